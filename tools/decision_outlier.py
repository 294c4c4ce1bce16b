"""attribute the isolated 5.2e-3 sample of `G blocks.2.0.conv1.weight` in the cfg3 decision replay
(tests/test_decision_replay_gpu.py).  Replays the cfg3_w96_r256_b16 step with the fp64 decisions imposed under several kernel routes and
prints, for the worst tensors, the worst sample: its position, the reference value in units of the tensor rms, the error in units of
the tensor rms and RELATIVE TO THE SAMPLE ITSELF.
    python tools/decision_outlier.py [case]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def detail(case, samples, top=6):
    import tests.helpers as H
    g64 = np.load(os.path.join(H.GOLDEN_DIR, "biggan_%s_f64.npz" % case), allow_pickle=False)
    rows = []
    rms_of = lambda r: np.sqrt((r ** 2).sum() / max(int(np.count_nonzero(r)), 1))
    top_rms = max(rms_of(r) for pf in ("step1/G_grad/", "step1/D_grad/") for r in g64[pf + "samp"])
    for tag in "GD":
        prefix = "step1/%s_grad/" % tag
        for i, n in enumerate(json.loads(str(g64[prefix + "names"]))):
            r64 = g64[prefix + "samp"][i]
            nz = max(int(np.count_nonzero(r64)), 1)
            rms = np.sqrt((r64 ** 2).sum() / nz)
            if (tag, n) not in samples or rms < 1e-6 * top_rms:      # (mathematically zero gradients: a bias feeding a BatchNorm)
                continue
            e = samples[(tag, n)] - r64
            j = int(np.abs(e).argmax())
            rows.append((float(np.abs(e).max() / rms), tag + " " + n, j, float(r64[j] / rms), float(e[j] / rms), float(abs(e[j]) / max(abs(r64[j]), 1e-300)),
                         float(np.sqrt((e ** 2).sum() / nz) / rms), float(np.sort(np.abs(e))[-2] / rms)))
    rows.sort(reverse=True)
    for r in rows[:top]:
        print("    %-40s max %.2e at sample %4d: ref = %+8.2f rms, err = %+.2e rms = %.2e of the sample | tensor rms err %.2e, 2nd largest %.2e" % (
            r[1], r[0], r[2], r[3], r[4], r[5], r[6], r[7]))
    return rows


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "cfg3_w96_r256_b16"
    from tests import decision_replay as R
    from ic_gan_amd import ops
    big = 10 ** 9
    saved = dict(rs=ops.RS_WINOGRAD_MIN_CHANNELS, w=ops.WINOGRAD_MIN_CHANNELS, w2=ops.WINOGRAD2_MIN_CHANNELS, w4=ops.WINOGRAD4_MIN_CHANNELS,
                 wg=ops.WINOGRAD4_WGRAD_MIN_CHANNELS)

    def restore():
        ops.RS_WINOGRAD_MIN_CHANNELS, ops.WINOGRAD_MIN_CHANNELS, ops.WINOGRAD2_MIN_CHANNELS = saved["rs"], saved["w"], saved["w2"]
        ops.WINOGRAD4_MIN_CHANNELS, ops.WINOGRAD4_WGRAD_MIN_CHANNELS = saved["w4"], saved["wg"]

    def only_rs_off():          # the resample-fused layers (GBlock conv1, DBlock conv2) leave the 25-plane domain; stride-1 layers stay in F(4x4,3x3)
        restore()
        ops.RS_WINOGRAD_MIN_CHANNELS = {True: (big, big, big), False: (big, big, big)}

    def only_wgrad_off():       # weight gradients leave the Winograd domain, forward / data gradients stay
        restore()
        ops.WINOGRAD4_WGRAD_MIN_CHANNELS = big
        ops.RS_WINOGRAD_MIN_CHANNELS = {k: (v[0], v[1], big) for k, v in saved["rs"].items()}

    routes = [("default routes", lambda: None)]
    routes.append(("resample-fused layers off the 25-plane domain (stride-1 layers stay in F(4x4,3x3))", only_rs_off))
    routes.append(("weight gradients off the Winograd domain (forward / data gradients stay)", only_wgrad_off))
    routes.append(("no Winograd anywhere (implicit-GEMM / phase kernels only)", lambda: (restore(), ops.disable_winograd())))
    for name, setup in routes:
        setup()
        samples, census, pools = R.hip_step(case, R.Nudger)
        print("%s: %d flipped ReLU signs, %d max-pool winners imposed" % (name, sum(c[2] for c in census), sum(c[1] for c in pools)), flush=True)
        detail(case, samples)


if __name__ == "__main__":
    main()
