#!/usr/bin/env python
"""A network snapshot written the way the reference's training loop writes it (stylegan2_ada_pytorch/training/
training_loop.py:613-640: `pickle.dump(dict(G, D, G_ema, augment_pipe, training_set_kwargs), f)` of the persistent-class
networks), plus what the reference's own reader + generator produce from it:

    tests/golden/sg2_snapshot/best-network-snapshot.pkl      the pickle (tiny 16x16 networks, class + instance conditioned)
    tests/golden/sg2_snapshot.npz                            state_dict of G_ema as read back by legacy.load_network_pkl,
                                                             its init_kwargs (JSON) and G_ema(z, c, feats, psi=0.7, const noise)

The pickle format embeds the SOURCE TEXT of training/networks.py in every persistent object (persistence.py:125-141); that
text is replaced by a one-line placeholder before dumping (ic_gan_amd's reader never executes it, and reference sources
must not be copied into this repository), so the reference's own loader is exercised on an un-elided in-memory copy.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sg2_snapshot.py"""
import copy
import io
import json
import os
import pickle
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/stylegan2_ada_pytorch")
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
from training import networks as ref_net
from torch_utils import persistence
sys.argv = sys.argv[:1]
import legacy as ref_legacy                      # stylegan2_ada_pytorch/legacy.py (imports click: present)

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.stylegan_cases import SG2_NETS, sg2_inputs, sg2_state   # noqa: E402


def load(m, seed):
    spec = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    cur = m.state_dict()
    m.load_state_dict({k: (cur[k] if v is None else v) for k, v in sg2_state(spec, seed).items()})


def main():
    cfg = SG2_NETS["cc_ic_r16_resnetG"]
    G = ref_net.Generator(**cfg["G"]).train().requires_grad_(False)
    D = ref_net.Discriminator(**cfg["D"]).train().requires_grad_(False)
    load(G, 1); load(D, 2)
    G_ema = copy.deepcopy(G).eval()
    load(G_ema, 3)
    snapshot = dict(training_set_kwargs=dict(class_name="training.dataset.ImageFolderDataset", resolution=16, use_labels=True))
    for name, module in [("G", G), ("D", D), ("G_ema", G_ema), ("augment_pipe", None)]:
        if module is not None:                   # training_loop.py:617-622
            module = copy.deepcopy(module).eval().requires_grad_(False).cpu()
        snapshot[name] = module
    # (1) the reference's reader on the genuine pickle (in memory)
    buf = io.BytesIO()
    pickle.dump(snapshot, buf)
    back = ref_legacy.load_network_pkl(io.BytesIO(buf.getvalue()))
    g = back["G_ema"]
    z, gc, gh, _, _, _ = sg2_inputs(cfg, 7, 1)
    with torch.no_grad():
        img = g(z=z, c=gc, feats=gh, truncation_psi=0.7, noise_mode="const")
    out = {"img": img.numpy(), "init_kwargs": json.dumps(dict(g.init_kwargs)), "training": np.array(int(g.training)),
           "names": json.dumps(list(g.state_dict().keys()))}
    for k, v in g.state_dict().items():
        out["sd/" + k] = v.numpy()
    for k, v in back["D"].state_dict().items():
        out["sdD/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "sg2_snapshot.npz"), **out)
    # (2) the committed fixture: same pickle with the embedded source text elided
    for obj in list(vars(ref_net).values()):
        if isinstance(obj, type) and persistence.is_persistent(obj):
            obj._orig_module_src = "# training/networks.py source text elided in this fixture (see make_golden_sg2_snapshot.py)\n"
    os.makedirs(os.path.join(HERE, "sg2_snapshot"), exist_ok=True)
    path = os.path.join(HERE, "sg2_snapshot", "best-network-snapshot.pkl")
    with open(path, "wb") as f:
        pickle.dump(snapshot, f)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", "image", tuple(img.shape))


if __name__ == "__main__":
    main()
