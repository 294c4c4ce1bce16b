"""Reading the reference's StyleGAN2 network snapshots (`legacy.load_network_pkl`, legacy.py:28-77, and the
StyleGAN2 branch of inference/utils.py:395-403).  The fixture was pickled by the reference's own persistent classes the way
training_loop.py:613-640 writes `best-network-snapshot.pkl` (tests/golden/make_golden_sg2_snapshot.py; embedded source text
elided); the expectations (state_dict, init_kwargs, a truncated sample) come from the reference's own reader + generator."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from tests.stylegan_cases import SG2_NETS, sg2_inputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PKL = os.path.join(GOLD, "sg2_snapshot", "best-network-snapshot.pkl")


def _load(**kw):
    from ic_gan_amd.stylegan2 import legacy
    with open(PKL, "rb") as f:
        return legacy.load_network_pkl(f, **kw)


def test_snapshot_reader_rebuilds_networks_with_the_pickled_state():
    from ic_gan_amd.stylegan2 import networks
    g = np.load(os.path.join(GOLD, "sg2_snapshot.npz"))
    data = _load()
    assert set(data) == {"G", "D", "G_ema", "augment_pipe", "training_set_kwargs"}
    assert data["augment_pipe"] is None and data["training_set_kwargs"]["resolution"] == 16
    G_ema, D = data["G_ema"], data["D"]
    assert isinstance(G_ema, networks.Generator) and isinstance(D, networks.Discriminator) and isinstance(data["G"], networks.Generator)
    assert G_ema.training == bool(int(g["training"])) and not any(p.requires_grad for p in G_ema.parameters())
    assert list(G_ema.state_dict().keys()) == json.loads(str(g["names"]))
    for k, v in G_ema.state_dict().items():
        assert np.array_equal(v.numpy(), g["sd/" + k]), k
    for k, v in D.state_dict().items():
        assert np.array_equal(v.numpy(), g["sdD/" + k]), k
    want = json.loads(str(g["init_kwargs"]))
    got = json.loads(json.dumps(dict(G_ema._init_kwargs)))
    assert got == want
    # G and G_ema carry different weights in the fixture: the reader must not alias them
    assert not torch.equal(data["G"].state_dict()["mapping.fc0.weight"], G_ema.state_dict()["mapping.fc0.weight"])


def test_snapshot_reader_never_resolves_foreign_globals(tmp_path):
    """the embedded source is not executed, and a pickle that names any other global is refused"""
    from ic_gan_amd.stylegan2 import legacy

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "x"),))

    with pytest.raises(pickle.UnpicklingError, match="does not resolve"):
        legacy.load_network_pkl(pickle.dumps({"G_ema": Evil()}))
    assert not (tmp_path / "x").exists()
    with pytest.raises(pickle.UnpicklingError):
        legacy.load_network_pkl(pickle.dumps([1, 2, 3]))


def _stack_global_pickle(module, name, arg):
    """protocol-4 pickle of `module.name(arg)` built by hand: STACK_GLOBAL takes ANY (module, dotted-name) pair, which is
    how a prefix-based allow-list is bypassed (the name is resolved by attribute traversal from the module)."""
    enc = lambda t: pickle.SHORT_BINUNICODE + bytes([len(t.encode())]) + t.encode()
    return (pickle.PROTO + b"\x04" + enc(module) + enc(name) + pickle.STACK_GLOBAL + enc(arg) + pickle.TUPLE1 + pickle.REDUCE
            + pickle.STOP)


@pytest.mark.parametrize("module,name", [
    ("torch.serialization", "os.system"),            # dotted name under a formerly allowed module prefix
    ("torch._utils", "_import_dotted_name"),         # helper that imports whatever its argument names
    ("torch.serialization", "load"),                 # the unrestricted loader itself
    ("torch.storage", "os.system"),
    ("numpy.core.multiarray", "os.system"),
])
def test_snapshot_reader_refuses_allow_list_bypasses(tmp_path, module, name):
    from ic_gan_amd.stylegan2 import legacy
    marker = tmp_path / "pwned"
    blob = _stack_global_pickle(module, name, "touch %s" % marker)
    with pytest.raises(pickle.UnpicklingError, match="does not resolve"):
        legacy.load_network_pkl(blob)
    assert not marker.exists()


def test_snapshot_reader_storage_helper_uses_the_restricted_loader(tmp_path):
    """`torch.storage._load_from_bytes` (what plain-pickled tensors call for their storage) is torch.load(weights_only=False)
    in torch 2.x; the reader maps it to a weights_only=True load, so an inner payload naming a foreign global is refused too."""
    from ic_gan_amd.stylegan2 import legacy
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (os.system, ("touch %s" % marker,))

    inner = pickle.dumps(Evil(), protocol=2)
    enc = lambda t: pickle.SHORT_BINUNICODE + bytes([len(t.encode())]) + t.encode()
    blob = (pickle.PROTO + b"\x04" + enc("torch.storage") + enc("_load_from_bytes") + pickle.STACK_GLOBAL
            + pickle.BINBYTES + len(inner).to_bytes(4, "little") + inner + pickle.TUPLE1 + pickle.REDUCE + pickle.STOP)
    with pytest.raises(Exception):
        legacy.load_network_pkl(blob)
    assert not marker.exists()
    # ... while a genuine plain-pickled tensor still loads through the same route
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    back = legacy._SnapshotUnpickler(__import__("io").BytesIO(pickle.dumps({"t": t}))).load()["t"]
    assert torch.equal(back, t)


@pytest.mark.gpu
def test_snapshot_sample_matches_reference_generator():
    """load_model_inference(model_backbone='stylegan2') -> G_ema; inference.sample with truncation reproduces the image the
    reference generator produced from the same snapshot (rel. L2 < 1e-3: north_star's sample bound)."""
    from ic_gan_amd import inference
    g = np.load(os.path.join(GOLD, "sg2_snapshot.npz"))
    cfg = SG2_NETS["cc_ic_r16_resnetG"]
    config = {"model_backbone": "stylegan2", "base_root": GOLD, "experiment_name": "sg2_snapshot", "n_classes": cfg["G"]["c_dim"]}
    G_ema, _ = inference.load_model_inference(config, device="cuda")
    z, gc, gh, _, _, _ = sg2_inputs(cfg, 7, 1)
    labels = gc.argmax(1)
    img, c, h = inference.sample(G_ema, lambda: (z, labels, gh), config, class_cond=True, instance_cond=True, device="cuda",
                                 backbone="stylegan2", truncation_value=0.7)
    ref = torch.from_numpy(g["img"]).double()
    rel = float((img.double().cpu() - ref).norm() / ref.norm())
    assert img.shape == ref.shape and rel < 1e-3, rel
