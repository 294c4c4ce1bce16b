"""Reader of the reference's StyleGAN2 network snapshots (`best-network-snapshot.pkl`, `last_net.pkl`).

Replaces `legacy.load_network_pkl` (stylegan2_ada_pytorch/legacy.py:28-77) for the pickles the IC-GAN training loop writes
(training/training_loop.py:613-640: `pickle.dump(dict(G, D, G_ema, augment_pipe, training_set_kwargs), f)`).

Format.  Every network class of training/networks.py is a `persistence.persistent_class`
(torch_utils/persistence.py:42-141): an instance pickles as a call of `persistence._reconstruct_persistent_obj(meta)` with
`meta = {type: 'class', version, module_src: <source text of training/networks.py>, class_name, state: obj.__dict__}`, and
the reference's loader `exec`s that source text to rebuild the object (persistence.py:198-219, 239-249).  Here the pickled
source is NEVER executed: each persistent object becomes a stub holding (class_name, state); the top-level networks are then
rebuilt as `ic_gan_amd.stylegan2.networks.<class_name>(*state._init_args, **state._init_kwargs)` (same constructor
signatures) and every parameter / buffer of the stub tree is copied in by name (`load_state_dict(strict=True)`: the two
implementations share the state_dict layout, tests/test_stylegan2.py).  The unpickler resolves only an EXACT
(module, name) allow-list (the four torch tensor / parameter rebuild helpers, numpy array / dtype / scalar reconstruction,
OrderedDict, EasyDict, the persistence hook), refuses dotted names, and routes the storage helper of plain-pickled tensors
(`torch.storage._load_from_bytes`) through `torch.load(..., weights_only=True)`; tests/test_sg2_snapshot.py feeds it
pickles that name `os.system` three different ways and checks that each is refused before anything is called.

Not supported (raise): TensorFlow-era pickles (`dnnlib.tflib.network.Network`, legacy.py:31-41,80-89: conversion of the
original TF StyleGAN2 weights, which IC-GAN never writes); `force_fp16` is honoured through the networks' own
`num_fp16_res` / `conv_clamp` arguments.
"""
from __future__ import annotations

import copy
import io
import pickle

import numpy as np
import torch


class EasyDict(dict):
    """dnnlib.util.EasyDict: attribute access to dict entries."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


class _PersistentStub:
    """A pickled persistent object: its class name and `__dict__` (children are stubs themselves)."""

    def __init__(self, meta):
        if meta.get("type") != "class":
            raise pickle.UnpicklingError("unsupported persistent object type %r" % (meta.get("type"),))
        self.class_name = meta["class_name"]
        self.state = dict(meta["state"] or {})

    def named_tensors(self, prefix=""):
        """(name, tensor) of every parameter and persistent buffer of the module tree, state_dict order."""
        st = self.state
        skip = set(st.get("_non_persistent_buffers_set", ()))
        for group in ("_parameters", "_buffers"):
            for k, v in (st.get(group) or {}).items():
                if v is not None and not (group == "_buffers" and k in skip):
                    yield prefix + k, v
        for k, child in (st.get("_modules") or {}).items():
            if isinstance(child, _PersistentStub):
                yield from child.named_tensors(prefix + k + ".")
            elif isinstance(child, torch.nn.Module):
                for n, t in child.state_dict().items():
                    yield prefix + k + "." + n, t


def _reconstruct_persistent_obj(meta):
    return _PersistentStub(meta)


class _TFNetworkStub(dict):
    pass


def _load_storage_from_bytes(b):
    """torch.storage._load_from_bytes is `torch.load(io.BytesIO(b), weights_only=False)` (torch 2.x): the helper plain-pickled
    tensors use for their storage would hand the inner payload to the unrestricted unpickler.  Same bytes, restricted loader."""
    return torch.load(io.BytesIO(b), weights_only=True)


def _exact(module, names):
    mod = __import__(module, fromlist=["_"])
    return {(module, n): getattr(mod, n) for n in names if hasattr(mod, n)}


# EXACT (module, name) pairs only.  No prefix rules: protocol-4 STACK_GLOBAL accepts dotted names, so a rule such as
# "anything under torch.serialization." also resolves ("torch.serialization", "os.system") by attribute traversal.
_ALLOWED = {
    ("torch_utils.persistence", "_reconstruct_persistent_obj"): _reconstruct_persistent_obj,
    ("dnnlib.util", "EasyDict"): EasyDict,
    ("dnnlib", "EasyDict"): EasyDict,
    ("dnnlib.tflib.network", "Network"): _TFNetworkStub,
    ("collections", "OrderedDict"): __import__("collections").OrderedDict,
    ("builtins", "set"): set, ("builtins", "frozenset"): frozenset, ("builtins", "dict"): dict, ("builtins", "list"): list,
    ("builtins", "tuple"): tuple, ("builtins", "slice"): slice, ("builtins", "complex"): complex,
    ("torch.storage", "_load_from_bytes"): _load_storage_from_bytes,
}
_ALLOWED.update(_exact("torch._utils", ("_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_parameter",
                                        "_rebuild_parameter_with_state")))
_ALLOWED.update(_exact("torch", ("FloatStorage", "HalfStorage", "DoubleStorage", "LongStorage", "IntStorage", "BoolStorage",
                                 "ByteStorage", "Size", "device", "float32", "float16", "float64", "int64", "int32", "bool",
                                 "uint8", "Tensor")))
_ALLOWED.update(_exact("numpy", ("ndarray", "dtype")))
for _m in ("numpy.core.multiarray", "numpy._core.multiarray"):
    try:
        _ALLOWED.update(_exact(_m, ("_reconstruct", "scalar")))
    except ImportError:
        pass


class _SnapshotUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        # exact pairs, resolved from the table above (never by importing what the pickle names); dotted names are refused
        # outright -- they are attribute traversals (protocol >= 4), not globals
        if "." not in name and (module, name) in _ALLOWED:
            return _ALLOWED[(module, name)]
        raise pickle.UnpicklingError("network snapshot references %s.%s, which this reader does not resolve" % (module, name))


def _build(stub, force_fp16=False, device="cpu"):
    """stub of a top-level network -> ic_gan_amd.stylegan2.networks instance with the pickled parameters and buffers."""
    from . import networks
    cls = getattr(networks, stub.class_name, None)
    if cls is None or not isinstance(cls, type) or not issubclass(cls, torch.nn.Module):
        raise pickle.UnpicklingError("no ic_gan_amd.stylegan2.networks class named %r" % (stub.class_name,))
    args = copy.deepcopy(tuple(stub.state.get("_init_args", ())))
    kwargs = EasyDict(copy.deepcopy(dict(stub.state.get("_init_kwargs", {}))))
    if force_fp16:                                  # legacy.py:60-76
        if stub.class_name == "Generator":
            kwargs.synthesis_kwargs = EasyDict(kwargs.get("synthesis_kwargs", {}))
            kwargs.synthesis_kwargs.num_fp16_res = 4
            kwargs.synthesis_kwargs.conv_clamp = 256
        elif stub.class_name == "Discriminator":
            kwargs.num_fp16_res = 4
            kwargs.conv_clamp = 256
    net = cls(*args, **kwargs)
    net.load_state_dict({k: (v.data if isinstance(v, torch.nn.Parameter) else v) for k, v in stub.named_tensors()}, strict=True)
    net._init_args, net._init_kwargs = args, kwargs          # `init_kwargs` of the reference's persistent classes
    training = stub.state.get("training", True)
    net.train(bool(training))
    for p in net.parameters():                               # snapshots store requires_grad_(False) copies (training_loop.py:617-621)
        p.requires_grad_(False)
    return net.to(device)


def load_network_pkl(f, force_fp16=False, device="cpu"):
    """-> dict(G, D, G_ema, training_set_kwargs, augment_pipe) like legacy.load_network_pkl (legacy.py:28-77); `f` is a binary
    file object or bytes.  Networks come back as ic_gan_amd.stylegan2.networks modules in the mode they were saved in."""
    if isinstance(f, (bytes, bytearray)):
        f = io.BytesIO(f)
    data = _SnapshotUnpickler(f).load()
    if isinstance(data, tuple) and len(data) == 3 and all(isinstance(n, _TFNetworkStub) for n in data):
        raise NotImplementedError("TensorFlow-era StyleGAN2 pickles (legacy.py:31-41) are not supported: IC-GAN snapshots are "
                                  "PyTorch persistent-class pickles")
    if not isinstance(data, dict):
        raise pickle.UnpicklingError("not a network snapshot: expected a dict with G / D / G_ema")
    out = dict(data)
    for key in ("G", "D", "G_ema"):
        if isinstance(out.get(key), _PersistentStub):
            out[key] = _build(out[key], force_fp16, device)
    if isinstance(out.get("augment_pipe"), _PersistentStub):
        # ADA is out of scope (every shipped IC-GAN StyleGAN2 config trains with aug = noaug): keep only its buffers
        out["augment_pipe"] = {k: v for k, v in out["augment_pipe"].named_tensors()}
    out.setdefault("training_set_kwargs", None)               # legacy.py:43-47
    out.setdefault("augment_pipe", None)
    assert isinstance(out["G_ema"], torch.nn.Module)          # legacy.py:52
    return out
