#!/usr/bin/env python
"""Generates tests/golden/sampler.npz by running the REFERENCE's own host sampler
(`/root/reference/data_utils/datasets_common.py::ILSVRC_HDF5_feats`, `data_utils/utils.py::
sample_conditioning_values / prepare_z_y / Distribution / make_weights_for_balanced_classes`) unmodified over the
synthetic tables of tests/sampler_cases.py.

Runs only in the build container (needs /root/reference).  `h5py`, `torchvision`, `PIL` are absent from the
image, so they are replaced by import stubs; `h5py.File` is replaced by an in-memory stand-in that serves the
synthetic arrays under the dataset names the reference reads (`imgs`, `labels`, `feats`, `feats_hflip`,
`sample_nns`, `sample_nns_radius`) and returns copies like h5py does.  No reference source is copied.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sampler.py
"""
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import sampler_cases as SC  # noqa: E402


# ---- import stubs ------------------------------------------------------------------------------------
class _Meta(type):
    def __getattr__(cls, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _mk(k)


def _mk(name):
    return _Meta(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _mk(k)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ("torchvision", "h5py", "PIL", "pycocotools", "skimage")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, m):
        pass


sys.meta_path.insert(0, _Finder())
sys.path[:0] = ["/root/reference"]
import data_utils.datasets_common as dc  # noqa: E402
import data_utils.utils as du  # noqa: E402


class _DS:
    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def __len__(self):
        return len(self.a)

    def __getitem__(self, i):
        return np.array(self.a[i])


_FILES = {}


class _File(dict):
    def __init__(self, path, mode="r"):
        super().__init__({k: _DS(v) for k, v in _FILES[path].items()})

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


dc.h5.File = _File


def reference_dataset(tab, ctor, tmpdir):
    c = dict(ctor)
    with_nns = c.pop("with_nns", True)
    if c.pop("kmeans", False):
        path = os.path.join(tmpdir, "kmeans.npy")
        np.save(path, {"center_examples": SC.KMEANS[:, None]}, allow_pickle=True)
        c["kmeans_file"] = path
    _FILES["root"] = dict(imgs=tab["imgs"], labels=tab["labels"])
    _FILES["feats"] = dict(feats=tab["feats"], feats_hflip=tab["feats_hflip"])
    _FILES["nns"] = dict(sample_nns=tab["sample_nns"], sample_nns_radius=tab["sample_nn_radius"])
    kw = dict(root="root", root_feats="feats", root_nns="nns" if with_nns else None, load_in_mem_images=True,
              load_in_mem_labels=True, feature_dim=SC.D, k_nn=SC.K, gpu_knn=False)
    kw.update(c)
    return dc.ILSVRC_HDF5_feats(**kw)


def resolve_weights(call):
    call = dict(call)
    for key in ("weights", "weights_sampling"):
        if call.get(key) == "instance":
            call[key] = SC.sampling_weights(SC.N)
        elif call.get(key) == "class":
            call[key] = SC.class_weights()
    return call


def main():
    tab = SC.make_table()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, case in SC.SAMPLER_CASES.items():
            np.random.seed(case["seed"])
            ds = reference_dataset(tab, case["ctor"], tmp)
            for c in range(case["ncalls"]):
                lab, feat = getattr(ds, case["method"])(**resolve_weights(case["call"]))
                if lab is not None:
                    out["%s/%d/labels" % (name, c)] = lab.numpy()
                out["%s/%d/feats" % (name, c)] = feat.numpy()
            out["%s/possible" % name] = np.asarray(ds.possible_sampling_idxs)

        for name, case in SC.ITEM_CASES.items():
            np.random.seed(case["seed"])
            ds = reference_dataset(tab, case["ctor"], tmp)
            for j, i in enumerate(case["indices"]):
                item = ds[i]
                item = item if isinstance(item, tuple) else (item,)
                for t, v in enumerate(item):
                    out["%s/%d/%d" % (name, j, t)] = v.numpy() if torch.is_tensor(v) else np.asarray(v)

        for name, case in SC.SCV_CASES.items():
            torch.manual_seed(case["seed"])
            np.random.seed(case["seed"])
            zy = dict(case.get("zy", {}))
            if zy:
                zy["class_probabilities"] = SC.class_probabilities()
            z_, y_ = du.prepare_z_y(SC.SCV_BATCH, SC.SCV_DIMZ, SC.NCLS, device="cpu", **zy)
            kw = resolve_weights(case["kw"])
            balance = kw.get("nn_sampling_strategy", "instance_balance")
            ds = reference_dataset(tab, dict(which_nn_balance=balance), tmp)
            # nnclass_balance takes num_classes=1000 by default through this entry point; give every class up to
            # NCLS a weight and none beyond, as the reference's long-tail configs do with 1000 classes
            if balance == "nnclass_balance":
                kw["weights_sampling"] = list(kw["weights_sampling"]) + [0.0] * (1000 - SC.NCLS)
            for c in range(SC.SCV_NCALLS):
                res = du.sample_conditioning_values(z_, y_, dataset=ds, batch_size=SC.SCV_BATCH, **kw)
                res = res if isinstance(res, tuple) else (res,)
                for t, v in enumerate(res):
                    out["%s/%d/%d" % (name, c, t)] = torch.Tensor(v).numpy().copy() if v.dtype.is_floating_point else v.numpy().copy()

        # kNN build of the reference (sklearn path: faiss is absent) — neighbour sets and radii
        np.random.seed(0)
        ds = reference_dataset(tab, dict(load_in_mem_feats=True, with_nns=False), tmp)
        out["knn/sets"] = np.sort(np.asarray(ds.sample_nns, dtype=np.int64), axis=1)
        out["knn/radius"] = np.asarray(ds.sample_nn_radius)

        # DataLoader weights
        spc = np.bincount(tab["labels"], minlength=SC.NCLS).tolist()
        out["weights/balanced"] = np.asarray(du.make_weights_for_balanced_classes(spc, tab["labels"], SC.NCLS), dtype=np.float64)
        out["weights/temperature"] = np.asarray(
            [float(w) for w in du.make_weights_for_balanced_classes(
                spc, tab["labels"], SC.NCLS, True, 2.0, class_probabilities=SC.class_probabilities())], dtype=np.float64)

    path = os.path.join(HERE, "sampler.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
