"""Host conditioning sampler: `ic_gan_amd.data_utils` against outputs of the reference's own
`ILSVRC_HDF5_feats` / `sample_conditioning_values` (tests/golden/sampler.npz, made by make_golden_sampler.py) and
against the loop restatement in oracle/sampler_oracle.py.  Bar: indices, labels AND feature values bit-exact."""
import numpy as np
import pytest
import torch

from ic_gan_amd import data_utils as DU
from oracle import sampler_oracle as SO
from tests import sampler_cases as SC
from tests.helpers import GOLDEN_DIR

import os

GOLD = np.load(os.path.join(GOLDEN_DIR, "sampler.npz"))
TAB = SC.make_table()


def _resolve(call):
    call = dict(call)
    for key in ("weights", "weights_sampling"):
        if call.get(key) == "instance":
            call[key] = SC.sampling_weights(SC.N)
        elif call.get(key) == "class":
            call[key] = SC.class_weights()
    return call


def _store(ctor, device=None):
    c = dict(ctor)
    with_nns = c.pop("with_nns", True)
    kmeans = SC.KMEANS if c.pop("kmeans", False) else None
    kw = dict(imgs=TAB["imgs"], labels=TAB["labels"], feats=TAB["feats"], feats_hflip=TAB["feats_hflip"],
              feature_dim=SC.D, k_nn=SC.K, kmeans_samples=kmeans, device=device)
    if with_nns:
        kw.update(sample_nns=TAB["sample_nns"], sample_nn_radius=TAB["sample_nn_radius"])
    kw.update(c)
    return DU.ConditioningStore(**kw)


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    assert a.dtype == b.dtype, "%s: dtype %s vs %s" % (what, a.dtype, b.dtype)
    assert np.array_equal(a, b), "%s differs (max |d| = %g)" % (what, np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


@pytest.mark.parametrize("name", sorted(SC.SAMPLER_CASES))
def test_sampler_matches_reference(name):
    case = SC.SAMPLER_CASES[name]
    np.random.seed(case["seed"])
    ds = _store(case["ctor"])
    _same(ds.possible_sampling_idxs, GOLD["%s/possible" % name], "possible_sampling_idxs")
    for c in range(case["ncalls"]):
        lab, feat = getattr(ds, case["method"])(**_resolve(case["call"]))
        key = "%s/%d/labels" % (name, c)
        if key in GOLD:
            _same(lab.numpy(), GOLD[key], key)
        else:
            assert lab is None
        _same(feat.numpy(), GOLD["%s/%d/feats" % (name, c)], "%s/%d/feats" % (name, c))


@pytest.mark.parametrize("name", sorted(SC.ITEM_CASES))
def test_getitem_matches_reference(name):
    case = SC.ITEM_CASES[name]
    np.random.seed(case["seed"])
    ds = _store(case["ctor"])
    for j, i in enumerate(case["indices"]):
        item = ds[i]
        item = item if isinstance(item, tuple) else (item,)
        n_ref = sum(1 for k in GOLD.files if k.startswith("%s/%d/" % (name, j)))
        assert len(item) == n_ref
        for t, v in enumerate(item):
            _same(v.numpy() if torch.is_tensor(v) else np.asarray(v), GOLD["%s/%d/%d" % (name, j, t)], "%s[%d][%d]" % (name, i, t))


@pytest.mark.parametrize("name", sorted(SC.SCV_CASES))
def test_sample_conditioning_values_matches_reference(name):
    case = SC.SCV_CASES[name]
    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"])
    zy = dict(case.get("zy", {}))
    if zy:
        zy["class_probabilities"] = SC.class_probabilities()
    z_, y_ = DU.prepare_z_y(SC.SCV_BATCH, SC.SCV_DIMZ, SC.NCLS, device="cpu", **zy)
    kw = _resolve(case["kw"])
    balance = kw.get("nn_sampling_strategy", "instance_balance")
    ds = _store(dict(which_nn_balance=balance))
    if balance == "nnclass_balance":
        kw["weights_sampling"] = list(kw["weights_sampling"]) + [0.0] * (1000 - SC.NCLS)
    for c in range(SC.SCV_NCALLS):
        res = DU.sample_conditioning_values(z_, y_, dataset=ds, batch_size=SC.SCV_BATCH, **kw)
        res = res if isinstance(res, tuple) else (res,)
        for t, v in enumerate(res):
            got = torch.Tensor(v).numpy() if v.dtype.is_floating_point else v.numpy()
            _same(got, GOLD["%s/%d/%d" % (name, c, t)], "%s/%d/%d" % (name, c, t))


def test_oracle_matches_reference():
    """pins oracle/sampler_oracle.py to the reference's outputs (instance_balance ± aug ± weights, nnclass ± weights)."""
    for name, aug, w in (("ib_hdf5", False, None), ("ib_hdf5_aug", True, None), ("ib_weights", False, SC.sampling_weights(SC.N))):
        case = SC.SAMPLER_CASES[name]
        np.random.seed(case["seed"])
        for c in range(case["ncalls"]):
            lab, feat, _, _ = SO.instance_balance(TAB, 16, weights=w, augmentation=aug)
            _same(lab, GOLD["%s/%d/labels" % (name, c)], name)
            _same(feat, GOLD["%s/%d/feats" % (name, c)], name)
    np.random.seed(SC.SAMPLER_CASES["ib_inmem"]["seed"])
    for c in range(2):
        lab, feat, _, _ = SO.instance_balance(TAB, 16, in_mem=True)
        _same(lab, GOLD["ib_inmem/%d/labels" % c], "ib_inmem")
        _same(feat, GOLD["ib_inmem/%d/feats" % c], "ib_inmem")
    for name, aug, w in (("ncb", False, None), ("ncb_weights_aug", True, SC.class_weights())):
        case = SC.SAMPLER_CASES[name]
        np.random.seed(case["seed"])
        for c in range(case["ncalls"]):
            lab, feat, _ = SO.nnclass_balance(TAB, 16, weights=w, num_classes=SC.NCLS, augmentation=aug)
            _same(lab, GOLD["%s/%d/labels" % (name, c)], name)
            _same(feat, GOLD["%s/%d/feats" % (name, c)], name)


def _ragged_table(seed, n, d, ncls):
    """neighbourhoods of different sizes (what `_obtain_nns` yields when the table holds duplicates)."""
    rs = np.random.RandomState(seed)
    feats = rs.randn(n, d).astype(np.float32)
    nns = [rs.choice(n, size=rs.randint(1, 9), replace=False).tolist() for _ in range(n)]
    return dict(feats=feats, feats_hflip=(feats + 0.1 * rs.randn(n, d)).astype(np.float32),
                labels=rs.permutation(np.arange(n) % ncls).astype(np.int64), sample_nns=nns,
                sample_nn_radius=rs.rand(n))


@pytest.mark.parametrize("seed,n,batch,aug", [(0, 50, 7, False), (1, 2000, 64, True), (2, 333, 1, True), (3, 4096, 256, False)])
def test_store_matches_oracle_ragged_and_large(seed, n, batch, aug):
    """sizes / neighbourhood shapes beyond the golden file: vectorised product vs per-sample oracle loop."""
    ncls = 13
    tab = _ragged_table(seed, n, 24, ncls)
    ds = DU.ConditioningStore(labels=tab["labels"], feats=tab["feats"], feats_hflip=tab["feats_hflip"],
                              sample_nns=tab["sample_nns"], sample_nn_radius=tab["sample_nn_radius"],
                              feature_dim=24, feature_augmentation=aug)
    for trial in range(3):
        np.random.seed(100 * seed + trial)
        lab, feat = ds.sample_conditioning_instance_balance(batch)
        np.random.seed(100 * seed + trial)
        lab_o, feat_o, _, _ = SO.instance_balance(tab, batch, augmentation=aug)
        _same(lab.numpy(), lab_o, "labels")
        _same(feat.numpy(), feat_o, "feats")
        np.random.seed(100 * seed + trial)
        lab, feat = ds.sample_conditioning_nnclass_balance(batch, num_classes=ncls)
        np.random.seed(100 * seed + trial)
        lab_o, feat_o, _ = SO.nnclass_balance(tab, batch, num_classes=ncls, augmentation=aug)
        _same(lab.numpy(), lab_o, "labels")
        _same(feat.numpy(), feat_o, "feats")
    # rectangular neighbourhoods take the fully vectorised branch
    rect = dict(tab, sample_nns=np.random.RandomState(seed).randint(0, n, size=(n, 5)).astype(np.int64))
    ds = DU.ConditioningStore(labels=rect["labels"], feats=rect["feats"], feats_hflip=rect["feats_hflip"],
                              sample_nns=rect["sample_nns"], sample_nn_radius=rect["sample_nn_radius"],
                              feature_dim=24, feature_augmentation=aug)
    np.random.seed(seed)
    lab, feat = ds.sample_conditioning_instance_balance(batch)
    np.random.seed(seed)
    lab_o, feat_o, _, _ = SO.instance_balance(rect, batch, augmentation=aug)
    _same(lab.numpy(), lab_o, "labels")
    _same(feat.numpy(), feat_o, "feats")


def test_empty_class_raises_like_numpy():
    tab = _ragged_table(5, 40, 8, 3)
    ds = DU.ConditioningStore(labels=tab["labels"], feats=tab["feats"], sample_nns=tab["sample_nns"],
                              sample_nn_radius=tab["sample_nn_radius"], feature_dim=8)
    np.random.seed(0)
    with pytest.raises(ValueError):
        ds.sample_conditioning_nnclass_balance(64, num_classes=6)      # classes 3..5 have no members
    with pytest.raises(ValueError):
        DU.ConditioningStore(labels=tab["labels"], feats=tab["feats"], feature_dim=8)   # no nns, feats not in memory


def test_knn_build_matches_reference_sets():
    """`_obtain_nns` of the reference (sklearn branch) gives unordered neighbour sets + the k-th distance."""
    nns, radius = DU.build_knn(DU._row_normalise(TAB["feats"], True), SC.K)
    got = np.sort(np.asarray(nns, dtype=np.int64), axis=1)
    _same(got, GOLD["knn/sets"], "kNN sets")
    np.testing.assert_allclose(radius, GOLD["knn/radius"], rtol=2e-4, atol=2e-6)   # fp32 Gram matrix vs fp64 sklearn
    # faiss ordering convention: ascending distance
    _same(np.asarray(nns, dtype=np.int64), TAB["sample_nns"], "kNN order")
    ds = _store(dict(load_in_mem_feats=True, with_nns=False))
    _same(np.asarray(ds.sample_nns, dtype=np.int64), TAB["sample_nns"], "store kNN")


def test_knn_duplicates_keep_k_plus_one():
    f = np.random.RandomState(0).randn(12, 6).astype(np.float32)
    f[1:6] = f[0]                                               # six identical rows, k=3: self may not be listed
    nns, radius = DU.build_knn(f, 3)
    assert all(i not in row for i, row in enumerate(nns))
    assert all(len(row) == 3 for row in nns)                    # self is forced to rank 0, so always removed
    assert radius[0] == 0.0


def test_dataloader_weights_match_reference():
    spc = np.bincount(TAB["labels"], minlength=SC.NCLS).tolist()
    _same(np.asarray(DU.make_weights_for_balanced_classes(spc, TAB["labels"], SC.NCLS)), GOLD["weights/balanced"], "balanced")
    _same(np.asarray(DU.make_weights_for_balanced_classes(spc, TAB["labels"], SC.NCLS, True, 2.0,
                                                          class_probabilities=SC.class_probabilities())),
          GOLD["weights/temperature"], "temperature")


@pytest.mark.gpu
def test_store_resident_on_device_matches_reference():
    """feature table + labels resident in HBM, gathers on the device: same bits as the reference's host path."""
    for name in ("ib_hdf5_aug", "ib_inmem", "ncb_weights_aug", "ib_kmeans"):
        case = SC.SAMPLER_CASES[name]
        np.random.seed(case["seed"])
        ds = _store(case["ctor"], device="cuda:0")
        for c in range(case["ncalls"]):
            lab, feat = getattr(ds, case["method"])(**_resolve(case["call"]))
            assert feat.is_cuda and lab.is_cuda
            _same(lab.cpu().numpy(), GOLD["%s/%d/labels" % (name, c)], name)
            _same(feat.cpu().numpy(), GOLD["%s/%d/feats" % (name, c)], name)
    nns, radius = DU.build_knn(DU._row_normalise(TAB["feats"], True), SC.K, device="cuda:0")
    _same(np.sort(np.asarray(nns, dtype=np.int64), axis=1), GOLD["knn/sets"], "kNN sets (device)")
    np.testing.assert_allclose(radius, GOLD["knn/radius"], rtol=2e-4, atol=2e-6)
