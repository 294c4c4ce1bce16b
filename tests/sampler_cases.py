"""Synthetic conditioning tables and the case list shared by tests/golden/make_golden_sampler.py (which runs the
REFERENCE's `ILSVRC_HDF5_feats` over them, in the build container) and tests/test_sampler_cpu.py (which runs
`ic_gan_amd.data_utils.ConditioningStore` and `oracle.sampler_oracle` over them, anywhere)."""
import numpy as np

N, D, K, NCLS, RES = 600, 48, 6, 10, 4


def make_table(seed=1234, n=N, d=D, k=K, ncls=NCLS, res=RES):
    """Deterministic table: clustered features (so that neighbourhoods are meaningful), their 'hflip' twins,
    labels, tiny images and a rectangular exact-kNN index with radii (fp64 brute force, self removed)."""
    rs = np.random.RandomState(seed)
    centres = rs.randn(ncls * 3, d).astype(np.float32)
    assign = rs.randint(0, ncls * 3, size=n)
    feats = (centres[assign] + 0.35 * rs.randn(n, d)).astype(np.float32) * rs.uniform(0.5, 3.0, size=(n, 1)).astype(np.float32)
    feats_hflip = (feats + 0.05 * rs.randn(n, d)).astype(np.float32)
    labels = (assign % ncls).astype(np.int64)
    noisy = rs.rand(n) < 0.15
    labels[noisy] = rs.randint(0, ncls, size=int(noisy.sum()))
    imgs = rs.randint(0, 256, size=(n, 3, res, res)).astype(np.uint8)
    fn = feats.astype(np.float64)
    fn /= np.linalg.norm(fn, axis=1, keepdims=True)
    d2 = ((fn[:, None, :] - fn[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, -1.0)
    order = np.argsort(d2, axis=1, kind="stable")[:, : k + 1]
    sample_nns = order[:, 1:].astype(np.int64)
    radius = np.sqrt(np.maximum(np.take_along_axis(d2, order[:, -1:], 1)[:, 0], 0))
    return dict(imgs=imgs, labels=labels, feats=feats, feats_hflip=feats_hflip, sample_nns=sample_nns,
                sample_nn_radius=radius)


def sampling_weights(n, seed=7):
    w = np.random.RandomState(seed).gamma(0.7, size=n)
    return w / w.sum()


KMEANS = np.random.RandomState(99).choice(N, 40, replace=False).astype(np.int64)

# name -> dict(ctor=kwargs for the dataset, method, call=kwargs, seed, ncalls)
SAMPLER_CASES = {
    "ib_hdf5": dict(ctor=dict(), method="sample_conditioning_instance_balance", call=dict(batch_size=16), seed=3, ncalls=3),
    "ib_hdf5_aug": dict(ctor=dict(feature_augmentation=True), method="sample_conditioning_instance_balance",
                        call=dict(batch_size=16), seed=4, ncalls=3),
    "ib_weights": dict(ctor=dict(), method="sample_conditioning_instance_balance",
                       call=dict(batch_size=16, weights="instance"), seed=5, ncalls=2),
    "ib_nolabels": dict(ctor=dict(load_labels=False), method="sample_conditioning_instance_balance",
                        call=dict(batch_size=8), seed=6, ncalls=2),
    "ib_inmem": dict(ctor=dict(load_in_mem_feats=True, with_nns=True), method="sample_conditioning_instance_balance",
                     call=dict(batch_size=16), seed=7, ncalls=2),
    "ib_subsampled": dict(ctor=dict(n_subsampled_data=100), method="sample_conditioning_instance_balance",
                          call=dict(batch_size=16), seed=8, ncalls=2),
    "ib_kmeans": dict(ctor=dict(kmeans=True), method="sample_conditioning_instance_balance",
                      call=dict(batch_size=16), seed=9, ncalls=2),
    "ncb": dict(ctor=dict(which_nn_balance="nnclass_balance"), method="sample_conditioning_nnclass_balance",
                call=dict(batch_size=16, num_classes=NCLS), seed=10, ncalls=3),
    "ncb_weights_aug": dict(ctor=dict(which_nn_balance="nnclass_balance", feature_augmentation=True),
                            method="sample_conditioning_nnclass_balance",
                            call=dict(batch_size=16, weights="class", num_classes=NCLS), seed=11, ncalls=2),
}

# __getitem__ sequences: name -> dict(ctor, seed, indices)
ITEM_CASES = {
    "item_ib": dict(ctor=dict(), seed=20, indices=[0, 5, 17, 599, 123]),
    "item_ib_aug": dict(ctor=dict(feature_augmentation=True), seed=21, indices=[1, 2, 3, 4, 400]),
    "item_ncb": dict(ctor=dict(which_nn_balance="nnclass_balance"), seed=22, indices=[10, 11, 12, 300]),
    "item_kmeans": dict(ctor=dict(kmeans=True), seed=23, indices=[0, 1, 39]),
    "item_nofeat": dict(ctor=dict(load_features=False), seed=24, indices=[7, 8]),
    "item_onehot": dict(ctor=dict(label_onehot=True, label_dim=NCLS), seed=25, indices=[7, 8, 9]),
}

# sample_conditioning_values: name -> dict(kwargs, seed, ncalls)
SCV_CASES = {
    "scv_cc_ic": dict(kw=dict(class_cond=True, instance_cond=True), seed=30),
    "scv_ic": dict(kw=dict(class_cond=False, instance_cond=True), seed=31),
    "scv_ic_ncb": dict(kw=dict(class_cond=True, instance_cond=True, nn_sampling_strategy="nnclass_balance",
                               weights_sampling="class"), seed=32),
    "scv_cc": dict(kw=dict(class_cond=True, instance_cond=False), seed=33),
    "scv_cc_const": dict(kw=dict(class_cond=True, instance_cond=False, constant_conditioning=True), seed=34),
    "scv_none": dict(kw=dict(class_cond=False, instance_cond=False), seed=35),
    "scv_longtail": dict(kw=dict(class_cond=True, instance_cond=False), seed=36, zy=dict(longtail_gen=True)),
    "scv_longtail_T": dict(kw=dict(class_cond=True, instance_cond=False), seed=37,
                           zy=dict(custom_distrib=True, longtail_temperature=2.0)),
}
SCV_BATCH, SCV_DIMZ, SCV_NCALLS = 12, 20, 2


def class_probabilities(ncls=NCLS):
    p = np.arange(1, ncls + 1, dtype=np.float64) ** -1.2
    return (p / p.sum()).tolist()


def class_weights(ncls=NCLS):
    return (np.arange(1, ncls + 1, dtype=np.float64) ** 0.5).tolist()
