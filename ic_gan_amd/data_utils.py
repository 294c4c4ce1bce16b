"""Host conditioning sampler of the IC-GAN step, with the feature table resident in HBM  (SURVEY §8f N2).

Mirrors the part of the reference's data layer that sits either side of the training step:

  reference                                                   here
  ---------------------------------------------------------   -------------------------------------------
  data_utils/datasets_common.py:240-821  ILSVRC_HDF5_feats    ConditioningStore
      sample_conditioning_instance_balance   (525-576)            .sample_conditioning_instance_balance
      sample_conditioning_nnclass_balance    (578-622)            .sample_conditioning_nnclass_balance
      get_label / get_instance_features      (624-679)            .get_label / .get_instance_features
      _get_instance_features_and_nn          (780-818)            ._get_instance_features_and_nn
      __getitem__ / __len__                  (476-523,820)        .__getitem__ / .__len__
      _obtain_nns / _get_kth_value_accurate  (695-769)            build_knn (exact L2 top-(k+1), drop self)
  data_utils/utils.py:830-901   sample_conditioning_values    sample_conditioning_values
  data_utils/utils.py:905-966   prepare_z_y                   prepare_z_y
  data_utils/utils.py:978-1021  Distribution                  Distribution
  data_utils/utils.py:227-300   make_weights_for_balanced_classes   make_weights_for_balanced_classes

What is different by design (MI355X-first):

* The reference opens the HDF5 file and reads + L2-normalises ONE feature row per sample, per step
  (datasets_common.py:655-679) — at 8 × 130+ img/s that host loop is the bottleneck either side of the
  hot path.  Here the feature table (and its hflip twin) is normalised once at construction and kept
  resident on the device (1.28 M × 2048 fp32 = 10.5 GB of the 288 GB HBM); a step's conditioning batch is
  one row gather on the device.  Labels live on the device too.
* Only the *index* draws stay on the host, and they consume numpy's legacy global RandomState in exactly
  the reference's order, so the chosen indices are bit-identical to the reference's under the same seed
  (tests/test_sampler_cpu.py pins this against outputs of the reference itself).  The per-sample python
  loops of the reference are replaced by vectorised draws that produce the same stream:
  `randint(0, n, size=B)` is the same stream as B × `randint(0, n)`, and `choice(a)` is `a[randint(0, len(a))]`.
* `nnclass_balance` uses a class → indices CSR built once instead of scanning all labels per sample
  (datasets_common.py:611 scans 1.28 M labels per drawn sample).

Normalisation arithmetic is the reference's, so that feature values are bit-identical as well:
HDF5 path = fp64 row / fp64 `sqrt(sum(x*x))`, rounded to fp32 by `torch.FloatTensor(...)` (datasets_common.py:678,574);
in-memory path = fp32 in-place division (datasets_common.py:425).
"""
from __future__ import annotations

import numpy as np
import torch

__all__ = [
    "ConditioningStore",
    "Distribution",
    "prepare_z_y",
    "sample_conditioning_values",
    "make_weights_for_balanced_classes",
    "build_knn",
]


# ------------------------------------------------------------------------------------------------------
# kNN build
# ------------------------------------------------------------------------------------------------------
def build_knn(feats, k_nn, device=None, block=4096):
    """Exact L2 k-NN of every row of `feats` [N, D] against the table itself.

    Follows datasets_common.py:695-746: search k+1 neighbours, remove the query itself, keep the distance to
    the (k+1)-th hit as the radius.  Rows are returned sorted by distance (the faiss convention,
    datasets_common.py:726-731); ties resolve to the lower index.  If the query is not among its own k+1
    nearest (exact duplicates), the row keeps k+1 entries exactly like `np.delete` does in the reference,
    hence the list-of-lists return type.

    The N×N distance matrix is produced block-wise as one GEMM per block on `device`
    (|a-b|² = |a|² + |b|² - 2 a·b, features are unit-norm in the reference so this is well conditioned).  On the GPU both
    the GEMM and the top-(k+1) selection are the hand-written kernels of csrc/knn.hip (C-ABI icg_knn_l2).

    Returns (sample_nns: list[list[int]] of length N, radius: float64 ndarray [N]).
    """
    f = torch.as_tensor(feats, dtype=torch.float32)
    if device is not None:
        f = f.to(device)
    N = f.shape[0]
    k = min(int(k_nn) + 1, N)
    if f.is_cuda and k <= 64:
        # hand-written path (csrc/knn.hip): inner products on the fp32 MFMA GEMM + one wavefront per query row keeping a
        # lane-sorted running top-(k+1); replaces faiss-gpu's IndexFlatL2.search (datasets_common.py:720-731)
        from . import _lib as L
        f = f.contiguous()
        nn_idx = torch.empty(N, k, dtype=torch.int64, device=f.device)
        nn_d2 = torch.empty(N, k, dtype=torch.float32, device=f.device)
        nb = L.query("icg_knn_l2_workspace_bytes", N, f.shape[1])
        L.call("icg_knn_l2", f, N, f.shape[1], k, nn_idx, nn_d2, torch.empty(nb, dtype=torch.uint8, device=f.device), nb)
    else:
        # host tables (the reference's own placement: north_star keeps the sampler on the host) and k + 1 > 64: library ops
        sq = (f * f).sum(1)
        nn_idx = torch.empty(N, k, dtype=torch.int64, device=f.device)
        nn_d2 = torch.empty(N, k, dtype=torch.float32, device=f.device)
        for s in range(0, N, block):
            e = min(N, s + block)
            d2 = sq[s:e, None] + sq[None, :] - 2.0 * (f[s:e] @ f.t())
            d2.clamp_(min=0)
            # the query is its own nearest hit by construction, not by rounding luck
            d2[torch.arange(e - s, device=f.device), torch.arange(s, e, device=f.device)] = -1.0
            v, i = torch.topk(d2, k, dim=1, largest=False, sorted=True)
            nn_idx[s:e], nn_d2[s:e] = i, v.clamp_(min=0)
    idx = nn_idx.cpu().numpy()
    radius = np.sqrt(nn_d2[:, -1].double().cpu().numpy())
    keep = idx != np.arange(N)[:, None]
    sample_nns = [idx[i][keep[i]].tolist() for i in range(N)]
    return sample_nns, radius


# ------------------------------------------------------------------------------------------------------
# the store
# ------------------------------------------------------------------------------------------------------
def _row_normalise(feats, in_mem):
    """The two normalisations of the reference (see module docstring); returns float32 [N, D]."""
    if in_mem:                                          # datasets_common.py:421-427
        f = np.array(feats, copy=True)
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        return np.ascontiguousarray(f, dtype=np.float32)
    out = np.empty(feats.shape, dtype=np.float32)       # datasets_common.py:666-678, chunked
    for s in range(0, feats.shape[0], 65536):
        blk = np.asarray(feats[s:s + 65536]).astype("float")
        blk /= np.linalg.norm(blk, axis=1, keepdims=True)
        out[s:s + 65536] = blk
    return out


class ConditioningStore(torch.utils.data.Dataset):
    """In-memory / in-HBM equivalent of `ILSVRC_HDF5_feats` (datasets_common.py:240-821).

    Arrays replace the HDF5 files (`root`, `root_feats`, `root_nns`); `from_hdf5` reads the reference's files
    when `h5py` is importable.  Constructor arguments keep the reference's names and meaning
    (datasets_common.py:354-378).  `device` is where the normalised feature table and the labels live
    (None = host; the sampler then returns host tensors exactly like the reference).
    """

    def __init__(
        self,
        imgs=None,
        labels=None,
        feats=None,
        feats_hflip=None,
        sample_nns=None,
        sample_nn_radius=None,
        transform=None,
        target_transform=None,
        load_labels=True,
        load_features=True,
        load_in_mem_feats=False,
        k_nn=4,
        which_nn_balance="instance_balance",
        kmeans_samples=None,
        n_subsampled_data=-1,
        label_dim=0,
        feature_dim=2048,
        feature_augmentation=False,
        apply_norm=True,
        label_onehot=False,
        device=None,
        **_ignored,
    ):
        if labels is None and imgs is None:
            raise ValueError("ConditioningStore needs at least `labels` or `imgs` to know the dataset size")
        self.data = imgs
        self.labels = None if labels is None else np.asarray(labels)
        self.load_labels = bool(load_labels) and labels is not None
        self.load_features = bool(load_features)
        self.load_in_mem_feats = bool(load_in_mem_feats)
        self._label_dim, self._feature_dim = label_dim, feature_dim
        self.label_onehot = label_onehot
        self.feature_augmentation = feature_augmentation
        self.transform, self.target_transform = transform, target_transform
        self.apply_norm = apply_norm
        self.which_nn_balance = which_nn_balance
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.num_imgs = len(labels) if labels is not None else len(imgs)

        self._raw = self._raw_hflip = None          # host rows for the per-item path
        self.feats = self._table = None
        if self.load_features:
            if feats is None:
                raise ValueError("load_features=True needs `feats`")
            self._raw = feats
            self._raw_hflip = feats_hflip if feature_augmentation else None
            if self.load_in_mem_feats:
                # the reference never reads feats_hflip on this branch (datasets_common.py:652-653)
                tab = _row_normalise(feats, True)
                self.feats = torch.from_numpy(tab)                     # reference attribute (host)
                self._n_tab = 1
            else:
                tab = _row_normalise(feats, False)
                if self._raw_hflip is not None:
                    tab = np.concatenate([tab, _row_normalise(self._raw_hflip, False)], 0)
                    self._n_tab = 2
                else:
                    self._n_tab = 1
            self._table = torch.from_numpy(tab).to(self.device)       # [n_tab*N, D] resident

            if sample_nns is None:
                if not self.load_in_mem_feats:
                    raise ValueError(
                        "If no pre-computed neighborhoods are provided, the features need to be loaded in "
                        "memory to extract them. Set load_in_mem_feats=True.")      # datasets_common.py:438-443
                sample_nns, sample_nn_radius = build_knn(self.feats, k_nn, device=self.device)
            self.sample_nns = sample_nns
            self.sample_nn_radius = np.asarray(sample_nn_radius)
            self._nn_rect = self._rectangular(sample_nns)

        self._labels_dev = None
        if self.load_labels:
            self._labels_dev = torch.from_numpy(self.labels.astype(np.int64).reshape(self.num_imgs, -1)).to(self.device)
        self._class_csr = None

        # datasets_common.py:445-468
        self.possible_sampling_idxs = range(self.num_imgs)
        self.kmeans_samples = None
        if kmeans_samples is not None:
            self.kmeans_samples = np.asarray(kmeans_samples)
            self.possible_sampling_idxs = self.kmeans_samples
        elif n_subsampled_data > -1:
            self.possible_sampling_idxs = np.random.choice(
                np.array(self.possible_sampling_idxs), int(n_subsampled_data), replace=False)
        self.possible_sampling_idxs = np.array(self.possible_sampling_idxs)
        self.num_imgs = len(self.possible_sampling_idxs)

    # ---- construction helpers -----------------------------------------------------------------------
    @classmethod
    def from_hdf5(cls, root, root_feats=None, root_nns=None, kmeans_file=None, **kw):
        """Read the reference's HDF5 layout (`imgs`,`labels` / `feats`,`feats_hflip` / `sample_nns`,
        `sample_nns_radius`; datasets_common.py:405-437) into arrays.  Needs h5py."""
        try:
            import h5py as h5
        except ImportError as e:                                        # pragma: no cover
            raise RuntimeError("ConditioningStore.from_hdf5 needs h5py; pass arrays to the constructor instead") from e
        with h5.File(root, "r") as f:
            kw["labels"] = f["labels"][:]
            if kw.pop("load_in_mem_images", False):
                kw["imgs"] = f["imgs"][:]
        if root_feats is not None:
            with h5.File(root_feats, "r") as f:
                kw["feats"] = f["feats"][:]
                if kw.get("feature_augmentation") and "feats_hflip" in f:
                    kw["feats_hflip"] = f["feats_hflip"][:]
        if root_nns is not None:
            with h5.File(root_nns, "r") as f:
                kw["sample_nns"] = f["sample_nns"][:]
                kw["sample_nn_radius"] = f["sample_nns_radius"][:]
        if kmeans_file is not None:
            kw["kmeans_samples"] = np.load(kmeans_file, allow_pickle=True).item()["center_examples"][:, 0]
        return cls(**kw)

    @staticmethod
    def _rectangular(sample_nns):
        """[N, k] int64 array when every neighbourhood has the same size (always true for the HDF5 files),
        else None (list-of-lists from a kNN build with duplicates)."""
        if isinstance(sample_nns, np.ndarray) and sample_nns.ndim == 2:
            return sample_nns.astype(np.int64, copy=False)
        k0 = len(sample_nns[0])
        if all(len(r) == k0 for r in sample_nns):
            return np.asarray(sample_nns, dtype=np.int64).reshape(len(sample_nns), k0)
        return None

    # ---- neighbour draw -----------------------------------------------------------------------------
    def _draw_neighbours(self, centres):
        """`np.random.choice(self.sample_nns[i])` for each i in `centres`, same RNG stream
        (datasets_common.py:560-562)."""
        centres = np.asarray(centres, dtype=np.int64)
        if self._nn_rect is not None:
            k = self._nn_rect.shape[1]
            return self._nn_rect[centres, np.random.randint(0, k, size=len(centres))]
        return np.asarray([self.sample_nns[i][np.random.randint(0, len(self.sample_nns[i]))] for i in centres],
                          dtype=np.int64)

    # ---- reference API ------------------------------------------------------------------------------
    def sample_conditioning_instance_balance(self, batch_size, weights=None):
        """datasets_common.py:525-576: h ~ p(h) (uniform or `weights`), then a neighbour's label.
        Returns (labels_gen int64 [B(,label_dim)] or None, instance_gen float32 [B, D]) on `self.device`."""
        if weights is None:
            sel = np.random.randint(0, len(self.possible_sampling_idxs), size=batch_size)
            sel = self.possible_sampling_idxs[sel]
        else:
            sel = np.random.choice(self.possible_sampling_idxs, batch_size, replace=True, p=weights)
        instance_gen = self._gather_features(sel)           # draws the hflip coins first, like 554
        chosen = self._draw_neighbours(sel)
        labels_gen = self._gather_labels(chosen) if self.load_labels else None
        return labels_gen, instance_gen

    def sample_conditioning_nnclass_balance(self, batch_size, weights=None, num_classes=1000):
        """datasets_common.py:578-622: y ~ p(y), x_nn ~ p(x|y), h ~ p(h | x_nn)."""
        if weights is not None:
            weights = np.array(weights) / sum(weights)
        chosen_class = np.random.choice(range(num_classes), batch_size, replace=True, p=weights)
        starts, members = self._classes()
        nn_idxs = np.empty(batch_size, dtype=np.int64)
        for j, lab in enumerate(chosen_class):
            lo, hi = starts[lab], starts[lab + 1]
            if hi == lo:
                raise ValueError("'a' cannot be empty unless no samples are taken")     # numpy's own message
            x_nn = members[lo + np.random.randint(0, hi - lo)]
            row = self.sample_nns[x_nn]
            nn_idxs[j] = row[np.random.randint(0, len(row))]
        instance_gen = self._gather_features(nn_idxs)
        labels_gen = torch.from_numpy(np.asarray(chosen_class, dtype=np.int64)).to(self.device)
        return labels_gen, instance_gen

    def get_label(self, index):
        """datasets_common.py:624-645."""
        if not self.load_labels:
            return np.zeros(self.label_dim, dtype=np.float32) if self.label_onehot else 0
        target = self.labels[index]
        if self.label_onehot:
            onehot = np.zeros(self.label_dim, dtype=np.float32)
            onehot[target] = 1
            target = onehot
        return target

    def get_instance_features(self, index):
        """datasets_common.py:647-679, host path (DataLoader workers / single rows).  Same return types as
        the reference: a float32 torch tensor on the in-memory branch, a float64 ndarray otherwise."""
        if not self.load_features:
            return np.zeros(self.feature_dim, dtype=np.float32)
        if self.load_in_mem_feats:
            return self.feats[index].clone().float()
        if isinstance(index, (int, np.int64)):
            hflip = np.random.randint(2) == 1
            src = self._raw_hflip if (self.feature_augmentation and hflip) else self._raw
            feat = np.asarray(src[index]).astype("float")
            feat /= np.linalg.norm(feat, keepdims=True)
            return feat
        flips = np.random.randint(0, 2, size=len(index)) == 1
        rows = [np.asarray((self._raw_hflip if (self.feature_augmentation and h) else self._raw)[i]).astype("float")
                for i, h in zip(index, flips)]
        feat = np.stack(rows)
        feat /= np.linalg.norm(feat, axis=1, keepdims=True)
        return feat

    def _get_image(self, index):
        return self.data[index]

    def _get_instance_features_and_nn(self, index):
        """datasets_common.py:780-818."""
        if self.which_nn_balance == "instance_balance":
            idx_h = index
            if self.kmeans_samples is not None:
                index = np.random.choice(self.kmeans_samples)
            idx_nn = np.random.choice(self.sample_nns[index])
        elif self.which_nn_balance == "nnclass_balance":
            idx_h = np.random.choice(self.sample_nns[index])
            idx_nn = index
        else:
            raise ValueError("which_nn_balance must be instance_balance or nnclass_balance")
        radii = self.sample_nn_radius[idx_h]
        return self._get_image(idx_nn), self.get_label(idx_nn), self.get_instance_features(idx_h), radii

    def __getitem__(self, index):
        """datasets_common.py:476-523."""
        index = self.possible_sampling_idxs[index]
        img = self._get_image(index)
        target = self.get_label(index)
        if self.load_features:
            img, target, feats, radii = self._get_instance_features_and_nn(index)
        img = torch.from_numpy(np.asarray(img))
        if self.apply_norm:
            img = ((img.float() / 255) - 0.5) * 2
        if self.transform is not None:
            img = self.transform(img)
        if self.target_transform is not None:
            target = self.target_transform(target)
        if not self.label_onehot:
            target = int(target)
        if self.load_features and self.load_labels:
            return img, target, feats, radii
        if self.load_features:
            return img, feats, radii
        if self.load_labels:
            return img, target
        return img

    def __len__(self):
        return self.num_imgs

    @property
    def resolution(self):
        return list(self.data[0].shape)[1]

    @property
    def label_dim(self):
        return self._label_dim

    @property
    def feature_dim(self):
        return self._feature_dim

    # ---- device gathers -----------------------------------------------------------------------------
    def _gather_features(self, idx):
        """Rows of the resident table for a batch of indices; consumes one hflip coin per sample on the
        HDF5 branch whether or not augmentation is on (datasets_common.py:664-666)."""
        idx = np.asarray(idx, dtype=np.int64)
        if not self.load_features:
            return torch.zeros(self.feature_dim, dtype=torch.float32, device=self.device)
        if not self.load_in_mem_feats:
            flips = np.random.randint(0, 2, size=len(idx))
            if self._n_tab == 2:
                idx = idx + flips.astype(np.int64) * (self._table.shape[0] // 2)
        sel = torch.from_numpy(idx)
        if self.device.type != "cpu":
            sel = sel.pin_memory().to(self.device, non_blocking=True)
        return self._table.index_select(0, sel)

    def _gather_labels(self, idx):
        sel = torch.from_numpy(np.asarray(idx, dtype=np.int64))
        if self.device.type != "cpu":
            sel = sel.pin_memory().to(self.device, non_blocking=True)
        out = self._labels_dev.index_select(0, sel)
        # reference: concatenate(get_label(i)[np.newaxis]) -> [B] for scalar labels, [B, L] otherwise
        return out[:, 0] if self.labels.ndim == 1 else out.view(len(idx), *self.labels.shape[1:])

    def _classes(self):
        """class -> ascending member indices, as CSR; `(labels == c).nonzero()[0]` without the scan."""
        if self._class_csr is None:
            lab = self.labels.reshape(-1).astype(np.int64)
            order = np.argsort(lab, kind="stable")
            n_cls = int(lab.max()) + 1 if lab.size else 0
            counts = np.bincount(lab, minlength=n_cls)
            starts = np.concatenate([[0], np.cumsum(counts)])
            self._class_csr = (_PaddedStarts(starts), order)
        return self._class_csr


class _PaddedStarts:
    """CSR offsets that answer `[c]`/`[c+1]` for classes beyond the largest label present (empty class)."""

    def __init__(self, starts):
        self._s = starts

    def __getitem__(self, i):
        return int(self._s[min(int(i), len(self._s) - 1)])


# ------------------------------------------------------------------------------------------------------
# noise / label distributions and the step-level sampler
# ------------------------------------------------------------------------------------------------------
class Distribution(torch.Tensor):
    """A tensor that knows how to resample itself in place (data_utils/utils.py:978-1021).

    `init_distribution("normal", mean=, var=)`, `("categorical", num_categories=)`,
    `("categorical_longtail", num_categories=, class_prob=)`,
    `("categorical_longtail_temperature", num_categories=, temperature=, class_prob=)`.
    NB like the reference, "var" is handed to `normal_` as the standard deviation.
    """

    def init_distribution(self, dist_type, class_prob=None, **kwargs):
        self.dist_type, self.dist_kwargs = dist_type, kwargs
        if dist_type == "normal":
            self.mean, self.var = kwargs["mean"], kwargs["var"]
            return
        if not dist_type.startswith("categorical"):
            raise ValueError("unknown distribution %r" % (dist_type,))
        self.num_categories = kwargs["num_categories"]
        if dist_type == "categorical_longtail":
            self.class_prob = torch.DoubleTensor(class_prob)
        elif dist_type == "categorical_longtail_temperature":
            logp = torch.log(torch.DoubleTensor(class_prob)) / kwargs["temperature"]
            self.class_prob = torch.exp(logp) / torch.sum(torch.exp(logp))

    def sample_(self):
        if self.dist_type == "normal":
            self.normal_(self.mean, self.var)
        elif self.dist_type == "categorical":
            self.random_(0, self.num_categories)
        else:
            self.data = torch.multinomial(self.class_prob, len(self), replacement=True).to(self.device)


def prepare_z_y(G_batch_size, dim_z, nclasses, device="cuda", fp16=False, z_var=1.0, longtail_gen=False,
                custom_distrib=False, longtail_temperature=1, class_probabilities=None):
    """data_utils/utils.py:905-966.  Like the reference, both distributions stay on the host (the `.to(device)`
    lines are commented out there) so that draws come from torch's CPU generator."""
    if fp16:
        raise NotImplementedError("fp16 noise is not part of the fp32 hot path")
    z_ = Distribution(torch.randn(G_batch_size, dim_z, requires_grad=False))
    z_.init_distribution("normal", mean=0, var=z_var)
    y_ = Distribution(torch.zeros(G_batch_size, requires_grad=False))
    if longtail_gen:
        y_.init_distribution("categorical_longtail", num_categories=nclasses, class_prob=class_probabilities)
    elif custom_distrib:
        y_.init_distribution("categorical_longtail_temperature", num_categories=nclasses,
                             temperature=longtail_temperature, class_prob=class_probabilities)
    else:
        y_.init_distribution("categorical", num_categories=nclasses)
    return z_, y_


def sample_conditioning_values(z_, y_, ddp=False, batch_size=1, weights_sampling=None, dataset=None,
                               constant_conditioning=False, class_cond=True, instance_cond=False,
                               nn_sampling_strategy="instance_balance"):
    """data_utils/utils.py:830-901: one draw of (z[, labels][, features]) for the generator."""
    with torch.no_grad():
        z_.sample_()
        if not instance_cond:
            if not class_cond:
                return z_
            y_.sample_()
            if constant_conditioning:
                return z_, torch.zeros_like(y_)
            return (z_, y_) if ddp else (z_, y_.data.clone())
        if nn_sampling_strategy == "instance_balance":
            draw = dataset.sample_conditioning_instance_balance
        elif nn_sampling_strategy == "nnclass_balance":
            draw = dataset.sample_conditioning_nnclass_balance
        else:
            raise ValueError("nn_sampling_strategy must be instance_balance or nnclass_balance")
        labels_g, f_g = draw(batch_size, weights_sampling)
        return (z_, labels_g, f_g) if class_cond else (z_, f_g)


def make_weights_for_balanced_classes(samples_per_class, labels=None, nclasses=None, custom_distrib_gen=False,
                                      longtail_temperature=1, class_probabilities=None):
    """data_utils/utils.py:227-287: per-sample weight p(x|y)·w(y) for the DataLoader's weighted sampler, with
    w(y) = N / count(y) (class balancing) or the temperature-softened class distribution.  Returns a list of
    python floats of length len(labels) (fp64 arithmetic, as in the reference's python loop)."""
    lab = np.asarray(labels).reshape(-1).astype(np.int64)
    if custom_distrib_gen:
        logp = torch.log(torch.DoubleTensor(class_probabilities)) / longtail_temperature
        per_class = (torch.exp(logp) / torch.sum(torch.exp(logp))).numpy()
    else:
        count = np.bincount(lab, minlength=nclasses)[:nclasses].astype(np.float64)
        with np.errstate(divide="raise"):
            per_class = float(count.sum()) / count          # ZeroDivisionError in the reference for an empty class
    spc = np.asarray(samples_per_class, dtype=np.float64)
    return ((1.0 / spc[lab]) * per_class[lab]).tolist()
