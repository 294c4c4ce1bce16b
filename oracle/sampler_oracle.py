"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's host conditioning sampler, one numpy RNG call per
reference RNG call, in the reference's order.  PINNED: tests/test_sampler_cpu.py checks it bit-for-bit against
tests/golden/sampler.npz, which holds outputs of the reference itself (tests/golden/make_golden_sampler.py).

Used to check `ic_gan_amd.data_utils.ConditioningStore` (which vectorises the draws and gathers on the device) at
sizes / neighbourhood shapes the golden file does not cover.  Never imported by the product.

Follows /root/reference/data_utils/datasets_common.py:
  fetch_features   647-679     instance_balance 525-576     nnclass_balance 578-622
"""
import numpy as np


def fetch_features(tab, index, in_mem=False, augmentation=False):
    """datasets_common.py:647-679 for a sequence of indices -> float32 [B, D] (after the caller's
    torch.FloatTensor cast, 574/619)."""
    if in_mem:                                                   # 652-653 with 421-427
        f = np.array(tab["feats"], copy=True)
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        return f[np.asarray(index)].astype(np.float32)
    rows = []
    for i in index:                                              # 663-677
        hflip = np.random.randint(2) == 1
        src = tab["feats_hflip"] if (augmentation and hflip) else tab["feats"]
        rows.append(src[i].astype("float")[np.newaxis, ...])
    feat = np.concatenate(rows)
    feat /= np.linalg.norm(feat, axis=1, keepdims=True)          # 678
    return feat.astype(np.float32)


def instance_balance(tab, batch_size, possible=None, weights=None, in_mem=False, augmentation=False, with_labels=True):
    """datasets_common.py:525-576 -> (labels int64 [B] or None, feats float32 [B, D], centres, neighbours)."""
    possible = np.arange(len(tab["labels"])) if possible is None else np.asarray(possible)
    if weights is None:
        sel = possible[np.random.randint(0, len(possible), size=batch_size)]          # 551-552
    else:
        sel = np.random.choice(possible, batch_size, replace=True, p=weights)        # 554-556
    feats = fetch_features(tab, sel, in_mem, augmentation)                            # 559
    chosen = [np.random.choice(tab["sample_nns"][i]) for i in sel]                    # 562-564
    labels = np.asarray([tab["labels"][c] for c in chosen], dtype=np.int64) if with_labels else None
    return labels, feats, np.asarray(sel), np.asarray(chosen)


def nnclass_balance(tab, batch_size, weights=None, num_classes=1000, in_mem=False, augmentation=False):
    """datasets_common.py:578-622 -> (labels int64 [B], feats float32 [B, D], instance indices)."""
    if weights is not None:
        weights = np.array(weights) / sum(weights)                                    # 603-604
    chosen_class = np.random.choice(range(num_classes), batch_size, replace=True, p=weights)   # 607-609
    nn_idxs = []
    for lab in chosen_class:
        x_nn = np.random.choice((tab["labels"] == lab).nonzero()[0])                  # 613
        nn_idxs.append(np.random.choice(tab["sample_nns"][x_nn]))                     # 615
    feats = fetch_features(tab, nn_idxs, in_mem, augmentation)                        # 617
    return np.asarray(chosen_class, dtype=np.int64), feats, np.asarray(nn_idxs)
