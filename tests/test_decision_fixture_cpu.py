"""The committed decision fixtures of tests/test_decision_replay_gpu.py are what the fp64 oracle produces (CPU): regenerated here
for configs[0] (14 s) and compared array by array; make_masks() itself asserts that the oracle's fp64 gradients EQUAL the
reference's fp64 goldens, i.e. that the recorded decisions are the reference's."""
import json
import os

import numpy as np

from tests import decision_replay as R


def test_decision_fixture_is_the_fp64_oracles(tmp_path):
    case = "cfg1_icgan_res64"
    out = str(tmp_path / "d.npz")
    R.make_masks(case, out)
    new, old = np.load(out), np.load(os.path.join(R.GOLDEN_DIR, "decisions_%s.npz" % case))
    assert sorted(new.files) == sorted(old.files)
    assert int(old["n"]) == 38 and int(old["npool"]) == 8
    for k in old.files:
        assert np.array_equal(new[k], old[k]), k
    shapes = json.loads(str(old["shapes"]))
    assert int(old["version"]) == 2 and old["digests"].shape == (38, 2)
    # sparse: only the elements inside the band are stored, with the digests of the full pattern beside them
    total, kept = sum(int(np.prod(s)) for s in shapes), sum(old["i%d" % i].size for i in range(38))
    assert 0 < kept < 1e-3 * total
    assert all(0 <= int(old["digests"][i][0]) <= int(np.prod(s)) for i, s in enumerate(shapes))
