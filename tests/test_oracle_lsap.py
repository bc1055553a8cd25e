"""Pin oracle/lsap.c (restated scipy `_lsap`) against known-answer vectors produced by scipy
(tests/golden/lsap_cases.npz, made by tests/golden/make_golden.py) and against scipy live."""
import os

import numpy as np
import pytest

from oracle import owl_oracle as O
from owl_vit_object_detection_amd import rng


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(os.path.join(golden_dir, "lsap_cases.npz"))


def test_golden_known_answers(cases):
    keys = sorted({k.split("/")[0] for k in cases.files})
    assert len(keys) >= 13
    for k in keys:
        c = cases[k + "/cost"]
        i, j = O.linear_sum_assignment(c)
        assert np.array_equal(i, cases[k + "/row"]), k
        if k.startswith("c"):                      # tie-free: indices are unique
            assert np.array_equal(j, cases[k + "/col"]), k
        assert c[i, j].sum() == pytest.approx(c[cases[k + "/row"], cases[k + "/col"]].sum(), abs=1e-12)


def test_against_scipy_live():
    sp = pytest.importorskip("scipy.optimize")
    for t, (nr, nc) in enumerate([(2304, 16), (16, 2304), (50, 50), (1, 9), (9, 1), (577, 90)]):
        c = rng.uniform(99, f"live/{t}", nr * nc).reshape(nr, nc) - 0.5
        i, j = O.linear_sum_assignment(c)
        si, sj = sp.linear_sum_assignment(c)
        assert np.array_equal(i, si) and np.array_equal(j, sj)
    # tie-heavy integer matrices: same indices thanks to the same tie rule
    for t, (nr, nc) in enumerate([(30, 30), (200, 12), (12, 200)]):
        c = rng.randint(99, f"liveint/{t}", nr * nc, 3).reshape(nr, nc).astype(np.float64)
        i, j = O.linear_sum_assignment(c)
        si, sj = sp.linear_sum_assignment(c)
        assert c[i, j].sum() == c[si, sj].sum()
        assert np.array_equal(i, si) and np.array_equal(j, sj)


def test_edge_cases():
    i, j = O.linear_sum_assignment(np.zeros((0, 5)))
    assert len(i) == 0 and len(j) == 0
    with pytest.raises(ValueError):
        O.linear_sum_assignment(np.array([[np.nan, 1.0], [1.0, 2.0]]))
    with pytest.raises(ValueError):
        O.linear_sum_assignment(np.array([[np.inf, np.inf], [1.0, 2.0]]))
