"""CPU: the densify / prune oracle (oracle/densify_oracle.py) against golden vectors produced by the reference's own
GaussianModel.densify_and_prune (tests/golden/densify.npz, G7 of tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import densify_oracle as O

NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "gaussian_feats"]
GOLD = os.path.join(os.path.dirname(__file__), "golden", "densify.npz")


def load_case(d, tag):
    params = {n: d[f"{tag}_in_{n}"] for n in NAMES}
    moments = {n: (d[f"{tag}_in_{n}_m"], d[f"{tag}_in_{n}_v"]) for n in NAMES}
    return params, moments


@pytest.mark.parametrize("tag,size_threshold", [("a", 20), ("b", None)])
def test_oracle_matches_reference_densify_and_prune(tag, size_threshold):
    d = np.load(GOLD)
    params, moments = load_case(d, tag)
    p, m, nc, ns = O.densify_and_prune(params, moments, d[f"{tag}_in_accum"], d[f"{tag}_in_denom"], float(d["percent_dense"]),
                                       float(d["extent"]), float(d["max_grad"]), float(d["min_opacity"]), size_threshold,
                                       d[f"{tag}_z"])
    assert nc == int(d[f"{tag}_num_clone"]) and ns == int(d[f"{tag}_num_split"]) and nc > 0 and ns > 0
    for n in NAMES:
        want = d[f"{tag}_out_{n}"]
        assert p[n].shape == want.shape, n
        if n in ("xyz", "scaling"):      # children rows carry float arithmetic (rotation, exp/log): 1e-6 relative
            np.testing.assert_allclose(p[n], want, rtol=2e-6, atol=2e-6)
        else:
            assert np.array_equal(p[n], want), n
        assert np.array_equal(m[n][0], d[f"{tag}_out_{n}_m"]) and np.array_equal(m[n][1], d[f"{tag}_out_{n}_v"]), n
    # something was pruned, cloned and split, and the screen-size variant prunes more
    assert want.shape[0] != params["xyz"].shape[0]


def test_stats_oracle_small_case():
    acc, den, mr = np.zeros((5, 1), np.float32), np.zeros((5, 1), np.float32), np.array([0, 9, 2, 0, 0], np.float32)
    g = np.array([[3, 4, 7], [1, 0, 0], [0, 0, 0], [6, 8, 1], [5, 12, 0]], np.float32)
    radii = np.array([2, 3, 0, 7, -1], np.int32)
    O.add_densification_stats(acc, den, mr, g, radii)
    assert acc[:, 0].tolist() == [5.0, 1.0, 0.0, 10.0, 0.0]
    assert den[:, 0].tolist() == [1.0, 1.0, 0.0, 1.0, 0.0]
    assert mr.tolist() == [2.0, 9.0, 2.0, 7.0, 0.0]
