"""CPU: the FEATURE-state head oracle (oracle/feature_head_oracle.py) against golden vectors from the imported reference
(tests/golden/feature_head.npz, G8 of tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import feature_head_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "feature_head.npz")


def test_sampler_weights_and_matrices_match_reference():
    d = np.load(GOLD)
    sam = torch.from_numpy(d["sam_masks"])
    torch.manual_seed(int(d["sampler_seed"]))
    sp, sm = O.sample_pixel_and_mask(sam, int(d["num_sampled_pixels"]), int(d["num_sampled_masks"]))
    assert torch.equal(sp, torch.from_numpy(d["sampled_pixel"])) and torch.equal(sm, torch.from_numpy(d["sampled_mask"]))
    assert int(sp.sum()) > 100 and 0 < int(sm.sum()) < sam.shape[0]
    assert np.array_equal(O.correspondence_matrix(sam, sp, sm).numpy(), d["C"])
    assert np.array_equal(O.pixel_weights(sam, sp).numpy(), d["weights"])
    np.testing.assert_allclose(O.feature_matrix(torch.from_numpy(d["features"]), sp).numpy(), d["C_F"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("mode,use_w", [("soft", True), ("all", True), ("hard", True), ("soft", False)])
def test_head_losses_and_gradients_match_reference(mode, use_w):
    d = np.load(GOLD)
    sam, sp, sm = torch.from_numpy(d["sam_masks"]), torch.from_numpy(d["sampled_pixel"]), torch.from_numpy(d["sampled_mask"])
    f = torch.from_numpy(d["features"]).requires_grad_(True)
    lp, ln, ps, ns = O.head(f, sam, sp, sm, mode, float(d["positive_th"]), float(d["negative_th"]), use_w)
    tag = mode + ("" if use_w else "_noweights")
    assert abs(float(lp.detach()) - float(d[f"{tag}_loss_pos"])) < 1e-6 and abs(float(ln.detach()) - float(d[f"{tag}_loss_neg"])) < 1e-6
    assert abs(float(ps) - float(d["pos_similarity"])) < 1e-6 and abs(float(ns) - float(d["neg_similarity"])) < 1e-6
    (lp + ln).backward()
    np.testing.assert_allclose(f.grad.numpy(), d[f"{tag}_grad"], rtol=1e-5, atol=1e-8)


def test_feature_norm_reg_matches_reference():
    d = np.load(GOLD)
    f = torch.from_numpy(d["features"]).requires_grad_(True)
    r = O.feature_norm_reg(f)
    r.backward()
    assert abs(float(r) - float(d["reg"])) < 1e-5 * float(d["reg"])
    np.testing.assert_allclose(f.grad.numpy(), d["reg_grad"], rtol=1e-5, atol=1e-9)
