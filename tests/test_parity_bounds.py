"""CPU: the frozen bounds of the GPU suite stay within 3x of what the hardware showed.

profiles/r05_parity_observed.json is the record a full `pytest -m gpu` run leaves (tests.util.observe -> gpurun_out/parity_observed.json,
copied into profiles/ by the builder): for every floating-point comparison its worst observed value and the bound it was asserted
against.  Here: every "<=" bound is at most 3x its observed worst (plus 1e-6 of slack for values that are exactly zero), every
">=" count bound is at least the observed count minus one, and the named constants of tests/parity_bounds.py are the ones in the record."""
import json
import os

import pytest

from tests import parity_bounds as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = os.path.join(ROOT, "profiles", "r05_parity_observed.json")


@pytest.fixture(scope="module")
def rec():
    assert os.path.exists(REC), "profiles/r05_parity_observed.json missing: run the GPU suite and copy gpurun_out/parity_observed.json"
    return json.load(open(REC))


def test_every_bound_is_within_three_times_the_observed_worst(rec):
    assert len(rec) >= 30
    loose = []
    for key, r in rec.items():
        if r["kind"] == "<=":
            if r["bound"] > 3.0 * r["worst"] + 1e-6 and not key.startswith(("vit/encode_end_to_end_mean", "checkpoint/")):
                loose.append((key, r["worst"], r["bound"]))
            assert r["worst"] <= r["bound"], key
        else:
            assert r["bound"] >= r["worst"] - 1, (key, r)
            assert r["worst"] >= r["bound"], key
    # shared constants serve several comparisons: the constant is sized by the WORST of them, the others are then tighter than 3x by
    # construction - what must hold is that each constant is within 3x of the largest value observed under it
    by_bound = {}
    for key, r in rec.items():
        if r["kind"] == "<=":
            by_bound.setdefault(r["bound"], []).append(r["worst"])
    for bound, worsts in by_bound.items():
        if bound in (1e-2, 2e-2):           # whole-path margins that include the ViT's near-tie drift: stated in the tests, not kernel bounds
            continue
        assert bound <= 3.0 * max(worsts) + 1e-6, (bound, max(worsts))


def test_the_named_constants_are_the_recorded_ones(rec):
    want = {"llm/batched_logits_max_err_over_scale": P.LOGIT_TOL, "configs/llama7b_width_2_layers_logits_over_scale": P.LOGIT_TOL_WIDE,
            "configs/llama7b_full_depth_logits_over_scale": P.LOGIT_TOL_DEEP, "configs/vit_h_every_layer_teacher_forced_rel_l2": P.FEAT_TOL,
            "skinny_lds/structure_0_vs_1_logits_over_scale": P.STRUCTURE_TOL,
            "golden/g7_mid_flip_boundary_gap": P.NEAR_TIE, "golden/g7_mid_frame_layers_with_reference_indices": P.G7_MID_AGREE,
            "configs/vit_h_free_run_agreeing_frame_layers": P.FREE_RUN_AGREE, "configs/vit_h_free_run_mean_feature_drift": P.FREE_RUN_DRIFT}
    for key, const in want.items():
        assert key in rec, key
        assert rec[key]["bound"] == pytest.approx(const), (key, rec[key]["bound"], const)
