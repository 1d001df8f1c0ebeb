"""CPU: the frozen bounds of the GPU suite stay within 3x of what the hardware showed - PER COMPARISON.

profiles/<tag>_parity_observed.json is the merged record of full `pytest -m gpu` runs (tests.util.observe -> gpurun_out/parity_observed.json
per run; tools/merge_parity_runs.py): for every floating-point comparison its worst observed value over the runs and the bound it was
asserted against.  Here, with NO exemption by name: every comparison's bound (the tighter of its shared constant and its own entry in
tests/parity_bounds.PER_COMPARISON) is at most 3x ITS OWN observed worst, every ">=" count bound is the observed count or one less, the
round-6 record comes from at least three runs, and the table in tests/parity_bounds.py is the one the record was asserted against."""
import json
import os

import pytest

from tests import parity_bounds as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record():
    for tag in ("r06", "r05"):
        p = os.path.join(ROOT, "profiles", f"{tag}_parity_observed.json")
        if os.path.exists(p):
            return tag, json.load(open(p))
    raise AssertionError("profiles/r0x_parity_observed.json missing: run the GPU suite and merge its records (tools/merge_parity_runs.py)")


@pytest.fixture(scope="module")
def rec():
    return _record()


def test_every_comparison_is_within_three_times_its_own_observation(rec):
    tag, r = rec
    keys = [k for k in r if not k.startswith("_")]
    assert len(keys) >= 30
    loose = []
    for key in keys:
        o = r[key]
        bound = P.bound_for(key, o.get("shared", o["bound"]), o["kind"] == ">=")      # what observe() asserts TODAY for this comparison
        if o["kind"] == "<=":
            assert o["worst"] <= bound, (key, o["worst"], bound)
            if bound > 3.0 * o["worst"] + 1e-9:
                loose.append((key, o["worst"], bound))
        else:
            assert o["worst"] >= bound, (key, o)
            if bound < o["worst"] - 1:
                loose.append((key, o["worst"], bound))
    assert not loose, loose
    if tag == "r06":
        assert len(r["_runs"]) >= 3, r["_runs"]
        few = [k for k in keys if r[k]["runs"] < 3]
        assert not few, few


def test_every_table_entry_was_recorded(rec):
    _, r = rec
    missing = [k for k in P.PER_COMPARISON if k not in r]
    assert not missing, missing                                   # an entry nobody records any more is a bound nobody checks


def test_the_named_constants_are_the_shared_ones_of_the_record(rec):
    _, r = rec
    want = {"llm/batched_logits_max_err_over_scale": P.LOGIT_TOL, "configs/llama7b_width_2_layers_logits_over_scale": P.LOGIT_TOL_WIDE,
            "configs/llama7b_full_depth_logits_over_scale": P.LOGIT_TOL_DEEP, "configs/vit_h_every_layer_teacher_forced_rel_l2": P.FEAT_TOL,
            "skinny_lds/structure_0_vs_1_logits_over_scale": P.STRUCTURE_TOL,
            "golden/g7_mid_flip_boundary_gap": P.NEAR_TIE, "golden/g7_mid_frame_layers_with_reference_indices": P.G7_MID_AGREE,
            "configs/vit_h_free_run_agreeing_frame_layers": P.FREE_RUN_AGREE, "configs/vit_h_free_run_mean_feature_drift": P.FREE_RUN_DRIFT}
    for key, const in want.items():
        assert key in r, key
        assert r[key].get("shared", r[key]["bound"]) == pytest.approx(const), (key, r[key], const)
