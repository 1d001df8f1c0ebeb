#!/usr/bin/env python3
"""Merge the parity records of several FULL `pytest -m gpu` runs (each run leaves gpurun_out/parity_observed.json; copy it to
profiles/<tag>_parity_runs/run<i>.json) into profiles/<tag>_parity_observed.json: per comparison the worst value over all runs, the
number of runs that recorded it, every run's own worst, and the bound it was asserted against in the LAST run.

    python tools/merge_parity_runs.py r06        # reads profiles/r06_parity_runs/*.json, writes profiles/r06_parity_observed.json
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def merge(tag: str) -> dict:
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_parity_runs", "*.json")))
    if not files:
        raise SystemExit(f"no run records under profiles/{tag}_parity_runs/")
    out = {}
    for f in files:
        for key, r in json.load(open(f)).items():
            o = out.setdefault(key, {"kind": r["kind"], "n": 0, "runs": 0, "per_run_worst": [], "worst": r["worst"]})
            assert o["kind"] == r["kind"], key
            o["n"] += r.get("n", 1)
            o["runs"] += 1
            o["per_run_worst"].append(r["worst"])
            o["worst"] = (min if r["kind"] == ">=" else max)(o["worst"], r["worst"])
            o["bound"] = r["bound"]
            if "shared" in r:
                o["shared"] = r["shared"]
    out["_runs"] = [os.path.basename(f) for f in files]
    return out


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    rec = merge(tag)
    path = os.path.join(ROOT, "profiles", f"{tag}_parity_observed.json")
    with open(path, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print(f"{path}: {len(rec) - 1} comparisons from {len(rec['_runs'])} run(s)")
