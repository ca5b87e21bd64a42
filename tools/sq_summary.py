#!/usr/bin/env python3
"""Per-kernel averages of the SQ counter passes of tools/collect_profiles.sh -> profiles/<tag>_sq_counters.json.

  python tools/sq_summary.py <tag> [--prefix prof_sq | prof_sec_sq]

Per launch, and per wave where the launch geometry gives the wave count (Grid_Size / 64 waves issued... the persistent
kernels loop, so "per wave" here is per wave LIFE, i.e. per launch / waves launched).  SQ_*_CYCLES counters tick once per
4 clocks except SQ_VALU_MFMA_BUSY_CYCLES (clocks); SQ_BUSY_CYCLES is per SE."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--prefix", default="prof_sq")
    a = ap.parse_args()
    from bench import engine_source_hash
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    grid = {}
    for sub in (a.prefix + "1", a.prefix + "2"):
        paths = glob.glob(os.path.join(ROOT, "gpurun_out", sub, "**", "*counter_collection*.csv"), recursive=True)
        for path in ([max(paths, key=os.path.getmtime)] if paths else []):      # gpurun merges: older passes stay around
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                if not k:
                    continue
                k = "%s grid=%s" % (k, r["Grid_Size"])                      # launches of different sizes are kept apart
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid[k] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]))
    out = {"tag": a.tag, "engine_source_sha256": engine_source_hash(), "unit_note": __doc__.split("\n\n")[-1], "kernels": {}}
    for k, cs in sorted(acc.items()):
        waves = grid[k][0] // 64
        per_launch = {c: sum(v) / len(v) for c, v in cs.items()}
        out["kernels"][k] = {"launches_averaged": max(len(v) for v in cs.values()), "grid_threads": grid[k][0],
                             "workgroup": grid[k][1], "waves_launched": waves,
                             "per_launch": per_launch, "per_wave_life": {c: v / waves for c, v in per_launch.items()}}
    dst = os.path.join(ROOT, "profiles", a.tag + "_sq_counters.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["kernels"].items():
        print(k, {c: round(x) for c, x in v["per_wave_life"].items()})


if __name__ == "__main__":
    main()
