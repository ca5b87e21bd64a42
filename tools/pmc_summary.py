#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the summaries kept under profiles/.

  python tools/pmc_summary.py <tag> [--mode witness|value] [--N 821] [--logB 20]

(rocprofv3 writes CSV with --output-format csv; its default rocpd database is converted first with
 `rocpd2csv -i <results.db> -d gpurun_out/prof_<x>/csv`.)
Reads  gpurun_out/prof_stats/**/**_kernel_stats.csv            (rocprofv3 --kernel-trace --stats)
       gpurun_out/prof_fetch/**/**_counter_collection.csv      (rocprofv3 --pmc FETCH_SIZE --kernel-trace)
       gpurun_out/prof_write/**/**_counter_collection.csv      (rocprofv3 --pmc WRITE_SIZE --kernel-trace)
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_hbm.json and profiles/pmc_hbm_latest.json.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE counts 64 B per 128-B request on gfx950
(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact."""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """rocprof's demangled kernel name -> the name ntru_engine_last_kernel / bench.py report."""
    m = re.match(r"(?:void )?(k_[a-z_0-9]+)(?:<([0-9a-z, ]+)>)?\(", name)
    if not m:
        return None
    fam = m.group(1)
    if m.group(2) is None:
        return fam
    args = [x.strip() for x in m.group(2).split(",")]
    if args[-1] in ("true", "false"):                      # k_decrypt_s<K, ME, D8>
        fam += "+dot8" if args.pop() == "true" else ""
    return "%s<%s>" % (fam, ",".join(args))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--mode", default="witness")
    ap.add_argument("--N", type=int, default=821)
    ap.add_argument("--logB", type=int, default=20)
    ap.add_argument("--items", type=int, default=0, help="items per launch when it is not 2^logB (per-item kernels)")
    ap.add_argument("--prefix", default="prof", help="gpurun_out/<prefix>_{stats,fetch,write}: prof (bench.py) or prof_sec (bench_configs.py)")
    ap.add_argument("--largest", action="store_true", help="average only the launches of the largest batch of each kernel")
    ap.add_argument("--no-latest", action="store_true", help="do not overwrite profiles/pmc_hbm_latest.json (secondary kernels)")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    sdir = os.path.join(ROOT, "gpurun_out", a.prefix + "_stats")
    stats = glob.glob(os.path.join(sdir, "**", "*_kernel_stats.csv"), recursive=True)
    trace = glob.glob(os.path.join(sdir, "**", "*kernel_trace.csv"), recursive=True)
    if stats:
        rows = list(csv.reader(open(max(stats, key=os.path.getmtime))))
    elif trace:      # rocpd output converted with rocpd2csv: rebuild the --stats table from the kernel trace
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(max(trace, key=os.path.getmtime))):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        total = float(sum(sum(v) for v in dur.values()))
        rows = [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]]
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            mean = sum(v) / len(v)
            sd = (sum((x - mean) ** 2 for x in v) / (len(v) - 1)) ** 0.5 if len(v) > 1 else 0.0
            rows.append([k, len(v), sum(v), "%.6f" % mean, "%.2f" % (100 * sum(v) / total), min(v), max(v), "%.6f" % sd])
    else:
        rows = None
    if rows:
        with open(os.path.join(out, a.tag + "_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            for r in rows:
                w.writerow([r[0][:110]] + r[1:])
    kern = {}
    for cname, sub in (("FETCH_SIZE", a.prefix + "_fetch"), ("WRITE_SIZE", a.prefix + "_write")):
        files = glob.glob(os.path.join(ROOT, "gpurun_out", sub, "**", "*counter_collection*.csv"), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):     # gpurun merges into gpurun_out/: older passes stay
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == cname:
                acc[k].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            # --largest: the persistent kernels launch the same grid for every batch size, so launches of different batches
            # (tools/bench_configs.py) can only be told apart by their traffic: keep the largest batch's launches
            if a.largest:
                v = [x for x in v if x >= 0.9 * max(v)]
            kern.setdefault(k, {})[cname + "_KiB_per_launch"] = sum(v) / len(v)
            kern[k][cname + "_launches_averaged"] = len(v)
    B = 1 << a.logB
    per_item = {"k_encrypt": (6, 4), "k_decrypt": (8, 3), "k_verify_keys": (17, 17), "k_polymul": (8, 8), "k_public_key": (5, 5)}
    if a.items:
        B = a.items
    for k, v in kern.items():
        if "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
            v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KiB_per_launch"] + v["WRITE_SIZE_KiB_per_launch"]) * 1024
            for fam, (wit, val) in per_item.items():
                if k.startswith(fam):
                    v["algorithmic_bytes_per_launch"] = (wit if a.mode == "witness" else val) * a.N * B
                    v["ratio_to_algorithmic"] = v["hbm_bytes_per_launch"] / v["algorithmic_bytes_per_launch"]
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    from bench import engine_source_hash
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "ntru-circom_amd/csrc"], capture_output=True,
                                    text=True).stdout.strip())
    except OSError:
        sha, dirty = None, None
    doc = {"tag": a.tag, "mode": a.mode, "N": a.N, "batch_log2": a.logB,
           "git_sha": (sha + ("+uncommitted csrc changes" if dirty else "")) if sha else None,
           "engine_source_sha256": engine_source_hash(),
           "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py",
           "kernels": kern}
    for name in (a.tag + "_pmc_hbm.json",) + (() if a.no_latest else ("pmc_hbm_latest.json",)):
        with open(os.path.join(out, name), "w") as fh:
            json.dump(doc, fh, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
