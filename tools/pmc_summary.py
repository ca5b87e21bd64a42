#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the summaries kept under profiles/.

  python tools/pmc_summary.py <tag> [--prefix prof|prof_sec] [--mode witness|value] [--logB 20] [--no-latest]

Reads  gpurun_out/<prefix>_stats/**/*kernel_trace.csv          (rocprofv3 --kernel-trace --stats)
       gpurun_out/<prefix>_fetch/**/*counter_collection.csv    (rocprofv3 --pmc FETCH_SIZE --kernel-trace)
       gpurun_out/<prefix>_write/**/*counter_collection.csv    (rocprofv3 --pmc WRITE_SIZE --kernel-trace)
       gpurun_out/<prefix>_{stats,fetch,write}_launches.jsonl  (NTRU_LAUNCH_LOG of the same runs: kernel, N, items, bytes per item
                                                                of every launch, in launch order -- ntru-circom_amd/engine.py)
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_hbm.json and (unless --no-latest) profiles/pmc_hbm_latest.json.

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE counts 64 B per 128-B request on gfx950
(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.

Scaling is PER LAUNCH: the k-th dispatch of a kernel in the counter file is matched with the k-th record of that kernel in the
launch log, so a kernel that one run launches at several sizes (bench.py: verify_keys at 2^18 beside the 2^20 headline;
bench_configs.py: k_encrypt_md at three N) gets each launch divided by ITS algorithmic bytes.  A kernel whose dispatch count
differs from its log (launched by a multi-kernel call that is not logged) gets no ratio instead of a wrong one.
Durations: the first two launches of every kernel are dropped (cold instruction cache, first-touch page faults of the output
arrays); mean, median, min and max of the rest are reported beside the all-launch mean rocprofv3 --stats prints."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """rocprof's demangled kernel name -> the name ntru_engine_last_kernel / bench.py report."""
    m = re.match(r"(?:void )?(k_[a-z_0-9]+)(?:<([0-9a-z, ]+)>)?\(", name)
    if not m:
        return None
    fam = m.group(1)
    if m.group(2) is None:
        return fam
    if fam == "k_encrypt_wp":                                      # k_encrypt_wp<BITS>: the engine reports the family name alone
        return fam
    args = [x.strip() for x in m.group(2).split(",")]
    if fam == "k_decrypt_s" and args[-1] in ("true", "false"):     # k_decrypt_s<K, ME, D8>: the engine reports "k_decrypt_s+dot8<K,ME>"
        fam += "+dot8" if args.pop() == "true" else ""
    return "%s<%s>" % (fam, ",".join(args))


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def load_log(path):
    """kernel -> [records in launch order].  k_public_key_m is reported by the engine under that name but dispatched as
    k_product_tern_m: fold the alias."""
    by = collections.defaultdict(list)
    if not path or not os.path.exists(path):
        return by
    for line in open(path):
        line = line.strip()
        if line:
            r = json.loads(line)
            by[{"k_public_key_m": "k_product_tern_m"}.get(r["kernel"], r["kernel"])].append(r)
    return by


def dispatches(csv_path, counter=None):
    """kernel -> [(dispatch id, value)] in dispatch order; value = counter value (summed over the rows of one dispatch) or
    duration in ns when counter is None."""
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(csv_path)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        did = int(r.get("Dispatch_Id") or r.get("Correlation_Id") or 0)
        if counter is None:
            per[k][did] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        elif r["Counter_Name"] == counter:
            per[k][did] = per[k].get(did, 0.0) + float(r["Counter_Value"])
    return {k: [v[d] for d in sorted(v)] for k, v in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--mode", default="witness")
    ap.add_argument("--logB", type=int, default=20, help="batch of the headline launches (what bench.py's pmc_traffic() keys on)")
    ap.add_argument("--prefix", default="prof", help="gpurun_out/<prefix>_{stats,fetch,write}: prof (bench.py) or prof_sec (bench_configs.py)")
    ap.add_argument("--no-latest", action="store_true", help="do not overwrite profiles/pmc_hbm_latest.json (secondary kernels)")
    ap.add_argument("--drop", type=int, default=2, help="launches of every kernel dropped from the duration statistics")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    go = os.path.join(ROOT, "gpurun_out")

    # ---- durations ------------------------------------------------------------------------------------------------------
    trace = newest(os.path.join(go, a.prefix + "_stats", "**", "*kernel_trace.csv"))
    stats_rows = []
    if trace:
        dur = dispatches(trace)
        total = float(sum(sum(v) for v in dur.values())) or 1.0
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            warm = v[a.drop:] if len(v) > a.drop else v
            stats_rows.append({"kernel": k, "calls": len(v), "total_ns": sum(v), "pct": 100.0 * sum(v) / total,
                               "mean_all_ns": sum(v) / len(v), "warm_calls": len(warm), "warm_mean_ns": sum(warm) / len(warm),
                               "warm_median_ns": statistics.median(warm), "warm_min_ns": min(warm), "warm_max_ns": max(warm)})
        slog = load_log(os.path.join(go, a.prefix + "_stats_launches.jsonl"))
        for r in stats_rows:                                # durations per launch size, where the log covers every dispatch
            v, lg = dur[r["kernel"]], slog.get(r["kernel"], [])
            if len(lg) == len(v) and len({(x["N"], x["items"]) for x in lg}) > 1:
                groups = collections.OrderedDict()
                for d_ns, x in zip(v, lg):
                    groups.setdefault((x["N"], x["items"]), []).append(d_ns)
                r["by_launch_size"] = [{"N": N, "items": items, "calls": len(ds), "warm_median_ms": statistics.median(ds[a.drop:] or ds) / 1e6,
                                        "warm_mean_ms": (sum(ds[a.drop:] or ds) / len(ds[a.drop:] or ds)) / 1e6} for (N, items), ds in groups.items()]
        with open(os.path.join(out, a.tag + "_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "Percentage", "AverageNs(all launches)",
                        "WarmCalls(first %d dropped)" % a.drop, "WarmAverageNs", "WarmMedianNs", "WarmMinNs", "WarmMaxNs"])
            for r in stats_rows:
                w.writerow([r["kernel"][:110], r["calls"], r["total_ns"], "%.2f" % r["pct"], "%.1f" % r["mean_all_ns"], r["warm_calls"],
                            "%.1f" % r["warm_mean_ns"], "%.1f" % r["warm_median_ns"], r["warm_min_ns"], r["warm_max_ns"]])

    # ---- counters, per launch -------------------------------------------------------------------------------------------
    kern = {}
    vals = {}
    for cname, sub in (("FETCH_SIZE", "_fetch"), ("WRITE_SIZE", "_write")):
        f = newest(os.path.join(go, a.prefix + sub, "**", "*counter_collection*.csv"))
        if f:
            vals[cname] = (dispatches(f, cname), load_log(os.path.join(go, a.prefix + sub + "_launches.jsonl")))
    names = set()
    for cname in vals:
        names |= set(vals[cname][0])
    for k in sorted(names):
        if not all(k in vals[c][0] for c in ("FETCH_SIZE", "WRITE_SIZE") if c in vals) or len(vals) < 2:
            continue
        fe, fe_log = vals["FETCH_SIZE"][0][k], vals["FETCH_SIZE"][1].get(k, [])
        wr, wr_log = vals["WRITE_SIZE"][0][k], vals["WRITE_SIZE"][1].get(k, [])
        if len(fe) != len(wr):
            continue
        hbm = [(2 * x + y) * 1024.0 for x, y in zip(fe, wr)]
        entry = {"launches": len(hbm), "FETCH_SIZE_KiB_per_launch": sum(fe) / len(fe), "WRITE_SIZE_KiB_per_launch": sum(wr) / len(wr)}
        matched = len(fe_log) == len(fe) and len(wr_log) == len(wr) and \
            [(r["N"], r["items"], r["bytes_per_item"]) for r in fe_log] == [(r["N"], r["items"], r["bytes_per_item"]) for r in wr_log]
        if matched:
            groups = collections.OrderedDict()
            for h, r in zip(hbm, fe_log):
                groups.setdefault((r["N"], r["items"], r["bytes_per_item"]), []).append(h)
            entry["by_launch_size"] = []
            for (N, items, bpi), hs in groups.items():
                alg = float(bpi) * items
                entry["by_launch_size"].append({"N": N, "items": items, "algorithmic_bytes_per_item": bpi, "launches": len(hs),
                                                "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_launch": sum(hs) / len(hs),
                                                "ratio_to_algorithmic": sum(hs) / len(hs) / alg})
            # the headline entry bench.py serves as roofline.traffic: the launches at the headline batch
            head = [g for g in entry["by_launch_size"] if g["items"] == 1 << a.logB] or entry["by_launch_size"][-1:]
            entry["hbm_bytes_per_launch"] = head[0]["hbm_bytes_per_launch"]
            entry["algorithmic_bytes_per_launch"] = head[0]["algorithmic_bytes_per_launch"]
            entry["ratio_to_algorithmic"] = head[0]["ratio_to_algorithmic"]
            entry["items_per_launch"] = head[0]["items"]
        else:
            entry["hbm_bytes_per_launch"] = sum(hbm) / len(hbm)
            entry["ratio_to_algorithmic"] = None
            entry["note"] = "dispatch count %d / %d does not match the launch log %d / %d: no per-launch scaling" % (
                len(fe), len(wr), len(fe_log), len(wr_log))
        kern[k] = entry

    sys.path.insert(0, ROOT)
    from bench import engine_source_hash
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "ntru-circom_amd/csrc"], capture_output=True,
                                    text=True).stdout.strip())
    except OSError:
        sha, dirty = None, None
    doc = {"tag": a.tag, "mode": a.mode, "batch_log2": a.logB,
           "git_sha": (sha + ("+uncommitted csrc changes" if dirty else "")) if sha else None,
           "engine_source_sha256": engine_source_hash(),
           "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 "
                      + ("bench.py" if a.prefix == "prof" else "tools/bench_configs.py"),
           "durations": {r["kernel"]: {"calls": r["calls"], "mean_all_ms": r["mean_all_ns"] / 1e6, "warm_mean_ms": r["warm_mean_ns"] / 1e6,
                                       "warm_median_ms": r["warm_median_ns"] / 1e6, "warm_min_ms": r["warm_min_ns"] / 1e6,
                                       "warm_max_ms": r["warm_max_ns"] / 1e6, "by_launch_size": r.get("by_launch_size")}
                         for r in stats_rows if r["kernel"].startswith("k_")},
           "kernels": kern}
    for name in (a.tag + "_pmc_hbm.json",) + (() if a.no_latest else ("pmc_hbm_latest.json",)):
        with open(os.path.join(out, name), "w") as fh:
            json.dump(doc, fh, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
