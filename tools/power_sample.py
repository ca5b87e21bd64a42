#!/usr/bin/env python3
"""Runs a command while a sampler (this process: it never touches HIP) reads sclk and socket power of every amdgpu card from sysfs;
the command prints `pci_bus_id <id>` (bench_micro/*.hip do) and the card with that PCI address is reported: mean power, mean
sclk over the middle of the run.   python3 tools/power_sample.py -- bench_micro/store_pattern 821 0 3"""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def rd(p):
    try:
        with open(p) as fh:
            return fh.read().strip()
    except OSError:
        return None


def main():
    cmd = sys.argv[sys.argv.index("--") + 1:]
    cards = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
        if hw:
            cards[os.path.basename(os.path.realpath(d)).lower()] = hw[0]
    samples, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            rec = {"t": time.time()}
            for pci, hw in cards.items():
                pw = rd(os.path.join(hw, "power1_average")) or rd(os.path.join(hw, "power1_input"))
                fq = rd(os.path.join(hw, "freq1_input"))
                rec[pci] = (float(pw) / 1e6 if pw else None, float(fq) / 1e6 if fq else None)
            samples.append(rec)
            time.sleep(0.05)

    th = threading.Thread(target=sample); th.start()
    t0 = time.time()
    proc = subprocess.run(cmd, stdout=subprocess.PIPE)
    t1 = time.time()
    stop.set(); th.join()
    out = proc.stdout.decode("utf-8", "replace")
    pci = None
    for line in out.splitlines():
        if line.startswith("pci_bus_id"):
            pci = line.split()[1].lower()
    mid = [s for s in samples if t0 + 0.35 * (t1 - t0) <= s["t"] <= t1 - 0.1 * (t1 - t0)]
    res = {"cmd": " ".join(cmd), "rc": proc.returncode, "seconds": t1 - t0, "pci_bus_id": pci}
    if pci in cards and mid:
        pw = [s[pci][0] for s in mid if s[pci][0] is not None]; fq = [s[pci][1] for s in mid if s[pci][1] is not None]
        res["power_W"] = sum(pw) / len(pw) if pw else None
        res["sclk_MHz"] = sum(fq) / len(fq) if fq else None
        res["samples"] = len(mid)
    res["stdout"] = [l for l in out.splitlines() if not l.startswith("pci_bus_id")][-4:]
    print(json.dumps(res)); sys.stdout.flush()
    return proc.returncode


if __name__ == "__main__":
    sys.exit(main())
