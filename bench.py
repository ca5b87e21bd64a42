#!/usr/bin/env python3
"""Headline benchmark: NTRU encrypt+decrypt round trips per second at N=821, q=4096 (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --workload {roundtrip,encrypt_n701,verify_keys} ...   (BASELINE.json configs 3 [default], 4 and 5, same launcher)
  python bench.py --gpus 8 --workload verify_keys --total-batch-log2 18     (config 5 AS WORDED: 2^18 key pairs over the 8 GPUs)
  python bench.py --gpus 8 --workload encrypt_n701 --total-batch-log2 23    (config 4 as worded: 2^23 encrypts over the 8 GPUs)
  python bench.py --gpus N --steps K --warmup W          (no RANK in the environment: bench.py starts the N ranks itself, as
                                                          a CHILD `python -m torch.distributed.run`, before anything touches HIP)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Every rank checks WORLD_SIZE == --gpus and refuses to run otherwise.

One "step" = one pass of the hot path over one batch of B synthetic plaintexts that are already resident in
HBM: ntru_encrypt_batch_dev (encryptBits, index.js:87-110) followed by ntru_decrypt_batch_dev of the fresh
ciphertexts (decryptBits, index.js:111-140), both with their full witness outputs (every array the reference
returns), under one shared golden key.  Batches shard across ranks with no data-path collective (weak scaling:
B per GPU is fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PK_MAC_PEAK_T = 75.6           # measured v_pk_mad_u16 roof, profiles/archive/r01_microbench_valu_lds.txt (T MAC/s)
MFMA_I8_PEAK_T = 5000.0        # dense int8 MFMA = 2 x bf16 (~2.5 PFLOP/s), MI355X_MICROARCH.md "Matrix cores"
DOT8_PEAK_T = 302.0            # measured v_dot8_u32_u4 roof: 37.9 T lane-instr/s x 8 nibble MACs
ADD_PEAK_T = 134.0             # measured v_add_u32 roof 67 T lane-adds/s x 2 packed 16-bit coefficients per add


# BASELINE.json configs that have a multi-rank entry point here; per-GPU batch sizes (weak scaling)
WORKLOADS = {
    "roundtrip": {"profile": "n821_q4096", "batch_log2": 20},
    "encrypt_n701": {"profile": "n701_q8192", "batch_log2": 20},
    "verify_keys": {"profile": "n821_q4096", "batch_log2": 18},
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="roundtrip",
                    help="roundtrip: BASELINE.json's metric (config 3: N=821 q=4096 encryptBits + decryptBits); encrypt_n701: config 4 "
                         "(N=701 q=8192 encryptBits, 2^20 per GPU); verify_keys: config 5 (N=821 verifyKeysInputs with per-item keys, "
                         "2^18 per GPU)")
    ap.add_argument("--batch-log2", type=int, default=None, help="items per GPU per step (2^k); default: the workload's")
    ap.add_argument("--total-batch-log2", type=int, default=None,
                    help="items per step over ALL ranks (2^k, split evenly: --gpus must be a power of two); the line then says "
                         "scaling = strong.  BASELINE.json words configs 4 and 5 this way (2^23 encrypts / 2^18 keys on 8 GPUs)")
    ap.add_argument("--synthetic-keys", action="store_true",
                    help="verify_keys: uniform random operands (every flag `invalid`) instead of key pairs generated on the device")
    ap.add_argument("--profile", default=None, help="golden key / parameter set under tests/golden; default: the workload's")
    ap.add_argument("--mode", choices=["witness", "value"], default="witness",
                    help="witness: every array the reference returns; value: ciphertext / plaintext only")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-path", choices=["auto", "mac", "add", "matrix", "lockstep", "rolesplit", "chunk", "dma", "lockstep-encrypt", "rowimage"], default="auto",
                    help="kernel family: packed-u16 MAC, ternary add path, matrix cores as two workgroups per CU / with the lock-step "
                         "decrypt, or the engine's choice (same results); rolesplit, chunk, dma, lockstep-encrypt need the library "
                         "built with `make experiments` (NTRU_ENGINE_LIB)")
    ap.add_argument("--row-pitch", type=int, default=0,
                    help="row pitch of the batch arrays in elements (0 = dense rows of N elements; -1 = N rounded up to "
                         "a multiple of 64, i.e. rows on cache-line boundaries: ntru_*_batch_pitched_dev)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only for rehearsing the launch path on one GPU")
    ap.add_argument("--device", type=int, default=None, help="override the HIP device (default LOCAL_RANK)")
    ap.add_argument("--sample-r", action="store_true",
                    help="draw r with the engine's on-device sampler (generateCustomArray on a ChaCha20 stream) "
                         "instead of torch; the timed region is unchanged")
    ap.add_argument("--gather", action="store_true",
                    help="after timing, gather the workload's result rows of every rank onto rank 0 (one gather-to-root, RCCL over "
                         "xGMI; with gloo through host memory); reported under `gather`.  Always on with more than one rank")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the timing all_reduce / barrier / gather even with ONE rank "
                         "(RCCL refuses two ranks on one device, so this is how the RCCL code path is exercised on a 1-GPU box)")
    ap.add_argument("--check-rows", type=int, default=1 << 14, help="rows of the batch verified against the CPU oracle")
    ap.add_argument("--power-seconds", type=float, default=6.0,
                    help="LAST phase of the run: keep stepping for this long while socket power and shader clock are sampled "
                         "from sysfs (reported under `power`; 0 = skip).  Several seconds, so that a coarse external GPU-activity "
                         "sampler sees the GPU busy too")
    args = ap.parse_args(argv)
    w = WORKLOADS[args.workload]
    if args.total_batch_log2 is not None:
        if args.batch_log2 is not None:
            ap.error("--batch-log2 and --total-batch-log2 exclude each other")
        if args.gpus & (args.gpus - 1) or (1 << args.total_batch_log2) < args.gpus:
            ap.error("--total-batch-log2 needs a power-of-two --gpus that does not exceed the batch")
        args.batch_log2 = args.total_batch_log2 - (args.gpus.bit_length() - 1)
    if args.batch_log2 is None:
        args.batch_log2 = w["batch_log2"]
    if args.profile is None:
        args.profile = w["profile"]
    return args


def launcher_command(args, argv, environ, port=None):
    """`python bench.py --gpus N` outside a launcher (no RANK in the environment) -> the command line of the child that
    starts the N ranks, else None.  One process per GPU over torch.distributed.run, rendezvous on 127.0.0.1."""
    if args.gpus <= 1 or "RANK" in environ:
        return None
    if port is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def pick_result_line(text):
    """The one JSON line of rank 0 among whatever else the ranks printed on stdout."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if isinstance(d, dict) and "metric" in d:
                return line
    return None


def spawn_ranks(cmd):
    """Runs the launcher as a CHILD process (this process has not imported torch or touched HIP, and never replaces
    itself: an exec from a GPU-initialised process is refused on the pool), relays rank 0's JSON line, returns its rc."""
    proc = subprocess.run(cmd, stdout=subprocess.PIPE)
    line = pick_result_line(proc.stdout.decode("utf-8", "replace"))
    if line is not None:
        sys.stdout.write(line + "\n"); sys.stdout.flush()
    elif proc.returncode == 0:
        sys.stderr.write("bench: the ranks exited 0 without a result line\n")
        return 1
    return proc.returncode


def check_world(args, environ):
    """WORLD_SIZE must be what --gpus says: a mismatch means the launch is not the one the caller asked for."""
    world = int(environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d: start the ranks with `python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d` or let `python bench.py --gpus %d` do it"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    return world


def load_key(profile):
    with open(os.path.join(ROOT, "tests", "golden", "scheme_%s.json" % profile)) as fh:
        g = json.load(fh)
    o, key = g["options"], g["keys"][0]
    N = o["N"]
    pad = lambda a, dt: np.array(list(a) + [0] * (N - len(a)), dtype=dt)
    return o, pad(key["h"], np.uint16), pad(key["f"], np.int8), pad(key["fp"], np.uint8)


def make_inputs(torch, dev, B, N, d, seed, ld=None):
    """m iid uniform {0,1}; r = d ones and d twos per row, shuffled.  Rows at a pitch of ld
    elements (default N); pad elements are zero."""
    ld = ld or N
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    r = torch.zeros((B, ld), dtype=torch.uint8, device=dev)
    chunk = min(B, 1 << 16)
    for o in range(0, B, chunk):
        n = min(chunk, B - o)
        idx = torch.rand((n, N), device=dev, generator=gen).argsort(dim=1)
        r[o:o + n].scatter_(1, idx[:, :d], 1)
        r[o:o + n].scatter_(1, idx[:, d:2 * d], 2)
        del idx
    m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev, generator=gen)
    if ld != N:
        mp = torch.zeros((B, ld), dtype=torch.uint8, device=dev)
        mp[:, :N] = m
        m = mp
    return r, m


def generate_key_pairs(torch, eng, dev, o, B, first_item, max_redraws=8, first_draw_minus=None):
    """B TRUE key pairs on the device (index.js:51-79): f, g from the on-device sampler (generateCustomArray's
    shuffle on a ChaCha20 stream: df ones and df - 1 minus ones / dg and dg), fq = f^-1 mod q and fp = f^-1 mod p from
    ntru_invert_key_batch_dev, h = p fq g from ntru_public_key_batch_dev.  Rows whose f is not a unit are drawn again (the
    reference's generatePrivateKeyF retries the same way, index.js:51-66).  Returns (f, g, fq, fp, h, info).
    first_draw_minus (tests): minus ones of the FIRST draw only -- df of them make f(1) = 0, i.e. every first draw a non-unit."""
    N, q, p, df, dg = o["N"], o["q"], o["p"], o["df"], o["dg"]
    key = (np.arange(8, dtype=np.uint32) * 0x85EBCA6B + 7).astype(np.uint32)
    fs = torch.empty((B, N), dtype=torch.uint8, device=dev); gs = torch.empty((B, N), dtype=torch.uint8, device=dev)
    eng.sample_ternary_dev(N, df, df - 1 if first_draw_minus is None else first_draw_minus, 255, key, first_item, B, fs.data_ptr())   # 255 = -1 as int8
    eng.sample_ternary_dev(N, dg, dg, 255, key, (1 << 40) + first_item, B, gs.data_ptr())
    f, g = fs.view(torch.int8), gs.view(torch.int8)
    fq = torch.empty((B, N), dtype=torch.int16, device=dev); fp = torch.empty((B, N), dtype=torch.uint8, device=dev)
    fl = torch.empty(B, dtype=torch.uint8, device=dev); h = torch.empty((B, N), dtype=torch.int16, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, fq.data_ptr(), fp.data_ptr(), fl.data_ptr())
    torch.cuda.synchronize(); t_inv = time.perf_counter() - t0
    redrawn, rounds = 0, 0
    bad = torch.nonzero(fl).flatten()
    while bad.numel():
        rounds += 1
        if rounds > max_redraws:
            raise SystemExit("bench: %d private keys still not invertible after %d redraws" % (bad.numel(), max_redraws))
        n = int(bad.numel()); redrawn += n
        f2 = torch.empty((n, N), dtype=torch.uint8, device=dev)
        eng.sample_ternary_dev(N, df, df - 1, 255, key, (rounds << 44) + first_item + int(bad[0]), n, f2.data_ptr())
        fq2 = torch.empty((n, N), dtype=torch.int16, device=dev); fp2 = torch.empty((n, N), dtype=torch.uint8, device=dev)
        fl2 = torch.empty(n, dtype=torch.uint8, device=dev)
        eng.invert_key_batch_dev(N, q, p, f2.data_ptr(), n, fq2.data_ptr(), fp2.data_ptr(), fl2.data_ptr())
        torch.cuda.synchronize()
        good = fl2 == 0
        f[bad[good]] = f2.view(torch.int8)[good]; fq[bad[good]] = fq2[good]; fp[bad[good]] = fp2[good]
        bad = bad[~good]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.public_key_batch_dev(N, q, p, fq.data_ptr(), g.data_ptr(), B, h.data_ptr())
    torch.cuda.synchronize(); t_h = time.perf_counter() - t0
    return f, g, fq, fp, h, {"source": "generated on the device: ntru_sample_ternary_dev (f: %d ones, %d minus ones; g: %d / %d) -> "
                                       "ntru_invert_key_batch_dev -> ntru_public_key_batch_dev" % (df, df - 1, dg, dg),
                             "invert_ms": t_inv * 1e3, "public_key_ms": t_h * 1e3, "non_units_redrawn": redrawn, "key_pairs": B}


def check_keys_against_oracle(orc, N, q, p, host_rows):
    """A strided sample of generated keys against the CPU oracle: h = p fq g, and verifyKeysInputs raises no flag -- i.e. the oracle's
    own products f fq = 1 (mod q) and f fp = 1 (mod p) hold, and the inverses are unique."""
    f, g, fq, fp, h = host_rows
    want = orc.verify_keys_batch(N, q, p, f, g, fq, fp, h)
    return bool(np.array_equal(orc.public_key_batch(N, q, p, fq, g), h)) and not want["flags"].any(), want


def engine_source_hash():
    """sha256 over the engine's sources: ties a PMC summary to the build it was measured on (tools/pmc_summary.py stamps
    the same value; .git does not travel to the GPU box, the sources do)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ntru-circom_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def pmc_traffic(kernel, mode, batch_log2):
    """(HBM bytes per launch of `kernel`, note) from the committed rocprofv3 PMC passes (profiles/pmc_hbm_latest.json:
    2*FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950).  Counters need their own rocprofv3 passes, so
    they cannot be collected inside this run; the summary carries the hash of the sources it was measured on and is only
    served when that is the build being benchmarked -- otherwise traffic is null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")) as fh:
            d = json.load(fh)
        k = d["kernels"].get(kernel)
        if not k or d.get("mode") != mode or d.get("batch_log2") != batch_log2:
            return None, "no PMC pass on record for this kernel / mode / batch"
        if d.get("engine_source_sha256") != engine_source_hash():
            return None, "profiles/pmc_hbm_latest.json (%s) was measured on other engine sources: stale, not served" % d.get("tag")
        return k["hbm_bytes_per_launch"], "rocprofv3 PMC passes %s on these sources (git %s)" % (d.get("tag"), d.get("git_sha"))
    except (OSError, ValueError, KeyError):
        return None, "profiles/pmc_hbm_latest.json missing or unreadable"


def device_sysfs_dir(index):
    """/sys/class/drm/cardK/device of HIP device `index` (matched by PCI address), or None.  Reading it needs no HIP call."""
    import ctypes
    import glob
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) != 0:
            return None
        want = buf.value.decode().lower()
    except OSError:
        return None
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.basename(os.path.realpath(d)).lower() == want:
            return d
    return None


class PowerSampler(threading.Thread):
    """Socket power, power cap and shader clock of one device from its amdgpu hwmon files, every 20 ms (the kernels run AT the
    power cap: profiles/archive/r03_clock_power.txt -- their speed is their energy, so the line carries what the driver reports)."""

    def __init__(self, dev_dir):
        super().__init__(daemon=True)
        import glob
        hw = sorted(glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*"))) if dev_dir else []
        self.hw = hw[0] if hw else None
        self.rows, self.stop_flag = [], threading.Event()

    @staticmethod
    def _rd(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return None

    def run(self):
        while self.hw and not self.stop_flag.is_set():
            pw = self._rd(os.path.join(self.hw, "power1_average")) or self._rd(os.path.join(self.hw, "power1_input"))
            self.rows.append((time.perf_counter(), pw, self._rd(os.path.join(self.hw, "freq1_input"))))
            time.sleep(0.02)

    def summary(self, t0, t1):
        self.stop_flag.set()
        if self.is_alive():
            self.join(timeout=2)
        mid = [r for r in self.rows if t0 + 0.3 * (t1 - t0) <= r[0] <= t1]
        pw = [r[1] / 1e6 for r in mid if r[1]]
        fq = [r[2] / 1e6 for r in mid if r[2]]
        cap = self._rd(os.path.join(self.hw, "power1_cap")) if self.hw else None
        if not pw:
            return None
        return {"socket_W": sum(pw) / len(pw), "cap_W": cap / 1e6 if cap else None, "sclk_MHz": sum(fq) / len(fq) if fq else None,
                "samples": len(pw), "window_s": t1 - t0}


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


GPU_SHARE_CORES = 16           # the pool's rule of thumb for worker pools next to ONE GPU; reported beside the all-cores figure


def cpu_baseline(unit, n_rows, seconds, what):
    """Times the CPU oracle on this host, on a bounded sample of the same workload, scaled to items/s:
    (a) reference-equivalent algorithm (double FFT product + long division with per-step brute-force inverse), 1 thread;
    (b) the same on GPU_SHARE_CORES threads and on EVERY core this process may use (measured, not assumed);
    (c) exact-integer restatement on every core.  unit(rows, mode) runs the workload's CPU path on rows `rows` of the sample."""
    from oracle import ntru_oracle as orc

    unit(slice(0, 2), orc.FAITHFUL)
    t0 = time.perf_counter(); unit(slice(0, 8), orc.FAITHFUL); per = (time.perf_counter() - t0) / 8
    n = int(max(8, min(n_rows, seconds / max(per, 1e-9))))
    t0 = time.perf_counter(); unit(slice(0, n), orc.FAITHFUL); dt = time.perf_counter() - t0
    per = dt / n
    faithful = {"value": n / dt, "unit": what + "/s", "cores": 1, "kind": "port",
                "sample": "%d %s of the same workload, oracle in reference-equivalent mode (double FFT product + long division "
                          "with per-step brute-force inverse), %.1f s" % (n, what, dt)}
    avail = usable_cores()

    def threaded(threads, mode, budget_s, per_item):
        per_thread = int(max(2, min(n_rows // threads, budget_s / max(per_item, 1e-9))))
        ths = [threading.Thread(target=unit, args=(slice(i * per_thread, (i + 1) * per_thread), mode)) for i in range(threads)]
        t0 = time.perf_counter()
        [t.start() for t in ths]; [t.join() for t in ths]
        return threads * per_thread, time.perf_counter() - t0

    share = max(1, min(GPU_SHARE_CORES, avail))
    n_s, dt_s = threaded(share, orc.FAITHFUL, seconds / 4, per)
    faithful_share = {"value": n_s / dt_s, "unit": what + "/s", "cores": share, "kind": "port",
                      "sample": "%d %s, oracle in reference-equivalent mode, %d threads, %.1f s" % (n_s, what, share, dt_s)}
    n_a, dt_a = threaded(avail, orc.FAITHFUL, seconds / 4, per)
    faithful_all = {"value": n_a / dt_a, "unit": what + "/s", "cores": avail, "kind": "port",
                    "sample": "%d %s, oracle in reference-equivalent mode, one thread per usable core (%d), %.1f s"
                              % (n_a, what, avail, dt_a)}
    t0 = time.perf_counter(); unit(slice(0, 8), orc.EXACT); per_x = (time.perf_counter() - t0) / 8
    n_x, dt_x = threaded(avail, orc.EXACT, seconds / 6, per_x)
    optimized = {"value": n_x / dt_x, "unit": what + "/s", "cores": avail, "kind": "port",
                 "sample": "%d %s, oracle in exact-integer mode (schoolbook + closed-form split), %d threads, %.2f s"
                           % (n_x, what, avail, dt_x)}
    return faithful, faithful_share, faithful_all, optimized


def hbm_roofline(kernel, bytes_per_item, items, ms, traffic, traffic_note, int8_macs_per_item=None):
    """ALGORITHMIC bytes of one launch over its HIP-event time, against the 8 TB/s HBM3E roof; beside it the
    algorithmic int8 rate (2 ops per MAC of the reference's products) against the dense int8 MFMA peak."""
    s = ms * 1e-3
    gbs = bytes_per_item * items / s / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": traffic_note, "algorithmic_bytes_per_item": bytes_per_item,
         "algorithmic_bytes_per_launch": bytes_per_item * items, "ms_per_launch": ms}
    if int8_macs_per_item:
        tops = 2.0 * int8_macs_per_item * items / s / 1e12
        r["int8"] = {"achieved": tops, "peak": MFMA_I8_PEAK_T, "unit": "TOP/s (int8)", "frac": tops / MFMA_I8_PEAK_T,
                     "note": "algorithmic: 2 ops per multiply-accumulate of the reference's products (%d MACs per item), against the "
                             "dense int8 MFMA peak" % int8_macs_per_item}
    return r


def main():
    args = parse_args()
    cmd = launcher_command(args, sys.argv[1:], os.environ)
    if cmd is not None:
        raise SystemExit(spawn_ranks(cmd))
    world = check_world(args, os.environ)
    # Rank 0 prints ONE JSON line on stdout.  RCCL prints a version banner to stdout when the process group starts, so
    # file descriptor 1 is pointed at stderr for the duration of the run and the line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from oracle import ntru_oracle as orc          # the checker and the cpu_baseline leg; never inside the timed region

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    dev_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        try:
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.ones(1, device=dev)             # the first collective is where RCCL really starts: fail HERE, loudly,
                dist.all_reduce(probe); torch.cuda.synchronize()     # not in the middle of the timed region
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce over %d ranks returned %r" % (world, probe.item()))
            else:
                dist.init_process_group("gloo")
        except Exception as exc:                              # noqa: BLE001 -- no fallback to another backend, no re-exec: a clear non-zero exit
            sys.stderr.write("bench: rank %s: process group (%s, world %d) did not come up: %r\n"
                             % (os.environ.get("RANK", "0"), args.dist_backend, world, exc))
            sys.stderr.flush()
            os._exit(3)
    red_dev = dev if args.dist_backend == "nccl" else None        # where the timing all_reduce lives

    o, h_np, f_np, fp_np = load_key(args.profile)
    N, q, p, d = o["N"], o["q"], o["p"], o["dr"]
    B = 1 << args.batch_log2
    sh = pkg.sharding
    seed = sh.shard_seed(20240, rank)
    LD = N if args.row_pitch == 0 else ((N + 63) // 64 * 64 if args.row_pitch < 0 else args.row_pitch)
    pitched = LD != N
    witness = args.mode == "witness"
    wl = args.workload
    if wl != "roundtrip" and (pitched or args.sample_r or not witness):
        raise SystemExit("bench: --row-pitch / --sample-r / --mode value belong to --workload roundtrip")
    ptr = lambda t: t.data_ptr() if t is not None else None
    b16 = lambda n=B: torch.empty((n, LD), dtype=torch.int16, device=dev)    # raw uint16 patterns
    b8 = lambda n=B: torch.empty((n, LD), dtype=torch.uint8, device=dev)

    eng = pkg.Engine(dev_index)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    path_no = {"auto": 0, "mac": 1, "add": 2, "matrix": 4, "lockstep": 5, "rolesplit": 6, "chunk": 7, "dma": 8, "lockstep-encrypt": 9,
               "rowimage": 10}[args.kernel_path]
    names = {}
    to_u16 = lambda a: a.view(np.uint16) if a.dtype == np.int16 else a

    # ---- the workload: device-resident inputs and outputs, step(ev), what is gathered, what the CPU does for one item --------
    if wl in ("roundtrip", "encrypt_n701"):
        r, m = make_inputs(torch, dev, B, N, d, seed, LD)
        h = torch.from_numpy(h_np.view(np.int16)).to(dev)
        f = torch.from_numpy(f_np).to(dev)
        fp = torch.from_numpy(fp_np).to(dev)
        if args.sample_r:
            key = np.arange(8, dtype=np.uint32) * 0x9E3779B1 + 20240
            if pitched:
                raise SystemExit("bench: --sample-r writes dense rows; use it with --row-pitch 0")
            eng.sample_ternary_dev(N, d, d, p - 1, key, rank * B, B, r.data_ptr())
            torch.cuda.synchronize()
        e = b16()
        quotE = b16() if witness else None
        if wl == "roundtrip":
            value = b8()
            quot1, rem1, quot2 = (b16(), b16(), b8()) if witness else (None, None, None)
            bufs = (r, m, e, quotE, value, quot1, rem1, quot2)
        else:
            bufs = (r, m, e, quotE)
        n_ev = 3 if wl == "roundtrip" else 2

        def step(ev=None, bufs=bufs, ld=LD if pitched else None):
            if ev: ev[0].record(stream)
            eng.encrypt_batch_dev(N, q, ptr(h), ptr(bufs[0]), ptr(bufs[1]), B, ptr(bufs[2]), ptr(bufs[3]), ld=ld)
            names["encrypt"] = eng.last_kernel()
            if ev: ev[1].record(stream)
            if wl == "roundtrip":
                eng.decrypt_batch_dev(N, q, p, ptr(f), ptr(fp), ptr(bufs[2]), B, ptr(bufs[4]), ptr(bufs[5]), ptr(bufs[6]), ptr(bufs[7]), ld=ld)
                names["decrypt"] = eng.last_kernel()
                if ev: ev[2].record(stream)

        gather_src = lambda: value if wl == "roundtrip" else e.view(torch.uint8)
        unit_name = "round_trips" if wl == "roundtrip" else "encrypts"

        def cpu_unit_factory(n_cpu):
            rr, mm = r[:n_cpu, :N].contiguous().cpu().numpy(), m[:n_cpu, :N].contiguous().cpu().numpy()

            def unit(rows, mode):
                e_c, _ = orc.encrypt_batch(N, q, h_np, rr[rows], mm[rows], mode)
                if wl == "roundtrip":
                    orc.decrypt_batch(N, q, p, f_np, fp_np, e_c, mode)
            return unit
    else:       # verify_keys (config 5): per-item operands -- TRUE key pairs generated on the device, outside the timed region
        keys_info = None
        if args.synthetic_keys:                                # uniform random operands: every output is still defined, every flag says invalid
            gk = torch.Generator(device=dev); gk.manual_seed(seed + 5)
            tern = lambda: (torch.randint(0, 3, (B, N), device=dev, generator=gk) - 1).to(torch.int8)
            kf, kg = tern(), tern()
            kfq = torch.randint(0, q, (B, N), device=dev, generator=gk).to(torch.int16)
            kh = torch.randint(0, q, (B, N), device=dev, generator=gk).to(torch.int16)
            kfp = torch.randint(0, p, (B, N), device=dev, generator=gk).to(torch.uint8)
        else:
            kf, kg, kfq, kfp, kh, keys_info = generate_key_pairs(torch, eng, dev, o, B, rank * B)
        kouts = [b16(), b16(), b8(), b8(), b16(), b16()]           # quotient / remainder of fq * f, fp * f, (p fq) * g
        kflags = torch.empty(B, dtype=torch.uint8, device=dev)
        n_ev = 2

        def step(ev=None):
            if ev: ev[0].record(stream)
            eng.verify_keys_batch_dev(N, q, p, kf.data_ptr(), kg.data_ptr(), kfq.data_ptr(), kfp.data_ptr(), kh.data_ptr(), B,
                                      *[t.data_ptr() for t in kouts], kflags.data_ptr())
            names["verify_keys"] = eng.last_kernel()
            if ev: ev[1].record(stream)

        gather_src = lambda: kouts[5].view(torch.uint8)            # remainderI of the h case
        unit_name = "key_pairs"

        def cpu_unit_factory(n_cpu):
            hs = [t[:n_cpu].contiguous().cpu().numpy() for t in (kf, kg, kfq, kfp, kh)]
            hs[2], hs[4] = to_u16(hs[2]), to_u16(hs[4])

            def unit(rows, mode):
                orc.verify_keys_batch(N, q, p, hs[0][rows], hs[1][rows], hs[2][rows], hs[3][rows], hs[4][rows], mode)
            return unit
    eng.set_kernel_path(path_no)

    # ---- CPU baseline FIRST (rank 0 at N = 1): the GPU phases then run back to back until the end of the process ----------------
    out_cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_cpu = max(4096, 64 * usable_cores())
        n_cpu = min(n_cpu, B)
        faithful, faithful_share, faithful_all, optimized = cpu_baseline(cpu_unit_factory(n_cpu), n_cpu, args.cpu_seconds, unit_name)
        out_cpu = {"cpu_baseline": faithful, "cpu_baseline_gpu_share": faithful_share, "cpu_baseline_all_cores": faithful_all,
                   "cpu_baseline_optimized": optimized}

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] for _ in range(args.steps)]

    def run_steps():
        for k in range(args.steps):
            step(events[k])

    elapsed, elapsed_local = sh.timed_region(run_steps, torch.cuda.synchronize, dist, red_dev, always=args.force_dist, return_local=True)
    seg_ms = [float(np.mean([ev[i].elapsed_time(ev[i + 1]) for ev in events])) for i in range(n_ev - 1)]

    # ---- bit-exact check of a strided sample against the CPU oracle (outside the timed region): every output array ---
    n_chk = max(1, min(B, args.check_rows))
    rows = torch.tensor(sorted(set(list(range(0, B, max(1, B // n_chk))) + [B - 1])), device=dev)
    host = lambda t: to_u16(t[rows][:, :N].contiguous().cpu().numpy())
    if wl in ("roundtrip", "encrypt_n701"):
        ins = (host(r), host(m))
        got = {"e": host(e)}
        if witness:
            got["quotE"] = host(quotE)
        if wl == "roundtrip":
            got["value"] = host(value)
            if witness:
                got.update(quot1=host(quot1), rem1=host(rem1), quot2=host(quot2))

        def oracle_rows(lo, hi):
            e_o, qe_o = orc.encrypt_batch(N, q, h_np, ins[0][lo:hi], ins[1][lo:hi])
            want = {"e": e_o, "quotE": qe_o}
            if wl == "roundtrip":
                v_o, q1_o, r1_o, q2_o = orc.decrypt_batch(N, q, p, f_np, fp_np, e_o)
                want.update(value=v_o, quot1=q1_o, rem1=r1_o, quot2=q2_o)
            return want
    else:
        ins = tuple(host(t) for t in (kf, kg, kfq, kfp, kh))
        knames = ("quot_fq", "rem_fq", "quot_fp", "rem_fp", "quot_h", "rem_h")
        got = {n_: host(t) for n_, t in zip(knames, kouts)}
        got["flags"] = kflags[rows].cpu().numpy()
        if keys_info is not None:                              # the generated keys themselves: h = p fq g and no flag from the oracle's own products
            keys_ok, _ = check_keys_against_oracle(orc, N, q, p, [a[:256] for a in ins])
            keys_info["sample_equals_oracle"] = keys_ok
            keys_info["flags_valid"] = int((kflags == 0).sum())
            if not keys_ok or keys_info["flags_valid"] != B:
                raise SystemExit("bench: generated keys fail verifyKeysInputs (%d of %d valid; oracle sample %s)" % (keys_info["flags_valid"], B, keys_ok))

        def oracle_rows(lo, hi):
            return orc.verify_keys_batch(N, q, p, *[a[lo:hi] for a in ins])
    n_thr = max(1, min(GPU_SHARE_CORES, usable_cores()))
    bad, done = [], []

    def check(lo, hi):
        try:
            want = oracle_rows(lo, hi)
            for k, a in got.items():
                if not np.array_equal(a[lo:hi], want[k]):
                    bad.append(k)
            done.append(hi - lo)
        except BaseException as exc:        # a worker that dies must not read as "nothing differed"
            bad.append("check(%d, %d) raised %r" % (lo, hi, exc))

    cuts = np.linspace(0, len(ins[0]), n_thr + 1).astype(int)
    ths = [threading.Thread(target=check, args=(int(cuts[i]), int(cuts[i + 1]))) for i in range(n_thr) if cuts[i] < cuts[i + 1]]
    [t.start() for t in ths]; [t.join() for t in ths]
    if sum(done) != len(ins[0]) and not bad:
        bad.append("only %d of %d rows were compared" % (sum(done), len(ins[0])))
    ok = not bad
    if not ok:
        raise SystemExit("bench: GPU results differ from the oracle (%s) -- refusing to report a number" % sorted(set(bad)))

    # ---- multi-rank evidence: who took part, and the final gather-to-root (outside the timed region) --------------------------
    dist_info, gathered = None, None
    if dist and (args.gather or world > 1):
        # The final gather: every rank's result rows onto rank 0, once (gather-to-root; nccl = RCCL over xGMI, GPU
        # memory to GPU memory; gloo = through host memory).  Its time is reported, not hidden.
        on_dev = args.dist_backend == "nccl"
        src = gather_src()
        local = src if on_dev else src.cpu()
        sums = sh.shard_checksums(local, dist, always=args.force_dist)
        torch.cuda.synchronize(); dist.barrier(); tg = time.perf_counter()
        allv = sh.gather_rows(local, dist, always=args.force_dist, root=0)
        torch.cuda.synchronize(); dist.barrier(); tgather = time.perf_counter() - tg
        if rank == 0:
            per = [sh.shard_checksums(allv[k * B:(k + 1) * B], None)[0].item() for k in range(world)]
            gathered = {"kind": "gather_to_root", "root": 0, "consumed_on": "GPU 0 (HBM)" if on_dev else "host memory of rank 0",
                        "rows": int(allv.shape[0]), "bytes_per_rank": int(src.numel() * src.element_size()), "seconds": tgather,
                        "backend": args.dist_backend, "ranks": world,
                        "rows_equal_local_shard": bool(torch.equal(allv[:B].to(src.device), src)),
                        "every_shard_checksum_matches_its_owner": per == [int(x) for x in sums.tolist()]}
            if not (gathered["rows_equal_local_shard"] and gathered["every_shard_checksum_matches_its_owner"]):
                raise SystemExit("bench: gathered rows differ from what the ranks computed")
        del allv

    out = None
    if rank == 0:
        total = world * B * args.steps
        common = {"value": total / elapsed, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                  "scaling": "strong" if args.total_batch_log2 is not None else "weak", "vs_baseline": None,
                  "data": "synthetic", "verified_bit_exact_rows": int(rows.numel()), "verified_arrays": sorted(got.keys()),
                  "variance": "device to device 3-5 % (the headline measured 259-269 M round trips/s on the boxes of rounds 1-5), run to run "
                              "< 1 % on one device: gains below that are only claimed from same-device A/B runs (EXPERIMENTS.md)"}
        par = "batch-sharded x%d, no collective" % world
    if rank == 0 and wl == "roundtrip":
        enc_ms, dec_ms = seg_ms
        dec_bytes, enc_bytes = (8 if witness else 3) * N, (6 if witness else 4) * N
        dec_s = dec_ms * 1e-3
        dname = names.get("decrypt", "k_decrypt")
        traffic, traffic_note = pmc_traffic(dname, args.mode, args.batch_log2)
        add_path = dname.startswith(("k_decrypt_s", "k_decrypt_t"))
        mfma = None
        if dname.startswith("k_decrypt_m"):
            # int8 matrix path: decrypt = 3 plane-products (e_lo, e_hi against f; lifted against fp), each 27 tile
            # products per 32x32 output tile on a 32-padded grid (26 x 26 tiles at N = 821)
            NT = (N + 31) // 32
            n_mfma = 3.0 * NT * (NT + 1) * ((B + 31) // 32)
            ops = 2.0 * 32768 * n_mfma
            mfma = {"kernel": dname, "executed": ops / dec_s / 1e12, "peak": MFMA_I8_PEAK_T, "unit": "TOP/s (int8, executed)",
                    "executed_frac": ops / dec_s / 1e12 / MFMA_I8_PEAK_T, "instructions_per_launch": n_mfma,
                    "note": "v_mfma_i32_32x32x32_i8 instructions actually issued (three digit-plane products on a 32-padded tile grid, "
                            "1.6x the algorithmic MACs): a utilisation figure, not a roofline fraction; = SQ_INSTS_MFMA of the PMC "
                            "pass; the instruction's measured issue roof is 75.4 G/s = 4.94 POP/s (profiles/archive/r03_clock_power.txt)"}
            valu = {"kernel": dname, "note": "matrix-core path: see `mfma`"}
        elif "+dot8" in dname:
            # product 1 steps over f (adds), product 2 is all N^2 nibble MACs on v_dot8_u32_u4
            w1, w2 = float(np.count_nonzero(f_np)) * N * B, float(N) * N * B
            ideal_s = w1 / (ADD_PEAK_T * 1e12) + w2 / (DOT8_PEAK_T * 1e12)
            valu = {"kernel": dname, "achieved": (w1 + w2) / dec_s / 1e12, "peak": (w1 + w2) / ideal_s / 1e12,
                    "unit": "T ops/s (coefficient-adds of product 1 + nibble-MACs of product 2)",
                    "note": "peak = this mix at the measured issue roofs: v_add_u32 67 T/s x 2 coefficients for the "
                            "stepping product, v_dot8_u32_u4 302 T nibble-MAC/s for the dot8 product "
                            "(profiles/archive/r01_microbench_valu_lds.txt); frac = ideal issue time / measured time"}
        elif add_path:
            nz_f = float(np.count_nonzero(f_np)) / N
            nz_b = 2.0 / 3.0                                  # the lifted message is ~uniform over {0,1,2}
            valu = {"kernel": dname, "achieved": (nz_f + nz_b) * N * N * B / dec_s / 1e12, "peak": ADD_PEAK_T,
                    "unit": "T coefficient-adds/s",
                    "note": "ternary add path: (non-zero steps) x N packed 16-bit adds per product; peak = measured "
                            "v_add_u32 issue roof x 2 coefficients; the scalar-issued step control is what keeps it "
                            "below that roof (DESIGN.md section 4.3)"}
        else:
            valu = {"kernel": dname, "achieved": 2.0 * N * N * B / dec_s / 1e12, "peak": PK_MAC_PEAK_T,
                    "unit": "T MAC/s", "note": "2*N^2 MACs per decrypt; peak = measured v_pk_mad_u16 issue roof"}
        if "achieved" in valu:
            valu["frac"] = valu["achieved"] / valu["peak"]
        # `roofline`: the dominant kernel (decrypt), ALGORITHMIC: 8N (3N value-only) bytes per item over its HIP-event time
        # against 8 TB/s, with the algorithmic int8 rate (2 N^2 MACs per decrypt) beside it
        roofline = hbm_roofline(dname, dec_bytes, B, dec_ms, traffic, traffic_note, 2 * N * N)
        roofline["note"] = ("decrypt is the longer of the step's two kernels; what binds both is the 1400 W socket power cap (see `power` and "
                            "profiles/r04_rowimage_encrypt.txt): their time is their energy over the cap")
        out = dict(common)
        out.update({
            "metric": "NTRU encrypt+decrypt round trips per second at N=%d, q=%d" % (N, q), "unit": "round_trips/s",
            "dtype": "i8" if mfma else "u16",
            "config": {"workload": "N=%d q=%d p=%d d=%d, batch=2^%d round trips per GPU per step, shared golden key, "
                                   "%s outputs" % (N, q, p, d, args.batch_log2, "full-witness" if witness else "value-only"),
                       "mode": args.mode, "kernel_path": args.kernel_path, "row_pitch_elements": LD,
                       "arithmetic": "int8 digit planes on the matrix cores, int32 accumulation (exact)" if mfma else "exact u16 vector ALU",
                       "seed": 20240, "r_source": "device sampler" if args.sample_r else "torch", "parallelism": par},
            "kernels_ms": {names.get("encrypt", "k_encrypt"): enc_ms, dname: dec_ms},
            "roofline": roofline,
            "roofline_encrypt": hbm_roofline(names.get("encrypt", "k_encrypt"), enc_bytes, B, enc_ms, None,
                                             "no PMC pass on record for this kernel in this line", N * N),
            "valu": valu, "mfma": mfma,
            "hbm_gbs_round_trip": (dec_bytes + enc_bytes) * B / ((dec_ms + enc_ms) * 1e-3) / 1e9,
        })
        if world == 1 and args.kernel_path == "auto" and dname.startswith("k_decrypt_m"):
            # The wavefront-per-ciphertext VALU families (BASELINE north_star's design), same buffers, outside the timed
            # region: a few steps, and their outputs must equal the ones just verified.
            cmp_names = ("e", "quotientE", "value", "quotient1", "remainder1", "quotient2") if witness else ("e", "value")
            cmp_idx = (2, 3, 4, 5, 6, 7) if witness else (2, 4)
            ref_out = [bufs[i][:, :N].clone() for i in cmp_idx]
            alt = bufs
            if pitched:      # those families read and write dense rows: give them dense copies of the same inputs
                dense = lambda t: None if t is None else torch.empty((B, N), dtype=t.dtype, device=dev)
                alt = (r[:, :N].contiguous(), m[:, :N].contiguous()) + tuple(dense(t) for t in bufs[2:])
            eng.set_kernel_path(2)
            step(None, alt, None); torch.cuda.synchronize()
            differ = [n_ for n_, a, i in zip(cmp_names, ref_out, cmp_idx) if not bool(torch.equal(a, alt[i][:, :N]))]
            alt_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(3)]
            for ev in alt_ev:
                step(ev, alt, None)
            torch.cuda.synchronize()
            a_enc = float(np.mean([ev[0].elapsed_time(ev[1]) for ev in alt_ev]))
            a_dec = float(np.mean([ev[1].elapsed_time(ev[2]) for ev in alt_ev]))
            out["valu_families"] = {"value": B / ((a_enc + a_dec) * 1e-3), "unit": "round_trips/s",
                                    "kernels_ms": {names["encrypt"]: a_enc, names["decrypt"]: a_dec},
                                    "outputs_equal_matrix_path": not differ, "arrays_compared_on_whole_batch": list(cmp_names),
                                    "rows_compared": B,
                                    "note": "ntru_engine_set_kernel_path(2): one ciphertext per wavefront, no MFMA; kernel "
                                            "times only, 3 steps"}
            names["encrypt"], names["decrypt"] = list(out["kernels_ms"])
            eng.set_kernel_path(0)
            del ref_out
        if world == 1 and args.kernel_path == "auto":
            # The other function of the path with per-item operands, verifyKeysInputs (index.js:141-197, BASELINE config 5),
            # outside the timed region: 2^18 TRUE key pairs generated on the device (config 5's size), kernel time of 10 launches,
            # a sample of the keys and of every witness array against the oracle (`--workload verify_keys` is the same thing as a
            # bench line of its own).
            Bk = 1 << 18
            kf, kg, kfq, kfp, kh, kinfo = generate_key_pairs(torch, eng, dev, o, Bk, 0)
            k16 = lambda: torch.empty((Bk, N), dtype=torch.int16, device=dev)
            k8 = lambda: torch.empty((Bk, N), dtype=torch.uint8, device=dev)
            kouts = [k16(), k16(), k8(), k8(), k16(), k16()]
            kflags = torch.empty(Bk, dtype=torch.uint8, device=dev)
            vk = lambda: eng.verify_keys_batch_dev(N, q, p, kf.data_ptr(), kg.data_ptr(), kfq.data_ptr(), kfp.data_ptr(),
                                                   kh.data_ptr(), Bk, *[t.data_ptr() for t in kouts], kflags.data_ptr())
            for _ in range(5):                                  # the first launches touch the output pages and meet the clocks the
                vk()                                            # previous (vector-ALU) leg left behind: not what is being measured
            torch.cuda.synchronize()
            kev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            kev[0].record(stream)
            for i in range(10):
                vk(); kev[i + 1].record(stream)
            torch.cuda.synchronize()
            vms = float(np.mean([kev[i].elapsed_time(kev[i + 1]) for i in range(10)]))
            krows = torch.arange(0, Bk, Bk // 256, device=dev)
            khost = lambda t: to_u16(t[krows].contiguous().cpu().numpy())
            keys_ok, kwant = check_keys_against_oracle(orc, N, q, p, [khost(t) for t in (kf, kg, kfq, kfp, kh)])
            kgot = [khost(t) for t in kouts] + [kflags[krows].cpu().numpy()]
            witness_ok = all(np.array_equal(a, b) for a, b in zip(kgot, kwant.values()))
            kinfo.update(sample_equals_oracle=keys_ok, flags_valid=int((kflags == 0).sum()))
            if not (keys_ok and witness_ok and kinfo["flags_valid"] == Bk):
                raise SystemExit("bench: verify_keys leg differs from the oracle (keys %s, witness %s, %d of %d valid)"
                                 % (keys_ok, witness_ok, kinfo["flags_valid"], Bk))
            out["verify_keys"] = {"value": Bk / (vms * 1e-3), "unit": "key_pairs/s", "kernel": eng.last_kernel(), "ms": vms,
                                  "batch": Bk, "hbm_gbs": 17.0 * N * Bk / (vms * 1e-3) / 1e9, "keys": kinfo,
                                  "flags_valid": kinfo["flags_valid"], "witness_rows_equal_oracle": int(krows.numel()),
                                  "roofline": hbm_roofline(eng.last_kernel(), 17 * N, Bk, vms, None,
                                                           "PMC passes: profiles/*_secondary_pmc_hbm.json", 3 * N * N),
                                  "note": "verifyKeysInputs on 2^18 true key pairs generated on the device (every flag valid), kernel time only"}
            del kf, kg, kfq, kh, kfp, kouts
    elif rank == 0 and wl == "encrypt_n701":
        enc_ms = seg_ms[0]
        ename = names.get("encrypt", "k_encrypt")
        out = dict(common)
        out.update({
            "metric": "NTRU encryptBits operations per second at N=%d, q=%d (BASELINE.json config 4)" % (N, q), "unit": "encrypts/s",
            "dtype": "i8" if ename.startswith("k_encrypt_m") or ename == "k_encrypt_w" else "u16",
            "config": {"workload": "N=%d q=%d d=%d, batch=2^%d encryptBits per GPU per step (config 4: 2^23 over 8 GPUs), shared golden "
                                   "key, full-witness outputs" % (N, q, d, args.batch_log2),
                       "kernel_path": args.kernel_path, "seed": 20240, "parallelism": par},
            "kernels_ms": {ename: enc_ms},
            "roofline": hbm_roofline(ename, 6 * N, B, enc_ms, None, "PMC passes: profiles/*_secondary_pmc_hbm.json", N * N),
        })
    elif rank == 0:
        vms = seg_ms[0]
        vname = names.get("verify_keys", "k_verify_keys")
        out = dict(common)
        out.update({
            "metric": "NTRU verifyKeysInputs key pairs per second at N=%d, q=%d (BASELINE.json config 5)" % (N, q), "unit": "key_pairs/s",
            "dtype": "i8" if vname.startswith("k_verify_keys_m") else "u16",
            "config": {"workload": "N=%d q=%d p=%d, batch=2^%d key pairs per GPU per step (%d over %d GPU%s; config 5: 2^18 keys on 8 GPUs) "
                                   "with PER-ITEM operands, every witness array; %s"
                                   % (N, q, p, args.batch_log2, world * B, world, "" if world == 1 else "s",
                                      "synthetic operands (f, g ternary, fq, h uniform mod q, fp uniform mod p)" if keys_info is None else
                                      "TRUE key pairs generated on the device"),
                       "kernel_path": args.kernel_path, "seed": 20240, "parallelism": par},
            "keys": keys_info,
            "kernels_ms": {vname: vms},
            "roofline": hbm_roofline(vname, 17 * N, B, vms, None, "PMC passes: profiles/*_secondary_pmc_hbm.json", 3 * N * N),
        })

    # ---- LAST phase: what limits the kernels -- socket power against its cap and the shader clock over a sustained run of the
    # same steps (outside the timed region; also long enough for a coarse external GPU-activity sampler to see the GPU busy)
    power = None
    if args.power_seconds > 0:
        sampler = PowerSampler(device_sysfs_dir(dev_index))      # every rank samples ITS device (reported per rank under `dist`)
        sampler.start()
        tp0 = time.perf_counter(); n_sus = 0
        while time.perf_counter() - tp0 < args.power_seconds:
            for _ in range(20):
                step(); n_sus += 1
            torch.cuda.synchronize()
        tp1 = time.perf_counter()
        if sampler:
            power = sampler.summary(tp0, tp1)
        if power:
            power["sustained_ms_per_step"] = (tp1 - tp0) / n_sus * 1e3
            if wl == "roundtrip" and out and out.get("mfma"):
                # Energy view of the step (what the power cap turns into time): measured dynamic energy against the floor that the
                # matrix instructions and the HBM bytes alone would cost, priced with this round's measurements
                # (profiles/r04_rowimage_encrypt.txt: 17.6 nJ per v_mfma_i32_32x32x32_i8 on live operands, 113 pJ per byte stored,
                # ~60 pJ per byte loaded; idle socket 300 W).
                NTt = (N + 31) // 32
                n_mfma_step = 5.0 * NTt * (NTt + 1) * ((B + 31) // 32)          # 2 planes encrypt + 3 planes decrypt
                stored = ((4 if witness else 2) + (6 if witness else 1)) * N * B
                loaded = (2 + 2) * N * B
                floor_j = n_mfma_step * 17.6e-9 + stored * 113e-12 + loaded * 60e-12
                dyn_j = (power["socket_W"] - 300.0) * power["sustained_ms_per_step"] * 1e-3
                power["energy"] = {"dynamic_J_per_step": dyn_j, "floor_J_per_step": floor_j, "frac": floor_j / dyn_j,
                                   "floor_ms_per_step_at_the_cap": floor_j / ((power["cap_W"] or 1400.0) - 300.0) * 1e3,
                                   "note": "floor = matrix instructions on live operands + HBM bytes only; frac = floor / measured: how "
                                           "close the step is to what the socket power cap allows for this instruction mix"}
            power["note"] = ("the workload's steps back to back; the matrix-core kernels run against the socket power cap with the shader "
                             "clock pulled below its 2.4 GHz top (profiles/archive/r03_clock_power.txt, r04_rowimage_encrypt.txt)")
    if dist:
        # who took part, and how each rank fared: its own seconds for the K steps (up to its own synchronize, before the closing
        # barrier), its kernels' HIP-event times, socket power and shader clock of ITS device over the sustained phase -- the
        # metric is a maximum over ranks, so a straggler must be visible by itself
        me = {"rank": rank, "local_rank": local_rank, "device": dev_index, "pid": os.getpid(),
              "pci_bus_id": os.path.basename(os.path.realpath(device_sysfs_dir(dev_index) or "?")),
              "name": torch.cuda.get_device_name(dev_index), "ms_per_step": elapsed_local / args.steps * 1e3,
              "kernels_ms": dict(zip([names.get(k, k) for k in (("encrypt", "decrypt") if wl == "roundtrip" else
                                                                 ("encrypt",) if wl == "encrypt_n701" else ("verify_keys",))], seg_ms)),
              "items_per_step": B,
              "power": {k: power[k] for k in ("socket_W", "cap_W", "sclk_MHz", "sustained_ms_per_step")} if power else None}
        everyone, straggler = sh.rank_reports(me, dist, always=args.force_dist)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:                                   # noqa: BLE001 -- reporting only
            rccl = None
        dist_info = {"backend": args.dist_backend, "world": world, "rccl_version": rccl if args.dist_backend == "nccl" else None,
                     "ranks": everyone, "straggler": straggler,
                     "timing": "barrier + synchronize on both sides of the K steps, all_reduce(MAX) of the ranks' seconds; "
                               "ranks[].ms_per_step is each rank's own time up to its synchronize"}
    if rank == 0:
        out["power"] = power
        if dist_info:
            out["dist"] = dist_info
        if gathered:
            out["gather"] = gathered
        out.update(out_cpu)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
