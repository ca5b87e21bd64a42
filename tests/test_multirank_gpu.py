"""More than one rank on the HIP kernels.  The GPU box has ONE device and RCCL refuses two ranks on one device
("Duplicate GPU detected", profiles/archive/r02_rccl_2ranks_1gpu_refused.txt), so the two-rank runs use gloo with both ranks on device
0: the sharding, the launch path of bench.py, the barrier / max-over-ranks timing and the gather-to-root all execute with
world_size 2 around the real kernels.  RCCL itself runs as a one-rank process group (--force-dist)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, q_out):
    """Config 4's shard of one rank (N=701, q=8192, encryptBits index.js:87-110) through ntru_encrypt_batch_dev."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import bench
    pkg = ge.load_package()
    sh = pkg.sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, h_np, _, _ = bench.load_key("n701_q8192")
        N, q, d = o["N"], o["q"], o["dr"]
        B = 1 << 18                                         # per rank (weak scaling); BASELINE config 4 is 2^20 per GPU
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        r, m = bench.make_inputs(torch, dev, B, N, d, sh.shard_seed(20240, rank))
        h = torch.from_numpy(h_np.view(np.int16)).to(dev)
        e = torch.empty((B, N), dtype=torch.int16, device=dev)
        quot = torch.empty((B, N), dtype=torch.int16, device=dev)
        eng = pkg.Engine(0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)

        def run():
            eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), quot.data_ptr())

        elapsed = sh.timed_region(run, torch.cuda.synchronize, dist, None)
        kernel = eng.last_kernel()
        # every rank hands rank 0 the INPUT rows of a strided sample so that rank 0 can replay them on the oracle
        rows = torch.arange(0, B, 1021, device=dev)
        sample = torch.cat([r[rows], m[rows]], dim=1).cpu()
        e_host = e.cpu().view(torch.uint8)                   # gloo moves bytes (it has no 16-bit integer type); RCCL takes any dtype
        sums = sh.shard_checksums(e_host, dist)
        all_e = sh.gather_rows(e_host, dist, root=0)
        all_in = sh.gather_rows(sample, dist, root=0)
        if rank == 0:
            all_e = all_e.view(torch.int16)
            from oracle import ntru_oracle as orc
            ok = True
            for k in range(world):
                ins = all_in[k * len(rows):(k + 1) * len(rows)].numpy()
                want, _ = orc.encrypt_batch(N, q, h_np, ins[:, :N], ins[:, N:])
                got = all_e[k * B:(k + 1) * B][rows.cpu()].numpy().view(np.uint16)
                ok = ok and bool(np.array_equal(got, want))
                ok = ok and int(sh.shard_checksums(all_e[k * B:(k + 1) * B].view(torch.uint8), None)[0]) == int(sums[k])
            distinct = not bool(torch.equal(all_e[:B], all_e[B:2 * B]))          # the ranks really own different shards
            q_out.put((ok, distinct, kernel, int(all_e.shape[0]), elapsed > 0.0, len(rows) * world))
        else:
            assert all_e is None and all_in is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_the_hip_kernels_gather_to_root_matches_the_oracle():
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(rk, 2, port, q_out)) for rk in range(2)]
    [p.start() for p in procs]
    import queue
    res, waited = None, 0
    while res is None:                                      # a rank that dies must fail the test at once, not after a silent timeout
        try:
            res = q_out.get(timeout=5)
        except queue.Empty:
            waited += 5
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or waited > 300:
                [p.kill() for p in procs if p.is_alive()]
                pytest.fail("ranks exited with %r before rank 0 reported (waited %d s)" % (dead, waited))
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    ok, distinct, kernel, rows, timed, checked = res
    assert ok and distinct and timed and rows == 2 << 18 and checked >= 500
    assert kernel.startswith("k_encrypt_m")


def _run_bench(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                           "--check-rows", "512"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]
    lines = [l for l in proc.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                     # ONE JSON line, whatever the ranks and RCCL print
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_by_itself_and_gathers():
    """`python bench.py --gpus 2` (the driver's command shape without a launcher) must really run two ranks."""
    d = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--batch-log2", "16", "--power-seconds", "0"])   # no --gather: it runs anyway
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["dist"]["backend"] == "gloo" and d["dist"]["world"] == 2 and len(d["dist"]["ranks"]) == 2
    # every rank reports its own time, kernel times and items: a straggler shows by itself, and the maximum is the line's ms_per_step
    ranks = d["dist"]["ranks"]
    assert all(r["ms_per_step"] > 0 and r["items_per_step"] == 1 << 16 and len(r["kernels_ms"]) == 2 for r in ranks)
    assert max(r["ms_per_step"] for r in ranks) <= d["ms_per_step"] * 1.0001
    assert d["dist"]["straggler"]["slowest_rank"] in (0, 1) and d["dist"]["straggler"]["ms_per_step_max"] == max(r["ms_per_step"] for r in ranks)
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s" and d["mfma"]["unit"].startswith("TOP/s")
    g = d["gather"]
    assert g["kind"] == "gather_to_root" and g["ranks"] == 2 and g["rows"] == 2 << 16
    assert g["rows_equal_local_shard"] and g["every_shard_checksum_matches_its_owner"]
    # whole-job value: both ranks' items over the slowest rank's time
    assert abs(d["value"] - 2 * (1 << 16) * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6


def test_bench_workloads_config_4_and_5_have_a_multi_rank_entry():
    """BASELINE.json configs 4 (N=701 encryptBits) and 5 (verifyKeysInputs with per-item keys) under the same launcher: two ranks
    on device 0 over gloo, oracle-checked rows, the gather-to-root runs without being asked for, and the line says who took part."""
    for wl, unit, kern in (("verify_keys", "key_pairs/s", "k_verify_keys_m"), ("encrypt_n701", "encrypts/s", "k_encrypt_m")):
        d = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--workload", wl, "--batch-log2", "14",
                        "--power-seconds", "0"])
        assert d["n_gpus"] == 2 and d["unit"] == unit and d["scaling"] == "weak"
        assert any(k.startswith(kern) for k in d["kernels_ms"]), d["kernels_ms"]
        assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["int8"]["unit"].startswith("TOP/s")
        assert d["verified_bit_exact_rows"] >= 500
        ranks = d["dist"]["ranks"]
        assert d["dist"]["world"] == 2 and [r["rank"] for r in ranks] == [0, 1] and ranks[0]["pid"] != ranks[1]["pid"]
        assert all(r["ms_per_step"] > 0 and any(k.startswith(kern) for k in r["kernels_ms"]) for r in ranks)
        if wl == "verify_keys":                             # config 5 runs on TRUE key pairs generated on the device: every flag valid
            assert d["keys"]["flags_valid"] == 1 << 14 and d["keys"]["sample_equals_oracle"] and "generated on the device" in d["keys"]["source"]
        g = d["gather"]
        assert g["kind"] == "gather_to_root" and g["ranks"] == 2 and g["rows"] == 2 << 14
        assert g["rows_equal_local_shard"] and g["every_shard_checksum_matches_its_owner"]
        assert abs(d["value"] - 2 * (1 << 14) * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6


def test_bench_refuses_gpus_that_do_not_match_the_world():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
                           "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert proc.returncode != 0 and b"WORLD_SIZE=1" in proc.stderr and not proc.stdout.strip()


def test_rccl_one_rank_process_group_times_and_gathers():
    """RCCL (backend nccl) with the one rank this box allows: init, barrier, all_reduce(MAX), the checksum all_gather and the
    gather-to-root all go through the library."""
    d = _run_bench(["--gpus", "1", "--force-dist", "--gather", "--batch-log2", "16", "--power-seconds", "0"])
    g = d["gather"]
    assert d["dist"]["backend"] == "nccl" and d["dist"]["rccl_version"] and d["dist"]["ranks"][0]["device"] == 0
    assert d["n_gpus"] == 1 and g["backend"] == "nccl" and g["consumed_on"].startswith("GPU 0")
    assert g["rows_equal_local_shard"] and g["every_shard_checksum_matches_its_owner"]


def test_bench_total_batch_is_config_5_as_worded_and_power_is_per_rank():
    """`--gpus 2 --workload verify_keys --total-batch-log2 15`: the TOTAL is split over the ranks (strong scaling, as BASELINE words
    config 5: 2^18 keys over 8 GPUs), and with the sustained phase on every rank reports the power and clock of its own device."""
    d = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--workload", "verify_keys", "--total-batch-log2", "15",
                    "--power-seconds", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["keys"]["key_pairs"] == 1 << 14
    ranks = d["dist"]["ranks"]
    assert all(r["items_per_step"] == 1 << 14 for r in ranks)
    assert all(r["power"] is None or (r["power"]["socket_W"] > 0 and r["power"]["sustained_ms_per_step"] > 0) for r in ranks)
    assert abs(d["value"] - (1 << 15) / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_exits_non_zero_when_the_process_group_does_not_come_up():
    """RCCL refuses two ranks on ONE device: bench.py must fail loudly (no fallback to another backend, no result line)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "0", "--steps", "1", "--warmup", "0",
                           "--no-cpu-baseline", "--batch-log2", "12", "--power-seconds", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          env=env, timeout=300)
    assert proc.returncode != 0 and not proc.stdout.strip()
    assert b"did not come up" in proc.stderr or b"Duplicate GPU" in proc.stderr


def test_bench_four_ranks_report_each_rank():
    """Four ranks (gloo, all on device 0: the box has one GPU and allows six processes on it) through the same launcher the 8-GPU run
    uses: the line carries four rank records with distinct pids, each with its own time and kernel times, and the whole-job value."""
    d = _run_bench(["--gpus", "4", "--dist-backend", "gloo", "--device", "0", "--batch-log2", "14", "--power-seconds", "0"])
    ranks = d["dist"]["ranks"]
    assert d["n_gpus"] == 4 and [r["rank"] for r in ranks] == [0, 1, 2, 3] and len({r["pid"] for r in ranks}) == 4
    assert all(r["ms_per_step"] > 0 and r["items_per_step"] == 1 << 14 and len(r["kernels_ms"]) == 2 for r in ranks)
    assert d["dist"]["straggler"]["ms_per_step_max"] <= d["ms_per_step"] * 1.0001
    assert d["gather"]["ranks"] == 4 and d["gather"]["rows"] == 4 << 14 and d["gather"]["every_shard_checksum_matches_its_owner"]
    assert abs(d["value"] - 4 * (1 << 14) / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
