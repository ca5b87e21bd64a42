"""N>1 path on CPU: two processes over gloo exercise the sharding, max-over-ranks timing and final gather that
bench.py uses on RCCL.  The product kernels need a GPU, so the CPU oracle stands in as the per-shard compute."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge

pkg = ge.load_package()
sh = pkg.sharding


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 8, 1 << 20, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sh.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert len({sh.shard_seed(1, r) for r in range(8)}) == 8


def _worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ntru_oracle as orc
        N, q, d, total = 17, 32, 2, 12                      # equal shards for world 2 and 4: gather_rows takes equal rows per rank
        rng = np.random.default_rng(3)                      # same global batch on every rank
        h = rng.integers(0, q, N)
        r = np.zeros((total, N), np.uint8)
        for b in range(total):
            perm = rng.permutation(N); r[b, perm[:d]] = 1; r[b, perm[d:2 * d]] = 2
        m = rng.integers(0, 2, (total, N), dtype=np.uint8)
        lo, hi = sh.shard_range(total, rank, world)
        e_local, _ = orc.encrypt_batch(N, q, h, r[lo:hi], m[lo:hi])
        import time as _t
        elapsed, mine = sh.timed_region(lambda: _t.sleep(0.05 * (1 + rank)), lambda: None, dist, None, return_local=True)
        reports, straggler = sh.rank_reports({"rank": rank, "ms_per_step": mine * 1e3, "kernels_ms": {"k": 1.0 + rank},
                                              "power": {"socket_W": 100.0 * (1 + rank), "sclk_MHz": 2400.0}}, dist)
        slow = sh.max_over_ranks(1.0 + rank, dist, None)
        local = torch.from_numpy(e_local.astype(np.int32))
        sums = sh.shard_checksums(local, dist)
        allrows = sh.gather_rows(local, dist, root=0)               # gather-to-root: only rank 0 receives
        if rank == 0:
            e_full, _ = orc.encrypt_batch(N, q, h, r, m)
            per = [int(sh.shard_checksums(allrows[a:b], None)[0]) for a, b in (sh.shard_range(total, k, world) for k in range(world))]
            per_rank_ok = ([x["rank"] for x in reports] == list(range(world)) and all("kernels_ms" in x and "power" in x for x in reports)
                           and straggler["slowest_rank"] == world - 1 and straggler["ms_per_step_max"] == reports[-1]["ms_per_step"]
                           and reports[0]["ms_per_step"] < reports[-1]["ms_per_step"] and elapsed >= mine
                           and elapsed * 1e3 >= straggler["ms_per_step_max"] and straggler["spread"] > 0.3)
            q_out.put((bool(np.array_equal(allrows.numpy(), e_full)) and per == sums.tolist() and per_rank_ok, slow, elapsed >= 0.0, (lo, hi)))
        else:
            assert allrows is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_shard_and_gather_over_gloo_matches_single_process(world):
    """world_size 2 and 4 over gloo on the CPU: shard ranges, max-over-ranks timing, per-rank reports with the straggler named, checksums
    and the gather-to-root -- the plumbing bench.py runs over RCCL with one rank per GPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q_out)) for r in range(world)]
    [p.start() for p in procs]
    res = q_out.get(timeout=180)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    same, slow, ok, span = res
    assert same and slow == float(world) and ok and span == sh.shard_range(12, 0, world)


# ---- bench.py's own launch logic (no GPU, no torch.distributed needed) -------------------------------------------------
import bench


def test_bench_gpus_flag_spawns_a_child_launcher_only_outside_one():
    args = bench.parse_args(["--gpus", "4", "--steps", "2"])
    cmd = bench.launcher_command(args, ["--gpus", "4", "--steps", "2"], {}, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    # inside a launcher (RANK set) or with one GPU nothing is spawned: `rocprofv3 -- python3 bench.py` makes no hop
    assert bench.launcher_command(args, [], {"RANK": "0", "WORLD_SIZE": "4"}) is None
    assert bench.launcher_command(bench.parse_args([]), [], {}) is None
    assert bench.launcher_command(bench.parse_args(["--gpus", "1"]), [], {}) is None


def test_bench_refuses_a_world_that_is_not_gpus():
    assert bench.check_world(bench.parse_args(["--gpus", "2"]), {"WORLD_SIZE": "2", "RANK": "1"}) == 2
    assert bench.check_world(bench.parse_args([]), {}) == 1
    with pytest.raises(SystemExit):
        bench.check_world(bench.parse_args(["--gpus", "8"]), {"WORLD_SIZE": "1", "RANK": "0"})
    with pytest.raises(SystemExit):
        bench.check_world(bench.parse_args(["--gpus", "1"]), {"WORLD_SIZE": "2", "RANK": "0"})


def test_bench_relays_exactly_the_result_line():
    noise = "RCCL version 2.26\n{\"not\": \"it\"}\n" + '{"metric": "x", "value": 1.0, "n_gpus": 2}' + "\ntrailing\n"
    assert json.loads(bench.pick_result_line(noise))["n_gpus"] == 2
    assert bench.pick_result_line("nothing here\n") is None


def test_bench_launcher_runs_the_child_and_returns_its_code(tmp_path, capsys):
    child = tmp_path / "child.py"
    child.write_text("import sys\nprint('banner')\nprint('{\"metric\": \"m\", \"n_gpus\": 2}')\nsys.exit(int(sys.argv[1]))\n")
    import sys as _sys
    assert bench.spawn_ranks([_sys.executable, str(child), "0"]) == 0
    assert json.loads(capsys.readouterr().out.strip())["n_gpus"] == 2
    assert bench.spawn_ranks([_sys.executable, str(child), "3"]) == 3


def test_bench_total_batch_splits_over_the_ranks_and_says_strong_scaling():
    """BASELINE.json words configs 4 and 5 as a TOTAL over 8 GPUs: --total-batch-log2 turns that into the per-rank batch."""
    a = bench.parse_args(["--gpus", "8", "--workload", "verify_keys", "--total-batch-log2", "18"])
    assert a.batch_log2 == 15 and a.total_batch_log2 == 18
    a = bench.parse_args(["--gpus", "8", "--workload", "encrypt_n701", "--total-batch-log2", "23"])
    assert a.batch_log2 == 20
    assert bench.parse_args(["--workload", "verify_keys"]).batch_log2 == 18
    with pytest.raises(SystemExit):
        bench.parse_args(["--gpus", "3", "--total-batch-log2", "18"])
    with pytest.raises(SystemExit):
        bench.parse_args(["--gpus", "2", "--total-batch-log2", "18", "--batch-log2", "17"])


def test_rank_reports_without_a_process_group():
    reports, s = sh.rank_reports({"rank": 0, "ms_per_step": 2.5})
    assert reports == [{"rank": 0, "ms_per_step": 2.5}] and s["slowest_rank"] == 0 and s["spread"] == 0.0
