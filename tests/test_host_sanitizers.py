"""Sanitizer builds of the HOST side of the engine (sanitizers belong on the CPU build; GPU ASan is not
available on the pool).  abi.hip and ntru_host.hip -- engine life cycle, the three-stage chunk pipeline with its pinned arenas and
staging threads, ntru_multi_* with one host thread per shard, the scratch buffer shared across streams -- are compiled as plain C++
against a test double of the HIP runtime whose streams are asynchronous worker threads (tests/hostcheck/), and a driver pushes
batches through every host-pointer entry point under AddressSanitizer + UBSan and under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_pipeline_under_sanitizers(kind):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    build = subprocess.run(["make", "-C", HERE, kind], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", ASAN_OPTIONS="detect_leaks=1")
    run = subprocess.run([os.path.join(HERE, "hostcheck_" + kind)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "hostcheck ok" in run.stdout
    assert "ThreadSanitizer" not in run.stderr and "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
