"""The C ABI from C: tests/c/abi_round_trip.c is compiled with plain gcc against include/ntru_engine.h and the in-tree
library (no HIP headers, no Python in the data path) and must reproduce the oracle bit for bit."""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_plain_c_consumer_round_trip(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc on this box")
    pkg = ge.load_package()
    lib = pkg.library_path()
    assert os.path.exists(lib), "the HIP engine library is not built"
    orc_dir = os.path.join(ROOT, "oracle")
    assert os.path.exists(os.path.join(orc_dir, "libntru_oracle.so")), "oracle not built (python -c 'import __graft_entry__ as g; g.build()')"
    exe = str(tmp_path / "abi_round_trip")
    libdir, libname = os.path.dirname(lib), os.path.basename(lib)
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_round_trip.c"), "-o", exe,
                           "-L" + libdir, "-l:" + libname, "-L" + orc_dir, "-lntru_oracle",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + orc_dir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("N=")]
    assert len(lines) == 4 and all(l.endswith("identical") for l in lines), out.stdout
    assert "k_decrypt_m" in lines[0]              # the matrix-core kernels served the headline parameter set
    assert "pinned vs pageable host buffers" in out.stdout and "DIFFERENT" not in out.stdout
    assert "pipeline + device buffers, N=821 B=9001: identical" in out.stdout
    assert "generic family: divide ok, refused divide ok, Euclid example ok" in out.stdout
