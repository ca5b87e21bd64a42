"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the header declares, does its
argument checking on the host, and fails loudly (no fallback) without a GPU."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as ge

pkg = ge.load_package()
ROOT = ge.ROOT


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return pkg.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ntru_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ntru_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), "libntru_engine.so does not export %s" % n


def test_supports_matrix(lib):
    ok = [(821, 4096), (701, 8192), (509, 2048), (167, 128), (17, 32), (821, 3), (2, 2), (1920, 65536), (167, 7)]
    bad = [(1, 32), (1921, 2048), (821, 131072), (821, 11), (821, 1), (821, 0), (821, 6000)]
    for N, mod in ok:
        assert lib.ntru_engine_supports(N, mod) == 1, (N, mod)
    for N, mod in bad:
        assert lib.ntru_engine_supports(N, mod) == 0, (N, mod)


def test_no_gpu_means_loud_failure(lib):
    if lib.ntru_engine_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.EngineError) as ei:
        pkg.Engine(0)
    assert ei.value.code == 1 and "no CPU fallback" in str(ei.value)
    ntru = pkg.NTRU({"N": 17, "q": 32, "dr": 2, "h": [1, 2, 3]})
    with pytest.raises(pkg.EngineError):
        ntru.encryptBits([1, 0, 1])


def test_null_engine_is_rejected(lib):
    rc = lib.ntru_encrypt_batch(None, 17, 32, None, None, None, 1, None, None)
    assert rc == 2 and b"NULL" in lib.ntru_last_error()


def test_product_package_never_touches_the_oracle():
    # the shipped path must not import / dlopen anything under oracle/
    for dirpath, _, files in os.walk(ge.PKG_DIR):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".c", ".h", ".mjs", ".js")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in src.replace("SURVEY", ""), "%s mentions the oracle" % fn


def test_host_helpers_match_reference(pure_golden):
    for v in pure_golden["misc"]["trim"]:
        assert pkg.trimPolynomial(v["a"]) == v["out"]
    for v in pure_golden["misc"]["degree"]:
        assert pkg.degree(v["a"]) == v["out"]
    for v in pure_golden["add"]:
        assert pkg.addPolynomials(v["a"], v["b"], v["p"]) == v["out"]
    for v in pure_golden["sampler"]:
        if v.get("error"):
            with pytest.raises(ValueError, match="cannot exceed"):
                pkg.generateCustomArray(v["len"], v["n1"], v["nm1"])
            continue
        it = iter(v["draws"])
        assert pkg.generateCustomArray(v["len"], v["n1"], v["nm1"], rand_u32=lambda: next(it)) == v["out"]
    g = pure_golden["misc"]["stringToBits"][0]
    assert pkg.stringToBits(g["s"]) == g["out"] and pkg.bitsToString(g["out"]) == g["s"]
    for v in pure_golden["misc"]["NqNp"]:
        n = pkg.NTRU({"N": v["N"], "q": v["q"]})
        assert (n.calculateNq(), n.calculateNp()) == (v["Nq"], v["Np"])
    with pytest.raises(ValueError, match="Invalid array length"):
        pkg.expandArray([1, 2, 3], 2)


def test_plain_c_consumer_builds_and_fails_loudly_without_a_gpu(lib, tmp_path):
    """tests/c/abi_round_trip.c (a C host of the ABI: gcc, no HIP headers) must compile and link against the in-tree
    library; on a box without a GPU it must stop at ntru_engine_create with the no-device error, not compute anything."""
    import shutil
    import subprocess
    import torch
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    so = pkg.library_path()
    libdir, libname, orc_dir = os.path.dirname(so), os.path.basename(so), os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "abi_round_trip")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_round_trip.c"), "-o", exe,
                           "-L" + libdir, "-l:" + libname, "-L" + orc_dir, "-lntru_oracle",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + orc_dir])
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_c_abi_gpu.py runs the program")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "no CPU fallback" in out.stderr, out.stdout + out.stderr
    assert "identical" not in out.stdout


def test_no_timing_only_code_paths_in_the_product_sources():
    """Round 5 removed the -DNTRU_ABLATE / -DRI_ABL timing-only variants (kernels with stores, loops or lookups compiled out, wrong
    values on purpose) from the product kernels: what they measured is in EXPERIMENTS.md, the code is in the git history.  The hot
    loops must stay free of them; -DNTRU_STAMPS (phase stamps, right values) is the only diagnostic build."""
    csrc = os.path.join(ROOT, "ntru-circom_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            src = open(os.path.join(csrc, name)).read()
            for word in ("NTRU_ABLATE", "RI_ABL", "ABL_STORE", "NTRU_SAMPLER_ABLATE", "TIMING_ONLY"):
                assert word not in src, (name, word)
