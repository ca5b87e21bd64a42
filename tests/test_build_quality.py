"""Build-time quality gates for the HIP kernels (CPU only: hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import subprocess

import __graft_entry__ as ge


import pytest

KERNEL_TUS = ("valu_families", "matrix_encrypt", "matrix_decrypt", "matrix_rowimage", "matrix_peritem", "keygen_sampler_pack",
              "ntru_generic")
# the shipped library, and the experiments build (kernel paths 6-10: most 16-byte result stores live there; it is tested for
# bit-exactness too, so its kernels get the same gates): (ASMDIR, EXTRA, fewest wide stores the scan must see)
BUILDS = {"default": ("/tmp/ntru_asm", "", 4), "experiments": ("/tmp/ntru_asm_exp", "-DNTRU_EXPERIMENTS", 60)}


def _asm_usage(build):
    """`make asm` (one .s + one .usage per kernel translation unit under ASMDIR, rebuilt when a source is newer); returns
    the concatenated resource-usage remarks."""
    asm_dir, extra, _ = BUILDS[build]
    src = os.path.join(ge.PKG_DIR, "csrc")
    cmd = ["make", "-C", src, "asm", "ASMDIR=" + asm_dir] + (["EXTRA=" + extra] if extra else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-2000:]
    return "".join(open(os.path.join(asm_dir, t + ".usage")).read() for t in KERNEL_TUS)


@pytest.mark.parametrize("build", sorted(BUILDS))
def test_no_kernel_spills_to_scratch(build):
    """Register spills go to scratch memory = extra HBM traffic (round 1 measured 2.7-3.2x the algorithmic bytes
    before they were removed): every kernel must report ScratchSize 0."""
    text = _asm_usage(build)
    names = re.findall(r"Function Name: (\S+)", text)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
    assert len(names) == len(scratch) and len(names) >= 40, (len(names), len(scratch))
    # (the two fp4 experiment kernels of path 11 keep a few loop-invariant addresses in scratch, 52-80 bytes per lane, outside
    # their loops: measured slower than the int8 kernels anyway, profiles/r04_fp4_product2.txt -- not worth a register diet)
    # (k_verify_keys_m16, the 16-row-tile experiment of path 12: 176 bytes per lane between its two walks, not inside them; measured slower)
    bad = [(n, s) for n, s in zip(names, scratch)
           if s and not (build == "experiments" and ((("k_decrypt_mq" in n or "k_decrypt_m8q" in n) and s <= 96) or ("k_verify_keys_m16" in n and s <= 208)))]
    assert not bad, bad
    for must in ("k_encrypt_t", "k_decrypt_s", "k_encrypt", "k_decrypt", "k_verify_keys", "k_polymul_split", "k_encrypt_wp", "k_decrypt_mp"):
        assert any(must in n for n in names), must
    if build == "experiments":
        for must in ("k_encrypt_m2", "k_encrypt_mc", "k_encrypt_m8", "k_decrypt_m8d", "k_encrypt_w", "k_decrypt_m8q"):
            assert any(must in n for n in names), must


@pytest.mark.parametrize("build", sorted(BUILDS))
def test_no_wide_store_followed_by_a_write_of_its_data_registers(build):
    """gfx950, measured (profiles/archive/r02_hazard_store_x4_soffset.txt): a buffer_store_dwordx4 whose data registers the very next
    instruction overwrites can store the NEW value of the first dword when the memory pipe is busy.  The compiler separates
    the two only when the store's soffset is not a register, so the kernels never pass a scalar offset to their 16-byte
    stores; this scans the generated ISA of every kernel translation unit for the pattern."""
    _asm_usage(build)
    lines = []
    for t in KERNEL_TUS:
        lines += open(os.path.join(BUILDS[build][0], t + ".s")).read().split("\n")
    kern, n, bad = None, 0, []
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
        if not re.search(r"\b(buffer|global|flat|scratch)_store_dwordx[34]\b", line):
            continue
        data = re.search(r"v\[(\d+):(\d+)\]", line)
        lo, hi = int(data.group(1)), int(data.group(2))
        j = i + 1
        while j < len(lines) and (lines[j].strip().startswith(";") or not lines[j].strip()):
            j += 1
        nxt = lines[j].strip()
        n += 1
        w = re.match(r"^v_\w+\s+v\[?(\d+)(?::(\d+))?", nxt)
        if w and not nxt.startswith(("v_cmp", "v_cmpx")):
            a = int(w.group(1)); b = int(w.group(2) or a)
            if not (b < lo or a > hi):
                bad.append((kern, line.strip(), nxt))
    assert n >= BUILDS[build][2], n                       # the scan saw the wide stores of this build
    assert not bad, bad[:5]
