"""Build-time quality gates for the HIP kernels (CPU only: hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import subprocess

import __graft_entry__ as ge


def test_no_kernel_spills_to_scratch():
    """Register spills go to scratch memory = extra HBM traffic (round 1 measured 2.7-3.2x the algorithmic bytes
    before they were removed): every kernel must report ScratchSize 0."""
    out = subprocess.run(["make", "-C", os.path.join(ge.PKG_DIR, "csrc"), "asm"], capture_output=True, text=True,
                         timeout=900)
    text = out.stdout + out.stderr
    names = re.findall(r"Function Name: (\S+)", text)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
    assert len(names) == len(scratch) and len(names) >= 40, (len(names), len(scratch))
    bad = [(n, s) for n, s in zip(names, scratch) if s]
    assert not bad, bad
    for must in ("k_encrypt_t", "k_decrypt_s", "k_encrypt", "k_decrypt", "k_verify_keys", "k_polymul_split"):
        assert any(must in n for n in names), must
