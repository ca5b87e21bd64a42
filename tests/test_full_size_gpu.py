"""Parity at the sizes BASELINE.json names for the per-item kernels (config 5: N=821, q=4096, 2^18 keys; config 4: N=701,
q=8192, 2^20 encryptions per GPU).  A batch of 2^18 makes every persistent grid loop many times over its row blocks
(`rb += gridDim.x`) and sends ntru_invert_key_batch_dev through four chunks of Newton temporaries -- code a 40-item test
never reaches.  Each test compares, on the WHOLE batch and on every output array, the matrix-core family against the
vector-ALU family (ntru_engine_set_kernel_path 4 vs 1: different kernels, different decompositions), checks a
size-independent property where the domain offers one, and checks a strided sample of >= 4096 rows (first and last row
included) against the CPU oracle."""
import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import ntru_oracle as orc

pytestmark = pytest.mark.gpu
pkg = ge.load_package()

N, Q, P, D = 821, 4096, 3, 273
LOGB = 18


@pytest.fixture(scope="module")
def ctx():
    import torch
    eng = pkg.Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    yield torch, eng, torch.device("cuda:0")
    eng.set_stream(None)
    eng.set_kernel_path(0)


def _ternary(torch, dev, gen, B, n, n1, n2):
    """[B][n] int8 rows with n1 ones and n2 minus-ones at random positions."""
    out = torch.zeros((B, n), dtype=torch.int8, device=dev)
    for o in range(0, B, 1 << 16):
        c = min(1 << 16, B - o)
        idx = torch.rand((c, n), device=dev, generator=gen).argsort(dim=1)
        out[o:o + c].scatter_(1, idx[:, :n1], 1)
        out[o:o + c].scatter_(1, idx[:, n1:n1 + n2], -1)
    return out


def _sample_rows(torch, dev, B, n=4096):
    return torch.tensor(sorted(set(list(range(0, B, max(1, B // n))) + [B - 1])), device=dev)


def _host(t, rows):
    a = t[rows].cpu().numpy()
    return a.view(np.uint16) if a.dtype == np.int16 else a


def _u16(torch, gen, dev, B, n, hi):
    return torch.randint(0, hi, (B, n), device=dev, generator=gen, dtype=torch.int32).to(torch.int16)


def test_verify_keys_full_batch(ctx):
    torch, eng, dev = ctx
    B = 1 << LOGB
    gen = torch.Generator(device=dev); gen.manual_seed(51)
    f, g = _ternary(torch, dev, gen, B, N, D, D - 1), _ternary(torch, dev, gen, B, N, D, D)
    fq, h = _u16(torch, gen, dev, B, N, Q), _u16(torch, gen, dev, B, N, Q)
    fp = torch.randint(0, P, (B, N), device=dev, generator=gen, dtype=torch.uint8)
    # a few items that ARE consistent, spread over the batch (first, last, some inside): fq = f = 1 etc.
    for b in (0, 1, B // 3, B - 1):
        f[b] = 0; f[b, 0] = 1; fq[b] = 0; fq[b, 0] = 1; fp[b] = 0; fp[b, 0] = 1
        h[b] = 0; h[b, :N] = (3 * g[b].to(torch.int32) % Q).to(torch.int16)          # h = p * fq * g = 3 g
    outs = {}
    for path in (4, 1):
        eng.set_kernel_path(path)
        o16 = lambda: torch.empty((B, N), dtype=torch.int16, device=dev)
        o8 = lambda: torch.empty((B, N), dtype=torch.uint8, device=dev)
        arrs = [o16(), o16(), o8(), o8(), o16(), o16()]
        flags = torch.empty(B, dtype=torch.uint8, device=dev)
        eng.verify_keys_batch_dev(N, Q, P, f.data_ptr(), g.data_ptr(), fq.data_ptr(), fp.data_ptr(), h.data_ptr(), B,
                                  *[t.data_ptr() for t in arrs], flags.data_ptr())
        torch.cuda.synchronize()
        outs[path] = (arrs + [flags], eng.last_kernel())
    eng.set_kernel_path(0)
    assert outs[4][1] == "k_verify_keys_m" and outs[1][1].startswith("k_verify_keys<")
    names = ("quot_fq", "rem_fq", "quot_fp", "rem_fp", "quot_h", "rem_h", "flags")
    for name, a, b in zip(names, outs[4][0], outs[1][0]):
        assert torch.equal(a, b), "matrix-core and vector-ALU kernels differ on %s" % name
    flags = outs[4][0][6]
    assert int(flags[0]) == 0 and int(flags[B - 1]) == 0 and int(flags[B // 3]) == 0       # the consistent items
    assert int((flags != 0).sum()) >= B - 8                                                  # random operands: invalid
    rows = _sample_rows(torch, dev, B)
    want = orc.verify_keys_batch(N, Q, P, _host(f, rows), _host(g, rows), _host(fq, rows), _host(fp, rows), _host(h, rows))
    for name, t in zip(names, outs[4][0]):
        assert np.array_equal(_host(t, rows), want[name]), name


def test_polymul_split_and_public_key_full_batch(ctx):
    torch, eng, dev = ctx
    B = 1 << LOGB
    gen = torch.Generator(device=dev); gen.manual_seed(52)
    a, b = _u16(torch, gen, dev, B, N, Q), _u16(torch, gen, dev, B, N, Q)
    a[B - 1] = Q - 1; b[B - 1] = Q - 1                                     # worst-case magnitudes in the last row
    g = _ternary(torch, dev, gen, B, N, D, D)
    res = {}
    for path in (4, 1):
        eng.set_kernel_path(path)
        quot, rem, h = (torch.empty((B, N), dtype=torch.int16, device=dev) for _ in range(3))
        eng.polymul_split_dev(N, Q, a.data_ptr(), b.data_ptr(), B, quot.data_ptr(), rem.data_ptr())
        k1 = eng.last_kernel()
        eng.public_key_batch_dev(N, Q, P, a.data_ptr(), g.data_ptr(), B, h.data_ptr())
        k2 = eng.last_kernel()
        torch.cuda.synchronize()
        res[path] = (quot, rem, h, k1, k2)
    eng.set_kernel_path(0)
    assert res[4][3] == "k_polymul_m" and res[4][4] == "k_public_key_m"
    assert res[1][3].startswith("k_polymul_split") and res[1][4].startswith("k_public_key<")
    for i, name in enumerate(("quotient", "remainder", "h")):
        assert torch.equal(res[4][i], res[1][i]), "families differ on %s" % name
    # x = 1 evaluation of c = a * b split by 1 - x^N: sum(rem) = sum(a) * sum(b) (mod q) for every row
    u = lambda t: t.to(torch.int64) & 0xFFFF
    assert torch.equal(u(res[4][1]).sum(1) % Q, (u(a).sum(1) * u(b).sum(1)) % Q)
    rows = _sample_rows(torch, dev, B)
    qo, ro = orc.polymul_split_batch(N, Q, _host(a, rows), _host(b, rows))
    assert np.array_equal(_host(res[4][0], rows), qo) and np.array_equal(_host(res[4][1], rows), ro)
    assert np.array_equal(_host(res[4][2], rows), orc.public_key_batch(N, Q, P, _host(a, rows), _host(g, rows)))


def test_invert_key_full_batch(ctx):
    """2^18 keys = 4 chunks of Newton temporaries.  Whole batch: both families agree on fq, fp and flags; f * fq = 1 (mod q)
    and f * fp = 1 (mod 3) for every unflagged key -- the inverse is unique, so that identity IS parity with the
    reference's Euclidean algorithm.  A few rows (first, last, chunk boundaries) against the Python restatement."""
    from oracle import ntru_keygen as kg
    torch, eng, dev = ctx
    B = 1 << LOGB
    gen = torch.Generator(device=dev); gen.manual_seed(53)
    f = _ternary(torch, dev, gen, B, N, D, D - 1)
    res = {}
    for path in (0, 1):
        eng.set_kernel_path(path)
        fq = torch.empty((B, N), dtype=torch.int16, device=dev)
        fp = torch.empty((B, N), dtype=torch.uint8, device=dev)
        flags = torch.empty(B, dtype=torch.uint8, device=dev)
        eng.invert_key_batch_dev(N, Q, P, f.data_ptr(), B, fq.data_ptr(), fp.data_ptr(), flags.data_ptr())
        torch.cuda.synchronize()
        res[path] = (fq, fp, flags)
    eng.set_kernel_path(0)
    for i, name in enumerate(("fq", "fp", "flags")):
        assert torch.equal(res[0][i], res[1][i]), "families differ on %s" % name
    fq, fp, flags = res[0]
    ok = flags == 0
    assert int(ok.sum()) > B // 2
    one = torch.zeros(N, dtype=torch.int64, device=dev); one[0] = 1
    f16 = (f.to(torch.int32) % Q).to(torch.int16)
    quot, rem = (torch.empty((B, N), dtype=torch.int16, device=dev) for _ in range(2))
    eng.polymul_split_dev(N, Q, f16.data_ptr(), fq.data_ptr(), B, quot.data_ptr(), rem.data_ptr())
    torch.cuda.synchronize()
    assert bool(((rem.to(torch.int64) & 0xFFFF) == one)[ok].all())
    f3 = (f.to(torch.int32) % 3).to(torch.int16)
    fp16 = fp.to(torch.int16)
    eng.polymul_split_dev(N, 3, f3.data_ptr(), fp16.data_ptr(), B, quot.data_ptr(), rem.data_ptr())
    torch.cuda.synchronize()
    assert bool(((rem.to(torch.int64) & 0xFFFF) == one)[ok].all())
    for b in (0, (1 << 16) - 1, 1 << 16, B - 1):
        row = f[b].cpu().numpy().astype(np.int64)
        u2, u3 = kg.is_unit(row, N, 2), kg.is_unit(row, N, 3)
        assert bool(int(flags[b]) & pkg.engine.FLAG_NOT_UNIT_MOD2) == (not u2)
        assert bool(int(flags[b]) & pkg.engine.FLAG_NOT_UNIT_MODP) == (not u3)
        if u2 and u3:
            fq_o, fp_o = kg.load_private_key(row, N, Q, P)
            assert (fq[b].cpu().numpy().view(np.uint16) == fq_o).all() and (fp[b].cpu().numpy() == fp_o).all()


def test_sample_ternary_full_batch(ctx):
    torch, eng, dev = ctx
    B = 1 << LOGB
    key = np.arange(8, dtype=np.uint32) * 0x01000193 + 7
    first = (1 << 33) + 5                                                   # item indices beyond 32 bits
    r = torch.empty((B, N), dtype=torch.uint8, device=dev)
    eng.sample_ternary_dev(N, D, D, P - 1, key, first, B, r.data_ptr())
    torch.cuda.synchronize()
    assert bool(((r == 1).sum(1) == D).all()) and bool(((r == 2).sum(1) == D).all()) and int(r.max()) == 2
    # the same items drawn in two separate launches (other grid, other first_item) are the same rows
    half = torch.empty((B // 2, N), dtype=torch.uint8, device=dev)
    eng.sample_ternary_dev(N, D, D, P - 1, key, first + B // 2, B // 2, half.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(half, r[B // 2:])
    rows = _sample_rows(torch, dev, B)
    got = _host(r, rows)
    for i, b in enumerate(rows.cpu().tolist()):
        assert np.array_equal(got[i], orc.sample_ternary_batch(N, D, D, P - 1, key, first + b, 1)[0]), b


def test_config4_encrypt_2e20(ctx):
    """BASELINE config 4, one GPU's share: N=701, q=8192, 2^20 encryptions.  Whole batch: matrix-core family = ternary
    add path on e and quotientE, linearity in m, the x = 1 identity; 4096 rows against the oracle."""
    import json
    import os
    torch, eng, dev = ctx
    with open(os.path.join(ge.ROOT, "tests", "golden", "scheme_n701_q8192.json")) as fh:
        gold = json.load(fh)
    o, key = gold["options"], gold["keys"][0]
    n, q, d = o["N"], o["q"], o["dr"]
    B = 1 << 20
    gen = torch.Generator(device=dev); gen.manual_seed(54)
    h = torch.tensor(list(key["h"]) + [0] * (n - len(key["h"])), dtype=torch.int32, device=dev).to(torch.int16)
    r = (_ternary(torch, dev, gen, B, n, d, d).to(torch.int16) % 3).to(torch.uint8)          # -1 -> 2 (index.js:89)
    m1 = torch.randint(0, 2, (B, n), dtype=torch.uint8, device=dev, generator=gen)
    m2 = torch.randint(0, 3, (B, n), dtype=torch.uint8, device=dev, generator=gen)
    new16 = lambda: torch.empty((B, n), dtype=torch.int16, device=dev)
    u = lambda t: t.to(torch.int32) & 0xFFFF
    res = {}
    from conftest import has_experiments
    for path in (4, 2, 5, 5) + ((7, 7) if has_experiments(eng) else ()):
        eng.set_kernel_path(path)
        e, qe = new16(), new16()
        eng.encrypt_batch_dev(n, q, h.data_ptr(), r.data_ptr(), m1.data_ptr(), B, e.data_ptr(), qe.data_ptr())
        torch.cuda.synchronize()
        if path >= 5:        # direct-to-LDS operand loads / chunked 16-byte stores: twice each -- the store hazard of
            # profiles/archive/r02_hazard_store_x4_soffset.txt only showed at this size, and the early row loads rest on completion order
            assert eng.last_kernel() == {5: "k_encrypt_md", 7: "k_encrypt_mc"}[path]
            assert torch.equal(e, res[4][0]) and torch.equal(qe, res[4][1])
        else:
            res[path] = (e, qe, eng.last_kernel())
    eng.set_kernel_path(5)                                  # and without the quotient array (other store counts behind the early loads)
    e = new16()
    eng.encrypt_batch_dev(n, q, h.data_ptr(), r.data_ptr(), m1.data_ptr(), B, e.data_ptr(), None)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "k_encrypt_md" and torch.equal(e, res[4][0])
    del e
    eng.set_kernel_path(0)
    assert res[4][2] == "k_encrypt_m" and res[2][2].startswith("k_encrypt_t")
    assert torch.equal(res[4][0], res[2][0]) and torch.equal(res[4][1], res[2][1])
    e1, qe1 = res[4][0], res[4][1]
    del res
    e2 = new16()
    eng.encrypt_batch_dev(n, q, h.data_ptr(), r.data_ptr(), m2.data_ptr(), B, e2.data_ptr(), None)
    torch.cuda.synchronize()
    assert torch.equal((u(e1) - u(e2)) % q, (m1.to(torch.int32) - m2.to(torch.int32)) % q)
    s = lambda t: t.to(torch.int64).sum(dim=1)
    assert torch.equal((s(m1) + s(r) * int(u(h).sum())) % q, s(u(e1)) % q)
    rows = _sample_rows(torch, dev, B)
    e_o, qe_o = orc.encrypt_batch(n, q, h.cpu().numpy().view(np.uint16), _host(r, rows), _host(m1, rows))
    assert np.array_equal(_host(e1, rows), e_o) and np.array_equal(_host(qe1, rows), qe_o)


@pytest.mark.parametrize("n,q,logb", [(821, 4096, 20), (701, 8192, 20)])
def test_fused_pack_kernels_full_batch(ctx, n, q, logb):
    """encryptBits + packOutput(q - 1, N, e) and decryptBits + packOutput(p - 1, N, value) at BASELINE batch sizes (2^20 per GPU): the
    fused kernels (k_encrypt_wp, k_decrypt_mp: every workgroup loops over many row blocks, the image is reused) against the two-launch
    forms on the WHOLE batch, and a strided sample of rows against the CPU oracle."""
    torch, eng, dev = ctx
    B = 1 << logb
    gen = torch.Generator(device=dev); gen.manual_seed(57 + n)
    d = n // 3
    r = (_ternary(torch, dev, gen, B, n, d, d).to(torch.int16) % 3).to(torch.uint8)       # d ones, d twos
    m = torch.randint(0, 2, (B, n), device=dev, generator=gen, dtype=torch.uint8)
    h = _u16(torch, gen, dev, 1, n, q)[0]
    f = _ternary(torch, dev, gen, 1, n, d, d - 1)[0]
    fp = torch.randint(0, 3, (n,), device=dev, generator=gen, dtype=torch.uint8)
    bits = (q - 1).bit_length()
    os_e, os_v = max(3, -(-n // (252 // bits))), max(3, -(-n // 126))
    e = torch.empty((B, n), dtype=torch.int16, device=dev)
    pe1, pe2 = (torch.empty((B, os_e, 4), dtype=torch.int64, device=dev) for _ in range(2))
    eng.encrypt_pack_batch_dev(n, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), pe1.data_ptr())       # two launches
    eng.encrypt_pack_batch_dev(n, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, None, pe2.data_ptr())
    torch.cuda.synchronize()
    assert eng.last_kernel() == "k_encrypt_wp" and torch.equal(pe1, pe2)
    v = torch.empty((B, n), dtype=torch.uint8, device=dev)
    pv1, pv2 = (torch.empty((B, os_v, 4), dtype=torch.int64, device=dev) for _ in range(2))
    eng.decrypt_batch_dev(n, q, P, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr())
    eng.pack_bytes_batch_dev(P - 1, n, v.data_ptr(), B, pv1.data_ptr())
    eng.decrypt_pack_batch_dev(n, q, P, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, None, pv2.data_ptr())
    torch.cuda.synchronize()
    assert eng.last_kernel() == "k_decrypt_mp" and torch.equal(pv1, pv2)
    rows = _sample_rows(torch, dev, B, 1024)
    e_o = orc.encrypt_batch(n, q, _host(h, slice(None)), _host(r, rows), _host(m, rows), want_quot=False)[0]
    assert np.array_equal(_host(e, rows), e_o)
    assert np.array_equal(pe2[rows].cpu().numpy().view(np.uint64), orc.pack_batch(q - 1, n, e_o).view(np.uint64).reshape(len(rows), os_e, 4))
    v_o = orc.decrypt_batch(n, q, P, f.cpu().numpy(), fp.cpu().numpy(), e_o, want_witness=False)[0]
    assert np.array_equal(pv2[rows].cpu().numpy().view(np.uint64),
                          orc.pack_batch(P - 1, n, v_o.astype(np.uint16)).view(np.uint64).reshape(len(rows), os_v, 4))


def test_config3_whole_batch_against_the_oracle(ctx):
    """BASELINE config 3 -- the configuration the headline metric is quoted on -- with EVERY row against the CPU oracle: 2^20 round
    trips at N=821, q=4096 under the golden key, all six witness arrays of the HIP path compared with orc.encrypt_batch /
    orc.decrypt_batch (exact-integer mode) row by row, in chunks of 2^16 rows on the host's cores.  The other tests of this size
    compare a strided sample with the oracle and the whole batch between kernel families; this one closes the gap."""
    import threading
    import bench
    torch, eng, dev = ctx
    o, h_np, f_np, fp_np = bench.load_key("n821_q4096")
    n, q, p, d = o["N"], o["q"], o["p"], o["dr"]
    B = 1 << 20
    r, m = bench.make_inputs(torch, dev, B, n, d, 20240)
    h = torch.from_numpy(h_np.view(np.int16)).to(dev); f = torch.from_numpy(f_np).to(dev); fp = torch.from_numpy(fp_np).to(dev)
    n16 = lambda: torch.empty((B, n), dtype=torch.int16, device=dev)
    n8 = lambda: torch.empty((B, n), dtype=torch.uint8, device=dev)
    e, quotE, value, quot1, rem1, quot2 = n16(), n16(), n8(), n16(), n16(), n8()
    eng.set_kernel_path(0)
    eng.encrypt_batch_dev(n, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), quotE.data_ptr())
    k_enc = eng.last_kernel()
    eng.decrypt_batch_dev(n, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, value.data_ptr(), quot1.data_ptr(), rem1.data_ptr(), quot2.data_ptr())
    k_dec = eng.last_kernel()
    torch.cuda.synchronize()
    assert k_enc.startswith("k_encrypt_m") and k_dec.startswith("k_decrypt_m")           # the default (headline) kernels
    threads = max(1, min(16, len(__import__("os").sched_getaffinity(0))))
    chunk, bad, rows_done = 1 << 16, [], []
    for c0 in range(0, B, chunk):
        sl = slice(c0, c0 + chunk)
        ins = (r[sl].cpu().numpy(), m[sl].cpu().numpy())
        got = {k: _host(t, sl) for k, t in (("e", e), ("quotE", quotE), ("value", value), ("quot1", quot1), ("rem1", rem1), ("quot2", quot2))}

        def check(lo, hi):
            try:
                e_o, qe_o = orc.encrypt_batch(n, q, h_np, ins[0][lo:hi], ins[1][lo:hi])
                v_o, q1_o, r1_o, q2_o = orc.decrypt_batch(n, q, p, f_np, fp_np, e_o)
                for k, w in (("e", e_o), ("quotE", qe_o), ("value", v_o), ("quot1", q1_o), ("rem1", r1_o), ("quot2", q2_o)):
                    if not np.array_equal(got[k][lo:hi], w):
                        bad.append((k, c0 + lo))
                rows_done.append(hi - lo)
            except BaseException as exc:                     # a worker that dies must not read as "nothing differed"
                bad.append(("raised %r" % (exc,), c0 + lo))

        cuts = np.linspace(0, chunk, threads + 1).astype(int)
        ths = [threading.Thread(target=check, args=(int(cuts[i]), int(cuts[i + 1]))) for i in range(threads)]
        [t.start() for t in ths]; [t.join() for t in ths]
        assert not bad, bad[:4]
    assert sum(rows_done) == B
    # q = 4096 is 1 mod 3: the reference's centred lift is off there, so the round trip is NOT asserted -- parity is


def test_config5_true_keys_generated_on_the_device(ctx):
    """BASELINE config 5 on VALID keys (what key generation is for): 2^18 key pairs sampled, inverted and completed on
    the device (index.js:51-79), then verifyKeysInputs (index.js:141-197) over them: no flag on any key, both kernel families agree on
    the whole batch, a strided sample of the keys (h = p fq g; f fq = 1, f fp = 1 by the oracle's own products) and of all six
    witness arrays equals the oracle."""
    import bench
    torch, eng, dev = ctx
    o, _, _, _ = bench.load_key("n821_q4096")
    B = 1 << LOGB
    eng.set_kernel_path(0)
    f, g, fq, fp, h, info = bench.generate_key_pairs(torch, eng, dev, o, B, 0)
    assert info["key_pairs"] == B
    # every f is its own draw with the reference's weights; the redraws replaced whole rows
    assert bool(((f == 1).sum(1) == o["df"]).all()) and bool(((f == -1).sum(1) == o["df"] - 1).all())
    assert bool(((g == 1).sum(1) == o["dg"]).all()) and bool(((g == -1).sum(1) == o["dg"]).all())
    outs = {}
    for path in (4, 1):
        eng.set_kernel_path(path)
        o16 = lambda: torch.empty((B, N), dtype=torch.int16, device=dev)
        o8 = lambda: torch.empty((B, N), dtype=torch.uint8, device=dev)
        arrs = [o16(), o16(), o8(), o8(), o16(), o16()]
        flags = torch.empty(B, dtype=torch.uint8, device=dev)
        eng.verify_keys_batch_dev(N, Q, P, f.data_ptr(), g.data_ptr(), fq.data_ptr(), fp.data_ptr(), h.data_ptr(), B,
                                  *[t.data_ptr() for t in arrs], flags.data_ptr())
        torch.cuda.synchronize()
        outs[path] = (arrs + [flags], eng.last_kernel())
    eng.set_kernel_path(0)
    assert outs[4][1] == "k_verify_keys_m" and outs[1][1].startswith("k_verify_keys<")
    names = ("quot_fq", "rem_fq", "quot_fp", "rem_fp", "quot_h", "rem_h", "flags")
    for name, a, b in zip(names, outs[4][0], outs[1][0]):
        assert torch.equal(a, b), "matrix-core and vector-ALU kernels differ on %s" % name
    arrs = outs[4][0]
    assert int((arrs[6] != 0).sum()) == 0                                   # flags_valid = B
    one16 = torch.zeros(N, dtype=torch.int16, device=dev); one16[0] = 1
    assert bool((arrs[1] == one16).all()) and bool((arrs[3] == one16.to(torch.uint8)).all())     # f fq = 1 (mod q), f fp = 1 (mod 3): EVERY key
    assert torch.equal(arrs[5], h)                                          # the remainder of (p fq) g IS h, every key
    rows = _sample_rows(torch, dev, B, 1024)
    ok, want = bench.check_keys_against_oracle(orc, N, Q, P, [_host(t, rows) for t in (f, g, fq, fp, h)])
    assert ok
    for name, t in zip(names, arrs):
        assert np.array_equal(_host(t, rows), want[name]), name


def test_generated_keys_redraw_non_units(ctx):
    """bench.generate_key_pairs draws a private key again when its f is not a unit (index.js:51-66 retries the same way).  With
    N = 821 that practically never happens, so the path is forced: a FIRST draw with df minus ones has f(1) = 0 -- no unit anywhere --
    and every row must come out valid after one redraw with the reference's weights."""
    import bench
    torch, eng, dev = ctx
    for profile, B in (("n167_q128", 300), ("n821_q4096", 1000)):
        o, _, _, _ = bench.load_key(profile)
        n, q, p = o["N"], o["q"], o["p"]
        f, g, fq, fp, h, info = bench.generate_key_pairs(torch, eng, dev, o, B, 7, first_draw_minus=o["df"])
        assert info["non_units_redrawn"] >= B                               # every first draw was rejected
        assert bool(((f == 1).sum(1) == o["df"]).all()) and bool(((f == -1).sum(1) == o["df"] - 1).all())
        ok, want = bench.check_keys_against_oracle(orc, n, q, p, [_host(t, slice(None)) for t in (f, g, fq, fp, h)])
        assert ok and not want["flags"].any()
