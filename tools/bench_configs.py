#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs other than the headline (1 GPU, inputs resident in HBM).
Prints one JSON object per config; bench.py remains the contract benchmark."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

pkg = ge.load_package()
dev = torch.device("cuda", 0)
eng = pkg.Engine(0)
stream = torch.cuda.current_stream()
eng.set_stream(stream.cuda_stream)


from oracle import ntru_oracle as orc  # noqa: E402  (the checker: every config below compares a strided sample with it)

N_CHECK = 64


def sample_rows(B):
    return torch.tensor(sorted(set(list(range(0, B, max(1, B // N_CHECK))) + [B - 1])), device=dev)


def host(t, rows):
    a = t[rows].contiguous().cpu().numpy()
    return a.view(np.uint16) if a.dtype == np.int16 else a


def timed(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record(stream)
    for k in range(steps):
        fn()
        ev[k + 1].record(stream)
    torch.cuda.synchronize()
    return float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]))


def encrypt_config(profile, logB):
    o, h_np, f_np, fp_np = bench.load_key(profile)
    N, q, d = o["N"], o["q"], o["dr"]
    B = 1 << logB
    r, m = bench.make_inputs(torch, dev, B, N, d, 1)
    h = torch.from_numpy(h_np.view(np.int16)).to(dev)
    e = torch.empty((B, N), dtype=torch.int16, device=dev)
    quot = torch.empty((B, N), dtype=torch.int16, device=dev)
    ms = timed(lambda: eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), quot.data_ptr()))
    rows = sample_rows(B)
    e_o, q_o = orc.encrypt_batch(N, q, h_np, host(r, rows), host(m, rows))
    ok = bool(np.array_equal(host(e, rows), e_o) and np.array_equal(host(quot, rows), q_o))
    return {"config": "N=%d q=%d batch=2^%d encryptBits (full witness), 1 GPU" % (N, q, logB), "kernel": eng.last_kernel(),
            "ms": ms, "encrypts_per_s": B / (ms * 1e-3), "hbm_GBps": 6 * N * B / (ms * 1e-3) / 1e9,
            "rows_equal_oracle": ok, "rows_checked": int(rows.numel())}


def verify_config(profile, logB):
    o, _, _, _ = bench.load_key(profile)
    N, q, p, d = o["N"], o["q"], o["p"], o["dr"]
    B = 1 << logB
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    tern = lambda: (torch.randint(0, 3, (B, N), device=dev, generator=gen) - 1).to(torch.int8)
    f, g = tern(), tern()
    fq = torch.randint(0, q, (B, N), device=dev, generator=gen).to(torch.int16)
    hh = torch.randint(0, q, (B, N), device=dev, generator=gen).to(torch.int16)
    fp = torch.randint(0, p, (B, N), device=dev, generator=gen).to(torch.uint8)
    o16 = lambda: torch.empty((B, N), dtype=torch.int16, device=dev)
    o8 = lambda: torch.empty((B, N), dtype=torch.uint8, device=dev)
    outs = [o16(), o16(), o8(), o8(), o16(), o16()]
    flags = torch.empty(B, dtype=torch.uint8, device=dev)
    ms = timed(lambda: eng.verify_keys_batch_dev(N, q, p, f.data_ptr(), g.data_ptr(), fq.data_ptr(), fp.data_ptr(),
                                                 hh.data_ptr(), B, *[t.data_ptr() for t in outs], flags.data_ptr()))
    rows = sample_rows(B)
    want = list(orc.verify_keys_batch(N, q, p, host(f, rows), host(g, rows), host(fq, rows), host(fp, rows), host(hh, rows)).values())
    got = [host(t, rows) for t in outs] + [host(flags, rows)]
    ok = all(np.array_equal(a, b) for a, b in zip(got, want))
    return {"config": "N=%d q=%d verifyKeysInputs batch=2^%d synthetic per-item keys, 1 GPU" % (N, q, logB),
            "kernel": eng.last_kernel(), "ms": ms, "keys_per_s": B / (ms * 1e-3), "hbm_GBps": 17 * N * B / (ms * 1e-3) / 1e9,
            "rows_equal_oracle": bool(ok), "rows_checked": int(rows.numel())}


def sampler_config(profile, logB):
    o, _, _, _ = bench.load_key(profile)
    N, d, p = o["N"], o["dr"], o["p"]
    B = 1 << logB
    r = torch.empty((B, N), dtype=torch.uint8, device=dev)
    key = np.arange(8, dtype=np.uint32) + 1
    per_rounds, ok = {}, True
    for rounds in (8, 12, 20):                               # ntru_engine_set_sampler_rounds: ChaCha8 / ChaCha12 / ChaCha20 (default, last)
        eng.set_sampler_rounds(rounds)
        ms = timed(lambda: eng.sample_ternary_dev(N, d, d, p - 1, key, 0, B, r.data_ptr()))
        first = orc.sample_ternary_batch(N, d, d, p - 1, key, 0, 8, rounds=rounds)
        last = orc.sample_ternary_batch(N, d, d, p - 1, key, B - 8, 8, rounds=rounds)
        ok = ok and bool(np.array_equal(r[:8].cpu().numpy(), first) and np.array_equal(r[B - 8:].cpu().numpy(), last))
        per_rounds[str(rounds)] = {"ms": ms, "samples_per_s": B / (ms * 1e-3)}
    return {"config": "N=%d d=%d batch=2^%d on-device generateCustomArray (ChaCha20 draws), 1 GPU" % (N, d, logB),
            "kernel": eng.last_kernel(), "ms": ms, "samples_per_s": B / (ms * 1e-3), "hbm_GBps": N * B / (ms * 1e-3) / 1e9,
            "per_chacha_rounds": per_rounds, "rows_equal_oracle": ok, "rows_checked": 48}


def keygen_config(profile, logB):
    """SURVEY.md 8d config 5 with REAL keys: f sampled on the device (df ones, df - 1 minus ones), inverted mod q and
    mod p, g sampled, h = p*fq*g; then verifyKeysInputs over the generated keys must raise no flag."""
    o, _, _, _ = bench.load_key(profile)
    N, q, p, df, dg = o["N"], o["q"], o["p"], o["df"], o["dg"]
    B = 1 << logB
    key = (np.arange(8, dtype=np.uint32) * 0x85EBCA6B + 7).astype(np.uint32)
    fs = torch.empty((B, N), dtype=torch.uint8, device=dev); gs = torch.empty((B, N), dtype=torch.uint8, device=dev)
    eng.sample_ternary_dev(N, df, df - 1, 255, key, 0, B, fs.data_ptr())          # 255 = -1 as int8
    eng.sample_ternary_dev(N, dg, dg, 255, key, 1 << 40, B, gs.data_ptr())
    f, g = fs.view(torch.int8), gs.view(torch.int8)
    fq = torch.empty((B, N), dtype=torch.int16, device=dev); fp = torch.empty((B, N), dtype=torch.uint8, device=dev)
    fl = torch.empty(B, dtype=torch.uint8, device=dev); h = torch.empty((B, N), dtype=torch.int16, device=dev)

    def gen():                                               # EVERYTHING a key pair takes: both draws, both inversions, h
        eng.sample_ternary_dev(N, df, df - 1, 255, key, 0, B, fs.data_ptr())
        eng.sample_ternary_dev(N, dg, dg, 255, key, 1 << 40, B, gs.data_ptr())
        eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, fq.data_ptr(), fp.data_ptr(), fl.data_ptr())
        eng.public_key_batch_dev(N, q, p, fq.data_ptr(), g.data_ptr(), B, h.data_ptr())

    def inv_h():
        eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, fq.data_ptr(), fp.data_ptr(), fl.data_ptr())
        eng.public_key_batch_dev(N, q, p, fq.data_ptr(), g.data_ptr(), B, h.data_ptr())
    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=dev)
    eng2 = pkg.Engine(0) if "--no-pipeline" not in sys.argv else None

    def gen_side():                                          # g drawn by a second engine on a side stream, beside f's draw and inversions
        side.wait_stream(main_stream)
        eng2.set_stream(side.cuda_stream)
        eng2.sample_ternary_dev(N, dg, dg, 255, key, 1 << 40, B, gs.data_ptr())
        eng.sample_ternary_dev(N, df, df - 1, 255, key, 0, B, fs.data_ptr())
        eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, fq.data_ptr(), fp.data_ptr(), fl.data_ptr())
        main_stream.wait_stream(side)
        eng.public_key_batch_dev(N, q, p, fq.data_ptr(), g.data_ptr(), B, h.data_ptr())
    ms_inv_h = timed(inv_h, steps=3, warmup=1)
    ms = timed(gen, steps=3, warmup=1)
    ms_side = timed(gen_side, steps=3, warmup=1) if "--no-pipeline" not in sys.argv else None     # (second stream: kept out of the rocprofv3 passes)
    units = int((fl == 0).sum())
    o16 = lambda: torch.empty((B, N), dtype=torch.int16, device=dev)
    o8 = lambda: torch.empty((B, N), dtype=torch.uint8, device=dev)
    outs = [o16(), o16(), o8(), o8(), o16(), o16()]
    vflags = torch.empty(B, dtype=torch.uint8, device=dev)
    vms = timed(lambda: eng.verify_keys_batch_dev(N, q, p, f.data_ptr(), g.data_ptr(), fq.data_ptr(), fp.data_ptr(), h.data_ptr(),
                                                  B, *[t.data_ptr() for t in outs], vflags.data_ptr()), steps=10, warmup=5)
    bad = int(((vflags != 0) & (fl == 0)).sum())
    # the generated keys against the oracle: h = p fq g and the verify_keys witness of a strided sample (which holds f fq = 1, f fp = 1)
    rows = sample_rows(B)
    h_o = orc.public_key_batch(N, q, p, host(fq, rows), host(g, rows))
    want = list(orc.verify_keys_batch(N, q, p, host(f, rows), host(g, rows), host(fq, rows), host(fp, rows), host(h, rows)).values())
    got = [host(t, rows) for t in outs] + [host(vflags, rows)]
    ok = bool(np.array_equal(host(h, rows), h_o)) and all(np.array_equal(a, b) for a, b in zip(got, want))
    return {"config": "N=%d q=%d key generation batch=2^%d (sample f, g; invert mod q and mod p; h) then verifyKeysInputs on the "
                      "generated keys, 1 GPU" % (N, q, logB), "keygen_ms": ms, "keys_per_s": B / (ms * 1e-3), "invert_and_h_ms": ms_inv_h,
            "g_drawn_on_a_side_stream_ms": ms_side, "units": units,
            "verify_ms": vms, "verify_keys_per_s": B / (vms * 1e-3), "verify_flags_on_valid_keys": bad,
            "rows_equal_oracle": ok, "rows_checked": int(rows.numel())}


def pipeline_dev_config(profile, logB, log_chunk=19):
    """Device-resident sampled round trip (no PCIe): generateCustomArray -> encryptBits -> decryptBits (value only), chunk by
    chunk; m and every result stay in HBM.  "serial": the three kernels of a chunk one after the other on ONE stream (what
    ntru_pipeline_batch enqueues on its compute stream); "overlapped": the sampler of chunk k+1 on a SECOND stream next to the
    encrypt / decrypt of chunk k (events both ways).  Socket power and shader clock over 2 s of each are reported: the
    matrix kernels run at the power cap, so a co-running VALU-bound sampler has no free watts to run on."""
    import time
    o, h_np, f_np, fp_np = bench.load_key(profile)
    N, q, p, d = o["N"], o["q"], o["p"], o["dr"]
    B, C = 1 << logB, 1 << log_chunk
    key = np.arange(8, dtype=np.uint32) + 11
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev, generator=gen)
    h = torch.from_numpy(h_np.view(np.int16)).to(dev); f = torch.from_numpy(f_np).to(dev); fp = torch.from_numpy(fp_np).to(dev)
    r = [torch.empty((C, N), dtype=torch.uint8, device=dev) for _ in range(2)]          # two r buffers: chunk k and chunk k+1
    e = torch.empty((C, N), dtype=torch.int16, device=dev)
    value = torch.empty((B, N), dtype=torch.uint8, device=dev)
    s_main, s_samp = torch.cuda.Stream(), torch.cuda.Stream()
    nch = B // C

    def chunk_rest(k):
        eng.encrypt_batch_dev(N, q, h.data_ptr(), r[k & 1].data_ptr(), m[k * C:].data_ptr(), C, e.data_ptr(), None)
        eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), C, value[k * C:].data_ptr())

    def serial():
        eng.set_stream(s_main.cuda_stream)
        for k in range(nch):
            eng.sample_ternary_dev(N, d, d, p - 1, key, k * C, C, r[k & 1].data_ptr())
            chunk_rest(k)

    def overlapped():
        sampled = [torch.cuda.Event() for _ in range(nch)]
        used = [torch.cuda.Event() for _ in range(nch)]
        for k in range(nch):
            eng.set_stream(s_samp.cuda_stream)
            if k >= 2:
                s_samp.wait_event(used[k - 2])                 # r[k & 1] is free once chunk k-2 has been encrypted
            eng.sample_ternary_dev(N, d, d, p - 1, key, k * C, C, r[k & 1].data_ptr())
            sampled[k].record(s_samp)
            eng.set_stream(s_main.cuda_stream)
            s_main.wait_event(sampled[k])
            eng.encrypt_batch_dev(N, q, h.data_ptr(), r[k & 1].data_ptr(), m[k * C:].data_ptr(), C, e.data_ptr(), None)
            used[k].record(s_main)
            eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), C, value[k * C:].data_ptr())

    def wall(fn, seconds=2.0):
        fn(); torch.cuda.synchronize()
        sampler = bench.PowerSampler(bench.device_sysfs_dir(0)); sampler.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            fn(); n += 1
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        pw = sampler.summary(t0, t1) or {}
        return (t1 - t0) / n * 1e3, pw.get("socket_W"), pw.get("sclk_MHz")
    rows = torch.tensor([0, 1, C - 1], device=dev)
    k = nch - 1

    def replay(rounds):                                      # oracle replay of a few rows of the last chunk: the stream position is the item index
        r_o = orc.sample_ternary_batch(N, d, d, p - 1, key, k * C + C - 8, 8, rounds=rounds)
        r_o = np.concatenate([orc.sample_ternary_batch(N, d, d, p - 1, key, k * C, 2, rounds=rounds), r_o[-1:]])
        e_o, _ = orc.encrypt_batch(N, q, h_np, r_o, host(m[k * C:(k + 1) * C], rows))
        v_o = orc.decrypt_batch(N, q, p, f_np, fp_np, e_o)[0]
        return bool(np.array_equal(host(value[k * C:(k + 1) * C], rows), v_o))
    reduced, ok = {}, True
    for rounds in (8, 12):                                   # the serial pipeline at the reduced round counts of the sampler
        eng.set_sampler_rounds(rounds)
        ms_r, w_r, f_r = wall(serial)
        ok = ok and replay(rounds)
        reduced[str(rounds)] = {"ms": ms_r, "round_trips_per_s": B / (ms_r * 1e-3), "socket_W": w_r, "sclk_MHz": f_r}
    eng.set_sampler_rounds(20)
    ms_s, w_s, f_s = wall(serial)
    v_serial = value.clone()
    ms_o, w_o, f_o = wall(overlapped)
    same = bool(torch.equal(v_serial, value))
    eng.set_stream(stream.cuda_stream)
    ok = ok and replay(20)
    return {"config": "N=%d q=%d batch=2^%d device-resident sampled round trip (sampler -> encryptBits -> decryptBits, value only), "
                      "chunks of 2^%d, 1 GPU" % (N, q, logB, log_chunk),
            "serial": {"ms": ms_s, "round_trips_per_s": B / (ms_s * 1e-3), "socket_W": w_s, "sclk_MHz": f_s},
            "sampler_on_second_stream": {"ms": ms_o, "round_trips_per_s": B / (ms_o * 1e-3), "socket_W": w_o, "sclk_MHz": f_o,
                                         "values_equal_serial": same},
            "serial_at_reduced_sampler_rounds": reduced,
            "rows_equal_oracle": ok, "rows_checked": 9}


def decrypt_pack_config(profile, logB):
    """decryptBits (value only) followed by packOutput(p - 1, N, value): two kernels with the values through HBM, against the fused
    k_decrypt_mp that writes only the packed field elements (32 ceil(N / 126) bytes per item instead of N + 32 ceil(N / 126))."""
    o, _, f_np, fp_np = bench.load_key(profile)
    N, q, p = o["N"], o["q"], o["p"]
    B = 1 << logB
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=gen).to(torch.int16)
    f = torch.from_numpy(f_np).to(dev); fp = torch.from_numpy(fp_np).to(dev)
    os_ = max(3, -(-N // 126))
    v = torch.empty((B, N), dtype=torch.uint8, device=dev)
    packed = torch.empty((B, os_, 4), dtype=torch.int64, device=dev); packed2 = torch.empty_like(packed)

    def separate():
        eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr())
        eng.pack_bytes_batch_dev(p - 1, N, v.data_ptr(), B, packed.data_ptr())
    ms_sep = timed(separate)
    ms_fused = timed(lambda: eng.decrypt_pack_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, None, packed2.data_ptr()))
    kern = eng.last_kernel()
    rows = sample_rows(B)
    v_o = orc.decrypt_batch(N, q, p, f_np, fp_np, host(e, rows), want_witness=False)[0]
    want = orc.pack_batch(p - 1, N, v_o.astype(np.uint16)).view(np.uint64)
    ok = bool(np.array_equal(packed2[rows].cpu().numpy().view(np.uint64), want.reshape(len(rows), os_, 4))) and bool(torch.equal(packed, packed2))
    return {"config": "N=%d q=%d batch=2^%d decryptBits + packOutput(%d, N, value), 1 GPU" % (N, q, logB, p - 1), "kernel": kern,
            "ms": ms_fused, "items_per_s": B / (ms_fused * 1e-3), "separate_kernels_ms": ms_sep,
            "bytes_written_per_item": {"fused": 32 * os_, "separate": N + 32 * os_},
            "rows_equal_oracle": ok, "rows_checked": int(rows.numel())}


def encrypt_pack_config(profile, logB):
    """encryptBits (e only) followed by packOutput(q - 1, N, e): two kernels with e through HBM, against the fused k_encrypt_wp that
    writes only the packed field elements (32 ceil(N / per) bytes per item instead of 2 N + 32 ceil(N / per))."""
    o, h_np, _, _ = bench.load_key(profile)
    N, q, p, d = o["N"], o["q"], o["p"], o["dr"]
    B = 1 << logB
    gen = torch.Generator(device=dev); gen.manual_seed(12)
    key = (np.arange(8, dtype=np.uint32) * 0x9E3779B1 + 3).astype(np.uint32)
    r = torch.empty((B, N), dtype=torch.uint8, device=dev)
    eng.sample_ternary_dev(N, d, d, p - 1, key, 0, B, r.data_ptr())
    m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev, generator=gen)
    h = torch.from_numpy(h_np.view(np.int16)).to(dev)
    bits = (q - 1).bit_length(); per = 252 // bits
    os_ = max(3, -(-N // per))
    e = torch.empty((B, N), dtype=torch.int16, device=dev)
    packed = torch.empty((B, os_, 4), dtype=torch.int64, device=dev); packed2 = torch.empty_like(packed)
    ms_sep = timed(lambda: eng.encrypt_pack_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), packed.data_ptr()))
    ms_fused = timed(lambda: eng.encrypt_pack_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, None, packed2.data_ptr()))
    kern = eng.last_kernel()
    rows = sample_rows(B)
    e_o = orc.encrypt_batch(N, q, h_np, host(r, rows), host(m, rows), want_quot=False)[0]
    want = orc.pack_batch(q - 1, N, e_o).view(np.uint64)
    ok = bool(np.array_equal(packed2[rows].cpu().numpy().view(np.uint64), want.reshape(len(rows), os_, 4))) and bool(torch.equal(packed, packed2))
    return {"config": "N=%d q=%d batch=2^%d encryptBits + packOutput(%d, N, e), 1 GPU" % (N, q, logB, q - 1), "kernel": kern,
            "ms": ms_fused, "items_per_s": B / (ms_fused * 1e-3), "separate_kernels_ms": ms_sep,
            "bytes_written_per_item": {"fused": 32 * os_, "separate": 2 * N + 32 * os_},
            "rows_equal_oracle": ok, "rows_checked": int(rows.numel())}


def add_config(N, q, logB):
    B = 1 << logB
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    a = torch.randint(0, q, (B, N), device=dev, generator=gen).to(torch.int16)
    b = torch.randint(0, q, (B, N), device=dev, generator=gen).to(torch.int16)
    out = torch.empty_like(a)
    ms = timed(lambda: eng.add_batch_dev(N, q, a.data_ptr(), b.data_ptr(), B, out.data_ptr()), steps=10)
    ok = bool(torch.equal(out.to(torch.int32) & 0xFFFF, ((a.to(torch.int32) & 0xFFFF) + (b.to(torch.int32) & 0xFFFF)) % q))
    return {"config": "N=%d q=%d batch=2^%d addPolynomials on ciphertexts (elementwise), 1 GPU" % (N, q, logB), "ms": ms,
            "adds_per_s": B / (ms * 1e-3), "hbm_GBps": 6 * N * B / (ms * 1e-3) / 1e9, "hbm_frac_of_8TBps": 6 * N * B / (ms * 1e-3) / 8e12,
            "correct": ok}


def polymul_config(N, q, logB):
    """multiplyPolynomials + dividePolynomials by I (index.js:319-401) for B independent operand pairs, both < q."""
    B = 1 << logB
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    a = torch.randint(0, q, (B, N), device=dev, generator=gen).to(torch.int16)
    b = torch.randint(0, q, (B, N), device=dev, generator=gen).to(torch.int16)
    quot = torch.empty((B, N), dtype=torch.int16, device=dev); rem = torch.empty((B, N), dtype=torch.int16, device=dev)
    out = {}
    for path, name in ((0, "auto"), (1, "vector ALU (packed MAC)")):
        eng.set_kernel_path(path)
        ms = timed(lambda: eng.polymul_split_dev(N, q, a.data_ptr(), b.data_ptr(), B, quot.data_ptr(), rem.data_ptr()))
        out[name] = {"kernel": eng.last_kernel(), "ms": ms, "products_per_s": B / (ms * 1e-3)}
        if path == 0:
            rows = sample_rows(B)
            q_o, r_o = orc.polymul_split_batch(N, q, host(a, rows), host(b, rows))
            ok = bool(np.array_equal(host(quot, rows), q_o) and np.array_equal(host(rem, rows), r_o))
    eng.set_kernel_path(0)
    return {"config": "N=%d q=%d batch=2^%d multiplyPolynomials + split by I, per-item operands, 1 GPU" % (N, q, logB),
            "kernel": out["auto"]["kernel"], "ms": out["auto"]["ms"], "products_per_s": out["auto"]["products_per_s"],
            "vector_alu": out["vector ALU (packed MAC)"], "rows_equal_oracle": ok, "rows_checked": int(rows.numel())}


if __name__ == "__main__":
    # python tools/bench_configs.py [--only NAME[,NAME...]] [--no-pipeline]     NAME: encrypt509 encrypt701 encrypt821 verify15 verify18
    #                                                                                keygen polymul sampler add decrypt_pack encrypt_pack pipeline
    configs = [("encrypt509", lambda: encrypt_config("n509_q2048", 20)), ("encrypt701", lambda: encrypt_config("n701_q8192", 20)),
               ("encrypt821", lambda: encrypt_config("n821_q4096", 20)), ("verify15", lambda: verify_config("n821_q4096", 15)),
               ("verify18", lambda: verify_config("n821_q4096", 18)), ("keygen", lambda: keygen_config("n821_q4096", 18)),
               ("polymul", lambda: polymul_config(821, 4096, 18)), ("sampler", lambda: sampler_config("n821_q4096", 20)),
               ("add", lambda: add_config(821, 4096, 20)), ("decrypt_pack", lambda: decrypt_pack_config("n821_q4096", 20)),
               ("encrypt_pack", lambda: encrypt_pack_config("n821_q4096", 20)),
               ("pipeline", lambda: pipeline_dev_config("n821_q4096", 20)),
               ("pipeline20", lambda: pipeline_dev_config("n821_q4096", 20, 20)), ("pipeline17", lambda: pipeline_dev_config("n821_q4096", 20, 17))]      # (uses its own streams: kept out of the rocprofv3 passes)
    only = None
    if "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1].split(",")
    for name, fn in configs:
        if (only is not None and name not in only) or (only is None and name.startswith("pipeline") and (name != "pipeline" or "--no-pipeline" in sys.argv)):
            continue
        print(json.dumps(fn()), flush=True)
