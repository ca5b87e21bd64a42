#!/usr/bin/env python3
"""Power limit or latency?  (missing #4.)

For each load -- one idle wave, every SIMD issuing int8 matrix instructions back to back, a sustained loop of the encrypt
kernel, of the decrypt kernel, of verify_keys -- this records

  * the average SHADER clock over the run from two in-stream probes: d(s_memtime) / d(s_memrealtime) x 100 MHz
    (bench_micro/clock_probe.hip; s_memtime is the counter the phase stamps are in, s_memrealtime the constant reference);
  * what the driver reports meanwhile, sampled by a SECOND process that never touches HIP: sclk / mclk / socket power from
    the amdgpu sysfs files of this device (and one `rocm-smi` / `amd-smi` snapshot per load as a cross-check).

  python3 tools/clock_power.py [--seconds 5] > gpurun_out/r03_clock_power.txt
"""
import argparse
import ctypes
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SAMPLER = r'''
import glob, json, os, sys, time
dev_dir, out_path, stop_path = sys.argv[1], sys.argv[2], sys.argv[3]
def rd(p):
    try:
        with open(p) as fh: return fh.read().strip()
    except OSError: return None
def cur_clk(txt):
    if not txt: return None
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            return line.split(":")[1].strip().rstrip("*").strip()
    return None
hw = sorted(glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*")))
hw = hw[0] if hw else None
with open(out_path, "w") as out:
    while not os.path.exists(stop_path):
        rec = {"t": time.time(), "sclk": cur_clk(rd(os.path.join(dev_dir, "pp_dpm_sclk"))),
               "mclk": cur_clk(rd(os.path.join(dev_dir, "pp_dpm_mclk"))), "busy": rd(os.path.join(dev_dir, "gpu_busy_percent"))}
        if hw:
            for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "power1_cap"):
                v = rd(os.path.join(hw, name))
                if v is not None: rec[name] = v
        out.write(json.dumps(rec) + "\n"); out.flush()
        time.sleep(0.05)
'''


def sysfs_dir_of(pci_bus_id):
    """/sys/class/drm/cardK/device of the HIP device (matched by PCI address), or None."""
    want = pci_bus_id.lower()
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            real = os.path.realpath(d)
        except OSError:
            continue
        if real.lower().endswith(want) or os.path.basename(real).lower() == want:
            return d
    return None


def smi_snapshot():
    outs = {}
    for cmd in (["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--json"],
                ["amd-smi", "metric", "--clock", "--power", "--json"]):
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=20)
            outs[cmd[0]] = {"rc": p.returncode, "out": p.stdout.decode("utf-8", "replace")[-3000:],
                            "err": p.stderr.decode("utf-8", "replace")[-300:]}
        except (OSError, subprocess.TimeoutExpired) as exc:
            outs[cmd[0]] = {"error": repr(exc)}
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--no-smi", action="store_true")
    ap.add_argument("--kernel-path", type=int, default=0, help="ntru_engine_set_kernel_path for the encrypt / decrypt loops")
    ap.add_argument("--row-pitch", type=int, default=0, help="row pitch in elements (0 = dense)")
    ap.add_argument("--loads", default="all", help="comma list of: idle,mfma,encrypt,decrypt,decrypt_value,verify (default all), sampler; with "
                                                   "NTRU_ENGINE_LIB pointing at another build of the library this prices its energy")
    args = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as ge
    import bench
    pkg = ge.load_package()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    props = torch.cuda.get_device_properties(0)
    probe = ctypes.CDLL(os.path.join(ROOT, "bench_micro", "libclock_probe.so"))
    probe.clock_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    probe.clock_probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong,
                                       ctypes.c_void_p, ctypes.c_void_p]
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)

    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    pci = None
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
        pci = buf.value.decode()
    dev_dir = sysfs_dir_of(pci) if pci else None
    print(json.dumps({"device": props.name, "cus": props.multi_processor_count, "pci_bus_id": pci, "sysfs": dev_dir,
                      "torch_clock_rate_khz": getattr(props, "clock_rate", None)}))

    o, h_np, f_np, fp_np = bench.load_key("n821_q4096")
    N, q, p, d = o["N"], o["q"], o["p"], o["dr"]
    B = 1 << args.batch_log2
    r, m = bench.make_inputs(torch, dev, B, N, d, 20240)
    h = torch.from_numpy(h_np.view(np.int16)).to(dev); f = torch.from_numpy(f_np).to(dev); fp = torch.from_numpy(fp_np).to(dev)
    b16 = lambda: torch.empty((B, N), dtype=torch.int16, device=dev)
    b8 = lambda: torch.empty((B, N), dtype=torch.uint8, device=dev)
    e, quotE, value, quot1, rem1, quot2 = b16(), b16(), b8(), b16(), b16(), b8()
    eng = pkg.Engine(0)
    eng.set_stream(stream.cuda_stream)
    enc = lambda: eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), quotE.data_ptr())
    dec = lambda: eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, value.data_ptr(), quot1.data_ptr(),
                                        rem1.data_ptr(), quot2.data_ptr())
    dec_v = lambda: eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, value.data_ptr())
    Bk = 1 << 18
    gk = torch.Generator(device=dev); gk.manual_seed(5)
    tern = lambda: (torch.randint(0, 3, (Bk, N), device=dev, generator=gk) - 1).to(torch.int8)
    kf, kg = tern(), tern()
    kfq = torch.randint(0, q, (Bk, N), device=dev, generator=gk).to(torch.int16)
    kh = torch.randint(0, q, (Bk, N), device=dev, generator=gk).to(torch.int16)
    kfp = torch.randint(0, p, (Bk, N), device=dev, generator=gk).to(torch.uint8)
    kouts = [torch.empty((Bk, N), dtype=t, device=dev) for t in (torch.int16, torch.int16, torch.uint8, torch.uint8, torch.int16, torch.int16)]
    kflags = torch.empty(Bk, dtype=torch.uint8, device=dev)
    vk = lambda: eng.verify_keys_batch_dev(N, q, p, kf.data_ptr(), kg.data_ptr(), kfq.data_ptr(), kfp.data_ptr(), kh.data_ptr(), Bk,
                                           *[t.data_ptr() for t in kouts], kflags.data_ptr())
    eng.set_kernel_path(args.kernel_path)
    enc(); dec(); torch.cuda.synchronize()
    eng.set_kernel_path(0); vk(); torch.cuda.synchronize(); eng.set_kernel_path(args.kernel_path)

    clk = torch.zeros(4, dtype=torch.int64, device=dev)
    spin_out = torch.zeros(3 * 4096, dtype=torch.int64, device=dev)
    spin_in = torch.arange(64, dtype=torch.int32, device=dev)
    ticks = int(args.seconds * 100e6)

    def sampled(name, body):
        stop = "/tmp/clock_power.stop"
        outp = "/tmp/clock_power_%s.jsonl" % name
        for pth in (stop, outp):
            if os.path.exists(pth): os.remove(pth)
        proc = None
        if dev_dir:
            proc = subprocess.Popen([sys.executable, "-c", SAMPLER, dev_dir, outp, stop])
            time.sleep(0.3)
        t0 = time.time()
        res = body()
        t1 = time.time()
        snap = None if args.no_smi else None
        if proc:
            open(stop, "w").close(); proc.wait(timeout=10)
            rows = [json.loads(l) for l in open(outp)]
            rows = [x for x in rows if t0 + 0.25 * (t1 - t0) <= x["t"] <= t1]      # skip the ramp
            def stat(key, conv=float):
                vals = []
                for x in rows:
                    v = x.get(key)
                    if v is None: continue
                    try: vals.append(conv(str(v).lower().replace("mhz", "").strip()))
                    except ValueError: pass
                return {"n": len(vals), "mean": float(np.mean(vals)), "min": float(np.min(vals)), "max": float(np.max(vals))} if vals else None
            res["driver"] = {"sclk_MHz": stat("sclk"), "mclk_MHz": stat("mclk"),
                             "power_W": (lambda s: s and {k: (v / 1e6 if k != "n" else v) for k, v in s.items()})(stat("power1_average") or stat("power1_input")),
                             "freq1_MHz": (lambda s: s and {k: (v / 1e6 if k != "n" else v) for k, v in s.items()})(stat("freq1_input")),
                             "power_cap_W": (lambda s: s and s["mean"] / 1e6)(stat("power1_cap")), "samples": len(rows)}
        res["load"] = name
        print(json.dumps(res)); sys.stdout.flush()

    def spin(mode, blocks, threads):
        def body():
            probe.clock_probe_spin(sp, mode, blocks, threads, ticks, ctypes.c_void_p(spin_out.data_ptr()), ctypes.c_void_p(spin_in.data_ptr()))
            torch.cuda.synchronize()
            a = spin_out[:3 * blocks].cpu().numpy().reshape(blocks, 3)
            ghz = a[:, 0] / a[:, 1] * 0.1
            out = {"shader_clock_GHz_from_memtime": {"mean": float(ghz.mean()), "min": float(ghz.min()), "max": float(ghz.max())},
                   "real_seconds": float(a[:, 1].mean() / 100e6)}
            if mode >= 3:
                ins = a[:, 2].astype(np.float64) * 256 * (threads // 64)     # wave-instructions per workgroup
                out["wave_instr_per_s_G"] = float((ins / (a[:, 1] / 100e6)).sum() / 1e9)
                out["clocks_per_instr_per_simd"] = float((a[:, 0] / (a[:, 2] * 256.0 * max(1, threads // 256))).mean())
            elif mode >= 1:
                mf = a[:, 2].astype(np.float64) * 256 * (threads // 64)      # matrix instructions per workgroup
                secs = a[:, 1] / 100e6
                out["mfma_per_s_G"] = float((mf / secs).sum() / 1e9)
                out["clocks_per_mfma_per_simd"] = float((a[:, 0] / (a[:, 2] * 256.0 * max(1, threads // 256))).mean())
            return out
        return body

    def loop(fn, per_launch_items):
        def body():
            n = 0
            torch.cuda.synchronize()
            probe.clock_probe_read(sp, ctypes.c_void_p(clk.data_ptr()))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
            t0 = time.time()
            while time.time() - t0 < args.seconds:
                for _ in range(50):
                    fn(); n += 1
                torch.cuda.synchronize()
            ev1.record(stream)
            probe.clock_probe_read(sp, ctypes.c_void_p(clk[2:].data_ptr()))
            torch.cuda.synchronize()
            c = clk.cpu().numpy()
            dt, dr = int(c[2] - c[0]), int(c[3] - c[1])
            ms = ev0.elapsed_time(ev1)
            return {"launches": n, "kernel": eng.last_kernel(), "ms_per_launch": ms / n, "items_per_launch": per_launch_items,
                    "shader_clock_GHz_from_memtime": dt / dr * 0.1, "real_seconds": dr / 100e6, "memtime_ticks": dt, "memrealtime_ticks": dr}
        return body

    cus = props.multi_processor_count
    want = lambda k: (args.loads == "all" and k != "kinds") or k in args.loads.split(",")
    if os.environ.get("NTRU_ENGINE_LIB"):
        print(json.dumps({"engine_lib": os.environ["NTRU_ENGINE_LIB"]}))
    if want("idle"): sampled("idle_one_wave", spin(0, 1, 64))
    if want("mfma"):
        sampled("mfma_i8_all_simds_1_wave_each", spin(1, cus, 256))
        sampled("mfma_i8_all_simds_2_waves_each", spin(1, cus, 512))
        sampled("mfma_fp4_k64_all_simds_1_wave_each", spin(2, cus, 256))
    if want("kinds"):
        # energy per instruction class: 256 instructions of one kind per loop iteration and wave, 2 waves per SIMD
        for mode, name in ((9, "s_nop"), (3, "v_perm_b32"), (4, "v_add_u32"), (5, "ds_read_b128"), (6, "ds_read_u8_scattered"),
                           (7, "ds_write_b16"), (8, "ds_write_b64")):
            sampled("kind_" + name, spin(mode, cus, 512))
    if want("encrypt"): sampled("encrypt_loop", loop(enc, B))
    if want("decrypt"): sampled("decrypt_loop_full_witness", loop(dec, B))
    if want("decrypt_value"): sampled("decrypt_loop_value_only", loop(dec_v, B))
    if want("verify"): sampled("verify_keys_loop", loop(vk, Bk))
    if "sampler" in args.loads.split(","):
        skey = np.arange(8, dtype=np.uint32) + 1
        sampled("sampler_loop", loop(lambda: eng.sample_ternary_dev(N, 273 if N == 821 else N // 3, 273 if N == 821 else N // 3, 2, skey, 0, B, r.data_ptr()), B))
    if want("idle"): sampled("idle_one_wave_again", spin(0, 1, 64))
    if not args.no_smi:
        print(json.dumps({"smi_snapshot_idle": smi_snapshot()}))


if __name__ == "__main__":
    main()
