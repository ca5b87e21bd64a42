// Host-inclusive throughput of the Node.js boundary: JS typed arrays -> N-API addon -> C ABI host-pointer entry points
// (ntru_host.hip: two streams, pinned arenas, chunked H2D / kernel / D2H) -> back.  This is what a Node caller sees;
// the device-resident headline is bench.py's.
//
//   node tools/bench_host.mjs [--log2 18] [--iters 5] [--profile n821_q4096]
//
// Reports, for encryptBatch + decryptBatch with every witness array (14N bytes per round trip across PCIe: 2N in + 4N
// out for encrypt, 2N in + 6N out for decrypt):
//   pinned    TypedArrays from NTRU.allocUint8 / allocUint16 (page-locked, DMA'd in place)
//   pageable  ordinary TypedArrays (staged through the engine's pinned arenas by a multi-threaded memcpy)
// and the latency of single encryptBits / decryptBits calls (the reference's own synchronous API).
import { readFileSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath } from 'url';
import { cpus } from 'os';

import NTRU from '../ntru-circom_amd/js/index.mjs';

const here = dirname(fileURLToPath(import.meta.url));
const arg = (name, dflt) => { const i = process.argv.indexOf(name); return i > 0 ? process.argv[i + 1] : dflt; };
const log2 = Number(arg('--log2', 18)), iters = Number(arg('--iters', 5)), profile = arg('--profile', 'n821_q4096');
const devices = arg('--devices', '');   // e.g. 0,1,2,3: shard the batches over several devices (ntru_multi_*)
const PCIE_GBS = 63.0;       // PCIe Gen5 x16, one direction (the figure DESIGN.md section 5 uses)

const g = JSON.parse(readFileSync(join(here, '..', 'tests', 'golden', `scheme_${profile}.json`), 'utf8'));
const key = g.keys[0];
const ntru = new NTRU({ ...g.options, f: key.f, fp: key.fp, fq: key.fq, g: key.g, h: key.h });
const { N, dr } = ntru;
const B = 1 << log2;
const now = () => Number(process.hrtime.bigint()) / 1e6;       // ms

// inputs: r from the engine's sampler (valid weights), m random bits
const rSrc = ntru.sampleR(Uint32Array.from([1, 2, 3, 4, 5, 6, 7, 8]), 0, B);
const mSrc = new Uint8Array(B * N);
for (let i = 0; i < mSrc.length; i++) mSrc[i] = (i * 2654435761 >>> 7) & 1;

function roundTrips(alloc8, alloc16) {
  const r = alloc8(B * N), m = alloc8(B * N);
  r.set(rSrc); m.set(mSrc);
  const encOut = { e: alloc16(B * N), quotientE: alloc16(B * N) };
  const decOut = { value: alloc8(B * N), quotient1: alloc16(B * N), remainder1: alloc16(B * N), quotient2: alloc8(B * N) };
  const run = () => { ntru.encryptBatch(r, m, B, true, encOut); ntru.decryptBatch(encOut.e, B, true, decOut); };
  run();                                                         // warm-up: arenas grow once
  const t = [], te = [], td = [];
  for (let i = 0; i < iters; i++) {
    const t0 = now(); ntru.encryptBatch(r, m, B, true, encOut);
    const t1 = now(); ntru.decryptBatch(encOut.e, B, true, decOut);
    const t2 = now();
    te.push(t1 - t0); td.push(t2 - t1); t.push(t2 - t0);
  }
  const med = a => a.slice().sort((x, y) => x - y)[a.length >> 1];
  const ms = med(t);
  return { ms_per_batch: ms, encrypt_ms: med(te), decrypt_ms: med(td), round_trips_per_s: B / (ms * 1e-3),
    pcie_bytes_per_round_trip: 14 * N, achieved_GBs: 14 * N * B / (ms * 1e-3) / 1e9,
    frac_of_14N_over_pcie: (14 * N * B / (ms * 1e-3) / 1e9) / PCIE_GBS,
    checksum: decOut.value.reduce((s, x) => (s + x) >>> 0, 0) + encOut.e[encOut.e.length - 1] };
}

if (devices) NTRU.useDevices(devices.split(',').map(Number));
const pinned = roundTrips(n => NTRU.allocUint8(n), n => NTRU.allocUint16(n));
const pageable = roundTrips(n => new Uint8Array(n), n => new Uint16Array(n));
if (pinned.checksum !== pageable.checksum) throw new Error('pinned and pageable paths disagree');

// Device-resident pipeline (ntru.pipeline): only m goes up, r is drawn on the GPU, e stays there, only value (or its packed form)
// comes down.  PCIe carries 2N bytes per round trip (N up, N down -- the two directions are separate links: full duplex).
function pipelineRate(opts, bytesUp, bytesDown) {
  const mPin = NTRU.allocUint8(B * N); mPin.set(mSrc);
  const out = {};
  const decrypt = opts.decrypt !== false;
  if (!opts.pack && decrypt) out.value = NTRU.allocUint8(B * N);
  if (!opts.pack && !decrypt) out.e = NTRU.allocUint16(B * N);
  const chacha = Uint32Array.from([1, 2, 3, 4, 5, 6, 7, 8]);
  const run = () => ntru.pipeline({ m: mPin, B, sampleR: { key: chacha, firstItem: 0 }, decrypt: true, out, ...opts });
  let res = run();
  if (opts.pack) out.packed = res.packed;
  const t = [];
  for (let i = 0; i < iters; i++) { const t0 = now(); res = run(); t.push(now() - t0); }
  const ms = t.slice().sort((x, y) => x - y)[t.length >> 1];
  const up = bytesUp * B / (ms * 1e-3) / 1e9, down = bytesDown * B / (ms * 1e-3) / 1e9;
  return { ms_per_batch: ms, round_trips_per_s: B / (ms * 1e-3), pcie_bytes_up_per_round_trip: bytesUp, pcie_bytes_down_per_round_trip: bytesDown,
    up_GBs: up, down_GBs: down, frac_of_pcie_one_direction: Math.max(up, down) / PCIE_GBS,
    checksum: opts.pack ? Number(res.packed[res.packed.length - 4] & 0xffffn) : (decrypt ? res.value : res.e).reduce((s, x) => (s + x) >>> 0, 0) };
}
const pipeValue = pipelineRate({}, N, N);
const pipePacked = pipelineRate({ pack: true }, N, 32 * Math.max(3, Math.ceil(N / Math.floor(252 / 2))));
// encrypt only: the ciphertext comes down plain (2N bytes) or as packOutput(q - 1, N, e) from the fused encrypt + pack kernel
const qBits = Math.floor(Math.log2(ntru.q - 1)) + 1;
const encPlain = pipelineRate({ decrypt: false }, N, 2 * N);
const encPacked = pipelineRate({ decrypt: false, pack: true }, N, 32 * Math.max(3, Math.ceil(N / Math.floor(252 / qBits))));

// single calls through the reference's own API (plain Arrays in, witness objects out)
const lat = (fn, n) => { const t = []; for (let i = 0; i < n; i++) { const t0 = now(); fn(); t.push(now() - t0); } t.sort((a, b) => a - b); return { median_ms: t[n >> 1], p90_ms: t[Math.floor(n * 0.9)], min_ms: t[0] }; };
const mBits = Array.from(mSrc.subarray(0, N));
let enc = ntru.encryptBits(mBits);
const encLat = lat(() => { enc = ntru.encryptBits(mBits); }, 200);
const decLat = lat(() => { ntru.decryptBits(enc.value); }, 200);
const verLat = lat(() => { ntru.verifyKeysInputs(); }, 50);

console.log(JSON.stringify({
  what: 'host-inclusive Node.js path: encryptBatch + decryptBatch, full witness, host TypedArrays in and out',
  N, q: ntru.q, batch: B, iters, devices: devices || 'single', host_cpus: cpus().length, pcie_roof_GBs_one_direction: PCIE_GBS,
  bound_round_trips_per_s_at_14N_over_pcie: PCIE_GBS * 1e9 / (14 * N),
  pinned, pageable,
  pipeline_value_only: { what: 'ntru.pipeline({sampleR, decrypt}): sampler -> encryptBits -> decryptBits on the GPU, m up, value down (page-locked arrays)',
    bound_round_trips_per_s_at_N_each_way: PCIE_GBS * 1e9 / N, ...pipeValue },
  pipeline_packed_value: { what: 'the same with pack: true: packOutput(2, N, value) comes down instead of value (decrypt + pack fused: k_decrypt_mp)', ...pipePacked },
  pipeline_encrypt_plain: { what: 'ntru.pipeline({sampleR, decrypt: false}): sampler -> encryptBits, m up, e (2N bytes) down', ...encPlain },
  pipeline_encrypt_packed: { what: 'the same with pack: true: packOutput(q - 1, N, e) comes down instead of e (encrypt + pack fused: k_encrypt_wp)', ...encPacked },
  single_call_latency: { encryptBits: encLat, decryptBits: decLat, verifyKeysInputs: verLat,
    note: 'includes sampling r in JS (N-1 CSPRNG draws), Array <-> TypedArray conversion, one H2D, one launch, one D2H' },
}));
