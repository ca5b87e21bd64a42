// Device-resident pipeline through the Node shim: ntru.pipeline({sampleR, decrypt, pack}) and the same stages composed by hand on
// device-buffer handles.  Writes every array to <out dir> as raw little-endian files; tests/test_js_shim.py replays the ChaCha20 draw
// stream, encryptBits, decryptBits and packOutput on the CPU oracle and compares.
//   node tests/js/shim_pipeline.mjs <profile> <B> <out dir>
import { readFileSync, writeFileSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath } from 'url';

import NTRU from '../../ntru-circom_amd/js/index.mjs';

const here = dirname(fileURLToPath(import.meta.url));
const [profile, Bs, outDir] = process.argv.slice(2);
const B = Number(Bs);
const g = JSON.parse(readFileSync(join(here, '..', 'golden', `scheme_${profile}.json`), 'utf8'));
const key = g.keys[0];
const ntru = new NTRU({ ...g.options, f: key.f, fp: key.fp, fq: key.fq, g: key.g, h: key.h });
const { N, p, q } = ntru;
const chacha = Uint32Array.from([11, 22, 33, 44, 55, 66, 77, 88]);
// NTRU_SAMPLER_ROUNDS=12|8: the reduced-round draw streams (NTRU.samplerRounds); default ChaCha20
const rounds = NTRU.samplerRounds(Number(process.env.NTRU_SAMPLER_ROUNDS || 0));
const firstItem = 4294967296 + 5;                      // item indices above 2^32: both nonce words matter
const m = new Uint8Array(B * N);
for (let i = 0; i < m.length; i++) m[i] = (Math.imul(i, 2654435761) >>> 9) & 1;
const dump = (name, a) => writeFileSync(join(outDir, name + '.bin'), Buffer.from(a.buffer, a.byteOffset, a.byteLength));

// 1. everything: sampled r back (to replay), ciphertext, value, packed value
const all = ntru.pipeline({ m, B, sampleR: { key: chacha, firstItem }, decrypt: true, pack: true, want: { r: true, e: true, value: true } });
dump('m', m); dump('r', all.r); dump('e', all.e); dump('value', all.value); dump('packed_value', all.packed);
// 2. the throughput shape: only m up, only value down (page-locked arrays)
const mPin = NTRU.allocUint8(B * N); mPin.set(m);
const vPin = NTRU.allocUint8(B * N);
const lean = ntru.pipeline({ m: mPin, B, sampleR: { key: chacha, firstItem }, decrypt: true, out: { value: vPin } });
if (lean.value !== vPin || Object.keys(lean).length !== 1) throw new Error('lean pipeline: unexpected outputs ' + Object.keys(lean));
for (let i = 0; i < vPin.length; i++) if (vPin[i] !== all.value[i]) throw new Error('lean pipeline: value differs at ' + i);
// 3. encrypt only with a given r, packed ciphertext
const enc = ntru.pipeline({ m, B, r: all.r, pack: true, want: { e: true } });
for (let i = 0; i < enc.e.length; i++) if (enc.e[i] !== all.e[i]) throw new Error('pipeline with given r: e differs at ' + i);
dump('packed_e', enc.packed);
// 4. the same stages by hand on device-buffer handles
const hDev = NTRU.devAlloc(2 * N), fDev = NTRU.devAlloc(N), fpDev = NTRU.devAlloc(N);
const pad = (a, T) => { const t = new T(N); t.set(a); return t; };
NTRU.devUpload(hDev, pad(ntru.h, Uint16Array)); NTRU.devUpload(fDev, pad(ntru.f, Int8Array)); NTRU.devUpload(fpDev, pad(ntru.fp, Uint8Array));
const rDev = NTRU.devAlloc(B * N), mDev = NTRU.devAlloc(B * N), eDev = NTRU.devAlloc(2 * B * N), vDev = NTRU.devAlloc(B * N);
const os = all.outputSize, pkDev = NTRU.devAlloc(B * os * 32);
NTRU.devUpload(mDev, m);
ntru.sampleRDev(chacha, firstItem, B, rDev);
ntru.encryptBatchDev(hDev, rDev, mDev, B, eDev);
ntru.decryptBatchDev(fDev, fpDev, eDev, B, vDev);
NTRU.packBatchDev(p - 1, N, vDev, B, pkDev, true);
const v2 = NTRU.devDownload(new Uint8Array(B * N), vDev), pk2 = NTRU.devDownload(new BigUint64Array(B * os * 4), pkDev);
for (let i = 0; i < v2.length; i++) if (v2[i] !== all.value[i]) throw new Error('device handles: value differs at ' + i);
for (let i = 0; i < pk2.length; i++) if (pk2[i] !== all.packed[i]) throw new Error('device handles: packed differs at ' + i);
let threw = false;
try { ntru.encryptBatchDev(hDev, rDev, mDev, B + 1, eDev); } catch (e) { threw = true; }       // a handle too small for the launch
if (!threw) throw new Error('an undersized device buffer must be refused before the launch');
for (const d of [hDev, fDev, fpDev, rDev, mDev, eDev, vDev, pkDev]) NTRU.devFree(d);
threw = false;
try { NTRU.devUpload(mDev, m); } catch (e) { threw = true; }
if (!threw) throw new Error('a freed handle must be refused');
(async () => {
// 5. the Promise-returning twin (libuv worker thread): same results, and the event loop turns while it runs
let ticks = 0;
const timer = setInterval(() => { ticks++; }, 0);
const viaPromise = await ntru.pipelineAsync({ m, B, sampleR: { key: chacha, firstItem }, decrypt: true, pack: true, want: { value: true } });
clearInterval(timer);
for (let i = 0; i < viaPromise.value.length; i++) if (viaPromise.value[i] !== all.value[i]) throw new Error('pipelineAsync: value differs at ' + i);
for (let i = 0; i < viaPromise.packed.length; i++) if (viaPromise.packed[i] !== all.packed[i]) throw new Error('pipelineAsync: packed differs at ' + i);
writeFileSync(join(outDir, 'meta.json'), JSON.stringify({ N, q, p, dr: ntru.dr, B, samplerRounds: rounds, outputSizeValue: os, outputSizeE: enc.outputSize,
  key: Array.from(chacha), firstItem }));
console.log('shim_pipeline: OK', profile, B);
})().catch(e => { console.error(e); process.exit(1); });
