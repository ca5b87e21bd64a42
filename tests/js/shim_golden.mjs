// Parity of the Node.js shim (ntru-circom_amd/js/index.mjs -> N-API addon -> C ABI -> HIP kernels) with the golden
// vectors captured from the reference.  Reads like the reference's own tests (test/reference.test.js): plain
// Arrays compared with deepStrictEqual.  Usage: node tests/js/shim_golden.mjs   (needs a GPU)
import { deepStrictEqual, strictEqual, notStrictEqual, throws, ok } from 'assert';
import { readFileSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath } from 'url';

import NTRU, * as lib from '../../ntru-circom_amd/js/index.mjs';

const here = dirname(fileURLToPath(import.meta.url));
const golden = name => JSON.parse(readFileSync(join(here, '..', 'golden', name), 'utf8'));

// replay tape for crypto.getRandomValues (index.js:481-482): the shim must consume draws exactly like the reference
let tape = [], pos = 0;
globalThis.crypto = { getRandomValues(a) { for (let i = 0; i < a.length; i++) a[i] = tape[pos++]; return a; } };

let checks = 0;
for (const profile of ['n17_q32', 'n167_q128', 'n509_q2048', 'n821_q4096', 'n701_q8192']) {
  const g = golden(`scheme_${profile}.json`);
  for (const key of g.keys) {
    const ntru = new NTRU({ ...g.options, f: key.f, fp: key.fp, fq: key.fq, g: key.g, h: key.h });
    deepStrictEqual(ntru.I, key.I);
    for (const c of key.cases) {
      tape = c.draws; pos = 0;
      const mBefore = c.m.slice();
      const enc = ntru.encryptBits(c.m);
      strictEqual(pos, c.draws.length);                 // exactly N-1 draws
      deepStrictEqual(c.m, mBefore);                    // inputs are never mutated
      deepStrictEqual(enc, c.encrypt);
      ok(Array.isArray(enc.value) && Array.isArray(enc.inputs.quotientE));
      deepStrictEqual(ntru.decryptBits(enc.value), c.decrypt);
      checks += 2;
    }
    for (const s of key.sums) {
      deepStrictEqual(lib.addPolynomials(s.e1, s.e2, g.options.q), s.eSum);
      deepStrictEqual(lib.addCiphertexts(s.e1, s.e2, g.options.q), s.eSum);
      deepStrictEqual(ntru.decryptBits(s.eSum), s.decrypt);
      checks += 3;
    }
    for (const d of key.degenerate) { deepStrictEqual(ntru.decryptBits(d.e), d.decrypt); checks++; }
    deepStrictEqual(ntru.verifyKeysInputs(), key.verifyKeysInputs);
    checks++;
    // generatePublicKeyH (index.js:72-79) on the device reproduces the captured h from fq and g
    const regen = new NTRU({ ...g.options, f: key.f, fq: key.fq, g: key.g });
    regen.generatePublicKeyH();
    deepStrictEqual(regen.h, key.h);
    // loadPrivateKeyF (index.js:30-49) on the device: fq and fp from f alone
    const inv = new NTRU({ ...g.options });
    inv.loadPrivateKeyF(key.f);
    deepStrictEqual(inv.fq, key.fq);
    deepStrictEqual(inv.fp, key.fp);
    checks += 3;
  }
  // test/reference.test.js:6-25 with a captured key; q = 1 mod 3 never round-trips in the reference (SURVEY.md 0.4)
  if (g.options.q % 3 === 2 && g.options.N >= 88) {
    const k = g.keys[0];
    const ntru = new NTRU({ ...g.options, f: k.f, fp: k.fp, h: k.h });
    globalThis.crypto = undefined;                      // fall back to node's CSPRNG for r
    strictEqual(ntru.decryptStr(ntru.encryptStr('Hello World')), 'Hello World');
    if (g.keys.length > 1) {
      const other = g.keys[g.keys.length - 1];
      const wrong = new NTRU({ ...g.options, f: other.f, fp: other.fp, h: k.h });
      notStrictEqual(wrong.decryptStr(ntru.encryptStr('Hello World')), 'Hello World');
    }
    // test/reference.test.js:6-14 verbatim: a key generated from scratch (all of it on the device) round-trips
    const fresh = new NTRU({ ...g.options });
    fresh.generatePrivateKeyF();
    fresh.generateNewPublicKeyGH();
    fresh.verifyKeysInputs();                             // throws if fq, fp or h were wrong
    strictEqual(fresh.decryptStr(fresh.encryptStr('Hello World')), 'Hello World');
    checks++;
    globalThis.crypto = { getRandomValues(a) { for (let i = 0; i < a.length; i++) a[i] = tape[pos++]; return a; } };
    checks += 2;
  }
}

// additive homomorphism literal of test/reference.test.js:50-52
{
  const g = golden('scheme_n167_q128.json');
  deepStrictEqual(g.keys[0].sums[0].decrypt.value, [1, 0, 2, 1, 1, 1, 0, 1]);
}

// pure functions
const pure = golden('pure_functions.json');
let nMul = 0, nDiv = 0;
for (const v of pure.multiply) {     // incl. the 2^20 products of test/circuits.test.js:60-72 (generic family)
  deepStrictEqual(lib.multiplyPolynomials(v.a, v.b, v.p), v.out); nMul++;
}
for (const v of pure.divide) {
  if (v.N === undefined) continue;
  deepStrictEqual(lib.dividePolynomials(v.a, v.b, v.p), v.out); nDiv++;
}
ok(nMul >= 25 && nDiv >= 60);
throws(() => lib.dividePolynomials([1, 2, 3], [0, 0], 7), /Cannot divide by zero polynomial\./);
for (const v of pure.sampler) {
  if (v.error) { throws(() => lib.generateCustomArray(v.len, v.n1, v.nm1), new RegExp(v.error.replace('.', '\\.'))); continue; }
  tape = v.draws; pos = 0;
  deepStrictEqual(lib.generateCustomArray(v.len, v.n1, v.nm1), v.out);
}
for (const v of pure.misc.trim) deepStrictEqual(lib.trimPolynomial(v.a), v.out);
for (const v of pure.misc.degree) strictEqual(lib.degree(v.a), v.out);
for (const v of pure.add) deepStrictEqual(lib.addPolynomials(v.a, v.b, v.p), v.out);
for (const v of pure.misc.NqNp) {
  const n = new NTRU({ N: v.N, q: v.q });
  strictEqual(n.calculateNq(), v.Nq); strictEqual(n.calculateNp(), v.Np);
}
deepStrictEqual(lib.stringToBits('Hello World'), pure.misc.stringToBits[0].out);

// error behaviour of the reference
{
  const n = new NTRU({ N: 17, q: 32, dr: 2, h: new Array(17).fill(1) });
  tape = new Array(64).fill(5); pos = 0;
  throws(() => n.encryptBits(new Array(18).fill(1)), RangeError);          // expandArray overflow, index.js:535
  throws(() => n.decryptBits([1, 2, 3]), TypeError);                       // this.f is null, index.js:112
  const full = { N: 17, q: 32, f: new Array(17).fill(1), fq: [1], fp: [1], g: new Array(17).fill(1), h: [1] };
  for (const [k, msg] of [['f', 'missing private key F'], ['fq', 'missing private key Fq'], ['fp', 'missing private key Fp'],
    ['g', 'missing private key G'], ['h', 'missing public key H']])
    throws(() => new NTRU({ ...full, [k]: null }).verifyKeysInputs(), new RegExp(msg));
  const e1 = [1].concat(new Array(16).fill(0));
  throws(() => new NTRU({ N: 17, q: 32, f: e1, fq: [1], fp: [1], g: e1, h: [5] }).verifyKeysInputs(), /invalid h/);
}

// additive batch API: typed arrays in, typed arrays out, equal to the per-item path
{
  const g = golden('scheme_n167_q128.json'); const key = g.keys[0]; const N = g.options.N;
  const ntru = new NTRU({ ...g.options, f: key.f, fp: key.fp, h: key.h });
  const B = key.cases.length;
  const r = new Uint8Array(B * N), m = new Uint8Array(B * N);
  key.cases.forEach((c, b) => { r.set(c.encrypt.inputs.r, b * N); m.set(c.encrypt.inputs.m, b * N); });
  const enc = ntru.encryptBatch(r, m, B);
  const dec = ntru.decryptBatch(enc.e, B);
  key.cases.forEach((c, b) => {
    deepStrictEqual(Array.from(enc.e.subarray(b * N, (b + 1) * N)).concat([0]), c.encrypt.inputs.remainderE);
    deepStrictEqual(Array.from(enc.quotientE.subarray(b * N, (b + 1) * N)).concat([0]), c.encrypt.inputs.quotientE);
    deepStrictEqual(Array.from(dec.value.subarray(b * N, (b + 1) * N)).concat([0]), c.decrypt.inputs.remainder2);
    deepStrictEqual(Array.from(dec.quotient1.subarray(b * N, (b + 1) * N)).concat([0]), c.decrypt.inputs.quotient1);
  });
}

// field packing (index.js:572-620) against the captured reference outputs
{
  const g = golden('pack_functions.json');
  for (const v of g.packOutput) {
    const got = lib.packOutput(v.maxVal, v.dataLen, v.data);
    deepStrictEqual({ ...got, expected: got.expected.map(x => x.toString(16)) },
      { maxInputBits: v.maxInputBits, maxOutputBits: v.maxOutputBits, outputSize: v.outputSize, arrLen: v.arrLen, expected: v.expected });
    strictEqual(typeof got.expected[0], 'bigint');
  }
  for (const v of g.unpackInput) {
    const got = lib.unpackInput(v.maxVal, v.packedBits, v.data.map(x => BigInt('0x' + x)));
    deepStrictEqual(got, { maxInputBits: v.maxInputBits, packedBits: v.packedBits, packedSize: v.packedSize,
      unpackedSize: v.unpackedSize, unpacked: v.unpacked });
  }
}

// on-device sampler: right weights, deterministic per (key, item), different per item; feeds the batch encrypt
{
  const g = golden('scheme_n167_q128.json'); const key = g.keys[0]; const N = g.options.N;
  const ntru = new NTRU({ ...g.options, f: key.f, fp: key.fp, h: key.h });
  const k = Uint32Array.from([1, 2, 3, 4, 5, 6, 7, 8]);
  const r1 = ntru.sampleR(k, 10, 3), r2 = ntru.sampleR(k, 10, 3), r3 = ntru.sampleR(k, 11, 2);
  deepStrictEqual(Array.from(r1), Array.from(r2));
  deepStrictEqual(Array.from(r1.subarray(N, 3 * N)), Array.from(r3));
  for (let b = 0; b < 3; b++) {
    const row = Array.from(r1.subarray(b * N, (b + 1) * N));
    strictEqual(row.filter(x => x === 1).length, g.options.dr);
    strictEqual(row.filter(x => x === 2).length, g.options.dr);
  }
  const m = new Uint8Array(3 * N).fill(1);
  const enc = ntru.encryptBatch(r1, m, 3);
  const dec = ntru.decryptBatch(enc.e, 3);
  deepStrictEqual(Array.from(dec.value), Array.from(m));       // q = 128 = 2 mod 3: round trip holds
}

// several devices in one process (NTRU.useDevices -> ntru_multi_*): the device is listed three times here (three engines,
// three host threads, three contiguous shards); results must equal the single-engine ones
{
  const g = golden('scheme_n167_q128.json'); const key = g.keys[0]; const N = g.options.N;
  const ntru = new NTRU({ ...g.options, f: key.f, fp: key.fp, h: key.h });
  const B = 1000;
  const r = ntru.sampleR(Uint32Array.from([9, 8, 7, 6, 5, 4, 3, 2]), 0, B);
  const m = new Uint8Array(B * N); for (let i = 0; i < m.length; i++) m[i] = (i * 7 + (i >> 5)) % 3;
  const enc1 = ntru.encryptBatch(r, m, B), dec1 = ntru.decryptBatch(enc1.e, B);
  strictEqual(NTRU.useDevices([0, 0, 0]), 3);
  const enc3 = ntru.encryptBatch(r, m, B), dec3 = ntru.decryptBatch(enc3.e, B);
  strictEqual(NTRU.useDevices([]), 0);
  for (const k of ['e', 'quotientE']) strictEqual(Buffer.compare(Buffer.from(enc1[k].buffer), Buffer.from(enc3[k].buffer)), 0);
  for (const k of ['value', 'quotient1', 'remainder1', 'quotient2'])
    strictEqual(Buffer.compare(Buffer.from(dec1[k].buffer), Buffer.from(dec3[k].buffer)), 0);
}

// Promise-returning batch calls: several in flight at once (serialised inside the addon), the event loop turns meanwhile,
// results equal the synchronous ones; a refused call rejects with the engine's message
const asyncChecks = (async () => {
  const g = golden('scheme_n167_q128.json'); const key = g.keys[0]; const N = g.options.N;
  const ntru = new NTRU({ ...g.options, f: key.f, fp: key.fp, h: key.h });
  const B = 3000;
  const r = ntru.sampleR(Uint32Array.from([3, 1, 4, 1, 5, 9, 2, 6]), 0, B);
  const m = new Uint8Array(B * N); for (let i = 0; i < m.length; i++) m[i] = (i * 5 + (i >> 7)) % 3;
  const encSync = ntru.encryptBatch(r, m, B), decSync = ntru.decryptBatch(encSync.e, B);
  let ticks = 0; const timer = setInterval(() => { ticks++; }, 0);
  const pending = [ntru.encryptBatchAsync(r, m, B), ntru.encryptBatchAsync(r, m, B, false), ntru.decryptBatchAsync(encSync.e, B)];
  const [enc, encValueOnly, dec] = await Promise.all(pending);
  clearInterval(timer);
  for (const k of ['e', 'quotientE']) strictEqual(Buffer.compare(Buffer.from(encSync[k].buffer), Buffer.from(enc[k].buffer)), 0);
  strictEqual(encValueOnly.quotientE, null);
  strictEqual(Buffer.compare(Buffer.from(encSync.e.buffer), Buffer.from(encValueOnly.e.buffer)), 0);
  for (const k of ['value', 'quotient1', 'remainder1', 'quotient2'])
    strictEqual(Buffer.compare(Buffer.from(decSync[k].buffer), Buffer.from(dec[k].buffer)), 0);
  const bad = new NTRU({ ...g.options, q: 12, f: key.f, fp: key.fp, h: key.h });
  let msg = null;
  try { await bad.encryptBatchAsync(r, m, 4); } catch (err) { msg = err.message; }
  ok(msg && /ntru engine error/.test(msg), `rejection message: ${msg}`);
  return ticks;
})();
asyncChecks.then(ticks => console.log(`shim_golden: async batch calls OK (${ticks} event-loop ticks while in flight)`),
  err => { console.error(err); process.exit(1); });

console.log(`shim_golden: ${checks} scheme checks, ${nMul} multiply and ${nDiv} divide vectors OK`);
