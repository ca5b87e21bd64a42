// Golden-vector generator for the NTRU hot path.
//
// Runs the UNMODIFIED reference (numtel/ntru-circom index.js) under Node in the
// build container and dumps inputs + outputs of the hot-path functions as JSON.
// Only the resulting *.json data is committed / shipped to the GPU box; the
// reference itself never travels.
//
//   node tests/golden/gen_golden.mjs [/root/reference] [outdir]
//
// Determinism: the reference draws randomness from the global
// crypto.getRandomValues (index.js:481-482), which Node 12 lacks.  We install a
// seeded xorshift32 shim before importing it, so every run reproduces the same
// keys, r vectors and ciphertexts bit for bit.  Every u32 the shim hands out can
// be recorded so the sampler (index.js:461-488) can be replayed by our own code.

import { writeFileSync, mkdirSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath, pathToFileURL } from 'url';

const here = dirname(fileURLToPath(import.meta.url));
const refDir = process.argv[2] || '/root/reference';
const outDir = process.argv[3] || here;

// ---- seeded WebCrypto shim -------------------------------------------------
let state = 1;
let tape = null; // when an array, every u32 drawn is appended
function reseed(s) { state = (s >>> 0) || 1; }
function nextU32() {
  let x = state;
  x ^= x << 13; x >>>= 0;
  x ^= x >>> 17;
  x ^= x << 5; x >>>= 0;
  state = x;
  return x;
}
globalThis.crypto = {
  getRandomValues(arr) {
    for (let i = 0; i < arr.length; i++) {
      const v = nextU32();
      arr[i] = v;
      if (tape) tape.push(arr[i]);
    }
    return arr;
  },
};

function record(fn) {
  tape = [];
  const out = fn();
  const t = tape;
  tape = null;
  return { out, draws: t };
}

function randInts(n, mod) {
  const a = new Array(n);
  for (let i = 0; i < n; i++) a[i] = nextU32() % mod;
  return a;
}

function dump(name, obj) {
  mkdirSync(outDir, { recursive: true });
  const p = join(outDir, name);
  writeFileSync(p, JSON.stringify(obj));
  console.log('wrote', p);
}

function main(ref) {
  const NTRU = ref.default;

  // ---- pure-function vectors ------------------------------------------------
  const pure = { multiply: [], divide: [], add: [], sampler: [], misc: {} };

  // tuples of test/circuits.test.js:60-66 (expected values come from the JS at run time there)
  const mulTuples = [
    [[1, 4], [0, 3], 7],
    [[1, 2, 3], [4, 3, 2], 7],
    [[1, 2, 3, 4], [5, 4, 3, 2], 11],
    [[1, 2, 3, 4, 5], [6, 5, 4, 3, 2], 13],
    [[1, 2, 3, 4, 5, 0], [7, 6, 5, 4, 3, 2], 13],
  ];
  for (const [a, b, p] of mulTuples) {
    pure.multiply.push({ a, b, p: Math.pow(2, 20), out: ref.multiplyPolynomials(a, b, Math.pow(2, 20)) });
    pure.multiply.push({ a, b, p, out: ref.multiplyPolynomials(a, b, p) });
  }
  // worked example of circuits/ntru.circom:36-71
  pure.multiply.push({ a: [1, 2, 3, 4], b: [6, 5, 4, 3], p: 1 << 20, out: ref.multiplyPolynomials([1, 2, 3, 4], [6, 5, 4, 3], 1 << 20) });
  // edge cases: empty operands, zeros, negative coefficients, trailing zeros
  for (const [a, b, p] of [
    [[], [1, 2], 7], [[1, 2], [], 7], [[0], [0], 5], [[0, 0, 0], [1, 2], 5],
    [[-1, 0, 1], [1, -1], 3], [[-1, 0, 1, 0, 0], [5, 127, 0, 64], 128], [[3], [4], 5],
  ]) pure.multiply.push({ a, b, p, out: ref.multiplyPolynomials(a, b, p) });
  // random mid-size and worst-case magnitudes (exactness of the float FFT, SURVEY §0.2)
  reseed(0xC0FFEE);
  for (const [n, q] of [[17, 32], [167, 128], [509, 2048], [821, 4096], [701, 8192]]) {
    const a = randInts(n, q), b = randInts(n, q);
    pure.multiply.push({ a, b, p: q, out: ref.multiplyPolynomials(a, b, q) });
    const am = new Array(n).fill(q - 1), bm = new Array(n).fill(q - 1);
    pure.multiply.push({ a: am, b: bm, p: q, out: ref.multiplyPolynomials(am, bm, q) });
    const a3 = randInts(n, 3), b3 = randInts(n, 3);
    pure.multiply.push({ a: a3, b: b3, p: 3, out: ref.multiplyPolynomials(a3, b3, 3) });
    const as = randInts(n, q).map(x => 3 * x);
    const t = randInts(n, 3).map(x => (x === 2 ? q - 1 : x));
    pure.multiply.push({ a: as, b: t, p: q, out: ref.multiplyPolynomials(as, t, q) });
  }

  // tuples of test/circuits.test.js:165-170 (generic divisors)
  const divTuples = [
    [[1, 2], [2, 3], 8],
    [[1, 2], [2, 3], 3],
    [[81, 2, 96], [48, 2, 31], 128],
    [[81, 2, 96], [48, 2, 31], 16],
  ];
  for (const [a, b, p] of divTuples) {
    let out = null, error = null;
    try { out = ref.dividePolynomials(a, b, p); } catch (e) { error = e.message; }
    pure.divide.push({ a, b, p, out, error });
  }
  {
    let error = null;
    try { ref.dividePolynomials([1, 2, 3], [0, 0], 7); } catch (e) { error = e.message; }
    pure.divide.push({ a: [1, 2, 3], b: [0, 0], p: 7, out: null, error });
  }
  // division by I = 1 - x^N (the only divisor on the hot path): dense, sparse and short dividends
  reseed(0xD1CE);
  for (const [n, q] of [[5, 8], [17, 32], [17, 3], [167, 128], [167, 3], [509, 2048], [821, 4096], [821, 3], [701, 8192]]) {
    const I = new Array(n + 1).fill(0); I[0] = 1; I[n] = -1;
    const lens = [2 * n - 1, 2 * n - 2, n + 1, n, n - 1, 1, 0];
    for (const len of lens) {
      const a = randInts(len, q);
      pure.divide.push({ a, b: I, p: q, N: n, out: ref.dividePolynomials(a, I, q) });
    }
    // sparse: zero high part, and a[N+k] hitting exactly 0 at the top
    const sp = new Array(2 * n - 1).fill(0); sp[0] = 1; sp[n] = q - 1; sp[2 * n - 3] = 1;
    pure.divide.push({ a: sp, b: I, p: q, N: n, out: ref.dividePolynomials(sp, I, q) });
    const zeros = new Array(n + 3).fill(0);
    pure.divide.push({ a: zeros, b: I, p: q, N: n, out: ref.dividePolynomials(zeros, I, q) });
  }

  for (const [a, b, p] of [
    [[1, 2, 1, 0, 1], [0, 1, 1, 1, 0, 1, 0, 1], 3], [[5, 6], [], 4], [[], [], 7], [[3, 3], [1, 1], 4], [[-1, -2], [0, 0, 9], 5],
  ]) pure.add.push({ a, b, p, out: ref.addPolynomials(a, b, p) });

  // sampler (index.js:461-488): record draws so the order "i descending, j = u32 % (i+1)" is pinned
  reseed(0x5EED);
  for (const [len, n1, nm1] of [[17, 3, 2], [17, 2, 2], [17, 0, 0], [17, 9, 8], [1, 1, 0], [2, 1, 1], [167, 18, 18], [167, 61, 60], [509, 169, 169], [821, 273, 273], [701, 233, 233]]) {
    const { out, draws } = record(() => ref.generateCustomArray(len, n1, nm1));
    pure.sampler.push({ len, n1, nm1, draws, out });
  }
  {
    let error = null;
    try { ref.generateCustomArray(4, 3, 2); } catch (e) { error = e.message; }
    pure.sampler.push({ len: 4, n1: 3, nm1: 2, draws: [], out: null, error });
  }

  pure.misc.trim = [[0, 0], [], [1, 0, 2, 0, 0], [0]].map(a => ({ a, out: ref.trimPolynomial(a) }));
  pure.misc.degree = [[0, 0], [], [1, 0, 2, 0, 0], [0], [5]].map(a => ({ a, out: ref.degree(a) }));
  pure.misc.modInverse = [[-1, 3], [-1, 128], [-1, 4096], [3, 11], [2, 4], [7, 8192]].map(([a, p]) => ({ a, p, out: ref.modInverse(a, p) }));
  pure.misc.expandArray = [{ a: [1, 2], len: 5, fill: 0, out: ref.expandArray([1, 2], 5, 0) }];
  pure.misc.stringToBits = [{ s: 'Hello World', out: ref.stringToBits('Hello World') }];
  pure.misc.NqNp = [[167, 128], [509, 2048], [821, 4096], [701, 8192], [17, 32], [677, 2048]].map(([N, q]) => {
    const n = new NTRU({ N, q });
    return { N, q, p: 3, Nq: n.calculateNq(), Np: n.calculateNp() };
  });
  dump('pure_functions.json', pure);

  // ---- field-element packing (index.js:572-620; circuits CombineArray/UnpackArray, test/circuits.test.js:20-58) ----
  {
    const hex = v => v.toString(16);
    const pack = { packOutput: [], unpackInput: [] };
    reseed(0xBACC);
    for (const [maxVal, dataLen] of [[8192, 701], [8192, 17], [4096, 821], [2048, 509], [128, 167], [3, 821], [3, 17],
      [2, 5], [8191, 40], [65535, 33], [1, 7], [4096, 1], [4095, 63]]) {
      const data = randInts(dataLen, Math.min(maxVal + 1, 65536));
      const out = ref.packOutput(maxVal, dataLen, data);
      pack.packOutput.push({ maxVal, dataLen, data, maxInputBits: out.maxInputBits, maxOutputBits: out.maxOutputBits,
        outputSize: out.outputSize, arrLen: out.arrLen, expected: out.expected.map(hex) });
      const un = ref.unpackInput(maxVal, out.maxOutputBits, out.expected);
      pack.unpackInput.push({ maxVal, packedBits: out.maxOutputBits, data: out.expected.map(hex),
        maxInputBits: un.maxInputBits, packedSize: un.packedSize, unpackedSize: un.unpackedSize, unpacked: un.unpacked });
    }
    // the literal case of test/circuits.test.js:20-58: max 8192, N = 701
    dump('pack_functions.json', pack);
  }

  // ---- scheme-level vectors -------------------------------------------------
  const profiles = [
    { name: 'n17_q32', opt: { N: 17, q: 32, df: 3, dg: 2, dr: 2 }, keys: 3, seed: 0x1701 },
    { name: 'n167_q128', opt: {}, keys: 2, seed: 0x1671 },
    { name: 'n509_q2048', opt: { N: 509, q: 2048, df: 169, dg: 169, dr: 169 }, keys: 1, seed: 0x5091 },
    { name: 'n821_q4096', opt: { N: 821, q: 4096, df: 273, dg: 273, dr: 273 }, keys: 1, seed: 0x8211 },
    { name: 'n701_q8192', opt: { N: 701, q: 8192, df: 233, dg: 233, dr: 233 }, keys: 1, seed: 0x7011 },
  ];
  for (const prof of profiles) {
    reseed(prof.seed);
    const file = { profile: prof.name, options: null, keys: [] };
    for (let k = 0; k < prof.keys; k++) {
      const ntru = new NTRU(prof.opt);
      ntru.generatePrivateKeyF();
      ntru.generateNewPublicKeyGH();
      const { N, p, q, df, dg, dr } = ntru;
      file.options = { N, p, q, df, dg, dr };
      const key = { f: ntru.f, fp: ntru.fp, fq: ntru.fq, g: ntru.g, h: ntru.h, I: ntru.I, cases: [], sums: [] };
      key.verifyKeysInputs = ntru.verifyKeysInputs();

      const msgs = [];
      msgs.push(randInts(N, 2));                                 // full-length binary
      msgs.push(ref.stringToBits(N >= 88 ? 'Hello World' : 'Hi')); // short message (trailing-zero padding)
      msgs.push(randInts(N, 3));                                 // ternary incl. 2s
      msgs.push(new Array(N).fill(0));                           // all-zero plaintext
      msgs.push([]);                                             // empty plaintext
      msgs.push([1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1]); // test/circuits.test.js:258
      if (k === 0) msgs.push(randInts(N - 1, 2).concat([1]));    // top coefficient set
      for (const m of msgs) {
        const mIn = m.slice();
        const { out: enc, draws } = record(() => ntru.encryptBits(m));
        const dec = ntru.decryptBits(enc.value);
        key.cases.push({ m: mIn, draws, encrypt: enc, decrypt: dec });
      }
      // homomorphic sums (test/reference.test.js:46-61): decrypt input shorter/other than a fresh ciphertext
      const pairs = [[[1, 2, 1, 0, 1], [0, 1, 1, 1, 0, 1, 0, 1]], [msgs[0], msgs[2]]];
      for (const [m1, m2] of pairs) {
        const e1 = ntru.encryptBits(m1).value;
        const e2 = ntru.encryptBits(m2).value;
        const eSum = ref.addPolynomials(e1, e2, q);
        key.sums.push({ m1, m2, e1, e2, eSum, decrypt: ntru.decryptBits(eSum) });
      }
      // decrypt of degenerate ciphertexts
      key.degenerate = [[0], [1], [q - 1], new Array(N).fill(q - 1)].map(e => ({ e, decrypt: ntru.decryptBits(e) }));
      file.keys.push(key);
    }
    dump(`scheme_${prof.name}.json`, file);
  }
}

import(pathToFileURL(join(refDir, 'index.js')).href).then(main).catch(e => { console.error(e); process.exit(1); });
