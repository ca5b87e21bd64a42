// ES-module shim: the hot-path surface of numtel/ntru-circom's index.js served by the MI355X HIP engine.
//
//   import NTRU, { multiplyPolynomials, dividePolynomials, addPolynomials, ... } from './index.mjs'
//
// Same contract as the reference for the functions it covers (file:line = reference index.js):
//   * every return value is a fresh plain Array of Numbers / plain object (test/reference.test.js:60 uses
//     deepStrictEqual on a plain Array; circomkit takes number arrays as signals);
//   * inputs are never mutated; everything is synchronous; errors are `throw new Error(msg)` with the
//     reference's messages; expandArray overflow surfaces as RangeError (index.js:535);
//   * constructor options / public mutable fields N,p,q,df,dg,dr,f,fp,fq,g,h,I (index.js:9-27).
// All polynomial products and splits run on the GPU through the N-API addon -> C ABI (include/ntru_engine.h).
// There is no JavaScript arithmetic fallback: without the addon or a GPU the first call throws.
//
// Key generation runs on the device too (loadPrivateKeyF, generatePrivateKeyF, generateNewPublicKeyGH,
// generatePublicKeyH); the stand-alone helpers polyInv / extendedEuclideanAlgorithm / generic long division are not exported.
// Keys are supplied through the options object, as README.md:81 of the reference already allows.
import { createRequire } from 'module';
import { randomFillSync } from 'crypto';

const require = createRequire(import.meta.url);
let addon = null;
let engineReady = false;

function engine() {
  if (!addon) addon = require('./ntru_addon.node');       // throws if the addon was not built
  if (!engineReady) {
    addon.create(Number(process.env.NTRU_HIP_DEVICE || 0)); // throws when no GPU: no CPU fallback
    engineReady = true;
  }
  return addon;
}

export function deviceCount() {
  if (!addon) addon = require('./ntru_addon.node');
  return addon.deviceCount();
}

// ---- helpers with the reference's semantics -----------------------------------------------------------------

export function degree(poly) {                                   // index.js:210-215
  for (let i = poly.length - 1; i >= 0; i--) if (poly[i] !== 0) return i;
  return -1;
}

export function trimPolynomial(poly) {                           // index.js:218-221
  const d = degree(poly);
  return d >= 0 ? Array.prototype.slice.call(poly, 0, d + 1) : [0];
}

export function expandArray(arr, len, fill) {                    // index.js:534-536 (RangeError when too long)
  return [...arr, ...Array(len - arr.length).fill(fill)];
}

export function expandArrayToMultiple(array, multiple) {         // index.js:516-532 (mutates, like the reference)
  if (!Array.isArray(array)) throw new Error('First argument must be an array.');
  if (typeof multiple !== 'number' || multiple <= 0 || !Number.isInteger(multiple))
    throw new Error('Multiple must be a positive integer.');
  const target = Math.ceil(array.length / multiple) * multiple;
  while (array.length < target) array.push(0);
  return array;
}

export function addPolynomials(a, b, p) {                        // index.js:235-244 (O(N) host glue)
  const n = Math.max(a.length, b.length), out = [];
  for (let i = 0; i < n; i++) out[i] = (((i < a.length ? a[i] : 0) + (i < b.length ? b[i] : 0)) % p + p) % p;
  return trimPolynomial(out);
}

export function stringToBits(str) {                              // index.js:538-546
  const bits = [];
  for (let i = 0; i < str.length; i++)
    for (const c of str.charCodeAt(i).toString(2).padStart(8, '0')) bits.push(Number(c));
  return bits;
}

export function bitsToString(bits) {                             // index.js:548-556
  let s = '';
  for (let i = 0; i < bits.length; i += 8) s += String.fromCharCode(parseInt(bits.slice(i, i + 8).join(''), 2));
  return s;
}

// index.js:461-488: exactly length-1 draws, i descending, j = u32 % (i+1).  Uses WebCrypto when the host
// provides it (so a seeded shim replays the reference bit for bit) and node's CSPRNG otherwise.
export function generateCustomArray(length, numOnes, numNegOnes) {
  if (numOnes + numNegOnes > length) throw new Error('The total of 1s and -1s cannot exceed the array length.');
  const array = new Array(length).fill(0);
  array.fill(1, 0, numOnes);
  array.fill(-1, numOnes, numOnes + numNegOnes);
  const u = new Uint32Array(1);
  const webcrypto = typeof globalThis !== 'undefined' && globalThis.crypto && globalThis.crypto.getRandomValues
    ? globalThis.crypto : null;
  for (let i = length - 1; i > 0; i--) {
    if (webcrypto) webcrypto.getRandomValues(u); else randomFillSync(u);
    const j = u[0] % (i + 1);
    const t = array[i]; array[i] = array[j]; array[j] = t;
  }
  return array;
}

const mod = (x, p) => ((x % p) + p) % p;

// index.js:319-355: linear product, coefficients into [0,p), trimmed.  One GPU polymul-split in a ring that
// holds both operands; c[N+k] = -quot[k], c[k] = rem[k] - c[N+k].
export function multiplyPolynomials(a, b, p) {
  if (a.length === 0 || b.length === 0) return [0];
  const N = Math.max(a.length, b.length, 2);
  const A = new Uint16Array(N), Bv = new Uint16Array(N), quot = new Uint16Array(N), rem = new Uint16Array(N);
  for (let i = 0; i < a.length; i++) A[i] = mod(a[i], p);
  for (let i = 0; i < b.length; i++) Bv[i] = mod(b[i], p);
  engine().polymulSplit(N, p, A, Bv, 1, quot, rem);
  const out = new Array(2 * N - 1);
  for (let k = 0; k < N; k++) {
    const hi = (p - quot[k]) % p;
    out[k] = mod(rem[k] - hi, p);
    if (k < N - 1) out[N + k] = hi;
  }
  return trimPolynomial(out);
}

// index.js:358-401 for the hot path's divisor b = I = 1 - x^N; generic long division is key-generation code.
export function dividePolynomials(a, b, p) {
  if (degree(b) === -1) throw new Error('Cannot divide by zero polynomial.');
  const N = b.length - 1;
  let isI = N >= 1 && mod(b[0], p) === 1 && mod(b[N], p) === p - 1;
  for (let i = 1; isI && i < N; i++) isI = mod(b[i], p) === 0;
  if (!isI || a.length > 2 * N || a.some(x => x < 0 || x >= p))
    throw new Error('ntru engine: dividePolynomials only supports b = 1 - x^N with a reduced dividend of length <= 2N');
  const A = new Uint16Array(2 * N), quot = new Uint16Array(N), rem = new Uint16Array(N);
  A.set(a);
  engine().splitByI(N, p, A, 1, quot, rem);
  const nq = Math.max(a.length - N, 0);
  return {
    quotient: nq ? trimPolynomial(Array.from(quot.subarray(0, nq))) : [0],
    remainder: trimPolynomial(Array.from(rem)),
  };
}

// addPolynomials on the GPU for reduced operands (the homomorphic sum of test/reference.test.js:46-61).
export function addCiphertexts(e1, e2, q) {
  const N = Math.max(e1.length, e2.length, 1);
  const A = new Uint16Array(N), Bv = new Uint16Array(N), out = new Uint16Array(N);
  A.set(e1); Bv.set(e2);
  engine().addBatch(N, q, A, Bv, 1, out);
  return trimPolynomial(Array.from(out));
}

const limbsToBigInt = (l, at) => l[at] | (l[at + 1] << 64n) | (l[at + 2] << 128n) | (l[at + 3] << 192n);

// index.js:572-596 on the GPU: same object as the reference (`expected` is an Array of BigInt).
export function packOutput(maxVal, dataLen, data) {
  const [maxInputBits, numInputsPerOutput, arrLen, outputSize] = engine().packParams(maxVal, dataLen);
  const inArr = Uint16Array.from(expandArray(data, dataLen, 0));
  const limbs = new BigUint64Array(outputSize * 4);
  engine().packBatch(maxVal, dataLen, inArr, 1, limbs);
  const expected = [];
  for (let o = 0; o < outputSize; o++) expected.push(limbsToBigInt(limbs, 4 * o));
  return { maxInputBits, maxOutputBits: numInputsPerOutput * maxInputBits, outputSize, arrLen, expected };
}

// index.js:598-620 on the GPU (`data`: Array of BigInt below 2^256).
export function unpackInput(maxVal, packedBits, data) {
  const [maxInputBits] = engine().packParams(maxVal, 0);
  const per = Math.floor(packedBits / maxInputBits);
  const limbs = new BigUint64Array(data.length * 4);
  const M = (1n << 64n) - 1n;
  data.forEach((v, i) => { for (let k = 0; k < 4; k++) limbs[4 * i + k] = (v >> BigInt(64 * k)) & M; });
  const out = new Uint16Array(per * data.length);
  if (data.length) engine().unpackBatch(maxVal, packedBits, limbs, data.length, 1, out);
  return { maxInputBits, packedBits, packedSize: data.length, unpackedSize: per * data.length,
    unpacked: trimPolynomial(Array.from(out)) };
}

const withZero = ta => { const a = Array.from(ta); a.push(0); return a; };

export default class NTRU {
  constructor(options) {
    Object.assign(this, {                                        // index.js:9-23
      N: 167, p: 3, q: 128, df: 61, dg: 20, dr: 18, f: null, fp: null, fq: null, g: null, h: null,
    }, options);
    this.I = (new Array(this.N + 1)).fill(0);                    // index.js:25-27
    this.I[0] = 1;
    this.I[this.I.length - 1] = -1;
  }

  calculateNq() { return Math.ceil(Math.log2(this.q * this.q * this.N)); }   // index.js:201-203
  calculateNp() { return Math.ceil(Math.log2(this.p * this.p * this.N)); }   // index.js:204-206

  encryptStr(inputPlain) { return this.encryptBits(stringToBits(inputPlain)).value; }               // :80-83
  decryptStr(encrypted) { return bitsToString(expandArrayToMultiple(this.decryptBits(encrypted).value, 8)); } // :84-86

  encryptBits(m) {                                               // index.js:87-110
    const { N, p, q } = this;
    const r = generateCustomArray(N, this.dr, this.dr).map(x => x === -1 ? p - 1 : x);
    const mPad = expandArray(m, N, 0), hPad = expandArray(this.h, N, 0);
    if (mPad.some(x => !(x >= 0 && x <= 255))) throw new Error('ntru engine: plaintext coefficients must be in 0..255');
    const e = new Uint16Array(N), quot = new Uint16Array(N);
    engine().encryptBatch(N, q, Uint16Array.from(hPad), Uint8Array.from(r), Uint8Array.from(mPad), 1, e, quot);
    return {
      value: trimPolynomial(Array.from(e)),
      inputs: { r, m: mPad, h: hPad, quotientE: withZero(quot), remainderE: withZero(e) },
      params: [q, this.calculateNq(), N],
    };
  }

  decryptBits(e) {                                               // index.js:111-140
    const { N, p, q } = this;
    const f = this.f.map(x => x === -1 ? q - 1 : x);             // TypeError when f is null, like the reference
    const ePad = expandArray(e, N, 0), fpPad = expandArray(this.fp, N, 0);
    const value = new Uint8Array(N), q1 = new Uint16Array(N), r1 = new Uint16Array(N), q2 = new Uint8Array(N);
    engine().decryptBatch(N, q, p, Int8Array.from(expandArray(this.f, N, 0)), Uint8Array.from(fpPad),
      Uint16Array.from(ePad), 1, value, q1, r1, q2);
    return {
      value: trimPolynomial(Array.from(value)),
      inputs: {
        f: expandArray(f, N, 0), fp: fpPad, e: ePad,
        quotient1: withZero(q1), remainder1: withZero(r1), quotient2: withZero(q2), remainder2: withZero(value),
      },
      params: [q, this.calculateNq(), p, this.calculateNp(), N],
    };
  }

  loadPrivateKeyF(fArr) {                                        // index.js:30-49: fq, fp by inversion on the device
    const { N, p, q } = this;
    const fq = new Uint16Array(N), fp = new Uint8Array(N), flags = new Uint8Array(1);
    addon.invertKeyBatch(N, q, p, Int8Array.from(expandArray(fArr, N, 0)), 1, fq, fp, flags);
    // not a unit: the reference throws 'invalid_gcd' / 'invalid fq' for most such f (and accepts a few by accident)
    if (flags[0] & 24) throw new Error('invalid_gcd');
    this.f = fArr;
    this.fq = trimPolynomial(Array.from(fq));
    this.fp = trimPolynomial(Array.from(fp));
    return true;
  }

  generatePrivateKeyF() {                                        // index.js:51-65
    for (let i = 0; i < 100; i++) {
      try { return this.loadPrivateKeyF(generateCustomArray(this.N, this.df, this.df - 1)); } catch (error) { /* next f */ }
    }
    throw new Error('Could not find invertible f');
  }

  generateNewPublicKeyGH() {                                     // index.js:67-70
    this.g = generateCustomArray(this.N, this.dg, this.dg);
    this.generatePublicKeyH();
  }

  generatePublicKeyH() {                                         // index.js:72-79
    if (!this.f) throw new Error('missing private key F');
    if (!this.g) throw new Error('missing private key G');
    const { N, p, q } = this;
    const h = new Uint16Array(N);
    addon.publicKeyBatch(N, q, p, Uint16Array.from(expandArray(this.fq, N, 0)), Int8Array.from(expandArray(this.g, N, 0)), 1, h);
    this.h = trimPolynomial(Array.from(h));
  }

  verifyKeysInputs() {                                           // index.js:141-197
    if (!this.f) throw new Error('missing private key F');
    if (!this.fq) throw new Error('missing private key Fq');
    if (!this.fp) throw new Error('missing private key Fp');
    if (!this.g) throw new Error('missing private key G');
    if (!this.h) throw new Error('missing public key H');
    const { N, p, q } = this;
    const nq = this.calculateNq(), np = this.calculateNp();
    const pad = a => expandArray(a, N, 0);
    const out = {
      qfq: new Uint16Array(N), rfq: new Uint16Array(N), qfp: new Uint8Array(N), rfp: new Uint8Array(N),
      qh: new Uint16Array(N), rh: new Uint16Array(N), flags: new Uint8Array(1),
    };
    engine().verifyKeysBatch(N, q, p, Int8Array.from(pad(this.f)), Int8Array.from(pad(this.g)),
      Uint16Array.from(pad(this.fq)), Uint8Array.from(pad(this.fp)), Uint16Array.from(pad(this.h)), 1,
      out.qfq, out.rfq, out.qfp, out.rfp, out.qh, out.rh, out.flags);
    if (out.flags[0] & 1) throw new Error('invalid fq');
    if (out.flags[0] & 2) throw new Error('invalid fp');
    const remH = trimPolynomial(Array.from(out.rh));            // index.js:165 walks h exactly as stored
    if (this.h.reduce((bad, cur, index) => bad || remH[index] !== cur, false)) throw new Error('invalid h');
    return {
      fq: { params: [q, nq, N], inputs: { f: pad(this.f.map(x => x === -1 ? q - 1 : x)), fq: pad(this.fq),
        quotientI: withZero(out.qfq), remainderI: withZero(out.rfq) } },
      fp: { params: [p, np, N], inputs: { f: pad(this.f.map(x => x === -1 ? p - 1 : x)), fq: pad(this.fp),
        quotientI: withZero(out.qfp), remainderI: withZero(out.rfp) } },
      h: { params: [q, nq, N], inputs: { f: pad(this.g.map(x => x === -1 ? q - 1 : x)), fq: pad(this.fq.map(x => x * p)),
        quotientI: withZero(out.qh), remainderI: withZero(out.rh) } },
    };
  }

  // ---- additive batch API (typed arrays, fixed stride N; see include/ntru_engine.h for the layout) -----------
  // r: Uint8Array[B*N] in {0,1,2}, m: Uint8Array[B*N]  ->  { e, quotientE } as Uint16Array[B*N]
  encryptBatch(r, m, B, wantWitness = true) {
    const { N, q } = this;
    const e = new Uint16Array(B * N), quot = wantWitness ? new Uint16Array(B * N) : null;
    engine().encryptBatch(N, q, Uint16Array.from(expandArray(this.h, N, 0)), r, m, B, e, quot);
    return { e, quotientE: quot };
  }

  // r for B encryptions drawn on the GPU: generateCustomArray(N, dr, dr) with -1 -> p-1 (index.js:89) on the ChaCha20
  // stream of `key` (Uint32Array[8]) for item indices firstItem .. firstItem+B-1; replayable on any host.
  sampleR(key, firstItem, B) {
    const r = new Uint8Array(B * this.N);
    engine().sampleTernary(this.N, this.dr, this.dr, this.p - 1, key, firstItem, B, r);
    return r;
  }

  // e: Uint16Array[B*N] -> { value, quotient1, remainder1, quotient2 }
  decryptBatch(e, B, wantWitness = true) {
    const { N, p, q } = this;
    const value = new Uint8Array(B * N);
    const q1 = wantWitness ? new Uint16Array(B * N) : null, r1 = wantWitness ? new Uint16Array(B * N) : null;
    const q2 = wantWitness ? new Uint8Array(B * N) : null;
    engine().decryptBatch(N, q, p, Int8Array.from(expandArray(this.f, N, 0)), Uint8Array.from(expandArray(this.fp, N, 0)),
      e, B, value, q1, r1, q2);
    return { value, quotient1: q1, remainder1: r1, quotient2: q2 };
  }
}
