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
// generatePublicKeyH).  The whole export list of the reference is here (default NTRU + its 19 named functions):
// polyInv / extendedEuclideanAlgorithm / dividePolynomials by an arbitrary divisor / products modulo more than 65536 run
// on the engine's generic family (ntru_generic_*: the reference's own algorithms, one item per wavefront), which also
// serves the f that are not units, so that loadPrivateKeyF behaves like the reference there too.
import { createRequire } from 'module';
import nodeCrypto from 'crypto';
const { randomFillSync } = nodeCrypto;

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
  // The reference draws one u32 per step from the global WebCrypto (index.js:481-482).  A WebCrypto the CALLER installed
  // (e.g. a seeded one, to replay) is called exactly like that.  Node's own CSPRNG -- no global at all (Node 12), or the
  // built-in global of Node >= 19, which IS crypto.webcrypto -- delivers the length - 1 draws in ONE call: same count, same
  // distribution, ~0.4 ms less per encryptBits at N = 821 (length - 1 <= 16384 draws stay under WebCrypto's 65536-byte quota).
  const g = typeof globalThis !== 'undefined' && globalThis.crypto && globalThis.crypto.getRandomValues ? globalThis.crypto : null;
  const builtin = g !== null && nodeCrypto.webcrypto !== undefined && g === nodeCrypto.webcrypto;
  const perStep = g !== null && !builtin;
  const u = new Uint32Array(perStep || length < 2 ? 1 : length - 1);
  if (!perStep && length > 1) {
    if (builtin && u.byteLength <= 65536) g.getRandomValues(u); else randomFillSync(u);
  }
  for (let i = length - 1, k = 0; i > 0; i--, k++) {
    if (perStep) g.getRandomValues(u);
    const j = u[perStep ? 0 : k] % (i + 1);
    const t = array[i]; array[i] = array[j]; array[j] = t;
  }
  return array;
}

const mod = (x, p) => ((x % p) + p) % p;

// ---- the reference's remaining O(N) helpers (host glue, like addPolynomials) ------------------------------------

export function modInverse(a, p) {                               // index.js:224-232
  a = ((a % p) + p) % p;
  for (let x = 1; x < p; x++) if ((a * x) % p === 1) return x;
  return null;
}

export function subtractPolynomials(a, b, p) {                   // index.js:247-256
  const n = Math.max(a.length, b.length), out = [];
  for (let i = 0; i < n; i++) out[i] = (((i < a.length ? a[i] : 0) - (i < b.length ? b[i] : 0)) % p + p) % p;
  return trimPolynomial(out);
}

export function multiplyPolynomialsByScalar(poly, scalar, p) {   // index.js:404-406
  return poly.map(coeff => (coeff * scalar) % p);
}

export function bigintToBits(bigint) {                           // index.js:558-566
  const bits = [];
  while (bigint > 0n) { bits.push(Number(bigint & 1n)); bigint >>= 1n; }
  return bits;
}

export function bitsToBigInt(bits) { return BigInt(`0b${bits.join('')}`); }   // index.js:568-570

// ---- generic family of the engine (include/ntru_engine.h, ntru_generic_*): one item, JS Numbers in and out --------
const GENERIC_ERRORS = [null, 'Cannot divide by zero polynomial.', 'No inverse exists for division.', 'invalid_gcd',
  'ntru engine: generic work area exhausted'];
const REFERENCE_KEYGEN_ERRORS = new Set(['invalid_gcd', 'invalid fq', 'invalid fp', GENERIC_ERRORS[1], GENERIC_ERRORS[2]]);

function genericOp(op, a, b, mod) {
  const A = Float64Array.from(a), Bv = Float64Array.from(b);
  const cap = engine().genericCapacity(A.length, Bv.length);
  const o0 = new Float64Array(cap), o1 = op === 1 || op === 2 ? new Float64Array(cap) : null;
  const [status, len0, len1] = engine().genericOp(op, A, Bv, mod, o0, o1);
  if (status) throw new Error(GENERIC_ERRORS[status]);
  return [Array.from(o0.subarray(0, len0)), o1 ? Array.from(o1.subarray(0, len1)) : null];
}

const isI = (b, p) => {                                           // b = [1, 0, ..., 0, -1] (or p-1) of length N+1 >= 2
  const N = b.length - 1;
  if (N < 1 || mod(b[0], p) !== 1 || mod(b[N], p) !== p - 1) return false;
  for (let i = 1; i < N; i++) if (mod(b[i], p) !== 0) return false;
  return true;
};
const fastModulus = (N, p) => Number.isInteger(p) && p >= 2 &&
  ((p & (p - 1)) === 0 ? p <= 65536 : N * (p - 1) * (p - 1) < 65536);

// index.js:425-459 -> { gcd, inverse }
export function extendedEuclideanAlgorithm(a, b, p) {
  const [gcd, inverse] = genericOp(2, a, b, p);
  return { gcd, inverse };
}

// index.js:491-514.  The key-generation case -- ternary f, polyI = 1 - x^N, modulus 3 or a power of two -- runs on the
// batched inversion kernels (k_invert_key + Newton rounds on the matrix cores); the inverse of a unit is unique, so that
// is the reference's result.  Everything else, and every f those kernels flag as a non-unit, goes through the generic
// family, which follows the reference step by step (and throws what it throws).
export function polyInv(polyIn, polyI, polyMod) {
  const N = polyI.length - 1;
  const ternary = polyIn.length <= N && polyIn.every(x => x === 0 || x === 1 || x === -1);
  const pow2 = Number.isInteger(polyMod) && polyMod >= 2 && polyMod <= 65536 && (polyMod & (polyMod - 1)) === 0;
  if (ternary && N >= 2 && N <= 1920 && (polyMod === 3 || pow2) && polyI[0] === 1 && polyI[N] === -1 &&
      polyI.every((x, i) => i === 0 || i === N || x === 0)) {
    const f = Int8Array.from(expandArray(polyIn, N, 0)), flags = new Uint8Array(1);
    if (polyMod === 3) {
      const fp = new Uint8Array(N);
      engine().invertKeyBatch(N, 2, 3, f, 1, null, fp, flags);
      if (!(flags[0] & 16)) return trimPolynomial(Array.from(fp));
    } else {
      const fq = new Uint16Array(N);
      engine().invertKeyBatch(N, polyMod, 3, f, 1, fq, null, flags);
      if (!(flags[0] & 8)) return trimPolynomial(Array.from(fq));
    }
  }
  return genericOp(3, polyIn, polyI, polyMod)[0];
}

// index.js:319-355: linear product, coefficients into [0,p), trimmed.  One GPU polymul-split in a ring that
// holds both operands; c[N+k] = -quot[k], c[k] = rem[k] - c[N+k].  Moduli the packed kernels do not take (above 65536,
// e.g. the 2^20 of test/circuits.test.js:72) run on the generic family.
export function multiplyPolynomials(a, b, p) {
  if (a.length === 0 || b.length === 0) return [0];
  const N = Math.max(a.length, b.length, 2);
  if (!fastModulus(N, p)) return genericOp(0, a, b, p)[0];
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

// index.js:358-401.  The hot path's divisor b = I = 1 - x^N with a reduced dividend is the closed-form split kernel;
// any other divisor (or an unreduced dividend) is long division on the generic family.
export function dividePolynomials(a, b, p) {
  if (degree(b) === -1) throw new Error('Cannot divide by zero polynomial.');
  const N = b.length - 1;
  if (!(Number.isInteger(p) && p >= 2 && p <= 65536 && isI(b, p) && a.length <= 2 * N &&
        a.every(x => Number.isInteger(x) && x >= 0 && x < p))) {
    const [quotient, remainder] = genericOp(1, a, b, p);
    return { quotient, remainder };
  }
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
    // addPolynomials(m, rhq, q) reduces any integer m[i] modulo q (index.js:91, :241); the device adds a byte, and
    // (m mod q) mod 256 is the same residue modulo q whenever q divides 256 -- otherwise m mod q must fit the byte.
    const mDev = Uint8Array.from(mPad, x => { const v = mod(x, q); return q <= 256 ? v : v & 255; });
    if (q > 256 && mPad.some(x => mod(x, q) > 255)) return this.encryptBitsWide(m, r, mPad, hPad);
    const e = new Uint16Array(N), quot = new Uint16Array(N);
    engine().encryptBatch(N, q, Uint16Array.from(hPad), Uint8Array.from(r), mDev, 1, e, quot);
    return {
      value: trimPolynomial(Array.from(e)),
      inputs: { r, m: mPad, h: hPad, quotientE: withZero(quot), remainderE: withZero(e) },
      params: [q, this.calculateNq(), N],
    };
  }

  // Plaintext coefficients that do not fit the kernel's byte operand (the reference accepts any integer): the same
  // three steps as index.js:90-92 as separate engine calls.
  encryptBitsWide(m, r, mPad, hPad) {
    const { N, q } = this;
    const rhqm = addPolynomials(m, multiplyPolynomials(r, this.h, q), q);
    const { quotient, remainder } = dividePolynomials(rhqm, this.I, q);
    return {
      value: trimPolynomial(remainder),
      inputs: { r, m: mPad, h: hPad, quotientE: expandArray(quotient.map(x => x % q), N + 1, 0),
        remainderE: expandArray(remainder, N + 1, 0) },
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

  // index.js:30-49.  Units (every f a key generator ends up with) take the batched inversion kernels: fq and fp are
  // unique, so they are the reference's, and its validity checks pass by construction.  For an f those kernels flag,
  // the reference's own sequence runs on the generic family: same assignments in the same order, same throws, same
  // acceptance of the non-units its `&&` checks let through (index.js:41-45, :451).
  loadPrivateKeyF(fArr) {
    const { N, p, q } = this;
    const ternary = fArr.length <= N && fArr.every(x => x === 0 || x === 1 || x === -1);
    if (ternary && p === 3 && N >= 2 && N <= 1920 && fastModulus(N, q) && (q & (q - 1)) === 0) {   // the engine's N range, as in polyInv
      const fq = new Uint16Array(N), fp = new Uint8Array(N), flags = new Uint8Array(1);
      engine().invertKeyBatch(N, q, p, Int8Array.from(expandArray(fArr, N, 0)), 1, fq, fp, flags);
      if (!(flags[0] & 24)) {
        this.f = fArr;
        this.fq = trimPolynomial(Array.from(fq));
        this.fp = trimPolynomial(Array.from(fp));
        return true;
      }
    }
    this.f = fArr;
    this.fq = polyInv(this.f, this.I, q);
    this.fp = polyInv(this.f, this.I, p);
    const fmodq = this.f.map(x => x === -1 ? q - 1 : x), fmodp = this.f.map(x => x === -1 ? p - 1 : x);
    const fqDiv = dividePolynomials(multiplyPolynomials(this.fq, fmodq, q), this.I, q);
    if (fqDiv.remainder.length !== 1 && fqDiv.remainder[0] !== 1) throw new Error('invalid fq');
    const fpDiv = dividePolynomials(multiplyPolynomials(this.fp, fmodp, p), this.I, p);
    if (fpDiv.remainder.length !== 1 && fpDiv.remainder[0] !== 1) throw new Error('invalid fp');
    return true;
  }

  generatePrivateKeyF() {                                        // index.js:51-65
    const maxTries = 100;
    let i = 0, retval;
    while ((!retval || !(this.fq && this.fp)) && i++ < maxTries) {
      try {
        retval = this.loadPrivateKeyF(generateCustomArray(this.N, this.df, this.df - 1));
      } catch (error) {
        // the reference swallows everything here; only what IT can throw means "try the next f" -- an engine, addon
        // or GPU failure must surface instead of ending as 'Could not find invertible f'
        if (!REFERENCE_KEYGEN_ERRORS.has(error.message)) throw error;
      }
    }
    if (!this.fq || !this.fp) throw new Error('Could not find invertible f');
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
    engine().publicKeyBatch(N, q, p, Uint16Array.from(expandArray(this.fq, N, 0)), Int8Array.from(expandArray(this.g, N, 0)), 1, h);
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
  // Page-locked typed arrays (ntru_host_alloc): the batch calls DMA straight from / to them, no staging copy.
  // NTRU.useDevices([0, 1, ...]): encryptBatch / decryptBatch / verifyKeysInputs batches are cut into contiguous shards, one
  // engine + host thread per listed device (ntru_multi_*); [] goes back to the single device.  Returns the engine count.
  static useDevices(ids) { return engine().useDevices(Int32Array.from(ids)); }
  static allocUint8(n) { return new Uint8Array(engine().allocPinned(n), 0, n); }
  static allocUint16(n) { return new Uint16Array(engine().allocPinned(2 * n), 0, n); }

  // r: Uint8Array[B*N] in {0,1,2}, m: Uint8Array[B*N]  ->  { e, quotientE } as Uint16Array[B*N]
  // `out` may carry preallocated result arrays (e.g. from NTRU.allocUint16, which are page-locked and DMA'd in place).
  encryptBatch(r, m, B, wantWitness = true, out = {}) {
    const { N, q } = this;
    const e = out.e || new Uint16Array(B * N), quot = wantWitness ? (out.quotientE || new Uint16Array(B * N)) : null;
    engine().encryptBatch(N, q, Uint16Array.from(expandArray(this.h, N, 0)), r, m, B, e, quot);
    return { e, quotientE: quot };
  }

  // r for B encryptions drawn on the GPU: generateCustomArray(N, dr, dr) with -1 -> p-1 (index.js:89) on the ChaCha20
  // stream of `key` (Uint32Array[8]) for item indices firstItem .. firstItem+B-1; replayable on any host.
  // NTRU.samplerRounds(rounds): 20 (ChaCha20 as in RFC 8439, the default), 12 or 8 rounds of the block function behind sampleR and the
  // sampler stage of pipeline(); returns the count in force (no argument: only reads it).  Additive: the reference has no such knob because
  // its draws come from crypto.getRandomValues; generateCustomArray's shuffle and `u32 % (i + 1)` reduction are unchanged.
  static samplerRounds(rounds = 0) { return engine().setSamplerRounds(rounds); }

  sampleR(key, firstItem, B) {
    const r = new Uint8Array(B * this.N);
    engine().sampleTernary(this.N, this.dr, this.dr, this.p - 1, key, firstItem, B, r);
    return r;
  }

  // e: Uint16Array[B*N] -> { value, quotient1, remainder1, quotient2 }
  decryptBatch(e, B, wantWitness = true, out = {}) {
    const { N, p, q } = this;
    const value = out.value || new Uint8Array(B * N);
    const q1 = wantWitness ? (out.quotient1 || new Uint16Array(B * N)) : null;
    const r1 = wantWitness ? (out.remainder1 || new Uint16Array(B * N)) : null;
    const q2 = wantWitness ? (out.quotient2 || new Uint8Array(B * N)) : null;
    engine().decryptBatch(N, q, p, Int8Array.from(expandArray(this.f, N, 0)), Uint8Array.from(expandArray(this.fp, N, 0)),
      e, B, value, q1, r1, q2);
    return { value, quotient1: q1, remainder1: r1, quotient2: q2 };
  }

  // Device-resident pipeline for a batch of plaintexts (additive): the stages of the reference's encrypt / decrypt flow that the
  // caller names run back to back on the GPU, chunk by chunk, and only what is asked for crosses PCIe:
  //   sampleR: { key: Uint32Array[8], firstItem }   r = generateCustomArray(N, dr, dr) with -1 -> p-1 (index.js:89, :461-488) drawn on the
  //                                                 device from the ChaCha20 stream of `key` (replayable) -- or  r: Uint8Array[B*N]
  //   encrypt                                       always (index.js:87-110, this.h)
  //   decrypt: true                                 decryptBits of the fresh ciphertexts (index.js:111-140, this.f / this.fp)
  //   pack: true                                    packOutput (index.js:572-596) of the last stage's result: value (maxVal p-1) when
  //                                                 decrypting, else the ciphertext (maxVal q-1); BigUint64Array[B*outputSize*4] limbs
  //   want: { r, e, value }                         which plain arrays come back (default: e without decrypt, value with it, nothing
  //                                                 but `packed` when pack is set)
  // m: Uint8Array[B*N].  Arrays in `out` (e.g. from NTRU.allocUint8, page-locked) are filled in place.
  pipeline(opts) { return this._pipeline(opts, false); }
  // The same on a libuv worker thread: a Promise of the same result object; the arrays must be left alone until it settles.
  pipelineAsync(opts) { return this._pipeline(opts, true); }
  _pipeline({ m, B, sampleR = null, r = null, decrypt = false, pack = false, want = null, out = {} }, asynchronous) {
    const { N, p, q, dr } = this;
    if (!(m instanceof Uint8Array) || m.length < B * N) throw new TypeError('pipeline: m must be a Uint8Array of B*N plaintext coefficients');
    if ((sampleR === null) === (r === null)) throw new TypeError('pipeline: give either sampleR: {key, firstItem} or r');
    const w = want || (pack ? {} : (decrypt ? { value: true } : { e: true }));
    const res = {};
    if (w.r && sampleR) res.r = out.r || new Uint8Array(B * N);
    if (w.e) res.e = out.e || new Uint16Array(B * N);
    if (w.value) {
      if (!decrypt) throw new TypeError('pipeline: `value` needs decrypt: true');
      res.value = out.value || new Uint8Array(B * N);
    }
    if (pack) {
      const outputSize = engine().packParams(decrypt ? p - 1 : q - 1, N)[3];
      res.packed = out.packed || new BigUint64Array(B * outputSize * 4);
      res.outputSize = outputSize;
    }
    const args = [N, q, p, Uint16Array.from(expandArray(this.h, N, 0)),
      decrypt ? Int8Array.from(expandArray(this.f, N, 0)) : null, decrypt ? Uint8Array.from(expandArray(this.fp, N, 0)) : null,
      sampleR ? sampleR.key : null, sampleR ? (sampleR.firstItem || 0) : 0, dr, dr, r, m, B,
      res.r || null, res.e || null, res.value || null, res.packed || null];
    if (asynchronous) return engine().pipelineBatchAsync(...args).then(() => res);
    engine().pipelineBatch(...args);
    return res;
  }

  // Opaque device buffers + the engine's *_dev entry points on them, for callers that compose the stages themselves
  // (sizes are checked against every handle before a kernel is launched).
  static devAlloc(bytes) { return engine().devAlloc(bytes); }
  static devFree(handle) { engine().devFree(handle); }
  static devUpload(handle, typedArray) { engine().devUpload(handle, typedArray); }
  static devDownload(typedArray, handle) { engine().devDownload(typedArray, handle); return typedArray; }
  sampleRDev(key, firstItem, B, rDev) { engine().sampleTernaryDev(this.N, this.dr, this.dr, this.p - 1, key, firstItem, B, rDev); }
  encryptBatchDev(hDev, rDev, mDev, B, eDev, quotDev = null) { engine().encryptBatchDev(this.N, this.q, hDev, rDev, mDev, B, eDev, quotDev); }
  decryptBatchDev(fDev, fpDev, eDev, B, valueDev, q1Dev = null, r1Dev = null, q2Dev = null) {
    engine().decryptBatchDev(this.N, this.q, this.p, fDev, fpDev, eDev, B, valueDev, q1Dev, r1Dev, q2Dev);
  }
  static packBatchDev(maxVal, dataLen, dataDev, B, outDev, bytes = false) { engine().packBatchDev(maxVal, dataLen, dataDev, B, outDev, bytes); }

  // Promise-returning twins of the two batch calls: the engine call runs on a libuv worker thread, the event loop keeps
  // turning meanwhile (calls are serialised inside the addon: one engine).  Same arguments, same results; the input and
  // output arrays must be left alone until the Promise settles.
  encryptBatchAsync(r, m, B, wantWitness = true, out = {}) {
    const { N, q } = this;
    const e = out.e || new Uint16Array(B * N), quot = wantWitness ? (out.quotientE || new Uint16Array(B * N)) : null;
    return engine().encryptBatchAsync(N, q, Uint16Array.from(expandArray(this.h, N, 0)), r, m, B, e, quot)
      .then(() => ({ e, quotientE: quot }));
  }

  decryptBatchAsync(e, B, wantWitness = true, out = {}) {
    const { N, p, q } = this;
    const value = out.value || new Uint8Array(B * N);
    const q1 = wantWitness ? (out.quotient1 || new Uint16Array(B * N)) : null;
    const r1 = wantWitness ? (out.remainder1 || new Uint16Array(B * N)) : null;
    const q2 = wantWitness ? (out.quotient2 || new Uint8Array(B * N)) : null;
    return engine().decryptBatchAsync(N, q, p, Int8Array.from(expandArray(this.f, N, 0)),
      Uint8Array.from(expandArray(this.fp, N, 0)), e, B, value, q1, r1, q2)
      .then(() => ({ value, quotient1: q1, remainder1: r1, quotient2: q2 }));
  }
}
