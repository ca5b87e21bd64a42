// Golden vectors for the reference's generic polynomial helpers: runs the UNMODIFIED reference under Node and records
// inputs + outputs (or the thrown message) of
//   modInverse (index.js:224), subtractPolynomials (:247), multiplyPolynomialsByScalar (:404),
//   dividePolynomials with arbitrary divisors and signed / unreduced operands (:358), multiplyPolynomials with moduli
//   above 65536 (:319), extendedEuclideanAlgorithm (:425, incl. the worked example of :411-423), polyInv (:491),
//   bigintToBits (:558), bitsToBigInt (:568).
// Only the resulting JSON is committed.  (-0, which the reference can produce through `%`, serialises as 0.)
//
//   node tests/golden/gen_generic_cases.mjs [/root/reference] [outdir]
import { writeFileSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath, pathToFileURL } from 'url';

const here = dirname(fileURLToPath(import.meta.url));
const refDir = process.argv[2] || '/root/reference';
const outDir = process.argv[3] || here;
let state = 0x9E3779B9;
function nextU32() { let x = state; x ^= x << 13; x >>>= 0; x ^= x >>> 17; x ^= x << 5; x >>>= 0; state = x; return x; }
globalThis.crypto = { getRandomValues(arr) { for (let i = 0; i < arr.length; i++) arr[i] = nextU32(); return arr; } };
const rnd = (n) => nextU32() % n;
const ints = (len, lo, hi) => Array.from({ length: len }, () => lo + rnd(hi - lo + 1));

function attempt(fn) {
  try { return { out: fn() }; } catch (e) { return { error: String(e.message) }; }
}

async function main() {
  const ref = await import(pathToFileURL(join(refDir, 'index.js')).href);
  const out = { generator: 'gen_generic_cases.mjs', modInverse: [], subtract: [], scalar: [], divide: [], multiply: [],
    eea: [], polyInv: [], bigintToBits: [], bitsToBigInt: [] };

  for (const [a, p] of [[3, 11], [-1, 3], [-1, 128], [2, 4], [0, 7], [7, 8192], [5, 1], [-9, 10], [123456, 1048576], [123457, 1048576],
    [14, 15], [10, 15], [1, 2], [64, 127]])
    out.modInverse.push({ a, p, out: ref.modInverse(a, p) });

  for (const [a, b, p] of [[[1, 2, 3], [3, 2, 1], 5], [[1], [0, 0, 4], 7], [[], [], 3], [[-5, 9], [2], 4], [[2, 2], [2, 2], 9],
    [[0, 0, 1], [0, 0, 1, 0], 2], [[7, -8, 100], [-3, 5], 128]])
    out.subtract.push({ a, b, p, out: ref.subtractPolynomials(a, b, p) });

  for (const [a, s, p] of [[[1, 2, 3], 2, 4], [[-1, 5, 0], 3, 4], [[], 2, 8], [[4095, 1, 0], 3, 4096], [[1, -2], -3, 5]])
    out.scalar.push({ a, s, p, out: ref.multiplyPolynomialsByScalar(a, s, p) });

  // long division: prime, power-of-two and composite moduli; signed and unreduced operands; short dividends; zero
  // divisors; leading coefficients without an inverse
  const divSets = [[7, 8, 3], [11, 12, 6], [2, 24, 9], [3, 17, 17], [32, 10, 4], [4096, 30, 7], [15, 9, 3], [1048576, 12, 5],
    [101, 40, 33], [128, 5, 9]];
  for (const [p, la, lb] of divSets) {
    for (let t = 0; t < 6; t++) {
      const a = ints(t === 4 ? Math.max(lb - 2, 0) : la, t % 2 ? -p : 0, t === 3 ? 2 * p : p - 1);
      const b = ints(lb, t % 3 === 2 ? -p : 0, p - 1);
      if (t === 5) b[b.length - 1] = 0;                              // trailing zero: the degree is below the length
      out.divide.push({ a, b, p, ...attempt(() => ref.dividePolynomials(a, b, p)) });
    }
  }
  out.divide.push({ a: [1, 2, 3], b: [0, 0, 0], p: 5, ...attempt(() => ref.dividePolynomials([1, 2, 3], [0, 0, 0], 5)) });
  out.divide.push({ a: [], b: [1, 1], p: 5, ...attempt(() => ref.dividePolynomials([], [1, 1], 5)) });
  out.divide.push({ a: [0, 0], b: [3], p: 7, ...attempt(() => ref.dividePolynomials([0, 0], [3], 7)) });
  out.divide.push({ a: [4, 4, 4], b: [2], p: 8, ...attempt(() => ref.dividePolynomials([4, 4, 4], [2], 8)) });

  // products with moduli above 65536 (test/circuits.test.js:72 uses 2^20) and other moduli outside the packed kernels
  for (const [p, la, lb, mag] of [[1048576, 6, 6, 1048576], [1048576, 33, 17, 1048576], [1048576, 64, 64, 4096], [65537, 20, 20, 65537],
    [1000003, 9, 31, 1000003], [67108864, 12, 12, 8192], [97, 50, 50, 97], [131072, 1, 40, 131072]]) {
    const a = ints(la, 0, mag - 1), b = ints(lb, 0, mag - 1);
    out.multiply.push({ a, b, p, out: ref.multiplyPolynomials(a, b, p) });
    const an = ints(la, -50, 50), bn = ints(lb, -50, 50);
    out.multiply.push({ a: an, b: bn, p, out: ref.multiplyPolynomials(an, bn, p) });
  }

  // extendedEuclideanAlgorithm: the worked example, then random pairs over primes, 2 and composites
  out.eea.push({ a: [4, 2, 0, 3], b: [3, 2, 1], p: 11, ...attempt(() => ref.extendedEuclideanAlgorithm([4, 2, 0, 3], [3, 2, 1], 11)) });
  for (const [p, la, lb, count] of [[11, 5, 4, 6], [2, 9, 10, 8], [3, 8, 9, 8], [7, 3, 12, 4], [13, 12, 5, 4], [4, 5, 6, 4], [9, 4, 5, 4],
    [127, 20, 21, 3]]) {
    for (let t = 0; t < count; t++) {
      const a = ints(la, t % 2 ? -1 : 0, t % 4 === 3 ? p : p - 1), b = ints(lb, 0, p - 1);
      if (t % 3 === 0) b[b.length - 1] = 1;
      out.eea.push({ a, b, p, ...attempt(() => ref.extendedEuclideanAlgorithm(a, b, p)) });
    }
  }
  out.eea.push({ a: [1, 1], b: [0], p: 3, ...attempt(() => ref.extendedEuclideanAlgorithm([1, 1], [0], 3)) });
  out.eea.push({ a: [0], b: [0, 0], p: 3, ...attempt(() => ref.extendedEuclideanAlgorithm([0], [0, 0], 3)) });

  // polyInv: ternary f against I = 1 - x^N for several moduli (units and non-units), then non-ternary inputs, moduli
  // that are neither 3 nor a power of two, modulus polynomials other than I
  const I = n => { const v = new Array(n + 1).fill(0); v[0] = 1; v[n] = -1; return v; };
  for (const [N, mod, count] of [[7, 3, 5], [7, 32, 5], [11, 2, 4], [11, 8, 4], [16, 128, 4], [17, 3, 4], [17, 2048, 4], [23, 65536, 2],
    [13, 7, 4], [13, 5, 3], [10, 3, 4], [12, 64, 4], [31, 4096, 2], [9, 1, 2], [19, 131072, 2]]) {
    for (let t = 0; t < count; t++) {
      const f = ints(N, -1, 1);
      if (t === count - 1) f.fill(0, Math.floor(N / 2));
      out.polyInv.push({ f, I: I(N), mod, ...attempt(() => ref.polyInv(f, I(N), mod)) });
    }
  }
  for (const [f, m, mod] of [[[4, 2, 0, 3], [3, 2, 1], 11], [[2, 5, 1], [1, 0, 0, 1], 7], [[1, 1, 0, 1], [1, 0, 0, 0, -1], 16],
    [[3, -2, 7, 1, 0], I(6), 8], [[1, 1], [1, 1, 1], 2], [[5, 6, 7], I(5), 13], [[2, 0, 2], I(4), 4], [[1, 2, 3, 4, 5, 6, 7, 8, 9], I(9), 1024]])
    out.polyInv.push({ f, I: m, mod, ...attempt(() => ref.polyInv(f, m, mod)) });

  for (const v of [0n, 1n, 2n, 5n, 255n, 256n, (1n << 64n) + 3n, (1n << 251n) - 1n])
    out.bigintToBits.push({ v: v.toString(), out: ref.bigintToBits(v) });
  for (const bits of [[1], [0], [1, 0, 1], [0, 0, 1, 1], ref.bigintToBits(123456789012345678901234567890n).reverse()])
    out.bitsToBigInt.push({ bits, out: ref.bitsToBigInt(bits).toString() });

  writeFileSync(join(outDir, 'generic_functions.json'), JSON.stringify(out));
  const count = k => `${out[k].length} ${k} (${out[k].filter(c => c.error).length} errors)`;
  console.log('generic_functions.json:', ['modInverse', 'subtract', 'scalar', 'divide', 'multiply', 'eea', 'polyInv'].map(count).join(', '));
  console.log('errors seen:', [...new Set([...out.divide, ...out.eea, ...out.polyInv].filter(c => c.error).map(c => c.error))]);
}
main().catch(e => { console.error(e); process.exit(1); });
