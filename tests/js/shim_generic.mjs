// The reference's generic exports through the shim against vectors captured from the reference
// (tests/golden/generic_functions.json, keygen_cases.json).  Usage: node tests/js/shim_generic.mjs   (needs a GPU)
import { deepStrictEqual, strictEqual, throws } from 'assert';
import { readFileSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath } from 'url';

import NTRU, * as lib from '../../ntru-circom_amd/js/index.mjs';

const here = dirname(fileURLToPath(import.meta.url));
const golden = name => JSON.parse(readFileSync(join(here, '..', 'golden', name), 'utf8'));
const g = golden('generic_functions.json');
const esc = s => new RegExp(s.replace(/[.*+?^${}()|[\]\\]/g, '\\$&'));
const check = (c, fn) => { if (c.error) throws(fn, esc(c.error)); else deepStrictEqual(fn(), c.out); };

for (const c of g.modInverse) strictEqual(lib.modInverse(c.a, c.p), c.out);
for (const c of g.subtract) deepStrictEqual(lib.subtractPolynomials(c.a, c.b, c.p), c.out);
for (const c of g.scalar) deepStrictEqual(lib.multiplyPolynomialsByScalar(c.a, c.s, c.p).map(x => x + 0), c.out);   // -0 -> 0 as in JSON
for (const c of g.multiply) deepStrictEqual(lib.multiplyPolynomials(c.a, c.b, c.p), c.out);
for (const c of g.divide) check(c, () => lib.dividePolynomials(c.a, c.b, c.p));
for (const c of g.eea) check(c, () => lib.extendedEuclideanAlgorithm(c.a, c.b, c.p));
for (const c of g.polyInv) check(c, () => lib.polyInv(c.f, c.I, c.mod));
for (const c of g.bigintToBits) deepStrictEqual(lib.bigintToBits(BigInt(c.v)), c.out);
for (const c of g.bitsToBigInt) strictEqual(lib.bitsToBigInt(c.bits).toString(), c.out);

// every 2^20 product of test/circuits.test.js:60-72 and the four divisions of :165-170
const pure = golden('pure_functions.json');
let big = 0;
for (const v of pure.multiply) if (v.p > 65536) { deepStrictEqual(lib.multiplyPolynomials(v.a, v.b, v.p), v.out); big++; }
strictEqual(big >= 6, true);
for (const v of pure.divide) {
  if (v.N !== undefined) continue;
  if (v.error) throws(() => lib.dividePolynomials(v.a, v.b, v.p), esc(v.error)); else deepStrictEqual(lib.dividePolynomials(v.a, v.b, v.p), v.out);
}

// loadPrivateKeyF on arbitrary ternary f, non-units included: what the reference returns or throws
let accepted = 0, thrown = 0;
for (const c of golden('keygen_cases.json').cases) {
  const ntru = new NTRU({ N: c.N, q: c.q, p: c.p });
  if (c.error) { throws(() => ntru.loadPrivateKeyF(c.f), esc(c.error)); deepStrictEqual(ntru.f, c.f); thrown++; }
  else { strictEqual(ntru.loadPrivateKeyF(c.f), true); deepStrictEqual(ntru.fq, c.fq); deepStrictEqual(ntru.fp, c.fp); accepted++; }
}
console.log(`shim_generic: ${g.divide.length} divisions, ${g.eea.length} EEA, ${g.polyInv.length} polyInv, ${big} products mod 2^20, ` +
  `${accepted} accepted + ${thrown} rejected keys OK`);
