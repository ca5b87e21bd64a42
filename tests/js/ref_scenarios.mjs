// The four scenarios the reference's own suite checks (test/reference.test.js:6-61), run against the shim -- ONE
// scenario per fresh node process, so that key generation is the very first engine call of the process:
//   node tests/js/ref_scenarios.mjs roundtrip | wrongkey | large | homomorphic | firstcall-<method>
// Randomness is node's CSPRNG, as for a user of the package (no seeded crypto shim here).
import { deepStrictEqual, strictEqual, notStrictEqual, ok } from 'assert';

import NTRU, * as lib from '../../ntru-circom_amd/js/index.mjs';

const scenario = process.argv[2];
const freshKeyPair = (options) => {
  const ntru = new NTRU(options);
  ntru.generatePrivateKeyF();                 // first engine call of the process
  ntru.generateNewPublicKeyGH();
  return ntru;
};

if (scenario === 'roundtrip') {                                  // reference.test.js:6-13, default parameters
  const ntru = freshKeyPair();
  ok(Array.isArray(ntru.fq) && Array.isArray(ntru.fp) && Array.isArray(ntru.h));
  strictEqual(ntru.decryptStr(ntru.encryptStr('Hello World')), 'Hello World');
  ntru.verifyKeysInputs();                                       // throws unless fq, fp, h fit f and g
} else if (scenario === 'wrongkey') {                            // reference.test.js:15-25
  const enc = freshKeyPair();
  const encrypted = enc.encryptStr('Hello World');
  const dec = new NTRU();
  dec.generatePrivateKeyF();
  notStrictEqual(dec.decryptStr(encrypted), 'Hello World');
} else if (scenario === 'large') {                               // reference.test.js:27-44 (its GO_LARGE case)
  const d = Math.floor(701 / 3);
  const ntru = freshKeyPair({ N: 701, q: 8192, df: d, dg: d, dr: d });
  strictEqual(ntru.decryptStr(ntru.encryptStr('Big polys')), 'Big polys');
} else if (scenario === 'homomorphic') {                         // reference.test.js:46-61 ("this test may fail")
  // The reference warns that the sum of two ciphertexts may fail to decrypt; with the default parameters the wrap
  // probability per attempt is small, so a handful of fresh keys is allowed before calling it a failure.
  let good = false;
  for (let attempt = 0; attempt < 5 && !good; attempt++) {
    const ntru = freshKeyPair();
    const e1 = ntru.encryptBits([1, 2, 1, 0, 1]).value, e2 = ntru.encryptBits([0, 1, 1, 1, 0, 1, 0, 1]).value;
    const decrypted = ntru.decryptBits(lib.addPolynomials(e1, e2, ntru.q));
    ok(Array.isArray(decrypted.value));
    try { deepStrictEqual(decrypted.value, [1, 0, 2, 1, 1, 1, 0, 1]); good = true; } catch (e) { /* decryption failure */ }
  }
  ok(good, 'additive homomorphism failed on 5 fresh keys');
} else if (scenario && scenario.startsWith('firstcall-')) {
  // every key-generation method as the first call of a process (round 1 failed here: the addon was still null)
  const m = scenario.slice('firstcall-'.length);
  const ntru = new NTRU({ N: 17, q: 32, df: 3, dg: 2, dr: 2 });
  if (m === 'loadPrivateKeyF') {
    strictEqual(ntru.loadPrivateKeyF([1, 1, 0, -1, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0]), true);
    ok(ntru.fq.length > 0 && ntru.fp.length > 0);
  } else if (m === 'generatePublicKeyH') {
    ntru.f = [1, 1, 0, -1, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0]; ntru.fq = [1, 2, 3]; ntru.g = [1, -1, 0, 1];
    ntru.generatePublicKeyH();
    ok(Array.isArray(ntru.h));
  } else if (m === 'polyInv') {
    deepStrictEqual(lib.polyInv([4, 2, 0, 3], [3, 2, 1], 11), [5, 8]);       // index.js:411-423
  } else if (m === 'allocUint16') {
    const a = NTRU.allocUint16(1000); a[999] = 65535; strictEqual(a[999], 65535); strictEqual(a.length, 1000);
  } else throw new Error('unknown first call ' + m);
} else {
  throw new Error('usage: node ref_scenarios.mjs roundtrip|wrongkey|large|homomorphic|firstcall-<method>');
}
console.log(`ref_scenarios: ${scenario} OK`);
