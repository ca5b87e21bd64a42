// Golden cases for key inversion (SURVEY.md 8f #1): runs the UNMODIFIED reference under Node and records, for seeded
// ternary f, what loadPrivateKeyF (index.js:30-49) produces: fq and fp, or the message of the error it throws.  Small and
// even N are included on purpose: there many f are not units, and the reference's `&&` checks (index.js:41-45, :451)
// accept some of them.  Only the resulting JSON is committed.
//
//   node tests/golden/gen_keygen_cases.mjs [/root/reference] [outdir]
import { writeFileSync } from 'fs';
import { dirname, join } from 'path';
import { fileURLToPath, pathToFileURL } from 'url';

const here = dirname(fileURLToPath(import.meta.url));
const refDir = process.argv[2] || '/root/reference';
const outDir = process.argv[3] || here;
let state = 1;
function nextU32() { let x = state; x ^= x << 13; x >>>= 0; x ^= x >>> 17; x ^= x << 5; x >>>= 0; state = x; return x; }
globalThis.crypto = { getRandomValues(arr) { for (let i = 0; i < arr.length; i++) arr[i] = nextU32(); return arr; } };

async function main() {
const { default: NTRU } = await import(pathToFileURL(join(refDir, 'index.js')).href);

const cases = [];
const sets = [[7, 32, 3, 20], [11, 64, 3, 20], [16, 128, 3, 20], [17, 32, 3, 30], [31, 4096, 3, 12], [33, 8192, 3, 8], [64, 2048, 3, 6],
              [101, 2048, 3, 4], [167, 128, 3, 3]];
for (const [N, q, p, count] of sets) {
  state = (N * 2654435761 + q) >>> 0 || 1;
  for (let t = 0; t < count; t++) {
    // arbitrary ternary f (not only the reference's df / df-1 shape): weights vary, so do f(1) mod 2 and mod 3
    const f = Array.from({ length: N }, () => [0, 1, -1][nextU32() % 3]);
    if (t % 5 === 4) f.fill(0, Math.floor(N / 2));             // low-degree f
    const ntru = new NTRU({ N, q, p });
    const c = { N, q, p, f };
    try {
      ntru.loadPrivateKeyF(f);
      c.fq = ntru.fq; c.fp = ntru.fp;
    } catch (e) {
      c.error = String(e.message);
    }
    cases.push(c);
  }
}
writeFileSync(join(outDir, 'keygen_cases.json'), JSON.stringify({ generator: 'gen_keygen_cases.mjs', cases }));
const ok = cases.filter(c => !c.error).length;
console.log(`keygen_cases.json: ${cases.length} cases, ${ok} accepted, errors:`, [...new Set(cases.filter(c => c.error).map(c => c.error))]);
}
main().catch(e => { console.error(e); process.exit(1); });
