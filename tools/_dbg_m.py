import sys, json, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import ntru_oracle as orc
eng = pkg.Engine(0)
g = json.load(open('tests/golden/scheme_n167_q128.json'))
opts = g["options"]
for key in g["keys"]:
    n = pkg.NTRU(dict(opts, f=key["f"], fp=key["fp"], fq=key["fq"], g=key["g"], h=key["h"]), engine=eng)
    for case in key["cases"]:
        it = iter(case["draws"])
        r = pkg.generateCustomArray(opts["N"], opts["dr"], opts["dr"], rand_u32=lambda: next(it))
        enc = n.encryptBits(list(case["m"]), r=r)
        w = case["encrypt"]
        for k in ("quotientE", "remainderE"):
            a, b = np.array(enc["inputs"][k]), np.array(w["inputs"][k])
            bad = np.argwhere(a != b).ravel()
            if len(bad): print(k, eng.last_kernel(), "mismatch idx", bad[:10].tolist(), "got", a[bad[:5]].tolist(), "want", b[bad[:5]].tolist(), "len m", len(case["m"]), "max m", max(case["m"]))

        dec = n.decryptBits(enc["value"])
        wd = case["decrypt"]
        if dec != wd:
            for k in wd["inputs"]:
                a, b = np.array(dec["inputs"][k]), np.array(wd["inputs"][k])
                if a.shape != b.shape or (a != b).any():
                    bad = np.argwhere(a != b).ravel() if a.shape == b.shape else []
                    print("decrypt", k, eng.last_kernel(), "len value", len(enc["value"]), "bad idx", bad[:10].tolist() if len(bad) else "shape", "got", a[bad[:5]].tolist() if len(bad) else a.shape, "want", b[bad[:5]].tolist() if len(bad) else b.shape)
            print("value equal", dec["value"] == wd["value"])
print("done")
