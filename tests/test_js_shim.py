"""The Node.js side of the boundary: N-API addon + ES-module shim (ntru-circom_amd/js)."""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as ge

NODE = shutil.which("node")
JS = os.path.join(ge.PKG_DIR, "js")


def _node(args, **kw):
    return subprocess.run([NODE] + args, cwd=ge.ROOT, capture_output=True, text=True, timeout=600, **kw)


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_addon_builds_loads_and_fails_loudly_without_gpu():
    ge.build()
    assert os.path.exists(os.path.join(JS, "ntru_addon.node"))
    code = ("import NTRU, * as lib from '../../ntru-circom_amd/js/index.mjs';"
            "const names=['degree','trimPolynomial','modInverse','addPolynomials','subtractPolynomials','multiplyPolynomials',"
            "'dividePolynomials','multiplyPolynomialsByScalar','extendedEuclideanAlgorithm','generateCustomArray','polyInv',"
            "'expandArrayToMultiple','expandArray','stringToBits','bitsToString','bigintToBits','bitsToBigInt','packOutput',"
            "'unpackInput'];"     # the reference's 19 named exports (index.js:210-598)
            "for (const n of names) if (typeof lib[n] !== 'function') throw new Error('missing export '+n);"
            "const n = new NTRU({N:17,q:32,dr:2,h:[1,2,3]});"
            "if (n.I.length !== 18 || n.I[17] !== -1 || n.calculateNq() !== 15) throw new Error('ctor');"
            "console.log('devices', lib.deviceCount());"
            "if (lib.deviceCount() === 0) { try { n.encryptBits([1,0,1]); throw new Error('no throw'); }"
            " catch (e) { if (!/no CPU fallback/.test(e.message)) throw e; console.log('loud failure ok'); } }")
    path = os.path.join(ge.ROOT, "tests", "js", "_probe.mjs")
    with open(path, "w") as fh:
        fh.write(code)
    try:
        r = _node([path])
    finally:
        os.remove(path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "devices" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_shim_reproduces_reference_witnesses():
    ge.build()
    r = _node([os.path.join(ge.ROOT, "tests", "js", "shim_golden.mjs")])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "shim_golden:" in r.stdout


# test/reference.test.js:6-61: each scenario in a FRESH node process, key generation as its first engine call
@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
@pytest.mark.parametrize("scenario", ["roundtrip", "wrongkey", "large", "homomorphic", "firstcall-loadPrivateKeyF",
                                      "firstcall-generatePublicKeyH", "firstcall-polyInv", "firstcall-allocUint16"])
def test_reference_scenarios_in_fresh_processes(scenario):
    ge.build()
    r = _node([os.path.join(ge.ROOT, "tests", "js", "ref_scenarios.mjs"), scenario])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ref_scenarios: %s OK" % scenario in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_generic_exports_reproduce_reference():
    """polyInv, extendedEuclideanAlgorithm, generic dividePolynomials, products modulo 2^20, the O(N) helpers and
    loadPrivateKeyF on non-units against vectors captured from the reference (generic_functions.json, keygen_cases.json)."""
    ge.build()
    r = _node([os.path.join(ge.ROOT, "tests", "js", "shim_generic.mjs")])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "shim_generic:" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed")
@pytest.mark.parametrize("profile,B,rounds", [("n167_q128", 9001, 20), ("n821_q4096", 2050, 20), ("n821_q4096", 300, 12)])
def test_device_resident_pipeline_through_the_shim_equals_oracle_replay(profile, B, rounds, tmp_path):
    """ntru.pipeline({sampleR, decrypt, pack}) keeps r, e and value on the GPU between the stages (index.js:461-488 -> :87-110 ->
    :111-140 -> :572-596); the CPU oracle replays the ChaCha20 draw stream and every stage from m alone.  B = 9001 runs as four
    chunks through the three-stage pipeline; the script also composes the stages by hand on device-buffer handles."""
    import json
    import numpy as np
    from oracle import ntru_oracle as orc
    from conftest import load_golden
    ge.build()
    r = _node([os.path.join(ge.ROOT, "tests", "js", "shim_pipeline.mjs"), profile, str(B), str(tmp_path)],
              env=dict(os.environ, NTRU_SAMPLER_ROUNDS=str(rounds)))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    meta = json.load(open(tmp_path / "meta.json"))
    assert meta["samplerRounds"] == rounds                  # NTRU.samplerRounds: ChaCha20 by default, 12 / 8 on request
    N, q, p, dr = meta["N"], meta["q"], meta["p"], meta["dr"]
    rd = lambda name, dt: np.fromfile(tmp_path / (name + ".bin"), dtype=dt)
    m = rd("m", np.uint8).reshape(B, N)
    key = load_golden("scheme_%s.json" % profile)["keys"][0]
    pad = lambda a, dt: np.array(list(a) + [0] * (N - len(a)), dtype=dt)
    h, f, fp = pad(key["h"], np.uint16), pad(key["f"], np.int8), pad(key["fp"], np.uint8)
    r_o = orc.sample_ternary_batch(N, dr, dr, p - 1, np.array(meta["key"], np.uint32), meta["firstItem"], B, rounds=rounds)
    assert np.array_equal(rd("r", np.uint8).reshape(B, N), r_o)
    e_o, _ = orc.encrypt_batch(N, q, h, r_o, m)
    assert np.array_equal(rd("e", np.uint16).reshape(B, N), e_o)
    v_o = orc.decrypt_batch(N, q, p, f, fp, e_o)[0]
    assert np.array_equal(rd("value", np.uint8).reshape(B, N), v_o)
    assert np.array_equal(rd("packed_value", np.uint64).reshape(B, meta["outputSizeValue"], 4), orc.pack_batch(p - 1, N, v_o.astype(np.uint16)))
    assert np.array_equal(rd("packed_e", np.uint64).reshape(B, meta["outputSizeE"], 4), orc.pack_batch(q - 1, N, e_o))
