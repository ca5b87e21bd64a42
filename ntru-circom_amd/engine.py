"""ctypes binding of include/ntru_engine.h (the engine's C ABI).  No computation happens here."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERR_NAMES = {1: "NTRU_ERR_NO_DEVICE", 2: "NTRU_ERR_ARG", 3: "NTRU_ERR_UNSUPPORTED", 4: "NTRU_ERR_HIP"}
FLAG_INVALID_FQ, FLAG_INVALID_FP, FLAG_INVALID_H = 1, 2, 4
FLAG_NOT_UNIT_MOD2, FLAG_NOT_UNIT_MODP = 8, 16
# status codes of the generic family = the errors the reference throws (include/ntru_engine.h NTRU_GENERIC_*)
GENERIC_ERRORS = {1: "Cannot divide by zero polynomial.", 2: "No inverse exists for division.", 3: "invalid_gcd",
                  4: "ntru engine: generic work area exhausted"}


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, "error %d" % code), msg))
        self.code = code


def library_path():
    """The in-tree build; NTRU_ENGINE_LIB selects another build of the same C ABI (A/B measurements)."""
    return os.environ.get("NTRU_ENGINE_LIB") or os.path.join(_HERE, "lib", "libntru_engine.so")


_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
_SIGS = {
    "ntru_engine_device_count": (C.c_int, []),
    "ntru_engine_create": (C.c_int, [_i, C.POINTER(_vp)]),
    "ntru_engine_destroy": (None, [_vp]),
    "ntru_engine_set_stream": (C.c_int, [_vp, _vp]),
    "ntru_engine_synchronize": (C.c_int, [_vp]),
    "ntru_engine_set_kernel_path": (C.c_int, [_vp, _i]),
    "ntru_engine_last_kernel": (C.c_char_p, [_vp]),
    "ntru_last_error": (C.c_char_p, []),
    "ntru_engine_supports": (C.c_int, [_i, _i]),
}
for _sfx in ("", "_dev"):
    _SIGS["ntru_public_key_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _i64, _vp])
    _SIGS["ntru_invert_key_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp])
_SIGS["ntru_engine_set_sampler_rounds"] = (C.c_int, [_vp, _i])
_SIGS["ntru_engine_get_sampler_rounds"] = (C.c_int, [_vp])
_SIGS["ntru_sample_ternary"] = (C.c_int, [_vp, _i, _i, _i, _i, _vp, C.c_uint64, _i64, _vp])
_SIGS["ntru_sample_ternary_dev"] = (C.c_int, [_vp, _i, _i, _i, _i, _vp, C.c_uint64, _i64, _vp])
_ip = C.POINTER(C.c_int)
_SIGS["ntru_pack_params"] = (C.c_int, [_i, _i, _ip, _ip, _ip, _ip])
_SIGS["ntru_decrypt_pack_batch_dev"] = (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp])
_SIGS["ntru_encrypt_pack_batch_dev"] = (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp])
for _sfx in ("", "_dev"):
    _SIGS["ntru_pack_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _vp, _i64, _vp])
    _SIGS["ntru_unpack_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _vp, _i, _i64, _vp])
for _sfx in ("", "_dev"):
    _SIGS["ntru_polymul_split" + _sfx] = (C.c_int, [_vp, _i, _i, _vp, _vp, _i64, _vp, _vp])
    _SIGS["ntru_split_by_I" + _sfx] = (C.c_int, [_vp, _i, _i, _vp, _i64, _vp, _vp])
    _SIGS["ntru_add_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _vp, _vp, _i64, _vp])
    _SIGS["ntru_encrypt_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp])
    _SIGS["ntru_decrypt_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp])
    _SIGS["ntru_verify_keys_batch" + _sfx] = (C.c_int, [_vp, _i, _i, _i] + [_vp] * 5 + [_i64] + [_vp] * 7)
_SIGS["ntru_pack_bytes_batch_dev"] = (C.c_int, [_vp, _i, _i, _vp, _i64, _vp])
_SIGS["ntru_pipeline_batch"] = (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_uint64, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp])
_SIGS["ntru_dev_alloc"] = (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)])
_SIGS["ntru_dev_free"] = (C.c_int, [_vp, _vp])
_SIGS["ntru_dev_upload"] = (C.c_int, [_vp, _vp, _vp, C.c_size_t])
_SIGS["ntru_dev_download"] = (C.c_int, [_vp, _vp, _vp, C.c_size_t])
_SIGS["ntru_host_alloc"] = (C.c_void_p, [C.c_size_t])
_SIGS["ntru_host_free"] = (None, [_vp])
_SIGS["ntru_generic_capacity"] = (C.c_int, [_i, _i])
_SIGS["ntru_generic_multiply"] = (C.c_int, [_vp, _i, _i, _i64, _vp, _vp, _i64, _vp, _vp])
_SIGS["ntru_generic_divide"] = (C.c_int, [_vp, _i, _i, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp])
_SIGS["ntru_generic_eea"] = (C.c_int, [_vp, _i, _i, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp])
_SIGS["ntru_generic_poly_inv"] = (C.c_int, [_vp, _i, _i, _i64, _vp, _vp, _i64, _vp, _vp, _vp])
_SIGS["ntru_multi_create"] = (C.c_int, [_vp, _i, C.POINTER(_vp)])
_SIGS["ntru_multi_destroy"] = (None, [_vp])
_SIGS["ntru_multi_engines"] = (C.c_int, [_vp])
_SIGS["ntru_multi_encrypt_batch"] = (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp])
_SIGS["ntru_multi_decrypt_batch"] = (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp])
_SIGS["ntru_multi_verify_keys_batch"] = (C.c_int, [_vp, _i, _i, _i] + [_vp] * 5 + [_i64] + [_vp] * 7)
_SIGS["ntru_multi_polymul_split"] = (C.c_int, [_vp, _i, _i, _vp, _vp, _i64, _vp, _vp])
_SIGS["ntru_multi_invert_key_batch"] = (C.c_int, [_vp, _i, _i, _i, _vp, _i64, _vp, _vp, _vp])
_SIGS["ntru_multi_public_key_batch"] = (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _i64, _vp])
_SIGS["ntru_encrypt_batch_pitched_dev"] = (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp])
_SIGS["ntru_decrypt_batch_pitched_dev"] = (C.c_int, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp])


def _preload_hip_runtime():
    """Keep ONE HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (same SONAME as
    /opt/rocm's, found through an RPATH under the plain name); if the engine pulled /opt/rocm's copy in first, a
    later `import torch` would load a second runtime that sees no device.  When torch is installed, load its copy
    first: the engine's NEEDED libamdhip64.so.7 then binds to it by SONAME.  Without torch (C, Node.js) the
    engine's RUNPATH finds /opt/rocm as usual."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """dlopen the HIP engine.  Fails loudly when it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise EngineError(0, "HIP engine library %s is missing; run `python -c 'import __graft_entry__ as g; "
                                 "g.build()'` or `make -C ntru-circom_amd/csrc`" % path)
        _preload_hip_runtime()
        lib = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


def _np(a, dt, shape=None):
    arr = np.ascontiguousarray(np.asarray(a, dtype=dt))
    return arr.reshape(shape) if shape is not None else arr


def _ptr(arr):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


class Engine:
    """One engine per HIP device.  Host-buffer methods take/return numpy arrays; *_dev methods take raw
    device pointers (ints, e.g. torch.Tensor.data_ptr()) and only enqueue work on the engine's stream."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        self._h = None
        self._chk(self._lib.ntru_engine_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def _chk(self, rc):
        if rc:
            raise EngineError(rc, self._lib.ntru_last_error().decode())

    def close(self):
        if self._h is not None:
            self._lib.ntru_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- launch log (profiling aid) ----------------------------------------------------------------------------
    # NTRU_LAUNCH_LOG=<path>: every single-kernel *_dev call appends {"kernel", "N", "items", "bytes_per_item"} in launch order;
    # tools/pmc_summary.py matches the k-th rocprofv3 dispatch of a kernel with the k-th record, so that per-launch counters are
    # scaled by the size of THAT launch (bench.py and tools/bench_configs.py launch one kernel at several sizes).
    def _note(self, N, items, bytes_per_item):
        path = os.environ.get("NTRU_LAUNCH_LOG")
        if path:
            with open(path, "a") as fh:
                fh.write('{"kernel": "%s", "N": %d, "items": %d, "bytes_per_item": %d}\n'
                         % (self.last_kernel(), int(N), int(items), int(bytes_per_item)))

    def set_stream(self, hip_stream):
        self._chk(self._lib.ntru_engine_set_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else None)))

    def set_kernel_path(self, path):
        """0 auto, 1 packed-u16 MAC kernels, 2 ternary add path, 3 add path without dot8, 4 int8 matrix-core path
        (each where applicable; same results)."""
        self._chk(self._lib.ntru_engine_set_kernel_path(self._h, int(path)))

    def last_kernel(self):
        return self._lib.ntru_engine_last_kernel(self._h).decode()

    def synchronize(self):
        self._chk(self._lib.ntru_engine_synchronize(self._h))

    def supports(self, N, mod):
        return bool(self._lib.ntru_engine_supports(int(N), int(mod)))

    # ---- host buffers ------------------------------------------------------------------------------------
    def polymul_split(self, N, mod, a, b):
        a, b = _np(a, np.uint16).reshape(-1, N), _np(b, np.uint16).reshape(-1, N)
        B = a.shape[0]
        quot, rem = np.empty((B, N), np.uint16), np.empty((B, N), np.uint16)
        self._chk(self._lib.ntru_polymul_split(self._h, N, mod, _ptr(a), _ptr(b), B, _ptr(quot), _ptr(rem)))
        return quot, rem

    def split_by_I(self, N, mod, a):
        a = _np(a, np.uint16).reshape(-1, 2 * N)
        B = a.shape[0]
        quot, rem = np.empty((B, N), np.uint16), np.empty((B, N), np.uint16)
        self._chk(self._lib.ntru_split_by_I(self._h, N, mod, _ptr(a), B, _ptr(quot), _ptr(rem)))
        return quot, rem

    def add_batch(self, N, mod, a, b):
        a, b = _np(a, np.uint16).reshape(-1, N), _np(b, np.uint16).reshape(-1, N)
        B = a.shape[0]
        out = np.empty((B, N), np.uint16)
        self._chk(self._lib.ntru_add_batch(self._h, N, mod, _ptr(a), _ptr(b), B, _ptr(out)))
        return out

    def pack_params(self, max_val, data_len):
        v = [C.c_int(0) for _ in range(4)]
        self._chk(self._lib.ntru_pack_params(int(max_val), int(data_len), *[C.byref(x) for x in v]))
        return dict(zip(("maxInputBits", "numInputsPerOutput", "arrLen", "outputSize"), (x.value for x in v)))

    def pack_batch(self, max_val, data_len, data):
        """[B][data_len] values -> [B][outputSize][4] little-endian uint64 limbs (index.js:572-596)."""
        data = _np(data, np.uint16).reshape(-1, data_len) if data_len else np.zeros((len(data), 0), np.uint16)
        B = data.shape[0]
        out = np.empty((B, self.pack_params(max_val, data_len)["outputSize"], 4), np.uint64)
        self._chk(self._lib.ntru_pack_batch(self._h, max_val, data_len, _ptr(data), B, _ptr(out)))
        return out

    def unpack_batch(self, max_val, packed_bits, limbs):
        """[B][packedSize][4] limbs -> [B][packedSize * per] values (index.js:598-620, untrimmed)."""
        limbs = _np(limbs, np.uint64)
        B, S = limbs.shape[0], limbs.shape[1]
        per = packed_bits // self.pack_params(max_val, 0)["maxInputBits"]
        out = np.empty((B, S * per), np.uint16)
        self._chk(self._lib.ntru_unpack_batch(self._h, max_val, packed_bits, _ptr(limbs), S, B, _ptr(out)))
        return out

    def set_sampler_rounds(self, rounds):
        """Rounds of the sampler's ChaCha block function: 20 (RFC 8439, default), 12 or 8."""
        self._chk(self._lib.ntru_engine_set_sampler_rounds(self._h, int(rounds)))

    def sampler_rounds(self):
        return int(self._lib.ntru_engine_get_sampler_rounds(self._h))

    def sample_ternary(self, N, n1, n2, other, key, first_item, B):
        """generateCustomArray on the device (ChaCha20 draw stream under `key`, 8 uint32)."""
        key = _np(key, np.uint32, (8,))
        out = np.empty((B, N), np.uint8)
        self._chk(self._lib.ntru_sample_ternary(self._h, N, n1, n2, other, _ptr(key), int(first_item), B, _ptr(out)))
        return out

    def sample_ternary_dev(self, N, n1, n2, other, key, first_item, B, d_out):
        key = _np(key, np.uint32, (8,))
        self._chk(self._lib.ntru_sample_ternary_dev(self._h, N, n1, n2, other, _ptr(key), int(first_item), B,
                                                    self._dp(d_out)))
        self._note(N, B, N)

    def pipeline_batch(self, N, q, p, h, m, f=None, fp=None, key=None, first_item=0, n1=0, n2=0, r=None,
                       want_r=False, want_e=False, want_value=False, want_packed=False):
        """ntru_pipeline_batch: sampler (key) or given r -> encryptBits -> decryptBits (f, fp) -> packOutput, device-resident between
        the stages; host arrays in and out.  Returns a dict of the outputs asked for."""
        h = _np(h, np.uint16, (N,))
        m = _np(m, np.uint8).reshape(-1, N)
        B = m.shape[0]
        f = None if f is None else _np(f, np.int8, (N,))
        fp = None if fp is None else _np(fp, np.uint8, (N,))
        key = None if key is None else _np(key, np.uint32, (8,))
        r = None if r is None else _np(r, np.uint8).reshape(-1, N)
        out = {}
        if want_r: out["r"] = np.empty((B, N), np.uint8)
        if want_e: out["e"] = np.empty((B, N), np.uint16)
        if want_value: out["value"] = np.empty((B, N), np.uint8)
        if want_packed:
            osz = self.pack_params(p - 1 if f is not None else q - 1, N)["outputSize"]
            out["packed"] = np.empty((B, osz, 4), np.uint64)
        self._chk(self._lib.ntru_pipeline_batch(self._h, N, q, p, _ptr(h), _ptr(f), _ptr(fp), _ptr(key), int(first_item), int(n1), int(n2),
                                                _ptr(r), _ptr(m), B, _ptr(out.get("r")), _ptr(out.get("e")), _ptr(out.get("value")),
                                                _ptr(out.get("packed"))))
        return out

    # ---- plain device buffers (for callers without torch: what the N-API addon hands to JavaScript) ----------------
    def dev_alloc(self, nbytes):
        p = _vp()
        self._chk(self._lib.ntru_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, d_ptr):
        self._chk(self._lib.ntru_dev_free(self._h, self._dp(d_ptr)))

    def dev_upload(self, d_ptr, host):
        host = np.ascontiguousarray(host)
        self._chk(self._lib.ntru_dev_upload(self._h, self._dp(d_ptr), _ptr(host), host.nbytes))

    def dev_download(self, d_ptr, shape, dtype):
        out = np.empty(shape, dtype)
        self._chk(self._lib.ntru_dev_download(self._h, _ptr(out), self._dp(d_ptr), out.nbytes))
        return out

    def pack_bytes_batch_dev(self, max_val, data_len, d_data, B, d_out):
        self._chk(self._lib.ntru_pack_bytes_batch_dev(self._h, int(max_val), int(data_len), self._dp(d_data), B, self._dp(d_out)))

    def encrypt_batch(self, N, q, h, r, m, want_quot=True):
        h = _np(h, np.uint16, (N,))
        r, m = _np(r, np.uint8).reshape(-1, N), _np(m, np.uint8).reshape(-1, N)
        B = r.shape[0]
        e = np.empty((B, N), np.uint16)
        quot = np.empty((B, N), np.uint16) if want_quot else None
        self._chk(self._lib.ntru_encrypt_batch(self._h, N, q, _ptr(h), _ptr(r), _ptr(m), B, _ptr(e), _ptr(quot)))
        return e, quot

    def decrypt_batch(self, N, q, p, f, fp, e, want_witness=True):
        f, fp = _np(f, np.int8, (N,)), _np(fp, np.uint8, (N,))
        e = _np(e, np.uint16).reshape(-1, N)
        B = e.shape[0]
        value = np.empty((B, N), np.uint8)
        q1 = np.empty((B, N), np.uint16) if want_witness else None
        r1 = np.empty((B, N), np.uint16) if want_witness else None
        q2 = np.empty((B, N), np.uint8) if want_witness else None
        self._chk(self._lib.ntru_decrypt_batch(self._h, N, q, p, _ptr(f), _ptr(fp), _ptr(e), B, _ptr(value),
                                               _ptr(q1), _ptr(r1), _ptr(q2)))
        return value, q1, r1, q2

    def verify_keys_batch(self, N, q, p, f, g, fq, fp, h):
        f, g = _np(f, np.int8).reshape(-1, N), _np(g, np.int8).reshape(-1, N)
        fq, h = _np(fq, np.uint16).reshape(-1, N), _np(h, np.uint16).reshape(-1, N)
        fp = _np(fp, np.uint8).reshape(-1, N)
        B = f.shape[0]
        out = {"quot_fq": np.empty((B, N), np.uint16), "rem_fq": np.empty((B, N), np.uint16),
               "quot_fp": np.empty((B, N), np.uint8), "rem_fp": np.empty((B, N), np.uint8),
               "quot_h": np.empty((B, N), np.uint16), "rem_h": np.empty((B, N), np.uint16),
               "flags": np.empty(B, np.uint8)}
        self._chk(self._lib.ntru_verify_keys_batch(self._h, N, q, p, _ptr(f), _ptr(g), _ptr(fq), _ptr(fp), _ptr(h), B,
                                                   _ptr(out["quot_fq"]), _ptr(out["rem_fq"]), _ptr(out["quot_fp"]),
                                                   _ptr(out["rem_fp"]), _ptr(out["quot_h"]), _ptr(out["rem_h"]),
                                                   _ptr(out["flags"])))
        return out

    def public_key_batch(self, N, q, p, fq, g):
        """generatePublicKeyH for B keys: rows of (p*fq mod q) * g mod (x^N - 1, q), untrimmed."""
        fq, g = _np(fq, np.uint16).reshape(-1, N), _np(g, np.int8).reshape(-1, N)
        B = fq.shape[0]
        h = np.empty((B, N), np.uint16)
        self._chk(self._lib.ntru_public_key_batch(self._h, N, q, p, _ptr(fq), _ptr(g), B, _ptr(h)))
        return h

    def invert_key_batch(self, N, q, p, f, want_fq=True, want_fp=True):
        """loadPrivateKeyF / polyInv for B keys: (fq [B][N] u16, fp [B][N] u8, flags [B]); a set flag = not a unit.
        want_fq / want_fp = False skips that half (its array comes back as None)."""
        f = _np(f, np.int8).reshape(-1, N)
        B = f.shape[0]
        fq = np.empty((B, N), np.uint16) if want_fq else None
        fp = np.empty((B, N), np.uint8) if want_fp else None
        flags = np.empty(B, np.uint8)
        self._chk(self._lib.ntru_invert_key_batch(self._h, N, q, p, _ptr(f), B, _ptr(fq), _ptr(fp), _ptr(flags)))
        return fq, fp, flags

    # ---- pinned host memory ---------------------------------------------------------------------------------
    def pinned_empty(self, shape, dtype):
        """numpy array over page-locked memory (ntru_host_alloc): the batch entry points DMA it in place.  The memory is
        released when the array (and every view of it) is gone."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = self._lib.ntru_host_alloc(n)
        if not p:
            raise EngineError(4, self._lib.ntru_last_error().decode())
        lib = self._lib

        class _Owner:
            def __init__(self, ptr): self.ptr = ptr
            def __del__(self): lib.ntru_host_free(self.ptr)
        buf = (C.c_char * max(n, 1)).from_address(p)
        buf._owner = _Owner(p)
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    # ---- generic family (ntru_generic_*): int64 coefficients, uniform operand lengths per batch ---------------
    def _generic(self, op, a, b, mod, two):
        a, b = np.ascontiguousarray(a, dtype=np.int64), np.ascontiguousarray(b, dtype=np.int64)
        if a.ndim == 1: a = a[None]
        if b.ndim == 1: b = b[None]
        B, la, lb = a.shape[0], a.shape[1], b.shape[1]
        cap = self._lib.ntru_generic_capacity(la, lb)
        o0, l0 = np.zeros((B, cap), np.int64), np.zeros(B, np.int32)
        o1, l1 = (np.zeros((B, cap), np.int64), np.zeros(B, np.int32)) if two else (None, None)
        st = np.zeros(B, np.uint8)
        fn = [self._lib.ntru_generic_multiply, self._lib.ntru_generic_divide, self._lib.ntru_generic_eea,
              self._lib.ntru_generic_poly_inv][op]
        if op == 0:
            rc = fn(self._h, la, lb, int(mod), _ptr(a), _ptr(b), B, _ptr(o0), _ptr(l0))
        elif op == 3:
            rc = fn(self._h, la, lb, int(mod), _ptr(a), _ptr(b), B, _ptr(o0), _ptr(l0), _ptr(st))
        else:
            rc = fn(self._h, la, lb, int(mod), _ptr(a), _ptr(b), B, _ptr(o0), _ptr(l0), _ptr(o1), _ptr(l1), _ptr(st))
        self._chk(rc)
        rows = lambda o, l: [o[i, :l[i]].tolist() for i in range(B)]
        return rows(o0, l0), (rows(o1, l1) if two else None), st

    def generic_multiply(self, a, b, mod):
        """multiplyPolynomials per row (index.js:319-355), any modulus up to 2^26 -> list of trimmed lists."""
        return self._generic(0, a, b, mod, False)[0]

    def generic_divide(self, a, b, mod):
        """dividePolynomials per row (index.js:358-401) -> (quotients, remainders, status[B])."""
        return self._generic(1, a, b, mod, True)

    def generic_eea(self, a, b, mod):
        """extendedEuclideanAlgorithm per row (index.js:425-459) -> (gcds, inverses, status[B])."""
        return self._generic(2, a, b, mod, True)

    def generic_poly_inv(self, a, poly_i, mod):
        """polyInv per row (index.js:491-514) -> (inverses, status[B])."""
        r = self._generic(3, a, poly_i, mod, False)
        return r[0], r[2]

    def invert_key_batch_dev(self, N, q, p, d_f, B, d_fq, d_fp, d_flags):
        dp = self._dp
        self._chk(self._lib.ntru_invert_key_batch_dev(self._h, N, q, p, dp(d_f), B, dp(d_fq), dp(d_fp), dp(d_flags)))

    def decrypt_pack_batch_dev(self, N, q, p, d_f, d_fp, d_e, B, d_value, d_packed):
        """decryptBits + packOutput(p - 1, N, value) (one kernel on the matrix path); d_value may be None there."""
        dp = self._dp
        self._chk(self._lib.ntru_decrypt_pack_batch_dev(self._h, N, q, p, dp(d_f), dp(d_fp), dp(d_e), B, dp(d_value), dp(d_packed)))
        self._note(N, B, 2 * N + 32 * max(3, -(-N // 126)) + (N if d_value else 0))      # e in; packed rows (+ the plain values) out

    def encrypt_pack_batch_dev(self, N, q, d_h, d_r, d_m, B, d_e, d_packed):
        """encryptBits + packOutput(q - 1, N, e) (one kernel with d_e None where the row-image matrix kernel applies)."""
        dp = self._dp
        self._chk(self._lib.ntru_encrypt_pack_batch_dev(self._h, N, q, dp(d_h), dp(d_r), dp(d_m), B, dp(d_e), dp(d_packed)))
        bits = max(1, (q - 1).bit_length())
        self._note(N, B, 2 * N + 32 * max(3, -(-N // (252 // bits))) + (2 * N if d_e else 0))   # r, m in; packed rows (+ e) out

    def public_key_batch_dev(self, N, q, p, d_fq, d_g, B, d_h):
        dp = self._dp
        self._chk(self._lib.ntru_public_key_batch_dev(self._h, N, q, p, dp(d_fq), dp(d_g), B, dp(d_h)))
        self._note(N, B, 5 * N)

    # ---- device pointers (asynchronous) -------------------------------------------------------------------
    @staticmethod
    def _dp(x):
        return C.c_void_p(int(x)) if x else None

    def polymul_split_dev(self, N, mod, d_a, d_b, B, d_quot, d_rem):
        dp = self._dp
        self._chk(self._lib.ntru_polymul_split_dev(self._h, N, mod, dp(d_a), dp(d_b), B, dp(d_quot), dp(d_rem)))
        self._note(N, B, 8 * N)

    def split_by_I_dev(self, N, mod, d_a, B, d_quot, d_rem):
        dp = self._dp
        self._chk(self._lib.ntru_split_by_I_dev(self._h, N, mod, dp(d_a), B, dp(d_quot), dp(d_rem)))

    def add_batch_dev(self, N, mod, d_a, d_b, B, d_out):
        dp = self._dp
        self._chk(self._lib.ntru_add_batch_dev(self._h, N, mod, dp(d_a), dp(d_b), B, dp(d_out)))

    def encrypt_batch_dev(self, N, q, d_h, d_r, d_m, B, d_e, d_quotE=None, ld=None):
        """ld: row pitch of r, m, e, quotE in elements (None = dense rows of N; see ntru_encrypt_batch_pitched_dev)."""
        dp = self._dp
        if ld is None:
            self._chk(self._lib.ntru_encrypt_batch_dev(self._h, N, q, dp(d_h), dp(d_r), dp(d_m), B, dp(d_e), dp(d_quotE)))
        else:
            self._chk(self._lib.ntru_encrypt_batch_pitched_dev(self._h, N, q, int(ld), dp(d_h), dp(d_r), dp(d_m), B,
                                                               dp(d_e), dp(d_quotE)))
        self._note(N, B, (6 if d_quotE else 4) * N)             # r, m in; e (+ quotientE) out

    def decrypt_batch_dev(self, N, q, p, d_f, d_fp, d_e, B, d_value, d_quot1=None, d_rem1=None, d_quot2=None, ld=None):
        """ld: row pitch of e, value, quot1, rem1, quot2 in elements (None = dense rows of N)."""
        dp = self._dp
        if ld is None:
            self._chk(self._lib.ntru_decrypt_batch_dev(self._h, N, q, p, dp(d_f), dp(d_fp), dp(d_e), B, dp(d_value),
                                                       dp(d_quot1), dp(d_rem1), dp(d_quot2)))
        else:
            self._chk(self._lib.ntru_decrypt_batch_pitched_dev(self._h, N, q, p, int(ld), dp(d_f), dp(d_fp), dp(d_e), B,
                                                               dp(d_value), dp(d_quot1), dp(d_rem1), dp(d_quot2)))
        self._note(N, B, (3 + (2 if d_quot1 else 0) + (2 if d_rem1 else 0) + (1 if d_quot2 else 0)) * N)

    def verify_keys_batch_dev(self, N, q, p, d_f, d_g, d_fq, d_fp, d_h, B, d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp,
                              d_quot_h, d_rem_h, d_flags):
        dp = self._dp
        self._chk(self._lib.ntru_verify_keys_batch_dev(self._h, N, q, p, dp(d_f), dp(d_g), dp(d_fq), dp(d_fp), dp(d_h),
                                                       B, dp(d_quot_fq), dp(d_rem_fq), dp(d_quot_fp), dp(d_rem_fp),
                                                       dp(d_quot_h), dp(d_rem_h), dp(d_flags)))
        self._note(N, B, 17 * N)


class MultiEngine:
    """Several devices in one process (ntru_multi_*): a host batch is cut into contiguous shards, one engine + host thread per
    listed device id (an id may repeat).  Host numpy arrays in and out, same results as Engine."""

    def __init__(self, device_ids):
        self._lib = load_library()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        self._h = None
        rc = self._lib.ntru_multi_create(ids, len(device_ids), C.byref(h))
        if rc:
            raise EngineError(rc, self._lib.ntru_last_error().decode())
        self._h = h

    def _chk(self, rc):
        if rc:
            raise EngineError(rc, self._lib.ntru_last_error().decode())

    def close(self):
        if self._h is not None:
            self._lib.ntru_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def engines(self):
        return int(self._lib.ntru_multi_engines(self._h))

    def encrypt_batch(self, N, q, h, r, m, want_quot=True):
        h = _np(h, np.uint16, (N,))
        r, m = _np(r, np.uint8).reshape(-1, N), _np(m, np.uint8).reshape(-1, N)
        B = r.shape[0]
        e = np.empty((B, N), np.uint16)
        quot = np.empty((B, N), np.uint16) if want_quot else None
        self._chk(self._lib.ntru_multi_encrypt_batch(self._h, N, q, _ptr(h), _ptr(r), _ptr(m), B, _ptr(e), _ptr(quot)))
        return e, quot

    def decrypt_batch(self, N, q, p, f, fp, e, want_witness=True):
        f, fp = _np(f, np.int8, (N,)), _np(fp, np.uint8, (N,))
        e = _np(e, np.uint16).reshape(-1, N)
        B = e.shape[0]
        value = np.empty((B, N), np.uint8)
        q1 = np.empty((B, N), np.uint16) if want_witness else None
        r1 = np.empty((B, N), np.uint16) if want_witness else None
        q2 = np.empty((B, N), np.uint8) if want_witness else None
        self._chk(self._lib.ntru_multi_decrypt_batch(self._h, N, q, p, _ptr(f), _ptr(fp), _ptr(e), B, _ptr(value),
                                                     _ptr(q1), _ptr(r1), _ptr(q2)))
        return value, q1, r1, q2

    def verify_keys_batch(self, N, q, p, f, g, fq, fp, h):
        f, g = _np(f, np.int8).reshape(-1, N), _np(g, np.int8).reshape(-1, N)
        fq, h = _np(fq, np.uint16).reshape(-1, N), _np(h, np.uint16).reshape(-1, N)
        fp = _np(fp, np.uint8).reshape(-1, N)
        B = f.shape[0]
        out = {"quot_fq": np.empty((B, N), np.uint16), "rem_fq": np.empty((B, N), np.uint16),
               "quot_fp": np.empty((B, N), np.uint8), "rem_fp": np.empty((B, N), np.uint8),
               "quot_h": np.empty((B, N), np.uint16), "rem_h": np.empty((B, N), np.uint16),
               "flags": np.empty(B, np.uint8)}
        self._chk(self._lib.ntru_multi_verify_keys_batch(self._h, N, q, p, _ptr(f), _ptr(g), _ptr(fq), _ptr(fp), _ptr(h), B,
                                                         _ptr(out["quot_fq"]), _ptr(out["rem_fq"]), _ptr(out["quot_fp"]),
                                                         _ptr(out["rem_fp"]), _ptr(out["quot_h"]), _ptr(out["rem_h"]),
                                                         _ptr(out["flags"])))
        return out

    def polymul_split(self, N, mod, a, b):
        a, b = _np(a, np.uint16).reshape(-1, N), _np(b, np.uint16).reshape(-1, N)
        B = a.shape[0]
        quot, rem = np.empty((B, N), np.uint16), np.empty((B, N), np.uint16)
        self._chk(self._lib.ntru_multi_polymul_split(self._h, N, mod, _ptr(a), _ptr(b), B, _ptr(quot), _ptr(rem)))
        return quot, rem

    def invert_key_batch(self, N, q, p, f):
        f = _np(f, np.int8).reshape(-1, N)
        B = f.shape[0]
        fq, fp, flags = np.empty((B, N), np.uint16), np.empty((B, N), np.uint8), np.empty(B, np.uint8)
        self._chk(self._lib.ntru_multi_invert_key_batch(self._h, N, q, p, _ptr(f), B, _ptr(fq), _ptr(fp), _ptr(flags)))
        return fq, fp, flags

    def public_key_batch(self, N, q, p, fq, g):
        fq, g = _np(fq, np.uint16).reshape(-1, N), _np(g, np.int8).reshape(-1, N)
        B = fq.shape[0]
        h = np.empty((B, N), np.uint16)
        self._chk(self._lib.ntru_multi_public_key_batch(self._h, N, q, p, _ptr(fq), _ptr(g), B, _ptr(h)))
        return h
