"""MI355X-native NTRU polynomial-ring engine: host-side mirror of the reference's hot-path interface.

Layout
  csrc/abi.hip           engine life cycle + the *_dev entry points of the C ABI declared in include/ntru_engine.h
  csrc/valu_families.hip, matrix_encrypt.hip, matrix_decrypt.hip, matrix_peritem.hip, keygen_sampler_pack.hip
                         the HIP kernels (gfx950), one translation unit per kernel family (csrc/engine_internal.h lists them)
  csrc/ntru_host.hip     host-pointer entry points: pinned staging, three stage streams, chunked H2D / kernel / D2H pipeline
  csrc/ntru_generic.hip  reference-faithful generic family (arbitrary divisors, moduli up to 2^26, EEA, polyInv)
  lib/libntru_engine.so  built artefact (make -C csrc, or __graft_entry__.build())
  engine.py              ctypes binding of the C ABI (numpy host buffers or raw device pointers)
  ntru.py                `NTRU` class + pure functions with the reference's names and semantics (index.js)
  js/                    N-API addon + ES-module shim exposing the same surface to Node.js

There is no CPU implementation in this package: everything that computes goes through the HIP library and
raises `EngineError` when it (or a GPU) is missing.
"""
from .engine import Engine, EngineError, MultiEngine, library_path, load_library  # noqa: F401
from . import sharding  # noqa: F401
from .ntru import (NTRU, addCiphertexts, addPolynomials, bigintToBits, bitsToBigInt, bitsToString, degree,  # noqa: F401
                   dividePolynomials, expandArray, extendedEuclideanAlgorithm, generateCustomArray, modInverse,
                   multiplyPolynomials, multiplyPolynomialsByScalar, packOutput, polyInv, stringToBits,
                   subtractPolynomials, trimPolynomial, unpackInput)
