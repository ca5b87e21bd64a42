#!/bin/bash
# Timing-only ablations of the matrix-core kernels (never shipped, results unchecked):
#   bit 1 = result stores disabled, 2 = matrix loops disabled, 4 = r / e not loaded, 8 = encrypt: m not loaded, 32 = no image
#   expansion, 64 = no mod-3 table, 128 / 256 = product 1 / 2 epilogue skipped, 512 = stores to one L2-resident row block,
#   262144 = no operand reads inside the matrix loops;
#   ABL_SET="3 7 11 15" selects the combinations (default 1 2 3).
# Build here:   tools/ablate.sh build      -> ntru-circom_amd/lib/ab/libntru_abl{1,2,3}.so
# Run on a GPU: tools/ablate.sh run [bench.py args]
export NTRU_ALLOW_TIMING_ONLY=1      # the libraries these scripts time compute wrong values on purpose (ntru_engine_create asks)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p ntru-circom_amd/lib/ab
  for a in ${ABL_SET:-1 2 3}; do
    make -s -C ntru-circom_amd/csrc EXTRA=-DNTRU_ABLATE=$a OBJDIR=../lib/ab/obj_abl$a OUT=../lib/ab/libntru_abl$a.so
  done
else
  shift || true
  python bench.py --no-cpu-baseline "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("full      ", d["kernels_ms"])'
  for a in ${ABL_SET:-1 2 3}; do
    NTRU_BENCH_ABLATION=1 NTRU_ENGINE_LIB=$PWD/ntru-circom_amd/lib/ab/libntru_abl$a.so python bench.py --no-cpu-baseline "$@" \
      | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ablation '$a'", d["kernels_ms"], d["results_match_oracle"])'
  done
fi
