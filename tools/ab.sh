#!/bin/bash
# Same-device A/B of engine builds (boxes differ by ~5 %, so only runs inside ONE gpurun call compare):
#   tools/ab.sh [bench.py args]     alternates ntru-circom_amd/lib/ab/libntru_base.so and the in-tree library, 3 rounds
# Build the baseline here first, e.g. from a commit:  git worktree add /tmp/base <ref> && make -C /tmp/base/ntru-circom_amd/csrc
#   && cp /tmp/base/ntru-circom_amd/lib/libntru_engine.so ntru-circom_amd/lib/ab/libntru_base.so
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export NTRU_ENGINE_LIB=$PWD/ntru-circom_amd/lib/ab/libntru_base.so; else unset NTRU_ENGINE_LIB; fi
    python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("'$which'", {k: round(v, 4) for k, v in d["kernels_ms"].items()}, round(d["value"] / 1e6, 1), "M rt/s", "verify_keys ms/2^18:", round(d.get("verify_keys", {}).get("ms", 0), 4))'
  done
done
