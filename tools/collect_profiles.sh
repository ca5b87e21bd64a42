#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ in ONE call on the GPU box (each counter set in its own pass:
# --pmc is never combined with anything but --kernel-trace / --stats):
#   tools/collect_profiles.sh            -> gpurun_out/prof_{stats,fetch,write,sq1,sq2}/ (bench.py, the headline kernels)
#                                           gpurun_out/prof_sec_{stats,fetch,write,sq1,sq2}/ (tools/bench_configs.py: verify_keys,
#                                           polymul, key generation, sampler, the other encrypt configs)
# then, back in the build container:  python tools/pmc_summary.py <tag>  and  python tools/sq_summary.py <tag>
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --power-seconds 0"
S="python3 tools/bench_configs.py --no-pipeline"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
SQ2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
# NTRU_LAUNCH_LOG: the engine's Python binding appends (kernel, N, items, bytes per item) of every launch, in order, so that
# tools/pmc_summary.py can scale each dispatch by its own size
run() { d=$1; shift; rm -rf gpurun_out/$d gpurun_out/${d}_launches.jsonl; export NTRU_LAUNCH_LOG=$PWD/gpurun_out/${d}_launches.jsonl; echo "== $d"; rocprofv3 "$@" > gpurun_out/$d.log 2>&1 || { tail -5 gpurun_out/$d.log; exit 1; }; }
run prof_stats --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $B
run prof_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- $B
run prof_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- $B
run prof_sq1 --pmc $SQ1 --kernel-trace --output-format csv -d gpurun_out/prof_sq1 -- $B
run prof_sq2 --pmc $SQ2 --kernel-trace --output-format csv -d gpurun_out/prof_sq2 -- $B
run prof_sec_stats --kernel-trace --stats --output-format csv -d gpurun_out/prof_sec_stats -- $S
run prof_sec_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_sec_fetch -- $S
run prof_sec_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_sec_write -- $S
run prof_sec_sq1 --pmc $SQ1 --kernel-trace --output-format csv -d gpurun_out/prof_sec_sq1 -- $S
run prof_sec_sq2 --pmc $SQ2 --kernel-trace --output-format csv -d gpurun_out/prof_sec_sq2 -- $S
# keep what is merged back small: the counter CSVs of torch's own kernels are not needed
find gpurun_out/prof_* -name "*_agent_info.csv" -delete 2>/dev/null || true
du -sh gpurun_out/prof_* | tail -12
