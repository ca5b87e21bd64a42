# J per GB of the store patterns of bench_micro/store_pattern (and the fp4 matrix instruction's rate / power)
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r03_power_store_patterns.txt
: > $OUT
for pat in ${PATS:-0 1 3 4 5 6 7}; do
  timeout -k 10 60 python3 tools/power_sample.py -- bench_micro/store_pattern 821 $pat 3 >> $OUT
done
[ -n "$SKIP_MFMA" ] || timeout -k 10 120 python3 tools/clock_power.py --seconds 3 --no-smi --loads mfma,idle >> $OUT 2>/dev/null
cat $OUT | cut -c1-700
