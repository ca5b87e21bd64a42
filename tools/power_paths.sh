# time / shader clock / socket power of the encrypt and decrypt kernels per kernel path (same device, one call)
mkdir -p gpurun_out
OUT=gpurun_out/r03_power_kernel_paths.txt
: > $OUT
for kp in ${PATHS:-0 4 7 6 5}; do
  echo "# kernel path $kp" >> $OUT
  timeout -k 10 100 python3 tools/clock_power.py --seconds 2.5 --no-smi --loads encrypt,decrypt --kernel-path $kp 2>/dev/null | python3 -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if "ms_per_launch" in d:
        dr = d.get("driver") or {}
        print("  %-28s %-14s %.3f ms  %.3f GHz  %.0f W" % (d["load"], d["kernel"], d["ms_per_launch"], d["shader_clock_GHz_from_memtime"], (dr.get("power_W") or {}).get("mean", 0)))
' >> $OUT
done
cat $OUT
