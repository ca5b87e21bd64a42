export NTRU_ALLOW_TIMING_ONLY=1      # the libraries these scripts time compute wrong values on purpose (ntru_engine_create asks)
set -e
mkdir -p gpurun_out
OUT=gpurun_out/r03_power_ablations.txt
: > $OUT
echo "# full build" >> $OUT
timeout -k 10 120 python3 tools/clock_power.py --seconds 3 --no-smi --loads encrypt,decrypt,decrypt_value >> $OUT 2>/dev/null
for a in 1 2 3 262144; do
  echo "# NTRU_ABLATE=$a" >> $OUT
  NTRU_ENGINE_LIB=$PWD/ntru-circom_amd/lib/ab/libntru_abl$a.so timeout -k 10 120 python3 tools/clock_power.py --seconds 3 --no-smi --loads encrypt,decrypt >> $OUT 2>/dev/null
done
python3 - <<'PY'
import json
for line in open("gpurun_out/r03_power_ablations.txt"):
    if line.startswith("#"): print(line.strip()); continue
    d = json.loads(line)
    if "load" in d and "ms_per_launch" in d:
        dr = d.get("driver") or {}
        print("  %-28s %-14s %.3f ms  %.3f GHz  %s W" % (d["load"], d["kernel"], d["ms_per_launch"], d["shader_clock_GHz_from_memtime"], (dr.get("power_W") or {}).get("mean")))
PY
