# time / shader clock / socket power of the headline kernels for several builds of the library on ONE device
#   LIBS="new ntru-circom_amd/lib/ab/libntru_base.so ..." LOADS=encrypt,decrypt KPS="0 4" bash tools/ab_power_libs.sh
for lib in ${LIBS:-new ntru-circom_amd/lib/ab/libntru_base.so}; do
 for kp in ${KPS:-0}; do
  echo "# lib=$lib kernel_path=$kp"
  if [ "$lib" != new ]; then export NTRU_ENGINE_LIB=$PWD/$lib; else unset NTRU_ENGINE_LIB; fi
  timeout -k 10 100 python3 tools/clock_power.py --seconds ${SECS:-2.5} --no-smi --loads ${LOADS:-encrypt} --kernel-path $kp 2>/dev/null | python3 -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if "ms_per_launch" in d:
        dr = d.get("driver") or {}
        print("  %-28s %-14s %.3f ms  %.3f GHz  %.0f W" % (d["load"], d["kernel"], d["ms_per_launch"], d["shader_clock_GHz_from_memtime"], (dr.get("power_W") or {}).get("mean", 0)))
'
 done
done
