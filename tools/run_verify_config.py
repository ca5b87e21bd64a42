import sys, json, os
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
sys.argv=['x']
import importlib.util
spec = importlib.util.spec_from_file_location('bc','tools/bench_configs.py'); bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
r = bc.verify_config('n821_q4096', 18)
print(os.environ.get('NTRU_ENGINE_LIB','new'), round(r['ms'],4), r['rows_equal_oracle'])
