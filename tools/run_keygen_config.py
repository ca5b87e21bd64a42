import sys, json
sys.argv=['x']
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import importlib.util
spec = importlib.util.spec_from_file_location('bc','tools/bench_configs.py'); bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
print(json.dumps(bc.keygen_config('n821_q4096', 18)), flush=True)
