import sys, importlib
sys.argv = ['bench.py','--no-cpu-baseline','--steps','2','--warmup','1']
sys.path.insert(0,'/root/repo')
import runpy
try:
    runpy.run_path('/root/repo/bench.py', run_name='__main__')
finally:
    pkg = importlib.import_module('alevin-fry_amd')
    lib = pkg.load_library()
    lib.afq_debug_dump_decode()
