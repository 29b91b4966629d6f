import os, sys, runpy
sys.path.insert(0, '/root/repo')
from vulkanhybridrenderer_amd import lib
if os.environ.get("VHR_LIB_VARIANT"): lib.LIB_PATH = os.path.abspath(os.environ["VHR_LIB_VARIANT"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path('/root/repo/bench.py', run_name='__main__')
