"""Cold start: time to build the cached tables of (n, m) in a fresh process, and the first client after it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 8192
api.bp_gens_prepare(8, 16)      # runtime / module load out of the way
t = time.perf_counter(); api.bp_gens_prepare(n, m); t1 = time.perf_counter()
print("tables (n=%d, m=%d): %.1f ms, %.2f GB" % (n, m, (t1 - t) * 1e3, api.bp_gens_table_bytes(n, m) / 1e9))
