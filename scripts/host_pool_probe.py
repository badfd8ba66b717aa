import ctypes,sys,os
sys.path.insert(0,'/root/repo' if os.path.isdir('/root/repo/rofl_project_code_amd') else '.')
from rofl_project_code_amd import api
L=api.lib(); ns=ctypes.c_double()
for what in (7,8,9):
    L.rofl_dbg_host_bench(what, 300, ctypes.byref(ns)); print(os.environ.get("ROFL_POOL_SPIN_US","150"), os.environ.get("ROFL_HOST_THREADS","auto"), what, round(ns.value/1e3,1),'us')
