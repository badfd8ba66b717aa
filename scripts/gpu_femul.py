import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rofl_project_code_amd as R
R.set_device(0)
print(os.environ.get("ROFL_FEMUL_LDS", "0"), "%.3e" % R.bench_femul(400))
