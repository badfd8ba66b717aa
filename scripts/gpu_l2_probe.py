import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("ROFL_LANES", sys.argv[1] if len(sys.argv) > 1 else "3")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
import rofl_project_code_amd as R
R.set_device(0)
for i in range(3):
    r = bench.l2_composite(R, reps=5, warm=3)
    print(os.environ["ROFL_LANES"], round(r["create_ms"], 2), round(r["verify_ms"], 2), flush=True)
