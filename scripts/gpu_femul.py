"""Field-arithmetic microbenchmarks (rofl_bench_femul), multiplications per second on the whole chip.
ROFL_FEMUL_MODE unset/0: chains of Fp multiplications (the ceiling `roofline.peak` is quoted against); ROFL_FEMUL_LDS=bytes of dynamic LDS
limits its occupancy.  1 / 2 / 3: the 7-multiplication mixed addition of k_msm_accumulate_fb at that kernel's shape (4 waves/SIMD, <= 128
VGPRs) with the table entry in registers / fetched per addition from a 32 KB table / gathered at random from ROFL_FEMUL_TABLE entries of
128 bytes (2^23 = the 1 GB of the window table)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rofl_project_code_amd as R
R.set_device(0)
print("mode", os.environ.get("ROFL_FEMUL_MODE", "0"), "lds", os.environ.get("ROFL_FEMUL_LDS", "0"), "table", os.environ.get("ROFL_FEMUL_TABLE", "-"), "%.3e" % R.bench_femul(400))
