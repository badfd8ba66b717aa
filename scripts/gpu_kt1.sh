#!/bin/bash
# kernel trace of a single sequential client (uncontended kernel durations); summary -> gpurun_out/kt1.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt1; mkdir -p gpurun_out/kt1
sed -e 's/os.environ\["ROFL_TRACE"\] = "2"/pass/' scripts/gpu_trace1.py > /tmp/kt1.py
cp /tmp/kt1.py scripts/_kt1_tmp.py
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt1 -- python3 scripts/_kt1_tmp.py ${1:-4} > gpurun_out/kt1.log 2>&1
python3 scripts/kt_summary.py gpurun_out/kt1 seq > gpurun_out/kt1.txt
rm -rf gpurun_out/kt1 scripts/_kt1_tmp.py
tail -45 gpurun_out/kt1.txt
