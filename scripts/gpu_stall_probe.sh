#!/bin/bash
# are the multi-millisecond outliers CFS throttling?  cpu.stat / cpu.max of the cgroup before and after a run of sequential clients
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; nproc
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
python scripts/gpu_lat.py 4 40
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
ROFL_BLOCKING_SYNC=1 python scripts/gpu_lat.py 4 40
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
cat /proc/pressure/cpu 2>/dev/null
uptime
