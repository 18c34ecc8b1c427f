#!/bin/bash
# A/B of HIP-graph replay vs plain launches under host CPU contention (N busy-loop processes), README config.
N=${1:-24}
pids=""
for i in $(seq $N); do python -c "
import time
t=time.time()
while time.time()-t<150: pass" & pids="$pids $!"; done
sleep 2
echo "graphs on, $N hogs";  timeout -k 10 120 python bench.py --no_cpu_baseline --no_kernel_timing --steps 20 | cut -c1-330
echo "graphs off, $N hogs"; VPHO_GRAPHS=0 timeout -k 10 120 python bench.py --no_cpu_baseline --no_kernel_timing --steps 20 | cut -c1-330
kill $pids 2>/dev/null
wait 2>/dev/null
