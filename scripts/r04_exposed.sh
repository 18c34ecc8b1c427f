# the pipelined kernel trace + exposed-time analysis only (the part of profile_round.sh that scripts/rocpd_exposed.py reads)
T=${T:-r04c}; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out
rm -rf $O/${T}_pipe
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_pipe -o p -- python3 $R/bench.py --no_cpu_baseline --no_opt_in --no_kernel_timing --steps 10 > $O/${T}_pipe.json 2> $O/${T}_pipe.err && echo pipe-done &&
cd $R && python3 scripts/rocpd_stats.py $(ls gpurun_out/${T}_pipe/*/*.db gpurun_out/${T}_pipe/*.db 2>/dev/null | head -1) 30 --pipelined > gpurun_out/${T}_pipe_ss.txt && python3 scripts/rocpd_exposed.py $(ls gpurun_out/${T}_pipe/*/*.db gpurun_out/${T}_pipe/*.db 2>/dev/null | head -1) --pipelined > gpurun_out/${T}_exposed.txt
rm -rf gpurun_out/${T}_pipe; head -12 gpurun_out/${T}_exposed.txt
