# round 4 (final): bench line + kernel stats of the stress config cfg4 (bs 128, 256 hypotheses, 100 stamps)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 500 python bench.py --bs 128 --sample_num 256 --sampling_steps 100 --steps 6 --warmup 2 --no_cpu_baseline --no_opt_in > $O/r04_bench_cfg4.json 2> $O/r04_bench_cfg4.err && echo cfg4-done &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/r04_cfg4_prof -o c -- python3 $R/bench.py --bs 128 --sample_num 256 --sampling_steps 100 --steps 4 --warmup 2 --no_cpu_baseline --no_opt_in --no_kernel_timing --pipeline 1 > $O/r04_cfg4_prof.json 2> $O/r04_cfg4_prof.err && echo cfg4-prof-done
cd $R && python3 scripts/rocpd_stats.py $(ls $O/r04_cfg4_prof/*/*.db $O/r04_cfg4_prof/*.db 2>/dev/null | head -1) 30 > $O/r04_cfg4_stats.txt
rm -rf $O/r04_cfg4_prof; head -8 $O/r04_cfg4_stats.txt | cut -c1-150
