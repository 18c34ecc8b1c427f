# round 4: the bench lines of the other BASELINE configs (cfg4 stress config, cfg3 training step, cfg5 force optimisation) + kernel stats of cfg4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 500 python bench.py --bs 128 --sample_num 256 --sampling_steps 100 --steps 6 --warmup 2 --no_cpu_baseline --no_opt_in > $O/r04_bench_cfg4.json 2> $O/r04_bench_cfg4.err && echo cfg4-done &&
timeout -k 10 300 python train.py --steps 10 --warmup 3 > $O/r04_train_step.json 2> $O/r04_train_step.err && echo train-done &&
timeout -k 10 300 python force_optim.py --pairs 10048 > $O/r04_force_optim.json 2> $O/r04_force_optim.err && echo fo-done &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/r04_cfg4_prof -o c -- python3 $R/bench.py --bs 128 --sample_num 256 --sampling_steps 100 --steps 4 --warmup 2 --no_cpu_baseline --no_opt_in --no_kernel_timing --pipeline 1 > $O/r04_cfg4_prof.json 2> $O/r04_cfg4_prof.err && echo cfg4-prof-done &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r04_train_prof -o t -- python3 $R/train.py --steps 5 --warmup 2 > $O/r04_train_prof.json 2> $O/r04_train_prof.err && echo train-prof-done
cd $R && python3 scripts/rocpd_stats.py $(ls $O/r04_cfg4_prof/*/*.db $O/r04_cfg4_prof/*.db 2>/dev/null | head -1) 30 > $O/r04_cfg4_stats.txt; python3 scripts/rocpd_stats.py $(ls $O/r04_train_prof/*/*.db $O/r04_train_prof/*.db 2>/dev/null | head -1) 34 --last-ms 370 > $O/r04_train_stats.txt; python3 scripts/rocpd_gaps.py $(ls $O/r04_train_prof/*/*.db $O/r04_train_prof/*.db 2>/dev/null | head -1) >> $O/r04_train_stats.txt 2>&1
rm -rf $O/r04_cfg4_prof $O/r04_train_prof; head -12 $O/r04_cfg4_stats.txt | cut -c1-150; head -12 $O/r04_train_stats.txt | cut -c1-150
