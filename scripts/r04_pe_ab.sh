# round 4: pose encoder A/B (VPHO_PE_RING=1: round 3's LDS-ring kernel; default: the register-ring kernel) on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_sampler.py tests/test_gpu_referee.py -m gpu -q -s -p no:cacheprovider > $O/r04_t2.log 2>&1; tail -5 $O/r04_t2.log
for i in 1 2; do
  VPHO_PE_RING=1 timeout -k 10 300 python bench.py --no_cpu_baseline --no_opt_in --steps 20 > $O/r04_pe_ring_$i.json 2> $O/r04_pe_ring_$i.err && echo ring-$i &&
  timeout -k 10 300 python bench.py --no_cpu_baseline --no_opt_in --steps 20 > $O/r04_pe_reg_$i.json 2> $O/r04_pe_reg_$i.err && echo reg-$i || exit 1
done
cd /tmp; export TMPDIR=/tmp
VPHO_PE_RING=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/r04_pe_prof_ring -o pe -- python3 $R/scripts/pe_bench.py > $O/r04_pe_prof_ring.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/r04_pe_prof_reg -o pe -- python3 $R/scripts/pe_bench.py > $O/r04_pe_prof_reg.log 2>&1
cd $R; for v in ring reg; do f=$(find $O/r04_pe_prof_$v -name "*kernel_stats.csv" | head -1); echo $v $f; [ -n "$f" ] && head -8 $f | cut -c1-160; done
