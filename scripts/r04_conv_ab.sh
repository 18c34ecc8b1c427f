# round 4: residual-prefetch A/B (VPHO_CONV_DBG=8: residual loaded in the epilogue, round 3) + conv tests, on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_conv.py tests/test_gpu_predict.py -m gpu -q -x -p no:cacheprovider > $O/r04_t3.log 2>&1; tail -5 $O/r04_t3.log
python scripts/conv_shortk.py > $O/r04_shortk_early.log 2>&1; VPHO_CONV_DBG=8 python scripts/conv_shortk.py > $O/r04_shortk_late.log 2>&1
tail -12 $O/r04_shortk_early.log; tail -12 $O/r04_shortk_late.log
for i in 1 2; do
  VPHO_CONV_DBG=8 timeout -k 10 300 python bench.py --no_cpu_baseline --no_opt_in --steps 20 > $O/r04_res_late_$i.json 2> $O/r04_res_late_$i.err && echo late-$i &&
  timeout -k 10 300 python bench.py --no_cpu_baseline --no_opt_in --steps 20 > $O/r04_res_early_$i.json 2> $O/r04_res_early_$i.err && echo early-$i || exit 1
done
