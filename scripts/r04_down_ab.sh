# round 4: projection shortcuts merged into conv3 (default) against two launches (VPHO_DOWN_FUSE=0), one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_conv.py tests/test_gpu_predict.py tests/test_gpu_glue.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py -m gpu -q -x -p no:cacheprovider > $O/r04_t6.log 2>&1; tail -5 $O/r04_t6.log
for i in 1 2; do
  VPHO_DOWN_FUSE=0 timeout -k 10 300 python bench.py --no_cpu_baseline --no_opt_in --steps 20 > $O/r04_down_sep_$i.json 2> $O/r04_down_sep_$i.err && echo sep-$i &&
  timeout -k 10 300 python bench.py --no_cpu_baseline --no_opt_in --steps 20 > $O/r04_down_fused_$i.json 2> $O/r04_down_fused_$i.err && echo fused-$i || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_down_*_?.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'], 1), round(d['ms_per_step'], 2), [round(x, 2) for x in d['step_ms_min_median_max']], d['roofline']['launches_per_step'], round(d['roofline']['kernel_ms_per_step'], 2))
PY
