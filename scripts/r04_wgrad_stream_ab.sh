# round 4: training step with the weight gradients on their own stream vs everything on one stream, same box, interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for i in 1 2; do
  for v in 1 0; do
    VPHO_WGRAD_STREAM=$v timeout -k 10 300 python train.py --steps 10 --warmup 3 > $O/r04_ws_${v}_${i}.json 2> $O/r04_ws.err || exit 1
    python - <<PY
import json
d=json.loads(open('$O/r04_ws_${v}_${i}.json').read().strip().splitlines()[-1]); print('wgrad_stream=$v run $i:', round(d['ms_per_step'],2), 'ms/step', round(d['value'],1), 'img/s', flush=True)
PY
  done
done
