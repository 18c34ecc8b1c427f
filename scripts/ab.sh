# Interleaved A/B of two environment settings over one command that prints a JSON line with ms_per_step (bench.py, train.py), on ONE box:
#   bash scripts/ab.sh NAME "VPHO_X=0" "VPHO_X=1" [REPS] -- python3 bench.py --no_cpu_baseline --no_opt_in --steps 20
# ("-" = no variable; REPS default 2).  Results: gpurun_out/ab_NAME_{a,b}_<i>.json and one summary line per run.  Replaces round 4's
# one-off A/B recipes (residual prefetch VPHO_CONV_DBG=8, VPHO_DOWN_FUSE=0, VPHO_FPN_FUSE=0, VPHO_PE_RING=1, VPHO_WGRAD_STREAM=0 ...).
NAME=$1; A=$2; Bv=$3; shift 3; REPS=2
if [ "$1" != "--" ]; then REPS=$1; shift; fi; shift
R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out; cd $R
for i in $(seq 1 $REPS); do
  for arm in a b; do
    v=$A; [ $arm = b ] && v=$Bv; [ "$v" = "-" ] && v="VPHO_AB_NONE=1"
    env $v bash -c 'source scripts/gstep.sh; gstep 400 "$0" "$@"' $O/ab_${NAME}_${arm}_$i.log "$@" || exit 1
    grep '^{' $O/ab_${NAME}_${arm}_$i.log > $O/ab_${NAME}_${arm}_$i.json
    python3 - "$O/ab_${NAME}_${arm}_$i.json" "$arm ($v) run $i" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2] + ':', round(d['ms_per_step'], 2), 'ms/step', round(d['value'], 1), d['unit'], flush=True)
PY
  done
done
