#!/bin/bash
# Training-step A/B on one box: in-launch finish of the column reductions (VPHO_COL_FINISH) x slice-sum kernel of the weight gradients
# (VPHO_WGRAD_REDUCE_GROUPS), interleaved, two runs each.   bash scripts/train_ab.sh [train.py arguments]
cd "$(dirname "$0")/.."
source scripts/gstep.sh
O=gpurun_out
for r in 1 2; do
  for cfg in "separate 1" "fused 1" "separate 0" "fused 0"; do
    set -- $cfg
    VPHO_COL_FINISH=$1 VPHO_WGRAD_REDUCE_GROUPS=$2 gstep 200 $O/train_ab_$1_$2_$r.log python3 train.py --steps 10 --warmup 3 --no_roofline || exit 1
    python3 - "$O/train_ab_$1_$2_$r.log" "$1 groups=$2 run $r" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[2], round(d['ms_per_step'], 2), 'ms/step', list(d['loss_last'].items())[:1])
PY
  done
done
