# One command under every value of one environment variable:  bash scripts/sweep_env.sh VAR v1 v2 ... -- python3 scripts/wgrad_layers.py
# Output of value v: gpurun_out/sweep_VAR_v.txt (first line echoed).  Replaces round 4's weight-gradient planning sweeps
# (VPHO_WGRAD_WANT 256 ... 1024, VPHO_WGRAD_STAGES 16 32 64).
VAR=$1; shift; VALS=(); while [ "$1" != "--" ]; do VALS+=("$1"); shift; done; shift
R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out; cd $R
for v in "${VALS[@]}"; do
  env $VAR=$v bash -c 'source scripts/gstep.sh; gstep 300 "$0" "$@"' $O/sweep_${VAR}_$v.txt "$@" || exit 1
  echo "$VAR=$v: $(head -1 $O/sweep_${VAR}_$v.txt)"
done
