# round 4: weight-gradient slices of at least 16 / 32 / 64 stages (VPHO_WGRAD_STAGES) with the default workgroup target, over the step's shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for st in 32 16 64; do
  VPHO_WGRAD_STAGES=$st timeout -k 10 300 python scripts/wgrad_layers.py > $O/wgrad_stages_$st.txt 2> $O/wgrad_want.err || exit 1
  echo "STAGES=$st: $(head -1 $O/wgrad_stages_$st.txt)"
done
