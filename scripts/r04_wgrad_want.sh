# round 4: sweep of the weight-gradient kernel's workgroup target (VPHO_WGRAD_WANT: 64x64-tile launches get 2 x the value) over the step's shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for w in 256 384 512 640 768 1024; do
  VPHO_WGRAD_WANT=$w timeout -k 10 300 python scripts/wgrad_layers.py > $O/wgrad_want_$w.txt 2> $O/wgrad_want.err || exit 1
  echo "WANT=$w: $(head -1 $O/wgrad_want_$w.txt)"
done
