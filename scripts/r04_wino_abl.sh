R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for m in 0 1 2 4 3 7 16 23; do echo "== VPHO_WINO_ABL=$m"; VPHO_WINO8=1 VPHO_WINO_ABL=$m python scripts/wino_bench.py 2>&1 | grep "H64 256->256\|H32 128->128" | sed 's/direct.*winograd/winograd/'; done
echo "== v3"; VPHO_WINO8=0 python scripts/wino_bench.py 2>&1 | grep "H64 256->256\|H32 128->128" | sed 's/direct.*winograd/winograd/'
