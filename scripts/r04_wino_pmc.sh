# round 4: PMC passes on one large Winograd launch for both kernels (VPHO_WINO_V3=1: round 3's).  usage: bash scripts/r04_wino_pmc.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/r04_wino_pmc; rm -rf $O; mkdir -p $O
for v in 4 3; do
  if [ $v = 3 ]; then export VPHO_WINO8=0; else export VPHO_WINO8=1; fi
  i=0
  for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
             "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/v${v}_p$i -o p -- python3 $R/scripts/wino_one.py 64 64 256 256 > $O/v${v}_p$i.log 2>&1 || echo "pass v$v $i failed"
  done
done
unset VPHO_WINO8
cd $R; python3 - "$O" <<'PY'
import csv, glob, sys, collections
for v in ('v4', 'v3'):
    tot = collections.defaultdict(lambda: [0.0, 0])
    dur = []
    for f in glob.glob(sys.argv[1] + f'/{v}_p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'winograd' not in r['Kernel_Name'] or 'weights' in r['Kernel_Name']: continue
            t = tot[r['Counter_Name']]; t[0] += float(r['Counter_Value']); t[1] += 1
    for f in glob.glob(sys.argv[1] + f'/{v}_p1/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_winograd' in r['Kernel_Name']: dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
    print(v, 'durations us', [round(d, 1) for d in dur])
    for k in sorted(tot): print(f'  {k:36s} {tot[k][0] / tot[k][1]:18.1f}  ({tot[k][1]} launches)')
PY
