# Counter passes over ONE kernel's stand-alone script (each counter set is its own rocprofv3 --pmc run with --kernel-trace only):
#   bash scripts/pmc_one.sh TAG KERNEL_SUBSTRING [sq|mem|all] -- python3 scripts/conv_one.py 64 32 128 512 1
#   bash scripts/pmc_one.sh wino conv_winograd sq -- python3 scripts/wino_one.py 64 64 256 256
# Prints the per-launch average of every counter over the launches whose name contains KERNEL_SUBSTRING.  Replaces the per-kernel
# recipes of rounds 3-4 (one convolution launch, the score head, the Winograd launch).
TAG=$1; KN=$2; SETS=${3:-all}; shift 3; shift
R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O; source $R/scripts/gstep.sh
SQ=("SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"
    "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
    "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU")
MEM=("TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
     "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum"
     "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "FETCH_SIZE" "WRITE_SIZE")
LIST=(); [ $SETS != mem ] && LIST+=("${SQ[@]}"); [ $SETS != sq ] && LIST+=("${MEM[@]}")
i=0
for set in "${LIST[@]}"; do
  i=$((i+1)); (cd /tmp && TMPDIR=/tmp gstep 150 $O/p$i.log rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- "$@") || exit 1
done
cd $R; python3 - "$O" "$KN" <<'PY'
import csv, glob, sys, collections
tot, dur = collections.defaultdict(lambda: [0.0, 0]), []
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            t = tot[r['Counter_Name']]; t[0] += float(r['Counter_Value']); t[1] += 1
for f in glob.glob(sys.argv[1] + '/p1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']: dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
print('durations (us) in pass 1:', [round(d, 1) for d in dur][:12])
for k in sorted(tot): print(f'{k:44s} {tot[k][0] / tot[k][1]:18.1f}  (avg over {tot[k][1]} launches)')
PY
