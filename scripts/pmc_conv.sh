# PMC passes over one convolution launch (scripts/conv_one.py).  usage on the GPU box: bash scripts/pmc_conv.sh TAG N H Cin Cout res
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" ; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/scripts/conv_one.py "$@" > $O/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R; python3 - "$O" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' not in r['Kernel_Name']: continue
        t = tot[r['Counter_Name']]; t[0] += float(r['Counter_Value']); t[1] += 1
for k in sorted(tot): print(f'{k:44s} {tot[k][0] / tot[k][1]:16.1f}  (avg over {tot[k][1]} launches)')
PY
