# PMC counters of the score head kernels in isolation (scripts/head_bench.py); usage: bash scripts/pmc_head.sh [f32|bf16x6|bf16x9]
R=$GRAFT_REPO_ROOT; M=${1:-f32}; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R VPHO_SCORE_MFMA=$M
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sh -o m -- python3 $R/scripts/head_bench.py > $R/gpurun_out/pmc_sh.log 2>&1
cd $R; python3 scripts/pmc_mfma_summary.py $(find gpurun_out/pmc_sh -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_sh -name "*kernel_trace.csv" | head -1) | head -6
cd /tmp; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sh2 -o m -- python3 $R/scripts/head_bench.py > $R/gpurun_out/pmc_sh2.log 2>&1
cd $R; python3 - <<PY
import csv, collections, glob
fs = glob.glob("gpurun_out/pmc_sh2/**/*counter_collection.csv", recursive=True)
if not fs:
    print(open("gpurun_out/pmc_sh2.log").read()[-1500:])
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
    for k in agg:
        if "score_head" in k: print(k, cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in agg[k].items()})
PY
rm -rf gpurun_out/pmc_sh gpurun_out/pmc_sh2
