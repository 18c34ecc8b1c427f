# The round's committed profiles, on the GPU box (via gpurun):   T=r05a [PART="cfg2 cfg4"] bash scripts/profile_round.sh
#   -> gpurun_out/${T}_*; then (here)  python scripts/install_profiles.py r05 r05a  copies the summaries into profiles/.
# PART selects what runs (space-separated list; default "cfg2"; "all" = everything):
#   cfg2     kernel stats of the sequential + the pipelined bench, FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes, exposed-time analysis, default bench line
#   exposed  only the pipelined kernel trace + exposed-time analysis
#   cfg4     bench line + kernel stats of the stress config (bs 128, 256 hypotheses, 100 stamps)
#   train    training-step bench lines (bs 64 and cfg3's own bs 32) + kernel trace on one stream + idle gaps + PMC traffic passes
#   traintrace  only the training step's one-stream kernel trace + statistics
#   force    force-optimisation bench line (cfg5) + PMC traffic passes
# Every GPU step runs under its own timeout; a step that had to be killed ends its chain (scripts/gstep.sh).
T=${T:-r05a}; PART=${PART:-cfg2}
R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out; source $R/scripts/gstep.sh
db() { ls $O/$1/*/*.db $O/$1/*.db 2>/dev/null | head -1; }
csv() { find $O/$1 -name "*$2.csv" | head -1; }
prof() { local out=$1; shift; (cd /tmp && TMPDIR=/tmp gstep 400 $O/${out}.log rocprofv3 "$@"); }
has() { case " $PART " in *" $1 "*|*" all "*) return 0;; esac; return 1; }
scrub() { for d in "$@"; do [ -n "$d" ] && [ -d "$O/$d" ] && rm -r "$O/$d"; done; return 0; }
B="python3 $R/bench.py --no_cpu_baseline --no_opt_in"
cd $R
if has cfg2; then
  scrub ${T}_seq ${T}_pipe ${T}_pmc_f ${T}_pmc_w ${T}_pmc_m
  prof ${T}_seq --kernel-trace --stats -d $O/${T}_seq -o s -- $B --pipeline 1 --steps 10 && grep '^{' $O/${T}_seq.log > $O/${T}_seq.json && cp $O/bench_detail.json $O/${T}_seq_detail.json &&
  prof ${T}_pipe --kernel-trace --stats -d $O/${T}_pipe -o p -- $B --no_kernel_timing --steps 10 && grep '^{' $O/${T}_pipe.log > $O/${T}_pipe.json && cp $O/bench_detail.json $O/${T}_pipe_detail.json &&
  VPHO_GRAPHS=0 prof ${T}_pmc_f --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_f -o f -- $B --steps 2 --warmup 1 --no_kernel_timing --pipeline 1 &&
  VPHO_GRAPHS=0 prof ${T}_pmc_w --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_w -o w -- $B --steps 2 --warmup 1 --no_kernel_timing --pipeline 1 &&
  VPHO_GRAPHS=0 prof ${T}_pmc_m --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/${T}_pmc_m -o m -- $B --steps 2 --warmup 1 --no_kernel_timing --pipeline 1 &&
  python3 scripts/rocpd_stats.py $(db ${T}_seq) 34 > $O/${T}_seq_all.txt && python3 scripts/rocpd_stats.py $(db ${T}_seq) 16 --last-ms 600 > $O/${T}_seq_ss.txt &&
  python3 scripts/rocpd_stats.py $(db ${T}_pipe) 30 --pipelined > $O/${T}_pipe_ss.txt && python3 scripts/rocpd_exposed.py $(db ${T}_pipe) --pipelined > $O/${T}_exposed.txt &&
  python3 scripts/pmc_summary.py $(csv ${T}_pmc_f counter_collection) $(csv ${T}_pmc_w counter_collection) $O/${T}_pmc_hbm.json > $O/${T}_pmc_hbm.txt &&
  python3 scripts/pmc_mfma_summary.py $(csv ${T}_pmc_m counter_collection) $(csv ${T}_pmc_m kernel_trace) > $O/${T}_pmc_mfma.txt &&
  gstep 400 $O/${T}_bench_default.log python3 bench.py && grep '^{' $O/${T}_bench_default.log > $O/${T}_bench_default.json && cp $O/bench_detail.json $O/${T}_bench_default_detail.json
  scrub ${T}_seq ${T}_pipe ${T}_pmc_f ${T}_pmc_w ${T}_pmc_m; head -12 $O/${T}_seq_ss.txt | cut -c1-160
fi
if has exposed; then
  scrub ${T}_pipe
  prof ${T}_pipe --kernel-trace --stats -d $O/${T}_pipe -o p -- $B --no_kernel_timing --steps 10 &&
  python3 scripts/rocpd_stats.py $(db ${T}_pipe) 30 --pipelined > $O/${T}_pipe_ss.txt && python3 scripts/rocpd_exposed.py $(db ${T}_pipe) --pipelined > $O/${T}_exposed.txt
  scrub ${T}_pipe; head -12 $O/${T}_exposed.txt
fi
if has cfg4; then
  # the stress config: bench line, kernel stats, ITS OWN counter passes (a line must never quote another workload's traffic), and interleaved
  # A/Bs of the three switches that could explain a drift between rounds (all bit-identical alternatives)
  C4="--bs 128 --sample_num 256 --sampling_steps 100 --warmup 2"
  scrub ${T}_cfg4_prof ${T}_cfg4_pmc_f ${T}_cfg4_pmc_w
  gstep 500 $O/${T}_bench_cfg4.log $B $C4 --steps 6 && grep '^{' $O/${T}_bench_cfg4.log > $O/${T}_bench_cfg4.json && cp $O/bench_detail.json $O/${T}_bench_cfg4_detail.json &&
  prof ${T}_cfg4_prof --kernel-trace --stats -d $O/${T}_cfg4_prof -o c -- $B $C4 --steps 4 --no_kernel_timing --pipeline 1 &&
  python3 scripts/rocpd_stats.py $(db ${T}_cfg4_prof) 30 > $O/${T}_cfg4_stats.txt &&
  VPHO_GRAPHS=0 prof ${T}_cfg4_pmc_f --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_cfg4_pmc_f -o f -- $B $C4 --steps 1 --warmup 1 --no_kernel_timing --pipeline 1 &&
  VPHO_GRAPHS=0 prof ${T}_cfg4_pmc_w --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_cfg4_pmc_w -o w -- $B $C4 --steps 1 --warmup 1 --no_kernel_timing --pipeline 1 &&
  python3 scripts/pmc_summary.py $(csv ${T}_cfg4_pmc_f counter_collection) $(csv ${T}_cfg4_pmc_w counter_collection) $O/${T}_cfg4_pmc_hbm.json > $O/${T}_cfg4_pmc_hbm.txt
  { for sw in VPHO_HEAD_CB VPHO_CONV_PERS VPHO_WINO_STAGED; do bash scripts/ab.sh cfg4_$sw "$sw=0" "$sw=1" 2 -- python3 bench.py --no_cpu_baseline --no_opt_in --no_kernel_timing $C4 --steps 6 2>&1 | grep run; done; } > $O/${T}_cfg4_ab.txt 2>&1
  scrub ${T}_cfg4_prof ${T}_cfg4_pmc_f ${T}_cfg4_pmc_w; head -8 $O/${T}_cfg4_stats.txt | cut -c1-150; cat $O/${T}_cfg4_ab.txt
fi
if has trainmfma; then
  # only the matrix-pipe counter pass of one training step
  scrub ${T}_train_pmc_m
  VPHO_WGRAD_STREAM=0 prof ${T}_train_pmc_m --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/${T}_train_pmc_m -o m -- python3 $R/train.py --steps 1 --warmup 1 --no_roofline &&
  python3 scripts/pmc_mfma_summary.py $(csv ${T}_train_pmc_m counter_collection) $(csv ${T}_train_pmc_m kernel_trace) > $O/${T}_train_pmc_mfma.txt
  scrub ${T}_train_pmc_m; head -24 $O/${T}_train_pmc_mfma.txt | cut -c1-170
fi
if has traintrace; then
  # only the one-stream kernel trace of the training step + statistics + idle gaps (quick look between kernel changes)
  scrub ${T}_train_prof
  VPHO_WGRAD_STREAM=0 prof ${T}_train_prof --kernel-trace --stats -d $O/${T}_train_prof -o t -- python3 $R/train.py --steps 5 --warmup 2 --no_roofline &&
  { python3 scripts/rocpd_stats.py $(db ${T}_train_prof) 40 --last-ms 370; python3 scripts/rocpd_gaps.py $(db ${T}_train_prof); } > $O/${T}_train_stats.txt 2>&1
  scrub ${T}_train_prof; head -34 $O/${T}_train_stats.txt | cut -c1-150
fi
if has train; then
  gstep 300 $O/${T}_train_step.log python3 train.py --steps 10 --warmup 3 && grep '^{' $O/${T}_train_step.log > $O/${T}_train_step.json &&
  gstep 300 $O/${T}_train_step_bs32.log python3 train.py --bs 32 --steps 10 --warmup 3 && grep '^{' $O/${T}_train_step_bs32.log > $O/${T}_train_step_bs32.json &&
  VPHO_WGRAD_STREAM=0 prof ${T}_train_prof --kernel-trace --stats -d $O/${T}_train_prof -o t -- python3 $R/train.py --steps 5 --warmup 2 &&
  { python3 scripts/rocpd_stats.py $(db ${T}_train_prof) 40 --last-ms 370; python3 scripts/rocpd_gaps.py $(db ${T}_train_prof); } > $O/${T}_train_stats.txt 2>&1 &&
  VPHO_WGRAD_STREAM=0 prof ${T}_train_prof32 --kernel-trace --stats -d $O/${T}_train_prof32 -o t -- python3 $R/train.py --bs 32 --steps 5 --warmup 2 &&
  python3 scripts/rocpd_stats.py $(db ${T}_train_prof32) 40 --last-ms 200 > $O/${T}_train_stats_bs32.txt 2>&1 &&
  VPHO_WGRAD_STREAM=0 prof ${T}_train_pmc_f --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_train_pmc_f -o f -- python3 $R/train.py --steps 1 --warmup 1 --no_roofline &&
  VPHO_WGRAD_STREAM=0 prof ${T}_train_pmc_w --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_train_pmc_w -o w -- python3 $R/train.py --steps 1 --warmup 1 --no_roofline &&
  python3 scripts/pmc_summary.py $(csv ${T}_train_pmc_f counter_collection) $(csv ${T}_train_pmc_w counter_collection) $O/${T}_train_pmc_hbm.json > $O/${T}_train_pmc_hbm.txt &&
  VPHO_WGRAD_STREAM=0 prof ${T}_train_pmc_m --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/${T}_train_pmc_m -o m -- python3 $R/train.py --steps 1 --warmup 1 --no_roofline &&
  python3 scripts/pmc_mfma_summary.py $(csv ${T}_train_pmc_m counter_collection) $(csv ${T}_train_pmc_m kernel_trace) > $O/${T}_train_pmc_mfma.txt
  scrub ${T}_train_prof ${T}_train_prof32 ${T}_train_pmc_f ${T}_train_pmc_w ${T}_train_pmc_m; head -30 $O/${T}_train_stats.txt | cut -c1-160
fi
if has force; then
  gstep 300 $O/${T}_force_optim.log python3 force_optim.py --pairs 10048 && grep '^{' $O/${T}_force_optim.log > $O/${T}_force_optim.json &&
  prof ${T}_fo_pmc_f --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_fo_pmc_f -o f -- python3 $R/force_optim.py --pairs 10048 &&
  prof ${T}_fo_pmc_w --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_fo_pmc_w -o w -- python3 $R/force_optim.py --pairs 10048 &&
  python3 scripts/pmc_summary.py $(csv ${T}_fo_pmc_f counter_collection) $(csv ${T}_fo_pmc_w counter_collection) $O/${T}_fo_pmc_hbm.json > $O/${T}_fo_pmc_hbm.txt
  scrub ${T}_fo_pmc_f ${T}_fo_pmc_w; tail -3 $O/${T}_fo_pmc_hbm.txt
fi
ls $O | grep ${T}_ | tr '\n' ' '
