# usage (on the GPU box, via gpurun):  T=r03b bash scripts/profile_round.sh     -> gpurun_out/${T}_*; then scripts/install_profiles.py r03 r03b
T=${T:-r03b}
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out
rm -rf $O/${T}_*; 
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_seq -o s -- python3 $R/bench.py --no_cpu_baseline --pipeline 1 --steps 10 > $O/${T}_seq.json 2> $O/${T}_seq.err && echo seq-done &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_pipe -o p -- python3 $R/bench.py --no_cpu_baseline --no_kernel_timing --steps 10 > $O/${T}_pipe.json 2> $O/${T}_pipe.err && echo pipe-done &&
VPHO_GRAPHS=0 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing --pipeline 1 > $O/${T}_pmc_f.json 2> $O/${T}_pmc_f.err && echo f-done &&
VPHO_GRAPHS=0 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing --pipeline 1 > $O/${T}_pmc_w.json 2> $O/${T}_pmc_w.err && echo w-done &&
VPHO_GRAPHS=0 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/${T}_pmc_m -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing --pipeline 1 > $O/${T}_pmc_m.json 2> $O/${T}_pmc_m.err && echo m-done &&
cd $R && python3 scripts/rocpd_stats.py $(ls gpurun_out/${T}_seq/*/*.db gpurun_out/${T}_seq/*.db 2>/dev/null | head -1) 34 > gpurun_out/${T}_seq_all.txt && python3 scripts/rocpd_stats.py $(ls gpurun_out/${T}_seq/*/*.db gpurun_out/${T}_seq/*.db 2>/dev/null | head -1) 16 --last-ms 600 > gpurun_out/${T}_seq_ss.txt && python3 scripts/rocpd_stats.py $(ls gpurun_out/${T}_pipe/*/*.db gpurun_out/${T}_pipe/*.db 2>/dev/null | head -1) 30 --last-ms 300 > gpurun_out/${T}_pipe_ss.txt &&
python3 scripts/pmc_summary.py $(find gpurun_out/${T}_pmc_f -name '*counter_collection.csv' | head -1) $(find gpurun_out/${T}_pmc_w -name '*counter_collection.csv' | head -1) gpurun_out/${T}_pmc_hbm.json > gpurun_out/${T}_pmc_hbm.txt &&
python3 scripts/pmc_mfma_summary.py $(find gpurun_out/${T}_pmc_m -name '*counter_collection.csv' | head -1) $(find gpurun_out/${T}_pmc_m -name '*kernel_trace.csv' | head -1) > gpurun_out/${T}_pmc_mfma.txt &&
timeout -k 10 400 python3 bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; 
rm -rf gpurun_out/${T}_seq gpurun_out/${T}_pipe gpurun_out/${T}_pmc_f gpurun_out/${T}_pmc_w gpurun_out/${T}_pmc_m; ls -la gpurun_out | grep r02b; head -12 gpurun_out/${T}_seq_ss.txt | cut -c1-160
