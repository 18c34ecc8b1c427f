"""Copy the summaries scripts/profile_round.sh left under gpurun_out/ (r02b_*) into profiles/ with their command headers."""
import json, shutil, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r02'
g = 'gpurun_out/'
seq, pipe = json.load(open(g + 'r02b_seq.json')), json.load(open(g + 'r02b_pipe.json'))
A, As, B = open(g + 'r02b_seq_all.txt').read(), open(g + 'r02b_seq_ss.txt').read(), open(g + 'r02b_pipe_ss.txt').read()
r = seq['roofline']
txt = f"""# (A) rocprofv3 --kernel-trace --stats -- python3 bench.py --no_cpu_baseline --pipeline 1 --steps 10   (MI355X, round 2, final kernels: FPN stride-4 outputs on the RoI windows, score-head tail tiles, pre-activation prologue on the direct-to-LDS kernel; sequential steps: 3 warm-up + 10 timed + the instrumented repeats that feed the roofline / hbm blocks; README config bs=64, sample_num=100, sampling_steps=50, T0=0.65; weights = vpho_amd.synth.bench_state_dict: nfev 51/51)
# this run's bench line:
#   value {seq['value']:.1f} images/s (sequential, under the profiler), roofline.avg_launch_us {r['avg_launch_us']:.2f} -> {r['achieved']:.1f} TFLOP/s, frac {r['frac']:.3f}; score_head {r['score_head']['achieved']:.1f} TFLOP/s in-run, {r['score_head']['samplers_serialised']['achieved']:.1f} with the samplers serialised
# summary produced from the rocpd database by scripts/rocpd_stats.py (whole trace); recipe: scripts/profile_round.sh + scripts/install_profiles.py
{A}
# (A') same trace, steady state only (--last-ms 600)
{As}
# (B) rocprofv3 --kernel-trace --stats -- python3 bench.py --no_cpu_baseline --no_kernel_timing --steps 10   (the DEFAULT evaluator: three batches in flight; kernels of different batches and of the two samplers overlap, so per-kernel durations are NOT exclusive times and their sum exceeds the wall time); steady state (--last-ms 300)
#   this run's bench line: {pipe['value']:.1f} images/s, {pipe['ms_per_step']:.2f} ms/step under the profiler
{B}"""
open(f'profiles/{rnd}_kernel_stats_bench_cfg2.txt', 'w').write(txt)
hdr = open(f'profiles/{rnd}_pmc_hbm_traffic.txt').read().split('\n')[:2]
open(f'profiles/{rnd}_pmc_hbm_traffic.txt', 'w').write('\n'.join(hdr) + '\n' + open(g + 'r02b_pmc_hbm.txt').read())
hdr = open(f'profiles/{rnd}_pmc_mfma_busy.txt').read().split('\n')[:1]
open(f'profiles/{rnd}_pmc_mfma_busy.txt', 'w').write('\n'.join(hdr) + '\n' + open(g + 'r02b_pmc_mfma.txt').read())
shutil.copy(g + 'r02b_pmc_hbm.json', f'profiles/{rnd}_pmc_hbm_traffic.json')
shutil.copy(g + 'r02b_bench_default.json', f'profiles/{rnd}_bench_default.json')
d = json.load(open(g + 'r02b_bench_default.json'))
print('default bench:', d['value'], d['ms_per_step'], 'roofline frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('head:', d['roofline']['score_head'])
print('hbm:', {k: (round(v['GB/s']), round(v['frac'], 3), round(v['avg_launch_us'], 1)) for k, v in d['hbm']['kernels'].items()})
print('cpu:', d['cpu_baseline']); p = d['parity']
print({k: v for k, v in p['end_to_end_vs_oracle'].items() if k != 'per_stage'}); print({k: v for k, v in p['aggregation_given_identical_candidates'].items() if k != 'per_stage'})
