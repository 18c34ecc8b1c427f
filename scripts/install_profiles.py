"""Copy the summaries scripts/profile_round.sh left under gpurun_out/ (<tag>_*) into profiles/ with their command headers:
    python scripts/install_profiles.py r05 r05b     (each part -- cfg2 / cfg4 / train / force -- is installed when its files are present)"""
import json, shutil, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
tag = sys.argv[2] if len(sys.argv) > 2 else rnd + 'b'
g = 'gpurun_out/'
import os


def _cfg2():
    # the bench prints a compact line (<= 6 KB) since round 6; the numbers quoted in the headers come from the full record of the same run
    seq, pipe = json.load(open(g + tag + '_seq_detail.json')), json.load(open(g + tag + '_pipe_detail.json'))
    A, As, B = open(g + tag + '_seq_all.txt').read(), open(g + tag + '_seq_ss.txt').read(), open(g + tag + '_pipe_ss.txt').read()
    r = seq['roofline']
    txt = f"""# (A) rocprofv3 --kernel-trace --stats -- python3 bench.py --no_cpu_baseline --no_opt_in --pipeline 1 --steps 10   (MI355X, {rnd}, default plan: Winograd F(2x2,3x3) for the 3x3 / stride-1 convolutions incl. the RoI-windowed FPN ones, score-head tail tiles; sequential steps: 3 warm-up + 10 timed + the instrumented repeats that feed the roofline / hbm blocks; README config bs=64, sample_num=100, sampling_steps=50, T0=0.65; weights = vpho_amd.synth.bench_state_dict: nfev 51/51)
    # this run's bench line:
    #   value {seq['value']:.1f} images/s (sequential, under the profiler), roofline.avg_launch_us {r['avg_launch_us']:.2f} -> {r['achieved']:.1f} TFLOP/s, frac {r['frac']:.3f}; score_head {r['score_head']['achieved']:.1f} TFLOP/s in-run, {r['score_head']['samplers_serialised']['achieved']:.1f} with the samplers serialised
    # summary produced from the rocpd database by scripts/rocpd_stats.py (whole trace); recipe: scripts/profile_round.sh + scripts/install_profiles.py
    {A}
    # (A') same trace, steady state only (--last-ms 600)
    {As}
    # (B) rocprofv3 --kernel-trace --stats -- python3 bench.py --no_cpu_baseline --no_opt_in --no_kernel_timing --steps 10   (the DEFAULT evaluator: three batches in flight; kernels of different batches and of the two samplers overlap, so per-kernel durations are NOT exclusive times and their sum exceeds the wall time); steady state of the PIPELINED steps only (scripts/_rocpd.py: the end of a bench trace is its sequential legs)
    #   this run's bench line: {pipe['value']:.1f} images/s, {pipe['ms_per_step']:.2f} ms/step under the profiler
    {B}"""
    open(f'profiles/{rnd}_kernel_stats_bench_cfg2.txt', 'w').write(txt)
    hdr = f"# HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of  VPHO_GRAPHS=0 python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_opt_in --no_kernel_timing --pipeline 1  ({rnd}; scripts/profile_round.sh, summary by scripts/pmc_summary.py)\n# per-launch averages over all launches of each kernel; HBM column = (2*FETCH_SIZE + WRITE_SIZE)*1024 B (counters in KB; gfx950: FETCH_SIZE counts half of wide coalesced reads, MI355X_MICROARCH.md)"
    open(f'profiles/{rnd}_pmc_hbm_traffic.txt', 'w').write(hdr + '\n' + open(g + tag + '_pmc_hbm.txt').read())
    hdr = f"# MFMA pipe utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT) of the same command ({rnd}; summary by scripts/pmc_mfma_summary.py)"
    open(f'profiles/{rnd}_pmc_mfma_busy.txt', 'w').write(hdr + '\n' + open(g + tag + '_pmc_mfma.txt').read())
    if os.path.exists(g + tag + '_exposed.txt'):
        open(f'profiles/{rnd}_exposed_time.txt', 'w').write(f"# time of the pipelined steps (trace B of the kernel-stats file, same window) in which no convolution / score-head kernel executes, by the kernels that run there ({rnd}; scripts/rocpd_exposed.py)\n" + open(g + tag + '_exposed.txt').read())
    shutil.copy(g + tag + '_pmc_hbm.json', f'profiles/{rnd}_pmc_hbm_traffic.json')
    shutil.copy(g + tag + '_bench_default.json', f'profiles/{rnd}_bench_default.json')              # the line as the driver parses it
    shutil.copy(g + tag + '_bench_default_detail.json', f'profiles/{rnd}_bench_detail.json')          # the full record of the same run
    d = json.load(open(g + tag + '_bench_default_detail.json'))
    print('default bench:', d['value'], d['ms_per_step'], 'roofline frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
    print('head:', d['roofline']['score_head'])
    print('hbm:', {k: (round(v['GB/s']), round(v['frac'], 3), round(v['avg_launch_us'], 1)) for k, v in d['hbm']['kernels'].items()})
    print('cpu:', d['cpu_baseline']); p = d['parity']
    print({k: v for k, v in p['end_to_end_vs_oracle'].items() if k != 'per_stage'}); print({k: v for k, v in p['aggregation_given_identical_candidates'].items() if k != 'per_stage'})



if os.path.exists(g + tag + '_seq_detail.json'):           # PART=cfg2 files (a tag may hold only the cfg4 / train / force parts)
    _cfg2()

# ---- the other configurations (profile_round.sh PART="cfg4 train force"), installed when their files exist
def _copy(src, dst, header=None):
    if not os.path.exists(g + src):
        return False
    body = open(g + src).read()
    open('profiles/' + dst, 'w').write((header + '\n' if header else '') + body)
    return True


if _copy(tag + '_bench_cfg4.json', f'{rnd}_bench_cfg4.json'):
    _copy(tag + '_cfg4_stats.txt', f'{rnd}_kernel_stats_bench_cfg4.txt',
          f'# rocprofv3 --kernel-trace --stats -- python3 bench.py --bs 128 --sample_num 256 --sampling_steps 100 --steps 4 --warmup 2 --no_cpu_baseline --no_opt_in --no_kernel_timing --pipeline 1   (MI355X, {rnd}; BASELINE configs[3]; bench line: profiles/{rnd}_bench_cfg4.json)')
    _copy(tag + '_bench_cfg4_detail.json', f'{rnd}_bench_cfg4_detail.json')
    if os.path.exists(g + tag + '_cfg4_pmc_hbm.json'):
        shutil.copy(g + tag + '_cfg4_pmc_hbm.json', f'profiles/{rnd}_pmc_hbm_traffic_cfg4.json')
        _copy(tag + '_cfg4_pmc_hbm.txt', f'{rnd}_pmc_hbm_traffic_cfg4.txt', f'# HBM traffic per launch at BASELINE configs[3] (two rocprofv3 --pmc passes: FETCH_SIZE; WRITE_SIZE) of  VPHO_GRAPHS=0 python3 bench.py --bs 128 --sample_num 256 --sampling_steps 100 --steps 1 --warmup 1 --no_cpu_baseline --no_opt_in --no_kernel_timing --pipeline 1  ({rnd})')
    _copy(tag + '_cfg4_ab.txt', f'{rnd}_cfg4_switch_ab.txt', f'# interleaved A/B on ONE box at BASELINE configs[3] (bs 128, 256 hypotheses, 100 stamps; 6 steps, pipelined): a = switch off, b = on (the default); scripts/ab.sh ({rnd})')
    c4 = json.load(open(g + tag + '_bench_cfg4_detail.json'))
    print('cfg4:', c4['metric'], round(c4['value'], 1), round(c4['ms_per_step'], 2), 'head', c4['roofline']['score_head']['samplers_serialised'], 'pose encoder', c4['roofline'].get('pose_encoder'))
for bs_tag in ('', '_bs32'):
    if _copy(tag + f'_train_step{bs_tag}.json', f'{rnd}_train_step{bs_tag}.json'):
        t = json.load(open(g + tag + f'_train_step{bs_tag}.json'))
        print(f'train{bs_tag}:', round(t['value'], 1), round(t['ms_per_step'], 2), (t.get('roofline') or {}).get('frac'), (t.get('roofline') or {}).get('traffic'))
_copy(tag + '_train_stats.txt', f'{rnd}_train_step_kernel_stats.txt',
      f'# VPHO_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python3 train.py --steps 5 --warmup 2   (MI355X, {rnd}; ONE stream, so durations are exclusive; steady state = the last 370 ms; then the idle-gap analysis of scripts/rocpd_gaps.py)')
_copy(tag + '_train_stats_bs32.txt', f'{rnd}_train_step_bs32_kernel_stats.txt',
      f'# VPHO_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python3 train.py --bs 32 --steps 5 --warmup 2   (MI355X, {rnd}; cfg3\'s own per-GPU batch; one stream; steady state = the last 200 ms)')
if os.path.exists(g + tag + '_train_pmc_hbm.json'):
    shutil.copy(g + tag + '_train_pmc_hbm.json', f'profiles/{rnd}_train_pmc_hbm_traffic.json')
    _copy(tag + '_train_pmc_hbm.txt', f'{rnd}_train_pmc_hbm_traffic.txt', f'# HBM traffic per launch of one training step (two rocprofv3 --pmc passes: FETCH_SIZE; WRITE_SIZE) of  VPHO_WGRAD_STREAM=0 python3 train.py --steps 1 --warmup 1 --no_roofline  ({rnd}; (2*FETCH_SIZE + WRITE_SIZE)*1024 B)')
_copy(tag + '_train_pmc_mfma.txt', f'{rnd}_train_pmc_mfma_busy.txt', f'# MFMA pipe utilisation per kernel of one training step (one rocprofv3 --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT) of  VPHO_WGRAD_STREAM=0 python3 train.py --steps 1 --warmup 1 --no_roofline  ({rnd}; summary by scripts/pmc_mfma_summary.py)')
if _copy(tag + '_force_optim.json', f'{rnd}_force_optim.json'):
    f = json.load(open(g + tag + '_force_optim.json'))
    print('force:', round(f['value'], 1), f.get('unit'), (f.get('roofline') or {}).get('frac'), (f.get('roofline') or {}).get('traffic'))
if os.path.exists(g + tag + '_fo_pmc_hbm.json'):
    shutil.copy(g + tag + '_fo_pmc_hbm.json', f'profiles/{rnd}_force_pmc_hbm_traffic.json')
    _copy(tag + '_fo_pmc_hbm.txt', f'{rnd}_force_pmc_hbm_traffic.txt', f'# HBM traffic per launch of  python3 force_optim.py --pairs 10048  (two rocprofv3 --pmc passes, {rnd})')
