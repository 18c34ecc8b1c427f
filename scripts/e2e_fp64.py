"""End-to-end fp64 judge over several (data, prior) seeds at the README config (oracle/judge_fp64.py): per seed one 64-image batch through
the HIP path, the fp32 oracle, and the float64 predict on either side's accepted step sequences.
    python scripts/e2e_fp64.py [--images 64] [--out gpurun_out/r06_e2e_fp64.json] [seed ...]      (default seeds 777 1 2 3; ~2 min of CPU per seed)
    python scripts/e2e_fp64.py --images 128 --sample_num 256 --sampling_steps 100 --chunk 4 5      (BASELINE cfg4, one seed: ~15 min)
-> one JSON line per seed (progress) and the file: per seed the table of oracle.judge_fp64.judge, plus the totals over the seeds
(profiles/r06_e2e_fp64.json is this file)."""
import argparse, json, os, sys, time
ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=64)
ap.add_argument('--sample_num', type=int, default=100)
ap.add_argument('--sampling_steps', type=int, default=50)
ap.add_argument('--chunk', type=int, default=8, help='images per aggregation call of the float64 predict (memory)')
ap.add_argument('--out', default=os.path.join('gpurun_out', 'r06_e2e_fp64.json'))
ap.add_argument('seeds', nargs='*', type=int)
args = ap.parse_args()
seeds = args.seeds or [777, 1, 2, 3]
sys.argv = sys.argv[:1]; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import bench_state_dict, synth_batch
from vpho_amd.hostcpu import usable_cpus
from oracle import vpho as OV, judge_fp64 as J
torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))
S, steps, KH, KO, T0, n = args.sample_num, args.sampling_steps, 30, 10, 0.65, args.images
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = S, steps, KH, KO, T0
kw = dict(sample_num=S, sample_T0=T0, sampling_steps=steps, topk_hand=KH, topk_obj=KO)
assets = synthetic_assets(0)
m = vpho_net(assets); sd = bench_state_dict(m, seed=1); m.load_state_dict(sd); m = m.cuda().eval()
per_seed = {}
for seed in seeds:
    data = synth_batch(n, assets, seed=seed)
    torch.manual_seed(99 + seed)
    nh, no = torch.randn(n * S, 96), torch.randn(n * S, 9)
    say = lambda what: print(f'[seed {seed}] {time.strftime("%H:%M:%S")} {what}', file=sys.stderr, flush=True)      # a long silent phase looks hung to a watchdog
    say('fp32 oracle ...')
    t0 = time.time()
    ref, info = OV.predict(sd, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, **kw)
    t_or = time.time() - t0
    say('HIP predict ...')
    gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    m(gdata, mode='predict')                                   # builds the engine
    eng = m._engine
    eng.keep_states = True
    out = eng.predict(gdata, noise_hand=nh, noise_obj=no)
    torch.cuda.synchronize()
    gi = eng.last_info
    eng.keep_states = False
    out = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in out.items()}
    say('float64 feature path ...')
    t0 = time.time()
    f64 = J.features64(sd, assets, data)
    t_feat = time.time() - t0
    say('float64 predict on the HIP step sequences ...')
    t0 = time.time()
    o64h, d64h = J.predict_fp64(sd, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, steps_hand=gi['hand_ode']['steps'], steps_obj=gi['obj_ode']['steps'], feat64=f64, chunk=args.chunk, **kw)
    say('float64 predict on the oracle step sequences ...')
    o64o, d64o = J.predict_fp64(sd, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, steps_hand=info['hand_ode']['steps'], steps_obj=info['obj_ode']['steps'], feat64=f64, chunk=args.chunk, **kw)
    t_64 = time.time() - t0
    rep = J.judge(out, gi['agg'], ref, info['agg'], o64h, d64h, o64o, d64o, S)
    rep['hypotheses_max_abs_vs_fp64'] = {'hip_hand': float((gi['hand_x6d'].cpu().double() - o64h['hand_x6d']).abs().max()), 'oracle_hand': float((info['hand_x6d'].double() - o64o['hand_x6d']).abs().max()),
                                         'hip_obj': float((out['diff_final_obj_6d'].double() - o64h['diff_final_obj_6d']).abs().max()), 'oracle_obj': float((ref['diff_final_obj_6d'].double() - o64o['diff_final_obj_6d']).abs().max())}
    rep['seconds'] = {'oracle_fp32': round(t_or, 1), 'features_fp64': round(t_feat, 1), 'two_fp64_predicts': round(t_64, 1)}
    per_seed[str(seed)] = rep
    print(json.dumps({'seed': seed, 'within_1e-3_of_fp64': rep['images_within_1e3_of_fp64'], 'lists_identical_to_fp64': rep['images_lists_identical_to_fp64'],
                      'hip_vs_oracle': rep['hip_vs_oracle'], 'by_first_stage': rep['by_first_stage'], 'seconds': rep['seconds']}), flush=True)
    tot = {'hip': [r['images_within_1e3_of_fp64']['hip'] for r in per_seed.values()], 'oracle': [r['images_within_1e3_of_fp64']['oracle'] for r in per_seed.values()]}
    by = {}
    for r in per_seed.values():
        for st, d in r['by_first_stage'].items():
            t = by.setdefault(st, {k: 0 for k in d})
            for k, v in d.items():
                t[k] += v
    summary = {'config': {'images': n, 'sample_num': S, 'sampling_steps': steps, 'topk_hand': KH, 'topk_obj': KO, 'sample_T0': T0},
               'what': '%d images per seed; within 1e-3 (21 joints, 778 vertices, object 6-DoF) of the float64 predict on the side\'s own accepted step '
                       'sequences (oracle/judge_fp64.py); by_first_stage: the images on which the two fp32 sides differ by more than 1e-3, by the first selection '
                       'that differs, and which side\'s list there the float64 order agrees with' % n,
               'seeds': list(per_seed), 'images_within_1e-3_of_fp64_per_seed': tot,
               'hip_minus_oracle_per_seed': [a - b for a, b in zip(tot['hip'], tot['oracle'])],
               'hip_vs_oracle_within_1e-3_per_seed': [r['hip_vs_oracle']['images_within_1e3'] for r in per_seed.values()],
               'lists_identical_to_fp64_per_seed': {'hip': [r['images_lists_identical_to_fp64']['hip'] for r in per_seed.values()],
                                                    'oracle': [r['images_lists_identical_to_fp64']['oracle'] for r in per_seed.values()]},
               'by_first_stage_all_seeds': by}
    os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
    with open(os.path.join(ROOT, args.out), 'w') as f:
        json.dump({'summary': summary, 'per_seed': per_seed}, f, indent=1)
print(json.dumps(summary))
