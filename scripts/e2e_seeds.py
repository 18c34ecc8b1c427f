"""End-to-end list agreement HIP vs oracle at the README config over several (data, prior) seeds, next to the reference's own run-to-run
agreement (tests/golden/golden_predict_readme64_selfcheck.npz).  One 64-image batch per seed (~40 s of oracle each).
python scripts/e2e_seeds.py [seed ...] -> one JSON line per seed + a summary (profiles/r05_e2e_seeds.json by hand)"""
import json, os, sys, time, torch
seeds = [int(a) for a in sys.argv[1:]] or [777, 1, 2, 3]
sys.argv = sys.argv[:1]; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import bench_state_dict, synth_batch
from vpho_amd.hostcpu import usable_cpus
from oracle import vpho as OV
from oracle.compare import parity_summary, E2E_TIE_REL, reference_self_agreement
torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))
S, steps, KH, KO, T0, n = 100, 50, 30, 10, 0.65, 64
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = S, steps, KH, KO, T0
assets = synthetic_assets(0)
m = vpho_net(assets); sd = bench_state_dict(m, seed=1); m.load_state_dict(sd); m = m.cuda().eval()
rows = []
for seed in seeds:
    data = synth_batch(n, assets, seed=seed)
    torch.manual_seed(99 + seed)
    nh, no = torch.randn(n * S, 96), torch.randn(n * S, 9)
    t0 = time.time()
    ref, info = OV.predict(sd, assets, ANCHOR_SKELETON, data, sample_num=S, sample_T0=T0, sampling_steps=steps, topk_hand=KH, topk_obj=KO, noise_hand=nh, noise_obj=no)
    gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
    m(gdata, mode='predict')
    out = m._engine.predict(gdata, noise_hand=nh, noise_obj=no)
    torch.cuda.synchronize()
    e2e, _ = parity_summary(out, ref, m._engine.last_info['agg'], info['agg'], S, bound=E2E_TIE_REL)
    r = dict(seed=seed, images=n, identical=e2e['images_all_selections_identical'], within_1e3=e2e['images_within_1e-3_on_joints_vertices_6dof'],
             max_gap=e2e['max_rel_score_gap_at_first_differences'], mpjpe_delta_mm=e2e['mpjpe_delta_mm_all'],
             upstream_hand_x=float((m._engine.last_info['hand_x6d'].cpu().double() - info['hand_x6d'].double()).abs().max()),
             upstream_obj_x=float((out['diff_final_obj_6d'].cpu() - ref['diff_final_obj_6d']).abs().max()), seconds=round(time.time() - t0, 1))
    rows.append(r)
    print(json.dumps(r), flush=True)
sc = reference_self_agreement(os.path.join(ROOT, 'tests', 'golden', 'golden_predict_readme64_selfcheck.npz'))
refc = {k: (v['images_all_selections_identical'], v['images_within_1e-3_on_joints_vertices_6dof']) for k, v in sc['variants'].items()}
print(json.dumps({'hip_vs_oracle': {'identical_per_seed': [r['identical'] for r in rows], 'within_1e-3_per_seed': [r['within_1e3'] for r in rows],
                                    'identical_mean': sum(r['identical'] for r in rows) / len(rows), 'within_1e-3_mean': sum(r['within_1e3'] for r in rows) / len(rows)},
                  'reference_vs_itself (identical, within 1e-3) of 64': refc}))
