"""(B) of the README-config parity statement -- on IDENTICAL candidates the HIP aggregation makes the oracle's selections -- on three
further batches of 64 images (other data seeds and prior draws) than the pinned one of tests/test_gpu_fullsize.py and bench.py.  The
candidates are the HIP path's own samples, so this needs no oracle solve: HIP predict + the oracle's aggregation on its hypotheses.

What it shows and asserts: the object lists (translation / rotation / heat-map / physics) and hand levels 0-1 are identical on every
image; first differences occur at hand levels 2-3 only, between candidates whose oracle scores lie within CASCADE_TIE_REL = 2e-4
(observed up to 7.6e-5; the pinned batch: 5.6e-7) -- the cascade's level scores are smooth functions of FK joints that the two sides
compute with different fp32 rounding (oracle/compare.py).  >= 7/8 of the images are identical in every list; on those joints /
vertices / 6-DoF agree to 1e-4 (observed 3e-7); a flipped near-tie moves the fused hand by millimetres on that image (the same
discontinuity the reference has between two of its own runs on different BLAS), MPJPE delta over a batch < 0.1 mm."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
BS, S, STEPS, KH, KO, T0 = 64, 100, 50, 30, 10, 0.65


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_identical_candidate_selection_on_other_batches(model_contrast_cpu, assets, seed):
    from oracle.aggregation import hoi_aggregate
    from oracle.compare import parity_summary, CASCADE_TIE_REL
    from vpho_amd.assets import ANCHOR_SKELETON
    from vpho_amd.configs.args import cfg
    from vpho_amd.synth import synth_batch
    saved = (cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0)
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = S, STEPS, KH, KO, T0
    try:
        data = synth_batch(BS, assets, seed=1000 + seed)
        g = torch.Generator().manual_seed(seed)
        nh, no = torch.randn(BS * S, 96, generator=g), torch.randn(BS * S, 9, generator=g)
        m = copy.deepcopy(model_contrast_cpu).cuda().eval()
        gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
        m(gdata, mode='predict')
        out = m._engine.predict(gdata, noise_hand=nh, noise_obj=no)
        torch.cuda.synchronize()
        gi = m._engine.last_info
    finally:
        cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = saved
    c = lambda t: t.detach().cpu()
    gf = gi['features']
    fl = c(out['diff_final_hand_mano']).reshape(-1, 58)
    same = hoi_aggregate(assets, ANCHOR_SKELETON, cam_intrinsic=data['cam_intr_crop_flip'], root_joint_flip=data['root_joint_flip'],
                         root_joint=data['root_joint'], is_right=data['is_right'], force_local=c(gf['force_local']),
                         is_grasped=data['is_grasped'], hand_pose_diff=fl[:, :48].clone(), hand_pose_regression=c(gf['mano_pose']),
                         hand_shape=fl[:, 48:], hand_heatmap=c(gf['hand_heatmap']), hand_bbox=data['bbox_hand'], hand_topk=KH,
                         obj_pose6d=c(out['diff_final_obj_6d']), obj_heatmap=c(gf['obj_heatmap']), obj_bbox=data['bbox_obj_rect'],
                         obj_topk=KO, obj_name=data['obj_name'])
    same_out = dict(agg_hand_joint=same['hand_agg_joint'], agg_hand_vert=same['hand_agg_vert'], agg_hand_mano=same['hand_agg_mano'],
                    agg_obj_6d=same['obj_agg_6d'])
    res, _ = parity_summary(out, same_out, gi['agg'], same['dbg'], S, bound=CASCADE_TIE_REL)
    print(f'seed {seed}:', {k: v for k, v in res.items() if k != 'per_stage' and not k.startswith('max_abs')}, 'STAGES', {k: v for k, v in res['per_stage'].items() if v['images_primary']})
    assert res['images_with_wrong_selection'] == 0 and res['max_rel_score_gap_at_first_differences'] <= CASCADE_TIE_REL, res
    assert res['images_object_selection_identical'] == BS, res
    assert all(v['images_primary'] == 0 for k, v in res['per_stage'].items() if k not in ('hand_level2', 'hand_level3')), res['per_stage']
    assert res['mpjpe_delta_mm_all'] < 0.1, res
    assert res['images_all_selections_identical'] >= (7 * BS) // 8, res
    for k in ('agg_hand_joint', 'agg_hand_vert', 'agg_obj_6d'):
        assert res[f'max_abs_{k}_where_identical'] < 1e-4, (k, res)
