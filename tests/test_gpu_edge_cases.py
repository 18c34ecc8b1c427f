"""Edge cases of the whole path against the oracle: batch of one (quirk Q3 degenerates to self-attention over a single
image), all-left / all-right hands (x-flips), HO3D joint ordering, nobody grasped (heat-map branch of the object fusion),
top-k equal to the candidate count, odd sizes."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    # name: (bs, S, steps, kh, ko, T0, overrides)
    'batch_of_one':        (1, 5, 3, 6, 3, 0.2, {}),
    'topk_equals_count':   (2, 4, 3, 8, 4, 0.2, {}),                                  # kh = 2S, ko = S
    'all_left_ungrasped':  (3, 6, 4, 5, 3, 0.2, {'is_right': False, 'is_grasped': False}),
    'all_right_ho3d':      (2, 7, 6, 9, 3, 0.2, {'is_right': True, 'is_ho3d': True}),
    'mixed_ho3d_T065':     (4, 5, 5, 7, 3, 0.65, {'is_ho3d': [True, False, True, False]}),
    'cfg4_sizes_512_cand': (3, 256, 100, 30, 10, 0.65, {}),                            # BASELINE configs[3] sample sizes: 2S = 512 = the 8-slot top-k's limit
    'top_k_16_slot_path':  (2, 300, 6, 30, 10, 0.65, {}),                              # 600 candidates per image: the 16-slot top-k kernels
}


@pytest.mark.parametrize('name', list(CASES))
def test_predict_edge_case_matches_oracle(model_cpu, sd, model_contrast_cpu, sd_contrast, assets, name):
    from oracle import vpho as OV
    from vpho_amd.assets import ANCHOR_SKELETON
    from vpho_amd.configs.args import cfg
    from vpho_amd.synth import synth_batch
    bs, S, steps, kh, ko, T0, over = CASES[name]
    if T0 > 0.2:                                  # the README's T0 needs score networks that pull hypotheses into the crop
        model_cpu, sd = model_contrast_cpu, sd_contrast
    data = synth_batch(bs, assets, seed=300 + bs)
    for k, v in over.items():
        data[k] = torch.tensor(v if isinstance(v, list) else [v] * bs)
    if 'is_right' in over:                        # keep root_joint consistent with the flip convention
        data['root_joint'] = data['root_joint_flip'].clone()
        data['root_joint'][~data['is_right'], 0] *= -1
    saved = (cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0)
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = S, steps, kh, ko, T0
    try:
        torch.manual_seed(21)
        nh, no = torch.randn(bs * S, 96), torch.randn(bs * S, 9)
        ref, rinfo = OV.predict(sd, assets, ANCHOR_SKELETON, data, sample_num=S, sample_T0=T0, sampling_steps=steps, topk_hand=kh,
                                topk_obj=ko, noise_hand=nh, noise_obj=no)
        m = copy.deepcopy(model_cpu).cuda().eval()
        gdata = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in data.items()}
        from vpho_amd.model.engine import Engine
        eng = Engine(m)
        out = eng.predict(gdata, noise_hand=nh, noise_obj=no)
        torch.cuda.synchronize()
    finally:
        cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = saved
    assert eng.last_info['hand_ode']['nfev'] == rinfo['hand_ode']['nfev']
    assert eng.last_info['obj_ode']['nfev'] == rinfo['obj_ode']['nfev']
    # everything up to the aggregation is well conditioned at any T0
    for k in ('reg_hand_vert', 'reg_hand_joint', 'hand_heatmap', 'obj_heatmap', 'force_local', 'diff_final_hand_mano',
              'diff_final_hand_joint', 'diff_final_obj_6d', 'diff_inprocess_obj_6d'):
        err = float((out[k].double().cpu() - ref[k].double()).abs().max())
        assert err < (5e-4 if k in ('diff_final_hand_mano', 'diff_final_hand_joint') and T0 > 0.2 else 1e-4), (name, k, err)
    ga, ra = eng.last_info['agg'], rinfo['agg']
    if T0 <= 0.2:                                 # clustered hypotheses: aggregation is conditioned -> exact indices, tight outputs
        for lvl in range(4):
            got = ga['hand_topk'][lvl].cpu()
            got = (got[:, 0] if lvl == 0 else got.permute(0, 2, 1)).numpy()
            ref_idx, ref_val = ra['hand']['topk'][lvl].numpy(), ra['hand']['val'][lvl].numpy()
            # the S regression candidates are identical after level 0 (quirk Q7), so their scores tie up to the last ulp
            # of torch's vectorised CPU kernels; indices are compared wherever the oracle's values are not (nearly) tied
            tied = np.zeros_like(ref_idx, dtype=bool)
            d = np.abs(np.diff(ref_val, axis=1)) < 1e-6
            tied[:, 1:] |= d
            tied[:, :-1] |= d
            assert np.array_equal(got[~tied], ref_idx[~tied]), (name, lvl)
            gv = ga['hand_val'][lvl].cpu()
            gv = (gv[:, 0] if lvl == 0 else gv.permute(0, 2, 1)).numpy()
            # fp32 noise of a level score (oracle/referee.py measures it: ~3e-6 of the sum at level 0, ~1e-5 where FK rounding enters)
            assert np.abs(gv - ref_val).max() < 2e-5 + 3e-6 * np.abs(ref_val).max(), (name, lvl)
        for k in ('transl_topk', 'rot_topk', 'phys_topk', 'heat_topk'):
            assert np.array_equal(ga[k].cpu().view(bs, -1).numpy(), ra[k].numpy()), (name, k)
        for k in ('agg_obj_6d', 'agg_hand_mano', 'agg_hand_vert', 'agg_hand_joint'):
            err = float((out[k].double().cpu() - ref[k].double()).abs().max())
            assert err < 1e-4, (name, k, err)
    else:                                         # T0 = 0.65: selections against the oracle's with the fixed tie bound
        from oracle.compare import parity_summary, E2E_TIE_REL
        res, _ = parity_summary(out, ref, ga, ra, S, bound=E2E_TIE_REL)
        assert res['images_with_gap_above_tie_bound'] == 0 and res['max_rel_score_gap_at_first_differences'] <= E2E_TIE_REL, res
        for k in ('agg_hand_joint', 'agg_hand_vert', 'agg_obj_6d'):
            e = res[f'max_abs_{k}_where_identical']                 # None: no image of this (small) batch has every list identical
            assert e is None or e < 1e-4, (name, k, res)


def test_documented_kernel_limits_raise(assets):
    """2 * sample_num <= 1024 candidates per image (wavefront top-k: 16 slots of 64 lanes), batch <= 256
    (CrossModule attends over the batch axis, quirk Q3: one sequence of bs tokens per workgroup), object / hand point clouds that fit
    LDS.  Each limit is an error with a message, not a wrong answer or a launch failure."""
    from vpho_amd import ops
    from vpho_amd.assets import ANCHOR_SKELETON
    agg = ops.Aggregation(assets, ANCHOR_SKELETON, 'cuda')
    # 2 * sample_num = 1026 candidates in the hand cascade (sample_num = 513)
    hv = torch.rand(1, 1026, 20, device='cuda')
    pose = torch.zeros(1, 1026, 48, device='cuda')
    with pytest.raises(ops.VphoError, match='at most 1024 candidates'):
        agg.hand_fuse_level(hv, pose, 30, 0)
    ok = agg.hand_fuse_level(hv[:, :1024].contiguous(), pose[:, :1024].contiguous(), 30, 0)      # the limit itself is served
    assert ok[1].shape == (1, 1, 30)
    with pytest.raises(ops.VphoError, match='at most 1024 candidates'):
        agg.topk(torch.rand(2, 1025, device='cuda'), 5)
    # batch of 257 images: the cross module's sequence axis
    with pytest.raises(ops.VphoError, match='<= 256'):
        ops.mha(torch.zeros(257 * 65, 3 * 512, device='cuda'), 257, 65, 512, 2)
    # hand physics: 16-byte LDS records per object vertex (the check said 12)
    fp = torch.zeros(1, 31, 32, 3, device='cuda')
    with pytest.raises(ops.VphoError, match='does not fit LDS'):
        agg.hand_phys_score(fp, fp.clone(), torch.zeros(1, 4200, 3, device='cuda'), 1, 31)
    torch.cuda.synchronize()
