"""oracle/judge_fp64.py on the CPU at a small size: the float64 predict runs the whole chain (feature path, both solves on a given
accepted step sequence, the aggregation) and the fp32 oracle -- the reference's arithmetic -- sits on it to fp32 rounding where no
selection is a near-tie; the report of ``judge`` is well formed.  The README-size runs against the HIP path: tests/test_gpu_judge_fp64.py
and scripts/e2e_fp64.py."""
import torch


def test_float64_predict_reproduces_the_fp32_oracle_and_the_report_is_well_formed(sd_contrast, assets):
    from oracle import vpho as OV, judge_fp64 as J
    from vpho_amd.assets import ANCHOR_SKELETON
    from vpho_amd.synth import synth_batch
    bs, S, steps, kh, ko, T0 = 3, 8, 5, 4, 3, 0.3
    kw = dict(sample_num=S, sample_T0=T0, sampling_steps=steps, topk_hand=kh, topk_obj=ko)
    data = synth_batch(bs, assets, seed=5)
    torch.manual_seed(3)
    nh, no = torch.randn(bs * S, 96), torch.randn(bs * S, 9)
    ref, info = OV.predict(sd_contrast, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, **kw)
    o64, d64 = J.predict_fp64(sd_contrast, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, steps_hand=info['hand_ode']['steps'],
                              steps_obj=info['obj_ode']['steps'], chunk=2, **kw)
    assert all(v.dtype == torch.float64 for k, v in o64.items())
    assert float((info['hand_x6d'].double() - o64['hand_x6d']).abs().max()) < 2e-6
    rep = J.judge(ref, J.as_tested(info['agg']), ref, info['agg'], o64, d64, o64, d64, S)
    assert rep['images'] == bs and rep['images_within_1e3_of_fp64'] == {'hip': bs, 'oracle': bs}
    assert rep['hip_vs_oracle']['images_lists_identical'] == bs and rep['images_outside_1e3_between_the_fp32_sides'] == []
    assert rep['max_abs_vs_fp64_where_lists_identical']['oracle'] < 1e-4
    # a perturbed side: swap two hypotheses' roles by feeding another prior draw -> images leave the bar and are located at a stage
    torch.manual_seed(4)
    nh2 = nh + 0.05 * torch.randn_like(nh)
    ref2, info2 = OV.predict(sd_contrast, assets, ANCHOR_SKELETON, data, noise_hand=nh2, noise_obj=no, **kw)
    rep2 = J.judge(ref2, J.as_tested(info2['agg']), ref, info['agg'], o64, d64, o64, d64, S)
    for r in rep2['images_outside_1e3_between_the_fp32_sides']:
        assert r['fp64_order_agrees_with'] in ('hip', 'oracle', 'both', 'neither') and r['max_abs_hip_vs_oracle'] > 1e-3
    assert sum(d['images'] for d in rep2['by_first_stage'].values()) == len(rep2['images_outside_1e3_between_the_fp32_sides'])
