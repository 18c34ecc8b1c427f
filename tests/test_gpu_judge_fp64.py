"""End-to-end fp64 judge (oracle/judge_fp64.py) at the README config's sizes on a batch of 8 images: the HIP path and the fp32 oracle are
each held against a float64 ``predict`` (feature path, both ODE solves, rot6d -> axis-angle, the nine selections, all in double) that
runs the accepted step sequences of the side it judges.  The four-seed, 64-image table is profiles/r06_e2e_fp64.json
(scripts/e2e_fp64.py): both fp32 sides land within 1e-3 of the float64 result on 59-62 of 64 images, the first selections that flip are
cascade levels 2 and 3, and the float64 order sides with either of them equally often."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._referee import run_hip, S, STEPS, KH, KO, T0  # noqa: E402


def test_hip_is_as_close_to_the_float64_predict_as_the_reference_arithmetic(model_contrast_cpu, sd_contrast, assets):
    from oracle import vpho as OV, judge_fp64 as J
    from vpho_amd.assets import ANCHOR_SKELETON
    from vpho_amd.synth import synth_batch
    n, seed = 8, 777
    kw = dict(sample_num=S, sample_T0=T0, sampling_steps=STEPS, topk_hand=KH, topk_obj=KO)
    data = synth_batch(n, assets, seed=seed)
    torch.manual_seed(99 + seed)
    nh, no = torch.randn(n * S, 96), torch.randn(n * S, 9)
    ref, info = OV.predict(sd_contrast, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, **kw)
    out, gi = run_hip(model_contrast_cpu, assets, data, nh, no)
    out = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in out.items()}
    f64 = J.features64(sd_contrast, assets, data)
    o64h, d64h = J.predict_fp64(sd_contrast, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, steps_hand=gi['hand_ode']['steps'],
                                steps_obj=gi['obj_ode']['steps'], feat64=f64, **kw)
    o64o, d64o = J.predict_fp64(sd_contrast, assets, ANCHOR_SKELETON, data, noise_hand=nh, noise_obj=no, steps_hand=info['hand_ode']['steps'],
                                steps_obj=info['obj_ode']['steps'], feat64=f64, **kw)
    # upstream of the selections both sides sit on their float64 scheme: hypotheses to a few 1e-7 of a rotation-6d component
    ex_h = float((gi['hand_x6d'].cpu().double() - o64h['hand_x6d']).abs().max())
    ex_o = float((info['hand_x6d'].double() - o64o['hand_x6d']).abs().max())
    assert ex_h < 5e-6 and ex_o < 5e-6 and ex_h < 2.0 * ex_o, (ex_h, ex_o)
    rep = J.judge(out, gi['agg'], ref, info['agg'], o64h, d64h, o64o, d64o, S)
    print('[fp64 judge]', {k: rep[k] for k in ('images_within_1e3_of_fp64', 'images_lists_identical_to_fp64', 'first_differing_stage_vs_fp64',
                                               'hip_vs_oracle', 'the_two_fp64_truths_within_1e3_of_each_other', 'max_abs_vs_fp64_where_lists_identical')})
    w = rep['images_within_1e3_of_fp64']
    # a flip is a coin toss between two fp32 roundings of a near-tie: over the 4 x 64 images of the committed table the two sides are 1-2
    # images apart either way; on 8 images HIP may trail the reference's arithmetic by one image, not more
    assert w['hip'] >= w['oracle'] - 1 and w['hip'] >= n - 2, w
    # where a side's lists ARE the float64 lists its outputs are the float64 outputs to fp32 rounding of FK (1e-5 is 100 x that)
    m = rep['max_abs_vs_fp64_where_lists_identical']
    assert m['hip'] is not None and m['hip'] < 1e-5 and (m['oracle'] is None or m['oracle'] < 1e-5), m
    # what the judge found in round 6 and what was changed for it: the S regression copies (half of the candidates) inherit the regression
    # head's rounding noise through the 6-D normalisation -- 2.1e-6 rad rms on the fp32-MFMA GEMM against 1.3e-6 for the reference's
    # arithmetic; with the head's four small linear layers accumulated in double (vpho_linear_acc64_f32) 0.83e-6 -- and the scores of the
    # picks follow: HIP must not be farther from float64 than the reference arithmetic again (measured 0.6-0.7 x at every level)
    c = rep['candidate_pose_abs_err_vs_fp64']
    assert c['hip']['regression_copies_joints_1_15']['rms'] <= 1.1 * c['oracle']['regression_copies_joints_1_15']['rms'], c
    assert c['hip']['diffusion_hypotheses']['rms'] <= 1.5 * c['oracle']['diffusion_hypotheses']['rms'], c
    v = rep['topk_value_rel_err_vs_fp64']
    for lvl in ('level0', 'level1'):
        assert v['hip'][lvl] is not None and v['hip'][lvl]['rms'] <= 1.15 * v['oracle'][lvl]['rms'], (lvl, v)
    # the two float64 results differ only through the step sequences (2e-5 relative in h): far below the bar
    assert rep['the_two_fp64_truths_within_1e3_of_each_other'] >= n - 1
