"""Top-k parity of the aggregation judged by a kernel-independent fp64 referee (oracle/referee.py) instead of a tie bound.

For every batch the HIP path runs the README config; every one of its nine selection stages is then re-scored ON THE HIP PATH'S OWN
CANDIDATES (its hypotheses, the fused joints each cascade level really saw, its forces) in fp64 and in the reference's fp32
arithmetic (the oracle).  Asserted, for every image, stage and finger:

* regret of the HIP list (fp64 score of the true k-th best minus fp64 score of its worst pick; per rank at the rank-consumed level 3)
  <= 2 x eps32, eps32 = the largest |fp32 - fp64| score error of the REFERENCE's arithmetic on these candidates -- the bound every
  top-k of scores that are within eps32 of the truth obeys, i.e. the HIP pick is one the reference's own arithmetic could have made;
* where the HIP list differs from the fp32 oracle's list on the same candidates, the exchanged candidates lie within 2 x eps32 of each
  other in fp64: lists differ only where the reference cannot resolve the fp64 order itself.

Round 4: every bound is taken per (image, finger) score vector, not as the batch maximum; the tested side's own score error
is bounded by a multiple of the reference's; the number of images whose every list is fp64-optimal has a floor relative to the fp32
oracle's count; and an EXACT-list assertion is back: the oracle's aggregation fed the HIP path's own candidates must pick identical
lists on at least 7 of 8 images (IDENTICAL_FLOOR), and where it does the outputs agree to 1e-4.  Four batches x 64 images with
different data and prior seeds; the trained-checkpoint case is in tests/test_gpu_trained_checkpoint.py."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


from tests._referee import run_hip, assert_within_reference_noise, BS, S, STEPS, KH, KO, T0  # noqa: E402,F401


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_hip_selections_are_within_the_reference_arithmetic_noise_of_the_fp64_order(model_contrast_cpu, assets, seed):
    from oracle import referee as RF
    from vpho_amd.assets import ANCHOR_SKELETON
    from vpho_amd.synth import synth_batch
    data = synth_batch(BS, assets, seed=1000 + seed)
    g = torch.Generator().manual_seed(seed)
    nh, no = torch.randn(BS * S, 96, generator=g), torch.randn(BS * S, 9, generator=g)
    out, info = run_hip(model_contrast_cpu, assets, data, nh, no)
    rec = RF.record_from_hip(out, info, data)
    rep = RF.referee(assets, ANCHOR_SKELETON, rec)
    s = assert_within_reference_noise(rep, f' seed {seed}')
    assert s['all_within_reference_noise']
    if seed < 2:
        given = aggregation_on_identical_candidates(assets, ANCHOR_SKELETON, data, out, info)
        print(f'[identical candidates] seed {seed}: all lists identical on {given["images_all_selections_identical"]}/{BS} images, hand '
              f'{given["images_hand_selection_identical"]}, object {given["images_object_selection_identical"]}; outputs where identical: joints '
              f'{given["max_abs_agg_hand_joint_where_identical"]:.1e}, 6-DoF {given["max_abs_agg_obj_6d_where_identical"]:.1e}')
        assert given['images_all_selections_identical'] >= IDENTICAL_FLOOR * BS, given
        assert given['max_abs_agg_hand_joint_where_identical'] < 1e-4 and given['max_abs_agg_hand_vert_where_identical'] < 1e-4
        assert given['max_abs_agg_obj_6d_where_identical'] < 1e-4


IDENTICAL_FLOOR = 7 / 8


def aggregation_on_identical_candidates(assets, skeleton, data, out, info):
    """the oracle's aggregation (fp32, the reference's arithmetic) fed the HIP path's OWN hypotheses, heat-maps and forces: identical
    inputs on both sides, so what differs is the fp32 rounding of FK / projection / bicubic sums inside the selection chain"""
    from oracle.aggregation import hoi_aggregate
    from oracle.compare import parity_summary, TIE_REL
    c = lambda t: t.detach().cpu()
    gf = info['features']
    fl = c(out['diff_final_hand_mano']).reshape(-1, 58)
    same = hoi_aggregate(assets, skeleton, cam_intrinsic=data['cam_intr_crop_flip'], root_joint_flip=data['root_joint_flip'],
                         root_joint=data['root_joint'], is_right=data['is_right'], force_local=c(gf['force_local']),
                         is_grasped=data['is_grasped'], hand_pose_diff=fl[:, :48].clone(), hand_pose_regression=c(gf['mano_pose']),
                         hand_shape=fl[:, 48:], hand_heatmap=c(gf['hand_heatmap']), hand_bbox=data['bbox_hand'],
                         hand_topk=KH, obj_pose6d=c(out['diff_final_obj_6d']), obj_heatmap=c(gf['obj_heatmap']),
                         obj_bbox=data['bbox_obj_rect'], obj_topk=KO, obj_name=data['obj_name'])
    same_out = dict(agg_hand_joint=same['hand_agg_joint'], agg_hand_vert=same['hand_agg_vert'], agg_hand_mano=same['hand_agg_mano'],
                    agg_obj_6d=same['obj_agg_6d'])
    return parity_summary(out, same_out, info['agg'], same['dbg'], S, bound=TIE_REL)[0]
