"""The fp64 referee of the selection chain (oracle/referee.py) on the oracle's own run: CPU only."""
import sys

import pytest
import torch


@pytest.fixture(scope='module')
def oracle_run():
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        from vpho_amd.model.VPHO import vpho_net
        from vpho_amd.synth import bench_state_dict, synth_batch
        from vpho_amd.assets import synthetic_assets, ANCHOR_SKELETON
        from oracle import vpho as OV
    finally:
        sys.argv = argv
    assets = synthetic_assets(0)
    model = vpho_net(assets)
    sd = bench_state_dict(model, 1)
    data = synth_batch(3, assets, seed=11)
    torch.manual_seed(5)
    S = 12
    out, info = OV.predict(sd, assets, ANCHOR_SKELETON, data, sample_num=S, sample_T0=0.65, sampling_steps=5, topk_hand=8, topk_obj=4)
    return assets, ANCHOR_SKELETON, data, out, info


def test_referee_reproduces_the_oracle_scores_and_bounds_its_regret(oracle_run):
    from oracle import referee as RF
    assets, skel, data, out, info = oracle_run
    rec = RF.record_from_oracle(out, info, data)
    # fp32 mode of the stage functions IS the oracle's arithmetic: same score vectors, bit for bit
    d = info['agg']
    for st, want in (('hand_level0', d['hand']['score'][0][:, :, None]), ('hand_level2', d['hand']['score'][2]), ('obj_transl', d['transl_score'][:, :, None]),
                     ('obj_rot', d['rot_score'][:, :, None]), ('obj_heat', d['heat_score'][:, :, None]), ('obj_physics', d['phys_score'][:, :, None]),
                     ('hand_physics', d['hand_phys']['score'].permute(0, 2, 1))):
        got = RF.stage_scores(assets, skel, rec, st, torch.float32, chunk=2)
        assert torch.equal(got, want), st
    rep = RF.referee(assets, skel, rec)
    for st, r in rep.items():
        # the record's lists ARE the fp32 oracle's: no difference, and the 2-eps theorem holds
        assert not bool(r['differs_from_o32'].any()), st
        assert torch.equal(r['regret_rel'], r['regret32_rel']), st
        assert float(r['regret_rel'].max()) <= r['bound_rel'], (st, float(r['regret_rel'].max()), r['bound_rel'])
        assert 0 < r['eps32_rel'] < 1e-4, (st, r['eps32_rel'])                # fp32 noise, not a formula difference
    s = RF.summary(rep)
    assert s['all_within_reference_noise'] and s['images'] == 3


def test_referee_flags_a_wrong_pick(oracle_run):
    """a list that swaps its best pick for the worst candidate has a regret far above the noise bound"""
    from oracle import referee as RF
    assets, skel, data, out, info = oracle_run
    rec = RF.record_from_oracle(out, info, data)
    s64 = RF.stage_scores(assets, skel, rec, 'hand_level0', torch.float64)
    bad = rec['lists']['hand_level0'].clone()
    bad[0, 0, 0] = int(s64[0, :, 0].argmin())
    rec['lists'] = dict(rec['lists'], hand_level0=bad)
    r = RF.referee(assets, skel, rec, stages=('hand_level0',))['hand_level0']
    assert float(r['regret_rel'][0]) > 100 * r['bound_rel'] and bool(r['differs_from_o32'][0])
    assert not RF.summary({'hand_level0': r})['all_within_reference_noise']
