"""Top-k parity of the aggregation judged by a kernel-independent fp64 referee (oracle/referee.py) instead of a tie bound.

For every batch the HIP path runs the README config; every one of its nine selection stages is then re-scored ON THE HIP PATH'S OWN
CANDIDATES (its hypotheses, the fused joints each cascade level really saw, its forces) in fp64 and in the reference's fp32
arithmetic (the oracle).  Asserted, for every image, stage and finger:

* regret of the HIP list (fp64 score of the true k-th best minus fp64 score of its worst pick; per rank at the rank-consumed level 3)
  <= 2 x eps32, eps32 = the largest |fp32 - fp64| score error of the REFERENCE's arithmetic on these candidates -- the bound every
  top-k of scores that are within eps32 of the truth obeys, i.e. the HIP pick is one the reference's own arithmetic could have made;
* where the HIP list differs from the fp32 oracle's list on the same candidates, the exchanged candidates lie within 2 x eps32 of each
  other in fp64: lists differ only where the reference cannot resolve the fp64 order itself.

No constant is fitted to the side under test and no assertion depends on which near-ties a batch contains: the counts (images whose
every list is fp64-optimal, for HIP and for the fp32 oracle) are printed, not asserted.  Four batches x 64 images with different
data and prior seeds; the trained-checkpoint case is in tests/test_gpu_trained_checkpoint.py."""
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
