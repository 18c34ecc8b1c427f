"""Register budget of the hot kernels, from the compiler's own report (vpho_amd/build.py keeps -Rpass-analysis=kernel-resource-usage of
every .hip next to its object).  An edit that pushes a kernel over its budget compiles without a word and shows up only as a slower
bench: in round 4 an epilogue change took the score head from 124 registers to 256 + spills, i.e. from two workgroups per CU to one,
and cost the headline 9 % until the report was read.  No GPU needed: gfx950 is cross-compiled."""
import glob
import os
import re
import subprocess

import pytest

OBJ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vpho_amd', 'csrc', '_obj')
PATTERNS = dict(vgpr=r' VGPRs: (\d+)', spill=r'VGPRs Spill: (\d+)', occupancy=r'Occupancy \[waves/SIMD\]: (\d+)', scratch=r'ScratchSize \[bytes/lane\]: (\d+)',
                lds=r'LDS Size \[bytes/block\]: (\d+)')


@pytest.fixture(scope='module')
def kernels():
    from vpho_amd.build import build_extension
    build_extension()                                        # compiles whatever has no report yet
    files = glob.glob(os.path.join(OBJ, '*.usage.txt'))
    assert len(files) >= 12, files
    raw = {}
    for f in files:
        name = None
        for line in open(f):
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                name = m.group(1)
                raw[name] = {}
            for key, pat in PATTERNS.items():
                m = re.search(pat, line)
                if m and name:
                    raw[name][key] = int(m.group(1))
    names = list(raw)
    dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
    return {re.sub(r'\(anonymous namespace\)::', '', d).split('(')[0].replace('void ', ''): raw[n] for n, d in zip(names, dem)}


def test_no_kernel_spills_vector_registers_unannounced(kernels):
    """Kernels known to spill say why in their sources: the persistent force optimiser (1024-thread workgroup: 128 registers, of which 49
    hold per-item state; since round 6 its two iteration loops are scratch-free -- what is left is loop-invariant values parked around
    the loop that does not use them, the table fill and the once-per-launch report: 18 registers, down from 179), the attention
    backward (2 registers) and the training-only epilogues with BatchNorm sums (1-3 registers, outside the loops)."""
    # the Winograd inference kernels are scratch-free since round 6 (the tile records of the output transform are fetched in two batches); the training
    # instantiations with the BatchNorm sums in the output transform (bn: forward sums, bnb: backward sums with the gate recomputed) park up to two
    # accumulator dwords, behind the main loop
    allowed = {'force_optim_kernel<false>': 20, 'force_optim_kernel<true>': 28, 'mha_bwd_kernel': 8,
               'conv_winograd_bn_kernel<0>': 2, 'conv_winograd_bn_kernel<1>': 2, 'conv_winograd_bnb_kernel<0>': 3, 'conv_winograd_bnb_kernel<1>': 3,
               # the persistent walk with the BatchNorm sums: the statistics pointer is parked across the tile loop (one 8-byte reload per TILE, none in the k loop)
               'conv_igemm_pers_bn_kernel<128, 128, 4, 2>': 2}
    bad = {k: v['spill'] for k, v in kernels.items() if v.get('spill', 0) > allowed.get(k, 0)}
    assert not bad, bad


# kernel -> (waves per SIMD at least, registers at most): the occupancy each launch configuration relies on
BUDGET = {
    'score_head_kernel<true>': (4, 128), 'score_head_kernel<false>': (4, 128),      # two 512-thread workgroups per CU (57 KB of LDS each)
    'pose_encoder_reg_kernel<1>': (4, 128), 'pose_encoder_reg_kernel<3>': (4, 128), 'pose_encoder_reg_kernel<4>': (4, 128),
    'pose_encoder_reg64_kernel<1>': (4, 128), 'pose_encoder_reg64_kernel<3>': (4, 128), 'pose_encoder_reg64_kernel<4>': (4, 128),
    'conv_igemm_glds_kernel<128, 128, 4, 2, false>': (4, 128), 'conv_igemm_glds_kernel<128, 64, 4, 2, false>': (4, 128),
    'conv_igemm_glds_kernel<64, 64, 2, 2, false>': (4, 128),
    # the persistent tile walk sizes its grid as 2 workgroups per CU (<= 128 registers, 73 KB of LDS): a compiler change that drops
    # the occupancy would leave half of the slots idle without a word
    'conv_igemm_pers_kernel<128, 128, 4, 2>': (4, 128), 'conv_igemm_pers_bn_kernel<128, 128, 4, 2>': (4, 128),
    'conv_winograd_kernel<0>': (1, 256), 'conv_winograd_kernel<1>': (1, 256), 'conv_winograd_kernel<2>': (1, 256),   # one wave per SIMD by design: 256 accumulators
    'conv_winograd_bn_kernel<0>': (1, 256), 'conv_winograd_bn_kernel<1>': (1, 256), 'conv_winograd_bnb_kernel<0>': (1, 256), 'conv_winograd_bnb_kernel<1>': (1, 256),
    'conv_winograd8_kernel': (2, 256),
    'conv_wgrad_tn_kernel<64, 64, 2, 2>': (4, 128), 'conv_wgrad_tn_kernel<128, 128, 4, 2>': (4, 128), 'conv_wgrad_tn_kernel<128, 64, 4, 2>': (4, 128),
    'mano_fk_kernel<16>': (3, 168),
}


@pytest.mark.parametrize('name', sorted(BUDGET))
def test_hot_kernels_keep_their_register_budget(kernels, name):
    assert name in kernels, sorted(kernels)[:20]
    waves, regs = BUDGET[name]
    k = kernels[name]
    spill_ok = 2 if name.startswith(('conv_winograd_bn_kernel<', 'conv_igemm_pers_bn_kernel<')) else 3 if name.startswith('conv_winograd_bnb_kernel<') else 0
    assert k['occupancy'] >= waves and k['vgpr'] <= regs and k.get('spill', 0) <= spill_ok, (name, k)


def test_force_optimiser_keeps_its_workgroup_shape(kernels):
    """One 1024-thread workgroup per batch: 4 waves per SIMD at 128 registers, 144 KB of dynamic LDS for the AdamW moments beside
    ~10.5 KB of static LDS (bias-correction table, gravity, reductions), scratch only outside the iteration loops (<= 96 B per lane)."""
    for name in ('force_optim_kernel<false>', 'force_optim_kernel<true>'):
        k = kernels[name]
        assert k['occupancy'] >= 4 and k['vgpr'] <= 128 and k['scratch'] <= 96, (name, k)
        assert k['lds'] + 2 * 9 * 2 * 1024 * 4 <= 160 * 1024, (name, k)


@pytest.mark.parametrize('name', ['hand_fuse_kernel<1>', 'hand_fuse_kernel<2>', 'hand_fuse_kernel<4>', 'hand_fuse_any_kernel', 'hand_phys_fuse_kernel',
                                  'obj_fuse_kernel', 'hand_metrics_kernel', 'obj_metrics_kernel'])
def test_eigen_solves_stay_in_registers(kernels, name):
    """The 4x4 / 3x3 Jacobi solves index their matrices with compile-time constants only (rot.h sym4_top_eigenvector, metrics.hip
    sym3_eig): loop-variable indices put them in scratch (80-144 B per lane until round 5; the fuse kernels' HBM traffic was 8.9 x
    their algorithmic bytes)."""
    assert kernels[name]['scratch'] == 0 and kernels[name].get('spill', 0) == 0, (name, kernels[name])
