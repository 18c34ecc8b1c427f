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
    """Three kernels are known to spill and say why in their sources: the persistent force optimiser (two items per thread of a
    1024-thread workgroup: 128 registers; its AdamW moments already live in LDS), the attention backward (2 registers) and the
    Winograd kernel's epilogue (1 register, outside the loop)."""
    # conv_winograd_kernel: ONE accumulator dword saved and restored in the output transform (after the main loop, 256 + 256 registers in use)
    allowed = {'force_optim_kernel': 200, 'mha_bwd_kernel': 8, 'conv_winograd_kernel<0>': 8, 'conv_winograd_kernel<1>': 8, 'conv_winograd_kernel<2>': 8}
    bad = {k: v['spill'] for k, v in kernels.items() if v.get('spill', 0) > allowed.get(k, 0)}
    assert not bad, bad


# kernel -> (waves per SIMD at least, registers at most): the occupancy each launch configuration relies on
BUDGET = {
    'score_head_kernel<true>': (4, 128), 'score_head_kernel<false>': (4, 128),      # two 512-thread workgroups per CU (57 KB of LDS each)
    'pose_encoder_reg_kernel<1>': (4, 128), 'pose_encoder_reg_kernel<3>': (4, 128), 'pose_encoder_reg_kernel<4>': (4, 128),
    'conv_igemm_glds_kernel<128, 128, 4, 2, false>': (4, 128), 'conv_igemm_glds_kernel<128, 64, 4, 2, false>': (4, 128),
    'conv_igemm_glds_kernel<64, 64, 2, 2, false>': (4, 128),
    'conv_winograd_kernel<0>': (1, 256), 'conv_winograd_kernel<1>': (1, 256), 'conv_winograd_kernel<2>': (1, 256),   # one wave per SIMD by design: 256 accumulators
    'conv_winograd8_kernel': (2, 256),
    'conv_wgrad_tn_kernel<64, 64, 2, 2>': (4, 128), 'conv_wgrad_tn_kernel<128, 128, 4, 2>': (4, 128),
    'mano_fk_kernel<16>': (3, 168),
}


@pytest.mark.parametrize('name', sorted(BUDGET))
def test_hot_kernels_keep_their_register_budget(kernels, name):
    assert name in kernels, sorted(kernels)[:20]
    waves, regs = BUDGET[name]
    k = kernels[name]
    # the Winograd kernel saves one accumulator dword in its output transform (behind the main loop; timed the same: DESIGN 4c)
    spill_ok = 1 if name.startswith('conv_winograd_kernel<') else 0
    assert k['occupancy'] >= waves and k['vgpr'] <= regs and k.get('spill', 0) <= spill_ok, (name, k)
