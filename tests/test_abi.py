"""CPU-side checks: the C-ABI library loads and exports every symbol include/vpho_hip.h declares; host-side packing."""
import ctypes
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'vpho_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vpho_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, 'vpho_amd', 'libvpho_hip.so'))
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/vpho_hip.h but not exported'
    lib.vpho_abi_version.restype = ctypes.c_int
    assert lib.vpho_abi_version() == 7


def test_state_dict_layout_matches_reference_contract(model_cpu):
    sd = model_cpu.state_dict()
    assert len([k for k in sd if k.startswith('feature_extractor.')]) == 530
    assert len([k for k in sd if k.startswith('encoder_hand.')]) == 170
    assert sd['denoiser_hand.head.head.0.weight'].shape == (32, 1408, 256)
    assert sd['denoiser_obj.head.head.2.weight'].shape == (3, 256, 3)
    assert sd['head_hm_hand.deconv_layers.0.weight'].shape == (128, 64, 4, 4)
    assert sd['cross_hand.pose_embedder.pe'].shape == (5000, 1, 512)
    assert sd['cross_obj.attn.layers.0.self_attn.in_proj_weight'].shape == (1536, 512)
    assert sd['head_physics.anchor'].shape == (8, 3)
    assert 'head_obj.point_002_master_chef_can' in sd and 'head_mano.mano_layer.th_posedirs' in sd


def test_deconv_phase_packing_is_equivalent_on_cpu():
    """pack_deconv4x4s2 (host logic) reproduces ConvTranspose2d(k4,s2,p1) with four stride-1 2x2 convolutions."""
    from vpho_amd.model.pack import pack_deconv4x4s2
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(1, 6, 5, 5, generator=g), torch.randn(6, 4, 4, 4, generator=g)
    ref = F.conv_transpose2d(x, w, None, 2, 1)
    out = torch.zeros_like(ref)
    for (py, px), (wp, pady, padx) in pack_deconv4x4s2(w).items():
        wk = wp.view(4, 2, 2, 6).permute(0, 3, 1, 2)
        xp = F.pad(x, (padx, 1 - padx, pady, 1 - pady))
        out[:, :, py::2, px::2] = F.conv2d(xp, wk)
    assert torch.allclose(out, ref, atol=1e-5)


def test_bn_folding_on_cpu(sd):
    from vpho_amd.model.pack import fold_conv_bn
    p = 'feature_extractor.layer1_h.0.0'
    w, b = fold_conv_bn(sd, p + '.conv1', p + '.bn1')
    x = torch.randn(2, 64, 5, 5)
    ref = F.batch_norm(F.conv2d(x, sd[p + '.conv1.weight']), sd[p + '.bn1.running_mean'], sd[p + '.bn1.running_var'],
                       sd[p + '.bn1.weight'], sd[p + '.bn1.bias'], False, 0.0, 1e-5)
    got = F.conv2d(x, w.view(64, 1, 1, 64).permute(0, 3, 1, 2), b)
    assert torch.allclose(got, ref, atol=1e-5)
