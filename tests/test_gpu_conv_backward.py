"""Convolution backward building blocks against torch autograd of torch.nn.functional.conv2d (fp32, CPU)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(N, H, W, cin, cout, k, stride, pad, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, w, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    return x.detach(), w.detach(), dy, x.grad, w.grad


# ResNet / FPN / head shapes in miniature: 3x3 s1 p1, 1x1 s1, 3x3 s2 p1 (bottleneck conv2 of a down-sampling block),
# 1x1 s2 (its shortcut), 2x2 p0 and 7x7 s1 p3 for generality; channel counts not multiples of 32, ragged pixel counts
@pytest.mark.parametrize('N,H,W,cin,cout,k,stride,pad', [(2, 16, 16, 8, 12, 3, 1, 1), (3, 10, 14, 16, 8, 1, 1, 0), (2, 16, 12, 8, 16, 3, 2, 1),
                                                         (2, 8, 8, 12, 20, 1, 2, 0), (1, 9, 9, 4, 8, 2, 1, 0), (1, 12, 12, 4, 8, 7, 1, 3),
                                                         (4, 32, 32, 64, 64, 3, 1, 1), (2, 32, 32, 128, 64, 3, 2, 1)])
def test_dgrad_and_wgrad_match_autograd(N, H, W, cin, cout, k, stride, pad):
    from vpho_amd import conv_backward as CB
    from vpho_amd.model.pack import pack_conv
    x, w, dy, dx_ref, dw_ref = _case(N, H, W, cin, cout, k, stride, pad, seed=N * 1000 + H * 10 + k)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyg = dy.permute(0, 2, 3, 1).contiguous().cuda()
    wp = pack_conv(w).cuda()
    dx = CB.conv2d_dgrad(dyg, wp, (H, W), k, k, stride, pad).permute(0, 3, 1, 2).cpu()
    dw = CB.conv2d_wgrad(xg, dyg, k, k, stride, pad).cpu().view(cout, k, k, cin).permute(0, 3, 1, 2)
    np.testing.assert_allclose(dx.numpy(), dx_ref.numpy(), atol=2e-5 * float(dx_ref.abs().max()), rtol=1e-4)
    np.testing.assert_allclose(dw.numpy(), dw_ref.numpy(), atol=2e-5 * float(dw_ref.abs().max()), rtol=1e-4)
    db = CB.conv2d_bias_grad(dyg).cpu()
    np.testing.assert_allclose(db.numpy(), dy.sum(dim=(0, 2, 3)).numpy(), atol=1e-4 * float(dy.abs().sum(dim=(0, 2, 3)).max()), rtol=1e-4)
