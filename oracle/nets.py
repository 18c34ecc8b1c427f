"""Functional torch-CPU restatement of the network blocks on the hot path.  Every function takes the model
``state_dict`` (reference key layout) and a key prefix.  Eval-mode semantics only.
TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import rotations as R
from .rk45 import solve_rk45


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, eps)


def _conv(sd, p, x, stride=1, pad=0):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, pad)


def _lin(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd[p + '.bias'])


# ---------------------------------------------------------------- backbone_FPN_HFL.py:70-109, 308-350
def bottleneck(sd, p, x, stride):
    out = F.leaky_relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x)), 0.01)
    out = F.leaky_relu(_bn(sd, p + '.bn2', _conv(sd, p + '.conv2', out, stride, 1)), 0.01)
    out = _bn(sd, p + '.bn3', _conv(sd, p + '.conv3', out))
    res = x
    if (p + '.downsample.0.weight') in sd:
        res = _bn(sd, p + '.downsample.1', _conv(sd, p + '.downsample.0', x, stride))
    return F.leaky_relu(out + res, 0.01)


def res_layer(sd, p, x, blocks, stride):
    for i in range(blocks):
        x = bottleneck(sd, f'{p}.0.{i}', x, stride if i == 0 else 1)
    return x


def fpn(sd, p, x):
    c1 = F.max_pool2d(F.leaky_relu(_bn(sd, p + '.layer0_h.1', _conv(sd, p + '.layer0_h.0', x, 2, 3)), 0.01), 3, 2, 1)
    c2 = res_layer(sd, p + '.layer1_h', c1, 3, 1)
    c3h, c3o = res_layer(sd, p + '.layer2_h', c2, 4, 2), res_layer(sd, p + '.layer2_o', c2, 4, 2)
    c4h, c4o = res_layer(sd, p + '.layer3_h', c3h, 6, 2), res_layer(sd, p + '.layer3_o', c3o, 6, 2)
    c5h, c5o = res_layer(sd, p + '.layer4_h', c4h, 3, 2), res_layer(sd, p + '.layer4_h', c4o, 3, 2)  # quirk Q6

    def up_add(a, b):
        return F.interpolate(a, size=b.shape[-2:], mode='bilinear', align_corners=False) + b

    out = []
    for br, c5, c4, c3 in (('h', c5h, c4h, c3h), ('o', c5o, c4o, c3o)):
        p5 = _conv(sd, f'{p}.toplayer_{br}', c5)
        p4 = up_add(p5, _conv(sd, f'{p}.latlayer1_{br}', c4))
        p3 = up_add(p4, _conv(sd, f'{p}.latlayer2_{br}', c3))
        p2 = up_add(p3, _conv(sd, f'{p}.latlayer3_{br}', c2))
        out.append(_conv(sd, f'{p}.smooth3_{br}', p2, 1, 1))
    return out[0], out[1]


# ---------------------------------------------------------------- head_inplane.py:102-107 (quirk Q1)
def head_heatmap2(sd, p, x):
    x = _conv(sd, p + '.conv_layers.0', x, 1, 1)
    x = _bn(sd, p + '.conv_layers.2', _conv(sd, p + '.conv_layers.1', x, 1, 1))      # LeakyReLU(slope 1.0) = identity
    x = F.conv_transpose2d(x, sd[p + '.deconv_layers.0.weight'], None, 2, 1, 0)
    x = F.relu(_bn(sd, p + '.deconv_layers.1', x))
    return _conv(sd, p + '.final_layer', x)


# ---------------------------------------------------------------- encoding.py:21-36, 58-73
def residual(sd, p, x):
    out = F.leaky_relu(_bn(sd, p + '.bn', x), 0.01)
    out = F.leaky_relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', out)), 0.01)
    out = F.leaky_relu(_bn(sd, p + '.bn2', _conv(sd, p + '.conv2', out, 1, 1)), 0.01)
    return _conv(sd, p + '.conv3', out) + x


def encoder(sd, p, x):
    x = _conv(sd, p + '.project', x)
    stages = []
    for i in range(4):
        for j in range(2):
            x = residual(sd, f'{p}.reg.{i * 2 + j}', x)
        x = F.max_pool2d(x, 2, 2)
        stages.append(x)
    return x.flatten(1), stages


# ---------------------------------------------------------------- head_mano.py:61-76
def head_mano(sd, p, x):
    h = F.leaky_relu(_lin(sd, p + '.base_layer.0', x), 0.01)
    h = F.leaky_relu(_lin(sd, p + '.base_layer.2', h), 0.01)
    r6 = _lin(sd, p + '.fc_pose', h).reshape(x.shape[0], -1, 6)
    aa = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(r6)).reshape(x.shape[0], -1)
    return aa, _lin(sd, p + '.fc_shape', h)


# ---------------------------------------------------------------- cross_module.py:18-46, 120-137 (quirk Q3)
def pos_embed_nerf(x, multires=10):
    outs = [x]
    for f in (2.0 ** torch.linspace(0.0, multires - 1, steps=multires)):
        outs += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(outs, -1)


def transformer_layer(sd, p, x, nhead=2, eps=1e-5):
    """Post-norm nn.TransformerEncoderLayer, eval mode, ReLU, x: (S, B, E) (batch_first=False)."""
    S, B, E = x.shape
    hd = E // nhead
    qkv = F.linear(x, sd[p + '.self_attn.in_proj_weight'], sd[p + '.self_attn.in_proj_bias'])
    q, k, v = qkv.split(E, dim=-1)
    sh = lambda t: t.reshape(S, B * nhead, hd).transpose(0, 1)          # (B*h, S, hd)
    q, k, v = sh(q), sh(k), sh(v)
    att = torch.softmax(torch.bmm(q * (1.0 / math.sqrt(hd)), k.transpose(1, 2)), dim=-1)
    o = torch.bmm(att, v).transpose(0, 1).reshape(S, B, E)
    o = _lin(sd, p + '.self_attn.out_proj', o)
    x = F.layer_norm(x + o, (E,), sd[p + '.norm1.weight'], sd[p + '.norm1.bias'], eps)
    ff = _lin(sd, p + '.linear2', F.relu(_lin(sd, p + '.linear1', x)))
    return F.layer_norm(x + ff, (E,), sd[p + '.norm2.weight'], sd[p + '.norm2.bias'], eps)


def cross_module(sd, p, x_hand, x_obj, gravity):
    bs = x_hand.shape[0]
    xh = _conv(sd, p + '.proj_hand', x_hand, 1, 1).reshape(bs, 32, -1)
    xo = _conv(sd, p + '.proj_obj', x_obj, 1, 1).reshape(bs, 32, -1)
    g = _lin(sd, p + '.gravity_proj', pos_embed_nerf(gravity))
    x = torch.cat([xh, xo, g], dim=1)                                   # (bs, 65, 512) fed as (S=bs, B=65, E)
    x = x + sd[p + '.pose_embedder.pe'][:bs]
    x = transformer_layer(sd, p + '.attn.layers.0', x)
    return x[:, :32], x[:, 32:64], x[:, 64:]


# ---------------------------------------------------------------- physics.py:546-557, 700-721 (quirk Q4)
def head_physics(sd, p, x_hand, x_obj, friction=0.8):
    mlp = lambda q, x: _lin(sd, f'{p}.{q}.2', F.leaky_relu(_lin(sd, f'{p}.{q}.0', x), 0.01))
    scale = mlp('fc_scale', x_hand).squeeze(-1).abs()
    weight = torch.softmax(torch.softmax(mlp('fc_weight', x_obj), -1), -1)
    anchor = sd[p + '.anchor'].clone()
    anchor[:, :2] *= friction
    d = torch.einsum('...ij,jk->...ik', weight, anchor)
    d = d / (d.norm(dim=-1, keepdim=True) + 1e-8)
    return d * scale[..., None]


# ---------------------------------------------------------------- denoiser.py:29-31, 68-82; sde.py:15-28
SIGMA_MIN, SIGMA_MAX, EPS_T = 0.01, 50.0, 1e-5


def denoiser(sd, p, feat, x, t):
    """feat (R,1024), x (R,D) f32, t (R,1) f32 -> score (R,D)."""
    W = sd[p + '.t_encoder.0.W']
    xp = t.squeeze(1)[:, None] * W[None, :] * 2 * np.pi
    tf = F.relu(_lin(sd, p + '.t_encoder.1', torch.cat([torch.sin(xp), torch.cos(xp)], -1)))
    pf = F.relu(_lin(sd, p + '.pose_encoder.2', F.relu(_lin(sd, p + '.pose_encoder.0', x))))
    tot = torch.cat([tf, pf, feat], -1)
    h = F.relu(torch.einsum('bc,ncd->bnd', tot, sd[p + '.head.head.0.weight']) + sd[p + '.head.head.0.bias'])
    o = torch.einsum('bnc,ncd->bnd', h, sd[p + '.head.head.2.weight']) + sd[p + '.head.head.2.bias']
    std = SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t
    return o.reshape(x.shape[0], -1) / (std + 1e-7)


def ve_diffusion(t):
    """sde.py:20-24 with the same dtype behaviour (t tensor f32)."""
    sigma = SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t
    # torch.tensor(np.float64) is a float64 0-d tensor: a 0-d f32 ``t`` promotes to f64, a dimensioned f32 ``t`` stays f32
    return sigma * torch.sqrt(torch.tensor(np.float64(2 * (np.log(SIGMA_MAX) - np.log(SIGMA_MIN)))))


def ode_sample(sd, p, feat, init_x, T0, num_steps, rtol=3e-3, atol=3e-4):
    """score_based_model.py:45-105.  init_x (R,D) f32 (already the prior draw).  Returns xs (R,steps,D) f64,
    x (R,D) f64, info dict (nfev incl. the denoise call, step log)."""
    Rr, D = init_x.shape
    t_calls = []

    def fun(t, y):
        t_calls.append(float(np.float32(t)))
        x = torch.tensor(y.reshape(-1, D)).float()
        ts = torch.ones(Rr).unsqueeze(-1) * t
        g = ve_diffusion(torch.tensor(t)).numpy()
        s = denoiser(sd, p, feat, x, ts)
        s = torch.nan_to_num(s, nan=0.0, posinf=0.0, neginf=0.0) if torch.isnan(s).any() else s
        # numpy 1.26 (reference environment.yaml:72) value-based casting: the f64 0-d coefficient is cast to f32
        # before it multiplies the f32 score, so every RK stage holds f32 values
        c = np.float32(0.5 * (g ** 2))
        return 0 - c * s.numpy().reshape(-1)

    t_eval = np.linspace(T0, EPS_T, num_steps)
    res = solve_rk45(fun, T0, EPS_T, init_x.reshape(-1).numpy(), rtol, atol, 10, t_eval)
    xs = torch.tensor(res['y']).T.view(-1, Rr, D)
    x = torch.tensor(res['y'][:, -1]).reshape(Rr, D)
    vec = torch.ones((Rr, 1)) * EPS_T
    g = ve_diffusion(vec)
    grad = denoiser(sd, p, feat, x.float(), vec)
    x = x + (0 - g ** 2 * grad) * ((1 - EPS_T) / num_steps)
    t_calls.append(float(np.float32(EPS_T)))
    return xs.permute(1, 0, 2), x, dict(nfev=res['nfev'] + 1, steps=res['steps'], t_calls=t_calls)


def ve_prior_sigma(T0):
    return SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** T0
