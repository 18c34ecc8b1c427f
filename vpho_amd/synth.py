"""Seeded synthetic weights and input batches for benchmarks, smoke and parity tests.

The reference's own ``init_weights`` (VPHO.py:34-45) zero-initialises the score heads (score == 0), which would not
exercise the sampler, so synthetic runs use Kaiming-scaled weights with non-zero score output layers (sigma 0.02) and
randomised BatchNorm statistics.  Every tensor is generated from ``crc32(key) ^ seed`` so the values do not depend on
module construction order.
"""
import zlib
import numpy as np
import torch

from .assets import YCB_NAMES


def _rng(key, seed):
    return np.random.default_rng((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)


HM_GAIN_FLAT = 0.002       # round-1 fixtures: heat-maps 0.5 +- 0.008 (nearly flat: candidate scores differ in the 4th digit)
HM_GAIN_CONTRAST = 0.7     # heat-maps 0.5 +- 0.15: candidate scores spread like a trained head's peaked maps do


def condition_denoisers(sd, seed=0, c_hand=1.0, c_obj=1.5, noise_hand=0.005, noise_obj=0.003):
    """Give the seeded score networks the gross behaviour of trained ones: hypotheses that contract towards a mode.

    With purely random weights the probability-flow ODE (score_based_model.py:45-105) leaves the prior N(0, sigma(T0)^2) about
    where it started: at the README's T0=0.65 (sigma = 2.5) object translations are metres away from the crop, every heat-map
    score is exactly 0 and the reference's own result hangs on torch.topk's unspecified order among equal values.  A trained
    network pulls samples towards the data.  Here that pull is written into the ReLU MLP exactly (denoiser.py:34-82,166-179,
    234-247): pose_encoder carries x as (relu(x), relu(-x)) through both of its layers, six hidden units of every
    ParallelLinear head rebuild +-(x - mu) and the output layer returns -c (x - mu) before the division by sigma(t), so
    dx/dt = sigma(t) ln(sigma_max / sigma_min) c (x - mu): a contraction by exp(-c (sigma(T0) - sigma(eps))) = 0.08 (hand,
    c = 1) / 0.02 (object, c = 1.5) around mu.  All other units keep their seeded random weights (output weights scaled to
    `noise_*`), which displace the mode per image and per hypothesis.  mu: rot6d of the identity + seeded spread; object
    translation near the hand root (root-relative, head_object.py:36-61).  Returns sd (modified in place)."""
    for name, D, n, c, ns in (('denoiser_hand', 96, 32, c_hand, noise_hand), ('denoiser_obj', 9, 3, c_obj, noise_obj)):
        r = _rng(name + '.mu', seed)
        mu = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), D // 6 if D == 96 else 1)
        if D == 96:
            mu = mu + r.normal(0, 0.15, 96).astype(np.float32)
        else:
            mu = np.concatenate([mu + r.normal(0, 0.3, 6).astype(np.float32), r.normal(0, 0.02, 3).astype(np.float32)])
        w0, b0 = sd[f'{name}.pose_encoder.0.weight'], sd[f'{name}.pose_encoder.0.bias']            # (256,D), (256)
        w0[:2 * D] = 0
        b0[:2 * D] = 0
        w0[:D] = torch.eye(D)
        w0[D:2 * D] = -torch.eye(D)
        w2, b2 = sd[f'{name}.pose_encoder.2.weight'], sd[f'{name}.pose_encoder.2.bias']            # (256,256)
        w2[:2 * D] = 0
        b2[:2 * D] = 0
        w2[:2 * D, :2 * D] = torch.eye(2 * D)
        h0, hb0 = sd[f'{name}.head.head.0.weight'], sd[f'{name}.head.head.0.bias']                 # (n,1408,256), (n,256)
        h2, hb2 = sd[f'{name}.head.head.2.weight'], sd[f'{name}.head.head.2.bias']                 # (n,256,3), (n,3)
        h2 *= ns / 0.02
        hb2 *= 0
        for i in range(n):
            h0[i, :, :6] = 0
            h2[i, :6, :] = 0
            for j in range(3):
                d = 3 * i + j
                h0[i, 128 + d, j], h0[i, 128 + D + d, j], hb0[i, j] = 1.0, -1.0, -float(mu[d])
                h0[i, 128 + d, 3 + j], h0[i, 128 + D + d, 3 + j], hb0[i, 3 + j] = -1.0, 1.0, float(mu[d])
                h2[i, j, j], h2[i, 3 + j, j] = -c, c
    return sd


def synth_state_dict(model, seed=0, hm_gain=HM_GAIN_FLAT, conditioned=False):
    """Return a new state_dict for ``model`` (a vpho_net or any sub-module) with seeded values.

    ``hm_gain``: Kaiming gain of the heat-map heads' final 1x1 layer = the spatial contrast of the synthetic heat-maps.  The
    aggregation ranks hypotheses by bicubic heat-map samples (aggregation.py:206-218); on a nearly flat map (HM_GAIN_FLAT)
    the 200 candidates of an image score within 1e-4 of each other and one rank in a thousand is decided by the last bit of
    the fp32 sums -- on any platform, the reference's own included.  HM_GAIN_CONTRAST gives the scores the spread of a
    trained head (Gaussian peaks on a zero background) and is what the README-size parity tests and bench.py use; the small
    golden fixtures of round 1 keep the flat maps they were generated with.
    ``conditioned``: see ``condition_denoisers`` (score networks whose hypotheses cluster like a trained model's)."""
    sd = model.state_dict()
    out = {}
    for k, v in sd.items():
        r = _rng(k, seed)
        shp = tuple(v.shape)
        if k.startswith(('head_obj.', 'head_mano.mano_layer.')) or k.endswith(('pose_embedder.pe', 'head_physics.anchor', '.anchor', '.pe')):
            out[k] = v.clone()
            continue
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros_like(v)
        elif k.endswith('running_mean'):
            out[k] = torch.from_numpy(r.normal(0, 0.1, shp).astype(np.float32))
        elif k.endswith('running_var'):
            out[k] = torch.from_numpy(r.uniform(0.5, 1.5, shp).astype(np.float32))
        elif k.endswith('t_encoder.0.W'):
            out[k] = torch.from_numpy((r.normal(0, 1, shp) * 30.0).astype(np.float32))
        elif ('.bn' in k or '.norm' in k or 'downsample.1' in k or k.endswith(('layer0_h.1.weight', 'layer0_h.1.bias'))
              or 'conv_layers.2.' in k or 'deconv_layers.1.' in k) and len(shp) == 1:
            if k.endswith('weight'):
                lo, hi = (0.15, 0.35) if ('.bn3.' in k) else ((0.4, 0.7) if 'downsample.1' in k else (0.7, 1.3))
                out[k] = torch.from_numpy(r.uniform(lo, hi, shp).astype(np.float32))
            else:
                out[k] = torch.from_numpy(r.normal(0, 0.1, shp).astype(np.float32))
        elif k.endswith('final_layer.bias'):                  # positive heat-maps (well-conditioned top-k weights)
            out[k] = torch.from_numpy(r.uniform(0.4, 0.6, shp).astype(np.float32))
        elif k.endswith('bias'):
            out[k] = torch.from_numpy(r.normal(0, 0.05, shp).astype(np.float32))
        elif 'head.head.2.weight' in k:
            out[k] = torch.from_numpy(r.normal(0, 0.02, shp).astype(np.float32))
        elif 'head.head.0.weight' in k:                       # ParallelLinear (n, in, out)
            out[k] = torch.from_numpy(r.normal(0, np.sqrt(2.0 / shp[1]), shp).astype(np.float32))
        elif len(shp) == 4 and 'deconv' in k:                 # ConvTranspose (cin, cout, kh, kw)
            out[k] = torch.from_numpy(r.normal(0, np.sqrt(2.0 / (shp[0] * 4)), shp).astype(np.float32))
        elif len(shp) >= 2:                                   # conv (cout,cin,kh,kw) / linear (out,in)
            fan_in = int(np.prod(shp[1:]))
            gain = 2.0
            if '.conv3.' in k and k.startswith('encoder_'):   # un-normalised residual branch of encoding.Residual
                gain = 0.1
            elif k.startswith(('feature_extractor.toplayer', 'feature_extractor.latlayer', 'feature_extractor.smooth')):
                gain = 0.02
            elif k.startswith('encoder_') and '.project.' in k:
                gain = 0.25
            elif 'final_layer' in k:
                gain = hm_gain
            elif 'fc_shape' in k or 'fc_pose' in k:
                gain = 0.5
            out[k] = torch.from_numpy(r.normal(0, np.sqrt(gain / fan_in), shp).astype(np.float32))
        else:
            out[k] = torch.from_numpy(r.normal(0, 0.1, shp).astype(np.float32))
    if conditioned and 'denoiser_hand.pose_encoder.0.weight' in out:
        condition_denoisers(out, seed)
    return out


def bench_state_dict(model, seed=1):
    """The weights of bench.py and of the README-size parity tests: high-contrast heat-maps + conditioned score networks."""
    return synth_state_dict(model, seed=seed, hm_gain=HM_GAIN_CONTRAST, conditioned=True)


def synth_batch(bs, assets, seed=206, rank=0, patch=256):
    """Synthetic batch dict with the schema ``vpho_net.forward`` consumes."""
    r = np.random.default_rng(seed + 1000003 * rank)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    root = np.stack([r.uniform(-0.1, 0.1, bs), r.uniform(-0.1, 0.1, bs), r.uniform(0.6, 0.9, bs)], -1)
    foc = r.uniform(400, 600, bs) * 1.2
    Kmat = np.zeros((bs, 3, 3))
    Kmat[:, 0, 0] = foc
    Kmat[:, 1, 1] = foc
    Kmat[:, 2, 2] = 1
    # principal point such that the hand root projects near the crop centre
    Kmat[:, 0, 2] = patch / 2 - foc * root[:, 0] / root[:, 2]
    Kmat[:, 1, 2] = patch / 2 - foc * root[:, 1] / root[:, 2]
    half = foc * 0.11 / root[:, 2]                                    # ~hand half-extent in pixels
    ctr = patch / 2 + r.uniform(-6, 6, size=(bs, 2))
    hw = half[:, None] * r.uniform(0.75, 1.15, size=(bs, 2))
    bbox_hand = np.clip(np.concatenate([ctr - hw, ctr + hw], -1), 0, patch)
    octr = ctr + r.uniform(-25, 25, size=(bs, 2))
    ohw = half[:, None] * r.uniform(0.6, 1.1, size=(bs, 2))
    bbox_obj = np.clip(np.concatenate([octr - ohw, octr + ohw], -1), 0, patch)

    def rect(b):
        c = (b[:, :2] + b[:, 2:]) / 2
        m = (b[:, 2:] - b[:, :2]).max(-1, keepdims=True)
        return np.concatenate([c - m / 2, c + m / 2], -1)

    is_right = r.random(bs) < 0.5
    g = r.normal(size=(bs, 1, 3))
    g /= np.linalg.norm(g, axis=-1, keepdims=True)
    root_unflip = root.copy()
    root_unflip[~is_right, 0] *= -1
    batch = {
        'rgb': f32(r.normal(size=(bs, 3, patch, patch))),
        'bbox_hand': f32(bbox_hand), 'bbox_obj': f32(bbox_obj),
        'bbox_hand_rect': f32(rect(bbox_hand)), 'bbox_obj_rect': f32(rect(bbox_obj)),
        'is_right': torch.from_numpy(is_right), 'is_ho3d': torch.zeros(bs, dtype=torch.bool),
        'is_grasped': torch.from_numpy(r.random(bs) < 0.8),
        'gravity': f32(g), 'obj_CoM': f32(r.normal(0, 0.05, size=(bs, 1, 3))),
        'root_joint_flip': f32(root), 'root_joint': f32(root_unflip),
        'cam_intr_crop_flip': f32(Kmat),
        'obj_name': [YCB_NAMES[i] for i in r.integers(0, len(YCB_NAMES), bs)],
    }
    # evaluation-only ground truth, drawn after everything the model consumes (keeps those draws unchanged):
    # object pose [R | t] in camera space near the hand root, and the camera matrix TesterObject projects with
    q = r.normal(size=(bs, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q.T
    Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                   2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                   2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(bs, 3, 3)
    t = root_unflip + r.normal(0, 0.03, size=(bs, 3))
    batch['gt_obj_rt'] = f32(np.concatenate([Rm, t[:, :, None]], -1))
    batch['cam_intr'] = f32(Kmat)
    return batch
