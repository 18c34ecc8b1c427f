"""Parameter containers for ``vpho_net`` -- every sub-module attribute name and tensor shape follows the reference's
checkpoint layout so that ``accelerate.load_state`` / ``load_state_dict`` of a reference
checkpoint keeps working.  These modules only HOLD parameters; their arithmetic runs in the HIP kernels
(``vpho_amd/model/VPHO.py`` packs the tensors for the kernels).  Calling ``forward`` on a container is an error.

Reference definitions: backbone_FPN_HFL.py:20-68,202-350; head_inplane.py:42-100; encoding.py:5-56;
denoiser.py:20-66,166-179,234-247; parallel_linear.py:9-25; head_mano.py:29-59; head_object.py:9-33;
cross_module.py:47-118; physics.py:648-698.
"""
import math
import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} only holds parameters; the arithmetic runs in vpho_amd HIP kernels')


def _conv(cin, cout, k, stride=1, pad=0, bias=True):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=bias)


class Bottleneck(_Holder):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, with_downsample=False):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.stride = stride
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(_conv(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))


def _res_layer(inplanes, planes, blocks, stride):
    layers = [Bottleneck(inplanes, planes, stride, with_downsample=(stride != 1 or inplanes != planes * 4))]
    layers += [Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    return nn.Sequential(nn.Sequential(*layers))      # -> "layerN_x.0.<blk>." keys


class FPN(_Holder):
    """Two-branch ResNet-50 + FPN (shared layer0/1/4, per-branch layer2/3)."""

    def __init__(self):
        super().__init__()
        self.toplayer_h, self.toplayer_o = _conv(2048, 256, 1), _conv(2048, 256, 1)
        self.layer0_h = nn.Sequential(_conv(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64),
                                      nn.LeakyReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        self.layer1_h = _res_layer(64, 64, 3, 1)
        self.layer2_h = _res_layer(256, 128, 4, 2)
        self.layer3_h = _res_layer(512, 256, 6, 2)
        self.layer4_h = _res_layer(1024, 512, 3, 2)
        self.layer2_o = _res_layer(256, 128, 4, 2)
        self.layer3_o = _res_layer(512, 256, 6, 2)
        self.smooth3_h, self.smooth3_o = _conv(256, 256, 3, 1, 1), _conv(256, 256, 3, 1, 1)
        self.latlayer1_h, self.latlayer2_h, self.latlayer3_h = _conv(1024, 256, 1), _conv(512, 256, 1), _conv(256, 256, 1)
        self.latlayer1_o, self.latlayer2_o, self.latlayer3_o = _conv(1024, 256, 1), _conv(512, 256, 1), _conv(256, 256, 1)


class HeadHeatmap2(_Holder):
    def __init__(self, in_dim, out_dim, hidden_dim=256):
        super().__init__()
        # conv (no BN/act) ; conv + BN + LeakyReLU(negative_slope=True == 1.0 -> identity, quirk Q1)
        self.conv_layers = nn.Sequential(_conv(in_dim, hidden_dim, 3, 1, 1), _conv(hidden_dim, hidden_dim, 3, 1, 1),
                                         nn.BatchNorm2d(hidden_dim), nn.LeakyReLU(True))
        self.deconv_layers = nn.Sequential(
            nn.ConvTranspose2d(hidden_dim, hidden_dim // 2, 4, 2, 1, 0, bias=False),
            nn.BatchNorm2d(hidden_dim // 2), nn.ReLU(inplace=True))
        self.final_layer = _conv(hidden_dim // 2, out_dim, 1)


class Residual(_Holder):
    def __init__(self, n_in, n_out):
        super().__init__()
        assert n_in == n_out
        self.bn = nn.BatchNorm2d(n_in)
        self.conv1, self.bn1 = _conv(n_in, n_out // 2, 1), nn.BatchNorm2d(n_out // 2)
        self.conv2, self.bn2 = _conv(n_out // 2, n_out // 2, 3, 1, 1), nn.BatchNorm2d(n_out // 2)
        self.conv3 = _conv(n_out // 2, n_out, 1)


class Encoder(_Holder):
    def __init__(self, in_dim, hid_dim, n_block=4, n_module=2):
        super().__init__()
        self.project = _conv(in_dim, hid_dim, 1)
        self.reg = nn.ModuleList([Residual(hid_dim, hid_dim) for _ in range(n_block * n_module)])
        self.n_block, self.n_module = n_block, n_module


class ParallelLinear(_Holder):
    def __init__(self, in_features, out_features, num):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num, in_features, out_features))
        self.bias = nn.Parameter(torch.empty(num, out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1 / math.sqrt(in_features), 1 / math.sqrt(in_features))


class GaussianFourierProjection(_Holder):
    def __init__(self, embed_dim, scale=30.):
        super().__init__()
        self.register_buffer('W', torch.randn(embed_dim // 2) * scale)


class _ScoreHead(_Holder):
    def __init__(self, n):
        super().__init__()
        self.head = nn.Sequential(ParallelLinear(128 + 256 + 1024, 256, n), nn.ReLU(True), ParallelLinear(256, 3, n))
        for p in self.head[2].parameters():
            p.detach().zero_()


class BaseDenoiser(_Holder):
    def __init__(self, head='mano_pose'):
        super().__init__()
        assert head in ('mano_pose', 'obj')
        n = 32 if head == 'mano_pose' else 3
        self.out_dim = 3 * n
        self.t_encoder = nn.Sequential(GaussianFourierProjection(128), nn.Linear(128, 128), nn.ReLU(True))
        self.head = _ScoreHead(n)
        self.pose_encoder = nn.Sequential(nn.Linear(self.out_dim, 256), nn.ReLU(True), nn.Linear(256, 256), nn.ReLU(True))


class ManoLayer(_Holder):
    """Buffers of manopth.ManoLayer (names kept for checkpoint compatibility)."""

    def __init__(self, mano):
        super().__init__()
        t = lambda a: torch.as_tensor(a, dtype=torch.float32)
        self.register_buffer('th_betas', torch.zeros(1, 10))
        self.register_buffer('th_shapedirs', t(mano['shapedirs']))
        self.register_buffer('th_posedirs', t(mano['posedirs']))
        self.register_buffer('th_v_template', t(mano['v_template'])[None])
        self.register_buffer('th_J_regressor', t(mano['J_regressor']))
        self.register_buffer('th_weights', t(mano['weights']))
        self.register_buffer('th_faces', torch.zeros(1538, 3, dtype=torch.long))
        self.register_buffer('th_hands_mean', torch.zeros(1, 45))
        self.register_buffer('th_comps', torch.eye(45))
        self.register_buffer('th_selected_comps', torch.eye(45))


class HeadMano(_Holder):
    def __init__(self, mano, in_dim=1024):
        super().__init__()
        self.base_layer = nn.Sequential(nn.Linear(in_dim, 1024), nn.LeakyReLU(inplace=True),
                                        nn.Linear(1024, 512), nn.LeakyReLU(inplace=True))
        self.fc_pose = nn.Linear(512, 96)
        self.fc_shape = nn.Linear(512, 10)
        self.mano_layer = ManoLayer(mano)


class HeadObject(_Holder):
    def __init__(self, ycb):
        super().__init__()
        self.names = list(ycb.keys())
        for k, v in ycb.items():
            self.register_buffer(f'point_{k}', torch.as_tensor(v['kpt3d'], dtype=torch.float32))
            self.register_buffer(f'vert_{k}', torch.as_tensor(v['verts_sampled'], dtype=torch.float32))
            self.register_buffer(f'CoM_{k}', torch.as_tensor(v['CoM'], dtype=torch.float32)[None])
            self.register_buffer(f'vert_full_{k}', torch.as_tensor(v['verts'], dtype=torch.float32))


class PositionalEncoding(_Holder):
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        self.register_buffer('pe', pe.unsqueeze(1))


class CrossModule(_Holder):
    def __init__(self, in_hw=8, hid_dim=512):
        super().__init__()
        proj_dim = int(hid_dim / (in_hw ** 2 / 32))
        self.proj_hand, self.proj_obj = _conv(256, proj_dim, 3, 1, 1), _conv(256, proj_dim, 3, 1, 1)
        self.gravity_proj = nn.Linear(63, hid_dim)
        self.pose_embedder = PositionalEncoding(hid_dim)
        self.attn = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=hid_dim, nhead=2), num_layers=1,
                                          enable_nested_tensor=False)


class HeadPhysics(_Holder):
    def __init__(self, hid_dim=512):
        super().__init__()
        mlp = lambda o, *tail: nn.Sequential(nn.Linear(hid_dim, hid_dim), nn.LeakyReLU(), nn.Linear(hid_dim, o), *tail)
        self.fc_scale = mlp(1)
        self.fc_weight = mlp(8, nn.Softmax(dim=-1))
        self.fc_CoM = mlp(3)
        a = torch.arange(0, 2 * torch.pi, 2 * torch.pi / 8)[:8]
        self.register_buffer('anchor', torch.stack([torch.cos(a), torch.sin(a), torch.ones_like(a)], dim=-1) / 8)
