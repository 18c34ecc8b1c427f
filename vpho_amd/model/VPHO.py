"""``vpho_net`` -- drop-in for the reference's lib/model/VPHO.py:48-304 (same constructor contract: no arguments, reads
the module-global ``cfg``; same ``forward(data, mode)`` contract and output dict; same checkpoint key layout).
The arithmetic of ``forward(mode='predict')`` runs in the hand-written HIP kernels of ``vpho_amd/csrc`` through the
C-ABI in ``include/vpho_hip.h``; there is no CPU / eager fallback.
"""
import torch
import torch.nn as nn

from ..configs.args import cfg
from ..assets import load_assets, ANCHOR_SKELETON
from . import layers as L


class vpho_net(nn.Module):
    def __init__(self, assets=None):
        super().__init__()
        self.cfg = cfg
        self.assets = assets if assets is not None else load_assets(getattr(cfg, 'asset_root', 'asset'))
        self.anchor_skeleton = ANCHOR_SKELETON
        self.feature_extractor = L.FPN()
        self.denoiser_hand = L.BaseDenoiser(head='mano_pose')
        self.denoiser_obj = L.BaseDenoiser(head='obj')
        self.head_hm_hand = L.HeadHeatmap2(256, 21, 128)
        self.head_hm_obj = L.HeadHeatmap2(256, 27, 128)
        self.encoder_hand = L.Encoder(256 + 21, 256)
        self.encoder_obj = L.Encoder(256 + 27, 256)
        self.head_mano = L.HeadMano(self.assets['mano'], in_dim=1024)
        self.head_obj = L.HeadObject(self.assets['ycb'])
        self.cross_hand = L.CrossModule(8, 512)
        self.cross_obj = L.CrossModule(8, 512)
        self.head_physics = L.HeadPhysics(hid_dim=512)
        self._engine = None

    def forward(self, data, mode='predict'):
        assert mode in ['train', 'score', 'sample', 'predict']
        if mode != 'predict':
            raise NotImplementedError("vpho_amd implements the inference hot path (mode='predict') only")
        from .engine import Engine
        if self._engine is None or self._engine.stale(self):
            self._engine = Engine(self)
        return self._engine.predict(data)
