"""vpho_amd -- MI355X-native implementation of VPHO's per-image inference hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).
Device side: hand-written HIP for gfx950 behind the C-ABI in ``include/vpho_hip.h``
(``vpho_amd/csrc`` -> ``vpho_amd/libvpho_hip.so``).  There is no CPU fallback: every op
raises if the extension is missing.
"""
__all__ = ['assets', 'ops']
