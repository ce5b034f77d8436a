"""MI355X-native PWN dense-registration path (depth image -> cloud with normals -> projective ICP),
drop-in behind the pwn_core Aligner / DepthImageConverter API of grisetti/g2o_frontend.

The compute path is the C-ABI library ``libpwn_hip.so`` (include/pwn_hip.h, hand-written gfx950
kernels in csrc/); ``api`` mirrors the reference's class names over it.
"""
from . import synth  # noqa: F401

__all__ = ["api", "synth", "build"]
