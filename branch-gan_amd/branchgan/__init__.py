"""branchgan — MI355X-native implementation of Branch-GAN's branch-sequence hot path.

Host side in Python (mirrors the reference's call surface), compute in hand-written gfx950 HIP kernels
behind the C-ABI of include/branchgan.h (lib/libbranchgan_hip.so).  No CPU fallback exists on purpose.
"""
__version__ = "0.1.0"
