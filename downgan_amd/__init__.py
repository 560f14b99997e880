"""downgan_amd: MI355X-native WGAN-GP train step (the DoWnGAN hot path) behind the reference's API.

The product path runs hand-written HIP kernels (gfx950) through the C ABI in
``include/downgan_hip.h``; there is no CPU fallback.  Importing this package does not
load the shared library; the first op call does and fails loudly when it is missing.
"""
__version__ = "0.1.0"
