"""MI355X-native hot path of darkAlert/sports-field-homography (import name: ``sfh_amd``).

UNet conv stack + STN homography warp behind the reference's ``Reconstructor`` API.
All arithmetic runs in hand-written HIP kernels (``csrc/``) reached through the C-ABI
library ``libsfh_amd.so`` (``include/sfh_amd.h``); there is no CPU or torch-op fallback.
"""
__version__ = "0.1.0"
