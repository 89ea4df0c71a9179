"""scd_amd - MI355X-native (gfx950) embedding-and-naming hot path of Visual-AI/SCD.

The compute lives in scd_amd/csrc (hand-written HIP + a C-ABI shared library,
include/scd_hip.h); the Python modules here mirror the reference's call
signatures (clip, local_utils.clip_lang_util, local_utils.sskm_constrained,
gcd.methods.clustering.faster_mix_k_means_pytorch) so the reference's
main_unsup.py / main_ptsup.py flow runs on it unchanged.
"""
__version__ = "0.1.0"
