"""scd_amd - MI355X-native (gfx950) embedding-and-naming hot path of Visual-AI/SCD.

The compute lives in scd_amd/csrc (hand-written HIP + a C-ABI shared library,
include/scd_hip.h); the Python modules here mirror the reference's call
signatures (clip, local_utils.clip_lang_util, local_utils.sskm_constrained,
gcd.methods.clustering.faster_mix_k_means_pytorch) so the reference's
main_unsup.py / main_ptsup.py flow runs on it unchanged.
"""
__version__ = "0.1.0"


def install():
    """Alias the HIP-backed modules under the import names the reference uses (main_unsup.py:11-27), so
    `import clip`, `from local_utils.sskm_constrained import K_Means`, ... resolve here."""
    import importlib
    import sys
    pairs = {
        "clip": "scd_amd.clip",
        "local_utils": "scd_amd.local_utils",
        "local_utils.clip_lang_util": "scd_amd.local_utils.clip_lang_util",
        "local_utils.sskm_constrained": "scd_amd.local_utils.sskm_constrained",
        "local_utils.faster_mix_k_means_pytorch": "scd_amd.local_utils.faster_mix_k_means_pytorch",
        "local_utils.util": "scd_amd.local_utils.util",
        "gcd": "scd_amd.gcd",
        "gcd.methods": "scd_amd.gcd.methods",
        "gcd.methods.clustering": "scd_amd.gcd.methods.clustering",
        "gcd.methods.clustering.faster_mix_k_means_pytorch": "scd_amd.gcd.methods.clustering.faster_mix_k_means_pytorch",
        "gcd.project_utils": "scd_amd.gcd.project_utils",
        "gcd.project_utils.cluster_utils": "scd_amd.gcd.project_utils.cluster_utils",
        "gcd.project_utils.cluster_and_log_utils": "scd_amd.gcd.project_utils.cluster_and_log_utils",
        "project_utils": "scd_amd.gcd.project_utils",
        "project_utils.cluster_utils": "scd_amd.gcd.project_utils.cluster_utils",
        "project_utils.cluster_and_log_utils": "scd_amd.gcd.project_utils.cluster_and_log_utils",
    }
    for alias, real in pairs.items():
        sys.modules[alias] = importlib.import_module(real)
