"""oracle/clip_oracle.py pinned against (a) the reference's own DINO ViT file and (b) transformers.CLIPModel,
both run in the build container on the seeded weights of scd_amd/clip/weights.py (tests/golden/encoders.npz)."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as co
from scd_amd.clip import weights as W


@pytest.mark.parametrize("layers,tag", [(2, "d2"), (12, "d12")])
def test_dino_matches_reference_file(golden, layers, tag):
    g = golden("encoders.npz")
    sd = W.synthetic_dino_state_dict(seed=1, layers=layers)
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(77))
    out = co.dino_forward(sd, img).numpy()
    assert np.allclose(out, g[tag + "_out"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("layers,tag", [(2, "c2"), (12, "c12")])
def test_clip_matches_hf(golden, layers, tag):
    g = golden("encoders.npz")
    sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(v_layers=layers, t_layers=layers))
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(78))
    io = co.clip_encode_image(sd, img).numpy()
    to = co.clip_encode_text(sd, torch.from_numpy(g[tag + "_tok"])).numpy()
    assert np.allclose(io, g[tag + "_img"], rtol=1e-4, atol=2e-4)
    assert np.allclose(to, g[tag + "_txt"], rtol=1e-4, atol=2e-4)
