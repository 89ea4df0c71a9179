"""Print the worst cosine between the HIP towers and the fp32 oracle (same inputs as tests/test_gpu_parity.py)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this script lives in tests/: it uses the oracle)
sys.path.insert(0, ROOT)
from scd_amd.clip import weights as W
from scd_amd.clip.model import CLIP, DinoViT
from oracle import clip_oracle as co

def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)

sd = W.synthetic_clip_state_dict(seed=0, cfg=dict(v_layers=12, t_layers=12))
sd16 = {k: (v.half().float() if v.dim() >= 2 and "positional" not in k and "class_emb" not in k else v) for k, v in sd.items()}
model = CLIP(sd).cuda().eval()
img = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(78))
out = model.encode_image(img.cuda()).float().cpu()
ref = co.clip_encode_image(sd16, img.half().float())
print("LN_FUSE=%s clip visual: 1-cos max %.3e, max abs err / max %.3e" % (os.environ.get("SCD_LN_FUSE", "1"), (1 - cos(out, ref)).max().item(),
      (out - ref).abs().max().item() / ref.abs().max().item()))
sd = W.synthetic_dino_state_dict(seed=1, layers=12)
sd16 = {k: (v.half().float() if v.dim() >= 2 and "pos_embed" not in k and "cls_token" not in k else v) for k, v in sd.items()}
img = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(77))
out = DinoViT(sd).cuda()(img.cuda()).cpu()
ref = co.dino_forward(sd16, img.half().float())
print("LN_FUSE=%s dino: 1-cos max %.3e" % (os.environ.get("SCD_LN_FUSE", "1"), (1 - cos(out, ref)).max().item()))
