import sys, os, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/scd_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
from scd_amd.clip import weights as W
from scd_amd.clip.model import CLIP
sd = W.synthetic_clip_state_dict(seed=3, cfg=dict(v_layers=2, t_layers=2))
model = CLIP(sd).cuda().eval()
img = torch.randn(13, 3, 224, 224, generator=torch.Generator().manual_seed(5)).cuda()
a = model.encode_image(img).float().cpu()
b = model.encode_image(img).float().cpu()
one = model.encode_image(img[:1]).float().cpu()
print("fuse=%s same-input twice equal: %s ; batch-invariant: %s ; max diff %.3e" % (os.environ.get("SCD_LN_FUSE", "1"), torch.equal(a, b), torch.equal(one[0], a[0]), (one[0] - a[0]).abs().max().item()))
