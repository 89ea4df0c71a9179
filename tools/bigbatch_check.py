import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scd_amd.clip as clip
clip.allow_synthetic()
model, _ = clip.load("ViT-B/16", device="cuda")
g = torch.Generator(device="cuda").manual_seed(1)
n = 8000
x = torch.randn(n, 3, 224, 224, device="cuda", generator=g).half()
enc = model.visual.enc
big = enc.encode_image(x)                                     # one batch of 8000 (1.58 M token rows)
mid = torch.cat([enc.encode_image(x[i:i + 3990]) for i in range(0, n, 3990)])
small = torch.cat([enc.encode_image(x[i:i + 665]) for i in range(0, n, 665)])
print("8000-in-one == 3990-chunks:", torch.equal(big, mid), " == 665-chunks:", torch.equal(big, small), "finite:", bool(torch.isfinite(big.float()).all()))
