"""Stress: the full-size encoder (12 layers, B images per launch: 665 by default; 3990 = the bench's launch size, non-temporal stores and
several rounds of tiles) must give bit-identical features on repeated runs (races in the LDS-DMA rings / barriers, store-data hazards or
order-dependent atomics would show up here).  python tests/stress_determinism.py [repeats] [B]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scd_amd.clip as clip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
clip.allow_synthetic()
model, _ = clip.load("ViT-B/16", device="cuda")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 665
x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(11)).half().cuda()
ref = model.encode_image(x).clone()
bad = 0
for i in range(reps):
    out = model.encode_image(x)
    if not torch.equal(out, ref):
        bad += 1
        print("run %d differs: max abs diff %.3e, rows differing %d" % (i, (out.float() - ref.float()).abs().max().item(), int((out != ref).any(1).sum())))
assert torch.isfinite(ref.float()).all()
print("determinism stress: B = %d, %d repeats, %d mismatches" % (B, reps, bad))
sys.exit(1 if bad else 0)
