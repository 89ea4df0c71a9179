"""Features of the towers under one setting of the A/B switches (SCD_PATCH_FROM_IMAGE, SCD_ASSEMBLE_ROWS, SCD_ATTN_SHORT, ...), saved for a
bit-for-bit comparison with another setting's:  python tools/patch_img_check.py save out.pt [--text]  |  python tools/patch_img_check.py cmp a.pt b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd.clip import weights as W
from scd_amd.clip.model import CLIP, DinoViT
if sys.argv[1] == "save":
    g = torch.Generator(device="cuda").manual_seed(5)
    out = {}
    clip = CLIP(W.synthetic_clip_state_dict(seed=0, text=False)).cuda()
    dino = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=12)).cuda()
    for b in (3, 257, 1000):             # 1 row tile with padding rows, several tiles, a batch whose last tile is partly padding
        x = torch.randn((b, 3, 224, 224), device="cuda", generator=g).half()
        out["clip%d" % b] = clip.visual.enc.encode_image(x).cpu()
        out["dino%d" % b] = dino._enc.encode_image(x).cpu()
    if "--text" in sys.argv:                  # the text tower at trimmed contexts (<= 32 positions: attention_short_kernel / attention_kernel<1>)
        import scd_amd.clip as cl
        cl.allow_synthetic()
        model = CLIP(W.synthetic_clip_state_dict(seed=0)).cuda()
        for tag, names in (("short", ["cat", "a dog", "x"] * 50), ("long", ["a fairly long name of seven words here", "golden retriever puppy"] * 333)):
            tok = cl.tokenize(["a photo of a %s." % n for n in names])
            out["text_" + tag] = model.encode_text(tok).cpu()
    torch.save(out, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        same = torch.equal(a[k], b[k])
        print(k, "bit-identical" if same else "DIFFERENT max |d| %.3g" % (a[k].float() - b[k].float()).abs().max().item(), "finite", bool(torch.isfinite(a[k].float()).all()))
        assert same
    print("patch-from-image == im2col on", len(a), "feature matrices")
