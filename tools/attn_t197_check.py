"""SCD_ATTN_T197 = 0 / 1 (generic persistent attention / the T = 197 specialisation): CLIP and DINO features of the same images must be
bit-identical; prints the encode time of each.  Usage: attn_t197_check.py [B]  (child mode: --child MODE B OUT)"""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    import scd_amd.clip as clip
    from scd_amd.clip import weights as W
    from scd_amd.clip.model import DinoViT
    clip.allow_synthetic()
    B = int(sys.argv[3])
    model, _ = clip.load("ViT-B/16", device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 3, 224, 224, device="cuda", generator=g).half()
    enc = model.visual.enc
    f = enc.encode_image(x)
    for _ in range(2): enc.encode_image(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): enc.encode_image(x)
    e1.record(); torch.cuda.synchronize()
    fd = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=12)).cuda()(x[:700])
    torch.save({"clip": f.cpu(), "dino": fd.float().cpu()}, sys.argv[4])
    print("SCD_ATTN_T197=%s B=%d: %.2f ms per encode (%.0f images/s)" % (sys.argv[2], B, e0.elapsed_time(e1) / 5, B / (e0.elapsed_time(e1) / 5e3)), flush=True)
    sys.exit(0)
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3990
outs = {}
for rep in range(2):
    for mode in ("0", "1"):
        env = dict(os.environ, SCD_ATTN_T197=mode)
        out = "/tmp/attn_t197_%s.pt" % mode
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, str(B), out], env=env, check=True)
        outs[mode] = torch.load(out)
for key in ("clip", "dino"):
    a, b = outs["0"][key], outs["1"][key]
    print("%s features bit-identical: %s (shape %s, max |diff| %.3g)" % (key, torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)), tuple(a.shape), (a.float() - b.float()).abs().max().item()))
