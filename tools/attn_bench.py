"""Time one CLIP visual encode (B=512) and report the attention kernel share through ablation env SCD_ATTN_X."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scd_amd.clip as clip
clip.allow_synthetic()
model, _ = clip.load("ViT-B/16", device="cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randn(B, 3, 224, 224, device="cuda").half()
enc = model.visual.enc
for _ in range(2): enc.encode_image(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): enc.encode_image(x)
e1.record(); torch.cuda.synchronize()
print("SCD_ATTN_X=%s  encode B=%d: %.2f ms" % (os.environ.get("SCD_ATTN_X", "0"), B, e0.elapsed_time(e1) / 5))
