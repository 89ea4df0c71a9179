"""Time the CLIP text tower (77 tokens, width 512, 12 layers) on synthetic token ids: python tools/text_bench.py [batch]."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scd_amd.clip as clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
clip.allow_synthetic()
model, _ = clip.load("ViT-B/16", device="cuda")
g = torch.Generator().manual_seed(0)
tok = torch.zeros(B, 77, dtype=torch.int32)
tok[:, 0] = 49406
ln = torch.randint(4, 20, (B,), generator=g)
for i in range(B):
    tok[i, 1:1 + ln[i]] = torch.randint(1, 49405, (int(ln[i]),), generator=g, dtype=torch.int32)
    tok[i, 1 + ln[i]] = 49407
tok = tok.cuda()
for _ in range(2): model.encode_text(tok)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): model.encode_text(tok)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("encode_text B=%d: %.2f ms  -> %.0f prompts/s" % (B, ms, B / ms * 1e3))
L = int(tok.argmax(-1).max()) + 1
for _ in range(2): model.encode_text(tok, ctx_len=L)
e0.record()
for _ in range(5): model.encode_text(tok, ctx_len=L)
e1.record(); torch.cuda.synchronize()
ms2 = e0.elapsed_time(e1) / 5
print("encode_text B=%d, only the first %d positions (scd_clip_encode_text_len): %.2f ms  -> %.0f prompts/s  (%.2fx)" % (B, L, ms2, B / ms2 * 1e3, ms / ms2))
