"""Vocabulary build: zeroshot_classifier over V names x 80 templates - host tokenisation + text tower + prompt
pooling.  python tools/vocab_bench.py [V] [names_per_batch].  Prints the wall time, the host tokenisation time alone and the
device time alone (same token batches replayed), so that the bound of the build is visible."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scd_amd.clip as clip
from scd_amd.local_utils.clip_lang_util import imagenet_templates, zeroshot_classifier
from scd_amd import ops

V = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
NPB = int(sys.argv[2]) if len(sys.argv) > 2 else 16
clip.allow_synthetic()
model, _ = clip.load("ViT-B/16", device="cuda")
syll = ["ba", "ri", "ton", "mek", "lu", "sha", "vor", "ine", "qua", "dro", "pel", "ast"]
names = ["%s%s%s %s" % (syll[i % 12], syll[(i // 12) % 12], syll[(i // 144) % 12], syll[(i // 1728) % 12]) for i in range(V)]
T = len(imagenet_templates)
zeroshot_classifier(names[:NPB * 2], imagenet_templates, model, NPB)          # warm-up
torch.cuda.synchronize()
t0 = time.time()
w = zeroshot_classifier(names, imagenet_templates, model, NPB)
torch.cuda.synchronize()
wall = time.time() - t0
t0 = time.time()
toks = []
for s in range(0, V, NPB):
    texts = [t.format(c) for c in names[s:s + NPB] for t in imagenet_templates]
    toks.append(clip.tokenize(texts))
host = time.time() - t0
dtoks = [t.cuda() for t in toks]
out = torch.empty((w.shape[0], V), dtype=torch.float16, device="cuda")
torch.cuda.synchronize()
t0 = time.time()
for i, t in enumerate(dtoks):
    emb = model.encode_text(t)
    ops.prompt_pool(emb.contiguous(), t.shape[0] // T, T, out, i * NPB)
torch.cuda.synchronize()
devt = time.time() - t0
assert torch.equal(out, w)
print("vocabulary build V=%d x %d templates (%d prompts, %d names per batch): wall %.2f s = %.0f prompts/s | host tokenisation alone %.2f s "
      "(%.0f prompts/s) | text tower + pooling alone %.2f s (%.0f prompts/s) | tokenizer: %s"
      % (V, T, V * T, NPB, wall, V * T / wall, host, V * T / host, devt, V * T / devt, type(clip._tokenizer).__name__))
