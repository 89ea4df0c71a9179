"""Vocabulary build: zeroshot_classifier over V names x 80 templates - host tokenisation + text tower + prompt
pooling.  python tools/vocab_bench.py [V] [names_per_batch] [bpe].  With `bpe` the real SimpleTokenizer runs, on a merges table
learned here from prompt-like text (the package's 16e6 file is absent offline): its host cost is what a real-data build pays.  Prints the wall time, the host tokenisation time alone and the
device time alone (same token batches replayed), so that the bound of the build is visible."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scd_amd.clip as clip
from scd_amd.local_utils.clip_lang_util import imagenet_templates, zeroshot_classifier
from scd_amd import ops

V = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
NPB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
LG = int(os.environ.get("VB_GROUPS", "4"))
MG = int(os.environ.get("VB_MIN_GROUP", "2048"))
NAMES = os.environ.get("VB_NAMES", "syllables")
clip.allow_synthetic()
model, _ = clip.load("ViT-B/16", device="cuda")
if len(sys.argv) > 3 and sys.argv[3] == "bpe":
    import gzip, json, tempfile
    from tokenizers import Regex, Tokenizer, models, normalizers, pre_tokenizers, trainers
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs, nn = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + nn); nn += 1
    pat = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""
    tkz = Tokenizer(models.BPE(end_of_word_suffix="</w>"))
    tkz.normalizer = normalizers.Lowercase()
    tkz.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(pat), behavior="removed", invert=True),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    sy = ["ba", "ri", "ton", "mek", "lu", "sha", "vor", "ine", "qua", "dro", "pel", "ast"]
    corpus = [t.format("%s%s %s" % (sy[i % 12], sy[(i // 12) % 12], sy[(i // 144) % 12])) for i in range(400) for t in imagenet_templates[:20]]
    tkz.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=3000, initial_alphabet=[chr(c) for c in cs], end_of_word_suffix="</w>",
                                                        special_tokens=[], show_progress=False))
    merges = [tuple(m) for m in json.loads(tkz.to_str())["model"]["merges"]]
    path = os.path.join(tempfile.mkdtemp(), "bpe_synth.txt.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    clip._tokenizer = clip.SimpleTokenizer(path)
syll = ["ba", "ri", "ton", "mek", "lu", "sha", "vor", "ine", "qua", "dro", "pel", "ast"]
names = ["%s%s%s %s" % (syll[i % 12], syll[(i // 12) % 12], syll[(i // 144) % 12], syll[(i // 1728) % 12]) for i in range(V)]
if NAMES == "numbered":
    names = ["name_%05d" % i for i in range(V)]          # bench.py --config c5's names
T = len(imagenet_templates)
zeroshot_classifier(names[:NPB * 2], imagenet_templates, model, NPB, LG, MG)          # warm-up
torch.cuda.synchronize()
t0 = time.time()
w = zeroshot_classifier(names, imagenet_templates, model, NPB, LG, MG)
torch.cuda.synchronize()
wall = time.time() - t0
t0 = time.time()
toks = []
for s in range(0, V, NPB):
    toks.append(clip.tokenize_templates(names[s:s + NPB], imagenet_templates))
host = time.time() - t0
t0 = time.time()
for s in range(0, min(V, 640), NPB):
    clip.tokenize([t.format(c) for c in names[s:s + NPB] for t in imagenet_templates])
plain = (time.time() - t0) / (min(V, 640) * len(imagenet_templates))
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
      "(%.0f prompts/s; prompt by prompt: %.0f prompts/s) | text tower + pooling at full length alone %.2f s (%.0f prompts/s) | tokenizer: %s"
      % (V, T, V * T, NPB, wall, V * T / wall, host, V * T / host, 1.0 / plain, devt, V * T / devt, type(clip._tokenizer).__name__))
