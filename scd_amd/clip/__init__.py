"""`clip`-compatible module surface backed by libscd_hip.so.

Mirrors the third-party openai/CLIP package API used by the reference
(main_unsup.py:237 `clip.load("ViT-B/16")`; clip_lang_util.py:101 `clip.tokenize`).
Checkpoints: `load` looks for $SCD_ROOT/clip/ViT-B-16.pt - the openai download as it is (a TorchScript archive) or a
torch.save of its state dict.  Without one it RAISES: random-init weights (and the hash tokenizer that stands in for the BPE
merges file) produce meaningless features, so they must be asked for explicitly - `load(..., synthetic=True)`,
`clip.allow_synthetic()` or SCD_SYNTHETIC=1 - as bench.py, smoke(), the --synthetic mode of the mains and the tests do
(there is no network in the build/benchmark environment).
"""
import gzip
import html
import os
import warnings

import numpy as np
import torch

from .model import CLIP, DinoViT
from . import weights

_MODELS = {"ViT-B/16": weights.CLIP_VITB16}
MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def available_models():
    return list(_MODELS)


def _preprocess(n_px):
    """Resize(n_px, bicubic) -> CenterCrop -> RGB -> ToTensor -> Normalize (SURVEY.md appendix B 'Preprocess')."""
    def run(img):
        from PIL import Image
        img = img.convert("RGB")
        w, h = img.size
        s = n_px / min(w, h)
        img = img.resize((max(n_px, int(round(w * s))), max(n_px, int(round(h * s)))), Image.BICUBIC)
        w, h = img.size
        l, t = (w - n_px) // 2, (h - n_px) // 2
        img = img.crop((l, t, l + n_px, t + n_px))
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (x - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
    return run


_allow_synthetic = False


def allow_synthetic(flag=True):
    """Opt in to seeded random-init weights / the hash tokenizer when the real files are absent (benchmarks, tests)."""
    global _allow_synthetic
    _allow_synthetic = bool(flag)


def _synthetic_ok(explicit=None):
    if explicit is not None:
        return bool(explicit)
    return _allow_synthetic or os.environ.get("SCD_SYNTHETIC", "") not in ("", "0")


def _read_checkpoint(path):
    """State dict of an openai/CLIP checkpoint: the stock download is a TorchScript archive, a converted one a pickle."""
    try:
        return torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu", weights_only=False)
        return sd.state_dict() if hasattr(sd, "state_dict") else sd


def load(name="ViT-B/16", device=None, jit=False, download_root=None, seed=0, synthetic=None):
    if name not in _MODELS:
        raise RuntimeError("Model %s not found; available models = %s" % (name, available_models()))
    root = download_root or os.environ.get("SCD_ROOT", "")
    path = os.path.join(root, "clip", name.replace("/", "-") + ".pt") if root else ""
    if path and os.path.exists(path):
        sd = {k: v for k, v in _read_checkpoint(path).items() if k not in ("input_resolution", "context_length", "vocab_size")}
        synthetic = False
    elif _synthetic_ok(synthetic):
        sd = weights.synthetic_clip_state_dict(seed=seed)
        synthetic = True
    else:
        raise FileNotFoundError("CLIP checkpoint %s not found: put the openai ViT-B-16.pt under $SCD_ROOT/clip/ (or pass "
                                "download_root); seeded random-init weights must be requested explicitly with "
                                "load(..., synthetic=True), clip.allow_synthetic() or SCD_SYNTHETIC=1" % (path or "$SCD_ROOT/clip/ViT-B-16.pt"))
    model = CLIP(sd)
    model.synthetic = synthetic
    if device is None or str(device).startswith("cuda"):
        if torch.cuda.is_available():
            model.cuda()
    return model, _preprocess(_MODELS[name]["image"])


# ----------------------------------------------------------------------------- tokenizer
class SimpleTokenizer:
    """Byte-level BPE of openai/CLIP simple_tokenizer (lower-cased, SOT 49406, EOT 49407).
    Needs the merges file bpe_simple_vocab_16e6.txt.gz ($SCD_CLIP_BPE or $SCD_ROOT/clip/)."""

    def __init__(self, bpe_path):
        import regex as re
        bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
        cs = bs[:]
        n = 0
        for b in range(256):
            if b not in bs:
                bs.append(b)
                cs.append(256 + n)
                n += 1
        self.byte_encoder = dict(zip(bs, [chr(c) for c in cs]))
        merges = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in merges[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        for m in merges:
            vocab.append("".join(m))
        vocab.extend(["<|startoftext|>", "<|endoftext|>"])
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                              re.IGNORECASE)
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = set(zip(word[:-1], word[1:]))
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                    new.extend(word[i:j])
                    i = j
                except ValueError:
                    new.extend(word[i:])
                    break
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = set(zip(word[:-1], word[1:]))
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        text = " ".join(html.unescape(html.unescape(text)).split()).strip().lower()
        ids = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def _cls(self, ch):
        import regex as re
        return "L" if re.fullmatch(r"\p{L}", ch) else "N" if re.fullmatch(r"\p{N}", ch) else "S" if ch.isspace() else "O"

    def separable(self, a, b):
        """True when no token of the pre-tokenisation pattern can span the junction of a text ending in character a and a text
        starting with b, so that encode(x + y) == encode(x) + encode(y): tokens are runs of letters, single digits, runs of other
        non-space characters, or the apostrophe contractions."""
        ca, cb = self._cls(a), self._cls(b)
        if ca == "S" or cb == "S":
            return True
        if a in "'&;#<|>" or b in "&;#<|>":          # contractions, html entities and the special tokens are matched as units
            return False
        return not ((ca == "L" and cb == "L") or (ca == "O" and cb == "O"))


class HashTokenizer:
    """Stand-in used ONLY when the BPE merges file is absent (synthetic benchmarks): the pre-tokenisation of the real tokenizer (runs of
    letters, single digits, runs of other characters; '_' counts as a space, as in the WordNet noun lists) with ONE id per piece, a
    stable hash into [1, 49405], instead of the BPE.  Results are not comparable with real CLIP tokenisation; prompt LENGTHS and the
    junction rules are (a trailing '.' is its own token, so tokenize_templates assembles prompts from pieces exactly as it does with
    the real merges - round 6: the word-level stand-in glued the template's '.' to the name and sent every prompt down the
    prompt-by-prompt path, 0.59 M prompts/s against 2-5 M with the real tokenizer)."""
    sot, eot = 49406, 49407

    def __init__(self):
        import re
        self.pat = re.compile(r"[^\W\d_]+|\d|[^\s\w]+")

    def encode(self, text):
        import zlib
        return [1 + zlib.crc32(w.encode("utf-8")) % 49405 for w in self.pat.findall(text.lower().replace("_", " "))]

    @staticmethod
    def _cls(ch):
        return "S" if (ch.isspace() or ch == "_") else "L" if ch.isalpha() else "N" if ch.isdigit() else "O"

    def separable(self, a, b):
        """encode(x + y) == encode(x) + encode(y) for x ending in a and y starting with b (the rule of SimpleTokenizer.separable)."""
        ca, cb = self._cls(a), self._cls(b)
        if ca == "S" or cb == "S":
            return True
        return not ((ca == "L" and cb == "L") or (ca == "O" and cb == "O"))


_tokenizer = None


def _get_tokenizer():
    global _tokenizer
    if _tokenizer is None:
        cands = [os.environ.get("SCD_CLIP_BPE", ""), os.path.join(os.environ.get("SCD_ROOT", ""), "clip", "bpe_simple_vocab_16e6.txt.gz")]
        path = next((p for p in cands if p and os.path.exists(p)), None)
        if path:
            _tokenizer = SimpleTokenizer(path)
        elif _synthetic_ok():
            warnings.warn("CLIP BPE merges file not found ($SCD_CLIP_BPE); using the synthetic hash tokenizer")
            _tokenizer = HashTokenizer()
        else:
            raise FileNotFoundError("CLIP BPE merges file bpe_simple_vocab_16e6.txt.gz not found ($SCD_CLIP_BPE or $SCD_ROOT/clip/); "
                                    "the hash tokenizer stand-in must be requested explicitly (clip.allow_synthetic() or "
                                    "SCD_SYNTHETIC=1)")
    return _tokenizer


def tokenize_templates(names, templates, context_length=77, truncate=False):
    """tokenize([t.format(n) for n in names for t in templates]) - the prompt set of zeroshot_classifier - without running the
    pre-tokeniser and the BPE over every prompt: a template's text before and after its single '{}' and every name are encoded
    once and the id lists concatenated, wherever no token can span a junction (Tokenizer.separable: encode(x + y) == encode(x) +
    encode(y) there).  Any other (name, template) pair goes through encode() of the formatted prompt.  Same rows, same order."""
    tk = _get_tokenizer()
    names = list(names)
    plans = []
    for t in templates:
        parts = t.split("{}")
        plain = len(parts) == 2 and "{" not in parts[0] + parts[1] and "}" not in parts[0] + parts[1]
        if plain and (parts[0] == "" or parts[0][-1].isspace() or parts[0][-1] in ".,!?:\"(") and "&" not in t:
            plans.append((parts[0], parts[1], tk.encode(parts[0]), tk.encode(parts[1])))
        else:
            plans.append(None)
    n_names, n_t = len(names), len(templates)
    out = np.zeros((n_names, n_t, context_length), dtype=np.int32)
    sep = {}

    def separable(a, b):
        r = sep.get((a, b))
        if r is None:
            r = sep[(a, b)] = tk.separable(a, b)
        return r

    def slow(i, j):
        ids = [tk.sot] + tk.encode(templates[j].format(names[i])) + [tk.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError("Input %s is too long for context length %d" % (templates[j].format(names[i]), context_length))
            ids = ids[:context_length]
            ids[-1] = tk.eot
        out[i, j, :len(ids)] = ids

    simple = np.array([n != "" and n == n.strip() and "&" not in n and "{" not in n and "}" not in n for n in names], dtype=bool)
    nid = [tk.encode(n) if ok else [] for n, ok in zip(names, simple)]
    nlen = np.array([len(x) for x in nid], dtype=np.int64)
    by_len = {}                                     # name-id matrices per id count: rows filled with one numpy assignment
    for ln in np.unique(nlen[simple]) if simple.any() else []:
        rows = np.nonzero(simple & (nlen == ln))[0]
        by_len[int(ln)] = (rows, np.array([nid[i] for i in rows], dtype=np.int32).reshape(len(rows), int(ln)))
    ufirst, ifirst = np.unique(np.array([n[:1] or " " for n in names]), return_inverse=True)
    ulast, ilast = np.unique(np.array([n[-1:] or " " for n in names]), return_inverse=True)
    for j, plan in enumerate(plans):
        if plan is None:
            for i in range(n_names):
                slow(i, j)
            continue
        pre, suf, pid, sid = plan
        ok = simple.copy()
        if pre != "":
            ok &= np.array([separable(pre[-1], c) for c in ufirst], dtype=bool)[ifirst]
        if suf != "":
            ok &= np.array([separable(c, suf[0]) for c in ulast], dtype=bool)[ilast]
        lp, ls = len(pid), len(sid)
        for ln, (rows, mat) in by_len.items():
            keep = ok[rows]
            if 2 + lp + ln + ls > context_length:          # too long: the slow path truncates or raises like tokenize()
                ok[rows] = False
                continue
            r = rows[keep]
            if not len(r):
                continue
            out[r, j, 0] = tk.sot
            out[r, j, 1:1 + lp] = pid
            out[r, j, 1 + lp:1 + lp + ln] = mat[keep]
            out[r, j, 1 + lp + ln:1 + lp + ln + ls] = sid
            out[r, j, 1 + lp + ln + ls] = tk.eot
        for i in np.nonzero(~ok)[0]:
            slow(i, j)
    return torch.from_numpy(out.reshape(n_names * n_t, context_length))


def tokenize(texts, context_length=77, truncate=False):
    """-> IntTensor [len(texts), context_length]; raises if a text is too long (like the package)."""
    if isinstance(texts, str):
        texts = [texts]
    tk = _get_tokenizer()
    out = np.zeros((len(texts), context_length), dtype=np.int32)      # filled on the host row by row: a torch op per prompt cost more than the BPE
    for i, t in enumerate(texts):
        ids = [tk.sot] + tk.encode(t) + [tk.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError("Input %s is too long for context length %d" % (t, context_length))
            ids = ids[:context_length]
            ids[-1] = tk.eot
        out[i, :len(ids)] = ids
    return torch.from_numpy(out)
