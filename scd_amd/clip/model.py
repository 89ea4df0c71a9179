"""CLIP / DINO towers on libscd_hip.so, with the third-party `clip` package's surface
(clip.load, model.encode_image, model.encode_text, model.cuda().eval()) that the reference
calls at main_unsup.py:237-238,127 and local_utils/clip_lang_util.py:101-102.
"""
import torch

from .. import ops
from . import weights as W


def _f16(t, dev):
    return t.detach().to(device=dev, dtype=torch.float16).contiguous()


def _f32(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _pack_layers(sd, prefix, keys, layers, dev):
    out = []
    for i in range(layers):
        p = "%s%d." % (prefix, i)
        g = lambda n: sd[p + keys[n]]
        out += [_f32(g("ln1_w"), dev), _f32(g("ln1_b"), dev), _f16(g("qkv_w"), dev), _f32(g("qkv_b"), dev),
                _f16(g("proj_w"), dev), _f32(g("proj_b"), dev), _f32(g("ln2_w"), dev), _f32(g("ln2_b"), dev),
                _f16(g("fc1_w"), dev), _f32(g("fc1_b"), dev), _f16(g("fc2_w"), dev), _f32(g("fc2_b"), dev)]
    return out


def _count(sd, prefix):
    i = 0
    while any(k.startswith("%s%d." % (prefix, i)) for k in sd):
        i += 1
    return i


def build_clip_visual(sd, dev):
    w = sd["visual.conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    tokens = sd["visual.positional_embedding"].shape[0]
    image = int(round((tokens - 1) ** 0.5)) * patch
    layers = _count(sd, "visual.transformer.resblocks.")
    out_dim = sd["visual.proj"].shape[1]
    ws = [_f16(w.reshape(width, -1), dev), None, _f32(sd["visual.class_embedding"], dev),
          _f32(sd["visual.positional_embedding"], dev), _f32(sd["visual.ln_pre.weight"], dev),
          _f32(sd["visual.ln_pre.bias"], dev), _f32(sd["visual.ln_post.weight"], dev), _f32(sd["visual.ln_post.bias"], dev),
          _f16(sd["visual.proj"].t(), dev)]
    ws += _pack_layers(sd, "visual.transformer.resblocks.", W.CLIP_BLOCK_KEYS, layers, dev)
    desc = dict(kind=0, width=width, layers=layers, heads=width // 64, mlp_dim=4 * width, tokens=tokens, patch=patch,
                image=image, vocab=0, out_dim=out_dim, act=0, ln_eps=1e-5)
    return ops.Encoder(desc, ws)


def build_clip_text(sd, dev):
    width = sd["token_embedding.weight"].shape[1]
    layers = _count(sd, "transformer.resblocks.")
    out_dim = sd["text_projection"].shape[1]
    ws = [_f16(sd["token_embedding.weight"], dev), None, None, _f32(sd["positional_embedding"], dev), None, None,
          _f32(sd["ln_final.weight"], dev), _f32(sd["ln_final.bias"], dev), _f16(sd["text_projection"].t(), dev)]
    ws += _pack_layers(sd, "transformer.resblocks.", W.CLIP_BLOCK_KEYS, layers, dev)
    desc = dict(kind=1, width=width, layers=layers, heads=width // 64, mlp_dim=4 * width,
                tokens=sd["positional_embedding"].shape[0], patch=0, image=0, vocab=sd["token_embedding.weight"].shape[0],
                out_dim=out_dim, act=0, ln_eps=1e-5)
    return ops.Encoder(desc, ws)


def build_dino(sd, dev):
    w = sd["patch_embed.proj.weight"]
    width, patch = w.shape[0], w.shape[-1]
    tokens = sd["pos_embed"].shape[1]
    image = int(round((tokens - 1) ** 0.5)) * patch
    layers = _count(sd, "blocks.")
    ws = [_f16(w.reshape(width, -1), dev), _f32(sd["patch_embed.proj.bias"], dev), _f32(sd["cls_token"].reshape(-1), dev),
          _f32(sd["pos_embed"].reshape(tokens, width), dev), None, None, _f32(sd["norm.weight"], dev),
          _f32(sd["norm.bias"], dev), None]
    ws += _pack_layers(sd, "blocks.", W.DINO_BLOCK_KEYS, layers, dev)
    desc = dict(kind=2, width=width, layers=layers, heads=width // 64, mlp_dim=4 * width, tokens=tokens, patch=patch,
                image=image, vocab=0, out_dim=0, act=1, ln_eps=1e-6)
    return ops.Encoder(desc, ws)


class _Visual:
    def __init__(self, enc, cfg):
        self.enc = enc
        self.input_resolution = cfg["image"]
        self.output_dim = enc.out_dim

    def __call__(self, x):
        return self.enc.encode_image(x)


class CLIP:
    """Drop-in for the object `clip.load` returns (only the members the reference touches)."""

    def __init__(self, state_dict):
        self._sd = state_dict
        self._dev = None
        self.visual = None
        self._text = None
        self.dtype = torch.float16
        self.context_length = state_dict["positional_embedding"].shape[0] if "positional_embedding" in state_dict else 77
        self.vocab_size = state_dict["token_embedding.weight"].shape[0] if "token_embedding.weight" in state_dict else 49408

    def cuda(self, device=None):
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        if self._dev != dev:
            self._dev = dev
            if "visual.conv1.weight" in self._sd:
                enc = build_clip_visual(self._sd, dev)
                self.visual = _Visual(enc, enc.desc)
            if "token_embedding.weight" in self._sd:
                self._text = build_clip_text(self._sd, dev)
        return self

    def to(self, device):
        return self.cuda(torch.device(device).index)

    def eval(self):
        return self

    def float(self):
        return self

    def _ready(self):
        if self._dev is None:
            self.cuda()

    def encode_image(self, image):
        self._ready()
        return self.visual.enc.encode_image(image.to(self._dev))

    def encode_text(self, text, ctx_len=None):
        """The package's encode_text.  Token ids handed over on the HOST (as clip.tokenize returns them) are trimmed for free: the
        tower is causal and only the EOT position is read, so positions behind the batch's last EOT are not computed
        (ops.Encoder.encode_text ctx_len: bit-identical features).  Ids already on the device are encoded at full length unless
        the caller passes ctx_len (finding it would cost a device round trip)."""
        self._ready()
        if ctx_len is None and not text.is_cuda and text.numel():
            ctx_len = int(text.numpy().argmax(axis=-1).max()) + 1      # numpy: one thread, ~0.3 ms (a torch CPU op wakes the whole pool)
        return self._text.encode_text(text.to(self._dev), ctx_len=ctx_len)

    def __call__(self, image):
        return self.encode_image(image)


class DinoViT:
    """Callable like the torch.hub dino_vitb16 module the reference uses (main_unsup.py:241,129)."""

    def __init__(self, state_dict):
        self._sd = state_dict
        self._enc = None

    def load_state_dict(self, sd, strict=True):
        self._sd = sd
        self._enc = None

    def cuda(self, device=None):
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._enc = build_dino(self._sd, dev)
        return self

    def eval(self):
        return self

    def features(self, x, normalize=False):
        """[B,3,224,224] -> float32 [B,768]; normalize=True fuses F.normalize(dim=-1) into the encoder's last kernel."""
        if self._enc is None:
            self.cuda()
        return self._enc.encode_image(x.to(self._enc.device), normalize=normalize).float()

    def __call__(self, x):
        return self.features(x)
