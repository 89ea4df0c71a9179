"""CPU oracle (torch, float32) for the encoder part of the hot path.

TEST INFRASTRUCTURE ONLY - never imported from scd_amd/.

* CLIP ViT-B/16 visual / text towers: the arithmetic lives in the third-party
  package `clip==1.0` (openai/CLIP, requirements.txt:27), which is NOT under
  /root/reference and not installed; call sites main_unsup.py:237,127 and
  local_utils/clip_lang_util.py:101-102.  This file restates the published
  model.py structure.  It is pinned against
  transformers.CLIPModel (an independent implementation of the same published
  architecture) on shared seeded random weights: tests/golden/clip_hf.npz,
  produced by oracle/gen_golden.py.  Parity against the true package/weights
  is unpinned (no checkpoint offline).
* DINO ViT-B/16: restates /root/reference/gcd/models/vision_transformer.py
  (VisionTransformer :135-219, Attention :67-91, Block :94-114, PatchEmbed
  :117-132), pinned against that file run in the build container
  (tests/golden/dino_ref.npz).

Weights are dicts keyed with the reference state-dict names.
"""
import math
import torch
import torch.nn.functional as F


def _ln(x, w, b, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def _act(x, kind):
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)        # CLIP QuickGELU
    return F.gelu(x)                                # nn.GELU (vision_transformer.py:49)


def _attention(x, qkv_w, qkv_b, out_w, out_b, heads, causal):
    b, t, c = x.shape
    qkv = x @ qkv_w.float().t() + qkv_b.float()
    q, k, v = qkv.view(b, t, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * (c // heads) ** -0.5
    if causal:
        s = s + torch.full((t, t), float("-inf")).triu_(1)
    p = s.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(b, t, c)
    return o @ out_w.float().t() + out_b.float()


def _block(x, g, heads, act, eps, causal):
    """g(name) -> tensor for the canonical per-block names."""
    x = x + _attention(_ln(x, g("ln1_w"), g("ln1_b"), eps), g("qkv_w"), g("qkv_b"), g("proj_w"), g("proj_b"),
                       heads, causal)
    h = _ln(x, g("ln2_w"), g("ln2_b"), eps) @ g("fc1_w").float().t() + g("fc1_b").float()
    return x + _act(h, act) @ g("fc2_w").float().t() + g("fc2_b").float()


_CLIP_BLOCK = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "qkv_w": "attn.in_proj_weight",
               "qkv_b": "attn.in_proj_bias", "proj_w": "attn.out_proj.weight", "proj_b": "attn.out_proj.bias",
               "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias", "fc1_w": "mlp.c_fc.weight", "fc1_b": "mlp.c_fc.bias",
               "fc2_w": "mlp.c_proj.weight", "fc2_b": "mlp.c_proj.bias"}
_DINO_BLOCK = {"ln1_w": "norm1.weight", "ln1_b": "norm1.bias", "qkv_w": "attn.qkv.weight", "qkv_b": "attn.qkv.bias",
               "proj_w": "attn.proj.weight", "proj_b": "attn.proj.bias", "ln2_w": "norm2.weight",
               "ln2_b": "norm2.bias", "fc1_w": "mlp.fc1.weight", "fc1_b": "mlp.fc1.bias",
               "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias"}


def _n_blocks(sd, prefix):
    i = 0
    while any(k.startswith("%s%d." % (prefix, i)) for k in sd):
        i += 1
    return i


@torch.no_grad()
def clip_encode_image(sd, images, heads=12):
    """VisionTransformer.forward of openai/CLIP model.py (appendix B 'Visual')."""
    x = F.conv2d(images.float(), sd["visual.conv1.weight"].float(), stride=sd["visual.conv1.weight"].shape[-1])
    b, c = x.shape[:2]
    x = x.reshape(b, c, -1).permute(0, 2, 1)
    cls = sd["visual.class_embedding"].float().expand(b, 1, c)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].float()
    x = _ln(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    for i in range(_n_blocks(sd, "visual.transformer.resblocks.")):
        p = "visual.transformer.resblocks.%d." % i
        x = _block(x, lambda n: sd[p + _CLIP_BLOCK[n]], heads, "quick_gelu", 1e-5, False)
    x = _ln(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    return x @ sd["visual.proj"].float()


@torch.no_grad()
def clip_encode_text(sd, tokens, heads=8):
    """CLIP.encode_text (appendix B 'Text'): EOT row = argmax token id."""
    x = sd["token_embedding.weight"].float()[tokens.long()] + sd["positional_embedding"].float()
    for i in range(_n_blocks(sd, "transformer.resblocks.")):
        p = "transformer.resblocks.%d." % i
        x = _block(x, lambda n: sd[p + _CLIP_BLOCK[n]], heads, "quick_gelu", 1e-5, True)
    x = _ln(x, sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    x = x[torch.arange(x.shape[0]), tokens.long().argmax(dim=-1)]
    return x @ sd["text_projection"].float()


@torch.no_grad()
def dino_forward(sd, images, heads=12):
    """gcd/models/vision_transformer.py:210-219 (prepare_tokens, blocks, norm, [:,0])."""
    w = sd["patch_embed.proj.weight"].float()
    x = F.conv2d(images.float(), w, sd["patch_embed.proj.bias"].float(), stride=w.shape[-1])
    b, c = x.shape[:2]
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["cls_token"].float().expand(b, -1, -1), x], dim=1) + sd["pos_embed"].float()
    for i in range(_n_blocks(sd, "blocks.")):
        p = "blocks.%d." % i
        x = _block(x, lambda n: sd[p + _DINO_BLOCK[n]], heads, "gelu", 1e-6, False)
    return _ln(x, sd["norm.weight"], sd["norm.bias"], 1e-6)[:, 0]
