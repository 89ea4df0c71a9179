"""Seeded random-init weights with the reference state-dict key names.

There are no checkpoints offline (the reference ships only download links:
zeroshot_weights/download_link.txt, GCD_pretrained_weights_VIT16/download_link.txt),
so benchmarks and parity tests use these.  The init follows the published
CLIP.initialize_parameters scales; biases and LayerNorm
parameters are perturbed so every fused epilogue term is exercised.
Real checkpoints with the same keys load through the same code path.
"""
import torch

CLIP_VITB16 = dict(embed_dim=512, image=224, patch=16, v_width=768, v_layers=12, v_heads=12,
                   context=77, vocab=49408, t_width=512, t_layers=12, t_heads=8)


def _block(sd, g, prefix, width, layers, names):
    attn_std = width ** -0.5
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    fc_std = (2 * width) ** -0.5
    r = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    sd[prefix + names["ln1_w"]] = 1.0 + 0.1 * r(width)
    sd[prefix + names["ln1_b"]] = 0.02 * r(width)
    sd[prefix + names["qkv_w"]] = r(3 * width, width, std=attn_std)
    sd[prefix + names["qkv_b"]] = 0.02 * r(3 * width)
    sd[prefix + names["proj_w"]] = r(width, width, std=proj_std)
    sd[prefix + names["proj_b"]] = 0.02 * r(width)
    sd[prefix + names["ln2_w"]] = 1.0 + 0.1 * r(width)
    sd[prefix + names["ln2_b"]] = 0.02 * r(width)
    sd[prefix + names["fc1_w"]] = r(4 * width, width, std=fc_std)
    sd[prefix + names["fc1_b"]] = 0.02 * r(4 * width)
    sd[prefix + names["fc2_w"]] = r(width, 4 * width, std=proj_std)
    sd[prefix + names["fc2_b"]] = 0.02 * r(width)


CLIP_BLOCK_KEYS = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "qkv_w": "attn.in_proj_weight",
                   "qkv_b": "attn.in_proj_bias", "proj_w": "attn.out_proj.weight",
                   "proj_b": "attn.out_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias",
                   "fc1_w": "mlp.c_fc.weight", "fc1_b": "mlp.c_fc.bias", "fc2_w": "mlp.c_proj.weight",
                   "fc2_b": "mlp.c_proj.bias"}
DINO_BLOCK_KEYS = {"ln1_w": "norm1.weight", "ln1_b": "norm1.bias", "qkv_w": "attn.qkv.weight",
                   "qkv_b": "attn.qkv.bias", "proj_w": "attn.proj.weight", "proj_b": "attn.proj.bias",
                   "ln2_w": "norm2.weight", "ln2_b": "norm2.bias", "fc1_w": "mlp.fc1.weight",
                   "fc1_b": "mlp.fc1.bias", "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias"}


def synthetic_clip_state_dict(seed=0, cfg=None, visual=True, text=True):
    """float32 state dict with openai/CLIP key names (appendix B 'State-dict keys')."""
    c = dict(CLIP_VITB16)
    c.update(cfg or {})
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    sd = {}
    if visual:
        w, p = c["v_width"], c["patch"]
        sd["visual.conv1.weight"] = r(w, 3, p, p, std=(3 * p * p) ** -0.5)
        sd["visual.class_embedding"] = r(w, std=w ** -0.5)
        sd["visual.positional_embedding"] = r((c["image"] // p) ** 2 + 1, w, std=w ** -0.5)
        sd["visual.ln_pre.weight"] = 1.0 + 0.1 * r(w)
        sd["visual.ln_pre.bias"] = 0.02 * r(w)
        for i in range(c["v_layers"]):
            _block(sd, g, "visual.transformer.resblocks.%d." % i, w, c["v_layers"], CLIP_BLOCK_KEYS)
        sd["visual.ln_post.weight"] = 1.0 + 0.1 * r(w)
        sd["visual.ln_post.bias"] = 0.02 * r(w)
        sd["visual.proj"] = r(w, c["embed_dim"], std=w ** -0.5)
    if text:
        w = c["t_width"]
        sd["token_embedding.weight"] = r(c["vocab"], w, std=0.02)
        sd["positional_embedding"] = r(c["context"], w, std=0.01)
        for i in range(c["t_layers"]):
            _block(sd, g, "transformer.resblocks.%d." % i, w, c["t_layers"], CLIP_BLOCK_KEYS)
        sd["ln_final.weight"] = 1.0 + 0.1 * r(w)
        sd["ln_final.bias"] = 0.02 * r(w)
        sd["text_projection"] = r(w, c["embed_dim"], std=w ** -0.5)
        sd["logit_scale"] = torch.tensor(4.6052)
    return sd


def synthetic_dino_state_dict(seed=1, width=768, layers=12, image=224, patch=16):
    """float32 state dict with the key names of gcd/models/vision_transformer.py:135-219."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    sd = {"patch_embed.proj.weight": r(width, 3, patch, patch, std=(3 * patch * patch) ** -0.5),
          "patch_embed.proj.bias": 0.02 * r(width),
          "cls_token": r(1, 1, width, std=0.02),
          "pos_embed": r(1, (image // patch) ** 2 + 1, width, std=0.02)}
    for i in range(layers):
        _block(sd, g, "blocks.%d." % i, width, layers, DINO_BLOCK_KEYS)
    sd["norm.weight"] = 1.0 + 0.1 * r(width)
    sd["norm.bias"] = 0.02 * r(width)
    return sd
