"""Seeded state dicts with the pathologies of TRAINED CLIP / DINO checkpoints (test infrastructure).

The real weights are unobtainable offline (main_unsup.py:237 `clip.load("ViT-B/16")`), and the synthetic
init of scd_amd/clip/weights.py is well-conditioned Gaussian.  Trained ViTs are not: a handful of residual channels carry values
50-100 times the rest ("massive activations"), LayerNorm gains spread over more than two decades, some heads produce attention
logits of several tens, single MLP channels carry a large bias.  Those are exactly the inputs the LayerNorm-folded GEMM epilogue
(out = rstd * (acc - mean * colsum) + b', gemm.hip) and the fixed-point row statistics are sensitive to.  `pathologise` plants them
in a synthetic state dict; `residual_stream_stats` runs the fp32 oracle block by block and reports what the planted weights do to
the residual stream, so that a test can assert that the pathologies are actually there and inside the fixed-point range.
"""
import math

import numpy as np
import torch

from oracle import clip_oracle as co

CLIP_KEYS = co._CLIP_BLOCK
DINO_KEYS = co._DINO_BLOCK


def _loguniform(rs, n, lo, hi):
    return torch.from_numpy(np.exp(rs.uniform(math.log(lo), math.log(hi), size=n))).float()


def pathologise(sd, prefix, keys, width, heads, seed, pre_ln=None, emb_keys=(), n_out=5, logit_target=18.0):
    """In place.  prefix: block prefix ("visual.transformer.resblocks."), keys: canonical -> state-dict names.
    * residual outliers: n_out channels of the embeddings / of every block's proj and fc2 output rows scaled x50-100 (the channels a
      trained model writes its "registers" into), one of them also through an fc2 bias spike;
    * every LayerNorm gain log-uniform in [0.02, 8] (the outlier channels get SMALL gains, as trained models learn: 0.02-0.1);
    * two heads per block with q / k rows scaled so that their logits reach several tens;
    * one fc1 bias spike per block (+12: QuickGELU / GELU on its linear branch) and one of -12 (its flat branch).
    Returns the outlier channel indices."""
    rs = np.random.RandomState(seed)
    out_ch = rs.choice(width, size=n_out, replace=False)
    scale = torch.from_numpy(rs.uniform(50.0, 100.0, size=n_out)).float()
    for k in emb_keys:                       # class / positional embeddings: the outlier channels exist from token 0 on
        t = sd[k]
        t.view(-1, width)[:, out_ch] *= scale * 0.2
    layers = co._n_blocks(sd, prefix)
    for i in range(layers):
        p = "%s%d." % (prefix, i)
        for ln in ("ln1_w", "ln2_w"):
            g = _loguniform(rs, width, 0.02, 8.0)
            g[out_ch] = _loguniform(rs, n_out, 0.02, 0.1)
            sd[p + keys[ln]] = g
        for ln in ("ln1_b", "ln2_b"):
            sd[p + keys[ln]] = sd[p + keys[ln]] * 5.0
        # the gains above multiply the GEMM's input by ~2.3 rms: keep the projections' OUTPUT scale where the init put it
        sd[p + keys["qkv_w"]] = sd[p + keys["qkv_w"]] / 2.3
        sd[p + keys["fc1_w"]] = sd[p + keys["fc1_w"]] / 2.3
        # residual writers: rows (output channels) out_ch of proj / fc2
        grow = scale * (0.5 if i else 1.0) / math.sqrt(layers)
        sd[p + keys["proj_w"]][out_ch] *= grow[:, None] * 0.3
        sd[p + keys["fc2_w"]][out_ch] *= grow[:, None] * 0.3
        sd[p + keys["fc2_b"]][out_ch[0]] += 20.0 if i == 0 else 2.0
        # two sharp heads: logits q.k / 8 with |q|, |k| scaled up
        hd = width // heads
        for h in rs.choice(heads, size=2, replace=False):
            f = math.sqrt(logit_target)
            sd[p + keys["qkv_w"]][h * hd:(h + 1) * hd] *= f
            sd[p + keys["qkv_w"]][width + h * hd: width + (h + 1) * hd] *= f
            sd[p + keys["qkv_b"]][h * hd:(h + 1) * hd] *= f
        j = rs.choice(4 * width, size=2, replace=False)
        sd[p + keys["fc1_b"]][j[0]] += 12.0
        sd[p + keys["fc1_b"]][j[1]] -= 12.0
    if pre_ln is not None:                  # CLIP's ln_pre: large gains on the outlier channels feed them into block 0
        g = _loguniform(rs, width, 0.05, 4.0)
        g[out_ch] = torch.from_numpy(rs.uniform(6.0, 8.0, size=n_out)).float()
        sd[pre_ln] = g
    return out_ch


def clip_outlier_state_dict(seed=0, layers=12):
    from scd_amd.clip import weights as W
    sd = W.synthetic_clip_state_dict(seed=seed, cfg=dict(v_layers=layers, t_layers=layers))
    vch = pathologise(sd, "visual.transformer.resblocks.", CLIP_KEYS, 768, 12, seed + 100, pre_ln="visual.ln_pre.weight",
                      emb_keys=("visual.class_embedding", "visual.positional_embedding"))
    tch = pathologise(sd, "transformer.resblocks.", CLIP_KEYS, 512, 8, seed + 200, emb_keys=("positional_embedding",))
    sd["token_embedding.weight"][:, tch] *= 30.0
    for k in ("visual.ln_post.weight", "ln_final.weight"):
        rs = np.random.RandomState(seed + 300 + len(k))
        sd[k] = _loguniform(rs, sd[k].numel(), 0.02, 8.0)
    return sd, vch, tch


def dino_outlier_state_dict(seed=1, layers=12):
    from scd_amd.clip import weights as W
    sd = W.synthetic_dino_state_dict(seed=seed, layers=layers)
    ch = pathologise(sd, "blocks.", DINO_KEYS, 768, 12, seed + 100, emb_keys=("cls_token", "pos_embed"))
    sd["patch_embed.proj.bias"][ch] += 3.0
    sd["norm.weight"] = _loguniform(np.random.RandomState(seed + 300), 768, 0.02, 8.0)
    return sd, ch


def round_like_the_device(sd, keep=("positional", "class_emb", "pos_embed", "cls_token")):
    """What the HIP towers hold: matrices in fp16, vectors (LayerNorm, biases) and embeddings' additive tables in fp32."""
    return {k: (v.half().float() if v.dim() >= 2 and not any(s in k for s in keep) else v) for k, v in sd.items()}


@torch.no_grad()
def residual_stream_stats(sd, kind, inputs):
    """fp32 oracle, block by block.  Returns dict(max_abs, median_abs, max_sumsq, max_abs_sum, max_logit): over every block's
    INPUT rows (what the LayerNorm-folded GEMMs read raw and the row statistics summarise)."""
    import torch.nn.functional as F
    if kind == "clip_visual":
        x = F.conv2d(inputs.float(), sd["visual.conv1.weight"].float(), stride=16)
        b, c = x.shape[:2]
        x = x.reshape(b, c, -1).permute(0, 2, 1)
        x = torch.cat([sd["visual.class_embedding"].float().expand(b, 1, c), x], dim=1) + sd["visual.positional_embedding"].float()
        x = co._ln(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
        prefix, keys, heads, act, eps, causal = "visual.transformer.resblocks.", CLIP_KEYS, 12, "quick_gelu", 1e-5, False
    elif kind == "clip_text":
        x = sd["token_embedding.weight"].float()[inputs.long()] + sd["positional_embedding"].float()
        prefix, keys, heads, act, eps, causal = "transformer.resblocks.", CLIP_KEYS, 8, "quick_gelu", 1e-5, True
    else:
        w = sd["patch_embed.proj.weight"].float()
        x = F.conv2d(inputs.float(), w, sd["patch_embed.proj.bias"].float(), stride=16).flatten(2).transpose(1, 2)
        x = torch.cat([sd["cls_token"].float().expand(x.shape[0], -1, -1), x], dim=1) + sd["pos_embed"].float()
        prefix, keys, heads, act, eps, causal = "blocks.", DINO_KEYS, 12, "gelu", 1e-6, False
    st = dict(max_abs=0.0, median_abs=[], max_sumsq=0.0, max_abs_sum=0.0, max_logit=0.0, max_hidden=0.0)
    for i in range(co._n_blocks(sd, prefix)):
        p = "%s%d." % (prefix, i)
        g = lambda n: sd[p + keys[n]]
        st["max_abs"] = max(st["max_abs"], x.abs().max().item())
        st["median_abs"].append(x.abs().median().item())
        st["max_sumsq"] = max(st["max_sumsq"], (x.double() ** 2).sum(-1).max().item())
        st["max_abs_sum"] = max(st["max_abs_sum"], x.double().sum(-1).abs().max().item())
        h = co._ln(x, g("ln1_w"), g("ln1_b"), eps)
        qkv = h @ g("qkv_w").float().t() + g("qkv_b").float()
        bsz, t, c = x.shape
        q, k, _ = qkv.view(bsz, t, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
        st["max_logit"] = max(st["max_logit"], ((q @ k.transpose(-2, -1)) * (c // heads) ** -0.5).abs().max().item())
        hid = co._ln(x + co._attention(h, g("qkv_w"), g("qkv_b"), g("proj_w"), g("proj_b"), heads, causal), g("ln2_w"), g("ln2_b"), eps)
        st["max_hidden"] = max(st["max_hidden"], (hid @ g("fc1_w").float().t() + g("fc1_b").float()).abs().max().item())
        x = co._block(x, g, heads, act, eps, causal)
    st["median_abs"] = float(np.median(st["median_abs"]))
    return st
