"""Drop-in for /root/reference/local_utils/clip_lang_util.py on libscd_hip.so: imagenet_templates (:13-94),
zeroshot_classifier (:96-108), get_wordnet_dict (:113-137), get_nouns (:139-149), accuracy (:151-154),
assign_name (:156-180), assign_name_on_leftover (:182-206), assign_name_logits (:208-234)."""
import os
from collections import defaultdict

import numpy as np
import torch

from .. import ops
from ..gcd.project_utils.cluster_utils import linear_assignment

imagenet_templates = [
    'a bad photo of a {}.',
    'a photo of many {}.',
    'a sculpture of a {}.',
    'a photo of the hard to see {}.',
    'a low resolution photo of the {}.',
    'a rendering of a {}.',
    'graffiti of a {}.',
    'a bad photo of the {}.',
    'a cropped photo of the {}.',
    'a tattoo of a {}.',
    'the embroidered {}.',
    'a photo of a hard to see {}.',
    'a bright photo of a {}.',
    'a photo of a clean {}.',
    'a photo of a dirty {}.',
    'a dark photo of the {}.',
    'a drawing of a {}.',
    'a photo of my {}.',
    'the plastic {}.',
    'a photo of the cool {}.',
    'a close-up photo of a {}.',
    'a black and white photo of the {}.',
    'a painting of the {}.',
    'a painting of a {}.',
    'a pixelated photo of the {}.',
    'a sculpture of the {}.',
    'a bright photo of the {}.',
    'a cropped photo of a {}.',
    'a plastic {}.',
    'a photo of the dirty {}.',
    'a jpeg corrupted photo of a {}.',
    'a blurry photo of the {}.',
    'a photo of the {}.',
    'a good photo of the {}.',
    'a rendering of the {}.',
    'a {} in a video game.',
    'a photo of one {}.',
    'a doodle of a {}.',
    'a close-up photo of the {}.',
    'a photo of a {}.',
    'the origami {}.',
    'the {} in a video game.',
    'a sketch of a {}.',
    'a doodle of the {}.',
    'a origami {}.',
    'a low resolution photo of a {}.',
    'the toy {}.',
    'a rendition of the {}.',
    'a photo of the clean {}.',
    'a photo of a large {}.',
    'a rendition of a {}.',
    'a photo of a nice {}.',
    'a photo of a weird {}.',
    'a blurry photo of a {}.',
    'a cartoon {}.',
    'art of a {}.',
    'a sketch of the {}.',
    'a embroidered {}.',
    'a pixelated photo of a {}.',
    'itap of the {}.',
    'a jpeg corrupted photo of the {}.',
    'a good photo of a {}.',
    'a plushie {}.',
    'a photo of the nice {}.',
    'a photo of the small {}.',
    'a photo of the weird {}.',
    'the cartoon {}.',
    'art of the {}.',
    'a drawing of the {}.',
    'a photo of the large {}.',
    'a black and white photo of a {}.',
    'the plushie {}.',
    'a dark photo of a {}.',
    'itap of a {}.',
    'graffiti of the {}.',
    'a toy {}.',
    'itap of my {}.',
    'a photo of a cool {}.',
    'a photo of a small {}.',
    'a tattoo of the {}.',
]


def zeroshot_classifier(classnames, templates, model, names_per_batch=1024, length_groups=4, min_group=2048):
    """[embed_dim, n_names] fp16 on the device: per name normalise(encode_text(prompts)) -> mean -> normalise, stacked
    along dim=1.  The reference runs one 80x77 forward per name; here names are batched (names_per_batch*len(templates)
    prompts per step), the prompts of a step are encoded in `length_groups` groups of similar length - each group only up to ITS
    longest prompt's EOT position (clip/model.py encode_text: the tower is causal, features are bit-identical whatever the
    grouping) - and the pooling is one fused kernel per step.  The host tokenisation of the next step runs behind the device work."""
    from .. import clip
    n, t = len(classnames), len(templates)
    out = None
    for s in range(0, n, names_per_batch):
        names = classnames[s:s + names_per_batch]
        # the prompts [template.format(c) for c in names for template in templates], ids on the host
        tok = clip.tokenize_templates(names, templates)
        groups = min(length_groups, tok.shape[0] // min_group)      # a group below ~2k prompts no longer fills the GEMMs
        if getattr(model, "_dev", None) is None or groups <= 1:
            emb = model.encode_text(tok)
        else:
            eot = tok.numpy().argmax(axis=-1)           # (numpy on the host: a torch CPU op wakes the whole intra-op pool)
            order = np.argsort(eot, kind="stable")
            # ONE upload per step (ids + permutation): a pageable copy blocks the host until the stream reaches it, and per
            # group that would serialise this step's encodes with the next step's tokenisation
            dev_in = torch.from_numpy(np.concatenate([tok.numpy().reshape(-1), order.astype(np.int32)])).to(model._dev)
            tok_d = dev_in[:tok.numel()].reshape(tok.shape)
            order_d = dev_in[tok.numel():].to(torch.int64)
            emb = None
            for part, part_d in zip(np.array_split(order, groups), torch.tensor_split(order_d, groups)):
                e = model.encode_text(tok_d.index_select(0, part_d), ctx_len=int(eot[part].max()) + 1)
                if emb is None:
                    emb = torch.empty((tok.shape[0], e.shape[1]), dtype=e.dtype, device=e.device)
                emb.index_copy_(0, part_d, e)
        if out is None:
            out = torch.empty((emb.shape[1], n), dtype=torch.float16, device=emb.device)
        ops.prompt_pool(emb.contiguous(), len(names), t, out, s)
    return out


def zeroshot_classifier_sharded(classnames, templates, model, group, names_per_batch=256, build=None):
    """The vocabulary sharded over the ranks of `group` (one process per GPU): rank r builds the classifier columns of the
    contiguous name range [r*ceil(n/W), (r+1)*ceil(n/W)) with `zeroshot_classifier`, then ONE all-gather (RCCL over xGMI;
    name-major rows, the short last shard padded) gives every rank the full [embed_dim, n_names] matrix in the original
    name order.  Not in the reference (it has no multi-GPU path): BASELINE config 5, 100k names x 80 prompts = 8 M text
    forward passes, is the largest compute item once the vocabulary is open, and it shards without any other exchange.
    `build(names, templates, model, names_per_batch)` defaults to zeroshot_classifier (a hook for the CPU gloo test)."""
    import torch.distributed as dist
    build = build or zeroshot_classifier
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(classnames)
    per = (n + world - 1) // world
    mine = classnames[rank * per:(rank + 1) * per]
    local = build(mine, templates, model, names_per_batch) if len(mine) else None          # [D, n_mine]
    if local is not None:
        dev = local.device
    else:   # a rank without names still takes part in the collectives: on the device the backend moves (RCCL: the GPU, gloo: the host)
        dev = torch.device("cuda", torch.cuda.current_device()) if str(dist.get_backend(group)) == "nccl" else torch.device("cpu")
    dvec = torch.tensor([0 if local is None else local.shape[0]], dtype=torch.int64, device=dev)
    dims = [torch.empty_like(dvec) for _ in range(world)]
    dist.all_gather(dims, dvec, group=group)
    dim = max(int(x) for x in dims)
    dtype = local.dtype if local is not None else torch.float16
    pad = torch.zeros((per, dim), dtype=dtype, device=dev)
    if local is not None:
        pad[: local.shape[1]] = local.t()
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat(parts, dim=0)[:n].t().contiguous()


def _data_file(name):
    for root in (os.environ.get("SCD_DATA", ""), os.path.join(os.environ.get("SCD_ROOT", ""), "data")):
        p = os.path.join(root, name)
        if root and os.path.exists(p):
            return p
    raise FileNotFoundError("%s not found under $SCD_DATA or $SCD_ROOT/data (the reference hard-codes "
                            "/disk/work/xhhuang/scd_v1/language_ncd_yandong/data, clip_lang_util.py:141-148)" % name)


def get_nouns(corpus='wordnet'):
    fname = {'wordnet': 'wordnet_all_noun.txt', 'wikibird': 'wiki_birdclass_names.txt',
             'wikidog': 'wiki_dogclass_names.txt'}[corpus]
    with open(_data_file(fname)) as f:
        return [line.rstrip('\n') for line in f]


def get_wordnet_dict():
    from nltk.corpus import wordnet as wn          # optional dependency, host-only metadata
    wnid_to_synset, wnid_to_name, name_to_wnids = {}, {}, defaultdict(list)
    for n in wn.all_synsets('n'):
        wnid = "n{:08d}".format(n.offset())
        wnid_to_synset[wnid] = n
        name = n.lemma_names()[0].lower().replace('-', '_')
        wnid_to_name[wnid] = name
        name_to_wnids[name].append(wnid)
    return wnid_to_synset, wnid_to_name, name_to_wnids


def accuracy(output, target, topk=(1,)):
    """Counts (not %) of targets within the top-k logits (evaluation helper, host-side glue)."""
    pred = output.topk(max(topk), 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [float(correct[:k].reshape(-1).float().sum(0, keepdim=True).cpu().numpy()) for k in topk]


def _assign(unique_name_idx, cluster_to_counter, picker):
    """w[i, col(name)] += count; ind = linear_assignment(w.max() - w) (:167-178).  The solve runs on the non-zero entries
    (scd_munkres_sparse: the reference's state machine and tie-breaking with implicit potentials), so D = 10,000-20,000 at
    K = 1000 clusters costs a second instead of the O(D^3) of the dense matrix; `w` is still returned, its untouched pages
    never materialise."""
    col = {uidx: nidx for nidx, uidx in enumerate(unique_name_idx)}
    keys = list(cluster_to_counter.keys())
    D = max(len(unique_name_idx), len(keys))
    w = np.zeros((D, D), dtype=int)
    rows, cols, vals = [], [], []
    for i, ck in enumerate(keys):
        for k, v in picker(cluster_to_counter[ck]):
            w[i, col[k]] += v
            rows.append(i)
            cols.append(col[k])
            vals.append(v)
    if not all(isinstance(v, (int, np.integer)) for v in vals):      # assign_name_logits: float "counts" go through w's int cast
        return linear_assignment(w.max() - w), w
    return ops.munkres_sparse(D, rows, cols, vals), w


def assign_name(unique_name_idx, cluster_to_counter, num_common=4):
    return _assign(unique_name_idx, cluster_to_counter, lambda ct: ct.most_common(num_common))


def assign_name_on_leftover(unique_name_idx, cluster_to_counter, voted_unique_name_idx):
    return _assign(unique_name_idx, cluster_to_counter,
                   lambda ct: [(k, v) for k, v in ct.most_common(5) if k not in voted_unique_name_idx])


def assign_name_logits(unique_name_idx, cluster_to_logitcounter):
    return _assign(unique_name_idx, cluster_to_logitcounter,
                   lambda ct: sorted(ct.items(), key=lambda kv: kv[1], reverse=True)[:4])
