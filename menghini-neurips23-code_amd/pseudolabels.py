"""Pseudolabel engine: batched pool encode (sharded over ranks) -> cached text features ->
fused head -> exact sequential leaderboard.  This is the compute behind
utils.clip_pseudolabels.{compute_pseudo_labels, pseudolabel_top_k} and the nine
assign_pseudo_labels of the reference; those keep their signatures in grip_amd.utils / grip_amd.methods.

Two algorithmic changes against the reference loop (utils/clip_pseudolabels.py:31-61), neither of
which changes a result: the class prompts are encoded once instead of once per image, and images
go through the tower in chunks instead of one by one.
"""
import numpy as np
import torch

from . import dist as gdist
from . import engine

K_ALL = 10000000   # utils/clip_pseudolabels.py:27


def path_ranks(paths):
    """Dense rank of every path among all paths under Python's string order (the leaderboard breaks
    score ties by comparing the path strings, utils/clip_pseudolabels.py:79-82)."""
    order = {p: i for i, p in enumerate(sorted(set(paths)))}
    return np.fromiter((order[p] for p in paths), dtype=np.int64, count=len(paths))


@torch.no_grad()
def encode_pool(visual_tower, images, chunk=880, prefix=None, out=None):
    """Encode an ordered pool.  `images` is a tensor [N,3,R,R] (any device) or a callable
    (lo, hi) -> tensor for that slice.  With torch.distributed initialised the pool is sharded
    contiguously and the embeddings are all-gathered; returns [N, E] f32 on the device."""
    n = images.shape[0] if torch.is_tensor(images) else images.n
    lo, hi, per = gdist.shard_range(n)
    dev = visual_tower.device
    local = torch.empty(max(hi - lo, 0), visual_tower.embed_dim, dtype=torch.float32, device=dev)
    visual_tower.encode_chunks(images, local, lo, hi, chunk, prefix)
    return gdist.allgather_rows(local, n, per)


def leaderboard(probs, pred, paths, class_labels, k):
    """(filepaths, labels) exactly as utils/clip_pseudolabels.py:103-112 rebuilds them."""
    if k == K_ALL:
        return list(paths), [class_labels[int(j)] for j in pred]
    img, cls = engine.leaderboard_scan(probs, pred, path_ranks(paths), k)
    return [paths[i] for i in img], [class_labels[int(c)] for c in cls]


@torch.no_grad()
def pseudolabel_from_features(img_emb, txt_emb, scale, paths, class_labels, k, argmax_on="probs"):
    """Head + scan.  argmax_on: "probs" (compute_pseudo_labels, :39) or "logits" (assign_pseudo_labels)."""
    logits, probs, am_l, am_p = engine.cosine_head(img_emb, txt_emb, scale)
    pred = (am_p if argmax_on == "probs" else am_l).cpu().numpy()
    return leaderboard(probs.cpu().numpy(), pred, paths, class_labels, k)
