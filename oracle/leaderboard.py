"""ORACLE (test infrastructure) -- pure-Python restatement of the sequential
per-class leaderboard of utils/clip_pseudolabels.py:49-112 (identical in the nine
`assign_pseudo_labels`, e.g. methods/transductive_zsl/multimodal_fpl.py:194-285).

Kept deliberately literal: Python lists, list.append, sorted(..., reverse=True)[:k]
on (score, path) tuples, and the descending walk over the other classes -- so that
it is obviously the same algorithm.  Scores are numpy float32 scalars (the
reference compares 0-d fp32 torch tensors).  Pinned against the reference function
itself by oracle/gen_golden.py -> tests/golden/leaderboard_*.json.
"""
import numpy as np

K_ALL = 10000000   # utils/clip_pseudolabels.py:27: "label everything by arg-max"


def leaderboard_scan(probs, pred_ids, paths, class_labels, k):
    """probs [N,C] float32, pred_ids [N] (arg-max the caller took, :39-41 / multimodal_fpl.py:222),
    paths: N strings in dataset order, class_labels: C global label ids in `classnames` order.
    Returns (filepaths, labels) exactly as :103-112 builds them."""
    probs = np.asarray(probs, dtype=np.float32)
    C = probs.shape[1]
    if k == K_ALL:                                                   # :27-44
        return list(paths), [class_labels[int(j)] for j in pred_ids]
    board = {class_labels[c]: [] for c in range(C)}                  # :49-51
    for i, path in enumerate(paths):
        pred_id = int(pred_ids[i])
        pred = class_labels[pred_id]
        score = probs[i, pred_id]
        if len(board[pred]) < k:                                     # :73-74
            board[pred].append((score, path))
        elif board[pred][-1][0] < score:                             # :75-82
            board[pred] = sorted(board[pred] + [(score, path)], reverse=True)[:k]
        else:                                                        # :83-101
            order = sorted([(probs[i, j], j) for j in range(C) if j != pred_id], reverse=True)
            for s, j in order:
                lab = class_labels[j]
                if len(board[lab]) < k:
                    board[lab].append((probs[i, j], path))
                elif board[lab][-1][0] < probs[i, j]:
                    board[lab] = sorted(board[lab] + [(probs[i, j], path)], reverse=True)[:k]
    new_imgs, new_labels = [], []
    for lab, lb in board.items():                                    # :103-109
        new_imgs += [t[1] for t in lb]
        new_labels += [lab for _ in lb]
    return new_imgs, new_labels


def softmax_argmax(logits):
    """utils/clip_pseudolabels.py:38-39: probs = softmax(logits), pred = argmax(probs) (first max wins)."""
    import torch
    lg = torch.as_tensor(np.asarray(logits, dtype=np.float32))
    p = lg.softmax(dim=-1)
    return p.numpy(), torch.argmax(p, dim=1).numpy()


def scan_margin(probs, pred_ids, k):
    """Decision margin of the scan above on `probs`: the smallest RELATIVE gap |a - b| / max(a, b) between two fp32 values
    whose ORDER the algorithm depends on -- (a) top-1 vs top-2 probability of an image (the arg-max, :39), (b) every strict
    comparison `board[-1].score < score` (:75, :95), (c) neighbours in every sorted board incl. the element the truncation
    drops (:79-82).  If another implementation's probabilities differ from `probs` by a relative error below half this
    margin everywhere, its lists are necessarily identical.  (Softmax outputs of two fp32 implementations differ by a
    roughly uniform RELATIVE error -- exp of a logit error -- so the margin is relative.  Order of the spill walk :83-90
    does not matter: boards are independent and there is no `break`.  Exact ties are resolved by the path string in both
    implementations alike and are skipped.)"""
    probs = np.asarray(probs, dtype=np.float32)
    n, C = probs.shape
    gap = np.inf

    def note(a, b):
        nonlocal gap
        a, b = float(a), float(b)
        d = abs(a - b)
        if d > 0.0:
            gap = min(gap, d / max(abs(a), abs(b)))

    if C > 1:
        top2 = np.sort(probs, axis=1)[:, -2:].astype(np.float64)
        d = top2[:, 1] - top2[:, 0]
        if (d > 0).any():
            gap = float(np.min((d / top2[:, 1])[d > 0]))
    if k == K_ALL:
        return gap
    board = [[] for _ in range(C)]

    def offer(j, score, i):
        b = board[j]
        if len(b) < k:
            b.append((score, i))
            return
        note(b[-1][0], score)
        if b[-1][0] < score:
            full = sorted(b + [(score, i)], reverse=True)
            for x, y in zip(full[:-1], full[1:]):
                note(x[0], y[0])
            board[j] = full[:k]

    for i in range(n):
        j0 = int(pred_ids[i])
        b = board[j0]
        if len(b) < k or b[-1][0] < probs[i, j0]:
            offer(j0, probs[i, j0], i)
        else:
            note(b[-1][0], probs[i, j0])
            for j in range(C):
                if j != j0:
                    offer(j, probs[i, j], i)
    return gap
