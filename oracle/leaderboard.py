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
