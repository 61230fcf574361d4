"""`pseudolabel_top_k` / `compute_pseudo_labels` with the reference's signatures
(utils/clip_pseudolabels.py:13-156), cache file name and pickle schema, on the native engine."""
import logging
import os
import pickle

import torch

from .. import clip
from .. import pseudolabels as pl

log = logging.getLogger(__name__)


def _pool_images(dataset, transform, device):
    """Images of `dataset.filepaths` in order.  A dataset may carry a pre-decoded tensor pool as
    `dataset.images` ([N,3,R,R], aligned with filepaths); otherwise the files go through the native input pipeline
    (`ClipPreprocess.decode_chunk` / `finish_chunk`: parallel decode into a page-locked staging buffer, one upload and one
    batched launch pair per chunk), with the NEXT chunk decoding on a background thread while the current one is uploaded
    and encoded.  A foreign transform is applied per image as the reference does (utils/clip_pseudolabels.py:24-29)."""
    images = getattr(dataset, "images", None)
    if images is not None:
        return images
    from PIL import Image
    paths = list(dataset.filepaths)
    native_pre = hasattr(transform, "decode_chunk")
    workers = int(os.environ.get("GRIP_DECODE_WORKERS", str(min(32, os.cpu_count() or 8))))
    from ..data.decode import default_processes
    # decode processes: a number, or "auto" (default) = 1.5 per usable CPU for pools of >= 512 files (the thread back end for smaller
    # pools: starting two dozen interpreters costs more than decoding a few hundred images); "0" = threads only
    procs = os.environ.get("GRIP_DECODE_PROCS", "auto")
    procs = (default_processes() if len(paths) >= 512 else 0) if procs == "auto" else int(procs)

    class _Lazy:
        n = len(paths)

        def __init__(self):
            self._ahead = None          # ((lo, hi), future of the decoded chunk)
            self._stop = self.n
            self._bg = None

        def plan(self, lo, hi, chunk):
            """Called by the encoder with the range it is about to walk: the look-ahead never decodes past `hi`."""
            self._stop = hi

        def __call__(self, lo, hi):
            if not native_pre:
                return torch.stack([transform(Image.open(p).convert("RGB")) for p in paths[lo:hi]])
            if self._bg is None:
                from concurrent.futures import ThreadPoolExecutor
                self._bg = ThreadPoolExecutor(max_workers=1)
            if self._ahead is not None and self._ahead[0] == (lo, hi):
                handle = self._ahead[1].result()
            else:
                handle = transform.decode_chunk(paths[lo:hi], workers=workers, processes=procs)
            nlo, nhi = hi, min(hi + (hi - lo), self._stop)
            self._ahead = ((nlo, nhi), self._bg.submit(transform.decode_chunk, paths[nlo:nhi], workers, procs)) if nlo < nhi else None
            return transform.finish_chunk(handle)

        def take(self, idx):
            """A scattered selection (the rows a screen-and-refine pass re-encodes): decoded again, nothing is kept."""
            sel = [paths[int(i)] for i in idx]
            if not native_pre:
                return torch.stack([transform(Image.open(p).convert("RGB")) for p in sel])
            # a handful of rows (a refinement round, the audit): the thread back end -- a job round trip through two dozen decode processes costs more
            return transform.finish_chunk(transform.decode_chunk(sel, workers=workers, processes=procs if len(sel) >= 256 else 0))
    return _Lazy()


def compute_pseudo_labels(k, template, dataset, classnames, transform, clip_model, label_to_idx, device, filename,
                          chunk=880):
    prompts = [f"{template}{' '.join(i.split('_'))}" for i in classnames]     # reference :24 (literal "{}" kept)
    text = clip.tokenize(prompts).to(device)
    class_labels = [label_to_idx[c] for c in classnames]
    images = _pool_images(dataset, transform, device)
    scale = clip_model.logit_scale.exp().item()
    log.info(f"Compute {k} pseudo-labeles")
    if pl.mode() == "identical" and not getattr(clip_model, "exact", False) and hasattr(clip_model, "exact_twin"):
        # the fp32 scan's lists at close to f16 throughput: f16 screen, exact re-encode of the undecidable rows only
        twin = clip_model.exact_twin()
        with torch.no_grad():
            txt = twin.encode_text(text)                                       # once, not once per image
        new_imgs, new_labels = pl.identical_lists(clip_model.visual.tower, twin.visual.tower, images, txt, scale, list(dataset.filepaths),
                                                  class_labels, k, chunk=chunk, argmax_on="probs", visual_mid=pl.mid_tower(clip_model, len(dataset.filepaths)))
        st = pl.LAST_REFINE_STATS
        log.info(f"screen and refine: {st['rows_refined']} of {st['rows']} rows re-encoded ({st['rows_mid']} split-f16, {st['rows_exact']} f32) in {st['rounds']} rounds "
                 f"(bound {st['eps']:.2e}; audit {st['audit_rows']} rows, largest deviation {st['audit_max_deviation']:.2e})")
    else:
        with torch.no_grad():
            txt = clip_model.encode_text(text)                                 # once, not once per image
            emb = pl.encode_pool(clip_model.visual.tower, images, chunk=chunk)
        new_imgs, new_labels = pl.pseudolabel_from_features(emb, txt, scale, list(dataset.filepaths), class_labels, k, argmax_on="probs")
    dataset.filepaths = new_imgs
    dataset.labels = new_labels
    if getattr(dataset, "images", None) is not None:
        dataset.images = None   # the pool tensor no longer lines up with the rebuilt lists
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
    with open(filename, "wb") as f:
        pickle.dump({"filepaths": new_imgs, "labels": new_labels}, f)
    return dataset


def pseudolabel_top_k(config, data_name, k, template, dataset, classnames, transform, clip_model, label_to_idx, device,
                      vis_encoder, split_seed):
    filename = f"pseudolabels/{data_name}_{vis_encoder.replace('/', '')}_{config.LEARNING_PARADIGM}_{config.MODEL}_{k}_pseudolabels_split_{split_seed}.pickle"
    if os.path.exists(filename):
        with open(filename, "rb") as f:
            pseudolabels = pickle.load(f)
        dataset.filepaths = pseudolabels["filepaths"]
        dataset.labels = pseudolabels["labels"]
    else:
        dataset = compute_pseudo_labels(k, template, dataset, classnames, transform, clip_model, label_to_idx, device, filename)
    return dataset
