from .clip_pseudolabels import compute_pseudo_labels, pseudolabel_top_k  # noqa: F401
