from .clip_pseudolabels import compute_pseudo_labels, pseudolabel_top_k  # noqa: F401
from .compute_metrics import (evaluate_predictions, load_parameters, save_parameters, save_predictions, save_pseudo_labels,  # noqa: F401
                              store_results)
