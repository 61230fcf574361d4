"""On-disk formats of the reference's `utils/compute_metrics.py` that touch the hot path's tensors:
`save_parameters` (:105-147: best prompts as a pickled list of numpy arrays; UPT sub-modules with torch.save),
`save_pseudo_labels` (:150-154) and `save_predictions` (:157-171).  Same file names and schemas, so existing
analysis notebooks keep working."""
import os
import pickle

import torch


def save_parameters(obj, config, iteration=None):
    """obj: list of numpy arrays (textual / visual prompt) or a dict of UPT tensors."""
    os.makedirs("trained_prompts", exist_ok=True)
    enc = config.VIS_ENCODER.replace("/", "")
    tag = f"_iter_{iteration}" if iteration is not None else ""
    base = f"trained_prompts/{config.DATASET_NAME}_{config.LEARNING_PARADIGM}_{config.MODEL}_{enc}{tag}_opt_{config.OPTIM_SEED}_spl_{config.SPLIT_SEED}"
    if isinstance(obj, dict):
        torch.save(obj, base + ".pt")
        return base + ".pt"
    with open(base + ".pickle", "wb") as f:
        pickle.dump(obj, f)
    return base + ".pickle"


def load_parameters(path):
    if path.endswith(".pt"):
        return torch.load(path, map_location="cpu")
    with open(path, "rb") as f:
        return pickle.load(f)


def save_pseudo_labels(imgs, labs, config, iteration):
    os.makedirs("pseudolabels", exist_ok=True)
    enc = config.VIS_ENCODER.replace("/", "")
    fn = f"pseudolabels/{config.DATASET_NAME}_{config.LEARNING_PARADIGM}_{config.MODEL}_{enc}_iter_{iteration}_pseudolabels_spl_{config.SPLIT_SEED}.pickle"
    with open(fn, "wb") as f:
        pickle.dump({"filepaths": imgs, "labels": labs}, f)
    return fn


def save_predictions(obj, config, iteration=None):
    """obj = {"images", "predictions", "labels", "logits"} as methods/main_SSL.py:420-427 builds it."""
    os.makedirs("evaluation", exist_ok=True)
    enc = config.VIS_ENCODER.replace("/", "")
    tag = f"_iter_{iteration}" if iteration is not None else ""
    fn = f"evaluation/{config.DATASET_NAME}_{config.LEARNING_PARADIGM}_{config.MODEL}_{enc}{tag}_opt_{config.OPTIM_SEED}_spl_{config.SPLIT_SEED}.pickle"
    with open(fn, "wb") as f:
        pickle.dump(obj, f)
    return fn
