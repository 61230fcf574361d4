"""On-disk formats of the reference's `utils/compute_metrics.py` that touch the hot path's tensors:
`save_parameters` (:105-147: best prompts as a pickled list of numpy arrays; UPT sub-modules with torch.save),
`save_pseudo_labels` (:150-154) and `save_predictions` (:157-171), plus the two result helpers every `methods/main_*.py`
calls: `evaluate_predictions` (:18-56) and `store_results` (:58-103).  Same names, argument order, file names and schemas, so
existing analysis notebooks keep working."""
import json
import os
import pickle

import numpy as np
import torch


def evaluate_predictions(config, df_predictions, test_labeled_files, labels, unseen_classes, seen_classes=None):
    """(accuracy, None, None) for ul / ssl; (unseen_accuracy, seen_accuracy, harmonic_mean) for trzsl.
    df_predictions: columns "id" (file name) and "class"; joined with the ground truth on "id"."""
    import pandas as pd
    df_test = pd.DataFrame({"id": [f.split("/")[-1] for f in test_labeled_files], "true": labels})
    df = pd.merge(df_predictions, df_test, on="id")
    if config.LEARNING_PARADIGM in ("ul", "ssl"):
        return float(np.sum(df["class"] == df["true"]) / df.shape[0]), None, None
    un = df[df["true"].isin(unseen_classes)]
    se = df[df["true"].isin(seen_classes)]
    ua = float(np.sum(un["class"] == un["true"]) / un.shape[0])
    sa = float(np.sum(se["class"] == se["true"]) / se.shape[0])
    hm = 2.0 * ua * sa / (ua + sa) if ua + sa > 0 else 0.0       # = scipy.stats.hmean([ua, sa]) for positive values
    return ua, sa, hm


def store_results(obj_conf, std_response):
    """Append one JSON line to results_model_{MODEL}.json in the working directory: {"model", "config", "accuracy"} for
    ul / ssl, {"model", "config", "harmonic_mean", "seen_accuracy", "unseen_accuracy"} for trzsl (std_response as returned by
    evaluate_predictions).  Rank 0 only under torch.distributed."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_rank() != 0:
        return None
    conf = {k: v for k, v in obj_conf.__dict__.items() if isinstance(v, (str, int, float, bool, list, dict, type(None)))}
    if obj_conf.LEARNING_PARADIGM == "trzsl":
        res = {"model": obj_conf.MODEL, "config": conf, "harmonic_mean": std_response[2], "seen_accuracy": std_response[1],
               "unseen_accuracy": std_response[0]}
    else:
        res = {"model": obj_conf.MODEL, "config": conf, "accuracy": std_response[0]}
    fn = f"results_model_{obj_conf.MODEL}.json"
    with open(fn, "a") as f:
        f.write(json.dumps(res) + "\n")
    return fn


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
