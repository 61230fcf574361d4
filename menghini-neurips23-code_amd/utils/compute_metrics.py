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
    from ..clip import clip as _clip      # what the run actually used: real or synthetic weights, BPE or the stand-in tokenizer
    conf["weights"], conf["tokenizer"] = _clip.PROVENANCE["weights"], _clip.PROVENANCE["tokenizer"]
    if obj_conf.LEARNING_PARADIGM == "trzsl":
        res = {"model": obj_conf.MODEL, "config": conf, "harmonic_mean": std_response[2], "seen_accuracy": std_response[1],
               "unseen_accuracy": std_response[0]}
    else:
        res = {"model": obj_conf.MODEL, "config": conf, "accuracy": std_response[0]}
    fn = f"results_model_{obj_conf.MODEL}.json"
    with open(fn, "a") as f:
        f.write(json.dumps(res) + "\n")
    return fn


UPT_NAMES = ["transformer", "proj_coop_pre", "proj_coop_post", "proj_vpt_pre", "proj_vpt_post", "coop_embeddings", "deep_vpt", "vpt_embeddings"]
UPT_TORCH_SAVED = set(UPT_NAMES[:5])     # state_dicts -> torch.save(.pt); the three prompt arrays -> pickle


def _prompt_base(config, iteration):
    enc = config.VIS_ENCODER.replace("/", "")
    tag = f"_iter_{iteration}" if iteration is not None else ""
    return f"trained_prompts/{config.DATASET_NAME}_{config.LEARNING_PARADIGM}_{config.MODEL}_{enc}{tag}_opt_{config.OPTIM_SEED}_spl_{config.SPLIT_SEED}"


def save_parameters(obj, config, iteration=None):
    """utils/compute_metrics.py:105-147 of the reference.  Textual / visual prompts: `obj` (a list holding the prompt as a
    numpy array) pickled to `{base}.pickle`.  MODALITY == "multi": `obj` is the positional list of the eight trainable pieces
    (UPT_NAMES order, methods/*/multimodal_prompt.py:149-158) and each goes to its own file -- the five state_dicts with
    torch.save to `{base}_{name}.pt`, the three prompt arrays pickled to `{base}_{name}.pickle`.  Returns the file(s) written."""
    os.makedirs("trained_prompts", exist_ok=True)
    base = _prompt_base(config, iteration)
    if getattr(config, "MODALITY", None) == "multi":
        files = []
        for name, piece in zip(UPT_NAMES, obj):
            if name in UPT_TORCH_SAVED:
                torch.save(piece, f"{base}_{name}.pt")
                files.append(f"{base}_{name}.pt")
            else:
                with open(f"{base}_{name}.pickle", "wb") as f:
                    pickle.dump(piece, f)
                files.append(f"{base}_{name}.pickle")
        return files
    with open(base + ".pickle", "wb") as f:
        pickle.dump(obj, f)
    return base + ".pickle"


def load_parameters(config_or_path, iteration=None):
    """Inverse of save_parameters: a path to one file, or a config (+ iteration) -> the list save_parameters was given."""
    if isinstance(config_or_path, str):
        path = config_or_path
        if path.endswith(".pt"):
            return torch.load(path, map_location="cpu")
        with open(path, "rb") as f:
            return pickle.load(f)
    base = _prompt_base(config_or_path, iteration)
    if getattr(config_or_path, "MODALITY", None) == "multi":
        return [load_parameters(f"{base}_{n}.pt" if n in UPT_TORCH_SAVED else f"{base}_{n}.pickle") for n in UPT_NAMES]
    return load_parameters(base + ".pickle")


def save_pseudo_labels(imgs, labs, config, iteration):
    """utils/compute_metrics.py:150-154 of the reference (file name includes OPTIM_SEED: runs with different optimisation
    seeds do not overwrite each other)."""
    os.makedirs("pseudolabels", exist_ok=True)
    enc = config.VIS_ENCODER.replace("/", "")
    fn = f"pseudolabels/{config.DATASET_NAME}_{config.LEARNING_PARADIGM}_{config.MODEL}_{enc}_iter_{iteration}_opt_{config.OPTIM_SEED}_spl_{config.SPLIT_SEED}.pickle"
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
