"""Entry points `run_main_{ssl,ul,trzsl}.py` -> main(paradigm): the same CLI and environment variables as
the reference pipelines (methods/main_SSL.py:430-505: --model_config, --learning_paradigm; OPTIM_SEED,
VIS_ENCODER, DATASET_NAME, SPLIT_SEED, MODEL, DATASET_DIR), the same 12-way MODEL dispatch, on the native
engine.  The dataset readers of the reference (`utils/prepare_data.py`, per-dataset file layouts, PIL) are
out of scope; `--synthetic N` builds a class-structured synthetic pool instead (DATASET_DIR unset), which is
what the offline boxes can run."""
import argparse
import json
import logging
import os
import random

import numpy as np
import torch
import yaml

from .. import config as gcfg, rng
from ..data import ImagePool, TensorPoolDataset
from ..utils import evaluate_predictions, save_parameters, save_predictions, store_results
from . import strategies as S

log = logging.getLogger(__name__)

DEFAULTS = dict(MODALITY="text", PREFIX_SIZE=16, TEXT_PREFIX_SIZE=4, VISION_PREFIX_SIZE=4, TRANSFORMER_DIM=128, MEAN_INIT=0, VAR_INIT=0.02,
                VIS_PREFIX_INIT="normal", N_LABEL=2, N_PSEUDOSHOTS=16, STEP_QUANTILE=10, BATCH_SIZE=16, EPOCHS=150, WARMUP_EPOCHS=5,
                ACCUMULATION_ITER=1, LR=0.1, DECAY=0.1, validation_seed=0, ratio_train_val=0.8, PROMPT_TEMPLATE="a photo of a {}")

MODELS = {   # MODEL env value -> (strategy class, training method) -- the dispatch of methods/main_SSL.py:203-396
    "textual_prompt": (S.TextualPrompt, "train"), "visual_prompt": (S.VisualPrompt, "train"), "multimodal_prompt": (S.MultimodalPrompt, "train"),
    "textual_fpl": (S.TextualFPL, "train"), "visual_fpl": (S.VisualFPL, "train"), "multimodal_fpl": (S.MultimodalFPL, "train"),
    "iterative_textual_fpl": (S.TextualFPL, "fixed_iterative_train"), "iterative_visual_fpl": (S.VisualFPL, "fixed_iterative_train"),
    "iterative_multimodal_fpl": (S.MultimodalFPL, "fixed_iterative_train"),
    "grip_textual": (S.TextualFPL, "grip_train"), "grip_visual": (S.VisualFPL, "grip_train"), "grip_multimodal": (S.MultimodalFPL, "grip_train"),
}


class Config:
    def __init__(self, **entries):
        self.__dict__.update(entries)


def split_classes(classes, seed, seen_fraction=0.62):
    """Seeded seen/unseen split (utils/prepare_data.py:84-185 uses a 62 % seen share)."""
    r = np.random.RandomState(seed)
    perm = r.permutation(len(classes))
    n_seen = int(round(seen_fraction * len(classes)))
    seen = [classes[i] for i in sorted(perm[:n_seen])]
    unseen = [classes[i] for i in sorted(perm[n_seen:])]
    return seen, unseen


def synthetic_pool(n_classes, n_per_class, res, seed):
    """Class-structured images: a per-class colour/gradient signature plus noise."""
    classes = [f"class_{i:03d}" for i in range(n_classes)]
    sig = torch.from_numpy(rng.normal(seed, rng.stream_id("syn.sig"), (n_classes, 3, 4, 4)))
    sig = torch.nn.functional.interpolate(sig, size=(res, res), mode="bilinear", align_corners=False) * 1.5
    noise = torch.from_numpy(rng.normal(seed, rng.stream_id("syn.noise"), (n_classes * n_per_class, 3, res, res))) * 0.5
    labels = [c for c in range(n_classes) for _ in range(n_per_class)]
    images = sig[torch.tensor(labels)] + noise
    files = [f"{classes[l]}_{i:05d}.jpg" for i, l in enumerate(labels)]
    return classes, files, images, [classes[l] for l in labels]


def workflow(obj_conf, device, n_synth, n_classes=10):
    paradigm = obj_conf.LEARNING_PARADIGM
    d = gcfg.get_dims(obj_conf.VIS_ENCODER)
    classes, files, images, names = synthetic_pool(n_classes, n_synth, d.image_resolution, int(obj_conf.SPLIT_SEED))
    images = images.to(device)
    if paradigm == "trzsl":
        seen, unseen = split_classes(classes, int(obj_conf.SPLIT_SEED))
    else:
        seen, unseen = classes, classes
    label_to_idx = {c: i for i, c in enumerate(classes)}
    idx = np.arange(len(files))
    r = np.random.RandomState(int(obj_conf.validation_seed))
    test = set(r.choice(idx, size=len(idx) // 5, replace=False).tolist())
    labeled, unlabeled = [], []
    per_class = {c: 0 for c in classes}
    for i in idx:
        if i in test:
            continue
        c = names[i]
        if c in seen and paradigm != "ul" and per_class[c] < (int(obj_conf.N_LABEL) if paradigm == "ssl" else 10 ** 9):
            labeled.append(i)
            per_class[c] += 1
        elif paradigm != "trzsl" or c in unseen:
            unlabeled.append(i)
    pool = ImagePool(files, images)
    sub = lambda ids, lab: TensorPoolDataset([files[i] for i in ids], pool, labels=[names[i] for i in ids] if lab else None, label_map=label_to_idx)
    n_val = max(1, len(labeled) // 5) if labeled else 0
    train_data, val_data = sub(labeled[n_val:], True), sub(labeled[:n_val], True)
    unlabeled_data = sub(unlabeled, False)
    test_ids = sorted(test)
    test_data = sub(test_ids, False)

    if obj_conf.MODEL == "clip_baseline":                  # methods/main_CLIP.py: zero-shot CLIP, nothing to train
        from .clip_baseline import ClipBaseline
        model = ClipBaseline(obj_conf, label_to_idx, classes, seen, unseen, device)
        df, images_e, preds_e, logits_e = model.test_predictions(test_data)
        truth = {files[i]: names[i] for i in test_ids}
        std_response = evaluate_predictions(obj_conf, df, [files[i] for i in test_ids], [names[i] for i in test_ids], unseen, seen)
        store_results(obj_conf, std_response)
        save_predictions({"images": images_e, "predictions": preds_e, "labels": [truth[i] for i in images_e], "logits": logits_e}, obj_conf)
        result = {"model": obj_conf.MODEL, "paradigm": paradigm, "encoder": obj_conf.VIS_ENCODER, "val_accuracy": 0.0,
                  "test_accuracy": float(np.mean([truth[i] == c for i, c in zip(df["id"], df["class"])])),
                  "n_train": 0, "n_unlabeled": len(unlabeled), "n_test": len(test_ids)}
        if paradigm == "trzsl":
            result.update(unseen_accuracy=std_response[0], seen_accuracy=std_response[1], harmonic_mean=std_response[2])
        return result
    cls, method = MODELS[obj_conf.MODEL]
    fpl = cls.fpl
    args = (obj_conf, label_to_idx) + ((obj_conf.DATASET_DIR,) if fpl else ()) + (classes, seen, unseen, device)
    model = cls(*args)
    only_seen = paradigm == "trzsl" and not fpl
    if method == "train":
        val_acc, prompt = model.train(train_data, val_data, unlabeled_data if fpl else None, only_seen=only_seen)
    else:
        val_acc, prompt = getattr(model, method)(train_data, val_data, unlabeled_data, only_seen=only_seen)
    save_parameters(prompt, obj_conf)                      # methods/main_SSL.py:400
    df = model.test_predictions(test_data, standard_zsl=False)
    truth = {files[i]: names[i] for i in test_ids}
    std_response = evaluate_predictions(obj_conf, df, [files[i] for i in test_ids], [names[i] for i in test_ids], unseen, seen)   # methods/main_SSL.py:404-414
    store_results(obj_conf, std_response)
    acc = float(np.mean([truth[i] == c for i, c in zip(df["id"], df["class"])]))
    result = {"model": obj_conf.MODEL, "paradigm": paradigm, "encoder": obj_conf.VIS_ENCODER, "val_accuracy": val_acc, "test_accuracy": acc,
              "n_train": len(train_data), "n_unlabeled": len(unlabeled), "n_test": len(test_ids)}
    images_e, preds_e, logits_e = model.evaluation(test_data)   # methods/main_SSL.py:418-427
    save_predictions({"images": images_e, "predictions": preds_e, "labels": [truth[i] for i in images_e], "logits": logits_e}, obj_conf)
    if paradigm == "trzsl":
        s = [truth[i] == c for i, c in zip(df["id"], df["class"]) if truth[i] in seen]
        u = [truth[i] == c for i, c in zip(df["id"], df["class"]) if truth[i] in unseen]
        sa, ua = float(np.mean(s)) if s else 0.0, float(np.mean(u)) if u else 0.0
        result.update(seen_accuracy=sa, unseen_accuracy=ua, harmonic_mean=(2 * sa * ua / (sa + ua)) if sa + ua else 0.0)
    return result


def main(paradigm, model=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--model_config", type=str, default=None, help="Name of model config file")
    parser.add_argument("--learning_paradigm", type=str, default=paradigm, help="ssl / ul / trzsl")
    parser.add_argument("--synthetic", type=int, default=40, help="synthetic images per class (datasets are not available offline)")
    parser.add_argument("--classes", type=int, default=10)
    args = parser.parse_args()
    conf = dict(DEFAULTS)
    if args.model_config:
        path = args.model_config if os.path.exists(args.model_config) else os.path.join("methods_config", args.model_config)
        with open(path) as f:
            conf.update({k: v for k, v in yaml.safe_load(f).items() if not (isinstance(v, str) and v.startswith("$"))})
    conf.update(OPTIM_SEED=int(os.environ.get("OPTIM_SEED", 1)), VIS_ENCODER=os.environ.get("VIS_ENCODER", "ViT-B/16"),
                DATASET_NAME=os.environ.get("DATASET_NAME", "Synthetic"), SPLIT_SEED=int(os.environ.get("SPLIT_SEED", 500)),
                MODEL=os.environ.get("MODEL", conf.get("MODEL", "textual_prompt")), DATASET_DIR=os.environ.get("DATASET_DIR", ""),
                LEARNING_PARADIGM=args.learning_paradigm)
    if isinstance(conf.get("MODEL"), str) and conf["MODEL"].startswith("$"):
        conf["MODEL"] = "textual_prompt"
    if model:
        conf["MODEL"] = model
    for k in ("EPOCHS", "BATCH_SIZE", "N_PSEUDOSHOTS", "STEP_QUANTILE", "N_LABEL"):
        if k in os.environ:
            conf[k] = int(os.environ[k])
    obj_conf = Config(**conf)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    np.random.seed(obj_conf.OPTIM_SEED)
    random.seed(obj_conf.OPTIM_SEED)
    torch.manual_seed(obj_conf.OPTIM_SEED)
    device = "cuda" if torch.cuda.is_available() else "cpu"
    result = workflow(obj_conf, device, args.synthetic, args.classes)
    os.makedirs("results", exist_ok=True)
    with open(f"results/results_model_{obj_conf.MODEL}.json", "a") as f:
        f.write(json.dumps(result) + "\n")
    print(json.dumps(result))
    return result
