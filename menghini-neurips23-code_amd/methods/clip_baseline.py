"""Zero-shot CLIP baseline with the reference's class name, constructor and return tuple (methods/clip_baseline.py:17-96):
text features of the hand-written prompts once, batched image features, softmax / arg-max, and the per-image logits.
The reference calls `self.model(img, text)` per batch, i.e. re-encodes the C prompts for every batch; here they are
encoded once (same numbers: the text tower is deterministic)."""
import logging

import torch

from .. import clip
from ..engine import cosine_head

log = logging.getLogger(__name__)


class ClipBaseline(object):
    def __init__(self, config, label_to_idx, classes, seen_classes, unseen_classes, device):
        self.config = config
        self.classes, self.seen_classes, self.unseen_classes = classes, seen_classes, unseen_classes
        self.label_to_idx = label_to_idx
        self.device = device
        self.model, self.transform = clip.load(config.VIS_ENCODER, device=device)
        self.template = config.PROMPT_TEMPLATE

    @torch.no_grad()
    def test_predictions(self, data):
        """-> (df_predictions[id, class], images, predictions, prob_preds [N, C] logits on the CPU)."""
        import pandas as pd
        if getattr(data, "transform", None) is None and not hasattr(data, "_pool"):
            data.transform = self.transform
        loader = torch.utils.data.DataLoader(data, batch_size=int(self.config.BATCH_SIZE))
        prompts = [self.template.format(" ".join(i.split("_"))) for i in self.classes]
        log.info(f"Number of prompts: {len(prompts)}")
        text_features = self.model.encode_text(clip.tokenize(prompts).to(self.device))
        scale = self.model.logit_scale.exp().item()
        predictions, images, logits_all = [], [], []
        for batch in loader:
            img, img_path = batch[0], batch[-1]
            logits, _, _, am_p = cosine_head(self.model.encode_image(img.to(self.device)), text_features, scale)
            predictions += [self.classes[int(i)] for i in am_p.cpu()]
            images += [i for i in img_path]
            logits_all.append(logits)
        prob_preds = torch.cat(logits_all, dim=0).detach().to("cpu")
        return pd.DataFrame({"id": images, "class": predictions}), images, predictions, prob_preds
