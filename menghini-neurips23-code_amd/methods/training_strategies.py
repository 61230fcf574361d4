"""Stand-in for the reference's MISSING `methods/*/training_strategies.py` (every strategy file does
`from .training_strategies import TrainingStrategy`, but the file is absent upstream: SURVEY.md 0.2).
The contract is reconstructed from the call sites and from `pseudo_iterative.py:62-125`
(GRIP schedule).  Everything that cannot be pinned from the snapshot is an explicit config knob with a
documented default: prompt-init seed (`OPTIM_SEED`), `MOMENTUM` (0), loader shuffle seed, best-epoch rule
(highest validation accuracy, first wins).  Training trajectories are therefore NOT claimed to match
upstream bit for bit; the per-step numerics are the parity-tested kernels.

One class covers the three prompt modalities (config.MODALITY in {"text", "image", "multi"}) and the three
learning paradigms (config.LEARNING_PARADIGM in {"ssl", "ul", "trzsl"}); the reference's eighteen strategy
classes are thin aliases of it (methods/strategies.py).
"""
import copy
import logging
import math

import numpy as np
import torch

from .. import clip, dist as gdist, pseudolabels as pl, steps
from ..engine import cosine_head
from ..models import CustomImageEncoder, CustomTextEncoder, ImagePrefixModel, TextPrefixModel, UPTModel
from ..utils import pseudolabel_top_k

log = logging.getLogger(__name__)


def make_scheduler(optimizer, config):
    """WarmupCosineSchedule stepped once per EPOCH (utils/schedulers.py:36-65 of the reference)."""
    warm, total = int(getattr(config, "WARMUP_EPOCHS", 0)), int(config.EPOCHS)

    def lr_lambda(step):
        if step < warm:
            return float(step) / float(max(1.0, warm))
        progress = float(step - warm) / float(max(1, total - warm))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


class TrainingStrategy:
    def __init__(self, config, label_to_idx, classes, seen_classes, unseen_classes, device, data_folder=None):
        self.config = config
        self.classes, self.seen_classes, self.unseen_classes = classes, seen_classes, unseen_classes
        self.label_to_idx = label_to_idx
        self.device = device
        self.data_folder = data_folder
        self.clip_model, self.transform = clip.load(config.VIS_ENCODER, device=device)
        self.template = config.PROMPT_TEMPLATE
        self.modality = config.MODALITY
        self.paradigm = getattr(config, "LEARNING_PARADIGM", "ssl")
        self.val_unseen_files = self.val_unseen_labs = None
        self.balance_param = 1.0
        self.check_unlabeled = set()
        self.declare_custom_encoder()
        self.initialize_prompts_parameters()

    # ------------------------------------------------------------------ encoders / prompts / model
    def declare_custom_encoder(self):
        self.image_encoder = CustomImageEncoder(self.clip_model.visual) if self.modality in ("image", "multi") else None
        self.text_encoder = CustomTextEncoder(self.clip_model, self.device, torch.float32) if self.modality in ("text", "multi") else None

    def initialize_prompts_parameters(self):
        c = self.config
        g = torch.Generator().manual_seed(int(getattr(c, "OPTIM_SEED", 0)))
        mean, std = float(getattr(c, "MEAN_INIT", 0.0)), float(getattr(c, "VAR_INIT", 0.02))
        d = self.clip_model.dims

        def init(*shape):
            if getattr(c, "VIS_PREFIX_INIT", "normal") == "uniform":
                return (torch.rand(*shape, generator=g) * 2 - 1) * std + mean
            return torch.randn(*shape, generator=g) * std + mean
        if self.modality == "text":
            self.initial_prefix = init(1, int(c.PREFIX_SIZE), d.transformer_width)
        elif self.modality == "image":
            self.initial_prefix = init(int(c.PREFIX_SIZE), d.vision_width)
        else:
            self.coop_init = init(1, int(c.TEXT_PREFIX_SIZE), d.transformer_width)
            self.vpt_init = init(1, int(c.VISION_PREFIX_SIZE), d.vision_width)

    def define_model(self, classes=None):
        c, dev = self.config, self.device
        classes = classes if classes is not None else self.classes
        if self.modality == "text":
            self.model = TextPrefixModel(self.initial_prefix.clone().to(dev), self.text_encoder, classes, device=dev)
        elif self.modality == "image":
            self.model = ImagePrefixModel(self.initial_prefix.clone().to(dev), self.image_encoder, device=dev)
        else:
            torch.manual_seed(int(getattr(c, "OPTIM_SEED", 0)))
            self.model = UPTModel(self.coop_init.clone().to(dev), self.vpt_init.clone().to(dev), None, self.image_encoder,
                                  self.text_encoder, classes, int(getattr(c, "TRANSFORMER_DIM", 128)), device=dev, dtype=torch.float32)
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.optimizer = torch.optim.SGD(params, lr=float(c.LR), weight_decay=float(c.DECAY), momentum=float(getattr(c, "MOMENTUM", 0.0)))
        self.scheduler = make_scheduler(self.optimizer, c)
        self.loss_func = torch.nn.CrossEntropyLoss()
        self._model_gen = getattr(self, "_model_gen", 0) + 1       # a new model / optimizer: captured step graphs of the old one are stale

    def unwrap_model(self):
        return self.model

    def training_model(self, img):
        return self.model(img)

    def backpropagate(self):
        self.optimizer.step()
        self.optimizer.zero_grad()

    def update_scheduler(self):
        self.scheduler.step()

    def prompt_snapshot(self):
        m = self.unwrap_model()
        if self.modality in ("text", "image"):
            return [m.prefix.detach().cpu().numpy()]
        # multimodal_prompt.py:149-158 of the reference: the eight trainable pieces, positionally -- NOT the whole UPTModel
        # state_dict, which would drag ~1 GB of frozen CLIP weights (registered sub-modules) to the host every epoch.
        sd = lambda mod: copy.deepcopy({k: v.detach().cpu() for k, v in mod.state_dict().items()})   # noqa: E731
        return [sd(m.transformer), sd(m.proj_coop_pre), sd(m.proj_coop_post), sd(m.proj_vpt_pre), sd(m.proj_vpt_post),
                m.coop_embeddings.detach().cpu().numpy(),
                None if m.vpt_embeddings_deep is None else m.vpt_embeddings_deep.detach().cpu().numpy(),
                m.vpt_embeddings.detach().cpu().numpy()]

    # ------------------------------------------------------------------ features
    def text_prompts(self, classes):
        return [self.template.format(" ".join(i.split("_"))) for i in classes]

    @torch.no_grad()
    def fixed_text_features(self, classes):
        return self.clip_model.encode_text(clip.tokenize(self.text_prompts(classes)).to(self.device))

    def scale(self):
        return self.clip_model.logit_scale.exp().item()

    def frozen_image_features(self, images, names=None):
        """Features of the FROZEN image tower.  Textual strategies re-encode the same images every epoch upstream
        (e.g. textual_prompt.py:100: 150 epochs x the same shots); the tower is deterministic and row-independent, so the
        features are cached per file name (`CACHE_FROZEN_FEATURES: False` restores the re-encode)."""
        if names is None or not getattr(self.config, "CACHE_FROZEN_FEATURES", True):
            with torch.no_grad():
                return self.clip_model.encode_image(images)
        cache = self.__dict__.setdefault("_frozen_cache", {})
        miss = [i for i, n in enumerate(names) if n not in cache]
        if miss:
            with torch.no_grad():
                f = self.clip_model.encode_image(images[miss])
            for j, i in enumerate(miss):
                cache[names[i]] = f[j]
        return torch.stack([cache[n] for n in names])

    def features(self, images, classes, names=None):
        """(image_features, text_features) of the CURRENT model, autograd-ready where prompts are involved."""
        if self.modality == "text":
            self.model.classes = classes
            return self.frozen_image_features(images, names), self.model(classes)
        if self.modality == "image":
            return self.model(images), self.fixed_text_features(classes)
        self.model.classes = classes
        txt, img = self.model(images, classes)
        return img, txt

    # ------------------------------------------------------------------ datasets
    def create_training_dataset(self, train_data, unlabeled_data=None):
        """Plain strategies train on the labeled seen data as is; FPL strategies (subclass hook `fpl`)
        add CLIP pseudolabels for the unlabeled pool (e.g. textual_fpl.py:51-121 of each paradigm)."""
        if not getattr(self, "fpl", False) or unlabeled_data is None:
            return train_data
        c = self.config
        pseudo_classes = self.classes if self.paradigm in ("ssl", "ul") else self.unseen_classes
        pseudolabel_top_k(c, c.DATASET_NAME, int(c.N_PSEUDOSHOTS), self.template, unlabeled_data, pseudo_classes, self.transform,
                          self.clip_model, self.label_to_idx, self.device, c.VIS_ENCODER, getattr(c, "SPLIT_SEED", 0))
        return self.merge_pseudolabels(train_data, unlabeled_data)

    def merge_pseudolabels(self, train_data, unlabeled_data):
        c = self.config
        unseen_imgs, unseen_labs = list(unlabeled_data.filepaths), [int(l) for l in unlabeled_data.labels]
        if int(c.N_PSEUDOSHOTS) >= 10:     # hold out part of the pseudolabels for validation (textual_fpl.py:84-103)
            np.random.seed(int(getattr(c, "validation_seed", 0)))
            tr = np.random.choice(range(len(unseen_imgs)), size=int(len(unseen_imgs) * float(c.ratio_train_val)), replace=False)
            va = sorted(set(range(len(unseen_imgs))).difference(set(tr.tolist())))
            self.val_unseen_files = [unseen_imgs[i] for i in va]
            self.val_unseen_labs = [unseen_labs[i] for i in va]
            unseen_imgs, unseen_labs = [unseen_imgs[i] for i in tr], [unseen_labs[i] for i in tr]
        else:
            self.val_unseen_files = self.val_unseen_labs = None
        self.check_unlabeled = set(p.split("/")[-1] for p in unseen_imgs)
        if self.paradigm == "ul":
            train_data.filepaths, train_data.labels = unseen_imgs, unseen_labs
            self.balance_param = 1.0
        else:
            seen_imgs = list(train_data.filepaths)
            seen_labs = [l if train_data.label_id else self.label_to_idx[l] for l in train_data.labels]
            n_u, n_s = max(len(unseen_imgs), 1), max(len(seen_imgs), 1)
            # ssl: |unseen| / |seen| (semi_supervised_learning/textual_fpl.py:115, visual_fpl.py:110), trzsl: |seen| / |unseen|
            # (transductive_zsl/textual_fpl.py:109, visual_fpl.py:105); the multimodal strategies take the square root in both
            # paradigms (semi_supervised_learning/multimodal_fpl.py:107, transductive_zsl/multimodal_fpl.py:104)
            ratio = n_u / n_s if self.paradigm == "ssl" else n_s / n_u
            self.balance_param = math.sqrt(ratio) if self.modality == "multi" else ratio
            train_data.filepaths, train_data.labels = unseen_imgs + seen_imgs, unseen_labs + seen_labs
        train_data.label_id = True
        return train_data

    def row_weights(self, labels, names):
        """FPL loss as per-row weights (steps.fpl_row_weights); plain strategies use the mean CE."""
        if not getattr(self, "fpl", False) or self.paradigm == "ul":
            return steps.fpl_row_weights([False] * len(labels))
        if self.paradigm == "ssl":     # membership by file name (textual_fpl.py:123-165)
            return steps.fpl_row_weights([n in self.check_unlabeled for n in names], gamma_seen=self.balance_param, gamma_pseudo=1.0)
        unseen_ids = {self.label_to_idx[c] for c in self.unseen_classes}   # membership by label id (transductive_zsl/textual_fpl.py:117-147)
        return steps.fpl_row_weights([int(l) in unseen_ids for l in labels], gamma_seen=1.0, gamma_pseudo=self.balance_param)

    # ------------------------------------------------------------------ loops
    def _loader(self, data, shuffle):
        """Data-parallel like the reference under `accelerate launch` (methods_config/accelerate_config.yml: MULTI_GPU): every rank gets its OWN
        batches of BATCH_SIZE -- batch j of the epoch's order goes to rank j % world_size, the tail is padded with samples from the start
        (dist.rank_batches = accelerate's BatchSamplerShard; `accelerator.prepare(loader)`, e.g. textual_prompt.py:239) -- so N ranks take one
        optimizer step on N x BATCH_SIZE samples with the gradients averaged (DDP behind accelerator.backward, :131).  The shuffle seed is the same
        on every rank (`LOADER_SEED`, default 0: unpinnable upstream)."""
        seed = int(getattr(self.config, "LOADER_SEED", 0))
        sampler = gdist.RankBatchSampler(len(data), int(self.config.BATCH_SIZE), shuffle, seed=seed)
        # (the loader gets its own generator: without one every iter(loader) draws a base seed from the GLOBAL torch RNG, which would shift the stream
        # the prompt re-initialisations of the GRIP iterations read -- ADVICE r5)
        return torch.utils.data.DataLoader(data, batch_sampler=sampler, generator=torch.Generator().manual_seed(seed + 1))

    def _class_space(self, only_seen):
        classes = self.seen_classes if only_seen else self.classes
        idx = torch.tensor([self.label_to_idx[c] for c in classes], device=self.device)
        lut = torch.full((int(idx.max()) + 1,), -1, dtype=torch.long, device=self.device)
        lut[idx] = torch.arange(len(classes), device=self.device)
        return classes, idx, lut

    def _graphed_step(self, classes):
        """The HIP-graph replay of this strategy's prompt step (steps.Graphed*Step) for the current model and class list; rebuilt
        whenever define_model made a new model / optimizer."""
        key = (self._model_gen, tuple(classes))
        if getattr(self, "_gkey", None) != key:
            if self.modality == "text":
                self.model.classes = classes
                g = steps.GraphedCoopFeatureStep(self.model, self.clip_model, self.optimizer)
            elif self.modality == "image":
                g = steps.GraphedVptStep(self.model, self.fixed_text_features(classes), self.scale(), self.optimizer)
            else:
                self.model.classes = classes
                g = steps.GraphedUptStep(self.model, self.scale(), self.optimizer)
            self._gkey, self._gstep = key, g
        return self._gstep

    def _train_epoch(self, train_loader, only_seen=False):
        """One epoch of the strategy's prompt step (the `_train_epoch` bodies of the reference, e.g. textual_prompt.py:63-159).
        Default: every full-sized batch replays the step's forward + backward from a HIP graph (steps.Graphed*Step; a batch of
        another shape -- the ragged last one -- runs the same kernels eagerly), the textual modality encodes the frozen image
        tower `IMAGE_LOOKAHEAD` (51) batches at a time when its features are not cached, and loss / accuracy accumulate on the
        device: one host synchronisation per epoch instead of three per batch.  `GRAPH_STEPS: False` or gradient accumulation
        (ACCUMULATION_ITER > 1) take the eager per-batch path."""
        classes, ids, lut = self._class_space(only_seen)
        if self.modality in ("text", "multi"):
            # features() / predict() / trained_features() re-point model.classes at their own class list between epochs; the eager fallback of a
            # graphed step (ragged last batch) reads it at call time, so it is re-set here every epoch, not only when the graph is (re)built
            self.model.classes = classes
        accum = int(getattr(self.config, "ACCUMULATION_ITER", 1))
        if accum != 1 or not getattr(self.config, "GRAPH_STEPS", True):
            return self._train_epoch_eager(train_loader, classes, ids, lut, accum)
        g = self._graphed_step(classes)
        total = torch.zeros((), dtype=torch.float64, device=self.device)      # the sum the eager loop forms in Python doubles
        correct = torch.zeros((), dtype=torch.int64, device=self.device)
        count = 0

        def batches():
            for img, _, _, label, names in train_loader:
                w = self.row_weights(label.tolist(), names)          # labels are still on the host here: no synchronisation
                yield img.to(self.device, non_blocking=True), label.to(self.device, non_blocking=True), w.to(self.device, non_blocking=True), list(names)

        if self.modality == "text":
            if getattr(self.config, "CACHE_FROZEN_FEATURES", True):
                stream = ((self.frozen_image_features(img, names), label, w) for img, label, w, names in batches())
            else:           # every image is encoded every time it is used, a group of batches per frozen-tower forward
                stream = steps.lookahead_image_features(self.clip_model, ((img, label, w) for img, label, w, _ in batches()),
                                                        int(getattr(self.config, "IMAGE_LOOKAHEAD", 51)))
        else:
            stream = ((img, label, w) for img, label, w, _ in batches())
        for x, label, w in stream:
            total += g(x, lut[label], w)
            correct += (ids[g.logits.argmax(1)] == label).sum()
            count += len(label)
        self.update_scheduler()
        return self._epoch_stats(total, correct, count, len(train_loader))

    def _epoch_stats(self, total, correct, count, n_batches):
        """(mean loss per batch, accuracy) over ALL ranks' batches: the reference gathers predictions and labels of every rank, padded duplicates
        included, for the accuracy (textual_prompt.py:146-150); its loss stays per rank -- here it is the mean over the ranks."""
        t = torch.tensor([float(total), float(correct), float(count), float(n_batches)], dtype=torch.float64, device=self.device)
        gdist.allreduce_sum_(t)
        total, correct, count, n_batches = t.tolist()
        return total / max(n_batches, 1.0), correct / max(count, 1.0)

    def _train_epoch_eager(self, train_loader, classes, ids, lut, accum):
        total, correct, count = 0.0, 0, 0
        for i, (img, _, _, label, names) in enumerate(train_loader):
            img, label = img.to(self.device), label.to(self.device)
            image_features, text_features = self.features(img, classes, list(names))
            logits = steps.CosineHeadFn.apply(image_features, text_features, self.scale())
            w = self.row_weights(label.tolist(), names).to(self.device)
            loss = steps.WeightedCEFn.apply(logits, lut[label], w) / accum
            loss.backward()
            total += float(loss.detach()) * accum
            if (i + 1) % accum == 0 or i + 1 == len(train_loader):
                gdist.allreduce_mean_([p.grad for p in self.model.parameters() if p.grad is not None])
                self.backpropagate()
            correct += int((ids[logits.argmax(1)] == label).sum())
            count += len(label)
        self.update_scheduler()
        return self._epoch_stats(total, correct, count, len(train_loader))

    @torch.no_grad()
    def predict(self, data, classes):
        """Global label ids predicted for every item of `data`, logits [N, len(classes)] on the CPU, in dataset order on every rank.  Sharded like the
        reference's evaluation loops under accelerate (`accelerator.prepare(test_loader)`, textual_prompt.py:239): each rank runs its own batches
        (dist.rank_batches over the dataset order), the per-rank logits are all-gathered and the samples accelerate's even sharding duplicated
        are dropped again (:285-294)."""
        ids = torch.tensor([self.label_to_idx[c] for c in classes], device=self.device)
        outs = []
        for batch in self._loader(data, False):
            img = batch[0].to(self.device)
            image_features, text_features = self.features(img, classes)
            logits, _, am, _ = cosine_head(image_features, text_features, self.scale(), want_probs=False)
            outs.append(logits)
        logits = torch.cat(outs) if outs else torch.empty(0, len(classes), device=self.device)
        logits = gdist.gather_in_dataset_order(logits, len(data), int(self.config.BATCH_SIZE))
        return ids[logits.argmax(1)].cpu(), logits.cpu()

    def _run_validation(self, val_data, only_seen=False):
        classes = self.seen_classes if only_seen and self.val_unseen_files is None else self.classes
        pred, _ = self.predict(val_data, classes)
        labels = torch.tensor([l if val_data.label_id else self.label_to_idx[l] for l in val_data.labels])
        return float((pred == labels).float().mean())

    def train(self, train_data, val_data, unlabeled_data=None, only_seen=False, iter_train=False):
        train_data = self.create_training_dataset(train_data, unlabeled_data) if not iter_train else train_data
        self.define_model(self.seen_classes if only_seen else self.classes)
        loader = self._loader(train_data, True)
        best_acc, best_prompt = -1.0, None
        for epoch in range(int(self.config.EPOCHS)):
            loss, acc = self._train_epoch(loader, only_seen=only_seen)
            val_acc = self._run_validation(val_data, only_seen=only_seen) if val_data is not None and len(val_data) else acc
            log.info(f"epoch {epoch}: loss {loss:.4f} train acc {acc:.3f} val acc {val_acc:.3f}")
            if val_acc > best_acc:
                best_acc, best_prompt = val_acc, self.prompt_snapshot()
        return best_acc, best_prompt

    # ------------------------------------------------------------------ evaluation
    def test_predictions(self, data, standard_zsl=False):
        import pandas as pd
        classes = self.unseen_classes if standard_zsl else self.classes
        pred, _ = self.predict(data, classes)
        idx_to_class = {v: k for k, v in self.label_to_idx.items()}
        df = pd.DataFrame({"id": [f.split("/")[-1] for f in data.filepaths], "class": [idx_to_class[int(p)] for p in pred]})
        df.drop_duplicates(subset=["id", "class"], inplace=True)
        return df

    def evaluation(self, data):
        pred, logits = self.predict(data, self.classes)
        idx_to_class = {v: k for k, v in self.label_to_idx.items()}
        return [f.split("/")[-1] for f in data.filepaths], [idx_to_class[int(p)] for p in pred], logits

    # ------------------------------------------------------------------ pseudolabels from the trained model
    @torch.no_grad()
    def trained_text_features(self, classes, clip_model):
        """Text features [C, E] of the CURRENT prompts through `clip_model`'s text tower (the f16 model or its exact twin),
        and the visual prompt (or None) the image side has to be encoded with."""
        from ..engine import text_prefix_forward
        if self.modality == "text":
            ids = self.text_encoder._token_ids(self.model.prefix.shape[1], classes)
            return text_prefix_forward(clip_model.text_tower, ids, self.model.prefix.detach()), None
        if self.modality == "image":
            return clip_model.encode_text(clip.tokenize(self.text_prompts(classes)).to(self.device)), self.model.prefix.detach()
        coop_embs, vpt_embs = self.model.mix()
        ids = self.text_encoder._token_ids(coop_embs.shape[1], classes)
        return text_prefix_forward(clip_model.text_tower, ids, coop_embs.detach()), vpt_embs.detach()

    @torch.no_grad()
    def trained_features(self, images, classes, chunk=440):
        """(image features [N, E] of the whole ordered pool, text features [C, E]) of the CURRENT model for the pseudolabel pass.
        The image side goes through pseudolabels.encode_pool: chunked, sharded contiguously over the ranks of one node, one
        all-gather of the embeddings -- with the trained visual prompt where the modality has one.  The text
        side is computed once per call on every rank (C x 6 GF, cheaper than a broadcast); the UPT mixer runs once.  The
        reference does all of this per image at batch 1 (textual_fpl.py:203-205 + :225; visual_fpl.py:250 + :264;
        multimodal_fpl.py:223 runs BOTH towers and the mixer for every image)."""
        tower = self.clip_model.visual.tower
        if self.modality == "text":
            self.model.classes = classes
            return pl.encode_pool(tower, images, chunk=chunk), self.model(classes)
        if self.modality == "image":
            return pl.encode_pool(tower, images, chunk=chunk, prefix=self.model.prefix.detach()), self.fixed_text_features(classes)
        self.model.classes = classes
        coop_embs, vpt_embs = self.model.mix()
        txt = self.model.text_encoder(coop_embs, classes)
        return pl.encode_pool(tower, images, chunk=chunk, prefix=vpt_embs.detach()), txt

    @torch.no_grad()
    def assign_pseudo_labels(self, k, unlabeled_data):
        """The nine `assign_pseudo_labels` of the reference (e.g. transductive_zsl/multimodal_fpl.py:194-285):
        leaderboard over the TRAINED model's logits on the unseen classes (arg-max on the logits, :222)."""
        from ..utils.clip_pseudolabels import _pool_images
        classes = self.unseen_classes if self.paradigm == "trzsl" else self.classes
        images = _pool_images(unlabeled_data, self.transform, self.device)
        labels = [self.label_to_idx[c] for c in classes]
        if pl.mode() == "identical" and not self.clip_model.exact:
            # the lists the reference's fp32 pass would produce with these prompts: f16 screen + exact refinement
            twin = self.clip_model.exact_twin()
            txt, vprompt = self.trained_text_features(classes, twin)
            fp, lab = pl.identical_lists(self.clip_model.visual.tower, twin.visual.tower, images, txt, self.scale(),
                                         list(unlabeled_data.filepaths), labels, k, chunk=440, prefix=vprompt, argmax_on="logits",
                                         visual_mid=pl.mid_tower(self.clip_model, len(unlabeled_data.filepaths)))
        else:
            img, txt = self.trained_features(images, classes)
            fp, lab = pl.pseudolabel_from_features(img, txt, self.scale(), list(unlabeled_data.filepaths), labels, k, argmax_on="logits")
        unlabeled_data.filepaths, unlabeled_data.labels, unlabeled_data.label_id = fp, lab, True
        return unlabeled_data

    def get_pseudo_labels(self, unlabeled_examples):
        return self.assign_pseudo_labels(int(self.config.N_PSEUDOSHOTS), unlabeled_examples)

    # ------------------------------------------------------------------ iterative schedules
    def _n_pseudoshots(self, niter, num_samples, n_unlabeled, n_classes):
        n = int(niter * num_samples / n_classes)            # pseudo_iterative.py:62-75, 113-125
        return n if n * n_classes <= n_unlabeled else math.floor(n_unlabeled / n_classes)

    def grip_train(self, train_data, val_data, unlabeled_data, only_seen=False):
        """GRIP: grow the number of pseudolabels per class by one STEP_QUANTILE of the pool per iteration,
        re-initialising the prompts each time; iteration 1 labels with frozen CLIP, later ones with the
        previously trained prompts."""
        c = self.config
        num_iter = int(100 / int(c.STEP_QUANTILE))
        n_cls = len(self.unseen_classes if self.paradigm == "trzsl" else self.classes)
        num_samples = int(len(unlabeled_data) / num_iter)
        original_train, original_unlabeled = copy.copy(train_data.filepaths), copy.copy(unlabeled_data.filepaths)
        original_labels, original_label_id = copy.copy(train_data.labels), train_data.label_id
        out = None
        for niter in range(1, num_iter + 1):
            c.N_PSEUDOSHOTS = max(1, self._n_pseudoshots(niter, num_samples, len(original_unlabeled), n_cls))
            train_data.filepaths, train_data.labels, train_data.label_id = list(original_train), list(original_labels), original_label_id
            unlabeled_data.filepaths, unlabeled_data.labels = list(original_unlabeled), None
            if niter == 1:
                data = self.create_training_dataset(train_data, unlabeled_data)
            else:
                data = self.merge_pseudolabels(train_data, self.get_pseudo_labels(unlabeled_data))
            out = self.train(data, val_data, only_seen=only_seen, iter_train=True)
            from ..utils import save_parameters, save_pseudo_labels
            save_pseudo_labels(list(data.filepaths), list(data.labels), c, niter)     # compute_metrics.py:150-154
            save_parameters(out[1], c, iteration=niter)
            log.info(f"GRIP iteration {niter}/{num_iter}: {c.N_PSEUDOSHOTS} pseudo-shots per class, val acc {out[0]:.3f}")
        return out

    def fixed_iterative_train(self, train_data, val_data, unlabeled_data, only_seen=False):
        """Iterative FPL: same loop with a FIXED number of pseudo-shots per class."""
        c = self.config
        k = int(c.N_PSEUDOSHOTS)
        original_train, original_unlabeled = copy.copy(train_data.filepaths), copy.copy(unlabeled_data.filepaths)
        original_labels, original_label_id = copy.copy(train_data.labels), train_data.label_id
        out = None
        for niter in range(1, int(100 / int(c.STEP_QUANTILE)) + 1):
            c.N_PSEUDOSHOTS = k
            train_data.filepaths, train_data.labels, train_data.label_id = list(original_train), list(original_labels), original_label_id
            unlabeled_data.filepaths, unlabeled_data.labels = list(original_unlabeled), None
            data = self.create_training_dataset(train_data, unlabeled_data) if niter == 1 else \
                self.merge_pseudolabels(train_data, self.get_pseudo_labels(unlabeled_data))
            out = self.train(data, val_data, only_seen=only_seen, iter_train=True)
        return out
