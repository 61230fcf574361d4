"""ORACLE (test infrastructure) -- CPU fp32 restatement of the reference's
prompt wrappers, cosine head and FPL losses.  Functional form: every function
takes the oracle CLIP (oracle/clip/model.py) and explicit tensors, and cites
the reference lines it follows.  oracle/gen_golden.py checks each of them against
the reference's own classes (imported unmodified) before fixtures are written.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- a1 / a2
def vision_forward(visual, x, image_prefix=None, pos_emb=True):
    """CustomVisionTransformer.forward, models/clip_encoders.py:123-194 (deep_embs branch is dead code).

    x [B,3,R,R]; image_prefix [P,d] or [1,P,d] or None.  The prefix is inserted between CLS and the
    patches AFTER the positional embedding was added (so it carries none), :146-155."""
    x = visual.conv1(x)                                      # :131
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)   # :132-133
    cls = visual.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1)                           # :135-144
    if pos_emb:
        x = x + visual.positional_embedding.to(x.dtype)      # :146
    if image_prefix is not None:
        if image_prefix.dim() == 2:
            image_prefix = image_prefix[None]
        image_prefix = image_prefix.expand(x.shape[0], -1, -1)   # :148
        x = torch.cat([x[:, :1, :], image_prefix, x[:, 1:, :]], dim=1)   # :150-155
    x = visual.ln_pre(x)                                     # :163
    x = visual.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)  # :165,186-187
    x = visual.ln_post(x[:, 0, :])                           # :189
    if visual.proj is not None:
        x = x @ visual.proj                                  # :191-192
    return x


# ---------------------------------------------------------------- a4
def coop_prompt_strings(n_prefix, classes):
    """models/clip_encoders.py:54-57: 'X X ... X <class>' with n_prefix X's."""
    return [" ".join([" ".join(["X"] * n_prefix).strip(), c]) for c in classes]


def text_forward(clip_model, token_ids, class_embeddings=None, enable_pos_emb=True):
    """CustomTextEncoder.forward, models/clip_encoders.py:43-90, given the token ids of :60.

    class_embeddings [1,P,dt] (broadcast over classes) or [C,P,dt]; positions 1..P of the token
    embedding are overwritten (:67).  On CPU the reference runs the transformer in fp32 (:82-83)."""
    x = clip_model.token_embedding(token_ids.long())         # :63
    if class_embeddings is not None:
        P = class_embeddings[0].size(0)
        x = x.clone()
        x[:, 1:P + 1, :] = class_embeddings                  # :67
    if enable_pos_emb:
        x = x + clip_model.positional_embedding              # :70-74
    x = clip_model.transformer(x.permute(1, 0, 2).float()).permute(1, 0, 2)   # :75-84
    x = clip_model.ln_final(x)                               # :85
    return x[torch.arange(x.shape[0]), token_ids.argmax(dim=-1)] @ clip_model.text_projection   # :86-89


# ---------------------------------------------------------------- a8
def upt_mixer(p, coop_embeddings, vpt_embeddings, dtype=torch.float32):
    """UPTModel.forward mixer part, models/prompts_models.py:129-146.

    p: dict of the mixer's parameters (proj_{coop,vpt}_{pre,post}.{weight,bias} and the 1-layer,
    1-head `transformer.resblocks.0.*`).  cat(dim=0) makes the pair look like an LND tensor with
    sequence length 2 and batch = prompt length (:135); the fp32 -> fp16 -> dtype round trip of
    :138-141 is reproduced."""
    from .clip.model import Transformer
    coop = F.linear(coop_embeddings, p["proj_coop_pre.weight"], p["proj_coop_pre.bias"])    # :131
    vpt = F.linear(vpt_embeddings, p["proj_vpt_pre.weight"], p["proj_vpt_pre.bias"])        # :135
    seq = torch.cat((coop, vpt), dim=0).to(torch.float32)                                   # :138
    tdim = seq.shape[-1]
    tr = Transformer(width=tdim, layers=1, heads=1)
    tr.load_state_dict({k[len("transformer."):]: v for k, v in p.items() if k.startswith("transformer.")})
    out = tr(seq).to(torch.float16)                                                         # :141
    n = len(coop_embeddings)
    coop_len, coop_dim = coop_embeddings.shape[1:]
    vpt_len, vpt_dim = vpt_embeddings.shape[1:]
    coop_embs = F.linear(out[:n].to(dtype), p["proj_coop_post.weight"], p["proj_coop_post.bias"]).reshape(-1, coop_len, coop_dim)
    vpt_embs = F.linear(out[n:].to(dtype), p["proj_vpt_post.weight"], p["proj_vpt_post.bias"]).reshape(-1, vpt_len, vpt_dim)
    return coop_embs, vpt_embs                                                              # :144-145


# ---------------------------------------------------------------- a9
def cosine_head(image_features, text_features, logit_scale_log):
    """methods/semi_supervised_learning/textual_prompt.py:98-109: normalise both, logits =
    logit_scale.exp() * img @ txt.T, argmax over classes."""
    t = text_features / text_features.norm(dim=-1, keepdim=True)
    i = image_features / image_features.norm(dim=-1, keepdim=True)
    logits = torch.as_tensor(logit_scale_log).exp() * i @ t.t()
    return logits, torch.argmax(logits, dim=1)


def zero_shot_prompt_strings(template, classnames):
    """utils/clip_pseudolabels.py:24 -- f-string concatenation, NOT .format: 'a photo of a {}forest'."""
    return [f"{template}{' '.join(i.split('_'))}" for i in classnames]


def format_prompt_strings(template, classnames):
    """every other site, e.g. methods/clip_baseline.py:57-59: template.format(name)."""
    return [template.format(" ".join(i.split("_"))) for i in classnames]


# ---------------------------------------------------------------- a10
def _ce(logits, labels, rows):
    if not rows:
        return 0
    return F.cross_entropy(logits[rows], labels[rows])


def fpl_loss_ssl(logits, labels, is_unlabeled, balance_param):
    """methods/semi_supervised_learning/textual_fpl.py:123-165: gamma*CE(labeled rows) + CE(pseudolabeled rows);
    membership is by path in check_unlabeled, passed here as a bool list."""
    seen = [i for i, u in enumerate(is_unlabeled) if not u]
    unseen = [i for i, u in enumerate(is_unlabeled) if u]
    return balance_param * _ce(logits, labels, seen) + _ce(logits, labels, unseen)


def fpl_loss_trzsl(logits, labels, seen_ids, unseen_ids, balance_param):
    """methods/transductive_zsl/textual_fpl.py:117-147: CE(rows with seen label) + gamma*CE(rows with unseen label)."""
    seen = [i for i, l in enumerate(labels.tolist()) if l in seen_ids]
    unseen = [i for i, l in enumerate(labels.tolist()) if l in unseen_ids]
    return _ce(logits, labels, seen) + balance_param * _ce(logits, labels, unseen)


def fpl_loss_ul(logits, labels):
    """methods/unsupervised_learning/visual_fpl.py:107-122: plain mean CE."""
    return F.cross_entropy(logits, labels)


def balance_ssl(n_unseen, n_seen):
    return n_unseen / n_seen                     # semi_supervised_learning/textual_fpl.py:115


def balance_ssl_multimodal(n_unseen, n_seen):
    return math.sqrt(n_unseen / n_seen)          # semi_supervised_learning/multimodal_fpl.py:107


def balance_trzsl(n_unseen, n_seen):
    return n_seen / n_unseen                     # transductive_zsl/textual_fpl.py:109
