"""CPU: cross-check the oracle's restatement of the openai-CLIP arithmetic against an independent
implementation of the same architecture (transformers.CLIPModel, "Oracle-B" of SURVEY.md 4) on
shared random weights.  The reference's `clip` dependency is absent, so this is the strongest pin
available for the un-prompted towers."""
import pytest
import torch

from conftest import oracle_clip

transformers = pytest.importorskip("transformers")


def _to_hf(sd, d):
    out = {}

    def blocks(src, dst, n, w):
        for i in range(n):
            s, t = f"{src}.resblocks.{i}", f"{dst}.encoder.layers.{i}"
            wq, wk, wv = sd[f"{s}.attn.in_proj_weight"].split(w, 0)
            bq, bk, bv = sd[f"{s}.attn.in_proj_bias"].split(w, 0)
            for nm, ww, bb in (("q_proj", wq, bq), ("k_proj", wk, bk), ("v_proj", wv, bv)):
                out[f"{t}.self_attn.{nm}.weight"], out[f"{t}.self_attn.{nm}.bias"] = ww, bb
            out[f"{t}.self_attn.out_proj.weight"] = sd[f"{s}.attn.out_proj.weight"]
            out[f"{t}.self_attn.out_proj.bias"] = sd[f"{s}.attn.out_proj.bias"]
            for a, b in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
                out[f"{t}.{b}.weight"], out[f"{t}.{b}.bias"] = sd[f"{s}.{a}.weight"], sd[f"{s}.{a}.bias"]

    blocks("visual.transformer", "vision_model", d.vision_layers, d.vision_width)
    blocks("transformer", "text_model", d.transformer_layers, d.transformer_width)
    out["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    out["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    out["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    for a, b in (("visual.ln_pre", "vision_model.pre_layrnorm"), ("visual.ln_post", "vision_model.post_layernorm"),
                 ("ln_final", "text_model.final_layer_norm")):
        out[f"{b}.weight"], out[f"{b}.bias"] = sd[f"{a}.weight"], sd[f"{a}.bias"]
    out["visual_projection.weight"] = sd["visual.proj"].t()
    out["text_projection.weight"] = sd["text_projection"].t()
    out["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out["logit_scale"] = sd["logit_scale"]
    return out


def test_oracle_clip_matches_transformers_clip():
    import grip_amd  # noqa: F401
    from grip_amd import config
    d = config.get_dims("tiny")
    om = oracle_clip().load("tiny")[0]
    cfg = transformers.CLIPConfig(
        text_config=dict(vocab_size=d.vocab_size, hidden_size=d.transformer_width, intermediate_size=4 * d.transformer_width,
                         num_hidden_layers=d.transformer_layers, num_attention_heads=d.transformer_heads,
                         max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=49407, bos_token_id=49406, pad_token_id=0,
                         projection_dim=d.embed_dim),
        vision_config=dict(hidden_size=d.vision_width, intermediate_size=4 * d.vision_width, num_hidden_layers=d.vision_layers,
                           num_attention_heads=d.vision_heads, image_size=d.image_resolution, patch_size=d.vision_patch_size,
                           hidden_act="quick_gelu", projection_dim=d.embed_dim),
        projection_dim=d.embed_dim)
    hf = transformers.CLIPModel(cfg).eval()
    sd = {k: v.detach() for k, v in om.state_dict().items() if "attn_mask" not in k}
    missing, unexpected = hf.load_state_dict(_to_hf(sd, d), strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "position_ids" not in k], missing
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 3, d.image_resolution, d.image_resolution, generator=g)
    tok = oracle_clip().tokenize(["a photo of a forest", "X X X river bank", "sea"])
    with torch.no_grad():
        want_i = hf.get_image_features(pixel_values=x)
        want_t = hf.get_text_features(input_ids=tok.long(), attention_mask=(tok != 0).long())
        if not torch.is_tensor(want_i):
            want_i, want_t = want_i.pooler_output, want_t.pooler_output
        got_i, got_t = om.encode_image(x), om.encode_text(tok)
    torch.testing.assert_close(got_i, want_i, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got_t, want_t, rtol=1e-4, atol=1e-4)
