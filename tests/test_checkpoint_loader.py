"""CPU: the OpenAI-checkpoint side of clip.load.  The published checkpoints are TorchScript archives whose
state_dict carries the OpenAI key names, f16 tensors and three scalar book-keeping entries; openai/CLIP's build_model infers
every dimension from tensor shapes.  No checkpoint exists offline, so the test builds archives of the same FORM: a scripted
module tree with the real key names (tiny dimensions for the archive round trip, true ViT-B/32 / B/16 / L/14 / L/14@336px
shapes -- unallocated storage -- for the key-set and dimension inference)."""
import os

import pytest
import torch
from torch import nn


def _module_tree(sd):
    """nn.Module whose state_dict() is exactly `sd` (nested sub-modules named after the dotted keys)."""
    root = nn.Module()
    for key, t in sd.items():
        node = root
        parts = key.split(".")
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, nn.Module())
            node = getattr(node, p)
        node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
    return root


class _Scriptable(nn.Module):
    """A scripted module needs a forward; the archives OpenAI publishes are whole scripted CLIP models."""

    def __init__(self, tree, extras):
        super().__init__()
        for n, m in tree.named_children():
            self.add_module(n, m)
        for n, p in tree.named_parameters(recurse=False):
            self.register_parameter(n, p)
        for k, v in extras.items():
            self.register_buffer(k, torch.tensor(v))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x


def test_torchscript_archive_round_trip(tmp_path):
    import grip_amd  # noqa: F401
    from grip_amd import config, weights
    d = config.get_dims("tiny")
    sd = {k: torch.from_numpy(v).half() if v.ndim >= 2 else torch.from_numpy(v) for k, v in weights.init_state_dict(d, 3).items()}
    path = str(tmp_path / "ViT-tiny.pt")
    m = _Scriptable(_module_tree(sd), {"input_resolution": d.image_resolution, "context_length": d.context_length, "vocab_size": d.vocab_size})
    torch.jit.save(torch.jit.script(m), path)
    got = weights.read_checkpoint(path)                     # torch.jit.load(...).state_dict() minus the book-keeping entries
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert got["visual.conv1.weight"].dtype == torch.float16
    assert weights.dims_from_state_dict(got, "tiny") == d
    weights.check_state_dict(got, d)
    # the same weights as a pickled state_dict and as a {"state_dict": ...} wrapper
    torch.save(sd, str(tmp_path / "plain.pt"))
    torch.save({"state_dict": sd, "epoch": 3}, str(tmp_path / "wrapped.pt"))
    for f in ("plain.pt", "wrapped.pt"):
        again = weights.read_checkpoint(str(tmp_path / f))
        assert set(again) == set(sd) and torch.equal(again["text_projection"], sd["text_projection"])


@pytest.mark.parametrize("name", ["ViT-B/32", "ViT-B/16", "ViT-L/14", "ViT-L/14@336px"])
def test_openai_key_sets_and_dimension_inference(name):
    """The key set and shapes of each encoder the reference names (VIS_ENCODER) against the published architecture numbers,
    and build_model-style inference of every dimension from the shapes alone."""
    import grip_amd  # noqa: F401
    from grip_amd import config, weights
    d = config.get_dims(name)
    sd = {k: torch.empty(s, device="meta") for k, s, _ in weights.weight_spec(d)}
    assert weights.dims_from_state_dict(sd, name) == d
    weights.check_state_dict(sd, d)
    n_params = sum(int(torch.tensor(v.shape).prod()) if v.ndim else 1 for v in sd.values())
    want = {"ViT-B/32": 151_277_313, "ViT-B/16": 149_620_737, "ViT-L/14": 427_616_513, "ViT-L/14@336px": 427_944_193}[name]
    assert n_params == want, n_params            # parameter counts of the published models
    vis = [k for k in sd if k.startswith("visual.")]
    assert len(vis) == 5 + 12 * d.vision_layers + 3 and len(sd) - len(vis) == 2 + 12 * d.transformer_layers + 4
    # a wrong-shaped or missing tensor is refused with a message that names it
    bad = dict(sd)
    bad["visual.proj"] = torch.empty(d.vision_width, d.embed_dim + 1, device="meta")
    with pytest.raises(RuntimeError, match="visual.proj"):
        weights.check_state_dict(bad, d)
    del bad["visual.proj"]
    with pytest.raises(RuntimeError, match="not a ViT CLIP"):
        weights.dims_from_state_dict(bad)
    short = {k: v for k, v in sd.items() if "resblocks.0.ln_1" not in k}
    with pytest.raises(RuntimeError, match="missing"):
        weights.check_state_dict(short, d)
