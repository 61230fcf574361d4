"""CPU: the C-ABI library loads and exports every symbol include/grip_amd.h declares (no GPU
compute is called), the host-only layout entry points behave, and the product refuses to run
without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, "include", "grip_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import grip_amd  # noqa: F401
    from grip_amd import native
    lib = native.lib()
    names = _declared()
    assert len(names) >= 15
    assert set(names) == set(native.EXPORTS), set(names) ^ set(native.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.grip_abi_version() == native.ABI_VERSION


def test_library_exports_the_declared_test_hooks():
    """include/grip_amd_debug.h (kernel test hooks + GEMM profiler, outside the drop-in ABI) matches the library too."""
    import grip_amd  # noqa: F401
    from grip_amd import native
    text = open(os.path.join(REPO, "include", "grip_amd_debug.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(grip_[a-z_]+)\s*\(", text)))
    assert names == sorted(native._DEBUG_SIGS), names
    lib = native.lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_layout_is_complete_and_non_overlapping():
    import numpy as np

    import grip_amd  # noqa: F401
    from grip_amd import config, native, weights
    lib = native.lib()
    for name in ("tiny", "ViT-B/16", "ViT-L/14@336px"):
        d = config.get_dims(name)
        for kind, dims in ((0, native.Dims(0, d.vision_width, d.vision_layers, d.vision_heads, d.embed_dim, d.vision_seq,
                                           d.vision_patch_size, d.image_resolution, 0, 16, 0)),
                           (1, native.Dims(1, d.transformer_width, d.transformer_layers, d.transformer_heads, d.embed_dim,
                                           d.context_length, 0, 0, d.vocab_size, 16, 0))):
            n16, n32 = ctypes.c_int64(), ctypes.c_int64()
            native.check(lib.grip_layout_size(ctypes.byref(dims), ctypes.byref(n16), ctypes.byref(n32)))
            slots, s, i = [], native.Slot(), 0
            while lib.grip_layout_slot(ctypes.byref(dims), i, ctypes.byref(s)) == 0:
                slots.append((s.name.decode(), s.dtype, s.derived, s.offset, s.rows, s.cols, s.ld))
                i += 1
            for dt, total in ((0, n16.value), (1, n32.value)):
                spans = sorted((o, o + r * ld) for _, t, _, o, r, _, ld in slots if t == dt)
                assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "overlapping slots"
                assert spans[-1][1] <= total
            # every OpenAI state_dict key of this tower has exactly one primary slot of the right shape
            prefix = "visual." if kind == 0 else ""
            primary = {n: (r, c) for n, _, der, _, r, c, _ in slots if not der}
            for key, shape, _ in weights.weight_spec(d):
                if key == "logit_scale" or key.startswith("visual.") != (kind == 0):
                    continue
                rel = key[len(prefix):]
                assert rel in primary, rel
                r, c = primary[rel]
                assert r * c == int(np.prod(shape)), (rel, shape, r, c)


def test_bad_dims_are_rejected_with_a_message():
    import grip_amd  # noqa: F401
    from grip_amd import native
    lib = native.lib()
    bad = native.Dims(0, 100, 2, 2, 128, 17, 8, 32, 0, 4, 0)     # width not a multiple of 128
    a, b = ctypes.c_int64(), ctypes.c_int64()
    assert lib.grip_layout_size(ctypes.byref(bad), ctypes.byref(a), ctypes.byref(b)) == 1
    assert b"width" in lib.grip_last_error()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import grip_amd  # noqa: F401
    from grip_amd import clip, native
    with pytest.raises(native.GripError):
        clip.load("tiny", device="cuda")
    with pytest.raises(native.GripError):
        clip.load("tiny", device="cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "menghini-neurips23-code_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(root, f)
                assert "/root/reference" not in src, os.path.join(root, f)
