"""ctypes binding of the C ABI in include/grip_amd.h (libgrip_amd.so, built in-tree by
`make -C menghini-neurips23-code_amd/csrc` / `__graft_entry__.build()`).

There is no CPU fallback: if the library is missing or reports another ABI version every entry
point raises.  Nothing here imports the oracle.
"""
import ctypes
import os
from ctypes import POINTER, c_char, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRIP_LIB") or os.path.join(_HERE, "libgrip_amd.so")      # GRIP_LIB: another build of the same ABI (developer A/B)
HOST_LIB_PATH = os.environ.get("GRIP_HOST_LIB")       # developer: a sanitizer build of the host-only sources (`make -C csrc sanitize`) whose grip_leaderboard_* / grip_bpe_* replace the library's
ABI_VERSION = 8
FWD_TRAIN, FWD_SHARED_PREFIX, FWD_NO_POS_EMB, FWD_STREAM_HILO = 1, 2, 4, 8      # grip_vit_forward / grip_text_forward flags


class GripError(RuntimeError):
    pass


class Dims(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("kind", "width", "layers", "heads", "embed_dim", "seq0", "patch",
                                       "resolution", "vocab", "max_prefix", "precision")]


class Slot(ctypes.Structure):
    _fields_ = [("name", c_char * 96), ("dtype", c_int32), ("derived", c_int32), ("offset", c_int64),
                ("rows", c_int64), ("cols", c_int64), ("ld", c_int64)]


class PreprocessItem(ctypes.Structure):
    """grip_preprocess_item (include/grip_amd.h); 88 bytes, natural alignment."""
    _fields_ = [("img", c_void_p), ("H", c_int32), ("W", c_int32), ("hcoef", c_void_p), ("hbounds", c_void_p), ("hksize", c_int32), ("W_out", c_int32),
                ("vcoef", c_void_p), ("vbounds", c_void_p), ("vksize", c_int32), ("H_out", c_int32), ("crop_left", c_int32), ("crop_top", c_int32),
                ("tmp", c_void_p), ("out", c_void_p)]


MIXER_TENSORS = ("coop", "vpt", "coop_pre_w", "coop_pre_b", "vpt_pre_w", "vpt_pre_b", "ln1_g", "ln1_b", "in_w", "in_b", "out_w", "out_b",
                 "ln2_g", "ln2_b", "fc_w", "fc_b", "proj_w", "proj_b", "coop_post_w", "coop_post_b", "vpt_post_w", "vpt_post_b")


class UptMixer(ctypes.Structure):
    """grip_upt_mixer (include/grip_amd.h)."""
    _fields_ = [(n, c_int32) for n in ("n_prompt", "text_width", "vision_width", "dim", "half_linears", "reserved_")] + [(n, c_void_p) for n in MIXER_TENSORS]


_SIGS = {
    "grip_last_error": (ctypes.c_char_p, []),
    "grip_abi_version": (c_int, []),
    "grip_layout_slot": (c_int, [POINTER(Dims), c_int, POINTER(Slot)]),
    "grip_layout_size": (c_int, [POINTER(Dims), POINTER(c_int64), POINTER(c_int64)]),
    "grip_tower_create": (c_int, [POINTER(Dims), c_void_p, c_void_p, POINTER(c_void_p)]),
    "grip_tower_finalize": (c_int, [c_void_p, c_void_p]),
    "grip_tower_destroy": (c_int, [c_void_p]),
    "grip_workspace_bytes": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "grip_vit_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_uint64), c_void_p]),
    "grip_vit_backward_prefix": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_uint64, c_void_p]),
    "grip_text_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_uint64), c_void_p]),
    "grip_text_backward_prefix": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_uint64, c_void_p]),
    "grip_cosine_head": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "grip_cosine_head_backward": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "grip_weighted_ce": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "grip_upt_mixer_workspace": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "grip_upt_mixer_forward": (c_int, [POINTER(UptMixer), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "grip_upt_mixer_backward": (c_int, [POINTER(UptMixer), c_void_p, c_void_p, POINTER(UptMixer), c_void_p, c_size_t, c_void_p]),
    "grip_preprocess_image": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "grip_preprocess_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "grip_leaderboard_scan": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, POINTER(c_int64)]),
    "grip_leaderboard_scan_bounded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_int, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                              POINTER(c_int64), c_void_p, POINTER(c_int64)]),
    "grip_bpe_create": (c_int, [ctypes.c_char_p, c_size_t, POINTER(c_void_p)]),
    "grip_bpe_destroy": (c_int, [c_void_p]),
    "grip_bpe_special_ids": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "grip_bpe_encode_word": (c_int, [c_void_p, ctypes.c_char_p, c_int, c_void_p, c_int, POINTER(c_int)]),
    "grip_bpe_encode_ascii": (c_int, [c_void_p, ctypes.c_char_p, c_int, c_void_p, c_int, POINTER(c_int)]),
    "grip_set_cu_budget": (c_int, [c_int]),
    "grip_comm_unique_id": (c_int, [c_void_p]),
    "grip_comm_init_rank": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "grip_comm_destroy": (c_int, [c_void_p]),
    "grip_allgather_embeddings": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "grip_allreduce_mean": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
}
EXPORTS = tuple(_SIGS)   # the drop-in ABI (include/grip_amd.h)
_DEBUG_SIGS = {          # kernel-level test hooks (csrc/tower.hip), not part of the ABI
    "grip_debug_gemm": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p]),
    "grip_debug_attention": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "grip_debug_gemm_ln": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_void_p]),
    "grip_debug_gemm_train": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "grip_debug_coop_split": (c_int, [c_int, c_int, c_int]),
    "grip_debug_gemm_splitk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "grip_debug_ln_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                   c_int, c_int, c_void_p]),
    "grip_debug_attention_exact": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "grip_profile_enable": (c_int, [c_int]),
    "grip_profile_collect": (c_int, [c_int, c_void_p, c_void_p, c_void_p]),
    "grip_debug_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "grip_debug_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "grip_debug_gemm_split": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "grip_debug_split_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "grip_debug_split_last_wlo": (c_int, []),
    "grip_debug_attention_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


def lib():
    """The loaded library; raises GripError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GripError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C menghini-neurips23-code_amd/csrc`). "
                "There is no CPU fallback.")
        import torch  # noqa: F401  -- load torch's bundled HIP runtime first so both share one libamdhip64
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in {**_SIGS, **_DEBUG_SIGS}.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if l.grip_abi_version() != ABI_VERSION:
            raise GripError(f"libgrip_amd.so ABI {l.grip_abi_version()} != host layer ABI {ABI_VERSION}; rebuild")
        if HOST_LIB_PATH:
            h = ctypes.CDLL(HOST_LIB_PATH)
            for name, (res, args) in _SIGS.items():
                if name.startswith(("grip_leaderboard_", "grip_bpe_")):
                    fn = getattr(h, name)
                    fn.restype = res
                    fn.argtypes = args
                    setattr(l, name, fn)
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise GripError(f"grip_amd native call failed (status {rc}): {lib().grip_last_error().decode()}")
