"""ORACLE (test infrastructure) -- ctypes binding of oracle/leaderboard_ref.c."""
import ctypes
import os
import subprocess

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
_so = os.path.join(_here, "_build", "libleaderboard_ref.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _here])
    return _so


def leaderboard_ref(probs, pred_ids, paths, class_labels, k):
    if not os.path.exists(_so):
        build()
    lib = ctypes.CDLL(_so)
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    pred = np.ascontiguousarray(pred_ids, dtype=np.int32)
    N, C = probs.shape
    arr = (ctypes.c_char_p * N)(*[p.encode() for p in paths])
    cap = C * min(k, N)
    out_img = np.empty(cap, dtype=np.int32)
    out_cls = np.empty(cap, dtype=np.int32)
    lib.leaderboard_ref.restype = ctypes.c_int
    m = lib.leaderboard_ref(probs.ctypes.data_as(ctypes.c_void_p), pred.ctypes.data_as(ctypes.c_void_p), arr,
                            ctypes.c_int(N), ctypes.c_int(C), ctypes.c_int(min(k, N)),
                            out_img.ctypes.data_as(ctypes.c_void_p), out_cls.ctypes.data_as(ctypes.c_void_p))
    return [paths[i] for i in out_img[:m]], [class_labels[c] for c in out_cls[:m]]
