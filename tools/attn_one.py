"""Developer probe: run the fused attention forward repeatedly (timing / rocprofv3 --pmc)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import native  # noqa: E402

lib = native.lib()
B, S, H, causal, reps = [int(a) for a in sys.argv[1:6]]
p = lambda t: ctypes.c_void_p(t.data_ptr())
D = H * 64
qkv = torch.randn(B * S, 3 * D, device="cuda").half()
out = torch.empty(B * S, D, device="cuda", dtype=torch.float16)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f = lambda: native.check(lib.grip_debug_attention(p(qkv), p(out), B, S, H, causal, s))
for _ in range(3):
    f()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    f()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print(f"attention B={B} S={S} H={H} causal={causal}: {ms * 1e3:.1f} us, {4.0 * B * H * S * S * 64 / ms / 1e9:.0f} TF/s (algorithmic)")
