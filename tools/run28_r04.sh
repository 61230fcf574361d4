mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainfold.py tests/test_gpu_towers.py tests/test_gpu_determinism.py -q -m gpu -x > gpurun_out/t_sub.log 2>&1
grep -E "passed|failed|rror|assert" gpurun_out/t_sub.log | tail -n 5
for m in 1 2; do
  GRIP_TRAIN_FOLD=$m timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_trajectory.py -q -m gpu -x > gpurun_out/t_f$m.log 2>&1; grep -E "passed|failed" gpurun_out/t_f$m.log | tail -n 1
  GRIP_TRAIN_FOLD=$m python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import grip_amd
from grip_amd import clip, rng, steps
from grip_amd.models import CustomImageEncoder, ImagePrefixModel
dev = "cuda"
m, _ = clip.load("ViT-B/16", device=dev)
B, C = 16, 45
x = torch.randn(B, 3, 224, 224, device=dev)
scale = m.logit_scale.exp().item()
w = torch.full((B,), 1.0 / B, device=dev)
txt = m.encode_text(clip.tokenize([f"a photo of a class {i}" for i in range(C)]).to(dev))
im = ImagePrefixModel(torch.from_numpy(rng.normal(1, rng.stream_id("v"), (16, 768), 0.0, 0.02)).to(dev), CustomImageEncoder(m.visual), device=dev)
opt = torch.optim.SGD([im.prefix], lr=0.01, weight_decay=0.1)
y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
g = steps.GraphedVptStep(im, txt, scale, opt)
for _ in range(10): g(x, y, w)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t = time.perf_counter()
    for _ in range(100): g(x, y, w)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / 100)
print(f"GRIP_TRAIN_FOLD={os.environ['GRIP_TRAIN_FOLD']}: graphed VPT step {best * 1e3:.3f} ms")
PY
done
