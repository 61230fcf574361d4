mkdir -p gpurun_out
for m in 1 2 1 2; do
  GRIP_TRAIN_FOLD=$m python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import grip_amd
from grip_amd import clip, rng, steps
from grip_amd.models import CustomImageEncoder, CustomTextEncoder, ImagePrefixModel, UPTModel
dev = "cuda"
m, _ = clip.load("ViT-B/16", device=dev)
B = 16
x = torch.randn(B, 3, 224, 224, device=dev)
scale = m.logit_scale.exp().item()
w = torch.full((B,), 1.0 / B, device=dev)
def N(name, shape, std=0.02): return torch.from_numpy(rng.normal(1, rng.stream_id(name), shape, 0.0, std)).to(dev)
def bench(g, *a):
    for _ in range(10): g(*a)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(100): g(*a)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 100)
    return best * 1e3
C = 45
txt = m.encode_text(clip.tokenize([f"a photo of a class {i}" for i in range(C)]).to(dev))
im = ImagePrefixModel(N("v", (16, 768)), CustomImageEncoder(m.visual), device=dev)
opt = torch.optim.SGD([im.prefix], lr=0.01, weight_decay=0.1)
y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
v = bench(steps.GraphedVptStep(im, txt, scale, opt), x, y, w)
C = 47
classes = [f"class {i}" for i in range(C)]
um = UPTModel(N("uc", (1, 4, 512)), N("uv", (1, 4, 768)), None, CustomImageEncoder(m.visual), CustomTextEncoder(m, dev, torch.float32), classes, 128, device=dev, dtype=torch.float32)
opt2 = torch.optim.SGD([p for p in um.parameters() if p.requires_grad], lr=0.01, weight_decay=0.1)
y2 = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
u = bench(steps.GraphedUptStep(um, scale, opt2), x, y2, w)
print(f"GRIP_TRAIN_FOLD={os.environ['GRIP_TRAIN_FOLD']}: graphed VPT step {v:.3f} ms, UPT step {u:.3f} ms")
PY
done 2>&1 | grep TRAIN_FOLD | tee gpurun_out/vision_fold_ab.txt
