import sys, torch
sys.path.insert(0, '.')
import grip_amd
from grip_amd import clip
m, _ = clip.load('ViT-B/16', device='cuda')
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn(2000, 3, 224, 224, device='cuda', generator=g)
t = m.visual.tower
outs = []
for chunk in (1320, 880, 700, 2000):
    o = torch.empty(2000, 512, device='cuda')
    with torch.no_grad():
        t.encode_chunks(x, o, 0, 2000, chunk, streams=1)
    torch.cuda.synchronize()
    outs.append(o.clone())
print('chunk invariance (bitwise):', [bool(torch.equal(outs[0], o)) for o in outs[1:]])
for chunk, o in zip((880, 700, 2000), outs[1:]):
    bad = (outs[0] != o).any(dim=1).nonzero().flatten()
    if len(bad):
        print(f"chunk {chunk}: {len(bad)} rows differ, first {int(bad[0])} last {int(bad[-1])}, max abs diff {float((outs[0] - o).abs().max()):.3e}")
