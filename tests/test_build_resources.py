"""Build-time guard for the hot kernels: hipcc's resource remarks must show no scratch (register spills) for the GEMM
instantiations and the ViT-B/16 attention kernels.  A spill here does not fail any numeric test -- it silently costs 10x
(seen once: a helper taking the accumulator array by reference kept 150 registers of the 256x128 GEMM in scratch)."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "menghini-neurips23-code_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _resources(src):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur and m.group(1) not in res[cur]:
            res[cur][m.group(1)] = int(m.group(2))
    return res


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_gemm_kernels_do_not_spill():
    res = _resources("gemm.hip")
    gemms = {k: v for k, v in res.items() if "gemm_" in k}
    assert len(gemms) >= 9 * 5, sorted(gemms)          # >= 9 epilogues x (128x128, 64x128, ring 256x256, ring 256x128, k64 / k64p)
    # <= 7 dwords tolerated (k64 residual epilogue).  The LayerNorm-folded persistent kernels (gemm_k64p_kernel<7|8>) carry the
    # preloaded row statistics through the K loop; the compiler makes room by parking 13-21 loop-INVARIANT epilogue address
    # registers in scratch before the tile loop and reloading them once per tile after the K loop (checked in the ISA: no
    # scratch instruction inside the K loop).  Anything beyond that -- or any spill in another kernel -- fails here.
    def limit(k):
        return 96 if ("gemm_k64p_kernelILi7E" in k or "gemm_k64p_kernelILi8E" in k) else 28
    spilled = {k: v for k, v in gemms.items() if v.get("ScratchSize [bytes/lane]", 0) > limit(k)}
    assert not spilled, spilled
    for k, v in gemms.items():
        if "gemm_big_kernel" in k or "gemm_k64_kernel" in k:
            assert v["Occupancy [waves/SIMD]"] >= 2, (k, v)    # two waves per SIMD: the design point of the 8-wave tiles


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_vit_b16_attention_kernels_do_not_spill():
    res = _resources("attention.hip")
    hot = [k for k in res if "attn_fwd_pipe_kernelILi7ELi16E" in k or "attn_fwd_kernelILi7ELb0ELi8E" in k or "attn_fwd_kernelILi1ELb1ELi4E" in k]
    assert len(hot) == 3, sorted(res)
    for k in hot:
        assert res[k].get("ScratchSize [bytes/lane]", 0) == 0, (k, res[k])
    pipe = [k for k in hot if "pipe" in k][0]
    assert res[pipe]["VGPRs"] + res[pipe].get("AGPRs", 0) <= 128, res[pipe]     # 16 waves per workgroup = 4 per SIMD
    resb = _resources("attention_bwd.hip")
    for k, v in resb.items():
        if "attn_bwd_kernelILi7ELb0ELi8E" in k or "attn_bwd_kernelILi1ELb1ELi4E" in k:
            assert v.get("ScratchSize [bytes/lane]", 0) == 0, (k, v)
