// Exact-mode multi-head self-attention, all f32 on the vector ALUs (head dim 64): the comparison mode's counterpart of
// attention.hip.  softmax(Q K^T / 8) V per (image, head) exactly as nn.MultiheadAttention computes it in the published
// openai/CLIP ResidualAttentionBlock (q scaled by 1/8 before the product, additive -inf causal mask in the text tower),
// with the running-maximum form of the softmax so any sequence length works from 16 KiB of LDS.
//
// One thread owns one query row: its 64 q values and 64 output accumulators live in registers; the keys / values of the
// head stream through LDS 32 rows at a time and are read as wave-wide broadcasts (every lane the same address, one LDS
// cycle group per ds_read_b128).  Per tile: 32 scores, tile maximum, one rescale of the accumulators, 32 exp, P.V.
// Throughput is bounded by the LDS broadcast reads (32 ds_read_b128 per key per wave); the attention core is 4 % of the
// tower's FLOPs and this path exists for bit-level comparison, not speed.
#include <math.h>

#include "common.h"

template <bool CAUSAL, bool SPLIT = false>     // SPLIT: `out` in the split layout of gemm_split.hip (precision-2 towers: it feeds the out-proj GEMM)
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S, int H, int qblocks) {
    __shared__ f32x4 Ks[32][16];
    __shared__ f32x4 Vs[32][16];
    const int tid = threadIdx.x;
    const int bh = blockIdx.x / qblocks, qb = blockIdx.x - bh * qblocks;
    const int b = bh / H, h = bh - b * H;
    const int D = H * 64;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)b * S * ld + h * 64;
    const int qi = qb * 256 + tid;
    const int qr = qi < S ? qi : S - 1;

    f32x4 q[16], o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        q[c] = ((const f32x4*)(base + (size_t)qr * ld))[c] * 0.125f;   // 1/sqrt(64), exact
        o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float m = -INFINITY, l = 0.f;
    const int kend = CAUSAL ? (qb * 256 + 256 < S ? qb * 256 + 256 : S) : S;   // keys any query of this block can see
    for (int k0 = 0; k0 < kend; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = (tid >> 4) + rr * 16, c = tid & 15, key = k0 + r;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < S) {
                kv = ((const f32x4*)(base + (size_t)key * ld + D))[c];
                vv = ((const f32x4*)(base + (size_t)key * ld + 2 * D))[c];
            }
            Ks[r][c] = kv;
            Vs[r][c] = vv;
        }
        __syncthreads();
        float s[32];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 kf = Ks[j][c];
                a = fmaf(q[c][0], kf[0], a); a = fmaf(q[c][1], kf[1], a); a = fmaf(q[c][2], kf[2], a); a = fmaf(q[c][3], kf[3], a);
            }
            const int key = k0 + j;
            if (key >= S || (CAUSAL && key > qi)) a = -INFINITY;
            s[j] = a;
            tmax = fmaxf(tmax, a);
        }
        const float mn = fmaxf(m, tmax);          // finite from the first tile on: key 0 is visible to every query
        const float corr = expf(m - mn);          // exp(-inf) = 0 on the first tile
        l *= corr;
#pragma unroll
        for (int c = 0; c < 16; ++c) o[c] = o[c] * corr;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float p = expf(s[j] - mn);
            l += p;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 vf = Vs[j][c];
                o[c][0] = fmaf(p, vf[0], o[c][0]); o[c][1] = fmaf(p, vf[1], o[c][1]); o[c][2] = fmaf(p, vf[2], o[c][2]); o[c][3] = fmaf(p, vf[3], o[c][3]);
            }
        }
        m = mn;
    }
    if (qi < S) {
        if constexpr (SPLIT) {
            const SplitRow r{(half_t*)out + ((size_t)b * S + qi) * 2 * D};
#pragma unroll
            for (int c = 0; c < 16; ++c) store4(r, h * 16 + c, o[c] / l);
        } else {
            f32x4* op = (f32x4*)(out + ((size_t)b * S + qi) * D + h * 64);
#pragma unroll
            for (int c = 0; c < 16; ++c) op[c] = o[c] / l;
        }
    }
}

int launch_attention_fwd_f32(const float* qkv, float* out, int B, int S, int H, int causal, hipStream_t s, int split_out) {
    GRIP_REQUIRE(S >= 1 && B >= 1 && H >= 1, "attention_f32: bad shape B=%d S=%d H=%d", B, S, H);
    const int qblocks = (S + 255) / 256;
    if (split_out) {
        if (causal) hipLaunchKernelGGL((attn_fwd_f32_kernel<true, true>), dim3(B * H * qblocks), dim3(256), 0, s, qkv, out, S, H, qblocks);
        else hipLaunchKernelGGL((attn_fwd_f32_kernel<false, true>), dim3(B * H * qblocks), dim3(256), 0, s, qkv, out, S, H, qblocks);
    } else if (causal)
        hipLaunchKernelGGL(attn_fwd_f32_kernel<true>, dim3(B * H * qblocks), dim3(256), 0, s, qkv, out, S, H, qblocks);
    else
        hipLaunchKernelGGL(attn_fwd_f32_kernel<false>, dim3(B * H * qblocks), dim3(256), 0, s, qkv, out, S, H, qblocks);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// ONE query row per sequence (the last block of an inference forward of the f32 / split-f16 towers: only the CLS / EOT row of the final stream is
// ever read, csrc/tower.hip run_blocks): out[b] = softmax(q_b K_b^T / 8) V_b per head, all f32 on the vector ALUs.  One wave per (sequence, head):
// lanes over keys for the scores (64 fused multiply-adds per key, accurate expf as in the kernel above), then lane = (key group, 8-wide slice of the
// head dim) for P.V with a fixed fold of the eight key groups.  qrows [B, D] f32 = the projected query rows; qkv [B * S, 3 D] f32 holds K and V.
template <bool SPLIT>
__global__ __launch_bounds__(64) void attn_row_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ qrows, const int32_t* __restrict__ row_index,
                                                          float* __restrict__ out, int S, int H, int causal) {
    extern __shared__ float sm_row[];     // [64] q, [S] probabilities
    float* qs = sm_row;
    float* ps = sm_row + 64;
    const int lane = threadIdx.x;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * 64;
    const size_t ld = (size_t)3 * D;
    const int r = row_index ? row_index[b] : 0;
    const float* base = qkv + (size_t)b * S * ld + h * 64;
    qs[lane] = qrows[(size_t)b * D + h * 64 + lane] * 0.125f;       // 1/sqrt(64), exact
    __syncthreads();
    const int n_keys = causal ? r + 1 : S;
    float m = -INFINITY;
    for (int j = lane; j < n_keys; j += 64) {
        const f32x4* kr = (const f32x4*)(base + (size_t)j * ld + D);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 kv = kr[c];
            a = fmaf(qs[c * 4 + 0], kv[0], a); a = fmaf(qs[c * 4 + 1], kv[1], a); a = fmaf(qs[c * 4 + 2], kv[2], a); a = fmaf(qs[c * 4 + 3], kv[3], a);
        }
        ps[j] = a;
        m = fmaxf(m, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int j = lane; j < n_keys; j += 64) {
        const float p = expf(ps[j] - m);
        ps[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int kg = lane >> 3, ch = lane & 7;          // key group, 8-wide slice of the head dim
    for (int j = kg; j < n_keys; j += 8) {
        const f32x4* vr = (const f32x4*)(base + (size_t)j * ld + 2 * D + ch * 8);
        const f32x4 v0 = vr[0], v1 = vr[1];
        const float p = ps[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[0][e] = fmaf(p, v0[e], acc[0][e]); acc[1][e] = fmaf(p, v1[e], acc[1][e]); }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[u][e] += __shfl_xor(acc[u][e], 8);
            acc[u][e] += __shfl_xor(acc[u][e], 16);
            acc[u][e] += __shfl_xor(acc[u][e], 32);
        }
    if (kg == 0) {
        const f32x4 o0 = acc[0] / sum, o1 = acc[1] / sum;
        if constexpr (SPLIT) {
            const SplitRow row{(half_t*)out + (size_t)b * 2 * D};
            store4(row, h * 16 + ch * 2, o0);
            store4(row, h * 16 + ch * 2 + 1, o1);
        } else {
            f32x4* op = (f32x4*)(out + (size_t)b * D + h * 64 + ch * 8);
            op[0] = o0;
            op[1] = o1;
        }
    }
}

int launch_attention_row_f32(const float* qkv, const float* qrows, const int32_t* row_index, float* out, int B, int S, int H, int causal, hipStream_t s, int split_out) {
    GRIP_REQUIRE(S >= 1 && B >= 1 && H >= 1 && qrows, "attention_row_f32: bad arguments B=%d S=%d H=%d", B, S, H);
    const size_t lds = (size_t)(64 + S) * sizeof(float);
    if (split_out) hipLaunchKernelGGL(attn_row_f32_kernel<true>, dim3(B * H), dim3(64), lds, s, qkv, qrows, row_index, out, S, H, causal);
    else hipLaunchKernelGGL(attn_row_f32_kernel<false>, dim3(B * H), dim3(64), lds, s, qkv, qrows, row_index, out, S, H, causal);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
