// Split-f16 multi-head self-attention (dims.precision = 2, the middle tier of screen-and-refine): softmax(Q K^T / 8) V per (image, head)
// with both matrix products on the f16 matrix pipes at ~22 mantissa bits, the arithmetic of gemm_split.hip:
//     x = hi + lo' / 2^11,  hi = f16(x),  lo' = f16((x - hi) 2^11);    a b ~ a_hi b_hi + (a_hi b_lo' + a_lo' b_hi) / 2^11     (three MFMAs)
// Same structure as attn_fwd_kernel (attention.hip): the K rows and a blocked image of V of one (image, head) in LDS -- here as a hi and a
// lo' plane each --, scores^T = K (Q/8)^T leaves every lane with the scores of ONE query row, f32 softmax in registers, P split in
// registers into the B operand of the second product, O^T = V^T P^T with the head dims permuted (vt_index_fwd) so that a lane ends with
// sixteen consecutive output columns.  Input: the packed f32 projection qkv [B*S, 3*H*64] (what the QKV GEMM of a precision-2 tower
// writes); output: [B*S, H*64] in the split layout of gemm_split.hip (it feeds the out-proj GEMM).  The f32 tower's attention
// (attention_f32.hip: one thread per query row on the vector ALUs, 2.8 TFLOP/s) was 26 % of a split-tower pass; this is < 3 %.
// S <= 320 (four LDS planes of S x 64 halfs); longer sequences (ViT-L/14@336px) keep the f32 kernel with split output.
#include <math.h>

#include "common.h"

__device__ __forceinline__ int vt_index_fwd_s(int key, int d) { return vt_index(key, ((d >> 2) & 3) * 16 + (d >> 4) * 4 + (d & 3)); }

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, half8& hi, half8& lo) {
    half4 h0, l0, h1, l1;
    split_f16x4(a, h0, l0);
    split_f16x4(b, h1, l1);
    hi = (half8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    lo = (half8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

template <int KVC, bool CAUSAL, int NW>       // NW waves per workgroup: 8 from 97 tokens on (two waves per SIMD: staging and the 13 - 20 query tiles in two or three rounds), else 4
__global__ __launch_bounds__(NW * 64) void attn_fwd_split_kernel(const float* __restrict__ qkv, half_t* __restrict__ out, int S, int H) {
    constexpr int SP = KVC * 32;
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr float INV = 1.0f / (float)GRIP_SPLIT_LO_SCALE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Kh = (half_t*)smem;           // [SP][64], 16-byte chunk index XOR (key & 7)
    half_t* Kl = Kh + SP * 64;
    half_t* Vh = Kl + SP * 64;            // blocked V image (vt_index_fwd_s)
    half_t* Vl = Vh + SP * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = H * 64;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)b * S * ld + h * 64;
    const int li = lane & 15, lg = lane >> 4;
    const int n_qt = (S + 15) >> 4;

    for (int idx = tid; idx < SP * 8; idx += NW * 64) {
        const int row = idx >> 3, chunk = idx & 7;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
        if (row < S) {
            const f32x4* p = (const f32x4*)(base + (size_t)row * ld + D + chunk * 8);
            a = p[0];
            c = p[1];
        }
        half8 hi, lo;
        split8(a, c, hi, lo);
        const int o = row * 64 + ((chunk ^ (row & 7)) * 8);
        *(half8*)(Kh + o) = hi;
        *(half8*)(Kl + o) = lo;
    }
    // V image: a lane takes a PAIR of keys (2r, 2r+1) and one 8-wide slice of the head dim and writes eight 32-bit words {V[2r][d], V[2r+1][d]} per plane
    for (int idx = tid; idx < ((SP / 2 + 31) / 32) * 256; idx += NW * 64) {
        const int lane_rp = idx & 31, chunk = ((idx >> 5) & 1) + 2 * ((idx >> 6) & 3), rblk = idx >> 8;
        const int r0 = 2 * (rblk * 32 + lane_rp);
        if (r0 >= SP) continue;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 a0 = z, a1 = z, b0 = z, b1 = z;
        if (r0 < S) { const f32x4* p = (const f32x4*)(base + (size_t)r0 * ld + 2 * D + chunk * 8); a0 = p[0]; a1 = p[1]; }
        if (r0 + 1 < S) { const f32x4* p = (const f32x4*)(base + (size_t)(r0 + 1) * ld + 2 * D + chunk * 8); b0 = p[0]; b1 = p[1]; }
        half8 h0, l0, h1, l1;
        split8(a0, a1, h0, l0);
        split8(b0, b1, h1, l1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = vt_index_fwd_s(r0, chunk * 8 + j);
            *(half2v*)(Vh + o) = (half2v){h0[j], h1[j]};
            *(half2v*)(Vl + o) = (half2v){l0[j], l1[j]};
        }
    }
    __syncthreads();

    for (int qt = wave; qt < n_qt; qt += NW) {
        asm volatile("" ::: "memory");   // keep the fragment reads inside the tile loop
        const int qrow = qt * 16 + li;
        const int qr = qrow < S ? qrow : S - 1;
        half8 qh[2], ql[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const f32x4* p = (const f32x4*)(base + (size_t)qr * ld + (kk * 4 + lg) * 8);
            split8(p[0] * 0.125f, p[1] * 0.125f, qh[kk], ql[kk]);      // 1/sqrt(64): exact scaling
        }
        f32x4 sc[2 * KVC];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * KVC; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, cor = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int o = (t * 16 + li) * 64 + (((kk * 4 + lg) ^ (lane & 7)) * 8);
                const half8 kh = *(const half8*)(Kh + o), kl = *(const half8*)(Kl + o);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[kk], acc, 0, 0, 0);
                cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[kk], cor, 0, 0, 0);
                cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[kk], cor, 0, 0, 0);
            }
            acc += cor * INV;
            if (CAUSAL || t >= 2 * (KVC - 1)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kv = t * 16 + lg * 4 + r;
                    if (kv >= S || (CAUSAL && kv > qrow)) acc[r] = -INFINITY;
                }
            }
            m = fmaxf(fmaxf(fmaxf(m, acc[0]), fmaxf(acc[1], acc[2])), acc[3]);
            sc[t] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float m2 = m * LOG2E;
#pragma unroll
        for (int t = 0; t < 2 * KVC; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][r], LOG2E, -m2));   // exp(s - m)

        // row sums out of the matrix pipe (all-ones A operand), over the same hi / lo' numerators the P.V product uses
        const half8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
        f32x4 sum_h = {0.f, 0.f, 0.f, 0.f}, sum_l = {0.f, 0.f, 0.f, 0.f};
        f32x4 o[4], oc[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) { o[nf] = (f32x4){0.f, 0.f, 0.f, 0.f}; oc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < KVC; ++c) {
            half8 ph, pl;
            split8(sc[2 * c], sc[2 * c + 1], ph, pl);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const int vo = (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8;
                const half8 vh = *(const half8*)(Vh + vo), vl = *(const half8*)(Vl + vo);
                o[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[nf], 0, 0, 0);
                oc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, oc[nf], 0, 0, 0);
                oc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, oc[nf], 0, 0, 0);
            }
            sum_h = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ph, sum_h, 0, 0, 0);
            sum_l = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pl, sum_l, 0, 0, 0);
        }
        if (qrow < S) {
            const float inv = 1.0f / (sum_h[0] + sum_l[0] * INV);
            // lane (li, lg) holds columns h*64 + lg*16 + nf*4 + r of its query row: sixteen consecutive columns inside one 32-column group
            half_t* orow = out + (size_t)((size_t)b * S + qrow) * 2 * D;
            const int col = h * 64 + lg * 16;
            half_t* op = orow + (col >> 5) * 64 + (col & 31);
            half8 hi0, lo0, hi1, lo1;
            split8((o[0] + oc[0] * INV) * inv, (o[1] + oc[1] * INV) * inv, hi0, lo0);
            split8((o[2] + oc[2] * INV) * inv, (o[3] + oc[3] * INV) * inv, hi1, lo1);
            *(half8*)op = hi0;
            *(half8*)(op + 8) = hi1;
            *(half8*)(op + 32) = lo0;
            *(half8*)(op + 40) = lo1;
        }
    }
}

bool attention_split_supported(int S) { return S >= 1 && S <= 320; }

int launch_attention_fwd_split(const float* qkv, void* out, int B, int S, int H, int causal, hipStream_t s) {
    GRIP_REQUIRE(attention_split_supported(S) && B >= 1 && H >= 1, "attention_split: bad shape B=%d S=%d H=%d (S <= 320)", B, S, H);
    const int kvc = (S + 31) / 32;
    const size_t lds = (size_t)4 * kvc * 32 * 64 * sizeof(half_t);
#define GRIP_ATT_CASE(K)                                                                                                                     \
    case K: {                                                                                                                                \
        static bool configured = false;                                                                                                      \
        if (!configured) {                                                                                                                   \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_split_kernel<K, false, (K >= 4 ? 8 : 4)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_split_kernel<K, true, (K >= 4 ? 8 : 4)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
            configured = true;                                                                                                               \
        }                                                                                                                                    \
        if (causal) hipLaunchKernelGGL((attn_fwd_split_kernel<K, true, (K >= 4 ? 8 : 4)>), dim3(B * H), dim3((K >= 4 ? 8 : 4) * 64), lds, s, qkv, (half_t*)out, S, H); \
        else hipLaunchKernelGGL((attn_fwd_split_kernel<K, false, (K >= 4 ? 8 : 4)>), dim3(B * H), dim3((K >= 4 ? 8 : 4) * 64), lds, s, qkv, (half_t*)out, S, H);       \
    } break;
    switch (kvc) {
        GRIP_ATT_CASE(1) GRIP_ATT_CASE(2) GRIP_ATT_CASE(3) GRIP_ATT_CASE(4) GRIP_ATT_CASE(5)
        GRIP_ATT_CASE(6) GRIP_ATT_CASE(7) GRIP_ATT_CASE(8) GRIP_ATT_CASE(9) GRIP_ATT_CASE(10)
        default: GRIP_REQUIRE(false, "attention_split: S = %d out of range", S);
    }
#undef GRIP_ATT_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
