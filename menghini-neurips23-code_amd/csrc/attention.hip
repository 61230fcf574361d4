// Fused multi-head self-attention for the CLIP towers (head dim 64, S <= 640): the whole K and V of
// one (image, head) live in LDS, a workgroup of 8 waves (4 for S <= 96) walks the 16-row query tiles: two workgroups
// fit a CU's LDS at S = 197, so 8 waves each puts 4 waves on every SIMD to cover the staging and softmax latency.
//
//   scores^T = K (Q/8)^T   per 16x16 tile with v_mfma_f32_16x16x32_f16 (A = K rows from LDS,
//                          B = Q rows straight from HBM); the transposed product leaves every lane
//                          with the scores of ONE query row (its lane&15), so the softmax row
//                          reductions are in-register + two xor-shuffles (lanes 16/32 apart).
//   P = exp(s - max)       f32, packed to f16 in place: two score tiles form one 32-wide K chunk of
//                          the second MFMA without any cross-lane movement.
//   O^T = V^T P^T          A = V^T fragments read as 2 x ds_read_b64 from a transposed LDS copy
//                          (row stride SP+8 halfs => conflict-free), B = the packed P registers.
//                          Each lane ends with 4 consecutive head-dim values of its own query row,
//                          divides by its own row sum and stores 8 bytes.
// K is stored [kv][64] with the 16-byte chunk index XOR (kv & 7): conflict-free ds_read_b128.
// Padding keys (kv >= S) are zero-filled and masked to -inf; causal masking for the text tower.
#include <math.h>

#include "common.h"

template <int KVC, bool CAUSAL, int ATT_NW>
__global__ __launch_bounds__(ATT_NW * 64) void attn_fwd_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int S, int H) {
    constexpr int SP = KVC * 32;
    constexpr int VST = SP + 8;
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Ks = (half_t*)smem;
    half_t* Vt = Ks + SP * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = H * 64;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t ld = (size_t)3 * D;
    const half_t* base = qkv + (size_t)b * S * ld + h * 64;
    const int li = lane & 15, lg = lane >> 4;
    const int n_qt = (S + 15) >> 4;

    // Q fragments of this wave's first tile go out before the K/V staging so their latency hides behind it;
    // inside the loop the NEXT tile's Q is fetched while the current one is being processed.
    auto load_q = [&](int qt, half8 (&qf)[2]) {
        const int qrow = qt * 16 + li;
        const int qr = qrow < S ? qrow : S - 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const half8*)(base + qr * ld + (kk * 4 + lg) * 8);
    };
    half8 q_next[2];
    load_q(wave < n_qt ? wave : 0, q_next);

    // K: [kv][64] rows, 16-byte chunk index XOR (kv & 7).
    for (int idx = tid; idx < SP * 8; idx += ATT_NW * 64) {
        const int row = idx >> 3, chunk = idx & 7;
        half8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < S) kv = *(const half8*)(base + row * ld + D + chunk * 8);
        *(half8*)(Ks + row * 64 + ((chunk ^ (row & 7)) * 8)) = kv;
    }
    // V^T: [64][VST].  A lane takes a PAIR of keys (2r, 2r+1) and one 8-wide slice of the head dim and writes
    // eight 32-bit words {V[2r][d], V[2r+1][d]}: lanes 0-31 cover 32 consecutive words of one d row (no bank
    // conflict), lanes 32-63 the neighbouring slice (other half-wave group of ds_write_b32).
    for (int idx = tid; idx < ((SP / 2 + 31) / 32) * 256; idx += ATT_NW * 64) {
        const int lane_rp = idx & 31, chunk = ((idx >> 5) & 1) + 2 * ((idx >> 6) & 3), rblk = idx >> 8;
        const int rp = rblk * 32 + lane_rp;          // key pair index
        const int r0 = 2 * rp;
        if (r0 >= SP) continue;
        half8 v0 = {0, 0, 0, 0, 0, 0, 0, 0}, v1 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 < S) v0 = *(const half8*)(base + r0 * ld + 2 * D + chunk * 8);
        if (r0 + 1 < S) v1 = *(const half8*)(base + (r0 + 1) * ld + 2 * D + chunk * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) *(half2v*)(Vt + (chunk * 8 + j) * VST + r0) = (half2v){v0[j], v1[j]};
    }
    __syncthreads();

    for (int qt = wave; qt < n_qt; qt += ATT_NW) {
        asm volatile("" ::: "memory");  // keep the K/V fragment reads inside the tile loop (hoisting them costs >100 VGPRs)
        const int qrow = qt * 16 + li;
        half8 qf[2] = {q_next[0], q_next[1]};
        if (qt + ATT_NW < n_qt) load_q(qt + ATT_NW, q_next);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] *= (half_t)0.125f;  // 1/sqrt(64), exact in f16
        f32x4 sc[2 * KVC];
        float m = -INFINITY;
        // The launcher guarantees (KVC-1)*32 < S <= KVC*32: every tile before the last 32-key chunk is full, so
        // only the last two tiles (and causal rows) pay for the mask; the code stays one straight-line block.
#pragma unroll
        for (int t = 0; t < 2 * KVC; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8 kf = *(const half8*)(Ks + (t * 16 + li) * 64 + (((kk * 4 + lg) ^ (lane & 7)) * 8));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], acc, 0, 0, 0);
            }
            if (CAUSAL || t >= 2 * (KVC - 1)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kv = t * 16 + lg * 4 + r;
                    if (kv >= S || (CAUSAL && kv > qrow)) acc[r] = -INFINITY;
                }
            }
            m = fmaxf(fmaxf(m, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
            sc[t] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float m2 = m * LOG2E;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * KVC; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][r], LOG2E, -m2));   // exp(s - m)
                sc[t][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);

        f32x4 o[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) o[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KVC; ++c) {
            const half8 pf = {(half_t)sc[2 * c][0], (half_t)sc[2 * c][1], (half_t)sc[2 * c][2], (half_t)sc[2 * c][3],
                              (half_t)sc[2 * c + 1][0], (half_t)sc[2 * c + 1][1], (half_t)sc[2 * c + 1][2], (half_t)sc[2 * c + 1][3]};
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const half_t* vp = Vt + (nf * 16 + li) * VST + c * 32 + lg * 4;
                const half4 v0 = *(const half4*)vp;
                const half4 v1 = *(const half4*)(vp + 16);
                const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[nf], 0, 0, 0);
            }
        }
        if (qrow < S) {
            const float inv = __builtin_amdgcn_rcpf(sum);
            half_t* op = out + ((size_t)b * S + qrow) * D + h * 64 + lg * 4;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const f32x4 v = o[nf] * inv;
                *(half4*)(op + nf * 16) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            }
        }
    }
}

template <int KVC, bool CAUSAL, int ATT_NW>
static int launch_one(const half_t* qkv, half_t* out, int B, int S, int H, hipStream_t s) {
    constexpr int SP = KVC * 32;
    constexpr size_t lds = (size_t)SP * 64 * 2 + (size_t)64 * (SP + 8) * 2;
    static bool configured = false;
    if (!configured) {
        GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel<KVC, CAUSAL, ATT_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<KVC, CAUSAL, ATT_NW>), dim3(B * H), dim3(ATT_NW * 64), lds, s, qkv, out, S, H);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

int launch_attention_fwd(const half_t* qkv, half_t* out, int B, int S, int H, int causal, hipStream_t s) {
    const int kvc = (S + 31) / 32;
    GRIP_REQUIRE(S >= 1 && kvc <= 19, "attention: sequence length %d unsupported (max 608)", S);
    // exact chunk count (the kernel relies on (KVC-1)*32 < S); 11..18 share the 19-chunk build via the slow mask path
#define GRIP_ATTN(N)                                                        \
    if (kvc == N) return causal ? launch_one<N, true, (N >= 4 ? 8 : 4)>(qkv, out, B, S, H, s) \
                                : launch_one<N, false, (N >= 4 ? 8 : 4)>(qkv, out, B, S, H, s);
    GRIP_ATTN(1) GRIP_ATTN(2) GRIP_ATTN(3) GRIP_ATTN(4) GRIP_ATTN(5) GRIP_ATTN(6) GRIP_ATTN(7) GRIP_ATTN(8) GRIP_ATTN(9) GRIP_ATTN(10)
    GRIP_ATTN(11) GRIP_ATTN(12) GRIP_ATTN(13) GRIP_ATTN(14) GRIP_ATTN(15) GRIP_ATTN(16) GRIP_ATTN(17) GRIP_ATTN(18) GRIP_ATTN(19)
#undef GRIP_ATTN
    return GRIP_ERR_ARG;
}
