// Row-wise, HBM-bound kernels of the CLIP towers: patch gather (im2col), sequence assembly with
// prompt insertion + LayerNorm, LayerNorm of the f16 residual stream (f32 statistics) -> f16 GEMM operand, CLS/EOT gather + LayerNorm, token-embedding
// gather with prompt splice, f16 transpose.  One wave (64 lanes) owns one row of width d and keeps
// it in registers as float4 chunks (lane l holds float4 index l + 64*i): every global access is a
// 16-byte, fully coalesced access, statistics are two in-register passes + a 6-step xor reduction.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows of x (residual stream, f16; or f32 for the test hook) -> f16.  row_index == null: row r reads x[r]; else row r reads
// x[r * row_stride + row_index[r]] (EOT gather); row_stride alone gathers x[r * row_stride] (CLS).
template <int NV, typename XT, typename OT = half_t, bool SPLIT = false>       // SPLIT: OT = half_t, rows written in the split layout (common.h)
__global__ __launch_bounds__(256) void ln_f16_kernel(const XT* __restrict__ x, const int32_t* __restrict__ row_index, int row_stride,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     OT* __restrict__ out, int n_rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int d4 = d >> 2;
    size_t src = (size_t)row * row_stride + (row_index ? row_index[row] : 0);
    const XT* xr = x + src * d;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) v[i] = load4(xr, lane + 64 * i);
    float mean, rstd;
    ln_normalize<NV>(v, lane, d4, d, mean, rstd);
    auto orow = [&]() {
        if constexpr (SPLIT) return SplitRow{(half_t*)out + (size_t)row * 2 * d};
        else return out + (size_t)row * d;
    };
    const auto o = orow();
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            const f32x4 g = ((const f32x4*)gamma)[lane + 64 * i], b = ((const f32x4*)beta)[lane + 64 * i];
            store4(o, lane + 64 * i, ln_apply(v[i], mean, rstd, g, b));
        }
}

#define DISPATCH_NV(d, CALL)                                                                   \
    do {                                                                                       \
        const int _nv = ((d) / 4 + 63) / 64;                                                   \
        GRIP_REQUIRE((d) % 4 == 0 && _nv >= 1 && _nv <= MAX_NV, "row kernel: unsupported width %d", (d)); \
        switch (_nv) {                                                                         \
            case 1: { constexpr int NV = 1; CALL; } break;                                     \
            case 2: { constexpr int NV = 2; CALL; } break;                                     \
            case 3: { constexpr int NV = 3; CALL; } break;                                     \
            case 4: { constexpr int NV = 4; CALL; } break;                                     \
            case 5: case 6: { constexpr int NV = 6; CALL; } break;                             \
            default: { constexpr int NV = 8; CALL; } break;                                    \
        }                                                                                      \
    } while (0)

int launch_layernorm_f16(const void* x, const float* gamma, const float* beta, void* out, int f32, int M, int d, hipStream_t s) {
    if (f32 == 2) {
        GRIP_REQUIRE(d % 32 == 0, "layernorm (split layout): width %d %% 32 != 0", d);
        DISPATCH_NV(d, hipLaunchKernelGGL((ln_f16_kernel<NV, float, half_t, true>), dim3((M + 3) / 4), dim3(256), 0, s, (const float*)x, (const int32_t*)nullptr, 1, gamma, beta, (half_t*)out, M, d));
    } else if (f32) {
        DISPATCH_NV(d, hipLaunchKernelGGL((ln_f16_kernel<NV, float, float>), dim3((M + 3) / 4), dim3(256), 0, s, (const float*)x, (const int32_t*)nullptr, 1, gamma, beta, (float*)out, M, d));
    } else {
        DISPATCH_NV(d, hipLaunchKernelGGL((ln_f16_kernel<NV, resid_t>), dim3((M + 3) / 4), dim3(256), 0, s, (const resid_t*)x, (const int32_t*)nullptr, 1, gamma, beta, (half_t*)out, M, d));
    }
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

int launch_layernorm_f16_from_f32(const float* x, const float* gamma, const float* beta, half_t* out, int M, int d, hipStream_t s) {
    DISPATCH_NV(d, hipLaunchKernelGGL((ln_f16_kernel<NV, float>), dim3((M + 3) / 4), dim3(256), 0, s, x, (const int32_t*)nullptr, 1, gamma, beta, out, M, d));
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

int launch_gather_ln_f16(const void* x, const int32_t* row_index, int row_stride, const float* gamma, const float* beta,
                         void* out, int f32, int n_rows, int d, hipStream_t s) {
    if (f32) {
        DISPATCH_NV(d, hipLaunchKernelGGL((ln_f16_kernel<NV, float, float>), dim3((n_rows + 3) / 4), dim3(256), 0, s, (const float*)x, row_index, row_stride, gamma, beta, (float*)out, n_rows, d));
    } else {
        DISPATCH_NV(d, hipLaunchKernelGGL((ln_f16_kernel<NV, resid_t>), dim3((n_rows + 3) / 4), dim3(256), 0, s, (const resid_t*)x, row_index, row_stride, gamma, beta, (half_t*)out, n_rows, d));
    }
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Vision sequence assembly + ln_pre  (models/clip_encoders.py:135-163):
//   row (b, 0)            = class_embedding + pos[0]
//   row (b, 1..P)         = prefix[s-1]                         (no positional embedding)
//   row (b, 1+P+j)        = patch_out[b*G2 + j] + pos[1+j]
// then LayerNorm -> x (residual stream), S = 1 + P + G2.
template <int NV, typename RT>
__global__ __launch_bounds__(256) void vit_assemble_ln_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, const float* __restrict__ prefix, int P,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              RT* __restrict__ x, float* __restrict__ rowstat, int B, int G2, int d, half_t* __restrict__ x_lo) {
    const int lane = threadIdx.x & 63;
    const int S = 1 + P + G2;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * S) return;
    const int b = row / S, s = row - b * S;
    const int d4 = d >> 2;
    f32x4 v[NV];
    const f32x4* src;
    const f32x4* add = nullptr;
    if (s == 0) { src = (const f32x4*)cls; add = (const f32x4*)pos; }     // pos == nullptr: pos_emb=False (models/clip_encoders.py:141)
    else if (s <= P) { src = (const f32x4*)(prefix + (size_t)(s - 1) * d); }
    else { const int j = s - 1 - P; src = (const f32x4*)(patch_out + ((size_t)b * G2 + j) * d); add = pos ? (const f32x4*)(pos + (size_t)(1 + j) * d) : nullptr; }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            v[i] = src[lane + 64 * i];
            if (add) v[i] += add[lane + 64 * i];
        }
    float mean, rstd;
    ln_normalize<NV>(v, lane, d4, d, mean, rstd);
    RT* o = x + (size_t)row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            const f32x4 g = ((const f32x4*)gamma)[lane + 64 * i], bb = ((const f32x4*)beta)[lane + 64 * i];
            v[i] = ln_apply(v[i], mean, rstd, g, bb);
            store4(o, lane + 64 * i, v[i]);
            if constexpr (sizeof(RT) == 2) {
                if (x_lo) {     // compensated stream: what the f16 rounding of the value just stored dropped
                    const f32x4 r = {v[i][0] - (float)(half_t)v[i][0], v[i][1] - (float)(half_t)v[i][1], v[i][2] - (float)(half_t)v[i][2], v[i][3] - (float)(half_t)v[i][3]};
                    store4(x_lo + (size_t)row * d, lane + 64 * i, r);
                }
            }
        }
    if (rowstat) {     // statistics of the row just written, for the LayerNorm folded into the first QKV GEMM
        ln_normalize<NV>(v, lane, d4, d, mean, rstd);
        if (lane == 0) ((float2*)rowstat)[row] = make_float2(mean, rstd);
    }
}

int launch_vit_assemble_ln(const float* patch_out, const float* cls, const float* pos, const float* prefix, int P,
                           const float* gamma, const float* beta, void* x, int f32, float* rowstat, int B, int G2, int d, hipStream_t s, half_t* x_lo) {
    const int rows = B * (1 + P + G2);
    if (f32) {
        DISPATCH_NV(d, hipLaunchKernelGGL((vit_assemble_ln_kernel<NV, float>), dim3((rows + 3) / 4), dim3(256), 0, s, patch_out, cls, pos, prefix, P, gamma, beta, (float*)x, rowstat, B, G2, d, (half_t*)nullptr));
    } else {
        DISPATCH_NV(d, hipLaunchKernelGGL((vit_assemble_ln_kernel<NV, resid_t>), dim3((rows + 3) / 4), dim3(256), 0, s, patch_out, cls, pos, prefix, P, gamma, beta, (resid_t*)x, rowstat, B, G2, d, x_lo));
    }
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Text embedding (models/clip_encoders.py:63-74): x[c, t] = (1 <= t <= P ? prefix[c or 0, t-1] : tok_emb[ids[c, t]]) + pos[t];
// pos == nullptr is the reference's enable_pos_emb=False branch (:70-74): no positional term.
template <typename RT>
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ ids, int ld_ids, const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos, const float* __restrict__ prefix, int P,
                                                         int prefix_classes, RT* __restrict__ x, float* __restrict__ rowstat, int C, int T, int d, int vocab,
                                                         int Ps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    // Ps > 0: shared-prefix layout (common.h, seq_row): rows 0 .. Ps-1 are positions 0 .. Ps-1 once (class 0's tokens), then T - Ps rows per class
    if (row >= Ps + C * (T - Ps)) return;
    const int c = row < Ps ? 0 : (row - Ps) / (T - Ps), t = row < Ps ? row : Ps + (row - Ps) - c * (T - Ps);
    const int d4 = d >> 2;
    const f32x4* src;
    if (t >= 1 && t <= P) {
        src = (const f32x4*)(prefix + ((size_t)(prefix_classes == 1 ? 0 : c) * P + (t - 1)) * d);
    } else {
        int id = ids[(size_t)c * ld_ids + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        src = (const f32x4*)(tok_emb + (size_t)id * d);
    }
    const f32x4* pp = pos ? (const f32x4*)(pos + (size_t)t * d) : nullptr;
    RT* o = x + (size_t)row * d;
    float sm = 0.f, sq = 0.f;
    for (int f = lane; f < d4; f += 64) {
        const f32x4 v = pp ? src[f] + pp[f] : src[f];
        store4(o, f, v);
        sm += (v[0] + v[1]) + (v[2] + v[3]);
        sq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (rowstat) {     // (mean, rstd) of the row, for the LayerNorm folded into the first QKV GEMM
        const float mean = wave_sum(sm) / (float)d;
        const float var = fmaxf(wave_sum(sq) / (float)d - mean * mean, 0.f);
        if (lane == 0) ((float2*)rowstat)[row] = make_float2(mean, rsqrtf(var + LN_EPS));
    }
}

int launch_text_embed(const int32_t* token_ids, int ld_ids, const float* tok_emb, const float* pos, const float* prefix, int P,
                      int prefix_classes, void* x, int f32, float* rowstat, int C, int T, int d, int vocab, hipStream_t s, int shared_rows) {
    GRIP_REQUIRE(d % 4 == 0, "text_embed: width %% 4 != 0");
    GRIP_REQUIRE(shared_rows == 0 || (shared_rows == P + 1 && prefix_classes == 1 && shared_rows < T), "text_embed: shared-prefix layout needs one shared context and shared rows = n_prefix + 1 < T");
    const int rows = shared_rows + C * (T - shared_rows);
    if (f32)
        hipLaunchKernelGGL(text_embed_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, s, token_ids, ld_ids, tok_emb, pos, prefix, P, prefix_classes, (float*)x, rowstat, C, T, d, vocab, shared_rows);
    else
        hipLaunchKernelGGL(text_embed_kernel<resid_t>, dim3((rows + 3) / 4), dim3(256), 0, s, token_ids, ld_ids, tok_emb, pos, prefix, P, prefix_classes, (resid_t*)x, rowstat, C, T, d, vocab, shared_rows);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Patch gather (the conv1 of models/clip_encoders.py:131 as a GEMM operand):
//   out[(b*G + py)*G + px][c*p*p + kh*p + kw] = img[b][c][py*p + kh][px*p + kw], zero-padded to Kpad.
// One thread writes 8 consecutive k (16 bytes).
template <typename OT>
__device__ __forceinline__ void store8(OT* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<half_t>(half_t* p, const float (&v)[8]) {
    *(half8*)p = (half8){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
}
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    ((f32x4*)p)[0] = (f32x4){v[0], v[1], v[2], v[3]};
    ((f32x4*)p)[1] = (f32x4){v[4], v[5], v[6], v[7]};
}

template <typename T, typename OT>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ img, OT* __restrict__ out, int B, int R, int p, int Kpad) {
    const int G = R / p;
    const int K = 3 * p * p;
    const int chunks = Kpad >> 3;
    const size_t total = (size_t)B * G * G * chunks;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int chunk = (int)(idx % chunks);
        const size_t prow = idx / chunks;
        const int px = (int)(prow % G);
        const int py = (int)((prow / G) % G);
        const int b = (int)(prow / ((size_t)G * G));
        const int k0 = chunk * 8;
        float h[8];
        if ((p & 7) == 0 && k0 + 8 <= K) {
            const int c = k0 / (p * p), rem = k0 - c * p * p, kh = rem / p, kw = rem - kh * p;
            const T* src = img + (((size_t)b * 3 + c) * R + (py * p + kh)) * R + px * p + kw;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (float)src[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                float v = 0.f;
                if (k < K) {
                    const int c = k / (p * p), rem = k - c * p * p, kh = rem / p, kw = rem - kh * p;
                    v = (float)img[(((size_t)b * 3 + c) * R + (py * p + kh)) * R + px * p + kw];
                }
                h[j] = v;
            }
        }
        store8<OT>(out + prow * Kpad + k0, h);
    }
}

int launch_im2col(const void* images, int images_f16, void* out, int out_f32, int B, int R, int patch, int Kpad, hipStream_t s) {
    GRIP_REQUIRE(R % patch == 0 && Kpad % 8 == 0, "im2col: bad geometry R=%d patch=%d", R, patch);
    const int G = R / patch;
    const size_t total = (size_t)B * G * G * (Kpad / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (out_f32) {
        if (images_f16)
            hipLaunchKernelGGL((im2col_kernel<half_t, float>), dim3(blocks), dim3(256), 0, s, (const half_t*)images, (float*)out, B, R, patch, Kpad);
        else
            hipLaunchKernelGGL((im2col_kernel<float, float>), dim3(blocks), dim3(256), 0, s, (const float*)images, (float*)out, B, R, patch, Kpad);
    } else if (images_f16)
        hipLaunchKernelGGL((im2col_kernel<half_t, half_t>), dim3(blocks), dim3(256), 0, s, (const half_t*)images, (half_t*)out, B, R, patch, Kpad);
    else
        hipLaunchKernelGGL((im2col_kernel<float, half_t>), dim3(blocks), dim3(256), 0, s, (const float*)images, (half_t*)out, B, R, patch, Kpad);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// out[c][r] = in[r][c] (f16), 64x64 tiles through LDS; used once per weight in grip_tower_finalize.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, int cols, int ld_in) {
    __shared__ T tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : (T)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][i];
    }
}

int launch_transpose(const void* in, void* out, int f32, int rows, int cols, int ld_in, hipStream_t s) {
    if (f32)
        hipLaunchKernelGGL(transpose_kernel<float>, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, s, (const float*)in, (float*)out, rows, cols, ld_in);
    else
        hipLaunchKernelGGL(transpose_kernel<half_t>, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, s, (const half_t*)in, (half_t*)out, rows, cols, ld_in);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm folded into its consumer GEMM (EPI_LNFOLD_*, gemm.hip).
// (1) Row statistics: the residual GEMM epilogues emit, per row and 64-column wave tile, (sum, sum of squares) of the values they
//     store, laid out [d/64][M] (a store instruction of the epilogue writes the pairs of 8 consecutive rows = 64 contiguous bytes; the
//     row-major layout scattered them over 8 lines); one thread per row adds the d/64 pairs in a fixed order and turns them into (mean, rstd).  96 B in, 8 B out per row
//     at d = 768 -- against 2 x 1 536 B for a stand-alone LayerNorm pass over the stream.
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, int parts, float* __restrict__ rowstat, int M, float inv_d) {
#pragma clang fp contract(off)      // (gemm.hip's row_stat computes the same pair inside the consuming GEMM: identical roundings)
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const float2* p = (const float2*)part + row;         // [parts][M]: consecutive threads read consecutive pairs
    float sm = 0.f, sq = 0.f;
    for (int i = 0; i < parts; ++i) {
        const float2 v = p[(size_t)i * M];
        sm += v.x;
        sq += v.y;
    }
    const float mean = sm * inv_d;
    const float var = fmaxf(sq * inv_d - mean * mean, 0.f);
    ((float2*)rowstat)[row] = make_float2(mean, rsqrtf(var + LN_EPS));
}

int launch_ln_stats_finalize(const float* stat_part, int parts, float* rowstat, int M, int d, hipStream_t s) {
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, stat_part, parts, rowstat, M, 1.0f / (float)d);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// (2) Weights, once per tower (grip_tower_finalize): W'[n][k] = f16(gamma[k] * W[n][k]), colsum[n] = sum_k W'[n][k] (of the ROUNDED
//     values, so that mean * colsum cancels the mean part of x W'^T exactly as accumulated), bias_out[n] = bias[n] + sum_k beta[k] W[n][k].
//     One wave per output row.
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const half_t* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ bias, half_t* __restrict__ Wg, float* __restrict__ colsum,
                                                              float* __restrict__ bias_out, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float cs = 0.f, bb = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = (float)W[(size_t)n * K + k];
        const half_t wg = (half_t)(gamma[k] * w);
        Wg[(size_t)n * K + k] = wg;
        cs += (float)wg;
        bb += beta[k] * w;
    }
    cs = wave_sum(cs);
    bb = wave_sum(bb);
    if (lane == 0) { colsum[n] = cs; bias_out[n] = bias[n] + bb; }
}

int launch_ln_fold_weights(const half_t* W, const float* gamma, const float* beta, const float* bias, half_t* Wg, float* colsum, float* bias_out,
                           int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, gamma, beta, bias, Wg, colsum, bias_out, N, K);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// out[b] = x[b * row_stride + (row_index ? row_index[b] : 0)]  (f16 rows of width d): the CLS / EOT rows of the stream, compacted.
__global__ __launch_bounds__(256) void gather_rows_kernel(const half_t* __restrict__ x, const int32_t* __restrict__ row_index, int row_stride,
                                                          half_t* __restrict__ out, int n_rows, int d8) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_rows * d8) return;
    const int b = idx / d8, c = idx - b * d8;
    const size_t src = (size_t)b * row_stride + (row_index ? row_index[b] : 0);
    ((half8*)out)[(size_t)b * d8 + c] = ((const half8*)x)[src * d8 + c];
}

// the same for rows of 4-byte elements (f32 values, or the split-f16 layout, whose row pitch is an f32 row's)
__global__ __launch_bounds__(256) void gather_rows4_kernel(const f32x4* __restrict__ x, const int32_t* __restrict__ row_index, int row_stride,
                                                           f32x4* __restrict__ out, int n_rows, int d4) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_rows * d4) return;
    const int b = idx / d4, c = idx - b * d4;
    const size_t src = (size_t)b * row_stride + (row_index ? row_index[b] : 0);
    out[(size_t)b * d4 + c] = x[src * d4 + c];
}
int launch_gather_rows4(const void* x, const int32_t* row_index, int row_stride, void* out, int n_rows, int d, hipStream_t s) {
    GRIP_REQUIRE(d % 4 == 0, "gather_rows4: width %% 4 != 0");
    hipLaunchKernelGGL(gather_rows4_kernel, dim3((n_rows * (d / 4) + 255) / 256), dim3(256), 0, s, (const f32x4*)x, row_index, row_stride, (f32x4*)out, n_rows, d / 4);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

int launch_gather_rows(const half_t* x, const int32_t* row_index, int row_stride, half_t* out, int n_rows, int d, hipStream_t s) {
    GRIP_REQUIRE(d % 8 == 0, "gather_rows: width %% 8 != 0");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_rows * (d / 8) + 255) / 256), dim3(256), 0, s, x, row_index, row_stride, out, n_rows, d / 8);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
