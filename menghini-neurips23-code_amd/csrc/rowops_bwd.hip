// Row-wise kernels of the backward (input-gradient) chain: LayerNorm backward fused with the
// residual-gradient add and the f16 re-quantisation for the next GEMM, the CLS/EOT scatter through
// ln_post / ln_final, the prompt-slice reductions, and the dynamic loss scale.
// Same one-wave-per-row, float4-per-lane structure as rowops.hip.
#include <math.h>

#include "common.h"

#define LN_EPS 1e-5f

__device__ __forceinline__ float wave_sum_b(float v) { return wave_sum(v); }

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd
__device__ __forceinline__ f32x4 load4b(const float* p, int i) { return ((const f32x4*)p)[i]; }
__device__ __forceinline__ f32x4 load4b(const half_t* p, int i) {
    const half4 h = ((const half4*)p)[i];
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// dx += the NU partial rows at dp, dp + stride, ...: all loads first, then the adds in index order (straight-line code per NU)
template <int NU, int NV>
__device__ __forceinline__ void add_parts(const f32x4* __restrict__ dp, size_t stride4, int lane, int d4, f32x4 (&dx)[NV]) {
    f32x4 t[NU][NV];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < d4) t[u][i] = dp[(size_t)u * stride4 + lane + 64 * i];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < d4) dx[i] += t[u][i];
}

// Every global load of the row (x, the dy partials, gamma) is issued before the first reduction: the four wave reductions then run on registers
// and the row costs ONE memory round trip (r04: with the dy loads behind the statistics the CoOp step's 24 ln_bwd_add launches took 8.7 us each).
template <int NV, typename XT>
__device__ __forceinline__ void ln_bwd_row(const XT* __restrict__ xr, const f32x4* __restrict__ dyr, const f32x4* __restrict__ gamma,
                                           int lane, int d4, int d, f32x4 (&dx)[NV], int parts = 1, size_t part_stride4 = 0) {
    f32x4 x[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            x[i] = load4b(xr, lane + 64 * i);
            dx[i] = dyr[lane + 64 * i];
            gm[i] = gamma[lane + 64 * i];
        }
    // split-K partials, added in index order, up to FOUR in flight at a time (the text tower's dgrad GEMMs split eight ways: as a one-load-per-trip loop
    // the 24 ln_bwd_add launches of a CoOp step spent 8.4 us each waiting for seven dependent trips).  The guards are wave-uniform branches: a slot past
    // the last partial loads nothing (re-reading the last partial instead cost the image tower's bandwidth-bound launches 2 us: 3 partials -> 5 reads).
    for (int p = 1; p < parts; p += 4) {
        const f32x4* dp = dyr + (size_t)p * part_stride4;
        switch (parts - p < 4 ? parts - p : 4) {
            case 1: add_parts<1, NV>(dp, part_stride4, lane, d4, dx); break;
            case 2: add_parts<2, NV>(dp, part_stride4, lane, d4, dx); break;
            case 3: add_parts<3, NV>(dp, part_stride4, lane, d4, dx); break;
            default: add_parts<4, NV>(dp, part_stride4, lane, d4, dx); break;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
    const float mean = wave_sum_b(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) { x[i] = x[i] - mean; q += x[i][0] * x[i][0] + x[i][1] * x[i][1] + x[i][2] * x[i][2] + x[i][3] * x[i][3]; }
    const float rstd = rsqrtf(wave_sum_b(q) / (float)d + LN_EPS);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            x[i] = x[i] * rstd;                                    // xhat
            dx[i] = dx[i] * gm[i];                                 // g
            a += dx[i][0] + dx[i][1] + dx[i][2] + dx[i][3];
            b += dx[i][0] * x[i][0] + dx[i][1] * x[i][1] + dx[i][2] * x[i][2] + dx[i][3] * x[i][3];
        }
    a = wave_sum_b(a) / (float)d;
    b = wave_sum_b(b) / (float)d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) dx[i] = (dx[i] - a - x[i] * b) * rstd;
}

// The same row with the loads where they are used (the form until r04): fewer registers in flight, more waves per CU -- what the bandwidth-bound
// launches want (image tower, M = 3 408 rows x 768, three partials: 63 MB per launch; 11.5 us this way, 13.0 with everything in flight).  Same
// arithmetic in the same order: the two forms return the same bits.
template <int NV, typename XT>
__device__ __forceinline__ void ln_bwd_row_stream(const XT* __restrict__ xr, const f32x4* __restrict__ dyr, const f32x4* __restrict__ gamma,
                                                  int lane, int d4, int d, f32x4 (&dx)[NV], int parts, size_t part_stride4) {
    f32x4 x[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) { x[i] = load4b(xr, lane + 64 * i); s += x[i][0] + x[i][1] + x[i][2] + x[i][3]; }
    const float mean = wave_sum_b(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) { x[i] = x[i] - mean; q += x[i][0] * x[i][0] + x[i][1] * x[i][1] + x[i][2] * x[i][2] + x[i][3] * x[i][3]; }
    const float rstd = rsqrtf(wave_sum_b(q) / (float)d + LN_EPS);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            x[i] = x[i] * rstd;                                    // xhat
            f32x4 dy = dyr[lane + 64 * i];
            for (int p = 1; p < parts; ++p) dy += dyr[p * part_stride4 + lane + 64 * i];   // split-K partials, fixed order
            dx[i] = dy * gamma[lane + 64 * i];                     // g
            a += dx[i][0] + dx[i][1] + dx[i][2] + dx[i][3];
            b += dx[i][0] * x[i][0] + dx[i][1] * x[i][1] + dx[i][2] * x[i][2] + dx[i][3] * x[i][3];
        }
    a = wave_sum_b(a) / (float)d;
    b = wave_sum_b(b) / (float)d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) dx[i] = (dx[i] - a - x[i] * b) * rstd;
}

// dx[r] += LNbwd(sum_p dln[p][r]; x[r]);  dxh[r] = f16(dx[r])            (r < M; p over the split-K partials of the producer)
// EARLY: every load of the row in flight before the first reduction (few rows: latency-bound); otherwise the streaming form above.
template <int NV, bool EARLY>
__global__ __launch_bounds__(256) void ln_bwd_add_kernel(const resid_t* __restrict__ x, const float* __restrict__ dln, const float* __restrict__ gamma,
                                                         float* __restrict__ dx, half_t* __restrict__ dxh, int M, int d, int parts, size_t part_stride4) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int d4 = d >> 2;
    f32x4 g[NV], old[NV];
    f32x4* o = (f32x4*)(dx + (size_t)row * d);
    half4* oh = (half4*)(dxh + (size_t)row * d);
    if constexpr (EARLY) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < d4) old[i] = o[lane + 64 * i];         // in flight with the row's other loads
        ln_bwd_row<NV>(x + (size_t)row * d, (const f32x4*)(dln + (size_t)row * d), (const f32x4*)gamma, lane, d4, d, g, parts, part_stride4);
    } else {
        ln_bwd_row_stream<NV>(x + (size_t)row * d, (const f32x4*)(dln + (size_t)row * d), (const f32x4*)gamma, lane, d4, d, g, parts, part_stride4);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < d4) old[i] = o[lane + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            const f32x4 v = old[i] + g[i];
            o[lane + 64 * i] = v;
            oh[lane + 64 * i] = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
}

// dx[r] = LNbwd(sum_p dln[p][r]; x[r]) + (r is the read row of its sequence ? rows_add[sequence] : 0);  dxh[r] = f16(dx[r]).
// The ln_1 backward of the LAST block when that block ran for the read rows only (csrc/tower.hip): the stream gradient entering the
// block is zero except at row b * stride + index[b] of sequence b, where it is rows_add[b] -- no zero fill, no scatter.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_init_kernel(const resid_t* __restrict__ x, const float* __restrict__ dln, const float* __restrict__ gamma,
                                                          const float* __restrict__ rows_add, const int32_t* __restrict__ index, int stride,
                                                          float* __restrict__ dx, half_t* __restrict__ dxh, int M, int d, int parts, size_t part_stride4) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int d4 = d >> 2;
    f32x4 g[NV];
    ln_bwd_row<NV>(x + (size_t)row * d, (const f32x4*)(dln + (size_t)row * d), (const f32x4*)gamma, lane, d4, d, g, parts, part_stride4);
    const int b = row / stride;
    const bool read = row - b * stride == (index ? index[b] : 0);
    const f32x4* add = (const f32x4*)(rows_add + (size_t)b * d);
    f32x4* o = (f32x4*)(dx + (size_t)row * d);
    half4* oh = (half4*)(dxh + (size_t)row * d);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            f32x4 v = g[i];
            if (read) v += add[lane + 64 * i];
            o[lane + 64 * i] = v;
            oh[lane + 64 * i] = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
}

// Final LayerNorm (ln_post on the CLS row / ln_final on the EOT row): dx[row_b] = LNbwd(dy[b]; x[row_b]),
// row_b = b * stride + (index ? index[b] : 0).  dx / dxh were zero-filled before.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_scatter_kernel(const resid_t* __restrict__ x, const float* __restrict__ dy, const int32_t* __restrict__ index,
                                                             int stride, const float* __restrict__ gamma, float* __restrict__ dx,
                                                             half_t* __restrict__ dxh, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int d4 = d >> 2;
    const size_t row = (size_t)b * stride + (index ? index[b] : 0);
    f32x4 g[NV];
    ln_bwd_row<NV>(x + row * d, (const f32x4*)(dy + (size_t)b * d), (const f32x4*)gamma, lane, d4, d, g);
    f32x4* o = (f32x4*)(dx + row * d);
    half4* oh = (half4*)(dxh + row * d);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            o[lane + 64 * i] = g[i];
            oh[lane + 64 * i] = (half4){(half_t)g[i][0], (half_t)g[i][1], (half_t)g[i][2], (half_t)g[i][3]};
        }
}

// The same with the zero fill folded in (r04: two memset nodes and this kernel were three launches of a prompt step): every row r < M of dx / dxh is
// written -- LNbwd(dy[b]; x[r]) where r is the read row of its sequence b (r = b * stride + index[b], sequences start at row `first`: 0, or Ps in the
// shared-prefix layout whose leading rows belong to no sequence's read position), zero elsewhere.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_scatter_fill_kernel(const resid_t* __restrict__ x, const float* __restrict__ dy, const int32_t* __restrict__ index,
                                                                  int stride, int first, const float* __restrict__ gamma, float* __restrict__ dx,
                                                                  half_t* __restrict__ dxh, int n, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int d4 = d >> 2;
    const int b = row >= first ? (row - first) / stride : -1;
    const bool read = b >= 0 && b < n && row == b * stride + (index ? index[b] : 0);
    f32x4 g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (read) ln_bwd_row<NV>(x + (size_t)row * d, (const f32x4*)(dy + (size_t)b * d), (const f32x4*)gamma, lane, d4, d, g);
    f32x4* o = (f32x4*)(dx + (size_t)row * d);
    half4* oh = (half4*)(dxh + (size_t)row * d);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            o[lane + 64 * i] = g[i];
            oh[lane + 64 * i] = (half4){(half_t)g[i][0], (half_t)g[i][1], (half_t)g[i][2], (half_t)g[i][3]};
        }
}

// Visual prompt slice through ln_pre: grad_prefix[s] = inv_scale * sum_b LNbwd(dx[b*S + 1 + s]; prefix[s]).
// One workgroup of 8 waves per prompt token: wave w sums the images b = w (mod 8) in order, the eight partial rows meet in LDS and wave 0
// adds them in wave order (deterministic).  (Until r03 one wave walked the whole batch: 16 dependent LayerNorm-backward rows = 32 us.)
template <int NV>
__global__ __launch_bounds__(512) void vit_prefix_grad_kernel(const float* __restrict__ dx, const float* __restrict__ prefix, const float* __restrict__ gamma,
                                                              const float* __restrict__ scale, float* __restrict__ grad, int B, int S, int P, int d) {
    __shared__ f32x4 part[7][64 * NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x;
    const int d4 = d >> 2;
    f32x4 acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int b = wave; b < B; b += 8) {
        f32x4 g[NV];
        ln_bwd_row<NV>(prefix + (size_t)s * d, (const f32x4*)(dx + ((size_t)b * S + 1 + s) * d), (const f32x4*)gamma, lane, d4, d, g);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < d4) acc[i] += g[i];
    }
    if (wave != 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < d4) part[wave - 1][lane + 64 * i] = acc[i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 7; ++w)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < d4) acc[i] += part[w][lane + 64 * i];
    const float inv = scale[1];
    f32x4* o = (f32x4*)(grad + (size_t)s * d);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) o[lane + 64 * i] = acc[i] * inv;
}

// Textual prompt slice: grad_prefix[pc][p] = inv_scale * sum_{c in group} dx[c*T + 1 + p]
// One wave per (prompt token, 256-float slice of the row): a shared prompt sums C rows in class order, sixteen loads in flight.
__global__ __launch_bounds__(256) void text_prefix_grad_kernel(const float* __restrict__ dx, const float* __restrict__ scale, float* __restrict__ grad,
                                                               int C, int T, int P, int prefix_classes, int d) {
    const int lane = threadIdx.x & 63;
    const int d4 = d >> 2;
    const int nf = (d4 + 63) >> 6;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int idx = w / nf, f = (w - idx * nf) * 64 + lane;
    if (idx >= prefix_classes * P || f >= d4) return;
    const int pc = idx / P, p = idx - pc * P;
    const float inv = scale[1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (prefix_classes == 1) {
        // added in class order (the sum is the same sequence of f32 adds as a plain loop)
        int c = 0;
        for (; c + 16 <= C; c += 16) {
            f32x4 r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = ((const f32x4*)(dx + ((size_t)(c + u) * T + 1 + p) * d))[f];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += r[u];
        }
        for (; c < C; ++c) acc += ((const f32x4*)(dx + ((size_t)c * T + 1 + p) * d))[f];
    } else {
        acc = ((const f32x4*)(dx + ((size_t)pc * T + 1 + p) * d))[f];
    }
    ((f32x4*)(grad + (size_t)idx * d))[f] = acc * inv;
}

// Dynamic loss scale: scale[0] = 2^k with amax(g) * 2^k in [32, 64), scale[1] = 2^-k; g16 = f16(g * scale).
// Keeps the f16 gradient operands of the dgrad GEMMs away from the subnormal range; the chain is
// linear in g, so the power-of-two scale is exact and is divided out at the prompt slice.
// Up to 64 Ki elements (every prompt step: classes x embed_dim or batch x embed_dim) stay in registers between the two passes, all 16 loads of a
// thread in flight at once (r04; as two loops over memory the single workgroup spent 9.3 us on 13 dependent trips).
__global__ __launch_bounds__(1024) void grad_scale_cast_kernel(const float* __restrict__ g, half_t* __restrict__ g16, float* __restrict__ scale, int n) {
    __shared__ float red[16];
    __shared__ float sc;
    constexpr int RC = 16;
    float m = 0.f;
    const int n4 = (n & 3) == 0 ? n >> 2 : 0;        // 16-byte path when the length allows (embedding blocks always do)
    const bool cached = n4 > 0 && n4 <= 1024 * RC;
    f32x4 v[RC];
    if (cached) {
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            const int i = threadIdx.x + u * 1024;
            v[u] = i < n4 ? ((const f32x4*)g)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < RC; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
    } else {
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const f32x4 w = ((const f32x4*)g)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
        }
        for (int i = n4 * 4 + threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(g[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int i = 0; i < 16; ++i) a = fmaxf(a, red[i]);
        int e = 0;
        if (a > 0.f && isfinite(a)) {
            frexpf(a, &e);         // a = f * 2^e, f in [0.5, 1)
            e = 6 - e;             // a * 2^e in [32, 64)
            e = e > 40 ? 40 : (e < -40 ? -40 : e);
        }
        sc = ldexpf(1.0f, e);
        scale[0] = sc;
        scale[1] = ldexpf(1.0f, -e);
    }
    __syncthreads();
    const float s = sc;
    if (cached) {
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            const int i = threadIdx.x + u * 1024;
            const f32x4 w = v[u] * s;
            if (i < n4) ((half4*)g16)[i] = (half4){(half_t)w[0], (half_t)w[1], (half_t)w[2], (half_t)w[3]};
        }
        return;
    }
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const f32x4 w = ((const f32x4*)g)[i] * s;
        ((half4*)g16)[i] = (half4){(half_t)w[0], (half_t)w[1], (half_t)w[2], (half_t)w[3]};
    }
    for (int i = n4 * 4 + threadIdx.x; i < n; i += 1024) g16[i] = (half_t)(g[i] * s);
}

#define DISPATCH_NV_B(d, CALL)                                                                 \
    do {                                                                                       \
        const int _nv = ((d) / 4 + 63) / 64;                                                   \
        GRIP_REQUIRE((d) % 4 == 0 && _nv >= 1 && _nv <= 8, "row kernel: unsupported width %d", (d)); \
        switch (_nv) {                                                                         \
            case 1: { constexpr int NV = 1; CALL; } break;                                     \
            case 2: { constexpr int NV = 2; CALL; } break;                                     \
            case 3: { constexpr int NV = 3; CALL; } break;                                     \
            case 4: { constexpr int NV = 4; CALL; } break;                                     \
            case 5: case 6: { constexpr int NV = 6; CALL; } break;                             \
            default: { constexpr int NV = 8; CALL; } break;                                    \
        }                                                                                      \
    } while (0)

int launch_ln_bwd_add(const resid_t* x, const float* dln, int parts, int64_t part_stride, const float* gamma, float* dx, half_t* dxh, int M, int d, hipStream_t s) {
    GRIP_REQUIRE(parts >= 1 && part_stride % 4 == 0, "ln_bwd_add: bad partial layout (parts=%d stride=%lld)", parts, (long long)part_stride);
    if (M <= 1024) {     // a few hundred rows (text tower): latency-bound, every load in flight at once; else the streaming form
        DISPATCH_NV_B(d, hipLaunchKernelGGL((ln_bwd_add_kernel<NV, true>), dim3((M + 3) / 4), dim3(256), 0, s, x, dln, gamma, dx, dxh, M, d, parts, (size_t)(part_stride / 4)));
    } else {
        DISPATCH_NV_B(d, hipLaunchKernelGGL((ln_bwd_add_kernel<NV, false>), dim3((M + 3) / 4), dim3(256), 0, s, x, dln, gamma, dx, dxh, M, d, parts, (size_t)(part_stride / 4)));
    }
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
int launch_ln_bwd_init(const resid_t* x, const float* dln, int parts, int64_t part_stride, const float* gamma, const float* rows_add, const int32_t* index, int stride,
                       float* dx, half_t* dxh, int M, int d, hipStream_t s) {
    GRIP_REQUIRE(parts >= 1 && part_stride % 4 == 0 && stride >= 1, "ln_bwd_init: bad layout (parts=%d stride=%lld rows per sequence=%d)", parts, (long long)part_stride, stride);
    DISPATCH_NV_B(d, hipLaunchKernelGGL(ln_bwd_init_kernel<NV>, dim3((M + 3) / 4), dim3(256), 0, s, x, dln, gamma, rows_add, index, stride, dx, dxh, M, d, parts,
                                        (size_t)(part_stride / 4)));
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
int launch_ln_bwd_scatter(const resid_t* x, const float* dy, const int32_t* index, int stride, const float* gamma, float* dx, half_t* dxh,
                          int n, int d, hipStream_t s) {
    DISPATCH_NV_B(d, hipLaunchKernelGGL(ln_bwd_scatter_kernel<NV>, dim3((n + 3) / 4), dim3(256), 0, s, x, dy, index, stride, gamma, dx, dxh, n, d));
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
int launch_ln_bwd_scatter_fill(const resid_t* x, const float* dy, const int32_t* index, int stride, int first, const float* gamma, float* dx, half_t* dxh,
                               int n, int M, int d, hipStream_t s) {
    GRIP_REQUIRE(stride >= 1 && first >= 0, "ln_bwd_scatter_fill: bad layout (stride=%d first=%d)", stride, first);
    DISPATCH_NV_B(d, hipLaunchKernelGGL(ln_bwd_scatter_fill_kernel<NV>, dim3((M + 3) / 4), dim3(256), 0, s, x, dy, index, stride, first, gamma, dx, dxh, n, M, d));
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
int launch_vit_prefix_grad(const float* dx, const float* prefix, const float* gamma, const float* scale, float* grad, int B, int S, int P, int d, hipStream_t s) {
    DISPATCH_NV_B(d, hipLaunchKernelGGL(vit_prefix_grad_kernel<NV>, dim3(P), dim3(512), 0, s, dx, prefix, gamma, scale, grad, B, S, P, d));
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
int launch_text_prefix_grad(const float* dx, const float* scale, float* grad, int C, int T, int P, int prefix_classes, int d, hipStream_t s) {
    const int waves = prefix_classes * P * ((d / 4 + 63) / 64);
    hipLaunchKernelGGL(text_prefix_grad_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, dx, scale, grad, C, T, P, prefix_classes, d);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
int launch_grad_scale_cast(const float* g, half_t* g16, float* scale, int n, hipStream_t s) {
    hipLaunchKernelGGL(grad_scale_cast_kernel, dim3(1), dim3(1024), 0, s, g, g16, scale, n);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
