// Exact-mode GEMM: C[M,N] = epi(A[M,K] * W[N,K]^T) with f32 operands, f32 accumulate, v_mfma_f32_16x16x4_f32 (bitwise an
// fmaf chain per output element; 157 TFLOP/s dense peak on gfx950 = 1/16 of the f16 rate).  This is the arithmetic of the
// comparison mode (dims.precision = 1): the reference decides pseudolabel membership with fp32 softmax values and a strict
// '<' (utils/clip_pseudolabels.py:38-41, :73-82), so "identical indices" is asserted on towers that compute in fp32 end to end.
//
// One design, one kernel: 128x128 block tile, 4 waves in a 2x2 grid (64x64 per wave = 4x4 MFMA fragments, 64 accumulator
// registers), K staged 32 floats (one 128-byte line per row) at a time in two 32 KiB LDS stages, two workgroups per CU.
// The byte layout of a stage is the one of gemm_f16_kernel (gemm.hip): tiles go HBM -> LDS with global_load_lds (16 B per
// lane, 8 rows x 128 B per wave instruction), the 16-byte chunk index is XOR-swizzled with (row & 7) on the SOURCE address
// and again on the ds_read_b128 address (conflict-free).  A lane's chunk (kk*4 + lane/16) holds four consecutive k of its
// row: MFMA step s of the chunk contracts k = 4*(kk*4 + g) + s over the four lane groups g, so one ds_read_b128 per
// fragment feeds four MFMAs and a stage costs each wave 16 reads for 128 MFMAs (4 096 matrix-pipe cycles): MFMA-bound.
// Operands are swapped (W fragment first) so a lane ends with four CONSECUTIVE output columns of one row and stores 16 bytes.
#include <math.h>

#include "common.h"

// s_setprio(1) over a K step's fragment reads + MFMAs (1) or its MFMAs only (2); 0 = off.  Two workgroups share a CU (two waves per SIMD);
// measured (r03, exact ViT-B/16 encode, same box, two runs each): off 3 254 / 3 241 img/s, 1: 3 324 / 3 322, 2: 3 206 / 3 212; results bit-identical.
#ifndef GRIP_F32_PRIO
#define GRIP_F32_PRIO 1
#endif

#define BKF 32                       // floats per LDS row
#define STAGE_F ((128 + 128) * BKF)  // floats per stage = 32 KiB

// QuickGELU x * sigmoid(1.702 x) with IEEE exp and division (the fast-math form of gemm.hip is 1 ulp + 1 ulp off).
__device__ __forceinline__ float quick_gelu_exact(float x) { return x / (1.0f + expf(-1.702f * x)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE_F];

    // XCD-aware, bijective tile remap (block b runs on XCD b % 8): every XCD owns a contiguous run of tiles, N-fastest
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // staging: wave w fills rows [w*32, +32) of the A tile and of the W tile, 8 rows per instruction
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const size_t K = (size_t)g.K;
    // (wave-uniform bases + one constant 32-bit byte offset per lane, as in gemm.hip: no 64-bit vector adds per DMA piece)
    const float* a_src = (const float*)g.A + (size_t)(m0 + wave * 32) * K;
    const float* w_src = (const float*)g.W + (size_t)(n0 + wave * 32) * K;
    const uint32_t lane_off = 4u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(schunk * 4));
    auto stage = [&](int buf, int kt) {
        float* abase = lds + buf * STAGE_F + wave * 32 * BKF;
        float* bbase = lds + buf * STAGE_F + 128 * BKF + wave * 32 * BKF;
        const float* as = a_src + (size_t)kt * BKF;
        const float* ws = w_src + (size_t)kt * BKF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(as + (size_t)i * 8 * K) + lane_off), (AS3 void*)(abase + i * 8 * BKF), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(ws + (size_t)i * 8 * K) + lane_off), (AS3 void*)(bbase + i * 8 * BKF), 16, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = (kk * 4 + fgrp) ^ (lane & 7);
        a_off[kk] = (wr * 64 + frow) * BKF + chunk * 4;
        b_off[kk] = 128 * BKF + (wc * 64 + frow) * BKF + chunk * 4;
    }

    f32x4 acc[4][4];
    constexpr bool HAS_BIAS = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RESID);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) b = *(const f32x4*)(g.bias + n0 + wc * 64 + j * 16 + fgrp * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = b;
    }

    const int nk = g.K / BKF;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        // The LDS-DMA of stage kt must have landed before the barrier certifies it to the other waves.  The wait is explicit: the
        // compiler does not order a global_load_lds against later LDS reads by itself (it emitted only lgkmcnt(0) here).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // stage kt visible to every wave, stage kt-1 fully read
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const float* st = lds + buf * STAGE_F;
        if (GRIP_F32_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const f32x4*)(st + a_off[kk] + i * 16 * BKF);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const f32x4*)(st + b_off[kk] + j * 16 * BKF);
            if (GRIP_F32_PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
            if (GRIP_F32_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        }
        if (GRIP_F32_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    }

    // epilogue: lane (frow, fgrp) holds columns col0 + j*16 + fgrp*4 .. +3 of row row0 + i*16 + frow
    const int row0 = m0 + wr * 64 + frow, col0 = n0 + wc * 64 + fgrp * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + i * 16;
        if (row >= g.M) continue;
        const size_t ro = (size_t)row * g.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t o = ro + col0 + j * 16;
            f32x4 v = acc[i][j];
            if constexpr (EPI == EPI_F32_SCALE) v = v * g.scalar;
            if constexpr (EPI == EPI_BIAS_RESID) v += *(const f32x4*)((const float*)g.resid + o);
            if constexpr (EPI == EPI_BIAS_GELU_F16) {
                if (g.out2) *(f32x4*)((float*)g.out2 + o) = v;
                v = (f32x4){quick_gelu_exact(v[0]), quick_gelu_exact(v[1]), quick_gelu_exact(v[2]), quick_gelu_exact(v[3])};
            }
            *(f32x4*)((float*)g.out + o) = v;
        }
    }
}

int launch_gemm_f32(int epi, const GemmArgs& a, hipStream_t s) {
    GRIP_REQUIRE(a.N % 128 == 0 && a.K % BKF == 0 && a.M > 0, "gemm_f32: need N %% 128 == 0 and K %% 32 == 0 (M=%d N=%d K=%d)", a.M, a.N, a.K);
    GRIP_REQUIRE(a.ldc % 4 == 0, "gemm_f32: ldc %% 4 != 0");
    const int tiles_m = (a.M + 127) / 128, tiles_n = a.N / 128;
    GRIP_REQUIRE(a.m_pad == 0 || a.m_pad >= (int64_t)tiles_m * 128, "gemm_f32: A must be allocated up to the 128-row tile (M=%d m_pad=%lld)", a.M, (long long)a.m_pad);
    dim3 grid(tiles_m * tiles_n), block(256);
#define GRIP_GEMM_CASE(E) \
    case E: hipLaunchKernelGGL((gemm_f32_kernel<E>), grid, block, 0, s, a, tiles_m, tiles_n); break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_F32_SCALE)
        default: GRIP_REQUIRE(false, "gemm_f32: epilogue %d is not part of the exact (inference) path", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
