// f16 MFMA GEMMs for gfx950: C[M,N] = epi(A[M,K] * W[N,K]^T), f32 accumulate, v_mfma_f32_16x16x32_f16.
//
// Five kernels share one design (this header) and one epilogue:
//   gemm_k64p_kernel<EPI>           large problems (the pool encode): 256x256x64 tile, 8 waves, two 64 KiB LDS stages fed
//                                   with whole-cache-line DMA, persistent over tiles    -- see its own header below
//   gemm_k64_kernel<EPI,8>          the same, one tile per workgroup (fewer than 512 tiles)
//   gemm_big_kernel<EPI,256,256,4>  the same tile on a 4-slot ring of 32-wide K tiles, two wave groups in anti-phase
//                                   (the round-1 default until r01-g; kept as variant 2 / GRIP_GEMM_BIG=2)
//   gemm_big_kernel<EPI,256,128,3>  ring with 4 waves and two workgroups per CU (used where 256x256 tile counts
//                                   quantise badly over the 256 CUs)
//   gemm_f16_kernel<EPI,WMF>        small M (training batches): 128x128x64 or 64x128x64 tile, 4 waves in a 2x2 grid,
//                                   two LDS stages, one barrier per K tile
// Common to all: A and W tiles go HBM -> LDS with direct global_load_lds (16 B per lane, 1 KiB per wave instruction,
// lane-linear LDS image); the 16-byte chunk index is XOR-swizzled on the SOURCE address and again on the
// ds_read_b128 address, which makes the fragment reads bank-conflict free (guide T2 / rule 21).  The MFMA is issued with
// swapped operands (W fragment first) so each lane ends up with four CONSECUTIVE output columns of one row; the
// accumulators are then transposed through a wave-private LDS slab so that bias / residual / C move as whole 128-byte
// lines (epilogue_rows).  Workgroup ids are remapped so every XCD (private 4 MiB L2) owns a contiguous run of tiles,
// N-fastest: the A row panel and the W panel stay L2-resident across the run.  The launcher picks the tile shape by
// (relative rate) x (fill of the last wave of workgroups).
#include <stdlib.h>


#include "common.h"

// Synchronisation of the cooperative split-K form (gemm_ringw_kernel): 2 = system-scope cache bits on the partial tiles (default), 0 = device-scope fences,
// 1 = plain accesses (developer A/B only: relies on workgroup placement)
#ifndef GRIP_COOP_MODE
#define GRIP_COOP_MODE 2
#endif


// Sub-step 1 of the persistent GEMM places its 12 fragment reads after MFMA pairs GRIP_RD0 .. GRIP_RD0 + 11 of the 16.  The compiler
// puts a second s_waitcnt lgkmcnt(0) before the fifth MFMA of the block; with the reads starting at pair 0 that wait also drains the
// two reads just issued.  Starting at pair 4 nothing newer is outstanding there: residual GEMM 1 038 -> 1 047 TF/s (0: 1 038, 2: 1 037).
#ifndef GRIP_RD0
#define GRIP_RD0 4
#endif
// s_setprio around the MFMA sub-steps of the persistent GEMM (r03).  Bit 0: sub-step 0, bit 1: sub-step 1, for every epilogue (developer
// A/B); 4 (default): sub-step 1 of the LayerNorm-folded instantiations only.  Measured on one box, f16 bench loop, two runs each (TF/s of
// QKV / c_fc / residual): off 1 097 / 997 / 1 045; sub-step 0: 1 031 / 949 / 1 034; sub-step 1: 1 119 / 1 006 / 1 031; both: 1 054 / 958 / 1 027.
// While a wave is in the sub-step that also issues the stage's DMA pieces and the next fragment reads, winning the issue arbitration
// against the SIMD's other wave shortens that burst for the wide-output GEMMs (+2.0 % QKV, +0.9 % c_fc); the narrow residual GEMMs lose
// 1.3 % with it (their leading tiles wait on memory either way: the other wave's MFMAs are what fills that wait), so they stay without.
#ifndef GRIP_SETPRIO
#define GRIP_SETPRIO 4
#endif
// Developer experiment (r03): s_setprio(1) over the K step of the small-M kernels (gemm_f16_kernel, gemm_ring_kernel): no effect on the
// prompt steps (CoOp 2.62 / VPT 3.46 / UPT 4.01 ms eager either way), off
#ifndef GRIP_SMALL_PRIO
#define GRIP_SMALL_PRIO 0
#endif
#ifndef GRIP_KROT
#define GRIP_KROT 1
#endif

#define BM 128
#define BN 128
#define BK 64
#define STAGE_HALFS ((BM + BN) * BK)  // 16384 halfs = 32 KiB

// QuickGELU x * sigmoid(1.702 x) and its derivative.  v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE
// division (ten instructions): the result is rounded to f16 right after, and at 3072 columns per token
// the activation is a third of the c_fc GEMM's time otherwise.
#define QGELU_C (-1.702f * 1.4426950408889634f)   // exp(-1.702 x) = exp2(QGELU_C x): one multiply in front of v_exp_f32
__device__ __forceinline__ float quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(QGELU_C * x)); }
// Four at a time with the three full-rate operations as vector expressions: they compile to v_pk_mul_f32 / v_pk_add_f32 (two values
// per instruction); written per scalar the transcendental builtins in the middle keep the compiler from packing them (c_fc's
// epilogue: 128 elements per lane and tile, 5 VALU issues each -> 3.5).
__device__ __forceinline__ f32x4 quick_gelu4(f32x4 x) {
    const f32x4 t = x * QGELU_C;
    f32x4 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    e = e + 1.0f;
    const f32x4 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1]), __builtin_amdgcn_rcpf(e[2]), __builtin_amdgcn_rcpf(e[3])};
    return x * r;
}
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(QGELU_C * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// (epilogue_rows: branch-free for workgroups whose rows are all < M; only the last M tile checks rows.)
// Wave-private LDS transposition of the accumulators so that global accesses of the epilogue are
// row-contiguous: a lane's fragment data (4 consecutive columns of 16 different rows per
// instruction = 64-byte pieces) is written to a 32-row x 64-column f32 slab and read back as
// 4 rows x 256 contiguous bytes per wave instruction; residual reads and C writes then move whole
// 128-byte lines (f16 output: 128 B per row, f32: 256 B).  One slab per wave, no workgroup barrier.
#define EPI_LDW 68   // slab row stride in floats: conflict-free ds_write_b128, <= 2-way ds_read_b128
#define EPI_SLAB_FLOATS (32 * EPI_LDW)

// (mean, rstd) of row r of A for the LayerNorm-folded epilogues: the finalised pair, or -- GemmArgs::stat_in -- the sum of the producer's stat_parts partial
// (sum, sum of squares) pairs in index order with ln_stats_finalize's arithmetic (train-mode forwards: no finalising launch between the GEMMs).
__device__ __forceinline__ float2 row_stat(const GemmArgs& g, int r) {
#pragma clang fp contract(off)      // the same roundings as ln_stats_finalize_kernel wherever this is inlined
    if (g.stat_parts <= 0) return ((const float2*)g.rowstat)[r];
    const float2* sp = (const float2*)g.stat_in + r;
    float sm = 0.f, sq = 0.f;
    // all pairs requested before the first add (up to 16 = width 1 024; slots past the last pair re-read it and add zero): as a one-load-per-trip loop in
    // an epilogue the image tower's folded QKV / c_fc GEMMs took 39 / 58 us instead of 18 / 26
    constexpr int MAXP = 16;
    float2 v[MAXP];
    const int np = g.stat_parts < MAXP ? g.stat_parts : MAXP;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) v[i] = sp[(size_t)(i < np ? i : np - 1) * g.M];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        sm += i < np ? v[i].x : 0.f;
        sq += i < np ? v[i].y : 0.f;
    }
    for (int i = MAXP; i < g.stat_parts; ++i) {       // (wider streams: the rest one by one)
        const float2 w = sp[(size_t)i * g.M];
        sm += w.x;
        sq += w.y;
    }
    const float inv_d = 1.0f / (float)g.K;
    const float mean = sm * inv_d;
    const float var = fmaxf(sq * inv_d - mean * mean, 0.f);
    return make_float2(mean, rsqrtf(var + 1e-5f));
}

// Accumulators start at the bias (epilogues with one) instead of zero: a lane's acc[i][j] holds columns col0 + j*16 +
// (lane>>4)*4 .. +3 of some row for every row fragment i, so four float4 loads at kernel entry replace one v_add per output
// element in the epilogue.
template <int EPI, int ROWFRAGS>
__device__ __forceinline__ void init_acc(const GemmArgs& g, f32x4 (&acc)[ROWFRAGS][4], int col0, int lane) {
    constexpr bool HAS_BIAS = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_RESID_STATS);
    if constexpr (HAS_BIAS) {
        f32x4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const f32x4*)(g.bias + col0 + j * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int i = 0; i < ROWFRAGS; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = b[j];
    } else {
#pragma unroll
        for (int i = 0; i < ROWFRAGS; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// PF = row fragments per pass: 2 -> 32-row slab with padded rows (EPI_LDW); 1 -> 16-row slab of exactly 4 KiB whose 16-byte
// chunks are XOR-swizzled with the row instead (the persistent kernel keeps its slabs beside the two 64 KiB stages:
// 8 x 4 KiB is all the LDS that is left).
template <int PF>
__device__ __forceinline__ int slab_off(int row, int chunk) {
    if constexpr (PF == 2) return row * EPI_LDW + chunk * 4;
    else return row * 64 + ((chunk ^ (row & 15)) * 4);
}

// PRE (persistent kernel, LayerNorm-folded epilogues): the (mean, rstd) pairs of the wave's 128 rows were loaded at the START of
// the tile -- lane l holds rows l and 64 + l in pre[0] / pre[1] -- and each row group fetches its pair with two ds_bpermute
// instead of a global load whose latency the first pass of the epilogue cannot hide (that exposure cost the folded QKV /
// c_fc GEMMs 8 % against their plain-bias forms).
// HILO (residual epilogues, GemmArgs::resid_lo): the stream is a COMPENSATED pair -- value = resid + resid_lo -- read as such and written back as
// hi = f16(v) -> out, lo = f16(v - hi) -> resid_lo (r06: the screen of the pseudolabel pass; the statistics are those of v either way).
template <int EPI, int ROWFRAGS, bool CHECK, int PF, bool PRE = false, bool HILO = false>
__device__ __forceinline__ void epilogue_rows_impl(const GemmArgs& g, f32x4 (&acc)[ROWFRAGS][4], float* slab, int row0, int col0, int lane,
                                                   const float2* pre = nullptr, const f32x4* cb = nullptr) {
    constexpr int NP = ROWFRAGS / PF;
    constexpr int RP = 16 * PF;        // rows per pass
    constexpr int NI = 4 * PF;         // row groups (4 rows each) per pass
    constexpr bool FOLD = (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16);
    constexpr bool RESID = (EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_RESID_STATS);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int rr = lane >> 4, cc = (lane & 15) * 4;
    const int col = col0 + cc;
    const int ldc = g.ldc;
    // operand prefetch (residual / GELU' argument / row statistics): all row loads of a pass are issued together, and
    // the loads of pass p+1 go out before the stores of pass p, so no load ever queues behind a store.
    half4 res[2][NI];
    half4 resl[2][HILO ? NI : 1];
    half4 aux[2][NI];
    float2 rst[2][NI];
    // Addresses: one 32-bit element offset per lane plus a wave-uniform step per row group, against the uniform base
    // pointers (saddr + voffset addressing, one VGPR per access).  With 64-bit per-row pointers the compiler materialises
    // all 64 of them at the top of the epilogue and spills them when the main loop leaves < 10 free registers.
    const uint32_t lane_off = (uint32_t)(row0 + rr) * (uint32_t)ldc + (uint32_t)col;      // M * ldc < 2^31 elements (launcher)
    auto elem_off = [&](int p, int it) -> uint32_t {
        if constexpr (CHECK) {
            int row = row0 + p * RP + it * 4 + rr;
            row = row < g.M ? row : g.M - 1;
            return (uint32_t)row * (uint32_t)ldc + (uint32_t)col;
        } else {
            return lane_off + (uint32_t)((p * RP + it * 4) * ldc);
        }
    };
    auto prefetch = [&](int p, int b) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const uint32_t o = elem_off(p, it);
            if constexpr (RESID) res[b][it] = *(const half4*)((const half_t*)g.resid + o);
            if constexpr (RESID && HILO) resl[b][it] = *(const half4*)(g.resid_lo + o);
            if constexpr (EPI == EPI_GELUGRAD_F16) aux[b][it] = *(const half4*)((const half_t*)g.aux + o);
            if constexpr (FOLD && !PRE) {
                int row = row0 + p * RP + it * 4 + rr;
                if constexpr (CHECK) row = row < g.M ? row : g.M - 1;
                rst[b][it] = row_stat(g, row);      // the 16 lanes of a row read one address: broadcast loads
            }
        }
    };
    f32x4 csum = {0.f, 0.f, 0.f, 0.f}, bfold = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FOLD) {
        if (cb) {       // loaded by the caller before its K loop (colsum / folded bias of THIS lane's four columns)
            csum = cb[0];
            bfold = cb[1];
        } else {
            csum = *(const f32x4*)(g.colsum + col);
            bfold = *(const f32x4*)(g.bias + col);
        }
    }
    if constexpr (RESID || EPI == EPI_GELUGRAD_F16 || (FOLD && !PRE)) prefetch(0, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int ii = 0; ii < PF; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(slab + slab_off<PF>(ii * 16 + frow, j * 4 + fgrp)) = acc[PF * p + ii][j];
        if constexpr (RESID || EPI == EPI_GELUGRAD_F16 || (FOLD && !PRE))
            if (p + 1 < NP) prefetch(p + 1, (p + 1) & 1);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int rl = it * 4 + rr;
            f32x4 v = *(const f32x4*)(slab + slab_off<PF>(rl, lane & 15));
            const int row = row0 + p * RP + rl;
            const uint32_t o = CHECK ? (uint32_t)row * (uint32_t)ldc + (uint32_t)col : elem_off(p, it);
            if constexpr (RESID) {
                // the add into the residual stream happens here in f32; the row statistics of what is about to be stored travel
                // with it (per 64-column wave tile), so the LayerNorm that follows never has to re-read the stream
                const half4 rh = res[p & 1][it];
                if constexpr (HILO) {       // hi + lo is exact in f32 (two 11-bit numbers, |lo| <= ulp(hi) / 2)
                    const half4 rl = resl[p & 1][it];
                    v += (f32x4){(float)rh[0] + (float)rl[0], (float)rh[1] + (float)rl[1], (float)rh[2] + (float)rl[2], (float)rh[3] + (float)rl[3]};
                } else {
                    v += (f32x4){(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                }
                if constexpr (EPI == EPI_BIAS_RESID_STATS) {
                    const float sm = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
                    const float sq = row16_sum(__builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], v[3] * v[3]))));
                    if ((lane & 15) == 0 && (!CHECK || row < g.M))
                        ((float2*)g.stat_part)[(uint32_t)(col0 >> 6) * (uint32_t)g.M + (uint32_t)row] = make_float2(sm, sq);   // [N / 64][M]: M * N / 64 < 2^31
                }
            }
            float2 stp = make_float2(0.f, 0.f);
            if constexpr (FOLD && PRE) {      // cross-lane fetch with every lane active (outside the row guard below)
                const int rloc = p * RP + it * 4 + rr;                     // row inside the wave's tile: lane rloc & 63 of pre[rloc >> 6] holds its pair
                const int src = (rloc & 63) << 2;                          // (bytes)
                const float2 pv = pre[rloc >> 6];
                stp.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.x)));
                stp.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.y)));
            }
            if (!CHECK || row < g.M) {
                if constexpr (EPI == EPI_F32) {
                    *(f32x4*)((float*)g.out + o) = v;
                } else if constexpr (EPI == EPI_F32_SCALE) {
                    *(f32x4*)((float*)g.out + o) = v * g.scalar;
                } else if constexpr (RESID) {
                    const half4 hi = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *(half4*)((half_t*)g.out + o) = hi;
                    if constexpr (HILO)
                        *(half4*)(g.resid_lo + o) = (half4){(half_t)(v[0] - (float)hi[0]), (half_t)(v[1] - (float)hi[1]), (half_t)(v[2] - (float)hi[2]), (half_t)(v[3] - (float)hi[3])};
                } else if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_F16) {
                    *(half4*)((half_t*)g.out + o) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                } else if constexpr (EPI == EPI_BIAS_GELU_F16) {
                    if (g.out2) *(half4*)((half_t*)g.out2 + o) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    const f32x4 gv = quick_gelu4(v);
                    *(half4*)((half_t*)g.out + o) = (half4){(half_t)gv[0], (half_t)gv[1], (half_t)gv[2], (half_t)gv[3]};
                } else if constexpr (FOLD) {
                    float2 st;
                    if constexpr (PRE) st = stp;
                    else st = rst[p & 1][it];
                    v = (v - csum * st.x) * st.y + bfold;
                    if constexpr (EPI == EPI_LNFOLD_GELU_F16) {
                        if (g.out2) *(half4*)((half_t*)g.out2 + o) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        v = quick_gelu4(v);
                    }
                    *(half4*)((half_t*)g.out + o) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                } else if constexpr (EPI == EPI_GELUGRAD_F16) {
                    const half4 x = aux[p & 1][it];
                    *(half4*)((half_t*)g.out + o) = (half4){(half_t)(v[0] * quick_gelu_grad((float)x[0])), (half_t)(v[1] * quick_gelu_grad((float)x[1])),
                                                            (half_t)(v[2] * quick_gelu_grad((float)x[2])), (half_t)(v[3] * quick_gelu_grad((float)x[3]))};
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI, int ROWFRAGS, int PF = 2, bool PRE = false>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& g, f32x4 (&acc)[ROWFRAGS][4], float* slab, int row0, int col0, int lane,
                                              const float2* pre = nullptr, const f32x4* cb = nullptr) {
    if constexpr (EPI == EPI_BIAS_RESID_STATS) {       // (the statistics-carrying form only: the inference path's; the train-mode residual kernels stay as they were)
        if (g.resid_lo) {       // (wave-uniform: a launch either carries a compensated stream or it does not)
            if (row0 + ROWFRAGS * 16 <= g.M)
                epilogue_rows_impl<EPI, ROWFRAGS, false, PF, PRE, true>(g, acc, slab, row0, col0, lane, pre, cb);
            else
                epilogue_rows_impl<EPI, ROWFRAGS, true, PF, PRE, true>(g, acc, slab, row0, col0, lane, pre, cb);
            return;
        }
    }
    if (row0 + ROWFRAGS * 16 <= g.M)
        epilogue_rows_impl<EPI, ROWFRAGS, false, PF, PRE>(g, acc, slab, row0, col0, lane, pre, cb);
    else
        epilogue_rows_impl<EPI, ROWFRAGS, true, PF, PRE>(g, acc, slab, row0, col0, lane, pre, cb);
}

// ---- 16-byte-store form of the 16-row-pass epilogue (persistent kernel; the three f16-output epilogues of the pool encode).
// A lane owns EIGHT consecutive columns of one row: 8 lanes cover the wave tile's 64 columns (one 128-byte line), one wave
// instruction stores 8 rows x 128 B -- half the store (and residual-load) instructions of the 8-byte form above, whose store tail
// is issue-bound (MI355X_MICROARCH.md: "8 x dwordx4 halves it").  Slab layout: the 16-byte chunk ch (4 floats) of row r sits at
// position ((ch >> 1) ^ (r & 7)) + 8 * (ch & 1): the even and the odd chunks of a row each fill one 128-byte half, so the fragment
// writes (8 consecutive lanes = 8 rows, same chunk) and both read-back instructions (8 consecutive lanes = one row's 8 even / odd
// chunks) touch all 32 banks once.  Row statistics: 3 DPP steps over the row's 8 lanes.
__device__ __forceinline__ float row8_sum(float v) {
#define GRIP_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    GRIP_DPP_ADD(0xB1);
    GRIP_DPP_ADD(0x4E);
    GRIP_DPP_ADD(0x141);
#undef GRIP_DPP_ADD
    return v;
}

template <int EPI, bool CHECK, bool PRE, bool HILO = false>
__device__ __forceinline__ void epilogue_rows8_impl(const GemmArgs& g, f32x4 (&acc)[8][4], float* slab, int row0, int col0, int lane, const float2* pre) {
    constexpr bool FOLD = (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16);
    constexpr bool RESID = (EPI == EPI_BIAS_RESID_STATS);
    static_assert(FOLD || RESID, "epilogue_rows8: f16-output epilogues of the pool encode only");
    const int frow = lane & 15, fgrp = lane >> 4;
    const int r8 = lane >> 3, c8 = lane & 7;
    const int col = col0 + c8 * 8;
    const int ldc = g.ldc;
    const uint32_t lane_off = (uint32_t)(row0 + r8) * (uint32_t)ldc + (uint32_t)col;
    auto elem_off = [&](int p, int it) -> uint32_t {
        if constexpr (CHECK) {
            int row = row0 + p * 16 + it * 8 + r8;
            row = row < g.M ? row : g.M - 1;
            return (uint32_t)row * (uint32_t)ldc + (uint32_t)col;
        } else {
            return lane_off + (uint32_t)((p * 16 + it * 8) * ldc);
        }
    };
    half8 res[2][2];
    half8 resl[2][HILO ? 2 : 1];
    auto prefetch = [&](int p, int b) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            res[b][it] = *(const half8*)((const half_t*)g.resid + elem_off(p, it));
            if constexpr (HILO) resl[b][it] = *(const half8*)(g.resid_lo + elem_off(p, it));
        }
    };
    f32x4 csum[2], bfold[2];
    if constexpr (FOLD) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            csum[h] = *(const f32x4*)(g.colsum + col + h * 4);
            bfold[h] = *(const f32x4*)(g.bias + col + h * 4);
        }
    }
    if constexpr (RESID) prefetch(0, 0);
    // write positions of this lane's four fragment chunks (ch = j * 4 + fgrp), read positions of its two column chunks
    const int wbase = frow * 64 + (fgrp & 1) * 32;
    const int wlow = (fgrp >> 1);
    const int rpos = (c8 ^ r8) * 4;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(slab + wbase + (((j * 2 + wlow) ^ (frow & 7)) * 4)) = acc[p][j];
        if constexpr (RESID)
            if (p + 1 < 8) prefetch(p + 1, (p + 1) & 1);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int rl = it * 8 + r8;
            f32x4 v[2];
            v[0] = *(const f32x4*)(slab + rl * 64 + rpos);
            v[1] = *(const f32x4*)(slab + rl * 64 + 32 + rpos);
            const int row = row0 + p * 16 + rl;
            const uint32_t o = elem_off(p, it);
            if constexpr (RESID) {
                const half8 rh = res[p & 1][it];
                if constexpr (HILO) {
                    const half8 rl = resl[p & 1][it];
                    v[0] += (f32x4){(float)rh[0] + (float)rl[0], (float)rh[1] + (float)rl[1], (float)rh[2] + (float)rl[2], (float)rh[3] + (float)rl[3]};
                    v[1] += (f32x4){(float)rh[4] + (float)rl[4], (float)rh[5] + (float)rl[5], (float)rh[6] + (float)rl[6], (float)rh[7] + (float)rl[7]};
                } else {
                    v[0] += (f32x4){(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                    v[1] += (f32x4){(float)rh[4], (float)rh[5], (float)rh[6], (float)rh[7]};
                }
                const float sm = row8_sum(((v[0][0] + v[0][1]) + (v[0][2] + v[0][3])) + ((v[1][0] + v[1][1]) + (v[1][2] + v[1][3])));
                // (two 4-column chains added, then the tree: the same association as the 8-byte form's 16-lane reduction, so the statistics
                // -- and with them every later value of the row -- do not depend on which kernel family serves a launch)
                const float sqa = __builtin_fmaf(v[0][0], v[0][0], __builtin_fmaf(v[0][1], v[0][1], __builtin_fmaf(v[0][2], v[0][2], v[0][3] * v[0][3])));
                const float sqb = __builtin_fmaf(v[1][0], v[1][0], __builtin_fmaf(v[1][1], v[1][1], __builtin_fmaf(v[1][2], v[1][2], v[1][3] * v[1][3])));
                const float sq = row8_sum(sqa + sqb);
                if (c8 == 0 && (!CHECK || row < g.M))
                    ((float2*)g.stat_part)[(uint32_t)(col0 >> 6) * (uint32_t)g.M + (uint32_t)row] = make_float2(sm, sq);       // 8 rows = 64 contiguous bytes per store
            }
            float2 st = make_float2(0.f, 0.f);
            if constexpr (FOLD) {
                if constexpr (PRE) {
                    const int src = (((p & 3) * 16 + rl) << 2);
                    const float2 pv = pre[p >> 2];
                    st.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.x)));
                    st.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.y)));
                } else {
                    int rr_ = row;
                    if constexpr (CHECK) rr_ = rr_ < g.M ? rr_ : g.M - 1;
                    st = row_stat(g, rr_);
                }
            }
            if (!CHECK || row < g.M) {
                if constexpr (FOLD) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[h] = (v[h] - csum[h] * st.x) * st.y + bfold[h];
                    if constexpr (EPI == EPI_LNFOLD_GELU_F16) {
                        if (g.out2)
                            *(half8*)((half_t*)g.out2 + o) = (half8){(half_t)v[0][0], (half_t)v[0][1], (half_t)v[0][2], (half_t)v[0][3],
                                                                    (half_t)v[1][0], (half_t)v[1][1], (half_t)v[1][2], (half_t)v[1][3]};
#pragma unroll
                        for (int h = 0; h < 2; ++h) v[h] = quick_gelu4(v[h]);
                    }
                }
                const half8 hi = (half8){(half_t)v[0][0], (half_t)v[0][1], (half_t)v[0][2], (half_t)v[0][3],
                                         (half_t)v[1][0], (half_t)v[1][1], (half_t)v[1][2], (half_t)v[1][3]};
                *(half8*)((half_t*)g.out + o) = hi;
                if constexpr (HILO)
                    *(half8*)(g.resid_lo + o) = (half8){(half_t)(v[0][0] - (float)hi[0]), (half_t)(v[0][1] - (float)hi[1]), (half_t)(v[0][2] - (float)hi[2]), (half_t)(v[0][3] - (float)hi[3]),
                                                        (half_t)(v[1][0] - (float)hi[4]), (half_t)(v[1][1] - (float)hi[5]), (half_t)(v[1][2] - (float)hi[6]), (half_t)(v[1][3] - (float)hi[7])};
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI, bool PRE>
__device__ __forceinline__ void epilogue_rows8(const GemmArgs& g, f32x4 (&acc)[8][4], float* slab, int row0, int col0, int lane, const float2* pre) {
    if constexpr (EPI == EPI_BIAS_RESID_STATS) {
        if (g.resid_lo) {
            if (row0 + 128 <= g.M)
                epilogue_rows8_impl<EPI, false, PRE, true>(g, acc, slab, row0, col0, lane, pre);
            else
                epilogue_rows8_impl<EPI, true, PRE, true>(g, acc, slab, row0, col0, lane, pre);
            return;
        }
    }
    if (row0 + 128 <= g.M)
        epilogue_rows8_impl<EPI, false, PRE>(g, acc, slab, row0, col0, lane, pre);
    else
        epilogue_rows8_impl<EPI, true, PRE>(g, acc, slab, row0, col0, lane, pre);
}

// ---- f16-slab form (persistent kernel, EMODE 4; LayerNorm-folded epilogues).  The fold arithmetic (and QuickGELU) needs only the
// row's (mean, rstd) and the column's coefficients, both of which a lane knows in the FRAGMENT layout -- so it is applied to the
// accumulators in place and the values are rounded to f16 BEFORE the transposition: a 16-row pass is 2 KiB instead of 4, the
// wave's slab holds TWO passes, and pass p + 1 is computed and written while pass p's read-back and stores are in flight (the
// f32 forms serialise write -> read -> store per pass on one LDS round trip each).  Slab row = 128 B; the 16-byte chunk index is
// XORed with (row & 7) and the two 8-byte halves of a chunk are swapped for rows >= 8, which keeps the 8-byte fragment writes
// (16 lanes = 16 rows per cycle) and the 16-byte read-backs conflict-free.  Stores as in EMODE 1: 8 rows x 128 B per instruction.
template <int EPI, bool CHECK, bool PRE>
__device__ __forceinline__ void epilogue_rows8h_impl(const GemmArgs& g, f32x4 (&acc)[8][4], half_t* slab, int row0, int col0, int lane, const float2* pre) {
    static_assert(EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16, "epilogue_rows8h: LayerNorm-folded epilogues only");
    const int frow = lane & 15, fgrp = lane >> 4;
    const int r8 = lane >> 3, c8 = lane & 7;
    const int ldc = g.ldc;
    f32x4 csum[4], bfold[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        csum[j] = *(const f32x4*)(g.colsum + col0 + j * 16 + fgrp * 4);
        bfold[j] = *(const f32x4*)(g.bias + col0 + j * 16 + fgrp * 4);
    }
    const uint32_t lane_off = (uint32_t)(row0 + r8) * (uint32_t)ldc + (uint32_t)(col0 + c8 * 8);
    // write offsets (halfs) of this lane's four 8-byte pieces inside a pass buffer, read offset of its 16-byte chunk
    const int whalf = ((fgrp & 1) ^ (frow >> 3)) * 4;
    const int wrow = frow * 64;
    const int wx = frow & 7;
    const int roff = r8 * 64 + ((c8 ^ r8) * 8);          // rows r8 and r8 + 8 share (row & 7)
    auto produce = [&](int p) {
        float2 st;
        if constexpr (PRE) {
            const int src = (((p & 3) * 16 + frow) << 2);
            const float2 pv = pre[p >> 2];
            st.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.x)));
            st.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.y)));
        } else {
            int rr_ = row0 + p * 16 + frow;
            rr_ = rr_ < g.M ? rr_ : g.M - 1;
            st = row_stat(g, rr_);
        }
        half_t* buf = slab + (p & 1) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = (acc[p][j] - csum[j] * st.x) * st.y + bfold[j];
            if constexpr (EPI == EPI_LNFOLD_GELU_F16) v = quick_gelu4(v);
            *(half4*)(buf + wrow + (((j * 2 + (fgrp >> 1)) ^ wx) * 8) + whalf) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
    };
    produce(0);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p + 1 < 8) produce(p + 1);
        __builtin_amdgcn_wave_barrier();
        const half_t* buf = slab + (p & 1) * 1024;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            half8 v = *(const half8*)(buf + it * 512 + roff);
            if (it == 1) v = (half8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};      // rows >= 8 keep their halves swapped
            const int row = row0 + p * 16 + it * 8 + r8;
            if (!CHECK || row < g.M) *(half8*)((half_t*)g.out + (CHECK ? (uint32_t)row * (uint32_t)ldc + (uint32_t)(col0 + c8 * 8) : lane_off + (uint32_t)((p * 16 + it * 8) * ldc))) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI, bool PRE>
__device__ __forceinline__ void epilogue_rows8h(const GemmArgs& g, f32x4 (&acc)[8][4], half_t* slab, int row0, int col0, int lane, const float2* pre) {
    if (row0 + 128 <= g.M)
        epilogue_rows8h_impl<EPI, false, PRE>(g, acc, slab, row0, col0, lane, pre);
    else
        epilogue_rows8h_impl<EPI, true, PRE>(g, acc, slab, row0, col0, lane, pre);
}

// ---- Direct form (no LDS transposition at all; persistent kernel, EMODE 2; LayerNorm-folded epilogues).  The kernel reads its
// W fragments with the rows of the wave's 64-column tile PERMUTED -- MFMA row m of column block j is tile column
// (m >> 2) * 16 + j * 4 + (m & 3) -- so a lane's acc[i][0..3] are SIXTEEN consecutive columns (lane >> 4) * 16 .. + 15 of row
// i * 16 + (lane & 15): the lane stores 32 contiguous bytes (two 16-byte stores back to back), the four lanes l, l + 16, l + 32,
// l + 48 complete the row's 128-byte line, and no slab write / wave barrier / slab read stands between the accumulators and the
// store.  Measured against the 16-byte slab form (r02, TF/s in the loop): c_fc (QuickGELU: the epilogue is VALU-heavy and the slab
// traffic competes with it) 869-878 -> 903; QKV 947-958 -> 858 and the residual epilogue 1 008-1 022 -> 954-962 (store-bound
// epilogues: whole-line stores win); with the lane's 32 bytes split into two column passes (half the coefficient registers) c_fc
// falls to 852-856.  So this form serves c_fc only.
template <int EPI, bool CHECK, bool PRE>
__device__ __forceinline__ void epilogue_direct_impl(const GemmArgs& g, f32x4 (&acc)[8][4], int row0, int col0, int lane, const float2* pre) {
    static_assert(EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16, "epilogue_direct: LayerNorm-folded epilogues only");
    const int frow = lane & 15, fgrp = lane >> 4;
    const int col = col0 + fgrp * 16;
    const int ldc = g.ldc;
    const uint32_t lane_off = (uint32_t)(row0 + frow) * (uint32_t)ldc + (uint32_t)col;
    auto pack8 = [](const f32x4& a, const f32x4& b) {
        return (half8){(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
    };
    f32x4 csum[4], bfold[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        csum[j] = *(const f32x4*)(g.colsum + col + j * 4);
        bfold[j] = *(const f32x4*)(g.bias + col + j * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = row0 + i * 16 + frow;
        const uint32_t o = CHECK ? (uint32_t)(row < g.M ? row : g.M - 1) * (uint32_t)ldc + (uint32_t)col : lane_off + (uint32_t)(i * 16 * ldc);
        float2 st;
        if constexpr (PRE) {
            const int src = (((i & 3) * 16 + frow) << 2);
            const float2 pv = pre[i >> 2];
            st.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.x)));
            st.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, pv.y)));
        } else {
            int rr_ = row;
            if constexpr (CHECK) rr_ = rr_ < g.M ? rr_ : g.M - 1;
            st = row_stat(g, rr_);
        }
        if (!CHECK || row < g.M) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (acc[i][j] - csum[j] * st.x) * st.y + bfold[j];
            if constexpr (EPI == EPI_LNFOLD_GELU_F16) {
                if (g.out2) {
                    *(half8*)((half_t*)g.out2 + o) = pack8(v[0], v[1]);
                    *(half8*)((half_t*)g.out2 + o + 8) = pack8(v[2], v[3]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = quick_gelu4(v[j]);
            }
#ifdef GRIP_ABLATE
            if (g.ablate & 4) {         // the stores are issued, but into a 5 120-row window (31 MB: they never have to leave the Infinity Cache)
                const uint32_t ow = (uint32_t)(row % 5120) * (uint32_t)ldc + (uint32_t)col;
                *(half8*)((half_t*)g.out + ow) = pack8(v[0], v[1]);
                *(half8*)((half_t*)g.out + ow + 8) = pack8(v[2], v[3]);
                continue;
            }
            if (g.ablate & 1) {         // everything but the store (the values stay live: they are stored when the row index is impossible)
                if (row == 0x7fffffff) {
                    *(half8*)((half_t*)g.out + o) = pack8(v[0], v[1]);
                    *(half8*)((half_t*)g.out + o + 8) = pack8(v[2], v[3]);
                }
                continue;
            }
#endif
            *(half8*)((half_t*)g.out + o) = pack8(v[0], v[1]);
            *(half8*)((half_t*)g.out + o + 8) = pack8(v[2], v[3]);
        }
    }
}

template <int EPI, bool PRE>
__device__ __forceinline__ void epilogue_direct(const GemmArgs& g, f32x4 (&acc)[8][4], int row0, int col0, int lane, const float2* pre) {
    if (row0 + 128 <= g.M)
        epilogue_direct_impl<EPI, false, PRE>(g, acc, row0, col0, lane, pre);
    else
        epilogue_direct_impl<EPI, true, PRE>(g, acc, row0, col0, lane, pre);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WMF = 16-row fragments per wave along M: 4 -> 128x128 block tile, 2 -> 64x128 (small-M problems such as the
// training batches, where 128-row tiles leave half of the 256 CUs without a workgroup).
// First K slice (64 wide) of a tile in the 256-column panel tn of an output with tiles_n such panels: see gemm_k64p_kernel (K rotation).
// EVERY kernel of this file walks its K slices cyclically from this slice, so a row's summation order -- and with it every bit of the
// result -- is the same whichever tile shape the launcher picks for the row count at hand (split-K launches excepted: they are
// backward-only and sum their partials elsewhere).
__device__ __forceinline__ int k_rot(int tn, int tiles_n, int nk) {
    if (GRIP_KROT == 0) return 0;
    if (tiles_n >= 6) return (tn * nk) / tiles_n;
    return tn % nk;
}
__device__ __forceinline__ int k_rot_cols(int n0, int N, int nk) { return (N & 255) ? 0 : k_rot(n0 >> 8, N >> 8, nk); }

template <int EPI, int WMF>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int BMT = 2 * WMF * 16;
    constexpr int STAGE = (BMT + BN) * BK;
    constexpr int GA = BMT / 32;                 // global_load_lds per wave per tile for A (8 rows each)
    __shared__ __attribute__((aligned(16))) half_t lds[2 * STAGE];

    // ---- XCD-aware, bijective tile remap (block b runs on XCD b % 8)
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BMT, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- staging addresses: wave w fills rows [w*BMT/4, +BMT/4) of the A tile and rows [w*32, +32) of the W
    // tile, 8 rows (1 KiB) per instruction; lane l -> row l>>3, LDS chunk l&7, source chunk (l&7)^(l>>3).
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const size_t K = (size_t)g.K;
    // (wave-uniform bases + one constant 32-bit byte offset per lane: see gemm_k64p_kernel)
    const half_t* a_src = (const half_t*)g.A + (size_t)(m0 + wave * (BMT / 4)) * K;
    const half_t* w_src = (const half_t*)g.W + (size_t)(n0 + wave * 32) * K;
    const uint32_t lane_off = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(schunk * 8));

    auto stage = [&](int buf, int kt) {
        half_t* abase = lds + buf * STAGE + wave * (BMT / 4) * BK;
        half_t* bbase = lds + buf * STAGE + BMT * BK + wave * 32 * BK;
        const half_t* as = a_src + (size_t)kt * BK;
        const half_t* ws = w_src + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(as + (size_t)i * 8 * K) + lane_off), (AS3 void*)(abase + i * 8 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(ws + (size_t)i * 8 * K) + lane_off), (AS3 void*)(bbase + i * 8 * BK), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (in halfs) inside a stage
    const int frow = lane & 15;        // row inside a 16-row fragment; (row & 7) == (lane & 7)
    const int fgrp = lane >> 4;        // k-chunk group 0..3
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = (kk * 4 + fgrp) ^ (lane & 7);
        a_off[kk] = (wr * WMF * 16 + frow) * BK + chunk * 8;
        b_off[kk] = BMT * BK + (wc * 64 + frow) * BK + chunk * 8;
    }

    f32x4 acc[WMF][4];
    init_acc<EPI, WMF>(g, acc, n0 + wc * 64, lane);
    // LayerNorm-folded epilogues: (mean, rstd) of the wave's WMF x 16 rows, lane l <- row l, fetched before the K loop (see gemm_k64_kernel)
    constexpr bool FOLD = (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16);
    float2 pre[1] = {make_float2(0.f, 0.f)};
    f32x4 cb[2] = {};
    if constexpr (FOLD) {
        int r = m0 + wr * WMF * 16 + (lane & (WMF * 16 - 1));
        r = r < g.M ? r : g.M - 1;
        pre[0] = row_stat(g, r);
        cb[0] = *(const f32x4*)(g.colsum + n0 + wc * 64 + (lane & 15) * 4);
        cb[1] = *(const f32x4*)(g.bias + n0 + wc * 64 + (lane & 15) * 4);
    }

    int nk = g.K / BK, kt0 = 0;
    if constexpr (EPI == EPI_F32) {
        if (gridDim.y > 1) {       // split-K: this workgroup contracts k tiles [kt0, kt0 + nk) into partial buffer blockIdx.y
            nk /= (int)gridDim.y;
            kt0 = (int)blockIdx.y * nk;
            g.out = (float*)g.out + (size_t)blockIdx.y * (size_t)g.split_stride;
        }
    }
    const int rot = ((gridDim.y > 1 ? 0 : k_rot_cols(n0, g.N, nk)) + tm * g.rot_rows) % nk;
    auto ks = [&](int k) { return k + rot < nk ? k + rot : k + rot - nk; };
    stage(0, kt0 + ks(0));
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of tile kt has landed (explicit: never left to the compiler)
        __syncthreads();  // tile kt visible to every wave, tile kt-1 fully read
        if (kt + 1 < nk) stage(buf ^ 1, kt0 + ks(kt + 1));
        const half_t* st = lds + buf * STAGE;
        if (GRIP_SMALL_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 af[WMF], bf[4];
#pragma unroll
            for (int i = 0; i < WMF; ++i) af[i] = *(const half8*)(st + a_off[kk] + i * 16 * BK);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(st + b_off[kk] + j * 16 * BK);
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (GRIP_SMALL_PRIO) __builtin_amdgcn_s_setprio(0);
    }

    __syncthreads();   // every wave is done with the stage buffers: reuse them as epilogue slabs
    epilogue_rows<EPI, WMF, 2, FOLD>(g, acc, (float*)lds + wave * EPI_SLAB_FLOATS, m0 + wr * WMF * 16, n0 + wc * 64, lane, pre, FOLD ? cb : nullptr);
}

// ------------------------------------------------------------------------------------------------
// Small-M problems with FEW output tiles (the prompt-step GEMMs: M = 2 142 text rows or 3 408 image rows, N = d: 136-324
// tiles of 64x128 on 256 CUs): one workgroup per CU at most, so the two-stage loop above keeps ONE 24 KiB tile in flight per
// CU and runs at (tile bytes) / (L2 latency) = 35 B/ns per CU, a quarter of what a CU can pull.  Same tile, same LDS byte
// layout, same fragments; the stages form a ring of NST tiles fed NST-1 tiles ahead and retired with counted waits
// (vmcnt = loads of the tiles younger than the one about to be multiplied), one raw barrier per K tile.
template <int EPI, int NST>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int WMF = 2;
    constexpr int BMT = 64;
    constexpr int STAGE = (BMT + BN) * BK;       // halfs: 24 KiB
    constexpr int G = 2 + 4;                     // global_load_lds per wave per tile
    extern __shared__ __attribute__((aligned(16))) half_t lds2[];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BMT, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const size_t K = (size_t)g.K;
    // (wave-uniform bases + one constant 32-bit byte offset per lane: see gemm_k64p_kernel)
    const half_t* a_src = (const half_t*)g.A + (size_t)(m0 + wave * 16) * K;
    const half_t* w_src = (const half_t*)g.W + (size_t)(n0 + wave * 32) * K;
    const uint32_t lane_off = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(schunk * 8));

    auto stage = [&](int buf, int kt) {
        half_t* abase = lds2 + buf * STAGE + wave * 16 * BK;
        half_t* bbase = lds2 + buf * STAGE + BMT * BK + wave * 32 * BK;
        const half_t* as = a_src + (size_t)kt * BK;
        const half_t* ws = w_src + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(as + (size_t)i * 8 * K) + lane_off), (AS3 void*)(abase + i * 8 * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(ws + (size_t)i * 8 * K) + lane_off), (AS3 void*)(bbase + i * 8 * BK), 16, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = (kk * 4 + fgrp) ^ (lane & 7);
        a_off[kk] = (wr * WMF * 16 + frow) * BK + chunk * 8;
        b_off[kk] = BMT * BK + (wc * 64 + frow) * BK + chunk * 8;
    }

    f32x4 acc[WMF][4];
    init_acc<EPI, WMF>(g, acc, n0 + wc * 64, lane);

    int nk = g.K / BK, kt0 = 0;                   // nk >= NST - 1 (launcher)
    if constexpr (EPI == EPI_F32) {
        if (gridDim.y > 1) {                      // split-K: k tiles [kt0, kt0 + nk) into partial buffer blockIdx.y
            nk /= (int)gridDim.y;
            kt0 = (int)blockIdx.y * nk;
            g.out = (float*)g.out + (size_t)blockIdx.y * (size_t)g.split_stride;
        }
    }
    const int rot = ((gridDim.y > 1 ? 0 : k_rot_cols(n0, g.N, nk)) + tm * g.rot_rows) % nk;
    auto ks = [&](int k) { return k + rot < nk ? k + rot : k + rot - nk; };
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) stage(t, kt0 + ks(t));
    int buf = 0, nbuf = NST - 1;                  // slot of tile kt, slot the tile kt + NST - 1 goes to
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = nk - 1 - kt;          // tiles issued after tile kt that may still be in flight (capped at NST - 2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (younger >= NST - 2) wait_vmcnt<(NST - 2) * G>();
        else if (NST > 3 && younger == 1) wait_vmcnt<G>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();             // tile kt landed for every wave; the slot of tile kt-1 has been read by every wave
        if (kt + NST - 1 < nk) stage(nbuf, kt0 + ks(kt + NST - 1));
        const half_t* st = lds2 + buf * STAGE;
        if (GRIP_SMALL_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 af[WMF], bf[4];
#pragma unroll
            for (int i = 0; i < WMF; ++i) af[i] = *(const half8*)(st + a_off[kk] + i * 16 * BK);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(st + b_off[kk] + j * 16 * BK);
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (GRIP_SMALL_PRIO) __builtin_amdgcn_s_setprio(0);
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }

    __syncthreads();   // every wave is done with the ring: reuse it as epilogue slabs
    epilogue_rows<EPI, WMF>(g, acc, (float*)lds2 + wave * EPI_SLAB_FLOATS, m0 + wr * WMF * 16, n0 + wc * 64, lane);
}

// ------------------------------------------------------------------------------------------------
// The ring with the feed on its own waves (r03).  Measured (tools/micro/dma_feed_small.hip, tools/exp_r03_5.sh): a CU pulls ~80 GB/s
// through global_load_lds whatever the number of issuing waves or the ring depth, and a wave that issues a piece while that path is
// busy sits in the issue stage until the piece is accepted -- so in gemm_ring_kernel, where the same four waves issue the loads AND the
// MFMAs, feed time and compute time ADD UP (64x128 tiles, K = 3 072, one workgroup per CU: 15.6 us of feed alone, 11.7 us of fragment
// reads and MFMAs alone, 26 us together).  Here waves 4..7 only issue the loads (and are the ones that block), waves 0..3 only read
// fragments and multiply, software-pipelined over the two 32-wide halves of a K tile (the reads of one half are in flight under the
// MFMAs of the other); one barrier per K tile joins the two groups.  WMF = 2: 64x128 tile (24 KiB stages, up to 6), WMF = 4: 128x128
// (32 KiB stages, up to 5).  Same products in the same order as every other kernel of this file: bit-identical results.
template <int EPI, int NST, int WMF>
__global__ __launch_bounds__(512) void gemm_ringw_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int BMT = 32 * WMF;
    constexpr int STAGE = (BMT + BN) * BK;       // halfs
    constexpr int GA = BMT / 32;
    constexpr int G = GA + 4;                    // global_load_lds per producer wave per tile
    extern __shared__ __attribute__((aligned(16))) half_t lds2[];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        // (the cooperative split-K form pads gridDim.x to a multiple of 8, so that the splits of a tile -- linear workgroup ids gridDim.x apart --
        // land on one XCD and meet in its L2; the padding workgroups leave here)
        if (idx >= (xcd < r ? q + 1 : q)) return;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BMT, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    constexpr bool COOP = (EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_RESID_STATS) && WMF == 2;
    int nk = g.K / BK, kt0 = 0;                   // nk >= NST - 1 (launcher)
    if constexpr (EPI == EPI_F32) {
        if (gridDim.y > 1) {                      // split-K: k tiles [kt0, kt0 + nk) into partial buffer blockIdx.y
            nk /= (int)gridDim.y;
            kt0 = (int)blockIdx.y * nk;
            g.out = (float*)g.out + (size_t)blockIdx.y * (size_t)g.split_stride;
        }
    }
    if constexpr (COOP) {
        if (gridDim.y > 1) {                      // cooperative split-K (GemmArgs::coop_scratch): k tiles [kt0, kt0 + nk), partial tile to the scratch
            nk /= (int)gridDim.y;
            kt0 = (int)blockIdx.y * nk;
        }
    }
    const int rot = ((gridDim.y > 1 ? 0 : k_rot_cols(n0, g.N, nk)) + tm * g.rot_rows) % nk;
    auto ks = [&](int k) { return k + rot < nk ? k + rot : k + rot - nk; };

    if (wave >= 4) {
        // ---- producers: wave 4 + w fills rows [w*BMT/4, +BMT/4) of the A tile and rows [w*32, +32) of the W tile, 8 rows (1 KiB) per
        // instruction; lane l -> row l>>3, LDS chunk l&7, source chunk (l&7)^(l>>3)  (the layout of gemm_f16_kernel)
        const int pw = wave - 4;
        const int srow = lane >> 3;
        const int schunk = (lane & 7) ^ srow;
        const size_t K = (size_t)g.K;
        const half_t* a_src = (const half_t*)g.A + (size_t)(m0 + pw * (BMT / 4)) * K;
        const half_t* w_src = (const half_t*)g.W + (size_t)(n0 + pw * 32) * K;
        const uint32_t lane_off = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(schunk * 8));
        auto stage = [&](int buf, int kt) {
            half_t* abase = lds2 + buf * STAGE + pw * (BMT / 4) * BK;
            half_t* bbase = lds2 + buf * STAGE + BMT * BK + pw * 32 * BK;
            const half_t* as = a_src + (size_t)kt * BK;
            const half_t* ws = w_src + (size_t)kt * BK;
#pragma unroll
            for (int i = 0; i < GA; ++i)
                __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(as + (size_t)i * 8 * K) + lane_off), (AS3 void*)(abase + i * 8 * BK), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(ws + (size_t)i * 8 * K) + lane_off), (AS3 void*)(bbase + i * 8 * BK), 16, 0, 0);
        };
#pragma unroll
        for (int t = 0; t < NST - 1; ++t) stage(t, kt0 + ks(t));
        wait_vmcnt<(NST - 2) * G>();                  // tile 0 (nk >= NST - 1: NST - 2 younger tiles are in flight)
        __builtin_amdgcn_s_barrier();
        int nbuf = NST - 1;                           // slot the tile kt + NST - 1 goes to (= the slot of tile kt - 1)
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                // tile kt + 1 has landed: issued so far are the tiles up to min(nk - 1, kt + NST - 2)
                const int younger = nk - 2 - kt;
                if (younger >= NST - 3) wait_vmcnt<(NST - 3) * G>();
                else if (NST > 5 && younger == 2) wait_vmcnt<2 * G>();
                else if (NST > 4 && younger == 1) wait_vmcnt<G>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();         // ... and the consumers are through step kt - 1: the slot of tile kt - 1 is free
            }
            if (kt + NST - 1 < nk) stage(nbuf, kt0 + ks(kt + NST - 1));
            nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
        }
        __builtin_amdgcn_s_barrier();                 // (the consumers' "ring is free" barrier)
        return;
    }

    // ---- consumers: 2 x 2 waves, each (WMF x 16) x 64 of the tile
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fgrp = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = (kk * 4 + fgrp) ^ (lane & 7);
        a_off[kk] = (wr * WMF * 16 + frow) * BK + chunk * 8;
        b_off[kk] = BMT * BK + (wc * 64 + frow) * BK + chunk * 8;
    }
    f32x4 acc[WMF][4];
    init_acc<EPI, WMF>(g, acc, n0 + wc * 64, lane);
    if constexpr (COOP) {
        if (blockIdx.y > 0) {                         // the bias enters once, with split 0
#pragma unroll
            for (int i = 0; i < WMF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // LayerNorm-folded epilogues: (mean, rstd) of the wave's WMF x 16 rows, lane l <- row l, fetched a whole K loop ahead of their use -- from the
    // finalised array, or straight from the producer's per-column-tile partial sums (GemmArgs::stat_in; ln_stats_finalize's order and arithmetic)
    constexpr bool FOLD = (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16);
    float2 pre[1] = {make_float2(0.f, 0.f)};
    f32x4 cb[2] = {};
    if constexpr (FOLD) {
        static_assert(WMF * 16 <= 64, "one (mean, rstd) pair per lane");
        cb[0] = *(const f32x4*)(g.colsum + n0 + wc * 64 + (lane & 15) * 4);      // epilogue_rows: a lane stores columns col0 + (lane & 15) * 4 .. + 3
        cb[1] = *(const f32x4*)(g.bias + n0 + wc * 64 + (lane & 15) * 4);
        int r = m0 + wr * WMF * 16 + (lane & (WMF * 16 - 1));
        r = r < g.M ? r : g.M - 1;
        pre[0] = row_stat(g, r);
    }
    half8 fa[2][WMF], fb[2][4];
    auto rd = [&](const half_t* st, int kk) {
#pragma unroll
        for (int i = 0; i < WMF; ++i) fa[kk][i] = *(const half8*)(st + a_off[kk] + i * 16 * BK);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[kk][j] = *(const half8*)(st + b_off[kk] + j * 16 * BK);
    };
    auto mm = [&](int kk) {
#pragma unroll
        for (int i = 0; i < WMF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
    };
    __builtin_amdgcn_s_barrier();                     // tile 0 is in LDS
    rd(lds2, 0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nxt = buf + 1 == NST ? 0 : buf + 1;
        if (kt + 1 < nk) __builtin_amdgcn_s_barrier();   // tile kt + 1 is in LDS; every consumer is through step kt - 1
        __builtin_amdgcn_sched_barrier(0);
        rd(lds2 + buf * STAGE, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) rd(lds2 + nxt * STAGE, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
        buf = nxt;
    }
    __builtin_amdgcn_s_barrier();   // every consumer is done with the ring: reuse it as epilogue slabs
    if constexpr (COOP) {
        if (gridDim.y > 1) {
            const int nsplit = (int)gridDim.y;
            constexpr int NF = WMF * 4;                                   // f32x4 per lane
            f32x4* mine = (f32x4*)g.coop_scratch + ((size_t)(bid * nsplit + (int)blockIdx.y) * 4 + wave) * (NF * 64) + lane;
            const f32x4* all = (const f32x4*)g.coop_scratch + ((size_t)(bid * nsplit) * 4 + wave) * (NF * 64) + lane;
            (void)mine;
            constexpr size_t SPLIT_STRIDE = (size_t)4 * NF * 64;
            int old = 0;
#if GRIP_COOP_MODE == 0
            // release / acquire fences at device scope: on gfx950 a write-back of the XCD's whole dirty L2 per wave (measured: the GEMM 13 -> 23 us)
#pragma unroll
            for (int f = 0; f < NF; ++f) mine[f * 64] = acc[f >> 2][f & 3];
            __threadfence();
            if (lane == 0) old = atomicAdd(g.coop_counter + bid * 4 + wave, 1);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != nsplit - 1) return;                                // not the last split of this (tile, wave): done
            __threadfence();
            auto ld = [&](const f32x4* p) { return *p; };
#elif GRIP_COOP_MODE == 1
            // plain stores, retired (= in the L2) before the ticket is drawn; plain loads.  Correct only while the splits of a tile share an XCD.
#pragma unroll
            for (int f = 0; f < NF; ++f) mine[f * 64] = acc[f >> 2][f & 3];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) old = __hip_atomic_fetch_add(g.coop_counter + bid * 4 + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != nsplit - 1) return;
            auto ld = [&](const f32x4* p) { return *p; };
#else
            // The partial tiles are stored WRITE-THROUGH (sc1) and read with sc1 loads (past the L1, served by the memory side), so they are coherent wherever
            // the splits of a tile run and no fence is needed -- a device-scope release on gfx950 writes back the XCD's whole dirty L2 (measured: this GEMM
            // 13 -> 23 us with __threadfence() per wave).  A store is retired (vmcnt) only when the memory side has it; the ticket is a relaxed device-scope
            // atomic drawn after that.  (/opt/skills/guides/cdna_hip_programming.md, Guideline 16: "sc1 stores and loads both sides".)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const size_t scratch_bytes = (size_t)nwg * nsplit * 4 * NF * 64 * 16;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.coop_scratch, 0, (int)scratch_bytes, 0x00020000);
            const uint32_t my_off = (uint32_t)(((size_t)(bid * nsplit + (int)blockIdx.y) * 4 + wave) * (NF * 64) + lane) * 16u;
            const uint32_t all_off = (uint32_t)(((size_t)(bid * nsplit) * 4 + wave) * (NF * 64) + lane) * 16u;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[f >> 2][f & 3]), rsrc, (int)(my_off + f * 64 * 16), 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) old = __hip_atomic_fetch_add(g.coop_counter + bid * 4 + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != nsplit - 1) return;                                // not the last split of this (tile, wave): done
            auto ld = [&](const f32x4* p) {
                return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(all_off + (uint32_t)(p - all) * 16u), 0, 16));
            };
#endif
            // splits 0 .. nsplit - 1 in index order (this wave's own partial read back like the others), up to three splits in flight
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f >> 2][f & 3] = ld(all + f * 64);
            for (int sp = 1; sp < nsplit; sp += 3) {
                f32x4 t[3][NF];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const size_t so = (size_t)(sp + u < nsplit ? sp + u : nsplit - 1) * SPLIT_STRIDE;
#pragma unroll
                    for (int f = 0; f < NF; ++f) t[u][f] = ld(all + so + f * 64);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[f >> 2][f & 3] += (sp + u < nsplit ? t[u][f] : (f32x4){0.f, 0.f, 0.f, 0.f});
            }
            if (lane == 0) g.coop_counter[bid * 4 + wave] = 0;            // ready for the next launch (stream order: nobody else touches it now)
        }
    }
    epilogue_rows<EPI, WMF, (WMF & 1) ? 1 : 2, FOLD>(g, acc, (float*)lds2 + wave * EPI_SLAB_FLOATS, m0 + wr * WMF * 16, n0 + wc * 64, lane, pre, FOLD ? cb : nullptr);
}

// ------------------------------------------------------------------------------------------------
// Large-problem variants: BM x BN x 32 block tile, one wave per 128x64 sub-tile (8x4 fragments, 128
// accumulator registers, 32 MFMA per 12 ds_read_b128 per K tile):
//     256x256: 8 waves, 4-stage ring (128 KiB LDS, one workgroup per CU), 128 FLOP per staged byte;
//     256x128: 4 waves, 3-stage ring ( 72 KiB LDS, two workgroups per CU), 85 FLOP per staged byte --
//              the two co-resident workgroups run out of phase, so one's barrier / ds_read / epilogue
//              time is covered by the other's MFMAs.
// The ring is fed by global_load_lds; the loads of tile t+D (D = stages-1) are issued while tile t is
// being multiplied and are retired with COUNTED waits (s_waitcnt vmcnt((D-1)*G), G = loads per wave per
// tile: "everything but the D-1 youngest tiles") followed by one raw s_barrier per K tile, so L2/HBM
// latency is covered by several tiles of MFMA work (guide: 3-buffer glds + raw barrier).
// Rows of a 32-wide K tile are 64 B = four 16-byte chunks; the chunk index is XORed with
// T[(row >> 2) & 3], T = {0,2,3,1}, on the SOURCE address and on the read: ds_read_b128 conflict-free.
#define BK2 32


template <int EPI, int BMT, int BNT, int NSTAGE>
__global__ __launch_bounds__((BMT / 128) * (BNT / 64) * 64, 2) void gemm_big_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int WN = BNT / 64;                  // waves along N
    constexpr int NW = (BMT / 128) * WN;          // waves per workgroup
    constexpr int STAGE = (BMT + BNT) * BK2;      // halfs per stage
    constexpr int GA = BMT / 16 / NW, GB = BNT / 16 / NW;   // global_load_lds per wave per tile (A rows, W rows)
    constexpr int G = GA + GB;
    extern __shared__ __attribute__((aligned(16))) half_t lds2[];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BMT, n0 = tn * BNT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave - wr * WN;

    // staging: wave w fills A rows [w*GA*16, +GA*16) and W rows [w*GB*16, +GB*16), 16 rows per instruction
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((0x1320 >> (((lane >> 4) & 3) * 4)) & 3);
    const size_t K = (size_t)g.K;
    const half_t* a_src = (const half_t*)g.A + (size_t)(m0 + wave * GA * 16) * K;
    const half_t* w_src = (const half_t*)g.W + (size_t)(n0 + wave * GB * 16) * K;
    const uint32_t lane_off = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(schunk * 8));

    const int rot2 = 2 * k_rot_cols(n0, g.N, g.K / BK), nk2 = g.K / BK2;       // the shared K rotation, in 32-wide steps
    auto stage = [&](int buf, int kt_) {
        const int kt = kt_ + rot2 < nk2 ? kt_ + rot2 : kt_ + rot2 - nk2;
        half_t* abase = lds2 + buf * STAGE + wave * GA * 16 * BK2;
        half_t* bbase = lds2 + buf * STAGE + BMT * BK2 + wave * GB * 16 * BK2;
        const half_t* as = a_src + (size_t)kt * BK2;
        const half_t* ws = w_src + (size_t)kt * BK2;
#pragma unroll
        for (int i = 0; i < GA; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(as + (size_t)i * 16 * K) + lane_off), (AS3 void*)(abase + i * 16 * BK2), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < GB; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(ws + (size_t)i * 16 * K) + lane_off), (AS3 void*)(bbase + i * 16 * BK2), 16, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int fchunk = fgrp ^ ((0x1320 >> (((lane >> 2) & 3) * 4)) & 3);
    const int a_off = (wr * 128 + frow) * BK2 + fchunk * 8;
    const int b_off = BMT * BK2 + (wc * 64 + frow) * BK2 + fchunk * 8;

    f32x4 acc[8][4];
    if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_RESID_STATS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b = *(const f32x4*)(g.bias + n0 + wc * 64 + j * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] = b;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // Fragment registers are double-buffered (set 0 / set 1): while the 32 MFMAs of tile t run on one
    // set, the 12 ds_read_b128 of tile t+1 fill the other, so the LDS read phase that all waves of a
    // workgroup enter together right after the barrier is hidden behind matrix work.
    half8 fa[2][8], fb[2][4];
    auto load_frags = [&](int set, int buf) {
        const half_t* st = lds2 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *(const half8*)(st + b_off + j * 16 * BK2);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[set][i] = *(const half8*)(st + a_off + i * 16 * BK2);
    };
    auto mfma_set = [&](int set) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };

    const int nk = g.K / BK2;   // >= NSTAGE (checked by the launcher)
    constexpr int D = NSTAGE;   // tiles in flight: a tile's LDS slot is free once its fragments are in registers
    // One K step.  On entry fragment set SET holds (or is receiving) tile kt.  WAITN = vmcnt that leaves
    // only the tiles younger than kt+1 outstanding; the barrier then certifies tile kt+1 for every wave
    // and -- because each wave drained its own LDS reads first -- frees the slot of tile kt.
    auto step = [&](auto set_c, auto wait_c, auto has_next_c, int kt) {
        constexpr int SET = decltype(set_c)::value;
        constexpr int WAITN = decltype(wait_c)::value;
        constexpr bool HAS_NEXT = decltype(has_next_c)::value;
        if constexpr (HAS_NEXT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vmcnt<WAITN>();
            __builtin_amdgcn_s_barrier();
            if (kt + D < nk) stage(kt % NSTAGE, kt + D);
            load_frags(SET ^ 1, (kt + 1) % NSTAGE);
        }
        mfma_set(SET);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using WS = std::integral_constant<int, (D - 2) * G>;   // steady state
    using YES = std::true_type;
    using NO = std::false_type;

    if constexpr (NW == 8) {
        // ---- two wave groups in anti-phase (256x256 tile).  Waves 0-3 (upper 128 rows, group A) and 4-7
        // (lower 128 rows, group B) sit pairwise on the four SIMDs.  Every wave runs the SAME loop body
        //     X-barrier | 32 MFMA of tile t | Y-barrier | issue DMA of tile t+4, read fragments of tile t+1
        // but group B enters it one barrier late, so B's X meets A's Y: while one group feeds the matrix
        // pipe the other does its LDS / DMA work, instead of both stalling on fragment reads together.
        // Hazards (barrier numbers: A's X(t) = 2t+1, Y(t) = 2t+2; B's are one higher):
        //   RAW  tile t+1 is read from 2t+2 on (A): every wave waits for its own DMA of tile t+1 before its
        //        X(t), i.e. no later than barrier 2t+2;
        //   WAR  slot t is refilled after Y(t) (2t+2 at the earliest): each wave drains its fragment reads
        //        (lgkmcnt 0) before its X(t), and the later group's X(t) is barrier 2t+2.
        auto pp_step = [&](auto wait_c, auto has_next_c, int kt) {
            constexpr int WAITN = decltype(wait_c)::value;
            constexpr bool HAS_NEXT = decltype(has_next_c)::value;
            if constexpr (HAS_NEXT) wait_vmcnt<WAITN>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                     // X
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mfma_set(0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                     // Y
            __builtin_amdgcn_sched_barrier(0);
            if (kt + D < nk) stage(kt % NSTAGE, kt + D);
            if constexpr (HAS_NEXT) load_frags(0, (kt + 1) % NSTAGE);
        };
#pragma unroll
        for (int t = 0; t < D; ++t) stage(t, t);
        wait_vmcnt<(D - 1) * G>();
        __builtin_amdgcn_s_barrier();        // tile 0 certified
        load_frags(0, 0);
        if (wr == 1) __builtin_amdgcn_s_barrier();            // group B runs one barrier behind
        int kt = 0;
        for (; kt < nk - (D - 1); ++kt) pp_step(WS{}, YES{}, kt);
        static_assert(D == 4 || NW != 8, "256x256 ring has 4 stages");
        pp_step(std::integral_constant<int, G>{}, YES{}, kt);
        pp_step(std::integral_constant<int, 0>{}, YES{}, kt + 1);
        pp_step(I0{}, NO{}, kt + 2);
        if (wr == 0) __builtin_amdgcn_s_barrier();            // group A catches up
    } else {
#pragma unroll
        for (int t = 0; t < D; ++t) stage(t, t);
        wait_vmcnt<(D - 1) * G>();
        __builtin_amdgcn_s_barrier();        // tile 0 certified
        const int R = nk - (D - 1);          // steady-state steps (kt = 0 .. nk-D); the last D-1 steps drain
        int kt = 0;
        if (R & 1) {
            load_frags(1, 0);
            step(I1{}, WS{}, YES{}, 0);
            kt = 1;
        } else {
            load_frags(0, 0);
        }
        for (; kt < R; kt += 2) {
            step(I0{}, WS{}, YES{}, kt);
            step(I1{}, WS{}, YES{}, kt + 1);
        }
        if constexpr (D == 4) {
            step(I0{}, std::integral_constant<int, G>{}, YES{}, kt);
            step(I1{}, std::integral_constant<int, 0>{}, YES{}, kt + 1);
            step(I0{}, I0{}, NO{}, kt + 2);
        } else {
            static_assert(D == 3, "ring depth 3 or 4");
            step(I0{}, std::integral_constant<int, 0>{}, YES{}, kt);
            step(I1{}, I0{}, NO{}, kt + 1);
        }

    }

    __builtin_amdgcn_s_barrier();   // every wave is done with the ring: reuse it as epilogue slabs
    // (the statistics-carrying and LayerNorm-folded epilogues hold more per-row operands: 16-row passes keep their prefetch within
    // the register budget -- with 32-row passes the 256x128 kernel spilled six dwords and returned wrong values in its last row group)
    epilogue_rows<EPI, 8, (EPI == EPI_BIAS_RESID_STATS || EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16 ? 1 : 2)>(g, acc, (float*)lds2 + wave * EPI_SLAB_FLOATS, m0 + wr * 128, n0 + wc * 64, lane);
}

// ---- 256x256 tile, K staged 64 wide: every DMA instruction moves 8 rows x 128 B, i.e. whole cache lines (the 32-wide
// ring above asks the L2 for 64-byte half lines; tools/micro/dma_feed.hip: 14 TB/s vs 21 TB/s into LDS over 256 CUs).
// Two 64 KiB stages; each stage is multiplied as two 32-wide sub-steps with register double-buffered fragments.  One
// raw barrier per stage sits BETWEEN the sub-steps: at that point every wave has all fragments of stage t in registers
// (so slot t&1 can be refilled with stage t+2) and stage t+1, issued one stage earlier, is certified, so its first
// fragments are fetched under the second sub-step's MFMAs.  NW = 8: waves 128x64, two per SIMD.  (A four-wave build of this and of the ring -- 128x128 per wave, accumulators in
// the 256 AGPRs, a third less LDS fragment traffic -- measured 5-25 % slower: one wave per SIMD leaves the barrier, DMA
// issue and epilogue uncovered.)
// RF = 16-row fragments per wave along M: 8 -> 256x256 tile; 6 -> 192x256 (r03: launches of fewer tiles than CUs run one tile-time whatever
// the tile holds, so M = 3 408 rows as 18 row panels of 192 on 216 CUs beat 14 panels of 256 on 168).
template <int EPI, int NW, int RF = 8>
__global__ __launch_bounds__(NW * 64) void gemm_k64_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int BMT = 32 * RF, BNT = 256;
    constexpr int WN = NW / 2;                    // waves along N
    constexpr int WCOLS = BNT / WN;               // 64 or 128
    constexpr int NJ = WCOLS / 16;
    constexpr int STAGE = (BMT + BNT) * BK;       // halfs per stage (BK = 64)
    constexpr int GI = (BMT + BNT) / 8 / NW;      // DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) half_t lds2[];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BMT, n0 = tn * BNT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave - wr * WN;

    // staging: wave w moves rows [w*GI*8, +GI*8) of the 512-row (A then W) stage, 8 rows per instruction;
    // lane l -> row l>>3, LDS chunk l&7, source chunk (l&7)^(l>>3)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const size_t K = (size_t)g.K;
    const int r0 = wave * GI * 8;                 // uniform; RF = 8: this wave's rows are all A rows or all W rows (GI*8 divides 256)
    const half_t* src = r0 < BMT ? (const half_t*)g.A + (size_t)(m0 + r0) * K : (const half_t*)g.W + (size_t)(n0 + r0 - BMT) * K;
    const half_t* src_w = (const half_t*)g.W + ((ptrdiff_t)n0 + r0 - BMT) * (ptrdiff_t)K;      // RF != 8: the wave whose pieces straddle the A / W boundary
    const uint32_t lane_off = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(schunk * 8));
    auto stage = [&](int buf, int kt) {
        half_t* dst = lds2 + buf * STAGE + r0 * BK;
        const half_t* sp = src + (size_t)kt * BK;
        const half_t* spw = src_w + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const half_t* row = (RF == 8 || r0 + i * 8 < BMT) ? sp : spw;     // (uniform select)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(row + (size_t)i * 8 * K) + lane_off), (AS3 void*)(dst + i * 8 * BK), 16, 0, 0);
        }
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = (kk * 4 + fgrp) ^ (lane & 7);
        a_off[kk] = (wr * RF * 16 + frow) * BK + chunk * 8;
        b_off[kk] = BMT * BK + (wc * WCOLS + frow) * BK + chunk * 8;
    }

    f32x4 acc[NJ / 4][RF][4];
#pragma unroll
    for (int h = 0; h < NJ / 4; ++h) init_acc<EPI, RF>(g, acc[h], n0 + wc * WCOLS + h * 64, lane);

    half8 fa[2][RF], fb[2][NJ];
    auto load_frags = [&](int set, int buf, int kk) {
        const half_t* st = lds2 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[set][j] = *(const half8*)(st + b_off[kk] + j * 16 * BK);
#pragma unroll
        for (int i = 0; i < RF; ++i) fa[set][i] = *(const half8*)(st + a_off[kk] + i * 16 * BK);
    };
    auto mfma_set = [&](int set) {
#pragma unroll
        for (int i = 0; i < RF; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[j >> 2][i][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[set][j], fa[set][i], acc[j >> 2][i][j & 3], 0, 0, 0);
    };
    auto spread = [&]() {     // interleave the (RF + NJ) fragment reads of the other set with this set's RF*NJ MFMAs
#pragma unroll
        for (int q = 0; q < RF + NJ; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, (RF * NJ) / (RF + NJ) > 3 ? 3 : 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };

    const int nk = g.K / BK;    // >= 2 (checked by the launcher)
    const int rot = (k_rot(tn, tiles_n, nk) + tm * g.rot_rows) % nk;
    auto ks = [&](int k) { return k + rot < nk ? k + rot : k + rot - nk; };
    // LayerNorm-folded epilogues: (mean, rstd) of the wave's RF x 16 rows, lane l <- rows l and 64 + l, before the K loop (row_stat: finalised pairs or the
    // producer's partial sums); the epilogue fetches a row's pair with two ds_bpermute (one load per ROW instead of one per row group and column quad)
    constexpr bool FOLD = (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16);
    float2 pre[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    f32x4 cb[NJ / 4][2] = {};       // colsum / folded bias of this lane's four columns, per 64-column half
    if constexpr (FOLD) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int r = m0 + wr * RF * 16 + h * 64 + lane;
            r = r < g.M ? r : g.M - 1;
            pre[h] = row_stat(g, r);
        }
#pragma unroll
        for (int h = 0; h < NJ / 4; ++h) {
            cb[h][0] = *(const f32x4*)(g.colsum + n0 + wc * WCOLS + h * 64 + (lane & 15) * 4);
            cb[h][1] = *(const f32x4*)(g.bias + n0 + wc * WCOLS + h * 64 + (lane & 15) * 4);
        }
    }
    stage(0, ks(0));
    stage(1, ks(1));
    wait_vmcnt<GI>();
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        // sub-step 0
        load_frags(1, buf, 1);
        mfma_set(0);
        spread();
        // stage boundary
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(buf, ks(kt + 2));
        // sub-step 1
        if (kt + 1 < nk) load_frags(0, buf ^ 1, 0);
        mfma_set(1);
        spread();
    }

    __builtin_amdgcn_s_barrier();   // every wave is done with the stages: reuse them as epilogue slabs
#pragma unroll
    for (int h = 0; h < NJ / 4; ++h)
        epilogue_rows<EPI, RF, (EPI == EPI_BIAS_RESID_STATS || FOLD ? 1 : 2), FOLD>(g, acc[h], (float*)lds2 + wave * EPI_SLAB_FLOATS, m0 + wr * RF * 16, n0 + wc * WCOLS + h * 64, lane, pre, FOLD ? cb[h] : nullptr);
}

// ---- Developer prototype (r03): the 192x256 tile of gemm_k64_kernel with the stage feed on FOUR DEDICATED WAVES (12 waves per workgroup:
// eight consumers of 96x64 -- 96 accumulator registers, ONE fragment set: three waves share a SIMD and 170 registers each -- and four
// producers that issue the 56 DMA pieces of a stage and are the only ones to sit in the vector-memory issue stage).  Variant 9 of the
// debug hook / GRIP_GEMM_BIG=9; one tile per workgroup.
template <int EPI>
__global__ __launch_bounds__(768) void gemm_k64w_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int RF = 6, BMT = 32 * RF, BNT = 256;
    constexpr int STAGE = (BMT + BNT) * BK;       // halfs per stage
    constexpr int PIECES = (BMT + BNT) / 8;       // 56
    constexpr int GP = PIECES / 4;                // per producer wave
    extern __shared__ __attribute__((aligned(16))) half_t lds2[];

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BMT, n0 = tn * BNT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.K / BK;    // >= 2 (launcher)
    const int rot = (k_rot(tn, tiles_n, nk) + tm * g.rot_rows) % nk;
    auto ks = [&](int k) { return k + rot < nk ? k + rot : k + rot - nk; };

    if (wave >= 8) {
        const int pw = wave - 8;
        const int srow = lane >> 3;
        const int schunk = (lane & 7) ^ srow;
        const size_t K = (size_t)g.K;
        const uint32_t lane_off = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(schunk * 8));
        auto stage = [&](int buf, int kt) {
#pragma unroll
            for (int i = 0; i < GP; ++i) {
                const int pc = pw * GP + i;                   // piece: rows pc * 8 .. + 7 of the (A then W) stage
                const half_t* row = pc * 8 < BMT ? (const half_t*)g.A + (size_t)(m0 + pc * 8) * K : (const half_t*)g.W + (size_t)(n0 + pc * 8 - BMT) * K;
                __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(row + (size_t)kt * BK) + lane_off), (AS3 void*)(lds2 + buf * STAGE + pc * 8 * BK), 16, 0, 0);
            }
        };
        stage(0, ks(0));
        stage(1, ks(1));
        wait_vmcnt<GP>();
        __builtin_amdgcn_s_barrier();                 // stage 0 landed
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt<0>();                          // stage kt + 1 landed
            __builtin_amdgcn_s_barrier();             // ... and every consumer holds the last fragments of stage kt: its slot is free
            if (kt + 2 < nk) stage(kt & 1, ks(kt + 2));
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        return;
    }

    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fgrp = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = (kk * 4 + fgrp) ^ (lane & 7);
        a_off[kk] = (wr * RF * 16 + frow) * BK + chunk * 8;
        b_off[kk] = BMT * BK + (wc * 64 + frow) * BK + chunk * 8;
    }
    f32x4 acc[RF][4];
    init_acc<EPI, RF>(g, acc, n0 + wc * 64, lane);
    half8 fa[RF], fb[4];
    auto rd = [&](int buf, int kk) {
        const half_t* st = lds2 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *(const half8*)(st + b_off[kk] + j * 16 * BK);
#pragma unroll
        for (int i = 0; i < RF; ++i) fa[i] = *(const half8*)(st + a_off[kk] + i * 16 * BK);
    };
    auto mm = [&]() {
#pragma unroll
        for (int i = 0; i < RF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };
    __builtin_amdgcn_s_barrier();                     // stage 0 landed
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        rd(buf, 0);
        mm();
        rd(buf, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // this stage's slot is free; stage kt + 1 has landed
        mm();
    }
    __builtin_amdgcn_s_barrier();   // every consumer is done with the stages: reuse them as epilogue slabs
    // (16-row passes: the operand prefetch of 32-row passes does not fit the 170-register budget of three waves per SIMD)
    epilogue_rows<EPI, RF, 1>(g, acc, (float*)lds2 + wave * EPI_SLAB_FLOATS, m0 + wr * RF * 16, n0 + wc * 64, lane);
}

// ---- Persistent form of gemm_k64_kernel: one workgroup per CU walks its XCD's run of tiles, and the two-stage K pipeline
// simply continues across tile boundaries -- the first two stages of tile t+1 are issued at the last two stage
// boundaries of tile t and land while tile t's epilogue runs, so a tile no longer starts with an exposed HBM/L2 round trip
// (~2 us of a ~32 us K = 768 tile) nor ends with an idle DMA queue.  The epilogue slabs therefore cannot reuse the stage
// buffers: they are 16-row, 4 KiB, swizzled slabs in the 32 KiB of LDS beside the two 64 KiB stages.
// EMODE: 0 = 8-byte-store slab epilogue (every epilogue), 1 = 16-byte-store slab epilogue, 2 = direct epilogue with permuted W
// fragment rows (1 and 2: the three f16-output epilogues of the pool encode).
template <int EPI, int EMODE = 0, bool SD = false>
__global__ __launch_bounds__(512) void gemm_k64p_kernel(GemmArgs g, int tiles_m, int tiles_n, int colgroup) {
    constexpr int BMT = 256, BNT = 256, NW = 8, WN = 4;
    constexpr int STAGE = (BMT + BNT) * BK;       // halfs per stage (BK = 64)
    constexpr int GI = (BMT + BNT) / 8 / NW;      // DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) half_t lds2[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave - wr * WN;
    float* slab = (float*)(lds2 + 2 * STAGE) + wave * 1024;

    // Tile walk.  XCD x (= blockIdx & 7; its own 4 MiB L2) owns a contiguous band of ROW panels; inside the band the tiles
    // are ordered column group by column group (colgroup tiles wide), rows fastest after the group's columns:
    //     for cg: for row in band: for c in cg's columns: tile (row, c)
    // and the XCD's workgroups take them round-robin, so the 32 tiles in flight on an XCD are (32 / colgroup) row panels x
    // colgroup column panels.  The W panels of one column group (colgroup x 256 x K halfs: 1.5 MB for 4 x K = 768) then stay
    // L2-resident for the whole sweep down the band while A panels stream through once per group.  With colgroup = tiles_n
    // this is the plain N-fastest walk, under which the 4.7 MB of c_fc's twelve W panels were re-fetched from the memory
    // side for every few row panels (round 1: 4.3 GB of fetches per launch against 0.4 GB of A).
    // colgroup = 0 (launches with fewer than 64 row panels, where whole-row bands would load the XCDs unevenly): the XCD owns a
    // contiguous run of tiles, N-fastest, split evenly at tile granularity.
    const int xcd = blockIdx.x & 7;
    const int per_xcd = gridDim.x >> 3;
    int row_start, nrows, xstart = 0, xcount, group_tiles = 1;
    if (colgroup > 0) {
        const int rows_q = tiles_m >> 3, rows_r = tiles_m & 7;
        row_start = xcd < rows_r ? xcd * (rows_q + 1) : rows_r * (rows_q + 1) + (xcd - rows_r) * rows_q;
        nrows = rows_q + (xcd < rows_r ? 1 : 0);
        xcount = nrows * tiles_n;
        group_tiles = nrows * colgroup;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int xq = nwg >> 3, xr = nwg & 7;
        xstart = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
        xcount = xq + (xcd < xr ? 1 : 0);
        row_start = 0;
        nrows = 0;
    }
    int t = blockIdx.x >> 3;
    if (t >= xcount) return;
    auto tile_coords = [&](int tile, int& tm, int& tn) {
        if (colgroup > 0) {
            const int cg = tile / group_tiles, rem = tile - cg * group_tiles;
            const int r = rem / colgroup;
            tm = row_start + r;
            tn = cg * colgroup + (rem - r * colgroup);
        } else {
            const int bid = xstart + tile;
            tm = bid / tiles_n;
            tn = bid - tm * tiles_n;
        }
    };

    const int srow = lane >> 3;
    const size_t K = (size_t)g.K;
    const int r0 = wave * GI * 8;
    // Source-side swizzle: LDS slot (row, c) receives source chunk c ^ key(row).  key = row & 7 for the A rows (and for W in
    // EMODE 0 / 1).  EMODE 2 reads the W fragments through the row permutation of epilogue_direct, under which the eight rows a
    // fragment read touches per LDS cycle are {a * 16 + j * 4 + b: a in 0..1, b in 0..3}: key = (row & 3) | ((row >> 4) & 1) << 2
    // keeps those reads conflict-free.  A wave's 64 staged rows are rows i * 8 + srow of a 64-row block, so bit 4 is (i >> 1) & 1.
    // Addresses of the DMA pieces: a wave-UNIFORM 64-bit base (tile, K slice, piece: scalar registers and scalar adds) plus a 32-bit
    // per-lane offset that never changes (row srow of the piece, swizzled chunk) -- the saddr + voffset form of global_load_lds.  With
    // per-lane 64-bit pointers every piece cost two 64-bit vector adds (16 VALU issues per stage and wave, between the MFMAs).
    uint32_t lane_off[2];       // in bytes (the builtin is still selected with a 64-bit vector address: one v_lshl_add_u64 per piece from the scalar
                                // base; the saddr form written as inline asm measured the same, so the compiler-visible builtin stays)
    if (EMODE == 2 && r0 >= BMT) {
        lane_off[0] = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(((lane & 7) ^ (srow & 3)) * 8));
        lane_off[1] = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)((((lane & 7) ^ (srow & 3)) * 8) ^ 32));
    } else {
        lane_off[0] = lane_off[1] = 2u * ((uint32_t)srow * (uint32_t)g.K + (uint32_t)(((lane & 7) ^ srow) * 8));
    }
    auto tile_src = [&](int tile) {         // uniform: first staged row of this wave in the tile's A or W panel
        int tm, tn;
        tile_coords(tile, tm, tn);
#ifdef GRIP_ABLATE
        if ((g.ablate & 2) && g.K >= 2048 && r0 < BMT) tm %= 20;       // A from a 20-panel window (5 120 rows x 6 KB = 31 MB: Infinity-Cache resident)
#endif
        return r0 < BMT ? (const half_t*)g.A + (size_t)(tm * BMT + r0) * K : (const half_t*)g.W + (size_t)(tn * BNT + r0 - BMT) * K;
    };
    auto stage = [&](int buf, const half_t* src, int kt) {
        half_t* dst = lds2 + buf * STAGE + r0 * BK;
        const half_t* sp = src + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < GI; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(sp + (size_t)i * 8 * K) + lane_off[(i >> 1) & 1]), (AS3 void*)(dst + i * 8 * BK), 16, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    constexpr int BJ = (EMODE == 2 ? 4 : 16);     // LDS rows between a wave's consecutive W fragments
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = (kk * 4 + fgrp) ^ (lane & 7);
        a_off[kk] = (wr * 128 + frow) * BK + chunk * 8;
        if constexpr (EMODE == 2) {
            const int key = (frow & 3) | (((frow >> 2) & 1) << 2);
            b_off[kk] = BMT * BK + (wc * 64 + (frow >> 2) * 16 + (frow & 3)) * BK + (((kk * 4 + fgrp) ^ key) * 8);
        } else {
            b_off[kk] = BMT * BK + (wc * 64 + frow) * BK + chunk * 8;
        }
    }

    f32x4 acc[8][4];
    half8 fa[2][8], fb[2][4];
    auto load_frags = [&](int set, int buf, int kk) {
        const half_t* st = lds2 + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *(const half8*)(st + b_off[kk] + j * BJ * BK);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[set][i] = *(const half8*)(st + a_off[kk] + i * 16 * BK);
    };
    auto mfma_set = [&](int set) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
    auto spread = [&]() {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };

    const int nk = g.K / BK;    // >= 2 (checked by the launcher)
    // K rotation: the tile of column panel tn walks its K slices cyclically from slice (tn * nk) / tiles_n.  The column tiles of one row
    // panel run side by side on an XCD and would otherwise ask for the SAME slice of their shared A panel at the same moment -- every one
    // of them then waits out the HBM / Infinity-Cache latency of every slice; rotated, a slice is fetched by one tile and found in the
    // L2 by the others.  The start depends on the column panel only, so an output element's summation order does not depend on where
    // its row sits in the launch (rows stay bit-identical under any chunking / sharding).
    auto tile_rot = [&](int tile) {
        int tm, tn;
        tile_coords(tile, tm, tn);
        // wide outputs (tiles_n >= 6: the XCD's 32 tiles span few row panels, every slice stays in the L2 for the followers): full spread.
        // Narrow outputs (N = 768: ~11 row panels of A in flight per XCD, more than the L2 holds for long): a skew of ONE slice per column
        // panel -- the follower arrives one stage after the leader's fetch has landed (residual GEMM 1 023 -> 1 040 TF/s; skews of 2 or 3
        // slices: 1 029; the full spread: 981)
        return k_rot(tn, tiles_n, nk);
    };
    auto ks = [&](int k, int rot) { return k + rot < nk ? k + rot : k + rot - nk; };
    const half_t* src_cur = tile_src(t);
    int rot_cur = tile_rot(t);
    stage(0, src_cur, ks(0, rot_cur));
    stage(1, src_cur, ks(1, rot_cur));
    wait_vmcnt<GI>();
    __builtin_amdgcn_s_barrier();
    int par = 0;                // LDS slot of the current tile's stage 0
    for (;;) {
        int tm, tn;
        tile_coords(t, tm, tn);
        const int m0 = tm * BMT, n0 = tn * BNT;
        const int t_next = t + per_xcd;
        const bool has_next = t_next < xcount;
        const half_t* src_next = has_next ? tile_src(t_next) : src_cur;
        const int rot_next = has_next ? tile_rot(t_next) : 0;
        if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_RESID_STATS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b = *(const f32x4*)(g.bias + n0 + wc * 64 + (EMODE == 2 ? (lane >> 4) * 16 + j * 4 : j * 16 + (lane >> 4) * 4));
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j] = b;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // (mean, rstd) of this wave's 128 rows for the LayerNorm-folded epilogues: issued here, a whole K loop ahead of their use
        float2 pre[2];
        if constexpr (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int r = m0 + wr * 128 + h * 64 + lane;
                r = r < g.M ? r : g.M - 1;
                pre[h] = ((const float2*)g.rowstat)[r];      // (finalised pairs only: the launcher finalises partial sums for this kernel -- keeps the hot loop's code as it was)
            }
        }
        load_frags(0, par, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = (par + kt) & 1;
            load_frags(1, buf, 1);
            constexpr bool PRIO0 = (GRIP_SETPRIO & 1) && GRIP_SETPRIO < 4;
            constexpr bool PRIO1 = ((GRIP_SETPRIO & 2) && GRIP_SETPRIO < 4) || (GRIP_SETPRIO == 4 && (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16));
            if constexpr (PRIO0) __builtin_amdgcn_s_setprio(1);
            mfma_set(0);
            spread();
            if constexpr (PRIO0) __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if constexpr (PRIO1) __builtin_amdgcn_s_setprio(1);
            if constexpr (SD) {
                // Sub-step 1 as ONE basic block, so that the scheduler can place the stage's 8 DMA pieces and the 12 fragment reads
                // BETWEEN the 32 MFMAs (as three separate blocks -- the branchy form below -- they are a DMA burst, then a read
                // burst, then the MFMAs, and both waves of a SIMD sit in their bursts at the same time).  Unconditional: the last
                // two stages of a workgroup's last tile re-stage the current tile's first K slice into a dead slot, and the last
                // stage's fragment reads fetch the next tile's first fragments early (dead when there is no next tile).
                const half_t* sp = kt + 2 < nk ? src_cur + (size_t)ks(kt + 2, rot_cur) * BK : (has_next ? src_next + (size_t)ks(kt + 2 - nk, rot_next) * BK : src_cur);
                half_t* dst = lds2 + buf * STAGE + r0 * BK;
                const half_t* st = lds2 + (buf ^ 1) * STAGE;
                constexpr int RD0 = GRIP_RD0;     // first of the 16 MFMA pairs after which a fragment read is placed
                // program order = the intended issue order (LDS reads and LDS-DMA writes may alias as far as the compiler knows, so it
                // keeps their order): 2 MFMAs, 1 fragment read, 1 DMA piece, ...
#pragma unroll
                for (int q = 0; q < 16; ++q) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int mi = (2 * q + h) >> 2, mj = (2 * q + h) & 3;
                        acc[mi][mj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][mj], fa[1][mi], acc[mi][mj], 0, 0, 0);
                    }
                    if (q >= RD0 && q < RD0 + 4) fb[0][q - RD0] = *(const half8*)(st + b_off[0] + (q - RD0) * BJ * BK);
                    else if (q >= RD0 + 4 && q < RD0 + 12) fa[0][q - RD0 - 4] = *(const half8*)(st + a_off[0] + (q - RD0 - 4) * 16 * BK);
                    if (q < GI)
                        __builtin_amdgcn_global_load_lds((const AS1 void*)((const char*)(sp + (size_t)q * 8 * K) + lane_off[(q >> 1) & 1]), (AS3 void*)(dst + q * 8 * BK), 16, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (q >= RD0 && q < RD0 + 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (q < GI) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            } else {
                if (kt + 2 < nk) stage(buf, src_cur, ks(kt + 2, rot_cur));
                else if (has_next) stage(buf, src_next, ks(kt + 2 - nk, rot_next));     // the next tile's first two stages
                if (kt + 1 < nk) load_frags(0, buf ^ 1, 0);
                mfma_set(1);
                spread();
            }
            if constexpr (PRIO1) __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (EMODE == 4)
            epilogue_rows8h<EPI, true>(g, acc, (half_t*)slab, m0 + wr * 128, n0 + wc * 64, lane, pre);
        else if constexpr (EMODE == 2)
            epilogue_direct<EPI, (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16)>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, pre);
        else if constexpr (EMODE == 1)
            epilogue_rows8<EPI, (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16)>(g, acc, slab, m0 + wr * 128, n0 + wc * 64, lane, pre);
        else
            epilogue_rows<EPI, 8, 1, (EPI == EPI_LNFOLD_F16 || EPI == EPI_LNFOLD_GELU_F16)>(g, acc, slab, m0 + wr * 128, n0 + wc * 64, lane, pre);
        if (!has_next) break;
        par = (par + nk) & 1;
        t = t_next;
        src_cur = src_next;
        rot_cur = rot_next;
    }
}

// ---- optional in-library timing of the GEMM launches (uniform 1-in-4 sample) (HIP events on the launch stream), used by
// bench.py for the live roofline figure.  Off by default.  Records sit in a bounded ring; when it is
// full the oldest half (long finished) is folded into per-epilogue accumulators.
#include <deque>
#include <vector>
namespace {
constexpr int PROF_RING = 4096, PROF_EPIS = 144;   // slot = variant * 16 + epilogue id (variants 0..8)
struct ProfRec { int epi; double flops; hipEvent_t a, b; };
bool g_prof = false;
std::deque<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
int64_t g_launches[PROF_EPIS];
double g_ms[PROF_EPIS], g_flops[PROF_EPIS];
hipEvent_t prof_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void prof_drain(size_t keep) {
    while (g_recs.size() > keep) {
        ProfRec r = g_recs.front();
        g_recs.pop_front();
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.epi >= 0 && r.epi < PROF_EPIS) {
            g_launches[r.epi] += 1;
            g_ms[r.epi] += ms;
            g_flops[r.epi] += r.flops;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
}
}  // namespace

extern "C" int grip_profile_enable(int on) {
    prof_drain(0);
    for (int i = 0; i < PROF_EPIS; ++i) { g_launches[i] = 0; g_ms[i] = 0.0; g_flops[i] = 0.0; }
    g_prof = on != 0;
    return GRIP_OK;
}

// Per slot (variant * 16 + epilogue id) in [0, n): launches[slot], total milliseconds, total algorithmic FLOPs (2*M*N*K) of
// every GEMM launched since grip_profile_enable(1).  Synchronises the outstanding events.
extern "C" int grip_profile_collect(int n, int64_t* launches, double* total_ms, double* total_flops) {
    prof_drain(0);
    for (int i = 0; i < n; ++i) {
        launches[i] = i < PROF_EPIS ? g_launches[i] : 0;
        total_ms[i] = i < PROF_EPIS ? g_ms[i] : 0.0;
        total_flops[i] = i < PROF_EPIS ? g_flops[i] : 0.0;
    }
    return GRIP_OK;
}

static int launch_gemm_impl(int epi, const GemmArgs& a, hipStream_t s, int* chosen);

int launch_gemm(int epi, const GemmArgs& a, hipStream_t s) {
    // sample every 4th launch: the launch sequence is periodic with an odd period (49 GEMMs per
    // encode chunk), so every kernel/shape is sampled uniformly while the markers cost < 1 %
    static unsigned g_tick = 0;
    int chosen = 0;
    if (!g_prof || (++g_tick & 3u)) return launch_gemm_impl(epi, a, s, &chosen);
    {   // never put timing events into a stream that is being captured into a HIP graph (steps.GraphedCoopStep)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return launch_gemm_impl(epi, a, s, &chosen);
    }
    if ((int)g_recs.size() >= PROF_RING) prof_drain(PROF_RING / 2);
    ProfRec r{epi, 2.0 * a.M * (double)a.N * a.K, prof_event(), prof_event()};
    if (!r.a || !r.b) return launch_gemm_impl(epi, a, s, &chosen);
    (void)hipEventRecord(r.a, s);
    const int rc = launch_gemm_impl(epi, a, s, &chosen);
    (void)hipEventRecord(r.b, s);
    r.epi = chosen * 16 + ((epi == EPI_BIAS_RESID && a.stat_part && !a.f32) ? (int)EPI_BIAS_RESID_STATS : epi);   // the instantiation that actually ran
    g_recs.push_back(r);
    return rc;
}

template <int BMT, int BNT, int NSTAGE>
static int launch_big(int epi, const GemmArgs& a, hipStream_t s) {
    const int tiles_m = (a.M + BMT - 1) / BMT, tiles_n = a.N / BNT;
    constexpr size_t lds = (size_t)NSTAGE * (BMT + BNT) * BK2 * 2;
    constexpr int threads = (BMT / 128) * (BNT / 64) * 64;
    dim3 grid(tiles_m * tiles_n), block(threads);
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<E, BMT, BNT, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_big_kernel<E, BMT, BNT, NSTAGE>), grid, block, lds, s, a, tiles_m, tiles_n);               \
    } break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_F16)
        GRIP_GEMM_CASE(EPI_GELUGRAD_F16)
        GRIP_GEMM_CASE(EPI_F32_SCALE)
        GRIP_GEMM_CASE(EPI_LNFOLD_F16)
        GRIP_GEMM_CASE(EPI_LNFOLD_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID_STATS)
        default: GRIP_REQUIRE(false, "gemm: unknown epilogue %d", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

template <int NW, int RF = 8>
static int launch_k64(int epi, const GemmArgs& a, hipStream_t s) {
    constexpr int BMT = 32 * RF;
    const int tiles_m = (a.M + BMT - 1) / BMT, tiles_n = a.N / 256;
    constexpr size_t lds = (size_t)2 * (BMT + 256) * BK * 2;
    dim3 grid(tiles_m * tiles_n), block(NW * 64);
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_k64_kernel<E, NW, RF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_k64_kernel<E, NW, RF>), grid, block, lds, s, a, tiles_m, tiles_n);                             \
    } break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_F16)
        GRIP_GEMM_CASE(EPI_GELUGRAD_F16)
        GRIP_GEMM_CASE(EPI_F32_SCALE)
        GRIP_GEMM_CASE(EPI_LNFOLD_F16)
        GRIP_GEMM_CASE(EPI_LNFOLD_GELU_F16)
        default: GRIP_REQUIRE(false, "gemm: unknown epilogue %d", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

static int launch_k64w(int epi, const GemmArgs& a, hipStream_t s) {
    const int tiles_m = (a.M + 191) / 192, tiles_n = a.N / 256;
    constexpr size_t lds = (size_t)2 * (192 + 256) * BK * 2;
    dim3 grid(tiles_m * tiles_n), block(768);
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_k64w_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_k64w_kernel<E>), grid, block, lds, s, a, tiles_m, tiles_n);                                \
    } break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_GELUGRAD_F16)
        default: GRIP_REQUIRE(false, "gemm: the loader-wave 192x256 prototype has no epilogue %d", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

#ifdef GRIP_ABLATE
static int g_ablate = 0;
extern "C" int grip_debug_ablate(int mask) { g_ablate = mask; return GRIP_OK; }
#endif
static int launch_k64p(int epi, const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
#ifdef GRIP_ABLATE
    a.ablate = g_ablate;
#endif
    const int tiles_m = (a.M + 255) / 256, tiles_n = a.N / 256;
    constexpr size_t lds = (size_t)2 * 512 * BK * 2 + 8 * 4096;       // two stages + eight 4 KiB slabs = the whole 160 KiB
    static int n_cu_dev = 0;
    if (!n_cu_dev) {
        int dev = 0;
        GRIP_CHECK_HIP(hipGetDevice(&dev));
        GRIP_CHECK_HIP(hipDeviceGetAttribute(&n_cu_dev, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu_dev &= ~7;
        GRIP_REQUIRE(n_cu_dev >= 8, "gemm: device reports %d CUs", n_cu_dev);
    }
    int n_cu = n_cu_dev;
    if (grip_cu_budget() > 0 && grip_cu_budget() < n_cu_dev) n_cu = grip_cu_budget() & ~7;     // the launch stream owns fewer CUs (grip_set_cu_budget)
    const int tiles = tiles_m * tiles_n;
    // every XCD owns ceil or floor(tiles_m / 8) row panels: the grid has enough workgroups per XCD for the largest band
    const int band = ((tiles_m + 7) / 8) * tiles_n;
    // Tile walk: the even N-fastest split (colgroup = 0) is the default.  The row-band / column-group walk (see the kernel) cuts
    // the memory-side fetches of the K = 768 GEMMs by keeping one group's W panels L2-resident, but measured SLOWER on the
    // pool encode (r02: c_fc 794 vs 839 TF/s, QKV 891 vs 922 with groups of 4 / 3 column tiles): the refetched W panels come
    // out of the 256 MiB Infinity Cache, not HBM, and the banded walk makes all 32 workgroups of an XCD start their tiles'
    // A panels at once.  GRIP_GEMM_COLGROUP=<n> (a divisor of N / 256; 1 = tiles_n wide bands) switches it on for A/B runs.
    static const int force_cg = getenv("GRIP_GEMM_COLGROUP") ? atoi(getenv("GRIP_GEMM_COLGROUP")) : 0;
    int colgroup = 0;
    if (force_cg > 0 && tiles_m >= 64) {
        colgroup = tiles_n;
        if (a.K <= 1024 && force_cg > 1)
            for (int c = force_cg; c >= 2; --c)
                if (tiles_n % c == 0 && tiles_n > c) { colgroup = c; break; }
    }
    const int grid_n = colgroup ? (band * 8 >= n_cu ? n_cu : band * 8) : (tiles >= n_cu ? n_cu : ((tiles + 7) & ~7));
    dim3 grid(grid_n), block(512);
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_k64p_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_k64p_kernel<E>), grid, block, lds, s, a, tiles_m, tiles_n, colgroup);                                \
    } break;
#define GRIP_GEMM_CASE_M(E, MODE)                                                                                           \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_k64p_kernel<E, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_k64p_kernel<E, MODE>), grid, block, lds, s, a, tiles_m, tiles_n, colgroup);                          \
    } break;
    // Epilogue form of the three pool-encode epilogues: 0 = 8-byte stores through the f32 slab, 1 = 16-byte stores through the f32
    // slab, 2 = direct with permuted W fragment rows, 4 = fold arithmetic in the fragment layout + double-buffered f16 slab (2 and 4:
    // LayerNorm-folded epilogues only).  Default: QKV 4, c_fc 2, residual 1 -- measured in the loop (TF/s, one box): QKV 942 (1) /
    // 929 (2) / 972 (4); c_fc 870 (1) / 892 (2) / 872 (4); residual 1 007 (1) / 954 (2).  GRIP_GEMM_EMODE=<m> forces one mode,
    // three digits one each for QKV / c_fc / residual (developer A/B).
    static const int emode_env = getenv("GRIP_GEMM_EMODE") ? atoi(getenv("GRIP_GEMM_EMODE")) : 421;
    int emode = emode_env < 100 ? emode_env : (epi == EPI_LNFOLD_F16 ? emode_env / 100 : epi == EPI_LNFOLD_GELU_F16 ? (emode_env / 10) % 10 : emode_env % 10);
    if (epi == EPI_BIAS_RESID_STATS && (emode == 2 || emode == 4)) emode = 1;
    if (emode == 4 && a.out2) emode = 1;      // the f16-slab form has no pre-activation copy (train-mode forwards)
    // single-block sub-step 1 (DMA pieces and fragment reads between the MFMAs) for the three default pool-encode instantiations:
    // loop +0.9 %, residual GEMM 987 -> 1 010 TF/s, QKV 954 -> 966; GRIP_GEMM_SD=0 = the branchy form (developer A/B)
    static const bool sd = !(getenv("GRIP_GEMM_SD") && atoi(getenv("GRIP_GEMM_SD")) == 0);
#define GRIP_GEMM_CASE_SD(E, MODE)                                                                                          \
    {                                                                                                                       \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_k64p_kernel<E, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_k64p_kernel<E, MODE, true>), grid, block, lds, s, a, tiles_m, tiles_n, colgroup);                    \
        GRIP_CHECK_HIP(hipGetLastError());                                                                                  \
        return GRIP_OK;                                                                                                     \
    }
    if (sd && epi == EPI_LNFOLD_F16 && emode == 4) GRIP_GEMM_CASE_SD(EPI_LNFOLD_F16, 4)
    if (sd && epi == EPI_LNFOLD_GELU_F16 && emode == 2) GRIP_GEMM_CASE_SD(EPI_LNFOLD_GELU_F16, 2)
    if (sd && epi == EPI_BIAS_RESID_STATS && emode == 1) GRIP_GEMM_CASE_SD(EPI_BIAS_RESID_STATS, 1)
#undef GRIP_GEMM_CASE_SD
    if (emode != 0 && (epi == EPI_LNFOLD_F16 || epi == EPI_LNFOLD_GELU_F16 || epi == EPI_BIAS_RESID_STATS)) {
        if (emode == 4 && epi != EPI_BIAS_RESID_STATS) {
            switch (epi) {
                GRIP_GEMM_CASE_M(EPI_LNFOLD_F16, 4)
                GRIP_GEMM_CASE_M(EPI_LNFOLD_GELU_F16, 4)
            }
        } else if (emode == 2) {
            switch (epi) {
                GRIP_GEMM_CASE_M(EPI_LNFOLD_F16, 2)
                GRIP_GEMM_CASE_M(EPI_LNFOLD_GELU_F16, 2)
            }
        } else {
            switch (epi) {
                GRIP_GEMM_CASE_M(EPI_LNFOLD_F16, 1)
                GRIP_GEMM_CASE_M(EPI_LNFOLD_GELU_F16, 1)
                GRIP_GEMM_CASE_M(EPI_BIAS_RESID_STATS, 1)
            }
        }
        GRIP_CHECK_HIP(hipGetLastError());
        return GRIP_OK;
    }
#undef GRIP_GEMM_CASE_M
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_F16)
        GRIP_GEMM_CASE(EPI_GELUGRAD_F16)
        GRIP_GEMM_CASE(EPI_F32_SCALE)
        GRIP_GEMM_CASE(EPI_LNFOLD_F16)
        GRIP_GEMM_CASE(EPI_LNFOLD_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID_STATS)
        default: GRIP_REQUIRE(false, "gemm: unknown epilogue %d", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// variant: 0 = choose, 1 = 128x128x64 (2-stage), 2 = 256x256x32 (4-stage ring), 3 = 256x128x32 (3-stage ring), 4 = 64x128x64 (2-stage),
//          5 = 256x256x64 (2-stage, whole-line DMA), 6 = the same, persistent
// Split-K factor for an EPI_F32 product whose output has too few 64x128 tiles while K is long (the input-gradient GEMMs of
// the prompt steps: 136 tiles x 32 k-steps at M = 2 142, N = 512, K = 2 048): the SMALLEST factor that puts a workgroup on every
// CU (>= 256 workgroups) with >= 4 k-steps each, and at least 2 below 512 tiles.  Measured (tools/small_gemm_bench.py,
// SWEEP=1): text 15.5 us unsplit -> 11.7 at 2 = 11.7 at 4; image (324 tiles) 30.7 -> 24.7 at 2, 26.7 at 4 -- beyond one
// workgroup per CU more partials only add traffic for the consumer (ln_bwd_add reads every partial).
// Cooperative split-K of an epilogue-carrying GEMM (GemmArgs::coop_scratch): worth it when a handful of 64-row tiles walk a long K each -- every tile
// stages its slices at the ~80 GB/s one CU pulls, whatever the other 200 CUs do.  The largest factor of {2, 4} (GRIP_COOP_SPLIT: developer A/B, 1 = off)
// that leaves every split >= 4 slices and the launch <= 256 workgroups.
int gemm_pick_coop_split(int M, int N, int K) {
    static const int force = getenv("GRIP_COOP_SPLIT") ? atoi(getenv("GRIP_COOP_SPLIT")) : 0;
    static const bool wspec = !(getenv("GRIP_GEMM_WSPEC") && atoi(getenv("GRIP_GEMM_WSPEC")) == 0);
    if (!wspec) return 1;                          // the form lives in the loader-wave kernel only
    const int64_t tiles = ((int64_t)((M + 63) / 64) * (N / BN) + 7) / 8 * 8;
    const int nk = K / BK;
    if (N % BN || K % BK || nk < 16 || tiles > 64) return 1;
    if (force >= 1) return (nk % force == 0 && nk / force >= 3 && tiles * force <= 256) ? force : 1;
    int best = 1;
    for (int f : {2, 4})
        if (nk % f == 0 && nk / f >= 4 && tiles * f <= 256) best = f;
    return best;
}

int gemm_pick_ksplit(int M, int N, int K) {
    static const int force = getenv("GRIP_GEMM_KSPLIT") ? atoi(getenv("GRIP_GEMM_KSPLIT")) : 0;   // developer A/B: 1 disables, n forces
    const int64_t tiles = (int64_t)((M + 63) / 64) * (N / BN);
    const int nk = K / BK;
    if (force >= 1) return nk % force == 0 ? force : 1;
    if (tiles >= 512) return 1;
    // More 64-row tiles than CUs (the image tower's input-gradient GEMMs: M = 3 408, N = 768 -> 324): 128-row tiles instead, split so that
    // about two of their 64-KiB workgroups share a CU -- the largest factor with <= 512 workgroups and >= 8 K tiles each.  VPT step in situ
    // (tools/exp_r03_5.sh): 2 x 324 workgroups of 64x128 23.2 us, 2 x 162 of 128x128 23.2 us, 3 x 162 of 128x128 20.0 us (the third partial
    // costs ln_bwd_add 1.5 us: 10.0 -> 11.5).
    {
        const int64_t t128 = (int64_t)((M + 127) / 128) * (N / BN);
        if (tiles > 256 && t128 <= 256) {
            int f128 = 1;
            for (int f : {2, 3, 4})
                if (nk % f == 0 && nk / f >= 8 && t128 * f <= 512) f128 = f;
            if (f128 > 1) return f128;
        }
    }
    int best = 1;
    const int64_t t32 = (int64_t)((M + 31) / 32) * (N / BN);
    for (int f : {2, 3, 4, 6, 8}) {
        if (nk % f || nk / f < 3) continue;
        best = f;                                   // a few dozen tiles (the shared-prefix text rows): as many K slices as keep 3 k-steps each
        // workgroups of the launch: the launcher runs small launches on 32-row tiles (r06, launch_gemm_impl), which doubles them -- so half the splits already
        // reach (nearly) every CU: M = 425, N = 512, K = 1 536 / 2 048: 4 splits x 56 tiles = 224 workgroups walking 6 - 8 slices instead of 8 x 28 walking 3 - 4,
        // and ln_bwd_add sums 4 partials instead of 8 (graphed CoOp step 1.075 -> 1.041 ms, profiles/r06_ksplit_ab.txt)
        const int64_t wgs = (tiles * f <= 128 && t32 * f <= 256) ? t32 * f : tiles * f;
        if (nk / f >= 4 && wgs >= 224) break;       // the smallest factor that reaches (7/8 of) every CU
    }
    return best;
}

static int launch_ring(int epi, const GemmArgs& a, int nst, dim3 grid, hipStream_t s) {
    const int tiles_m = (a.M + 63) / 64, tiles_n = a.N / BN;
    const size_t lds = (size_t)nst * (64 + BN) * BK * 2;
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<E, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (64 + BN) * BK * 2)); \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<E, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (64 + BN) * BK * 2)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        if (nst == 3) hipLaunchKernelGGL((gemm_ring_kernel<E, 3>), grid, dim3(256), lds, s, a, tiles_m, tiles_n);           \
        else hipLaunchKernelGGL((gemm_ring_kernel<E, 4>), grid, dim3(256), lds, s, a, tiles_m, tiles_n);                    \
    } break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_F16)
        GRIP_GEMM_CASE(EPI_GELUGRAD_F16)
        GRIP_GEMM_CASE(EPI_F32_SCALE)
        GRIP_GEMM_CASE(EPI_LNFOLD_F16)
        GRIP_GEMM_CASE(EPI_LNFOLD_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID_STATS)
        default: GRIP_REQUIRE(false, "gemm: unknown epilogue %d", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// gemm_ringw_kernel: WMF = 2 (64-row tiles) runs a 4-slot ring, WMF = 4 (128-row tiles) a 3-slot ring for short K walks and a 5-slot one
// (the whole LDS) for long ones
template <int WMF, int NST>
static int launch_ringw(int epi, const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int BMT = 32 * WMF;
    const int tiles_m = (a.M + BMT - 1) / BMT, tiles_n = a.N / BN;
    constexpr size_t lds = (size_t)NST * (BMT + BN) * BK * 2;
    // the ring's prologue fills NST - 1 slots: the K walk (of one split) must be at least that long
    GRIP_REQUIRE((a.K / BK) / (a.ksplit > 1 ? a.ksplit : 1) >= NST - 1, "gemm_ringw: K walk of %d tiles is shorter than the %d-slot ring's prologue", (a.K / BK) / (a.ksplit > 1 ? a.ksplit : 1), NST);
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_ringw_kernel<E, NST, WMF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_ringw_kernel<E, NST, WMF>), grid, dim3(512), lds, s, a, tiles_m, tiles_n);                  \
    } break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_F16)
        GRIP_GEMM_CASE(EPI_GELUGRAD_F16)
        GRIP_GEMM_CASE(EPI_F32_SCALE)
        GRIP_GEMM_CASE(EPI_LNFOLD_F16)
        GRIP_GEMM_CASE(EPI_LNFOLD_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID_STATS)
        default: GRIP_REQUIRE(false, "gemm: unknown epilogue %d", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// The 96-row form of the loader-wave ring (WMF = 3; r06): residual and plain f16 epilogues only (the LayerNorm-folded ones fetch their row statistics with a
// power-of-two lane map).  For the K = 4 d residual GEMM of an image-tower prompt step -- M = 3 408, N = 768: 27 x 6 = 162 tiles of 128 rows leave 94 CUs idle for the whole
// 48-slice walk; 36 x 6 = 216 tiles of 96 rows put 84 % of the chip on a walk that is a quarter shorter per tile.
static int launch_ringw96(int epi, const GemmArgs& a, hipStream_t s) {
    constexpr int NST = 5, WMF = 3, BMT = 96;
    const int tiles_m = (a.M + BMT - 1) / BMT, tiles_n = a.N / BN;
    constexpr size_t lds = (size_t)NST * (BMT + BN) * BK * 2;
    GRIP_REQUIRE((int64_t)tiles_m * BMT <= a.m_pad && a.K / BK >= NST - 1 && a.ksplit <= 1, "gemm_ringw96: shape (M=%d K=%d m_pad=%lld ksplit=%d)", a.M, a.K, (long long)a.m_pad, a.ksplit);
    dim3 grid(tiles_m * tiles_n);
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E: {                                                                                                               \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_ringw_kernel<E, NST, WMF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((gemm_ringw_kernel<E, NST, WMF>), grid, dim3(512), lds, s, a, tiles_m, tiles_n);                  \
    } break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_BIAS_RESID_STATS)
        GRIP_GEMM_CASE(EPI_F16)
        default: GRIP_REQUIRE(false, "gemm_ringw96: residual / plain f16 epilogues only (epi %d)", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

static int launch_gemm_impl(int epi, const GemmArgs& a_in, hipStream_t s, int* chosen) {
    // Row-dependent K rotation (GemmArgs::rot_rows = the caller's permission: train-mode launches only).  In a prompt step every GEMM
    // reads weights nobody has touched since the previous step; with all tile rows of a column panel walking K in lockstep, each of them
    // waits out the memory latency of every slice.  Staggered, a slice is fetched by one tile row and found in the L2 by the next: VPT
    // step in situ (rocprofv3, tools/exp_r03_5.sh), GEMM time per step 2 245 us -> 2 025 us with a stride of 2 slices per tile row
    // (1: 2 050, 3: 2 057, 5: 2 055, 11: 2 053); the split-K launches like 1 best (their walks are half or a third as long).
    // GRIP_KROT_M: developer A/B (-1 = off, n > 0 = stride n).
    static const int rot_m = getenv("GRIP_KROT_M") ? atoi(getenv("GRIP_KROT_M")) : 0;
    GemmArgs a = a_in;
    a.rot_rows = (a.rot_rows && !a.f32 && rot_m >= 0) ? (rot_m > 0 ? rot_m : (a.ksplit > 1 ? 1 : 2)) : 0;
    if (a.f32 == 2) {   // split-f16 tier (gemm_split.hip); profiler variant 7
        *chosen = 7;
        return launch_gemm_split(epi, a, s);
    }
    if (a.f32) {        // exact mode: f32 operands (gemm_f32.hip); profiler variant 0
        *chosen = 0;
        return launch_gemm_f32(epi, a, s);
    }
    if (epi == EPI_BIAS_RESID && a.stat_part) epi = EPI_BIAS_RESID_STATS;
    GRIP_REQUIRE(epi != EPI_BIAS_RESID_STATS || (a.stat_part && a.N % 64 == 0), "gemm: row statistics need stat_part and N %% 64 == 0");
    GRIP_REQUIRE(a.N % BN == 0 && a.K % BK == 0 && a.M > 0, "gemm: need N %% 128 == 0 and K %% 64 == 0 (M=%d N=%d K=%d)", a.M, a.N, a.K);
    GRIP_REQUIRE(a.ldc % 4 == 0, "gemm: ldc %% 4 != 0");
    GRIP_REQUIRE(((int64_t)a.M + 256) * a.ldc < ((int64_t)1 << 31), "gemm: output larger than 2^31 elements (M=%d ldc=%d)", a.M, a.ldc);
    const int64_t m256 = (int64_t)((a.M + 255) / 256) * 256;
    const bool can_big = a.m_pad >= m256 && a.K >= 4 * BK2;       // A must be padded to the 256-row tile
    int variant = a.variant;
    if (variant == 0) {
        // pick the tile shape by (measured relative rate) x (fill of the last wave of workgroups over
        // 256 CUs): 256x256 runs one workgroup per CU, the other two shapes two per CU.
        auto fill = [](int64_t tiles, int64_t slots) { return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots); };
        const int64_t tm128 = (a.M + 127) / 128, tm256 = m256 / 256;
        double best = 0.85 * fill(tm128 * (a.N / 128), 512);
        variant = 1;
        {
            double s4 = 0.70 * fill((int64_t)((a.M + 63) / 64) * (a.N / 128), 768);   // three 48-KiB workgroups per CU
            // fewer 128-row tiles than CUs: the launch is one workgroup per CU whatever the shape, and its time is k-steps x (bytes
            // a CU stages per step), which the 64-row tile cuts by a quarter (measured at M = 425 and 2 142: 6-25 % faster)
            if (tm128 * (a.N / 128) <= 256) s4 = 1.0;
            if (s4 > best) { best = s4; variant = 4; }
            // ... unless the 64-row tiles outnumber the CUs while the 128-row ones do not: one round of 128-row tiles on dedicated loader
            // waves (gemm_ringw_kernel) beats a round and a bit of 64-row ones (M = 3 408, N = 768, K = 3 072: 25.6 us against 31)
            static const bool r128 = !(getenv("GRIP_GEMM_R128") && atoi(getenv("GRIP_GEMM_R128")) == 0);     // developer A/B
            if (r128 && variant == 4 && a.ksplit <= 1 && tm128 * (a.N / 128) <= 256 && (int64_t)((a.M + 63) / 64) * (a.N / 128) > 256 && a.K > 12 * BK) { best = 1.0; variant = 1; }
        }
        if (can_big) {
            const double s3 = 0.93 * fill(tm256 * (a.N / 128), 512);
            if (s3 > best) { best = s3; variant = 3; }
            if (a.N % 256 == 0) {
                const double s2 = 1.0 * fill(tm256 * (a.N / 256), 256);
                // same tile, two feeds: the 64-wide two-stage kernel (whole-line DMA) is 2-6 % faster than the 32-wide ring
                // ... and persistent (one workgroup per CU walking its XCD's tiles) once every CU gets several tiles
                static const int force = getenv("GRIP_GEMM_BIG") ? atoi(getenv("GRIP_GEMM_BIG")) : 0;    // developer A/B: 2, 5 or 6
                if (s2 > best) { best = s2; variant = force ? force : (a.K < 2 * BK ? 2 : (tm256 * (a.N / 256) >= 512 ? 6 : 5)); }
                // fewer tiles than CUs: one tile-time whatever the tile holds, so the 192-row form of the same kernel when it fills more CUs
                const int64_t tm192 = (a.M + 191) / 192;
                if (variant == 5 && !force && tm192 * 192 <= a.m_pad && tm192 * (a.N / 256) <= 256 && epi != EPI_BIAS_RESID_STATS &&
                    fill(tm192 * (a.N / 256), 256) > best) { best = fill(tm192 * (a.N / 256), 256); variant = 8; }
            }
        }
    }
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    const bool coop = ksplit > 1 && (epi == EPI_BIAS_RESID || epi == EPI_BIAS_RESID_STATS);
    if (coop) {     // cooperative split-K: loader-wave kernel on 64-row tiles only (GemmArgs::coop_scratch)
        const int64_t t64 = (int64_t)((a.M + 63) / 64) * (a.N / BN);
        GRIP_REQUIRE(a.coop_scratch && a.coop_counter && (a.K / BK) % ksplit == 0 && (a.K / BK) / ksplit >= 3 && ((t64 + 7) / 8 * 8) * ksplit <= 256,
                     "gemm: cooperative split-K needs the scratch and counter buffers, (K/64) %% ksplit == 0 with >= 3 slices per split and <= 256 workgroups (M=%d N=%d K=%d ksplit=%d)",
                     a.M, a.N, a.K, ksplit);
        variant = 4;
    } else if (ksplit > 1) {
        GRIP_REQUIRE(epi == EPI_F32 && (a.K / BK) % ksplit == 0 && a.split_stride >= (int64_t)a.M * a.ldc,
                     "gemm: split-K needs EPI_F32, (K/64) %% ksplit == 0 and a partial stride >= M*ldc (K=%d ksplit=%d)", a.K, ksplit);
        // 128-row tiles where gemm_pick_ksplit sized the split for them (more 64-row tiles than CUs: about two 128-row workgroups per CU)
        const int64_t t64 = (int64_t)((a.M + 63) / 64) * (a.N / BN), t128 = (int64_t)((a.M + 127) / 128) * (a.N / BN);
        if (a.variant == 0 && t64 > 256 && t128 * ksplit <= 512) variant = 1;
        else if (variant != 1) variant = 4;
    }
    if (variant == 5 && epi == EPI_BIAS_RESID_STATS) variant = 6;   // the one-tile-per-workgroup 64-wide kernel has no registers left for the statistics
    *chosen = variant;
    // At most one workgroup per CU: the ring with the feed on its own waves (gemm_ringw_kernel).  GRIP_GEMM_WSPEC=0: developer A/B.
    static const bool wspec = !(getenv("GRIP_GEMM_WSPEC") && atoi(getenv("GRIP_GEMM_WSPEC")) == 0);
    GRIP_REQUIRE(a.stat_parts <= 0 || a.stat_in, "gemm: stat_parts without stat_in");      // (every kernel but the persistent one reads the partial sums itself: row_stat)
    if (a.stat_parts > 0 && variant == 6) {      // the persistent kernel (pool-sized M) takes finalised statistics only
        GRIP_REQUIRE(a.rowstat, "gemm: partial row sums on the persistent kernel need a rowstat buffer to finalise into");
        const int rc = launch_ln_stats_finalize(a.stat_in, a.stat_parts, const_cast<float*>(a.rowstat), a.M, a.K, s);
        if (rc) return rc;
        a.stat_parts = 0;
    }
    if (variant == 2) {
        GRIP_REQUIRE(can_big && a.N % 256 == 0, "gemm: 256x256 tile needs N %% 256 == 0 and A padded to 256 rows");
        return launch_big<256, 256, 4>(epi, a, s);
    }
    if (variant == 3) {
        GRIP_REQUIRE(can_big, "gemm: 256x128 tile needs A padded to 256 rows");
        return launch_big<256, 128, 3>(epi, a, s);
    }
    if (variant == 5) {
        GRIP_REQUIRE(can_big && a.N % 256 == 0 && a.K >= 2 * BK, "gemm: 256x256x64 tile needs N %% 256 == 0, K >= 128 and A padded to 256 rows");
        return launch_k64<8>(epi, a, s);
    }
    if (variant == 6) {
        GRIP_REQUIRE(can_big && a.N % 256 == 0 && a.K >= 2 * BK, "gemm: 256x256x64 tile needs N %% 256 == 0, K >= 128 and A padded to 256 rows");
        return launch_k64p(epi, a, s);
    }
    {
        static const bool k64w = getenv("GRIP_K64W") && atoi(getenv("GRIP_K64W")) != 0;     // developer A/B
        if (k64w && variant == 8 && (epi == EPI_BIAS_GELU_F16 || epi == EPI_GELUGRAD_F16)) variant = 9;
    }
    if (variant == 9) {      // developer prototype: 192x256 tile with loader waves
        GRIP_REQUIRE(a.N % 256 == 0 && a.K >= 2 * BK && (int64_t)((a.M + 191) / 192) * 192 <= a.m_pad, "gemm: 192x256x64 loader-wave tile: shape");
        return launch_k64w(epi, a, s);
    }
    if (variant == 8) {      // (7 is the f32 kernel in the debug hook)
        GRIP_REQUIRE(a.N % 256 == 0 && a.K >= 2 * BK && (int64_t)((a.M + 191) / 192) * 192 <= a.m_pad && epi != EPI_BIAS_RESID_STATS,
                     "gemm: 192x256x64 tile needs N %% 256 == 0, K >= 128, A padded to a multiple of 192 rows and an epilogue without row statistics");
        return launch_k64<8, 6>(epi, a, s);
    }
    const int bmt = variant == 4 ? 64 : 128;
    const int tiles_m = (a.M + bmt - 1) / bmt, tiles_n = a.N / BN;
    dim3 grid(tiles_m * tiles_n, ksplit), block(256);
    if (coop) {
        GRIP_REQUIRE(wspec, "gemm: cooperative split-K needs the loader-wave kernels (GRIP_GEMM_WSPEC=0 is set)");
        grid.x = (grid.x + 7) / 8 * 8;       // the splits of a tile on one XCD (gemm_ringw_kernel)
        return launch_ringw<2, 4>(epi, a, grid, s);
    }
    {   // the 96-row loader-wave tile also for SHORT walks whose 64-row tiles outnumber the CUs (M = 3 408, N = K = 768 -- the out-proj forward of an image-tower
        // prompt step and its input gradient: 324 tiles of 64 rows at two workgroups per CU against 216 of 96 rows at one: VPT step 2.82 -> 2.78 ms, UPT
        // 3.15 -> 3.12, profiles/r06_r96_ab.txt).  GRIP_GEMM_R96 (developer A/B): 0 = no 96-row tiles, 1 = long walks only, 2 (default) = both
        static const int r96mode = getenv("GRIP_GEMM_R96") ? atoi(getenv("GRIP_GEMM_R96")) : 2;
        const int64_t t96 = (int64_t)((a.M + 95) / 96) * (a.N / BN), t64 = (int64_t)((a.M + 63) / 64) * (a.N / BN);
        if (r96mode >= 2 && wspec && ksplit == 1 && a.variant == 0 && variant == 4 && (epi == EPI_BIAS_RESID || epi == EPI_BIAS_RESID_STATS || epi == EPI_F16) &&
            t96 <= 256 && t64 > 256 && (int64_t)((a.M + 95) / 96) * 96 <= a.m_pad && a.K / BK >= 4)
            return launch_ringw96(epi, a, s);
    }
    if (wspec && (int64_t)grid.x * ksplit <= 256) {
        const int nk = a.K / BK / ksplit;
        {   // 32-row tiles (WMF = 1; r06) where the 64-row ones fill at most half the chip -- the text tower's M = 425 GEMMs of a CoOp step: 28 - 112 tiles of 64 rows
            // become 56 - 224 of 32 rows, each staging 160 instead of 192 rows per K slice through its CU's LDS-DMA path: graphed CoOp step 1.11 - 1.16 -> 1.076 ms
            // (profiles/r06_r32_ab.txt).  Same products in the same order: bit-identical (tests/test_gpu_kernels.py).  GRIP_GEMM_R32=0: developer A/B
            static const bool r32 = !(getenv("GRIP_GEMM_R32") && atoi(getenv("GRIP_GEMM_R32")) == 0);
            const int64_t t32 = (int64_t)((a.M + 31) / 32) * (a.N / BN);
            // (split-K input-gradient GEMMs too -- EPI_F32 partials, one grid row per split -- as long as all splits of the 32-row tiles still fit one per CU)
            if (r32 && variant == 4 && nk >= 3 && !coop && (ksplit == 1 || epi == EPI_F32) && a.variant == 0 && (int64_t)grid.x * ksplit <= 128 && t32 * ksplit <= 256) {
                dim3 g32((unsigned)t32, (unsigned)ksplit);
                return launch_ringw<1, 4>(epi, a, g32, s);
            }
        }
        if (variant == 4 && nk >= 3) return launch_ringw<2, 4>(epi, a, grid, s);
        if (variant == 1 && nk > 12 && ksplit == 1 && a.variant == 0 && (epi == EPI_BIAS_RESID || epi == EPI_BIAS_RESID_STATS)) {
            // long walk, residual epilogue, fewer 128-row tiles than CUs: the 96-row form when it fills more of them (and its rows are allocated)
            static const bool r96 = !(getenv("GRIP_GEMM_R96") && atoi(getenv("GRIP_GEMM_R96")) == 0);     // developer A/B
            const int64_t t96 = (int64_t)((a.M + 95) / 96) * (a.N / BN);
            if (r96 && t96 <= 256 && t96 > (int64_t)grid.x && (int64_t)((a.M + 95) / 96) * 96 <= a.m_pad) return launch_ringw96(epi, a, s);
        }
        if (variant == 1 && nk > 12) return launch_ringw<4, 5>(epi, a, grid, s);
        if (variant == 1 && nk >= 2) return launch_ringw<4, 3>(epi, a, grid, s);
    }

    if (variant == 4) {
        // ring depth by workgroups per CU: <= 1 -> four stages (96 KiB), <= 2 -> three (72 KiB, two per CU); beyond that three
        // co-resident two-stage workgroups already keep three tiles in flight per CU
        static const int force = getenv("GRIP_GEMM_RING") ? atoi(getenv("GRIP_GEMM_RING")) : -1;    // developer A/B: 0 (off), 3, 4
        const int64_t wgs = (int64_t)grid.x * ksplit;
        const int nk = a.K / BK / ksplit;
        int nst = force >= 0 ? force : (wgs <= 256 ? 4 : (wgs <= 512 ? 3 : 0));
        if (nst && nk < nst - 1) nst = 0;
        if (nst) return launch_ring(epi, a, nst, grid, s);
    }
#define GRIP_GEMM_CASE(E)                                                                                  \
    case E:                                                                                                \
        if (variant == 4) hipLaunchKernelGGL((gemm_f16_kernel<E, 2>), grid, block, 0, s, a, tiles_m, tiles_n); \
        else hipLaunchKernelGGL((gemm_f16_kernel<E, 4>), grid, block, 0, s, a, tiles_m, tiles_n);          \
        break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        GRIP_GEMM_CASE(EPI_F16)
        GRIP_GEMM_CASE(EPI_GELUGRAD_F16)
        GRIP_GEMM_CASE(EPI_F32_SCALE)
        GRIP_GEMM_CASE(EPI_LNFOLD_F16)
        GRIP_GEMM_CASE(EPI_LNFOLD_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID_STATS)
        default: GRIP_REQUIRE(false, "gemm: unknown epilogue %d", epi);
    }
#undef GRIP_GEMM_CASE
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
