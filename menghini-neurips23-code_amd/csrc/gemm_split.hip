// Split-f16 GEMM (dims.precision = 2, the MIDDLE tier of screen-and-refine): C[M,N] = epi(A[M,K] * W[N,K]^T) to ~22 mantissa bits
// on the f16 matrix pipes.  Every operand element x is carried as two f16 numbers
//     hi = f16(x),   lo = f16(x - hi)                       (x - hi is exact in f32)
// and the product is formed from THREE v_mfma_f32_16x16x32_f16 per fragment pair into ONE f32 accumulator:
//     acc += a_hi w_hi + a_hi w_lo + a_lo w_hi                                             (dropped: a_lo w_lo, 2^-22 relative)
// so a term's relative error is ~3 x 2^-23 against f32's 2^-24: the deviations of the tower's probabilities from the f32 twin's are
// of the size two f32 summation orders differ by (measured: DESIGN.md 10.2), 100x below the f16 towers', at a third of the f16 MFMA
// rate instead of a sixteenth (v_mfma_f32_16x16x4_f32).  The reference decides on fp32 values (utils/clip_pseudolabels.py:38-41,
// 73-101); this tier only SCREENS for it more finely -- whatever it cannot decide still goes to the f32 tower.
//
// THE DEFAULT BUILD (GRIP_SPLIT_LO_SCALE == 1, `gemm_split1_kernel` below) keeps the lo parts UNSCALED, so they are often f16 SUBNORMALS:
// it relies on gfx950's f16 MFMA taking subnormal inputs unflushed (measured, tools/micro/mfma_denorm.hip; this file is built for gfx950
// only -- the #error below).  To keep the lo part of a typical weight (|w| ~ 0.02) normal, WEIGHTS are stored scaled by 2^8 (SP1_W_SCALE,
// exact) and the epilogue multiplies by 2^-8; a weight with |w| >= 255.9 would overflow its hi part: grip_tower_finalize checks the range
// and fails (launch_split_rows' overflow flag).  Activations are unscaled; an activation lo below the f16 normal range is quantised to
// 2^-24 absolute.  The developer variant GRIP_SPLIT_LO_SCALE = 2048 (`gemm_split_kernel`: lo' = f16((x - hi) * 2^11), two accumulators,
// 256 x 128 tile, nothing subnormal where hi is normal) is kept for A/B; it measured 10 - 15 % slower.
//
// Data layout ("split layout"), activations and weights alike: row-major, per row and per group of 32 consecutive k one 128-byte
// line [32 x hi | 32 x lo] -- the row pitch equals an f32 row's (4 K bytes), a K step of the GEMM moves whole cache lines, and
// the LDS image of a stage is byte for byte gemm_f32.hip's (128-byte rows, 16-byte chunk index XOR (row & 7) on the DMA source
// address and on the ds_read_b128 address: conflict-free).  A lane's MFMA fragment (8 consecutive k of one row) is chunk
// (lane >> 4) of the row's hi half and chunk 4 + (lane >> 4) of its lo half: one ds_read_b128 each.
//
// Kernels: default = 256 x 256 tile, 8 waves as 4 x 2 (64 x 128 per wave), two 64-KiB stages (see `gemm_split1_kernel`); scaled variant =
// 256 x 128 tile, 8 waves as 4 x 2, K staged 32 wide in a 3-slot ring of 48 KiB stages with counted vmcnt.  Operands are swapped
// (W fragment first) so a lane ends with four consecutive output columns of one row.
#include <math.h>

#include "common.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "gemm_split.hip relies on gfx950 MFMA semantics (unflushed f16 subnormal inputs); build with --offload-arch=gfx950 only"
#endif

#define SPL_BM 256
#define SPL_BN 128
#define SPL_ROWB 128                                   // bytes per LDS row: 32 hi + 32 lo' halfs
#define SPL_STAGE_B ((SPL_BM + SPL_BN) * SPL_ROWB)     // 48 KiB
#define SPL_NST 3
#ifndef SPL_PIPE
#define SPL_PIPE 1          // 1 = register-pipelined K loop (hi fragments of the next step prefetched), 0 = all 16 reads at the top of the step
#endif
#define SPL_LO_INV (1.0f / (float)GRIP_SPLIT_LO_SCALE)

// QuickGELU x * sigmoid(1.702 x) on v_exp_f32 + v_rcp_f32 (1 ulp each: far inside this tier's 2^-22; the f32 tower keeps IEEE exp and division)
__device__ __forceinline__ float quick_gelu_exact_s(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x)); }

// (hi / lo' of a value: split_f16x4 / store4(SplitRow, ...) in common.h -- the LayerNorm, the f32 attention and the GELU epilogue below
// all write the layout through it)

#if GRIP_SPLIT_LO_SCALE != 1
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_split_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    // XCD-aware, bijective tile remap (block b runs on XCD b % 8): every XCD owns a contiguous run of tiles, N-fastest
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * SPL_BM, n0 = tn * SPL_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // staging: wave w fills rows [w*32, +32) of the A tile (4 pieces of 8 rows x 128 B) and rows [w*16, +16) of the W tile (2 pieces)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const size_t pitch = (size_t)g.K * 4;              // bytes per operand row (K/32 lines of 128 B)
    const char* a_src = (const char*)g.A + (size_t)(m0 + wave * 32) * pitch;
    const char* w_src = (const char*)g.W + (size_t)(n0 + wave * 16) * pitch;
    const uint32_t lane_off = (uint32_t)srow * (uint32_t)pitch + (uint32_t)(schunk * 16);
    auto stage = [&](int slot, int kt) {
        char* abase = lds + slot * SPL_STAGE_B + wave * 32 * SPL_ROWB;
        char* bbase = lds + slot * SPL_STAGE_B + SPL_BM * SPL_ROWB + wave * 16 * SPL_ROWB;
        const char* as = a_src + (size_t)kt * SPL_ROWB;
        const char* ws = w_src + (size_t)kt * SPL_ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)(as + (size_t)i * 8 * pitch + lane_off), (AS3 void*)(abase + i * 8 * SPL_ROWB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const AS1 void*)(ws + (size_t)i * 8 * pitch + lane_off), (AS3 void*)(bbase + i * 8 * SPL_ROWB), 16, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int swz_hi = (fgrp ^ (lane & 7)) * 16, swz_lo = ((4 + fgrp) ^ (lane & 7)) * 16;
    const int a_row = (wr * 64 + frow) * SPL_ROWB;
    const int b_row = SPL_BM * SPL_ROWB + (wc * 64 + frow) * SPL_ROWB;

    f32x4 acc[4][4], cor[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    const int nk = g.K / 32;
    // K rotation per column panel (as in gemm.hip): the column tiles of one row panel run side by side on an XCD and would all ask for the same
    // slice of their shared A panel at the same moment; tile (., tn) walks its slices cyclically from slice tn * nk / tiles_n, so a slice is
    // fetched by one tile and found in the L2 by the others.  The start depends on the column panel only: a row's result does not depend on the launch.
    int ks = (int)(((int64_t)tn * nk) / tiles_n);
    auto next_slice = [&](int k) { return k + 1 == nk ? 0 : k + 1; };
    int ks_stage = ks;                      // slice the next DMA stage fetches
    stage(0, ks_stage);
    ks_stage = next_slice(ks_stage);
    stage(1, ks_stage);                     // (nk == 1: the same slice again, into a slot nobody reads)
    ks_stage = next_slice(ks_stage);
#if SPL_PIPE
    // Software pipeline over the K steps: the hi fragments of step kt + 1 are read from LDS while step kt multiplies, so a step starts on operands
    // that are already in registers (two hi sets of 32 + one lo set of 32 fragment registers beside the 128 accumulators) and its 8 lo reads land
    // under the 16 main products.  Costs DMA distance: stage kt + 1 must be visible at the top of step kt (one step of prefetch instead of two;
    // with the K rotation most slices come out of the L2).  The loop is unrolled by two so the two hi sets swap roles without copies (K % 64 == 0);
    // sched_barrier pins the intended order: all 16 reads, then the 48 MFMAs with one DMA piece of stage kt + 2 after every sixth.
    half8 ah0[4], wh0[4], ah1[4], wh1[4];
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) ah0[i] = *(const half8*)(lds + a_row + i * 16 * SPL_ROWB + swz_hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) wh0[j] = *(const half8*)(lds + b_row + j * 16 * SPL_ROWB + swz_hi);
    // (a use of the prologue's fragments in front of the loop: otherwise the loop head inherits "reads pending" from this path and every step
    // would wait for all of its own 16 reads before its first product)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah0[i]), "v"(wh0[i]));
    auto step = [&](int kt, half8 (&ah)[4], half8 (&wh)[4], half8 (&ahn)[4], half8 (&whn)[4]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of stage kt + 1 (issued one step ago)
        __builtin_amdgcn_s_barrier();                             // stage kt + 1 visible; slot (kt + 2) % 3 fully read
        const char* st = lds + (kt % SPL_NST) * SPL_STAGE_B;
        const char* sn = lds + ((kt + 1) % SPL_NST) * SPL_STAGE_B;
        half8 al[4], wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wl[j] = *(const half8*)(st + b_row + j * 16 * SPL_ROWB + swz_lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) al[i] = *(const half8*)(st + a_row + i * 16 * SPL_ROWB + swz_lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) ahn[i] = *(const half8*)(sn + a_row + i * 16 * SPL_ROWB + swz_hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) whn[j] = *(const half8*)(sn + b_row + j * 16 * SPL_ROWB + swz_hi);
        __builtin_amdgcn_sched_barrier(0);
        char* abase = lds + ((kt + 2) % SPL_NST) * SPL_STAGE_B + wave * 32 * SPL_ROWB;
        char* bbase = lds + ((kt + 2) % SPL_NST) * SPL_STAGE_B + SPL_BM * SPL_ROWB + wave * 16 * SPL_ROWB;
        const char* as = a_src + (size_t)ks_stage * SPL_ROWB;
        const char* ws = w_src + (size_t)ks_stage * SPL_ROWB;
        ks_stage = next_slice(ks_stage);
#pragma unroll
        for (int q = 0; q < 48; ++q) {        // q = 0..15 main (j-major), 16..31 w_lo a_hi, 32..47 w_hi a_lo
            const int i = q < 32 ? (q & 3) : ((q >> 2) & 3), j = q < 32 ? ((q >> 2) & 3) : (q & 3);
            if (q < 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
            else if (q < 32) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah[i], cor[i][j], 0, 0, 0);
            else cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], al[i], cor[i][j], 0, 0, 0);
            if (q % 6 == 5 && q < 36) {
                const int pc = q / 6;
                if (pc < 4) __builtin_amdgcn_global_load_lds((const AS1 void*)(as + (size_t)pc * 8 * pitch + lane_off), (AS3 void*)(abase + pc * 8 * SPL_ROWB), 16, 0, 0);
                else __builtin_amdgcn_global_load_lds((const AS1 void*)(ws + (size_t)(pc - 4) * 8 * pitch + lane_off), (AS3 void*)(bbase + (pc - 4) * 8 * SPL_ROWB), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, ah0, wh0, ah1, wh1);
        step(kt + 1, ah1, wh1, ah0, wh0);
    }
#else
    for (int kt = 0; kt < nk; ++kt) {
        // ONE basic block per K step: stage kt has landed when at most the 6 pieces of stage kt + 1 are in flight; the barrier certifies it to the
        // other waves and frees slot (kt + 2) % 3 = (kt - 1) % 3; the 16 fragment reads go out first, in the order the MFMAs need them, and the
        // 6 DMA pieces of stage kt + 2 are issued BETWEEN the MFMAs (as a burst in front of them they cost both waves of a SIMD 6 x 60-185
        // cycles of issue at the same time).  Unconditional: past the end a valid slice is re-staged into a slot nobody reads.
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* st = lds + (kt % SPL_NST) * SPL_STAGE_B;
        half8 ah[4], al[4], wh[4], wl[4];
        wh[0] = *(const half8*)(st + b_row + swz_hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) ah[i] = *(const half8*)(st + a_row + i * 16 * SPL_ROWB + swz_hi);
#pragma unroll
        for (int j = 1; j < 4; ++j) wh[j] = *(const half8*)(st + b_row + j * 16 * SPL_ROWB + swz_hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) wl[j] = *(const half8*)(st + b_row + j * 16 * SPL_ROWB + swz_lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) al[i] = *(const half8*)(st + a_row + i * 16 * SPL_ROWB + swz_lo);
        stage((kt + 2) % SPL_NST, ks_stage);
        ks_stage = next_slice(ks_stage);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah[i], cor[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], al[i], cor[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);       // the fragment reads
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);    // 6 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    // one DMA piece
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing (dead) stages must have landed before the workgroup gives its LDS back

    // epilogue: lane (frow, fgrp) holds columns col0 + j*16 + fgrp*4 .. +3 of row row0 + i*16 + frow
    constexpr bool HAS_BIAS = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RESID);
    const int row0 = m0 + wr * 64 + frow, col0 = n0 + wc * 64 + fgrp * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) b = *(const f32x4*)(g.bias + col0 + j * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + i * 16;
            if (row >= g.M) continue;
            const int col = col0 + j * 16;
            f32x4 v = acc[i][j] + cor[i][j] * SPL_LO_INV + b;
            if constexpr (EPI == EPI_BIAS_RESID) v += *(const f32x4*)((const float*)g.resid + (size_t)row * g.ldc + col);
            if constexpr (EPI == EPI_BIAS_GELU_F16) {
                // the MLP hidden feeds the next split GEMM: written in the split layout ([32 hi | 32 lo'] per 32 columns)
                v = (f32x4){quick_gelu_exact_s(v[0]), quick_gelu_exact_s(v[1]), quick_gelu_exact_s(v[2]), quick_gelu_exact_s(v[3])};
                store4(SplitRow{(half_t*)g.out + (size_t)row * 2 * g.ldc}, col >> 2, v);
            } else {
                *(f32x4*)((float*)g.out + (size_t)row * g.ldc + col) = v;
            }
        }
    }
}

#endif   // two-accumulator form

// (r04 experiment, removed: the LDS-DMA pieces as inline asm.  Through the builtin the compiler's wait-count pass turns EVERY lgkmcnt wait of a kernel
// that contains an LDS-DMA instruction into lgkmcnt(0) -- the first MFMA of a K step then waits for all the fragment reads issued before it; with the
// pieces hidden in asm the same reads get lgkmcnt(4), (7), (10) ...  Correct, and not faster: 354 / 336 TF/s either way.  Like every restructuring of
// the f16 kernels, it lands on the ~1.05 PF/s of MFMA rate this power envelope gives a kernel that also feeds itself.)

#if GRIP_SPLIT_LO_SCALE == 1
// ---------------------------------------------------------------------------------------------------------------------------------------
// Single-accumulator form (the default): with the lo parts UNSCALED (lo = f16(x - hi); gfx950's f16 MFMA takes subnormal inputs unflushed,
// tools/micro/mfma_denorm.hip) all three products have the same scale and add into ONE accumulator, so the kernel affords the 256 x 256 tile
// of the f16 kernels (8 waves, 32 fragment pairs per wave, 128 accumulator registers): per K step of 32 a wave issues 24
// fragment reads for 96 MFMAs and the workgroup stages 64 KiB for 768 MFMAs -- a third less LDS traffic and DMA feed per MFMA than the
// 256 x 128 two-accumulator form, which measured 290 - 330 TFLOP/s f32-equivalent (0.39 of the f16 MFMA peak; an LDS / feed limit: pipelining
// its fragment reads through registers changed nothing).  Weights are stored scaled by 2^8 (exact) so that the lo part of a typical |w| ~ 0.02
// is a normal f16 (2^8 w ~ 5, lo ~ 2e-3); the epilogue multiplies the accumulator by 2^-8.  Activations are unscaled: a lo part below the
// normal range carries an absolute error <= 3e-8, far inside 2^-22 of the O(1) activations it sits among.
// Two 64-KiB LDS stages; the DMA of stage kt + 1 is issued between the MFMAs of stage kt; all 24 reads of a step go out at its top in the
// order the products need them (the MFMAs trail the reads; the exposed latency is the first read's).
#define SP1_BM 256
#define SP1_BN 256
#define SP1_STAGE_B ((SP1_BM + SP1_BN) * SPL_ROWB)     // 64 KiB
#define SP1_W_SCALE 256.0f
// WLO = false (r06, GemmArgs::w_exact): every element of W is an f16 number already (its lo part is zero) -- what the block weights of every
// published CLIP checkpoint are (fp16 archives; the reference's CPU path is clip.load(..., "cpu") = those values cast up, models/clip_encoders.py:
// 108-119) -- so the a_hi w_lo product is dropped: two MFMA passes instead of three per fragment pair, the W lo fragments are never read.
// grip_tower_finalize sets the flag when the split of the weights left no non-zero lo part.
// Wave layout (r06): the 8 waves as 4 x 2 (64 rows x 128 columns per wave = 4 x 8 fragments), not the f16 kernels' 2 x 4 (128 x 64).  Only the A side needs both
// of its halves in the two-pass form, so the A side is the short one: 4 a_hi + 4 a_lo + 8 w_hi = 16 fragment reads per K step and wave instead of 20 (three-pass:
// 24 either way).  An output element's sum does not depend on the wave that forms it (same K order, same pass order): bit-identical to the 2 x 4 layout, whole
// split tower 10.6k -> 10.95k img/s on fp16-checkpoint weights, 9.35k -> 9.53k otherwise (profiles/r06_split_wave_layout_ab.txt).
template <int EPI, bool WLO>
__global__ __launch_bounds__(512, 2) void gemm_split1_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int FI = 4, FJ = 8;          // A / W fragments (16 rows each) per wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * SP1_BM, n0 = tn * SP1_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // staging: wave w fills rows [w*32, +32) of the A tile and of the W tile (4 + 4 pieces of 8 rows x 128 B)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const size_t pitch = (size_t)g.K * 4;
    const char* a_src = (const char*)g.A + (size_t)(m0 + wave * 32) * pitch;
    const char* w_src = (const char*)g.W + (size_t)(n0 + wave * 32) * pitch;
    const uint32_t lane_off = (uint32_t)srow * (uint32_t)pitch + (uint32_t)(schunk * 16);

    const int frow = lane & 15, fgrp = lane >> 4;
    const int swz_hi = (fgrp ^ (lane & 7)) * 16, swz_lo = ((4 + fgrp) ^ (lane & 7)) * 16;
    const int a_row = (wr * FI * 16 + frow) * SPL_ROWB;
    const int b_row = SP1_BM * SPL_ROWB + (wc * FJ * 16 + frow) * SPL_ROWB;

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / 32;
    int ks_stage = (int)(((int64_t)tn * nk) / tiles_n);       // K rotation per column panel (see the two-accumulator kernel / gemm.hip)
    auto next_slice = [&](int k) { return k + 1 == nk ? 0 : k + 1; };
    {
        char* abase = lds + wave * 32 * SPL_ROWB;
        char* bbase = lds + SP1_BM * SPL_ROWB + wave * 32 * SPL_ROWB;
        const char* as = a_src + (size_t)ks_stage * SPL_ROWB;
        const char* ws = w_src + (size_t)ks_stage * SPL_ROWB;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            __builtin_amdgcn_global_load_lds((const AS1 void*)(as + (size_t)pc * 8 * pitch + lane_off), (AS3 void*)(abase + pc * 8 * SPL_ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const AS1 void*)(ws + (size_t)pc * 8 * pitch + lane_off), (AS3 void*)(bbase + pc * 8 * SPL_ROWB), 16, 0, 0);
        }
        ks_stage = next_slice(ks_stage);
    }
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage kt
        __builtin_amdgcn_s_barrier();                         // stage kt visible; the other slot fully read
        const char* st = lds + (kt & 1) * SP1_STAGE_B;
        half8 ah[FI], al[FI], wh[FJ], wl[FJ];
        // The hi fragments (12 reads) go out first; the lo reads (12, two-pass: the A side's 8 / 4) follow BETWEEN the first 24 MFMAs, which only need
        // hi operands.  (All 24 up front would exceed the 4-bit lgkmcnt: the compiler then waits for every read before the first product -- 192
        // ds_read_b128 per workgroup and K step in front of idle matrix pipes.)
        wh[0] = *(const half8*)(st + b_row + swz_hi);
#pragma unroll
        for (int i = 0; i < FI; ++i) ah[i] = *(const half8*)(st + a_row + i * 16 * SPL_ROWB + swz_hi);
#pragma unroll
        for (int j = 1; j < FJ; ++j) wh[j] = *(const half8*)(st + b_row + j * 16 * SPL_ROWB + swz_hi);
        __builtin_amdgcn_sched_barrier(0);
        // stage kt + 1 into the other slot, one piece after every twelfth (WLO: eighth) MFMA (past the end: a valid slice into a slot nobody reads)
        char* abase = lds + ((kt + 1) & 1) * SP1_STAGE_B + wave * 32 * SPL_ROWB;
        char* bbase = lds + ((kt + 1) & 1) * SP1_STAGE_B + SP1_BM * SPL_ROWB + wave * 32 * SPL_ROWB;
        const char* as = a_src + (size_t)ks_stage * SPL_ROWB;
        const char* ws = w_src + (size_t)ks_stage * SPL_ROWB;
        ks_stage = next_slice(ks_stage);
        constexpr int NQ = WLO ? 96 : 64, PIECE_EVERY = NQ / 8, LO_READS = WLO ? 12 : FI;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {        // passes over the 32 fragment pairs (j-major): w_hi a_hi, [w_lo a_hi,] w_hi a_lo
            const int pass = q >> 5, j = ((q & 31) / FI), i = q % FI;
            if (pass == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
            else if (WLO && pass == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah[i], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], al[i], acc[i][j], 0, 0, 0);
            if (q < 2 * LO_READS && !(q & 1)) {         // a lo read after every other MFMA of the first pass: [the w_lo fragments, then] the a_lo fragments
                const int r = (q >> 1) + (WLO ? 0 : FJ);
                if (r < FJ) wl[r] = *(const half8*)(st + b_row + r * 16 * SPL_ROWB + swz_lo);
                else al[r - FJ] = *(const half8*)(st + a_row + (r - FJ) * 16 * SPL_ROWB + swz_lo);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q % PIECE_EVERY == PIECE_EVERY - 1) {
                const int pc = q / PIECE_EVERY;
                if (pc < 4) __builtin_amdgcn_global_load_lds((const AS1 void*)(as + (size_t)pc * 8 * pitch + lane_off), (AS3 void*)(abase + pc * 8 * SPL_ROWB), 16, 0, 0);
                else __builtin_amdgcn_global_load_lds((const AS1 void*)(ws + (size_t)(pc - 4) * 8 * pitch + lane_off), (AS3 void*)(bbase + (pc - 4) * 8 * SPL_ROWB), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing (dead) stage must have landed before the workgroup gives its LDS back

    constexpr bool HAS_BIAS = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RESID);
    const int row0 = m0 + wr * FI * 16 + frow, col0 = n0 + wc * FJ * 16 + fgrp * 4;
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) b = *(const f32x4*)(g.bias + col0 + j * 16);
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int row = row0 + i * 16;
            if (row >= g.M) continue;
            const int col = col0 + j * 16;
            f32x4 v = acc[i][j] * (1.0f / SP1_W_SCALE) + b;
            if constexpr (EPI == EPI_BIAS_RESID) v += *(const f32x4*)((const float*)g.resid + (size_t)row * g.ldc + col);
            if constexpr (EPI == EPI_BIAS_GELU_F16) {
                v = (f32x4){quick_gelu_exact_s(v[0]), quick_gelu_exact_s(v[1]), quick_gelu_exact_s(v[2]), quick_gelu_exact_s(v[3])};
                store4(SplitRow{(half_t*)g.out + (size_t)row * 2 * g.ldc}, col >> 2, v);
            } else {
                *(f32x4*)((float*)g.out + (size_t)row * g.ldc + col) = v;
            }
        }
    }
}
#endif

// x [rows, K] f32 (row pitch ld_in floats) -> split layout [rows, K/32, 64] halfs (weights at grip_tower_finalize; activations whose
// producer does not write the layout itself).  One thread per 4 consecutive k.
// `overflow` (weights only, may be null): set to 1 when a scaled element leaves the finite f16 range -- its hi part would be inf and every row
// of the tower non-finite (that used to end as a silent escalation of the whole pool to the f32 tower).
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, half_t* __restrict__ out, int64_t rows, int K, int64_t ld_in, float scale,
                                                         int* __restrict__ overflow, int* __restrict__ lo_nonzero) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k4 = K >> 2;
    if (i >= rows * k4) return;
    const int64_t r = i / k4;
    const int c = (int)(i - r * k4) * 4;
    const f32x4 v = *(const f32x4*)(x + r * ld_in + c) * scale;
    if (overflow) {
        const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (!(m < 65520.0f)) *overflow = 1;            // 65520 rounds to inf in f16; NaN lands here too (benign race: every writer stores 1)
    }
    if (lo_nonzero) {       // weights: does any element need its lo part?  (none does when the checkpoint holds f16 numbers: GemmArgs::w_exact)
        const bool exact = (float)(half_t)v.x == v.x && (float)(half_t)v.y == v.y && (float)(half_t)v.z == v.z && (float)(half_t)v.w == v.w;
        if (!exact) *lo_nonzero = 1;
    }
    store4(SplitRow{out + r * 2 * (int64_t)K}, c >> 2, v);
}

float gemm_split_weight_scale() {
#if GRIP_SPLIT_LO_SCALE == 1
    return SP1_W_SCALE;
#else
    return 1.0f;
#endif
}

int launch_split_rows(const float* x, void* out, int64_t rows, int K, int64_t ld_in, hipStream_t s, int is_weight, int* overflow_flag, int* lo_nonzero_flag) {
    GRIP_REQUIRE(K % 32 == 0 && rows > 0 && ld_in >= K && ld_in % 4 == 0, "split_rows: need K %% 32 == 0 (rows=%lld K=%d ld=%lld)", (long long)rows, K, (long long)ld_in);
    const int64_t n = rows * (K / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (half_t*)out, rows, K, ld_in, is_weight ? gemm_split_weight_scale() : 1.0f,
                       overflow_flag, lo_nonzero_flag);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

static int g_last_wlo = -1;
int gemm_split_last_wlo() { return g_last_wlo; }     // test hook: did the last launch form the a_hi w_lo product (1) or skip it (0)?

int launch_gemm_split(int epi, const GemmArgs& a, hipStream_t s) {
    GRIP_REQUIRE(a.K % 64 == 0 && a.M > 0, "gemm_split: need K %% 64 == 0 (M=%d N=%d K=%d)", a.M, a.N, a.K);
    GRIP_REQUIRE(a.ldc % 32 == 0, "gemm_split: ldc %% 32 != 0");
    GRIP_REQUIRE((int64_t)a.K * 4 * 8 < ((int64_t)1 << 31), "gemm_split: K too large for 32-bit lane offsets");
#if GRIP_SPLIT_LO_SCALE == 1
    constexpr int BM = SP1_BM, BN = SP1_BN;
    constexpr size_t lds = (size_t)2 * SP1_STAGE_B;
#define GRIP_SPLIT_KERNEL(E, WLO) gemm_split1_kernel<E, WLO>
#else
    constexpr int BM = SPL_BM, BN = SPL_BN;
    constexpr size_t lds = (size_t)SPL_NST * SPL_STAGE_B;
#define GRIP_SPLIT_KERNEL(E, WLO) gemm_split_kernel<E>
#endif
    GRIP_REQUIRE(a.N % BN == 0, "gemm_split: need N %% %d == 0 (N=%d)", BN, a.N);
    static const bool force_wlo = getenv("GRIP_SPLIT_WLO") && atoi(getenv("GRIP_SPLIT_WLO")) == 1;      // developer A/B: always form the a_hi w_lo product
    const bool wlo = !a.w_exact || force_wlo;
    g_last_wlo = wlo;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    GRIP_REQUIRE(a.m_pad >= (int64_t)tiles_m * BM, "gemm_split: A must be allocated up to the 256-row tile (M=%d m_pad=%lld)", a.M, (long long)a.m_pad);
    dim3 grid(tiles_m * tiles_n), block(512);
#define GRIP_GEMM_LAUNCH(E, WLO)                                                                                            \
    {                                                                                                                       \
        static bool configured = false;                                                                                     \
        if (!configured) {                                                                                                  \
            GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)GRIP_SPLIT_KERNEL(E, WLO), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            configured = true;                                                                                              \
        }                                                                                                                   \
        hipLaunchKernelGGL((GRIP_SPLIT_KERNEL(E, WLO)), grid, block, lds, s, a, tiles_m, tiles_n);                          \
    }
#define GRIP_GEMM_CASE(E)                                                                                                   \
    case E:                                                                                                                 \
        if (wlo) GRIP_GEMM_LAUNCH(E, true) else GRIP_GEMM_LAUNCH(E, false)                                                  \
        break;
    switch (epi) {
        GRIP_GEMM_CASE(EPI_F32)
        GRIP_GEMM_CASE(EPI_BIAS_F16)
        GRIP_GEMM_CASE(EPI_BIAS_GELU_F16)
        GRIP_GEMM_CASE(EPI_BIAS_RESID)
        default: GRIP_REQUIRE(false, "gemm_split: epilogue %d is not part of the split-f16 (inference) path", epi);
    }
#undef GRIP_GEMM_CASE
#undef GRIP_GEMM_LAUNCH
#undef GRIP_SPLIT_KERNEL
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
