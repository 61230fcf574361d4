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
//   O^T = V^T P^T          A = V^T fragments, one ds_read_b128 each from a blocked LDS image of V (vt_index, common.h),
//                          B = the packed P registers.  The image holds the head dims PERMUTED (vt_index_fwd): MFMA row m of
//                          block nf is dim (m >> 2) * 16 + nf * 4 + (m & 3), so a lane ends with the SIXTEEN consecutive
//                          head-dim values (lane >> 4) * 16 .. + 15 of its own query row, divides by its own row sum and
//                          stores them as two 16-byte pieces (the store tail is issue-bound: four 8-byte stores per lane
//                          and tile before, 433 -> see DESIGN.md).
// K is stored [kv][64] with the 16-byte chunk index XOR (kv & 7): conflict-free ds_read_b128.
// Padding keys (kv >= S) are zero-filled and masked to -inf; causal masking for the text tower.
#include <math.h>
#include <stdlib.h>

#include "common.h"

// Forward V image: dim d sits where vt_index would put dim ((d >> 2) & 3) * 16 + (d >> 4) * 4 + (d & 3).
__device__ __forceinline__ int vt_index_fwd(int key, int d) { return vt_index(key, ((d >> 2) & 3) * 16 + (d >> 4) * 4 + (d & 3)); }

// IEEE-754-2019 maximum: compiles to v_maximum3_f32 (two ops per four scores); fmaxf chains cost a v_max per pair plus a
// canonicalising v_max per MFMA result.
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }

// One 16-row query tile against the K / V^T of one (image, head) held in LDS: scores, softmax, P.V, store.
// qf = the tile's two Q fragments (unscaled); orow = the output row pointer of this lane's query (+ head offset).
// s_setprio(1) around phases of a query tile (r03).  Bit 0: the score MFMAs, bit 1: the exponentials, bit 2: P.V.  Measured (one box, two
// runs each, tools/attn_one.py; us per launch at 1320 x 12 x S = 197 (persistent kernel) / 128 x 16 x S = 577 (plain kernel)): off 416 / 380;
// 1: 412 / 352; 2: 443 / 362; 4: 444 / 361; 5: 439 / 359; 7: 440 / 361.  A wave that wins the issue arbitration while it issues its 2 x 14 (36 at
// S = 577) score MFMAs reaches its exponentials sooner and leaves the matrix pipe to the SIMD's other waves; prioritising the VALU or P.V phases
// only reorders waves that are all in the same phase.  Default 1: -7 % at S = 577 (ViT-L/14@336px), -1 % at S = 197.
#ifndef GRIP_ATTN_PRIO
#define GRIP_ATTN_PRIO 1
#endif
template <int KVC, bool CAUSAL>
__device__ __forceinline__ void attn_tile(const half_t* Ks, const half_t* Vt, half8 (&qf)[2], int qrow, int S, half_t* orow, int lane, bool do_store = true) {
    constexpr float LOG2E = 1.4426950408889634f;
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] *= (half_t)0.125f;  // 1/sqrt(64), exact in f16
    f32x4 sc[2 * KVC];
    float m = -INFINITY;
    if (GRIP_ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(1);
    // The launcher guarantees (KVC-1)*32 < S <= KVC*32: every tile before the last 32-key chunk is full, so
    // only the last two tiles (and causal rows) pay for the mask; the code stays one straight-line block.
    // The LAST 16-key tile holds no key at all when S <= (2 KVC - 1) * 16 (S = 197: keys 208 .. 223): its scores would all be masked to -inf and its
    // probabilities exact zeros, so neither its two score MFMAs nor its four exponentials are issued (r04; a wave-uniform branch; same bits).
    const bool last_dead = S <= (2 * KVC - 1) * 16;
#pragma unroll
    for (int t = 0; t < 2 * KVC; ++t) {
        if (t == 2 * KVC - 1 && last_dead) { sc[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}; continue; }
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
        m = max3(max3(m, acc[0], acc[1]), acc[2], acc[3]);
        sc[t] = acc;
    }
    if (GRIP_ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(0);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    const float m2 = m * LOG2E;
    if (GRIP_ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 2 * KVC; ++t) {
        if (t == 2 * KVC - 1 && last_dead) { sc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }       // exp(-inf - m)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][r], LOG2E, -m2));   // exp(s - m)
    }
    if (GRIP_ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(0);

    // The row sums come out of the matrix pipe: a fifth accumulator takes an all-ones A operand, so every row of it is
    // sum_k P[q][k] for the lane's own query (56 v_add and two cross-lane steps per tile less on the VALU, which is the
    // busier pipe here; the sum is over the f16-rounded numerators, i.e. exactly what the P.V product uses).
    const half8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
    f32x4 osum = {0.f, 0.f, 0.f, 0.f};
    f32x4 o[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) o[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (GRIP_ATTN_PRIO & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int c = 0; c < KVC; ++c) {
        const half8 pf = {(half_t)sc[2 * c][0], (half_t)sc[2 * c][1], (half_t)sc[2 * c][2], (half_t)sc[2 * c][3],
                          (half_t)sc[2 * c + 1][0], (half_t)sc[2 * c + 1][1], (half_t)sc[2 * c + 1][2], (half_t)sc[2 * c + 1][3]};
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            const half8 vf = *(const half8*)(Vt + (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8);
            o[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[nf], 0, 0, 0);
        }
        osum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf, osum, 0, 0, 0);
    }
    if (GRIP_ATTN_PRIO & 4) __builtin_amdgcn_s_setprio(0);
    if (qrow < S && do_store) {
        const float inv = __builtin_amdgcn_rcpf(osum[0]);
        half_t* op = orow + lg * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 a = o[2 * h] * inv, b = o[2 * h + 1] * inv;
            *(half8*)(op + h * 8) = (half8){(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
        }
    }
}

template <int KVC, bool CAUSAL, int ATT_NW>
__global__ __launch_bounds__(ATT_NW * 64) void attn_fwd_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int S, int H, int Ps) {
    constexpr int SP = KVC * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Ks = (half_t*)smem;
    half_t* Vt = Ks + SP * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = H * 64;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t ld = (size_t)3 * D;
    const half_t* base = qkv + h * 64;
    const int li = lane & 15, lg = lane >> 4;
    const int n_qt = (S + 15) >> 4;
    // shared-prefix layout: the shared rows are produced by sequence 0's workgroups only
    const int q_min = (Ps > 0 && b > 0) ? Ps : 0;
    const int qt0 = q_min >> 4;

    // Q fragments of this wave's first tile go out before the K/V staging so their latency hides behind it;
    // inside the loop the NEXT tile's Q is fetched while the current one is being processed.
    auto load_q = [&](int qt, half8 (&qf)[2]) {
        const int qrow = qt * 16 + li;
        const int qr = qrow < S ? qrow : S - 1;
        const half_t* qp = base + seq_row(b, qr, S, Ps) * ld;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const half8*)(qp + (kk * 4 + lg) * 8);
    };
    half8 q_next[2];
    load_q(qt0 + wave < n_qt ? qt0 + wave : 0, q_next);

    // K: [kv][64] rows, 16-byte chunk index XOR (kv & 7).
    for (int idx = tid; idx < SP * 8; idx += ATT_NW * 64) {
        const int row = idx >> 3, chunk = idx & 7;
        half8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < S) kv = *(const half8*)(base + seq_row(b, row, S, Ps) * ld + D + chunk * 8);
        *(half8*)(Ks + row * 64 + ((chunk ^ (row & 7)) * 8)) = kv;
    }
    // V image (vt_index).  A lane takes a PAIR of keys (2r, 2r+1) and one 8-wide slice of the head dim and writes
    // eight 32-bit words {V[2r][d], V[2r+1][d]}; lanes 0-31 are 32 consecutive key pairs, lanes 32-63 the next slice.
    for (int idx = tid; idx < ((SP / 2 + 31) / 32) * 256; idx += ATT_NW * 64) {
        const int lane_rp = idx & 31, chunk = ((idx >> 5) & 1) + 2 * ((idx >> 6) & 3), rblk = idx >> 8;
        const int rp = rblk * 32 + lane_rp;          // key pair index
        const int r0 = 2 * rp;
        if (r0 >= SP) continue;
        half8 v0 = {0, 0, 0, 0, 0, 0, 0, 0}, v1 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 < S) v0 = *(const half8*)(base + seq_row(b, r0, S, Ps) * ld + 2 * D + chunk * 8);
        if (r0 + 1 < S) v1 = *(const half8*)(base + seq_row(b, r0 + 1, S, Ps) * ld + 2 * D + chunk * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) *(half2v*)(Vt + vt_index_fwd(r0, chunk * 8 + j)) = (half2v){v0[j], v1[j]};
    }
    __syncthreads();

    for (int qt = qt0 + wave; qt < n_qt; qt += ATT_NW) {
        asm volatile("" ::: "memory");  // keep the K/V fragment reads inside the tile loop (hoisting them costs >100 VGPRs)
        const int qrow = qt * 16 + li;
        half8 qf[2] = {q_next[0], q_next[1]};
        if (qt + ATT_NW < n_qt) load_q(qt + ATT_NW, q_next);
        const int qs = qrow < S ? qrow : S - 1;
        attn_tile<KVC, CAUSAL>(Ks, Vt, qf, qrow, S, out + seq_row(b, qs, S, Ps) * D + h * 64, lane, qrow >= q_min);
    }
}

// ---- Pipelined variant for the pool encode (non-causal, S <= 288): persistent workgroups, one per CU, walk the
// (image, head) items with TWO K / V^T buffers in LDS.  While the waves work on item i, item i+1 is already on its way:
//   K    HBM -> LDS by global_load_lds (no registers, no waits; the XOR swizzle is applied to the SOURCE chunk index),
//   V    HBM -> registers at the top of the item, transposed into the other buffer's V^T after the item's last tile,
//   Q    the wave's query fragments for item i+1.
// One barrier per item.  The plain kernel above runs staging and compute back to back (measured 96 us + 98 us of a
// 176 us launch at 440 x 12 heads, S = 197: no overlap, the two co-resident workgroups move in lock step); this one
// takes 169 us.  (An 8-wave variant with batched fragment reads, for more registers per wave, was slower: 191 us.)
template <int KVC, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_pipe_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int S, int H, int n_items, int dbg) {
    constexpr int SP = KVC * 32;
    constexpr int BUF = 2 * SP * 64;                            // halfs per buffer (K rows + V image)
    constexpr int NQ = (2 * KVC + NW - 1) / NW;                 // query tiles per wave
    constexpr int V_ITEMS = ((SP / 2 + 31) / 32) * 256;         // (key pair, 8-wide head-dim slice) units
    constexpr int NV = (V_ITEMS + NW * 64 - 1) / (NW * 64);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* lds = (half_t*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = H * 64;
    const size_t ld = (size_t)3 * D;
    const int li = lane & 15, lg = lane >> 4;
    const int n_qt = (S + 15) >> 4;

    auto item_base = [&](int item) {
        const int b = item / H, h = item - b * H;
        return qkv + (size_t)b * S * ld + h * 64;
    };
    // K rows 8 at a time per wave instruction; LDS slot (row, c) receives source chunk c ^ (row & 7).  Rows >= S repeat
    // row S-1 (their scores are masked to -inf whatever they hold).
    auto issue_k = [&](const half_t* base, int buf) {
        for (int g = wave; g < SP / 8; g += NW) {
            int row = g * 8 + (lane >> 3);
            row = row < S ? row : S - 1;
            const half_t* src = base + (size_t)row * ld + D + (((lane & 7) ^ (lane >> 3)) * 8);
            __builtin_amdgcn_global_load_lds((const AS1 void*)src, (AS3 void*)(lds + buf * BUF + g * 512), 16, 0, 0);
        }
    };
    half8 v0[NV], v1[NV];
    auto load_v = [&](const half_t* base) {
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int idx = tid + n * NW * 64;
            const int lane_rp = idx & 31, chunk = ((idx >> 5) & 1) + 2 * ((idx >> 6) & 3), rblk = idx >> 8;
            const int r0 = 2 * (rblk * 32 + lane_rp);
            v0[n] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
            v1[n] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
            if (r0 < S) v0[n] = *(const half8*)(base + (size_t)r0 * ld + 2 * D + chunk * 8);
            if (r0 + 1 < S) v1[n] = *(const half8*)(base + (size_t)(r0 + 1) * ld + 2 * D + chunk * 8);
        }
    };
    auto store_v = [&](int buf) {
        half_t* Vt = lds + buf * BUF + SP * 64;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int idx = tid + n * NW * 64;
            const int lane_rp = idx & 31, chunk = ((idx >> 5) & 1) + 2 * ((idx >> 6) & 3), rblk = idx >> 8;
            const int r0 = 2 * (rblk * 32 + lane_rp);
            if (r0 < SP) {
#pragma unroll
                for (int j = 0; j < 8; ++j) *(half2v*)(Vt + vt_index_fwd(r0, chunk * 8 + j)) = (half2v){v0[n][j], v1[n][j]};
            }
        }
    };
    half8 q[NQ][2];
    auto load_q = [&](const half_t* base) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int qrow = (wave + n * NW) * 16 + li;
            const int qr = qrow < S ? qrow : S - 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) q[n][kk] = *(const half8*)(base + (size_t)qr * ld + (kk * 4 + lg) * 8);
        }
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    {
        const half_t* base = item_base(item);
        issue_k(base, 0);
        load_v(base);
        load_q(base);
        store_v(0);
    }
    for (int it = 0;; ++it) {
        const int buf = it & 1;
        // This wave's share of the item's K (LDS-DMA) has landed: it was issued BEFORE the item's V loads, and store_v() at the end
        // of the previous iteration (the prologue for item 0) consumed those V registers -- the compiler's wait for them
        // covers every older vector-memory operation.  No explicit vmcnt(0) here: it would also drain the output stores the
        // wave issued a moment ago and expose their write latency once per item (measured: 433 -> see DESIGN.md us per launch).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the V^T image writes (ds_write) of this wave are done
        __builtin_amdgcn_s_barrier();                       // ... everyone's K has landed, V^T is written, and buffer buf^1 is free
        asm volatile("" ::: "memory");
        const int next = item + gridDim.x;
        const bool has_next = next < n_items;
        half8 qf[NQ][2];
#pragma unroll
        for (int n = 0; n < NQ; ++n) { qf[n][0] = q[n][0]; qf[n][1] = q[n][1]; }
        if (has_next) {
            const half_t* nb = item_base(next);
            issue_k(nb, buf ^ 1);
            load_v(nb);
            load_q(nb);
        }
        const int b = item / H, h = item - b * H;
        const half_t* Ks = lds + buf * BUF;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int qt = wave + n * NW;
            if (qt < n_qt) {
                asm volatile("" ::: "memory");
                const int qrow = qt * 16 + li;
                attn_tile<KVC, false>(Ks, Ks + SP * 64, qf[n], qrow, S, out + ((size_t)b * S + qrow) * D + h * 64, lane, !(dbg & 1) || item < 64);
            }
        }
        if (!has_next) break;
        store_v(buf ^ 1);
        item = next;
    }
}

template <int KVC, int NW>
static int launch_pipe(const half_t* qkv, half_t* out, int B, int S, int H, hipStream_t s) {
    constexpr int SP = KVC * 32;
    constexpr size_t lds = 2 * ((size_t)2 * SP * 64 * 2);
    static int resident = 0, per_cu_s = 0;     // workgroups the device holds at once (LDS- or register-limited), = the persistent grid
    if (!resident) {
        GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel<KVC, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0, n_cu = 0, per_cu = 0;
        GRIP_CHECK_HIP(hipGetDevice(&dev));
        GRIP_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        GRIP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_fwd_pipe_kernel<KVC, NW>, NW * 64, lds));
        GRIP_REQUIRE(per_cu >= 1, "attention: pipelined kernel does not fit a CU (KVC %d)", KVC);
        resident = n_cu * per_cu;
        per_cu_s = per_cu;
    }
    const int n_items = B * H;
    const int width = (grip_cu_budget() > 0 && grip_cu_budget() * per_cu_s < resident) ? grip_cu_budget() * per_cu_s : resident;     // a CU-masked launch stream
    const int grid = n_items < width ? n_items : width;
    static const int dbg = getenv("GRIP_ATTN_DBG") ? atoi(getenv("GRIP_ATTN_DBG")) : 0;     // developer experiments (bit 0: skip the output stores)
    hipLaunchKernelGGL((attn_fwd_pipe_kernel<KVC, NW>), dim3(grid), dim3(NW * 64), lds, s, qkv, out, S, H, n_items, dbg);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

template <int KVC, bool CAUSAL, int ATT_NW>
static int launch_one(const half_t* qkv, half_t* out, int B, int S, int H, hipStream_t s, int Ps) {
    constexpr int SP = KVC * 32;
    constexpr size_t lds = (size_t)2 * SP * 64 * 2;
    static bool configured = false;
    if (!configured) {
        GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel<KVC, CAUSAL, ATT_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<KVC, CAUSAL, ATT_NW>), dim3(B * H), dim3(ATT_NW * 64), lds, s, qkv, out, S, H, Ps);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

int launch_attention_fwd(const half_t* qkv, half_t* out, int B, int S, int H, int causal, hipStream_t s, int shared_rows) {
    const int kvc = (S + 31) / 32;
    GRIP_REQUIRE(shared_rows == 0 || (causal && shared_rows > 0 && shared_rows < S), "attention: the shared-prefix layout needs a causal mask and 0 < shared rows < S");
    GRIP_REQUIRE(S >= 1 && kvc <= 19, "attention: sequence length %d unsupported (max 608)", S);
    // exact chunk count (the kernel relies on (KVC-1)*32 < S)
    // The persistent double-buffered kernel serves the pool encode of the 197..224-token towers (16 waves fit the
    // 128-VGPR budget up to 7 chunks); GRIP_ATTN_PIPE=0 switches it off (developer A/B).
    static const bool pipe = !(getenv("GRIP_ATTN_PIPE") && atoi(getenv("GRIP_ATTN_PIPE")) == 0);
    if (!causal && pipe && B * H >= 1024) {      // enough items for every CU to pipeline over several
        if (kvc == 5) return launch_pipe<5, 16>(qkv, out, B, S, H, s);
        if (kvc == 6) return launch_pipe<6, 16>(qkv, out, B, S, H, s);
        if (kvc == 7) return launch_pipe<7, 16>(qkv, out, B, S, H, s);
    }
#define GRIP_ATTN(N)                                                        \
    if (kvc == N) return causal ? launch_one<N, true, (N >= 4 ? 8 : 4)>(qkv, out, B, S, H, s, shared_rows) \
                                : launch_one<N, false, (N >= 4 ? 8 : 4)>(qkv, out, B, S, H, s, 0);
    GRIP_ATTN(1) GRIP_ATTN(2) GRIP_ATTN(3) GRIP_ATTN(4) GRIP_ATTN(5) GRIP_ATTN(6) GRIP_ATTN(7) GRIP_ATTN(8) GRIP_ATTN(9) GRIP_ATTN(10)
    GRIP_ATTN(11) GRIP_ATTN(12) GRIP_ATTN(13) GRIP_ATTN(14) GRIP_ATTN(15) GRIP_ATTN(16) GRIP_ATTN(17) GRIP_ATTN(18) GRIP_ATTN(19)
#undef GRIP_ATTN
    return GRIP_ERR_ARG;
}


// ---- Attention for ONE query row per sequence (f16 in, f32 math, f16 out): the last block of a tower at inference.  Only the
// CLS row (vision) / EOT row (text) of the final residual stream is ever read (models/clip_encoders.py:189 and :86-89), and after
// the last block's attention rows no longer mix, so that block needs its attention output, out-proj and MLP for that row alone
// (keys and values still come from every row).  One wave per (sequence, head): lanes over keys for the scores (q broadcast
// from LDS), wave-wide softmax, lanes over the 64 head dims for P.V.
__global__ __launch_bounds__(64) void attn_row_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ qrows, const int32_t* __restrict__ row_index,
                                                     half_t* __restrict__ out, int S, int H, int causal) {
    extern __shared__ float sm[];         // [64] q, [S] probabilities
    float* qs = sm;
    float* ps = sm + 64;
    const int lane = threadIdx.x;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * 64;
    const size_t ld = (size_t)3 * D;
    const int r = row_index ? row_index[b] : 0;
    const half_t* base = qkv + (size_t)b * S * ld + h * 64;
    // the query row: from the compact [B, D] matrix when the caller projected only those rows, else from the packed qkv
    qs[lane] = (float)(qrows ? qrows[(size_t)b * D + h * 64 + lane] : base[(size_t)r * ld + lane]) * 0.125f;
    __syncthreads();
    const int n_keys = causal ? r + 1 : S;
    float m = -INFINITY;
    for (int j = lane; j < n_keys; j += 64) {
        const half8* kr = (const half8*)(base + (size_t)j * ld + D);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half8 kv = kr[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) a = __builtin_fmaf(qs[c * 8 + e], (float)kv[e], a);
        }
        ps[j] = a;
        m = fmaxf(m, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int j = lane; j < n_keys; j += 64) {
        const float p = __expf(ps[j] - m);
        ps[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    // P.V: lane = (key group lane >> 3, 8-wide slice of the head dim lane & 7): 16-byte loads, eight keys in flight per step; the
    // eight key groups are then folded with three xor-shuffles (lanes 8, 16, 32 apart hold the same slice).
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int kg = lane >> 3, ch = lane & 7;
    for (int j = kg; j < n_keys; j += 8) {
        const half8 v = *(const half8*)(base + (size_t)j * ld + 2 * D + ch * 8);
        const float p = ps[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)v[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 8);
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    if (kg == 0) {
        const float inv = 1.0f / sum;
        *(half8*)(out + (size_t)b * D + h * 64 + ch * 8) = (half8){(half_t)(acc[0] * inv), (half_t)(acc[1] * inv), (half_t)(acc[2] * inv), (half_t)(acc[3] * inv),
                                                                   (half_t)(acc[4] * inv), (half_t)(acc[5] * inv), (half_t)(acc[6] * inv), (half_t)(acc[7] * inv)};
    }
}

// ---- The same with FOUR waves per (sequence, head): the train-mode last block (csrc/tower.hip), where a prompt-step batch gives the
// one-wave form only B x H = 192 waves of ~200 dependent key trips (16 us for a 16-image batch).  Keys are dealt over 256 threads for the
// scores and over 32 key groups for P.V; the block-wide max / sum / P.V partials meet in LDS and are combined in wave order.
// Another summation order than attn_row_kernel, so the MODE picks the kernel, never the batch size (inference stays on the one-wave form:
// a row is computed the same way in whatever chunk it arrives).
__device__ __forceinline__ float block4_max(float v, float* red, int wave, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if (lane == 0) red[wave] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block4_sum(float v, float* red, int wave, int lane) {
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    v = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(256) void attn_row4_kernel(const half_t* __restrict__ qkv, const int32_t* __restrict__ row_index, half_t* __restrict__ out, int S, int H,
                                                       int causal) {
    extern __shared__ float sm[];         // [64] q / 8, [4] reduction, [4 x 64] P.V partials, [S] probabilities
    float* qs = sm;
    float* red = sm + 64;
    float* pv = sm + 68;
    float* ps = sm + 68 + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * 64;
    const size_t ld = (size_t)3 * D;
    const int r = row_index ? row_index[b] : 0;
    const half_t* base = qkv + (size_t)b * S * ld + h * 64;
    if (tid < 64) qs[tid] = (float)base[(size_t)r * ld + tid] * 0.125f;
    __syncthreads();
    const int n_keys = causal ? r + 1 : S;
    float m = -INFINITY;
    for (int j = tid; j < n_keys; j += 256) {
        const half8* kr = (const half8*)(base + (size_t)j * ld + D);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half8 kv = kr[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) a = __builtin_fmaf(qs[c * 8 + e], (float)kv[e], a);
        }
        ps[j] = a;
        m = fmaxf(m, a);
    }
    m = block4_max(m, red, wave, lane);
    float sum = 0.f;
    for (int j = tid; j < n_keys; j += 256) {
        const float p = __expf(ps[j] - m);
        ps[j] = p;
        sum += p;
    }
    sum = block4_sum(sum, red, wave, lane);      // (its barriers also publish ps)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int kg = tid >> 3, ch = tid & 7;
    for (int j = kg; j < n_keys; j += 32) {
        const half8 v = *(const half8*)(base + (size_t)j * ld + 2 * D + ch * 8);
        const float p = ps[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, (float)v[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 8);
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    if (lane < 8)
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[wave * 64 + ch * 8 + e] = acc[e];
    __syncthreads();
    if (tid < 64) {
        const float inv = 1.0f / sum;
        out[(size_t)b * D + h * 64 + tid] = (half_t)((((pv[tid] + pv[64 + tid]) + pv[128 + tid]) + pv[192 + tid]) * inv);
    }
}

// ---- Backward of attn_row_kernel (train-mode last block, csrc/tower.hip run_blocks_backward): ONE query row per sequence attends to
// all keys, so dQ is one row, dK_j = dS_j q / 8 and dV_j = P_j dO are rank-one in the row's (q, dO), and every other query row of dQ is
// zero.  Writes the whole packed [B, S, 3, H*64] gradient (zeros included), as the full attention backward would for a d_out that is
// zero outside the read rows.  Four waves per (sequence, head), the thread maps of attn_row4_kernel: threads over keys for the scores and
// dP = dO . V_j, (key group, 8-wide slice) for the outputs; dQ folds its 32 key groups with three xor-shuffles and one pass through LDS.
__global__ __launch_bounds__(256) void attn_row_bwd_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ o_rows, const half_t* __restrict__ do_rows,
                                                          const int32_t* __restrict__ row_index, half_t* __restrict__ dqkv, int S, int H, int causal) {
    extern __shared__ float sm[];         // [64] q / 8, [64] dO, [4] reduction, [4 x 64] dQ partials, [S] probabilities, [S] dS
    float* qs = sm;
    float* dos = sm + 64;
    float* red = sm + 128;
    float* dqp = sm + 132;
    float* ps = sm + 132 + 256;
    float* dss = ps + S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * 64;
    const size_t ld = (size_t)3 * D;
    const int r = row_index ? row_index[b] : 0;
    const half_t* base = qkv + (size_t)b * S * ld + h * 64;
    half_t* dbase = dqkv + (size_t)b * S * ld + h * 64;
    float dd = 0.f;
    if (tid < 64) {
        const float dov = (float)do_rows[(size_t)b * D + h * 64 + tid];
        qs[tid] = (float)base[(size_t)r * ld + tid] * 0.125f;
        dos[tid] = dov;
        dd = dov * (float)o_rows[(size_t)b * D + h * 64 + tid];
    }
    const float delta = block4_sum(dd, red, wave, lane);      // rowsum(dO o O); its barriers publish qs / dos
    const int n_keys = causal ? r + 1 : S;
    float m = -INFINITY;
    for (int j = tid; j < n_keys; j += 256) {
        const half8* kr = (const half8*)(base + (size_t)j * ld + D);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half8 kv = kr[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) a = __builtin_fmaf(qs[c * 8 + e], (float)kv[e], a);
        }
        ps[j] = a;
        m = fmaxf(m, a);
    }
    m = block4_max(m, red, wave, lane);
    float sum = 0.f;
    for (int j = tid; j < n_keys; j += 256) {
        const float p = __expf(ps[j] - m);
        ps[j] = p;
        sum += p;
    }
    sum = block4_sum(sum, red, wave, lane);
    const float inv = 1.0f / sum;
    for (int j = tid; j < n_keys; j += 256) {      // (each thread revisits the keys it wrote)
        const half8* vr = (const half8*)(base + (size_t)j * ld + 2 * D);
        float dp = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half8 vv = vr[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) dp = __builtin_fmaf(dos[c * 8 + e], (float)vv[e], dp);
        }
        const float pn = ps[j] * inv;
        ps[j] = pn;
        dss[j] = pn * (dp - delta);
    }
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int kg = tid >> 3, ch = tid & 7;
    const half8 zero = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    for (int j = kg; j < S; j += 32) {
        half8 dk = zero, dv = zero;
        if (j < n_keys) {
            const half8 k = *(const half8*)(base + (size_t)j * ld + D + ch * 8);
            const float ds = dss[j], p = ps[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] = __builtin_fmaf(ds, (float)k[e], acc[e]);
                dk[e] = (half_t)(ds * qs[ch * 8 + e]);          // qs carries the 1/8
                dv[e] = (half_t)(p * dos[ch * 8 + e]);
            }
        }
        *(half8*)(dbase + (size_t)j * ld + D + ch * 8) = dk;
        *(half8*)(dbase + (size_t)j * ld + 2 * D + ch * 8) = dv;
        if (j != r) *(half8*)(dbase + (size_t)j * ld + ch * 8) = zero;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 8);
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    if (lane < 8)
#pragma unroll
        for (int e = 0; e < 8; ++e) dqp[wave * 64 + ch * 8 + e] = acc[e];
    __syncthreads();
    if (tid < 64) dbase[(size_t)r * ld + tid] = (half_t)((((dqp[tid] + dqp[64 + tid]) + dqp[128 + tid]) + dqp[192 + tid]) * 0.125f);
}

int launch_attention_row_bwd(const half_t* qkv, const half_t* o_rows, const half_t* do_rows, const int32_t* row_index, half_t* dqkv, int B, int S, int H, int causal,
                             hipStream_t s) {
    GRIP_REQUIRE(B >= 1 && S >= 1 && H >= 1, "attention_row_bwd: bad shape");
    hipLaunchKernelGGL(attn_row_bwd_kernel, dim3(B * H), dim3(256), (size_t)(132 + 256 + 2 * S) * sizeof(float), s, qkv, o_rows, do_rows, row_index, dqkv, S, H, causal);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

int launch_attention_row(const half_t* qkv, const half_t* qrows, const int32_t* row_index, half_t* out, int B, int S, int H, int causal, hipStream_t s, int train) {
    GRIP_REQUIRE(B >= 1 && S >= 1 && H >= 1, "attention_row: bad shape");
    if (train) {      // four waves per (sequence, head): the prompt steps' small batches
        GRIP_REQUIRE(!qrows, "attention_row: the train-mode form reads its queries from the packed projection");
        hipLaunchKernelGGL(attn_row4_kernel, dim3(B * H), dim3(256), (size_t)(68 + 256 + S) * sizeof(float), s, qkv, row_index, out, S, H, causal);
        GRIP_CHECK_HIP(hipGetLastError());
        return GRIP_OK;
    }
    hipLaunchKernelGGL(attn_row_kernel, dim3(B * H), dim3(64), (size_t)(64 + S) * sizeof(float), s, qkv, qrows, row_index, out, S, H, causal);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
