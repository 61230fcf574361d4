// Backward of the fused self-attention (input gradients only): dqkv from qkv, the saved output O
// and dO.  One workgroup of 8 waves (4 for S <= 96) per (image, head): a training batch is 16 x 12 = 192 workgroups on 256
// CUs, so the waves inside a workgroup are the only parallelism a CU sees; K and V of that head stay in LDS for both
// phases, probabilities are RECOMPUTED from q, k (nothing but O was saved by the forward).
//
//   phase 1, per 16-row query tile (wave-parallel):  S^T = K (Q/8)^T, row max / sum  -> P^T;
//            dP^T = V dO^T;  delta = rowsum(dO * O);  dS^T = P^T * (dP^T - delta);
//            dQ^T = K^T dS^T / 8   (A = transposed K copy in LDS, B = packed dS registers);
//            the row statistics (max, 1/sum, delta) go to LDS for phase 2.
//   phase 2, per 16-row key tile (wave-parallel), looping over 32 queries at a time:
//            S = (Q/8) K^T and dP = dO V^T recomputed in the [query][key] orientation, which puts
//            the QUERY index on the MFMA contraction axis:  dV^T += dO^T P,  dK^T += (Q/8)^T dS
//            (A = transposed dO / Q copies in LDS, B = packed P / dS registers).
// All contractions are v_mfma_f32_16x16x32_f16; softmax statistics and dS are f32.
#include <math.h>

#include "common.h"

// Developer experiment (r03): s_setprio(1) around the score MFMAs of phase 1 (bit 0), the dQ MFMAs (bit 1), phase 2's S / dP MFMAs (bit 2), its dK / dV MFMAs (bit 3)
#ifndef GRIP_ATTNB_PRIO
#define GRIP_ATTNB_PRIO 0
#endif

template <int KVC, bool CAUSAL, int NWB>
__global__ __launch_bounds__(NWB * 64) void attn_bwd_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ o_saved,
                                                       const half_t* __restrict__ d_out, half_t* __restrict__ dqkv, int S, int H,
                                                       int Ps, float* __restrict__ kv_part) {
    constexpr int SP = KVC * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Ks = (half_t*)smem;          // [SP][64] swizzled rows
    half_t* Vs = Ks + SP * 64;           // [SP][64] swizzled rows
    half_t* T0 = Vs + SP * 64;           // blocked transposed image (vt_index): K in phase 1, Q/8 in phase 2
    half_t* T1 = T0 + SP * 64;           // blocked transposed image of dO (phase 2)
    float* st_m = (float*)(T1 + SP * 64);    // [SP] row max
    float* st_il = st_m + SP;                // [SP] 1 / row sum
    float* st_d = st_il + SP;                // [SP] delta

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int D = H * 64;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t ld = (size_t)3 * D;
    // rows through seq_row (common.h): plain layout b*S + r, or the shared-prefix layout of the text tower (Ps > 0), where the
    // shared query rows belong to sequence 0 alone (q_min) and every sequence adds its share to the shared keys' dK / dV
    const half_t* base = qkv + h * 64;
    const half_t* obase = o_saved + h * 64;
    const half_t* dobase = d_out + h * 64;
    half_t* dbase = dqkv + h * 64;
    const int q_min = (Ps > 0 && b > 0) ? Ps : 0;
    auto R = [&](int r) { return seq_row(b, r, S, Ps); };

    // A thread takes a PAIR of rows (2 r, 2 r + 1) and one 8-wide slice: the transposed images hold two consecutive keys of a head dim in one 32-bit
    // word (vt_index), so they are written as eight 4-byte words instead of sixteen 2-byte ones (r04: the 2-byte writes were the 39 % LDS bank
    // conflicts of profiles/r03_sq_vpt.csv), the forward's V staging pattern.
    // r05: lanes run over ROW PAIRS (32 per 32-lane group), the 8-wide slice comes from the higher index bits -- the forward's staging map.  With eight
    // consecutive lanes on the eight slices of ONE row pair (r04) every 4-byte write of a group landed in 4 banks (the slice only moves the address by whole
    // 1-KiB blocks): 8-way conflicts on all 24 image writes per thread, most of the 31 % of profiles/r04_sq_vpt.csv.  Now 32 lanes hit 16 banks twice (free).
    constexpr int ST_ITEMS = ((SP / 2 + 31) / 32) * 256;
    for (int idx = tid; idx < ST_ITEMS; idx += NWB * 64) {
        const int r0 = 2 * ((idx >> 8) * 32 + (idx & 31)), chunk = (idx >> 5) & 7;
        if (r0 >= SP) continue;
        half8 k0 = {0, 0, 0, 0, 0, 0, 0, 0}, k1 = k0, v0 = k0, v1 = k0;
        if (r0 < S) {
            const half_t* rp = base + R(r0) * ld;
            k0 = *(const half8*)(rp + D + chunk * 8);
            v0 = *(const half8*)(rp + 2 * D + chunk * 8);
        }
        if (r0 + 1 < S) {
            const half_t* rp = base + R(r0 + 1) * ld;
            k1 = *(const half8*)(rp + D + chunk * 8);
            v1 = *(const half8*)(rp + 2 * D + chunk * 8);
        }
        const int sw0 = (chunk ^ (r0 & 7)) * 8, sw1 = (chunk ^ ((r0 + 1) & 7)) * 8;
        *(half8*)(Ks + r0 * 64 + sw0) = k0;
        *(half8*)(Ks + (r0 + 1) * 64 + sw1) = k1;
        *(half8*)(Vs + r0 * 64 + sw0) = v0;
        *(half8*)(Vs + (r0 + 1) * 64 + sw1) = v1;
#pragma unroll
        for (int j = 0; j < 8; ++j) *(half2v*)(T0 + vt_index(r0, chunk * 8 + j)) = (half2v){k0[j], k1[j]};
    }
    __syncthreads();

    const int n_qt = (S + 15) >> 4;
    // ------------------------------------------------------------------ phase 1: dQ and row statistics
    for (int qt = (q_min >> 4) + wave; qt < n_qt; qt += NWB) {
        asm volatile("" ::: "memory");
        const int qrow = qt * 16 + li;
        const int qr = qrow < S ? qrow : S - 1;
        const size_t qg = R(qr);
        half8 qf[2], dof[2];
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[kk] = *(const half8*)(base + qg * ld + (kk * 4 + lg) * 8);
            qf[kk] *= (half_t)0.125f;
            dof[kk] = *(const half8*)(dobase + qg * D + (kk * 4 + lg) * 8);
            const half8 of = *(const half8*)(obase + qg * D + (kk * 4 + lg) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)dof[kk][j] * (float)of[j];
        }
        dl += __shfl_xor(dl, 16);
        dl += __shfl_xor(dl, 32);

        f32x4 sc[2 * KVC];
        float m = -INFINITY;
        if (GRIP_ATTNB_PRIO & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 2 * KVC; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8 kf = *(const half8*)(Ks + (t * 16 + li) * 64 + (((kk * 4 + lg) ^ (lane & 7)) * 8));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kv = t * 16 + lg * 4 + r;
                if (kv >= S || (CAUSAL && kv > qrow)) acc[r] = -INFINITY;
                m = fmaxf(m, acc[r]);
            }
            sc[t] = acc;
        }
        if (GRIP_ATTNB_PRIO & 1) __builtin_amdgcn_s_setprio(0);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * KVC; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[t][r] - m);
                sc[t][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float il = 1.0f / sum;
        if (lg == 0 && qrow < SP) { st_m[qrow] = m; st_il[qrow] = il; st_d[qrow] = dl; }

        // dS^T = P^T * (dP^T - delta), dP^T tile = V dO^T
#pragma unroll
        for (int t = 0; t < 2 * KVC; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8 vf = *(const half8*)(Vs + (t * 16 + li) * 64 + (((kk * 4 + lg) ^ (lane & 7)) * 8));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, dof[kk], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t][r] = sc[t][r] * il * (acc[r] - dl);
        }
        // dQ^T[dh][q] = sum_kv K^T[dh][kv] dS^T[kv][q]
        f32x4 dq[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) dq[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (GRIP_ATTNB_PRIO & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int c = 0; c < KVC; ++c) {
            const half8 sf = {(half_t)sc[2 * c][0], (half_t)sc[2 * c][1], (half_t)sc[2 * c][2], (half_t)sc[2 * c][3],
                              (half_t)sc[2 * c + 1][0], (half_t)sc[2 * c + 1][1], (half_t)sc[2 * c + 1][2], (half_t)sc[2 * c + 1][3]};
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const half8 kf = *(const half8*)(T0 + (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8);
                dq[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, sf, dq[nf], 0, 0, 0);
            }
        }
        if (GRIP_ATTNB_PRIO & 2) __builtin_amdgcn_s_setprio(0);
        if (qrow < S && qrow >= q_min) {
            half_t* op = dbase + qg * ld + lg * 4;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const f32x4 v = dq[nf] * 0.125f;
                *(half4*)(op + nf * 16) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            }
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ restage: T0 = (Q/8)^T, T1 = dO^T
    for (int idx = tid; idx < ST_ITEMS; idx += NWB * 64) {
        const int r0 = 2 * ((idx >> 8) * 32 + (idx & 31)), chunk = (idx >> 5) & 7;
        if (r0 >= SP) continue;
        half8 q0 = {0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0, d0 = q0, d1 = q0;
        if (r0 < S && r0 >= q_min) {      // (queries below q_min are not this sequence's: zero rows add nothing to dK / dV)
            q0 = *(const half8*)(base + R(r0) * ld + chunk * 8);
            q0 *= (half_t)0.125f;
            d0 = *(const half8*)(dobase + R(r0) * D + chunk * 8);
        }
        if (r0 + 1 < S && r0 + 1 >= q_min) {
            q1 = *(const half8*)(base + R(r0 + 1) * ld + chunk * 8);
            q1 *= (half_t)0.125f;
            d1 = *(const half8*)(dobase + R(r0 + 1) * D + chunk * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *(half2v*)(T0 + vt_index(r0, chunk * 8 + j)) = (half2v){q0[j], q1[j]};
            *(half2v*)(T1 + vt_index(r0, chunk * 8 + j)) = (half2v){d0[j], d1[j]};
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase 2: dK, dV per key tile
    const int n_kt = (S + 15) >> 4;
    // A wave takes TWO adjacent key tiles at a time (r03): the recomputed S / dP of a query step feed two independent chains (MFMA ->
    // exp / dS -> MFMA) that the scheduler interleaves, and the query-side operands (the Q / dO rows from L2, the transposed Q / dO
    // fragments from LDS) are fetched once for both.  At S = 213 the 14 key tiles are one round of seven waves instead of two rounds of
    // eight (the second one three quarters empty); a tile past the last one works on zero rows and stores nothing.
    constexpr int NT = 2;
    for (int kt0 = wave * NT; kt0 < n_kt; kt0 += NWB * NT) {
        asm volatile("" ::: "memory");
        int kvrow[NT];            // this lane's key (B-operand row / output column) in either tile
        half8 kf[NT][2], vf[NT][2];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            kvrow[u] = (kt0 + u) * 16 + li;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int sw = ((kk * 4 + lg) ^ (lane & 7)) * 8;
                kf[u][kk] = *(const half8*)(Ks + kvrow[u] * 64 + sw);      // (rows S .. SP - 1 of Ks / Vs are zero)
                vf[u][kk] = *(const half8*)(Vs + kvrow[u] * 64 + sw);
            }
        }
        f32x4 dk[NT][4], dv[NT][4];
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) { dk[u][nf] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[u][nf] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const int c_begin = CAUSAL ? (kt0 * 16) / 32 : 0;   // queries before the first key tile see none of these keys
        // A operands of the recomputed S and dP: query rows of Q and dO straight from HBM / L2 (row = q0 + li); the LDS holds the
        // transposed images only.  The rows of 16-query step `it + 1` are requested before step `it` is computed (r03: the loads of a
        // step used to be issued at its top, a full L2 round trip exposed per step and wave -- 14 steps per key tile at S = 213).
        auto load_rows = [&](int it, half8 (&qa_f)[2], half8 (&do_f)[2]) {
            const int qa = (it * 16 + li) < S ? (it * 16 + li) : S - 1;
            const size_t qag = R(qa);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                qa_f[kk] = *(const half8*)(base + qag * ld + (kk * 4 + lg) * 8);
                do_f[kk] = *(const half8*)(dobase + qag * D + (kk * 4 + lg) * 8);
            }
        };
        half8 q_nx[2], do_nx[2];
        load_rows(2 * c_begin, q_nx, do_nx);
#pragma unroll 1
        for (int c = c_begin; c < KVC; ++c) {     // not unrolled: KVC copies of this body cost > 256 VGPRs
            half8 pf[NT], sf[NT];
#pragma unroll
            for (int half_i = 0; half_i < 2; ++half_i) {
                const int q0 = c * 32 + half_i * 16;
                half8 qa_f[2] = {q_nx[0], q_nx[1]}, do_f[2] = {do_nx[0], do_nx[1]};
                if (2 * c + half_i + 1 < 2 * KVC) load_rows(2 * c + half_i + 1, q_nx, do_nx);
                f32x4 s_acc[NT], p_acc[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) { s_acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; p_acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                if (GRIP_ATTNB_PRIO & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    qa_f[kk] *= (half_t)0.125f;
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        s_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa_f[kk], kf[u][kk], s_acc[u], 0, 0, 0);   // S[q][kv]
                        p_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(do_f[kk], vf[u][kk], p_acc[u], 0, 0, 0);   // dP[q][kv]
                    }
                }
                if (GRIP_ATTNB_PRIO & 4) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + lg * 4 + r;
                    const bool qok = q < S && q >= q_min;
                    const float mq = st_m[q < SP ? q : SP - 1], ilq = st_il[q < SP ? q : SP - 1], dq_ = st_d[q < SP ? q : SP - 1];
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        float p = 0.f, ds = 0.f;   // padding rows carry no statistics: keep them exactly zero
                        if (qok && kvrow[u] < S && !(CAUSAL && kvrow[u] > q)) {
                            p = __expf(s_acc[u][r] - mq) * ilq;
                            ds = p * (p_acc[u][r] - dq_);
                        }
                        pf[u][half_i * 4 + r] = (half_t)p;
                        sf[u][half_i * 4 + r] = (half_t)ds;
                    }
                }
            }
            if (GRIP_ATTNB_PRIO & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                const half8 dof = *(const half8*)(T1 + (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8);
                const half8 qtf = *(const half8*)(T0 + (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8);
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    dv[u][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dof, pf[u], dv[u][nf], 0, 0, 0);   // dV^T[dh][kv]
                    dk[u][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qtf, sf[u], dk[u][nf], 0, 0, 0);   // dK^T[dh][kv]
                }
            }
            if (GRIP_ATTNB_PRIO & 8) __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (kvrow[u] < Ps) {
                // a shared key: this sequence's share goes to kv_part [b][key][dK | dV][D] in f32; attn_shared_kv_reduce adds the
                // shares of all sequences in index order
                float* pp = kv_part + (((size_t)b * Ps + kvrow[u]) * 2) * D + h * 64 + lg * 4;
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    *(f32x4*)(pp + nf * 16) = dk[u][nf];
                    *(f32x4*)(pp + D + nf * 16) = dv[u][nf];
                }
            } else if (kvrow[u] < S) {
                half_t* kp = dbase + R(kvrow[u]) * ld + D + lg * 4;
                half_t* vp = kp + D;
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    *(half4*)(kp + nf * 16) = (half4){(half_t)dk[u][nf][0], (half_t)dk[u][nf][1], (half_t)dk[u][nf][2], (half_t)dk[u][nf][3]};
                    *(half4*)(vp + nf * 16) = (half4){(half_t)dv[u][nf][0], (half_t)dv[u][nf][1], (half_t)dv[u][nf][2], (half_t)dv[u][nf][3]};
                }
            }
        }
    }
}

// dqkv[r][D + j] = f16(sum_b kv_part[b][r][j]) for the shared keys r < Ps, j over [dK | dV] (2D columns).  One workgroup of 8 waves per 64 float4
// columns: wave w adds the classes b = w (mod 8) in index order (13 independent loads for 102 classes), the eight partial sums meet in LDS and wave 0
// adds them in wave order -- a fixed summation tree, so the result is deterministic.  (Until r04 one thread walked all classes: 17 workgroups on the
// chip and 13 dependent load batches, 8.7 us per layer of the CoOp step.)
__global__ __launch_bounds__(512) void attn_shared_kv_reduce_kernel(const float* __restrict__ kv_part, half_t* __restrict__ dqkv, int B, int Ps, int D) {
    __shared__ f32x4 part[8][64];
    const int n4 = Ps * 2 * D / 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const size_t stride4 = (size_t)n4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int b = wv;
        for (; b + 56 < B; b += 64) {
            f32x4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = ((const f32x4*)kv_part)[(size_t)(b + 8 * u) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += r[u];
        }
        for (; b < B; b += 8) acc += ((const f32x4*)kv_part)[(size_t)b * stride4 + i];
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv != 0 || i >= n4) return;
#pragma unroll
    for (int u = 1; u < 8; ++u) acc += part[u][lane];
    const int row = (i * 4) / (2 * D), col = i * 4 - row * 2 * D;
    *(half4*)(dqkv + (size_t)row * 3 * D + D + col) = (half4){(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
}

template <int KVC, bool CAUSAL, int NWB>
static int launch_bwd_one(const half_t* qkv, const half_t* o, const half_t* d_out, half_t* dqkv, int B, int S, int H, hipStream_t s, int Ps, float* kv_part) {
    constexpr int SP = KVC * 32;
    constexpr size_t lds = (size_t)4 * SP * 64 * 2 + (size_t)3 * SP * 4;
    static_assert(lds <= 160 * 1024, "attention backward tile does not fit LDS");
    static bool configured = false;
    if (!configured) {
        GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_kernel<KVC, CAUSAL, NWB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL((attn_bwd_kernel<KVC, CAUSAL, NWB>), dim3(B * H), dim3(NWB * 64), lds, s, qkv, o, d_out, dqkv, S, H, Ps, kv_part);
    if (Ps > 0) {
        const int D = H * 64;
        hipLaunchKernelGGL(attn_shared_kv_reduce_kernel, dim3((Ps * 2 * D / 4 + 63) / 64), dim3(512), 0, s, kv_part, dqkv, B, Ps, D);
    }
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

int launch_attention_bwd(const half_t* qkv, const half_t* o, const half_t* d_out, half_t* dqkv, int B, int S, int H, int causal, hipStream_t s,
                         int shared_rows, float* kv_part) {
    const int kvc = (S + 31) / 32;
    GRIP_REQUIRE(shared_rows == 0 || (causal && shared_rows > 0 && shared_rows < S && kv_part && kvc <= 9),
                 "attention backward: the shared-prefix layout needs a causal mask, 0 < shared rows < S <= 288 and the partial buffer");
    if (kvc > 9) return launch_attention_bwd_tiled(qkv, o, d_out, dqkv, B, S, H, causal, s);    // S > 288: block-tiled kernel (attention_bwd_tiled.hip)
    GRIP_REQUIRE(S >= 1, "attention backward: sequence length %d unsupported", S);
#define GRIP_ATTN(N)                                                                        \
    if (kvc <= N) return causal ? launch_bwd_one<N, true, (N >= 4 ? 8 : 4)>(qkv, o, d_out, dqkv, B, S, H, s, shared_rows, kv_part)  \
                                : launch_bwd_one<N, false, (N >= 4 ? 8 : 4)>(qkv, o, d_out, dqkv, B, S, H, s, 0, nullptr);
    GRIP_ATTN(1) GRIP_ATTN(2) GRIP_ATTN(3) GRIP_ATTN(4) GRIP_ATTN(5) GRIP_ATTN(6) GRIP_ATTN(7) GRIP_ATTN(8) GRIP_ATTN(9)
#undef GRIP_ATTN
    return GRIP_ERR_ARG;
}
