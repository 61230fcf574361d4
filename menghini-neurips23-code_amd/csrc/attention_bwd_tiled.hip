// Backward of the fused self-attention for LONG sequences (288 < S <= 672; visual / multimodal prompts on ViT-L/14@336px give
// S = 581 / 593: CustomVisionTransformer inserts the prompt tokens into the 577-token sequence, models/clip_encoders.py:148-155).
// attention_bwd.hip keeps K, V and two transposed images of one whole head in LDS (4 x S x 128 B), which stops at S = 288; this
// kernel walks the head in BLOCKS of BS = 32 * KB rows and keeps only one block of each operand resident:
//
//   phase 0  row statistics over ALL keys: per key block stage K, every wave updates the running (max, sum) of its query
//            tiles (registers); delta = rowsum(dO * O).  Results go to LDS (3 floats per query row).
//   phase 1  dQ: per key block stage K, V and the blocked transposed image of K; every wave recomputes P^T for its query tiles
//            against the block, dP^T = V dO^T, dS^T = P^T (dP^T - delta), and accumulates dQ^T += K^T dS^T in registers
//            (<= 5 query tiles x 16 accumulator registers per wave).
//   phase 2  dK, dV: per key block each wave owns <= 2 key tiles (K / V fragments straight from HBM, accumulators in registers)
//            and sweeps the QUERY blocks: stage the transposed images of Q/8 and dO of the block, recompute S and dP in the
//            [query][key] orientation, dV^T += dO^T P, dK^T += (Q/8)^T dS.
// Same MFMA formulation (v_mfma_f32_16x16x32_f16), LDS images (vt_index) and f32 statistics as attention_bwd.hip; probabilities
// are recomputed, nothing but O was saved by the forward.  One workgroup of 8 waves per (image, head).
#include <math.h>

#include "common.h"

template <int KB, bool CAUSAL, int NWB>
__global__ __launch_bounds__(NWB * 64) void attn_bwd_tiled_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ o_saved,
                                                                  const half_t* __restrict__ d_out, half_t* __restrict__ dqkv, int S, int H) {
    constexpr int BS = KB * 32;                       // rows per block
    constexpr int SMAX = 3 * BS;                      // longest padded sequence (three blocks)
    constexpr int NQT = (SMAX / 16 + NWB - 1) / NWB;  // query tiles a wave can own
    constexpr int NKT = (2 * KB + NWB - 1) / NWB;     // key tiles of one block a wave can own
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Ks = (half_t*)smem;          // [BS][64] swizzled rows
    half_t* Vs = Ks + BS * 64;           // [BS][64] swizzled rows
    half_t* T0 = Vs + BS * 64;           // blocked transposed image: K (phase 1), Q/8 (phase 2)
    half_t* T1 = T0 + BS * 64;           // blocked transposed image of dO (phase 2)
    float* st_m = (float*)(T1 + BS * 64);    // [SMAX] row max
    float* st_il = st_m + SMAX;              // [SMAX] 1 / row sum
    float* st_d = st_il + SMAX;              // [SMAX] delta

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int D = H * 64;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t ld = (size_t)3 * D;
    const half_t* base = qkv + (size_t)b * S * ld + h * 64;
    const half_t* obase = o_saved + (size_t)b * S * D + h * 64;
    const half_t* dobase = d_out + (size_t)b * S * D + h * 64;
    half_t* dbase = dqkv + (size_t)b * S * ld + h * 64;
    const int n_qt = (S + 15) >> 4;
    const int n_blk = (S + BS - 1) / BS;

    // rows [r0, r0 + BS) of K (and V) -> swizzled LDS rows; optionally the blocked transposed image of K
    auto stage_kv = [&](int r0, bool with_v, bool with_t) {
        for (int idx = tid; idx < BS * 8; idx += NWB * 64) {
            const int row = idx >> 3, chunk = idx & 7, gr = r0 + row;
            half8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gr < S) {
                kv = *(const half8*)(base + gr * ld + D + chunk * 8);
                if (with_v) vv = *(const half8*)(base + gr * ld + 2 * D + chunk * 8);
            }
            const int sw = (chunk ^ (row & 7)) * 8;
            *(half8*)(Ks + row * 64 + sw) = kv;
            if (with_v) *(half8*)(Vs + row * 64 + sw) = vv;
            if (with_t) {
#pragma unroll
                for (int j = 0; j < 8; ++j) T0[vt_index(row, chunk * 8 + j)] = kv[j];
            }
        }
    };
    auto load_qf = [&](int qr, half8 (&qf)[2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[kk] = *(const half8*)(base + qr * ld + (kk * 4 + lg) * 8);
            qf[kk] *= (half_t)0.125f;
        }
    };
    // S^T tile t of the resident key block against the wave's query fragments, masked: acc[r] = score of key k0 + t*16 + lg*4 + r
    auto score_tile = [&](int t, int k0, int qrow, const half8 (&qf)[2]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const half8 kf = *(const half8*)(Ks + (t * 16 + li) * 64 + (((kk * 4 + lg) ^ (lane & 7)) * 8));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kv = k0 + t * 16 + lg * 4 + r;
            if (kv >= S || (CAUSAL && kv > qrow)) acc[r] = -INFINITY;
        }
        return acc;
    };

    // ------------------------------------------------------------------ phase 0: row statistics over all keys
    {
        float run_m[NQT], run_l[NQT];
#pragma unroll
        for (int j = 0; j < NQT; ++j) { run_m[j] = -INFINITY; run_l[j] = 0.f; }
        for (int kb = 0; kb < n_blk; ++kb) {
            __syncthreads();
            stage_kv(kb * BS, false, false);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NQT; ++j) {
                const int qt = wave + j * NWB;
                if (qt >= n_qt) continue;
                asm volatile("" ::: "memory");
                const int qrow = qt * 16 + li;
                half8 qf[2];
                load_qf(qrow < S ? qrow : S - 1, qf);
                f32x4 sc[2 * KB];
                float m = run_m[j];
#pragma unroll
                for (int t = 0; t < 2 * KB; ++t) {
                    sc[t] = score_tile(t, kb * BS, qrow, qf);
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[t][r]);
                }
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));           // finite from block 0 on: key 0 is visible to every query
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 2 * KB; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum += __expf(sc[t][r] - m);
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                run_l[j] = run_l[j] * __expf(run_m[j] - m) + sum;
                run_m[j] = m;
            }
        }
#pragma unroll
        for (int j = 0; j < NQT; ++j) {
            const int qt = wave + j * NWB;
            if (qt >= n_qt) continue;
            const int qrow = qt * 16 + li;
            const int qr = qrow < S ? qrow : S - 1;
            float dl = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8 dof = *(const half8*)(dobase + (size_t)qr * D + (kk * 4 + lg) * 8);
                const half8 of = *(const half8*)(obase + (size_t)qr * D + (kk * 4 + lg) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)dof[e] * (float)of[e];
            }
            dl += __shfl_xor(dl, 16);
            dl += __shfl_xor(dl, 32);
            if (lg == 0) { st_m[qrow] = run_m[j]; st_il[qrow] = 1.0f / run_l[j]; st_d[qrow] = dl; }
        }
    }

    // ------------------------------------------------------------------ phase 1: dQ, accumulated over the key blocks
    {
        f32x4 dq[NQT][4];
#pragma unroll
        for (int j = 0; j < NQT; ++j)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) dq[j][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < n_blk; ++kb) {
            __syncthreads();               // (first pass: also publishes the statistics of phase 0)
            stage_kv(kb * BS, true, true);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NQT; ++j) {
                const int qt = wave + j * NWB;
                if (qt >= n_qt) continue;
                if (CAUSAL && kb * BS > qt * 16 + 15) continue;        // the whole key block lies after this query tile
                asm volatile("" ::: "memory");
                const int qrow = qt * 16 + li;
                const int qr = qrow < S ? qrow : S - 1;
                half8 qf[2], dof[2];
                load_qf(qr, qf);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) dof[kk] = *(const half8*)(dobase + (size_t)qr * D + (kk * 4 + lg) * 8);
                const float m = st_m[qrow], il = st_il[qrow], dl = st_d[qrow];
                f32x4 sc[2 * KB];
#pragma unroll
                for (int t = 0; t < 2 * KB; ++t) {
                    sc[t] = score_tile(t, kb * BS, qrow, qf);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[t][r] = __expf(sc[t][r] - m);       // masked keys: exp(-inf) = 0
                }
#pragma unroll
                for (int t = 0; t < 2 * KB; ++t) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const half8 vf = *(const half8*)(Vs + (t * 16 + li) * 64 + (((kk * 4 + lg) ^ (lane & 7)) * 8));
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, dof[kk], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[t][r] = sc[t][r] * il * (acc[r] - dl);
                }
#pragma unroll
                for (int c = 0; c < KB; ++c) {
                    const half8 sf = {(half_t)sc[2 * c][0], (half_t)sc[2 * c][1], (half_t)sc[2 * c][2], (half_t)sc[2 * c][3],
                                      (half_t)sc[2 * c + 1][0], (half_t)sc[2 * c + 1][1], (half_t)sc[2 * c + 1][2], (half_t)sc[2 * c + 1][3]};
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const half8 kf = *(const half8*)(T0 + (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8);
                        dq[j][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, sf, dq[j][nf], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NQT; ++j) {
            const int qt = wave + j * NWB;
            const int qrow = qt * 16 + li;
            if (qt < n_qt && qrow < S) {
                half_t* op = dbase + qrow * ld + lg * 4;
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    const f32x4 v = dq[j][nf] * 0.125f;
                    *(half4*)(op + nf * 16) = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                }
            }
        }
    }

    // ------------------------------------------------------------------ phase 2: dK, dV per key block, swept over the query blocks
    for (int kb = 0; kb < n_blk; ++kb) {
        half8 kf[NKT][2], vf[NKT][2];
        f32x4 dk[NKT][4], dv[NKT][4];
        bool own[NKT];
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int ktl = wave + i * NWB;
            own[i] = ktl < 2 * KB && kb * BS + ktl * 16 < S;
            const int kvrow = kb * BS + ktl * 16 + li;
            const int kr = kvrow < S ? kvrow : S - 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                kf[i][kk] = *(const half8*)(base + kr * ld + D + (kk * 4 + lg) * 8);
                vf[i][kk] = *(const half8*)(base + kr * ld + 2 * D + (kk * 4 + lg) * 8);
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) { dk[i][nf] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i][nf] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
        for (int qb = CAUSAL ? kb : 0; qb < n_blk; ++qb) {     // causal: query blocks before the key block see none of its keys
            __syncthreads();
            for (int idx = tid; idx < BS * 8; idx += NWB * 64) {
                const int row = idx >> 3, chunk = idx & 7, gr = qb * BS + row;
                half8 qv = {0, 0, 0, 0, 0, 0, 0, 0}, dvv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (gr < S) {
                    qv = *(const half8*)(base + gr * ld + chunk * 8);
                    qv *= (half_t)0.125f;
                    dvv = *(const half8*)(dobase + (size_t)gr * D + chunk * 8);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    T0[vt_index(row, chunk * 8 + j)] = qv[j];
                    T1[vt_index(row, chunk * 8 + j)] = dvv[j];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NKT; ++i) {
                if (!own[i]) continue;
                const int kvrow = kb * BS + (wave + i * NWB) * 16 + li;
#pragma unroll 1
                for (int c = 0; c < KB; ++c) {          // not unrolled: KB copies of this body cost > 256 VGPRs
                    if (qb * BS + c * 32 >= S) break;
                    half8 pf, sf;
#pragma unroll
                    for (int half_i = 0; half_i < 2; ++half_i) {
                        const int q0 = qb * BS + c * 32 + half_i * 16;
                        const int qa = (q0 + li) < S ? (q0 + li) : S - 1;
                        f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            half8 qa_f = *(const half8*)(base + qa * ld + (kk * 4 + lg) * 8);
                            qa_f *= (half_t)0.125f;
                            const half8 do_f = *(const half8*)(dobase + (size_t)qa * D + (kk * 4 + lg) * 8);
                            s_acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa_f, kf[i][kk], s_acc, 0, 0, 0);   // S[q][kv]
                            p_acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(do_f, vf[i][kk], p_acc, 0, 0, 0);   // dP[q][kv]
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = q0 + lg * 4 + r;
                            float p = 0.f, ds = 0.f;   // padding rows carry no statistics: keep them exactly zero
                            if (q < S && kvrow < S && !(CAUSAL && kvrow > q)) {
                                p = __expf(s_acc[r] - st_m[q]) * st_il[q];
                                ds = p * (p_acc[r] - st_d[q]);
                            }
                            pf[half_i * 4 + r] = (half_t)p;
                            sf[half_i * 4 + r] = (half_t)ds;
                        }
                    }
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const half8 dof = *(const half8*)(T1 + (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8);
                        dv[i][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dof, pf, dv[i][nf], 0, 0, 0);   // dV^T[dh][kv]
                        const half8 qtf = *(const half8*)(T0 + (c * 4 + nf) * 512 + (lg * 16 + (li ^ lg)) * 8);
                        dk[i][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qtf, sf, dk[i][nf], 0, 0, 0);   // dK^T[dh][kv]
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int kvrow = kb * BS + (wave + i * NWB) * 16 + li;
            if (own[i] && kvrow < S) {
                half_t* kp = dbase + kvrow * ld + D + lg * 4;
                half_t* vp = dbase + kvrow * ld + 2 * D + lg * 4;
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    *(half4*)(kp + nf * 16) = (half4){(half_t)dk[i][nf][0], (half_t)dk[i][nf][1], (half_t)dk[i][nf][2], (half_t)dk[i][nf][3]};
                    *(half4*)(vp + nf * 16) = (half4){(half_t)dv[i][nf][0], (half_t)dv[i][nf][1], (half_t)dv[i][nf][2], (half_t)dv[i][nf][3]};
                }
            }
        }
    }
}

template <int KB, bool CAUSAL>
static int launch_tiled(const half_t* qkv, const half_t* o, const half_t* d_out, half_t* dqkv, int B, int S, int H, hipStream_t s) {
    constexpr int BS = KB * 32, NWB = 8;
    constexpr size_t lds = (size_t)4 * BS * 64 * 2 + (size_t)3 * (3 * BS) * 4;
    static_assert(lds <= 160 * 1024, "tiled attention backward does not fit LDS");
    static bool configured = false;
    if (!configured) {
        GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_tiled_kernel<KB, CAUSAL, NWB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL((attn_bwd_tiled_kernel<KB, CAUSAL, NWB>), dim3(B * H), dim3(NWB * 64), lds, s, qkv, o, d_out, dqkv, S, H);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// S in (288, 672]: blocks of 224 rows (three at most).
int launch_attention_bwd_tiled(const half_t* qkv, const half_t* o, const half_t* d_out, half_t* dqkv, int B, int S, int H, int causal, hipStream_t s) {
    GRIP_REQUIRE(S >= 1 && S <= 672, "attention backward (tiled): sequence length %d unsupported (max 672)", S);
    return causal ? launch_tiled<7, true>(qkv, o, d_out, dqkv, B, S, H, s) : launch_tiled<7, false>(qkv, o, d_out, dqkv, B, S, H, s);
}
