// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, MFMA 16x16x32 f16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "host_common.h"

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef half_t half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Residual stream element type.  f16, as the reference's own GPU path keeps it (clip.load on a GPU converts the model to
// f16): every add into it happens in f32 inside a GEMM epilogue and is rounded once; LayerNorm statistics are f32.
// Halves the bytes of the HBM-bound LayerNorm kernels and of the out-proj / c_proj epilogues.
typedef half_t resid_t;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

#define GRIP_CHECK_HIP(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            grip_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return GRIP_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

int grip_cu_budget();      // grip_set_cu_budget (tower.hip): CUs the persistent kernels may size their grids to, 0 = all

static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------
// GEMM: C[M,N] = epilogue(A[M,K] * W[N,K]^T).  A and W are f16, K-contiguous; accumulate f32.
enum GemmEpi {
    EPI_F32 = 0,             // out_f32 = acc
    EPI_BIAS_F16 = 1,        // out_f16 = acc + bias
    EPI_BIAS_GELU_F16 = 2,   // out_f16 = quickgelu(acc + bias); if out2 != null, out2_f16 = acc + bias (pre-activation)
    EPI_BIAS_RESID = 3,      // out_resid = resid + acc + bias   (residual stream, resid_t)
    EPI_F16 = 4,             // out_f16 = acc
    EPI_GELUGRAD_F16 = 5,    // out_f16 = acc * quickgelu'(aux_f16)       (backward of c_fc activation)
    EPI_F32_SCALE = 6,       // out_f32 = acc * scalar
    // LayerNorm folded into the GEMM that consumes it (A = the RAW residual stream, W = gamma-scaled weights W' = f16(gamma o W)):
    //   LN(x) W^T + b  =  rstd_r * (x W'^T - mean_r * colsum(W')) + (W beta + b)
    // rowstat[r] = (mean_r, rstd_r), colsum[n] = sum_k W'[n][k], bias[n] = (W beta + b)[n]
    EPI_LNFOLD_F16 = 7,      // out_f16 = rstd * (acc - mean * colsum) + bias
    EPI_LNFOLD_GELU_F16 = 8, // out_f16 = quickgelu(that); if out2 != null, out2_f16 = that (pre-activation)
    EPI_BIAS_RESID_STATS = 9,// EPI_BIAS_RESID + the row statistics (GemmArgs.stat_part); chosen by the launcher, never passed in by callers
    EPI_COUNT = 10
};

#ifdef __HIPCC__
// LDS image of a [key][64] matrix consumed TRANSPOSED as an MFMA A operand (16 head dims x 32 keys per fragment): V in the
// attention forward, K / Q / dO in the backward.  Per (32-key chunk c, 16-dim block
// nf) one 1 KiB block in which lane (li, lg) of the consuming wave finds its 8 halfs -- keys c*32 + lg*4 + {0..3} and
// c*32 + 16 + lg*4 + {0..3} of head dim nf*16 + li -- at 16-byte unit lg*16 + (li ^ lg): one ds_read_b128 per MFMA,
// conflict-free in the instruction's four 16-lane groups, and the staging writes (32 lanes = 32 key pairs of one dim)
// spread over 16 banks.  (The earlier [dim][key] image needed two ds_read_b64, which the compiler fuses into a
// ds_read2_b64: 8 LDS cycles + 2-way conflicts instead of 4.)
__device__ __forceinline__ int vt_index(int key, int d) {
    const int c = key >> 5, kk = key & 31, lg = (kk >> 2) & 3, e = (kk >> 4) * 4 + (kk & 3);
    return (((c * 4 + (d >> 4)) * 64 + lg * 16 + ((d & 15) ^ lg)) * 8) + e;
}

// Row-kernel helpers (rowops.hip): one wave owns a row, lane l holds float4 index l + 64*i; statistics are two in-register passes + a 6-step xor reduction.
#define LN_EPS 1e-5f
#define MAX_NV 8  // d <= 2048

// Sum over the 64 lanes of a wave, returned in every lane, without touching the LDS: four DPP steps inside each 16-lane row
// (xor 1, xor 2, half-mirror = the other quad, mirror = the other eight), then the four row sums are read out as scalars.
// (__shfl_xor compiles to ds_bpermute_b32: six dependent LDS round trips per reduction, the whole cost of a row kernel
// when a wave has a SIMD to itself.)  Needs all 64 lanes active: callers branch per wave, never per lane, before it.
__device__ __forceinline__ float wave_sum(float v) {
#define GRIP_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    GRIP_DPP_ADD(0xB1);     // quad_perm [1,0,3,2]
    GRIP_DPP_ADD(0x4E);     // quad_perm [2,3,0,1]
    GRIP_DPP_ADD(0x141);    // row_half_mirror
    GRIP_DPP_ADD(0x140);    // row_mirror
#undef GRIP_DPP_ADD
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// Sum over each 16-lane row of the wave (the four DPP steps above), returned in every lane of the row.
__device__ __forceinline__ float row16_sum(float v) {
#define GRIP_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    GRIP_DPP_ADD(0xB1);
    GRIP_DPP_ADD(0x4E);
    GRIP_DPP_ADD(0x141);
    GRIP_DPP_ADD(0x140);
#undef GRIP_DPP_ADD
    return v;
}

__device__ __forceinline__ f32x4 load4(const float* p, int i) { return ((const f32x4*)p)[i]; }
__device__ __forceinline__ f32x4 load4(const half_t* p, int i) {
    const half4 h = ((const half4*)p)[i];
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ void store4(float* p, int i, f32x4 v) { ((f32x4*)p)[i] = v; }
__device__ __forceinline__ void store4(half_t* p, int i, f32x4 v) { ((half4*)p)[i] = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; }
// Split layout of the precision-2 tier (gemm_split.hip): per row and 32 consecutive columns one 128-byte line [32 x hi | 32 x lo'],
// hi = f16(x), lo' = f16((x - hi) * GRIP_SPLIT_LO_SCALE).  SplitRow tags an output row pointer of that layout for store4.
#ifndef GRIP_SPLIT_LO_SCALE
#define GRIP_SPLIT_LO_SCALE 1       // lo' = f16((x - hi) * this): 1 = unscaled (single-accumulator kernels; needs unflushed f16 subnormals in the MFMA),
#endif                              // 2048 = the two-accumulator forms (nothing subnormal where hi is normal)
struct SplitRow { half_t* p; };
__device__ __forceinline__ void split_f16x4(f32x4 v, half4& hi, half4& lo) {
    // hi and lo must come from ONE value.  Left to itself (HIP compiles with fp-contract=fast) the compiler fuses the subtraction with the multiply
    // that produced v (v_fma_mix: lo = f16(x r - h) from the exact product) while a second copy of hi comes from the f32-rounded product
    // (v_cvt_pk_f16_f32): at an f16 rounding tie of fl32(x r) the two disagree by an f16 ulp and hi + lo is off by 2^-11 (seen in the GELU
    // epilogue: 2 of 65 536 elements).  The empty asm makes v opaque, so nothing upstream can be folded into either use.
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        asm volatile("" : "+v"(x));
        const half_t h = (half_t)x;
        hi[e] = h;
        lo[e] = (half_t)((x - (float)h) * (float)GRIP_SPLIT_LO_SCALE);
    }
}
__device__ __forceinline__ void store4(SplitRow r, int i, f32x4 v) {       // columns 4 i .. 4 i + 3 of the row
    half4 hi, lo;
    split_f16x4(v, hi, lo);
    half_t* o = r.p + (i >> 3) * 64 + (i & 7) * 4;
    *(half4*)o = hi;
    *(half4*)(o + 32) = lo;
}

// The LayerNorm arithmetic is compiled with floating-point contraction OFF: it is inlined into several kernels (stand-alone,
// gather, sequence assembly) and every one of them must round identically whatever the surrounding code lets the
// optimiser fuse.
template <int NV>
__device__ __forceinline__ void ln_normalize(f32x4 (&v)[NV], int lane, int d4, int d, float& mean, float& rstd) {
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < d4) {
            f32x4 c = v[i] - mean;
            q += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
        }
    rstd = rsqrtf(wave_sum(q) / (float)d + LN_EPS);
}
__device__ __forceinline__ f32x4 ln_apply(f32x4 v, float mean, float rstd, f32x4 g, f32x4 b) {
#pragma clang fp contract(off)
    return (v - mean) * rstd * g + b;
}
#endif

struct GemmArgs {
    const void* A;      // [Mpad, K], lda = K; rows >= M may hold anything finite or not (never stored).  f16, or f32 when f32 != 0
    const void* W;      // [N, K], same element type as A
    int M, N, K;
    int variant;        // 0 = let the launcher choose the tile shape; 1/2/3 force 128x128 / 256x256 / 256x128 (tests, tuning)
    int64_t m_pad;      // rows allocated for A (>= M); the 256-row tile is used only when m_pad covers it
    const float* bias;  // [N] or null
    const void* resid;  // [M, ldc] residual stream (EPI_BIAS_RESID), activation type
    const void* aux;    // [M, ldc] activation type (EPI_GELUGRAD_F16)
    void* out;          // [M, ldc] activation type or f32
    void* out2;         // optional second output
    int ldc;
    float scalar;
    int f32;            // 1 = exact mode: A, W, resid and every activation output are f32 (gemm_f32.hip, v_mfma_f32_16x16x4_f32)
                        // 2 = split-f16 tier: A and W in the split layout, resid / bias / output f32 (gemm_split.hip)
    // LayerNorm statistics travelling with the residual stream (f16 towers):
    float* stat_part;      // EPI_BIAS_RESID, optional: [N/64, M, 2] per-row partial (sum, sum of squares) of the values written, one pair
                           // per 64-column wave tile; ln_stats_finalize turns them into rowstat for the consuming GEMM
    const float* rowstat;  // EPI_LNFOLD_*: [M, 2] (mean, rstd) of A's rows
    const float* stat_in;  // EPI_LNFOLD_*, optional: the producer's stat_part array ([stat_parts, M, 2]) when nobody has finalised it yet: every kernel adds the
    int stat_parts;        // pairs itself (gemm.hip, row_stat: same order and arithmetic as ln_stats_finalize), so a prompt step pays no launch for it
    const float* colsum;   // EPI_LNFOLD_*: [N]
    // Split-K (EPI_F32 on the small-M kernels only): ksplit > 1 launches ksplit workgroups per output tile, each contracting
    // K/ksplit and writing its partial product to out + split * split_stride (floats); the consumer sums the partials
    // in a fixed order (launch_ln_bwd_add).  gemm_pick_ksplit chooses the factor.
    int ksplit;
    int64_t split_stride;
    // Cooperative split-K (EPI_BIAS_RESID[_STATS] on gemm_ringw_kernel, 64-row tiles): ksplit > 1 with these two set.  Every split writes its partial
    // tile to coop_scratch ([tile][split][wave][2048] f32), the LAST wave to arrive at a (tile, wave) counter adds the partials in split order (a fixed
    // order whoever arrives last: deterministic) and runs the epilogue, then puts the counter back to zero.  For the K = 4 d GEMM of a few hundred rows
    // (a text tower's c_proj: 28 tiles walking 32 K slices at the ~80 GB/s one CU stages).  coop_counter must be zero before the first such launch.
    float* coop_scratch;
    int* coop_counter;
    // One-tile-per-workgroup kernels: tile row tm starts its K walk rot_rows * tm slices further on (0: every tile row starts where its
    // column panel says).  Changes the summation order with the tile row, so only train-mode launches may set it (the inference
    // forwards stay bit-identical under any chunking).
    int rot_rows;
    half_t* resid_lo;   // EPI_BIAS_RESID[_STATS], f16 kernels, optional: the COMPENSATED residual stream of the screen (r06) -- the stream value of an element is
                        // resid + resid_lo (two f16 numbers, ~22 mantissa bits): the epilogue adds both in f32, stores hi = f16(v) to `out` and lo = f16(v - hi) back
                        // to resid_lo (same index as `out`: the stream is updated in place).  The next GEMM still multiplies the hi part alone (one operand
                        // rounding, which does not accumulate); what no longer accumulates is the rounding of the stream itself, 24 of them per ViT-B/16 image
    int w_exact;        // f32 == 2 only: every element of W is an f16 number (zero lo parts: fp16 checkpoints) -- gemm_split.hip drops the a_hi w_lo product
    int ablate;         // developer builds only (-DGRIP_ABLATE, tools/mlp_ablation.sh): bit 0 = the c_fc epilogue issues no global store, bit 1 = the K = 4 d
                        // residual GEMM reads its A operand from a 31-MB window (Infinity-Cache resident); timing experiments, results are wrong by design
};
int gemm_pick_ksplit(int M, int N, int K);
int gemm_pick_coop_split(int M, int N, int K);   // split factor of the cooperative form (1 = not worth it / not applicable)

int launch_gemm(int epi, const GemmArgs& a, hipStream_t s);
int launch_gemm_f32(int epi, const GemmArgs& a, hipStream_t s);
// split-f16 tier (gemm_split.hip; GemmArgs.f32 == 2): A and W in the split layout ([32 x hi | 32 x lo'] f16 per 32 consecutive k, row pitch
// 4 K bytes), f32 bias / residual / output -- EPI_BIAS_GELU_F16 writes its output in the split layout (it feeds the next split GEMM)
int launch_gemm_split(int epi, const GemmArgs& a, hipStream_t s);
int launch_split_rows(const float* x, void* out, int64_t rows, int K, int64_t ld_in, hipStream_t s, int is_weight = 0, int* overflow_flag = nullptr,
                      int* lo_nonzero_flag = nullptr);   // weights carry gemm_split_weight_scale(); lo_nonzero_flag: set when an element is not an f16 number
float gemm_split_weight_scale();

// Activation buffers are f16 (default) or f32 (exact mode): the row kernels take untyped pointers plus the flag.  f32 == 2 (split-f16
// tier): inputs f32 as in exact mode; launch_layernorm_f16 writes its OUTPUT (a GEMM operand) in the split layout of gemm_split.hip.
// row-wise kernels (rowops.hip)
int launch_im2col(const void* images, int images_f16, void* out, int out_f32, int B, int R, int patch, int Kpad, hipStream_t s);
int launch_vit_assemble_ln(const float* patch_out, const float* cls, const float* pos, const float* prefix, int P,   /* x_lo (last argument, optional): the lo parts of a compensated stream */
                           const float* gamma, const float* beta, void* x, int f32, float* rowstat, int B, int G2, int d, hipStream_t s, half_t* x_lo = nullptr);
// rowstat [M, 2] = (mean, rstd) of every row from the [M, parts, 2] partial sums the residual GEMM epilogues emit
int launch_ln_stats_finalize(const float* stat_part, int parts, float* rowstat, int M, int d, hipStream_t s);
// W' = f16(gamma o W) [N, K]; colsum[n] = sum_k W'[n][k]; bias_out[n] = bias[n] + sum_k beta[k] W[n][k]
int launch_ln_fold_weights(const half_t* W, const float* gamma, const float* beta, const float* bias, half_t* Wg, float* colsum, float* bias_out,
                           int N, int K, hipStream_t s);
int launch_layernorm_f16(const void* x, const float* gamma, const float* beta, void* out, int f32, int M, int d, hipStream_t s);
int launch_gather_ln_f16(const void* x, const int32_t* row_index, int row_stride, const float* gamma, const float* beta,
                         void* out, int f32, int n_rows, int d, hipStream_t s);
int launch_text_embed(const int32_t* token_ids, int ld_ids, const float* tok_emb, const float* pos, const float* prefix, int P,
                      int prefix_classes, void* x, int f32, float* rowstat, int C, int T, int d, int vocab, hipStream_t s, int shared_rows = 0);
int launch_transpose(const void* in, void* out, int f32, int rows, int cols, int ld_in, hipStream_t s);

// attention (attention.hip / attention_f32.hip): qkv [B*S, 3*D] -> out [B*S, D]
// Row addressing shared by the small-S forward and backward kernels.  Plain layout: row r of sequence b is b*S + r.
// Shared-prefix layout (text tower with ONE learned context for every class, Ps = 1 + n_prefix > 0): the first Ps positions
// (SOT + context) are the same tokens at the same positions for every class and the mask is causal, so their activations are
// identical for every class in every layer; the stream holds them ONCE (rows 0 .. Ps-1) followed by the L = S - Ps
// class-specific positions of each class (row Ps + b*L + (r - Ps)).  Sequence b attends to the shared keys and its own.
#ifdef __HIPCC__
__device__ __forceinline__ size_t seq_row(int b, int r, int S, int Ps) {
    return r < Ps ? (size_t)r : (size_t)Ps + (size_t)b * (S - Ps) + (r - Ps);
}
#endif

// shared_rows > 0: shared-prefix row layout (attention.hip, seq_row)
int launch_attention_fwd(const half_t* qkv, half_t* out, int B, int S, int H, int causal, hipStream_t s, int shared_rows = 0);
// train != 0: the four-waves-per-(sequence, head) form of the prompt steps (another summation order: the MODE picks it, never the batch size)
int launch_attention_row(const half_t* qkv, const half_t* qrows, const int32_t* row_index, half_t* out, int B, int S, int H, int causal, hipStream_t s, int train = 0);
int launch_gather_rows(const half_t* x, const int32_t* row_index, int row_stride, half_t* out, int n_rows, int d, hipStream_t s);
// backward of launch_attention_row: o_rows / do_rows [B, H*64] (the forward's output rows and their gradient) -> the whole packed dqkv [B, S, 3, H*64]
int launch_attention_row_bwd(const half_t* qkv, const half_t* o_rows, const half_t* do_rows, const int32_t* row_index, half_t* dqkv, int B, int S, int H, int causal,
                             hipStream_t s);
// split_out != 0: `out` is written in the split layout of gemm_split.hip (it feeds the out-proj GEMM of a precision-2 tower)
int launch_attention_fwd_f32(const float* qkv, float* out, int B, int S, int H, int causal, hipStream_t s, int split_out = 0);
// one query row per sequence (qrows [B, H*64] f32: the projected rows), K / V from the packed f32 qkv; out [B, H*64] f32 or, split_out != 0, the split layout
int launch_attention_row_f32(const float* qkv, const float* qrows, const int32_t* row_index, float* out, int B, int S, int H, int causal, hipStream_t s, int split_out = 0);
// rows of 4-byte elements (f32, or the split-f16 layout): out[b] = x[b * row_stride + (row_index ? row_index[b] : 0)]
int launch_gather_rows4(const void* x, const int32_t* row_index, int row_stride, void* out, int n_rows, int d, hipStream_t s);
// precision-2 towers, S <= 320 (attention_split.hip): f32 qkv in, both products as three f16 MFMAs on hi / lo' operand pairs, split-layout out
bool attention_split_supported(int S);
int launch_attention_fwd_split(const float* qkv, void* out, int B, int S, int H, int causal, hipStream_t s);
// shared_rows > 0: shared-prefix layout; kv_part [B, shared_rows, 2, H*64] f32 scratch for the per-sequence dK / dV of the shared keys
int launch_attention_bwd(const half_t* qkv, const half_t* o, const half_t* d_out, half_t* dqkv, int B, int S, int H, int causal, hipStream_t s,
                         int shared_rows = 0, float* kv_part = nullptr);
int launch_attention_bwd_tiled(const half_t* qkv, const half_t* o, const half_t* d_out, half_t* dqkv, int B, int S, int H, int causal, hipStream_t s);
// backward row kernels (rowops_bwd.hip)
int launch_layernorm_f16_from_f32(const float* x, const float* gamma, const float* beta, half_t* out, int M, int d, hipStream_t s);
int launch_ln_bwd_add(const resid_t* x, const float* dln, int parts, int64_t part_stride, const float* gamma, float* dx, half_t* dxh, int M, int d, hipStream_t s);
// ln_bwd_add into a stream gradient that is rows_add[b] at row b * stride + index[b] and zero elsewhere (writes dx / dxh, reads neither)
int launch_ln_bwd_init(const resid_t* x, const float* dln, int parts, int64_t part_stride, const float* gamma, const float* rows_add, const int32_t* index, int stride,
                       float* dx, half_t* dxh, int M, int d, hipStream_t s);
int launch_ln_bwd_scatter(const resid_t* x, const float* dy, const int32_t* index, int stride, const float* gamma, float* dx, half_t* dxh,
                          int n, int d, hipStream_t s);
// ... with the zero fill of every other row of dx / dxh [M, d] folded in (sequences start at row `first`)
int launch_ln_bwd_scatter_fill(const resid_t* x, const float* dy, const int32_t* index, int stride, int first, const float* gamma, float* dx, half_t* dxh,
                               int n, int M, int d, hipStream_t s);
int launch_vit_prefix_grad(const float* dx, const float* prefix, const float* gamma, const float* scale, float* grad, int B, int S, int P, int d, hipStream_t s);
int launch_text_prefix_grad(const float* dx, const float* scale, float* grad, int C, int T, int P, int prefix_classes, int d, hipStream_t s);
int launch_grad_scale_cast(const float* g, half_t* g16, float* scale, int n, hipStream_t s);
