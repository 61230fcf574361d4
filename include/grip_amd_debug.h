/* Test hooks and the in-library GEMM profiler of libgrip_amd.so.  NOT part of the drop-in ABI (include/grip_amd.h): nothing
 * in the reference binds to these.  They exist so that tests/test_gpu_kernels.py can check each kernel against a PyTorch fp32
 * reference through exactly the launchers the towers use, and so that bench.py can time the GEMM launches of the timed
 * region with HIP events on the launch stream (roofline block).  All return 0 or a GRIP_ERR_* code (grip_last_error()). */
#ifndef GRIP_AMD_DEBUG_H
#define GRIP_AMD_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* out[M,N] = epilogue(A[M,K] W[N,K]^T).  epi: 0 f32 out, 1 +bias -> f16, 2 +bias, QuickGELU -> f16 (out2 = pre-activation or
 * NULL), 3 +bias +resid(f16) -> f16, 4 f16, 5 * QuickGELU'(aux) -> f16, 6 f32 * scalar.  A has m_pad >= M rows allocated.
 * variant: 0 = launcher's choice, 1..6 = a specific tile kernel (csrc/gemm.hip); 7 = the exact-mode kernel (csrc/gemm_f32.hip):
 * A, W, resid and every output are f32 (epi 0..3 and 6 only). */
int grip_debug_gemm(int epi, const void* A, const void* W, int M, int N, int K, const float* bias, const void* resid,
                    const void* aux, void* out, void* out2, float scalar, int m_pad, int variant, void* stream);
/* The LayerNorm-carrying epilogues (csrc/gemm.hip).  epi 3 with stat_part != NULL: the residual epilogue also writes, per row
 * and 64-column tile, (sum, sum of squares) of the stored values to stat_part [M, N/64, 2].  epi 7 / 8: LayerNorm folded into
 * the GEMM: out = [quickgelu](rstd_r (A W'^T - mean_r colsum) + bias) with rowstat [M, 2] = (mean, rstd), A = the raw rows. */
int grip_debug_gemm_ln(int epi, const void* A, const void* W, int M, int N, int K, const float* bias, const void* resid, void* out, void* out2,
                       float* stat_part, const float* rowstat, const float* colsum, int m_pad, int variant, void* stream);
/* The prompt-step forms of the text tower's train-mode forward (r04).  epi 7 / 8 with stat_in != NULL: the LayerNorm-folded GEMM reads the producer's
 * stat_parts partial (sum, sum of squares) pairs ([stat_parts, M, 2]) instead of finalised statistics -- inside the kernel on the loader-wave kernels,
 * through ln_stats_finalize into `rowstat` (writable, [M, 2]) otherwise.  epi 3 with ksplit > 1: cooperative split-K of the residual GEMM (partial tiles
 * in coop_scratch, >= tiles_of_64x128 * ksplit * 32 KiB; tickets in coop_counter, 4 ints per tile, zero on entry and zero again on return).
 * grip_debug_coop_split: the factor the tower would choose for this shape (1 = form not used). */
int grip_debug_gemm_train(int epi, const void* A, const void* W, int M, int N, int K, const float* bias, const void* resid, void* out, void* out2,
                          float* stat_part, float* rowstat, const float* colsum, const float* stat_in, int stat_parts, int ksplit,
                          float* coop_scratch, int* coop_counter, int m_pad, void* stream);
int grip_debug_coop_split(int M, int N, int K);
/* Split-K product (the input-gradient GEMMs of the prompt steps): out = ksplit partial [M, N] f32 buffers, split_stride floats apart,
 * partial p = A[:, Kp] W[:, Kp]^T over the p-th K range.  ksplit = 0: the launcher's own choice; *ksplit_used receives the factor. */
int grip_debug_gemm_splitk(const void* A, const void* W, int M, int N, int K, float* out, int ksplit, int64_t split_stride, int* ksplit_used,
                           int m_pad, int variant, void* stream);
/* Wg = f16(gamma o W) [N, K], colsum[n] = sum_k Wg[n][k], bias_out = bias + W beta; then, if stat_part != NULL,
 * rowstat [M, 2] = (mean, rstd) from the [M, parts, 2] partial sums over rows of width d. */
int grip_debug_ln_fold(const void* W, const float* gamma, const float* beta, const float* bias, void* Wg, float* colsum, float* bias_out,
                       int N, int K, const float* stat_part, int parts, float* rowstat, int M, int d, void* stream);
/* The split-f16 GEMM of precision-2 towers (csrc/gemm_split.hip) on f32 inputs: A [m_pad, K] and W [N, K] are rewritten in the split layout
 * ([32 x hi | 32 x lo'] f16 per 32 consecutive k) into the scratch buffers a_split / w_split (4 bytes per element), then out = epi(A W^T):
 * epi 0 f32, 1 +bias -> f32, 3 +bias +resid(f32) -> f32, 2 +bias, QuickGELU -> the split layout (4 bytes per element).  m_pad: rows of A
 * allocated, a multiple of 256.  grip_debug_split_rows: f32 rows -> the split layout. */
int grip_debug_gemm_split(int epi, const float* A, const float* W, int M, int N, int K, const float* bias, const float* resid, void* out,
                          void* a_split, void* w_split, int m_pad, void* stream);
int grip_debug_split_rows(const float* x, void* out, int64_t rows, int K, void* stream);
/* 1 when the last split-f16 GEMM launch formed the a_hi w_lo product, 0 when it ran the two-pass kernel (every element of W an f16 number:
 * grip_debug_gemm_split checks W itself, a tower at grip_tower_finalize), -1 before any launch. */
int grip_debug_split_last_wlo(void);
/* Attention of a precision-2 tower: qkv [B*S, 3*H*64] f32 -> out [B*S, H*64] in the split layout (4 bytes per element).  mfma != 0: the matrix-pipe
 * kernel (csrc/attention_split.hip, S <= 320), else the f32 vector-ALU kernel writing the split layout (any S). */
int grip_debug_attention_split(const void* qkv, void* out, int B, int S, int H, int causal, int mfma, void* stream);
/* out[B*S, H*64] = softmax(q k^T / 8 [+ causal mask]) v for qkv[B*S, 3*H*64] (f16). */
int grip_debug_attention(const void* qkv, void* out, int B, int S, int H, int causal, void* stream);
/* The same in f32 (exact mode, csrc/attention_f32.hip): qkv and out f32, any S. */
int grip_debug_attention_exact(const void* qkv, void* out, int B, int S, int H, int causal, void* stream);
/* dqkv from qkv, the saved forward output o and d_out (S <= 288). */
int grip_debug_attention_bwd(const void* qkv, const void* o, const void* d_out, void* dqkv, int B, int S, int H, int causal, void* stream);
/* out[M,d] (f16) = LayerNorm(x[M,d] f32; gamma, beta), eps 1e-5. */
int grip_debug_layernorm(const float* x, const float* gamma, const float* beta, void* out, int M, int d, void* stream);

/* GEMM launch profiler: while enabled, every 4th GEMM launch is bracketed by HIP events on its stream. */
int grip_profile_enable(int on);
/* Per slot (variant * 8 + epilogue id) in [0, n): launches sampled, their total milliseconds and total 2*M*N*K. */
int grip_profile_collect(int n, int64_t* launches, double* total_ms, double* total_flops);

#ifdef __cplusplus
}
#endif
#endif
