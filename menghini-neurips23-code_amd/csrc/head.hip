// Cosine x logit-scale head with fused softmax / arg-max (all f32): the block the reference inlines
// 45 times (e.g. methods/semi_supervised_learning/textual_prompt.py:98-109) and the tail of
// clip_model(image, text) + softmax + argmax in utils/clip_pseudolabels.py:35-41.
// HBM-bound: n*e*4 bytes in, n*c*8 bytes out; the c*e text matrix is L2-resident.
// One wave per image row: the row is normalised and pre-multiplied by the scale in registers
// (the reference computes (scale * img_n) @ txt_n.T), then each class is one 64-lane dot product.
#include <math.h>

#include "common.h"

__device__ __forceinline__ float wsum(float v) { return wave_sum(v); }

// out[r] = in[r] / ||in[r]||
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int e) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = in + (size_t)row * e;
    float q = 0.f;
    for (int i = lane; i < e; i += 64) q += x[i] * x[i];
    const float nrm = sqrtf(wsum(q));
    for (int i = lane; i < e; i += 64) out[(size_t)row * e + i] = x[i] / nrm;
}

#define HEAD_MAX_EV 8  // e <= 64 * 4 * 8 = 2048
#define HEAD_MAX_CV 16 // c <= 1024

// One workgroup (4 waves) per image row: the row is normalised and pre-multiplied by the scale in every wave's registers, wave w
// takes the classes j = w (mod 4) in groups of four (independent row loads and wave reductions in flight), the logits meet in
// LDS, and wave 0 does the softmax / arg-max tail.  (Round 1 gave a whole row to ONE wave: 102 dependent class trips = 85 us
// for a 16-row training batch, on the critical path between the text tower's forward and backward.)  NW = waves per row: 4 for pools (many rows fill
// the chip), 16 for training batches (a few rows: 102 classes are then 2 trips of 4 per wave instead of 7 -- 25.9 us -> r04).  A class's logit is
// computed by one wave in the same order whatever NW is: the two forms return the same bits.
template <int NW>
__global__ __launch_bounds__(NW * 64) void cosine_head_kernel(const float* __restrict__ img, const float* __restrict__ txtn, float scale,
                                                          int n, int c, int e, float* __restrict__ logits, float* __restrict__ probs,
                                                          int32_t* __restrict__ am_logits, int32_t* __restrict__ am_probs) {
    __shared__ float lgs[64 * HEAD_MAX_CV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const int e4 = e >> 2;
    const f32x4* x = (const f32x4*)(img + (size_t)row * e);
    f32x4 v[HEAD_MAX_EV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAX_EV; ++i)
        if (lane + 64 * i < e4) {
            v[i] = x[lane + 64 * i];
            q += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
    const float nrm = sqrtf(wsum(q));
#pragma unroll
    for (int i = 0; i < HEAD_MAX_EV; ++i)
        if (lane + 64 * i < e4) v[i] = scale * (v[i] / nrm);

    for (int j0 = wave * 4; j0 < c; j0 += NW * 4) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u < c ? j0 + u : c - 1;
            const f32x4* t = (const f32x4*)(txtn + (size_t)j * e);
#pragma unroll
            for (int i = 0; i < HEAD_MAX_EV; ++i)
                if (lane + 64 * i < e4) {
                    const f32x4 w = t[lane + 64 * i];
                    d[u] += v[i][0] * w[0] + v[i][1] * w[1] + v[i][2] * w[2] + v[i][3] * w[3];
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = wsum(d[u]);
        if (lane < 4 && j0 + lane < c) lgs[j0 + lane] = lane == 0 ? d[0] : (lane == 1 ? d[1] : (lane == 2 ? d[2] : d[3]));
    }
    __syncthreads();
    if (wave != 0) return;
    float lg[HEAD_MAX_CV];  // lane l keeps classes l, l+64, ...
    // row max + first arg-max over logits
    float m = -INFINITY;
    int am = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < HEAD_MAX_CV; ++s) {
        const int j = s * 64 + lane;
        if (j < c) {
            lg[s] = lgs[j];
            logits[(size_t)row * c + j] = lg[s];
            if (lg[s] > m) { m = lg[s]; am = j; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o);
        const int oa = __shfl_xor(am, o);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    if (lane == 0 && am_logits) am_logits[row] = am;
    if (!probs && !am_probs) return;
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < HEAD_MAX_CV; ++s)
        if (s * 64 + lane < c) { lg[s] = expf(lg[s] - m); sum += lg[s]; }
    sum = wsum(sum);
    float pm = -1.f;
    int pa = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < HEAD_MAX_CV; ++s) {
        const int j = s * 64 + lane;
        if (j < c) {
            const float p = lg[s] / sum;
            if (probs) probs[(size_t)row * c + j] = p;
            if (p > pm) { pm = p; pa = j; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(pm, o);
        const int oa = __shfl_xor(pa, o);
        if (om > pm || (om == pm && oa < pa)) { pm = om; pa = oa; }
    }
    if (lane == 0 && am_probs) am_probs[row] = pa;
}

extern "C" int grip_cosine_head(const float* img_emb, const float* txt_emb, float scale, int n, int c, int e,
                                float* logits, float* probs, int32_t* argmax_logits, int32_t* argmax_probs,
                                float* txt_norm_scratch, void* stream) {
    GRIP_REQUIRE(img_emb && txt_emb && logits && txt_norm_scratch, "cosine_head: null pointer");
    GRIP_REQUIRE(n > 0 && c > 0 && c <= 64 * HEAD_MAX_CV && e % 4 == 0 && e <= 256 * HEAD_MAX_EV, "cosine_head: unsupported shape n=%d c=%d e=%d", n, c, e);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((c + 3) / 4), dim3(256), 0, s, txt_emb, txt_norm_scratch, c, e);
    if (n <= 64 && e <= 1024)
        hipLaunchKernelGGL(cosine_head_kernel<16>, dim3(n), dim3(1024), 0, s, img_emb, txt_norm_scratch, scale, n, c, e, logits, probs, argmax_logits, argmax_probs);
    else
        hipLaunchKernelGGL(cosine_head_kernel<4>, dim3(n), dim3(256), 0, s, img_emb, txt_norm_scratch, scale, n, c, e, logits, probs, argmax_logits, argmax_probs);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of logits = scale * ihat @ that^T (ihat, that = L2-normalised rows).
//   d ihat = scale * dL @ that;  d img = (d ihat - ihat * <ihat, d ihat>) / ||img||      (same for txt with dL^T)
// One workgroup (4 waves) per output row; wave w takes the `other` rows j = w (mod 4), two at a time (independent loads and wave
// reductions in flight), normalising them on the fly (n and c are small in training); the four partial rows meet in LDS and wave 0
// adds them in wave order and projects.  (Until r03 one wave walked all of a row's `other` rows: 45 dependent trips = 41 us of a VPT
// step for a 16 x 45 head.)
__global__ __launch_bounds__(256) void cosine_head_bwd_kernel(const float* __restrict__ self, const float* __restrict__ other, float scale,
                                                              int n_self, int n_other, int e, const float* __restrict__ dl, int transposed,
                                                              int ld_dl, float* __restrict__ grad) {
    __shared__ f32x4 part[3][64 * HEAD_MAX_EV];      // partial rows of waves 1..3 (wave 0 keeps its own in registers)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const int e4 = e >> 2;
    f32x4 acc[HEAD_MAX_EV];
#pragma unroll
    for (int i = 0; i < HEAD_MAX_EV; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto dlv = [&](int j) { return transposed ? dl[(size_t)j * ld_dl + row] : dl[(size_t)row * ld_dl + j]; };
    for (int j0 = wave; j0 < n_other; j0 += 8) {
        const int j1 = j0 + 4;
        const bool two = j1 < n_other;
        const f32x4* o0 = (const f32x4*)(other + (size_t)j0 * e);
        const f32x4* o1 = (const f32x4*)(other + (size_t)(two ? j1 : j0) * e);
        f32x4 v0[HEAD_MAX_EV], v1[HEAD_MAX_EV];
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int i = 0; i < HEAD_MAX_EV; ++i)
            if (lane + 64 * i < e4) {
                v0[i] = o0[lane + 64 * i];
                v1[i] = o1[lane + 64 * i];
                q0 += v0[i][0] * v0[i][0] + v0[i][1] * v0[i][1] + v0[i][2] * v0[i][2] + v0[i][3] * v0[i][3];
                q1 += v1[i][0] * v1[i][0] + v1[i][1] * v1[i][1] + v1[i][2] * v1[i][2] + v1[i][3] * v1[i][3];
            }
        const float g0 = scale * dlv(j0) / sqrtf(wsum(q0));
        const float g1 = two ? scale * dlv(j1) / sqrtf(wsum(q1)) : 0.f;
#pragma unroll
        for (int i = 0; i < HEAD_MAX_EV; ++i)
            if (lane + 64 * i < e4) {
                acc[i] += v0[i] * g0;
                acc[i] += v1[i] * g1;
            }
    }
    if (wave != 0) {
#pragma unroll
        for (int i = 0; i < HEAD_MAX_EV; ++i)
            if (lane + 64 * i < e4) part[wave - 1][lane + 64 * i] = acc[i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < HEAD_MAX_EV; ++i)
        if (lane + 64 * i < e4) acc[i] = ((acc[i] + part[0][lane + 64 * i]) + part[1][lane + 64 * i]) + part[2][lane + 64 * i];
    const f32x4* x = (const f32x4*)(self + (size_t)row * e);
    f32x4 xs[HEAD_MAX_EV];
    float q = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAX_EV; ++i)
        if (lane + 64 * i < e4) {
            xs[i] = x[lane + 64 * i];
            q += xs[i][0] * xs[i][0] + xs[i][1] * xs[i][1] + xs[i][2] * xs[i][2] + xs[i][3] * xs[i][3];
            dot += xs[i][0] * acc[i][0] + xs[i][1] * acc[i][1] + xs[i][2] * acc[i][2] + xs[i][3] * acc[i][3];
        }
    q = wsum(q);
    dot = wsum(dot);
    const float inv = 1.0f / sqrtf(q);
    const float proj = dot / q;   // <xhat, d xhat> / ||x||  =  <x, d xhat> / ||x||^2
    f32x4* o = (f32x4*)(grad + (size_t)row * e);
#pragma unroll
    for (int i = 0; i < HEAD_MAX_EV; ++i)
        if (lane + 64 * i < e4) o[lane + 64 * i] = (acc[i] - xs[i] * proj) * inv;
}

extern "C" int grip_cosine_head_backward(const float* img_emb, const float* txt_emb, float scale, int n, int c, int e,
                                         const float* grad_logits, float* grad_img, float* grad_txt, void* stream) {
    GRIP_REQUIRE(img_emb && txt_emb && grad_logits, "cosine_head_backward: null pointer");
    GRIP_REQUIRE(n > 0 && c > 0 && e % 4 == 0 && e <= 256 * HEAD_MAX_EV, "cosine_head_backward: unsupported shape");
    hipStream_t s = (hipStream_t)stream;
    if (grad_img)
        hipLaunchKernelGGL(cosine_head_bwd_kernel, dim3(n), dim3(256), 0, s, img_emb, txt_emb, scale, n, c, e, grad_logits, 0, c, grad_img);
    if (grad_txt)
        hipLaunchKernelGGL(cosine_head_bwd_kernel, dim3(c), dim3(256), 0, s, txt_emb, img_emb, scale, c, n, e, grad_logits, 1, c, grad_txt);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ------------------------------------------------------------------------------------------------
// loss = sum_i w_i * (logsumexp(logits_i) - logits_i[label_i]);  grad_i = w_i * (softmax(logits_i) - onehot(label_i)).
// One block, wave w walks rows w, w+4, ...; partial sums are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void weighted_ce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                          const float* __restrict__ weight, int n, int c, float* __restrict__ loss,
                                                          float* __restrict__ grad) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float total = 0.f;
    for (int row = wave; row < n; row += 4) {
        const float* x = logits + (size_t)row * c;
        const float w = weight[row];
        const int lab = labels[row];
        float m = -INFINITY;
        for (int j = lane; j < c; j += 64) m = fmaxf(m, x[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int j = lane; j < c; j += 64) sum += expf(x[j] - m);
        sum = wsum(sum);
        const float lse = m + logf(sum);
        if (grad)
            for (int j = lane; j < c; j += 64) grad[(size_t)row * c + j] = w * (expf(x[j] - m) / sum - (j == lab ? 1.f : 0.f));
        if (w != 0.f && lab >= 0 && lab < c) total += w * (lse - x[lab]);
    }
    if (lane == 0) part[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((part[0] + part[1]) + part[2]) + part[3];
}

extern "C" int grip_weighted_ce(const float* logits, const int32_t* labels, const float* row_weight, int n, int c,
                                float* loss, float* grad_logits, void* stream) {
    GRIP_REQUIRE(logits && labels && row_weight && loss, "weighted_ce: null pointer");
    GRIP_REQUIRE(n > 0 && c > 0, "weighted_ce: empty input");
    hipLaunchKernelGGL(weighted_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, row_weight, n, c, loss, grad_logits);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
