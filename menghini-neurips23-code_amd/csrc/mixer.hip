// UPT prompt mixer (reference models/prompts_models.py:99-146): the only TRAINABLE arithmetic on the hot path and the only weight
// gradients.  proj_coop_pre / proj_vpt_pre (Linear -> D = 128), one residual attention block of width D with ONE head over the
// sequence [2, P, D] (sequence length 2 = (text prompt n, visual prompt n), batch P), the fp32 -> fp16 -> dtype round trip of
// :138-145, proj_coop_post / proj_vpt_post.  < 10 MFLOP and 2 MB of weights per direction: nothing here is bound by a roof, the
// cost is launch count and serial depth.  So every Linear is ONE small launch of one kernel, spread over the chip by output
// column (one wave per column, the R = 2P <= 32 rows held in LDS, lanes striding over K), and everything between two Linears
// -- LayerNorm, the 2-token attention, QuickGELU, the residual adds, and in the backward their derivatives -- runs as that
// kernel's PROLOGUE on the LDS tile (recomputed by every workgroup: it is R x D numbers) or as its per-element epilogue.
// Forward = 6 launches; backward = 1 transpose of the weights + 6 input-gradient launches (the same kernel on W^T) + 1 launch that
// forms all weight / bias / LayerNorm-affine gradients.  No atomics: results are bit-reproducible.
#include "common.h"

namespace {
enum Pro { PRO_NONE = 0, PRO_LN = 1, PRO_ATTN = 2, PRO_LNBWD_ADD = 3, PRO_ATTNBWD = 4 };
enum Epi { MEPI_NONE = 0, MEPI_GELU = 1, MEPI_RESID = 2, MEPI_RESID_F16 = 3, MEPI_F16ROUND = 4, MEPI_GELUGRAD = 5 };

struct MixDesc {          // Y[R, N] = epi(pro(X)[R, K] * W[N, K]^T + bias)
    const float* X; int ldx;
    const float* W; int ldw;
    const float* bias;
    float* Y; int ldy;
    int R, N, K;
    int P;                // prompts per modality (R = 2P for the block's rows)
    const float *p0, *p1, *p2, *p3;     // prologue operands
    float *save0, *save1;               // prologue results written once (workgroup 0)
    const float* e0; float* e1;         // epilogue operands
};
struct MixLaunch { MixDesc d[2]; int n; int pro, epi; int round_pro; };      // round_pro: PRO_LNBWD_ADD rounds its result to fp16 (see half_linears)

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = 1.f / (1.f + __expf(-1.702f * x));
    return s * (1.f + 1.702f * x * (1.f - s));
}

// One workgroup = 4 waves = 4 output columns of problem blockIdx.y.  Dynamic LDS: R * K floats.
__global__ __launch_bounds__(256) void mixer_linear_kernel(MixLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];
    const MixDesc& D = L.d[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = D.R, K = D.K, N = D.N, P = D.P;
    if ((int)blockIdx.x * 4 >= N) return;
    const bool writer = blockIdx.x == 0;
    // ---------------------------------------------------------------- prologue: the R x K operand rows in LDS
    if (L.pro == PRO_NONE) {
        for (int i = threadIdx.x; i < R * K; i += 256) Xs[i] = D.X[(size_t)(i / K) * D.ldx + (i % K)];
    } else if (L.pro == PRO_LN) {                  // y = LayerNorm(x) (gamma = p0, beta = p1); save0 <- y, save1 <- (mean, rstd)
        for (int r = wave; r < R; r += 4) {
            const float* x = D.X + (size_t)r * D.ldx;
            float s = 0.f;
            for (int k = lane; k < K; k += 64) s += x[k];
            const float mean = wave_sum(s) / (float)K;
            float q = 0.f;
            for (int k = lane; k < K; k += 64) { const float c = x[k] - mean; q += c * c; }
            const float rstd = rsqrtf(wave_sum(q) / (float)K + LN_EPS);
            for (int k = lane; k < K; k += 64) {
                const float y = (x[k] - mean) * rstd * D.p0[k] + D.p1[k];
                Xs[r * K + k] = y;
                if (writer && D.save0) D.save0[r * K + k] = y;
            }
            if (writer && lane == 0 && D.save1) { D.save1[2 * r] = mean; D.save1[2 * r + 1] = rstd; }
        }
    } else if (L.pro == PRO_ATTN) {                // X = qkv [2P, 3K]; o[l, n] = sum_m softmax_m(q[l,n].k[m,n] / sqrt(K)) v[m, n]; save0 <- o
        const float scale = rsqrtf((float)K);
        for (int n = wave; n < P; n += 4) {
            const float* q0 = D.X + (size_t)n * D.ldx;
            const float* q1 = D.X + (size_t)(P + n) * D.ldx;
            float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
            for (int k = lane; k < K; k += 64) {
                const float a0 = q0[k], a1 = q1[k], k0 = q0[K + k], k1 = q1[K + k];
                s00 += a0 * k0; s01 += a0 * k1; s10 += a1 * k0; s11 += a1 * k1;
            }
            s00 = wave_sum(s00) * scale; s01 = wave_sum(s01) * scale; s10 = wave_sum(s10) * scale; s11 = wave_sum(s11) * scale;
            const float m0 = fmaxf(s00, s01), m1 = fmaxf(s10, s11);
            const float e00 = __expf(s00 - m0), e01 = __expf(s01 - m0), e10 = __expf(s10 - m1), e11 = __expf(s11 - m1);
            const float p00 = e00 / (e00 + e01), p01 = e01 / (e00 + e01), p10 = e10 / (e10 + e11), p11 = e11 / (e10 + e11);
            for (int k = lane; k < K; k += 64) {
                const float v0 = q0[2 * K + k], v1 = q1[2 * K + k];
                const float o0 = p00 * v0 + p01 * v1, o1 = p10 * v0 + p11 * v1;
                Xs[n * K + k] = o0; Xs[(P + n) * K + k] = o1;
                if (writer && D.save0) { D.save0[n * K + k] = o0; D.save0[(P + n) * K + k] = o1; }
            }
        }
    } else if (L.pro == PRO_LNBWD_ADD) {           // X = dL/d(LN output); p0 = LN input x, p1 = gamma, p2 = (mean, rstd), p3 = gradient arriving on the residual path
        for (int r = wave; r < R; r += 4) {
            const float mean = D.p2[2 * r], rstd = D.p2[2 * r + 1];
            const float* dz = D.X + (size_t)r * D.ldx;
            const float* x = D.p0 + (size_t)r * K;
            float c1 = 0.f, c2 = 0.f;
            for (int k = lane; k < K; k += 64) {
                const float g = dz[k] * D.p1[k], xh = (x[k] - mean) * rstd;
                c1 += g; c2 += g * xh;
            }
            c1 = wave_sum(c1) / (float)K; c2 = wave_sum(c2) / (float)K;
            for (int k = lane; k < K; k += 64) {
                const float g = dz[k] * D.p1[k], xh = (x[k] - mean) * rstd;
                float dx = D.p3[r * K + k] + rstd * (g - c1 - xh * c2);
                if (L.round_pro) dx = (float)(half_t)dx;       // the gradient crosses the fp16 -> fp32 cast in front of the block (autograd casts it back)
                Xs[r * K + k] = dx;
                if (writer && D.save0) D.save0[r * K + k] = dx;
            }
        }
    } else {                                       // PRO_ATTNBWD: X = d o [2P, Dh]; p0 = qkv [2P, 3 Dh]; rows of Xs: (dq | dk | dv), K = 3 Dh; save0 <- them
        const int Dh = K / 3;
        const float scale = rsqrtf((float)Dh);
        for (int n = wave; n < P; n += 4) {
            const float* q0 = D.p0 + (size_t)n * K;
            const float* q1 = D.p0 + (size_t)(P + n) * K;
            const float* g0 = D.X + (size_t)n * D.ldx;
            const float* g1 = D.X + (size_t)(P + n) * D.ldx;
            float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f, d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
            for (int k = lane; k < Dh; k += 64) {
                const float a0 = q0[k], a1 = q1[k], k0 = q0[Dh + k], k1 = q1[Dh + k], v0 = q0[2 * Dh + k], v1 = q1[2 * Dh + k];
                s00 += a0 * k0; s01 += a0 * k1; s10 += a1 * k0; s11 += a1 * k1;
                d00 += g0[k] * v0; d01 += g0[k] * v1; d10 += g1[k] * v0; d11 += g1[k] * v1;       // dP[l][m] = d o[l] . v[m]
            }
            s00 = wave_sum(s00) * scale; s01 = wave_sum(s01) * scale; s10 = wave_sum(s10) * scale; s11 = wave_sum(s11) * scale;
            d00 = wave_sum(d00); d01 = wave_sum(d01); d10 = wave_sum(d10); d11 = wave_sum(d11);
            const float m0 = fmaxf(s00, s01), m1 = fmaxf(s10, s11);
            const float e00 = __expf(s00 - m0), e01 = __expf(s01 - m0), e10 = __expf(s10 - m1), e11 = __expf(s11 - m1);
            const float p00 = e00 / (e00 + e01), p01 = e01 / (e00 + e01), p10 = e10 / (e10 + e11), p11 = e11 / (e10 + e11);
            const float t0 = p00 * d00 + p01 * d01, t1 = p10 * d10 + p11 * d11;
            const float ds00 = p00 * (d00 - t0) * scale, ds01 = p01 * (d01 - t0) * scale, ds10 = p10 * (d10 - t1) * scale, ds11 = p11 * (d11 - t1) * scale;
            for (int k = lane; k < Dh; k += 64) {
                const float a0 = q0[k], a1 = q1[k], k0 = q0[Dh + k], k1 = q1[Dh + k];
                float* r0 = Xs + (size_t)n * K;
                float* r1 = Xs + (size_t)(P + n) * K;
                r0[k] = ds00 * k0 + ds01 * k1;                  r1[k] = ds10 * k0 + ds11 * k1;                     // dq
                r0[Dh + k] = ds00 * a0 + ds10 * a1;             r1[Dh + k] = ds01 * a0 + ds11 * a1;                // dk
                r0[2 * Dh + k] = p00 * g0[k] + p10 * g1[k];     r1[2 * Dh + k] = p01 * g0[k] + p11 * g1[k];        // dv
            }
        }
        __syncthreads();
        if (writer && D.save0)
            for (int i = threadIdx.x; i < R * K; i += 256) D.save0[i] = Xs[i];
    }
    __syncthreads();
    // ---------------------------------------------------------------- one output column per wave
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const float* w = D.W + (size_t)n * D.ldw;
    const float b = D.bias ? D.bias[n] : 0.f;
    for (int r0 = 0; r0 < R; r0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = lane; k < K; k += 64) {
            const float wk = w[k];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (r0 + j < R) acc[j] += Xs[(r0 + j) * K + k] * wk;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (r0 + j >= R) break;
            const float y = wave_sum(acc[j]) + b;
            if (lane != 0) continue;
            const int r = r0 + j;
            const size_t o = (size_t)r * D.ldy + n;
            switch (L.epi) {
                case MEPI_NONE: D.Y[o] = y; break;
                case MEPI_GELU: D.e1[o] = y; D.Y[o] = quick_gelu(y); break;
                case MEPI_RESID: D.Y[o] = D.e0[o] + y; break;
                case MEPI_RESID_F16: { const float x2 = D.e0[o] + y; D.Y[o] = x2; D.e1[o] = (float)(half_t)x2; } break;
                case MEPI_F16ROUND: D.Y[o] = (float)(half_t)y; break;
                default: D.Y[o] = y * quick_gelu_grad(D.e0[o]); break;        // MEPI_GELUGRAD
            }
        }
    }
}

// ---- transposes of the eight weight matrices (they are trained: once per backward) ----------------------------------------
struct TrDesc { const float* W; float* WT; int N, K; };      // W [N, K] -> WT [K, N]
struct TrLaunch { TrDesc d[8]; };
__global__ __launch_bounds__(256) void mixer_transpose_kernel(TrLaunch L) {
    __shared__ float tile[32][33];
    const TrDesc& D = L.d[blockIdx.z];
    const int tn = blockIdx.y * 32, tk = blockIdx.x * 32;
    if (tn >= D.N || tk >= D.K) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (tn + j < D.N && tk + tx < D.K) tile[j][tx] = D.W[(size_t)(tn + j) * D.K + tk + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (tk + j < D.K && tn + tx < D.N) D.WT[(size_t)(tk + j) * D.N + tn + tx] = tile[tx][j];
}

// ---- all parameter gradients in one launch ---------------------------------------------------------------------------------
// kind 0: dW[n, k] = sum_r dY[r, n] X[r, k], db[n] = sum_r dY[r, n];  kind 1 (LayerNorm affine): dgamma[k] = sum_r dY[r, k] xhat[r, k],
// dbeta[k] = sum_r dY[r, k] with xhat from the LN input X and its saved (mean, rstd).
struct GradDesc { int kind; const float* dY; int ldy; const float* X; int ldx; const float* stats; float* dW; float* db; int R, N, K; };
struct GradLaunch { GradDesc d[10]; };
__global__ __launch_bounds__(256) void mixer_param_grad_kernel(GradLaunch L) {
    const GradDesc& D = L.d[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (D.kind == 1) {
        if (i >= D.K) return;
        float g = 0.f, bb = 0.f;
        for (int r = 0; r < D.R; ++r) {
            const float dy = D.dY[(size_t)r * D.ldy + i];
            g += dy * (D.X[(size_t)r * D.ldx + i] - D.stats[2 * r]) * D.stats[2 * r + 1];
            bb += dy;
        }
        D.dW[i] = g; D.db[i] = bb;
        return;
    }
    if (i >= (int64_t)D.N * D.K) return;
    const int n = (int)(i / D.K), k = (int)(i % D.K);
    float s = 0.f;
    for (int r = 0; r < D.R; ++r) s += D.dY[(size_t)r * D.ldy + n] * D.X[(size_t)r * D.ldx + k];
    D.dW[i] = s;
    if (k == 0) {
        float bb = 0.f;
        for (int r = 0; r < D.R; ++r) bb += D.dY[(size_t)r * D.ldy + n];
        D.db[n] = bb;
    }
}

// Workspace carve-up (floats).  Saved by the forward for the backward: x0 .. out16; the rest is backward scratch.
struct MixWs {
    float *x0, *st1, *y1, *qkv, *o, *x1, *st2, *z, *h, *g, *x2, *out16;
    float *d_x2, *dh, *dz, *d_x1, *d_o, *dqkv, *dy, *d_x0;
    float *T_cq, *T_vq, *T_pr, *T_fc, *T_o, *T_in, *T_cp, *T_vp;
    size_t floats;
};
MixWs carve_mixer(float* base, int P, int dt, int dv, int D) {
    MixWs w{};
    const size_t R = 2 * (size_t)P;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    w.x0 = take(R * D); w.st1 = take(R * 2); w.y1 = take(R * D); w.qkv = take(R * 3 * D); w.o = take(R * D); w.x1 = take(R * D);
    w.st2 = take(R * 2); w.z = take(R * D); w.h = take(R * 4 * D); w.g = take(R * 4 * D); w.x2 = take(R * D); w.out16 = take(R * D);
    w.d_x2 = take(R * D); w.dh = take(R * 4 * D); w.dz = take(R * D); w.d_x1 = take(R * D); w.d_o = take(R * D); w.dqkv = take(R * 3 * D);
    w.dy = take(R * D); w.d_x0 = take(R * D);
    w.T_cq = take((size_t)dt * D); w.T_vq = take((size_t)dv * D); w.T_pr = take((size_t)4 * D * D); w.T_fc = take((size_t)4 * D * D);
    w.T_o = take((size_t)D * D); w.T_in = take((size_t)3 * D * D); w.T_cp = take((size_t)dt * D); w.T_vp = take((size_t)dv * D);
    w.floats = off;
    return w;
}

int check_shape(const grip_upt_mixer* m) {
    GRIP_REQUIRE(m, "upt_mixer: null pointer");
    GRIP_REQUIRE(m->n_prompt >= 1 && m->n_prompt <= 16, "upt_mixer: n_prompt = %d (1 .. 16 prompt tokens per modality)", m->n_prompt);
    GRIP_REQUIRE(m->dim >= 64 && m->dim <= 256 && m->dim % 64 == 0, "upt_mixer: dim = %d (64, 128, 192 or 256)", m->dim);
    GRIP_REQUIRE(m->text_width >= 64 && m->text_width <= 1280 && m->vision_width >= 64 && m->vision_width <= 1280, "upt_mixer: widths %d / %d out of range",
                 m->text_width, m->vision_width);
    return GRIP_OK;
}

int launch_linear(const MixLaunch& L, hipStream_t s) {
    int max_n = 0;
    size_t lds = 0;
    for (int i = 0; i < L.n; ++i) {
        max_n = L.d[i].N > max_n ? L.d[i].N : max_n;
        const size_t need = (size_t)L.d[i].R * L.d[i].K * sizeof(float);
        lds = need > lds ? need : lds;
    }
    static size_t configured = 0;
    if (lds > configured) {
        GRIP_CHECK_HIP(hipFuncSetAttribute((const void*)mixer_linear_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    hipLaunchKernelGGL(mixer_linear_kernel, dim3((max_n + 3) / 4, L.n), dim3(256), lds, s, L);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
}  // namespace

#define RUNM(x) do { int _rc = (x); if (_rc != GRIP_OK) return _rc; } while (0)

extern "C" int grip_upt_mixer_workspace(int n_prompt, int text_width, int vision_width, int dim, size_t* bytes) {
    GRIP_REQUIRE(bytes, "upt_mixer_workspace: null pointer");
    grip_upt_mixer m{};
    m.n_prompt = n_prompt; m.text_width = text_width; m.vision_width = vision_width; m.dim = dim;
    RUNM(check_shape(&m));
    *bytes = carve_mixer(nullptr, n_prompt, text_width, vision_width, dim).floats * sizeof(float) + 256;
    return GRIP_OK;
}

extern "C" int grip_upt_mixer_forward(const grip_upt_mixer* m, float* coop_out, float* vpt_out, void* workspace, size_t workspace_bytes, void* stream) {
    try {
        RUNM(check_shape(m));
        GRIP_REQUIRE(coop_out && vpt_out && workspace && m->coop && m->vpt && m->coop_pre_w && m->vpt_pre_w && m->in_w && m->out_w && m->fc_w && m->proj_w &&
                     m->coop_post_w && m->vpt_post_w && m->ln1_g && m->ln1_b && m->ln2_g && m->ln2_b, "upt_mixer_forward: null pointer");
        const int P = m->n_prompt, dt = m->text_width, dv = m->vision_width, D = m->dim, R = 2 * P;
        float* base = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
        const MixWs w = carve_mixer(base, P, dt, dv, D);
        GRIP_REQUIRE((char*)base + w.floats * sizeof(float) <= (char*)workspace + workspace_bytes, "upt_mixer_forward: workspace too small");
        hipStream_t s = (hipStream_t)stream;
        MixLaunch L{};
        // x0 = [proj_coop_pre(coop); proj_vpt_pre(vpt)]   (prompts_models.py:130-135; the cat along dim 0 makes the sequence axis)
        const int hl = m->half_linears != 0;       // fp16 Linears (multimodal_prompt.py:46): their outputs are fp16 tensors
        L.n = 2; L.pro = PRO_NONE; L.epi = hl ? MEPI_F16ROUND : MEPI_NONE;
        L.d[0] = MixDesc{m->coop, dt, m->coop_pre_w, dt, m->coop_pre_b, w.x0, D, P, D, dt, P};
        L.d[1] = MixDesc{m->vpt, dv, m->vpt_pre_w, dv, m->vpt_pre_b, w.x0 + (size_t)P * D, D, P, D, dv, P};
        RUNM(launch_linear(L, s));
        // qkv = in_proj(ln_1(x0))
        L = MixLaunch{}; L.n = 1; L.pro = PRO_LN; L.epi = MEPI_NONE;
        L.d[0] = MixDesc{w.x0, D, m->in_w, D, m->in_b, w.qkv, 3 * D, R, 3 * D, D, P, m->ln1_g, m->ln1_b, nullptr, nullptr, w.y1, w.st1};
        RUNM(launch_linear(L, s));
        // x1 = x0 + out_proj(attention over the 2-token sequences)
        L = MixLaunch{}; L.n = 1; L.pro = PRO_ATTN; L.epi = MEPI_RESID;
        L.d[0] = MixDesc{w.qkv, 3 * D, m->out_w, D, m->out_b, w.x1, D, R, D, D, P, nullptr, nullptr, nullptr, nullptr, w.o, nullptr, w.x0, nullptr};
        RUNM(launch_linear(L, s));
        // h = c_fc(ln_2(x1)); g = QuickGELU(h)
        L = MixLaunch{}; L.n = 1; L.pro = PRO_LN; L.epi = MEPI_GELU;
        L.d[0] = MixDesc{w.x1, D, m->fc_w, D, m->fc_b, w.g, 4 * D, R, 4 * D, D, P, m->ln2_g, m->ln2_b, nullptr, nullptr, w.z, w.st2, nullptr, w.h};
        RUNM(launch_linear(L, s));
        // x2 = x1 + c_proj(g); out16 = x2 rounded to fp16 (:138-145)
        L = MixLaunch{}; L.n = 1; L.pro = PRO_NONE; L.epi = MEPI_RESID_F16;
        L.d[0] = MixDesc{w.g, 4 * D, m->proj_w, 4 * D, m->proj_b, w.x2, D, R, D, 4 * D, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, w.x1, w.out16};
        RUNM(launch_linear(L, s));
        // coop_embs = proj_coop_post(out16[0]); vpt_embs = proj_vpt_post(out16[1])
        L = MixLaunch{}; L.n = 2; L.pro = PRO_NONE; L.epi = hl ? MEPI_F16ROUND : MEPI_NONE;
        L.d[0] = MixDesc{w.out16, D, m->coop_post_w, D, m->coop_post_b, coop_out, dt, P, dt, D, P};
        L.d[1] = MixDesc{w.out16 + (size_t)P * D, D, m->vpt_post_w, D, m->vpt_post_b, vpt_out, dv, P, dv, D, P};
        RUNM(launch_linear(L, s));
        return GRIP_OK;
    } catch (...) { grip_set_error("upt_mixer_forward: exception"); return GRIP_ERR_ARG; }
}

extern "C" int grip_upt_mixer_backward(const grip_upt_mixer* m, const float* d_coop_out, const float* d_vpt_out, const grip_upt_mixer* grads,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    try {
        RUNM(check_shape(m));
        GRIP_REQUIRE(d_coop_out && d_vpt_out && grads && workspace, "upt_mixer_backward: null pointer");
        const grip_upt_mixer* g = grads;
        GRIP_REQUIRE(g->coop && g->vpt && g->coop_pre_w && g->coop_pre_b && g->vpt_pre_w && g->vpt_pre_b && g->ln1_g && g->ln1_b && g->in_w && g->in_b && g->out_w &&
                     g->out_b && g->ln2_g && g->ln2_b && g->fc_w && g->fc_b && g->proj_w && g->proj_b && g->coop_post_w && g->coop_post_b && g->vpt_post_w &&
                     g->vpt_post_b, "upt_mixer_backward: every gradient buffer must be given");
        const int P = m->n_prompt, dt = m->text_width, dv = m->vision_width, D = m->dim, R = 2 * P;
        float* base = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
        const MixWs w = carve_mixer(base, P, dt, dv, D);
        GRIP_REQUIRE((char*)base + w.floats * sizeof(float) <= (char*)workspace + workspace_bytes, "upt_mixer_backward: workspace too small");
        hipStream_t s = (hipStream_t)stream;
        // W^T of the eight matrices: the input-gradient products dX = dY W run through the forward kernel as dX = dY (W^T)^T
        TrLaunch T{};
        T.d[0] = TrDesc{m->coop_post_w, w.T_cq, dt, D}; T.d[1] = TrDesc{m->vpt_post_w, w.T_vq, dv, D};
        T.d[2] = TrDesc{m->proj_w, w.T_pr, D, 4 * D};   T.d[3] = TrDesc{m->fc_w, w.T_fc, 4 * D, D};
        T.d[4] = TrDesc{m->out_w, w.T_o, D, D};         T.d[5] = TrDesc{m->in_w, w.T_in, 3 * D, D};
        T.d[6] = TrDesc{m->coop_pre_w, w.T_cp, D, dt};  T.d[7] = TrDesc{m->vpt_pre_w, w.T_vp, D, dv};
        int maxd = dt > dv ? dt : dv;
        maxd = maxd > 4 * D ? maxd : 4 * D;
        hipLaunchKernelGGL(mixer_transpose_kernel, dim3((maxd + 31) / 32, (maxd + 31) / 32, 8), dim3(256), 0, s, T);
        GRIP_CHECK_HIP(hipGetLastError());
        MixLaunch L{};
        // d out16 = [d_coop W_cq ; d_vpt W_vq], rounded to fp16: the gradient of an fp16 tensor is fp16 in the reference's autograd (:141)
        L.n = 2; L.pro = PRO_NONE; L.epi = MEPI_F16ROUND;
        L.d[0] = MixDesc{d_coop_out, dt, w.T_cq, dt, nullptr, w.d_x2, D, P, D, dt, P};
        L.d[1] = MixDesc{d_vpt_out, dv, w.T_vq, dv, nullptr, w.d_x2 + (size_t)P * D, D, P, D, dv, P};
        RUNM(launch_linear(L, s));
        // dh = (d_x2 W_proj) * QuickGELU'(h)
        L = MixLaunch{}; L.n = 1; L.pro = PRO_NONE; L.epi = MEPI_GELUGRAD;
        L.d[0] = MixDesc{w.d_x2, D, w.T_pr, D, nullptr, w.dh, 4 * D, R, 4 * D, D, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, w.h, nullptr};
        RUNM(launch_linear(L, s));
        // dz = dh W_fc
        L = MixLaunch{}; L.n = 1; L.pro = PRO_NONE; L.epi = MEPI_NONE;
        L.d[0] = MixDesc{w.dh, 4 * D, w.T_fc, 4 * D, nullptr, w.dz, D, R, D, 4 * D, P};
        RUNM(launch_linear(L, s));
        // d_x1 = d_x2 + LN2'(dz);  d o = d_x1 W_out
        L = MixLaunch{}; L.n = 1; L.pro = PRO_LNBWD_ADD; L.epi = MEPI_NONE;
        L.d[0] = MixDesc{w.dz, D, w.T_o, D, nullptr, w.d_o, D, R, D, D, P, w.x1, m->ln2_g, w.st2, w.d_x2, w.d_x1, nullptr};
        RUNM(launch_linear(L, s));
        // dqkv = attention'(d o);  dy = dqkv W_in
        L = MixLaunch{}; L.n = 1; L.pro = PRO_ATTNBWD; L.epi = MEPI_NONE;
        L.d[0] = MixDesc{w.d_o, D, w.T_in, 3 * D, nullptr, w.dy, D, R, D, 3 * D, P, w.qkv, nullptr, nullptr, nullptr, w.dqkv, nullptr};
        RUNM(launch_linear(L, s));
        // d_x0 = d_x1 + LN1'(dy);  d coop = d_x0[0] W_cp, d vpt = d_x0[1] W_vp   (both problems recompute d_x0's P rows of their half)
        // (fp16 Linears: d_x0 is rounded to fp16 where it crosses the cast in front of the block, and so are the prompt gradients the fp16 Linears return)
        L = MixLaunch{}; L.n = 2; L.pro = PRO_LNBWD_ADD; L.epi = m->half_linears ? MEPI_F16ROUND : MEPI_NONE; L.round_pro = m->half_linears != 0;
        L.d[0] = MixDesc{w.dy, D, w.T_cp, D, nullptr, g->coop, dt, P, dt, D, P, w.x0, m->ln1_g, w.st1, w.d_x1, w.d_x0, nullptr};
        L.d[1] = MixDesc{w.dy + (size_t)P * D, D, w.T_vp, D, nullptr, g->vpt, dv, P, dv, D, P, w.x0 + (size_t)P * D, m->ln1_g, w.st1 + 2 * P, w.d_x1 + (size_t)P * D,
                         w.d_x0 + (size_t)P * D, nullptr};
        RUNM(launch_linear(L, s));
        // every parameter gradient
        GradLaunch G{};
        G.d[0] = GradDesc{0, d_coop_out, dt, w.out16, D, nullptr, g->coop_post_w, g->coop_post_b, P, dt, D};
        G.d[1] = GradDesc{0, d_vpt_out, dv, w.out16 + (size_t)P * D, D, nullptr, g->vpt_post_w, g->vpt_post_b, P, dv, D};
        G.d[2] = GradDesc{0, w.d_x2, D, w.g, 4 * D, nullptr, g->proj_w, g->proj_b, R, D, 4 * D};
        G.d[3] = GradDesc{0, w.dh, 4 * D, w.z, D, nullptr, g->fc_w, g->fc_b, R, 4 * D, D};
        G.d[4] = GradDesc{0, w.d_x1, D, w.o, D, nullptr, g->out_w, g->out_b, R, D, D};
        G.d[5] = GradDesc{0, w.dqkv, 3 * D, w.y1, D, nullptr, g->in_w, g->in_b, R, 3 * D, D};
        G.d[6] = GradDesc{0, w.d_x0, D, m->coop, dt, nullptr, g->coop_pre_w, g->coop_pre_b, P, D, dt};
        G.d[7] = GradDesc{0, w.d_x0 + (size_t)P * D, D, m->vpt, dv, nullptr, g->vpt_pre_w, g->vpt_pre_b, P, D, dv};
        G.d[8] = GradDesc{1, w.dz, D, w.x1, D, w.st2, g->ln2_g, g->ln2_b, R, 1, D};
        G.d[9] = GradDesc{1, w.dy, D, w.x0, D, w.st1, g->ln1_g, g->ln1_b, R, 1, D};
        int64_t most = (int64_t)4 * D * D;
        most = (int64_t)dt * D > most ? (int64_t)dt * D : most;
        most = (int64_t)dv * D > most ? (int64_t)dv * D : most;
        hipLaunchKernelGGL(mixer_param_grad_kernel, dim3((unsigned)((most + 255) / 256), 10), dim3(256), 0, s, G);
        GRIP_CHECK_HIP(hipGetLastError());
        return GRIP_OK;
    } catch (...) { grip_set_error("upt_mixer_backward: exception"); return GRIP_ERR_ARG; }
}
