// CLIP image preprocessing on the GPU: the `_transform` of openai-CLIP that the reference applies
// three times per item on the host (data/dataset.py:64-79): Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor ->
// Normalize(mean, std).  The resize reproduces Pillow's ImagingResample for 8-bit images bit for bit: two separable
// passes (horizontal, then vertical) with per-output-pixel coefficient windows precomputed by the host in Pillow's
// 22-bit fixed point, an 8-bit intermediate image between the passes, round-half-up and clip to [0, 255].  Only the
// rows / columns that survive the centre crop are produced.  HBM-bound: H*W*3 bytes in, n_px^2 * 3 * 4 bytes out.
#include "common.h"

#define PRECISION_BITS 22   // Pillow: 32 - 8 - 2

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: in [H, W, 3] u8 -> tmp [H, WC, 3] u8 for output columns [x0, x0 + WC)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ in, int H, int W, const int32_t* __restrict__ coef,
                                                         const int32_t* __restrict__ bounds, int ksize, int x0, int WC,
                                                         uint8_t* __restrict__ tmp) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * WC) return;
    const int y = idx / WC, xc = idx - y * WC, xx = x0 + xc;
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int32_t* k = coef + (size_t)xx * ksize;
    const uint8_t* row = in + ((size_t)y * W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const int w = k[x];
        s0 += row[3 * x] * w;
        s1 += row[3 * x + 1] * w;
        s2 += row[3 * x + 2] * w;
    }
    uint8_t* o = tmp + ((size_t)y * WC + xc) * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

// vertical pass + crop + ToTensor + Normalize: tmp [H, WC, 3] u8 -> out [3, HC, WC] f32 for output rows [y0, y0 + HC)
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int H, int WC, const int32_t* __restrict__ coef,
                                                              const int32_t* __restrict__ bounds, int ksize, int y0, int HC,
                                                              float m0, float m1, float m2, float i0, float i1, float i2,
                                                              float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HC * WC) return;
    const int yc = idx / WC, x = idx - yc * WC, yy = y0 + yc;
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int32_t* k = coef + (size_t)yy * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < cnt; ++y) {
        const uint8_t* p = tmp + ((size_t)(ymin + y) * WC + x) * 3;
        const int w = k[y];
        s0 += p[0] * w;
        s1 += p[1] * w;
        s2 += p[2] * w;
    }
    const size_t plane = (size_t)HC * WC;
    out[idx] = ((float)clip8(s0) / 255.0f - m0) * i0;
    out[plane + idx] = ((float)clip8(s1) / 255.0f - m1) * i1;
    out[2 * plane + idx] = ((float)clip8(s2) / 255.0f - m2) * i2;
}

// no resize needed along an axis: coefficient tables may be null (identity), handled by the host passing ksize = 0
__global__ __launch_bounds__(256) void crop_norm_kernel(const uint8_t* __restrict__ in, int W, int x0, int y0, int HC, int WC,
                                                        float m0, float m1, float m2, float i0, float i1, float i2, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HC * WC) return;
    const int yc = idx / WC, xc = idx - yc * WC;
    const uint8_t* p = in + ((size_t)(y0 + yc) * W + x0 + xc) * 3;
    const size_t plane = (size_t)HC * WC;
    out[idx] = ((float)p[0] / 255.0f - m0) * i0;
    out[plane + idx] = ((float)p[1] / 255.0f - m1) * i1;
    out[2 * plane + idx] = ((float)p[2] / 255.0f - m2) * i2;
}

extern "C" int grip_preprocess_image(const uint8_t* img, int H, int W,
                                     const int32_t* hcoef, const int32_t* hbounds, int hksize, int W_out,
                                     const int32_t* vcoef, const int32_t* vbounds, int vksize, int H_out,
                                     int crop_left, int crop_top, int n_px, const float* mean3, const float* std3,
                                     uint8_t* tmp, float* out, void* stream) {
    GRIP_REQUIRE(img && out && mean3 && std3 && H > 0 && W > 0 && n_px > 0, "preprocess: bad arguments");
    GRIP_REQUIRE(crop_left >= 0 && crop_top >= 0 && crop_left + n_px <= W_out && crop_top + n_px <= H_out, "preprocess: crop window outside the resized image");
    hipStream_t s = (hipStream_t)stream;
    const float m0 = mean3[0], m1 = mean3[1], m2 = mean3[2], i0 = 1.f / std3[0], i1 = 1.f / std3[1], i2 = 1.f / std3[2];
    const int blocks = (n_px * n_px + 255) / 256;
    if (hksize == 0 && vksize == 0) {           // already the right size: crop + normalise
        hipLaunchKernelGGL(crop_norm_kernel, dim3(blocks), dim3(256), 0, s, img, W, crop_left, crop_top, n_px, n_px, m0, m1, m2, i0, i1, i2, out);
    } else {
        GRIP_REQUIRE(hksize > 0 && vksize > 0 && hcoef && hbounds && vcoef && vbounds && tmp, "preprocess: coefficient tables missing");
        hipLaunchKernelGGL(resample_h_kernel, dim3((H * n_px + 255) / 256), dim3(256), 0, s, img, H, W, hcoef, hbounds, hksize, crop_left, n_px, tmp);
        hipLaunchKernelGGL(resample_v_norm_kernel, dim3(blocks), dim3(256), 0, s, tmp, H, n_px, vcoef, vbounds, vksize, crop_top, n_px,
                           m0, m1, m2, i0, i1, i2, out);
    }
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}

// ---- batched form: one launch pair for a whole batch of decoded images of DIFFERENT sizes.  blockIdx.y selects the image; its
// descriptor (pointers into one packed upload buffer, its own Pillow coefficient tables, crop window) is read from a device array.
__global__ __launch_bounds__(256) void resample_h_batch_kernel(const grip_preprocess_item* __restrict__ items, int n_px) {
    const grip_preprocess_item it = items[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= it.H * n_px || it.hksize == 0) return;
    const int y = idx / n_px, xc = idx - y * n_px, xx = it.crop_left + xc;
    const int xmin = it.hbounds[2 * xx], cnt = it.hbounds[2 * xx + 1];
    const int32_t* k = it.hcoef + (size_t)xx * it.hksize;
    const uint8_t* row = it.img + ((size_t)y * it.W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const int w = k[x];
        s0 += row[3 * x] * w;
        s1 += row[3 * x + 1] * w;
        s2 += row[3 * x + 2] * w;
    }
    uint8_t* o = it.tmp + ((size_t)y * n_px + xc) * 3;
    o[0] = (uint8_t)clip8(s0);
    o[1] = (uint8_t)clip8(s1);
    o[2] = (uint8_t)clip8(s2);
}

__global__ __launch_bounds__(256) void resample_v_norm_batch_kernel(const grip_preprocess_item* __restrict__ items, int n_px,
                                                                    float m0, float m1, float m2, float i0, float i1, float i2) {
    const grip_preprocess_item it = items[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_px * n_px) return;
    const int yc = idx / n_px, x = idx - yc * n_px;
    int v0, v1, v2;
    if (it.hksize == 0) {                      // already the resized size: crop only
        const uint8_t* p = it.img + ((size_t)(it.crop_top + yc) * it.W + it.crop_left + x) * 3;
        v0 = p[0]; v1 = p[1]; v2 = p[2];
    } else {
        const int yy = it.crop_top + yc;
        const int ymin = it.vbounds[2 * yy], cnt = it.vbounds[2 * yy + 1];
        const int32_t* k = it.vcoef + (size_t)yy * it.vksize;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < cnt; ++y) {
            const uint8_t* p = it.tmp + ((size_t)(ymin + y) * n_px + x) * 3;
            const int w = k[y];
            s0 += p[0] * w;
            s1 += p[1] * w;
            s2 += p[2] * w;
        }
        v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
    }
    const size_t plane = (size_t)n_px * n_px;
    it.out[idx] = ((float)v0 / 255.0f - m0) * i0;
    it.out[plane + idx] = ((float)v1 / 255.0f - m1) * i1;
    it.out[2 * plane + idx] = ((float)v2 / 255.0f - m2) * i2;
}

extern "C" int grip_preprocess_batch(const grip_preprocess_item* items_device, int n_items, int max_H, int n_px,
                                     const float* mean3, const float* std3, void* stream) {
    GRIP_REQUIRE(items_device && mean3 && std3 && n_items > 0 && max_H > 0 && n_px > 0, "preprocess_batch: bad arguments");
    GRIP_REQUIRE(n_items <= 65535, "preprocess_batch: at most 65535 images per launch");
    hipStream_t s = (hipStream_t)stream;
    const float m0 = mean3[0], m1 = mean3[1], m2 = mean3[2], i0 = 1.f / std3[0], i1 = 1.f / std3[1], i2 = 1.f / std3[2];
    hipLaunchKernelGGL(resample_h_batch_kernel, dim3((max_H * n_px + 255) / 256, n_items), dim3(256), 0, s, items_device, n_px);
    hipLaunchKernelGGL(resample_v_norm_batch_kernel, dim3((n_px * n_px + 255) / 256, n_items), dim3(256), 0, s, items_device, n_px, m0, m1, m2, i0, i1, i2);
    GRIP_CHECK_HIP(hipGetLastError());
    return GRIP_OK;
}
