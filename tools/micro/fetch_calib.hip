// FETCH_SIZE calibration on the GEMM's own DMA pattern.  MI355X_MICROARCH.md says rocprofv3's FETCH_SIZE
// reports half the bytes of a wide coalesced streaming read on gfx950 and asks to calibrate other patterns on a known byte
// count.  This program reads every byte of an [R, 768] f16 matrix (R x 1 536 B, far larger than L2 + the 256 MiB Infinity
// Cache) EXACTLY ONCE with the access pattern of the 256x256x64 GEMM's A-operand feed: global_load_lds, 16 B per lane, one
// wave instruction = 8 rows x 128 B at a row stride of 1 536 B, walking K in 64-element steps and then down the rows --
// and, for reference, with plain 16-byte global loads of the same bytes.  Run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
// and compare FETCH_SIZE (KiB) x 1024 per launch with the bytes printed here.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int K = 768;          // halfs per row

// one workgroup of 8 waves per 256-row panel; wave w feeds rows [w*32, +32) of the panel, 8 rows per instruction (as gemm_k64p)
__global__ __launch_bounds__(512) void dma_pattern_kernel(const _Float16* A, int panels) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    for (int p = blockIdx.x; p < panels; p += gridDim.x) {
        const _Float16* src = A + (size_t)(p * 256 + wave * 32 + srow) * K + schunk * 8;
        for (int kt = 0; kt < K / 64; ++kt) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const AS1 void*)(src + (size_t)i * 8 * K + kt * 64), (AS3 void*)(lds + (wave * 32 + i * 8) * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

// the same bytes with plain coalesced 16-byte loads (the pattern the guide's x2 rule was calibrated on)
__global__ __launch_bounds__(256) void stream_kernel(const float4* A, size_t n16, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const float4 v = A[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}

int main() {
    const int panels = 1 << 12;                       // 4096 panels x 256 rows = 1 048 576 rows
    const size_t rows = (size_t)panels * 256, bytes = rows * K * 2;   // 1.61 GB
    _Float16* A;
    float* sink;
    CHECK(hipMalloc(&A, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(A, 0x11, bytes));
    CHECK(hipFuncSetAttribute((const void*)dma_pattern_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 2));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(dma_pattern_kernel, dim3(256), dim3(512), 256 * 64 * 2, 0, A, panels);
        hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)A, bytes / 16, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("bytes read per launch (each kernel, every byte once): %zu  = %.1f KiB\n", bytes, bytes / 1024.0);
    return 0;
}
