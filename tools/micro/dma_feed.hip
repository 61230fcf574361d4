// Developer microbenchmark: how fast can 256 workgroups stream GEMM operand tiles L2/HBM -> LDS with global_load_lds,
// (A) as 16 rows x 64 B per wave instruction (K tile 32, the ring of gemm_big_kernel) or (B) as 8 rows x 128 B (K tile 64,
// whole cache lines)?  No MFMA, no LDS reads: just the DMA stream with counted waits and one barrier per step.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_feed dma_feed.hip ; run: ./dma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half_t;
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void feed(const half_t* A, const half_t* W, int M, int N, int K, int tiles_m, int tiles_n, float* sink) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int BKT = MODE == 0 ? 32 : 64;
    constexpr int ROWS_PER_INST = MODE == 0 ? 16 : 8;
    constexpr int INST = 512 / ROWS_PER_INST / NW;          // per wave per stage (A and W rows together)
    constexpr int STAGE = 512 * BKT;                        // halfs
    constexpr int NST = MODE == 0 ? 4 : 2;
    const int srow = MODE == 0 ? lane >> 2 : lane >> 3;
    const int schunk = MODE == 0 ? lane & 3 : lane & 7;
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < INST; ++i) {
            const int r = (wave * INST + i) * ROWS_PER_INST + srow;     // 0..511: first 256 = A rows, rest = W rows
            const half_t* src = r < 256 ? A + (size_t)(m0 + r) * K : W + (size_t)(n0 + r - 256) * K;
            __builtin_amdgcn_global_load_lds((const AS1 void*)(src + (size_t)kt * BKT + schunk * 8),
                                             (AS3 void*)(lds + buf * STAGE + (wave * INST + i) * ROWS_PER_INST * BKT), 16, 0, 0);
        }
    };
    const int nk = K / BKT;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) stage(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NST - 1 < nk) { stage((kt + NST - 1) % NST, kt + NST - 1); wait_vmcnt<(NST - 1) * INST>(); }
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = (float)lds[threadIdx.x];
}

int main() {
    const int M = 86784, shapes[4][2] = {{768, 3072}, {2304, 768}, {3072, 768}, {768, 768}};
    half_t *A, *W; float* sink;
    hipMalloc(&A, (size_t)M * 3072 * 2); hipMalloc(&W, (size_t)3072 * 3072 * 2); hipMalloc(&sink, 1 << 20);
    hipMemset(A, 0x11, (size_t)M * 3072 * 2); hipMemset(W, 0x22, (size_t)3072 * 3072 * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1], tiles_m = M / 256, tiles_n = N / 256;
        const double bytes = (double)tiles_m * tiles_n * 512.0 * K * 2;
        auto run = [&](auto kern, int nw, size_t ldsb, const char* name) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(nw * 64), ldsb, 0, A, W, M, N, K, tiles_m, tiles_n, sink);
            hipEventRecord(a);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(nw * 64), ldsb, 0, A, W, M, N, K, tiles_m, tiles_n, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
            printf("N=%d K=%d %-28s %.3f ms  %.2f TB/s into LDS  (= %.0f TF/s of GEMM if fully hidden)\n", N, K, name, ms, bytes / ms / 1e9,
                   2.0 * M * N * K / ms / 1e9);
        };
        run(feed<0, 8>, 8, 131072, "64B rows, 8 waves, 4x32K");
        run(feed<0, 4>, 4, 131072, "64B rows, 4 waves, 4x32K");
        run(feed<1, 8>, 8, 131072, "128B rows, 8 waves, 2x64K");
        run(feed<1, 4>, 4, 131072, "128B rows, 4 waves, 2x64K");
    }
    return 0;
}
