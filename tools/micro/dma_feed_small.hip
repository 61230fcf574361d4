// Developer microbenchmark: the operand feed of the small-M GEMMs (prompt steps: M = 2 688 / 3 408 rows, N = 768, K = 3 072) on its
// own -- global_load_lds of (BM + 128) rows x 128 B per K step into an NST-deep LDS ring, one barrier per step, no fragment reads,
// no MFMA -- as a function of the number of waves that issue the pieces (1 KiB = 8 rows x 128 B per wave instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_feed_small tools/micro/dma_feed_small.hip ; run: /tmp/dma_feed_small
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half_t;
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int NW, int NST>
__global__ __launch_bounds__(NW * 64) void feed(const half_t* A, const half_t* W, int K, int tiles_m, int tiles_n, float* sink) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * 128;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PIECES = (BM + 128) / 8;          // 1 KiB pieces per stage
    constexpr int INST = PIECES / NW;               // per wave
    constexpr int STAGE = (BM + 128) * 64;          // halfs
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < INST; ++i) {
            const int r = (wave * INST + i) * 8 + srow;
            const half_t* src = r < BM ? A + (size_t)(m0 + r) * K : W + (size_t)(n0 + r - BM) * K;
            __builtin_amdgcn_global_load_lds((const AS1 void*)(src + (size_t)kt * 64 + schunk * 8), (AS3 void*)(lds + buf * STAGE + (wave * INST + i) * 8 * 64), 16, 0, 0);
        }
    };
    const int nk = K / 64;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) stage(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NST - 1 < nk) { wait_vmcnt<(NST - 2) * INST>(); __builtin_amdgcn_s_barrier(); stage((kt + NST - 1) % NST, kt + NST - 1); }
        else { wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }
    }
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = (float)lds[threadIdx.x];
}

int main() {
    const int K = 3072, N = 768;
    half_t *A, *W; float* sink;
    hipMalloc(&A, (size_t)4096 * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&sink, 1 << 20);
    hipMemset(A, 0x11, (size_t)4096 * K * 2); hipMemset(W, 0x22, (size_t)N * K * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int M : {2688, 3408}) {
        auto run = [&](auto kern, int bm, int nw, int nst, const char* name) {
            const int tiles_m = (M + bm - 1) / bm, tiles_n = N / 128;
            const size_t ldsb = (size_t)nst * (bm + 128) * 64 * 2;
            const double bytes = (double)tiles_m * tiles_n * (bm + 128) * K * 2.0;
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(nw * 64), ldsb, 0, A, W, K, tiles_m, tiles_n, sink);
            hipEventRecord(a);
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(nw * 64), ldsb, 0, A, W, K, tiles_m, tiles_n, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 50;
            printf("M=%d %-34s %3d wgs  %6.2f us  %5.2f TB/s into LDS  %5.1f GB/s per wg\n", M, name, tiles_m * tiles_n, ms * 1e3, bytes / ms / 1e9,
                   (bm + 128) * K * 2.0 / ms / 1e6);
        };
        run(feed<64, 4, 3>, 64, 4, 3, "64x128, 4 waves, 3 stages");
        run(feed<64, 4, 6>, 64, 4, 6, "64x128, 4 waves, 6 stages");
        run(feed<64, 8, 3>, 64, 8, 3, "64x128, 8 waves, 3 stages");
        run(feed<64, 8, 6>, 64, 8, 6, "64x128, 8 waves, 6 stages");
        run(feed<64, 12, 3>, 64, 12, 3, "64x128, 12 waves, 3 stages");
        run(feed<64, 12, 6>, 64, 12, 6, "64x128, 12 waves, 6 stages");
        run(feed<128, 4, 4>, 128, 4, 4, "128x128, 4 waves, 4 stages");
        run(feed<128, 8, 4>, 128, 8, 4, "128x128, 8 waves, 4 stages");
        run(feed<128, 16, 4>, 128, 16, 4, "128x128, 16 waves, 4 stages");
        run(feed<128, 16, 5>, 128, 16, 5, "128x128, 16 waves, 5 stages");
    }
    return 0;
}
