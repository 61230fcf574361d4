// Pure-MFMA loop for the clock / power trace: what does v_mfma_f32_16x16x32_f16 reach on this box when
// nothing but the matrix pipes work -- with ZERO operands (no data toggling: the clock stays up) and with RANDOM f16 operands
// (full toggling: the package runs into its power cap and the clock drops) -- and what the shader clock is meanwhile
// (tools/clock_trace.py samples it; this program prints `##clock_trace <label>` markers on stderr around each phase).
//     hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && python tools/clock_trace.py --out ... -- /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ACC = 16;      // independent accumulators per wave: no MFMA waits on the previous one's result

__global__ __launch_bounds__(256) void mfma_loop(const _Float16* src, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // four operand pairs per lane (register-resident; nothing is loaded inside the loop)
        a[i] = *(const half8*)(src + ((size_t)(i * 2) * 64 + lane) * 8);
        b[i] = *(const half8*)(src + ((size_t)(i * 2 + 1) * 64 + lane) * 8);
    }
    f32x4 acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;      // keep the loop alive
}

static double run(const char* label, const _Float16* src, float* sink, double seconds) {
    fprintf(stderr, "##clock_trace %s\n", label);
    fflush(stderr);
    const int iters = 20000, blocks = 256 * 2;       // 2 workgroups of 4 waves per CU: two waves per SIMD
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, src, sink, 100);
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    double el = 0.0;
    do {
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, src, sink, iters);
        CHECK(hipDeviceSynchronize());
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double flops = (double)launches * blocks * 4 * (double)iters * ACC * 2.0 * 16 * 16 * 32;
    printf("%-14s %.1f TFLOP/s over %.2f s (%d launches)\n", label, flops / el / 1e12, el, launches);
    fflush(stdout);
    return flops / el;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t n = 8 * 64 * 8;
    _Float16* h = (_Float16*)malloc(n * 2);
    _Float16 *zero, *rnd;
    float* sink;
    CHECK(hipMalloc(&zero, n * 2));
    CHECK(hipMalloc(&rnd, n * 2));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(zero, 0, n * 2));
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (_Float16)(((rand() % 2001) - 1000) * 1e-3f);
    CHECK(hipMemcpy(rnd, h, n * 2, hipMemcpyHostToDevice));
    fprintf(stderr, "##clock_trace idle\n");
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    run("mfma_zero", zero, sink, seconds);
    run("mfma_random", rnd, sink, seconds);
    run("mfma_zero_2", zero, sink, seconds / 2);
    fprintf(stderr, "##clock_trace idle_after\n");
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    return 0;
}
