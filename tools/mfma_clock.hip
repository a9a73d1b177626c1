// Diagnostic: how many s_memtime ticks one v_mfma_f32_32x32x16_bf16 costs when a
// SIMD runs nothing else (1 or 2 waves per SIMD, whole chip busy), and the tick
// rate against the 100 MHz real-time counter. Calibrates tools/conv_trace numbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void spin(unsigned long long* out, int iters) {
    f32x16 acc[4] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.5f); }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0;
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 12345.f) out[0] = 0;   // keep the accumulators alive
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

int main() {
    const int iters = getenv("MFMA_ITERS") ? atoi(getenv("MFMA_ITERS")) : 50000;
    for (int threads : {256}) {
        const int blocks = 256;
        const size_t waves = (size_t)blocks * threads / 64;
        unsigned long long* d;
        hipMalloc(&d, waves * 16);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        spin<<<blocks, threads>>>(d, 1000);
        hipEventRecord(e0);
        spin<<<blocks, threads>>>(d, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(waves * 2);
        hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
        double st = 0, sr = 0;
        for (size_t w = 0; w < waves; ++w) { st += h[2 * w]; sr += h[2 * w + 1]; }
        st /= waves; sr /= waves;
        const double nm = 4.0 * iters;
        printf("%d waves/SIMD: %.3f ms; per MFMA: %.2f memtime ticks, %.3f realtime ticks (%.2f ns by events); "
               "memtime/realtime = %.2f; chip %.1f TFLOP/s\n", threads / 256, ms, st / nm, sr / nm,
               ms * 1e6 / nm / (threads / 256), st / sr, waves * nm * 32768.0 / (ms * 1e-3) * 1e-12);
        hipFree(d);
    }
    return 0;
}
