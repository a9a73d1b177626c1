// Diagnostic: throughput and (via rocm-smi, sampled by the caller) power of an MFMA
// loop fed from LDS at a chosen ds_read_b128 : MFMA ratio. 2 waves per SIMD.
//   mfma_power <reads per 18 MFMAs> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int R>
__global__ __launch_bounds__(256, 2) void feed(const uint4* __restrict__ src, float* out, int iters) {
    __shared__ uint4 lds[4096];   // 64 KiB
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
    __syncthreads();
    f32x16 acc[6] = {};
    const int lane = threadIdx.x & 63;
    uint4 fr[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) fr[k] = lds[(lane * 7 + k * 64) & 4095];
    for (int i = 0; i < iters; ++i) {
        const int base = (i * 192 + lane) & 4095;
#pragma unroll
        for (int k = 0; k < R; ++k) fr[k] = lds[(base + k * 64) & 4095];
#pragma unroll
        for (int k = 0; k < 18; ++k)
            acc[k % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[(k + 5) % 18]),
                                                                __builtin_bit_cast(bf16x8, fr[k]), acc[k % 6], 0, 0, 0);
    }
    float s = 0;
    for (int k = 0; k < 6; ++k) for (int j = 0; j < 16; ++j) s += acc[k][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int r = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 1.0;
    uint4* src; float* out;
    hipMalloc(&src, (1 << 20) * 16);
    hipMemset(src, 0x3c, (1 << 20) * 16);   // bf16 0x3c3c = small normal numbers
    hipMalloc(&out, 512 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int iters) {
        switch (r) {
            case 0: feed<0><<<512, 256>>>(src, out, iters); break;
            case 6: feed<6><<<512, 256>>>(src, out, iters); break;
            case 11: feed<11><<<512, 256>>>(src, out, iters); break;
            case 18: feed<18><<<512, 256>>>(src, out, iters); break;
            default: printf("reads must be 0, 6, 11 or 18\n"); exit(2);
        }
    };
    run(1000);
    hipDeviceSynchronize();
    int iters = 200000;
    hipEventRecord(e0); run(iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * secs * 1e3 / ms);
    hipEventRecord(e0); run(iters); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    printf("reads %d per 18 MFMAs: %.1f TFLOP/s over %.2f s\n", r, 512.0 * 4 * iters * 18 * 32768 / (ms * 1e-3) * 1e-12, ms * 1e-3);
    return 0;
}
