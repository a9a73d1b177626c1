// Diagnostic: sustained throughput of LDS-fed MFMA loops on RANDOM f16 operands (the chip
// lowers its clock under load, and by how much depends on the data and on the MFMA shape:
// MI355X_MICROARCH.md, DVFS give-back items 1 and 7). Two waves per SIMD, every CU busy.
//   mfma_shape <variant> <seconds>
//     0: v_mfma_f32_32x32x16_f16, no LDS reads in the loop
//     1: v_mfma_f32_32x32x16_f16, 11 ds_read_b128 per 18 MFMAs (the z-column kernel's ratio)
//     2: v_mfma_f32_16x16x32_f16, no LDS reads
//     3: v_mfma_f32_16x16x32_f16, 21 ds_read_b128 per 36 MFMAs (same FLOPs as variant 1's 18)
//     4: v_mfma_f32_16x16x32_f16, 11 ds_read_b128 per 36 MFMAs
//     5: v_mfma_f32_32x32x16_f16, 18 reads per 18 MFMAs
//     6: v_mfma_f32_16x16x32_f16, 36 reads per 36 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int R>
__global__ __launch_bounds__(256, 2) void feed(const uint4* __restrict__ src, float* out, int iters) {
    __shared__ uint4 lds[4096];   // 64 KiB
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 0xfffff];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    constexpr int NF = 24;
    uint4 fr[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) fr[k] = lds[(lane * 7 + k * 64) & 4095];
    float s = 0;
    if (SHAPE == 32) {
        f32x16 acc[6] = {};
        for (int i = 0; i < iters; ++i) {
            const int base = (i * 192 + lane) & 4095;
#pragma unroll
            for (int k = 0; k < R; ++k) fr[k % NF] = lds[(base + k * 64) & 4095];
#pragma unroll
            for (int k = 0; k < 18; ++k)
                acc[k % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(f16x8, fr[(k + 5) % NF]), __builtin_bit_cast(f16x8, fr[k % NF]), acc[k % 6], 0, 0, 0);
        }
        for (int k = 0; k < 6; ++k) for (int j = 0; j < 16; ++j) s += acc[k][j];
    } else {
        f32x4 acc[24] = {};
        for (int i = 0; i < iters; ++i) {
            const int base = (i * 192 + lane) & 4095;
#pragma unroll
            for (int k = 0; k < R; ++k) fr[k % NF] = lds[(base + k * 64) & 4095];
#pragma unroll
            for (int k = 0; k < 36; ++k)
                acc[k % 24] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    __builtin_bit_cast(f16x8, fr[(k + 5) % NF]), __builtin_bit_cast(f16x8, fr[k % NF]), acc[k % 24], 0, 0, 0);
        }
        for (int k = 0; k < 24; ++k) for (int j = 0; j < 4; ++j) s += acc[k][j];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int v = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 1.0;
    uint4* src; float* out;
    const size_t n = 1 << 20;
    hipMalloc(&src, n * 16);
    std::vector<unsigned short> h(n * 8);
    unsigned long long s = 88172645463325252ull;
    for (auto& x : h) {   // random halves in +-[0.25, 1): random sign and mantissa, exponent 13 or 14
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x = (unsigned short)(((s >> 20) & 0x8000) | ((13 + ((s >> 40) & 1)) << 10) | (s & 0x3ff));
    }
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int iters) {
        switch (v) {
            case 0: feed<32, 0><<<512, 256>>>(src, out, iters); break;
            case 1: feed<32, 11><<<512, 256>>>(src, out, iters); break;
            case 2: feed<16, 0><<<512, 256>>>(src, out, iters); break;
            case 3: feed<16, 21><<<512, 256>>>(src, out, iters); break;
            case 4: feed<16, 11><<<512, 256>>>(src, out, iters); break;
            case 5: feed<32, 18><<<512, 256>>>(src, out, iters); break;
            case 6: feed<16, 36><<<512, 256>>>(src, out, iters); break;
            default: printf("variant 0..6\n"); exit(2);
        }
    };
    run(1000);
    hipDeviceSynchronize();
    int iters = 100000;
    hipEventRecord(e0); run(iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * secs * 1e3 / ms);
    hipEventRecord(e0); run(iters); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_iter = (v == 0 || v == 1 || v == 5) ? 18.0 * 32768 : 36.0 * 16384;
    printf("variant %d: %.1f TFLOP/s over %.2f s\n", v, 512.0 * 4 * iters * flop_per_iter / (ms * 1e-3) * 1e-12, ms * 1e-3);
    return 0;
}
