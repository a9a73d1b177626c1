// Measurement aid (not part of the library): what would an F(2, 3) Winograd transform along z buy the
// z-column convolution under the package power limit? Two LDS-fed inner loops on random IEEE halves, one
// iteration = one in-plane tap (dy, dx) of a six-plane tile of one wave, i.e. the same useful work:
//   direct : 8 operand reads + 3 weight reads, 18 MFMAs into 6 accumulator tiles, two waves per SIMD
//            (512 workgroups of 4 waves, 256 registers each) -- the shape of conv3x3x3_zpipe's tap loop
//   wino   : 8 operand reads + 4 transformed-weight reads, 12 transformed operands (4 packed-half
//            additions each), 12 MFMAs into 12 accumulator tiles (192 registers: AGPRs), one wave per SIMD
//            (256 workgroups of 4 waves, up to 512 registers each)
// Reported: iterations per second and the "useful" rate 18 x 32768 FLOP per iteration in both cases.
//   hipcc -O3 --offload-arch=gfx950 tools/wino_probe.hip -o tools/wino_probe && tools/wino_probe [seconds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint4 pk_add(const uint4& a, const uint4& b) {
    auto add = [](unsigned x, unsigned y) {
        return __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, x) + __builtin_bit_cast(f16x2, y));
    };
    return make_uint4(add(a.x, b.x), add(a.y, b.y), add(a.z, b.z), add(a.w, b.w));
}
__device__ __forceinline__ uint4 pk_sub(const uint4& a, const uint4& b) {
    auto sub = [](unsigned x, unsigned y) {
        return __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, x) - __builtin_bit_cast(f16x2, y));
    };
    return make_uint4(sub(a.x, b.x), sub(a.y, b.y), sub(a.z, b.z), sub(a.w, b.w));
}
__device__ __forceinline__ f32x16 mma(const f32x16& acc, const uint4& a, const uint4& b) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

#ifndef DIRECT_ORDER
#define DIRECT_ORDER 2
#endif
constexpr int kSlots = 4096;   // 64 KiB of LDS per workgroup

__global__ __launch_bounds__(256, 2) void direct(const uint4* __restrict__ src, float* out, int iters) {
    __shared__ uint4 lds[kSlots];
    for (int i = threadIdx.x; i < kSlots; i += 256) lds[i] = src[(blockIdx.x * kSlots + i) & 0xfffff];
    __syncthreads();
    f32x16 acc[6] = {};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < iters; ++i) {
        const int base = (i * 704 + wave * 1024 + lane) & (kSlots - 1);
        uint4 d[8], w[3];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = lds[(base + k * 64) & (kSlots - 1)];
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = lds[(base + (8 + k) * 64) & (kSlots - 1)];
#if DIRECT_ORDER == 1      // weight-stationary: one weight fragment for six consecutive MFMAs
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int z = 0; z < 6; ++z) acc[z] = mma(acc[z], w[dz], d[z + dz]);
#elif DIRECT_ORDER == 2    // operand-stationary: one input plane for up to three consecutive MFMAs (the kernel's order)
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
                if (k - dz >= 0 && k - dz < 6) acc[k - dz] = mma(acc[k - dz], w[dz], d[k]);
#else                      // accumulator-stationary
#pragma unroll
        for (int z = 0; z < 6; ++z)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) acc[z] = mma(acc[z], w[dz], d[z + dz]);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int k = 0; k < 6; ++k) for (int j = 0; j < 16; ++j) s += acc[k][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 1) void wino(const uint4* __restrict__ src, float* out, int iters) {
    __shared__ uint4 lds[kSlots];
    for (int i = threadIdx.x; i < kSlots; i += 256) lds[i] = src[(blockIdx.x * kSlots + i) & 0xfffff];
    __syncthreads();
    f32x16 m[3][4] = {};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (operands of the next iteration are read before this iteration's MFMAs: one wave per SIMD has nobody
    // else to hide the LDS latency)
    uint4 dn[8], un[4];
    {
        const int base = (wave * 1024 + lane) & (kSlots - 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) dn[k] = lds[(base + k * 64) & (kSlots - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) un[k] = lds[(base + (8 + k) * 64) & (kSlots - 1)];
    }
    for (int i = 0; i < iters; ++i) {
        const int base = ((i + 1) * 768 + wave * 1024 + lane) & (kSlots - 1);
        uint4 d[8], u[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = dn[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = un[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) dn[k] = lds[(base + k * 64) & (kSlots - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) un[k] = lds[(base + (8 + k) * 64) & (kSlots - 1)];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint4 t0 = pk_sub(d[2 * p], d[2 * p + 2]);
            const uint4 t1 = pk_add(d[2 * p + 1], d[2 * p + 2]);
            const uint4 t2 = pk_sub(d[2 * p + 2], d[2 * p + 1]);
            const uint4 t3 = pk_sub(d[2 * p + 1], d[2 * p + 3]);
            m[p][0] = mma(m[p][0], u[0], t0);
            m[p][1] = mma(m[p][1], u[1], t1);
            m[p][2] = mma(m[p][2], u[2], t2);
            m[p][3] = mma(m[p][3], u[3], t3);
        }
    }
    float s = 0;
    for (int p = 0; p < 3; ++p)
        for (int j = 0; j < 16; ++j) s += (m[p][0][j] + m[p][1][j] + m[p][2][j]) + (m[p][1][j] - m[p][2][j] - m[p][3][j]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    uint4* src;
    float* out;
    std::vector<unsigned short> h((1 << 20) * 8);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) {   // halves of magnitude 2^-3 .. 2^0 with random sign and mantissa
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = (unsigned short)(((s >> 20) & 0x8000) | (0x3000 + ((s >> 8) & 0x0fff)));
    }
    hipMalloc(&src, h.size() * 2);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        const int wgs = which == 0 ? 512 : 256;
        auto run = [&](int iters) {
            if (which == 0) direct<<<wgs, 256>>>(src, out, iters); else wino<<<wgs, 256>>>(src, out, iters);
        };
        run(1000);
        hipDeviceSynchronize();
        int iters = 100000;
        float ms;
        hipEventRecord(e0); run(iters); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        iters = (int)(iters * secs * 1e3 / ms);
        hipEventRecord(e0); run(iters); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        const double tile_steps = (double)wgs * 4 * iters / (ms * 1e-3);
        printf("%-6s %d workgroups: %.3e tile-steps/s = %.1f useful TFLOP/s (issued %.1f) over %.2f s\n",
               which == 0 ? "direct" : "wino", wgs, tile_steps, tile_steps * 18 * 32768 * 1e-12,
               tile_steps * (which == 0 ? 18 : 12) * 32768 * 1e-12, ms * 1e-3);
    }
    return 0;
}
