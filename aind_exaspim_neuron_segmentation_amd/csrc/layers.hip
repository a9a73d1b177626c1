// The layers of the U-Net around the 3x3x3 convolutions, on the blocked
// channels-last layout of conv3d.hip, (N, C / chunk, D, H, W, 32 B):
//   conv_first : inc.0, Conv3d(1 -> C0, k3, p1) + folded BN + LeakyReLU
//                (machine_learning/unet3d.py:64,143-145) from the float32 patch;
//                exact fp32 MFMA in fp32 mode, split 16-bit operands otherwise
//   convt2     : nn.ConvTranspose3d(k=2, s=2) of UNet3D(trilinear=False) (unet3d.py:254)
//   maxpool2   : nn.MaxPool3d(2)                          (unet3d.py:195)
//   upsample2  : nn.Upsample(x2, trilinear, align_corners=True) (unet3d.py:248)
//   head       : OutConv 1x1x1 (+ sigmoid of inference.py:158) -> NCDHW float32
// Every thread moves 16-byte channel groups; consecutive lanes touch
// consecutive addresses. Max-pool and interpolation act per channel, so they see
// the tensor as N * C / chunk independent volumes of 32-byte voxel records.

#include <cmath>
#include <cstdlib>

#include "common.h"

// This file is compiled with -ffp-contract=off (Makefile): the interpolation coordinates
// below must come out of a rounded product and a rounded difference like ATen's, and the
// same in every inlined copy; every fused multiply-add these kernels want is written out.

namespace exaspim {

struct F32T {
    static constexpr int kG = 4;
    using vec = float4;
    __device__ static void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
        f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    __device__ static uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]),
                          __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
struct BF16T {
    static constexpr int kG = 8;
    __device__ static void unpack(const uint4& u, float* f) {
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static uint4 pack(const float* f) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __bf16 lo = (__bf16)f[2 * i], hi = (__bf16)f[2 * i + 1];
            w[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) |
                   ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};
struct F16T {
    static constexpr int kG = 8;
    __device__ static void unpack(const uint4& u, float* f) {
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] & 0xffffu));
            f[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] >> 16));
        }
    }
    __device__ static uint4 pack(const float* f) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // saturating conversion (largest finite half instead of +-inf)
            const _Float16 lo = (_Float16)__builtin_amdgcn_fmed3f(f[2 * i], -65504.f, 65504.f);
            const _Float16 hi = (_Float16)__builtin_amdgcn_fmed3f(f[2 * i + 1], -65504.f, 65504.f);
            w[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) |
                   ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// ---- inc.0 ------------------------------------------------------------------
// Cin = 1, so the contraction is only over the 27 taps: per 32 voxels x 32
// output channels it is a 32 x 28 x 32 GEMM (27 taps + one zero column), run as
// 14 exact-fp32 v_mfma_f32_32x32x2_f32. A = weights W[cout][tap] (14 floats per
// lane, loaded once per wave), B = X[tap][voxel] gathered straight from a copy
// of the float32 patch with a one-voxel zero border (pad_input_kernel), so a tap
// is one add and one load with no bounds arithmetic. The patch is read in fp32
// whatever the network's storage type: inc.0 adds no input rounding.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// (n, d, h, w) float32 -> (n, d+2, h+2, w+2) with a zero border. One block per padded plane.
__global__ __launch_bounds__(256) void pad_input_kernel(const float* __restrict__ x,
                                                        float* __restrict__ xp, int d, int h, int w) {
    const int nb = blockIdx.x / (d + 2), pz = blockIdx.x - nb * (d + 2);
    const int z = pz - 1;
    const int pw = w + 2, plane = (h + 2) * pw;
    float* const dst = xp + (size_t)blockIdx.x * plane;
    if ((unsigned)z >= (unsigned)d) {
        for (int i = threadIdx.x; i < plane; i += blockDim.x) dst[i] = 0.f;
        return;
    }
    const float* const src = x + ((size_t)nb * d + z) * h * w;
    for (int i = threadIdx.x; i < plane; i += blockDim.x) {
        const int py = i / pw, px = i - py * pw;
        const int y = py - 1, xx = px - 1;
        dst[i] = ((unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)w) ? src[y * w + xx] : 0.f;
    }
}

template <typename T, int MT>
__global__ __launch_bounds__(256) void conv_first_kernel(
    const float* __restrict__ xp, const float* __restrict__ w,
    const float* __restrict__ bias, void* __restrict__ dst, int nvox, int d, int h, int wd,
    int c0p, float slope) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, r = lane & 31;
    const int co_tile = blockIdx.y * 32;
    const int hw = h * wd, dhw = d * hw;
    const int pw = wd + 2, phw = (h + 2) * pw;   // padded row / plane strides

    // this lane's 14 taps (k = 2 * ks + half): weight and offset in the padded patch
    float wk[14];
    int rel[14];
#pragma unroll
    for (int ks = 0; ks < 14; ++ks) {
        const int t = 2 * ks + half;
        const bool real = t < 27;
        const int tt = real ? t : 0;
        wk[ks] = real ? w[tt * c0p + co_tile + r] : 0.f;   // zero weight for the 28th column
        rel[ks] = (tt / 9) * phw + ((tt / 3) % 3) * pw + tt % 3;
    }

    const int v0 = (blockIdx.x * 4 + wave) * (MT * 32);
    f32x16_t acc[MT];
    float xv[MT][14];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int v = v0 + mt * 32 + r;
        const int vc = v < nvox ? v : nvox - 1;
        const int nb = vc / dhw, sp = vc - nb * dhw;
        const int zz = sp / hw, yy = (sp - zz * hw) / wd, xx = sp - zz * hw - yy * wd;
        // padded address of tap (0,0,0) = voxel (zz-1, yy-1, xx-1)
        const float* base = xp + ((size_t)nb * (d + 2) + zz) * phw + yy * pw + xx;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) xv[mt][ks] = base[rel[ks]];
    }
    // accumulators start from the folded bias (register 4q+k = channel 8q + 4*half + k)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(bias + co_tile + 8 * q + 4 * half);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt][4 * q + 0] = b.x; acc[mt][4 * q + 1] = b.y;
            acc[mt][4 * q + 2] = b.z; acc[mt][4 * q + 3] = b.w;
        }
    }
#pragma unroll
    for (int ks = 0; ks < 14; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[ks], xv[mt][ks], acc[mt], 0, 0, 0);

    // epilogue: bias + LeakyReLU, then through LDS so that each store instruction
    // writes whole 16-byte pieces of consecutive voxel records
    constexpr int G = T::kG;
    constexpr int ES = 16 / G;
    constexpr int RECB = 32 * ES;           // bytes of one voxel's 32-channel slice
    constexpr int RECP = RECB + 16;         // padded LDS stride (2-way instead of 8-way conflicts)
    constexpr int CPT = RECB / 32;          // chunk planes of a 32-channel slice
    __shared__ __attribute__((aligned(16))) char tr[4 * 32 * RECP];
    char* wl = tr + wave * (32 * RECP);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = 8 * q + 4 * half;
            float o[4] = {acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], o[j] * slope);  // LeakyReLU, 0 <= slope <= 1
            char* slot = wl + r * RECP + cl * ES;
            if (G == 4) {
                *reinterpret_cast<float4*>(slot) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                float o8[8] = {o[0], o[1], o[2], o[3], 0.f, 0.f, 0.f, 0.f};
                const uint4 pk = T::pack(o8);
                *reinterpret_cast<uint2*>(slot) = make_uint2(pk.x, pk.y);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // one store instruction = one chunk plane's 32 voxel records
        const int vv = lane >> 1, sub = lane & 1;
        const int v = v0 + mt * 32 + vv;
        const int vc = v < nvox ? v : nvox - 1;
        const int nb = vc / dhw, sp = vc - nb * dhw;
        char* const dplane = static_cast<char*>(dst) +
                             ((size_t)nb * (c0p * ES / 32) + blockIdx.y * CPT) * dhw * 32;
#pragma unroll
        for (int ck = 0; ck < CPT; ++ck) {
            const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + (ck * 2 + sub) * 16);
            if (v < nvox)
                *reinterpret_cast<uint4*>(dplane + ((size_t)ck * dhw + sp) * 32 + sub * 16) = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- ConvTranspose3d(k=2, s=2) ------------------------------------------------
// Up block of UNet3D(trilinear=False) (unet3d.py:254-258). Kernel 2 with stride
// 2 has no tap overlap: out[2z+dz, 2y+dy, 2x+dx, :] = b + W[:, :, dz, dy, dx]^T in[z, y, x, :],
// i.e. eight independent 1x1x1 GEMMs that scatter to the 2x2x2 output phases.
// One wave takes 32 input voxels x 32 output channels x 8 phases (8 accumulator
// tiles); the B operand is the voxel's 16-byte channel group straight from
// global memory (one load feeds the eight MFMAs of a chunk), the A operand the
// phase's weight fragment (plan.cpp). ~4 % of the 3x3x3 FLOPs of the network.
typedef float f32x16_ct __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_ct __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_ct __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ void mma_ct(f32x16_ct& acc, const uint4& wf, const uint4& xf);
template <>
__device__ __forceinline__ void mma_ct<F32T>(f32x16_ct& acc, const uint4& wf, const uint4& xf) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.x), __uint_as_float(xf.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.y), __uint_as_float(xf.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.z), __uint_as_float(xf.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.w), __uint_as_float(xf.w), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_ct<BF16T>(f32x16_ct& acc, const uint4& wf, const uint4& xf) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_ct, wf),
                                                  __builtin_bit_cast(bf16x8_ct, xf), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_ct<F16T>(f32x16_ct& acc, const uint4& wf, const uint4& xf) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_ct, wf),
                                                 __builtin_bit_cast(f16x8_ct, xf), acc, 0, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(256) void convt2_kernel(const uint4* __restrict__ src,
                                                     const uint4* __restrict__ wts,
                                                     const float* __restrict__ bias,
                                                     void* __restrict__ dst, int nvox, int d, int h,
                                                     int w, int cin, int cout) {
    constexpr int G = T::kG;
    constexpr int ES = 16 / G;
    constexpr int KC = 2 * G;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, r = lane & 31;
    const int ntiles = cout >> 5, ntile = blockIdx.y;
    const int nchunks = cin / KC;
    const int v = (blockIdx.x * 4 + wave) * 32 + r;   // input voxel (flat over the batch)
    const int vc = v < nvox ? v : nvox - 1;

    f32x16_ct acc[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(bias + ntile * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int ph = 0; ph < 8; ++ph) {
            acc[ph][4 * q + 0] = b.x; acc[ph][4 * q + 1] = b.y;
            acc[ph][4 * q + 2] = b.z; acc[ph][4 * q + 3] = b.w;
        }
    }
    const int hw = h * w, dhw = d * hw;
    const int nbc = vc / dhw, spc = vc - nbc * dhw;
    // this lane's 16-byte group of the voxel's record in chunk plane 0; planes are dhw records apart
    const uint4* xrow = src + ((size_t)nbc * nchunks * dhw + spc) * 2 + half;
    for (int c = 0; c < nchunks; ++c) {
        const uint4 xf = xrow[(size_t)c * dhw * 2];
        const uint4* wp = wts + (((size_t)c * 8) * ntiles + ntile) * 64 + lane;
#pragma unroll
        for (int ph = 0; ph < 8; ++ph) mma_ct<T>(acc[ph], wp[(size_t)ph * ntiles * 64], xf);
    }
    if (v >= nvox) return;
    const int nb = nbc, sp = spc;
    const int z = sp / hw, y = (sp - z * hw) / w, x = sp - z * hw - y * w;
    const int oh = 2 * h, ow = 2 * w;
    const size_t odhw = (size_t)8 * dhw;
    char* const oplane = static_cast<char*>(dst) + ((size_t)nb * (cout / KC) + ntile * (32 / KC)) * odhw * 32;
#pragma unroll
    for (int ph = 0; ph < 8; ++ph) {
        const size_t ovox = ((size_t)(2 * z + (ph >> 2)) * oh + 2 * y + ((ph >> 1) & 1)) * ow +
                            2 * x + (ph & 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float o4[4] = {acc[ph][4 * q], acc[ph][4 * q + 1], acc[ph][4 * q + 2], acc[ph][4 * q + 3]};
            const int byte = (8 * q + 4 * half) * ES;   // inside the 32-channel slice
            char* out = oplane + ((size_t)(byte / 32) * odhw + ovox) * 32 + byte % 32;
            if (G == 4) {
                *reinterpret_cast<float4*>(out) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            } else {
                float o8[8] = {o4[0], o4[1], o4[2], o4[3], 0.f, 0.f, 0.f, 0.f};
                const uint4 pk = T::pack(o8);
                *reinterpret_cast<uint2*>(out) = make_uint2(pk.x, pk.y);
            }
        }
    }
}

// ---- inc.0 for the 16-bit modes ------------------------------------------------------
// The fp32 MFMA above shares the vector ALUs and takes 14 x 16 passes per 32 voxels.
// With 16-bit storage the layer's result is rounded to 8 or 11 significant bits
// anyway, so the contraction runs on the 16-bit matrix pipe with split operands:
// x = x_hi + x_lo and w = w_hi + w_lo (each half a bf16 / f16 number), and
// x w ~ x_hi w_hi + x_lo w_hi + x_hi w_lo in fp32 accumulation (the dropped terms are
// below 2^-16 |x w|, still far under the output rounding). K = 27 taps padded to 32 =
// two MFMA steps; a lane holds taps 8 * (half + 2 * step) .. + 7 of its voxel.
// LeakyReLU with 0 <= slope <= 1 is max(v, slope * v): a packed multiply and one bare
// v_max_f32 per value (fmaxf would add a canonicalising v_max in front of each)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t leaky2(f32x2_t v, float slope) {
    const f32x2_t sv = v * slope;
    f32x2_t r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r.x) : "v"(v.x), "v"(sv.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(r.y) : "v"(v.y), "v"(sv.y));
    return r;
}

template <typename T>
struct Half16;
template <>
struct Half16<BF16T> {
    static __device__ __forceinline__ unsigned short bits(float v) {
        return __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    static __device__ __forceinline__ float value(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
};
template <>
struct Half16<F16T> {
    static __device__ __forceinline__ unsigned short bits(float v) {
        return __builtin_bit_cast(unsigned short, (_Float16)v);
    }
    static __device__ __forceinline__ float value(unsigned short b) {
        return (float)__builtin_bit_cast(_Float16, b);
    }
};

// splits eight floats into packed 16-bit high parts and packed 16-bit remainders
template <typename T>
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned short h0 = Half16<T>::bits(v[2 * i]), h1 = Half16<T>::bits(v[2 * i + 1]);
        const unsigned short l0 = Half16<T>::bits(v[2 * i] - Half16<T>::value(h0));
        const unsigned short l1 = Half16<T>::bits(v[2 * i + 1] - Half16<T>::value(h1));
        h[i] = (unsigned)h0 | ((unsigned)h1 << 16);
        l[i] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// (n, d, h, w) float32 -> (n, d+2, h+2, w+2) with a zero border, every voxel already split
// into its two 16-bit parts, hi | lo << 16: a voxel is read by 27 taps of up to four
// 32-cout slices, so the split (two conversions and a subtraction) is done here once per
// voxel instead of once per use. One block per padded plane.
template <typename T>
__global__ __launch_bounds__(256) void pad_split_kernel(const float* __restrict__ x,
                                                        unsigned* __restrict__ xp, int d, int h, int w) {
    const int nb = blockIdx.x / (d + 2), pz = blockIdx.x - nb * (d + 2);
    const int z = pz - 1;
    const int pw = w + 2, plane = (h + 2) * pw;
    unsigned* const dst = xp + (size_t)blockIdx.x * plane;
    if ((unsigned)z >= (unsigned)d) {
        for (int i = threadIdx.x; i < plane; i += blockDim.x) dst[i] = 0u;
        return;
    }
    const float* const src = x + ((size_t)nb * d + z) * h * w;
    for (int i = threadIdx.x; i < plane; i += blockDim.x) {
        const int py = i / pw, px = i - py * pw;
        const int y = py - 1, xx = px - 1;
        unsigned v = 0u;
        if ((unsigned)y < (unsigned)h && (unsigned)xx < (unsigned)w) {
            const float f = src[y * w + xx];
            const unsigned short hi = Half16<T>::bits(f);
            const unsigned short lo = Half16<T>::bits(f - Half16<T>::value(hi));
            v = (unsigned)hi | ((unsigned)lo << 16);
        }
        dst[i] = v;
    }
}

#ifndef EXASPIM_FIRST_HIDDEN_STORES
#define EXASPIM_FIRST_HIDDEN_STORES 0   // inc.0: 1 = stores the compiler's wait counts do not see (measurement aid)
#endif
#ifndef EXASPIM_ABLATE_FIRST
#define EXASPIM_ABLATE_FIRST 0   // tools/layer_bench only (results wrong on purpose): 1 = no stores, 2 = no LDS
#endif                           // transposition, 4 = one MFMA instead of six, 8 = no tap loads
// raw buffer access: 32-bit offsets against a descriptor, wave-uniform part in the scalar offset
typedef unsigned int u32x4_b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t layer_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// (no minimum-occupancy hint: asking for 4 / 6 / 8 waves per SIMD measured 365 / 1362 / 1537 us
// against 302 us -- the two tap register sets want their ~100 registers)
// ROWS: the patch width is a multiple of 32, so a 32-voxel group is a piece of one row: its
// position is decoded once per wave with scalar arithmetic, every tap load is a buffer load with a
// lane-constant offset (tap offset + lane's x) plus the group's scalar offset, and every store a
// buffer store likewise -- no per-group vector address arithmetic at all (r03; before, 16 64-bit
// address pairs were rebuilt per group: 148 registers, three waves per SIMD).
// The next group's 16 tap loads are issued UNCONDITIONALLY (the group index is clamped, the last
// prefetch of a wave is simply not used): behind a branch hipcc's wait-count insertion assumed the
// not-taken path at the join and made every group wait for its successor's loads as well
// (s_waitcnt vmcnt(0) in front of the matrix instructions) -- the double buffering never overlapped.
template <typename T, bool ROWS>
__global__ __launch_bounds__(256) void conv_first16_kernel(
    const unsigned* __restrict__ xp, const float* __restrict__ w,
    const float* __restrict__ bias, void* __restrict__ dst, int nvox, int d, int h, int wd,
    int c0p, float slope) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, r = lane & 31;
    const int co_tile = blockIdx.y * 32;
    const int hw = h * wd, dhw = d * hw;
    const int pw = wd + 2, phw = (h + 2) * pw;   // padded row / plane strides

    // A operands (weights of output channel co_tile + r) and the taps' offsets in the padded patch
    uint4 whi[2], wlo[2];
    int rel[2][8];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 8 * (half + 2 * st) + j;
            const bool real = t < 27;
            const int tt = real ? t : 0;
            wv[j] = real ? w[tt * c0p + co_tile + r] : 0.f;   // zero weight for the padding taps
            rel[st][j] = (tt / 9) * phw + ((tt / 3) % 3) * pw + tt % 3;
            if (ROWS) rel[st][j] = (rel[st][j] + r) * 4;      // byte offset incl. the lane's x
        }
        split8<T>(wv, whi[st], wlo[st]);
    }
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        bq[q] = *reinterpret_cast<const float4*>(bias + co_tile + 8 * q + 4 * half);

    constexpr int G = T::kG;
    constexpr int ES = 16 / G;
    constexpr int RECB = 32 * ES;
    constexpr int RECP = RECB + 16;
    constexpr int CPT = RECB / 32;
    __shared__ __attribute__((aligned(16))) char tr[4 * 32 * RECP];
    char* wl = tr + wave * (32 * RECP);

    // a wave walks 32-voxel groups: the weight operands above are set up once, and the 16
    // tap loads of the next group are in flight while the current one goes through the
    // matrix pipe, LDS and the stores (a group alone is a chain of latencies)
    const int ngroups = (nvox + 31) / 32;
    struct Pos {
        const unsigned* base;   // !ROWS: padded address of tap (0,0,0) = voxel (zz-1, yy-1, xx-1)
        int v, nb, sp;          // !ROWS: per lane; ROWS: nb and sp (first voxel of the group) are scalar
        unsigned soff;          // ROWS: byte offset of the group's tap (0,0,0) inside its padded patch
    };
    const size_t xpatch_bytes = (size_t)(d + 2) * phw * 4;               // one padded patch
    const size_t opatch_bytes = (size_t)(c0p * ES / 32) * dhw * 32;      // one output patch, all chunk planes
    auto locate = [&](int grp) {
        Pos p;
        if (ROWS) {
            const int g0 = __builtin_amdgcn_readfirstlane(grp) * 32;   // first voxel of the group
            p.nb = g0 / dhw;
            p.sp = g0 - p.nb * dhw;
            const int zz = p.sp / hw, yy = (p.sp - zz * hw) / wd, xx = p.sp - zz * hw - yy * wd;
            p.soff = (unsigned)((zz * phw + yy * pw + xx) * 4);
            p.v = g0;
            p.base = nullptr;
        } else {
            p.v = grp * 32 + r;
            const int vc = p.v < nvox ? p.v : nvox - 1;
            p.nb = vc / dhw; p.sp = vc - p.nb * dhw;
            const int zz = p.sp / hw, yy = (p.sp - zz * hw) / wd, xx = p.sp - zz * hw - yy * wd;
            p.base = xp + ((size_t)p.nb * (d + 2) + zz) * phw + yy * pw + xx;
            p.soff = 0;
        }
        return p;
    };
    const int gstep = gridDim.x * 4;
    int grp = blockIdx.x * 4 + wave;
    if (grp >= ngroups) return;
    auto load_taps = [&](const Pos& p, unsigned (*x)[8]) {   // hi | lo << 16 per tap (pad_split_kernel)
        if (ROWS) {
            const __amdgpu_buffer_rsrc_t rs =
                layer_rsrc(reinterpret_cast<const char*>(xp) + (size_t)p.nb * xpatch_bytes, xpatch_bytes);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[st][j] = (EXASPIM_ABLATE_FIRST & 8) ? (unsigned)(rel[st][j] + p.soff)
                                                          : __builtin_amdgcn_raw_buffer_load_b32(rs, rel[st][j], (int)p.soff, 0);
        } else {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[st][j] = p.base[rel[st][j]];
        }
    };
    const int vv = lane >> 1, sub = lane & 1;
    auto process = [&](const Pos& cur, int g, const unsigned (*x)[8]) {
        f32x16_ct acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[4 * q + 0] = bq[q].x; acc[4 * q + 1] = bq[q].y;
            acc[4 * q + 2] = bq[q].z; acc[4 * q + 3] = bq[q].w;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            // v_perm_b32: gather the low halves (hi parts) resp. the high halves (lo parts) of two taps
            uint4 xhi, xlo;
            xhi.x = __builtin_amdgcn_perm(x[st][1], x[st][0], 0x05040100u); xlo.x = __builtin_amdgcn_perm(x[st][1], x[st][0], 0x07060302u);
            xhi.y = __builtin_amdgcn_perm(x[st][3], x[st][2], 0x05040100u); xlo.y = __builtin_amdgcn_perm(x[st][3], x[st][2], 0x07060302u);
            xhi.z = __builtin_amdgcn_perm(x[st][5], x[st][4], 0x05040100u); xlo.z = __builtin_amdgcn_perm(x[st][5], x[st][4], 0x07060302u);
            xhi.w = __builtin_amdgcn_perm(x[st][7], x[st][6], 0x05040100u); xlo.w = __builtin_amdgcn_perm(x[st][7], x[st][6], 0x07060302u);
            mma_ct<T>(acc, whi[st], xhi);
            if ((EXASPIM_ABLATE_FIRST & 4) && st == 0) { acc[1] += __uint_as_float(xlo.x ^ xhi.y); continue; }
            if (EXASPIM_ABLATE_FIRST & 4) { acc[2] += __uint_as_float(xlo.z ^ xhi.w); continue; }
            mma_ct<T>(acc, whi[st], xlo);
            mma_ct<T>(acc, wlo[st], xhi);
        }
        // LeakyReLU, then through LDS so that each store instruction writes the 32 voxel
        // records of one chunk plane
        uint2 direct[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = 8 * q + 4 * half;
            const f32x2_t a0 = leaky2((f32x2_t){acc[4 * q], acc[4 * q + 1]}, slope);
            const f32x2_t a1 = leaky2((f32x2_t){acc[4 * q + 2], acc[4 * q + 3]}, slope);
            const float o8[8] = {a0.x, a0.y, a1.x, a1.y, 0.f, 0.f, 0.f, 0.f};
            const uint4 pk = T::pack(o8);
            direct[q] = make_uint2(pk.x, pk.y);
            if (!(EXASPIM_ABLATE_FIRST & 2)) *reinterpret_cast<uint2*>(wl + r * RECP + cl * ES) = direct[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ROWS && (EXASPIM_ABLATE_FIRST & 3)) {
            const __amdgpu_buffer_rsrc_t rs =
                layer_rsrc(static_cast<char*>(dst) + (size_t)cur.nb * opatch_bytes, opatch_bytes);
#pragma unroll
            for (int ck = 0; ck < CPT; ++ck) {
                uint4 val = (EXASPIM_ABLATE_FIRST & 2)
                                ? make_uint4(direct[2 * ck].x, direct[2 * ck].y, direct[2 * ck + 1].x, direct[2 * ck + 1].y)
                                : *reinterpret_cast<const uint4*>(wl + vv * RECP + (ck * 2 + sub) * 16);
                if (!(EXASPIM_ABLATE_FIRST & 1) || val.x == 0x12345678u)
                    buf_store16(val, rs, vv * 32 + sub * 16, ((unsigned)(blockIdx.y * CPT + ck) * (unsigned)dhw + (unsigned)cur.sp) * 32u);
            }
        } else if (ROWS) {
            // one store instruction = the group's 32 voxel records of one chunk plane, 1 KiB contiguous
            const __amdgpu_buffer_rsrc_t rs =
                layer_rsrc(static_cast<char*>(dst) + (size_t)cur.nb * opatch_bytes, opatch_bytes);
#pragma unroll
            for (int ck = 0; ck < CPT; ++ck) {
                const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + (ck * 2 + sub) * 16);
#if EXASPIM_FIRST_HIDDEN_STORES
                buf_store16(val, rs, vv * 32 + sub * 16, ((unsigned)(blockIdx.y * CPT + ck) * (unsigned)dhw + (unsigned)cur.sp) * 32u);
#else
                // counted: the wait for the next group's taps (issued before these stores) leaves them in flight
                buf_store16_counted(val, rs, vv * 32 + sub * 16, ((unsigned)(blockIdx.y * CPT + ck) * (unsigned)dhw + (unsigned)cur.sp) * 32u);
#endif
            }
        } else {
            const int vo = g * 32 + vv;
            const int voc = vo < nvox ? vo : nvox - 1;
            const int nbo = voc / dhw, spo = voc - nbo * dhw;
            char* const dplane = static_cast<char*>(dst) +
                                 ((size_t)nbo * (c0p * ES / 32) + blockIdx.y * CPT) * dhw * 32;
#pragma unroll
            for (int ck = 0; ck < CPT; ++ck) {
                const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + (ck * 2 + sub) * 16);
                if (vo < nvox)
                    *reinterpret_cast<uint4*>(dplane + ((size_t)ck * dhw + spo) * 32 + sub * 16) = val;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // two tap buffers in turn: the loads of group i + 1 are issued before group i is processed
    unsigned xa[2][8], xb[2][8];
    Pos pa = locate(grp), pb = pa;
    load_taps(pa, xa);
    for (;;) {
        const int g1 = grp + gstep;
        pb = locate(g1 < ngroups ? g1 : ngroups - 1);
        load_taps(pb, xb);
        process(pa, grp, xa);
        if (g1 >= ngroups) break;
        const int g2 = g1 + gstep;
        pa = locate(g2 < ngroups ? g2 : ngroups - 1);
        load_taps(pa, xa);
        process(pb, g1, xb);
        if (g2 >= ngroups) break;
        grp = g2;
    }
}

// inc.0 on row strips (r03): a workgroup of four waves owns 4 rows x the whole width of one plane of a
// patch and keeps the 3 planes x 6 rows x (width + 2) input words its 27 taps touch in LDS. The ablations
// of conv_first16_kernel say its 16 tap loads per 32-voxel group and its stores each
// take their own time and do not overlap; here a thread issues 7 global loads per STRIP (12 groups at a
// width of 96) instead of 48, for the next strip and before the current strip's stores, the taps are
// ds_read_b32 (consecutive lanes, consecutive words: no conflicts), and the only wait for global data sits
// behind the strip's stores, which hipcc counts (s_waitcnt vmcnt(#stores)). Same operands to the same six
// MFMAs per group as conv_first16_kernel: same bits.
constexpr int kStripRows = 4;
template <typename T, int GPW>   // GPW: 32-voxel groups per wave and strip = width / 32
__global__ __launch_bounds__(256) void conv_first16_strip_kernel(
    const unsigned* __restrict__ xp, const float* __restrict__ w, const float* __restrict__ bias,
    void* __restrict__ dst, int n, int d, int h, int wd, int c0p, float slope) {
    constexpr int G = T::kG;
    constexpr int ES = 16 / G;
    constexpr int RECB = 32 * ES;
    constexpr int RECP = RECB + 16;
    constexpr int CPT = RECB / 32;
    constexpr int R = kStripRows;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, r = lane & 31;
    const int co_tile = blockIdx.y * 32;
    const int hw = h * wd, dhw = d * hw;
    const int pw = wd + 2, phw = (h + 2) * pw;    // padded row / plane strides (global)
    const int trow = pw, tplane = (R + 2) * pw;   // row / plane strides of the LDS tile
    const int twords = 3 * tplane;                // words of one input tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* const tile0 = reinterpret_cast<unsigned*>(smem);
    char* const tr = smem + 2 * ((twords * 4 + 15) / 16 * 16);
    char* const wl = tr + wave * (32 * RECP);

    // A operands and the taps' offsets inside the LDS tile (words)
    uint4 whi[2], wlo[2];
    int rel[2][8];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 8 * (half + 2 * st) + j;
            const bool real = t < 27;
            const int tt = real ? t : 0;
            wv[j] = real ? w[tt * c0p + co_tile + r] : 0.f;
            rel[st][j] = (tt / 9) * tplane + ((tt / 3) % 3) * trow + tt % 3 + r;
        }
        split8<T>(wv, whi[st], wlo[st]);
    }
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        bq[q] = *reinterpret_cast<const float4*>(bias + co_tile + 8 * q + 4 * half);

    // this thread's words of a tile: word i = tid + k * 256 -> (plane, row, x) of the tile
    constexpr int NLOAD = (3 * (R + 2) * (GPW * 32 + 2) + 255) / 256;   // words of a tile per thread (7 at a width of 96)
    unsigned goff[NLOAD];                          // byte offset inside the padded patch relative to the strip origin
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
        const int i = threadIdx.x + k * 256;
        const int pl = i / tplane, rem = i - pl * tplane;
        const int ry = rem / trow, x = rem - ry * trow;
        goff[k] = i < twords ? (unsigned)((pl * phw + ry * pw + x) * 4) : 0x80000000u;
    }
    const size_t xpatch_bytes = (size_t)(d + 2) * phw * 4;
    const size_t opatch_bytes = (size_t)(c0p * ES / 32) * dhw * 32;
    const int strips_y = h / R;
    const int nstrips = n * d * strips_y;
    const int gpr = wd / 32;                       // groups per row (= GPW: four rows, four waves)

    struct Strip { int nb, z, y0; };
    auto locate = [&](int sidx) {
        Strip t;
        const int si = __builtin_amdgcn_readfirstlane(sidx);
        t.nb = si / (d * strips_y);
        const int rem = si - t.nb * (d * strips_y);
        t.z = rem / strips_y;
        t.y0 = (rem - t.z * strips_y) * R;
        return t;
    };
    unsigned stage[NLOAD];
    auto fetch = [&](const Strip& t) {             // padded coordinates: tap (0,0,0) of voxel (z, y0, 0) is (z, y0, 0)
        const __amdgpu_buffer_rsrc_t rs =
            layer_rsrc(reinterpret_cast<const char*>(xp) + (size_t)t.nb * xpatch_bytes, xpatch_bytes);
        const unsigned so = (unsigned)((t.z * phw + t.y0 * pw) * 4);
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) stage[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)goff[k], (int)so, 0);
    };
    auto park = [&](unsigned* tile) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < twords) tile[i] = stage[k];
        }
    };

    int sidx = blockIdx.x;
    if (sidx >= nstrips) return;
    Strip cur = locate(sidx);
    fetch(cur);
    park(tile0);
    __syncthreads();
    const int vv = lane >> 1, sub = lane & 1;
    int buf = 0;
    for (;;) {
        const int snext = sidx + gridDim.x;
        const bool more = snext < nstrips;
        const Strip nxt = locate(more ? snext : sidx);
        fetch(nxt);                                // next strip's words, BEFORE this strip's stores
        const unsigned* const tile = tile0 + buf * ((twords + 3) / 4 * 4);
        const __amdgpu_buffer_rsrc_t ors =
            layer_rsrc(static_cast<char*>(dst) + (size_t)cur.nb * opatch_bytes, opatch_bytes);
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {         // (a compile-time count: hipcc must be able to count the stores)
            const int q = wave + 4 * gi;
            const int ry = q / gpr, xs = (q - ry * gpr) * 32;
            const int base = ry * trow + xs;
            f32x16_ct acc;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                acc[4 * qq + 0] = bq[qq].x; acc[4 * qq + 1] = bq[qq].y;
                acc[4 * qq + 2] = bq[qq].z; acc[4 * qq + 3] = bq[qq].w;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                unsigned x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = tile[base + rel[st][j]];
                uint4 xhi, xlo;
                xhi.x = __builtin_amdgcn_perm(x[1], x[0], 0x05040100u); xlo.x = __builtin_amdgcn_perm(x[1], x[0], 0x07060302u);
                xhi.y = __builtin_amdgcn_perm(x[3], x[2], 0x05040100u); xlo.y = __builtin_amdgcn_perm(x[3], x[2], 0x07060302u);
                xhi.z = __builtin_amdgcn_perm(x[5], x[4], 0x05040100u); xlo.z = __builtin_amdgcn_perm(x[5], x[4], 0x07060302u);
                xhi.w = __builtin_amdgcn_perm(x[7], x[6], 0x05040100u); xlo.w = __builtin_amdgcn_perm(x[7], x[6], 0x07060302u);
                mma_ct<T>(acc, whi[st], xhi);
                mma_ct<T>(acc, whi[st], xlo);
                mma_ct<T>(acc, wlo[st], xhi);
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int cl = 8 * qq + 4 * half;
                const f32x2_t a0 = leaky2((f32x2_t){acc[4 * qq], acc[4 * qq + 1]}, slope);
                const f32x2_t a1 = leaky2((f32x2_t){acc[4 * qq + 2], acc[4 * qq + 3]}, slope);
                const float o8[8] = {a0.x, a0.y, a1.x, a1.y, 0.f, 0.f, 0.f, 0.f};
                const uint4 pk = T::pack(o8);
                *reinterpret_cast<uint2*>(wl + r * RECP + cl * ES) = make_uint2(pk.x, pk.y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned sp = (unsigned)((cur.z * h + cur.y0 + ry) * wd + xs);   // first voxel of the group inside the patch
#pragma unroll
            for (int ck = 0; ck < CPT; ++ck) {
                const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + (ck * 2 + sub) * 16);
                buf_store16_counted(val, ors, vv * 32 + sub * 16,
                                    ((unsigned)(blockIdx.y * CPT + ck) * (unsigned)dhw + sp) * 32u);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (!more) break;
        park(tile0 + (buf ^ 1) * ((twords + 3) / 4 * 4));   // (waits for the fetched words: they are older than the stores)
        __syncthreads();
        buf ^= 1;
        sidx = snext;
        cur = nxt;
    }
}

// ---- max-pool 2x2x2 ----------------------------------------------------------
// Grid: x = (volume, output plane), y = blocks of 16-byte pieces of that plane, so
// small pyramid levels still fill their blocks. Index arithmetic is 32-bit; the
// volume/plane split is scalar.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const uint4* __restrict__ src,
                                                       uint4* __restrict__ dst, int d, int h,
                                                       int w, int cg) {
    // d,h,w: input size; cg: 16-byte groups per voxel record
    const int od = d / 2, oh = h / 2, ow = w / 2;
    const int i = blockIdx.y * blockDim.x + threadIdx.x;  // piece inside the output plane
    if (i >= oh * ow * cg) return;
    const int y = i / (ow * cg), ix = i - y * (ow * cg);
    const int x = ix / cg, g = ix - x * cg;
    const int nb = blockIdx.x / od, z = blockIdx.x - nb * od;
    float m[T::kG];
#pragma unroll
    for (int j = 0; j < T::kG; ++j) m[j] = -INFINITY;
    const size_t plane = (size_t)h * w;
    const uint4* base = src + (((size_t)nb * d + 2 * z) * plane + (size_t)(2 * y) * w + 2 * x) * cg + g;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float f[T::kG];
                T::unpack(base[(dz * plane + (size_t)dy * w + dx) * cg], f);
#pragma unroll
                for (int j = 0; j < T::kG; ++j) m[j] = fmaxf(m[j], f[j]);
            }
    dst[((((size_t)nb * od + z) * oh + y) * ow + x) * cg + g] = T::pack(m);
}

// ---- trilinear x2, align_corners=True ---------------------------------------
// torch (ATen UpSample.h): scale = (in - 1) / (out - 1) in float; src = scale *
// dst_index; i0 = floor(src) clamped; lambda = src - i0 clamped to [0, 1];
// i1 = min(i0 + 1, in - 1). The three scales are computed on the host with the
// same float division.
__device__ __forceinline__ void lerp_coord(int o, int in, float scale, int& i0, int& i1, float& l1) {
    // rounded product, then a rounded difference, as ATen computes them (no contraction
    // into one fma: see the note on -ffp-contract=off at the top of the file)
    const float s = scale * (float)o;
    i0 = min((int)floorf(s), in - 1);
    i1 = min(i0 + 1, in - 1);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}

// A thread owns a pair of output rows at one x (one 16-byte channel group) and walks a
// run of output planes, two at a time; lanes run along (x, group), so a store instruction
// writes 1 KiB of consecutive voxel records. Along z and y two consecutive outputs read the
// sources i0, i0 + 1 and, for the second one, possibly i0 + 2 (the source coordinate grows
// by scale < 1/2 per output). The interpolation runs separably in torch's own nesting,
// d0 * (h0 * (w0 v000 + w1 v001) + h1 * (...)) + d1 * (...): a source plane is interpolated
// along x and y once (3 rows x 2 columns of 16-byte pieces, each converted to float once)
// into the thread's two output rows and kept in registers for the 2-4 output planes that
// blend it -- the last three source planes are live, a pair of output planes needs about
// one new one. Per axis the first output of a pair is fma(w1, v[1], w0 * v[0]); the second
// one takes (v[0], v[1]) or (v[1], v[2]) depending on whether its i0 moved on: it is
// evaluated as fma(c2, v[2], fma(c1, v[1], c0 * v[0])) with (c0, c1, c2) = (w0, w1, 0) or
// (0, w0, w1) -- the zero term adds nothing, so the value is the two-term expression bit for
// bit and no lane ever selects between registers. Sources are taken at min(i0 + k, n - 1),
// which is where torch's clamped i1 points when i0 is the last element.
// Grid: x = (volume, run of kUpsZRun pairs of output planes), y = blocks of (row pair, x, group).
constexpr int kUpsZRun = 12;
#ifndef UPS_MINW
#define UPS_MINW 3   // waves per SIMD the register allocation aims at (tools/layer_bench.hip)
#endif

struct LerpPair {
    int s0, s1, s2;        // source indices i0, min(i0 + 1, n - 1), min(i0 + 2, n - 1) of the first output
    float a0, a1;          // first output: a0 * v[0] + a1 * v[1]
    float b0, b1, b2;      // second output (see above)
};

__device__ __forceinline__ LerpPair lerp_pair(int o, bool has_second, int in, float scale) {
    int i0, i1, j0, j1;
    float la, lb;
    lerp_coord(o, in, scale, i0, i1, la);
    lerp_coord(has_second ? o + 1 : o, in, scale, j0, j1, lb);
    LerpPair p;
    p.s0 = i0; p.s1 = min(i0 + 1, in - 1); p.s2 = min(i0 + 2, in - 1);
    p.a0 = 1.f - la; p.a1 = la;
    const bool moved = j0 != i0;   // j0 is i0 or i0 + 1
    p.b0 = moved ? 0.f : 1.f - lb;
    p.b1 = moved ? 1.f - lb : lb;
    p.b2 = moved ? lb : 0.f;
    return p;
}

template <typename T>
__global__ __launch_bounds__(256, UPS_MINW) void upsample2_kernel(const uint4* __restrict__ src,
                                                           uint4* __restrict__ dst, int d, int h,
                                                           int w, float sz, float sy, float sx,
                                                           int margin) {
    // output voxels closer than "margin" to a face are not needed by the caller
    constexpr int cg = 2;       // 16-byte groups of a 32-byte record
    constexpr int NP = T::kG / 2;   // float pairs of a group
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int od = d * 2, oh = h * 2, ow = w * 2;
    const int nz = od - 2 * margin, ny = oh - 2 * margin, nx = ow - 2 * margin;
    const int nzp = (nz + 1) >> 1, nyp = (ny + 1) >> 1;
    const int nruns = (nzp + kUpsZRun - 1) / kUpsZRun;
    const unsigned item = blockIdx.y * blockDim.x + threadIdx.x;
    if (item >= (unsigned)(nyp * nx * cg)) return;
    const int g = item & 1;
    const int yp = (int)(item >> 1) / nx, x = margin + (int)(item >> 1) - yp * nx;
    const int nb = blockIdx.x / nruns, run = blockIdx.x - nb * nruns;
    const int ya = margin + 2 * yp;
    const bool has_yb = ya + 1 < oh - margin;
    const LerpPair py = lerp_pair(ya, has_yb, h, sy);
    int x0, x1;
    float lx;
    lerp_coord(x, w, sx, x0, x1, lx);
    const f2 wx0 = {1.f - lx, 1.f - lx}, wx1 = {lx, lx};
    const uint4* base = src + (size_t)nb * d * h * w * cg + g;
    const int pos[3][2] = {{(py.s0 * w + x0) * cg, (py.s0 * w + x1) * cg},
                           {(py.s1 * w + x0) * cg, (py.s1 * w + x1) * cg},
                           {(py.s2 * w + x0) * cg, (py.s2 * w + x1) * cg}};
    const float yA[3] = {py.a0, py.a1, 0.f}, yB[3] = {py.b0, py.b1, py.b2};

    // source plane -> this thread's two output rows (x and y interpolation)
    auto load_plane = [&](int p, f2 (*q)[NP]) {
        const uint4* plane = base + (size_t)p * h * w * cg;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float v0[T::kG], v1[T::kG];
            T::unpack(plane[pos[r][0]], v0);
            T::unpack(plane[pos[r][1]], v1);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f2 xr = __builtin_elementwise_fma(wx1, (f2){v1[2 * j], v1[2 * j + 1]},
                                                        wx0 * (f2){v0[2 * j], v0[2 * j + 1]});
                const f2 wa = {yA[r], yA[r]}, wb = {yB[r], yB[r]};
                if (r == 0) {
                    q[0][j] = wa * xr;
                    q[1][j] = wb * xr;
                } else {
                    if (r == 1) q[0][j] = __builtin_elementwise_fma(wa, xr, q[0][j]);
                    q[1][j] = __builtin_elementwise_fma(wb, xr, q[1][j]);
                }
            }
        }
    };
    auto store = [&](int z, const f2 (*o)[NP]) {
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            if (yy == 1 && !has_yb) continue;
            float out[T::kG];
#pragma unroll
            for (int j = 0; j < NP; ++j) { out[2 * j] = o[yy][j].x; out[2 * j + 1] = o[yy][j].y; }
            dst[((((size_t)nb * od + z) * oh + ya + yy) * ow + x) * cg + g] = T::pack(out);
        }
    };

    // the last three interpolated source planes, q2 the newest (plane "top")
    f2 q0[2][NP], q1[2][NP], q2[2][NP];
#pragma unroll
    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
        for (int j = 0; j < NP; ++j) q0[yy][j] = q1[yy][j] = q2[yy][j] = (f2){0.f, 0.f};
    int top = -1;
    const int pair_end = min(nzp, (run + 1) * kUpsZRun);
    for (int pr = run * kUpsZRun; pr < pair_end; ++pr) {     // wave-uniform
        const int za = margin + 2 * pr;
        const bool has_zb = za + 1 < od - margin;
        const LerpPair pz = lerp_pair(za, has_zb, d, sz);
        const int need[3] = {pz.s0, pz.s1, pz.s2};           // non-decreasing, steps of 0 or 1
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (need[k] > top) {
#pragma unroll
                for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                    for (int j = 0; j < NP; ++j) { q0[yy][j] = q1[yy][j]; q1[yy][j] = q2[yy][j]; }
                load_plane(need[k], q2);
                top = need[k];
            }
        }
        // top == s2; plane s0 sits (top - s0) slots back, s1 one in front of it (clamped to q2)
        f2 oa[2][NP], ob[2][NP];
        auto blend = [&](const f2 (*v0)[NP], const f2 (*v1)[NP], const f2 (*v2)[NP]) {
#pragma unroll
            for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    oa[yy][j] = __builtin_elementwise_fma((f2){pz.a1, pz.a1}, v1[yy][j],
                                                          (f2){pz.a0, pz.a0} * v0[yy][j]);
                    ob[yy][j] = __builtin_elementwise_fma(
                        (f2){pz.b2, pz.b2}, v2[yy][j],
                        __builtin_elementwise_fma((f2){pz.b1, pz.b1}, v1[yy][j],
                                                  (f2){pz.b0, pz.b0} * v0[yy][j]));
                }
        };
        const int back = top - pz.s0;
        if (back == 2) blend(q0, q1, q2);
        else if (back == 1) blend(q1, q2, q2);
        else blend(q2, q2, q2);
        store(za, oa);
        if (has_zb) store(za + 1, ob);
    }
}

// The same interpolation, software-pipelined (r03). In upsample2_kernel a pair of output planes first
// loads the source plane it is missing and uses it at once, and because vector memory operations retire
// in order that wait also covers the four stores of the pair before: every pair pays a load round trip
// plus a store round trip, three waves per SIMD cannot hide it, and the kernel sat at 2.5-2.9 TB/s with
// its arithmetic-only and store-only halves each at ~175 us. For x2 with align_corners the
// schedule is regular whenever the first output plane is even (margin even): pair p (outputs 2p, 2p + 1)
// reads source planes p - 1, p, p + 1 (clamped), i.e. exactly ONE new plane per pair from p = 2 on. So
// the loop body is branch-free: the new plane's six 16-byte pieces are fetched one pair AHEAD, before the
// current pair's stores are issued, every store is issued unconditionally (range-checked buffer stores:
// a row or plane outside the region gets an out-of-range offset / a zero-record descriptor), and hipcc's
// wait for the prefetched pieces is s_waitcnt vmcnt(4): the four stores stay in flight. Same arithmetic
// in the same order as upsample2_kernel, bit for bit (test_pipelined_upsampling_equals_the_plain_kernel).
template <typename T>
__global__ __launch_bounds__(256, UPS_MINW) void upsample2_pipe_kernel(const uint4* __restrict__ src,
                                                                uint4* __restrict__ dst, int d, int h,
                                                                int w, float sz, float sy, float sx,
                                                                int margin) {
    constexpr int cg = 2;
    constexpr int NP = T::kG / 2;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int od = d * 2, oh = h * 2, ow = w * 2;
    const int nz = od - 2 * margin, ny = oh - 2 * margin, nx = ow - 2 * margin;
    const int nzp = (nz + 1) >> 1, nyp = (ny + 1) >> 1;
    const int nruns = (nzp + kUpsZRun - 1) / kUpsZRun;
    const unsigned item = blockIdx.y * blockDim.x + threadIdx.x;
    // (a thread beyond the last item works on the last one and stores nothing)
    const bool live = item < (unsigned)(nyp * nx * cg);
    const unsigned it = live ? item : (unsigned)(nyp * nx * cg) - 1u;
    const int g = it & 1;
    const int yp = (int)(it >> 1) / nx, x = margin + (int)(it >> 1) - yp * nx;
    const int nb = blockIdx.x / nruns, run = blockIdx.x - nb * nruns;
    const int ya = margin + 2 * yp;
    const bool has_yb = ya + 1 < oh - margin;
    const LerpPair py = lerp_pair(ya, has_yb, h, sy);
    int x0, x1;
    float lx;
    lerp_coord(x, w, sx, x0, x1, lx);
    const f2 wx0 = {1.f - lx, 1.f - lx}, wx1 = {lx, lx};
    // byte offsets of the six pieces inside a source plane (lane constants), plane in the scalar offset
    const size_t splane = (size_t)h * w * cg * 16;
    const __amdgpu_buffer_rsrc_t srs = layer_rsrc(reinterpret_cast<const char*>(src) + (size_t)nb * d * splane,
                                                  (size_t)d * splane);
    const unsigned pos[3][2] = {{(unsigned)((py.s0 * w + x0) * cg + g) * 16u, (unsigned)((py.s0 * w + x1) * cg + g) * 16u},
                                {(unsigned)((py.s1 * w + x0) * cg + g) * 16u, (unsigned)((py.s1 * w + x1) * cg + g) * 16u},
                                {(unsigned)((py.s2 * w + x0) * cg + g) * 16u, (unsigned)((py.s2 * w + x1) * cg + g) * 16u}};
    const float yA[3] = {py.a0, py.a1, 0.f}, yB[3] = {py.b0, py.b1, py.b2};
    // output rows of this thread: byte offsets inside an output plane, out of range when not stored
    const size_t oplane = (size_t)oh * ow * cg * 16;
    char* const obase = reinterpret_cast<char*>(dst) + (size_t)nb * od * oplane;
    const unsigned orow[2] = {live ? (unsigned)(((ya + 0) * ow + x) * cg + g) * 16u : 0x80000000u,
                              live && has_yb ? (unsigned)(((ya + 1) * ow + x) * cg + g) * 16u : 0x80000000u};

    struct Raw { uint4 v[3][2]; };
    auto fetch = [&](int p) {                      // the six pieces of source plane p (clamped)
        Raw r;
        const unsigned so = (unsigned)(p < d - 1 ? p : d - 1) * (unsigned)splane;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u32x4_b t = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)pos[rr][c], (int)so, 0);
                r.v[rr][c] = make_uint4(t.x, t.y, t.z, t.w);
            }
        return r;
    };
    auto interp = [&](const Raw& r, f2 (*q)[NP]) {  // x and y interpolation, as upsample2_kernel's load_plane
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float v0[T::kG], v1[T::kG];
            T::unpack(r.v[rr][0], v0);
            T::unpack(r.v[rr][1], v1);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f2 xr = __builtin_elementwise_fma(wx1, (f2){v1[2 * j], v1[2 * j + 1]},
                                                        wx0 * (f2){v0[2 * j], v0[2 * j + 1]});
                const f2 wa = {yA[rr], yA[rr]}, wb = {yB[rr], yB[rr]};
                if (rr == 0) {
                    q[0][j] = wa * xr;
                    q[1][j] = wb * xr;
                } else {
                    if (rr == 1) q[0][j] = __builtin_elementwise_fma(wa, xr, q[0][j]);
                    q[1][j] = __builtin_elementwise_fma(wb, xr, q[1][j]);
                }
            }
        }
    };
    auto emit_pair = [&](int za, const LerpPair& pz, const f2 (*v0)[NP], const f2 (*v1)[NP], const f2 (*v2)[NP]) {
        const bool has_zb = za + 1 < od - margin;
#pragma unroll
        for (int zz = 0; zz < 2; ++zz) {
            // (zero-record descriptor for a plane that does not exist: a scalar select, no branch)
            const __amdgpu_buffer_rsrc_t ors = layer_rsrc(obase, zz == 0 || has_zb ? (size_t)od * oplane : (size_t)0);
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                float out[T::kG];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const f2 o = zz == 0
                        ? __builtin_elementwise_fma((f2){pz.a1, pz.a1}, v1[yy][j], (f2){pz.a0, pz.a0} * v0[yy][j])
                        : __builtin_elementwise_fma((f2){pz.b2, pz.b2}, v2[yy][j],
                              __builtin_elementwise_fma((f2){pz.b1, pz.b1}, v1[yy][j], (f2){pz.b0, pz.b0} * v0[yy][j]));
                    out[2 * j] = o.x; out[2 * j + 1] = o.y;
                }
                buf_store16_counted(T::pack(out), ors, orow[yy], (unsigned)(za + zz) * (unsigned)oplane);
            }
        }
    };

    f2 q0[2][NP], q1[2][NP], q2[2][NP];
    const int pair_end = min(nzp, (run + 1) * kUpsZRun);
    int pr = run * kUpsZRun;
    int p = (margin >> 1) + pr;                 // pair index in the un-trimmed output: planes 2p, 2p + 1
    // ---- prologue: pairs 0 and 1 read planes 0, 1, 2 and do not advance
    {
        const int first = p < 2 ? 0 : p - 2;
        interp(fetch(first), q0);
        interp(fetch(first + 1), q1);
        interp(fetch(first + 2), q2);
        for (; p < 2 && pr < pair_end; ++p, ++pr) {
            const int za = margin + 2 * pr;
            const LerpPair pz = lerp_pair(za, za + 1 < od - margin, d, sz);
            emit_pair(za, pz, q0, q1, q2);
        }
    }
    // ---- steady state: ring = planes (p - 2, p - 1, p) on entry; pair p wants (p - 1, p, p + 1).
    // "cur" has arrived when an iteration starts: the wait for the pieces fetched in an iteration sits
    // at its END, behind its four stores, where hipcc can count them (straight-line code: s_waitcnt
    // vmcnt(4)); at the loop top the entry edge and the back edge would disagree and it would have to
    // wait for everything. The empty asm is that wait's anchor (it "uses" every fetched register).
    auto settle = [](Raw& r) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                asm volatile("" : "+v"(r.v[rr][c].x), "+v"(r.v[rr][c].y), "+v"(r.v[rr][c].z), "+v"(r.v[rr][c].w));
    };
    Raw cur = fetch(p + 1);
    settle(cur);
    for (; pr < pair_end; ++pr, ++p) {          // wave-uniform
        const int za = margin + 2 * pr;
        const LerpPair pz = lerp_pair(za, za + 1 < od - margin, d, sz);
#pragma unroll
        for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int j = 0; j < NP; ++j) { q0[yy][j] = q1[yy][j]; q1[yy][j] = q2[yy][j]; }
        Raw nxt = fetch(p + 2);                 // one pair ahead, BEFORE this pair's stores
        interp(cur, q2);
        emit_pair(za, pz, q0, q1, q2);
        settle(nxt);
        cur = nxt;
    }
}

// The same interpolation with the SOURCE rows of a workgroup shared through LDS and a wave of its own to
// fetch them (r03). In the per-thread pipeline above every thread fetches its own six pieces of the new
// source plane -- 1 536 loads per workgroup and plane for at most 480 distinct pieces -- and waits for them
// in the wave that also stores (vector memory operations retire in order, one counter for loads and stores
// on gfx9: that wait is also a wait for every older store). Here a workgroup is five waves. Waves 0-3
// (256 threads = 256 consecutive (row pair, x, group) items = up to three row pairs) only read LDS,
// interpolate and store; they never wait on the vector memory counter. Wave 4 only loads: the contiguous
// run of source rows the workgroup shares is at most 512 pieces per plane (eight loads per lane),
// fetched kUpsAhead planes ahead into registers and parked in one of four LDS slots the step before it
// is read. One s_barrier per step keeps the two sides one plane apart (no vmcnt(0) at a workgroup barrier
// on gfx950). Step t: wave 4 parks plane p0 - 2 + t; waves 0-3 interpolate plane p0 - 3 + t (t >= 1) and,
// from t = 4 on, emit pair p0 + t - 4, whose newest plane that is. Needs the regular schedule, a first
// pair >= 2 (no pairs that do not advance) and a whole number of runs; same arithmetic in the same order
// as upsample2_kernel, bit for bit.
// Measured (level-0 launch of the trimmed forward, batch 16; tools/layer_bench.hip, tools/store_ceiling.hip,
// UPS_ABLATE): stand-alone 208-215 us against 247-261 us for the per-thread pipeline on the same box; the
// bare stores of this pattern take 147 us, compute + LDS 81 us, compute + loads 87 us, compute + stores
// 157-170 us. The same four numbers came out with the loads in the storing waves (prefetch 2 ... 8 planes
// deep: 222 ... 204 us), so what the loads cost on top of the stores is not the shared counter but the
// memory system serving reads between the writes; inside a step, where the source was written by the
// launch before, the kernel takes 170 us (rocprofv3) against 209 us for the per-thread pipeline.
#ifndef UPS_AHEAD
#define UPS_AHEAD 3
#endif
#ifndef UPS_ABLATE
#define UPS_ABLATE 0   // measurement aid: 1 = no global loads, 2 = every store dropped by its range check
#endif
constexpr int kUpsAhead = UPS_AHEAD;
constexpr int kUpsSlotPieces = 512;     // 16-byte pieces of one LDS plane slot (eight per lane of the loading wave)

template <typename T, int RUN>
__global__ __launch_bounds__(320) void upsample2_strip_kernel(const uint4* __restrict__ src,
                                                       uint4* __restrict__ dst, int d, int h,
                                                       int w, float sz, float sy, float sx,
                                                       int margin) {
    constexpr int cg = 2;
    constexpr int NP = T::kG / 2;
    constexpr int A = kUpsAhead;
    constexpr int STEPS = RUN + 4;
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ uint4 slots[4][kUpsSlotPieces];
    const int od = d * 2, oh = h * 2, ow = w * 2;
    const int nz = od - 2 * margin, ny = oh - 2 * margin, nx = ow - 2 * margin;
    const int nzp = (nz + 1) >> 1, nyp = (ny + 1) >> 1;
    const int nruns = nzp / RUN;                 // (the launcher checks nzp % RUN == 0)
    const int nb = blockIdx.x / nruns, run = blockIdx.x - nb * nruns;
    const int pr0 = run * RUN;
    const int p0 = (margin >> 1) + pr0;         // >= 2: pair p wants planes (p - 1, p, p + 1)
    // the workgroup's source rows start at the first item's first row
    const int yp_first = (int)((blockIdx.y * 256u) >> 1) / nx;
    int rmin, r1;
    float l1;
    lerp_coord(margin + 2 * yp_first, h, sy, rmin, r1, l1);
    const unsigned image0 = (unsigned)(rmin * w * cg);           // first piece of the image inside a source plane
    const size_t splane = (size_t)h * w * cg * 16;
    // both sides synchronise with the bare barrier instruction: LDS traffic settled (lgkmcnt), nothing else
    auto step_barrier = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    if (threadIdx.x >= 256) {
        // ---- wave 4: global -> registers -> LDS, nothing else
        const int lane = threadIdx.x - 256;
        const __amdgpu_buffer_rsrc_t srs = layer_rsrc(reinterpret_cast<const char*>(src) + (size_t)nb * d * splane,
                                                      (size_t)d * splane);
        unsigned gpos[8];                          // byte offsets inside a plane; past the plane: dropped
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned piece = image0 + lane + 64u * k;
            gpos[k] = piece < (unsigned)(h * w * cg) ? piece * 16u : 0x80000000u;
        }
        struct Eight { uint4 v[8]; };
        auto fetch = [&](int p) {                  // the image of source plane p (clamped)
            Eight r;
            const unsigned so = (unsigned)(p < d - 1 ? p : d - 1) * (unsigned)splane;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#if UPS_ABLATE & 1
                r.v[k] = make_uint4(so, gpos[k], 0, 0);
#else
                const u32x4_b t = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)gpos[k], (int)so, 0);
                r.v[k] = make_uint4(t.x, t.y, t.z, t.w);
#endif
            }
            return r;
        };
        Eight fl[A];
#pragma unroll
        for (int k = 0; k < A; ++k) fl[k] = fetch(p0 - 2 + k);
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            if (t < STEPS - 1) {                   // (the last step has nothing left to park)
#pragma unroll
                for (int k = 0; k < 8; ++k) slots[(p0 - 2 + t) & 3][lane + 64 * k] = fl[t % A].v[k];
                if (t + A < STEPS - 1) fl[t % A] = fetch(p0 - 2 + t + A);
            }
            step_barrier();
        }
        return;
    }

    // ---- waves 0-3: LDS -> interpolation -> stores
    const int nitems = nyp * nx * cg;
    const unsigned item = blockIdx.y * 256u + threadIdx.x;
    const bool live = item < (unsigned)nitems;
    const unsigned it = live ? item : (unsigned)nitems - 1u;
    const int g = it & 1;
    const int yp = (int)(it >> 1) / nx, x = margin + (int)(it >> 1) - yp * nx;
    const int ya = margin + 2 * yp;
    const bool has_yb = ya + 1 < oh - margin;
    const LerpPair py = lerp_pair(ya, has_yb, h, sy);
    int x0, x1;
    float lx;
    lerp_coord(x, w, sx, x0, x1, lx);
    const f2 wx0 = {1.f - lx, 1.f - lx}, wx1 = {lx, lx};
    // this thread's six pieces inside a slot
    const unsigned lpos[3][2] = {{(unsigned)((py.s0 * w + x0) * cg + g) - image0, (unsigned)((py.s0 * w + x1) * cg + g) - image0},
                                 {(unsigned)((py.s1 * w + x0) * cg + g) - image0, (unsigned)((py.s1 * w + x1) * cg + g) - image0},
                                 {(unsigned)((py.s2 * w + x0) * cg + g) - image0, (unsigned)((py.s2 * w + x1) * cg + g) - image0}};
    const float yA[3] = {py.a0, py.a1, 0.f}, yB[3] = {py.b0, py.b1, py.b2};
    const size_t oplane = (size_t)oh * ow * cg * 16;
    char* const obase = reinterpret_cast<char*>(dst) + (size_t)nb * od * oplane;
    const unsigned orow[2] = {live ? (unsigned)(((ya + 0) * ow + x) * cg + g) * 16u : 0x80000000u,
                              live && has_yb ? (unsigned)(((ya + 1) * ow + x) * cg + g) * 16u : 0x80000000u};

    auto interp = [&](int p, f2 (*q)[NP]) {        // x and y interpolation of plane p, as upsample2_kernel's load_plane
        const uint4* const sl = slots[p & 3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float v0[T::kG], v1[T::kG];
            T::unpack(sl[lpos[rr][0]], v0);
            T::unpack(sl[lpos[rr][1]], v1);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const f2 xr = __builtin_elementwise_fma(wx1, (f2){v1[2 * j], v1[2 * j + 1]},
                                                        wx0 * (f2){v0[2 * j], v0[2 * j + 1]});
                const f2 wa = {yA[rr], yA[rr]}, wb = {yB[rr], yB[rr]};
                if (rr == 0) {
                    q[0][j] = wa * xr;
                    q[1][j] = wb * xr;
                } else {
                    if (rr == 1) q[0][j] = __builtin_elementwise_fma(wa, xr, q[0][j]);
                    q[1][j] = __builtin_elementwise_fma(wb, xr, q[1][j]);
                }
            }
        }
    };
    auto emit_pair = [&](int za, const LerpPair& pz, const f2 (*v0)[NP], const f2 (*v1)[NP], const f2 (*v2)[NP]) {
        const bool has_zb = za + 1 < od - margin;
#pragma unroll
        for (int zz = 0; zz < 2; ++zz) {
            const __amdgpu_buffer_rsrc_t ors =
                layer_rsrc(obase, ((UPS_ABLATE & 2) ? d < 0 : true) && (zz == 0 || has_zb) ? (size_t)od * oplane : (size_t)0);
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                float out[T::kG];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const f2 o = zz == 0
                        ? __builtin_elementwise_fma((f2){pz.a1, pz.a1}, v1[yy][j], (f2){pz.a0, pz.a0} * v0[yy][j])
                        : __builtin_elementwise_fma((f2){pz.b2, pz.b2}, v2[yy][j],
                              __builtin_elementwise_fma((f2){pz.b1, pz.b1}, v1[yy][j], (f2){pz.b0, pz.b0} * v0[yy][j]));
                    out[2 * j] = o.x; out[2 * j + 1] = o.y;
                }
                buf_store16_counted(T::pack(out), ors, orow[yy], (unsigned)(za + zz) * (unsigned)oplane);
            }
        }
    };

    f2 q0[2][NP], q1[2][NP], q2[2][NP];
#pragma unroll
    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
        for (int j = 0; j < NP; ++j) q0[yy][j] = q1[yy][j] = q2[yy][j] = (f2){0.f, 0.f};
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
        if (t >= 1) {
#pragma unroll
            for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                for (int j = 0; j < NP; ++j) { q0[yy][j] = q1[yy][j]; q1[yy][j] = q2[yy][j]; }
            interp(p0 - 3 + t, q2);
        }
        if (t >= 4) {
            const int za = margin + 2 * (pr0 + t - 4);
            const LerpPair pz = lerp_pair(za, za + 1 < od - margin, d, sz);
            emit_pair(za, pz, q0, q1, q2);
        }
        step_barrier();
    }
}

// rows of source a workgroup of upsample2_strip_kernel shares, over all its workgroups (host copy of the
// kernel's float arithmetic)
static int upsample_strip_rows(int h, int w, int margin, float sy) {
    const int oh = 2 * h, ny = oh - 2 * margin, nx = 2 * w - 2 * margin;
    const int nyp = (ny + 1) / 2, nitems = nyp * nx * 2;
    auto i0 = [&](int o) { const int v = (int)floorf(sy * (float)o); return v < h - 1 ? v : h - 1; };
    int rows = 0;
    for (int first = 0; first < nitems; first += 256) {
        const int last = (first + 255 < nitems ? first + 255 : nitems - 1);
        const int ypf = (first >> 1) / nx, ypl = (last >> 1) / nx;
        const int lo = i0(margin + 2 * ypf);
        int hi = i0(margin + 2 * ypl) + 2;
        if (hi > h - 1) hi = h - 1;
        if (hi - lo + 1 > rows) rows = hi - lo + 1;
    }
    return rows;
}

// does pair p of a x2 upsampling of n planes read source planes max(p - 1, 0) .. + 2, in float arithmetic too?
static bool upsample_schedule_is_regular(int n, float scale) {
    for (int p = 0; p < n; ++p) {
        const float s = scale * (float)(2 * p);
        int i0 = (int)floorf(s);
        if (i0 > n - 1) i0 = n - 1;
        if (i0 != (p > 0 ? p - 1 : 0)) return false;
        const float s1 = scale * (float)(2 * p + 1);
        int j0 = (int)floorf(s1);
        if (j0 > n - 1) j0 = n - 1;
        if (j0 != i0 && j0 != i0 + 1) return false;
    }
    return true;
}

// ---- head: 1x1x1 conv (+ sigmoid), channels-last -> NCDHW float32 -----------
template <typename T, int OC>
__global__ __launch_bounds__(256) void head_kernel(const uint4* __restrict__ src,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ out, unsigned nvox_per_patch,
                                                   int c0p, int apply_sigmoid) {
    const unsigned sp = blockIdx.x * blockDim.x + threadIdx.x;  // voxel inside the patch
    if (sp >= nvox_per_patch) return;
    const unsigned nb = blockIdx.y;
    const int cg = c0p / T::kG;
    // group g of the voxel: half (g & 1) of its record in chunk plane g >> 1
    const uint4* rec = src + ((size_t)nb * (cg / 2) * nvox_per_patch + sp) * 2;
    float acc[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) acc[o] = bias[o];
    for (int g = 0; g < cg; ++g) {
        float f[T::kG];
        T::unpack(rec[(size_t)(g >> 1) * nvox_per_patch * 2 + (g & 1)], f);
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int j = 0; j < T::kG; ++j)
                acc[o] = fmaf(f[j], w[o * c0p + g * T::kG + j], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < OC; ++o) {
        float r = acc[o];
        if (apply_sigmoid) r = 1.f / (1.f + expf(-r));
        out[((size_t)nb * OC + o) * nvox_per_patch + sp] = r;
    }
}

// ---- range probe: largest |value| of an activation tensor ----------------------
// (exaspim_unet_forward_absmax: what a checkpoint's activations reach, per layer, before a
// 16-bit storage type is trusted with them). Non-negative floats order like their bit
// patterns, so the maximum is an atomicMax on unsigned.
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const uint4* __restrict__ src, size_t pieces,
                                                     unsigned* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pieces;
         i += (size_t)gridDim.x * blockDim.x) {
        float f[T::kG];
        T::unpack(src[i], f);
#pragma unroll
        for (int j = 0; j < T::kG; ++j) {
            const float a = fabsf(f[j]);
            m = a > m || a != a ? a : m;      // (a NaN is kept and reported)
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = o > m || o != o ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// ---- launchers ----------------------------------------------------------------
static inline unsigned stream_grid(size_t items) {
    const size_t blocks = (items + 255) / 256;
    return (unsigned)(blocks < 8192 ? (blocks ? blocks : 1) : 8192);
}

#define DISPATCH_T(dtype, CALL)                                  \
    switch (dtype) {                                             \
        case EXASPIM_DT_F32: { using T = F32T; CALL; } break;    \
        case EXASPIM_DT_BF16: { using T = BF16T; CALL; } break;  \
        case EXASPIM_DT_F16: { using T = F16T; CALL; } break;    \
        default: set_error("unknown dtype %d", dtype); return EXASPIM_E_INVALID; \
    }

int launch_conv_first(int dtype, const float* x, float* xpad, const float* w, const float* bias,
                      void* dst, int n, int d, int h, int wd, int c0p, float slope,
                      hipStream_t stream, bool first_no_strips) {
    const size_t nvox = (size_t)n * d * h * wd;
    constexpr int MT = 4;
    const size_t blocks = (nvox + 4 * MT * 32 - 1) / (4 * MT * 32);
    EXA_CHECK_ARG(nvox > 0 && nvox < 0x7fffffffULL && c0p % 32 == 0, "conv_first: bad size");
    EXA_CHECK_ARG((long long)n * (d + 2) <= 65535 && h + 2 <= 65535, "conv_first: grid too large");
    const unsigned pad_blocks = (unsigned)((long long)n * (d + 2));
    unsigned* const xsplit = reinterpret_cast<unsigned*>(xpad);   // 16-bit modes: hi | lo << 16 per voxel
    // x == nullptr: xpad already holds the operand layout (exaspim_gather_patches_as wrote it)
    if (x) switch (dtype) {
        case EXASPIM_DT_F32: pad_input_kernel<<<pad_blocks, 256, 0, stream>>>(x, xpad, d, h, wd); break;
        case EXASPIM_DT_BF16: pad_split_kernel<BF16T><<<pad_blocks, 256, 0, stream>>>(x, xsplit, d, h, wd); break;
        case EXASPIM_DT_F16: pad_split_kernel<F16T><<<pad_blocks, 256, 0, stream>>>(x, xsplit, d, h, wd); break;
        default: set_error("unknown dtype %d", dtype); return EXASPIM_E_INVALID;
    }
    EXA_CHECK_HIP(hipGetLastError());
    dim3 grid((unsigned)blocks, c0p / 32);
    // the 16-bit kernel's waves walk the 32-voxel groups: 8 workgroups per CU are plenty
    const size_t groups = (nvox + 31) / 32;
    dim3 grid16((unsigned)(groups / 4 < 2048 ? (groups + 3) / 4 : 2048), c0p / 32);
    // ROWS variant: whole 32-voxel groups inside a row, one padded / output patch per 32-bit descriptor
    const bool rows = wd % 32 == 0 && (size_t)(d + 2) * (h + 2) * (wd + 2) * 4 < 0x7fffffffULL &&
                      (size_t)c0p * 2 * d * h * wd < 0x7fffffffULL;
    // row strips when the rows variant applies, a strip is whole rows and the tile fits the staging registers
    const bool strips = rows && !first_no_strips && h % kStripRows == 0 && wd <= 128;
    if (strips && dtype != EXASPIM_DT_F32) {
        const int nstrips = n * d * (h / kStripRows);
        const size_t tile_bytes = ((size_t)3 * (kStripRows + 2) * (wd + 2) * 4 + 15) / 16 * 16;
        const size_t lds = 2 * tile_bytes + 4 * 32 * (32 * 2 + 16);
        dim3 sgrid((unsigned)(nstrips < 1024 ? nstrips : 1024), c0p / 32);
#define STRIP(TT, GG) conv_first16_strip_kernel<TT, GG><<<sgrid, 256, lds, stream>>>(xsplit, w, bias, dst, n, d, h, wd, c0p, slope)
        const bool bf = dtype == EXASPIM_DT_BF16;
        switch (wd / 32) {
            case 1: if (bf) STRIP(BF16T, 1); else STRIP(F16T, 1); break;
            case 2: if (bf) STRIP(BF16T, 2); else STRIP(F16T, 2); break;
            case 3: if (bf) STRIP(BF16T, 3); else STRIP(F16T, 3); break;
            default: if (bf) STRIP(BF16T, 4); else STRIP(F16T, 4); break;
        }
#undef STRIP
        EXA_CHECK_HIP(hipGetLastError());
        return EXASPIM_OK;
    }
    switch (dtype) {
        case EXASPIM_DT_F32:
            conv_first_kernel<F32T, MT><<<grid, 256, 0, stream>>>(xpad, w, bias, dst, (int)nvox, d, h, wd, c0p, slope);
            break;
        case EXASPIM_DT_BF16:
            if (rows)
                conv_first16_kernel<BF16T, true><<<grid16, 256, 0, stream>>>(xsplit, w, bias, dst, (int)nvox, d, h, wd, c0p, slope);
            else
                conv_first16_kernel<BF16T, false><<<grid16, 256, 0, stream>>>(xsplit, w, bias, dst, (int)nvox, d, h, wd, c0p, slope);
            break;
        case EXASPIM_DT_F16:
            if (rows)
                conv_first16_kernel<F16T, true><<<grid16, 256, 0, stream>>>(xsplit, w, bias, dst, (int)nvox, d, h, wd, c0p, slope);
            else
                conv_first16_kernel<F16T, false><<<grid16, 256, 0, stream>>>(xsplit, w, bias, dst, (int)nvox, d, h, wd, c0p, slope);
            break;
        default: set_error("unknown dtype %d", dtype); return EXASPIM_E_INVALID;
    }
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

int launch_convt2(int dtype, const void* src, const void* weights, const float* bias, void* dst,
                  int n, int d, int h, int w, int cin, int cout, hipStream_t stream) {
    const size_t nvox = (size_t)n * d * h * w;
    EXA_CHECK_ARG(nvox > 0 && nvox < 0x7fffffffULL && cin % 32 == 0 && cout % 32 == 0,
                  "convt: bad size (%zu voxels, %d -> %d channels)", nvox, cin, cout);
    dim3 grid((unsigned)((nvox + 127) / 128), cout / 32);
    DISPATCH_T(dtype, (convt2_kernel<T><<<grid, 256, 0, stream>>>(
                          static_cast<const uint4*>(src), static_cast<const uint4*>(weights), bias,
                          dst, (int)nvox, d, h, w, cin, cout)));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

int launch_maxpool2(int dtype, const void* src, void* dst, int n, int d, int h, int w,
                    int c, hipStream_t stream) {
    EXA_CHECK_ARG(d % 2 == 0 && h % 2 == 0 && w % 2 == 0, "maxpool: odd size %dx%dx%d", d, h, w);
    constexpr int cg = 2;                              // 16-byte groups of a 32-byte record
    const int nv = n * (c * dtype_size(dtype) / 32);   // chunk planes = independent volumes
    const long long plane = (long long)(h / 2) * (w / 2) * cg;
    EXA_CHECK_ARG((long long)nv * (d / 2) <= 0x7fffffffLL && (plane + 255) / 256 <= 65535,
                  "maxpool: grid too large");
    dim3 grid(nv * (d / 2), (unsigned)((plane + 255) / 256));
    DISPATCH_T(dtype, (maxpool2_kernel<T><<<grid, 256, 0, stream>>>(
                          static_cast<const uint4*>(src), static_cast<uint4*>(dst), d, h, w, cg)));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

int launch_upsample2(int dtype, const void* src, void* dst, int n, int d, int h, int w,
                     int c, int margin, hipStream_t stream, bool plain_kernel, bool per_thread) {
    if (margin < 0 || margin >= d || margin >= h || margin >= w) margin = 0;
    const int nv = n * (c * dtype_size(dtype) / 32);   // chunk planes = independent volumes
    // items of one pair of output planes: (row pair, column, 16-byte group)
    const long long items = (long long)((h * 2 - 2 * margin + 1) / 2) * (w * 2 - 2 * margin) * 2;
    const int nzp = (d * 2 - 2 * margin + 1) / 2;                              // pairs of output planes
    const long long planes = (long long)nv * ((nzp + kUpsZRun - 1) / kUpsZRun);  // runs of pairs
    EXA_CHECK_ARG(planes <= 0x7fffffffLL && (items + 255) / 256 <= 65535 &&
                      (long long)nv * d * h * w * 2 < 0x7fffffffLL,
                  "upsample: grid too large");
    auto scale = [](int in) { return in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f; };
    dim3 grid((unsigned)planes, (unsigned)((items + 255) / 256));
    // the software-pipelined kernel when its schedule holds: even first plane, at least three source
    // planes, one volume per 32-bit descriptor ("plain_kernel", EXASPIM_OPT_PLAIN_UPSAMPLE: the tests
    // hold the two kernels to each other bit for bit)
    const bool pipe = !plain_kernel && margin % 2 == 0 && d >= 3 && upsample_schedule_is_regular(d, scale(d)) &&
                      (size_t)8 * d * h * w * 32 < 0x7fffffffULL;
    // ... and with the workgroup's source rows shared through LDS when, on top of that, no pair of the launch
    // is one of the two that do not advance, the pairs are a whole number of runs and the rows fit a slot
    // ("per_thread", EXASPIM_OPT_UPSAMPLE_PER_THREAD: the per-thread pipeline instead, for the tests)
    if (pipe && !per_thread && margin >= 4 && upsample_strip_rows(h, w, margin, scale(h)) * w * 2 <= kUpsSlotPieces) {
        const int run = nzp % 14 == 0 ? 14 : nzp % 12 == 0 ? 12 : 0;
        if (run) {
            dim3 sgrid((unsigned)(nv * (nzp / run)), (unsigned)((items + 255) / 256));
#define UPS_STRIP(RR) DISPATCH_T(dtype, (upsample2_strip_kernel<T, RR><<<sgrid, 320, 0, stream>>>(   \
                          static_cast<const uint4*>(src), static_cast<uint4*>(dst), d, h, w, scale(d), \
                          scale(h), scale(w), margin)))
            if (run == 14) { UPS_STRIP(14); } else { UPS_STRIP(12); }
#undef UPS_STRIP
            EXA_CHECK_HIP(hipGetLastError());
            return EXASPIM_OK;
        }
    }
    if (pipe) {
        DISPATCH_T(dtype, (upsample2_pipe_kernel<T><<<grid, 256, 0, stream>>>(
                              static_cast<const uint4*>(src), static_cast<uint4*>(dst), d, h, w,
                              scale(d), scale(h), scale(w), margin)));
        EXA_CHECK_HIP(hipGetLastError());
        return EXASPIM_OK;
    }
    DISPATCH_T(dtype, (upsample2_kernel<T><<<grid, 256, 0, stream>>>(
                          static_cast<const uint4*>(src), static_cast<uint4*>(dst), d, h, w,
                          scale(d), scale(h), scale(w), margin)));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

int launch_absmax(int dtype, const void* src, size_t bytes, float* out, hipStream_t stream) {
    EXA_CHECK_ARG(bytes % 16 == 0, "absmax: %zu bytes is not a whole number of 16-byte pieces", bytes);
    if (bytes == 0) return EXASPIM_OK;
    const size_t pieces = bytes / 16;
    DISPATCH_T(dtype, (absmax_kernel<T><<<stream_grid(pieces), 256, 0, stream>>>(
                          static_cast<const uint4*>(src), pieces, reinterpret_cast<unsigned*>(out))));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

template <typename T>
static int launch_head_t(const void* src, const float* w, const float* bias, float* out,
                         size_t nvox, int n, int c0p, int oc, int sig, hipStream_t stream) {
    EXA_CHECK_ARG(nvox < 0x7fffffffULL && n <= 65535, "head: patch too large");
    dim3 grid((unsigned)((nvox + 255) / 256), n);
    const uint4* s = static_cast<const uint4*>(src);
    const unsigned nv = (unsigned)nvox;
    switch (oc) {
        case 1: head_kernel<T, 1><<<grid, 256, 0, stream>>>(s, w, bias, out, nv, c0p, sig); break;
        case 2: head_kernel<T, 2><<<grid, 256, 0, stream>>>(s, w, bias, out, nv, c0p, sig); break;
        case 3: head_kernel<T, 3><<<grid, 256, 0, stream>>>(s, w, bias, out, nv, c0p, sig); break;
        case 4: head_kernel<T, 4><<<grid, 256, 0, stream>>>(s, w, bias, out, nv, c0p, sig); break;
        default: set_error("head: out_channels %d unsupported", oc); return EXASPIM_E_INVALID;
    }
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

int launch_head(int dtype, const void* src, const float* w, const float* bias, float* out,
                int n, int d, int h, int wd, int c0p, int out_channels, int apply_sigmoid,
                hipStream_t stream) {
    const size_t nvox = (size_t)d * h * wd;
    DISPATCH_T(dtype, return (launch_head_t<T>(src, w, bias, out, nvox, n, c0p, out_channels,
                                               apply_sigmoid, stream)));
    return EXASPIM_OK;
}

}  // namespace exaspim
