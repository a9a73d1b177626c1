// 3x3x3 convolution (+ folded BatchNorm bias + LeakyReLU) as an implicit GEMM
// on the gfx950 matrix cores. Replaces every nn.Conv3d(k=3,p=1) ->
// BatchNorm3d(eval) -> LeakyReLU(0.01) triple of the reference's DoubleConv
// (machine_learning/unet3d.py:142-149) except inc.0 (Cin = 1, layers.hip), and
// torch.cat([skip, up], dim=1) (unet3d.py:288) by reading two sources.
//
// Layout. Activations are channels-last (N, D, H, W, C) with C padded to 32.
// One workgroup owns a TZ x TY x TX block of output voxels of one patch and
// NWG = WAVES_N * NT * 32 output channels. The K dimension (27 taps x Cin) is
// walked in chunks of 32 bytes of input channels (8 x f32 / 16 x 16-bit): per
// chunk the (TZ+2)(TY+2)(TX+2) halo block is staged in LDS as two planes of
// 16-byte channel groups, [group][halo voxel]; zero padding at patch borders
// is written as zeros. A wave's MFMA B operand (activations, voxel on the lane)
// is then ONE ds_read_b128 per (tap, 32 voxels): lanes 0-31 read plane 0, lanes
// 32-63 plane 1, 32 consecutive voxels -> conflict-free. The A operand
// (weights) comes straight from global memory in fragment order (plan.cpp), 1
// KiB per wave-instruction, shared by all MT voxel tiles of the wave.
//
// D = W(32 cout x K) * X(K x 32 voxels): the accumulator keeps the voxel on
// the lane and 4-channel runs in registers, so the epilogue stores 16 bytes
// (f32) / 8 bytes (16-bit) of consecutive channels per lane.
//
// f32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 4 per chunk-tap),
// bf16/f16 use v_mfma_f32_32x32x16_{bf16,f16} (one per chunk-tap).

#include <cstdlib>

#include "common.h"

namespace exaspim {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct F32Tag { static constexpr int kG = 4; };
struct BF16Tag { static constexpr int kG = 8; };
struct F16Tag { static constexpr int kG = 8; };

template <typename Tag>
__device__ __forceinline__ void mma(f32x16& acc, const uint4& wf, const uint4& xf);

template <>
__device__ __forceinline__ void mma<F32Tag>(f32x16& acc, const uint4& wf, const uint4& xf) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.x), __uint_as_float(xf.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.y), __uint_as_float(xf.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.z), __uint_as_float(xf.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wf.w), __uint_as_float(xf.w), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma<BF16Tag>(f32x16& acc, const uint4& wf, const uint4& xf) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf),
                                                  __builtin_bit_cast(bf16x8, xf), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma<F16Tag>(f32x16& acc, const uint4& wf, const uint4& xf) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf),
                                                 __builtin_bit_cast(f16x8, xf), acc, 0, 0, 0);
}

// 16-byte buffer load with hardware range check: an offset at or beyond the
// descriptor's size returns zeros, which is how the conv's zero padding (and
// the tail of the staging list) is produced without branches.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOutOfRange = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// store 4 consecutive output channels of one voxel
template <typename Tag>
__device__ __forceinline__ void store4(void* dst, size_t elem_off, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<F32Tag>(void* dst, size_t off, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(static_cast<float*>(dst) + off) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<BF16Tag>(void* dst, size_t off, float a, float b, float c, float d) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(dst) + off) = v;
}
template <>
__device__ __forceinline__ void store4<F16Tag>(void* dst, size_t off, float a, float b, float c, float d) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    *reinterpret_cast<f16x4*>(static_cast<_Float16*>(dst) + off) = v;
}

template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int MT, int NT, int MINW, int PD = 3>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv3x3x3_kernel(
    ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int G = Tag::kG;            // elements per 16 B
    constexpr int KC = 2 * G;             // channels per chunk
    constexpr int ES = 16 / G;            // element size in bytes
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HV = HZ * HY * HX;      // halo voxels; LDS image = [2 groups][HV] x 16 B
    constexpr int NTHREADS = WAVES_M * WAVES_N * 64;
    constexpr int TILE_VOX = TZ * TY * TX;
    constexpr int NITEMS = (2 * HV + NTHREADS - 1) / NTHREADS;  // 16-byte pieces per thread
    static_assert(WAVES_M * MT * 32 >= TILE_VOX, "tile not covered by the waves");

    __shared__ __attribute__((aligned(16))) uint4 lds[2 * HV];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int half = lane >> 5;
    const int r = lane & 31;

    // XCD-aware block order: workgroups b, b+8, ... share an XCD (round-robin
    // dispatch), so give each XCD a contiguous run of tiles -- neighbouring
    // tiles then find each other's halo voxels in the same 4 MiB L2.
    int bid;
    {
        const int nblk = gridDim.x, q = nblk >> 3, rem = nblk & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int tz = bid % tiles_z; bid /= tiles_z;
    const int nb = bid;  // patch index in the batch
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;

    const int ntiles = a.cout >> 5;
    const int ntile0 = (blockIdx.y * WAVES_N + wn) * NT;

    int base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wm * MT + mt) * 32 + r;
        m = m < TILE_VOX ? m : TILE_VOX - 1;
        const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
        base[mt] = (z * HY + y) * HX + x + half * HV;
    }

    // Per-thread staging descriptors: piece i = tid + it * NTHREADS of the LDS
    // image is channel group (i >= HV) of halo voxel (i mod HV); its source is
    // voxel vidx[it] of this patch, or nothing (conv zero padding / tile tail).
    const size_t patch_vox = (size_t)a.d * a.h * a.w;
    int vidx[NITEMS];
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) {
        const int i = tid + it * NTHREADS;
        const int hv = i >= HV ? i - HV : i;
        const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = i < 2 * HV && (unsigned)gz < (unsigned)a.d &&
                        (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        vidx[it] = ok ? (gz * a.h + gy) * a.w + gx : -1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nchunks = (a.ca + a.cb) / KC;
    for (int c = 0; c < nchunks; ++c) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const uint4* wp = static_cast<const uint4*>(a.weights) +
                          ((size_t)c * 27 * ntiles + ntile0) * 64 + lane;
        // first weight fragments of the chunk: in flight while the halo streams in
        uint4 wring[PD + 1][NT];
#pragma unroll
        for (int t = 0; t < PD; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wring[t][nt] = wp[((size_t)t * ntiles + nt) * 64];

        __syncthreads();  // every wave is done reading the previous chunk
        {
            // LDS-DMA staging: buffer_load_dwordx4 ... lds writes 16 B per lane at
            // (wave base + lane * 16) with no register round trip; out-of-range
            // offsets return zeros = the conv's zero padding.
            const size_t rec_bytes = patch_vox * cs * ES;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(src) + (size_t)nb * rec_bytes, 0, (int)rec_bytes, 0x00020000);
#pragma unroll
            for (int it = 0; it < NITEMS; ++it) {
                const int i = tid + it * NTHREADS;
                if (i < 2 * HV) {
                    const int kg = i >= HV ? 1 : 0;
                    const unsigned off = vidx[it] >= 0
                                             ? (unsigned)((vidx[it] * cs + ch0 + kg * G) * ES)
                                             : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rsrc, (__attribute__((address_space(3))) void*)(lds + i), 16, off, 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int tapoff = ((t / 9) * HY + (t / 3) % 3) * HX + t % 3;
            if (t + PD < 27) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wring[(t + PD) % (PD + 1)][nt] = wp[((size_t)(t + PD) * ntiles + nt) * 64];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint4 xf = lds[base[mt] + tapoff];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma<Tag>(acc[mt][nt], wring[t % (PD + 1)][nt], xf);
            }
        }
    }

    // epilogue: bias + LeakyReLU (unet3d.py:145,148), 4-channel runs per lane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = (wm * MT + mt) * 32 + r;
        const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
        const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
        const bool ok = m < TILE_VOX && gz < a.d && gy < a.h && gx < a.w;
        if (!ok) continue;
        const size_t vox = (((size_t)nb * a.d + gz) * a.h + gy) * a.w + gx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = (ntile0 + nt) * 32 + 8 * q + 4 * half;
                const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
                float v0 = acc[mt][nt][4 * q + 0] + b.x;
                float v1 = acc[mt][nt][4 * q + 1] + b.y;
                float v2 = acc[mt][nt][4 * q + 2] + b.z;
                float v3 = acc[mt][nt][4 * q + 3] + b.w;
                v0 = v0 > 0.f ? v0 : v0 * a.slope;
                v1 = v1 > 0.f ? v1 : v1 * a.slope;
                v2 = v2 > 0.f ? v2 : v2 * a.slope;
                v3 = v3 > 0.f ? v3 : v3 * a.slope;
                store4<Tag>(a.dst, vox * a.cout + co, v0, v1, v2, v3);
            }
        }
    }
}

// ---- v3: register-staged prefetch (async-STAGE split), deeper operand
// pipelining and an LDS-transposed epilogue -------------------------------------
// Same tiling and LDS image as above. Differences:
//  * the next chunk's halo pieces are loaded global -> VGPR late in the current
//    chunk's tap loop (after every weight load of the chunk has been issued, so
//    the in-order vmcnt never makes a weight wait behind the prefetch), and are
//    written to LDS after the chunk's last MFMA: HBM/L2 latency hides under MFMAs
//    of the same workgroup instead of relying on a second workgroup;
//  * x fragments are double-buffered per tap (all MT reads of tap t+1 in flight
//    under the MFMAs of tap t);
//  * outputs go through LDS so every store instruction writes whole 16-byte
//    pieces of consecutive voxel records (1 KiB contiguous per instruction when
//    the tile row is 16 voxels of 32 channels).
template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int MT, int NT, int MINW, int PD>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv3x3x3_t14(
    ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int G = Tag::kG;
    constexpr int KC = 2 * G;
    constexpr int ES = 16 / G;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HV = HZ * HY * HX;
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int TILE_VOX = TZ * TY * TX;
    constexpr int NITEMS = (2 * HV + NTHREADS - 1) / NTHREADS;
    constexpr int RECB = NT * 32 * ES;              // bytes of one voxel's output slice
    constexpr int EPI_UNITS = NWAVES * 32 * RECB / 16;
    constexpr int LDS_UNITS = 2 * HV > EPI_UNITS ? 2 * HV : EPI_UNITS;
    constexpr int ISSUE_T = 26 - PD > 0 ? 26 - PD : 0;  // tap at which the prefetch is issued
    static_assert(WAVES_M * MT * 32 >= TILE_VOX, "tile not covered by the waves");

    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_UNITS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int half = lane >> 5;
    const int r = lane & 31;

    int bid;
    {
        const int nblk = gridDim.x, q = nblk >> 3, rem = nblk & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int tz = bid % tiles_z; bid /= tiles_z;
    const int nb = bid;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;

    const int ntiles = a.cout >> 5;
    const int ntile0 = (blockIdx.y * WAVES_N + wn) * NT;

    int base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wm * MT + mt) * 32 + r;
        m = m < TILE_VOX ? m : TILE_VOX - 1;
        const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
        base[mt] = (z * HY + y) * HX + x + half * HV;
    }

    const size_t patch_vox = (size_t)a.d * a.h * a.w;
    int vidx[NITEMS];
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) {
        const int i = tid + it * NTHREADS;
        const int hv = i >= HV ? i - HV : i;
        const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = i < 2 * HV && (unsigned)gz < (unsigned)a.d &&
                        (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        vidx[it] = ok ? (gz * a.h + gy) * a.w + gx : -1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nchunks = (a.ca + a.cb) / KC;
    uint4 stg[NITEMS];

    auto stage_load = [&](int c) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const unsigned rowb = cs * ES;  // bytes of one voxel record of this source
        const __amdgpu_buffer_rsrc_t rsrc =
            make_rsrc(src + (size_t)nb * patch_vox * rowb, patch_vox * rowb);
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const int i = tid + it * NTHREADS;
            const unsigned voff = vidx[it] >= 0 ? (unsigned)vidx[it] * rowb + (i >= HV ? 16u : 0u)
                                                : kOutOfRange;
            stg[it] = buf_load16(rsrc, voff, ch0 * ES);
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const int i = tid + it * NTHREADS;
            if (i < 2 * HV) lds[i] = stg[it];
        }
    };

    stage_load(0);
    stage_store();
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const uint4* wp = static_cast<const uint4*>(a.weights) +
                          ((size_t)c * 27 * ntiles + ntile0) * 64 + lane;
        uint4 wring[PD + 1][NT];
#pragma unroll
        for (int t = 0; t < PD; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wring[t][nt] = wp[((size_t)t * ntiles + nt) * 64];

        uint4 xf[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[0][mt] = lds[base[mt]];

        const bool more = c + 1 < nchunks;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            if (t + PD < 27) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wring[(t + PD) % (PD + 1)][nt] = wp[((size_t)(t + PD) * ntiles + nt) * 64];
            }
            if (t == ISSUE_T && more) stage_load(c + 1);
            if (t + 1 < 27) {
                const int tapoff = (((t + 1) / 9) * HY + ((t + 1) / 3) % 3) * HX + (t + 1) % 3;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[(t + 1) & 1][mt] = lds[base[mt] + tapoff];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    mma<Tag>(acc[mt][nt], wring[t % (PD + 1)][nt], xf[t & 1][mt]);
            // keep each tap's {prefetch issue, fragment reads, MFMAs} together: without
            // this fence hipcc hoists and sinks them across taps and the loop runs ~20 % slower
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // every wave is done reading this chunk's image
        if (more) {
            stage_store();
            __syncthreads();
        }
    }

    // ---- epilogue: bias + LeakyReLU, transposed through LDS ------------------
    char* wl = reinterpret_cast<char*>(lds) + wave * (32 * RECB);
    // bias first: a load inside the store loop would wait (vmcnt counts stores on
    // gfx950) for every store issued before it
    float4 bq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bq[nt][q] = *reinterpret_cast<const float4*>(a.bias + ntile0 * 32 + nt * 32 + 8 * q + 4 * half);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = nt * 32 + 8 * q + 4 * half;  // channel inside the slice
                const float4 b = bq[nt][q];
                float v0 = acc[mt][nt][4 * q + 0] + b.x;
                float v1 = acc[mt][nt][4 * q + 1] + b.y;
                float v2 = acc[mt][nt][4 * q + 2] + b.z;
                float v3 = acc[mt][nt][4 * q + 3] + b.w;
                v0 = v0 > 0.f ? v0 : v0 * a.slope;
                v1 = v1 > 0.f ? v1 : v1 * a.slope;
                v2 = v2 > 0.f ? v2 : v2 * a.slope;
                v3 = v3 > 0.f ? v3 : v3 * a.slope;
                store4<Tag>(wl, (size_t)(r * RECB) / ES + cl, v0, v1, v2, v3);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int PPV = RECB / 16;           // 16-byte pieces per voxel slice
        constexpr int ROUNDS = 32 * PPV / 64;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int p = k * 64 + lane;
            const int vv = p / PPV, part = p % PPV;
            const int m = (wm * MT + mt) * 32 + vv;
            const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
            const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
            const uint4 val = *reinterpret_cast<const uint4*>(wl + p * 16);
            if (m < TILE_VOX && gz < a.d && gy < a.h && gx < a.w) {
                const size_t vox = (((size_t)nb * a.d + gz) * a.h + gy) * a.w + gx;
                *reinterpret_cast<uint4*>(static_cast<char*>(a.dst) +
                                          (vox * a.cout + ntile0 * 32) * ES + part * 16) = val;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- v4: z-column tiles ---------------------------------------------------
// As v3, but a wave owns COLUMNS of the tile: YXW groups of 32 (y, x) positions
// times all TZ planes (MT = TZ * YXW accumulators). For a fixed in-plane tap
// (dy, dx) the operand fragment of input plane zin is the B operand of up to
// three MFMAs (dz = 0, 1, 2 -> output planes zin, zin-1, zin-2), so the N = 32
// GEMM needs (TZ + 2) LDS reads per 3 * TZ MFMAs instead of one read per MFMA.
// Halo rows are padded to HXP = TX (mod 16) 16-byte slots so that the two
// half-rows of a 32-voxel group never share a bank (no LDS conflicts for 16-
// wide rows); padding slots are never written or read.
// With WLDS the chunk's 27 weight fragments are staged in LDS as well (prefetched
// global -> VGPR with the halo, one copy per workgroup instead of one L2 read per
// wave): with 4 x 32-voxel tiles per wave the per-wave weight stream would
// otherwise be 3x the activation traffic.
template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int YXW, int NT, int MINW, int PDG, bool WLDS, bool PADX = true>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv3x3x3_zcol(
    ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int G = Tag::kG;
    constexpr int KC = 2 * G;
    constexpr int ES = 16 / G;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HXP = (TX == 32 || !PADX) ? HX : (HX + 15) / 16 * 16 + (TX % 16);  // row stride (slots)
    constexpr int HV = HZ * HY * HX;       // real halo voxels (staging items per group)
    constexpr int HVP = HZ * HY * HXP;     // slots per channel group
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int YXT = TY * TX / 32;      // 32-voxel (y, x) groups per plane
    constexpr int MT = TZ * YXW;
    constexpr int NITEMS = (2 * HV + NTHREADS - 1) / NTHREADS;
    constexpr int RECB = NT * 32 * ES;
    constexpr int EPI_UNITS = NWAVES * 32 * RECB / 16;
    constexpr int WUNITS = WLDS ? 27 * NT * 64 : 0;           // weight fragments in LDS
    constexpr int WITEMS = (WUNITS + NTHREADS - 1) / NTHREADS;
    constexpr int XUNITS = 2 * HVP > EPI_UNITS ? 2 * HVP : EPI_UNITS;
    constexpr int LDS_UNITS = XUNITS + WUNITS;
    static_assert(TY * TX % 32 == 0 && YXT == WAVES_M * YXW, "plane not covered by the waves");
    static_assert(!WLDS || WAVES_N == 1, "LDS weights assume one cout slice per workgroup");

    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_UNITS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int half = lane >> 5;
    const int r = lane & 31;

    int bid;
    {
        const int nblk = gridDim.x, q = nblk >> 3, rem = nblk & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int tz = bid % tiles_z; bid /= tiles_z;
    const int nb = bid;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;

    const int ntiles = a.cout >> 5;
    const int ntile0 = (blockIdx.y * WAVES_N + wn) * NT;

    // slot of this lane's voxel in plane 0 for each of the wave's (y, x) groups
    int col[YXW];
#pragma unroll
    for (int j = 0; j < YXW; ++j) {
        const int p = (wm * YXW + j) * 32 + r;  // position inside the plane
        col[j] = (p / TX) * HXP + (p % TX) + half * HVP;
    }

    const size_t patch_vox = (size_t)a.d * a.h * a.w;
    int vidx[NITEMS];   // source voxel of staging item, or -1
    int slot[NITEMS];   // its LDS slot
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) {
        const int i = tid + it * NTHREADS;
        const int kg = i >= HV ? 1 : 0;
        const int hv = i - kg * HV;
        const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool inr = i < 2 * HV;
        const bool ok = inr && (unsigned)gz < (unsigned)a.d && (unsigned)gy < (unsigned)a.h &&
                        (unsigned)gx < (unsigned)a.w;
        vidx[it] = ok ? (gz * a.h + gy) * a.w + gx : -1;
        slot[it] = inr ? kg * HVP + (hz * HY + hy) * HXP + hx : -1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nchunks = (a.ca + a.cb) / KC;
    uint4 stg[NITEMS + WITEMS];  // halo pieces, then weight fragments
    uint4* const wlds = lds + XUNITS;

    auto stage_load = [&](int c) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const unsigned rowb = cs * ES;  // bytes of one voxel record of this source
        const __amdgpu_buffer_rsrc_t rsrc =
            make_rsrc(src + (size_t)nb * patch_vox * rowb, patch_vox * rowb);
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const int i = tid + it * NTHREADS;
            const unsigned voff = vidx[it] >= 0 ? (unsigned)vidx[it] * rowb + (i >= HV ? 16u : 0u)
                                                : kOutOfRange;
            stg[it] = buf_load16(rsrc, voff, ch0 * ES);
        }
        if (WLDS) {
            const __amdgpu_buffer_rsrc_t wrsrc =
                make_rsrc(a.weights, (size_t)nchunks * 27 * ntiles * 1024);
#pragma unroll
            for (int it = 0; it < WITEMS; ++it) {
                const int i = tid + it * NTHREADS;
                const int tap = i / (NT * 64), rest = i % (NT * 64);
                const unsigned voff = i < WUNITS ? ((tap * ntiles + ntile0) * 64 + rest) * 16u : kOutOfRange;
                stg[NITEMS + it] = buf_load16(wrsrc, voff, c * 27 * ntiles * 1024);
            }
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int it = 0; it < NITEMS; ++it)
            if (slot[it] >= 0) lds[slot[it]] = stg[it];
        if (WLDS) {
#pragma unroll
            for (int it = 0; it < WITEMS; ++it) {
                const int i = tid + it * NTHREADS;
                if (i < WUNITS) wlds[i] = stg[NITEMS + it];
            }
        }
    };

    stage_load(0);
    stage_store();
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const uint4* wp = static_cast<const uint4*>(a.weights) +
                          ((size_t)c * 27 * ntiles + ntile0) * 64 + lane;
        // weight ring over in-plane taps g = dy * 3 + dx: three fragments (dz) each
        uint4 wring[PDG + 1][3][NT];
        if (!WLDS)
#pragma unroll
        for (int g = 0; g < PDG; ++g)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wring[g][dz][nt] = wp[((size_t)(dz * 9 + g) * ntiles + nt) * 64];

        const bool more = c + 1 < nchunks;
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if (WLDS) {
#pragma unroll
                for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wring[g % (PDG + 1)][dz][nt] = wlds[((dz * 9 + g) * NT + nt) * 64 + lane];
            } else if (g + PDG < 9) {
#pragma unroll
                for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wring[(g + PDG) % (PDG + 1)][dz][nt] =
                            wp[((size_t)(dz * 9 + g + PDG) * ntiles + nt) * 64];
            }
            if (g == (WLDS ? 0 : 9 - PDG - 1) && more) stage_load(c + 1);
            const int goff = (g / 3) * HXP + g % 3;
#pragma unroll
            for (int j = 0; j < YXW; ++j) {
                uint4 xf[HZ];
#pragma unroll
                for (int zin = 0; zin < HZ; ++zin) xf[zin] = lds[col[j] + zin * HY * HXP + goff];
#pragma unroll
                for (int zin = 0; zin < HZ; ++zin)
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz) {
                        const int z = zin - dz;
                        if (z >= 0 && z < TZ) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                mma<Tag>(acc[j * TZ + z][nt], wring[g % (PDG + 1)][dz][nt], xf[zin]);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more) {
            stage_store();
            __syncthreads();
        }
    }

    // ---- epilogue: bias + LeakyReLU, transposed through LDS ------------------
    char* wl = reinterpret_cast<char*>(lds) + wave * (32 * RECB);
    float4 bq[NT][4];  // bias before the stores (vmcnt counts stores on gfx950)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bq[nt][q] = *reinterpret_cast<const float4*>(a.bias + ntile0 * 32 + nt * 32 + 8 * q + 4 * half);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int j = mt / TZ, z = mt % TZ;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = nt * 32 + 8 * q + 4 * half;
                const float4 b = bq[nt][q];
                float v0 = acc[mt][nt][4 * q + 0] + b.x;
                float v1 = acc[mt][nt][4 * q + 1] + b.y;
                float v2 = acc[mt][nt][4 * q + 2] + b.z;
                float v3 = acc[mt][nt][4 * q + 3] + b.w;
                v0 = v0 > 0.f ? v0 : v0 * a.slope;
                v1 = v1 > 0.f ? v1 : v1 * a.slope;
                v2 = v2 > 0.f ? v2 : v2 * a.slope;
                v3 = v3 > 0.f ? v3 : v3 * a.slope;
                store4<Tag>(wl, (size_t)(r * RECB) / ES + cl, v0, v1, v2, v3);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int PPV = RECB / 16;
        constexpr int ROUNDS = 32 * PPV / 64;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int p = k * 64 + lane;
            const int vv = p / PPV, part = p % PPV;
            const int pos = (wm * YXW + j) * 32 + vv;
            const int gz = z0 + z, gy = y0 + pos / TX, gx = x0 + pos % TX;
            const uint4 val = *reinterpret_cast<const uint4*>(wl + p * 16);
            if (gz < a.d && gy < a.h && gx < a.w) {
                const size_t vox = (((size_t)nb * a.d + gz) * a.h + gy) * a.w + gx;
                *reinterpret_cast<uint4*>(static_cast<char*>(a.dst) +
                                          (vox * a.cout + ntile0 * 32) * ES + part * 16) = val;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int YXW, int NT, int MINW, int PDG, bool WLDS = false, bool PADX = true>
static int launch_zcol(const ConvArgs& a, hipStream_t stream) {
    constexpr int NWG = WAVES_N * NT * 32;
    if (a.cout % NWG != 0) {
        set_error("conv: cout %d not a multiple of the %d-channel tile", a.cout, NWG);
        return EXASPIM_E_INVALID;
    }
    const int tz = (a.d + TZ - 1) / TZ, ty = (a.h + TY - 1) / TY, tx = (a.w + TX - 1) / TX;
    const long long blocks = (long long)tz * ty * tx * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        set_error("conv: grid of %lld blocks out of range", blocks);
        return EXASPIM_E_INVALID;
    }
    dim3 grid((unsigned)blocks, a.cout / NWG);
    conv3x3x3_zcol<Tag, TZ, TY, TX, WAVES_M, WAVES_N, YXW, NT, MINW, PDG, WLDS, PADX>
        <<<grid, WAVES_M * WAVES_N * 64, 0, stream>>>(a, tz, ty, tx);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

// ---- v5: persistent workgroups, one wave per SIMD ----------------------------
// For the 32-cout layers (53 % of the network's FLOPs). One 256-thread
// workgroup per CU walks a list of 4 x 8 x 32 tiles (1024 voxels; z-column
// mapping: wave w owns rows 2w, 2w+1 of every plane -> 8 accumulator tiles).
// Every (tile, chunk) step prefetches the NEXT step's halo pieces and weight
// fragments global -> VGPR at its start -- also across tile boundaries -- so the
// only exposed memory time is the very first step of a workgroup. A step's LDS
// work is (6 x 2 x 9 fragment reads + 27 weight reads) per wave for 216 MFMAs,
// about half of what the 512-voxel tiles need per FLOP, which is what bounds
// those. launch_bounds(256, 1): the wave may use the whole 512-register file.
template <typename Tag, int TY, int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void conv3x3x3_persist(ConvArgs a, int tiles_z, int tiles_y,
                                                            int tiles_x, int total_tiles) {
    constexpr int TZ = 4, TX = 32, YXW = TY / 4, NT = 1;
    constexpr int G = Tag::kG;
    constexpr int KC = 2 * G;
    constexpr int ES = 16 / G;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HV = HZ * HY * HX;
    constexpr int NTHREADS = 256;
    constexpr int MT = TZ * YXW;
    constexpr int NITEMS = (2 * HV + NTHREADS - 1) / NTHREADS;   // 16
    constexpr int WUNITS = 27 * 64;
    constexpr int WITEMS = (WUNITS + NTHREADS - 1) / NTHREADS;   // 7
    constexpr int RECB = NT * 32 * ES;
    constexpr int XUNITS = 2 * HV;
    constexpr int LDS_UNITS = XUNITS + WUNITS;

    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_UNITS];
    uint4* const wlds = lds + XUNITS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int r = lane & 31;

    // tile list: XCD k (workgroups b with b % 8 == k) takes a contiguous run of
    // tiles, dealt to its workgroups round-robin so neighbours run concurrently
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int wg_per_xcd = (nwg + 7 - xcd) >> 3;       // workgroups on this XCD
    const int run_lo = (int)((long long)total_tiles * xcd / 8);
    const int run_hi = (int)((long long)total_tiles * (xcd + 1) / 8);

    const int ntiles = a.cout >> 5;
    const int ntile0 = blockIdx.y;
    const int nchunks = (a.ca + a.cb) / KC;
    const size_t patch_vox = (size_t)a.d * a.h * a.w;

    int col[YXW];
#pragma unroll
    for (int j = 0; j < YXW; ++j) {
        const int p = (wave * YXW + j) * 32 + r;
        col[j] = (p / TX) * HX + (p % TX) + half * HV;
    }
    // Staging item i = tid + it * 256 = channel group (i >= HV) of halo voxel
    // (i mod HV). Tile-independent per item: its voxel offset relative to the
    // tile origin and a mask of the tile faces it lies beyond (bits 0-5: low/high
    // z, y, x; bit 6: past the end of the list; bit 7: second channel group).
    int rel[NITEMS], iflag[NITEMS];
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) {
        const int i = tid + it * NTHREADS;
        const int kg = i >= HV ? 1 : 0;
        const int hv = i - kg * HV;
        const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
        rel[it] = ((hz - 1) * a.h + (hy - 1)) * a.w + (hx - 1);
        iflag[it] = (hz == 0 ? 1 : 0) | (hz == HZ - 1 ? 2 : 0) | (hy == 0 ? 4 : 0) |
                    (hy == HY - 1 ? 8 : 0) | (hx == 0 ? 16 : 0) | (hx == HX - 1 ? 32 : 0) |
                    (i >= 2 * HV ? 64 : 0) | (kg << 7);
    }
    unsigned wvoff[WITEMS];  // byte offset of the thread's weight fragments inside a chunk
#pragma unroll
    for (int it = 0; it < WITEMS; ++it) {
        const int i = tid + it * NTHREADS;
        wvoff[it] = i < WUNITS ? (((i >> 6) * ntiles + ntile0) * 64 + (i & 63)) * 16u : kOutOfRange;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(a.weights, (size_t)nchunks * 27 * ntiles * 1024);

    uint4 stg[NITEMS + WITEMS];
    int tz0 = 0, ty0 = 0, tx0 = 0, tnb = 0;  // origin of the tile being prefetched

    auto decode = [&](int tile, int& z0, int& y0, int& x0, int& nb) {
        int t = tile;
        x0 = (t % tiles_x) * TX; t /= tiles_x;
        y0 = (t % tiles_y) * TY; t /= tiles_y;
        z0 = (t % tiles_z) * TZ; t /= tiles_z;
        nb = t;
    };
    auto stage_load = [&](int c) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const unsigned rowb = cs * ES;
        const __amdgpu_buffer_rsrc_t rsrc =
            make_rsrc(src + (size_t)tnb * patch_vox * rowb, patch_vox * rowb);
        // faces of the volume this tile touches (tiles are exact: w % 32 == 0 etc.)
        const int tflag = (tz0 == 0 ? 1 : 0) | (tz0 + TZ >= a.d ? 2 : 0) | (ty0 == 0 ? 4 : 0) |
                          (ty0 + TY >= a.h ? 8 : 0) | (tx0 == 0 ? 16 : 0) |
                          (tx0 + TX >= a.w ? 32 : 0) | 64;
        const int origin = (tz0 * a.h + ty0) * a.w + tx0;
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const unsigned voff = (iflag[it] & tflag)
                                      ? kOutOfRange
                                      : (unsigned)(origin + rel[it]) * rowb + ((iflag[it] >> 3) & 16u);
            stg[it] = buf_load16(rsrc, voff, ch0 * ES);
        }
#pragma unroll
        for (int it = 0; it < WITEMS; ++it)
            stg[NITEMS + it] = buf_load16(wrsrc, wvoff[it], c * 27 * ntiles * 1024);
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const int i = tid + it * NTHREADS;
            if (i < 2 * HV) lds[i] = stg[it];
        }
#pragma unroll
        for (int it = 0; it < WITEMS; ++it) {
            const int i = tid + it * NTHREADS;
            if (i < WUNITS) wlds[i] = stg[NITEMS + it];
        }
    };

    int tile = run_lo + jx;
    if (tile >= run_hi) return;
    // stagger the second workgroup of each CU so the two do not run their MFMA
    // and their epilogue phases in lockstep (a.debug = number of ~4 us sleeps)
    if (WGS_PER_CU > 1 && (int)blockIdx.x >= (int)gridDim.x / 2)
        for (int i = 0; i < a.debug; ++i) __builtin_amdgcn_s_sleep(127);
    decode(tile, tz0, ty0, tx0, tnb);
    stage_load(0);
    stage_store();
    __syncthreads();

    float4 bq[4];  // this lane's 16 bias values, loaded once (never between stores)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        bq[q] = *reinterpret_cast<const float4*>(a.bias + ntile0 * 32 + 8 * q + 4 * half);

    f32x16 acc[MT];
    int cz0 = tz0, cy0 = ty0, cx0 = tx0, cnb = tnb;  // tile being computed
    int c = 0;
    for (;;) {
        if (c == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        }
        // ---- prefetch the next step -------------------------------------------
        const bool last_chunk = c + 1 == nchunks;
        const int next_tile = last_chunk ? tile + wg_per_xcd : tile;
        const bool has_next = next_tile < run_hi;
        if (has_next) {
            if (last_chunk) decode(next_tile, tz0, ty0, tx0, tnb);
            stage_load(last_chunk ? 0 : c + 1);
        }
        // ---- this step's 216 MFMAs ----------------------------------------------
        // Software-pipelined over the 18 (in-plane tap g, column j) groups: the six
        // plane fragments (and, per tap, the three weight fragments) of group k+1
        // are issued before the 12 MFMAs of group k; the scheduling fences keep
        // hipcc from re-serialising read -> wait -> MFMA on one register set.
        uint4 xf[2][HZ];
        uint4 wf[2][3];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) wf[0][dz] = wlds[(dz * 9) * 64 + lane];
#pragma unroll
        for (int zin = 0; zin < HZ; ++zin) xf[0][zin] = lds[col[0] + zin * HY * HX];
#pragma unroll
        for (int k = 0; k < 9 * YXW; ++k) {
            const int g = k / YXW, j = k % YXW;
            if (k + 1 < 9 * YXW) {
                const int g1 = (k + 1) / YXW, j1 = (k + 1) % YXW;
                const int goff1 = (g1 / 3) * HX + g1 % 3;
                if (j1 == 0) {
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz)
                        wf[g1 & 1][dz] = wlds[(dz * 9 + g1) * 64 + lane];
                }
#pragma unroll
                for (int zin = 0; zin < HZ; ++zin)
                    xf[(k + 1) & 1][zin] = lds[col[j1] + zin * HY * HX + goff1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int zin = 0; zin < HZ; ++zin)
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int z = zin - dz;
                    if (z >= 0 && z < TZ) mma<Tag>(acc[j * TZ + z], wf[g & 1][dz], xf[k & 1][zin]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // all waves are done reading this step's LDS image
        if (last_chunk) {
            // ---- epilogue of the finished tile: bias + LeakyReLU, through LDS ------
            char* wl = reinterpret_cast<char*>(lds) + wave * (32 * RECB);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int j = mt / TZ, z = mt % TZ;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = 8 * q + 4 * half;
                    const float4 b = bq[q];
                    float v0 = acc[mt][4 * q + 0] + b.x;
                    float v1 = acc[mt][4 * q + 1] + b.y;
                    float v2 = acc[mt][4 * q + 2] + b.z;
                    float v3 = acc[mt][4 * q + 3] + b.w;
                    v0 = v0 > 0.f ? v0 : v0 * a.slope;
                    v1 = v1 > 0.f ? v1 : v1 * a.slope;
                    v2 = v2 > 0.f ? v2 : v2 * a.slope;
                    v3 = v3 > 0.f ? v3 : v3 * a.slope;
                    store4<Tag>(wl, (size_t)(r * RECB) / ES + cl, v0, v1, v2, v3);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                constexpr int PPV = RECB / 16;
                constexpr int ROUNDS = 32 * PPV / 64;
#pragma unroll
                for (int k = 0; k < ROUNDS; ++k) {
                    const int p = k * 64 + lane;
                    const int vv = p / PPV, part = p % PPV;
                    const int pos = (wave * YXW + j) * 32 + vv;
                    const int gz = cz0 + z, gy = cy0 + pos / TX, gx = cx0 + pos % TX;
                    const uint4 val = *reinterpret_cast<const uint4*>(wl + p * 16);
                    if (gz < a.d && gy < a.h && gx < a.w) {
                        const size_t vox = (((size_t)cnb * a.d + gz) * a.h + gy) * a.w + gx;
                        *reinterpret_cast<uint4*>(static_cast<char*>(a.dst) +
                                                  (vox * a.cout + ntile0 * 32) * ES + part * 16) = val;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();  // the transposition buffers alias the LDS image
        }
        if (!has_next) break;
        stage_store();
        __syncthreads();
        if (last_chunk) {
            tile = next_tile; c = 0;
            cz0 = tz0; cy0 = ty0; cx0 = tx0; cnb = tnb;
        } else {
            ++c;
        }
    }
}

template <typename Tag, int TY, int WGS_PER_CU>
static int launch_persist(const ConvArgs& a, hipStream_t stream) {
    const int tz = (a.d + 3) / 4, ty = (a.h + TY - 1) / TY, tx = (a.w + 31) / 32;
    const long long total = (long long)tz * ty * tx * a.n;
    if (total <= 0 || total > 0x7fffffffLL) {
        set_error("conv: %lld tiles out of range", total);
        return EXASPIM_E_INVALID;
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        EXA_CHECK_HIP(hipGetDevice(&dev));
        EXA_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount;
    }
    const int nwg = (int)(total < ncu * WGS_PER_CU ? total : ncu * WGS_PER_CU);
    dim3 grid(nwg, a.cout / 32);
    conv3x3x3_persist<Tag, TY, WGS_PER_CU><<<grid, 256, 0, stream>>>(a, tz, ty, tx, (int)total);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

// ---- host side: pick a tile configuration per layer -----------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int MT, int NT, int MINW, int PD = 3>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
    constexpr int NWG = WAVES_N * NT * 32;
    if (a.cout % NWG != 0) {
        set_error("conv: cout %d not a multiple of the %d-channel tile", a.cout, NWG);
        return EXASPIM_E_INVALID;
    }
    const int tz = cdiv(a.d, TZ), ty = cdiv(a.h, TY), tx = cdiv(a.w, TX);
    const long long blocks = (long long)tz * ty * tx * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        set_error("conv: grid of %lld blocks out of range", blocks);
        return EXASPIM_E_INVALID;
    }
    dim3 grid((unsigned)blocks, a.cout / NWG);
    static const int impl = getenv("EXASPIM_CONV_IMPL") ? atoi(getenv("EXASPIM_CONV_IMPL")) : 1;
    if (impl == 0)
        conv3x3x3_kernel<Tag, TZ, TY, TX, WAVES_M, WAVES_N, MT, NT, MINW, (PD > 3 ? 3 : PD)>
            <<<grid, WAVES_M * WAVES_N * 64, 0, stream>>>(a, tz, ty, tx);
    else
        conv3x3x3_t14<Tag, TZ, TY, TX, WAVES_M, WAVES_N, MT, NT, MINW, PD>
            <<<grid, WAVES_M * WAVES_N * 64, 0, stream>>>(a, tz, ty, tx);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

template <typename Tag>
static int launch_typed(const ConvArgs& a, hipStream_t stream) {
    // Widest x extent first: the tile shapes follow the 96/48/24/12/6 pyramid of
    // a 96^3 patch; any other size runs on the closest shape with masking.
    static const int l0_variant = getenv("EXASPIM_L0_VARIANT") ? atoi(getenv("EXASPIM_L0_VARIANT")) : 0;
    static const int l1_variant = getenv("EXASPIM_L1_VARIANT") ? atoi(getenv("EXASPIM_L1_VARIANT")) : 0;
    if (a.w >= 16 && a.w % 16 == 0) {
        // z-column tiles, weights shared through LDS, one 32-cout slice per workgroup
        if (a.cout % 64 != 0) {
            if (l0_variant == 1) return launch_cfg<Tag, 4, 8, 16, 4, 1, 4, 1, 2, 8>(a, stream);
            if (l0_variant == 2 && a.w % 32 == 0) return launch_persist<Tag, 8, 1>(a, stream);
            if (l0_variant == 3 && a.w % 32 == 0) return launch_persist<Tag, 4, 2>(a, stream);
            return launch_zcol<Tag, 4, 8, 16, 4, 1, 1, 1, 2, 1, true, false>(a, stream);
        }
        if (l1_variant == 1) return launch_zcol<Tag, 4, 8, 16, 4, 1, 1, 1, 2, 1, true, false>(a, stream);
        return launch_cfg<Tag, 4, 8, 16, 4, 1, 4, 2, 2>(a, stream);
    }
    if (a.w > 12) {
        if (a.cout % 64 == 0) return launch_cfg<Tag, 4, 4, 24, 4, 1, 3, 2, 2>(a, stream);
        return launch_cfg<Tag, 4, 4, 24, 4, 1, 3, 1, 2>(a, stream);
    }
    if (a.w > 6) {
        if (a.cout % 128 == 0) return launch_cfg<Tag, 4, 4, 12, 2, 2, 3, 2, 2>(a, stream);
        if (a.cout % 64 == 0) return launch_cfg<Tag, 4, 4, 12, 2, 2, 3, 1, 2>(a, stream);
        return launch_cfg<Tag, 4, 4, 12, 2, 1, 3, 1, 2>(a, stream);
    }
    if (a.cout % 64 == 0) return launch_cfg<Tag, 6, 6, 6, 2, 2, 4, 1, 2>(a, stream);
    return launch_cfg<Tag, 6, 6, 6, 2, 1, 4, 1, 2>(a, stream);
}

int launch_conv3x3x3(int dtype, const ConvArgs& a, hipStream_t stream) {
    const int kc = dtype == EXASPIM_DT_F32 ? 8 : 16;
    EXA_CHECK_ARG(a.ca % kc == 0 && a.cb % kc == 0 && a.cout % 32 == 0 && a.ca > 0,
                  "conv: channels (%d,%d)->%d not padded", a.ca, a.cb, a.cout);
    EXA_CHECK_ARG(a.n > 0 && a.d > 0 && a.h > 0 && a.w > 0, "conv: empty input");
    {   // the LDS-DMA staging addresses one patch of one source with 32-bit offsets
        const unsigned long long rec = (unsigned long long)a.d * a.h * a.w *
                                       (a.ca > a.cb ? a.ca : a.cb) * (dtype == EXASPIM_DT_F32 ? 4 : 2);
        EXA_CHECK_ARG(rec < 0x80000000ULL, "conv: one patch of one source is %llu bytes (>= 2 GiB)", rec);
    }
    static const int debug = getenv("EXASPIM_CONV_DEBUG") ? atoi(getenv("EXASPIM_CONV_DEBUG")) : 0;
    ConvArgs b = a;
    b.debug = debug;
    switch (dtype) {
        case EXASPIM_DT_F32: return launch_typed<F32Tag>(b, stream);
        case EXASPIM_DT_BF16: return launch_typed<BF16Tag>(b, stream);
        case EXASPIM_DT_F16: return launch_typed<F16Tag>(b, stream);
    }
    set_error("conv: unknown dtype %d", dtype);
    return EXASPIM_E_INVALID;
}

}  // namespace exaspim
