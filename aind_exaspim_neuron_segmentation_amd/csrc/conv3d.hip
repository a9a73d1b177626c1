// 3x3x3 convolution (+ folded BatchNorm bias + LeakyReLU) as an implicit GEMM
// on the gfx950 matrix cores. Replaces every nn.Conv3d(k=3,p=1) ->
// BatchNorm3d(eval) -> LeakyReLU(0.01) triple of the reference's DoubleConv
// (machine_learning/unet3d.py:142-149) except inc.0 (Cin = 1, layers.hip), and
// torch.cat([skip, up], dim=1) (unet3d.py:288) by reading two sources.
//
// Layout. Activations are channels-last (N, D, H, W, C) with C padded to 32.
// One workgroup owns a TZ x TY x TX block of output voxels of one patch and a
// slice of 32 * NT output channels. The K dimension (27 taps x Cin) is walked in
// chunks of 32 bytes of input channels (8 x f32 / 16 x 16-bit): per chunk the
// (TZ+2)(TY+2)(TX+2) halo block sits in LDS as two planes of 16-byte channel
// groups, [group][halo voxel]; the conv's zero padding comes from range-checked
// buffer loads that return zeros outside the patch. A wave's MFMA B operand
// (activations, voxel on the lane) is ONE ds_read_b128 per (tap, 32 voxels):
// lanes 0-31 read group 0, lanes 32-63 group 1. The A operand is a weight
// fragment in the order plan.cpp packs (1 KiB per wave-instruction).
//
// D = W(32 cout x K) * X(K x 32 voxels): the accumulator keeps the voxel on the
// lane and 4-channel runs in registers; the epilogue (bias, LeakyReLU, convert)
// goes through LDS so every global store is a whole 16-byte piece of
// consecutive voxel records.
//
// f32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 4 per chunk-tap),
// bf16/f16 use v_mfma_f32_32x32x16_{bf16,f16} (one per chunk-tap).
//
// Two kernels share this scheme:
//   conv3x3x3_zcol  32-cout slices (53 % of the FLOPs): wave = columns of the
//                   tile, one LDS read feeds the three dz taps, the chunk's
//                   weights are shared through LDS;
//   conv3x3x3_t14   wider slices and the small pyramid levels: weights stream
//                   from L2 through a register ring.
// Both prefetch the next chunk global -> VGPR under the current chunk's MFMAs.

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

// ---- conv3x3x3_t14: register-staged prefetch (async-STAGE split), deeper operand
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
    constexpr int RECP = RECB + 16;                 // padded LDS stride (8-way -> 2-way conflicts)
    constexpr int EPI_UNITS = NWAVES * 32 * RECP / 16;
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
    // 16-wide rows: second row of a 32-voxel group in rotated x order, x = (i - HX)
    // mod 16, so its lanes use the bank slots the first row leaves free (see zcol)
    const int r = (TX == 16 && (lane & 16)) ? 16 + (((lane & 15) - HX) & 15) : (lane & 31);

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

    // accumulators start from the folded bias: register 4q+k of a lane is channel
    // 8q + 4*half + k of its slice (no bias pass in the epilogue)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias + (ntile0 + nt) * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][nt][4 * q + 0] = b.x; acc[mt][nt][4 * q + 1] = b.y;
                acc[mt][nt][4 * q + 2] = b.z; acc[mt][nt][4 * q + 3] = b.w;
            }
        }

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
    char* wl = reinterpret_cast<char*>(lds) + wave * (32 * RECP);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = nt * 32 + 8 * q + 4 * half;  // channel inside the slice
                // LeakyReLU with 0 <= slope <= 1 is max(v, slope * v)
                float v0 = acc[mt][nt][4 * q + 0], v1 = acc[mt][nt][4 * q + 1];
                float v2 = acc[mt][nt][4 * q + 2], v3 = acc[mt][nt][4 * q + 3];
                v0 = fmaxf(v0, v0 * a.slope);
                v1 = fmaxf(v1, v1 * a.slope);
                v2 = fmaxf(v2, v2 * a.slope);
                v3 = fmaxf(v3, v3 * a.slope);
                store4<Tag>(wl, (size_t)(r * RECP) / ES + cl, v0, v1, v2, v3);
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
            const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + part * 16);
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

// Phase stamps for tools/conv_trace.hip (compiled out of the library).
#ifdef EXASPIM_TRACE
#define EXA_TRACE(ev)                                                                          \
    do {                                                                                       \
        if (a.trace && lane == 0)                                                              \
            a.trace[((size_t)blockIdx.x * NWAVES + wave) * 16 + (ev)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define EXA_TRACE(ev) do { } while (0)
#endif

// ---- conv3x3x3_zcol: z-column tiles ---------------------------------------------------
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
template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int YXW, int NT, int MINW, int PDG, bool WLDS, bool PADX = true, int HEAD = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv3x3x3_zcol(
    ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int G = Tag::kG;
    constexpr int KC = 2 * G;
    constexpr int ES = 16 / G;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HXP = (TX == 32 || !PADX) ? HX : (HX + 15) / 16 * 16 + (TX % 16);  // row stride (slots)
    constexpr int HVP = HZ * HY * HXP;     // slots per channel group
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int YXT = TY * TX / 32;      // 32-voxel (y, x) groups per plane
    constexpr int MT = TZ * YXW;
    constexpr int NCOL = HY * HX;          // (y, x) columns of the halo block
    constexpr int NITEMS = 2 * HZ;         // staging pieces per thread: one column, 2 groups x HZ planes
    constexpr int RECB = NT * 32 * ES;
    constexpr int RECP = RECB + 16;  // padded LDS stride of the output transposition
    constexpr int EPI_UNITS = NWAVES * 32 * RECP / 16;
    constexpr int WUNITS = WLDS ? 27 * NT * 64 : 0;           // weight fragments in LDS
    constexpr int WITEMS = (WUNITS + NTHREADS - 1) / NTHREADS;
    constexpr int XUNITS = 2 * HVP > EPI_UNITS ? 2 * HVP : EPI_UNITS;
    constexpr int LDS_UNITS = XUNITS + WUNITS;
    static_assert(TY * TX % 32 == 0 && YXT == WAVES_M * YXW, "plane not covered by the waves");
    static_assert(!WLDS || WAVES_N == 1, "LDS weights assume one cout slice per workgroup");
    static_assert(NCOL <= NTHREADS, "one halo column per thread");

    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_UNITS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int half = lane >> 5;
    // Voxel of the 32-group this lane works on. With 16-wide rows the group is two
    // rows whose LDS slots differ by HXP; taking the second row's x in rotated
    // order, x = (i - HXP) mod 16, puts lane 16+i on bank slot i (mod 16), the
    // complement of what its ds_read_b128 lane group already uses: no conflicts.
    const int r = (TX == 16 && (lane & 16)) ? 16 + (((lane & 15) - HXP) & 15) : (lane & 31);

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

    // Column staging: thread t < NCOL owns halo column (hy, hx) = (t / HX, t % HX)
    // and moves its 2 x HZ pieces. All pieces of a thread share ONE vector offset
    // (the column inside a plane); plane, channel group and chunk go into the
    // scalar offset of the buffer load, so staging costs no per-piece VALU work.
    const size_t patch_vox = (size_t)a.d * a.h * a.w;
    const int plane_vox = a.h * a.w;
    const bool colok = tid < NCOL;
    const int chy = tid / HX, chx = tid % HX;
    const int cgy = y0 + chy - 1, cgx = x0 + chx - 1;
    const bool col_in = colok && (unsigned)cgy < (unsigned)a.h && (unsigned)cgx < (unsigned)a.w;
    const int colvox = cgy * a.w + cgx;        // voxel of the column inside a plane
    const int colslot = chy * HXP + chx;       // its LDS slot inside a plane
    // weight fragments: piece i = tid + it * NTHREADS is element (i & 63) of tap i >> 6
    const unsigned wvoff = (((tid >> 6) * ntiles) * 64 + (tid & 63)) * 16u;

    // accumulators start from the folded bias: register 4q+k of a lane is channel
    // 8q + 4*half + k of its slice (no bias pass in the epilogue)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias + (ntile0 + nt) * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][nt][4 * q + 0] = b.x; acc[mt][nt][4 * q + 1] = b.y;
                acc[mt][nt][4 * q + 2] = b.z; acc[mt][nt][4 * q + 3] = b.w;
            }
        }

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
        const unsigned voff = col_in ? (unsigned)colvox * rowb : kOutOfRange;
        const unsigned planeb = (unsigned)plane_vox * rowb;
#pragma unroll
        for (int hz = 0; hz < HZ; ++hz) {
            const int gz = z0 + hz - 1;           // wave-uniform
            const bool zin = (unsigned)gz < (unsigned)a.d;
#pragma unroll
            for (int kg = 0; kg < 2; ++kg)
                stg[kg * HZ + hz] = zin ? buf_load16(rsrc, voff, (unsigned)gz * planeb + ch0 * ES + kg * 16)
                                        : make_uint4(0, 0, 0, 0);
        }
        if (WLDS) {
            const __amdgpu_buffer_rsrc_t wrsrc =
                make_rsrc(a.weights, (size_t)nchunks * 27 * ntiles * 1024);
#pragma unroll
            for (int it = 0; it < WITEMS; ++it) {
                // taps it * NWAVES + wave; the last round covers taps < 27 only
                const bool live = it * NWAVES + wave < 27;
                stg[NITEMS + it] = live ? buf_load16(wrsrc, wvoff,
                                                     ((c * 27 + it * NWAVES) * ntiles + ntile0) * 1024)
                                        : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto stage_store = [&]() {
        if (colok) {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                for (int hz = 0; hz < HZ; ++hz)
                    lds[kg * HVP + hz * HY * HXP + colslot] = stg[kg * HZ + hz];
        }
        if (WLDS) {
#pragma unroll
            for (int it = 0; it < WITEMS; ++it) {
                const int i = tid + it * NTHREADS;
                if (i < WUNITS) wlds[i] = stg[NITEMS + it];
            }
        }
    };

#ifdef EXASPIM_TRACE
    if (a.trace && lane == 0)
        a.trace[((size_t)blockIdx.x * NWAVES + wave) * 16 + 15] =
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
            (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    EXA_TRACE(0);
    stage_load(0);
    EXA_TRACE(1);
    stage_store();
    __syncthreads();
    EXA_TRACE(2);

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
        if (c < 4) EXA_TRACE(3 + 3 * c);
        __syncthreads();
        if (c < 4) EXA_TRACE(4 + 3 * c);
        if (more) {
            stage_store();
            __syncthreads();
            if (c < 3) EXA_TRACE(5 + 3 * c);
        }
    }

    if (HEAD > 0) {
        // ---- fused head: OutConv 1x1x1 (+ sigmoid) on the accumulators -----------
        // lane (voxel r, half h) holds channels 8q + 4h + j of its voxel: a 16-term
        // partial dot product per output, completed by the other half-wave.
        static_assert(HEAD == 0 || NT == 1, "the fused head needs the whole 32-channel record");
        float hw[HEAD > 0 ? HEAD : 1][16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = 0; o < HEAD; ++o) {
                const float4 t = *reinterpret_cast<const float4*>(a.head_w + o * 32 + 8 * q + 4 * half);
                hw[o][4 * q + 0] = t.x; hw[o][4 * q + 1] = t.y;
                hw[o][4 * q + 2] = t.z; hw[o][4 * q + 3] = t.w;
            }
        }
        float hb[HEAD > 0 ? HEAD : 1];
#pragma unroll
        for (int o = 0; o < HEAD; ++o) hb[o] = a.head_b[o];
        const size_t plane = (size_t)a.h * a.w;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int j = mt / TZ, z = mt % TZ;
            float part[HEAD > 0 ? HEAD : 1];
#pragma unroll
            for (int o = 0; o < HEAD; ++o) part[o] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = acc[mt][0][4 * q + k];
                    v = fmaxf(v, v * a.slope);
#pragma unroll
                    for (int o = 0; o < HEAD; ++o) part[o] = fmaf(v, hw[o][4 * q + k], part[o]);
                }
            }
            const int pos = (wm * YXW + j) * 32 + r;
            const int gz = z0 + z, gy = y0 + pos / TX, gx = x0 + pos % TX;
            const bool ok = gz < a.d && gy < a.h && gx < a.w;
#pragma unroll
            for (int o = 0; o < HEAD; ++o) {
                float t = part[o] + __shfl_xor(part[o], 32) + hb[o];
                if (a.head_sigmoid) t = 1.f / (1.f + expf(-t));
                // outputs are dealt to the two half-waves so both store
                if (ok && (o & 1) == half)
                    a.head_out[(((size_t)nb * HEAD + o) * a.d + gz) * plane + (size_t)gy * a.w + gx] = t;
            }
        }
        return;
    }

    // ---- epilogue: bias + LeakyReLU, transposed through LDS ------------------
    char* wl = reinterpret_cast<char*>(lds) + wave * (32 * RECP);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int j = mt / TZ, z = mt % TZ;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = nt * 32 + 8 * q + 4 * half;
                // LeakyReLU with 0 <= slope <= 1 is max(v, slope * v)
                float v0 = acc[mt][nt][4 * q + 0], v1 = acc[mt][nt][4 * q + 1];
                float v2 = acc[mt][nt][4 * q + 2], v3 = acc[mt][nt][4 * q + 3];
                v0 = fmaxf(v0, v0 * a.slope);
                v1 = fmaxf(v1, v1 * a.slope);
                v2 = fmaxf(v2, v2 * a.slope);
                v3 = fmaxf(v3, v3 * a.slope);
                store4<Tag>(wl, (size_t)(r * RECP) / ES + cl, v0, v1, v2, v3);
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
            const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + part * 16);
            if (gz < a.d && gy < a.h && gx < a.w) {
                const size_t vox = (((size_t)nb * a.d + gz) * a.h + gy) * a.w + gx;
                *reinterpret_cast<uint4*>(static_cast<char*>(a.dst) +
                                          (vox * a.cout + ntile0 * 32) * ES + part * 16) = val;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    EXA_TRACE(14);
}

template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int YXW, int NT, int MINW, int PDG, bool WLDS = false, bool PADX = true, int HEAD = 0>
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
    conv3x3x3_zcol<Tag, TZ, TY, TX, WAVES_M, WAVES_N, YXW, NT, MINW, PDG, WLDS, PADX, HEAD>
        <<<grid, WAVES_M * WAVES_N * 64, 0, stream>>>(a, tz, ty, tx);
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
    conv3x3x3_t14<Tag, TZ, TY, TX, WAVES_M, WAVES_N, MT, NT, MINW, PD>
        <<<grid, WAVES_M * WAVES_N * 64, 0, stream>>>(a, tz, ty, tx);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

// 32-cout slice on z-column tiles of TZ planes, with or without the fused head
template <typename Tag, int TZ>
static int launch_zcol_head(const ConvArgs& a, hipStream_t stream) {
    if (a.head_out && a.cout == 32) {
        switch (a.head_oc) {
            case 1: return launch_zcol<Tag, TZ, 8, 16, 4, 1, 1, 1, 2, 1, true, false, 1>(a, stream);
            case 2: return launch_zcol<Tag, TZ, 8, 16, 4, 1, 1, 1, 2, 1, true, false, 2>(a, stream);
            case 3: return launch_zcol<Tag, TZ, 8, 16, 4, 1, 1, 1, 2, 1, true, false, 3>(a, stream);
            case 4: return launch_zcol<Tag, TZ, 8, 16, 4, 1, 1, 1, 2, 1, true, false, 4>(a, stream);
        }
    }
    return launch_zcol<Tag, TZ, 8, 16, 4, 1, 1, 1, 2, 1, true, false>(a, stream);
}

template <typename Tag>
static int launch_typed(const ConvArgs& a, hipStream_t stream) {
    // Widest x extent first: the tile shapes follow the 96/48/24/12/6 pyramid of
    // a 96^3 patch; any other size runs on the closest shape with masking.
    if (a.w >= 16 && a.w % 16 == 0) {
        // 32-cout slices: z-column tiles with the chunk's weights shared through LDS
        if (a.cout % 64 != 0) {
            // 6-plane tiles when the depth divides (96, 48, 24): more dz reuse per LDS read
            if (a.d % 6 == 0) return launch_zcol_head<Tag, 6>(a, stream);
            return launch_zcol_head<Tag, 4>(a, stream);
        }
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

bool conv_can_fuse_head(int cout, int w, int head_oc) {
    return cout == 32 && w >= 16 && w % 16 == 0 && head_oc >= 1 && head_oc <= 4;
}

int launch_conv3x3x3(int dtype, const ConvArgs& a, hipStream_t stream) {
    const int kc = dtype == EXASPIM_DT_F32 ? 8 : 16;
    EXA_CHECK_ARG(a.ca % kc == 0 && a.cb % kc == 0 && a.cout % 32 == 0 && a.ca > 0,
                  "conv: channels (%d,%d)->%d not padded", a.ca, a.cb, a.cout);
    EXA_CHECK_ARG(a.n > 0 && a.d > 0 && a.h > 0 && a.w > 0, "conv: empty input");
    EXA_CHECK_ARG(a.slope >= 0.f && a.slope <= 1.f, "conv: LeakyReLU slope %g outside [0, 1]", a.slope);
    EXA_CHECK_ARG(!a.head_out || conv_can_fuse_head(a.cout, a.w, a.head_oc),
                  "conv: fused head needs cout 32, w %% 16 == 0, 1..4 outputs");
    {   // the LDS-DMA staging addresses one patch of one source with 32-bit offsets
        const unsigned long long rec = (unsigned long long)a.d * a.h * a.w *
                                       (a.ca > a.cb ? a.ca : a.cb) * (dtype == EXASPIM_DT_F32 ? 4 : 2);
        EXA_CHECK_ARG(rec < 0x80000000ULL, "conv: one patch of one source is %llu bytes (>= 2 GiB)", rec);
    }
    switch (dtype) {
        case EXASPIM_DT_F32: return launch_typed<F32Tag>(a, stream);
        case EXASPIM_DT_BF16: return launch_typed<BF16Tag>(a, stream);
        case EXASPIM_DT_F16: return launch_typed<F16Tag>(a, stream);
    }
    set_error("conv: unknown dtype %d", dtype);
    return EXASPIM_E_INVALID;
}

}  // namespace exaspim
