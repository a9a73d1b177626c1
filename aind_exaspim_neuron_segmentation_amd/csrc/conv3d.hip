// 3x3x3 convolution (+ folded BatchNorm bias + LeakyReLU) as an implicit GEMM
// on the gfx950 matrix cores. Replaces every nn.Conv3d(k=3,p=1) ->
// BatchNorm3d(eval) -> LeakyReLU(0.01) triple of the reference's DoubleConv
// (machine_learning/unet3d.py:142-149) except inc.0 (Cin = 1, layers.hip), and
// torch.cat([skip, up], dim=1) (unet3d.py:288) by reading two sources.
//
// Layout. Activations are blocked channels-last: the (padded) channels are cut
// into chunks of 32 bytes (8 x f32 / 16 x 16-bit = the K of one MFMA step) and
// every chunk is its own plane, (N, C/chunk, D, H, W, 32 B). A row of voxels of
// one chunk is contiguous, which is what both sides of the kernel want: the K
// dimension (27 taps x Cin) is walked chunk by chunk, and the texture addresser
// handles a load instruction per 64-byte segment, so staging whole halo rows of
// one chunk costs a quarter of gathering one 16-byte piece per voxel record.
// One workgroup owns a TZ x TY x TX block of output voxels of one patch and a
// slice of 32 * NT output channels. Per chunk the (TZ+2)(TY+2)(TX+2) halo block
// sits in LDS as two planes of 16-byte channel groups, [group][halo voxel]; the
// conv's zero padding comes from range-checked buffer loads that return zeros
// outside the patch. A wave's MFMA B operand (activations, voxel on the lane) is
// ONE ds_read_b128 per (tap, 32 voxels): lanes 0-31 read group 0, lanes 32-63
// group 1. The A operand is a weight fragment in the order plan.cpp packs
// (1 KiB per wave-instruction).
//
// D = W(32 cout x K) * X(K x 32 voxels): the accumulator keeps the voxel on the
// lane and 4-channel runs in registers; the epilogue (LeakyReLU, convert; the
// folded bias is the accumulators' initial value) goes through LDS so every
// global store instruction writes 32 whole voxel records of one chunk plane.
//
// f32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 4 per chunk-tap),
// bf16/f16 use v_mfma_f32_32x32x16_{bf16,f16} (one per chunk-tap).
//
// Two kernels share this scheme:
//   conv3x3x3_zpipe 32-cout slices (53 % of the FLOPs): wave = a column of the
//                   tile, one LDS read feeds the three dz taps, the chunk's
//                   weights are shared through LDS, operand reads run a fixed
//                   distance ahead of the MFMAs;
//   conv3x3x3_t14   wider slices and the small pyramid levels: weights stream
//                   from L2 through a register ring.
// Both prefetch the next chunk global -> VGPR under the current chunk's MFMAs.

#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace exaspim {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct F32Tag { static constexpr int kG = 4; static constexpr int kCode = EXASPIM_DT_F32; };
struct BF16Tag { static constexpr int kG = 8; static constexpr int kCode = EXASPIM_DT_BF16; };
struct F16Tag { static constexpr int kG = 8; static constexpr int kCode = EXASPIM_DT_F16; };

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

// In-place form for the z-column kernel: destination tied to the addend ("+v"). Left to the
// register allocator, many MFMAs of the unrolled tap loop got a destination different from
// their addend (both accumulator copies live for a while) and the kernel, already at its
// 256 registers, spilled; a spilled value comes back through a scratch load whose wait also
// waits for every prefetch load and store still in flight.
template <typename Tag>
__device__ __forceinline__ void mma_inplace(f32x16& acc, const uint4& wf, const uint4& xf) { mma<Tag>(acc, wf, xf); }
template <>
__device__ __forceinline__ void mma_inplace<BF16Tag>(f32x16& acc, const uint4& wf, const uint4& xf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                 : "+v"(acc) : "v"(__builtin_bit_cast(u32x4_t, wf)), "v"(__builtin_bit_cast(u32x4_t, xf)));
}
template <>
__device__ __forceinline__ void mma_inplace<F16Tag>(f32x16& acc, const uint4& wf, const uint4& xf) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0"
                 : "+v"(acc) : "v"(__builtin_bit_cast(u32x4_t, wf)), "v"(__builtin_bit_cast(u32x4_t, xf)));
}

// LeakyReLU with 0 <= slope <= 1 is max(v, slope * v): one multiply and one bare v_max_f32
// (fmaxf would put a canonicalising v_max in front of it)
__device__ __forceinline__ float leaky(float v, float slope) {
    const float sv = v * slope;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(sv));
    return r;
}

// A copy of a value the compiler cannot see through. The persistent z-column kernel sits at its
// register ceiling; hipcc hoists every lane-derived constant of the per-tile prologue and of the
// epilogue (LDS addresses of the bias, row / column of the lane, ...) out of the tile loop and
// then SPILLS them: each came back through a scratch_load whose s_waitcnt vmcnt(0) also waited
// for the previous tile's output stores and the prefetch in flight (four serialised round trips
// at every tile top, three in every epilogue). Deriving such values from an opaque copy of the
// lane index (fresh_lane) inside the loop makes them a few VALU instructions per tile instead.
// lane index (= threadIdx.x & 63 for the 1-D workgroups here) from the hardware, two VALU
// instructions without any input register; volatile, so never hoisted and never kept
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// element-wise maximum of two 16-byte channel groups in the storage type (exact: the
// inputs are already rounded, the larger one is returned bit for bit)
template <typename Tag>
__device__ __forceinline__ uint4 max16(const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ uint4 max16<F32Tag>(const uint4& a, const uint4& b) {
    return make_uint4(__float_as_uint(fmaxf(__uint_as_float(a.x), __uint_as_float(b.x))),
                      __float_as_uint(fmaxf(__uint_as_float(a.y), __uint_as_float(b.y))),
                      __float_as_uint(fmaxf(__uint_as_float(a.z), __uint_as_float(b.z))),
                      __float_as_uint(fmaxf(__uint_as_float(a.w), __uint_as_float(b.w))));
}
// bf16 (and f16) order like sign-magnitude integers: flipping the magnitude bits of negative
// values, x ^ ((x >> 15) & 0x7fff) per 16-bit half, makes them order like two's-complement
// shorts, so a packed integer maximum picks the larger float; the map is its own inverse.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned key16x2(unsigned x) {
    const s16x2 m = __builtin_bit_cast(s16x2, x) >> (short)15;   // 0 or -1 per half
    return x ^ (__builtin_bit_cast(unsigned, m) & 0x7fff7fffu);
}
__device__ __forceinline__ unsigned maxkey16x2(unsigned ka, unsigned kb) {
    return __builtin_bit_cast(
        unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ka), __builtin_bit_cast(s16x2, kb)));
}
__device__ __forceinline__ uint4 key16(const uint4& v) {
    return make_uint4(key16x2(v.x), key16x2(v.y), key16x2(v.z), key16x2(v.w));
}
__device__ __forceinline__ uint4 maxkey16(const uint4& a, const uint4& b) {
    return make_uint4(maxkey16x2(a.x, b.x), maxkey16x2(a.y, b.y), maxkey16x2(a.z, b.z), maxkey16x2(a.w, b.w));
}
template <>
__device__ __forceinline__ uint4 max16<BF16Tag>(const uint4& a, const uint4& b) {
    return key16(maxkey16(key16(a), key16(b)));
}
template <>
__device__ __forceinline__ uint4 max16<F16Tag>(const uint4& a, const uint4& b) {
    return key16(maxkey16(key16(a), key16(b)));
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
    // saturate to the largest finite half: an activation beyond +-65504 is stored as
    // +-65504 instead of +-inf (one v_med3_f32 per value, epilogue only)
    a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
    c = __builtin_amdgcn_fmed3f(c, -65504.f, 65504.f);
    d = __builtin_amdgcn_fmed3f(d, -65504.f, 65504.f);
    f16x4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    *reinterpret_cast<f16x4*>(static_cast<_Float16*>(dst) + off) = v;
}

// the same four channels packed into 8 bytes (16-bit storage types), for stores straight from
// registers
template <typename Tag>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d);
template <>
__device__ __forceinline__ uint2 pack4<BF16Tag>(float a, float b, float c, float d) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return __builtin_bit_cast(uint2, v);
}
template <>
__device__ __forceinline__ uint2 pack4<F16Tag>(float a, float b, float c, float d) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);   // saturating, like store4
    b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
    c = __builtin_amdgcn_fmed3f(c, -65504.f, 65504.f);
    d = __builtin_amdgcn_fmed3f(d, -65504.f, 65504.f);
    const f16x4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    return __builtin_bit_cast(uint2, v);
}
template <>
__device__ __forceinline__ uint2 pack4<F32Tag>(float, float, float, float) { return make_uint2(0u, 0u); }   // (unused)

// Tap loops run at a raised wave priority (s_setprio): a CU holds two workgroups, and while one is in
// its prologue / staging / epilogue (VALU, LDS writes, stores) the other one's MFMA issue should not
// queue behind it. Measured inside 512^3 steps (us per launch, two alternating repeats, r03): levels
// 0 / 1 / 2 / 3 of the z-column kernel: inc.3 803 / 786 / 783 / 785, up4.0 803 / 788 / 788 / 789,
// up4.3 499 / 486 / 484 / 486, up3.3 171 / 166 / 166 / 166; level 2 in conv3x3x3_t14 as well: the
// 17 convolutions sum to 4617 instead of 4644 us per batch. 16-bit types only: with float32 operands
// (four 16-pass MFMAs per chunk-tap) the same hint makes conv3x3x3_t14 8 - 30 % SLOWER (512^3, batch 8:
// down2.0 458 -> 608 us, up2.0 1808 -> 2439, the 17 convolutions 19.7 -> 21.5 ms per batch) and leaves
// the z-column kernel where it was.
#ifndef EXASPIM_SETPRIO
#define EXASPIM_SETPRIO 2   // conv3x3x3_zpipe (0 = off)
#endif
#ifndef EXASPIM_SETPRIO_T14
#define EXASPIM_SETPRIO_T14 2   // conv3x3x3_t14
#endif
#ifndef EXASPIM_BUFFER_STORES
#define EXASPIM_BUFFER_STORES 7   // z-column kernel epilogues: branch-free range-checked buffer stores (1 direct, 2 head, 4 transposed + pool)
#endif
#ifndef EXASPIM_SETTLE_FIRST
#define EXASPIM_SETTLE_FIRST 1    // z-column kernel: wait for the next tile's prefetch before the epilogue's stores
#endif
#ifndef EXASPIM_STAGE_FIRST
#define EXASPIM_STAGE_FIRST 0   // 1: next tile's first chunk goes to LDS before the epilogue's stores (A/B aid)
#endif
#ifndef EXASPIM_DIRECT_EPILOGUE
#define EXASPIM_DIRECT_EPILOGUE 1   // 0: every epilogue goes through LDS (measurement aid)
#endif
// Phase stamps for tools/conv_trace.hip (compiled out of the library).
#ifndef EXASPIM_ABLATE
#define EXASPIM_ABLATE 0   // tools only: 1 = no prefetch loads, 2 = no output stores, 4 = no LDS staging writes
#endif
#ifdef EXASPIM_TRACE
#define EXA_TRACE(ev)                                                                          \
    do {                                                                                       \
        if (a.trace && lane == 0)                                                              \
            a.trace[trace_rec + (ev)] = __builtin_readcyclecounter();                          \
    } while (0)
#else
#define EXA_TRACE(ev) do { } while (0)
#endif

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
// ZORD: walk the 27 taps in the z-column kernel's order (in-plane tap outermost, dz
// innermost) instead of dz-major, so that a voxel gets the same bits from either kernel
// (the thin remainders of a region next to z-column tiles).
// DMA: the halo image is double-buffered and filled by LDS-DMA (buffer_load ... lds, 1 KiB per
// wave-instruction = 64 consecutive slots of one channel-group plane): the next chunk lands in
// the other buffer while this one is read, nothing is staged through registers, no ds_write,
// one barrier per chunk instead of two. The DMA is issued as inline assembly -- through the
// builtin hipcc puts a vmcnt(0) wait in front of every later LDS read, because it cannot see
// that they touch the other buffer -- and waited for explicitly before the chunk's barrier.
// POOL: the epilogue also writes the layer's MaxPool3d(2) (the input of the next Down block,
// unet3d.py:194-196) to a.pool_dst: every wave parks all its output groups in LDS, and after one
// workgroup barrier any thread can take the maximum over a 2 x 2 x 2 block of the tile (planes z
// and z + 1 belong to different waves). The skip tensor is not read again and the separate
// max-pool launch disappears. Same bits as maxpool2_kernel: the maximum of stored values.
template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int MT, int NT, int MINW, int PD,
          bool ZORD = false, bool DMA = false, bool POOL = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv3x3x3_t14(
    ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int G = Tag::kG;
    constexpr int KC = 2 * G;
    constexpr int ES = 16 / G;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    // Row stride of the LDS image (slots). A 32-voxel group of a 24-wide tile is a
    // row tail plus a row head; with a stride of 8 (mod 16) slots the two pieces fall
    // on complementary banks for every ds_read_b128 lane group (PMC: bank-conflict
    // cycles 50 % -> 17 % of the LDS-active cycles, which drop by 39 %; the launch time
    // does not move, LDS is not what limits this kernel). Padding is never touched.
    // (12-wide rows: a ds_read_b128 lane group of 16 voxels always wraps a 12-voxel row, and with the dense
    // row stride of 14 slots the two pieces share bank slots -- the counters show 50 % conflict cycles at
    // the 12^3 level. A row stride of 28 slots (= 12 mod 16, conflict-free for all 27 taps by enumeration,
    // channel groups 8 slots apart mod 16 for the staging writes) was measured in round 3 and, like round
    // 2's attempt, is SLOWER: down3.0 58 -> 66 us, down3.3 107 -> 127, up1.0 204 -> 243 per batch -- twice
    // the LDS image and a wider staging scatter cost more than the conflicts. Dense rows stay.)
    constexpr int HXS = TX == 24 ? 40 : HX;
    constexpr int PLS = HY * HXS;                   // plane stride
    constexpr int HVR = HZ * PLS;                   // slots of a channel-group plane that hold voxels
    // plane stride: with DMA a plane is written in whole 64-slot blocks (the tail lanes write zeros)
    constexpr int HV = DMA ? (HVR + 63) / 64 * 64 : HVR;
    constexpr int HVD = HZ * HY * HX;               // halo voxels (staging enumerates these)
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int TILE_VOX = TZ * TY * TX;
    constexpr int NITEMS = (2 * HVD + NTHREADS - 1) / NTHREADS;
    constexpr int RECB = NT * 32 * ES;              // bytes of one voxel's output slice
    constexpr int RECP = RECB + 16;                 // padded LDS stride (8-way -> 2-way conflicts)
    constexpr int EPI_UNITS = NWAVES * (POOL ? MT : 1) * 32 * RECP / 16;   // POOL: all MT groups at once
    constexpr int IMG = 2 * HV;                     // slots of one image (two channel groups)
    constexpr int LDS_UNITS = (DMA ? 2 : 1) * IMG > EPI_UNITS ? (DMA ? 2 : 1) * IMG : EPI_UNITS;
    static_assert(!POOL || (!DMA && WAVES_M * MT * 32 == TZ * TY * TX && TZ % 2 == 0 && TY % 2 == 0 && TX % 2 == 0),
                  "pooled tile shape");
    constexpr int NBLK = HV / 64;                   // DMA: 64-slot blocks per plane
    constexpr int NDMA = DMA ? (2 * NBLK + NWAVES - 1) / NWAVES : 1;   // blocks per wave and image
    static_assert(!DMA || HXS == HX, "DMA staging wants dense rows");
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
    // mod 16, so its lanes use the bank slots the first row leaves free (see zpipe)
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
    const int z0 = a.org[0] + tz * TZ, y0 = a.org[1] + ty * TY, x0 = a.org[2] + tx * TX;
    const int zend = a.org[0] + a.ext[0], yend = a.org[1] + a.ext[1], xend = a.org[2] + a.ext[2];

    const int ntiles = a.cout >> 5;
    const int ntile0 = (blockIdx.y * WAVES_N + wn) * NT;

    int base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wm * MT + mt) * 32 + r;
        m = m < TILE_VOX ? m : TILE_VOX - 1;
        const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
        base[mt] = z * PLS + y * HXS + x + half * HV;
    }

    // staging piece i = tid + it * NTHREADS is 16-byte group i & 1 of halo voxel
    // i >> 1: consecutive lanes read consecutive bytes of a halo row of the chunk plane
    const size_t patch_vox = (size_t)a.d * a.h * a.w;
    unsigned voffs[NITEMS];
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) {
        const int i = tid + it * NTHREADS;
        const int hv = i >> 1;
        const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = i < 2 * HVD && (unsigned)gz < (unsigned)a.d &&
                        (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        voffs[it] = ok ? (unsigned)((gz * a.h + gy) * a.w + gx) * 32u + (i & 1) * 16u : kOutOfRange;
    }

    // DMA block j = wave + k * NWAVES of an image: plane (group) j / NBLK, slots (j % NBLK) * 64 + lane
    unsigned dvoff[NDMA];
    if (DMA) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int j = wave + k * NWAVES;
            const int g = j / NBLK, slot = (j % NBLK) * 64 + lane;
            const int hz = slot / (HY * HX), hy = (slot / HX) % HY, hx = slot % HX;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = j < 2 * NBLK && slot < HVR && (unsigned)gz < (unsigned)a.d &&
                            (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            dvoff[k] = ok ? (unsigned)((gz * a.h + gy) * a.w + gx) * 32u + g * 16u : kOutOfRange;
        }
    }

    // accumulators start from the folded bias: register 4q+k of a lane is channel
    // 8q + 4*half + k of its slice (no bias pass in the epilogue)
    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (split-K ranges start from zero; the reduction adds the bias)
            float4 b = *reinterpret_cast<const float4*>(a.bias + (ntile0 + nt) * 32 + 8 * q + 4 * half);
            if (a.ksplit > 1) b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][nt][4 * q + 0] = b.x; acc[mt][nt][4 * q + 1] = b.y;
                acc[mt][nt][4 * q + 2] = b.z; acc[mt][nt][4 * q + 3] = b.w;
            }
        }

    // this workgroup's range of input-channel chunks (all of them unless split-K)
    const int nchunks_all = (a.ca + a.cb) / KC;
    const int cbeg = (int)blockIdx.z * nchunks_all / a.ksplit;
    const int nchunks = ((int)blockIdx.z + 1) * nchunks_all / a.ksplit;
    uint4 stg[NITEMS];

    auto stage_load = [&](int c) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const size_t patchb = patch_vox * cs * ES;  // bytes of one patch of this source
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(src + (size_t)nb * patchb, patchb);
        const unsigned cbase = (unsigned)(ch0 / KC) * (unsigned)patch_vox * 32u;  // chunk plane
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) stg[it] = buf_load16(rsrc, voffs[it], cbase);
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int it = 0; it < NITEMS; ++it) {
            const int i = tid + it * NTHREADS;
            const int hv = i >> 1;
            if (i < 2 * HVD) lds[(i & 1) * HV + (hv / (HY * HX)) * PLS + ((hv / HX) % HY) * HXS + hv % HX] = stg[it];
        }
    };

    // chunk c -> image buffer "buf", by LDS-DMA (this wave's blocks)
    const unsigned lds_base = (unsigned)(size_t)lds;
    auto dma_load = [&](int c, int buf) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const size_t patchb = patch_vox * cs * ES;
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(src + (size_t)nb * patchb, patchb);
        const unsigned cbase = (unsigned)(ch0 / KC) * (unsigned)patch_vox * 32u;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int j = wave + k * NWAVES;
            if (j < 2 * NBLK) {   // wave-uniform
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    lds_base + (unsigned)((buf * IMG + (j / NBLK) * HV + (j % NBLK) * 64) * 16));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(dvoff[k]), "s"(rsrc), "s"(cbase) : "memory", "m0");   // m0 is not allocatable: nothing of the compiler's lives in it
            }
        }
    };

#ifdef EXASPIM_TRACE
    const size_t trace_rec = ((size_t)blockIdx.x * NWAVES + wave) * 16;
    if (a.trace && lane == 0)
        a.trace[trace_rec + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                  (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    // step t of the tap loop handles tap tap_of(t) = dz * 9 + dy * 3 + dx
    constexpr auto tap_of = [](int t) { return ZORD ? (t % 3) * 9 + t / 3 : t; };
    // Weight ring, primed for the first PD taps of a chunk BEFORE the barriers in front of it
    // (in the prologue next to the staging loads, later right after the previous chunk's last
    // tap): the L2 latency of a chunk's first fragments passes under the wait for the staged
    // image instead of after it.
    uint4 wring[PD + 1][NT];
    auto prime_weights = [&](int c) {
        const uint4* wp = static_cast<const uint4*>(a.weights) + ((size_t)c * 27 * ntiles + ntile0) * 64 + lane;
#pragma unroll
        for (int t = 0; t < PD; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wring[t][nt] = wp[((size_t)tap_of(t) * ntiles + nt) * 64];
    };
    EXA_TRACE(0);
    if (DMA) dma_load(cbeg, 0); else stage_load(cbeg);
    prime_weights(cbeg);
    EXA_TRACE(1);
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else stage_store();
    __syncthreads();
    EXA_TRACE(2);

    for (int c = cbeg; c < nchunks; ++c) {
        const uint4* wp = static_cast<const uint4*>(a.weights) +
                          ((size_t)c * 27 * ntiles + ntile0) * 64 + lane;

        // DMA: chunks alternate between the two image buffers
        const int cur = DMA ? (c - cbeg) & 1 : 0;
        const uint4* const img = lds + cur * IMG;
        uint4 xf[2][MT];
        {
            constexpr int t0 = tap_of(0);
            constexpr int tapoff0 = (t0 / 9) * PLS + ((t0 / 3) % 3) * HXS + t0 % 3;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[0][mt] = img[base[mt] + tapoff0];
        }

        const bool more = c + 1 < nchunks;
        if (ES == 2 && EXASPIM_SETPRIO_T14) __builtin_amdgcn_s_setprio(EXASPIM_SETPRIO_T14);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            if (t + PD < 27) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wring[(t + PD) % (PD + 1)][nt] = wp[((size_t)tap_of(t + PD) * ntiles + nt) * 64];
            }
            if (t == ISSUE_T && more) {
                if (DMA) dma_load(c + 1, cur ^ 1); else stage_load(c + 1);
            }
#ifdef EXASPIM_TRACE
            // 2-chunk layers leave stamps 9..11 free: marks after taps 7, 14 and 21 of the first chunk
            if (nchunks_all == 2 && c == cbeg && t > 0 && t % 7 == 0 && t / 7 <= 3) EXA_TRACE(8 + t / 7);
#endif
            if (t + 1 < 27) {
                const int tn = tap_of(t + 1);
                const int tapoff = (tn / 9) * PLS + ((tn / 3) % 3) * HXS + tn % 3;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[(t + 1) & 1][mt] = img[base[mt] + tapoff];
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
        if (ES == 2 && EXASPIM_SETPRIO_T14) __builtin_amdgcn_s_setprio(0);
        if (more) prime_weights(c + 1);
        if (c - cbeg < 4) EXA_TRACE(3 + 3 * (c - cbeg));
        if (DMA) {
            // this wave's DMA blocks (and the primed weights) have landed; behind the barrier so have
            // everyone's, and everyone is done reading this chunk's buffer
            if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c - cbeg < 4) EXA_TRACE(4 + 3 * (c - cbeg));
            if (more && c - cbeg < 3) EXA_TRACE(5 + 3 * (c - cbeg));
        } else {
            __syncthreads();  // every wave is done reading this chunk's image
            if (c - cbeg < 4) EXA_TRACE(4 + 3 * (c - cbeg));
            if (more) {
                stage_store();
                __syncthreads();
                if (c - cbeg < 3) EXA_TRACE(5 + 3 * (c - cbeg));
            }
        }
    }

    if (a.ksplit > 1) {
        // ---- split-K: float32 partial sums, [range][patch][voxel][cout] ------------------
        float* const part = a.partial + ((size_t)blockIdx.z * a.n + nb) * patch_vox * a.cout;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = (wm * MT + mt) * 32 + r;
            const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
            const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
            if (m < TILE_VOX && gz < zend && gy < yend && gx < xend) {
                float* rec = part + (((size_t)gz * a.h + gy) * a.w + gx) * a.cout;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(rec + (ntile0 + nt) * 32 + 8 * q + 4 * half) =
                            make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2],
                                        acc[mt][nt][4 * q + 3]);
            }
        }
        EXA_TRACE(14);
        return;
    }

    // ---- epilogue: bias + LeakyReLU, transposed through LDS ------------------
    char* wl = reinterpret_cast<char*>(lds) + wave * ((POOL ? MT : 1) * 32 * RECP);
    if (POOL) {
        // all groups first: group mt of wave w sits at ((w * MT + mt) * 32 + voxel) * RECP
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = nt * 32 + 8 * q + 4 * half;
                    store4<Tag>(wl + mt * (32 * RECP), (size_t)(r * RECP) / ES + cl,
                                leaky(acc[mt][nt][4 * q + 0], a.slope), leaky(acc[mt][nt][4 * q + 1], a.slope),
                                leaky(acc[mt][nt][4 * q + 2], a.slope), leaky(acc[mt][nt][4 * q + 3], a.slope));
                }
        __syncthreads();
        constexpr int NPL = RECB / 32;
        {   // the layer's own output, as below
            const int vv = lane >> 1, sub = lane & 1;
            char* const dplane = static_cast<char*>(a.dst) +
                                 ((size_t)nb * (a.cout / KC) + ntile0 * (32 / KC)) * patch_vox * 32;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = (wm * MT + mt) * 32 + vv;
                const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
                const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
                const bool ok = gz < zend && gy < yend && gx < xend;
                const size_t vox = ((size_t)gz * a.h + gy) * a.w + gx;
#pragma unroll
                for (int ck = 0; ck < NPL; ++ck) {
                    const uint4 val = *reinterpret_cast<const uint4*>(
                        wl + mt * (32 * RECP) + vv * RECP + (ck * 2 + sub) * 16);
                    if (ok)
                        *reinterpret_cast<uint4*>(dplane + ((size_t)ck * patch_vox + vox) * 32 + sub * 16) = val;
                }
            }
        }
        // MaxPool3d(2): piece p = 16-byte group "sub" of pooled voxel (pz, py, px) in chunk plane ck
        // of cout slice wn; lanes run along (px, sub), so a row of the pooled tile is one run of
        // TX / 2 x 32 contiguous bytes
        constexpr int PX = TX / 2, PY = TY / 2, PZ = TZ / 2;
        constexpr int NPIECE = 2 * PX * PY * PZ * NPL * WAVES_N;
        const int pd = a.d >> 1, ph = a.h >> 1, pw2 = a.w >> 1;
        const size_t pvox = (size_t)pd * ph * pw2;
        const char* const lb = reinterpret_cast<const char*>(lds);
#pragma unroll
        for (int p0 = 0; p0 < NPIECE; p0 += NTHREADS) {
            const int pp = p0 + tid;
            const int sub = pp & 1, px = (pp >> 1) % PX;
            int rest = (pp >> 1) / PX;
            const int py = rest % PY; rest /= PY;
            const int pz = rest % PZ; rest /= PZ;
            const int ck = rest % NPL, pwn = rest / NPL;
            const int qz = (z0 >> 1) + pz, qy = (y0 >> 1) + py, qx = (x0 >> 1) + px;
            if (pp < NPIECE && qz < pd && qy < ph && qx < pw2) {
                uint4 mx;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = ((2 * pz + (k >> 2)) * TY + 2 * py + ((k >> 1) & 1)) * TX + 2 * px + (k & 1);
                    const int w_src = (m / (MT * 32)) * WAVES_N + pwn;      // wave that produced it
                    const uint4 v = *reinterpret_cast<const uint4*>(
                        lb + (size_t)((w_src * MT + (m / 32) % MT) * 32 + m % 32) * RECP + (ck * 2 + sub) * 16);
                    if (ES == 2) mx = k == 0 ? key16(v) : maxkey16(mx, key16(v));   // order-preserving keys
                    else mx = k == 0 ? v : max16<Tag>(mx, v);
                }
                if (ES == 2) mx = key16(mx);
                const int ptile = (blockIdx.y * WAVES_N + pwn) * NT;   // first 32-cout tile of that slice
                char* const pplane = static_cast<char*>(a.pool_dst) +
                                     ((size_t)nb * (a.cout / KC) + ptile * (32 / KC) + ck) * pvox * 32;
                *reinterpret_cast<uint4*>(pplane + (((size_t)qz * ph + qy) * pw2 + qx) * 32 + sub * 16) = mx;
            }
        }
        EXA_TRACE(14);
        return;
    }
    if (ES == 2 && EXASPIM_DIRECT_EPILOGUE) {
        // 16-bit types: records assembled with v_permlane32_swap (record_half), no LDS round trip
        char* const dplane = static_cast<char*>(a.dst) +
                             ((size_t)nb * (a.cout / KC) + ntile0 * (32 / KC)) * patch_vox * 32;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = (wm * MT + mt) * 32 + r;
            const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
            const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
            const bool ok = m < TILE_VOX && gz < zend && gy < yend && gx < xend;
            char* const dvox = dplane + (((size_t)gz * a.h + gy) * a.w + gx) * 32 + half * 16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                uint2 grp[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    grp[q] = pack4<Tag>(leaky(acc[mt][nt][4 * q + 0], a.slope), leaky(acc[mt][nt][4 * q + 1], a.slope),
                                        leaky(acc[mt][nt][4 * q + 2], a.slope), leaky(acc[mt][nt][4 * q + 3], a.slope));
#pragma unroll
                for (int ck = 0; ck < 2; ++ck) {
                    const uint4 rec = record_half(grp[2 * ck], grp[2 * ck + 1]);
                    if (ok) *reinterpret_cast<uint4*>(dvox + (size_t)(nt * 2 + ck) * patch_vox * 32) = rec;
                }
            }
        }
        EXA_TRACE(14);
        return;
    }
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
                v0 = leaky(v0, a.slope);
                v1 = leaky(v1, a.slope);
                v2 = leaky(v2, a.slope);
                v3 = leaky(v3, a.slope);
                store4<Tag>(wl, (size_t)(r * RECP) / ES + cl, v0, v1, v2, v3);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // one store instruction = one chunk plane's 32 voxel records (32 B each)
        constexpr int NPL = RECB / 32;           // chunk planes of this wave's output slice
        const int vv = lane >> 1, sub = lane & 1;
        const int m = (wm * MT + mt) * 32 + vv;
        const int z = m / (TY * TX), y = (m / TX) % TY, x = m % TX;
        const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
        const bool ok = m < TILE_VOX && gz < zend && gy < yend && gx < xend;
        const size_t vox = ((size_t)gz * a.h + gy) * a.w + gx;
        char* const dplane = static_cast<char*>(a.dst) +
                             ((size_t)nb * (a.cout / KC) + ntile0 * (32 / KC)) * patch_vox * 32;
#pragma unroll
        for (int ck = 0; ck < NPL; ++ck) {
            const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + (ck * 2 + sub) * 16);
            if (ok)
                *reinterpret_cast<uint4*>(dplane + ((size_t)ck * patch_vox + vox) * 32 + sub * 16) = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    EXA_TRACE(14);
}

// ---- conv3x3x3_zpipe: z-column tiles for the 32-cout slices ----------------------------
// A wave owns one 32-voxel (y, x) group of the tile times all TZ planes (TZ
// accumulators). For a fixed in-plane tap g = (dy, dx) the operand fragment of
// input plane zin is the B operand of up to three MFMAs (dz = 0, 1, 2 -> output
// planes zin, zin-1, zin-2): (TZ + 2) LDS reads per 3 * TZ MFMAs. The chunk's 27
// weight fragments are staged in LDS too (one copy per workgroup instead of one L2
// read per wave).
//
// Shaped by two measurements (tools/conv_trace.hip): a workgroup spends ~45 % of
// its life outside the tap loops, so most of the time a SIMD has ONE wave feeding
// its matrix pipe, and a wave whose operand reads sit right before the MFMAs that
// use them reaches only ~70 % alone. Hence
//  * the tap loop is one flat sequence of 9 * (TZ + 2) steps (g, zin); the operand
//    fragment of step s + D is read from LDS before the MFMAs of step s into a ring
//    of D + 1 registers, the next tap's three weight fragments are read one tap
//    ahead, and a scheduling fence per step pins that order, so LDS latency hides
//    under the wave's own MFMAs;
//  * the next chunk's global loads are dealt one per step;
//  * staging maps (column, 16-byte group) pairs to lanes, so with the blocked
//    layout a load instruction covers whole halo rows of contiguous bytes (the
//    texture addresser works per 64-byte segment: 16 cycles per instruction
//    instead of 64 with one voxel record per lane).
template <typename Tag, int TZ, int TY, int TX, int MINW, int D, int HEAD = 0, bool POOL = false>
__global__ __launch_bounds__(TY* TX * 2, MINW) void conv3x3x3_zpipe(
    ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int G = Tag::kG;
    constexpr int KC = 2 * G;
    constexpr int ES = 16 / G;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HXP = HX;                // row stride (slots)
    constexpr int PLANE = HY * HXP;        // slots per halo plane
    constexpr int HVP = HZ * PLANE;        // slots per channel group
    // stride between the two channel-group planes: an odd multiple of 128 bytes, so the
    // lane pair that stages one voxel (group 0, group 1) writes different LDS banks
    constexpr int GS = HVP + (24 - HVP % 16) % 16;
    constexpr int NWAVES = TY * TX / 32;
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int NPAIR = 2 * HY * HX;     // (column, group) pairs of the halo block
    constexpr int REM = NPAIR > NTHREADS ? NPAIR - NTHREADS : 0;
    constexpr int SEC = (REM * HZ + NTHREADS - 1) / NTHREADS;
    constexpr int NITEMS = HZ + SEC;       // halo pieces per thread
    constexpr int RECB = 32 * ES;
    constexpr int RECP = RECB + 16;        // padded LDS stride of the output transposition
    constexpr int EPI_UNITS = NWAVES * 32 * RECP / 16;
    constexpr int WUNITS = 27 * 64;        // the chunk's weight fragments in LDS
    constexpr int WITEMS = (WUNITS + NTHREADS - 1) / NTHREADS;
    constexpr int XUNITS = 2 * GS > EPI_UNITS ? 2 * GS : EPI_UNITS;
    constexpr int LDS_UNITS = XUNITS + WUNITS;
    constexpr int NS = 9 * HZ;             // steps per chunk
    constexpr int R = D + 1;               // operand ring
#ifndef EXASPIM_POOL_DIRECT
#define EXASPIM_POOL_DIRECT 1   // z-column kernel, 16-bit, fused max-pool: output from registers, pair maxima through LDS (0: all planes through LDS)
#endif
#ifndef EXASPIM_HEAD_TZ
#define EXASPIM_HEAD_TZ 5   // planes per tile of the trimmed fused-head launch (0: the 6-plane tiles of every other launch; A/B builds)
#endif
#ifndef EXASPIM_LOAD_STRIDE
// stride 1 / 2 / 3 inside a 1024^3 step (us per launch, same box): up3.3 + up4.0 510 / 490 / 496,
// inc.3 831 / 801 / 801, up4.3 with the fused head 514 / 496 / 536
#define EXASPIM_LOAD_STRIDE 2
#endif
    constexpr int LOAD_STRIDE = EXASPIM_LOAD_STRIDE * (NITEMS + WITEMS) <= NS ? EXASPIM_LOAD_STRIDE : 1;
    static_assert(TY * TX % 32 == 0 && NPAIR <= 2 * NTHREADS, "tile shape");
    static_assert(LOAD_STRIDE * (NITEMS + WITEMS) <= NS, "the staged pieces fit into the steps");

    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_UNITS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    // Voxel of the 32-group this lane works on. With 16-wide rows the group is two
    // rows whose LDS slots differ by HXP; taking the second row's x in rotated
    // order, x = (i - HXP) mod 16, puts lane 16+i on bank slot i (mod 16), the
    // complement of what its ds_read_b128 lane group already uses: no conflicts.
    const int r = (TX == 16 && (lane & 16)) ? 16 + (((lane & 15) - HXP) & 15) : (lane & 31);

    // Tiles of this workgroup. The tile list is cut into 8 contiguous ranges, one per
    // XCD (workgroups are dealt round-robin to the XCDs, so blockIdx.x & 7 is the XCD);
    // the workgroups of an XCD walk their range together, slot by slot, so tiles that
    // share halo planes are resident in the same L2 at the same time. With as many
    // workgroups as tiles this is the plain one-tile-per-workgroup order.
    const int total = tiles_z * tiles_y * tiles_x * a.n;
    int t_first, t_count, t_step;
    {
        const int q = total >> 3, rem = total & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        t_first = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
        t_count = q + (xcd < rem ? 1 : 0) - slot;        // tiles left from t_first on
        t_step = (gridDim.x + 7 - xcd) >> 3;             // workgroups on this XCD
    }
    if (t_count <= 0) return;
    const int ntiles = a.cout >> 5;
    const int ntile0 = blockIdx.y;

    struct Tile {
        int z0, y0, x0, nb;
    };
    auto tile_at = [&](int id) {
        Tile t;
        // (x fastest. Measured alternative, z fastest -- whole z-columns resident in an XCD's L2
        // together so that neighbours share their two halo planes: HBM reads 1363 -> 1429 MB per
        // inc.3 launch, 1 % slower.) The divisions run on the vector ALU; readfirstlane puts the
        // wave-uniform results back into scalar registers.
        t.x0 = __builtin_amdgcn_readfirstlane(a.org[2] + (id % tiles_x) * TX); id /= tiles_x;
        t.y0 = __builtin_amdgcn_readfirstlane(a.org[1] + (id % tiles_y) * TY); id /= tiles_y;
        t.z0 = __builtin_amdgcn_readfirstlane(a.org[0] + (id % tiles_z) * TZ); id /= tiles_z;
        id = __builtin_amdgcn_readfirstlane(id);
        t.nb = id;
        return t;
    };

    const int pos = wave * 32 + r;  // this lane's position inside the plane
    const int col = (pos / TX) * HXP + (pos % TX) + half * GS;

    // ---- staging map ----------------------------------------------------------
    // primary: thread t < NPAIR moves pair t = (column t / 2, group t & 1), all HZ
    // planes (one vector offset; plane and chunk ride in the scalar offset).
    // secondary: the REM pairs beyond NTHREADS, piece q = t + k * NTHREADS is
    // plane q / REM of pair NTHREADS + q % REM (own vector offset each).
    // LDS slots do not depend on the tile; the global offsets are set per tile.
    const int plane_vox = a.h * a.w;
    const size_t patch_vox = (size_t)a.d * plane_vox;
    const bool p_ok = tid < NPAIR;
    const int p_hy = (tid >> 1) / HX, p_hx = (tid >> 1) % HX, p_kg = tid & 1;
    const int p_slot = p_kg * GS + p_hy * HXP + p_hx;
    unsigned p_voff;      // byte offset inside a z-plane of a chunk plane (or out of range)
    unsigned s_voff[SEC > 0 ? SEC : 1];
    int s_slot[SEC > 0 ? SEC : 1];
#pragma unroll
    for (int k = 0; k < SEC; ++k) {
        const int q = tid + k * NTHREADS;
        const int pr = NTHREADS + q % REM, hz = q / REM;
        const int c = pr >> 1, kg = pr & 1;
        s_slot[k] = q < REM * HZ ? kg * GS + hz * PLANE + (c / HX) * HXP + c % HX : -1;
    }
    auto set_offsets = [&](const Tile& t) {
        {
            const int gy = t.y0 + p_hy - 1, gx = t.x0 + p_hx - 1;
            const bool in = p_ok && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            p_voff = in ? (unsigned)(gy * a.w + gx) * 32u + p_kg * 16u : kOutOfRange;
        }
#pragma unroll
        for (int k = 0; k < SEC; ++k) {
            const int q = tid + k * NTHREADS;
            const int pr = NTHREADS + q % REM, hz = q / REM;
            const int c = pr >> 1, kg = pr & 1;
            const int gz = t.z0 + hz - 1, gy = t.y0 + c / HX - 1, gx = t.x0 + c % HX - 1;
            const bool in = q < REM * HZ && (unsigned)gz < (unsigned)a.d &&
                            (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            s_voff[k] = in ? (unsigned)((gz * a.h + gy) * a.w + gx) * 32u + kg * 16u : kOutOfRange;
        }
    };
    // weight fragments: piece i = tid + it * NTHREADS is element (i & 63) of tap i >> 6
    const unsigned wvoff = (((tid >> 6) * ntiles) * 64 + (tid & 63)) * 16u;

    // the slice's folded bias, kept in LDS: every tile's accumulators start from it
    __shared__ __attribute__((aligned(16))) float bias_s[32];
    if (tid < 32) bias_s[tid] = a.bias[ntile0 * 32 + tid];
    // the fused head's weights and bias live there too (read back once per tile)
    __shared__ __attribute__((aligned(16))) float head_s[HEAD > 0 ? HEAD * 32 + 4 : 4];
    if (HEAD > 0) {
        if (tid < HEAD * 32) head_s[tid] = a.head_w[tid];
        if (tid < HEAD) head_s[HEAD * 32 + tid] = a.head_b[tid];
    }

    const int nchunks = (a.ca + a.cb) / KC;
    uint4 stg[NITEMS + WITEMS];  // halo pieces, then weight fragments
    uint4* const wlds = lds + XUNITS;
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(a.weights, (size_t)nchunks * 27 * ntiles * 1024);

    // where chunk c of patch nb lives: descriptor of the patch of its source, offset of its plane
    struct ChunkSrc {
        __amdgpu_buffer_rsrc_t rsrc;
        __amdgpu_buffer_rsrc_t none;   // the same with zero records: every load returns zeros
        unsigned cbase;
    };
    auto chunk_src = [&](int c, int nb) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const size_t patchb = patch_vox * cs * ES;  // bytes of one patch of this source
        return ChunkSrc{make_rsrc(src + (size_t)nb * patchb, patchb), make_rsrc(src + (size_t)nb * patchb, 0),
                        (unsigned)(ch0 / KC) * (unsigned)patch_vox * 32u};
    };
    // piece i of chunk c of the tile whose first plane is z0: global -> stg[i]
    auto load_piece = [&](const ChunkSrc& cs, int c, int z0, int i) {
        if (i < NITEMS) {
            if (i < HZ) {
                // A z-halo plane outside the patch is loaded through the zero-record descriptor (the
                // range check returns zeros) rather than set to zero in a branch: writing the staging
                // registers there made hipcc wait for EVERY load in flight (s_waitcnt vmcnt(0) in the
                // middle of the tap loop of every first and last tile of a column).
                const int gz = z0 + i - 1;  // wave-uniform
                stg[i] = (unsigned)gz < (unsigned)a.d
                             ? buf_load16(cs.rsrc, p_voff, cs.cbase + (unsigned)gz * plane_vox * 32u)
                             : buf_load16(cs.none, p_voff, 0);
            } else {
                stg[i] = buf_load16(cs.rsrc, s_voff[i - HZ], cs.cbase);
            }
        } else {
            const int it = i - NITEMS;
            // taps it * NWAVES + wave; the last round covers taps < 27 only
            stg[i] = it * NWAVES + wave < 27
                         ? buf_load16(wrsrc, wvoff, ((c * 27 + it * NWAVES) * ntiles + ntile0) * 1024)
                         : make_uint4(0, 0, 0, 0);
        }
    };
    auto stage_store = [&]() {
        if ((EXASPIM_ABLATE & 4) && stg[0].x != 0x12345u) return;
        if (p_ok) {
#pragma unroll
            for (int hz = 0; hz < HZ; ++hz) lds[p_slot + hz * PLANE] = stg[hz];
        }
#pragma unroll
        for (int k = 0; k < SEC; ++k)
            if (s_slot[k] >= 0) lds[s_slot[k]] = stg[HZ + k];
#pragma unroll
        for (int it = 0; it < WITEMS; ++it) {
            const int i = tid + it * NTHREADS;
            if (i < WUNITS) wlds[i] = stg[NITEMS + it];
        }
    };

    int tile_id = t_first;
    Tile cur = tile_at(tile_id);
    set_offsets(cur);
    {
        const ChunkSrc cs0 = chunk_src(0, cur.nb);
#pragma unroll
        for (int i = 0; i < NITEMS + WITEMS; ++i) load_piece(cs0, 0, cur.z0, i);
    }
    stage_store();

    for (;;) {
#ifdef EXASPIM_TRACE
        const size_t trace_rec = ((size_t)tile_id * NWAVES + wave) * 16;
        if (a.trace && lane == 0)
            a.trace[trace_rec + 15] =
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
        EXA_TRACE(0);
        EXA_TRACE(1);
        __syncthreads();   // this tile's first chunk (and, the first time, the bias) is in LDS
        // register 4q+k of a lane is channel 8q + 4*half + k of the slice
        f32x16 acc[TZ];
        const int half_t = fresh_lane() >> 5;   // (recomputed per tile, see fresh_lane())
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(bias_s + 8 * q + 4 * half_t);
#pragma unroll
            for (int mt = 0; mt < TZ; ++mt) {
                acc[mt][4 * q + 0] = b.x; acc[mt][4 * q + 1] = b.y;
                acc[mt][4 * q + 2] = b.z; acc[mt][4 * q + 3] = b.w;
            }
        }
        EXA_TRACE(2);

        // During the last chunk the first chunk of the workgroup's NEXT tile is
        // prefetched, so only the first tile of a workgroup pays the global latency.
        t_count -= t_step;
        const bool has_next = t_count > 0;
        Tile nxt = cur;
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            const bool pre = more || has_next;
            if (!more && has_next) {
                nxt = tile_at(tile_id + t_step);
                set_offsets(nxt);   // every load of the current tile has been issued
            }
            const ChunkSrc csn = chunk_src(more ? c + 1 : 0, more ? cur.nb : nxt.nb);
            const int cn = more ? c + 1 : 0, zn = more ? cur.z0 : nxt.z0;
            uint4 xr[R];       // operand ring: fragment of step s lives in xr[s % R]
            // weight fragments (dz) of tap g: fragment dz is used in steps zin = dz ..
            // dz + TZ - 1 of its tap, so the next tap's fragment takes over the register
            // as soon as that window closes (two steps before its own window opens)
            uint4 wb[3];
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) wb[dz] = wlds[(dz * 9) * 64 + lane];
#pragma unroll
            for (int s = 0; s < D; ++s)
                xr[s % R] = lds[col + (s % HZ) * PLANE + ((s / HZ) / 3) * HXP + (s / HZ) % 3];
            __builtin_amdgcn_sched_barrier(0);
            if (ES == 2 && EXASPIM_SETPRIO) __builtin_amdgcn_s_setprio(EXASPIM_SETPRIO);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int g = s / HZ, zin = s % HZ;
                if (s + D < NS) {
                    const int g2 = (s + D) / HZ, z2 = (s + D) % HZ;
                    xr[(s + D) % R] = lds[col + z2 * PLANE + (g2 / 3) * HXP + g2 % 3];
                }
#ifdef EXASPIM_TRACE
                // 2-chunk layers leave stamps 9..11 free: quarter marks inside the first chunk's loop
                if (nchunks == 2 && c == 0 && s > 0 && s % (NS / 4) == 0 && s / (NS / 4) <= 3) EXA_TRACE(8 + s / (NS / 4));
#endif
                if (g + 1 < 9 && zin >= TZ) wb[zin - TZ] = wlds[((zin - TZ) * 9 + g + 1) * 64 + lane];
                if (g > 0 && zin == 0) wb[2] = wlds[(2 * 9 + g) * 64 + lane];
                // one staged piece every LOAD_STRIDE steps: issued back to back in the first steps
                // the loads of all eight waves of a CU queue up in the texture addresser, and the
                // MFMAs behind a load that cannot issue wait with it (the first quarter of the loop
                // took 5.1 k cycles, the others 1.3-1.8 k; spread out 3.2 k: tools/conv_trace.hip)
                if (!(EXASPIM_ABLATE & 1) && pre && s % LOAD_STRIDE == 0 && s / LOAD_STRIDE < NITEMS + WITEMS)
                    load_piece(csn, cn, zn, s / LOAD_STRIDE);
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int z = zin - dz;
                    if (z >= 0 && z < TZ) mma_inplace<Tag>(acc[z], wb[dz], xr[s % R]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ES == 2 && EXASPIM_SETPRIO) __builtin_amdgcn_s_setprio(0);
            // The 16-bit MFMAs above are inline assembly (mma_inplace), so hipcc's hazard recognizer does
            // not know that the accumulators were written by the matrix pipe: the wait states between an
            // 8-pass MFMA and the first vector-ALU read of its destination (the epilogue) are spelled out
            // here instead of being left to whatever happens to stand in between.
            if (ES == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1");
            if (c < 4) EXA_TRACE(3 + 3 * c);
            __syncthreads();   // every wave is done reading this chunk's image
            if (c < 4) EXA_TRACE(4 + 3 * c);
            if (more) {
                stage_store();
                __syncthreads();
                if (c < 3) EXA_TRACE(5 + 3 * c);
            }
        }
        // The next tile's prefetched pieces are waited for HERE, before the epilogue issues its stores
        // (the empty asm makes every staged register "used", so hipcc puts the s_waitcnt for the
        // prefetch loads in front of it; afterwards they are plain values). Left to the staging behind
        // the epilogue, that wait is s_waitcnt vmcnt(0) and also covers the epilogue's stores -- a tile
        // boundary then costs a full store round trip. (A counted wait, vmcnt(#stores), behind
        // unconditional range-checked stores would do as well -- range-dropped stores retire in order
        // with older loads, tools/vmcnt_order.hip -- but the epilogue's stores go through buf_store16,
        // whose inline assembly hipcc cannot count.)
        if (EXASPIM_SETTLE_FIRST && has_next) {
#pragma unroll
            for (int i = 0; i < NITEMS + WITEMS; ++i)
                asm volatile("" : "+v"(stg[i].x), "+v"(stg[i].y), "+v"(stg[i].z), "+v"(stg[i].w));
        }
        // epilogues that do not go through LDS leave the image free from here on
        constexpr bool kLdsFreeEpilogue = HEAD > 0 || (ES == 2 && !POOL && EXASPIM_DIRECT_EPILOGUE);
        if (EXASPIM_STAGE_FIRST && kLdsFreeEpilogue && has_next) stage_store();

        // the lane's coordinates inside the tile, recomputed per tile (see fresh_lane())
        const int lane_e = fresh_lane();
        const int half_e = lane_e >> 5;
        const int r_e = (TX == 16 && (lane_e & 16)) ? 16 + (((lane_e & 15) - HXP) & 15) : (lane_e & 31);
        const int pos_e = wave * 32 + r_e;
        if (HEAD > 0) {
            // ---- fused head: OutConv 1x1x1 (+ sigmoid) on the accumulators -----------
            // lane (voxel r, half h) holds channels 8q + 4h + j of its voxel: a 16-term
            // partial dot product per output, completed by the other half-wave.
            // Channel quads outermost, so only 4 weights per output are live at a time
            // (the next tile's staged pieces occupy most of the register file here).
            float part[TZ][HEAD > 0 ? HEAD : 1];
#pragma unroll
            for (int z = 0; z < TZ; ++z)
#pragma unroll
                for (int o = 0; o < HEAD; ++o) part[z][o] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 hw[HEAD > 0 ? HEAD : 1];
#pragma unroll
                for (int o = 0; o < HEAD; ++o)
                    hw[o] = *reinterpret_cast<const float4*>(head_s + o * 32 + 8 * q + 4 * half_e);
#pragma unroll
                for (int z = 0; z < TZ; ++z) {
                    float v0 = acc[z][4 * q + 0], v1 = acc[z][4 * q + 1];
                    float v2 = acc[z][4 * q + 2], v3 = acc[z][4 * q + 3];
                    v0 = leaky(v0, a.slope);
                    v1 = leaky(v1, a.slope);
                    v2 = leaky(v2, a.slope);
                    v3 = leaky(v3, a.slope);
#pragma unroll
                    for (int o = 0; o < HEAD; ++o)
                        part[z][o] = fmaf(v3, hw[o].w, fmaf(v2, hw[o].z, fmaf(v1, hw[o].y, fmaf(v0, hw[o].x, part[z][o]))));
                }
            }
            const size_t plane = (size_t)a.h * a.w;
            const int gy = cur.y0 + pos_e / TX, gx = cur.x0 + pos_e % TX;
#if EXASPIM_BUFFER_STORES & 2
            // (unconditional range-checked stores, see the direct epilogue below; outputs are dealt
            // to the two half-waves: even ones are stored by lanes 0-31, odd ones by lanes 32-63)
            const bool okyx = gy < a.org[1] + a.ext[1] && gx < a.org[2] + a.ext[2];
            float* const hpatch = a.head_out + (size_t)cur.nb * HEAD * a.d * plane;
            const size_t hbytes = (size_t)HEAD * a.d * plane * sizeof(float);
            const unsigned hvoff[2] = {okyx && half_e == 0 ? (unsigned)(gy * a.w + gx) * 4u : kOutOfRange,
                                       okyx && half_e == 1 ? (unsigned)(gy * a.w + gx) * 4u : kOutOfRange};
#endif
#pragma unroll
            for (int z = 0; z < TZ; ++z) {
                const int gz = cur.z0 + z;
#if EXASPIM_BUFFER_STORES & 2
                const __amdgpu_buffer_rsrc_t hrsrc = make_rsrc(hpatch, gz < a.org[0] + a.ext[0] ? hbytes : (size_t)0);
#else
                const bool ok = gz < a.org[0] + a.ext[0] && gy < a.org[1] + a.ext[1] && gx < a.org[2] + a.ext[2];
#endif
#pragma unroll
                for (int o = 0; o < HEAD; ++o) {
                    float t = part[z][o] + __shfl_xor(part[z][o], 32) + head_s[HEAD * 32 + o];
                    if (a.head_sigmoid) t = 1.f / (1.f + expf(-t));
#if EXASPIM_BUFFER_STORES & 2
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), hrsrc, (int)hvoff[o & 1],
                                                          (int)((unsigned)(o * a.d + gz) * (unsigned)plane * 4u), 0);
#else
                    // outputs are dealt to the two half-waves so both store
                    if (ok && (o & 1) == half_e)
                        a.head_out[(((size_t)cur.nb * HEAD + o) * a.d + gz) * plane + (size_t)gy * a.w + gx] = t;
#endif
                }
            }
        } else if (ES == 2 && !POOL && EXASPIM_DIRECT_EPILOGUE) {
            // ---- epilogue: LeakyReLU, records assembled with v_permlane32_swap ------
            // (16-bit types without the fused max-pool: nothing goes through LDS, so the next
            // tile's first chunk can be written to the image right behind this)
            char* const dplane = static_cast<char*>(a.dst) +
                                 ((size_t)cur.nb * (a.cout / KC) + ntile0 * 2) * patch_vox * 32;
            const int gy = cur.y0 + pos_e / TX, gx = cur.x0 + pos_e % TX;
            const bool okyx = gy < a.org[1] + a.ext[1] && gx < a.org[2] + a.ext[2];
#if EXASPIM_BUFFER_STORES & 1
            // Every store is ISSUED, as a range-checked buffer store: a lane outside the region gets
            // an out-of-range offset and a plane outside it the zero-record descriptor, and the
            // hardware drops the write -- no branch, no per-store address arithmetic on the vector ALU.
            // (buf_store16: hazard-safe and invisible to hipcc's wait counts, see common.h.)
            const unsigned ovoff = okyx ? (unsigned)(gy * a.w + gx) * 32u + half_e * 16u : kOutOfRange;
#else
            char* const dvox = dplane + ((size_t)gy * a.w + gx) * 32 + half_e * 16;
#endif
#pragma unroll
            for (int z = 0; z < TZ; ++z) {
                uint2 grp[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    grp[q] = pack4<Tag>(leaky(acc[z][4 * q + 0], a.slope), leaky(acc[z][4 * q + 1], a.slope),
                                        leaky(acc[z][4 * q + 2], a.slope), leaky(acc[z][4 * q + 3], a.slope));
                const int gz = cur.z0 + z;     // wave-uniform
#pragma unroll
                for (int ck = 0; ck < 2; ++ck) {
                    const uint4 rec = record_half(grp[2 * ck], grp[2 * ck + 1]);
#if EXASPIM_BUFFER_STORES & 1
                    const unsigned soff = ((unsigned)ck * (unsigned)patch_vox + (unsigned)gz * (unsigned)plane_vox) * 32u;
                    // (descriptor with zero records for a plane outside the region: a scalar select, no branch)
                    const __amdgpu_buffer_rsrc_t orsrc =
                        make_rsrc(dplane, gz < a.org[0] + a.ext[0] ? (size_t)2 * patch_vox * 32 : (size_t)0);
                    buf_store16(rec, orsrc, ovoff, soff);
#else
                    if (okyx && gz < a.org[0] + a.ext[0] && (!(EXASPIM_ABLATE & 2) || rec.x == 0x12345u))
                        *reinterpret_cast<uint4*>(dvox + ((size_t)ck * patch_vox + (size_t)gz * plane_vox) * 32) = rec;
#endif
                }
            }
        } else if (ES == 2 && POOL && EXASPIM_POOL_DIRECT) {
            // ---- epilogue with the fused max-pool, 16-bit types: the layer's own output leaves the
            // registers like in the branch above (v_permlane32_swap, no LDS); of a PAIR of planes only the
            // element-wise maximum goes to LDS, as order-preserving keys, and a pooled piece is the maximum
            // over the 2 x 2 records of its row pair there. Against parking all six planes: half the LDS
            // writes, a third of the reads, no read-back for the 12 output stores. Maximum of the stored
            // (rounded, saturated) values like maxpool2_kernel: same bits.
            static_assert(!(ES == 2 && POOL) || (TZ % 2 == 0 && TY % 2 == 0 && TX == 16), "pooled tile shape");
            constexpr int CPT = 2;
            char* wl = reinterpret_cast<char*>(lds) + wave * ((TZ / 2) * 32 * RECP);
            char* const dplane = static_cast<char*>(a.dst) +
                                 ((size_t)cur.nb * (a.cout / KC) + ntile0 * 2) * patch_vox * 32;
            const int gy = cur.y0 + pos_e / TX, gx = cur.x0 + pos_e % TX;
            const bool okyx = gy < a.org[1] + a.ext[1] && gx < a.org[2] + a.ext[2];
            const unsigned ovoff = okyx ? (unsigned)(gy * a.w + gx) * 32u + half_e * 16u : kOutOfRange;
#pragma unroll
            for (int zp = 0; zp < TZ / 2; ++zp) {
                uint2 grp[2][4];
#pragma unroll
                for (int zz = 0; zz < 2; ++zz)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        grp[zz][q] = pack4<Tag>(leaky(acc[2 * zp + zz][4 * q + 0], a.slope), leaky(acc[2 * zp + zz][4 * q + 1], a.slope),
                                                leaky(acc[2 * zp + zz][4 * q + 2], a.slope), leaky(acc[2 * zp + zz][4 * q + 3], a.slope));
#pragma unroll
                for (int q = 0; q < 4; ++q) {      // the pair's maximum, as keys, where the pooling pass finds it
                    const uint2 km = make_uint2(maxkey16x2(key16x2(grp[0][q].x), key16x2(grp[1][q].x)),
                                                maxkey16x2(key16x2(grp[0][q].y), key16x2(grp[1][q].y)));
                    *reinterpret_cast<uint2*>(wl + zp * (32 * RECP) + r_e * RECP + (8 * q + 4 * half_e) * ES) = km;
                }
#pragma unroll
                for (int zz = 0; zz < 2; ++zz) {
                    const int gz = cur.z0 + 2 * zp + zz;     // wave-uniform
                    const __amdgpu_buffer_rsrc_t orsrc =
                        make_rsrc(dplane, gz < a.org[0] + a.ext[0] ? (size_t)2 * patch_vox * 32 : (size_t)0);
#pragma unroll
                    for (int ck = 0; ck < 2; ++ck) {
                        const uint4 rec = record_half(grp[zz][2 * ck], grp[zz][2 * ck + 1]);
                        buf_store16(rec, orsrc, ovoff, ((unsigned)ck * (unsigned)patch_vox + (unsigned)gz * (unsigned)plane_vox) * 32u);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                // the wave's 2 rows x 16 voxels x TZ / 2 plane pairs give TZ / 2 x 8 pooled voxels; a piece is
                // one 16-byte group of one of them (record = row * 16 + x)
                constexpr int NP = (TZ / 2) * 8 * CPT * 2;
                const int pd = a.d >> 1, ph = a.h >> 1, pw2 = a.w >> 1;
                const size_t pvox = (size_t)pd * ph * pw2;
                char* const pplane = static_cast<char*>(a.pool_dst) +
                                     ((size_t)cur.nb * (a.cout / KC) + ntile0 * CPT) * pvox * 32;
                const __amdgpu_buffer_rsrc_t prsrc = make_rsrc(pplane, (size_t)CPT * pvox * 32);
#pragma unroll
                for (int p0 = 0; p0 < NP; p0 += 64) {
                    const int p = p0 + lane_e;
                    const bool live = p < NP;        // (a lane without a piece works on piece 0; its store is dropped)
                    const int pc = live ? p : 0;
                    const int zp = pc / (8 * CPT * 2), rem = pc % (8 * CPT * 2);
                    const int ck = rem / 16, xp = (rem % 16) >> 1, sb = rem & 1;
                    const char* rec = wl + zp * (32 * RECP) + (2 * xp) * RECP + (ck * 2 + sb) * 16;
                    uint4 m = *reinterpret_cast<const uint4*>(rec);
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        m = maxkey16(m, *reinterpret_cast<const uint4*>(rec + (k >> 1) * 16 * RECP + (k & 1) * RECP));
                    m = key16(m);
                    const int qz = cur.z0 / 2 + zp, qy = cur.y0 / 2 + wave, qx = cur.x0 / 2 + xp;
                    const unsigned pvoff = live && qz < pd && qy < ph && qx < pw2
                                               ? (unsigned)((ck * (int)pvox + (qz * ph + qy) * pw2 + qx) * 32 + sb * 16)
                                               : kOutOfRange;
                    buf_store16(m, prsrc, pvoff, 0u);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            // ---- epilogue: LeakyReLU, transposed through LDS ----------------------
            // One store instruction writes one chunk plane's 32 voxel records (32 B
            // each): two runs of 512 contiguous bytes when the tile row is 16 voxels.
            // The (dead) halo and weight image gives every wave room for TB planes at
            // once, so the LDS round trips and the stores of a batch overlap.
            constexpr int CPT = RECB / 32;                       // chunk planes of a 32-cout slice
            constexpr int TB_MAX = LDS_UNITS * 16 / (NWAVES * 32 * RECP);
            // with the fused max-pool a batch must hold whole pairs of planes
            constexpr int TB = POOL ? ((TB_MAX >= TZ ? TZ : TB_MAX) & ~1)
                                    : (TB_MAX >= TZ ? TZ : (TB_MAX >= (TZ + 1) / 2 ? (TZ + 1) / 2 : 1));
            static_assert(!POOL || (TB >= 2 && TZ % 2 == 0 && TY % 2 == 0 && TX == 16), "pooled tile shape");
            char* wl = reinterpret_cast<char*>(lds) + wave * (TB * 32 * RECP);
            char* const dplane = static_cast<char*>(a.dst) +
                                 ((size_t)cur.nb * (a.cout / KC) + ntile0 * CPT) * patch_vox * 32;
            const int vv = lane_e >> 1, sub = lane_e & 1;
            const int po = wave * 32 + vv;
            const int ogy = cur.y0 + po / TX, ogx = cur.x0 + po % TX;
#pragma unroll
            for (int zb = 0; zb < TZ; zb += TB) {
#pragma unroll
                for (int z = zb; z < zb + TB && z < TZ; ++z) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = 8 * q + 4 * half_e;
                        // LeakyReLU with 0 <= slope <= 1 is max(v, slope * v)
                        float v0 = acc[z][4 * q + 0], v1 = acc[z][4 * q + 1];
                        float v2 = acc[z][4 * q + 2], v3 = acc[z][4 * q + 3];
                        v0 = leaky(v0, a.slope);
                        v1 = leaky(v1, a.slope);
                        v2 = leaky(v2, a.slope);
                        v3 = leaky(v3, a.slope);
                        store4<Tag>(wl + (z - zb) * (32 * RECP), (size_t)(r_e * RECP) / ES + cl, v0, v1, v2, v3);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int z = zb; z < zb + TB && z < TZ; ++z) {
                    const int gz = cur.z0 + z;
#if EXASPIM_BUFFER_STORES & 4
                    // (unconditional range-checked stores, see the direct epilogue above)
                    const bool tokyx = ogy < a.org[1] + a.ext[1] && ogx < a.org[2] + a.ext[2];
                    const unsigned tvoff = tokyx ? (unsigned)(ogy * a.w + ogx) * 32u + sub * 16u : kOutOfRange;
                    const __amdgpu_buffer_rsrc_t trsrc =
                        make_rsrc(dplane, gz < a.org[0] + a.ext[0] ? (size_t)CPT * patch_vox * 32 : (size_t)0);
#else
                    const bool ok = gz < a.org[0] + a.ext[0] && ogy < a.org[1] + a.ext[1] && ogx < a.org[2] + a.ext[2];
                    const size_t vox = ((size_t)gz * a.h + ogy) * a.w + ogx;
#endif
#pragma unroll
                    for (int ck = 0; ck < CPT; ++ck) {
                        const uint4 val = *reinterpret_cast<const uint4*>(
                            wl + (z - zb) * (32 * RECP) + vv * RECP + (ck * 2 + sub) * 16);
#if EXASPIM_BUFFER_STORES & 4
                        buf_store16(val, trsrc, tvoff,
                                    ((unsigned)ck * (unsigned)patch_vox + (unsigned)gz * (unsigned)plane_vox) * 32u);
#else
                        if (ok && (!(EXASPIM_ABLATE & 2) || val.x == 0x12345u))
                            *reinterpret_cast<uint4*>(dplane + ((size_t)ck * patch_vox + vox) * 32 + sub * 16) = val;
#endif
                    }
                }
                if (POOL) {
                    // MaxPool3d(2) of the planes in LDS: the wave's 2 rows x 16 voxels x TB planes
                    // give TB/2 x 8 pooled voxels; a piece is one 16-byte group of one of them,
                    // the maximum over its 2 x 2 x 2 source records (record = row * 16 + x).
                    constexpr int NP = (TB / 2) * 8 * CPT * 2;
                    const int pd = a.d >> 1, ph = a.h >> 1, pw2 = a.w >> 1;
                    const size_t pvox = (size_t)pd * ph * pw2;
                    char* const pplane = static_cast<char*>(a.pool_dst) +
                                         ((size_t)cur.nb * (a.cout / KC) + ntile0 * CPT) * pvox * 32;
#if EXASPIM_BUFFER_STORES & 4
                    const __amdgpu_buffer_rsrc_t prsrc = make_rsrc(pplane, (size_t)CPT * pvox * 32);
#endif
#pragma unroll
                    for (int p0 = 0; p0 < NP; p0 += 64) {
                        const int p = p0 + lane_e;
#if EXASPIM_BUFFER_STORES & 4
                        // a lane without a piece works on piece 0 and its store is dropped by the range
                        // check: no branch around the LDS reads or the store (see the direct epilogue)
                        const bool live0 = p < NP;
                        const int pc = live0 ? p : 0;
                        const int zp = pc / (8 * CPT * 2), rem = pc % (8 * CPT * 2);
                        const int ck = rem / 16, xp = (rem % 16) >> 1, sb = rem & 1;
                        const bool live = live0 && zb + 2 * zp + 1 < TZ;
                        const int zr = zb + 2 * zp + 1 < TZ ? zp : 0;     // (rows that exist in the batch)
                        {
                            const char* rec = wl + (2 * zr) * (32 * RECP) + (2 * xp) * RECP + (ck * 2 + sb) * 16;
#else
                        const int zp = p / (8 * CPT * 2), rem = p % (8 * CPT * 2);
                        const int ck = rem / 16, xp = (rem % 16) >> 1, sb = rem & 1;
                        if (p < NP && zb + 2 * zp + 1 < TZ) {
                            const char* rec = wl + (2 * zp) * (32 * RECP) + (2 * xp) * RECP + (ck * 2 + sb) * 16;
#endif
                            uint4 m = *reinterpret_cast<const uint4*>(rec);
                            if (ES == 2) m = key16(m);   // 16-bit types: compare order-preserving keys
#pragma unroll
                            for (int k = 1; k < 8; ++k) {
                                const uint4 v = *reinterpret_cast<const uint4*>(
                                    rec + (k >> 2) * (32 * RECP) + ((k >> 1) & 1) * 16 * RECP + (k & 1) * RECP);
                                m = ES == 2 ? maxkey16(m, key16(v)) : max16<Tag>(m, v);
                            }
                            if (ES == 2) m = key16(m);
                            const int qz = (cur.z0 + zb) / 2 + zp, qy = cur.y0 / 2 + wave, qx = cur.x0 / 2 + xp;
#if EXASPIM_BUFFER_STORES & 4
                            const unsigned pvoff =
                                live && qz < pd && qy < ph && qx < pw2
                                    ? (unsigned)((ck * (int)pvox + (qz * ph + qy) * pw2 + qx) * 32 + sb * 16)
                                    : kOutOfRange;
                            buf_store16(m, prsrc, pvoff, 0u);
#else
                            if (qz < pd && qy < ph && qx < pw2)
                                *reinterpret_cast<uint4*>(pplane + ((size_t)ck * pvox +
                                                                    ((size_t)qz * ph + qy) * pw2 + qx) * 32 + sb * 16) = m;
#endif
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        EXA_TRACE(14);
        if (!has_next) break;
        // the transposition buffers are free again (the direct epilogue never used them)
        if (!kLdsFreeEpilogue) __syncthreads();
        if (!(EXASPIM_STAGE_FIRST && kLdsFreeEpilogue)) stage_store();
        tile_id += t_step;
        cur = nxt;
    }
}

// ---- measured-and-not-adopted kernels ------------------------------------
// conv3x3x3_zpair and conv3x3x3_t16 are compiled only with -DEXASPIM_VARIANTS (make variant
// NAME=variants VFLAGS=-DEXASPIM_VARIANTS: tools/ and tests/test_gpu_parity.py's variant test load that
// build through EXASPIM_LIB); the product library carries neither them nor their switches.
#ifdef EXASPIM_VARIANTS
// ---- conv3x3x3_zpair: the z-column kernel on v_mfma_f32_16x16x32 (16-bit modes) ----------
// Under the package power cap the 16x16x32 shape sustains ~15 % more FLOP/s than 32x32x16
// at equal operand traffic (tools/mfma_shape.hip). Its K is 32: with
// 16-channel chunks in the LDS image that is two taps per instruction. Same tiles, same
// image, same staging as conv3x3x3_zpipe; what changes is how the 27 taps are walked:
//   * in-plane taps in pairs (0,3) (1,4) (2,5) (6,7) (g = dy * 3 + dx): the B operand of a
//     (pair, input plane, 16-voxel row) is ONE ds_read_b128 -- lanes 0-31 read the first
//     tap's 16 channels, lanes 32-63 the second tap's, HXP or 1 slot further -- and feeds
//     the three dz taps x two 16-cout halves = 6 MFMAs (96 cycles), the same reads per FLOP
//     as before;
//   * tap 8 is chained over two consecutive input planes (lanes 32-63 read PLANE slots
//     further): [X(p); X(p+1)] serves out planes p (weights [dz0; dz1]) and p+1 ([-; dz0]) in
//     a first pass over p = 0, 2, .., and p-2 ([dz2; -]) and p-1 ([dz1; dz2]) in a second
//     one: 14 instruction pairs per output plane where 13.5 would be ideal (3.7 %).
// A wave owns two rows of 16 voxels x TZ planes x 2 cout halves = the same 16 * TZ
// accumulator registers. Weight fragments come in the paired order plan.cpp packs
// (common.h, kPairedFrags = 32 per chunk, 32 KB in LDS).
// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- the step index of
// the tap loop must be a constant in every register-array subscript
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <typename Tag>
__device__ __forceinline__ void mma16(f32x4& acc, const uint4& wf, const uint4& xf);
// The accumulators are pinned to the accumulation registers ("+a") and updated in place:
// left to itself hipcc kept them in VGPRs, gave many MFMAs a destination different from
// their addend (40 accumulator quads live instead of 24) and spilled.
template <>
__device__ __forceinline__ void mma16<BF16Tag>(f32x4& acc, const uint4& wf, const uint4& xf) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                 : "+v"(acc) : "v"(__builtin_bit_cast(u32x4, wf)), "v"(__builtin_bit_cast(u32x4, xf)));
}
template <>
__device__ __forceinline__ void mma16<F16Tag>(f32x4& acc, const uint4& wf, const uint4& xf) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                 : "+v"(acc) : "v"(__builtin_bit_cast(u32x4, wf)), "v"(__builtin_bit_cast(u32x4, xf)));
}

// The same with the accumulator in the accumulation register file: for one-wave-per-SIMD
// kernels, whose 128 accumulators would otherwise crowd the operands out of the 256 VGPRs.
template <typename Tag>
__device__ __forceinline__ void mma16_agpr(f32x4& acc, const uint4& wf, const uint4& xf);
template <>
__device__ __forceinline__ void mma16_agpr<BF16Tag>(f32x4& acc, const uint4& wf, const uint4& xf) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                 : "+a"(acc) : "v"(__builtin_bit_cast(u32x4, wf)), "v"(__builtin_bit_cast(u32x4, xf)));
}
template <>
__device__ __forceinline__ void mma16_agpr<F16Tag>(f32x4& acc, const uint4& wf, const uint4& xf) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                 : "+a"(acc) : "v"(__builtin_bit_cast(u32x4, wf)), "v"(__builtin_bit_cast(u32x4, xf)));
}

template <typename Tag, int TZ, int TY, int MINW, int D, int HEAD = 0, bool POOL = false>
__global__ __launch_bounds__(TY * 16 * 2, MINW) void conv3x3x3_zpair(
    ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int TX = 16;
    constexpr int KC = 16, ES = 2;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HXP = HX;                // row stride (slots)
    constexpr int PLANE = HY * HXP;        // slots per halo plane
    constexpr int HVP = HZ * PLANE;        // slots per channel group
    // stride between the two channel-group planes: a multiple of 16 slots, so the lane
    // quarters of one ds_read_b128 group (16 voxels of group 0 next to 16 of group 1) fall
    // on complementary banks
    constexpr int GS = (HVP + 15) / 16 * 16;
    constexpr int NWAVES = TY / 2;
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int NPAIR = 2 * HY * HX;     // (column, group) pairs of the halo block
    constexpr int REM = NPAIR > NTHREADS ? NPAIR - NTHREADS : 0;
    constexpr int SEC = (REM * HZ + NTHREADS - 1) / NTHREADS;
    constexpr int NITEMS = HZ + SEC;       // halo pieces per thread
    constexpr int RECB = 32 * ES;
    constexpr int RECP = RECB + 16;        // padded LDS stride of the output transposition
    constexpr int EPI_UNITS = NWAVES * 32 * RECP / 16;
    constexpr int WUNITS = kPairedFrags * 64;   // the chunk's weight fragments in LDS
    constexpr int WITEMS = (WUNITS + NTHREADS - 1) / NTHREADS;
    constexpr int XUNITS = 2 * GS > EPI_UNITS ? 2 * GS : EPI_UNITS;
    constexpr int LDS_UNITS = XUNITS + WUNITS;
    constexpr int NSAME = 4 * HZ * 2;      // steps of the four in-plane pairs: (pair, plane, row)
    constexpr int NS = NSAME + 2 * TZ;     // + the two passes over the chained tap
    constexpr int R = D + 1;               // operand ring
    static_assert(TZ % 2 == 0 && TY % 2 == 0 && NPAIR <= 2 * NTHREADS, "tile shape");
    static_assert(NITEMS + WITEMS <= NS, "one staged piece per step");
    static_assert(WUNITS % NTHREADS == 0, "weight fragments split evenly over the threads");

    __shared__ __attribute__((aligned(16))) uint4 lds[LDS_UNITS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quarter = lane >> 4;         // K slice of the MFMA operands: (tap slot, channel group)
    const int lx = lane & 15;              // voxel column inside the 16-wide row

    const int total = tiles_z * tiles_y * tiles_x * a.n;
    int t_first, t_count, t_step;
    {
        const int q = total >> 3, rem = total & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        t_first = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
        t_count = q + (xcd < rem ? 1 : 0) - slot;        // tiles left from t_first on
        t_step = (gridDim.x + 7 - xcd) >> 3;             // workgroups on this XCD
    }
    if (t_count <= 0) return;
    const int ntiles = a.cout >> 5;
    const int ntile0 = blockIdx.y;

    struct Tile {
        int z0, y0, x0, nb;
    };
    auto tile_at = [&](int id) {
        // (the divisions run on the vector ALU; readfirstlane puts the wave-uniform results back
        // into scalar registers, where every later use -- selects, scalar load offsets -- wants them)
        Tile t;
        t.x0 = __builtin_amdgcn_readfirstlane(a.org[2] + (id % tiles_x) * TX); id /= tiles_x;
        t.y0 = __builtin_amdgcn_readfirstlane(a.org[1] + (id % tiles_y) * TY); id /= tiles_y;
        t.z0 = __builtin_amdgcn_readfirstlane(a.org[0] + (id % tiles_z) * TZ); id /= tiles_z;
        t.nb = __builtin_amdgcn_readfirstlane(id);
        return t;
    };

    // LDS slot of this lane's voxel of row 0 of the wave, tap (0,0), plane 0, in its channel
    // group; lanes 32-63 (second tap of a pair) sit one row, one column or one plane further
    const int cbase = (quarter & 1) * GS + (wave * 2) * HXP + lx;
    const int colH = cbase + (quarter >> 1) * HXP;
    const int col1 = cbase + (quarter >> 1);
    const int colP = cbase + (quarter >> 1) * PLANE;

    // ---- staging map (as in conv3x3x3_zpipe) ------------------------------------------
    // primary: thread t < NPAIR moves pair t = (column t / 2, group t & 1), all HZ planes;
    // secondary: the REM pairs beyond NTHREADS, piece q = t + k * NTHREADS is plane q / REM of
    // pair NTHREADS + q % REM. Nothing of this map is kept in registers: slots and global
    // offsets are derived from the thread index where they are needed (a few multiply-shifts
    // per piece). This kernel has no register to spare, and a spilled value comes back through
    // a scratch load whose wait also waits for every prefetch load and store still in flight.
    const int plane_vox = a.h * a.w;
    const size_t patch_vox = (size_t)a.d * plane_vox;
    // (the asm keeps the decoding next to its use: hoisted out of the loops it would live in registers)
    auto my_tid = [&]() { int v = tid; asm volatile("" : "+v"(v)); return v; };
    struct Item { int hz, hy, hx, kg; bool valid; };
    auto primary = [&]() {
        const int t = my_tid();
        return Item{0, (t >> 1) / HX, (t >> 1) % HX, t & 1, t < NPAIR};
    };
    auto secondary = [&](int k) {
        const int q = my_tid() + k * NTHREADS;
        const int pr = NTHREADS + q % (REM > 0 ? REM : 1), c = pr >> 1;
        return Item{q / (REM > 0 ? REM : 1), c / HX, c % HX, pr & 1, q < REM * HZ};
    };
    auto slot_of = [&](const Item& it) { return it.kg * GS + it.hz * PLANE + it.hy * HXP + it.hx; };
    // byte offset of an item inside a chunk plane of the tile at (z0, y0, x0); primary items get
    // their plane through the scalar offset (z = 0 here)
    auto voff_of = [&](const Item& it, int z0, int y0, int x0, bool with_z) {
        const int gz = z0 + it.hz - 1, gy = y0 + it.hy - 1, gx = x0 + it.hx - 1;
        const bool in = it.valid && (!with_z || (unsigned)gz < (unsigned)a.d) &&
                        (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
        return in ? (unsigned)(((with_z ? gz : 0) * a.h + gy) * a.w + gx) * 32u + it.kg * 16u : kOutOfRange;
    };

    __shared__ __attribute__((aligned(16))) float bias_s[32];
    if (tid < 32) bias_s[tid] = a.bias[ntile0 * 32 + tid];
    __shared__ __attribute__((aligned(16))) float head_s[HEAD > 0 ? HEAD * 32 + 4 : 4];
    if (HEAD > 0) {
        if (tid < HEAD * 32) head_s[tid] = a.head_w[tid];
        if (tid < HEAD) head_s[HEAD * 32 + tid] = a.head_b[tid];
    }

    const int nchunks = (a.ca + a.cb) / KC;
    uint4 stg[NITEMS + WITEMS];  // halo pieces, then weight fragments
    uint4* const wlds = lds + XUNITS;
    const __amdgpu_buffer_rsrc_t wrsrc =
        make_rsrc(a.weights_paired, (size_t)nchunks * kPairedFrags * ntiles * 1024);

    struct ChunkSrc {
        __amdgpu_buffer_rsrc_t rsrc;
        unsigned cbase;
    };
    auto chunk_src = [&](int c, int nb) {
        const char* src;
        int cs, ch0;
        if (c * KC < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * KC;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * KC - a.ca;
        }
        const size_t patchb = patch_vox * cs * ES;
        return ChunkSrc{make_rsrc(src + (size_t)nb * patchb, patchb),
                        (unsigned)(ch0 / KC) * (unsigned)patch_vox * 32u};
    };
    // piece i of chunk c of the tile at (z0, y0, x0): global -> stg[i]; p_voff = voff_of(primary)
    auto load_piece = [&](const ChunkSrc& cs, int c, int z0, int y0, int x0, unsigned p_voff, int i) {
        if (i < NITEMS) {
            if (i < HZ) {
                const int gz = z0 + i - 1;  // wave-uniform
                stg[i] = (unsigned)gz < (unsigned)a.d
                             ? buf_load16(cs.rsrc, p_voff, cs.cbase + (unsigned)gz * plane_vox * 32u)
                             : make_uint4(0, 0, 0, 0);
            } else {
                stg[i] = buf_load16(cs.rsrc, voff_of(secondary(i - HZ), z0, y0, x0, true), cs.cbase);
            }
        } else {
            const int it = i - NITEMS;   // fragments it * NWAVES + wave
            stg[i] = buf_load16(wrsrc, (unsigned)lane * 16u,
                                ((c * kPairedFrags + it * NWAVES + wave) * ntiles + ntile0) * 1024);
        }
    };
    auto stage_store = [&]() {
        {
            const Item it = primary();
            if (it.valid) {
                const int p_slot = slot_of(it);
#pragma unroll
                for (int hz = 0; hz < HZ; ++hz) lds[p_slot + hz * PLANE] = stg[hz];
            }
        }
#pragma unroll
        for (int k = 0; k < SEC; ++k) {
            const Item it = secondary(k);
            if (it.valid) lds[slot_of(it)] = stg[HZ + k];
        }
#pragma unroll
        for (int it = 0; it < WITEMS; ++it) wlds[tid + it * NTHREADS] = stg[NITEMS + it];
    };

    int tile_id = t_first;
    Tile cur = tile_at(tile_id);
    {
        const ChunkSrc cs0 = chunk_src(0, cur.nb);
        const unsigned pv = voff_of(primary(), cur.z0, cur.y0, cur.x0, false);
#pragma unroll
        for (int i = 0; i < NITEMS + WITEMS; ++i) load_piece(cs0, 0, cur.z0, cur.y0, cur.x0, pv, i);
    }
    stage_store();

    // ---- step tables (all compile-time after unrolling) ------------------------------
    // LDS slot offset (beyond colH / col1 / colP) of the B operand of step s
    constexpr auto x_off = [](int s) constexpr {
        if (s < NSAME) {
            const int pair = s / (2 * HZ), p = (s % (2 * HZ)) / 2, v = s % 2;
            const int dy0 = pair < 3 ? 0 : 2, dx0 = pair < 3 ? pair : 0;
            return p * PLANE + (v + dy0) * HXP + dx0;
        }
        const int c = s - NSAME;
        const int p = c < TZ ? (c / 2) * 2 : ((c - TZ) / 2) * 2 + 2, v = c % 2;
        return p * PLANE + (v + 2) * HXP + 2;
    };
    // which of colH / col1 / colP the step reads through
    constexpr auto x_col = [](int s) constexpr { return s >= NSAME ? 2 : s / (2 * HZ) < 3 ? 0 : 1; };

    for (;;) {
        __syncthreads();   // this tile's first chunk (and, the first time, the bias) is in LDS
        // accumulators: register k of acc[z][v][t] is channel 16 t + 4 quarter + k of voxel (row v, lx)
        f32x4 acc[TZ][2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 b = *reinterpret_cast<const float4*>(bias_s + 16 * t + 4 * quarter);
#pragma unroll
            for (int z = 0; z < TZ; ++z)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    acc[z][v][t][0] = b.x; acc[z][v][t][1] = b.y;
                    acc[z][v][t][2] = b.z; acc[z][v][t][3] = b.w;
                }
        }

        t_count -= t_step;
        const bool has_next = t_count > 0;
        Tile nxt = cur;
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            const bool pre = more || has_next;
            if (!more && has_next) nxt = tile_at(tile_id + t_step);
            // the prefetch target: the next chunk of this tile, or the first chunk of the next tile
            Tile tgt;
            tgt.z0 = more ? cur.z0 : nxt.z0; tgt.y0 = more ? cur.y0 : nxt.y0;
            tgt.x0 = more ? cur.x0 : nxt.x0; tgt.nb = more ? cur.nb : nxt.nb;
            const ChunkSrc csn = chunk_src(more ? c + 1 : 0, tgt.nb);
            const int cn = more ? c + 1 : 0;
            const unsigned pvn = voff_of(primary(), tgt.z0, tgt.y0, tgt.x0, false);
            uint4 xr[R];          // operand ring: fragment of step s lives in xr[s % R]
            uint4 wb[3][2];       // weight fragments (dz, cout half) of the current pair / chain kinds
            auto wfrag = [&](int f) { return wlds[f * 64 + lane]; };
            auto xread = [&](auto S) {
                constexpr int s = decltype(S)::value;
                constexpr int which = x_col(s), off = x_off(s);
                return lds[(which == 0 ? colH : which == 1 ? col1 : colP) + off];
            };
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                for (int t = 0; t < 2; ++t) wb[dz][t] = wfrag(dz * 2 + t);
            static_for<0, D>([&](auto S) { xr[decltype(S)::value % R] = xread(S); });
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NS>([&](auto S) {
                constexpr int s = decltype(S)::value;
                if constexpr (s + D < NS) xr[(s + D) % R] = xread(std::integral_constant<int, s + D>{});
                if constexpr (s < NITEMS + WITEMS) {
                    if (pre) load_piece(csn, cn, tgt.z0, tgt.y0, tgt.x0, pvn, s);
                }
                if constexpr (s < NSAME) {
                    constexpr int pair = s / (2 * HZ), p = (s % (2 * HZ)) / 2, v = s % 2;
                    // weight slot dz is free once plane dz + TZ - 1 is done: the next pair's
                    // fragments (after the last pair: chain kinds 0 and 2) move in
                    if constexpr (v == 0 && p >= TZ) {
                        constexpr int dz = p - TZ;    // 0 or 1
                        constexpr int f = pair < 3 ? (pair + 1) * 6 + dz * 2 : 24 + (dz == 0 ? 0 : 2) * 2;
                        wb[dz][0] = wfrag(f);
                        wb[dz][1] = wfrag(f + 1);
                    }
                    if constexpr (v == 0 && p == 0 && pair > 0) {
                        wb[2][0] = wfrag(pair * 6 + 4);
                        wb[2][1] = wfrag(pair * 6 + 5);
                    }
                    if constexpr (p >= 0 && p < TZ) {
                        mma16<Tag>(acc[p][v][0], wb[0][0], xr[s % R]);
                        mma16<Tag>(acc[p][v][1], wb[0][1], xr[s % R]);
                    }
                    if constexpr (p - 1 >= 0 && p - 1 < TZ) {
                        mma16<Tag>(acc[p - 1][v][0], wb[1][0], xr[s % R]);
                        mma16<Tag>(acc[p - 1][v][1], wb[1][1], xr[s % R]);
                    }
                    if constexpr (p - 2 >= 0 && p - 2 < TZ) {
                        mma16<Tag>(acc[p - 2][v][0], wb[2][0], xr[s % R]);
                        mma16<Tag>(acc[p - 2][v][1], wb[2][1], xr[s % R]);
                    }
                } else if constexpr (s < NSAME + TZ) {
                    // chain, first pass: [X(p); X(p+1)] -> out p (kind 0, wb[0]) and p + 1 (kind 2, wb[1])
                    constexpr int c1 = s - NSAME, p = (c1 / 2) * 2, v = c1 % 2;
                    if constexpr (c1 == 0) {   // kind 1 for the second pass: wb[2] is free since the last pair ended
                        wb[2][0] = wfrag(24 + 1 * 2);
                        wb[2][1] = wfrag(24 + 1 * 2 + 1);
                    }
                    mma16<Tag>(acc[p][v][0], wb[0][0], xr[s % R]);
                    mma16<Tag>(acc[p][v][1], wb[0][1], xr[s % R]);
                    if constexpr (c1 == TZ - 1) {   // last use of kind 0: kind 3 takes its registers
                        wb[0][0] = wfrag(24 + 3 * 2);
                        wb[0][1] = wfrag(24 + 3 * 2 + 1);
                    }
                    mma16<Tag>(acc[p + 1][v][0], wb[1][0], xr[s % R]);
                    mma16<Tag>(acc[p + 1][v][1], wb[1][1], xr[s % R]);
                } else {
                    // chain, second pass: [X(p); X(p+1)], p = 2, 4, .. -> out p - 2 (kind 1, wb[2]) and
                    // p - 1 (kind 3, wb[0])
                    constexpr int c2 = s - NSAME - TZ, p = (c2 / 2) * 2 + 2, v = c2 % 2;
                    mma16<Tag>(acc[p - 2][v][0], wb[2][0], xr[s % R]);
                    mma16<Tag>(acc[p - 2][v][1], wb[2][1], xr[s % R]);
                    mma16<Tag>(acc[p - 1][v][0], wb[0][0], xr[s % R]);
                    mma16<Tag>(acc[p - 1][v][1], wb[0][1], xr[s % R]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();   // every wave is done reading this chunk's image
            if (more) {
                stage_store();
                __syncthreads();
            }
        }

        const int row0 = wave * 2;     // the wave's rows inside the tile
        if (HEAD > 0) {
            // ---- fused head: OutConv 1x1x1 (+ sigmoid) on the accumulators -----------
            // a lane holds 8 of its voxel's 32 channels (16 t + 4 quarter + k): partial dot
            // products, completed over the four lane quarters
            const size_t plane = (size_t)a.h * a.w;
#pragma unroll
            for (int z = 0; z < TZ; ++z) {
                const int gz = cur.z0 + z;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int gy = cur.y0 + row0 + v, gx = cur.x0 + lx;
                    const bool ok = gz < a.org[0] + a.ext[0] && gy < a.org[1] + a.ext[1] &&
                                    gx < a.org[2] + a.ext[2];
                    float part[HEAD > 0 ? HEAD : 1];
#pragma unroll
                    for (int o = 0; o < HEAD; ++o) part[o] = 0.f;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float v0 = leaky(acc[z][v][t][0], a.slope), v1 = leaky(acc[z][v][t][1], a.slope);
                        const float v2 = leaky(acc[z][v][t][2], a.slope), v3 = leaky(acc[z][v][t][3], a.slope);
#pragma unroll
                        for (int o = 0; o < HEAD; ++o) {
                            const float4 hw = *reinterpret_cast<const float4*>(head_s + o * 32 + 16 * t + 4 * quarter);
                            part[o] = fmaf(v3, hw.w, fmaf(v2, hw.z, fmaf(v1, hw.y, fmaf(v0, hw.x, part[o]))));
                        }
                    }
#pragma unroll
                    for (int o = 0; o < HEAD; ++o) {
                        float s1 = part[o] + __shfl_xor(part[o], 16);
                        float tsum = s1 + __shfl_xor(s1, 32) + head_s[HEAD * 32 + o];
                        if (a.head_sigmoid) tsum = 1.f / (1.f + expf(-tsum));
                        // outputs are dealt to the lane quarters so all of them store
                        if (ok && o == quarter)
                            a.head_out[(((size_t)cur.nb * HEAD + o) * a.d + gz) * plane + (size_t)gy * a.w + gx] = tsum;
                    }
                }
            }
        } else {
            // ---- epilogue: LeakyReLU, transposed through LDS (as in conv3x3x3_zpipe) --------
            constexpr int CPT = RECB / 32;
            constexpr int TB_MAX = LDS_UNITS * 16 / (NWAVES * 32 * RECP);
            constexpr int TB = POOL ? ((TB_MAX >= TZ ? TZ : TB_MAX) & ~1)
                                    : (TB_MAX >= TZ ? TZ : (TB_MAX >= (TZ + 1) / 2 ? (TZ + 1) / 2 : 1));
            static_assert(!POOL || (TB >= 2 && TZ % 2 == 0 && TY % 2 == 0), "pooled tile shape");
            char* wl = reinterpret_cast<char*>(lds) + wave * (TB * 32 * RECP);
            char* const dplane = static_cast<char*>(a.dst) +
                                 ((size_t)cur.nb * (a.cout / KC) + ntile0 * CPT) * patch_vox * 32;
            const int vv = lane >> 1, sub = lane & 1;
            const int po = wave * 32 + vv;
            const int ogy = cur.y0 + po / TX, ogx = cur.x0 + po % TX;
#pragma unroll
            for (int zb = 0; zb < TZ; zb += TB) {
#pragma unroll
                for (int z = zb; z < zb + TB && z < TZ; ++z) {
#pragma unroll
                    for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int rr = v * 16 + lx, cl = 16 * t + 4 * quarter;
                            store4<Tag>(wl + (z - zb) * (32 * RECP), (size_t)(rr * RECP) / ES + cl,
                                        leaky(acc[z][v][t][0], a.slope), leaky(acc[z][v][t][1], a.slope),
                                        leaky(acc[z][v][t][2], a.slope), leaky(acc[z][v][t][3], a.slope));
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int z = zb; z < zb + TB && z < TZ; ++z) {
                    const int gz = cur.z0 + z;
                    const bool ok = gz < a.org[0] + a.ext[0] && ogy < a.org[1] + a.ext[1] && ogx < a.org[2] + a.ext[2];
                    const size_t vox = ((size_t)gz * a.h + ogy) * a.w + ogx;
#pragma unroll
                    for (int ck = 0; ck < CPT; ++ck) {
                        const uint4 val = *reinterpret_cast<const uint4*>(
                            wl + (z - zb) * (32 * RECP) + vv * RECP + (ck * 2 + sub) * 16);
                        if (ok) *reinterpret_cast<uint4*>(dplane + ((size_t)ck * patch_vox + vox) * 32 + sub * 16) = val;
                    }
                }
                if (POOL) {
                    constexpr int NP = (TB / 2) * 8 * CPT * 2;
                    const int pd = a.d >> 1, ph = a.h >> 1, pw2 = a.w >> 1;
                    const size_t pvox = (size_t)pd * ph * pw2;
                    char* const pplane = static_cast<char*>(a.pool_dst) +
                                         ((size_t)cur.nb * (a.cout / KC) + ntile0 * CPT) * pvox * 32;
#pragma unroll
                    for (int p0 = 0; p0 < NP; p0 += 64) {
                        const int p = p0 + lane;
                        const int zp = p / (8 * CPT * 2), rem = p % (8 * CPT * 2);
                        const int ck = rem / 16, xp = (rem % 16) >> 1, sb = rem & 1;
                        if (p < NP && zb + 2 * zp + 1 < TZ) {
                            const char* rec = wl + (2 * zp) * (32 * RECP) + (2 * xp) * RECP + (ck * 2 + sb) * 16;
                            uint4 m = key16(*reinterpret_cast<const uint4*>(rec));
#pragma unroll
                            for (int k = 1; k < 8; ++k) {
                                const uint4 vq = *reinterpret_cast<const uint4*>(
                                    rec + (k >> 2) * (32 * RECP) + ((k >> 1) & 1) * 16 * RECP + (k & 1) * RECP);
                                m = maxkey16(m, key16(vq));
                            }
                            m = key16(m);
                            const int qz = (cur.z0 + zb) / 2 + zp, qy = cur.y0 / 2 + wave, qx = cur.x0 / 2 + xp;
                            if (qz < pd && qy < ph && qx < pw2)
                                *reinterpret_cast<uint4*>(pplane + ((size_t)ck * pvox +
                                                                    ((size_t)qz * ph + qy) * pw2 + qx) * 32 + sb * 16) = m;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!has_next) break;
        if (HEAD == 0) __syncthreads();   // the transposition buffers are free again
        stage_store();
        tile_id += t_step;
        cur = nxt;
    }
}

// ---- conv3x3x3_t16: 64-cout layers on v_mfma_f32_16x16x32 (16-bit modes, 16-wide levels) ------
// Under the package power cap the 16x16x32 shape sustains ~15 % more FLOP/s than 32x32x16 at
// equal operand traffic (tools/mfma_shape.hip). Its K is 32 = two 16-channel chunks, so the LDS
// image holds a PAIR of chunks (four 16-byte channel-group planes); double-buffered and filled by
// LDS-DMA that is 2 x 69.6 KB, i.e. one workgroup of four waves per CU with the whole register
// file (512 per lane) to itself. With nobody else on the CU to hide a prologue or an epilogue
// behind, the workgroup is persistent: the image of the next tile's first pair is fetched during
// the last pair of the current tile, and the epilogue goes through a small private LDS region
// per wave, so the MFMA pipe only idles for the epilogue's own instructions.
// A wave owns one z-plane of the 4 x 8 x 16 tile = 8 rows of 16 voxels (8 B fragments per tap
// and pair, each one ds_read_b128: lanes 16 q .. 16 q + 15 read plane q) x 64 couts (4 A
// fragments per tap and pair, streamed from L2 through a register ring): 32 MFMAs of 16 cycles
// per 8 + 4 operand fragments. A B fragment's register is refilled for the next tap as soon as
// its four MFMAs have been issued. Weight fragments in the K = 32 order of plan.cpp
// (ConvLayer::w3_off). Whole patches only (no region), cout == 64, ca and cb multiples of 32.
template <typename Tag, int PD>
__global__ __launch_bounds__(256, 1) void conv3x3x3_t16(ConvArgs a, int tiles_z, int tiles_y, int tiles_x) {
    constexpr int TZ = 4, TY = 8, TX = 16;
    constexpr int ES = 2;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HVR = HZ * HY * HX;               // halo voxels = slots of a plane that hold voxels
    constexpr int HV = (HVR + 63) / 64 * 64;        // plane stride (whole 64-slot DMA blocks)
    constexpr int NBLK = HV / 64;
    constexpr int IMG = 4 * HV;                     // slots of a pair image: [chunk 2][group 2][HV]
    constexpr int NWAVES = 4;
    constexpr int NDMA = (4 * NBLK + NWAVES - 1) / NWAVES;   // DMA blocks per wave and pair image
    constexpr int RECB = 64 * ES;                   // bytes of one voxel's 64-cout record
    constexpr int RECP = RECB + 16;
    constexpr int EPI_UNITS = 32 * RECP / 16;       // per wave: two rows of 16 voxel records
    constexpr int ISSUE_T = 26 - PD > 0 ? 26 - PD : 0;

    __shared__ __attribute__((aligned(16))) uint4 lds[2 * IMG + NWAVES * EPI_UNITS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = the tile's z-plane this wave computes
    const int quarter = lane >> 4, lx = lane & 15;

    // Workgroup b runs on XCD b % 8; every XCD gets a contiguous range of tiles and its
    // workgroups walk it side by side, so neighbouring tiles meet in the same L2.
    const int ntiles = tiles_z * tiles_y * tiles_x * a.n;
    const int nwg_xcd = gridDim.x >> 3;             // the launcher keeps gridDim.x a multiple of 8
    int tile, tile_end;
    {
        const int q = ntiles >> 3, rem = ntiles & 7;
        const int xcd = blockIdx.x & 7;
        const int first = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
        tile = first + (blockIdx.x >> 3);
        tile_end = first + (xcd < rem ? q + 1 : q);
    }
    if (tile >= tile_end) return;

    const size_t patch_vox = (size_t)a.d * a.h * a.w;
    const int npairs = (a.ca + a.cb) / 32;

    // slot of this lane's voxel of row 0 of the wave's plane, tap (0,0,0), in plane "quarter"
    const int xbase = quarter * HV + (wave * HY) * HX + lx;

    // DMA block j = wave + k * NWAVES of a pair image: plane j / NBLK, slots (j % NBLK) * 64 + lane
    unsigned dvoff[NDMA];
    auto aim = [&](int t, int& nb, int& z0, int& y0, int& x0) {
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; t /= tiles_y;
        const int tz = t % tiles_z;
        nb = t / tiles_z;
        z0 = tz * TZ; y0 = ty * TY; x0 = tx * TX;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int j = wave + k * NWAVES;
            const int pl = j / NBLK, slot = (j % NBLK) * 64 + lane;
            const int hz = slot / (HY * HX), hy = (slot / HX) % HY, hx = slot % HX;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = j < 4 * NBLK && slot < HVR && (unsigned)gz < (unsigned)a.d &&
                            (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
            // the chunk of the pair (pl >> 1) rides in the scalar offset
            dvoff[k] = ok ? (unsigned)((gz * a.h + gy) * a.w + gx) * 32u + (pl & 1) * 16u : kOutOfRange;
        }
    };
    const unsigned lds_base = (unsigned)(size_t)lds;
    auto dma_load = [&](int nb, int pr, int buf) {
        const int c = 2 * pr;                       // first chunk of the pair (both in one source)
        const char* src;
        int cs, ch0;
        if (c * 16 < a.ca) {
            src = static_cast<const char*>(a.src_a); cs = a.ca; ch0 = c * 16;
        } else {
            src = static_cast<const char*>(a.src_b); cs = a.cb; ch0 = c * 16 - a.ca;
        }
        const size_t patchb = patch_vox * cs * ES;
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(src + (size_t)nb * patchb, patchb);
        const unsigned cbase = (unsigned)(ch0 / 16) * (unsigned)patch_vox * 32u;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int j = wave + k * NWAVES;
            if ((4 * NBLK) % NWAVES == 0 || j < 4 * NBLK) {   // wave-uniform
                const int pl = j / NBLK;
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    lds_base + (unsigned)((buf * IMG + pl * HV + (j % NBLK) * 64) * 16));
                const unsigned soff = cbase + (unsigned)(pl >> 1) * (unsigned)patch_vox * 32u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(dvoff[k]), "s"(rsrc), "s"(soff) : "memory", "m0");   // m0 is not allocatable: nothing of the compiler's lives in it
            }
        }
    };

    uint4 wring[PD + 1][4];
    auto weights_at = [&](int pr, int t) {
        return static_cast<const uint4*>(a.weights_k32) + ((size_t)pr * 27 + t) * 4 * 64 + lane;
    };
    auto prime_weights = [&](int pr) {
#pragma unroll
        for (int t = 0; t < PD; ++t) {
            const uint4* wp = weights_at(pr, t);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) wring[t][ct] = wp[ct * 64];
        }
    };

    float4 bias4[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bias4[ct] = *reinterpret_cast<const float4*>(a.bias + ct * 16 + 4 * quarter);

    int nb, z0, y0, x0;
    aim(tile, nb, z0, y0, x0);
    dma_load(nb, 0, 0);
    prime_weights(0);
    int buf = 0;   // the LDS buffer the pair about to be computed sits in

    for (;;) {
        // accumulators: register k of acc[vg][ct] is cout 16 ct + 4 quarter + k of voxel (row vg, lx)
        f32x4 acc[8][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int vg = 0; vg < 8; ++vg) {
                acc[vg][ct][0] = bias4[ct].x; acc[vg][ct][1] = bias4[ct].y;
                acc[vg][ct][2] = bias4[ct].z; acc[vg][ct][3] = bias4[ct].w;
            }
        const int next = tile + nwg_xcd;
        const bool more_tiles = next < tile_end;    // workgroup-uniform
        const int cz0 = z0, cy0 = y0, cx0 = x0, cnb = nb;

        for (int pr = 0; pr < npairs; ++pr) {
            // this wave's DMA blocks (and weights, and the last tile's stores) have landed; after
            // the barrier so have everyone's, and nobody still reads the other buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const uint4* const img = lds + buf * IMG;
            const bool more = pr + 1 < npairs;
            uint4 bf[8];
#pragma unroll
            for (int vg = 0; vg < 8; ++vg) bf[vg] = img[xbase + vg * HX];
            static_for<0, 27>([&](auto T) {
                constexpr int t = decltype(T)::value;
                if constexpr (t + PD < 27) {
                    const uint4* wp = weights_at(pr, t + PD);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) wring[(t + PD) % (PD + 1)][ct] = wp[ct * 64];
                }
                if constexpr (t == ISSUE_T) {
                    if (more) {
                        dma_load(cnb, pr + 1, buf ^ 1);
                    } else if (more_tiles) {
                        aim(next, nb, z0, y0, x0);
                        dma_load(nb, 0, buf ^ 1);
                    }
                }
                constexpr int tapn = ((t + 1) / 9 * HY + ((t + 1) / 3) % 3) * HX + (t + 1) % 3;
#pragma unroll
                for (int vg = 0; vg < 8; ++vg) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) mma16_agpr<Tag>(acc[vg][ct], wring[t % (PD + 1)][ct], bf[vg]);
                    // this row's fragment of the next tap takes over the register
                    if constexpr (t + 1 < 27) bf[vg] = img[xbase + vg * HX + tapn];
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            prime_weights(more ? pr + 1 : 0);
            buf ^= 1;
        }

        // ---- epilogue: LeakyReLU, two rows at a time transposed through the wave's LDS region ----
        {
            char* wl = reinterpret_cast<char*>(lds + 2 * IMG + wave * EPI_UNITS);
            const int vv = lane >> 1, sub = lane & 1;   // a store instruction = one chunk plane's 32 records
            const int gz = cz0 + wave;
            char* const dplane = static_cast<char*>(a.dst) + (size_t)cnb * 4 * patch_vox * 32;
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const f32x4 v = acc[2 * mg + r][ct];
                        store4<Tag>(wl, (size_t)((r * 16 + lx) * RECP) / ES + 16 * ct + 4 * quarter,
                                    leaky(v[0], a.slope), leaky(v[1], a.slope), leaky(v[2], a.slope),
                                    leaky(v[3], a.slope));
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int gy = cy0 + 2 * mg + vv / 16, gx = cx0 + vv % 16;
                const bool ok = gz < a.d && gy < a.h && gx < a.w;
                const size_t vox = ((size_t)gz * a.h + gy) * a.w + gx;
#pragma unroll
                for (int ck = 0; ck < 4; ++ck) {
                    const uint4 val = *reinterpret_cast<const uint4*>(wl + vv * RECP + (ck * 2 + sub) * 16);
                    if (ok) *reinterpret_cast<uint4*>(dplane + ((size_t)ck * patch_vox + vox) * 32 + sub * 16) = val;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!more_tiles) break;
        tile = next;
    }
}

#endif  // EXASPIM_VARIANTS

#ifdef EXASPIM_TRACE
int g_variant = 0;   // tools/conv_trace.hip: 3/5/6 = operand prefetch distance, +10 = one tile per workgroup
#endif

// workgroup slots of the device for a kernel that runs MINW workgroups per CU
static int resident_workgroups(int per_cu) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    return cus * per_cu;
}

template <typename Tag, int TZ, int TY, int TX, int MINW, int D, int HEAD = 0, bool POOL = false>
static int launch_zpipe(const ConvArgs& a, hipStream_t stream) {
    // tiles cover the voxels the caller needs, [org, org + ext) on every axis
    const int tz = (a.ext[0] + TZ - 1) / TZ, ty = (a.ext[1] + TY - 1) / TY, tx = (a.ext[2] + TX - 1) / TX;
    const long long blocks = (long long)tz * ty * tx * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        set_error("conv: grid of %lld blocks out of range", blocks);
        return EXASPIM_E_INVALID;
    }
    // persistent workgroups: as many as the device holds at once (a multiple of the 8
    // XCDs), each walking its share of the tile list with cross-tile prefetch
    const int slices = a.cout / 32;
    long long wgs = resident_workgroups(MINW) / slices / 8 * 8;
    if (wgs < 8) wgs = 8;
    if (wgs > blocks) wgs = blocks;
#ifdef EXASPIM_TRACE
    if (g_variant >= 10 && g_variant < 20) wgs = blocks;
    // (tools/conv_trace.hip: one workgroup per CU shows what a wave's tap loop does with the matrix pipe to itself)
    if (const char* e = getenv("EXASPIM_TRACE_WGS_PER_CU")) wgs = resident_workgroups(1) * (long long)atoi(e) / slices / 8 * 8;
#endif
    dim3 grid((unsigned)wgs, slices);
    conv3x3x3_zpipe<Tag, TZ, TY, TX, MINW, D, HEAD, POOL><<<grid, TY * TX * 2, 0, stream>>>(a, tz, ty, tx);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

#ifdef EXASPIM_VARIANTS
template <typename Tag, int TZ, int D, int HEAD = 0, bool POOL = false>
static int launch_zpair(const ConvArgs& a, hipStream_t stream) {
    constexpr int TY = 8, TX = 16, MINW = 2;
    const int tz = (a.ext[0] + TZ - 1) / TZ, ty = (a.ext[1] + TY - 1) / TY, tx = (a.ext[2] + TX - 1) / TX;
    const long long blocks = (long long)tz * ty * tx * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        set_error("conv: grid of %lld blocks out of range", blocks);
        return EXASPIM_E_INVALID;
    }
    const int slices = a.cout / 32;
    long long wgs = resident_workgroups(MINW) / slices / 8 * 8;
    if (wgs < 8) wgs = 8;
    if (wgs > blocks) wgs = blocks;
    dim3 grid((unsigned)wgs, slices);
    conv3x3x3_zpair<Tag, TZ, TY, MINW, D, HEAD, POOL><<<grid, TY * TX * 2, 0, stream>>>(a, tz, ty, tx);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

template <typename Tag, int TZ>
static int launch_zpair_head(const ConvArgs& a, hipStream_t stream) {
    if (a.head_out && a.cout == 32) {
        switch (a.head_oc) {
            case 1: return launch_zpair<Tag, TZ, 4, 1>(a, stream);
            case 2: return launch_zpair<Tag, TZ, 4, 2>(a, stream);
            case 3: return launch_zpair<Tag, TZ, 4, 3>(a, stream);
            case 4: return launch_zpair<Tag, TZ, 4, 4>(a, stream);
        }
    }
    if (a.pool_dst) return launch_zpair<Tag, TZ, 4, 0, true>(a, stream);
    return launch_zpair<Tag, TZ, 4>(a, stream);
}

// the 16x16x32 z-column kernel exists for the 16-bit types only
template <typename Tag> struct HasPaired { static constexpr bool value = true; };
template <> struct HasPaired<F32Tag> { static constexpr bool value = false; };

// EXASPIM_ZPAIR=1 routes the 32-cout-slice layers of the 16-bit modes to conv3x3x3_zpair. Off by
// default: measured on MI355X it is not faster than conv3x3x3_zpipe (stand-alone 0.773 vs 0.754 ms
// on the inc.3 shape, 1.369 vs 1.334 ms on the up4.0 shape; a 1024^3 step 1.505 vs 1.476 s), and its
// sums are associated differently from the thin-tile kernel's, so the trimmed forward would no
// longer match the full one bit for bit next to the thin remainders.
static bool paired_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("EXASPIM_ZPAIR");
        on = e && e[0] == '1';
    }
    return on != 0;
}

#endif  // EXASPIM_VARIANTS

// Split-K reduction: adds the float32 partial sums of the chunk ranges in range order, then
// bias, LeakyReLU and the conversion, and writes four channels of one voxel in the blocked
// layout. One thread per (voxel, 4 channels).
template <typename Tag>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                            const float* __restrict__ bias,
                                                            void* __restrict__ dst, size_t nvox_all,
                                                            size_t patch_vox, int cout, int ksplit,
                                                            float slope) {
    constexpr int ES = 16 / Tag::kG;
    constexpr int KC = 2 * Tag::kG;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int quads = cout >> 2;
    if (i >= nvox_all * quads) return;
    const size_t v = i / quads;
    const int c = (int)(i - v * quads) * 4;
    float4 s = *reinterpret_cast<const float4*>(bias + c);
    for (int k = 0; k < ksplit; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(partial + ((size_t)k * nvox_all + v) * cout + c);
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    const size_t nb = v / patch_vox, vox = v - nb * patch_vox;
    char* out = static_cast<char*>(dst) + (((size_t)nb * (cout / KC) + c / KC) * patch_vox + vox) * 32 +
                (c % KC) * ES;
    store4<Tag>(out, 0, leaky(s.x, slope), leaky(s.y, slope), leaky(s.z, slope), leaky(s.w, slope));
}

// ---- host side: pick a tile configuration per layer -----------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int MT, int NT, int MINW, int PD = 3,
          bool ZORD = false, bool DMA = false, bool POOL = false>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
    constexpr int NWG = WAVES_N * NT * 32;
    if (a.cout % NWG != 0) {
        set_error("conv: cout %d not a multiple of the %d-channel tile", a.cout, NWG);
        return EXASPIM_E_INVALID;
    }
    const int tz = cdiv(a.ext[0], TZ), ty = cdiv(a.ext[1], TY), tx = cdiv(a.ext[2], TX);
    const long long blocks = (long long)tz * ty * tx * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        set_error("conv: grid of %lld blocks out of range", blocks);
        return EXASPIM_E_INVALID;
    }
    const bool whole = a.ext[0] == a.d && a.ext[1] == a.h && a.ext[2] == a.w;
    // Split-K when a launch of a nominal batch (16 patches) cannot give every CU two
    // workgroups: up to 4 ranges of chunks (one at least per range), if the scratch holds
    // the partial sums. The split changes the order in which a voxel's products are
    // summed, so it is a function of the layer and the patch size only, never of the batch
    // size: a patch gets the same bits whichever batch it travels in (predict_streaming
    // relies on that; short batches merely fill the device less well).
    ConvArgs b = a;
    b.ksplit = 1;
    constexpr int kNominalBatch = 16;
    const long long wgs = (long long)tz * ty * tx * kNominalBatch * (a.cout / NWG);
    const int nchunks = (a.ca + a.cb) / (2 * Tag::kG);
    const size_t patch_vox_all = (size_t)a.d * a.h * a.w;
    const size_t nvox_all = (size_t)a.n * patch_vox_all;
    if (a.partial && whole && !a.head_out && wgs * 2 <= resident_workgroups(2)) {
        int ks = (int)(resident_workgroups(2) / wgs);
        if (ks > 4) ks = 4;
        if (ks > nchunks) ks = nchunks;
        while (ks > 1 && (size_t)ks * patch_vox_all * a.cout * sizeof(float) > a.partial_patch_bytes) --ks;
        b.ksplit = ks;
    }
    dim3 grid((unsigned)blocks, a.cout / NWG, b.ksplit);
    if (POOL && b.ksplit > 1) {
        // a split layer's output exists only after the reduction: its max-pool stays a launch of
        // its own (tiny layers; the split is a function of the layer, so is this choice)
        b.pool_dst = nullptr;
        conv3x3x3_t14<Tag, TZ, TY, TX, WAVES_M, WAVES_N, MT, NT, MINW, PD, ZORD, DMA, false>
            <<<grid, WAVES_M * WAVES_N * 64, 0, stream>>>(b, tz, ty, tx);
    } else {
        conv3x3x3_t14<Tag, TZ, TY, TX, WAVES_M, WAVES_N, MT, NT, MINW, PD, ZORD, DMA, POOL>
            <<<grid, WAVES_M * WAVES_N * 64, 0, stream>>>(b, tz, ty, tx);
    }
    EXA_CHECK_HIP(hipGetLastError());
    if (b.ksplit > 1) {
        const size_t items = nvox_all * (a.cout / 4);
        splitk_reduce_kernel<Tag><<<(unsigned)((items + 255) / 256), 256, 0, stream>>>(
            a.partial, a.bias, a.dst, nvox_all, (size_t)a.d * a.h * a.w, a.cout, b.ksplit, a.slope);
        EXA_CHECK_HIP(hipGetLastError());
        if (POOL)
            return launch_maxpool2(Tag::kCode, a.dst, a.pool_dst, a.n, a.d, a.h, a.w, a.cout, stream);
    }
    return EXASPIM_OK;
}

template <typename Tag> struct ES_of { static constexpr int value = 16 / Tag::kG; };   // bytes per element

// t14 configuration with or without the fused max-pool
template <typename Tag, int TZ, int TY, int TX, int WAVES_M, int WAVES_N, int MT, int NT, int MINW>
static int launch_cfg_pool(const ConvArgs& a, hipStream_t stream) {
    // (16-bit types only: with float32 records the parked groups of the 64-cout shapes take
    // 139 KB of LDS and the CU would hold one workgroup; conv_can_fuse_pool says no for those)
    if constexpr (ES_of<Tag>::value == 2) {
        if (a.pool_dst)
            return launch_cfg<Tag, TZ, TY, TX, WAVES_M, WAVES_N, MT, NT, MINW, 3, false, false, true>(a, stream);
    }
    return launch_cfg<Tag, TZ, TY, TX, WAVES_M, WAVES_N, MT, NT, MINW>(a, stream);
}

// 32-cout slice on z-column tiles of TZ planes, with or without the fused head
template <typename Tag, int TZ, int D>
static int launch_zpipe_head(const ConvArgs& a, hipStream_t stream) {
    if (a.head_out && a.cout == 32) {
        switch (a.head_oc) {
            case 1: return launch_zpipe<Tag, TZ, 8, 16, 2, D, 1>(a, stream);
            case 2: return launch_zpipe<Tag, TZ, 8, 16, 2, D, 2>(a, stream);
            case 3: return launch_zpipe<Tag, TZ, 8, 16, 2, D, 3>(a, stream);
            case 4: return launch_zpipe<Tag, TZ, 8, 16, 2, D, 4>(a, stream);
        }
    }
    if (a.pool_dst) return launch_zpipe<Tag, TZ, 8, 16, 2, D, 0, true>(a, stream);
    return launch_zpipe<Tag, TZ, 8, 16, 2, D>(a, stream);
}

template <typename Tag, int TZ>
static int launch_zpipe_d(const ConvArgs& a, hipStream_t stream) {
#ifdef EXASPIM_TRACE
    if (g_variant % 10 == 3) return launch_zpipe_head<Tag, TZ, 3>(a, stream);
    if (g_variant % 10 == 5) return launch_zpipe_head<Tag, TZ, 5>(a, stream);
#endif
    return launch_zpipe_head<Tag, TZ, 4>(a, stream);
}

template <typename Tag>
static int launch_typed(const ConvArgs& a, hipStream_t stream) {
    // Widest x extent first: the tile shapes follow the 96/48/24/12/6 pyramid of
    // a 96^3 patch; any other size runs on the closest shape with masking.
    if (a.w >= 16 && a.w % 16 == 0) {
        // 32-cout slices: z-column tiles with the chunk's weights shared through LDS
#ifdef EXASPIM_TRACE
        if (g_variant >= 20) return launch_zpipe_d<Tag, 6>(a, stream);
#endif
        if (a.cout % 64 != 0) {
#ifdef EXASPIM_VARIANTS
            if constexpr (HasPaired<Tag>::value) {
                if (a.weights_paired && paired_enabled()) {
                    if (a.d % 6 == 0) return launch_zpair_head<Tag, 6>(a, stream);
                    return launch_zpair_head<Tag, 4>(a, stream);
                }
            }
#endif
#if EXASPIM_HEAD_TZ
            // the fused-head launch of the trimmed forward covers 80 planes: 16 tiles of 5 instead of 14 of 6
            // (4.8 % fewer planes; up4.3 459 -> 439 us inside 512^3 steps, same bits: a voxel's taps and
            // chunks accumulate in the same order whatever the tile)
            if (a.head_out && a.cout == 32 && a.ext[0] % 6 != 0 && a.ext[0] % EXASPIM_HEAD_TZ == 0) {
                switch (a.head_oc) {
                    case 1: return launch_zpipe<Tag, EXASPIM_HEAD_TZ, 8, 16, 2, 4, 1>(a, stream);
                    case 2: return launch_zpipe<Tag, EXASPIM_HEAD_TZ, 8, 16, 2, 4, 2>(a, stream);
                    case 3: return launch_zpipe<Tag, EXASPIM_HEAD_TZ, 8, 16, 2, 4, 3>(a, stream);
                    case 4: return launch_zpipe<Tag, EXASPIM_HEAD_TZ, 8, 16, 2, 4, 4>(a, stream);
                }
            }
#endif
            // 6-plane tiles when the depth divides (96, 48, 24): more dz reuse per LDS read
            if (a.d % 6 == 0) return launch_zpipe_d<Tag, 6>(a, stream);
            return launch_zpipe_d<Tag, 4>(a, stream);
        }
#ifdef EXASPIM_VARIANTS
        if constexpr (HasPaired<Tag>::value) {
            static int t16 = -1;
            if (t16 < 0) { const char* e = getenv("EXASPIM_T16"); t16 = e && e[0] == '1'; }
            const bool whole = a.ext[0] == a.d && a.ext[1] == a.h && a.ext[2] == a.w;
            if (t16 && !a.pool_dst && a.weights_k32 && whole && a.cout == 64 && a.ca % 32 == 0 && a.cb % 32 == 0) {
                const int tz = cdiv(a.d, 4), ty = cdiv(a.h, 8), tx = cdiv(a.w, 16);
                const int ntiles = tz * ty * tx * a.n;
                long long nwg = resident_workgroups(1) / 8 * 8;   // one persistent workgroup per CU
                if (nwg < 8) nwg = 8;
                if (nwg > ntiles) nwg = (ntiles + 7) / 8 * 8;
                conv3x3x3_t16<Tag, 2><<<(unsigned)nwg, 256, 0, stream>>>(a, tz, ty, tx);
                EXA_CHECK_HIP(hipGetLastError());
                return EXASPIM_OK;
            }
        }
#endif
#ifndef EXASPIM_T14_DMA
#define EXASPIM_T14_DMA 0
#endif
        if (a.pool_dst) return launch_cfg_pool<Tag, 4, 8, 16, 4, 1, 4, 2, 2>(a, stream);
        return launch_cfg<Tag, 4, 8, 16, 4, 1, 4, 2, 2, 3, false, EXASPIM_T14_DMA != 0>(a, stream);
    }
    if (a.w > 12) {
        if (a.cout % 64 == 0) return launch_cfg_pool<Tag, 4, 4, 24, 4, 1, 3, 2, 2>(a, stream);
        return launch_cfg_pool<Tag, 4, 4, 24, 4, 1, 3, 1, 2>(a, stream);
    }
    if (a.w > 6) {
        // 256 couts and more: 32-cout slices on two-wave workgroups fill the 256 CUs better
        // than 128-cout slices (144 tiles per batch of 16 at the 12^3 level)
        if (a.cout % 256 == 0) return launch_cfg_pool<Tag, 4, 4, 12, 2, 1, 3, 1, 2>(a, stream);
        if (a.cout % 128 == 0) return launch_cfg_pool<Tag, 4, 4, 12, 2, 2, 3, 2, 2>(a, stream);
        if (a.cout % 64 == 0) return launch_cfg_pool<Tag, 4, 4, 12, 2, 2, 3, 1, 2>(a, stream);
        return launch_cfg_pool<Tag, 4, 4, 12, 2, 1, 3, 1, 2>(a, stream);
    }
    // 6^3 level: the whole patch is one tile; 32-cout slices on four waves give the most
    // workgroups (16 patches x 8 slices for 256 couts)
    return launch_cfg<Tag, 6, 6, 6, 4, 1, 2, 1, 2>(a, stream);
}

bool conv_can_fuse_pool(int dtype, int cout, int d, int h, int w) {
    if (d % 2 != 0 || h % 2 != 0 || w % 2 != 0) return false;
    // the layers launch_typed sends to the z-column kernel (any dtype) ...
    if (cout % 64 != 0 && w >= 16 && w % 16 == 0) return true;
    // ... and, in the 16-bit modes, every other tile shape but the single-tile 6^3 one
    return dtype != EXASPIM_DT_F32 && w > 6;
}

bool conv_can_fuse_head(int cout, int w, int head_oc) {
    return cout == 32 && w >= 16 && w % 16 == 0 && head_oc >= 1 && head_oc <= 4;
}

// ext = 0 stands for the whole axis; the region must lie inside the patch
static int resolve_region(ConvArgs& a) {
    const int dims[3] = {a.d, a.h, a.w};
    for (int i = 0; i < 3; ++i) {
        if (a.ext[i] == 0 && a.org[i] == 0) a.ext[i] = dims[i];
        EXA_CHECK_ARG(a.org[i] >= 0 && a.ext[i] > 0 && a.org[i] + a.ext[i] <= dims[i],
                      "conv: region [%d, %d) outside axis %d of a %dx%dx%d patch", a.org[i],
                      a.org[i] + a.ext[i], i, a.d, a.h, a.w);
    }
    return EXASPIM_OK;
}

int conv_zcol_main_extent(int ext, int axis) {
    const int tile = axis == 1 ? 8 : 16;   // the z-column kernel's TY / TX
    const int rem = ext % tile;
    return (axis != 0 && ext > tile && rem >= 1 && rem <= 4) ? ext - rem : ext;
}

int launch_conv3x3x3(int dtype, const ConvArgs& a_in, hipStream_t stream) {
    ConvArgs a = a_in;
    const int kc = dtype == EXASPIM_DT_F32 ? 8 : 16;
    EXA_CHECK_ARG(a.ca % kc == 0 && a.cb % kc == 0 && a.cout % 32 == 0 && a.ca > 0,
                  "conv: channels (%d,%d)->%d not padded", a.ca, a.cb, a.cout);
    EXA_CHECK_ARG(a.n > 0 && a.d > 0 && a.h > 0 && a.w > 0, "conv: empty input");
    EXA_CHECK_ARG(a.slope >= 0.f && a.slope <= 1.f, "conv: LeakyReLU slope %g outside [0, 1]", a.slope);
    if (int rc = resolve_region(a)) return rc;
    const bool whole = a.ext[0] == a.d && a.ext[1] == a.h && a.ext[2] == a.w;
    EXA_CHECK_ARG(!a.pool_dst || (conv_can_fuse_pool(dtype, a.cout, a.d, a.h, a.w) && !a.head_out && whole),
                  "conv: fused max-pool needs an even, untrimmed patch (16-bit modes: wider than 6 voxels; "
                  "float32: a 32-cout-slice layer)");
    EXA_CHECK_ARG(!a.head_out || conv_can_fuse_head(a.cout, a.w, a.head_oc),
                  "conv: fused head needs cout 32, w %% 16 == 0, 1..4 outputs");
    {   // the staging loads address one patch of one source with 32-bit buffer offsets
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

template <typename Tag>
static int launch_thin_typed(const ConvArgs& a, hipStream_t stream) {
    // 2-voxel-thick tiles: thin along y (TZ x 2 x 16) or along x (TZ x 16 x 2), four waves. A wave owns
    // TZ / 4 groups of 32 voxels, so a weight fragment fetched through the L2 ring feeds that many
    // MFMAs and the z halo is shared by more planes: measured inside 512^3 steps on up4.0's two
    // remainders (82 planes, rocprofv3, us per launch) thin along y 54.2 / 42.3 - 44.5 / 36.0 for
    // TZ = 4 / 8 / 12 (three workgroups per CU for 12), thin along x 56.6 / 45.7 / 47.4 - 51.9 (its
    // 2-voxel rows keep 50 % LDS bank conflicts). The depth is picked per launch: padded planes x
    // the relative cost per plane from those measurements.
    const int ez = a.ext[0];
    auto planes = [&](int tz) { return (ez + tz - 1) / tz * tz; };
    if (a.ext[1] <= a.ext[2]) {
        const float c4 = planes(4) * 1.00f, c8 = planes(8) * 0.76f, c12 = planes(12) * 0.66f;
        if (c12 <= c8 && c12 <= c4) return launch_cfg<Tag, 12, 2, 16, 4, 1, 3, 1, 3, 3, true>(a, stream);
        if (c8 <= c4) return launch_cfg<Tag, 8, 2, 16, 4, 1, 2, 1, 4, 3, true>(a, stream);
        return launch_cfg<Tag, 4, 2, 16, 4, 1, 1, 1, 4, 3, true>(a, stream);
    }
    if (planes(8) * 0.78f <= planes(4) * 1.00f) return launch_cfg<Tag, 8, 16, 2, 4, 1, 2, 1, 4, 3, true>(a, stream);
    return launch_cfg<Tag, 4, 16, 2, 4, 1, 1, 1, 4, 3, true>(a, stream);
}

int launch_conv3x3x3_thin(int dtype, const ConvArgs& a_in, hipStream_t stream) {
    ConvArgs a = a_in;
    const int kc = dtype == EXASPIM_DT_F32 ? 8 : 16;
    EXA_CHECK_ARG(a.ca % kc == 0 && a.cb % kc == 0 && a.cout % 32 == 0 && a.ca > 0,
                  "conv: channels (%d,%d)->%d not padded", a.ca, a.cb, a.cout);
    EXA_CHECK_ARG(a.n > 0 && a.d > 0 && a.h > 0 && a.w > 0, "conv: empty input");
    EXA_CHECK_ARG(!a.pool_dst && !a.head_out, "conv: thin tiles have no fused pool or head");
    if (int rc = resolve_region(a)) return rc;
    a.partial = nullptr;   // no split-K on a partial region
    switch (dtype) {
        case EXASPIM_DT_F32: return launch_thin_typed<F32Tag>(a, stream);
        case EXASPIM_DT_BF16: return launch_thin_typed<BF16Tag>(a, stream);
        case EXASPIM_DT_F16: return launch_thin_typed<F16Tag>(a, stream);
    }
    set_error("conv: unknown dtype %d", dtype);
    return EXASPIM_E_INVALID;
}

}  // namespace exaspim
