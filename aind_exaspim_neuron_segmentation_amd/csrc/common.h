// Shared declarations of the exaspim_affinity HIP extension (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "../../include/exaspim_affinity.h"

namespace exaspim {

// ---- error reporting (thread-local message behind exaspim_last_error) ----
void set_error(const char* fmt, ...);
const char* get_error();

#define EXA_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::exaspim::set_error(__VA_ARGS__);   \
            return EXASPIM_E_INVALID;            \
        }                                        \
    } while (0)

#define EXA_CHECK_HIP(expr)                                                  \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) {                                              \
            ::exaspim::set_error("%s failed: %s (%s:%d)", #expr,             \
                                 hipGetErrorString(_e), __FILE__, __LINE__); \
            return EXASPIM_E_HIP;                                            \
        }                                                                    \
    } while (0)

// ---- device helper shared by the convolution epilogues ------------------------
// One voxel's 16-channel record of a chunk plane (32 B in a 16-bit type) leaves the accumulators
// of a 32 x 32 MFMA tile split over the two half-waves: lane r holds channels 8q .. 8q + 3 of its
// voxel, lane r + 32 channels 8q + 4 .. 8q + 7 (q = 0 .. 3). v_permlane32_swap exchanges the upper
// half of group q with the lower half of group q + 1, after which lane r holds the record's first
// 16 bytes (channels 0 .. 7 of the pair) and lane r + 32 the second 16 (channels 8 .. 15): one
// 16-byte store per lane and chunk plane, 32 whole records per instruction, no LDS round trip.
__device__ __forceinline__ uint4 record_half(uint2 lo_group, uint2 hi_group) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo_group.x, hi_group.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(lo_group.y, hi_group.y, false, false);
    return make_uint4(r0[0], r1[0], r0[1], r1[1]);
}


// 16-byte MUBUF store with a scalar offset (range-checked against the descriptor: an offset at or
// beyond its size is dropped by the hardware). gfx950 needs a wait state between such a store and a
// vector-ALU write of its data registers -- tools/store_hazard.hip: without one 0.5 % of the stored
// words are the NEW register contents (5 % with an immediate soffset, where one wait state is still
// not enough and two are). hipcc's hazard recognizer guards only the form without a register soffset,
// and with a single wait state, so the store carries its own s_nop 1. (Being inline assembly it is also
// invisible to hipcc's s_waitcnt bookkeeping: nothing ever waits on these stores but the end of the
// kernel -- which is what the epilogues want -- AND to its hazard recognizer: a descriptor or offset
// SGPR written by the vector ALU right before (v_readlane of a spilled SGPR, v_readfirstlane) needs
// five wait states before a memory instruction reads it, which hipcc inserts for its own instructions
// only; without the leading s_nop 4 the store of the pooled epilogue went out with a half-restored
// descriptor and faulted.)
typedef unsigned int exa_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_store16(const uint4& v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff,
                                            unsigned soff) {
    const exa_u32x4 d = {v.x, v.y, v.z, v.w};
    const unsigned ssoff = __builtin_amdgcn_readfirstlane(soff);   // (wave-uniform by contract)
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1"
                 :: "v"(d), "v"(voff), "s"(rsrc), "s"(ssoff) : "memory");
}

// The same store through the compiler's builtin, for code whose later waits should COUNT it (a load
// issued before it is then waited for with s_waitcnt vmcnt(#younger operations) instead of being made
// to wait for the store as well). The scheduling fences keep any vector-ALU write of the data registers
// from being placed between the store and its wait states.
__device__ __forceinline__ void buf_store16_counted(const uint4& v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff,
                                                    unsigned soff) {
    const exa_u32x4 d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)voff, (int)__builtin_amdgcn_readfirstlane(soff), 0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}

// ---- network plan ---------------------------------------------------------
// Channel counts are padded to multiples of 32 inside the workspace so that
// every MFMA convolution sees whole 32-wide output tiles and whole 32-byte
// input chunks; padded weights and biases are zero, so padded channels carry
// exact zeros through LeakyReLU, max-pool and interpolation.
constexpr int kChannelPad = 32;
constexpr int kNumMfmaConvs = 17;  // every 3x3x3 conv except inc.0 (Cin = 1)

// Paired-tap fragments of the z-column kernel on v_mfma_f32_16x16x32 (conv3d.hip,
// conv3x3x3_zpair): per 16-channel chunk 32 A fragments [16 couts x (2 taps x 16 channels)].
// Fragments 0..23 = pair * 6 + dz * 2 + t for the in-plane tap pairs (0,3) (1,4) (2,5) (6,7)
// (g = dy * 3 + dx; both taps in plane dz); 24..31 = 24 + kind * 2 + t for tap g = 8 chained
// over two consecutive input planes: kind 0 = [dz 0; dz 1], 1 = [dz 2; -], 2 = [-; dz 0],
// 3 = [dz 1; dz 2]. t = 16-cout half of the 32-cout slice.
constexpr int kPairedFrags = 32;
// (tap of the first, tap of the second 16-channel half; -1 = zero weights) of a fragment
inline void paired_frag_taps(int frag, int* tap0, int* tap1) {
    static const int pair_g[4][2] = {{0, 3}, {1, 4}, {2, 5}, {6, 7}};
    if (frag < 24) {
        const int pair = frag / 6, dz = (frag % 6) / 2;
        *tap0 = dz * 9 + pair_g[pair][0];
        *tap1 = dz * 9 + pair_g[pair][1];
    } else {
        static const int kind_dz[4][2] = {{0, 1}, {2, -1}, {-1, 0}, {1, 2}};
        const int kind = (frag - 24) / 2;
        *tap0 = kind_dz[kind][0] < 0 ? -1 : kind_dz[kind][0] * 9 + 8;
        *tap1 = kind_dz[kind][1] < 0 ? -1 : kind_dz[kind][1] * 9 + 8;
    }
}

inline int pad_channels(int c) { return (c + kChannelPad - 1) / kChannelPad * kChannelPad; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int dtype_size(int dtype) { return dtype == EXASPIM_DT_F32 ? 4 : 2; }

struct ConvLayer {
    int ca_real = 0, cb_real = 0;  // real input channels from source A / B
    int ca = 0, cb = 0;            // padded
    int cout_real = 0, cout = 0;   // real / padded output channels
    size_t w_off = 0;              // packed weights (compute dtype), bytes
    size_t w2_off = 0;             // the same weights in paired-tap fragment order (16-bit modes,
                                   // 32-cout-slice layers: conv3x3x3_zpair), 0 if not packed
    size_t w3_off = 0;             // the same weights as K = 32 fragments for conv3x3x3_t16 (16-bit modes,
                                   // the 64-cout-slice layers of pyramid level 1), 0 if not packed
    size_t b_off = 0;              // folded bias, float[cout], bytes
    size_t p_off = 0;              // offset of this conv's block in params
};

// ConvTranspose3d(k=2, s=2) of an Up block (UNet3D(trilinear=False))
struct ConvTLayer {
    int cin_real = 0, cin = 0;     // real / padded input channels
    int cout_real = 0, cout = 0;   // real / padded output channels
    size_t w_off = 0;              // packed weights [chunk][phase 8][tile][lane][16 B]
    size_t b_off = 0;              // bias, float[cout]
    size_t p_off = 0;              // offset of (weight, bias) in params
};

struct UNetPlan {
    int channels[5] = {0, 0, 0, 0, 0};
    int out_channels = 0;
    int dtype = 0;                  // EXASPIM_DT_* (without flags)
    bool convt = false;             // EXASPIM_UP_CONVT
    ConvTLayer up[4];               // up1.up .. up4.up (convt only)
    int c0 = 0, c0p = 0;            // inc.0 output channels real / padded
    size_t first_p_off = 0;         // params offset of inc.0
    size_t first_w_off = 0;         // float[27][c0p]
    size_t first_b_off = 0;         // float[c0p]
    ConvLayer conv[kNumMfmaConvs];  // inc.3, down1.0 ... up4.3
    size_t head_p_off = 0;
    size_t head_w_off = 0;          // float[out_channels][c0p]
    size_t head_b_off = 0;          // float[out_channels]
    size_t packed_bytes = 0;
    size_t n_params = 0;
};

// Builds the plan; returns false (and sets the error) on invalid arguments.
bool make_plan(const int32_t channels[5], int32_t out_channels, int32_t dtype,
               UNetPlan* plan);

int pack_weights(const UNetPlan& plan, const float* params, void* packed_host);

// ---- kernel launchers (conv3d.hip / layers.hip / prepost.hip) -------------
struct ConvArgs {
    const void* src_a;
    const void* src_b;
    int ca, cb;          // padded channel counts of the two sources (cb may be 0)
    const void* weights; // packed fragments
    const void* weights_paired = nullptr;   // paired-tap fragments (ConvLayer::w2_off), or null
    const void* weights_k32 = nullptr;      // K = 32 fragments (ConvLayer::w3_off), or null
    const float* bias;
    void* dst;
    int cout;            // padded
    int n, d, h, w;      // batch of patches and their spatial size at this level
    float slope;
    // Optional fused OutConv (unet3d.py:318) + sigmoid (inference.py:158): when
    // head_out is set (32-cout layers only) the conv's activations are not stored;
    // the 1x1x1 head runs on the accumulators and writes NCDHW float32.
    const float* head_w = nullptr;  // float[head_oc][32]
    const float* head_b = nullptr;  // float[head_oc]
    float* head_out = nullptr;      // float (n, head_oc, d, h, w)
    int head_oc = 0;
    int head_sigmoid = 0;
    // Region of output voxels the caller needs, [org, org + ext) per axis (z, y, x);
    // ext = 0 means the whole patch. predict() trims the outputs of every patch
    // (inference.py:161-162), so the last two convolutions only produce what survives:
    // tiles start at org, whole tiles outside the region are never launched and
    // stores are masked to it. Voxels outside the region are left untouched.
    int org[3] = {0, 0, 0};
    int ext[3] = {0, 0, 0};
    // Optional fused MaxPool3d(2) (unet3d.py:195) of this conv's output, written to
    // pool_dst as (n, cout, d/2, h/2, w/2) in the same layout (every tile shape but 6^3:
    // ask conv_can_fuse_pool first); saves re-reading the whole skip tensor.
    void* pool_dst = nullptr;
    // Optional scratch for split-K (t14 kernel): launches with too few workgroups to fill
    // the device cut the input-channel chunks into up to 4 ranges, every range writes its
    // float32 partial sums here and a second kernel adds them in a fixed order.
    float* partial = nullptr;
    size_t partial_patch_bytes = 0;   // scratch bytes per patch of the batch
    int ksplit = 1;      // set by the launcher
#ifdef EXASPIM_TRACE
    // tools/conv_trace.hip only: 16 x 64-bit cycle stamps per wave (never in the library build)
    unsigned long long* trace = nullptr;
#endif
};

int launch_conv3x3x3(int dtype, const ConvArgs& a, hipStream_t stream);
// Thin remainders of a region along y or x (at most 4 voxels thick) on 2-voxel-thick
// tiles: what is left when the z-column kernel's 8 x 16 tiles cover only the multiple-of-
// tile part of a trimmed region. 32-cout slices only.
int launch_conv3x3x3_thin(int dtype, const ConvArgs& a, hipStream_t stream);
// y / x extent of "ext" the z-column kernel should cover with whole tiles when the
// remainder goes to launch_conv3x3x3_thin: the largest multiple of the tile if the
// remainder is 1..4 voxels, else ext itself
int conv_zcol_main_extent(int ext, int axis);
bool conv_can_fuse_head(int cout, int w, int head_oc);
bool conv_can_fuse_pool(int dtype, int cout, int d, int h, int w);

// xpad: scratch for the zero-bordered copy of x, n * (d+2)(h+2)(wd+2) floats
// first_no_strips: keep the per-group kernel (the tests hold it to the row-strip kernel bit for bit)
int launch_conv_first(int dtype, const float* x, float* xpad, const float* w, const float* bias,
                      void* dst, int n, int d, int h, int wd, int c0p, float slope,
                      hipStream_t stream, bool first_no_strips = false);
// ConvTranspose3d(k=2, s=2): (n, d, h, w, cin) -> (n, 2d, 2h, 2w, cout), bias, no activation
int launch_convt2(int dtype, const void* src, const void* weights, const float* bias, void* dst,
                  int n, int d, int h, int w, int cin, int cout, hipStream_t stream);
int launch_maxpool2(int dtype, const void* src, void* dst, int n, int d, int h, int w,
                    int c, hipStream_t stream);  // d,h,w = INPUT size
// d,h,w = INPUT size; output voxels within "margin" of a face are not computed
int launch_upsample2(int dtype, const void* src, void* dst, int n, int d, int h, int w,
                     int c, int margin, hipStream_t stream, bool plain_kernel = false, bool per_thread = false);
// *out = max(*out, largest |value| in the tensor) as float bits (out zeroed by the caller)
int launch_absmax(int dtype, const void* src, size_t bytes, float* out, hipStream_t stream);
int launch_head(int dtype, const void* src, const float* w, const float* bias,
                float* out, int n, int d, int h, int wd, int c0p, int out_channels,
                int apply_sigmoid, hipStream_t stream);

}  // namespace exaspim
