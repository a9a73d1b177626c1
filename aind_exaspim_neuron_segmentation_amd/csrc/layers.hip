// HBM-bound layers of the U-Net on channels-last activations:
//   conv_first : inc.0, Conv3d(1 -> C0, k3, p1) + folded BN + LeakyReLU
//                (machine_learning/unet3d.py:64,143-145) from the float32 patch
//   maxpool2   : nn.MaxPool3d(2)                          (unet3d.py:195)
//   upsample2  : nn.Upsample(x2, trilinear, align_corners=True) (unet3d.py:248)
//   head       : OutConv 1x1x1 (+ sigmoid of inference.py:158) -> NCDHW float32
// Every thread moves 16-byte channel groups; consecutive lanes touch
// consecutive addresses.

#include "common.h"

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
            const _Float16 lo = (_Float16)f[2 * i], hi = (_Float16)f[2 * i + 1];
            w[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) |
                   ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// ---- inc.0 ------------------------------------------------------------------
// One thread = one voxel x 32 output channels (blockIdx.y selects the group of
// 32), so the 27 x 32 weights are wave-uniform scalar loads and every input
// voxel is read once per tap; loads are clamped + selected instead of branched.
template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(
    const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ bias, void* __restrict__ dst, int n, int d, int h, int wd,
    int c0p, float slope) {
    const size_t nvox = (size_t)n * d * h * wd;
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int co0 = blockIdx.y * 32;
    const int xx = (int)(v % wd);
    size_t t = v / wd;
    const int yy = (int)(t % h); t /= h;
    const int zz = (int)(t % d);
    const size_t nb = t / d;
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = bias[co0 + j];
    const float* xp = x + nb * (size_t)d * h * wd;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
        const int z = zz + kz - 1;
        const bool zok = (unsigned)z < (unsigned)d;
        const int zc = min(max(z, 0), d - 1);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = yy + ky - 1;
            const bool yok = zok && (unsigned)y < (unsigned)h;
            const int yc = min(max(y, 0), h - 1);
            const float* row = xp + ((size_t)zc * h + yc) * wd;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xq = xx + kx - 1;
                const bool ok = yok && (unsigned)xq < (unsigned)wd;
                const float xv = ok ? row[min(max(xq, 0), wd - 1)] : 0.f;
                const float* wt = w + (size_t)((kz * 3 + ky) * 3 + kx) * c0p + co0;
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = fmaf(xv, wt[j], acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = acc[j] > 0.f ? acc[j] : acc[j] * slope;
    constexpr int G = T::kG;
    uint4* out = reinterpret_cast<uint4*>(static_cast<char*>(dst) +
                                          (v * c0p + co0) * (16 / G));
#pragma unroll
    for (int g = 0; g < 32 / G; ++g) out[g] = T::pack(acc + g * G);
}

// ---- max-pool 2x2x2 ----------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const uint4* __restrict__ src,
                                                       uint4* __restrict__ dst, int n, int d,
                                                       int h, int w, int cg) {
    // d,h,w: input size; cg: 16-byte groups per voxel
    const int od = d / 2, oh = h / 2, ow = w / 2;
    const size_t total = (size_t)n * od * oh * ow * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        size_t t = i / cg;
        const int x = (int)(t % ow); t /= ow;
        const int y = (int)(t % oh); t /= oh;
        const int z = (int)(t % od);
        const size_t nb = t / od;
        float m[T::kG];
#pragma unroll
        for (int j = 0; j < T::kG; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t vox = ((nb * d + 2 * z + dz) * h + 2 * y + dy) * w + 2 * x + dx;
                    float f[T::kG];
                    T::unpack(src[vox * cg + g], f);
#pragma unroll
                    for (int j = 0; j < T::kG; ++j) m[j] = fmaxf(m[j], f[j]);
                }
        dst[i] = T::pack(m);
    }
}

// ---- trilinear x2, align_corners=True ---------------------------------------
// torch (ATen UpSample.h): scale = (in - 1) / (out - 1) in float; src = scale *
// dst_index; i0 = floor(src) clamped; lambda = src - i0 clamped to [0, 1];
// i1 = min(i0 + 1, in - 1).
__device__ __forceinline__ void lerp_coord(int o, int in, int out, int& i0, int& i1, float& l1) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float s = scale * (float)o;
    i0 = min((int)floorf(s), in - 1);
    i1 = min(i0 + 1, in - 1);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(const uint4* __restrict__ src,
                                                        uint4* __restrict__ dst, int n, int d,
                                                        int h, int w, int cg) {
    const int od = d * 2, oh = h * 2, ow = w * 2;
    const size_t total = (size_t)n * od * oh * ow * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        size_t t = i / cg;
        const int x = (int)(t % ow); t /= ow;
        const int y = (int)(t % oh); t /= oh;
        const int z = (int)(t % od);
        const size_t nb = t / od;
        int z0, z1, y0, y1, x0, x1;
        float lz, ly, lx;
        lerp_coord(z, d, od, z0, z1, lz);
        lerp_coord(y, h, oh, y0, y1, ly);
        lerp_coord(x, w, ow, x0, x1, lx);
        float acc[T::kG];
#pragma unroll
        for (int j = 0; j < T::kG; ++j) acc[j] = 0.f;
        const int zs[2] = {z0, z1}, ys[2] = {y0, y1}, xs[2] = {x0, x1};
        const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const size_t vox = ((nb * d + zs[a]) * h + ys[b]) * w + xs[c];
                    float f[T::kG];
                    T::unpack(src[vox * cg + g], f);
                    const float wgt = wz[a] * wy[b] * wx[c];
#pragma unroll
                    for (int j = 0; j < T::kG; ++j) acc[j] = fmaf(wgt, f[j], acc[j]);
                }
        dst[i] = T::pack(acc);
    }
}

// ---- head: 1x1x1 conv (+ sigmoid), channels-last -> NCDHW float32 -----------
template <typename T, int OC>
__global__ __launch_bounds__(256) void head_kernel(const uint4* __restrict__ src,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ out, size_t nvox_per_patch,
                                                   int n, int c0p, int apply_sigmoid) {
    const size_t total = nvox_per_patch * n;
    const int cg = c0p / T::kG;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total;
         v += (size_t)gridDim.x * blockDim.x) {
        float acc[OC];
#pragma unroll
        for (int o = 0; o < OC; ++o) acc[o] = bias[o];
        for (int g = 0; g < cg; ++g) {
            float f[T::kG];
            T::unpack(src[v * cg + g], f);
#pragma unroll
            for (int o = 0; o < OC; ++o)
#pragma unroll
                for (int j = 0; j < T::kG; ++j)
                    acc[o] = fmaf(f[j], w[o * c0p + g * T::kG + j], acc[o]);
        }
        const size_t nb = v / nvox_per_patch, sp = v % nvox_per_patch;
#pragma unroll
        for (int o = 0; o < OC; ++o) {
            float r = acc[o];
            if (apply_sigmoid) r = 1.f / (1.f + expf(-r));
            out[(nb * OC + o) * nvox_per_patch + sp] = r;
        }
    }
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

int launch_conv_first(int dtype, const float* x, const float* w, const float* bias,
                      void* dst, int n, int d, int h, int wd, int c0p, float slope,
                      hipStream_t stream) {
    const size_t nvox = (size_t)n * d * h * wd;
    const size_t blocks = (nvox + 255) / 256;
    EXA_CHECK_ARG(blocks > 0 && blocks < 0x7fffffffULL && c0p % 32 == 0, "conv_first: bad size");
    dim3 grid((unsigned)blocks, c0p / 32);
    DISPATCH_T(dtype, (conv_first_kernel<T><<<grid, 256, 0, stream>>>(x, w, bias, dst, n, d, h, wd, c0p, slope)));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

int launch_maxpool2(int dtype, const void* src, void* dst, int n, int d, int h, int w,
                    int c, hipStream_t stream) {
    EXA_CHECK_ARG(d % 2 == 0 && h % 2 == 0 && w % 2 == 0, "maxpool: odd size %dx%dx%d", d, h, w);
    const int cg = c * dtype_size(dtype) / 16;
    const size_t total = (size_t)n * (d / 2) * (h / 2) * (w / 2) * cg;
    DISPATCH_T(dtype, (maxpool2_kernel<T><<<stream_grid(total), 256, 0, stream>>>(
                          static_cast<const uint4*>(src), static_cast<uint4*>(dst), n, d, h, w, cg)));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

int launch_upsample2(int dtype, const void* src, void* dst, int n, int d, int h, int w,
                     int c, hipStream_t stream) {
    const int cg = c * dtype_size(dtype) / 16;
    const size_t total = (size_t)n * d * h * w * 8 * cg;
    DISPATCH_T(dtype, (upsample2_kernel<T><<<stream_grid(total), 256, 0, stream>>>(
                          static_cast<const uint4*>(src), static_cast<uint4*>(dst), n, d, h, w, cg)));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

template <typename T>
static int launch_head_t(const void* src, const float* w, const float* bias, float* out,
                         size_t nvox, int n, int c0p, int oc, int sig, hipStream_t stream) {
    const unsigned grid = stream_grid(nvox * n);
    const uint4* s = static_cast<const uint4*>(src);
    switch (oc) {
        case 1: head_kernel<T, 1><<<grid, 256, 0, stream>>>(s, w, bias, out, nvox, n, c0p, sig); break;
        case 2: head_kernel<T, 2><<<grid, 256, 0, stream>>>(s, w, bias, out, nvox, n, c0p, sig); break;
        case 3: head_kernel<T, 3><<<grid, 256, 0, stream>>>(s, w, bias, out, nvox, n, c0p, sig); break;
        case 4: head_kernel<T, 4><<<grid, 256, 0, stream>>>(s, w, bias, out, nvox, n, c0p, sig); break;
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
