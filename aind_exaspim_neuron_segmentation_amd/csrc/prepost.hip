// Device-side pre/post-processing of predict() (inference.py:79-126), all
// HBM-bound byte/float streaming:
//   histogram         -> exact order statistics for np.percentile (img_util.py:526)
//   gather_patches    -> np.minimum + normalize + get_patch_slices + reflect
//                        add_padding + float32 cast (inference.py:79-80,188-192)
//   stitch_accumulate -> trimmed overlap-add (inference.py:99-116)
//   stitch_finalize   -> divide by the patch count (inference.py:120-125)
//   synth_volume_u16  -> benchmark / test input (utils/synthetic.py)

#include "common.h"

namespace exaspim {

// ---------------------------------------------------------------- synthetic --
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_kernel(uint16_t* __restrict__ vol, exaspim_block blk,
                                                    unsigned long long seed) {
    const size_t total = (size_t)blk.dims[0] * blk.dims[1] * blk.dims[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % blk.dims[2]);
        size_t t = i / blk.dims[2];
        const int y = (int)(t % blk.dims[1]);
        const int z = (int)(t / blk.dims[1]);
        const unsigned long long lin =
            ((unsigned long long)(z + blk.origin[0]) * blk.global[1] + (y + blk.origin[1])) *
                blk.global[2] + (x + blk.origin[2]);
        vol[i] = (uint16_t)(splitmix64(lin + seed) % 2000ULL);
    }
}

// ---------------------------------------------------------------- histogram --
__device__ __forceinline__ unsigned f32_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving
}

// bin of one voxel, or -1 if it does not take part in this pass
template <int VOX>
__device__ __forceinline__ int bin_of(const void* vol, size_t i, double clip, int has_clip,
                                      int pass, unsigned prefix);
template <>
__device__ __forceinline__ int bin_of<EXASPIM_VOX_U8>(const void* vol, size_t i, double clip,
                                                      int has_clip, int, unsigned) {
    int v = static_cast<const uint8_t*>(vol)[i];
    if (has_clip && (double)v > clip) v = (int)ceil(clip);   // a fractional clip gets the bin above its floor
    return v;
}
template <>
__device__ __forceinline__ int bin_of<EXASPIM_VOX_U16>(const void* vol, size_t i, double clip,
                                                       int has_clip, int, unsigned) {
    int v = static_cast<const uint16_t*>(vol)[i];
    if (has_clip && (double)v > clip) v = (int)ceil(clip);   // a fractional clip gets the bin above its floor
    return v;
}
template <>
__device__ __forceinline__ int bin_of<EXASPIM_VOX_I16>(const void* vol, size_t i, double clip,
                                                       int has_clip, int, unsigned) {
    int v = static_cast<const int16_t*>(vol)[i];
    if (has_clip && (double)v > clip) v = (int)ceil(clip);   // a fractional clip gets the bin above its floor
    return v + 32768;
}
template <>
__device__ __forceinline__ int bin_of<EXASPIM_VOX_F32>(const void* vol, size_t i, double clip,
                                                       int has_clip, int pass, unsigned prefix) {
    float v = static_cast<const float*>(vol)[i];
    if (has_clip && (double)v > clip) {
        // a clip float32 cannot hold (float64 images) gets the key of the float32 just above
        // it, which no voxel can have after clipping: that bin stands for the clip itself
        float cf = (float)clip;
        if ((double)cf < clip) cf = __uint_as_float(__float_as_uint(cf) + (cf >= 0.f ? 1u : -1u));
        v = cf;
    }
    const unsigned k = f32_key(v);
    if (pass == 0) return (int)(k >> 16);
    return (k >> 16) == prefix ? (int)(k & 0xffffu) : -1;
}

// float64 voxels (images float32 cannot carry exactly: float64, or 32/64-bit integers converted on
// the host): order-preserving 64-bit key, binned 16 bits per pass -- pass p bins key bits
// [48 - 16p, 64 - 16p) of the voxels whose bits above that field equal "prefix".
__device__ __forceinline__ unsigned long long f64_key(double f) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(f);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ __forceinline__ int bin_of_f64(const void* vol, size_t i, double clip, int has_clip, int pass,
                                          unsigned long long prefix) {
    double v = static_cast<const double*>(vol)[i];
    if (has_clip && v > clip) v = clip;      // np.minimum: exact, the clip is a float64 itself
    const unsigned long long k = f64_key(v);
    if (pass > 0 && (k >> (64 - 16 * pass)) != prefix) return -1;
    return (int)((k >> (48 - 16 * pass)) & 0xffffULL);
}

constexpr int kLdsBins = 16384;  // privatised window of the 65536 bins

template <int VOX>
__global__ __launch_bounds__(256) void histogram_kernel(const void* __restrict__ vol, size_t n,
                                                        double clip, int has_clip, int pass,
                                                        unsigned prefix, int window_lo,
                                                        unsigned long long* __restrict__ hist) {
    __shared__ unsigned local[kLdsBins];
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x) local[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = bin_of<VOX>(vol, i, clip, has_clip, pass, prefix);
        if (b < 0) continue;
        const int lb = b - window_lo;
        if ((unsigned)lb < (unsigned)kLdsBins)
            atomicAdd(&local[lb], 1u);
        else
            atomicAdd(&hist[b], 1ULL);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x) {
        const unsigned c = local[i];
        if (c) atomicAdd(&hist[window_lo + i], (unsigned long long)c);
    }
}

// 8- and 16-bit integer voxels, pass 0 (the whole job for these dtypes): sixteen bytes per load, the
// clip as an integer threshold (v > clip <=> v > floor(clip) for an integer v) instead of a double
// compare per voxel, and the clip's own bin -- where every saturated voxel lands, half of the
// synthetic volume -- counted with one ballot per wave instead of 64 serialised LDS atomics.
// `thr`: largest value that is kept (INT_MAX without a clip); `cbin`: bin of a clipped voxel, before `bias`.
template <typename V>
__global__ __launch_bounds__(256) void histogram_int_kernel(const V* __restrict__ vol, size_t n, int thr, int cbin,
                                                            int bias, int window_lo,
                                                            unsigned long long* __restrict__ hist) {
    constexpr int PER = 16 / (int)sizeof(V);
    __shared__ unsigned local[kLdsBins];
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x) local[i] = 0;
    __syncthreads();
    unsigned clipped = 0;                         // this lane's share of the wave's clipped voxels
    auto count = [&](int v, bool live) {
        const bool sat = live && v > thr;
        const unsigned long long m = __ballot(sat);
        if ((threadIdx.x & 63) == 0) clipped += (unsigned)__popcll(m);
        if (live && !sat) {
            const int b = v + bias, lb = b - window_lo;
            if ((unsigned)lb < (unsigned)kLdsBins)
                atomicAdd(&local[lb], 1u);
            else
                atomicAdd(&hist[b], 1ULL);
        }
    };
    const size_t nvec = n / PER;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // (every lane of a wave runs the same number of iterations: the ballots see whole waves)
    const size_t rounds = (nvec + stride - 1) / stride;
    for (size_t r = 0; r < rounds; ++r) {
        const size_t i = r * stride + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = i < nvec;
        const uint4 q = live ? reinterpret_cast<const uint4*>(vol)[i] : make_uint4(0, 0, 0, 0);
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const unsigned word = w[k * (int)sizeof(V) / 4];
            const unsigned raw = (word >> (8 * ((k * (int)sizeof(V)) % 4))) & (sizeof(V) == 1 ? 0xffu : 0xffffu);
            count((int)(V)raw, live);
        }
    }
    // the last n % PER voxels
    {
        const size_t t = nvec * PER + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = blockIdx.x == 0 && t < n;
        count(live ? (int)vol[t] : 0, live);
    }
    if ((threadIdx.x & 63) == 0 && clipped) {
        const int b = cbin + bias, lb = b - window_lo;
        if ((unsigned)lb < (unsigned)kLdsBins)
            atomicAdd(&local[lb], clipped);
        else
            atomicAdd(&hist[b], (unsigned long long)clipped);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x) {
        const unsigned c = local[i];
        if (c) atomicAdd(&hist[window_lo + i], (unsigned long long)c);
    }
}

__global__ __launch_bounds__(256) void histogram_f64_kernel(const void* __restrict__ vol, size_t n, double clip,
                                                            int has_clip, int pass, unsigned long long prefix,
                                                            int window_lo, unsigned long long* __restrict__ hist) {
    __shared__ unsigned local[kLdsBins];
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x) local[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = bin_of_f64(vol, i, clip, has_clip, pass, prefix);
        if (b < 0) continue;
        const int lb = b - window_lo;
        if ((unsigned)lb < (unsigned)kLdsBins)
            atomicAdd(&local[lb], 1u);
        else
            atomicAdd(&hist[b], 1ULL);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x) {
        const unsigned c = local[i];
        if (c) atomicAdd(&hist[window_lo + i], (unsigned long long)c);
    }
}

// ------------------------------------------------------------------- gather --
__device__ __forceinline__ int reflect_index(int j, int n) {
    // numpy 'reflect' on the high side of a length-n axis (period 2(n-1))
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    const int m = j % p;
    return m < n ? m : p - m;
}

template <int VOX>
__device__ __forceinline__ double load_voxel(const void* vol, size_t i);
template <> __device__ __forceinline__ double load_voxel<EXASPIM_VOX_U8>(const void* v, size_t i) { return (double)static_cast<const uint8_t*>(v)[i]; }
template <> __device__ __forceinline__ double load_voxel<EXASPIM_VOX_U16>(const void* v, size_t i) { return (double)static_cast<const uint16_t*>(v)[i]; }
template <> __device__ __forceinline__ double load_voxel<EXASPIM_VOX_I16>(const void* v, size_t i) { return (double)static_cast<const int16_t*>(v)[i]; }
template <> __device__ __forceinline__ double load_voxel<EXASPIM_VOX_F32>(const void* v, size_t i) { return (double)static_cast<const float*>(v)[i]; }
template <> __device__ __forceinline__ double load_voxel<EXASPIM_VOX_F64>(const void* v, size_t i) { return static_cast<const double*>(v)[i]; }

// How a gathered value leaves the kernel (LAYOUT = EXASPIM_IN_*): the float32 patch the ABI's
// forward takes, or the first convolution's own operand layout -- a (pz + 2, py + 2, px + 2)
// copy with a zero border, float32 or every voxel already split into two 16-bit parts,
// hi | lo << 16 (what pad_input_kernel / pad_split_kernel of layers.hip make of the float32
// patch, bit for bit) -- so that the padding pass and a 4-byte-per-voxel round trip disappear.
template <int LAYOUT>
__device__ __forceinline__ void emit_input(float* out, size_t i, float r) {
    if (LAYOUT == EXASPIM_IN_F32 || LAYOUT == EXASPIM_IN_PADDED_F32) {
        out[i] = r;
    } else if (LAYOUT == EXASPIM_IN_PADDED_SPLIT_F16) {
        const _Float16 hi = (_Float16)r;
        const _Float16 lo = (_Float16)(r - (float)hi);
        reinterpret_cast<unsigned*>(out)[i] = (unsigned)__builtin_bit_cast(unsigned short, hi) |
                                              ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    } else {
        const unsigned short hi = __builtin_bit_cast(unsigned short, (__bf16)r);
        const unsigned short lo = __builtin_bit_cast(unsigned short, (__bf16)(r - __uint_as_float((unsigned)hi << 16)));
        reinterpret_cast<unsigned*>(out)[i] = (unsigned)hi | ((unsigned)lo << 16);
    }
}

// Grid: x = voxels of one (padded) patch row, y = row, z = patch * (padded) depth.
template <int VOX, int LAYOUT>
__global__ __launch_bounds__(128) void gather_kernel(const void* __restrict__ vol,
                                                     exaspim_block blk,
                                                     const int* __restrict__ starts, int pz,
                                                     int py, int px, double clip, int has_clip,
                                                     double mn, double denom,
                                                     float* __restrict__ out) {
    constexpr int B = LAYOUT == EXASPIM_IN_F32 ? 0 : 1;   // border
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    if (ox >= px + 2 * B) return;
    const int oy = blockIdx.y;
    const int p = blockIdx.z / (pz + 2 * B), oz = blockIdx.z - p * (pz + 2 * B);
    const size_t oidx = (((size_t)p * (pz + 2 * B) + oz) * (py + 2 * B) + oy) * (px + 2 * B) + ox;
    const int x = ox - B, y = oy - B, z = oz - B;
    if (B && ((unsigned)x >= (unsigned)px || (unsigned)y >= (unsigned)py || (unsigned)z >= (unsigned)pz)) {
        out[oidx] = 0.f;    // (all-zero bits in the split layouts as well)
        return;
    }
    const int sz = starts[3 * p], sy = starts[3 * p + 1], sx = starts[3 * p + 2];
    // in-volume extent of the patch (img_util.py:424-428), then reflect
    const int nz = min(sz + pz, blk.global[0]) - sz;
    const int ny = min(sy + py, blk.global[1]) - sy;
    const int nx = min(sx + px, blk.global[2]) - sx;
    const int lz = sz + reflect_index(z, nz) - blk.origin[0];
    const int ly = sy + reflect_index(y, ny) - blk.origin[1];
    const int lx = sx + reflect_index(x, nx) - blk.origin[2];
    float r = 0.f;
    if ((unsigned)lz < (unsigned)blk.dims[0] && (unsigned)ly < (unsigned)blk.dims[1] &&
        (unsigned)lx < (unsigned)blk.dims[2]) {
        double v = load_voxel<VOX>(vol, ((size_t)lz * blk.dims[1] + ly) * blk.dims[2] + lx);
        if (has_clip) v = fmin(v, clip);
        double q = (v - mn) / denom;         // float64, like numpy (img_util.py:527)
        q = fmin(fmax(q, 0.0), 1.0);         // np.clip(img, 0, 1)
        r = (float)q;                        // cast on assignment (inference.py:191)
    }
    emit_input<LAYOUT>(out, oidx, r);
}

// Unsigned integer volumes with a brightness clip take at most clip + 1 distinct values
// after np.minimum (inference.py:79), so the float64 normalisation is evaluated once per
// value into an LDS table (the same expression, entry for entry) and a voxel costs one
// lookup instead of a float64 division. One block = one (patch, z) plane.
template <typename V, int LAYOUT>
__global__ __launch_bounds__(256) void gather_lut_kernel(const V* __restrict__ vol, exaspim_block blk,
                                                         const int* __restrict__ starts, int pz,
                                                         int py, int px, int clip, double mn,
                                                         double denom, float* __restrict__ out) {
    extern __shared__ float lut[];
    for (int v = threadIdx.x; v <= clip; v += blockDim.x) {
        double q = ((double)v - mn) / denom;   // float64, like numpy (img_util.py:527)
        q = fmin(fmax(q, 0.0), 1.0);           // np.clip(img, 0, 1)
        lut[v] = (float)q;                     // cast on assignment (inference.py:191)
    }
    __syncthreads();
    constexpr int B = LAYOUT == EXASPIM_IN_F32 ? 0 : 1;   // border: one block per PADDED plane
    const int p = blockIdx.x / (pz + 2 * B), z = blockIdx.x - p * (pz + 2 * B) - B;
    const int opx = px + 2 * B, oplane_vox = (py + 2 * B) * opx;
    float* const oplane = out + (size_t)blockIdx.x * oplane_vox;
    if (B && (unsigned)z >= (unsigned)pz) {
        for (int i = threadIdx.x; i < oplane_vox; i += blockDim.x) oplane[i] = 0.f;
        return;
    }
    const int sz = starts[3 * p], sy = starts[3 * p + 1], sx = starts[3 * p + 2];
    const int nz = min(sz + pz, blk.global[0]) - sz;
    const int ny = min(sy + py, blk.global[1]) - sy;
    const int nx = min(sx + px, blk.global[2]) - sx;
    const int lz = sz + reflect_index(z, nz) - blk.origin[0];
    const bool zok = (unsigned)lz < (unsigned)blk.dims[0];
    // Rows: source row of every output row of the plane, once per block (-1: border row or outside the
    // block). Columns: a thread keeps its x, so its source column is computed once. What is left per voxel
    // is a 2-byte load, a table lookup and a store -- the integer divisions and modulos of a flat index
    // (two of each per voxel) were what this kernel spent its time on: 150 instructions per voxel, 52 us
    // per batch of 16, 38 us now. Of those the table costs 2 us and the 2-byte loads 19 us (ablations):
    // a patch row is 192 bytes of a 2 KiB volume row.
    int* const rowmap = reinterpret_cast<int*>(lut + clip + 1);
    const int opy = py + 2 * B;
    for (int oy = threadIdx.x; oy < opy; oy += blockDim.x) {
        const int y = oy - B;
        int ly = -1;
        if (!(B && (unsigned)y >= (unsigned)py)) {
            ly = sy + reflect_index(y, ny) - blk.origin[1];
            if ((unsigned)ly >= (unsigned)blk.dims[1]) ly = -2;     // inside the patch, outside the block: 0.f
        }
        rowmap[oy] = ly;
    }
    __syncthreads();
    const V* const zplane = vol + (size_t)(zok ? lz : 0) * blk.dims[1] * blk.dims[2];
    const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;      // 128 columns x 2 rows per round
    constexpr int ILP = 8;
    for (int xb = 0; xb < opx; xb += 128) {
        const int ox = xb + tx, x = ox - B;
        if (ox >= opx) continue;
        const bool xreal = !(B && (unsigned)x >= (unsigned)px);
        const int lx = sx + reflect_index(xreal ? x : 0, nx) - blk.origin[2];
        const bool xin = xreal && zok && (unsigned)lx < (unsigned)blk.dims[2];
        for (int oy0 = ty; oy0 < opy; oy0 += 2 * ILP) {
            int v[ILP], row[ILP];
#pragma unroll
            for (int k = 0; k < ILP; ++k) {      // the loads of eight rows before the first lookup
                const int oy = oy0 + 2 * k;
                row[k] = oy < opy ? rowmap[oy] : -3;
                v[k] = xin && row[k] >= 0 ? (int)zplane[(size_t)row[k] * blk.dims[2] + lx] : -1;
            }
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const int oy = oy0 + 2 * k;
                if (row[k] == -3) break;
                const size_t i = (size_t)oy * opx + ox;
                if (row[k] == -1 || !xreal)
                    oplane[i] = 0.f;    // zero border (all-zero bits in the split layouts as well)
                else
                    emit_input<LAYOUT>(oplane, i, v[k] < 0 ? 0.f : lut[v[k] < clip ? v[k] : clip]);
            }
        }
    }
}

// ------------------------------------------------------------------- stitch --
struct Coverage {
    int lo[3], hi[3];
};

__device__ __forceinline__ bool covered(const int* s, const exaspim_window& w, const int g[3],
                                        int z, int y, int x) {
    const int oz = w.patch[0] - 2 * w.trim, oy = w.patch[1] - 2 * w.trim,
              ox = w.patch[2] - 2 * w.trim;
    const int z0 = s[0] + w.trim, y0 = s[1] + w.trim, x0 = s[2] + w.trim;
    return z >= z0 && z < min(z0 + oz, g[0]) && y >= y0 && y < min(y0 + oy, g[1]) && x >= x0 &&
           x < min(x0 + ox, g[2]);
}

// Grid: x = blocks of voxels of one trimmed output plane, y = patch * trimmed depth.
template <int C>
__global__ __launch_bounds__(256) void stitch_kernel(const float* __restrict__ pred,
                                                     const int* __restrict__ starts, int n,
                                                     exaspim_window win,
                                                     float* __restrict__ accum,
                                                     exaspim_block blk) {
    const int oz = win.patch[0] - 2 * win.trim, oy = win.patch[1] - 2 * win.trim,
              ox = win.patch[2] - 2 * win.trim;
    const size_t pvox = (size_t)win.patch[0] * win.patch[1] * win.patch[2];
    const size_t avox = (size_t)blk.dims[0] * blk.dims[1] * blk.dims[2];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // voxel inside the trimmed plane
    const int y = i / ox, x = i - y * ox;
    const int p = blockIdx.y / oz, z = blockIdx.y - p * oz;
    const int* sp = starts + 3 * p;

    // Patches of the batch whose trimmed box meets patch p's (one ballot per
    // block): the per-voxel coverage tests below then run for the 1-7 neighbours
    // instead of for the whole batch. Batches of more than 64 test everything.
    __shared__ unsigned long long near_mask;
    const bool wide = n > 64;
    if (!wide) {
        if (threadIdx.x < 64) {
            const int j = threadIdx.x;
            bool hit = false;
            if (j < n) {
                const int* sj = starts + 3 * j;
                const int dz = sj[0] - sp[0], dy = sj[1] - sp[1], dx = sj[2] - sp[2];
                hit = dz > -oz && dz < oz && dy > -oy && dy < oy && dx > -ox && dx < ox;
            }
            const unsigned long long m = __ballot(hit);
            if (threadIdx.x == 0) near_mask = m;
        }
        __syncthreads();
    }
    if (y >= oy) return;
    const int gz = sp[0] + win.trim + z, gy = sp[1] + win.trim + y, gx = sp[2] + win.trim + x;
    if (gz >= blk.global[0] || gy >= blk.global[1] || gx >= blk.global[2]) return;
    const int lz = gz - blk.origin[0], ly = gy - blk.origin[1], lx = gx - blk.origin[2];
    if ((unsigned)lz >= (unsigned)blk.dims[0] || (unsigned)ly >= (unsigned)blk.dims[1] ||
        (unsigned)lx >= (unsigned)blk.dims[2])
        return;
    const size_t a = ((size_t)lz * blk.dims[1] + ly) * blk.dims[2] + lx;
    // The first patch of the batch that covers this voxel owns it and adds every
    // covering patch in batch order (= the reference's loop order).
    if (wide) {
        for (int j = 0; j < p; ++j)
            if (covered(starts + 3 * j, win, blk.global, gz, gy, gx)) return;
        float s[C];
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] = accum[c * avox + a];
        for (int j = p; j < n; ++j) {
            const int* sj = starts + 3 * j;
            if (j != p && !covered(sj, win, blk.global, gz, gy, gx)) continue;
            const size_t o = ((size_t)(gz - sj[0]) * win.patch[1] + (gy - sj[1])) * win.patch[2] + (gx - sj[2]);
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] += pred[((size_t)j * C + c) * pvox + o];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) accum[c * avox + a] = s[c];
        return;
    }
    const unsigned long long near = near_mask;
    unsigned long long before = near & ((1ULL << p) - 1ULL);
    while (before) {
        const int j = __ffsll((long long)before) - 1;
        before &= before - 1ULL;
        if (covered(starts + 3 * j, win, blk.global, gz, gy, gx)) return;
    }
    float s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = accum[c * avox + a];
    unsigned long long rest = near & ~((1ULL << p) - 1ULL);  // p itself and later neighbours
    while (rest) {
        const int j = __ffsll((long long)rest) - 1;
        rest &= rest - 1ULL;
        const int* sj = starts + 3 * j;
        if (j != p && !covered(sj, win, blk.global, gz, gy, gx)) continue;
        const size_t o = ((size_t)(gz - sj[0]) * win.patch[1] + (gy - sj[1])) * win.patch[2] + (gx - sj[2]);
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] += pred[((size_t)j * C + c) * pvox + o];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) accum[c * avox + a] = s[c];
}

// number of patch starts along one axis whose trimmed output covers g
__device__ __forceinline__ int axis_count(int g, int dim, int patch, int overlap, int trim) {
    const int stride = patch - overlap;
    const int span = dim - patch + stride;             // range(0, span, stride)
    if (span <= 0) return 0;
    const int nstarts = (span + stride - 1) / stride;
    const int out = patch - 2 * trim;
    int cnt = 0;
    int k = (g - trim) / stride;
    if (g - trim < 0) return 0;
    if (k > nstarts - 1) k = nstarts - 1;
    for (; k >= 0; --k) {
        const int s0 = k * stride + trim;
        if (s0 + out <= g) break;
        if (g >= s0 && g < min(s0 + out, dim)) ++cnt;
    }
    return cnt;
}

// The same overlap-add with four x-neighbours per thread (r03): 16-byte loads and stores of the patch
// outputs and of the accumulator instead of 4-byte ones. Four neighbours share their set of covering
// patches whenever every box edge along x falls on a multiple of four relative to the accumulator --
// starts that agree mod 4, a trim and row lengths that are multiples of 4 (the default geometry: starts
// are multiples of 64, trim 8) -- which a block checks for the patches near its own with one ballot;
// otherwise its threads walk their four voxels one by one through the same code (V = 1). Ownership and
// the order of the additions are the scalar kernel's: same bits.
template <int C, int V>
__device__ __forceinline__ void stitch_voxels(const float* __restrict__ pred, const int* __restrict__ starts,
                                              const exaspim_window& win, float* __restrict__ accum,
                                              const exaspim_block& blk, int p, unsigned long long near,
                                              int z, int y, int x, size_t pvox, size_t avox) {
    const int* sp = starts + 3 * p;
    const int gz = sp[0] + win.trim + z, gy = sp[1] + win.trim + y, gx = sp[2] + win.trim + x;
    if (gz >= blk.global[0] || gy >= blk.global[1] || gx >= blk.global[2]) return;
    const int lz = gz - blk.origin[0], ly = gy - blk.origin[1], lx = gx - blk.origin[2];
    if ((unsigned)lz >= (unsigned)blk.dims[0] || (unsigned)ly >= (unsigned)blk.dims[1] ||
        (unsigned)lx >= (unsigned)blk.dims[2])
        return;
    const size_t a = ((size_t)lz * blk.dims[1] + ly) * blk.dims[2] + lx;
    unsigned long long before = near & ((1ULL << p) - 1ULL);
    while (before) {
        const int j = __ffsll((long long)before) - 1;
        before &= before - 1ULL;
        if (covered(starts + 3 * j, win, blk.global, gz, gy, gx)) return;
    }
    float s[C][V];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(accum + c * avox + a);
            s[c][0] = t.x; s[c][V > 1 ? 1 : 0] = t.y; s[c][V > 2 ? 2 : 0] = t.z; s[c][V > 3 ? 3 : 0] = t.w;
        } else {
            s[c][0] = accum[c * avox + a];
        }
    }
    unsigned long long rest = near & ~((1ULL << p) - 1ULL);  // p itself and later neighbours
    while (rest) {
        const int j = __ffsll((long long)rest) - 1;
        rest &= rest - 1ULL;
        const int* sj = starts + 3 * j;
        if (j != p && !covered(sj, win, blk.global, gz, gy, gx)) continue;
        const size_t o = ((size_t)(gz - sj[0]) * win.patch[1] + (gy - sj[1])) * win.patch[2] + (gx - sj[2]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* q = pred + ((size_t)j * C + c) * pvox + o;
            if (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(q);
                s[c][0] += t.x; s[c][V > 1 ? 1 : 0] += t.y; s[c][V > 2 ? 2 : 0] += t.z; s[c][V > 3 ? 3 : 0] += t.w;
            } else {
                s[c][0] += q[0];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (V == 4)
            *reinterpret_cast<float4*>(accum + c * avox + a) =
                make_float4(s[c][0], s[c][V > 1 ? 1 : 0], s[c][V > 2 ? 2 : 0], s[c][V > 3 ? 3 : 0]);
        else
            accum[c * avox + a] = s[c][0];
    }
}

// Grid: x = blocks of 4-voxel groups of one trimmed output plane, y = patch * trimmed depth (n <= 64,
// trimmed row length a multiple of 4: the launcher sends everything else to stitch_kernel).
template <int C>
__global__ __launch_bounds__(256) void stitch4_kernel(const float* __restrict__ pred,
                                                      const int* __restrict__ starts, int n,
                                                      exaspim_window win, float* __restrict__ accum,
                                                      exaspim_block blk) {
    const int oz = win.patch[0] - 2 * win.trim, oy = win.patch[1] - 2 * win.trim,
              ox = win.patch[2] - 2 * win.trim;
    const size_t pvox = (size_t)win.patch[0] * win.patch[1] * win.patch[2];
    const size_t avox = (size_t)blk.dims[0] * blk.dims[1] * blk.dims[2];
    const int gpr = ox >> 2;                                  // groups per row
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // group inside the trimmed plane
    const int y = i / gpr, x = (i - y * gpr) * 4;
    const int p = blockIdx.y / oz, z = blockIdx.y - p * oz;
    const int* sp = starts + 3 * p;
    __shared__ unsigned long long near_mask, odd_mask;
    if (threadIdx.x < 64) {
        const int j = threadIdx.x;
        bool hit = false, odd = false;
        if (j < n) {
            const int* sj = starts + 3 * j;
            const int dz = sj[0] - sp[0], dy = sj[1] - sp[1], dx = sj[2] - sp[2];
            hit = dz > -oz && dz < oz && dy > -oy && dy < oy && dx > -ox && dx < ox;
            odd = hit && (dx & 3) != 0;
        }
        const unsigned long long m = __ballot(hit), mo = __ballot(odd);
        if (threadIdx.x == 0) { near_mask = m; odd_mask = mo; }
    }
    __syncthreads();
    if (y >= oy) return;
    const unsigned long long near = near_mask;
    // every x edge of the near boxes on a multiple of four of the accumulator's rows, 16-byte aligned bases
    const bool fast = odd_mask == 0 && (win.trim & 3) == 0 && (blk.dims[2] & 3) == 0 && ((blk.global[2] - sp[2] - win.trim) & 3) == 0 &&
                      ((sp[2] + win.trim - blk.origin[2]) & 3) == 0 && (win.patch[2] & 3) == 0 &&
                      (((uintptr_t)pred | (uintptr_t)accum) & 15) == 0;
    if (fast) {
        stitch_voxels<C, 4>(pred, starts, win, accum, blk, p, near, z, y, x, pvox, avox);
    } else {
        for (int v = 0; v < 4; ++v)
            stitch_voxels<C, 1>(pred, starts, win, accum, blk, p, near, z, y, x + v, pvox, avox);
    }
}

// Grid: x = voxels of one row (VEC per thread), y = row, z = plane; the z and y counts are scalar.
// VEC = 4 when rows are whole float4s (launcher): a voxel covered once keeps its bits under a
// division by 1.0f, so a thread divides its four voxels alike and skips the row segment only when
// none of them is covered more than once.
template <int VEC>
__global__ __launch_bounds__(256) void finalize_kernel(float* __restrict__ accum, int channels,
                                                       exaspim_window win, exaspim_block blk) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (x >= blk.dims[2]) return;
    const int y = blockIdx.y, z = blockIdx.z;
    const size_t avox = (size_t)blk.dims[0] * blk.dims[1] * blk.dims[2];
    const int cz = axis_count(z + blk.origin[0], blk.global[0], win.patch[0], win.overlap[0], win.trim);
    const int cy = axis_count(y + blk.origin[1], blk.global[1], win.patch[1], win.overlap[1], win.trim);
    float wgt[VEC];
    bool any = false;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int cnt = cz * cy * axis_count(x + k + blk.origin[2], blk.global[2], win.patch[2], win.overlap[2], win.trim);
        any |= cnt > 1;
        // the reference counts in float16 (inference.py:92, accum_wgt += 1): 2048 + 1 rounds
        // back to 2048, so its weights stop there (only strides of a voxel or two get that far)
        wgt[k] = (float)max(min(cnt, 2048), 1);
    }
    if (!any) return;
    const size_t i = ((size_t)z * blk.dims[1] + y) * blk.dims[2] + x;
    for (int c = 0; c < channels; ++c) {
        float* const p = accum + c * avox + i;
        if (VEC == 4) {
            float4 v = *reinterpret_cast<float4*>(p);
            v.x = __fdiv_rn(v.x, wgt[0]); v.y = __fdiv_rn(v.y, wgt[1]);
            v.z = __fdiv_rn(v.z, wgt[VEC > 2 ? 2 : 0]); v.w = __fdiv_rn(v.w, wgt[VEC > 3 ? 3 : 0]);
            *reinterpret_cast<float4*>(p) = v;
        } else {
            *p = __fdiv_rn(*p, wgt[0]);
        }
    }
}

// float32 -> IEEE half, round to nearest even (what numpy's astype(float16) does), eight
// values per thread: 32 B in, 16 B out
__global__ __launch_bounds__(256) void export_f16_kernel(const float* __restrict__ src,
                                                        unsigned short* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n8 = n / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i];
        const float4 b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            u[k] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * k]) |
                   ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * k + 1]) << 16);
        reinterpret_cast<uint4*>(dst)[i] = make_uint4(u[0], u[1], u[2], u[3]);
    }
    // the last n % 8 values
    const size_t t = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = __builtin_bit_cast(unsigned short, (_Float16)src[t]);
}

static inline unsigned stream_grid(size_t items) {
    const size_t blocks = (items + 255) / 256;
    return (unsigned)(blocks < 8192 ? (blocks ? blocks : 1) : 8192);
}

static int check_block(const exaspim_block* b, const char* who) {
    EXA_CHECK_ARG(b != nullptr, "%s: NULL block", who);
    for (int i = 0; i < 3; ++i)
        EXA_CHECK_ARG(b->dims[i] > 0 && b->global[i] > 0 && b->origin[i] >= 0 &&
                          b->origin[i] + b->dims[i] <= b->global[i],
                      "%s: block axis %d: dims %d origin %d global %d", who, i, b->dims[i],
                      b->origin[i], b->global[i]);
    return EXASPIM_OK;
}

static int check_window(const exaspim_window* w, const char* who) {
    EXA_CHECK_ARG(w != nullptr, "%s: NULL window", who);
    for (int i = 0; i < 3; ++i)
        EXA_CHECK_ARG(w->patch[i] > 0 && w->overlap[i] >= 0 && w->overlap[i] < w->patch[i] &&
                          w->trim >= 0 && 2 * w->trim < w->patch[i],
                      "%s: window axis %d: patch %d overlap %d trim %d", who, i, w->patch[i],
                      w->overlap[i], w->trim);
    return EXASPIM_OK;
}

}  // namespace exaspim

using namespace exaspim;

extern "C" int exaspim_synth_volume_u16(uint16_t* vol_dev, const exaspim_block* blk,
                                        uint64_t seed, void* stream) {
    if (int rc = check_block(blk, "synth")) return rc;
    EXA_CHECK_ARG(vol_dev != nullptr, "synth: NULL volume");
    const size_t total = (size_t)blk->dims[0] * blk->dims[1] * blk->dims[2];
    synth_kernel<<<stream_grid(total), 256, 0, (hipStream_t)stream>>>(vol_dev, *blk, seed);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

extern "C" int exaspim_histogram(const void* vol_dev, int32_t vox_dtype, size_t n, double clip,
                                 int32_t has_clip, int32_t pass, uint32_t prefix,
                                 uint64_t* hist_dev, void* stream) {
    EXA_CHECK_ARG(vol_dev && hist_dev, "histogram: NULL pointer");
    EXA_CHECK_ARG(pass == 0 || (pass == 1 && vox_dtype == EXASPIM_VOX_F32),
                  "histogram: pass %d invalid for voxel dtype %d", pass, vox_dtype);
    if (n == 0) return EXASPIM_OK;
    unsigned long long* h = reinterpret_cast<unsigned long long*>(hist_dev);
    const unsigned grid = (unsigned)((n + 256 * 64 - 1) / (256 * 64) < 2048
                                         ? ((n + 256 * 64 - 1) / (256 * 64) ? (n + 256 * 64 - 1) / (256 * 64) : 1)
                                         : 2048);
    hipStream_t s = (hipStream_t)stream;
    // integer voxels: the clip as an integer threshold and the bin of a clipped voxel (a fractional clip
    // gets the bin above its floor); a clip below the dtype has no bin to stand in
    int thr = 0x7fffffff, cbin = 0;
    if (vox_dtype != EXASPIM_VOX_F32 && has_clip) {
        const double lo = vox_dtype == EXASPIM_VOX_I16 ? -32768.0 : 0.0, hi = vox_dtype == EXASPIM_VOX_U8 ? 255.0 : vox_dtype == EXASPIM_VOX_U16 ? 65535.0 : 32767.0;
        EXA_CHECK_ARG(clip == clip && ceil(clip) >= lo, "histogram: clip %g lies below every voxel of dtype %d", clip, vox_dtype);
        if (clip < hi) { thr = (int)floor(clip); cbin = (int)ceil(clip); }
    }
    const bool vec = ((uintptr_t)vol_dev & 15) == 0;
    switch (vox_dtype) {
        case EXASPIM_VOX_U8:
            if (vec) histogram_int_kernel<uint8_t><<<grid, 256, 0, s>>>(static_cast<const uint8_t*>(vol_dev), n, thr, cbin, 0, 0, h);
            else histogram_kernel<EXASPIM_VOX_U8><<<grid, 256, 0, s>>>(vol_dev, n, clip, has_clip, pass, prefix, 0, h);
            break;
        case EXASPIM_VOX_U16:
            if (vec) histogram_int_kernel<uint16_t><<<grid, 256, 0, s>>>(static_cast<const uint16_t*>(vol_dev), n, thr, cbin, 0, 0, h);
            else histogram_kernel<EXASPIM_VOX_U16><<<grid, 256, 0, s>>>(vol_dev, n, clip, has_clip, pass, prefix, 0, h);
            break;
        case EXASPIM_VOX_I16:
            if (vec) histogram_int_kernel<int16_t><<<grid, 256, 0, s>>>(static_cast<const int16_t*>(vol_dev), n, thr, cbin, 32768, 32768 - 4096, h);
            else histogram_kernel<EXASPIM_VOX_I16><<<grid, 256, 0, s>>>(vol_dev, n, clip, has_clip, pass, prefix, 32768 - 4096, h);
            break;
        case EXASPIM_VOX_F32:
            // non-negative floats up to ~1e5 have key halves 0x8000..0xC7C3
            histogram_kernel<EXASPIM_VOX_F32><<<grid, 256, 0, s>>>(vol_dev, n, clip, has_clip, pass, prefix,
                                                                 pass == 0 ? 0x8000 + 0x0800 : 0, h);
            break;
        default:
            set_error("histogram: unknown voxel dtype %d", vox_dtype);
            return EXASPIM_E_INVALID;
    }
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

extern "C" int exaspim_histogram_wide(const void* vol_dev, int32_t vox_dtype, size_t n, double clip,
                                      int32_t has_clip, int32_t pass, uint64_t prefix,
                                      uint64_t* hist_dev, void* stream) {
    EXA_CHECK_ARG(vol_dev && hist_dev, "histogram: NULL pointer");
    EXA_CHECK_ARG(vox_dtype == EXASPIM_VOX_F64, "histogram_wide: voxel dtype %d (only EXASPIM_VOX_F64)", vox_dtype);
    EXA_CHECK_ARG(pass >= 0 && pass <= 3, "histogram_wide: pass %d outside 0..3", pass);
    EXA_CHECK_ARG(pass == 0 ? prefix == 0 : (prefix >> (16 * pass)) == 0, "histogram_wide: prefix wider than %d bits", 16 * pass);
    if (n == 0) return EXASPIM_OK;
    const size_t blocks = (n + 256 * 64 - 1) / (256 * 64);
    const unsigned grid = (unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048);
    // pass 0: non-negative doubles from ~1e-60 up to ~1e240 have key halves 0xB000..0xEFFF
    histogram_f64_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
        vol_dev, n, clip, has_clip, pass, (unsigned long long)prefix, pass == 0 ? 0xB000 : 0,
        reinterpret_cast<unsigned long long*>(hist_dev));
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

template <int LAYOUT>
static int gather_launch(const void* vol_dev, int32_t vox_dtype, const exaspim_block* blk,
                         const int32_t* starts_dev, int32_t n, const int32_t patch[3], double clip,
                         int32_t has_clip, double mn, double denom, float* out_dev, hipStream_t s) {
    constexpr int B = LAYOUT == EXASPIM_IN_F32 ? 0 : 1;
    EXA_CHECK_ARG((long long)n * (patch[0] + 2 * B) <= 65535 && patch[1] + 2 * B <= 65535, "gather: grid too large");
    // table path: unsigned integers clipped to a small integer maximum
    // (table + row map within the 64 KiB a launch gets without asking for more)
    if (has_clip && clip >= 0.0 && clip <= 16383.0 && clip == (double)(int)clip &&
        ((size_t)clip + 1) * sizeof(float) + (size_t)(patch[1] + 2 * B) * sizeof(int) <= 65536 &&
        (vox_dtype == EXASPIM_VOX_U8 || vox_dtype == EXASPIM_VOX_U16)) {
        const int ci = (int)clip;
        const size_t lds = ((size_t)ci + 1) * sizeof(float) + (size_t)(patch[1] + 2 * B) * sizeof(int);   // table + row map
        const unsigned blocks = (unsigned)((long long)n * (patch[0] + 2 * B));
        if (vox_dtype == EXASPIM_VOX_U8)
            gather_lut_kernel<uint8_t, LAYOUT><<<blocks, 256, lds, s>>>(
                static_cast<const uint8_t*>(vol_dev), *blk, starts_dev, patch[0], patch[1], patch[2], ci, mn,
                denom, out_dev);
        else
            gather_lut_kernel<uint16_t, LAYOUT><<<blocks, 256, lds, s>>>(
                static_cast<const uint16_t*>(vol_dev), *blk, starts_dev, patch[0], patch[1], patch[2], ci, mn,
                denom, out_dev);
        EXA_CHECK_HIP(hipGetLastError());
        return EXASPIM_OK;
    }
    const dim3 grid((patch[2] + 2 * B + 127) / 128, patch[1] + 2 * B, n * (patch[0] + 2 * B));
#define GATHER(V) gather_kernel<V, LAYOUT><<<grid, 128, 0, s>>>(vol_dev, *blk, starts_dev, patch[0], patch[1], patch[2], clip, has_clip, mn, denom, out_dev)
    switch (vox_dtype) {
        case EXASPIM_VOX_U8: GATHER(EXASPIM_VOX_U8); break;
        case EXASPIM_VOX_U16: GATHER(EXASPIM_VOX_U16); break;
        case EXASPIM_VOX_I16: GATHER(EXASPIM_VOX_I16); break;
        case EXASPIM_VOX_F32: GATHER(EXASPIM_VOX_F32); break;
        case EXASPIM_VOX_F64: GATHER(EXASPIM_VOX_F64); break;
        default:
            set_error("gather: unknown voxel dtype %d", vox_dtype);
            return EXASPIM_E_INVALID;
    }
#undef GATHER
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

extern "C" int exaspim_gather_patches_as(const void* vol_dev, int32_t vox_dtype,
                                         const exaspim_block* blk, const int32_t* starts_dev,
                                         int32_t n, const int32_t patch[3], double clip,
                                         int32_t has_clip, double mn, double denom, int32_t layout,
                                         void* out_dev, void* stream) {
    if (int rc = check_block(blk, "gather")) return rc;
    EXA_CHECK_ARG(vol_dev && starts_dev && out_dev && patch, "gather: NULL pointer");
    EXA_CHECK_ARG(n > 0 && patch[0] > 0 && patch[1] > 0 && patch[2] > 0, "gather: empty batch");
    hipStream_t s = (hipStream_t)stream;
    float* const out = static_cast<float*>(out_dev);
    switch (layout) {
        case EXASPIM_IN_F32:
            return gather_launch<EXASPIM_IN_F32>(vol_dev, vox_dtype, blk, starts_dev, n, patch, clip, has_clip, mn, denom, out, s);
        case EXASPIM_IN_PADDED_F32:
            return gather_launch<EXASPIM_IN_PADDED_F32>(vol_dev, vox_dtype, blk, starts_dev, n, patch, clip, has_clip, mn, denom, out, s);
        case EXASPIM_IN_PADDED_SPLIT_F16:
            return gather_launch<EXASPIM_IN_PADDED_SPLIT_F16>(vol_dev, vox_dtype, blk, starts_dev, n, patch, clip, has_clip, mn, denom, out, s);
        case EXASPIM_IN_PADDED_SPLIT_BF16:
            return gather_launch<EXASPIM_IN_PADDED_SPLIT_BF16>(vol_dev, vox_dtype, blk, starts_dev, n, patch, clip, has_clip, mn, denom, out, s);
    }
    set_error("gather: unknown input layout %d", layout);
    return EXASPIM_E_INVALID;
}

extern "C" int exaspim_gather_patches(const void* vol_dev, int32_t vox_dtype,
                                      const exaspim_block* blk, const int32_t* starts_dev,
                                      int32_t n, const int32_t patch[3], double clip,
                                      int32_t has_clip, double mn, double denom, float* out_dev,
                                      void* stream) {
    return exaspim_gather_patches_as(vol_dev, vox_dtype, blk, starts_dev, n, patch, clip, has_clip, mn,
                                     denom, EXASPIM_IN_F32, out_dev, stream);
}

extern "C" int exaspim_stitch_accumulate(const float* pred_dev, const int32_t* starts_dev,
                                         int32_t n, int32_t channels, const exaspim_window* win,
                                         float* accum_dev, const exaspim_block* blk,
                                         void* stream) {
    if (int rc = check_block(blk, "stitch")) return rc;
    if (int rc = check_window(win, "stitch")) return rc;
    EXA_CHECK_ARG(pred_dev && starts_dev && accum_dev, "stitch: NULL pointer");
    EXA_CHECK_ARG(n > 0 && channels >= 1 && channels <= 4, "stitch: n %d channels %d", n, channels);
    const int oz = win->patch[0] - 2 * win->trim, oy = win->patch[1] - 2 * win->trim,
              ox = win->patch[2] - 2 * win->trim;
    EXA_CHECK_ARG((long long)n * oz <= 65535 && (long long)oy * ox < 0x7fffffffLL, "stitch: grid too large");
    const dim3 grid((unsigned)(((long long)oy * ox + 255) / 256), n * oz);
    hipStream_t s = (hipStream_t)stream;
    if (n <= 64 && ox % 4 == 0) {       // four x-neighbours per thread (falls back per block where they differ)
        const dim3 grid4((unsigned)(((long long)oy * (ox / 4) + 255) / 256), n * oz);
        switch (channels) {
            case 1: stitch4_kernel<1><<<grid4, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
            case 2: stitch4_kernel<2><<<grid4, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
            case 3: stitch4_kernel<3><<<grid4, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
            case 4: stitch4_kernel<4><<<grid4, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
        }
        EXA_CHECK_HIP(hipGetLastError());
        return EXASPIM_OK;
    }
    switch (channels) {
        case 1: stitch_kernel<1><<<grid, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
        case 2: stitch_kernel<2><<<grid, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
        case 3: stitch_kernel<3><<<grid, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
        case 4: stitch_kernel<4><<<grid, 256, 0, s>>>(pred_dev, starts_dev, n, *win, accum_dev, *blk); break;
    }
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

extern "C" int exaspim_stitch_finalize(float* accum_dev, int32_t channels,
                                       const exaspim_window* win, const exaspim_block* blk,
                                       void* stream) {
    if (int rc = check_block(blk, "finalize")) return rc;
    if (int rc = check_window(win, "finalize")) return rc;
    EXA_CHECK_ARG(accum_dev && channels >= 1, "finalize: bad arguments");
    EXA_CHECK_ARG(blk->dims[0] <= 65535 && blk->dims[1] <= 65535, "finalize: block too large");
    if (blk->dims[2] % 4 == 0 && ((uintptr_t)accum_dev & 15) == 0) {
        const dim3 grid((blk->dims[2] / 4 + 255) / 256, blk->dims[1], blk->dims[0]);
        finalize_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(accum_dev, channels, *win, *blk);
    } else {
        const dim3 grid((blk->dims[2] + 255) / 256, blk->dims[1], blk->dims[0]);
        finalize_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(accum_dev, channels, *win, *blk);
    }
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}

extern "C" int exaspim_export_f16(const float* src_dev, void* dst_dev, size_t n, void* stream) {
    EXA_CHECK_ARG(src_dev && dst_dev, "export_f16: NULL buffer");
    EXA_CHECK_ARG(((uintptr_t)src_dev & 15) == 0 && ((uintptr_t)dst_dev & 15) == 0,
                  "export_f16: buffers must be 16-byte aligned");
    if (n == 0) return EXASPIM_OK;
    export_f16_kernel<<<stream_grid(n / 8 + 8), 256, 0, (hipStream_t)stream>>>(
        src_dev, static_cast<unsigned short*>(dst_dev), n);
    EXA_CHECK_HIP(hipGetLastError());
    return EXASPIM_OK;
}
