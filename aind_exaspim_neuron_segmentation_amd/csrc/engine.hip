// UNet3D.forward (machine_learning/unet3d.py:77-105) as a fixed sequence of
// kernel launches on the caller's stream, plus the extern "C" model API.
//
// Workspace: per pyramid level l (voxels n*d*h*w / 8^l) three channels-last
// buffers -- skip[l] (x1..x4, alive until the decoder consumes them), A[l] and
// B[l] (ping-pong) -- each sized for the widest tensor stored at that level.

#include <algorithm>
#include <cstdlib>
#include <new>
#include <string_view>

#include "common.h"

namespace exaspim {

constexpr float kLeakySlope = 0.01f;  // unet3d.py:145,148

struct Workspace {
    size_t skip[5], a[5], b[5];
    size_t xpad;  // zero-bordered float32 copy of the input patches (inc.0)
    size_t bytes;
};

}  // namespace exaspim

// Optional per-layer timing with HIP events recorded on the launch stream
// (exaspim_unet_timing_*): a ring of event pairs around the MFMA convolutions
// selected by a layer mask.
struct LayerTimer {
    static constexpr int kRing = 16384;
    uint32_t mask = 0;
    int next = 0, used = 0;
    hipEvent_t start[kRing], stop[kRing];
    int layer[kRing];
    bool created = false;
};

struct exaspim_unet {
    exaspim::UNetPlan plan;
    int device;
    const char* packed;  // device image (caller-owned)
    LayerTimer* timer = nullptr;
    uint32_t options = 0;   // EXASPIM_OPT_* (exaspim_unet_set_options)
};

namespace exaspim {

static bool level_dims_ok(int d, int h, int w) {
    return d > 0 && h > 0 && w > 0 && d % 16 == 0 && h % 16 == 0 && w % 16 == 0;
}

// pyramid level of the i-th MFMA conv (inc.3 = 0 ... up4.3 = 16)
static int conv_level(int i) {
    if (i == 0) return 0;
    if (i <= 8) return (i + 1) / 2;          // down1..down4 -> 1..4
    return 3 - (i - 9) / 2;                  // up1..up4 -> 3..0
}

static Workspace make_workspace(const UNetPlan& p, int n, int d, int h, int w) {
    // widest tensor stored at each level: every input and output of its convs
    // (the pooled / upsampled tensors are inputs of the level's first conv)
    int maxc[5] = {p.c0p, 0, 0, 0, 0};
    for (int i = 0; i < kNumMfmaConvs; ++i) {
        const ConvLayer& L = p.conv[i];
        int& m = maxc[conv_level(i)];
        m = std::max(m, std::max(std::max(L.ca, L.cb), L.cout));
    }
    const size_t es = dtype_size(p.dtype);
    Workspace ws;
    size_t off = 0;
    for (int l = 0; l < 5; ++l) {
        const size_t vox = (size_t)n * (d >> l) * (h >> l) * (w >> l);
        const size_t sz = align_up(vox * maxc[l] * es, 256);
        ws.skip[l] = off; off += (l < 4 ? sz : 0);
        ws.a[l] = off; off += sz;
        ws.b[l] = off; off += sz;
    }
    ws.xpad = off;
    off += align_up((size_t)n * (d + 2) * (h + 2) * (w + 2) * sizeof(float), 256);
    ws.bytes = off;
    return ws;
}

// x: float32 patches, or (x == nullptr) x_prepared: the first convolution's operand layout
// absmax (optional): float[1 + kNumMfmaConvs + 4], the largest |activation| inc.0, every MFMA
// convolution and every ConvTranspose3d stored (range probe: nothing is fused away, nothing trimmed)
static int forward(exaspim_unet* e, const float* x, float* out, int n, int d, int h, int w,
                   int apply_sigmoid, int trim, void* workspace, size_t workspace_bytes,
                   hipStream_t stream, const void* x_prepared = nullptr, float* absmax = nullptr) {
    const UNetPlan& p = e->plan;
    const Workspace ws = make_workspace(p, n, d, h, w);
    if (workspace_bytes < ws.bytes) {
        set_error("forward: workspace %zu bytes < required %zu", workspace_bytes, ws.bytes);
        return EXASPIM_E_WORKSPACE;
    }
    char* base = static_cast<char*>(workspace);
    auto skip = [&](int l) { return (void*)(base + ws.skip[l]); };
    auto A = [&](int l) { return (void*)(base + ws.a[l]); };
    auto B = [&](int l) { return (void*)(base + ws.b[l]); };
    const int dt = p.dtype;
    int rc;

    // the last conv (up4.3) can run the 1x1x1 head on its accumulators
    const bool fuse_head = !absmax && conv_can_fuse_head(p.conv[kNumMfmaConvs - 1].cout, w, p.out_channels);
    // With the head fused, voxels within "trim" of a patch face are never read again:
    // up4.3 skips them, and up4.0 everything its 3x3x3 consumer does not reach.
    const bool trimmed = fuse_head && trim > 0 && 2 * trim < d && 2 * trim < h && 2 * trim < w;
    // conv 2l (inc.3, down1.3, down2.3, down3.3) can write its own max-pool, the input of level l + 1
    // (EXASPIM_OPT_SEPARATE_POOL, set per handle: run the stand-alone max-pool launches instead -- the
    // tests hold the two to each other bit for bit)
    const bool separate_pool = (e->options & EXASPIM_OPT_SEPARATE_POOL) != 0;
    const bool separate_deep = (e->options & EXASPIM_OPT_SEPARATE_DEEP_POOLS) != 0;   // only inc.3 keeps its pool
    bool fuse_pool[4];
    for (int l = 0; l < 4; ++l)
        fuse_pool[l] = !separate_pool && !(separate_deep && l > 0) && conv_can_fuse_pool(p.dtype, p.conv[2 * l].cout, d >> l, h >> l, w >> l);
    auto conv = [&](int idx, const void* sa, const void* sb, void* dst, int l) -> int {
        const ConvLayer& L = p.conv[idx];
        ConvArgs a;
        // the last conv of an encoder level writes its 2x2x2 max-pool (the next level's input) too
        // split-K scratch: the zero-bordered input copy is dead once inc.0 has run
        a.partial = reinterpret_cast<float*>(base + ws.xpad);
        a.partial_patch_bytes = (size_t)(d + 2) * (h + 2) * (w + 2) * sizeof(float);
        if (idx <= 6 && idx % 2 == 0 && fuse_pool[idx / 2]) a.pool_dst = A(idx / 2 + 1);
        // up4.3 produces [trim, size - trim), up4.0 one voxel more on every face
        const int margin = !trimmed ? 0 : idx == kNumMfmaConvs - 1 ? trim : idx == kNumMfmaConvs - 2 ? trim - 1 : 0;
        const int full[3] = {d >> l, h >> l, w >> l};
        for (int i = 0; i < 3; ++i) {
            a.org[i] = margin;
            a.ext[i] = full[i] - 2 * margin;
        }
        // When the region is a few voxels more than whole z-column tiles along y or x (82 =
        // 10 x 8 + 2 = 5 x 16 + 2 for the default patch), the z-column kernel covers the whole
        // tiles and two launches on 2-voxel-thick tiles the rest, instead of a ninth row and a
        // sixth column of mostly masked 8 x 16 tiles (924 tiles for 718 tiles' worth of voxels).
        int rem_y = 0, rem_x = 0;
        const bool has_head = idx == kNumMfmaConvs - 1 && fuse_head;   // the head runs on z-column tiles only
        if (margin > 0 && !has_head && L.cout % 64 != 0 && full[2] % 16 == 0) {
            rem_y = a.ext[1] - conv_zcol_main_extent(a.ext[1], 1);
            rem_x = a.ext[2] - conv_zcol_main_extent(a.ext[2], 2);
            a.ext[1] -= rem_y;
            a.ext[2] -= rem_x;
        }
        if (idx == kNumMfmaConvs - 1 && fuse_head) {
            a.head_w = reinterpret_cast<const float*>(e->packed + p.head_w_off);
            a.head_b = reinterpret_cast<const float*>(e->packed + p.head_b_off);
            a.head_out = out;
            a.head_oc = p.out_channels;
            a.head_sigmoid = apply_sigmoid;
        }
        a.src_a = sa; a.src_b = sb; a.ca = L.ca; a.cb = L.cb;
        a.weights = e->packed + L.w_off;
        a.weights_paired = L.w2_off ? e->packed + L.w2_off : nullptr;
        a.weights_k32 = L.w3_off ? e->packed + L.w3_off : nullptr;
        a.bias = reinterpret_cast<const float*>(e->packed + L.b_off);
        a.dst = dst; a.cout = L.cout;
        a.n = n; a.d = d >> l; a.h = h >> l; a.w = w >> l;
        a.slope = kLeakySlope;
        LayerTimer* t = e->timer;
        const bool timed = t && (t->mask >> idx & 1u) && t->used < LayerTimer::kRing;
        int slot = 0;
        if (timed) {
            slot = t->next;
            EXA_CHECK_HIP(hipEventRecord(t->start[slot], stream));
        }
        int r = launch_conv3x3x3(dt, a, stream);
        if (timed && r == EXASPIM_OK) {
            EXA_CHECK_HIP(hipEventRecord(t->stop[slot], stream));
            t->layer[slot] = idx;
            t->next = (slot + 1) % LayerTimer::kRing;
            t->used++;
        }
        if (r == EXASPIM_OK && rem_y > 0) {      // rows [org_y + main, org_y + main + rem_y), every x of the region
            ConvArgs b = a;
            b.org[1] = a.org[1] + a.ext[1]; b.ext[1] = rem_y;
            b.ext[2] = a.ext[2] + rem_x;
            r = launch_conv3x3x3_thin(dt, b, stream);
        }
        if (r == EXASPIM_OK && rem_x > 0) {      // columns beyond the whole tiles, rows of the main part
            ConvArgs b = a;
            b.org[2] = a.org[2] + a.ext[2]; b.ext[2] = rem_x;
            r = launch_conv3x3x3_thin(dt, b, stream);
        }
        if (r == EXASPIM_OK && absmax)
            r = launch_absmax(dt, dst, (size_t)n * (d >> l) * (h >> l) * (w >> l) * L.cout * dtype_size(dt),
                              absmax + 1 + idx, stream);
        return r;
    };
#define RUN(expr) do { rc = (expr); if (rc) return rc; } while (0)

    // encoder (unet3d.py:93-97)
    RUN(launch_conv_first(dt, x, x ? reinterpret_cast<float*>(base + ws.xpad)
                                   : const_cast<float*>(static_cast<const float*>(x_prepared)),
                          reinterpret_cast<const float*>(e->packed + p.first_w_off),
                          reinterpret_cast<const float*>(e->packed + p.first_b_off), A(0), n, d,
                          h, w, p.c0p, kLeakySlope, stream, (e->options & EXASPIM_OPT_FIRST_PER_GROUP) != 0));
    if (absmax) RUN(launch_absmax(dt, A(0), (size_t)n * d * h * w * p.c0p * dtype_size(dt), absmax, stream));
    RUN(conv(0, A(0), nullptr, skip(0), 0));                      // x1
    for (int l = 1; l <= 4; ++l) {
        const ConvLayer& L0 = p.conv[2 * l - 1];
        const void* prev = skip(l - 1);
        if (!fuse_pool[l - 1])
            RUN(launch_maxpool2(dt, prev, A(l), n, d >> (l - 1), h >> (l - 1), w >> (l - 1), L0.ca,
                                stream));
        RUN(conv(2 * l - 1, A(l), nullptr, B(l), l));
        RUN(conv(2 * l, B(l), nullptr, l < 4 ? skip(l) : A(l), l));  // x2..x4, x5 in A(4)
    }
    // decoder (unet3d.py:100-103): y = DoubleConv(cat[skip, up(prev)])
    const void* prev = A(4);
    for (int l = 3; l >= 0; --l) {
        const int i0 = 9 + 2 * (3 - l);  // up1.0 = conv[9], up2.0 = conv[11], ...
        const ConvLayer& L0 = p.conv[i0];
        if (p.convt) {
            const ConvTLayer& U = p.up[3 - l];
            RUN(launch_convt2(dt, prev, e->packed + U.w_off,
                              reinterpret_cast<const float*>(e->packed + U.b_off), A(l), n,
                              d >> (l + 1), h >> (l + 1), w >> (l + 1), U.cin, U.cout, stream));
            if (absmax)     // (a transposed convolution is not bounded by its input like the interpolation is)
                RUN(launch_absmax(dt, A(l), (size_t)n * (d >> l) * (h >> l) * (w >> l) * U.cout * dtype_size(dt),
                                  absmax + 1 + kNumMfmaConvs + (3 - l), stream));
        } else {
            // up4.0 reads the upsampled tensor one voxel beyond its own trimmed output
            RUN(launch_upsample2(dt, prev, A(l), n, d >> (l + 1), h >> (l + 1), w >> (l + 1), L0.cb,
                                 trimmed && l == 0 && trim > 2 ? trim - 2 : 0, stream,
                                 (e->options & EXASPIM_OPT_PLAIN_UPSAMPLE) != 0,
                                 (e->options & EXASPIM_OPT_UPSAMPLE_PER_THREAD) != 0));
        }
        RUN(conv(i0, skip(l), A(l), B(l), l));
        RUN(conv(i0 + 1, B(l), nullptr, A(l), l));
        prev = A(l);
    }
    if (!fuse_head)
    RUN(launch_head(dt, prev, reinterpret_cast<const float*>(e->packed + p.head_w_off),
                    reinterpret_cast<const float*>(e->packed + p.head_b_off), out, n, d, h, w,
                    p.c0p, p.out_channels, apply_sigmoid, stream));
#undef RUN
    return EXASPIM_OK;
}

}  // namespace exaspim

using namespace exaspim;

extern "C" int exaspim_abi_version(void) { return EXASPIM_ABI_VERSION; }
extern "C" const char* exaspim_last_error(void) { return get_error(); }

extern "C" size_t exaspim_unet_param_count(const int32_t channels[5], int32_t out_channels,
                                           int32_t dtype) {
    UNetPlan p;
    if (!make_plan(channels, out_channels, dtype, &p)) return 0;
    return p.n_params;
}

extern "C" size_t exaspim_unet_packed_bytes(const int32_t channels[5], int32_t out_channels,
                                            int32_t dtype) {
    UNetPlan p;
    if (!make_plan(channels, out_channels, dtype, &p)) return 0;
    return p.packed_bytes;
}

extern "C" int exaspim_unet_pack_weights(const int32_t channels[5], int32_t out_channels,
                                         int32_t dtype, const float* params, size_t n_params,
                                         void* packed_host, size_t packed_bytes) {
    UNetPlan p;
    if (!make_plan(channels, out_channels, dtype, &p)) return EXASPIM_E_INVALID;
    EXA_CHECK_ARG(params && packed_host, "pack_weights: NULL pointer");
    EXA_CHECK_ARG(n_params == p.n_params, "pack_weights: got %zu parameters, expected %zu",
                  n_params, p.n_params);
    EXA_CHECK_ARG(packed_bytes == p.packed_bytes, "pack_weights: buffer %zu bytes, expected %zu",
                  packed_bytes, p.packed_bytes);
    return pack_weights(p, params, packed_host);
}

extern "C" int exaspim_unet_create(const int32_t channels[5], int32_t out_channels,
                                   int32_t dtype, int32_t device, const void* packed_dev,
                                   size_t packed_bytes, exaspim_unet** out) {
    EXA_CHECK_ARG(out != nullptr && packed_dev != nullptr, "create: NULL pointer");
    UNetPlan p;
    if (!make_plan(channels, out_channels, dtype, &p)) return EXASPIM_E_INVALID;
    EXA_CHECK_ARG(packed_bytes == p.packed_bytes, "create: packed image %zu bytes, expected %zu",
                  packed_bytes, p.packed_bytes);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        set_error("create: device %d not available (%d HIP devices visible)", device, ndev);
        return EXASPIM_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    EXA_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string_view(prop.gcnArchName).substr(0, 6) != "gfx950") {
        set_error("create: device %d is %s; this library is built for gfx950 only", device,
                  prop.gcnArchName);
        return EXASPIM_E_NODEVICE;
    }
    exaspim_unet* e = new (std::nothrow) exaspim_unet;
    EXA_CHECK_ARG(e != nullptr, "create: out of host memory");
    e->plan = p;
    e->device = device;
    e->packed = static_cast<const char*>(packed_dev);
    *out = e;
    return EXASPIM_OK;
}

extern "C" void exaspim_unet_destroy(exaspim_unet* h) {
    if (!h) return;
    if (h->timer) {
        if (h->timer->created)
            for (int i = 0; i < LayerTimer::kRing; ++i) {
                (void)hipEventDestroy(h->timer->start[i]);
                (void)hipEventDestroy(h->timer->stop[i]);
            }
        delete h->timer;
    }
    delete h;
}

extern "C" int exaspim_unet_timing_begin(exaspim_unet* h, uint32_t conv_mask) {
    EXA_CHECK_ARG(h != nullptr, "timing_begin: NULL handle");
    if (!h->timer) h->timer = new (std::nothrow) LayerTimer;
    EXA_CHECK_ARG(h->timer != nullptr, "timing_begin: out of host memory");
    LayerTimer* t = h->timer;
    if (!t->created) {
        for (int i = 0; i < LayerTimer::kRing; ++i) {
            EXA_CHECK_HIP(hipEventCreate(&t->start[i]));
            EXA_CHECK_HIP(hipEventCreate(&t->stop[i]));
        }
        t->created = true;
    }
    t->mask = conv_mask;
    t->next = 0;
    t->used = 0;
    return EXASPIM_OK;
}

extern "C" int exaspim_unet_timing_read(exaspim_unet* h, double ms_sum[17], int32_t count[17]) {
    EXA_CHECK_ARG(h && h->timer && ms_sum && count, "timing_read: timing was not started");
    LayerTimer* t = h->timer;
    for (int i = 0; i < kNumMfmaConvs; ++i) { ms_sum[i] = 0.0; count[i] = 0; }
    for (int s = 0; s < t->used; ++s) {
        EXA_CHECK_HIP(hipEventSynchronize(t->stop[s]));
        float ms = 0.f;
        EXA_CHECK_HIP(hipEventElapsedTime(&ms, t->start[s], t->stop[s]));
        ms_sum[t->layer[s]] += ms;
        count[t->layer[s]]++;
    }
    t->mask = 0;
    t->used = 0;
    t->next = 0;
    return EXASPIM_OK;
}

extern "C" size_t exaspim_unet_workspace_bytes(const exaspim_unet* h, int32_t n, int32_t d,
                                               int32_t hgt, int32_t w) {
    if (!h || n <= 0 || !level_dims_ok(d, hgt, w)) {
        set_error("workspace_bytes: bad arguments (n %d, patch %dx%dx%d must be multiples of 16)",
                  n, d, hgt, w);
        return 0;
    }
    return make_workspace(h->plan, n, d, hgt, w).bytes;
}

extern "C" int exaspim_unet_forward(exaspim_unet* h, const float* x_dev, float* out_dev,
                                    int32_t n, int32_t d, int32_t hgt, int32_t w,
                                    int32_t apply_sigmoid, void* workspace_dev,
                                    size_t workspace_bytes, void* stream) {
    EXA_CHECK_ARG(h && x_dev && out_dev && workspace_dev, "forward: NULL pointer");
    EXA_CHECK_ARG(n > 0, "forward: empty batch");
    // The reference's Up.forward pads only two axes by the wrong differences
    // (unet3d.py:281-287), so torch.cat raises for sizes that are not multiples
    // of 16; the same sizes are rejected here.
    EXA_CHECK_ARG(level_dims_ok(d, hgt, w),
                  "forward: patch %dx%dx%d: every dimension must be a positive multiple of 16",
                  d, hgt, w);
    return forward(h, x_dev, out_dev, n, d, hgt, w, apply_sigmoid, 0, workspace_dev, workspace_bytes,
                   (hipStream_t)stream);
}

extern "C" int exaspim_unet_input_layout(const exaspim_unet* h) {
    if (!h) return EXASPIM_E_INVALID;
    switch (h->plan.dtype) {
        case EXASPIM_DT_F32: return EXASPIM_IN_PADDED_F32;
        case EXASPIM_DT_F16: return EXASPIM_IN_PADDED_SPLIT_F16;
        case EXASPIM_DT_BF16: return EXASPIM_IN_PADDED_SPLIT_BF16;
    }
    return EXASPIM_E_INVALID;
}

extern "C" int exaspim_unet_forward_prepared(exaspim_unet* h, const void* x_prepared_dev, float* out_dev,
                                             int32_t n, int32_t d, int32_t hgt, int32_t w,
                                             int32_t apply_sigmoid, int32_t trim,
                                             void* workspace_dev, size_t workspace_bytes,
                                             void* stream) {
    EXA_CHECK_ARG(h && x_prepared_dev && out_dev && workspace_dev, "forward: NULL pointer");
    EXA_CHECK_ARG(n > 0, "forward: empty batch");
    EXA_CHECK_ARG(trim >= 0, "forward: negative trim %d", trim);
    EXA_CHECK_ARG(level_dims_ok(d, hgt, w),
                  "forward: patch %dx%dx%d: every dimension must be a positive multiple of 16",
                  d, hgt, w);
    return forward(h, nullptr, out_dev, n, d, hgt, w, apply_sigmoid, trim, workspace_dev,
                   workspace_bytes, (hipStream_t)stream, x_prepared_dev);
}

extern "C" int exaspim_unet_forward_trimmed(exaspim_unet* h, const float* x_dev, float* out_dev,
                                            int32_t n, int32_t d, int32_t hgt, int32_t w,
                                            int32_t apply_sigmoid, int32_t trim,
                                            void* workspace_dev, size_t workspace_bytes,
                                            void* stream) {
    EXA_CHECK_ARG(h && x_dev && out_dev && workspace_dev, "forward: NULL pointer");
    EXA_CHECK_ARG(n > 0, "forward: empty batch");
    EXA_CHECK_ARG(trim >= 0, "forward: negative trim %d", trim);
    EXA_CHECK_ARG(level_dims_ok(d, hgt, w),
                  "forward: patch %dx%dx%d: every dimension must be a positive multiple of 16",
                  d, hgt, w);
    return forward(h, x_dev, out_dev, n, d, hgt, w, apply_sigmoid, trim, workspace_dev,
                   workspace_bytes, (hipStream_t)stream);
}

extern "C" int exaspim_unet_forward_absmax(exaspim_unet* h, const float* x_dev, float* out_dev,
                                           int32_t n, int32_t d, int32_t hgt, int32_t w,
                                           int32_t apply_sigmoid, float* absmax_dev,
                                           void* workspace_dev, size_t workspace_bytes, void* stream) {
    EXA_CHECK_ARG(h && x_dev && out_dev && workspace_dev && absmax_dev, "forward_absmax: NULL pointer");
    EXA_CHECK_ARG(n > 0, "forward: empty batch");
    EXA_CHECK_ARG(level_dims_ok(d, hgt, w),
                  "forward: patch %dx%dx%d: every dimension must be a positive multiple of 16",
                  d, hgt, w);
    return forward(h, x_dev, out_dev, n, d, hgt, w, apply_sigmoid, 0, workspace_dev, workspace_bytes,
                   (hipStream_t)stream, nullptr, absmax_dev);
}

extern "C" int exaspim_unet_set_options(exaspim_unet* h, uint32_t options) {
    EXA_CHECK_ARG(h != nullptr, "set_options: NULL handle");
    EXA_CHECK_ARG((options & ~(uint32_t)(EXASPIM_OPT_SEPARATE_POOL | EXASPIM_OPT_SEPARATE_DEEP_POOLS | EXASPIM_OPT_PLAIN_UPSAMPLE | EXASPIM_OPT_FIRST_PER_GROUP |
                                          EXASPIM_OPT_UPSAMPLE_PER_THREAD)) == 0,
                  "set_options: unknown option bits 0x%x", options);
    h->options = options;
    return EXASPIM_OK;
}
