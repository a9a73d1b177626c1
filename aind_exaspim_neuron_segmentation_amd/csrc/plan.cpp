// Host-only part of the engine: network plan, BatchNorm folding and packing of
// the weights into MFMA fragment order. No device code, so the non-GPU test
// suite can exercise it.
//
// Reference semantics folded here (machine_learning/unet3d.py:142-149):
//   y = LeakyReLU(BN_eval(conv(x, W) + b))
//   BN_eval(v) = (v - running_mean) / sqrt(running_var + 1e-5) * gamma + beta
// =>  y = LeakyReLU(conv(x, W * s) + (b - running_mean) * s + beta),
//     s = gamma / sqrt(running_var + 1e-5)            (computed in float64)

#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace exaspim {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

const char* get_error() { return g_error; }

static size_t conv_param_count(int cout, int cin) {
    return (size_t)cout * cin * 27 + 5 * (size_t)cout;
}

bool make_plan(const int32_t channels[5], int32_t out_channels, int32_t dtype,
               UNetPlan* plan) {
    if (!channels || !plan) {
        set_error("make_plan: NULL argument");
        return false;
    }
    for (int i = 0; i < 5; ++i) {
        if (channels[i] < 1 || channels[i] > 4096) {
            set_error("channels[%d] = %d out of range", i, channels[i]);
            return false;
        }
    }
    if (channels[4] % 2 || channels[3] % 2 || channels[2] % 2 || channels[1] % 2) {
        set_error("channels[1..4] must be even (trilinear U-Net halves them)");
        return false;
    }
    const bool convt = (dtype & EXASPIM_UP_CONVT) != 0;
    dtype &= 0xff;
    if (channels[3] != channels[4] / 2 || channels[2] != channels[3] / 2 ||
        channels[1] != channels[2] / 2 || channels[0] != channels[1] / 2) {
        set_error("channels must double per level (skip + upsampled = next width)");
        return false;
    }
    if (out_channels < 1 || out_channels > 4) {
        set_error("out_channels = %d unsupported (1..4)", out_channels);
        return false;
    }
    if (dtype != EXASPIM_DT_F32 && dtype != EXASPIM_DT_BF16 && dtype != EXASPIM_DT_F16) {
        set_error("unknown compute dtype %d", dtype);
        return false;
    }
    UNetPlan p;
    for (int i = 0; i < 5; ++i) p.channels[i] = channels[i];
    p.out_channels = out_channels;
    p.dtype = dtype;
    p.convt = convt;
    const int* c = channels;
    // trilinear U-Net halves the bottleneck and the decoder widths (factor 2,
    // unet3d.py:53,68-74); the ConvTranspose3d variant keeps them (factor 1)
    const int half4 = convt ? c[4] : c[4] / 2;

    // (ca_real, cb_real, cout_real) of the 17 MFMA convs in state_dict order
    // (unet3d.py:64-74; Up: DoubleConv(in, out, mid=in//2) on cat[skip, up]).
    struct Spec { int ca, cb, co; };
    const Spec specs[kNumMfmaConvs] = {
        {c[0], 0, c[0]},                                  // inc.3
        {c[0], 0, c[1]}, {c[1], 0, c[1]},                 // down1
        {c[1], 0, c[2]}, {c[2], 0, c[2]},                 // down2
        {c[2], 0, c[3]}, {c[3], 0, c[3]},                 // down3
        {c[3], 0, half4}, {half4, 0, half4},              // down4
        // Up blocks on cat[skip, up(prev)]. trilinear: DoubleConv(in, out/2, mid=in/2);
        // conv-transpose: up halves the channels first, DoubleConv(in, out) (mid = out)
        {c[3], convt ? c[4] / 2 : half4, convt ? c[3] : c[4] / 2},
        {convt ? c[3] : c[4] / 2, 0, convt ? c[3] : c[3] / 2},                      // up1
        {c[2], c[3] / 2, convt ? c[2] : c[3] / 2}, {convt ? c[2] : c[3] / 2, 0, convt ? c[2] : c[2] / 2},  // up2
        {c[1], c[2] / 2, convt ? c[1] : c[2] / 2}, {convt ? c[1] : c[2] / 2, 0, convt ? c[1] : c[1] / 2},  // up3
        {c[0], c[1] / 2, convt ? c[0] : c[1] / 2}, {convt ? c[0] : c[1] / 2, 0, c[0]},                     // up4
    };

    size_t poff = 0, woff = 0;
    const int es = dtype_size(dtype);
    p.c0 = c[0];
    p.c0p = pad_channels(c[0]);
    p.first_p_off = poff;
    poff += conv_param_count(c[0], 1);
    p.first_w_off = woff;
    woff = align_up(woff + (size_t)27 * p.c0p * sizeof(float), 256);
    p.first_b_off = woff;
    woff = align_up(woff + (size_t)p.c0p * sizeof(float), 256);
    for (int i = 0; i < kNumMfmaConvs; ++i) {
        if (convt && i >= 9 && (i - 9) % 2 == 0) {
            // upN.up.weight (Cin, Cin/2, 2, 2, 2) + bias come before the block's DoubleConv
            ConvTLayer& U = p.up[(i - 9) / 2];
            U.cin_real = c[4 - (i - 9) / 2];
            U.cout_real = U.cin_real / 2;
            U.cin = pad_channels(U.cin_real);
            U.cout = pad_channels(U.cout_real);
            U.p_off = poff;
            poff += (size_t)U.cin_real * U.cout_real * 8 + U.cout_real;
            U.w_off = woff;
            woff = align_up(woff + (size_t)8 * U.cin * U.cout * es, 256);
            U.b_off = woff;
            woff = align_up(woff + (size_t)U.cout * sizeof(float), 256);
        }
        ConvLayer& L = p.conv[i];
        L.ca_real = specs[i].ca;
        L.cb_real = specs[i].cb;
        L.cout_real = specs[i].co;
        L.ca = pad_channels(L.ca_real);
        L.cb = L.cb_real ? pad_channels(L.cb_real) : 0;
        L.cout = pad_channels(L.cout_real);
        L.p_off = poff;
        poff += conv_param_count(L.cout_real, L.ca_real + L.cb_real);
        L.w_off = woff;
        woff = align_up(woff + (size_t)27 * (L.ca + L.cb) * L.cout * es, 256);
        L.b_off = woff;
        woff = align_up(woff + (size_t)L.cout * sizeof(float), 256);
#ifdef EXASPIM_VARIANTS   // (the measured-and-not-adopted kernels' own fragment orders: variant builds only)
        // level-1 layers with 64-cout slices (down1.0, down1.3, up3.0): K = 32 fragments for the
        // 16x16x32 kernel, [pair of chunks][tap 27][16-cout group][lane 64][8 x 16 bit]
        if (dtype != EXASPIM_DT_F32 && L.cout % 64 == 0 && (i == 1 || i == 2 || i == 13)) {
            L.w3_off = woff;
            woff = align_up(woff + (size_t)27 * (L.ca + L.cb) * L.cout * es, 256);
        }
        // 32-cout-slice layers of the 16-bit modes also get the paired-tap order
        if (dtype != EXASPIM_DT_F32 && L.cout % 64 != 0) {
            L.w2_off = woff;
            woff = align_up(woff + (size_t)kPairedFrags * 1024 * ((L.ca + L.cb) / 16) * (L.cout / 32), 256);
        }
#endif
    }
    p.head_p_off = poff;
    poff += (size_t)out_channels * c[0] + out_channels;
    p.head_w_off = woff;
    woff = align_up(woff + (size_t)out_channels * p.c0p * sizeof(float), 256);
    p.head_b_off = woff;
    woff = align_up(woff + (size_t)out_channels * sizeof(float), 256);
    p.n_params = poff;
    p.packed_bytes = woff;
    *plan = p;
    return true;
}

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline uint16_t f32_to_f16_rne(float f) {
    // saturating, like the kernels' stores: a folded weight beyond +-65504 stays finite
    if (f > 65504.f) f = 65504.f;
    if (f < -65504.f) f = -65504.f;
    _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

struct Folded {
    std::vector<double> scale, bias;
};

// params block layout: W(cout,cin,27) b(cout) gamma beta mean var
static Folded fold_bn(const float* blk, int cout, int cin) {
    const float* b = blk + (size_t)cout * cin * 27;
    const float* gamma = b + cout;
    const float* beta = gamma + cout;
    const float* mean = beta + cout;
    const float* var = mean + cout;
    Folded f;
    f.scale.resize(cout);
    f.bias.resize(cout);
    for (int o = 0; o < cout; ++o) {
        const double s = (double)gamma[o] / std::sqrt((double)var[o] + 1e-5);
        f.scale[o] = s;
        f.bias[o] = ((double)b[o] - (double)mean[o]) * s + (double)beta[o];
    }
    return f;
}

int pack_weights(const UNetPlan& plan, const float* params, void* packed_host) {
    char* out = static_cast<char*>(packed_host);
    std::memset(out, 0, plan.packed_bytes);

    {  // inc.0: float [27][c0p]
        const float* blk = params + plan.first_p_off;
        Folded f = fold_bn(blk, plan.c0, 1);
        float* w = reinterpret_cast<float*>(out + plan.first_w_off);
        float* b = reinterpret_cast<float*>(out + plan.first_b_off);
        for (int o = 0; o < plan.c0; ++o) {
            for (int t = 0; t < 27; ++t)
                w[(size_t)t * plan.c0p + o] = (float)((double)blk[(size_t)o * 27 + t] * f.scale[o]);
            b[o] = (float)f.bias[o];
        }
    }

    const int es = dtype_size(plan.dtype);
    const int G = 16 / es;   // elements per 16-byte fragment
    const int KC = 2 * G;    // input channels per 32-byte chunk
    for (int i = 0; i < kNumMfmaConvs; ++i) {
        const ConvLayer& L = plan.conv[i];
        const int cin_real = L.ca_real + L.cb_real;
        const float* blk = params + L.p_off;
        Folded f = fold_bn(blk, L.cout_real, cin_real);
        float* b = reinterpret_cast<float*>(out + L.b_off);
        for (int o = 0; o < L.cout_real; ++o) b[o] = (float)f.bias[o];
        char* wbase = out + L.w_off;
        const int nchunks = (L.ca + L.cb) / KC;
        const int ntiles = L.cout / 32;
        for (int c = 0; c < nchunks; ++c)
            for (int t = 0; t < 27; ++t)
                for (int n = 0; n < ntiles; ++n)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = 32 * n + (lane & 31);
                        const size_t frag = (((size_t)c * 27 + t) * ntiles + n) * 64 + lane;
                        for (int j = 0; j < G; ++j) {
                            const int pc = KC * c + G * (lane >> 5) + j;
                            int ci = -1;
                            if (pc < L.ca) {
                                if (pc < L.ca_real) ci = pc;
                            } else {
                                const int q = pc - L.ca;
                                if (q < L.cb_real) ci = L.ca_real + q;
                            }
                            float v = 0.f;
                            if (ci >= 0 && co < L.cout_real)
                                v = (float)((double)blk[((size_t)co * cin_real + ci) * 27 + t] *
                                            f.scale[co]);
                            char* dst = wbase + (frag * G + j) * es;
                            if (plan.dtype == EXASPIM_DT_F32) {
                                std::memcpy(dst, &v, 4);
                            } else {
                                const uint16_t hbits = plan.dtype == EXASPIM_DT_BF16
                                                           ? f32_to_bf16_rne(v)
                                                           : f32_to_f16_rne(v);
                                std::memcpy(dst, &hbits, 2);
                            }
                        }
                    }
    }

    // K = 32 fragments: lane l holds cout 16 ct + l % 16 and the 8 channels
    // 32 pair + 16 * (l / 32) + 8 * ((l / 16) % 2) .. + 7 (pair = two consecutive 16-channel chunks of
    // the concatenated, padded input: sources A and B are multiples of 32 channels, so a pair never
    // straddles them)
    for (int i = 0; i < kNumMfmaConvs; ++i) {
        const ConvLayer& L = plan.conv[i];
        if (!L.w3_off) continue;
        const int cin_real = L.ca_real + L.cb_real;
        const float* blk = params + L.p_off;
        Folded f = fold_bn(blk, L.cout_real, cin_real);
        const int npairs = (L.ca + L.cb) / 32, nct = L.cout / 16;
        for (int pr = 0; pr < npairs; ++pr)
            for (int t = 0; t < 27; ++t)
                for (int ct = 0; ct < nct; ++ct)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = 16 * ct + (lane & 15);
                        const size_t frag = (((size_t)pr * 27 + t) * nct + ct) * 64 + lane;
                        for (int j = 0; j < 8; ++j) {
                            const int pc = 32 * pr + 16 * (lane >> 5) + 8 * ((lane >> 4) & 1) + j;
                            int ci = -1;
                            if (pc < L.ca) {
                                if (pc < L.ca_real) ci = pc;
                            } else {
                                const int q = pc - L.ca;
                                if (q < L.cb_real) ci = L.ca_real + q;
                            }
                            float v = 0.f;
                            if (ci >= 0 && co < L.cout_real)
                                v = (float)((double)blk[((size_t)co * cin_real + ci) * 27 + t] * f.scale[co]);
                            const uint16_t hbits = plan.dtype == EXASPIM_DT_BF16 ? f32_to_bf16_rne(v)
                                                                                  : f32_to_f16_rne(v);
                            std::memcpy(out + L.w3_off + (frag * 8 + j) * 2, &hbits, 2);
                        }
                    }
    }

    // paired-tap fragments [chunk][frag 32][32-cout slice][lane 64][8 x 16 bit]: lane l holds
    // cout 16 t + l % 16 and channels 8 * ((l / 16) % 2) .. + 7 of tap (l / 32 ? tap1 : tap0)
    for (int i = 0; i < kNumMfmaConvs; ++i) {
        const ConvLayer& L = plan.conv[i];
        if (!L.w2_off) continue;
        const int cin_real = L.ca_real + L.cb_real;
        const float* blk = params + L.p_off;
        Folded f = fold_bn(blk, L.cout_real, cin_real);
        const int nchunks = (L.ca + L.cb) / 16, ntiles = L.cout / 32;
        for (int c = 0; c < nchunks; ++c)
            for (int fr = 0; fr < kPairedFrags; ++fr) {
                int taps[2];
                paired_frag_taps(fr, &taps[0], &taps[1]);
                const int t16 = fr % 2;
                for (int n = 0; n < ntiles; ++n)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = 32 * n + 16 * t16 + (lane & 15);
                        const int tap = taps[lane >> 5];
                        const size_t frag = (((size_t)c * kPairedFrags + fr) * ntiles + n) * 64 + lane;
                        for (int j = 0; j < 8; ++j) {
                            const int pc = 16 * c + 8 * ((lane >> 4) & 1) + j;
                            int ci = -1;
                            if (pc < L.ca) {
                                if (pc < L.ca_real) ci = pc;
                            } else {
                                const int q = pc - L.ca;
                                if (q < L.cb_real) ci = L.ca_real + q;
                            }
                            float v = 0.f;
                            if (tap >= 0 && ci >= 0 && co < L.cout_real)
                                v = (float)((double)blk[((size_t)co * cin_real + ci) * 27 + tap] * f.scale[co]);
                            const uint16_t hbits = plan.dtype == EXASPIM_DT_BF16 ? f32_to_bf16_rne(v)
                                                                                  : f32_to_f16_rne(v);
                            std::memcpy(out + L.w2_off + (frag * 8 + j) * 2, &hbits, 2);
                        }
                    }
            }
    }

    if (plan.convt) {
        // ConvTranspose3d(k=2, s=2): out[2z+dz, 2y+dy, 2x+dx, co] = b[co] + sum_ci in[z,y,x,ci] *
        // W[ci][co][dz][dy][dx] -- eight 1x1x1 GEMMs; fragments [chunk][phase][tile][lane]
        for (int u = 0; u < 4; ++u) {
            const ConvTLayer& U = plan.up[u];
            const float* w = params + U.p_off;
            const float* bsrc = w + (size_t)U.cin_real * U.cout_real * 8;
            float* b = reinterpret_cast<float*>(out + U.b_off);
            for (int o = 0; o < U.cout_real; ++o) b[o] = bsrc[o];
            const int nchunks = U.cin / KC, ntiles = U.cout / 32;
            for (int c = 0; c < nchunks; ++c)
                for (int ph = 0; ph < 8; ++ph)
                    for (int n = 0; n < ntiles; ++n)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = 32 * n + (lane & 31);
                            const size_t frag = (((size_t)c * 8 + ph) * ntiles + n) * 64 + lane;
                            for (int j = 0; j < G; ++j) {
                                const int ci = KC * c + G * (lane >> 5) + j;
                                float v = 0.f;
                                if (ci < U.cin_real && co < U.cout_real)
                                    v = w[((size_t)ci * U.cout_real + co) * 8 + ph];
                                char* dst = out + U.w_off + (frag * G + j) * es;
                                if (plan.dtype == EXASPIM_DT_F32) {
                                    std::memcpy(dst, &v, 4);
                                } else {
                                    const uint16_t hbits = plan.dtype == EXASPIM_DT_BF16
                                                               ? f32_to_bf16_rne(v)
                                                               : f32_to_f16_rne(v);
                                    std::memcpy(dst, &hbits, 2);
                                }
                            }
                        }
        }
    }

    {  // head: float [out_channels][c0p] + float[out_channels] (unet3d.py:318)
        const float* blk = params + plan.head_p_off;
        float* w = reinterpret_cast<float*>(out + plan.head_w_off);
        float* b = reinterpret_cast<float*>(out + plan.head_b_off);
        for (int o = 0; o < plan.out_channels; ++o) {
            for (int ci = 0; ci < plan.c0; ++ci)
                w[(size_t)o * plan.c0p + ci] = blk[(size_t)o * plan.c0 + ci];
            b[o] = blk[(size_t)plan.out_channels * plan.c0 + o];
        }
    }
    return EXASPIM_OK;
}

}  // namespace exaspim
