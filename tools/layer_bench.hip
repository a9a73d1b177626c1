// Diagnostic (not part of the library): times the non-MFMA layer kernels of the network
// (trilinear upsampling, inc.0, max-pool) on the shapes of a batch of 16 patches of 96^3.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/layer_bench.hip -ffp-contract=off \
//         aind_exaspim_neuron_segmentation_amd/csrc/plan.cpp -x hip -o tools/layer_bench
#include "../aind_exaspim_neuron_segmentation_amd/csrc/layers.hip"

#include <cmath>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    const int n = 16;
    const int dt = EXASPIM_DT_F16;
    const size_t big = (size_t)n * 96 * 96 * 96 * 32 * 2;   // level-0 tensor, 32 channels
    void *a, *b;
    float *x, *xp, *w, *bias;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big));
    std::vector<uint16_t> h(big / 2);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = 0x3800 + (uint16_t)(s & 0x3ff); }
    CK(hipMemcpy(a, h.data(), big, hipMemcpyHostToDevice));
    CK(hipMalloc(&x, (size_t)n * 96 * 96 * 96 * 4));
    CK(hipMemset(x, 0, (size_t)n * 96 * 96 * 96 * 4));
    CK(hipMalloc(&xp, (size_t)n * 98 * 98 * 98 * 4));
    CK(hipMalloc(&w, 27 * 32 * 4)); CK(hipMemset(w, 0, 27 * 32 * 4));
    CK(hipMalloc(&bias, 32 * 4)); CK(hipMemset(bias, 0, 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double bytes, auto&& fn) {
        for (int i = 0; i < 3; ++i) fn();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) fn();
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us   %6.2f TB/s (algorithmic bytes)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) * 1e-12);
    };
    using namespace exaspim;
    const bool plain = getenv("EXASPIM_PLAIN_UPSAMPLE") != nullptr;   // (tool only: the un-pipelined upsampling kernel)
    if (argc > 1) {   // check: an output voxel must not depend on the margin (role inside its pair)
        const int c = 8, e = 8, oe = 16;                       // one fp32 chunk plane, 8^3 -> 16^3
        std::vector<float> hs((size_t)e * e * e * c), o0((size_t)oe * oe * oe * c), o1(o0.size());
        for (auto& v : hs) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) / 9007199254740992.0 - 0.5); }
        CK(hipMemcpy(a, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        void* b2; CK(hipMalloc(&b2, o0.size() * 4));
        CK(hipMemset(b, 0, o0.size() * 4)); CK(hipMemset(b2, 0, o0.size() * 4));
        launch_upsample2(EXASPIM_DT_F32, a, b, 1, e, e, e, c, 0, 0);
        launch_upsample2(EXASPIM_DT_F32, a, b2, 1, e, e, e, c, 1, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o0.data(), b, o0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o1.data(), b2, o1.size() * 4, hipMemcpyDeviceToHost));
        {   // which run deviates from the float64 evaluation of torch's formula?
            auto coord = [&](int o, int& i0, int& i1, double& l) {
                const float sc = (float)(e - 1) / (float)(oe - 1);
                const float sf = sc * (float)o;
                i0 = (int)sf; if (i0 > e - 1) i0 = e - 1; i1 = i0 + 1 < e ? i0 + 1 : e - 1;
                l = (double)(sf - (float)i0);
            };
            double e0 = 0, e1 = 0;
            for (int z = 1; z < oe - 1; ++z) for (int y = 1; y < oe - 1; ++y) for (int x = 1; x < oe - 1; ++x) {
                int z0, z1, y0, y1, x0, x1; double lz, ly, lx;
                coord(z, z0, z1, lz); coord(y, y0, y1, ly); coord(x, x0, x1, lx);
                for (int k = 0; k < c; ++k) {
                    auto S = [&](int zz, int yy, int xx) { return (double)hs[(((size_t)zz * e + yy) * e + xx) * c + k]; };
                    auto P = [&](int zz) {
                        const double r0 = (1 - lx) * S(zz, y0, x0) + lx * S(zz, y0, x1);
                        const double r1 = (1 - lx) * S(zz, y1, x0) + lx * S(zz, y1, x1);
                        return (1 - ly) * r0 + ly * r1; };
                    const double ref = (1 - lz) * P(z0) + lz * P(z1);
                    const size_t i = (((size_t)z * oe + y) * oe + x) * c + k;
                    e0 = fmax(e0, fabs(o0[i] - ref)); e1 = fmax(e1, fabs(o1[i] - ref));
                }
            }
            printf("max |gpu - float64 reference|: margin 0 run %.3g, margin 1 run %.3g\n", e0, e1);
            // float32 emulation of the kernel's nesting, lambda fused (fma(scale, o, -floor)) or plain
            for (int fused = 0; fused < 2; ++fused) {
                auto coordf = [&](int o, int& i0, int& i1, float& l) {
                    const float sc = (float)(e - 1) / (float)(oe - 1);
                    const float sf = sc * (float)o;
                    i0 = (int)floorf(sf); if (i0 > e - 1) i0 = e - 1; i1 = i0 + 1 < e ? i0 + 1 : e - 1;
                    l = fused ? fmaf(sc, (float)o, -(float)i0) : sf - (float)i0;
                    l = fminf(fmaxf(l, 0.f), 1.f);
                };
                long m0 = 0, m1 = 0, tot = 0;
                for (int z = 1; z < oe - 1; ++z) for (int y = 1; y < oe - 1; ++y) for (int x = 1; x < oe - 1; ++x) {
                    int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
                    coordf(z, z0, z1, lz); coordf(y, y0, y1, ly); coordf(x, x0, x1, lx);
                    for (int k = 0; k < c; ++k) {
                        auto S = [&](int zz, int yy, int xx) { return hs[(((size_t)zz * e + yy) * e + xx) * c + k]; };
                        auto P = [&](int zz) {
                            const float r0 = fmaf(lx, S(zz, y0, x1), (1.f - lx) * S(zz, y0, x0));
                            const float r1 = fmaf(lx, S(zz, y1, x1), (1.f - lx) * S(zz, y1, x0));
                            return fmaf(ly, r1, (1.f - ly) * r0); };
                        const float ref = fmaf(lz, P(z1), (1.f - lz) * P(z0));
                        const size_t i = (((size_t)z * oe + y) * oe + x) * c + k;
                        m0 += o0[i] == ref; m1 += o1[i] == ref; ++tot;
                    }
                }
                printf("float32 emulation (lambda %s): margin 0 run matches %ld, margin 1 run %ld of %ld\n",
                       fused ? "fused" : "plain", m0, m1, tot);
            }
        }
        int bad = 0;
        for (int z = 1; z < oe - 1; ++z) for (int y = 1; y < oe - 1; ++y) for (int x = 1; x < oe - 1; ++x)
            for (int k = 0; k < c; ++k) {
                const size_t i = (((size_t)z * oe + y) * oe + x) * c + k;
                if (o0[i] != o1[i] && bad++ < 12) printf("z %d y %d x %d ch %d: %.9g vs %.9g\n", z, y, x, k, o0[i], o1[i]);
            }
        printf("margin 0 vs margin 1: %d values differ\n", bad);
        return 0;
    }
    // up4.up: 32 ch, 48^3 -> 96^3, margin 6 (trimmed forward) and 0
    time("upsample 32ch 48->96 margin 6", 16.0 * 64 * (84.0 * 84 * 84 + 48 * 48 * 48),
         [&] { launch_upsample2(dt, a, b, n, 48, 48, 48, 32, 6, 0, plain); });
    time("upsample 32ch 48->96 margin 6, per-thread pipeline", 16.0 * 64 * (84.0 * 84 * 84 + 48 * 48 * 48),
         [&] { launch_upsample2(dt, a, b, n, 48, 48, 48, 32, 6, 0, plain, true); });
    time("upsample 32ch 48->96 margin 0", 16.0 * 64 * (96.0 * 96 * 96 + 48 * 48 * 48),
         [&] { launch_upsample2(dt, a, b, n, 48, 48, 48, 32, 0, 0, plain); });
    time("upsample 64ch 24->48", 16.0 * 128 * (48.0 * 48 * 48 + 24 * 24 * 24),
         [&] { launch_upsample2(dt, a, b, n, 24, 24, 24, 64, 0, 0, plain); });
    time("upsample 128ch 12->24", 16.0 * 256 * (24.0 * 24 * 24 + 12 * 12 * 12),
         [&] { launch_upsample2(dt, a, b, n, 12, 12, 12, 128, 0, 0, plain); });
    time("upsample 256ch 6->12", 16.0 * 512 * (12.0 * 12 * 12 + 6 * 6 * 6),
         [&] { launch_upsample2(dt, a, b, n, 6, 6, 6, 256, 0, 0, plain); });
    time("maxpool 64ch 48->24", 16.0 * 128 * (48.0 * 48 * 48 + 24 * 24 * 24),
         [&] { launch_maxpool2(dt, a, b, n, 48, 48, 48, 64, 0); });
    time("inc.0 (pad + conv_first16) 96^3 -> 32ch", 16.0 * 96 * 96 * 96 * (4 + 64),
         [&] { launch_conv_first(dt, x, xp, w, bias, b, n, 96, 96, 96, 32, 0.01f, 0, getenv("EXASPIM_NO_STRIPS") != nullptr); });
    return 0;
}
