// Diagnostic (not part of the library): runs one 3x3x3 MFMA convolution of the
// network on synthetic data with EXASPIM_TRACE phase stamps and dumps them, so
// tools/analyze_trace.py can rebuild a per-CU timeline of where a workgroup's
// life goes (prologue latency, MFMA loop, barriers, epilogue).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DEXASPIM_TRACE tools/conv_trace.hip \
//         aind_exaspim_neuron_segmentation_amd/csrc/plan.cpp aind_exaspim_neuron_segmentation_amd/csrc/layers.hip -x hip -o tools/conv_trace
//   tools/conv_trace <ca> <cb> <cout> <edge> <batch> <out.bin> [kernel variant]
#include "../aind_exaspim_neuron_segmentation_amd/csrc/conv3d.hip"

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
    if (argc < 7) return 2;
    if (argc > 7) exaspim::g_variant = atoi(argv[7]);
    const int ca = atoi(argv[1]), cb = atoi(argv[2]), cout = atoi(argv[3]);
    const int edge = atoi(argv[4]), n = atoi(argv[5]);
    const size_t vox = (size_t)n * edge * edge * edge;
    const int cin = ca + cb, nchunks = cin / 16, ntiles = cout / 32;
    const size_t wbytes = (size_t)nchunks * 27 * ntiles * 1024;
    size_t helems = (size_t)vox * (ca > cb ? ca : cb);
    if (helems < wbytes / 2) helems = wbytes / 2;   // the same host buffer seeds the weights
    std::vector<uint16_t> h(helems);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = 0x3c00 + (uint16_t)(s & 0x1ff);  // bf16 around 0.008..0.03
    }
    void *a_dev, *b_dev = nullptr, *w_dev, *dst;
    float* bias;
    CK(hipMalloc(&a_dev, vox * ca * 2));
    CK(hipMemcpy(a_dev, h.data(), vox * ca * 2, hipMemcpyHostToDevice));
    if (cb) {
        CK(hipMalloc(&b_dev, vox * cb * 2));
        CK(hipMemcpy(b_dev, h.data(), vox * cb * 2, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&w_dev, wbytes));
    CK(hipMemcpy(w_dev, h.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&bias, cout * 4));
    CK(hipMemset(bias, 0, cout * 4));
    CK(hipMalloc(&dst, vox * cout * 2));
    const size_t nwg = (size_t)n * ((edge + 3) / 4) * ((edge + 3) / 4) * ((edge + 11) / 12) + 64;   // tiles (upper bound over the tile shapes)
    unsigned long long* trace;
    const size_t tbytes = nwg * 4 * 16 * 8;
    CK(hipMalloc(&trace, tbytes));
    CK(hipMemset(trace, 0, tbytes));

    float* partial;
    const size_t pbytes = (size_t)4 * vox * cout * 4;
    CK(hipMalloc(&partial, pbytes));
    exaspim::ConvArgs a{};
    a.src_a = a_dev; a.src_b = b_dev; a.ca = ca; a.cb = cb; a.weights = w_dev; a.bias = bias;
    a.partial = partial; a.partial_patch_bytes = pbytes / n;
    void* w2_dev = nullptr;   // paired-tap fragments (timing only: random bits)
    if (cout % 64 != 0) {
        const size_t w2bytes = (size_t)nchunks * 32 * ntiles * 1024;
        CK(hipMalloc(&w2_dev, w2bytes));
        CK(hipMemcpy(w2_dev, h.data(), w2bytes < h.size() * 2 ? w2bytes : h.size() * 2, hipMemcpyHostToDevice));
        a.weights_paired = w2_dev;
    }
    void* w3_dev = nullptr;   // K = 32 fragments for conv3x3x3_t16 (timing only: random bits)
    if (cout % 64 == 0 && cin % 32 == 0) {
        const size_t w3bytes = (size_t)27 * cin * cout * 2;
        CK(hipMalloc(&w3_dev, w3bytes));
        CK(hipMemcpy(w3_dev, h.data(), w3bytes < h.size() * 2 ? w3bytes : h.size() * 2, hipMemcpyHostToDevice));
        a.weights_k32 = w3_dev;
    }
    a.dst = dst; a.cout = cout; a.n = n; a.d = a.h = a.w = edge; a.slope = 0.01f;
    if (exaspim::g_variant != 0) {   // the variant must reproduce the library kernel bit for bit
        const int v = exaspim::g_variant;
        void* ref;
        CK(hipMalloc(&ref, vox * cout * 2));
        exaspim::g_variant = 0;
        a.dst = ref;
        if (exaspim::launch_conv3x3x3(EXASPIM_DT_BF16, a, 0)) return 1;
        exaspim::g_variant = v;
        a.dst = dst;
        CK(hipMemset(dst, 0xff, vox * cout * 2));
        if (exaspim::launch_conv3x3x3(EXASPIM_DT_BF16, a, 0)) return 1;
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> h0(vox * cout), h1(vox * cout);
        CK(hipMemcpy(h0.data(), ref, vox * cout * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), dst, vox * cout * 2, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
        printf("variant %d vs library kernel: %zu of %zu values differ\n", v, bad, h0.size());
        CK(hipFree(ref));
        if (bad > h0.size() / 1000) return 3;   // different kernels round differently in a few places
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 4; ++it) {   // warm-up, untraced
        if (exaspim::launch_conv3x3x3(EXASPIM_DT_BF16, a, 0)) { fprintf(stderr, "%s\n", exaspim::get_error()); return 1; }
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = getenv("CONV_TRACE_REPEAT") ? atoi(getenv("CONV_TRACE_REPEAT")) : 10;
    for (int it = 0; it < reps; ++it) exaspim::launch_conv3x3x3(EXASPIM_DT_BF16, a, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 27 * cin * cout * (double)vox;
    printf("untraced: %.3f ms per launch, %.1f TFLOP/s\n", ms / reps, flop / (ms / reps * 1e-3) * 1e-12);
    a.trace = trace;
    CK(hipEventRecord(e0, 0));
    exaspim::launch_conv3x3x3(EXASPIM_DT_BF16, a, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("traced: %.3f ms, %zu workgroups\n", ms, nwg);
    std::vector<unsigned long long> t(nwg * 64);
    CK(hipMemcpy(t.data(), trace, tbytes, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[6], "wb");
    if (!f) return 1;
    fwrite(t.data(), 1, tbytes, f);
    fclose(f);
    return 0;
}
