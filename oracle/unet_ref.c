/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into or called by the product.
 *
 * Plain-C restatement of the tensor operators the reference's U-Net is made of
 * (AllenNeuralDynamics/aind-exaspim-neuron-segmentation, paths relative to
 * src/aind_exaspim_neuron_segmentation/machine_learning/unet3d.py). The
 * reference delegates these to torch; this file spells their published
 * definitions out as loops (float32 data, float64 accumulation, NCDHW) so the
 * torch-based oracle in reference_path.py can be pinned without torch:
 * tests/test_oracle_c.py compares the two on small cases.
 *
 *   conv3d_k3p1      nn.Conv3d(kernel_size=3, padding=1)            unet3d.py:143,146
 *   conv3d_k1        nn.Conv3d(kernel_size=1)                       unet3d.py:318
 *   batchnorm_eval   nn.BatchNorm3d in eval mode, eps = 1e-5        unet3d.py:144,147
 *   leaky_relu       nn.LeakyReLU(negative_slope)                   unet3d.py:145,148
 *   maxpool2         nn.MaxPool3d(2)                                unet3d.py:195
 *   upsample2        nn.Upsample(scale_factor=2, mode="trilinear",
 *                                align_corners=True)                unet3d.py:248-250
 *   sigmoid          torch.sigmoid                                  inference.py:158
 *
 * Build: make -C oracle   (gcc -O2 -shared; output oracle/_build/libunet_ref.so)
 */
#include <math.h>
#include <stddef.h>

#define IDX5(n, c, z, y, x, C, D, H, W) \
    (((((size_t)(n) * (C) + (c)) * (D) + (z)) * (H) + (y)) * (W) + (x))

/* out[n,co,z,y,x] = b[co] + sum_{ci,kz,ky,kx} w[co,ci,kz,ky,kx] * in[n,ci,z+kz-1,y+ky-1,x+kx-1]
 * with zeros outside the volume (cross-correlation, as torch). */
void conv3d_k3p1(const float* in, const float* w, const float* b, float* out, int N, int Cin,
                 int Cout, int D, int H, int W) {
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int z = 0; z < D; ++z)
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        double acc = b ? (double)b[co] : 0.0;
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int kz = 0; kz < 3; ++kz) {
                                const int zz = z + kz - 1;
                                if (zz < 0 || zz >= D) continue;
                                for (int ky = 0; ky < 3; ++ky) {
                                    const int yy = y + ky - 1;
                                    if (yy < 0 || yy >= H) continue;
                                    for (int kx = 0; kx < 3; ++kx) {
                                        const int xx = x + kx - 1;
                                        if (xx < 0 || xx >= W) continue;
                                        acc += (double)w[((((size_t)co * Cin + ci) * 3 + kz) * 3 + ky) * 3 + kx] *
                                               (double)in[IDX5(n, ci, zz, yy, xx, Cin, D, H, W)];
                                    }
                                }
                            }
                        out[IDX5(n, co, z, y, x, Cout, D, H, W)] = (float)acc;
                    }
}

void conv3d_k1(const float* in, const float* w, const float* b, float* out, int N, int Cin,
               int Cout, int D, int H, int W) {
    const size_t vol = (size_t)D * H * W;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (size_t v = 0; v < vol; ++v) {
                double acc = b ? (double)b[co] : 0.0;
                for (int ci = 0; ci < Cin; ++ci)
                    acc += (double)w[(size_t)co * Cin + ci] * (double)in[((size_t)n * Cin + ci) * vol + v];
                out[((size_t)n * Cout + co) * vol + v] = (float)acc;
            }
}

/* y = (x - mean) / sqrt(var + eps) * gamma + beta, per channel, in place */
void batchnorm_eval(float* x, const float* gamma, const float* beta, const float* mean,
                    const float* var, double eps, int N, int C, size_t vol) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const double s = (double)gamma[c] / sqrt((double)var[c] + eps);
            float* p = x + ((size_t)n * C + c) * vol;
            for (size_t v = 0; v < vol; ++v)
                p[v] = (float)(((double)p[v] - (double)mean[c]) * s + (double)beta[c]);
        }
}

void leaky_relu(float* x, double slope, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (x[i] < 0.f) x[i] = (float)((double)x[i] * slope);
}

void sigmoid(float* x, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] = (float)(1.0 / (1.0 + exp(-(double)x[i])));
}

/* kernel 2, stride 2, floor mode: out dims D/2, H/2, W/2 */
void maxpool2(const float* in, float* out, int N, int C, int D, int H, int W) {
    const int OD = D / 2, OH = H / 2, OW = W / 2;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int z = 0; z < OD; ++z)
                for (int y = 0; y < OH; ++y)
                    for (int x = 0; x < OW; ++x) {
                        float m = -INFINITY;
                        for (int dz = 0; dz < 2; ++dz)
                            for (int dy = 0; dy < 2; ++dy)
                                for (int dx = 0; dx < 2; ++dx) {
                                    const float v = in[IDX5(n, c, 2 * z + dz, 2 * y + dy, 2 * x + dx, C, D, H, W)];
                                    if (v > m) m = v;
                                }
                        out[IDX5(n, c, z, y, x, C, OD, OH, OW)] = m;
                    }
}

/* trilinear, scale 2, align_corners=True: src = i * (n - 1) / (2n - 1),
 * i0 = floor(src), i1 = min(i0 + 1, n - 1), lambda = src - i0 (separable). */
static void lerp_axis(int o, int n, int* i0, int* i1, double* l) {
    const int on = 2 * n;
    const double src = on > 1 ? (double)o * (double)(n - 1) / (double)(on - 1) : 0.0;
    int a = (int)floor(src);
    if (a > n - 1) a = n - 1;
    *i0 = a;
    *i1 = a + 1 < n ? a + 1 : n - 1;
    *l = src - (double)a;
}

void upsample2(const float* in, float* out, int N, int C, int D, int H, int W) {
    const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int z = 0; z < OD; ++z) {
                int z0, z1; double lz; lerp_axis(z, D, &z0, &z1, &lz);
                for (int y = 0; y < OH; ++y) {
                    int y0, y1; double ly; lerp_axis(y, H, &y0, &y1, &ly);
                    for (int x = 0; x < OW; ++x) {
                        int x0, x1; double lx; lerp_axis(x, W, &x0, &x1, &lx);
                        double acc = 0.0;
                        const int zs[2] = {z0, z1}, ys[2] = {y0, y1}, xs[2] = {x0, x1};
                        const double wz[2] = {1.0 - lz, lz}, wy[2] = {1.0 - ly, ly}, wx[2] = {1.0 - lx, lx};
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b)
                                for (int d = 0; d < 2; ++d)
                                    acc += wz[a] * wy[b] * wx[d] *
                                           (double)in[IDX5(n, c, zs[a], ys[b], xs[d], C, D, H, W)];
                        out[IDX5(n, c, z, y, x, C, OD, OH, OW)] = (float)acc;
                    }
                }
            }
}
