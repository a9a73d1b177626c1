// Measurement aid (not part of the library): what does the device sustain for the STORE patterns of
// the two store-bound kernels, with nothing else in the kernel?
//   hipcc -O3 --offload-arch=gfx950 tools/store_ceiling.hip -o tools/store_ceiling && tools/store_ceiling
//  a) one contiguous 906 MB range (inc.0's output of a batch of 16), 16 B per lane, wave = 1 KiB run
//  b) the level-0 upsampling's output: per volume (16 patches x 2 chunk planes of 96^3 x 32 B) the
//     84^3 box at margin 6: rows of 2 688 B every 3 072 B; one wave-store = 1 KiB of a row
//  c) b) issued like upsample2_pipe_kernel: a thread owns a row pair at one (x, group) and walks the
//     planes two at a time -- four stores to 2 rows x 2 planes per iteration
//  d) c) with all the stores of a wave's run issued by plane-major blocks of 256 threads = 1.5 row pairs
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_contiguous(uint4* dst, size_t n16) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

// one thread per 16-byte piece of the box, pieces enumerated row-major inside the box
__global__ void fill_box(uint4* dst, int vols, int edge, int m) {
    const int n = edge - 2 * m;
    const size_t per_vol = (size_t)n * n * n * 2;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_vol * vols; i += (size_t)gridDim.x * blockDim.x) {
        const size_t vol = i / per_vol, r = i % per_vol;
        const int xg = (int)(r % (2 * n)), y = (int)((r / (2 * n)) % n), z = (int)(r / ((size_t)2 * n * n));
        dst[((vol * edge + z + m) * edge + y + m) * edge * 2 + 2 * m + xg] = v;
    }
}

// the upsampling kernel's issue order: thread = (row pair, x, group), a run of 12 plane pairs, 4 stores per pair
__global__ void fill_like_upsample(uint4* dst, int edge, int m) {
    const int n = edge - 2 * m, nyp = n / 2, nzp = n / 2, run_len = 12, nruns = (nzp + run_len - 1) / run_len;
    const unsigned item = blockIdx.y * blockDim.x + threadIdx.x;
    if (item >= (unsigned)(nyp * n * 2)) return;
    const int yp = (int)(item >> 1) / n, xg = (int)item - yp * n * 2;
    const int vol = blockIdx.x / nruns, run = blockIdx.x % nruns;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    const int pend = min(nzp, (run + 1) * run_len);
    for (int pr = run * run_len; pr < pend; ++pr)
#pragma unroll
        for (int zz = 0; zz < 2; ++zz)
#pragma unroll
            for (int yy = 0; yy < 2; ++yy)
                dst[(((size_t)vol * edge + m + 2 * pr + zz) * edge + m + 2 * yp + yy) * edge * 2 + 2 * m + xg] = v;
}

int main() {
    const int edge = 96, m = 6, vols = 32;
    const size_t bytes = (size_t)vols * edge * edge * edge * 32;
    uint4* dst;
    CK(hipMalloc(&dst, bytes));
    CK(hipMemset(dst, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = edge - 2 * m;
    const double box = (double)vols * n * n * n * 32;
    auto time = [&](const char* name, double b, auto&& launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %8.1f us  %5.2f TB/s\n", name, ms / 20 * 1e3, b / (ms / 20 * 1e-3) * 1e-12);
    };
    for (int wgs : {1024, 2048, 4096, 16384})
        time(("a) contiguous 906 MB, grid " + std::to_string(wgs)).c_str(), (double)bytes,
             [&] { fill_contiguous<<<wgs, 256>>>(dst, bytes / 16); });
    time("a') contiguous, first 607 MB", box, [&] { fill_contiguous<<<4096, 256>>>(dst, (size_t)(box / 16)); });
    for (int wgs : {2048, 4096, 16384})
        time(("b) 84^3 boxes row-major, grid " + std::to_string(wgs)).c_str(), box, [&] { fill_box<<<wgs, 256>>>(dst, vols, edge, m); });
    {
        const int items = (n / 2) * n * 2, nruns = (n / 2 + 11) / 12;
        dim3 grid(vols * nruns, (items + 255) / 256);
        time("c) 84^3 boxes in upsample2_pipe_kernel's order", box, [&] { fill_like_upsample<<<grid, 256>>>(dst, edge, m); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
