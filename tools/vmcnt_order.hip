// Do range-dropped MUBUF stores retire in order with older loads on gfx950? A counted wait
// (s_waitcnt vmcnt(#younger stores)) behind range-checked stores is only safe if they do.
//   hipcc -O2 --offload-arch=gfx950 tools/vmcnt_order.hip -o tools/vmcnt_order
// Per lane: a load that misses every cache, then four stores (all dropped by the range check, or all
// real), then s_waitcnt vmcnt(4), then the load's register is copied; a copy that still holds the
// sentinel means the wait let the load through unfinished.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int DROPPED>
__global__ void k(const unsigned* big, unsigned nbig, unsigned* sink, unsigned* out, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(big), 0, (int)(nbig * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(sink, 0, DROPPED ? 0 : (int)(gridDim.x * blockDim.x * 16), 0x00020000);
    unsigned bad = 0;
    unsigned state = tid * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        state = state * 1664525u + 1013904223u;
        const unsigned idx = (state >> 4) % nbig;
        unsigned loff = idx * 4, soff = tid * 16, x, y, zero = 0;
        asm volatile(
            "v_mov_b32 %0, 0xdeadbeef\n\t"
            "s_nop 4\n\t"
            "buffer_load_dword %0, %2, %4, 0 offen\n\t"
            "buffer_store_dword %6, %3, %5, 0 offen\n\t"
            "buffer_store_dword %6, %3, %5, 0 offen offset:4\n\t"
            "buffer_store_dword %6, %3, %5, 0 offen offset:8\n\t"
            "buffer_store_dword %6, %3, %5, 0 offen offset:12\n\t"
            "s_waitcnt vmcnt(4)\n\t"
            "v_mov_b32 %1, %0\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(x), "=&v"(y) : "v"(loff), "v"(soff), "s"(lrs), "s"(srs), "v"(zero) : "memory");
        bad += (y == 0xdeadbeefu) || (y != idx);
    }
    out[tid] = bad;
}
int main() {
    const unsigned nbig = 1u << 28;     // 1 GiB of words: every load misses
    unsigned *big, *sink, *out;
    hipMalloc(&big, (size_t)nbig * 4);
    std::vector<unsigned> h(1 << 20);
    const int blocks = 4096, threads = 256, iters = 200;
    hipMalloc(&sink, (size_t)blocks * threads * 16);
    hipMalloc(&out, (size_t)blocks * threads * 4);
    // big[i] = i
    {
        std::vector<unsigned> chunk(1 << 24);
        for (unsigned base = 0; base < nbig; base += (1u << 24)) {
            for (unsigned i = 0; i < (1u << 24); ++i) chunk[i] = base + i;
            hipMemcpy(big + base, chunk.data(), chunk.size() * 4, hipMemcpyHostToDevice);
        }
    }
    std::vector<unsigned> res((size_t)blocks * threads);
    for (int rep = 0; rep < 2; ++rep) {
        for (int dropped = 0; dropped < 2; ++dropped) {
            if (dropped) k<1><<<blocks, threads>>>(big, nbig, sink, out, iters);
            else k<0><<<blocks, threads>>>(big, nbig, sink, out, iters);
            hipDeviceSynchronize();
            hipMemcpy(res.data(), out, res.size() * 4, hipMemcpyDeviceToHost);
            unsigned long long bad = 0;
            for (unsigned v : res) bad += v;
            printf("%s stores behind the load: %llu of %llu loads seen unfinished after s_waitcnt vmcnt(4)\n",
                   dropped ? "range-dropped" : "real         ", bad, (unsigned long long)res.size() * iters);
        }
    }
    return 0;
}
