// Does a >64-bit MUBUF store with an SGPR soffset need a wait state before its data VGPRs are
// overwritten on gfx950? (LLVM's hazard recognizer says: only without a register soffset.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NOPS, int SOFF_REG>
__global__ void k(unsigned* out, int n_per_wave, unsigned soff_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
    for (int i = 0; i < n_per_wave; ++i) {
        const unsigned base = ((unsigned)wave * n_per_wave + i) * 64u + lane;   // 16-byte slot index
        u32x4 d = {base * 4 + 0, base * 4 + 1, base * 4 + 2, base * 4 + 3};
        unsigned voff = base * 16u + 4096u - (SOFF_REG ? soff_bytes : 0u);
        unsigned so = __builtin_amdgcn_readfirstlane(soff_bytes);
#define BODY(SOFF, NOP)                                                                                  \
        asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\ts_nop 4\n\t" \
                     "buffer_store_dwordx4 v[10:13], %4, %5, " SOFF " offen\n\t" NOP                            \
                     "v_mov_b32 v10, 0xdeadbeef\n\tv_mov_b32 v11, 0xdeadbeef\n\tv_mov_b32 v12, 0xdeadbeef\n\t"  \
                     "v_mov_b32 v13, 0xdeadbeef"                                                             \
                     :: "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w), "v"(voff), "s"(rs), "s"(so)                  \
                     : "v10", "v11", "v12", "v13", "memory")
        if (SOFF_REG) {
            if (NOPS == 0) BODY("%6", ""); else if (NOPS == 1) BODY("%6", "s_nop 0\n\t"); else BODY("%6", "s_nop 1\n\t");
        } else {
            if (NOPS == 0) BODY("0", ""); else if (NOPS == 1) BODY("0", "s_nop 0\n\t"); else BODY("0", "s_nop 1\n\t");
        }
    }
}
template <int NOPS, int SOFF_REG>
long run(unsigned* dev, size_t words, int blocks, int per) {
    hipMemset(dev, 0, words * 4);
    k<NOPS, SOFF_REG><<<blocks, 256>>>(dev, per, 4096);
    hipDeviceSynchronize();
    std::vector<unsigned> h(words);
    hipMemcpy(h.data(), dev, words * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < words; ++i) bad += i >= 1024 && h[i] != (unsigned)(i - 1024);
    return bad;
}
int main() {
    const int blocks = 2048, per = 64;
    const size_t words = (size_t)blocks * 4 * per * 64 * 4 + 1024;
    unsigned* dev;
    hipMalloc(&dev, words * 4);
    for (int rep = 0; rep < 2; ++rep) {
        printf("soffset SGPR: no nop %ld, s_nop 0 %ld, s_nop 1 %ld bad words of %zu\n", run<0, 1>(dev, words, blocks, per),
               run<1, 1>(dev, words, blocks, per), run<2, 1>(dev, words, blocks, per), words);
        printf("soffset 0   : no nop %ld, s_nop 0 %ld, s_nop 1 %ld bad words\n", run<0, 0>(dev, words, blocks, per),
               run<1, 0>(dev, words, blocks, per), run<2, 0>(dev, words, blocks, per));
    }
    return 0;
}
