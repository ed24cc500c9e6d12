// Checks the semantics the conv kernel relies on for `buffer_load_dwordx4 ... offen lds` on gfx950:
// lane-linear LDS destination at M0, soffset added to the per-lane voffset, out-of-range reads return zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bufload_lds16(u32x4 srd, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}
__global__ void k(const char* src, char* dst, int nrec, int soff, int lds_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    u32x4 srd;
    const uint64_t p = (uint64_t)src;
    srd[0] = __builtin_amdgcn_readfirstlane((uint32_t)p);
    srd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xffffu);
    srd[2] = __builtin_amdgcn_readfirstlane((uint32_t)nrec);
    srd[3] = 0x00020000u;
    // lanes read a permuted 16-B unit (reverse order inside the wave) to prove the destination is lane-linear
    bufload_lds16(srd, (uint32_t)((wave * 64 + (63 - lane)) * 16), (uint32_t)soff, (uint32_t)(lds_base + wave * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096 / 16; i += blockDim.x)
        reinterpret_cast<uint4*>(dst)[i] = *reinterpret_cast<const uint4*>(smem + lds_base + i * 16);
}
int main() {
    const int N = 8192;
    std::vector<uint8_t> h(N); for (int i = 0; i < N; ++i) h[i] = (uint8_t)(i * 7 + 3);
    char *src, *dst; hipMalloc(&src, N); hipMalloc(&dst, 4096); hipMemcpy(src, h.data(), N, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int lds_base : {0, 70000, 150000}) for (int soff : {0, 256}) for (int nrec : {8192, 3000}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, src, dst, nrec, soff, lds_base);
        std::vector<uint8_t> o(4096); hipMemcpy(o.data(), dst, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) {
            const int off = (w * 64 + (63 - l)) * 16 + soff + e;           // byte offset read by lane l of wave w
            const uint8_t want = off < nrec ? h[off] : 0;
            if (o[w * 1024 + l * 16 + e] != want) ++bad;
        }
        printf("lds_base %6d soffset %3d num_records %4d: %s (%d bad bytes)\n", lds_base, soff, nrec, bad ? "MISMATCH" : "ok", bad);
        bad_total += bad;
    }
    printf("RESULT %s\n", bad_total ? "FAIL" : "PASS");
    return bad_total != 0;
}
