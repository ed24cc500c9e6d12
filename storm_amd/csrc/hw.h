// Hardware primitives of the gfx950 kernels - the ONLY file of libstorm_hip that knows about anything but the device:
//   * section DEVICE: what the kernels run on (inline asm, amdgcn builtins);
//   * section HOST: the same names for (a) the host pass of a hipcc compilation, where kernel bodies are parsed but never
//     run, and (b) the CPU test build (-DSTORM_HOST_SIM, tests/sim/), where they are functional models on top of the fiber
//     runtime of tests/sim/hip_host_shim.h (test infrastructure: the asynchronous LDS-DMA queue model lives there).
// Kernel sources call these functions and carry no preprocessor branches of their own.  Included by common.h.
#pragma once

namespace storm {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t BUF_OOB = 0x80000000u;     // a per-lane offset past every buffer used here: the load returns zeros

#if defined(__HIP_DEVICE_COMPILE__)
// ===================================================== DEVICE ===========================================================
// two fp32 -> packed bf16x2 (lo in bits 0-15), round-to-nearest-even: one v_cvt_pk_bf16_f32.  (The compiler's own conversion,
// not inline asm: behind an asm statement the hazard recognizer cannot see the consumer, and a v_cvt_pk_bf16_f32 issued straight
// after the v_dot2c_f32_bf16 that produced its operand read the STALE register on gfx950.)
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, v2bf_));
}
// raw hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp): used where the result is rounded to 16 bits anyway
__device__ inline float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ inline float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// acc + x.lo * w.lo + x.hi * w.hi on a dword of two 16-bit values (v_dot2c_f32_bf16 / v_dot2c_f32_f16): a filter tap on packed
// 16-bit data without unpacking - w = (weight, 0) adds the low channel's tap, (0, weight) the high channel's.  With one half
// of w zero and a weight of few mantissa bits the product is exact in fp32: the same value as fmaf(weight, x, acc).
__device__ __forceinline__ float dot2_acc(uint32_t x, uint32_t w, float acc, bf16_t*) {
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, x), __builtin_bit_cast(v2bf, w), acc, false);
}
__device__ __forceinline__ float dot2_acc(uint32_t x, uint32_t w, float acc, half_t*) {
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h, x), __builtin_bit_cast(v2h, w), acc, false);
}
// Wave-level ordering point: the LDS executes one wave's requests in issue order, so data a wave wrote is visible to its own
// later reads without a workgroup barrier; this only stops compiler reordering.
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// wave vote: true in every lane if the predicate holds in any lane
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// ---- opaque values: what the optimiser must not look through ------------------------------------------------------------
// keep a wave-uniform value in an SGPR (stops re-materialisation from the kernarg segment inside a loop)
__device__ __forceinline__ int pin(int x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ unsigned long long pin(unsigned long long x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// a per-lane value made opaque at this point: address math derived from it is not hoisted out of the enclosing loop and kept
// live in registers (one hoisted table entry, spilled, put a scratch reload + vmcnt(0) into conv_pipe's pipelined loop)
template <typename V> __device__ __forceinline__ void launder(V& v) { asm volatile("" : "+v"(v)); }
// a value that must exist in a VGPR here (work-skipping profiling instantiations keep their operands alive with it; a packed
// 16-bit literal operand of v_dot2c must come from a register: as a 32-bit literal only its low half is honoured)
template <typename V> __device__ __forceinline__ void keep(const V& v) { asm volatile("" ::"v"(v)); }
template <typename V> __device__ __forceinline__ void keep_rw(V& v) { asm volatile("" : "+v"(v)); }
// a * b + c for a, b < 2^24 (pixel indices, per-pixel byte strides): ONE 32-bit instruction (v_mad_u32_u24).  A plain 32-bit
// product goes through v_mad_u64_u32 and a 64-bit register pair.
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }
// A kernel's first argument (a parameter block passed by value) read in place through the kernarg segment: the constant
// address space keeps every field access a scalar load and nothing of the block is copied to registers or scratch.
template <typename P> struct KArg { typedef const P __attribute__((address_space(4)))* Ptr; };
template <typename P> __device__ __forceinline__ typename KArg<P>::Ptr kernarg_of(const P&) { return (typename KArg<P>::Ptr)__builtin_amdgcn_kernarg_segment_ptr(); }
template <typename Q> __device__ __forceinline__ void relaunder(Q& p) { asm volatile("" : "+s"(p)); }     // (re-opaque: its scalar loads stay inside the loop)
// a read-only table in device memory viewed the same way (grouped launches: one parameter block per problem): scalar loads, no copies
template <typename P> __device__ __forceinline__ typename KArg<P>::Ptr const_table(const P* p) { return (typename KArg<P>::Ptr)(unsigned long long)p; }
// a global pointer whose origin the compiler cannot see (keeps the stores global_store instead of flat_store)
__device__ __forceinline__ char* as_global(unsigned long long u) { asm volatile("" : "+s"(u)); typedef __attribute__((address_space(1))) char G; return (char*)(G*)u; }
__device__ __forceinline__ unsigned long long hw_memtime() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long hw_ids() {      // (XCC id, HW id) of this wave, for the wave-timeline tool
    return (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
}

// ---- buffer loads --------------------------------------------------------------------------------------------------------
// Raw buffer resource (stride 0): a 16-byte load whose offset lies past `bytes` returns zeros.  Padding pixels, ragged channel
// counts and rows past a matrix become an out-of-range OFFSET instead of a branch around the load.
struct BufRsrc { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ BufRsrc make_buf(const void* base, uint32_t bytes) {
    BufRsrc b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    return b;
}
__device__ __forceinline__ uint4 buf_load16(const BufRsrc& b, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, (int)soff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
// the same resource as four SGPRs, for the inline-asm LDS-DMA below
__device__ __forceinline__ u32x4 make_srd(const void* base, uint32_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)p);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
// ---- asynchronous copy: 16 B per lane, buffer (srd) at voff + soff  ->  LDS at (uniform lds_wave + 16 * lane) -------------
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t voff, uint32_t soff, char* lds_wave, int lane) {
    (void)lane;
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave);
    // M0 is written and consumed inside this one statement (the compiler keeps nothing live in M0 in these kernels)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(srd), "s"(soff), "s"(la) : "memory");
}
// counted wait: at most N of this wave's vector-memory instructions still in flight (they retire in issue order)
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier WITHOUT a vmcnt drain: LDS traffic of this wave retired (lgkmcnt), loads keep flying
__device__ __forceinline__ void raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// issue priority of this wave (raised around MFMA clusters: the partner wave on the SIMD is staging then)
__device__ __forceinline__ void prio(int p) { if (p) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }

#else
// ====================================================== HOST =============================================================
// (the host pass of a device build never runs these; the test simulator does)
__device__ inline uint32_t pack_bf16x2(float lo, float hi) { return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16); }
__device__ inline float hw_exp2(float x) { return exp2f(x); }
__device__ inline float hw_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float dot2_acc(uint32_t x, uint32_t w, float acc, bf16_t*) {
    return fmaf(bf16_bits_to_f32((uint16_t)(x & 0xffffu)), bf16_bits_to_f32((uint16_t)(w & 0xffffu)),
                fmaf(bf16_bits_to_f32((uint16_t)(x >> 16)), bf16_bits_to_f32((uint16_t)(w >> 16)), acc));
}
__device__ __forceinline__ float dot2_acc(uint32_t x, uint32_t w, float acc, half_t*) {
    return fmaf(f16_bits_to_f32((uint16_t)(x & 0xffffu)), f16_bits_to_f32((uint16_t)(w & 0xffffu)),
                fmaf(f16_bits_to_f32((uint16_t)(x >> 16)), f16_bits_to_f32((uint16_t)(w >> 16)), acc));
}
__device__ __forceinline__ int pin(int x) { return x; }
__device__ __forceinline__ unsigned long long pin(unsigned long long x) { return x; }
__device__ __forceinline__ int uniform(int x) { return x; }
template <typename V> __device__ __forceinline__ void launder(V&) {}
template <typename V> __device__ __forceinline__ void keep(const V&) {}
template <typename V> __device__ __forceinline__ void keep_rw(V&) {}
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return a * b + c; }
template <typename P> struct KArg { typedef const P* Ptr; };
template <typename P> __device__ __forceinline__ typename KArg<P>::Ptr kernarg_of(const P& a) { return &a; }
template <typename Q> __device__ __forceinline__ void relaunder(Q&) {}
template <typename P> __device__ __forceinline__ typename KArg<P>::Ptr const_table(const P* p) { return p; }
__device__ __forceinline__ char* as_global(unsigned long long u) { return reinterpret_cast<char*>(u); }
__device__ __forceinline__ unsigned long long hw_memtime() { return 0ull; }
__device__ __forceinline__ unsigned long long hw_ids() { return 0ull; }
struct BufRsrc { const char* base; uint32_t bytes; };
__device__ __forceinline__ BufRsrc make_buf(const void* base, uint32_t bytes) { return BufRsrc{static_cast<const char*>(base), bytes}; }
__device__ __forceinline__ uint4 buf_load16(const BufRsrc& b, uint32_t voff, uint32_t soff) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((uint64_t)voff + soff + 16 <= b.bytes) memcpy(&v, b.base + voff + soff, 16);
    return v;
}
__device__ __forceinline__ u32x4 make_srd(const void* base, uint32_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    u32x4 r;
    r[0] = (uint32_t)p; r[1] = (uint32_t)(p >> 32); r[2] = bytes; r[3] = 0;
    return r;
}
#if defined(STORM_HOST_SIM)
__device__ inline void wave_sync() { simrt::wave_rendezvous(); }
__device__ inline bool wave_any(bool p) {
    int in = p ? 1 : 0, out = 0;
    simrt::wave_collective(&in, &out, sizeof(out), &simrt::any_fn, 0);
    return out != 0;
}
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t voff, uint32_t soff, char* lds_wave, int lane) {
    const uint64_t off = (uint64_t)voff + soff;
    const char* base = reinterpret_cast<const char*>(((uint64_t)srd[1] << 32) | srd[0]);
    simdma::Entry e;
    e.dst = lds_wave + 16 * lane;
    if (off + 16 <= srd[2]) memcpy(e.data, base + off, 16); else memset(e.data, 0, 16);
    if (simdma::late()) simdma::queue().push_back(e);        // lands at the latest legal moment (see simdma)
    else memcpy(e.dst, e.data, 16);                          // lands at once
}
template <int N> __device__ __forceinline__ void vm_wait() {
    simdma::retire(N);
    simrt::wave_rendezvous();          // simulator lanes are not in lockstep: every lane's copy is done past this point
}
#else
__device__ inline void wave_sync() {}
__device__ inline bool wave_any(bool p) { return p; }
__device__ __forceinline__ void dma16(u32x4, uint32_t, uint32_t, char*, int) {}
template <int N> __device__ __forceinline__ void vm_wait() {}
#endif
__device__ __forceinline__ void raw_barrier() { __syncthreads(); }
__device__ __forceinline__ void prio(int) {}
#endif

}  // namespace storm
