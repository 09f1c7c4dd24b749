// bf16_pipe.hpp -- what the one-wave-per-SIMD bf16 field kernels (field_bf16w.hip, field_bf16x3.hip) share: the vector types, the
// hand-issued LDS reads with their counted waits, the bias-as-C-operand helpers and the double-buffered LDS-DMA weight-chunk context.
//
// Hand-issued reads and the compiler.  An A fragment or bias row is read by `asm volatile ds_read_b128` several MFMA steps ahead of
// its use and retired by a counted `s_waitcnt lgkmcnt(n)` (LDS operations of a wave complete in issue order).  The compiler knows
// nothing of that: to it the read's destination is defined the moment the asm is issued.  Two rules make the scheme safe:
//   1. the wait that retires a read takes the destination as a read-write operand (wait_retire): the value every later instruction sees
//      is "defined" by the wait, so no use can be scheduled in front of it, and the register is live from the read to the wait, so it
//      cannot be handed to another value in between -- also when NO instruction uses the data (a read issued only to keep the counts
//      uniform), which is how the first 16x16x32 port overwrote a DMA address;
//   2. tools/check_lds_inflight.py checks the compiled ISA of every such kernel -- no instruction between a ds_read_b128 and the wait that
//      retires it (found by counting LDS operations against each wait's N) may read or write its destination -- and build.py runs it on
//      every build, so an object that violates it is never linked.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "sahs_common.hpp"

namespace SAHS_NS {
namespace bfp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ uint32_t lds_addr_of(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p; }
__device__ __forceinline__ bf16x8 frag(const u32x4 &v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }

template <int OFF, class V>      // V: u32x4 (A fragments) or f32x4 (bias rows)
__device__ __forceinline__ void lds_read16(V &dst, uint32_t addr)
{
    static_assert(sizeof(V) == 16, "one ds_read_b128");
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// s_waitcnt lgkmcnt(N) that RETIRES the given destinations (rule 1 above): one, two (hi + lo fragment), or those plus a bias batch
template <int N, class V>
__device__ __forceinline__ void wait_retire(V &a)
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N, class V>
__device__ __forceinline__ void wait_retire(V &a, V &b)
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N, class V>
__device__ __forceinline__ void wait_retire(V &a, f32x4 (&t)[4])
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "n"(N));
}
template <int N, class V>
__device__ __forceinline__ void wait_retire(V &a, V &b, f32x4 (&t)[4])
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "n"(N));
}

// this lane's 16 bias rows of a 32-row tile (accumulator register r <-> row (r&3) + 8(r>>2) + 4h): four ds_read_b128 at +32 B steps.
// The raw destinations become the MFMA's C operand only AFTER the wait that retires them (bias_as_c).
template <int OFF>
__device__ __forceinline__ void bias_read(f32x4 (&t)[4], uint32_t addr)
{
    lds_read16<OFF>(t[0], addr);
    lds_read16<OFF + 32>(t[1], addr);
    lds_read16<OFF + 64>(t[2], addr);
    lds_read16<OFF + 96>(t[3], addr);
}
// (float-typed reads, plain element copies: __builtin_bit_cast applied to an ELEMENT of an ext_vector reads element 0 for every index
// with this compiler -- found on the ISA; whole-vector bit_casts are fine)
__device__ __forceinline__ f32x16 bias_as_c(const f32x4 (&t)[4])
{
    f32x16 b;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) b[4 * g + r] = t[g][r];
    return b;
}

// The weight stream of a level as <= 64 KB chunks through two LDS buffers: while the MFMAs of a chunk run from one buffer the next
// chunk is fetched into the other by LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging), one 4 KB piece per call.
// NODMA / NOBARRIER: timing-only ablations (tools/ablate.py), results wrong by construction.
template <int THREADS, int BUF_BYTES, int BIAS_BYTE_OFF, bool NODMA = false, bool NOBARRIER = false>
struct PipeCtx {
    const unsigned short *stream;   // this level's packed stream
    char *lds;
    int buf;
    int lane, h, wave;
    const f32x4 *nx_src; f32x4 *nx_dst;   // the chunk being prefetched: this lane's first source granule, the LDS buffer
    uint32_t off;                   // halfword offset of the NEXT chunk to prefetch (uniform)
    uint32_t wrap_at, wrap_to;      // the stream wraps for the next sample tile: whole network [0, STREAM_HW), deformation nets
                                    // [0, T0), radiance nets [T0, STREAM_HW) (the kernel's MODE)
    uint32_t bias_addr;             // LDS byte address of this lane's first bias row (+4h rows)

    __device__ __forceinline__ void prepare(int hw, int b)
    {
        if (off >= wrap_at) off = wrap_to;
        nx_src = reinterpret_cast<const f32x4 *>(stream + off) + lane;
        nx_dst = reinterpret_cast<f32x4 *>(lds + b * BUF_BYTES);
        off += (uint32_t)hw;
    }
    __device__ __forceinline__ void issue_piece(int p)
    {
        // (A hand-issued scalar-base form -- global_load_lds_dwordx4 voffset, s[base:base+1], no VALU address arithmetic -- was measured
        // and is not faster: 22.6 against 22.0 ms.)
        if constexpr (!NODMA) {
            const int base = p * THREADS + wave * WAVE;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(nx_src + base), (lds_ptr_t)(nx_dst + base), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void begin_chunk(int next_hw) { prepare(next_hw, buf ^ 1); }
    __device__ __forceinline__ void end_chunk()
    {
        if constexpr (NOBARRIER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __syncthreads();       // vmcnt(0) (the next chunk has landed) + barrier (every wave is done with this one)
        buf ^= 1;
    }
    __device__ __forceinline__ uint32_t cur_addr() const { return lds_addr_of(lds + buf * BUF_BYTES) + 16 * lane; }
    // re-materialised opaquely once per sample tile: keeps the ~140 per-tile bias addresses out of LICM (they would be hoisted and spilled)
    __device__ __forceinline__ void refresh_bias_base()
    {
        uint32_t a = lds_addr_of(lds) + BIAS_BYTE_OFF + 16 * h;
        asm volatile("" : "+v"(a));
        bias_addr = a;
    }
};

}  // namespace bfp
}  // namespace SAHS_NS
