// field_bf16.hip -- the per-sample field with bf16 MFMA operands and fp32 accumulation
// (BASELINE.json configs[2]).  Same structure as field_f32.hip, re-tiled for
// v_mfma_f32_32x32x16_bf16:
//  * one wave owns 32 samples (MFMA columns); lane = 32*h + j holds sample j.  A 32x32 fp32
//    accumulator tile (lane: rows (r&3) + 8(r>>2) + 4h, r = 0..15) becomes the next layer's B operand
//    by converting registers 8s..8s+7 to a bf16x8 fragment for k-step s -- no lane movement, no LDS
//    (the weights are packed in the matching k order, sahs_layout.hpp).  Activations are rounded to
//    bf16 once per layer; accumulation, biases, positional encodings, the grid interpolation and the
//    final [rgb|seg|sigma] tile are fp32.
//  * 8 waves x 32 samples = 256 samples per workgroup tile share each weight chunk (<= 64 KB,
//    LDS-DMA double buffered): 1 ds_read_b128 per MFMA per wave = 128 B/clk/CU (half the LDS rate).
//  * skip layers are not split here: a tile's accumulator runs over [hidden | re-injected encoding].
// Roofline: MFMA-bound against the dense bf16 peak (~2.5 PFLOP/s).
#include <hip/hip_runtime.h>
#include <utility>
#include "../sahs_common.hpp"
#include "../sahs_layout.hpp"

namespace sahs {
namespace hb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Blk { bf16x8 s[2]; };   // 32 features of this lane's sample, as the two k-step B fragments

#ifndef SAHS_BF16_WAVES
#define SAHS_BF16_WAVES 8
#endif
#ifndef SAHS_BF16_APF
#define SAHS_BF16_APF 0       // 0: A-fragment reads left to the compiler; N > 0: inline-asm ds_read_b128, N fragments ahead, counted lgkmcnt
#endif
constexpr int H_THREADS = 64 * SAHS_BF16_WAVES;
constexpr int H_PTS_PER_WAVE = 32;
constexpr int H_PTS_PER_WG = (H_THREADS / WAVE) * H_PTS_PER_WAVE;   // 256
constexpr int LDS_BUF_BYTES = CHUNK_HW_MAX * 2;                       // 64 KB each, two of them
constexpr int LDS_BIAS_BYTE_OFF = 2 * LDS_BUF_BYTES;
constexpr int LDS_STASH_BYTE_OFF = LDS_BIAS_BYTE_OFF + ((BIAS_FLOATS + 3) / 4) * 16;
constexpr int STASH_FLOATS = 8;                                       // per sample: x'[3], w[2] (+pad): cold across the trunk
constexpr int LDS_BYTES = LDS_STASH_BYTE_OFF + H_PTS_PER_WG * STASH_FLOATS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int DBG_STRIDE_H = 56;

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

struct CtxH {
    const unsigned short *stream;   // this level's packed bf16 stream
    const uint32_t *table;          // chunk start offsets (halfwords)
    char *lds;
    int chunk, buf;
    int lane, h, wave;
#ifdef SAHS_STAMP
    // diagnostic build only (tools/stamp_bf16.py): s_memtime stamps of waves 0 and 4 of workgroup 0 for one sample tile, written to
    // the dbg buffer (which the normal dbg writes then leave alone); no output value is computed from them
    unsigned long long *stamps; int sidx; bool stamp_on;
    __device__ __forceinline__ void stamp()
    {
        if (stamp_on) { if (lane == 0) stamps[sidx] = __builtin_amdgcn_s_memtime(); ++sidx; }
    }
#else
    __device__ __forceinline__ void stamp() {}
#endif

    // LDS-DMA of chunk c into buffer b, cut into pieces of 8 KB (one 1-KB global_load_lds per wave).
    // A piece costs the issuing wave ~100+ cycles of issue time, so the pieces of the NEXT chunk are spread
    // between the MFMAs of the current one (issue_piece) instead of being issued back to back.
    // The chunk sequence is static, so offsets are tracked arithmetically (no table loads on the critical path):
    // off = halfword offset of the chunk being prefetched; its size is passed by the layer code.
    const f32x4 *nx_src; f32x4 *nx_dst; int nx_n16;
    uint32_t off;          // halfword offset of the NEXT chunk to prefetch (uniform)
    __device__ __forceinline__ void prepare(int hw, int b)
    {
        if (off >= (uint32_t)STREAM_HW) off = 0;                   // the stream wraps for the next sample tile
        nx_n16 = hw >> 3;                                          // 16-byte units
        nx_src = reinterpret_cast<const f32x4 *>(stream + off) + lane;
        nx_dst = reinterpret_cast<f32x4 *>(lds + b * LDS_BUF_BYTES);
        off += (uint32_t)hw;
    }
    // piece p of the next chunk; the caller knows at compile time how many pieces that chunk has (no branch here, so a
    // chunk's MFMA run stays one basic block and the scheduler may interleave across it)
    __device__ __forceinline__ void issue_piece(int p)
    {
#ifndef SAHS_ABLATE_NODMA
        const int base = p * H_THREADS + wave * WAVE;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(nx_src + base), (lds_ptr_t)(nx_dst + base), 16, 0, 0);
#endif
    }
    __device__ __forceinline__ void begin_chunk(int next_hw) { prepare(next_hw, buf ^ 1); }
    __device__ __forceinline__ void end_chunk()
    {
#ifndef SAHS_ABLATE_NOBARRIER
        __syncthreads();
#endif
        chunk = (chunk + 1 == NUM_CHUNKS_H) ? 0 : chunk + 1;
        buf ^= 1;
    }
    __device__ __forceinline__ const bf16x8 *cur() const { return reinterpret_cast<const bf16x8 *>(lds + buf * LDS_BUF_BYTES); }
    // accumulator register r of a 32-row tile <-> row (r&3) + 8(r>>2) + 4h: this lane's bias rows start at +4h, stride 8
    // bias_lane = LDS address of this lane's first bias row; it is re-materialised (opaquely) once per sample tile so that
    // the per-tile addresses are NOT hoisted out of the persistent loop (LICM would precompute ~140 of them and spill them;
    // a spill reload inside a layer waits on vmcnt and thereby on the LDS-DMA prefetch).  All reads are base + immediate.
    typedef const __attribute__((address_space(3))) float *lds_cfloat;   // stays an LDS pointer: a generic one would become flat_load (vmcnt!)
    lds_cfloat bias_lane;
    __device__ __forceinline__ void refresh_bias_base()
    {
        uint32_t a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)lds) + LDS_BIAS_BYTE_OFF + 16 * h;
        asm volatile("" : "+v"(a));
        bias_lane = (lds_cfloat)(uintptr_t)a;
    }
    __device__ __forceinline__ lds_cfloat bias_ptr(int off) const { return bias_lane + off; }
    __device__ __forceinline__ f32x16 bias16(int off) const
    {
        lds_cfloat b = bias_lane + off;
        f32x16 v;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(b + 8 * g);
            v[4 * g + 0] = t[0]; v[4 * g + 1] = t[1]; v[4 * g + 2] = t[2]; v[4 * g + 3] = t[3];
        }
        return v;
    }
    // 16-row layers (WF, HF, FINAL) occupy rows 0..15 of a 32-row tile: rows 16..31 (regs 8..15) carry no bias
    __device__ __forceinline__ f32x16 bias16_half(int off) const
    {
        lds_cfloat b = bias_lane + off;
        f32x16 v;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const f32x4 t = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(b + 8 * g);
            v[4 * g + 0] = t[0]; v[4 * g + 1] = t[1]; v[4 * g + 2] = t[2]; v[4 * g + 3] = t[3];
        }
#pragma unroll
        for (int r = 8; r < 16; ++r) v[r] = 0.0f;
        return v;
    }
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// bias, activation (slope in [0,1]: leaky relu == max(x, slope*x); 0 = relu, 1 = none) and fp32 -> bf16 pairs (v_cvt_pk_bf16_f32).
// Written on float pairs so that the adds and the slope products become v_pk_add_f32 / v_pk_mul_f32 (two values per VALU issue):
// the repack is the largest non-MFMA item of a wave's instruction stream (tools/stamp_bf16.py).
__device__ __forceinline__ Blk pack_act(const f32x16 acc, const __attribute__((address_space(3))) float *bias, float slope)   // bias: this lane's rows, LDS
{
    Blk o;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint32_t w[4];
        typedef const __attribute__((address_space(3))) f32x4 *lds_cf4;
        const f32x4 b0 = *reinterpret_cast<lds_cf4>(bias + 16 * s), b1 = *reinterpret_cast<lds_cf4>(bias + 16 * s + 8);
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const f32x2 bb = (jp < 2) ? f32x2{b0[2 * jp], b0[2 * jp + 1]} : f32x2{b1[2 * jp - 4], b1[2 * jp - 3]};
            const f32x2 v = f32x2{acc[8 * s + 2 * jp], acc[8 * s + 2 * jp + 1]} + bb;
#ifdef SAHS_ABLATE_NOACT
            const f32x2 a = v;
#else
            f32x2 a;
            if (slope == 1.0f) {
                a = v;
            } else if (slope == 0.0f) {
                a = f32x2{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
            } else {
                const f32x2 u = v * f32x2{slope, slope};
                a = f32x2{fmaxf(v[0], u[0]), fmaxf(v[1], u[1])};
            }
#endif
            w[jp] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2));
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        o.s[s] = __builtin_bit_cast(bf16x8, (u32x4){w[0], w[1], w[2], w[3]});
    }
    return o;
}

constexpr int DMA_PIECES = LDS_BUF_BYTES / (H_THREADS * 16);   // 8 pieces of 8 KB cover the largest chunk
constexpr int A_PREFETCH = SAHS_BF16_APF > 0 ? SAHS_BF16_APF : 3;   // A fragments (ds_read_b128 each) kept in flight ahead of the MFMA that consumes them

// ---- hand-issued A-fragment reads ---------------------------------------------------------------------------------------------
// Left to the compiler, the reads of the "ring" below are sunk next to their uses (ds_read, ds_read, s_waitcnt lgkmcnt(0), MFMA ...:
// the full LDS latency is exposed every two or three MFMAs; tools/check_isa.py).  Issued as volatile asm they stay where they are
// written -- A_PREFETCH fragments ahead of the MFMA that consumes them -- and each MFMA waits only for ITS fragment with a counted
// lgkmcnt (LDS returns in order; the persistent loop has no scalar loads that could share the counter out of order).  The
// destination registers are unprotected until that wait (cdna_hip_programming.md section 5.7): nothing but the MFMA reads them.
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p; }
template <int OFF>       // OFF: byte offset, an instruction immediate (< 64 KB: one chunk buffer)
__device__ __forceinline__ void a_read(bf16x8 &dst, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
    __builtin_amdgcn_sched_barrier(0);      // the MFMA that consumes a LATER fragment must not be scheduled above this read's wait
}
template <int N>
__device__ __forceinline__ void wait_lgkm()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);      // hipcc would otherwise hoist a register-only MFMA above the wait (guide rule 18)
}
__device__ __forceinline__ void wait_lgkm_n(int n)      // n is a constant after unrolling: the switch folds to one s_waitcnt
{
    switch (n) {
    case 0: wait_lgkm<0>(); break;
    case 1: wait_lgkm<1>(); break;
    case 2: wait_lgkm<2>(); break;
    case 3: wait_lgkm<3>(); break;
    case 4: wait_lgkm<4>(); break;
    case 5: wait_lgkm<5>(); break;
    case 6: wait_lgkm<6>(); break;
    default: wait_lgkm<7>(); break;
    }
}

// hidden layer: NT32 output tiles, activation, bf16 repack.  A chunk is one flat run of G*KB*2 MFMAs;
// the A-fragment reads run A_PREFETCH steps ahead, across tile boundaries.
// NEXT_HW: size (halfwords) of the chunk that follows this layer's last one (compile-time: the program is static)
template <int K0, int K1, int K2, int NT32, int NEXT_HW>
__device__ __forceinline__ void dense_h(CtxH &cx, const Blk *in0, const Blk *in1, const Blk *in2, Blk *out, int bias_off, float slope)
{
    constexpr int KB = K0 + K1 + K2;
    constexpr int G = pick_G32(KB, NT32);
    constexpr int STEPS = KB * 2, TOTAL = G * STEPS;
    // one DMA piece of the next chunk every PSTEP MFMAs, all within the first half of this chunk so that the last one has
    // half a chunk of MFMA time to land before the vmcnt(0) + barrier that ends the chunk
    constexpr int PSTEP = TOTAL >= 2 * DMA_PIECES ? TOTAL / (2 * DMA_PIECES) : 1;
#pragma unroll
    for (int c = 0; c < NT32 / G; ++c) {
        const int nhw = (c + 1 < NT32 / G) ? G * KB * 1024 : NEXT_HW;          // constant after unrolling
        const int npieces = (nhw + H_THREADS * 8 - 1) / (H_THREADS * 8);        // 8 KB = 4096 halfwords per piece
        cx.begin_chunk(nhw);
        cx.stamp();                                   // chunk start
        const bf16x8 *A = cx.cur() + cx.lane;
        bf16x8 a[A_PREFETCH];
#if SAHS_BF16_APF
        const uint32_t abase = lds_addr_of(A);
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (a_read<Is * 1024>(a[Is], abase), ...); }(std::make_integer_sequence<int, A_PREFETCH>{});
#else
#pragma unroll
        for (int i = 0; i < A_PREFETCH; ++i) a[i] = A[i * 64];
#endif
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;      // bias is added at repack time: the chain starts from an inline zero
#if SAHS_BF16_APF
        auto mfma_step = [&]<int I>() {       // I is a compile-time constant: the read offsets are instruction immediates
            constexpr int k = I % STEPS, b = k >> 1, st = k & 1;
            const Blk &x = (b < K0) ? in0[b] : ((b < K0 + K1) ? in1[b - K0] : in2[b - K0 - K1]);
            wait_lgkm<((TOTAL - 1 - I) < (A_PREFETCH - 1) ? (TOTAL - 1 - I) : (A_PREFETCH - 1))>();    // fragment I has landed
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[I % A_PREFETCH], x.s[st], acc, 0, 0, 0);
            if constexpr (I + A_PREFETCH < TOTAL) a_read<(I + A_PREFETCH) * 1024>(a[I % A_PREFETCH], abase);
        };
        auto full_step = [&]<int I>() {
            constexpr int g = I / STEPS, k = I % STEPS;
            if (I % PSTEP == 0 && I / PSTEP < npieces) cx.issue_piece(I / PSTEP);
            mfma_step.template operator()<I>();
            if constexpr (k == STEPS - 1) {
                Blk o = pack_act(acc, cx.bias_ptr(bias_off + 32 * (c * G + g)), slope);
                asm volatile("" : "+v"(o.s[0]), "+v"(o.s[1]));
                out[c * G + g] = o;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            }
        };
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (full_step.template operator()<Is>(), ...); }(std::make_integer_sequence<int, TOTAL>{});
#else
#pragma unroll
        for (int i = 0; i < TOTAL; ++i) {
            const int g = i / STEPS, k = i % STEPS, b = k >> 1, st = k & 1;
            const Blk &x = (b < K0) ? in0[b] : ((b < K0 + K1) ? in1[b - K0] : in2[b - K0 - K1]);
            if (i % PSTEP == 0 && i / PSTEP < npieces) cx.issue_piece(i / PSTEP);
#ifdef SAHS_ABLATE_NOMFMA
            if (i % 16 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % A_PREFETCH], x.s[st], acc, 0, 0, 0);
#else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % A_PREFETCH], x.s[st], acc, 0, 0, 0);
#endif
#ifdef SAHS_ABLATE_NOLDSREAD
            if (i + A_PREFETCH < TOTAL && (i % 16 == 0)) a[i % A_PREFETCH] = A[(i + A_PREFETCH) * 64];
#else
            if (i + A_PREFETCH < TOTAL) a[i % A_PREFETCH] = A[(i + A_PREFETCH) * 64];
#endif
            if (k == STEPS - 1) {
#ifdef SAHS_ABLATE_NOPACK
                asm volatile("" :: "v"(acc));
                Blk o = in0[0];
#else
                Blk o = pack_act(acc, cx.bias_ptr(bias_off + 32 * (c * G + g)), slope);
#endif
                // pin the repack HERE: otherwise LLVM sinks it to the next layer's first use and the 16-register fp32
                // accumulator of every finished tile stays live instead of its 8-register bf16 form (=> spills)
                asm volatile("" : "+v"(o.s[0]), "+v"(o.s[1]));
                out[c * G + g] = o;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            }
        }
#endif
#pragma unroll
        for (int pc = (TOTAL + PSTEP - 1) / PSTEP; pc < DMA_PIECES; ++pc)
            if (pc < npieces) cx.issue_piece(pc);
        cx.stamp();                                   // work done, before the barrier
        cx.end_chunk();
        cx.stamp();                                   // after the barrier
    }
}

template <int K0, int NPIECES>
__device__ __forceinline__ f32x16 tile_mac(CtxH &cx, const bf16x8 *A, const Blk *in0, f32x16 acc)
{
    bf16x8 a[A_PREFETCH];
    constexpr int TOTAL = K0 * 2;
#if SAHS_BF16_APF
    const uint32_t abase = lds_addr_of(A);
    constexpr int PRE = A_PREFETCH < TOTAL ? A_PREFETCH : TOTAL;
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (a_read<Is * 1024>(a[Is], abase), ...); }(std::make_integer_sequence<int, PRE>{});
    auto step = [&]<int I>() {
        if (I < NPIECES) cx.issue_piece(I);
        wait_lgkm<((TOTAL - 1 - I) < (A_PREFETCH - 1) ? (TOTAL - 1 - I) : (A_PREFETCH - 1))>();
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[I % A_PREFETCH], in0[I >> 1].s[I & 1], acc, 0, 0, 0);
        if constexpr (I + A_PREFETCH < TOTAL) a_read<(I + A_PREFETCH) * 1024>(a[I % A_PREFETCH], abase);
    };
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (step.template operator()<Is>(), ...); }(std::make_integer_sequence<int, TOTAL>{});
#else
#pragma unroll
    for (int i = 0; i < A_PREFETCH; ++i) a[i] = A[i * 64];
#pragma unroll
    for (int i = 0; i < K0 * 2; ++i) {
        if (i < NPIECES) cx.issue_piece(i);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % A_PREFETCH], in0[i >> 1].s[i & 1], acc, 0, 0, 0);
        if (i + A_PREFETCH < K0 * 2) a[i % A_PREFETCH] = A[(i + A_PREFETCH) * 64];
    }
#endif
#pragma unroll
    for (int pc = K0 * 2; pc < NPIECES; ++pc) cx.issue_piece(pc);
    return acc;
}

// 16-row output layer accumulated in fp32: init = bias (first) or the running tile (ALPHA -> RGB -> SEG)
template <int K0, int NEXT_HW>
__device__ __forceinline__ void dense_h_out(CtxH &cx, const Blk *in0, f32x16 &acc, int bias_off, bool first)
{
    cx.begin_chunk(NEXT_HW);
    cx.stamp();
    const bf16x8 *A = cx.cur() + cx.lane;
    if (first) acc = cx.bias16_half(bias_off);
    acc = tile_mac<K0, (NEXT_HW + H_THREADS * 8 - 1) / (H_THREADS * 8)>(cx, A, in0, acc);
    cx.stamp();
    cx.end_chunk();
    cx.stamp();
}

// ---- positional encoding for the bf16 path ------------------------------------------------------
// Features are rounded to bf16 (8 significant bits) right after, so the hardware sine is ample:
// v_sin_f32 takes its argument in revolutions, sin(2^k x) = v_sin(fract(2^k * x/2pi)), cos = sin(. + 1/4).
// The power-of-two scaling is exact; the error is that of x/2pi (6e-8 relative, i.e. <= 2e-5 rad at the
// top octave for |x| < 1) plus the instruction's ~1e-6.  The fp32 kernel keeps ocml's sincosf.
struct PeSlot { float scale; float phase; int axis; int kind; };   // kind: 0 zero pad, 1 raw input, 2 sinusoid
template <int D, int L>
constexpr PeSlot pe_slot(int f)
{
    constexpr int W = D + 2 * D * L;
    if (f >= W) return PeSlot{0.0f, 0.0f, 0, 0};
    if (f < D) return PeSlot{1.0f, 0.0f, f, 1};
    const int g = f - D, k = g / (2 * D), rem = g % (2 * D);
    return PeSlot{(float)(1 << k), (rem / D) ? 0.25f : 0.0f, rem % D, 2};
}

template <int D, int L, int NB>
__device__ __forceinline__ void pe_blocks_h(const float *v, int h, Blk *out)
{
    float rev[3];
#pragma unroll
    for (int i = 0; i < D; ++i) rev[i] = v[i] * 0.15915494309189535f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f0 = 32 * b + 16 * s + 8 * (j >> 2) + (j & 3);
                constexpr PeSlot Z{0.0f, 0.0f, 0, 0};
                const PeSlot a = pe_slot<D, L>(f0), c = pe_slot<D, L>(f0 + 4);   // lane half 0 / 1
                float r;
                if (a.kind == 0 && c.kind == 0) {
                    r = 0.0f;
                } else {
                    const float xa = (a.kind == 1) ? v[a.axis] : rev[a.axis], xc = (c.kind == 1) ? v[c.axis] : rev[c.axis];
                    const float x = h ? xc : xa;
                    const float t = x * (h ? c.scale : a.scale) + (h ? c.phase : a.phase);
                    const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t));
                    const int kind = h ? c.kind : a.kind;
                    r = (kind == 2) ? sn : ((kind == 1) ? x : 0.0f);
                }
                (void)Z;
                out[b].s[s][j] = (__bf16)r;
            }
}

// trilinear lookup (fp32, ATen corner order, zeros padding); this lane takes channels 16s + 8g + 4h + 0..3
__device__ __forceinline__ void grid_block_h(const float *__restrict__ grid, float x, float y, float z, int h, Blk &out, float *dbg)
{
    const float R1 = (float)(G_RES - 1);
    const float ix = ((x + 1.0f) / 2.0f) * R1, iy = ((y + 1.0f) / 2.0f) * R1, iz = ((z + 1.0f) / 2.0f) * R1;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float wx[2] = {(fx + 1.0f) - ix, ix - fx}, wy[2] = {(fy + 1.0f) - iy, iy - fy}, wz[2] = {(fz + 1.0f) - iz, iz - fz};
    const bool ok = fx >= -1.0f && fx <= (float)G_RES && fy >= -1.0f && fy <= (float)G_RES && fz >= -1.0f && fz <= (float)G_RES;
    const int xi = ok ? (int)fx : -2, yi = ok ? (int)fy : -2, zi = ok ? (int)fz : -2;
    f32x4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int cx = xi + (n & 1), cy = yi + ((n >> 1) & 1), cz = zi + (n >> 2);
        const bool inb = cx >= 0 && cx < G_RES && cy >= 0 && cy < G_RES && cz >= 0 && cz < G_RES;
        const float wt = (wx[n & 1] * wy[(n >> 1) & 1]) * wz[n >> 2];
        const long vox = inb ? (((long)cz * G_RES + cy) * G_RES + cx) : 0;
        const f32x4 *g = reinterpret_cast<const f32x4 *>(grid + vox * D_GRID) + h;   // channels 4h.., 8+4h.., 16+4h.., 24+4h..
        // Branch-free: voxel 0 stands in for a corner outside the grid and its weight is zeroed (x + 0 == x, so this is the
        // zeros-padding sum exactly).  Guarded corners compile to eight exec-masked blocks, i.e. eight DEPENDENT round trips of the
        // feature fetch (8.5 k cycles per tile, both SIMD partners waiting together); this way all 32 loads are in flight at once.
        const float we = inb ? wt : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 gv = g[2 * k];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[k][r] = a[k][r] + gv[r] * we;
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) out.s[s][j] = (__bf16)a[2 * s + (j >> 2)][j & 3];
    if (dbg != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4 *>(dbg + 8 * k + 4 * h) = a[k];
    }
}

#define CH(id) (kProgH.layer[id].G32 * kProgH.layer[id].KB32 * 1024)   /* halfwords in one chunk of layer id */

__device__ __forceinline__ float bcast32(float v, int lane) { return __shfl(v, lane & 31, WAVE); }

__global__ void __launch_bounds__(H_THREADS, 2)
field_forward_bf16_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                          const float *__restrict__ rays, int ray_stride, const float *__restrict__ zvals,
                          float *__restrict__ raw, float *__restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) char lds_h[];
    CtxH cx;
    cx.stream = reinterpret_cast<const unsigned short *>(packed + PACKH_STREAM_OFF) + (long)level * STREAM_HW;
    cx.table = reinterpret_cast<const uint32_t *>(packed + PACKH_TABLE_OFF);
    cx.lds = lds_h;
    cx.chunk = 0;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.h = cx.lane >> 5;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *grid = packed + PACKH_GRID_OFF;
    const int h = cx.h;
#ifndef SAHS_ABLATE_NOPRIO
    // The two waves of a SIMD run the same program in lock step (one barrier per chunk), so their MFMA chains and their
    // VALU repack phases coincide and nothing overlaps.  A static priority for the second-dispatched half lets that wave
    // take the matrix pipe whenever it wants it; its partner fills the gaps, which shifts the pair into anti-phase.
    if (cx.wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    {
        const float *bsrc = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        float *bl = reinterpret_cast<float *>(lds_h + LDS_BIAS_BYTE_OFF);
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += H_THREADS) bl[i] = bsrc[i];
        cx.off = 0;
        cx.prepare(CH(H_W0), 0);
#pragma unroll
        for (int pc = 0; pc < (CH(H_W0) + H_THREADS * 8 - 1) / (H_THREADS * 8); ++pc) cx.issue_piece(pc);
        __syncthreads();
    }
    constexpr const LayerH *Ly = kProgH.layer;

#ifdef SAHS_STAMP
    unsigned long long *stamp_base = reinterpret_cast<unsigned long long *>(dbg);
    dbg = nullptr;
    int tile_no = 0;
#endif
    const long ntiles = (P + H_PTS_PER_WG - 1) / H_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef SAHS_STAMP
        cx.stamp_on = stamp_base != nullptr && blockIdx.x == 0 && tile_no == 3 && (cx.wave == 0 || cx.wave == 4);
        cx.stamps = stamp_base + (cx.wave == 4 ? 512 : 0);
        cx.sidx = 0;
        ++tile_no;
        cx.stamp();                                   // tile start
#endif
        cx.refresh_bias_base();
        const long p_raw = tile * H_PTS_PER_WG + cx.wave * H_PTS_PER_WAVE + (cx.lane & 31);
        const long p = p_raw < P ? p_raw : P - 1;
        const long ray = p / S;
        const float *rp = rays + ray * ray_stride;
        const float z = zvals[p];
        float rd[3] = {rp[3], rp[4], rp[5]};
        float x[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) x[i] = rp[i] + rd[i] * z;

        Blk pe_x[2];
        pe_blocks_h<3, 10, 2>(x, h, pe_x);
        // x' (3) and w (2) are parked in LDS between their uses (trunk input, skip-layer input, grid lookup): every VGPR that
        // is live across the 256-wide layers is one the register allocator would otherwise spill to scratch, and a scratch
        // reload is a VMEM op whose s_waitcnt also drains the in-flight LDS-DMA weight prefetch.
        float *stash = reinterpret_cast<float *>(lds_h + LDS_STASH_BYTE_OFF) + (cx.wave * H_PTS_PER_WAVE + (cx.lane & 31)) * STASH_FLOATS;
        {   // warp field
            Blk hh[4], hn[4];
            dense_h<2, 0, 0, 4, CH(H_W1)>(cx, pe_x, nullptr, nullptr, hh, Ly[H_W0].bias_off, 0.0f);
#pragma unroll 1
            for (int l = 0; l < 2; ++l) {
                dense_h<4, 0, 0, 4, CH(H_W1)>(cx, hh, nullptr, nullptr, hn, Ly[H_W1].bias_off + 128 * l, 0.0f);
#pragma unroll
                for (int i = 0; i < 4; ++i) hh[i] = hn[i];
            }
            dense_h<4, 0, 0, 4, CH(H_W4)>(cx, hh, nullptr, nullptr, hn, Ly[H_W3].bias_off, 0.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i) hh[i] = hn[i];
            dense_h<4, 2, 0, 4, CH(H_W5)>(cx, hh, pe_x, nullptr, hn, Ly[H_W4].bias_off, 0.0f);
            dense_h<4, 0, 0, 4, CH(H_WF)>(cx, hn, nullptr, nullptr, hh, Ly[H_W5].bias_off, 0.0f);
            f32x16 o;
            dense_h_out<4, CH(H_H0)>(cx, hh, o, Ly[H_WF].bias_off, true);
            if (h == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) stash[i] = x[i] + tanhf(o[i]);            // models.py:305 (rows 0..2 live in lane half 0)
            }
        }
        {   // hyper sheet
            Blk hh[2], hn[2];
            dense_h<2, 0, 0, 2, CH(H_H1)>(cx, pe_x, nullptr, nullptr, hh, Ly[H_H0].bias_off, 0.0f);
#pragma unroll 1
            for (int l = 0; l < 2; ++l) {
                dense_h<2, 0, 0, 2, CH(H_H1)>(cx, hh, nullptr, nullptr, hn, Ly[H_H1].bias_off + 64 * l, 0.0f);
#pragma unroll
                for (int i = 0; i < 2; ++i) hh[i] = hn[i];
            }
            dense_h<2, 0, 0, 2, CH(H_H4)>(cx, hh, nullptr, nullptr, hn, Ly[H_H3].bias_off, 0.0f);
#pragma unroll
            for (int i = 0; i < 2; ++i) hh[i] = hn[i];
            dense_h<2, 2, 0, 2, CH(H_H5)>(cx, hh, pe_x, nullptr, hn, Ly[H_H4].bias_off, 0.0f);
            dense_h<2, 0, 0, 2, CH(H_HF)>(cx, hn, nullptr, nullptr, hh, Ly[H_H5].bias_off, 0.0f);
            f32x16 o;
            dense_h_out<2, CH(H_T0)>(cx, hh, o, Ly[H_HF].bias_off, true);
            if (h == 0) { stash[3] = o[0]; stash[4] = o[1]; }
        }
        __builtin_amdgcn_wave_barrier();
        if (dbg != nullptr && h == 0 && p_raw < P) {
            float *dsl = dbg + p * DBG_STRIDE_H;
            dsl[0] = stash[0] - x[0]; dsl[1] = stash[1] - x[1]; dsl[2] = stash[2] - x[2]; dsl[3] = stash[3]; dsl[4] = stash[4];
        }
        // radiance trunk
        Blk feat[8];
        f32x16 fin;
        {
            Blk hh[8];
            {   // the re-injected encoding [PE(x') | PE(w)] is rebuilt at the skip layer instead of staying live (24 VGPRs)
                Blk in_tr[3];
                const float xw[3] = {stash[0], stash[1], stash[2]}, amb[2] = {stash[3], stash[4]};
                pe_blocks_h<3, 10, 2>(xw, h, in_tr);
                pe_blocks_h<2, 4, 1>(amb, h, in_tr + 2);
                dense_h<2, 1, 0, 8, CH(H_T1)>(cx, in_tr, in_tr + 2, nullptr, hh, Ly[H_T0].bias_off, 0.01f);
            }
            // T1, T2, T3 (skip), T4..T7, FEAT: 256-wide layers; hh <- feat between them
            dense_h<8, 0, 0, 8, CH(H_T2)>(cx, hh, nullptr, nullptr, feat, Ly[H_T1].bias_off, 0.01f);
#pragma unroll
            for (int i = 0; i < 8; ++i) hh[i] = feat[i];
            dense_h<8, 0, 0, 8, CH(H_T3)>(cx, hh, nullptr, nullptr, feat, Ly[H_T2].bias_off, 0.01f);
#pragma unroll
            for (int i = 0; i < 8; ++i) hh[i] = feat[i];
            {
                Blk in_tr[3];
                const float xw[3] = {stash[0], stash[1], stash[2]}, amb[2] = {stash[3], stash[4]};
                pe_blocks_h<3, 10, 2>(xw, h, in_tr);
                pe_blocks_h<2, 4, 1>(amb, h, in_tr + 2);
                dense_h<8, 2, 1, 8, CH(H_T4)>(cx, hh, in_tr, in_tr + 2, feat, Ly[H_T3].bias_off, 0.01f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) hh[i] = feat[i];
#pragma unroll 1
            for (int l = 4; l <= 7; ++l) {
                dense_h<8, 0, 0, 8, CH(H_T5)>(cx, hh, nullptr, nullptr, feat, Ly[H_T4].bias_off + 256 * (l - 4), 0.01f);
#pragma unroll
                for (int i = 0; i < 8; ++i) hh[i] = feat[i];
            }
            dense_h<8, 0, 0, 8, CH(H_ALPHA)>(cx, hh, nullptr, nullptr, feat, Ly[H_FEAT].bias_off, 1.0f);
        }
        dense_h_out<8, CH(H_D0)>(cx, feat, fin, Ly[H_ALPHA].bias_off, true);
        {   // colour branch
            Blk in_d[2];
            {
                const long pr = tile * H_PTS_PER_WG + cx.wave * H_PTS_PER_WAVE + (cx.lane & 31);
                const long pp = pr < P ? pr : P - 1;
                const float *rq = rays + (pp / S) * ray_stride;
                const float rdir[3] = {rq[3], rq[4], rq[5]};                         // re-read: cheaper than keeping it live
                pe_blocks_h<3, 4, 1>(rdir, h, in_d);
                grid_block_h(grid, stash[0], stash[1], stash[2], h, in_d[1], (dbg != nullptr && pr < P) ? dbg + P * DBG_STRIDE_H + pp * 32 : nullptr);
            }
            Blk c[4], cn[4];
            dense_h<8, 1, 1, 4, CH(H_D1)>(cx, feat, in_d, in_d + 1, c, Ly[H_D0].bias_off, 0.01f);
#pragma unroll 1
            for (int l = 0; l < 2; ++l) {
                dense_h<4, 0, 0, 4, CH(H_D1)>(cx, c, nullptr, nullptr, cn, Ly[H_D1].bias_off + 128 * l, 0.01f);
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = cn[i];
            }
            dense_h<4, 0, 0, 4, CH(H_RGB)>(cx, c, nullptr, nullptr, cn, Ly[H_D3].bias_off, 0.01f);
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = cn[i];
            dense_h_out<4, CH(H_S0)>(cx, c, fin, 0, false);
        }
        {   // seg branch
            Blk s[4], sn[4];
            dense_h<8, 0, 0, 4, CH(H_S1)>(cx, feat, nullptr, nullptr, s, Ly[H_S0].bias_off, 0.01f);
#pragma unroll 1
            for (int l = 0; l < 2; ++l) {
                dense_h<4, 0, 0, 4, CH(H_S1)>(cx, s, nullptr, nullptr, sn, Ly[H_S1].bias_off + 128 * l, 0.01f);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] = sn[i];
            }
            dense_h<4, 0, 0, 4, CH(H_SEG)>(cx, s, nullptr, nullptr, sn, Ly[H_S3].bias_off, 0.01f);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = sn[i];
            dense_h_out<4, CH(H_W0)>(cx, s, fin, 0, false);
        }
        const long p_out = tile * H_PTS_PER_WG + cx.wave * H_PTS_PER_WAVE + (cx.lane & 31);
        if (p_out < P) {   // rows 4h..4h+3 and 8+4h..8+4h+3 of [rgb3 | seg12 | sigma]
#ifdef SAHS_ABLATE_NTSTORE
            __builtin_nontemporal_store(f32x4{fin[0], fin[1], fin[2], fin[3]}, reinterpret_cast<f32x4 *>(raw + p_out * D_RAW + 4 * h));
            __builtin_nontemporal_store(f32x4{fin[4], fin[5], fin[6], fin[7]}, reinterpret_cast<f32x4 *>(raw + p_out * D_RAW + 8 + 4 * h));
#else
            *reinterpret_cast<f32x4 *>(raw + p_out * D_RAW + 4 * h) = f32x4{fin[0], fin[1], fin[2], fin[3]};
            *reinterpret_cast<f32x4 *>(raw + p_out * D_RAW + 8 + 4 * h) = f32x4{fin[4], fin[5], fin[6], fin[7]};
#endif
        }
    }
}

}  // namespace hb
}  // namespace sahs

using namespace sahs;
using namespace sahs::hb;

extern "C" int sahs_field_forward_bf16_launch(const float *packed, const float *frame, int level, long P, int S, const float *rays,
                                              int ray_stride, const float *zvals, float *raw, float *dbg, int num_cu,
                                              hipStream_t stream)
{
    if (P <= 0) return 0;
    const long ntiles = (P + H_PTS_PER_WG - 1) / H_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;       // the large-LDS attribute is per device
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_forward_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    field_forward_bf16_kernel<<<grid, H_THREADS, LDS_BYTES, stream>>>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg);
    return (int)hipGetLastError();
}
