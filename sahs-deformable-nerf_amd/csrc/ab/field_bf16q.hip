// field_bf16q.hip -- the bf16 field kernel on v_mfma_f32_16x16x32_bf16 (precision SAHS_BF16_Q; experiment of round 2).
//
// Why: under a dense bf16 MFMA load the MI355X holds its clock down, and how far depends on the MFMA shape: a bare loop of
// v_mfma_f32_32x32x16_bf16 runs at 1.75-1.82 GHz on random data, the same FLOPs as v_mfma_f32_16x16x32_bf16 at 2.14 GHz (+16 %:
// tools/micro/mfma_shape_clock.hip; MI355X_MICROARCH.md).  This is field_bf16w.hip (one wave per SIMD, 64 samples per wave: read that file
// for the structure) re-tiled onto the 16x16x32 shape; everything a chunk / a step / a wait count means is unchanged:
//   * a wave's 64 samples are FOUR quarters of 16 (lane = 16 g + c: column c of a quarter, row group g);
//   * one step is still one 1-KB A fragment -- now 16 output rows x 32 input features -- and feeds four MFMAs (one per quarter); a 32-row
//     output tile is two row tiles rt = 0, 1, so a tile still takes 2 * KB steps, step k = (k-block k >> 1, row tile k & 1);
//   * the D tile of a quarter (lane (c, g): rows 4g..4g+3, 4 registers) is the B operand of the next layer: the 8 values a lane holds of a
//     32-row tile -- rows {4g..4g+3} u {16+4g..16+4g+3} -- become one bf16x8 fragment, element j <-> feature 16 (j >> 2) + 4 g + (j & 3) of
//     the block (the packed stream is in the matching k order: pack_stream_bf16q_kernel), so again no lane movement;
//   * the bias is the C operand of each row tile's first MFMA (one ds_read_b128 per row tile: rows 16 rt + 4g .. + 3).
// AudioFaceModel only.  Arithmetic: the products, fp32 accumulation and bf16 roundings of field_bf16w.hip; the summation order over k
// differs (32 products per MFMA in another grouping), so results agree with it to fp32 rounding of the accumulations, not bit for bit.
#include <hip/hip_runtime.h>
#include <utility>
#include "../sahs_common.hpp"
#include "../sahs_layout.hpp"

#if SAHS_MODEL != 0
#error "field_bf16q.hip is built for the AudioFaceModel only"
#endif

namespace sahs {
namespace hq {
using namespace hb;      // the bf16 layer program, chunking and packed sizes of sahs_layout.hpp (the stream differs only inside a tile)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

enum { FIELD_ALL = 0, FIELD_DEFORM = 1, FIELD_RADIANCE = 2 };
constexpr int NQ = 4;                              // 16-sample quarters per wave
struct Blk { u32x4 s[NQ]; };                       // 32 features of this lane's four samples: one bf16x8 fragment per quarter (as dwords)

constexpr int Q_THREADS = 256;
constexpr int Q_PTS_PER_WAVE = 16 * NQ;
constexpr int Q_PTS_PER_WG = (Q_THREADS / WAVE) * Q_PTS_PER_WAVE;     // 256
constexpr int LDS_BUF_BYTES = CHUNK_HW_MAX * 2;                       // 64 KB each, two of them
constexpr int LDS_BIAS_BYTE_OFF = 2 * LDS_BUF_BYTES;
constexpr int LDS_STASH_BYTE_OFF = LDS_BIAS_BYTE_OFF + ((BIAS_FLOATS + 3) / 4) * 16;
constexpr int STASH_FLOATS = 8;                                       // per sample: x'[3], w[2] (+pad)
constexpr int LDS_BYTES = LDS_STASH_BYTE_OFF + Q_PTS_PER_WG * STASH_FLOATS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int AP = 4;                                                  // A fragments in flight ahead of the MFMAs that consume them
constexpr int PIECE_HW = Q_THREADS * 8;                                // one LDS-DMA piece: 4 KB = 2048 halfwords (1 KB per wave)
constexpr int DMA_PIECES = LDS_BUF_BYTES / (Q_THREADS * 16);           // 16

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ uint32_t lds_addr_of(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p; }

template <int OFF, class V>
__device__ __forceinline__ void lds_read16(V &dst, uint32_t addr)
{
    static_assert(sizeof(V) == 16, "one ds_read_b128");
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_lgkm()
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}
__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }

// run f(integral_constant<int, q>) for q = 0 .. NQ-1 with q a compile-time constant (per-quarter state must only ever be indexed by constants)
template <class F>
__device__ __forceinline__ void for_quarters(F &&f)
{
    [&]<int... Qs>(std::integer_sequence<int, Qs...>) { (f(std::integral_constant<int, Qs>{}), ...); }(std::make_integer_sequence<int, NQ>{});
}

struct Ctx {
    const unsigned short *stream;
    char *lds;
    int buf;
    int lane, g, wave;              // g = lane >> 4: row group of the MFMA layouts
    const f32x4 *nx_src; f32x4 *nx_dst;
    uint32_t off;                   // halfword offset of the NEXT chunk to prefetch (uniform)
    uint32_t wrap_at, wrap_to;
    uint32_t bias_addr;             // LDS byte address of this lane's first bias row (+4g rows)

    __device__ __forceinline__ void prepare(int hw, int b)
    {
        if (off >= wrap_at) off = wrap_to;
        nx_src = reinterpret_cast<const f32x4 *>(stream + off) + lane;
        nx_dst = reinterpret_cast<f32x4 *>(lds + b * LDS_BUF_BYTES);
        off += (uint32_t)hw;
    }
    __device__ __forceinline__ void issue_piece(int p)
    {
        const int base = p * Q_THREADS + wave * WAVE;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(nx_src + base), (lds_ptr_t)(nx_dst + base), 16, 0, 0);
    }
    __device__ __forceinline__ void begin_chunk(int next_hw) { prepare(next_hw, buf ^ 1); }
    __device__ __forceinline__ void end_chunk()
    {
        __syncthreads();            // vmcnt(0) (the next chunk has landed) + barrier (every wave is done with this one)
        buf ^= 1;
    }
    __device__ __forceinline__ uint32_t cur_addr() const { return lds_addr_of(lds + buf * LDS_BUF_BYTES) + 16 * lane; }
    __device__ __forceinline__ void refresh_bias_base()
    {
        uint32_t a = lds_addr_of(lds) + LDS_BIAS_BYTE_OFF + 16 * g;
        asm volatile("" : "+v"(a));
        bias_addr = a;
    }
};

// this lane's bias rows of a 32-row tile: rows 16 rt + 4g .. + 3 for rt = 0, 1 (two ds_read_b128, 64 bytes apart)
template <int OFF>
__device__ __forceinline__ void bias_read(f32x4 (&t)[2], uint32_t addr)
{
    lds_read16<OFF>(t[0], addr);
    lds_read16<OFF + 64>(t[1], addr);
}

__device__ __forceinline__ uint32_t cvt_pair(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2)); }

// ---- activation + bf16 conversion of a finished tile (32 values per lane: [row tile 2][quarter 4][register 4]), software-pipelined as in
// field_bf16w.hip.  Value U: pair P = U >> 1 -> quarter P & 3, fragment dword jp = P >> 2 (row tile jp >> 1, registers 2 (jp & 1) + (U & 1)).
struct Acc { f32x4 t[2][NQ]; };
struct PackState { f32x2 m2[2]; float r[4]; uint32_t d[2]; };
constexpr int NV = 32;
constexpr int PACK_TICKS = NV + 4;
template <int T>
__device__ __forceinline__ void pack_tick(const Acc &acc, Blk &o, float slope, PackState &ps)
{
    if (slope == 0.0f) {       // ReLU: round first, clamp the packed pair as signed 16-bit integers (field_bf16w.hip)
        if constexpr (T - 2 >= 1 && T - 2 < NV && ((T - 2) & 1)) {
            constexpr int U = T - 2, P = U >> 1, qq = P & 3, jp = P >> 2, rt = jp >> 1, rp = jp & 1;
            ps.d[P & 1] = cvt_pair(acc.t[rt][qq][2 * rp], acc.t[rt][qq][2 * rp + 1]);
        }
        if constexpr (T - 4 >= 1 && T - 4 < NV && ((T - 4) & 1)) {
            constexpr int U = T - 4, P = U >> 1, qq = P & 3, jp = P >> 2;
            o.s[qq][jp] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ps.d[P & 1]), s16x2{0, 0}));
        }
        return;
    }
#ifndef SAHS_BF16W_EXACT_LEAKY      // LeakyReLU on the packed bf16 bit patterns (field_bf16w.hip: pack_tick)
    if (slope != 1.0f) {
        if constexpr (T - 2 >= 1 && T - 2 < NV && ((T - 2) & 1)) {
            constexpr int U = T - 2, P = U >> 1, qq = P & 3, jp = P >> 2, rt = jp >> 1, rp = jp & 1;
            ps.d[P & 1] = cvt_pair(acc.t[rt][qq][2 * rp], acc.t[rt][qq][2 * rp + 1]);
        }
        if constexpr (T - 3 >= 1 && T - 3 < NV && ((T - 3) & 1)) {
            constexpr int U = T - 3, P = U >> 1;
            uint32_t sg;
            asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(sg) : "v"(ps.d[P & 1]));
            ps.r[P & 1] = __builtin_bit_cast(float, sg);
        }
        if constexpr (T - 4 >= 1 && T - 4 < NV && ((T - 4) & 1)) {
            constexpr int U = T - 4, P = U >> 1, qq = P & 3, jp = P >> 2;
            const uint32_t kk = 0x03520352u;
            uint32_t o_;
            asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(o_) : "v"(__builtin_bit_cast(uint32_t, ps.r[P & 1])), "s"(kk), "v"(ps.d[P & 1]));
            o.s[qq][jp] = o_;
        }
        return;
    }
#endif
    if constexpr (T >= 0 && T < NV && !(T & 1)) {           // A(T), T even: both values of the pair
        constexpr int P = T >> 1, qq = P & 3, jp = P >> 2, rt = jp >> 1, rp = jp & 1;
        if (slope != 1.0f) ps.m2[P & 1] = f32x2{acc.t[rt][qq][2 * rp], acc.t[rt][qq][2 * rp + 1]} * f32x2{slope, slope};
    }
    if constexpr (T - 1 >= 0 && T - 1 < NV) {               // B(T-1)
        constexpr int U = T - 1, P = U >> 1, qq = P & 3, jp = P >> 2, rt = jp >> 1, rp = jp & 1, e = U & 1;
        const float v = acc.t[rt][qq][2 * rp + e];
        ps.r[U & 3] = slope == 1.0f ? v : fmaxf(v, ps.m2[P & 1][e]);
    }
    if constexpr (T - 2 >= 1 && T - 2 < NV && ((T - 2) & 1)) {    // C(T-2): the pair (T-3, T-2) is complete
        constexpr int U = T - 2, P = U >> 1, qq = P & 3, jp = P >> 2;
        o.s[qq][jp] = cvt_pair(ps.r[(U - 1) & 3], ps.r[U & 3]);
    }
}
template <int LO, int HI>
__device__ __forceinline__ void pack_ticks(const Acc &acc, Blk &o, float slope, PackState &ps)
{
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (pack_tick<LO + Is>(acc, o, slope, ps), ...); }(std::make_integer_sequence<int, HI - LO>{});
}

__device__ __forceinline__ bf16x8 frag(const u32x4 &v) { return __builtin_bit_cast(bf16x8, v); }

// LGKM bookkeeping of field_bf16w.hip with a bias batch of TWO reads: step I issues [wait] [4 MFMAs] [bias batch of the next tile, when I
// starts a tile that has a successor] [A read I+AP]; the wait of step I allows exactly the reads issued after A(I).
template <int STEPS, int TOTAL, int NT32, int T0>
struct Sched {
    static constexpr bool bias_at(int s) { return s >= 0 && s < TOTAL && (s % STEPS == 0) && (T0 + s / STEPS + 1 < NT32); }
    static constexpr bool aread_at(int s) { return s >= 0 && s + AP < TOTAL; }
    static constexpr int cnt(int I)
    {
        int n = 0, lo = 0;
        if (I < AP) n += ((AP < TOTAL ? AP : TOTAL) - 1 - I);
        else lo = I - AP + 1;
        for (int s = lo; s < I; ++s) n += (aread_at(s) ? 1 : 0) + (bias_at(s) ? 2 : 0);
        return n;
    }
};

struct St {
    Acc acc[2];          // two accumulator sets: tile t accumulates into one while tile t-1 is converted from the other
    PackState ps;
};

// hidden layer: NT32 output tiles of 32 rows.  The layer's last tile stays in st.acc[1]; the NEXT layer converts it (PEND) into in0[K0-1].
template <int K0, int K1, int K2, int NT32, int NEXT_HW, bool PEND>
__device__ __forceinline__ void dense_q(Ctx &cx, St &st, Blk *in0, const Blk *in1, const Blk *in2, Blk *out, int bias_off, float slope, float pslope)
{
    constexpr int KB = K0 + K1 + K2;
    constexpr int G = pick_G32(KB, NT32);
    constexpr int STEPS = KB * 2, TOTAL = G * STEPS, NCH = NT32 / G;
    static_assert(NT32 % 2 == 0, "the last tile of a layer must land in accumulator set 1");
    const uint32_t baddr = cx.bias_addr + 4u * (uint32_t)bias_off;
    f32x4 braw[2][2];                                         // raw bias reads of the current / next tile: [tile parity][row tile]
    bias_read<0>(braw[0], baddr);

    auto chunk = [&]<int C>() {
        constexpr int T0 = C * G;
        constexpr int nhw = (C + 1 < NCH) ? G * KB * 1024 : NEXT_HW;
        constexpr int npieces = (nhw + PIECE_HW - 1) / PIECE_HW;
        constexpr int PSTEP = (TOTAL * 3 / 4) / (npieces > 0 ? npieces : 1) > 0 ? (TOTAL * 3 / 4) / (npieces > 0 ? npieces : 1) : 1;
        using S = Sched<STEPS, TOTAL, NT32, T0>;
        cx.begin_chunk(nhw);
        const uint32_t abase = cx.cur_addr();
        u32x4 a[AP];
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (lds_read16<Is * 1024>(a[Is], abase), ...); }(
            std::make_integer_sequence<int, (AP < TOTAL ? AP : TOTAL)>{});
        fence();
        auto step = [&]<int I>() {
            constexpr int g = I / STEPS, k = I % STEPS, b = k >> 1, rt = k & 1, t = T0 + g, set = t & 1;
            const Blk &x = (b < K0) ? in0[b] : ((b < K0 + K1) ? in1[b - K0] : in2[b - K0 - K1]);
            wait_lgkm<S::cnt(I)>();
            fence();
            for_quarters([&](auto Q) {
                constexpr int qq = decltype(Q)::value;
                // the row tile's chains start from its bias (b == 0; landed: retired by the wait of step k = 0 of this tile)
                if constexpr (b == 0) st.acc[set].t[rt][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[I % AP]), frag(x.s[qq]), braw[set][rt], 0, 0, 0);
                else st.acc[set].t[rt][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[I % AP]), frag(x.s[qq]), st.acc[set].t[rt][qq], 0, 0, 0);
                fence();
                if constexpr (qq == 0) {
                    if constexpr (S::bias_at(I)) bias_read<128 * (t + 1)>(braw[set ^ 1], baddr);
                }
                if constexpr (qq == NQ - 1) {
                    if constexpr (S::aread_at(I)) lds_read16<(I + AP) * 1024>(a[I % AP], abase);
                    if constexpr (I % PSTEP == 0 && I / PSTEP < npieces) cx.issue_piece(I / PSTEP);
                }
                if constexpr (t > 0 && k >= 1) {              // the finished tile t-1 -> out[t-1]: this gap's share of its conversion ticks
                    constexpr int NSLOT = NQ * (STEPS - 1), slot = NQ * (k - 1) + qq;
                    constexpr int lo = PACK_TICKS * slot / NSLOT, hi = PACK_TICKS * (slot + 1) / NSLOT;
                    if constexpr (hi > lo) pack_ticks<lo, hi>(st.acc[set ^ 1], out[t - 1], slope, st.ps);
                } else if constexpr (PEND && t == 0 && k >= 1 && k <= 2 * K0 - 3) {     // the previous layer's last tile -> in0[K0-1]
                    constexpr int NSLOT = NQ * (2 * K0 - 3), slot = NQ * (k - 1) + qq;
                    constexpr int lo = PACK_TICKS * slot / NSLOT, hi = PACK_TICKS * (slot + 1) / NSLOT;
                    if constexpr (hi > lo) pack_ticks<lo, hi>(st.acc[1], in0[K0 - 1], pslope, st.ps);
                }
                fence();
            });
        };
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (step.template operator()<Is>(), ...); }(std::make_integer_sequence<int, TOTAL>{});
        [&]<int... Ps>(std::integer_sequence<int, Ps...>) {
            ((Ps >= (TOTAL + PSTEP - 1) / PSTEP && Ps < npieces ? cx.issue_piece(Ps) : (void)0), ...);
        }(std::make_integer_sequence<int, DMA_PIECES>{});
        cx.end_chunk();
    };
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) { (chunk.template operator()<Cs>(), ...); }(std::make_integer_sequence<int, NCH>{});
}

// 16-row output layer (WF, HF; ALPHA -> RGB -> SEG into one tile): row tile 0 of its 32-row tile; first = start from the bias, else from the
// running tile.  Row tile 1 of the packed tile is all zero: its steps read their fragment (the counted waits stay those of dense_q) and
// issue no MFMA.  The previous layer's last tile (st.acc[1]) is converted into in0[K0-1] here.
template <int K0, int NEXT_HW>
__device__ __forceinline__ void dense_q_out(Ctx &cx, St &st, Blk *in0, f32x4 (&acc)[NQ], int bias_off, bool first, float pslope)
{
    constexpr int TOTAL = K0 * 2;
    constexpr int npieces = (NEXT_HW + PIECE_HW - 1) / PIECE_HW;
    cx.begin_chunk(NEXT_HW);
    const uint32_t abase = cx.cur_addr();
    if (first) {       // rows 4g .. 4g+3 of the layer's bias
        const uint32_t baddr = cx.bias_addr + 4u * (uint32_t)bias_off;
        f32x4 t0;
        lds_read16<0>(t0, baddr);
        wait_lgkm<0>();
        fence();
        for_quarters([&](auto Q) { acc[decltype(Q)::value] = t0; });
    }
    u32x4 a[AP];
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (lds_read16<Is * 1024>(a[Is], abase), ...); }(std::make_integer_sequence<int, (AP < TOTAL ? AP : TOTAL)>{});
    fence();
    auto step = [&]<int I>() {
        wait_lgkm<((TOTAL - 1 - I) < (AP - 1) ? (TOTAL - 1 - I) : (AP - 1))>();
        fence();
        // An odd step's fragment (row tile 1: zero rows) feeds no MFMA -- but its read WAS issued, and a destination the compiler
        // believes dead is handed out again while the LDS data is still on its way (found on the ISA: a DMA address computed into
        // those registers two instructions after the ds_read, then overwritten by the landing data: garbage addresses, a memory
        // fault).  This use, placed after the wait that retires the read, keeps the registers reserved until the data has landed.
        if constexpr ((I & 1) == 1) asm volatile("" :: "v"(a[I % AP]));
        for_quarters([&](auto Q) {
            constexpr int qq = decltype(Q)::value;
            if constexpr ((I & 1) == 0) {
                acc[qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[I % AP]), frag(in0[I >> 1].s[qq]), acc[qq], 0, 0, 0);
                fence();
            }
            if constexpr (qq == NQ - 1) {
                if constexpr (I + AP < TOTAL) lds_read16<(I + AP) * 1024>(a[I % AP], abase);
                if constexpr (I < npieces) cx.issue_piece(I);
            }
            if constexpr (I >= 1 && I <= 2 * K0 - 3) {        // the previous layer's last tile -> in0[K0-1], before step 2 (K0 - 1) reads it
                constexpr int NSLOT = NQ * (2 * K0 - 3), slot = NQ * (I - 1) + qq;
                constexpr int lo = PACK_TICKS * slot / NSLOT, hi = PACK_TICKS * (slot + 1) / NSLOT;
                if constexpr (hi > lo) pack_ticks<lo, hi>(st.acc[1], in0[K0 - 1], pslope, st.ps);
            }
            fence();
        });
    };
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (step.template operator()<Is>(), ...); }(std::make_integer_sequence<int, TOTAL>{});
    [&]<int... Ps>(std::integer_sequence<int, Ps...>) { ((Ps >= TOTAL && Ps < npieces ? cx.issue_piece(Ps) : (void)0), ...); }(
        std::make_integer_sequence<int, DMA_PIECES>{});
    cx.end_chunk();
}

// ---- positional encoding (v_sin_f32 in revolutions, as field_bf16w.hip).  Element j of lane row group g in block b is feature
// 32 b + 16 (j >> 2) + 4 g + (j & 3): the slot (scale, phase, axis, kind) depends on g, a run-time value -- four-way selects -------------
struct PeSlot { float scale; float phase; int axis; int kind; };   // kind: 0 zero pad, 1 raw input, 2 sinusoid
template <int D, int L>
constexpr PeSlot pe_slot(int f)
{
    constexpr int W = D + 2 * D * L;
    if (f >= W) return PeSlot{0.0f, 0.0f, 0, 0};
    if (f < D) return PeSlot{1.0f, 0.0f, f, 1};
    const int gg = f - D, k = gg / (2 * D), rem = gg % (2 * D);
    return PeSlot{(float)(1 << k), (rem / D) ? 0.25f : 0.0f, rem % D, 2};
}
template <class T>
__device__ __forceinline__ T sel4(int g, T a, T b, T c, T d) { return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d)); }

template <int D, int L, int NB>
__device__ __forceinline__ void pe_blocks_q(const float (*v)[3], int g, Blk *out)
{
    for_quarters([&](auto Q) {
        constexpr int qq = decltype(Q)::value;
        float rev[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) rev[i] = i < D ? v[qq][i] * 0.15915494309189535f : 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f0 = 32 * b + 16 * (j >> 2) + (j & 3);
                const PeSlot s0 = pe_slot<D, L>(f0), s1 = pe_slot<D, L>(f0 + 4), s2 = pe_slot<D, L>(f0 + 8), s3 = pe_slot<D, L>(f0 + 12);
                if (s0.kind == 0 && s1.kind == 0 && s2.kind == 0 && s3.kind == 0) {
                    r[j] = 0.0f;
                } else {
                    const int kind = sel4(g, s0.kind, s1.kind, s2.kind, s3.kind), axis = sel4(g, s0.axis, s1.axis, s2.axis, s3.axis);
                    const float scale = sel4(g, s0.scale, s1.scale, s2.scale, s3.scale), phase = sel4(g, s0.phase, s1.phase, s2.phase, s3.phase);
                    const float xr = axis == 0 ? rev[0] : (axis == 1 ? rev[1] : rev[2]);
                    const float xv = axis == 0 ? v[qq][0] : (axis == 1 ? v[qq][1] : v[qq][2]);
                    const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(xr * scale + phase));
                    r[j] = (kind == 2) ? sn : ((kind == 1) ? xv : 0.0f);
                }
            }
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) out[b].s[qq][jp] = cvt_pair(r[2 * jp], r[2 * jp + 1]);
        }
    });
}

// trilinear lookup (fp32, ATen corner order, zeros padding) for one sample; this lane takes channels 4g..4g+3 and 16+4g..16+4g+3
__device__ __forceinline__ void grid_block_q(const float *__restrict__ grid, float x, float y, float z, int g, u32x4 &out)
{
    const float R1 = (float)(G_RES - 1);
    const float ix = ((x + 1.0f) / 2.0f) * R1, iy = ((y + 1.0f) / 2.0f) * R1, iz = ((z + 1.0f) / 2.0f) * R1;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float wx[2] = {(fx + 1.0f) - ix, ix - fx}, wy[2] = {(fy + 1.0f) - iy, iy - fy}, wz[2] = {(fz + 1.0f) - iz, iz - fz};
    const bool ok = fx >= -1.0f && fx <= (float)G_RES && fy >= -1.0f && fy <= (float)G_RES && fz >= -1.0f && fz <= (float)G_RES;
    const int xi = ok ? (int)fx : -2, yi = ok ? (int)fy : -2, zi = ok ? (int)fz : -2;
    f32x4 a[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int cx = xi + (n & 1), cy = yi + ((n >> 1) & 1), cz = zi + (n >> 2);
        const bool inb = cx >= 0 && cx < G_RES && cy >= 0 && cy < G_RES && cz >= 0 && cz < G_RES;
        const float wt = (wx[n & 1] * wy[(n >> 1) & 1]) * wz[n >> 2];
        const long vox = inb ? (((long)cz * G_RES + cy) * G_RES + cx) : 0;
        const f32x4 *gp = reinterpret_cast<const f32x4 *>(grid + vox * D_GRID) + g;   // channels 4g.., 16+4g..
        const float we = inb ? wt : 0.0f;        // branch-free zeros padding: all loads in flight at once
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x4 gv = gp[4 * k];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[k][r] = a[k][r] + gv[r] * we;
        }
    }
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) out[jp] = cvt_pair(a[jp >> 1][2 * (jp & 1)], a[jp >> 1][2 * (jp & 1) + 1]);
}

#define CH(id) (kProgH.layer[id].G32 * kProgH.layer[id].KB32 * 1024)   /* halfwords in one chunk of layer id */

// MODE: the split evaluation of field_f32.hip / field_bf16w.hip, same contract
template <int MODE>
__global__ void __launch_bounds__(Q_THREADS, 1)
field_forward_bf16q_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                           const float *__restrict__ rays, int ray_stride, const float *__restrict__ zvals,
                           float *__restrict__ raw, float *__restrict__ xw, int xw_row, int xw_col0, const int *__restrict__ src)
{
    constexpr uint32_t RAD_OFF = (uint32_t)kProgH.layer[H_T0].stream_off;
    constexpr int L_FIRST = MODE == FIELD_RADIANCE ? H_T0 : H_W0;
    constexpr int AFTER_DEFORM = MODE == FIELD_DEFORM ? H_W0 : H_T0;
    constexpr int AFTER_RADIANCE = MODE == FIELD_RADIANCE ? H_T0 : H_W0;
    extern __shared__ __attribute__((aligned(16))) char lds_q[];
    Ctx cx;
    cx.stream = reinterpret_cast<const unsigned short *>(packed + PACKH_STREAM_OFF) + (long)level * STREAM_HW;
    cx.lds = lds_q;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.g = cx.lane >> 4;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *grid = packed + PACKH_GRID_OFF;
    const int g = cx.g, col = cx.lane & 15;
    {
        const float *bsrc = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        float *bl = reinterpret_cast<float *>(lds_q + LDS_BIAS_BYTE_OFF);
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += Q_THREADS) bl[i] = bsrc[i];
        cx.wrap_at = MODE == FIELD_DEFORM ? RAD_OFF : (uint32_t)STREAM_HW;
        cx.wrap_to = MODE == FIELD_RADIANCE ? RAD_OFF : 0u;
        cx.off = cx.wrap_to;
        cx.prepare(CH(L_FIRST), 0);
#pragma unroll
        for (int pc = 0; pc < (CH(L_FIRST) + PIECE_HW - 1) / PIECE_HW; ++pc) cx.issue_piece(pc);
        __syncthreads();
    }
    constexpr const LayerH *Ly = kProgH.layer;

    const long ntiles = (P + Q_PTS_PER_WG - 1) / Q_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cx.refresh_bias_base();
        asm volatile("" : "+s"(cx.off));      // (chunk addresses are loop-invariant: keep them from being hoisted and spilled)
        St st;
        // this lane's four samples: quarter qq -> sample (wave*64 + 16 qq + col) of the workgroup tile
        long p_raw[NQ], p[NQ];
        float x[NQ][3];
        typedef __attribute__((address_space(3))) float *lds_float;
        const lds_float stash0 = (lds_float)(__attribute__((address_space(3))) char *)(lds_q + LDS_STASH_BYTE_OFF) + (cx.wave * Q_PTS_PER_WAVE + col) * STASH_FLOATS;
        auto stash = [&](int qq) { return stash0 + qq * 16 * STASH_FLOATS; };
        for_quarters([&](auto Q) {
            constexpr int qq = decltype(Q)::value;
            p_raw[qq] = tile * Q_PTS_PER_WG + cx.wave * Q_PTS_PER_WAVE + qq * 16 + col;
            p[qq] = p_raw[qq] < P ? p_raw[qq] : P - 1;
            if constexpr (MODE != FIELD_RADIANCE) {
                const float *rp = rays + (p[qq] / S) * ray_stride;
                const float z = zvals[p[qq]];
#pragma unroll
                for (int i = 0; i < 3; ++i) x[qq][i] = rp[i] + rp[3 + i] * z;
            }
        });
        if constexpr (MODE == FIELD_RADIANCE) {     // x', w come from the launches that deformed these samples
            if (g == 0) {
                for_quarters([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    const float *row = xw + ((p[q] / S) * (long)xw_row + (src != nullptr ? src[p[q]] : (int)(p[q] % S))) * 8;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(row);
                    stash(q)[0] = v[0]; stash(q)[1] = v[1]; stash(q)[2] = v[2]; stash(q)[3] = v[3];
                    stash(q)[4] = row[4];
                });
            }
        } else {
            Blk pe_x[2];
            pe_blocks_q<3, 10, 2>(x, g, pe_x);
            {   // warp field (layers alternate between two register sets: no activation copies)
                Blk A[4], B[4];
                dense_q<2, 0, 0, 4, CH(H_W1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_W0].bias_off, 0.0f, 0.0f);
                dense_q<4, 0, 0, 4, CH(H_W1), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W1].bias_off, 0.0f, 0.0f);
                dense_q<4, 0, 0, 4, CH(H_W1), true>(cx, st, B, nullptr, nullptr, A, Ly[H_W1].bias_off + 128, 0.0f, 0.0f);
                dense_q<4, 0, 0, 4, CH(H_W4), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W3].bias_off, 0.0f, 0.0f);
                dense_q<4, 2, 0, 4, CH(H_W5), true>(cx, st, B, pe_x, nullptr, A, Ly[H_W4].bias_off, 0.0f, 0.0f);
                dense_q<4, 0, 0, 4, CH(H_WF), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W5].bias_off, 0.0f, 0.0f);
                f32x4 o[NQ];
                dense_q_out<4, CH(H_H0)>(cx, st, B, o, Ly[H_WF].bias_off, true, 0.0f);
                if (g == 0) {      // rows 0..2 of the output tile live in row group 0
                    for_quarters([&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        stash(q)[0] = x[q][0] + tanhf(o[q][0]);            // models.py:305
                        stash(q)[1] = x[q][1] + tanhf(o[q][1]);
                        stash(q)[2] = x[q][2] + tanhf(o[q][2]);
                    });
                }
            }
            {   // hyper sheet
                Blk A[2], B[2];
                dense_q<2, 0, 0, 2, CH(H_H1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_H0].bias_off, 0.0f, 0.0f);
                dense_q<2, 0, 0, 2, CH(H_H1), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H1].bias_off, 0.0f, 0.0f);
                dense_q<2, 0, 0, 2, CH(H_H1), true>(cx, st, B, nullptr, nullptr, A, Ly[H_H1].bias_off + 64, 0.0f, 0.0f);
                dense_q<2, 0, 0, 2, CH(H_H4), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H3].bias_off, 0.0f, 0.0f);
                dense_q<2, 2, 0, 2, CH(H_H5), true>(cx, st, B, pe_x, nullptr, A, Ly[H_H4].bias_off, 0.0f, 0.0f);
                dense_q<2, 0, 0, 2, CH(H_HF), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H5].bias_off, 0.0f, 0.0f);
                f32x4 o[NQ];
                dense_q_out<2, CH(AFTER_DEFORM)>(cx, st, B, o, Ly[H_HF].bias_off, true, 0.0f);
                if (g == 0) {
                    for_quarters([&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        stash(q)[3] = o[q][0];
                        stash(q)[4] = o[q][1];
                    });
                }
            }
            if (xw != nullptr && g == 0) {      // hand x', w to the fine pass (the lane that wrote the stash reads it back: no barrier needed)
                for_quarters([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    if (p_raw[q] < P) {
                        float *row = xw + ((p[q] / S) * (long)xw_row + xw_col0 + (p[q] % S)) * 8;
                        *reinterpret_cast<f32x4 *>(row) = f32x4{stash(q)[0], stash(q)[1], stash(q)[2], stash(q)[3]};
                        *reinterpret_cast<f32x4 *>(row + 4) = f32x4{stash(q)[4], 0.0f, 0.0f, 0.0f};
                    }
                });
            }
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (MODE == FIELD_DEFORM) continue;
        // radiance trunk (two 256-wide register sets A, B alternate; feat ends up in A)
        Blk A[8];
        f32x4 fin[NQ];
        {
            Blk B[8];
            {
                Blk in_tr[3];
                float xp[NQ][3], amb[NQ][3];
                for_quarters([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    xp[q][0] = stash(q)[0]; xp[q][1] = stash(q)[1]; xp[q][2] = stash(q)[2];
                    amb[q][0] = stash(q)[3]; amb[q][1] = stash(q)[4]; amb[q][2] = 0.0f;
                });
                pe_blocks_q<3, 10, 2>(xp, g, in_tr);
                pe_blocks_q<2, 4, 1>(amb, g, in_tr + 2);
                dense_q<2, 1, 0, 8, CH(H_T1), false>(cx, st, in_tr, in_tr + 2, nullptr, A, Ly[H_T0].bias_off, 0.01f, 0.0f);
            }
            dense_q<8, 0, 0, 8, CH(H_T2), true>(cx, st, A, nullptr, nullptr, B, Ly[H_T1].bias_off, 0.01f, 0.01f);
            dense_q<8, 0, 0, 8, CH(H_T3), true>(cx, st, B, nullptr, nullptr, A, Ly[H_T2].bias_off, 0.01f, 0.01f);
            {   // the re-injected encoding [PE(x') | PE(w)] is rebuilt at the skip layer instead of staying live
                Blk in_tr[3];
                float xp[NQ][3], amb[NQ][3];
                for_quarters([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    xp[q][0] = stash(q)[0]; xp[q][1] = stash(q)[1]; xp[q][2] = stash(q)[2];
                    amb[q][0] = stash(q)[3]; amb[q][1] = stash(q)[4]; amb[q][2] = 0.0f;
                });
                pe_blocks_q<3, 10, 2>(xp, g, in_tr);
                pe_blocks_q<2, 4, 1>(amb, g, in_tr + 2);
                dense_q<8, 2, 1, 8, CH(H_T4), true>(cx, st, A, in_tr, in_tr + 2, B, Ly[H_T3].bias_off, 0.01f, 0.01f);
            }
#pragma unroll 1
            for (int j = 0; j < 2; ++j) {     // T4, T5 | T6, T7 (identical shapes: one copy of the code, run twice)
                dense_q<8, 0, 0, 8, CH(H_T5), true>(cx, st, B, nullptr, nullptr, A, Ly[H_T4].bias_off + 512 * j, 0.01f, 0.01f);
                dense_q<8, 0, 0, 8, CH(H_T5), true>(cx, st, A, nullptr, nullptr, B, Ly[H_T4].bias_off + 512 * j + 256, 0.01f, 0.01f);
            }
            dense_q<8, 0, 0, 8, CH(H_ALPHA), true>(cx, st, B, nullptr, nullptr, A, Ly[H_FEAT].bias_off, 1.0f, 0.01f);
        }
        dense_q_out<8, CH(H_D0)>(cx, st, A, fin, Ly[H_ALPHA].bias_off, true, 1.0f);
        {   // colour branch
            Blk in_d[2];
            {
                float rdir[NQ][3];
                for_quarters([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    const float *rq = rays + (p[q] / S) * ray_stride;
                    rdir[q][0] = rq[3]; rdir[q][1] = rq[4]; rdir[q][2] = rq[5];
                });
                pe_blocks_q<3, 4, 1>(rdir, g, in_d);
                for_quarters([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    grid_block_q(grid, stash(q)[0], stash(q)[1], stash(q)[2], g, in_d[1].s[q]);
                });
            }
            Blk c[4], cn[4];
            dense_q<8, 1, 1, 4, CH(H_D1), false>(cx, st, A, in_d, in_d + 1, c, Ly[H_D0].bias_off, 0.01f, 0.0f);
            dense_q<4, 0, 0, 4, CH(H_D1), true>(cx, st, c, nullptr, nullptr, cn, Ly[H_D1].bias_off, 0.01f, 0.01f);
            dense_q<4, 0, 0, 4, CH(H_D1), true>(cx, st, cn, nullptr, nullptr, c, Ly[H_D1].bias_off + 128, 0.01f, 0.01f);
            dense_q<4, 0, 0, 4, CH(H_RGB), true>(cx, st, c, nullptr, nullptr, cn, Ly[H_D3].bias_off, 0.01f, 0.01f);
            dense_q_out<4, CH(H_S0)>(cx, st, cn, fin, 0, false, 0.01f);
        }
        {   // seg branch
            Blk s[4], sn[4];
            dense_q<8, 0, 0, 4, CH(H_S1), false>(cx, st, A, nullptr, nullptr, s, Ly[H_S0].bias_off, 0.01f, 0.0f);
            dense_q<4, 0, 0, 4, CH(H_S1), true>(cx, st, s, nullptr, nullptr, sn, Ly[H_S1].bias_off, 0.01f, 0.01f);
            dense_q<4, 0, 0, 4, CH(H_S1), true>(cx, st, sn, nullptr, nullptr, s, Ly[H_S1].bias_off + 128, 0.01f, 0.01f);
            dense_q<4, 0, 0, 4, CH(H_SEG), true>(cx, st, s, nullptr, nullptr, sn, Ly[H_S3].bias_off, 0.01f, 0.01f);
            dense_q_out<4, CH(AFTER_RADIANCE)>(cx, st, sn, fin, 0, false, 0.01f);
        }
        for_quarters([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if (p_raw[q] < P)    // rows 4g .. 4g+3 of [rgb3 | seg12 | sigma]
                *reinterpret_cast<f32x4 *>(raw + p_raw[q] * D_RAW + 4 * g) = fin[q];
        });
    }
}

}  // namespace hq
}  // namespace sahs

using namespace sahs;
using namespace sahs::hq;

template <int MODE>
static int launch_q(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride, const float *zvals,
                    float *raw, float *xw, int xw_row, int xw_col0, const int *src, int num_cu, hipStream_t stream)
{
    if (P <= 0) return 0;
    const long ntiles = (P + Q_PTS_PER_WG - 1) / Q_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_forward_bf16q_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    field_forward_bf16q_kernel<MODE><<<grid, Q_THREADS, LDS_BYTES, stream>>>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, xw, xw_row, xw_col0, src);
    return (int)hipGetLastError();
}

// mode as sahs_field_forward_f32_split_launch; mode 0 with xw == nullptr is the plain whole-network evaluation
extern "C" int sahs_field_forward_bf16q_split_launch(const float *packed, const float *frame, int level, int mode, long P, int S,
                                                     const float *rays, int ray_stride, const float *zvals, float *raw, float *xw, int xw_row,
                                                     int xw_col0, const int *src, int num_cu, hipStream_t stream)
{
    switch (mode) {
    case FIELD_ALL:
        return launch_q<FIELD_ALL>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, xw, xw_row, xw_col0, nullptr, num_cu, stream);
    case FIELD_DEFORM:
        return launch_q<FIELD_DEFORM>(packed, frame, level, P, S, rays, ray_stride, zvals, nullptr, xw, xw_row, xw_col0, nullptr, num_cu, stream);
    case FIELD_RADIANCE:
        return launch_q<FIELD_RADIANCE>(packed, frame, level, P, S, rays, ray_stride, nullptr, raw, xw, xw_row, 0, src, num_cu, stream);
    default:
        return -2;
    }
}
