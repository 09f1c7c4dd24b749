// field_bf16w.hip -- the bf16-MFMA field kernel re-tiled for ONE WAVE PER SIMD (BASELINE.json configs[2], round 2).
//
// Why: in field_bf16.hip (8 waves x 32 samples, 2 waves per SIMD, <= 256 VGPRs) a wave's instruction stream per weight chunk is
// its MFMAs PLUS its LDS-DMA issue, its activation repack (VALU) and its A-fragment reads, back to back; the two waves of a SIMD
// run that same stream in lock step, so the launch time is the SUM of the matrix time and of everything else (ablations, DESIGN.md
// section 3.1b), and a lone wave per SIMD already does 78 % of the pair's work (tools/ablate.py "w4").  Here a workgroup is 4 waves,
// one per SIMD, each owning 64 samples (two 32-sample halves that share every A fragment) and the whole 512-register file:
//   * activations of both halves stay in registers as the next layer's B operand, exactly as before (same packed stream);
//   * every A fragment (one ds_read_b128) feeds TWO MFMAs -- half the LDS read traffic per FLOP;
//   * A fragments are read by volatile asm AP fragments ahead with a counted lgkmcnt (the compiler sinks plain loads next to
//     their use and drains with lgkmcnt(0));
//   * the bias of a tile is the C operand of its first MFMA (no bias add in the repack), read one tile ahead;
//   * TWO accumulator sets: while tile t accumulates into one, the finished tile t-1 is converted (activation + bf16 pack) from the
//     other in small pieces placed in the MFMA gaps of tile t -- the VALU work rides under the matrix pipe instead of after it;
//   * the 16 LDS-DMA pieces of the next chunk are spread over the chunk's steps.
// Numerics: the products, k order and bf16 roundings of field_bf16.hip; the bias enters as the chain's initial value instead of being
// added to the finished sum (one fp32 rounding placed differently).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <utility>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"
#include "bf16_pipe.hpp"

namespace SAHS_NS {
namespace hw {
using namespace hb;      // the bf16 layer program, chunking and packed layout of sahs_layout.hpp
using namespace bfp;     // vector types, hand-issued LDS reads + retiring waits, bias helpers, the LDS-DMA chunk context (bf16_pipe.hpp)

enum { FIELD_ALL = 0, FIELD_DEFORM = 1, FIELD_RADIANCE = 2 };
constexpr int KX32 = (KB_XYZ + 1) / 2, KA32 = (KB_AMB + 1) / 2;     // 32-feature blocks of PE(x') and PE(w): 2, 1 | 3, 1 (NeRFaceModel)      // the kernel's MODE (same values as field_f32.hip / SAHS_FIELD_*)
constexpr int NH = 2;                              // 32-sample halves per wave (1 also builds: the tick / slot maps below are written over NH)
constexpr int NV = 16 * NH;                        // accumulator values of an output tile per lane
struct Blk { u32x4 s[NH][2]; };                    // 32 features of this lane's two samples: [half][k-step] bf16x8 fragments (as dwords)

constexpr int W_THREADS = 256;
constexpr int W_PTS_PER_WAVE = 32 * NH;
constexpr int W_PTS_PER_WG = (W_THREADS / WAVE) * W_PTS_PER_WAVE;     // 256
constexpr int LDS_BUF_BYTES = CHUNK_HW_MAX * 2;                       // 64 KB each, two of them
constexpr int LDS_BIAS_BYTE_OFF = 2 * LDS_BUF_BYTES;
constexpr int LDS_STASH_BYTE_OFF = LDS_BIAS_BYTE_OFF + ((BIAS_FLOATS + 3) / 4) * 16;
constexpr int STASH_FLOATS = 8;                                       // per sample: x'[3], w[2] (+pad)
constexpr int LDS_BYTES = LDS_STASH_BYTE_OFF + W_PTS_PER_WG * STASH_FLOATS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(true, "");
constexpr int DBG_STRIDE_H = 56;
constexpr int AP = 4;                                                  // A fragments in flight ahead of the MFMAs that consume them
constexpr int PIECE_HW = W_THREADS * 8;                                // one LDS-DMA piece: 4 KB = 2048 halfwords (1 KB per wave)
constexpr int DMA_PIECES = LDS_BUF_BYTES / (W_THREADS * 16);           // 16

// run f(integral_constant<int, q>) for q = 0 .. NH-1 with q a compile-time constant: per-half state lives in small arrays that must
// only ever be indexed by constants (a run-time index sends them to scratch and turns tanhf into an out-of-line call)
template <class F>
__device__ __forceinline__ void for_halves(F &&f)
{
    [&]<int... Qs>(std::integer_sequence<int, Qs...>) { (f(std::integral_constant<int, Qs>{}), ...); }(std::make_integer_sequence<int, NH>{});
}

constexpr bool kNoDma = false;
constexpr bool kNoBarrier = false;
struct Ctx : PipeCtx<W_THREADS, LDS_BUF_BYTES, LDS_BIAS_BYTE_OFF, kNoDma, kNoBarrier> {
#if defined(SAHS_DIAG) && defined(SAHS_STAMP_W)
    // diagnostic build only (tools/stamp_bf16w.py): s_memtime stamps of wave 0 of workgroup 0 for one sample tile, written to the dbg
    // buffer (which the normal dbg writes then leave alone); no output value is computed from them
    unsigned long long *stamps; int sidx; bool stamp_on;
    __device__ __forceinline__ void stamp()
    {
        if (stamp_on) { if (lane == 0) stamps[sidx] = __builtin_amdgcn_s_memtime(); ++sidx; }
    }
#else
    __device__ __forceinline__ void stamp() {}
#endif
};

// ---- activation + bf16 conversion of a finished accumulator tile, software-pipelined -------------------------------------------------
// Beside a wave's own MFMAs, INDEPENDENT VALU instructions are nearly free (about six per MFMA), but a dependent chain is not: v_mul ->
// v_max -> v_cvt_pk issued back to back stalls the wave's in-order issue on every result, and the next MFMA with it (tools/micro/
// mfma_valu_overlap2.hip: one such unit per MFMA costs 15 %, two 74 %).  The 32 values of a tile (half U&1 .. see unit_of) therefore go
// through a three-stage pipeline, one "tick" at a time, so that the instructions of one tick never depend on each other:
//   tick T:   A(T)    m = v_{T,T+1} * slope                 (leaky only; T even, one packed multiply per pair)
//             B(T-1)  r = max(v_{T-1}, m)    | max(v, 0) for relu | v for no activation
//             C(T-2)  dword = cvt_pk_bf16(r_{T-3}, r_{T-2})   when T-2 is odd
// 36 ticks convert a tile; they are dealt out over the MFMA slots of the following tile.
struct PackState { f32x2 m2[2]; float r[4]; uint32_t d[2]; };
// value U (0..31): pair P = U>>1 (half P&1, accumulator registers 2(P>>1) + (U&1)) -> dword (P>>1)&3 of fragment [half][(P>>1)>>2]
// ReLU layers (slope 0) take a shorter route: round first, clamp after -- bf16(max(v,0)) == max(bf16(v),0), and on bf16 BIT PATTERNS
// max(.,0) is a signed 16-bit integer max (negative floats are negative integers), so one v_pk_max_i16 clamps a converted PAIR:
// 1 instruction per value instead of 1.5 (two v_max_f32 + one v_cvt_pk per pair).
//   tick T:   C'(T-2)  d = cvt_pk_bf16(v_{T-3}, v_{T-2})        when T-2 is odd
//             D'(T-4)  dword = pk_max_i16(d, 0)                 when T-4 is odd
template <int T>
__device__ __forceinline__ void pack_tick(const f32x16 (&acc)[NH], Blk &o, float slope, PackState &ps)
{
    const bool exact = slope < 0.0f;       // (-0.01: the reference's fp32 LeakyReLU, see the kernel's EXACT)
    if (exact) slope = -slope;
    // LeakyReLU on the packed bf16 BIT PATTERNS, after rounding: 1.5 VALU instructions per value instead of the 2.5 of multiply + max +
    // half a convert in fp32 (build with -DSAHS_BF16W_EXACT_LEAKY for that form; tools/ablate.py "wexact").  The conversion work beside
    // the MFMAs is what this kernel is bound by, and this is worth 15 % of a launch.  A negative bf16 value has
    // its sign bit set, i.e. it is a negative int16 whose low 15 bits grow with the magnitude; subtracting K = round(128 log2(1/slope)) =
    // 850 from the pattern, saturating at int16 min = -0.0, scales the magnitude by 2^-(K/128) read piecewise-linearly over the mantissa:
    // the slope applied is 0.0095 .. 0.0106 (depending on the mantissa) where the reference's is 0.01.  That is an error of <= 6e-4 |x| on
    // the negative side -- below the rounding step of bf16 itself, 2e-3 |x|, which every positive value already carries -- but a
    // systematic one: this precision's contract is the PSNR bound of BASELINE.json (0.05 dB), not the activation function bit for bit,
    // and bench.py / tests/test_gpu_bf16.py measure it with this arithmetic.  Positive values pass unchanged:
    //   tick T:  C'(T-2) d = cvt_pk_bf16(v_{T-3}, v_{T-2});  D'(T-3) s = d >> 15 (arithmetic, per half: 0 | -1);  E'(T-4) dword = sat_i16(s * K + d)
    if (!exact && slope != 0.0f && slope != 1.0f) {
        if constexpr (T - 2 >= 1 && T - 2 < NV && ((T - 2) & 1)) {
            constexpr int U = T - 2, P = U >> 1, hh = P % NH, q = P / NH;
            ps.d[P & 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{acc[hh][2 * q], acc[hh][2 * q + 1]}, bf16x2));
        }
        if constexpr (T - 3 >= 1 && T - 3 < NV && ((T - 3) & 1)) {
            constexpr int U = T - 3, P = U >> 1;
            uint32_t sg;
            asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(sg) : "v"(ps.d[P & 1]));      // (the inline constant sits in the low half only)
            ps.r[P & 1] = __builtin_bit_cast(float, sg);
        }
        if constexpr (T - 4 >= 1 && T - 4 < NV && ((T - 4) & 1)) {
            constexpr int U = T - 4, P = U >> 1, hh = P % NH, q = P / NH, s = q >> 2, jp = q & 3;
            const uint32_t kk = 0x03520352u;      // K = 850 = round(128 * log2(100)) in both halves
            uint32_t o_;
            asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(o_) : "v"(__builtin_bit_cast(uint32_t, ps.r[P & 1])), "s"(kk), "v"(ps.d[P & 1]));
            o.s[hh][s][jp] = o_;
        }
        return;
    }
    if (slope == 0.0f) {
        if constexpr (T - 2 >= 1 && T - 2 < NV && ((T - 2) & 1)) {
            constexpr int U = T - 2, P = U >> 1, hh = P % NH, q = P / NH;
            ps.d[P & 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{acc[hh][2 * q], acc[hh][2 * q + 1]}, bf16x2));
        }
        if constexpr (T - 4 >= 1 && T - 4 < NV && ((T - 4) & 1)) {
            constexpr int U = T - 4, P = U >> 1, hh = P % NH, q = P / NH, s = q >> 2, jp = q & 3;
            o.s[hh][s][jp] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, ps.d[P & 1]), s16x2{0, 0}));
        }
        return;
    }
    if constexpr (T >= 0 && T < NV && !(T & 1)) {           // A(T), T even: both values of the pair
        constexpr int P = T >> 1, hh = P % NH, q = P / NH;
        if (slope != 0.0f && slope != 1.0f) {
            const f32x2 pr = f32x2{acc[hh][2 * q], acc[hh][2 * q + 1]}, sl = f32x2{slope, slope};
            ps.m2[P & 1] = pr * sl;      // the compiler scalarises this into two v_mul_f32
        }
    }
    constexpr int DB = 1, DC = 2;       // stage distances in ticks (2 and 4 measured: no change, 46.1-48.2 vs 46.9-48.0 cycles per MFMA)
    if constexpr (T - DB >= 0 && T - DB < NV) {             // B(T-1)
        constexpr int U = T - DB, P = U >> 1, hh = P % NH, q = P / NH, e = U & 1;
        const float v = acc[hh][2 * q + e];
        ps.r[U & 3] = slope == 1.0f ? v : (slope == 0.0f ? fmaxf(v, 0.0f) : fmaxf(v, ps.m2[P & 1][e]));
    }
    if constexpr (T - DC >= 1 && T - DC < NV && ((T - DC) & 1)) {    // C(T-2): the pair (T-3, T-2) is complete
        constexpr int U = T - DC, P = U >> 1, hh = P % NH, q = P / NH, s = q >> 2, jp = q & 3;
        o.s[hh][s][jp] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{ps.r[(U - 1) & 3], ps.r[U & 3]}, bf16x2));
    }
}
constexpr int PACK_TICKS = NV + 4;
template <int LO, int HI>
__device__ __forceinline__ void pack_ticks(const f32x16 (&acc)[NH], Blk &o, float slope, PackState &ps)
{
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (pack_tick<LO + Is>(acc, o, slope, ps), ...); }(std::make_integer_sequence<int, HI - LO>{});
}

// ---------------------------------------------------------------------------------------------------------------------------------
// hidden layer: NT32 output tiles of 32 rows for both halves.  Per chunk: TOTAL = G*STEPS A fragments, two MFMAs each.
// LGKM bookkeeping (all reads are asm, in issue order): step I issues  [wait] [2 MFMAs] [bias batch of the next tile: 4 reads, when I
// starts a tile that has a successor] [A read I+AP]; the wait of step I allows exactly the reads issued after A(I).  The batch comes
// BEFORE the step's A read, so the wait that retires A(first step of the next tile) -- issued at least AP <= STEPS steps later --
// retires the batch too (simulated for every layer shape: tools/check_bf16w_sched.py).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int STEPS, int TOTAL, int NT32, int T0>      // T0: index (within the layer) of the chunk's first tile
struct Sched {
    static constexpr bool bias_at(int s) { return s >= 0 && s < TOTAL && (s % STEPS == 0) && (T0 + s / STEPS + 1 < NT32); }
    static constexpr bool aread_at(int s) { return s >= 0 && s + AP < TOTAL; }
    static constexpr int cnt(int I)
    {
        int n = 0, lo = 0;
        if (I < AP) n += ((AP < TOTAL ? AP : TOTAL) - 1 - I);    // read in the prologue: the later prologue reads, then steps 0..I-1
        else lo = I - AP + 1;                                   // read at the END of step I-AP (after that step's bias batch)
        for (int s = lo; s < I; ++s) n += (aread_at(s) ? 1 : 0) + (bias_at(s) ? 4 : 0);
        return n;
    }
};

// The two accumulator sets and the conversion state live across layers: a layer ends with its LAST tile still unconverted in set 1
// (every layer has an even number of tiles), and the NEXT layer -- whose first input segment in0 is that layer's output -- converts it
// into in0[K0-1] under its own first MFMAs, finishing before the step that first reads that block (k = 2 (K0 - 1)).
struct St {
    f32x16 acc[2][NH];
    PackState ps;        // the conversion pipeline's registers
};

// PEND: the previous layer left its last tile in st.acc[1]; it belongs in in0[K0-1] with activation slope pslope
template <int K0, int K1, int K2, int NT32, int NEXT_HW, bool PEND>
__device__ __forceinline__ void dense_w(Ctx &cx, St &st, Blk *in0, const Blk *in1, const Blk *in2, Blk *out, int bias_off, float slope, float pslope)
{
    constexpr int KB = K0 + K1 + K2;
    constexpr int G = pick_G32(KB, NT32);
    constexpr int STEPS = KB * 2, TOTAL = G * STEPS, NCH = NT32 / G;
    static_assert(NT32 % 2 == 0, "the last tile of a layer must land in accumulator set 1");
    const uint32_t baddr = cx.bias_addr + 4u * (uint32_t)bias_off;
    f32x4 braw[2][4];                                         // raw bias reads of the current / next tile
    bias_read<0>(braw[0], baddr);                             // tile 0 (older than every A read below: any wait that retires A(0) retires it)

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
        // One step = one A fragment = two MFMAs (half 0, half 1).  Order inside a step, pinned by scheduling fences:
        //     wait | MFMA half 0 | conversion work, bias reads | MFMA half 1 | conversion work, A read, LDS-DMA piece
        // Plain VALU work placed behind an MFMA rides under it (up to ~6 instructions per MFMA are free: tools/micro/
        // mfma_valu_overlap.hip); two MFMAs back to back would block the wave's issue -- and the VALU work queued behind them -- until
        // the pipe is free.  The 32 accumulator values of the finished tile t-1 are spread over the 2 STEPS - 2 slots behind the MFMAs of
        // steps 1..STEPS-1; the tile carried over from the previous layer over the slots of steps 1..2 K0 - 3 of tile 0.
        auto step = [&]<int I>() {
            constexpr int g = I / STEPS, k = I % STEPS, b = k >> 1, st_ = k & 1, t = T0 + g, set = t & 1;
            const Blk &x = (b < K0) ? in0[b] : ((b < K0 + K1) ? in1[b - K0] : in2[b - K0 - K1]);
            // retires A(I) and, at a tile's first step, the tile's bias batch (issued before A(I): Sched)
            if constexpr (k == 0) wait_retire<S::cnt(I)>(a[I % AP], braw[set]);
            else wait_retire<S::cnt(I)>(a[I % AP]);
            fence();
            f32x16 cb;
            if constexpr (k == 0) cb = bias_as_c(braw[set]);      // the tile's chains start from its bias
            for_halves([&](auto Q) {
                constexpr int hh = decltype(Q)::value;
                if constexpr (k == 0) st.acc[set][hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[I % AP]), frag(x.s[hh][st_]), cb, 0, 0, 0);
                else st.acc[set][hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[I % AP]), frag(x.s[hh][st_]), st.acc[set][hh], 0, 0, 0);
                fence();
                if constexpr (hh == 0) {
                    if constexpr (S::bias_at(I)) bias_read<128 * (t + 1)>(braw[set ^ 1], baddr);
                }
                if constexpr (hh == NH - 1) {
                    if constexpr (S::aread_at(I)) lds_read16<(I + AP) * 1024>(a[I % AP], abase);
                    if constexpr (I % PSTEP == 0 && I / PSTEP < npieces) cx.issue_piece(I / PSTEP);
                }
                if constexpr (t > 0 && k >= 1) {              // the finished tile t-1 -> out[t-1]: this slot's share of its conversion ticks
                    constexpr int NSLOT = NH * (STEPS - 1), slot = NH * (k - 1) + hh;
                    constexpr int lo = PACK_TICKS * slot / NSLOT, hi = PACK_TICKS * (slot + 1) / NSLOT;
                    if constexpr (hi > lo) pack_ticks<lo, hi>(st.acc[set ^ 1], out[t - 1], slope, st.ps);
                } else if constexpr (PEND && t == 0 && k >= 1 && k <= 2 * K0 - 3) {     // the previous layer's last tile -> in0[K0-1]
                    constexpr int NSLOT = NH * (2 * K0 - 3), slot = NH * (k - 1) + hh;
                    constexpr int lo = PACK_TICKS * slot / NSLOT, hi = PACK_TICKS * (slot + 1) / NSLOT;
                    if constexpr (hi > lo) pack_ticks<lo, hi>(st.acc[1], in0[K0 - 1], pslope, st.ps);
                }
                fence();
            });
        };
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (step.template operator()<Is>(), ...); }(std::make_integer_sequence<int, TOTAL>{});
        // pieces that did not get a step (short chunks in front of long ones)
        [&]<int... Ps>(std::integer_sequence<int, Ps...>) {
            ((Ps >= (TOTAL + PSTEP - 1) / PSTEP && Ps < npieces ? cx.issue_piece(Ps) : (void)0), ...);
        }(std::make_integer_sequence<int, DMA_PIECES>{});
        cx.end_chunk();
    };
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) { (chunk.template operator()<Cs>(), ...); }(std::make_integer_sequence<int, NCH>{});
    // the layer's last tile stays in st.acc[1]: the next layer converts it (PEND)
}

// 16-row output layer (WF, HF; ALPHA -> RGB -> SEG into one tile) accumulated in fp32: first = start from the bias, else from the running
// tile.  Its input in0 is always the previous layer's output: that layer's last tile (st.acc[1]) is converted into in0[K0-1] here.
template <int K0, int NEXT_HW>
__device__ __forceinline__ void dense_w_out(Ctx &cx, St &st, Blk *in0, f32x16 (&acc)[NH], int bias_off, bool first, float pslope)
{
    constexpr int TOTAL = K0 * 2;
    constexpr int npieces = (NEXT_HW + PIECE_HW - 1) / PIECE_HW;
    cx.begin_chunk(NEXT_HW);
    const uint32_t abase = cx.cur_addr();
    if (first) {       // rows 0..15 of the 32-row tile carry the layer's bias (registers 0..7), rows 16..31 are unused
        const uint32_t baddr = cx.bias_addr + 4u * (uint32_t)bias_off;
        f32x4 t0, t1;
        lds_read16<0>(t0, baddr);
        lds_read16<32>(t1, baddr);
        wait_retire<0>(t0, t1);
        fence();
        for_halves([&](auto Q) {
            constexpr int hh = decltype(Q)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[hh][r] = t0[r]; acc[hh][4 + r] = t1[r]; }
#pragma unroll
            for (int r = 8; r < 16; ++r) acc[hh][r] = 0.0f;
        });
    }
    u32x4 a[AP];
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (lds_read16<Is * 1024>(a[Is], abase), ...); }(std::make_integer_sequence<int, (AP < TOTAL ? AP : TOTAL)>{});
    fence();
    auto step = [&]<int I>() {
        wait_retire<((TOTAL - 1 - I) < (AP - 1) ? (TOTAL - 1 - I) : (AP - 1))>(a[I % AP]);
        fence();
        for_halves([&](auto Q) {
            constexpr int hh = decltype(Q)::value;
            acc[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a[I % AP]), frag(in0[I >> 1].s[hh][I & 1]), acc[hh], 0, 0, 0);
            fence();
            if constexpr (hh == NH - 1) {
                if constexpr (I + AP < TOTAL) lds_read16<(I + AP) * 1024>(a[I % AP], abase);
                if constexpr (I < npieces) cx.issue_piece(I);
            }
            if constexpr (I >= 1 && I <= 2 * K0 - 3) {        // the previous layer's last tile -> in0[K0-1], before step 2 (K0 - 1) reads it
                constexpr int NSLOT = NH * (2 * K0 - 3), slot = NH * (I - 1) + hh;
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

// ---- positional encoding (v_sin_f32 in revolutions, as field_bf16.hip) for both halves -------------------------------------------
struct PeSlot { float scale; float phase; int axis; int kind; };   // kind: 0 zero pad, 1 raw input, 2 sinusoid
template <int D, int L, int INC = 1>      // INC: the encoding starts with its input (nerf_helpers.py:305-349 include_input)
constexpr PeSlot pe_slot(int f)
{
    constexpr int W = INC * D + 2 * D * L;
    if (f >= W) return PeSlot{0.0f, 0.0f, 0, 0};
    if (INC && f < D) return PeSlot{1.0f, 0.0f, f, 1};
    const int g = f - INC * D, k = g / (2 * D), rem = g % (2 * D);
    return PeSlot{(float)(1 << k), (rem / D) ? 0.25f : 0.0f, rem % D, 2};
}

template <int D, int L, int NB, int INC = 1>
__device__ __forceinline__ void pe_blocks_w(const float (*v)[3], int h, Blk *out)
{
    for_halves([&](auto Q) {
        constexpr int hh = decltype(Q)::value;
        float rev[3];
#pragma unroll
        for (int i = 0; i < D; ++i) rev[i] = v[hh][i] * 0.15915494309189535f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f0 = 32 * b + 16 * s + 8 * (j >> 2) + (j & 3);
                    const PeSlot a = pe_slot<D, L, INC>(f0), c = pe_slot<D, L, INC>(f0 + 4);   // lane half 0 / 1
                    if (a.kind == 0 && c.kind == 0) {
                        r[j] = 0.0f;
                    } else {
                        const float xa = (a.kind == 1) ? v[hh][a.axis] : rev[a.axis], xc = (c.kind == 1) ? v[hh][c.axis] : rev[c.axis];
                        const float x = h ? xc : xa;
                        const float t = x * (h ? c.scale : a.scale) + (h ? c.phase : a.phase);
                        const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t));
                        const int kind = h ? c.kind : a.kind;
                        r[j] = (kind == 2) ? sn : ((kind == 1) ? x : 0.0f);
                    }
                }
#pragma unroll
                for (int jp = 0; jp < 4; ++jp)
                    out[b].s[hh][s][jp] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r[2 * jp], r[2 * jp + 1]}, bf16x2));
            }
    });
}

// trilinear lookup (fp32, ATen corner order, zeros padding) for one half; this lane takes channels 16s + 8g + 4h + 0..3
__device__ __forceinline__ void grid_block_w(const float *__restrict__ grid, float x, float y, float z, int h, u32x4 (&out)[2], float *dbg)
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
        const float we = inb ? wt : 0.0f;        // branch-free zeros padding: all 32 loads in flight at once
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
        for (int jp = 0; jp < 4; ++jp) {
            const int j = 2 * jp;
            out[s][jp] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a[2 * s + (j >> 2)][j & 3], a[2 * s + (j >> 2)][(j & 3) + 1]}, bf16x2));
        }
    if (dbg != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4 *>(dbg + 8 * k + 4 * h) = a[k];
    }
}

#define CH(id) (kProgH.layer[id].G32 * kProgH.layer[id].KB32 * 1024)   /* halfwords in one chunk of layer id */

// MODE (the split evaluation of field_f32.hip, same contract): FIELD_ALL = whole network, additionally writing x', w of every sample
// to xw when xw != nullptr; FIELD_DEFORM = warp + hyper nets only, on the depths zvals, results to xw columns xw_col0..; FIELD_RADIANCE =
// radiance nets only, (x', w) of sample p read from xw column src[p].  xw: (rays, xw_row, 8) floats [x'0 x'1 x'2 w0 w1 . . .].
// EXACT: LeakyReLU as the reference's max(v, 0.01 v) in fp32 (2.5 VALU instructions per value) instead of the packed-integer form on the bf16
// bit patterns (1.5; slope 0.0095 .. 0.0106): selected at run time (sahs_bf16_exact_leaky / SAHS_BF16_EXACT_LEAKY=1), e.g. to A/B a real
// checkpoint.  The kernel hands pack_tick the slope as -0.01 for "exact" (a compile-time constant either way).
template <int MODE, bool EXACT>
__global__ void __launch_bounds__(W_THREADS, 1)
field_forward_bf16w_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                           const float *__restrict__ rays, int ray_stride, const float *__restrict__ zvals,
                           float *__restrict__ raw, float *__restrict__ dbg, float *__restrict__ xw, int xw_row, int xw_col0,
                           const int *__restrict__ src)
{
    constexpr uint32_t RAD_OFF = (uint32_t)kProgH.layer[H_T0].stream_off;      // first radiance-net chunk
    constexpr float LK = EXACT ? -0.01f : 0.01f;                                // NeRFMLP's LeakyReLU (modules.py:252), see pack_tick
#if SAHS_MODEL == 2      // no deformation nets (config/expression/person_1.yml): the whole network is the radiance net, queried at the raw point
    static_assert(MODE == FIELD_ALL, "this model has no deformation nets to split off");
    constexpr int L_FIRST = H_T0, AFTER_RADIANCE = H_T0;
#else
    constexpr int L_FIRST = MODE == FIELD_RADIANCE ? H_T0 : H_W0;               // the launch's first layer
    constexpr int AFTER_DEFORM = MODE == FIELD_DEFORM ? H_W0 : H_T0;            // the chunk prefetched under the last deformation layer
    constexpr int AFTER_RADIANCE = MODE == FIELD_RADIANCE ? H_T0 : H_W0;        // ... under the last radiance layer
#endif
    extern __shared__ __attribute__((aligned(16))) char lds_w[];
    Ctx cx;
    cx.stream = reinterpret_cast<const unsigned short *>(packed + PACKH_STREAM_OFF) + (long)level * STREAM_HW;
    cx.lds = lds_w;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.h = cx.lane >> 5;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *grid = packed + PACKH_GRID_OFF;
    const int h = cx.h, col = cx.lane & 31;
    {
        const float *bsrc = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        float *bl = reinterpret_cast<float *>(lds_w + LDS_BIAS_BYTE_OFF);
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += W_THREADS) bl[i] = bsrc[i];
        cx.wrap_at = MODE == FIELD_DEFORM ? RAD_OFF : (uint32_t)STREAM_HW;
        cx.wrap_to = MODE == FIELD_RADIANCE ? RAD_OFF : 0u;
        cx.off = cx.wrap_to;
        cx.prepare(CH(L_FIRST), 0);
#pragma unroll
        for (int pc = 0; pc < (CH(L_FIRST) + PIECE_HW - 1) / PIECE_HW; ++pc) cx.issue_piece(pc);
        __syncthreads();
    }
    constexpr const LayerH *Ly = kProgH.layer;

    const long ntiles = (P + W_PTS_PER_WG - 1) / W_PTS_PER_WG;
#if defined(SAHS_DIAG) && defined(SAHS_STAMP_W)
    unsigned long long *stamp_base = reinterpret_cast<unsigned long long *>(dbg);
    dbg = nullptr;
    int tile_no = 0;
#endif
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#if defined(SAHS_DIAG) && defined(SAHS_STAMP_W)
        cx.stamp_on = stamp_base != nullptr && blockIdx.x == 0 && tile_no == 3 && cx.wave == 0;
        cx.stamps = stamp_base;
        cx.sidx = 0;
        ++tile_no;
#endif
        cx.stamp();                                   // 0 tile start
        cx.refresh_bias_base();
        // the chunk sequence is static, so every chunk's source address is loop-invariant: without this the compiler hoists all ~120 of
        // them out of the persistent loop and spills them (960 bytes of scratch in the NeRFaceModel build)
        asm volatile("" : "+s"(cx.off));
        St st;
        // this lane's two samples: halves hh = 0, 1 -> sample (wave*64 + hh*32 + col) of the workgroup tile
        long p_raw[NH], p[NH];
        float x[NH][3];
        // x' (3) and w (2) of a sample are parked in LDS between their uses; sample (wave*64 + 32*half + col) of the workgroup tile
        typedef __attribute__((address_space(3))) float *lds_float;
        const lds_float stash0 = (lds_float)(__attribute__((address_space(3))) char *)(lds_w + LDS_STASH_BYTE_OFF) + (cx.wave * W_PTS_PER_WAVE + col) * STASH_FLOATS;
        auto stash = [&](int hh) { return stash0 + hh * 32 * STASH_FLOATS; };
        for_halves([&](auto Q) {
            constexpr int hh = decltype(Q)::value;
            p_raw[hh] = tile * W_PTS_PER_WG + cx.wave * W_PTS_PER_WAVE + hh * 32 + col;
            p[hh] = p_raw[hh] < P ? p_raw[hh] : P - 1;
            if constexpr (MODE != FIELD_RADIANCE) {
                const float *rp = rays + (p[hh] / S) * ray_stride;
                const float z = zvals[p[hh]];
#pragma unroll
                for (int i = 0; i < 3; ++i) x[hh][i] = rp[i] + rp[3 + i] * z;
            }
        });
        if constexpr (MODE == FIELD_RADIANCE) {     // x', w come from the launches that deformed these samples
            if (h == 0) {
                for_halves([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    const float *row = xw + ((p[q] / S) * (long)xw_row + (src != nullptr ? src[p[q]] : (int)(p[q] % S))) * 8;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(row);
                    stash(q)[0] = v[0]; stash(q)[1] = v[1]; stash(q)[2] = v[2]; stash(q)[3] = v[3];
                    stash(q)[4] = row[4];
                });
            }
        } else {
#if SAHS_MODEL == 2
        if (h == 0) {
            for_halves([&](auto Q) {      // models.py:316-327 with use_warp False, use_ambient False: x' = x, no ambient coordinate
                constexpr int q = decltype(Q)::value;
                stash(q)[0] = x[q][0]; stash(q)[1] = x[q][1]; stash(q)[2] = x[q][2]; stash(q)[3] = 0.0f; stash(q)[4] = 0.0f;
            });
        }
#else
        Blk pe_x[2];
        pe_blocks_w<3, 10, 2>(x, h, pe_x);
        cx.stamp();                                   // 1 sample points + PE(x)
        {   // warp field (layers alternate between two register sets: no activation copies)
            Blk A[4], B[4];
            dense_w<2, 0, 0, 4, CH(H_W1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_W0].bias_off, 0.0f, 0.0f);
            dense_w<4, 0, 0, 4, CH(H_W1), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W1].bias_off, 0.0f, 0.0f);
            dense_w<4, 0, 0, 4, CH(H_W1), true>(cx, st, B, nullptr, nullptr, A, Ly[H_W1].bias_off + 128, 0.0f, 0.0f);
            dense_w<4, 0, 0, 4, CH(H_W4), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W3].bias_off, 0.0f, 0.0f);
            dense_w<4, 2, 0, 4, CH(H_W5), true>(cx, st, B, pe_x, nullptr, A, Ly[H_W4].bias_off, 0.0f, 0.0f);
            dense_w<4, 0, 0, 4, CH(H_WF), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W5].bias_off, 0.0f, 0.0f);
            f32x16 o[NH];
            dense_w_out<4, CH(H_H0)>(cx, st, B, o, Ly[H_WF].bias_off, true, 0.0f);
            if (h == 0) {
                for_halves([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    stash(q)[0] = x[q][0] + tanhf(o[q][0]);            // models.py:305 (rows 0..2 live in lane half 0)
                    stash(q)[1] = x[q][1] + tanhf(o[q][1]);
                    stash(q)[2] = x[q][2] + tanhf(o[q][2]);
                });
            }
        }
        cx.stamp();                                   // 2 warp net
        {   // hyper sheet
            Blk A[2], B[2];
            dense_w<2, 0, 0, 2, CH(H_H1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_H0].bias_off, 0.0f, 0.0f);
            dense_w<2, 0, 0, 2, CH(H_H1), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H1].bias_off, 0.0f, 0.0f);
            dense_w<2, 0, 0, 2, CH(H_H1), true>(cx, st, B, nullptr, nullptr, A, Ly[H_H1].bias_off + 64, 0.0f, 0.0f);
            dense_w<2, 0, 0, 2, CH(H_H4), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H3].bias_off, 0.0f, 0.0f);
            dense_w<2, 2, 0, 2, CH(H_H5), true>(cx, st, B, pe_x, nullptr, A, Ly[H_H4].bias_off, 0.0f, 0.0f);
            dense_w<2, 0, 0, 2, CH(H_HF), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H5].bias_off, 0.0f, 0.0f);
            f32x16 o[NH];
            dense_w_out<2, CH(AFTER_DEFORM)>(cx, st, B, o, Ly[H_HF].bias_off, true, 0.0f);
            if (h == 0) {
                for_halves([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    stash(q)[3] = o[q][0];
                    stash(q)[4] = o[q][1];
                });
            }
        }
        if (xw != nullptr && h == 0) {      // hand x', w to the fine pass (the lane that wrote the stash reads it back: no barrier needed)
            for_halves([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if (p_raw[q] < P) {
                    float *row = xw + ((p[q] / S) * (long)xw_row + xw_col0 + (p[q] % S)) * 8;
                    *reinterpret_cast<f32x4 *>(row) = f32x4{stash(q)[0], stash(q)[1], stash(q)[2], stash(q)[3]};
                    *reinterpret_cast<f32x4 *>(row + 4) = f32x4{stash(q)[4], 0.0f, 0.0f, 0.0f};
                }
            });
        }
#endif
        }
        __builtin_amdgcn_wave_barrier();
        cx.stamp();                                   // 3 hyper net (RADIANCE: x', w fetched)
        if (dbg != nullptr && h == 0 && MODE != FIELD_RADIANCE) {
            for_halves([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if (p_raw[q] < P) {
                    float *dsl = dbg + p[q] * DBG_STRIDE_H;
                    dsl[0] = stash(q)[0] - x[q][0]; dsl[1] = stash(q)[1] - x[q][1]; dsl[2] = stash(q)[2] - x[q][2];
                    dsl[3] = stash(q)[3]; dsl[4] = stash(q)[4];
                }
            });
        }
        if constexpr (MODE == FIELD_DEFORM) continue;
        // radiance trunk (two 256-wide register sets A, B alternate; feat ends up in A)
        Blk A[8];
        f32x16 fin[NH];
        {
            Blk B[8];
            {
                Blk in_tr[KX32 + KA32];
                float xw[NH][3], amb[NH][3];
                for_halves([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    xw[q][0] = stash(q)[0]; xw[q][1] = stash(q)[1]; xw[q][2] = stash(q)[2];
                    amb[q][0] = stash(q)[3]; amb[q][1] = stash(q)[4]; amb[q][2] = 0.0f;
                });
                pe_blocks_w<3, L_XYZ, KX32>(xw, h, in_tr);
                if constexpr (KA32 > 0) pe_blocks_w<(AMB_DIM > 0 ? AMB_DIM : 1), L_AMB, KA32, AMB_INC>(amb, h, in_tr + KX32);
                cx.stamp();                           // 4 PE(x'), PE(w)
                dense_w<KX32, KA32, 0, 8, CH(H_T1), false>(cx, st, in_tr, in_tr + KX32, nullptr, A, Ly[H_T0].bias_off, LK, 0.0f);
                cx.stamp();                           // 5 T0
            }
            dense_w<8, 0, 0, 8, CH(H_T2), true>(cx, st, A, nullptr, nullptr, B, Ly[H_T1].bias_off, LK, LK);
            cx.stamp();                               // 6 T1
            dense_w<8, 0, 0, 8, CH(H_T3), true>(cx, st, B, nullptr, nullptr, A, Ly[H_T2].bias_off, LK, LK);
            cx.stamp();                               // 7 T2
            {   // the re-injected encoding [PE(x') | PE(w)] is rebuilt at the skip layer instead of staying live
                Blk in_tr[KX32 + KA32];
                float xw[NH][3], amb[NH][3];
                for_halves([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    xw[q][0] = stash(q)[0]; xw[q][1] = stash(q)[1]; xw[q][2] = stash(q)[2];
                    amb[q][0] = stash(q)[3]; amb[q][1] = stash(q)[4]; amb[q][2] = 0.0f;
                });
                pe_blocks_w<3, L_XYZ, KX32>(xw, h, in_tr);
                if constexpr (KA32 > 0) pe_blocks_w<(AMB_DIM > 0 ? AMB_DIM : 1), L_AMB, KA32, AMB_INC>(amb, h, in_tr + KX32);
                cx.stamp();                           // 8 PE again
#if SAHS_MODEL == 0
                dense_w<8, KX32, KA32, 8, CH(H_T4), true>(cx, st, A, in_tr, in_tr + KX32, B, Ly[H_T3].bias_off, LK, LK);
#else           // NeRFaceModel: a 4-layer trunk, the skip layer is its last (modules.py:176)
                dense_w<8, KX32, KA32, 8, CH(H_FEAT), true>(cx, st, A, in_tr, in_tr + KX32, B, Ly[H_T3].bias_off, LK, LK);
#endif
                cx.stamp();                           // 9 T3 (skip)
            }
#if SAHS_MODEL == 0
#pragma unroll 1
            for (int j = 0; j < 2; ++j) {     // T4, T5 | T6, T7 (identical shapes: one copy of the code, run twice)
                dense_w<8, 0, 0, 8, CH(H_T5), true>(cx, st, B, nullptr, nullptr, A, Ly[H_T4].bias_off + 512 * j, LK, LK);
                dense_w<8, 0, 0, 8, CH(H_T5), true>(cx, st, A, nullptr, nullptr, B, Ly[H_T4].bias_off + 512 * j + 256, LK, LK);
#if defined(SAHS_DIAG) && defined(SAHS_STAMP_W)
                if (j == 0) cx.stamp();               // (diagnostic only) first pass through the loop body: instruction-cache cold
#endif
            }
#endif
            cx.stamp();                               // 10 T4..T7
            dense_w<8, 0, 0, 8, CH(H_ALPHA), true>(cx, st, B, nullptr, nullptr, A, Ly[H_FEAT].bias_off, 1.0f, LK);
            cx.stamp();                               // 11 feat
        }
        dense_w_out<8, CH(H_D0)>(cx, st, A, fin, Ly[H_ALPHA].bias_off, true, 1.0f);
        cx.stamp();                                   // 12 sigma
        {   // colour branch
            Blk in_d[2];
            {
                float rdir[NH][3];
                for_halves([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    const float *rq = rays + (p[q] / S) * ray_stride;
                    rdir[q][0] = rq[3]; rdir[q][1] = rq[4]; rdir[q][2] = rq[5];
                });
                pe_blocks_w<3, 4, 1>(rdir, h, in_d);
                for_halves([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    grid_block_w(grid, stash(q)[0], stash(q)[1], stash(q)[2], h, in_d[1].s[q],
                                 (dbg != nullptr && p_raw[q] < P) ? dbg + P * DBG_STRIDE_H + p[q] * 32 : nullptr);
                });
            }
            cx.stamp();                               // 13 PE(dir) + grid lookup
            Blk c[4], cn[4];
            dense_w<8, 1, 1, 4, CH(H_D1), false>(cx, st, A, in_d, in_d + 1, c, Ly[H_D0].bias_off, LK, 0.0f);
            dense_w<4, 0, 0, 4, CH(H_D1), true>(cx, st, c, nullptr, nullptr, cn, Ly[H_D1].bias_off, LK, LK);
            dense_w<4, 0, 0, 4, CH(H_D1), true>(cx, st, cn, nullptr, nullptr, c, Ly[H_D1].bias_off + 128, LK, LK);
            dense_w<4, 0, 0, 4, CH(H_RGB), true>(cx, st, c, nullptr, nullptr, cn, Ly[H_D3].bias_off, LK, LK);
            dense_w_out<4, CH(H_S0)>(cx, st, cn, fin, 0, false, LK);
        }
        cx.stamp();                                   // 14 colour branch
        {   // seg branch
            Blk s[4], sn[4];
            dense_w<8, 0, 0, 4, CH(H_S1), false>(cx, st, A, nullptr, nullptr, s, Ly[H_S0].bias_off, LK, 0.0f);
            dense_w<4, 0, 0, 4, CH(H_S1), true>(cx, st, s, nullptr, nullptr, sn, Ly[H_S1].bias_off, LK, LK);
            dense_w<4, 0, 0, 4, CH(H_S1), true>(cx, st, sn, nullptr, nullptr, s, Ly[H_S1].bias_off + 128, LK, LK);
            dense_w<4, 0, 0, 4, CH(H_SEG), true>(cx, st, s, nullptr, nullptr, sn, Ly[H_S3].bias_off, LK, LK);
            dense_w_out<4, CH(AFTER_RADIANCE)>(cx, st, sn, fin, 0, false, LK);
        }
        cx.stamp();                                   // 15 seg branch
        for_halves([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if (p_raw[q] < P) {   // rows 4h..4h+3 and 8+4h..8+4h+3 of [rgb3 | seg12 | sigma]
                *reinterpret_cast<f32x4 *>(raw + p_raw[q] * D_RAW + 4 * h) = f32x4{fin[q][0], fin[q][1], fin[q][2], fin[q][3]};
                *reinterpret_cast<f32x4 *>(raw + p_raw[q] * D_RAW + 8 + 4 * h) = f32x4{fin[q][4], fin[q][5], fin[q][6], fin[q][7]};
            }
        });
        cx.stamp();                                   // 16 store
    }
}

}  // namespace hw
}  // namespace SAHS_NS

using namespace SAHS_NS;
using namespace SAHS_NS::hw;

// Process-wide: 0 = packed-integer LeakyReLU (default), 1 = the reference's fp32 form; SAHS_BF16_EXACT_LEAKY=1 in the environment selects 1
// at first use.  set < 0 queries.  (One state per model build; sahs_bf16_exact_leaky of the C ABI sets all three.)
extern "C" int SAHS_SYM(sahs_bf16w_exact_leaky_state)(int set)
{
    static std::atomic<int> state{-1};
    int cur = state.load(std::memory_order_relaxed);
    if (cur < 0) {
        const char *e = getenv("SAHS_BF16_EXACT_LEAKY");
        cur = (e != nullptr && e[0] == '1') ? 1 : 0;
        state.store(cur, std::memory_order_relaxed);
    }
    if (set >= 0) { cur = set ? 1 : 0; state.store(cur, std::memory_order_relaxed); }
    return cur;
}

template <int MODE, bool EXACT>
static int launch_w2(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride, const float *zvals,
                     float *raw, float *dbg, float *xw, int xw_row, int xw_col0, const int *src, int num_cu, hipStream_t stream)
{
    if (P <= 0) return 0;
    const long ntiles = (P + W_PTS_PER_WG - 1) / W_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;       // the large-LDS attribute is per device (and per kernel instance)
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_forward_bf16w_kernel<MODE, EXACT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    field_forward_bf16w_kernel<MODE, EXACT><<<grid, W_THREADS, LDS_BYTES, stream>>>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg, xw, xw_row,
                                                                                   xw_col0, src);
    return (int)hipGetLastError();
}
template <int MODE>
static int launch_w(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride, const float *zvals,
                    float *raw, float *dbg, float *xw, int xw_row, int xw_col0, const int *src, int num_cu, hipStream_t stream)
{
    return SAHS_SYM(sahs_bf16w_exact_leaky_state)(-1)
               ? launch_w2<MODE, true>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg, xw, xw_row, xw_col0, src, num_cu, stream)
               : launch_w2<MODE, false>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg, xw, xw_row, xw_col0, src, num_cu, stream);
}

#if SAHS_MODEL != 1      // whole-network launches: AudioFaceModel, and NeRFaceModel without deformation nets (all of it is the radiance net)
extern "C" int SAHS_SYM(sahs_field_forward_bf16w_launch)(const float *packed, const float *frame, int level, long P, int S, const float *rays,
                                                         int ray_stride, const float *zvals, float *raw, float *dbg, int num_cu,
                                                         hipStream_t stream)
{
    return launch_w<FIELD_ALL>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg, nullptr, 0, 0, nullptr, num_cu, stream);
}
#endif

#if SAHS_MODEL != 2
// the split evaluation (field_f32.hip: sahs_field_forward_f32_split_launch, same arguments).  Built for NeRFaceModel too, radiance nets
// only (mode 2; src may be null = sample s of a ray is column s of xw): that model's deformation nets stay fp32.
extern "C" int SAHS_SYM(sahs_field_forward_bf16w_split_launch)(const float *packed, const float *frame, int level, int mode, long P, int S,
                                                               const float *rays, int ray_stride, const float *zvals, float *raw, float *xw,
                                                               int xw_row, int xw_col0, const int *src, int num_cu, hipStream_t stream)
{
    switch (mode) {
#if SAHS_MODEL == 0
    case FIELD_ALL:
        return launch_w<FIELD_ALL>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, nullptr, xw, xw_row, xw_col0, nullptr, num_cu, stream);
    case FIELD_DEFORM:
        return launch_w<FIELD_DEFORM>(packed, frame, level, P, S, rays, ray_stride, zvals, nullptr, nullptr, xw, xw_row, xw_col0, nullptr, num_cu, stream);
#endif
    case FIELD_RADIANCE:
        return launch_w<FIELD_RADIANCE>(packed, frame, level, P, S, rays, ray_stride, nullptr, raw, nullptr, xw, xw_row, 0, src, num_cu, stream);
    default:
        return -2;
    }
}
#endif
