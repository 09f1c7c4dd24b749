// bf16x3_pipe.hpp -- the split-operand dense layer on the bf16 matrix pipe, shared by the forward kernels of field_bf16x3.hip and the
// backward chain kernels of field_bwd_fused.hip.
//
// Every operand is split into two bf16 numbers, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16-17 significant bits together),
// and a product is three MFMAs with fp32 accumulation,
//     W x  ~=  W_hi x_hi + W_hi x_lo + W_lo x_hi            (the dropped W_lo x_lo term is ~2^-18 of the product).
// One wave owns 32 samples through a whole chain of layers: the 32x32 accumulator tile of a layer (16 values per lane) is converted in
// registers into the hi/lo B fragments of the next layer (Blk), under the MFMAs of the tile that follows it.  What happens to a finished
// value on its way -- the forward's activation, or the backward's derivative mask and the store of the pre-activation gradient -- is an
// EPILOGUE POLICY (EP) handed to dense_x:
//     ep.slope                      the factor of the "other" branch (A stage: m = v * slope, skipped when slope == 1)
//     ep.value<TILE, U>(v, m)       the value that is converted (B stage): forward max(v, m); backward bit ? v : m
//     ep.done4<TILE, G>(r, aux)     called once values 4G .. 4G+3 of tile TILE are final (r[0..3]): the saving forward and the backward store
//                                   them; aux = two registers of scratch that live across the calls of a tile
//     EP::stores(tile)              constexpr: done4 of that tile is ONE vector-memory store (so that a chunk's closing wait can be a COUNTED
//                                   vmcnt that retires the next chunk's LDS-DMA but not the younger stores: end_chunk below)
// Derived from field_bf16w.hip (one wave per SIMD: chunked weight stream through LDS-DMA, hand-issued A reads with counted lgkmcnt, bias
// as the C operand of a tile's first MFMA, conversion "ticks" dealt into the MFMA gaps).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"
#include "bf16_pipe.hpp"

namespace SAHS_NS {
namespace hx3 {
using namespace hb;      // the bf16 layer program of sahs_layout.hpp (layers, blocks, bias offsets); offsets of the doubled stream derived below
using namespace bfp;     // vector types, hand-issued LDS reads + retiring waits, bias helpers, the LDS-DMA chunk context (bf16_pipe.hpp)

struct Blk { u32x4 s[2]; u32x4 l[2]; };            // 32 features of this lane's sample: hi (s) and lo (l) bf16x8 fragments per k-step (as dwords)

constexpr int X_THREADS = 256;
constexpr int X_PTS_PER_WAVE = 32;
constexpr int X_PTS_PER_WG = (X_THREADS / WAVE) * X_PTS_PER_WAVE;     // 128
constexpr int LDS_BUF_BYTES = CHUNK_HW_MAX * 2;                       // 64 KB each, two of them
constexpr int LDS_BIAS_BYTE_OFF = 2 * LDS_BUF_BYTES;
constexpr int LDS_STASH_BYTE_OFF = LDS_BIAS_BYTE_OFF + ((BIAS_FLOATS + 3) / 4) * 16;
constexpr int STASH_FLOATS = 8;                                       // per sample: x'[3], w[2] (+pad)
constexpr int LDS_BYTES = LDS_STASH_BYTE_OFF + X_PTS_PER_WG * STASH_FLOATS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int AP = 4;                                                  // k-steps whose A fragments (hi and lo) are in flight ahead of their MFMAs
constexpr int PIECE_HW = X_THREADS * 8;                                // one LDS-DMA piece: 4 KB = 2048 halfwords (1 KB per wave)
constexpr int DMA_PIECES = LDS_BUF_BYTES / (X_THREADS * 16);           // 16
constexpr int FRAG_BYTES = 1024;                                       // one fragment of 64 lanes x 8 halfwords
constexpr int STEP_BYTES = 2 * FRAG_BYTES;                             // hi + lo
// chunking of the doubled stream: a k-step of a tile is 2 KB, a chunk at most 64 KB
constexpr int pick_GX(int KB32, int NT32)
{
    int g = (CHUNK_HW_MAX / 2) / (KB32 * 1024);
    if (g < 1) g = 1;
    if (g > NT32) g = NT32;
    while (NT32 % g) --g;
    return g;
}
constexpr long STREAM_HWX = 2 * STREAM_HW;                             // halfwords per level; layer i starts at 2 * kProgH.layer[i].stream_off
static_assert(11 * 2048 * 2 <= LDS_BUF_BYTES, "one tile of the widest layer (11 k-blocks) must fit a buffer");

#if defined(SAHS_DIAG) && defined(SAHS_X3_NODMA)            // timing-only experiments (tools/ablate.py x3*): results wrong by construction
constexpr bool kNoDma = true;
#else
constexpr bool kNoDma = false;
#endif
#if defined(SAHS_DIAG) && defined(SAHS_X3_NOBARRIER)
constexpr bool kNoBarrier = true;
#else
constexpr bool kNoBarrier = false;
#endif
typedef PipeCtx<X_THREADS, LDS_BUF_BYTES, LDS_BIAS_BYTE_OFF, kNoDma, kNoBarrier> Ctx;

// hi/lo split of a pair of fp32 values: hi = bf16 pair (RNE), lo = bf16 pair of the remainders
__device__ __forceinline__ uint32_t cvt_pair(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ void split_pair(float a, float b, uint32_t &hi, uint32_t &lo)
{
    hi = cvt_pair(a, b);
    lo = cvt_pair(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}

// the forward's epilogue: (leaky-)ReLU as max(v, slope v); slope 1 = no activation
struct FwdAct {
    static constexpr bool stores(int) { return false; }      // done4 issues no vector-memory instruction
    float slope;
    template <int TILE, int U> __device__ __forceinline__ float value(float v, float m) const { return slope == 1.0f ? v : fmaxf(v, m); }
    template <int TILE, int G> __device__ __forceinline__ void done4(const float (&)[4], uint32_t (&)[2]) const {}
};

// the SAVING forward's epilogue (training with the forward on this pipe): the activation, the post-activation values stored to the layer's
// plane of the saved-activation buffer (sahs_layout.hpp: act), and their signs as the bytes of the lane's two sign words (sbits: this lane
// (sample, h) owns words q = h and q = 2 + h outright; byte TILE of word h = groups 0 | 2, of word 2 + h = groups 1 | 3).
// The values do NOT go straight to memory: in the accumulator layout every store instruction would write 32 bytes into each of 32 rows
// (measured: the radiance launch of a training step at 3.1 ms instead of ~0.7, the vector-memory address path busy with quarter lines).
// A per-wave 32 x 32 staging tile in LDS (as in the backward chain, field_bwd_chain.hip) turns them around: the lane writes its four
// values, after the tile's fourth group the wave reads it back row-major and stores WHOLE 128-byte lines, non-temporal -- lane l -> rows
// (l >> 3) + 8 i, i = 0..3, 16 bytes (l & 7).  (Half tiles stored as 64-byte runs, the first version: 20 % more bytes at the memory --
// WRITE_SIZE 4.59 against 3.94 GB for the fine radiance launch -- and default-policy stores, which merge in L2, evict the weight stream:
// 1.31 against 1.04 ms.)  LDS instructions of one wave execute in issue order: no barrier between the writes and the read-back.
// Where the four tiles (18 KB) live is the kernel's business (field_bf16x3.hip: the saving kernels give up the biases of the nets they do
// not run, the radiance kernel also its x', w stash).
typedef __attribute__((address_space(3))) f32x4 *lds_f4_t;
constexpr int SAVE_ROW_BYTES = 144;                                      // 32 floats + 4 of padding: 16-byte aligned rows, spread over the banks
constexpr int SAVE_WAVE_BYTES = X_PTS_PER_WAVE * SAVE_ROW_BYTES;         // 4608
constexpr int LDS_BYTES_SAVE = 160 * 1024;
struct SaveStage {
    uint32_t wr, rd;          // LDS byte addresses: this lane's row (+16 h) for writing; row l >> 3, piece l & 7 for reading back
    uint32_t pc;              // 16 (l & 7): the piece's byte offset inside a 128-byte line
    uint32_t prow[4];         // the sample index of read-back row i (clamped to P - 1 like the lane's own sample)
    uint32_t soff;            // byte offset of this lane's sign word q = h in a plane of ONE word per (sample, q): 4 (4 sample + h)
};
// tile_byte_off: where this wave's staging tile starts in the workgroup's LDS
__device__ __forceinline__ SaveStage make_save_stage(const char *lds, uint32_t tile_byte_off, int lane, long wave_sample0, long sample, long P)
{
    SaveStage sc;
    const uint32_t base = lds_addr_of(lds) + tile_byte_off;
    sc.wr = base + (uint32_t)(lane & 31) * SAVE_ROW_BYTES + 16u * (uint32_t)(lane >> 5);
    sc.rd = base + (uint32_t)(lane >> 3) * SAVE_ROW_BYTES + 16u * (uint32_t)(lane & 7);
    sc.pc = 16u * (uint32_t)(lane & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long r = wave_sample0 + (lane >> 3) + 8 * i;
        sc.prow[i] = (uint32_t)(r < P ? r : P - 1);
    }
    sc.soff = 4u * (4u * (uint32_t)sample + (uint32_t)(lane >> 5));
    return sc;
}
template <int WIDTH, bool SIGN>
struct SaveAct {
    static constexpr bool stores(int) { return true; }      // (the counted wait of dense_x: a suffix of a tile's groups issues at least as many stores as it has groups: 4 + sign bytes at the fourth)
    static constexpr int NW = WIDTH >= 256 ? WIDTH / 128 : 1;      // sign words per (sample, q)
    float slope;
    float *plane;             // uniform: the layer's activation plane
    unsigned char *sign;      // uniform: the layer's sign plane (SIGN)
    SaveStage sc;
    template <int TILE, int U> __device__ __forceinline__ float value(float v, float m) const { return slope == 1.0f ? v : fmaxf(v, m); }
    template <int TILE, int G> __device__ __forceinline__ void done4(const float (&r)[4], uint32_t (&aux)[2]) const
    {
        *(lds_f4_t)(uintptr_t)(sc.wr + 32 * G) = f32x4{r[0], r[1], r[2], r[3]};
        if constexpr (G == 3) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *(lds_f4_t)(uintptr_t)(sc.rd + i * 8 * SAVE_ROW_BYTES);
#if defined(SAHS_DIAG) && defined(SAHS_X3_NOREADBACK)      // timing-only (results wrong by construction): the stores without waiting for the read-back
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = f32x4{r[0], r[1], r[2], r[3]};
#endif
#if defined(SAHS_DIAG) && defined(SAHS_X3_SAVE_PLAIN)      // A/B: default cache policy instead of non-temporal
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(plane) + (sc.prow[i] * (uint32_t)(WIDTH * 4) + sc.pc + (uint32_t)(128 * TILE))) = v[i];
#else
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(plane) + (sc.prow[i] * (uint32_t)(WIDTH * 4) + sc.pc + (uint32_t)(128 * TILE))));
#endif
        }
        if constexpr (SIGN) {
            const uint32_t nib = (r[0] > 0.0f ? 1u : 0u) | (r[1] > 0.0f ? 2u : 0u) | (r[2] > 0.0f ? 4u : 0u) | (r[3] > 0.0f ? 8u : 0u);
            if constexpr (G < 2) aux[G] = nib;
            else sign[sc.soff * (uint32_t)NW + (uint32_t)((G & 1) * 8 * NW + TILE)] = (unsigned char)(aux[G - 2] | (nib << 4));
        }
    }
};

// ---- epilogue + hi/lo conversion of a finished accumulator tile, software-pipelined (field_bf16w.hip: pack_tick) -------------------
// 16 values per lane and tile.  tick T:  A(T)    m = v_{T,T+1} * slope              (slope != 1 only, T even)
//                                        B(T-1)  r = ep.value(v, m);  after every fourth value: ep.done4
//                                        C(T-2)  hi dword = cvt_pk(r_{T-3}, r_{T-2})                       when T-2 is odd
//                                        D(T-3)  e = r - float(hi) for both values of the pair              when T-3 is odd
//                                        E(T-4)  lo dword = cvt_pk(e0, e1)                                  when T-4 is odd
struct PackState { f32x2 m2[2]; float r[4]; uint32_t hi[2]; float e[2][2]; uint32_t aux[2]; };      // aux: scratch of the epilogue policy (sign nibbles)
constexpr int NV = 16;
constexpr int PACK_TICKS = NV + 5;
template <int T, int TILE, class EP>
__device__ __forceinline__ void pack_tick(const f32x16 &acc, Blk &o, const EP ep, PackState &ps)
{
    if constexpr (T >= 0 && T < NV && !(T & 1)) {           // A(T), T even: both values of the pair
        constexpr int P = T >> 1;
        if (ep.slope != 1.0f) ps.m2[P & 1] = f32x2{acc[2 * P], acc[2 * P + 1]} * f32x2{ep.slope, ep.slope};
    }
    if constexpr (T - 1 >= 0 && T - 1 < NV) {               // B(T-1)
        constexpr int U = T - 1, P = U >> 1, e = U & 1;
        const float v = acc[2 * P + e];
        ps.r[U & 3] = ep.template value<TILE, U>(v, ps.m2[P & 1][e]);
        if constexpr ((U & 3) == 3) ep.template done4<TILE, (U >> 2)>(ps.r, ps.aux);
    }
    if constexpr (T - 2 >= 1 && T - 2 < NV && ((T - 2) & 1)) {    // C(T-2): the pair (U-1, U) is complete -> hi
        constexpr int U = T - 2, P = U >> 1, s = P >> 2, jp = P & 3;
        ps.hi[P & 1] = cvt_pair(ps.r[(U - 1) & 3], ps.r[U & 3]);
        o.s[s][jp] = ps.hi[P & 1];
    }
    if constexpr (T - 3 >= 1 && T - 3 < NV && ((T - 3) & 1)) {    // D(T-3): remainders (r[U-1], r[U] are still theirs: B has since
        constexpr int U = T - 3, P = U >> 1;                      //          written r[(U+1)&3] and r[(U+2)&3] only)
        ps.e[P & 1][0] = ps.r[(U - 1) & 3] - __builtin_bit_cast(float, ps.hi[P & 1] << 16);
        ps.e[P & 1][1] = ps.r[U & 3] - __builtin_bit_cast(float, ps.hi[P & 1] & 0xffff0000u);
    }
    if constexpr (T - 4 >= 1 && T - 4 < NV && ((T - 4) & 1)) {    // E(T-4): lo
        constexpr int U = T - 4, P = U >> 1, s = P >> 2, jp = P & 3;
        o.l[s][jp] = cvt_pair(ps.e[P & 1][0], ps.e[P & 1][1]);
    }
}
template <int LO, int HI, int TILE, class EP>
__device__ __forceinline__ void pack_ticks(const f32x16 &acc, Blk &o, const EP ep, PackState &ps)
{
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (pack_tick<LO + Is, TILE>(acc, o, ep, ps), ...); }(std::make_integer_sequence<int, HI - LO>{});
}

// LGKM bookkeeping as in field_bf16w.hip (Sched), with TWO A reads (hi, lo) per k-step: step I issues [wait] [3 MFMAs] [bias batch of the
// next tile: 4 reads, when I starts a tile that has a successor] [A reads of step I+AP: hi then lo]; the wait of step I allows exactly
// the reads issued after A_lo(I).
template <int STEPS, int TOTAL, int NT32, int T0>
struct Sched {
    static constexpr bool bias_at(int s) { return s >= 0 && s < TOTAL && (s % STEPS == 0) && (T0 + s / STEPS + 1 < NT32); }
    static constexpr bool aread_at(int s) { return s >= 0 && s + AP < TOTAL; }
    static constexpr int cnt(int I)
    {
        int n = 0, lo = 0;
        if (I < AP) n += 2 * ((AP < TOTAL ? AP : TOTAL) - 1 - I);   // read in the prologue: the later prologue reads, then steps 0..I-1
        else lo = I - AP + 1;                                        // read at the END of step I-AP (after that step's bias batch)
        for (int s = lo; s < I; ++s) n += (aread_at(s) ? 2 : 0) + (bias_at(s) ? 4 : 0);
        return n;
    }
    static constexpr int max_cnt()
    {
        int m = 0;
        for (int i = 0; i < TOTAL; ++i) m = cnt(i) > m ? cnt(i) : m;
        return m;
    }
};

// the conversion ticks a (step, gap) pair is dealt: slot of NSLOT -> [lo, hi)
constexpr int tick_lo(int slot, int nslot) { return PACK_TICKS * slot / nslot; }
constexpr int tick_hi(int slot, int nslot) { return PACK_TICKS * (slot + 1) / nslot; }
// done4 calls among ticks [lo, hi): tick T finishes value T - 1; every fourth value (T = 4, 8, 12, 16) ends a group
constexpr int groups_done(int lo, int hi)
{
    int n = 0;
    for (int T = lo; T < hi; ++T) n += (T >= 1 && T - 1 < NV && ((T - 1) & 3) == 3) ? 1 : 0;
    return n;
}
// end of a chunk when the layer's epilogue stores: the wave's LDS-DMA pieces of the next chunk must have landed, the N vector-memory
// instructions it issued AFTER the last of them (stores, in issue order behind it: MI355X_MICROARCH.md, vmcnt) may stay in flight --
// __syncthreads() would drain them all (vmcnt(0)) once per chunk, and a store-heavy kernel then runs compute and stores in turns
template <int N>
__device__ __forceinline__ void barrier_after_dma()
{
    static_assert(N >= 0, "count");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N < 63 ? N : 63) : "memory");
}

struct St {
    f32x16 acc[2];       // two accumulator sets: tile t accumulates into one while tile t-1 is converted from the other
    PackState ps;
};

// hidden layer: NT32 output tiles of 32 rows.  The layer's last tile stays in st.acc[1]; the NEXT layer converts it (PEND) into
// in0[K0-1] with the PREVIOUS layer's policy pep (PTILE = that layer's NT32 - 1: the tile index its policy sees) under its own first
// MFMAs, before the step that first reads that block.
template <int K0, int K1, int K2, int NT32, int NEXT_HW, bool PEND, int PTILE = 0, class EP, class PEP>
__device__ __forceinline__ void dense_x(Ctx &cx, St &st, Blk *in0, const Blk *in1, const Blk *in2, Blk *out, int bias_off, const EP ep, const PEP pep)
{
    constexpr int KB = K0 + K1 + K2;
    constexpr int G = pick_GX(KB, NT32);
    constexpr int STEPS = KB * 2, TOTAL = G * STEPS, NCH = NT32 / G;
    static_assert(NT32 % 2 == 0, "the last tile of a layer must land in accumulator set 1");
    static_assert(TOTAL * STEP_BYTES <= LDS_BUF_BYTES, "chunk does not fit its buffer");
    static_assert(!PEND || 2 * K0 - 3 >= 1, "no room for the pending tile's conversion");
    static_assert(STEPS >= AP || NT32 == 1, "the counted waits assume a tile's bias batch is issued before the A reads of the next tile's first step");
    constexpr bool STORING = EP::stores(0) || EP::stores(NT32 - 1) || PEP::stores(PTILE);      // the epilogue issues vector-memory stores
    constexpr int PPS = 4;                                                                     // DMA pieces per step when front-loaded
    const uint32_t baddr = cx.bias_addr + 4u * (uint32_t)bias_off;
    f32x4 braw[2][4];
    bias_read<0>(braw[0], baddr);

    auto chunk = [&]<int C>() {
        constexpr int T0 = C * G;
        constexpr int nhw = (C + 1 < NCH) ? G * KB * 2048 : NEXT_HW;
        constexpr int npieces = (nhw + PIECE_HW - 1) / PIECE_HW;
        constexpr int PSTEP = (TOTAL * 3 / 4) / (npieces > 0 ? npieces : 1) > 0 ? (TOTAL * 3 / 4) / (npieces > 0 ? npieces : 1) : 1;
        using S = Sched<STEPS, TOTAL, NT32, T0>;
        static_assert(S::max_cnt() <= 15, "lgkmcnt is a 4-bit counter");
        static_assert(!STORING || (npieces + PPS - 1) / PPS <= TOTAL, "front-loaded DMA pieces must fit the chunk's steps");
        cx.begin_chunk(nhw);
        const uint32_t abase = cx.cur_addr();
        u32x4 ah[AP], al[AP];
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
            ((lds_read16<Is * STEP_BYTES>(ah[Is], abase), lds_read16<Is * STEP_BYTES + FRAG_BYTES>(al[Is], abase)), ...);
        }(std::make_integer_sequence<int, (AP < TOTAL ? AP : TOTAL)>{});
        fence();
        auto step = [&]<int I>() {
            constexpr int g = I / STEPS, k = I % STEPS, b = k >> 1, st_ = k & 1, t = T0 + g, set = t & 1;
            const Blk &x = (b < K0) ? in0[b] : ((b < K0 + K1) ? in1[b - K0] : in2[b - K0 - K1]);
            // retires A_hi(I), A_lo(I) and, at a tile's first step, the tile's bias batch (bf16_pipe.hpp: wait_retire)
            if constexpr (k == 0) wait_retire<S::cnt(I)>(ah[I % AP], al[I % AP], braw[set]);
            else wait_retire<S::cnt(I)>(ah[I % AP], al[I % AP]);
            fence();
            f32x16 cb;
            if constexpr (k == 0) cb = bias_as_c(braw[set]);
            // this step's share of the conversion ticks, dealt over its THREE MFMA gaps (a gap hides ~6 VALU instructions, not more):
            // the finished tile t-1 -> out[t-1] over steps 1..STEPS-1, or the previous layer's last tile -> in0[K0-1] over steps 1..2K0-3
            auto ticks = [&]<int J>() {
                if constexpr (t > 0 && k >= 1) {
                    constexpr int NSLOT = 3 * (STEPS - 1), slot = 3 * (k - 1) + J;
                    constexpr int lo = tick_lo(slot, NSLOT), hi = tick_hi(slot, NSLOT);
                    if constexpr (hi > lo) pack_ticks<lo, hi, t - 1>(st.acc[set ^ 1], out[t - 1], ep, st.ps);
                } else if constexpr (PEND && t == 0 && k >= 1 && k <= 2 * K0 - 3) {
                    constexpr int NSLOT = 3 * (2 * K0 - 3), slot = 3 * (k - 1) + J;
                    constexpr int lo = tick_lo(slot, NSLOT), hi = tick_hi(slot, NSLOT);
                    if constexpr (hi > lo) pack_ticks<lo, hi, PTILE>(st.acc[1], in0[K0 - 1], pep, st.ps);
                }
            };
            // W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi on ONE accumulator: back-to-back dependent MFMAs of this shape run at the pipe's rate
            // (three separate accumulation chains were measured: 59.6 against 56.8 ms per fine launch -- slower, not faster)
            if constexpr (k == 0) st.acc[set] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ah[I % AP]), frag(x.s[st_]), cb, 0, 0, 0);
            else st.acc[set] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ah[I % AP]), frag(x.s[st_]), st.acc[set], 0, 0, 0);
            fence();
            if constexpr (S::bias_at(I)) bias_read<128 * (t + 1)>(braw[set ^ 1], baddr);
            ticks.template operator()<0>();
            fence();
            st.acc[set] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ah[I % AP]), frag(x.l[st_]), st.acc[set], 0, 0, 0);
            fence();
            ticks.template operator()<1>();
            fence();
            st.acc[set] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(al[I % AP]), frag(x.s[st_]), st.acc[set], 0, 0, 0);
            fence();
#if !(defined(SAHS_DIAG) && defined(SAHS_X3_NOAREAD))
            if constexpr (S::aread_at(I)) {
                lds_read16<(I + AP) * STEP_BYTES>(ah[I % AP], abase);
                lds_read16<(I + AP) * STEP_BYTES + FRAG_BYTES>(al[I % AP], abase);
            }
#endif
            if constexpr (STORING) {      // front-loaded: PPS pieces per step from step 0 (see `after` below)
                [&]<int... Qs>(std::integer_sequence<int, Qs...>) { ((I * PPS + Qs < npieces ? cx.issue_piece(I * PPS + Qs) : (void)0), ...); }(
                    std::make_integer_sequence<int, PPS>{});
            } else {
                if constexpr (I % PSTEP == 0 && I / PSTEP < npieces) cx.issue_piece(I / PSTEP);
            }
            ticks.template operator()<2>();
            fence();
        };
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (step.template operator()<Is>(), ...); }(std::make_integer_sequence<int, TOTAL>{});
        if constexpr (!STORING) {
            [&]<int... Ps>(std::integer_sequence<int, Ps...>) {
                ((Ps >= (TOTAL + PSTEP - 1) / PSTEP && Ps < npieces ? cx.issue_piece(Ps) : (void)0), ...);
            }(std::make_integer_sequence<int, DMA_PIECES>{});
        }
        // A storing epilogue and the single in-order vmcnt: the wait that retires the next chunk's LDS-DMA also waits for every OLDER store.
        // So the DMA pieces go out at the START of the chunk (PPS per step), ahead of the chunk's own stores: what the closing wait then has
        // to see finished, besides the pieces, are only stores of EARLIER chunks, which have had a whole chunk's MFMAs to drain -- with the
        // pieces spread over the chunk (the forward's schedule) every chunk stalled for most of a store's round trip.  `after` = the stores
        // this chunk issues behind its last piece (the same dealing of ticks as in `ticks` above, evaluated at compile time; a lower bound).
        constexpr int after = [] {
            if (npieces == 0) return 63;                                                       // nothing was fetched
            const int LP = (npieces - 1) / PPS;                                                // the step that issues the last piece (before its gap 2)
            int n = 0;
            for (int I = LP; I < TOTAL; ++I) {
                const int g = I / STEPS, k = I % STEPS, t = T0 + g;
                for (int J = (I == LP ? 2 : 0); J < 3; ++J) {
                    if (t > 0 && k >= 1) {
                        const int NSLOT = 3 * (STEPS - 1), slot = 3 * (k - 1) + J;
                        if (EP::stores(t - 1)) n += groups_done(tick_lo(slot, NSLOT), tick_hi(slot, NSLOT));
                    } else if (PEND && t == 0 && k >= 1 && k <= 2 * K0 - 3) {
                        const int NSLOT = 3 * (2 * K0 - 3), slot = 3 * (k - 1) + J;
                        if (PEP::stores(PTILE)) n += groups_done(tick_lo(slot, NSLOT), tick_hi(slot, NSLOT));
                    }
                }
            }
            return n;
        }();
        if constexpr (STORING) {
            barrier_after_dma<after>();
            cx.buf ^= 1;
        } else {
            cx.end_chunk();
        }
    };
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) { (chunk.template operator()<Cs>(), ...); }(std::make_integer_sequence<int, NCH>{});
}

// 16-row output layer (ALPHA -> RGB -> SEG into one tile) accumulated in fp32: first = start from the bias, else from the running tile.
template <int K0, int NEXT_HW, int PTILE = 0, class PEP>
__device__ __forceinline__ void dense_x_out(Ctx &cx, St &st, Blk *in0, f32x16 &acc, int bias_off, bool first, const PEP pep)
{
    constexpr int TOTAL = K0 * 2;
    constexpr int npieces = (NEXT_HW + PIECE_HW - 1) / PIECE_HW;
    static_assert(TOTAL * STEP_BYTES <= LDS_BUF_BYTES, "chunk does not fit its buffer");
    cx.begin_chunk(NEXT_HW);
    const uint32_t abase = cx.cur_addr();
    if (first) {       // rows 0..15 of the 32-row tile carry the layer's bias (registers 0..7), rows 16..31 are unused
        const uint32_t baddr = cx.bias_addr + 4u * (uint32_t)bias_off;
        f32x4 t0, t1;
        lds_read16<0>(t0, baddr);
        lds_read16<32>(t1, baddr);
        wait_retire<0>(t0, t1);
        fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] = t0[r]; acc[4 + r] = t1[r]; }
#pragma unroll
        for (int r = 8; r < 16; ++r) acc[r] = 0.0f;
    }
    u32x4 ah[AP], al[AP];
    [&]<int... Is>(std::integer_sequence<int, Is...>) {
        ((lds_read16<Is * STEP_BYTES>(ah[Is], abase), lds_read16<Is * STEP_BYTES + FRAG_BYTES>(al[Is], abase)), ...);
    }(std::make_integer_sequence<int, (AP < TOTAL ? AP : TOTAL)>{});
    fence();
    auto step = [&]<int I>() {
        wait_retire<2 * ((TOTAL - 1 - I) < (AP - 1) ? (TOTAL - 1 - I) : (AP - 1))>(ah[I % AP], al[I % AP]);
        fence();
        const Blk &x = in0[I >> 1];
        auto ticks = [&]<int J>() {      // the previous layer's last tile -> in0[K0-1], before step 2 (K0 - 1) reads it
            if constexpr (I >= 1 && I <= 2 * K0 - 3) {
                constexpr int NSLOT = 3 * (2 * K0 - 3), slot = 3 * (I - 1) + J;
                constexpr int lo = PACK_TICKS * slot / NSLOT, hi = PACK_TICKS * (slot + 1) / NSLOT;
                if constexpr (hi > lo) pack_ticks<lo, hi, PTILE>(st.acc[1], in0[K0 - 1], pep, st.ps);
            }
        };
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ah[I % AP]), frag(x.s[I & 1]), acc, 0, 0, 0);
        fence();
        ticks.template operator()<0>();
        fence();
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(ah[I % AP]), frag(x.l[I & 1]), acc, 0, 0, 0);
        fence();
        ticks.template operator()<1>();
        fence();
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(al[I % AP]), frag(x.s[I & 1]), acc, 0, 0, 0);
        fence();
        if constexpr (I + AP < TOTAL) {
            lds_read16<(I + AP) * STEP_BYTES>(ah[I % AP], abase);
            lds_read16<(I + AP) * STEP_BYTES + FRAG_BYTES>(al[I % AP], abase);
        }
        if constexpr (I < npieces) cx.issue_piece(I);
        ticks.template operator()<2>();
        fence();
    };
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (step.template operator()<Is>(), ...); }(std::make_integer_sequence<int, TOTAL>{});
    [&]<int... Ps>(std::integer_sequence<int, Ps...>) { ((Ps >= TOTAL && Ps < npieces ? cx.issue_piece(Ps) : (void)0), ...); }(
        std::make_integer_sequence<int, DMA_PIECES>{});
    cx.end_chunk();
}

}  // namespace hx3
}  // namespace SAHS_NS
