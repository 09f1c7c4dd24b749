// field_bwd_chain_f32.hip -- the data-gradient chain of the field backward in exact fp32 products (the reference's arithmetic).
//
// Autograd of the reference's per-sample networks (modules.py:254-295 NeRFMLP, :371-390 WarpFieldMLP, :444-462 HyperSheetMLP, as driven by
// train_stage_rays_auto.py:437-499): dZ_l = (W_{l+1}^T dZ_{l+1}) * act'(z_l) for every dense layer.  field_bwd_chain.hip walks a net
// backwards on the bf16 pipe with split operands; this file is the same walk for ops.backward_gemm_precision("fp32") on
// v_mfma_f32_16x16x4_f32, built from the forward kernel's own layer routine (f32_pipe.hpp: dense_ep): one wave owns 16 samples, the D
// tile of a layer (lane (q, j): gradients 4q..4q+3 of sample j) is the B operand of the next, the transposed weights stream L2 -> LDS by
// LDS-DMA in <= 32 KB chunks.  The epilogue policy is the backward's: start from zero, multiply by the (leaky-)ReLU derivative read from
// the sign bits the saving forward wrote (sahs_layout.hpp: sbits -- the forward's lane (q, j) wrote the very word this lane (q, j) reads),
// store the dZ tile into the plane of its layer (the operand of the weight-gradient launch, field_bwd.hip: gemm_tn_jobs*_f32_kernel).
// Same planes, seam buffers and launch order as the split-operand chain; sums in a different order than the per-layer GEMMs it replaces
// (gemm_dma_kernel<false, false>), same exact products.  AudioFaceModel only (SAHS_MODEL 0).
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"
#include "f32_pipe.hpp"

#if SAHS_MODEL != 0
#error "field_bwd_chain_f32.hip is built for the AudioFaceModel only"
#endif

namespace SAHS_NS {
namespace bwf {

// ---- the backward layer program, in 16-row tiles and 16-gradient k-blocks ------------------------------------------------------------
// A backward layer multiplies A = (part of) W^T: output row = an INPUT feature of the forward layer (W's column col0 + row), K index = an
// OUTPUT feature of the forward layer (W's row k - kshift).  Up to two K segments (d feat sums the seg and colour branches), up to two row
// ranges (the encodings' gradient: PE(x') columns, then PE(w) columns).  The heads take the 16-float d_raw row [drgb3 | dseg12 | dsigma]
// as their one k-block: kshift places the head's weight rows (fc_seg: k = 3..14, fc_alpha: k = 15).
struct SegF { long w_off[2]; int ld, kshift, krows; };
struct RowF { int rows, col0, valid; };
struct LayerF {
    int NT, KB, nseg; SegF seg[2]; int blocks[2];
    int nrow; RowF row[2];
    long stream_off;      // floats, in this part's stream
    int chunk;            // floats per LDS chunk
};
enum RadF { R_RGBH, R_D3, R_D2, R_D1, R_GRIDF, R_SEGH, R_S3, R_S2, R_S1, R_FEATA, R_FEATB, R_FEAT, R_T7, R_T6, R_T5, R_T4, R_T3IN, R_T3, R_T2, R_T1,
            R_T0IN, R_COUNT };
enum DefF { D_HF, D_H5, D_H4, D_H3, D_H2, D_H1, D_WF, D_W5, D_W4, D_W3, D_W2, D_W1, D_COUNT };
template <int N> struct ProgF { LayerF layer[N]; long stream_floats; };

constexpr LayerF mkf(SegF s0, RowF r0, SegF s1 = {{0, 0}, 0, 0, 0}, RowF r1 = {0, 0, 0})
{
    LayerF L{};
    L.seg[0] = s0; L.seg[1] = s1;
    L.nseg = 1 + (s1.krows > 0);
    L.blocks[0] = (s0.kshift + s0.krows + 15) / 16;
    L.blocks[1] = (s1.kshift + s1.krows + 15) / 16;
    L.KB = L.blocks[0] + L.blocks[1];
    L.row[0] = r0; L.row[1] = r1;
    L.nrow = 1 + (r1.rows > 0);
    L.NT = (r0.rows + r1.rows) / 16;
    return L;
}
template <int N> constexpr void finish(ProgF<N> &P)
{
    long off = 0;
    for (int i = 0; i < N; ++i) {
        P.layer[i].stream_off = off;
        P.layer[i].chunk = pick_G(P.layer[i].KB, P.layer[i].NT) * P.layer[i].KB * 256;
        off += (long)P.layer[i].NT * P.layer[i].KB * 256;
    }
    P.stream_floats = off;
}
constexpr ProgF<R_COUNT> make_rad()
{
    ProgF<R_COUNT> P{};
    const FlatOffsets::Lvl &c = kFlat.lvl[0], &n = kFlat.lvl[1];
    auto sq = [](long w0, long w1, int ld, int krows) { return SegF{{w0, w1}, ld, 0, krows}; };
    LayerF *L = P.layer;
    // colour branch, from its head back (modules.py:276-287)
    L[R_RGBH] = mkf(SegF{{c.rgb_w, n.rgb_w}, BR_H, 0, 3}, RowF{BR_H, 0, BR_H});
    for (int i = 3; i >= 1; --i) L[R_D3 + (3 - i)] = mkf(sq(c.dir_w[i], n.dir_w[i], BR_H, BR_H), RowF{BR_H, 0, BR_H});
    L[R_GRIDF] = mkf(sq(c.dir_w[0], n.dir_w[0], D_DIR_IN, BR_H), RowF{D_GRID, TR_H + D_DIR, D_GRID});
    // seg branch (modules.py:289-294)
    L[R_SEGH] = mkf(SegF{{c.segout_w, n.segout_w}, BR_H, 3, N_SEG}, RowF{BR_H, 0, BR_H});
    for (int i = 3; i >= 1; --i) L[R_S3 + (3 - i)] = mkf(sq(c.seg_w[i], n.seg_w[i], BR_H, BR_H), RowF{BR_H, 0, BR_H});
    // d feat = w_alpha dsigma (A) + W_S0^T dS0 + W_D0[:, :256]^T dC0 (B, accumulating)
    L[R_FEATA] = mkf(SegF{{c.alpha_w, n.alpha_w}, TR_H, 15, 1}, RowF{TR_H, 0, TR_H});
    L[R_FEATB] = mkf(sq(c.seg_w[0], n.seg_w[0], TR_H, BR_H), RowF{TR_H, 0, TR_H}, sq(c.dir_w[0], n.dir_w[0], D_DIR_IN, BR_H));
    // trunk (modules.py:267-274), skip layer 3: [h | PE(x') | PE(w) | pose]
    L[R_FEAT] = mkf(sq(c.feat_w, n.feat_w, TR_H, TR_H), RowF{TR_H, 0, TR_H});
    for (int i = 7; i >= 4; --i) L[R_T7 + (7 - i)] = mkf(sq(c.xyz_w[i], n.xyz_w[i], TR_H, TR_H), RowF{TR_H, 0, TR_H});
    L[R_T3IN] = mkf(sq(c.xyz_w[3], n.xyz_w[3], TR_H + D_TR_IN, TR_H), RowF{16 * KB_XYZ, TR_H, D_XYZ}, SegF{{0, 0}, 0, 0, 0},
                    RowF{16 * KB_AMB, TR_H + D_XYZ, D_AMB});
    L[R_T3] = mkf(sq(c.xyz_w[3], n.xyz_w[3], TR_H + D_TR_IN, TR_H), RowF{TR_H, 0, TR_H});
    L[R_T2] = mkf(sq(c.xyz_w[2], n.xyz_w[2], TR_H, TR_H), RowF{TR_H, 0, TR_H});
    L[R_T1] = mkf(sq(c.xyz_w[1], n.xyz_w[1], TR_H, TR_H), RowF{TR_H, 0, TR_H});
    L[R_T0IN] = mkf(sq(c.xyz_w[0], n.xyz_w[0], D_TR_IN, TR_H), RowF{16 * KB_XYZ, 0, D_XYZ}, SegF{{0, 0}, 0, 0, 0}, RowF{16 * KB_AMB, D_XYZ, D_AMB});
    finish(P);
    return P;
}
constexpr ProgF<D_COUNT> make_def()
{
    ProgF<D_COUNT> P{};
    const FlatOffsets &f = kFlat;
    auto sq = [](long w, int ld, int krows) { return SegF{{w, w}, ld, 0, krows}; };
    LayerF *L = P.layer;
    // hyper sheet (modules.py:444-462): w = fc_ambient(g5); skip layer 4: [g | PE(x) | driving | pose]
    L[D_HF] = mkf(sq(f.hyp_fw, HYP_H, AMB_DIM), RowF{HYP_H, 0, HYP_H});
    for (int i = 5; i >= 1; --i) L[D_H5 + (5 - i)] = mkf(sq(f.hyp_w[i], i == 4 ? HYP_H + D_DEF_IN : HYP_H, HYP_H), RowF{HYP_H, 0, HYP_H});
    // warp field (modules.py:371-390): dx = tanh(fc_final(h5))
    L[D_WF] = mkf(sq(f.warp_fw, WARP_H, 3), RowF{WARP_H, 0, WARP_H});
    for (int i = 5; i >= 1; --i) L[D_W5 + (5 - i)] = mkf(sq(f.warp_w[i], i == 4 ? WARP_H + D_DEF_IN : WARP_H, WARP_H), RowF{WARP_H, 0, WARP_H});
    finish(P);
    return P;
}
constexpr ProgF<R_COUNT> kRad = make_rad();
constexpr ProgF<D_COUNT> kDef = make_def();
__device__ const ProgF<R_COUNT> dRad = make_rad();
__device__ const ProgF<D_COUNT> dDef = make_def();
constexpr long RAD_FLOATS = kRad.stream_floats, DEF_FLOATS = kDef.stream_floats;
// (a DMA piece is 8 KB: a chunk shorter than that is over-read into what follows it in the stream -- never past the stream's end)
static_assert(kRad.layer[R_T0IN].chunk >= PIECE_FLOATS && kDef.layer[D_W1].chunk >= PIECE_FLOATS, "the stream's last chunk is whole DMA pieces");

// ---- transposed stream of one part: [layer][tile16][k-block][lane 64][4]; lane = 16 q + i holds A[16 t + i][16 b + 4 q + r], r = 0..3 (the
// A-fragment order of f32_pipe.hpp, as pack.hip writes the forward's) ----------------------------------------------------------------------
template <int N>
__device__ __forceinline__ float pack_one(const ProgF<N> &Pg, const float *__restrict__ flat, int level, long e)
{
    int li = 0;
    while (li + 1 < N && Pg.layer[li + 1].stream_off <= e) ++li;
    const LayerF &L = Pg.layer[li];
    const long w = e - L.stream_off;
    const int per_tile = L.KB * 256;
    const int t = (int)(w / per_tile);
    const int rem = (int)(w - (long)t * per_tile);
    const int b = rem >> 8, lane = (rem & 255) >> 2, r = rem & 3;
    const int row = 16 * t + (lane & 15), q = lane >> 4;
    int col = -1, r0 = 0;
    for (int rs = 0; rs < L.nrow; ++rs) {
        if (row < r0 + L.row[rs].rows) { if (row - r0 < L.row[rs].valid) col = L.row[rs].col0 + (row - r0); break; }
        r0 += L.row[rs].rows;
    }
    int bb = b, sg = 0;
    if (bb >= L.blocks[0]) { bb -= L.blocks[0]; sg = 1; }
    const int k = 16 * bb + 4 * q + r - L.seg[sg].kshift;
    if (col < 0 || k < 0 || k >= L.seg[sg].krows) return 0.0f;
    return flat[L.seg[sg].w_off[level] + (long)k * L.seg[sg].ld + col];
}
__global__ void __launch_bounds__(256) pack_bwd_stream_f32_kernel(const float *__restrict__ flat, float *__restrict__ out, int level, int part)
{
    const long total = part == 1 ? DEF_FLOATS : RAD_FLOATS;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
        out[e] = part == 1 ? pack_one(dDef, flat, 0, e) : pack_one(dRad, flat, level, e);
}

// ---- the backward epilogue ------------------------------------------------------------------------------------------------------------
// MASKED: the layer behind this gradient has a (leaky-)ReLU -- bit 4 (t & 7) + r of word t >> 3 of this lane's sign words says whether value
// r of tile t was > 0 (derivative 1, else `slope`).  save (STORE): this lane's slot (row p, column 4 q) of the plane the tile is stored to.
// accum: the tile continues the sum a previous layer left in out[] (d feat).
template <bool MASKED, bool STORE = true>
struct BwdEpF {
    static constexpr int kStores = STORE ? 1 : 0;      // (unconditional: lanes past the end redo the last sample and store the same values again)
    bool accum; float slope; uint32_t m[2]; float *save;
    __device__ __forceinline__ f32x4 first(const Ctx &, const f32x4 *out, int t) const { return accum ? out[t] : f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    template <int NT> __device__ __forceinline__ void done(const f32x4 &acc, f32x4 *out, int t)
    {
        f32x4 o = acc;
        if constexpr (MASKED) {
            const uint32_t w = m[t >> 3];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (w & (1u << (4 * (t & 7) + r))) ? acc[r] : acc[r] * slope;
        }
        asm volatile("" : "+v"(o));     // pin: keep the finished tile from being sunk into the next layer
        out[t] = o;
        if constexpr (STORE) *reinterpret_cast<f32x4 *>(save + 16 * t) = o;      // (non-temporal: measured no gain here, 2.31 vs 2.30 ms per launch)
    }
};

constexpr int RAD_MASK_WORDS = 2 * TR_LAYERS + 8, DEF_MASK_WORDS = 12;      // sign words per lane and sample tile
static_assert(TR_LAYERS == 8, "slots 16.. of the radiance mask words follow the trunk's 16");
constexpr int CHAIN_LDS_BYTES = 2 * LDS_BUF_FLOATS * 4 + (F32_THREADS / WAVE) * RAD_MASK_WORDS * WAVE * 4;      // the two weight-chunk buffers + the waves' mask words

#define CHR(id) (kRad.layer[id].chunk)
#define CHD(id) (kDef.layer[id].chunk)

__device__ __forceinline__ void start_stream(Ctx &cx, const float *stream, float *lds, long floats, int first_chunk)
{
    cx.stream = stream;
    cx.lds = lds;
    cx.buf = 1;                       // so that the first chunk lands in buffer 0
    cx.lane = threadIdx.x & 63;
    cx.q = cx.lane >> 4;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cx.wrap_to = 0u;
    cx.wrap_at = (uint32_t)floats;
    cx.off = 0u;
    cx.begin_chunk(first_chunk);
#pragma unroll
    for (int pc = 0; pc < MAX_PIECES; ++pc)
        if (pc < (first_chunk + PIECE_FLOATS - 1) / PIECE_FLOATS) cx.issue_piece(pc);
    cx.end_chunk();
}

// Radiance nets of one level, backwards.  d_raw (P,16); bits: the radiance sign planes (sbits::BR_*); dact: plane c of the act:: table at
// dact + c * P (written: C, S, FEAT, T planes); dgridf (P,32); din_a, din_b (P,96): the encodings' gradient through the skip layer and
// through layer 0 (summed by encode_backward).
__global__ void __launch_bounds__(F32_THREADS, 2)
field_backward_chain_rad_f32_kernel(const float *__restrict__ stream, long P, const float *__restrict__ d_raw, const uint32_t *__restrict__ bits,
                                    float *__restrict__ dact, float *__restrict__ dgridf, float *__restrict__ din_a, float *__restrict__ din_b)
{
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    Ctx cx;
    start_stream(cx, stream, lds_f, RAD_FLOATS, CHR(R_RGBH));
    const int q = cx.q;
    const long ntiles = (P + F32_PTS_PER_WG - 1) / F32_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p_raw = tile * F32_PTS_PER_WG + cx.wave * F32_PTS_PER_WAVE + (cx.lane & 15);
        const long p = p_raw < P ? p_raw : P - 1;      // lanes past the end redo the last sample (and store its values again)
        long Pq = P;
        asm volatile("" : "+s"(Pq));                   // (the ~25 plane bases c * P are loop-invariant: keep them from being hoisted and spilled)
        const uint32_t sb_lane = (uint32_t)(p * 4 + q);
#define DZ(c, w) (dact + (long)(c) * Pq + p * (long)(w) + 4 * q)
        // This lane's 24 sign words of the tile, fetched ONCE and parked in LDS (word k at mw[64 k]: trunk layer l -> 2 l, 2 l + 1; colour
        // layer i -> 16 + i; seg layer i -> 20 + i).  Fetched where a layer needs them, each was an ordinary global load whose use the compiler
        // guards with vmcnt(0) while LDS-DMA is in flight -- a full drain of the weight prefetch (and of the tile stores) in the middle of a
        // chunk, 16 times per sample tile; from LDS the layers read them on the lgkm counter.
        uint32_t *const mw = reinterpret_cast<uint32_t *>(lds_f + 2 * LDS_BUF_FLOATS) + cx.wave * (RAD_MASK_WORDS * WAVE) + cx.lane;
        {
            uint32_t w[RAD_MASK_WORDS];
#pragma unroll
            for (int l = 0; l < TR_LAYERS; ++l) {
                const uint32_t *a = bits + (long)(sbits::BR_T + 8 * l) * Pq + sb_lane * 2u;
                w[2 * l] = a[0]; w[2 * l + 1] = a[1];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[16 + i] = bits[(long)(sbits::BR_C + 4 * i) * Pq + sb_lane];
                w[20 + i] = bits[(long)(sbits::BR_S + 4 * i) * Pq + sb_lane];
            }
#pragma unroll
            for (int k = 0; k < RAD_MASK_WORDS; ++k) mw[WAVE * k] = w[k];
        }
        auto ep128 = [&](int slot, int c) {            // a 128-wide leaky-ReLU layer: one sign word per lane
            BwdEpF<true> e;
            e.accum = false; e.slope = 0.01f;
            e.m[0] = mw[WAVE * slot]; e.m[1] = 0u;
            e.save = DZ(c, 128);
            return e;
        };
        auto ep256 = [&](int l, int c) {               // trunk layer l: two sign words
            BwdEpF<true> e;
            e.accum = false; e.slope = 0.01f;
            e.m[0] = mw[WAVE * (2 * l)]; e.m[1] = mw[WAVE * (2 * l + 1)];
            e.save = DZ(c, 256);
            return e;
        };
        f32x4 draw[1];
        draw[0] = *reinterpret_cast<const f32x4 *>(d_raw + p * D_RAW + 4 * q);
        f32x4 dC0[8], dS0[8];
        {   // colour branch: d_raw -> dC3 -> dC2 -> dC1 -> dC0 -> d grid features
            f32x4 cA[8], cB[8];
            auto e3 = ep128(16 + 3, act::C + 384);
            dense_ep<1, 0, 8, CHR(R_D3)>(cx, draw, nullptr, cA, e3);
            auto e2 = ep128(16 + 2, act::C + 256);
            dense_ep<8, 0, 8, CHR(R_D2)>(cx, cA, nullptr, cB, e2);
            auto e1 = ep128(16 + 1, act::C + 128);
            dense_ep<8, 0, 8, CHR(R_D1)>(cx, cB, nullptr, cA, e1);
            auto e0 = ep128(16 + 0, act::C + 0);
            dense_ep<8, 0, 8, CHR(R_GRIDF)>(cx, cA, nullptr, dC0, e0);
            BwdEpF<false> eg{false, 1.0f, {0u, 0u}, dgridf + p * 32 + 4 * q};
            f32x4 g2[2];
            dense_ep<8, 0, 2, CHR(R_SEGH)>(cx, dC0, nullptr, g2, eg);
        }
        {   // seg branch: d_raw -> dS3 -> dS2 -> dS1 -> dS0
            f32x4 sA[8], sB[8];
            auto e3 = ep128(20 + 3, act::S + 384);
            dense_ep<1, 0, 8, CHR(R_S3)>(cx, draw, nullptr, sA, e3);
            auto e2 = ep128(20 + 2, act::S + 256);
            dense_ep<8, 0, 8, CHR(R_S2)>(cx, sA, nullptr, sB, e2);
            auto e1 = ep128(20 + 1, act::S + 128);
            dense_ep<8, 0, 8, CHR(R_S1)>(cx, sB, nullptr, sA, e1);
            auto e0 = ep128(20 + 0, act::S + 0);
            dense_ep<8, 0, 8, CHR(R_FEATA)>(cx, sA, nullptr, dS0, e0);
        }
        f32x4 F[16], G[16];
        {   // d feat (no activation behind it)
            BwdEpF<false, false> ea{false, 1.0f, {0u, 0u}, nullptr};
            dense_ep<1, 0, 16, CHR(R_FEATB)>(cx, draw, nullptr, F, ea);
            BwdEpF<false> ef{true, 1.0f, {0u, 0u}, DZ(act::FEAT, 256)};
            dense_ep<8, 8, 16, CHR(R_FEAT)>(cx, dS0, dC0, F, ef);
        }
        {   // trunk: d feat -> dT7 -> ... -> dT3 (-> the encodings through the skip layer) -> dT2 -> dT1 -> dT0 (-> the encodings through layer 0)
            auto e7 = ep256(7, act::T + 7 * 256);
            dense_ep<16, 0, 16, CHR(R_T7)>(cx, F, nullptr, G, e7);
#pragma unroll 1
            for (int l = 6; l >= 3; --l) {             // layers T7..T4 (next chunks: T6, T5, T4, T3IN, all 32 KB) leave dT6..dT3
                auto el = ep256(l, act::T + l * 256);
                dense_ep<16, 0, 16, CHR(R_T6)>(cx, G, nullptr, F, el);
#pragma unroll
                for (int i = 0; i < 16; ++i) G[i] = F[i];
            }
            static_assert(CHR(R_T6) == CHR(R_T5) && CHR(R_T6) == CHR(R_T4) && CHR(R_T6) == CHR(R_T3IN), "rolled trunk layers");
            {
                BwdEpF<false> ei{false, 1.0f, {0u, 0u}, din_a + p * 96 + 4 * q};
                f32x4 d6[6];
                dense_ep<16, 0, 6, CHR(R_T3)>(cx, G, nullptr, d6, ei);
            }
            auto e2 = ep256(2, act::T + 2 * 256);
            dense_ep<16, 0, 16, CHR(R_T2)>(cx, G, nullptr, F, e2);
            auto e1 = ep256(1, act::T + 1 * 256);
            dense_ep<16, 0, 16, CHR(R_T1)>(cx, F, nullptr, G, e1);
            auto e0 = ep256(0, act::T + 0);
            dense_ep<16, 0, 16, CHR(R_T0IN)>(cx, G, nullptr, F, e0);
            {
                BwdEpF<false> ei{false, 1.0f, {0u, 0u}, din_b + p * 96 + 4 * q};
                f32x4 d6[6];
                dense_ep<16, 0, 6, CHR(R_RGBH)>(cx, F, nullptr, d6, ei);
            }
        }
#undef DZ
    }
}
static_assert(16 * (KB_XYZ + KB_AMB) == 96, "the encodings' gradient rows (din_a, din_b) are 96 floats");

// Deformation nets, backwards.  xwg (P,8): the seam gradient [dx'0 dx'1 dx'2 . dw0 dw1 . .]; actbuf: the saved activations (DX plane:
// tanh'); bits: the deformation sign planes (sbits::BD_*); dact: dZ planes WH, HH; g3, dw4 (P,4): the heads' pre-activation gradients
// [dx' (1 - dx^2) | 0], [dw | 0 0] -- the dY operands of the two final layers' weight-gradient jobs.
__global__ void __launch_bounds__(F32_THREADS, 2)
field_backward_chain_def_f32_kernel(const float *__restrict__ stream, long P, const float *__restrict__ xwg, const float *__restrict__ actbuf,
                                    const uint32_t *__restrict__ bits, float *__restrict__ dact, float *__restrict__ g3, float *__restrict__ dw4)
{
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    Ctx cx;
    start_stream(cx, stream, lds_f, DEF_FLOATS, CHD(D_HF));
    const int q = cx.q;
    const long ntiles = (P + F32_PTS_PER_WG - 1) / F32_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p_raw = tile * F32_PTS_PER_WG + cx.wave * F32_PTS_PER_WAVE + (cx.lane & 15);
        const long p = p_raw < P ? p_raw : P - 1;
        const bool on = p_raw < P;
        long Pq = P;
        asm volatile("" : "+s"(Pq));
        const uint32_t sb_lane = (uint32_t)(p * 4 + q);
        // (this lane's 12 sign words, parked in LDS as in the radiance kernel: hyper-sheet layer i -> i, warp-field layer i -> 6 + i)
        uint32_t *const mw = reinterpret_cast<uint32_t *>(lds_f + 2 * LDS_BUF_FLOATS) + cx.wave * (DEF_MASK_WORDS * WAVE) + cx.lane;
        {
            uint32_t w[DEF_MASK_WORDS];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                w[i] = bits[(long)(sbits::BD_HH + 4 * i) * Pq + sb_lane];
                w[6 + i] = bits[(long)(sbits::BD_WH + 4 * i) * Pq + sb_lane];
            }
#pragma unroll
            for (int k = 0; k < DEF_MASK_WORDS; ++k) mw[WAVE * k] = w[k];
        }
        auto epr = [&](int slot, int c, int w) {       // a ReLU layer of width w (64 or 128): one sign word per lane
            BwdEpF<true> e;
            e.accum = false; e.slope = 0.0f;
            e.m[0] = mw[WAVE * slot]; e.m[1] = 0u;
            e.save = dact + (long)c * Pq + p * (long)w + 4 * q;
            return e;
        };
        f32x4 hd_x[1], hd_w[1];
        {
            const f32x4 *r = reinterpret_cast<const f32x4 *>(xwg + p * 8);
            const f32x4 gx = r[0], gw = r[1];
            const float *dxp = actbuf + (long)act::DX * Pq + p * 16;
            const float d0 = dxp[0], d1 = dxp[1], d2 = dxp[2];
            const f32x4 t3 = f32x4{gx[0] * (1.0f - d0 * d0), gx[1] * (1.0f - d1 * d1), gx[2] * (1.0f - d2 * d2), 0.0f};     // x' = x + tanh(.) (models.py:304-305)
            const f32x4 tw = f32x4{gw[0], AMB_DIM > 1 ? gw[1] : 0.0f, 0.0f, 0.0f};
            const f32x4 z4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            hd_x[0] = q == 0 ? t3 : z4;                // gradients 0..3 of the head's k-block live in lane quarter 0
            hd_w[0] = q == 0 ? tw : z4;
            if (q == 0 && on) {
                *reinterpret_cast<f32x4 *>(g3 + p * 4) = t3;
                *reinterpret_cast<f32x4 *>(dw4 + p * 4) = tw;
            }
        }
        {   // hyper sheet: dw -> dG5 -> ... -> dG0
            f32x4 A[4], B[4];
            auto e5 = epr(5, act::HH + 64 * 5, 64);
            dense_ep<1, 0, 4, CHD(D_H5)>(cx, hd_w, nullptr, A, e5);
#pragma unroll 1
            for (int l = 4; l >= 1; --l) {             // layers H5..H2 (next chunks: H4, H3, H2, H1) leave dG4..dG1
                auto el = epr(l, act::HH + 64 * l, 64);
                dense_ep<4, 0, 4, CHD(D_H4)>(cx, A, nullptr, B, el);
#pragma unroll
                for (int i = 0; i < 4; ++i) A[i] = B[i];
            }
            static_assert(CHD(D_H4) == CHD(D_H3) && CHD(D_H4) == CHD(D_H2) && CHD(D_H4) == CHD(D_H1), "rolled hyper-sheet layers");
            auto e0 = epr(0, act::HH + 0, 64);
            dense_ep<4, 0, 4, CHD(D_WF)>(cx, A, nullptr, B, e0);
        }
        {   // warp field: dx' (1 - dx^2) -> dH5 -> ... -> dH0
            f32x4 A[8], B[8];
            auto e5 = epr(6 + 5, act::WH + 128 * 5, 128);
            dense_ep<1, 0, 8, CHD(D_W5)>(cx, hd_x, nullptr, A, e5);
#pragma unroll 1
            for (int l = 4; l >= 1; --l) {             // layers W5..W2 leave dH4..dH1
                auto el = epr(6 + l, act::WH + 128 * l, 128);
                dense_ep<8, 0, 8, CHD(D_W4)>(cx, A, nullptr, B, el);
#pragma unroll
                for (int i = 0; i < 8; ++i) A[i] = B[i];
            }
            static_assert(CHD(D_W4) == CHD(D_W3) && CHD(D_W4) == CHD(D_W2) && CHD(D_W4) == CHD(D_W1), "rolled warp-field layers");
            auto e0 = epr(6, act::WH + 0, 128);
            dense_ep<8, 0, 8, CHD(D_HF)>(cx, A, nullptr, B, e0);
        }
    }
}

}  // namespace bwf
}  // namespace SAHS_NS

using namespace SAHS_NS;
using namespace SAHS_NS::bwf;

// floats of the transposed fp32 stream of `part` (1 deformation nets, 2 radiance nets of one level)
extern "C" long sahs_bwd_chain_f32_stream_floats(int part) { return part == 1 ? DEF_FLOATS : RAD_FLOATS; }

extern "C" int sahs_bwd_chain_f32_pack_launch(const float *flat, float *stream_out, int level, int part, hipStream_t stream)
{
    const long total = part == 1 ? DEF_FLOATS : RAD_FLOATS;
    pack_bwd_stream_f32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(flat, stream_out, level, part);
    return (int)hipGetLastError();
}

template <class K, class... A>
static int launch_chain_f32(K kernel, long P, int num_cu, hipStream_t stream, A... args)
{
    if (P <= 0) return 0;
    const long ntiles = (P + F32_PTS_PER_WG - 1) / F32_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    constexpr int LDS_BYTES = CHAIN_LDS_BYTES;
    static sahs_once::Flags attr_set;       // (one per instantiation = per kernel)
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    kernel<<<grid, F32_THREADS, LDS_BYTES, stream>>>(args...);
    return (int)hipGetLastError();
}

extern "C" int sahs_bwd_chain_f32_rad_launch(const float *bstream, long P, const float *d_raw, const uint32_t *bits, float *dact, float *dgridf,
                                             float *din_a, float *din_b, int num_cu, hipStream_t stream)
{
    return launch_chain_f32(field_backward_chain_rad_f32_kernel, P, num_cu, stream, bstream, P, d_raw, bits, dact, dgridf, din_a, din_b);
}

extern "C" int sahs_bwd_chain_f32_def_launch(const float *bstream, long P, const float *xwg, const float *actbuf, const uint32_t *bits, float *dact,
                                             float *g3, float *dw4, int num_cu, hipStream_t stream)
{
    return launch_chain_f32(field_backward_chain_def_f32_kernel, P, num_cu, stream, bstream, P, xwg, actbuf, bits, dact, g3, dw4);
}
