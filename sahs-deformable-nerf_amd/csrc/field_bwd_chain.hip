// field_bwd_chain.hip -- the data-gradient chain of the field backward, sample-major, on the bf16 matrix pipe with split operands.
//
// Autograd of the reference's per-sample networks (modules.py:254-295 NeRFMLP, :371-390 WarpFieldMLP, :444-462 HyperSheetMLP, as driven by
// train_stage_rays_auto.py:437-499) needs, for every dense layer, the gradient of its pre-activation, dZ_l = (W_{l+1}^T dZ_{l+1}) * act'(z_l).
// Rounds 1-3 ran that as one GEMM launch per layer over all samples (field_bwd.hip: nn()), every dZ going through HBM twice.  Here ONE
// launch walks a whole net backwards for 32 samples per wave, exactly as the split-operand forward of field_bf16x3.hip walks it forwards
// (bf16x3_pipe.hpp: dense_x -- the 32x32 accumulator tile of one layer becomes the hi/lo B fragments of the next in registers), with
//   * transposed weight streams (A = W^T in MFMA fragment order, split into bf16 hi + lo by pack_bwd_stream_kernel once per walk),
//   * the (leaky-)ReLU derivative taken from the sign bits the saving forward wrote (sahs_layout.hpp: sbits; 1 bit per value instead of the
//     saved fp32 activation), applied as v_bfe_i32 + v_bfi_b32 on the value's way into the conversion,
//   * every dZ tile stored once, fp32, into the plane of its layer (the act:: layout: dZ of a layer sits where its activation sits in
//     the saved-activation buffer) -- the operands of the weight-gradient launch that follows (field_bwd.hip: gemm_tn_jobs_kernel).
// Same arithmetic as the per-layer split-operand GEMMs it replaces (hi*hi + hi*lo + lo*hi, fp32 accumulation), a different summation
// order.  AudioFaceModel only (SAHS_MODEL 0); the other models and SAHS_BWD_GEMM=f32 keep the per-layer walk.
#include <hip/hip_runtime.h>
#include <utility>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"
#include "bf16_pipe.hpp"
#include "bf16x3_pipe.hpp"

#if SAHS_MODEL != 0
#error "field_bwd_chain.hip is built for the AudioFaceModel only"
#endif

namespace SAHS_NS {
namespace bwc {
using namespace hx3;

// ---- the backward layer program ----------------------------------------------------------------------------------------------------
// A backward layer multiplies A = (part of) W^T: its output rows are INPUT features of the forward layer (W's columns col0 ..), its K
// index runs over the forward layer's OUTPUT features (W's rows).  Up to three K segments (dfeat sums three branches), up to two row
// ranges (the encodings' gradient: PE(x') columns, then PE(w) columns).  The heads read the 16-float d_raw row [drgb3 | dseg12 | dsigma]
// as a 32-feature block whose k = d_raw column: kshift places the head's weight rows (fc_seg: rows 0..11 at k = 3..14).  A head layer on its
// own runs over TWO blocks, the second all zeros: dense_x's counted LDS waits assume at least AP = 4 k-steps per tile.
struct SegB { long w_off[2]; int ld, kshift, krows, blocks; };
struct RowB { int rows, col0, valid; };
struct LayerB {
    int NT32, KB32, nseg; SegB seg[3];
    int nrow; RowB row[2];
    long stream_off;      // halfwords, in this part's stream
    int chunk_hw;
};
enum RadLayer { R_RGBH, R_D3, R_D2, R_D1, R_GRIDF, R_SEGH, R_S3, R_S2, R_S1, R_FEATIN, R_FEAT, R_T7, R_T6, R_T5, R_T4, R_T3IN, R_T3, R_T2, R_T1, R_T0IN, R_COUNT };
enum DefLayer { D_HF, D_H5, D_H4, D_H3, D_H2, D_H1, D_WF, D_W5, D_W4, D_W3, D_W2, D_W1, D_COUNT };
template <int N> struct ProgB { LayerB layer[N]; long stream_hw; };

constexpr LayerB mkb(int NT32, SegB s0, RowB r0, SegB s1 = {{0, 0}, 0, 0, 0, 0}, SegB s2 = {{0, 0}, 0, 0, 0, 0}, RowB r1 = {0, 0, 0})
{
    LayerB L{};
    L.NT32 = NT32;
    L.seg[0] = s0; L.seg[1] = s1; L.seg[2] = s2;
    L.nseg = 1 + (s1.blocks > 0) + (s2.blocks > 0);
    L.KB32 = s0.blocks + s1.blocks + s2.blocks;
    L.row[0] = r0; L.row[1] = r1;
    L.nrow = 1 + (r1.rows > 0);
    return L;
}
template <int N> constexpr void finish(ProgB<N> &P)
{
    long off = 0;
    for (int i = 0; i < N; ++i) {
        P.layer[i].stream_off = off;
        P.layer[i].chunk_hw = pick_GX(P.layer[i].KB32, P.layer[i].NT32) * P.layer[i].KB32 * 2048;
        off += (long)P.layer[i].NT32 * P.layer[i].KB32 * 2048;
    }
    P.stream_hw = off;
}
constexpr ProgB<R_COUNT> make_rad()
{
    ProgB<R_COUNT> P{};
    const FlatOffsets::Lvl &c = kFlat.lvl[0], &n = kFlat.lvl[1];
    auto sq = [](long w0, long w1, int ld, int krows) { return SegB{{w0, w1}, ld, 0, krows, (krows + 31) / 32}; };
    LayerB *L = P.layer;
    // colour branch, from its head back (modules.py:276-287)
    L[R_RGBH] = mkb(4, SegB{{c.rgb_w, n.rgb_w}, BR_H, 0, 3, 2}, RowB{BR_H, 0, BR_H});
    L[R_D3] = mkb(4, sq(c.dir_w[3], n.dir_w[3], BR_H, BR_H), RowB{BR_H, 0, BR_H});
    L[R_D2] = mkb(4, sq(c.dir_w[2], n.dir_w[2], BR_H, BR_H), RowB{BR_H, 0, BR_H});
    L[R_D1] = mkb(4, sq(c.dir_w[1], n.dir_w[1], BR_H, BR_H), RowB{BR_H, 0, BR_H});
    L[R_GRIDF] = mkb(2, sq(c.dir_w[0], n.dir_w[0], D_DIR_IN, BR_H), RowB{D_GRID, TR_H + D_DIR, D_GRID});       // d grid features (tile 1: padding)
    // seg branch (modules.py:289-294)
    L[R_SEGH] = mkb(4, SegB{{c.segout_w, n.segout_w}, BR_H, 3, N_SEG, 2}, RowB{BR_H, 0, BR_H});
    L[R_S3] = mkb(4, sq(c.seg_w[3], n.seg_w[3], BR_H, BR_H), RowB{BR_H, 0, BR_H});
    L[R_S2] = mkb(4, sq(c.seg_w[2], n.seg_w[2], BR_H, BR_H), RowB{BR_H, 0, BR_H});
    L[R_S1] = mkb(4, sq(c.seg_w[1], n.seg_w[1], BR_H, BR_H), RowB{BR_H, 0, BR_H});
    // d feat = W_S0^T dS0 + W_D0[:, :256]^T dC0 + w_alpha dsigma
    L[R_FEATIN] = mkb(8, sq(c.seg_w[0], n.seg_w[0], TR_H, BR_H), RowB{TR_H, 0, TR_H}, sq(c.dir_w[0], n.dir_w[0], D_DIR_IN, BR_H),
                      SegB{{c.alpha_w, n.alpha_w}, TR_H, 15, 1, 1});
    // trunk (modules.py:267-274), skip layer 3: [h | PE(x') | PE(w) | pose]
    L[R_FEAT] = mkb(8, sq(c.feat_w, n.feat_w, TR_H, TR_H), RowB{TR_H, 0, TR_H});
    for (int i = 7; i >= 4; --i) L[R_T7 + (7 - i)] = mkb(8, sq(c.xyz_w[i], n.xyz_w[i], TR_H, TR_H), RowB{TR_H, 0, TR_H});
    L[R_T3IN] = mkb(4, sq(c.xyz_w[3], n.xyz_w[3], TR_H + D_TR_IN, TR_H), RowB{16 * KB_XYZ, TR_H, D_XYZ}, SegB{{0, 0}, 0, 0, 0, 0}, SegB{{0, 0}, 0, 0, 0, 0},
                    RowB{16 * KB_AMB, TR_H + D_XYZ, D_AMB});                                                      // d [PE(x') | PE(w)] (tile 3: padding)
    L[R_T3] = mkb(8, sq(c.xyz_w[3], n.xyz_w[3], TR_H + D_TR_IN, TR_H), RowB{TR_H, 0, TR_H});
    L[R_T2] = mkb(8, sq(c.xyz_w[2], n.xyz_w[2], TR_H, TR_H), RowB{TR_H, 0, TR_H});
    L[R_T1] = mkb(8, sq(c.xyz_w[1], n.xyz_w[1], TR_H, TR_H), RowB{TR_H, 0, TR_H});
    L[R_T0IN] = mkb(4, sq(c.xyz_w[0], n.xyz_w[0], D_TR_IN, TR_H), RowB{16 * KB_XYZ, 0, D_XYZ}, SegB{{0, 0}, 0, 0, 0, 0}, SegB{{0, 0}, 0, 0, 0, 0},
                    RowB{16 * KB_AMB, D_XYZ, D_AMB});
    finish(P);
    return P;
}
constexpr ProgB<D_COUNT> make_def()
{
    ProgB<D_COUNT> P{};
    const FlatOffsets &f = kFlat;
    auto sq = [](long w, int ld, int krows) { return SegB{{w, w}, ld, 0, krows, (krows + 31) / 32}; };
    LayerB *L = P.layer;
    // hyper sheet (modules.py:444-462): w = fc_ambient(g5); skip layer 4: [g | PE(x) | driving | pose]
    L[D_HF] = mkb(2, SegB{{f.hyp_fw, f.hyp_fw}, HYP_H, 0, AMB_DIM, 2}, RowB{HYP_H, 0, HYP_H});
    L[D_H5] = mkb(2, sq(f.hyp_w[5], HYP_H, HYP_H), RowB{HYP_H, 0, HYP_H});
    L[D_H4] = mkb(2, sq(f.hyp_w[4], HYP_H + D_DEF_IN, HYP_H), RowB{HYP_H, 0, HYP_H});
    L[D_H3] = mkb(2, sq(f.hyp_w[3], HYP_H, HYP_H), RowB{HYP_H, 0, HYP_H});
    L[D_H2] = mkb(2, sq(f.hyp_w[2], HYP_H, HYP_H), RowB{HYP_H, 0, HYP_H});
    L[D_H1] = mkb(2, sq(f.hyp_w[1], HYP_H, HYP_H), RowB{HYP_H, 0, HYP_H});
    // warp field (modules.py:371-390): dx = tanh(fc_final(h5))
    L[D_WF] = mkb(4, SegB{{f.warp_fw, f.warp_fw}, WARP_H, 0, 3, 2}, RowB{WARP_H, 0, WARP_H});
    L[D_W5] = mkb(4, sq(f.warp_w[5], WARP_H, WARP_H), RowB{WARP_H, 0, WARP_H});
    L[D_W4] = mkb(4, sq(f.warp_w[4], WARP_H + D_DEF_IN, WARP_H), RowB{WARP_H, 0, WARP_H});
    L[D_W3] = mkb(4, sq(f.warp_w[3], WARP_H, WARP_H), RowB{WARP_H, 0, WARP_H});
    L[D_W2] = mkb(4, sq(f.warp_w[2], WARP_H, WARP_H), RowB{WARP_H, 0, WARP_H});
    L[D_W1] = mkb(4, sq(f.warp_w[1], WARP_H, WARP_H), RowB{WARP_H, 0, WARP_H});
    finish(P);
    return P;
}
constexpr ProgB<R_COUNT> kRad = make_rad();
constexpr ProgB<D_COUNT> kDef = make_def();
__device__ const ProgB<R_COUNT> dRad = make_rad();
__device__ const ProgB<D_COUNT> dDef = make_def();
constexpr long RAD_HW = kRad.stream_hw, DEF_HW = kDef.stream_hw;      // halfwords of a level's radiance stream / of the deformation stream
static_assert(RAD_HW % 8 == 0 && DEF_HW % 8 == 0, "16-byte granules");

// ---- transposed, split stream of one part: [layer][tile32][k-step][hi 64 x 8 | lo 64 x 8]; lane 32 h + i of k-step (block b, step st)
// holds A[32 t + i][32 b + 16 st + 8 (j >> 2) + 4 h + (j & 3)], j = 0..7 (the B-operand order of bf16x3_pipe.hpp) -----------------------
template <int N>
__device__ __forceinline__ void pack_one(const ProgB<N> &Pg, const float *__restrict__ flat, unsigned short *__restrict__ out, int level, long e)
{
    const long hw = e * 16;                                // one thread: the hi AND lo fragment granule of one lane (2 x 8 halfwords)
    int li = 0;
    while (li + 1 < N && Pg.layer[li + 1].stream_off <= hw) ++li;
    const LayerB &L = Pg.layer[li];
    const long w = hw - L.stream_off;
    const int per_tile = L.KB32 * 2048;
    const int t = (int)(w / per_tile);
    const int rem = (int)(w - (long)t * per_tile);
    const int f = rem >> 10, lane = (rem & 1023) >> 4;     // fragment pair f = 2 b + st; 16 halfwords (hi 8 + lo 8) per lane and pair
    const int b = f >> 1, st = f & 1, i = lane & 31, h = lane >> 5;
    const int row = 32 * t + i;
    int col = -1, r0 = 0;
    for (int rs = 0; rs < L.nrow; ++rs) {
        if (row >= r0 && row < r0 + L.row[rs].rows) { if (row - r0 < L.row[rs].valid) col = L.row[rs].col0 + (row - r0); break; }
        r0 += L.row[rs].rows;
    }
    int bb = b, sg = -1;
    for (int s = 0; s < L.nseg; ++s) {
        if (bb < L.seg[s].blocks) { sg = s; break; }
        bb -= L.seg[s].blocks;
    }
    unsigned short vh[8], vl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 32 * bb + 16 * st + 8 * (j >> 2) + 4 * h + (j & 3);
        float x = 0.0f;
        if (sg >= 0 && col >= 0) {
            const int k = c - L.seg[sg].kshift;
            if (k >= 0 && k < L.seg[sg].krows) x = flat[L.seg[sg].w_off[level] + (long)k * L.seg[sg].ld + col];
        }
        const __bf16 hi = (__bf16)x;
        vh[j] = __builtin_bit_cast(unsigned short, hi);
        vl[j] = __builtin_bit_cast(unsigned short, (__bf16)(x - (float)hi));
    }
    unsigned short *dst = out + L.stream_off + (long)t * per_tile + (long)f * 1024 + lane * 8;
    uint4 q;
    q.x = vh[0] | ((unsigned)vh[1] << 16); q.y = vh[2] | ((unsigned)vh[3] << 16);
    q.z = vh[4] | ((unsigned)vh[5] << 16); q.w = vh[6] | ((unsigned)vh[7] << 16);
    *reinterpret_cast<uint4 *>(dst) = q;
    q.x = vl[0] | ((unsigned)vl[1] << 16); q.y = vl[2] | ((unsigned)vl[3] << 16);
    q.z = vl[4] | ((unsigned)vl[5] << 16); q.w = vl[6] | ((unsigned)vl[7] << 16);
    *reinterpret_cast<uint4 *>(dst + 512) = q;
}
__global__ void __launch_bounds__(256) pack_bwd_stream_kernel(const float *__restrict__ flat, unsigned short *__restrict__ out, int level, int part)
{
    const long total = (part == 1 ? DEF_HW : RAD_HW) / 16;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        if (part == 1) pack_one(dDef, flat, out, 0, e);
        else pack_one(dRad, flat, out, level, e);
    }
}

// ---- the backward epilogue: derivative mask from the forward's sign bits, dZ stored once --------------------------------------------
// This lane (sample, h) of 32-row tile TILE holds, as value U = 4 g + i, feature 32 TILE + 8 g + 4 h + i: bit 8 TILE + 4 (g >> 1) + i of
// sign word q = h (g even) or q = 2 + h (g odd) -- sahs_layout.hpp: sbits.  MASKED = false: no activation behind this gradient (d feat, the
// encodings' gradient).  Tiles >= NVALID are padding (not stored).  dst = (uniform plane base, this lane's byte offset of its row + 4 h).
// Where a finished group of four values goes: NOT straight to memory -- the accumulator layout would make every store instruction write 32
// bytes into each of 32 rows (quarter cache lines; measured: the chain kernel then runs at 2.4 TB/s of stores, compute and stores adding up) --
// but through a per-wave 32 x 32 staging tile in LDS: the lane writes its four values (ds_write_b128), and once the tile's fourth group is in,
// the wave reads the tile back row-major and stores it as WHOLE 128-byte lines: lane l -> rows (l >> 3) + 8 i, i = 0..3, 16 bytes (l & 7).
struct StageCtx {
    uint32_t wr, rd;          // LDS byte addresses: this lane's row (+16 h) for writing; row l >> 3, piece l & 7 for reading back
    uint32_t pc;              // 16 (l & 7): the piece's byte offset inside a 128-byte tile row
    uint32_t prow[4];         // the sample index of read-back row i (clamped to P - 1 like the lane's own sample)
    uint32_t np;              // P
};
constexpr int STAGE_ROW_BYTES = 144;                                      // 32 floats + 4 of padding: 16-byte aligned rows, spread over the banks
constexpr int STAGE_WAVE_BYTES = 32 * STAGE_ROW_BYTES;

template <bool MASKED, int NVALID, int WIDTH>
struct BwdEp {
#if defined(SAHS_DIAG) && (defined(SAHS_BWC_NOSTORE) || defined(SAHS_BWC_NOGSTORE))
    static constexpr bool stores(int) { return false; }
#else
    static constexpr bool stores(int tile) { return tile < NVALID; }      // (the counted wait of bf16x3_pipe.hpp: at least one store per group)
#endif
    float slope;
    uint32_t mh[2], m2[2];
    float *base;              // the layer's dZ plane (uniform)
    StageCtx sc;
    template <int TILE, int U> __device__ __forceinline__ float value(float v, float m) const
    {
        if constexpr (!MASKED) {
            return v;
        } else {
            constexpr int g = U >> 2, i = U & 3, bit = 8 * TILE + 4 * (g >> 1) + i;
            const uint32_t w = (g & 1) ? m2[bit >> 5] : mh[bit >> 5];
            const uint32_t t = (uint32_t)((int32_t)(w << (31 - (bit & 31))) >> 31);      // v_bfe_i32: all ones where the activation was > 0
            if (slope == 0.0f) return __builtin_bit_cast(float, t & __builtin_bit_cast(uint32_t, v));
            return __builtin_bit_cast(float, (t & __builtin_bit_cast(uint32_t, v)) | (~t & __builtin_bit_cast(uint32_t, m)));      // v_bfi_b32
        }
    }
    template <int TILE, int G> __device__ __forceinline__ void done4(const float (&r)[4], uint32_t (&)[2]) const
    {
#if defined(SAHS_DIAG) && defined(SAHS_BWC_NOSTORE)      // timing-only: results wrong by construction
        asm volatile("" :: "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
#else
        if constexpr (TILE < NVALID) {
            *(lds_f4_t)(uintptr_t)(sc.wr + 32 * G) = f32x4{r[0], r[1], r[2], r[3]};
            if constexpr (G == 3) {
                f32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *(lds_f4_t)(uintptr_t)(sc.rd + i * 8 * STAGE_ROW_BYTES);
#if defined(SAHS_DIAG) && defined(SAHS_X3_NOREADBACK)      // timing-only (results wrong by construction): the stores without waiting for the read-back
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = f32x4{r[0], r[1], r[2], r[3]};
#endif
#if defined(SAHS_DIAG) && defined(SAHS_BWC_NOGSTORE)      // timing-only: the staging round trip without the global stores
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(v[i]));
#elif defined(SAHS_DIAG) && defined(SAHS_BWC_TILEMAJOR)      // timing-only: the same bytes as 4 KB contiguous runs per wave and tile (the weight-gradient launch would read garbage)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(base) + (size_t)TILE * (sc.np * 128u) + (sc.prow[i] * 128u + sc.pc)) = v[i];
#elif defined(SAHS_DIAG) && defined(SAHS_BWC_PLAINSTORE)      // A/B: default cache policy instead of non-temporal
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(base) + (sc.prow[i] * (uint32_t)(WIDTH * 4) + sc.pc + 128u * TILE)) = v[i];
#else
                // non-temporal: 4.4 GB of dZ per launch stream THROUGH the L2 that holds the 3 MB weight stream every workgroup re-reads
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(base) + (sc.prow[i] * (uint32_t)(WIDTH * 4) + sc.pc + 128u * TILE)));
#endif
            }
        }
#endif
    }
};
struct NoEp {      // (the pending-tile policy of a layer that has none)
    static constexpr bool stores(int) { return false; }
    float slope;
    template <int TILE, int U> __device__ __forceinline__ float value(float v, float) const { return v; }
    template <int TILE, int G> __device__ __forceinline__ void done4(const float (&)[4], uint32_t (&)[2]) const {}
};

// the last tile of a chain's last layer: nothing follows that would convert it under its MFMAs
template <int PTILE, class PEP>
__device__ __forceinline__ void flush_x(St &st, const PEP pep)
{
    Blk dummy;
    pack_ticks<0, PACK_TICKS, PTILE>(st.acc[1], dummy, pep, st.ps);
    fence();
}

constexpr int BWC_ZERO_BYTES_ = 2048;
constexpr int BWC_STAGE_OFF_ = LDS_BIAS_BYTE_OFF + BWC_ZERO_BYTES_;
// this lane's part in staging its wave's tiles (StageCtx): the wave's 32 samples are consecutive, p0 = 128 tile + 32 wave
__device__ __forceinline__ StageCtx make_stage(const char *lds, int wave, int lane, long tile, long P)
{
    StageCtx sc;
    const uint32_t base = lds_addr_of(lds) + BWC_STAGE_OFF_ + (uint32_t)wave * STAGE_WAVE_BYTES;
    sc.wr = base + (uint32_t)(lane & 31) * STAGE_ROW_BYTES + 16u * (lane >> 5);
    sc.rd = base + (uint32_t)(lane >> 3) * STAGE_ROW_BYTES + 16u * (lane & 7);
    sc.pc = 16u * (lane & 7);
    const long p0 = tile * X_PTS_PER_WG + wave * X_PTS_PER_WAVE + (lane >> 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) sc.prow[i] = (uint32_t)(p0 + 8 * i < P ? p0 + 8 * i : P - 1);
    sc.np = (uint32_t)P;
    return sc;
}

constexpr int BWC_ZERO_BYTES = 2048;                                       // the "bias" of every backward layer: zeros
constexpr int BWC_STAGE_BYTE_OFF = LDS_BIAS_BYTE_OFF + BWC_ZERO_BYTES;
static_assert(BWC_STAGE_BYTE_OFF == BWC_STAGE_OFF_, "staging tiles behind the zero bias page");
constexpr int BWC_LDS_BYTES = BWC_STAGE_BYTE_OFF + (X_THREADS / WAVE) * STAGE_WAVE_BYTES;
static_assert(BWC_LDS_BYTES <= 160 * 1024, "LDS budget");

// a (P,4)/(P,16) gradient row as the one 32-feature block of a head layer: k-step 0 = columns 4 h .. 4 h + 3 and 8 + 4 h .. 8 + 4 h + 3
__device__ __forceinline__ void head_block(const f32x4 a, const f32x4 b, Blk &o)
{
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        uint32_t hi, lo;
        split_pair(v[2 * jp], v[2 * jp + 1], hi, lo);
        o.s[0][jp] = hi; o.l[0][jp] = lo;
    }
    o.s[1] = u32x4{0u, 0u, 0u, 0u};
    o.l[1] = u32x4{0u, 0u, 0u, 0u};
}

__device__ __forceinline__ void zero_block(Blk &o)
{
    o.s[0] = o.s[1] = o.l[0] = o.l[1] = u32x4{0u, 0u, 0u, 0u};
}

#define CHR(id) (kRad.layer[id].chunk_hw)
#define CHD(id) (kDef.layer[id].chunk_hw)

// Radiance nets of one level, backwards.  d_raw (P,16); bits: the radiance sign planes (sbits::BR_*); dact: plane c of the act:: table at
// dact + c * P (written: C, S, FEAT, T planes); dgridf (P,32); din_a, din_b (P,96): the encodings' gradient through the skip layer and
// through layer 0 (summed by encode_backward).
__global__ void __launch_bounds__(X_THREADS, 1)
field_backward_chain_rad_kernel(const unsigned short *__restrict__ stream, long P, const float *__restrict__ d_raw, const uint32_t *__restrict__ bits,
                                float *__restrict__ dact, float *__restrict__ dgridf, float *__restrict__ din_a, float *__restrict__ din_b)
{
    extern __shared__ __attribute__((aligned(16))) char lds_x[];
    Ctx cx;
    cx.stream = stream;
    cx.lds = lds_x;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.h = cx.lane >> 5;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = cx.h, col = cx.lane & 31;
    {
        float *zl = reinterpret_cast<float *>(lds_x + LDS_BIAS_BYTE_OFF);
        for (int i = threadIdx.x; i < BWC_ZERO_BYTES / 4; i += X_THREADS) zl[i] = 0.0f;
        cx.wrap_at = (uint32_t)RAD_HW;
        cx.wrap_to = 0u;
        cx.off = 0u;
        cx.prepare(CHR(R_RGBH), 0);
#pragma unroll
        for (int pc = 0; pc < (CHR(R_RGBH) + PIECE_HW - 1) / PIECE_HW; ++pc) cx.issue_piece(pc);
        __syncthreads();
    }
    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cx.refresh_bias_base();
        asm volatile("" : "+s"(cx.off));      // (chunk addresses are loop-invariant: keep them from being hoisted and spilled)
        long Pq = P;
        asm volatile("" : "+s"(Pq));          // (and the ~30 plane bases c * P)
        St st;
        const long p_raw = tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE + col;
        const long p = p_raw < P ? p_raw : P - 1;      // lanes past the end redo the last sample: identical values to identical addresses
        const uint32_t pl = (uint32_t)p;
        const StageCtx sc = make_stage(lds_x, cx.wave, cx.lane, tile, P);
        // policies: sign words of plane (word offset b, NW words per q) + the dZ plane at act:: column c
        auto ep128 = [&](int b, int c) {
            BwdEp<true, 4, 128> e;
            e.slope = 0.01f;
            const uint32_t *a = bits + (long)b * Pq + ((long)pl * 4 + h);
#if defined(SAHS_DIAG) && defined(SAHS_BWC_NOMASK)       // timing-only
            e.mh[0] = pl; e.m2[0] = pl + h; (void)a;
#else
            e.mh[0] = a[0]; e.m2[0] = a[2];
#endif
            e.mh[1] = 0u; e.m2[1] = 0u;
            e.base = dact + (long)c * Pq;
            e.sc = sc;
            return e;
        };
        auto ep256 = [&](int b, int c) {
            BwdEp<true, 8, 256> e;
            e.slope = 0.01f;
            const uint32_t *a = bits + (long)b * Pq + ((long)pl * 4 + h) * 2;
#if defined(SAHS_DIAG) && defined(SAHS_BWC_NOMASK)
            e.mh[0] = pl; e.mh[1] = pl ^ 5u; e.m2[0] = pl + h; e.m2[1] = pl * 3u; (void)a;
#else
            e.mh[0] = a[0]; e.mh[1] = a[1]; e.m2[0] = a[4]; e.m2[1] = a[5];
#endif
            e.base = dact + (long)c * Pq;
            e.sc = sc;
            return e;
        };
        Blk draw[2];
        {
            const f32x4 *r = reinterpret_cast<const f32x4 *>(d_raw + p * D_RAW);
            head_block(r[h], r[2 + h], draw[0]);
            zero_block(draw[1]);
        }
        Blk dC0[4], dS0[4];
        {   // colour branch: d_raw -> dC3 -> dC2 -> dC1 -> dC0 (-> d grid features)
            Blk cA[4];
            auto e3 = ep128(sbits::BR_C + 12, act::C + 384);
            auto e2 = ep128(sbits::BR_C + 8, act::C + 256);
            dense_x<2, 0, 0, 4, CHR(R_D3), false>(cx, st, draw, nullptr, nullptr, cA, 0, e3, NoEp{1.0f});
            auto e1 = ep128(sbits::BR_C + 4, act::C + 128);
            dense_x<4, 0, 0, 4, CHR(R_D2), true, 3>(cx, st, cA, nullptr, nullptr, dC0, 0, e2, e3);
            auto e0 = ep128(sbits::BR_C + 0, act::C + 0);
            dense_x<4, 0, 0, 4, CHR(R_D1), true, 3>(cx, st, dC0, nullptr, nullptr, cA, 0, e1, e2);
            dense_x<4, 0, 0, 4, CHR(R_GRIDF), true, 3>(cx, st, cA, nullptr, nullptr, dC0, 0, e0, e1);
            BwdEp<false, 1, 32> eg;
            eg.slope = 1.0f; eg.base = dgridf; eg.sc = sc;
            Blk dummy[2];
            dense_x<4, 0, 0, 2, CHR(R_SEGH), true, 3>(cx, st, dC0, nullptr, nullptr, dummy, 0, eg, e0);
        }
        BwdEp<true, 4, 128> es0;
        {   // seg branch: d_raw -> dS3 -> dS2 -> dS1 -> dS0
            Blk sA[4];
            auto e3 = ep128(sbits::BR_S + 12, act::S + 384);
            auto e2 = ep128(sbits::BR_S + 8, act::S + 256);
            dense_x<2, 0, 0, 4, CHR(R_S3), false>(cx, st, draw, nullptr, nullptr, sA, 0, e3, NoEp{1.0f});
            auto e1 = ep128(sbits::BR_S + 4, act::S + 128);
            dense_x<4, 0, 0, 4, CHR(R_S2), true, 3>(cx, st, sA, nullptr, nullptr, dS0, 0, e2, e3);
            es0 = ep128(sbits::BR_S + 0, act::S + 0);
            dense_x<4, 0, 0, 4, CHR(R_S1), true, 3>(cx, st, dS0, nullptr, nullptr, sA, 0, e1, e2);
            dense_x<4, 0, 0, 4, CHR(R_FEATIN), true, 3>(cx, st, sA, nullptr, nullptr, dS0, 0, es0, e1);
        }
        Blk F[8], G[8];
        {   // d feat (no activation behind it), then the trunk
            BwdEp<false, 8, 256> ef;
            ef.slope = 1.0f; ef.base = dact + (long)act::FEAT * Pq; ef.sc = sc;
            auto e7 = ep256(sbits::BR_T + 8 * 7, act::T + 7 * 256);
            dense_x<4, 4, 1, 8, CHR(R_FEAT), true, 3>(cx, st, dS0, dC0, draw, F, 0, ef, es0);
            auto e6 = ep256(sbits::BR_T + 8 * 6, act::T + 6 * 256);
            dense_x<8, 0, 0, 8, CHR(R_T7), true, 7>(cx, st, F, nullptr, nullptr, G, 0, e7, ef);
            auto e5 = ep256(sbits::BR_T + 8 * 5, act::T + 5 * 256);
            dense_x<8, 0, 0, 8, CHR(R_T6), true, 7>(cx, st, G, nullptr, nullptr, F, 0, e6, e7);
            auto e4 = ep256(sbits::BR_T + 8 * 4, act::T + 4 * 256);
            dense_x<8, 0, 0, 8, CHR(R_T5), true, 7>(cx, st, F, nullptr, nullptr, G, 0, e5, e6);
            auto e3 = ep256(sbits::BR_T + 8 * 3, act::T + 3 * 256);
            dense_x<8, 0, 0, 8, CHR(R_T4), true, 7>(cx, st, G, nullptr, nullptr, F, 0, e4, e5);
            auto e2 = ep256(sbits::BR_T + 8 * 2, act::T + 2 * 256);
            dense_x<8, 0, 0, 8, CHR(R_T3IN), true, 7>(cx, st, F, nullptr, nullptr, G, 0, e3, e4);      // G = dT3 (last tile pending)
            BwdEp<false, 3, 96> ea;
            ea.slope = 1.0f; ea.base = din_a; ea.sc = sc;
            {
                Blk dummy[4];
                dense_x<8, 0, 0, 4, CHR(R_T3), true, 7>(cx, st, G, nullptr, nullptr, dummy, 0, ea, e3);  // d [PE(x') | PE(w)] through the skip layer
            }
            auto e1 = ep256(sbits::BR_T + 8 * 1, act::T + 1 * 256);
            dense_x<8, 0, 0, 8, CHR(R_T2), false>(cx, st, G, nullptr, nullptr, F, 0, e2, NoEp{1.0f});
            auto e0 = ep256(sbits::BR_T + 0, act::T + 0);
            dense_x<8, 0, 0, 8, CHR(R_T1), true, 7>(cx, st, F, nullptr, nullptr, G, 0, e1, e2);
            dense_x<8, 0, 0, 8, CHR(R_T0IN), true, 7>(cx, st, G, nullptr, nullptr, F, 0, e0, e1);      // F = dT0
            BwdEp<false, 3, 96> eb;
            eb.slope = 1.0f; eb.base = din_b; eb.sc = sc;
            {
                Blk dummy[4];
                dense_x<8, 0, 0, 4, CHR(R_RGBH), true, 7>(cx, st, F, nullptr, nullptr, dummy, 0, eb, e0);
            }
        }
    }
}

// Deformation nets, backwards.  xwg (P,8): the seam gradient [dx'0 dx'1 dx'2 . dw0 dw1 . .]; actbuf: the saved activations (DX plane:
// tanh'); bits: the deformation sign planes (sbits::BD_*); dact: dZ planes WH, HH; g3, dw4 (P,4): the heads' pre-activation gradients
// [dx' (1 - dx^2) | 0], [dw | 0 0] -- the dY operands of the two final layers' weight-gradient jobs.
__global__ void __launch_bounds__(X_THREADS, 1)
field_backward_chain_def_kernel(const unsigned short *__restrict__ stream, long P, const float *__restrict__ xwg, const float *__restrict__ actbuf,
                                const uint32_t *__restrict__ bits, float *__restrict__ dact, float *__restrict__ g3, float *__restrict__ dw4)
{
    extern __shared__ __attribute__((aligned(16))) char lds_x[];
    Ctx cx;
    cx.stream = stream;
    cx.lds = lds_x;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.h = cx.lane >> 5;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = cx.h, col = cx.lane & 31;
    {
        float *zl = reinterpret_cast<float *>(lds_x + LDS_BIAS_BYTE_OFF);
        for (int i = threadIdx.x; i < BWC_ZERO_BYTES / 4; i += X_THREADS) zl[i] = 0.0f;
        cx.wrap_at = (uint32_t)DEF_HW;
        cx.wrap_to = 0u;
        cx.off = 0u;
        cx.prepare(CHD(D_HF), 0);
#pragma unroll
        for (int pc = 0; pc < (CHD(D_HF) + PIECE_HW - 1) / PIECE_HW; ++pc) cx.issue_piece(pc);
        __syncthreads();
    }
    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cx.refresh_bias_base();
        asm volatile("" : "+s"(cx.off));
        long Pq = P;
        asm volatile("" : "+s"(Pq));
        St st;
        const long p_raw = tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE + col;
        const long p = p_raw < P ? p_raw : P - 1;
        const uint32_t pl = (uint32_t)p;
        const StageCtx sc = make_stage(lds_x, cx.wave, cx.lane, tile, P);
        auto fill = [&](auto &e, int b, int c) {       // one sign word per q for the 128- and the 64-wide nets alike
            e.slope = 0.0f;
            const uint32_t *a = bits + (long)b * Pq + ((long)pl * 4 + h);
            e.mh[0] = a[0]; e.m2[0] = a[2]; e.mh[1] = 0u; e.m2[1] = 0u;
            e.base = dact + (long)c * Pq;
            e.sc = sc;
        };
        auto epw = [&](int b, int c) { BwdEp<true, 4, 128> e; fill(e, b, c); return e; };
        Blk hd_w[2], hd_x[2];
        {
            const f32x4 *r = reinterpret_cast<const f32x4 *>(xwg + p * 8);
            const f32x4 gx = r[0], gw = r[1];
            const float *dxp = actbuf + (long)act::DX * Pq + p * 16;
            const float d0 = dxp[0], d1 = dxp[1], d2 = dxp[2];
            const f32x4 t3 = f32x4{gx[0] * (1.0f - d0 * d0), gx[1] * (1.0f - d1 * d1), gx[2] * (1.0f - d2 * d2), 0.0f};     // x' = x + tanh(.) (models.py:304-305)
            const f32x4 tw = f32x4{gw[0], AMB_DIM > 1 ? gw[1] : 0.0f, 0.0f, 0.0f};
            const f32x4 z4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            head_block(h == 0 ? t3 : z4, z4, hd_x[0]);
            head_block(h == 0 ? tw : z4, z4, hd_w[0]);
            zero_block(hd_x[1]);
            zero_block(hd_w[1]);
            if (h == 0) {
                *reinterpret_cast<f32x4 *>(g3 + p * 4) = t3;
                *reinterpret_cast<f32x4 *>(dw4 + p * 4) = tw;
            }
        }
        {   // hyper sheet: dw -> dG5 -> ... -> dG0 (the 64-wide layers: two 32-row tiles)
            Blk A[2], B[2];
            auto mk = [&](int i) { BwdEp<true, 2, 64> e; fill(e, sbits::BD_HH + 4 * i, act::HH + 64 * i); return e; };
            auto e5 = mk(5);
            auto e4 = mk(4);
            dense_x<2, 0, 0, 2, CHD(D_H5), false>(cx, st, hd_w, nullptr, nullptr, A, 0, e5, NoEp{1.0f});
            auto e3 = mk(3);
            dense_x<2, 0, 0, 2, CHD(D_H4), true, 1>(cx, st, A, nullptr, nullptr, B, 0, e4, e5);
            auto e2 = mk(2);
            dense_x<2, 0, 0, 2, CHD(D_H3), true, 1>(cx, st, B, nullptr, nullptr, A, 0, e3, e4);
            auto e1 = mk(1);
            dense_x<2, 0, 0, 2, CHD(D_H2), true, 1>(cx, st, A, nullptr, nullptr, B, 0, e2, e3);
            auto e0 = mk(0);
            dense_x<2, 0, 0, 2, CHD(D_H1), true, 1>(cx, st, B, nullptr, nullptr, A, 0, e1, e2);
            dense_x<2, 0, 0, 2, CHD(D_WF), true, 1>(cx, st, A, nullptr, nullptr, B, 0, e0, e1);
            flush_x<1>(st, e0);
        }
        {   // warp field: dx' (1 - dx^2) -> dH5 -> ... -> dH0
            Blk A[4], B[4];
            auto e5 = epw(sbits::BD_WH + 4 * 5, act::WH + 128 * 5);
            auto e4 = epw(sbits::BD_WH + 4 * 4, act::WH + 128 * 4);
            dense_x<2, 0, 0, 4, CHD(D_W5), false>(cx, st, hd_x, nullptr, nullptr, A, 0, e5, NoEp{1.0f});
            auto e3 = epw(sbits::BD_WH + 4 * 3, act::WH + 128 * 3);
            dense_x<4, 0, 0, 4, CHD(D_W4), true, 3>(cx, st, A, nullptr, nullptr, B, 0, e4, e5);
            auto e2 = epw(sbits::BD_WH + 4 * 2, act::WH + 128 * 2);
            dense_x<4, 0, 0, 4, CHD(D_W3), true, 3>(cx, st, B, nullptr, nullptr, A, 0, e3, e4);
            auto e1 = epw(sbits::BD_WH + 4 * 1, act::WH + 128 * 1);
            dense_x<4, 0, 0, 4, CHD(D_W2), true, 3>(cx, st, A, nullptr, nullptr, B, 0, e2, e3);
            auto e0 = epw(sbits::BD_WH + 0, act::WH + 0);
            dense_x<4, 0, 0, 4, CHD(D_W1), true, 3>(cx, st, B, nullptr, nullptr, A, 0, e1, e2);
            dense_x<4, 0, 0, 4, CHD(D_HF), true, 3>(cx, st, A, nullptr, nullptr, B, 0, e0, e1);
            flush_x<3>(st, e0);
        }
    }
}

}  // namespace bwc
}  // namespace SAHS_NS

using namespace SAHS_NS;
using namespace SAHS_NS::bwc;

// halfwords of the transposed stream of `part` (1 deformation nets, 2 radiance nets of one level)
extern "C" long sahs_bwd_chain_stream_hw(int part) { return part == 1 ? DEF_HW : RAD_HW; }

extern "C" int sahs_bwd_chain_pack_launch(const float *flat, void *stream_out, int level, int part, hipStream_t stream)
{
    const long total = (part == 1 ? DEF_HW : RAD_HW) / 16;
    pack_bwd_stream_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(flat, reinterpret_cast<unsigned short *>(stream_out), level, part);
    return (int)hipGetLastError();
}

template <class K, class... A>
static int launch_chain(K kernel, long P, int num_cu, hipStream_t stream, A... args)
{
    if (P <= 0) return 0;
    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;       // (one per instantiation = per kernel)
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BWC_LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    kernel<<<grid, X_THREADS, BWC_LDS_BYTES, stream>>>(args...);
    return (int)hipGetLastError();
}

extern "C" int sahs_bwd_chain_rad_launch(const void *bstream, long P, const float *d_raw, const uint32_t *bits, float *dact, float *dgridf,
                                         float *din_a, float *din_b, int num_cu, hipStream_t stream)
{
    return launch_chain(field_backward_chain_rad_kernel, P, num_cu, stream, reinterpret_cast<const unsigned short *>(bstream), P, d_raw, bits, dact,
                        dgridf, din_a, din_b);
}

extern "C" int sahs_bwd_chain_def_launch(const void *bstream, long P, const float *xwg, const float *actbuf, const uint32_t *bits, float *dact,
                                         float *g3, float *dw4, int num_cu, hipStream_t stream)
{
    return launch_chain(field_backward_chain_def_kernel, P, num_cu, stream, reinterpret_cast<const unsigned short *>(bstream), P, xwg, actbuf, bits,
                        dact, g3, dw4);
}
