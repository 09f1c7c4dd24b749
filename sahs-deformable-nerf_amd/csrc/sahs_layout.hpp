// sahs_layout.hpp -- static description of the per-point network as the field kernels execute it.
//
// The reference evaluates, per sample point (models.py:514-528 -> modules.py:371-390, 444-462,
// 254-295): WarpFieldMLP, HyperSheetMLP, a trilinear feature-grid lookup and NeRFMLP.  The field
// kernels run these as ONE fixed sequence of dense layers ("layer program") whose activations
// never leave registers: a layer's MFMA output tile (16 features x 16 points) is, register for
// register, the B operand of the next layer's MFMA (see field_f32.hip).  This header is the
// single source of truth for that program: which state_dict tensor feeds each layer, how its
// columns map onto 16-feature k-blocks, where its bias lives, and how it is cut into the LDS
// chunks the kernel streams.  pack.hip (weight packer, conditioning fold) and field_*.hip are
// all generated from this table, so they cannot disagree.
//
// Per-frame constants (driving[76] from AudioNet, pose36) are NOT network inputs here: the
// reference concatenates them to every point (models.py:518,521); we fold W[:, const]*c into the
// bias once per frame (fold_conditioning in pack.hip), which removes 6.6 % of the MACs and all
// of the replication traffic.  Algorithmic FLOPs are still counted as the reference writes them.
#pragma once
#include <cstdint>
#include "sahs_model.hpp"

namespace SAHS_NS {

// ---- architecture ----
#if SAHS_MODEL == 0
// AudioFaceModel, config/audio/person_2_auto.yml: 10-octave position encoding, 2-D ambient coordinate
// encoded with its input (4 octaves), 8-layer trunk fed [PE63(x') | PE18(w) | pose36]; driving = AudioNet(audio window).
constexpr int L_XYZ = 10, AMB_DIM = 2, L_AMB = 4, AMB_INC = 1, TR_LAYERS = 8;
constexpr bool TRUNK_SEES_POSE = true, HAS_AUDIONET = true, USE_DEFORM = true;
#elif SAHS_MODEL == 1
// NeRFaceModel, config/expression/person_2.yml / person_3.yml (models.py:189-299): 15 octaves, 1-D ambient coordinate
// encoded WITHOUT its input (15 octaves), 4-layer trunk fed [PE93(x') | PE30(w) | expression76] (include_driving, no
// pose); the deformation nets still take [PE93 | expression76 | pose36] (modules.py:344: dim_pose = include_pose + 36).
constexpr int L_XYZ = 15, AMB_DIM = 1, L_AMB = 15, AMB_INC = 0, TR_LAYERS = 4;
constexpr bool TRUNK_SEES_POSE = false, HAS_AUDIONET = false, USE_DEFORM = true;
#else
// NeRFaceModel, config/expression/person_1.yml ("nohyper_nodeform": warp.use_warp False, hyper.use_ambient False): no
// deformation nets in the state_dict, the grid is sampled at the raw point, 4-layer trunk fed [PE63(x) | expression76].
constexpr int L_XYZ = 10, AMB_DIM = 0, L_AMB = 0, AMB_INC = 0, TR_LAYERS = 4;
constexpr bool TRUNK_SEES_POSE = false, HAS_AUDIONET = false, USE_DEFORM = false;
#endif
constexpr int D_XYZ = 3 + 6 * L_XYZ;                             // 63 | 93
constexpr int D_AMB = AMB_INC * AMB_DIM + 2 * AMB_DIM * L_AMB;   // 18 | 30
constexpr int KB_XYZ = (D_XYZ + 15) / 16, KB_AMB = (D_AMB + 15) / 16;   // 16-feature k-blocks: 4 | 6, 2 | 2
constexpr int D_DIR = 27, D_POSE = 36, D_DRV = 76, D_GRID = 32, G_RES = 32;
constexpr int WARP_H = 128, HYP_H = 64, TR_H = 256, BR_H = 128, N_SEG = 12, D_RAW = 16;
constexpr int D_TR_CONST = TRUNK_SEES_POSE ? D_POSE : D_DRV;    // the per-frame constant the trunk sees
constexpr int TR_CONST_WHICH = TRUNK_SEES_POSE ? 1 : 0;         // Fold::which
constexpr int D_DEF_IN = D_XYZ + D_DRV + D_POSE;   // 175 | 205
constexpr int D_TR_IN = D_XYZ + D_AMB + D_TR_CONST;    // 117 | 199
constexpr int D_DIR_IN = TR_H + D_DIR + D_GRID;    // 315
constexpr long GRID_FLOATS = (long)D_GRID * G_RES * G_RES * G_RES;

// ---- canonical flat-buffer offsets (state_dict order; must equal weights.py::canonical_spec) ----
struct FlatOffsets {
    long grid;
    long warp_w[6], warp_b[6], warp_fw, warp_fb;
    long hyp_w[6], hyp_b[6], hyp_fw, hyp_fb;
    struct Lvl {
        long xyz_w[8], xyz_b[8], feat_w, feat_b, alpha_w, alpha_b;
        long dir_w[4], dir_b[4], rgb_w, rgb_b, seg_w[4], seg_b[4], segout_w, segout_b;
    } lvl[2];
    long conv_w[4], conv_b[4], fc_w[2], fc_b[2];
    long total;
};

constexpr FlatOffsets make_flat_offsets()
{
    FlatOffsets f{};
    long p = 0;
    f.grid = p; p += GRID_FLOATS;
    if (USE_DEFORM) {
        for (int i = 0; i < 6; ++i) {
            int in = (i == 0) ? D_DEF_IN : (i == 4 ? WARP_H + D_DEF_IN : WARP_H);
            f.warp_w[i] = p; p += (long)WARP_H * in; f.warp_b[i] = p; p += WARP_H;
        }
        f.warp_fw = p; p += 3 * WARP_H; f.warp_fb = p; p += 3;
        for (int i = 0; i < 6; ++i) {
            int in = (i == 0) ? D_DEF_IN : (i == 4 ? HYP_H + D_DEF_IN : HYP_H);
            f.hyp_w[i] = p; p += (long)HYP_H * in; f.hyp_b[i] = p; p += HYP_H;
        }
        f.hyp_fw = p; p += AMB_DIM * HYP_H; f.hyp_fb = p; p += AMB_DIM;
    }
    for (int l = 0; l < 2; ++l) {
        for (int i = 0; i < TR_LAYERS; ++i) {
            int in = (i == 0) ? D_TR_IN : (i == 3 ? TR_H + D_TR_IN : TR_H);
            f.lvl[l].xyz_w[i] = p; p += (long)TR_H * in; f.lvl[l].xyz_b[i] = p; p += TR_H;
        }
        f.lvl[l].feat_w = p; p += TR_H * TR_H; f.lvl[l].feat_b = p; p += TR_H;
        f.lvl[l].alpha_w = p; p += TR_H; f.lvl[l].alpha_b = p; p += 1;
        for (int i = 0; i < 4; ++i) {
            int in = (i == 0) ? D_DIR_IN : BR_H;
            f.lvl[l].dir_w[i] = p; p += (long)BR_H * in; f.lvl[l].dir_b[i] = p; p += BR_H;
        }
        f.lvl[l].rgb_w = p; p += 3 * BR_H; f.lvl[l].rgb_b = p; p += 3;
        for (int i = 0; i < 4; ++i) {
            int in = (i == 0) ? TR_H : BR_H;
            f.lvl[l].seg_w[i] = p; p += (long)BR_H * in; f.lvl[l].seg_b[i] = p; p += BR_H;
        }
        f.lvl[l].segout_w = p; p += N_SEG * BR_H; f.lvl[l].segout_b = p; p += N_SEG;
    }
    if (HAS_AUDIONET) {
        constexpr int cin[4] = {29, 32, 32, 64}, cout[4] = {32, 32, 64, 64};
        for (int i = 0; i < 4; ++i) { f.conv_w[i] = p; p += (long)cout[i] * cin[i] * 3; f.conv_b[i] = p; p += cout[i]; }
        f.fc_w[0] = p; p += 64 * 64; f.fc_b[0] = p; p += 64;
        f.fc_w[1] = p; p += D_DRV * 64; f.fc_b[1] = p; p += D_DRV;
    }
    f.total = p;
    return f;
}
constexpr FlatOffsets kFlat = make_flat_offsets();
static_assert(kFlat.total == (SAHS_MODEL == 0 ? 2775633 : (SAHS_MODEL == 1 ? 2311140 : 2066976)),
              "flat parameter count must match the reference state_dict");

// ---- layer program -------------------------------------------------------------------------
// A layer consumes up to 2 input segments (each a whole number of 16-feature k-blocks, zero
// padded) and produces NT 16-row output tiles.  Source columns that are per-frame constants are
// listed separately (folded into the bias by fold_conditioning).
//
// Skip layers are split in two so that every wide layer has the same shape: the "B" part
// multiplies the re-injected input (PE blocks) and carries the bias (+ fold), leaving
// pre-activations in registers; the "A" part multiplies the hidden state and accumulates onto
// them, then applies the activation.  Same for the colour branch's first layer (D0B: dirPE+grid,
// D0A: feat).  ALPHA/RGB/SEG all accumulate into ONE 16-row tile = raw[0:16] = [rgb3|seg12|sigma].
struct Seg { int blocks; int src_col; int valid; };          // k-blocks, first source column, #valid columns
struct Fold { int src_col; int count; int which; };          // which: 0 = driving[76], 1 = pose36
enum LayerId {   // enum order == execution order == stream order
#if SAHS_MODEL != 2
    L_W0, L_W1, L_W2, L_W3, L_W4B, L_W4A, L_W5, L_WF,
    L_H0, L_H1, L_H2, L_H3, L_H4B, L_H4A, L_H5, L_HF,
#endif
    L_T0, L_T1, L_T2, L_T3B, L_T3A,
#if SAHS_MODEL == 0
    L_T4, L_T5, L_T6, L_T7,
#endif
    L_FEAT, L_ALPHA,
    L_D0B, L_D0A, L_D1, L_D2, L_D3, L_RGB,
    L_S0, L_S1, L_S2, L_S3, L_SEG,
    NUM_LAYERS,
    L_FIRST = 0      // the stream wraps from L_SEG to this layer
};
struct Layer {
    long w_off[2];     // offset of the weight tensor in the flat buffer, per level
    long b_off[2];     // offset of the bias tensor (unused when !has_bias)
    int src_ld;        // row length (in_features) of the source tensor
    int src_rows;      // out_features of the source tensor
    int row_shift;     // source row r lands on output row r + row_shift (FINAL tile packing)
    int NT;            // 16-row output tiles
    int KB;            // total k-blocks (sum of seg blocks)
    int G;             // tiles per LDS chunk
    int nseg; Seg seg[2];
    int nfold; Fold fold[2];
    int has_bias;      // 0: accumulates onto the previous layer's pre-activations
    int bias_off;      // offset (floats) into the per-level bias array
    int bias_shared;   // 1: writes its rows of the shared FINAL bias slot
    long stream_off;   // offset (floats) of the layer's first chunk in the per-level weight stream
    int first_chunk;   // index of its first chunk in the chunk table
};

constexpr int CHUNK_FLOATS_MAX = 8192;   // 32 KB LDS staging buffer
constexpr int pick_G(int KB, int NT)
{
    int g = CHUNK_FLOATS_MAX / (KB * 256);
    if (g > NT) g = NT;
    while (NT % g) --g;   // largest divisor of NT not above g
    return g;
}

struct Program {
    Layer layer[NUM_LAYERS];
    int bias_floats;       // per level
    long stream_floats;    // per level
    int num_chunks;        // per level
};

struct Src { long w0, w1, b0, b1; int ld, rows; };

constexpr Layer mk(Src s, int has_bias, int shift, int NT, Seg s0, Seg s1 = {0, 0, 0}, Fold f0 = {0, 0, 0}, Fold f1 = {0, 0, 0})
{
    Layer L{};
    L.w_off[0] = s.w0; L.w_off[1] = s.w1; L.b_off[0] = s.b0; L.b_off[1] = s.b1;
    L.src_ld = s.ld; L.src_rows = s.rows; L.row_shift = shift; L.NT = NT; L.has_bias = has_bias;
    L.seg[0] = s0; L.seg[1] = s1;
    L.nseg = 1 + (s1.blocks > 0);
    L.KB = s0.blocks + s1.blocks;
    L.fold[0] = f0; L.fold[1] = f1;
    L.nfold = (f0.count > 0) + (f1.count > 0);
    L.G = pick_G(L.KB, NT);
    return L;
}

constexpr Program make_program()
{
    Program P{};
    const FlatOffsets &f = kFlat;
    Layer *L = P.layer;
    // deformation nets (shared by both levels).  Source columns: [PE63 | driving76 | pose36],
    // skip layer 4: [h | PE63 | driving76 | pose36]  (modules.py:372-387, 445-459)
#if SAHS_MODEL != 2
    auto warp = [&](int i, int ld) { return Src{f.warp_w[i], f.warp_w[i], f.warp_b[i], f.warp_b[i], ld, WARP_H}; };
    auto hyp = [&](int i, int ld) { return Src{f.hyp_w[i], f.hyp_w[i], f.hyp_b[i], f.hyp_b[i], ld, HYP_H}; };
    L[L_W0] = mk(warp(0, D_DEF_IN), 1, 0, 8, {KB_XYZ, 0, D_XYZ}, {0, 0, 0}, {D_XYZ, D_DRV, 0}, {D_XYZ + D_DRV, D_POSE, 1});
    L[L_W1] = mk(warp(1, WARP_H), 1, 0, 8, {8, 0, 128});
    L[L_W2] = mk(warp(2, WARP_H), 1, 0, 8, {8, 0, 128});
    L[L_W3] = mk(warp(3, WARP_H), 1, 0, 8, {8, 0, 128});
    L[L_W4B] = mk(warp(4, WARP_H + D_DEF_IN), 1, 0, 8, {KB_XYZ, WARP_H, D_XYZ}, {0, 0, 0}, {WARP_H + D_XYZ, D_DRV, 0},
                  {WARP_H + D_XYZ + D_DRV, D_POSE, 1});
    L[L_W4A] = mk(warp(4, WARP_H + D_DEF_IN), 0, 0, 8, {8, 0, 128});
    L[L_W5] = mk(warp(5, WARP_H), 1, 0, 8, {8, 0, 128});
    L[L_WF] = mk(Src{f.warp_fw, f.warp_fw, f.warp_fb, f.warp_fb, WARP_H, 3}, 1, 0, 1, {8, 0, 128});
    L[L_H0] = mk(hyp(0, D_DEF_IN), 1, 0, 4, {KB_XYZ, 0, D_XYZ}, {0, 0, 0}, {D_XYZ, D_DRV, 0}, {D_XYZ + D_DRV, D_POSE, 1});
    L[L_H1] = mk(hyp(1, HYP_H), 1, 0, 4, {4, 0, 64});
    L[L_H2] = mk(hyp(2, HYP_H), 1, 0, 4, {4, 0, 64});
    L[L_H3] = mk(hyp(3, HYP_H), 1, 0, 4, {4, 0, 64});
    L[L_H4B] = mk(hyp(4, HYP_H + D_DEF_IN), 1, 0, 4, {KB_XYZ, HYP_H, D_XYZ}, {0, 0, 0}, {HYP_H + D_XYZ, D_DRV, 0},
                  {HYP_H + D_XYZ + D_DRV, D_POSE, 1});
    L[L_H4A] = mk(hyp(4, HYP_H + D_DEF_IN), 0, 0, 4, {4, 0, 64});
    L[L_H5] = mk(hyp(5, HYP_H), 1, 0, 4, {4, 0, 64});
    L[L_HF] = mk(Src{f.hyp_fw, f.hyp_fw, f.hyp_fb, f.hyp_fb, HYP_H, AMB_DIM}, 1, 0, 1, {4, 0, 64});
#endif
    // radiance trunk.  Source columns: [PE(x') | PE(w) | pose36 or driving76 (model)]; skip layer 3: [h | same]
    // (modules.py:255-273; skip index is NeRFMLP's default 3, models.py never forwards the YAML's 4)
    const FlatOffsets::Lvl &c = f.lvl[0], &n = f.lvl[1];
    auto tr = [&](int i, int ld) { return Src{c.xyz_w[i], n.xyz_w[i], c.xyz_b[i], n.xyz_b[i], ld, TR_H}; };
    L[L_T0] = mk(tr(0, D_TR_IN), 1, 0, 16, {KB_XYZ, 0, D_XYZ}, {KB_AMB, D_XYZ, D_AMB}, {D_XYZ + D_AMB, D_TR_CONST, TR_CONST_WHICH});
    L[L_T1] = mk(tr(1, TR_H), 1, 0, 16, {16, 0, 256});
    L[L_T2] = mk(tr(2, TR_H), 1, 0, 16, {16, 0, 256});
    L[L_T3B] = mk(tr(3, TR_H + D_TR_IN), 1, 0, 16, {KB_XYZ, TR_H, D_XYZ}, {KB_AMB, TR_H + D_XYZ, D_AMB},
                  {TR_H + D_XYZ + D_AMB, D_TR_CONST, TR_CONST_WHICH});
    L[L_T3A] = mk(tr(3, TR_H + D_TR_IN), 0, 0, 16, {16, 0, 256});
#if SAHS_MODEL == 0
    for (int i = 4; i < 8; ++i) L[L_T4 + (i - 4)] = mk(tr(i, TR_H), 1, 0, 16, {16, 0, 256});
#endif
    L[L_FEAT] = mk(Src{c.feat_w, n.feat_w, c.feat_b, n.feat_b, TR_H, TR_H}, 1, 0, 16, {16, 0, 256});
    L[L_ALPHA] = mk(Src{c.alpha_w, n.alpha_w, c.alpha_b, n.alpha_b, TR_H, 1}, 1, 15, 1, {16, 0, 256});
    // colour branch: [feat256 | dirPE27 | grid32] (modules.py:276-287)
    auto dir = [&](int i, int ld) { return Src{c.dir_w[i], n.dir_w[i], c.dir_b[i], n.dir_b[i], ld, BR_H}; };
    L[L_D0B] = mk(dir(0, D_DIR_IN), 1, 0, 8, {2, 256, 27}, {2, 283, 32});
    L[L_D0A] = mk(dir(0, D_DIR_IN), 0, 0, 8, {16, 0, 256});
    for (int i = 1; i < 4; ++i) L[L_D0A + i] = mk(dir(i, BR_H), 1, 0, 8, {8, 0, 128});
    L[L_RGB] = mk(Src{c.rgb_w, n.rgb_w, c.rgb_b, n.rgb_b, BR_H, 3}, 1, 0, 1, {8, 0, 128});
    // seg branch (modules.py:289-294)
    auto sg = [&](int i, int ld) { return Src{c.seg_w[i], n.seg_w[i], c.seg_b[i], n.seg_b[i], ld, BR_H}; };
    L[L_S0] = mk(sg(0, TR_H), 1, 0, 8, {16, 0, 256});
    for (int i = 1; i < 4; ++i) L[L_S0 + i] = mk(sg(i, BR_H), 1, 0, 8, {8, 0, 128});
    L[L_SEG] = mk(Src{c.segout_w, n.segout_w, c.segout_b, n.segout_b, BR_H, N_SEG}, 1, 3, 1, {8, 0, 128});

    int boff = 0; long soff = 0; int chunk = 0;
    int final_bias = -1;
    for (int i = 0; i < NUM_LAYERS; ++i) {
        const bool fin = (i == L_ALPHA || i == L_RGB || i == L_SEG);
        if (fin) {
            if (final_bias < 0) { final_bias = boff; boff += 16; }
            L[i].bias_off = final_bias; L[i].bias_shared = 1;
        } else if (L[i].has_bias) {
            L[i].bias_off = boff; boff += L[i].NT * 16;
        } else {
            L[i].bias_off = -1;
        }
        L[i].stream_off = soff; L[i].first_chunk = chunk;
        soff += (long)L[i].NT * L[i].KB * 256;
        chunk += L[i].NT / L[i].G;
    }
    P.bias_floats = boff; P.stream_floats = soff; P.num_chunks = chunk;
    return P;
}
constexpr Program kProg = make_program();

constexpr int BIAS_FLOATS = kProg.bias_floats;             // per level
constexpr long STREAM_FLOATS = kProg.stream_floats;        // per level
constexpr int NUM_CHUNKS = kProg.num_chunks;

// ---- device buffers -------------------------------------------------------------------------
// packed (f32): [grid channel-last: G_RES^3 x 32 floats][level0 stream][level1 stream][chunk table: (NUM_CHUNKS+1) uint32 float-offsets]
constexpr long PACK_GRID_OFF = 0;
constexpr long PACK_STREAM_OFF = GRID_FLOATS;
constexpr long PACK_TABLE_OFF = PACK_STREAM_OFF + 2 * STREAM_FLOATS;
constexpr long PACK_FLOATS = PACK_TABLE_OFF + ((NUM_CHUNKS + 1 + 3) / 4) * 4;
// frame: [driving 76 (pad 80)][pose36 (pad 48)][bias level0 BIAS_FLOATS][bias level1]
constexpr int FRAME_DRV_OFF = 0, FRAME_POSE_OFF = 80, FRAME_BIAS_OFF = 128;
constexpr int FRAME_FLOATS = FRAME_BIAS_OFF + 2 * BIAS_FLOATS;

}  // namespace SAHS_NS

// =============================================================================================
// bf16 program (field_bf16.hip).  Same network, same bias array (the per-frame fold is shared), but
// the MFMA is v_mfma_f32_32x32x16_bf16: tiles are 32 rows, k-blocks are 32 features (= one D tile
// = two MFMA k-steps), and skip layers are NOT split (the accumulator of a tile simply runs over
// all input segments), so a layer has up to three input segments.
// Stream (bf16 halfwords): [layer][tile32][block32][step 0/1][lane 64][8]; lane = 32*h + i holds
// W[32t+i][32b + 16s + 8(j>>2) + 4h + (j&3)], j = 0..7: the k order in which a 32x32 accumulator
// tile turns into the next MFMA's B operand without moving between lanes.
// =============================================================================================
namespace SAHS_NS {
namespace hb {

enum LayerIdH {
#if SAHS_MODEL != 2
    H_W0, H_W1, H_W2, H_W3, H_W4, H_W5, H_WF,
    H_H0, H_H1, H_H2, H_H3, H_H4, H_H5, H_HF,
#endif
    H_T0, H_T1, H_T2, H_T3,
#if SAHS_MODEL == 0
    H_T4, H_T5, H_T6, H_T7,
#endif
    H_FEAT, H_ALPHA,
    H_D0, H_D1, H_D2, H_D3, H_RGB,
    H_S0, H_S1, H_S2, H_S3, H_SEG,
    NUM_LAYERS_H
};
struct LayerH {
    long w_off[2];
    int src_ld, src_rows, row_shift;
    int NT32, KB32, G32;
    int nseg; Seg seg[3];          // Seg.blocks in units of 32 features here
    int bias_off;                  // into the shared per-level bias array (floats)
    long stream_off;               // halfwords
    int first_chunk;
};
constexpr int CHUNK_HW_MAX = 32768;          // 64 KB LDS staging buffer
constexpr int pick_G32(int KB32, int NT32)
{
    int g = CHUNK_HW_MAX / (KB32 * 1024);
    if (g > NT32) g = NT32;
    while (NT32 % g) --g;
    return g;
}
struct ProgramH {
    LayerH layer[NUM_LAYERS_H];
    long stream_hw;
    int num_chunks;
};
constexpr Seg half_seg(Seg s) { return Seg{s.blocks / 2, s.src_col, s.valid}; }
constexpr LayerH from1(const Layer &a)     // plain layer
{
    LayerH h{};
    h.w_off[0] = a.w_off[0]; h.w_off[1] = a.w_off[1];
    h.src_ld = a.src_ld; h.src_rows = a.src_rows; h.row_shift = a.row_shift;
    h.NT32 = (a.NT + 1) / 2;
    h.nseg = a.nseg;
    h.seg[0] = half_seg(a.seg[0]); h.seg[1] = half_seg(a.seg[1]);
    h.KB32 = h.seg[0].blocks + h.seg[1].blocks;
    h.bias_off = a.bias_off;
    return h;
}
constexpr LayerH from2(const Layer &b, const Layer &a)   // split pair: hidden part (A) first, then the injected part (B)
{
    LayerH h = from1(a);
    h.seg[1] = half_seg(b.seg[0]); h.seg[2] = half_seg(b.seg[1]);
    h.nseg = 1 + b.nseg;
    h.KB32 = h.seg[0].blocks + h.seg[1].blocks + h.seg[2].blocks;
    h.bias_off = b.bias_off;
    return h;
}
constexpr ProgramH make_program_h()
{
    ProgramH P{};
    const Layer *L = kProg.layer;
    LayerH *H = P.layer;
#if SAHS_MODEL != 2
    H[H_W0] = from1(L[L_W0]); H[H_W1] = from1(L[L_W1]); H[H_W2] = from1(L[L_W2]); H[H_W3] = from1(L[L_W3]);
    H[H_W4] = from2(L[L_W4B], L[L_W4A]); H[H_W5] = from1(L[L_W5]); H[H_WF] = from1(L[L_WF]);
    H[H_H0] = from1(L[L_H0]); H[H_H1] = from1(L[L_H1]); H[H_H2] = from1(L[L_H2]); H[H_H3] = from1(L[L_H3]);
    H[H_H4] = from2(L[L_H4B], L[L_H4A]); H[H_H5] = from1(L[L_H5]); H[H_HF] = from1(L[L_HF]);
#endif
    H[H_T0] = from1(L[L_T0]); H[H_T1] = from1(L[L_T1]); H[H_T2] = from1(L[L_T2]);
    H[H_T3] = from2(L[L_T3B], L[L_T3A]);
#if SAHS_MODEL == 0
    H[H_T4] = from1(L[L_T4]); H[H_T5] = from1(L[L_T5]); H[H_T6] = from1(L[L_T6]); H[H_T7] = from1(L[L_T7]);
#endif
    H[H_FEAT] = from1(L[L_FEAT]); H[H_ALPHA] = from1(L[L_ALPHA]);
    H[H_D0] = from2(L[L_D0B], L[L_D0A]); H[H_D1] = from1(L[L_D1]); H[H_D2] = from1(L[L_D2]); H[H_D3] = from1(L[L_D3]);
    H[H_RGB] = from1(L[L_RGB]);
    H[H_S0] = from1(L[L_S0]); H[H_S1] = from1(L[L_S1]); H[H_S2] = from1(L[L_S2]); H[H_S3] = from1(L[L_S3]);
    H[H_SEG] = from1(L[L_SEG]);
    long soff = 0; int chunk = 0;
    for (int i = 0; i < NUM_LAYERS_H; ++i) {
        H[i].G32 = pick_G32(H[i].KB32, H[i].NT32);
        H[i].stream_off = soff; H[i].first_chunk = chunk;
        soff += (long)H[i].NT32 * H[i].KB32 * 1024;
        chunk += H[i].NT32 / H[i].G32;
    }
    P.stream_hw = soff; P.num_chunks = chunk;
    return P;
}
constexpr ProgramH kProgH = make_program_h();
constexpr long STREAM_HW = kProgH.stream_hw;      // per level, halfwords
constexpr int NUM_CHUNKS_H = kProgH.num_chunks;
// packed (bf16), in 4-byte words: [grid channel-last fp32][level0 stream][level1 stream][chunk table: halfword offsets / 8]
constexpr long PACKH_GRID_OFF = 0;
constexpr long PACKH_STREAM_OFF = GRID_FLOATS;                       // words
constexpr long PACKH_TABLE_OFF = PACKH_STREAM_OFF + STREAM_HW;       // 2 levels * STREAM_HW halfwords = STREAM_HW words
constexpr long PACKH_WORDS = PACKH_TABLE_OFF + ((NUM_CHUNKS_H + 1 + 3) / 4) * 4;
// packed (bf16x3, field_bf16x3.hip), in 4-byte words: [grid channel-last fp32][level0 hi/lo stream][level1 hi/lo stream]; a level's stream is the
// bf16 stream with every 1-KB fragment followed by the fragment of the remainders (layer i at 2 * stream_off[i], 2 * STREAM_HW halfwords)
constexpr long PACKX_GRID_OFF = 0;
constexpr long PACKX_STREAM_OFF = GRID_FLOATS;
constexpr long PACKX_WORDS = PACKX_STREAM_OFF + 2 * STREAM_HW;

}  // namespace hb
}  // namespace SAHS_NS

// =============================================================================================
// Saved activations for the backward pass (field_bwd.hip), fp32: ONE DENSE [P x width] ARRAY PER LAYER -- the array listed at
// column c below starts at float c * P of the buffer and is indexed [sample][feature] -- so that the backward GEMMs stream their
// operands sequentially (a row-per-sample interleaving made every operand a 19-KB-strided gather).  Every array is a whole
// number of 16-feature blocks in the field kernel's B layout: the forward kernel stores each finished tile with one float4 per lane.  Hidden activations are stored POST activation (relu/leaky-relu keep the sign, so the
// derivative mask is recovered from them); PE rows hold sin and cos of every octave, so the PE derivative needs
// no trigonometry.
// =============================================================================================
namespace SAHS_NS {
namespace act {
constexpr int E = 0;                    // PE(x), KB_XYZ blocks
constexpr int WH = E + 16 * KB_XYZ;     // warp hidden h0..h5, 6 x 128
constexpr int DX = WH + 6 * 128;        // dx = tanh(.), 3 (+13 pad)
constexpr int HH = DX + 16;             // hyper hidden g0..g5, 6 x 64
constexpr int AW = HH + 6 * 64;         // ambient w, AMB_DIM (+pad to 16)
constexpr int XW = AW + 16;             // warped point x', 3 (+13 pad)
constexpr int PEX = XW + 16;            // PE(x'), KB_XYZ blocks
constexpr int PEW = PEX + 16 * KB_XYZ;  // PE(w), KB_AMB blocks
constexpr int T = PEW + 16 * KB_AMB;    // trunk t0.., TR_LAYERS x 256
constexpr int FEAT = T + TR_LAYERS * 256;   // 256
constexpr int DIR = FEAT + 256;         // PE27(rd), 32
constexpr int GRID = DIR + 32;          // grid features, 32
constexpr int C = GRID + 32;            // colour hidden c0..c3, 4 x 128
constexpr int S = C + 4 * 128;          // seg hidden s0..s3, 4 x 128
constexpr int STRIDE = S + 4 * 128;     // sum of the array widths; AudioFaceModel: 4752 floats = 19 KB per sample
}  // namespace act
}  // namespace SAHS_NS

// =============================================================================================
// Sign-bit planes (round 4): field_forward_f32_kernel<SAVE> also writes, for every (leaky-)ReLU layer, which of its outputs are > 0 --
// the derivative mask of the backward chain (field_bwd_fused.hip), 1 bit instead of the 32 of the saved activation.  One plane per
// layer, [P][4 q][NW] 32-bit words: the lane that holds features 16 t + 4 q + r (r = 0..3) of every 16-row tile t of its sample sets
// bit 4 t + r of ITS OWN word(s) -- no cross-lane work in the forward; NW = NT / 8 words (at least 1).  The chain kernel's lane
// (sample, h) reads words q = h and q = 2 + h: its 16 values of 32-row tile T are features 32 T + 8 g + 4 h + i, i.e. 16-row tile
// 2 T + (g >> 1), q = 2 (g & 1) + h, r = i: bit 8 T + 4 (g >> 1) + i of word (g & 1 ? 2 + h : h).
// A part's planes start at word offset B*_ * P of that part's bits buffer (deformation nets / radiance nets; a whole-network save
// holds [deformation planes][radiance planes]).
// =============================================================================================
namespace SAHS_NS {
namespace sbits {
constexpr int words(int width) { return 4 * ((width / 16 + 7) / 8); }      // per sample: 256 -> 8, 128 -> 4, 64 -> 4
constexpr int BD_WH = 0;                                                    // warp hidden h0..h5
constexpr int BD_HH = BD_WH + (USE_DEFORM ? 6 * words(WARP_H) : 0);         // hyper hidden g0..g5
constexpr int BD_WORDS = BD_HH + (USE_DEFORM ? 6 * words(HYP_H) : 0);       // 48 words per sample
constexpr int BR_T = 0;                                                     // trunk t0..
constexpr int BR_C = BR_T + TR_LAYERS * words(TR_H);                        // colour hidden c0..c3
constexpr int BR_S = BR_C + 4 * words(BR_H);                                // seg hidden s0..s3
constexpr int BR_WORDS = BR_S + 4 * words(BR_H);                            // 96 words per sample (AudioFaceModel)
}  // namespace sbits
}  // namespace SAHS_NS
