// field_f32.hip -- the per-sample field (deformation MLPs + feature grid + radiance MLP), exact fp32.
//
// Replaces, for one level, the reference's run_network -> AudioFaceModel.forward chain
// (train_utils.py:9-50, models.py:514-528, modules.py:371-390 / 444-462 / 254-295, grid lookup
// models.py:346-365) including the point construction pts = ro + rd*z (train_utils.py:115,168).
// Input is the ray table and the per-ray depths; output is raw (N,S,16) = [rgb3, seg12, sigma].
// Nothing else touches HBM: the (P,18) point rows, the three positional encodings, the
// replicated conditioning vectors and every hidden activation of the reference (~50-100 KB per
// sample) never exist.
//
// Design (gfx950):
//  * one wave owns 16 sample points for the whole network.  v_mfma_f32_16x16x4_f32 computes
//    D[16 out-features x 16 points] += A[16 x 4] * B[4 x 16]; the D tile of layer l (lane (q,j):
//    features 4q..4q+3 of point j in its 4 registers) is exactly the B operand layout of layer
//    l+1 (lane (q,j) supplies k = feature 16b+4q+r at step r), so activations stay in VGPRs and
//    never visit LDS.  f32 MFMA is a k-ordered fmaf chain: results are exact fp32.
//  * weights are pre-packed in A-fragment order (pack.hip) and streamed L2 -> LDS in <=32 KB
//    chunks with global_load_lds (no VGPR staging), double buffered, one barrier per chunk; all 8
//    waves of the workgroup (2 per SIMD: MFMA of one overlaps VALU/LDS of the other) consume the
//    same chunk, so each weight byte is fetched once per 128 points.
//  * persistent workgroups (one per CU) walk 128-point tiles; the weight stream simply wraps.
// Roofline: MFMA-bound (1.86 MFLOP per sample against ~100 B of HBM traffic).
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"
#include "f32_pipe.hpp"

namespace SAHS_NS {


// ---- positional encoding in B layout ----------------------------------------------------------
// feature f of positional_encoding(v[0:d], L, include_input=True): [v | sin(2^0 v) | cos(2^0 v) | ...]
// (nerf_helpers.py:322-349); f >= d+2dL is zero padding.
//
// sin/cos(2^k x) to ~1e-7 absolute, without a general-purpose range reduction: the octave scaling is a power
// of two, so the angle is reduced in REVOLUTIONS.  u = x/(2 pi) is formed once per coordinate as an unevaluated
// sum p + lo (error-free product with a two-term 1/(2 pi), ~48 bits); 2^k p and 2^k lo are exact, 2^k p - rint(2^k p)
// is exact, so the fraction f of a revolution carries only lo's error (<= 2^-48 * 2^k |u|).  Then the quadrant is
// peeled off and sin/cos(2 pi g), |g| <= 1/8, are the classic minimax polynomials on [-pi/4, pi/4] (1 ulp).
// The reference evaluates sin(fl(2^k x)) with SLEEF/libm (1 ulp): both are within rounding of the true value.
struct RevArg { float p, lo; };
__device__ __forceinline__ RevArg rev_arg(float x)
{
    const float C_HI = 0.15915494f;                 // fl(1/(2 pi))
    const float C_LO = 6.4206382e-09f;              // 1/(2 pi) - C_HI
    RevArg r;
    r.p = x * C_HI;
    r.lo = fmaf(x, C_LO, fmaf(x, C_HI, -r.p));
    return r;
}
// sin(2^k x + fn*pi/2) given rev_arg(x) and scale = 2^k
__device__ __forceinline__ float sin_octave(RevArg u, float scale, int fn)
{
    const float P = u.p * scale, Lo = u.lo * scale;                 // exact
    const float f = (P - rintf(P)) + Lo;                            // fraction of a revolution, [-1/2, 1/2]
    const float qf = rintf(4.0f * f);
    const float th = fmaf(qf, -0.25f, f) * 6.2831855f;              // |th| <= pi/4
    const float z = th * th;
    const float sp = fmaf(th * z, fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), th);
    const float cp = fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), fmaf(z, -0.5f, 1.0f));
    const int m = ((int)qf + fn) & 3;                               // sin(th + m pi/2)
    const float v = (m & 1) ? cp : sp;
    return (m & 2) ? -v : v;
}

template <int D, int L, int NB, int INC = 1>   // INC: include_input (the raw coordinates come first)
__device__ __forceinline__ void pe_blocks(const float *v, int q, f32x4 *out)
{
    constexpr int D0 = INC ? D : 0;
    constexpr int W = D0 + 2 * D * L;
    // plain scalars, selected with ternaries: an array of structs indexed by the lane-dependent axis ends up in scratch
    const float v0 = v[0], v1 = v[D > 1 ? 1 : 0], v2 = v[D > 2 ? 2 : 0];
    const RevArg u0 = rev_arg(v0), u1 = rev_arg(v1), u2 = rev_arg(v2);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * b + 4 * q + r;
            float val = 0.0f;
            if (f < D0) {
                val = (f == 0) ? v0 : ((f == 1) ? v1 : v2);
            } else if (f < W) {
                const int g = f - D0;
                const int k = g / (2 * D), rem = g % (2 * D);
                const int fn = rem / D, ax = rem % D;
                RevArg a;
                a.p = (ax == 0) ? u0.p : ((ax == 1) ? u1.p : u2.p);
                a.lo = (ax == 0) ? u0.lo : ((ax == 1) ? u1.lo : u2.lo);
                val = sin_octave(a, (float)(1 << k), fn);
            }
            out[b][r] = val;
        }
}

// ---- trilinear feature-grid lookup (models.py:346-365; ATen grid_sampler_3d, align_corners=True,
// zeros padding; x indexes W, y H, z D).  grid is channel-last [D][H][W][32]; this lane fetches
// channels 16b+4q..+3 (b = 0,1) = its B-layout share.
__device__ __forceinline__ void grid_blocks(const float *__restrict__ grid, float x, float y, float z, int q, f32x4 *out)
{
    const float R1 = (float)(G_RES - 1);
    const float ix = ((x + 1.0f) / 2.0f) * R1, iy = ((y + 1.0f) / 2.0f) * R1, iz = ((z + 1.0f) / 2.0f) * R1;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float x1 = fx + 1.0f, y1 = fy + 1.0f, z1 = fz + 1.0f;
    const float wx[2] = {x1 - ix, ix - fx}, wy[2] = {y1 - iy, iy - fy}, wz[2] = {z1 - iz, iz - fz};
    const bool ok = fx >= -1.0f && fx <= (float)G_RES && fy >= -1.0f && fy <= (float)G_RES && fz >= -1.0f && fz <= (float)G_RES;
    const int xi = ok ? (int)fx : -2, yi = ok ? (int)fy : -2, zi = ok ? (int)fz : -2;
    out[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    out[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int cx = xi + (n & 1), cy = yi + ((n >> 1) & 1), cz = zi + (n >> 2);
        const bool inb = cx >= 0 && cx < G_RES && cy >= 0 && cy < G_RES && cz >= 0 && cz < G_RES;
        const float wt = (wx[n & 1] * wy[(n >> 1) & 1]) * wz[n >> 2];
        const long vox = inb ? (((long)cz * G_RES + cy) * G_RES + cx) : 0;
        const f32x4 *g = reinterpret_cast<const f32x4 *>(grid + vox * D_GRID) + q;
        const f32x4 g0 = g[0], g1 = g[4];
        // branch-free (weight 0 for a corner outside the grid; x + 0 == x): guarded corners become eight dependent fetch round trips
        const float we = inb ? wt : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { out[0][r] = out[0][r] + g0[r] * we; out[1][r] = out[1][r] + g1[r] * we; }
    }
}

#define CHF(id) (kProg.layer[id].G * kProg.layer[id].KB * 256)   /* floats in one chunk of layer id */

// SAVE: also store every layer's activations (sahs::act layout, 19 KB/sample) for field_bwd.hip
// MODE: which part of the per-sample network a launch evaluates.  The deformation nets (warp field, hyper sheet) are shared by the
// coarse and the fine level, and the fine pass's depths are sort(cat(coarse depths, new depths)) (train_utils.py:166): the reference
// evaluates the deformation of the coarse samples a second time in its fine pass.  Here the coarse launch (FIELD_ALL) also writes every
// sample's deformed point x' and ambient coordinate w to `xw`, a FIELD_DEFORM launch computes them for the new depths only, and the
// fine launch (FIELD_RADIANCE) reads them through the merge permutation `src` and runs only the radiance net: the same arithmetic on
// the same operands, hence bit-identical results, for 6 % fewer matrix instructions per frame.
//   xw: (rays, xw_row, 8) floats [x'0 x'1 x'2 w0 w1 . . .]; a launch over (N, S) depths owns columns xw_col0 .. xw_col0 + S - 1
//   src: (N, S) int32, FIELD_RADIANCE only: column of xw holding sample s of the ray
enum { FIELD_ALL = 0, FIELD_DEFORM = 1, FIELD_RADIANCE = 2 };
template <bool SAVE, int MODE>
__global__ void __launch_bounds__(F32_THREADS, 2)
field_forward_f32_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                         const float *__restrict__ rays, int ray_stride, const float *__restrict__ zvals,
                         float *__restrict__ raw, float *__restrict__ dbg, float *__restrict__ actbuf,
                         float *__restrict__ xw, int xw_row, int xw_col0, const int *__restrict__ src, uint32_t *__restrict__ bits)
{
#if SAHS_MODEL == 2
    static_assert(MODE == FIELD_ALL, "this model has no deformation nets to split off");
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Ctx cx;
    cx.stream = packed + PACK_STREAM_OFF + (long)level * STREAM_FLOATS;
    cx.lds = lds;
    cx.buf = 1;                       // so that the first chunk lands in buffer 0
    cx.lane = threadIdx.x & 63;
    cx.q = cx.lane >> 4;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr const Layer *Ly = kProg.layer;
    constexpr int L_START = (MODE == FIELD_RADIANCE) ? L_T0 : L_FIRST;      // first layer of this launch's walk through the stream
    cx.wrap_to = (MODE == FIELD_RADIANCE) ? (uint32_t)Ly[L_T0].stream_off : 0u;
    cx.wrap_at = (MODE == FIELD_DEFORM) ? (uint32_t)Ly[L_T0].stream_off : (uint32_t)STREAM_FLOATS;
    cx.off = cx.wrap_to;
    const float *grid = packed + PACK_GRID_OFF;
    const int q = cx.q;

    {   // per-level biases (static + folded conditioning) -> LDS, first weight chunk -> buffer 0
        const float *bsrc = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += F32_THREADS) lds[LDS_BIAS_OFF + i] = bsrc[i];
        cx.begin_chunk(CHF(L_START));
#pragma unroll
        for (int pc = 0; pc < (CHF(L_START) + PIECE_FLOATS - 1) / PIECE_FLOATS; ++pc) cx.issue_piece(pc);
        cx.end_chunk();
    }

    const long ntiles = (P + F32_PTS_PER_WG - 1) / F32_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cx.refresh();
        const long p_raw = tile * F32_PTS_PER_WG + cx.wave * F32_PTS_PER_WAVE + (cx.lane & 15);
        const long p = p_raw < P ? p_raw : P - 1;
        // x' (3) and w (2) are parked in LDS between their uses; the ray direction is re-read where needed: values that stay
        // live across the 256-wide layers get spilled to scratch, and a scratch reload drains the LDS-DMA prefetch (vmcnt).
        float *stash = lds + LDS_STASH_OFF + (cx.wave * F32_PTS_PER_WAVE + (cx.lane & 15)) * STASH_FLOATS;
        const bool dump = MODE == FIELD_ALL && dbg != nullptr && q == 0 && p_raw < P;   // lanes holding feature 0 of tile 0
        float *dsl = dbg + p * DBG_STRIDE;
        // saved activations are one dense [P x width] array per layer (array at column c of the act:: table starts at c * P):
        // this lane's slot in its sample's row of array (c, w)
        const bool sv_on = SAVE && p_raw < P;
        // (the plane bases actbuf + c * P are loop-invariant: unless P is opaque per tile the compiler hoists all ~40 of them out of the
        // persistent loop and spills them -- 176 bytes of scratch whose reloads drain the LDS-DMA prefetch)
        long Psv = P;
        if constexpr (SAVE) asm volatile("" : "+s"(Psv));
#define SVP(c, w) (actbuf + (long)(c) * Psv + p * (long)(w) + 4 * q)
#define SV(c, w) SVP(c, w)      // (a layer's tile stores are unconditional: lanes past the end redo sample P - 1 and store the same values again -- f32_pipe.hpp: kStores)
        // sign-bit planes (sbits): this lane's NW = w / 128 (at least 1) words of plane b; a whole-network save holds [deformation | radiance] planes
        const bool sb_on = sv_on && bits != nullptr;
        uint32_t *const bits_r = bits + (MODE == FIELD_ALL ? (long)sbits::BD_WORDS * Psv : 0L);
        const uint32_t sb_lane = (uint32_t)(p * 4 + q);      // (a uniform plane base + a 32-bit lane offset: one live register, not a pointer pair)
#define SBD(b, w) (sb_on ? bits + (long)(b) * Psv + sb_lane * (uint32_t)((w) >= 128 ? (w) / 128 : 1) : nullptr)
#define SBR(b, w) (sb_on ? bits_r + (long)(b) * Psv + sb_lane * (uint32_t)((w) >= 128 ? (w) / 128 : 1) : nullptr)
        float x[3] = {0.0f, 0.0f, 0.0f};
        if constexpr (MODE != FIELD_RADIANCE) {
            const float *rp = rays + (p / S) * ray_stride;
            const float z = zvals[p];
#pragma unroll
            for (int i = 0; i < 3; ++i) x[i] = rp[i] + rp[3 + i] * z;          // train_utils.py:115
        } else if (q == 0) {     // x', w of this sample were computed by the coarse or the deformation launch: fetch through the merge permutation
            const float *row = xw + ((p / S) * (long)xw_row + src[p]) * 8;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(row);
            stash[0] = a[0]; stash[1] = a[1]; stash[2] = a[2]; stash[3] = a[3]; stash[4] = row[4];
            if (sv_on) { float *d = SVP(act::XW, 16); d[0] = a[0]; d[1] = a[1]; d[2] = a[2]; }   // the grid backward reads x'
        }
        if constexpr (MODE != FIELD_RADIANCE) {
#if SAHS_MODEL == 2
        // no deformation nets (use_warp False, use_ambient False): the template is queried at the raw point (models.py:316-327)
        if (q == 0) {
            stash[0] = x[0]; stash[1] = x[1]; stash[2] = x[2]; stash[3] = 0.0f; stash[4] = 0.0f;
            if (sv_on) { float *d = SVP(act::XW, 16); d[0] = x[0]; d[1] = x[1]; d[2] = x[2]; }   // the grid backward reads it
        }
#else
        f32x4 pe_x[KB_XYZ];
        {
            pe_blocks<3, L_XYZ, KB_XYZ>(x, q, pe_x);
            if (sv_on) {
#pragma unroll
                for (int b = 0; b < KB_XYZ; ++b) *reinterpret_cast<f32x4 *>(SVP(act::E, 16 * KB_XYZ) + 16 * b) = pe_x[b];
            }
        }
        // ---- warp field: dx = tanh(MLP) (modules.py:371-390) ----
        {
            f32x4 h[8], hn[8];
            dense_sv<SAVE, KB_XYZ, 0, 8, CHF(L_W1)>(cx, pe_x, nullptr, h, Ly[L_W0].bias_off, false, 0.0f, SV(act::WH, 128), SBD(sbits::BD_WH + 4 * (0), 128));
            // W1..W3; the chunk after each is W2, W3, W4B.  One rolled loop where those have the same size (AudioFaceModel: 32 KB)
            constexpr int W_ROLLED = (CHF(L_W4B) == CHF(L_W2)) ? 3 : 2;
#pragma unroll 1
            for (int l = 0; l < W_ROLLED; ++l) {
                dense_sv<SAVE, 8, 0, 8, CHF(L_W2)>(cx, h, nullptr, hn, Ly[L_W1].bias_off + 128 * l, false, 0.0f, SV(act::WH + 128 * (l + 1), 128), SBD(sbits::BD_WH + 4 * ((l + 1)), 128));
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] = hn[i];
            }
            if (W_ROLLED == 2) {
                dense_sv<SAVE, 8, 0, 8, CHF(L_W4B)>(cx, h, nullptr, hn, Ly[L_W3].bias_off, false, 0.0f, SV(act::WH + 128 * 3, 128), SBD(sbits::BD_WH + 4 * (3), 128));
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] = hn[i];
            }
            dense<KB_XYZ, 0, 8, CHF(L_W4A)>(cx, pe_x, nullptr, hn, Ly[L_W4B].bias_off, false, 1.0f);
            dense_sv<SAVE, 8, 0, 8, CHF(L_W5)>(cx, h, nullptr, hn, 0, true, 0.0f, SV(act::WH + 4 * 128, 128), SBD(sbits::BD_WH + 4 * (4), 128));
            dense_sv<SAVE, 8, 0, 8, CHF(L_WF)>(cx, hn, nullptr, h, Ly[L_W5].bias_off, false, 0.0f, SV(act::WH + 5 * 128, 128), SBD(sbits::BD_WH + 4 * (5), 128));
            f32x4 o[1];
            dense<8, 0, 1, CHF(L_H0)>(cx, h, nullptr, o, Ly[L_WF].bias_off, false, 1.0f);
            if (q == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float dxv = tanhf(o[0][i]);
                    stash[i] = x[i] + dxv;                                            // models.py:305 (rows 0..2 live in lane quarter 0)
                    if (sv_on) { SVP(act::DX, 16)[i] = dxv; SVP(act::XW, 16)[i] = x[i] + dxv; }
                }
            }
        }
        // ---- hyper sheet: ambient w (modules.py:444-462) ----
        {
            f32x4 h[4], hn[4];
            dense_sv<SAVE, KB_XYZ, 0, 4, CHF(L_H1)>(cx, pe_x, nullptr, h, Ly[L_H0].bias_off, false, 0.0f, SV(act::HH, 64), SBD(sbits::BD_HH + 4 * (0), 64));
            constexpr int H_ROLLED = (CHF(L_H4B) == CHF(L_H2)) ? 3 : 2;   // H1..H3 (next chunks: H2, H3, H4B)
#pragma unroll 1
            for (int l = 0; l < H_ROLLED; ++l) {
                dense_sv<SAVE, 4, 0, 4, CHF(L_H2)>(cx, h, nullptr, hn, Ly[L_H1].bias_off + 64 * l, false, 0.0f, SV(act::HH + 64 * (l + 1), 64), SBD(sbits::BD_HH + 4 * ((l + 1)), 64));
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = hn[i];
            }
            if (H_ROLLED == 2) {
                dense_sv<SAVE, 4, 0, 4, CHF(L_H4B)>(cx, h, nullptr, hn, Ly[L_H3].bias_off, false, 0.0f, SV(act::HH + 64 * 3, 64), SBD(sbits::BD_HH + 4 * (3), 64));
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = hn[i];
            }
            dense<KB_XYZ, 0, 4, CHF(L_H4A)>(cx, pe_x, nullptr, hn, Ly[L_H4B].bias_off, false, 1.0f);
            dense_sv<SAVE, 4, 0, 4, CHF(L_H5)>(cx, h, nullptr, hn, 0, true, 0.0f, SV(act::HH + 4 * 64, 64), SBD(sbits::BD_HH + 4 * (4), 64));
            dense_sv<SAVE, 4, 0, 4, CHF(L_HF)>(cx, hn, nullptr, h, Ly[L_H5].bias_off, false, 0.0f, SV(act::HH + 5 * 64, 64), SBD(sbits::BD_HH + 4 * (5), 64));
            f32x4 o[1];
            dense<4, 0, 1, (MODE == FIELD_DEFORM ? CHF(L_FIRST) : CHF(L_T0))>(cx, h, nullptr, o, Ly[L_HF].bias_off, false, 1.0f);
            if (q == 0) {      // rows 0..AMB_DIM-1 of the one output tile
                stash[3] = o[0][0]; stash[4] = (AMB_DIM > 1) ? o[0][1] : 0.0f;
                if (sv_on) { SVP(act::AW, 16)[0] = o[0][0]; SVP(act::AW, 16)[1] = stash[4]; }
            }
        }
#endif
        if (xw != nullptr && q == 0 && p_raw < P) {   // hand x', w to the fine pass (the lane that wrote the stash reads it back: no barrier needed)
            float *row = xw + ((p / S) * (long)xw_row + xw_col0 + (p % S)) * 8;
            *reinterpret_cast<f32x4 *>(row) = f32x4{stash[0], stash[1], stash[2], stash[3]};
            row[4] = stash[4];
        }
        }   // MODE != FIELD_RADIANCE
        if constexpr (MODE == FIELD_DEFORM) continue;      // the stream has wrapped to the warp field's first layer
        __builtin_amdgcn_wave_barrier();
        if (dump) { dsl[0] = stash[0] - x[0]; dsl[1] = stash[1] - x[1]; dsl[2] = stash[2] - x[2]; dsl[3] = stash[3]; dsl[4] = stash[4]; }

        // ---- radiance trunk (modules.py:254-275) ----
        f32x4 fin[1];      // FINAL tile: raw[4q..4q+3] of this lane's point
        f32x4 feat[16];
        {
            f32x4 h[16];
            {
                f32x4 in_tr[KB_XYZ + KB_AMB];
                const float xw[3] = {stash[0], stash[1], stash[2]}, amb[2] = {stash[3], stash[4]};
                pe_blocks<3, L_XYZ, KB_XYZ>(xw, q, in_tr);
#if SAHS_MODEL != 2
                pe_blocks<AMB_DIM, L_AMB, KB_AMB, AMB_INC>(amb, q, in_tr + KB_XYZ);
#else
                (void)amb;
#endif
                if (sv_on) {
#pragma unroll
                    for (int b = 0; b < KB_XYZ + KB_AMB; ++b)
                        *reinterpret_cast<f32x4 *>(b < KB_XYZ ? SVP(act::PEX, 16 * KB_XYZ) + 16 * b : SVP(act::PEW, 16 * KB_AMB) + 16 * (b - KB_XYZ)) = in_tr[b];
                }
                dense_sv<SAVE, KB_XYZ, KB_AMB, 16, CHF(L_T1)>(cx, in_tr, in_tr + KB_XYZ, h, Ly[L_T0].bias_off, false, 0.01f, SV(act::T, 256), SBR(sbits::BR_T + 8 * (0), 256));
            }
            if (dump) dsl[5] = h[0][0];
            dense_sv<SAVE, 16, 0, 16, CHF(L_T2)>(cx, h, nullptr, feat, Ly[L_T1].bias_off, false, 0.01f, SV(act::T + 256, 256), SBR(sbits::BR_T + 8 * (1), 256));
            if (dump) dsl[6] = feat[0][0];
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = feat[i];
            dense_sv<SAVE, 16, 0, 16, CHF(L_T3B)>(cx, h, nullptr, feat, Ly[L_T2].bias_off, false, 0.01f, SV(act::T + 512, 256), SBR(sbits::BR_T + 8 * (2), 256));
            if (dump) dsl[7] = feat[0][0];
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = feat[i];
            {   // skip layer: the re-injected encoding is rebuilt here instead of staying live (24 VGPRs) through T1, T2
                f32x4 in_tr[KB_XYZ + KB_AMB];
                const float xw[3] = {stash[0], stash[1], stash[2]}, amb[2] = {stash[3], stash[4]};
                pe_blocks<3, L_XYZ, KB_XYZ>(xw, q, in_tr);
#if SAHS_MODEL != 2
                pe_blocks<AMB_DIM, L_AMB, KB_AMB, AMB_INC>(amb, q, in_tr + KB_XYZ);
#else
                (void)amb;
#endif
                dense<KB_XYZ, KB_AMB, 16, CHF(L_T3A)>(cx, in_tr, in_tr + KB_XYZ, feat, Ly[L_T3B].bias_off, false, 1.0f);
            }
#if SAHS_MODEL == 0
            dense_sv<SAVE, 16, 0, 16, CHF(L_T4)>(cx, h, nullptr, feat, 0, true, 0.01f, SV(act::T + 768, 256), SBR(sbits::BR_T + 8 * (3), 256));
#else           // 4-layer trunk: the skip layer is the last one, fc_feat follows
            dense_sv<SAVE, 16, 0, 16, CHF(L_FEAT)>(cx, h, nullptr, feat, 0, true, 0.01f, SV(act::T + 768, 256), SBR(sbits::BR_T + 8 * (3), 256));
#endif
            if (dump) dsl[8] = feat[0][0];
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = feat[i];
#if SAHS_MODEL == 0
#pragma unroll 1
            for (int l = 4; l <= 7; ++l) {     // T4..T7 (next chunks: T5, T6, T7, FEAT, all 32 KB)
                dense_sv<SAVE, 16, 0, 16, CHF(L_T5)>(cx, h, nullptr, feat, Ly[L_T4].bias_off + 256 * (l - 4), false, 0.01f, SV(act::T + 256 * l, 256), SBR(sbits::BR_T + 8 * (l), 256));
                if (dump) dsl[5 + l] = feat[0][0];
#pragma unroll
                for (int i = 0; i < 16; ++i) h[i] = feat[i];
            }
#endif
            dense_sv<SAVE, 16, 0, 16, CHF(L_ALPHA)>(cx, h, nullptr, feat, Ly[L_FEAT].bias_off, false, 1.0f, SV(act::FEAT, 256));
            if (dump) dsl[13] = feat[0][0];
        }
        dense<16, 0, 1, CHF(L_D0B)>(cx, feat, nullptr, fin, Ly[L_ALPHA].bias_off, false, 1.0f);
        if (dbg != nullptr && p_raw < P) *reinterpret_cast<f32x4 *>(dsl + 24 + 4 * q) = fin[0];
        // ---- colour branch (modules.py:276-287) ----
        {
            f32x4 in_d[4];
            {
                const float *rp = rays + (p / S) * ray_stride;
                const float rd[3] = {rp[3], rp[4], rp[5]};
                pe_blocks<3, 4, 2>(rd, q, in_d);                                  // models.py:340 (raw, un-normalised direction)
                grid_blocks(grid, stash[0], stash[1], stash[2], q, in_d + 2);     // models.py:525
                if (sv_on) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4 *>(b < 2 ? SVP(act::DIR, 32) + 16 * b : SVP(act::GRID, 32) + 16 * (b - 2)) = in_d[b];
                }
            }
            if (dbg != nullptr && p_raw < P) {
                float *d = dbg + P * DBG_STRIDE + p * 32;
                *reinterpret_cast<f32x4 *>(d + 4 * q) = in_d[2];
                *reinterpret_cast<f32x4 *>(d + 16 + 4 * q) = in_d[3];
            }
            f32x4 c[8], cn[8];
            dense<2, 2, 8, CHF(L_D0A)>(cx, in_d, in_d + 2, c, Ly[L_D0B].bias_off, false, 1.0f);
            dense_sv<SAVE, 16, 0, 8, CHF(L_D1)>(cx, feat, nullptr, c, 0, true, 0.01f, SV(act::C, 128), SBR(sbits::BR_C + 4 * (0), 128));
            if (dump) dsl[14] = c[0][0];
#pragma unroll 1
            for (int l = 0; l < 2; ++l) {      // D1, D2
                dense_sv<SAVE, 8, 0, 8, CHF(L_D2)>(cx, c, nullptr, cn, Ly[L_D1].bias_off + 128 * l, false, 0.01f, SV(act::C + 128 * (l + 1), 128), SBR(sbits::BR_C + 4 * ((l + 1)), 128));
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = cn[i];
            }
            dense_sv<SAVE, 8, 0, 8, CHF(L_RGB)>(cx, c, nullptr, cn, Ly[L_D3].bias_off, false, 0.01f, SV(act::C + 384, 128), SBR(sbits::BR_C + 4 * (3), 128));
            if (dump) dsl[15] = cn[0][0];
            dense<8, 0, 1, CHF(L_S0)>(cx, cn, nullptr, fin, 0, true, 1.0f);
            if (dbg != nullptr && p_raw < P) *reinterpret_cast<f32x4 *>(dsl + 40 + 4 * q) = fin[0];
        }
        // ---- seg branch (modules.py:289-294) ----
        {
            f32x4 s[8], sn[8];
            dense_sv<SAVE, 16, 0, 8, CHF(L_S1)>(cx, feat, nullptr, s, Ly[L_S0].bias_off, false, 0.01f, SV(act::S, 128), SBR(sbits::BR_S + 4 * (0), 128));
            if (dump) dsl[16] = s[0][0];
#pragma unroll 1
            for (int l = 0; l < 2; ++l) {      // S1, S2
                dense_sv<SAVE, 8, 0, 8, CHF(L_S2)>(cx, s, nullptr, sn, Ly[L_S1].bias_off + 128 * l, false, 0.01f, SV(act::S + 128 * (l + 1), 128), SBR(sbits::BR_S + 4 * ((l + 1)), 128));
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = sn[i];
            }
            dense_sv<SAVE, 8, 0, 8, CHF(L_SEG)>(cx, s, nullptr, sn, Ly[L_S3].bias_off, false, 0.01f, SV(act::S + 384, 128), SBR(sbits::BR_S + 4 * (3), 128));
            if (dump) dsl[17] = sn[0][0];
            dense<8, 0, 1, CHF(L_START)>(cx, sn, nullptr, fin, 0, true, 1.0f);
        }
        if (p_raw < P) *reinterpret_cast<f32x4 *>(raw + p * D_RAW + 4 * q) = fin[0];   // cat((rgb, seg, alpha)) modules.py:295
    }
}

}  // namespace SAHS_NS

using namespace SAHS_NS;

// dbg (optional, may be null): [P x 24: see DBG_STRIDE][P x 32: grid features]
template <bool SAVE, int MODE>
static int launch_field(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride, const float *zvals,
                        float *raw, float *dbg, float *actbuf, float *xw, int xw_row, int xw_col0, const int *src, int num_cu, hipStream_t stream,
                        uint32_t *bits = nullptr)
{
    const long ntiles = (P + F32_PTS_PER_WG - 1) / F32_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    const size_t lds_bytes = (size_t)LDS_TOTAL_FLOATS * sizeof(float);
    static sahs_once::Flags attr_set;       // the large-LDS attribute is per device (and per instantiation)
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_forward_f32_kernel<SAVE, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes);
    });
    if (ae != hipSuccess) return (int)ae;
    field_forward_f32_kernel<SAVE, MODE><<<grid, F32_THREADS, lds_bytes, stream>>>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg, actbuf,
                                                                                   xw, xw_row, xw_col0, src, bits);
    return (int)hipGetLastError();
}

extern "C" int SAHS_SYM(sahs_field_forward_f32_launch)(const float *packed, const float *frame, int level, long P, int S, const float *rays,
                                             int ray_stride, const float *zvals, float *raw, float *dbg, float *actbuf, int num_cu,
                                             hipStream_t stream)
{
    if (P <= 0) return 0;
    if (actbuf != nullptr)
        return launch_field<true, FIELD_ALL>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg, actbuf, nullptr, 0, 0, nullptr, num_cu, stream);
    return launch_field<false, FIELD_ALL>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg, nullptr, nullptr, 0, 0, nullptr, num_cu, stream);
}

// The split evaluation (see the kernel's MODE): mode 0 = whole network, additionally writing x', w of every sample to xw; mode 1 =
// deformation nets only (raw unused); mode 2 = radiance net only on x', w fetched through src (zvals unused).
// actbuf != nullptr: also save the activations of the layers the launch runs (training).  A saved array of act:: column c starts at
// actbuf + c * P as for whole-network launches, so a radiance-only launch touches columns >= act::XW only and a deformation-only launch
// columns < act::XW + 16: the caller may pass a base such that only that range is backed by memory (sahs_layout_act_part_words).
// bits (with actbuf): the sign-bit planes of the layers the launch runs (sahs_layout.hpp: sbits; mode 0: [deformation | radiance] planes),
// (P * sahs_layout_bits_part_words(mode)) 32-bit words, or null
extern "C" int SAHS_SYM(sahs_field_forward_f32_split_bits_launch)(const float *packed, const float *frame, int level, int mode, long P, int S,
                                                   const float *rays, int ray_stride, const float *zvals, float *raw, float *xw, int xw_row,
                                                   int xw_col0, const int *src, float *actbuf, uint32_t *bits, int num_cu, hipStream_t stream);
extern "C" int SAHS_SYM(sahs_field_forward_f32_split_launch)(const float *packed, const float *frame, int level, int mode, long P, int S,
                                                   const float *rays, int ray_stride, const float *zvals, float *raw, float *xw, int xw_row,
                                                   int xw_col0, const int *src, float *actbuf, int num_cu, hipStream_t stream)
{
    return SAHS_SYM(sahs_field_forward_f32_split_bits_launch)(packed, frame, level, mode, P, S, rays, ray_stride, zvals, raw, xw, xw_row, xw_col0, src, actbuf,
                                                              nullptr, num_cu, stream);
}
extern "C" int SAHS_SYM(sahs_field_forward_f32_split_bits_launch)(const float *packed, const float *frame, int level, int mode, long P, int S,
                                                   const float *rays, int ray_stride, const float *zvals, float *raw, float *xw, int xw_row,
                                                   int xw_col0, const int *src, float *actbuf, uint32_t *bits, int num_cu, hipStream_t stream)
{
    if (P <= 0) return 0;
#if SAHS_MODEL == 2
    return -3;      // no deformation nets in this model
#else
    if (actbuf != nullptr) {
        if (mode == FIELD_ALL)
            return launch_field<true, FIELD_ALL>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, nullptr, actbuf, xw, xw_row, xw_col0, nullptr, num_cu, stream, bits);
        if (mode == FIELD_DEFORM)
            return launch_field<true, FIELD_DEFORM>(packed, frame, level, P, S, rays, ray_stride, zvals, nullptr, nullptr, actbuf, xw, xw_row, xw_col0, nullptr, num_cu, stream, bits);
        if (mode == FIELD_RADIANCE)
            return launch_field<true, FIELD_RADIANCE>(packed, frame, level, P, S, rays, ray_stride, nullptr, raw, nullptr, actbuf, xw, xw_row, 0, src, num_cu, stream, bits);
        return -2;
    }
    if (mode == FIELD_ALL)
        return launch_field<false, FIELD_ALL>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, nullptr, nullptr, xw, xw_row, xw_col0, nullptr, num_cu, stream);
    if (mode == FIELD_DEFORM)
        return launch_field<false, FIELD_DEFORM>(packed, frame, level, P, S, rays, ray_stride, zvals, nullptr, nullptr, nullptr, xw, xw_row, xw_col0, nullptr, num_cu, stream);
    if (mode == FIELD_RADIANCE)
        return launch_field<false, FIELD_RADIANCE>(packed, frame, level, P, S, rays, ray_stride, nullptr, raw, nullptr, nullptr, xw, xw_row, 0, src, num_cu, stream);
    return -2;
#endif
}

// words per sample of the saved activations a launch of `part` writes / a backward of `part` reads, and the first act:: column of that
// range: part 0 whole network [0, STRIDE); 1 deformation nets [0, XW + 16); 2 radiance nets [XW, STRIDE)
extern "C" int SAHS_SYM(sahs_layout_act_part_words)(int part) { return part == 1 ? act::XW + 16 : (part == 2 ? act::STRIDE - act::XW : act::STRIDE); }
extern "C" int SAHS_SYM(sahs_layout_act_part_col0)(int part) { return part == 2 ? act::XW : 0; }
// 32-bit words per sample of the sign-bit planes of `part` (0: both, deformation planes first)
extern "C" int SAHS_SYM(sahs_layout_bits_part_words)(int part) { return part == 1 ? sbits::BD_WORDS : (part == 2 ? sbits::BR_WORDS : sbits::BD_WORDS + sbits::BR_WORDS); }
