// field_bf16x3.hip -- near-fp32 radiance nets on the bf16 matrix pipe (precision SAHS_BF16X3).
//
// Every operand is split into two bf16 numbers, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16-17 significant bits together),
// and a product is three MFMAs with fp32 accumulation,
//     W x  ~=  W_hi x_hi + W_hi x_lo + W_lo x_hi            (the dropped W_lo x_lo term is ~2^-18 of the product).
// Derived from field_bf16w.hip (one wave per SIMD: read that file for the structure -- chunked weight stream through LDS-DMA, hand-issued
// A reads with counted lgkmcnt, bias as the C operand of a tile's first MFMA, conversion "ticks" dealt into the MFMA gaps) with these
// differences:
//   * TWO KERNELS for the split evaluation's launches: field_radiance_bf16x3_kernel (FIELD_RADIANCE) and, since round 3,
//     field_deform_bf16x3_kernel (FIELD_DEFORM; round 2 left the deformation nets on the fp32 kernel because the deformed point feeds
//     sin/cos(2^9 x') -- measured, the split-operand error on x' = x + tanh(.) is 6e-7 and the frame stays within 4x the fp32 tolerance;
//     see the kernel's header).  AudioFaceModel only.
//   * ONE 32-sample half per wave: an activation block holds hi AND lo fragments -- the registers two halves of plain bf16 would take.
//   * The packed stream holds [hi fragment | lo fragment] per k-step (2 KB), so a 64 KB chunk holds half as many tiles.
//   * The conversion produces hi and lo (about 5 VALU per value instead of 2.5); with three MFMAs per value it hides under them.
#include <hip/hip_runtime.h>
#include <utility>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"
#include "bf16_pipe.hpp"
#include "bf16x3_pipe.hpp"      // Blk, dense_x, dense_x_out, the conversion ticks and their epilogue policies

#if SAHS_MODEL == 2
#error "field_bf16x3.hip: the model without deformation nets has no use for it (its whole net is the plain-bf16 radiance kernel)"
#endif
// SAHS_MODEL 0 (AudioFaceModel): both kernels = precision SAHS_BF16X3.  SAHS_MODEL 1 (NeRFaceModel with deformation nets): the deformation
// kernel only -- the deformation launches of that model's mixed-precision path (SAHS_BF16), whose radiance nets are plain bf16 anyway.

namespace SAHS_NS {
namespace hx3 {

// ---- positional encoding (v_sin_f32 on an fp32 revolution count reduced exactly: the value keeps ~fp32 accuracy, then hi/lo split) ------
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
// sin(2 pi t) for t = 2^k u + phase with u = x / (2 pi) held as an unevaluated fp32 sum (uh + ul): the product by a power of two and the
// reduction to [-0.5, 0.5) are exact, so the argument error is that of u (2^-24 relative), not 2^k times it
__device__ __forceinline__ float sin_rev(float uh, float ul, float scale, float phase)
{
    const float a = uh * scale;                     // exact (power of two)
    const float r = a - rintf(a);                   // exact
    const float t = (r + phase) + ul * scale;       // |t| < 1: fp32 rounding of a small number
    return __builtin_amdgcn_sinf(t - rintf(t));     // v_sin_f32 takes revolutions
}
template <int D, int L, int NB>
__device__ __forceinline__ void pe_blocks_x(const float *v, int h, Blk *out)
{
    float uh[3], ul[3];
#pragma unroll
    for (int i = 0; i < D; ++i) {       // u = v / (2 pi) as hi + lo (two-product with the split constant 1/(2 pi) = C_HI + C_LO)
        const float C_HI = 0.15915494f, C_LO = 6.4206382e-09f;
        uh[i] = v[i] * C_HI;
        ul[i] = fmaf(v[i], C_HI, -uh[i]) + v[i] * C_LO;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f0 = 32 * b + 16 * s + 8 * (j >> 2) + (j & 3);
                const PeSlot a = pe_slot<D, L>(f0), c = pe_slot<D, L>(f0 + 4);   // lane half 0 / 1
                if (a.kind == 0 && c.kind == 0) {
                    r[j] = 0.0f;
                } else {
                    const int kind = h ? c.kind : a.kind;
                    const float sn = sin_rev(h ? uh[c.axis] : uh[a.axis], h ? ul[c.axis] : ul[a.axis], h ? c.scale : a.scale, h ? c.phase : a.phase);
                    r[j] = (kind == 2) ? sn : ((kind == 1) ? (h ? v[c.axis] : v[a.axis]) : 0.0f);
                }
            }
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                uint32_t hi, lo;
                split_pair(r[2 * jp], r[2 * jp + 1], hi, lo);
                out[b].s[s][jp] = hi; out[b].l[s][jp] = lo;
            }
        }
}

#define CHX(id) (pick_GX(kProgH.layer[id].KB32, kProgH.layer[id].NT32) * kProgH.layer[id].KB32 * 2048)   /* halfwords in one chunk of layer id */

#if SAHS_MODEL == 0      // ---- the radiance kernel (AudioFaceModel) ----
// trilinear lookup (fp32, ATen corner order, zeros padding); this lane takes channels 16s + 8g + 4h + 0..3
__device__ __forceinline__ void grid_block_x(const float *__restrict__ grid, float x, float y, float z, int h, Blk &out)
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
        const f32x4 *g = reinterpret_cast<const f32x4 *>(grid + vox * D_GRID) + h;
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
        for (int jp = 0; jp < 4; ++jp) {
            const int j = 2 * jp;
            uint32_t hi, lo;
            split_pair(a[2 * s + (j >> 2)][j & 3], a[2 * s + (j >> 2)][(j & 3) + 1], hi, lo);
            out.s[s][jp] = hi; out.l[s][jp] = lo;
        }
}


// The radiance nets of `level` on S samples per ray whose (x', w) are xw[ray][src ? src[ray][s] : s] (field_f32.hip, FIELD_RADIANCE).
__global__ void __launch_bounds__(X_THREADS, 1)
field_radiance_bf16x3_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                             const float *__restrict__ rays, int ray_stride, float *__restrict__ raw, const float *__restrict__ xw, int xw_row,
                             const int *__restrict__ src)
{
    constexpr uint32_t RAD_OFF = (uint32_t)(2 * kProgH.layer[H_T0].stream_off);
    extern __shared__ __attribute__((aligned(16))) char lds_x[];
    Ctx cx;
    cx.stream = reinterpret_cast<const unsigned short *>(packed + PACKX_STREAM_OFF) + (long)level * STREAM_HWX;
    cx.lds = lds_x;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.h = cx.lane >> 5;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *grid = packed + PACKX_GRID_OFF;
    const int h = cx.h, col = cx.lane & 31;
    {
        const float *bsrc = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        float *bl = reinterpret_cast<float *>(lds_x + LDS_BIAS_BYTE_OFF);
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += X_THREADS) bl[i] = bsrc[i];
        cx.wrap_at = (uint32_t)STREAM_HWX;
        cx.wrap_to = RAD_OFF;
        cx.off = RAD_OFF;
        cx.prepare(CHX(H_T0), 0);
#pragma unroll
        for (int pc = 0; pc < (CHX(H_T0) + PIECE_HW - 1) / PIECE_HW; ++pc) cx.issue_piece(pc);
        __syncthreads();
    }
    constexpr const LayerH *Ly = kProgH.layer;

    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cx.refresh_bias_base();
        asm volatile("" : "+s"(cx.off));      // (chunk addresses are loop-invariant: keep them from being hoisted and spilled)
        St st;
        const long p_raw = tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE + col;
        const long p = p_raw < P ? p_raw : P - 1;
        typedef __attribute__((address_space(3))) float *lds_float;
        const lds_float stash = (lds_float)(__attribute__((address_space(3))) char *)(lds_x + LDS_STASH_BYTE_OFF) + (cx.wave * X_PTS_PER_WAVE + col) * STASH_FLOATS;
        if (h == 0) {
            const float *row = xw + ((p / S) * (long)xw_row + (src != nullptr ? src[p] : (int)(p % S))) * 8;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row);
            stash[0] = v[0]; stash[1] = v[1]; stash[2] = v[2]; stash[3] = v[3];
            stash[4] = row[4];
        }
        __builtin_amdgcn_wave_barrier();
        Blk A[8];
        f32x16 fin;
        {
            Blk B[8];
            {
                Blk in_tr[3];
                const float xp[3] = {stash[0], stash[1], stash[2]}, amb[3] = {stash[3], stash[4], 0.0f};
                pe_blocks_x<3, 10, 2>(xp, h, in_tr);
                pe_blocks_x<2, 4, 1>(amb, h, in_tr + 2);
                dense_x<2, 1, 0, 8, CHX(H_T1), false>(cx, st, in_tr, in_tr + 2, nullptr, A, Ly[H_T0].bias_off, FwdAct{0.01f}, FwdAct{1.0f});
            }
            dense_x<8, 0, 0, 8, CHX(H_T2), true>(cx, st, A, nullptr, nullptr, B, Ly[H_T1].bias_off, FwdAct{0.01f}, FwdAct{0.01f});
            dense_x<8, 0, 0, 8, CHX(H_T3), true>(cx, st, B, nullptr, nullptr, A, Ly[H_T2].bias_off, FwdAct{0.01f}, FwdAct{0.01f});
            {   // the re-injected encoding [PE(x') | PE(w)] is rebuilt at the skip layer instead of staying live
                Blk in_tr[3];
                const float xp[3] = {stash[0], stash[1], stash[2]}, amb[3] = {stash[3], stash[4], 0.0f};
                pe_blocks_x<3, 10, 2>(xp, h, in_tr);
                pe_blocks_x<2, 4, 1>(amb, h, in_tr + 2);
                dense_x<8, 2, 1, 8, CHX(H_T4), true>(cx, st, A, in_tr, in_tr + 2, B, Ly[H_T3].bias_off, FwdAct{0.01f}, FwdAct{0.01f});
            }
#pragma unroll 1
            for (int j = 0; j < 2; ++j) {     // T4, T5 | T6, T7 (identical shapes: one copy of the code, run twice)
                dense_x<8, 0, 0, 8, CHX(H_T5), true>(cx, st, B, nullptr, nullptr, A, Ly[H_T4].bias_off + 512 * j, FwdAct{0.01f}, FwdAct{0.01f});
                dense_x<8, 0, 0, 8, CHX(H_T5), true>(cx, st, A, nullptr, nullptr, B, Ly[H_T4].bias_off + 512 * j + 256, FwdAct{0.01f}, FwdAct{0.01f});
            }
            dense_x<8, 0, 0, 8, CHX(H_ALPHA), true>(cx, st, B, nullptr, nullptr, A, Ly[H_FEAT].bias_off, FwdAct{1.0f}, FwdAct{0.01f});
        }
        dense_x_out<8, CHX(H_D0)>(cx, st, A, fin, Ly[H_ALPHA].bias_off, true, FwdAct{1.0f});
        {   // colour branch
            Blk in_d[2];
            {
                const float *rq = rays + (p / S) * ray_stride;
                const float rdir[3] = {rq[3], rq[4], rq[5]};
                pe_blocks_x<3, 4, 1>(rdir, h, in_d);
                grid_block_x(grid, stash[0], stash[1], stash[2], h, in_d[1]);
            }
            Blk c[4], cn[4];
            dense_x<8, 1, 1, 4, CHX(H_D1), false>(cx, st, A, in_d, in_d + 1, c, Ly[H_D0].bias_off, FwdAct{0.01f}, FwdAct{1.0f});
            dense_x<4, 0, 0, 4, CHX(H_D1), true>(cx, st, c, nullptr, nullptr, cn, Ly[H_D1].bias_off, FwdAct{0.01f}, FwdAct{0.01f});
            dense_x<4, 0, 0, 4, CHX(H_D1), true>(cx, st, cn, nullptr, nullptr, c, Ly[H_D1].bias_off + 128, FwdAct{0.01f}, FwdAct{0.01f});
            dense_x<4, 0, 0, 4, CHX(H_RGB), true>(cx, st, c, nullptr, nullptr, cn, Ly[H_D3].bias_off, FwdAct{0.01f}, FwdAct{0.01f});
            dense_x_out<4, CHX(H_S0)>(cx, st, cn, fin, 0, false, FwdAct{0.01f});
        }
        {   // seg branch
            Blk s[4], sn[4];
            dense_x<8, 0, 0, 4, CHX(H_S1), false>(cx, st, A, nullptr, nullptr, s, Ly[H_S0].bias_off, FwdAct{0.01f}, FwdAct{1.0f});
            dense_x<4, 0, 0, 4, CHX(H_S1), true>(cx, st, s, nullptr, nullptr, sn, Ly[H_S1].bias_off, FwdAct{0.01f}, FwdAct{0.01f});
            dense_x<4, 0, 0, 4, CHX(H_S1), true>(cx, st, sn, nullptr, nullptr, s, Ly[H_S1].bias_off + 128, FwdAct{0.01f}, FwdAct{0.01f});
            dense_x<4, 0, 0, 4, CHX(H_SEG), true>(cx, st, s, nullptr, nullptr, sn, Ly[H_S3].bias_off, FwdAct{0.01f}, FwdAct{0.01f});
            dense_x_out<4, CHX(H_T0)>(cx, st, sn, fin, 0, false, FwdAct{0.01f});
        }
        if (p_raw < P) {   // rows 4h..4h+3 and 8+4h..8+4h+3 of [rgb3 | seg12 | sigma]
            *reinterpret_cast<f32x4 *>(raw + p_raw * D_RAW + 4 * h) = f32x4{fin[0], fin[1], fin[2], fin[3]};
            *reinterpret_cast<f32x4 *>(raw + p_raw * D_RAW + 8 + 4 * h) = f32x4{fin[4], fin[5], fin[6], fin[7]};
        }
    }
}

#endif      // SAHS_MODEL == 0

// The deformation nets (warp field + hyper sheet, level-independent inputs) on the depths zvals: x' = x + tanh(warp(PE(x))), w = hyper(PE(x))
// to xw[ray][xw_col0 + s] (field_f32.hip, FIELD_DEFORM; same arguments).  Round 3: with these launches on the split-operand pipe too the
// bf16x3 frame no longer contains an fp32-MFMA launch.  What that costs in accuracy was measured before it was built
// (tools/experiments/emulate_x3_deform.py, the eager restatement with split-operand linear layers, HDR weights): the coarse outputs move
// from 1.7e-5 to 2.8e-5 of the fp32 frame at worst -- inside four times the fp32 tolerance for every ray, the criterion of
// tests/test_gpu_bf16.py -- because x' = x + tanh(.) adds a SMALL correction to an exact x: the 6e-6 relative error of the nets lands on
// |dx| << 1, not on x' itself, before sin(2^9 x') amplifies it.
__global__ void __launch_bounds__(X_THREADS, 1)
field_deform_bf16x3_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                           const float *__restrict__ rays, int ray_stride, const float *__restrict__ zvals, float *__restrict__ xw, int xw_row,
                           int xw_col0)
{
    constexpr uint32_t RAD_OFF = (uint32_t)(2 * kProgH.layer[H_T0].stream_off);
    extern __shared__ __attribute__((aligned(16))) char lds_x[];
    Ctx cx;
    cx.stream = reinterpret_cast<const unsigned short *>(packed + PACKX_STREAM_OFF) + (long)level * STREAM_HWX;
    cx.lds = lds_x;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.h = cx.lane >> 5;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = cx.h, col = cx.lane & 31;
    {
        const float *bsrc = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        float *bl = reinterpret_cast<float *>(lds_x + LDS_BIAS_BYTE_OFF);
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += X_THREADS) bl[i] = bsrc[i];
        cx.wrap_at = RAD_OFF;          // the deformation nets are the stream's first layers: [0, RAD_OFF)
        cx.wrap_to = 0u;
        cx.off = 0u;
        cx.prepare(CHX(H_W0), 0);
#pragma unroll
        for (int pc = 0; pc < (CHX(H_W0) + PIECE_HW - 1) / PIECE_HW; ++pc) cx.issue_piece(pc);
        __syncthreads();
    }
    constexpr const LayerH *Ly = kProgH.layer;

    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cx.refresh_bias_base();
        asm volatile("" : "+s"(cx.off));      // (chunk addresses are loop-invariant: keep them from being hoisted and spilled)
        St st;
        const long p_raw = tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE + col;
        const long p = p_raw < P ? p_raw : P - 1;
        float x[3], xp[3], amb[2];
        {
            const float *rp = rays + (p / S) * ray_stride;
            const float z = zvals[p];
#pragma unroll
            for (int i = 0; i < 3; ++i) x[i] = rp[i] + rp[3 + i] * z;
        }
        constexpr int KX32 = (KB_XYZ + 1) / 2;        // 32-feature blocks of PE(x): 2 (10 octaves) | 3 (15 octaves, NeRFaceModel)
        Blk pe_x[KX32];
        pe_blocks_x<3, L_XYZ, KX32>(x, h, pe_x);
        {   // warp field (models.py:296-305; layers alternate between two register sets)
            Blk A[4], B[4];
            dense_x<KX32, 0, 0, 4, CHX(H_W1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_W0].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<4, 0, 0, 4, CHX(H_W1), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W1].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<4, 0, 0, 4, CHX(H_W1), true>(cx, st, B, nullptr, nullptr, A, Ly[H_W1].bias_off + 128, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<4, 0, 0, 4, CHX(H_W4), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W3].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<4, KX32, 0, 4, CHX(H_W5), true>(cx, st, B, pe_x, nullptr, A, Ly[H_W4].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<4, 0, 0, 4, CHX(H_WF), true>(cx, st, A, nullptr, nullptr, B, Ly[H_W5].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            f32x16 o;
            dense_x_out<4, CHX(H_H0)>(cx, st, B, o, Ly[H_WF].bias_off, true, FwdAct{0.0f});
            xp[0] = x[0] + tanhf(o[0]);            // models.py:305 (rows 0..2 live in lane half 0)
            xp[1] = x[1] + tanhf(o[1]);
            xp[2] = x[2] + tanhf(o[2]);
        }
        {   // hyper sheet (models.py:307-314)
            Blk A[2], B[2];
            dense_x<KX32, 0, 0, 2, CHX(H_H1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_H0].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<2, 0, 0, 2, CHX(H_H1), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H1].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<2, 0, 0, 2, CHX(H_H1), true>(cx, st, B, nullptr, nullptr, A, Ly[H_H1].bias_off + 64, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<2, 0, 0, 2, CHX(H_H4), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H3].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<2, KX32, 0, 2, CHX(H_H5), true>(cx, st, B, pe_x, nullptr, A, Ly[H_H4].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            dense_x<2, 0, 0, 2, CHX(H_HF), true>(cx, st, A, nullptr, nullptr, B, Ly[H_H5].bias_off, FwdAct{0.0f}, FwdAct{0.0f});
            f32x16 o;
            dense_x_out<2, CHX(H_W0)>(cx, st, B, o, Ly[H_HF].bias_off, true, FwdAct{0.0f});
            amb[0] = o[0];
            amb[1] = AMB_DIM > 1 ? o[1] : 0.0f;      // (NeRFaceModel: one ambient coordinate)
        }
        if (h == 0 && p_raw < P) {
            float *row = xw + ((p / S) * (long)xw_row + xw_col0 + (p % S)) * 8;
            *reinterpret_cast<f32x4 *>(row) = f32x4{xp[0], xp[1], xp[2], amb[0]};
            *reinterpret_cast<f32x4 *>(row + 4) = f32x4{amb[1], 0.0f, 0.0f, 0.0f};
        }
    }
}

}  // namespace hx3
}  // namespace SAHS_NS

using namespace SAHS_NS;
using namespace SAHS_NS::hx3;

#if SAHS_MODEL == 0
// the radiance launch of the split evaluation (field_f32.hip: sahs_field_forward_f32_split_launch mode 2, same arguments)
extern "C" int sahs_field_radiance_bf16x3_launch(const float *packed, const float *frame, int level, long P, int S, const float *rays,
                                                 int ray_stride, float *raw, const float *xw, int xw_row, const int *src, int num_cu,
                                                 hipStream_t stream)
{
    if (P <= 0) return 0;
    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_radiance_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    field_radiance_bf16x3_kernel<<<grid, X_THREADS, LDS_BYTES, stream>>>(packed, frame, level, P, S, rays, ray_stride, raw, xw, xw_row, src);
    return (int)hipGetLastError();
}

#endif

// the deformation launch of the split evaluation (field_f32.hip: sahs_field_forward_f32_split_launch mode 1, same arguments)
extern "C" int SAHS_SYM(sahs_field_deform_bf16x3_launch)(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride,
                                               const float *zvals, float *xw, int xw_row, int xw_col0, int num_cu, hipStream_t stream)
{
    if (P <= 0) return 0;
    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_deform_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    field_deform_bf16x3_kernel<<<grid, X_THREADS, LDS_BYTES, stream>>>(packed, frame, level, P, S, rays, ray_stride, zvals, xw, xw_row, xw_col0);
    return (int)hipGetLastError();
}
