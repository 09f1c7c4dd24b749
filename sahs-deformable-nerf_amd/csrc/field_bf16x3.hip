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
// save (optional): this lane's slot of the sample's row of the encoding's saved-activation plane (row + 4 h floats): the fp32 values, natural
// feature order -- what the backward's PE derivative and weight-gradient jobs read (training with the forward on this pipe)
template <int D, int L, int NB>
__device__ __forceinline__ void pe_blocks_x(const float *v, int h, Blk *out, float *save = nullptr)
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
            if (save != nullptr) {
                *reinterpret_cast<f32x4 *>(save + 32 * b + 16 * s) = f32x4{r[0], r[1], r[2], r[3]};
                *reinterpret_cast<f32x4 *>(save + 32 * b + 16 * s + 8) = f32x4{r[4], r[5], r[6], r[7]};
            }
        }
}

// the epilogue policy of a layer: the plain activation, or (SAVE: training) the activation + its plane of the saved activations + its sign
// words.  col: the layer's column of the act:: table; BOFF: its word offset in the part's sign planes (< 0: no mask behind this layer)
template <bool SAVE, int WIDTH, int BOFF>
__device__ __forceinline__ auto layer_policy(float slope, float *actbuf, uint32_t *bits, long P, const SaveStage &sc, int col, int boff)
{
    if constexpr (SAVE) {
        SaveAct<WIDTH, (BOFF >= 0)> e;
        e.slope = slope;
        e.plane = actbuf + (long)col * P;
        e.sign = reinterpret_cast<unsigned char *>(bits + (long)boff * P);
        e.sc = sc;
        return e;
    } else {
        return FwdAct{slope};
    }
}

#define CHX(id) (pick_GX(kProgH.layer[id].KB32, kProgH.layer[id].NT32) * kProgH.layer[id].KB32 * 2048)   /* halfwords in one chunk of layer id */

#if SAHS_MODEL == 0      // ---- the radiance kernel (AudioFaceModel) ----
// trilinear lookup (fp32, ATen corner order, zeros padding); this lane takes channels 16s + 8g + 4h + 0..3
__device__ __forceinline__ void grid_block_x(const float *__restrict__ grid, float x, float y, float z, int h, Blk &out, float *save = nullptr)
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
    if (save != nullptr) {      // channels 8 k + 4 h .. + 3 of the sample's row of the GRID plane
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4 *>(save + 8 * k) = a[k];
    }
}


// The radiance nets of `level` on S samples per ray whose (x', w) are xw[ray][src ? src[ray][s] : s] (field_f32.hip, FIELD_RADIANCE).
// SAVE (training with the forward on this pipe): also the saved activations of the radiance part (actbuf: column c of the act:: table at
// actbuf + c * P) and its sign-bit planes (bits), exactly the buffers field_forward_f32_kernel<true, 2> writes for the backward.
template <bool SAVE>
__global__ void __launch_bounds__(X_THREADS, 1)
field_radiance_bf16x3_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                             const float *__restrict__ rays, int ray_stride, float *__restrict__ raw, const float *__restrict__ xw, int xw_row,
                             const int *__restrict__ src, float *__restrict__ actbuf, uint32_t *__restrict__ bits)
{
    constexpr uint32_t RAD_OFF = (uint32_t)(2 * kProgH.layer[H_T0].stream_off);
    // SAVE: the four staging tiles of the epilogue (bf16x3_pipe.hpp: SaveAct) -- wave 0's where the deformation nets' biases would be, the
    // others' from the x', w stash on (those five values stay in registers instead) to the end of the 160 KB
    constexpr int RAD_BIAS0 = kProgH.layer[H_T0].bias_off;
    static_assert(!SAVE || (RAD_BIAS0 * 4 >= SAVE_WAVE_BYTES && LDS_STASH_BYTE_OFF + (X_THREADS / WAVE - 1) * SAVE_WAVE_BYTES <= LDS_BYTES_SAVE), "staging tiles of the saving radiance kernel");
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
        // (SAVE: the biases of the deformation nets, which this kernel does not run, make room for a staging tile: RAD_BIAS0 floats)
        for (int i = threadIdx.x + (SAVE ? RAD_BIAS0 : 0); i < BIAS_FLOATS; i += X_THREADS) bl[i] = bsrc[i];
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
        // (SAVE: the backend moves the stream-offset chain to the VALU and then cannot feed the "s" constraint below -- "illegal VGPR to SGPR copy")
        if constexpr (SAVE) cx.off = (uint32_t)__builtin_amdgcn_readfirstlane((int)cx.off);
        asm volatile("" : "+s"(cx.off));      // (chunk addresses are loop-invariant: keep them from being hoisted and spilled)
        St st;
        const long p_raw = tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE + col;
        const long p = p_raw < P ? p_raw : P - 1;      // (SAVE: lanes past the end redo the last sample -- identical values to identical addresses)
        long Pq = P;
        if constexpr (SAVE) asm volatile("" : "+s"(Pq));      // (the ~25 plane bases c * P: keep them from being hoisted out of the tile loop and spilled)
        SaveStage sst;
        if constexpr (SAVE)
            sst = make_save_stage(lds_x, cx.wave == 0 ? (uint32_t)LDS_BIAS_BYTE_OFF : (uint32_t)(LDS_STASH_BYTE_OFF + (cx.wave - 1) * SAVE_WAVE_BYTES), cx.lane,
                                  tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE, p, P);
#define pol(slope, c, width, boff) layer_policy<SAVE, width, 0>(slope, actbuf, bits, Pq, sst, c, boff)
#define pol_nosign(slope, c, width) layer_policy<SAVE, width, -1>(slope, actbuf, bits, Pq, sst, c, 0)
#define plane(c, width) (SAVE ? actbuf + (long)(c) * Pq + p * (width) + 4 * h : nullptr)
        typedef __attribute__((address_space(3))) float *lds_float;
        const lds_float stash = (lds_float)(__attribute__((address_space(3))) char *)(lds_x + LDS_STASH_BYTE_OFF) + (cx.wave * X_PTS_PER_WAVE + col) * STASH_FLOATS;
        float sxw[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#define xwv(i) (SAVE ? sxw[i] : stash[i])      /* x'[3], w[2] of this lane's sample: SAVE in registers (both lanes of the sample load them), else parked in LDS */
        if (SAVE || h == 0) {
            const float *row = xw + ((p / S) * (long)xw_row + (src != nullptr ? src[p] : (int)(p % S))) * 8;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row);
            if constexpr (SAVE) {
                sxw[0] = v[0]; sxw[1] = v[1]; sxw[2] = v[2]; sxw[3] = v[3];
                sxw[4] = row[4];
                if (h == 0) { float *d = actbuf + (long)act::XW * Pq + p * 16; d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }      // the grid backward reads x'
            } else {
                stash[0] = v[0]; stash[1] = v[1]; stash[2] = v[2]; stash[3] = v[3];
                stash[4] = row[4];
            }
        }
        __builtin_amdgcn_wave_barrier();
        Blk A[8];
        f32x16 fin;
        {
            Blk B[8];
            const auto e0 = pol(0.01f, act::T, 256, sbits::BR_T);
            {
                Blk in_tr[3];
                const float xp[3] = {xwv(0), xwv(1), xwv(2)}, amb[3] = {xwv(3), xwv(4), 0.0f};
                pe_blocks_x<3, 10, 2>(xp, h, in_tr, plane(act::PEX, 64));
                pe_blocks_x<2, 4, 1>(amb, h, in_tr + 2, plane(act::PEW, 32));
                dense_x<2, 1, 0, 8, CHX(H_T1), false>(cx, st, in_tr, in_tr + 2, nullptr, A, Ly[H_T0].bias_off, e0, FwdAct{1.0f});
            }
            const auto e1 = pol(0.01f, act::T + 256, 256, sbits::BR_T + 8);
            dense_x<8, 0, 0, 8, CHX(H_T2), true, 7>(cx, st, A, nullptr, nullptr, B, Ly[H_T1].bias_off, e1, e0);
            const auto e2 = pol(0.01f, act::T + 512, 256, sbits::BR_T + 16);
            dense_x<8, 0, 0, 8, CHX(H_T3), true, 7>(cx, st, B, nullptr, nullptr, A, Ly[H_T2].bias_off, e2, e1);
            {   // the re-injected encoding [PE(x') | PE(w)] is rebuilt at the skip layer instead of staying live
                Blk in_tr[3];
                const float xp[3] = {xwv(0), xwv(1), xwv(2)}, amb[3] = {xwv(3), xwv(4), 0.0f};
                pe_blocks_x<3, 10, 2>(xp, h, in_tr);
                pe_blocks_x<2, 4, 1>(amb, h, in_tr + 2);
                dense_x<8, 2, 1, 8, CHX(H_T4), true, 7>(cx, st, A, in_tr, in_tr + 2, B, Ly[H_T3].bias_off, pol(0.01f, act::T + 768, 256, sbits::BR_T + 24), e2);
            }
#pragma unroll 1
            for (int j = 0; j < 2; ++j) {     // T4, T5 | T6, T7 (identical shapes: one copy of the code, run twice)
                const auto ea = pol(0.01f, act::T + 256 * (4 + 2 * j), 256, sbits::BR_T + 8 * (4 + 2 * j));
                dense_x<8, 0, 0, 8, CHX(H_T5), true, 7>(cx, st, B, nullptr, nullptr, A, Ly[H_T4].bias_off + 512 * j, ea,
                                                       pol(0.01f, act::T + 256 * (3 + 2 * j), 256, sbits::BR_T + 8 * (3 + 2 * j)));
                dense_x<8, 0, 0, 8, CHX(H_T5), true, 7>(cx, st, A, nullptr, nullptr, B, Ly[H_T4].bias_off + 512 * j + 256,
                                                       pol(0.01f, act::T + 256 * (5 + 2 * j), 256, sbits::BR_T + 8 * (5 + 2 * j)), ea);
            }
            dense_x<8, 0, 0, 8, CHX(H_ALPHA), true, 7>(cx, st, B, nullptr, nullptr, A, Ly[H_FEAT].bias_off, pol_nosign(1.0f, act::FEAT, 256),
                                                      pol(0.01f, act::T + 256 * 7, 256, sbits::BR_T + 8 * 7));
        }
        dense_x_out<8, CHX(H_D0), 7>(cx, st, A, fin, Ly[H_ALPHA].bias_off, true, pol_nosign(1.0f, act::FEAT, 256));
        {   // colour branch
            Blk in_d[2];
            {
                const float *rq = rays + (p / S) * ray_stride;
                const float rdir[3] = {rq[3], rq[4], rq[5]};
                pe_blocks_x<3, 4, 1>(rdir, h, in_d, plane(act::DIR, 32));
                grid_block_x(grid, xwv(0), xwv(1), xwv(2), h, in_d[1], plane(act::GRID, 32));
            }
            Blk c[4], cn[4];
            const auto d0 = pol(0.01f, act::C, 128, sbits::BR_C);
            dense_x<8, 1, 1, 4, CHX(H_D1), false>(cx, st, A, in_d, in_d + 1, c, Ly[H_D0].bias_off, d0, FwdAct{1.0f});
            const auto d1 = pol(0.01f, act::C + 128, 128, sbits::BR_C + 4);
            dense_x<4, 0, 0, 4, CHX(H_D1), true, 3>(cx, st, c, nullptr, nullptr, cn, Ly[H_D1].bias_off, d1, d0);
            const auto d2 = pol(0.01f, act::C + 256, 128, sbits::BR_C + 8);
            dense_x<4, 0, 0, 4, CHX(H_D1), true, 3>(cx, st, cn, nullptr, nullptr, c, Ly[H_D1].bias_off + 128, d2, d1);
            const auto d3 = pol(0.01f, act::C + 384, 128, sbits::BR_C + 12);
            dense_x<4, 0, 0, 4, CHX(H_RGB), true, 3>(cx, st, c, nullptr, nullptr, cn, Ly[H_D3].bias_off, d3, d2);
            dense_x_out<4, CHX(H_S0), 3>(cx, st, cn, fin, 0, false, d3);
        }
        {   // seg branch
            Blk s[4], sn[4];
            const auto s0 = pol(0.01f, act::S, 128, sbits::BR_S);
            dense_x<8, 0, 0, 4, CHX(H_S1), false>(cx, st, A, nullptr, nullptr, s, Ly[H_S0].bias_off, s0, FwdAct{1.0f});
            const auto s1 = pol(0.01f, act::S + 128, 128, sbits::BR_S + 4);
            dense_x<4, 0, 0, 4, CHX(H_S1), true, 3>(cx, st, s, nullptr, nullptr, sn, Ly[H_S1].bias_off, s1, s0);
            const auto s2 = pol(0.01f, act::S + 256, 128, sbits::BR_S + 8);
            dense_x<4, 0, 0, 4, CHX(H_S1), true, 3>(cx, st, sn, nullptr, nullptr, s, Ly[H_S1].bias_off + 128, s2, s1);
            const auto s3 = pol(0.01f, act::S + 384, 128, sbits::BR_S + 12);
            dense_x<4, 0, 0, 4, CHX(H_SEG), true, 3>(cx, st, s, nullptr, nullptr, sn, Ly[H_S3].bias_off, s3, s2);
            dense_x_out<4, CHX(H_T0), 3>(cx, st, sn, fin, 0, false, s3);
        }
        if (p_raw < P) {   // rows 4h..4h+3 and 8+4h..8+4h+3 of [rgb3 | seg12 | sigma]
            *reinterpret_cast<f32x4 *>(raw + p_raw * D_RAW + 4 * h) = f32x4{fin[0], fin[1], fin[2], fin[3]};
            *reinterpret_cast<f32x4 *>(raw + p_raw * D_RAW + 8 + 4 * h) = f32x4{fin[4], fin[5], fin[6], fin[7]};
        }
    }
}

#undef pol
#undef xwv
#undef pol_nosign
#undef plane
#endif      // SAHS_MODEL == 0

// The deformation nets (warp field + hyper sheet, level-independent inputs) on the depths zvals: x' = x + tanh(warp(PE(x))), w = hyper(PE(x))
// to xw[ray][xw_col0 + s] (field_f32.hip, FIELD_DEFORM; same arguments).  Round 3: with these launches on the split-operand pipe too the
// bf16x3 frame no longer contains an fp32-MFMA launch.  What that costs in accuracy was measured before it was built
// (tools/experiments/emulate_x3_deform.py, the eager restatement with split-operand linear layers, HDR weights): the coarse outputs move
// from 1.7e-5 to 2.8e-5 of the fp32 frame at worst -- inside four times the fp32 tolerance for every ray, the criterion of
// tests/test_gpu_bf16.py -- because x' = x + tanh(.) adds a SMALL correction to an exact x: the 6e-6 relative error of the nets lands on
// |dx| << 1, not on x' itself, before sin(2^9 x') amplifies it.
// SAVE (training with the forward on this pipe, AudioFaceModel): also the saved activations and sign-bit planes of the deformation part, the
// buffers field_forward_f32_kernel<true, 1> writes for the backward.
template <bool SAVE>
__global__ void __launch_bounds__(X_THREADS, 1)
field_deform_bf16x3_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                           const float *__restrict__ rays, int ray_stride, const float *__restrict__ zvals, float *__restrict__ xw, int xw_row,
                           int xw_col0, float *__restrict__ actbuf, uint32_t *__restrict__ bits)
{
    constexpr uint32_t RAD_OFF = (uint32_t)(2 * kProgH.layer[H_T0].stream_off);
    constexpr int DEF_BIAS1 = kProgH.layer[H_T0].bias_off;      // the deformation nets' biases are the first of the level's array
    static_assert(!SAVE || LDS_BIAS_BYTE_OFF + DEF_BIAS1 * 4 + (X_THREADS / WAVE) * SAVE_WAVE_BYTES <= LDS_BYTES_SAVE, "staging tiles of the saving deformation kernel");
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
        // (SAVE: only the deformation nets' biases; the staging tiles of the epilogue take the place of the others': SaveAct)
        for (int i = threadIdx.x; i < (SAVE ? DEF_BIAS1 : BIAS_FLOATS); i += X_THREADS) bl[i] = bsrc[i];
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
        // (SAVE: the backend moves the stream-offset chain to the VALU and then cannot feed the "s" constraint below -- "illegal VGPR to SGPR copy")
        if constexpr (SAVE) cx.off = (uint32_t)__builtin_amdgcn_readfirstlane((int)cx.off);
        asm volatile("" : "+s"(cx.off));      // (chunk addresses are loop-invariant: keep them from being hoisted and spilled)
        St st;
        const long p_raw = tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE + col;
        const long p = p_raw < P ? p_raw : P - 1;
        long Pq = P;
        if constexpr (SAVE) asm volatile("" : "+s"(Pq));      // (plane bases c * P: not hoisted out of the tile loop)
        SaveStage sst;
        if constexpr (SAVE)
            sst = make_save_stage(lds_x, (uint32_t)(LDS_BIAS_BYTE_OFF + DEF_BIAS1 * 4 + cx.wave * SAVE_WAVE_BYTES), cx.lane, tile * X_PTS_PER_WG + cx.wave * X_PTS_PER_WAVE, p, P);
#define pold(c, width, boff) layer_policy<SAVE, width, 0>(0.0f, actbuf, bits, Pq, sst, c, boff)
        float x[3], xp[3], amb[2], dx[3];
        {
            const float *rp = rays + (p / S) * ray_stride;
            const float z = zvals[p];
#pragma unroll
            for (int i = 0; i < 3; ++i) x[i] = rp[i] + rp[3 + i] * z;
        }
        constexpr int KX32 = (KB_XYZ + 1) / 2;        // 32-feature blocks of PE(x): 2 (10 octaves) | 3 (15 octaves, NeRFaceModel)
        Blk pe_x[KX32];
        pe_blocks_x<3, L_XYZ, KX32>(x, h, pe_x, SAVE ? actbuf + (long)act::E * Pq + p * (16 * KB_XYZ) + 4 * h : nullptr);
        {   // warp field (models.py:296-305; layers alternate between two register sets)
            Blk A[4], B[4];
            const auto w0 = pold(act::WH, 128, sbits::BD_WH);
            dense_x<KX32, 0, 0, 4, CHX(H_W1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_W0].bias_off, w0, FwdAct{0.0f});
            const auto w1 = pold(act::WH + 128, 128, sbits::BD_WH + 4);
            dense_x<4, 0, 0, 4, CHX(H_W1), true, 3>(cx, st, A, nullptr, nullptr, B, Ly[H_W1].bias_off, w1, w0);
            const auto w2 = pold(act::WH + 256, 128, sbits::BD_WH + 8);
            dense_x<4, 0, 0, 4, CHX(H_W1), true, 3>(cx, st, B, nullptr, nullptr, A, Ly[H_W1].bias_off + 128, w2, w1);
            const auto w3 = pold(act::WH + 384, 128, sbits::BD_WH + 12);
            dense_x<4, 0, 0, 4, CHX(H_W4), true, 3>(cx, st, A, nullptr, nullptr, B, Ly[H_W3].bias_off, w3, w2);
            const auto w4 = pold(act::WH + 512, 128, sbits::BD_WH + 16);
            dense_x<4, KX32, 0, 4, CHX(H_W5), true, 3>(cx, st, B, pe_x, nullptr, A, Ly[H_W4].bias_off, w4, w3);
            const auto w5 = pold(act::WH + 640, 128, sbits::BD_WH + 20);
            dense_x<4, 0, 0, 4, CHX(H_WF), true, 3>(cx, st, A, nullptr, nullptr, B, Ly[H_W5].bias_off, w5, w4);
            f32x16 o;
            dense_x_out<4, CHX(H_H0), 3>(cx, st, B, o, Ly[H_WF].bias_off, true, w5);
            dx[0] = tanhf(o[0]); dx[1] = tanhf(o[1]); dx[2] = tanhf(o[2]);
            xp[0] = x[0] + dx[0];            // models.py:305 (rows 0..2 live in lane half 0)
            xp[1] = x[1] + dx[1];
            xp[2] = x[2] + dx[2];
        }
        {   // hyper sheet (models.py:307-314)
            Blk A[2], B[2];
            const auto g0 = pold(act::HH, 64, sbits::BD_HH);
            dense_x<KX32, 0, 0, 2, CHX(H_H1), false>(cx, st, pe_x, nullptr, nullptr, A, Ly[H_H0].bias_off, g0, FwdAct{0.0f});
            const auto g1 = pold(act::HH + 64, 64, sbits::BD_HH + 4);
            dense_x<2, 0, 0, 2, CHX(H_H1), true, 1>(cx, st, A, nullptr, nullptr, B, Ly[H_H1].bias_off, g1, g0);
            const auto g2 = pold(act::HH + 128, 64, sbits::BD_HH + 8);
            dense_x<2, 0, 0, 2, CHX(H_H1), true, 1>(cx, st, B, nullptr, nullptr, A, Ly[H_H1].bias_off + 64, g2, g1);
            const auto g3 = pold(act::HH + 192, 64, sbits::BD_HH + 12);
            dense_x<2, 0, 0, 2, CHX(H_H4), true, 1>(cx, st, A, nullptr, nullptr, B, Ly[H_H3].bias_off, g3, g2);
            const auto g4 = pold(act::HH + 256, 64, sbits::BD_HH + 16);
            dense_x<2, KX32, 0, 2, CHX(H_H5), true, 1>(cx, st, B, pe_x, nullptr, A, Ly[H_H4].bias_off, g4, g3);
            const auto g5 = pold(act::HH + 320, 64, sbits::BD_HH + 20);
            dense_x<2, 0, 0, 2, CHX(H_HF), true, 1>(cx, st, A, nullptr, nullptr, B, Ly[H_H5].bias_off, g5, g4);
            f32x16 o;
            dense_x_out<2, CHX(H_W0), 1>(cx, st, B, o, Ly[H_HF].bias_off, true, g5);
            amb[0] = o[0];
            amb[1] = AMB_DIM > 1 ? o[1] : 0.0f;      // (NeRFaceModel: one ambient coordinate)
        }
        if (h == 0 && p_raw < P) {
            float *row = xw + ((p / S) * (long)xw_row + xw_col0 + (p % S)) * 8;
            *reinterpret_cast<f32x4 *>(row) = f32x4{xp[0], xp[1], xp[2], amb[0]};
            *reinterpret_cast<f32x4 *>(row + 4) = f32x4{amb[1], 0.0f, 0.0f, 0.0f};
            if constexpr (SAVE) {      // what the backward reads of the heads: dx = tanh(.), x', w (field_f32.hip writes the same three slots)
                float *d = actbuf + (long)act::DX * Pq + p * 16, *q = actbuf + (long)act::XW * Pq + p * 16, *w = actbuf + (long)act::AW * Pq + p * 16;
                d[0] = dx[0]; d[1] = dx[1]; d[2] = dx[2];
                q[0] = xp[0]; q[1] = xp[1]; q[2] = xp[2];
                w[0] = amb[0]; w[1] = amb[1];
            }
        }
    }
}

#undef pold
}  // namespace hx3
}  // namespace SAHS_NS

using namespace SAHS_NS;
using namespace SAHS_NS::hx3;

#if SAHS_MODEL == 0
template <bool SAVE>
static int launch_radiance_x3(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride, float *raw,
                              const float *xw, int xw_row, const int *src, float *actbuf, uint32_t *bits, int num_cu, hipStream_t stream)
{
    if (P <= 0) return 0;
    if (SAVE && P * (long)(4 * 256) >= (1L << 32)) return -4;      // (SaveAct: 32-bit byte offsets inside a layer's plane)
    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_radiance_bf16x3_kernel<SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, SAVE ? LDS_BYTES_SAVE : LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    field_radiance_bf16x3_kernel<SAVE><<<grid, X_THREADS, SAVE ? LDS_BYTES_SAVE : LDS_BYTES, stream>>>(packed, frame, level, P, S, rays, ray_stride, raw, xw, xw_row, src, actbuf, bits);
    return (int)hipGetLastError();
}
// the radiance launch of the split evaluation (field_f32.hip: sahs_field_forward_f32_split_launch mode 2, same arguments)
extern "C" int sahs_field_radiance_bf16x3_launch(const float *packed, const float *frame, int level, long P, int S, const float *rays,
                                                 int ray_stride, float *raw, const float *xw, int xw_row, const int *src, int num_cu,
                                                 hipStream_t stream)
{
    return launch_radiance_x3<false>(packed, frame, level, P, S, rays, ray_stride, raw, xw, xw_row, src, nullptr, nullptr, num_cu, stream);
}
// ... that also saves what the backward reads (field_f32.hip: sahs_field_forward_f32_split_bits_launch mode 2 with actbuf and bits, same buffers)
extern "C" int sahs_field_radiance_bf16x3_save_launch(const float *packed, const float *frame, int level, long P, int S, const float *rays,
                                                      int ray_stride, float *raw, const float *xw, int xw_row, const int *src, float *actbuf,
                                                      uint32_t *bits, int num_cu, hipStream_t stream)
{
    if (actbuf == nullptr || bits == nullptr) return -2;
    return launch_radiance_x3<true>(packed, frame, level, P, S, rays, ray_stride, raw, xw, xw_row, src, actbuf, bits, num_cu, stream);
}
#endif

template <bool SAVE>
static int launch_deform_x3(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride, const float *zvals,
                            float *xw, int xw_row, int xw_col0, float *actbuf, uint32_t *bits, int num_cu, hipStream_t stream)
{
    if (P <= 0) return 0;
    if (SAVE && P * (long)(4 * 128) >= (1L << 32)) return -4;
    const long ntiles = (P + X_PTS_PER_WG - 1) / X_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    static sahs_once::Flags attr_set;
    hipError_t ae = sahs_once::per_device(attr_set, [&]() {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(field_deform_bf16x3_kernel<SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, SAVE ? LDS_BYTES_SAVE : LDS_BYTES);
    });
    if (ae != hipSuccess) return (int)ae;
    field_deform_bf16x3_kernel<SAVE><<<grid, X_THREADS, SAVE ? LDS_BYTES_SAVE : LDS_BYTES, stream>>>(packed, frame, level, P, S, rays, ray_stride, zvals, xw, xw_row, xw_col0, actbuf, bits);
    return (int)hipGetLastError();
}
// the deformation launch of the split evaluation (field_f32.hip: sahs_field_forward_f32_split_launch mode 1, same arguments)
extern "C" int SAHS_SYM(sahs_field_deform_bf16x3_launch)(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride,
                                               const float *zvals, float *xw, int xw_row, int xw_col0, int num_cu, hipStream_t stream)
{
    return launch_deform_x3<false>(packed, frame, level, P, S, rays, ray_stride, zvals, xw, xw_row, xw_col0, nullptr, nullptr, num_cu, stream);
}
#if SAHS_MODEL == 0
extern "C" int sahs_field_deform_bf16x3_save_launch(const float *packed, const float *frame, int level, long P, int S, const float *rays, int ray_stride,
                                                    const float *zvals, float *xw, int xw_row, int xw_col0, float *actbuf, uint32_t *bits, int num_cu,
                                                    hipStream_t stream)
{
    if (actbuf == nullptr || bits == nullptr) return -2;
    return launch_deform_x3<true>(packed, frame, level, P, S, rays, ray_stride, zvals, xw, xw_row, xw_col0, actbuf, bits, num_cu, stream);
}
#endif
