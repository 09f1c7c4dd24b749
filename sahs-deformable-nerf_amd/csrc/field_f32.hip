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

namespace sahs {

constexpr int F32_THREADS = 512;
constexpr int F32_PTS_PER_WAVE = 16;
constexpr int F32_PTS_PER_WG = (F32_THREADS / WAVE) * F32_PTS_PER_WAVE;   // 128
constexpr int DBG_STRIDE = 56;   // test seam: [dx3, w2, T0[0], trunk layers 1..8 [0], D0[0], D3[0], S0[0], S3[0], pad]
constexpr int LDS_BUF_FLOATS = CHUNK_FLOATS_MAX;                           // 32 KB each, two of them
constexpr int LDS_BIAS_OFF = 2 * LDS_BUF_FLOATS;
constexpr int LDS_FLOATS = LDS_BIAS_OFF + ((BIAS_FLOATS + 3) / 4) * 4;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };
enum Init { INIT_BIAS = 0, INIT_ACCUM = 1 };

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

struct Ctx {
    const float *stream;      // this level's packed weight stream (global)
    const uint32_t *table;    // chunk start offsets (floats), NUM_CHUNKS + 1 entries
    float *lds;               // dynamic LDS base
    int chunk;                // index of the chunk currently resident (uniform)
    int buf;                  // LDS buffer (0/1) holding it; toggles per chunk (NUM_CHUNKS is odd, the stream wraps)
    int lane, q, wave;

    // async copy of chunk c into buffer b: global_load_lds writes LDS at (wave-uniform base + lane*16)
    __device__ __forceinline__ void issue(int c, int b)
    {
        const uint32_t o0 = table[c], o1 = table[c + 1];
        const int n16 = (int)(o1 - o0) >> 2;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(stream + o0);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds + b * LDS_BUF_FLOATS);
        for (int base = wave * WAVE; base < n16; base += F32_THREADS)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + base + lane), (lds_ptr_t)(dst + base), 16, 0, 0);
    }
    __device__ __forceinline__ void begin_chunk()
    {
        int nxt = chunk + 1;
        if (nxt == NUM_CHUNKS) nxt = 0;
        issue(nxt, buf ^ 1);
    }
    __device__ __forceinline__ void end_chunk()
    {
        __syncthreads();   // drains the in-flight global_load_lds (vmcnt(0)) and orders buffer reuse
        chunk = (chunk + 1 == NUM_CHUNKS) ? 0 : chunk + 1;
        buf ^= 1;
    }
    __device__ __forceinline__ const f32x4 *cur() const { return reinterpret_cast<const f32x4 *>(lds + buf * LDS_BUF_FLOATS); }
    __device__ __forceinline__ f32x4 bias4(int off) const   // off: float offset of a 16-row tile's bias
    {
        return *reinterpret_cast<const f32x4 *>(lds + LDS_BIAS_OFF + off + 4 * q);
    }
};

__device__ __forceinline__ f32x4 act4(f32x4 v, float slope)
{
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = v[r] > 0.0f ? v[r] : v[r] * slope;
    return o;
}

__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 acc)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[r], acc, 0, 0, 0);
    return acc;
}

// One dense layer.  in0[KB0] ++ in1[KB1] are the input k-blocks; out[NT] the output tiles.
// slope: 1 = no activation, 0 = relu, 0.01 = leaky relu.  accum: start from out[] instead of the bias.
template <int KB0, int KB1, int NT>
__device__ __forceinline__ void dense(Ctx &cx, const f32x4 *in0, const f32x4 *in1, f32x4 *out, int bias_off, bool accum, float slope)
{
    constexpr int KB = KB0 + KB1;
    constexpr int G = pick_G(KB, NT);
#pragma unroll
    for (int c = 0; c < NT / G; ++c) {
        cx.begin_chunk();
        const f32x4 *A = cx.cur() + cx.lane;
        if constexpr (G >= 2) {
#pragma unroll
            for (int g = 0; g < G; g += 2) {
                const int t0 = c * G + g;
                f32x4 acc0 = accum ? out[t0] : cx.bias4(bias_off + 16 * t0);
                f32x4 acc1 = accum ? out[t0 + 1] : cx.bias4(bias_off + 16 * (t0 + 1));
#pragma unroll
                for (int b = 0; b < KB; ++b) {
                    const f32x4 x = (b < KB0) ? in0[b] : in1[b - KB0];
                    acc0 = mfma4(A[(g * KB + b) * 64], x, acc0);
                    acc1 = mfma4(A[((g + 1) * KB + b) * 64], x, acc1);
                }
                out[t0] = act4(acc0, slope);
                out[t0 + 1] = act4(acc1, slope);
            }
        } else {
            // single tile per chunk: two accumulation chains over alternating k-blocks
            const int t0 = c;
            f32x4 acc0 = accum ? out[t0] : cx.bias4(bias_off + 16 * t0);
            f32x4 acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int b = 0; b < KB; b += 2) {
                acc0 = mfma4(A[b * 64], (b < KB0) ? in0[b] : in1[b - KB0], acc0);
                if (b + 1 < KB) acc1 = mfma4(A[(b + 1) * 64], (b + 1 < KB0) ? in0[b + 1] : in1[b + 1 - KB0], acc1);
            }
            out[t0] = act4(acc0 + acc1, slope);
        }
        cx.end_chunk();
    }
}

// ---- positional encoding in B layout ----------------------------------------------------------
// feature f of positional_encoding(v[0:d], L, include_input=True): [v | sin(2^0 v) | cos(2^0 v) | ...]
// (nerf_helpers.py:322-349); f >= d+2dL is zero padding.
template <int D, int L>
__device__ __forceinline__ float pe_feature(const float *v, int f)
{
    constexpr int W = D + 2 * D * L;
    float r = 0.0f;
    if (f < D) {
        r = (f == 0) ? v[0] : ((f == 1) ? v[1] : v[D > 2 ? 2 : 0]);
    } else if (f < W) {
        const int g = f - D;
        const int k = g / (2 * D), rem = g % (2 * D);
        const int fn = rem / D, ax = rem % D;
        const float x = (ax == 0) ? v[0] : ((ax == 1) ? v[1] : v[D > 2 ? 2 : 0]);
        const float t = x * (float)(1 << k);
        float s, c;
        sincosf(t, &s, &c);
        r = fn ? c : s;
    }
    return r;
}

template <int D, int L, int NB>
__device__ __forceinline__ void pe_blocks(const float *v, int q, f32x4 *out)
{
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[b][r] = pe_feature<D, L>(v, 16 * b + 4 * q + r);
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
        if (inb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { out[0][r] = out[0][r] + g0[r] * wt; out[1][r] = out[1][r] + g1[r] * wt; }
        }
    }
}

__device__ __forceinline__ float bcast16(float v, int lane) { return __shfl(v, lane & 15, WAVE); }

__global__ void __launch_bounds__(F32_THREADS, 2)
field_forward_f32_kernel(const float *__restrict__ packed, const float *__restrict__ frame, int level, long P, int S,
                         const float *__restrict__ rays, int ray_stride, const float *__restrict__ zvals,
                         float *__restrict__ raw, float *__restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Ctx cx;
    cx.stream = packed + PACK_STREAM_OFF + (long)level * STREAM_FLOATS;
    cx.table = reinterpret_cast<const uint32_t *>(packed + PACK_TABLE_OFF);
    cx.lds = lds;
    cx.chunk = 0;
    cx.buf = 0;
    cx.lane = threadIdx.x & 63;
    cx.q = cx.lane >> 4;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *grid = packed + PACK_GRID_OFF;
    const int q = cx.q;

    {   // per-level biases (static + folded conditioning) -> LDS, first weight chunk -> buffer 0
        const float *bsrc = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        for (int i = threadIdx.x; i < BIAS_FLOATS; i += F32_THREADS) lds[LDS_BIAS_OFF + i] = bsrc[i];
        cx.issue(0, 0);
        __syncthreads();
    }
    constexpr const Layer *Ly = kProg.layer;

    const long ntiles = (P + F32_PTS_PER_WG - 1) / F32_PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p_raw = tile * F32_PTS_PER_WG + cx.wave * F32_PTS_PER_WAVE + (cx.lane & 15);
        const long p = p_raw < P ? p_raw : P - 1;
        const long ray = p / S;
        const float *rp = rays + ray * ray_stride;
        const float z = zvals[p];
        float rd[3] = {rp[3], rp[4], rp[5]};
        float x[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) x[i] = rp[i] + rd[i] * z;          // train_utils.py:115

        f32x4 pe_x[4];
        pe_blocks<3, 10, 4>(x, q, pe_x);

        // ---- warp field: dx = tanh(MLP) (modules.py:371-390) ----
        float xw[3];
        {
            f32x4 h[8], hn[8];
            dense<4, 0, 8>(cx, pe_x, nullptr, h, Ly[L_W0].bias_off, false, 0.0f);
#pragma unroll 1
            for (int l = 0; l < 3; ++l) {
                dense<8, 0, 8>(cx, h, nullptr, hn, Ly[L_W1].bias_off + 128 * l, false, 0.0f);
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] = hn[i];
            }
            dense<4, 0, 8>(cx, pe_x, nullptr, hn, Ly[L_W4B].bias_off, false, 1.0f);
            dense<8, 0, 8>(cx, h, nullptr, hn, 0, true, 0.0f);
            dense<8, 0, 8>(cx, hn, nullptr, h, Ly[L_W5].bias_off, false, 0.0f);
            f32x4 o[1];
            dense<8, 0, 1>(cx, h, nullptr, o, Ly[L_WF].bias_off, false, 1.0f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float dx = tanhf(bcast16(o[0][i], cx.lane));
                xw[i] = x[i] + dx;                                       // models.py:305
            }
        }
        // ---- hyper sheet: ambient w (modules.py:444-462) ----
        float amb[2];
        {
            f32x4 h[4], hn[4];
            dense<4, 0, 4>(cx, pe_x, nullptr, h, Ly[L_H0].bias_off, false, 0.0f);
#pragma unroll 1
            for (int l = 0; l < 3; ++l) {
                dense<4, 0, 4>(cx, h, nullptr, hn, Ly[L_H1].bias_off + 64 * l, false, 0.0f);
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = hn[i];
            }
            dense<4, 0, 4>(cx, pe_x, nullptr, hn, Ly[L_H4B].bias_off, false, 1.0f);
            dense<4, 0, 4>(cx, h, nullptr, hn, 0, true, 0.0f);
            dense<4, 0, 4>(cx, hn, nullptr, h, Ly[L_H5].bias_off, false, 0.0f);
            f32x4 o[1];
            dense<4, 0, 1>(cx, h, nullptr, o, Ly[L_HF].bias_off, false, 1.0f);
            amb[0] = bcast16(o[0][0], cx.lane);
            amb[1] = bcast16(o[0][1], cx.lane);
        }
        const bool dump = dbg != nullptr && q == 0 && p_raw < P;   // lanes holding feature 0 of tile 0
        float *dsl = dbg + p * DBG_STRIDE;
        if (dump) { dsl[0] = xw[0] - x[0]; dsl[1] = xw[1] - x[1]; dsl[2] = xw[2] - x[2]; dsl[3] = amb[0]; dsl[4] = amb[1]; }

        // ---- radiance trunk (modules.py:254-275) ----
        f32x4 fin[1];      // FINAL tile: raw[4q..4q+3] of this lane's point
        f32x4 feat[16];
        {
            f32x4 in_tr[6];
            pe_blocks<3, 10, 4>(xw, q, in_tr);
            pe_blocks<2, 4, 2>(amb, q, in_tr + 4);
            f32x4 h[16];
            dense<4, 2, 16>(cx, in_tr, in_tr + 4, h, Ly[L_T0].bias_off, false, 0.01f);
            if (dump) dsl[5] = h[0][0];
            // T1, T2, [T3B], T3A, T4..T7, FEAT: eight 256x256 layers
#pragma unroll 1
            for (int l = 1; l <= 8; ++l) {
                if (l == 3) dense<4, 2, 16>(cx, in_tr, in_tr + 4, feat, Ly[L_T3B].bias_off, false, 1.0f);
                const int boff = (l < 3) ? Ly[L_T1].bias_off + 256 * (l - 1) : (l == 3 ? 0 : Ly[L_T4].bias_off + 256 * (l - 4));
                dense<16, 0, 16>(cx, h, nullptr, feat, boff, l == 3, l == 8 ? 1.0f : 0.01f);
                if (dump) dsl[5 + l] = feat[0][0];
                if (l < 8) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) h[i] = feat[i];
                }
            }
        }
        dense<16, 0, 1>(cx, feat, nullptr, fin, Ly[L_ALPHA].bias_off, false, 1.0f);
        if (dbg != nullptr && p_raw < P) *reinterpret_cast<f32x4 *>(dsl + 24 + 4 * q) = fin[0];
        // ---- colour branch (modules.py:276-287) ----
        {
            f32x4 in_d[4];
            pe_blocks<3, 4, 2>(rd, q, in_d);                          // models.py:340 (raw, un-normalised direction)
            grid_blocks(grid, xw[0], xw[1], xw[2], q, in_d + 2);      // models.py:525
            if (dbg != nullptr && p_raw < P) {
                float *d = dbg + P * DBG_STRIDE + p * 32;
                *reinterpret_cast<f32x4 *>(d + 4 * q) = in_d[2];
                *reinterpret_cast<f32x4 *>(d + 16 + 4 * q) = in_d[3];
            }
            f32x4 c[8], cn[8];
            dense<2, 2, 8>(cx, in_d, in_d + 2, c, Ly[L_D0B].bias_off, false, 1.0f);
            dense<16, 0, 8>(cx, feat, nullptr, c, 0, true, 0.01f);
            if (dump) dsl[14] = c[0][0];
#pragma unroll 1
            for (int l = 0; l < 3; ++l) {
                dense<8, 0, 8>(cx, c, nullptr, cn, Ly[L_D1].bias_off + 128 * l, false, 0.01f);
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = cn[i];
            }
            if (dump) dsl[15] = c[0][0];
            dense<8, 0, 1>(cx, c, nullptr, fin, 0, true, 1.0f);
            if (dbg != nullptr && p_raw < P) *reinterpret_cast<f32x4 *>(dsl + 40 + 4 * q) = fin[0];
        }
        // ---- seg branch (modules.py:289-294) ----
        {
            f32x4 s[8], sn[8];
            dense<16, 0, 8>(cx, feat, nullptr, s, Ly[L_S0].bias_off, false, 0.01f);
            if (dump) dsl[16] = s[0][0];
#pragma unroll 1
            for (int l = 0; l < 3; ++l) {
                dense<8, 0, 8>(cx, s, nullptr, sn, Ly[L_S1].bias_off + 128 * l, false, 0.01f);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = sn[i];
            }
            if (dump) dsl[17] = s[0][0];
            dense<8, 0, 1>(cx, s, nullptr, fin, 0, true, 1.0f);
        }
        if (p_raw < P) *reinterpret_cast<f32x4 *>(raw + p * D_RAW + 4 * q) = fin[0];   // cat((rgb, seg, alpha)) modules.py:295
    }
}

}  // namespace sahs

using namespace sahs;

// dbg (optional, may be null): [P x 24: see DBG_STRIDE][P x 32: grid features]
extern "C" int sahs_field_forward_f32_launch(const float *packed, const float *frame, int level, long P, int S, const float *rays,
                                             int ray_stride, const float *zvals, float *raw, float *dbg, int num_cu,
                                             hipStream_t stream)
{
    if (P <= 0) return 0;
    const long ntiles = (P + F32_PTS_PER_WG - 1) / F32_PTS_PER_WG;
    const int grid = (int)(ntiles < num_cu ? ntiles : num_cu);
    const size_t lds_bytes = (size_t)LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(field_forward_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    field_forward_f32_kernel<<<grid, F32_THREADS, lds_bytes, stream>>>(packed, frame, level, P, S, rays, ray_stride, zvals, raw, dbg);
    return (int)hipGetLastError();
}
