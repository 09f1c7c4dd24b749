// train_bwd.hip -- backward of the compositing stage and of the per-frame conditioning (AudioNet + fold).
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"

namespace sahs {

// ---- volume_render_radiance_field backward (volume_rendering_utils.py:7-78 under autograd) ------------------------
// One wave per ray, lane <-> sample.  Forward quantities are recomputed; the cumprod gradient is autograd's
// (suffix sum of dT*T divided by the factor).  Gradients: d_rgb (N,15), d_disp, d_acc, d_depth, d_wlast (N) -> d_raw (N,S,16).
__device__ __forceinline__ float rl(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
__device__ __forceinline__ float wsum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

struct Smp { float col[15]; float alpha, f, pre, dist, zs; };

// ---- Stage-I loss (train_stage_rays_auto.py:455-468 with nerf_helpers.py:14-62) fused around the compositing -----------------
//   loss = sum over levels {coarse, fine} of  mean_r l2_r + 0.02 mean_r ce_r + 0.005 sum_{c in 7,8} (M_c[l2] + M_c[ce])
//   l2_r = sum_ch (rgb_r,ch - target_r,ch)^2,  ce_r = -sum_c mask_r,c log(seg_r,c + 1e-10),  M_c[x] = sum_r x_r mask_r,c / max(1, #{r: mask_r,c != 0})
//   new sample_prob_c ~ sum over levels of w_c (M_c[l2] + M_c[ce])        (w = the loss modules' class weights, ones(12) with [7:9] = 2)
// stats (64 floats): [0] loss, [1] mse of the last level (the script's psnr), [2..13] new sample_prob, [14..25] class counts (>= 1), [26] rays
enum { STAT_LOSS = 0, STAT_MSE = 1, STAT_PROB = 2, STAT_CNT = 14, STAT_RAYS = 26, STAT_WORDS = 64 };
struct LossGrad {        // composite_backward_kernel adds d loss / d [rgb3 | seg12] of its level itself when map != nullptr
    const float *map;    // (N,15) the level's rendered [rgb3 | seg12] (the forward's output)
    const float *target; int target_ld;    // (N, >=3)
    const float *mask;   // (N,12)
    const float *stats;  // STAT_* above, of the whole batch
    const float *gscale; // device scalar: d (final objective) / d loss, or nullptr for 1
};

// one workgroup, fixed summation order (deterministic): thread t takes rays t, t+1024, ...; wave butterfly; waves summed in order
__global__ void __launch_bounds__(1024) stage1_loss_forward_kernel(long N, const float *__restrict__ map_c, const float *__restrict__ map_f,
                                                                   const float *__restrict__ target, int target_ld,
                                                                   const float *__restrict__ mask, const float *__restrict__ class_w,
                                                                   float *__restrict__ stats)
{
    constexpr int NV = 12 + 2 * 26;        // counts | per level: total l2, total ce, l2 per class, ce per class
    __shared__ float part[16][NV];
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
    for (long r = threadIdx.x; r < N; r += 1024) {
        const float *mk = mask + r * 12, *tg = target + r * target_ld;
        float m[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) { m[c] = mk[c]; acc[c] += (m[c] != 0.0f) ? 1.0f : 0.0f; }
#pragma unroll
        for (int L = 0; L < 2; ++L) {
            const float *mp = L == 0 ? map_c : map_f;
            if (mp == nullptr) continue;
            mp += r * 15;
            float l2 = 0.0f, ce = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float d = mp[c] - tg[c]; l2 += d * d; }
#pragma unroll
            for (int c = 0; c < 12; ++c) ce += m[c] * logf(mp[3 + c] + 1e-10f);
            ce = -ce;
            float *a = acc + 12 + 26 * L;
            a[0] += l2; a[1] += ce;
#pragma unroll
            for (int c = 0; c < 12; ++c) { a[2 + c] += l2 * m[c]; a[14 + c] += ce * m[c]; }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float v = wsum(acc[i]); if (lane == 0) part[wave][i] = v; }
    __syncthreads();
    if (threadIdx.x < NV) {
        float v = 0.0f;
        for (int w = 0; w < 16; ++w) v += part[w][threadIdx.x];
        part[0][threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float *t = part[0];
        float cnt[12], loss = 0.0f, mse = 0.0f, wsumv = 0.0f, wc[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) { cnt[c] = t[c] > 0.0f ? t[c] : 1.0f; wc[c] = 0.0f; }
        for (int L = 0; L < 2; ++L) {
            if ((L == 0 ? map_c : map_f) == nullptr) continue;
            const float *a = t + 12 + 26 * L;
            mse = a[0] / (float)N;
            float mouth = 0.0f;
            for (int c = 0; c < 12; ++c) {
                const float pl2 = a[2 + c] / cnt[c], pce = a[14 + c] / cnt[c];
                if (c == 7 || c == 8) mouth += pl2 + pce;
                wc[c] += class_w[c] * pl2 + class_w[c] * pce;
            }
            loss += mse + 0.02f * (a[1] / (float)N) + 0.005f * mouth;
        }
        for (int c = 0; c < 12; ++c) wsumv += wc[c];
        stats[STAT_LOSS] = loss;
        stats[STAT_MSE] = mse;
        for (int c = 0; c < 12; ++c) { stats[STAT_PROB + c] = wc[c] / wsumv; stats[STAT_CNT + c] = cnt[c]; }
        stats[STAT_RAYS] = (float)N;
    }
}

__device__ __forceinline__ void sample_fwd(const float *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ noise,
                                           const float *__restrict__ bg, long ray, int S, int sc, float nrm, Smp &o)
{
    const float *q = raw + (ray * S + sc) * D_RAW;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const f32x4 *>(q)[k];
    const bool last = (sc == S - 1);
    if (bg != nullptr) {
        float seg[12], mx = v[0][3];
#pragma unroll
        for (int c = 0; c < 12; ++c) { seg[c] = v[(3 + c) >> 2][(3 + c) & 3]; mx = seg[c] > mx ? seg[c] : mx; }
        float es = 0.0f;
#pragma unroll
        for (int c = 0; c < 12; ++c) { seg[c] = expf(seg[c] - mx); es += seg[c]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) o.col[c] = 1.0f / (1.0f + expf(-v[0][c]));
#pragma unroll
        for (int c = 0; c < 12; ++c) o.col[3 + c] = seg[c] / es;
        if (last) {
#pragma unroll
            for (int c = 0; c < 15; ++c) o.col[c] = bg[ray * 15 + c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 15; ++c) o.col[c] = 1.0f / (1.0f + expf(-v[c >> 2][c & 3]));
    }
    o.zs = z[ray * S + sc];
    o.dist = (last ? 1e10f : (z[ray * S + sc + 1] - o.zs)) * nrm;
    o.pre = v[3][3] + (noise != nullptr ? noise[ray * S + sc] : 0.0f);
    float sg = o.pre > 0.0f ? o.pre : 0.0f;
    if (last) sg += 1e-6f;
    o.alpha = 1.0f - expf(-sg * o.dist);
    o.f = (1.0f - o.alpha) + 1e-10f;
}

__global__ void __launch_bounds__(256) composite_backward_kernel(long N, int S, const float *__restrict__ raw, const float *__restrict__ z,
                                                                 const float *__restrict__ rays, int ray_stride,
                                                                 const float *__restrict__ noise, const float *__restrict__ bg,
                                                                 int white_bkgd, const float *__restrict__ d_rgb,
                                                                 const float *__restrict__ d_disp, const float *__restrict__ d_acc,
                                                                 const float *__restrict__ d_depth, const float *__restrict__ d_wlast,
                                                                 const float *__restrict__ d_weights, float *__restrict__ d_raw, LossGrad lg)
{
    const int lane = threadIdx.x & 63;
    const long stride = (long)gridDim.x * (blockDim.x >> 6);
    const int NI = (S + 63) >> 6;
    for (long ray = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); ray < N; ray += stride) {
        const float *rp = rays + ray * ray_stride;
        const float nrm = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
        // pass 1: totals and the transmittance at the start of every 64-sample block
        float Tstart[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        float T = 1.0f, dsum = 0.0f, asum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < NI) {
                Tstart[i] = T;
                const int s = i * 64 + lane, sc = s < S ? s : S - 1;
                Smp m; sample_fwd(raw, z, noise, bg, ray, S, sc, nrm, m);
                const int cnt = (S - i * 64) < 64 ? (S - i * 64) : 64;
                float myT = 1.0f;
                for (int k = 0; k < cnt; ++k) { if (lane == k) myT = T; T = T * rl(m.f, k); }
                const float w = (s < S) ? m.alpha * myT : 0.0f;
                dsum += w * m.zs; asum += w;
            }
        }
        dsum = wsum(dsum); asum = wsum(asum);
        float g[15], gsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 15; ++c) g[c] = d_rgb ? d_rgb[ray * 15 + c] : 0.0f;
        if (lg.map != nullptr) {      // the Stage-I loss's own gradient w.r.t. this ray's rendered [rgb3 | seg12] (see stage1_loss_forward_kernel)
            const float *mp = lg.map + ray * 15, *tg = lg.target + ray * lg.target_ld, *mk = lg.mask + ray * 12;
            const float gs = lg.gscale ? lg.gscale[0] : 1.0f;
            const float mouth = 0.005f * (mk[7] / lg.stats[STAT_CNT + 7] + mk[8] / lg.stats[STAT_CNT + 8]);
            const float inv = 1.0f / lg.stats[STAT_RAYS];
            const float a = gs * (inv + mouth), b = gs * (0.02f * inv + mouth);
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] += a * 2.0f * (mp[c] - tg[c]);
#pragma unroll
            for (int c = 0; c < 12; ++c) g[3 + c] -= b * mk[c] / (mp[3 + c] + 1e-10f);
        }
#pragma unroll
        for (int c = 0; c < 15; ++c) gsum += g[c];
        float gd = d_depth ? d_depth[ray] : 0.0f, ga = d_acc ? d_acc[ray] : 0.0f;
        const float gdisp = d_disp ? d_disp[ray] : 0.0f, gwl = d_wlast ? d_wlast[ray] : 0.0f;
        const float mq = dsum / asum;
        if (mq > 1e-10f) {            // disp = 1/max(1e-10, depth/acc): gradient only through the active branch
            const float dm = -gdisp / (mq * mq);
            gd += dm / asum;
            ga += -dm * dsum / (asum * asum);
        }
        if (white_bkgd) ga -= gsum;   // rgb += 1 - acc
        // pass 2: blocks in reverse, suffix carry of dT*T
        float carry = 0.0f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 3 - ii;
            if (i < NI) {
                const int s = i * 64 + lane, sc = s < S ? s : S - 1;
                const bool valid = s < S, last = (sc == S - 1);
                Smp m; sample_fwd(raw, z, noise, bg, ray, S, sc, nrm, m);
                const int cnt = (S - i * 64) < 64 ? (S - i * 64) : 64;
                float Tl = Tstart[i], myT = 1.0f;
                for (int k = 0; k < cnt; ++k) { if (lane == k) myT = Tl; Tl = Tl * rl(m.f, k); }
                const float w = valid ? m.alpha * myT : 0.0f;
                float dwv = gd * m.zs + ga + ((last && valid) ? gwl : 0.0f);
                if (d_weights != nullptr && valid) dwv += d_weights[ray * S + s];
#pragma unroll
                for (int c = 0; c < 15; ++c) dwv += g[c] * m.col[c];
                if (!valid) dwv = 0.0f;
                const float dT = dwv * m.alpha, tt = valid ? dT * myT : 0.0f;
                // exclusive suffix sum over samples (this block's later lanes + all later blocks)
                float suf = carry, mine = 0.0f;
                for (int k = cnt - 1; k >= 0; --k) { if (lane == k) mine = suf; suf += rl(tt, k); }
                carry = suf;
                float dalpha = dwv * myT - mine / m.f;
                const float dsig = dalpha * m.dist * (1.0f - m.alpha);
                f32x4 o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (!(bg != nullptr && last)) {
                    if (bg != nullptr) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) o[0][c] = w * g[c] * m.col[c] * (1.0f - m.col[c]);
                        float dot = 0.0f;
#pragma unroll
                        for (int c = 3; c < 15; ++c) dot += w * g[c] * m.col[c];
#pragma unroll
                        for (int c = 3; c < 15; ++c) o[c >> 2][c & 3] = m.col[c] * (w * g[c] - dot);
                    } else {
#pragma unroll
                        for (int c = 0; c < 15; ++c) o[c >> 2][c & 3] = w * g[c] * m.col[c] * (1.0f - m.col[c]);
                    }
                }
                o[3][3] = (m.pre > 0.0f) ? dsig : 0.0f;
                if (valid) {
                    f32x4 *dst = reinterpret_cast<f32x4 *>(d_raw + (ray * S + s) * D_RAW);
#pragma unroll
                    for (int k = 0; k < 4; ++k) dst[k] = o[k];
                }
            }
        }
    }
}

// ---- conditioning backward: d_driving[76] -> AudioNet parameters (+ d_audio) ------------------------------------
// AudioNet (modules.py:43-73) is 4 x [Conv1d(k3,s2,p1) + LeakyReLU(0.02)] + Linear 64->64 + LeakyReLU + Linear 64->76.
__device__ __forceinline__ float dl02(float y) { return y > 0.0f ? 1.0f : 0.02f; }

__global__ void __launch_bounds__(1024) conditioning_backward_kernel(const float *__restrict__ flat, const float *__restrict__ audio,
                                                                    const float *__restrict__ grad_cond, float *__restrict__ grad_flat,
                                                                    float *__restrict__ grad_audio)
{
    const FlatOffsets F = make_flat_offsets();
    __shared__ float a[5][64 * 16];      // a[0] = input (29x16), a[l+1] = output of conv l (post activation)
    __shared__ float d[2][64 * 16];
    __shared__ float h1[64], dh[76], dh1[64];
    const int tid = threadIdx.x;
    const int cin[4] = {29, 32, 32, 64}, cout[4] = {32, 32, 64, 64};
    for (int e = tid; e < 29 * 16; e += 1024) { int c = e / 16, t = e % 16; a[0][c * 16 + t] = audio[t * 29 + c]; }
    __syncthreads();
    int L = 16;
    for (int l = 0; l < 4; ++l) {
        const int Lo = L / 2;
        const float *w = flat + F.conv_w[l], *bs = flat + F.conv_b[l];
        for (int e = tid; e < cout[l] * Lo; e += 1024) {
            const int o = e / Lo, t = e % Lo;
            float s = bs[o];
            for (int c = 0; c < cin[l]; ++c)
                for (int k = 0; k < 3; ++k) {
                    const int ti = 2 * t + k - 1;
                    if (ti >= 0 && ti < L) s = fmaf(w[(o * cin[l] + c) * 3 + k], a[l][c * L + ti], s);
                }
            a[l + 1][o * Lo + t] = s > 0.0f ? s : s * 0.02f;
        }
        __syncthreads();
        L = Lo;
    }
    if (tid < 64) {
        float s = flat[F.fc_b[0] + tid];
        for (int k = 0; k < 64; ++k) s = fmaf(flat[F.fc_w[0] + tid * 64 + k], a[4][k], s);
        h1[tid] = s > 0.0f ? s : s * 0.02f;
    }
    if (tid < D_DRV) dh[tid] = grad_cond[tid];
    __syncthreads();
    // fc1.2: 64 -> 76
    for (int e = tid; e < D_DRV * 64; e += 1024) grad_flat[F.fc_w[1] + e] += dh[e / 64] * h1[e % 64];
    if (tid < D_DRV) grad_flat[F.fc_b[1] + tid] += dh[tid];
    if (tid < 64) {
        float s = 0.0f;
        for (int o = 0; o < D_DRV; ++o) s += flat[F.fc_w[1] + o * 64 + tid] * dh[o];
        dh1[tid] = s * dl02(h1[tid]);
    }
    __syncthreads();
    // fc1.0: 64 -> 64 (input a[4])
    for (int e = tid; e < 64 * 64; e += 1024) grad_flat[F.fc_w[0] + e] += dh1[e / 64] * a[4][e % 64];
    if (tid < 64) grad_flat[F.fc_b[0] + tid] += dh1[tid];
    if (tid < 64) {
        float s = 0.0f;
        for (int o = 0; o < 64; ++o) s += flat[F.fc_w[0] + o * 64 + tid] * dh1[o];
        d[0][tid] = s * dl02(a[4][tid]);     // grad wrt pre-activation of conv 3, shape (64,1)
    }
    __syncthreads();
    int cur = 0;
    int Lout = 1;
    for (int l = 3; l >= 0; --l) {
        const int Lin = Lout * 2;
        const float *w = flat + F.conv_w[l];
        float *gw = grad_flat + F.conv_w[l], *gb = grad_flat + F.conv_b[l];
        const float *dout = d[cur];
        float *din = d[cur ^ 1];
        for (int e = tid; e < cout[l] * cin[l] * 3; e += 1024) {
            const int o = e / (cin[l] * 3), c = (e / 3) % cin[l], k = e % 3;
            float s = 0.0f;
            for (int t = 0; t < Lout; ++t) {
                const int ti = 2 * t + k - 1;
                if (ti >= 0 && ti < Lin) s += dout[o * Lout + t] * a[l][c * Lin + ti];
            }
            gw[e] += s;
        }
        for (int o = tid; o < cout[l]; o += 1024) {
            float s = 0.0f;
            for (int t = 0; t < Lout; ++t) s += dout[o * Lout + t];
            gb[o] += s;
        }
        for (int e = tid; e < cin[l] * Lin; e += 1024) {
            const int c = e / Lin, ti = e % Lin;
            float s = 0.0f;
            for (int o = 0; o < cout[l]; ++o)
                for (int k = 0; k < 3; ++k) {
                    const int t2 = ti + 1 - k;           // 2t + k - 1 == ti
                    if (t2 >= 0 && (t2 & 1) == 0 && (t2 >> 1) < Lout) s += w[(o * cin[l] + c) * 3 + k] * dout[o * Lout + (t2 >> 1)];
                }
            din[e] = (l > 0) ? s * dl02(a[l][e]) : s;
        }
        __syncthreads();
        cur ^= 1;
        Lout = Lin;
    }
    if (grad_audio != nullptr)
        for (int e = tid; e < 29 * 16; e += 1024) { int c = e / 16, t = e % 16; grad_audio[t * 29 + c] += d[cur][c * 16 + t]; }
}

}  // namespace sahs

using namespace sahs;

extern "C" int sahs_composite_backward_launch(long N, int S, const float *raw, const float *z, const float *rays, int ray_stride,
                                              const float *noise, const float *bg, int white_bkgd, const float *d_rgb, const float *d_disp,
                                              const float *d_acc, const float *d_depth, const float *d_wlast, const float *d_weights,
                                              float *d_raw, const float *loss_map, const float *loss_target, int target_ld,
                                              const float *loss_mask, const float *loss_stats, const float *loss_gscale, hipStream_t stream)
{
    if (N <= 0) return 0;
    if (S < 1 || S > 256) return -2;
    long blocks = (N + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    composite_backward_kernel<<<(int)blocks, 256, 0, stream>>>(N, S, raw, z, rays, ray_stride, noise, bg, white_bkgd, d_rgb, d_disp, d_acc,
                                                               d_depth, d_wlast, d_weights, d_raw,
                                                               LossGrad{loss_map, loss_target, target_ld, loss_mask, loss_stats, loss_gscale});
    return (int)hipGetLastError();
}

extern "C" int sahs_stage1_loss_forward_launch(long N, const float *map_c, const float *map_f, const float *target, int target_ld,
                                               const float *mask, const float *class_w, float *stats, hipStream_t stream)
{
    stage1_loss_forward_kernel<<<1, 1024, 0, stream>>>(N, map_c, map_f, target, target_ld, mask, class_w, stats);
    return (int)hipGetLastError();
}

extern "C" int sahs_conditioning_backward_launch(const float *flat, const float *audio, const float *grad_cond, float *grad_flat,
                                                 float *grad_audio, hipStream_t stream)
{
    conditioning_backward_kernel<<<1, 1024, 0, stream>>>(flat, audio, grad_cond, grad_flat, grad_audio);
    return (int)hipGetLastError();
}
