// render_ops.hip -- the HBM-bound stages around the field kernel: ray generation, stratified
// depths, alpha compositing, importance resampling + merge-sort.  Each is one pass over its
// input with coalesced rows; none of them is on the critical path (<1 % of a frame), so they are
// written for exact, order-defined arithmetic first (sequential transmittance product, sequential
// pdf sum / cumsum -- the same orders the CPU oracle uses) and bandwidth second.
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"

namespace sahs {

// ---- get_ray_bundle: nerf_helpers.py:178-233 -------------------------------------------------
__global__ void ray_bundle_kernel(int H, int W, float fx, float fy, float cx, float cy, const float *__restrict__ c2w, int ld,
                                  float *__restrict__ ro, float *__restrict__ rd)
{
    const long n = (long)H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e / W), w = (int)(e % W);
        const float d0 = ((float)w - (float)W * cx) / fx;
        const float d1 = -((float)h - (float)H * cy) / fy;
        const float d2 = -1.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float s = d0 * c2w[i * ld + 0];
            s = s + d1 * c2w[i * ld + 1];
            s = s + d2 * c2w[i * ld + 2];
            rd[e * 3 + i] = s;
            ro[e * 3 + i] = c2w[i * ld + 3];
        }
    }
}

// ---- partition-invariant uniform draws -------------------------------------------
// The reference draws t_rand / u with torch.rand over the rays of a chunk, so the value a ray gets depends on how the frame
// was chunked or sharded.  Here a draw is a pure function of (seed, stream id, GLOBAL ray index, sample index):
// Philox4x32-10 (Salmon et al., SC'11) with key = seed, counter = (ray lo, ray hi, sample / 4, stream id); element s of a ray
// is word s % 4 of that block, mapped to [0,1) by its top 24 bits.  Integer work: bit-exact against the oracle.
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4])
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

__global__ void ray_uniforms_kernel(unsigned long long seed, int stream_id, long ray0, long N, int S, float *__restrict__ out)
{
    const int S4 = (S + 3) / 4;
    const long total = N * S4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / S4; const int b = (int)(e % S4);
        const unsigned long long gr = (unsigned long long)(ray0 + r);
        uint32_t c[4] = {(uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)b, (uint32_t)stream_id};
        philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * b + i < S) out[r * S + 4 * b + i] = (float)(c[i] >> 8) * 5.9604644775390625e-08f;   // 2^-24
    }
}

// ---- coarse depths: train_utils.py:93-113 ----------------------------------------------------
__device__ __forceinline__ float linspace01(int i, int n)   // torch.linspace(0, 1, n)[i], bit for bit: ATen's CPU kernel forms the second
{                                                            // half `end - step * k` as one fused multiply-add (oracle: aten_linspace01)
    if (n == 1) return 0.0f;
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? step * (float)i : __builtin_fmaf(-step, (float)(n - 1 - i), 1.0f);
}

__global__ void stratified_depths_kernel(long N, int S, const float *__restrict__ rays, int ray_stride, int lindisp,
                                         const float *__restrict__ t_rand, float *__restrict__ z)
{
    const long total = N * S;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / S; const int i = (int)(e % S);
        const float nr = rays[r * ray_stride + 6], fr = rays[r * ray_stride + 7];
        auto zat = [&](int k) {
            const float t = linspace01(k, S);
            return lindisp ? 1.0f / (1.0f / nr * (1.0f - t) + 1.0f / fr * t) : nr * (1.0f - t) + fr * t;
        };
        float v = zat(i);
        if (t_rand != nullptr) {
            const float upper = (i + 1 < S) ? 0.5f * (zat(i + 1) + v) : v;
            const float lower = (i > 0) ? 0.5f * (v + zat(i - 1)) : v;
            v = lower + (upper - lower) * t_rand[e];
        }
        z[e] = v;
    }
}

// ---- volume_render_radiance_field: volume_rendering_utils.py:7-78 ------------------------------
// One wave per ray, lane <-> sample (s = 64*i + lane).  Colour/alpha per sample in parallel; the
// exclusive transmittance product runs in the reference's order (cumprod: ((1*f0)*f1)*...), one
// step per sample with the factor broadcast from its lane; the 17 weighted sums use a wave
// butterfly.  bg != null: last sample's 15 colour channels are the prior (train_utils.py:135-136).
constexpr int COMP_MAX_I = 4;   // S <= 256

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__global__ void __launch_bounds__(256) composite_forward_kernel(long N, int S, const float *__restrict__ raw, const float *__restrict__ z,
                                                                const float *__restrict__ rays, int ray_stride,
                                                                const float *__restrict__ noise, const float *__restrict__ bg,
                                                                int white_bkgd, float *__restrict__ rgb_map, float *__restrict__ disp,
                                                                float *__restrict__ acc_map, float *__restrict__ weights,
                                                                float *__restrict__ depth, float *__restrict__ w_last, int rgb_ld, int sc_ld)
{   // rgb_ld / sc_ld: row strides of the 15-channel and of the per-ray scalar outputs (15 / 1 for separate dense arrays; 36 / 36
    // when they are columns of one (N,36) row block = the 8-tuple of a ray side by side, the unit the multi-GPU all-gather moves)
    const int lane = threadIdx.x & 63;
    const long ray0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * (blockDim.x >> 6);
    const int NI = (S + 63) >> 6;
    for (long ray = ray0; ray < N; ray += stride) {
        const float *rp = rays + ray * ray_stride;
        const float nrm = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
        float out[15];
#pragma unroll
        for (int c = 0; c < 15; ++c) out[c] = 0.0f;
        float dsum = 0.0f, asum = 0.0f, T = 1.0f;
        for (int i = 0; i < NI; ++i) {
            const int s = i * 64 + lane;
            const bool valid = s < S;
            const int sc = valid ? s : S - 1;
            const float *q = raw + (ray * S + sc) * D_RAW;
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const f32x4 *>(q)[k];
            float col[15];
            const bool last = (sc == S - 1);
            if (bg != nullptr) {
                float seg[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) seg[c] = v[(3 + c) >> 2][(3 + c) & 3];
                float mx = seg[0];
#pragma unroll
                for (int c = 1; c < 12; ++c) mx = seg[c] > mx ? seg[c] : mx;
                float es = 0.0f;
#pragma unroll
                for (int c = 0; c < 12; ++c) { seg[c] = expf(seg[c] - mx); es += seg[c]; }
#pragma unroll
                for (int c = 0; c < 3; ++c) col[c] = 1.0f / (1.0f + expf(-v[0][c]));
#pragma unroll
                for (int c = 0; c < 12; ++c) col[3 + c] = seg[c] / es;
                if (last) {
#pragma unroll
                    for (int c = 0; c < 15; ++c) col[c] = bg[ray * 15 + c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 15; ++c) col[c] = 1.0f / (1.0f + expf(-v[c >> 2][c & 3]));
            }
            const float zs = z[ray * S + sc];
            float dist = last ? 1e10f : (z[ray * S + sc + 1] - zs);
            dist = dist * nrm;
            float sg = v[3][3] + (noise != nullptr ? noise[ray * S + sc] : 0.0f);
            sg = sg > 0.0f ? sg : 0.0f;
            if (last) sg += 1e-6f;
            const float alpha = 1.0f - expf(-sg * dist);
            const float f = (1.0f - alpha) + 1e-10f;
            // exclusive cumprod in sample order, the reference's association ((T f0) f1) ...: P_k = P_{k-1} * f_k with P_{-1} = T carried in from
            // the previous 64 samples.  One in-place DPP multiply per step on ALL lanes (wave_shr:1 hands lane k the value of lane k - 1): after
            // step s lanes <= s hold their final product and every later step recomputes the same value from a final left neighbour -- 63
            // multiplies (+ their s_nop) instead of the 256 instructions of a broadcast (v_readlane), select and multiply per sample; same bits.
            const float fv = valid ? f : 1.0f;                    // lanes past S multiply by one: the carry stays the product of the real samples
            float Pk = T * fv;                                    // lane 0 is final from the start; lane k after step k
#pragma unroll
            for (int k = 0; k < 63; ++k)      // in place: lanes >= 1 take the left neighbour's product times their factor, lane 0 (no source lane) keeps its value
                asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(Pk) : "v"(fv));
            const float sh = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(T), __float_as_int(Pk), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
            const float myT = sh;                                 // T_k = P_{k-1}
            T = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Pk), 63));
            const float w = valid ? alpha * myT : 0.0f;
            if (valid) weights[ray * S + s] = w;
            if (valid && last && w_last != nullptr) w_last[ray * sc_ld] = w;
#pragma unroll
            for (int c = 0; c < 15; ++c) out[c] += w * col[c];
            dsum += w * zs;
            asum += w;
        }
#pragma unroll
        for (int c = 0; c < 15; ++c) out[c] = wave_sum(out[c]);
        dsum = wave_sum(dsum);
        asum = wave_sum(asum);
        if (white_bkgd) {
#pragma unroll
            for (int c = 0; c < 15; ++c) out[c] = out[c] + (1.0f - asum);
        }
        if (lane < 15) {
            float o = out[0];
#pragma unroll
            for (int c = 1; c < 15; ++c) o = (lane == c) ? out[c] : o;
            rgb_map[ray * rgb_ld + lane] = o;
        }
        if (lane == 0) {
            if (depth != nullptr) depth[ray * sc_ld] = dsum;
            acc_map[ray * sc_ld] = asum;
            const float dd = dsum / asum;
            disp[ray * sc_ld] = (dd != dd) ? dd : 1.0f / (dd > 1e-10f ? dd : 1e-10f);
        }
    }
}

// ---- sample_pdf_2 + cat + sort: nerf_helpers.py:454-497, train_utils.py:157-166 -----------------
// One wave per ray.  LDS per wave: cdf[S-1], bins[S-1], merge buffer[S+nf].
constexpr int RS_MAX = 256;           // S, nf <= 256
constexpr int RS_WAVES = 4;

// torch.sum of a contiguous fp32 row (n < 512) in the order ATen's CPU kernel adds it (SumKernel.cpp, vectorized_inner_sum / row_sum): n/8
// vectors of 8 lanes into four interleaved vector accumulators (vector 4i+k -> k; those past the last full group of four -> 0), accumulators
// 1..3 added to 0, then one scalar takes the n%8 trailing elements followed by the eight lanes, in order; rows shorter than a vector: the
// same with four scalar accumulators.  = oracle/sahs_oracle.c: aten_sum_f32.
__device__ inline float aten_row_sum(const float *x, int n)
{
    if (n < 8) {
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        const int g = n >> 2;
        for (int i = 0; i < g; ++i) { p0 += x[4 * i]; p1 += x[4 * i + 1]; p2 += x[4 * i + 2]; p3 += x[4 * i + 3]; }
        for (int i = 4 * g; i < n; ++i) p0 += x[i];
        p0 += p1; p0 += p2; p0 += p3;
        return p0;
    }
    const int vs = n >> 3, g = vs >> 2;
    float acc = 0.0f;
    for (int k = vs * 8; k < n; ++k) acc += x[k];
    for (int l = 0; l < 8; ++l) {
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        for (int i = 0; i < g; ++i) {
            p0 += x[(4 * i) * 8 + l]; p1 += x[(4 * i + 1) * 8 + l]; p2 += x[(4 * i + 2) * 8 + l]; p3 += x[(4 * i + 3) * 8 + l];
        }
        for (int i = 4 * g; i < vs; ++i) p0 += x[i * 8 + l];
        p0 += p1; p0 += p2; p0 += p3;
        acc += p0;
    }
    return acc;
}

// from_z = 1: z (N,S), weights (N,S) are the coarse depths / composite weights; bins = mids(z),
//             pdf weights = weights[:, 1:-1]; z_out (N,S+nf) = sort(cat(z, samples)).
// from_z = 0: plain sample_pdf_2 seam: z is bins (N,S-1 columns used as nb = S-1), weights is (N,nb-1); no merge.
__global__ void __launch_bounds__(RS_WAVES * 64) resample_kernel(long N, int S, int nf, int from_z, const float *__restrict__ z,
                                                                 const float *__restrict__ weights, const float *__restrict__ u_in,
                                                                 float *__restrict__ z_samples, float *__restrict__ z_out,
                                                                 long long *__restrict__ inds_out, int *__restrict__ src_out)
{   // src_out (N, S+nf), optional: the merge permutation -- position `rank` of the sorted row holds element src of cat(z, samples)
    __shared__ __attribute__((aligned(16))) float s_cdf[RS_WAVES][RS_MAX], s_bins[RS_WAVES][RS_MAX], s_val[RS_WAVES][2 * RS_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *cdf = s_cdf[wv], *bins = s_bins[wv], *val = s_val[wv];
    const int nb = S - 1;                     // bins; pdf has nb-1 entries
    const long stride = (long)gridDim.x * RS_WAVES;
    for (long ray = (long)blockIdx.x * RS_WAVES + wv; ray < N; ray += stride) {
        if (from_z) {
            const float *zr = z + ray * S, *wr = weights + ray * S;
            for (int i = lane; i < S; i += 64) val[i] = zr[i];
            for (int i = lane; i < nb; i += 64) bins[i] = 0.5f * (zr[i + 1] + zr[i]);
            for (int i = lane; i < nb - 1; i += 64) cdf[i + 1] = wr[i + 1] + 1e-5f;     // stash w' in cdf[1..]
        } else {
            const float *br = z + ray * nb, *wr = weights + ray * (nb - 1);
            for (int i = lane; i < nb; i += 64) bins[i] = br[i];
            for (int i = lane; i < nb - 1; i += 64) cdf[i + 1] = wr[i] + 1e-5f;
        }
        __builtin_amdgcn_wave_barrier();      // LDS ops of one wave execute in order; this only pins the compiler
        // torch.sum in ATen's own summation order, torch.cumsum accumulated in double and rounded per prefix (ATen CPU's acc_type<float>):
        // a cdf knot that moves by an ulp moves a searchsorted index.  Round 4: the same numbers with the work spread over the lanes --
        // (a) the eight vector-lane partial sums of ATen's row sum by eight lanes, (b) the quotients w'/sum one per lane, (c) the float64
        // prefix sums by a lane scan WHEN THAT IS EXACT: every partial sum of <= 255 floats whose exponents span <= 2^20 fits a double's 53
        // bits, so any order of additions gives the sequential cumsum's bits; otherwise (weights spanning more) the sequential loop.
        const int np = nb - 1;                 // pdf entries, w' in cdf[1 .. np]
        float sum;
        if (np >= 8) {
            const int vs = np >> 3, g = vs >> 2;
            if (lane < 8) {
                const float *x = cdf + 1;
                float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
                for (int i = 0; i < g; ++i) {
                    p0 += x[(4 * i) * 8 + lane]; p1 += x[(4 * i + 1) * 8 + lane]; p2 += x[(4 * i + 2) * 8 + lane]; p3 += x[(4 * i + 3) * 8 + lane];
                }
                for (int i = 4 * g; i < vs; ++i) p0 += x[i * 8 + lane];
                p0 += p1; p0 += p2; p0 += p3;
                val[S + lane] = p0;            // (val[S ..] is free until the samples are written)
            }
            __builtin_amdgcn_wave_barrier();
            float acc = 0.0f;
            for (int k = vs * 8; k < np; ++k) acc += cdf[1 + k];
            for (int l = 0; l < 8; ++l) acc += val[S + l];
            sum = acc;
        } else {
            sum = aten_row_sum(cdf + 1, np);
        }
        __builtin_amdgcn_wave_barrier();
        // quotients (one IEEE division per entry, as torch's weights / sum) and their exponent span
        int emin = 255, emax = 0;
        for (int i = lane; i < np; i += 64) {
            const float q = cdf[1 + i] / sum;
            cdf[1 + i] = q;
            const int e = (__float_as_int(q) >> 23) & 255;
            if (q != 0.0f) { emin = e < emin ? e : emin; emax = e > emax ? e : emax; }
            if (!(q == q) || e == 255) { emin = 0; emax = 255; }      // NaN / inf: take the sequential path
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int a_ = __shfl_xor(emin, o, WAVE), b_ = __shfl_xor(emax, o, WAVE);
            emin = a_ < emin ? a_ : emin; emax = b_ > emax ? b_ : emax;
        }
        __builtin_amdgcn_wave_barrier();
        if (emax - emin <= 20) {
            double carry = 0.0;
            for (int base = 0; base < np; base += 64) {
                const int i = base + lane;
                double c = i < np ? (double)cdf[1 + i] : 0.0;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double up = __shfl_up(c, o, WAVE);
                    if (lane >= o) c += up;
                }
                c += carry;
                carry = __shfl(c, 63, WAVE);
                __builtin_amdgcn_wave_barrier();
                if (i < np) cdf[1 + i] = (float)c;
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            double c = 0.0;
            for (int base = 1; base < nb; base += 64) {
                const int cnt = (nb - base) < 64 ? (nb - base) : 64;
                float keep = 0.0f;
                for (int t = 0; t < cnt; ++t) {
                    c += (double)cdf[base + t];
                    if (t == lane) keep = (float)c;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < cnt) cdf[base + lane] = keep;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) cdf[0] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < nf; j += 64) {
            const float u = (u_in != nullptr) ? u_in[ray * nf + j] : linspace01(j, nf);
            int lo = 0, hi = nb;                   // searchsorted(right=True): first idx with cdf[idx] > u
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
            const int below = lo - 1 > 0 ? lo - 1 : 0;
            const int above = lo < nb - 1 ? lo : nb - 1;
            float denom = cdf[above] - cdf[below];
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (u - cdf[below]) / denom;
            const float smp = bins[below] + t * (bins[above] - bins[below]);
            val[S + j] = smp;
            if (z_samples != nullptr) z_samples[ray * nf + j] = smp;
            if (inds_out != nullptr) inds_out[ray * nf + j] = lo;
        }
        __builtin_amdgcn_wave_barrier();
        // rank sort of the S+nf values (ascending; equal values keep index order).  The depths z of a ray arrive sorted (stratified bins), so
        // rank(z_a) = a + #{samples < z_a} and rank(sample_j) = #{z <= sample_j} + #{samples before it in the stable order}: half the
        // comparisons, four values per LDS read; rows that are not sorted (the seam accepts any z) take the general count.
        const int M = from_z ? S + nf : 0;
        bool sorted = true;
        for (int i = lane; i + 1 < S && M > 0; i += 64) sorted = sorted && (val[i] <= val[i + 1]);
        sorted = __all(sorted);
        if (M > 0 && sorted && (nf & 3) == 0 && (S & 3) == 0) {      // (16-byte LDS reads of the samples: S and nf multiples of 4)
            for (int a = lane; a < M; a += 64) {
                const float v = val[a];
                int rank;
                if (a < S) {
                    rank = a;
                    for (int b = 0; b < nf; b += 4) {
                        const f32x4 o = *reinterpret_cast<const f32x4 *>(val + S + b);
                        rank += (o[0] < v) + (o[1] < v) + (o[2] < v) + (o[3] < v);
                    }
                } else {
                    int lo = 0, hi = S;              // #{z_b <= v}: first index with z > v
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (val[mid] <= v) lo = mid + 1; else hi = mid; }
                    rank = lo;
                    const int j = a - S;
                    for (int b = 0; b < nf; b += 4) {
                        const f32x4 o = *reinterpret_cast<const f32x4 *>(val + S + b);
                        rank += (o[0] < v || (o[0] == v && b < j)) + (o[1] < v || (o[1] == v && b + 1 < j)) + (o[2] < v || (o[2] == v && b + 2 < j)) +
                                (o[3] < v || (o[3] == v && b + 3 < j));
                    }
                }
                z_out[ray * M + rank] = v;
                if (src_out != nullptr) src_out[ray * M + rank] = a;
            }
        } else {
            for (int a = lane; a < M; a += 64) {
                const float v = val[a];
                int rank = 0;
                for (int b = 0; b < M; ++b) {
                    const float o = val[b];
                    rank += (o < v || (o == v && b < a)) ? 1 : 0;
                }
                z_out[ray * M + rank] = v;
                if (src_out != nullptr) src_out[ray * M + rank] = a;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace sahs

using namespace sahs;

static inline int blocks_for(long n, int per_block, int cap = 4096)
{
    long b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" int sahs_ray_bundle_launch(int H, int W, float fx, float fy, float cx, float cy, const float *c2w, int ld, float *ro,
                                      float *rd, hipStream_t stream)
{
    ray_bundle_kernel<<<blocks_for((long)H * W, 256), 256, 0, stream>>>(H, W, fx, fy, cx, cy, c2w, ld, ro, rd);
    return (int)hipGetLastError();
}

extern "C" int sahs_stratified_depths_launch(long N, int S, const float *rays, int ray_stride, int lindisp, const float *t_rand,
                                             float *z, hipStream_t stream)
{
    if (N <= 0) return 0;
    stratified_depths_kernel<<<blocks_for(N * S, 256), 256, 0, stream>>>(N, S, rays, ray_stride, lindisp, t_rand, z);
    return (int)hipGetLastError();
}

extern "C" int sahs_composite_forward_launch(long N, int S, const float *raw, const float *z, const float *rays, int ray_stride,
                                             const float *noise, const float *bg, int white_bkgd, float *rgb_map, float *disp,
                                             float *acc_map, float *weights, float *depth, float *w_last, int rgb_ld, int sc_ld,
                                             hipStream_t stream)
{
    if (N <= 0) return 0;
    if (S < 1 || S > 64 * COMP_MAX_I) return -2;
    composite_forward_kernel<<<blocks_for(N, 4, 8192), 256, 0, stream>>>(N, S, raw, z, rays, ray_stride, noise, bg, white_bkgd,
                                                                        rgb_map, disp, acc_map, weights, depth, w_last, rgb_ld, sc_ld);
    return (int)hipGetLastError();
}

// The fine pass's gradient w.r.t. (x', w) of its samples, (N, Sc+nf, 8) rows in sorted-depth order, routed back through the merge
// permutation to the samples that produced them: slots < Sc are the coarse pass's samples, the others the new depths (training with
// the deformation nets evaluated once per depth; src is a permutation per ray, so every output row is written exactly once).
__global__ void __launch_bounds__(256) route_xw_grad_kernel(long N, int Sc, int nf, const int *__restrict__ src, const float *__restrict__ g_fine,
                                                            float *__restrict__ g_coarse, float *__restrict__ g_new)
{
    const int Sf = Sc + nf;
    const long total = N * Sf * 2;       // two float4 per row
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e >> 1;
        const int half = (int)(e & 1);
        const long ray = row / Sf;
        const int slot = src[row];
        const f32x4 v = reinterpret_cast<const f32x4 *>(g_fine)[e];
        float *dst = slot < Sc ? g_coarse + (ray * Sc + slot) * 8 : g_new + (ray * nf + (slot - Sc)) * 8;
        reinterpret_cast<f32x4 *>(dst)[half] = v;
    }
}

extern "C" int sahs_route_xw_grad_launch(long N, int Sc, int nf, const int *src, const float *g_fine, float *g_coarse, float *g_new, hipStream_t stream)
{
    const long total = N * (Sc + nf) * 2;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    route_xw_grad_kernel<<<(int)blocks, 256, 0, stream>>>(N, Sc, nf, src, g_fine, g_coarse, g_new);
    return (int)hipGetLastError();
}

extern "C" int sahs_resample_launch(long N, int S, int nf, int from_z, const float *z, const float *weights, const float *u,
                                    float *z_samples, float *z_out, long long *inds, int *src, hipStream_t stream)
{
    if (N <= 0) return 0;
    if (S < 3 || S > RS_MAX || nf < 1 || nf > RS_MAX) return -2;
    resample_kernel<<<blocks_for(N, RS_WAVES, 8192), RS_WAVES * 64, 0, stream>>>(N, S, nf, from_z, z, weights, u, z_samples, z_out,
                                                                                 inds, src);
    return (int)hipGetLastError();
}

extern "C" int sahs_ray_uniforms_launch(unsigned long long seed, int stream_id, long ray0, long N, int S, float *out, hipStream_t stream)
{
    if (N <= 0) return 0;
    const long total = N * ((S + 3) / 4);
    ray_uniforms_kernel<<<(unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), 256, 0, stream>>>(seed, stream_id, ray0, N, S, out);
    return (int)hipGetLastError();
}
