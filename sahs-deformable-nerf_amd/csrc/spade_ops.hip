// spade_ops.hip -- the elementwise core of the Stage-II refiner's SPADE layer (reference
// nerf/_init_spade.py:114-160):   out = act( InstanceNorm2d(x) * (1 + gamma) + beta ),   act = LeakyReLU(slope) of the SPADEBlock that
// follows every SPADE layer (:262-279; slope 1 = none).  The reference runs this as InstanceNorm (a batch-norm kernel), an add, a multiply,
// an add and an activation -- six passes over the (N, C, H, W) tensor; here: two statistics passes (the plane comes from L2 the second
// time) and ONE fused modulate pass.  HBM-bound: 4 B read for the statistics + 12 B read + 4 B written per element = 20 B
// per element (bench.py's `spade` leg reports the achieved GB/s against that).  The 3x3
// convolutions around it (label map -> 128 -> gamma / beta, and the block's spectral-normalised convolutions) are library work and
// stay on MIOpen through PyTorch (hand-written convolutions would buy nothing on this path).
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"

namespace sahs {

// Instance statistics in two deterministic passes over (chunk, plane) workgroups -- a workgroup per plane left 3/4 of the chip idle on the
// refiner's 64- and 128-channel maps: pass 0 writes each chunk's sum, pass 1 re-adds the plane's chunk sums in a fixed order (the mean),
// and writes each chunk's sum of squares about it; the modulate pass re-adds those.  part: [planes][SPADE_CHUNKS][2] floats.
constexpr int SPADE_CHUNKS = 8;

__device__ __forceinline__ float block_sum256(float v, float *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float plane_mean(const float *part, long hw)
{
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < SPADE_CHUNKS; ++c) s += part[2 * c];
    return s / (float)hw;
}

template <bool VEC>
__global__ void __launch_bounds__(256) instance_stats_kernel(long hw, const float *__restrict__ x, float *__restrict__ part, int pass)
{
    __shared__ float red[4];
    const long plane = blockIdx.y;
    const float *p = x + plane * hw;
    float *pp = part + plane * (2 * SPADE_CHUNKS);
    const long n = VEC ? hw / 4 : hw, per = (n + SPADE_CHUNKS - 1) / SPADE_CHUNKS;
    const long lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const float mean = pass ? plane_mean(pp, hw) : 0.0f;
    float s = 0.0f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        if (VEC) {
            const f32x4 q = reinterpret_cast<const f32x4 *>(p)[i];
            if (pass) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = q[k] - mean; s += d * d; }
            } else {
                s += (q[0] + q[1]) + (q[2] + q[3]);
            }
        } else {
            const float d = p[i] - mean;
            s += pass ? d * d : p[i];
        }
    }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) pp[2 * blockIdx.x + pass] = s;
}

// the fused modulate pass: 2-D grid (chunk of the plane, plane) -- the plane index is blockIdx.y, no per-element divide -- and one 16-byte
// load per operand and thread where the plane allows it: 12 B read + 4 B written per element (the statistics pass reads 4 B more)
template <bool VEC>
__global__ void __launch_bounds__(256) spade_modulate_kernel(long hw, const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, const float *__restrict__ stats, float eps, float slope,
                                                             float *__restrict__ out)
{
    const long plane = blockIdx.y, base = plane * hw;
    const float *pp = stats + plane * (2 * SPADE_CHUNKS);
    const float mean = plane_mean(pp, hw);
    float var = 0.0f;
#pragma unroll
    for (int c = 0; c < SPADE_CHUNKS; ++c) var += pp[2 * c + 1];
    const float rstd = 1.0f / sqrtf(var / (float)hw + eps);
    if (VEC) {
        const long n4 = hw >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            const f32x4 xv = reinterpret_cast<const f32x4 *>(x + base)[i], gv = reinterpret_cast<const f32x4 *>(gamma + base)[i],
                        bv = reinterpret_cast<const f32x4 *>(beta + base)[i];
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = ((xv[k] - mean) * rstd) * (1.0f + gv[k]) + bv[k];
                o[k] = v > 0.0f ? v : v * slope;
            }
            reinterpret_cast<f32x4 *>(out + base)[i] = o;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
            const float v = ((x[base + i] - mean) * rstd) * (1.0f + gamma[base + i]) + beta[base + i];
            out[base + i] = v > 0.0f ? v : v * slope;
        }
    }
}

}  // namespace sahs

extern "C" long sahs_spade_stats_words(long planes) { return planes * 2 * sahs::SPADE_CHUNKS; }

extern "C" int sahs_spade_modulate_launch(long planes, long hw, const float *x, const float *gamma, const float *beta, float eps, float slope,
                                          float *out, float *stats, hipStream_t stream)
{
    if (planes <= 0 || hw <= 0) return 0;
    if (planes > 65535) return -2;      // gridDim.y
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = (hw % 4 == 0) && al(x) && al(gamma) && al(beta) && al(out);
    const dim3 sgrid(sahs::SPADE_CHUNKS, (unsigned)planes);
    for (int pass = 0; pass < 2; ++pass) {
        if (vec) sahs::instance_stats_kernel<true><<<sgrid, 256, 0, stream>>>(hw, x, stats, pass);
        else sahs::instance_stats_kernel<false><<<sgrid, 256, 0, stream>>>(hw, x, stats, pass);
    }
    const long per = vec ? hw / 4 : hw;
    long bx = (per + 255) / 256;
    const long cap = (8192 + planes - 1) / planes;      // ~8 k workgroups in all
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    const dim3 grid((unsigned)bx, (unsigned)planes);
    if (vec) sahs::spade_modulate_kernel<true><<<grid, 256, 0, stream>>>(hw, x, gamma, beta, stats, eps, slope, out);
    else sahs::spade_modulate_kernel<false><<<grid, 256, 0, stream>>>(hw, x, gamma, beta, stats, eps, slope, out);
    return (int)hipGetLastError();
}
