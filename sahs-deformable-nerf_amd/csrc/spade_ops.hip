// spade_ops.hip -- the elementwise core of the Stage-II refiner's SPADE layer (reference
// nerf/_init_spade.py:114-160):   out = act( InstanceNorm2d(x) * (1 + gamma) + beta ),   act = LeakyReLU(slope) of the SPADEBlock that
// follows every SPADE layer (:262-279; slope 1 = none).  The reference runs this as InstanceNorm (a batch-norm kernel), an add, a multiply,
// an add and an activation -- six passes over the (N, C, H, W) tensor; here: one statistics pass (a plane is read twice, from L2 the
// second time) and ONE fused modulate pass.  HBM-bound: 4 B read for the statistics + 12 B read + 4 B written per element = 20 B
// per element (bench.py's `spade` leg reports the achieved GB/s against that).  The 3x3
// convolutions around it (label map -> 128 -> gamma / beta, and the block's spectral-normalised convolutions) are library work and
// stay on MIOpen through PyTorch (hand-written convolutions would buy nothing on this path).
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"

namespace sahs {

// one workgroup per (n, c) plane: mean, then the biased variance about that mean (two passes: the plane is L2-resident), in fp32;
// 16-byte loads when the plane allows (hw a multiple of 4 and a 16-byte aligned base)
__global__ void __launch_bounds__(256) instance_stats_kernel(long hw, const float *__restrict__ x, float eps, float *__restrict__ stats, int vec)
{
    __shared__ float red[4];
    __shared__ float s_mean;
    const float *p = x + (long)blockIdx.x * hw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.0f;
    if (vec) {
        for (long i = threadIdx.x; i < hw / 4; i += 256) { const f32x4 q = reinterpret_cast<const f32x4 *>(p)[i]; s += (q[0] + q[1]) + (q[2] + q[3]); }
    } else {
        for (long i = threadIdx.x; i < hw; i += 256) s += p[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) s_mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)hw;
    __syncthreads();
    const float mean = s_mean;
    float v = 0.0f;
    if (vec) {
        for (long i = threadIdx.x; i < hw / 4; i += 256) {
            const f32x4 q = reinterpret_cast<const f32x4 *>(p)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = q[k] - mean; v += d * d; }
        }
    } else {
        for (long i = threadIdx.x; i < hw; i += 256) { const float d = p[i] - mean; v += d * d; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float var = ((red[0] + red[1]) + (red[2] + red[3])) / (float)hw;
        stats[2 * blockIdx.x] = mean;
        stats[2 * blockIdx.x + 1] = 1.0f / sqrtf(var + eps);
    }
}

// the fused modulate pass: 2-D grid (chunk of the plane, plane) -- the plane index is blockIdx.y, no per-element divide -- and one 16-byte
// load per operand and thread where the plane allows it: 12 B read + 4 B written per element (the statistics pass reads 4 B more)
template <bool VEC>
__global__ void __launch_bounds__(256) spade_modulate_kernel(long hw, const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, const float *__restrict__ stats, float slope,
                                                             float *__restrict__ out)
{
    const long plane = blockIdx.y, base = plane * hw;
    const float mean = stats[2 * plane], rstd = stats[2 * plane + 1];
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

extern "C" int sahs_spade_modulate_launch(long planes, long hw, const float *x, const float *gamma, const float *beta, float eps, float slope,
                                          float *out, float *stats, hipStream_t stream)
{
    if (planes <= 0 || hw <= 0) return 0;
    if (planes > 65535) return -2;      // gridDim.y
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = (hw % 4 == 0) && al(x) && al(gamma) && al(beta) && al(out);
    sahs::instance_stats_kernel<<<(unsigned)planes, 256, 0, stream>>>(hw, x, eps, stats, vec ? 1 : 0);
    const long per = vec ? hw / 4 : hw;
    long bx = (per + 255) / 256;
    const long cap = (8192 + planes - 1) / planes;      // ~8 k workgroups in all
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    const dim3 grid((unsigned)bx, (unsigned)planes);
    if (vec) sahs::spade_modulate_kernel<true><<<grid, 256, 0, stream>>>(hw, x, gamma, beta, stats, slope, out);
    else sahs::spade_modulate_kernel<false><<<grid, 256, 0, stream>>>(hw, x, gamma, beta, stats, slope, out);
    return (int)hipGetLastError();
}
