// spade_ops.hip -- the elementwise core of the Stage-II refiner's SPADE layer (reference
// nerf/_init_spade.py:114-160):   out = act( InstanceNorm2d(x) * (1 + gamma) + beta ),   act = LeakyReLU(slope) of the SPADEBlock that
// follows every SPADE layer (:262-279; slope 1 = none).  The reference runs this as InstanceNorm (a batch-norm kernel), an add, a multiply,
// an add and an activation -- six passes over the (N, C, H, W) tensor; here: one statistics pass (a plane is read twice, from L2 the
// second time) and ONE fused modulate pass.  HBM-bound: 4 B read for the statistics + 12 B read + 4 B written per element.  The 3x3
// convolutions around it (label map -> 128 -> gamma / beta, and the block's spectral-normalised convolutions) are library work and
// stay on MIOpen through PyTorch (hand-written convolutions would buy nothing on this path).
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"

namespace sahs {

// one workgroup per (n, c) plane: mean, then the biased variance about that mean (two passes: the plane is L2-resident), in fp32
__global__ void __launch_bounds__(256) instance_stats_kernel(long hw, const float *__restrict__ x, float eps, float *__restrict__ stats)
{
    __shared__ float red[4];
    __shared__ float s_mean;
    const float *p = x + (long)blockIdx.x * hw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.0f;
    for (long i = threadIdx.x; i < hw; i += 256) s += p[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) s_mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)hw;
    __syncthreads();
    const float mean = s_mean;
    float v = 0.0f;
    for (long i = threadIdx.x; i < hw; i += 256) { const float d = p[i] - mean; v += d * d; }
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

__global__ void __launch_bounds__(256) spade_modulate_kernel(long total, long hw, const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, const float *__restrict__ stats, float slope,
                                                             float *__restrict__ out)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long plane = e / hw;
        const float nrm = (x[e] - stats[2 * plane]) * stats[2 * plane + 1];
        const float v = nrm * (1.0f + gamma[e]) + beta[e];
        out[e] = v > 0.0f ? v : v * slope;
    }
}

}  // namespace sahs

extern "C" int sahs_spade_modulate_launch(long planes, long hw, const float *x, const float *gamma, const float *beta, float eps, float slope,
                                          float *out, float *stats, hipStream_t stream)
{
    if (planes <= 0 || hw <= 0) return 0;
    sahs::instance_stats_kernel<<<(unsigned)planes, 256, 0, stream>>>(hw, x, eps, stats);
    const long total = planes * hw;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    sahs::spade_modulate_kernel<<<(unsigned)blocks, 256, 0, stream>>>(total, hw, x, gamma, beta, stats, slope, out);
    return (int)hipGetLastError();
}
