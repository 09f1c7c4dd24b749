// Micro-benchmark 3: which bf16 MFMA shape sustains more FLOP/s on THIS chip (random operands held in registers, one wave per SIMD,
// every CU busy)?  v_mfma_f32_32x32x16_bf16 (what field_bf16w.hip uses) against
// v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md: ~1.15x the sustained clock).  Same FLOPs per loop iteration in both kernels.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_clock mfma_shape_clock.hip && ./mfma_shape_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 32x32x16: 8 MFMAs per iteration on 2 accumulators (2 x 16 regs)
__global__ void __launch_bounds__(256, 1) k32(const bf16x8 *__restrict__ g, float *out, int iters, unsigned long long *cyc, unsigned long long *rt)
{
    __shared__ bf16x8 lds[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 256) lds[i] = g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 b0 = g[512 + threadIdx.x], b1 = g[768 + threadIdx.x];
    bf16x8 av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) av[u] = lds[u * 64 + lane];
    f32x16 acc0 = {}, acc1 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bf16x8 a = av[u];
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3];
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

// 16x16x32: 16 MFMAs per iteration (same FLOPs: 16 x 16384 = 8 x 32768) on 16 accumulators (16 x 4 regs)
__global__ void __launch_bounds__(256, 1) k16(const bf16x8 *__restrict__ g, float *out, int iters, unsigned long long *cyc, unsigned long long *rt)
{
    __shared__ bf16x8 lds[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 256) lds[i] = g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 b0 = g[512 + threadIdx.x], b1 = g[768 + threadIdx.x];
    bf16x8 av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) av[u] = lds[u * 64 + lane];
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bf16x8 a = av[u];
            acc[2 * u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[2 * u], 0, 0, 0);
            acc[2 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[2 * u + 1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <class K>
void run(const char *name, K kern, const bf16x8 *g, float *out, unsigned long long *cyc, unsigned long long *rt, int iters)
{
    for (int w = 0; w < 3; ++w) kern<<<256, 256>>>(g, out, iters, cyc, rt);      // ~1.5 s of back-to-back load before the timed launch
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<<<256, 256>>>(g, out, iters, cyc, rt);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(256), hr(256);
    hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), rt, 256 * 8, hipMemcpyDeviceToHost);
    const double flop = (double)iters * 8 * 32768.0 * 256 * 4;
    printf("%s: %.1f ms, %.0f TFLOP/s, %.2f cycles per 32768 FLOP, in-kernel clock %.2f GHz\n", name, ms, flop / (ms * 1e-3) / 1e12,
           (double)hc[128] / ((double)iters * 8), (double)hc[128] / ((double)hr[128] * 10.0) );
}

int main()
{
    bf16x8 *g; float *out; unsigned long long *cyc, *rt;
    hipMalloc(&g, 1024 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&rt, 256 * 8);
    std::vector<unsigned short> h(1024 * 8);
    unsigned s = 12345;
    for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 3) & 0x8000)); }   // random signs and mantissas around 0.01
    hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 400000;
    run("32x32x16", k32, g, out, cyc, rt, iters);
    run("16x16x32", k16, g, out, cyc, rt, iters);
    run("32x32x16", k32, g, out, cyc, rt, iters);
    run("16x16x32", k16, g, out, cyc, rt, iters);
    return 0;
}
