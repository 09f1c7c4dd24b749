// Micro-benchmark: does VALU work issued between a wave's own MFMAs hide under them?  One wave per SIMD (256-thread workgroups, one per
// CU), two alternating accumulation chains of v_mfma_f32_32x32x16_bf16, K independent VALU fillers after every MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int MODE>      // MODE 0: v_max_f32 on VGPRs; 1: v_accvgpr_read of the OTHER chain's accumulator; 2: v_cvt_pk_bf16_f32
__global__ void __launch_bounds__(256, 1) kern(const bf16x8 *__restrict__ g, float *out, int iters, unsigned long long *cyc)
{
    bf16x8 a = g[threadIdx.x], b = g[threadIdx.x + 256];
    f32x16 acc0 = {}, acc1 = {};
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)threadIdx.x * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k % 8]) : "v"(f[(k + 1) % 8]));
                else if (MODE == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[k % 8]) : "a"(((u & 1) ? acc0 : acc1)[k % 16]));
                else { unsigned w; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(f[k % 8]), "v"(f[(k + 1) % 8])); asm volatile("" :: "v"(w)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3] + s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int MODE>
void run(const bf16x8 *g, float *out, unsigned long long *cyc, int iters)
{
    kern<K, MODE><<<256, 256>>>(g, out, 4, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<K, MODE><<<256, 256>>>(g, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    unsigned long long med = h[128];
    const double mf = (double)iters * 8;
    printf("mode %d fillers/MFMA %d: %.2f ms, %.1f cycles/MFMA (s_memtime), %.0f TFLOP/s\n", MODE, K, ms, (double)med / mf,
           mf * 256 * 4 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    bf16x8 *g; float *out; unsigned long long *cyc;
    hipMalloc(&g, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<unsigned short> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 22);   // random-ish bf16 around 0.01
    hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    run<0, 0>(g, out, cyc, iters); run<2, 0>(g, out, cyc, iters); run<4, 0>(g, out, cyc, iters); run<6, 0>(g, out, cyc, iters); run<8, 0>(g, out, cyc, iters);
    run<2, 1>(g, out, cyc, iters); run<4, 1>(g, out, cyc, iters); run<2, 2>(g, out, cyc, iters); run<4, 2>(g, out, cyc, iters);
    return 0;
}
