// Micro-benchmark 2: what the activation conversion costs beside a wave's own MFMAs (one wave per SIMD), by where the data lives.
// Two alternating chains of v_mfma_f32_32x32x16_bf16; after every MFMA, K "conversion units" = {fetch one value, v_mul, v_max} and a
// v_cvt_pk every second unit -- the real kernel's work per accumulator value.  SRC 0: the value is in an idle arch VGPR; SRC 1: in an
// idle AGPR block (v_accvgpr_read first).  Build twice: default (MFMA accumulators in AGPRs) and -mllvm -amdgpu-mfma-vgpr-form=1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int SRC>
__global__ void __launch_bounds__(256, 1) kern(const bf16x8 *__restrict__ g, float *out, int iters)
{
    bf16x8 a = g[threadIdx.x], b = g[threadIdx.x + 256];
    f32x16 acc0 = {}, acc1 = {};
    float idle[16], res = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) idle[i] = (float)threadIdx.x * 0.001f + i;
    float ag[16];
    if (SRC == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[i]) : "v"(idle[i]));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float prev = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float v, m;
                if (SRC == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(ag[(2 * u + k) % 16]));
                else { v = idle[(2 * u + k) % 16]; asm volatile("" : "+v"(v)); }
                asm volatile("v_mul_f32 %0, 0x3c23d70a, %1" : "=v"(m) : "v"(v));
                asm volatile("v_max_f32 %0, %1, %0" : "+v"(m) : "v"(v));
                if (k & 1) { unsigned w; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(prev), "v"(m)); asm volatile("" :: "v"(w)); }
                prev = m;
            }
            asm volatile("" :: "v"(prev));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3] + res;
}

template <int K, int SRC>
void run(const bf16x8 *g, float *out, int iters)
{
    kern<K, SRC><<<256, 256>>>(g, out, 4);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    kern<K, SRC><<<256, 256>>>(g, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 8;
    printf("value source %s, units/MFMA %d: %.2f ms, %.0f TFLOP/s\n", SRC ? "idle AGPR" : "idle VGPR", K, ms, mf * 256 * 4 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    bf16x8 *g; float *out;
    (void)hipMalloc(&g, 512 * 16); (void)hipMalloc(&out, 256 * 256 * 4);
    std::vector<unsigned short> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 22);
    (void)hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;
    run<0, 0>(g, out, iters); run<1, 0>(g, out, iters); run<2, 0>(g, out, iters); run<4, 0>(g, out, iters);
    run<1, 1>(g, out, iters); run<2, 1>(g, out, iters); run<4, 1>(g, out, iters);
    return 0;
}
