// mfma_f32_kloop.hip -- what bounds a K loop of v_mfma_f32_16x16x4_f32 (the fp32 weight-gradient kernels: 128 MFMAs per wave and K-step, two
// waves per SIMD, one workgroup barrier per K-step)?  Bare loops, 256 workgroups x 512 threads, same accumulator count (32 tiles):
//   mode 0  128 independent MFMAs per iteration, operands in registers, no barrier
//   mode 1  + one s_barrier per iteration
//   mode 2  + 48 ds_read_b32 per iteration feeding the MFMAs (the kernel's operand reads), no barrier
//   mode 3  barrier + reads (the kernel's loop without its DMA)
//   mode 4  as 3 with one barrier per TWO iterations
//   mode 5  as 3 + the bias-gradient column sums (threads < 256: 16 ds_read_b32 + adds per iteration)
//   mode 6  as 3 + the operand fetch: 4 global_load_lds (16 B per lane) per wave and iteration into a ring of four stages, counted vmcnt
//   mode 7  as 6 + 5
// Prints cycles per MFMA per SIMD (32 = the pipe's rate).   hipcc --offload-arch=gfx950 -O3 mfma_f32_kloop.hip -o /tmp/kloop && /tmp/kloop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) kloop(float *out, int iters, float seed, const float *src, long src_floats)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 4 * 16 * 128];      // ring of four stages x four [16 k][128 cols] tiles
    const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, c16 = lane & 15, wave = tid >> 6;
    for (int i = tid; i < 4 * 4 * 16 * 128; i += 512) lds[i] = seed * (float)(i & 7);
    float cs = 0.f;
    const float *g = src + ((long)blockIdx.x * 8192 * 64 + wave * 1024 + lane * 4) % src_floats;
    __syncthreads();
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = seed + i + lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = seed - j + lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3 || MODE == 5 || (MODE == 4 && (it & 1) == 0)) asm volatile("s_barrier" ::: "memory");
        if (MODE >= 6) {
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            float *dst = lds + ((it + 3) & 3) * 8192 + wave * 1024;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)(dst + u * 256), 16, 0, 0);
                g += 8192;
                if (g >= src + src_floats - 8192 * 64) g -= src_floats - 8192 * 64 - 4096;
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (MODE >= 2) {
                const float *st = lds + (it & 3) * 8192 + (4 * s4 + q) * 128;
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = st[(16 * i + c16 + 64 * (wave & 1)) & 127];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = st[4096 + ((16 * j + c16) & 127)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if ((MODE == 5 || MODE == 7) && tid < 256) {
            const float *st = lds + (it & 3) * 8192 + (tid >> 7) * 2048;
#pragma unroll
            for (int k = 0; k < 16; ++k) cs += st[k * 128 + (tid & 127)];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s + cs;
}

static float *g_src; static long g_src_floats;
template <int MODE> static void run(float *out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kloop<MODE><<<256, 512>>>(out, iters, 1.0f, g_src, g_src_floats);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kloop<MODE><<<256, 512>>>(out, iters, 1.0f, g_src, g_src_floats);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = 2.0 * 128.0 * iters;      // two waves per SIMD
    printf("mode %d: %.3f ms, %.1f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz; %.1f TFLOP/s\n", MODE, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4,
           256.0 * 8 * 128.0 * iters * 2048.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    g_src_floats = 1L << 30;      // 4 GB of operands to stream
    hipMalloc(&g_src, g_src_floats * sizeof(float));
    hipMemset(g_src, 0, g_src_floats * sizeof(float));
    const int iters = 4000;
    run<0>(out, iters); run<1>(out, iters); run<2>(out, iters); run<3>(out, iters); run<4>(out, iters); run<5>(out, iters); run<6>(out, iters); run<7>(out, iters);
    return 0;
}
