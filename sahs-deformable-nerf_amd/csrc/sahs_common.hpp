// sahs_common.hpp -- small shared helpers for the HIP sources.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "sahs_model.hpp"

// Once per (call site, device), safe from several host threads: kernel attributes and device properties belong to a DEVICE,
// and a process may drive more than one (the library keeps no other state).  The guarded action is idempotent, so two threads
// racing on the first call both perform it; nothing is cached until it has succeeded.
namespace sahs_once {
constexpr int MAX_DEVICES = 64;
struct Flags { std::atomic<int> v[MAX_DEVICES]; };
template <class F>
inline hipError_t per_device(Flags &flags, F &&action)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return action();
    if (flags.v[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = action();
    if (e == hipSuccess) flags.v[dev].store(1, std::memory_order_release);
    return e;
}
}  // namespace sahs_once

// precision ids of the A/B kernels (csrc/ab/, development builds with -DSAHS_AB_KERNELS only; reserved values of include/sahs_nerf.h)
#ifndef SAHS_BF16_2W
#define SAHS_BF16_2W 2   /* SAHS_BF16's arithmetic and packed stream through the round-1 two-waves-per-SIMD kernel (ab/field_bf16.hip) */
#define SAHS_BF16_Q 4    /* SAHS_BF16's arithmetic on v_mfma_f32_16x16x32_bf16 (ab/field_bf16q.hip) */
#endif

namespace SAHS_NS {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64;   // gfx950 wavefront
}  // namespace SAHS_NS
