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

namespace SAHS_NS {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64;   // gfx950 wavefront
}  // namespace SAHS_NS
