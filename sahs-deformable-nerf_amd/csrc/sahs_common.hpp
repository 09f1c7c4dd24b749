// sahs_common.hpp -- small shared helpers for the HIP sources.
#pragma once
#include <hip/hip_runtime.h>
#include "sahs_model.hpp"

namespace SAHS_NS {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64;   // gfx950 wavefront
}  // namespace SAHS_NS
