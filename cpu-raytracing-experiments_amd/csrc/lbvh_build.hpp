// lbvh_build.hpp — GPU construction of the internal traversal tree (see lbvh_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace mirt_gpu {
// Builds an LBVH over `spheres` ({pos.xyz, radius_sq}, BVH order, n >= 2, device memory) on stream `st` and writes its n-1
// child-pair records in the layouts of bvh_layout.hpp: 64-B f32 records to `recs32` and/or 32-B binary16 records to
// `recs16` (either may be null; device memory).  `*depth_out` = levels of the tree including the leaf level.
// `recs_wide` (optional, room for n-1 records of 64 B): the same tree as 4-wide binary16 records (bvh_layout.hpp
// build_wide_half_records: a node and its inner children), `*n_wide_out` of them, breadth-first.
// Returns false with *err set if a HIP call failed.  Synchronises `st` before returning.
bool build_lbvh(hipStream_t st, const float4* spheres, uint32_t n, float* recs32, uint32_t* recs16, uint32_t* depth_out, std::string* err,
                uint32_t* recs_wide = nullptr, uint32_t* n_wide_out = nullptr);
}
