// bvh_build.hpp — internal entry point of csrc/bvh_build.cpp (the public ones are mirt_bvh_build / mirt_light_list in mirt.h).
#pragma once
#include "../../include/mirt.h"
#include <cstdint>
#include <vector>

namespace mirt_host {
// SAH sweep tree (true half-area cost, one prim per leaf) over `prims`; leaf node k with first_id = s refers to
// prims[prim_of_slot[s]].  Used for the GPU-internal traversal structure only.
void build_sah_tree(const mirt_sphere* prims, uint32_t n, std::vector<mirt_bvh_node>& nodes, std::vector<uint32_t>& prim_of_slot);
}
