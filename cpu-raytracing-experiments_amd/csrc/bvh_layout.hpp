// bvh_layout.hpp — host-side conversion of the reference's BVH arrays (BVH.hpp:18-31, 85-86) into the
// GPU-internal traversal layout.  The input contract stays the reference's Node[] / Sphere[]; this is
// only how the copy in HBM/LDS is arranged.
//
// Record (64 B, one per inner node, breadth-first order so the top of the tree is one contiguous
// block that the trace kernels stage in LDS):
//     q0 = (lo0.x, lo1.x, hi0.x, hi1.x)      child 0 / child 1 boxes, grouped per axis
//     q1 = (lo0.y, lo1.y, hi0.y, hi1.y)
//     q2 = (lo0.z, lo1.z, hi0.z, hi1.z)
//     q3 = (bits c0, bits c1, 0, 0)          child reference: inner -> record index,
//                                            leaf -> kLeafBit | (prim_count-1) << 24 | first prim (BVH order)
// One 64-B fetch therefore serves both of a node's children (the reference stores them adjacently at
// first_id, first_id+1 for the same reason).
//
// Boxes are CONSERVATIVE: each leaf box is its spheres' bbox grown by pad_rel*(max|centre|+radius) and
// rounded outward, inner boxes are unions.  The growth absorbs the rounding of the slab test and of the
// reference's sphere test (BVH.hpp:251-267), so the traversal never culls a sphere the reference's
// brute-force loop (BVH.hpp:312) would accept; together with the (dist, prim index) tie rule the BVH is
// a pure acceleration of the as-shipped result (DESIGN.md "Traversal semantics").
#pragma once
#include "../../include/mirt.h"

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace mirt_host {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr float kPadRel = 0x1p-18f;

struct PadBox { float lo[3], hi[3]; };

inline uint32_t leaf_ref(uint32_t first, uint32_t count) { return kLeafBit | ((count - 1u) << 24) | first; }

// Returns "" on success, otherwise the reason the tree cannot be laid out.
inline std::string build_records(const mirt_bvh_node* nodes, uint32_t n_nodes, const mirt_sphere* prims, uint32_t n_prims,
                                 std::vector<float>& recs /* 16 floats per record */, uint32_t* max_depth_out) {
	recs.clear();
	*max_depth_out = 0;
	if (n_nodes == 0) return "";
	if (n_prims > (1u << 24)) return "more than 2^24 spheres";
	// conservative boxes, children before parents (children always have larger indices; validated by the caller)
	std::vector<PadBox> box(n_nodes);
	for (uint32_t k = n_nodes; k-- > 0;) {
		const mirt_bvh_node& nd = nodes[k];
		PadBox b{ { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
		if (nd.prim_count != 0) {
			if (nd.prim_count > 128) return "leaf with more than 128 prims";
			for (uint32_t p = nd.first_id; p < nd.first_id + nd.prim_count; p++) {
				const float* c = prims[p].position;
				const float r = std::sqrt(prims[p].radius_sq);
				float amax = std::fabs(c[0]);
				if (amax < std::fabs(c[1])) amax = std::fabs(c[1]);
				if (amax < std::fabs(c[2])) amax = std::fabs(c[2]);
				const float pad = kPadRel * (amax + r);
				for (int a = 0; a < 3; a++) {
					const float lo = std::nextafter((c[a] - r) - pad, -INFINITY), hi = std::nextafter((c[a] + r) + pad, INFINITY);
					if (lo < b.lo[a]) b.lo[a] = lo;
					if (b.hi[a] < hi) b.hi[a] = hi;
				}
			}
		} else {
			const PadBox &x = box[nd.first_id], &y = box[nd.first_id + 1];
			for (int a = 0; a < 3; a++) { b.lo[a] = (y.lo[a] < x.lo[a]) ? y.lo[a] : x.lo[a]; b.hi[a] = (x.hi[a] < y.hi[a]) ? y.hi[a] : x.hi[a]; }
		}
		box[k] = b;
	}
	auto put = [&](size_t rec, int child, const PadBox& b, uint32_t ref) {
		float* q = recs.data() + rec * 16;
		for (int a = 0; a < 3; a++) { q[a * 4 + child] = b.lo[a]; q[a * 4 + 2 + child] = b.hi[a]; }
		std::memcpy(&q[12 + child], &ref, 4);
	};
	const PadBox nothing{ { FLT_MAX, FLT_MAX, FLT_MAX }, { FLT_MAX, FLT_MAX, FLT_MAX } };   // degenerate box at +max: every slab test misses it
	if (nodes[0].prim_count != 0) {                       // single-leaf tree: one record, second child empty
		recs.assign(16, 0.0f);
		put(0, 0, box[0], leaf_ref(nodes[0].first_id, nodes[0].prim_count));
		put(0, 1, nothing, leaf_ref(nodes[0].first_id, 1));
		*max_depth_out = 1;
		return "";
	}
	// breadth-first numbering of inner nodes
	std::vector<uint32_t> order;                           // record -> node
	std::vector<uint32_t> rec_of(n_nodes, 0xffffffffu), depth(n_nodes, 0);
	order.reserve(n_nodes / 2 + 1);
	order.push_back(0); rec_of[0] = 0; depth[0] = 1;
	for (size_t head = 0; head < order.size(); head++) {
		const uint32_t nd = order[head];
		for (uint32_t c = nodes[nd].first_id; c <= nodes[nd].first_id + 1; c++) {
			depth[c] = depth[nd] + 1;
			if (depth[c] > *max_depth_out) *max_depth_out = depth[c];
			if (nodes[c].prim_count == 0) { rec_of[c] = static_cast<uint32_t>(order.size()); order.push_back(c); }
		}
	}
	if (*max_depth_out >= MIRT_BVH_STACK) return "tree deeper than the 64-entry traversal stack (BVH.hpp:321)";
	recs.assign(order.size() * 16, 0.0f);
	for (size_t r = 0; r < order.size(); r++) {
		const uint32_t nd = order[r];
		for (int child = 0; child < 2; child++) {
			const uint32_t c = nodes[nd].first_id + child;
			const uint32_t ref = nodes[c].prim_count ? leaf_ref(nodes[c].first_id, nodes[c].prim_count) : rec_of[c];
			put(r, child, box[c], ref);
		}
	}
	return "";
}

} // namespace mirt_host
