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
//                                            leaf -> kLeafBit | prim (BVH order); every leaf of the layout holds ONE prim
//                                            (a caller's leaf of k prims becomes a small subtree of k leaves, split_multi_prim_leaves)
// One 64-B fetch therefore serves both of a node's children (the reference stores them adjacently at
// first_id, first_id+1 for the same reason).
//
// DEFAULT LAYOUT: 4-wide binary16 records (64 B: a node and its inner children, build_wide_half_records below) whenever binary16
// is precise enough and the tree is not too deep (from the host builder here or from lbvh_build.hip); the child-pair records
// described here remain for deep trees and f32 planes.
//
// Half-precision variant (32 B per record, used when it is precise enough for the scene): the same twelve planes as
// IEEE binary16, lo planes rounded toward -inf and hi planes toward +inf so the boxes only ever grow:
//     words 0..5 = (lo0|lo1<<16, hi0|hi1<<16) for x, y, z;  words 6,7 = child references.
// Halving the record halves the LDS bytes and read instructions per traversal step and lets two 16-wave workgroups
// share a CU's LDS.  The slab arithmetic itself stays f32.
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

// ---- IEEE binary16 with directed rounding (host side; the device decodes with v_cvt_f32_f16) -------------------------
inline float half_to_float(uint16_t h) {
	const uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
	uint32_t bits;
	if (exp == 0) {
		if (man == 0) bits = sign;
		else { int e = -1; uint32_t m = man; do { e++; m <<= 1; } while (!(m & 0x400u)); bits = sign | ((127 - 15 - e) << 23) | ((m & 0x3ffu) << 13); }
	} else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
	else bits = sign | ((exp - 15 + 127) << 23) | (man << 13);
	float f; std::memcpy(&f, &bits, 4); return f;
}
inline uint16_t half_next_up(uint16_t h) {                 // smallest half greater than h (h finite or -inf)
	if ((h & 0x7fffu) == 0) return 0x0001u;                // +-0 -> smallest positive subnormal
	return (h & 0x8000u) ? static_cast<uint16_t>(h - 1) : static_cast<uint16_t>(h + 1);
}
inline uint16_t half_next_down(uint16_t h) {
	if ((h & 0x7fffu) == 0) return 0x8001u;
	return (h & 0x8000u) ? static_cast<uint16_t>(h + 1) : static_cast<uint16_t>(h - 1);
}
inline uint16_t float_to_half_nearest(float f) {           // round-to-nearest-even, overflow -> inf
	uint32_t x; std::memcpy(&x, &f, 4);
	const uint32_t sign = (x >> 16) & 0x8000u;
	x &= 0x7fffffffu;
	if (x >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));
	if (x >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);                   // >= 65520 rounds to inf
	if (x < 0x33000001u) return static_cast<uint16_t>(sign);                               // < 2^-25 rounds to 0
	const int e = static_cast<int>(x >> 23) - 127;
	uint32_t man = (x & 0x7fffffu) | 0x800000u;
	int shift = (e < -14) ? (13 + (-14 - e)) : 13;
	uint32_t h = man >> shift;
	const uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
	if (rem > halfway || (rem == halfway && (h & 1u))) h++;
	if (e < -14) return static_cast<uint16_t>(sign | h);                                   // subnormal (carry into exp 1 is correct)
	h = (h & 0x3ffu) + ((h & 0x400u) ? 0u : 0u) + (static_cast<uint32_t>(e + 15) << 10) + ((h >> 11) << 10);   // h has the implicit 1 at bit 10; a carry-out bumps the exponent
	return static_cast<uint16_t>(sign | h);
}
inline uint16_t float_to_half_down(float f) {              // largest half <= f
	uint16_t h = float_to_half_nearest(f);
	if (half_to_float(h) > f) h = half_next_down(h);
	return h;
}
inline uint16_t float_to_half_up(float f) {                // smallest half >= f
	uint16_t h = float_to_half_nearest(f);
	if (half_to_float(h) < f) h = half_next_up(h);
	return h;
}
inline float half_ulp_at(float magnitude) {                // spacing of binary16 values around |magnitude|
	if (magnitude < 6.103515625e-05f) return 5.9604644775390625e-08f;
	int e; std::frexp(magnitude, &e);                       // magnitude = m * 2^e, m in [0.5, 1)
	return std::ldexp(1.0f, e - 11);
}

inline uint32_t leaf_ref(uint32_t prim) { return kLeafBit | prim; }

// The reference's builder only makes one-prim leaves (BVH.hpp:201-205) but its traversal loops over prim_count
// (BVH.hpp:343-345), so a caller's tree may hold larger leaves.  The kernels only know one-prim leaves: a leaf of k prims
// [first, first+k) is replaced by a median-split subtree over the same range (children appended at the end of the array,
// so child indices stay larger than their parent's).  Boxes are recomputed from the prims by build_records anyway, and the
// traversal result does not depend on the tree (DESIGN.md "Traversal semantics").
inline void split_multi_prim_leaves(std::vector<mirt_bvh_node>& nodes) {
	for (size_t i = 0; i < nodes.size(); i++) {                 // nodes appended below are visited as well
		const uint32_t k = nodes[i].prim_count, first = nodes[i].first_id;
		if (k <= 1) continue;
		const uint32_t child = static_cast<uint32_t>(nodes.size()), half = k / 2;
		mirt_bvh_node a = nodes[i], b = nodes[i];
		a.first_id = first; a.prim_count = half;
		b.first_id = first + half; b.prim_count = k - half;
		nodes.push_back(a); nodes.push_back(b);
		nodes[i].first_id = child; nodes[i].prim_count = 0;
	}
}

// 32-B half-precision records from the 64-B ones.  Returns false (and leaves `out` empty) when binary16 is not adequate:
// a coordinate beyond +-60000, or a leaf box whose smallest extent is under 8 quantisation steps (the box would grow by
// more than ~25 %).  (With more than 32768 records or spheres the traversal stack keeps u32 entries, see mirt_capi.hip.)
inline bool build_half_records(const std::vector<float>& recs, std::vector<uint32_t>& out) {
	out.clear();
	const size_t n = recs.size() / 16;
	if (n == 0) return false;
	for (size_t r = 0; r < n; r++) {
		const float* q = recs.data() + r * 16;
		for (int child = 0; child < 2; child++) {
			uint32_t ref; std::memcpy(&ref, &q[12 + child], 4);
			float amax = 0.0f, min_extent = INFINITY;
			for (int a = 0; a < 3; a++) {
				const float lo = q[a * 4 + child], hi = q[a * 4 + 2 + child];
				if (lo == FLT_MAX && hi == FLT_MAX) { min_extent = INFINITY; amax = 0.0f; break; }     // the empty child of a single-leaf tree
				amax = std::fmax(amax, std::fmax(std::fabs(lo), std::fabs(hi)));
				min_extent = std::fmin(min_extent, hi - lo);
			}
			if (amax > 60000.0f) return false;
			if ((ref & kLeafBit) && min_extent < 8.0f * half_ulp_at(amax)) return false;
		}
	}
	out.resize(n * 8);
	for (size_t r = 0; r < n; r++) {
		const float* q = recs.data() + r * 16;
		uint32_t* w = out.data() + r * 8;
		for (int a = 0; a < 3; a++) {
			uint16_t lo[2], hi[2];
			for (int child = 0; child < 2; child++) {
				const float l = q[a * 4 + child], h = q[a * 4 + 2 + child];
				if (l == FLT_MAX && h == FLT_MAX) { lo[child] = hi[child] = 0x7c00u; continue; }       // +inf box: every slab test misses
				lo[child] = float_to_half_down(l); hi[child] = float_to_half_up(h);
			}
			w[a * 2] = lo[0] | (static_cast<uint32_t>(lo[1]) << 16);
			w[a * 2 + 1] = hi[0] | (static_cast<uint32_t>(hi[1]) << 16);
		}
		std::memcpy(&w[6], &q[12], 8);
	}
	return true;
}

// 4-WIDE binary16 records (64 B) from the binary 64-B ones: every inner node absorbs its inner children, so a wide node holds
// up to four children — the grandchildren of the binary node it stands for, or a child itself where that child is a leaf:
//     q0 = x planes, as binary16 pairs: (lo k0 | lo k1), (lo k2 | lo k3), (hi k0 | hi k1), (hi k2 | hi k3)
//     q1 = y planes, q2 = z planes likewise
//     q3 = the four child references (wide record index, or kLeafBit | prim); slots in the order [children of c0 ..., children of c1 ...]
// Unused slots carry the box (+inf, +inf) that every slab test misses and repeat the first slot's reference (never followed).
// Wide records are numbered breadth-first.  One 64-B fetch then serves two levels of the binary tree: on S(100000) a closest-hit
// ray makes 10.9 instead of 19.3 dependent record fetches and a shadow ray 15.5 instead of 30.7, with as many box tests as before
// (40.9 vs 38.5 / 58.5 vs 61.4; oracle twin, orc_wide_stats).  Returns false when binary16 is not adequate (build_half_records'
// rules) or the tree is too deep for the traversal stack: a wide node can leave three entries behind, so 3 x (wide levels) must
// stay below MIRT_BVH_STACK.
inline bool build_wide_half_records(const std::vector<float>& recs, std::vector<uint32_t>& out, uint32_t* wide_levels_out) {
	std::vector<uint32_t> probe;
	out.clear();
	if (!build_half_records(recs, probe)) return false;                 // adequacy only; the binary half records are not kept
	const size_t n = recs.size() / 16;
	struct Kid { float lo[3], hi[3]; uint32_t ref; };
	auto kid_of = [&](size_t rec, int child) {
		Kid k; const float* q = recs.data() + rec * 16;
		for (int a = 0; a < 3; a++) { k.lo[a] = q[a * 4 + child]; k.hi[a] = q[a * 4 + 2 + child]; }
		std::memcpy(&k.ref, &q[12 + child], 4);
		return k;
	};
	std::vector<uint32_t> wide_of(n, 0xffffffffu), level(n, 0), order;
	order.reserve(n / 2 + 1);
	order.push_back(0); wide_of[0] = 0; level[0] = 1;
	uint32_t levels = 1;
	std::vector<Kid> kids; kids.reserve(4 * (n / 2 + 1));
	std::vector<uint8_t> n_kids; n_kids.reserve(n / 2 + 1);
	for (size_t head = 0; head < order.size(); head++) {
		const uint32_t r = order[head];
		uint8_t nk = 0;
		for (int child = 0; child < 2; child++) {
			const Kid c = kid_of(r, child);
			if (c.ref & kLeafBit) { kids.push_back(c); nk++; }
			else { kids.push_back(kid_of(c.ref, 0)); kids.push_back(kid_of(c.ref, 1)); nk += 2; }
		}
		n_kids.push_back(nk);
		for (size_t k = kids.size() - nk; k < kids.size(); k++) {
			const uint32_t b = kids[k].ref;
			if (b & kLeafBit) continue;
			if (b >= n || wide_of[b] != 0xffffffffu) return false;      // (validated trees never get here)
			wide_of[b] = static_cast<uint32_t>(order.size()); level[b] = level[r] + 1; order.push_back(b);
			if (level[b] > levels) levels = level[b];
		}
	}
	if (3u * levels >= MIRT_BVH_STACK) return false;
	*wide_levels_out = levels;
	out.assign(order.size() * 16, 0u);
	size_t at = 0;
	for (size_t w = 0; w < order.size(); w++) {
		uint32_t* q = out.data() + w * 16;
		uint16_t lo[3][4], hi[3][4]; uint32_t ref[4];
		const uint8_t nk = n_kids[w];
		for (int k = 0; k < 4; k++) {
			if (k < nk) {
				const Kid& c = kids[at + k];
				const bool nothing = c.lo[0] == FLT_MAX && c.hi[0] == FLT_MAX;         // the empty child of a single-leaf tree
				for (int a = 0; a < 3; a++) { lo[a][k] = nothing ? 0x7c00u : float_to_half_down(c.lo[a]); hi[a][k] = nothing ? 0x7c00u : float_to_half_up(c.hi[a]); }
				ref[k] = (c.ref & kLeafBit) ? c.ref : wide_of[c.ref];
			} else {
				for (int a = 0; a < 3; a++) lo[a][k] = hi[a][k] = 0x7c00u;              // +inf box: every slab test misses
				ref[k] = ref[0];
			}
		}
		at += nk;
		for (int a = 0; a < 3; a++) {
			q[a * 4 + 0] = lo[a][0] | (static_cast<uint32_t>(lo[a][1]) << 16);
			q[a * 4 + 1] = lo[a][2] | (static_cast<uint32_t>(lo[a][3]) << 16);
			q[a * 4 + 2] = hi[a][0] | (static_cast<uint32_t>(hi[a][1]) << 16);
			q[a * 4 + 3] = hi[a][2] | (static_cast<uint32_t>(hi[a][3]) << 16);
		}
		for (int k = 0; k < 4; k++) q[12 + k] = ref[k];
	}
	return true;
}

// Lays out 64-B records for `nodes`; returns "" on success, otherwise the reason the tree is rejected.
// `prim_of_slot` (optional): leaf slot s of `nodes` refers to prims[prim_of_slot[s]] (internal tree); without it slot s is
// prims[s] (the caller's tree, BVH.hpp:201-205).  Leaf references always carry the index into `prims` (the BVH-order
// array hit.primID refers to); mapped leaves must hold a single prim.
// Records are numbered breadth-first, so the top of the tree is one contiguous block (what the trace kernels stage in LDS).
// (A treelet order for the records below that block — four likely-co-visited inner nodes per 128-B line, treelets laid out
// depth-first — was measured on S(100000) and changed nothing: k_trace 399.9 ms/step breadth-first vs 405.0 with treelets; the
// kernel is bound by VALU issue, not by where its L2 hits land.  Removed.)
inline std::string build_records(const mirt_bvh_node* nodes, uint32_t n_nodes, const mirt_sphere* prims, uint32_t n_prims,
                                 std::vector<float>& recs /* 16 floats per record */, uint32_t* max_depth_out,
                                 const std::vector<uint32_t>* prim_of_slot = nullptr) {
	recs.clear();
	*max_depth_out = 0;
	if (n_nodes == 0) return "";
	if (n_prims >= (1u << 26)) return "more than 2^26 spheres";       // record byte offsets are 32-bit in the trace kernels, leaf references carry a flag bit
	// conservative boxes, children before parents (children always have larger indices; validated by the caller)
	std::vector<PadBox> box(n_nodes);
	for (uint32_t k = n_nodes; k-- > 0;) {
		const mirt_bvh_node& nd = nodes[k];
		PadBox b{ { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
		if (nd.prim_count != 0) {
			if (nd.prim_count != 1) return "leaf with more than one prim (split_multi_prim_leaves first)";
			for (uint32_t slot = nd.first_id; slot < nd.first_id + nd.prim_count; slot++) {
				const uint32_t p = prim_of_slot ? (*prim_of_slot)[slot] : slot;
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
	auto leaf_of = [&](const mirt_bvh_node& nd) { return leaf_ref(prim_of_slot ? (*prim_of_slot)[nd.first_id] : nd.first_id); };
	if (nodes[0].prim_count != 0) {                       // single-leaf tree: one record, second child empty
		recs.assign(16, 0.0f);
		put(0, 0, box[0], leaf_of(nodes[0]));
		put(0, 1, nothing, leaf_of(nodes[0]));
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
			if (nodes[c].prim_count == 0) {
				if (rec_of[c] != 0xffffffffu || order.size() >= n_nodes) return "node referenced by more than one parent";   // also bounds this loop
				rec_of[c] = static_cast<uint32_t>(order.size()); order.push_back(c);
			}
		}
	}
	if (*max_depth_out >= MIRT_BVH_STACK) return "tree deeper than the 64-entry traversal stack (BVH.hpp:321)";
	recs.assign(order.size() * 16, 0.0f);
	for (size_t r = 0; r < order.size(); r++) {
		const uint32_t nd = order[r];
		for (int child = 0; child < 2; child++) {
			const uint32_t c = nodes[nd].first_id + child;
			const uint32_t ref = nodes[c].prim_count ? leaf_of(nodes[c]) : rec_of[c];
			put(r, child, box[c], ref);
		}
	}
	return "";
}

} // namespace mirt_host
