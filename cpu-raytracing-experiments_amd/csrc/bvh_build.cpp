// bvh_build.cpp — host-side SAH sweep builder producing the reference's node / prim layout.
//
// Stands in for BoundingVolumeHierarchy<Sphere>::BoundingVolumeHierarchy (BVH.hpp:90-206), which the
// reference runs on the host at start-up and after every geometry edit (Application.cpp:233,508).  The
// GPU traversal consumes exactly these arrays, and NEE compares geometry-order light ids with BVH-order
// hit ids (Renderer.hpp:261-263), so the tree must be the reference's tree, not just a valid one:
//   * cost uses Node::half_area() as written, which sums only d.y*d.z          (BVH.hpp:58-67)
//   * the "chunked" right sweep degenerates to a full suffix sweep and its early-out never fires, leaving
//     a plain SAH sweep with the single left-side break                        (BVH.hpp:146-171)
//   * initial candidate = median split on the widest axis at cost half_area*(count-1)   (BVH.hpp:144)
//   * child with the larger half-area is stored first, smaller range is refined first    (BVH.hpp:190-197)
//   * leaves hold one prim; prims are emitted in axis-0 order                   (BVH.hpp:133,201-205)
// The reference sorts centroids with the unstable std::ranges::sort (BVH.hpp:121); ties are broken by
// primitive index here so the result is defined.
#include "../../include/mirt.h"
#include "bvh_build.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

struct Bounds {
	float lo[3], hi[3];
	static Bounds nothing() { return { { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } }; }
	void include(const Bounds& o) {
		for (int a = 0; a < 3; a++) {
			lo[a] = (o.lo[a] < lo[a]) ? o.lo[a] : lo[a];     // glm::min(lo, o.lo)
			hi[a] = (hi[a] < o.hi[a]) ? o.hi[a] : hi[a];     // glm::max(hi, o.hi)
		}
	}
	float extent(int a) const { return hi[a] - lo[a]; }
	float sah_area(bool full) const {
		if (full) return (extent(0) * extent(1) + extent(1) * extent(2)) + extent(2) * extent(0);   // true half surface area (internal tree)
		float area = 0.0f;                                    // Node::half_area() as written: 0 + d.y * d.z (Q15)
		area += extent(1) * extent(2);
		return area;
	}
	int widest_axis() const {
		int best = 0;
		for (int a = 1; a < 3; a++) if (extent(best) < extent(a)) best = a;
		return best;
	}
};

struct Candidate { size_t position; int axis; float cost; };
struct Pending { uint32_t node; size_t first, count; };

class SweepBuilder {
public:
	SweepBuilder(const mirt_sphere* spheres, uint32_t n, bool full_area = false) : src_(spheres), n_(n), full_area_(full_area), box_(n), suffix_cost_(n), left_side_(n) {
		for (int a = 0; a < 3; a++) order_[a].resize(n);
		std::vector<float> centre(static_cast<size_t>(n) * 3);
		for (uint32_t i = 0; i < n; i++) {
			const float r = std::sqrt(spheres[i].radius_sq);                       // Sphere::bounds(), Primitives.hpp:13-16
			for (int a = 0; a < 3; a++) {
				box_[i].lo[a] = spheres[i].position[a] - r;
				box_[i].hi[a] = spheres[i].position[a] + r;
				centre[static_cast<size_t>(i) * 3 + a] = (box_[i].hi[a] + box_[i].lo[a]) * 0.5f;   // Node::centroid()
			}
		}
		for (int a = 0; a < 3; a++) {
			std::iota(order_[a].begin(), order_[a].end(), 0u);
			std::stable_sort(order_[a].begin(), order_[a].end(), [&](uint32_t l, uint32_t r) {
				return centre[static_cast<size_t>(l) * 3 + a] < centre[static_cast<size_t>(r) * 3 + a];
			});
		}
	}

	// prims_out (optional): primitives in leaf order; order_out (optional): source index of the primitive at each leaf slot
	void run(std::vector<mirt_bvh_node>& nodes, mirt_sphere* prims_out, std::vector<uint32_t>* order_out = nullptr) {
		nodes.clear();
		bounds_.clear();
		if (n_ == 0) return;
		nodes.reserve(2 * static_cast<size_t>(n_));
		bounds_.reserve(2 * static_cast<size_t>(n_));
		emit(nodes, range_bounds(0, n_));
		std::vector<Pending> todo{ Pending{ 0, 0, n_ } };
		while (!todo.empty()) {
			const Pending job = todo.back();
			todo.pop_back();
			if (job.count <= 1) {                                                   // leaf
				nodes[job.node].first_id = static_cast<uint32_t>(job.first);
				nodes[job.node].prim_count = static_cast<uint32_t>(job.count);
				continue;
			}
			const uint32_t kids = static_cast<uint32_t>(nodes.size());
			nodes[job.node].first_id = kids;
			const size_t first = job.first, last = job.first + job.count;
			const Candidate split = choose_split(bounds_[job.node], first, last);
			partition_other_axes(split, first, last);

			const size_t lo_n = split.position - first, hi_n = last - split.position;
			const Bounds lo_box = range_bounds(first, split.position), hi_box = range_bounds(split.position, last);
			const bool hi_is_bigger_box = lo_box.sah_area(full_area_) < hi_box.sah_area(full_area_);
			const bool hi_is_bigger_range = lo_n < hi_n;
			// slot 0 gets the child with the larger (y*z) area
			emit(nodes, hi_is_bigger_box ? hi_box : lo_box);
			emit(nodes, hi_is_bigger_box ? lo_box : hi_box);
			const uint32_t lo_node = kids + (hi_is_bigger_box ? 1u : 0u), hi_node = kids + (hi_is_bigger_box ? 0u : 1u);
			const Pending lo_job{ lo_node, first, lo_n }, hi_job{ hi_node, split.position, hi_n };
			// the larger range is queued first so the smaller one is refined next (bounded stack depth)
			todo.push_back(hi_is_bigger_range ? hi_job : lo_job);
			todo.push_back(hi_is_bigger_range ? lo_job : hi_job);
		}
		if (prims_out) for (uint32_t i = 0; i < n_; i++) prims_out[i] = src_[order_[0][i]];
		if (order_out) *order_out = order_[0];
	}

private:
	Bounds range_bounds(size_t first, size_t last) const {
		Bounds b = Bounds::nothing();
		for (size_t i = first; i < last; i++) b.include(box_[order_[0][i]]);
		return b;
	}
	void emit(std::vector<mirt_bvh_node>& nodes, const Bounds& b) {
		mirt_bvh_node n;
		std::memset(&n, 0, sizeof n);
		for (int a = 0; a < 3; a++) { n.min_bound[a] = b.lo[a]; n.max_bound[a] = b.hi[a]; }
		nodes.push_back(n);
		bounds_.push_back(b);
	}
	Candidate choose_split(const Bounds& parent, size_t first, size_t last) {
		const size_t count = last - first;
		Candidate best{ first + (count + 1) / 2, parent.widest_axis(), parent.sah_area(full_area_) * (static_cast<float>(count) - 1.0f) };
		for (int axis = 0; axis < 3; axis++) {
			const std::vector<uint32_t>& ids = order_[axis];
			Bounds right = Bounds::nothing();
			for (size_t i = last - 1; i > first; i--) {                              // suffix costs for splits at i
				right.include(box_[ids[i]]);
				suffix_cost_[i] = right.sah_area(full_area_) * static_cast<float>(last - i);
			}
			Bounds left = Bounds::nothing();
			for (size_t i = first; i + 1 < last; i++) {
				left.include(box_[ids[i]]);
				const float left_cost = left.sah_area(full_area_) * static_cast<float>(i + 1 - first);
				if (left_cost > best.cost) break;
				const float total = left_cost + suffix_cost_[i + 1];
				if (total < best.cost) best = Candidate{ i + 1, axis, total };
			}
		}
		return best;
	}
	void partition_other_axes(const Candidate& split, size_t first, size_t last) {
		const std::vector<uint32_t>& ids = order_[split.axis];
		for (size_t i = first; i < split.position; i++) left_side_[ids[i]] = 1;
		for (size_t i = split.position; i < last; i++) left_side_[ids[i]] = 0;
		for (int axis = 0; axis < 3; axis++) {
			if (axis == split.axis) continue;
			std::stable_partition(order_[axis].begin() + first, order_[axis].begin() + last,
			                      [&](uint32_t id) { return left_side_[id] != 0; });
		}
	}

	const mirt_sphere* src_;
	uint32_t n_;
	bool full_area_;
	std::vector<Bounds> box_;
	std::vector<Bounds> bounds_;          // per emitted node
	std::vector<float> suffix_cost_;
	std::vector<uint8_t> left_side_;
	std::vector<uint32_t> order_[3];
};

} // namespace

// Internal traversal tree: the same sweep builder with the true half surface area as cost (the reference's formula drops
// the x extent, Q15, which roughly doubles the boxes a ray has to test).  Only the GPU-internal copy uses it; the
// traversal result does not depend on the tree (DESIGN.md "Traversal semantics").
void mirt_host::build_sah_tree(const mirt_sphere* prims, uint32_t n, std::vector<mirt_bvh_node>& nodes, std::vector<uint32_t>& prim_of_slot) {
	SweepBuilder builder(prims, n, /*full_area=*/true);
	builder.run(nodes, nullptr, &prim_of_slot);
}

extern "C" int mirt_bvh_build(const mirt_sphere* geometry, uint32_t n, mirt_bvh_node* nodes_out, uint32_t* n_nodes_out, mirt_sphere* prims_out) {
	if ((!geometry && n) || !nodes_out || !n_nodes_out || (!prims_out && n)) return MIRT_ERR_ARG;
	for (uint32_t i = 0; i < n; i++)                                                     // NaNs would break the centroid sorts' ordering
		if (!std::isfinite(geometry[i].position[0]) || !std::isfinite(geometry[i].position[1]) || !std::isfinite(geometry[i].position[2]) ||
		    !std::isfinite(geometry[i].radius_sq) || geometry[i].radius_sq < 0.0f) return MIRT_ERR_ARG;
	std::vector<mirt_bvh_node> nodes;
	SweepBuilder builder(geometry, n);
	builder.run(nodes, prims_out);
	if (!nodes.empty()) std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(mirt_bvh_node));
	*n_nodes_out = static_cast<uint32_t>(nodes.size());
	return MIRT_OK;
}

extern "C" int mirt_light_list(const mirt_sphere* geometry, uint32_t n, const mirt_material* materials, uint32_t n_materials,
                               int32_t* lights_out, uint32_t* n_lights_out) {
	if ((!geometry && n) || (!materials && n_materials) || !lights_out || !n_lights_out) return MIRT_ERR_ARG;
	uint32_t count = 0;
	for (uint32_t i = 0; i < n; i++) {                                                 // Scene.hpp:13-15
		const int32_t m = geometry[i].material_ID;
		if (m < 0 || static_cast<uint32_t>(m) >= n_materials) return MIRT_ERR_ARG;
		const float* e = materials[m].emission;
		const float energy = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];               // glm::dot(em, em)
		if (energy > 0.0f) lights_out[count++] = static_cast<int32_t>(i);
	}
	*n_lights_out = count;
	return MIRT_OK;
}
