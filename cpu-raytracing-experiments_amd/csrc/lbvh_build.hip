// lbvh_build.hip — the GPU-internal traversal tree built ON the GPU (policy.gpu_build).
//
// the reference rebuilds its BVH on the host, single-threaded, after every geometry edit
// (BoundingVolumeHierarchy ctor BVH.hpp:90-206, called from Application.cpp:233,508).  The prim ORDER of that tree is part
// of the reference's results (hit.primID is compared with geometry-order light ids, Renderer.hpp:261-263), so the
// caller's tree keeps coming from mirt_bvh_build; what is built here is the tree the trace kernels actually walk, over
// the same BVH-order prims.  Traversal results do not depend on that tree (DESIGN.md "Traversal semantics"), which is what
// makes a different — here: Morton-order — tree admissible.
//
// Linear BVH after Karras 2012: 30-bit Morton codes of the sphere centres made unique by the prim index (64-bit keys),
// one device radix sort, one thread per inner node to find its key range and split, a bottom-up pass for the boxes, a
// second sort by (depth, node) for the breadth-first record order of bvh_layout.hpp, and one pass that writes the
// child-pair records (f32 and binary16) exactly as the host layout does: leaf boxes grown by 2^-18 (max|c| + r) and
// rounded outward, inner boxes unions, child with the larger half area first.
//
// The Morton grid is CUBIC and laid over the ordinary spheres only (round 3; round 2 normalised each axis by the bounds of all centres,
// so S(n)'s 1000-unit ground sphere stretched the y axis to 1046 units for a 46-unit layer of spheres: y was first split where x and z
// cells were already 6 units wide, and every node above that was a 46-unit column).  Spheres much larger than the typical one — radius
// above 16 x the median radius' power of two, found with an exponent histogram — get a key bit of their own and are split off at the
// root, which is where a SAH builder puts them too; the grid spans the centres of the others, with one cell size for the three axes.
// What remains is what LBVH trees are: median splits of a space-filling curve instead of surface-area-guided ones — in exchange for
// milliseconds instead of 0.1-0.3 s per 100 k spheres.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <string>

#include "lbvh_build.hpp"

namespace mirt_gpu {
namespace {

constexpr uint32_t kBlock = 256;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr float kPadRel = 0x1p-18f;                     // bvh_layout.hpp kPadRel

__device__ __forceinline__ uint32_t ordered_bits(float f) {               // monotone float -> uint
	const uint32_t b = __float_as_uint(f);
	return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered(uint32_t u) {
	return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ float next_up(float x) {                       // finite x
	if (x == 0.0f) return __uint_as_float(1u);
	uint32_t b = __float_as_uint(x);
	return __uint_as_float(x > 0.0f ? b + 1u : b - 1u);
}
__device__ __forceinline__ float next_down(float x) { return -next_up(-x); }

struct Box { float lo[3], hi[3]; };

// Leaf boxes as in mirt_host::build_records, plus a histogram of the radii's binary exponents (integer atomics: the same tree every run).
__global__ __launch_bounds__(kBlock) void k_leaf_boxes(const float4* __restrict__ spheres, uint32_t n, Box* __restrict__ leaf_box, uint32_t* __restrict__ exp_hist /* [256] */) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const float4 s = spheres[i];
	const float c[3] = { s.x, s.y, s.z };
	const float r = __builtin_sqrtf(s.w);
	float amax = fabsf(c[0]);
	if (amax < fabsf(c[1])) amax = fabsf(c[1]);
	if (amax < fabsf(c[2])) amax = fabsf(c[2]);
	const float pad = kPadRel * (amax + r);
	Box b;
	for (int a = 0; a < 3; a++) {
		b.lo[a] = next_down((c[a] - r) - pad);
		b.hi[a] = next_up((c[a] + r) + pad);
	}
	leaf_box[i] = b;
	atomicAdd(&exp_hist[(__float_as_uint(r) >> 23) & 0xffu], 1u);
}
// "Large" = radius of 16 x 2^(median exponent + 1) or more (one thread walks the 256 bins); then the bounds of the ordinary spheres' centres.
__global__ void k_large_threshold(const uint32_t* __restrict__ exp_hist, uint32_t n, float* __restrict__ threshold) {
	uint32_t seen = 0, e = 0;
	for (; e < 256u; e++) { seen += exp_hist[e]; if (2u * seen >= n) break; }
	const uint32_t te = e + 5u;                                              // 2^(e - 127 + 1) x 16
	*threshold = te >= 255u ? __builtin_inff() : __uint_as_float(te << 23);
}
__global__ __launch_bounds__(kBlock) void k_centre_bounds(const float4* __restrict__ spheres, uint32_t n, const float* __restrict__ threshold, uint32_t* centre_bounds /* lo[3], hi[3] */) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const float4 s = spheres[i];
	if (!(__builtin_sqrtf(s.w) < *threshold)) return;
	const float c[3] = { s.x, s.y, s.z };
	for (int a = 0; a < 3; a++) { atomicMin(&centre_bounds[a], ordered_bits(c[a])); atomicMax(&centre_bounds[3 + a], ordered_bits(c[a])); }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {                // 10 bits -> every third bit
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
__global__ __launch_bounds__(kBlock) void k_morton(const float4* __restrict__ spheres, uint32_t n, const uint32_t* __restrict__ centre_bounds, const float* __restrict__ threshold,
                                                   uint64_t* __restrict__ keys) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const float4 s = spheres[i];
	const float c[3] = { s.x, s.y, s.z };
	float lo[3], cell = 0.0f;                                                // one cell size for the three axes: the largest extent / 1024
	for (int a = 0; a < 3; a++) { lo[a] = from_ordered(centre_bounds[a]); cell = fmaxf(cell, from_ordered(centre_bounds[3 + a]) - lo[a]); }
	uint32_t q[3];
	for (int a = 0; a < 3; a++) {
		float u = cell > 0.0f ? (c[a] - lo[a]) / cell : 0.0f;
		u = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);                         // (a large sphere's centre may lie far outside the grid)
		q[a] = static_cast<uint32_t>(u);
	}
	const uint32_t large = (__builtin_sqrtf(s.w) < *threshold) ? 0u : 1u;
	const uint32_t code = (large << 30) | (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
	keys[i] = (static_cast<uint64_t>(code) << 32) | i;                      // unique: ties broken by the prim index
}

// Karras 2012, one thread per inner node i in [0, n-1): children are inner nodes (index) or leaves (kLeafBit | sorted position).
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
	if (j < 0 || j >= n) return -1;
	return __clzll(static_cast<long long>(keys[i] ^ keys[j]));
}
__global__ __launch_bounds__(kBlock) void k_karras(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ child0, uint32_t* __restrict__ child1,
                                                   uint32_t* __restrict__ parent_of_inner, uint32_t* __restrict__ parent_of_leaf) {
	const int i = static_cast<int>(blockIdx.x * kBlock + threadIdx.x);
	const int nn = static_cast<int>(n);
	if (i >= nn - 1) return;
	const int d = (delta(keys, nn, i, i + 1) - delta(keys, nn, i, i - 1)) >= 0 ? 1 : -1;
	const int dmin = delta(keys, nn, i, i - d);
	int lmax = 2;
	while (delta(keys, nn, i, i + lmax * d) > dmin) lmax *= 2;
	int l = 0;
	for (int t = lmax / 2; t >= 1; t /= 2) if (delta(keys, nn, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	const int dnode = delta(keys, nn, i, j);
	int s = 0;
	for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
		if (delta(keys, nn, i, i + (s + t) * d) > dnode) s += t;
		if (t <= 1) break;
	}
	const int gamma = i + s * d + min(d, 0);
	const int first = min(i, j), last = max(i, j);
	const uint32_t c0 = (first == gamma) ? (kLeafBit | static_cast<uint32_t>(gamma)) : static_cast<uint32_t>(gamma);
	const uint32_t c1 = (last == gamma + 1) ? (kLeafBit | static_cast<uint32_t>(gamma + 1)) : static_cast<uint32_t>(gamma + 1);
	child0[i] = c0; child1[i] = c1;
	if (c0 & kLeafBit) parent_of_leaf[gamma] = i; else parent_of_inner[gamma] = i;
	if (c1 & kLeafBit) parent_of_leaf[gamma + 1] = i; else parent_of_inner[gamma + 1] = i;
	if (i == 0) parent_of_inner[0] = 0xffffffffu;
}

__device__ __forceinline__ Box box_union(const Box& x, const Box& y) {
	Box b;
	for (int a = 0; a < 3; a++) { b.lo[a] = (y.lo[a] < x.lo[a]) ? y.lo[a] : x.lo[a]; b.hi[a] = (x.hi[a] < y.hi[a]) ? y.hi[a] : x.hi[a]; }
	return b;
}
// Bottom-up: one thread per leaf climbs; the second thread to reach a node owns it (both child boxes are then complete).
__global__ __launch_bounds__(kBlock) void k_inner_boxes(const uint64_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ child0, const uint32_t* __restrict__ child1,
                                                        const uint32_t* __restrict__ parent_of_inner, const uint32_t* __restrict__ parent_of_leaf,
                                                        const Box* __restrict__ leaf_box, Box* inner_box, uint32_t* arrivals) {
	const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
	if (j >= n) return;
	uint32_t node = parent_of_leaf[j];
	while (node != 0xffffffffu) {
		__threadfence();
		if (atomicAdd(&arrivals[node], 1u) == 0u) return;                     // first arrival: the sibling subtree is not finished yet
		__threadfence();
		const uint32_t a = child0[node], b = child1[node];
		const volatile Box* ib = inner_box;
		Box ba, bb;
		if (a & kLeafBit) ba = leaf_box[static_cast<uint32_t>(keys[a & ~kLeafBit])];
		else for (int k = 0; k < 3; k++) { ba.lo[k] = ib[a].lo[k]; ba.hi[k] = ib[a].hi[k]; }
		if (b & kLeafBit) bb = leaf_box[static_cast<uint32_t>(keys[b & ~kLeafBit])];
		else for (int k = 0; k < 3; k++) { bb.lo[k] = ib[b].lo[k]; bb.hi[k] = ib[b].hi[k]; }
		inner_box[node] = box_union(ba, bb);
		node = parent_of_inner[node];
	}
}

// depth of every inner node (root = 1) and the (depth, node) keys of the breadth-first order
__global__ __launch_bounds__(kBlock) void k_depth_keys(uint32_t n_inner, const uint32_t* __restrict__ parent_of_inner, uint64_t* __restrict__ order_keys, uint32_t* max_depth) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_inner) return;
	uint32_t depth = 1;
	for (uint32_t p = parent_of_inner[i]; p != 0xffffffffu; p = parent_of_inner[p]) depth++;
	order_keys[i] = (static_cast<uint64_t>(depth) << 32) | i;
	atomicMax(max_depth, depth + 1u);                                       // + the leaf level below
}
__global__ __launch_bounds__(kBlock) void k_record_index(uint32_t n_inner, const uint64_t* __restrict__ order_keys, uint32_t* __restrict__ rec_of) {
	const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
	if (r >= n_inner) return;
	rec_of[static_cast<uint32_t>(order_keys[r])] = r;
}

__device__ __forceinline__ float half_area(const Box& b) {
	const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
	return (dx * dy + dy * dz) + dz * dx;
}
__device__ __forceinline__ uint32_t half_down(float f) {                  // largest binary16 <= f
	return static_cast<uint32_t>(__half_as_ushort(__float2half_rd(f)));
}
__device__ __forceinline__ uint32_t half_up(float f) {
	return static_cast<uint32_t>(__half_as_ushort(__float2half_ru(f)));
}
// One child-pair record per inner node, in the layouts of bvh_layout.hpp (64-B f32 and 32-B binary16).
__global__ __launch_bounds__(kBlock) void k_emit_records(const uint64_t* __restrict__ keys, uint32_t n_inner, const uint32_t* __restrict__ child0, const uint32_t* __restrict__ child1,
                                                         const uint32_t* __restrict__ rec_of, const Box* __restrict__ leaf_box, const Box* __restrict__ inner_box,
                                                         float* __restrict__ recs32, uint32_t* __restrict__ recs16) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_inner) return;
	uint32_t ref[2] = { child0[i], child1[i] };
	Box b[2];
	for (int c = 0; c < 2; c++) {
		if (ref[c] & kLeafBit) {
			const uint32_t prim = static_cast<uint32_t>(keys[ref[c] & ~kLeafBit]);   // index into the BVH-order prims = hit.primID
			b[c] = leaf_box[prim];
			ref[c] = kLeafBit | prim;                                               // one prim per leaf: count-1 = 0 in bits 24..30
		} else {
			b[c] = inner_box[ref[c]];
			ref[c] = rec_of[ref[c]];
		}
	}
	const int first = (half_area(b[0]) < half_area(b[1])) ? 1 : 0;                 // the child with the larger half area goes first
	const uint32_t r = rec_of[i];
	if (recs32) {
		float* q = recs32 + static_cast<size_t>(r) * 16;
		for (int a = 0; a < 3; a++) {
			q[a * 4 + 0] = b[first].lo[a]; q[a * 4 + 1] = b[first ^ 1].lo[a];
			q[a * 4 + 2] = b[first].hi[a]; q[a * 4 + 3] = b[first ^ 1].hi[a];
		}
		q[12] = __uint_as_float(ref[first]); q[13] = __uint_as_float(ref[first ^ 1]); q[14] = 0.0f; q[15] = 0.0f;
	}
	if (recs16) {
		uint32_t* w = recs16 + static_cast<size_t>(r) * 8;
		for (int a = 0; a < 3; a++) {
			w[a * 2] = half_down(b[first].lo[a]) | (half_down(b[first ^ 1].lo[a]) << 16);
			w[a * 2 + 1] = half_up(b[first].hi[a]) | (half_up(b[first ^ 1].hi[a]) << 16);
		}
		w[6] = ref[first]; w[7] = ref[first ^ 1];
	}
}

// 4-WIDE binary16 records (bvh_layout.hpp build_wide_half_records): a wide node is an inner node at an odd depth (root = 1) — it
// absorbs its inner children.  wide_flag[r] = 1 for the wide nodes in breadth-first record order r; its exclusive sum numbers them.
__global__ __launch_bounds__(kBlock) void k_wide_flags(uint32_t n_inner, const uint64_t* __restrict__ order_sorted, uint32_t* __restrict__ flag) {
	const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
	if (r >= n_inner) return;
	flag[r] = static_cast<uint32_t>(order_sorted[r] >> 32) & 1u;
}
__global__ __launch_bounds__(kBlock) void k_emit_wide(const uint64_t* __restrict__ keys, uint32_t n_inner, const uint64_t* __restrict__ order_sorted, const uint32_t* __restrict__ child0,
                                                      const uint32_t* __restrict__ child1, const uint32_t* __restrict__ rec_of, const uint32_t* __restrict__ wide_index,
                                                      const Box* __restrict__ leaf_box, const Box* __restrict__ inner_box, uint32_t* __restrict__ recs_wide, uint32_t* n_wide) {
	const uint32_t r = blockIdx.x * kBlock + threadIdx.x;                   // breadth-first record number of a binary node
	if (r >= n_inner) return;
	const uint64_t key = order_sorted[r];
	if (((key >> 32) & 1u) == 0u) return;                                   // even depth: absorbed by its parent
	const uint32_t i = static_cast<uint32_t>(key);
	atomicMax(n_wide, wide_index[r] + 1u);                                  // the count of wide records = the last index + 1
	Box b[4]; uint32_t ref[4]; int nk = 0;
	auto child_pair = [&](uint32_t node, uint32_t out[2]) {                // a node's children, the one with the larger half area first (as k_emit_records)
		const uint32_t c[2] = { child0[node], child1[node] };
		Box cb[2];
		for (int k = 0; k < 2; k++) cb[k] = (c[k] & kLeafBit) ? leaf_box[static_cast<uint32_t>(keys[c[k] & ~kLeafBit])] : inner_box[c[k]];
		const int first = (half_area(cb[0]) < half_area(cb[1])) ? 1 : 0;
		out[0] = c[first]; out[1] = c[first ^ 1];
	};
	uint32_t c[2]; child_pair(i, c);
	for (int k = 0; k < 2; k++) {
		if (c[k] & kLeafBit) {
			const uint32_t prim = static_cast<uint32_t>(keys[c[k] & ~kLeafBit]);
			b[nk] = leaf_box[prim]; ref[nk] = kLeafBit | prim; nk++;
		} else {
			uint32_t g[2]; child_pair(c[k], g);
			for (int j = 0; j < 2; j++) {
				if (g[j] & kLeafBit) { const uint32_t prim = static_cast<uint32_t>(keys[g[j] & ~kLeafBit]); b[nk] = leaf_box[prim]; ref[nk] = kLeafBit | prim; }
				else { b[nk] = inner_box[g[j]]; ref[nk] = wide_index[rec_of[g[j]]]; }      // an inner grandchild is two levels down: a wide node again
				nk++;
			}
		}
	}
	uint32_t lo[3][4], hi[3][4];
	for (int k = 0; k < 4; k++) {
		for (int a = 0; a < 3; a++) { lo[a][k] = k < nk ? half_down(b[k].lo[a]) : 0x7c00u; hi[a][k] = k < nk ? half_up(b[k].hi[a]) : 0x7c00u; }   // unused slot: the +inf box no slab test hits
		if (k >= nk) ref[k] = ref[0];
	}
	uint32_t* q = recs_wide + static_cast<size_t>(wide_index[r]) * 16;
	for (int a = 0; a < 3; a++) {
		q[a * 4 + 0] = lo[a][0] | (lo[a][1] << 16); q[a * 4 + 1] = lo[a][2] | (lo[a][3] << 16);
		q[a * 4 + 2] = hi[a][0] | (hi[a][1] << 16); q[a * 4 + 3] = hi[a][2] | (hi[a][3] << 16);
	}
	for (int k = 0; k < 4; k++) q[12 + k] = ref[k];
}

struct Scratch {
	void* p = nullptr;
	hipError_t get(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
	~Scratch() { if (p) (void)hipFree(p); }
	template <class T> T* as() const { return static_cast<T*>(p); }
};

#define LBVH_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { if (err) *err = std::string(#expr) + ": " + hipGetErrorString(_e); return false; } } while (0)

} // namespace

bool build_lbvh(hipStream_t st, const float4* spheres, uint32_t n, float* recs32, uint32_t* recs16, uint32_t* depth_out, std::string* err,
                uint32_t* recs_wide, uint32_t* n_wide_out) {
	if (n < 2) { if (err) *err = "fewer than two spheres"; return false; }
	const uint32_t n_inner = n - 1;
	Scratch leaf_box, inner_box, keys, keys_sorted, order_keys, order_sorted, c0, c1, par_inner, par_leaf, small, sort_tmp, hist;
	LBVH_TRY(leaf_box.get(sizeof(Box) * n)); LBVH_TRY(inner_box.get(sizeof(Box) * n_inner));
	LBVH_TRY(keys.get(8ull * n)); LBVH_TRY(keys_sorted.get(8ull * n));
	LBVH_TRY(order_keys.get(8ull * n_inner)); LBVH_TRY(order_sorted.get(8ull * n_inner));
	LBVH_TRY(c0.get(4ull * n_inner)); LBVH_TRY(c1.get(4ull * n_inner));
	LBVH_TRY(par_inner.get(4ull * n_inner)); LBVH_TRY(par_leaf.get(4ull * n));
	// small: centre bounds lo[3] (init all ones) hi[3] (init 0), max depth, then the per-node arrival counters
	LBVH_TRY(small.get(4ull * (8 + n_inner)));
	uint32_t* centre_bounds = small.as<uint32_t>();
	uint32_t* max_depth = centre_bounds + 6;
	uint32_t* arrivals = centre_bounds + 8;
	LBVH_TRY(hist.get(4ull * 260));                                         // 256 exponent bins + the threshold
	uint32_t* exp_hist = hist.as<uint32_t>();
	float* threshold = reinterpret_cast<float*>(exp_hist + 256);
	LBVH_TRY(hipMemsetAsync(exp_hist, 0, 4ull * 260, st));
	LBVH_TRY(hipMemsetAsync(centre_bounds, 0xff, 12, st));
	LBVH_TRY(hipMemsetAsync(centre_bounds + 3, 0, 4ull * (5 + n_inner), st));
	size_t tmp_bytes = 0, tmp2 = 0;
	LBVH_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys.as<uint64_t>(), keys_sorted.as<uint64_t>(), static_cast<int>(n), 0, 63, st));
	LBVH_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp2, order_keys.as<uint64_t>(), order_sorted.as<uint64_t>(), static_cast<int>(n_inner), 0, 40, st));
	if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
	LBVH_TRY(sort_tmp.get(tmp_bytes));

	const dim3 gl((n + kBlock - 1) / kBlock), gi((n_inner + kBlock - 1) / kBlock), blk(kBlock);
	hipLaunchKernelGGL(k_leaf_boxes, gl, blk, 0, st, spheres, n, leaf_box.as<Box>(), exp_hist);
	hipLaunchKernelGGL(k_large_threshold, dim3(1), dim3(1), 0, st, exp_hist, n, threshold);
	hipLaunchKernelGGL(k_centre_bounds, gl, blk, 0, st, spheres, n, threshold, centre_bounds);
	hipLaunchKernelGGL(k_morton, gl, blk, 0, st, spheres, n, centre_bounds, threshold, keys.as<uint64_t>());
	LBVH_TRY(hipcub::DeviceRadixSort::SortKeys(sort_tmp.p, tmp_bytes, keys.as<uint64_t>(), keys_sorted.as<uint64_t>(), static_cast<int>(n), 0, 63, st));
	hipLaunchKernelGGL(k_karras, gi, blk, 0, st, keys_sorted.as<uint64_t>(), n, c0.as<uint32_t>(), c1.as<uint32_t>(), par_inner.as<uint32_t>(), par_leaf.as<uint32_t>());
	hipLaunchKernelGGL(k_inner_boxes, gl, blk, 0, st, keys_sorted.as<uint64_t>(), n, c0.as<uint32_t>(), c1.as<uint32_t>(), par_inner.as<uint32_t>(), par_leaf.as<uint32_t>(),
	                   leaf_box.as<Box>(), inner_box.as<Box>(), arrivals);
	hipLaunchKernelGGL(k_depth_keys, gi, blk, 0, st, n_inner, par_inner.as<uint32_t>(), order_keys.as<uint64_t>(), max_depth);
	LBVH_TRY(hipcub::DeviceRadixSort::SortKeys(sort_tmp.p, tmp_bytes, order_keys.as<uint64_t>(), order_sorted.as<uint64_t>(), static_cast<int>(n_inner), 0, 40, st));
	uint32_t* rec_of = par_leaf.as<uint32_t>();                            // parent_of_leaf is dead after k_inner_boxes: reuse (n >= n_inner words)
	hipLaunchKernelGGL(k_record_index, gi, blk, 0, st, n_inner, order_sorted.as<uint64_t>(), rec_of);
	hipLaunchKernelGGL(k_emit_records, gi, blk, 0, st, keys_sorted.as<uint64_t>(), n_inner, c0.as<uint32_t>(), c1.as<uint32_t>(), rec_of, leaf_box.as<Box>(), inner_box.as<Box>(),
	                   recs32, recs16);
	LBVH_TRY(hipGetLastError());
	uint32_t depth = 0, n_wide = 0;
	Scratch wide_flag, wide_index, scan_tmp;
	if (recs_wide) {
		// number the wide nodes (odd depth) in breadth-first order and emit one 64-B record each
		LBVH_TRY(wide_flag.get(4ull * n_inner)); LBVH_TRY(wide_index.get(4ull * n_inner));
		size_t scan_bytes = 0;
		LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, wide_flag.as<uint32_t>(), wide_index.as<uint32_t>(), static_cast<int>(n_inner), st));
		LBVH_TRY(scan_tmp.get(scan_bytes));
		hipLaunchKernelGGL(k_wide_flags, gi, blk, 0, st, n_inner, order_sorted.as<uint64_t>(), wide_flag.as<uint32_t>());
		LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, wide_flag.as<uint32_t>(), wide_index.as<uint32_t>(), static_cast<int>(n_inner), st));
		uint32_t* n_wide_dev = centre_bounds + 7;                           // a spare word of `small` (zeroed above)
		hipLaunchKernelGGL(k_emit_wide, gi, blk, 0, st, keys_sorted.as<uint64_t>(), n_inner, order_sorted.as<uint64_t>(), c0.as<uint32_t>(), c1.as<uint32_t>(), rec_of, wide_index.as<uint32_t>(),
		                   leaf_box.as<Box>(), inner_box.as<Box>(), recs_wide, n_wide_dev);
		LBVH_TRY(hipGetLastError());
		LBVH_TRY(hipMemcpyAsync(&n_wide, n_wide_dev, 4, hipMemcpyDeviceToHost, st));
	}
	LBVH_TRY(hipMemcpyAsync(&depth, max_depth, 4, hipMemcpyDeviceToHost, st));
	LBVH_TRY(hipStreamSynchronize(st));
	if (depth_out) *depth_out = depth;
	if (n_wide_out) *n_wide_out = n_wide;
	return true;
}

} // namespace mirt_gpu
