// kernels.hpp — HIP kernels of the wavefront path tracer (gfx950 / CDNA4, wave64).
//
// One Renderer::Accumulate() call of the reference (Renderer.hpp:73-434) runs a 256-ray stream per
// 16x16 tile through raygen -> [intersect -> closest-hit shade -> sort -> NEE -> shadow trace ->
// emissive -> BRDF sample/RR/compaction -> miss -> accumulate] on one CPU thread.  Here the same
// path state is a set of SoA ray streams in HBM spanning every pixel this GPU owns times the
// accumulations in flight, and one bounce is two kernels (plus a few-microsecond k_trace_fat for the rare stretched rays):
//
//   k_trace           Traverse (BVH.hpp:309-360) for the rays of bounce b: reads p,dir, writes tfar,primID — and, in the
//                     same launch, Traverse_shadow (BVH.hpp:362-404) for the NEE rays of bounce b-1, each followed by
//                     Renderer.hpp:304-314 and the deferred radiance finalisation ((R + unoccluded NEE) + emissive) in the
//                     reference's add order
//   k_shade           Renderer.hpp:169-431 except the shadow-dependent adds; compacts survivors into the next stream and
//                     NEE candidates into the shadow stream (wave64 ballot + mbcnt prefix sums, one atomic per workgroup)
//   k_primary_cand /  bounce 0 only (RAY GENERATION + the first Traverse, Renderer.hpp:113-127,165): the camera rays have no stream — a
//   k_primary_hits    ray is a function of its index — and the jittered samples of a pixel (one per accumulation of the batch, up to 256) share ONE cone traversal that lists
//                     the spheres they can hit; each sample then tests only its pixel's list (kCollect below)
//
// Per-path results do not depend on stream slot or scheduling: every random draw is re-derived from
// (accumulations, seed[pixel], bounce) (Renderer.hpp:107,117,255,362), which is what lets the
// wavefront reorder rays freely and still reproduce the reference bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_math.hpp"

namespace mirt {

constexpr uint32_t kBlock = 256;
constexpr uint32_t kShadeBlock = 512;           // shade: 73-79 VGPRs = 24 waves per CU = three 8-wave workgroups (forced to 64 VGPRs for a fourth: 15-20 spills, 92 -> 109 ms per cfg4 step; 1024-thread workgroups: one per CU, every barrier stalls the CU)
constexpr uint32_t kTraceBlock = 1024;          // trace kernels: one workgroup per CU stages the BVH into LDS once per launch
constexpr uint32_t kLdsStack = 16;              // traversal-stack entries per lane kept in LDS (deeper ones spill to scratch)
constexpr uint32_t kLdsStackWide = 12;          // ... with binary16 records but u32 entries (> 32768 records or spheres): 48 KB, so that two workgroups still share a CU
constexpr uint32_t kLeafBit = 0x80000000u;      // child reference flag (bvh_layout.hpp)
constexpr uint32_t kTileSize = 256;
constexpr uint32_t kTileRoot = 16;
constexpr uint32_t kStack = 64;                 // Stack<StackFrame,64>, BVH.hpp:321
constexpr uint32_t kDestAccum = 0x80000000u;    // shadow-entry destination flag: accumulator instead of stream slot

struct SceneDev {
	const float4* recs;         // GPU-internal child-pair records, 4 float4 each, breadth-first (bvh_layout.hpp)
	const float4* spheres;      // acceleration_structure.prims (BVH order): {pos.xyz, radius_sq}
	const int32_t* prim_mat;    //   "  material_ID
	const float4* light_sphere; // per light (lighting_acceleration.prims order): scene.geometry[light] = {pos.xyz, radius_sq} (Renderer.hpp:261-262)
	const float4* light_emit;   //   "   {emission of its material .xyz, bits of the geometry-order prim id} (Renderer.hpp:263,283)
	const float4* mat_albedo;   // scene.material[].albedo
	const float4* mat_emission; // scene.material[].emission
	const float4* hdri;         // sky.hdri_data RGBA
	uint32_t n_spheres, n_recs, n_mat, n_lights;
	uint32_t lds_recs;          // records [0, lds_recs) are staged in LDS by the trace kernels (top of the tree)
	uint32_t lds_spheres;       // spheres [0, lds_spheres) likewise (all of them, or none)
	uint32_t half_boxes;        // 1: recs are the 32-B binary16 records (2 float4 each)
	uint32_t wide;              // 1 (with half_boxes): recs are 64-B binary16 records of up to FOUR children (bvh_layout.hpp build_wide_half_records)
	uint32_t chunk_max;         // rays per reservation from a launch's work counter, upper limit (pick_chunk)
	uint32_t leaf_batch;        // lanes of a wave that must stand at a leaf before it runs a leaf pass (trace_persistent)
	uint32_t refill_idle;       // idle lanes of a wave that trigger a refill (trace_persistent)
	uint32_t stack16;           // 1: record and prim indices fit 15 bits and the LDS stack holds u16 entries (binary16 records, <= 32768 records and spheres)
	float ambient[3];
	int32_t hdri_w, hdri_h;
	float hdri_fw, hdri_fh;
	uint32_t has_ambient;       // Renderer.hpp:79
	uint32_t use_bvh;
};

// RayStream<>::Buffer, DataStreams.hpp:75-88 — one SoA plane per member, `capacity` rays each.
struct StreamBuf {
	float *px, *py, *pz, *dx, *dy, *dz;
	float *tr, *tg, *tb;        // throughput
	float *rr, *rg, *rb;        // radiance
	                            // (`pdf` has no plane: Closure::pdf of the sampled direction is (1/pi) max(0, dir.z) of the WORLD-space dir
	                            //  (Q8, Renderer.hpp:386,401), a function of the stored direction, recomputed by the bounce that needs it)
	uint32_t* path;             // (batch slot << FrameParams::pix_bits) | local pixel index (tile_local*256 + ID); stands in for pixelID + seed[]
};
// RayStream<>::ShadowStream, DataStreams.hpp:113-126, plus the deferred-add operands.  A record is 32 B for the common case
// (dir, tfar, NEE radiance, destination): the origin is the surviving ray's own (read from the next stream through `dest`;
// stored here only for paths that Russian roulette ended), and the path radiance waits in the destination word itself.
struct ShadowBuf {
	float *px, *py, *pz;        // origin — written only when dest is the accumulator (no surviving ray to share it with)
	float *dx, *dy, *dz, *tfar;
	float *sr, *sg, *sb;        // NEE radiance carried by the shadow ray
	float *rr, *rg, *rb;        // kDestFull records only: path radiance before this bounce's adds
	float *er, *eg, *eb;        // kDestFull records only: emissive add of this bounce
	uint32_t* dest;             // next-stream slot, or kDestAccum | path id; | kDestFull
};
struct FrameParams {
	CameraParams cam;
	uint32_t h_tiles;           // Renderer.hpp:59
	uint32_t first_tile;        // first global LaunchIndex owned by this context
	uint32_t run_tiles;         // the context owns runs of run_tiles consecutive LaunchIndices, stride_tiles apart (interleaved tile rows,
	uint32_t stride_tiles;      // mirt_set_tile_rows); stride_tiles = 0: one contiguous range (mirt_set_tile_range)
	uint32_t n_pix;             // local pixels = local tiles * 256
	uint32_t pix_bits;          // a path id is (batch slot << pix_bits) | local pixel, below 2^30: the fewer pixels a context owns, the more
	uint32_t pix_mask;          // accumulations fit a batch ((1 << pix_bits) - 1)
	uint32_t acc_base;          // `accumulations` before this batch
	uint32_t batch_n;           // accumulations in flight in this batch
	uint32_t idx_base;          // where the paths' radiance is added (accum_index): straight into the accumulator (idx_base = acc_base,
	uint32_t idx_buckets;       // idx_buckets = buckets) or into this batch's contribution buffer [tile][slot][rgb][256] (0xffffffff, batch_n)
	uint32_t max_bounces;
	uint32_t buckets;
	uint32_t mis;               // MIS && light_count > 0 (Q12 guard)
	uint32_t n_lights;
	uint32_t first_groups;      // k_shade<FIRST>: a chunk of 512 pixels x the batch's accumulations is handed out in this many pieces (small images: enough pieces for an even load)
	float inv_n_pix, inv_h_tiles, inv_run_tiles;   // 1/n_pix, 1/h_tiles, 1/run_tiles for udiv_f (no integer division in the kernels)
};
// RayStream<>::Hit, DataStreams.hpp:90-111 (tfar, primID; matID is looked up by the shader): ONE 8-B record per ray, written by the lane that finished
// the ray and read back by k_shade with one instruction each (two dword planes until round 3: two scattered 4-B stores and loads per ray).
struct alignas(8) HitRec { float tfar; int32_t prim; };
struct DevCounters {
	unsigned long long rays, shadow_rays, nodes, spheres, shadow_nodes, shadow_spheres, terminated, dropped;
};

// ------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------
MIRT_DI uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
MIRT_DI uint32_t mask_rank(unsigned long long m) {            // set bits of m below this lane
	return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}
MIRT_DI void wave_sum(uint32_t v, unsigned long long* counter) {
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	if (lane_id() == 0 && v) atomicAdd(counter, static_cast<unsigned long long>(v));
}

// ---- ray queues -----------------------------------------------------------------------------------------------
// A ray queue (the rays of one bounce, or its NEE shadow rays) is kSegs dense SEGMENTS of the stream planes at fixed offsets:
// segment k holds slots [k * seg_cap, k * seg_cap + n[k]).  Producers (k_shade's compaction) append block-iteration t to
// segment t % kSegs, so the one returning atomicAdd per workgroup and 512 rays is spread over kSegs counters, each on a
// cache line of its own: same-address returning atomics serialise at ~11 ns each (MI355X_MICROARCH.md "dequeue": one word
// saturates at ~88 per microsecond), and with a single counter that alone was 1.9 ms of the 3.3 ms primary-ray k_shade launch
// of a 4096x4096 batch (164 k appends).  A segment receives at most ceil(T / kSegs) iterations of <= 512 rays, T =
// ceil(input rays / 512), hence seg_cap = ceil(ceil(capacity / 512) / kSegs) * 512 never overflows.  Consumers number the
// rays 0 .. total-1 across the segments in order (queue_view / queue_slot); per-path results do not depend on slot order.
constexpr uint32_t kSegs = 8;
constexpr uint32_t kSegPitch = 32;              // u32 words between two segment counters (128 B)
struct Queue { uint32_t* n; uint32_t seg_cap; };  // n[k * kSegPitch] = rays in segment k
struct QueueView { uint32_t pre[kSegs + 1]; uint32_t seg_cap; };     // pre[k] = rays before segment k, pre[kSegs] = total
MIRT_DI QueueView queue_view(const Queue& q) {
	QueueView v; v.seg_cap = q.seg_cap; v.pre[0] = 0;
	for (uint32_t k = 0; k < kSegs; k++) v.pre[k + 1] = v.pre[k] + q.n[k * kSegPitch];     // wave-uniform: scalar loads
	return v;
}
MIRT_DI QueueView queue_identity(uint32_t total) {     // bounce 0: ray i is slot i (the rays are generated from their index)
	QueueView v; v.seg_cap = 0; v.pre[0] = 0;
	for (uint32_t k = 1; k <= kSegs; k++) v.pre[k] = total;
	return v;
}
// The piece [first, end) of the numbering that lies in first's segment, and the offset that turns its numbers into slots.
struct QueuePiece { uint32_t end, offset; };
MIRT_DI QueuePiece queue_piece(const QueueView& v, uint32_t first) {
	uint32_t seg = 0, lo = 0, hi = v.pre[1];
	for (uint32_t k = 1; k < kSegs; k++) { const bool ge = first >= v.pre[k]; seg = ge ? k : seg; lo = ge ? v.pre[k] : lo; hi = ge ? v.pre[k + 1] : hi; }
	return QueuePiece{ hi, seg * v.seg_cap - lo };
}
// The same straight from the counters in memory (k_trace calls it once per reserved chunk of rays and keeps no view in registers:
// the kernel sits at its 64-VGPR budget).  q.n == nullptr: the identity numbering of bounce 0, `total` rays.
MIRT_DI QueuePiece queue_piece(const Queue& q, uint32_t total, uint32_t first) {
	if (q.n == nullptr) return QueuePiece{ total, 0u };
	uint32_t lo = 0, seg = 0, end = 0;
	for (uint32_t k = 0; k < kSegs; k++) {
		const uint32_t cnt = q.n[k * kSegPitch];
		if (first >= lo + cnt && k + 1 < kSegs) { lo += cnt; continue; }
		seg = k; end = lo + cnt; break;
	}
	return QueuePiece{ end, seg * q.seg_cap - lo };
}
MIRT_DI uint32_t queue_total(const Queue& q) { uint32_t t = 0; for (uint32_t k = 0; k < kSegs; k++) t += q.n[k * kSegPitch]; return t; }
MIRT_DI uint32_t queue_slot(const QueueView& v, uint32_t i) { return i + queue_piece(v, i).offset; }     // per-lane form
// The same for consecutive numbers starting at the wave-uniform `first`: scalar segment search, per-lane work only for the
// lanes (if any) that fall into a later segment.
MIRT_DI uint32_t queue_slot(const QueueView& v, uint32_t first, uint32_t i) {
	const QueuePiece p = queue_piece(v, first);
	return i < p.end ? i + p.offset : queue_slot(v, i);
}

// Two-queue compaction for a whole workgroup: wave64 ballots give per-wave counts, one lane per queue adds the workgroup
// total to the counter of segment `seg` (ONE global atomic per queue per `blockDim.x` rays), and every lane gets
// segment start + base + (counts of earlier waves) + (rank inside its wave).  Must be called by all threads of the block,
// converged.  `scratch` = 2*2*16+2*2 words of LDS, double-buffered on `parity` so consecutive calls need no third barrier.
MIRT_DI void block_append2(bool flag_a, bool flag_b, const Queue& qa, const Queue& qb, uint32_t seg, uint32_t* scratch, uint32_t parity,
                           uint32_t& slot_a, uint32_t& slot_b) {
	const unsigned long long ma = __ballot(flag_a), mb = __ballot(flag_b);
	const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6, lane = lane_id();
	uint32_t* cnt_a = scratch + parity * 36u;       // [16] a, [16] b, base a, base b, pad
	uint32_t* cnt_b = cnt_a + 16u;
	uint32_t* base = cnt_a + 32u;
	if (lane == 0) { cnt_a[wave] = static_cast<uint32_t>(__popcll(ma)); cnt_b[wave] = static_cast<uint32_t>(__popcll(mb)); }
	__syncthreads();
	if (threadIdx.x < 2) {
		const uint32_t* cnt = threadIdx.x ? cnt_b : cnt_a;
		uint32_t total = 0;
		for (uint32_t w = 0; w < n_waves; w++) total += cnt[w];
		base[threadIdx.x] = total ? atomicAdd((threadIdx.x ? qb.n : qa.n) + seg * kSegPitch, total) : 0u;
	}
	__syncthreads();
	uint32_t pa = base[0] + seg * qa.seg_cap, pb = base[1] + seg * qb.seg_cap;
	for (uint32_t w = 0; w < wave; w++) { pa += cnt_a[w]; pb += cnt_b[w]; }
	slot_a = pa + mask_rank(ma);
	slot_b = pb + mask_rank(mb);
}

// ------------------------------------------------------------------------------------------------
// Intersection arithmetic
// ------------------------------------------------------------------------------------------------
// 8-ray x 1-sphere AVX2 body of intersect_prims, BVH.hpp:251-267, one lane per ray: the FMA chain as written
// there; accept iff dist < tfar, sign(dist) clear, sign(sqrt(disc)) clear (= disc not negative / -0).
MIRT_DI void sphere_closest(float4 s, int32_t prim, float px, float py, float pz, float dx, float dy, float dz,
                            float& tfar, int32_t& primID) {
	float tx = s.x - px;
	float b = dx * tx;
	float disc = __builtin_fmaf(-tx, tx, s.w);
	float ty = s.y - py;
	b = __builtin_fmaf(dy, ty, b);
	disc = __builtin_fmaf(-ty, ty, disc);
	float tz = s.z - pz;
	b = __builtin_fmaf(dz, tz, b);
	disc = __builtin_fmaf(-tz, tz, disc);
	disc = __builtin_fmaf(b, b, disc);
	if (__float_as_uint(disc) & 0x80000000u) return;
	float sq = __builtin_sqrtf(disc);
	float dist = b - sq;
	if (__float_as_uint(dist) & 0x80000000u) dist = b + sq;
	if ((dist < tfar) && !(__float_as_uint(dist) & 0x80000000u)) { tfar = dist; primID = prim; }
}
// intersect_prims_shadow, BVH.hpp:294-300 (scalar, unfused)
MIRT_DI bool sphere_occludes(float4 s, float px, float py, float pz, float dx, float dy, float dz, float tfar) {
	f3 P{ s.x - px, s.y - py, s.z - pz };
	float b = dot3(f3{ dx, dy, dz }, P);
	float disc = b * b - dot3(P, P) + s.w;
	if (disc < 0.0f) return false;
	disc = __builtin_sqrtf(disc);
	float dist = (b >= disc ? b - disc : b + disc);
	if (dist < 0.0f || dist >= tfar) return false;
	return true;
}
// ---- BVH traversal -----------------------------------------------------------------------------
// Semantics (DESIGN.md "Traversal semantics"): the BVH is a pure acceleration of the reference's shipped
// brute-force loops (BVH.hpp:312 / :365).  Boxes are conservative (bvh_layout.hpp), the ray interval is
// [0, tfar], and the closest hit is the lexicographic minimum of (dist, BVH-order prim index) — which is
// what the ascending strict-'<' scan of intersect_prims returns — so visiting order, pruning and the slab
// arithmetic cannot change the result.  oracle/oracle.cpp mode 2 is the CPU twin of this routine.
// Conservative "cone" slab test.  The reference's sphere tests assume a unit direction (disc = b^2 - |oc|^2 + r^2,
// BVH.hpp:251-260,295-296), but its tangent frame is ill-conditioned near N.z = -1 (Sampling.hpp:150-159) and to_world can
// return a stretched direction (|D| = 1.125 seen in cfg2); for such a ray a sphere it geometrically misses can still pass
// the reference's test, by a margin that grows with distance — no fixed box padding covers that.  Algebra: with d the
// reported hit parameter, |p + d*D - c|^2 = r^2 + d^2 (|D|^2 - 1) (+ rounding of the f32 sphere arithmetic, <= 2^-19 d^2),
// so the reported hit point lies within alpha*d of the sphere, alpha^2 = max(|D|^2-1, 0) + 2^-19.  The boxes are therefore
// tested against the ray inflated by alpha*t (L-inf) at parameter t:
//     lo_i - alpha t <= p_i + t d_i <= hi_i + alpha t   <=>   t (d_i+alpha) >= lo_i - p_i   and   t (d_i-alpha) <= hi_i - p_i
// i.e. the usual slab test with reciprocal 1/(d_i+alpha) for the lo plane and 1/(d_i-alpha) for the hi plane (still one FMA
// per plane).  An axis with |d_i| < alpha has d_i+alpha > 0 > d_i-alpha: BOTH planes give lower bounds on t and there is no
// upper bound (the cone widens faster than the ray drifts).  Such axes cannot simply be dropped: every camera ray near the
// image's centre row/column is nearly axis-parallel, and without its lateral bounds it visits every box in front of it
// (64 438 boxes for one ray of S(100000), a 20 ms tail per launch).  Per axis a constant c = -inf (ordinary) / +inf (both
// lower bounds) turns the two cases into the same two instructions:
//     enter_i = med3(l, h, c)   (min(l,h) | max(l,h))        leave_i = max3(l, h, c)   (max(l,h) | +inf)
// |d_i| == alpha exactly: the axis gives no bound (l = -inf, h = +inf).  For unit directions alpha = 1.4e-3: +4 % box tests on S(1000).
struct RaySlab { float iax, iay, iaz, nax, nay, naz, ibx, iby, ibz, nbx, nby, nbz, cx, cy, cz; };
MIRT_DI void slab_axis(float p, float d, float alpha, float& ia, float& na, float& ib, float& nb, float& c) {
	const float prod = (d - alpha) * (d + alpha);               // one division per axis: 1/(d+a) = (d-a)/(d^2-a^2), 1/(d-a) = (d+a)/(d^2-a^2)
	const bool keep = prod != 0.0f;
	const float r = 1.0f / prod;
	const float ra = (d - alpha) * r, rb = (d + alpha) * r;
	ia = keep ? ra : 0.0f; ib = keep ? rb : 0.0f;
	na = keep ? -(p * ra) : -__builtin_inff();
	nb = keep ? -(p * rb) : __builtin_inff();
	c = (prod < 0.0f) ? __builtin_inff() : -__builtin_inff();
}
MIRT_DI RaySlab make_slab(float px, float py, float pz, float dx, float dy, float dz, float& alpha_out, float alpha_extra = 0.0f) {
	const float L2 = (dx * dx + dy * dy) + dz * dz;
	const float a2 = __builtin_fmaxf(L2 - 1.0f, 0.0f) * 1.0009765625f + 0x1p-19f;
	const float alpha = __builtin_sqrtf(a2) * 1.0009765625f + alpha_extra;     // alpha_extra: a BUNDLE of rays around this axis (k_primary_cand)
	alpha_out = alpha;
	RaySlab s;
	slab_axis(px, dx, alpha, s.iax, s.nax, s.ibx, s.nbx, s.cx);
	slab_axis(py, dy, alpha, s.iay, s.nay, s.iby, s.nby, s.cy);
	slab_axis(pz, dz, alpha, s.iaz, s.naz, s.ibz, s.nbz, s.cz);
	return s;
}
// v_med3_f32 / v_max3_f32 / v_min3_f32; no NaN can reach them (finite boxes, finite or zeroed reciprocals).
MIRT_DI bool slab_hit(const RaySlab& rs, float lx, float hx, float ly, float hy, float lz, float hz, float tfar, float& tnear) {
	const float ex = __builtin_amdgcn_fmed3f(lx, hx, rs.cx), ey = __builtin_amdgcn_fmed3f(ly, hy, rs.cy), ez = __builtin_amdgcn_fmed3f(lz, hz, rs.cz);
	const float ox = __builtin_fmaxf(__builtin_fmaxf(lx, hx), rs.cx), oy = __builtin_fmaxf(__builtin_fmaxf(ly, hy), rs.cy), oz = __builtin_fmaxf(__builtin_fmaxf(lz, hz), rs.cz);
	const float tmin = __builtin_fmaxf(__builtin_fmaxf(ex, ey), __builtin_fmaxf(ez, 0.0f));
	float oz_far;                                                             // min(oz, tfar) as the bare instruction: __builtin_fminf first canonicalises tfar (a v_max_f32 x, x per step) in case it were a signalling NaN
	asm("v_min_f32 %0, %1, %2" : "=v"(oz_far) : "v"(oz), "v"(tfar));
	const float tmax = __builtin_fminf(__builtin_fminf(ox, oy), oz_far);
	tnear = tmin;
	return tmin <= tmax;
}
// intersect_prims acceptance made order-independent: candidate against an open tfar, then (dist, index) order.
MIRT_DI void sphere_closest_tie(float4 s, int32_t prim, float px, float py, float pz, float dx, float dy, float dz,
                                float& tfar, int32_t& primID) {
	float t = MIRT_FLT_MAX; int32_t id = -1;
	sphere_closest(s, prim, px, py, pz, dx, dy, dz, t, id);
	if (id < 0) return;
	if (t < tfar || (t == tfar && (primID < 0 || prim < primID))) { tfar = t; primID = prim; }
}

// LDS views are typed with address_space(3) so every access is a ds_read/ds_write.  (With generic pointers hipcc merged
// the "staged record" and "record in HBM" paths into one pointer select + flat_load_dwordx4, which is several times
// slower than ds_read_b128 and waits on both vmcnt and lgkmcnt.)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
// Select-only forms of the two sphere tests for the traversal loop (same arithmetic and acceptance as sphere_closest /
// sphere_occludes above; a divergent branch costs several scalar exec-mask instructions, and the CU's single scalar unit
// was issuing as many instructions as its four SIMDs).
// sqrt of the two tests below: correctly rounded, bit for bit __builtin_sqrtf (hipcc's expansion around v_sqrt_f32: one fix-up step
// towards each neighbour, decided by an FMA residual) WITHOUT that expansion's seven instructions of denormal scaling and 0 / inf
// pass-through.  Only inputs in [+0, 2^-100) need the scaling (their residuals would be denormal); a wave that holds such a
// discriminant (none in practice) takes the library path as a whole — one unsigned compare on the bits, which negative inputs
// pass.  +inf and the large values fall out right (NaN residuals select nothing); negative and NaN inputs give NaN-or-garbage that
// both callers mask.  Checked against __builtin_sqrtf for EVERY f32 bit pattern >= 0 by profiles/experiments/sqrt_check.hip, and
// by tests (mirt_debug_math fn 10).
MIRT_DI float sqrt_trav(float x) {
	if (__ballot(__float_as_uint(x) < 0x0d800000u) != 0ull) return __builtin_sqrtf(x);     // 0x0d800000 = 2^-100
	const float s = __builtin_amdgcn_sqrtf(x);                                // within 1 ulp
	const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
	const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
	float r = (r_dn <= 0.0f) ? s_dn : s;
	r = (r_up > 0.0f) ? s_up : r;
	return r;
}
MIRT_DI void sphere_closest_sel(bool lane_on, float4 s, int32_t prim, const float px, const float py, const float pz, const float dx, const float dy,
                                const float dz, float& tfar, int32_t& primID) {
	const float tx = s.x - px;
	float b = dx * tx;
	float disc = __builtin_fmaf(-tx, tx, s.w);
	const float ty = s.y - py;
	b = __builtin_fmaf(dy, ty, b);
	disc = __builtin_fmaf(-ty, ty, disc);
	const float tz = s.z - pz;
	b = __builtin_fmaf(dz, tz, b);
	disc = __builtin_fmaf(-tz, tz, disc);
	disc = __builtin_fmaf(b, b, disc);
	const bool disc_ok = !(__float_as_uint(disc) & 0x80000000u);
	const float sq = sqrt_trav(disc);                             // NaN for a negative discriminant: masked by disc_ok
	float dist = b - sq;
	dist = (__float_as_uint(dist) & 0x80000000u) ? b + sq : dist;
	// bitwise &,| on purpose: short-circuit &&,|| come back as nested exec-mask branches
	const bool cand = lane_on & disc_ok & (dist < MIRT_FLT_MAX) & !(__float_as_uint(dist) & 0x80000000u);
	const bool better = cand & ((dist < tfar) | ((dist == tfar) & ((primID < 0) | (prim < primID))));
	tfar = better ? dist : tfar;
	primID = better ? prim : primID;
}
MIRT_DI bool sphere_occludes_sel(float4 s, const float px, const float py, const float pz, const float dx, const float dy, const float dz, const float tfar) {
	const f3 P{ s.x - px, s.y - py, s.z - pz };
	const float b = dot3(f3{ dx, dy, dz }, P);
	const float disc = b * b - dot3(P, P) + s.w;
	const float sq = sqrt_trav(disc);
	const float dist = (b >= sq ? b - sq : b + sq);
	return !(disc < 0.0f) & !((dist < 0.0f) | (dist >= tfar));
}

struct TraceLds { const lds_v4f* recs; const lds_v4f* spheres; lds_u32* stack; };   // stack: [kLdsStack][blockDim.x] entries (u32, or u16 with half records)
MIRT_DI float half_lo(uint32_t w) { return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(w & 0xffffu))); }
MIRT_DI float half_hi(uint32_t w) { return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(w >> 16))); }
MIRT_DI float4 to_float4(v4f v) { return make_float4(v.x, v.y, v.z, v.w); }

// One lane's traversal state.  A lane keeps it in registers across refills of OTHER lanes (persistent waves below).
struct Trav {
	float px, py, pz, dx, dy, dz;
	RaySlab rs;
	float tfar;
	int32_t prim;
	uint32_t cur, sp;
};
// Stack entries beyond the LDS-resident ones.  Kept OUTSIDE Trav: a dynamically indexed member would pin the whole struct
// in scratch memory (every step would then reload the ray through VMEM); alone, only this rarely-touched array lives there.
struct TravSpill { uint32_t e[kStack - kLdsStackWide]; };
// Rays whose cone half-width exceeds kAlphaFat per unit of ray parameter (|D|^2 - 1 > ~1e-3: a few per million, produced
// by the reference's ill-conditioned tangent frame) are not traversed: the inflated ray would touch most of the tree and one
// lane would walk it serially (measured: 20-57 ms per launch on a 100k-sphere scene).  They go to a "fat ray" list and
// k_trace_fat intersects them with every sphere, a whole workgroup per ray — literally the reference's brute-force loop.
// (|D|^2 - 1 over the rays of S(100000), CPU twin, per million: 38 in (1e-4, 3e-4], 16 in (3e-4, 1e-3], 6 above.  With the limit
// at 0.01 the detour took the 60 per million and 1.6 % of the cfg4 step; at 0.03 it takes the 6.)
constexpr float kAlphaFat = 0.03f;
MIRT_DI bool trav_begin(Trav& t, float px, float py, float pz, float dx, float dy, float dz, float tfar, float alpha_extra = 0.0f) {     // true = fat ray
	t.px = px; t.py = py; t.pz = pz; t.dx = dx; t.dy = dy; t.dz = dz;
	float alpha;
	t.rs = make_slab(px, py, pz, dx, dy, dz, alpha, alpha_extra);
	t.tfar = tfar; t.prim = -1; t.cur = 0; t.sp = 0;
	return alpha > kAlphaFat;
}
// One step = one 64-B record: slab-test both children against the current tfar, intersect hit leaf children at once,
// re-check inner children against the shrunken tfar, enter the nearer, push the other (or pop).  Returns true when this
// ray is finished (stack empty, or ANYHIT occluder found -> occluded = true).
// MODE: kClosest (Traverse), kAnyHit (Traverse_shadow), kCollect (candidate spheres of a pixel's bundle of camera rays, below).
constexpr int kClosest = 0, kAnyHit = 1, kCollect = 2;
// kCollect — what the samples of ONE PIXEL can hit.  Within a batch a pixel is sampled once per accumulation (up to 256 times) with sub-pixel jitter
// (Renderer.hpp:117-118); all those camera rays leave cam.pos inside a cone of half-angle rho around the ray through the pixel
// centre.  Traversing that cone once (the same conservative slab test, widened by rho) and listing the spheres it can touch turns
// the primary traversal of every sample into a few exact sphere tests (k_primary_hits).  The list is complete for the closest hit:
//   * a sphere some sample reports a hit on at parameter d has its reported hit point within alpha_s d of the sphere and within
//     rho d of the axis point at d, so its box inflated by (alpha_s + rho) d meets the axis — the cone test with alpha + rho;
//   * the search is cut at F once a sphere is found that EVERY sample must hit: centre within the silhouette by a margin that
//     covers the cone's reach and the reference's f32 discriminant error, origin outside.  F bounds every sample's first-hit
//     distance on it from above, hence every sample's closest-hit distance: a sphere whose box starts beyond F cannot be a closest hit.
// A pixel whose list would exceed kCandMax entries (silhouettes of many small spheres) is marked and its samples are traced normally.
// (15 entries until round 3: 9 % of the cfg4 pixels overflowed; with 31 it is 3.8 % and the step 1 % faster — a sample of k_primary_hits reads its
//  list from registers and L1 —; 63 entries buy nothing more.)
constexpr uint32_t kCandMax = 31, kCandStride = 32, kCandOverflow = 0xffffffffu;
struct Collect { uint32_t* cand; float rho; uint32_t n_pix; };      // cand[k * n_pix + pixel]: k = 0 the count, k = 1.. up to kCandMax BVH-order prim indices (plane-major: neighbouring pixels, neighbouring words)
MIRT_DI void collect_leaf(bool on, float4 s, uint32_t prim, uint32_t pix, const Collect& col, Trav& t) {
	uint32_t cnt = static_cast<uint32_t>(t.prim);
	if (on & (cnt < kCandMax)) col.cand[static_cast<size_t>(1u + cnt) * col.n_pix + pix] = prim;
	cnt += on ? 1u : 0u;
	t.prim = static_cast<int32_t>(cnt);
	// full cover -> every sample hits this sphere no later than F
	const f3 oc{ s.x - t.px, s.y - t.py, s.z - t.pz };
	const float oc2 = dot3(oc, oc), b = dot3(f3{ t.dx, t.dy, t.dz }, oc);
	const float len = __builtin_sqrtf(oc2);
	const float reach = __builtin_sqrtf(__builtin_fmaxf(oc2 - b * b, 0.0f)) + col.rho * len;       // farthest a sample ray can pass from the centre
	// The discriminant every sample is left with after the worst rounding of the reference's f32 evaluation of b^2 - |oc|^2 + r^2
	// (15 roundings of magnitude 2^-24 |oc|^2, directions that are unit only to 1e-7, and this function's own cancellation in
	// oc2 - b*b: together < 2^-19.5 |oc|^2; 2^-17 leaves a factor 5).  Positive: the reference reports a hit for every sample, at a
	// parameter no larger than (b + rho |oc|) - sqrt(slack).
	const float slack = (s.w - reach * reach) - (0x1p-17f * oc2 + 1e-6f * s.w);
	const bool cover = on & (b > 0.0f) & (oc2 > s.w * 1.002f) & (slack > 0.0f);
	const float F = ((b + col.rho * len) - __builtin_sqrtf(__builtin_fmaxf(slack, 0.0f))) * 1.001f + 1e-4f;
	t.tfar = (cover & (F < t.tfar)) ? F : t.tfar;
}
// Per-lane stack: the first kLdsStack entries live in LDS, entry-major ([entry][thread]: a wave's accesses to one depth are
// consecutive, conflict-free); deeper entries (rare) use the scratch array.  An entry is a child reference: a record index, or
// kLeafBit | prim.  ST16 packs it into 16 bits (index < 2^15, the leaf flag moved to bit 15; read back sign-extended).
typedef __attribute__((address_space(3))) int16_t lds_i16;
template <bool HALF, bool ST16>
MIRT_DI void stack_put(const TraceLds lds, TravSpill& spill, uint32_t sp, uint32_t ref) {
	constexpr uint32_t lds_entries = (HALF && !ST16) ? kLdsStackWide : kLdsStack;
	constexpr uint32_t lstride = kTraceBlock;     // every trace launch uses kTraceBlock threads (a runtime blockDim.x costs a quarter-rate v_mul_lo_u32 per push and per pop)
	if (sp < lds_entries) { if (ST16) ((lds_u16*)lds.stack)[sp * lstride + threadIdx.x] = static_cast<uint16_t>(ref | (ref >> 16)); else lds.stack[sp * lstride + threadIdx.x] = ref; }
	else if (sp < kStack) spill.e[sp - lds_entries] = ref;       // depth < kStack is validated on the host
}
template <bool HALF, bool ST16>
MIRT_DI uint32_t stack_get(const TraceLds lds, const TravSpill& spill, uint32_t sp) {
	constexpr uint32_t lds_entries = (HALF && !ST16) ? kLdsStackWide : kLdsStack;
	constexpr uint32_t lstride = kTraceBlock;
	if (sp < lds_entries) {
		if (ST16) return static_cast<uint32_t>(static_cast<int32_t>(((lds_i16*)lds.stack)[sp * lstride + threadIdx.x])) & 0x80007fffu;
		return lds.stack[sp * lstride + threadIdx.x];
	}
	return spill.e[sp - lds_entries];
}
// A ray's traversal is a depth-first walk in which LEAVES ARE STACK ITEMS like inner nodes: t.cur is either a record (node_step:
// slab-test both children against the current tfar, enter the nearer hit child — leaf or not —, push the other, or pop) or a leaf
// (leaf_step: intersect its sphere, then pop).  The two kinds of step are separate so that a wave can run them as separate,
// DENSE passes (trace_persistent): with the sphere test inlined in the node step, as it was, ~58 % of the loop's VALU
// instructions were sphere tests executed for the whole wave on behalf of the ~13 % of lanes that had a hit leaf in that step.
// Both leave t.cur = kHalt when this ray is finished (stack empty, or ANYHIT occluder found -> occluded = true, or a full kCollect list).
constexpr uint32_t kHalt = 0x7fffffffu;         // Trav::cur of a lane that is not walking: no ray, or a finished one whose result waits for the next refill
template <int MODE, bool COUNT, bool ALL_LDS, bool HALF, bool ST16>
MIRT_DI void node_step(const SceneDev& sc, const TraceLds lds, Trav& t, TravSpill& spill, uint32_t& n_nodes) {
	const uint32_t cur = t.cur;
	// child boxes: (lo, hi) per axis for child 0 (a) and child 1 (b)
	float ax0, ax1, ay0, ay1, az0, az1, bx0, bx1, by0, by1, bz0, bz1;
	uint32_t c0, c1;
	if (HALF) {
		v4f q0, q1;                           // 32-B record: 12 binary16 planes + 2 child references
		// Only the top of a large tree is staged.  The choice is made per WAVE: a step whose lanes are all in the staged block reads LDS,
		// any other step reads every lane's record from memory (the top records are the hottest lines of L1 / L2) — one path per step
		// instead of two exec-masked halves.
		if (ALL_LDS || cur < sc.lds_recs) { const lds_v4f* r = lds.recs + cur; q0 = r[0]; q1 = r[sc.lds_recs]; }     // plane-major in LDS
		else { const v4f* r = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sc.recs) + (cur << 5)); q0 = r[0]; q1 = r[1]; }   // 32-bit byte offset from a uniform base (n_recs < 2^27, checked on the host)
		const uint32_t w0 = __float_as_uint(q0.x), w1 = __float_as_uint(q0.y), w2 = __float_as_uint(q0.z), w3 = __float_as_uint(q0.w);
		const uint32_t w4 = __float_as_uint(q1.x), w5 = __float_as_uint(q1.y);
		ax0 = half_lo(w0); bx0 = half_hi(w0); ax1 = half_lo(w1); bx1 = half_hi(w1);
		ay0 = half_lo(w2); by0 = half_hi(w2); ay1 = half_lo(w3); by1 = half_hi(w3);
		az0 = half_lo(w4); bz0 = half_hi(w4); az1 = half_lo(w5); bz1 = half_hi(w5);
		c0 = __float_as_uint(q1.z); c1 = __float_as_uint(q1.w);
	} else {
		v4f q0, q1, q2, q3;
		if (ALL_LDS || __ballot(cur >= sc.lds_recs) == 0ull) {     // staged records are stored plane-major (q0[], q1[], q2[], q3[]): 16-B stride between lanes' addresses
			const lds_v4f* r = lds.recs + cur; const uint32_t ns = sc.lds_recs;
			q0 = r[0]; q1 = r[ns]; q2 = r[2u * ns]; q3 = r[3u * ns];
		}
		else { const v4f* r = reinterpret_cast<const v4f*>(sc.recs) + 4ull * cur; q0 = r[0]; q1 = r[1]; q2 = r[2]; q3 = r[3]; }
		ax0 = q0.x; bx0 = q0.y; ax1 = q0.z; bx1 = q0.w;
		ay0 = q1.x; by0 = q1.y; ay1 = q1.z; by1 = q1.w;
		az0 = q2.x; bz0 = q2.y; az1 = q2.z; bz1 = q2.w;
		c0 = __float_as_uint(q3.x); c1 = __float_as_uint(q3.y);
	}
	if (COUNT) n_nodes += 2;
	const RaySlab& rs = t.rs;
	float ta, tb;
	const bool ha = slab_hit(rs, __builtin_fmaf(ax0, rs.iax, rs.nax), __builtin_fmaf(ax1, rs.ibx, rs.nbx),
	                         __builtin_fmaf(ay0, rs.iay, rs.nay), __builtin_fmaf(ay1, rs.iby, rs.nby),
	                         __builtin_fmaf(az0, rs.iaz, rs.naz), __builtin_fmaf(az1, rs.ibz, rs.nbz), t.tfar, ta);
	const bool hb = slab_hit(rs, __builtin_fmaf(bx0, rs.iax, rs.nax), __builtin_fmaf(bx1, rs.ibx, rs.nbx),
	                         __builtin_fmaf(by0, rs.iay, rs.nay), __builtin_fmaf(by1, rs.iby, rs.nby),
	                         __builtin_fmaf(bz0, rs.iaz, rs.naz), __builtin_fmaf(bz1, rs.ibz, rs.nbz), t.tfar, tb);
	// ---- next item: selects, plus one exec region each for the conditional stack write and read ----
	const bool both = ha && hb, none = !ha && !hb;
	const bool a_first = ta <= tb;                                            // any-hit rays too: an occluder is most likely close to the origin (the result does not depend on the order)
	const uint32_t near = (ha && (a_first || !hb)) ? c0 : c1;                 // the child entered when at least one child is hit
	const uint32_t far = a_first ? c1 : c0;
	uint32_t sp = t.sp;
	if (both) { stack_put<HALF, ST16>(lds, spill, sp, far); sp += 1u; }
	uint32_t next = none ? kHalt : near;                                     // nothing hit and nothing stacked: finished
	if (none & (sp != 0u)) { --sp; next = stack_get<HALF, ST16>(lds, spill, sp); }
	t.sp = sp;
	t.cur = next;
}
// The node pass over 4-wide binary16 records (bvh_layout.hpp build_wide_half_records): one 64-B fetch, four slab tests, the nearest
// hit child next (the lowest slot on ties), the other hit children pushed in slot order — sorting them by distance as well buys 2 %
// fewer visits on S(100000) for three times the selection code (oracle twin, orc_wide_stats 1 vs 2).  Unused slots never hit.
template <int MODE, bool COUNT, bool ALL_LDS, bool ST16>
MIRT_DI void node_step_wide(const SceneDev& sc, const TraceLds lds, Trav& t, TravSpill& spill, uint32_t& n_nodes) {
	const uint32_t cur = t.cur;
	v4f q0, q1, q2, q3;
	if (ALL_LDS || cur < sc.lds_recs) { const lds_v4f* r = lds.recs + cur; const uint32_t ns = sc.lds_recs; q0 = r[0]; q1 = r[ns]; q2 = r[2u * ns]; q3 = r[3u * ns]; }
	else { const v4f* r = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(sc.recs) + (cur << 6)); q0 = r[0]; q1 = r[1]; q2 = r[2]; q3 = r[3]; }
	const uint32_t x0 = __float_as_uint(q0.x), x1 = __float_as_uint(q0.y), x2 = __float_as_uint(q0.z), x3 = __float_as_uint(q0.w);
	const uint32_t y0 = __float_as_uint(q1.x), y1 = __float_as_uint(q1.y), y2 = __float_as_uint(q1.z), y3 = __float_as_uint(q1.w);
	const uint32_t z0 = __float_as_uint(q2.x), z1 = __float_as_uint(q2.y), z2 = __float_as_uint(q2.z), z3 = __float_as_uint(q2.w);
	const uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y), r2 = __float_as_uint(q3.z), r3 = __float_as_uint(q3.w);
	if (COUNT) {                                                              // boxes tested = the slots in use (an unused slot's lo.x is +inf)
		n_nodes += ((x0 & 0x7fffu) != 0x7c00u) + (((x0 >> 16) & 0x7fffu) != 0x7c00u) + ((x1 & 0x7fffu) != 0x7c00u) + (((x1 >> 16) & 0x7fffu) != 0x7c00u);
	}
	const RaySlab& rs = t.rs;
	float ta, tb, tc, td;
	const bool ha = slab_hit(rs, __builtin_fmaf(half_lo(x0), rs.iax, rs.nax), __builtin_fmaf(half_lo(x2), rs.ibx, rs.nbx),
	                         __builtin_fmaf(half_lo(y0), rs.iay, rs.nay), __builtin_fmaf(half_lo(y2), rs.iby, rs.nby),
	                         __builtin_fmaf(half_lo(z0), rs.iaz, rs.naz), __builtin_fmaf(half_lo(z2), rs.ibz, rs.nbz), t.tfar, ta);
	const bool hb = slab_hit(rs, __builtin_fmaf(half_hi(x0), rs.iax, rs.nax), __builtin_fmaf(half_hi(x2), rs.ibx, rs.nbx),
	                         __builtin_fmaf(half_hi(y0), rs.iay, rs.nay), __builtin_fmaf(half_hi(y2), rs.iby, rs.nby),
	                         __builtin_fmaf(half_hi(z0), rs.iaz, rs.naz), __builtin_fmaf(half_hi(z2), rs.ibz, rs.nbz), t.tfar, tb);
	const bool hc = slab_hit(rs, __builtin_fmaf(half_lo(x1), rs.iax, rs.nax), __builtin_fmaf(half_lo(x3), rs.ibx, rs.nbx),
	                         __builtin_fmaf(half_lo(y1), rs.iay, rs.nay), __builtin_fmaf(half_lo(y3), rs.iby, rs.nby),
	                         __builtin_fmaf(half_lo(z1), rs.iaz, rs.naz), __builtin_fmaf(half_lo(z3), rs.ibz, rs.nbz), t.tfar, tc);
	const bool hd = slab_hit(rs, __builtin_fmaf(half_hi(x1), rs.iax, rs.nax), __builtin_fmaf(half_hi(x3), rs.ibx, rs.nbx),
	                         __builtin_fmaf(half_hi(y1), rs.iay, rs.nay), __builtin_fmaf(half_hi(y3), rs.iby, rs.nby),
	                         __builtin_fmaf(half_hi(z1), rs.iaz, rs.naz), __builtin_fmaf(half_hi(z3), rs.ibz, rs.nbz), t.tfar, td);
	// ---- the nearest hit child (lowest slot on ties) is next; the other hit children are pushed in slot order ----
	const float inf = __builtin_inff();
	ta = ha ? ta : inf; tb = hb ? tb : inf; tc = hc ? tc : inf; td = hd ? td : inf;
	const bool b_ab = tb < ta, d_cd = td < tc;                                // within each pair the later slot wins only when strictly nearer
	const float tab = b_ab ? tb : ta, tcd = d_cd ? td : tc;
	const uint32_t rab = b_ab ? r1 : r0, rcd = d_cd ? r3 : r2;
	const bool cd = tcd < tab;
	const uint32_t near = cd ? rcd : rab;
	const bool any = ha || hb || hc || hd;
	uint32_t sp = t.sp;
	constexpr uint32_t lds_entries = ST16 ? kLdsStack : kLdsStackWide;
	if (__ballot(sp + 3u > lds_entries) == 0ull) {
		// the usual case, decided for the wave: whatever a lane pushes stays within its LDS entries — four plain conditional writes
		// instead of four times the LDS / scratch / overflow nest of stack_put
		constexpr uint32_t lstride = kTraceBlock;
		if (ha && (cd || b_ab)) { if (ST16) ((lds_u16*)lds.stack)[sp * lstride + threadIdx.x] = static_cast<uint16_t>(r0 | (r0 >> 16)); else lds.stack[sp * lstride + threadIdx.x] = r0; sp += 1u; }
		if (hb && (cd || !b_ab)) { if (ST16) ((lds_u16*)lds.stack)[sp * lstride + threadIdx.x] = static_cast<uint16_t>(r1 | (r1 >> 16)); else lds.stack[sp * lstride + threadIdx.x] = r1; sp += 1u; }
		if (hc && (!cd || d_cd)) { if (ST16) ((lds_u16*)lds.stack)[sp * lstride + threadIdx.x] = static_cast<uint16_t>(r2 | (r2 >> 16)); else lds.stack[sp * lstride + threadIdx.x] = r2; sp += 1u; }
		if (hd && (!cd || !d_cd)) { if (ST16) ((lds_u16*)lds.stack)[sp * lstride + threadIdx.x] = static_cast<uint16_t>(r3 | (r3 >> 16)); else lds.stack[sp * lstride + threadIdx.x] = r3; sp += 1u; }
	} else {
		if (ha && (cd || b_ab)) { stack_put<true, ST16>(lds, spill, sp, r0); sp += 1u; }
		if (hb && (cd || !b_ab)) { stack_put<true, ST16>(lds, spill, sp, r1); sp += 1u; }
		if (hc && (!cd || d_cd)) { stack_put<true, ST16>(lds, spill, sp, r2); sp += 1u; }
		if (hd && (!cd || !d_cd)) { stack_put<true, ST16>(lds, spill, sp, r3); sp += 1u; }
	}
	uint32_t next = any ? near : kHalt;                                       // nothing hit and nothing stacked: finished
	if (!any & (sp != 0u)) { --sp; next = stack_get<true, ST16>(lds, spill, sp); }
	t.sp = sp;
	t.cur = next;
}
template <int MODE, bool COUNT, bool ALL_LDS, bool HALF, bool ST16>
MIRT_DI void leaf_step(const SceneDev& sc, const TraceLds lds, Trav& t, TravSpill& spill, bool& occluded, uint32_t& n_spheres, uint32_t pix, const Collect& col) {
	const uint32_t first = t.cur & ~kLeafBit;
	// (Requesting the sphere when the lane ARRIVES at the leaf, into four registers kept until the pass, was measured: 22 VGPR
	// spills and k_trace 253 -> 283 ms per cfg4 step.  The other seven waves of the SIMD cover this load.)
	float4 s;
	if (ALL_LDS) s = to_float4(lds.spheres[first]); else s = sc.spheres[first];     // spheres are staged all (ALL_LDS) or none
	if (COUNT) n_spheres += 1u;
	bool fin = false;
	if (MODE == kAnyHit) { occluded = sphere_occludes_sel(s, t.px, t.py, t.pz, t.dx, t.dy, t.dz, t.tfar); fin = occluded; }
	else if (MODE == kCollect) { collect_leaf(true, s, first, pix, col, t); fin = static_cast<uint32_t>(t.prim) > kCandMax; }   // a full list: the pixel falls back to tracing
	else sphere_closest_sel(true, s, static_cast<int32_t>(first), t.px, t.py, t.pz, t.dx, t.dy, t.dz, t.tfar, t.prim);
	uint32_t sp = t.sp;
	t.cur = kHalt;
	if (!fin & (sp != 0u)) { --sp; t.cur = stack_get<HALF, ST16>(lds, spill, sp); }
	t.sp = sp;
}

// ---- persistent waves with in-kernel lane refill ------------------------------------------------------------
// Secondary and shadow rays have a heavy-tailed traversal length: with one fixed ray per lane, PMC showed 10-24 % VALU
// lane utilisation (one long ray keeps 63 finished lanes waiting).  Instead every wave keeps a window of ray indices
// [wbeg, wend) that it reserves from a per-launch work counter (one atomic per kChunk rays); whenever at least
// kRefillIdle lanes are idle they are given the next rays of the window — slot = wbeg + rank among idle lanes, from a
// wave64 ballot + mbcnt prefix sum — and the wave goes back to stepping all lanes together.
constexpr uint32_t kChunkMax = 512;    // default of SceneDev::chunk_max.  Measured (k_trace ms per step, 256 / 512 / 4096): cfg2 14.9 / 13.6 / 14.1, cfg4 - / 381.5 / 402.3 (larger chunks: uneven tails)
constexpr uint32_t kLeafBatch = 16;    // default of SceneDev::leaf_batch.  Measured (k_trace ms per cfg4 step; 8 / 16 / 24 / 32 / 40): 252.9 / 247.7 / 253.1 / 274.2 / 314.8; before the split 284.9
constexpr uint32_t kRefillIdle = 32;   // measured on cfg2: 8 -> 15.3 ms of trace per step, 16 -> 13.9, 24..40 -> 13.5 (a refill runs the ~200-instruction ray set-up on the whole wave)
constexpr uint32_t kNone = 0xffffffffu;
struct FatList { uint32_t* count; uint32_t* rays; uint32_t capacity; };
// [beg, end): ray numbers the wave still has to hand out, all in one queue segment (slot = number + offset); [end, chunk_end):
// the rest of the reserved chunk, in later segments.
struct WaveWindow { uint32_t beg, end, chunk_end, offset, chunk; bool more; };
// Rays per reservation: large enough that the work counter sees one atomic per a few thousand rays, small enough that
// every wave of the grid gets a few chunks even on the thin late-bounce streams.
MIRT_DI uint32_t pick_chunk(uint32_t n, uint32_t chunk_max) {
	const uint32_t waves = gridDim.x * (blockDim.x >> 6);
	uint32_t c = n / (waves * 4u);
	c = (c + 63u) & ~63u;
	return c < 64u ? 64u : (c > chunk_max ? chunk_max : c);
}
// Gives the lanes with `want` the slots of the next rays of the wave's window (number = beg + rank among wanting lanes, from a
// wave64 ballot + mbcnt prefix sum), reserving a new chunk from the launch's work counter when the window is empty.
MIRT_DI uint32_t wave_take(bool want, WaveWindow& w, const Queue& q, uint32_t n, uint32_t* work_next) {
	const unsigned long long m = __ballot(want);
	const uint32_t n_want = static_cast<uint32_t>(__popcll(m));
	if (n_want == 0u) return kNone;
	if (w.beg == w.end) {
		if (w.end == w.chunk_end && w.more) {
			uint32_t base = 0;
			if (lane_id() == 0) base = atomicAdd(work_next, w.chunk);
			base = __builtin_amdgcn_readfirstlane(base);
			if (base >= n) { w.more = false; w.beg = w.end = w.chunk_end = 0; }
			else { w.beg = w.end = base; w.chunk_end = min(base + w.chunk, n); }
		}
		if (w.end != w.chunk_end) {                                  // next piece of the chunk: the part that lies in one segment
			const QueuePiece p = queue_piece(q, n, w.beg);
			w.end = min(w.chunk_end, p.end); w.offset = p.offset;
		}
	}
	const uint32_t take = min(n_want, w.end - w.beg);
	uint32_t got = kNone;
	if (want) { const uint32_t r = mask_rank(m); if (r < take) got = w.beg + r + w.offset; }
	w.beg += take;
	return got;
}

// Persistent wave loop shared by the closest-hit and the any-hit kernels.
//   * a lane is WALKING (t.cur is a record or a leaf), DONE (ri != kNone, t.cur == kHalt: result waiting to be flushed) or EMPTY (ri == kNone);
//   * a refill event (>= kRefillIdle lanes not running) flushes the finished lanes' results, hands the idle lanes the next
//     ray indices of the window and loads + sets up their rays — all as batched, mostly coalesced accesses.  (An earlier
//     version also kept one prefetched ray per lane in registers; with the cone slab constants that pushed the kernel past
//     64 VGPRs, i.e. from two 16-wave workgroups per CU to one, which cost far more than the prefetch saved.)
template <int MODE, bool COUNT, bool ALL_LDS, bool HALF, bool ST16, bool WIDE, class LoadRay, class StoreResult>
MIRT_DI void trace_persistent(const SceneDev& sc, const TraceLds tl, const Queue& q, uint32_t n, uint32_t* work_next, FatList fat, uint32_t& c_nodes, uint32_t& c_spheres,
                              LoadRay load_ray, StoreResult store_result, const Collect col = Collect{ nullptr, 0.0f, 0u }) {
	WaveWindow w{ 0, 0, 0, 0, pick_chunk(n, sc.chunk_max), true };
	Trav t;
	TravSpill spill;
	uint32_t ri = kNone;
	bool occluded = false;
	t.cur = kHalt;
	for (;;) {
		// ---- refill event: flush results, hand the idle lanes the next ray indices of the window, load and set up their rays ----
		if ((ri != kNone) & (t.cur == kHalt)) { store_result(ri, t, occluded); ri = kNone; }
		{
			const uint32_t got = wave_take(ri == kNone, w, q, n, work_next);   // a slot of the stream planes
			if (got != kNone) {
				float px, py, pz, dx, dy, dz, tf;
				load_ray(got, px, py, pz, dx, dy, dz, tf);
				ri = got; occluded = false;
				if (MODE == kCollect) {
					if (trav_begin(t, px, py, pz, dx, dy, dz, tf, col.rho)) { t.prim = static_cast<int32_t>(kCandMax + 1u); t.cur = kHalt; }   // a bundle too wide for the tree: the pixel is traced normally
					else t.prim = 0;                                           // candidates listed so far
				} else if (trav_begin(t, px, py, pz, dx, dy, dz, tf)) {
					const uint32_t k = atomicAdd(fat.count, 1u);               // rare: a few rays per million
					if (k < fat.capacity) { fat.rays[k] = got; ri = kNone; t.cur = kHalt; }   // list full -> traverse it after all (correct, only slow)
				}
			}
		}
		const bool work_left = w.more || w.beg != w.chunk_end;
		if (__ballot(ri != kNone) == 0ull) { if (!work_left) break; continue; }
		const bool can_refill = work_left;
		// ---- step every walking lane until enough lanes have finished to make the next refill worthwhile ----
		// A lane's state is its t.cur: a record index (at a node), kLeafBit | prim (at a leaf) or kHalt.  One pass per iteration, chosen
		// for the whole wave: a LEAF pass (the lanes standing at a leaf intersect their sphere) once at least sc.leaf_batch lanes wait
		// for one or no lane stands at a record, a NODE pass otherwise.  A waiting lane loses the node passes it sits out; the sphere
		// test (~50 VALU instructions with its correctly rounded sqrt) runs at several times the lane density it had inside the node step.
		// (Measured and dropped: REQUESTING THE NEXT RECORD EARLY — as soon as a lane knows its next node, into two register quads
		// live across the passes: 51 VGPR spills around the loop and k_trace 249 -> 276 ms per cfg4 step; the other seven waves of
		// the SIMD already cover the load at the head of each pass.)
		// (Measured and dropped: the same request as an LDS-DMA load — global_load_lds_dwordx4 into one 16-B LDS slot per lane when the
		// lane ARRIVES at a leaf, read by the leaf pass after s_waitcnt vmcnt(0); no registers, but 16 KB less of staged records:
		// k_trace 226 -> 242 ms per cfg4 step.  Halving the resident waves costs 1.43x (227 -> 325 ms), so the loop is neither purely
		// latency-bound nor purely issue-bound.)
		// (Measured and dropped: SPECULATION — a lane parks the leaf it meets in a register and walks on until the wave's next leaf pass,
		// so that node passes keep ~0.68 instead of ~0.61 of their lanes and leaf passes fill up.  The box tests made against the stale
		// tfar cost more than that gains: 46 instead of 41 VALU wave-instructions per ray, k_trace 267 instead of 248 ms per cfg4 step.)
		for (;;) {
			const bool at_leaf = static_cast<int32_t>(t.cur) < 0, at_node = t.cur < kHalt;      // kLeafBit is the sign bit
			const unsigned long long leaf_m = __ballot(at_leaf), node_m = __ballot(at_node);
			const unsigned long long walking = leaf_m | node_m;
			if (walking == 0ull) break;
			if (can_refill && 64u - static_cast<uint32_t>(__popcll(walking)) >= sc.refill_idle) break;
			if (node_m == 0ull || static_cast<uint32_t>(__popcll(leaf_m)) >= sc.leaf_batch) {
				if (at_leaf) leaf_step<MODE, COUNT, ALL_LDS, HALF, ST16>(sc, tl, t, spill, occluded, c_spheres, ri, col);
			} else {
				if (at_node) { if (WIDE) node_step_wide<MODE, COUNT, ALL_LDS, ST16>(sc, tl, t, spill, c_nodes); else node_step<MODE, COUNT, ALL_LDS, HALF, ST16>(sc, tl, t, spill, c_nodes); }
			}
		}
	}
}

// Brute force over all prims (the reference as shipped, BVH.hpp:312 / :365), sphere packets staged
// through LDS one chunk at a time by the whole workgroup.
constexpr uint32_t kBruteChunk = 1024;   // 16 KiB of float4
template <bool ANYHIT, bool COUNT>
MIRT_DI bool traverse_brute(const SceneDev& sc, float4* lds, bool active, float px, float py, float pz, float dx, float dy, float dz,
                            float& tfar, int32_t& primID, uint32_t& n_spheres) {
	bool occluded = false;
	for (uint32_t base = 0; base < sc.n_spheres; base += kBruteChunk) {
		const uint32_t cnt = min(kBruteChunk, sc.n_spheres - base);
		__syncthreads();
		for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x) lds[j] = sc.spheres[base + j];
		__syncthreads();
		if (active && !(ANYHIT && occluded)) {
			for (uint32_t j = 0; j < cnt; j++) {
				if (COUNT) n_spheres++;
				const float4 s = lds[j];
				if (ANYHIT) { if (sphere_occludes(s, px, py, pz, dx, dy, dz, tfar)) { occluded = true; break; } }
				else sphere_closest(s, static_cast<int32_t>(base + j), px, py, pz, dx, dy, dz, tfar, primID);
			}
		}
	}
	return occluded;
}

// Dynamic LDS of the trace kernels: [records 0..lds_recs) | spheres 0..lds_spheres) | stack kLdsStack x blockDim] for
// the BVH path, or one kBruteChunk sphere buffer for the brute-force path.
MIRT_DI TraceLds stage_bvh(const SceneDev& sc, float4* lds_generic) {
	lds_v4f* lds = (lds_v4f*)lds_generic;
	const v4f* recs = reinterpret_cast<const v4f*>(sc.recs);
	const v4f* sph = reinterpret_cast<const v4f*>(sc.spheres);
	const uint32_t planes = (sc.half_boxes && !sc.wide) ? 2u : 4u;   // float4 per record
	const uint32_t nq = sc.lds_recs * planes;
	if (planes == 2u) { for (uint32_t j = threadIdx.x; j < nq; j += blockDim.x) lds[(j & 1u) * sc.lds_recs + (j >> 1)] = recs[j]; }
	else { for (uint32_t j = threadIdx.x; j < nq; j += blockDim.x) lds[(j & 3u) * sc.lds_recs + (j >> 2)] = recs[j]; }   // AoS in HBM -> plane-major in LDS
	for (uint32_t j = threadIdx.x; j < sc.lds_spheres; j += blockDim.x) lds[nq + j] = sph[j];
	__syncthreads();
	return TraceLds{ lds, lds + nq, (lds_u32*)(lds + nq + sc.lds_spheres) };
}
MIRT_DI bool bvh_all_in_lds(const SceneDev& sc) { return sc.lds_recs == sc.n_recs && sc.lds_spheres == sc.n_spheres; }

// ------------------------------------------------------------------------------------------------
// RAY GENERATION — Renderer.hpp:97-127 (stream init is implicit: radiance 0 / throughput 1 are
// supplied by k_shade<FIRST>, so only p, dir and the path id are written)
// ------------------------------------------------------------------------------------------------
// n / d for n < 2^31 and quotients below 2^20 (slots, tile rows, tile columns) with a precomputed 1/d: the float quotient is
// within one of the truth, one correction step either way makes it exact (hipcc's u32 division is ~40 instructions).
MIRT_DI uint32_t udiv_f(uint32_t n, uint32_t d, float inv_d, uint32_t& rem) {
	uint32_t q = static_cast<uint32_t>(static_cast<float>(n) * inv_d);
	int32_t r = static_cast<int32_t>(n - q * d);
	if (r < 0) { q -= 1u; r += static_cast<int32_t>(d); }
	else if (r >= static_cast<int32_t>(d)) { q += 1u; r -= static_cast<int32_t>(d); }
	rem = static_cast<uint32_t>(r);
	return q;
}
// Global LaunchIndex (Renderer.hpp:75,84-88) of this context's local tile.
MIRT_DI uint32_t global_tile(uint32_t first_tile, uint32_t run_tiles, uint32_t stride_tiles, uint32_t local_tile) {     // k_resolve (once per pixel and frame)
	if (stride_tiles == 0u) return first_tile + local_tile;
	const uint32_t run = local_tile / run_tiles;
	return first_tile + run * stride_tiles + (local_tile - run * run_tiles);
}
MIRT_DI uint32_t global_tile(const FrameParams& fp, uint32_t local_tile) {
	if (fp.stride_tiles == 0u) return fp.first_tile + local_tile;             // wave-uniform branch
	uint32_t in_run;
	const uint32_t run = udiv_f(local_tile, fp.run_tiles, fp.inv_run_tiles, in_run);
	return fp.first_tile + run * fp.stride_tiles + in_run;
}
MIRT_DI uint32_t path_seed(const FrameParams& fp, uint32_t pix) {      // seed[ID], Renderer.hpp:107 (wraps like the int32 cast)
	return (global_tile(fp, pix >> 8) * kTileSize + (pix & 255u)) * (fp.max_bounces * 2u + 1u);
}
// RAY GENERATION, Renderer.hpp:97-127, for stream index i = slot * n_pix + pix of a batch: path id and direction (the origin
// is cam.pos).  Bounce 0 has no ray stream in HBM: k_trace<PRIMARY> and k_shade<FIRST> both derive the ray from its index
// (~90 VALU instructions each, against 28 B written + 52 B read per primary ray).
MIRT_DI void pixel_xy(const FrameParams& fp, uint32_t pix, uint32_t& tile, int32_t& x, int32_t& y) {     // Renderer.hpp:84-88,115-116
	tile = global_tile(fp, pix >> 8);
	const uint32_t ID = pix & 255u;
	uint32_t col;
	const uint32_t row = udiv_f(tile, fp.h_tiles, fp.inv_h_tiles, col);
	x = static_cast<int32_t>(kTileRoot * col + (ID & 15u));
	y = static_cast<int32_t>(kTileRoot * row + (ID >> 4));
}
MIRT_DI void primary_ray(const FrameParams& fp, uint32_t i, uint32_t& path, float& dx, float& dy, float& dz) {
	uint32_t pix, tile;
	const uint32_t slot = udiv_f(i, fp.n_pix, fp.inv_n_pix, pix);
	int32_t x, y;
	pixel_xy(fp, pix, tile, x, y);
	const uint32_t ID = pix & 255u;
	const uint32_t acc = fp.acc_base + slot + 1u;                           // ++accumulations, Renderer.hpp:74
	uint32_t rng = hash_2d(acc, (tile * kTileSize + ID) * (fp.max_bounces * 2u + 1u));
	const float s0 = rand_unit_float(rng);
	const float s1 = rand_unit_float(rng);
	const f3 d = camera_ray_dir(fp.cam, x, y, s0, s1);
	path = (slot << fp.pix_bits) | pix;
	dx = d.x; dy = d.y; dz = d.z;
}
// The same rays written out as a stream (mirt_debug_raygen only).
__global__ __launch_bounds__(kBlock) void k_raygen(FrameParams fp, StreamBuf out, uint32_t* stream_count) {
	const uint32_t total = fp.n_pix * fp.batch_n;
	if (blockIdx.x == 0 && threadIdx.x == 0) stream_count[0] = total;
	for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
		uint32_t path; float dx, dy, dz;
		primary_ray(fp, i, path, dx, dy, dz);
		out.px[i] = fp.cam.pos[0]; out.py[i] = fp.cam.pos[1]; out.pz[i] = fp.cam.pos[2];
		out.dx[i] = dx; out.dy[i] = dy; out.dz[i] = dz;
		out.path[i] = path;
	}
}

// ------------------------------------------------------------------------------------------------
// INTERSECTION — Traverse, BVH.hpp:309-360
// ------------------------------------------------------------------------------------------------
// Drain one ray queue with the persistent-wave loop (dispatch on the staged-BVH variant).
template <int MODE, bool COUNT, class LoadRay, class StoreResult>
MIRT_DI void trace_queue(const SceneDev& sc, const TraceLds tl, const Queue& q, uint32_t n, uint32_t* work_next, FatList fat, uint32_t& c_nodes, uint32_t& c_spheres,
                         LoadRay load_ray, StoreResult store_result, const Collect col = Collect{ nullptr, 0.0f, 0u }) {
	if (n == 0) return;
	const bool all = bvh_all_in_lds(sc);
	if (sc.half_boxes && sc.wide) {
		if (!sc.stack16) trace_persistent<MODE, COUNT, false, true, false, true>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);   // > 32768 records or spheres: never all in LDS
		else if (all) trace_persistent<MODE, COUNT, true, true, true, true>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);
		else trace_persistent<MODE, COUNT, false, true, true, true>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);
	} else if (sc.half_boxes) {                                              // binary 32-B records: trees too deep for the wide layout
		if (!sc.stack16) trace_persistent<MODE, COUNT, false, true, false, false>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);
		else if (all) trace_persistent<MODE, COUNT, true, true, true, false>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);
		else trace_persistent<MODE, COUNT, false, true, true, false>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);
	} else {
		if (all) trace_persistent<MODE, COUNT, true, false, false, false>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);
		else trace_persistent<MODE, COUNT, false, false, false, false>(sc, tl, q, n, work_next, fat, c_nodes, c_spheres, load_ray, store_result, col);
	}
}

// ------------------------------------------------------------------------------------------------
// Accumulator addressing + the deferred shadow-ray adds
// ------------------------------------------------------------------------------------------------
MIRT_DI size_t accum_index(uint32_t acc_base, uint32_t buckets, uint32_t pix_bits, uint32_t path) {      // path: below 2^30 (no flag bits)
	const uint32_t slot = path >> pix_bits;
	const uint32_t pix = path & ((1u << pix_bits) - 1u);
	const uint32_t bucket = (acc_base + slot + 1u) % buckets;               // Renderer.hpp:82
	return (static_cast<size_t>(pix >> 8) * buckets + bucket) * 3u * kTileSize + (pix & 255u);
}
MIRT_DI size_t accum_index(const FrameParams& fp, uint32_t path) { return accum_index(fp.idx_base, fp.idx_buckets, fp.pix_bits, path); }
MIRT_DI void accumulate_add(float* __restrict__ accum, size_t idx, float r, float g, float b) {     // Renderer.hpp:427-429
	// (pixel, bucket) is unique within a batch and batches are stream-ordered: plain read-modify-write, no atomics,
	// and each bucket sees its adds in accumulation order exactly like the reference.
	accum[idx] += r; accum[idx + kTileSize] += g; accum[idx + 2 * kTileSize] += b;
}
// Where a finished shadow ray's radiance goes.  The adds that had to wait for the occlusion test — (R + unoccluded NEE) +
// emissive, the reference's order (Renderer.hpp:307-311, then 339-341 / 348-350) — are made by the lane that traced the
// ray, straight into the next stream's radiance planes or the accumulator: k_trace is VALU-bound, so the few memory
// instructions per shadow ray ride along for free, where a separate pass over the shadow stream cost 4.8 ms per cfg2 step.
//   * light record (the common case: the hit was not emissive): k_shade has already put R where the result belongs — the
//     surviving ray's radiance word, or the path's own word of the zeroed contribution buffer — so an occluded ray (most of
//     them) touches nothing and an unoccluded one adds its NEE radiance to that word: R + S, and the emissive term is +0;
//   * kDestFull record (emissive hit, or a path that ended while batches add straight into the accumulator, whose words hold
//     earlier samples): R and E travel in the record and (R + S) + E is formed here.
// occ != nullptr (mirt_debug_trace_shadow): only the occlusion flag is stored.
constexpr uint32_t kDestFull = 0x40000000u;
constexpr uint32_t kDestSlot = 0x3fffffffu;     // stream slots stay below 2^30 (capacity check in mirt_capi.hip); path ids use bits 0-29 too (batch slot << pix_bits | pixel)
struct ShadowSink {
	float *rr, *rg, *rb;        // radiance planes of the stream k_shade reads next
	const float *px, *py, *pz;  // its origin planes (shared with the shadow rays of the surviving paths)
	float* accum;
	uint32_t acc_base, buckets, pix_bits;
	uint32_t* occ;
};
MIRT_DI void shadow_origin(const ShadowBuf& sh, const ShadowSink& sink, uint32_t i, float& px, float& py, float& pz) {
	if (sink.occ) { px = sh.px[i]; py = sh.py[i]; pz = sh.pz[i]; return; }        // stage-level entry point: plain ray list
	const uint32_t dw = sh.dest[i];
	if (dw & kDestAccum) { px = sh.px[i]; py = sh.py[i]; pz = sh.pz[i]; }
	else { const uint32_t d = dw & kDestSlot; px = sink.px[d]; py = sink.py[d]; pz = sink.pz[d]; }
}
MIRT_DI void shadow_finish(const ShadowBuf& sh, const ShadowSink& sink, uint32_t i, bool occluded, uint32_t& c_term) {
	if (sink.occ) { sink.occ[i] = occluded ? 1u : 0u; return; }
	const uint32_t dw = sh.dest[i];
	const uint32_t dest = dw & (kDestSlot | kDestAccum);
	if (dw & kDestAccum) c_term++;
	if (dw & kDestFull) {
		f3 R{ sh.rr[i], sh.rg[i], sh.rb[i] };
		const f3 E{ sh.er[i], sh.eg[i], sh.eb[i] };
		if (!occluded) { R.x += sh.sr[i]; R.y += sh.sg[i]; R.z += sh.sb[i]; }
		R.x += E.x; R.y += E.y; R.z += E.z;
		if (dest & kDestAccum) accumulate_add(sink.accum, accum_index(sink.acc_base, sink.buckets, sink.pix_bits, dest & ~kDestAccum), R.x, R.y, R.z);
		else { sink.rr[dest] = R.x; sink.rg[dest] = R.y; sink.rb[dest] = R.z; }
	} else if (!occluded) {
		const f3 S{ sh.sr[i], sh.sg[i], sh.sb[i] };
		if (dest & kDestAccum) accumulate_add(sink.accum, accum_index(sink.acc_base, sink.buckets, sink.pix_bits, dest & ~kDestAccum), S.x, S.y, S.z);   // word = R + S
		else { sink.rr[dest] += S.x; sink.rg[dest] += S.y; sink.rb[dest] += S.z; }
	}
}

// INTERSECTION + SHADOW RAY TRACING in one launch: Traverse (BVH.hpp:309-360) for the rays of bounce b and
// Traverse_shadow (BVH.hpp:362-404) for the NEE rays emitted at bounce b-1 — the two queues are independent, and every
// launch ends with a tail while its longest rays finish (~180 us at 16k+ rays, measured), so draining both in one
// persistent kernel halves the number of tails: a wave that runs out of closest-hit rays moves on to the shadow queue.
// Either count pointer may refer to a zero word (first bounce: no shadow rays yet; after the last extension: shadow only).
// 8 waves/SIMD (= two 16-wave workgroups per CU, the LDS plan of the binary16 layout) caps the kernel at 64 VGPRs.
// PRIMARY != 0 = bounce 0: the rays are generated from their index (primary_ray), nothing is read; no shadow rays are pending.
//   kPrimaryAll: every ray i of the batch, numbered 0 .. n_pix * batch_n - 1;  kPrimaryList: every sample of the pixels k_primary_cand
//   listed in in.path (pixels without a candidate list; closest_queue.n[0] = how many); results are stored under the ray's index.
constexpr int kPrimaryNone = 0, kPrimaryAll = 1, kPrimaryList = 2;
// kPrimaryList numbering: ray j of n_ov * batch_n is sample slot j / n_ov of the j % n_ov-th listed pixel (slot-major: the lanes of a wave
// take neighbouring pixels of one accumulation, like everywhere else); returns its index in the batch, slot * n_pix + pixel.
MIRT_DI uint32_t primary_list_ray(const FrameParams& fp, const uint32_t* __restrict__ ov_pix, uint32_t n_ov, float inv_n_ov, uint32_t j) {
	uint32_t k;
	const uint32_t slot = udiv_f(j, n_ov, inv_n_ov, k);
	return slot * fp.n_pix + ov_pix[k];
}
template <bool COUNT, int PRIMARY>
__global__ __launch_bounds__(kTraceBlock, 8) void k_trace(SceneDev sc, FrameParams fp,
                                                       StreamBuf in, HitRec* __restrict__ hit_out,
                                                       Queue closest_queue, uint32_t* closest_work,
                                                       ShadowBuf sh, ShadowSink sink,
                                                       Queue shadow_queue, uint32_t* shadow_work, FatList fat_closest, FatList fat_shadow,
                                                       DevCounters* ctr) {
	extern __shared__ float4 lds[];
	const uint32_t n_ov = PRIMARY == kPrimaryList ? closest_queue.n[0] : 0u;     // listed pixels (k_primary_cand)
	const float inv_n_ov = 1.0f / static_cast<float>(n_ov ? n_ov : 1u);
	if (PRIMARY) closest_queue.n = nullptr;                                    // identity numbering: ray j is slot j
	const uint32_t nc = PRIMARY == kPrimaryAll ? fp.n_pix * fp.batch_n : PRIMARY == kPrimaryList ? n_ov * fp.batch_n : queue_total(closest_queue), ns = PRIMARY ? 0u : queue_total(shadow_queue);
	if (nc + ns == 0) return;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		if (nc && PRIMARY != kPrimaryList) atomicAdd(&ctr->rays, static_cast<unsigned long long>(nc));     // (k_primary_hits has counted the whole batch)
		if (ns) atomicAdd(&ctr->shadow_rays, static_cast<unsigned long long>(ns));
	}
	uint32_t c_nodes = 0, c_spheres = 0, s_nodes = 0, s_spheres = 0, c_term = 0;
	if (sc.use_bvh && sc.n_recs != 0) {
		if (static_cast<uint64_t>(blockIdx.x) * 64u >= static_cast<uint64_t>(nc) + ns) return;   // more workgroups than minimum-size chunks: skip the staging too
		const TraceLds tl = stage_bvh(sc, lds);
		{
			auto load_ray = [&](uint32_t i, float& px, float& py, float& pz, float& dx, float& dy, float& dz, float& tf) {
				if (PRIMARY) { uint32_t path; primary_ray(fp, PRIMARY == kPrimaryList ? primary_list_ray(fp, in.path, n_ov, inv_n_ov, i) : i, path, dx, dy, dz); px = fp.cam.pos[0]; py = fp.cam.pos[1]; pz = fp.cam.pos[2]; }
				else { px = in.px[i]; py = in.py[i]; pz = in.pz[i]; dx = in.dx[i]; dy = in.dy[i]; dz = in.dz[i]; }
				tf = MIRT_FLT_MAX;                                                 // hit reset, Renderer.hpp:150-158
			};
			auto store_result = [&](uint32_t i, const Trav& t, bool) { const uint32_t o = PRIMARY == kPrimaryList ? primary_list_ray(fp, in.path, n_ov, inv_n_ov, i) : i; hit_out[o] = HitRec{ t.tfar, t.prim }; };
			trace_queue<kClosest, COUNT>(sc, tl, closest_queue, nc, closest_work, fat_closest, c_nodes, c_spheres, load_ray, store_result);
		}
		if (!PRIMARY) {
			auto load_ray = [&](uint32_t i, float& px, float& py, float& pz, float& dx, float& dy, float& dz, float& tf) {
				shadow_origin(sh, sink, i, px, py, pz); dx = sh.dx[i]; dy = sh.dy[i]; dz = sh.dz[i]; tf = sh.tfar[i];
			};
			auto store_result = [&](uint32_t i, const Trav&, bool occluded) { shadow_finish(sh, sink, i, occluded, c_term); };
			trace_queue<kAnyHit, COUNT>(sc, tl, shadow_queue, ns, shadow_work, fat_shadow, s_nodes, s_spheres, load_ray, store_result);
		}
	} else {
		// brute force over all prims (the reference as shipped); also the no-spheres case
		const QueueView qc = PRIMARY ? queue_identity(nc) : queue_view(closest_queue);      // (kPrimaryList is never launched without a tree)
		const QueueView qs = PRIMARY ? queue_identity(0u) : queue_view(shadow_queue);
		for (uint32_t base = blockIdx.x * kTraceBlock; base < nc; base += gridDim.x * kTraceBlock) {
			const bool active = base + threadIdx.x < nc;
			const uint32_t i = active ? queue_slot(qc, base, base + threadIdx.x) : 0u;
			float px = 0, py = 0, pz = 0, dx = 1, dy = 1, dz = 1;
			if (active) {
				if (PRIMARY) { uint32_t path; primary_ray(fp, i, path, dx, dy, dz); px = fp.cam.pos[0]; py = fp.cam.pos[1]; pz = fp.cam.pos[2]; }
				else { px = in.px[i]; py = in.py[i]; pz = in.pz[i]; dx = in.dx[i]; dy = in.dy[i]; dz = in.dz[i]; }
			}
			float tfar = MIRT_FLT_MAX;             // hit reset, Renderer.hpp:150-158
			int32_t prim = -1;
			if (sc.use_bvh == 0) traverse_brute<false, COUNT>(sc, lds, active, px, py, pz, dx, dy, dz, tfar, prim, c_spheres);
			if (active) hit_out[i] = HitRec{ tfar, prim };
		}
		for (uint32_t base = blockIdx.x * kTraceBlock; base < ns; base += gridDim.x * kTraceBlock) {
			const bool active = base + threadIdx.x < ns;
			const uint32_t i = active ? queue_slot(qs, base, base + threadIdx.x) : 0u;
			float px = 0, py = 0, pz = 0, dx = 1, dy = 1, dz = 1, tfar = 0;
			if (active) { shadow_origin(sh, sink, i, px, py, pz); dx = sh.dx[i]; dy = sh.dy[i]; dz = sh.dz[i]; tfar = sh.tfar[i]; }
			bool occluded = false;
			int32_t dummy = -1;
			if (sc.use_bvh == 0) occluded = traverse_brute<true, COUNT>(sc, lds, active, px, py, pz, dx, dy, dz, tfar, dummy, s_spheres);
			if (active) shadow_finish(sh, sink, i, occluded, c_term);
		}
	}
	wave_sum(c_term, &ctr->terminated);
	if (COUNT) { wave_sum(c_nodes, &ctr->nodes); wave_sum(c_spheres, &ctr->spheres); wave_sum(s_nodes, &ctr->shadow_nodes); wave_sum(s_spheres, &ctr->shadow_spheres); }
}

// Fat rays (see trav_begin): one workgroup per ray runs the reference's brute-force loops over ALL prims —
// intersect_prims (BVH.hpp:236-288; closest = lexicographic minimum of (dist, prim index), i.e. the ascending strict-'<' scan)
// for the closest-hit list, intersect_prims_shadow (BVH.hpp:290-305) for the shadow list.
template <bool COUNT, int PRIMARY>
__global__ __launch_bounds__(1024) void k_trace_fat(SceneDev sc, FrameParams fp, StreamBuf in, HitRec* __restrict__ hit_out, FatList fat_closest,
                                                    ShadowBuf sh, ShadowSink sink, FatList fat_shadow, DevCounters* ctr, const uint32_t* ov_count) {
	__shared__ float s_t[16];
	__shared__ int32_t s_p[16];
	const uint32_t nc = min(*fat_closest.count, fat_closest.capacity), ns = min(*fat_shadow.count, fat_shadow.capacity);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
	const uint32_t n_ov = PRIMARY == kPrimaryList ? *ov_count : 0u;
	const float inv_n_ov = 1.0f / static_cast<float>(n_ov ? n_ov : 1u);
	for (uint32_t k = blockIdx.x; k < nc; k += gridDim.x) {
		const uint32_t i = PRIMARY == kPrimaryList ? primary_list_ray(fp, in.path, n_ov, inv_n_ov, fat_closest.rays[k]) : fat_closest.rays[k];
		float px, py, pz, dx, dy, dz;
		if (PRIMARY) { uint32_t path; primary_ray(fp, i, path, dx, dy, dz); px = fp.cam.pos[0]; py = fp.cam.pos[1]; pz = fp.cam.pos[2]; }
		else { px = in.px[i]; py = in.py[i]; pz = in.pz[i]; dx = in.dx[i]; dy = in.dy[i]; dz = in.dz[i]; }
		float tfar = MIRT_FLT_MAX; int32_t prim = -1;
		for (uint32_t p = threadIdx.x; p < sc.n_spheres; p += blockDim.x) sphere_closest(sc.spheres[p], static_cast<int32_t>(p), px, py, pz, dx, dy, dz, tfar, prim);
		// lexicographic (dist, prim) minimum across the workgroup; prim -1 = no hit, always loses (its tfar is FLT_MAX and every hit is < FLT_MAX)
		for (int off = 32; off > 0; off >>= 1) {
			const float ot = __shfl_down(tfar, off, 64); const int32_t op = __shfl_down(prim, off, 64);
			const bool take = (op >= 0) & ((prim < 0) | (ot < tfar) | ((ot == tfar) & (op < prim)));
			tfar = take ? ot : tfar; prim = take ? op : prim;
		}
		__syncthreads();
		if (lane == 0) { s_t[wave] = tfar; s_p[wave] = prim; }
		__syncthreads();
		if (threadIdx.x == 0) {
			for (uint32_t w = 1; w < n_waves; w++) {
				const float ot = s_t[w]; const int32_t op = s_p[w];
				const bool take = (op >= 0) && ((prim < 0) || (ot < tfar) || ((ot == tfar) && (op < prim)));
				if (take) { tfar = ot; prim = op; }
			}
			hit_out[i] = HitRec{ tfar, prim };
			if (COUNT) atomicAdd(&ctr->spheres, static_cast<unsigned long long>(sc.n_spheres));
		}
	}
	for (uint32_t k = blockIdx.x; k < ns; k += gridDim.x) {
		const uint32_t i = fat_shadow.rays[k];
		float px, py, pz;
		shadow_origin(sh, sink, i, px, py, pz);
		const float dx = sh.dx[i], dy = sh.dy[i], dz = sh.dz[i], tfar = sh.tfar[i];
		bool occ = false;
		for (uint32_t p = threadIdx.x; p < sc.n_spheres && !occ; p += blockDim.x) occ = sphere_occludes(sc.spheres[p], px, py, pz, dx, dy, dz, tfar);
		const int any = __syncthreads_or(occ ? 1 : 0);
		if (threadIdx.x == 0) {
			uint32_t c_term = 0;
			shadow_finish(sh, sink, i, any != 0, c_term);
			if (c_term) atomicAdd(&ctr->terminated, 1ull);
			if (COUNT) atomicAdd(&ctr->shadow_spheres, static_cast<unsigned long long>(sc.n_spheres));
		}
	}
}

// ------------------------------------------------------------------------------------------------
// PRIMARY RAYS THROUGH PER-PIXEL CANDIDATE LISTS (see kCollect above)
// ------------------------------------------------------------------------------------------------
// One cone per local pixel: axis = Camera::generate_ray (Camera.hpp:80-88) through the pixel centre, half-angle rho (host:
// 0.7072 / |projection.z| + margins — the jitter moves a sample by at most half a pixel per axis in the image plane, whose
// points are at least |z| from the eye).  Same persistent-wave traversal as k_trace; the result is the pixel's candidate list.
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock, 8) void k_primary_cand(SceneDev sc, FrameParams fp, uint32_t* __restrict__ cand, float rho, uint32_t* work, FatList unused, DevCounters* ctr,
                                                                uint32_t* __restrict__ ov_pix, uint32_t* __restrict__ ov_count) {
	extern __shared__ float4 lds[];
	const uint32_t n = fp.n_pix;
	if (n == 0 || static_cast<uint64_t>(blockIdx.x) * 64u >= n) return;
	const TraceLds tl = stage_bvh(sc, lds);
	uint32_t c_nodes = 0, c_spheres = 0;
	auto load_ray = [&](uint32_t pix, float& px, float& py, float& pz, float& dx, float& dy, float& dz, float& tf) {
		uint32_t tile; int32_t x, y;
		pixel_xy(fp, pix, tile, x, y);
		const f3 d = camera_ray_dir(fp.cam, x, y, 0.5f, 0.5f);
		px = fp.cam.pos[0]; py = fp.cam.pos[1]; pz = fp.cam.pos[2]; dx = d.x; dy = d.y; dz = d.z; tf = MIRT_FLT_MAX;
	};
	auto store_result = [&](uint32_t pix, const Trav& t, bool) {
		const bool over = static_cast<uint32_t>(t.prim) > kCandMax;
		cand[pix] = over ? kCandOverflow : static_cast<uint32_t>(t.prim);
		// pixels without a list: every sample is traced by k_trace<kPrimaryList>.  One atomic per wave and flush for the lanes that report one.
		const unsigned long long m = __ballot(over);
		if (over) {
			const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(m))) - 1u;
			uint32_t first = 0;
			if (lane_id() == leader) first = atomicAdd(ov_count, static_cast<uint32_t>(__popcll(m)));
			first = __builtin_amdgcn_readlane(first, leader);
			if (ov_pix) ov_pix[first + mask_rank(m)] = pix;
		}
	};
	trace_queue<kCollect, COUNT>(sc, tl, Queue{ nullptr, 0u }, n, work, unused, c_nodes, c_spheres, load_ray, store_result, Collect{ cand, rho, n });
	if (COUNT) { wave_sum(c_nodes, &ctr->nodes); wave_sum(c_spheres, &ctr->spheres); }
}
// Traverse (BVH.hpp:309-360) for the camera rays of a batch, given the lists: one lane per PIXEL runs the batch's accumulations over it.  What
// depends on the pixel alone — tile, x, y, seed[ID], the list and its first kCandRegs spheres — is set up once, so a sample costs its ray
// (hash_2d, two PCG draws, the quaternion rotation and normalisation of Camera::generate_ray, Camera.hpp:80-88) + the exact sphere tests on
// registers (sphere_closest_tie: the reference's arithmetic; the (dist, prim index) minimum does not depend on the order of the list) + one 8-B
// hit record, written for 64 neighbouring pixels at a time.  (Round 2 ran one lane per SAMPLE: every ray paid the pixel's set-up and walked the
// list through three dependent loads — 7.3 ms per cfg4 batch of 537 M rays.)  Pixels without a list are skipped: k_trace<kPrimaryList> traces
// all their samples.
constexpr uint32_t kCandRegs = 3;
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_primary_hits(SceneDev sc, FrameParams fp, const uint32_t* __restrict__ cand, HitRec* __restrict__ hit_out, DevCounters* ctr) {
	if (blockIdx.x == 0 && threadIdx.x == 0 && fp.n_pix) atomicAdd(&ctr->rays, static_cast<unsigned long long>(fp.n_pix) * fp.batch_n);     // Renderer.hpp:165: every camera ray of the batch
	uint32_t c_spheres = 0;
	const float ox = fp.cam.pos[0], oy = fp.cam.pos[1], oz = fp.cam.pos[2];
	for (uint32_t pix = blockIdx.x * kBlock + threadIdx.x; pix < fp.n_pix; pix += gridDim.x * kBlock) {
		const uint32_t cnt = cand[pix];
		if (cnt == kCandOverflow) continue;
		uint32_t tile; int32_t x, y;
		pixel_xy(fp, pix, tile, x, y);
		const uint32_t seed = (tile * kTileSize + (pix & 255u)) * (fp.max_bounces * 2u + 1u);      // seed[ID], Renderer.hpp:107
		float4 s[kCandRegs]; int32_t id[kCandRegs];
		for (uint32_t k = 0; k < kCandRegs; k++) {
			id[k] = k < cnt ? static_cast<int32_t>(cand[static_cast<size_t>(k + 1u) * fp.n_pix + pix]) : -1;
			s[k] = sc.spheres[id[k] < 0 ? 0 : id[k]];
		}
		for (uint32_t slot = 0; slot < fp.batch_n; slot++) {
			uint32_t rng = hash_2d(fp.acc_base + slot + 1u, seed);                 // ++accumulations, Renderer.hpp:74,117
			const float s0 = rand_unit_float(rng);
			const float s1 = rand_unit_float(rng);
			const f3 d = camera_ray_dir(fp.cam, x, y, s0, s1);
			float tfar = MIRT_FLT_MAX; int32_t prim = -1;                          // hit reset, Renderer.hpp:150-158
			for (uint32_t k = 0; k < kCandRegs; k++) if (id[k] >= 0) sphere_closest_tie(s[k], id[k], ox, oy, oz, d.x, d.y, d.z, tfar, prim);
			for (uint32_t k = kCandRegs; k < kCandMax; k++) {                      // the rest of a long list, from L1
				if (__ballot(k < cnt) == 0ull) break;
				if (k < cnt) {
					const uint32_t j = cand[static_cast<size_t>(k + 1u) * fp.n_pix + pix];
					sphere_closest_tie(sc.spheres[j], static_cast<int32_t>(j), ox, oy, oz, d.x, d.y, d.z, tfar, prim);
				}
			}
			if (COUNT) c_spheres += cnt;
			const size_t i = static_cast<size_t>(slot) * fp.n_pix + pix;
			hit_out[i] = HitRec{ tfar, prim };
		}
	}
	if (COUNT) wave_sum(c_spheres, &ctr->spheres);
}

// ------------------------------------------------------------------------------------------------
// Accumulator addressing — AccumulationTile<k>, Renderer.hpp:43-46,84: [tile][bucket][r,g,b][256]
// ------------------------------------------------------------------------------------------------
// Sky::operator(), Primitives.hpp:35-46
MIRT_DI f3 sky_eval(const SceneDev& sc, float x, float y, float z) {
	const float ex = sc.hdri_fw * (0.5f + MIRT_INV_TWO_PI * fast_atan2(z, x));
	const float ey = sc.hdri_fh * (0.5f - MIRT_INV_PI * fast_asin(y));
	const float4 t = sc.hdri[static_cast<int32_t>(ey) * sc.hdri_w + static_cast<int32_t>(ex)];
	return { t.x * sc.ambient[0], t.y * sc.ambient[1], t.z * sc.ambient[2] };
}

// ------------------------------------------------------------------------------------------------
// SHADE — closest-hit shader, NEE, emissive, BRDF sample + Russian roulette + compaction, miss, accumulate
// (Renderer.hpp:169-431).  FIRST = bounce 0: radiance 0, throughput 1 (Renderer.hpp:98-101) without reading them.
// ------------------------------------------------------------------------------------------------
// Dense list of the rays for which `flag` is set, in LDS: list[rank] = value; returns the number of entries.  All threads
// of the block, converged.  scratch = 17 words.  (Two barriers; the list may be read after return.)
MIRT_DI uint32_t block_compact(bool flag, uint32_t value, uint32_t* scratch, uint32_t* list) {
	const unsigned long long m = __ballot(flag);
	const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
	if (lane_id() == 0) scratch[wave] = static_cast<uint32_t>(__popcll(m));
	__syncthreads();
	uint32_t before = 0, total = 0;
	for (uint32_t w = 0; w < n_waves; w++) { const uint32_t c = scratch[w]; before += (w < wave) ? c : 0u; total += c; }
	if (flag) list[before + mask_rank(m)] = value;
	__syncthreads();
	return total;
}

// Block-iterations.  A later bounce reads a stream: iteration t covers its ray numbers [512 t, 512 t + 512), t = blockIdx.x, + gridDim.x, ...
// Bounce 0 (FIRST) has no stream — ray i = slot * n_pix + pixel is a function of its index — and runs PIXEL-MAJOR: a workgroup takes chunks of
// 512 local pixels (two tiles) and runs all the batch's accumulations over each, so what depends on the pixel alone (tile, x, y, seed[ID],
// seed[ID]) is set up once per chunk.  Its hit records come from k_primary_hits (pixels with a candidate list) or k_trace.
// (Measured and dropped twice: the candidate tests inside this kernel instead of k_primary_hits — ray-major in round 2, pixel-major in
// round 3: at 6 waves per SIMD and 80 VGPRs the list's dependent loads cost k_shade<FIRST> 5 ms per cfg4 batch, as much as the kernel saved.)
template <bool FIRST>
__global__ __launch_bounds__(kShadeBlock, 6) void k_shade(SceneDev sc, FrameParams fp, StreamBuf in, const HitRec* __restrict__ hit_in, StreamBuf out, ShadowBuf sh, uint32_t bounce,
                                                  Queue in_queue, Queue next_queue, Queue shadow_queue, float* __restrict__ accum, DevCounters* ctr) {
	const QueueView qin = FIRST ? queue_identity(fp.n_pix * fp.batch_n) : queue_view(in_queue);
	const uint32_t n = qin.pre[kSegs];
	const uint32_t n_chunks = (fp.n_pix + kShadeBlock - 1u) / kShadeBlock;      // FIRST
	const bool last_bounce = !(bounce < fp.max_bounces - 1u);                 // Renderer.hpp:358
	const float light_selection_pdf = 1.0f / static_cast<float>(fp.n_lights);  // Renderer.hpp:78
	__shared__ uint32_t append_scratch[72];
	__shared__ uint32_t compact_scratch[17];
	__shared__ uint32_t hit_list[kShadeBlock];
	__shared__ float4 s_albedo[MIRT_MAX_MATERIALS + 1], s_emission[MIRT_MAX_MATERIALS + 1];      // scene.material: 2 KB, read by every hit
	uint32_t c_term = 0, c_drop = 0, parity = 0;
	const uint32_t n_units = n_chunks * fp.first_groups;                       // FIRST: pieces of work = (chunk, group of accumulations)
	if (FIRST ? blockIdx.x >= n_units : blockIdx.x * kShadeBlock >= n) return;
	for (uint32_t m = threadIdx.x; m < sc.n_mat; m += kShadeBlock) { s_albedo[m] = sc.mat_albedo[m]; s_emission[m] = sc.mat_emission[m]; }
	__syncthreads();            // the table is read by every wave in phase 2; k_shade<FIRST> reaches no other barrier before that (the early return above is block-uniform)

	// !FIRST: this lane's ray of the stream for the block-iteration at hand, and its hit record: requested one iteration ahead, so that the
	// first of the iteration's three dependent memory round trips is already under way
	uint32_t next_slot = 0u; int32_t next_prim = -1;
	if (!FIRST) {
		next_slot = (blockIdx.x * kShadeBlock + threadIdx.x < n) ? queue_slot(qin, blockIdx.x * kShadeBlock, blockIdx.x * kShadeBlock + threadIdx.x) : 0u;
		next_prim = hit_in[next_slot].prim;
	}
	// FIRST: the pixel of this lane in the chunk at hand, and what depends on it alone
	uint32_t unit = blockIdx.x, chunk = 0u, slot_it = 0u, slot_end = 0u, pix = 0u, pix_seed = 0u;
	int32_t pix_x = 0, pix_y = 0;
	for (uint32_t base = blockIdx.x * kShadeBlock; FIRST ? unit < n_units : base < n; parity ^= 1u) {
		const bool new_unit = FIRST && slot_it == slot_end;                    // wave-uniform
		if (new_unit) {
			chunk = unit / fp.first_groups;
			const uint32_t g = unit - chunk * fp.first_groups;
			slot_it = g * fp.batch_n / fp.first_groups; slot_end = (g + 1u) * fp.batch_n / fp.first_groups;
		}
		const uint32_t iteration = FIRST ? chunk * fp.batch_n + slot_it : base / kShadeBlock;     // FIRST: every (chunk, slot) exactly once
		// ---- phase 1, one lane per ray of the stream: misses end here; hits are only listed ----
		bool is_hit = false;
		uint32_t my_slot = next_slot;
		int32_t my_prim = next_prim;
		float my_tfar = 0.0f;
		f3 my_D{0, 0, 0};
		uint32_t my_path = 0u;
		bool lane_on;
		if (FIRST) {
			if (new_unit) {                                                      // a new chunk of pixels
				pix = chunk * kShadeBlock + threadIdx.x;
				if (pix < fp.n_pix) {
					uint32_t tile;
					pixel_xy(fp, pix, tile, pix_x, pix_y);
					pix_seed = (tile * kTileSize + (pix & 255u)) * (fp.max_bounces * 2u + 1u);      // seed[ID], Renderer.hpp:107
				}
			}
			lane_on = pix < fp.n_pix;
			my_slot = slot_it * fp.n_pix + pix;                                 // the ray's index in the batch = where k_trace stored a hit record for it
			if (lane_on) {
				// RAY GENERATION, Renderer.hpp:113-127 (primary_ray with the pixel's part taken from the chunk set-up)
				uint32_t rng = hash_2d(fp.acc_base + slot_it + 1u, pix_seed);
				const float s0 = rand_unit_float(rng);
				const float s1 = rand_unit_float(rng);
				my_D = camera_ray_dir(fp.cam, pix_x, pix_y, s0, s1);
				my_path = (slot_it << fp.pix_bits) | pix;
				{ const HitRec h = hit_in[my_slot]; my_prim = h.prim; my_tfar = h.tfar; }      // the hit record of k_primary_hits / k_trace
			}
			if (++slot_it == slot_end) unit += gridDim.x;
		} else {
			lane_on = base + threadIdx.x < n;
			const uint32_t nb = base + gridDim.x * kShadeBlock;
			next_slot = (nb + threadIdx.x < n) ? queue_slot(qin, nb, nb + threadIdx.x) : 0u;
			next_prim = hit_in[next_slot].prim;
			base = nb;
		}
		{
			if (lane_on) {
				const uint32_t i = my_slot;
				const int32_t prim = my_prim;
				if (prim < 0) {
					// MISS SHADER, Renderer.hpp:408-420 (Q10: throughput.r scales all three channels)
					f3 R{0.0f, 0.0f, 0.0f};
					float thr_x = 1.0f;
					uint32_t mpath = my_path; f3 md = my_D;
					if (!FIRST) { R = { in.rr[i], in.rg[i], in.rb[i] }; thr_x = in.tr[i]; mpath = in.path[i]; }
					if (sc.has_ambient) {
						if (!FIRST) md = { in.dx[i], in.dy[i], in.dz[i] };
						const f3 sky = sky_eval(sc, md.x, md.y, md.z);
						R.x += thr_x * sky.x; R.y += thr_x * sky.y; R.z += thr_x * sky.z;
					}
					accumulate_add(accum, accum_index(fp, mpath), R.x + 0.0f, R.y + 0.0f, R.z + 0.0f);   // ACCUMULATION, Renderer.hpp:424-430 (+ the zero emissive term)
					c_term++;
				} else if (last_bounce) {
					c_drop++;                                                         // Q5: still alive after the last bounce -> never accumulated
				} else is_hit = true;
			}
		}
		// ---- regroup: the closest-hit shader is ~800 VALU instructions per ray and only 40-60 % of a secondary stream hits;
		// packing the hits of the block into its first waves runs that code on full waves (lane utilisation 0.42 -> ~0.9) ----
		// (Not for primary rays: ~95 % of them hit, the stream is already dense, and the two barriers cost more than they save.)
		const uint32_t n_hits = FIRST ? 0u : block_compact(is_hit, my_slot, compact_scratch, hit_list);

		// ---- phase 2, one lane per hit ----
		bool survive = false, has_shadow = false, terminated = false, has_E = false;
		uint32_t path = 0;
		f3 P{0, 0, 0}, ndir{0, 0, 0}, L{0, 0, 0}, srad{0, 0, 0}, E{0, 0, 0};
		f3 R{0.0f, 0.0f, 0.0f}, thr{1.0f, 1.0f, 1.0f};
		float light_distance = 0.0f;
		if (FIRST ? is_hit : threadIdx.x < n_hits) {
			const uint32_t i = FIRST ? my_slot : hit_list[threadIdx.x];
			f3 D = my_D;                                                       // bounce 0 has no stream: the ray is a function of its index (phase 1)
			path = my_path;
			if (!FIRST) { path = in.path[i]; D = { in.dx[i], in.dy[i], in.dz[i] }; }
			float pdf_in = 0.0f;
			if (!FIRST) {
				R = { in.rr[i], in.rg[i], in.rb[i] };
				thr = { in.tr[i], in.tg[i], in.tb[i] };
				pdf_in = MIRT_INV_PI * max_sel(0.0f, D.z);                        // out->pdf of the bounce that sampled D (Q8), bit for bit
			}
			const HitRec hrec = FIRST ? HitRec{ my_tfar, my_prim } : hit_in[i];
			const int32_t prim = hrec.prim;
			{
				// CLOSEST HIT SHADER, Renderer.hpp:169-214
				const float depth = hrec.tfar;
				const float4 hs = sc.spheres[prim];
				const int32_t mat = sc.prim_mat[prim];
				const f3 O = FIRST ? f3{ fp.cam.pos[0], fp.cam.pos[1], fp.cam.pos[2] } : f3{ in.px[i], in.py[i], in.pz[i] };
				const f3 hit{ O.x + D.x * depth, O.y + D.y * depth, O.z + D.z * depth };
				f3 N = normalize3(f3{ hit.x - hs.x, hit.y - hs.y, hit.z - hs.z });
				if (dot3(N, D) >= 0.0f) N = f3{ -N.x, -N.y, -N.z };
				const quat T = tangent_space(N);
				const f3 Vl = to_local(T, f3{ -D.x, -D.y, -D.z });
				P = { hit.x + N.x * 1e-4f, hit.y + N.y * 1e-4f, hit.z + N.z * 1e-4f };
				const float4 em = s_emission[mat];
				const float4 alb = s_albedo[mat];
				const bool is_emissive = max_sel(em.x, max_sel(em.y, em.z)) > MIRT_FLT_EPSILON;
				const uint32_t acc = fp.acc_base + (path >> fp.pix_bits) + 1u;
				const uint32_t seed = FIRST ? pix_seed : path_seed(fp, path & fp.pix_mask);

				// NEXT EVENT ESTIMATION, Renderer.hpp:247-298
				if (fp.mis) {
					uint32_t rng = hash_2d(acc, seed + bounce * 2u);
					const float u0 = rand_unit_float(rng);
					const float u1 = rand_unit_float(rng);
					const int32_t selected = static_cast<int32_t>(rand_bounded_int(rng, fp.n_lights));
					const float4 lp = sc.light_sphere[selected];                    // scene.geometry[lighting_acceleration.prims[selected]]
					const float4 lem = sc.light_emit[selected];                     // its material's emission, and the prim id
					const int32_t light_primID = static_cast<int32_t>(__float_as_uint(lem.w));
					do {
						if (light_primID == prim) break;                             // Q11: geometry-order id vs BVH-order id
						f3 Wc{ lp.x - P.x, lp.y - P.y, lp.z - P.z };
						const float center_dist2 = dot3(Wc, Wc);
						if (center_dist2 <= lp.w) break;
						const float center_dist = __builtin_sqrtf(center_dist2);
						{ const float inv = 1.0f / center_dist; Wc.x *= inv; Wc.y *= inv; Wc.z *= inv; }
						const float sinThetaMax2 = lp.w / center_dist2;
						{
							const float NdotW = (2.0f * T.w) * (Wc.z * T.w + Wc.x * T.y - T.x * Wc.y) - Wc.z;
							if (NdotW < 0.0f && sinThetaMax2 < NdotW * NdotW) break;
						}
						float ldist, lpdf;
						const f3 Ld = sample_direction_to_sphere(Wc, sinThetaMax2, center_dist, lp.w, u0, u1, ldist, lpdf);
						const f3 Ll = to_local(T, Ld);
						if (Ll.z < 0.0f) break;
						f3 rad{ lem.x * thr.x, lem.y * thr.y, lem.z * thr.z };
						{   // Closure<LambertianDiffuse>::eval, DataStreams.hpp:169-172
							const float f = MIRT_INV_PI * max_sel(0.0f, Ll.z);
							rad.x *= alb.x * f; rad.y *= alb.y * f; rad.z *= alb.z * f;
						}
						lpdf *= light_selection_pdf;
						const float brdf_pdf = MIRT_INV_PI * max_sel(0.0f, Ll.z);  // DataStreams.hpp:173-176
						const float w = powerHeuristic_over_f(lpdf, brdf_pdf);
						rad.x *= w; rad.y *= w; rad.z *= w;
						if (max_sel(max_sel(rad.x, rad.y), rad.z) <= 0.0f) break;
						has_shadow = true; L = Ld; light_distance = ldist; srad = rad;
					} while (false);
				}
				// EMISSIVE PRIMITIVE HIT, Renderer.hpp:319-353
				has_E = is_emissive;
				if (is_emissive) {
					if (fp.mis && bounce > 0) {
						const float radius2 = hs.w;
						const float center_dist2 = depth * (depth + Vl.z * (2.0f * __builtin_sqrtf(radius2))) + radius2;
						const float weight = powerHeuristic(pdf_in, light_selection_pdf * spherePdf(radius2, center_dist2));
						E = { (thr.x * weight) * em.x, (thr.y * weight) * em.y, (thr.z * weight) * em.z };
					} else {
						E = { em.x, em.y, em.z };                                   // Q9: no throughput
					}
				}
				// BRDF SAMPLING - BOUNCE, Renderer.hpp:357-404
				{
					uint32_t rng = hash_2d(acc, seed + bounce * 2u + 1u);
					const float b0 = rand_unit_float(rng);
					const float b1 = rand_unit_float(rng);
					f3 sd = hemisphere(b0, b1);                                     // Closure::sample, DataStreams.hpp:177-181
					thr = { thr.x * alb.x, thr.y * alb.y, thr.z * alb.z };
					const float q = 1.0f - max_sel(thr.x, max_sel(thr.y, thr.z));
					if (rand_unit_float(rng) < q) {
						terminated = true;                                          // Russian roulette, Renderer.hpp:377-383
					} else {
						const float inv = 1.0f / max_sel(MIRT_FLT_EPSILON, 1.0f - q);
						thr = { thr.x * inv, thr.y * inv, thr.z * inv };
						ndir = to_world(T, sd);
						survive = true;
					}
				}
			}
		}
		// ---- stream compaction: wave64 ballot + mbcnt prefix inside each wave, one atomic per workgroup and stream ----
		uint32_t slot, sslot;
		block_append2(survive, has_shadow, next_queue, shadow_queue, iteration % kSegs, append_scratch, parity, slot, sslot);
		// (R + unoccluded NEE) + E is finished by k_trace's shadow_finish once the occlusion is known.  Non-emissive hits (E = +0)
		// leave R where that result belongs and send a light record; the others send R and E along (kDestFull).
		const bool direct = fp.idx_base != 0xffffffffu;                          // paths add straight into accumulator words that hold earlier samples
		const bool full = has_shadow & (has_E | (terminated & direct));
		if (survive) {
			out.px[slot] = P.x; out.py[slot] = P.y; out.pz[slot] = P.z;
			out.dx[slot] = ndir.x; out.dy[slot] = ndir.y; out.dz[slot] = ndir.z;
			out.tr[slot] = thr.x; out.tg[slot] = thr.y; out.tb[slot] = thr.z;
			out.path[slot] = path;
		}
		if (has_shadow) {
			if (!survive) { sh.px[sslot] = P.x; sh.py[sslot] = P.y; sh.pz[sslot] = P.z; }          // else: the surviving ray's origin, found through dest
			sh.dx[sslot] = L.x; sh.dy[sslot] = L.y; sh.dz[sslot] = L.z;
			sh.tfar[sslot] = light_distance;
			sh.sr[sslot] = srad.x; sh.sg[sslot] = srad.y; sh.sb[sslot] = srad.z;
			if (full) {
				sh.rr[sslot] = R.x; sh.rg[sslot] = R.y; sh.rb[sslot] = R.z;
				sh.er[sslot] = E.x; sh.eg[sslot] = E.y; sh.eb[sslot] = E.z;
			}
			sh.dest[sslot] = (survive ? slot : (kDestAccum | path)) | (full ? kDestFull : 0u);
		}
		if ((survive | terminated) & !full) {
			const f3 Rf{ R.x + E.x, R.y + E.y, R.z + E.z };                       // E is +0 when the hit is not emissive (exact no-op); with a light record pending E is +0 too
			if (survive) { out.rr[slot] = Rf.x; out.rg[slot] = Rf.y; out.rb[slot] = Rf.z; }
			else accumulate_add(accum, accum_index(fp, path), Rf.x, Rf.y, Rf.z);   // ACCUMULATION, Renderer.hpp:424-430 (a pending light record adds its NEE term to the same word)
		}
		c_term += (terminated && !has_shadow) ? 1u : 0u;
	}
	wave_sum(c_term, &ctr->terminated);
	wave_sum(c_drop, &ctr->dropped);
}

// In-order merge of one batch's contribution buffer ([tile][slot][rgb][256], slot = accumulation index inside the batch)
// into the accumulator (see launch_batch in mirt_capi.hip): exactly the `output_color[px] += radiance` of
// Renderer.hpp:427-429, one add per (pixel, bucket) per Accumulate() call.  Accumulation acc_base+k+1 lands in bucket
// (acc_base+k+1) % buckets (Renderer.hpp:82); each accumulator word is owned by one thread, which applies that bucket's
// contributions in ascending k, i.e. in accumulation order.  Entries a path did not touch hold +0 (exact no-op).
__global__ __launch_bounds__(kBlock) void k_merge_contrib(float4* __restrict__ accum, const float4* __restrict__ contrib, uint32_t n_tiles, uint32_t buckets,
                                                          uint32_t batch_n, uint32_t acc_base) {
	constexpr uint32_t kQuads = 3u * kTileSize / 4u;                            // float4 per (tile, bucket)
	const size_t n_items = static_cast<size_t>(n_tiles) * kQuads;
	const uint32_t first = min(buckets, batch_n);
	for (size_t item = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; item < n_items; item += static_cast<size_t>(gridDim.x) * kBlock) {
		const size_t tile = item / kQuads; const uint32_t q = static_cast<uint32_t>(item % kQuads);
		for (uint32_t k0 = 0; k0 < first; k0++) {
			const uint32_t bucket = (acc_base + k0 + 1u) % buckets;
			float4* dst = accum + (tile * buckets + bucket) * kQuads + q;
			float4 a = *dst;
			for (uint32_t k = k0; k < batch_n; k += buckets) {
				const float4 c = contrib[(tile * batch_n + k) * kQuads + q];
				a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
			}
			*dst = a;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// MEDIAN OF MEANS & TONEMAPPING — Renderer::Render, Renderer.hpp:436-478
// ------------------------------------------------------------------------------------------------
MIRT_DI float median_k(float* v, uint32_t k) {
	if (k == 5) return median5(v[0], v[1], v[2], v[3], v[4]);               // the reference network, Sampling.hpp:13-21
	for (uint32_t a = 1; a < k; a++) {                                      // generalisation (Q19): insertion sort
		float key = v[a]; int b = static_cast<int>(a) - 1;
		while (b >= 0 && key < v[b]) { v[b + 1] = v[b]; b--; }
		v[b + 1] = key;
	}
	return (k & 1u) ? v[k / 2] : (v[k / 2 - 1] + v[k / 2]) * 0.5f;
}
__global__ __launch_bounds__(kBlock) void k_resolve(const float* __restrict__ accum, float4* __restrict__ fb, uint32_t n_pix, uint32_t first_tile,
                                                    uint32_t run_tiles, uint32_t stride_tiles, uint32_t h_tiles, uint32_t width, uint32_t buckets, float scale) {
	for (uint32_t pix = blockIdx.x * kBlock + threadIdx.x; pix < n_pix; pix += gridDim.x * kBlock) {
		const float* src = accum + static_cast<size_t>(pix >> 8) * buckets * 3u * kTileSize + (pix & 255u);
		float ch[3];
		for (uint32_t c = 0; c < 3; c++) {
			float v[16];
			for (uint32_t b = 0; b < 16; b++) if (b < buckets) v[b] = src[(static_cast<size_t>(b) * 3u + c) * kTileSize];
			ch[c] = scale * median_k(v, buckets);                               // Renderer.hpp:453-455
		}
		tonemapping(ch[0], ch[1], ch[2]);                                       // Renderer.hpp:461
		const uint32_t tile = global_tile(first_tile, run_tiles, stride_tiles, pix >> 8), ID = pix & 255u;
		const uint32_t x = kTileRoot * (tile % h_tiles) + (ID & 15u);
		const uint32_t y = kTileRoot * (tile / h_tiles) + (ID >> 4);
		fb[static_cast<size_t>(y) * width + x] = make_float4(ch[0], ch[1], ch[2], 1.0f);   // Renderer.hpp:447,465
	}
}

// ------------------------------------------------------------------------------------------------
// Stage-level debug kernels (mirt_debug_*)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_debug_math(int fn, uint32_t n, const float* in, float* out) {
	for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
		switch (fn) {
		case 0: { float s, c; fast_sincos(in[i], s, c); out[i] = s; out[n + i] = c; } break;
		case 1: out[i] = fast_atan2(in[i], in[n + i]); break;
		case 2: out[i] = fast_asin(in[i]); break;
		case 3: out[i] = 1.0f / in[i]; out[n + i] = __builtin_sqrtf(fabs_bits(in[i])); out[2 * n + i] = in[i] / in[n + i]; break;
		case 4: { f3 h = hemisphere(in[i], in[n + i]); out[i] = h.x; out[n + i] = h.y; out[2 * n + i] = h.z; } break;
		case 5: {
			f3 N{ in[i], in[n + i], in[2 * n + i] }, v{ in[3 * n + i], in[4 * n + i], in[5 * n + i] };
			quat T = tangent_space(N); f3 l = to_local(T, v); f3 w = to_world(T, v);
			out[i] = T.x; out[n + i] = T.y; out[2 * n + i] = T.z; out[3 * n + i] = T.w;
			out[4 * n + i] = l.x; out[5 * n + i] = l.y; out[6 * n + i] = l.z;
			out[7 * n + i] = w.x; out[8 * n + i] = w.y; out[9 * n + i] = w.z;
		} break;
		case 6: {
			f3 Wc{ in[i], in[n + i], in[2 * n + i] };
			float dist, pdf;
			f3 Ld = sample_direction_to_sphere(Wc, in[3 * n + i], in[4 * n + i], in[5 * n + i], in[6 * n + i], in[7 * n + i], dist, pdf);
			out[i] = Ld.x; out[n + i] = Ld.y; out[2 * n + i] = Ld.z; out[3 * n + i] = dist; out[4 * n + i] = pdf;
		} break;
		case 7: {
			uint32_t s = hash_2d(__float_as_uint(in[i]), __float_as_uint(in[n + i]));
			out[i] = __uint_as_float(s);
			out[n + i] = rand_unit_float(s); out[2 * n + i] = rand_unit_float(s);
			uint32_t s2 = s;
			out[3 * n + i] = rand_unit_float(s);
			out[4 * n + i] = __uint_as_float(rand_bounded_int(s2, __float_as_uint(in[2 * n + i])));
		} break;
		case 10: out[i] = sqrt_trav(in[i]); break;
		case 8: {   // Closure<GGX>::eval: in F0(3), alpha, L(3), V(3) -> out 3
			const f3 r = ggx_eval(f3{ in[i], in[n + i], in[2 * n + i] }, in[3 * n + i], f3{ in[4 * n + i], in[5 * n + i], in[6 * n + i] }, f3{ in[7 * n + i], in[8 * n + i], in[9 * n + i] });
			out[i] = r.x; out[n + i] = r.y; out[2 * n + i] = r.z;
		} break;
		case 9: {   // Closure<GGX>::sample: in F0(3), alpha, V(3), u0, u1 -> out dir(3), estimator(3)
			f3 d, e;
			ggx_sample(f3{ in[i], in[n + i], in[2 * n + i] }, in[3 * n + i], f3{ in[4 * n + i], in[5 * n + i], in[6 * n + i] }, in[7 * n + i], in[8 * n + i], d, e);
			out[i] = d.x; out[n + i] = d.y; out[2 * n + i] = d.z; out[3 * n + i] = e.x; out[4 * n + i] = e.y; out[5 * n + i] = e.z;
		} break;
		default: break;
		}
	}
}

} // namespace mirt
