// mirt_capi.hip — C-ABI (include/mirt.h) over the HIP kernels in kernels.hpp.
//
// Host-side stand-in for the state Renderer<Policy> keeps (Renderer.hpp:38-49): scene reference,
// accumulator, width/height, accumulations, h_tiles/v_tiles — with the scene copied to HBM, the
// accumulator resident in HBM in the reference's AccumulationTile layout, and the per-tile ray
// streams replaced by one set of frame-wide SoA streams (see kernels.hpp).
//
// There is no CPU fallback: every entry point that computes needs a gfx950 device and fails with
// MIRT_ERR_NO_DEVICE / MIRT_ERR_HIP otherwise.
#include "../../include/mirt.h"
#include "kernels.hpp"
#include "bvh_layout.hpp"
#include "bvh_build.hpp"
#include "lbvh_build.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mirt;

namespace {

thread_local std::string g_create_error;

struct DeviceBuffer {
	void* ptr = nullptr;
	size_t bytes = 0;
	hipError_t ensure(size_t want) {
		if (want <= bytes && ptr) return hipSuccess;
		if (ptr) { (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
		if (want == 0) return hipSuccess;
		hipError_t e = hipMalloc(&ptr, want);
		if (e == hipSuccess) bytes = want;
		return e;
	}
	void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
	template <class T> T* as() const { return static_cast<T*>(ptr); }
};

// Function-local buffers of the debug entry points: freed on every return path.
struct ScopedBuffer : DeviceBuffer {
	ScopedBuffer() = default;
	ScopedBuffer(const ScopedBuffer&) = delete;
	ScopedBuffer& operator=(const ScopedBuffer&) = delete;
	~ScopedBuffer() { release(); }
};

struct TimedLaunch { int klass; hipEvent_t start, stop; };

// One batch in flight: its own HIP stream, ray streams and (when more than one slot exists) contribution buffer.
struct PipeSlot {
	hipStream_t stream = nullptr;
	DeviceBuffer arena;              // ray streams
	DeviceBuffer counts;             // ray queues (kSegs counters each): stream[nb+1] | shadow[nb] | one always-empty queue; then work_next[nb] | work_next_shadow[nb] | fat counts[2 nb]
	DeviceBuffer contrib;            // this batch's adds, same layout as the accumulator
	DeviceBuffer fat;                // fat-ray index lists: [closest kFatCapacity][shadow kFatCapacity]
	DeviceBuffer cand;               // per local pixel: candidate spheres of its bundle of camera rays (k_primary_cand), kCandStride words
	hipEvent_t batch_done = nullptr; // recorded on `stream` after the batch's last kernel
	hipEvent_t merged = nullptr;     // recorded on the main stream after the batch was merged (slot reusable)
	hipEvent_t zeroed = nullptr;     // recorded on the side stream after the contribution buffer was zeroed again (one batch in flight)
	bool contrib_is_zero = false;    // the whole contribution buffer holds +0 (or will, once `zeroed` has fired)
	bool zero_pending = false;       // `zeroed` must be waited for before the next write into the buffer
	bool in_use = false;
	StreamBuf stream_buf[2]{};
	ShadowBuf shadow_buf{};
	HitRec* hit = nullptr;           // RayStream<>::Hit: one 8-B {tfar, primID} record per ray (two planes' worth of the arena)
};

} // namespace

struct mirt_ctx {
	int device = 0;
	int n_cu = 256;
	hipStream_t stream = nullptr;
	hipStream_t side_stream = nullptr;      // re-zeroes a contribution buffer while the next batch's camera-ray phase runs (launch_batch)
	std::string error;

	mirt_policy policy{ 16, 5, 1, 0, 0, 0, 0, 0, 0, 0, 0, { 0 } };     // RendererPolicy defaults, Renderer.hpp:19-26,41,71; USEBVH false BVH.hpp:307
	uint32_t width = 0, height = 0, h_tiles = 0, v_tiles = 0;
	uint32_t first_tile = 0, n_tiles = 0;
	uint32_t run_tiles = 0, stride_tiles = 0;   // interleaved tile rows (mirt_set_tile_rows); stride 0 = one contiguous range
	uint32_t accumulations = 0;
	bool have_scene = false, have_camera = false;

	// scene
	DeviceBuffer recs, recs_wide, spheres, prim_mat, light_sphere, light_emit, mat_albedo, mat_emission, hdri;
	SceneDev scene{};
	CameraParams camera{};
	uint32_t trace_lds_bytes = 0;    // dynamic LDS of the BVH trace kernels (staged records + spheres)
	uint32_t bvh_depth = 0;
	bool allow_half = true;           // binary16 records when adequate (mirt_debug_set(ctx, "half_boxes", 0) forces f32)

	// frame state
	DeviceBuffer accumulator;        // [local tile][bucket][3][256] f32
	DeviceBuffer framebuffer;        // width*height float4
	float* frame_host = nullptr;     // pinned staging copy of the framebuffer for mirt_render (pageable memory halves the copy rate)
	size_t frame_host_bytes = 0;
	DeviceBuffer counters;           // DevCounters
	std::vector<PipeSlot> slots;     // batches in flight (policy.streams)
	uint64_t planned_for = 0;        // local pixel count batch_mem_cap was planned for (0 = plan again)
	uint32_t batch_mem_cap = 256;     // accumulations per batch the device memory allows (lowered by ensure_streams when the plan does not fit)
	uint32_t capacity = 0;           // rays per stream plane (= kSegs * seg_cap)
	uint32_t seg_cap = 0;            // slots per queue segment
	uint32_t arena_bounces = 0;
	uint64_t batch_seq = 0;

	uint32_t deferred = 0;            // Accumulate() calls accepted by mirt_accumulate_async but not launched yet (fewer than a batch)
	// launch-shape knobs for measurements (profiles/gpu_cycle.sh A/B runs), read from the environment at mirt_create: MIRT_TUNE_TRACE_WGS /
	// MIRT_TUNE_SHADE_WGS = workgroups per CU, MIRT_TUNE_CHUNK = rays per work reservation, MIRT_TUNE_LEAF_BATCH = lanes at a leaf that
	// trigger a leaf pass, MIRT_TUNE_REFILL_IDLE = idle lanes that trigger a refill.  They never change results.
	uint32_t tune_trace_wgs = 2, tune_shade_wgs = 3, tune_chunk = kChunkMax, tune_leaf_batch = kLeafBatch, tune_refill_idle = kRefillIdle, tune_wide = 1, tune_zero_ahead = 1;
	// profiling
	std::vector<TimedLaunch> pending;
	std::vector<hipEvent_t> free_events;
	mirt_kernel_times times{};
};

namespace {

int fail(mirt_ctx* ctx, int code, const char* fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	if (ctx) ctx->error = buf; else g_create_error = buf;
	return code;
}
#define HIP_TRY(ctx, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(ctx, MIRT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); } while (0)

// Accumulations traced together as one batch (path id = (slot << pix_bits) | pixel below 2^30: bits 30 and 31 stay free for flags).
// Every launch of a batch ends in a tail while its longest rays finish and starts with the staging of the tree top, so
// launches want to be LARGE — and 288 GB of HBM is there to be used for ray streams (180 B per ray and batch in flight).
// Measured on one MI355X (Mray/s; accumulations per batch x batches in flight on separate HIP streams):
//   cfg4 4096^2 S(100000):  5 x 3: 4919   8 x 1: 5435   16 x 1: 5726   32 x 1: 5838   16 x 2: 5013
//   one eighth of cfg4:     16 x 3: 4367  32 x 3: 4981  64 x 3: 5103   64 x 1: 5498
//   cfg3 1920x1088 S(10000): 16 x 3: 5378  32 x 3: 5753  64 x 3: 5700  64 x 1: 6354
//   cfg2 1024^2 S(1000):    32 x 3: 7510  64 x 3: 8011  64 x 1: 7654
// Hence: aim at 1 G primary rays per batch (round 2: 512 M; see kBatchRays) (at most max_slots() accumulations, at most what the free device memory holds), and keep
// ONE batch in flight once a batch carries 96 M primary rays or more — launches that fill the chip for milliseconds gain
// nothing from sharing it with another stream's kernels and lose to the interleaving — three below that (other batches fill the tails).
constexpr uint32_t kMaxBatch = 256;         // upper limit of accumulations per batch; a context's own limit is what its path ids hold (max_slots)
constexpr uint64_t kBatchRays = 1024ull << 20;   // round 3, cfg4 on the whole image (Mray/s, one batch in flight): 32 accumulations per batch 7970, 48: 8154, 63: 8218; 2 x 32: 7232, 2 x 24: 7192
constexpr uint64_t kSerialRays = 96ull << 20;
constexpr size_t kStreamPlanes = 2 * 13 + 2 + 17;      // two ray streams, hit (tfar, prim), shadow stream: 4-byte planes per ray of capacity
// Path id = (batch slot << pix_bits) | local pixel, below 2^30 (bits 30 and 31 of the words that carry it are flags): a context that owns
// all 2^24 pixels of a 4096^2 image has room for 64 slots, one that owns an eighth of it (a rank of an 8-GPU run) for 512.
uint32_t pix_bits_of(const mirt_ctx* c) { uint32_t b = 8; while ((1ull << b) < static_cast<uint64_t>(c->n_tiles) * kTileSize) b++; return b; }
// ... and a stream slot must stay below 2^30 as well (two flag bits in the words that carry it): pixels x slots + the padding of the kSegs queue segments
uint32_t max_slots(const mirt_ctx* c) {
	const uint64_t n_pix = std::max<uint64_t>(static_cast<uint64_t>(c->n_tiles) * kTileSize, 1);
	const uint64_t by_slots = ((1ull << 30) - 3ull * kSegs * kShadeBlock) / n_pix;
	return static_cast<uint32_t>(std::max<uint64_t>(std::min<uint64_t>(std::min<uint64_t>(kMaxBatch, 1u << (30u - std::min<uint32_t>(pix_bits_of(c), 24u))), by_slots), 1));
}
uint32_t batch_floor(const mirt_ctx* c) { return std::max<uint32_t>(std::min<uint32_t>(c->policy.buckets, 5u), 1u); }   // the reference's natural group: five calls, five buckets
uint32_t batch_limit(const mirt_ctx* c) {
	if (c->policy.max_batch) return std::min(c->policy.max_batch, max_slots(c));
	const uint64_t n_pix = static_cast<uint64_t>(c->n_tiles) * kTileSize;
	if (n_pix == 0) return 1;
	const uint64_t b = std::max<uint64_t>((kBatchRays + n_pix / 2) / n_pix, batch_floor(c));
	return static_cast<uint32_t>(std::min<uint64_t>(std::min<uint64_t>(b, max_slots(c)), std::max<uint32_t>(c->batch_mem_cap, 1u)));
}
// Paths add straight into the accumulator only when a batch cannot touch a (pixel, bucket) word twice and no other
// batch is in flight; otherwise every batch adds into its own contribution buffer, merged in accumulation order.
bool uses_contrib(const mirt_ctx* c, uint32_t n_slots) { return n_slots > 1 || batch_limit(c) > c->policy.buckets; }

uint32_t grid_for(const mirt_ctx* c, uint64_t work_items) {
	uint64_t blocks = (work_items + kBlock - 1) / kBlock;
	const uint64_t cap = static_cast<uint64_t>(c->n_cu) * 8u;      // grid-stride beyond 8 workgroups per CU
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return static_cast<uint32_t>(blocks);
}

// Trace kernels are launched as a resident grid (as many 512-thread workgroups as their LDS footprint lets a
// CU hold) and grid-stride over the stream, so each workgroup stages the BVH into LDS once per launch.
constexpr uint32_t kLdsPerCu = 160u * 1024u;
// LDS plan of a trace workgroup (1024 lanes): staged BVH bytes + a per-lane traversal stack.
//   binary16 records, <= 32768 records and spheres: 16 u16 entries (32 KB) + up to 48 KB staged  -> TWO workgroups (32 waves) per CU
//   binary16 records, more of them:     12 u32 entries (48 KB) + up to 32 KB staged  -> two workgroups per CU
//   f32 records:                        16 u32 entries (64 KB) + up to 96 KB staged  -> one workgroup per CU
uint32_t stack_bytes(bool half, bool stack16) { return half ? (stack16 ? kLdsStack * kTraceBlock * 2u : kLdsStackWide * kTraceBlock * 4u) : kLdsStack * kTraceBlock * 4u; }
uint32_t stack_bytes(const mirt_ctx* c) { return stack_bytes(c->scene.half_boxes != 0, c->scene.stack16 != 0); }
uint32_t stage_budget(bool half, bool stack16) { return half ? (stack16 ? 48u * 1024u : 32u * 1024u) : 96u * 1024u; }
uint32_t trace_lds(const mirt_ctx* c) { return c->policy.use_bvh ? c->trace_lds_bytes + stack_bytes(c) : kBruteChunk * 16u; }
uint32_t trace_grid(const mirt_ctx* c, uint64_t work_items) {
	uint32_t per_cu = kLdsPerCu / trace_lds(c);
	if (per_cu > c->tune_trace_wgs) per_cu = c->tune_trace_wgs;      // 2 x 1024 threads = 32 waves, the CU's limit
	if (per_cu < 1) per_cu = 1;
	uint64_t blocks = (work_items + kTraceBlock - 1) / kTraceBlock;
	const uint64_t cap = static_cast<uint64_t>(c->n_cu) * per_cu;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return static_cast<uint32_t>(blocks);
}

hipError_t sync_all(mirt_ctx* c) {
	for (PipeSlot& sl : c->slots) if (sl.stream) { hipError_t e = hipStreamSynchronize(sl.stream); if (e != hipSuccess) return e; }
	if (c->side_stream) { hipError_t e = hipStreamSynchronize(c->side_stream); if (e != hipSuccess) return e; }
	return hipStreamSynchronize(c->stream);
}
constexpr uint32_t kQueueWords = kSegs * kSegPitch;     // one ray queue's counters
size_t counts_words(uint32_t nb) { return static_cast<size_t>(2 * nb + 2) * kQueueWords + static_cast<size_t>(nb) * 4 + 8; }
constexpr uint32_t kFatCapacity = 1u << 16;   // rays per list and launch that may take the brute-force detour (a few per million qualify)
uint32_t wanted_slots(const mirt_ctx* c) {
	if (c->policy.streams) return std::min<uint32_t>(c->policy.streams, 8u);
	return static_cast<uint64_t>(c->n_tiles) * kTileSize * batch_limit(c) >= kSerialRays ? 1u : 3u;
}
// Device bytes of the batches in flight for the current plan (ray streams + contribution buffers).
uint64_t streams_bytes(const mirt_ctx* c) {
	const uint64_t rays = static_cast<uint64_t>(c->n_tiles) * kTileSize * batch_limit(c);
	return wanted_slots(c) * (rays * 4u * kStreamPlanes + (uses_contrib(c, wanted_slots(c)) ? rays * 12u : 0u));
}

// Automatic batch size: as large as kBatchRays asks, but within 80 % of the device memory that is free plus what this context's own ray
// streams hold already (other contexts and processes may share the device).  Allocates nothing: mirt_get_policy reports the plan before
// the first launch; ensure_streams re-carves the arena when the plan changed.  Planned once per pixel count (mirt_set_policy resets it
// when a field the plan depends on changes).
void plan_batches(mirt_ctx* c) {
	const uint64_t n_pix = static_cast<uint64_t>(c->n_tiles) * kTileSize;
	if (c->policy.max_batch || n_pix == 0 || c->planned_for == n_pix) return;
	c->batch_mem_cap = kMaxBatch;
	size_t free_b = 0, total_b = 0, own = 0;
	for (const PipeSlot& sl : c->slots) own += sl.arena.bytes + sl.contrib.bytes;
	if (hipSetDevice(c->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
		while (batch_limit(c) > batch_floor(c) && streams_bytes(c) > (free_b + own) / 5 * 4) c->batch_mem_cap = std::max(batch_limit(c) / 2, batch_floor(c));
	c->planned_for = n_pix;
}

// Carve each slot's frame-wide ray streams out of one allocation.
int ensure_streams(mirt_ctx* c) {
	const uint64_t n_pix = static_cast<uint64_t>(c->n_tiles) * kTileSize;
	if (n_pix * batch_limit(c) == 0) return MIRT_OK;
	plan_batches(c);
	const uint64_t cap64 = n_pix * batch_limit(c);
	if (n_pix > (1u << 24)) return fail(c, MIRT_ERR_ARG, "more than 2^24 pixels per context (%llu); shard the tile range", (unsigned long long)n_pix);
	if (cap64 + 3ull * kSegs * kShadeBlock > (1ull << 30)) return fail(c, MIRT_ERR_ARG, "stream capacity %llu too large", (unsigned long long)cap64);   // stream slots carry two flag bits (kDestAccum, kDestFull)
	// a ray queue is kSegs segments of seg_cap slots (kernels.hpp "ray queues"): a plane holds kSegs * seg_cap entries
	// (+ 2 blocks: k_shade<FIRST> iterates pixel-major, and when the pixel count is an odd multiple of 256 its half-filled last chunk adds iterations)
	const uint32_t seg_cap = static_cast<uint32_t>((((cap64 + kShadeBlock - 1) / kShadeBlock + kSegs - 1) / kSegs + 2) * kShadeBlock);
	const uint32_t cap = seg_cap * kSegs;
	const uint32_t nb = c->policy.max_bounces;
	const uint32_t want = wanted_slots(c);
	const bool contrib = uses_contrib(c, want);
	const size_t acc_bytes = static_cast<size_t>(c->n_tiles) * batch_limit(c) * 3 * kTileSize * sizeof(float);     // contribution buffer: [tile][slot][rgb][256]
	if (cap == c->capacity && nb == c->arena_bounces && c->slots.size() == want && (!contrib || c->slots[0].contrib.bytes >= acc_bytes)) return MIRT_OK;
	HIP_TRY(c, sync_all(c));
	while (c->slots.size() > want) {
		PipeSlot& sl = c->slots.back();
		sl.arena.release(); sl.counts.release(); sl.contrib.release(); sl.fat.release(); sl.cand.release();
		if (sl.batch_done) (void)hipEventDestroy(sl.batch_done);
		if (sl.merged) (void)hipEventDestroy(sl.merged);
		if (sl.zeroed) (void)hipEventDestroy(sl.zeroed);
		if (sl.stream) (void)hipStreamDestroy(sl.stream);
		c->slots.pop_back();
	}
	while (c->slots.size() < want) {
		PipeSlot sl;
		HIP_TRY(c, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
		HIP_TRY(c, hipEventCreateWithFlags(&sl.batch_done, hipEventDisableTiming));
		HIP_TRY(c, hipEventCreateWithFlags(&sl.merged, hipEventDisableTiming));
		HIP_TRY(c, hipEventCreateWithFlags(&sl.zeroed, hipEventDisableTiming));
		c->slots.push_back(sl);
	}
	const size_t planes = kStreamPlanes;
	const size_t plane_bytes = (static_cast<size_t>(cap) * 4 + 255) & ~static_cast<size_t>(255);
	for (PipeSlot& sl : c->slots) {
		sl.in_use = false; sl.contrib_is_zero = false; sl.zero_pending = false;      // (the caller has synchronised every stream)
		hipError_t e = sl.arena.ensure(planes * plane_bytes);
		const char* what = "ray streams";
		if (e == hipSuccess) { e = sl.counts.ensure(counts_words(nb) * sizeof(uint32_t)); what = "queue counters"; }
		if (e == hipSuccess) { e = sl.fat.ensure(2u * kFatCapacity * sizeof(uint32_t)); what = "fat-ray lists"; }
		if (e == hipSuccess) { e = sl.cand.ensure(static_cast<size_t>(n_pix) * kCandStride * sizeof(uint32_t)); what = "candidate lists"; }
		if (e == hipSuccess) { if (contrib) { e = sl.contrib.ensure(acc_bytes); what = "contribution buffer"; } else sl.contrib.release(); }
		if (e != hipSuccess) {
			// not enough device memory after all (someone else took it meanwhile — another context or process planning against the same
			// free memory): halve the automatic batch and plan again
			(void)hipGetLastError();
			if (e == hipErrorOutOfMemory && !c->policy.max_batch && batch_limit(c) > batch_floor(c)) {
				for (PipeSlot& other : c->slots) { other.arena.release(); other.contrib.release(); }
				c->capacity = 0;
				c->batch_mem_cap = std::max(batch_limit(c) / 2, batch_floor(c));
				return ensure_streams(c);
			}
			return fail(c, MIRT_ERR_HIP, "%s (%zu bytes of ray streams per batch in flight): %s", what, planes * plane_bytes, hipGetErrorString(e));
		}
		char* p = sl.arena.as<char>();
		auto take = [&]() { void* r = p; p += plane_bytes; return r; };
		for (int b = 0; b < 2; b++) {
			StreamBuf& s = sl.stream_buf[b];
			s.px = (float*)take(); s.py = (float*)take(); s.pz = (float*)take();
			s.dx = (float*)take(); s.dy = (float*)take(); s.dz = (float*)take();
			s.tr = (float*)take(); s.tg = (float*)take(); s.tb = (float*)take();
			s.rr = (float*)take(); s.rg = (float*)take(); s.rb = (float*)take();
			s.path = (uint32_t*)take();
		}
		sl.hit = (HitRec*)take(); (void)take();                            // planes are adjacent: 8 B per ray
		ShadowBuf& h = sl.shadow_buf;
		h.px = (float*)take(); h.py = (float*)take(); h.pz = (float*)take();
		h.dx = (float*)take(); h.dy = (float*)take(); h.dz = (float*)take(); h.tfar = (float*)take();
		h.sr = (float*)take(); h.sg = (float*)take(); h.sb = (float*)take();
		h.rr = (float*)take(); h.rg = (float*)take(); h.rb = (float*)take();
		h.er = (float*)take(); h.eg = (float*)take(); h.eb = (float*)take();
		h.dest = (uint32_t*)take();
	}
	c->capacity = cap;
	c->seg_cap = seg_cap;
	c->arena_bounces = nb;
	return MIRT_OK;
}

int alloc_accumulator(mirt_ctx* c) {
	const size_t floats = static_cast<size_t>(c->n_tiles) * c->policy.buckets * 3 * kTileSize;
	HIP_TRY(c, sync_all(c));
	HIP_TRY(c, c->accumulator.ensure(floats * sizeof(float)));
	if (floats) HIP_TRY(c, hipMemsetAsync(c->accumulator.ptr, 0, floats * sizeof(float), c->stream));
	HIP_TRY(c, hipMemsetAsync(c->counters.ptr, 0, sizeof(DevCounters), c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	for (PipeSlot& sl : c->slots) sl.in_use = false;
	c->accumulations = 0;
	return MIRT_OK;
}

// ---- launch bracketing (policy.profile) ---------------------------------------------------------
hipEvent_t take_event(mirt_ctx* c) {
	if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
	hipEvent_t e = nullptr;
	(void)hipEventCreate(&e);
	return e;
}
void harvest(mirt_ctx* c) {
	if (c->pending.empty()) return;
	(void)sync_all(c);
	for (TimedLaunch& t : c->pending) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) { c->times.ms[t.klass] += ms; c->times.launches[t.klass]++; }
		c->free_events.push_back(t.start); c->free_events.push_back(t.stop);
	}
	c->pending.clear();
}
struct Bracket {
	mirt_ctx* c; int klass; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
	Bracket(mirt_ctx* ctx, int k, hipStream_t stream = nullptr) : c(ctx), klass(k), st(stream ? stream : ctx->stream) {
		if (c->policy.profile) { a = take_event(c); b = take_event(c); (void)hipEventRecord(a, st); }
	}
	~Bracket() {
		if (c->policy.profile) { (void)hipEventRecord(b, st); c->pending.push_back(TimedLaunch{ klass, a, b }); }
	}
};

float bundle_half_angle(const mirt_ctx* c) { return (0.7072f / std::fabs(c->camera.z)) * 1.01f + 1e-4f; }

FrameParams frame_params(const mirt_ctx* c, uint32_t acc_base, uint32_t batch_n) {
	FrameParams fp{};
	fp.cam = c->camera;
	fp.h_tiles = c->h_tiles;
	fp.first_tile = c->first_tile;
	fp.run_tiles = c->run_tiles ? c->run_tiles : 1u;
	fp.stride_tiles = c->stride_tiles;
	fp.n_pix = c->n_tiles * kTileSize;
	fp.pix_bits = pix_bits_of(c); fp.pix_mask = (1u << fp.pix_bits) - 1u;
	fp.acc_base = acc_base;
	fp.batch_n = batch_n;
	fp.idx_base = acc_base;
	fp.idx_buckets = c->policy.buckets;
	fp.max_bounces = c->policy.max_bounces;
	fp.buckets = c->policy.buckets;
	fp.n_lights = c->scene.n_lights;
	fp.mis = (c->policy.mis && c->scene.n_lights > 0) ? 1u : 0u;       // Q12 guard
	fp.first_groups = 1;
	fp.inv_n_pix = fp.n_pix ? 1.0f / static_cast<float>(fp.n_pix) : 0.0f;
	fp.inv_h_tiles = fp.h_tiles ? 1.0f / static_cast<float>(fp.h_tiles) : 0.0f;
	fp.inv_run_tiles = 1.0f / static_cast<float>(fp.run_tiles);
	return fp;
}

// One batch = up to batch_limit() consecutive Accumulate() calls traced together (path id = (slot << pix_bits) | pixel).
// Consecutive accumulation indices land in buckets (acc % buckets, Renderer.hpp:82), and the ORDER of the adds into a
// bucket word is part of the result.  With at most `buckets` accumulations per batch and one batch at a time every
// (pixel, bucket) word is touched once per batch and paths add straight into the accumulator.  Otherwise each batch adds
// into a zeroed contribution buffer [tile][slot][rgb][256] and k_merge_contrib, enqueued on the main stream in batch
// order, applies the slots to their buckets in ascending order — so larger batches (launches several times longer than
// their tails) and up to policy.streams batches in flight on their own HIP streams (other batches fill those tails)
// leave every bucket's add order, hence the result, exactly as in the reference.
int launch_batch(mirt_ctx* c, uint32_t batch_n) {
	FrameParams fp = frame_params(c, c->accumulations, batch_n);
	const uint32_t nb = c->policy.max_bounces;
	const uint64_t total = static_cast<uint64_t>(fp.n_pix) * batch_n;
	if (total == 0) { c->accumulations += batch_n; return MIRT_OK; }                   // no tile owned (an image below 16 px, a group member beyond the last tile row): ++accumulations over an empty parallel_for, Renderer.hpp:74-75
	const bool pipelined = c->slots.size() > 1;
	PipeSlot& sl = c->slots[c->batch_seq % c->slots.size()];
	hipStream_t st = pipelined ? sl.stream : c->stream;
	const bool contrib = uses_contrib(c, static_cast<uint32_t>(c->slots.size()));
	const size_t contrib_floats = static_cast<size_t>(c->n_tiles) * batch_n * 3 * kTileSize;     // [tile][slot][rgb][256]
	if (contrib) { fp.idx_base = 0xffffffffu; fp.idx_buckets = batch_n; }            // bucket of slot k inside the buffer = k
	uint32_t* counts = sl.counts.as<uint32_t>();
	auto stream_queue = [&](uint32_t b) { return Queue{ counts + static_cast<size_t>(b) * kQueueWords, c->seg_cap }; };                 // rays entering bounce b (b >= 1)
	auto shadow_queue = [&](uint32_t b) { return Queue{ counts + static_cast<size_t>(nb + 1 + b) * kQueueWords, c->seg_cap }; };        // NEE rays emitted at bounce b
	const Queue empty_queue{ counts + static_cast<size_t>(2 * nb + 1) * kQueueWords, c->seg_cap };                                      // "no shadow rays pending"
	uint32_t* work_next = counts + static_cast<size_t>(2 * nb + 2) * kQueueWords;     // per-launch work counters of the persistent trace kernels
	uint32_t* work_next_shadow = work_next + nb;
	uint32_t* fat_n_closest = work_next_shadow + nb;    // per-launch fat-ray counts (closest-hit list, shadow list)
	uint32_t* fat_n_shadow = fat_n_closest + nb;
	uint32_t* misc = fat_n_shadow + nb;                 // [0] work counter of k_primary_cand, [1] an unused fat-ray count, [2] pixels without a candidate list (their list: in.path of bounce 0)
	// Camera rays of a batch go through per-pixel candidate lists when a pixel is sampled often enough to pay for its cone traversal
	// (policy.trace_primary_rays = 1 switches that off: every primary ray then walks the tree; results are identical either way).
	// (The half-angle bound assumes view.orient rotates: a non-unit quaternion, which the reference's View never holds (Camera.hpp:48-50),
	// would shear the image plane — such a camera gets no lists.)
	const float qn = c->camera.orient[0] * c->camera.orient[0] + c->camera.orient[1] * c->camera.orient[1] + c->camera.orient[2] * c->camera.orient[2] + c->camera.orient[3] * c->camera.orient[3];
	const bool bundle = c->policy.use_bvh && c->scene.n_recs != 0 && !c->policy.trace_primary_rays && batch_n >= 3 && c->camera.z != 0.0f && std::fabs(qn - 1.0f) < 1e-4f;
	// half-angle of a pixel's bundle: half a pixel diagonal (0.7072) at distance >= |z|; + 1e-4: a sample's own cone half-width (1.38e-3 ..
	// 1.47e-3, from |D|^2 - 1 of its normalised direction) may exceed the axis ray's by 8.5e-5
	const float rho = bundle ? bundle_half_angle(c) : 0.0f;
	DevCounters* ctr = c->counters.as<DevCounters>();
	float* accum = contrib ? sl.contrib.as<float>() : c->accumulator.as<float>();
	SceneDev sc = c->scene;
	sc.use_bvh = c->policy.use_bvh;
	sc.chunk_max = c->tune_chunk; sc.leaf_batch = c->tune_leaf_batch; sc.refill_idle = c->tune_refill_idle;
	const bool count = c->policy.count_traffic != 0;
	const uint32_t tgrid = trace_grid(c, total);
	const uint32_t sgrid = static_cast<uint32_t>(std::min<uint64_t>((total + kShadeBlock - 1) / kShadeBlock, static_cast<uint64_t>(c->n_cu) * c->tune_shade_wgs));     // three 512-thread workgroups are resident per CU (k_shade: ~80 VGPRs); more only adds passes
	const uint32_t tlds = trace_lds(c);
	{	// k_shade<FIRST> hands out (512-pixel chunk, group of accumulations) pieces: at least ~8 per workgroup, so that a small image loads the grid evenly
		const uint64_t n_chunks = (static_cast<uint64_t>(fp.n_pix) + kShadeBlock - 1) / kShadeBlock;
		const uint64_t want = (8ull * sgrid + n_chunks - 1) / n_chunks;
		fp.first_groups = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(want, 1), batch_n));
	}
	// one workgroup per fat ray: a large scene (100 k spheres: ~100 us per ray) wants as many of them in flight as there are rays (a few hundred per launch)
	const uint32_t fat_grid = sc.n_spheres > 4096 ? static_cast<uint32_t>(c->n_cu) * 2u : 64u;

	if (pipelined && sl.in_use) HIP_TRY(c, hipStreamWaitEvent(st, sl.merged, 0));   // the slot's previous batch has been merged: buffers are free
	// The contribution buffer must hold +0 where this batch's paths add.  With one batch in flight it is zeroed AHEAD of time: once a batch has been
	// merged, the side stream clears what that batch dirtied (a prefix: [tile][slot][rgb][256] with the batch's slot count) while the next batch's
	// camera-ray kernels, which do not touch it, run on the main stream (12.7 GB = 4 ms per 63-accumulation batch of cfg4 that no longer sit between
	// two batches); k_shade of bounce 0, the first writer, waits for that.  Several batches in flight: zeroed in line, as before.
	const bool zero_ahead = contrib && !pipelined && c->tune_zero_ahead;
	if (contrib && !(zero_ahead && sl.contrib_is_zero)) {
		HIP_TRY(c, hipMemsetAsync(sl.contrib.ptr, 0, zero_ahead ? sl.contrib.bytes : contrib_floats * sizeof(float), st));
		sl.contrib_is_zero = zero_ahead; sl.zero_pending = false;
	}
	HIP_TRY(c, hipMemsetAsync(counts, 0, counts_words(nb) * sizeof(uint32_t), st));
	// bounce 0 has no ray stream: k_trace<PRIMARY> and k_shade<FIRST> derive the camera ray from its index (RAY GENERATION, Renderer.hpp:113-127)
	for (uint32_t bounce = 0; bounce < nb; bounce++) {
		const StreamBuf& in = sl.stream_buf[bounce & 1u];
		const StreamBuf& out = sl.stream_buf[(bounce & 1u) ^ 1u];
		const bool shadow_pending = fp.mis && bounce > 0;           // NEE rays emitted by k_shade(bounce-1)
		{ Bracket t(c, MIRT_K_TRACE, st);
		  const Queue sq = shadow_pending ? shadow_queue(bounce - 1) : empty_queue;
		  uint32_t* sc_work = work_next_shadow + (shadow_pending ? bounce - 1 : 0);
		  const FatList fc{ fat_n_closest + bounce, sl.fat.as<uint32_t>(), kFatCapacity };
		  const FatList fs{ fat_n_shadow + bounce, sl.fat.as<uint32_t>() + kFatCapacity, kFatCapacity };
		  // the adds of bounce-1 that waited for occlusion land in stream `in` (= out of bounce-1) or the accumulator, before k_shade reads them
		  const ShadowSink sink{ in.rr, in.rg, in.rb, in.px, in.py, in.pz, accum, fp.idx_base, fp.idx_buckets, fp.pix_bits, nullptr };
		  const Queue cq = (bounce == 0 && bundle) ? Queue{ misc + 2, 0u } : stream_queue(bounce);      // kPrimaryList: n[0] = listed pixels
		  auto launch_trace = [&](auto kernel, auto fat_kernel) {
		    hipLaunchKernelGGL(kernel, dim3(tgrid), dim3(kTraceBlock), tlds, st, sc, fp, in, sl.hit, cq, work_next + bounce,
		                       sl.shadow_buf, sink, sq, sc_work, fc, fs, ctr);
		    // the few rays too "fat" for the tree: brute force, one workgroup each
		    if (sc.use_bvh) hipLaunchKernelGGL(fat_kernel, dim3(fat_grid), dim3(1024), 0, st, sc, fp, in, sl.hit, fc, sl.shadow_buf, sink, fs, ctr, misc + 2);
		  };
		  if (bounce == 0 && bundle) {
		    // camera rays through per-pixel candidate lists (kernels.hpp kCollect): one cone traversal per pixel, then k_primary_hits intersects every
		    // sample with its pixel's list.  Pixels without a list are listed in in.path (count: misc[2]) and all their samples traced like any other ray
		    const FatList none{ misc + 1, sl.fat.as<uint32_t>(), 0u };
		    if (count) hipLaunchKernelGGL(k_primary_cand<true>, dim3(trace_grid(c, fp.n_pix)), dim3(kTraceBlock), tlds, st, sc, fp, sl.cand.as<uint32_t>(), rho, misc, none, ctr, in.path, misc + 2);
		    else       hipLaunchKernelGGL(k_primary_cand<false>, dim3(trace_grid(c, fp.n_pix)), dim3(kTraceBlock), tlds, st, sc, fp, sl.cand.as<uint32_t>(), rho, misc, none, ctr, in.path, misc + 2);
		    const uint32_t hgrid = static_cast<uint32_t>(std::min<uint64_t>((static_cast<uint64_t>(fp.n_pix) + kBlock - 1) / kBlock, static_cast<uint64_t>(c->n_cu) * 64u));
		    if (count) hipLaunchKernelGGL(k_primary_hits<true>, dim3(hgrid), dim3(kBlock), 0, st, sc, fp, sl.cand.as<uint32_t>(), sl.hit, ctr);
		    else       hipLaunchKernelGGL(k_primary_hits<false>, dim3(hgrid), dim3(kBlock), 0, st, sc, fp, sl.cand.as<uint32_t>(), sl.hit, ctr);
		    if (count) launch_trace((k_trace<true, kPrimaryList>), (k_trace_fat<true, kPrimaryList>)); else launch_trace((k_trace<false, kPrimaryList>), (k_trace_fat<false, kPrimaryList>));
		  }
		  else if (bounce == 0) { if (count) launch_trace((k_trace<true, kPrimaryAll>), (k_trace_fat<true, kPrimaryAll>)); else launch_trace((k_trace<false, kPrimaryAll>), (k_trace_fat<false, kPrimaryAll>)); }
		  else                  { if (count) launch_trace((k_trace<true, kPrimaryNone>), (k_trace_fat<true, kPrimaryNone>)); else launch_trace((k_trace<false, kPrimaryNone>), (k_trace_fat<false, kPrimaryNone>)); } }
		if (bounce == 0 && zero_ahead && sl.zero_pending) { HIP_TRY(c, hipStreamWaitEvent(st, sl.zeroed, 0)); sl.zero_pending = false; }
		{ Bracket t(c, MIRT_K_SHADE, st);
		  if (bounce == 0) hipLaunchKernelGGL(k_shade<true>, dim3(sgrid), dim3(kShadeBlock), 0, st, sc, fp, in, sl.hit, out, sl.shadow_buf, bounce, stream_queue(bounce), stream_queue(bounce + 1), shadow_queue(bounce), accum, ctr);
		  else             hipLaunchKernelGGL(k_shade<false>, dim3(sgrid), dim3(kShadeBlock), 0, st, sc, fp, in, sl.hit, out, sl.shadow_buf, bounce, stream_queue(bounce), stream_queue(bounce + 1), shadow_queue(bounce), accum, ctr); }
	}
	HIP_TRY(c, hipGetLastError());
	if (contrib) {
		// merges are enqueued on the main stream in batch order => every bucket receives its adds in accumulation order
		if (pipelined) {
			HIP_TRY(c, hipEventRecord(sl.batch_done, st));
			HIP_TRY(c, hipStreamWaitEvent(c->stream, sl.batch_done, 0));
		}
		{ Bracket t(c, MIRT_K_RESOLVE);
		  hipLaunchKernelGGL(k_merge_contrib, dim3(grid_for(c, static_cast<uint64_t>(c->n_tiles) * (3u * kTileSize / 4u))), dim3(kBlock), 0, c->stream,
		                     c->accumulator.as<float4>(), sl.contrib.as<float4>(), c->n_tiles, c->policy.buckets, batch_n, fp.acc_base); }
		HIP_TRY(c, hipGetLastError());
		if (pipelined) { HIP_TRY(c, hipEventRecord(sl.merged, c->stream)); sl.in_use = true; }
		if (zero_ahead) {
			HIP_TRY(c, hipEventRecord(sl.merged, c->stream));
			HIP_TRY(c, hipStreamWaitEvent(c->side_stream, sl.merged, 0));
			HIP_TRY(c, hipMemsetAsync(sl.contrib.ptr, 0, contrib_floats * sizeof(float), c->side_stream));
			HIP_TRY(c, hipEventRecord(sl.zeroed, c->side_stream));
			sl.zero_pending = true;                                              // contrib_is_zero stays true: it will be by the time anyone may write
		}
	}
	c->batch_seq++;
	c->accumulations += batch_n;
	if (c->policy.profile && c->pending.size() > 512) harvest(c);
	return MIRT_OK;
}

template <class T>
int upload(mirt_ctx* c, DeviceBuffer& buf, const std::vector<T>& host) {
	HIP_TRY(c, buf.ensure(std::max<size_t>(host.size(), 1) * sizeof(T)));
	if (!host.empty()) HIP_TRY(c, hipMemcpyAsync(buf.ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
	return MIRT_OK;
}

int check_ready(mirt_ctx* c) {
	if (!c) return MIRT_ERR_ARG;
	if (!c->have_scene) return fail(c, MIRT_ERR_STATE, "mirt_set_scene has not been called");
	if (!c->have_camera) return fail(c, MIRT_ERR_STATE, "mirt_set_camera has not been called");
	if (c->width == 0 && c->height == 0) return fail(c, MIRT_ERR_STATE, "mirt_resize has not been called");
	return MIRT_OK;
}

// Launches what mirt_accumulate_async has deferred.  Every entry point that observes results or changes what a launch
// reads (scene, camera, policy, size, tile range) calls this first, so deferral is invisible except in timing.
int flush_deferred(mirt_ctx* c) {
	if (!c->deferred) return MIRT_OK;
	const uint32_t n = c->deferred;
	c->deferred = 0;
	HIP_TRY(c, hipSetDevice(c->device));
	int r = ensure_streams(c); if (r) return r;
	return launch_batch(c, n);
}

} // namespace

extern "C" {

const char* mirt_last_error(const mirt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int mirt_create(int device, mirt_ctx** out) {
	if (!out) return fail(nullptr, MIRT_ERR_ARG, "out is NULL");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) return fail(nullptr, MIRT_ERR_NO_DEVICE, "no HIP device available (%s); mirt has no CPU path", hipGetErrorString(e));
	if (device < 0 || device >= n) return fail(nullptr, MIRT_ERR_ARG, "device %d out of range (have %d)", device, n);
	e = hipSetDevice(device);
	if (e != hipSuccess) return fail(nullptr, MIRT_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
	mirt_ctx* c = new mirt_ctx();
	c->device = device;
	if (const char* e = std::getenv("MIRT_TUNE_CHUNK")) c->tune_chunk = static_cast<uint32_t>(std::min(std::max(std::atoi(e), 64), 65536)) & ~63u;
	if (const char* e = std::getenv("MIRT_TUNE_REFILL_IDLE")) c->tune_refill_idle = static_cast<uint32_t>(std::min(std::max(std::atoi(e), 1), 64));
	if (const char* e = std::getenv("MIRT_TUNE_ZERO_AHEAD")) c->tune_zero_ahead = std::atoi(e) != 0 ? 1u : 0u;
	if (const char* e = std::getenv("MIRT_TUNE_WIDE")) c->tune_wide = std::atoi(e) != 0 ? 1u : 0u;
	if (const char* e = std::getenv("MIRT_TUNE_LEAF_BATCH")) c->tune_leaf_batch = static_cast<uint32_t>(std::min(std::max(std::atoi(e), 1), 64));
	if (const char* e = std::getenv("MIRT_TUNE_TRACE_WGS")) c->tune_trace_wgs = std::atoi(e) == 1 ? 1u : 2u;
	if (const char* e = std::getenv("MIRT_TUNE_SHADE_WGS")) c->tune_shade_wgs = static_cast<uint32_t>(std::min(std::max(std::atoi(e), 1), 16));
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = c->counters.ensure(sizeof(DevCounters));
	if (e == hipSuccess) e = hipMemsetAsync(c->counters.ptr, 0, sizeof(DevCounters), c->stream);
	if (e != hipSuccess) { int r = fail(nullptr, MIRT_ERR_HIP, "context setup: %s", hipGetErrorString(e)); delete c; return r; }
	*out = c;
	return MIRT_OK;
}

int mirt_destroy(mirt_ctx* c) {
	if (!c) return MIRT_ERR_ARG;
	(void)hipSetDevice(c->device);
	(void)sync_all(c);
	harvest(c);
	for (PipeSlot& sl : c->slots) {
		sl.arena.release(); sl.counts.release(); sl.contrib.release(); sl.fat.release(); sl.cand.release();
		if (sl.batch_done) (void)hipEventDestroy(sl.batch_done);
		if (sl.merged) (void)hipEventDestroy(sl.merged);
		if (sl.zeroed) (void)hipEventDestroy(sl.zeroed);
		if (sl.stream) (void)hipStreamDestroy(sl.stream);
	}
	c->slots.clear();
	for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
	DeviceBuffer* bufs[] = { &c->recs, &c->recs_wide, &c->spheres, &c->prim_mat, &c->light_sphere, &c->light_emit, &c->mat_albedo, &c->mat_emission,
	                         &c->hdri, &c->accumulator, &c->framebuffer, &c->counters };
	for (DeviceBuffer* b : bufs) b->release();
	if (c->frame_host) (void)hipHostFree(c->frame_host);
	if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
	return MIRT_OK;
}

int mirt_set_scene(mirt_ctx* c, const mirt_sphere* geometry, const mirt_sphere* bvh_prims, uint32_t n_spheres,
                   const mirt_bvh_node* nodes, uint32_t n_nodes, const mirt_material* materials, uint32_t n_materials,
                   const int32_t* lights, uint32_t n_lights, const float ambient_color[3],
                   const float* hdri_rgba, uint32_t hdri_w, uint32_t hdri_h) {
	if (!c) return MIRT_ERR_ARG;
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	if (n_spheres && (!geometry || !bvh_prims)) return fail(c, MIRT_ERR_ARG, "geometry / bvh_prims is NULL");
	if (n_spheres >= (1u << 26)) return fail(c, MIRT_ERR_ARG, "more than 2^26 spheres");      // 32-bit record offsets in the trace kernels (GPU-built trees do not pass build_records)
	if (n_nodes && !nodes) return fail(c, MIRT_ERR_ARG, "nodes is NULL");
	if (!materials || n_materials == 0 || n_materials > MIRT_MAX_MATERIALS) return fail(c, MIRT_ERR_ARG, "need 1..%u materials, got %u", MIRT_MAX_MATERIALS, n_materials);
	if (n_lights && !lights) return fail(c, MIRT_ERR_ARG, "lights is NULL");
	if (!ambient_color || !hdri_rgba || hdri_w == 0 || hdri_h == 0) return fail(c, MIRT_ERR_ARG, "sky needs ambient_color and an hdri of at least 1x1");
	// Validate everything the kernels index with: an out-of-range id would fault the GPU.
	for (uint32_t i = 0; i < n_spheres; i++) {
		if (geometry[i].material_ID < 0 || static_cast<uint32_t>(geometry[i].material_ID) >= n_materials ||
		    bvh_prims[i].material_ID < 0 || static_cast<uint32_t>(bvh_prims[i].material_ID) >= n_materials)
			return fail(c, MIRT_ERR_ARG, "sphere %u: material_ID out of range", i);
		// non-finite centres or radii would reach the tree builders' sorts and the box arithmetic as NaN (the reference has no check; UB there)
		for (const mirt_sphere* s : { &geometry[i], &bvh_prims[i] })
			if (!std::isfinite(s->position[0]) || !std::isfinite(s->position[1]) || !std::isfinite(s->position[2]) || !std::isfinite(s->radius_sq) || s->radius_sq < 0.0f)
				return fail(c, MIRT_ERR_ARG, "sphere %u: position / radius_sq must be finite, radius_sq >= 0", i);
	}
	for (uint32_t i = 0; i < n_lights; i++)
		if (lights[i] < 0 || static_cast<uint32_t>(lights[i]) >= n_spheres) return fail(c, MIRT_ERR_ARG, "light %u: index out of range", i);
	{
		// children point forward and every node has at most one parent: the node array is a forest, so the breadth-first
		// re-layout (bvh_layout.hpp) visits at most n_nodes nodes (shared children would make it grow like Fibonacci numbers)
		std::vector<uint8_t> has_parent(n_nodes, 0);
		for (uint32_t i = 0; i < n_nodes; i++) {
			const mirt_bvh_node& nd = nodes[i];
			if (nd.prim_count == 0) {
				if (nd.first_id <= i || static_cast<uint64_t>(nd.first_id) + 1 >= n_nodes) return fail(c, MIRT_ERR_ARG, "node %u: child index %u invalid", i, nd.first_id);
				if (has_parent[nd.first_id] || has_parent[nd.first_id + 1]) return fail(c, MIRT_ERR_ARG, "node %u: child pair %u is referenced by more than one parent", i, nd.first_id);
				has_parent[nd.first_id] = has_parent[nd.first_id + 1] = 1;
			} else if (static_cast<uint64_t>(nd.first_id) + nd.prim_count > n_spheres) return fail(c, MIRT_ERR_ARG, "node %u: prim range out of bounds", i);
		}
	}
	if (n_spheres && n_nodes == 0) return fail(c, MIRT_ERR_ARG, "spheres without BVH nodes");
	HIP_TRY(c, hipSetDevice(c->device));

	std::vector<float4> sph(n_spheres), lsp(n_lights), lem(n_lights), alb(n_materials), emi(n_materials), sky(static_cast<size_t>(hdri_w) * hdri_h);
	std::vector<int32_t> pm(n_spheres);
	for (uint32_t i = 0; i < n_spheres; i++) {
		sph[i] = make_float4(bvh_prims[i].position[0], bvh_prims[i].position[1], bvh_prims[i].position[2], bvh_prims[i].radius_sq);
		pm[i] = bvh_prims[i].material_ID;
	}
	// NEE reads scene.geometry[light], then material[geometry[light].material_ID].emission (Renderer.hpp:261-263,283):
	// flattened to one table per light so the kernel makes two independent loads instead of four dependent ones
	for (uint32_t l = 0; l < n_lights; l++) {
		const mirt_sphere& g = geometry[lights[l]];
		const float* e = materials[g.material_ID].emission;
		lsp[l] = make_float4(g.position[0], g.position[1], g.position[2], g.radius_sq);
		float id_bits; const int32_t id = lights[l]; std::memcpy(&id_bits, &id, 4);
		lem[l] = make_float4(e[0], e[1], e[2], id_bits);
	}
	for (uint32_t i = 0; i < n_materials; i++) {
		alb[i] = make_float4(materials[i].albedo[0], materials[i].albedo[1], materials[i].albedo[2], 0.0f);
		emi[i] = make_float4(materials[i].emission[0], materials[i].emission[1], materials[i].emission[2], 0.0f);
	}
	std::memcpy(sky.data(), hdri_rgba, sky.size() * sizeof(float4));
	std::vector<float> recs;
	std::vector<uint32_t> half_recs, wide_recs;
	uint32_t depth = 0, n_recs = 0;
	bool half = false, wide = false, wide_gpu = false;
	int r;
	if ((r = upload(c, c->spheres, sph))) return r;
	const bool gpu_tree = c->policy.gpu_build && !c->policy.reference_tree && n_spheres >= 2;
	if (gpu_tree) {
		// the tree the kernels walk, built where it is used (lbvh_build.hip); binary16 records under the same conditions as on the host
		n_recs = n_spheres - 1;
		half = c->allow_half;
		for (uint32_t i = 0; half && i < n_spheres; i++) {
			const float rad = std::sqrt(sph[i].w);
			const float amax = std::fmax(std::fmax(std::fabs(sph[i].x), std::fabs(sph[i].y)), std::fabs(sph[i].z)) + rad * 1.0001f;
			if (amax > 60000.0f || 2.0f * rad < 8.0f * mirt_host::half_ulp_at(amax)) half = false;
		}
		HIP_TRY(c, c->recs.ensure(static_cast<size_t>(n_recs) * (half ? 32u : 64u)));
		const bool want_wide = half && c->tune_wide;                         // the 4-wide binary16 records as well (used when the tree is shallow enough)
		if (want_wide) HIP_TRY(c, c->recs_wide.ensure(static_cast<size_t>(n_recs) * 64u));
		std::string why;
		uint32_t n_wide = 0;
		if (!mirt_gpu::build_lbvh(c->stream, c->spheres.as<float4>(), n_spheres, half ? nullptr : c->recs.as<float>(), half ? c->recs.as<uint32_t>() : nullptr, &depth, &why,
		                          want_wide ? c->recs_wide.as<uint32_t>() : nullptr, &n_wide))
			return fail(c, MIRT_ERR_HIP, "GPU BVH build: %s", why.c_str());
		if (depth >= kStack) return fail(c, MIRT_ERR_ARG, "GPU-built BVH is %u levels deep (limit %u): set policy.gpu_build = 0 for this scene", depth, kStack);
		if (want_wide && n_wide && 3u * (depth / 2u) < kStack) { wide = true; wide_gpu = true; n_recs = n_wide; }     // depth counts the leaf level: depth / 2 = wide levels, rounded up
	} else {
		std::vector<mirt_bvh_node> own; std::vector<uint32_t> prim_of_slot;
		const bool caller_tree = c->policy.reference_tree || n_spheres == 0;
		if (caller_tree) {
			// traverse the caller's tree exactly as handed over (BVH.hpp:18-31 nodes over the BVH-order prims)
			own.assign(nodes, nodes + n_nodes);
			mirt_host::split_multi_prim_leaves(own);                         // the kernels know one-prim leaves only
		} else {
			// default: GPU-internal SAH tree over the same BVH-order prims (hit.primID keeps its meaning; results are identical)
			mirt_host::build_sah_tree(bvh_prims, n_spheres, own, prim_of_slot);
		}
		const std::vector<uint32_t>* slot_map = caller_tree ? nullptr : &prim_of_slot;
		const std::string why = mirt_host::build_records(own.data(), static_cast<uint32_t>(own.size()), bvh_prims, n_spheres, recs, &depth, slot_map);
		if (!why.empty()) return fail(c, MIRT_ERR_ARG, "%s BVH rejected: %s", caller_tree ? "caller's" : "internal", why.c_str());
		n_recs = static_cast<uint32_t>(recs.size() / 16);
		half = c->allow_half && mirt_host::build_half_records(recs, half_recs);
		// 4-wide binary16 records when the tree allows (bvh_layout.hpp build_wide_half_records): half as many dependent fetches per ray
		uint32_t wide_levels = 0;
		wide = half && c->tune_wide && mirt_host::build_wide_half_records(recs, wide_recs, &wide_levels);
		if (wide) n_recs = static_cast<uint32_t>(wide_recs.size() / 16);
		if ((r = wide ? upload(c, c->recs, wide_recs) : half ? upload(c, c->recs, half_recs) : upload(c, c->recs, recs))) return r;
	}
	if ((r = upload(c, c->prim_mat, pm)) ||
	    (r = upload(c, c->light_sphere, lsp)) || (r = upload(c, c->light_emit, lem)) || (r = upload(c, c->mat_albedo, alb)) ||
	    (r = upload(c, c->mat_emission, emi)) || (r = upload(c, c->hdri, sky))) return r;
	HIP_TRY(c, hipStreamSynchronize(c->stream));

	SceneDev& s = c->scene;
	s.recs = wide_gpu ? c->recs_wide.as<float4>() : c->recs.as<float4>(); s.spheres = c->spheres.as<float4>(); s.prim_mat = c->prim_mat.as<int32_t>();
	s.light_sphere = c->light_sphere.as<float4>(); s.light_emit = c->light_emit.as<float4>();
	s.mat_albedo = c->mat_albedo.as<float4>(); s.mat_emission = c->mat_emission.as<float4>();
	s.hdri = c->hdri.as<float4>();
	s.n_spheres = n_spheres; s.n_recs = n_recs; s.n_mat = n_materials; s.n_lights = n_lights;
	c->bvh_depth = depth;
	// LDS staging plan: the whole tree and every sphere packet when they fit the budget (1k spheres: 64 + 16 KB),
	// otherwise the top of the tree only (records are breadth-first) and spheres from L2.
	s.half_boxes = half ? 1u : 0u;
	s.wide = wide ? 1u : 0u;
	s.stack16 = (half && n_recs <= 32768 && n_spheres <= 32768) ? 1u : 0u;     // a stack entry = 15-bit record or prim index + the leaf flag
	const uint32_t rec_bytes = (half && !wide) ? 32u : 64u, budget = stage_budget(half, s.stack16 != 0);
	if (static_cast<uint64_t>(n_recs) * rec_bytes + static_cast<uint64_t>(n_spheres) * 16u <= budget) { s.lds_recs = n_recs; s.lds_spheres = n_spheres; }
	else { s.lds_recs = std::min<uint32_t>(n_recs, budget / rec_bytes); s.lds_spheres = 0; }
	c->trace_lds_bytes = s.lds_recs * rec_bytes + s.lds_spheres * 16u;
	{
		const int lds_max = static_cast<int>(kLdsPerCu);
		const void* dyn_lds_kernels[] = { reinterpret_cast<const void*>(&k_trace<true, kPrimaryNone>), reinterpret_cast<const void*>(&k_trace<false, kPrimaryNone>),
		                                  reinterpret_cast<const void*>(&k_trace<true, kPrimaryAll>), reinterpret_cast<const void*>(&k_trace<false, kPrimaryAll>),
		                                  reinterpret_cast<const void*>(&k_trace<true, kPrimaryList>), reinterpret_cast<const void*>(&k_trace<false, kPrimaryList>),
		                                  reinterpret_cast<const void*>(&k_primary_cand<true>), reinterpret_cast<const void*>(&k_primary_cand<false>) };
		for (const void* k : dyn_lds_kernels) HIP_TRY(c, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
	}
	for (int k = 0; k < 3; k++) s.ambient[k] = ambient_color[k];
	s.hdri_w = static_cast<int32_t>(hdri_w); s.hdri_h = static_cast<int32_t>(hdri_h);
	s.hdri_fw = static_cast<float>(static_cast<int32_t>(hdri_w) - 1);                 // Application.cpp:230-231
	s.hdri_fh = static_cast<float>(static_cast<int32_t>(hdri_h) - 1);
	{ const float a = ambient_color[0], b = ambient_color[1], d = ambient_color[2];
	  const float m1 = (b < d) ? d : b; const float m0 = (a < m1) ? m1 : a; s.has_ambient = (m0 > 0.0f) ? 1u : 0u; }   // Renderer.hpp:79
	c->have_scene = true;
	return MIRT_OK;
}

int mirt_set_camera(mirt_ctx* c, const float pos[3], const float orient_xyzw[4], float half_width, float half_height, float z, float exposure) {
	if (!c) return MIRT_ERR_ARG;
	if (!pos || !orient_xyzw) return fail(c, MIRT_ERR_ARG, "pos / orient is NULL");
	if (c->have_camera && std::memcmp(c->camera.pos, pos, 12) == 0 && std::memcmp(c->camera.orient, orient_xyzw, 16) == 0 &&
	    c->camera.half_width == half_width && c->camera.half_height == half_height && c->camera.z == z && c->camera.exposure == exposure)
		return MIRT_OK;                                                                // unchanged (a host that re-sends it every frame): nothing to flush
	{ const int fr = flush_deferred(c); if (fr) return fr; }                        // deferred accumulations belong to the camera they were issued under
	for (int k = 0; k < 3; k++) c->camera.pos[k] = pos[k];
	for (int k = 0; k < 4; k++) c->camera.orient[k] = orient_xyzw[k];
	c->camera.half_width = half_width; c->camera.half_height = half_height; c->camera.z = z; c->camera.exposure = exposure;
	c->have_camera = true;
	return MIRT_OK;
}

int mirt_set_policy(mirt_ctx* c, const mirt_policy* p) {
	if (!c || !p) return MIRT_ERR_ARG;
	if (p->max_bounces < 1 || p->max_bounces > 1024) return fail(c, MIRT_ERR_ARG, "max_bounces %u out of range", p->max_bounces);
	if (p->buckets < 1 || p->buckets > MIRT_MAX_BUCKETS) return fail(c, MIRT_ERR_ARG, "buckets %u out of range 1..%u", p->buckets, MIRT_MAX_BUCKETS);
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	const bool realloc_acc = p->buckets != c->policy.buckets;
	// batch size / batches in flight are planned again only when something the plan depends on changes (a toggle like trace_primary_rays
	// leaves the ray-stream arena, tens of GB, where it is)
	const bool replan = p->max_batch != c->policy.max_batch || p->streams != c->policy.streams || p->buckets != c->policy.buckets || p->max_bounces != c->policy.max_bounces;
	c->policy = *p;
	if (replan) c->planned_for = 0;
	if (realloc_acc && c->n_tiles) { int r = alloc_accumulator(c); if (r) return r; }
	return MIRT_OK;
}
int mirt_get_policy(const mirt_ctx* c, mirt_policy* p) {
	if (!c || !p) return MIRT_ERR_ARG;
	*p = c->policy;
	plan_batches(const_cast<mirt_ctx*>(c));        // the plan is a cache: made here if no launch has made it yet, so that the values below are the ones launches will use
	p->max_batch = batch_limit(c);                 // the values in effect where the caller left 0 = auto
	p->streams = wanted_slots(c);
	return MIRT_OK;
}

int mirt_resize(mirt_ctx* c, uint32_t width, uint32_t height) {
	if (!c) return MIRT_ERR_ARG;
	if (width > 65536 || height > 65536) return fail(c, MIRT_ERR_ARG, "size %ux%u too large", width, height);
	c->deferred = 0;                                                                   // the accumulator is about to be zeroed (Renderer.hpp:61-62)
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	c->width = width; c->height = height;
	c->h_tiles = width / MIRT_TILE_ROOT; c->v_tiles = height / MIRT_TILE_ROOT;          // Renderer.hpp:59-60
	c->first_tile = 0; c->n_tiles = c->h_tiles * c->v_tiles; c->run_tiles = 0; c->stride_tiles = 0;
	HIP_TRY(c, c->framebuffer.ensure(std::max<size_t>(static_cast<size_t>(width) * height, 1) * sizeof(float4)));
	HIP_TRY(c, hipMemsetAsync(c->framebuffer.ptr, 0, c->framebuffer.bytes, c->stream));
	if (c->frame_host_bytes < c->framebuffer.bytes) {
		if (c->frame_host) (void)hipHostFree(c->frame_host);
		c->frame_host = nullptr; c->frame_host_bytes = 0;
		HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->frame_host), c->framebuffer.bytes, hipHostMallocDefault));
		c->frame_host_bytes = c->framebuffer.bytes;
	}
	return alloc_accumulator(c);                                                      // Renderer.hpp:61-62
}

int mirt_set_tile_range(mirt_ctx* c, uint32_t first_tile, uint32_t n_tiles) {
	if (!c) return MIRT_ERR_ARG;
	const uint64_t all = static_cast<uint64_t>(c->h_tiles) * c->v_tiles;
	if (static_cast<uint64_t>(first_tile) + n_tiles > all) return fail(c, MIRT_ERR_ARG, "tile range [%u,+%u) exceeds %llu tiles", first_tile, n_tiles, (unsigned long long)all);
	c->deferred = 0;                                                                   // zeroes the accumulator
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	c->first_tile = first_tile; c->n_tiles = n_tiles; c->run_tiles = 0; c->stride_tiles = 0;
	return alloc_accumulator(c);
}

int mirt_set_tile_rows(mirt_ctx* c, uint32_t first_row, uint32_t row_stride) {
	if (!c) return MIRT_ERR_ARG;
	if (row_stride == 0) return fail(c, MIRT_ERR_ARG, "row_stride must be at least 1");
	c->deferred = 0;                                                                   // zeroes the accumulator
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	const uint32_t rows = first_row < c->v_tiles ? (c->v_tiles - first_row + row_stride - 1) / row_stride : 0u;
	c->first_tile = first_row * c->h_tiles; c->n_tiles = rows * c->h_tiles;
	c->run_tiles = c->h_tiles; c->stride_tiles = row_stride > 1 ? row_stride * c->h_tiles : 0u;
	return alloc_accumulator(c);
}

int mirt_reset(mirt_ctx* c) {
	if (!c) return MIRT_ERR_ARG;
	c->deferred = 0;                                                                   // what was not launched yet would be wiped anyway
	HIP_TRY(c, hipSetDevice(c->device));
	return alloc_accumulator(c);                                                      // Renderer.hpp:64-67
}

int mirt_accumulate_async(mirt_ctx* c, uint32_t n_calls) {
	int r = check_ready(c); if (r) return r;
	HIP_TRY(c, hipSetDevice(c->device));
	if ((r = ensure_streams(c))) return r;
	// Whole batches are launched now; a remainder waits for more calls (a frame loop that calls this once per frame still gets
	// full-size launches) and is launched by the next call that needs it: mirt_synchronize, a read, a state change.
	if (c->n_tiles == 0) { c->accumulations += n_calls; return MIRT_OK; }
	const uint32_t limit = batch_limit(c);
	uint64_t total = static_cast<uint64_t>(c->deferred) + n_calls;
	c->deferred = 0;
	while (total >= limit) {
		if ((r = launch_batch(c, limit))) return r;
		total -= limit;
	}
	c->deferred = static_cast<uint32_t>(total);
	return MIRT_OK;
}
int mirt_synchronize(mirt_ctx* c) {
	if (!c) return MIRT_ERR_ARG;
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	return MIRT_OK;
}
int mirt_accumulate(mirt_ctx* c, uint32_t n_calls) {
	int r = mirt_accumulate_async(c, n_calls);
	if (r) return r;
	return mirt_synchronize(c);
}
int mirt_get_accumulations(const mirt_ctx* c, uint32_t* a) { if (!c || !a) return MIRT_ERR_ARG; *a = c->accumulations + c->deferred; return MIRT_OK; }

int mirt_accumulator_floats(const mirt_ctx* c, size_t* n) {
	if (!c || !n) return MIRT_ERR_ARG;
	*n = static_cast<size_t>(c->n_tiles) * c->policy.buckets * 3 * kTileSize;
	return MIRT_OK;
}
int mirt_read_accumulator(mirt_ctx* c, float* dst) {
	if (!c || !dst) return MIRT_ERR_ARG;
	size_t n = 0; mirt_accumulator_floats(c, &n);
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	if (n) HIP_TRY(c, hipMemcpy(dst, c->accumulator.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
	return MIRT_OK;
}
int mirt_accumulator_device(mirt_ctx* c, void** ptr, size_t* bytes) {
	if (!c || !ptr || !bytes) return MIRT_ERR_ARG;
	size_t n = 0; mirt_accumulator_floats(c, &n);
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));                                                           // the caller reads the slab on a stream of its own (RCCL gather): everything enqueued here has landed
	*ptr = c->accumulator.ptr; *bytes = n * sizeof(float);
	return MIRT_OK;
}
int mirt_load_accumulator(mirt_ctx* c, const float* src, int src_is_device, uint32_t accumulations) {
	if (!c || !src) return MIRT_ERR_ARG;
	size_t n = 0; mirt_accumulator_floats(c, &n);
	c->deferred = 0;                                                                   // overwritten below
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	if (n && src != c->accumulator.ptr)                                                // src == the slab itself (filled in place by mirt_group_gather): only `accumulations` changes
		HIP_TRY(c, hipMemcpy(c->accumulator.ptr, src, n * sizeof(float), src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
	c->accumulations = accumulations;
	return MIRT_OK;
}

int mirt_render(mirt_ctx* c, float* rgba_host) {
	int r = check_ready(c); if (r) return r;
	if (!rgba_host) return fail(c, MIRT_ERR_ARG, "rgba_host is NULL");
	const uint32_t k = c->policy.buckets;
	const uint32_t issued = c->accumulations + c->deferred;
	if (issued == 0 || (issued % k) != 0) return MIRT_NOT_READY;                                    // Renderer.hpp:437 (nothing is launched for this)
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	HIP_TRY(c, hipSetDevice(c->device));
	const float scale = c->camera.exposure / static_cast<float>(c->accumulations / k);              // Renderer.hpp:439
	const uint32_t n_pix = c->n_tiles * kTileSize;
	HIP_TRY(c, sync_all(c));
	{ Bracket t(c, MIRT_K_RESOLVE);
	  hipLaunchKernelGGL(k_resolve, dim3(grid_for(c, n_pix)), dim3(kBlock), 0, c->stream, c->accumulator.as<float>(), c->framebuffer.as<float4>(),
	                     n_pix, c->first_tile, c->run_tiles ? c->run_tiles : 1u, c->stride_tiles, c->h_tiles, c->width, k, scale); }
	HIP_TRY(c, hipGetLastError());
	const size_t frame_floats = static_cast<size_t>(c->width) * c->height * 4;
	const bool whole = c->n_tiles == c->h_tiles * c->v_tiles && c->width == c->h_tiles * MIRT_TILE_ROOT && c->height == c->v_tiles * MIRT_TILE_ROOT;
	HIP_TRY(c, hipMemcpyAsync(c->frame_host, c->framebuffer.ptr, frame_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if (whole) std::memcpy(rgba_host, c->frame_host, frame_floats * sizeof(float));   // every pixel is this context's
	else for (uint32_t local = 0; local < c->n_tiles; local++) {                                    // only this context's tiles
		const uint32_t t = c->stride_tiles ? c->first_tile + (local / c->run_tiles) * c->stride_tiles + local % c->run_tiles : c->first_tile + local;
		const uint32_t x0 = MIRT_TILE_ROOT * (t % c->h_tiles), y0 = MIRT_TILE_ROOT * (t / c->h_tiles);
		for (uint32_t row = 0; row < MIRT_TILE_ROOT; row++) {
			const size_t off = (static_cast<size_t>(y0 + row) * c->width + x0) * 4;
			std::memcpy(rgba_host + off, c->frame_host + off, MIRT_TILE_ROOT * 4 * sizeof(float));
		}
	}
	return MIRT_OK;
}

int mirt_get_counters(mirt_ctx* c, mirt_counters* out) {
	if (!c || !out) return MIRT_ERR_ARG;
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, sync_all(c));
	DevCounters d;
	HIP_TRY(c, hipMemcpy(&d, c->counters.ptr, sizeof d, hipMemcpyDeviceToHost));
	out->rays = d.rays; out->shadow_rays = d.shadow_rays; out->nodes = d.nodes; out->spheres = d.spheres;
	out->shadow_nodes = d.shadow_nodes; out->shadow_spheres = d.shadow_spheres; out->terminated = d.terminated; out->dropped = d.dropped;
	return MIRT_OK;
}
int mirt_get_kernel_times(mirt_ctx* c, mirt_kernel_times* out, int reset) {
	if (!c || !out) return MIRT_ERR_ARG;
	HIP_TRY(c, hipSetDevice(c->device));
	harvest(c);
	*out = c->times;
	if (reset) c->times = mirt_kernel_times{};
	return MIRT_OK;
}
int mirt_get_stream(mirt_ctx* c, void** s) { if (!c || !s) return MIRT_ERR_ARG; *s = c->stream; return MIRT_OK; }

// ---- stage-level entry points -------------------------------------------------------------------
int mirt_debug_raygen(mirt_ctx* c, uint32_t accumulations, float* p_xyz, float* dir_xyz) {
	int r = check_ready(c); if (r) return r;
	if (!p_xyz || !dir_xyz || accumulations == 0) return fail(c, MIRT_ERR_ARG, "bad arguments");
	if (c->n_tiles == 0) return fail(c, MIRT_ERR_STATE, "no tiles owned");
	HIP_TRY(c, hipSetDevice(c->device));
	if ((r = ensure_streams(c))) return r;
	const FrameParams fp = frame_params(c, accumulations - 1, 1);
	const size_t n = fp.n_pix;
	HIP_TRY(c, sync_all(c));
	hipLaunchKernelGGL(k_raygen, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, fp, c->slots[0].stream_buf[0], c->slots[0].counts.as<uint32_t>());
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	const StreamBuf& s = c->slots[0].stream_buf[0];
	float* srcs[6] = { s.px, s.py, s.pz, s.dx, s.dy, s.dz };
	for (int k = 0; k < 3; k++) {
		HIP_TRY(c, hipMemcpy(p_xyz + k * n, srcs[k], n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(c, hipMemcpy(dir_xyz + k * n, srcs[3 + k], n * 4, hipMemcpyDeviceToHost));
	}
	return MIRT_OK;
}

int mirt_debug_trace_closest(mirt_ctx* c, size_t n, const float* p_xyz, const float* dir_xyz, float* tfar_out, int32_t* prim_out) {
	if (!c) return MIRT_ERR_ARG;
	if (!c->have_scene) return fail(c, MIRT_ERR_STATE, "mirt_set_scene has not been called");
	if (!p_xyz || !dir_xyz || !tfar_out || !prim_out || n == 0 || n >= (1ull << 31)) return fail(c, MIRT_ERR_ARG, "bad arguments");
	HIP_TRY(c, hipSetDevice(c->device));
	ScopedBuffer rays, res, cnt, ctr, fat;
	constexpr size_t kCntWords = 2 * kQueueWords + 8;      // closest queue | shadow queue | work counter x2, fat count x2
	HIP_TRY(c, rays.ensure(n * 6 * 4)); HIP_TRY(c, res.ensure(n * 8)); HIP_TRY(c, cnt.ensure(kCntWords * 4));
	HIP_TRY(c, hipMemset(cnt.ptr, 0, kCntWords * 4));
	float* d = rays.as<float>();
	HIP_TRY(c, hipMemcpy(d, p_xyz, n * 12, hipMemcpyHostToDevice));
	HIP_TRY(c, hipMemcpy(d + 3 * n, dir_xyz, n * 12, hipMemcpyHostToDevice));
	const uint32_t n32 = static_cast<uint32_t>(n);
	uint32_t* cn = cnt.as<uint32_t>();
	HIP_TRY(c, hipMemcpy(cn, &n32, 4, hipMemcpyHostToDevice));                 // all n rays in segment 0 of the closest queue: slot = ray number
	StreamBuf in{};
	in.px = d; in.py = d + n; in.pz = d + 2 * n; in.dx = d + 3 * n; in.dy = d + 4 * n; in.dz = d + 5 * n;
	SceneDev sc = c->scene; sc.use_bvh = c->policy.use_bvh; sc.chunk_max = c->tune_chunk; sc.leaf_batch = c->tune_leaf_batch; sc.refill_idle = c->tune_refill_idle;
	DevCounters* scratch_ctr = nullptr;
	HIP_TRY(c, ctr.ensure(sizeof(DevCounters))); scratch_ctr = ctr.as<DevCounters>();
	HIP_TRY(c, hipMemset(scratch_ctr, 0, sizeof(DevCounters)));
	HIP_TRY(c, fat.ensure(2u * kFatCapacity * sizeof(uint32_t)));
	uint32_t* misc = cn + 2 * kQueueWords;
	const FatList fc{ misc + 2, fat.as<uint32_t>(), kFatCapacity }, fs{ misc + 3, fat.as<uint32_t>() + kFatCapacity, kFatCapacity };
	{ Bracket t(c, MIRT_K_TRACE);                                               // policy.profile: the launch is timed like those of a batch (mirt_get_kernel_times)
	hipLaunchKernelGGL((k_trace<false, kPrimaryNone>), dim3(trace_grid(c, n)), dim3(kTraceBlock), trace_lds(c), c->stream, sc, FrameParams{}, in, res.as<HitRec>(),
	                   Queue{ cn, 0u }, misc, ShadowBuf{}, ShadowSink{}, Queue{ cn + kQueueWords, 0u }, misc + 1, fc, fs, scratch_ctr);
	if (sc.use_bvh) hipLaunchKernelGGL((k_trace_fat<false, kPrimaryNone>), dim3(64), dim3(1024), 0, c->stream, sc, FrameParams{}, in, res.as<HitRec>(), fc, ShadowBuf{}, ShadowSink{}, fs, scratch_ctr, static_cast<const uint32_t*>(nullptr)); }
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	std::vector<HitRec> host_hits(n);
	if (e == hipSuccess) e = hipMemcpy(host_hits.data(), res.ptr, n * sizeof(HitRec), hipMemcpyDeviceToHost);
	if (e == hipSuccess) for (size_t i = 0; i < n; i++) { tfar_out[i] = host_hits[i].tfar; prim_out[i] = host_hits[i].prim; }
	if (e != hipSuccess) return fail(c, MIRT_ERR_HIP, "debug_trace_closest: %s", hipGetErrorString(e));
	return MIRT_OK;
}

int mirt_debug_trace_shadow(mirt_ctx* c, size_t n, const float* p_xyz, const float* dir_xyz, const float* tfar, uint8_t* occluded_out) {
	if (!c) return MIRT_ERR_ARG;
	if (!c->have_scene) return fail(c, MIRT_ERR_STATE, "mirt_set_scene has not been called");
	if (!p_xyz || !dir_xyz || !tfar || !occluded_out || n == 0 || n >= (1ull << 31)) return fail(c, MIRT_ERR_ARG, "bad arguments");
	HIP_TRY(c, hipSetDevice(c->device));
	ScopedBuffer rays, occ, cnt, ctr, fat;
	constexpr size_t kCntWords = 2 * kQueueWords + 8;      // closest queue (empty) | shadow queue | work counter x2, fat count x2
	HIP_TRY(c, rays.ensure(n * 7 * 4)); HIP_TRY(c, occ.ensure(n * 4)); HIP_TRY(c, cnt.ensure(kCntWords * 4)); HIP_TRY(c, ctr.ensure(sizeof(DevCounters)));
	HIP_TRY(c, fat.ensure(2u * kFatCapacity * sizeof(uint32_t)));
	float* d = rays.as<float>();
	HIP_TRY(c, hipMemcpy(d, p_xyz, n * 12, hipMemcpyHostToDevice));
	HIP_TRY(c, hipMemcpy(d + 3 * n, dir_xyz, n * 12, hipMemcpyHostToDevice));
	HIP_TRY(c, hipMemcpy(d + 6 * n, tfar, n * 4, hipMemcpyHostToDevice));
	HIP_TRY(c, hipMemset(cnt.ptr, 0, kCntWords * 4));
	uint32_t* cn = cnt.as<uint32_t>();
	const uint32_t n32 = static_cast<uint32_t>(n);
	HIP_TRY(c, hipMemcpy(cn + kQueueWords, &n32, 4, hipMemcpyHostToDevice));   // all n rays in segment 0 of the shadow queue
	HIP_TRY(c, hipMemset(ctr.ptr, 0, sizeof(DevCounters)));
	SceneDev sc = c->scene; sc.use_bvh = c->policy.use_bvh; sc.chunk_max = c->tune_chunk; sc.leaf_batch = c->tune_leaf_batch; sc.refill_idle = c->tune_refill_idle;
	ShadowBuf sh{}; sh.px = d; sh.py = d + n; sh.pz = d + 2 * n; sh.dx = d + 3 * n; sh.dy = d + 4 * n; sh.dz = d + 5 * n; sh.tfar = d + 6 * n;
	uint32_t* misc = cn + 2 * kQueueWords;
	const FatList fc{ misc + 2, fat.as<uint32_t>(), kFatCapacity }, fs{ misc + 3, fat.as<uint32_t>() + kFatCapacity, kFatCapacity };
	// the product's own kernels: the shadow queue of k_trace, then the fat-ray pass; the sink only records the occlusion flags
	ShadowSink sink{}; sink.occ = occ.as<uint32_t>();
	{ Bracket t(c, MIRT_K_TRACE);
	hipLaunchKernelGGL((k_trace<false, kPrimaryNone>), dim3(trace_grid(c, n)), dim3(kTraceBlock), trace_lds(c), c->stream, sc, FrameParams{}, StreamBuf{}, static_cast<HitRec*>(nullptr),
	                   Queue{ cn, 0u }, misc, sh, sink, Queue{ cn + kQueueWords, 0u }, misc + 1, fc, fs, ctr.as<DevCounters>());
	if (sc.use_bvh) hipLaunchKernelGGL((k_trace_fat<false, kPrimaryNone>), dim3(64), dim3(1024), 0, c->stream, sc, FrameParams{}, StreamBuf{}, static_cast<HitRec*>(nullptr), fc, sh, sink, fs, ctr.as<DevCounters>(), static_cast<const uint32_t*>(nullptr)); }
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	std::vector<uint32_t> host_occ(n);
	if (e == hipSuccess) e = hipMemcpy(host_occ.data(), occ.ptr, n * 4, hipMemcpyDeviceToHost);
	if (e == hipSuccess) for (size_t i = 0; i < n; i++) occluded_out[i] = host_occ[i] ? 1 : 0;
	if (e != hipSuccess) return fail(c, MIRT_ERR_HIP, "debug_trace_shadow: %s", hipGetErrorString(e));
	return MIRT_OK;
}

int mirt_debug_math(mirt_ctx* c, int fn, size_t n, const float* in, float* out) {
	if (!c) return MIRT_ERR_ARG;
	static const int n_in[11] = { 1, 2, 1, 2, 2, 6, 8, 3, 10, 9, 1 }, n_out[11] = { 2, 1, 1, 3, 3, 10, 5, 5, 3, 6, 1 };
	if (fn < 0 || fn > 10 || !in || !out || n == 0 || n >= (1u << 28)) return fail(c, MIRT_ERR_ARG, "bad arguments");
	HIP_TRY(c, hipSetDevice(c->device));
	ScopedBuffer din, dout;
	HIP_TRY(c, din.ensure(n * n_in[fn] * 4)); HIP_TRY(c, dout.ensure(n * n_out[fn] * 4));
	HIP_TRY(c, hipMemcpy(din.ptr, in, n * n_in[fn] * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_debug_math, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, fn, static_cast<uint32_t>(n), din.as<float>(), dout.as<float>());
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e == hipSuccess) e = hipMemcpy(out, dout.ptr, n * n_out[fn] * 4, hipMemcpyDeviceToHost);
	if (e != hipSuccess) return fail(c, MIRT_ERR_HIP, "debug_math: %s", hipGetErrorString(e));
	return MIRT_OK;
}

int mirt_debug_primary_lists(mirt_ctx* c, uint32_t hist[10]) {
	int r = check_ready(c); if (r) return r;
	if (!hist) return fail(c, MIRT_ERR_ARG, "hist is NULL");
	if (!c->policy.use_bvh || c->scene.n_recs == 0 || c->n_tiles == 0) return fail(c, MIRT_ERR_STATE, "needs policy.use_bvh, a tree and at least one tile");
	{ const int fr = flush_deferred(c); if (fr) return fr; }
	HIP_TRY(c, hipSetDevice(c->device));
	if ((r = ensure_streams(c))) return r;
	HIP_TRY(c, sync_all(c));
	PipeSlot& sl = c->slots[0];
	const FrameParams fp = frame_params(c, 0, 1);
	SceneDev sc = c->scene; sc.use_bvh = 1; sc.chunk_max = c->tune_chunk; sc.leaf_batch = c->tune_leaf_batch; sc.refill_idle = c->tune_refill_idle;
	uint32_t* misc = sl.counts.as<uint32_t>();
	HIP_TRY(c, hipMemsetAsync(misc, 0, 64, c->stream));
	const float rho = bundle_half_angle(c);
	const FatList none{ misc + 1, sl.fat.as<uint32_t>(), 0u };
	hipLaunchKernelGGL(k_primary_cand<false>, dim3(trace_grid(c, fp.n_pix)), dim3(kTraceBlock), trace_lds(c), c->stream, sc, fp, sl.cand.as<uint32_t>(), rho, misc, none, c->counters.as<DevCounters>(), static_cast<uint32_t*>(nullptr), misc + 2);
	HIP_TRY(c, hipGetLastError());
	std::vector<uint32_t> host(static_cast<size_t>(fp.n_pix));                            // plane 0 of the lists: the counts
	HIP_TRY(c, hipMemcpyAsync(host.data(), sl.cand.ptr, host.size() * 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	for (int k = 0; k < 10; k++) hist[k] = 0;
	for (size_t p = 0; p < fp.n_pix; p++) { const uint32_t n = host[p]; hist[n == kCandOverflow ? 9 : std::min<uint32_t>(n, 8u)]++; }     // hist[8]: 8 or more
	return MIRT_OK;
}

int mirt_debug_info(mirt_ctx* c, uint32_t out[8]) {
	if (!c || !out) return MIRT_ERR_ARG;
	const SceneDev& s = c->scene;
	uint32_t per_cu = kLdsPerCu / std::max<uint32_t>(trace_lds(c), 1u);
	per_cu = std::min<uint32_t>(std::max<uint32_t>(per_cu, 1u), c->tune_trace_wgs);
	out[0] = s.n_recs; out[1] = s.lds_recs; out[2] = s.lds_spheres; out[3] = c->bvh_depth; out[4] = s.half_boxes | (s.wide << 1);
	out[5] = trace_lds(c); out[6] = per_cu; out[7] = static_cast<uint32_t>(c->n_cu);
	return MIRT_OK;
}
int mirt_debug_allow_half_boxes(mirt_ctx* c, int allow) { if (!c) return MIRT_ERR_ARG; c->allow_half = allow != 0; return MIRT_OK; }

} // extern "C"
