// mirt_group.hip — multi-GPU inside the library (include/mirt.h, "mirt_group_*").
//
// The reference host is ONE C++ object: `Renderer<> renderer{scene}` (Application.cpp:514) whose Accumulate() is a
// parallel_for over the 16x16 tiles (Renderer.hpp:75).  A group stands for that object on n GPUs of one node:
//   * one mirt_ctx per device, each with the whole scene (a few MB) and its share of the tile ROWS, interleaved: member i renders
//     tile rows i, i+n, i+2n, ... (every GPU sees sky and ground alike; contiguous stripes differ by 1.5x in cost);
//   * no exchange while rendering: tiles are independent and every random draw is keyed on the global LaunchIndex
//     (Renderer.hpp:107,117), so the union of the members' accumulators is the single-GPU accumulator bit for bit;
//   * ONE gather of the accumulator slabs to the first device — RCCL point-to-point (a grouped ncclSend / ncclRecv per peer,
//     i.e. ncclGather) over xGMI, one link per peer — followed by a device-side un-interleave into the full-image
//     AccumulationTile layout, where Render() resolves the whole frame.
// librccl is loaded on first use (dlopen), so single-GPU users of libmirt.so do not pay for it.  Members that name the SAME
// device (a rehearsal on a one-GPU box; RCCL refuses two ranks on one device) exchange their slabs with plain device copies.
// Everything else here goes through the public C-ABI of the contexts.
#include "../../include/mirt.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rccl {
	void* lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	std::string load() {
		if (lib) return "";
		for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
		if (!lib) return std::string("cannot load librccl: ") + dlerror();
		auto sym = [&](const char* n) { return dlsym(lib, n); };
		CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
		CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
		GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
		GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
		Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
		Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
		GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
		if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) { dlclose(lib); lib = nullptr; return "librccl lacks a required symbol"; }
		return "";
	}
};
Rccl g_rccl;
thread_local std::string g_group_create_error;

// Member r's slab holds its tile rows r, r+n, ... in ascending order: local row j is tile row r + j*n of the image.
// One thread per float4 of the slab; a tile row is h_tiles * buckets * 3 * 256 floats in both layouts.
__global__ __launch_bounds__(256) void k_uninterleave(float4* __restrict__ full, const float4* __restrict__ slab, size_t quads_per_row, uint32_t rows, uint32_t member, uint32_t n_members) {
	const size_t total = quads_per_row * rows;
	for (size_t q = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; q < total; q += static_cast<size_t>(gridDim.x) * 256) {
		const size_t j = q / quads_per_row, within = q - j * quads_per_row;
		full[(member + j * n_members) * quads_per_row + within] = slab[q];
	}
}

} // namespace

struct mirt_group {
	std::vector<int> devices;
	std::vector<mirt_ctx*> members;
	mirt_ctx* full = nullptr;              // on devices[0]: the gathered full-image accumulator and its Render()
	std::vector<ncclComm_t> comms;         // one per member, distinct devices only
	bool distinct = true;                  // every member on a device of its own (RCCL); otherwise plain device copies
	std::vector<void*> staging;            // on devices[0]: where member r's slab lands (r >= 1)
	std::vector<size_t> staging_bytes;
	uint32_t width = 0, height = 0, buckets = 5;
	bool gathered = false;                 // `full` holds the members' current accumulators
	double last_gather_ms = 0.0;
	std::string error;
};

namespace {

int gfail(mirt_group* g, int code, const char* fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	if (g) g->error = buf; else g_group_create_error = buf;
	return code;
}
// Runs `call` on every member; the first failure is reported with the member's own error text.
#define FOR_MEMBERS(g, what, call) do { for (size_t _i = 0; _i < (g)->members.size(); _i++) { mirt_ctx* ctx = (g)->members[_i]; const int _rc = (call); \
	if (_rc < 0) return gfail((g), _rc, "%s on member %zu (device %d): %s", (what), _i, (g)->devices[_i], mirt_last_error(ctx)); } } while (0)
#define FULL_TRY(g, what, call) do { const int _rc = (call); if (_rc < 0) return gfail((g), _rc, "%s on the gather context: %s", (what), mirt_last_error((g)->full)); } while (0)
#define GHIP(g, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return gfail((g), MIRT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); } while (0)
#define GNCCL(g, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return gfail((g), MIRT_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(_r)); } while (0)

} // namespace

extern "C" {

const char* mirt_group_last_error(const mirt_group* g) { return g ? g->error.c_str() : g_group_create_error.c_str(); }

int mirt_group_create(const int* devices, int n, mirt_group** out) {
	if (!out) return gfail(nullptr, MIRT_ERR_ARG, "out is NULL");
	*out = nullptr;
	if (!devices || n < 1 || n > 64) return gfail(nullptr, MIRT_ERR_ARG, "need 1..64 devices");
	mirt_group* g = new mirt_group();
	g->devices.assign(devices, devices + n);
	for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (devices[i] == devices[j]) g->distinct = false;
	auto bail = [&](int rc, const std::string& why) { g_group_create_error = why; for (mirt_ctx* c : g->members) mirt_destroy(c); if (g->full) mirt_destroy(g->full); delete g; return rc; };
	for (int i = 0; i < n; i++) {
		mirt_ctx* c = nullptr;
		const int rc = mirt_create(devices[i], &c);
		if (rc != MIRT_OK) return bail(rc, std::string("mirt_create(device ") + std::to_string(devices[i]) + "): " + mirt_last_error(nullptr));
		g->members.push_back(c);
	}
	if (n > 1) {
		const int rc = mirt_create(devices[0], &g->full);
		if (rc != MIRT_OK) return bail(rc, std::string("mirt_create(gather context): ") + mirt_last_error(nullptr));
		g->staging.assign(n, nullptr); g->staging_bytes.assign(n, 0);
		if (g->distinct) {
			const std::string why = g_rccl.load();
			if (!why.empty()) return bail(MIRT_ERR_HIP, why);
			g->comms.assign(n, nullptr);
			const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n, devices);      // rccl.h:236 — one communicator per device, this process drives them all
			if (r != ncclSuccess) { g->comms.clear(); return bail(MIRT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); }
		}
	}
	*out = g;
	return MIRT_OK;
}

int mirt_group_destroy(mirt_group* g) {
	if (!g) return MIRT_ERR_ARG;
	for (ncclComm_t c : g->comms) if (c) (void)g_rccl.CommDestroy(c);
	if (!g->staging.empty()) { (void)hipSetDevice(g->devices[0]); for (void* p : g->staging) if (p) (void)hipFree(p); }
	for (mirt_ctx* c : g->members) mirt_destroy(c);
	if (g->full) mirt_destroy(g->full);
	delete g;
	return MIRT_OK;
}

int mirt_group_size(const mirt_group* g, int* n) { if (!g || !n) return MIRT_ERR_ARG; *n = static_cast<int>(g->members.size()); return MIRT_OK; }
int mirt_group_member(mirt_group* g, int index, mirt_ctx** ctx) {
	if (!g || !ctx || index < 0 || index >= static_cast<int>(g->members.size())) return MIRT_ERR_ARG;
	*ctx = g->members[index];
	return MIRT_OK;
}

int mirt_group_set_scene(mirt_group* g, const mirt_sphere* geometry, const mirt_sphere* bvh_prims, uint32_t n_spheres, const mirt_bvh_node* nodes, uint32_t n_nodes,
                         const mirt_material* materials, uint32_t n_materials, const int32_t* lights, uint32_t n_lights,
                         const float ambient_color[3], const float* hdri_rgba, uint32_t hdri_w, uint32_t hdri_h) {
	if (!g) return MIRT_ERR_ARG;
	FOR_MEMBERS(g, "mirt_set_scene", mirt_set_scene(ctx, geometry, bvh_prims, n_spheres, nodes, n_nodes, materials, n_materials, lights, n_lights, ambient_color, hdri_rgba, hdri_w, hdri_h));
	// the gather context only resolves frames, but mirt_render wants a scene: the sky and materials are tiny, the geometry is not needed
	if (g->full) FULL_TRY(g, "mirt_set_scene", mirt_set_scene(g->full, nullptr, nullptr, 0, nullptr, 0, materials, n_materials, nullptr, 0, ambient_color, hdri_rgba, hdri_w, hdri_h));
	return MIRT_OK;
}
int mirt_group_set_camera(mirt_group* g, const float pos[3], const float orient_xyzw[4], float half_width, float half_height, float z, float exposure) {
	if (!g) return MIRT_ERR_ARG;
	FOR_MEMBERS(g, "mirt_set_camera", mirt_set_camera(ctx, pos, orient_xyzw, half_width, half_height, z, exposure));
	if (g->full) FULL_TRY(g, "mirt_set_camera", mirt_set_camera(g->full, pos, orient_xyzw, half_width, half_height, z, exposure));
	return MIRT_OK;
}
int mirt_group_set_policy(mirt_group* g, const mirt_policy* p) {
	if (!g || !p) return MIRT_ERR_ARG;
	FOR_MEMBERS(g, "mirt_set_policy", mirt_set_policy(ctx, p));
	if (g->full) FULL_TRY(g, "mirt_set_policy", mirt_set_policy(g->full, p));
	g->buckets = p->buckets;
	g->gathered = false;
	return MIRT_OK;
}
int mirt_group_resize(mirt_group* g, uint32_t width, uint32_t height) {
	if (!g) return MIRT_ERR_ARG;
	const uint32_t n = static_cast<uint32_t>(g->members.size());
	FOR_MEMBERS(g, "mirt_resize", mirt_resize(ctx, width, height));
	if (n > 1) {
		for (uint32_t i = 0; i < n; i++) {
			const int rc = mirt_set_tile_rows(g->members[i], i, n);                   // Renderer.hpp:75: the parallel_for range, split by tile row
			if (rc < 0) return gfail(g, rc, "mirt_set_tile_rows on member %u: %s", i, mirt_last_error(g->members[i]));
		}
		FULL_TRY(g, "mirt_resize", mirt_resize(g->full, width, height));
	}
	g->width = width; g->height = height; g->gathered = false;
	return MIRT_OK;
}
int mirt_group_reset(mirt_group* g) {
	if (!g) return MIRT_ERR_ARG;
	FOR_MEMBERS(g, "mirt_reset", mirt_reset(ctx));
	g->gathered = false;
	return MIRT_OK;
}
int mirt_group_accumulate_async(mirt_group* g, uint32_t n_calls) {
	if (!g) return MIRT_ERR_ARG;
	FOR_MEMBERS(g, "mirt_accumulate_async", mirt_accumulate_async(ctx, n_calls));     // every device gets its launches before anyone is waited for
	g->gathered = false;
	return MIRT_OK;
}
int mirt_group_synchronize(mirt_group* g) {
	if (!g) return MIRT_ERR_ARG;
	FOR_MEMBERS(g, "mirt_synchronize", mirt_synchronize(ctx));
	return MIRT_OK;
}
int mirt_group_accumulate(mirt_group* g, uint32_t n_calls) {
	const int rc = mirt_group_accumulate_async(g, n_calls);
	return rc ? rc : mirt_group_synchronize(g);
}
int mirt_group_get_accumulations(const mirt_group* g, uint32_t* a) { if (!g || g->members.empty()) return MIRT_ERR_ARG; return mirt_get_accumulations(g->members[0], a); }

int mirt_group_get_counters(mirt_group* g, mirt_counters* out) {
	if (!g || !out) return MIRT_ERR_ARG;
	std::memset(out, 0, sizeof *out);
	for (size_t i = 0; i < g->members.size(); i++) {
		mirt_counters c{};
		const int rc = mirt_get_counters(g->members[i], &c);
		if (rc < 0) return gfail(g, rc, "mirt_get_counters on member %zu: %s", i, mirt_last_error(g->members[i]));
		out->rays += c.rays; out->shadow_rays += c.shadow_rays; out->nodes += c.nodes; out->spheres += c.spheres;
		out->shadow_nodes += c.shadow_nodes; out->shadow_spheres += c.shadow_spheres; out->terminated += c.terminated; out->dropped += c.dropped;
	}
	return MIRT_OK;
}

// The one exchange of the path: every member's slab to the first device, then the un-interleave into the full-image layout.
int mirt_group_gather(mirt_group* g) {
	if (!g) return MIRT_ERR_ARG;
	const uint32_t n = static_cast<uint32_t>(g->members.size());
	if (n == 1 || g->gathered) return MIRT_OK;
	if (g->width == 0) return gfail(g, MIRT_ERR_STATE, "mirt_group_resize has not been called");
	const uint32_t h_tiles = g->width / MIRT_TILE_ROOT, v_tiles = g->height / MIRT_TILE_ROOT;
	std::vector<void*> slab(n, nullptr); std::vector<size_t> bytes(n, 0); std::vector<void*> stream(n, nullptr);
	for (uint32_t i = 0; i < n; i++) {
		int rc = mirt_accumulator_device(g->members[i], &slab[i], &bytes[i]);               // launches anything deferred and waits for the member's GPU
		if (rc >= 0) rc = mirt_get_stream(g->members[i], &stream[i]);
		if (rc < 0) return gfail(g, rc, "accumulator of member %u: %s", i, mirt_last_error(g->members[i]));
	}
	void* full_ptr = nullptr; size_t full_bytes = 0; void* full_stream = nullptr;
	FULL_TRY(g, "mirt_accumulator_device", mirt_accumulator_device(g->full, &full_ptr, &full_bytes));
	FULL_TRY(g, "mirt_get_stream", mirt_get_stream(g->full, &full_stream));
	const size_t row_bytes = static_cast<size_t>(h_tiles) * g->buckets * 3 * MIRT_TILE_SIZE * sizeof(float);
	if (full_bytes != row_bytes * v_tiles) return gfail(g, MIRT_ERR_STATE, "gather context holds %zu bytes, the image needs %zu", full_bytes, row_bytes * v_tiles);
	GHIP(g, hipSetDevice(g->devices[0]));
	for (uint32_t i = 1; i < n; i++) if (g->staging_bytes[i] < bytes[i]) {
		if (g->staging[i]) (void)hipFree(g->staging[i]);
		g->staging[i] = nullptr; g->staging_bytes[i] = 0;
		if (bytes[i]) { GHIP(g, hipMalloc(&g->staging[i], bytes[i])); g->staging_bytes[i] = bytes[i]; }
	}
	// From here on every failure is recorded in `rc` and the sequence runs to its end: an ncclGroupStart is always matched by its
	// ncclGroupEnd and both events are destroyed whatever happened in between.
	int rc = MIRT_OK;
	auto hstep = [&](hipError_t e, const char* what) { if (rc == MIRT_OK && e != hipSuccess) rc = gfail(g, MIRT_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); return rc == MIRT_OK; };
	auto nstep = [&](ncclResult_t r, const char* what) { if (rc == MIRT_OK && r != ncclSuccess) rc = gfail(g, MIRT_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString(r)); return rc == MIRT_OK; };
	hipEvent_t t0 = nullptr, t1 = nullptr;
	hipStream_t root = static_cast<hipStream_t>(full_stream);
	hstep(hipEventCreate(&t0), "hipEventCreate"); hstep(hipEventCreate(&t1), "hipEventCreate");
	if (rc == MIRT_OK) hstep(hipEventRecord(t0, root), "hipEventRecord");
	if (rc == MIRT_OK && g->distinct) {
		// ncclGather spelled as its point-to-point form (rccl.h:700,722): the root posts one receive per peer, every peer one send;
		// each transfer rides the xGMI link between that peer and the root
		if (nstep(g_rccl.GroupStart(), "ncclGroupStart")) {
			for (uint32_t i = 1; i < n && rc == MIRT_OK; i++) {
				if (!bytes[i]) continue;
				if (hstep(hipSetDevice(g->devices[0]), "hipSetDevice"))
					nstep(g_rccl.Recv(g->staging[i], bytes[i] / sizeof(float), ncclFloat, static_cast<int>(i), g->comms[0], root), "ncclRecv");
				if (rc == MIRT_OK && hstep(hipSetDevice(g->devices[i]), "hipSetDevice"))
					nstep(g_rccl.Send(slab[i], bytes[i] / sizeof(float), ncclFloat, 0, g->comms[i], static_cast<hipStream_t>(stream[i])), "ncclSend");
			}
			const ncclResult_t end = g_rccl.GroupEnd();                              // always: an open group would swallow every later RCCL call of this thread
			nstep(end, "ncclGroupEnd");
		}
		for (uint32_t i = 1; i < n && rc == MIRT_OK; i++) if (hstep(hipSetDevice(g->devices[i]), "hipSetDevice")) hstep(hipStreamSynchronize(static_cast<hipStream_t>(stream[i])), "hipStreamSynchronize");
		(void)hipSetDevice(g->devices[0]);
	} else if (rc == MIRT_OK) {
		for (uint32_t i = 1; i < n && rc == MIRT_OK; i++) if (bytes[i]) hstep(hipMemcpyAsync(g->staging[i], slab[i], bytes[i], hipMemcpyDeviceToDevice, root), "hipMemcpyAsync");
	}
	const size_t quads_per_row = row_bytes / sizeof(float4);
	for (uint32_t i = 0; i < n && rc == MIRT_OK; i++) {
		const uint32_t rows = static_cast<uint32_t>(bytes[i] / row_bytes);
		if (!rows) continue;
		const float4* src = static_cast<const float4*>(i == 0 ? slab[0] : g->staging[i]);
		const size_t total = quads_per_row * rows;
		const uint32_t grid = static_cast<uint32_t>(std::min<size_t>((total + 255) / 256, 256u * 16u));
		hipLaunchKernelGGL(k_uninterleave, dim3(grid), dim3(256), 0, root, static_cast<float4*>(full_ptr), src, quads_per_row, rows, i, n);
	}
	if (rc == MIRT_OK) hstep(hipGetLastError(), "k_uninterleave");
	if (rc == MIRT_OK) hstep(hipEventRecord(t1, root), "hipEventRecord");
	if (rc == MIRT_OK) hstep(hipStreamSynchronize(root), "hipStreamSynchronize");
	if (rc == MIRT_OK) { float ms = 0.0f; (void)hipEventElapsedTime(&ms, t0, t1); g->last_gather_ms = ms; }
	if (t0) (void)hipEventDestroy(t0);
	if (t1) (void)hipEventDestroy(t1);
	if (rc != MIRT_OK) return rc;
	uint32_t acc = 0;
	(void)mirt_get_accumulations(g->members[0], &acc);
	FULL_TRY(g, "mirt_load_accumulator", mirt_load_accumulator(g->full, static_cast<const float*>(full_ptr), 1, acc));   // in place: only `accumulations` changes
	g->gathered = true;
	return MIRT_OK;
}
// Diagnostic: is RCCL usable from this process?  Loads librccl, makes a one-device communicator on `device` and sends n_floats to
// itself through a grouped ncclSend / ncclRecv pair (the calls mirt_group_gather makes between distinct devices); 0 = the data arrived intact.
int mirt_group_rccl_selftest(int device, size_t n_floats) {
	const std::string why = g_rccl.load();
	if (!why.empty()) return gfail(nullptr, MIRT_ERR_HIP, "%s", why.c_str());
	if (n_floats == 0) return gfail(nullptr, MIRT_ERR_ARG, "n_floats is 0");
	GHIP(nullptr, hipSetDevice(device));
	ncclComm_t comm = nullptr;
	GNCCL(nullptr, g_rccl.CommInitAll(&comm, 1, &device));
	float *src = nullptr, *dst = nullptr;
	hipStream_t st = nullptr;
	std::vector<float> host(n_floats), back(n_floats, 0.0f);
	for (size_t i = 0; i < n_floats; i++) host[i] = static_cast<float>(i % 8191) * 0.5f;
	int rc = MIRT_OK;
	auto step = [&](hipError_t e, const char* what) { if (rc == MIRT_OK && e != hipSuccess) rc = gfail(nullptr, MIRT_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); };
	auto nstep = [&](ncclResult_t r, const char* what) { if (rc == MIRT_OK && r != ncclSuccess) rc = gfail(nullptr, MIRT_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString(r)); };
	step(hipMalloc(&src, n_floats * 4), "hipMalloc"); step(hipMalloc(&dst, n_floats * 4), "hipMalloc"); step(hipStreamCreate(&st), "hipStreamCreate");
	if (rc == MIRT_OK) step(hipMemcpy(src, host.data(), n_floats * 4, hipMemcpyHostToDevice), "hipMemcpy");
	if (rc == MIRT_OK) {
		nstep(g_rccl.GroupStart(), "ncclGroupStart");
		nstep(g_rccl.Recv(dst, n_floats, ncclFloat, 0, comm, st), "ncclRecv");
		nstep(g_rccl.Send(src, n_floats, ncclFloat, 0, comm, st), "ncclSend");
		nstep(g_rccl.GroupEnd(), "ncclGroupEnd");
		step(hipStreamSynchronize(st), "hipStreamSynchronize");
	}
	if (rc == MIRT_OK) step(hipMemcpy(back.data(), dst, n_floats * 4, hipMemcpyDeviceToHost), "hipMemcpy");
	if (rc == MIRT_OK && std::memcmp(host.data(), back.data(), n_floats * 4) != 0) rc = gfail(nullptr, MIRT_ERR_HIP, "data sent through RCCL came back changed");
	if (st) (void)hipStreamDestroy(st);
	if (src) (void)hipFree(src);
	if (dst) (void)hipFree(dst);
	(void)g_rccl.CommDestroy(comm);
	return rc;
}
int mirt_group_last_gather_ms(const mirt_group* g, double* ms) { if (!g || !ms) return MIRT_ERR_ARG; *ms = g->last_gather_ms; return MIRT_OK; }

int mirt_group_accumulator_floats(const mirt_group* g, size_t* n) {
	if (!g || !n) return MIRT_ERR_ARG;
	*n = static_cast<size_t>(g->width / MIRT_TILE_ROOT) * (g->height / MIRT_TILE_ROOT) * g->buckets * 3 * MIRT_TILE_SIZE;
	return MIRT_OK;
}
int mirt_group_read_accumulator(mirt_group* g, float* host_dst) {
	if (!g || !host_dst) return MIRT_ERR_ARG;
	if (g->members.size() == 1) { const int rc = mirt_read_accumulator(g->members[0], host_dst); return rc < 0 ? gfail(g, rc, "%s", mirt_last_error(g->members[0])) : rc; }
	const int rc = mirt_group_gather(g);
	if (rc) return rc;
	FULL_TRY(g, "mirt_read_accumulator", mirt_read_accumulator(g->full, host_dst));
	return MIRT_OK;
}
int mirt_group_render(mirt_group* g, float* rgba_host) {
	if (!g || !rgba_host) return MIRT_ERR_ARG;
	if (g->members.size() == 1) { const int rc = mirt_render(g->members[0], rgba_host); return rc < 0 ? gfail(g, rc, "%s", mirt_last_error(g->members[0])) : rc; }
	uint32_t acc = 0;
	(void)mirt_get_accumulations(g->members[0], &acc);
	if (acc == 0 || acc % g->buckets != 0) return MIRT_NOT_READY;                          // Renderer.hpp:437: nothing is gathered for a frame that is not due
	const int rc = mirt_group_gather(g);
	if (rc) return rc;
	const int rr = mirt_render(g->full, rgba_host);
	return rr < 0 ? gfail(g, rr, "mirt_render on the gather context: %s", mirt_last_error(g->full)) : rr;
}

} // extern "C"
