/* mirt.h — C-ABI of the MI355X-native replacement for the reference's
 * Renderer<Policy>::{Resize, ResetAccumulator, Accumulate, Render} hot path
 * (Borx25/CPU-Raytracing-experiments, Renderer.hpp:53-67,73-434,436-478).
 *
 * The reference has no FFI/plugin layer: the boundary is the C++ class `Renderer`
 * holding `const Scene&` (Renderer.hpp:38,51; instantiated Application.cpp:514).
 * Each entry point below names the reference member it stands in for.  Structs are
 * passed in the reference's exact byte layout so a host can hand over its
 * std::vector<Sphere>/<Material>/<Node> storage unchanged.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every call returns an
 * int status (0 = MIRT_OK, >0 informational, <0 error) and never aborts
 * (the reference returns void and asserts/terminates: App.cpp:43-48, Application.cpp:226-229);
 * one context = one caller thread at a time; calls are synchronous unless named *_async;
 * host buffers are copied, never retained.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_OK              0
#define MIRT_NOT_READY       1   /* mirt_render before accumulations % buckets == 0 (Renderer.hpp:437) */
#define MIRT_ERR_ARG        -1
#define MIRT_ERR_HIP        -2
#define MIRT_ERR_STATE      -3
#define MIRT_ERR_NO_DEVICE  -4

#define MIRT_TILE_ROOT     16u   /* Renderer.hpp:32  (Policy.log_tile = 4) */
#define MIRT_TILE_SIZE    256u   /* Renderer.hpp:33 */
#define MIRT_MAX_MATERIALS 63u   /* Renderer.hpp:23,92 (max_materialID = 64, key -1 = miss) */
#define MIRT_MAX_BUCKETS   16u
#define MIRT_BVH_STACK     64u   /* DataStructures.hpp:26-44, BVH.hpp:127,321 */

/* Primitives.hpp:7-17 — alignas(16) {vec3 position; float radius_sq; int32 material_ID} = 32 B */
typedef struct mirt_sphere {
	float   position[3];
	float   radius_sq;
	int32_t material_ID;
	int32_t _pad[3];
} mirt_sphere;

/* Primitives.hpp:18-27 — alignas(32), 96 B; the path reads albedo and emission only */
typedef struct mirt_material {
	float albedo[3];
	float F0[3];
	float F80[3];
	float emission[3];
	float transmission[3];
	float roughness;
	float IOR_minus_one;
	float _pad[7];
} mirt_material;

/* BVH.hpp:18-31 — alignas(32) {alignas(16) vec3 min; u32 first_id; alignas(16) vec3 max; u32 prim_count} = 32 B.
 * Leaf iff prim_count != 0 (prims [first_id, first_id+prim_count) of the BVH-order array);
 * inner: children at first_id and first_id+1. */
typedef struct mirt_bvh_node {
	float    min_bound[3];
	uint32_t first_id;
	float    max_bound[3];
	uint32_t prim_count;
} mirt_bvh_node;

/* RendererPolicy (Renderer.hpp:19-26) + the compile-time switches of the path, made runtime. */
typedef struct mirt_policy {
	uint32_t max_bounces;   /* Policy.max_bounces, Renderer.hpp:24 (bounce-loop iterations; seed stride 2*max_bounces+1, :107) */
	uint32_t buckets;       /* AccumulationBuckets, Renderer.hpp:41 (reference: 5; 1..16 accepted) */
	uint32_t mis;           /* #define MIS, Renderer.hpp:71 */
	uint32_t use_bvh;       /* #define USEBVH, BVH.hpp:307 (reference ships 0 = brute force) */
	uint32_t count_traffic; /* 1: kernels also count BVH nodes / spheres visited (slower; for the roofline's algorithmic bytes) */
	uint32_t profile;       /* 1: bracket every kernel launch with HIP events (mirt_get_kernel_times) */
	uint32_t max_batch;     /* Accumulate() calls traced together as one batch: 1..256, clamped to what the context's path ids and stream slots hold —
	                           2^30 / (its pixel count rounded up to a power of two), and pixels x batch + 12288 <= 2^30 (a context that owns all 2^24 pixels of a
	                           4096 x 4096 image: 63); 0 = auto, about 1 G primary rays per batch within the free device memory.  mirt_get_policy
	                           reports the value in effect.  Results do not depend on it: adds reach every bucket in accumulation order. */
	uint32_t reference_tree;/* 0 (default): traverse a GPU-internal SAH tree built over the same BVH-order prims; 1: traverse the caller's
	                         * nodes as handed over.  Results are identical either way (DESIGN.md "Traversal semantics"); read at mirt_set_scene. */
	uint32_t streams;       /* batches of accumulations kept in flight on separate HIP streams (0 = default 3, 1 = one kernel at a time).
	                         * Each in-flight batch renders into its own contribution buffer; the buffers are added to the accumulator in
	                         * accumulation order, so results do not depend on this value. */
	uint32_t gpu_build;     /* 1: the GPU-internal traversal tree is built on the GPU at mirt_set_scene (Morton-order LBVH, milliseconds) instead of
	                         * the host SAH sweep (better tree, 0.3 s per 100 k spheres): for the edit-rebuild loop (Application.cpp:508).  Results
	                         * are identical either way; ignored with reference_tree = 1 or fewer than 2 spheres.  Read at mirt_set_scene. */
	uint32_t trace_primary_rays; /* 0 (default): within a batch, the camera rays of a pixel (one jittered sample per accumulation of the batch, up to 256) share ONE cone traversal that lists the
	                         * spheres they can hit; each sample then tests only those, with the reference's arithmetic.  1: every primary ray walks the
	                         * tree by itself (measurements; the traversal-twin counter checks).  Results are identical either way. */
	uint32_t _reserved[1];
} mirt_policy;

typedef struct mirt_counters {
	uint64_t rays;            /* rays handed to closest-hit traversal: primary + extension (Renderer.hpp:165) */
	uint64_t shadow_rays;     /* rays handed to any-hit traversal (Renderer.hpp:302) */
	uint64_t nodes;           /* BVH nodes box-tested by closest-hit traversal (count_traffic) */
	uint64_t spheres;         /* spheres tested by closest-hit traversal (count_traffic) */
	uint64_t shadow_nodes;
	uint64_t shadow_spheres;
	uint64_t terminated;      /* paths added into the accumulator (Renderer.hpp:424-430) */
	uint64_t dropped;         /* paths still alive after the last bounce, radiance dropped (Q5) */
} mirt_counters;

/* Kernel classes for mirt_get_kernel_times.  MIRT_K_TRACE includes the shadow rays and their deferred adds (traced in the
 * same launches); MIRT_K_SHADOW is kept for ABI stability and stays 0. */
enum { MIRT_K_RAYGEN = 0, MIRT_K_TRACE = 1, MIRT_K_SHADE = 2, MIRT_K_SHADOW = 3, MIRT_K_RESOLVE = 4, MIRT_K_COUNT = 5 };
typedef struct mirt_kernel_times {
	double   ms[MIRT_K_COUNT];        /* summed HIP-event time per kernel class since the last reset of the timers */
	uint64_t launches[MIRT_K_COUNT];
} mirt_kernel_times;

typedef struct mirt_ctx mirt_ctx;

/* Renderer(const Scene&) ctor, Renderer.hpp:51.  device = HIP device ordinal. */
int mirt_create(int device, mirt_ctx** out);
int mirt_destroy(mirt_ctx* ctx);
/* Last error text of ctx (or of the failed mirt_create when ctx == NULL). */
const char* mirt_last_error(const mirt_ctx* ctx);

/* BoundingVolumeHierarchy<Sphere> ctor, BVH.hpp:90-206 (host, runs on every scene edit, Application.cpp:233,508).
 * nodes_out capacity >= 2*n; prims_out capacity n (BVH-order copy of geometry, BVH.hpp:201-205). */
int mirt_bvh_build(const mirt_sphere* geometry, uint32_t n, mirt_bvh_node* nodes_out, uint32_t* n_nodes_out, mirt_sphere* prims_out);
/* LightingAcceleration ctor, Scene.hpp:12-16: geometry-order indices with dot(emission,emission) > 0. lights_out capacity n. */
int mirt_light_list(const mirt_sphere* geometry, uint32_t n, const mirt_material* materials, uint32_t n_materials,
                    int32_t* lights_out, uint32_t* n_lights_out);

/* The `const Scene& scene` the renderer reads (Scene.hpp:19-26): geometry (authoring order, used by NEE
 * Renderer.hpp:262), acceleration_structure.{nodes,prims}, material, lighting_acceleration.prims, sky
 * (Primitives.hpp:29-47; hdri_rgba = RGBA f32 texels, >= 1x1).  Call again after any scene edit
 * (Application.cpp:508-510); it does not reset the accumulator. */
int mirt_set_scene(mirt_ctx* ctx,
                   const mirt_sphere* geometry, const mirt_sphere* bvh_prims, uint32_t n_spheres,
                   const mirt_bvh_node* nodes, uint32_t n_nodes,
                   const mirt_material* materials, uint32_t n_materials,
                   const int32_t* lights, uint32_t n_lights,
                   const float ambient_color[3], const float* hdri_rgba, uint32_t hdri_w, uint32_t hdri_h);

/* scene.camera fields the path reads (Camera.hpp:80-88; Renderer.hpp:439): view.pos, view.orient (x,y,z,w),
 * projection.half_width / half_height / z, exp. */
int mirt_set_camera(mirt_ctx* ctx, const float pos[3], const float orient_xyzw[4],
                    float half_width, float half_height, float z, float exposure);

int mirt_set_policy(mirt_ctx* ctx, const mirt_policy* policy);
int mirt_get_policy(const mirt_ctx* ctx, mirt_policy* policy);   /* with the values in effect for max_batch / streams left at 0 */

/* Renderer::Resize, Renderer.hpp:53-63: h_tiles = w/16, v_tiles = h/16 (truncating), allocates and zeroes
 * the accumulator, accumulations = 0.  Owns all tiles until mirt_set_tile_range says otherwise. */
int mirt_resize(mirt_ctx* ctx, uint32_t width, uint32_t height);
/* Multi-GPU sharding of the parallel_for range (Renderer.hpp:75): this context renders LaunchIndex in
 * [first_tile, first_tile + n_tiles).  RNG seeds use the global LaunchIndex (Renderer.hpp:107), so any
 * partition reproduces the single-context result bit for bit.  Reallocates and zeroes the accumulator. */
int mirt_set_tile_range(mirt_ctx* ctx, uint32_t first_tile, uint32_t n_tiles);
/* The same with interleaved tile rows: this context renders the tile rows first_row, first_row + row_stride, ... (a tile row =
 * width/16 consecutive LaunchIndices).  With rank r of N taking (r, N) every GPU sees the whole image height, sky and ground
 * alike: contiguous eighths of the weak-scaling image differ by 1.5x in cost (profiles/experiments/shard_balance.py).  The
 * accumulator slab holds the owned rows in ascending order.  Reallocates and zeroes the accumulator. */
int mirt_set_tile_rows(mirt_ctx* ctx, uint32_t first_row, uint32_t row_stride);
/* Renderer::ResetAccumulator, Renderer.hpp:64-67 */
int mirt_reset(mirt_ctx* ctx);

/* n_calls x Renderer::Accumulate(), Renderer.hpp:73-434 (each call = 1 sample per pixel, ++accumulations first). */
int mirt_accumulate(mirt_ctx* ctx, uint32_t n_calls);
/* Same, without waiting for the GPU.  Whole batches (policy.max_batch) are enqueued on the context's HIP streams at once; a
 * remainder is kept until later calls complete the batch or something needs it — mirt_synchronize, mirt_render when a frame
 * is due, any read of results, any change of scene / camera / policy (the deferred calls are launched with the state they
 * were issued under).  A host that calls this once per frame, as the reference's UI loop calls Accumulate() (Application.cpp:379),
 * therefore still gets full-size launches: 2.1 ms -> 0.8 ms per 1024x1024 frame with Render() called every frame (due every 5th). */
int mirt_accumulate_async(mirt_ctx* ctx, uint32_t n_calls);
/* Launches anything deferred and waits for the GPU. */
int mirt_synchronize(mirt_ctx* ctx);
/* Accumulate() calls issued so far (launched or deferred). */
int mirt_get_accumulations(const mirt_ctx* ctx, uint32_t* accumulations);

/* `accumulator` member, Renderer.hpp:43-46: [local tile][bucket][r,g,b][256] f32. */
int mirt_accumulator_floats(const mirt_ctx* ctx, size_t* n_floats);
int mirt_read_accumulator(mirt_ctx* ctx, float* host_dst);
/* Device address of that slab (for an RCCL gather by the caller); valid until the next resize / tile-range call.
 * Launches anything deferred and waits for the GPU first, so the slab is complete whatever stream the caller reads it on. */
int mirt_accumulator_device(mirt_ctx* ctx, void** device_ptr, size_t* bytes);
/* Checkpoint/resume and post-gather resolve: overwrite the slab (src on host or device) and set `accumulations`. */
int mirt_load_accumulator(mirt_ctx* ctx, const float* src, int src_is_device, uint32_t accumulations);

/* Renderer::Render, Renderer.hpp:436-478: median over buckets * exposure/(accumulations/buckets), ACES tonemap,
 * RGBA f32 row-major width*height (row 0 = y 0), A = 1.  Returns MIRT_NOT_READY and leaves rgba_host untouched
 * when accumulations % buckets != 0.  Only the context's own tiles are written. */
int mirt_render(mirt_ctx* ctx, float* rgba_host);

int mirt_get_counters(mirt_ctx* ctx, mirt_counters* out);
int mirt_get_kernel_times(mirt_ctx* ctx, mirt_kernel_times* out, int reset);
/* HIP stream the context launches on (hipStream_t), for callers that time with their own events. */
int mirt_get_stream(mirt_ctx* ctx, void** hip_stream);

/* ---- multi-GPU: the same renderer on n devices of one node ------------------------------------------------------
 * The reference host is ONE object, `Renderer<> renderer{scene}` (Application.cpp:514), whose Accumulate() is a parallel_for over
 * the tiles (Renderer.hpp:75).  A group is that object on n GPUs, driven by one host thread of one process: one context per
 * device with the whole scene and an interleaved share of the tile rows (member i: rows i, i+n, ...; mirt_set_tile_rows), no
 * exchange while rendering, and ONE gather of the accumulator slabs to devices[0] — RCCL point-to-point over xGMI (librccl is loaded
 * on first use) — with a device-side un-interleave into the full-image AccumulationTile layout, where Render() resolves the whole
 * frame.  Results equal the single-context ones bit for bit (random draws are keyed on the global LaunchIndex, Renderer.hpp:107).
 * Entries mirror their mirt_* namesakes; errors: mirt_group_last_error.  devices[] may name one device more than once (rehearsal on
 * a one-GPU box: slabs are then exchanged with device copies, RCCL needs distinct devices).
 * (One process PER GPU — torch.distributed / MPI ranks — uses plain contexts with mirt_set_tile_rows + mirt_accumulator_device instead.) */
typedef struct mirt_group mirt_group;
int mirt_group_create(const int* devices, int n, mirt_group** out);
int mirt_group_destroy(mirt_group* group);
const char* mirt_group_last_error(const mirt_group* group);       /* NULL: of the failed mirt_group_create / selftest */
int mirt_group_size(const mirt_group* group, int* n);
int mirt_group_member(mirt_group* group, int index, mirt_ctx** ctx);   /* borrowed: counters, kernel times, debug entry points */
int mirt_group_set_scene(mirt_group* group,
                         const mirt_sphere* geometry, const mirt_sphere* bvh_prims, uint32_t n_spheres,
                         const mirt_bvh_node* nodes, uint32_t n_nodes,
                         const mirt_material* materials, uint32_t n_materials,
                         const int32_t* lights, uint32_t n_lights,
                         const float ambient_color[3], const float* hdri_rgba, uint32_t hdri_w, uint32_t hdri_h);
int mirt_group_set_camera(mirt_group* group, const float pos[3], const float orient_xyzw[4], float half_width, float half_height, float z, float exposure);
int mirt_group_set_policy(mirt_group* group, const mirt_policy* policy);
int mirt_group_resize(mirt_group* group, uint32_t width, uint32_t height);      /* Renderer::Resize + the tile-row split */
int mirt_group_reset(mirt_group* group);                                         /* Renderer::ResetAccumulator */
int mirt_group_accumulate(mirt_group* group, uint32_t n_calls);                  /* n x Renderer::Accumulate on every device, then waits */
int mirt_group_accumulate_async(mirt_group* group, uint32_t n_calls);
int mirt_group_synchronize(mirt_group* group);
int mirt_group_get_accumulations(const mirt_group* group, uint32_t* accumulations);
int mirt_group_get_counters(mirt_group* group, mirt_counters* out);              /* summed over the members */
/* The one exchange: slabs -> devices[0] (ncclSend / ncclRecv per peer, rccl.h:700,722), un-interleaved there.  Implied by the two
 * reads below; a no-op for one member or when nothing was accumulated since the last gather. */
int mirt_group_gather(mirt_group* group);
int mirt_group_last_gather_ms(const mirt_group* group, double* ms);              /* device time of the last gather incl. un-interleave */
int mirt_group_accumulator_floats(const mirt_group* group, size_t* n_floats);    /* of the whole image */
int mirt_group_read_accumulator(mirt_group* group, float* host_dst);             /* whole image, [tile][bucket][r,g,b][256] in LaunchIndex order */
int mirt_group_render(mirt_group* group, float* rgba_host);                      /* Renderer::Render of the whole frame; MIRT_NOT_READY as mirt_render */
/* Diagnostic: one-device RCCL communicator on `device`, n_floats sent to itself through a grouped ncclSend / ncclRecv. */
int mirt_group_rccl_selftest(int device, size_t n_floats);

/* ---- stage-level entry points (parity tests of single kernels) -------------------------------------
 * SoA planes: p_xyz / dir_xyz hold x[n], y[n], z[n] back to back. */
/* RAY GENERATION, Renderer.hpp:113-127 for Accumulate() number `accumulations`; ray order tile*256 + ID over local tiles. */
int mirt_debug_raygen(mirt_ctx* ctx, uint32_t accumulations, float* p_xyz, float* dir_xyz);
/* Traverse (BVH.hpp:309-360): tfar starts at FLT_MAX, primID at -1 (BVH-order index). */
int mirt_debug_trace_closest(mirt_ctx* ctx, size_t n, const float* p_xyz, const float* dir_xyz, float* tfar_out, int32_t* primID_out);
/* Traverse_shadow (BVH.hpp:362-404): occluded_out[i] in {0,1}. */
int mirt_debug_trace_shadow(mirt_ctx* ctx, size_t n, const float* p_xyz, const float* dir_xyz, const float* tfar, uint8_t* occluded_out);
/* Device math used by the shading kernels, evaluated on the GPU for n inputs.
 * fn: 0 fast_sincos(x)->(sin,cos)  [VectorMath.hpp:644-662]   in: x[n]            out: 2n
 *     1 fast_atan2(y,x)            [VectorMath.hpp:632-642]   in: y[n],x[n]       out: n
 *     2 fast_asin(x)               [VectorMath.hpp:625-630]   in: x[n]            out: n
 *     3 1/x, sqrt(x), a/b checks                               in: a[n],b[n]       out: 3n (1/a, sqrt(|a|), a/b)
 *     4 hemisphere(t,s)            [Sampling.hpp:92-94]       in: t[n],s[n]       out: 3n
 *     5 tangent_space(N)+to_local/to_world round trip [Sampling.hpp:150-179]  in: N(3n),v(3n)  out: 10n (T xyzw, local xyz, world xyz)
 *     6 sample_direction_to_sphere [Sampling.hpp:220-239]     in: Wc(3n),sin2[n],dist[n],r2[n],t[n],s[n]  out: 5n (L xyz, distance, pdf)
 *     7 rng: hash_2d(x,y) then 3 pcg draws as float + bounded int  [Random.hpp:5-50]  in: x[n],y[n],range[n] as u32 bits  out: 5n (hash bits, f0,f1,f2, bounded bits)
 *     8 Closure<GGX>::eval         [DataStreams.hpp:189-195, Sampling.hpp:272-296]  in: F0(3n),alpha[n],Llocal(3n),Vlocal(3n)  out: 3n
 *     9 Closure<GGX>::sample       [DataStreams.hpp:200-218, Sampling.hpp:254-270,297-309]  in: F0(3n),alpha[n],Vlocal(3n),u0[n],u1[n]  out: 6n (dir, estimator)
 *    10 the sphere tests' sqrt (kernels.hpp sqrt_trav) — must equal IEEE sqrt for every input >= 0   in: x[n]  out: n
 *       (8, 9: the defined part of the reference's compiled-out GGX closure; not used by the path — DESIGN.md §7)
 */
int mirt_debug_math(mirt_ctx* ctx, int fn, size_t n, const float* in, float* out);
/* Introspection of the GPU-internal BVH layout (tests, bench, DESIGN.md numbers):
 * out[0] records, out[1] records staged in LDS, out[2] spheres staged in LDS, out[3] tree depth, out[4] bit 0: binary16
 * records are in use, bit 1: they are the 64-B records of up to four children (else 32-B child pairs), out[5] dynamic LDS
 * bytes of a trace workgroup, out[6] trace workgroups per CU, out[7] CUs. */
int mirt_debug_info(mirt_ctx* ctx, uint32_t out[8]);
/* Length histogram of the per-pixel candidate lists of the current scene / camera / size (policy.trace_primary_rays = 0 path):
 * hist[n] = local pixels whose bundle of camera rays can hit n spheres for n = 0..7, hist[8] = 8 or more (lists hold up to 31), hist[9] = pixels without a list (traced normally). */
int mirt_debug_primary_lists(mirt_ctx* ctx, uint32_t hist[10]);
/* Test knob: forbid (0) / allow (1, default) the binary16 records; takes effect at the next mirt_set_scene. */
int mirt_debug_allow_half_boxes(mirt_ctx* ctx, int allow);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
