// mirt_host.hpp — C++ host-side mirror of the reference's Renderer / Scene interface over the C-ABI (mirt.h).
//
// The reference host is C++ (Application.cpp); this header gives it the same names with the same meaning:
//   mirt::Sphere / Material / Node      Primitives.hpp:7-27, BVH.hpp:18-31 (byte-identical)
//   mirt::Camera                        Camera.hpp:5-89 (fields the path reads + Projection::Resize/UpdateLens + quatLookAt)
//   mirt::Scene                         Scene.hpp:19-26 (geometry, material, lighting_acceleration, camera, sky, acceleration_structure)
//   mirt::Renderer                      Renderer.hpp:28-68,73,436: Resize / ResetAccumulator / Accumulate / Render / GetFrame
// Failures surface as std::runtime_error carrying mirt_last_error() (the reference asserts / terminates instead).
#pragma once
#include "../../include/mirt.h"

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace mirt {

using Sphere = mirt_sphere;
using Material = mirt_material;
using Node = mirt_bvh_node;
static_assert(sizeof(Sphere) == 32 && sizeof(Material) == 96 && sizeof(Node) == 32, "reference layouts");

struct vec3 { float x = 0, y = 0, z = 0; };
struct quat { float x = 0, y = 0, z = 0, w = 1; };

// Camera.hpp:5-89 — pinhole camera; aperture / focus fields are unused by the path (Q18) and omitted.
struct Camera {
	vec3 pos;
	quat orient;
	float half_width = 0.5f, half_height = 0.5f, z = 0.0f;
	float focal_length = 50.0f, exp = 1.0f;

	Camera() { Resize(1, 1); }
	Camera(vec3 eye, vec3 direction, float focal_length_mm = 50.0f, float exposure = 1.0f) : pos(eye), focal_length(focal_length_mm), exp(exposure) {
		const float inv = 1.0f / std::sqrt(direction.x * direction.x + direction.y * direction.y + direction.z * direction.z);
		orient = look_at(vec3{ direction.x * inv, direction.y * inv, direction.z * inv });
		Resize(1, 1);
	}
	void Resize(uint32_t width, uint32_t height) {                         // Projection::Resize + UpdateLens, Camera.hpp:20-31
		half_height = static_cast<float>(height) * 0.5f;
		half_width = static_cast<float>(width) * 0.5f;
		z = half_height * ((-2.0f / 24.0f) * focal_length);
	}
	static quat look_at(vec3 d) {                                          // glm::quatLookAt(direction, up = +y), right-handed
		const vec3 c2{ -d.x, -d.y, -d.z }, up{ 0, 1, 0 };
		vec3 r{ up.y * c2.z - c2.y * up.z, up.z * c2.x - c2.z * up.x, up.x * c2.y - c2.x * up.y };
		const float rr = r.x * r.x + r.y * r.y + r.z * r.z;
		const float s = 1.0f / std::sqrt(rr > 0.00001f ? rr : 0.00001f);
		const vec3 c0{ r.x * s, r.y * s, r.z * s };
		const vec3 c1{ c2.y * c0.z - c0.y * c2.z, c2.z * c0.x - c0.z * c2.x, c2.x * c0.y - c0.x * c2.y };
		const float m00 = c0.x, m01 = c0.y, m02 = c0.z, m10 = c1.x, m11 = c1.y, m12 = c1.z, m20 = c2.x, m21 = c2.y, m22 = c2.z;
		const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
		int idx = 0; float big = fw;
		if (fx > big) { big = fx; idx = 1; }
		if (fy > big) { big = fy; idx = 2; }
		if (fz > big) { big = fz; idx = 3; }
		const float b = std::sqrt(big + 1.0f) * 0.5f, mult = 0.25f / b;
		switch (idx) {
		case 0: return quat{ (m12 - m21) * mult, (m20 - m02) * mult, (m01 - m10) * mult, b };
		case 1: return quat{ b, (m01 + m10) * mult, (m20 + m02) * mult, (m12 - m21) * mult };
		case 2: return quat{ (m01 + m10) * mult, b, (m12 + m21) * mult, (m20 - m02) * mult };
		default: return quat{ (m20 + m02) * mult, (m12 + m21) * mult, b, (m01 - m10) * mult };
		}
	}
};

struct Sky {                                                               // Primitives.hpp:29-47
	float ambient_color[3] = { 0, 0, 0 };
	int32_t hdri_width = 1, hdri_height = 1;
	std::vector<float> hdri_data = { 1, 1, 1, 1 };                         // RGBA f32 (synthetic 1x1 texel by default)
};

struct BoundingVolumeHierarchy {                                           // BVH.hpp:85-86,90
	std::vector<Node> nodes;
	std::vector<Sphere> prims;
	BoundingVolumeHierarchy() = default;
	explicit BoundingVolumeHierarchy(const std::vector<Sphere>& primitives) {
		nodes.resize(primitives.size() * 2 + 1);
		prims.resize(primitives.size());
		uint32_t n = 0;
		if (mirt_bvh_build(primitives.data(), static_cast<uint32_t>(primitives.size()), nodes.data(), &n, prims.data()) != MIRT_OK)
			throw std::runtime_error("mirt_bvh_build failed");
		nodes.resize(n);
	}
};
struct LightingAcceleration {                                              // Scene.hpp:9-17
	std::vector<int32_t> prims;
	LightingAcceleration() = default;
	LightingAcceleration(const std::vector<Sphere>& src, const std::vector<Material>& material) {
		prims.resize(src.size() + 1);
		uint32_t n = 0;
		if (mirt_light_list(src.data(), static_cast<uint32_t>(src.size()), material.data(), static_cast<uint32_t>(material.size()), prims.data(), &n) != MIRT_OK)
			throw std::runtime_error("mirt_light_list failed (material_ID out of range?)");
		prims.resize(n);
	}
};
struct Scene {                                                             // Scene.hpp:19-26
	std::vector<Sphere> geometry;
	std::vector<Material> material;
	LightingAcceleration lighting_acceleration;
	Camera camera;
	Sky sky;
	BoundingVolumeHierarchy acceleration_structure;
	void RebuildAcceleration() {                                           // Application.cpp:233-234, 508-509
		acceleration_structure = BoundingVolumeHierarchy{ geometry };
		lighting_acceleration = LightingAcceleration{ geometry, material };
	}
};

struct RendererPolicy {                                                    // Renderer.hpp:19-26 (+ the path's #defines)
	uint32_t max_bounces = 16;
	uint32_t buckets = 5;                                                  // AccumulationBuckets, Renderer.hpp:41
	bool mis = true;                                                       // #define MIS true, Renderer.hpp:71
	bool use_bvh = true;                                                   // reference ships USEBVH false (BVH.hpp:307); results are identical
	bool reference_tree = false;                                           // true: traverse scene.acceleration_structure.nodes as is instead of the internal SAH tree
	bool gpu_build = false;                                                // true: internal tree built on the GPU (LBVH) at SceneChanged(): faster rebuild, slower rays
};

class Renderer {
public:
	static constexpr size_t RequiredTiling() { return MIRT_TILE_ROOT; }    // Renderer.hpp:36

	// One renderer object, as in the reference (Application.cpp:514) — on one GPU or on several of the node: `devices` lists the HIP
	// ordinals; the library splits the tile rows over them and gathers the accumulator with one RCCL exchange (mirt.h, mirt_group_*).
	explicit Renderer(const Scene& scene_ref, RendererPolicy policy = {}, std::vector<int> devices = { 0 }) : scene(scene_ref) {
		if (mirt_group_create(devices.data(), static_cast<int>(devices.size()), &group_) != MIRT_OK) throw std::runtime_error(std::string("mirt_group_create: ") + mirt_group_last_error(nullptr));
		mirt_policy p{};
		p.max_bounces = policy.max_bounces; p.buckets = policy.buckets; p.mis = policy.mis; p.use_bvh = policy.use_bvh; p.reference_tree = policy.reference_tree; p.gpu_build = policy.gpu_build;
		if (mirt_group_set_policy(group_, &p) < 0) {                           // no destructor runs for a constructor that throws: release the group here
			const std::string why = std::string("mirt_group_set_policy: ") + mirt_group_last_error(group_);
			mirt_group_destroy(group_); group_ = nullptr;
			throw std::runtime_error(why);
		}
	}
	~Renderer() { if (group_) mirt_group_destroy(group_); }
	Renderer(const Renderer&) = delete;
	Renderer& operator=(const Renderer&) = delete;

	// The reference reads `scene` live; a copy in HBM has to be told about edits (Application.cpp:508-510).
	void SceneChanged() {
		const auto& s = scene;
		check(mirt_group_set_scene(group_, s.geometry.data(), s.acceleration_structure.prims.data(), static_cast<uint32_t>(s.geometry.size()),
		                           s.acceleration_structure.nodes.data(), static_cast<uint32_t>(s.acceleration_structure.nodes.size()),
		                           s.material.data(), static_cast<uint32_t>(s.material.size()),
		                           s.lighting_acceleration.prims.data(), static_cast<uint32_t>(s.lighting_acceleration.prims.size()),
		                           s.sky.ambient_color, s.sky.hdri_data.data(), static_cast<uint32_t>(s.sky.hdri_width), static_cast<uint32_t>(s.sky.hdri_height)),
		      "mirt_group_set_scene");
		CameraChanged();
	}
	void CameraChanged() {
		const Camera& c = scene.camera;
		check(mirt_group_set_camera(group_, &c.pos.x, &c.orient.x, c.half_width, c.half_height, c.z, c.exp), "mirt_group_set_camera");
	}
	void Resize(uint32_t new_width, uint32_t new_height) {                 // Renderer.hpp:53-63
		width = new_width; height = new_height;
		framebuffer.assign(static_cast<size_t>(width) * height * 4, 0.0f);
		check(mirt_group_resize(group_, width, height), "mirt_group_resize");
	}
	void ResetAccumulator() { check(mirt_group_reset(group_), "mirt_group_reset"); }    // Renderer.hpp:64-67
	// Renderer.hpp:73-434.  Asynchronous: Render() / counters() / the destructor wait for the GPUs; called once per frame the library
	// still batches the frames between two Render()s that are due (mirt.h, mirt_accumulate_async).
	void Accumulate(uint32_t n_calls = 1) { check(mirt_group_accumulate_async(group_, n_calls), "mirt_group_accumulate_async"); }
	void Synchronize() { check(mirt_group_synchronize(group_), "mirt_group_synchronize"); }
	bool Render() {                                                        // Renderer.hpp:436-478; false = frame unchanged (:437)
		const int rc = mirt_group_render(group_, framebuffer.data());
		check(rc, "mirt_group_render");
		return rc == MIRT_OK;
	}
	const std::vector<float>& GetFrame() const { return framebuffer; }     // RGBA f32 rows, row 0 = y 0 (Renderer.hpp:40,68)
	uint32_t accumulations() const { uint32_t a = 0; mirt_group_get_accumulations(group_, &a); return a; }
	mirt_counters counters() { mirt_counters c{}; check(mirt_group_get_counters(group_, &c), "mirt_group_get_counters"); return c; }
	std::vector<float> accumulator() {                                     // the whole image's AccumulationTile slab (Renderer.hpp:43-46)
		size_t n = 0; mirt_group_accumulator_floats(group_, &n);
		std::vector<float> acc(n);
		check(mirt_group_read_accumulator(group_, acc.data()), "mirt_group_read_accumulator");
		return acc;
	}
	double gather_ms() const { double ms = 0; mirt_group_last_gather_ms(group_, &ms); return ms; }
	mirt_group* handle() { return group_; }

	const Scene& scene;
	std::vector<float> framebuffer;
	uint32_t width = 0, height = 0;

private:
	void check(int rc, const char* what) const {
		if (rc < 0) throw std::runtime_error(std::string(what) + ": " + mirt_group_last_error(group_));
	}
	mirt_group* group_ = nullptr;
};

} // namespace mirt
