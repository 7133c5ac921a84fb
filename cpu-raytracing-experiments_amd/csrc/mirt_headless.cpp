// mirt_headless.cpp — headless stand-in for RaytracingApp (Application.cpp:32-235, 361-386): builds a scene the way
// the reference's constructor does, then drives the renderer with the reference's call protocol
// (Resize -> [Accumulate, Render] per frame) through mirt_host.hpp, and writes the resolved frame in the reference's own screenshot
// format — Radiance .hdr, flipped vertically like Image::Store (Image.cpp:71-74) — or as a PFM (by the extension of --out).  --hdri loads the
// environment map the way the constructor does (stbi_loadf(..., 4), Application.cpp:225-231); --ambient sets sky.ambient_color, which scales it.
//
//   mirt_headless --scene default9|furnace|bvh_test|synthetic:N [--size WxH] [--spp N | --frames N] [--bounces B] [--buckets K] [--brute] [--devices 0,1,..]
//                 [--hdri env.hdr] [--ambient A] [--out frame.hdr|frame.pfm]
// --devices: the GPUs the one Renderer object uses (tile rows split over them inside the library, one RCCL gather per frame read).
// --frames N is the UI loop itself (Application.cpp:373-380): N frames of { Accumulate(); Render(); }; the report lists the frames on
// which Render() produced output (every `buckets`-th, Renderer.hpp:437) and a hash of the last frame shown.
#include "mirt_host.hpp"
#include "hdr_io.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using namespace mirt;

static Material make_material(float ar, float ag, float ab, float er = 0, float eg = 0, float eb = 0) {
	Material m{};
	m.albedo[0] = ar; m.albedo[1] = ag; m.albedo[2] = ab;
	m.emission[0] = er; m.emission[1] = eg; m.emission[2] = eb;
	return m;
}
static Sphere make_sphere(float x, float y, float z, float radius_sq, int32_t mat) {
	Sphere s{};
	s.position[0] = x; s.position[1] = y; s.position[2] = z; s.radius_sq = radius_sq; s.material_ID = mat;
	return s;
}

// Scenes::Default, Application.cpp:33-101
static void scene_default9(Scene& sc) {
	sc.camera = Camera{ vec3{ -0.2f, 0.3f, 1.0f }, vec3{ 0.1f, -0.4f, -1.0f }, 40.0f, 1.0f };
	sc.material = {
		make_material(1, 1, 1), make_material(1, 1, 1, 0.1f * 25.0f, 0.1f * 25.0f, 0.1f * 200.0f), make_material(1, 1, 1, 0.1f * 150.0f, 0.1f * 150.0f, 0.1f * 150.0f),
		make_material(1, 1, 1, 200.0f, 17.0f, 25.0f), make_material(0.793f, 0.793f, 0.664f), make_material(0.05f, 0.05f, 0.05f),
		make_material(1, 1, 1), make_material(1, 1, 1), make_material(1, 1, 1) };
	sc.geometry = {
		make_sphere(0.3f, -1.47f, 0.0f, 1.5f * 1.5f, 0), make_sphere(0.29999f, 0.0801f, 0.0f, 0.05f * 0.05f, 1), make_sphere(0.3302f, 0.36165f, 0.7119f, 0.05f * 0.05f, 2),
		make_sphere(-0.4857f, -0.0242f, -0.41383f, 0.05f * 0.05f, 3), make_sphere(0.3f, 1.7f, 0.0f, 1.5f * 1.5f, 4), make_sphere(0.018f, 0.022f, 0.07f, 0.02f * 0.02f, 5),
		make_sphere(-0.037f, 0.022f, 0.0f, 0.03f * 0.03f, 6), make_sphere(-0.0846f, -0.0334f, 0.283f, 0.012f * 0.012f, 7), make_sphere(0.03863f, -0.00788f, 0.2835f, 0.012f * 0.012f, 8) };
}
// Scenes::White_Furnace, Application.cpp:218-223
static void scene_furnace(Scene& sc) {
	sc.camera = Camera{ vec3{ 0, 0, 3 }, vec3{ 0, 0, -1 } };
	sc.material = { make_material(1, 1, 1) };
	sc.geometry = { make_sphere(0, 0, 0, 1.0f, 0) };
	sc.sky.ambient_color[0] = sc.sky.ambient_color[1] = sc.sky.ambient_color[2] = 1.0f;
}
// S(n) of SURVEY.md §8d, generator = the reference's PCG (Random.hpp:5-43) — same scene as scene.py's synthetic()
static uint32_t hash_u32(uint32_t i) { i ^= i >> 16; i *= 0x21f0aaadu; i ^= i >> 15; i *= 0xd35a2d97u; i ^= i >> 15; return i ^ 0xe6fe3bebu; }
static float next_unit(uint32_t& s) {
	uint32_t v = s; s = s * 747796405u + 2891336453u;
	v = ((v >> ((v >> 28u) + 4u)) ^ v) * 277803737u; v = (v >> 22u) ^ v;
	return static_cast<float>(v) * 0x1p-32f;
}
static void scene_synthetic(Scene& sc, uint32_t n, float ambient) {
	const float cb = static_cast<float>(std::cbrt(static_cast<double>(n))), H = cb, L = cb * 2.0f;
	uint32_t st = hash_u32(1);
	sc.material.assign(17, Material{});
	for (int m = 0; m < 16; m++) for (int c = 0; c < 3; c++) sc.material[m].albedo[c] = 0.2f + 0.6f * next_unit(st);
	sc.material[16] = make_material(1, 1, 1, 20, 20, 20);
	sc.geometry.assign(n, Sphere{});
	sc.geometry[0] = make_sphere(0, -1000.0f, 0, 1000.0f * 1000.0f, 0);
	for (uint32_t i = 1; i < n; i++) {
		const float u0 = next_unit(st), u1 = next_unit(st), u2 = next_unit(st), u3 = next_unit(st);
		const float r = 0.3f + 0.7f * u3;
		sc.geometry[i] = make_sphere(-L + (2.0f * L) * u0, 0.2f + (H - 0.2f) * u1, -L + (2.0f * L) * u2, r * r, static_cast<int32_t>(i % 16));
	}
	for (uint32_t i = 0; i < n; i++) if ((i % 64) == 63 || (n < 64 && i == n - 1)) sc.geometry[i].material_ID = 16;
	sc.camera = Camera{ vec3{ 0.0f, H, 3.0f * L }, vec3{ 0.0f, -0.3f, -1.0f }, 40.0f, 1.0f };
	sc.sky.ambient_color[0] = sc.sky.ambient_color[1] = sc.sky.ambient_color[2] = ambient;
}

// Scenes::BVH_test, Application.cpp:102-122, fixed version (see scene.py bvh_test(): the shipped scene has an empty material list and
// a std::mt19937 whose distributions differ between standard libraries): eight Lambertian materials, the reference's PCG seeded with
// hash_u32 of the low 32 bits of the shipped seed, draws per sphere in the shipped order radius, x, y, z, material.
static void scene_bvh_test(Scene& sc) {
	uint32_t st = hash_u32(0x04d15a07u);
	sc.material.assign(8, Material{});
	for (int m = 0; m < 8; m++) for (int c = 0; c < 3; c++) sc.material[m].albedo[c] = 0.2f + 0.7f * next_unit(st);
	sc.geometry.clear();
	for (int i = 0; i < 255; i++) {
		const float r = 0.3f + 19.7f * next_unit(st);
		const float x = -100.0f + 200.0f * next_unit(st);
		const float y = 100.0f * next_unit(st);
		const float z = -100.0f + 200.0f * next_unit(st);
		const uint32_t m = static_cast<uint32_t>(next_unit(st) * 8.0f);
		sc.geometry.push_back(make_sphere(x, y, z, r * r, static_cast<int32_t>(m < 7u ? m : 7u)));
	}
	sc.camera = Camera{ vec3{ 0, 60, 300 }, vec3{ 0, 0, -1 } };
	sc.sky.ambient_color[0] = sc.sky.ambient_color[1] = sc.sky.ambient_color[2] = 1.0f;
}

static uint64_t fnv1a(const std::vector<float>& v) {
	uint64_t hsh = 1469598103934665603ull;
	for (float f : v) { uint32_t u; std::memcpy(&u, &f, 4); for (int b = 0; b < 4; b++) { hsh ^= (u >> (8 * b)) & 0xffu; hsh *= 1099511628211ull; } }
	return hsh;
}

static bool write_pfm(const std::string& path, const std::vector<float>& rgba, uint32_t w, uint32_t h) {
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) return false;
	std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);                 // little-endian; PFM rows run bottom-to-top = framebuffer order (y 0 = bottom)
	std::vector<float> row(static_cast<size_t>(w) * 3);
	for (uint32_t y = 0; y < h; y++) {
		for (uint32_t x = 0; x < w; x++) for (int c = 0; c < 3; c++) row[x * 3 + c] = rgba[(static_cast<size_t>(y) * w + x) * 4 + c];
		std::fwrite(row.data(), sizeof(float), row.size(), f);
	}
	std::fclose(f);
	return true;
}

int main(int argc, char** argv) {
	std::string scene_name = "default9", out, hdri;
	uint32_t w = 512, h = 512, spp = 10, n = 0;
	RendererPolicy policy;
	float ambient = 0.0f;
	bool ambient_set = false;
	std::vector<int> devices = { 0 };
	if (argc == 4 && std::string(argv[1]) == "--convert-hdr") {
		// file-format check without a GPU: read a picture like stbi_loadf does (top-down) and store it again like Image::Store does (which
		// flips, so the rows are handed over bottom-up): the output decodes to the same texels
		std::vector<float> texels; int32_t iw = 0, ih = 0;
		const std::string why = mirt_hdr::read(argv[2], texels, iw, ih);
		if (!why.empty()) { std::fprintf(stderr, "mirt_headless: %s: %s\n", argv[2], why.c_str()); return 1; }
		std::vector<float> flipped(texels.size());
		for (int32_t y = 0; y < ih; y++) std::memcpy(&flipped[static_cast<size_t>(y) * iw * 4], &texels[static_cast<size_t>(ih - 1 - y) * iw * 4], static_cast<size_t>(iw) * 16);
		return mirt_hdr::write_flipped(argv[3], flipped.data(), static_cast<uint32_t>(iw), static_cast<uint32_t>(ih)) ? 0 : 1;
	}
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
		if (a == "--scene") scene_name = next();
		else if (a == "--size") { if (std::sscanf(next(), "%ux%u", &w, &h) != 2) return 2; }
		else if (a == "--spp" || a == "--frames") spp = static_cast<uint32_t>(std::atoi(next()));
		else if (a == "--bounces") policy.max_bounces = static_cast<uint32_t>(std::atoi(next()));
		else if (a == "--buckets") policy.buckets = static_cast<uint32_t>(std::atoi(next()));
		else if (a == "--ambient") { ambient = static_cast<float>(std::atof(next())); ambient_set = true; }
		else if (a == "--hdri") hdri = next();
		else if (a == "--brute") policy.use_bvh = false;
		else if (a == "--out") out = next();
		else if (a == "--devices") { devices.clear(); for (const char* p = next(); *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; } if (devices.empty()) return 2; }
		else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
	}
	try {
		Scene scene;
		if (scene_name == "default9") scene_default9(scene);
		else if (scene_name == "furnace") scene_furnace(scene);
		else if (scene_name == "bvh_test") scene_bvh_test(scene);
		else if (scene_name.rfind("synthetic:", 0) == 0) { n = static_cast<uint32_t>(std::atoi(scene_name.c_str() + 10)); if (n < 2) return 2; scene_synthetic(scene, n, ambient); }
		else { std::fprintf(stderr, "unknown scene %s\n", scene_name.c_str()); return 2; }
		if (ambient_set) scene.sky.ambient_color[0] = scene.sky.ambient_color[1] = scene.sky.ambient_color[2] = ambient;
		if (!hdri.empty()) {                                           // Application.cpp:225-231
			const std::string why = mirt_hdr::read(hdri, scene.sky.hdri_data, scene.sky.hdri_width, scene.sky.hdri_height);
			if (!why.empty()) { std::fprintf(stderr, "mirt_headless: %s: %s\n", hdri.c_str(), why.c_str()); return 1; }
		}
		scene.RebuildAcceleration();                                   // Application.cpp:233-234

		Renderer renderer{ scene, policy, devices };
		// pad the viewport to the tile requirement like UIRender does (Application.cpp:365-372)
		const uint32_t t = static_cast<uint32_t>(Renderer::RequiredTiling());
		w = (w + t - 1) & ~(t - 1); h = (h + t - 1) & ~(t - 1);
		scene.camera.Resize(w, h);                                     // Application.cpp:375-376
		renderer.SceneChanged();
		renderer.Resize(w, h);

		const auto t0 = std::chrono::steady_clock::now();
		bool have_frame = false;
		std::string frames_due;                                        // 1-based frame numbers on which Render() produced output
		for (uint32_t frame = 0; frame < spp; frame++) {               // one UIRender per frame: Accumulate(); Render();  (Application.cpp:379-380)
			renderer.Accumulate();
			if (renderer.Render()) { have_frame = true; frames_due += (frames_due.empty() ? "" : ", ") + std::to_string(frame + 1); }
		}
		const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		const mirt_counters c = renderer.counters();

		// FNV-1a over the raw accumulator words: lets a test compare this C++ host against the Python host bit for bit
		const std::vector<float> acc = renderer.accumulator();
		const uint64_t hsh = fnv1a(acc);
		const uint64_t frame_hsh = have_frame ? fnv1a(renderer.GetFrame()) : 0ull;

		std::printf("{\"scene\": \"%s\", \"spheres\": %zu, \"nodes\": %zu, \"lights\": %zu, \"width\": %u, \"height\": %u, \"accumulations\": %u, "
		            "\"rays\": %llu, \"shadow_rays\": %llu, \"terminated\": %llu, \"dropped\": %llu, \"seconds\": %.6f, \"mray_per_s\": %.3f, "
		            "\"accumulator_fnv1a\": \"%016llx\", \"frame_ready\": %s, \"frames_due\": [%s], \"last_frame_fnv1a\": \"%016llx\", \"gpus\": %zu, \"gather_ms\": %.3f}\n",
		            scene_name.c_str(), scene.geometry.size(), scene.acceleration_structure.nodes.size(), scene.lighting_acceleration.prims.size(), w, h,
		            renderer.accumulations(), (unsigned long long)c.rays, (unsigned long long)c.shadow_rays, (unsigned long long)c.terminated,
		            (unsigned long long)c.dropped, sec, c.rays / sec / 1e6, (unsigned long long)hsh, have_frame ? "true" : "false", frames_due.c_str(), (unsigned long long)frame_hsh, devices.size(), renderer.gather_ms());
		if (!out.empty() && have_frame) {
			const bool as_hdr = out.size() >= 4 && out.compare(out.size() - 4, 4, ".hdr") == 0;
			const bool ok = as_hdr ? mirt_hdr::write_flipped(out, renderer.GetFrame().data(), w, h) : write_pfm(out, renderer.GetFrame(), w, h);
			if (!ok) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
		}
	} catch (const std::exception& e) {
		std::fprintf(stderr, "mirt_headless: %s\n", e.what());
		return 1;
	}
	return 0;
}
