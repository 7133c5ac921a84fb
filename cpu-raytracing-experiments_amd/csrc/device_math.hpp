// device_math.hpp — gfx950 device functions for the shading side of the path tracer.
//
// Decision parity with the reference's CPU arithmetic needs the same IEEE-754
// operation sequence, so this TU is built with -ffp-contract=off (hipcc's device default is
// `fast`), IEEE div/sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), denormals
// kept, and __builtin_fmaf only where the reference fuses.  min/max are the compare-selects
// std::max/std::min/glm::max/glm::min expand to (NaN behaviour included), not v_max/v_min.
// Each function cites the reference lines it implements.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirt {

struct f3 { float x, y, z; };
struct quat { float x, y, z, w; };      // glm::quat storage order; T.z is always 0 for tangent frames

#define MIRT_DI __device__ __forceinline__

MIRT_DI float max_sel(float a, float b) { return (a < b) ? b : a; }   // std::max(a,b) == glm::max(a,b)
MIRT_DI float min_sel(float a, float b) { return (b < a) ? b : a; }   // std::min(a,b) == glm::min(a,b)
MIRT_DI float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }        // glm::dot(vec3, vec3)
MIRT_DI f3 cross3(f3 x, f3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }  // glm::cross
MIRT_DI f3 normalize3(f3 v) { float inv = 1.0f / __builtin_sqrtf(dot3(v, v)); return { v.x * inv, v.y * inv, v.z * inv }; } // glm::normalize

// glm constants (double literals narrowed to float)
#define MIRT_PI          3.14159265358979323846264338327950288f
#define MIRT_HALF_PI     1.57079632679489661923132169163975144f
#define MIRT_TWO_PI      6.28318530717958647692528676655900576f
#define MIRT_INV_PI      0.318309886183790671537767526745028724f
#define MIRT_INV_TWO_PI  0.159154943091895335768883763372514362f
#define MIRT_FLT_EPSILON 1.1920928955078125e-7f
#define MIRT_FLT_MAX     3.402823466e+38f

// ---- Random.hpp:5-50 ---------------------------------------------------------------------------
MIRT_DI float make_unit_float(uint32_t x) { return static_cast<float>(x) * 0x1p-32f; }
MIRT_DI uint32_t pcg_generate(uint32_t& s) {
	uint32_t p = s;
	s = p * 747796405u + 2891336453u;
	p = ((p >> ((p >> 28u) + 4u)) ^ p) * 277803737u;
	return (p >> 22u) ^ p;
}
MIRT_DI float rand_unit_float(uint32_t& s) { return make_unit_float(pcg_generate(s)); }
MIRT_DI uint32_t rand_bounded_int(uint32_t& s, uint32_t range) {
	uint32_t v = static_cast<uint32_t>(rand_unit_float(s) * static_cast<float>(range));
	uint32_t hi = range - 1u;
	return v < hi ? v : hi;
}
MIRT_DI uint32_t hash_2d(uint32_t x, uint32_t y) {
	const uint32_t qx = 0x41c64e6du * ((x >> 1u) ^ y);
	const uint32_t qy = 0x41c64e6du * ((y >> 1u) ^ x);
	return 0x41c64e6du * (qx ^ (qy >> 3u));
}

// ---- VectorMath.hpp:581-662 --------------------------------------------------------------------
MIRT_DI float fabs_bits(float v) { return __uint_as_float(__float_as_uint(v) & 0x7fffffffu); }
MIRT_DI float copysign_bits(float v, float s) { return __uint_as_float((__float_as_uint(v) & 0x7fffffffu) | (__float_as_uint(s) & 0x80000000u)); }
MIRT_DI float xor_bits(float a, float b) { return __uint_as_float(__float_as_uint(a) ^ __float_as_uint(b)); }
MIRT_DI float and_bits(float a, float b) { return __uint_as_float(__float_as_uint(a) & __float_as_uint(b)); }

MIRT_DI float fast_asin(float x) {                        // :625-630
	float f = fabs_bits(x);
	f = (f < 1.0f) ? 1.0f - (1.0f - f) : 1.0f;
	f = MIRT_HALF_PI - __builtin_sqrtf(1.0f - f) * (1.5707963267f + f * (-0.213300989f + f * (0.077980478f + f * -0.02164095f)));
	return copysign_bits(f, x);
}
MIRT_DI float fast_atan2(float y, float x) {              // :632-642
	const float a = fabs_bits(x); const float b = fabs_bits(y);
	float lo = min_sel(a, b), hi = max_sel(a, b);
	float k = hi == 0.0f ? 0.0f : lo / hi;
	k = 1.0f - (1.0f - k);
	const float k2 = k * k;
	float r = k * (0.43157974f * k2 + 1.0f) / ((0.05831938f * k2 + 0.76443945f) * k2 + 1.0f);
	if (b > a) r = MIRT_HALF_PI - r;
	if (x < 0.0f) r = MIRT_PI - r;
	return copysign_bits(r, y);
}
MIRT_DI void fast_sincos(float x, float& sine, float& cosine) {   // :644-662
	const float qf = __builtin_rintf(x * MIRT_INV_PI);            // _mm_round_ss nearest-even
	const uint32_t sign_mask = static_cast<uint32_t>(static_cast<int32_t>(qf)) << 31;
	x += qf * (-0.78515625f * 4);
	x += qf * (-0.00024187564849853515625f * 4);
	x += qf * (-3.7747668102383613586e-08f * 4);
	x += qf * (-1.2816720341285448015e-12f * 4);
	x = MIRT_HALF_PI - (MIRT_HALF_PI - x);
	float x2 = x * x;
	x = __uint_as_float(__float_as_uint(x) ^ sign_mask);
	float su = 2.6083159809786593541503e-06f;     float cu = -2.71811842367242206819355e-07f;
	su = su * x2 - 0.0001981069071916863322258f;  cu = (cu * x2 + 2.47990446951007470488548e-05f);
	su = su * x2 + 0.00833307858556509017944336f; cu = (cu * x2 - 0.00138888787478208541870117f);
	su = su * x2 - 0.166666597127914428710938f;   cu = (cu * x2 + 0.0416666641831398010253906f);
	su = x2 * (su * x) + x;                       cu = (cu * x2 - 0.5f); cu = (cu * x2 + 1.0f);
	cu = __uint_as_float(__float_as_uint(cu) ^ sign_mask);
	if (fabs_bits(su) > 1.0f) su = 0.0f;
	if (fabs_bits(cu) > 1.0f) cu = 0.0f;
	sine = su; cosine = cu;
}

// ---- Sampling.hpp ------------------------------------------------------------------------------
MIRT_DI f3 spherical_to_cartesian(float phi_over_2pi, float sin_theta, float cos_theta) {   // :77-84
	float cos_phi, sin_phi; fast_sincos(phi_over_2pi * MIRT_TWO_PI, sin_phi, cos_phi);
	return { sin_theta * cos_phi, sin_theta * sin_phi, cos_theta };
}
MIRT_DI f3 hemisphere(float t, float s) {                                                   // :92-94
	return spherical_to_cartesian(s, __builtin_sqrtf(t), __builtin_sqrtf(max_sel(0.0f, 1.0f - t)));
}
MIRT_DI void orthonormal_basis(f3 n, f3& v2, f3& v3) {                                      // :116-130
	float sign = and_bits(-0.0f, n.z);
	float s = xor_bits(1.0f, sign);
	float z = -1.0f / (s + n.z);
	float s_nx = xor_bits(sign, n.x);
	float ny_z = n.y * z;
	float t = n.x * ny_z;
	v2 = { 1.0f + (s_nx * n.x) * z, xor_bits(sign, t), -s_nx };
	v3 = { t, s + ny_z * n.y, -n.y };
}
MIRT_DI quat tangent_space(f3 N) {                                                          // :150-159
	if (N.z < -1.0f + MIRT_FLT_EPSILON) return quat{ 0.0f, 1.0f, 0.0f, 0.0f };
	float s = __builtin_sqrtf(2.0f * (N.z + 1.0f));
	float invs = 1.0f / s;
	return quat{ -N.y * invs, N.x * invs, 0.0f, s * 0.5f };
}
MIRT_DI f3 to_local(quat T, f3 v) {                                                         // :161-169
	float temp = 2.0f * (v.z * T.w + v.x * T.y - T.x * v.y);
	return { v.x - T.y * temp, v.y + T.x * temp, temp * T.w - v.z };
}
MIRT_DI f3 to_world(quat T, f3 v) {                                                         // :171-179
	float temp = 2.0f * (v.z * T.w - v.x * T.y + T.x * v.y);
	return { v.x + T.y * temp, v.y - T.x * temp, temp * T.w - v.z };
}
MIRT_DI float conePdf(float cosThetaMax) { return MIRT_INV_TWO_PI / max_sel(1e-6f, 1.0f - cosThetaMax); }   // :192-194
MIRT_DI float spherePdf(float radius_sq, float dist_sq) {                                                     // :196-200
	float sinThetaMax2 = radius_sq / dist_sq;
	float cosThetaMax = __builtin_sqrtf(max_sel(0.0f, 1.0f - sinThetaMax2));
	return conePdf(cosThetaMax);
}
MIRT_DI f3 sample_direction_to_sphere(f3 Wc, float sinThetaMax2, float center_dist, float radius2,
                                      float t, float s, float& out_distance, float& out_pdf) {               // :220-239
	float cosThetaMax = __builtin_sqrtf(max_sel(0.0f, 1.0f - sinThetaMax2));
	out_pdf = conePdf(cosThetaMax);
	float cosTheta = 1.0f - t * (1.0f - cosThetaMax);
	float sinTheta = __builtin_sqrtf(sinThetaMax2 * t);
	const bool small = sinThetaMax2 < 0.00068523f;
	float src_blend = small ? sinTheta : cosTheta;
	float invert = __builtin_sqrtf(max_sel(0.0f, 1.0f - src_blend * src_blend));
	cosTheta = small ? invert : cosTheta;
	sinTheta = small ? sinTheta : invert;
	float temp = center_dist * sinTheta;
	out_distance = center_dist * cosTheta - __builtin_sqrtf(max_sel(0.0f, radius2 - temp * temp)) - 1e-5f;
	f3 Ll = spherical_to_cartesian(s, sinTheta, cosTheta);
	f3 wcX, wcY; orthonormal_basis(Wc, wcX, wcY);
	return { wcX.x * Ll.x + wcY.x * Ll.y + Wc.x * Ll.z,
	         wcX.y * Ll.x + wcY.y * Ll.y + Wc.y * Ll.z,
	         wcX.z * Ll.x + wcY.z * Ll.y + Wc.z * Ll.z };
}
// ---- GGX closure, function level (DataStreams.hpp:184-219, Sampling.hpp:85-91,102-104,254-309) --------------
// The reference's path with this closure does not build (`#define BRDF 0`; `gloss_decay_table` of Renderer.hpp:212 is declared
// nowhere) and Closure<GGX>::pdf returns 0 ("TODO"), so these functions are NOT wired into k_shade; they are the defined part of the
// closure, checked bit for bit against the oracle through mirt_debug_math (fn 8, 9).
MIRT_DI float mix_glm(float x, float y, float a) { return x * (1.0f - a) + y * a; }           // glm::mix
MIRT_DI float clamp_std(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
MIRT_DI void disk(float t, float s, float& x, float& y) {                                       // Sampling.hpp:85-91,102-104
	float cos_phi, sin_phi; fast_sincos(s * MIRT_TWO_PI, sin_phi, cos_phi);
	const float rho = __builtin_sqrtf(t);
	x = rho * cos_phi; y = rho * sin_phi;
}
MIRT_DI f3 distribution_visible_normals(f3 Vlocal, float alpha, float u, float v) {             // :254-270
	const f3 V = normalize3(f3{ alpha * Vlocal.x, alpha * Vlocal.y, Vlocal.z });
	float sx, sy; disk(u, v, sx, sy);
	const float t = 1.0f - sx * sx;
	sy = mix_glm(__builtin_sqrtf(t), sy, V.z * 0.5f + 0.5f);
	f3 X, Y; orthonormal_basis(V, X, Y);
	const float k = __builtin_sqrtf(max_sel(0.0f, t - sy * sy));
	const f3 H{ (X.x * sx + Y.x * sy) + V.x * k, (X.y * sx + Y.y * sy) + V.y * k, (X.z * sx + Y.z * sy) + V.z * k };
	return normalize3(f3{ alpha * H.x, alpha * H.y, max_sel(0.0f, H.z) });
}
MIRT_DI float pow5(float x) { float t = x * x; t *= t; return x * t; }                          // :272
MIRT_DI f3 Fresnel(f3 F0, float HdotV) {                                                        // :273-275
	const float a = pow5(clamp_std(1.0f - HdotV, 0.0f, 1.0f));
	return { mix_glm(F0.x, 1.0f, a), mix_glm(F0.y, 1.0f, a), mix_glm(F0.z, 1.0f, a) };
}
MIRT_DI float GGX_D(float alpha2, float NdotH2) { const float temp = (1.0f + (alpha2 - 1.0f) * NdotH2); return alpha2 / (MIRT_PI * temp * temp); }   // :278-281
MIRT_DI float smith_g2_lagarde(float alpha2, float NdotL, float NdotV) {                        // :287-291
	const float a = NdotV * __builtin_sqrtf(alpha2 + NdotL * (NdotL - alpha2 * NdotL));
	const float b = NdotL * __builtin_sqrtf(alpha2 + NdotV * (NdotV - alpha2 * NdotV));
	return 0.5f / (a + b);
}
MIRT_DI f3 microfacet_brdf(f3 F0, float alpha, float NdotV, float NdotL, float NdotH, float HdotV) {   // :293-296
	const float alpha2 = alpha * alpha;
	const f3 F = Fresnel(F0, HdotV);
	const float k = NdotL * GGX_D(max_sel(0.00001f, alpha2), NdotH * NdotH) * smith_g2_lagarde(alpha2, NdotL, NdotV);
	return { F.x * k, F.y * k, F.z * k };
}
MIRT_DI float G1_GGX(float alpha2, float NdotS2) { return 2.0f / (1.0f + __builtin_sqrtf(((alpha2 * (1.0f - NdotS2)) + NdotS2) / NdotS2)); }   // :297-299
MIRT_DI float smith_g2_over_g1(float alpha2, float NdotL, float NdotV) {                        // :301-305
	const float G1V = G1_GGX(alpha2, NdotV * NdotV), G1L = G1_GGX(alpha2, NdotL * NdotL);
	return G1L / (G1V + G1L - G1V * G1L);
}
MIRT_DI f3 vndf_estimator(f3 F0, float alpha, float NdotV, float NdotL, float HdotV) {          // :307-309
	const f3 F = Fresnel(F0, HdotV);
	const float k = smith_g2_over_g1(alpha * alpha, NdotL, NdotV);
	return { F.x * k, F.y * k, F.z * k };
}
MIRT_DI f3 ggx_eval(f3 F0, float alpha, f3 Llocal, f3 Vlocal) {                                 // Closure<GGX>::eval, DataStreams.hpp:189-195
	const float NdotL = max_sel(0.0f, Llocal.z), NdotV = max_sel(0.0f, Vlocal.z);
	const f3 Hn = normalize3(f3{ Llocal.x + Vlocal.x, Llocal.y + Vlocal.y, Llocal.z + Vlocal.z });
	const float NdotH = max_sel(0.0f, Hn.z), HdotV = max_sel(0.0f, dot3(Hn, Vlocal));
	return microfacet_brdf(F0, alpha, NdotV, NdotL, NdotH, HdotV);
}
MIRT_DI void ggx_sample(f3 F0, float alpha, f3 Vlocal, float u0, float u1, f3& dir, f3& estimator) {   // Closure<GGX>::sample, DataStreams.hpp:200-218
	const float NdotV = max_sel(0.0f, Vlocal.z);
	float HdotV;
	if (alpha == 0.0f) { dir = f3{ -Vlocal.x, -Vlocal.y, Vlocal.z }; HdotV = NdotV; }
	else {
		const f3 Hl = distribution_visible_normals(Vlocal, alpha, u0, u1);
		HdotV = dot3(Hl, Vlocal);
		const float k = 2.0f * HdotV;
		dir = f3{ k * Hl.x - Vlocal.x, k * Hl.y - Vlocal.y, k * Hl.z - Vlocal.z };
		HdotV = max_sel(0.0f, HdotV);
	}
	const float NdotL = max_sel(0.0f, dir.z);
	estimator = vndf_estimator(F0, alpha, NdotV, NdotL, HdotV);
}

MIRT_DI float powerHeuristic(float f, float g) { float f2 = f * f; return f2 / max_sel(1e-6f, f2 + g * g); }   // :241-244
MIRT_DI float powerHeuristic_over_f(float f, float g) { return f / max_sel(1e-6f, f * f + g * g); }            // :245-247
MIRT_DI float median3(float a, float b, float c) { return max_sel(min_sel(a, b), min_sel(max_sel(a, b), c)); } // :8-12
MIRT_DI float median5(float a, float b, float c, float d, float e) {                                           // :13-21
	return median3(max_sel(min_sel(a, b), min_sel(c, d)), min_sel(max_sel(a, b), max_sel(c, d)), e);
}

// ---- Color.hpp:47-49,66-73 (VCL Vec8f min/max = _mm256_min/max_ps lane semantics) --------------------
MIRT_DI float aces_fit(float x) { return (x * (x + 0.0245786f) - 0.000090537f) / (x * (0.983729f * x + 0.4329510f) + 0.238081f); }
MIRT_DI float clamp01_vcl(float v) { float m = (0.0f > v) ? 0.0f : v; return (1.0f < m) ? 1.0f : m; }   // min(1, max(0, v))
MIRT_DI void tonemapping(float& r, float& g, float& b) {
	float x = aces_fit(r * 0.59719f + g * 0.35458f + b * 0.04823f);
	float y = aces_fit(r * 0.07600f + g * 0.90834f + b * 0.01566f);
	float z = aces_fit(r * 0.02840f + g * 0.13383f + b * 0.83777f);
	r = clamp01_vcl(x * 1.604750f + y * -0.53108f + z * -0.07367f);
	g = clamp01_vcl(x * -0.10208f + y * 1.10813f + z * -0.00605f);
	b = clamp01_vcl(x * -0.00327f + y * -0.07276f + z * 1.07602f);
}

// ---- Camera.hpp:80-88 -------------------------------------------------------------------------------
struct CameraParams { float pos[3]; float orient[4]; float half_width, half_height, z, exposure; };
MIRT_DI f3 camera_ray_dir(const CameraParams& c, int32_t x, int32_t y, float s0, float s1) {
	f3 v{ static_cast<float>(x) + s0 - c.half_width, static_cast<float>(y) + s1 - c.half_height, c.z };
	f3 qv{ c.orient[0], c.orient[1], c.orient[2] };      // glm operator*(quat, vec3)
	f3 uv = cross3(qv, v);
	f3 uuv = cross3(qv, uv);
	const float w = c.orient[3];
	f3 r{ v.x + ((uv.x * w) + uuv.x) * 2.0f, v.y + ((uv.y * w) + uuv.y) * 2.0f, v.z + ((uv.z * w) + uuv.z) * 2.0f };
	return normalize3(r);
}

} // namespace mirt
