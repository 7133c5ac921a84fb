// Micro-benchmark (round 2): how fast can a k_shade-shaped grid (3 x 512-thread workgroups per CU, grid-stride) write a ray stream,
// as a function of the stream's layout?  And what does one returning same-address atomicAdd per 512 rays cost?
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int P = 24;   // dword fields per ray written by k_shade (next stream 14 + shadow record ~10)

__global__ __launch_bounds__(512) void planes(float* out, size_t n, size_t pitch) {          // field-major planes, one dword per lane and plane
	for (size_t i = blockIdx.x * 512ull + threadIdx.x; i < n; i += gridDim.x * 512ull) {
		const float v = static_cast<float>(i);
#pragma unroll
		for (int p = 0; p < P; p++) out[p * pitch + i] = v + p;
	}
}
__global__ __launch_bounds__(512) void tiles64(float* out, size_t n) {                        // [ray / 64][field][ray % 64]: a wave's 24 x 256 B are one 6 KB burst
	for (size_t i = blockIdx.x * 512ull + threadIdx.x; i < n; i += gridDim.x * 512ull) {
		const float v = static_cast<float>(i);
		float* t = out + (i >> 6) * (P * 64) + (i & 63);
#pragma unroll
		for (int p = 0; p < P; p++) t[p * 64] = v + p;
	}
}
__global__ __launch_bounds__(512) void planes4(float4* out, size_t n, size_t pitch) {         // field-major planes of float4 (16 B per lane)
	for (size_t i = blockIdx.x * 512ull + threadIdx.x; i < n; i += gridDim.x * 512ull) {
		const float v = static_cast<float>(i);
#pragma unroll
		for (int p = 0; p < P / 4; p++) out[p * pitch + i] = make_float4(v, v + 1, v + 2, v + p);
	}
}
__global__ __launch_bounds__(512) void tiles64x4(float4* out, size_t n) {                     // [ray / 64][field4][ray % 64] float4
	for (size_t i = blockIdx.x * 512ull + threadIdx.x; i < n; i += gridDim.x * 512ull) {
		const float v = static_cast<float>(i);
		float4* t = out + (i >> 6) * (P / 4 * 64) + (i & 63);
#pragma unroll
		for (int p = 0; p < P / 4; p++) t[p * 64] = make_float4(v, v + 1, v + 2, v + p);
	}
}
// one returning atomicAdd per block iteration, spread over `spread` counters 128 B apart; every lane then stores one dword at base + tid
__global__ __launch_bounds__(512) void append(uint32_t* counters, int spread, float* out, size_t n) {
	__shared__ uint32_t base;
	for (size_t i = blockIdx.x * 512ull; i < n; i += gridDim.x * 512ull) {
		if (threadIdx.x == 0) base = atomicAdd(&counters[((i >> 9) % spread) * 32], 512u);
		__syncthreads();
		out[(static_cast<size_t>(base) + threadIdx.x) % n] = 1.0f;
		__syncthreads();
	}
}
int main() {
	const size_t n = 64ull << 20;
	float* buf; CHECK(hipMalloc(&buf, n * P * 4 + 4096));
	uint32_t* ctr; CHECK(hipMalloc(&ctr, 64 * 128));
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	const int grid = prop.multiProcessorCount * 3;
	auto time = [&](const char* name, auto launch, double bytes) {
		launch(); hipDeviceSynchronize();
		hipEventRecord(a); for (int r = 0; r < 3; r++) launch(); hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
		printf("%-28s %8.3f ms  %7.2f TB/s\n", name, ms, bytes / ms / 1e9);
	};
	const double bytes = double(n) * P * 4;
	time("planes (dword, 24 planes)", [&] { hipLaunchKernelGGL(planes, dim3(grid), dim3(512), 0, 0, buf, n, n); }, bytes);
	time("tiles of 64 rays (dword)", [&] { hipLaunchKernelGGL(tiles64, dim3(grid), dim3(512), 0, 0, buf, n); }, bytes);
	time("planes (float4, 6 planes)", [&] { hipLaunchKernelGGL(planes4, dim3(grid), dim3(512), 0, 0, (float4*)buf, n, n); }, bytes);
	time("tiles of 64 rays (float4)", [&] { hipLaunchKernelGGL(tiles64x4, dim3(grid), dim3(512), 0, 0, (float4*)buf, n); }, bytes);
	for (int spread : { 1, 2, 8, 32 }) {
		hipMemset(ctr, 0, 64 * 128);
		char name[64]; snprintf(name, sizeof name, "append, %d counter(s)", spread);
		time(name, [&] { hipLaunchKernelGGL(append, dim3(grid), dim3(512), 0, 0, ctr, spread, buf, n); }, double(n) * 4);
	}
	printf("(append: %zu atomics per launch)\n", n / 512);
	return 0;
}
