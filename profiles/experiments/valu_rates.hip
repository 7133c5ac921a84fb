// Micro-benchmark (round 2): issue rate of the VALU instructions k_trace's traversal step is made of, 8 waves per SIMD, all CUs busy.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o valu_rates valu_rates.hip && ./valu_rates
// Each kernel runs ITER x 32 independent instances of one instruction per wave; cycles per wave-instruction per SIMD =
// time x clock x 1024 SIMDs / (waves x ITER x 32).  (Clock taken as 2.4 GHz; the ratio between rows is what matters.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 4096;
#define R8(x) x x x x x x x x
#define BODY(INSTR) \
	float a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = a0 * 0.5f, c = a0 * 0.25f; \
	for (int i = 0; i < ITER; i++) { R8(INSTR(a0) INSTR(a1) INSTR(a2) INSTR(a3)) } \
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
#define K(name, INSTR) __global__ __launch_bounds__(1024, 8) void name(const float* in, float* out) { BODY(INSTR) }
#define I_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_FMAMIX(x) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(b), "v"(c));
#define I_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define I_CMPCND(x) asm volatile("v_cmp_le_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define I_CMP64(x) asm volatile("v_cmp_le_f32 s[20:21], %1, %0\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(x) : "v"(b), "v"(c) : "s20", "s21");
#define I_SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define I_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define I_VALU_SALU(x) asm volatile("v_add_f32 %0, %0, %1\n s_and_b64 s[20:21], s[20:21], exec" : "+v"(x) : "v"(b) : "s20", "s21");
#define I_VALU_2SALU(x) asm volatile("v_add_f32 %0, %0, %1\n s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[20:21], exec" : "+v"(x) : "v"(b) : "s20", "s21", "s22", "s23");
K(k_fma, I_FMA) K(k_fmamix, I_FMAMIX) K(k_med3, I_MED3) K(k_max3, I_MAX3) K(k_max, I_MAX) K(k_add, I_ADD) K(k_cmpcnd, I_CMPCND) K(k_cmp64, I_CMP64)
K(k_sqrt, I_SQRT) K(k_lshladd, I_LSHLADD) K(k_valu_salu, I_VALU_SALU) K(k_valu_2salu, I_VALU_2SALU)
int main() {
	float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 512 * 1024 * 4); hipMemset(in, 0, 4096);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	auto run = [&](const char* name, auto kern, int per_instance) {
		hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, in, out); hipDeviceSynchronize();
		hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, in, out); hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		const double waves = 512.0 * 16, inst = waves * ITER * 32.0 * per_instance;
		printf("%-34s %7.3f ms  %5.2f cycles per wave-instruction per SIMD (%d instr per instance)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / inst, per_instance);
	};
	run("v_fma_f32", k_fma, 1); run("v_fma_mix_f32 (f16 operand)", k_fmamix, 1); run("v_med3_f32", k_med3, 1); run("v_max3_f32", k_max3, 1);
	run("v_max_f32", k_max, 1); run("v_add_f32", k_add, 1); run("v_cmp(vcc) + v_cndmask", k_cmpcnd, 2); run("v_cmp(sgpr pair) + v_cndmask e64", k_cmp64, 2);
	run("v_sqrt_f32", k_sqrt, 1); run("v_lshl_add_u32", k_lshladd, 1); run("v_add_f32 + 1 s_and_b64", k_valu_salu, 2); run("v_add_f32 + 2 SALU", k_valu_2salu, 3);
	return 0;
}
