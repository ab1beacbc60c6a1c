// What feeds v_mfma_f64_16x16x4_f64 matters: accumulators in ArchVGPRs or AccVGPRs, the same or changing A / B registers.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_operands.hip -o /tmp/mfma_ops && /tmp/mfma_ops
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

// NOPS operand pairs in rotation, NACC accumulators; AGPR: accumulators pinned to AccVGPRs ("a" constraint)
template <int NACC, int NOPS, bool AGPR>
__global__ void rate(double *out, int n, double seed)
{
	double4_t c[NACC];
	for (int k = 0; k < NACC; k++) c[k] = double4_t{0, 0, 0, 0};
	double a[NOPS], b[NOPS];
	for (int o = 0; o < NOPS; o++) { a[o] = threadIdx.x * 1e-3 + seed + o; b[o] = 1.0 + threadIdx.x * 1e-4 + o * seed; }
	for (int i = 0; i < n; i++) {
#pragma unroll
		for (int o = 0; o < NOPS; o++)
#pragma unroll
			for (int k = 0; k < NACC; k++) {
				if (AGPR) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c[k]) : "v"(a[o]), "v"(b[o]));
				else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a[o]), "v"(b[o]));
			}
	}
	double s = 0;
	for (int k = 0; k < NACC; k++) s += c[k][k & 3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int NOPS, bool AGPR>
static void run(int wgs_per_cu)
{
	const int cus = 256, n = 4096 / NOPS;
	double *dout;
	(void) hipMalloc(&dout, (size_t) cus * wgs_per_cu * 256 * 8);
	hipEvent_t e0, e1;
	(void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
	const int lds = 160 * 1024 / wgs_per_cu - 1024;
	hipLaunchKernelGGL((rate<NACC, NOPS, AGPR>), dim3(cus * wgs_per_cu), dim3(256), lds, 0, dout, 16, 0.0);
	(void) hipEventRecord(e0, 0);
	hipLaunchKernelGGL((rate<NACC, NOPS, AGPR>), dim3(cus * wgs_per_cu), dim3(256), lds, 0, dout, n, 0.5);
	(void) hipEventRecord(e1, 0);
	(void) hipEventSynchronize(e1);
	float ms = 0;
	(void) hipEventElapsedTime(&ms, e0, e1);
	const double flops = 2.0 * 1024.0 * NACC * NOPS * n * (double) cus * wgs_per_cu * 4;
	printf("accumulators %d in %s, %d operand pairs in rotation, %d waves per SIMD: %.1f TFLOP/s (%.3f ms)\n", NACC, AGPR ? "AccVGPRs" : "ArchVGPRs",
	       NOPS, wgs_per_cu, flops / (ms * 1e-3) / 1e12, ms);
	(void) hipFree(dout);
}

int main()
{
	for (int k : {2, 8}) { run<4, 1, false>(k); run<4, 1, true>(k); }
	for (int k : {2, 8}) { run<8, 1, false>(k); run<8, 1, true>(k); }
	for (int k : {2}) { run<8, 4, false>(k); run<8, 4, true>(k); run<8, 8, false>(k); }
	return 0;
}
