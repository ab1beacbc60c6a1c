// How many waves per SIMD (and independent accumulators per wave) v_mfma_f64_16x16x4_f64 needs to reach
// its rate on gfx950, and what a v_mul_f64 between the multiplications costs.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_occupancy.hip -o /tmp/mfma_occ && /tmp/mfma_occ
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int MULS>
__global__ __launch_bounds__(256) void rate(double *out, int n, double seed)
{
	double4_t c[NACC];
	for (int k = 0; k < NACC; k++) c[k] = double4_t{0, 0, 0, 0};
	double a = threadIdx.x * 1e-3 + seed, b = 1.0 + threadIdx.x * 1e-4;
	for (int i = 0; i < n; i++) {
#pragma unroll
		for (int k = 0; k < NACC; k++) {
			if (MULS && (k & 1)) b = b * 1.0000001;
			c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
		}
	}
	double s = 0;
	for (int k = 0; k < NACC; k++) s += c[k][k & 3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MULS>
static void run(int wgs_per_cu, int lds_bytes)
{
	const int cus = 256, n = 2048;
	double *dout;
	hipMalloc(&dout, (size_t) cus * wgs_per_cu * 256 * 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((rate<NACC, MULS>), dim3(cus * wgs_per_cu), dim3(256), lds_bytes, 0, dout, 16, 0.0);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((rate<NACC, MULS>), dim3(cus * wgs_per_cu), dim3(256), lds_bytes, 0, dout, n, 0.5);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double flops = 2.0 * 1024.0 * NACC * n * (double) cus * wgs_per_cu * 4;
	printf("accumulators %d, v_mul between %d, waves per SIMD %d: %.1f TFLOP/s (%.3f ms)\n", NACC, MULS, wgs_per_cu, flops / (ms * 1e-3) / 1e12, ms);
	hipFree(dout);
}

int main()
{
	// (a workgroup of 256 threads = one wave per SIMD; k workgroups per CU, kept apart from more by dynamic LDS)
	for (int k : {1, 2, 3, 4, 8}) run<4, 0>(k, 160 * 1024 / k - 1024);
	for (int k : {1, 2, 4}) run<8, 0>(k, 160 * 1024 / k - 1024);
	for (int k : {1, 2, 4}) run<8, 1>(k, 160 * 1024 / k - 1024);
	for (int k : {1, 2}) run<2, 0>(k, 160 * 1024 / k - 1024);
	return 0;
}
