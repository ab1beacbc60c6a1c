// Layout probe for v_mfma_f64_16x16x4_f64 on gfx950: D[16x16] += A[16x4] B[4x16].
// Hypothesis (CDNA3 ISA guide, matrix instruction calculator): lane l holds A[i = l % 16][k = l / 16],
// B[k = l / 16][j = l % 16], and D[i = 4 * v + l / 16][j = l % 16] in its v-th result register
// (measured: the other guess, i = 4 * (l / 16) + v -- the f32 16x16x4 layout -- matches only where the two coincide).
//   hipcc --offload-arch=gfx950 tools/probes/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <chrono>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void probe(const double *A, const double *B, double *D)
{
	const int l = threadIdx.x;
	const double a = A[(l % 16) * 4 + l / 16];          // A[i][k], row-major 16x4
	const double b = B[(l / 16) * 16 + l % 16];         // B[k][j], row-major 4x16
	double4_t c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	for (int v = 0; v < 4; v++) D[(4 * v + l / 16) * 16 + l % 16] = c[v];
}

// throughput: each wave issues N dependent-free MFMAs on 4 accumulators
__global__ void rate(double *out, int n)
{
	double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
	for (int i = 0; i < n; i++) {
		c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
		c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
		c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
		c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main()
{
	double hA[64], hB[64], hD[256], ref[256];
	for (int i = 0; i < 64; i++) { hA[i] = sin(i * 1.7) + 0.01 * i; hB[i] = cos(i * 0.9) - 0.02 * i; }
	for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s = fma(hA[i * 4 + k], hB[k * 16 + j], s); ref[i * 16 + j] = s; }
	double *dA, *dB, *dD;
	hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
	hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
	double worst = 0; int exact = 0;
	for (int e = 0; e < 256; e++) { worst = fmax(worst, fabs(hD[e] - ref[e])); exact += hD[e] == ref[e]; }
	printf("layout hypothesis: max abs difference %.3e, %d / 256 entries bit-equal to a k-ascending fma chain\n", worst, exact);
	// rate
	const int blocks = 256 * 8, threads = 256, n = 4096;
	double *dout; hipMalloc(&dout, (size_t) blocks * threads * 8);
	hipLaunchKernelGGL(rate, dim3(blocks), dim3(threads), 0, 0, dout, 16);
	hipDeviceSynchronize();
	auto t0 = std::chrono::steady_clock::now();
	hipLaunchKernelGGL(rate, dim3(blocks), dim3(threads), 0, 0, dout, n);
	hipDeviceSynchronize();
	const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	const double flops = 2.0 * 1024.0 * 4.0 * n * (double) blocks * (threads / 64);
	printf("v_mfma_f64_16x16x4_f64: %.1f TFLOP/s (%d waves, %d x 4 MFMAs each, %.3f ms)\n", flops / s / 1e12, blocks * threads / 64, n, s * 1e3);
	return 0;
}
