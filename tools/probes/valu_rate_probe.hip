// Issue rate of the vector instructions K6's inner loop is made of, gfx950: cycles per wave-instruction
// with 4 and 8 waves per SIMD, independent chains.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

#define CHAIN8(OP) \
	asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
	             OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
	             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))

template <int WHICH> __global__ void rate(double *out, int n)
{
	double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	double b = 1.0 + 1e-9 * threadIdx.x;
	for (int i = 0; i < n; i++) {
		if (WHICH == 0) CHAIN8("v_min_f64");
		if (WHICH == 1) CHAIN8("v_max_f64");
		if (WHICH == 2) CHAIN8("v_add_f64");
		if (WHICH == 3) CHAIN8("v_mul_f64");
		if (WHICH == 4) asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\n"
		                             "v_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8\n"
		                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
		if (WHICH == 5) {
			unsigned x0 = i, x1 = i + 1, x2 = i + 2, x3 = i + 3, x4 = i + 4, x5 = i + 5, x6 = i + 6, x7 = i + 7, y = threadIdx.x;
			asm volatile("v_or_b32 %0, %0, %8\nv_or_b32 %1, %1, %8\nv_or_b32 %2, %2, %8\nv_or_b32 %3, %3, %8\n"
			             "v_or_b32 %4, %4, %8\nv_or_b32 %5, %5, %8\nv_or_b32 %6, %6, %8\nv_or_b32 %7, %7, %8\n"
			             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
			a0 += x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345u;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int WHICH> static void run(const char *name, double *dout, int waves_per_simd)
{
	const int blocks = 256 * waves_per_simd, threads = 256, n = 20000;       // 4 waves per workgroup: one per SIMD
	hipLaunchKernelGGL(rate<WHICH>, dim3(blocks), dim3(threads), 0, 0, dout, 10);
	hipDeviceSynchronize();
	auto t0 = std::chrono::steady_clock::now();
	hipLaunchKernelGGL(rate<WHICH>, dim3(blocks), dim3(threads), 0, 0, dout, n);
	hipDeviceSynchronize();
	const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	const double per_simd = 8.0 * n * waves_per_simd;                          // wave-instructions per SIMD
	printf("%-10s %d waves/SIMD: %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, waves_per_simd, s / per_simd * 1e9, s / per_simd * 2.4e9);
}

int main()
{
	double *dout; hipMalloc(&dout, (size_t) 256 * 8 * 256 * 8);
	for (int w = 4; w <= 8; w += 4) {
		run<0>("v_min_f64", dout, w); run<1>("v_max_f64", dout, w); run<2>("v_add_f64", dout, w);
		run<3>("v_mul_f64", dout, w); run<4>("v_fma_f64", dout, w); run<5>("v_or_b32", dout, w);
	}
	return 0;
}
