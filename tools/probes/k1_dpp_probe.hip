// Probe (not part of the product): does `v_fmac_f64_dpp ... row_newbcast:n` issue at the plain
// v_fmac_f64 rate on gfx950, and does a lane-per-spectrum cross-term kernel fed that way
// (templates in VGPRs, 16 values per register pair, no scalar loads) beat the SGPR-fed
// residual-form kernel?   hipcc -O3 --offload-arch=gfx950 k1_dpp_probe.hip -o k1_dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int N>
__device__ __forceinline__ void fmac_bcast(double &acc, double m, double y)
{
#define CASE(n) if constexpr (N == n) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "v"(y));
	CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7)
	CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
}

template <int C, int B>
struct Unroll {
	template <int c, int b>
	static __device__ __forceinline__ void step(double (&acc)[B], const double (&mv)[C * B / 16], const double (&y)[C])
	{
		fmac_bcast<(c * B + b) % 16>(acc[b], mv[(c * B + b) / 16], y[c]);
		if constexpr (b + 1 < B) step<c, b + 1>(acc, mv, y);
		else if constexpr (c + 1 < C) step<c + 1, 0>(acc, mv, y);
	}
};

// YT [tiles][nxp][64], MT [nbt][nxp][8]; out C[b][i] = sum_j m[b][j] y[i][j]
template <bool DPP>
__global__ __launch_bounds__(256) void k_cross(const double *__restrict__ YT, int nxp, const double *__restrict__ MT,
                                               int B, int M, int ntiles, int nbt, double *__restrict__ out)
{
	constexpr int BT = 8, CH = 8;
	const int lane = threadIdx.x & 63;
	const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int tile = wave % ntiles, bt = wave / ntiles;
	if (bt >= nbt) return;
	const double *yp = YT + ((size_t) tile * nxp << 6) + lane;
	const double *mp = MT + (size_t) bt * nxp * BT;
	double acc[BT];
#pragma unroll
	for (int b = 0; b < BT; b++) acc[b] = 0.0;
	double ya[CH], yb[CH], ma[4], mb[4];
#pragma unroll
	for (int c = 0; c < CH; c++) ya[c] = yp[c * 64];
#pragma unroll
	for (int r = 0; r < 4; r++) ma[r] = mp[r * 16 + (lane & 15)];
	auto stage = [&](const double (&cy)[CH], double (&ny)[CH], const double (&cm)[4], double (&nm)[4], bool last) {
		yp += last ? 0 : CH * 64;
		mp += last ? 0 : CH * BT;
#pragma unroll
		for (int c = 0; c < CH; c++) ny[c] = yp[c * 64];
		if (DPP) {
#pragma unroll
			for (int r = 0; r < 4; r++) nm[r] = mp[r * 16 + (lane & 15)];
			Unroll<CH, BT>::template step<0, 0>(acc, cm, cy);
		} else {
			const double *sp = mp - (last ? 0 : CH * BT);
#pragma unroll
			for (int c = 0; c < CH; c++)
#pragma unroll
				for (int b = 0; b < BT; b++) acc[b] = fma(sp[c * BT + b], cy[c], acc[b]);
		}
	};
	int st = nxp / CH;
#pragma unroll 1
	for (; st >= 2; st -= 2) {
		stage(ya, yb, ma, mb, false);
		stage(yb, ya, mb, ma, st == 2);
	}
	if (st == 1) stage(ya, yb, ma, mb, true);
	const int k = tile * 64 + lane;
	if (k < M) {
#pragma unroll
		for (int b = 0; b < BT; b++)
			if (bt * BT + b < B) out[(size_t) (bt * BT + b) * M + k] = acc[b];
	}
}

int main(int argc, char **argv)
{
	const int M = argc > 1 ? atoi(argv[1]) : 10000, B = argc > 2 ? atoi(argv[2]) : 256, nx = 200;
	const int nxp = (nx + 7) / 8 * 8, ntiles = (M + 63) / 64, nbt = (B + 7) / 8;
	std::vector<double> y((size_t) M * nx), m((size_t) B * nx);
	srand(1);
	for (auto &v : y) v = (rand() / (double) RAND_MAX - 0.5) * 0.1;
	for (auto &v : m) v = rand() / (double) RAND_MAX;
	std::vector<double> YT((size_t) ntiles * nxp * 64, 0.0), MT((size_t) nbt * nxp * 8, 0.0);
	for (int i = 0; i < M; i++) for (int j = 0; j < nx; j++) YT[((size_t) (i / 64) * nxp + j) * 64 + i % 64] = y[(size_t) i * nx + j];
	for (int b = 0; b < B; b++) for (int j = 0; j < nx; j++) MT[((size_t) (b / 8) * nxp + j) * 8 + b % 8] = m[(size_t) b * nx + j];
	double *dY, *dM, *dO;
	CHECK(hipMalloc(&dY, YT.size() * 8)); CHECK(hipMalloc(&dM, MT.size() * 8)); CHECK(hipMalloc(&dO, (size_t) B * M * 8));
	CHECK(hipMemcpy(dY, YT.data(), YT.size() * 8, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(dM, MT.data(), MT.size() * 8, hipMemcpyHostToDevice));
	const int blocks = (ntiles * nbt + 3) / 4;
	std::vector<double> out((size_t) B * M);
	for (int variant = 0; variant < 2; variant++) {
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		CHECK(hipMemset(dO, 0, (size_t) B * M * 8));
		for (int rep = 0; rep < 25; rep++) {
			if (rep == 5) CHECK(hipEventRecord(e0));
			if (variant == 0) hipLaunchKernelGGL((k_cross<true>), dim3(blocks), dim3(256), 0, 0, dY, nxp, dM, B, M, ntiles, nbt, dO);
			else hipLaunchKernelGGL((k_cross<false>), dim3(blocks), dim3(256), 0, 0, dY, nxp, dM, B, M, ntiles, nbt, dO);
		}
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		CHECK(hipMemcpy(out.data(), dO, out.size() * 8, hipMemcpyDeviceToHost));
		double worst = 0;
		for (int t = 0; t < 2000; t++) {
			const int b = rand() % B, i = rand() % M;
			double ref = 0;
			for (int j = 0; j < nx; j++) ref = fma(m[(size_t) b * nx + j], y[(size_t) i * nx + j], ref);
			const double err = fabs(out[(size_t) b * M + i] - ref);
			if (err > worst) worst = err;
		}
		const double us = 1e3 * ms / 20;
		printf("%s: %.1f us per launch, %.1f TFLOP/s (2 flop per fma), max abs err vs host fma chain %.3g\n",
		       variant == 0 ? "dpp row_newbcast" : "sgpr operand    ", us, 2.0 * nx * B * M / us / 1e6, worst);
	}
	return 0;
}
