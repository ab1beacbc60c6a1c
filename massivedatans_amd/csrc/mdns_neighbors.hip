// RadFriends geometry kernels for gfx950 (MI355X): the all-pairs distance tests that replace
// clustering/cneighbors.c.  Members (the live-point pool) are tiled through LDS; every lane
// owns one candidate (K3/K4) or one pool point (K5/K6) and reads the tile by LDS broadcast.
//
// Integer / bit exactness.  The squared distance is accumulated from 0 over the dimensions in
// ascending order with separate multiply and add (no FMA: this file is compiled with
// -ffp-contract=off and says so again below), which is the arithmetic of cneighbors.c:55-58.
// The reference then tests  sqrt(d) < r  (cneighbors.c:88,109); since the correctly rounded
// square root is monotone that is  d < T  with T = the smallest double whose root is >= r,
// found on the host (mdns::sqrt_threshold), so no device sqrt is needed.  For the radii
// (cneighbors.c:64-71,160-174) the root is taken once, on the host, after the max of the
// min squared distances -- the same number because sqrt is monotone.
#include "mdns_internal.h"

#pragma clang fp contract(off)

namespace mdns {

static constexpr int kBlock = 256;
static constexpr int kTile = 512;          // members per LDS tile
static constexpr int kMaxRegDim = 8;       // dimensions kept in registers
static constexpr int kRounds = 16;         // bootstrap rounds per pass (cneighbors uses 10)

__device__ __forceinline__ double sq_distance(const double *a, const double *b, int ndim)
{
	double acc = 0.0;
	for (int k = 0; k < ndim; k++) {
		const double diff = a[k] - b[k];
		acc = acc + diff * diff;
	}
	return acc;
}

template <int D>
__device__ __forceinline__ double sq_distance_fixed(const double *a, const double (&c)[D])
{
	double acc = 0.0;
#pragma unroll
	for (int k = 0; k < D; k++) {
		const double diff = a[k] - c[k];
		acc = acc + diff * diff;
	}
	return acc;
}

// ---------------------------------------------------------------------------------------
// K3 / K4: how many members lie strictly within the radius of each candidate
// ---------------------------------------------------------------------------------------
// grid.x tiles the candidates, grid.y splits the members; partial counts are combined with
// integer atomics (exact whatever the order).  D == 0: runtime ndim (slow generic path).
template <int D>
__global__ __launch_bounds__(kBlock) void k_count_within(
    const double *__restrict__ members, int K, int ndim, double thresh_sq,
    const double *__restrict__ cands, int M, int *__restrict__ counts, int kchunk)
{
	extern __shared__ double tile[];                    // [kTile][ndim]
	const int j = blockIdx.x * kBlock + threadIdx.x;
	const int jj = j < M ? j : M - 1;
	const int kbeg = blockIdx.y * kchunk;
	const int kend = min(K, kbeg + kchunk);

	double c[D > 0 ? D : 1];
	if (D > 0) {
#pragma unroll
		for (int k = 0; k < D; k++) c[k] = cands[(size_t) jj * D + k];
	}
	int hits = 0;
	for (int t0 = kbeg; t0 < kend; t0 += kTile) {
		const int n = min(kTile, kend - t0);
		__syncthreads();
		for (int e = threadIdx.x; e < n * ndim; e += kBlock) tile[e] = members[(size_t) t0 * ndim + e];
		__syncthreads();
		if (D > 0) {
			for (int i = 0; i < n; i++)
				hits += sq_distance_fixed<(D > 0 ? D : 1)>(tile + i * D, c) < thresh_sq ? 1 : 0;
		} else {
			const double *cj = cands + (size_t) jj * ndim;
			for (int i = 0; i < n; i++)
				hits += sq_distance(tile + i * ndim, cj, ndim) < thresh_sq ? 1 : 0;
		}
	}
	if (j < M && hits) atomicAdd(counts + j, hits);
}

// ---------------------------------------------------------------------------------------
// K5 / K6: nearest "chosen" pool point of every "left-out" pool point, max over the left-out
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
	return v;
}

// non-negative doubles order like their bit patterns
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v)
{
	atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long) __double_as_longlong(v));
}

// NN == false (K6, cneighbors.c:137-168): rounds [b0, b0+nb) of the bootstrap; a point is
//   "chosen" in round b when chosen[i*nboot + b] != 0; left-out points with index >= 1
//   contribute (the reference's max loop starts at 1, :162).
// NN == true  (K5, cneighbors.c:47-71): one round, everybody chosen, self excluded, every
//   point contributes.
template <int D, bool NN>
__global__ __launch_bounds__(kBlock) void k_nearest_chosen(
    const double *__restrict__ members, int K, int ndim, const double *__restrict__ chosen,
    int nboot, int b0, int nb, double *__restrict__ round_sq)
{
	extern __shared__ double smem[];
	double *tile = smem;                                          // [kTile][ndim]
	unsigned *tmask = reinterpret_cast<unsigned *>(smem + (size_t) kTile * ndim);   // [kTile]

	const int i = blockIdx.x * kBlock + threadIdx.x;
	const int ii = i < K ? i : K - 1;
	double c[D > 0 ? D : 1];
	if (D > 0) {
#pragma unroll
		for (int k = 0; k < D; k++) c[k] = members[(size_t) ii * D + k];
	}
	unsigned mymask = 0;
	if (!NN)
		for (int b = 0; b < nb; b++) mymask |= (chosen[(size_t) ii * nboot + b0 + b] != 0.0 ? 1u : 0u) << b;

	double nearest[kRounds];
#pragma unroll
	for (int b = 0; b < kRounds; b++) nearest[b] = 1e300;         // cneighbors.c:51,148

	for (int t0 = 0; t0 < K; t0 += kTile) {
		const int n = min(kTile, K - t0);
		__syncthreads();
		for (int e = threadIdx.x; e < n * ndim; e += kBlock) tile[e] = members[(size_t) t0 * ndim + e];
		for (int e = threadIdx.x; e < n; e += kBlock) {
			unsigned mk = 0;
			if (NN) mk = 1u;
			else for (int b = 0; b < nb; b++) mk |= (chosen[(size_t) (t0 + e) * nboot + b0 + b] != 0.0 ? 1u : 0u) << b;
			tmask[e] = mk;
		}
		__syncthreads();
		for (int jn = 0; jn < n; jn++) {
			const unsigned mk = tmask[jn];                        // wave-uniform
			if (mk == 0) continue;
			double d;
			if (D > 0) d = sq_distance_fixed<(D > 0 ? D : 1)>(tile + jn * D, c);
			else d = sq_distance(members + (size_t) ii * ndim, tile + jn * ndim, ndim);
			if (NN) {
				if (t0 + jn != ii && d < nearest[0]) nearest[0] = d;
			} else {
#pragma unroll
				for (int b = 0; b < kRounds; b++)
					if (((mk >> b) & 1u) && d < nearest[b]) nearest[b] = d;
			}
		}
	}
	// max over the contributing points of this wave, then one atomic per wave and round
#pragma unroll
	for (int b = 0; b < kRounds; b++) {
		if (b < nb) {                                             // nb is uniform over the grid
			const bool contributes = i < K && (NN ? true : (i >= 1 && !((mymask >> b) & 1u)));
			const double v = wave_max(contributes ? nearest[b] : 0.0);
			if ((threadIdx.x & 63) == 0 && v > 0.0) atomic_max_nonneg(round_sq + b0 + b, v);
		}
	}
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static bool launched(const char *name)
{
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) return true;
	set_error("launch of %s failed: %s", name, hipGetErrorString(e));
	return false;
}

bool launch_count_within(const double *d_members, int K, int ndim, double thresh_sq,
                         const double *d_cands, int M, int *d_counts)
{
	Context *c = ctx();
	const int gx = (M + kBlock - 1) / kBlock;
	// split the members until the grid covers the chip about twice (tiles stay whole)
	int want = (2 * c->num_cus + gx - 1) / gx;
	int max_split = (K + kTile - 1) / kTile;
	int gy = want < max_split ? want : max_split;
	if (gy < 1) gy = 1;
	if (gy > 65535) gy = 65535;
	int kchunk = (K + gy - 1) / gy;
	kchunk = ((kchunk + kTile - 1) / kTile) * kTile;
	gy = (K + kchunk - 1) / kchunk;
	const size_t lds = (size_t) kTile * ndim * sizeof(double);
	if (lds > 64 * 1024) { set_error("ndim=%d too large for the member tile", ndim); return false; }
	dim3 grid(gx, gy);
	ProfileScope prof(2);
#define COUNT_LAUNCH(D) hipLaunchKernelGGL((k_count_within<D>), grid, dim3(kBlock), lds, c->stream, \
	d_members, K, ndim, thresh_sq, d_cands, M, d_counts, kchunk)
	switch (ndim <= kMaxRegDim ? ndim : 0) {
	case 1: COUNT_LAUNCH(1); break;
	case 2: COUNT_LAUNCH(2); break;
	case 3: COUNT_LAUNCH(3); break;
	case 4: COUNT_LAUNCH(4); break;
	case 5: COUNT_LAUNCH(5); break;
	case 6: COUNT_LAUNCH(6); break;
	case 7: COUNT_LAUNCH(7); break;
	case 8: COUNT_LAUNCH(8); break;
	default: COUNT_LAUNCH(0); break;
	}
#undef COUNT_LAUNCH
	return launched("k_count_within");
}

template <bool NN>
static bool launch_nearest(const double *d_members, int K, int ndim, const double *d_chosen,
                           int nboot, double *d_round_sq)
{
	Context *c = ctx();
	const size_t lds = (size_t) kTile * ndim * sizeof(double) + kTile * sizeof(unsigned);
	if (lds > 64 * 1024) { set_error("ndim=%d too large for the member tile", ndim); return false; }
	dim3 grid((K + kBlock - 1) / kBlock);
	for (int b0 = 0; b0 < nboot; b0 += kRounds) {
		const int nb = nboot - b0 < kRounds ? nboot - b0 : kRounds;
		ProfileScope prof(3);
#define NEAR_LAUNCH(D) hipLaunchKernelGGL((k_nearest_chosen<D, NN>), grid, dim3(kBlock), lds, c->stream, \
	d_members, K, ndim, d_chosen, nboot, b0, nb, d_round_sq)
		switch (ndim <= kMaxRegDim ? ndim : 0) {
		case 1: NEAR_LAUNCH(1); break;
		case 2: NEAR_LAUNCH(2); break;
		case 3: NEAR_LAUNCH(3); break;
		case 4: NEAR_LAUNCH(4); break;
		case 5: NEAR_LAUNCH(5); break;
		case 6: NEAR_LAUNCH(6); break;
		case 7: NEAR_LAUNCH(7); break;
		case 8: NEAR_LAUNCH(8); break;
		default: NEAR_LAUNCH(0); break;
		}
#undef NEAR_LAUNCH
		if (!launched("k_nearest_chosen")) return false;
	}
	return true;
}

bool launch_bootstrap(const double *d_members, int K, int ndim, const double *d_chosen,
                      int nbootstraps, double *d_round_sq)
{
	return launch_nearest<false>(d_members, K, ndim, d_chosen, nbootstraps, d_round_sq);
}

bool launch_nn_maxsq(const double *d_members, int K, int ndim, double *d_out)
{
	return launch_nearest<true>(d_members, K, ndim, nullptr, 1, d_out);
}

}  // namespace mdns
