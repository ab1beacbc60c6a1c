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

static constexpr int kBlock = 256;          // 64 points (one per lane) x 4 waves splitting the members
static constexpr int kMaxTile = 512;        // members per LDS tile (fewer when ndim is large)
static constexpr int kMaxRegDim = 8;        // dimensions kept in registers
static constexpr int kRounds = 16;          // bootstrap rounds per pass (cneighbors uses 10)
static constexpr size_t kLdsBudget = 60 * 1024;

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
// A workgroup owns 64 candidates (one per lane); its four waves each scan a quarter of every
// member tile and meet in LDS.  grid.x tiles the candidates, grid.y splits the members between
// workgroups when the pool is large (partial counts then combine with integer atomics into a
// zeroed buffer -- exact whatever the order).  D == 0: runtime ndim (generic path).
template <int D>
__global__ __launch_bounds__(kBlock) void k_count_within(
    const double *__restrict__ members, int K, int ndim, double thresh_sq,
    const double *__restrict__ cands, int M, int *__restrict__ counts, int kchunk, int tile_n,
    int accumulate)
{
	extern __shared__ double smem[];
	double *tile = smem;                                                  // [tile_n][ndim]
	int *part = reinterpret_cast<int *>(smem + (size_t) tile_n * ndim);   // [4][64]
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int j = blockIdx.x * 64 + lane;
	const int jj = j < M ? j : M - 1;
	const int kbeg = blockIdx.y * kchunk;
	const int kend = min(K, kbeg + kchunk);

	double c[D > 0 ? D : 1];
	if (D > 0) {
#pragma unroll
		for (int k = 0; k < D; k++) c[k] = cands[(size_t) jj * D + k];
	}
	int hits = 0;
	for (int t0 = kbeg; t0 < kend; t0 += tile_n) {
		const int n = min(tile_n, kend - t0);
		__syncthreads();
		for (int e = threadIdx.x; e < n * ndim; e += kBlock) tile[e] = members[(size_t) t0 * ndim + e];
		__syncthreads();
		const int q = (n + 3) / 4;
		const int ibeg = wv * q, iend = min(n, ibeg + q);
		if (D > 0) {
			for (int i = ibeg; i < iend; i++)
				hits += sq_distance_fixed<(D > 0 ? D : 1)>(tile + i * D, c) < thresh_sq ? 1 : 0;
		} else {
			const double *cj = cands + (size_t) jj * ndim;
			for (int i = ibeg; i < iend; i++)
				hits += sq_distance(tile + i * ndim, cj, ndim) < thresh_sq ? 1 : 0;
		}
	}
	part[wv * 64 + lane] = hits;
	__syncthreads();
	if (wv == 0 && j < M) {
		const int total = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
		if (accumulate) { if (total) atomicAdd(counts + j, total); }
		else counts[j] = total;
	}
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

// chosen f64[K][nboot] (cneighbors.c:146 tests != 0) -> one bit per round of the window
// [b0, b0+nb) for every pool point
__global__ void k_pack_chosen(const double *__restrict__ chosen, int K, int nboot, int b0, int nb,
                              unsigned *__restrict__ mask)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= K) return;
	const double *row = chosen + (size_t) i * nboot + b0;
	unsigned m = 0;
#pragma unroll
	for (int b = 0; b < kRounds; b++)
		if (b < nb) m |= (row[b] != 0.0 ? 1u : 0u) << b;
	mask[i] = m;
}

// NN == false (K6, cneighbors.c:137-168): rounds of the window packed in `mask`; a left-out
//   point with index >= 1 contributes its nearest chosen point (the reference's max loop starts
//   at 1, :162).
// NN == true  (K5, cneighbors.c:47-71): one round, everybody chosen, self excluded, every
//   point contributes.
// A workgroup owns 64 pool points (one per lane); its four waves each scan a quarter of every
// member tile, so a pool of K points runs on 4*ceil(K/64) waves.  The per-round minima of the
// four waves meet in LDS (min is exact, so the split cannot change the result).
template <int D, bool NN>
__global__ __launch_bounds__(kBlock) void k_nearest_chosen(
    const double *__restrict__ members, int K, int ndim, const unsigned *__restrict__ mask,
    int nb, double *__restrict__ round_sq, int tile_n)
{
	extern __shared__ double smem[];
	double *tile = smem;                                          // [tile_n][ndim]
	double *part = smem + (size_t) tile_n * ndim;                 // [4][kRounds][64]
	unsigned *tmask = reinterpret_cast<unsigned *>(part + 4 * kRounds * 64);   // [tile_n]

	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int i = blockIdx.x * 64 + lane;
	const int ii = i < K ? i : K - 1;
	double c[D > 0 ? D : 1];
	if (D > 0) {
#pragma unroll
		for (int k = 0; k < D; k++) c[k] = members[(size_t) ii * D + k];
	}
	const unsigned mymask = NN ? 0u : mask[ii];

	double nearest[kRounds];
#pragma unroll
	for (int b = 0; b < kRounds; b++) nearest[b] = 1e300;         // cneighbors.c:51,148

	for (int t0 = 0; t0 < K; t0 += tile_n) {
		const int n = min(tile_n, K - t0);
		__syncthreads();
		for (int e = threadIdx.x; e < n * ndim; e += kBlock) tile[e] = members[(size_t) t0 * ndim + e];
		for (int e = threadIdx.x; e < n; e += kBlock) tmask[e] = NN ? 1u : mask[t0 + e];
		__syncthreads();
		const int q = (n + 3) / 4;
		const int jbeg = wv * q, jend = min(n, jbeg + q);
		for (int jn = jbeg; jn < jend; jn++) {
			const unsigned mk = __builtin_amdgcn_readfirstlane(tmask[jn]);   // same for the whole wave
			if (mk == 0) continue;
			double d;
			if (D > 0) d = sq_distance_fixed<(D > 0 ? D : 1)>(tile + jn * D, c);
			else d = sq_distance(members + (size_t) ii * ndim, tile + jn * ndim, ndim);
			if (NN) {
				if (t0 + jn != ii) nearest[0] = fmin(nearest[0], d);
			} else {
#pragma unroll
				for (int b = 0; b < kRounds; b++)
					if ((mk >> b) & 1u) nearest[b] = fmin(nearest[b], d);   // scalar branch per round
			}
		}
	}
	// meet the four partial minima, then max over the contributing points: one atomic per
	// wave and round (each wave finishes a quarter of the rounds)
#pragma unroll
	for (int b = 0; b < kRounds; b++) part[(wv * kRounds + b) * 64 + lane] = nearest[b];
	__syncthreads();
	for (int b = wv; b < nb; b += 4) {
		double v = fmin(fmin(part[(0 * kRounds + b) * 64 + lane], part[(1 * kRounds + b) * 64 + lane]),
		                fmin(part[(2 * kRounds + b) * 64 + lane], part[(3 * kRounds + b) * 64 + lane]));
		const bool contributes = i < K && (NN ? true : (i >= 1 && !((mymask >> b) & 1u)));
		v = wave_max(contributes ? v : 0.0);
		if (lane == 0 && v > 0.0) atomic_max_nonneg(round_sq + b, v);
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

// members per LDS tile: a multiple of 4, as many as fit beside `fixed_bytes`
static int pick_tile(int ndim, size_t per_member_extra, size_t fixed_bytes)
{
	const size_t per = (size_t) ndim * sizeof(double) + per_member_extra;
	if (fixed_bytes + 4 * per > kLdsBudget) return 0;
	size_t n = (kLdsBudget - fixed_bytes) / per;
	if (n > (size_t) kMaxTile) n = kMaxTile;
	return (int) (n & ~(size_t) 3);
}

bool launch_count_within(const double *d_members, int K, int ndim, double thresh_sq,
                         const double *d_cands, int M, int *d_counts)
{
	Context *c = ctx();
	const size_t fixed = 4 * 64 * sizeof(int);
	const int tile_n = pick_tile(ndim, 0, fixed);
	if (tile_n <= 0) { set_error("ndim=%d too large for the member tile", ndim); return false; }
	const int gx = (M + 63) / 64;
	// split the members between workgroups until the grid covers the chip about twice
	int want = (2 * c->num_cus + gx - 1) / gx;
	int max_split = (K + tile_n - 1) / tile_n;
	int gy = want < max_split ? want : max_split;
	if (gy < 1) gy = 1;
	if (gy > 65535) gy = 65535;
	int kchunk = (K + gy - 1) / gy;
	kchunk = ((kchunk + tile_n - 1) / tile_n) * tile_n;
	gy = (K + kchunk - 1) / kchunk;
	const int accumulate = gy > 1 ? 1 : 0;
	if (accumulate && !MDNS_HIP(hipMemsetAsync(d_counts, 0, (size_t) M * sizeof(int), c->stream))) return false;
	const size_t lds = (size_t) tile_n * ndim * sizeof(double) + fixed;
	dim3 grid(gx, gy);
	ProfileScope prof(2);
#define COUNT_LAUNCH(D) hipLaunchKernelGGL((k_count_within<D>), grid, dim3(kBlock), lds, c->stream, \
	d_members, K, ndim, thresh_sq, d_cands, M, d_counts, kchunk, tile_n, accumulate)
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
	const size_t fixed = 4 * kRounds * 64 * sizeof(double);
	const int tile_n = pick_tile(ndim, sizeof(unsigned), fixed);
	if (tile_n <= 0) { set_error("ndim=%d too large for the member tile", ndim); return false; }
	const size_t lds = (size_t) tile_n * ndim * sizeof(double) + fixed + (size_t) tile_n * sizeof(unsigned);
	unsigned *d_mask = nullptr;
	if (!NN) {
		d_mask = (unsigned *) mask_scratch((size_t) K * sizeof(unsigned));
		if (!d_mask) return false;
	}
	dim3 grid((K + 63) / 64);
	for (int b0 = 0; b0 < nboot; b0 += kRounds) {
		const int nb = nboot - b0 < kRounds ? nboot - b0 : kRounds;
		if (!NN) {
			hipLaunchKernelGGL(k_pack_chosen, dim3((K + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
			                   d_chosen, K, nboot, b0, nb, d_mask);
			if (!launched("k_pack_chosen")) return false;
		}
		ProfileScope prof(3);
#define NEAR_LAUNCH(D) hipLaunchKernelGGL((k_nearest_chosen<D, NN>), grid, dim3(kBlock), lds, c->stream, \
	d_members, K, ndim, d_mask, nb, d_round_sq + b0, tile_n)
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
