// K6 (bootstrapped_maxdistance, cneighbors.c:125-179) below K^2: the pool is put into MORTON ORDER on
// the device, and a point only looks at the tiles of 64 members that can still hold the nearest CHOSEN
// member of one of its rounds.
//
// The all-pairs kernels (mdns_neighbors.hip) execute K^2 (point, member) steps of 28 vector
// instructions; the answer per (point, round) is a minimum over the chosen members, and with 63 % of the
// members chosen in a round that minimum sits among the point's few nearest members.  Here:
//
//   k_morton_sort<D>    ONE workgroup of 1024 threads: bounding box, a Morton key per point (the
//                       quantised coordinates interleaved), bitonic sort of (key, index) in LDS, then the
//                       members, their choice masks and the bounding box of every tile of 64 consecutive
//                       members, in sorted order (a scratch copy: the region keeps its members as given)
//   k_nearest_culled<D> a workgroup = 64 consecutive sorted points (lane = point) x 8 waves.  Every wave
//                       first meets the points' OWN tile (their nearest members, in all likelihood: good
//                       bounds at once), then the waves share the other tiles, nearer ones first, and skip
//                       every tile whose box lies farther from the points' box than the worst bound any
//                       lane still holds for a round it is left out of.  The per-pair arithmetic is
//                       k_nearest_uniform's: the member is wave-uniform, its choice bits are scalar
//                       operands of v_max_f64 / v_min_f64; the squared distance is the same separate
//                       multiply-and-add chain (-ffp-contract=off).
//
// Exactness: min over a set of members that provably contains every member closer than the current
// bound is the min over all members -- the distances themselves are computed exactly as before and min /
// max select, they do not round -- so the radius is the all-pairs kernels' bit for bit (tests:
// test_hip_parity.py k6 golden / sweep / quirk cases run both paths).  The reference's quirk that the
// point with index 0 never contributes to a round's maximum (cneighbors.c:162) travels with the point as
// a flag bit through the sort.
#include "mdns_internal.h"
#include "mdns_radius.h"
#include <cstdlib>
#include <cstring>

#pragma clang fp contract(off)

namespace mdns {

static constexpr int kSortThreads = 1024;
static constexpr int kMostSorted = 16384;                 // points at most (128 KB of keys in LDS)
static constexpr int kTile = 64;
static constexpr int kWaves = 8;                          // waves per group of 64 points
static constexpr unsigned kFirstPointBit = 1u << 31;      // the pool's point 0 (cneighbors.c:162)

__device__ __forceinline__ double wave_min_d(double v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
	return v;
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
	return v;
}

template <int D>
__global__ __launch_bounds__(kSortThreads) void k_morton_sort(
    const double *__restrict__ members, const unsigned *__restrict__ mask, int K, int N /* power of two >= K */,
    double *__restrict__ smembers, unsigned *__restrict__ smask, double *__restrict__ boxes /* [ntiles][2 D] */)
{
	extern __shared__ unsigned long long keys[];              // [N]
	__shared__ double red[2 * D][kSortThreads / 64];
	__shared__ double lo[D], inv[D];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// 1. bounding box
	double mn[D], mx[D];
#pragma unroll
	for (int k = 0; k < D; k++) { mn[k] = 1e300; mx[k] = -1e300; }
	for (int i = tid; i < K; i += kSortThreads) {
#pragma unroll
		for (int k = 0; k < D; k++) { const double v = members[(size_t) i * D + k]; mn[k] = fmin(mn[k], v); mx[k] = fmax(mx[k], v); }
	}
#pragma unroll
	for (int k = 0; k < D; k++) {
		const double a = wave_min_d(mn[k]), b = wave_max_d(mx[k]);
		if (lane == 0) { red[k][wave] = a; red[D + k][wave] = b; }
	}
	__syncthreads();
	if (tid < D) {
		double a = 1e300, b = -1e300;
		for (int w = 0; w < kSortThreads / 64; w++) { a = fmin(a, red[tid][w]); b = fmax(b, red[D + tid][w]); }
		constexpr int BITS = 48 / D;
		lo[tid] = a;
		inv[tid] = b > a ? (double) (1ull << BITS) / (b - a) : 0.0;
	}
	__syncthreads();
	// 2. keys: Morton code of the quantised coordinates above the index
	{
		constexpr int BITS = 48 / D;
		for (int i = tid; i < N; i += kSortThreads) {
			unsigned long long key = ~0ull;                       // padding sorts last
			if (i < K) {
				unsigned q[D];
#pragma unroll
				for (int k = 0; k < D; k++) {
					double t = (members[(size_t) i * D + k] - lo[k]) * inv[k];
					const double top = (double) ((1ull << BITS) - 1);
					t = t < 0.0 ? 0.0 : (t > top ? top : t);
					q[k] = (unsigned) t;
				}
				unsigned long long code = 0;
				for (int bit = BITS - 1; bit >= 0; bit--)
#pragma unroll
					for (int k = 0; k < D; k++) code = (code << 1) | ((q[k] >> bit) & 1u);
				key = (code << 16) | (unsigned long long) i;
			}
			keys[i] = key;
		}
	}
	__syncthreads();
	// 3. bitonic sort, ascending
	for (int size = 2; size <= N; size <<= 1)
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int t = tid; t < N / 2; t += kSortThreads) {
				const int i = 2 * t - (t & (stride - 1));             // lower index of the pair
				const int j = i + stride;
				const bool up = (i & size) == 0;
				const unsigned long long a = keys[i], b = keys[j];
				if ((a > b) == up) { keys[i] = b; keys[j] = a; }
			}
			__syncthreads();
		}
	// 4. members and masks in sorted order
	for (int p = tid; p < K; p += kSortThreads) {
		const int i = (int) (keys[p] & 0xffffull);
#pragma unroll
		for (int k = 0; k < D; k++) smembers[(size_t) p * D + k] = members[(size_t) i * D + k];
		smask[p] = (mask[i] & ~kFirstPointBit) | (i == 0 ? kFirstPointBit : 0u);
	}
	// 5. the bounding box of every tile of 64 (a wave per tile)
	const int ntiles = (K + kTile - 1) / kTile;
	for (int t = wave; t < ntiles; t += kSortThreads / 64) {
		const int p = t * kTile + lane;
		const bool live = p < K;
		const int i = live ? (int) (keys[p] & 0xffffull) : 0;
#pragma unroll
		for (int k = 0; k < D; k++) {
			const double v = members[(size_t) i * D + k];
			const double a = wave_min_d(live ? v : 1e300), b = wave_max_d(live ? v : -1e300);
			if (lane == 0) { boxes[(size_t) t * 2 * D + k] = a; boxes[(size_t) t * 2 * D + D + k] = b; }
		}
	}
}

__device__ __forceinline__ double min_skip(double a, double b)
{
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// the wave's LDS writes before its LDS reads (no workgroup barrier inside the divergent tile loop)
#define MDNS_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <int D, int RT>
__global__ __launch_bounds__(kWaves * 64) void k_nearest_culled(
    const double *__restrict__ smembers, const unsigned *__restrict__ smask, const double *__restrict__ boxes, int K, int nb,
    double *__restrict__ round_sq, BootstrapFinish fin, int nround_all)
{
	__shared__ double tiles[kWaves][kTile * D];
	__shared__ unsigned tmasks[kWaves][kTile];
	__shared__ double meet[kWaves][RT][64];
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int t0 = blockIdx.x;                                    // the tile these 64 points are
	const int ntiles = (K + kTile - 1) / kTile;
	const int i = t0 * kTile + lane;
	const bool live = i < K;
	const int ii = live ? i : K - 1;
	double c[D];
#pragma unroll
	for (int k = 0; k < D; k++) c[k] = smembers[(size_t) ii * D + k];
	const unsigned mymask = smask[ii];
	// rounds this point is left out of (and may contribute to): the only ones whose nearest matters
	bool counts[RT];
	double nearest[RT];
#pragma unroll
	for (int b = 0; b < RT; b++) {
		counts[b] = live && b < nb && !((mymask >> b) & 1u) && !(mymask & kFirstPointBit);
		nearest[b] = counts[b] ? 1e300 : 0.0;                    // cneighbors.c:148
	}
	// the points' own bounding box
	double glo[D], ghi[D];
#pragma unroll
	for (int k = 0; k < D; k++) { glo[k] = boxes[(size_t) t0 * 2 * D + k]; ghi[k] = boxes[(size_t) t0 * 2 * D + D + k]; }
	const double PINF = __longlong_as_double(0x7ff0000000000000LL), NINF = __longlong_as_double((long long) 0xfff0000000000000ULL);
	double *tile = tiles[wv];
	unsigned *tmask = tmasks[wv];
	auto process = [&](int t) {
		const int first = t * kTile;
		const int n = min(kTile, K - first);
		MDNS_WAVE_LDS_SYNC();                                     // (the previous tile has been read)
		if (lane < n) {
#pragma unroll
			for (int k = 0; k < D; k++) tile[lane * D + k] = smembers[(size_t) (first + lane) * D + k];
			tmask[lane] = smask[first + lane];
		}
		MDNS_WAVE_LDS_SYNC();
		for (int jn = 0; jn < n; jn++) {
			double d = 0.0;
#pragma unroll
			for (int k = 0; k < D; k++) {
				const double diff = tile[jn * D + k] - c[k];
				d = d + diff * diff;
			}
			const unsigned m = (unsigned) __builtin_amdgcn_readfirstlane((int) tmask[jn]);
#pragma unroll
			for (int b = 0; b < RT; b++) {
				const double S = (m >> b & 1u) ? NINF : PINF;         // scalar: s_bitcmp1 + s_cselect_b64
				double tt;
				asm("v_max_f64 %0, %1, %2" : "=v"(tt) : "v"(d), "s"(S));
				nearest[b] = min_skip(nearest[b], tt);
			}
		}
	};
	auto worst_of_wave = [&]() {
		double w = 0.0;
#pragma unroll
		for (int b = 0; b < RT; b++) w = fmax(w, counts[b] ? nearest[b] : 0.0);
		return wave_max_d(w);
	};
	// every wave meets the own tile first, then the waves share the others, nearer ones first
	process(t0);
	double worst = worst_of_wave();
	for (int dd = 1 + wv; dd < ntiles && worst > 0.0; dd += kWaves) {
#pragma unroll
		for (int side = 0; side < 2; side++) {
			const int t = side == 0 ? t0 + dd : t0 - dd;
			if (t < 0 || t >= ntiles) continue;
			double dist2 = 0.0;
#pragma unroll
			for (int k = 0; k < D; k++) {
				const double blo = boxes[(size_t) t * 2 * D + k], bhi = boxes[(size_t) t * 2 * D + D + k];
				const double gap = fmax(0.0, fmax(blo - ghi[k], glo[k] - bhi));
				dist2 = dist2 + gap * gap;
			}
			// (a tile farther than the worst bound cannot lower any minimum that matters; the margin
			// covers the rounding of the box arithmetic)
			if (dist2 > worst * (1.0 + 1e-9)) continue;
			process(t);
			worst = worst_of_wave();
		}
	}
	// min over the waves, then per round the max over the contributing points
#pragma unroll
	for (int b = 0; b < RT; b++) meet[wv][b][lane] = nearest[b];
	__syncthreads();
	for (int b = wv; b < nb; b += kWaves) {
		double v = meet[0][b][lane];
#pragma unroll
		for (int w = 1; w < kWaves; w++) v = fmin(v, meet[w][b][lane]);
		const bool contributes = live && !((mymask >> b) & 1u) && !(mymask & kFirstPointBit);
		v = wave_max_d(contributes ? v : 0.0);
		if (lane == 0 && v > 0.0)
			atomicMax(reinterpret_cast<unsigned long long *>(round_sq + b), (unsigned long long) __double_as_longlong(v));
	}
	if (!fin.counter) return;
	// the workgroup that finishes last turns the maxima into {radius, threshold} (as k_nearest_chosen)
	handover_release();
	__syncthreads();
	if (wv != 0) return;
	unsigned ticket = 0;
	if (lane == 0) ticket = atomicAdd(fin.counter, 1u);
	if (__shfl(ticket, 0, 64) != gridDim.x - 1) return;
	handover_acquire();
	double best = 0.0;
	for (int b = lane; b < nround_all; b += 64)
		best = fmax(best, __hip_atomic_load(round_sq + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	best = wave_max_d(best);
	for (int b = lane; b < nround_all; b += 64) round_sq[b] = 0.0;
	if (lane != 0) return;
	double radius, thresh;
	radius_and_threshold(best, radius, thresh);
	fin.d_res->radius = radius;
	fin.d_res->thresh = thresh;
	*fin.counter = 0;
	mail_store(&fin.h_res->radius, radius);
	mail_store(&fin.h_res->thresh, thresh);
	mail_raise(&fin.h_res->seq, fin.seq);
}

static bool launched(const char *name)
{
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) return true;
	set_error("launch of %s failed: %s", name, hipGetErrorString(e));
	return false;
}

// whether the sorted form applies (pools of 1 024 .. 16 384 points in at most 5 dimensions, 10 or 16 rounds)
bool bootstrap_sorted_applies(int K, int ndim, int nbootstraps)
{
	// Opt-in (MDNS_K6_PATH=sorted): measured on the MI355X (profiles/r04_k6_sorted.txt) the pair of kernels
	// takes 104 + 101 us at 5 000 points against 51 us for the all-pairs pair -- the sort is one workgroup on
	// one CU (91 bitonic stages of 1 us), and a wave that walks its tiles alone has nothing to hide the
	// latency of its 28-instruction chain behind (300 cycles per member instead of the 28 x 5 the all-pairs
	// kernel reaches with five waves per SIMD).  Kept for its tests and as the starting point of a version
	// with a multi-workgroup sort and four members in flight per wave.
	static const char *forced = getenv("MDNS_K6_PATH");
	if (!forced || strcmp(forced, "sorted") != 0) return false;
	return ndim >= 1 && ndim <= 5 && K >= 1024 && K <= kMostSorted && nbootstraps >= 1 && nbootstraps <= 16;
}

bool launch_bootstrap_sorted(const double *d_members, int K, int ndim, const unsigned *d_packed, int nbootstraps,
                             double *d_round_sq, const BootstrapFinish *finish)
{
	Context *c = ctx();
	int N = 1024;
	while (N < K) N <<= 1;
	const int ntiles = (K + kTile - 1) / kTile;
	const size_t mbytes = ((size_t) K * ndim * sizeof(double) + 255) & ~(size_t) 255;
	const size_t kbytes = ((size_t) K * sizeof(unsigned) + 255) & ~(size_t) 255;
	const size_t bbytes = (size_t) ntiles * 2 * ndim * sizeof(double);
	char *scratch = (char *) device_scratch(mbytes + kbytes + bbytes);
	if (!scratch) return false;
	double *sm = (double *) scratch;
	unsigned *sk = (unsigned *) (scratch + mbytes);
	double *boxes = (double *) (scratch + mbytes + kbytes);
	const size_t lds = (size_t) N * sizeof(unsigned long long);
	const int rt = nbootstraps <= 10 ? 10 : 16;
	ProfileScope prof(3);
	note_kernel(3, "k_nearest_culled<%d, %d>", ndim, rt);
#define SORT_LAUNCH(D) do { \
	static bool attr_set_##D = false; \
	if (!attr_set_##D) { (void) hipFuncSetAttribute((const void *) k_morton_sort<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); attr_set_##D = true; } \
	hipLaunchKernelGGL((k_morton_sort<D>), dim3(1), dim3(kSortThreads), lds, c->stream, d_members, d_packed, K, N, sm, sk, boxes); \
	if (!launched("k_morton_sort")) return false; \
	if (rt == 10) hipLaunchKernelGGL((k_nearest_culled<D, 10>), dim3(ntiles), dim3(kWaves * 64), 0, c->stream, (const double *) sm, (const unsigned *) sk, (const double *) boxes, K, nbootstraps, d_round_sq, *finish, nbootstraps); \
	else hipLaunchKernelGGL((k_nearest_culled<D, 16>), dim3(ntiles), dim3(kWaves * 64), 0, c->stream, (const double *) sm, (const unsigned *) sk, (const double *) boxes, K, nbootstraps, d_round_sq, *finish, nbootstraps); \
	} while (0)
	switch (ndim) { case 1: SORT_LAUNCH(1); break; case 2: SORT_LAUNCH(2); break; case 3: SORT_LAUNCH(3); break;
	                case 4: SORT_LAUNCH(4); break; default: SORT_LAUNCH(5); break; }
#undef SORT_LAUNCH
	return launched("k_nearest_culled");
}

}  // namespace mdns
