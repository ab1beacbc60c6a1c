// RadFriends geometry kernels for gfx950 (MI355X): the all-pairs distance tests that replace
// clustering/cneighbors.c.  Members (the live-point pool) are tiled through LDS; a lane is a
// (point, member-slice) pair: one candidate (K3/K4) or one pool point (K5/K6) and a share of
// the members (see Geo below).
//
// Integer / bit exactness.  The squared distance is accumulated from 0 over the dimensions in
// ascending order with separate multiply and add (no FMA: this file is compiled with
// -ffp-contract=off and says so again below), which is the arithmetic of cneighbors.c:55-58.
// The reference then tests  sqrt(d) < r  (cneighbors.c:88,109); since the correctly rounded
// square root is monotone that is  d < T  with T = the smallest double whose root is >= r,
// found on the host (mdns::sqrt_threshold) or, at the end of a radius computation, by one lane
// of the bootstrap kernel's last workgroup (radius_and_threshold); the inner loops need no sqrt.  For the
// radii (cneighbors.c:64-71,160-174) the root is taken once after the max of the min squared
// distances -- the same number because sqrt is monotone.
#include "mdns_internal.h"
#include "mdns_radius.h"
#include <cstdlib>
#include <cstring>

#pragma clang fp contract(off)

namespace mdns {

static constexpr int kBlock = 256;          // 4 waves
static constexpr int kMaxTile = 512;        // members per LDS tile (fewer when ndim is large)
static constexpr int kMaxRegDim = 8;        // dimensions kept in registers
static constexpr int kRounds = 16;          // bootstrap rounds per pass (cneighbors uses 10)
static constexpr size_t kLdsBudget = 60 * 1024;

// Lane geometry shared by the kernels below.  A wave is PTS points x SL member-slices
// (PTS * SL = 64): lane = point + PTS * slice.  With the 4 waves of a workgroup that makes
// 4*SL slices; slice q scans members q, q + 4*SL, q + 8*SL, ... of every LDS tile.  SL = 1 is
// the throughput shape (64 points per workgroup, LDS reads are pure broadcasts); SL = 4 is the
// latency shape for small problems (4x the workgroups, a quarter of the trip count per lane).
template <int SL> struct Geo {
	static constexpr int PTS = 64 / SL;
	static constexpr int NSLICE = 4 * SL;
};

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

// Copies `count` words from global memory into LDS with all of a thread's loads in flight at
// once (a plain strided copy loop is compiled to load -> wait -> store per trip, i.e. one full
// memory latency per 256 words: that was 10 of the 14 us of the K = 400 bootstrap kernel).
template <typename T>
__device__ __forceinline__ void stage_to_lds(T *__restrict__ dst, const T *__restrict__ src, int count)
{
	constexpr int U = 8;
	for (int base = 0; base < count; base += U * kBlock) {
		T r[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const int e = base + u * kBlock + (int) threadIdx.x;
			r[u] = src[e < count ? e : count - 1];
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			const int e = base + u * kBlock + (int) threadIdx.x;
			if (e < count) dst[e] = r[u];
		}
	}
}


// ---------------------------------------------------------------------------------------
// K3 / K4: how many members lie strictly within the radius of each candidate
// ---------------------------------------------------------------------------------------
// grid.x tiles the candidates (PTS per workgroup), grid.y splits the members between
// workgroups when the pool is large (partial counts then combine with integer atomics into a
// zeroed buffer -- exact whatever the order).  D == 0: runtime ndim (generic path).
template <int D, int SL>
__global__ __launch_bounds__(kBlock) void k_count_within(
    const double *__restrict__ members, int K, int ndim, double thresh_value,
    const RegionResult *__restrict__ res, const double *__restrict__ cands, int M,
    int *__restrict__ counts, int kchunk, int tile_n, int accumulate, CountMail mail)
{
	constexpr int PTS = Geo<SL>::PTS, NSLICE = Geo<SL>::NSLICE;
	// the threshold either came with the launch (host-known radius) or was left in device
	// memory by the radius computation that precedes this launch in stream order
	const double thresh_sq = res ? res->thresh : thresh_value;
	extern __shared__ double smem[];
	double *tile = smem;                                                  // [tile_n][ndim]
	int *part = reinterpret_cast<int *>(smem + (size_t) tile_n * ndim);   // [4][PTS]
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int pt = lane % PTS;
	const int slice = wv * SL + lane / PTS;
	const int j = blockIdx.x * PTS + pt;
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
		stage_to_lds(tile, members + (size_t) t0 * ndim, n * ndim);
		__syncthreads();
		if (D > 0) {
#pragma unroll 4
			for (int i = slice; i < n; i += NSLICE)
				hits += sq_distance_fixed<(D > 0 ? D : 1)>(tile + i * D, c) < thresh_sq ? 1 : 0;
		} else {
			const double *cj = cands + (size_t) jj * ndim;
			for (int i = slice; i < n; i += NSLICE)
				hits += sq_distance(tile + i * ndim, cj, ndim) < thresh_sq ? 1 : 0;
		}
	}
	// sum over the slices of this wave (lanes pt, pt + PTS, ...), then over the waves
#pragma unroll
	for (int off = PTS; off < 64; off <<= 1) hits += __shfl_xor(hits, off, 64);
	if (lane < PTS) part[wv * PTS + pt] = hits;
	__syncthreads();
	if (wv == 0 && lane < PTS && j < M) {
		const int total = (part[pt] + part[PTS + pt]) + (part[2 * PTS + pt] + part[3 * PTS + pt]);
		if (accumulate) { if (total) atomicAdd(counts + j, total); }
		else if (mail.seq_at) __hip_atomic_store(counts + j, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		else counts[j] = total;
	}
	if (mail.seq_at) {
		// `counts` is host memory mapped into the device (no member split; system-scope stores): once every
		// workgroup's stores are out -- waited for, not fenced: see handover_release -- the last one to get here
		// raises `seq` for the polling host
		if (wv == 0) handover_release();                           // (only wave 0 stored)
		__syncthreads();
		if (threadIdx.x == 0) {
			const int done = atomicAdd(mail.ticket, 1);
			if (done == (int) (gridDim.x * gridDim.y) - 1) {
				__hip_atomic_store(mail.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch (stream order)
				mail_raise(mail.seq_at, mail.seq);
			}
		}
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

// min(a, b) where b may be a QUIET NaN meaning "not a candidate": v_min_f64 returns the other
// operand for a quiet NaN.  Written as asm so that the compiler does not put a canonicalising
// v_max_f64 in front of every use (it cannot know the operand is already quiet).
__device__ __forceinline__ double min_or_skip(double a, double b)
{
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// non-negative doubles order like their bit patterns
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v)
{
	atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long) __double_as_longlong(v));
}

// chosen f64[K][nboot] (cneighbors.c:146 tests != 0) -> one bit per round of the window
// [b0, b0+nb) for every pool point; also clears the window's slots of round_sq, which
// k_nearest_chosen then raises with atomic max
__global__ void k_pack_chosen(const double *__restrict__ chosen, int K, int nboot, int b0, int nb,
                              unsigned *__restrict__ mask, double *__restrict__ round_sq)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nb) round_sq[b0 + i] = 0.0;
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
// The distance of a pair is computed once and offered to all rounds without branches: for a
// round in which member j is NOT chosen the exponent and quiet bits of the distance are forced
// to ones (-> quiet NaN), which v_min_f64 discards.  Partial minima of slices and waves meet by
// shuffles and LDS (min is exact, the split cannot change the result).
// RT = rounds carried per pass when !NN: 10 (what the reference's callers ask for,
// radfriendsregion.py:59: nbootstraps = 10) or kRounds.
template <int D, bool NN, int SL, int RT>
__global__ __launch_bounds__(kBlock) void k_nearest_chosen(
    const double *__restrict__ members, int K, int ndim, const unsigned *__restrict__ mask,
    int nb, double *__restrict__ round_sq, int tile_n, BootstrapFinish fin, double *round_all, int nround_all,
    const double *__restrict__ chosen, int nboot, int b0)
{
	constexpr int PTS = Geo<SL>::PTS, NSLICE = Geo<SL>::NSLICE;
	constexpr int NR = NN ? 1 : RT;
	extern __shared__ double smem[];
	double *tile = smem;                                          // [tile_n][ndim]
	double *part = smem + (size_t) tile_n * ndim;                 // [4][NR][PTS]
	unsigned *tmask = reinterpret_cast<unsigned *>(part + 4 * NR * PTS);   // [tile_n]

	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int pt = lane % PTS;
	const int slice = wv * SL + lane / PTS;
	const int i = blockIdx.x * PTS + pt;
	const int ii = i < K ? i : K - 1;
	double c[D > 0 ? D : 1];
	if (D > 0) {
#pragma unroll
		for (int k = 0; k < D; k++) c[k] = members[(size_t) ii * D + k];
	}
	// `chosen` != nullptr: small pools read the reference's f64[K][nboot] choice matrix
	// themselves (cneighbors.c:146 tests != 0) instead of the bit masks of k_pack_chosen
	unsigned mymask = (NN || chosen) ? 0u : mask[ii];
	if (!NN && chosen) {
		// own flags now, together with the coordinates: nothing global is left for the epilogue
		const double *row = chosen + (size_t) ii * nboot + b0;
		double flag[NR];
#pragma unroll
		for (int b = 0; b < NR; b++) flag[b] = row[b < nb ? b : 0];
#pragma unroll
		for (int b = 0; b < NR; b++) mymask |= (b < nb && flag[b] != 0.0 ? 1u : 0u) << b;
	}

	double nearest[NR];
#pragma unroll
	for (int b = 0; b < NR; b++) nearest[b] = 1e300;              // cneighbors.c:51,148

	for (int t0 = 0; t0 < K; t0 += tile_n) {
		const int n = min(tile_n, K - t0);
		__syncthreads();
		stage_to_lds(tile, members + (size_t) t0 * ndim, n * ndim);
		if (!NN && !chosen) stage_to_lds(tmask, mask + t0, n);
		if (!NN && chosen) {
			// the rows of all this thread's tile members requested before the first is used
			constexpr int TRIPS = kMaxTile / kBlock;
			double flag[TRIPS][NR];
#pragma unroll
			for (int t = 0; t < TRIPS; t++) {
				const int e = min((int) threadIdx.x + t * kBlock, n - 1);
				const double *row = chosen + (size_t) (t0 + e) * nboot + b0;
#pragma unroll
				for (int b = 0; b < NR; b++) flag[t][b] = row[b < nb ? b : 0];
			}
#pragma unroll
			for (int t = 0; t < TRIPS; t++) {
				const int e = (int) threadIdx.x + t * kBlock;
				unsigned m = 0;
#pragma unroll
				for (int b = 0; b < NR; b++) m |= (b < nb && flag[t][b] != 0.0 ? 1u : 0u) << b;
				if (e < n) tmask[e] = m;
			}
		}
		__syncthreads();
		// one member: its squared distance offered to every round
		auto offer = [&](int jn) {
			double d;
			if (D > 0) d = sq_distance_fixed<(D > 0 ? D : 1)>(tile + jn * D, c);
			else d = sq_distance(members + (size_t) ii * ndim, tile + jn * ndim, ndim);
			if (NN) {
				if (t0 + jn != ii) nearest[0] = fmin(nearest[0], d);
			} else {
				const int notchosen = (int) ~tmask[jn];
				const unsigned lo = (unsigned) __double2loint(d);
				const unsigned hi = (unsigned) __double2hiint(d);
#pragma unroll
				for (int b = 0; b < NR; b++) {
					// 0 where chosen in round b, all ones where not (1-bit signed field extract);
					// exponent all ones + quiet bit = quiet NaN whatever the mantissa
					const unsigned kill = (unsigned) __builtin_amdgcn_sbfe(notchosen, b, 1);
					const double cand = __hiloint2double((int) (hi | (kill & 0x7ff80000u)), (int) lo);
					nearest[b] = min_or_skip(nearest[b], cand);
				}
			}
		};
		// two members per trip: the second one's LDS reads and distance overlap the first one's
		// chain of minima (one wave per SIMD at the usual pool sizes: nothing else hides them)
		int jn = slice;
		for (; jn + NSLICE < n; jn += 2 * NSLICE) {
			offer(jn);
			offer(jn + NSLICE);
		}
		if (jn < n) offer(jn);
	}
	// min over the slices of this wave, then over the waves; finally the max over the
	// contributing points: one atomic per wave and round
#pragma unroll
	for (int b = 0; b < NR; b++) {
		double v = nearest[b];
#pragma unroll
		for (int off = PTS; off < 64; off <<= 1) v = fmin(v, __shfl_xor(v, off, 64));
		if (lane < PTS) part[(wv * NR + b) * PTS + pt] = v;
	}
	__syncthreads();
	for (int b = wv; b < (NN ? 1 : nb); b += 4) {
		double v = 0.0;
		if (lane < PTS) {
			v = fmin(fmin(part[(0 * NR + b) * PTS + pt], part[(1 * NR + b) * PTS + pt]),
			         fmin(part[(2 * NR + b) * PTS + pt], part[(3 * NR + b) * PTS + pt]));
			const bool contributes = i < K && (NN ? true : (i >= 1 && !((mymask >> b) & 1u)));
			if (!contributes) v = 0.0;
		}
		v = wave_max(v);
		if (lane == 0 && v > 0.0) atomic_max_nonneg(round_sq + b, v);
	}
	if (!fin.counter) return;
	// The workgroup that finishes last turns the maxima into {radius, threshold} for the
	// membership kernel (device copy) and for the host (mapped memory, `seq` written last).
	handover_release();                                   // this workgroup's atomics before its ticket
	__syncthreads();
	if (wv != 0) return;
	unsigned ticket = 0;
	if (lane == 0) ticket = atomicAdd(fin.counter, 1u);
	if (__shfl(ticket, 0, 64) != gridDim.x - 1) return;
	handover_acquire();
	double best = 0.0;
	for (int b = lane; b < nround_all; b += 64)
		best = fmax(best, __hip_atomic_load(round_all + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	best = wave_max(best);
	// the slots go back to zero for the next computation (nobody else reads them any more)
	for (int b = lane; b < nround_all; b += 64) round_all[b] = 0.0;
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

// K6 again, for packed choices (the shape every region of a run has): the member a wave looks at
// is the same for all of its lanes.  A workgroup owns 64 points (lane = point) and a chunk of the
// members (grid.y); its four waves walk that chunk interleaved.  Because the member is
// wave-uniform its choice bits are a SCALAR: per round  v_max_f64 t, d, S  with S = -inf where the
// member is chosen (t = d) and +inf where it is not (t = +inf), then  v_min_f64 nearest, t  --
// two vector instructions per round and pair instead of the four of the masked-NaN form above
// (sign-extracting the bit, and-ing, or-ing into the exponent, min), with the distance still
// computed once per pair; the scalar unit prepares S (s_bitcmp1 + s_cselect) in the shadow of
// another wave's vector work.  Same minima, bit for bit (min and max select, they do not round).
// Per-point minima of a chunk go to part[y][K][RT] with plain stores; k_nearest_finish takes the
// min over the chunks and the max over the left-out points (cneighbors.c:160-168).
template <int D, int RT>
__global__ __launch_bounds__(kBlock) void k_nearest_uniform(
    const double *__restrict__ members, int K, const unsigned *__restrict__ mask, int kchunk, int tile_n,
    double *__restrict__ part, unsigned *__restrict__ block_tickets, int nb, double *__restrict__ round_sq,
    BootstrapFinish fin, int nround_all)
{
	extern __shared__ double smem[];
	double *tile = smem;                                                   // [tile_n][D]
	double *meet = smem + (size_t) tile_n * D;                             // [4][RT][64]
	unsigned *tmask = reinterpret_cast<unsigned *>(meet + 4 * RT * 64);    // [tile_n]
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int i = blockIdx.x * 64 + lane;
	const int ii = i < K ? i : K - 1;
	double c[D];
#pragma unroll
	for (int k = 0; k < D; k++) c[k] = members[(size_t) ii * D + k];
	double nearest[RT];
#pragma unroll
	for (int b = 0; b < RT; b++) nearest[b] = 1e300;                      // cneighbors.c:148
	const int kbeg = blockIdx.y * kchunk, kend = min(K, kbeg + kchunk);
	const double PINF = __longlong_as_double(0x7ff0000000000000LL), NINF = __longlong_as_double((long long) 0xfff0000000000000ULL);
	for (int t0 = kbeg; t0 < kend; t0 += tile_n) {
		const int n = min(tile_n, kend - t0);
		__syncthreads();
		stage_to_lds(tile, members + (size_t) t0 * D, n * D);
		stage_to_lds(tmask, mask + t0, n);
		__syncthreads();
		auto offer = [&](int jn) {
			const double d = sq_distance_fixed<D>(tile + jn * D, c);
			const unsigned m = (unsigned) __builtin_amdgcn_readfirstlane((int) tmask[jn]);
			// (tried: `if (m >> b & 1u) nearest[b] = min(nearest[b], d)` -- a scalar branch around ONE
			// v_min_f64, 14 vector instructions per step instead of 28: 47.7 / 124 / 409 / 2198 us at
			// 5 000 / 9 000 / 20 000 / 50 000 points against 51 / 111 / 400 / 2130 -- the ten
			// s_bitcmp1 + s_cbranch pairs per step cost what the skipped instructions save)
#pragma unroll
			for (int b = 0; b < RT; b++) {
				const double S = (m >> b & 1u) ? NINF : PINF;             // scalar: s_bitcmp1 + s_cselect_b64
				// (written as asm: left to itself the compiler turns max(d, +-inf) into two
				// v_cndmask_b32 per round -- three vector instructions instead of two; so does the
				// quiet-NaN form d.hi | 0x7ff80000 of the classic kernel with a scalar mask: it needs a
				// copy of d.lo per round to form the register pair)
				double t;
				asm("v_max_f64 %0, %1, %2" : "=v"(t) : "v"(d), "s"(S));
				nearest[b] = min_or_skip(nearest[b], t);
			}
		};
		int jn = wv;
		for (; jn + 4 < n; jn += 8) { offer(jn); offer(jn + 4); }
		if (jn < n) offer(jn);
	}
#pragma unroll
	for (int b = 0; b < RT; b++) meet[(wv * RT + b) * 64 + lane] = nearest[b];
	__syncthreads();
	// [RT][64] minima over the four waves, written as rows of part[y][i][RT]
	for (int e = threadIdx.x; e < RT * 64; e += kBlock) {
		const int b = e / 64, l = e % 64;
		const double v = fmin(fmin(meet[(0 * RT + b) * 64 + l], meet[(1 * RT + b) * 64 + l]),
		                      fmin(meet[(2 * RT + b) * 64 + l], meet[(3 * RT + b) * 64 + l]));
		const int p = blockIdx.x * 64 + l;
		// (agent scope when another workgroup of this launch merges them: see handover_release)
		if (p < K) { if (block_tickets) __hip_atomic_store(&part[((size_t) blockIdx.y * K + p) * RT + b], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else part[((size_t) blockIdx.y * K + p) * RT + b] = v; }
	}
	if (!block_tickets) return;                        // (the merge is a kernel of its own: k_nearest_finish)
	// Round 4: the merge rides in this kernel.  Of the gridDim.y workgroups that share these 64 points the
	// one that arrives LAST takes the min over the member chunks and the max over the left-out points
	// (cneighbors.c:160-168); of those, the last one finishes the radius (k_nearest_chosen's epilogue).
	__shared__ int s_last;
	__shared__ double wmax[4][RT];
	handover_release();
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned t = atomicAdd(block_tickets + blockIdx.x, 1u);
		s_last = t == gridDim.y - 1 ? 1 : 0;
		if (s_last) block_tickets[blockIdx.x] = 0;       // for the next computation (stream order)
	}
	__syncthreads();
	if (!s_last) return;
	handover_acquire();
	{
		const int ny = (int) gridDim.y;
		const int i2 = blockIdx.x * 64 + ((int) threadIdx.x >> 2), yq = threadIdx.x & 3;
		double v[RT];
#pragma unroll
		for (int b = 0; b < RT; b++) v[b] = 0.0;
		const bool counts = i2 < K && i2 >= 1;
		if (counts) {
#pragma unroll
			for (int b = 0; b < RT; b++) v[b] = 1e300;
			for (int y = yq; y < ny; y += 4) {
				const double *row = part + ((size_t) y * K + i2) * RT;
#pragma unroll
				for (int b = 0; b < RT; b++) v[b] = fmin(v[b], __hip_atomic_load(row + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
			}
		}
#pragma unroll
		for (int b = 0; b < RT; b++) {
			v[b] = fmin(v[b], __shfl_xor(v[b], 1, 64));
			v[b] = fmin(v[b], __shfl_xor(v[b], 2, 64));
		}
		if (counts) {
			const unsigned m = mask[i2];
#pragma unroll
			for (int b = 0; b < RT; b++) if (b >= nb || (m >> b & 1u)) v[b] = 0.0;
		}
#pragma unroll
		for (int b = 0; b < RT; b++) {
			const double w = wave_max(v[b]);
			if (lane == 0) wmax[wv][b] = w;
		}
		__syncthreads();
		if ((int) threadIdx.x < RT && (int) threadIdx.x < nb) {
			const double w = fmax(fmax(wmax[0][threadIdx.x], wmax[1][threadIdx.x]), fmax(wmax[2][threadIdx.x], wmax[3][threadIdx.x]));
			if (w > 0.0) atomic_max_nonneg(round_sq + threadIdx.x, w);
		}
	}
	if (!fin.counter) return;
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
	best = wave_max(best);
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

// min over the member chunks, then per round the max over the left-out points with index >= 1
// (cneighbors.c:160-168; the reference's loop starts at 1), and -- last workgroup -- radius and
// threshold for the membership kernel and the host (see k_nearest_chosen)
template <int RT>
__global__ __launch_bounds__(kBlock) void k_nearest_finish(
    const double *__restrict__ part, int K, int ny, const unsigned *__restrict__ mask, int nb,
    double *__restrict__ round_sq, BootstrapFinish fin, int nround_all)
{
	__shared__ double wmax[4][RT];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	// four lanes per point, each with every fourth member chunk (one lane per point walking all
	// chunks: 12 us at 5 000 points in 20 workgroups -- a chain of dependent row reads)
	const int i = blockIdx.x * (kBlock / 4) + (threadIdx.x >> 2), yq = threadIdx.x & 3;
	double v[RT];
#pragma unroll
	for (int b = 0; b < RT; b++) v[b] = 0.0;
	const bool counts = i < K && i >= 1;
	if (counts) {
#pragma unroll
		for (int b = 0; b < RT; b++) v[b] = 1e300;
		for (int y = yq; y < ny; y += 4) {
			const double *row = part + ((size_t) y * K + i) * RT;
#pragma unroll
			for (int b = 0; b < RT; b++) v[b] = fmin(v[b], row[b]);
		}
	}
#pragma unroll
	for (int b = 0; b < RT; b++) {
		v[b] = fmin(v[b], __shfl_xor(v[b], 1, 64));
		v[b] = fmin(v[b], __shfl_xor(v[b], 2, 64));
	}
	if (counts) {
		const unsigned m = mask[i];
#pragma unroll
		for (int b = 0; b < RT; b++) if (b >= nb || (m >> b & 1u)) v[b] = 0.0;     // chosen points do not contribute
	}
#pragma unroll
	for (int b = 0; b < RT; b++) {
		const double w = wave_max(v[b]);
		if (lane == 0) wmax[wv][b] = w;
	}
	__syncthreads();
	if (threadIdx.x < RT && threadIdx.x < nb) {
		const double w = fmax(fmax(wmax[0][threadIdx.x], wmax[1][threadIdx.x]), fmax(wmax[2][threadIdx.x], wmax[3][threadIdx.x]));
		if (w > 0.0) atomic_max_nonneg(round_sq + threadIdx.x, w);
	}
	if (!fin.counter) return;
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
	best = wave_max(best);
	for (int b = lane; b < nround_all; b += 64) round_sq[b] = 0.0;      // the slots go back to zero
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

// members per LDS tile: a multiple of 16, as many as fit beside `fixed_bytes`
static int pick_tile(int ndim, size_t per_member_extra, size_t fixed_bytes)
{
	const size_t per = (size_t) ndim * sizeof(double) + per_member_extra;
	if (fixed_bytes + 16 * per > kLdsBudget) return 0;
	size_t n = (kLdsBudget - fixed_bytes) / per;
	if (n > (size_t) kMaxTile) n = kMaxTile;
	return (int) (n & ~(size_t) 15);
}

#define MDNS_DIM_SWITCH(ndim, LAUNCH) \
	switch ((ndim) <= kMaxRegDim ? (ndim) : 0) { \
	case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; \
	case 4: LAUNCH(4); break; case 5: LAUNCH(5); break; case 6: LAUNCH(6); break; \
	case 7: LAUNCH(7); break; case 8: LAUNCH(8); break; default: LAUNCH(0); break; }

bool launch_count_within(const double *d_members, int K, int ndim, double thresh_sq,
                         const RegionResult *d_res, const double *d_cands, int M, int *d_counts, const CountMail *mail)
{
	Context *c = ctx();
	const CountMail none = {nullptr, nullptr, 0};
	const CountMail post = mail ? *mail : none;
	const size_t fixed = 4 * 64 * sizeof(int);
	const int tile_n = pick_tile(ndim, 0, fixed);
	if (tile_n <= 0) { set_error("ndim=%d too large for the member tile", ndim); return false; }
	// few candidates: 16 per workgroup and 16 member slices (latency shape)
	const bool small = (M + 63) / 64 < 2 * c->num_cus;
	// with a mailbox the members are not split between workgroups (no partial counts): 4 points per
	// workgroup and 64 member slices instead, so that the ~1000 proposals of a membership call
	// still make 250 workgroups (16 points x 16 slices: 63 workgroups, 20 us per call in a real run)
	const bool fine = mail != nullptr && small;
	const int pts = fine ? 4 : (small ? 16 : 64);
	const int gx = (M + pts - 1) / pts;
	// split the members between workgroups until the grid covers the chip about twice
	int want = (2 * c->num_cus + gx - 1) / gx;
	int max_split = (K + tile_n - 1) / tile_n;
	int gy = want < max_split ? want : max_split;
	if (mail) gy = 1;                                  // results go straight to the host: no partial counts
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
#define COUNT_LAUNCH(D) do { if (fine) hipLaunchKernelGGL((k_count_within<D, 16>), grid, dim3(kBlock), lds, c->stream, \
		d_members, K, ndim, thresh_sq, d_res, d_cands, M, d_counts, kchunk, tile_n, accumulate, post); \
	else if (small) hipLaunchKernelGGL((k_count_within<D, 4>), grid, dim3(kBlock), lds, c->stream, \
		d_members, K, ndim, thresh_sq, d_res, d_cands, M, d_counts, kchunk, tile_n, accumulate, post); \
	else hipLaunchKernelGGL((k_count_within<D, 1>), grid, dim3(kBlock), lds, c->stream, \
		d_members, K, ndim, thresh_sq, d_res, d_cands, M, d_counts, kchunk, tile_n, accumulate, post); } while (0)
	MDNS_DIM_SWITCH(ndim, COUNT_LAUNCH)
#undef COUNT_LAUNCH
	return launched("k_count_within");
}

// d_packed != nullptr: the caller already holds the choice as one bit per round and point
// (nboot <= kRounds, finishing computation: its slots are zero), so nothing has to be packed.
template <bool NN>
static bool launch_nearest(const double *d_members, int K, int ndim, const double *d_chosen,
                           int nboot, double *d_round_sq, const BootstrapFinish *finish,
                           const unsigned *d_packed = nullptr)
{
	Context *c = ctx();
	const bool small = (K + 63) / 64 < 2 * c->num_cus;          // latency shape below ~32k points
	// partial minima [4 waves][rounds carried][points per workgroup]: exactly what the kernel
	// instantiation uses (the LDS a workgroup asks for bounds the workgroups per CU: 3 at 46 KB)
	const size_t fixed = (size_t) 4 * (NN ? 1 : (nboot > 10 ? kRounds : 10)) * (small ? 16 : 64) * sizeof(double);
	const int tile_n = pick_tile(ndim, sizeof(unsigned), fixed);
	if (tile_n <= 0) { set_error("ndim=%d too large for the member tile", ndim); return false; }
	const size_t lds = (size_t) tile_n * ndim * sizeof(double) + fixed + (size_t) tile_n * sizeof(unsigned);
	unsigned *d_mask = const_cast<unsigned *>(d_packed);
	if (!NN && !d_packed) {
		d_mask = (unsigned *) mask_scratch((size_t) K * sizeof(unsigned));
		if (!d_mask) return false;
	}
	// A finishing computation leaves its slots zeroed (see the kernel), so no launch has to
	// clear them; small pools then also read the choice matrix directly: one launch in all.
	const bool fused = !NN && finish && K <= 2048 && !d_packed;
	const int pts = small ? 16 : 64;
	dim3 grid((K + pts - 1) / pts);
	for (int b0 = 0; b0 < nboot; b0 += kRounds) {
		const int nb = nboot - b0 < kRounds ? nboot - b0 : kRounds;
		if (!NN && !fused && !d_packed) {
			const int nthreads = K > nb ? K : nb;
			hipLaunchKernelGGL(k_pack_chosen, dim3((nthreads + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
			                   d_chosen, K, nboot, b0, nb, d_mask, d_round_sq);
			if (!launched("k_pack_chosen")) return false;
		}
		// only the launch of the last window finishes the computation
		BootstrapFinish fin = {nullptr, nullptr, nullptr, 0};
		if (finish && b0 + kRounds >= nboot) fin = *finish;
		ProfileScope prof(3);
#define NEAR_LAUNCH_R(D, SL, RT) hipLaunchKernelGGL((k_nearest_chosen<D, NN, SL, RT>), grid, dim3(kBlock), lds, c->stream, \
		d_members, K, ndim, d_mask, nb, d_round_sq + b0, tile_n, fin, d_round_sq, nboot, fused ? d_chosen : nullptr, nboot, b0)
#define NEAR_LAUNCH(D) do { \
	if (small) { if (NN || nb <= 10) NEAR_LAUNCH_R(D, 4, 10); else NEAR_LAUNCH_R(D, 4, kRounds); } \
	else       { if (NN || nb <= 10) NEAR_LAUNCH_R(D, 1, 10); else NEAR_LAUNCH_R(D, 1, kRounds); } } while (0)
		MDNS_DIM_SWITCH(ndim, NEAR_LAUNCH)
#undef NEAR_LAUNCH
#undef NEAR_LAUNCH_R
		if (!launched("k_nearest_chosen")) return false;
	}
	return true;
}

bool launch_bootstrap(const double *d_members, int K, int ndim, const double *d_chosen,
                      int nbootstraps, double *d_round_sq, const BootstrapFinish *finish)
{
	return launch_nearest<false>(d_members, K, ndim, d_chosen, nbootstraps, d_round_sq, finish);
}

// grid.y of k_nearest_uniform: member chunks.  Measured (tools/k6_gy_sweep.py, us for the pair of
// kernels at 10 rounds; 1 / 4 / 8 / 16 / 32 / 64 chunks):
//     K =  5 000:  174 /  68 /  55 /  51 /  59 /  62        K = 20 000:  813 / 430 / 407 / 405 / 411 / 418
//     K =  9 000:  302 / 139 / 118 / 110 / 114 / 127        K = 50 000: 3258 / 2331 / 2193 / 2155 / 2194 / 2195
// i.e. many small workgroups (12 500 at 50 000 points) beat the few long ones a model of "fill the
// CUs three deep" picks (it chose 1 chunk at 50 000: 782 workgroups on 256 CUs, 3 217 us): sixteen
// chunks wherever a chunk still has 128 members (fewer is all prologue).  At 50 000 points the pair
// then runs within 12 % of the issue bound of its 28 vector instructions per pair and round set.
static int uniform_chunks(int K, int num_cus)
{
	(void) num_cus;
	int most = K / 128;
	if (most < 1) most = 1;
	int best = most < 16 ? most : 16;
	static const char *forced = getenv("MDNS_K6_GY");                    // experiments only
	if (forced && atoi(forced) > 0) best = atoi(forced) < most ? atoi(forced) : most;
	return best;
}

bool launch_bootstrap_packed(const double *d_members, int K, int ndim, const unsigned *d_packed,
                             int nbootstraps, double *d_round_sq, const BootstrapFinish *finish)
{
	if (nbootstraps > kRounds || !finish) { set_error("packed bootstrap: at most %d rounds, finishing only", kRounds); return false; }
	// pools of a thousand points and more: Morton order + tile culling (mdns_k6sort.hip)
	if (bootstrap_sorted_applies(K, ndim, nbootstraps)) return launch_bootstrap_sorted(d_members, K, ndim, d_packed, nbootstraps, d_round_sq, finish);
	Context *c = ctx();
	static const char *forced = getenv("MDNS_K6_PATH");                   // "classic": the masked-NaN kernel (experiments)
	const bool classic = forced && !strcmp(forced, "classic");
	// (pools of a few hundred points: one launch of the masked-NaN kernel is quicker than two of these)
	if (ndim >= 1 && ndim <= 5 && K >= 640 && !classic) {
		const int rt = nbootstraps <= 10 ? 10 : kRounds;
		const int gy = uniform_chunks(K, c->num_cus);
		int kchunk = (K + gy - 1) / gy;
		kchunk = (kchunk + 3) & ~3;
		const int ny = (K + kchunk - 1) / kchunk;
		const size_t fixed = (size_t) 4 * rt * 64 * sizeof(double);
		int tile_n = pick_tile(ndim, sizeof(unsigned), fixed);
		if (tile_n <= 0) { set_error("ndim=%d too large for the member tile", ndim); return false; }
		if (tile_n > kchunk) tile_n = (kchunk + 15) & ~15;
		const size_t lds = (size_t) tile_n * ndim * sizeof(double) + fixed + (size_t) tile_n * sizeof(unsigned);
		double *d_part = (double *) device_scratch((size_t) ny * K * rt * sizeof(double));
		if (!d_part) return false;
		dim3 grid((K + 63) / 64, ny);
		ProfileScope prof(3);
		note_kernel(3, "k_nearest_uniform<%d, %d>", ndim, rt);
		// MDNS_K6_MERGE=fold: the merge inside the kernel (its last workgroups) instead of k_nearest_finish.
		// With a device-scope fence per workgroup before its ticket (profiles/r04_k6_sorted.txt): 33.9 / 87.9 /
		// 263.6 / 417.6 us at 1 000 / 2 000 / 5 000 / 9 000 points against 20.1 / 24.2 / 50.9 / 113.6 for the pair of
		// kernels.  With the hand-over of mdns_internal.h (agent-scope stores, a wait instead of the fence): 20.4 /
		// 26.2 / 48.1 / 107.1 against 19.1 / 22.3 / 47.6 / 106.3 -- level, so the pair stays the default.
		static const char *merge = getenv("MDNS_K6_MERGE");
		const bool fold = merge && !strcmp(merge, "fold");
		static unsigned *d_tickets = nullptr;
		static int tickets_cap = 0;
		if (fold && (int) grid.x > tickets_cap) {
			if (d_tickets) { (void) hipStreamSynchronize(c->stream); (void) hipFree(d_tickets); d_tickets = nullptr; }
			const int cap = (int) grid.x + 1024;
			if (!MDNS_HIP(hipMalloc((void **) &d_tickets, (size_t) cap * sizeof(unsigned))) ||
			    !MDNS_HIP(hipMemsetAsync(d_tickets, 0, (size_t) cap * sizeof(unsigned), c->stream))) { d_tickets = nullptr; tickets_cap = 0; return false; }
			tickets_cap = cap;
		}
		unsigned *tickets = fold ? d_tickets : nullptr;
#define UNI_LAUNCH(D) do { if (rt == 10) hipLaunchKernelGGL((k_nearest_uniform<D, 10>), grid, dim3(kBlock), lds, c->stream, d_members, K, d_packed, kchunk, tile_n, d_part, tickets, nbootstraps, d_round_sq, *finish, nbootstraps); \
		else hipLaunchKernelGGL((k_nearest_uniform<D, kRounds>), grid, dim3(kBlock), lds, c->stream, d_members, K, d_packed, kchunk, tile_n, d_part, tickets, nbootstraps, d_round_sq, *finish, nbootstraps); } while (0)
		switch (ndim) { case 1: UNI_LAUNCH(1); break; case 2: UNI_LAUNCH(2); break; case 3: UNI_LAUNCH(3); break;
		                case 4: UNI_LAUNCH(4); break; default: UNI_LAUNCH(5); break; }
#undef UNI_LAUNCH
		if (!launched("k_nearest_uniform")) return false;
		if (fold) return true;
		const dim3 fgrid((K + kBlock / 4 - 1) / (kBlock / 4));
		if (rt == 10) hipLaunchKernelGGL((k_nearest_finish<10>), fgrid, dim3(kBlock), 0, c->stream, (const double *) d_part, K, ny, d_packed, nbootstraps, d_round_sq, *finish, nbootstraps);
		else hipLaunchKernelGGL((k_nearest_finish<kRounds>), fgrid, dim3(kBlock), 0, c->stream, (const double *) d_part, K, ny, d_packed, nbootstraps, d_round_sq, *finish, nbootstraps);
		return launched("k_nearest_finish");
	}
	return launch_nearest<false>(d_members, K, ndim, nullptr, nbootstraps, d_round_sq, finish, d_packed);
}

bool launch_nn_maxsq(const double *d_members, int K, int ndim, double *d_out)
{
	return launch_nearest<true>(d_members, K, ndim, nullptr, 1, d_out, nullptr);
}

}  // namespace mdns
