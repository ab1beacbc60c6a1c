// The floating-point state of the joint sampler on the device, and the constrained draw that is
// decided there (include/mdns.h, Part 2b).
//
// The reference keeps, per data set, the likelihoods of its live points (`live_pointsL[nlive,
// ndata]`, multi_nested_sampler.py:111) and of the accepted points waiting on its shelf (:117),
// and derives from them, for every draw, the thresholds `Lmins_higher` (:438-447).  Here all of
// that lives in HBM, one column per data set (lane = data set: every access is coalesced over
// the data sets), indexed by the ORIGINAL data-set index for the whole run:
//
//   live   [nlive][ndata]   slot p of data set d at p * ndata + d
//   shelfL [cap][ndata]     FIFO, entry e at e * ndata + d, shelfn[d] entries
//   higher [ndata]          threshold of the next draw: with n entries waiting, the (n+1)-th
//                           smallest of live + shelf (find_nsmallest, :44-47)
//
// Thresholds are order statistics of ~100 numbers; they are always computed from the two
// arrays above (no auxiliary sorted lists that could go out of step): by repeated
// "smallest value above the previous one" passes at the start of an iteration, and by ONE such
// pass when an accepted point joins a shelf (k_gauss_cols_commit in mdns_like.hip).
#include "mdns_internal.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

namespace mdns {

static constexpr int kBlock = 256;
static constexpr int kFlagInts = MDNS_JOINT_MAX_BATCH;            // accept flags, one int per candidate
static constexpr int kZeroInts = kFlagInts + (int) (sizeof(JointHeader) / sizeof(int));

// Start of an iteration, one lane per running data set (multi_nested_sampler.py:130-143):
// lowest live likelihood and its slot (first occurrence, as numpy.argmin), shelf entries that
// do not beat it dropped in order, threshold for what is still waiting.
//
// SIXTEEN lanes per data set, four data sets per wave (lane = 4 slice + data set: the four data
// sets of a wave are neighbours in `running`, so a load touches 16 rows x 32 B): slice s holds live
// slots and shelf entries s, s + 16, ... in registers, and everything that concerns one data set
// -- minimum, purge, threshold -- is settled among its sixteen lanes with cross-lane operations:
// no LDS, no barrier, and a data set that is done only waits for the other three of its wave.
// History, per iteration of a real run whose shelves hold tens of entries (rocprofv3, first 600
// iterations of C2; one lane per data set and slice, 64 data sets x 4 slices per workgroup):
// walking up the distinct values with one pass over the column per waiting entry 416 us; sorted
// lists of the 16-24 smallest per slice in LDS, merged in rounds, 62 us (the insertions are chains
// of dependent LDS accesses); quickselect on counts with the values in registers 95 us (two
// barriers and three scans of 64 registers per round); this form: see profiles/.
static constexpr int kSlices = 16, kHeld = 8;                  // kHeld values per lane in registers: 128 slots / entries

// reductions over the sixteen lanes of a data set (lanes q, q + 4, ..., q + 60)
template <typename T, typename F> __device__ __forceinline__ T over_slices(T v, F f)
{
	v = f(v, __shfl_xor(v, 4));
	v = f(v, __shfl_xor(v, 8));
	v = f(v, __shfl_xor(v, 16));
	v = f(v, __shfl_xor(v, 32));
	return v;
}

__global__ __launch_bounds__(kBlock) void k_joint_prepare(JointArrays st, const int *__restrict__ running, int nrun,
                                                          double *__restrict__ Lmin, int *__restrict__ argmin_run,
                                                          int *__restrict__ argmin, unsigned long long *__restrict__ keep,
                                                          int keep_words)
{
	const int lane = threadIdx.x & 63;
	const int q = lane & 3, sl = lane >> 2;
	const int r = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 4 + q;
	const bool real = r < nrun;
	const int d = running[real ? r : nrun - 1];
	const size_t nd = (size_t) st.ndata;
	// live slots sl, sl + 16, ...: all loads in flight at once; a NaN stands for "no such slot" (it
	// is never the minimum and never counted); slots beyond 128 are read again where needed
	double lv[kHeld];
#pragma unroll
	for (int u = 0; u < kHeld; u++) {
		const int p = sl + kSlices * u;
		lv[u] = st.live[(size_t) (p < st.nlive ? p : 0) * nd + d];
	}
	const int n0 = st.shelfn[d];
	double sv[kHeld];                                            // shelf entries sl, sl + 16, ...
#pragma unroll
	for (int u = 0; u < kHeld; u++) {
		const int e = sl + kSlices * u;
		sv[u] = st.shelfL[(size_t) (e < n0 ? e : 0) * nd + d];
	}
	double m = INFINITY;
	int am = 0x7fffffff;
#pragma unroll
	for (int u = 0; u < kHeld; u++) if (sl + kSlices * u >= n0) sv[u] = __builtin_nan("");
#pragma unroll
	for (int u = 0; u < kHeld; u++) {
		const int p = sl + kSlices * u;
		if (p >= st.nlive) lv[u] = __builtin_nan("");
		if (lv[u] < m) { m = lv[u]; am = p; }
	}
	for (int p = sl + kSlices * kHeld; p < st.nlive; p += kSlices) { const double v = st.live[(size_t) p * nd + d]; if (v < m) { m = v; am = p; } }
	// the minimum and its first slot (numpy.argmin)
	{
		const double mm = over_slices(m, [](double x, double y) { return y < x ? y : x; });
		am = over_slices(m == mm ? am : 0x7fffffff, [](int x, int y) { return y < x ? y : x; });
		m = mm;
		if (am == 0x7fffffff) am = 0;                            // a column of NaNs: numpy.argmin's answer is moot
	}
	// purge (multi_nested_sampler.py:137-138: keep entries with L > Lmin, order kept): the sixteen
	// lanes take sixteen consecutive entries; an entry moves down by the number of dropped ones
	// before it.  (A batch is read whole before any of it is written, and kept entries only move
	// down, to slots of this or an earlier batch.)
	const unsigned long long mine = 0x1111111111111111ull << q;   // the lanes of this data set
	const unsigned long long before = mine & ((1ull << lane) - 1ull);
	int w = 0;                                                     // kept so far (the same in all sixteen lanes)
	int w_held = -1;                                               // ... when the entries held in registers were through
	unsigned long long word = 0;                                   // this lane's keep bits of the current 64 entries
	for (int e0 = 0; e0 < keep_words * 64; e0 += kSlices) {        // (wave-uniform bounds: n0 differs between the four data sets)
		const int e = e0 + sl;
		const int u = e0 / kSlices;
		if (u == kHeld) w_held = w;
		double v;
		if (u < kHeld) {
			v = sv[0];
#pragma unroll
			for (int k = 1; k < kHeld; k++) if (u == k) v = sv[k];
		} else v = st.shelfL[(size_t) (e < n0 ? e : 0) * nd + d];
		const bool kept = real && e < n0 && v > m;
		const unsigned long long votes = __ballot(kept) & mine;
		const int at = w + __popcll(votes & before);
		if (kept && at != e) st.shelfL[(size_t) at * nd + d] = v;
		if (kept) word |= 1ull << (e & 63);
		w += __popcll(votes);
		// what this lane keeps for the selection below: its own entries that were kept
		if (u < kHeld) {
#pragma unroll
			for (int k = 0; k < kHeld; k++) if (u == k && !kept) sv[k] = __builtin_nan("");
		}
		if (((e0 + kSlices) & 63) == 0) {
			const unsigned long long all = over_slices(word, [](unsigned long long x, unsigned long long y) { return x | y; });
			if (real && sl == 0) keep[(size_t) r * keep_words + (e0 >> 6)] = all;
			word = 0;
		}
	}
	if (real && sl == 0) st.shelfn[d] = w;
	if (w_held < 0) w_held = w;
	// (shelves longer than the registers hold: the selection below reads entries the purge above has
	// just moved -- stores of other lanes of this wave -- from memory: make them visible first)
	if (n0 > kSlices * kHeld) __threadfence();
	// The (w+1)-th smallest of live + shelf (find_nsmallest, :44-47) by quickselect on counts: the
	// answer is the smallest value x with at least w + 1 values at or below it.  It is known to lie
	// in (lo, hi]; a round takes a pivot strictly inside -- every lane proposes the middle one, in
	// slot order, of its own values in there, and the lane with most of them wins -- counts the
	// values at or below the pivot and moves one end.  About log2(100 + w) rounds settle it.
	// (Entries beyond the 128 held in registers are read from the purged shelf, where they follow
	// the w_held kept ones of the first 128.)
	double thr = m;                                                // nothing waits: the threshold is the minimum
	bool done = !real || w == 0;
	const int target = w + 1;
	const bool far = n0 > kSlices * kHeld || st.nlive > kSlices * kHeld;   // values beyond the registers
	double lo = __builtin_nan(""), hi = __builtin_nan("");        // NaN: no bound on that side yet
	while (__any(!done)) {
		int inside = 0;
		double proposal = 0.0;
		auto in_range = [&](double v) { return v == v && !(v <= lo) && !(v >= hi); };
#pragma unroll
		for (int u = 0; u < kHeld; u++) inside += in_range(lv[u]);
#pragma unroll
		for (int u = 0; u < kHeld; u++) inside += in_range(sv[u]);
		if (far) {
			for (int p = sl + kSlices * kHeld; p < st.nlive; p += kSlices) inside += in_range(st.live[(size_t) p * nd + d]);
			for (int e = w_held + sl; e < w; e += kSlices) inside += in_range(st.shelfL[(size_t) e * nd + d]);
		}
		{
			int countdown = inside / 2;                            // the middle one in slot order
			auto pick = [&](double v) { if (in_range(v) && countdown-- == 0) proposal = v; };
#pragma unroll
			for (int u = 0; u < kHeld; u++) pick(lv[u]);
#pragma unroll
			for (int u = 0; u < kHeld; u++) pick(sv[u]);
			if (far) {
				for (int p = sl + kSlices * kHeld; p < st.nlive; p += kSlices) pick(st.live[(size_t) p * nd + d]);
				for (int e = w_held + sl; e < w; e += kSlices) pick(st.shelfL[(size_t) e * nd + d]);
			}
		}
		// the proposal of the lane with most values inside (ties: the lowest slice)
		const int most = over_slices(inside, [](int x, int y) { return y > x ? y : x; });
		const int from = over_slices(inside == most ? sl : kSlices, [](int x, int y) { return y < x ? y : x; });
		const double pivot = __shfl(proposal, 4 * from + q);
		if (!done && most == 0) {
			// nothing strictly between: hi it is -- or there are fewer than w + 1 comparable values (NaNs)
			thr = hi == hi ? hi : INFINITY;
			done = true;
		}
		int count = 0;
#pragma unroll
		for (int u = 0; u < kHeld; u++) count += lv[u] <= pivot;  // (a NaN is never counted)
#pragma unroll
		for (int u = 0; u < kHeld; u++) count += sv[u] <= pivot;
		if (far) {
			for (int p = sl + kSlices * kHeld; p < st.nlive; p += kSlices) count += st.live[(size_t) p * nd + d] <= pivot;
			for (int e = w_held + sl; e < w; e += kSlices) count += st.shelfL[(size_t) e * nd + d] <= pivot;
		}
		const int total = over_slices(count, [](int x, int y) { return x + y; });
		if (!done) { if (total >= target) hi = pivot; else lo = pivot; }
	}
	if (!real || sl != 0) return;
	st.higher[d] = thr;
	Lmin[r] = m;
	argmin_run[r] = am;
	argmin[d] = am;
}

// End of an iteration (multi_nested_sampler.py:494-534): the worst live point of every running
// data set is replaced by the head of its shelf.
__global__ __launch_bounds__(kBlock) void k_joint_advance(JointArrays st, const int *__restrict__ running, int nrun,
                                                          const int *__restrict__ argmin, int *__restrict__ status)
{
	const int r = blockIdx.x * kBlock + threadIdx.x;
	if (r >= nrun) return;
	const int d = running[r];
	const size_t nd = (size_t) st.ndata;
	const int n = st.shelfn[d];
	if (n <= 0) { atomicOr(status, 1); return; }
	st.live[argmin[d] * nd + d] = st.shelfL[d];
	for (int e = 1; e < n; e++) st.shelfL[(e - 1) * nd + d] = st.shelfL[e * nd + d];
	st.shelfn[d] = n - 1;
}

// Takes back the last advance: the slot every running data set gave to its shelf head gets the
// likelihood back that prepare found there, and the shelves are emptied.  (bench.py re-runs the
// same iteration; 10 000 rewritten entries instead of an 8 MB copy.)
__global__ __launch_bounds__(kBlock) void k_joint_undo_advance(JointArrays st, const int *__restrict__ running, int nrun,
                                                               const int *__restrict__ argmin, const double *__restrict__ Lmin)
{
	const int r = blockIdx.x * kBlock + threadIdx.x;
	if (r >= nrun) return;
	const int d = running[r];
	st.live[(size_t) argmin[d] * st.ndata + d] = Lmin[r];
	st.shelfn[d] = 0;
}

// Runs right behind the commit pass (stream order makes its results visible): one workgroup
// copies {accepted, status} and the fill words into the mapped block and writes `seq` last.
// (Doing this in the commit kernel itself -- last workgroup to finish -- needs a device- or
// system-scope fence in every workgroup, i.e. an L2 write-back each: 16 -> 32 us, measured.)
__global__ __launch_bounds__(kBlock) void k_joint_publish(const JointHeader *__restrict__ header,
                                                          const unsigned long long *__restrict__ fillbits, int nwords,
                                                          JointMailbox *__restrict__ box, unsigned long long seq)
{
	const int accepted = header->accepted;
	if (accepted >= 0)
		for (int w = threadIdx.x; w < nwords; w += kBlock) mail_store(&box->bits[w], fillbits[w]);
	handover_release();
	__syncthreads();
	if (threadIdx.x != 0) return;
	mail_store(&box->accepted, accepted);
	mail_store(&box->status, header->status);
	mail_raise(&box->seq, seq);
}

// ---- the draw for scorers that leave a dense L[B, M] (the scale-marginalised likelihood, K2) ----
// L += jitter (musefuse.py:535), then the accept test `any(L > Lmins)` (hiermetriclearn.py:193): one
// thread per (candidate, selected data set); an accepted candidate gets flags[b] = flag (every
// writer stores the same value)
__global__ __launch_bounds__(kBlock) void k_joint_accept_dense(double *__restrict__ L, const double *__restrict__ jitter, int B, int M,
                                                               const int *__restrict__ thr_rows, const double *__restrict__ higher,
                                                               int *__restrict__ flags, int flag, JointHeader *__restrict__ header)
{
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) header->status = 0;
	const int k = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
	if (k >= M) return;
	const size_t at = (size_t) b * M + k;
	double v = L[at];
	if (jitter) { v = v + jitter[at]; L[at] = v; }
	const int d = thr_rows ? thr_rows[k] : k;
	if (v > higher[d]) flags[b] = flag;
}

// first flagged candidate -> shelf appends, next thresholds, fill bits (k_joint_commit_trail with the
// likelihoods read from the dense block)
__global__ __launch_bounds__(kBlock) void k_joint_commit_dense(
    const double *__restrict__ L, const int *__restrict__ thr_rows, int M, int B, int ntiles, const int *__restrict__ flags, int flag,
    JointArrays st, JointHeader *__restrict__ header, unsigned long long *__restrict__ fillbits)
{
	__shared__ int s_first;
	if (threadIdx.x == 0) s_first = 0x7fffffff;
	__syncthreads();
	for (int b = threadIdx.x; b < B; b += kBlock)
		if (flags[b] == flag) { atomicMin(&s_first, b); break; }
	__syncthreads();
	const int bstar = s_first;
	if (blockIdx.x == 0 && threadIdx.x == 0) header->accepted = bstar < B ? bstar : -1;
	if (bstar >= B) return;
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (tile >= ntiles) return;
	const int k = tile * 64 + lane;
	bool beats = false;
	if (k < M) {
		const int d = thr_rows ? thr_rows[k] : k;
		const double v = L[(size_t) bstar * M + k];
		const double thr = st.higher[d];
		beats = v > thr;
		if (beats) {
			const int n = st.shelfn[d];
			if (n >= st.cap) atomicOr(&header->status, 1);
			else {
				int at_most = 0;
				double next = INFINITY;
				for (int p = 0; p < st.nlive; p++) {
					const double w = st.live[(size_t) p * st.ndata + d];
					if (w <= thr) at_most++; else next = fmin(next, w);
				}
				for (int e = 0; e < n; e++) {
					const double w = st.shelfL[(size_t) e * st.ndata + d];
					if (w <= thr) at_most++; else next = fmin(next, w);
				}
				st.shelfL[(size_t) n * st.ndata + d] = v;
				st.shelfn[d] = n + 1;
				st.higher[d] = at_most >= n + 2 ? thr : fmin(v, next);
			}
		}
	}
	const unsigned long long word = __ballot(beats);
	if (lane == 0) fillbits[tile] = word;
}

// ---- the likelihood noise in band form (mdns.h: draw_band / draw_band_commit) ----
// every likelihood against its threshold +- (1.01 bound[b] + 1e-12 (|L| + |thr|)): a pair above the band
// is beaten whatever the noise (clear[b] = 1), a pair inside it is listed for the host, which alone
// makes the exact deviates
// (BandBox, BandScratch, band_vote, band_publish: mdns_internal.h -- the small-chunk row kernel of mdns_like.hip votes and
// publishes by itself)
// box != nullptr: the last workgroup to finish publishes (votes and pairs go through agent-scope stores: the
// hand-over of mdns_internal.h) -- one launch less per chunk than k_joint_band_publish behind it
__global__ __launch_bounds__(kBlock) void k_joint_band(const double *__restrict__ L, const double *__restrict__ bound, int B, int M,
                                                       const int *__restrict__ thr_rows, const double *__restrict__ higher,
                                                       BandScratch *__restrict__ sc, JointHeader *__restrict__ header,
                                                       BandBox *__restrict__ box, unsigned long long seq)
{
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) header->status = 0;
	const int k = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
	if (k < M) {
		const int d = thr_rows ? thr_rows[k] : k;
		band_vote(sc, b, k, L[(size_t) b * M + k], higher[d], bound[b]);
	}
	if (!box) return;
	__shared__ int s_last;
	handover_release();
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(&sc->ticket, 1) == (int) (gridDim.x * gridDim.y) - 1 ? 1 : 0;
	__syncthreads();
	if (!s_last) return;
	handover_acquire();
	band_publish(sc, B, box, seq);
}

// (behind the matrix-core filter, whose workgroups do not count themselves)
__global__ __launch_bounds__(kBlock) void k_joint_band_publish(BandScratch *__restrict__ sc, int B, BandBox *__restrict__ box, unsigned long long seq)
{
	band_publish(sc, B, box, seq);
}

// k_joint_commit_dense for a candidate the HOST names, with its noise row added first
__global__ __launch_bounds__(kBlock) void k_joint_commit_band(
    const double *__restrict__ L, const double *__restrict__ jrow, const int *__restrict__ thr_rows, int M, int bstar, int ntiles,
    JointArrays st, JointHeader *__restrict__ header, unsigned long long *__restrict__ fillbits)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) header->accepted = bstar;
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (tile >= ntiles) return;
	const int k = tile * 64 + lane;
	bool beats = false;
	if (k < M) {
		const int d = thr_rows ? thr_rows[k] : k;
		const double v = L[(size_t) bstar * M + k] + jrow[k];
		const double thr = st.higher[d];
		beats = v > thr;
		if (beats) {
			const int n = st.shelfn[d];
			if (n >= st.cap) atomicOr(&header->status, 1);
			else {
				int at_most = 0;
				double next = INFINITY;
				int p = 0;
				for (; p + 20 <= st.nlive; p += 20) {                   // 20 loads in flight (as in k_joint_commit_trail: the latency of a round trip is what this costs)
					double w[20];
#pragma unroll
					for (int u = 0; u < 20; u++) w[u] = st.live[(size_t) (p + u) * st.ndata + d];
#pragma unroll
					for (int u = 0; u < 20; u++) { if (w[u] <= thr) at_most++; else next = fmin(next, w[u]); }
				}
				for (; p < st.nlive; p++) {
					const double w = st.live[(size_t) p * st.ndata + d];
					if (w <= thr) at_most++; else next = fmin(next, w);
				}
				for (int e = 0; e < n; e++) {
					const double w = st.shelfL[(size_t) e * st.ndata + d];
					if (w <= thr) at_most++; else next = fmin(next, w);
				}
				st.shelfL[(size_t) n * st.ndata + d] = v;
				st.shelfn[d] = n + 1;
				st.higher[d] = at_most >= n + 2 ? thr : fmin(v, next);
			}
		}
	}
	const unsigned long long word = __ballot(beats);
	if (lane == 0) fillbits[tile] = word;
}

__global__ void k_joint_add(double *__restrict__ p, const double *__restrict__ q, size_t n)
{
	for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t) gridDim.x * blockDim.x) p[e] = p[e] + q[e];
}

__global__ void k_joint_fill(double *__restrict__ p, size_t n, double value)
{
	for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t) gridDim.x * blockDim.x) p[e] = value;
}

}  // namespace mdns

using namespace mdns;

struct mdns_joint {
	mdns_spectra *s = nullptr;
	int nlive = 0, cap = 0, ndata = 0;
	JointArrays st = {};
	int *d_running = nullptr;  int nrun = 0;
	double *d_Lmin = nullptr;          // [ndata] by position in the running list
	int *d_argmin_run = nullptr;       // [ndata] by position in the running list
	int *d_argmin = nullptr;           // [ndata] by data set
	unsigned long long *d_keep = nullptr;  size_t keep_cap = 0;
	int *d_status = nullptr;           // sticky failure bits of advance
	// one allocation: accept flags | header | fill bits | likelihood row  (the model kernel of a
	// chunk clears flags + header; header onward is what the host reads)
	int *d_flags = nullptr;
	char *d_result = nullptr;          // = (char *) (d_flags + kFlagInts)
	// the outcome of a commit without the likelihood row: mapped host memory a kernel behind the
	// commit pass writes and the host polls (JointMailbox)
	JointMailbox *h_box = nullptr, *h_box_dev = nullptr;
	unsigned long long box_seq = 0;
	bool box_pending = false;          // a commit was launched whose mailbox has not been read
	// the trail of the accept pass (JointTrail): grown to candidates x tiles of the largest chunk
	int *d_trail_stamp = nullptr;  unsigned long long *d_trail_word = nullptr;  double *d_trail_L = nullptr;
	size_t trail_cap = 0;              // entries (candidate, tile)
	int trail_stamp = 0;
	bool trail_valid = false;          // the last score left a trail for its chunk
	// staging of the host-pointer draw
	double *d_params = nullptr;
	int *d_rows = nullptr;             // inside d_params' block, behind the candidates
	char *h_pin = nullptr;  size_t pin_bytes = 0;
	// what the last score launched with (commit uses the same spectra replica and templates)
	const double *last_yT = nullptr;
	const int *last_gather = nullptr;
	int last_bt = 0, last_B = 0;
	double last_scale = 0;
	bool prepared = false;
	// what mdns_joint_score staged for the commit that follows
	bool staged_rows = false;
	bool staged_valid = false;                              // a host-pointer score is waiting for its commit
	int scored_M = -1;                                      // selection size of the score whose flags / trail are in place (-1: none)
	int staged_M = 0;
	size_t staged_in_bytes = 0;
	double noise_level = 0;            // of mdns_joint_init_gauss (the backend entry points score with it)
	// which likelihood: 0 = the Gaussian line (clike.c; lane kernels, trail), 1 = the three-line
	// template scored with the scale-marginalised chi^2 (cmuselike.c; dense L[B, M] block)
	int kind = 0;
	int nparams = 3;
	double *d_msq = nullptr;                                // templates' sums of squares (guarded accept filter, kind 0)
	int *d_filter_scratch = nullptr;                        // matrix-core filter: ambiguous marks
	double *d_dense = nullptr;  size_t dense_cap = 0;       // L[B, M] of a chunk (kind 1)
	double *d_jitter = nullptr;  size_t jitter_cap = 0;
	// the draw in progress through the mdns_backend_* entry points: its selection, uploaded once
	int *d_sel_rows = nullptr;  size_t sel_rows_cap = 0;
	bool sel_rows = false;
	int sel_M = 0;
	bool sel_open = false;
	bool sel_on_device = false;        // d_sel_rows holds the selection (a chunk kernel or a copy put it there)
	// accepted candidates of a two-launch chunk are flagged with the chunk's own number (never reused,
	// never 1 -- the flag value of the other path -- so the flag buffer needs no clearing)
	int chunk_seq = 1;
	// what the chunk kernels read directly: candidates, then the selection's row ids, in host
	// memory mapped into the device
	char *h_in = nullptr, *h_in_dev = nullptr;
	// no shelf holds more than this many entries (set by prepare from the purge's keep bits, +1 per
	// accepted chunk, -1 per advance): when it reaches the capacity the shelves are grown
	int shelf_bound = 0;
	// the first batch of a region without a host look in between (mdns_backend_chain_*, mdns_chain.hip)
	ChainBox *h_chain = nullptr, *h_chain_dev = nullptr;   // mapped
	double *d_chain_props = nullptr;
	int *d_chain_counts = nullptr, *d_chain_ticket = nullptr;
	int *d_commit_ticket = nullptr;    // workgroups of a commit pass that are done (k_joint_commit_trail publishing by itself)
	// the likelihood noise in band form (mdns_backend_draw_band / _commit)
	BandBox *h_band = nullptr, *h_band_dev = nullptr;
	BandScratch *d_band = nullptr;
	double *d_bound = nullptr;                // [MDNS_JOINT_MAX_BATCH] bounds, then room for one noise row [ndata]
	unsigned long long band_seq = 0;
	int band_B = 0;
	bool band_pending = false;        // a chunk begun and not collected (mdns_backend_draw_band_begin / _end)
	bool band_exact = true;           // d_dense holds the exact likelihoods of the chunk (not the matrix-core filter's)
	// a chunk in two halves (mdns_backend_draw_score / _commit): one 0 / 1 vote per candidate, what the ranks
	// of a sharded run MAX-reduce in between
	int *d_votes = nullptr;
	int half_path = 0, half_B = 0, half_flag = 0;       // 0: none; 1: dense block (kind 1); 2: two launches; 3: lane kernels
	unsigned long long chain_seq = 0;
	int chain_state = 0;               // 0 none, 1 counts only (poll h_chain->seq), 2 full (poll the commit's mailbox)
	int chain_n = 0;
};

static size_t result_bytes(int M) { return sizeof(JointHeader) + (size_t) ((M + 63) / 64) * 8 + (size_t) M * 8; }
extern "C" size_t mdns_joint_result_bytes(int M) { return result_bytes(M > 0 ? M : 0); }
static int keep_words_of(int cap) { return (cap + 63) / 64; }

static bool joint_sync(Context *c) { return MDNS_HIP(hipStreamSynchronize(c->stream)); }

extern "C" void mdns_joint_destroy(mdns_joint *j)
{
	if (!j) return;
	Context *c = ctx();
	if (c) (void) hipStreamSynchronize(c->stream);
	void *bufs[] = {j->st.live, j->st.shelfL, j->st.shelfn, j->st.higher, j->d_running, j->d_Lmin, j->d_argmin_run,
	                j->d_argmin, j->d_keep, j->d_status, j->d_flags, j->d_params};
	for (void *b : bufs) if (b) (void) hipFree(b);
	if (j->d_sel_rows) (void) hipFree(j->d_sel_rows);
	if (j->d_dense) (void) hipFree(j->d_dense);
	if (j->d_msq) (void) hipFree(j->d_msq);
	if (j->d_filter_scratch) (void) hipFree(j->d_filter_scratch);
	if (j->d_jitter) (void) hipFree(j->d_jitter);
	if (j->h_in) (void) hipHostFree(j->h_in);
	if (j->h_chain) (void) hipHostFree(j->h_chain);
	if (j->d_chain_props) (void) hipFree(j->d_chain_props);
	if (j->d_chain_counts) (void) hipFree(j->d_chain_counts);
	if (j->d_chain_ticket) (void) hipFree(j->d_chain_ticket);
	if (j->d_commit_ticket) (void) hipFree(j->d_commit_ticket);
	if (j->d_votes) (void) hipFree(j->d_votes);
	if (j->h_band) (void) hipHostFree(j->h_band);
	if (j->d_band) (void) hipFree(j->d_band);
	if (j->d_bound) (void) hipFree(j->d_bound);
	void *trail[] = {j->d_trail_stamp, j->d_trail_word, j->d_trail_L};
	for (void *b : trail) if (b) (void) hipFree(b);
	if (j->h_box) (void) hipHostFree(j->h_box);
	if (j->h_pin) (void) hipHostFree(j->h_pin);
	delete j;
}

extern "C" mdns_joint *mdns_joint_create(mdns_spectra *s, int nlive, int shelf_cap)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (!s || nlive <= 0 || s->ndata <= 0) { set_error("mdns_joint_create: bad arguments (nlive=%d)", nlive); return nullptr; }
	if (!s->d_x || (!s->d_yT && !s->d_w)) { set_error("mdns_joint_create: the spectra need a wavelength grid"); return nullptr; }
	if (shelf_cap < 4) shelf_cap = 4;
	mdns_joint *j = new mdns_joint();
	j->s = s; j->nlive = nlive; j->cap = shelf_cap; j->ndata = s->ndata;
	// spectra with variances: the scale-marginalised likelihood against the three-line template
	j->kind = s->d_w ? 1 : 0;
	j->nparams = j->kind == 1 ? 5 : 3;
	const size_t nd = (size_t) s->ndata;
	const size_t res = (size_t) kFlagInts * sizeof(int) + result_bytes(s->ndata);
	bool ok =
	    MDNS_HIP(hipMalloc((void **) &j->st.live, (size_t) nlive * nd * sizeof(double))) &&
	    MDNS_HIP(hipMalloc((void **) &j->st.shelfL, (size_t) shelf_cap * nd * sizeof(double))) &&
	    MDNS_HIP(hipMalloc((void **) &j->st.shelfn, nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &j->st.higher, nd * sizeof(double))) &&
	    MDNS_HIP(hipMalloc((void **) &j->d_running, nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &j->d_Lmin, nd * sizeof(double))) &&
	    MDNS_HIP(hipMalloc((void **) &j->d_argmin_run, nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &j->d_argmin, nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &j->d_status, sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &j->d_flags, res)) &&
	    // candidates [B, 3] followed by the selection's row ids: one staging block
	    MDNS_HIP(hipMalloc((void **) &j->d_params, (size_t) MDNS_JOINT_MAX_BATCH * 5 * sizeof(double) + nd * sizeof(int) + 16)) &&
	    MDNS_HIP(hipHostMalloc((void **) &j->h_box, sizeof(JointMailbox) + ((nd + 63) / 64) * 8, hipHostMallocMapped | hipHostMallocCoherent)) &&
	    MDNS_HIP(hipHostGetDevicePointer((void **) &j->h_box_dev, j->h_box, 0));
	if (ok) {
		memset(j->h_box, 0, sizeof(JointMailbox) + ((nd + 63) / 64) * 8);
		j->d_result = (char *) (j->d_flags + kFlagInts);
		j->st.nlive = nlive; j->st.cap = shelf_cap; j->st.ndata = s->ndata;
		std::vector<int> all(nd);
		for (size_t i = 0; i < nd; i++) all[i] = (int) i;
		ok = MDNS_HIP(hipMemsetAsync(j->st.shelfn, 0, nd * sizeof(int), c->stream)) &&
		     MDNS_HIP(hipMemsetAsync(j->d_status, 0, sizeof(int), c->stream)) &&
		     MDNS_HIP(hipMemsetAsync(j->d_flags, 0, res, c->stream)) &&
		     MDNS_HIP(hipMemsetAsync(j->d_argmin, 0, nd * sizeof(int), c->stream)) &&
		     MDNS_HIP(hipMemcpyAsync(j->d_running, all.data(), nd * sizeof(int), hipMemcpyHostToDevice, c->stream));
		if (ok) {
			hipLaunchKernelGGL(k_joint_fill, dim3(256), dim3(kBlock), 0, c->stream, j->st.higher, nd, (double) NAN);
			ok = MDNS_HIP(hipGetLastError()) && joint_sync(c);
		}
		j->nrun = s->ndata;
	}
	if (!ok) { mdns_joint_destroy(j); return nullptr; }
	return j;
}

static char *joint_pin(mdns_joint *j, size_t bytes)
{
	if (bytes <= j->pin_bytes) return j->h_pin;
	Context *c = ctx();
	if (j->h_pin) { (void) hipStreamSynchronize(c->stream); (void) hipHostFree(j->h_pin); j->h_pin = nullptr; j->pin_bytes = 0; }
	const size_t want = bytes + bytes / 2 + 4096;
	if (!MDNS_HIP(hipHostMalloc((void **) &j->h_pin, want, hipHostMallocDefault))) return nullptr;
	j->pin_bytes = want;
	return j->h_pin;
}

extern "C" int mdns_joint_shelf_cap(const mdns_joint *j) { return j ? j->cap : -1; }
extern "C" int mdns_joint_keep_words(const mdns_joint *j) { return j ? keep_words_of(j->cap) : -1; }

extern "C" int mdns_joint_reserve(mdns_joint *j, int shelf_cap)
{
	Context *c = ctx();
	if (!c || !j) return 1;
	if (shelf_cap <= j->cap) return 0;
	int cap = j->cap;
	while (cap < shelf_cap) cap *= 2;
	const size_t nd = (size_t) j->ndata;
	double *bigger = nullptr;
	if (!MDNS_HIP(hipMalloc((void **) &bigger, (size_t) cap * nd * sizeof(double)))) return 1;
	// entry-major layout: the old array is a prefix of the new one
	if (!MDNS_HIP(hipMemcpyAsync(bigger, j->st.shelfL, (size_t) j->cap * nd * sizeof(double), hipMemcpyDeviceToDevice, c->stream)) ||
	    !joint_sync(c)) { (void) hipFree(bigger); return 1; }
	(void) hipFree(j->st.shelfL);
	j->st.shelfL = bigger;
	j->cap = j->st.cap = cap;
	return 0;
}

static int joint_reset(mdns_joint *j)
{
	Context *c = ctx();
	j->prepared = false;
	j->shelf_bound = 0;
	return MDNS_HIP(hipMemsetAsync(j->st.shelfn, 0, (size_t) j->ndata * sizeof(int), c->stream)) &&
	       MDNS_HIP(hipMemsetAsync(j->d_status, 0, sizeof(int), c->stream)) ? 0 : 1;
}

extern "C" int mdns_joint_init_gauss(mdns_joint *j, const double *params, double noise_level)
{
	Context *c = ctx();
	if (!c || !j || !params) return 1;
	if (j->kind != 0) { set_error("mdns_joint_init_gauss: these spectra carry variances (mdns_joint_init_muse3)"); return 1; }
	if (j->nlive > MDNS_JOINT_MAX_BATCH) { set_error("mdns_joint_init_gauss: nlive=%d > %d", j->nlive, MDNS_JOINT_MAX_BATCH); return 1; }
	char *pin = joint_pin(j, (size_t) j->nlive * 24);
	if (!pin) return 1;
	j->noise_level = noise_level;
	memcpy(pin, params, (size_t) j->nlive * 24);
	if (!MDNS_HIP(hipMemcpyAsync(j->d_params, pin, (size_t) j->nlive * 24, hipMemcpyHostToDevice, c->stream))) return 1;
	// always the lane kernel: every likelihood of a run is then the same chain of operations
	if (gauss_loglike_cols_dev(j->s, j->d_params, j->nlive, noise_level, nullptr, j->ndata, j->st.live) != 0) return 1;
	if (joint_reset(j) != 0) return 1;
	return joint_sync(c) ? 0 : 1;
}

static bool joint_grow(double **p, size_t *cap, size_t need)
{
	if (need <= *cap) return true;
	Context *c = ctx();
	if (*p) { (void) hipStreamSynchronize(c->stream); (void) hipFree(*p); *p = nullptr; *cap = 0; }
	const size_t n = need + need / 2 + 1024;
	if (!MDNS_HIP(hipMalloc((void **) p, n * sizeof(double)))) return false;
	*cap = n;
	return true;
}

extern "C" int mdns_joint_init_muse3(mdns_joint *j, const double *params, const double *jitter)
{
	Context *c = ctx();
	if (!c || !j || !params) return 1;
	if (j->kind != 1) { set_error("mdns_joint_init_muse3: the spectra carry no variances"); return 1; }
	if (j->nlive > MDNS_JOINT_MAX_BATCH) { set_error("mdns_joint_init_muse3: nlive=%d > %d", j->nlive, MDNS_JOINT_MAX_BATCH); return 1; }
	const size_t pbytes = (size_t) j->nlive * 5 * sizeof(double), n = (size_t) j->nlive * j->ndata;
	char *pin = joint_pin(j, pbytes);
	if (!pin) return 1;
	memcpy(pin, params, pbytes);
	if (!MDNS_HIP(hipMemcpyAsync(j->d_params, pin, pbytes, hipMemcpyHostToDevice, c->stream))) return 1;
	if (mdns_muse3_loglike_batch_dev(j->s, j->d_params, j->nlive, nullptr, j->ndata, j->st.live) != 0) return 1;
	if (jitter) {
		// (musefuse.py:535 adds its noise to the initial points' likelihoods too)
		if (!joint_grow(&j->d_jitter, &j->jitter_cap, n)) return 1;
		if (!MDNS_HIP(hipMemcpyAsync(j->d_jitter, jitter, n * sizeof(double), hipMemcpyHostToDevice, c->stream))) return 1;
		hipLaunchKernelGGL(k_joint_add, dim3(1024), dim3(kBlock), 0, c->stream, j->st.live, (const double *) j->d_jitter, n);
		if (!MDNS_HIP(hipGetLastError())) return 1;
	}
	if (joint_reset(j) != 0) return 1;
	return joint_sync(c) ? 0 : 1;
}

extern "C" int mdns_joint_restore_live_dev(mdns_joint *j, const double *d_liveL)
{
	Context *c = ctx();
	if (!c || !j || !d_liveL) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(j->st.live, d_liveL, (size_t) j->nlive * j->ndata * sizeof(double), hipMemcpyDeviceToDevice, c->stream))) return 1;
	return joint_reset(j);
}

extern "C" int mdns_joint_undo_advance_dev(mdns_joint *j)
{
	Context *c = ctx();
	if (!c || !j) return 1;
	if (!j->prepared) { set_error("mdns_joint_undo_advance_dev: nothing to take back"); return 1; }
	if (j->nrun == 0) return 0;
	hipLaunchKernelGGL(k_joint_undo_advance, dim3((j->nrun + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
	                   j->st, j->d_running, j->nrun, j->d_argmin, j->d_Lmin);
	return MDNS_HIP(hipGetLastError()) ? 0 : 1;
}

extern "C" const double *mdns_joint_live_dev(mdns_joint *j) { return j ? j->st.live : nullptr; }

extern "C" int mdns_joint_set_live(mdns_joint *j, const double *liveL)
{
	Context *c = ctx();
	if (!c || !j || !liveL) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(j->st.live, liveL, (size_t) j->nlive * j->ndata * sizeof(double), hipMemcpyHostToDevice, c->stream))) return 1;
	if (joint_reset(j) != 0) return 1;
	return joint_sync(c) ? 0 : 1;
}

extern "C" int mdns_joint_get_live(mdns_joint *j, double *liveL)
{
	Context *c = ctx();
	if (!c || !j || !liveL) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(liveL, j->st.live, (size_t) j->nlive * j->ndata * sizeof(double), hipMemcpyDeviceToHost, c->stream))) return 1;
	return joint_sync(c) ? 0 : 1;
}

extern "C" int mdns_joint_get_thresholds(mdns_joint *j, double *higher, int *shelf_n)
{
	Context *c = ctx();
	if (!c || !j) return 1;
	if (higher && !MDNS_HIP(hipMemcpyAsync(higher, j->st.higher, (size_t) j->ndata * sizeof(double), hipMemcpyDeviceToHost, c->stream))) return 1;
	if (shelf_n && !MDNS_HIP(hipMemcpyAsync(shelf_n, j->st.shelfn, (size_t) j->ndata * sizeof(int), hipMemcpyDeviceToHost, c->stream))) return 1;
	return joint_sync(c) ? 0 : 1;
}

extern "C" int mdns_joint_set_running(mdns_joint *j, const int *rows, int nrun)
{
	Context *c = ctx();
	if (!c || !j) return 1;
	if (nrun < 0 || nrun > j->ndata || (nrun > 0 && !rows)) { set_error("mdns_joint_set_running: nrun=%d", nrun); return 1; }
	for (int i = 0; i < nrun; i++)
		if (rows[i] < 0 || rows[i] >= j->ndata || (i > 0 && rows[i] <= rows[i - 1])) {
			set_error("mdns_joint_set_running: rows must be ascending indices below %d", j->ndata);
			return 1;
		}
	if (nrun > 0 && (!MDNS_HIP(hipMemcpyAsync(j->d_running, rows, (size_t) nrun * sizeof(int), hipMemcpyHostToDevice, c->stream)) ||
	                 !joint_sync(c))) return 1;
	j->nrun = nrun;
	return 0;
}

extern "C" int mdns_joint_prepare_dev(mdns_joint *j)
{
	Context *c = ctx();
	if (!c || !j) return 1;
	const int kw = keep_words_of(j->cap);
	const size_t need = (size_t) (j->nrun > 0 ? j->nrun : 1) * kw;
	if (need > j->keep_cap) {
		if (j->d_keep) { (void) hipStreamSynchronize(c->stream); (void) hipFree(j->d_keep); j->d_keep = nullptr; j->keep_cap = 0; }
		const size_t want = (size_t) j->ndata * kw;
		if (!MDNS_HIP(hipMalloc((void **) &j->d_keep, want * sizeof(unsigned long long)))) return 1;
		j->keep_cap = want;
	}
	j->prepared = true;
	if (j->nrun == 0) return 0;
	hipLaunchKernelGGL(k_joint_prepare, dim3((j->nrun + 15) / 16), dim3(kBlock), 0, c->stream,
	                   j->st, j->d_running, j->nrun, j->d_Lmin, j->d_argmin_run, j->d_argmin, j->d_keep, kw);
	return MDNS_HIP(hipGetLastError()) ? 0 : 1;
}

extern "C" int mdns_joint_prepare(mdns_joint *j, double *Lmin, int *argmin, unsigned long long *keep)
{
	Context *c = ctx();
	if (mdns_joint_prepare_dev(j) != 0) return 1;
	const int kw = keep_words_of(j->cap);
	const size_t n = (size_t) j->nrun;
	if (n == 0) return 0;
	// one pinned block, three copies, one wait
	const size_t o1 = n * 8, o2 = o1 + ((n * 4 + 7) & ~(size_t) 7);
	char *pin = joint_pin(j, o2 + n * kw * 8);
	if (!pin) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(pin, j->d_Lmin, n * 8, hipMemcpyDeviceToHost, c->stream)) ||
	    !MDNS_HIP(hipMemcpyAsync(pin + o1, j->d_argmin_run, n * 4, hipMemcpyDeviceToHost, c->stream)) ||
	    !MDNS_HIP(hipMemcpyAsync(pin + o2, j->d_keep, n * kw * 8, hipMemcpyDeviceToHost, c->stream)) ||
	    !joint_sync(c)) return 1;
	if (Lmin) memcpy(Lmin, pin, n * 8);
	if (argmin) memcpy(argmin, pin + o1, n * 4);
	if (keep) memcpy(keep, pin + o2, n * kw * 8);
	// what stays on the longest shelf (entries kept = set bits)
	int longest = 0;
	const unsigned long long *kwords = (const unsigned long long *) (pin + o2);
	for (size_t r = 0; r < n; r++) {
		int kept = 0;
		for (int w = 0; w < kw; w++) kept += __builtin_popcountll(kwords[r * kw + w]);
		if (kept > longest) longest = kept;
	}
	j->shelf_bound = longest;
	return 0;
}

extern "C" int mdns_joint_advance_dev(mdns_joint *j)
{
	Context *c = ctx();
	if (!c || !j) return 1;
	if (!j->prepared) { set_error("mdns_joint_advance: no prepare since the state was set"); return 1; }
	if (j->nrun == 0) return 0;
	hipLaunchKernelGGL(k_joint_advance, dim3((j->nrun + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
	                   j->st, j->d_running, j->nrun, j->d_argmin, j->d_status);
	return MDNS_HIP(hipGetLastError()) ? 0 : 1;
}

extern "C" int mdns_joint_advance(mdns_joint *j)
{
	Context *c = ctx();
	if (mdns_joint_advance_dev(j) != 0) return 1;
	int status = 0;
	if (!MDNS_HIP(hipMemcpyAsync(&status, j->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream)) || !joint_sync(c)) return 1;
	if (status) { set_error("mdns_joint_advance: a running data set had an empty shelf"); return 1; }
	if (j->shelf_bound > 0) j->shelf_bound--;
	return 0;
}

// ---------------------------------------------------------------------------------------
// the draw
// ---------------------------------------------------------------------------------------
// room for the trail of a chunk of B candidates over M selected spectra; a fresh stamp
static bool joint_trail(mdns_joint *j, int B, int M, JointTrail *out)
{
	Context *c = ctx();
	const size_t need = (size_t) B * ((M + 63) / 64);
	if (need > j->trail_cap) {
		(void) hipStreamSynchronize(c->stream);
		void *old[] = {j->d_trail_stamp, j->d_trail_word, j->d_trail_L};
		for (void *b : old) if (b) (void) hipFree(b);
		j->d_trail_stamp = nullptr; j->d_trail_word = nullptr; j->d_trail_L = nullptr; j->trail_cap = 0;
		const size_t cap = need + need / 2 + 1024;
		if (!MDNS_HIP(hipMalloc((void **) &j->d_trail_stamp, cap * sizeof(int))) ||
		    !MDNS_HIP(hipMalloc((void **) &j->d_trail_word, cap * sizeof(unsigned long long))) ||
		    !MDNS_HIP(hipMalloc((void **) &j->d_trail_L, cap * 64 * sizeof(double))) ||
		    !MDNS_HIP(hipMemsetAsync(j->d_trail_stamp, 0, cap * sizeof(int), c->stream))) return false;
		j->trail_cap = cap;
		j->trail_stamp = 0;
	}
	if (j->trail_stamp == 0x7fffffff) {                                 // stamps start over
		if (!MDNS_HIP(hipMemsetAsync(j->d_trail_stamp, 0, j->trail_cap * sizeof(int), c->stream))) return false;
		j->trail_stamp = 0;
	}
	out->stamp_of = j->d_trail_stamp; out->word = j->d_trail_word; out->L = j->d_trail_L;
	out->stamp = ++j->trail_stamp;
	return true;
}

// what a chunk's status word says (bit 0: commit, bit 1: the matrix-core filter's pick pass)
static void status_error(const char *who, int status, int cap)
{
	if (status & 2) set_error("%s: more ambiguous candidates than the exact pass lists (guarded filter)", who);
	else set_error("%s: a shelf overflowed its capacity %d (mdns_joint_reserve)", who, cap);
}

static bool check_draw(const mdns_joint *j, int B, int M, const char *who)
{
	if (!j) { set_error("%s: null handle", who); return false; }
	if (B < 0 || B > MDNS_JOINT_MAX_BATCH || M < 0 || M > j->ndata) {
		set_error("%s: bad sizes B=%d (<= %d) M=%d (ndata=%d)", who, B, MDNS_JOINT_MAX_BATCH, M, j->ndata);
		return false;
	}
	return true;
}

extern "C" int mdns_joint_score_dev(mdns_joint *j, const double *d_params, int B, double noise_level,
                                    const int *d_row_ids, int M)
{
	Context *c = ctx();
	if (!c || !check_draw(j, B, M, "mdns_joint_score_dev")) return 1;
	if (j->kind != 0) { set_error("mdns_joint_score_dev: Gaussian-line states only (use the mdns_backend_* entry points)"); return 1; }
	mdns_spectra *s = j->s;
	j->last_B = 0;
	j->trail_valid = false;
	if (B == 0 || M == 0) {
		// nothing to score: no flag can be set; still hand commit a clean header
		j->scored_M = M;
		return MDNS_HIP(hipMemsetAsync(j->d_flags, 0, (size_t) kZeroInts * sizeof(int), c->stream)) ? 0 : 1;
	}
	const double scale = -0.5 / (noise_level * noise_level);
	const int filter = gauss_filter_pays(s, M, B);
	const int bt = filter == 2 ? 16 : filter ? gauss_filter_tile(M, B) : gauss_cols_tile(M, B);
	bool gemm_form = false;
	if (!ensure_model(s, (size_t) cols_nx(s->nx) * (B + bt))) return 1;
	if (filter) {
		if (!j->d_msq && !MDNS_HIP(hipMalloc((void **) &j->d_msq, (size_t) (MDNS_JOINT_MAX_BATCH + 16) * sizeof(double)))) return 1;
		if (filter == 2 && !j->d_filter_scratch) {
			const size_t bytes = (size_t) (MDNS_JOINT_MAX_BATCH + 16) * sizeof(int);
			if (!MDNS_HIP(hipMalloc((void **) &j->d_filter_scratch, bytes)) || !MDNS_HIP(hipMemsetAsync(j->d_filter_scratch, 0, bytes, c->stream))) return 1;
		}
		// (the matrix-core filter with tiled operands wants the templates in its tiling too)
		double *model_g = nullptr;
		if (filter == 2 && gauss_mfma_form() == 2 && !d_row_ids && !s->d_yG) {
			// (the whole set in tiles of 16 rows: made on first use)
			if (!MDNS_HIP(hipMalloc((void **) &s->d_yG, (size_t) ((s->ndata + 15) / 16) * 16 * tiled16_nx(s->nx) * sizeof(double))) ||
			    !launch_tile_rows16(s->d_y, s->ld, s->ndata, s->nx, nullptr, s->d_yG)) return 1;
		}
		if (filter == 2 && gauss_mfma_form() == 2 && (d_row_ids || s->d_yG)) {
			const size_t need = (size_t) ((B + 15) / 16) * 16 * tiled16_nx(s->nx);
			if (need > s->model_g_cap) {
				if (s->d_model_g) { if (!joint_sync(c)) return 1; (void) hipFree(s->d_model_g); s->d_model_g = nullptr; s->model_g_cap = 0; }
				if (!MDNS_HIP(hipMalloc((void **) &s->d_model_g, (need + need / 2) * sizeof(double)))) return 1;
				s->model_g_cap = need + need / 2;
			}
			model_g = s->d_model_g;
		}
		if (!launch_gauss_model_tsq(s->d_x, s->nx, d_params, B, bt, s->d_model, j->d_msq, j->d_flags, kZeroInts, model_g)) return 1;
		gemm_form = model_g != nullptr;
	} else if (!launch_gauss_model_t(s->d_x, s->nx, d_params, B, bt, s->d_model, j->d_flags, kZeroInts)) return 1;
	const double *yT = s->d_yT;
	const int *gather = d_row_ids;
	// a sparse selection, or many candidate tiles over a selection: first a compact replica of
	// the selected spectra (one coalesced pass) instead of gathering columns in every tile
	const bool sparse = (size_t) M * 8 < (size_t) s->ndata;
	const double *yG = s->d_yG;
	if (d_row_ids && gemm_form) {
		// (its own compact replica of the selection, in its tiling)
		const size_t need = (size_t) ((M + 15) / 16) * 16 * tiled16_nx(s->nx);
		if (need > s->selG_cap) {
			if (s->d_selG) { if (!joint_sync(c)) return 1; (void) hipFree(s->d_selG); s->d_selG = nullptr; s->selG_cap = 0; }
			if (!MDNS_HIP(hipMalloc((void **) &s->d_selG, (need + need / 2) * sizeof(double)))) return 1;
			s->selG_cap = need + need / 2;
		}
		if (!launch_tile_rows16(s->d_y, s->ld, M, s->nx, d_row_ids, s->d_selG)) return 1;
		yG = s->d_selG;
	} else if (d_row_ids && (B >= 128 || sparse || filter == 2)) {
		if (!ensure_selection(s, (size_t) ((M + 63) / 64) * 64 * cols_nx(s->nx))) return 1;
		if (!launch_tile_columns(s->d_y, s->ld, M, s->nx, d_row_ids, s->d_sel)) return 1;
		yT = s->d_sel;
		gather = nullptr;
	}
	JointTrail trail;
	if (!joint_trail(j, B, M, &trail)) return 1;
	if (filter == 2) {
		int *lowest = (int *) &((JointHeader *) j->d_result)->pad;
		if (!launch_gauss_mfma_filter(s, yT, s->d_model, B, scale, d_row_ids, M, j->st.higher, j->d_flags, j->d_msq, trail, lowest,
		                              j->d_filter_scratch, j->d_result, gemm_form ? yG : nullptr, gemm_form ? s->d_model_g : nullptr)) return 1;
	} else if (filter) {
		// issue-bound launch: the guarded filter decides -- same flags and trail, bit for bit.  Its
		// "lowest flagged candidate so far" lives in the header's spare word (cleared with the flags).
		int *lowest = (int *) &((JointHeader *) j->d_result)->pad;
		if (!launch_gauss_cols_filter(s, yT, s->d_model, bt, B, scale, gather, d_row_ids, M, j->st.higher, j->d_flags, j->d_msq, trail, lowest)) return 1;
	} else if (!launch_gauss_cols_accept(s, yT, s->d_model, bt, B, scale, gather, d_row_ids, M, j->st.higher, j->d_flags, trail)) return 1;
	j->trail_valid = true;
	j->last_yT = yT; j->last_gather = gather; j->last_bt = bt; j->last_B = B; j->last_scale = scale;
	j->scored_M = M;
	return 0;
}

extern "C" int *mdns_joint_flags_dev(mdns_joint *j) { return j ? j->d_flags : nullptr; }
extern "C" const void *mdns_joint_result_dev(mdns_joint *j) { return j ? j->d_result : nullptr; }

static bool commit_ticket(mdns_joint *j)
{
	if (j->d_commit_ticket) return true;
	Context *c = ctx();
	return MDNS_HIP(hipMalloc((void **) &j->d_commit_ticket, sizeof(int))) && MDNS_HIP(hipMemsetAsync(j->d_commit_ticket, 0, sizeof(int), c->stream));
}

static int joint_commit_dev(mdns_joint *j, const int *d_row_ids, int M, bool want_row, const char *who)
{
	Context *c = ctx();
	if (!c || !check_draw(j, 0, M, who)) return 1;
	// flags and trail belong to ONE score: the commit must name the same selection size, once
	if (j->scored_M != M) {
		set_error("%s: M=%d, but the score that precedes had M=%d (-1: none, or already committed)", who, M, j->scored_M);
		return 1;
	}
	j->scored_M = -1;
	if (j->last_B == 0 || M == 0) {
		// an empty chunk accepts nothing (no kernel runs: the mailbox is filled from here, once
		// whatever may still be writing to it has finished)
		static const JointHeader none = {-1, 0, 0};
		if (!MDNS_HIP(hipMemcpyAsync(j->d_result, &none, sizeof none, hipMemcpyHostToDevice, c->stream)) || !joint_sync(c)) return 1;
		j->h_box->accepted = -1; j->h_box->status = 0;
		j->h_box->seq = ++j->box_seq;
		j->box_pending = true;
		return 0;
	}
	char *base = j->d_result;
	unsigned long long *bits = (unsigned long long *) (base + sizeof(JointHeader));
	double *Lrow = (double *) (base + sizeof(JointHeader) + (size_t) ((M + 63) / 64) * 8);
	if (!want_row && j->trail_valid) {
		// who beats its threshold, and with which likelihood, is in the trail of the accept pass:
		// nothing is computed again
		const JointTrail trail = {j->d_trail_stamp, j->d_trail_word, j->d_trail_L, j->trail_stamp};
		// (its last workgroup fills the mailbox)
		if (!commit_ticket(j) || !launch_joint_commit_trail(d_row_ids, M, j->last_B, j->d_flags, trail, j->st, base, bits, 1, j->h_box_dev,
		                                                    ++j->box_seq, j->d_commit_ticket)) return 1;
	} else {
		if (!launch_gauss_cols_commit(j->s, j->last_yT, j->s->d_model, j->last_bt, j->last_B, j->last_scale, j->last_gather,
		                              d_row_ids, M, j->d_flags, j->st, base, bits, Lrow)) return 1;
		hipLaunchKernelGGL(k_joint_publish, dim3(1), dim3(kBlock), 0, c->stream, (const JointHeader *) base, bits, (M + 63) / 64,
		                   j->h_box_dev, ++j->box_seq);
	}
	j->trail_valid = false;                                             // a chunk is committed once
	j->box_pending = true;
	return MDNS_HIP(hipGetLastError()) ? 0 : 1;
}

extern "C" int mdns_joint_commit_dev(mdns_joint *j, const int *d_row_ids, int M)
{
	return joint_commit_dev(j, d_row_ids, M, true, "mdns_joint_commit_dev");
}

extern "C" int mdns_joint_commit_bits_dev(mdns_joint *j, const int *d_row_ids, int M)
{
	return joint_commit_dev(j, d_row_ids, M, false, "mdns_joint_commit_bits_dev");
}

// waits for the mailbox of the last commit (see JointMailbox)
static bool joint_wait_box(mdns_joint *j, const char *who)
{
	Context *c = ctx();
	if (!j->box_pending) { set_error("%s: no commit to wait for", who); return false; }
	volatile unsigned long long *seq = &j->h_box->seq;
	// While polling, look at the stream now and then: a failed launch shows up as an error
	// instead of a hang.
	long long started = 0;
	for (unsigned spin = 0; *seq != j->box_seq; spin++) {
		if ((spin & 1023) != 1023) continue;
		const hipError_t e = hipStreamQuery(c->stream);
		if (e == hipErrorNotReady) {
			if (poll_expired(&started)) { set_error("%s: no outcome within MDNS_POLL_TIMEOUT_S", who); return false; }
			continue;
		}
		if (e != hipSuccess) { set_error("%s: the commit failed: %s", who, hipGetErrorString(e)); return false; }
		if (*seq != j->box_seq) { set_error("%s: the commit finished without an outcome", who); return false; }
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	j->box_pending = false;
	return true;
}

extern "C" int mdns_joint_fetch(mdns_joint *j, int M, int *accepted, unsigned long long *fillbits)
{
	if (!ctx() || !j || !accepted) return 1;
	if (M < 0 || M > j->ndata) { set_error("mdns_joint_fetch: M=%d", M); return 1; }
	if (!joint_wait_box(j, "mdns_joint_fetch")) return 1;
	if (j->h_box->status) { status_error("mdns_joint_fetch", j->h_box->status, j->cap); return 1; }
	*accepted = j->h_box->accepted;
	if (fillbits && j->h_box->accepted >= 0) memcpy(fillbits, (const void *) j->h_box->bits, (size_t) ((M + 63) / 64) * 8);
	return 0;
}

// host-pointer halves of the draw: `score` stages candidates and selection and leaves the accept
// flags on the device; `commit` finishes with whatever the flags say by then (a multi-GPU host
// reduces them over the ranks in between) and copies the outcome back
static int joint_stage_and_score(mdns_joint *j, const double *params, int B, double noise_level,
                                 const int *row_ids, int M, const char *who)
{
	Context *c = ctx();
	if (!c || !check_draw(j, B, M, who)) return 1;
	if (!j->prepared) { set_error("%s: thresholds are not set (call mdns_joint_prepare first)", who); return 1; }
	if (row_ids) {
		for (int k = 0; k < M; k++)
			if (row_ids[k] < 0 || row_ids[k] >= j->ndata || (k > 0 && row_ids[k] <= row_ids[k - 1])) {
				set_error("%s: row_ids must be ascending indices below %d (row_ids[%d]=%d)", who, j->ndata, k, row_ids[k]);
				return 1;
			}
	} else if (M != j->ndata && M != 0) {
		set_error("%s: M=%d without row_ids (ndata=%d)", who, M, j->ndata);
		return 1;
	}
	// candidates and selection travel together: one pinned block, one copy
	const size_t pbytes = (size_t) B * 24, rbytes = row_ids ? (size_t) M * 4 : 0;
	const size_t in_bytes = (pbytes + rbytes + 15) & ~(size_t) 15;
	char *pin = joint_pin(j, in_bytes + result_bytes(j->ndata));
	if (!pin) return 1;
	if (pbytes) memcpy(pin, params, pbytes);
	if (rbytes) memcpy(pin + pbytes, row_ids, rbytes);
	if (pbytes + rbytes &&
	    !MDNS_HIP(hipMemcpyAsync(j->d_params, pin, pbytes + rbytes, hipMemcpyHostToDevice, c->stream))) return 1;
	j->d_rows = (int *) ((char *) j->d_params + pbytes);
	j->staged_rows = row_ids != nullptr;
	j->staged_M = M;
	j->staged_in_bytes = in_bytes;
	const int rc = mdns_joint_score_dev(j, j->d_params, B, noise_level, row_ids ? j->d_rows : nullptr, M);
	j->staged_valid = rc == 0;
	return rc;
}

static int joint_commit_and_fetch(mdns_joint *j, int *accepted, double *Lrow, unsigned long long *fillbits, const char *who)
{
	Context *c = ctx();
	if (!c || !j) return 1;
	// (the selection, its size and the pinned block are those of the mdns_joint_score that precedes)
	if (!j->staged_valid) { set_error("%s: no mdns_joint_score with host pointers precedes this commit", who); return 1; }
	j->staged_valid = false;
	const int M = j->staged_M;
	if (joint_commit_dev(j, j->staged_rows ? j->d_rows : nullptr, M, Lrow != nullptr, who) != 0) return 1;
	if (!Lrow) {
		// no likelihood row wanted: the kernel leaves {accepted, status, fill words} in mapped host
		// memory -- no copy, no stream synchronisation
		if (!joint_wait_box(j, who)) return 1;
		if (j->h_box->status) { status_error(who, j->h_box->status, j->cap); return 1; }
		*accepted = j->h_box->accepted;
		if (fillbits && j->h_box->accepted >= 0) memcpy(fillbits, (const void *) j->h_box->bits, (size_t) ((M + 63) / 64) * 8);
		return 0;
	}
	// the header says whether the rest matters, but one copy of at most 80 KB costs less than a
	// second round trip; a caller that does not ask for the likelihood row gets header + bits
	const size_t out_bytes = Lrow ? result_bytes(M) : sizeof(JointHeader) + (size_t) ((M + 63) / 64) * 8;
	char *out = j->h_pin + j->staged_in_bytes;
	if (!MDNS_HIP(hipMemcpyAsync(out, j->d_result, out_bytes, hipMemcpyDeviceToHost, c->stream)) || !joint_sync(c)) return 1;
	const JointHeader *h = (const JointHeader *) out;
	if (h->status) { status_error(who, h->status, j->cap); return 1; }
	*accepted = h->accepted;
	if (h->accepted >= 0) {
		const size_t nb = (size_t) ((M + 63) / 64) * 8;
		if (fillbits) memcpy(fillbits, out + sizeof(JointHeader), nb);
		if (Lrow) memcpy(Lrow, out + sizeof(JointHeader) + nb, (size_t) M * 8);
	}
	return 0;
}

extern "C" int mdns_joint_score(mdns_joint *j, const double *params, int B, double noise_level,
                                const int *row_ids, int M)
{
	return joint_stage_and_score(j, params, B, noise_level, row_ids, M, "mdns_joint_score");
}

extern "C" int mdns_joint_commit(mdns_joint *j, int *accepted, double *Lrow, unsigned long long *fillbits)
{
	if (!accepted) { set_error("mdns_joint_commit: null output"); return 1; }
	*accepted = -1;
	return joint_commit_and_fetch(j, accepted, Lrow, fillbits, "mdns_joint_commit");
}

extern "C" int mdns_joint_draw_gauss(mdns_joint *j, const double *params, int B, double noise_level,
                                     const int *row_ids, int M, int *accepted, double *Lrow,
                                     unsigned long long *fillbits)
{
	if (!accepted) { set_error("mdns_joint_draw_gauss: null output"); return 1; }
	*accepted = -1;
	if (!check_draw(j, B, M, "mdns_joint_draw_gauss")) return 1;
	if (B == 0 || M == 0) return 0;
	if (joint_stage_and_score(j, params, B, noise_level, row_ids, M, "mdns_joint_draw_gauss") != 0) return 1;
	return joint_commit_and_fetch(j, accepted, Lrow, fillbits, "mdns_joint_draw_gauss");
}

// ---------------------------------------------------------------------------------------
// the device work of a native constrainer (include/mdns.h Part 5: mdns_draw_backend), user = the
// joint handle
// ---------------------------------------------------------------------------------------
extern "C" void *mdns_backend_region_create(void *joint, const double *members, int K, int ndim,
                                            const unsigned *packed, int nbootstraps, double *radius)
{
	(void) joint;
	if (!radius) { set_error("mdns_backend_region_create: null radius"); return nullptr; }
	if (packed) return mdns_region_create_bootstrapped(members, K, ndim, packed, nbootstraps, radius);
	mdns_region *r = mdns_region_create(members, K, ndim);
	if (r && mdns_region_set_radius(r, *radius) != 0) { mdns_region_destroy(r); return nullptr; }
	return r;
}

extern "C" void *mdns_backend_region_begin(void *joint, const double *members, int K, int ndim, const unsigned *packed, int nbootstraps)
{
	(void) joint;
	return region_begin_bootstrapped(members, K, ndim, packed, nbootstraps);
}

extern "C" int mdns_backend_region_radius(void *joint, void *region, double *radius)
{
	(void) joint;
	if (!region || !radius) { set_error("mdns_backend_region_radius: null argument"); return 1; }
	*radius = mdns_region_radius((mdns_region *) region);
	return *radius != *radius;
}

extern "C" void mdns_backend_region_destroy(void *joint, void *region)
{
	(void) joint;
	mdns_region_destroy((mdns_region *) region);
}

extern "C" int mdns_backend_region_count(void *joint, void *region, const double *points, int n, int *counts)
{
	(void) joint;
	return mdns_region_count_polled((mdns_region *) region, points, n, counts);
}

static constexpr size_t kInParams = (size_t) MDNS_JOINT_MAX_BATCH * 24;      // bytes of the candidates' slot in h_in

extern "C" int mdns_backend_draw_begin(void *joint, const int *rows, int M)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !check_draw(j, 0, M, "mdns_backend_draw_begin")) return 1;
	if (!j->prepared) { set_error("mdns_backend_draw_begin: thresholds are not set (call mdns_joint_prepare first)"); return 1; }
	j->sel_open = false;
	if (!j->h_in) {
		if (!MDNS_HIP(hipHostMalloc((void **) &j->h_in, kInParams + (size_t) j->ndata * sizeof(int), hipHostMallocMapped)) ||
		    !MDNS_HIP(hipHostGetDevicePointer((void **) &j->h_in_dev, j->h_in, 0))) return 1;
	}
	if (!j->d_sel_rows) {
		if (!MDNS_HIP(hipMalloc((void **) &j->d_sel_rows, (size_t) j->ndata * sizeof(int)))) return 1;
		j->sel_rows_cap = (size_t) j->ndata;
	}
	if (rows) {
		int *dst = (int *) (j->h_in + kInParams);
		int prev = -1;
		for (int k = 0; k < M; k++) {
			const int r = rows[k];
			if (r <= prev || r >= j->ndata) {
				set_error("mdns_backend_draw_begin: rows must be ascending indices below %d (rows[%d]=%d)", j->ndata, k, r);
				return 1;
			}
			dst[k] = prev = r;
		}
	} else if (M != j->ndata && M != 0) {
		set_error("mdns_backend_draw_begin: M=%d without rows (ndata=%d)", M, j->ndata);
		return 1;
	}
	j->sel_rows = rows != nullptr;
	j->sel_on_device = false;
	j->sel_M = M;
	j->sel_open = true;
	return 0;
}

// the chunk for kind 1: templates + K2 into the dense block, jitter, accept flags, commit, mailbox
static int backend_chunk_muse(mdns_joint *j, const double *params, int B, const double *jitter, int M)
{
	Context *c = ctx();
	const size_t pbytes = (size_t) B * 5 * sizeof(double), n = (size_t) B * M;
	char *pin = joint_pin(j, pbytes);
	if (!pin) return 1;
	memcpy(pin, params, pbytes);
	if (!MDNS_HIP(hipMemcpyAsync(j->d_params, pin, pbytes, hipMemcpyHostToDevice, c->stream))) return 1;
	if (j->sel_rows && !j->sel_on_device) {
		if (!MDNS_HIP(hipMemcpyAsync(j->d_sel_rows, j->h_in + kInParams, (size_t) M * sizeof(int), hipMemcpyHostToDevice, c->stream))) return 1;
		j->sel_on_device = true;
	}
	const int *d_rows = j->sel_rows ? j->d_sel_rows : nullptr;
	if (!joint_grow(&j->d_dense, &j->dense_cap, n)) return 1;
	if (jitter) {
		if (!joint_grow(&j->d_jitter, &j->jitter_cap, n)) return 1;
		// (pageable source: the runtime stages it and returns when the caller's buffer is free)
		if (!MDNS_HIP(hipMemcpyAsync(j->d_jitter, jitter, n * sizeof(double), hipMemcpyHostToDevice, c->stream))) return 1;
	}
	if (mdns_muse3_loglike_batch_dev(j->s, j->d_params, B, d_rows, M, j->d_dense) != 0) return 1;
	if (j->chunk_seq == 0x7fffffff) {
		if (!MDNS_HIP(hipMemsetAsync(j->d_flags, 0, (size_t) kFlagInts * sizeof(int), c->stream))) return 1;
		j->chunk_seq = 1;
	}
	const int flag = ++j->chunk_seq;
	char *base = j->d_result;
	unsigned long long *bits = (unsigned long long *) (base + sizeof(JointHeader));
	hipLaunchKernelGGL(k_joint_accept_dense, dim3((M + kBlock - 1) / kBlock, B), dim3(kBlock), 0, c->stream,
	                   j->d_dense, jitter ? (const double *) j->d_jitter : nullptr, B, M, d_rows, (const double *) j->st.higher,
	                   j->d_flags, flag, (JointHeader *) base);
	const int ntiles = (M + 63) / 64;
	hipLaunchKernelGGL(k_joint_commit_dense, dim3((ntiles + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, c->stream,
	                   (const double *) j->d_dense, d_rows, M, B, ntiles, (const int *) j->d_flags, flag, j->st, (JointHeader *) base, bits);
	hipLaunchKernelGGL(k_joint_publish, dim3(1), dim3(kBlock), 0, c->stream, (const JointHeader *) base, bits, ntiles, j->h_box_dev, ++j->box_seq);
	if (!MDNS_HIP(hipGetLastError())) return 1;
	j->box_pending = true;
	j->trail_valid = false;
	j->last_B = 0;
	return 0;
}

extern "C" int mdns_backend_draw_chunk(void *joint, const double *params, int B, const double *jitter, int *accepted,
                                       unsigned long long *fillbits, int *nscored)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !j || !accepted) return 1;
	*accepted = -1;
	if (!j->sel_open) { set_error("mdns_backend_draw_chunk: no draw begun"); return 1; }
	const int M = j->sel_M;
	if (!check_draw(j, B, M, "mdns_backend_draw_chunk")) return 1;
	if (nscored) *nscored = B;
	if (B == 0 || M == 0) return 0;
	if (j->shelf_bound + 1 > j->cap && mdns_joint_reserve(j, j->shelf_bound + 1) != 0) return 1;
	if (j->kind == 1) {
		if (backend_chunk_muse(j, params, B, jitter, M) != 0) return 1;
		if (mdns_joint_fetch(j, M, accepted, fillbits) != 0) return 1;
		if (*accepted >= 0) j->shelf_bound++;
		return 0;
	}
	if (jitter) { set_error("mdns_backend_draw_chunk: likelihood jitter is not part of the Gaussian-line problem"); return 1; }
	const size_t pbytes = (size_t) B * 24;
	static const char *chunk_path = getenv("MDNS_CHUNK_PATH");        // "classic": the five-command chunk (experiments)
	if (chunk_fits(j->s, M, B) && !(chunk_path && !strcmp(chunk_path, "classic"))) {
		// two launches: the kernels read candidates and (first chunk of the draw) the selection
		// from the mapped block; nothing is copied, nothing is cleared
		memcpy(j->h_in, params, pbytes);
		JointTrail trail;
		if (!joint_trail(j, B, M, &trail)) return 1;
		const int *rows_in = nullptr;
		int *rows_out = nullptr;
		if (j->sel_rows) {
			if (j->sel_on_device) rows_in = j->d_sel_rows;
			else { rows_in = (const int *) (j->h_in_dev + kInParams); rows_out = j->d_sel_rows; }
		}
		const double scale = -0.5 / (j->noise_level * j->noise_level);
		if (j->chunk_seq == 0x7fffffff) {
			if (!MDNS_HIP(hipMemsetAsync(j->d_flags, 0, (size_t) kFlagInts * sizeof(int), c->stream))) return 1;
			j->chunk_seq = 1;
		}
		const int flag = ++j->chunk_seq;
		char *base = j->d_result;
		unsigned long long *bits = (unsigned long long *) (base + sizeof(JointHeader));
		if (!launch_chunk_accept(j->s, (const double *) j->h_in_dev, B, scale, rows_in, rows_out, M, j->st.higher,
		                         j->d_flags, flag, trail, base)) return 1;
		if (j->sel_rows) j->sel_on_device = true;
		const int *thr_rows = j->sel_rows ? j->d_sel_rows : nullptr;
		if (M <= 128) {
			// a tile or two: shelf appends, thresholds and the mailbox in ONE workgroup
			if (!launch_chunk_commit(thr_rows, M, B, j->d_flags, flag, trail, j->st, base, bits, j->h_box_dev, ++j->box_seq)) return 1;
		} else {
			if (!commit_ticket(j) || !launch_joint_commit_trail(thr_rows, M, B, j->d_flags, trail, j->st, base, bits, flag, j->h_box_dev,
			                                                    ++j->box_seq, j->d_commit_ticket)) return 1;
		}
		j->box_pending = true;
		j->trail_valid = false;
		j->last_B = 0;
	} else {
		char *pin = joint_pin(j, pbytes);
		if (!pin) return 1;
		memcpy(pin, params, pbytes);
		if (!MDNS_HIP(hipMemcpyAsync(j->d_params, pin, pbytes, hipMemcpyHostToDevice, c->stream))) return 1;
		if (j->sel_rows && !j->sel_on_device) {
			// (the mapped block is pinned: a plain asynchronous copy)
			if (!MDNS_HIP(hipMemcpyAsync(j->d_sel_rows, j->h_in + kInParams, (size_t) M * sizeof(int), hipMemcpyHostToDevice, c->stream))) return 1;
			j->sel_on_device = true;
		}
		const int *d_rows = j->sel_rows ? j->d_sel_rows : nullptr;
		if (mdns_joint_score_dev(j, j->d_params, B, j->noise_level, d_rows, M) != 0) return 1;
		if (joint_commit_dev(j, d_rows, M, false, "mdns_backend_draw_chunk") != 0) return 1;
	}
	if (mdns_joint_fetch(j, M, accepted, fillbits) != 0) return 1;
	if (*accepted >= 0) j->shelf_bound++;
	return 0;
}

// candidates per chunk: four times the tries the last draw needed, within a budget of (candidate,
// spectrum) pairs (~50 us of GPU time) -- a speed choice only
extern "C" int mdns_backend_chunk_size(void *joint, int offered, int M, int hint)
{
	const mdns_joint *j = (const mdns_joint *) joint;
	// (a K2 pair reads 16 bytes x 4096 channels against K1's 8 x 200: a twentieth of the pairs,
	// and each speculative candidate also costs its jitter deviates on the host)
	const bool muse = j && j->kind == 1;
	const long long EVAL_BUDGET = muse ? 400000 : 2560000;
	// (over thousands of spectra at least 8: early in a run two candidates in five are accepted and a
	// chunk of 32 scored 30 of them for nothing -- with 8 a chunk over all 10 000 spectra is two
	// launches and 22 us instead of five commands and 44.  Over a few hundred spectra at least 32:
	// there a chunk costs its round trip whatever it holds, and 8 meant a fifth more chunks.)
	// (K2 with the noise in band form: a speculative candidate costs the host 2 ns per data set and the
	// device a share of a launch that is latency-bound below ~64 candidates -- round 3, with a Gaussian
	// deviate per pair on the host, kept these chunks at 4)
	const int MIN_CHUNK = muse ? 32 : (M > 4096 ? 8 : 32);
	long long budget = EVAL_BUDGET / (M > 0 ? M : 1);
	if (budget < MIN_CHUNK) budget = MIN_CHUNK;
	long long want = (muse ? 8LL : 4LL) * (hint > 0 ? hint : 1);
	if (want < MIN_CHUNK) want = MIN_CHUNK;
	if (budget > want) budget = want;
	if (budget > MDNS_JOINT_MAX_BATCH) budget = MDNS_JOINT_MAX_BATCH;
	return offered < budget ? offered : (int) budget;
}

// ---------------------------------------------------------------------------------------
// the first batch of a fresh region without a host look in between (mdns.h Part 5, chain_begin /
// chain_end; kernels in mdns_chain.hip)
// ---------------------------------------------------------------------------------------
extern "C" int mdns_backend_chain_begin(void *joint, void *region, const mdns_chain_request *rq)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !j || !region || !rq || !rq->u || !rq->mn || !rq->mx || !rq->prior) { set_error("mdns_backend_chain_begin: null argument"); return 1; }
	if (!j->sel_open) { set_error("mdns_backend_chain_begin: no draw begun"); return 1; }
	if (j->chain_state != 0) { set_error("mdns_backend_chain_begin: a chain is in flight"); return 1; }
	RegionView rv;
	if (!region_view((mdns_region *) region, &rv)) { set_error("mdns_backend_chain_begin: the region's radius is not on the device"); return 1; }
	const mdns_prior *prior = rq->prior;
	if (rq->n <= 0 || rq->n > kChainMost || rq->ndim != rv.ndim || rv.ndim > 5 || prior->ndim != rq->ndim) {
		set_error("mdns_backend_chain_begin: %d proposals, %d dimensions (region: %d)", rq->n, rq->ndim, rv.ndim);
		return 1;
	}
	if (!j->h_chain) {
		if (!MDNS_HIP(hipHostMalloc((void **) &j->h_chain, sizeof(ChainBox), hipHostMallocMapped)) ||
		    !MDNS_HIP(hipHostGetDevicePointer((void **) &j->h_chain_dev, j->h_chain, 0)) ||
		    !MDNS_HIP(hipMalloc((void **) &j->d_chain_props, (size_t) kChainMost * kChainDim * sizeof(double))) ||
		    !MDNS_HIP(hipMalloc((void **) &j->d_chain_counts, (size_t) kChainMost * sizeof(int))) ||
		    !MDNS_HIP(hipMalloc((void **) &j->d_chain_ticket, sizeof(int))) ||
		    !MDNS_HIP(hipMemsetAsync(j->d_chain_ticket, 0, sizeof(int), c->stream))) return 1;
		memset(j->h_chain, 0, sizeof(ChainBox));
	}
	const int M = j->sel_M;
	// the chunk rides along when the problem is the Gaussian line with the library's own prior
	// transform and the chunk would take the two-launch path anyway
	int limit = rq->limit;
	if (limit > MDNS_JOINT_MAX_BATCH) limit = MDNS_JOINT_MAX_BATCH;
	const bool full = limit > 0 && j->kind == 0 && !prior->custom && prior->nparams == 3 && prior->jitter_sigma == 0 && M > 0 &&
	                  chunk_fits(j->s, M, limit);
	ChainSpec spec;
	memset(&spec, 0, sizeof spec);
	spec.n = rq->n; spec.ndim = rq->ndim; spec.nparams = 3; spec.limit = full ? limit : 0;
	spec.identity = rq->identity || !rq->mean || !rq->scale;
	for (int k = 0; k < rq->ndim; k++) {
		spec.mn[k] = rq->mn[k]; spec.mx[k] = rq->mx[k];
		spec.mean[k] = spec.identity ? 0.0 : rq->mean[k];
		spec.scale[k] = spec.identity ? 1.0 : rq->scale[k];
		spec.a[k] = prior->a[k]; spec.b[k] = prior->b[k];
		spec.pow10[k] = prior->pow10[k]; spec.kernel_pow10[k] = prior->kernel_pow10[k];
	}
	memcpy(j->h_chain->u, rq->u, (size_t) rq->n * rq->ndim * sizeof(double));
	j->h_chain->nkept = -1; j->h_chain->B = -1;
	j->chain_n = rq->n;
	if (!full) {
		const CountMail mail = {j->d_chain_ticket, &j->h_chain_dev->seq, ++j->chain_seq};
		if (!launch_box_count(rv, spec, j->h_chain_dev, j->d_chain_props, j->d_chain_counts, &mail)) return 1;
		j->chain_state = 1;
		return 0;
	}
	if (j->shelf_bound + 1 > j->cap && mdns_joint_reserve(j, j->shelf_bound + 1) != 0) return 1;
	if (!launch_box_count(rv, spec, j->h_chain_dev, j->d_chain_props, j->d_chain_counts, nullptr)) return 1;
	JointTrail trail;
	if (!joint_trail(j, limit, M, &trail)) return 1;
	const int *rows_in = nullptr;
	int *rows_out = nullptr;
	if (j->sel_rows) {
		if (j->sel_on_device) rows_in = j->d_sel_rows;
		else { rows_in = (const int *) (j->h_in_dev + kInParams); rows_out = j->d_sel_rows; }
	}
	const double scale = -0.5 / (j->noise_level * j->noise_level);
	if (j->chunk_seq == 0x7fffffff) {
		if (!MDNS_HIP(hipMemsetAsync(j->d_flags, 0, (size_t) kFlagInts * sizeof(int), c->stream))) return 1;
		j->chunk_seq = 1;
	}
	const int flag = ++j->chunk_seq;
	char *base = j->d_result;
	unsigned long long *bits = (unsigned long long *) (base + sizeof(JointHeader));
	if (!launch_chain_accept(j->s, spec, j->d_chain_props, j->d_chain_counts, j->h_chain_dev, scale, rows_in, rows_out, M,
	                         j->st.higher, j->d_flags, flag, trail, base)) return 1;
	if (j->sel_rows) j->sel_on_device = true;
	const int *thr_rows = j->sel_rows ? j->d_sel_rows : nullptr;
	if (M <= 128) {
		if (!launch_chunk_commit(thr_rows, M, limit, j->d_flags, flag, trail, j->st, base, bits, j->h_box_dev, ++j->box_seq)) return 1;
	} else {
		if (!commit_ticket(j) || !launch_joint_commit_trail(thr_rows, M, limit, j->d_flags, trail, j->st, base, bits, flag, j->h_box_dev,
		                                                    ++j->box_seq, j->d_commit_ticket)) return 1;
	}
	j->box_pending = true;
	j->trail_valid = false;
	j->last_B = 0;
	j->chain_state = 2;
	return 0;
}

extern "C" int mdns_backend_chain_end(void *joint, void *region, int *counts, int *nkept, int *B, int *accepted,
                                      unsigned long long *fillbits, double *params)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	(void) region;
	if (!c || !j || !counts || !nkept || !B || !accepted) { set_error("mdns_backend_chain_end: null argument"); return 1; }
	const int state = j->chain_state;
	j->chain_state = 0;
	*accepted = -1; *nkept = -1; *B = 0;
	if (state == 1) {
		volatile unsigned long long *at = &j->h_chain->seq;
		long long started = 0;
		for (unsigned spin = 0; *at != j->chain_seq; spin++) {
			if ((spin & 1023) != 1023) continue;
			const hipError_t e = hipStreamQuery(c->stream);
			if (e == hipErrorNotReady) {
				if (poll_expired(&started)) { set_error("chain: no membership counts within MDNS_POLL_TIMEOUT_S"); return 1; }
				continue;
			}
			if (e != hipSuccess) { set_error("chain: the membership count failed: %s", hipGetErrorString(e)); return 1; }
			if (*at != j->chain_seq) {
				(void) hipMemsetAsync(j->d_chain_ticket, 0, sizeof(int), c->stream);
				set_error("chain: the membership count finished without a result");
				return 1;
			}
		}
		std::atomic_thread_fence(std::memory_order_acquire);
		memcpy(counts, (const void *) j->h_chain->counts, (size_t) j->chain_n * sizeof(int));
		return 0;
	}
	if (state != 2) { set_error("mdns_backend_chain_end: no chain in flight"); return 1; }
	if (mdns_joint_fetch(j, j->sel_M, accepted, fillbits) != 0) return 1;
	memcpy(counts, (const void *) j->h_chain->counts, (size_t) j->chain_n * sizeof(int));
	*nkept = j->h_chain->nkept;
	*B = j->h_chain->B;
	if (*B < 0 || *B > MDNS_JOINT_MAX_BATCH || *nkept < *B || *accepted >= *B) {
		set_error("chain: the device reports %d kept proposals, a chunk of %d, accepted %d", *nkept, *B, *accepted);
		return 1;
	}
	if (params && *B > 0) memcpy(params, (const void *) j->h_chain->params, (size_t) *B * 3 * sizeof(double));
	if (*accepted >= 0) j->shelf_bound++;
	return 0;
}

// ---------------------------------------------------------------------------------------
// mdns_backend_draw_chunk in two halves, for hosts that put something between them: with the data sets
// sharded over ranks, the MAX all-reduce of the candidates' votes (mdns_joint_votes_dev)
// ---------------------------------------------------------------------------------------
__global__ void k_flags_to_votes(const int *__restrict__ flags, int value, int *__restrict__ votes, int B)
{
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < B) votes[b] = flags[b] == value ? 1 : 0;
}

extern "C" int *mdns_joint_votes_dev(mdns_joint *j) { return j ? j->d_votes : nullptr; }

extern "C" int mdns_backend_draw_score(void *joint, const double *params, int B, const double *jitter)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !j || (!params && B > 0)) return 1;
	if (!j->sel_open) { set_error("mdns_backend_draw_score: no draw begun"); return 1; }
	const int M = j->sel_M;
	if (!check_draw(j, B, M, "mdns_backend_draw_score")) return 1;
	if (!j->d_votes && !MDNS_HIP(hipMalloc((void **) &j->d_votes, (size_t) MDNS_JOINT_MAX_BATCH * sizeof(int)))) return 1;
	j->half_path = 0; j->half_B = B;
	if (!MDNS_HIP(hipMemsetAsync(j->d_votes, 0, (size_t) MDNS_JOINT_MAX_BATCH * sizeof(int), c->stream))) return 1;
	if (B == 0 || M == 0) { j->half_path = -1; return 0; }          // (nothing of this rank's is selected: no vote)
	if (j->shelf_bound + 1 > j->cap && mdns_joint_reserve(j, j->shelf_bound + 1) != 0) return 1;
	if (j->chunk_seq == 0x7fffffff) {
		if (!MDNS_HIP(hipMemsetAsync(j->d_flags, 0, (size_t) kFlagInts * sizeof(int), c->stream))) return 1;
		j->chunk_seq = 1;
	}
	char *base = j->d_result;
	if (j->kind == 1) {
		const size_t pbytes = (size_t) B * 5 * sizeof(double), n = (size_t) B * M;
		char *pin = joint_pin(j, pbytes);
		if (!pin) return 1;
		memcpy(pin, params, pbytes);
		if (!MDNS_HIP(hipMemcpyAsync(j->d_params, pin, pbytes, hipMemcpyHostToDevice, c->stream))) return 1;
		if (j->sel_rows && !j->sel_on_device) {
			if (!MDNS_HIP(hipMemcpyAsync(j->d_sel_rows, j->h_in + kInParams, (size_t) M * sizeof(int), hipMemcpyHostToDevice, c->stream))) return 1;
			j->sel_on_device = true;
		}
		const int *d_rows = j->sel_rows ? j->d_sel_rows : nullptr;
		if (!joint_grow(&j->d_dense, &j->dense_cap, n)) return 1;
		if (jitter) {
			if (!joint_grow(&j->d_jitter, &j->jitter_cap, n)) return 1;
			if (!MDNS_HIP(hipMemcpyAsync(j->d_jitter, jitter, n * sizeof(double), hipMemcpyHostToDevice, c->stream))) return 1;
		}
		if (mdns_muse3_loglike_batch_dev(j->s, j->d_params, B, d_rows, M, j->d_dense) != 0) return 1;
		const int flag = ++j->chunk_seq;
		hipLaunchKernelGGL(k_joint_accept_dense, dim3((M + kBlock - 1) / kBlock, B), dim3(kBlock), 0, c->stream,
		                   j->d_dense, jitter ? (const double *) j->d_jitter : nullptr, B, M, d_rows, (const double *) j->st.higher,
		                   j->d_flags, flag, (JointHeader *) base);
		j->half_path = 1; j->half_flag = flag;
	} else {
		if (jitter) { set_error("mdns_backend_draw_score: likelihood jitter is not part of the Gaussian-line problem"); return 1; }
		const size_t pbytes = (size_t) B * 24;
		if (chunk_fits(j->s, M, B)) {
			memcpy(j->h_in, params, pbytes);
			JointTrail trail;
			if (!joint_trail(j, B, M, &trail)) return 1;
			const int *rows_in = nullptr;
			int *rows_out = nullptr;
			if (j->sel_rows) {
				if (j->sel_on_device) rows_in = j->d_sel_rows;
				else { rows_in = (const int *) (j->h_in_dev + kInParams); rows_out = j->d_sel_rows; }
			}
			const double scale = -0.5 / (j->noise_level * j->noise_level);
			const int flag = ++j->chunk_seq;
			if (!launch_chunk_accept(j->s, (const double *) j->h_in_dev, B, scale, rows_in, rows_out, M, j->st.higher,
			                         j->d_flags, flag, trail, base)) return 1;
			if (j->sel_rows) j->sel_on_device = true;
			j->half_path = 2; j->half_flag = flag;
		} else {
			char *pin = joint_pin(j, pbytes);
			if (!pin) return 1;
			memcpy(pin, params, pbytes);
			if (!MDNS_HIP(hipMemcpyAsync(j->d_params, pin, pbytes, hipMemcpyHostToDevice, c->stream))) return 1;
			if (j->sel_rows && !j->sel_on_device) {
				if (!MDNS_HIP(hipMemcpyAsync(j->d_sel_rows, j->h_in + kInParams, (size_t) M * sizeof(int), hipMemcpyHostToDevice, c->stream))) return 1;
				j->sel_on_device = true;
			}
			if (mdns_joint_score_dev(j, j->d_params, B, j->noise_level, j->sel_rows ? j->d_sel_rows : nullptr, M) != 0) return 1;
			j->half_path = 3; j->half_flag = 1;
		}
	}
	hipLaunchKernelGGL(k_flags_to_votes, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, (const int *) j->d_flags, j->half_flag, j->d_votes, B);
	return MDNS_HIP(hipGetLastError()) ? 0 : 1;
}

// second half: the first candidate with a vote (after whatever the caller did to the votes) is the
// accepted one; *accepted and the fill bits of THIS handle's selected data sets as mdns_backend_draw_chunk
extern "C" int mdns_backend_draw_commit(void *joint, int *accepted, unsigned long long *fillbits)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !j || !accepted) return 1;
	*accepted = -1;
	const int path = j->half_path, B = j->half_B, M = j->sel_M;
	j->half_path = 0;
	if (path == 0) { set_error("mdns_backend_draw_commit: no mdns_backend_draw_score precedes"); return 1; }
	if (path < 0) {
		// nothing of this handle's was scored: the accepted candidate is whoever the votes name
		std::vector<int> votes((size_t) (B > 0 ? B : 1), 0);
		if (B > 0 && (!MDNS_HIP(hipMemcpyAsync(votes.data(), j->d_votes, (size_t) B * sizeof(int), hipMemcpyDeviceToHost, c->stream)) || !joint_sync(c))) return 1;
		for (int b = 0; b < B; b++) if (votes[b]) { *accepted = b; break; }
		return 0;
	}
	char *base = j->d_result;
	unsigned long long *bits = (unsigned long long *) (base + sizeof(JointHeader));
	const int ntiles = (M + 63) / 64;
	const int *thr_rows = j->sel_rows ? j->d_sel_rows : nullptr;
	if (path == 1) {
		hipLaunchKernelGGL(k_joint_commit_dense, dim3((ntiles + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, c->stream,
		                   (const double *) j->d_dense, thr_rows, M, B, ntiles, (const int *) j->d_votes, 1, j->st, (JointHeader *) base, bits);
		hipLaunchKernelGGL(k_joint_publish, dim3(1), dim3(kBlock), 0, c->stream, (const JointHeader *) base, bits, ntiles, j->h_box_dev, ++j->box_seq);
		if (!MDNS_HIP(hipGetLastError())) return 1;
		j->box_pending = true;
	} else if (path == 2) {
		const JointTrail trail = {j->d_trail_stamp, j->d_trail_word, j->d_trail_L, j->trail_stamp};
		if (M <= 128) {
			if (!launch_chunk_commit(thr_rows, M, B, j->d_votes, 1, trail, j->st, base, bits, j->h_box_dev, ++j->box_seq)) return 1;
		} else {
			if (!commit_ticket(j) || !launch_joint_commit_trail(thr_rows, M, B, j->d_votes, trail, j->st, base, bits, 1, j->h_box_dev, ++j->box_seq,
			                                                    j->d_commit_ticket)) return 1;
		}
		j->box_pending = true;
	} else {
		// the lane kernels' flags are 0 / 1 themselves: the votes go back into them
		if (!MDNS_HIP(hipMemcpyAsync(j->d_flags, j->d_votes, (size_t) B * sizeof(int), hipMemcpyDeviceToDevice, c->stream))) return 1;
		if (joint_commit_dev(j, thr_rows, M, false, "mdns_backend_draw_commit") != 0) return 1;
	}
	j->trail_valid = false;
	j->last_B = 0;
	if (mdns_joint_fetch(j, M, accepted, fillbits) != 0) return 1;
	if (*accepted >= 0) j->shelf_bound++;
	return 0;
}

// ---------------------------------------------------------------------------------------
// the likelihood noise in band form (mdns.h Part 5: draw_band / draw_band_commit)
// ---------------------------------------------------------------------------------------
// candidates and their bounds travel as ONE block [B x 5 | B] into the head of d_bound; the noise row of a
// commit lives behind it
static constexpr size_t kBandRowAt = (size_t) MDNS_JOINT_MAX_BATCH * 6;

// launches of one scoring of the chunk in place (by the matrix-core filter or by the exact kernels)
static int band_launch(mdns_joint *j, bool filtered)
{
	Context *c = ctx();
	const int B = j->band_B, M = j->sel_M;
	const int *d_rows = j->sel_rows ? j->d_sel_rows : nullptr;
	const double *d_p = j->d_bound, *d_b = j->d_bound + (size_t) B * 5;
	if (filtered) {
		const int ldm = model_ld(j->s->nx) + 16;            // (not a power of two: mdns_k2gemm.hip, muse_filter_ld)
		if (!ensure_model(j->s, (size_t) B * ldm) || !launch_muse3_model(j->s->d_x, j->s->nx, d_p, B, j->s->d_model, ldm)) return 1;
		const MuseBandOut out = {&j->d_band->counter, j->d_band->clear, j->d_band->maybe, j->d_band->pair_b, j->d_band->pair_k,
		                         j->d_band->pair_L, j->d_band->pair_thr, kBandCap, &((JointHeader *) j->d_result)->status};
		if (!launch_muse_filter(j->s, j->s->d_model, ldm, B, d_rows, M, j->st.higher, d_b, out)) return 1;
	} else if (muse_rows_variant(B, M) == 1 && j->s->d_w && j->s->d_x) {
		// small chunks (pairs of candidates per workgroup): templates, then ONE kernel that scores, votes and publishes
		const int ldm = model_ld(j->s->nx);
		if (!ensure_model(j->s, (size_t) B * ldm) || !launch_muse3_model(j->s->d_x, j->s->nx, d_p, B, j->s->d_model, ldm)) return 1;
		const MuseBandFused fused = {j->d_band, j->h_band_dev, ++j->band_seq, j->st.higher, d_b, &((JointHeader *) j->d_result)->status};
		if (!launch_muse_rows(j->s, j->s->d_model, ldm, B, d_rows, M, j->d_dense, 0, &fused)) return 1;
	} else {
		if (mdns_muse3_loglike_batch_dev(j->s, d_p, B, d_rows, M, j->d_dense) != 0) return 1;
		hipLaunchKernelGGL(k_joint_band, dim3((M + kBlock - 1) / kBlock, B), dim3(kBlock), 0, c->stream, (const double *) j->d_dense,
		                   d_b, B, M, d_rows, (const double *) j->st.higher, j->d_band, (JointHeader *) j->d_result, j->h_band_dev, ++j->band_seq);
	}
	if (filtered) hipLaunchKernelGGL(k_joint_band_publish, dim3(1), dim3(kBlock), 0, c->stream, j->d_band, B, j->h_band_dev, ++j->band_seq);
	if (!MDNS_HIP(hipGetLastError())) return 1;
	j->band_exact = !filtered;
	return 0;
}

// the chunk is on its way when this returns; mdns_backend_draw_band_ready tells whether its outcome has
// arrived (a look at mapped memory), mdns_backend_draw_band_end waits for it and hands it over
extern "C" int mdns_backend_draw_band_begin(void *joint, const double *params, int B, const double *bound)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !j || !params || !bound) { set_error("mdns_backend_draw_band: null argument"); return 1; }
	if (j->kind != 1) { set_error("mdns_backend_draw_band: a state of the scale-marginalised likelihood is needed"); return 1; }
	if (!j->sel_open) { set_error("mdns_backend_draw_band: no draw begun"); return 1; }
	if (j->band_pending) { set_error("mdns_backend_draw_band_begin: the last chunk was not collected"); return 1; }
	const int M = j->sel_M;
	if (!check_draw(j, B, M, "mdns_backend_draw_band") || B == 0 || M == 0) return 1;
	if (!j->h_band) {
		if (!MDNS_HIP(hipHostMalloc((void **) &j->h_band, sizeof(BandBox), hipHostMallocMapped)) ||
		    !MDNS_HIP(hipHostGetDevicePointer((void **) &j->h_band_dev, j->h_band, 0)) ||
		    !MDNS_HIP(hipMalloc((void **) &j->d_band, sizeof(BandScratch))) ||
		    !MDNS_HIP(hipMemsetAsync(j->d_band, 0, sizeof(BandScratch), c->stream)) ||
		    !MDNS_HIP(hipMalloc((void **) &j->d_bound, (kBandRowAt + j->ndata) * sizeof(double)))) return 1;
		memset(j->h_band, 0, sizeof(BandBox));
	}
	if (j->shelf_bound + 1 > j->cap && mdns_joint_reserve(j, j->shelf_bound + 1) != 0) return 1;
	const size_t pbytes = (size_t) B * 5 * sizeof(double), bbytes = (size_t) B * sizeof(double), n = (size_t) B * M;
	char *pin = joint_pin(j, pbytes + bbytes);
	if (!pin) return 1;
	memcpy(pin, params, pbytes);
	memcpy(pin + pbytes, bound, bbytes);
	if (!MDNS_HIP(hipMemcpyAsync(j->d_bound, pin, pbytes + bbytes, hipMemcpyHostToDevice, c->stream))) return 1;
	if (j->sel_rows && !j->sel_on_device) {
		if (!MDNS_HIP(hipMemcpyAsync(j->d_sel_rows, j->h_in + kInParams, (size_t) M * sizeof(int), hipMemcpyHostToDevice, c->stream))) return 1;
		j->sel_on_device = true;
	}
	if (!joint_grow(&j->d_dense, &j->dense_cap, n)) return 1;
	j->band_B = B;
	j->trail_valid = false;
	j->last_B = 0;
	// Large chunks first as two matrix products with a widened band (mdns_k2gemm.hip): whatever that settles
	// is settled as the exact kernels would; a chunk it lists a pair of is scored again by those (_end).
	if (band_launch(j, muse_filter_applies(j->s, B, M)) != 0) { j->band_B = 0; return 1; }
	j->band_pending = true;
	return 0;
}

extern "C" int mdns_backend_draw_band_ready(void *joint)
{
	mdns_joint *j = (mdns_joint *) joint;
	if (!j || !j->band_pending || !j->h_band) return 1;
	return *(volatile unsigned long long *) &j->h_band->seq == j->band_seq ? 1 : 0;
}

extern "C" int mdns_backend_draw_band_end(void *joint, int *status, int *npairs, int *pair_b, int *pair_k, double *pair_L, double *pair_thr,
                                          int cap)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !j || !status || !npairs || !pair_b || !pair_k || !pair_L || !pair_thr) { set_error("mdns_backend_draw_band: null argument"); return 1; }
	if (!j->band_pending) { set_error("mdns_backend_draw_band_end: no chunk begun"); return 1; }
	j->band_pending = false;
	const int B = j->band_B;
	for (;;) {
		volatile unsigned long long *at = &j->h_band->seq;
		long long started = 0;
		for (unsigned spin = 0; *at != j->band_seq; spin++) {
			if ((spin & 1023) != 1023) continue;
			const hipError_t e = hipStreamQuery(c->stream);
			if (e == hipErrorNotReady) {
				if (poll_expired(&started)) { set_error("mdns_backend_draw_band: no outcome within MDNS_POLL_TIMEOUT_S"); return 1; }
				continue;
			}
			if (e != hipSuccess) { set_error("mdns_backend_draw_band: %s", hipGetErrorString(e)); return 1; }
			if (*at != j->band_seq) { set_error("mdns_backend_draw_band: finished without an outcome"); return 1; }
		}
		std::atomic_thread_fence(std::memory_order_acquire);
		if (j->band_exact || j->h_band->npairs == 0) break;
		muse_filter_note(1);
		if (band_launch(j, false) != 0) return 1;
	}
	memcpy(status, (const void *) j->h_band->status, (size_t) B * sizeof(int));
	*npairs = j->h_band->npairs;
	const int m = *npairs < cap ? (*npairs < kBandCap ? *npairs : kBandCap) : cap;
	if (*npairs > kBandCap && *npairs <= cap) *npairs = cap + 1;          // (more than the device lists: the caller falls back)
	memcpy(pair_b, (const void *) j->h_band->pair_b, (size_t) m * sizeof(int));
	memcpy(pair_k, (const void *) j->h_band->pair_k, (size_t) m * sizeof(int));
	memcpy(pair_L, (const void *) j->h_band->pair_L, (size_t) m * sizeof(double));
	memcpy(pair_thr, (const void *) j->h_band->pair_thr, (size_t) m * sizeof(double));
	return 0;
}

extern "C" int mdns_backend_draw_band(void *joint, const double *params, int B, const double *bound, int *status, int *npairs,
                                      int *pair_b, int *pair_k, double *pair_L, double *pair_thr, int cap)
{
	if (mdns_backend_draw_band_begin(joint, params, B, bound) != 0) return 1;
	return mdns_backend_draw_band_end(joint, status, npairs, pair_b, pair_k, pair_L, pair_thr, cap);
}

extern "C" int mdns_backend_draw_band_commit(void *joint, int b, const double *jitter_row, unsigned long long *fillbits)
{
	Context *c = ctx();
	mdns_joint *j = (mdns_joint *) joint;
	if (!c || !j || !jitter_row) { set_error("mdns_backend_draw_band_commit: null argument"); return 1; }
	if (j->kind != 1 || !j->sel_open || j->band_B <= 0 || b < 0 || b >= j->band_B) { set_error("mdns_backend_draw_band_commit: candidate %d of a chunk of %d", b, j->band_B); return 1; }
	const int M = j->sel_M, ntiles = (M + 63) / 64;
	if (!j->band_exact) {
		// the chunk went through the matrix-core filter: what the state keeps is the exact kernel's row --
		// from the instantiation that would have scored the whole block, bit for bit (the templates of
		// the chunk are still in place)
		const int ldm = model_ld(j->s->nx) + 16, Bc = j->band_B;
		const int lo = muse_rows_variant(Bc, M) == 1 ? (b & ~1) : b;
		const int nb = lo == b && muse_rows_variant(Bc, M) != 1 ? 1 : (Bc - lo < 2 ? Bc - lo : 2);
		if (!launch_muse_rows(j->s, j->s->d_model + (size_t) lo * ldm, ldm, nb, j->sel_rows ? j->d_sel_rows : nullptr, M,
		                      j->d_dense + (size_t) lo * M, Bc)) return 1;
		muse_filter_note(2);
	}
	j->band_B = 0;
	double *d_row = j->d_bound + kBandRowAt;
	char *pin = joint_pin(j, (size_t) M * sizeof(double));
	if (!pin) return 1;
	memcpy(pin, jitter_row, (size_t) M * sizeof(double));
	if (!MDNS_HIP(hipMemcpyAsync(d_row, pin, (size_t) M * sizeof(double), hipMemcpyHostToDevice, c->stream))) return 1;
	char *base = j->d_result;
	unsigned long long *bits = (unsigned long long *) (base + sizeof(JointHeader));
	const int *d_rows = j->sel_rows ? j->d_sel_rows : nullptr;
	hipLaunchKernelGGL(k_joint_commit_band, dim3((ntiles + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, c->stream,
	                   (const double *) j->d_dense, (const double *) d_row, d_rows, M, b, ntiles, j->st, (JointHeader *) base, bits);
	hipLaunchKernelGGL(k_joint_publish, dim3(1), dim3(kBlock), 0, c->stream, (const JointHeader *) base, bits, ntiles, j->h_box_dev, ++j->box_seq);
	if (!MDNS_HIP(hipGetLastError())) return 1;
	j->box_pending = true;
	int accepted = -1;
	if (mdns_joint_fetch(j, M, &accepted, fillbits) != 0) return 1;
	if (accepted != b) { set_error("mdns_backend_draw_band_commit: committed %d, asked for %d", accepted, b); return 1; }
	j->shelf_bound++;
	return 0;
}
