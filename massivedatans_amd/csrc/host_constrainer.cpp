// The MLFriends constrainer of the reference behind ONE native call per constrained draw
// (include/mdns.h, Part 5).  Plain host C++: no GPU code here -- the kernels are reached through
// the mdns_draw_backend table (libmdns_hip.so on a GPU box, the CPU oracle in tests).
//
// What is restated, and from where (paths under the reference checkout):
//   hiermetriclearn.py:27-211        MetricLearningFriendsConstrainer: rebuild policy, region
//                                    construction, candidate generator, accept loop
//   clustering/radfriendsregion.py:58-182   RadFriendsRegion: bootstrap radius, bounding box,
//                                    box / ball proposals
//   clustering/sdml.py:25-88         the axis-scaling metrics (their FIT is left to numpy through
//                                    mdns_numpy_ops: numpy's log2 is not the C library's)
//   clustering/neighbors.py:170-177  the bootstrap choice
// The sequence of random numbers is part of the results: every draw is taken
// from numpy's own mt19937 state with numpy's LEGACY algorithms (numpy/random/src/legacy/
// legacy-distributions.c, distributions.c) in the reference's call order, so the global stream is
// left exactly where the reference would leave it.  Floating-point operations are written one per
// statement in numpy's order (this file is compiled with -ffp-contract=off).
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <deque>
#include <memory>
#include <pthread.h>
#include <time.h>
#include <vector>

#include "mdns.h"

extern "C" int mdns_host_bootstrap_masks_mt(void *state, int64_t K, int rounds, uint32_t *masks);
extern "C" int mdns_host_bootstrap_skip_mt(void *state, int64_t K, int rounds);

namespace {

char g_error[512] = "";

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof g_error, fmt, ap);
	va_end(ap);
}

// ---------------------------------------------------------------------------------------
// numpy's legacy stream on its own Mersenne Twister state
// ---------------------------------------------------------------------------------------
struct MT { uint32_t key[624]; int pos; };        // numpy/random/src/mt19937/mt19937.h

// (inlined into its callers so that their vector clones get a vector twist as well: the loops have
// no dependency shorter than 227 elements)
static inline __attribute__((always_inline)) void mt_refill(MT *s)
{
	const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
	uint32_t y;
	int i;
	for (i = 0; i < 624 - 397; i++) {
		y = (s->key[i] & UP) | (s->key[i + 1] & LO);
		s->key[i] = s->key[i + 397] ^ (y >> 1) ^ (-(y & 1) & A);
	}
	for (; i < 623; i++) {
		y = (s->key[i] & UP) | (s->key[i + 1] & LO);
		s->key[i] = s->key[i + (397 - 624)] ^ (y >> 1) ^ (-(y & 1) & A);
	}
	y = (s->key[623] & UP) | (s->key[0] & LO);
	s->key[623] = s->key[396] ^ (y >> 1) ^ (-(y & 1) & A);
	s->pos = 0;
}

inline uint32_t mt_next(MT *s)
{
	if (s->pos == 624) mt_refill(s);
	uint32_t y = s->key[s->pos++];
	y ^= (y >> 11);
	y ^= (y << 7) & 0x9d2c5680u;
	y ^= (y << 15) & 0xefc60000u;
	y ^= (y >> 18);
	return y;
}

// mt19937_next_double: 53 bits from two draws
inline double mt_double(MT *s)
{
	const int32_t a = (int32_t) (mt_next(s) >> 5), b = (int32_t) (mt_next(s) >> 6);
	return (a * 67108864.0 + b) / 9007199254740992.0;
}

// n doubles at once: the same numbers as n calls of mt_double, in three flat loops the compiler
// vectorises (tempering of the words of a refill, then pairs -> doubles); the proposals of a
// batch are 3000 of them
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
void mt_fill_doubles(MT *s, double *out, size_t n)
{
	enum { BLOCK = 1024 };                                   // doubles per pass: 8 KB of words on the stack
	uint32_t w[2 * BLOCK];
	while (n > 0) {
		const size_t m = n < (size_t) BLOCK ? n : (size_t) BLOCK;
		size_t have = 0;
		while (have < 2 * m) {
			if (s->pos == 624) mt_refill(s);
			size_t take = 624 - (size_t) s->pos;
			if (take > 2 * m - have) take = 2 * m - have;
			const uint32_t *key = s->key + s->pos;
			uint32_t *dst = w + have;
			for (size_t i = 0; i < take; i++) {
				uint32_t y = key[i];
				y ^= (y >> 11);
				y ^= (y << 7) & 0x9d2c5680u;
				y ^= (y << 15) & 0xefc60000u;
				y ^= (y >> 18);
				dst[i] = y;
			}
			s->pos += (int) take;
			have += take;
		}
		for (size_t i = 0; i < m; i++) {
			const int32_t a = (int32_t) (w[2 * i] >> 5), b = (int32_t) (w[2 * i + 1] >> 6);
			out[i] = (a * 67108864.0 + b) / 9007199254740992.0;
		}
		out += m;
		n -= m;
	}
}

// legacy_gauss keeps the second deviate of a pair (aug_bitgen_t.has_gauss / .gauss): one cache per
// process, like the global RandomState's
int g_has_gauss = 0;
double g_gauss = 0.0;

inline double legacy_gauss(MT *s)
{
	if (g_has_gauss) {
		const double t = g_gauss;
		g_has_gauss = 0;
		g_gauss = 0.0;
		return t;
	}
	double f, x1, x2, r2;
	do {
		x1 = 2.0 * mt_double(s) - 1.0;
		x2 = 2.0 * mt_double(s) - 1.0;
		r2 = x1 * x1 + x2 * x2;
	} while (r2 >= 1.0 || r2 == 0.0);
	f = std::sqrt(-2.0 * std::log(r2) / r2);
	g_gauss = f * x1;
	g_has_gauss = 1;
	return f * x2;
}

// ---- the double stream in blocks, with what the polar method makes of their pairs -------------
// numpy's legacy Gaussian takes the doubles two at a time, (x1, x2) = 2 u - 1, and rejects the pair unless
// 0 < r2 = x1^2 + x2^2 < 1: which pairs of a block survive, and their r2, are flat loops over the block.
enum { kBlockDoubles = 1024, kBlockPairs = kBlockDoubles / 2 };

#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
void pairs_r2(const double *buf, double *r2, int npairs)
{
	for (int p = 0; p < npairs; p++) {
		const double x1 = 2.0 * buf[2 * p] - 1.0, x2 = 2.0 * buf[2 * p + 1] - 1.0;
		const double v = x1 * x1 + x2 * x2;
		r2[p] = (v >= 1.0 || v == 0.0) ? 3.0 : v;           // 3: rejected (no minimum takes it)
	}
}

// (a loop of its own: with the byte store in the loop above the compiler gives up its wide form)
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
void pairs_ok(const double *r2, uint8_t *ok, int npairs)
{
	for (int p = 0; p < npairs; p++) ok[p] = r2[p] < 1.5;
}

// one block of the stream and what the polar method makes of its pairs
struct BlockData {
	MT start;                    // the state the block was made from
	double buf[kBlockDoubles];
	double r2[kBlockPairs];      // of the block's pairs: x1^2 + x2^2, or 3 where the pair is rejected
	alignas(8) uint8_t ok[kBlockPairs];          // 1: accepted
	uint16_t cum8[kBlockPairs / 8 + 1];          // pairs accepted before group g of eight
	uint64_t group(int g) const { uint64_t w; memcpy(&w, ok + 8 * g, 8); return w; }
	void analyse()
	{
		pairs_r2(buf, r2, kBlockPairs);
		pairs_ok(r2, ok, kBlockPairs);
		unsigned n = 0;
		for (int g = 0; g < kBlockPairs / 8; g++) { cum8[g] = (uint16_t) n; n += (unsigned) __builtin_popcountll(group(g)); }
		cum8[kBlockPairs / 8] = (uint16_t) n;
	}
	int total() const { return cum8[kBlockPairs / 8]; }
	// pairs accepted before pair p
	int before(int p) const
	{
		if (p >= kBlockPairs) return total();
		const int k = p & 7;
		return cum8[p >> 3] + (k ? __builtin_popcountll(group(p >> 3) & ((1ull << (8 * k)) - 1)) : 0);
	}
	// smallest pe with `pairs` (>= 1) accepted pairs in [p, pe)  (there are that many)
	int end_of(int p, int pairs) const
	{
		const unsigned target = (unsigned) before(p) + (unsigned) pairs;
		int lo = (p >> 3) + 1, hi = kBlockPairs / 8;             // first group boundary g with cum8[g] >= target
		while (lo < hi) {
			const int mid = (lo + hi) >> 1;
			if (cum8[mid] >= target) hi = mid; else lo = mid + 1;
		}
		const int g = lo - 1;
		unsigned n = cum8[g];
		for (int j = 0; j < 8; j++) { n += ok[8 * g + j]; if (n >= target) return 8 * g + j + 1; }
		return 8 * g + 8;
	}
	double min_r2(int p, int pe, double m) const
	{
		double a = m, b = m, c = m, d = m;
		int i = p;
		for (; i + 4 <= pe; i += 4) {
			a = r2[i] < a ? r2[i] : a; b = r2[i + 1] < b ? r2[i + 1] : b;
			c = r2[i + 2] < c ? r2[i + 2] : c; d = r2[i + 3] < d ? r2[i + 3] : d;
		}
		for (; i < pe; i++) a = r2[i] < a ? r2[i] : a;
		a = b < a ? b : a; c = d < c ? d : c;
		return c < a ? c : a;
	}
};

// the caller's generator in blocks
struct DoubleSource : BlockData {
	MT *mt;
	int pos = 0;                 // doubles consumed of the block (always even: they go in pairs)
	bool filled = false;
	explicit DoubleSource(MT *m) : mt(m) { start = *m; }
	void refill()
	{
		start = *mt;
		mt_fill_doubles(mt, buf, kBlockDoubles);
		pos = 0;
		filled = true;
		analyse();
	}
};

// ---- blocks made AHEAD by two helper threads (the noise bounds of MUSE-style draws, BandLook) ----------
// The generator is sequential, but what is done to its output is not its business: thread 1 runs mt19937 and
// turns words into doubles (0.74 ns per double), thread 2 tests the pairs (r2, accepted bytes, prefix counts:
// 0.3 ns), the caller only scans (binary search + a vector minimum).  Blocks travel through a ring of slots,
// each handed from one stage to the next with a release / acquire flag; `restart` parks both helpers, empties
// the ring and points thread 1 at a new state (a candidate was accepted, or another constrainer's stream is
// wanted).  The blocks are a pure function of the state they start from: timing changes nothing but time.
// Opt-in (MDNS_BAND_THREADS=1): measured 1.9 ns per deviate against 2.1-2.4 in the caller alone -- a block is 15 KB
// that crosses two cores' caches, and the scan does not wait long for the generator -- so the default stays the
// caller alone (tools/probes/band_advance_bench.cpp).
struct ProducedBlock : BlockData {
	MT end;                      // the generator behind the block
	std::atomic<int> stage{0};   // 0 empty (thread 1's), 1 doubles made (thread 2's), 2 analysed (the caller's)
};

struct BlockProducer {
	static constexpr int kSlots = 48;
	ProducedBlock ring[kSlots];
	std::atomic<unsigned> want{0};                 // generation the caller asked for
	std::atomic<unsigned> parked1{0}, parked2{0};  // ... the helpers have stopped for
	std::atomic<unsigned> go{0};                   // ... they may run
	std::atomic<bool> stop{false};
	MT state;                                      // where thread 1 starts generation `go`
	unsigned long long stream = 0;                 // whose stream the ring holds (BandLook::stream)
	void *holder = nullptr;                        // ... and who may still have a slot of it in hand
	void (*evict)(void *) = nullptr;               // tells him to keep a copy: the ring is about to start over
	int take = 0;                                  // the caller's next slot
	pthread_t t1, t2;
	bool started = false, failed = false;

	// (no PAUSE in the spins: under the hypervisor of these boxes a loop of them traps -- a flag passed between
	// two threads took 6.2 us with it and 0.14 us without)
	static void idle(unsigned &spins)
	{
		if (++spins < 200000) return;
		struct timespec ts = {0, spins < 400000 ? 20000 : 200000};
		nanosleep(&ts, nullptr);
	}
	// a helper with nothing to do for generation `mine`: parks when a new one is wanted; false when it moved on
	bool settle(unsigned &mine, std::atomic<unsigned> &parked, int &at)
	{
		const unsigned w = want.load(std::memory_order_acquire);
		if (w == mine) return true;
		parked.store(w, std::memory_order_release);
		unsigned spins = 0;
		while (go.load(std::memory_order_acquire) != w && !stop.load(std::memory_order_relaxed)) idle(spins);
		mine = w;
		at = 0;
		return false;
	}
	void run1()
	{
		unsigned mine = 0, spins = 0;
		int at = 0;
		MT mt{};
		while (!stop.load(std::memory_order_relaxed)) {
			if (!settle(mine, parked1, at)) { mt = state; spins = 0; continue; }
			if (mine == 0) { idle(spins); continue; }
			ProducedBlock &b = ring[at % kSlots];
			if (b.stage.load(std::memory_order_acquire) != 0) { idle(spins); continue; }
			spins = 0;
			b.start = mt;
			mt_fill_doubles(&mt, b.buf, kBlockDoubles);
			b.end = mt;
			b.stage.store(1, std::memory_order_release);
			at++;
		}
	}
	void run2()
	{
		unsigned mine = 0, spins = 0;
		int at = 0;
		while (!stop.load(std::memory_order_relaxed)) {
			if (!settle(mine, parked2, at)) { spins = 0; continue; }
			ProducedBlock &b = ring[at % kSlots];
			if (mine == 0 || b.stage.load(std::memory_order_acquire) != 1) { idle(spins); continue; }
			spins = 0;
			b.analyse();
			b.stage.store(2, std::memory_order_release);
			at++;
		}
	}
	static void *entry1(void *p) { ((BlockProducer *) p)->run1(); return nullptr; }
	static void *entry2(void *p) { ((BlockProducer *) p)->run2(); return nullptr; }
	bool start()
	{
		if (started || failed) return started;
		if (pthread_create(&t1, nullptr, entry1, this) != 0) { failed = true; return false; }
		if (pthread_create(&t2, nullptr, entry2, this) != 0) { stop.store(true); pthread_join(t1, nullptr); failed = true; return false; }
		started = true;
		return true;
	}
	// the ring emptied and thread 1 pointed at `from`
	void restart(const MT &from, unsigned long long whose, void *who, void (*keep)(void *))
	{
		if (holder && holder != who && evict) evict(holder);
		holder = who;
		evict = keep;
		const unsigned g = want.load(std::memory_order_relaxed) + 1;
		want.store(g, std::memory_order_release);
		unsigned spins = 0;
		while (parked1.load(std::memory_order_acquire) != g || parked2.load(std::memory_order_acquire) != g) {
			if (++spins >= 2000000) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
		}
		for (int i = 0; i < kSlots; i++) ring[i].stage.store(0, std::memory_order_relaxed);
		state = from;
		stream = whose;
		take = 0;
		go.store(g, std::memory_order_release);
	}
	// the next analysed block (the previous one goes back to thread 1)
	const ProducedBlock *next(const ProducedBlock *done)
	{
		if (done) const_cast<ProducedBlock *>(done)->stage.store(0, std::memory_order_release);
		ProducedBlock &b = ring[take % kSlots];
		unsigned spins = 0;
		while (b.stage.load(std::memory_order_acquire) != 2) { if (++spins >= 20000000) { struct timespec ts = {0, 5000}; nanosleep(&ts, nullptr); } }
		take++;
		return &b;
	}
};

BlockProducer *block_producer()
{
	static int mode = -1;                          // 0 off, 1 on
	static BlockProducer *p = nullptr;
	if (mode < 0) {
		const char *e = getenv("MDNS_BAND_THREADS");
		mode = e && e[0] == '1' ? 1 : 0;
	}
	if (!mode) return nullptr;
	if (!p) p = new BlockProducer();               // (lives as long as the process: its threads never see it go)
	return p->start() ? p : nullptr;
}

// the stream `pos` doubles behind `start`
void mt_place(MT *mt, const MT &start, int pos)
{
	*mt = start;
	if (pos > 0) { double skip[kBlockDoubles]; mt_fill_doubles(mt, skip, (size_t) pos); }
}

// out[i] = legacy_gauss(s), i = 0 .. n-1: the same numbers, the same cache and the same place in the stream
// afterwards.  Pairs are tested a block at a time, the logarithms and roots of the accepted ones follow in a
// flat loop (glibc's scalar log, as numpy calls it).
void gauss_fill(MT *s, double *out, size_t n)
{
	size_t i = 0;
	if (n < 48) { for (; i < n; i++) out[i] = legacy_gauss(s); return; }
	if (g_has_gauss) { out[i++] = g_gauss; g_has_gauss = 0; g_gauss = 0.0; }
	DoubleSource src(s);
	int idx[kBlockPairs];
	double f[kBlockPairs];
	while (i < n) {
		src.refill();
		const size_t want = (n - i + 1) / 2;                  // accepted pairs still needed
		const int have = src.total();
		const int take = (size_t) have < want ? have : (int) want;
		const int pe = (size_t) have < want ? kBlockPairs : src.end_of(0, take);   // (ends BEHIND its last accepted pair)
		int k = 0;
		for (int p = 0; p < pe; p++) { idx[k] = p; k += src.ok[p]; }
		for (int t = 0; t < take; t++) { const double r2 = src.r2[idx[t]]; f[t] = std::sqrt(-2.0 * std::log(r2) / r2); }
		for (int t = 0; t < take; t++) {
			const double x1 = 2.0 * src.buf[2 * idx[t]] - 1.0, x2 = 2.0 * src.buf[2 * idx[t] + 1] - 1.0;
			out[i++] = f[t] * x2;
			if (i < n) out[i++] = f[t] * x1;
			else { g_gauss = f[t] * x1; g_has_gauss = 1; }
		}
		if (pe < kBlockPairs) mt_place(s, src.start, 2 * pe);
	}
}

// RandomState.randint(0, K, size=n) (random_bounded_uint64_fill, masked rejection, 32-bit draws)
void legacy_randint(MT *s, int64_t K, int n, int32_t *out)
{
	const uint32_t top = (uint32_t) (K - 1);
	if (top == 0) { for (int i = 0; i < n; i++) out[i] = 0; return; }
	uint32_t cover = top;
	cover |= cover >> 1; cover |= cover >> 2; cover |= cover >> 4; cover |= cover >> 8; cover |= cover >> 16;
	for (int i = 0; i < n; i++) {
		uint32_t v;
		do { v = mt_next(s) & cover; } while (v > top);
		out[i] = (int32_t) v;
	}
}

// numpy's pairwise summation of a contiguous run (numpy/_core/src/umath/loops_utils.h.src):
// what `a.sum(axis=-1)` performs per row, added to the identity 0.0
double pairwise_sum(const double *a, int n)
{
	if (n < 8) {
		double res = 0.;
		for (int i = 0; i < n; i++) res += a[i];
		return res;
	}
	if (n <= 128) {
		double r[8];
		for (int j = 0; j < 8; j++) r[j] = a[j];
		int i;
		for (i = 8; i < n - (n % 8); i += 8)
			for (int j = 0; j < 8; j++) r[j] += a[i + j];
		double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
		for (; i < n; i++) res += a[i];
		return res;
	}
	int n2 = n / 2;
	n2 -= n2 % 8;
	return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

// ---------------------------------------------------------------------------------------
// metric (clustering/sdml.py)
// ---------------------------------------------------------------------------------------
struct Metric {
	bool identity = true;
	std::vector<double> mean, scale;
	// (x - mean) / scale
	void transform(const double *x, int n, int ndim, double *w) const
	{
		if (identity) { memcpy(w, x, (size_t) n * ndim * sizeof(double)); return; }
		for (int i = 0; i < n; i++)
			for (int k = 0; k < ndim; k++) {
				const double d = x[(size_t) i * ndim + k] - mean[k];
				w[(size_t) i * ndim + k] = d / scale[k];
			}
	}
	// y * scale + mean
	void untransform(const double *y, int n, int ndim, double *x) const
	{
		if (identity) { memcpy(x, y, (size_t) n * ndim * sizeof(double)); return; }
		for (int i = 0; i < n; i++)
			for (int k = 0; k < ndim; k++) {
				const double p = y[(size_t) i * ndim + k] * scale[k];
				x[(size_t) i * ndim + k] = p + mean[k];
			}
	}
};

// ---------------------------------------------------------------------------------------
// region (clustering/radfriendsregion.py)
// ---------------------------------------------------------------------------------------
// index of a counter in mdns_constrainer_stats' output
enum { N_DRAWS, N_CHUNKS, N_CANDIDATES, N_PAIRS, N_REGIONS, N_RADII, N_COUNTS, N_PROPOSALS, N_INSIDE, N_TRIES,
       // nanoseconds spent in: the bootstrap choice, region_create (K6 + upload), region_count (K3),
       // proposal arithmetic + random numbers, prior transform, draw_chunk, the whole draw call
       T_BOOTSTRAP, T_REGION, T_COUNT, T_PROPOSE, T_TRANSFORM, T_CHUNK, T_DRAW, T_JITTER,
       // first batches chained on the device: with their chunk / counts only; accepted candidates whose device
       // parameters were not bit for bit the host's (10**v: mdns_pow10.h); nanoseconds from chain_begin to chain_end
       N_CHAINS, N_CHAIN_COUNTS, N_PARAM_MISMATCH, T_CHAIN,
       // jitter in band form: pairs the device could not decide without their noise, candidates whose noise was replayed for them
       N_BAND_PAIRS, N_BAND_REPLAYS,
       // candidates whose bound was ready before their chunk (made while the previous chunk was scored)
       N_BAND_AHEAD, N_COUNTERS };

inline long long now_ns()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (long long) ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

struct Counters {
	long long v[N_COUNTERS] = {0};
	long long *totals = nullptr;           // optional: the same counters summed over all constrainers of a sampler
	void add(int which, long long n)
	{
		v[which] += n;
		if (totals) totals[which] += n;
	}
};

struct Region {
	std::shared_ptr<const std::vector<double>> points;   // [K, ndim], the metric's coordinates (shared: the regions a
	                                                     // cluster() call makes of the same live points hold ONE copy)
	const double *members = nullptr;      // = points->data()
	int K = 0, ndim = 0;
	bool has_radius = false;
	double radius = 0;
	std::vector<uint32_t> masks;          // the bootstrap choice, drawn at construction, until K6 has run
	int nbootstraps = 10;
	void *handle = nullptr;               // the backend's region (members resident), made when first needed
	const mdns_draw_backend *be = nullptr;
	std::vector<double> lo, hi;
	bool has_box = false;
	Counters *stat = nullptr;

	~Region() { if (handle && be) be->region_destroy(be->user, handle); }

	bool started = false;                 // K6 is running (region_begin): region_radius delivers

	// K6 launched, nobody waits: what the caller does until it asks for `maxdistance` overlaps with it
	// (backends without region_begin compute the radius when it is asked for)
	bool start()
	{
		if (has_radius || started || !be->region_begin || !be->region_radius) return true;
		const long long t0 = now_ns();
		handle = be->region_begin(be->user, members, K, ndim, masks.data(), nbootstraps);
		if (stat) stat->add(T_REGION, now_ns() - t0);
		if (!handle) { set_error("region_begin (K6) failed for %d points", K); return false; }
		started = true;
		return true;
	}
	// `maxdistance` of the reference: the bootstrapped radius, computed when first asked for
	bool maxdistance(double *out)
	{
		if (!has_radius) {
			double r = 0;
			const long long t0 = now_ns();
			if (started) {
				if (be->region_radius(be->user, handle, &r) != 0) { set_error("region_radius (K6) failed for %d points", K); return false; }
			} else handle = be->region_create(be->user, members, K, ndim, masks.data(), nbootstraps, &r);
			if (stat) stat->add(T_REGION, now_ns() - t0);
			if (!handle) { set_error("region_create (K6) failed for %d points", K); return false; }
			radius = r;
			has_radius = true;
			std::vector<uint32_t>().swap(masks);
			if (stat) stat->add(N_RADII, 1);
		}
		*out = radius;
		return true;
	}
	bool member_set()
	{
		double r;
		if (!maxdistance(&r)) return false;
		if (!handle) {
			const long long t0 = now_ns();
			handle = be->region_create(be->user, members, K, ndim, nullptr, 0, &r);
			if (stat) stat->add(T_REGION, now_ns() - t0);
			if (!handle) { set_error("region_create failed for %d points", K); return false; }
		}
		return true;
	}
	bool count(const double *points, int n, int *counts)
	{
		if (!member_set()) return false;
		if (stat) stat->add(N_COUNTS, 1);
		const long long t0 = now_ns();
		const int rc = be->region_count(be->user, handle, points, n, counts);
		if (stat) stat->add(T_COUNT, now_ns() - t0);
		if (rc != 0) { set_error("region_count failed"); return false; }
		return true;
	}
	// lo = min(members) - maxdistance, hi = max(members) + maxdistance (radfriendsregion.py:69-70)
	bool box()
	{
		if (has_box) return true;
		double r;
		if (!maxdistance(&r)) return false;
		lo.assign(ndim, 0.0);
		hi.assign(ndim, 0.0);
		for (int k = 0; k < ndim; k++) { lo[k] = members[k]; hi[k] = members[k]; }
		for (int i = 1; i < K; i++)
			for (int k = 0; k < ndim; k++) {
				const double v = members[(size_t) i * ndim + k];
				if (v < lo[k]) lo[k] = v;
				if (v > hi[k]) hi[k] = v;
			}
		for (int k = 0; k < ndim; k++) { lo[k] = lo[k] - r; hi[k] = hi[k] + r; }
		has_box = true;
		return true;
	}
};

typedef std::shared_ptr<Region> RegionRef;

}  // namespace

// ---------------------------------------------------------------------------------------
// the constrainer (hiermetriclearn.py)
// ---------------------------------------------------------------------------------------
struct mdns_constrainer {
	int ndim = 0;
	int metriclearner = MDNS_METRIC_TRUNCATEDSCALING;
	int rebuild_every = 50, metric_rebuild_every = 50;
	bool force_shrink = false;
	Metric metric;
	RegionRef region;                      // `self.region` (may be empty: None)
	bool has_prev = false;                 // `self.prev_maxdistance is not None`
	double prev_maxdistance = 0;
	RegionRef prev_region;                 // set: prev_maxdistance is THIS region's radius, not asked for yet
	bool has_last = false;                 // `self.last_cluster_points`
	std::vector<double> last_cluster_points;
	int last_K = 0;
	long long iter_since_metric_rebuild = 0, ndraws_since_rebuild = 0;
	bool direct_draws_efficient = true;
	long long last_ntoaccept = 1;
	// the generator of hiermetriclearn.py:104-137 with the one of radfriendsregion.py:117-182 inside,
	// as a state machine
	bool has_generator = false;
	enum Phase { START_ROUND, BOX, BALL, COIN } phase = START_ROUND;
	RegionRef gen_region;                  // the region whose generate() is running
	double gen_maxdistance = 0;            // captured at its first step (radfriendsregion.py:118-120)
	bool gen_started = false;
	long long ntotal = 0, spent = 0, proposed = 0;
	// candidates proposed and not yet consumed
	std::vector<double> buf;
	int buf_n = 0, buf_pos = 0;
	long long buf_ntotal = 0;
	bool has_buf = false;
	Counters stat;
	// scratch
	std::vector<double> us, ws, dir, rad, coin, coins, xs, params, wtmp;
	std::vector<int32_t> idx;
	std::vector<int> counts;
	// the draw in progress (what the chained first batch needs to know of the accept loop's state)
	long long cur_tries = 0;
	bool cur_region_rebuilt = false, cur_metric_rebuilt = false;
	int cur_M = 0;
	bool in_draw = false;
	// outcome of a first chunk that rode along with the region's first batch (chain_begin / chain_end)
	bool chain_valid = false;
	int chain_B = 0, chain_accepted = -1;
	std::vector<unsigned long long> chain_bits;
	std::vector<double> chain_params;
	int chain_nkept = -1;
	// the band form of the likelihood jitter (band_chunk)
	std::shared_ptr<void> look;                // BandLook (band_chunk)
	std::vector<double> band_bound, band_row, band_pL, band_pthr;
	std::vector<int> band_status, band_pb, band_pk;
	// likelihood jitter of a chunk and the stream's state after each candidate's share of it
	struct Snapshot { MT mt; int has_gauss; double gauss; };
	std::vector<double> jitter;
	std::vector<Snapshot> snap;
};

namespace {

const int REGION_BATCH = 10000;          // hiermetriclearn.py:106
const int BATCH = 1000;                  // radfriendsregion.py:124
const double CUBE_PROBABILITY = 0.1;     // hiermetriclearn.py:126

struct Env {
	mdns_constrainer *c;
	const mdns_draw_backend *be;
	const mdns_prior *prior;
	const mdns_numpy_ops *np;
	MT *mt;
};

// RadFriendsRegion(members, maxdistance=None | value) (radfriendsregion.py:59-70)
// `discarded`: the caller replaces this region before anybody can ask it for anything -- its
// bootstrap choice is drawn from the stream (the position of the draws is part of the results) but
// not kept
typedef std::shared_ptr<const std::vector<double>> Points;

RegionRef new_region(Env &e, const Points &members, int K, bool given, double maxdistance, bool discarded = false)
{
	RegionRef r = std::make_shared<Region>();
	r->points = members;
	r->members = members->data();
	r->K = K;
	r->ndim = e.c->ndim;
	r->be = e.be;
	r->stat = &e.c->stat;
	e.c->stat.add(N_REGIONS, 1);
	if (given) {
		r->has_radius = true;
		r->radius = maxdistance;
	} else {
		// nbootstraps x numpy.random.choice(arange(K), size=K) NOW (neighbors.py:173-174): the
		// position of these draws in the stream is part of the results; K6 itself, which draws
		// nothing, waits until somebody asks for the radius
		if (!discarded) r->masks.assign(K, 0u);
		const long long t0 = now_ns();
		const int rc = discarded ? mdns_host_bootstrap_skip_mt(e.mt, K, r->nbootstraps)
		                         : mdns_host_bootstrap_masks_mt(e.mt, K, r->nbootstraps, r->masks.data());
		e.c->stat.add(T_BOOTSTRAP, now_ns() - t0);
		if (rc != 0) {
			set_error("bootstrap choice for %d points failed", K);
			return RegionRef();
		}
	}
	return r;
}

// `force_shrink`: a rebuilt region may not have a larger radius than the one it replaces
// (hiermetriclearn.py:53-54,88-90); `prev_maxdistance is None` counts as smaller than anything
// (Python 2) and then the replacement draws its own bootstrap choice
// `self.prev_maxdistance`: the radius of the region the last cluster() installed, whose K6 may
// still be running
bool prev_value(mdns_constrainer *c)
{
	if (!c->prev_region) return true;
	double r;
	if (!c->prev_region->maxdistance(&r)) return false;
	c->prev_maxdistance = r;
	c->prev_region.reset();
	return true;
}

RegionRef never_grow(Env &e, RegionRef region, const Points &members_old_metric, int K)
{
	mdns_constrainer *c = e.c;
	if (!c->force_shrink) return region;
	if (c->has_prev) {
		if (!prev_value(c)) return RegionRef();
		double r;
		if (!region->maxdistance(&r)) return RegionRef();
		if (!(r > c->prev_maxdistance)) return region;
		return new_region(e, members_old_metric, K, true, c->prev_maxdistance);
	}
	return new_region(e, members_old_metric, K, false, 0);
}

// cluster(u, keepMetric) (hiermetriclearn.py:48-92)
bool cluster(Env &e, const double *u, int K, bool keepMetric)
{
	mdns_constrainer *c = e.c;
	const int ndim = c->ndim;
	auto w_old_store = std::make_shared<std::vector<double>>((size_t) K * ndim);
	c->metric.transform(u, K, ndim, w_old_store->data());
	const Points w_old = w_old_store;
	RegionRef region;
	if (keepMetric) {
		// (with force_shrink and no previous radius never_grow replaces it at once, hiermetriclearn.py:53-54)
		region = new_region(e, w_old, K, false, 0, c->force_shrink && !c->has_prev);
		if (!region) return false;
		region = never_grow(e, region, w_old, K);
		if (!region) return false;
	} else {
		bool changed = false;
		if (c->metriclearner != MDNS_METRIC_NONE) {
			Metric m;
			m.identity = false;
			m.mean.assign(ndim, 0.0);
			m.scale.assign(ndim, 1.0);
			if (!e.np || !e.np->fit_metric ||
			    e.np->fit_metric(e.np->user, c->metriclearner, u, K, ndim, m.mean.data(), m.scale.data()) != 0) {
				set_error("fit_metric failed");
				return false;
			}
			if (c->metriclearner == MDNS_METRIC_SIMPLESCALING) changed = true;
			else {
				changed = c->metric.identity;
				if (!changed)
					for (int k = 0; k < ndim; k++) if (c->metric.scale[k] != m.scale[k]) changed = true;
			}
			c->metric = m;
		}
		auto w_new = std::make_shared<std::vector<double>>((size_t) K * ndim);
		c->metric.transform(u, K, ndim, w_new->data());
		region = new_region(e, Points(w_new), K, false, 0);
		if (!region) return false;
		// only a region in the SAME metric is comparable with the previous radius
		if (!changed && c->has_prev) {
			region = never_grow(e, region, w_old, K);
			if (!region) return false;
		}
	}
	c->region = region;
	// self.prev_maxdistance = self.region.maxdistance (hiermetriclearn.py:91): K6 starts here and
	// the value is taken when somebody needs it -- the proposals of the generator draw their random
	// numbers first (next_batch_inner)
	if (!region->start()) return false;
	c->prev_region = region;
	c->has_prev = true;
	return true;
}

void reset_buffer(mdns_constrainer *c) { c->has_buf = false; c->buf_n = 0; c->buf_pos = 0; c->buf_ntotal = 0; }

// rebuild(u, keepMetric) (hiermetriclearn.py:139-150)
// (`points` is given up when a new region is built -- it becomes `last_cluster_points` without
// another copy -- and `*taken` says so: a later rebuild of the same draw with the same points
// would find them identical and do nothing, hiermetriclearn.py:141-143)
bool rebuild(Env &e, std::vector<double> &points, int K, bool keepMetric, bool *taken)
{
	mdns_constrainer *c = e.c;
	const double *u = points.data();
	const size_t n = (size_t) K * c->ndim;
	if (c->has_last && c->last_K == K && memcmp(c->last_cluster_points.data(), u, n * sizeof(double)) == 0) {
		// (memcmp equality is numpy's element-wise == for the finite unit-cube points of a pile)
		return true;                        // identical live points: keep region AND generator
	}
	if (!cluster(e, u, K, keepMetric)) return false;
	c->last_cluster_points.swap(points);
	*taken = true;
	c->last_K = K;
	c->has_last = true;
	c->has_generator = true;
	c->phase = mdns_constrainer::START_ROUND;
	c->gen_region.reset();
	c->gen_started = false;
	c->ntotal = 0;
	reset_buffer(c);
	return true;
}

// hands the candidates ws[n, ndim] (metric coordinates) of one region.generate() yield to the outer
// generator (hiermetriclearn.py:111-119); true when that yields a batch
bool deliver(Env &e, const double *ws, int n, long long nspent)
{
	mdns_constrainer *c = e.c;
	const int ndim = c->ndim;
	c->wtmp.resize((size_t) n * ndim);
	c->metric.untransform(ws, n, ndim, c->wtmp.data());
	c->ntotal = c->ntotal + nspent;
	c->buf.clear();
	int kept = 0;
	for (int i = 0; i < n; i++) {
		bool inside = true;
		for (int k = 0; k < ndim; k++) {
			const double v = c->wtmp[(size_t) i * ndim + k];
			if (!(v < 1 && v > 0)) inside = false;
		}
		if (inside) {
			c->buf.insert(c->buf.end(), c->wtmp.begin() + (size_t) i * ndim, c->wtmp.begin() + (size_t) (i + 1) * ndim);
			kept++;
		}
	}
	if (!kept) return false;
	c->buf_n = kept;
	c->buf_pos = 0;
	c->buf_ntotal = c->ntotal;
	c->has_buf = true;
	c->ntotal = 0;
	return true;
}

// how many candidates the next chunk of the draw in progress may hold when `room` are on offer: never
// past the candidate after which the reference would rebuild its region (hiermetriclearn.py:198-211)
int chunk_room(Env &e, long long room)
{
	mdns_constrainer *c = e.c;
	if (!c->cur_region_rebuilt) {
		long long lim = c->rebuild_every - c->ndraws_since_rebuild + 1;
		if (lim < 1) lim = 1;
		if (room > lim) room = lim;
	}
	if (!c->cur_metric_rebuilt) {
		long long lim = 201 - c->cur_tries;
		if (lim < 1) lim = 1;
		if (room > lim) room = lim;
	}
	int B = (int) (room > 0x3fffffff ? 0x3fffffff : room);
	if (e.be->chunk_size) {
		const long long hint = c->last_ntoaccept > 1 ? c->last_ntoaccept : 1;
		const int got = e.be->chunk_size(e.be->user, B, c->cur_M, (int) (hint > 0x3fffffff ? 0x3fffffff : hint));
		if (got < 1 || got > B) return -1;
		B = got;
	}
	return B;
}

// next(self.generator): fills the buffer with the next batch of candidates (consumes RNG, exactly
// on demand)
bool next_batch_inner(Env &e);

// (time spent here minus the membership kernels = random numbers + proposal arithmetic)
bool next_batch(Env &e)
{
	const long long t0 = now_ns(), k0 = e.c->stat.v[T_COUNT] + e.c->stat.v[T_REGION];
	const bool ok = next_batch_inner(e);
	e.c->stat.add(T_PROPOSE, now_ns() - t0 - (e.c->stat.v[T_COUNT] + e.c->stat.v[T_REGION] - k0));
	return ok;
}

bool next_batch_inner(Env &e)
{
	mdns_constrainer *c = e.c;
	const int ndim = c->ndim, N = BATCH;
	if (!c->has_generator) { set_error("draw without a generator"); return false; }
	for (;;) {
		switch (c->phase) {
		case mdns_constrainer::START_ROUND:
			if (ndim < 40) {
				if (!c->region) { set_error("the constrainer's region was dropped while its generator is in use (the reference raises AttributeError here)"); return false; }
				c->gen_region = c->region;
				c->gen_started = false;
				c->spent = 0;
				c->proposed = 0;
				c->phase = mdns_constrainer::BOX;
			} else c->phase = mdns_constrainer::COIN;
			break;
		case mdns_constrainer::BOX: {
			Region *r = c->gen_region.get();
			// numpy.random.uniform(lo, hi, size=(N, ndim)): lo + (hi - lo) * double, row by row.  The
			// doubles do not depend on the box: a generator that starts draws them BEFORE it waits
			// for the radius of its region, whose K6 cluster() has just launched (nothing else takes
			// numbers from the stream in between, and a generator that starts always proposes:
			// `proposed` is 0)
			const bool starting = !c->gen_started;
			bool chained = false;
			if (starting) {
				c->us.resize((size_t) N * ndim);
				mt_fill_doubles(e.mt, c->us.data(), (size_t) N * ndim);
				// K6 of this region is running and the backend can carry on by itself: the proposals
				// from these doubles, their membership counts and -- inside a draw -- the first chunk are
				// queued behind it and the host waits ONCE (mdns.h, chain_begin / chain_end)
				if (r->started && !r->has_radius && r->handle && e.be->chain_begin && e.be->chain_end && !e.prior->custom &&
				    ndim <= 5 && N <= 1024) {
					double mn[MDNS_MAX_DIM], mx[MDNS_MAX_DIM];
					for (int k = 0; k < ndim; k++) { mn[k] = r->members[k]; mx[k] = r->members[k]; }
					for (int i = 1; i < r->K; i++)
						for (int k = 0; k < ndim; k++) {
							const double v = r->members[(size_t) i * ndim + k];
							if (v < mn[k]) mn[k] = v;
							if (v > mx[k]) mx[k] = v;
						}
					int limit = 0;
					if (c->in_draw) {
						limit = chunk_room(e, N);
						if (limit < 0) { set_error("chunk_size failed"); return false; }
					}
					mdns_chain_request rq;
					rq.n = N; rq.ndim = ndim; rq.u = c->us.data(); rq.mn = mn; rq.mx = mx;
					rq.identity = c->metric.identity ? 1 : 0;
					rq.mean = c->metric.identity ? nullptr : c->metric.mean.data();
					rq.scale = c->metric.identity ? nullptr : c->metric.scale.data();
					rq.prior = e.prior; rq.limit = limit;
					const long long t0 = now_ns();
					if (e.be->chain_begin(e.be->user, r->handle, &rq) != 0) { set_error("chain_begin failed"); return false; }
					c->counts.resize(N);
					c->chain_bits.resize((size_t) (c->cur_M + 63) / 64 + 1);
					c->chain_params.resize((size_t) 1024 * 3);
					int nkept = -1, B = 0, accepted = -1;
					if (e.be->chain_end(e.be->user, r->handle, c->counts.data(), &nkept, &B, &accepted, c->chain_bits.data(),
					                    c->chain_params.data()) != 0) { set_error("chain_end failed"); return false; }
					c->stat.add(T_CHAIN, now_ns() - t0);
					chained = true;
					c->stat.add(N_COUNTS, 1);
					if (nkept >= 0) {
						c->stat.add(N_CHAINS, 1);
						c->chain_valid = B > 0;
						c->chain_B = B;
						c->chain_accepted = accepted;
						c->chain_nkept = nkept;          // (checked against the host's own count below)
					} else {
						c->stat.add(N_CHAIN_COUNTS, 1);
						c->chain_nkept = -1;
					}
				}
				// like the reference, the ball proposals keep the members and the radius the generator
				// started with (radfriendsregion.py:118-120)
				if (!r->maxdistance(&c->gen_maxdistance)) return false;
				if (c->prev_region.get() == r) { c->prev_maxdistance = c->gen_maxdistance; c->prev_region.reset(); }
				c->gen_started = true;
			}
			if (!(c->proposed < REGION_BATCH)) {
				c->gen_region.reset();
				c->phase = mdns_constrainer::COIN;
				break;
			}
			c->spent += N;
			c->proposed += N;
			c->stat.add(N_PROPOSALS, N);
			if (!r->box()) return false;
			if (!starting) {
				c->us.resize((size_t) N * ndim);
				mt_fill_doubles(e.mt, c->us.data(), (size_t) N * ndim);
			}
			double range[MDNS_MAX_DIM];
			for (int k = 0; k < ndim; k++) range[k] = r->hi[k] - r->lo[k];
			for (int i = 0; i < N; i++)
				for (int k = 0; k < ndim; k++) {
					const double t = range[k] * c->us[(size_t) i * ndim + k];
					c->us[(size_t) i * ndim + k] = r->lo[k] + t;
				}
			if (!chained) {
				c->counts.resize(N);
				if (!r->count(c->us.data(), N, c->counts.data())) return false;
			}
			c->phase = mdns_constrainer::BALL;
			c->ws.clear();
			int n = 0;
			for (int i = 0; i < N; i++)
				if (c->counts[i] > 0) {
					c->ws.insert(c->ws.end(), c->us.begin() + (size_t) i * ndim, c->us.begin() + (size_t) (i + 1) * ndim);
					n++;
				}
			c->stat.add(N_INSIDE, n);
			bool got = false;
			if (n) {
				const long long sp = c->spent;
				c->spent = 0;
				got = deliver(e, c->ws.data(), n, sp);
			}
			if (chained && c->chain_nkept >= 0 && (got ? c->buf_n : 0) != c->chain_nkept) {
				set_error("chain: the device kept %d of the proposals, the host %d", c->chain_nkept, got ? c->buf_n : 0);
				return false;
			}
			if (got) return true;
			c->chain_valid = false;
			break;
		}
		case mdns_constrainer::BALL: {
			Region *r = c->gen_region.get();
			// members[numpy.random.randint(0, K, N)]
			c->idx.resize(N);
			legacy_randint(e.mt, r->K, N, c->idx.data());
			c->spent += N;
			c->proposed += N;
			c->stat.add(N_PROPOSALS, N);
			// direction = normal(0, 1, (N, ndim)); direction / sqrt((direction ** 2).sum(axis=1))
			c->dir.resize((size_t) N * ndim);
			gauss_fill(e.mt, c->dir.data(), (size_t) N * ndim);
			for (size_t t = 0; t < (size_t) N * ndim; t++) {
				const double g = 1.0 * c->dir[t];
				c->dir[t] = 0.0 + g;
			}
			double sq[MDNS_MAX_DIM];
			for (int i = 0; i < N; i++) {
				double *d = &c->dir[(size_t) i * ndim];
				for (int k = 0; k < ndim; k++) sq[k] = d[k] * d[k];
				const double s = 0.0 + pairwise_sum(sq, ndim);
				const double norm = std::sqrt(s);
				for (int k = 0; k < ndim; k++) d[k] = d[k] / norm;
			}
			// radius = maxdistance * uniform(0, 1, (N, 1)) ** (1. / ndim)
			c->rad.resize(N);
			mt_fill_doubles(e.mt, c->rad.data(), (size_t) N);
			for (int i = 0; i < N; i++) {
				const double t = 1.0 * c->rad[i];
				c->rad[i] = 0.0 + t;
			}
			// numpy's scalar-exponent fast paths: ** 1.0 leaves the values, ** 0.5 is sqrt
			const double expo = 1. / ndim;
			if (expo == 1.0) {
			} else if (expo == 0.5) {
				for (int i = 0; i < N; i++) c->rad[i] = std::sqrt(c->rad[i]);
			} else {
				if (!e.np || !e.np->vec_pow) { set_error("vec_pow missing"); return false; }
				e.np->vec_pow(e.np->user, c->rad.data(), N, expo);
			}
			for (int i = 0; i < N; i++) c->rad[i] = c->gen_maxdistance * c->rad[i];
			// us = centres + direction * radius
			c->us.resize((size_t) N * ndim);
			for (int i = 0; i < N; i++)
				for (int k = 0; k < ndim; k++) {
					const double p = c->dir[(size_t) i * ndim + k] * c->rad[i];
					c->us[(size_t) i * ndim + k] = r->members[(size_t) c->idx[i] * ndim + k] + p;
				}
			c->counts.resize(N);
			if (!r->count(c->us.data(), N, c->counts.data())) return false;
			// accept = uniform(size=N) < 1. / nnear
			c->ws.clear();
			int n = 0;
			c->coins.resize(N);
			mt_fill_doubles(e.mt, c->coins.data(), (size_t) N);
			for (int i = 0; i < N; i++) {
				const double t = 1.0 * c->coins[i];
				const double coin = 0.0 + t;
				const double inv = 1. / (double) c->counts[i];          // 1/0 = inf: accepted (never occurs: the centre is within reach)
				if (coin < inv) {
					c->ws.insert(c->ws.end(), c->us.begin() + (size_t) i * ndim, c->us.begin() + (size_t) (i + 1) * ndim);
					n++;
				}
			}
			c->phase = mdns_constrainer::BOX;
			c->stat.add(N_INSIDE, n);
			if (n) {
				const long long sp = c->spent;
				c->spent = 0;
				if (deliver(e, c->ws.data(), n, sp)) return true;
			}
			break;
		}
		case mdns_constrainer::COIN: {
			c->phase = mdns_constrainer::START_ROUND;
			const double t = 1.0 * mt_double(e.mt);
			const double coin = 0.0 + t;
			if (coin < CUBE_PROBABILITY) {
				// occasionally propose from the whole unit cube (hiermetriclearn.py:126-137)
				const int NN = REGION_BATCH;
				c->ntotal = c->ntotal + NN;
				c->stat.add(N_PROPOSALS, NN);
				c->us.resize((size_t) NN * ndim);
				mt_fill_doubles(e.mt, c->us.data(), (size_t) NN * ndim);
				for (size_t q = 0; q < (size_t) NN * ndim; q++) {
					const double v = 1.0 * c->us[q];
					c->us[q] = 0.0 + v;
				}
				if (!c->region) { set_error("the constrainer's region was dropped while its generator is in use"); return false; }
				c->ws.resize((size_t) NN * ndim);
				c->metric.transform(c->us.data(), NN, ndim, c->ws.data());
				c->counts.resize(NN);
				if (!c->region->count(c->ws.data(), NN, c->counts.data())) return false;
				c->buf.clear();
				int n = 0;
				for (int i = 0; i < NN; i++)
					if (c->counts[i] > 0) {
						c->buf.insert(c->buf.end(), c->us.begin() + (size_t) i * ndim, c->us.begin() + (size_t) (i + 1) * ndim);
						n++;
					}
				c->stat.add(N_INSIDE, n);
				if (n) {
					c->buf_n = n;
					c->buf_pos = 0;
					c->buf_ntotal = c->ntotal;
					c->has_buf = true;
					c->ntotal = 0;
					return true;
				}
			}
			break;
		}
		}
	}
}

// priortransform + kernel parameters of candidates us[B, ndim]
void transform(const mdns_prior *p, const double *us, int B, double *xs, double *params)
{
	if (p->custom) { p->custom(p->user, us, B, xs, params); return; }
	const int ndim = p->ndim;
	for (int i = 0; i < B; i++)
		for (int k = 0; k < ndim; k++) {
			double v = p->a[k] * us[(size_t) i * ndim + k];
			if (p->b[k] != 0.0) v = v + p->b[k];
			if (p->pow10[k]) v = std::pow(10.0, v);
			xs[(size_t) i * ndim + k] = v;
			if (k < p->nparams) params[(size_t) i * p->nparams + k] = p->kernel_pow10[k] ? std::pow(10.0, v) : v;
		}
}

// ---------------------------------------------------------------------------------------
// the likelihood noise of musefuse.py:535 WITHOUT drawing every deviate
// ---------------------------------------------------------------------------------------
// `Lout[mask] + numpy.random.normal(0, 1e-5, size=mask.sum())` consumes one Gaussian deviate per
// (candidate, data set) evaluation -- billions per run -- and only pairs whose likelihood lies within a
// few 1e-5 of the threshold can be decided by it.  numpy's legacy generator makes a PAIR of deviates
// f x1, f x2 from two uniforms with r2 = x1^2 + x2^2 < 1, f = sqrt(-2 ln(r2) / r2): their magnitude is
// at most sqrt(-2 ln r2).  So the stream is only ADVANCED -- the uniforms and the rejection test, no
// logarithm, no root -- and each candidate gets a rigorous bound on its deviates from its smallest r2.
// The device decides every pair outside  threshold +- bound  without the noise and lists the rest
// (draw_band); exact deviates are made, by replaying the candidate's part of the stream with
// legacy_gauss itself, only for candidates with listed pairs and for the accepted candidate's row,
// which is what the state keeps (draw_band_commit).  Decisions, kept values and the position of the
// stream are those of the deviate-per-evaluation path (tests/test_muse.py on the reference's traces).
// a place in the stream: `pos` doubles into block `block` of a BandLook, with numpy's cached deviate
struct BandSnap { long long block; int pos; int has_gauss; double gauss; };

// The candidates of a batch take their noise from the stream one after the other, M deviates each, and
// nothing else draws from it between two chunks of the same batch: the bound of a candidate depends on its
// place behind the chunk's first one only.  So the bounds are made by a generator of their own that runs AHEAD
// of the caller's: while a chunk is being scored it goes on with the candidates that follow in the batch, and
// the next chunk -- if this one accepts nobody -- finds its bounds made.  The caller's stream is only ever
// PUT to a remembered place (band_restore); what was looked at beyond it is dropped.
struct BandLook {
	bool valid = false;
	int M = 0;
	double sigma = 0.0;
	MT own;                              // the generator that runs ahead: the state BEHIND the block in hand
	BlockData local;                     // its blocks when the caller makes them itself
	const BlockData *blk = &local;       // the block in hand: `local`, or a slot of the helpers' ring
	const ProducedBlock *slot = nullptr; // ... that slot
	int pos = 0;                         // doubles consumed of it (always even: they go in pairs)
	bool filled = false;
	unsigned long long stream = 0;       // which of the helpers' streams this one is (0: none yet)
	int has_gauss = 0;                   // numpy's cached second deviate behind the last candidate looked at
	double gauss = 0.0;
	size_t base = 0;                     // entries before `base` belong to chunks that are done
	std::deque<MT> blocks;               // the generator at the start of the blocks it has made: number block_first onwards
	long long block_first = 0;
	std::vector<BandSnap> snap;          // snap[base + i]: the stream before candidate i; one more than bounds
	std::vector<double> bound;           // bound[base + i]
	MT expect;                           // the caller's stream where candidate 0 starts: what it must still be
	int expect_has = 0;
	double expect_gauss = 0.0;
	size_t count() const { return bound.size() - base; }
	void reset(const MT *mt, int M_, double sigma_)
	{
		own = *mt;
		drop_slot();
		stream = 0;
		pos = 0; filled = false;
		has_gauss = g_has_gauss; gauss = g_gauss;
		snap.clear(); bound.clear(); blocks.clear(); base = 0; block_first = 0;
		blocks.push_back(own);
		snap.push_back(here());
		M = M_; sigma = sigma_;
	}
	// the slot in hand goes back (to a ring that still holds this stream)
	void drop_slot()
	{
		if (slot) const_cast<ProducedBlock *>(slot)->stage.store(0, std::memory_order_release);
		slot = nullptr;
		blk = &local;
	}
	// the ring is about to start over with another stream: what is in hand is kept as a copy
	static void keep_copy(void *self)
	{
		BandLook *L = (BandLook *) self;
		if (!L->slot) return;
		L->local = *static_cast<const BlockData *>(L->slot);
		L->slot = nullptr;
		L->blk = &L->local;
	}
	~BandLook()
	{
		if (!slot && stream == 0) return;
		BlockProducer *bp = block_producer();
		if (bp && bp->holder == this) { drop_slot(); bp->holder = nullptr; }
	}
	// the next block of the stream behind `own`
	void next_block()
	{
		static unsigned long long streams = 0;
		BlockProducer *bp = block_producer();
		if (bp) {
			if (stream == 0 || bp->stream != stream) {
				// the ring holds somebody else's stream (or none): it starts over from where this one stands
				slot = nullptr;
				blk = &local;
				stream = ++streams;
				bp->restart(own, stream, this, &BandLook::keep_copy);
			}
			slot = bp->next(slot);
			blk = slot;
			own = slot->end;
		} else {
			local.start = own;
			mt_fill_doubles(&own, local.buf, kBlockDoubles);
			local.analyse();
			blk = &local;
		}
		pos = 0;
		filled = true;
		blocks.push_back(blk->start);
	}
	BandSnap here() const { return BandSnap{block_first + (long long) blocks.size() - 1, filled ? pos : 0, has_gauss, gauss}; }
	// the caller's stream (and numpy's cache) at a remembered place
	void restore(MT *mt, const BandSnap &at) const
	{
		mt_place(mt, blocks[(size_t) (at.block - block_first)], at.pos);
		g_has_gauss = at.has_gauss;
		g_gauss = at.gauss;
	}
	// forget what lies before snap[base]
	void compact()
	{
		while (block_first < snap[base].block) { blocks.pop_front(); block_first++; }
		if (count() == 0 || base > 8192) {
			snap.erase(snap.begin(), snap.begin() + (long) base);
			bound.erase(bound.begin(), bound.begin() + (long) base);
			base = 0;
		}
	}
	bool matches(const MT *mt, int M_, double sigma_) const
	{
		return valid && M == M_ && sigma == sigma_ && expect_has == g_has_gauss &&
		       (!expect_has || memcmp(&expect_gauss, &g_gauss, sizeof(double)) == 0) && memcmp(&expect, mt, sizeof(MT)) == 0;
	}
	// one more candidate: M deviates further, the largest of them bounded from the smallest r2 of its pairs
	// (an accepted pair gives two deviates, f x2 and then, from the cache, f x1)
	void advance()
	{
		double minr2 = 2.0, cached_abs = 0.0;
		int need = M;
		if (has_gauss && need > 0) { cached_abs = std::fabs(gauss); has_gauss = 0; gauss = 0.0; need--; }
		const bool odd = need & 1;
		int pairs = (need + 1) / 2;
		while (pairs > 0) {
			if (!filled || pos == kBlockDoubles) next_block();
			const BlockData &src = *blk;
			const int p = pos >> 1;
			const int avail = src.total() - src.before(p);
			int pe = kBlockPairs;
			if (avail < pairs) pairs -= avail;
			else { pe = src.end_of(p, pairs); pairs = 0; }
			minr2 = src.min_r2(p, pe, minr2);
			pos = 2 * pe;
			if (pairs == 0 && odd) {
				// the last pair's other deviate waits in the cache beyond this candidate: its value is needed
				const double r2 = src.r2[pe - 1];
				const double x1 = 2.0 * src.buf[2 * (pe - 1)] - 1.0;
				const double f = std::sqrt(-2.0 * std::log(r2) / r2);
				gauss = f * x1;
				has_gauss = 1;
			}
		}
		double most = cached_abs;
		if (minr2 < 2.0) { const double g = std::sqrt(-2.0 * std::log(minr2)); if (g > most) most = g; }
		bound.push_back(sigma * most);
		snap.push_back(here());
	}
};

// 0: done (*accepted, fillbits, stream positioned); 1: failed; 2: not possible for this chunk (stream
// back where it was: the caller draws the whole block)
int band_chunk(Env &e, const double *params, int B, int M, int *accepted, unsigned long long *fillbits)
{
	mdns_constrainer *c = e.c;
	const mdns_draw_backend *be = e.be;
	const double sigma = e.prior->jitter_sigma;
	MT *mt = e.mt;
	const long long t0 = now_ns();
	if (!c->look) c->look = std::make_shared<BandLook>();
	BandLook &L = *static_cast<BandLook *>(c->look.get());
	if (L.matches(mt, M, sigma)) c->stat.add(N_BAND_AHEAD, (long long) (L.count() < (size_t) B ? L.count() : (size_t) B));
	else L.reset(mt, M, sigma);
	L.valid = false;                                    // (until this chunk ends without an accepted candidate)
	while (L.count() < (size_t) B) L.advance();
	const long long t_band = now_ns();
	c->stat.add(T_JITTER, t_band - t0);
	const int cap = 4096;
	c->band_status.resize(B);
	c->band_pb.resize(cap); c->band_pk.resize(cap); c->band_pL.resize(cap); c->band_pthr.resize(cap);
	int npairs = 0;
	if (be->draw_band_begin && be->draw_band_ready && be->draw_band_end) {
		if (be->draw_band_begin(be->user, params, B, &L.bound[L.base]) != 0) { set_error("draw_band failed"); return 1; }
		// what follows in the batch (the next chunk starts there if this one accepts nobody)
		long long ahead = (long long) c->buf_n - c->buf_pos - B;
		if (ahead > 2048) ahead = 2048;
		while ((long long) L.count() < B + ahead && !be->draw_band_ready(be->user)) L.advance();
		if (be->draw_band_end(be->user, c->band_status.data(), &npairs, c->band_pb.data(), c->band_pk.data(), c->band_pL.data(),
		                      c->band_pthr.data(), cap) != 0) { set_error("draw_band failed"); return 1; }
	} else if (be->draw_band(be->user, params, B, &L.bound[L.base], c->band_status.data(), &npairs, c->band_pb.data(), c->band_pk.data(),
	                         c->band_pL.data(), c->band_pthr.data(), cap) != 0) { set_error("draw_band failed"); return 1; }
	const BandSnap *snap = &L.snap[L.base];
	if (npairs > cap) { L.restore(mt, snap[0]); return 2; }
	const long long t1 = now_ns();
	c->stat.add(T_CHUNK, t1 - t_band);
	// the exact noise of candidate b: its part of the stream again, through legacy_gauss itself
	auto replay = [&](int b) {
		L.restore(mt, snap[b]);
		c->band_row.resize(M);
		gauss_fill(mt, c->band_row.data(), (size_t) M);
		for (int k = 0; k < M; k++) {
			const double g = sigma * c->band_row[k];
			c->band_row[k] = 0.0 + g;
		}
	};
	int bstar = -1;
	for (int b = 0; b < B && bstar < 0; b++) {
		const int st = c->band_status[b];
		if (st == 0) continue;
		if (st == 1) { bstar = b; break; }
		replay(b);
		for (int t = 0; t < npairs; t++) {
			if (c->band_pb[t] != b) continue;
			const int k = c->band_pk[t];
			if (k < 0 || k >= M) { set_error("draw_band: pair of data set %d of %d", k, M); return 1; }
			const double v = c->band_pL[t] + c->band_row[k];
			if (v > c->band_pthr[t]) { bstar = b; break; }
		}
		c->stat.add(N_BAND_REPLAYS, 1);
	}
	c->stat.add(N_BAND_PAIRS, npairs);
	if (bstar >= 0) {
		replay(bstar);
		if (be->draw_band_commit(be->user, bstar, c->band_row.data(), fillbits) != 0) { set_error("draw_band_commit failed"); return 1; }
		L.restore(mt, snap[(size_t) bstar + 1]);
	} else {
		L.restore(mt, snap[B]);
		// the candidates looked at beyond this chunk stay, for a chunk that starts exactly here
		L.base += (size_t) B;
		L.compact();
		L.expect = *mt;
		L.expect_has = g_has_gauss;
		L.expect_gauss = g_gauss;
		L.valid = true;
	}
	c->stat.add(T_JITTER, now_ns() - t1);
	*accepted = bstar;
	return 0;
}

}  // namespace

extern "C" const char *mdns_host_last_error(void) { return g_error; }

extern "C" void mdns_host_rng_get_gauss(int *has_gauss, double *gauss)
{
	if (has_gauss) *has_gauss = g_has_gauss;
	if (gauss) *gauss = g_gauss;
}

extern "C" void mdns_host_rng_set_gauss(int has_gauss, double gauss)
{
	g_has_gauss = has_gauss ? 1 : 0;
	g_gauss = has_gauss ? gauss : 0.0;
}

extern "C" mdns_constrainer *mdns_constrainer_create(int ndim, int metriclearner, int rebuild_every,
                                                     int metric_rebuild_every, int force_shrink)
{
	if (ndim <= 0 || ndim > MDNS_MAX_DIM || metriclearner < 0 || metriclearner > 2) {
		set_error("mdns_constrainer_create: ndim=%d (1..%d), metriclearner=%d", ndim, MDNS_MAX_DIM, metriclearner);
		return nullptr;
	}
	mdns_constrainer *c = new mdns_constrainer();
	c->ndim = ndim;
	c->metriclearner = metriclearner;
	c->rebuild_every = rebuild_every;
	c->metric_rebuild_every = metric_rebuild_every;
	c->force_shrink = force_shrink != 0;
	return c;
}

extern "C" void mdns_constrainer_destroy(mdns_constrainer *c, const mdns_draw_backend *be)
{
	(void) be;                                  // regions carry the backend they were made with
	delete c;
}

extern "C" void mdns_constrainer_forget_region(mdns_constrainer *c)
{
	if (c) c->region.reset();
}

extern "C" void mdns_constrainer_stats(const mdns_constrainer *c, long long *out)
{
	if (!c || !out) return;
	memcpy(out, c->stat.v, sizeof c->stat.v);
}

extern "C" void mdns_constrainer_share_stats(mdns_constrainer *c, long long *totals)
{
	if (c) c->stat.totals = totals;
}

static int constrainer_draw(mdns_constrainer *c, const mdns_draw_backend *be, const mdns_prior *prior,
                            const mdns_numpy_ops *np, void *mt19937_state,
                            const double *pile_u, const void *ids, int ids_itemsize, int K,
                            const int *rows, int M,
                            double *u_out, double *x_out, long long *ntries, unsigned long long *fillbits);

extern "C" int mdns_constrainer_draw(mdns_constrainer *c, const mdns_draw_backend *be, const mdns_prior *prior,
                                     const mdns_numpy_ops *np, void *mt19937_state,
                                     const double *pile_u, const void *ids, int ids_itemsize, int K,
                                     const int *rows, int M,
                                     double *u_out, double *x_out, long long *ntries, unsigned long long *fillbits)
{
	const long long t0 = now_ns();
	const int rc = constrainer_draw(c, be, prior, np, mt19937_state, pile_u, ids, ids_itemsize, K, rows, M, u_out, x_out, ntries, fillbits);
	if (c) c->stat.add(T_DRAW, now_ns() - t0);
	return rc;
}

static int constrainer_draw(mdns_constrainer *c, const mdns_draw_backend *be, const mdns_prior *prior,
                            const mdns_numpy_ops *np, void *mt19937_state,
                            const double *pile_u, const void *ids, int ids_itemsize, int K,
                            const int *rows, int M,
                            double *u_out, double *x_out, long long *ntries, unsigned long long *fillbits)
{
	if (!c || !be || !prior || !mt19937_state || !pile_u || !ids || K <= 0 || M <= 0 || !u_out || !x_out || !ntries || !fillbits) {
		set_error("mdns_constrainer_draw: bad arguments (K=%d M=%d)", K, M);
		return 1;
	}
	if (prior->ndim != c->ndim || prior->nparams <= 0 || prior->nparams > MDNS_MAX_DIM || (ids_itemsize != 4 && ids_itemsize != 8)) {
		set_error("mdns_constrainer_draw: prior of %d dimensions for a constrainer of %d (ids of %d bytes)", prior->ndim, c->ndim, ids_itemsize);
		return 1;
	}
	MT *mt = (MT *) mt19937_state;
	if (mt->pos < 0 || mt->pos > 624) { set_error("mdns_constrainer_draw: not an mt19937 state (pos=%d)", mt->pos); return 1; }
	Env e = {c, be, prior, np, mt};
	const int ndim = c->ndim;
	// live_pointsu = pointpile[ids]
	std::vector<double> u((size_t) K * ndim);
	for (int i = 0; i < K; i++) {
		const long long id = ids_itemsize == 4 ? (long long) ((const int32_t *) ids)[i] : (long long) ((const int64_t *) ids)[i];
		memcpy(&u[(size_t) i * ndim], pile_u + (size_t) id * ndim, (size_t) ndim * sizeof(double));
	}
	c->stat.add(N_DRAWS, 1);
	// rebuild policy at the start of a draw (hiermetriclearn.py:152-166)
	c->iter_since_metric_rebuild += 1;
	const bool region_due = !c->region || c->ndraws_since_rebuild > c->rebuild_every;
	const bool metric_due = c->iter_since_metric_rebuild > c->metric_rebuild_every;
	bool region_rebuilt = false, metric_rebuilt = false;
	bool u_taken = false;                  // `u` went into the constrainer (the points of its current region)
	if (region_due) {
		if (!rebuild(e, u, K, !metric_due, &u_taken)) return 1;
		c->ndraws_since_rebuild = 0;
		if (metric_due) c->iter_since_metric_rebuild = 0;
		region_rebuilt = true;
		metric_rebuilt = metric_due;
	}
	if (!c->has_generator) { set_error("mdns_constrainer_draw: no generator"); return 1; }
	if (be->draw_begin && be->draw_begin(be->user, rows, M) != 0) { set_error("draw_begin failed"); return 1; }
	struct InDraw { mdns_constrainer *c; ~InDraw() { c->in_draw = false; c->chain_valid = false; } } in_draw_guard = {c};
	c->in_draw = true;
	c->chain_valid = false;
	c->cur_M = M;
	// the accept loop of hiermetriclearn.py:181-211 with the candidates handed over in chunks: a
	// chunk never reaches past the candidate after which the reference would rebuild its region
	// (:198-211), so regions, RNG draws and results are those of the one-candidate-at-a-time loop
	long long tries = 0;
	for (;;) {
		c->cur_tries = tries;
		c->cur_region_rebuilt = region_rebuilt;
		c->cur_metric_rebuilt = metric_rebuilt;
		if (!c->has_buf || c->buf_pos >= c->buf_n) {
			c->chain_valid = false;
			if (!next_batch(e)) return 1;
			if (c->buf_ntotal > 100000) c->direct_draws_efficient = false;
		} else c->chain_valid = false;
		const long long room = c->buf_n - c->buf_pos;
		const int B = chunk_room(e, room);
		if (B < 1 || B > room) { set_error("chunk_size returned %d of %lld", B, room); return 1; }
		const double *chunk = &c->buf[(size_t) c->buf_pos * ndim];
		c->xs.resize((size_t) B * ndim);
		c->params.resize((size_t) B * prior->nparams);
		const long long t0 = now_ns();
		transform(prior, chunk, B, c->xs.data(), c->params.data());
		const long long t1 = now_ns();
		// the tie-breaking noise of musefuse.py:535, candidate by candidate, with the state of the
		// stream remembered after each of them
		const double *jitter = nullptr;
		int band = 2;                                   // 2: the noise goes as a block (or there is none)
		int band_accepted = -1;
		if (prior->jitter_sigma > 0 && be->draw_band && be->draw_band_commit && !c->chain_valid) {
			band = band_chunk(e, c->params.data(), B, M, &band_accepted, fillbits);
			if (band == 1) return 1;
		}
		if (prior->jitter_sigma > 0 && band == 2) {
			c->jitter.resize((size_t) B * M);
			c->snap.resize(B);
			for (int b = 0; b < B; b++) {
				double *row = &c->jitter[(size_t) b * M];
				gauss_fill(mt, row, (size_t) M);
				for (int k = 0; k < M; k++) {
					const double g = prior->jitter_sigma * row[k];
					row[k] = 0.0 + g;
				}
				c->snap[b].mt = *mt;
				c->snap[b].has_gauss = g_has_gauss;
				c->snap[b].gauss = g_gauss;
			}
			jitter = c->jitter.data();
			c->stat.add(T_JITTER, now_ns() - t1);
		}
		const long long t2 = now_ns();
		int accepted = -1, nscored = B;
		int rc_chunk = 0;
		if (c->chain_valid) {
			// this chunk rode along with the region's first batch: scored and committed already
			c->chain_valid = false;
			if (c->chain_B != B || c->buf_pos != 0) { set_error("chain: a chunk of %d was scored, the host's is %d (position %d)", c->chain_B, B, c->buf_pos); return 1; }
			accepted = c->chain_accepted;
			if (accepted >= 0) {
				memcpy(fillbits, c->chain_bits.data(), (size_t) ((M + 63) / 64) * sizeof(unsigned long long));
				// the parameters the device scored the accepted candidate with against the host's own
				if (memcmp(&c->chain_params[(size_t) accepted * 3], &c->params[(size_t) accepted * prior->nparams], 3 * sizeof(double)) != 0)
					c->stat.add(N_PARAM_MISMATCH, 1);
			}
		} else if (band == 0) accepted = band_accepted;
		else rc_chunk = be->draw_chunk(be->user, c->params.data(), B, jitter, &accepted, fillbits, &nscored);
		if (jitter && rc_chunk == 0) {
			// the reference evaluated exactly the candidates up to the accepted one (or all `nscored`)
			const int last = accepted >= 0 ? accepted : nscored - 1;
			if (last >= 0 && last < B - 1) {
				*mt = c->snap[last].mt;
				g_has_gauss = c->snap[last].has_gauss;
				g_gauss = c->snap[last].gauss;
			}
		}
		c->stat.add(T_TRANSFORM, t1 - t0);
		c->stat.add(T_CHUNK, now_ns() - t2);
		if (rc_chunk != 0) { set_error("draw_chunk failed"); return 1; }
		c->stat.add(N_CHUNKS, 1);
		c->stat.add(N_CANDIDATES, nscored);
		c->stat.add(N_PAIRS, (long long) nscored * M);
		const long long used = accepted >= 0 ? accepted + 1 : nscored;
		if (used <= 0 || used > B || accepted >= B) { set_error("draw_chunk: accepted %d, scored %d of %d", accepted, nscored, B); return 1; }
		tries += used;
		c->stat.add(N_TRIES, used);
		c->ndraws_since_rebuild += used;
		c->buf_pos += (int) used;
		if (accepted >= 0) {
			c->last_ntoaccept = tries;
			memcpy(u_out, chunk + (size_t) accepted * ndim, (size_t) ndim * sizeof(double));
			memcpy(x_out, &c->xs[(size_t) accepted * ndim], (size_t) ndim * sizeof(double));
			*ntries = tries;
			return 0;
		}
		// a long unsuccessful streak tightens the region -- each kind at most once per draw
		// (hiermetriclearn.py:198-211); the candidate stream restarts from the new region
		if (!region_rebuilt && c->ndraws_since_rebuild > c->rebuild_every) {
			region_rebuilt = true;
			if (!u_taken && !rebuild(e, u, K, true, &u_taken)) return 1;
			c->ndraws_since_rebuild = 0;
		} else if (!metric_rebuilt && tries > 200) {
			metric_rebuilt = true;
			if (!u_taken && !rebuild(e, u, K, false, &u_taken)) return 1;
			c->iter_since_metric_rebuild = 0;
		}
	}
}
