/* Host-side helpers (plain C, no GPU) around the two scalar-heavy spots of the Python
 * orchestration that have to stay bit-identical with the reference:
 *
 *  - the bootstrap choice of a RadFriends region (clustering/neighbors.py:170-174 of the
 *    reference: per round `numpy.random.choice(arange(K), size=K, replace=True)` on the GLOBAL
 *    legacy stream), drawn here straight from numpy's own bit generator through the C
 *    interface numpy publishes for that purpose (`BitGenerator.ctypes`), so the stream advances
 *    exactly as if numpy had made the calls;
 *  - `10 ** v` of the prior transform and of `sig = 10**log_sig` (sample.py:54,103): the C
 *    library's pow(), the function numpy's and Python's scalar power end in.
 */
#include <math.h>
#include <stdint.h>

/* numpy/random/bitgen.h */
typedef struct {
	void *state;
	uint64_t (*next_uint64)(void *st);
	uint32_t (*next_uint32)(void *st);
	double (*next_double)(void *st);
	uint64_t (*next_raw)(void *st);
} mdns_bitgen;

/* Legacy `RandomState.randint(0, K, size=(rounds, K))` -- which is what `choice(arange(K), K)`
 * per round amounts to, value by value -- packed: bit b of masks[i] is set when point i is
 * drawn in round b.  numpy's algorithm for a 32-bit range (distributions.c,
 * random_bounded_uint64_fill with use_masked): smallest bit mask covering K-1, one 32-bit
 * draw per attempt, rejected while above K-1.  K = 1 draws nothing.  masks must be zeroed by
 * the caller; rounds <= 32.  Returns 0, or 1 when the range needs numpy's 64-bit path. */
int mdns_host_bootstrap_masks(const mdns_bitgen *bg, int64_t K, int rounds, uint32_t *masks)
{
	if (K <= 0 || rounds < 0 || rounds > 32) return 1;
	const uint64_t top = (uint64_t) K - 1;
	if (top > 0xFFFFFFFEull) return 1;
	if (top == 0) {
		for (int b = 0; b < rounds; b++) masks[0] |= 1u << b;
		return 0;
	}
	uint32_t cover = (uint32_t) top;
	cover |= cover >> 1; cover |= cover >> 2; cover |= cover >> 4; cover |= cover >> 8; cover |= cover >> 16;
	void *st = bg->state;
	uint32_t (*next32)(void *) = bg->next_uint32;
	for (int b = 0; b < rounds; b++) {
		const uint32_t bit = 1u << b;
		for (int64_t i = 0; i < K; i++) {
			uint32_t v;
			do { v = next32(st) & cover; } while (v > (uint32_t) top);
			masks[v] |= bit;
		}
	}
	return 0;
}

/* The same draws with the Mersenne Twister stepped in place instead of one call through a
 * function pointer per number (the bootstrap choices of a 10 000-spectra run are 6e9 draws).  `state` must be numpy's `mt19937_state` -- { uint32_t key[624]; int pos; },
 * numpy/random/src/mt19937/mt19937.h -- which is what `bitgen_t.state` of the MT19937 bit
 * generator points at; the Python side checks this entry point against the one above on a copy
 * of the state before it trusts it (massivedatans_amd/_host.py). */
typedef struct { uint32_t key[624]; int pos; } mdns_mt19937;

/* (inlined into its callers so that their AVX2 clones get an AVX2 twist as well: the three loops
 * have no dependency shorter than 227 elements) */
static inline __attribute__((always_inline)) void mt_refill(mdns_mt19937 *s)
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

/* The inner loops are free of unpredictable branches (at K = 4600 almost every second draw is
 * rejected: 9 ns per draw with a branch, 3 with one branch-free loop, under 2 split in three).  A run of draws is never longer than the number of points still to be drawn, so
 * the generator stops exactly where numpy's one-at-a-time loop stops. */
#if defined(__x86_64__) && defined(__GNUC__)
#define MDNS_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))      /* wider tempering loops where the CPU has them */
#else
#define MDNS_CLONES
#endif

/* Tempering of n state words and the accepted values (<= top after masking with `cover`) packed
 * together, in draw order: returns how many.  AVX-512: sixteen words per step, the accepted ones
 * compressed to the front of a register (the scalar form spends a store and an add per word on it). */
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#include <stdlib.h>
__attribute__((target("avx512f")))
static int64_t temper_pack_avx512(const uint32_t *key, int64_t n, uint32_t cover, uint32_t top, uint32_t *taken)
{
	const __m512i c1 = _mm512_set1_epi32((int) 0x9d2c5680u), c2 = _mm512_set1_epi32((int) 0xefc60000u);
	const __m512i vcover = _mm512_set1_epi32((int) cover), vtop = _mm512_set1_epi32((int) top);
	int64_t got = 0;
	for (int64_t i = 0; i < n; i += 16) {
		const __mmask16 live = n - i >= 16 ? (__mmask16) 0xffff : (__mmask16) ((1u << (n - i)) - 1u);
		__m512i y = _mm512_maskz_loadu_epi32(live, key + i);
		y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
		y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 7), c1));
		y = _mm512_xor_si512(y, _mm512_and_si512(_mm512_slli_epi32(y, 15), c2));
		y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
		y = _mm512_and_si512(y, vcover);
		const __mmask16 ok = _mm512_mask_cmple_epu32_mask(live, y, vtop);
		/* (compress in a register + one full store: the store form of vpcompressd is microcoded on
		 * Zen 4/5; `taken` has sixteen words of slack) */
		_mm512_storeu_si512((void *) (taken + got), _mm512_maskz_compress_epi32(ok, y));
		got += __builtin_popcount((unsigned) ok);
	}
	return got;
}
static int have_avx512(void)
{
	static int known = -1;
	if (known < 0) {
		const char *off = getenv("MDNS_HOST_NO_AVX512");                 /* tests compare the two forms */
		known = (off && off[0] == '1') ? 0 : (__builtin_cpu_supports("avx512f") ? 1 : 0);
	}
	return known;
}
#else
static int have_avx512(void) { return 0; }
static int64_t temper_pack_avx512(const uint32_t *key, int64_t n, uint32_t cover, uint32_t top, uint32_t *taken)
{
	(void) key; (void) n; (void) cover; (void) top; (void) taken;
	return 0;
}
#endif

/* The same draws WITHOUT the choice: the stream is left where mdns_host_bootstrap_masks_mt would
 * leave it.  For regions whose bootstrap choice is drawn -- the reference draws it in the
 * constructor, clustering/radfriendsregion.py:62-64 -- and never used: the first region of a new
 * constrainer is replaced at once (hiermetriclearn.py:53-54), half of all regions of a run. */
MDNS_CLONES
int mdns_host_bootstrap_skip_mt(void *state, int64_t K, int rounds)
{
	mdns_mt19937 *s = (mdns_mt19937 *) state;
	if (!s || K <= 0 || rounds < 0 || rounds > 32 || s->pos < 0 || s->pos > 624) return 1;
	const uint64_t top64 = (uint64_t) K - 1;
	if (top64 > 0xFFFFFFFEull) return 1;
	if (top64 == 0) return 0;
	const uint32_t top = (uint32_t) top64;
	uint32_t cover = top;
	cover |= cover >> 1; cover |= cover >> 2; cover |= cover >> 4; cover |= cover >> 8; cover |= cover >> 16;
	for (int b = 0; b < rounds; b++) {
		int64_t need = K;
		while (need > 0) {
			if (s->pos == 624) mt_refill(s);
			int64_t n = 624 - s->pos;
			if (n > need) n = need;
			const uint32_t *key = s->key + s->pos;
			int64_t got = 0;
			for (int64_t i = 0; i < n; i++) {
				uint32_t y = key[i];
				y ^= (y >> 11);
				y ^= (y << 7) & 0x9d2c5680u;
				y ^= (y << 15) & 0xefc60000u;
				y ^= (y >> 18);
				got += (y & cover) <= top;
			}
			s->pos += (int) n;
			need -= got;
		}
	}
	return 0;
}

MDNS_CLONES
int mdns_host_bootstrap_masks_mt(void *state, int64_t K, int rounds, uint32_t *masks)
{
	mdns_mt19937 *s = (mdns_mt19937 *) state;
	if (!s || K <= 0 || rounds < 0 || rounds > 32 || s->pos < 0 || s->pos > 624) return 1;
	const uint64_t top64 = (uint64_t) K - 1;
	if (top64 > 0xFFFFFFFEull) return 1;
	if (top64 == 0) {
		for (int b = 0; b < rounds; b++) masks[0] |= 1u << b;
		return 0;
	}
	const uint32_t top = (uint32_t) top64;
	uint32_t cover = top;
	cover |= cover >> 1; cover |= cover >> 2; cover |= cover >> 4; cover |= cover >> 8; cover |= cover >> 16;
	uint32_t vals[624], taken[624 + 16];
	const int wide = have_avx512();
	for (int b = 0; b < rounds; b++) {
		const uint32_t bit = 1u << b;
		int64_t need = K;
		while (need > 0) {
			if (s->pos == 624) mt_refill(s);
			int64_t n = 624 - s->pos;
			if (n > need) n = need;
			const uint32_t *key = s->key + s->pos;
			/* three short loops instead of one: tempering (vectorised by the compiler), the
			 * accepted values packed together without a branch, then the scatter */
			int64_t got = 0;
			if (wide) got = temper_pack_avx512(key, n, cover, top, taken);
			else {
				for (int64_t i = 0; i < n; i++) {
					uint32_t y = key[i];
					y ^= (y >> 11);
					y ^= (y << 7) & 0x9d2c5680u;
					y ^= (y << 15) & 0xefc60000u;
					y ^= (y >> 18);
					vals[i] = y & cover;
				}
				for (int64_t i = 0; i < n; i++) {
					taken[got] = vals[i];
					got += vals[i] <= top;
				}
			}
			for (int64_t i = 0; i < got; i++) masks[taken[i]] |= bit;
			s->pos += (int) n;
			need -= got;
		}
	}
	return 0;
}

/* lo[k] = min_i a[i][k], hi[k] = max_i a[i][k] of a f64[n][ndim] (numpy.min / numpy.max along
 * axis 0: exact selections, so the same numbers) */
void mdns_host_minmax(const double *a, int64_t n, int ndim, double *lo, double *hi)
{
	for (int k = 0; k < ndim; k++) { lo[k] = a[k]; hi[k] = a[k]; }
	for (int64_t i = 1; i < n; i++) {
		const double *row = a + i * ndim;
		for (int k = 0; k < ndim; k++) {
			if (row[k] < lo[k]) lo[k] = row[k];
			if (row[k] > hi[k]) hi[k] = row[k];
		}
	}
}

/* out[i] = 10 ** in[i] */
void mdns_host_pow10(const double *in, int64_t n, double *out)
{
	for (int64_t i = 0; i < n; i++) out[i] = pow(10.0, in[i]);
}
