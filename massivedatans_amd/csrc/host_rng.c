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

/* out[i] = 10 ** in[i] */
void mdns_host_pow10(const double *in, int64_t n, double *out)
{
	for (int64_t i = 0; i < n; i++) out[i] = pow(10.0, in[i]);
}
