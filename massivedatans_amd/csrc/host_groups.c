/* Host-side helper (plain C, no GPU): the grouping of data sets that share live points,
 * multi_nested_sampler.py:237-266 of the reference (generate_subsets_nograph), restated over
 * arrays.  Late in a run this walk is called a hundred times per nested-sampling iteration, on
 * anything from one data set to a component of thousands; in Python it was two thirds of the
 * wall-clock.
 *
 * The result has to be what the reference's walk produces, ORDER included (the order of a
 * group's points decides which of them the bootstrap rounds leave out):
 *   - groups in the order of their first (lowest-index) data set;
 *   - a group's points: the live points of its first data set in live-slot order, then, for
 *     every listed point in turn, the not yet listed live points of the data sets it brings
 *     in (those still unplaced that hold it), ascending and each once.
 * Integer work only.  The cost of a call is proportional to the SELECTED data sets: the
 * per-id work arrays belong to the caller, arrive zeroed and are handed back zeroed (only the
 * entries of the ids the selection holds are touched). */
#include <stdint.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

static int cmp_i64(const void *a, const void *b)
{
	const int64_t x = *(const int64_t *) a, y = *(const int64_t *) b;
	return (x > y) - (x < y);
}

/* ascending sort of n non-negative ids; `tmp` has room for n.  Byte-wise counting sort over as
 * many bytes as the largest id needs (the big batches -- a component's first point can bring in
 * a thousand data sets with 10^5 new ids -- made qsort the most expensive part of the walk). */
static void sort_ids(int64_t *v, int64_t n, int64_t *tmp)
{
	if (n < 2) return;
	if (n < 192) { qsort(v, (size_t) n, sizeof(int64_t), cmp_i64); return; }
	int64_t top = 0;
	for (int64_t i = 0; i < n; i++) if (v[i] > top) top = v[i];
	int64_t *src = v, *dst = tmp;
	for (int shift = 0; shift < 64 && (top >> shift) != 0; shift += 8) {
		int64_t count[257];
		memset(count, 0, sizeof count);
		for (int64_t i = 0; i < n; i++) count[((src[i] >> shift) & 255) + 1]++;
		for (int b = 0; b < 256; b++) count[b + 1] += count[b];
		for (int64_t i = 0; i < n; i++) dst[count[(src[i] >> shift) & 255]++] = src[i];
		int64_t *t = src; src = dst; dst = t;
	}
	if (src != v) memcpy(v, src, (size_t) n * sizeof(int64_t));
}

/* lpT       int64[ndata][nlive]   live-point ids, one ROW per data set (the transpose of the
 *                                 sampler's matrix: the walk reads whole data sets)
 * mask      uint8[ndata]          the data sets to group
 * npoints                         ids are in [0, npoints)
 * cnt       int32[npoints]  work  zero on entry and on return: holders of an id in the selection
 * first     int64[npoints]  work  any content: where an id's holder list starts
 * known     uint8[npoints]  work  zero on entry and on return
 * group_of  int32[ndata]   out    group index of every selected data set, -1 for the others
 * points    int64[cap]     out    the groups' point lists, one after the other
 * offsets   int64[ndata+1] out    group g owns points[offsets[g] : offsets[g+1]]
 * distinct  int64[npoints] out    the distinct ids held by the selection (numpy.unique of the
 *                                 selected columns), ascending if `sorted_distinct` or if there
 *                                 are fewer than `sort_below` of them; *ndistinct their number
 * Returns the number of groups, -1 if out of memory, -2 if `cap` is too small (cap >= the
 * number of distinct ids + nlive is enough when the ids of a data set are distinct). */
int mdns_host_group_walk(const int64_t *lpT, int nlive, int ndata, const uint8_t *mask, int64_t npoints,
                         int32_t *cnt, int64_t *first, uint8_t *known,
                         int32_t *group_of, int64_t *points, int64_t cap, int64_t *offsets,
                         int64_t *distinct, int64_t *ndistinct, int sorted_distinct, int64_t sort_below)
{
	(void) npoints;
	uint8_t *todo = (uint8_t *) malloc((size_t) ndata);
	int32_t *holders = NULL, *sel = NULL;
	int64_t *fresh = NULL, *sort_tmp = NULL;
	int ngroups = -1;
	int64_t nt = 0;                                   /* ids touched = distinct ids */
	if (!todo) goto done;
	int64_t nsel = 0;
	for (int d = 0; d < ndata; d++) { todo[d] = mask[d] != 0; group_of[d] = -1; nsel += todo[d]; }
	sel = (int32_t *) malloc((size_t) (nsel > 0 ? nsel : 1) * sizeof(int32_t));
	holders = (int32_t *) malloc((size_t) (nsel > 0 ? nsel : 1) * nlive * sizeof(int32_t));
	fresh = (int64_t *) malloc((size_t) (nsel > 0 ? nsel : 1) * nlive * sizeof(int64_t));
	sort_tmp = (int64_t *) malloc((size_t) (nsel > 0 ? nsel : 1) * nlive * sizeof(int64_t));
	if (!sel || !holders || !fresh || !sort_tmp) goto done;
	{
		int64_t j = 0;
		for (int d = 0; d < ndata; d++) if (todo[d]) sel[j++] = d;               /* ascending */
	}
	/* holders per id, the ids in the order they are first met (`distinct`) */
	for (int64_t j = 0; j < nsel; j++) {
		const int64_t *ids = lpT + (size_t) sel[j] * nlive;
		for (int k = 0; k < nlive; k++) {
			if (cnt[ids[k]]++ == 0) distinct[nt++] = ids[k];
		}
	}
	*ndistinct = nt;
	/* An id with a single holder in the selection can never bring anybody in (its holder is the
	 * data set that listed it): late in a run that is nine ids in ten.  Only shared ids get a
	 * holder list. */
	{
		int64_t at = 0;
		for (int64_t t = 0; t < nt; t++)
			if (cnt[distinct[t]] >= 2) { first[distinct[t]] = at; at += cnt[distinct[t]]; }
	}
	/* the running fill position is kept in first[] itself and taken back afterwards */
	for (int64_t j = 0; j < nsel; j++) {
		const int64_t *ids = lpT + (size_t) sel[j] * nlive;
		for (int k = 0; k < nlive; k++)
			if (cnt[ids[k]] >= 2) holders[first[ids[k]]++] = sel[j];
	}
	for (int64_t t = 0; t < nt; t++)
		if (cnt[distinct[t]] >= 2) first[distinct[t]] -= cnt[distinct[t]];      /* back to the list starts */

	{
		int64_t used = 0, left = nsel;
		int next_first = 0;
		ngroups = 0;
		offsets[0] = 0;
		while (left > 0) {
			while (!todo[next_first]) next_first++;
			const int lead = next_first;
			todo[lead] = 0; left--;
			group_of[lead] = ngroups;
			const int64_t begin = used;
			if (used + nlive > cap) { ngroups = -2; goto done; }
			for (int k = 0; k < nlive; k++) {
				const int64_t p = lpT[(size_t) lead * nlive + k];
				points[used++] = p;
				known[p] = 1;
			}
			for (int64_t i = begin; i < used && left > 0; i++) {
				const int64_t p = points[i];
				if (cnt[p] < 2) continue;
				int64_t nnew = 0;
				for (int64_t h = first[p]; h < first[p] + cnt[p]; h++) {
					const int d = holders[h];
					if (todo[d]) { todo[d] = 0; left--; group_of[d] = ngroups; sel[nnew++] = d; }
				}
				if (!nnew) continue;
				int64_t nfresh = 0, lo = INT64_MAX, hi = -1;
				for (int64_t m = 0; m < nnew; m++) {
					const int64_t *ids = lpT + (size_t) sel[m] * nlive;
					for (int k = 0; k < nlive; k++) {
						const int64_t q = ids[k];
						if (!known[q]) {                                                        /* each once */
							known[q] = 2; fresh[nfresh++] = q;
							if (q < lo) lo = q;
							if (q > hi) hi = q;
						}
					}
				}
				/* ascending: a dense batch is read off the marks in id order, a sparse one is sorted */
				if (nfresh >= 192 && hi - lo < 24 * nfresh) {
					int64_t n = 0;
					for (int64_t q = lo; q <= hi; q++) if (known[q] == 2) { known[q] = 1; fresh[n++] = q; }
				} else {
					for (int64_t f = 0; f < nfresh; f++) known[fresh[f]] = 1;
					sort_ids(fresh, nfresh, sort_tmp);
				}
				if (used + nfresh > cap) { ngroups = -2; goto done; }
				memcpy(points + used, fresh, (size_t) nfresh * sizeof(int64_t));
				used += nfresh;
			}
			offsets[++ngroups] = used;
		}
	}
done:
	/* the work arrays go back zeroed (also after a failure) */
	for (int64_t t = 0; t < nt; t++) { cnt[distinct[t]] = 0; known[distinct[t]] = 0; }
	if (ngroups >= 0 && (sorted_distinct || nt < sort_below) && nt > 1 && sort_tmp)
		sort_ids(distinct, nt, sort_tmp);
	free(todo); free(holders); free(sel); free(fresh); free(sort_tmp);
	return ngroups;
}
