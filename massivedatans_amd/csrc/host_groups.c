/* Host-side helper (plain C, no GPU): the grouping of data sets that share live points,
 * multi_nested_sampler.py:237-266 of the reference (generate_subsets_nograph), restated over
 * arrays.  Late in a run this walk is called hundreds of times per nested-sampling iteration on
 * a component of a thousand data sets; in Python it was two thirds of the wall-clock.
 *
 * The result has to be what the reference's walk produces, ORDER included (the order of a
 * group's points decides which of them the bootstrap rounds leave out):
 *   - groups in the order of their first (lowest-index) data set;
 *   - a group's points: the live points of its first data set in live-slot order, then, for
 *     every listed point in turn, the not yet listed live points of the data sets it brings
 *     in (those still unplaced that hold it), ascending and each once.
 * Integer work only. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_i64(const void *a, const void *b)
{
	const int64_t x = *(const int64_t *) a, y = *(const int64_t *) b;
	return (x > y) - (x < y);
}

/* lpT       int64[ndata][nlive]   live-point ids, one ROW per data set (the transpose of the
 *                                 sampler's matrix: the walk reads whole data sets)
 * mask      uint8[ndata]          the data sets to group
 * npoints                         ids are in [0, npoints)
 * group_of  int32[ndata]   out    group index of every selected data set, -1 for the others
 * points    int64[cap]     out    the groups' point lists, one after the other
 * offsets   int64[ndata+1] out    group g owns points[offsets[g] : offsets[g+1]]
 * distinct  int64[npoints] out    the distinct ids held by the selection, ascending
 *                                 (numpy.unique of the selected columns); *ndistinct their number
 * Returns the number of groups, -1 if out of memory, -2 if `cap` is too small (cap >= the
 * number of distinct ids held by the selection is always enough). */
int mdns_host_group_walk(const int64_t *lpT, int nlive, int ndata, const uint8_t *mask, int64_t npoints,
                         int32_t *group_of, int64_t *points, int64_t cap, int64_t *offsets,
                         int64_t *distinct, int64_t *ndistinct)
{
	int64_t *start = (int64_t *) calloc((size_t) npoints + 1, sizeof(int64_t));   /* CSR: id -> holders */
	uint8_t *known = (uint8_t *) calloc((size_t) npoints, 1);
	uint8_t *todo = (uint8_t *) malloc((size_t) ndata);
	int32_t *holders = NULL, *fresh_members = NULL;
	int64_t *fresh = NULL;
	int ngroups = -1;
	if (!start || !known || !todo) goto done;
	int64_t nsel = 0;
	for (int d = 0; d < ndata; d++) { todo[d] = mask[d] != 0; group_of[d] = -1; nsel += todo[d]; }
	/* only the selected columns are ever touched: late in a run they are a tenth of the matrix */
	fresh_members = (int32_t *) malloc((size_t) (nsel > 0 ? nsel : 1) * sizeof(int32_t));
	if (!fresh_members) goto done;
	{
		int64_t j = 0;
		for (int d = 0; d < ndata; d++) if (todo[d]) fresh_members[j++] = d;      /* ascending */
	}
	for (int64_t j = 0; j < nsel; j++) {
		const int64_t *ids = lpT + (size_t) fresh_members[j] * nlive;
		for (int k = 0; k < nlive; k++) start[ids[k] + 1]++;
	}
	{
		int64_t n = 0;
		for (int64_t p = 0; p < npoints; p++) if (start[p + 1]) distinct[n++] = p;
		*ndistinct = n;
	}
	for (int64_t p = 0; p < npoints; p++) start[p + 1] += start[p];
	holders = (int32_t *) malloc((size_t) (start[npoints] > 0 ? start[npoints] : 1) * sizeof(int32_t));
	fresh = (int64_t *) malloc((size_t) (nsel > 0 ? nsel : 1) * nlive * sizeof(int64_t));
	int64_t *fill = (int64_t *) malloc((size_t) (npoints > 0 ? npoints : 1) * sizeof(int64_t));
	if (!holders || !fresh || !fill) { free(fill); goto done; }
	memcpy(fill, start, (size_t) npoints * sizeof(int64_t));
	for (int64_t j = 0; j < nsel; j++) {          /* ascending data sets: holder lists come out sorted */
		const int64_t *ids = lpT + (size_t) fresh_members[j] * nlive;
		for (int k = 0; k < nlive; k++) holders[fill[ids[k]]++] = fresh_members[j];
	}
	free(fill);

	int64_t used = 0, left = nsel;
	int next_first = 0;
	ngroups = 0;
	offsets[0] = 0;
	while (left > 0) {
		while (!todo[next_first]) next_first++;
		const int first = next_first;
		todo[first] = 0; left--;
		group_of[first] = ngroups;
		const int64_t begin = used;
		if (used + nlive > cap) { ngroups = -2; goto done; }
		for (int k = 0; k < nlive; k++) {
			const int64_t p = lpT[(size_t) first * nlive + k];
			points[used++] = p;
			known[p] = 1;
		}
		for (int64_t i = begin; i < used && left > 0; i++) {
			const int64_t p = points[i];
			int nnew = 0;
			for (int64_t h = start[p]; h < start[p + 1]; h++) {
				const int d = holders[h];
				if (todo[d]) { todo[d] = 0; left--; group_of[d] = ngroups; fresh_members[nnew++] = d; }
			}
			if (!nnew) continue;
			int64_t nfresh = 0;
			for (int m = 0; m < nnew; m++) {
				const int64_t *ids = lpT + (size_t) fresh_members[m] * nlive;
				for (int k = 0; k < nlive; k++)
					if (!known[ids[k]]) { known[ids[k]] = 1; fresh[nfresh++] = ids[k]; }      /* each once */
			}
			if (nfresh > 1) qsort(fresh, (size_t) nfresh, sizeof(int64_t), cmp_i64);   /* ascending */
			if (used + nfresh > cap) { ngroups = -2; goto done; }
			memcpy(points + used, fresh, (size_t) nfresh * sizeof(int64_t));
			used += nfresh;
		}
		for (int64_t i = begin; i < used; i++) known[points[i]] = 0;       /* next group starts clean */
		offsets[++ngroups] = used;
	}
done:
	free(start); free(known); free(todo); free(holders); free(fresh_members); free(fresh);
	return ngroups;
}
