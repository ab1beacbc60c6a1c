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

/* ------------------------------------------------------------------------------------------
 * The same walk, incremental over the passes of ONE nested-sampling iteration.
 *
 * Within an iteration the id matrix is fixed and the selections asked for -- the data sets whose
 * shelf is still empty -- only shrink, a few data sets per pass, for a hundred passes.  The
 * stateless walk above rebuilds the id -> holders index of the whole selection every time (most
 * of its cost); this one builds it for the first selection it sees (the BASE), keeps the holder
 * counts of the CURRENT selection up to date by subtracting the data sets that left, and walks
 * the base's holder lists skipping the departed.  A selection that is not a subset of the base,
 * or has shrunk below half of it, becomes the new base.  Labels are int32.  Results are those of
 * mdns_host_group_walk (tests/test_sampler_units.py compares them on shrinking selections).
 * ------------------------------------------------------------------------------------------ */
typedef struct mdns_walk {
	const int32_t *lpT;             /* [ndata][nlive] labels, borrowed until the next reset */
	int nlive, ndata;
	int64_t npoints;
	/* per id */
	int32_t *cnt;                   /* holders in the CURRENT selection */
	int64_t *first;                 /* start of the id's holder list (ids with >= 2 holders in the base) */
	int32_t *len;                   /* its length in the base */
	int32_t *seen;                  /* call tag: listed / pending in this call */
	int64_t cap_points;
	/* per data set */
	uint8_t *cur;                   /* in the current selection */
	uint8_t *todo;
	int64_t cap_data;
	/* base */
	int32_t *holders;  int64_t cap_holders;
	int32_t *touched;  int64_t ntouched, cap_touched;      /* ids with a holder in the base */
	int64_t base_nsel, cur_nsel, ndistinct;
	int have_base;
	int32_t tag;
	/* scratch */
	int32_t *sel;  int32_t *fresh;  int32_t *sort_tmp;  int64_t cap_scratch;
} mdns_walk;

mdns_walk *mdns_host_walk_create(void) { return (mdns_walk *) calloc(1, sizeof(mdns_walk)); }

void mdns_host_walk_destroy(mdns_walk *w)
{
	if (!w) return;
	free(w->cnt); free(w->first); free(w->len); free(w->seen); free(w->cur); free(w->todo);
	free(w->holders); free(w->touched); free(w->sel); free(w->fresh); free(w->sort_tmp);
	free(w);
}

static int walk_fit(void **p, int64_t *cap, int64_t need, size_t elem, int zero)
{
	if (need <= *cap) return 1;
	int64_t n = need + need / 2 + 64;
	void *q = zero ? calloc((size_t) n, elem) : malloc((size_t) n * elem);
	if (!q) return 0;
	free(*p);
	*p = q; *cap = n;
	return 1;
}

/* A new iteration: the id matrix changed.  lpT stays borrowed until the next reset. */
int mdns_host_walk_reset(mdns_walk *w, const int32_t *lpT, int nlive, int ndata, int64_t npoints)
{
	if (!w) return -1;
	/* counts and tags of the ids the old base touched go back to zero */
	for (int64_t t = 0; t < w->ntouched; t++) { w->cnt[w->touched[t]] = 0; w->seen[w->touched[t]] = 0; }
	w->ntouched = 0; w->have_base = 0; w->tag = 0;
	w->lpT = lpT; w->nlive = nlive; w->ndata = ndata; w->npoints = npoints;
	if (npoints > w->cap_points) {
		int64_t c1 = w->cap_points, c2 = w->cap_points, c3 = w->cap_points, c4 = w->cap_points;
		if (!walk_fit((void **) &w->cnt, &c1, npoints, sizeof(int32_t), 1) ||
		    !walk_fit((void **) &w->first, &c2, npoints, sizeof(int64_t), 0) ||
		    !walk_fit((void **) &w->len, &c3, npoints, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->seen, &c4, npoints, sizeof(int32_t), 1)) return -1;
		w->cap_points = c1 < c2 ? c1 : c2;
		if (c3 < w->cap_points) w->cap_points = c3;
		if (c4 < w->cap_points) w->cap_points = c4;
	}
	if (ndata > w->cap_data) {
		int64_t c1 = w->cap_data, c2 = w->cap_data;
		if (!walk_fit((void **) &w->cur, &c1, ndata, 1, 1) || !walk_fit((void **) &w->todo, &c2, ndata, 1, 1)) return -1;
		w->cap_data = c1 < c2 ? c1 : c2;
	}
	memset(w->cur, 0, (size_t) ndata);
	return 0;
}

static void sort_i32(int32_t *v, int64_t n, int32_t *tmp)
{
	if (n < 2) return;
	if (n < 64) {                                   /* insertion sort */
		for (int64_t i = 1; i < n; i++) {
			const int32_t x = v[i];
			int64_t j = i;
			while (j > 0 && v[j - 1] > x) { v[j] = v[j - 1]; j--; }
			v[j] = x;
		}
		return;
	}
	int32_t top = 0;
	for (int64_t i = 0; i < n; i++) if (v[i] > top) top = v[i];
	int32_t *src = v, *dst = tmp;
	for (int shift = 0; shift < 32 && (top >> shift) != 0; shift += 8) {
		int64_t count[257];
		memset(count, 0, sizeof count);
		for (int64_t i = 0; i < n; i++) count[((src[i] >> shift) & 255) + 1]++;
		for (int b = 0; b < 256; b++) count[b + 1] += count[b];
		for (int64_t i = 0; i < n; i++) dst[count[(src[i] >> shift) & 255]++] = src[i];
		int32_t *t = src; src = dst; dst = t;
	}
	if (src != v) memcpy(v, src, (size_t) n * sizeof(int32_t));
}

/* make `mask` the base: holder index over exactly these data sets */
static int walk_rebase(mdns_walk *w, const uint8_t *mask, int64_t nsel)
{
	const int nlive = w->nlive;
	for (int64_t t = 0; t < w->ntouched; t++) { w->cnt[w->touched[t]] = 0; w->seen[w->touched[t]] = 0; }
	w->ntouched = 0; w->tag = 0;
	if (!walk_fit((void **) &w->holders, &w->cap_holders, nsel * nlive, sizeof(int32_t), 0)) return 0;
	{
		const int64_t bound = nsel * nlive < w->npoints ? nsel * nlive : w->npoints;
		if (!walk_fit((void **) &w->touched, &w->cap_touched, bound, sizeof(int32_t), 0)) return 0;
	}
	if (nsel * nlive > w->cap_scratch) {
		int64_t c1 = w->cap_scratch, c2 = w->cap_scratch, c3 = w->cap_scratch;
		if (!walk_fit((void **) &w->sel, &c1, nsel * nlive, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->fresh, &c2, nsel * nlive, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->sort_tmp, &c3, nsel * nlive, sizeof(int32_t), 0)) return 0;
		w->cap_scratch = c1 < c2 ? c1 : c2;
		if (c3 < w->cap_scratch) w->cap_scratch = c3;
	}
	int64_t nt = 0;
	for (int d = 0; d < w->ndata; d++) {
		w->cur[d] = mask[d] != 0;
		if (!w->cur[d]) continue;
		const int32_t *ids = w->lpT + (size_t) d * nlive;
		for (int k = 0; k < nlive; k++) if (w->cnt[ids[k]]++ == 0) w->touched[nt++] = ids[k];
	}
	w->ntouched = nt; w->ndistinct = nt;
	int64_t at = 0;
	for (int64_t t = 0; t < nt; t++) {
		const int32_t p = w->touched[t];
		w->len[p] = w->cnt[p];
		if (w->cnt[p] >= 2) { w->first[p] = at; at += w->cnt[p]; }
	}
	for (int d = 0; d < w->ndata; d++) {                       /* ascending data sets: lists come out sorted */
		if (!w->cur[d]) continue;
		const int32_t *ids = w->lpT + (size_t) d * nlive;
		for (int k = 0; k < nlive; k++) if (w->len[ids[k]] >= 2) w->holders[w->first[ids[k]]++] = d;
	}
	for (int64_t t = 0; t < nt; t++) { const int32_t p = w->touched[t]; if (w->len[p] >= 2) w->first[p] -= w->len[p]; }
	w->base_nsel = w->cur_nsel = nsel;
	w->have_base = 1;
	return 1;
}

/* mask      uint8[ndata]          the data sets to group
 * group_of  int32[ndata]   out    group index of every selected data set, -1 for the others
 * points    int32[cap]     out    the groups' point lists (labels), one after the other
 * offsets   int64[ndata+1] out    group g owns points[offsets[g] : offsets[g+1]]
 * ndistinct               out     number of distinct ids the selection holds
 * When `sorted_distinct` is set, or there are fewer than `sort_below` distinct ids, the walk is
 * not made at all (the caller needs ONE group with the ids ascending,
 * multi_nested_sampler.py:206-235): points[0 : ndistinct] gets them and the return value is 0.
 * Otherwise returns the number of groups; -1 out of memory, -2 `cap` too small. */
int mdns_host_walk_groups(mdns_walk *w, const uint8_t *mask, int32_t *group_of, int32_t *points, int64_t cap,
                          int64_t *offsets, int64_t *ndistinct, int sorted_distinct, int64_t sort_below)
{
	if (!w || !w->lpT) return -1;
	const int nlive = w->nlive, ndata = w->ndata;
	int64_t nsel = 0;
	int subset = w->have_base;
	for (int d = 0; d < ndata; d++) {
		const int m = mask[d] != 0;
		nsel += m;
		if (m && !w->cur[d]) subset = 0;
	}
	if (!subset || 2 * nsel < w->base_nsel) {
		if (!walk_rebase(w, mask, nsel)) return -1;
	} else if (nsel != w->cur_nsel) {
		/* the data sets that left take their ids' counts with them */
		for (int d = 0; d < ndata; d++) {
			if (!w->cur[d] || mask[d]) continue;
			w->cur[d] = 0;
			const int32_t *ids = w->lpT + (size_t) d * nlive;
			for (int k = 0; k < nlive; k++) if (--w->cnt[ids[k]] == 0) w->ndistinct--;
		}
		w->cur_nsel = nsel;
	}
	*ndistinct = w->ndistinct;
	if (w->tag > 0x3ffffff0) {                                  /* tags wrap: start over (never in practice) */
		for (int64_t t = 0; t < w->ntouched; t++) w->seen[w->touched[t]] = 0;
		w->tag = 0;
	}
	const int32_t listed = ++w->tag, pending = ++w->tag;
	if (sorted_distinct || w->ndistinct < sort_below) {
		if (w->ndistinct > cap) return -2;
		int64_t n = 0;
		for (int d = 0; d < ndata; d++) {
			if (!w->cur[d]) continue;
			const int32_t *ids = w->lpT + (size_t) d * nlive;
			for (int k = 0; k < nlive; k++) if (w->seen[ids[k]] != listed) { w->seen[ids[k]] = listed; points[n++] = ids[k]; }
		}
		sort_i32(points, n, w->sort_tmp);
		return 0;
	}
	for (int d = 0; d < ndata; d++) { w->todo[d] = w->cur[d]; group_of[d] = -1; }
	int64_t used = 0, left = nsel;
	int next_first = 0, ngroups = 0;
	offsets[0] = 0;
	int32_t *sel = w->sel, *fresh = w->fresh;
	while (left > 0) {
		while (!w->todo[next_first]) next_first++;
		const int lead = next_first;
		w->todo[lead] = 0; left--;
		group_of[lead] = ngroups;
		const int64_t begin = used;
		if (used + nlive > cap) return -2;
		for (int k = 0; k < nlive; k++) {
			const int32_t p = w->lpT[(size_t) lead * nlive + k];
			points[used++] = p;
			w->seen[p] = listed;
		}
		for (int64_t i = begin; i < used && left > 0; i++) {
			const int32_t p = points[i];
			if (w->cnt[p] < 2) continue;                        /* held by nobody else in the selection */
			int64_t nnew = 0;
			const int32_t *list = w->holders + w->first[p];
			for (int32_t h = 0; h < w->len[p]; h++) {
				const int d = list[h];
				if (w->todo[d]) { w->todo[d] = 0; left--; group_of[d] = ngroups; sel[nnew++] = d; }
			}
			if (!nnew) continue;
			int64_t nfresh = 0;
			int32_t lo = INT32_MAX, hi = -1;
			for (int64_t m = 0; m < nnew; m++) {
				const int32_t *ids = w->lpT + (size_t) sel[m] * nlive;
				for (int k = 0; k < nlive; k++) {
					const int32_t q = ids[k];
					if (w->seen[q] != listed && w->seen[q] != pending) {            /* each once */
						w->seen[q] = pending; fresh[nfresh++] = q;
						if (q < lo) lo = q;
						if (q > hi) hi = q;
					}
				}
			}
			/* ascending: a dense batch is read off the marks in id order, a sparse one is sorted */
			if (nfresh >= 192 && (int64_t) hi - lo < 24 * nfresh) {
				int64_t n = 0;
				for (int32_t q = lo; q <= hi; q++) if (w->seen[q] == pending) { w->seen[q] = listed; fresh[n++] = q; }
			} else {
				for (int64_t f = 0; f < nfresh; f++) w->seen[fresh[f]] = listed;
				sort_i32(fresh, nfresh, w->sort_tmp);
			}
			if (used + nfresh > cap) return -2;
			memcpy(points + used, fresh, (size_t) nfresh * sizeof(int32_t));
			used += nfresh;
		}
		offsets[++ngroups] = used;
	}
	return ngroups;
}
