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
	uint64_t *listed;               /* bit per id: in the point list of this call (clean between calls) */
	uint64_t *pending;              /* bit per id: found in the batch being assembled */
	int32_t *pos;                   /* where a listed id stands in the point list (valid while its bit is set) */
	int32_t *qbatch;                /* the batch an id was listed in, in the kept result */
	int64_t cap_points;
	/* per data set */
	uint8_t *cur;                   /* in the current selection */
	uint8_t *todo;
	uint8_t *isnew;                 /* joined in the batch being assembled (clean between batches) */
	int32_t *dbatch;                /* the batch a data set joined in, in the kept result */
	int32_t *dgroup;                /* its group there (-1: not selected) */
	int64_t cap_data;
	/* base */
	int32_t *holders;  int64_t cap_holders;
	int32_t *touched;  int64_t ntouched, cap_touched;      /* ids with a holder in the base, ASCENDING */
	int32_t *unlisted; int64_t cap_unlisted;               /* ids of the current selection not yet listed, ascending */
	/* the result of the last walk, kept to derive the next one when only a few data sets left */
	int32_t *res_points; int64_t res_used, cap_res;
	int64_t *res_offsets; int32_t *res_lead; int64_t cap_groups;
	int res_ngroups, have_result;
	int32_t *gone; int64_t cap_gone;
	int64_t base_nsel, cur_nsel, ndistinct;
	int have_base;
	/* scratch */
	int32_t *sel;  int32_t *fresh;  int32_t *sort_tmp;  int64_t cap_scratch;
} mdns_walk;

mdns_walk *mdns_host_walk_create(void) { return (mdns_walk *) calloc(1, sizeof(mdns_walk)); }

void mdns_host_walk_destroy(mdns_walk *w)
{
	if (!w) return;
	free(w->cnt); free(w->first); free(w->len); free(w->listed); free(w->pending); free(w->pos); free(w->cur);
	free(w->todo); free(w->isnew); free(w->unlisted); free(w->qbatch); free(w->dbatch); free(w->dgroup);
	free(w->res_points); free(w->res_offsets); free(w->res_lead); free(w->gone);
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
	/* counts of the ids the old base touched go back to zero */
	for (int64_t t = 0; t < w->ntouched; t++) w->cnt[w->touched[t]] = 0;
	w->ntouched = 0; w->have_base = 0; w->have_result = 0;
	w->lpT = lpT; w->nlive = nlive; w->ndata = ndata; w->npoints = npoints;
	if (npoints > w->cap_points) {
		/* (the bit maps are clean between calls, so fresh zeroed ones are as good as the old) */
		int64_t c1 = w->cap_points, c2 = w->cap_points, c3 = w->cap_points, c6 = w->cap_points, c7 = w->cap_points;
		int64_t c4 = (w->cap_points + 63) / 64, c5 = (w->cap_points + 63) / 64;
		if (!walk_fit((void **) &w->cnt, &c1, npoints, sizeof(int32_t), 1) ||
		    !walk_fit((void **) &w->first, &c2, npoints, sizeof(int64_t), 0) ||
		    !walk_fit((void **) &w->len, &c3, npoints, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->pos, &c6, npoints, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->qbatch, &c7, npoints, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->listed, &c4, (npoints + 63) / 64 + 1, sizeof(uint64_t), 1) ||
		    !walk_fit((void **) &w->pending, &c5, (npoints + 63) / 64 + 1, sizeof(uint64_t), 1)) return -1;
		w->cap_points = c1 < c2 ? c1 : c2;
		if (c3 < w->cap_points) w->cap_points = c3;
		if (c6 < w->cap_points) w->cap_points = c6;
		if (c7 < w->cap_points) w->cap_points = c7;
		if ((c4 - 1) * 64 < w->cap_points) w->cap_points = (c4 - 1) * 64;
		if ((c5 - 1) * 64 < w->cap_points) w->cap_points = (c5 - 1) * 64;
	}
	if (ndata > w->cap_data) {
		int64_t c1 = w->cap_data, c2 = w->cap_data, c3 = w->cap_data, c4 = w->cap_data, c5 = w->cap_data;
		int64_t c6 = w->cap_groups, c7 = w->cap_groups, c8 = w->cap_gone;
		if (!walk_fit((void **) &w->cur, &c1, ndata, 1, 1) || !walk_fit((void **) &w->todo, &c2, ndata, 1, 1) ||
		    !walk_fit((void **) &w->isnew, &c3, ndata, 1, 1) ||
		    !walk_fit((void **) &w->dbatch, &c4, ndata, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->dgroup, &c5, ndata, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->res_offsets, &c6, ndata + 1, sizeof(int64_t), 0) ||
		    !walk_fit((void **) &w->res_lead, &c7, ndata + 1, sizeof(int32_t), 0) ||
		    !walk_fit((void **) &w->gone, &c8, ndata, sizeof(int32_t), 0)) return -1;
		w->cap_data = c1 < c2 ? c1 : c2;
		if (c3 < w->cap_data) w->cap_data = c3;
		if (c4 < w->cap_data) w->cap_data = c4;
		if (c5 < w->cap_data) w->cap_data = c5;
		w->cap_groups = c6 < c7 ? c6 : c7;
		w->cap_gone = c8;
	}
	memset(w->cur, 0, (size_t) ndata);
	return 0;
}

/* make `mask` the base: holder index over exactly these data sets */
static int walk_rebase(mdns_walk *w, const uint8_t *mask, int64_t nsel)
{
	const int nlive = w->nlive;
	for (int64_t t = 0; t < w->ntouched; t++) w->cnt[w->touched[t]] = 0;
	w->ntouched = 0;
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
	if (!walk_fit((void **) &w->unlisted, &w->cap_unlisted, nt, sizeof(int32_t), 0)) return 0;
	/* the touched ids ascending: through the (clean) bit map */
	if (nt > 0) {
		int32_t lo = INT32_MAX, hi = -1;
		for (int64_t t = 0; t < nt; t++) {
			const int32_t q = w->touched[t];
			w->listed[q >> 6] |= 1ull << (q & 63);
			if (q < lo) lo = q;
			if (q > hi) hi = q;
		}
		int64_t n = 0;
		for (int64_t wd = lo >> 6; wd <= hi >> 6; wd++) {
			uint64_t bits = w->listed[wd];
			w->listed[wd] = 0;
			while (bits) { w->touched[n++] = (int32_t) (wd * 64 + __builtin_ctzll(bits)); bits &= bits - 1; }
		}
	}
	int64_t at = 0;
	for (int64_t t = 0; t < nt; t++) {
		const int32_t p = w->touched[t];
		w->len[p] = w->cnt[p];
		if (w->cnt[p] >= 2) { w->first[p] = at; at += w->cnt[p]; }
	}
	for (int d = 0; d < w->ndata; d++) {                       /* ascending data sets: lists come out sorted */
		if (!w->cur[d]) continue;
		const int32_t *ids = w->lpT + (size_t) d * nlive;
		for (int k = 0; k < nlive; k++) {
			const int32_t q = ids[k];
			if (w->len[q] >= 2) w->holders[w->first[q]++] = d;
			else w->first[q] = d;                               /* a single holder: kept instead of a list */
		}
	}
	for (int64_t t = 0; t < nt; t++) { const int32_t p = w->touched[t]; if (w->len[p] >= 2) w->first[p] -= w->len[p]; }
	w->base_nsel = w->cur_nsel = nsel;
	w->have_base = 1;
	w->have_result = 0;
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
	int64_t nsel = 0, ngone = 0;
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
			w->gone[ngone++] = d;
			const int32_t *ids = w->lpT + (size_t) d * nlive;
			for (int k = 0; k < nlive; k++) if (--w->cnt[ids[k]] == 0) w->ndistinct--;
		}
		w->cur_nsel = nsel;
	}
	*ndistinct = w->ndistinct;
	uint64_t *listed = w->listed, *pending = w->pending;
#define BIT(map, q) ((map)[(q) >> 6] >> ((q) & 63) & 1)
#define SET(map, q) ((map)[(q) >> 6] |= 1ull << ((q) & 63))
	if (sorted_distinct || w->ndistinct < sort_below) {
		if (w->ndistinct > cap) return -2;
		w->have_result = 0;
		/* the distinct ids, ascending: marked in a bit map, read off word by word */
		int32_t lo = INT32_MAX, hi = -1;
		for (int d = 0; d < ndata; d++) {
			if (!w->cur[d]) continue;
			const int32_t *ids = w->lpT + (size_t) d * nlive;
			for (int k = 0; k < nlive; k++) {
				const int32_t q = ids[k];
				SET(listed, q);
				if (q < lo) lo = q;
				if (q > hi) hi = q;
			}
		}
		int64_t n = 0;
		if (hi >= 0)
			for (int64_t wd = lo >> 6; wd <= hi >> 6; wd++) {
				uint64_t bits = listed[wd];
				listed[wd] = 0;
				while (bits) { points[n++] = (int32_t) (wd * 64 + __builtin_ctzll(bits)); bits &= bits - 1; }
			}
		return 0;
	}
	/* Only a few data sets left since the last walk of this base?  Then the new lists are the
	 * old ones without the ids nobody holds any more -- provided no group lost its first data
	 * set and every id a departed data set shares with the remaining ones was listed in a batch
	 * that still has one of its holders: then every remaining data set is brought in by the same
	 * point as before and every id is found in the same batch (see the header comment). */
	if (w->have_result && ngone * 8 < w->cur_nsel + ngone) {
		int ok = 1;
		for (int64_t gi = 0; gi < ngone && ok; gi++) {
			const int x = w->gone[gi];
			if (w->res_lead[w->dgroup[x]] == x) { ok = 0; break; }
			const int32_t *ids = w->lpT + (size_t) x * nlive;
			for (int k = 0; k < nlive && ok; k++) {
				const int32_t q = ids[k];
				if (w->cnt[q] == 0) continue;                       /* leaves the list with x */
				const int32_t want = w->qbatch[q];
				const int32_t *list = w->holders + w->first[q];      /* cnt > 0 and x held it: at least two base holders */
				const int32_t n = w->len[q];
				int32_t h = 0;
				for (; h < n; h++) { const int y = list[h]; if (w->cur[y] && w->dbatch[y] == want) break; }
				ok = h < n;
			}
		}
		if (ok) {
			for (int64_t gi = 0; gi < ngone; gi++) w->dgroup[w->gone[gi]] = -1;
			int64_t used = 0;
			for (int g = 0; g < w->res_ngroups; g++) {
				const int64_t from = w->res_offsets[g], to = w->res_offsets[g + 1];
				w->res_offsets[g] = used;
				for (int64_t i = from; i < to; i++) {
					const int32_t q = w->res_points[i];
					if (w->cnt[q] > 0) w->res_points[used++] = q;
				}
			}
			w->res_offsets[w->res_ngroups] = used;
			if (used > cap) return -2;
			memcpy(points, w->res_points, (size_t) used * sizeof(int32_t));
			memcpy(offsets, w->res_offsets, (size_t) (w->res_ngroups + 1) * sizeof(int64_t));
			memcpy(group_of, w->dgroup, (size_t) ndata * sizeof(int32_t));
			w->res_used = used;
			return w->res_ngroups;
		}
	}
	w->have_result = 0;
	for (int d = 0; d < ndata; d++) { w->todo[d] = w->cur[d]; group_of[d] = -1; }
	int64_t used = 0, left = nsel;
	int next_first = 0, ngroups = 0;
	offsets[0] = 0;
	int32_t *sel = w->sel, *pos = w->pos;
	uint8_t *todo = w->todo, *isnew = w->isnew;
	/* ids of the selection not yet in a list, ascending (those that got listed meanwhile are
	 * dropped when the array is next walked) */
	int64_t nunl = 0;
	for (int64_t t = 0; t < w->ntouched; t++) if (w->cnt[w->touched[t]] > 0) w->unlisted[nunl++] = w->touched[t];
	enum { STRAGGLERS = 16 };
	int32_t rest[STRAGGLERS];
	int32_t batch = 0;                                          /* counts leads and bringing points */
	while (left > 0) {
		while (!todo[next_first]) next_first++;
		const int lead = next_first;
		todo[lead] = 0; left--;
		group_of[lead] = ngroups;
		w->res_lead[ngroups] = lead;
		w->dbatch[lead] = ++batch;
		const int64_t begin = used;
		if (used + nlive > cap) { ngroups = -2; break; }
		for (int k = 0; k < nlive; k++) {
			const int32_t p = w->lpT[(size_t) lead * nlive + k];
			pos[p] = (int32_t) used;
			w->qbatch[p] = batch;
			points[used++] = p;
			SET(listed, p);
		}
		int64_t i = begin;
		int nrest = -1;                                         /* >= 0: the data sets still to place are in rest[] */
		while (i < used && left > 0) {
			int64_t nnew = 0;
			if (left <= STRAGGLERS) {
				/* A handful of data sets are left: instead of scanning the holder lists of all the
				 * points still to come, look where each of them first meets the list. */
				if (nrest < 0) {
					nrest = 0;
					for (int d = next_first; d < ndata && nrest < left; d++) if (todo[d]) rest[nrest++] = d;
				}
				int64_t best = INT64_MAX;
				int64_t meets[STRAGGLERS];
				for (int r = 0; r < nrest; r++) {
					const int32_t *ids = w->lpT + (size_t) rest[r] * nlive;
					int64_t e = INT64_MAX;
					for (int k = 0; k < nlive; k++) {
						const int32_t q = ids[k];
						if (BIT(listed, q) && pos[q] < e) e = pos[q];
					}
					meets[r] = e;
					if (e < best) best = e;
				}
				if (best == INT64_MAX) break;                   /* nobody left shares a point with this group */
				i = best + 1;                                   /* (a point before i would have brought them in already) */
				int keep = 0;
				for (int r = 0; r < nrest; r++) {
					if (meets[r] == best) { const int d = rest[r]; todo[d] = 0; left--; group_of[d] = ngroups; sel[nnew++] = d; }
					else rest[keep++] = rest[r];
				}
				nrest = keep;
			} else {
				const int32_t p = points[i++];
				if (w->cnt[p] < 2) continue;                    /* held by nobody else in the selection */
				const int32_t *list = w->holders + w->first[p];
				for (int32_t h = 0; h < w->len[p]; h++) {
					const int d = list[h];
					if (todo[d]) { todo[d] = 0; left--; group_of[d] = ngroups; sel[nnew++] = d; }
				}
				if (!nnew) continue;
			}
			++batch;
			for (int64_t m = 0; m < nnew; m++) w->dbatch[sel[m]] = batch;
			/* The not yet listed ids of the newcomers, ascending and each once: marked in the
			 * `pending` bit map, then read off word by word.  Marking goes through the newcomers'
			 * rows (nnew * nlive ids), or -- when a batch brings in so many data sets that this
			 * is the larger number -- through the unlisted ids, each asking its holders. */
			int32_t lo = INT32_MAX, hi = -1;
			int by_rows = 1;
			if (nnew * nlive > 4 * nunl && nunl > 0) {
				for (int64_t m = 0; m < nnew; m++) isnew[sel[m]] = 1;
				int64_t budget = nnew * nlive, keep = 0;
				by_rows = 0;
				for (int64_t t = 0; t < nunl; t++) {
					const int32_t q = w->unlisted[t];
					if (BIT(listed, q) || w->cnt[q] == 0) continue;   /* listed meanwhile */
					int found = 0;
					if (w->len[q] < 2) found = isnew[w->first[q]];
					else {
						const int32_t *list = w->holders + w->first[q];
						const int32_t n = w->len[q];
						int32_t h = 0;
						for (; h < n && !isnew[list[h]]; h++) { }
						found = h < n;
						budget -= h;
					}
					if (found) { SET(pending, q); if (q < lo) lo = q; hi = q; }
					else w->unlisted[keep++] = q;
					if (budget < 0) {                                  /* not worth it after all */
						for (int64_t u = t + 1; u < nunl; u++) w->unlisted[keep++] = w->unlisted[u];
						by_rows = 1;
						break;
					}
				}
				nunl = keep;
				for (int64_t m = 0; m < nnew; m++) isnew[sel[m]] = 0;
				if (by_rows && hi >= 0) {
					/* the marks made so far are valid (a marked id is fresh either way) but the
					 * ids dropped from `unlisted` with them must stay findable: they are, through
					 * the rows pass below, which marks every fresh id again */
				}
			}
			if (by_rows) {
				for (int64_t m = 0; m < nnew; m++) {
					const int32_t *ids = w->lpT + (size_t) sel[m] * nlive;
					for (int k = 0; k < nlive; k++) {
						const int32_t q = ids[k];
						const uint64_t fresh = ~listed[q >> 6] >> (q & 63) & 1;
						pending[q >> 6] |= fresh << (q & 63);
						lo = fresh && q < lo ? q : lo;
						hi = fresh && q > hi ? q : hi;
					}
				}
			}
			if (hi < 0) continue;
			int overflow = 0;
			for (int64_t wd = lo >> 6; wd <= hi >> 6; wd++) {
				uint64_t bits = pending[wd];
				if (!bits) continue;
				pending[wd] = 0;
				if (used + __builtin_popcountll(bits) > cap) { overflow = 1; continue; }
				listed[wd] |= bits;
				while (bits) {
					const int32_t q = (int32_t) (wd * 64 + __builtin_ctzll(bits));
					pos[q] = (int32_t) used;
					w->qbatch[q] = batch;
					points[used++] = q;
					bits &= bits - 1;
				}
			}
			if (overflow) { ngroups = -2; break; }
		}
		if (ngroups < 0) break;
		offsets[++ngroups] = used;
	}
	/* the bit maps go back clean: every listed id is in the point list */
	for (int64_t i = 0; i < used; i++) listed[points[i] >> 6] = 0;
	if (ngroups > 0 && walk_fit((void **) &w->res_points, &w->cap_res, used, sizeof(int32_t), 0)) {
		memcpy(w->res_points, points, (size_t) used * sizeof(int32_t));
		memcpy(w->res_offsets, offsets, (size_t) (ngroups + 1) * sizeof(int64_t));
		memcpy(w->dgroup, group_of, (size_t) ndata * sizeof(int32_t));
		w->res_used = used; w->res_ngroups = ngroups; w->have_result = 1;
	}
	if (ngroups < 0) {
		/* an aborted call may have left marks outside the list: clear what the selection holds */
		for (int d = 0; d < ndata; d++) {
			if (!w->cur[d]) continue;
			const int32_t *ids = w->lpT + (size_t) d * nlive;
			for (int k = 0; k < nlive; k++) { listed[ids[k] >> 6] = 0; pending[ids[k] >> 6] = 0; }
		}
	}
#undef BIT
#undef SET
	return ngroups;
}
