// The integer side of one nested-sampling iteration behind ONE native call (include/mdns.h, Part 6):
// what the reference's MultiNestedSampler does between `prepare` and the replacement of the dead
// points (multi_nested_sampler.py:365-534) -- the passes over the data sets whose shelf is empty, the
// grouping of data sets that share live points, the choice of a constrainer for every group
// (cachedconstrainer.py:19-116), the constrained draws themselves (mdns_constrainer_draw, Part 5) and
// the shelves' queues of point ids.  Plain host C++: the likelihood side stays with the joint state
// behind the mdns_draw_backend table, the connected components of big selections with the device
// behind mdns_group_backend (small ones: a union-find here).
//
// What is restated, and from where (paths under the reference checkout):
//   multi_nested_sampler.py:365-491   the fill loop: superset passes, focussed passes, per-group draws
//   multi_nested_sampler.py:268-355   generate_subsets_graph: connected components of the bipartite
//                                     graph {data sets} -- {live points}, clusters in igraph's order
//                                     (by lowest data set), ids ascending
//   multi_nested_sampler.py:204-266   generate_subsets_nograph (the walk of csrc/host_groups.c)
//   multi_nested_sampler.py:474-489   a drawn point joins the pile and the shelves it beats
//   multi_nested_sampler.py:494-534   the end of an iteration
//   multi_nested_sampler.py:148-173   cut_down
//   cachedconstrainer.py:19-116       CachedConstrainer.get, generate_individual_constrainer
// Every decision is an integer one; the random numbers are consumed inside mdns_constrainer_draw
// only, in the order of the group loop, which is the reference's.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <time.h>
#include <unordered_map>
#include <vector>

#include "mdns.h"

extern "C" {
typedef struct mdns_walk mdns_walk;
mdns_walk *mdns_host_walk_create(void);
void mdns_host_walk_destroy(mdns_walk *w);
int mdns_host_walk_reset(mdns_walk *w, const int32_t *lpT, int nlive, int ndata, int64_t npoints);
int mdns_host_walk_groups(mdns_walk *w, const uint8_t *mask, int32_t *group_of, int32_t *points, int64_t cap,
                          int64_t *offsets, int64_t *ndistinct, int sorted_distinct, int64_t sort_below);
}

namespace {

char g_core_error[512] = "";

void core_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_core_error, sizeof g_core_error, fmt, ap);
	va_end(ap);
}

inline long long now_ns()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (long long) ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

// a group of data sets that share live points: members = positions in the running list, ascending;
// ids = the live points of the group (ascending, or in slot / discovery order where the reference's are)
struct Group {
	std::vector<int32_t> members;
	std::vector<int32_t> ids;
};

struct Cons {
	mdns_constrainer *h = nullptr;
	long long last_used = 0;
	~Cons() { if (h) mdns_constrainer_destroy(h, nullptr); }
};
typedef std::shared_ptr<Cons> ConsRef;

// index of a counter in mdns_core_stats' output
enum { C_NDRAWS, C_NDRAW_CALLS, C_NEVALS, C_NPOINTS, C_ITER, C_NRUN, C_NSUPER, C_PASSES, C_GROUPINGS, C_GROUPINGS_HOST,
       C_GROUPINGS_DEVICE, C_GROUPINGS_WALK, C_CONSTRAINERS, C_NS_DRAW, C_NS_GROUP, C_NS_FILL, C_SIMILAR,
       // groupings by selection size (< 2, 8, 32, 128, 512, 2048, 8192, more): calls and nanoseconds
       C_SIZE_CALLS, C_SIZE_NS = C_SIZE_CALLS + 8,
       // focussed groupings kept up to date (host_sampler_inc.h): analyses from scratch, updates, updates that split a component
       C_INC_BUILDS = C_SIZE_NS + 8, C_INC_UPDATES, C_INC_SPLITS, C_COUNTERS };

}  // namespace

struct mdns_core {
	int nlive = 0, ndata_total = 0, ndim = 0, nsuperset = 10;
	bool use_graph = true;
	int metric = MDNS_METRIC_TRUNCATEDSCALING, rebuild_every = 1000, metric_rebuild_every = 20, force_shrink = 1;
	const mdns_draw_backend *be = nullptr;
	const mdns_prior *prior = nullptr;
	const mdns_numpy_ops *np = nullptr;
	void *mt = nullptr;
	mdns_group_backend gb = {};
	bool have_gb = false;
	long long host_edges_max = 32768;          // selections with at most this many (data set, id) pairs: union-find here
	long long *shelf_mirror = nullptr;         // int64[ndata_total], by ORIGINAL index: +1 per point shelved (the joint state's mirror)
	long long *cons_totals = nullptr;          // the constrainers' shared counters
	// the running data sets
	int nrun = 0;
	std::vector<int32_t> running;              // [nrun] original index by position, ascending
	std::vector<int32_t> lp;                   // [nrun][nlive] live-point ids by position
	std::vector<std::vector<int32_t>> shelf;   // by position: ids waiting, FIFO from `head`
	std::vector<int32_t> head;
	// every point ever accepted
	std::vector<double> pile_u, pile_x;
	long long npile = 0;
	std::vector<int32_t> superpoints;          // ascending
	long long global_iter = 0;
	long long stat[C_COUNTERS] = {0};
	// constrainers
	ConsRef superset;
	std::unordered_map<int, ConsRef> individual;
	std::unordered_map<std::string, ConsRef> generations[4];
	long long cache_iter = -1;
	bool have_last = false;
	std::vector<int32_t> last_mask, last_points;
	std::string last_key;
	std::vector<int32_t> last_stamp;           // by id: == last_token when the id is among last_points
	int32_t last_token = 0;
	bool last_table = false;
	// grouping scratch
	std::vector<int32_t> seen, owner;          // by id (union-find over a selection)
	int32_t seen_token = 0;
	std::vector<uint64_t> bitmap;              // by id, clean between calls
	std::vector<int32_t> parent, sel, label_buf, idlabel_buf, distinct_buf, grp_of;
	std::vector<int32_t> rows_buf;
	mdns_walk *walk = nullptr;
	bool walk_stale = true;
	std::vector<uint8_t> mask8;
	std::vector<int32_t> group_of, walk_points;
	std::vector<int64_t> walk_offsets;
	std::vector<int32_t> empty_list;
	// the components of the focussed passes of the current iteration, kept up to date (host_sampler_inc.h)
	void *inc = nullptr;
	void (*inc_free)(void *) = nullptr;
	long long inc_edges_max = 150000;          // first focussed selections beyond this many pairs are not analysed here (the device is quicker)
	bool check_groups = false;                 // MDNS_CORE_CHECK_GROUPS=1: every incremental result against a fresh one
	// draw outputs
	std::vector<double> u_out, x_out;
	std::vector<unsigned long long> bits;

	~mdns_core() { if (walk) mdns_host_walk_destroy(walk); if (inc && inc_free) inc_free(inc); }
	int shelf_n(int pos) const { return (int) shelf[pos].size() - head[pos]; }
};

namespace {

ConsRef fresh_constrainer(mdns_core *c)
{
	ConsRef r = std::make_shared<Cons>();
	r->h = mdns_constrainer_create(c->ndim, c->metric, c->rebuild_every, c->metric_rebuild_every, c->force_shrink);
	if (!r->h) { core_error("mdns_constrainer_create: %s", mdns_host_last_error()); return ConsRef(); }
	if (c->cons_totals) mdns_constrainer_share_stats(r->h, c->cons_totals);
	c->stat[C_CONSTRAINERS]++;
	return r;
}

void grow_id_arrays(mdns_core *c)
{
	const size_t need = (size_t) c->npile + 1;
	if (c->seen.size() < need) {
		const size_t n = need + need / 2 + 1024;
		c->seen.resize(n, 0);
		c->owner.resize(n, 0);
		c->last_stamp.resize(n, 0);
		c->bitmap.resize(n / 64 + 2, 0ull);
	}
}

// ---------------------------------------------------------------------------------------
// grouping
// ---------------------------------------------------------------------------------------
// Connected components over the selected data sets `sel` (positions, ascending), here: union-find
// over the ids (a data set is joined with the first selected data set seen holding the same id),
// roots = lowest member, so that components come out in igraph's cluster order.  Fills
// label_buf[k] = selection index of the lowest member of k's component, distinct_buf (ascending)
// and idlabel_buf (label of every distinct id, same order).
void host_components(mdns_core *c, const std::vector<int32_t> &sel, int *ncomp)
{
	const int M = (int) sel.size(), nlive = c->nlive;
	grow_id_arrays(c);
	if (c->seen_token == 0x7fffffff) { std::fill(c->seen.begin(), c->seen.end(), 0); c->seen_token = 0; }
	const int32_t token = ++c->seen_token;
	c->parent.resize(M);
	for (int k = 0; k < M; k++) c->parent[k] = k;
	int32_t *parent = c->parent.data();
	int32_t lo = 0x7fffffff, hi = -1;
	uint64_t *bm = c->bitmap.data();
	for (int k = 0; k < M; k++) {
		const int32_t *ids = &c->lp[(size_t) sel[k] * nlive];
		for (int s = 0; s < nlive; s++) {
			const int32_t q = ids[s];
			if (c->seen[q] != token) {
				c->seen[q] = token;
				c->owner[q] = k;
				bm[q >> 6] |= 1ull << (q & 63);
				if (q < lo) lo = q;
				if (q > hi) hi = q;
			} else {
				int a = k, b = c->owner[q];
				while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; }
				while (parent[b] != b) { parent[b] = parent[parent[b]]; b = parent[b]; }
				if (a < b) parent[b] = a; else if (b < a) parent[a] = b;
			}
		}
	}
	c->label_buf.resize(M);
	int n = 0;
	for (int k = 0; k < M; k++) {
		int a = k;
		while (parent[a] != a) a = parent[a];
		c->label_buf[k] = a;
		if (a == k) n++;
	}
	*ncomp = n;
	c->distinct_buf.clear();
	c->idlabel_buf.clear();
	if (hi >= 0)
		for (int64_t w = lo >> 6; w <= hi >> 6; w++) {
			uint64_t bits = bm[w];
			bm[w] = 0;
			while (bits) {
				const int32_t q = (int32_t) (w * 64 + __builtin_ctzll(bits));
				bits &= bits - 1;
				c->distinct_buf.push_back(q);
				c->idlabel_buf.push_back(c->label_buf[c->owner[q]]);
			}
		}
}

// generate_subsets_graph (multi_nested_sampler.py:268-355) over the selection `sel` (positions)
bool groups_graph(mdns_core *c, const std::vector<int32_t> &sel, std::vector<Group> &out)
{
	out.clear();
	const int M = (int) sel.size(), nlive = c->nlive;
	if (M == 1) {
		// one data set: its own live points, in slot order (:206-213)
		Group g;
		g.members = sel;
		g.ids.assign(&c->lp[(size_t) sel[0] * nlive], &c->lp[(size_t) sel[0] * nlive] + nlive);
		out.push_back(std::move(g));
		return true;
	}
	int ncomp = 0;
	const bool on_device = c->have_gb && (long long) M * nlive > c->host_edges_max;
	const int32_t *labels = nullptr, *idlabels = nullptr;
	if (on_device) {
		c->stat[C_GROUPINGS_DEVICE]++;
		const int32_t *rows = nullptr;
		if (M != c->ndata_total) {
			c->rows_buf.resize(M);
			for (int k = 0; k < M; k++) c->rows_buf[k] = c->running[sel[k]];
			rows = c->rows_buf.data();
		}
		long long cap = (long long) M * nlive;
		if (cap > c->npile) cap = c->npile;
		c->distinct_buf.resize((size_t) cap + 1);
		long long nd = 0;
		if (c->gb.components(c->gb.user, rows, M, c->npile, &ncomp, &nd, c->distinct_buf.data(), cap, nullptr) != 0) {
			core_error("the device grouping failed (mdns_groups_components)");
			return false;
		}
		c->distinct_buf.resize((size_t) nd);
	} else {
		c->stat[C_GROUPINGS_HOST]++;
		host_components(c, sel, &ncomp);
		labels = c->label_buf.data();
		idlabels = c->idlabel_buf.data();
	}
	const long long nd = (long long) c->distinct_buf.size();
	// The reference's two shortcuts (:283-297) return ONE group without looking at the graph: fewer
	// than 2 nlive distinct ids, or superpoints known (which does not imply one component: see
	// MultiNestedSampler.generate_subsets_graph in multi_nested_sampler.py of this package)
	if (ncomp == 1 || nd < 2LL * c->nlive || !c->superpoints.empty()) {
		{
			static const char *log_path = getenv("MDNS_CORE_GROUP_LOG");
			if (log_path) {
				static FILE *f = fopen((std::string(log_path) + ".single").c_str(), "w");
				if (f) fprintf(f, "%lld %d %lld %d %zu\n", c->global_iter, M, nd, ncomp, c->superpoints.size());
			}
		}
		Group g;
		g.members = sel;
		g.ids = c->distinct_buf;
		out.push_back(std::move(g));
		return true;
	}
	if (on_device) {
		c->label_buf.resize(M);
		c->idlabel_buf.resize((size_t) nd);
		if (c->gb.id_labels(c->gb.user, c->label_buf.data(), c->idlabel_buf.data(), nd) != 0) {
			core_error("the device grouping failed (mdns_groups_id_labels)");
			return false;
		}
		labels = c->label_buf.data();
		idlabels = c->idlabel_buf.data();
	}
	// Components in ascending order of their label (the lowest data set of each: igraph's cluster
	// order), members and ids ascending inside.  A label's first occurrence among the members IS its
	// lowest member, so numbering labels by first occurrence gives that order.
	// (labels are selection indices here, original data-set indices from the device: both below this)
	const size_t nlabel = (size_t) (on_device ? c->ndata_total : M) + 1;
	if (c->grp_of.size() < nlabel) c->grp_of.resize(nlabel);
	std::vector<int32_t> &table = c->grp_of;         // label -> group number, cleared for the labels that occur
	for (int k = 0; k < M; k++) {
		if (labels[k] < 0 || (size_t) labels[k] >= nlabel) { core_error("grouping: label %d out of range", labels[k]); return false; }
		table[labels[k]] = -1;
	}
	for (int k = 0; k < M; k++) {
		const int32_t L = labels[k];
		if (table[L] < 0) {
			table[L] = (int32_t) out.size();
			out.emplace_back();
		}
		out[table[L]].members.push_back(sel[k]);
	}
	{
		// MDNS_CORE_GROUP_LOG=<file>: M, distinct ids, components, the three largest (analysis only)
		static const char *log_path = getenv("MDNS_CORE_GROUP_LOG");
		if (log_path) {
			static FILE *f = fopen(log_path, "w");
			if (f) {
				size_t top[3] = {0, 0, 0};
				for (const Group &g : out) {
					size_t v = g.members.size();
					for (int t = 0; t < 3; t++) if (v > top[t]) { const size_t w = top[t]; top[t] = v; v = w; }
				}
				fprintf(f, "%lld %d %lld %d %zu %zu %zu\n", c->global_iter, M, nd, ncomp, top[0], top[1], top[2]);
			}
		}
	}
	for (long long t = 0; t < nd; t++) {
		const int32_t L = idlabels[t];
		if (L < 0 || (size_t) L >= nlabel || table[L] < 0 || table[L] >= (int32_t) out.size()) {
			core_error("grouping: live point %d carries label %d of no selected data set", c->distinct_buf[t], L);
			return false;
		}
		out[table[L]].ids.push_back(c->distinct_buf[t]);
	}
	return true;
}

}  // namespace

#include <algorithm>
#include "host_sampler_inc.h"

namespace {

Incremental &inc_state(mdns_core *c)
{
	if (!c->inc) {
		c->inc = new Incremental();
		c->inc_free = [](void *p) { delete (Incremental *) p; };
	}
	return *(Incremental *) c->inc;
}

// generate_subsets_graph for the focussed passes: the first one of an iteration analyses its
// selection, the later ones update that analysis (host_sampler_inc.h)
bool groups_focussed(mdns_core *c, const std::vector<int32_t> &sel, std::vector<Group> &out)
{
	Incremental &I = inc_state(c);
	if (sel.size() == 1) return groups_graph(c, sel, out);          // (its own live points, in slot order)
	const long long t_start = now_ns();
	if (!I.valid) {
		if ((long long) sel.size() * c->nlive > c->inc_edges_max) return groups_graph(c, sel, out);
		inc_build(c, I, sel);
		c->stat[C_INC_BUILDS]++;
	} else {
		const long long before = I.splits;
		inc_update(c, I, sel);
		c->stat[C_INC_UPDATES]++;
		c->stat[C_INC_SPLITS] += I.splits - before;
	}
	static long long t_upd = 0, t_emit = 0;
	const long long tu = now_ns();
	t_upd += tu - t_start;
	inc_emit(c, I, sel, out);
	t_emit += now_ns() - tu;
	{
		static const bool trace = getenv("MDNS_CORE_INC_TRACE") != nullptr;
		if (trace)
			fprintf(stderr, "inc: M %zu comps %zu orphans %lld chain %lld scanned %lld splits %lld update %lld us emit %lld us\n", sel.size(), I.comps.size(),
			        I.orphans, I.chain_steps, I.scanned_ids, I.splits, t_upd / 1000, t_emit / 1000);
	}
	if (c->check_groups) {
		std::vector<Group> fresh;
		if (!groups_graph(c, sel, fresh)) return false;
		bool same = fresh.size() == out.size();
		for (size_t g = 0; same && g < fresh.size(); g++) same = fresh[g].members == out[g].members && fresh[g].ids == out[g].ids;
		if (!same) {
			core_error("incremental grouping differs from a fresh one (iteration %lld, %zu data sets: %zu groups against %zu)",
			           c->global_iter, sel.size(), out.size(), fresh.size());
			return false;
		}
	}
	return true;
}

// generate_subsets_nograph (multi_nested_sampler.py:204-266) through the incremental walk
bool groups_walk(mdns_core *c, const std::vector<int32_t> &sel, std::vector<Group> &out)
{
	out.clear();
	const int M = (int) sel.size(), nlive = c->nlive, nrun = c->nrun;
	if (M == 1) {
		Group g;
		g.members = sel;
		g.ids.assign(&c->lp[(size_t) sel[0] * nlive], &c->lp[(size_t) sel[0] * nlive] + nlive);
		out.push_back(std::move(g));
		return true;
	}
	c->stat[C_GROUPINGS_WALK]++;
	if (!c->walk) {
		c->walk = mdns_host_walk_create();
		if (!c->walk) { core_error("mdns_host_walk_create"); return false; }
		c->walk_stale = true;
	}
	if (c->walk_stale) {
		if (mdns_host_walk_reset(c->walk, c->lp.data(), nlive, nrun, c->npile) != 0) { core_error("mdns_host_walk_reset"); return false; }
		c->walk_stale = false;
	}
	c->mask8.assign(nrun, 0);
	for (int k = 0; k < M; k++) c->mask8[sel[k]] = 1;
	c->group_of.resize(nrun);
	c->walk_offsets.resize((size_t) nrun + 1);
	const int64_t cap = (int64_t) M * nlive + nlive;
	c->walk_points.resize((size_t) cap);
	int64_t ndistinct = 0;
	const int n = mdns_host_walk_groups(c->walk, c->mask8.data(), c->group_of.data(), c->walk_points.data(), cap,
	                                    c->walk_offsets.data(), &ndistinct, c->superpoints.empty() ? 0 : 1, 2LL * nlive);
	if (n < 0) { core_error("mdns_host_walk_groups (%d)", n); return false; }
	if (n == 0 || n == 1) {
		Group g;
		g.members = sel;
		const int64_t len = n == 0 ? ndistinct : c->walk_offsets[1];
		g.ids.assign(c->walk_points.begin(), c->walk_points.begin() + len);
		out.push_back(std::move(g));
		return true;
	}
	out.resize(n);
	for (int k = 0; k < M; k++) {
		const int g = c->group_of[sel[k]];
		if (g < 0 || g >= n) { core_error("walk: data set without a group"); return false; }
		out[g].members.push_back(sel[k]);
	}
	for (int g = 0; g < n; g++)
		out[g].ids.assign(c->walk_points.begin() + c->walk_offsets[g], c->walk_points.begin() + c->walk_offsets[g + 1]);
	return true;
}

// ---------------------------------------------------------------------------------------
// which constrainer draws for a group (cachedconstrainer.py)
// ---------------------------------------------------------------------------------------
// generate_individual_constrainer (cachedconstrainer.py:92-109): one per data set, its region dropped
// when it was last used more than five iterations ago
ConsRef individual_constrainer(mdns_core *c, int original, long long it)
{
	auto found = c->individual.find(original);
	ConsRef r;
	if (found == c->individual.end()) {
		r = fresh_constrainer(c);
		if (!r) return r;
		r->last_used = it;
		c->individual.emplace(original, r);
	} else r = found->second;
	if (it > r->last_used + 5) mdns_constrainer_forget_region(r->h);
	r->last_used = it;
	return r;
}

// CachedConstrainer.get(mask, realmask, points, it) (cachedconstrainer.py:35-90); mask = ORIGINAL
// indices of the group's data sets
ConsRef cached_constrainer(mdns_core *c, const int32_t *mask, int nmask, const std::vector<int32_t> &points, long long it)
{
	while (c->cache_iter < it) {
		c->generations[3] = std::move(c->generations[2]);
		c->generations[2] = std::move(c->generations[1]);
		c->generations[1] = std::move(c->generations[0]);
		c->generations[0].clear();
		c->have_last = false;
		c->last_table = false;
		c->cache_iter++;
	}
	if (c->have_last) {
		// "similar to the call just before": slightly fewer data sets and live points, all of them
		// among the last call's (:54-62); the data-set mask object is the sampler's own every time
		const double nm = (double) nmask, lm = (double) c->last_mask.size();
		const double np = (double) points.size(), lpn = (double) c->last_points.size();
		if (nmask < (int) c->last_mask.size() && nm > 0.80 * lm && points.size() <= c->last_points.size() && np > 0.90 * lpn) {
			if (!c->last_table) {
				grow_id_arrays(c);
				if (c->last_token == 0x7fffffff) { std::fill(c->last_stamp.begin(), c->last_stamp.end(), 0); c->last_token = 0; }
				c->last_token++;
				for (int32_t q : c->last_points) c->last_stamp[q] = c->last_token;
				c->last_table = true;
			}
			bool all = true;
			for (int32_t q : points) if (c->last_stamp[q] != c->last_token) { all = false; break; }
			if (all) {
				c->stat[C_SIMILAR]++;
				return c->generations[0][c->last_key];
			}
		}
	}
	std::string key((const char *) mask, (size_t) nmask * sizeof(int32_t));
	c->have_last = true;
	c->last_mask.assign(mask, mask + nmask);
	c->last_points = points;
	c->last_table = false;
	c->last_key = key;
	auto &current = c->generations[0];
	auto found = current.find(key);
	if (found != current.end()) return found->second;
	for (int g = 1; g < 4; g++) {
		auto older = c->generations[g].find(key);
		if (older != c->generations[g].end()) {
			current.emplace(key, older->second);
			return older->second;
		}
	}
	ConsRef r = fresh_constrainer(c);
	if (r) current.emplace(key, r);
	return r;
}

// ---------------------------------------------------------------------------------------
// one constrained draw for a group and what follows from it (multi_nested_sampler.py:449-489)
// ---------------------------------------------------------------------------------------
bool draw_for_group(mdns_core *c, const Group &g, bool rebuilding)
{
	const int njoints = (int) g.members.size();
	const int32_t *rows = nullptr;
	if (!(njoints == c->nrun && c->nrun == c->ndata_total)) {
		c->rows_buf.resize(njoints);
		for (int k = 0; k < njoints; k++) c->rows_buf[k] = c->running[g.members[k]];
		rows = c->rows_buf.data();
	}
	ConsRef cons;
	if (njoints == 1) cons = individual_constrainer(c, c->running[g.members[0]], c->global_iter);
	else if (rebuilding) {
		// (the cache is keyed by the ORIGINAL indices of the group's data sets)
		std::vector<int32_t> key_rows;
		const int32_t *mask = rows;
		if (!mask) {
			key_rows.resize(njoints);
			for (int k = 0; k < njoints; k++) key_rows[k] = c->running[g.members[k]];
			mask = key_rows.data();
		}
		cons = cached_constrainer(c, mask, njoints, g.ids, c->global_iter);
	} else cons = c->superset;
	if (!cons) return false;
	long long ntries = 0;
	const size_t words = (size_t) (njoints + 63) / 64;
	if (c->bits.size() < words + 1) c->bits.resize(words + 1);
	const long long t0 = now_ns();
	const int rc = mdns_constrainer_draw(cons->h, c->be, c->prior, c->np, c->mt, c->pile_u.data(), g.ids.data(), 4, (int) g.ids.size(),
	                                     rows, njoints, c->u_out.data(), c->x_out.data(), &ntries, c->bits.data());
	c->stat[C_NS_DRAW] += now_ns() - t0;
	if (rc != 0) { core_error("mdns_constrainer_draw failed: %s", mdns_host_last_error()); return false; }
	// the point joins the pile ...
	const int ndim = c->ndim;
	const long long ppi = c->npile;
	if ((size_t) (ppi + 1) * ndim > c->pile_u.size()) {
		const size_t n = ((size_t) ppi + 1 + (size_t) ppi / 2 + 4096) * ndim;
		c->pile_u.resize(n);
		c->pile_x.resize(n);
	}
	memcpy(&c->pile_u[(size_t) ppi * ndim], c->u_out.data(), (size_t) ndim * sizeof(double));
	memcpy(&c->pile_x[(size_t) ppi * ndim], c->x_out.data(), (size_t) ndim * sizeof(double));
	c->npile = ppi + 1;
	// ... and the shelves of the data sets whose threshold it beats (:482-485)
	int nfilled = 0;
	for (size_t w = 0; w < words; w++) {
		unsigned long long bits = c->bits[w];
		if (w == words - 1 && (njoints & 63)) bits &= (1ull << (njoints & 63)) - 1ull;
		while (bits) {
			const int k = (int) (w * 64 + __builtin_ctzll(bits));
			bits &= bits - 1;
			const int pos = g.members[k];
			c->shelf[pos].push_back((int32_t) ppi);
			if (c->shelf_mirror) c->shelf_mirror[c->running[pos]] += 1;
			nfilled++;
		}
	}
	if (nfilled == c->nrun) c->superpoints.push_back((int32_t) ppi);       // (:486-488; ppi exceeds every id so far)
	c->stat[C_NDRAWS] += ntries;
	c->stat[C_NDRAW_CALLS] += 1;
	c->stat[C_NEVALS] += ntries * njoints;
	return true;
}

}  // namespace

extern "C" const char *mdns_core_last_error(void) { return g_core_error; }

extern "C" mdns_core *mdns_core_create(int nlive, int ndata, int ndim, int nsuperset_draws, int use_graph,
                                       int metriclearner, int rebuild_every, int metric_rebuild_every, int force_shrink,
                                       const mdns_draw_backend *be, const mdns_prior *prior, const mdns_numpy_ops *np,
                                       void *mt19937_state, const mdns_group_backend *gb, long long *shelf_mirror,
                                       long long *constrainer_totals)
{
	if (nlive <= 0 || ndata <= 0 || ndim <= 0 || ndim > MDNS_MAX_DIM || !be || !prior || !mt19937_state) {
		core_error("mdns_core_create: bad arguments (nlive=%d ndata=%d ndim=%d)", nlive, ndata, ndim);
		return nullptr;
	}
	mdns_core *c = new mdns_core();
	c->nlive = nlive; c->ndata_total = ndata; c->ndim = ndim; c->nsuperset = nsuperset_draws;
	c->use_graph = use_graph != 0;
	c->metric = metriclearner; c->rebuild_every = rebuild_every; c->metric_rebuild_every = metric_rebuild_every;
	c->force_shrink = force_shrink;
	c->be = be; c->prior = prior; c->np = np; c->mt = mt19937_state;
	if (gb && gb->components && gb->id_labels) { c->gb = *gb; c->have_gb = true; }
	c->shelf_mirror = shelf_mirror;
	c->cons_totals = constrainer_totals;
	c->nrun = ndata;
	c->running.resize(ndata);
	for (int d = 0; d < ndata; d++) c->running[d] = d;
	c->lp.resize((size_t) ndata * nlive);
	c->shelf.resize(ndata);
	c->head.assign(ndata, 0);
	c->u_out.resize(ndim);
	c->x_out.resize(ndim);
	c->superset = fresh_constrainer(c);
	if (!c->superset) { delete c; return nullptr; }
	return c;
}

extern "C" void mdns_core_destroy(mdns_core *c) { delete c; }

extern "C" void mdns_core_set_host_edges(mdns_core *c, long long edges) { if (c) c->host_edges_max = edges; }
extern "C" void mdns_core_set_incremental(mdns_core *c, long long edges_max, int check)
{
	if (!c) return;
	c->inc_edges_max = edges_max;
	c->check_groups = check != 0;
}

// the nlive prior draws every data set starts from (multi_nested_sampler.py:88-103)
extern "C" int mdns_core_set_initial(mdns_core *c, const double *u, const double *x)
{
	if (!c || !u || !x) return 1;
	const int nlive = c->nlive, ndim = c->ndim;
	c->pile_u.assign(u, u + (size_t) nlive * ndim);
	c->pile_x.assign(x, x + (size_t) nlive * ndim);
	c->pile_u.resize((size_t) (nlive + 4096) * ndim);
	c->pile_x.resize((size_t) (nlive + 4096) * ndim);
	c->npile = nlive;
	for (int d = 0; d < c->nrun; d++)
		for (int p = 0; p < nlive; p++) c->lp[(size_t) d * nlive + p] = p;
	c->superpoints.resize(nlive);
	for (int p = 0; p < nlive; p++) c->superpoints[p] = p;
	c->stat[C_NDRAWS] = nlive;
	c->stat[C_NEVALS] = (long long) nlive * c->nrun;
	return 0;
}

// shelf entries that no longer beat their data set's threshold leave, in order
// (multi_nested_sampler.py:137-138): keep uint8[nrun][width], entry e of the data set at position r
// stays when e < width and keep[r][e]
extern "C" int mdns_core_purge(mdns_core *c, const unsigned char *keep, int width)
{
	if (!c || (!keep && width > 0)) return 1;
	for (int pos = 0; pos < c->nrun; pos++) {
		std::vector<int32_t> &q = c->shelf[pos];
		const int h = c->head[pos], n = (int) q.size() - h;
		if (n == 0) continue;
		int w = 0;
		for (int e = 0; e < n; e++)
			if (e < width && keep[(size_t) pos * width + e]) q[w++] = q[h + e];
		q.resize(w);
		c->head[pos] = 0;
	}
	return 0;
}

// multi_nested_sampler.py:365-491: draws until no running data set has an empty shelf
extern "C" int mdns_core_fill(mdns_core *c)
{
	if (!c) return 1;
	const long long t_fill = now_ns();
	std::vector<Group> superset_groups, tmp;
	bool have_superset = false;
	if (c->inc) ((Incremental *) c->inc)->valid = false;       // the id matrix changed since the last iteration
	std::vector<int32_t> everybody;
	long long passes = 0;
	bool first_list = true;
	for (;;) {
		passes++;
		// the data sets whose shelf is empty (only ever fewer within an iteration)
		if (first_list) {
			c->empty_list.clear();
			for (int pos = 0; pos < c->nrun; pos++) if (c->shelf_n(pos) == 0) c->empty_list.push_back(pos);
			first_list = false;
		} else {
			size_t w = 0;
			for (int32_t pos : c->empty_list) if (c->shelf_n(pos) == 0) c->empty_list[w++] = pos;
			c->empty_list.resize(w);
		}
		if (c->empty_list.empty()) break;
		c->stat[C_PASSES]++;
		const bool focussed = passes > c->nsuperset;
		const std::vector<Group> *groups;
		if (!focussed && have_superset) groups = &superset_groups;
		else {
			const std::vector<int32_t> *sel = &c->empty_list;
			if (!focussed) {
				if (everybody.empty()) { everybody.resize(c->nrun); for (int pos = 0; pos < c->nrun; pos++) everybody[pos] = pos; }
				sel = &everybody;
			}
			std::vector<Group> &dst = focussed ? tmp : superset_groups;
			{
				// MDNS_CORE_DUMP=<iteration>:<file>: the id matrix and the selections of that iteration's passes
				// (analysis of the grouping off line: tools/groups_replay.py)
				static const char *dump = getenv("MDNS_CORE_DUMP");
				if (dump && atoll(dump) == c->global_iter && strchr(dump, ':')) {
					static FILE *f = nullptr;
					if (!f) {
						f = fopen(strchr(dump, ':') + 1, "wb");
						if (f) {
							const long long head[3] = {c->nrun, c->nlive, c->npile};
							fwrite(head, sizeof head, 1, f);
							fwrite(c->lp.data(), sizeof(int32_t), c->lp.size(), f);
						}
					}
					if (f) {
						const long long rec[2] = {(long long) sel->size(), focussed ? 1 : 0};
						fwrite(rec, sizeof rec, 1, f);
						fwrite(sel->data(), sizeof(int32_t), sel->size(), f);
						fflush(f);
					}
				}
			}
			const long long t0 = now_ns();
			c->stat[C_GROUPINGS]++;
			const bool ok = !c->use_graph ? groups_walk(c, *sel, dst)
			                : (focussed && c->inc_edges_max > 0 ? groups_focussed(c, *sel, dst) : groups_graph(c, *sel, dst));
			const long long dt = now_ns() - t0;
			c->stat[C_NS_GROUP] += dt;
			{
				static const size_t edges[8] = {2, 8, 32, 128, 512, 2048, 8192, (size_t) -1};
				int bucket = 0;
				while (sel->size() >= edges[bucket]) bucket++;
				c->stat[C_SIZE_CALLS + bucket]++;
				c->stat[C_SIZE_NS + bucket] += dt;
			}
			if (!ok) return 1;
			if (!focussed) have_superset = true;
			groups = &dst;
		}
		if (groups->empty()) { core_error("grouping returned no group"); return 1; }
		const bool rebuilding = focussed || groups->size() > 1;
		for (const Group &g : *groups) {
			if (groups->size() > 1 && !focussed) {
				bool all_waiting = true;
				for (int32_t pos : g.members) if (c->shelf_n(pos) == 0) { all_waiting = false; break; }
				if (all_waiting) continue;                       // this group needs nothing (:434-436)
			}
			if (!draw_for_group(c, g, rebuilding)) return 1;
		}
	}
	c->stat[C_NS_FILL] += now_ns() - t_fill;
	return 0;
}

// multi_nested_sampler.py:494-534: every running data set gives up the live point in slot
// argmin[position] and takes the head of its shelf.  dead_u / dead_x f64[nrun][ndim] receive the
// coordinates of the points given up, dead_ids / new_ids int32[nrun] the ids (any may be NULL).
extern "C" int mdns_core_advance(mdns_core *c, const int32_t *argmin, double *dead_u, double *dead_x,
                                 int32_t *dead_ids, int32_t *new_ids)
{
	if (!c || !argmin) return 1;
	const int nlive = c->nlive, ndim = c->ndim, nrun = c->nrun;
	for (int pos = 0; pos < nrun; pos++) {
		if (c->shelf_n(pos) <= 0) { core_error("mdns_core_advance: data set %d has an empty shelf", c->running[pos]); return 1; }
		if (argmin[pos] < 0 || argmin[pos] >= nlive) { core_error("mdns_core_advance: slot %d", argmin[pos]); return 1; }
	}
	c->global_iter++;
	grow_id_arrays(c);
	if (c->seen_token == 0x7fffffff) { std::fill(c->seen.begin(), c->seen.end(), 0); c->seen_token = 0; }
	const int32_t token = ++c->seen_token;           // marks the ids that die somewhere
	std::vector<int32_t> &newp = c->label_buf;
	newp.resize(nrun);
	for (int pos = 0; pos < nrun; pos++) {
		int32_t *ids = &c->lp[(size_t) pos * nlive];
		const int32_t dead = ids[argmin[pos]];
		if (dead_u) memcpy(dead_u + (size_t) pos * ndim, &c->pile_u[(size_t) dead * ndim], (size_t) ndim * sizeof(double));
		if (dead_x) memcpy(dead_x + (size_t) pos * ndim, &c->pile_x[(size_t) dead * ndim], (size_t) ndim * sizeof(double));
		if (dead_ids) dead_ids[pos] = dead;
		c->seen[dead] = token;
		std::vector<int32_t> &q = c->shelf[pos];
		const int32_t fresh = q[c->head[pos]++];
		if (c->head[pos] == (int) q.size()) { q.clear(); c->head[pos] = 0; }
		ids[argmin[pos]] = fresh;
		newp[pos] = fresh;
		if (new_ids) new_ids[pos] = fresh;
	}
	if (!c->superpoints.empty()) {
		size_t w = 0;
		for (int32_t q : c->superpoints) if (c->seen[q] != token) c->superpoints[w++] = q;
		c->superpoints.resize(w);
	}
	c->walk_stale = true;
	if (c->have_gb && c->gb.replace) {
		if (c->gb.replace(c->gb.user, c->running.data(), argmin, newp.data(), nrun) != 0) {
			core_error("the device grouping failed (mdns_groups_replace)");
			return 1;
		}
	}
	return 0;
}

// multi_nested_sampler.py:148-173: surviving uint8[nrun] by position
extern "C" int mdns_core_cut_down(mdns_core *c, const unsigned char *surviving)
{
	if (!c || !surviving) return 1;
	const int nlive = c->nlive;
	int w = 0;
	for (int pos = 0; pos < c->nrun; pos++) {
		if (!surviving[pos]) continue;
		if (w != pos) {
			c->running[w] = c->running[pos];
			memmove(&c->lp[(size_t) w * nlive], &c->lp[(size_t) pos * nlive], (size_t) nlive * sizeof(int32_t));
			c->shelf[w] = std::move(c->shelf[pos]);
			c->head[w] = c->head[pos];
		}
		w++;
	}
	c->nrun = w;
	c->running.resize(w);
	c->lp.resize((size_t) w * nlive);
	c->shelf.resize(w);
	c->head.resize(w);
	c->walk_stale = true;
	return 0;
}

// ---- what the Python object shows of the state ----
extern "C" long long mdns_core_npoints(const mdns_core *c) { return c ? c->npile : -1; }
extern "C" int mdns_core_nrunning(const mdns_core *c) { return c ? c->nrun : -1; }
extern "C" const double *mdns_core_pile_u(const mdns_core *c) { return c ? c->pile_u.data() : nullptr; }
extern "C" const double *mdns_core_pile_x(const mdns_core *c) { return c ? c->pile_x.data() : nullptr; }

// live_pointsp int32[nlive][nrun] (the reference's orientation, multi_nested_sampler.py:108)
extern "C" int mdns_core_get_ids(const mdns_core *c, int32_t *out)
{
	if (!c || !out) return 1;
	for (int pos = 0; pos < c->nrun; pos++)
		for (int p = 0; p < c->nlive; p++) out[(size_t) p * c->nrun + pos] = c->lp[(size_t) pos * c->nlive + p];
	return 0;
}

// shelf sizes int32[nrun]; with ids != NULL also the waiting ids, queue after queue (cap entries at most;
// returns the number written, -1 when cap is too small)
extern "C" long long mdns_core_get_shelves(const mdns_core *c, int32_t *sizes, int32_t *ids, long long cap)
{
	if (!c || !sizes) return -1;
	long long n = 0;
	for (int pos = 0; pos < c->nrun; pos++) {
		const int k = c->shelf_n(pos);
		sizes[pos] = k;
		if (ids) {
			if (n + k > cap) return -1;
			memcpy(ids + n, c->shelf[pos].data() + c->head[pos], (size_t) k * sizeof(int32_t));
		}
		n += k;
	}
	return n;
}

extern "C" int mdns_core_get_superpoints(const mdns_core *c, int32_t *out, int cap)
{
	if (!c) return -1;
	const int n = (int) c->superpoints.size();
	if (out) memcpy(out, c->superpoints.data(), (size_t) (n < cap ? n : cap) * sizeof(int32_t));
	return n;
}

extern "C" void mdns_core_stats(const mdns_core *c, long long *out)
{
	if (!c || !out) return;
	memcpy(out, c->stat, sizeof c->stat);
	out[C_NPOINTS] = c->npile;
	out[C_ITER] = c->global_iter;
	out[C_NRUN] = c->nrun;
	out[C_NSUPER] = (long long) c->superpoints.size();
}

// ---- test entry points (tests/test_core.py): the grouping on a planted id matrix ----
// ids int32[nlive][nrunning] (the reference's orientation); every id below npoints
extern "C" int mdns_core_debug_set_ids(mdns_core *c, const int32_t *ids, long long npoints, int nsuperpoints)
{
	if (!c || !ids || npoints <= 0) return 1;
	for (int pos = 0; pos < c->nrun; pos++)
		for (int p = 0; p < c->nlive; p++) {
			const int32_t q = ids[(size_t) p * c->nrun + pos];
			if (q < 0 || q >= npoints) { core_error("mdns_core_debug_set_ids: id %d", q); return 1; }
			c->lp[(size_t) pos * c->nlive + p] = q;
		}
	c->npile = npoints;
	c->pile_u.resize((size_t) npoints * c->ndim);
	c->pile_x.resize((size_t) npoints * c->ndim);
	c->superpoints.assign((size_t) (nsuperpoints > 0 ? nsuperpoints : 0), 0);
	c->walk_stale = true;
	if (c->inc) ((Incremental *) c->inc)->valid = false;
	return 0;
}

// the groups of the selection sel int32[M] (positions, ascending): group_of int32[M] = group number of
// every selected data set (groups in output order), ids = the groups' id lists one after the other,
// offsets int64[ngroups + 1].  focussed: through the incremental path (a sequence of shrinking
// selections; `restart` begins a new sequence).  Returns the number of groups, negative on failure.
extern "C" int mdns_core_debug_groups(mdns_core *c, const int32_t *sel, int M, int focussed, int restart,
                                      int32_t *group_of, int32_t *ids, long long cap, long long *offsets)
{
	if (!c || !sel || M <= 0 || !group_of || !ids || !offsets) return -1;
	std::vector<int32_t> s(sel, sel + M);
	std::vector<Group> out;
	if (restart && c->inc) ((Incremental *) c->inc)->valid = false;
	const bool ok = !c->use_graph ? groups_walk(c, s, out) : (focussed ? groups_focussed(c, s, out) : groups_graph(c, s, out));
	if (!ok) return -1;
	std::vector<int32_t> at((size_t) c->nrun, -1);
	for (int k = 0; k < M; k++) at[s[k]] = k;
	long long n = 0;
	offsets[0] = 0;
	for (size_t g = 0; g < out.size(); g++) {
		for (int32_t pos : out[g].members) group_of[at[pos]] = (int32_t) g;
		if (n + (long long) out[g].ids.size() > cap) return -2;
		memcpy(ids + n, out[g].ids.data(), out[g].ids.size() * sizeof(int32_t));
		n += (long long) out[g].ids.size();
		offsets[g + 1] = n;
	}
	return (int) out.size();
}
