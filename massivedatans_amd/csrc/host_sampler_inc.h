// Connected components of the FOCUSSED passes of one iteration, kept up to date instead of recomputed
// (included by host_sampler.cpp).
//
// Within an iteration the live-point matrix is fixed and the focussed selections only shrink: a pass
// selects the data sets whose shelf is still empty (multi_nested_sampler.py:373-388), and every draw
// of the pass fills a few shelves.  In the middle of a C2 run that is ~90 passes per iteration over
// ~1500 data sets that stay ONE component while a handful leave per pass (profiles/r04_groups_log.txt)
// -- and each pass paid a full components computation: 105 us on the device, 3 ns per (data set, id)
// pair on the host.
//
// Here the first focussed selection S1 of an iteration is analysed once -- union-find over its
// (data set, id) pairs, which also leaves: per id the chain of its holders, per id the number of
// holders still selected, and a rooted SPANNING FOREST of the data-set graph whose every edge is a real
// one (two data sets sharing a live point).  A later pass removes the data sets that left:
//   * their ids lose a holder each (ids without holders leave the component's id list),
//   * the forest falls into pieces -- the data sets whose path to the root is intact, and the subtrees
//     that hung below a data set that left; found by chasing parent pointers: O(remaining data sets),
//   * a piece is hung back into the forest through a REPLACEMENT edge: a live point one of its data sets
//     shares with a remaining data set outside the piece (found by walking holder chains; the first id
//     tried usually gives one); pieces are scanned smallest first, merged as they connect, and a set of
//     pieces that exhausts its data sets' ids without an edge out is a component of its own.
// The result -- groups in ascending order of their lowest data set, members and ids ascending, the
// reference's one-group shortcuts applied on top (multi_nested_sampler.py:283-297) -- is what a fresh
// computation gives (MDNS_CORE_CHECK_GROUPS=1 compares every pass; tests/test_core.py stresses it
// against the union-find on random graphs).
namespace {

struct IncComp {
	std::vector<int32_t> members;   // selection indices k, ascending
	std::vector<int32_t> ids;       // live-point ids, ascending
};

struct Incremental {
	bool valid = false;
	int M1 = 0;
	std::vector<int32_t> sel;             // S1: position of selection index k
	std::vector<int32_t> k_of_pos;        // by position: k, or -1
	std::vector<uint8_t> alive;           // by k
	std::vector<int32_t> parent;          // by k: spanning forest (k of the parent, -1: root)
	std::vector<int32_t> uf;              // by k: union-find of the build
	std::vector<int32_t> head, cnt, first_k;   // by compact id: holder chain, holders still selected, first holder
	std::vector<int32_t> next;            // by edge (k * nlive + slot): next holder of the same id
	std::vector<int32_t> cid_of, cid_stamp;    // by id
	int32_t cid_token = 0;
	std::vector<IncComp> comps;
	// scratch
	std::vector<int32_t> mark, top, top_stamp, path, piece_of;   // by k
	std::vector<int32_t> moved;                                   // by compact id
	int32_t pass_token = 0;
	long long builds = 0, updates = 0, splits = 0, scanned_ids = 0;
};

inline int inc_find(std::vector<int32_t> &uf, int a)
{
	while (uf[a] != a) { uf[a] = uf[uf[a]]; a = uf[a]; }
	return a;
}

// S1: everything from scratch.  Returns false when the selection is too large to be worth it.
bool inc_build(mdns_core *c, Incremental &I, const std::vector<int32_t> &sel)
{
	const int M = (int) sel.size(), nlive = c->nlive;
	I.valid = false;
	I.M1 = M;
	I.sel = sel;
	I.k_of_pos.assign((size_t) c->nrun, -1);
	for (int k = 0; k < M; k++) I.k_of_pos[sel[k]] = k;
	I.alive.assign(M, 1);
	I.parent.assign(M, -1);
	I.uf.resize(M);
	for (int k = 0; k < M; k++) I.uf[k] = k;
	I.next.resize((size_t) M * nlive);
	I.head.clear(); I.cnt.clear(); I.first_k.clear();
	const size_t nid = (size_t) c->npile + 1;
	if (I.cid_of.size() < nid) { I.cid_of.resize(nid + nid / 2 + 1024, 0); I.cid_stamp.resize(I.cid_of.size(), 0); }
	if (I.cid_token == 0x7fffffff) { std::fill(I.cid_stamp.begin(), I.cid_stamp.end(), 0); I.cid_token = 0; }
	const int32_t token = ++I.cid_token;
	grow_id_arrays(c);
	uint64_t *bm = c->bitmap.data();
	int32_t lo = 0x7fffffff, hi = -1;
	std::vector<int32_t> tree;                // pairs (k, j): real edges that joined two sets
	for (int k = 0; k < M; k++) {
		const int32_t *ids = &c->lp[(size_t) sel[k] * nlive];
		for (int s = 0; s < nlive; s++) {
			const int32_t q = ids[s];
			int32_t cid;
			if (I.cid_stamp[q] != token) {
				I.cid_stamp[q] = token;
				cid = (int32_t) I.head.size();
				I.cid_of[q] = cid;
				I.head.push_back(-1);
				I.cnt.push_back(1);
				I.first_k.push_back(k);
				bm[q >> 6] |= 1ull << (q & 63);
				if (q < lo) lo = q;
				if (q > hi) hi = q;
			} else {
				cid = I.cid_of[q];
				I.cnt[cid]++;
				const int a = inc_find(I.uf, k), b = inc_find(I.uf, I.first_k[cid]);
				if (a != b) {
					tree.push_back(k);
					tree.push_back(I.first_k[cid]);
					if (a < b) I.uf[b] = a; else I.uf[a] = b;
				}
			}
			const int32_t e = k * nlive + s;
			I.next[e] = I.head[cid];
			I.head[cid] = e;
		}
	}
	// components: roots are the lowest members, so ascending k meets every root first
	I.comps.clear();
	std::vector<int32_t> &comp_of = I.piece_of;       // (scratch: by k)
	comp_of.assign(M, -1);
	for (int k = 0; k < M; k++) {
		const int r = inc_find(I.uf, k);
		if (comp_of[r] < 0) { comp_of[r] = (int32_t) I.comps.size(); I.comps.emplace_back(); }
		comp_of[k] = comp_of[r];
		I.comps[comp_of[k]].members.push_back(k);
	}
	// ids per component, ascending (the bit map is read off and left clean)
	if (hi >= 0)
		for (int64_t w = lo >> 6; w <= hi >> 6; w++) {
			uint64_t bits = bm[w];
			bm[w] = 0;
			while (bits) {
				const int32_t q = (int32_t) (w * 64 + __builtin_ctzll(bits));
				bits &= bits - 1;
				I.comps[comp_of[I.first_k[I.cid_of[q]]]].ids.push_back(q);
			}
		}
	// the tree edges as a rooted forest: breadth first from every component's lowest member
	{
		const size_t ne = tree.size() / 2;
		std::vector<int32_t> start((size_t) M + 1, 0), adj(2 * ne);
		for (size_t t = 0; t < 2 * ne; t++) start[tree[t] + 1]++;
		for (int k = 0; k < M; k++) start[k + 1] += start[k];
		std::vector<int32_t> fill(start.begin(), start.end() - 1);
		for (size_t t = 0; t < ne; t++) {
			const int32_t a = tree[2 * t], b = tree[2 * t + 1];
			adj[fill[a]++] = b;
			adj[fill[b]++] = a;
		}
		std::vector<int32_t> queue;
		queue.reserve(M);
		std::vector<uint8_t> seen(M, 0);
		for (const IncComp &comp : I.comps) {
			const int32_t root = comp.members[0];
			seen[root] = 1;
			I.parent[root] = -1;
			queue.clear();
			queue.push_back(root);
			for (size_t at = 0; at < queue.size(); at++) {
				const int32_t x = queue[at];
				for (int32_t t = start[x]; t < start[x + 1]; t++) {
					const int32_t y = adj[t];
					if (!seen[y]) { seen[y] = 1; I.parent[y] = x; queue.push_back(y); }
				}
			}
		}
	}
	I.mark.assign(M, 0);
	I.top.assign(M, 0);
	I.top_stamp.assign(M, 0);
	I.moved.assign(I.head.size(), 0);
	I.pass_token = 0;
	I.valid = true;
	I.builds++;
	return true;
}

// the tree that holds x re-rooted at x (parent pointers reversed along the path to its root)
inline void inc_reroot(std::vector<int32_t> &parent, int32_t x)
{
	int32_t prev = -1;
	while (x >= 0) {
		const int32_t up = parent[x];
		parent[x] = prev;
		prev = x;
		x = up;
	}
}

// One component lost members: `mem` are the ones still alive.  Appends the components they form now to
// `out` (members only; ids are settled by the caller).
void inc_resolve(mdns_core *c, Incremental &I, const std::vector<int32_t> &mem, std::vector<std::vector<int32_t>> &out)
{
	const int nlive = c->nlive;
	const int32_t pass = I.pass_token;
	std::vector<int32_t> roots;               // piece = the alive data sets with the same top
	for (int32_t m : mem) {
		int32_t x = m, t;
		I.path.clear();
		for (;;) {
			if (I.top_stamp[x] == pass) { t = I.top[x]; break; }
			const int32_t p = I.parent[x];
			if (p < 0 || !I.alive[p]) {
				I.parent[x] = -1;
				t = x;
				I.top[x] = x; I.top_stamp[x] = pass;
				roots.push_back(x);
				break;
			}
			I.path.push_back(x);
			x = p;
		}
		for (int32_t y : I.path) { I.top[y] = t; I.top_stamp[y] = pass; }
	}
	if (roots.size() == 1) { out.push_back(mem); return; }
	// several pieces: merge what is connected
	const int np = (int) roots.size();
	for (int p = 0; p < np; p++) I.piece_of[roots[p]] = p;
	std::vector<std::vector<int32_t>> members(np);
	for (int32_t m : mem) members[I.piece_of[I.top[m]]].push_back(m);
	std::vector<int32_t> set_of(np), size(np);
	std::vector<std::vector<int32_t>> pieces_of(np);          // by set root: its pieces
	std::vector<uint8_t> closed(np, 0);
	// scanning position of every piece: (member, slot)
	std::vector<int32_t> cur_m(np, 0), cur_s(np, 0);
	for (int p = 0; p < np; p++) { set_of[p] = p; size[p] = (int32_t) members[p].size(); pieces_of[p].push_back(p); }
	auto find = [&](int p) { while (set_of[p] != p) { set_of[p] = set_of[set_of[p]]; p = set_of[p]; } return p; };
	for (;;) {
		// the smallest open set; when only one is open it is a component as it stands
		int X = -1, nopen = 0;
		for (int p = 0; p < np; p++)
			if (set_of[p] == p && !closed[p]) { nopen++; if (X < 0 || size[p] < size[X]) X = p; }
		if (nopen <= 1) break;
		bool linked = false;
		for (size_t pi = 0; pi < pieces_of[X].size() && !linked; pi++) {
			const int P = pieces_of[X][pi];
			while (cur_m[P] < (int32_t) members[P].size() && !linked) {
				const int32_t m = members[P][cur_m[P]];
				const int32_t *ids = &c->lp[(size_t) I.sel[m] * nlive];
				while (cur_s[P] < nlive && !linked) {
					const int32_t cid = I.cid_of[ids[cur_s[P]]];
					I.scanned_ids++;
					if (I.cnt[cid] > 1)
						for (int32_t e = I.head[cid]; e >= 0; e = I.next[e]) {
							const int32_t h = e / nlive;
							if (!I.alive[h]) continue;
							const int Y = find(I.piece_of[I.top[h]]);
							if (Y == X) continue;
							// a real edge (m, h) out of X: X's tree hangs below h from now on
							inc_reroot(I.parent, m);
							I.parent[m] = h;
							set_of[X] = Y;
							size[Y] += size[X];
							pieces_of[Y].insert(pieces_of[Y].end(), pieces_of[X].begin(), pieces_of[X].end());
							linked = true;
							break;
						}
					if (!linked) cur_s[P]++;
				}
				if (!linked) { cur_m[P]++; cur_s[P] = 0; }
			}
		}
		if (!linked) closed[X] = 1;
	}
	int nsets = 0;
	for (int p = 0; p < np; p++) {
		if (set_of[p] != p) continue;
		nsets++;
		std::vector<int32_t> all;
		for (int q : pieces_of[p]) all.insert(all.end(), members[q].begin(), members[q].end());
		std::sort(all.begin(), all.end());
		out.push_back(std::move(all));
	}
	if (nsets > 1) I.splits++;
}

// the selection shrank to `sel` (positions, ascending; a subset of the last one)
void inc_update(mdns_core *c, Incremental &I, const std::vector<int32_t> &sel)
{
	const int nlive = c->nlive;
	if (I.pass_token == 0x7fffffff) {
		std::fill(I.mark.begin(), I.mark.end(), 0);
		std::fill(I.top_stamp.begin(), I.top_stamp.end(), 0);
		std::fill(I.moved.begin(), I.moved.end(), 0);
		I.pass_token = 0;
	}
	const int32_t pass = ++I.pass_token;
	for (int32_t pos : sel) I.mark[I.k_of_pos[pos]] = pass;
	I.updates++;
	std::vector<IncComp> next;
	next.reserve(I.comps.size() + 4);
	std::vector<int32_t> kept;
	std::vector<std::vector<int32_t>> parts;
	for (IncComp &comp : I.comps) {
		kept.clear();
		bool any_dead_id = false, lost = false;
		for (int32_t m : comp.members) {
			if (I.mark[m] == pass) { kept.push_back(m); continue; }
			lost = true;
			I.alive[m] = 0;
			const int32_t *ids = &c->lp[(size_t) I.sel[m] * nlive];
			for (int s = 0; s < nlive; s++) if (--I.cnt[I.cid_of[ids[s]]] == 0) any_dead_id = true;
		}
		if (!lost) { next.push_back(std::move(comp)); continue; }
		if (kept.empty()) continue;
		parts.clear();
		inc_resolve(c, I, kept, parts);
		// ids: the largest part keeps the old list minus what nobody holds any more and minus what
		// the other parts hold; those are listed from their members' ids
		size_t big = 0;
		for (size_t t = 1; t < parts.size(); t++) if (parts[t].size() > parts[big].size()) big = t;
		std::vector<IncComp> made(parts.size());
		for (size_t t = 0; t < parts.size(); t++) {
			made[t].members = std::move(parts[t]);
			if (t == big) continue;
			for (int32_t m : made[t].members) {
				const int32_t *ids = &c->lp[(size_t) I.sel[m] * nlive];
				for (int s = 0; s < nlive; s++) {
					const int32_t cid = I.cid_of[ids[s]];
					if (I.moved[cid] != pass) { I.moved[cid] = pass; made[t].ids.push_back(ids[s]); }
				}
			}
			std::sort(made[t].ids.begin(), made[t].ids.end());
		}
		if (any_dead_id || parts.size() > 1) {
			std::vector<int32_t> &ids = comp.ids;
			size_t w = 0;
			for (int32_t q : ids) {
				const int32_t cid = I.cid_of[q];
				if (I.cnt[cid] > 0 && I.moved[cid] != pass) ids[w++] = q;
			}
			ids.resize(w);
		}
		made[big].ids = std::move(comp.ids);
		for (IncComp &m : made) next.push_back(std::move(m));
	}
	I.comps.swap(next);
}

// the groups of the current selection from the components kept: ascending lowest member, the
// reference's one-group shortcuts on top (see groups_graph)
void inc_emit(mdns_core *c, Incremental &I, const std::vector<int32_t> &sel, std::vector<Group> &out)
{
	out.clear();
	std::vector<const IncComp *> order;
	order.reserve(I.comps.size());
	long long nd = 0;
	for (const IncComp &comp : I.comps) { order.push_back(&comp); nd += (long long) comp.ids.size(); }
	std::sort(order.begin(), order.end(), [](const IncComp *a, const IncComp *b) { return a->members[0] < b->members[0]; });
	if (order.size() == 1 || nd < 2LL * c->nlive || !c->superpoints.empty()) {
		Group g;
		g.members = sel;
		if (order.size() == 1) g.ids = order[0]->ids;
		else {
			g.ids.reserve((size_t) nd);
			for (const IncComp *comp : order) g.ids.insert(g.ids.end(), comp->ids.begin(), comp->ids.end());
			std::sort(g.ids.begin(), g.ids.end());
		}
		out.push_back(std::move(g));
		return;
	}
	out.resize(order.size());
	for (size_t t = 0; t < order.size(); t++) {
		out[t].members.reserve(order[t]->members.size());
		for (int32_t k : order[t]->members) out[t].members.push_back(I.sel[k]);
		out[t].ids = order[t]->ids;
	}
}

}  // namespace
