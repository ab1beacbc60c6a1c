// Connected components of the FOCUSSED passes of one iteration, kept up to date instead of recomputed
// (included by host_sampler.cpp).
//
// Within an iteration the live-point matrix is fixed and the focussed selections only shrink: a pass
// selects the data sets whose shelf is still empty (multi_nested_sampler.py:373-388), and every draw
// of the pass fills a few shelves.  In the middle of a C2 run that is 90 - 190 passes per iteration
// over thousands of data sets that stay ONE component while a few dozen leave per pass -- and each pass
// paid a full components computation: 105 us on the device, 3 ns per (data set, id) pair on the host.
//
// Here the first focussed selection S1 of an iteration is analysed once -- union-find over its
// (data set, id) pairs, which also leaves: per id the chain of its holders, per id the number of
// holders still selected, and a rooted SPANNING FOREST of the data-set graph (parent pointers and child
// lists) whose every edge is a real one: two data sets sharing a live point.  A later pass removes the
// data sets that left:
//   * their ids lose a holder each (ids nobody holds any more leave the component's id list),
//   * the subtrees that hung below a data set that left are ORPHANS: pieces of the forest cut off from
//     their component's root,
//   * an orphan is hung back through a REPLACEMENT edge -- a live point one of its data sets shares with
//     a remaining data set outside it -- found by walking its data sets (depth first from its root, only
//     as far as needed) and the holder chains of their ids; the first id tried usually gives one.
//     Orphans that reach each other merge and go on looking together; a set of orphans that exhausts
//     all ids of all of its data sets without an edge out is a component of its own.  The piece with the
//     component's old root is never walked, so a pass costs what LEFT, not what stayed.
// The result -- groups in ascending order of their lowest data set, members and ids ascending, the
// reference's one-group shortcuts applied on top (multi_nested_sampler.py:283-297) -- is what a fresh
// computation gives: MDNS_CORE_CHECK_GROUPS=1 compares every pass, tests/test_core.py stresses it
// against scipy on planted graphs, tools/groups_replay.py replays an iteration of a real run.
namespace {

struct IncComp {
	std::vector<int32_t> members;   // selection indices k, ascending
	std::vector<int32_t> ids;       // live-point ids, ascending
	int32_t root = -1;              // root of its tree in the forest
};

struct Incremental {
	bool valid = false;
	int M1 = 0;
	std::vector<int32_t> sel;             // S1: position of selection index k
	std::vector<int32_t> k_of_pos;        // by position: k, or -1
	std::vector<uint8_t> alive;           // by k
	// the spanning forest, by k: parent (-1: root) and the list of children (first / next / previous sibling)
	std::vector<int32_t> parent, child, sib_next, sib_prev;
	std::vector<int32_t> uf;              // by k: union-find of the build
	std::vector<int32_t> head, cnt, first_k;   // by compact id: holder chain, holders still selected, first holder
	std::vector<int32_t> next;            // by edge (k * nlive + slot): next holder of the same id
	std::vector<int32_t> cid_of, cid_stamp;    // by id
	int32_t cid_token = 0;
	std::vector<IncComp> comps;
	// scratch
	std::vector<int32_t> mark, piece_of, piece_stamp, visited, slot_cur;   // by k
	std::vector<int32_t> moved;                                             // by compact id
	int32_t pass_token = 0;
	long long builds = 0, updates = 0, splits = 0, scanned_ids = 0, orphans = 0, chain_steps = 0;
};

inline int inc_find(std::vector<int32_t> &uf, int a)
{
	while (uf[a] != a) { uf[a] = uf[uf[a]]; a = uf[a]; }
	return a;
}

inline void inc_detach(Incremental &I, int32_t x)
{
	const int32_t p = I.parent[x];
	if (p < 0) return;
	const int32_t nx = I.sib_next[x], pv = I.sib_prev[x];
	if (pv >= 0) I.sib_next[pv] = nx; else I.child[p] = nx;
	if (nx >= 0) I.sib_prev[nx] = pv;
	I.parent[x] = -1;
}

inline void inc_attach(Incremental &I, int32_t x, int32_t p)
{
	I.parent[x] = p;
	I.sib_prev[x] = -1;
	I.sib_next[x] = I.child[p];
	if (I.child[p] >= 0) I.sib_prev[I.child[p]] = x;
	I.child[p] = x;
}

// the tree that holds x re-rooted at x
inline void inc_reroot(Incremental &I, int32_t x, std::vector<int32_t> &path)
{
	path.clear();
	for (int32_t y = x; y >= 0; y = I.parent[y]) path.push_back(y);
	for (size_t i = path.size(); i-- > 1;) {
		inc_detach(I, path[i - 1]);                    // path[i-1] was a child of path[i] ...
	}
	for (size_t i = 1; i < path.size(); i++) inc_attach(I, path[i], path[i - 1]);     // ... which now hangs below it
}

inline int32_t inc_root_of(const Incremental &I, int32_t x)
{
	while (I.parent[x] >= 0) x = I.parent[x];
	return x;
}

// S1: everything from scratch
void inc_build(mdns_core *c, Incremental &I, const std::vector<int32_t> &sel)
{
	const int M = (int) sel.size(), nlive = c->nlive;
	I.valid = false;
	I.M1 = M;
	I.sel = sel;
	I.k_of_pos.assign((size_t) c->nrun, -1);
	for (int k = 0; k < M; k++) I.k_of_pos[sel[k]] = k;
	I.alive.assign(M, 1);
	I.parent.assign(M, -1);
	I.child.assign(M, -1);
	I.sib_next.assign(M, -1);
	I.sib_prev.assign(M, -1);
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
				const int32_t j = I.first_k[cid];
				const int a = inc_find(I.uf, k), b = inc_find(I.uf, j);
				if (a != b) {
					tree.push_back(k);
					tree.push_back(j);
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
	std::vector<int32_t> comp_of(M, -1);
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
		for (IncComp &comp : I.comps) {
			const int32_t root = comp.members[0];
			comp.root = root;
			seen[root] = 1;
			queue.clear();
			queue.push_back(root);
			for (size_t at = 0; at < queue.size(); at++) {
				const int32_t x = queue[at];
				for (int32_t t = start[x]; t < start[x + 1]; t++) {
					const int32_t y = adj[t];
					if (!seen[y]) { seen[y] = 1; inc_attach(I, y, x); queue.push_back(y); }
				}
			}
		}
	}
	I.mark.assign(M, 0);
	I.piece_of.assign(M, 0);
	I.piece_stamp.assign(M, 0);
	I.visited.assign(M, 0);
	I.slot_cur.assign(M, 0);
	I.moved.assign(I.head.size(), 0);
	I.pass_token = 0;
	I.valid = true;
	I.builds++;
}

// A component lost the data sets `gone` (already marked dead; their ids' counts lowered).  Hangs the
// orphaned subtrees back where they still connect and appends, for every part that did NOT find its way
// back, its members (ascending) to `parts`; the component's remaining members are then `kept` minus
// those.  *root receives the root of the remaining main part.
void inc_resolve(mdns_core *c, Incremental &I, IncComp &comp, const std::vector<int32_t> &gone,
                 std::vector<std::vector<int32_t>> &parts, std::vector<int32_t> &path)
{
	const int nlive = c->nlive;
	const int32_t pass = I.pass_token;
	// pieces: the orphaned subtrees, and -- last, never walked -- the tree of the old root if that stayed
	std::vector<int32_t> roots;
	for (int32_t L : gone) {
		inc_detach(I, L);
		for (int32_t ch = I.child[L]; ch >= 0;) {
			const int32_t nx = I.sib_next[ch];
			I.parent[ch] = -1;
			I.sib_next[ch] = I.sib_prev[ch] = -1;
			if (I.alive[ch]) roots.push_back(ch);
			ch = nx;
		}
		I.child[L] = -1;
	}
	const bool has_main = I.alive[comp.root];
	if (roots.empty()) return;
	I.orphans += (long long) roots.size();
	if (has_main) roots.push_back(comp.root);
	const int np = (int) roots.size();
	if (np == 1) { comp.root = roots[0]; return; }
	for (int p = 0; p < np; p++) { I.piece_of[roots[p]] = p; I.piece_stamp[roots[p]] = pass; }
	std::vector<int32_t> set_of(np), next_piece(np, -1), last_piece(np);
	std::vector<uint8_t> closed(np, 0);
	std::vector<std::vector<int32_t>> stack(np), walked(np);       // per piece: nodes to expand, nodes done
	for (int p = 0; p < np; p++) {
		set_of[p] = p; last_piece[p] = p;
		stack[p].push_back(roots[p]);
		I.slot_cur[roots[p]] = 0;
		I.visited[roots[p]] = pass;
	}
	auto find = [&](int p) { while (set_of[p] != p) { set_of[p] = set_of[set_of[p]]; p = set_of[p]; } return p; };
	auto set_of_node = [&](int32_t h) { return find(I.piece_of[inc_root_of(I, h)]); };
	// NOTE: inc_root_of(h) is the root of h's TREE, which after a merge is the root of the piece the tree
	// was hung into: its piece number leads to the merged set through `set_of`.
	int nopen = np;
	const int nscan = has_main ? np - 1 : np;                      // (the main piece is not walked)
	for (int oi = 0; oi < nscan && nopen > 1; oi++) {
		if (set_of[oi] != oi) continue;                             // merged: walked in the turn of its set's root
		int X = oi;
		bool linked = false;
		for (int P = X; P >= 0 && !linked; P = next_piece[P]) {
			std::vector<int32_t> &st = stack[P];
			while (!st.empty() && !linked) {
				const int32_t v = st.back();
				const int32_t *ids = &c->lp[(size_t) I.sel[v] * nlive];
				int s = I.slot_cur[v];
				for (; s < nlive && !linked; s++) {
					const int32_t cid = I.cid_of[ids[s]];
					I.scanned_ids++;
					if (I.cnt[cid] <= 1 || I.moved[cid] == -pass) continue;
					// the chain of the id's holders: those that left are unlinked on the way (each once); an
					// id whose remaining holders all sit in ONE set is of no use to anybody any more in this
					// call (sets only grow) and is marked so
					int32_t prev = -1;
					bool all_inside = true;
					for (int32_t e = I.head[cid]; e >= 0; e = I.next[e]) {
						const int32_t h = e / nlive;
						I.chain_steps++;
						if (!I.alive[h]) {
							if (prev < 0) I.head[cid] = I.next[e]; else I.next[prev] = I.next[e];
							continue;
						}
						prev = e;
						if (h == v) continue;
						const int Y = set_of_node(h);
						if (Y == X) continue;
						all_inside = false;
						// a real edge (v, h) out of X: X's tree hangs below h from now on
						inc_reroot(I, v, path);
						inc_attach(I, v, h);
						set_of[X] = Y;
						next_piece[last_piece[Y]] = X;
						last_piece[Y] = last_piece[X];
						nopen--;
						linked = true;
						break;
					}
					if (all_inside) I.moved[cid] = -pass;
				}
				if (linked) { I.slot_cur[v] = s - 1; break; }
				// all ids of v lead nowhere new: v is done, its children are next
				st.pop_back();
				walked[P].push_back(v);
				// (a re-rooted tree lists a data set's former parent among its children: walked already)
				for (int32_t ch = I.child[v]; ch >= 0; ch = I.sib_next[ch])
					if (I.visited[ch] != pass) { I.visited[ch] = pass; I.slot_cur[ch] = 0; st.push_back(ch); }
			}
		}
		if (!linked) { closed[X] = 1; nopen--; }
	}
	// closed sets are components of their own; whatever is still open is ONE set (the rest): it keeps
	// the component
	int main_set = -1;
	for (int p = 0; p < np; p++) if (set_of[p] == p && !closed[p]) main_set = p;
	if (main_set < 0) {
		// every set closed (no main piece, or it was walked): the last closed one keeps the component
		for (int p = np - 1; p >= 0; p--) if (set_of[p] == p) { main_set = p; break; }
	}
	for (int p = 0; p < np; p++) {
		if (set_of[p] != p || p == main_set) continue;
		std::vector<int32_t> all;
		for (int q = p; q >= 0; q = next_piece[q]) all.insert(all.end(), walked[q].begin(), walked[q].end());
		std::sort(all.begin(), all.end());
		parts.push_back(std::move(all));
	}
	comp.root = inc_root_of(I, roots[main_set]);
	if (!parts.empty()) I.splits++;
}

// the selection shrank to `sel` (positions, ascending; a subset of the last one)
void inc_update(mdns_core *c, Incremental &I, const std::vector<int32_t> &sel)
{
	const int nlive = c->nlive;
	if (I.pass_token == 0x3fffffff) {
		std::fill(I.mark.begin(), I.mark.end(), 0);
		std::fill(I.piece_stamp.begin(), I.piece_stamp.end(), 0);
		std::fill(I.visited.begin(), I.visited.end(), 0);
		std::fill(I.moved.begin(), I.moved.end(), 0);
		I.pass_token = 0;
	}
	const int32_t pass = ++I.pass_token;
	for (int32_t pos : sel) I.mark[I.k_of_pos[pos]] = pass;
	I.updates++;
	std::vector<IncComp> next;
	next.reserve(I.comps.size() + 4);
	std::vector<int32_t> kept, gone, dead, path;
	std::vector<std::vector<int32_t>> parts;
	for (IncComp &comp : I.comps) {
		kept.clear(); gone.clear(); dead.clear();
		for (int32_t m : comp.members) {
			if (I.mark[m] == pass) { kept.push_back(m); continue; }
			gone.push_back(m);
			I.alive[m] = 0;
			const int32_t *ids = &c->lp[(size_t) I.sel[m] * nlive];
			for (int s = 0; s < nlive; s++) if (--I.cnt[I.cid_of[ids[s]]] == 0) dead.push_back(ids[s]);
		}
		if (gone.empty()) { next.push_back(std::move(comp)); continue; }
		if (kept.empty()) continue;
		parts.clear();
		inc_resolve(c, I, comp, gone, parts, path);
		if (parts.empty()) {
			// still one component: the ids nobody holds any more leave the (ascending) list -- one
			// sequential pass against their sorted list
			if (!dead.empty()) {
				std::sort(dead.begin(), dead.end());
				std::vector<int32_t> &ids = comp.ids;
				size_t w = 0, d = 0;
				const size_t nd = dead.size();
				for (int32_t q : ids) {
					if (d < nd && dead[d] == q) { d++; continue; }
					ids[w++] = q;
				}
				ids.resize(w);
			}
			comp.members.swap(kept);
			next.push_back(std::move(comp));
			continue;
		}
		// parts split off: they list their ids from their members' (marked `moved`), the rest keeps the
		// old list minus those and minus what nobody holds any more
		for (int32_t m : kept) I.piece_stamp[m] = 0;                // (reused below as "split off" marks: cleared first)
		std::vector<IncComp> made(parts.size());
		for (size_t t = 0; t < parts.size(); t++) {
			made[t].members = std::move(parts[t]);
			made[t].root = inc_root_of(I, made[t].members[0]);
			for (int32_t m : made[t].members) {
				I.piece_stamp[m] = -pass;
				const int32_t *ids = &c->lp[(size_t) I.sel[m] * nlive];
				for (int s = 0; s < nlive; s++) {
					const int32_t cid = I.cid_of[ids[s]];
					if (I.moved[cid] != pass) { I.moved[cid] = pass; made[t].ids.push_back(ids[s]); }
				}
			}
			std::sort(made[t].ids.begin(), made[t].ids.end());
		}
		{
			std::vector<int32_t> &ids = comp.ids;
			size_t w = 0;
			for (int32_t q : ids) {
				const int32_t cid = I.cid_of[q];
				if (I.cnt[cid] > 0 && I.moved[cid] != pass) ids[w++] = q;
			}
			ids.resize(w);
			size_t wm = 0;
			for (int32_t m : kept) if (I.piece_stamp[m] != -pass) kept[wm++] = m;
			kept.resize(wm);
			comp.members.swap(kept);
		}
		next.push_back(std::move(comp));
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
