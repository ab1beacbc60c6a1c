// Grouping of data sets that share live points, on the device (include/mdns.h, Part 4).
//
// The reference's default grouping (`generate_subsets_graph`, multi_nested_sampler.py:268-355)
// builds the bipartite graph {data sets} -- {live points they hold} with igraph and asks for
// its connected components (`graph.clusters()`, :325); with ONE component -- the rule late in a
// run -- it hands on the selection and the ascending distinct ids (`numpy.unique`, :279,:331).
// Both are order-free integer results, which is what a GPU can produce without a sequential
// walk: minimum-label propagation over the edges.
//
//   idsT   int32[ndata][nlive]   the id matrix `live_pointsp` (:108) transposed, resident: the ids
//                                of a data set are contiguous; one entry per running data set
//                                changes per iteration (mdns_groups_replace)
//   label  int32[ndata]          label of a data set: starts as its own index
//   plabel {stamp, label}[npoints]  label of a live point, valid when `stamp` is the number of the
//                                current call -- otherwise the point is unclaimed: "above every
//                                index" -- so that no call has to clear npoints entries first (the
//                                600 KB fill per call of the first version); one aligned 8-byte
//                                word, read and written whole
//
// One wave per selected data set, one kernel per ROUND: the wave pulls the minimum over
// label[d], the plabel of its ids and the label of that minimum, and pushes it to whatever
// stands higher.
// Labels only ever take the index of a data set of the same component, the component's lowest
// data set keeps its own, so the only state no edge wants to change is "everything in a
// component carries its lowest index" -- reached after a few rounds (the graphs are dense: a
// live point is shared by ~100 data sets), detected by a round that changes nothing.
//
// No read-modify-write atomics and no device-scope traffic in the rounds.  Measured on this chip:
// a million atomicMin on ~10^4 addresses take 70-100 us and ten thousand on ONE address 3 ms
// (the hot case is real -- early in a run every data set holds the same initial points); a
// lock-free union-find spends milliseconds chasing and halving paths through a few hundred hot
// roots; device-scope (sc1) stores cost 250 us per million.  So labels are read and written with
// PLAIN cached accesses: an XCD sees its own stores at once and the others' at the next kernel
// boundary.  That only costs rounds, never correctness: a store that loses a race against a
// higher label is redone in the next round by the wave that still sees the difference, and the
// deciding round starts from coherent memory (kernel boundary) and, storing nothing, has read a
// consistent state.  Only the per-round "something moved" flag is a device-scope store, at
// most one per wave.
#include "mdns_internal.h"

#include <atomic>
#include <cstddef>
#include <cstring>

namespace mdns {

static constexpr int kBlock = 256;
static constexpr int kSweeps = 2;                     // pull-and-push passes per launch of k_groups_round
static constexpr int kMaxRounds = 64;                 // rounds per batch: their "changed" flags are in the header
static constexpr int kRoundLimit = 1 << 20;           // a call gives up after this many rounds (never seen)

// counts, failure bits (1 = id out of range, 2 = bad replacement) and, per round, whether it
// still moved a label
struct GroupsHeader { int status; int pad; int ncomponents; int ndistinct; int changed[kMaxRounds]; };   // a call clears [ncomponents, end)
// the same and the list of distinct ids in host memory mapped into the device: written by the
// last kernel of a call (k_groups_compact), `seq` last; the host polls instead of copying
struct GroupsBox { unsigned long long seq; unsigned long long pad; GroupsHeader header; int distinct[1]; };

// Device-scope relaxed accesses (the rounds' "moved" flag).
__device__ __forceinline__ int load_relaxed(const int *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_relaxed(int *p, int v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a point's label in this call: its stored label when the stamp is this call's, else unclaimed
typedef unsigned long long PLabel;
__device__ __forceinline__ int plabel_of_point(PLabel w, int call) { return (int) (w >> 32) == call ? (int) (unsigned) w : 0x7f7f7f7f; }
__device__ __forceinline__ PLabel plabel_make(int call, int label) { return ((PLabel) (unsigned) call << 32) | (unsigned) label; }

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const int other = __shfl_xor(v, o, 64);
		v = other < v ? other : v;
	}
	return v;
}

// One WAVE per selected data set d (its ids, idsT[d][0..nlive), are contiguous: lanes take slots
// lane, lane + 64, ...): the wave pulls m = min(label[d], plabel of its ids, label[that minimum]),
// lane 0 lowers label[d], every lane lowers the plabel of its ids that stand above m.  Data
// sets are dispatched in ascending order, so a popular id is usually claimed by one of its lowest
// holders before the others look: a handful of stores per id instead of one per holder (a
// store per edge made a round of 10^6 edges 250 us; a round that only looks takes 6).
__global__ __launch_bounds__(kBlock) void k_groups_round(const int *__restrict__ idsT, int nlive,
                                                         int *rows, const int *__restrict__ rows_host, int M, long long npoints,
                                                         PLabel *plabel, int *label, int first, int call, int flag,
                                                         int *__restrict__ changed, int *__restrict__ status, int sweeps)
{
	const int lane = threadIdx.x & 63;
	const int i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (i >= M) return;                                               // whole waves
	// the first launch of a call takes the selection from the host block mapped into the device
	// (one 4-byte read per wave instead of a copy command in front of the launch) and leaves it in
	// device memory for the launches that follow
	int d = i;
	if (rows_host) {
		d = rows_host[i];
		if (lane == 0) rows[i] = d;
	} else if (rows) d = rows[i];
	const int *mine = idsT + (size_t) d * nlive;
	bool moved = false;
	// `sweeps` passes of the same pull-and-push per launch: what a wave sees of the others' stores
	// within a launch is whatever has reached its caches -- stale values only cost progress, as
	// between launches -- and a launch in which NO sweep moved anything started from coherent memory
	// and found the fixed point, exactly like a single-sweep round (each launch costs ~8 us of
	// dispatch against ~5 us of work: two sweeps per launch halve the launches a call needs)
	for (int sweep = 0; sweep < sweeps; sweep++) {
		const bool start = first && sweep == 0;
		// the first round of a call starts every data set from its own index (label[] still holds
		// the previous call's values: nobody reads another data set's label in this round)
		const int l = start ? d : label[d];
		int m = l;
		bool bad = false;
		for (int p = lane; p < nlive; p += 64) {
			const int q = mine[p];
			if (q < 0 || q >= npoints) { bad = true; continue; }
			const int pl = plabel_of_point(plabel[q], call);
			m = pl < m ? pl : m;
		}
		if (__ballot(bad) != 0ull) { if (lane == 0) atomicOr(status, 1); return; }
		m = wave_min(m);
		if (!start) {
			// m is a data set of this component: follow ITS label down to a data set that keeps its
			// own (labels only point to lower indices, so this ends) ...
			while (true) {
				const int lm = label[m];
				if (lm >= m) break;
				m = lm;
			}
			// ... and hang the data set d stood under below it too: everybody who still points at
			// that one gets there with the next jump (without this a path of n data sets in random
			// order needs ~n rounds; with it a few dozen for n = 20 000)
			if (lane == 0 && m < l && m < label[l]) label[l] = m;
		}
		for (int p = lane; p < nlive; p += 64) {
			const int q = mine[p];
			if (m < plabel_of_point(plabel[q], call)) { plabel[q] = plabel_make(call, m); moved = true; }
		}
		if (lane == 0 && (m < l || start)) { label[d] = m; moved = moved || m < l; }
		if (first && sweeps > 1 && sweep == 0) {
			// the sweep that follows reads other data sets' labels: all of them must have been
			// initialised by THIS launch first (label[] holds the previous call's values), which
			// only the next launch guarantees -- so the first launch of a call makes one sweep
			break;
		}
	}
	// one flag for the round -- the call's number: nothing has to be cleared between calls -- raised
	// by a wave that moved something unless it is up already
	if (__ballot(moved) != 0ull && lane == 0 && load_relaxed(changed) != flag) store_relaxed(changed, flag);
}

// [rows][cols] -> [cols][rows]
__global__ __launch_bounds__(kBlock) void k_groups_transpose(const int *__restrict__ src, int nrows, int ncols, int *__restrict__ dst)
{
	__shared__ int tile[32][33];
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;               // 32 x 8
	const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
	for (int k = ty; k < 32; k += 8)
		if (r0 + k < nrows && c0 + tx < ncols) tile[k][tx] = src[(size_t) (r0 + k) * ncols + c0 + tx];
	__syncthreads();
	for (int k = ty; k < 32; k += 8)
		if (c0 + k < ncols && r0 + tx < nrows) dst[(size_t) (c0 + k) * nrows + r0 + tx] = tile[tx][k];
}

// After the rounds, one launch: workgroups [0, finish_blocks) note the label of every selected data
// set and count the components (a data set that kept its own index); the others build the bit map
// of the ids held -- bit q = some selected data set holds live point q: a wave's ballot is the word.
__global__ __launch_bounds__(kBlock) void k_groups_finish(const int *__restrict__ label, const int *__restrict__ rows, int M,
                                                          int *__restrict__ labels, GroupsHeader *__restrict__ header, int finish_blocks,
                                                          const PLabel *__restrict__ plabel, long long npoints, int call,
                                                          unsigned long long *__restrict__ touched)
{
	if ((int) blockIdx.x < finish_blocks) {
		const int i = blockIdx.x * kBlock + threadIdx.x;
		if (i >= M) return;
		const int d = rows ? rows[i] : i;
		const int l = label[d];
		labels[i] = l;
		if (l == d) atomicAdd(&header->ncomponents, 1);
		return;
	}
	const long long q = (long long) (blockIdx.x - finish_blocks) * kBlock + threadIdx.x;
	const bool held = q < npoints && (int) (plabel[q] >> 32) == call;
	const unsigned long long word = __ballot(held);
	if ((threadIdx.x & 63) == 0 && q < npoints) touched[q >> 6] = word;
}

// The set bits of the map as an ascending list of ids -- numpy.unique of the selected columns
// (multi_nested_sampler.py:279) -- by ONE workgroup: popcounts of contiguous runs of words, a scan
// over the 1024 run totals, then every thread writes out its run.
__global__ __launch_bounds__(1024) void k_groups_compact(const unsigned long long *__restrict__ touched, long long nwords,
                                                         int *scratch, GroupsHeader *__restrict__ header,
                                                         GroupsBox *__restrict__ box, unsigned long long seq)
{
	__shared__ int wave_total[16];
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const long long run = (nwords + 1023) / 1024;
	const long long w0 = t * run, w1 = w0 + run < nwords ? w0 + run : nwords;
	int mine = 0;
	for (long long w = w0; w < w1; w++) mine += __popcll(touched[w]);
	int incl = mine;                                                   // inclusive scan inside the wave
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int up = __shfl_up(incl, o, 64);
		if (lane >= o) incl += up;
	}
	if (lane == 63) wave_total[wave] = incl;
	__syncthreads();
	int before = 0;
	for (int k = 0; k < wave; k++) before += wave_total[k];
	int at = before + incl - mine;
	for (long long w = w0; w < w1; w++) {
		unsigned long long bits = touched[w];
		while (bits) {
			scratch[at++] = (int) (w * 64 + __builtin_ctzll(bits));
			bits &= bits - 1;
		}
	}
	__shared__ int s_total;
	if (t == 1023) s_total = before + incl;
	__threadfence_block();
	__syncthreads();
	// the list goes to the host in one sweep of coalesced stores (`box` is host memory mapped into
	// the device; every thread writing its own run there 4 bytes at a time took 3x as long)
	const int total = s_total;
	for (int e = t; e < total; e += 1024) mail_store(&box->distinct[e], scratch[e]);
	// counts, failure bits and the rounds' flags with it; `seq` last: the host polls for it
	if (t < kMaxRounds) mail_store(&box->header.changed[t], header->changed[t]);
	if (t == 1023) {
		mail_store(&box->header.ncomponents, header->ncomponents);
		mail_store(&box->header.ndistinct, total);
		mail_store(&box->header.status, header->status);
		header->ncomponents = 0;                                       // k_groups_finish of the next call counts from here
	}
	handover_release();
	__syncthreads();
	if (t == 0) mail_raise(&box->seq, seq);
}

__global__ __launch_bounds__(kBlock) void k_groups_point_labels(const PLabel *__restrict__ plabel, long long npoints, int call,
                                                                int *__restrict__ point_labels)
{
	const long long q = (long long) blockIdx.x * kBlock + threadIdx.x;
	if (q >= npoints) return;
	const PLabel w = plabel[q];
	point_labels[q] = (int) (w >> 32) == call ? (int) (unsigned) w : -1;
}

// labels of the ids a components call listed (its ascending list is still in `list`), in that order
__global__ __launch_bounds__(kBlock) void k_groups_id_labels(const PLabel *__restrict__ plabel, const int *list, long long n,
                                                             int call, int *out)
{
	const long long e = (long long) blockIdx.x * kBlock + threadIdx.x;
	if (e >= n) return;
	const PLabel w = plabel[list[e]];
	out[e] = (int) (w >> 32) == call ? (int) (unsigned) w : -1;
}

__global__ __launch_bounds__(kBlock) void k_groups_replace(int *__restrict__ idsT, int ndata, int nlive,
                                                           const int *__restrict__ rows, const int *__restrict__ slots,
                                                           const int *__restrict__ new_ids, int n, int *__restrict__ status)
{
	const int i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const int d = rows[i], p = slots[i];
	if (d < 0 || d >= ndata || p < 0 || p >= nlive) { atomicOr(status, 2); return; }
	idsT[(size_t) d * nlive + p] = new_ids[i];
}

}  // namespace mdns

using namespace mdns;

struct mdns_groups {
	int nlive = 0, ndata = 0;
	int *d_idsT = nullptr;             // [ndata][nlive]
	int *d_tmp = nullptr;              // [nlive][ndata] staging of set_ids / get_ids
	int *d_label = nullptr;            // [ndata] by data set
	int *d_rows = nullptr;             // [3 * ndata]: selection, or (rows | slots | new ids) of a replacement
	int *d_labels = nullptr;           // [ndata] by position in the selection
	// per id; grown with the pile of accepted points.  One block:
	//   header | touched bit map | plabel | point labels (out)
	char *d_points = nullptr;  long long cap_points = 0;
	char *h_pin = nullptr;  size_t pin_bytes = 0;
	GroupsBox *h_box = nullptr, *h_box_dev = nullptr;  long long box_cap = 0;   // room for box_cap ids
	unsigned long long box_seq = 0;
	bool have_ids = false;
	int last_M = -1;  long long last_npoints = 0, last_ndistinct = 0;      // of the last components call (-1: ids changed since)
	int rounds_hint = 4;                               // rounds the next call launches before it looks
	int call = 0;                                      // number of the current components call (stamps)
	int label_call = 0;                                // the stamp the point labels of the last call carry
	char *h_rows = nullptr, *h_rows_dev = nullptr;  size_t rows_bytes = 0;    // pinned + mapped block of a call's selection (its own:
	                                                   // every call ends by polling, so it is free at the next one)
	long long rounds_total = 0, calls_total = 0;
};

static constexpr size_t kHeaderBytes = 512;            // sizeof(GroupsHeader) rounded up: keeps what follows aligned
static_assert(sizeof(GroupsHeader) <= kHeaderBytes, "header grew");
static size_t words_of(long long npoints) { return (size_t) ((npoints + 63) / 64); }
static GroupsHeader *hdr_of(mdns_groups *g) { return (GroupsHeader *) g->d_points; }
static unsigned long long *touched_of(mdns_groups *g) { return (unsigned long long *) (g->d_points + kHeaderBytes); }
static PLabel *plabel_of(mdns_groups *g) { return (PLabel *) (touched_of(g) + words_of(g->cap_points)); }
static int *pout_of(mdns_groups *g) { return (int *) (plabel_of(g) + g->cap_points); }

static char *groups_pin(mdns_groups *g, size_t bytes)
{
	if (bytes <= g->pin_bytes) return g->h_pin;
	Context *c = ctx();
	if (g->h_pin) { (void) hipStreamSynchronize(c->stream); (void) hipHostFree(g->h_pin); g->h_pin = nullptr; g->pin_bytes = 0; }
	const size_t want = bytes + bytes / 2 + 4096;
	if (!MDNS_HIP(hipHostMalloc((void **) &g->h_pin, want, hipHostMallocDefault))) return nullptr;
	g->pin_bytes = want;
	return g->h_pin;
}

static bool groups_fit_points(mdns_groups *g, long long npoints)
{
	if (npoints <= g->cap_points) return true;
	Context *c = ctx();
	long long cap = g->cap_points > 0 ? g->cap_points : 4096;
	while (cap < npoints) cap *= 2;                                    // (a multiple of 64: the regions stay 8-byte aligned)
	if (g->d_points) { (void) hipStreamSynchronize(c->stream); (void) hipFree(g->d_points); g->d_points = nullptr; g->cap_points = 0; }
	const size_t bytes = kHeaderBytes + words_of(cap) * 8 + (size_t) cap * (sizeof(PLabel) + sizeof(int));
	if (!MDNS_HIP(hipMalloc((void **) &g->d_points, bytes))) return false;
	g->cap_points = cap;
	g->last_M = -1;
	// header and stamps start at zero: call numbers start at 1
	return MDNS_HIP(hipMemsetAsync(g->d_points, 0, kHeaderBytes + words_of(cap) * 8 + (size_t) cap * sizeof(PLabel), c->stream));
}

// mapped host block with room for `ids` distinct ids
static bool groups_fit_box(mdns_groups *g, long long ids)
{
	if (g->h_box && ids <= g->box_cap) return true;
	Context *c = ctx();
	long long cap = g->box_cap > 0 ? g->box_cap : 4096;
	while (cap < ids) cap *= 2;
	if (g->h_box) { (void) hipStreamSynchronize(c->stream); (void) hipHostFree(g->h_box); g->h_box = nullptr; g->box_cap = 0; }
	const size_t bytes = sizeof(GroupsBox) + (size_t) cap * sizeof(int);
	if (!MDNS_HIP(hipHostMalloc((void **) &g->h_box, bytes, hipHostMallocMapped | hipHostMallocCoherent)) ||
	    !MDNS_HIP(hipHostGetDevicePointer((void **) &g->h_box_dev, g->h_box, 0))) return false;
	memset(g->h_box, 0, sizeof(GroupsBox));
	g->h_box->seq = g->box_seq;
	g->box_cap = cap;
	return true;
}

extern "C" void mdns_groups_destroy(mdns_groups *g)
{
	if (!g) return;
	Context *c = ctx();
	if (c) (void) hipStreamSynchronize(c->stream);
	if (g->h_box) (void) hipHostFree(g->h_box);
	if (g->h_rows) (void) hipHostFree(g->h_rows);
	void *bufs[] = {g->d_idsT, g->d_tmp, g->d_label, g->d_rows, g->d_labels, g->d_points};
	for (void *b : bufs) if (b) (void) hipFree(b);
	if (g->h_pin) (void) hipHostFree(g->h_pin);
	delete g;
}

extern "C" mdns_groups *mdns_groups_create(int nlive, int ndata)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (nlive <= 0 || ndata <= 0) { set_error("mdns_groups_create: nlive=%d ndata=%d", nlive, ndata); return nullptr; }
	mdns_groups *g = new mdns_groups();
	g->nlive = nlive; g->ndata = ndata;
	const size_t nd = (size_t) ndata;
	const bool ok =
	    MDNS_HIP(hipMalloc((void **) &g->d_idsT, (size_t) nlive * nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &g->d_tmp, (size_t) nlive * nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &g->d_label, nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &g->d_rows, 3 * nd * sizeof(int))) &&
	    MDNS_HIP(hipMalloc((void **) &g->d_labels, nd * sizeof(int))) &&
	    groups_fit_points(g, 4096);
	if (!ok) { mdns_groups_destroy(g); return nullptr; }
	return g;
}

extern "C" int mdns_groups_set_ids(mdns_groups *g, const int32_t *ids)
{
	Context *c = ctx();
	if (!c || !g || !ids) return 1;
	const size_t bytes = (size_t) g->nlive * g->ndata * sizeof(int);
	if (!MDNS_HIP(hipMemcpyAsync(g->d_tmp, ids, bytes, hipMemcpyHostToDevice, c->stream))) return 1;
	hipLaunchKernelGGL(k_groups_transpose, dim3((g->ndata + 31) / 32, (g->nlive + 31) / 32), dim3(kBlock), 0, c->stream,
	                   g->d_tmp, g->nlive, g->ndata, g->d_idsT);
	if (!MDNS_HIP(hipGetLastError()) || !MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
	g->have_ids = true;
	g->last_M = -1;
	return 0;
}

extern "C" int mdns_groups_get_ids(mdns_groups *g, int32_t *ids)
{
	Context *c = ctx();
	if (!c || !g || !ids) return 1;
	const size_t bytes = (size_t) g->nlive * g->ndata * sizeof(int);
	hipLaunchKernelGGL(k_groups_transpose, dim3((g->nlive + 31) / 32, (g->ndata + 31) / 32), dim3(kBlock), 0, c->stream,
	                   g->d_idsT, g->ndata, g->nlive, g->d_tmp);
	return MDNS_HIP(hipGetLastError()) &&
	       MDNS_HIP(hipMemcpyAsync(ids, g->d_tmp, bytes, hipMemcpyDeviceToHost, c->stream)) &&
	       MDNS_HIP(hipStreamSynchronize(c->stream)) ? 0 : 1;
}

extern "C" int mdns_groups_replace(mdns_groups *g, const int32_t *rows, const int32_t *slots, const int32_t *new_ids, int n)
{
	Context *c = ctx();
	if (!c || !g) return 1;
	if (n < 0 || n > g->ndata || (n > 0 && (!rows || !slots || !new_ids))) { set_error("mdns_groups_replace: n=%d", n); return 1; }
	if (!g->have_ids) { set_error("mdns_groups_replace: no id matrix yet (mdns_groups_set_ids)"); return 1; }
	if (n == 0) return 0;
	// rows | slots | new ids: one pinned block, one copy.  Nothing waits: a bad index raises a
	// bit in the header that the next mdns_groups_components reports.
	char *pin = groups_pin(g, (size_t) 3 * n * sizeof(int));
	if (!pin) return 1;
	// (the staging block may still feed the previous copy)
	if (!MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
	memcpy(pin, rows, (size_t) n * 4);
	memcpy(pin + (size_t) n * 4, slots, (size_t) n * 4);
	memcpy(pin + (size_t) n * 8, new_ids, (size_t) n * 4);
	if (!MDNS_HIP(hipMemcpyAsync(g->d_rows, pin, (size_t) 3 * n * 4, hipMemcpyHostToDevice, c->stream))) return 1;
	hipLaunchKernelGGL(k_groups_replace, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
	                   g->d_idsT, g->ndata, g->nlive, g->d_rows, g->d_rows + n, g->d_rows + 2 * n, n, &hdr_of(g)->status);
	g->last_M = -1;
	return MDNS_HIP(hipGetLastError()) ? 0 : 1;
}

extern "C" int mdns_groups_components(mdns_groups *g, const int32_t *rows, int M, long long npoints,
                                      int *ncomponents, long long *ndistinct, int32_t *distinct, long long cap,
                                      unsigned long long *touched)
{
	Context *c = ctx();
	if (!c || !g || !ncomponents) return 1;
	if (!g->have_ids) { set_error("mdns_groups_components: no id matrix yet (mdns_groups_set_ids)"); return 1; }
	if (M <= 0 || M > g->ndata || npoints <= 0 || (!rows && M != g->ndata) || (distinct && cap < 0)) {
		set_error("mdns_groups_components: M=%d (ndata=%d) npoints=%lld", M, g->ndata, npoints);
		return 1;
	}
	if (rows)
		for (int i = 0; i < M; i++)
			if (rows[i] < 0 || rows[i] >= g->ndata || (i > 0 && rows[i] <= rows[i - 1])) {
				set_error("mdns_groups_components: rows must be ascending indices below %d", g->ndata);
				return 1;
			}
	// (a pending replacement wrote its failure bit into the old block: keep it across a regrowth)
	if (npoints > g->cap_points) {
		int status = 0;
		if (!MDNS_HIP(hipMemcpyAsync(&status, &hdr_of(g)->status, sizeof(int), hipMemcpyDeviceToHost, c->stream)) ||
		    !MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
		if (!groups_fit_points(g, npoints)) return 1;
		if (status && !MDNS_HIP(hipMemcpyAsync(&hdr_of(g)->status, &status, sizeof(int), hipMemcpyHostToDevice, c->stream))) return 1;
		if (!MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
	}
	const size_t nw = words_of(npoints);
	// the most this selection can hold
	const long long most = (long long) M * g->nlive < npoints ? (long long) M * g->nlive : npoints;
	if (!groups_fit_box(g, most)) return 1;
	if (rows) {
		// (the selection's own pinned block: the previous call ended by polling for its outcome, so
		// nothing reads the block any more -- no stream synchronisation)
		if ((size_t) M * 4 > g->rows_bytes) {
			if (g->h_rows) { (void) hipStreamSynchronize(c->stream); (void) hipHostFree(g->h_rows); g->h_rows = nullptr; g->rows_bytes = 0; }
			if (!MDNS_HIP(hipHostMalloc((void **) &g->h_rows, (size_t) g->ndata * 4, hipHostMallocMapped)) ||
			    !MDNS_HIP(hipHostGetDevicePointer((void **) &g->h_rows_dev, g->h_rows, 0))) return 1;
			g->rows_bytes = (size_t) g->ndata * 4;
		}
		memcpy(g->h_rows, rows, (size_t) M * 4);
	}
	int *d_rows = rows ? g->d_rows : nullptr;
	// Nothing is cleared: point labels and the rounds' "moved" flags carry the number of the batch
	// of rounds they belong to, the component counter is put back to zero by the kernel that reads it.
	if (g->call >= 0x7ffffff0) {                                     // stamps start over (once in 2^31 batches)
		if (!MDNS_HIP(hipMemsetAsync(plabel_of(g), 0, (size_t) g->cap_points * sizeof(PLabel), c->stream)) ||
		    !MDNS_HIP(hipMemsetAsync(hdr_of(g)->changed, 0, sizeof(hdr_of(g)->changed), c->stream))) return 1;
		g->call = 0;
	}
	const int call = ++g->call;
	const GroupsHeader *h = &g->h_box->header;
	long long total = 0;
	int batch = g->rounds_hint + 1, needed = 0;                      // one spare round costs 5 us, a second look 40
	int flag = call;                                                 // what a round of the current batch writes into its flag
	while (true) {
		if (total > 0) flag = ++g->call;                              // (labels keep the stamp `call`; only the flags move on)
		for (int r = 0; r < batch; r++)
			hipLaunchKernelGGL(k_groups_round, dim3((M + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, c->stream,
			                   g->d_idsT, g->nlive, d_rows, rows && total == 0 && r == 0 ? (const int *) g->h_rows_dev : nullptr, M, npoints,
			                   plabel_of(g), g->d_label, total == 0 && r == 0 ? 1 : 0, call,
			                   flag, &hdr_of(g)->changed[r], &hdr_of(g)->status, kSweeps);
		total += batch;
		// optimistically everything that follows a converged state, in the same round trip
		const int finish_blocks = (M + kBlock - 1) / kBlock;
		hipLaunchKernelGGL(k_groups_finish, dim3(finish_blocks + (unsigned) ((npoints + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
		                   g->d_label, (const int *) d_rows, M, g->d_labels, hdr_of(g), finish_blocks, plabel_of(g), npoints, call, touched_of(g));
		hipLaunchKernelGGL(k_groups_compact, dim3(1), dim3(1024), 0, c->stream, touched_of(g), (long long) nw, pout_of(g), hdr_of(g),
		                   g->h_box_dev, ++g->box_seq);
		if (!MDNS_HIP(hipGetLastError())) return 1;
		// counts, flags and the list arrive in mapped host memory: poll for `seq` (looking at the
		// stream now and then: a failed launch shows up as an error instead of a hang)
		volatile unsigned long long *seq = &g->h_box->seq;
		long long started = 0;
		for (unsigned spin = 0; *seq != g->box_seq; spin++) {
			if ((spin & 1023) != 1023) continue;
			const hipError_t e = hipStreamQuery(c->stream);
			if (e == hipErrorNotReady) {
				if (poll_expired(&started)) { set_error("mdns_groups_components: no result within MDNS_POLL_TIMEOUT_S"); return 1; }
				continue;
			}
			if (e != hipSuccess) { set_error("mdns_groups_components: %s", hipGetErrorString(e)); return 1; }
			if (*seq != g->box_seq) { set_error("mdns_groups_components: finished without a result"); return 1; }
		}
		std::atomic_thread_fence(std::memory_order_acquire);
		if (h->status) {
			const int status = h->status;
			(void) hipMemsetAsync(&hdr_of(g)->status, 0, sizeof(int), c->stream);
			set_error(status & 2 ? "mdns_groups_components: an earlier mdns_groups_replace named a data set or slot that does not exist"
			                     : "mdns_groups_components: an id outside [0, %lld) was met", npoints);
			return 1;
		}
		if (h->changed[batch - 1] != flag) {                         // the last round found nothing to do
			int clean = batch - 1;
			while (clean > 0 && h->changed[clean - 1] != flag) clean--;
			needed = (int) (total - batch) + clean + 1;              // the deciding round is part of it
			break;
		}
		if (total >= kRoundLimit) { set_error("mdns_groups_components: labels still moving after %lld rounds", total); return 1; }
		// not there yet: a few more rounds, then more and more (a slow graph: look less often)
		batch = total <= g->rounds_hint + 1 ? 2 : (batch * 2 > kMaxRounds ? kMaxRounds : batch * 2);
	}
	g->rounds_hint = needed < 2 ? 2 : (needed > 16 ? 16 : needed);
	g->rounds_total += needed; g->calls_total++;
	*ncomponents = h->ncomponents;
	const long long nd = h->ndistinct;
	if (ndistinct) *ndistinct = nd;
	if (distinct) {
		if (nd > cap) { set_error("mdns_groups_components: %lld distinct ids, room for %lld", nd, cap); return 1; }
		memcpy(distinct, (const void *) g->h_box->distinct, (size_t) nd * 4);
	}
	if (touched) {
		if (!MDNS_HIP(hipMemcpyAsync(touched, touched_of(g), nw * 8, hipMemcpyDeviceToHost, c->stream)) ||
		    !MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
	}
	g->last_M = M; g->last_npoints = npoints; g->last_ndistinct = nd;
	g->label_call = call;
	return 0;
}

/* rounds the label propagation needed per call so far (the clean round included) */
extern "C" double mdns_groups_mean_rounds(const mdns_groups *g)
{
	return g && g->calls_total ? (double) g->rounds_total / (double) g->calls_total : 0.0;
}

extern "C" int mdns_groups_id_labels(mdns_groups *g, int32_t *labels, int32_t *id_labels, long long ndistinct)
{
	Context *c = ctx();
	if (!c || !g) return 1;
	if (g->last_M <= 0) { set_error("mdns_groups_id_labels: no components computed since the ids last changed"); return 1; }
	if (ndistinct != g->last_ndistinct) { set_error("mdns_groups_id_labels: %lld ids asked for, the last call listed %lld", ndistinct, g->last_ndistinct); return 1; }
	const int M = g->last_M;
	// the list of the last call is still in the compaction's scratch; every entry is replaced by its
	// label in place (a thread reads its own entry, then writes it), so this works once per call
	int *list = pout_of(g), *out = pout_of(g);
	if (id_labels && ndistinct > 0) {
		g->last_ndistinct = -1;
		hipLaunchKernelGGL(k_groups_id_labels, dim3((unsigned) ((ndistinct + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
		                   plabel_of(g), (const int *) list, ndistinct, g->label_call, out);
		if (!MDNS_HIP(hipGetLastError())) return 1;
	}
	if (labels && !MDNS_HIP(hipMemcpyAsync(labels, g->d_labels, (size_t) M * 4, hipMemcpyDeviceToHost, c->stream))) return 1;
	if (id_labels && ndistinct > 0 && !MDNS_HIP(hipMemcpyAsync(id_labels, out, (size_t) ndistinct * 4, hipMemcpyDeviceToHost, c->stream))) return 1;
	return MDNS_HIP(hipStreamSynchronize(c->stream)) ? 0 : 1;
}

extern "C" int mdns_groups_labels(mdns_groups *g, int32_t *labels, int32_t *point_labels)
{
	Context *c = ctx();
	if (!c || !g) return 1;
	if (g->last_M <= 0) { set_error("mdns_groups_labels: no components computed since the ids last changed"); return 1; }
	const int M = g->last_M;
	const long long np = g->last_npoints;
	if (point_labels) g->last_ndistinct = -1;                          // (the labels overwrite the list of the last call)
	if (point_labels)
		hipLaunchKernelGGL(k_groups_point_labels, dim3((unsigned) ((np + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
		                   plabel_of(g), np, g->label_call, pout_of(g));
	if (!MDNS_HIP(hipGetLastError())) return 1;
	if (labels && !MDNS_HIP(hipMemcpyAsync(labels, g->d_labels, (size_t) M * 4, hipMemcpyDeviceToHost, c->stream))) return 1;
	if (point_labels && !MDNS_HIP(hipMemcpyAsync(point_labels, pout_of(g), (size_t) np * 4, hipMemcpyDeviceToHost, c->stream))) return 1;
	return MDNS_HIP(hipStreamSynchronize(c->stream)) ? 0 : 1;
}
