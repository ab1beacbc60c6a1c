// The first batch of a fresh RadFriends region WITHOUT a host look in between: behind the radius
// computation (K6, mdns_neighbors.hip) the stream carries
//
//   k_box_count      the box proposals  lo + (hi - lo) u  (radfriendsregion.py:135; lo / hi = the members'
//                    extent widened by the radius K6 has just left in device memory, :69-70) from the raw
//                    doubles the host drew while K6 ran, and their membership counts (K3,
//                    cneighbors.c:95-119) -- to the host's mapped block and to device memory
//   k_chain_accept   k_chunk_accept (mdns_chunk.hip) whose candidates are the first min(kept, limit)
//                    proposals that lie in the region AND, after the metric's inverse transform, in the
//                    unit cube (hiermetriclearn.py:111-119), in proposal order; prior transform and
//                    kernel parameters (sample.py:52-58,103) computed here; accept test as epilogue
//   commit           the commit kernels of any other chunk (k_chunk_commit / k_joint_commit_trail)
//
// so that the host, which so far waited for the radius, then for the counts, then for the chunk,
// polls ONE mailbox.  Everything the host needs to carry on as if it had done the steps itself comes
// back: the radius (the region's own result slot), the counts, the number of kept proposals, the
// chunk size used, the accepted candidate and its fill bits, and the parameters the device scored
// with (the host compares the accepted candidate's with its own).
//
// Arithmetic: the proposals and the inverse transform are single IEEE operations in the host's order
// (this file is compiled with -ffp-contract=off), the counts use the squared-distance test of
// mdns_neighbors.hip; 10**v is mdns_pow10.h (correctly rounded but for one argument in ~10^4; the C
// library's own pow is off by an ulp more often).  The sum of a (candidate, spectrum) pair is the chain
// of every other K1 form: channels ascending, d = m - y, acc = fma(d, d, acc).
#include "mdns_internal.h"
#include "mdns_pow10.h"

#pragma clang fp contract(off)

namespace mdns {

static constexpr int kCH = 8;              // channels per stage (as mdns_chunk.hip)

template <int Q>
__device__ __forceinline__ double quad_bcast(double v)
{
	constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}

// proposal i, dimension k: lo + (hi - lo) u with lo = mn - r, hi = mx + r (the host's operations, one
// rounding each: host_constrainer.cpp, Region::box and the BOX phase)
__device__ __forceinline__ double box_proposal(double mn, double mx, double r, double u)
{
	const double lo = mn - r, hi = mx + r;
	const double range = hi - lo;
	const double t = range * u;
	return lo + t;
}

// ---------------------------------------------------------------------------------------
// proposals + membership counts: 4 proposals x 64 member slices per workgroup (the "fine" shape of
// k_count_within: ~1000 proposals still make 250 workgroups), members tiled through LDS
// ---------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_box_count(
    const double *__restrict__ members, int K, const RegionResult *__restrict__ res, ChainSpec spec,
    ChainBox *__restrict__ box, double *__restrict__ props, int *__restrict__ counts, int tile_n, CountMail mail)
{
	constexpr int PTS = 4, NSLICE = 64;
	extern __shared__ double smem[];
	double *tile = smem;                                                  // [tile_n][D]
	int *part = reinterpret_cast<int *>(smem + (size_t) tile_n * D);      // [4][PTS]
	const double radius = res->radius, thresh_sq = res->thresh;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int pt = lane % PTS;
	const int slice = wv * 16 + lane / PTS;
	const int j = blockIdx.x * PTS + pt;
	const int jj = j < spec.n ? j : spec.n - 1;
	double c[D];
#pragma unroll
	for (int k = 0; k < D; k++) c[k] = box_proposal(spec.mn[k], spec.mx[k], radius, box->u[(size_t) jj * D + k]);
	if (threadIdx.x < PTS && j < spec.n) {
#pragma unroll
		for (int k = 0; k < D; k++) props[(size_t) j * D + k] = c[k];
	}
	int hits = 0;
	for (int t0 = 0; t0 < K; t0 += tile_n) {
		const int n = min(tile_n, K - t0);
		__syncthreads();
		for (int e = threadIdx.x; e < n * D; e += 256) tile[e] = members[(size_t) t0 * D + e];
		__syncthreads();
#pragma unroll 4
		for (int i = slice; i < n; i += NSLICE) {
			double acc = 0.0;
#pragma unroll
			for (int k = 0; k < D; k++) {
				const double diff = tile[i * D + k] - c[k];
				acc = acc + diff * diff;
			}
			hits += acc < thresh_sq ? 1 : 0;
		}
	}
#pragma unroll
	for (int off = PTS; off < 64; off <<= 1) hits += __shfl_xor(hits, off, 64);
	if (lane < PTS) part[wv * PTS + pt] = hits;
	__syncthreads();
	if (wv == 0 && lane < PTS && j < spec.n) {
		const int total = (part[pt] + part[PTS + pt]) + (part[2 * PTS + pt] + part[3 * PTS + pt]);
		counts[j] = total;
		__hip_atomic_store(&box->counts[j], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (mail.seq_at) {
		if (wv == 0) handover_release();                           // (mdns_internal.h)
		__syncthreads();
		if (threadIdx.x == 0) {
			const int done = atomicAdd(mail.ticket, 1);
			if (done == (int) gridDim.x - 1) {
				__hip_atomic_store(mail.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				mail_raise(mail.seq_at, mail.seq);
			}
		}
	}
}

// ---------------------------------------------------------------------------------------
// the accept pass of the first chunk: candidates = kept proposals
// ---------------------------------------------------------------------------------------
// Every workgroup finds ITS four candidates itself: a proposal is kept when a member lies within
// the radius (count > 0) and its inverse-transformed coordinates lie strictly inside the unit cube;
// candidate number = rank among the kept ones, in proposal order; B = min(kept, limit).
template <int NST>
__global__ __launch_bounds__(256) void k_chain_accept(
    const double *__restrict__ Y, int ld, int nx, int nxp, const double *__restrict__ xgrid,
    ChainSpec spec, const double *__restrict__ props, const int *__restrict__ counts, ChainBox *__restrict__ box, double scale,
    const int *__restrict__ rows, int *__restrict__ rows_dev, int M, int ntiles,
    const double *__restrict__ higher, int *__restrict__ flags, int stamp, JointTrail trail, JointHeader *__restrict__ header)
{
	extern __shared__ __attribute__((aligned(16))) double lds[];
	if (blockIdx.x == 0 && threadIdx.x == 0) header->status = 0;
	double2 *tpl = reinterpret_cast<double2 *>(lds);            // [nxp / 2][4 candidates] pairs of channels
	double *par = lds + (size_t) nxp * 4;                       // [4][3]
	unsigned long long *votes = reinterpret_cast<unsigned long long *>(par + 12);   // [4 waves]
	int *wave_kept = reinterpret_cast<int *>(votes + 4);        // [4]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int q = lane & 3;
	const int bt = blockIdx.x / ntiles, tile = blockIdx.x - bt * ntiles;
	// 1. this quad's spectrum: all of its row requested at once (in flight during the prologue)
	const int r = wave * 16 + (lane >> 2);
	const int k = tile * 64 + r;
	const bool live = k < M;
	const int kk = live ? k : M - 1;
	const int row = rows ? rows[kk] : kk;
	if (rows_dev && bt == 0 && live && q == 0) rows_dev[k] = row;
	const double *yr = Y + (size_t) row * ld;
	const int nst = nxp / kCH;
	double2 y[NST];
#pragma unroll
	for (int s = 0; s < NST; s++) {
		const int j = s * kCH + 2 * q;
		const double2 v = *reinterpret_cast<const double2 *>(yr + (j < ld ? j : 0));
		y[s].x = j < ld ? v.x : 0.0;
		y[s].y = j < ld ? v.y : 0.0;
	}
	const double thr = live ? higher[row] : __builtin_nan("");
	// 2. which proposals are candidates, and this tile's four
	if (threadIdx.x < 12) par[threadIdx.x] = 0.0;
	int base = 0;
	const int D = spec.ndim;
	for (int i0 = 0; i0 < spec.n; i0 += 256) {
		const int i = i0 + (int) threadIdx.x;
		bool kept = false;
		double x[kChainDim];
#pragma unroll
		for (int d = 0; d < kChainDim; d++) x[d] = 0.5;
		if (i < spec.n && counts[i] > 0) {
			kept = true;
#pragma unroll
			for (int d = 0; d < kChainDim; d++) {
				if (d < D) {
					const double yv = props[(size_t) i * D + d];
					double v = yv;
					if (!spec.identity) {
						const double p = yv * spec.scale[d];
						v = p + spec.mean[d];
					}
					x[d] = v;
					if (!(v < 1 && v > 0)) kept = false;
				}
			}
		}
		const unsigned long long m = __ballot(kept);
		if (lane == 0) wave_kept[wave] = __popcll(m);
		__syncthreads();
		int before = base;
		for (int w = 0; w < wave; w++) before += wave_kept[w];
		const int rank = before + __popcll(m & ((1ull << lane) - 1ull));
		if (kept && rank >= bt * 4 && rank < bt * 4 + 4 && rank < spec.limit) {
			// prior transform and kernel parameters (mdns_prior; host_constrainer.cpp transform())
			double p3[3] = {0.0, 0.0, 0.0};
#pragma unroll
			for (int d = 0; d < kChainDim; d++) {
				if (d < D) {
					double v = spec.a[d] * x[d];
					if (spec.b[d] != 0.0) v = v + spec.b[d];
					if (spec.pow10[d]) v = mdns_pow10::pow10_dd(v);
					if (d < 3) p3[d] = spec.kernel_pow10[d] ? mdns_pow10::pow10_dd(v) : v;
				}
			}
			const int slot = rank - bt * 4;
			par[slot * 3] = p3[0]; par[slot * 3 + 1] = p3[1]; par[slot * 3 + 2] = p3[2];
			if (tile == 0) { box->params[rank][0] = p3[0]; box->params[rank][1] = p3[1]; box->params[rank][2] = p3[2]; }
		}
		base += wave_kept[0] + wave_kept[1] + wave_kept[2] + wave_kept[3];
		__syncthreads();
	}
	const int B = base < spec.limit ? base : spec.limit;
	if (blockIdx.x == 0 && threadIdx.x == 0) { box->nkept = base; box->B = B; }
	if (bt * 4 >= B) return;                                    // (whole workgroups)
	// 3. templates of the candidate tile (clike.c:65)
	for (int e = threadIdx.x; e < nxp * 4; e += 256) {
		const int j = e >> 2, bb = e & 3;
		double m = 0.0;
		if (j < nx && bt * 4 + bb < B) {
			const double A = par[bb * 3], mu = par[bb * 3 + 1], sig = par[bb * 3 + 2];
			const double t = (mu - xgrid[j]) / sig;
			m = A * exp(-0.5 * (t * t));
		}
		lds[((size_t) (j >> 1) * 4 + bb) * 2 + (j & 1)] = m;
	}
	__syncthreads();
	// 4. the sum of (candidate q, this spectrum)
	double acc = 0.0;
#pragma unroll
	for (int s = 0; s < NST; s++) {
		if (s < nst) {
			const double2 *m = tpl + (size_t) s * 16 + q;
			double d;
#define QUARTER(QQ) { const double2 mv = m[QQ * 4]; \
			d = mv.x - quad_bcast<QQ>(y[s].x); acc = fma(d, d, acc); \
			d = mv.y - quad_bcast<QQ>(y[s].y); acc = fma(d, d, acc); }
			QUARTER(0) QUARTER(1) QUARTER(2) QUARTER(3)
#undef QUARTER
		}
	}
	// 5. accept test
	const double L = acc * scale;
	const bool beat = L > thr && bt * 4 + q < B;
	const unsigned long long vote = __ballot(beat);
	if (lane == 0) votes[wave] = vote;
	if (beat) trail.L[((size_t) (bt * 4 + q) * ntiles + tile) * 64 + r] = L;
	__syncthreads();
	{
		const int cand = wave;
		const unsigned long long word = __ballot((votes[lane >> 4] >> (4 * (lane & 15) + cand)) & 1ull);
		if (word != 0ull && lane == 0) {
			const size_t at = (size_t) (bt * 4 + cand) * ntiles + tile;
			flags[bt * 4 + cand] = stamp;
			trail.word[at] = word;
			trail.stamp_of[at] = trail.stamp;
		}
	}
}

static bool launched(const char *name)
{
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) return true;
	set_error("launch of %s failed: %s", name, hipGetErrorString(e));
	return false;
}

bool launch_box_count(const RegionView &rv, const ChainSpec &spec, ChainBox *box_dev, double *d_props, int *d_counts,
                      const CountMail *mail)
{
	Context *c = ctx();
	if (spec.n <= 0 || spec.n > kChainMost || spec.ndim != rv.ndim || rv.ndim < 1 || rv.ndim > 5) {
		set_error("chain: %d proposals in %d dimensions", spec.n, rv.ndim);
		return false;
	}
	const CountMail none = {nullptr, nullptr, 0};
	const CountMail post = mail ? *mail : none;
	int tile_n = 512;
	const size_t lds = (size_t) tile_n * rv.ndim * sizeof(double) + 4 * 4 * sizeof(int);
	const dim3 grid((spec.n + 3) / 4);
	ProfileScope prof(2);
	note_kernel(2, "k_box_count<%d>", rv.ndim);
#define BOX_LAUNCH(D) hipLaunchKernelGGL((k_box_count<D>), grid, dim3(256), lds, c->stream, rv.d_members, rv.K, rv.d_res, spec, \
	box_dev, d_props, d_counts, tile_n, post)
	switch (rv.ndim) { case 1: BOX_LAUNCH(1); break; case 2: BOX_LAUNCH(2); break; case 3: BOX_LAUNCH(3); break;
	                   case 4: BOX_LAUNCH(4); break; default: BOX_LAUNCH(5); break; }
#undef BOX_LAUNCH
	return launched("k_box_count");
}

bool launch_chain_accept(const mdns_spectra *s, const ChainSpec &spec, const double *d_props, const int *d_counts,
                         ChainBox *box_dev, double scale, const int *d_rows_in, int *d_rows_dev, int M,
                         const double *d_higher, int *d_flags, int stamp, const JointTrail &trail, void *d_header)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	const int nbt = (spec.limit + 3) / 4;
	const int nxp = cols_nx(s->nx);
	const size_t lds = ((size_t) nxp * 4 + 12 + 4 + 2) * sizeof(double);
	const int nst = nxp / kCH;
	ProfileScope prof(0);
#define CHAIN_LAUNCH(NST) hipLaunchKernelGGL((k_chain_accept<NST>), dim3(ntiles * nbt), dim3(256), lds, c->stream, \
	s->d_y, s->ld, s->nx, nxp, s->d_x, spec, d_props, d_counts, box_dev, scale, d_rows_in, d_rows_dev, M, ntiles, d_higher, d_flags, stamp, trail, \
	(JointHeader *) d_header)
	if (nst <= 8) { note_kernel(0, "k_chain_accept<8>"); CHAIN_LAUNCH(8); }
	else if (nst <= 16) { note_kernel(0, "k_chain_accept<16>"); CHAIN_LAUNCH(16); }
	else if (nst <= 26) { note_kernel(0, "k_chain_accept<26>"); CHAIN_LAUNCH(26); }
	else { note_kernel(0, "k_chain_accept<32>"); CHAIN_LAUNCH(32); }
#undef CHAIN_LAUNCH
	return launched("k_chain_accept");
}

}  // namespace mdns
