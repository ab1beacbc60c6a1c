// One chunk of a constrained draw in TWO launches, for the launches a real run is made of: a few
// dozen candidates against a few hundred to a few thousand selected spectra
// (hiermetriclearn.py:181-196 + multi_nested_sampler.py:462-485).
//
//   k_chunk_accept   candidates' parameters and the selection's row ids are read straight from host
//                    memory mapped into the device (no copy command in front of the kernel); every
//                    workgroup computes the templates of ITS candidate tile into LDS itself
//                    (clike.c:65) and scores them against ITS 256 selected spectra, read from the
//                    [n_datasets x n_channels] rows (one cache line per lane and stage: a sparse
//                    selection costs exactly its own bytes, no replica is built); the epilogue is the
//                    accept test (`any(L > Lmins)`) -- one stamped flag per accepted candidate and the
//                    trail of its likelihoods leave the kernel, nothing else
//   k_chunk_commit   ONE workgroup: first flagged candidate, shelf appends, next thresholds
//                    (multi_nested_sampler.py:482-485,438-447), fill bits, and the outcome written
//                    to the mapped mailbox the host polls (no copy, no stream synchronisation)
//
// Per (candidate, spectrum) the sum is the same chain as in the big lane kernel (mdns_like.hip):
// channels in ascending order, d = m - y, acc = fma(d, d, acc), padding channels contributing
// fma(0, 0, acc) -- so a likelihood does not depend on which kernel computed it.
#include "mdns_internal.h"
#include <cstdlib>

namespace mdns {

static constexpr int kCH = 8;              // channels per stage

typedef JointMailbox ChunkMailbox;          // what the host finds in mapped memory after a chunk

// quad broadcast: every lane of a quad gets the value lane Q of the quad holds (DPP quad_perm)
template <int Q>
__device__ __forceinline__ double quad_bcast(double v)
{
	constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}

// Work of one workgroup: 64 selected spectra (one tile of the selection) x 4 candidates (one
// candidate tile).  A quad of lanes shares ONE spectrum: lane q of the quad LOADS the q-th quarter
// of every 64-byte stage of the row (so four adjacent lanes read one cache line and a wave's load
// touches 16 lines -- with one lane per spectrum it would be 64 lines four times over, and the
// texture-address unit, one line per clock, was the bound: 33 us measured), and SCORES candidate q
// of the tile against the whole spectrum, the other three quarters of a stage arriving by quad
// broadcasts (DPP moves inside the VALU).  The chain of a (candidate, spectrum) pair is unchanged:
// one accumulator, channels ascending, d = m - y, acc = fma(d, d, acc).
// NST = stages held in registers: the whole row is requested before the first sum starts.
template <int NST>
__global__ __launch_bounds__(256) void k_chunk_accept(
    const double *__restrict__ Y, int ld, int nx, int nxp, const double *__restrict__ xgrid,
    const double *__restrict__ params, int B, double scale,
    const int *__restrict__ rows, int *__restrict__ rows_dev, int M, int ntiles,
    const double *__restrict__ higher, int *__restrict__ flags, int stamp, JointTrail trail, JointHeader *__restrict__ header)
{
	extern __shared__ __attribute__((aligned(16))) double lds[];
	if (blockIdx.x == 0 && threadIdx.x == 0) header->status = 0;       // the commit pass may raise it
	double2 *tpl = reinterpret_cast<double2 *>(lds);            // [nxp / 2][4 candidates] pairs of channels
	double *par = lds + (size_t) nxp * 4;                       // [4][3]
	unsigned long long *votes = reinterpret_cast<unsigned long long *>(par + 12);   // [4 waves]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int q = lane & 3;                                     // quarter loaded, candidate scored
	const int bt = blockIdx.x / ntiles, tile = blockIdx.x - bt * ntiles;
	// 1. the candidates of this tile (host memory: issued first, the latency is the PCIe round trip)
	double pv = 0.0;
	if (threadIdx.x < 12) {
		const int b = bt * 4 + threadIdx.x / 3;
		pv = b < B ? params[(size_t) b * 3 + threadIdx.x % 3] : 0.0;
	}
	// 2. this quad's spectrum: all of its row requested at once
	const int r = wave * 16 + (lane >> 2);                     // position in the tile
	const int k = tile * 64 + r;
	const bool live = k < M;
	const int kk = live ? k : M - 1;
	const int row = rows ? rows[kk] : kk;
	if (rows_dev && bt == 0 && live && q == 0) rows_dev[k] = row;      // for the commit kernel
	const double *yr = Y + (size_t) row * ld;
	const int nst = nxp / kCH;
	double2 y[NST];
#pragma unroll
	for (int s = 0; s < NST; s++) {
		const int j = s * kCH + 2 * q;
		const double2 v = *reinterpret_cast<const double2 *>(yr + (j < ld ? j : 0));
		// channels at or beyond the row's length are padding: zeros
		y[s].x = j < ld ? v.x : 0.0;
		y[s].y = j < ld ? v.y : 0.0;
	}
	const double thr = live ? higher[row] : __builtin_nan("");          // NaN compares false: no vote
	// 3. templates of the candidate tile, computed here (clike.c:65: A exp(-0.5 ((mu - x)/sig)^2))
	if (threadIdx.x < 12) par[threadIdx.x] = pv;
	__syncthreads();
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
		if (s < nst) {                                              // wave-uniform
			const double2 *m = tpl + (size_t) s * 16 + q;           // 4 channel pairs x 4 candidates per stage
			double d;
#define QUARTER(QQ) { const double2 mv = m[QQ * 4]; \
			d = mv.x - quad_bcast<QQ>(y[s].x); acc = fma(d, d, acc); \
			d = mv.y - quad_bcast<QQ>(y[s].y); acc = fma(d, d, acc); }
			QUARTER(0) QUARTER(1) QUARTER(2) QUARTER(3)
#undef QUARTER
		}
	}
	// 5. accept test: lane (spectrum r, candidate q)
	const double L = acc * scale;
	const bool beat = L > thr && bt * 4 + q < B;
	const unsigned long long vote = __ballot(beat);                 // bit 4 i + q: spectrum wave * 16 + i, candidate q
	if (lane == 0) votes[wave] = vote;
	if (beat) trail.L[((size_t) (bt * 4 + q) * ntiles + tile) * 64 + r] = L;
	__syncthreads();
	// wave w' puts together the word of candidate w': lane l = spectrum l of the tile
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

// The second half in one workgroup of 1024 threads (16 waves, each walking tiles wave, wave + 16,
// ...): nothing crosses a workgroup, so the outcome can go to the host from here.
__global__ __launch_bounds__(1024) void k_chunk_commit(
    const int *__restrict__ thr_rows, int M, int B, int ntiles, const int *__restrict__ flags, int stamp,
    JointTrail trail, JointArrays st, JointHeader *__restrict__ header, unsigned long long *__restrict__ fillbits,
    ChunkMailbox *__restrict__ box, unsigned long long seq)
{
	__shared__ int s_first, s_status;
	if (threadIdx.x == 0) { s_first = 0x7fffffff; s_status = 0; }
	__syncthreads();
	for (int b = threadIdx.x; b < B; b += 1024)
		if (flags[b] == stamp) { atomicMin(&s_first, b); break; }
	__syncthreads();
	const int bstar = s_first;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (bstar < B) {
		for (int tile = wave; tile < ntiles; tile += 16) {
			const size_t at = (size_t) bstar * ntiles + tile;
			const unsigned long long word = trail.stamp_of[at] == trail.stamp ? trail.word[at] : 0ull;
			const int k = tile * 64 + lane;
			if (k < M && (word >> lane & 1ull)) {
				const int d = thr_rows ? thr_rows[k] : k;
				const double L = trail.L[at * 64 + lane];
				const double thr = st.higher[d];
				const int n = st.shelfn[d];
				if (n >= st.cap) {
					atomicOr(&s_status, 1);
				} else {
					// (see k_joint_commit_trail: the (n+2)-th smallest of the enlarged set)
					int at_most = 0;
					double next = INFINITY;
					int p = 0;
					for (; p + 16 <= st.nlive; p += 16) {                 // sixteen loads in flight
						double v[16];
#pragma unroll
						for (int u = 0; u < 16; u++) v[u] = st.live[(size_t) (p + u) * st.ndata + d];
#pragma unroll
						for (int u = 0; u < 16; u++) { if (v[u] <= thr) at_most++; else next = fmin(next, v[u]); }
					}
					for (; p < st.nlive; p++) {
						const double v = st.live[(size_t) p * st.ndata + d];
						if (v <= thr) at_most++; else next = fmin(next, v);
					}
					for (int e = 0; e < n; e++) {
						const double v = st.shelfL[(size_t) e * st.ndata + d];
						if (v <= thr) at_most++; else next = fmin(next, v);
					}
					st.shelfL[(size_t) n * st.ndata + d] = L;
					st.shelfn[d] = n + 1;
					st.higher[d] = at_most >= n + 2 ? thr : fmin(L, next);
				}
			}
			if (lane == 0) { fillbits[tile] = word; box->bits[tile] = word; }
		}
	}
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x != 0) return;
	const int accepted = bstar < B ? bstar : -1;
	header->accepted = accepted;
	header->status = s_status;
	box->accepted = accepted;
	box->status = s_status;
	__hip_atomic_store(&box->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static bool launched(const char *name)
{
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) return true;
	set_error("launch of %s failed: %s", name, hipGetErrorString(e));
	return false;
}

// whether a chunk of B candidates over M selected spectra takes the two-launch path
bool chunk_fits(const mdns_spectra *s, int M, int B)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	const long long groups = (long long) ntiles * ((B + 3) / 4);
	static const char *limit = getenv("MDNS_CHUNK_GROUPS");            // experiments: most workgroups of the accept kernel
	const long long most = limit && atoll(limit) > 0 ? atoll(limit) : 16LL * c->num_cus;
	// Measured (tools/chunk_bench.py, us per chunk, this path / the five-command one): 300 spectra x 32
	// candidates 24.7 / 39.5, 4096 x 32: 26.9 / 37.2, 3000 x 128: 44.6 / 47.6, 10 000 x 4: 22.3 / 31.6 -- but
	// 10 000 x 32: 43.2 / 36.9 and 10 000 x 128: 49.5 / 44.8: with every spectrum selected and several
	// candidate tiles the lane kernel on the channel-major replica reads perfectly coalesced.
	if (M > 4096 && B > 8) return false;
	return groups <= most && cols_nx(s->nx) <= 8 * 32 && s->d_x != nullptr;
}

bool launch_chunk_accept(const mdns_spectra *s, const double *d_params_mapped, int B, double scale,
                         const int *d_rows_in, int *d_rows_dev, int M, const double *d_higher,
                         int *d_flags, int stamp, const JointTrail &trail, void *d_header)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	const int nbt = (B + 3) / 4;
	const int nxp = cols_nx(s->nx);
	const size_t lds = ((size_t) nxp * 4 + 12 + 4) * sizeof(double);
	const int nst = nxp / kCH;
	ProfileScope prof(0);
#define CHUNK_LAUNCH(NST) hipLaunchKernelGGL((k_chunk_accept<NST>), dim3(ntiles * nbt), dim3(256), lds, c->stream, \
	s->d_y, s->ld, s->nx, nxp, s->d_x, d_params_mapped, B, scale, d_rows_in, d_rows_dev, M, ntiles, d_higher, d_flags, stamp, trail, (JointHeader *) d_header)
	if (nst <= 8) { note_kernel(0, "k_chunk_accept<8>"); CHUNK_LAUNCH(8); }
	else if (nst <= 16) { note_kernel(0, "k_chunk_accept<16>"); CHUNK_LAUNCH(16); }
	else if (nst <= 26) { note_kernel(0, "k_chunk_accept<26>"); CHUNK_LAUNCH(26); }
	else { note_kernel(0, "k_chunk_accept<32>"); CHUNK_LAUNCH(32); }
#undef CHUNK_LAUNCH
	return launched("k_chunk_accept");
}

bool launch_chunk_commit(const int *d_thr_rows, int M, int B, const int *d_flags, int stamp, const JointTrail &trail,
                         const JointArrays &st, void *d_header, unsigned long long *d_fillbits, void *box_dev,
                         unsigned long long seq)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	hipLaunchKernelGGL(k_chunk_commit, dim3(1), dim3(1024), 0, c->stream, d_thr_rows, M, B, ntiles, d_flags, stamp, trail, st,
	                   (JointHeader *) d_header, d_fillbits, (ChunkMailbox *) box_dev, seq);
	return launched("k_chunk_commit");
}

}  // namespace mdns
