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

namespace mdns {

static constexpr int kCH = 8;              // channels per stage

typedef JointMailbox ChunkMailbox;          // what the host finds in mapped memory after a chunk

template <int BT, int NB>
__global__ __launch_bounds__(256) void k_chunk_accept(
    const double *__restrict__ Y, int ld, int nx, int nxp, const double *__restrict__ xgrid,
    const double *__restrict__ params, int B, double scale,
    const int *__restrict__ rows, int *__restrict__ rows_dev, int M, int ntiles,
    const double *__restrict__ higher, int *__restrict__ flags, int stamp, JointTrail trail)
{
	extern __shared__ __attribute__((aligned(16))) double lds[];
	double *tpl = lds;                                  // [nxp][BT]
	double *par = lds + (size_t) nxp * BT;              // [BT][3]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int nquads = (ntiles + 3) >> 2;
	const int bt = blockIdx.x / nquads, quad = blockIdx.x - bt * nquads;
	const int tile = quad * 4 + wave;
	// 1. the candidates of this tile (host memory: issue first, the latency is the PCIe round trip)
	double pv = 0.0;
	if (threadIdx.x < BT * 3) {
		const int b = bt * BT + threadIdx.x / 3;
		pv = b < B ? params[(size_t) b * 3 + threadIdx.x % 3] : 0.0;
	}
	// 2. this lane's spectrum
	const int k = tile * 64 + lane;
	const bool live = tile < ntiles && k < M;
	int row = 0;
	if (tile < ntiles) {
		const int kk = k < M ? k : M - 1;
		row = rows ? rows[kk] : kk;
		if (rows_dev && bt == 0 && k < M) rows_dev[k] = row;          // for the commit kernel
	}
	const double *yr = Y + (size_t) row * ld;
	const int nst = nxp / kCH;
	double y[NB][kCH];
	// channels at or beyond the row's length are padding: read a valid pair, use zeros
	auto load_stage = [&](double (&dst)[kCH], int s) {
#pragma unroll
		for (int c = 0; c < kCH; c += 2) {
			const int j = s * kCH + c;
			const int jj = j < ld ? j : 0;
			const double2 v = *reinterpret_cast<const double2 *>(yr + jj);
			dst[c] = j < ld ? v.x : 0.0;
			dst[c + 1] = j < ld ? v.y : 0.0;
		}
	};
	if (tile < ntiles) {
#pragma unroll
		for (int i = 0; i < NB - 1; i++) load_stage(y[i], i < nst ? i : nst - 1);
	}
	const double thr = live ? higher[row] : __builtin_nan("");          // NaN compares false: no vote
	// 3. templates of the candidate tile, computed here (clike.c:65: A exp(-0.5 ((mu - x)/sig)^2))
	if (threadIdx.x < BT * 3) par[threadIdx.x] = pv;
	__syncthreads();
	for (int e = threadIdx.x; e < nxp * BT; e += 256) {
		const int j = e / BT, bb = e - j * BT;
		double m = 0.0;
		if (j < nx && bt * BT + bb < B) {
			const double A = par[bb * 3], mu = par[bb * 3 + 1], sig = par[bb * 3 + 2];
			const double t = (mu - xgrid[j]) / sig;
			m = A * exp(-0.5 * (t * t));
		}
		tpl[e] = m;
	}
	__syncthreads();
	if (tile >= ntiles) return;
	// 4. the sums, NB - 1 stages of spectra in flight
	double acc[BT];
#pragma unroll
	for (int b = 0; b < BT; b++) acc[b] = 0.0;
#pragma unroll 1
	for (int s0 = 0; s0 < nst; s0 += NB) {
#pragma unroll
		for (int i = 0; i < NB; i++) {
			const int s = s0 + i;                                   // wave-uniform
			if (s < nst) {
				const int ahead = s + NB - 1;
				load_stage(y[(i + NB - 1) % NB], ahead < nst ? ahead : nst - 1);
				const double *m = tpl + s * kCH * BT;
#pragma unroll
				for (int c = 0; c < kCH; c++)
#pragma unroll
					for (int b = 0; b < BT; b++) {
						const double d = m[c * BT + b] - y[i][c];
						acc[b] = fma(d, d, acc[b]);
					}
			}
		}
	}
	// 5. accept test
#pragma unroll
	for (int b = 0; b < BT; b++) {
		const double L = acc[b] * scale;
		const unsigned long long word = __ballot(L > thr);
		if (word != 0ull && bt * BT + b < B) {
			const size_t at = (size_t) (bt * BT + b) * ntiles + tile;
			trail.L[at * 64 + lane] = L;
			if (lane == 0) {
				flags[bt * BT + b] = stamp;
				trail.word[at] = word;
				trail.stamp_of[at] = trail.stamp;
			}
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

// candidates per workgroup of the small-chunk accept kernel
int chunk_tile(int M, int B)
{
	(void) M;
	return B >= 4 ? 4 : (B >= 2 ? 2 : 1);
}

// whether a chunk of B candidates over M selected spectra takes the two-launch path
bool chunk_fits(const mdns_spectra *s, int M, int B)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	const int bt = chunk_tile(M, B);
	const long long waves = (long long) ntiles * ((B + bt - 1) / bt);
	const size_t lds = ((size_t) cols_nx(s->nx) * bt + 3 * bt) * sizeof(double);
	return ntiles <= 64 && waves <= 16LL * c->num_cus && lds <= 60 * 1024 && s->d_x != nullptr;
}

bool launch_chunk_accept(const mdns_spectra *s, const double *d_params_mapped, int B, double scale,
                         const int *d_rows_mapped, int *d_rows_dev, int M, const double *d_higher,
                         int *d_flags, int stamp, const JointTrail &trail)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64, nquads = (ntiles + 3) / 4;
	const int bt = chunk_tile(M, B);
	const int nbt = (B + bt - 1) / bt;
	const int nxp = cols_nx(s->nx);
	const size_t lds = ((size_t) nxp * bt + 3 * bt) * sizeof(double);
	// stages in flight: all of a 200-channel spectrum in two rounds
	const int nst = nxp / kCH;
	ProfileScope prof(0);
#define CHUNK_LAUNCH(BT, NB) hipLaunchKernelGGL((k_chunk_accept<BT, NB>), dim3(nquads * nbt), dim3(256), lds, c->stream, \
	s->d_y, s->ld, s->nx, nxp, s->d_x, d_params_mapped, B, scale, d_rows_mapped, d_rows_dev, M, ntiles, d_higher, d_flags, stamp, trail)
	if (nst > 8) {
		note_kernel(0, "k_chunk_accept<%d, 13>", bt);
		switch (bt) { case 4: CHUNK_LAUNCH(4, 13); break; case 2: CHUNK_LAUNCH(2, 13); break; default: CHUNK_LAUNCH(1, 13); break; }
	} else {
		note_kernel(0, "k_chunk_accept<%d, 4>", bt);
		switch (bt) { case 4: CHUNK_LAUNCH(4, 4); break; case 2: CHUNK_LAUNCH(2, 4); break; default: CHUNK_LAUNCH(1, 4); break; }
	}
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
