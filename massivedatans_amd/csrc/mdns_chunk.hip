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
			if (lane == 0) { fillbits[tile] = word; mail_store(&box->bits[tile], word); }
		}
	}
	handover_release();
	__syncthreads();
	if (threadIdx.x != 0) return;
	const int accepted = bstar < B ? bstar : -1;
	header->accepted = accepted;
	header->status = s_status;
	mail_store(&box->accepted, accepted);
	mail_store(&box->status, s_status);
	mail_raise(&box->seq, seq);
}

// ---------------------------------------------------------------------------------------
// The accept test of an issue-bound chunk (hundreds of candidates x thousands of spectra) as a
// GUARDED FILTER ON THE MATRIX CORES.
//
// sum_j (m_bj - y_ij)^2 = sum_j m_bj^2 - 2 sum_j m_bj y_ij + sum_j y_ij^2.  The first sum belongs to the
// candidate (`msq`, from the template kernel), the last one to the spectrum (`ysq`, computed at
// upload); the middle one is a matrix product [candidates x channels] . [channels x spectra] -- the
// one piece of this hot path that IS GEMM-shaped -- and runs as v_mfma_f64_16x16x4_f64: a wave owns
// 16 candidates x 64 spectra, both operands arrive as plain coalesced vector loads (the vector-FMA
// form of the same sum was bound by the delivery of its scalar template operands: SQ counters in
// profiles/), 1024 multiply-adds per instruction.
// The filtered value Lf = scale (msq - 2 S + ysq) is NOT the chain value L the library works with
// (the expanded form cancels), but  |Lf - L| <= E := |scale| (nx + 8) 2^-52 (msq + 2 |S| + ysq)  (forward
// error of three sums of nx terms in any order with fused multiply-adds, of their combination, and
// of the chain itself: (3 nx + 6) u against the (2 nx + 16) u of E, and 4 E is what is used).  So
// Lf > thr + 4 E  implies  L > thr  -- a CLEAR vote: the candidate is flagged, exactly the decision
// of the chain kernel -- and  Lf < thr - 4 E  implies  L <= thr.  A candidate with a pair in between
// (about one pair in 10^9) and no clear vote is AMBIGUOUS.  k_exact_list then lists the ambiguous
// candidates below the lowest clear one and that one itself, and scores the listed candidates
// with the chain (votes, and the trail of likelihoods the commit pass keeps).  Flags,
// accepted index and every likelihood that is KEPT are therefore the chain kernel's, bit for bit.
// ---------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double4_t gg_join(double2_t a, double2_t b) { return double4_t{a[0], a[1], b[0], b[1]}; }

// lane l holds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16], D[i = 4 v + l / 16][j = l % 16] in
// its v-th result (tools/probes/mfma_f64_probe.hip checks this on the hardware).
//
// A workgroup (4 waves) owns one tile of SPEC spectra and FOUR tiles of 16 candidates, one per wave.
// The spectra -- 16 MB at 10 000 x 200, against 0.4 MB of templates -- arrive from L2 once per
// workgroup: chunks of CHUNK channels x SPEC spectra (rows of the tiled replica) are
// staged through LDS, two chunks ahead, and every wave feeds its MFMAs from there; its own
// templates come straight from global memory (512 B per k-step, coalesced).  Workgroups of the same
// spectrum tile run on the same XCD (blockIdx % 8), so the tile also reaches that L2 only once.
template <int SPEC, int CHUNK, int DEPTH, int PROBE>
__global__ __launch_bounds__(256) void k_gauss_mfma_filter(
    const double *__restrict__ YT, int nxp, int nx, const double *__restrict__ model_t, const double *__restrict__ msq, int B,
    double scale, const int *__restrict__ thr_rows, int M, int ntiles, int nbt, int ngroups,
    const double *__restrict__ higher, const double *__restrict__ ysq, int *__restrict__ flags, int *__restrict__ ambiguous,
    int stamp, int *__restrict__ lowest)
{
	constexpr int NB = SPEC / 16;                                         // 16-spectrum blocks per wave: MFMAs per k-step
	constexpr int kRow2 = SPEC / 2;                                       // 16-byte pieces per channel row
	constexpr int kPieces = (CHUNK * kRow2 + 255) / 256;                  // per thread and chunk (the last one may be partial)
	constexpr bool kWhole = (CHUNK * kRow2) % 256 == 0;                   // ... or not
	constexpr int kRowsPerPass = 256 / kRow2;
	constexpr int kSteps = CHUNK / 4;
	__shared__ __attribute__((aligned(16))) double stage[DEPTH + 1][CHUNK * SPEC];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
	const int tile = (local / ngroups) * 8 + xcd, grp = local - (local / ngroups) * ngroups;       // tiles of SPEC spectra
	if (tile >= ntiles) return;                                           // whole workgroups
	const int bt = grp * 4 + wave;
	const bool working = bt < nbt;                                        // wave-uniform; an idle wave still stages
	const int kq = lane >> 4, jq = lane & 15;
	const double *ap = model_t + (size_t) (working ? bt : 0) * nxp * 16 + lane;     // A of k-step c: ap[c * 64]
	// this thread's pieces of a chunk: channel rows (threadIdx / kRow2) + i kRowsPerPass, 16 bytes at (threadIdx % kRow2)
	const int first = tile * SPEC;                                        // first spectrum
	const int prow = threadIdx.x / kRow2, pcol = threadIdx.x % kRow2;
	const double2 *src = reinterpret_cast<const double2 *>(YT + ((size_t) (first >> 6) * nxp << 6) + (first & 63)) + (size_t) prow * 32 + pcol;
	// DEPTH + 1 register sets and LDS stages, used in rotation (the chunk loop is unrolled by that many
	// so that every set is named at compile time -- a copy between sets would wait for the loads):
	// while chunk c is multiplied, chunk c + 1 is on its way into LDS and, at DEPTH 2, the loads of
	// chunk c + 2 are in flight.
	double2 piece[DEPTH + 1][kPieces];
	double a[DEPTH + 1][kSteps];
	double4_t acc[NB];
#pragma unroll
	for (int t = 0; t < NB; t++) acc[t] = double4_t{0, 0, 0, 0};
	const int nchunks = (nxp + CHUNK - 1) / CHUNK;
	const int nk = nxp / 4;
	// (loads are unconditional and nothing looks at their results before the stage that needs them
	// -- a chunk number past the end fetches the last chunk again, a channel past the end the last
	// row, which is written to LDS as zeros, and a k-step past the end the last templates, which
	// then meet those zeros -- so that the compiler can COUNT the loads in flight: behind a branch, or
	// with a select on the loaded value, it waits for all of them, vmcnt(0), and the second chunk in
	// flight buys nothing)
#define FILTER_FETCH(SET, CHUNK_NO) { \
	const int cn_ = (CHUNK_NO) < nchunks ? (CHUNK_NO) : nchunks - 1; \
	_Pragma("unroll") for (int i = 0; i < kPieces; i++) { \
		const int row = kWhole || prow + i * kRowsPerPass < CHUNK ? prow + i * kRowsPerPass : CHUNK - 1; \
		const int ch = cn_ * CHUNK + row; \
		piece[SET][i] = src[(size_t) ((ch < nxp ? ch : nxp - 1) - prow) * 32]; \
	} \
	_Pragma("unroll") for (int i = 0; i < kSteps; i++) { \
		const int k = cn_ * kSteps + i; \
		a[SET][i] = ap[(size_t) (k < nk ? k : nk - 1) * 64]; \
	} }
#define FILTER_STAGE(SET, CHUNK_NO) { \
	const int cn_ = (CHUNK_NO) < nchunks ? (CHUNK_NO) : nchunks - 1; \
	_Pragma("unroll") for (int i = 0; i < kPieces; i++) { \
		const int row = prow + i * kRowsPerPass; \
		const bool inside = cn_ * CHUNK + row < nxp; \
		double2 v = piece[SET][i]; \
		v.x = inside ? v.x : 0.0; v.y = inside ? v.y : 0.0; \
		if (kWhole || row < CHUNK) reinterpret_cast<double2 *>(stage[SET])[i * 256 + threadIdx.x] = v; \
	} }
	// the B operands of a k-step are read from LDS one k-step ahead of the MFMAs that use them
#define FILTER_BODY(R, R1, RF, CHUNK_NO) { \
	const int c_ = (CHUNK_NO); \
	if (PROBE != 2 && PROBE != 3) FILTER_FETCH(RF, c_ + DEPTH) \
	if (working && c_ < nchunks && PROBE != 1) { \
		const double *cur = stage[R]; \
		double b[2][NB]; \
		_Pragma("unroll") for (int t = 0; t < NB; t++) b[0][t] = cur[kq * SPEC + jq + 16 * t]; \
		_Pragma("unroll") for (int i = 0; i < kSteps; i++) { \
			if (i + 1 < kSteps) { \
				_Pragma("unroll") for (int t = 0; t < NB; t++) b[(i + 1) & 1][t] = cur[((i + 1) * 4 + kq) * SPEC + jq + 16 * t]; \
			} \
			/* (k-steps past the last channel -- 24 of the 224 channel slots at 200 channels -- multiply \
			   zeros; skipping them with a branch per k-step measured SLOWER: 37.0 against 34.6 us) */ \
			_Pragma("unroll") for (int t = 0; t < NB; t++) \
				acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[R][i], PROBE == 3 ? a[R][(i + t + 1) % kSteps] : b[i & 1][t], acc[t], 0, 0, 0); \
		} \
	} \
	if (PROBE != 2 && PROBE != 3) FILTER_STAGE(R1, c_ + 1) \
	__syncthreads(); }
	FILTER_FETCH(0, 0)
	if (DEPTH == 2) FILTER_FETCH(1, 1)
	FILTER_STAGE(0, 0)
	__syncthreads();
	// (whole rounds of the register sets: past the last chunk a body still fetches, stages and meets
	// the barrier -- straight-line memory operations keep the load counts exact -- but multiplies nothing)
	if (DEPTH == 2) {
#pragma unroll 1
		for (int c = 0; c < nchunks; c += 3) {
			FILTER_BODY(0, 1, 2, c)
			FILTER_BODY(1, 2 % (DEPTH + 1), 0, c + 1)
			FILTER_BODY(2 % (DEPTH + 1), 0, 1, c + 2)
		}
	} else {
#pragma unroll 1
		for (int c = 0; c < nchunks; c += 2) {
			FILTER_BODY(0, 1, 1, c)
			FILTER_BODY(1, 0, 0, c + 1)
		}
	}
#undef FILTER_BODY
#undef FILTER_STAGE
#undef FILTER_FETCH
	if (!working) return;
	if (PROBE == 3) { double sum = 0.0; for (int t = 0; t < NB; t++) sum += acc[t][t]; if (sum == 12345.678) flags[0] = 1; return; }
	// votes: lane (kq, jq) holds, of block t, candidates 4 v + kq (v = 0..3) for spectrum 16 t + jq
	const double unit = fabs(scale) * (double) (nx + 8) * 0x1p-52;
	double mm[4];
#pragma unroll
	for (int v = 0; v < 4; v++) mm[v] = msq[bt * 16 + 4 * v + kq];
	bool hit[4] = {false, false, false, false}, maybe[4] = {false, false, false, false};
#pragma unroll
	for (int t = 0; t < NB; t++) {
		const int k = first + 16 * t + jq;
		const bool live = k < M;
		const int kk = live ? k : M - 1;
		const int d = thr_rows ? thr_rows[kk] : kk;
		const double thr = live ? higher[d] : __builtin_nan("");       // NaN compares false: no vote
		const double yy = ysq[d];
#pragma unroll
		for (int v = 0; v < 4; v++) {
			const double S = acc[t][v];
			const double Lf = scale * ((mm[v] - 2.0 * S) + yy);
			const double E4 = 4.0 * unit * ((mm[v] + 2.0 * fabs(S)) + yy);
			const bool valid = bt * 16 + 4 * v + kq < B;
			const bool h = valid && Lf > thr + E4;
			hit[v] = hit[v] || h;
			maybe[v] = maybe[v] || (valid && !h && Lf >= thr - E4);
		}
	}
	int best = 0x7fffffff;                                                // lowest clearly accepted candidate of this wave
#pragma unroll
	for (int v = 0; v < 4; v++) {
		const unsigned long long hm = __ballot(hit[v]), mb = __ballot(maybe[v]);
		if (lane == 0) {
#pragma unroll
			for (int g = 0; g < 4; g++) {
				const int cand = bt * 16 + 4 * v + g;
				if ((hm >> (16 * g)) & 0xffffull) { flags[cand] = 1; best = cand < best ? cand : best; }
				else if ((mb >> (16 * g)) & 0xffffull) ambiguous[cand] = stamp;    // (never cleared: stamped with the call)
			}
		}
	}
	if (lane == 0 && best != 0x7fffffff) atomicMax(lowest, B - best);
}

// The same filter with BOTH operands read straight from memory into the registers the instruction wants (round 4;
// what took K2's filter from 183 to 140 us; here it did NOT pay -- see launch_gauss_mfma_filter -- opt-in): the channel-major replica in tiles of 64 spectra and the templates in
// tiles of 16 candidates are both laid out so that the 16 lanes of a quarter wave -- one channel of 16 neighbouring
// spectra / candidates -- read 128 contiguous bytes.  No LDS, no barrier per chunk of channels; a wave owns
// 16 spectra x 16 NC candidates (A = spectra, B = candidates), the four waves of a workgroup share the spectra (L1).
// Channels past nxp come from 16 zeros.
template <int NC>
__global__ __launch_bounds__(256) void k_gauss_mfma_direct(
    const double *__restrict__ YT, int nxp, int nx, const double *__restrict__ model_t, const double *__restrict__ msq, int B,
    double scale, const int *__restrict__ thr_rows, int M, int nbt,
    const double *__restrict__ higher, const double *__restrict__ ysq, int *__restrict__ flags, int *__restrict__ ambiguous,
    int stamp, int *__restrict__ lowest, const double *__restrict__ zeros)
{
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
	const int i = lane & 15, q = lane >> 4;
	const int first = blockIdx.x * 16;                                    // first spectrum (place in the selection)
	const int ct0 = (blockIdx.y * 4 + wave) * NC;                         // first candidate tile of this wave
	if (ct0 >= nbt) return;                                               // whole waves
	const int sp = first + i;                                             // (a spectrum past M reads the replica's zero padding)
	const double *pa = YT + ((size_t) (sp >> 6) * nxp << 6) + (sp & 63);
	const double *pb[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) pb[c] = model_t + (size_t) (ct0 + c < nbt ? ct0 + c : nbt - 1) * nxp * 16 + i;
	double4_t acc[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) acc[c] = double4_t{0, 0, 0, 0};
	const int ng = (nxp + 15) >> 4;
	double a[2][4], b[2][NC][4];
#define GD_FETCH(SET, G) { \
	const int g_ = (G) < ng ? (G) : ng - 1; \
	_Pragma("unroll") for (int t = 0; t < 4; t++) { \
		const int ch = 16 * g_ + 4 * q + t; \
		const bool in = ch < nxp && (G) < ng; \
		a[SET][t] = *(in ? pa + ((size_t) ch << 6) : zeros + i); \
		const int cb = ch < nxp ? ch : nxp - 1; \
		_Pragma("unroll") for (int c = 0; c < NC; c++) b[SET][c][t] = pb[c][(size_t) cb << 4]; \
	} }
#define GD_BODY(SET) { \
	__builtin_amdgcn_sched_barrier(0); \
	_Pragma("unroll") for (int t = 0; t < 4; t++) \
		_Pragma("unroll") for (int c = 0; c < NC; c++) \
			acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET][t], b[SET][c][t], acc[c], 0, 0, 0); \
	__builtin_amdgcn_sched_barrier(0); }
	GD_FETCH(0, 0)
#pragma unroll 1
	for (int g = 0; g < ng; g += 2) {
		GD_FETCH(1, g + 1)
		GD_BODY(0)
		GD_FETCH(0, g + 2)
		GD_BODY(1)
	}
#undef GD_BODY
#undef GD_FETCH
	// votes: lane (i, q) holds, of candidate tile c, candidate 16 (ct0 + c) + i for the spectra first + 4 v + q
	const double unit = fabs(scale) * (double) (nx + 8) * 0x1p-52;
	double thr[4], yy[4];
#pragma unroll
	for (int v = 0; v < 4; v++) {
		const int k = first + 4 * v + q;
		const bool live = k < M;
		const int kk = live ? k : M - 1;
		const int d = thr_rows ? thr_rows[kk] : kk;
		thr[v] = live ? higher[d] : __builtin_nan("");                    // NaN compares false: no vote
		yy[v] = ysq[d];
	}
	int best = 0x7fffffff;                                                // lowest clearly accepted candidate of this wave
#pragma unroll
	for (int c = 0; c < NC; c++) {
		const int cand = (ct0 + c) * 16 + i;
		const bool valid = ct0 + c < nbt && cand < B;
		const double mm = msq[valid ? cand : 0];
		bool hit = false, maybe = false;
#pragma unroll
		for (int v = 0; v < 4; v++) {
			const double S = acc[c][v];
			const double Lf = scale * ((mm - 2.0 * S) + yy[v]);
			const double E4 = 4.0 * unit * ((mm + 2.0 * fabs(S)) + yy[v]);
			const bool h = valid && Lf > thr[v] + E4;
			hit = hit || h;
			maybe = maybe || (valid && !h && Lf >= thr[v] - E4);
		}
		// a candidate's four lanes (q = 0..3) are 16 apart
		const unsigned long long hm = __ballot(hit), mb = __ballot(maybe);
		if (lane < 16) {
			const unsigned long long mine = 0x0001000100010001ull << lane;
			if (hm & mine) { flags[cand] = 1; best = cand < best ? cand : best; }
			else if (mb & mine) ambiguous[cand] = stamp;                      // (never cleared: stamped with the call)
		}
	}
	// lowest accepted candidate of the wave (lanes 0..15 hold one each)
#pragma unroll
	for (int off = 8; off > 0; off >>= 1) { const int o = __shfl_xor(best, off, 64); best = o < best ? o : best; }
	if (lane == 0 && best != 0x7fffffff) atomicMax(lowest, B - best);
}

// The filter with both operands straight from memory in the TILED16 layout (round 4, after K2's: every load of a quarter
// wave is 256 contiguous bytes, a lane's four channels of a group are two 16-byte pieces).  A workgroup owns 16 spectra x
// 16 NC candidates and its four waves split the channel groups (a quarter each: the 200 channels of the bench shape are
// 13 groups -- too few to pipeline deeply, so 10 000 short waves instead of 2 500 long ones, which also evens out the
// SIMDs); the partial sums of waves 1-3 meet wave 0's in LDS behind ONE barrier, and wave 0 votes.
// Measured in that form: 37.0 / 34.4 us at 10 000 x 256 against 34.4 / 31.8 staged through LDS, 138.6 against 128.9 at
// 50 000 x 256 -- every 16-spectrum tile pulls all the templates through L2 again.  KSPLIT = false, the default: each wave
// takes 16 spectra of its own with ALL the channels, the four waves of a workgroup read the same templates at about the same
// time (L1): 29.8 / 28.3 us and 112.8 / 108.3 -- the fastest of the matrix-core forms, decisions identical (tests/test_joint.py).
template <int NC, bool KSPLIT>
__global__ __launch_bounds__(256) void k_gauss_gemm_filter(
    const double *__restrict__ YG, int nxg, int nx, const double *__restrict__ model_g, const double *__restrict__ msq, int B,
    double scale, const int *__restrict__ thr_rows, int M, int nbt,
    const double *__restrict__ higher, const double *__restrict__ ysq, int *__restrict__ flags, int *__restrict__ ambiguous,
    int stamp, int *__restrict__ lowest)
{
	__shared__ double red[KSPLIT ? 3 : 1][KSPLIT ? NC * 256 : 1];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
	const int i = lane & 15, q = lane >> 4;
	// KSPLIT: the four waves share 16 spectra and split the channel groups; else each wave has 16 spectra of its own and
	// all the groups -- the waves then read the SAME templates at about the same time (L1)
	const int rtile = KSPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
	const int first = rtile * 16;                                         // first spectrum (place in the selection)
	if (first >= M) return;                                               // (whole waves; no barrier on this path)
	const int ct0 = blockIdx.y * NC;                                      // first candidate tile
	const int ncp = nxg >> 1, ng = nxg >> 4;
	const int g0 = KSPLIT ? (wave * ng) >> 2 : 0, g1 = KSPLIT ? ((wave + 1) * ng) >> 2 : ng;   // this wave's channel groups
	const double *pa = YG + (((size_t) rtile * ncp + 2 * q) << 5) + 2 * i;
	const double *pb[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) pb[c] = model_g + (((size_t) (ct0 + c < nbt ? ct0 + c : nbt - 1) * ncp + 2 * q) << 5) + 2 * i;
	double4_t acc[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) acc[c] = double4_t{0, 0, 0, 0};
	double4_t a[2], b[2][NC];
#define GG_LOAD4(P, G) gg_join(*reinterpret_cast<const double2_t *>((P) + ((size_t) (G) << 8)), *reinterpret_cast<const double2_t *>((P) + ((size_t) (G) << 8) + 32))
#define GG_FETCH(SET, G) { \
	const int g_ = (G) < g1 ? (G) : g1 - 1; \
	a[SET] = GG_LOAD4(pa, g_); \
	_Pragma("unroll") for (int c = 0; c < NC; c++) b[SET][c] = GG_LOAD4(pb[c], g_); }
#define GG_BODY(SET, G) { \
	__builtin_amdgcn_sched_barrier(0); \
	if ((G) < g1) { \
		_Pragma("unroll") for (int t = 0; t < 4; t++) \
			_Pragma("unroll") for (int c = 0; c < NC; c++) \
				acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET][t], b[SET][c][t], acc[c], 0, 0, 0); \
	} \
	__builtin_amdgcn_sched_barrier(0); }
	if (g0 < g1) {
		GG_FETCH(0, g0)
#pragma unroll 1
		for (int g = g0; g < g1; g += 2) {
			GG_FETCH(1, g + 1)
			GG_BODY(0, g)
			GG_FETCH(0, g + 2)
			GG_BODY(1, g + 1)
		}
	}
#undef GG_BODY
#undef GG_FETCH
#undef GG_LOAD4
	if (KSPLIT) {
		if (wave != 0) {
#pragma unroll
			for (int c = 0; c < NC; c++)
#pragma unroll
				for (int v = 0; v < 4; v++) red[wave - 1][(c * 4 + v) * 64 + lane] = acc[c][v];
		}
		__syncthreads();
		if (wave != 0) return;
#pragma unroll
		for (int c = 0; c < NC; c++)
#pragma unroll
			for (int v = 0; v < 4; v++)
				acc[c][v] = ((acc[c][v] + red[0][(c * 4 + v) * 64 + lane]) + red[1][(c * 4 + v) * 64 + lane]) + red[2][(c * 4 + v) * 64 + lane];
	}
	// votes: lane (i, q) holds, of candidate tile c, candidate 16 (ct0 + c) + i for the spectra first + 4 v + q
	const double unit = fabs(scale) * (double) (nx + 8) * 0x1p-52;
	double thr[4], yy[4];
#pragma unroll
	for (int v = 0; v < 4; v++) {
		const int k = first + 4 * v + q;
		const bool live = k < M;
		const int kk = live ? k : M - 1;
		const int d = thr_rows ? thr_rows[kk] : kk;
		thr[v] = live ? higher[d] : __builtin_nan("");                    // NaN compares false: no vote
		yy[v] = ysq[d];
	}
	int best = 0x7fffffff;                                                // lowest clearly accepted candidate of this wave
#pragma unroll
	for (int c = 0; c < NC; c++) {
		const int cand = (ct0 + c) * 16 + i;
		const bool valid = ct0 + c < nbt && cand < B;
		const double mm = msq[valid ? cand : 0];
		bool hit = false, maybe = false;
#pragma unroll
		for (int v = 0; v < 4; v++) {
			const double S = acc[c][v];
			const double Lf = scale * ((mm - 2.0 * S) + yy[v]);
			const double E4 = 4.0 * unit * ((mm + 2.0 * fabs(S)) + yy[v]);
			const bool h = valid && Lf > thr[v] + E4;
			hit = hit || h;
			maybe = maybe || (valid && !h && Lf >= thr[v] - E4);
		}
		// a candidate's four lanes (q = 0..3) are 16 apart
		const unsigned long long hm = __ballot(hit), mb = __ballot(maybe);
		if (lane < 16) {
			const unsigned long long mine = 0x0001000100010001ull << lane;
			if (hm & mine) { flags[cand] = 1; best = cand < best ? cand : best; }
			else if (mb & mine) ambiguous[cand] = stamp;                      // (never cleared: stamped with the call)
		}
	}
#pragma unroll
	for (int off = 8; off > 0; off >>= 1) { const int o = __shfl_xor(best, off, 64); best = o < best ? o : best; }
	if (lane == 0 && best != 0x7fffffff) atomicMax(lowest, B - best);
}

// The chain's own sums for the candidates the filter could not settle -- the ambiguous ones below
// the lowest clear vote -- and for that one itself (its likelihoods are wanted): votes (flags) and
// the trail of likelihoods, exactly as the chain accept kernels leave them.  Every workgroup (one
// tile of 64 spectra) first lists those candidates (a few hundred flags: cheaper than a launch of
// its own), then scores them four at a time in the layout of k_chunk_accept: a quad of lanes shares
// one spectrum row and scores four listed candidates.
static constexpr int kExactMost = 64;
template <int NST>
__global__ __launch_bounds__(256) void k_exact_list(
    const double *__restrict__ Y, int ld, int nxp, const double *__restrict__ model_t, double scale,
    const int *__restrict__ rows, int M, int ntiles, const double *__restrict__ higher, int B,
    const int *__restrict__ ambiguous, int stamp, const int *__restrict__ lowest, int *__restrict__ flags, JointTrail trail,
    JointHeader *__restrict__ header)
{
	extern __shared__ __attribute__((aligned(16))) double lds[];
	double2 *tpl = reinterpret_cast<double2 *>(lds);                     // [nxp / 2][4] pairs of channels
	unsigned long long *votes = reinterpret_cast<unsigned long long *>(lds + (size_t) nxp * 4);
	__shared__ int list[kExactMost];
	__shared__ int s_n;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (wave == 0) {
		const int low = *lowest > 0 ? B - *lowest : B;                  // lowest clearly accepted candidate, or B
		int n = 0;
		for (int b0 = 0; b0 < B && b0 <= low; b0 += 64) {
			const int b = b0 + lane;
			// (below `low` nobody has a clear vote; flags are not read here: other workgroups of this launch write them)
			const bool take = b < B && ((b < low && ambiguous[b] == stamp) || b == low);
			const unsigned long long m = __ballot(take);
			if (take) {
				const int at = n + __popcll(m & ((1ull << lane) - 1ull));
				if (at < kExactMost) list[at] = b;
			}
			n += __popcll(m);
		}
		if (lane == 0) {
			s_n = n < kExactMost ? n : kExactMost;
			// more ambiguous candidates than the list holds: never seen; the chunk fails loudly
			if (n > kExactMost && blockIdx.x == 0) atomicOr(&header->status, 2);
		}
	}
	__syncthreads();
	const int n = s_n;
	if (n == 0) return;
	const int q = lane & 3;
	const int tile = blockIdx.x;
	const int r = wave * 16 + (lane >> 2);
	const int k = tile * 64 + r;
	const bool live = k < M;
	const int kk = live ? k : M - 1;
	const int row = rows ? rows[kk] : kk;
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
	for (int g = 0; g < n; g += 4) {
		const int mine = g + q < n ? list[g + q] : -1;
		// the listed candidates' template columns (tiled templates MT[tile16][channel][16]) into LDS
		for (int e = threadIdx.x; e < nxp * 4; e += 256) {
			const int j = e >> 2, bb = e & 3;
			const int c = g + bb < n ? list[g + bb] : -1;
			const double m = c >= 0 ? model_t[((size_t) (c >> 4) * nxp + j) * 16 + (c & 15)] : 0.0;
			lds[((size_t) (j >> 1) * 4 + bb) * 2 + (j & 1)] = m;
		}
		__syncthreads();
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
		const double L = acc * scale;
		const bool beat = mine >= 0 && L > thr;
		const unsigned long long vote = __ballot(beat);
		if (lane == 0) votes[wave] = vote;
		if (beat) trail.L[((size_t) mine * ntiles + tile) * 64 + r] = L;
		__syncthreads();
		{
			const int c = g + wave < n ? list[g + wave] : -1;
			const unsigned long long word = __ballot((votes[lane >> 4] >> (4 * (lane & 15) + wave)) & 1ull);
			if (c >= 0 && word != 0ull && lane == 0) {
				const size_t at = (size_t) c * ntiles + tile;
				flags[c] = 1;
				trail.word[at] = word;
				trail.stamp_of[at] = trail.stamp;
			}
		}
		__syncthreads();
	}
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

// the accept pass of an issue-bound chunk on the matrix cores (see k_gauss_mfma_filter): the same
// flags and trail as launch_gauss_cols_accept.  d_yT: the tiled replica of exactly the M spectra
// scored (no column gather); d_model_t must be tiled 16 candidates wide, d_msq
// from launch_gauss_model_tsq; d_scratch int32[MDNS_JOINT_MAX_BATCH + 16], zeroed once (ambiguous
// marks, stamped with the trail's stamp); d_lowest: an int the template kernel cleared.
// MDNS_K1_FILTER_FORM: gemm (default) | lds | direct
int gauss_mfma_form()
{
	static int form = -1;
	if (form < 0) {
		const char *e = getenv("MDNS_K1_FILTER_FORM");
		form = e && e[0] == 'd' ? 1 : (e && e[0] == 'l' ? 0 : 2);
	}
	return form;
}

bool launch_gauss_mfma_filter(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int B, double scale,
                              const int *d_thr_rows, int M, const double *d_higher, int *d_flags,
                              const double *d_msq, const JointTrail &trail, int *d_lowest, int *d_scratch, void *d_header,
                              const double *d_yG, const double *d_model_g)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64, nbt = (B + 15) / 16;
	const int nxp = cols_nx(s->nx);
	int *d_amb = d_scratch;
	{
		// Tiles of 32 spectra, chunks of 32 channels, one chunk ahead (100 VGPRs, 16 KB of LDS).  Measured
		// at 10 000 x 256 (rocprofv3, us): 64 spectra x 40 channels 30.3-32.4; 32 x 32 29-32; two chunks
		// ahead 36.9 (132 VGPRs: three waves per SIMD instead of four or five -- this kernel wants
		// independent workgroups around its barriers more than it wants distance to its loads; with the
		// L2s flushed between launches it takes the same 31); 16-channel chunks 30-31; 48-channel chunks
		// (five barriers instead of seven, 128 VGPRs) 37.0 against 34.7 on the same box.  The reads of
		// the B operands out of LDS forced ahead of the MFMAs that use them (scheduling barriers; left alone
		// the scheduler puts every read right in front of its use): a pair of k-steps ahead, still 128
		// VGPRs: 33.8 against 33.9 -- the LDS latency is not what the waves wait for; a whole chunk ahead
		// (152 VGPRs, three waves per SIMD): 35.8.
		// MDNS_FILTER_PROBE (experiments; results are wrong): 1 no MFMAs 12.8, 2 no loads after the first
		// chunk 26.5, 3 MFMAs on registers only 23.5 -- against 14.8 for the 502 400 MFMAs at the rate the
		// instruction sustains alone (69.7 TFLOP/s) on perfectly balanced SIMDs.
		static const char *probe = getenv("MDNS_FILTER_PROBE");
		const int pr = probe ? atoi(probe) : 0;
		// MDNS_K1_FILTER_FORM=direct: both operands straight from memory (k_gauss_mfma_direct).  Measured at 10 000 x 256:
		// 40.5 / 38.7 us against 34.4 / 31.8 staged through LDS; 50 000 x 256: 140.9 against 128.9 -- twenty 8-byte loads
		// per lane and 16 multiplications (the layouts give a lane one channel per load) cost more than the barriers
		// they save.  Not the default.
		if (pr == 0 && d_yG && d_model_g && gauss_mfma_form() == 2) {
			const int nxg = tiled16_nx(s->nx);
			// (32 candidates per wave: 29.8 / 28.3 us at 10 000 x 256 against 33.0 / 31.4 with 64 -- twice the waves, evener SIMDs --
			// and 112.8 / 108.3 against 112.2 / 108.4 at 50 000 x 256; 16: 42 / 179)
			static const char *ncf = getenv("MDNS_K1_GEMM_NC");                 // experiments only
			int nc = nbt >= 2 ? 2 : 1;
			if (ncf && (atoi(ncf) == 1 || atoi(ncf) == 2 || atoi(ncf) == 4) && atoi(ncf) <= nbt) nc = atoi(ncf);
			ProfileScope prof(0);
			note_kernel(0, "k_gauss_gemm_filter<%d>", nc);
			// MDNS_K1_GEMM_KSPLIT=1: the four waves of a workgroup share 16 spectra and split the channels (35.7 / 34.3 us at
			// 10 000 x 256, 138.7 / 135.6 at 50 000 x 256); default: a wave per 16 spectra with all the channels
			static const char *ks = getenv("MDNS_K1_GEMM_KSPLIT");
			const bool ksplit = ks && ks[0] == '1';
			const int rt = (M + 15) / 16;
#define GG_LAUNCH(NC) do { if (ksplit) hipLaunchKernelGGL((k_gauss_gemm_filter<NC, true>), dim3(rt, (nbt + NC - 1) / NC), dim3(256), 0, c->stream, \
			d_yG, nxg, s->nx, d_model_g, d_msq, B, scale, d_thr_rows, M, nbt, d_higher, (const double *) s->d_ysq, d_flags, d_amb, trail.stamp, d_lowest); \
		else hipLaunchKernelGGL((k_gauss_gemm_filter<NC, false>), dim3((rt + 3) / 4, (nbt + NC - 1) / NC), dim3(256), 0, c->stream, \
			d_yG, nxg, s->nx, d_model_g, d_msq, B, scale, d_thr_rows, M, nbt, d_higher, (const double *) s->d_ysq, d_flags, d_amb, trail.stamp, d_lowest); } while (0)
			if (nc == 4) GG_LAUNCH(4); else if (nc == 2) GG_LAUNCH(2); else GG_LAUNCH(1);
#undef GG_LAUNCH
			if (!launched("k_gauss_gemm_filter")) return false;
		} else if (pr == 0 && gauss_mfma_form() == 1) {
			static double *d_zeros = nullptr;
			if (!d_zeros && (!MDNS_HIP(hipMalloc((void **) &d_zeros, 16 * sizeof(double))) ||
			                 !MDNS_HIP(hipMemsetAsync(d_zeros, 0, 16 * sizeof(double), c->stream)))) return false;
			const int nc = nbt >= 16 ? 4 : (nbt >= 8 ? 2 : 1);                // candidate tiles per wave
			const int gy = (nbt + 4 * nc - 1) / (4 * nc);
			ProfileScope prof(0);
			note_kernel(0, "k_gauss_mfma_direct<%d>", nc);
#define GD_LAUNCH(NC) hipLaunchKernelGGL((k_gauss_mfma_direct<NC>), dim3((M + 15) / 16, gy), dim3(256), 0, c->stream, \
			d_yT, nxp, s->nx, d_model_t, d_msq, B, scale, d_thr_rows, M, nbt, d_higher, (const double *) s->d_ysq, d_flags, d_amb, trail.stamp, d_lowest, \
			(const double *) d_zeros)
			if (nc == 4) GD_LAUNCH(4); else if (nc == 2) GD_LAUNCH(2); else GD_LAUNCH(1);
#undef GD_LAUNCH
			if (!launched("k_gauss_mfma_direct")) return false;
		} else {
		const int nspec = (M + 31) / 32;
		const int ngroups = (nbt + 3) / 4;
		const int blocks = 8 * ((nspec + 7) / 8) * ngroups;
		ProfileScope prof(0);
		note_kernel(0, "k_gauss_mfma_filter");
#define MFMA_LAUNCH(D, P) hipLaunchKernelGGL((k_gauss_mfma_filter<32, 32, D, P>), dim3(blocks), dim3(256), 0, c->stream, \
		                   d_yT, nxp, s->nx, d_model_t, d_msq, B, scale, d_thr_rows, M, nspec, nbt, ngroups, d_higher, \
		                   (const double *) s->d_ysq, d_flags, d_amb, trail.stamp, d_lowest)
		// chunks of 40 channels where they leave fewer padded channels than chunks of 32 (200 channels:
		// none against 24 -- the padded k-steps multiply zeros: 34.1 against 38.9 us on the same box)
		const bool forty = pr == 0 && (nxp + 39) / 40 * 40 < (nxp + 31) / 32 * 32;
		if (forty) hipLaunchKernelGGL((k_gauss_mfma_filter<32, 40, 1, 0>), dim3(blocks), dim3(256), 0, c->stream,
		                              d_yT, nxp, s->nx, d_model_t, d_msq, B, scale, d_thr_rows, M, nspec, nbt, ngroups, d_higher,
		                              (const double *) s->d_ysq, d_flags, d_amb, trail.stamp, d_lowest);
		else switch (pr) {
		case 1: MFMA_LAUNCH(1, 1); break;
		case 2: MFMA_LAUNCH(1, 2); break;
		case 3: MFMA_LAUNCH(1, 3); break;
		case 4: MFMA_LAUNCH(2, 0); break;       // two chunks ahead (correct results)
		case 5: MFMA_LAUNCH(1, 0); break;       // 32-channel chunks whatever the padding
		default: MFMA_LAUNCH(1, 0); break;
		}
#undef MFMA_LAUNCH
		if (!launched("k_gauss_mfma_filter")) return false;
		}
	}
	const size_t lds = ((size_t) nxp * 4 + 4) * sizeof(double);
	const int nst = nxp / kCH;
#define EXACT_LAUNCH(NST) hipLaunchKernelGGL((k_exact_list<NST>), dim3(ntiles), dim3(256), lds, c->stream, \
	s->d_y, s->ld, nxp, d_model_t, scale, d_thr_rows, M, ntiles, d_higher, B, (const int *) d_amb, trail.stamp, (const int *) d_lowest, d_flags, trail, \
	(JointHeader *) d_header)
	if (nst <= 8) EXACT_LAUNCH(8); else if (nst <= 16) EXACT_LAUNCH(16); else if (nst <= 26) EXACT_LAUNCH(26); else EXACT_LAUNCH(32);
#undef EXACT_LAUNCH
	return launched("k_exact_list");
}

}  // namespace mdns
