// K2 accept pass of a chunk of candidates on the matrix cores: a GUARDED FILTER in front of the exact
// row kernels of mdns_like.hip (cmuselike.c:45-64 is what those compute and what is ever KEPT).
//
// With w = 1/v and the best-fit scale s = S1 / (1e-10 + S2) (cmuselike.c:52,57):
//     chi2 = sum_j (y_j - s m_j)^2 w_j = A - 2 s S1 + s^2 S2,
//     A = sum_j y_j^2 w_j (of the spectrum),  S1 = sum_j (y w)_j m_j,  S2 = sum_j w_j m_j^2:
// two matrix products [spectra x channels] . [channels x candidates] with y w and A made once per
// upload -- 4 flops per (candidate, channel, spectrum) where the residual form needs 10, and
// 1024 multiply-adds per v_mfma_f64_16x16x4_f64 with both operands arriving as plain 32-byte loads.
//
// The expanded form cancels, so its value Lf = -chi2 / 2 is NOT the library's likelihood L; but every
// one of its three terms is at most A in magnitude (Cauchy-Schwarz: |S1| <= sqrt(A S2)), each is a sum
// of at most nx + 16 products accumulated with fused multiply-adds in some order, and the exact
// kernel's own value carries (nx + 4) u chi2 <= (nx + 4) u A: |Lf - L| <= E := 32 (nx + 16) 2^-52 A with
// room to spare (the first-order bound is about 4 (nx + 16) u A).  The caller's band test
// (mdns_joint.hip, k_joint_band) is widened by E: a pair above  thr + band + E  certainly beats its
// threshold whatever the noise, one below  thr - band - E  certainly does not, and a chunk with a pair in
// between -- or a likelihood that is not finite -- is scored again by the exact kernels (about one chunk in
// 10^5).  The likelihood row of the ACCEPTED candidate, which is what the state keeps, always comes from
// the exact kernel (mdns_backend_draw_band_commit).
#include "mdns_internal.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace mdns {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double4_t k2_join(double2_t a, double2_t b) { return double4_t{a[0], a[1], b[0], b[1]}; }
// the spectrum operands are read once per launch: past the caches' replacement (the templates are re-read by everybody)
#define K2_ROWLOAD(p) __builtin_nontemporal_load(p)

// element (row r, channel c) of a TILED operand: tiles of 16 rows, inside a tile channel pair by channel pair,
// 16 rows x 2 doubles each -- so that the 16 lanes of a quarter wave, which hold the same channels of 16
// different rows, read 256 contiguous bytes (row-major they make 16 requests of 16 bytes each)
__host__ __device__ inline size_t tiled_at(size_t r, int c, int ncp) { return (((r >> 4) * (size_t) ncp + (size_t) (c >> 1)) << 5) + ((r & 15) << 1) + (size_t) (c & 1); }

// one workgroup per spectrum: y w and w, zero padded to ldf channels, row-major and tiled, and A = sum y^2 w
__global__ __launch_bounds__(256) void k_muse_filter_prepare(const double *__restrict__ Y, const double *__restrict__ W, int ld, int nx,
                                                            double *__restrict__ YW, double *__restrict__ WF, int ldf,
                                                            double *__restrict__ YWt, double *__restrict__ Wt, double *__restrict__ A)
{
	__shared__ double part[4];
	const size_t r = blockIdx.x;
	double acc = 0.0;
	for (int j = threadIdx.x; j < ldf; j += 256) {
		const double y = j < nx ? Y[r * ld + j] : 0.0, w = j < nx ? W[r * ld + j] : 0.0;
		const double yw = y * w;
		YW[r * ldf + j] = yw;
		WF[r * ldf + j] = w;
		YWt[tiled_at(r, j, ldf >> 1)] = yw;
		Wt[tiled_at(r, j, ldf >> 1)] = w;
		acc = fma(y, yw, acc);
	}
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) A[r] = (part[0] + part[1]) + (part[2] + part[3]);
}

// templates [B][ldm] -> tiled (16 candidates per tile; the candidates past B of the last tile stay as they are:
// a column of the product depends on its own operand column only, and nobody looks at theirs)
__global__ void k_muse_tile_templates(const double *__restrict__ model, int ldm, int B, int ldf, double *__restrict__ Mt)
{
	const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < ldf) Mt[tiled_at((size_t) b, j, ldf >> 1)] = model[(size_t) b * ldm + j];
}

// lane l holds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16], D[i = 4 v + l / 16][j = l % 16] in its v-th
// result (tools/probes/mfma_f64_probe.hip).  Any assignment of channels to (k-step, k) serves a sum over
// channels as long as both operands use the same one: lane (i, q) loads channels 16 g + 4 q .. + 3 of ITS
// spectrum row and of ITS candidate's template -- 32 contiguous bytes per operand, the four q of a row one
// 128-byte line -- and feeds element t of them to k-step t of group g.
//
// A workgroup of KW waves owns 16 selected spectra x 16 NC candidates; wave w takes the channel groups
// g = w, w + KW, ... (so that a row is read once, 128 KW contiguous bytes per step of the workgroup) and
// the partial sums meet in LDS (ds_add_f64).  The epilogue is the band test of k_joint_band on Lf.
template <int NC, int KW>
__global__ __launch_bounds__(64 * KW) void k_muse_gemm_band(
    const double *__restrict__ YW, const double *__restrict__ WF, int ldf, const double *__restrict__ A,
    const double *__restrict__ model, int ldm, int B, const int *__restrict__ rows, int M,
    const double *__restrict__ higher, const double *__restrict__ bound, double gamma, MuseBandOut out)
{
	if (out.zero_at && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *out.zero_at = 0;
	__shared__ double red[NC * 2 * 256];                                  // [c][S1 | S2][v * 64 + lane]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int i = lane & 15, q = lane >> 4;
	const int k0 = blockIdx.x * 16, b0 = blockIdx.y * 16 * NC;
	for (int t = threadIdx.x; t < NC * 2 * 256; t += 64 * KW) red[t] = 0.0;
	const int krow = k0 + i < M ? k0 + i : M - 1;
	const size_t r = rows ? rows[krow] : krow;
	const double *pyw = YW + r * ldf + 4 * q, *pw = WF + r * ldf + 4 * q;
	const double *pm[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) {
		const int b = b0 + 16 * c + i < B ? b0 + 16 * c + i : B - 1;
		pm[c] = model + (size_t) b * ldm + 4 * q;
	}
	double4_t acc1[NC], acc2[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) { acc1[c] = double4_t{0, 0, 0, 0}; acc2[c] = double4_t{0, 0, 0, 0}; }
	const int ng = (ldf >> 4) - 1;                                        // (the last 16 of a row are padding: muse_filter_ld)
	// two register sets in rotation: the loads of the wave's next group are in flight while this one is
	// multiplied (a group past the end fetches the last one again: straight-line loads, exact counts)
	double4_t yw[2], w[2], m[2][NC];
#define K2_FETCH(SET, G) { \
	const size_t o_ = (size_t) ((G) < ng ? (G) : ng - 1) << 4; \
	yw[SET] = *reinterpret_cast<const double4_t *>(pyw + o_); \
	w[SET] = *reinterpret_cast<const double4_t *>(pw + o_); \
	_Pragma("unroll") for (int c = 0; c < NC; c++) m[SET][c] = *reinterpret_cast<const double4_t *>(pm[c] + o_); }
#define K2_BODY(SET) { \
	_Pragma("unroll") for (int t = 0; t < 4; t++) { \
		_Pragma("unroll") for (int c = 0; c < NC; c++) { \
			const double mv = m[SET][c][t]; \
			acc1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(yw[SET][t], mv, acc1[c], 0, 0, 0); \
			acc2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[SET][t], mv * mv, acc2[c], 0, 0, 0); \
		} } }
	K2_FETCH(0, wave)
#pragma unroll 1
	for (int g = wave; g < ng; g += 2 * KW) {
		K2_FETCH(1, g + KW)
		K2_BODY(0)
		K2_FETCH(0, g + 2 * KW)
		if (g + KW < ng) K2_BODY(1)
	}
#undef K2_BODY
#undef K2_FETCH
	__syncthreads();                                                      // (the zeros are in place)
#pragma unroll
	for (int c = 0; c < NC; c++)
#pragma unroll
		for (int v = 0; v < 4; v++) {
			unsafeAtomicAdd(&red[(c * 2) * 256 + v * 64 + lane], acc1[c][v]);
			unsafeAtomicAdd(&red[(c * 2 + 1) * 256 + v * 64 + lane], acc2[c][v]);
		}
	__syncthreads();
	for (int e = threadIdx.x; e < NC * 256; e += 64 * KW) {
		const int c = e >> 8, rem = e & 255, v = rem >> 6, l = rem & 63;
		const int k = k0 + 4 * v + (l >> 4), b = b0 + 16 * c + (l & 15);
		if (k >= M || b >= B) continue;
		const double S1 = red[(c * 2) * 256 + rem], S2 = red[(c * 2 + 1) * 256 + rem];
		const int d = rows ? rows[k] : k;
		const double a = A[d], thr = higher[d];
		const double s = S1 / (1e-10 + S2);
		const double Lf = -0.5 * ((a - 2.0 * s * S1) + s * s * S2);
		const double band = (1.01 * bound[b] + 1e-12 * (fabs(Lf) + fabs(thr))) + gamma * a;
		if (Lf > thr + band) out.clear[b] = 1;
		else if (!(Lf < thr - band)) {                                   // (NaN lands here)
			out.maybe[b] = 1;
			const int at = atomicAdd(out.counter, 1);
			if (at < out.cap) { out.pair_b[at] = b; out.pair_k[at] = k; out.pair_L[at] = Lf; out.pair_thr[at] = thr; }
		}
	}
}

// The same products with the work spread evenly (stream-K): tiles x channel groups form ONE line of units,
// workgroup w of P (one per CU) takes the stretch [w U / P, (w + 1) U / P) of it whatever tiles that crosses --
// 391 row tiles on 256 CUs would otherwise leave half the chip idle for the second half of the launch.  A
// stretch that covers a whole tile ends in the band test as above; a piece of a split tile goes into a
// slot of its workgroup in device memory behind a count of the groups delivered, and the workgroup whose
// piece completes the count -- it alone knows that all pieces are there -- adds the slots up in a fixed order,
// tests the tile and leaves the count at zero for the next launch.  (Adding the pieces into one scratch tile
// with f64 atomics measured 365 us against 221 for whole tiles, 6250 x 4096 x 64: 1.6 million device-scope
// atomics.)
template <int NC, bool TILED>
__global__ __launch_bounds__(512) void k_muse_gemm_band_sk(
    const double *__restrict__ YW, const double *__restrict__ WF, int ldf, const double *__restrict__ A,
    const double *__restrict__ model, int ldm, int B, const int *__restrict__ rows, int M, int bt,
    const double *__restrict__ higher, const double *__restrict__ bound, double gamma, MuseBandOut out,
    double *__restrict__ scratch, unsigned *__restrict__ delivered, const double *__restrict__ zeros)
{
	constexpr int KW = 8, NE = NC * 2 * 256;
	if (out.zero_at && blockIdx.x == 0 && threadIdx.x == 0) *out.zero_at = 0;
	__shared__ double red[NE];                                            // [c][S1 | S2][v * 64 + lane]
	__shared__ int last_piece;
	// (the wave number as a scalar: with it in a vector register the loop below counts as divergent and the
	// compiler waits for every load in flight, vmcnt(0), at its head)
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
	const int i = lane & 15, q = lane >> 4;
	const int ng = (ldf >> 4) - 1, rt = (M + 15) >> 4;                    // (the last 16 of a row are padding: muse_filter_ld)
	const double *zr = zeros + 4 * q;
	const long long U = (long long) rt * bt * ng;
	long long u = U * blockIdx.x / gridDim.x;
	const long long u1 = U * (blockIdx.x + 1) / gridDim.x;
	while (u < u1) {
		const int tile = (int) (u / ng), g0 = (int) (u - (long long) tile * ng);
		const int g1 = u1 - u < ng - g0 ? g0 + (int) (u1 - u) : ng;       // this workgroup's groups [g0, g1) of the tile
		u += g1 - g0;
		const int k0 = (tile / bt) * 16, b0 = (tile % bt) * 16 * NC;
		for (int t = threadIdx.x; t < NE; t += 64 * KW) red[t] = 0.0;
		// TILED (the whole set of spectra, rows == nullptr): YW, WF and model are the tiled replicas, a lane's four
		// channels of a group are two pairs 256 bytes apart, and a quarter wave reads 256 contiguous bytes
		const int krow = k0 + i < M ? k0 + i : M - 1;
		const size_t r = rows ? rows[krow] : krow;
		const int ncp = ldf >> 1;
		const double *pyw = TILED ? YW + (((size_t) (tile / bt) * ncp + 2 * q) << 5) + 2 * i : YW + r * ldf + 4 * q;
		const double *pw = TILED ? WF + (((size_t) (tile / bt) * ncp + 2 * q) << 5) + 2 * i : WF + r * ldf + 4 * q;
		const double *pm[NC];
#pragma unroll
		for (int c = 0; c < NC; c++) {
			const int b = b0 + 16 * c + i < B ? b0 + 16 * c + i : B - 1;
			pm[c] = TILED ? model + (((size_t) ((b0 >> 4) + c) * ncp + 2 * q) << 5) + 2 * i : model + (size_t) b * ldm + 4 * q;
		}
		double4_t acc1[NC], acc2[NC];
#pragma unroll
		for (int c = 0; c < NC; c++) { acc1[c] = double4_t{0, 0, 0, 0}; acc2[c] = double4_t{0, 0, 0, 0}; }
		double4_t yw[2], w[2], m[2][NC];
		// whole rounds of both register sets, straight-line: a group past the end of the stretch takes its spectrum
		// operands from 16 zeros (and any template), so the loads in flight can be counted and waited for one set at a time
#define K2_PTR(P, O) (TILED ? (P) + ((O) << 4) : (P) + (O))
#define K2_LOAD4(PTR) (TILED ? k2_join(K2_ROWLOAD(reinterpret_cast<const double2_t *>(PTR)), K2_ROWLOAD(reinterpret_cast<const double2_t *>((PTR) + 32))) \
                             : K2_ROWLOAD(reinterpret_cast<const double4_t *>(PTR)))
#define K2_TLOAD4(PTR) (TILED ? k2_join(*reinterpret_cast<const double2_t *>(PTR), *reinterpret_cast<const double2_t *>((PTR) + 32)) \
                              : *reinterpret_cast<const double4_t *>(PTR))
#define K2_FETCH(SET, G) { \
		const bool ok_ = (G) < g1; \
		const size_t o_ = (size_t) (ok_ ? (G) : g1 - 1) << 4; \
		const double *a_ = ok_ ? K2_PTR(pyw, o_) : zr, *b_ = ok_ ? K2_PTR(pw, o_) : zr; \
		yw[SET] = K2_LOAD4(a_); \
		w[SET] = K2_LOAD4(b_); \
		_Pragma("unroll") for (int c = 0; c < NC; c++) { const double *t_ = K2_PTR(pm[c], o_); m[SET][c] = K2_TLOAD4(t_); } }
		// (the squares first, all of them: a multiplication waiting for the square made just before it held up the
		// matrix pipe)
#define K2_BODY(SET) { \
		double4_t m2[NC]; \
		_Pragma("unroll") for (int c = 0; c < NC; c++) m2[c] = m[SET][c] * m[SET][c]; \
		__builtin_amdgcn_sched_barrier(0); \
		_Pragma("unroll") for (int t = 0; t < 4; t++) { \
			_Pragma("unroll") for (int c = 0; c < NC; c++) { \
				acc1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(yw[SET][t], m[SET][c][t], acc1[c], 0, 0, 0); \
				acc2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[SET][t], m2[c][t], acc2[c], 0, 0, 0); \
			} } }
		K2_FETCH(0, g0 + wave)
#pragma unroll 1
		for (int g = g0 + wave; g < g1; g += 2 * KW) {
#ifdef MDNS_K2_PROBE
			// (experiments: 1 no loads in the loop; 2 no multiplications)
#if MDNS_K2_PROBE == 1
			K2_BODY(0) K2_BODY(1)
			continue;
#elif MDNS_K2_PROBE == 2
			K2_FETCH(1, g + KW)
			K2_FETCH(0, g + 2 * KW)
			acc1[0] += yw[0] + w[0] + yw[1] + w[1];
			_Pragma("unroll") for (int c = 0; c < NC; c++) acc2[c] += m[0][c] + m[1][c];
			continue;
#endif
#endif
			// (the fence keeps the loads of the next group at the head of the 32 multiplications they overlap
			// with; left alone the scheduler sinks them to a few instructions before their use)
			K2_FETCH(1, g + KW)
			__builtin_amdgcn_sched_barrier(0);
			K2_BODY(0)
			__builtin_amdgcn_sched_barrier(0);
			K2_FETCH(0, g + 2 * KW)
			__builtin_amdgcn_sched_barrier(0);
			K2_BODY(1)
			__builtin_amdgcn_sched_barrier(0);
		}
#undef K2_BODY
#undef K2_FETCH
#undef K2_LOAD4
#undef K2_TLOAD4
#undef K2_PTR
		__syncthreads();                                                  // (the zeros are in place)
		if (g0 + wave < g1) {
#pragma unroll
			for (int c = 0; c < NC; c++)
#pragma unroll
				for (int v = 0; v < 4; v++) {
					unsafeAtomicAdd(&red[(c * 2) * 256 + v * 64 + lane], acc1[c][v]);
					unsafeAtomicAdd(&red[(c * 2 + 1) * 256 + v * 64 + lane], acc2[c][v]);
				}
		}
		__syncthreads();
		bool finish = true;
		if (g1 - g0 < ng) {
			// a piece of a tile: into this workgroup's own slot (0: the piece its stretch starts with, 1: the
			// one it ends with); the piece that completes the count of groups delivered adds up the slots of
			// the workgroups the tile is spread over, in their order
			const long long ubeg = U * blockIdx.x / gridDim.x;
			const long long upiece = (long long) tile * ng + g0;
			double *slot = scratch + ((size_t) blockIdx.x * 2 + (upiece == ubeg ? 0 : 1)) * NE;
			// (agent-scope stores go through to memory and the loads below come from there; what has to be waited
			// for is their completion -- a device-scope fence would write back and invalidate this XCD's whole L2,
			// templates included, once per piece: 365 us a launch against 221 without split tiles)
			for (int e = threadIdx.x; e < NE; e += 64 * KW) __hip_atomic_store(&slot[e], red[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__syncthreads();
			if (threadIdx.x == 0) {
				const unsigned before = atomicAdd(&delivered[tile], (unsigned) (g1 - g0));
				last_piece = before + (unsigned) (g1 - g0) == (unsigned) ng ? 1 : 0;
			}
			__syncthreads();
			finish = last_piece != 0;
			if (finish) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				const long long t0 = (long long) tile * ng, t1 = t0 + ng;
				// workgroups w with [U w / P, U (w + 1) / P) meeting [t0, t1)
				long long wlo = t0 * gridDim.x / U, whi = (t1 * gridDim.x + U - 1) / U;
				while (wlo > 0 && U * wlo / gridDim.x > t0) wlo--;
				while (U * (wlo + 1) / gridDim.x <= t0) wlo++;
				if (whi > gridDim.x) whi = gridDim.x;
				for (int e = threadIdx.x; e < NE; e += 64 * KW) red[e] = 0.0;
				for (long long wg = wlo; wg < whi; wg++) {
					const long long a0 = U * wg / gridDim.x, a1 = U * (wg + 1) / gridDim.x;
					const long long p0 = a0 > t0 ? a0 : t0, p1 = a1 < t1 ? a1 : t1;
					if (p0 >= p1) continue;
					const double *from = scratch + ((size_t) wg * 2 + (p0 == a0 ? 0 : 1)) * NE;
					for (int e = threadIdx.x; e < NE; e += 64 * KW) red[e] += __hip_atomic_load(&from[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				if (threadIdx.x == 0) delivered[tile] = 0;
				__syncthreads();
			}
		}
		if (finish) {
			for (int e = threadIdx.x; e < NC * 256; e += 64 * KW) {
				const int c = e >> 8, rem = e & 255, v = rem >> 6, l = rem & 63;
				const int k = k0 + 4 * v + (l >> 4), b = b0 + 16 * c + (l & 15);
				if (k >= M || b >= B) continue;
				const double S1 = red[(c * 2) * 256 + rem], S2 = red[(c * 2 + 1) * 256 + rem];
				const int d = rows ? rows[k] : k;
				const double a = A[d], thr = higher[d];
				const double s = S1 / (1e-10 + S2);
				const double Lf = -0.5 * ((a - 2.0 * s * S1) + s * s * S2);
				const double band = (1.01 * bound[b] + 1e-12 * (fabs(Lf) + fabs(thr))) + gamma * a;
				if (Lf > thr + band) out.clear[b] = 1;
				else if (!(Lf < thr - band)) {                               // (NaN lands here)
					out.maybe[b] = 1;
					const int at = atomicAdd(out.counter, 1);
					if (at < out.cap) { out.pair_b[at] = b; out.pair_k[at] = k; out.pair_L[at] = Lf; out.pair_thr[at] = thr; }
				}
			}
		}
		__syncthreads();                                                  // (red is free for the next stretch)
	}
}

// scratch tiles and delivery counts of the stream-K form: grow-only, zero whenever no launch is in flight
static double *g_sk_scratch = nullptr, *g_sk_zeros = nullptr, *g_sk_templ = nullptr;
static size_t g_sk_templ_cap = 0;
static unsigned *g_sk_delivered = nullptr;
static size_t g_sk_tiles = 0, g_sk_doubles = 0;

static bool sk_reserve(size_t tiles, size_t doubles)
{
	Context *c = ctx();
	if (!g_sk_zeros && (!MDNS_HIP(hipMalloc((void **) &g_sk_zeros, 64 * sizeof(double))) ||
	                    !MDNS_HIP(hipMemsetAsync(g_sk_zeros, 0, 64 * sizeof(double), c->stream)))) return false;
	if (tiles > g_sk_tiles) {
		if (g_sk_delivered) { (void) hipStreamSynchronize(c->stream); (void) hipFree(g_sk_delivered); g_sk_delivered = nullptr; g_sk_tiles = 0; }
		const size_t want = tiles + tiles / 2 + 64;
		if (!MDNS_HIP(hipMalloc((void **) &g_sk_delivered, want * sizeof(unsigned))) ||
		    !MDNS_HIP(hipMemsetAsync(g_sk_delivered, 0, want * sizeof(unsigned), c->stream))) return false;
		g_sk_tiles = want;
	}
	if (doubles > g_sk_doubles) {
		if (g_sk_scratch) { (void) hipStreamSynchronize(c->stream); (void) hipFree(g_sk_scratch); g_sk_scratch = nullptr; g_sk_doubles = 0; }
		const size_t want = doubles + doubles / 2;
		if (!MDNS_HIP(hipMalloc((void **) &g_sk_scratch, want * sizeof(double))) ||
		    !MDNS_HIP(hipMemsetAsync(g_sk_scratch, 0, want * sizeof(double), c->stream))) return false;
		g_sk_doubles = want;
	}
	return true;
}

static long long g_filter_stats[4];        // chunks filtered | scored again exactly | exact rows for a commit | prepared handles

void muse_filter_note(int which) { if (which >= 0 && which < 4) g_filter_stats[which]++; }

// -1: by shape; 0: never; 1: every chunk (mdns_muse_filter_mode; MDNS_K2_FILTER sets the start value)
static int g_filter_mode = -2;

bool muse_filter_applies(const mdns_spectra *s, int B, int M)
{
	if (g_filter_mode == -2) {
		const char *forced = getenv("MDNS_K2_FILTER");
		g_filter_mode = forced && forced[0] == '0' ? 0 : (forced && forced[0] == '1' ? 1 : -1);
	}
	if (!s || !s->d_w || !s->d_x || s->nx < 1 || B < 1 || M < 1) return false;
	if (g_filter_mode == 0) return false;
	if (g_filter_mode == 1) return true;
	// where the two-row kernel is what the exact path takes (mdns_like.hip, launch_muse_rows) and a pass
	// is long against the exact row a commit then needs
	return B >= 8 && M >= 512 && s->nx >= 256;
}

// row stride of the filter's operands (spectra and templates alike): nx rounded up to 16, plus 16.  A wave's
// load touches 16 rows at once; with rows a power of two apart (4096 channels = 32 KB) all 16 lines fall into
// the same L2 channel (measured: the loads alone took as long as the whole kernel)
int muse_filter_ld(int nx) { return ((nx + 15) & ~15) + 16; }

bool muse_filter_prepare(mdns_spectra *s)
{
	if (s->d_fyw) return true;
	Context *c = ctx();
	const int ldf = muse_filter_ld(s->nx);
	const size_t elems = (size_t) s->ndata * ldf, telems = (size_t) ((s->ndata + 15) / 16) * 16 * ldf;
	double *buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // y w, w (row-major), y w, w (tiled), A
	const size_t want[5] = {elems, elems, telems, telems, (size_t) s->ndata};
	bool ok = true;
	for (int t = 0; t < 5 && ok; t++) ok = MDNS_HIP(hipMalloc((void **) &buf[t], (want[t] ? want[t] : 1) * sizeof(double)));
	// (the rows past the last one of the last tile multiply as zeros)
	for (int t = 2; t < 4 && ok; t++) ok = MDNS_HIP(hipMemsetAsync(buf[t], 0, (telems ? telems : 1) * sizeof(double), c->stream));
	if (ok && s->ndata > 0) {
		hipLaunchKernelGGL(k_muse_filter_prepare, dim3(s->ndata), dim3(256), 0, c->stream, (const double *) s->d_y, (const double *) s->d_w,
		                   s->ld, s->nx, buf[0], buf[1], ldf, buf[2], buf[3], buf[4]);
		ok = MDNS_HIP(hipGetLastError());
	}
	if (!ok) { for (double *b : buf) if (b) (void) hipFree(b); return false; }
	s->d_fyw = buf[0]; s->d_fw = buf[1]; s->fw_owned = true; s->d_fyw_t = buf[2]; s->d_fw_t = buf[3]; s->d_fa = buf[4]; s->ldf = ldf;
	muse_filter_note(3);
	return true;
}

bool launch_muse_filter(mdns_spectra *s, const double *d_model, int ldm, int B, const int *d_rows, int M,
                        const double *d_higher, const double *d_bound, const MuseBandOut &out)
{
	Context *c = ctx();
	if (!muse_filter_prepare(s)) return false;
	if (ldm < s->ldf) { set_error("launch_muse_filter: templates of %d channels for rows of %d", ldm, s->ldf); return false; }
	const double gamma = 32.0 * (double) (s->nx + 16) * 0x1p-52;
	const int rt = (M + 15) / 16;
	static const char *kw_forced = getenv("MDNS_K2_FILTER_KW");          // experiments only
	static const char *nc_forced = getenv("MDNS_K2_FILTER_NC");
	int nc = B > 32 ? 4 : (B > 16 ? 2 : 1);
	if (nc_forced) { const int f = atoi(nc_forced); if (f == 1 || f == 2 || f == 4) nc = f; }
	const int bt = (B + 16 * nc - 1) / (16 * nc);
	static const char *sk_forced = getenv("MDNS_K2_FILTER_SK");          // "0": one workgroup per tile (experiments)
	const long long tiles = (long long) rt * bt;
	if (!(sk_forced && sk_forced[0] == '0')) {
		// stream-K: one workgroup per CU, fewer when that would leave a workgroup less than 16 groups of channels
		const long long U = tiles * ((s->ldf >> 4) - 1);
		long long P = c->num_cus;
		if (P > (U + 15) / 16) P = (U + 15) / 16;
		if (P < 1) P = 1;
		static const char *p_forced = getenv("MDNS_K2_FILTER_P");        // experiments only; 0: one per tile
		if (p_forced) { const long long f = atoll(p_forced); P = f > 0 ? f : tiles; }
		if (!sk_reserve((size_t) tiles, (size_t) P * 2 * nc * 512)) return false;
		// the whole set of spectra in its stored order: tiled operands (every load of a quarter wave contiguous);
		// the templates are tiled on the way (2 MB: one more small launch)
		static const char *tiled_forced = getenv("MDNS_K2_FILTER_TILED");    // "0": row-major operands always (experiments)
		const bool tiled = !d_rows && M == s->ndata && !(tiled_forced && tiled_forced[0] == '0');
		const double *d_templ = d_model;
		if (tiled) {
			const size_t need = (size_t) bt * nc * 16 * s->ldf;
			if (need > g_sk_templ_cap) {
				if (g_sk_templ) { (void) hipStreamSynchronize(c->stream); (void) hipFree(g_sk_templ); g_sk_templ = nullptr; g_sk_templ_cap = 0; }
				if (!MDNS_HIP(hipMalloc((void **) &g_sk_templ, (need + need / 2) * sizeof(double)))) return false;
				if (!MDNS_HIP(hipMemsetAsync(g_sk_templ, 0, (need + need / 2) * sizeof(double), c->stream))) return false;
				g_sk_templ_cap = need + need / 2;
			}
			hipLaunchKernelGGL(k_muse_tile_templates, dim3((s->ldf + 255) / 256, B), dim3(256), 0, c->stream, d_model, ldm, B, s->ldf, g_sk_templ);
			d_templ = g_sk_templ;
		}
		note_kernel(1, tiled ? "k_muse_gemm_band_sk<%d, tiled>" : "k_muse_gemm_band_sk<%d>", nc);
#define K2_SK(NC, T) hipLaunchKernelGGL((k_muse_gemm_band_sk<NC, T>), dim3((unsigned) P), dim3(512), 0, c->stream, \
		(const double *) (T ? s->d_fyw_t : s->d_fyw), (const double *) (T ? s->d_fw_t : s->d_fw), s->ldf, (const double *) s->d_fa, d_templ, ldm, B, d_rows, M, bt, \
		d_higher, d_bound, gamma, out, g_sk_scratch, g_sk_delivered, (const double *) g_sk_zeros)
		{
			ProfileScope prof(1);
			if (tiled) { if (nc == 4) K2_SK(4, true); else if (nc == 2) K2_SK(2, true); else K2_SK(1, true); }
			else { if (nc == 4) K2_SK(4, false); else if (nc == 2) K2_SK(2, false); else K2_SK(1, false); }
		}
#undef K2_SK
		if (!MDNS_HIP(hipGetLastError())) return false;
		muse_filter_note(0);
		return true;
	}
	// waves over the channels of a tile: as many as it takes to give every SIMD a few waves
	int kw = tiles * 4 >= 3LL * 4 * c->num_cus ? 4 : 8;
	if (kw_forced) { const int f = atoi(kw_forced); if (f == 4 || f == 8) kw = f; }
	while (kw > 4 && (s->ldf >> 4) - 1 < 2 * kw) kw >>= 1;
	note_kernel(1, "k_muse_gemm_band<%d, %d>", nc, kw);
#define K2_LAUNCH(NC, KW) hipLaunchKernelGGL((k_muse_gemm_band<NC, KW>), dim3(rt, bt), dim3(64 * KW), 0, c->stream, \
		(const double *) s->d_fyw, (const double *) s->d_fw, s->ldf, (const double *) s->d_fa, d_model, ldm, B, d_rows, M, d_higher, d_bound, gamma, out)
#define K2_PICK(NC) do { if (kw == 4) K2_LAUNCH(NC, 4); else K2_LAUNCH(NC, 8); } while (0)
	{
		ProfileScope prof(1);
		if (nc == 4) K2_PICK(4); else if (nc == 2) K2_PICK(2); else K2_PICK(1);
	}
#undef K2_PICK
#undef K2_LAUNCH
	if (!MDNS_HIP(hipGetLastError())) return false;
	muse_filter_note(0);
	return true;
}

}  // namespace mdns

// Part 3 (raw device pointers): the accept pass alone, for benches and tests
extern "C" int mdns_muse_filter_dev(mdns_spectra *s, const double *d_ypred, int B, const int *d_row_ids, int M, const double *d_thr,
                                    const double *d_bound, int *d_out)
{
	using namespace mdns;
	if (!ctx() || !s || !d_ypred || !d_thr || !d_bound || !d_out || B < 1 || M < 1 || M > s->ndata) { set_error("mdns_muse_filter_dev: bad arguments"); return 1; }
	if (!s->d_w) { set_error("spectra were created without variances"); return 1; }
	const int ldm = model_ld(s->nx) + 16;
	if (!ensure_model(s, (size_t) B * ldm) || !launch_pad_model(d_ypred, s->nx, B, s->d_model, ldm)) return 1;
	const MuseBandOut out = {d_out + 2 * B, d_out, d_out + B, nullptr, nullptr, nullptr, nullptr, 0, nullptr};
	return launch_muse_filter(s, s->d_model, ldm, B, d_row_ids, M, d_thr, d_bound, out) ? 0 : 1;
}

extern "C" void mdns_muse_filter_mode(int mode) { mdns::g_filter_mode = mode == 0 || mode == 1 ? mode : -1; }

extern "C" void mdns_muse_filter_stats(long long *out4)
{
	if (!out4) return;
	for (int t = 0; t < 4; t++) out4[t] = mdns::g_filter_stats[t];
}
