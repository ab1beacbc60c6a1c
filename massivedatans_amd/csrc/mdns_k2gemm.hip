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

// one workgroup per spectrum: y w and w, zero padded to ldf channels, and A = sum y^2 w
__global__ __launch_bounds__(256) void k_muse_filter_prepare(const double *__restrict__ Y, const double *__restrict__ W, int ld, int nx,
                                                            double *__restrict__ YW, double *__restrict__ WF, int ldf,
                                                            double *__restrict__ A)
{
	__shared__ double part[4];
	const size_t r = blockIdx.x;
	double acc = 0.0;
	for (int j = threadIdx.x; j < ldf; j += 256) {
		const double y = j < nx ? Y[r * ld + j] : 0.0, w = j < nx ? W[r * ld + j] : 0.0;
		const double yw = y * w;
		YW[r * ldf + j] = yw;
		if (WF) WF[r * ldf + j] = w;
		acc = fma(y, yw, acc);
	}
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) A[r] = (part[0] + part[1]) + (part[2] + part[3]);
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
	const int ng = ldf >> 4;
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

bool muse_filter_prepare(mdns_spectra *s)
{
	if (s->d_fyw) return true;
	Context *c = ctx();
	const int ldf = (s->nx + 15) & ~15;
	const size_t elems = (size_t) s->ndata * ldf;
	double *yw = nullptr, *wf = nullptr, *a = nullptr;
	const bool own_w = ldf != s->ld;
	if (!MDNS_HIP(hipMalloc((void **) &yw, (elems ? elems : 1) * sizeof(double))) ||
	    !MDNS_HIP(hipMalloc((void **) &a, (size_t) (s->ndata ? s->ndata : 1) * sizeof(double))) ||
	    (own_w && !MDNS_HIP(hipMalloc((void **) &wf, (elems ? elems : 1) * sizeof(double))))) {
		if (yw) (void) hipFree(yw);
		if (a) (void) hipFree(a);
		return false;
	}
	if (s->ndata > 0)
		hipLaunchKernelGGL(k_muse_filter_prepare, dim3(s->ndata), dim3(256), 0, c->stream, (const double *) s->d_y, (const double *) s->d_w,
		                   s->ld, s->nx, yw, own_w ? wf : (double *) nullptr, ldf, a);
	if (!MDNS_HIP(hipGetLastError())) { (void) hipFree(yw); (void) hipFree(a); if (wf) (void) hipFree(wf); return false; }
	s->d_fyw = yw; s->d_fw = own_w ? wf : s->d_w; s->fw_owned = own_w; s->d_fa = a; s->ldf = ldf;
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
	// waves over the channels of a tile: as many as it takes to give every SIMD a few waves
	const long long tiles = (long long) rt * bt;
	int kw = tiles * 4 >= 3LL * 4 * c->num_cus ? 4 : 8;
	if (kw_forced) { const int f = atoi(kw_forced); if (f == 4 || f == 8) kw = f; }
	while (kw > 4 && (s->ldf >> 4) < 2 * kw) kw >>= 1;
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

extern "C" void mdns_muse_filter_mode(int mode) { mdns::g_filter_mode = mode == 0 || mode == 1 ? mode : -1; }

extern "C" void mdns_muse_filter_stats(long long *out4)
{
	if (!out4) return;
	for (int t = 0; t < 4; t++) out4[t] = mdns::g_filter_stats[t];
}
