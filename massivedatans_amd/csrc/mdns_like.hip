// Likelihood kernels for gfx950 (MI355X): spectral templates and the per-spectrum
// residual reductions that replace the reference's clike.c / cmuselike.c loops.
//
// Data layout in HBM: spectra are rows, Y[ndata][ld] (ld = nx rounded up to even, so every
// row starts 16-byte aligned and lanes read double2 = 16 B: a wave reads 1 KiB contiguous per
// load instruction).  Templates ("models") are M[B][ldm], ldm a multiple of 512 and
// zero-padded past nx, so lanes beyond the last channel contribute (0-0)^2 = 0.
//
// K1 (gauss rows): one wavefront scores R spectra at a time.  Lane l owns channel pairs
//   {p*128 + 2l, p*128 + 2l + 1}, p < NP; the spectrum values stay in registers while the
//   wave walks over the candidates in tiles of BT, so each spectrum is fetched from HBM
//   exactly once per launch however many candidates are scored.  Partial sums are reduced
//   across the 64 lanes with shuffles and written as log-likelihoods.
// K2 (muse rows): one 256-thread workgroup per spectrum (rows are 32 KiB at 4096 channels);
//   y and 1/v stay in registers over both passes of cmuselike.c:50-61 (scale, then chi), so
//   the reference's two strided passes become one HBM pass.
#include "mdns_internal.h"
#include <climits>
#include <cstdlib>

namespace mdns {

static constexpr int kBlock = 256;

// Sum over the 64 lanes of a wavefront, returned in every lane (wave-uniform).
// Data-parallel primitives (DPP) move the partial sums inside the VALU -- no LDS crossbar
// round trips as with ds_bpermute shuffles, which made reductions the longest part of the row
// kernels.  Steps: the two quad permutations, row_shr:4, row_shr:8 (each 16-lane row now has its
// sum in lanes 12-15), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lane 63
// holds the total.  Lanes without a source receive 0 (bound_ctrl), the identity of the sum.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
	return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
	v += dpp_move<0xb1, 0xf>(v);      // quad_perm:[1,0,3,2]
	v += dpp_move<0x4e, 0xf>(v);      // quad_perm:[2,3,0,1]
	v += dpp_move<0x114, 0xf>(v);     // row_shr:4
	v += dpp_move<0x118, 0xf>(v);     // row_shr:8
	v += dpp_move<0x142, 0xa>(v);     // row_bcast:15 -> rows 1, 3
	v += dpp_move<0x143, 0xc>(v);     // row_bcast:31 -> rows 2, 3
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
	return __hiloint2double(hi, lo);
}

// Several sums at once (gfx950): v_permlane32_swap / v_permlane16_swap exchange half-waves and
// 16-lane rows between TWO registers, so one swap + one add halves two sums together instead of
// one.  a' = swap32(a, b) gives  lanes 0-31: a_i + a_{i+32},  lanes 32-63: the same for b;
// swap16 does it again on rows.  Two values cost 18 instructions instead of 36, four cost 21
// instead of 72 (per wave, doubles: every move is two 32-bit moves).
__device__ __forceinline__ double swap_add32(double a, double b)
{
	const auto lo = __builtin_amdgcn_permlane32_swap((unsigned) __double2loint(a), (unsigned) __double2loint(b), false, false);
	const auto hi = __builtin_amdgcn_permlane32_swap((unsigned) __double2hiint(a), (unsigned) __double2hiint(b), false, false);
	return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b)
{
	const auto lo = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(a), (unsigned) __double2loint(b), false, false);
	const auto hi = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(a), (unsigned) __double2hiint(b), false, false);
	return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}
// sum inside every 16-lane row; the row total ends in its lanes 12-15
__device__ __forceinline__ double row_sum(double v)
{
	v += dpp_move<0xb1, 0xf>(v);
	v += dpp_move<0x4e, 0xf>(v);
	v += dpp_move<0x114, 0xf>(v);
	v += dpp_move<0x118, 0xf>(v);
	return v;
}
// wave totals of a and b: a's in lane 31, b's in lane 63 of the returned register
__device__ __forceinline__ double wave_sums2(double a, double b)
{
	double v = row_sum(swap_add32(a, b));         // rows 0,1: a   rows 2,3: b
	v += dpp_move<0x142, 0xa>(v);                 // row_bcast:15 -> rows 1 and 3 take rows 0 and 2
	return v;
}
// wave totals of a, b, c, d in lanes 15, 47, 31, 63 of the returned register
__device__ __forceinline__ double wave_sums4(double a, double b, double c, double d)
{
	return row_sum(swap_add16(swap_add32(a, b), swap_add32(c, d)));   // rows: a, c, b, d
}

// ---------------------------------------------------------------------------------------
// templates
// ---------------------------------------------------------------------------------------

// single Gaussian line, clike.c:65:  A * exp(-0.5 * ((mu - x_j)/sig)^2)
__global__ void k_gauss_model(const double *__restrict__ x, int nx, const double *__restrict__ params,
                              double *__restrict__ model, int ldm)
{
	const int b = blockIdx.y;
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= ldm) return;
	double m = 0.0;
	if (j < nx) {
		const double A = params[3 * b], mu = params[3 * b + 1], sig = params[3 * b + 2];
		const double t = (mu - x[j]) / sig;
		m = A * exp(-0.5 * (t * t));
	}
	model[(size_t) b * ldm + j] = m;
}

// the same line in candidate tiles: MT[tile][j][BT] with tile = b / BT, zero for b >= B and
// for the padding channels j >= nx (one thread per element).  A wave of k_gauss_cols walks one
// tile front to back, so its template stream is contiguous.
__global__ void k_gauss_model_t(const double *__restrict__ x, int nx, int nxp, const double *__restrict__ params,
                                int B, int bt_size, int ntile, double *__restrict__ model_t,
                                int *__restrict__ zero, int nzero)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	// first kernel of a draw chunk: also clears the accept flags and the result header
	if (e < nzero) zero[e] = 0;
	if (e >= ntile * nxp * bt_size) return;
	const int bin = e % bt_size;
	const int j = (e / bt_size) % nxp;
	const int b = (e / (bt_size * nxp)) * bt_size + bin;
	double m = 0.0;
	if (b < B && j < nx) {
		const double A = params[3 * b], mu = params[3 * b + 1], sig = params[3 * b + 2];
		const double t = (mu - x[j]) / sig;
		m = A * exp(-0.5 * (t * t));
	}
	model_t[e] = m;
}

// k_gauss_model_t for the filter: one workgroup per candidate tile writes the tile and the sum of
// squares of every candidate's template (deterministic: per-thread partial sums over the channels
// 256 / BT apart, then one thread per candidate adds them in a fixed order)
// (gridDim.y workgroups share a tile's channels: sixteen workgroups of thirteen exponentials per thread took 9 us at 256
// templates x 200 channels.  Each leaves its partial sums of squares; the last one of a tile to finish -- hand-over of
// mdns_internal.h -- adds them in the order of the channel shares, so msq does not depend on who was last)
__global__ __launch_bounds__(256) void k_gauss_model_tsq(const double *__restrict__ x, int nx, int nxp, const double *__restrict__ params,
                                                         int B, int bt_size, double *__restrict__ model_t, double *__restrict__ msq,
                                                         int *__restrict__ zero, int nzero, double *__restrict__ model_g, int nxg,
                                                         double *__restrict__ msq_part, int *__restrict__ tickets)
{
	__shared__ double partial[256];
	__shared__ int s_last;
	const int tile = blockIdx.x, t = threadIdx.x, ny = gridDim.y, share = blockIdx.y;
	for (int e = (share * gridDim.x + tile) * 256 + t; e < nzero; e += gridDim.x * ny * 256) zero[e] = 0;      // accept flags + result header
	const int bin = t % bt_size, b = tile * bt_size + bin;
	double A = 0.0, mu = 0.0, sig = 1.0;
	if (b < B) { A = params[3 * b]; mu = params[3 * b + 1]; sig = params[3 * b + 2]; }
	double acc = 0.0;
	const int nj = model_g && nxg > nxp ? nxg : nxp;
	const int per = (nj + ny - 1) / ny, j0 = share * per, j1 = j0 + per < nj ? j0 + per : nj;       // this workgroup's channels
	for (int j = j0 + t / bt_size; j < j1; j += 256 / bt_size) {
		double m = 0.0;
		if (b < B && j < nx) {
			const double u = (mu - x[j]) / sig;
			m = A * exp(-0.5 * (u * u));
		}
		if (j < nxp) model_t[((size_t) tile * nxp + j) * bt_size + bin] = m;
		// (the same value in tiles of 16 candidates, channel pair by channel pair: k_gauss_gemm_filter; bt_size is 16 then)
		if (model_g && j < nxg) model_g[tiled16_at((size_t) tile * 16 + bin, j, nxg >> 1)] = m;
		acc = fma(m, m, acc);
	}
	partial[t] = acc;
	__syncthreads();
	double mine = 0.0;
	if (t < bt_size) for (int q = t; q < 256; q += bt_size) mine += partial[q];
	if (ny == 1) { if (t < bt_size) msq[tile * bt_size + t] = mine; return; }
	if (t < bt_size) __hip_atomic_store(&msq_part[((size_t) tile * ny + share) * bt_size + t], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	handover_release();
	__syncthreads();
	if (t == 0) {
		const int done = atomicAdd(&tickets[tile], 1);
		s_last = done == ny - 1 ? 1 : 0;
		if (s_last) tickets[tile] = 0;                                   // for the next launch (stream order)
	}
	__syncthreads();
	if (!s_last || t >= bt_size) return;
	handover_acquire();
	double sum = 0.0;
	for (int sh = 0; sh < ny; sh++) sum += __hip_atomic_load(&msq_part[((size_t) tile * ny + sh) * bt_size + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	msq[tile * bt_size + t] = sum;
}

// rows (d_rows: which, or the first M) of Y [.][ld] -> tiled16 replica (zero padded rows and channels); one workgroup per tile
__global__ __launch_bounds__(256) void k_tile_rows16(const double *__restrict__ Y, int ld, int M, int nx, int nxg,
                                                     const int *__restrict__ rows, double *__restrict__ out)
{
	const int tile = blockIdx.x, ncp = nxg >> 1;
	for (int e = threadIdx.x; e < 16 * nxg; e += 256) {
		const int i = e / nxg, c = e - i * nxg;                           // (reads run along a row)
		const int k = tile * 16 + i;
		double v = 0.0;
		if (k < M && c < nx) v = Y[(size_t) (rows ? rows[k] : k) * ld + c];
		out[tiled16_at((size_t) k, c, ncp)] = v;
	}
}

// three lines on a flat continuum (config C5; massivedatans_amd/gen.py muse_template)
// (four channels per thread: the two 10**p of a candidate are the larger part of a thread's work -- one channel
// per thread measured 7.1 us per launch of 57 templates x 4096 channels)
static constexpr int kMuseModelPer = 4;
__global__ void k_muse3_model(const double *__restrict__ x, int nx, const double *__restrict__ params,
                              double *__restrict__ model, int ldm)
{
	const int b = blockIdx.y;
	const int j0 = blockIdx.x * (int) blockDim.x * kMuseModelPer + threadIdx.x;
	if (j0 >= ldm) return;
	const double *p = params + 5 * b;
	const double amp = pow(10.0, p[0]), z = p[1], ws = pow(10.0, p[2]);
	const double ratio[3] = {p[3], 1.0, p[4]};
	const double mu0[3] = {4861.3, 5006.8, 6562.8};
	const double a0[3] = {0.35, 1.0, 0.8};
	const double sg[3] = {4.0, 4.0, 5.0};
#pragma unroll
	for (int i = 0; i < kMuseModelPer; i++) {
		const int j = j0 + i * (int) blockDim.x;
		if (j >= ldm) break;
		double m = 0.0;
		if (j < nx) {
			const double xj = x[j];
			m = 1.0;
#pragma unroll
			for (int g = 0; g < 3; g++) {
				const double t = (xj - mu0[g] * (1 + z)) / (sg[g] * ws);
				m = m + amp * ratio[g] * a0[g] * exp(-0.5 * (t * t));
			}
		}
		model[(size_t) b * ldm + j] = m;
	}
}

// caller-supplied templates [B][nx] -> zero padded [B][ldm]
__global__ void k_pad_model(const double *__restrict__ src, int nx, double *__restrict__ dst, int ldm)
{
	const int b = blockIdx.y;
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= ldm) return;
	dst[(size_t) b * ldm + j] = j < nx ? src[(size_t) b * nx + j] : 0.0;
}

// ---------------------------------------------------------------------------------------
// re-laying spectra at upload
// ---------------------------------------------------------------------------------------

// src [nx][ndata] (reference layout, sample.py:31 / clike.c:72) -> dst [ndata][ld]
__global__ void k_transpose(const double *__restrict__ src, int nx, int ndata, double *__restrict__ dst,
                            int ld, int invert, int lds)
{
	__shared__ double tile[32][33];
	const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
	for (int jj = ty; jj < 32; jj += 8) {
		const int j = j0 + jj, i = i0 + tx;
		if (j < nx && i < ndata) tile[jj][tx] = src[(size_t) j * lds + i];
	}
	__syncthreads();
	for (int ii = ty; ii < 32; ii += 8) {
		const int i = i0 + ii, j = j0 + tx;
		if (i < ndata && j < nx) {
			const double val = tile[tx][ii];
			dst[(size_t) i * ld + j] = invert ? 1.0 / val : val;
		}
	}
}

// spectra rows Y[ndata][ld] -> the tiled channel-major replica of k_gauss_cols:
// element (channel j, spectrum i) at ((i / 64) * nxp + j) * 64 + i % 64, zeros in the padding
// (spectra >= ndata of the last tile, channels nx..nxp-1).  One workgroup = 64 spectra x 32
// channels through an LDS tile: reads run along channels, writes along spectra.
// With `rows` (int32[ndata]) spectrum i of the replica is row rows[i] of Y: the compact replica
// of a selection.
__global__ __launch_bounds__(kBlock) void k_tile_columns(const double *__restrict__ Y, int ld, int ndata, int nx,
                                                         int nxp, const int *__restrict__ rows,
                                                         double *__restrict__ YT)
{
	__shared__ double tile[64][33];
	const int t = blockIdx.x, j0 = blockIdx.y * 32;
	for (int r = threadIdx.x >> 5; r < 64; r += 8) {
		const int i = t * 64 + r, j = j0 + (threadIdx.x & 31);
		const int src = (rows && i < ndata) ? rows[i] : i;
		tile[r][threadIdx.x & 31] = (i < ndata && j < nx) ? Y[(size_t) src * ld + j] : 0.0;
	}
	__syncthreads();
	for (int jj = threadIdx.x >> 6; jj < 32; jj += 4) {
		const int j = j0 + jj;
		if (j < nxp) YT[((size_t) t * nxp + j) * 64 + (threadIdx.x & 63)] = tile[threadIdx.x & 63][jj];
	}
}

__global__ void k_copy_rows(const double *__restrict__ src, int nx, int ndata, double *__restrict__ dst,
                            int ld, int invert)
{
	const size_t n = (size_t) nx * ndata;
	for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t) gridDim.x * blockDim.x) {
		const size_t i = e / nx, j = e % nx;
		const double val = src[e];
		dst[i * ld + j] = invert ? 1.0 / val : val;
	}
}

// ---------------------------------------------------------------------------------------
// K1: Gaussian-line rows
// ---------------------------------------------------------------------------------------
// NP  = channel pairs per lane (nx <= 128*NP);  BT = candidates per register tile;
// R   = spectra in flight per wave;  HOIST = all candidates fit one tile, keep it in registers
template <int NP, int BT, int R, bool HOIST>
__global__ __launch_bounds__(kBlock) void k_gauss_rows(
    const double *__restrict__ Y, int ld, int nx, const double *__restrict__ model, int ldm, int B,
    double scale, const int *__restrict__ rows, int M, double *__restrict__ out)
{
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
	const int nwaves = (gridDim.x * kBlock) >> 6;
	const int ch = 2 * lane;

	bool valid[NP];
#pragma unroll
	for (int p = 0; p < NP; p++) valid[p] = (p * 128 + ch) < nx;

	double2 m[BT][NP];
	if (HOIST) {
#pragma unroll
		for (int b = 0; b < BT; b++) {
			const int bb = b < B ? b : B - 1;
#pragma unroll
			for (int p = 0; p < NP; p++)
				m[b][p] = *reinterpret_cast<const double2 *>(model + (size_t) bb * ldm + p * 128 + ch);
		}
	}

	for (int k0 = wave * R; k0 < M; k0 += nwaves * R) {
		double2 y[R][NP];
#pragma unroll
		for (int r = 0; r < R; r++) {
			const int k = (k0 + r < M) ? k0 + r : M - 1;
			const int row = rows ? rows[k] : k;
			const double *yr = Y + (size_t) row * ld + ch;
#pragma unroll
			for (int p = 0; p < NP; p++)
				y[r][p] = valid[p] ? *reinterpret_cast<const double2 *>(yr + p * 128) : make_double2(0.0, 0.0);
		}
		for (int b0 = 0; b0 < B; b0 += BT) {
			if (!HOIST) {
#pragma unroll
				for (int b = 0; b < BT; b++) {
					const int bb = (b0 + b < B) ? b0 + b : B - 1;
#pragma unroll
					for (int p = 0; p < NP; p++)
						m[b][p] = *reinterpret_cast<const double2 *>(model + (size_t) bb * ldm + p * 128 + ch);
				}
			}
			double acc[BT][R];
#pragma unroll
			for (int b = 0; b < BT; b++)
#pragma unroll
				for (int r = 0; r < R; r++) {
					double a = 0.0;
#pragma unroll
					for (int p = 0; p < NP; p++) {
						const double d0 = m[b][p].x - y[r][p].x;
						const double d1 = m[b][p].y - y[r][p].y;
						a = fma(d0, d0, a);
						a = fma(d1, d1, a);
					}
					acc[b][r] = a;
				}
			// reduce over the 64 lanes; lane (b*R + r) keeps value (b, r) so that the
			// BT*R results leave in one store instruction
			double mine = 0.0;
#pragma unroll
			for (int b = 0; b < BT; b++)
#pragma unroll
				for (int r = 0; r < R; r++) {
					const double s = wave_sum(acc[b][r]);
					if (lane == b * R + r) mine = s;
				}
			if (lane < BT * R) {
				const int b = b0 + lane / R, k = k0 + lane % R;
				if (b < B && k < M) out[(size_t) b * M + k] = mine * scale;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------
// K1, dense selections: one LANE per spectrum on the channel-major replica
// ---------------------------------------------------------------------------------------
// YT[nx][ldT]: lane i of a wave reads YT[j][tile*64 + i] -- 512 contiguous bytes per load and
// no cross-lane reduction at all: every lane sums its own spectrum over the channels in
// ascending order (the order of clike.c:64-76).  A wave scores BT candidates; their template
// values MT[j][bt*BT .. +BT) are wave-uniform and arrive through the scalar cache as SGPR
// operands, so the inner loop is exactly one v_add_f64 + one v_fma_f64 per (candidate,
// channel, spectrum).  Work items are (spectrum tile, candidate tile) pairs, one per wave.
// SP = spectra per lane (1 or 2): with 2 every template value read through the scalar cache
// feeds two v_add/v_fmac pairs.
// Which (spectrum tile, candidate tile) a wave works on -- see the comments in k_gauss_cols.
// Returns false when the wave has nothing to do.
__device__ __forceinline__ bool cols_item(int ntiles, int tile_limit, int nq_xcd, int nbt, int cu_slots, int &tile, int &bt)
{
	// A workgroup takes 4 adjacent spectrum tiles (a "quad") of ONE candidate tile, so its
	// waves pull the same template values through the scalar cache.  Workgroups are dealt
	// round-robin over the 8 XCDs (blockIdx % 8 shares an XCD): quad q is always given to XCD
	// q % 8, for every candidate tile, so each XCD re-reads only its own eighth of the spectra
	// from its own L2.  This is a speed-only mapping; any placement gives the same results.
	const int xcd = blockIdx.x & 7;
	const int local = blockIdx.x >> 3;
	// Within an XCD the dispatcher deals workgroups round-robin over the CUs (measured: `local`
	// and `local + cu_slots` always share a CU), so slot = local % cu_slots names a CU.  Each
	// slot takes a contiguous run of the XCD's items in candidate-tile-major order: the
	// workgroups resident on a CU then read the SAME template tile through the scalar cache
	// (12.8 KB at BT = 8) instead of five different ones (-12 % at B = 256).
	const int slot = local % cu_slots, round = local / cu_slots;
	const int nitems = nbt * nq_xcd;
	const int first = (int) ((long long) slot * nitems / cu_slots);
	const int next = (int) ((long long) (slot + 1) * nitems / cu_slots);
	if (first + round >= next) return false;
	bt = (first + round) / nq_xcd;
	const int quad = ((first + round) % nq_xcd) * 8 + xcd;
	const int wpb = blockDim.x >> 6;                          // waves (spectrum tiles) per workgroup: 4
	tile = quad * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	(void) ntiles;
	return tile < tile_limit;         // (INT_MAX: the caller has a workgroup barrier to reach first)
}

// The sums of one wave: acc[s][b] = sum over the channels, in ascending order, of
// (template b - spectrum)^2 for the lane's spectrum (positions k[s] of the selection).
// MS = distance in doubles between the template values of consecutive channels: BT for a
// wave that scores a whole candidate tile (compile time), or, with RUNTIME_STRIDE, the tile
// width the templates were laid out for while the wave scores ONE candidate of it (BT == 1).
// Every (candidate, spectrum) sum is the same chain of v_add_f64 / v_fma_f64 whatever BT is,
// so the likelihood of a pair does not depend on the tile shape it was computed in.
template <int BT, int SP, bool RUNTIME_STRIDE, bool DOT = false>
__device__ __forceinline__ void cols_accumulate(const double *__restrict__ YT, int nxp,
                                                const double *mp, int mstride,
                                                const int *__restrict__ rows, int M, int tile, int lane,
                                                int (&k)[SP], double (&acc)[SP][BT])
{
	constexpr int CH = 8;                     // channels per software-pipeline stage (nxp % CH == 0)
	// YT is stored in tiles of 64 spectra: element (channel j, spectrum i) at
	// ((i / 64) * nxp + j) * 64 + i % 64.  Whatever column a lane owns (its own tile when all
	// spectra are selected, a gathered one otherwise), consecutive channels are exactly 512
	// bytes apart, so the eight loads of a stage are immediate offsets of one pointer.
	const double *yp[SP];
#pragma unroll
	for (int s = 0; s < SP; s++) {
		k[s] = (tile * SP + s) * 64 + lane;
		int col = k[s];                       // the replica is padded: positions >= M stay readable
		if (rows) col = rows[k[s] < M ? k[s] : M - 1];
		else if (col >= ((M + 63) & ~63)) col = M - 1;
		yp[s] = YT + ((size_t) (col >> 6) * nxp << 6) + (col & 63);
	}
#pragma unroll
	for (int s = 0; s < SP; s++)
#pragma unroll
		for (int b = 0; b < BT; b++) acc[s][b] = 0.0;
	// Software pipeline over stages of CH channels with two register buffers that swap roles
	// (no copies): while `cur` is consumed the next stage's spectra values land in `nxt`.
	// The last stage prefetches its own channels again (in bounds, unused) so that every
	// stage is the same straight-line code.
	double ya[SP][CH], yb[SP][CH];
#pragma unroll
	for (int s = 0; s < SP; s++)
#pragma unroll
		for (int c = 0; c < CH; c++) ya[s][c] = yp[s][c * 64];
	const int ms = RUNTIME_STRIDE ? mstride : BT;
	auto stage = [&](const double (&cur)[SP][CH], double (&nxt)[SP][CH], bool last) {
#pragma unroll
		for (int s = 0; s < SP; s++) {
			yp[s] += last ? 0 : CH * 64;
#pragma unroll
			for (int c = 0; c < CH; c++) nxt[s][c] = yp[s][c * 64];
		}
#pragma unroll
		for (int c = 0; c < CH; c++) {
			// (forcing all differences of a channel ahead of the squares was measured 9 % slower:
			// it costs 34 VGPRs = three waves per SIMD)
#pragma unroll
			for (int b = 0; b < BT; b++) {
				const double mv = mp[c * ms + b];
#pragma unroll
				for (int s = 0; s < SP; s++) {
					if constexpr (DOT) {
						// the cross term of the expanded square only: ONE v_fmac_f64 with the template
						// value as scalar operand per (candidate, channel, spectrum)
						acc[s][b] = fma(mv, cur[s][c], acc[s][b]);
					} else {
						const double d = mv - cur[s][c];
						acc[s][b] = fma(d, d, acc[s][b]);
					}
				}
			}
		}
		mp += CH * ms;
	};
	int st = nxp / CH;
#pragma unroll 1
	for (; st >= 2; st -= 2) {
		stage(ya, yb, false);
		stage(yb, ya, st == 2);
	}
	if (st == 1) stage(ya, yb, true);
}

// The same sums for launches too small to hide memory latency behind other waves (a draw over
// a few hundred spectra, the one-candidate commit pass).  Two changes, none to the arithmetic
// (same operations in the same order per (candidate, spectrum): the same bits as
// cols_accumulate): a ring of NB stage buffers keeps NB - 1 stages of spectra values in flight
// instead of one, and the template values come from an LDS copy of the candidate tile
// (`tpl`, [channel][BT], filled by the workgroup before the loop) as broadcast ds_read_b64 -- the
// scalar loads of the big kernel expose their full latency once per stage when nothing else
// runs on the SIMD (measured on the constrained draws of a real run: 29 us per launch of which
// 3 are arithmetic).
template <int BT, int NB>
__device__ __forceinline__ void cols_accumulate_deep(const double *__restrict__ YT, int nxp,
                                                     const double *tpl,
                                                     const int *__restrict__ rows, int M, int tile, int lane,
                                                     int &k, double (&acc)[BT])
{
	constexpr int CH = 8;
	k = tile * 64 + lane;
	int col = k;
	if (rows) col = rows[k < M ? k : M - 1];
	else if (col >= ((M + 63) & ~63)) col = M - 1;
	const double *yp = YT + ((size_t) (col >> 6) * nxp << 6) + (col & 63);
#pragma unroll
	for (int b = 0; b < BT; b++) acc[b] = 0.0;
	const int nst = nxp / CH;
	double y[NB][CH];
#pragma unroll
	for (int i = 0; i < NB - 1; i++) {
		const double *p = yp + (size_t) min(i, nst - 1) * CH * 64;
#pragma unroll
		for (int c = 0; c < CH; c++) y[i][c] = p[c * 64];
	}
#pragma unroll 1
	for (int s0 = 0; s0 < nst; s0 += NB) {
#pragma unroll
		for (int i = 0; i < NB; i++) {
			const int s = s0 + i;                                // wave-uniform
			if (s < nst) {
				const double *p = yp + (size_t) min(s + NB - 1, nst - 1) * CH * 64;
#pragma unroll
				for (int c = 0; c < CH; c++) y[(i + NB - 1) % NB][c] = p[c * 64];
				const double *m = tpl + s * CH * BT;
#pragma unroll
				for (int c = 0; c < CH; c++)
#pragma unroll
					for (int b = 0; b < BT; b++) {
						const double d = m[c * BT + b] - y[i][c];
						acc[b] = fma(d, d, acc[b]);
					}
			}
		}
	}
}

// copies n template values from global memory (stride `stride` doubles apart) into LDS, all of a
// thread's loads in flight at once; ends with the workgroup barrier
__device__ __forceinline__ void stage_templates(double *__restrict__ dst, const double *__restrict__ src, int n, int stride)
{
	constexpr int U = 4;
	for (int base = 0; base < n; base += U * (int) blockDim.x) {
		double r[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const int e = base + u * (int) blockDim.x + (int) threadIdx.x;
			r[u] = src[(size_t) (e < n ? e : n - 1) * stride];
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			const int e = base + u * (int) blockDim.x + (int) threadIdx.x;
			if (e < n) dst[e] = r[u];
		}
	}
	__syncthreads();
}

template <int BT, int SP>
__global__ __launch_bounds__(kBlock) void k_gauss_cols(
    const double *__restrict__ YT, int nxp, const double *__restrict__ model_t, int B,
    double scale, const int *__restrict__ rows, int M, int ntiles, int nq_xcd, int nbt, int cu_slots, double *__restrict__ out)
{
	const int lane = threadIdx.x & 63;
	// Work item of a wave = (spectrum tile, candidate tile), a spectrum tile being 64*SP
	// spectra, one per workgroup wave (cols_item).
	int tile, bt;
	if (!cols_item(ntiles, ntiles, nq_xcd, nbt, cu_slots, tile, bt)) return;
	const double *mp = model_t + (size_t) bt * nxp * BT;     // wave-uniform: CH*BT contiguous doubles per stage
	int k[SP];                                // positions in the (compacted) output
	double acc[SP][BT];
	cols_accumulate<BT, SP, false>(YT, nxp, mp, BT, rows, M, tile, lane, k, acc);
#pragma unroll
	for (int s = 0; s < SP; s++) {
		if (k[s] < M) {
#pragma unroll
			for (int b = 0; b < BT; b++)
				if (bt * BT + b < B) out[(size_t) (bt * BT + b) * M + k[s]] = acc[s][b] * scale;
		}
	}
}

// The same sums with the accept test of the constrained draw as epilogue
// (hiermetriclearn.py:193 `any(L > Lmins)`): nothing but one flag per candidate leaves the
// kernel.  `thr_rows` names the data set behind every position of the selection (NULL: the
// position itself), `higher` holds the thresholds of all data sets (multi_nested_sampler.py:438-447).
// DEEP: the launch has too few waves to hide latency (see cols_accumulate_deep)
template <int BT, bool DEEP>
__global__ __launch_bounds__(2 * kBlock) void k_gauss_cols_accept(
    const double *__restrict__ YT, int nxp, const double *__restrict__ model_t, int B,
    double scale, const int *__restrict__ rows, const int *__restrict__ thr_rows, int M,
    int ntiles, int nq_xcd, int nbt, int cu_slots, const double *__restrict__ higher, int *__restrict__ flags,
    JointTrail trail)
{
	const int lane = threadIdx.x & 63;
	int tile, bt;
	int k[1];
	double acc[1][BT];
	if constexpr (DEEP) {
		extern __shared__ __attribute__((aligned(16))) double tpl[];      // [nxp][BT]
		const bool mine = cols_item(ntiles, INT_MAX, nq_xcd, nbt, cu_slots, tile, bt);   // false for whole workgroups only
		if (!mine) return;
		stage_templates(tpl, model_t + (size_t) bt * nxp * BT, nxp * BT, 1);
		if (tile >= ntiles) return;
		cols_accumulate_deep<BT, 4>(YT, nxp, tpl, rows, M, tile, lane, k[0], acc[0]);
	} else {
		if (!cols_item(ntiles, ntiles, nq_xcd, nbt, cu_slots, tile, bt)) return;
		const double *mp = model_t + (size_t) bt * nxp * BT;
		cols_accumulate<BT, 1, false>(YT, nxp, mp, BT, rows, M, tile, lane, k, acc);
	}
	const bool live = k[0] < M;
	const int kk = live ? k[0] : M - 1;
	// a quiet NaN compares false with everything: lanes past the selection never vote
	const double thr = live ? higher[thr_rows ? thr_rows[kk] : kk] : __builtin_nan("");
#pragma unroll
	for (int b = 0; b < BT; b++) {
		const double L = acc[0][b] * scale;
		const unsigned long long word = __ballot(L > thr);
		if (word != 0ull && bt * BT + b < B) {                         // rare: a candidate some data set accepts
			const size_t at = (size_t) (bt * BT + b) * ntiles + tile;
			if (trail.stamp_of) trail.L[at * 64 + lane] = L;
			if (lane == 0) {
				flags[bt * BT + b] = 1;
				if (trail.stamp_of) { trail.word[at] = word; trail.stamp_of[at] = trail.stamp; }
			}
		}
	}
}

// The accept test as a GUARDED FILTER, for launches that are bound by vector issue (hundreds of
// candidates over thousands of spectra).  sum_j (m_j - y_j)^2 = sum m^2 - 2 sum m y + sum y^2: the
// first sum is a property of the candidate (`msq`, from the template kernel), the last one of the
// spectrum (`ysq`, computed once at upload), so the launch only has to accumulate the cross term --
// one v_fmac_f64 per (candidate, channel, spectrum) instead of v_add_f64 + v_fmac_f64.  The value
// so obtained, Lf, is NOT the chain value L the rest of the library works with (the expanded form
// cancels), but it lies within a bound that is cheap to compute:
//     |Lf - L| <= |scale| (nx + 8) 2^-52 (msq + 2 |sum m y| + ysq) =: E     (forward error of three
//     sums of nx terms and their combination, and of the chain itself)
// so  Lf > thr + 4 E  implies  L > thr  (flag the candidate: exactly the decision of the chain
// kernel) and  Lf < thr - 4 E  implies  L <= thr  (no vote).  For anything in between -- one pair in
// ~10^9 -- and for a candidate with a clear vote that may be THE accepted one (no lower candidate
// flagged yet) the wave computes the chain's own sums for that one candidate on its tile (1/BT of a
// chain pass, rare) and votes / leaves the trail exactly as k_gauss_cols_accept does.  Flags, the
// accepted index and every likelihood that is KEPT therefore equal the chain kernel's bit for bit.
template <int BT>
__global__ __launch_bounds__(2 * kBlock) void k_gauss_cols_filter(
    const double *__restrict__ YT, int nxp, int nx, const double *__restrict__ model_t, const double *__restrict__ msq, int B,
    double scale, const int *__restrict__ rows, const int *__restrict__ thr_rows, int M,
    int ntiles, int nq_xcd, int nbt, int cu_slots, const double *__restrict__ higher, const double *__restrict__ ysq,
    int *__restrict__ flags, JointTrail trail, int *__restrict__ lowest)
{
	const int lane = threadIdx.x & 63;
	int tile, bt;
	int k[1];
	double acc[1][BT];
	if (!cols_item(ntiles, ntiles, nq_xcd, nbt, cu_slots, tile, bt)) return;
	const double *mp = model_t + (size_t) bt * nxp * BT;
	cols_accumulate<BT, 1, false, true>(YT, nxp, mp, BT, rows, M, tile, lane, k, acc);
	const bool live = k[0] < M;
	const int kk = live ? k[0] : M - 1;
	const int d = thr_rows ? thr_rows[kk] : kk;
	const double thr = live ? higher[d] : __builtin_nan("");          // NaN compares false: no vote
	const double yy = ysq[d];
	const double unit = fabs(scale) * (double) (nx + 8) * 0x1p-52;
	unsigned hits = 0, maybes = 0;                                     // wave-uniform: one bit per candidate of the tile
#pragma unroll
	for (int b = 0; b < BT; b++) {
		const int cand = bt * BT + b;
		const double mm = msq[cand];
		const double Lf = scale * ((mm - 2.0 * acc[0][b]) + yy);
		const double E4 = 4.0 * unit * ((mm + 2.0 * fabs(acc[0][b])) + yy);
		const bool valid = cand < B;
		if (__ballot(valid && Lf > thr + E4) != 0ull) hits |= 1u << b;
		else if (__ballot(valid && Lf >= thr - E4) != 0ull) maybes |= 1u << b;
	}
	// the rule: nobody accepts any candidate of the tile and the wave is done.  Otherwise, candidate
	// by candidate (a rolled loop: the chain pass is inlined ONCE):
	unsigned todo = hits | maybes;
#pragma unroll 1
	while (todo) {
		const int b = __builtin_ctz(todo);
		todo &= todo - 1;
		const int cand = bt * BT + b;
		// `lowest` holds B - (lowest candidate index flagged so far), 0 = none (atomicMax): a candidate
		// above it cannot be the accepted one, so a clear vote for it needs no likelihoods
		const int seen = B - __hip_atomic_load(lowest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if ((hits >> b & 1u) && cand > seen) { if (lane == 0) flags[cand] = 1; continue; }
		// the chain's own sums for this candidate on this tile: the decision inside the band, and
		// the likelihoods the commit pass wants (the trail) for a candidate that may be THE one
		int k1[1];
		double a1[1][1];
		cols_accumulate<1, 1, true>(YT, nxp, mp + b, BT, rows, M, tile, lane, k1, a1);
		const double L = a1[0][0] * scale;
		const unsigned long long word = __ballot(L > thr);
		if (word != 0ull) {
			const size_t at = (size_t) cand * ntiles + tile;
			if (trail.stamp_of) trail.L[at * 64 + lane] = L;
			if (lane == 0) {
				flags[cand] = 1;
				atomicMax(lowest, B - cand);
				if (trail.stamp_of) { trail.word[at] = word; trail.stamp_of[at] = trail.stamp; }
			}
		}
	}
}

// sum of squares of every spectrum (the filter's third sum), one thread per spectrum on the rows
__global__ void k_row_sumsq(const double *__restrict__ Y, int ld, int nx, int ndata, double *__restrict__ out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ndata) return;
	const double *yr = Y + (size_t) i * ld;
	double acc = 0.0;
	for (int j = 0; j < nx; j++) acc = fma(yr[j], yr[j], acc);
	out[i] = acc;
}

// The commit when nobody wants the whole likelihood row: the first flagged candidate is THE
// accepted point (hiermetriclearn.py:193-196); what its data sets need -- who beats the threshold,
// and with which likelihood -- the accept pass left in the trail.  One wave per tile of 64
// positions of the selection, one lane per data set.  Same shelf / threshold update as
// k_gauss_cols_commit (multi_nested_sampler.py:482-485, :438-447).
__global__ __launch_bounds__(kBlock) void k_joint_commit_trail(
    const int *__restrict__ thr_rows, int M, int B, int ntiles, const int *__restrict__ flags, int flag_value, JointTrail trail,
    JointArrays st, JointHeader *__restrict__ header, unsigned long long *__restrict__ fillbits,
    JointMailbox *__restrict__ box, unsigned long long seq, int *__restrict__ ticket)
{
	__shared__ int s_first;
	__shared__ int s_last;
	if (threadIdx.x == 0) s_first = 0x7fffffff;
	__syncthreads();
	for (int b = threadIdx.x; b < B; b += kBlock)
		if (flags[b] == flag_value) { atomicMin(&s_first, b); break; }      // ascending per thread: its first is its lowest
	__syncthreads();
	const int bstar = s_first;
	if (blockIdx.x == 0 && threadIdx.x == 0) header->accepted = bstar < B ? bstar : -1;
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const bool mine = bstar < B && tile < ntiles;
	if (!mine && !box) return;
	const size_t at = mine ? (size_t) bstar * ntiles + tile : 0;
	// (a candidate flagged by another rank's data sets only has no entry here: nobody beats)
	const unsigned long long word = mine && trail.stamp_of[at] == trail.stamp ? trail.word[at] : 0ull;
	const int k = tile * 64 + lane;
	if (mine && k < M && (word >> lane & 1ull)) {
		const int d = thr_rows ? thr_rows[k] : k;
		const double L = trail.L[at * 64 + lane];
		const double thr = st.higher[d];
		const int n = st.shelfn[d];
		if (n >= st.cap) {
			atomicOr(&header->status, 1);
		} else {
			// With n waiting the threshold was the (n+1)-th smallest of live + shelf, and L lies
			// above it: the (n+2)-th smallest of the enlarged set is the old threshold again when
			// it occurs more than once, else the smaller of L and the next value above it.
			int at_most = 0;
			double next = INFINITY;
			int p = 0;
			for (; p + 25 <= st.nlive; p += 25) {                    // 25 loads in flight (the latency of a round trip, not the count, is what this pass costs)
				double v[25];
#pragma unroll
				for (int u = 0; u < 25; u++) v[u] = st.live[(size_t) (p + u) * st.ndata + d];
#pragma unroll
				for (int u = 0; u < 25; u++) { if (v[u] <= thr) at_most++; else next = fmin(next, v[u]); }
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
	if (!box) { if (lane == 0) fillbits[tile] = word; return; }
	// box != nullptr: the mailbox is filled by the last workgroup to finish (hand-over of mdns_internal.h: the fill
	// words go through agent-scope stores) -- one launch less than k_joint_publish behind this kernel
	if (mine && lane == 0) __hip_atomic_store(&fillbits[tile], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	handover_release();
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1) == (int) gridDim.x - 1 ? 1 : 0;
	__syncthreads();
	if (!s_last) return;
	handover_acquire();
	if (bstar < B)
		for (int w = threadIdx.x; w < ntiles; w += kBlock)
			mail_store(&box->bits[w], __hip_atomic_load(&fillbits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	handover_release();
	__syncthreads();
	if (threadIdx.x != 0) return;
	*ticket = 0;
	mail_store(&box->accepted, bstar < B ? bstar : -1);
	mail_store(&box->status, __hip_atomic_load(&header->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	mail_raise(&box->seq, seq);
}

// Second half of a draw chunk: the first flagged candidate is THE accepted point
// (hiermetriclearn.py:193-196).  Its likelihood row is computed again -- the same chain of
// operations as in k_gauss_cols_accept, so the same bits -- and compared with the thresholds
// per data set (multi_nested_sampler.py:482-485): the row, one fill bit per selected data set,
// and, for the data sets it beats, the append to their shelf and their next threshold.
//   live   [nlive][ndata]  live-point likelihoods, slot p of data set d at p * ndata + d
//   shelfL [cap][ndata]    likelihoods waiting, entry e of data set d at e * ndata + d
//   shelfn [ndata]         how many are waiting
//   higher [ndata]         threshold = (n+1)-th smallest of live + shelf with n waiting
// One wave = 64 positions of the selection, one lane per data set: no two lanes share state.
__global__ __launch_bounds__(kBlock) void k_gauss_cols_commit(
    const double *__restrict__ YT, int nxp, const double *__restrict__ model_t, int mstride, int B,
    double scale, const int *__restrict__ rows, const int *__restrict__ thr_rows, int M,
    int ntiles, int nq_xcd, int cu_slots, const int *__restrict__ flags, JointArrays st,
    JointHeader *__restrict__ header, unsigned long long *__restrict__ fillbits, double *__restrict__ Lrow)
{
	// all LDS in the dynamic region (a static variable in front of it would leave the template
	// column 4 bytes off its 8-byte alignment): [nxp] template doubles, then one int
	extern __shared__ __attribute__((aligned(16))) double tpl[];
	int &s_first = *reinterpret_cast<int *>(tpl + nxp);
	if (threadIdx.x == 0) s_first = 0x7fffffff;
	__syncthreads();
	for (int b = threadIdx.x; b < B; b += kBlock)
		if (flags[b]) { atomicMin(&s_first, b); break; }      // ascending per thread: its first is its lowest
	__syncthreads();
	const int bstar = s_first;
	if (blockIdx.x == 0 && threadIdx.x == 0) header->accepted = bstar < B ? bstar : -1;
	if (bstar >= B) return;
	const int lane = threadIdx.x & 63;
	int tile, bt_unused;
	if (!cols_item(ntiles, INT_MAX, nq_xcd, 1, cu_slots, tile, bt_unused)) return;      // whole workgroups only
	// templates are laid out [candidate tile][channel][mstride candidates]: the accepted
	// candidate's column goes to LDS
	stage_templates(tpl, model_t + (size_t) (bstar / mstride) * nxp * mstride + bstar % mstride, nxp, mstride);
	if (tile >= ntiles) return;
	int k[1];
	double acc[1][1];
	cols_accumulate_deep<1, 4>(YT, nxp, tpl, rows, M, tile, lane, k[0], acc[0]);
	const bool live = k[0] < M;
	const double L = acc[0][0] * scale;
	bool beats = false;
	if (live) {
		const int d = thr_rows ? thr_rows[k[0]] : k[0];
		Lrow[k[0]] = L;
		const double thr = st.higher[d];
		beats = L > thr;
		if (beats) {
			const int n = st.shelfn[d];
			if (n >= st.cap) {
				atomicOr(&header->status, 1);
			} else {
				// With n waiting the threshold was the (n+1)-th smallest of live + shelf, and L lies
				// above it: the (n+2)-th smallest of the enlarged set is the old threshold again when
				// it occurs more than once, else the smaller of L and the next value above it.
				int at_most = 0;
				double next = INFINITY;
				int p = 0;
				for (; p + 8 <= st.nlive; p += 8) {                  // eight loads in flight
					double v[8];
#pragma unroll
					for (int u = 0; u < 8; u++) v[u] = st.live[(size_t) (p + u) * st.ndata + d];
#pragma unroll
					for (int u = 0; u < 8; u++) { if (v[u] <= thr) at_most++; else next = fmin(next, v[u]); }
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
	}
	const unsigned long long word = __ballot(beats);
	if (lane == 0) fillbits[tile] = word;
}

// any nx: channels walked in chunks of 128, templates re-read per chunk (L2 resident)
template <int BT>
__global__ __launch_bounds__(kBlock) void k_gauss_rows_generic(
    const double *__restrict__ Y, int ld, int nx, const double *__restrict__ model, int ldm, int B,
    double scale, const int *__restrict__ rows, int M, double *__restrict__ out)
{
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
	const int nwaves = (gridDim.x * kBlock) >> 6;
	const int ch = 2 * lane;
	for (int k = wave; k < M; k += nwaves) {
		const int row = rows ? rows[k] : k;
		const double *yr = Y + (size_t) row * ld;
		for (int b0 = 0; b0 < B; b0 += BT) {
			double acc[BT];
#pragma unroll
			for (int b = 0; b < BT; b++) acc[b] = 0.0;
			for (int c0 = 0; c0 < nx; c0 += 128) {
				const bool ok = c0 + ch < nx;
				const double2 yv = ok ? *reinterpret_cast<const double2 *>(yr + c0 + ch) : make_double2(0.0, 0.0);
#pragma unroll
				for (int b = 0; b < BT; b++) {
					const int bb = (b0 + b < B) ? b0 + b : B - 1;
					const double2 mv = *reinterpret_cast<const double2 *>(model + (size_t) bb * ldm + c0 + ch);
					const double d0 = mv.x - yv.x, d1 = mv.y - yv.y;
					acc[b] = fma(d0, d0, acc[b]);
					acc[b] = fma(d1, d1, acc[b]);
				}
			}
#pragma unroll
			for (int b = 0; b < BT; b++) {
				const double s = wave_sum(acc[b]);
				if (lane == 0 && b0 + b < B) out[(size_t) (b0 + b) * M + k] = s * scale;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------
// K2: scale-marginalised chi^2 rows (cmuselike.c:45-64)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum4(double v, double *slot /* [4] in LDS */)
{
	v = wave_sum(v);
	if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
	__syncthreads();
	return (slot[0] + slot[1]) + (slot[2] + slot[3]);
}

// sums N values over the 256 threads with ONE barrier: wave shuffles, then 4 x N LDS words.
// `slot` [4][N] must not be in use by a reduction that other waves may still be reading.
// SWAPS: use the permlane-swap reductions (k_muse_rows2<8> sits at the 256-VGPR limit: with
// both of its reductions on swaps it ran 32 % slower, with only the four-value one 14 % faster)
template <int N, bool SWAPS = true>
__device__ __forceinline__ void block_sums(double (&v)[N], double *slot)
{
	const int wv = threadIdx.x >> 6;
	const int lane = threadIdx.x & 63;
	if constexpr (N == 4 && SWAPS) {
		const double t = wave_sums4(v[0], v[1], v[2], v[3]);
		// lanes 15, 31, 47, 63 hold the totals of v[0], v[2], v[1], v[3]
		if ((lane & 15) == 15) slot[wv * 4 + ((lane >> 5) | ((lane >> 3) & 2))] = t;
	} else if constexpr (N == 2 && SWAPS) {
		const double t = wave_sums2(v[0], v[1]);
		if ((lane & 31) == 31) slot[wv * 2 + (lane >> 5)] = t;
	} else {
#pragma unroll
		for (int i = 0; i < N; i++) {
			const double t = wave_sum(v[i]);
			if (lane == 0) slot[wv * N + i] = t;
		}
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < N; i++) v[i] = (slot[i] + slot[N + i]) + (slot[2 * N + i] + slot[3 * N + i]);
}

// NP = channel pairs per thread (nx <= 512*NP); CB = candidates per barrier pair; W holds 1/v.
// Per candidate and channel: s1 += (y w) m, s2 += (m m) w, then chi += (y - s m)^2 w -- six
// VALU operations, with y, w and y w resident in registers for all candidates of the row.
template <int NP, int CB>
__global__ __launch_bounds__(kBlock) void k_muse_rows(
    const double *__restrict__ Y, const double *__restrict__ W, int ld, int nx,
    const double *__restrict__ model, int ldm, int B, const int *__restrict__ rows, int M,
    double *__restrict__ out, int bchunk, MuseBandFused band)
{
	if (band.sc && band.status_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *band.status_zero = 0;
	// grid.y splits the candidates (chunks of `bchunk`, a multiple of CB) when there are too
	// few rows to fill the chip: a row is then loaded by several workgroups
	const int bbeg = blockIdx.y * bchunk;
	const int bend = min(B, bbeg + bchunk);
	__shared__ double redA[4 * 2 * CB], redB[4 * CB];
	const int ch = 2 * threadIdx.x;
	bool valid[NP];
#pragma unroll
	for (int p = 0; p < NP; p++) valid[p] = (p * 512 + ch) < nx;

	for (int k = blockIdx.x; k < M; k += gridDim.x) {
		const int row = rows ? rows[k] : k;
		const size_t base = (size_t) row * ld + ch;
		double2 y[NP], w[NP], yw[NP];
#pragma unroll
		for (int p = 0; p < NP; p++) {
			// lanes past the last channel hold zeros; for odd nx the pad channel of the last
			// pair has y = w = 0 too (the buffers are zero-filled before the upload)
			y[p] = valid[p] ? *reinterpret_cast<const double2 *>(Y + base + p * 512) : make_double2(0.0, 0.0);
			w[p] = valid[p] ? *reinterpret_cast<const double2 *>(W + base + p * 512) : make_double2(0.0, 0.0);
			yw[p] = make_double2(y[p].x * w[p].x, y[p].y * w[p].y);
		}
		for (int b0 = bbeg; b0 < bend; b0 += CB) {
			double2 m[CB][NP];
			double sums[2 * CB];
#pragma unroll
			for (int c = 0; c < CB; c++) {
				const int b = (b0 + c < B) ? b0 + c : B - 1;
				double s1 = 0.0, s2 = 0.0;
#pragma unroll
				for (int p = 0; p < NP; p++) {
					m[c][p] = *reinterpret_cast<const double2 *>(model + (size_t) b * ldm + p * 512 + ch);
					s1 = fma(yw[p].x, m[c][p].x, s1);
					s1 = fma(yw[p].y, m[c][p].y, s1);
					s2 = fma(m[c][p].x * m[c][p].x, w[p].x, s2);
					s2 = fma(m[c][p].y * m[c][p].y, w[p].y, s2);
				}
				sums[2 * c] = s1;
				sums[2 * c + 1] = s2;
			}
			block_sums<2 * CB>(sums, redA);
			double chi[CB];
#pragma unroll
			for (int c = 0; c < CB; c++) {
				const double s = sums[2 * c] / (1e-10 + sums[2 * c + 1]);      // cmuselike.c:52,57
				double acc = 0.0;
#pragma unroll
				for (int p = 0; p < NP; p++) {
					const double r0 = fma(-s, m[c][p].x, y[p].x);
					const double r1 = fma(-s, m[c][p].y, y[p].y);
					acc = fma(r0 * r0, w[p].x, acc);
					acc = fma(r1 * r1, w[p].y, acc);
				}
				chi[c] = acc;
			}
			// redA may be rewritten by the next round only after everybody read it: they all
			// did before arriving at the barrier inside this call
			block_sums<CB>(chi, redB);
			if (threadIdx.x < CB && b0 + threadIdx.x < B) {
				const double L = -0.5 * (threadIdx.x == 0 ? chi[0] : chi[CB - 1]);
				out[(size_t) (b0 + threadIdx.x) * M + k] = L;
				// (the band test of the likelihood noise rides along: mdns_internal.h, band_vote)
				if (band.sc) band_vote(band.sc, b0 + (int) threadIdx.x, k, L, band.higher[row], band.bound[b0 + threadIdx.x]);
			}
		}
	}
	if (!band.sc) return;
	// the last workgroup to finish publishes (hand-over of mdns_internal.h)
	__shared__ int s_last;
	handover_release();
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(&band.sc->ticket, 1) == (int) (gridDim.x * gridDim.y) - 1 ? 1 : 0;
	__syncthreads();
	if (!s_last) return;
	handover_acquire();
	band_publish(band.sc, B, band.box, band.seq);
}

// Many candidates: the limit of k_muse_rows is the 32 KiB template that every (candidate,
// row) pair pulls from L2 (15.5 TB/s at B = 64).  Here a workgroup keeps RB = 2 spectra in
// registers and applies each template to both, halving that traffic; y w is formed on the fly
// to make room (8 VALU operations per candidate and channel instead of 6 -- the VALU was idle).
template <int NP>
__global__ __launch_bounds__(kBlock) void k_muse_rows2(
    const double *__restrict__ Y, const double *__restrict__ W, int ld, int nx,
    const double *__restrict__ model, int ldm, int B, const int *__restrict__ rows, int M,
    double *__restrict__ out)
{
	constexpr int RB = 2;
	__shared__ double redA[4 * 2 * RB], redB[4 * RB];
	const int ch = 2 * threadIdx.x;
	bool valid[NP];
#pragma unroll
	for (int p = 0; p < NP; p++) valid[p] = (p * 512 + ch) < nx;

	for (int k0 = blockIdx.x * RB; k0 < M; k0 += gridDim.x * RB) {
		double2 y[RB][NP], w[RB][NP];
#pragma unroll
		for (int r = 0; r < RB; r++) {
			const int k = (k0 + r < M) ? k0 + r : M - 1;
			const size_t base = (size_t) (rows ? rows[k] : k) * ld + ch;
#pragma unroll
			for (int p = 0; p < NP; p++) {
				y[r][p] = valid[p] ? *reinterpret_cast<const double2 *>(Y + base + p * 512) : make_double2(0.0, 0.0);
				w[r][p] = valid[p] ? *reinterpret_cast<const double2 *>(W + base + p * 512) : make_double2(0.0, 0.0);
			}
		}
		for (int b = 0; b < B; b++) {
			double2 m[NP];
			double sums[2 * RB];
#pragma unroll
			for (int i = 0; i < 2 * RB; i++) sums[i] = 0.0;
#pragma unroll
			for (int p = 0; p < NP; p++) {
				m[p] = *reinterpret_cast<const double2 *>(model + (size_t) b * ldm + p * 512 + ch);
				const double mmx = m[p].x * m[p].x, mmy = m[p].y * m[p].y;
#pragma unroll
				for (int r = 0; r < RB; r++) {
					sums[2 * r] = fma(y[r][p].x * w[r][p].x, m[p].x, sums[2 * r]);
					sums[2 * r] = fma(y[r][p].y * w[r][p].y, m[p].y, sums[2 * r]);
					sums[2 * r + 1] = fma(mmx, w[r][p].x, sums[2 * r + 1]);
					sums[2 * r + 1] = fma(mmy, w[r][p].y, sums[2 * r + 1]);
				}
			}
			block_sums<2 * RB>(sums, redA);
			double chi[RB];
#pragma unroll
			for (int r = 0; r < RB; r++) {
				const double s = sums[2 * r] / (1e-10 + sums[2 * r + 1]);      // cmuselike.c:52,57
				double acc = 0.0;
#pragma unroll
				for (int p = 0; p < NP; p++) {
					const double r0 = fma(-s, m[p].x, y[r][p].x);
					const double r1 = fma(-s, m[p].y, y[r][p].y);
					acc = fma(r0 * r0, w[r][p].x, acc);
					acc = fma(r1 * r1, w[r][p].y, acc);
				}
				chi[r] = acc;
			}
			block_sums<RB, false>(chi, redB);
			if (threadIdx.x < RB && k0 + threadIdx.x < M)
				out[(size_t) b * M + k0 + threadIdx.x] = -0.5 * (threadIdx.x == 0 ? chi[0] : chi[RB - 1]);
		}
	}
}

// any nx: two passes over the row from memory (the row is L2-hot for the second pass)
__global__ __launch_bounds__(kBlock) void k_muse_rows_generic(
    const double *__restrict__ Y, const double *__restrict__ W, int ld, int nx,
    const double *__restrict__ model, int ldm, int B, const int *__restrict__ rows, int M,
    double *__restrict__ out)
{
	__shared__ double red[3][4];
	for (int k = blockIdx.x; k < M; k += gridDim.x) {
		const int row = rows ? rows[k] : k;
		const double *yr = Y + (size_t) row * ld, *wr = W + (size_t) row * ld;
		for (int b = 0; b < B; b++) {
			const double *mr = model + (size_t) b * ldm;
			double s1 = 0.0, s2 = 0.0;
			for (int j = threadIdx.x; j < nx; j += kBlock) {
				s1 = fma(yr[j] * mr[j], wr[j], s1);
				s2 = fma(mr[j] * mr[j], wr[j], s2);
			}
			const double t1 = block_sum4(s1, red[0]);
			const double t2 = block_sum4(s2, red[1]);
			const double s = t1 / (1e-10 + t2);
			double chi = 0.0;
			for (int j = threadIdx.x; j < nx; j += kBlock) {
				const double r = yr[j] - s * mr[j];
				chi = fma(r * r, wr[j], chi);
			}
			const double tot = block_sum4(chi, red[2]);
			if (threadIdx.x == 0) out[(size_t) b * M + k] = -0.5 * tot;
		}
	}
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static bool launched(const char *name)
{
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) return true;
	set_error("launch of %s failed: %s", name, hipGetErrorString(e));
	return false;
}

bool launch_gauss_model(const double *d_x, int nx, const double *d_params, int B, double *d_model, int ldm)
{
	Context *c = ctx();
	dim3 grid((ldm + kBlock - 1) / kBlock, B);
	hipLaunchKernelGGL(k_gauss_model, grid, dim3(kBlock), 0, c->stream, d_x, nx, d_params, d_model, ldm);
	return launched("k_gauss_model");
}

// grid of the lane kernels for `ntiles` spectrum tiles and `nbt` candidate tiles (see cols_item)
static int cols_grid(const Context *c, int ntiles, int nbt, int &nq_xcd, int &cu_slots, int wpb = 4)
{
	const int nquads = (ntiles + wpb - 1) / wpb;
	nq_xcd = (nquads + 7) / 8;                              // quads per XCD (some may be empty)
	cu_slots = c->num_cus >= 8 ? c->num_cus / 8 : 1;
	const int rounds = (nbt * nq_xcd + cu_slots - 1) / cu_slots;
	return 8 * cu_slots * rounds;
}

int gauss_cols_tile(int M, int B)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	// candidates per wave, from the waves per SIMD a tile size leaves (4 SIMDs per CU).
	// Measured on MI355X (us per launch, 10 000 spectra unless noted; BT = 4 / 8 / 16):
	//   B = 64: 25.7 / 28.2 / 46.5     B = 128: 40.0 / 31.8 / 46.7    B = 256: 76.0 / 49.4 / 56.8
	//   B = 512: 144 / 97.1 / 92.6     B = 1024: 272 / 185 / 176      100 000 x 16: 56.8 / 40.5 / 51.0
	// 8 wins from ~2.4 waves per SIMD on (its template tile, 12.8 KB at 200 channels, stays in
	// the scalar cache and it needs half the L1 bandwidth of 4), 16 from ~4.5 on.
	const long long simds = 4LL * c->num_cus;
	auto waves10 = [&](int t) { return 10LL * ntiles * ((B + t - 1) / t); };
	int bt = 4;
	if (waves10(16) >= 45 * simds) bt = 16;
	else if (waves10(8) >= 24 * simds) bt = 8;
	while (bt > B && bt > 1) bt >>= 1;
	static const char *forced_bt = getenv("MDNS_K1_BT");      // experiments only
	if (forced_bt) { const int f = atoi(forced_bt); if (f == 1 || f == 2 || f == 4 || f == 8 || f == 16) bt = f; }
	return bt;
}

bool launch_gauss_model_t(const double *d_x, int nx, const double *d_params, int B, int bt, double *d_model_t,
                          int *d_zero, int nzero)
{
	Context *c = ctx();
	const int nxp = cols_nx(nx);
	const int ntile = (B + bt - 1) / bt;
	const int n = ntile * nxp * bt;
	const int threads = n > nzero ? n : nzero;
	hipLaunchKernelGGL(k_gauss_model_t, dim3((threads + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
	                   d_x, nx, nxp, d_params, B, bt, ntile, d_model_t, d_zero, d_zero ? nzero : 0);
	return launched("k_gauss_model_t");
}

bool launch_gauss_cols(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int bt, int B,
                       double scale, const int *d_rows, int M, double *d_out)
{
	Context *c = ctx();
	// one spectrum per lane: two per lane (template values reused twice) measured 18 % slower
	// (207 vs 176 us at B = 1024) -- half the waves, and the scalar path is not the limit
	constexpr int sp = 1;
	const int ntiles = (M + 64 * sp - 1) / (64 * sp);
	const int nbt = (B + bt - 1) / bt;
	// per XCD: nbt * nq_xcd items dealt to cu_slots CUs in contiguous runs (see cols_item).
	// Against the plain order (item = local) measured on one box: 49.4 vs 54.7 us at B = 256,
	// 31.8 vs 36.1 at B = 128, 391 vs 432 at 100 000 x 256; only 10 000 x 1024 lost (184 vs 176).
	int nq_xcd, cu_slots;
	const int blocks = cols_grid(c, ntiles, nbt, nq_xcd, cu_slots);
	ProfileScope prof(0);
	note_kernel(0, "k_gauss_cols<%d, %d>", bt, (int) sp);
#define COLS_LAUNCH(BT) hipLaunchKernelGGL((k_gauss_cols<BT, sp>), dim3(blocks), dim3(kBlock), 0, c->stream, \
	d_yT, cols_nx(s->nx), d_model_t, B, scale, d_rows, M, ntiles, nq_xcd, nbt, cu_slots, d_out)
	switch (bt) {
	case 16: COLS_LAUNCH(16); break;
	case 8: COLS_LAUNCH(8); break;
	case 4: COLS_LAUNCH(4); break;
	case 2: COLS_LAUNCH(2); break;
	default: COLS_LAUNCH(1); break;
	}
#undef COLS_LAUNCH
	return launched("k_gauss_cols");
}

bool launch_gauss_cols_accept(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int bt, int B,
                              double scale, const int *d_rows, const int *d_thr_rows, int M,
                              const double *d_higher, int *d_flags, const JointTrail &trail)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	const int nbt = (B + bt - 1) / bt;
	int nq_xcd, cu_slots;
	// (512-thread workgroups -- 8 spectrum tiles of a candidate tile per workgroup, fewer template
	// streams per CU -- measured slower: 48.7 vs 43.4 us at 10 000 x 256)
	const int wpb = 4;
	const int blocks = cols_grid(c, ntiles, nbt, nq_xcd, cu_slots, wpb);
	ProfileScope prof(0);
	note_kernel(0, "k_gauss_cols_accept<%d, %s>", bt, ((long long) ntiles * nbt < 8LL * c->num_cus && bt <= 4 && (size_t) cols_nx(s->nx) * bt * sizeof(double) <= 48 * 1024) ? "true" : "false");
	// fewer than two waves per SIMD: nothing hides a wave's memory latency but its own loads
	const size_t tpl_bytes = (size_t) cols_nx(s->nx) * bt * sizeof(double);
	const bool deep = (long long) ntiles * nbt < 8LL * c->num_cus && bt <= 4 && tpl_bytes <= 48 * 1024;
#define ACCEPT_LAUNCH(BT, DEEP) hipLaunchKernelGGL((k_gauss_cols_accept<BT, DEEP>), dim3(blocks), dim3(64 * wpb), DEEP ? tpl_bytes : 0, c->stream, \
	d_yT, cols_nx(s->nx), d_model_t, B, scale, d_rows, d_thr_rows, M, ntiles, nq_xcd, nbt, cu_slots, d_higher, d_flags, trail)
	switch (bt) {
	case 16: ACCEPT_LAUNCH(16, false); break;
	case 8: ACCEPT_LAUNCH(8, false); break;
	case 4: if (deep) ACCEPT_LAUNCH(4, true); else ACCEPT_LAUNCH(4, false); break;
	case 2: if (deep) ACCEPT_LAUNCH(2, true); else ACCEPT_LAUNCH(2, false); break;
	default: if (deep) ACCEPT_LAUNCH(1, true); else ACCEPT_LAUNCH(1, false); break;
	}
#undef ACCEPT_LAUNCH
	return launched("k_gauss_cols_accept");
}

bool launch_row_sumsq(const double *d_y, int ld, int nx, int ndata, double *d_out)
{
	Context *c = ctx();
	if (ndata <= 0) return true;
	hipLaunchKernelGGL(k_row_sumsq, dim3((ndata + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, d_y, ld, nx, ndata, d_out);
	return launched("k_row_sumsq");
}

// whether the guarded filter pays for a chunk of B candidates over M spectra: launches the lane
// kernel runs issue-bound (8+ candidates per wave)
int gauss_filter_pays(const mdns_spectra *s, int M, int B)
{
	// MDNS_K1_FILTER=1: the vector-FMA form.  It gains little -- 33.9 us against 41-43 at
	// 10 000 x 256 (rocprofv3), because the kernel is bound by the delivery of its scalar template
	// operands, not by issue (SQ counters: VALU active 49 %, waves 65 % of their time in s_waitcnt) --
	// and the chain sub-pass of the accepted candidate (one wave-length of latency-bound loads at
	// the END of its waves) gives the gain back: 57 us.
	// The matrix-core form (mdns_chunk.hip; MDNS_K1_FILTER=mfma forces it, 0 turns it off): the filter
	// kernel itself takes 29-32 us against the chain kernel's 40-45 at 10 000 x 256, but its route has
	// two more passes -- templates with sums of squares 7.7 us against 3.9, and the chain score of the
	// accepted candidate 9.9 -- so that a sampler step there gains nothing (rocprofv3 of bench.py:
	// 49.5 us against 43.9 for the three / two kernels).  It is the default where those passes are
	// small beside the product: 122 us against 189 at 50 000 x 256, 101 against 162 at 10 000 x 1024.
	static const char *forced = getenv("MDNS_K1_FILTER");
	if (s->d_ysq == nullptr || gauss_cols_tile(M, B) < 8) return 0;
	if (forced && forced[0] == '1') return 1;
	if (forced && forced[0] == '0') return 0;
	if (forced && forced[0] == 'm') return 2;
	// (round 4: with both operands straight from memory in tiles of 16 rows -- k_gauss_gemm_filter -- the filter kernel takes
	// 29 us against the chain's 43 at 10 000 x 256 and a chunk alone 74.6 / 72.1 us against 86.9 / 81.3 (tools/filter_bench.py);
	// but inside a whole sampler step the route still loses there: bench.py 96.8 us per step against 92.0 on the same box.  So
	// the threshold stays where the passes around the product are small beside it.)
	return B >= 128 && (long long) M * B >= 8000000LL ? 2 : 0;
}

// candidates per wave of the filter: with ONE vector instruction per (candidate, channel, spectrum)
// a wave of 8 candidates asks the L1 for 4 KB of spectra per 64 instructions -- 64 B per clock and
// CU, all the L1 delivers (measured: 39.6 us at 10 000 x 256, hardly better than the 45 of the
// add + fma kernel) -- so 16 wherever that still leaves two waves per SIMD
int gauss_filter_tile(int M, int B)
{
	Context *c = ctx();
	static const char *forced = getenv("MDNS_K1_FILTER_BT");               // experiments only
	if (forced && (atoi(forced) == 8 || atoi(forced) == 16)) return atoi(forced);
	const long long waves = (long long) ((M + 63) / 64) * ((B + 15) / 16);
	return waves >= 2LL * 4 * c->num_cus ? 16 : 8;
}

// templates + their sums of squares for the filter (also clears d_zero[0 .. nzero))
bool launch_gauss_model_tsq(const double *d_x, int nx, const double *d_params, int B, int bt, double *d_model_t, double *d_msq,
                            int *d_zero, int nzero, double *d_model_g)
{
	Context *c = ctx();
	const int ntile = (B + bt - 1) / bt;
	if (d_model_g && bt != 16) { set_error("launch_gauss_model_tsq: tiled templates come 16 candidates wide"); return false; }
	// partial sums of squares and tickets of the shared tiles: grow-only, the tickets zero between launches
	static double *d_part = nullptr;
	static int *d_tickets = nullptr;
	static int cap_tiles = 0;
	constexpr int kShares = 4;
	static const char *one = getenv("MDNS_TSQ_SHARES");                  // "1": one workgroup per tile (experiments)
	const int ny = one && one[0] == '1' ? 1 : kShares;
	if (ny > 1 && ntile > cap_tiles) {
		if (d_part) { (void) hipStreamSynchronize(c->stream); (void) hipFree(d_part); (void) hipFree(d_tickets); d_part = nullptr; d_tickets = nullptr; cap_tiles = 0; }
		const int cap = ntile + 64;
		if (!MDNS_HIP(hipMalloc((void **) &d_part, (size_t) cap * kShares * 64 * sizeof(double))) ||
		    !MDNS_HIP(hipMalloc((void **) &d_tickets, (size_t) cap * sizeof(int))) ||
		    !MDNS_HIP(hipMemsetAsync(d_tickets, 0, (size_t) cap * sizeof(int), c->stream))) return false;
		cap_tiles = cap;
	}
	hipLaunchKernelGGL(k_gauss_model_tsq, dim3(ntile, ny), dim3(256), 0, c->stream, d_x, nx, cols_nx(nx), d_params, B, bt, d_model_t, d_msq,
	                   d_zero, d_zero ? nzero : 0, d_model_g, tiled16_nx(nx), d_part, d_tickets);
	return launched("k_gauss_model_tsq");
}

bool launch_tile_rows16(const double *d_y, int ld, int M, int nx, const int *d_rows, double *d_out)
{
	Context *c = ctx();
	if (M <= 0 || nx <= 0) return true;
	hipLaunchKernelGGL(k_tile_rows16, dim3((M + 15) / 16), dim3(256), 0, c->stream, d_y, ld, M, nx, tiled16_nx(nx), d_rows, d_out);
	return launched("k_tile_rows16");
}

// the accept pass as guarded filter (see k_gauss_cols_filter): same flags and trail as
// launch_gauss_cols_accept; d_msq f64[>= B + 16] from launch_gauss_model_tsq; d_lowest: an int the
// template kernel cleared
bool launch_gauss_cols_filter(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int bt, int B,
                              double scale, const int *d_rows, const int *d_thr_rows, int M,
                              const double *d_higher, int *d_flags, const double *d_msq, const JointTrail &trail, int *d_lowest)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	const int nbt = (B + bt - 1) / bt;
	const int nxp = cols_nx(s->nx);
	int nq_xcd, cu_slots;
	const int blocks = cols_grid(c, ntiles, nbt, nq_xcd, cu_slots, 4);
	ProfileScope prof(0);
	note_kernel(0, "k_gauss_cols_filter<%d>", bt);
#define FILTER_LAUNCH(BT) hipLaunchKernelGGL((k_gauss_cols_filter<BT>), dim3(blocks), dim3(256), 0, c->stream, \
	d_yT, nxp, s->nx, d_model_t, d_msq, B, scale, d_rows, d_thr_rows, M, ntiles, nq_xcd, nbt, cu_slots, d_higher, (const double *) s->d_ysq, d_flags, trail, d_lowest)
	if (bt == 16) FILTER_LAUNCH(16); else FILTER_LAUNCH(8);
#undef FILTER_LAUNCH
	return launched("k_gauss_cols_filter");
}

bool launch_gauss_cols_commit(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int mstride, int B,
                              double scale, const int *d_rows, const int *d_thr_rows, int M, const int *d_flags,
                              const JointArrays &st, void *d_header, unsigned long long *d_fillbits, double *d_Lrow)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	int nq_xcd, cu_slots;
	const int blocks = cols_grid(c, ntiles, 1, nq_xcd, cu_slots);
	hipLaunchKernelGGL(k_gauss_cols_commit, dim3(blocks), dim3(kBlock), (size_t) (cols_nx(s->nx) + 2) * sizeof(double), c->stream,
	                   d_yT, cols_nx(s->nx), d_model_t, mstride, B, scale, d_rows, d_thr_rows, M, ntiles, nq_xcd, cu_slots,
	                   d_flags, st, (JointHeader *) d_header, d_fillbits, d_Lrow);
	return launched("k_gauss_cols_commit");
}

bool launch_joint_commit_trail(const int *d_thr_rows, int M, int B, const int *d_flags, const JointTrail &trail,
                               const JointArrays &st, void *d_header, unsigned long long *d_fillbits, int flag_value,
                               void *box_dev, unsigned long long seq, int *d_ticket)
{
	Context *c = ctx();
	const int ntiles = (M + 63) / 64;
	hipLaunchKernelGGL(k_joint_commit_trail, dim3((ntiles + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, c->stream,
	                   d_thr_rows, M, B, ntiles, d_flags, flag_value, trail, st, (JointHeader *) d_header, d_fillbits,
	                   (JointMailbox *) box_dev, seq, d_ticket);
	return launched("k_joint_commit_trail");
}

bool launch_muse3_model(const double *d_x, int nx, const double *d_params, int B, double *d_model, int ldm)
{
	Context *c = ctx();
	dim3 grid((ldm + kBlock * kMuseModelPer - 1) / (kBlock * kMuseModelPer), B);
	hipLaunchKernelGGL(k_muse3_model, grid, dim3(kBlock), 0, c->stream, d_x, nx, d_params, d_model, ldm);
	return launched("k_muse3_model");
}

bool launch_pad_model(const double *d_src, int nx, int B, double *d_dst, int ldm)
{
	Context *c = ctx();
	dim3 grid((ldm + kBlock - 1) / kBlock, B);
	hipLaunchKernelGGL(k_pad_model, grid, dim3(kBlock), 0, c->stream, d_src, nx, d_dst, ldm);
	return launched("k_pad_model");
}

bool launch_transpose(const double *d_src, int nx, int ndata, double *d_dst, int ld, bool invert, int lds)
{
	Context *c = ctx();
	if (nx == 0 || ndata == 0) return true;
	dim3 grid((ndata + 31) / 32, (nx + 31) / 32);
	hipLaunchKernelGGL(k_transpose, grid, dim3(kBlock), 0, c->stream, d_src, nx, ndata, d_dst, ld, invert ? 1 : 0, lds);
	return launched("k_transpose");
}

bool launch_tile_columns(const double *d_y, int ld, int ndata, int nx, const int *d_rows, double *d_yt)
{
	Context *c = ctx();
	if (nx == 0 || ndata == 0) return true;
	const int nxp = cols_nx(nx);
	dim3 grid((ndata + 63) / 64, (nxp + 31) / 32);
	hipLaunchKernelGGL(k_tile_columns, grid, dim3(kBlock), 0, c->stream, d_y, ld, ndata, nx, nxp, d_rows, d_yt);
	return launched("k_tile_columns");
}

bool launch_copy_rows(const double *d_src, int nx, int ndata, double *d_dst, int ld, bool invert)
{
	Context *c = ctx();
	if (nx == 0 || ndata == 0) return true;
	const size_t n = (size_t) nx * ndata;
	const int blocks = (int) ((n + kBlock - 1) / kBlock < 8192 ? (n + kBlock - 1) / kBlock : 8192);
	hipLaunchKernelGGL(k_copy_rows, dim3(blocks), dim3(kBlock), 0, c->stream, d_src, nx, ndata, d_dst, ld, invert ? 1 : 0);
	return launched("k_copy_rows");
}

template <int NP, int BT, int R>
static void launch_gauss_rows_t(const mdns_spectra *s, const double *d_model, int ldm, int B, double scale,
                                const int *d_rows, int M, double *d_out, hipStream_t stream, int num_cus)
{
	const int waves_needed = (M + R - 1) / R;
	int blocks = (waves_needed + 3) / 4;
	const int cap = num_cus * 8;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	note_kernel(0, "k_gauss_rows<%d, %d, %d, %s>", NP, BT, R, B <= BT ? "true" : "false");
	if (B <= BT)
		hipLaunchKernelGGL((k_gauss_rows<NP, BT, R, true>), dim3(blocks), dim3(kBlock), 0, stream,
		                   s->d_y, s->ld, s->nx, d_model, ldm, B, scale, d_rows, M, d_out);
	else
		hipLaunchKernelGGL((k_gauss_rows<NP, BT, R, false>), dim3(blocks), dim3(kBlock), 0, stream,
		                   s->d_y, s->ld, s->nx, d_model, ldm, B, scale, d_rows, M, d_out);
}

bool launch_gauss_rows(const mdns_spectra *s, const double *d_model, int ldm, int B, double scale,
                       const int *d_rows, int M, double *d_out)
{
	Context *c = ctx();
	const int nx = s->nx;
	ProfileScope prof(0);
	if (nx <= 128) {
		if (B == 1) launch_gauss_rows_t<1, 1, 4>(s, d_model, ldm, B, scale, d_rows, M, d_out, c->stream, c->num_cus);
		else        launch_gauss_rows_t<1, 4, 4>(s, d_model, ldm, B, scale, d_rows, M, d_out, c->stream, c->num_cus);
	} else if (nx <= 256) {
		if (B == 1) launch_gauss_rows_t<2, 1, 4>(s, d_model, ldm, B, scale, d_rows, M, d_out, c->stream, c->num_cus);
		else        launch_gauss_rows_t<2, 4, 4>(s, d_model, ldm, B, scale, d_rows, M, d_out, c->stream, c->num_cus);
	} else if (nx <= 512) {
		if (B == 1) launch_gauss_rows_t<4, 1, 2>(s, d_model, ldm, B, scale, d_rows, M, d_out, c->stream, c->num_cus);
		else        launch_gauss_rows_t<4, 4, 2>(s, d_model, ldm, B, scale, d_rows, M, d_out, c->stream, c->num_cus);
	} else {
		int blocks = (M + 3) / 4;
		if (blocks > c->num_cus * 8) blocks = c->num_cus * 8;
		hipLaunchKernelGGL((k_gauss_rows_generic<4>), dim3(blocks), dim3(kBlock), 0, c->stream,
		                   s->d_y, s->ld, nx, d_model, ldm, B, scale, d_rows, M, d_out);
	}
	return launched("k_gauss_rows");
}

// which instantiation scores a block of B candidates x M spectra: 2 two rows per workgroup, 1 pairs of
// candidates, 0 one candidate.  (Their reductions associate differently: the last bits of a likelihood
// depend on the instantiation and, for pairs, on the candidate's place in its pair.)
int muse_rows_variant(int B, int M)
{
	Context *c = ctx();
	static const char *k2v = getenv("MDNS_K2_ROWS2");         // experiments only: "0" disables
	if (B >= 4 && M >= 2 * c->num_cus && !(k2v && k2v[0] == '0')) return 2;
	return B >= 2 ? 1 : 0;
}

// B_shape > 0: score these B candidates with the instantiation a block of B_shape candidates would take
// (for variant 1 the caller passes whole pairs of that block)
bool launch_muse_rows(const mdns_spectra *s, const double *d_model, int ldm, int B, const int *d_rows,
                      int M, double *d_out, int B_shape, const MuseBandFused *band)
{
	const MuseBandFused none = {nullptr, nullptr, 0, nullptr, nullptr, nullptr};
	if (band && muse_rows_variant(B_shape > 0 ? B_shape : B, M) != 1) { set_error("launch_muse_rows: the band test rides along with pairs of candidates only"); return false; }
	const MuseBandFused fused = band ? *band : none;
	Context *c = ctx();
	const int nx = s->nx;
	int blocks = M < c->num_cus * 8 ? M : c->num_cus * 8;
	if (blocks < 1) blocks = 1;
	ProfileScope prof(1);
	const int variant = muse_rows_variant(B_shape > 0 ? B_shape : B, M);
	const bool two_rows = variant == 2;
	// few rows, several candidates: split the candidates over grid.y until ~2 workgroups per CU
	// (45 rows x 64 candidates: 69 us as one workgroup per row)
	// (with more than half a workgroup per CU it measured slower: 407 rows x 16: 29.7 vs 24.2 us)
	int gy = 2 * blocks <= c->num_cus ? (2 * c->num_cus + blocks - 1) / blocks : 1;
	if (gy > (B + 1) / 2) gy = (B + 1) / 2;
	if (gy < 1) gy = 1;
	const int bchunk = 2 * (((B + gy - 1) / gy + 1) / 2);         // even: k_muse_rows<NP, 2> walks pairs
	gy = (B + bchunk - 1) / bchunk;
	note_kernel(1, two_rows ? "k_muse_rows2<%d>" : (variant == 1 ? "k_muse_rows<%d, 2>" : "k_muse_rows<%d, 1>"),
	            nx <= 512 ? 1 : nx <= 1024 ? 2 : nx <= 2048 ? 4 : 8);
#define MUSE_LAUNCH(NP) do { if (two_rows) hipLaunchKernelGGL((k_muse_rows2<NP>), dim3((blocks + 1) / 2), dim3(kBlock), 0, c->stream, \
		s->d_y, s->d_w, s->ld, nx, d_model, ldm, B, d_rows, M, d_out); \
	else if (variant == 1) hipLaunchKernelGGL((k_muse_rows<NP, 2>), dim3(blocks, gy), dim3(kBlock), 0, c->stream, \
		s->d_y, s->d_w, s->ld, nx, d_model, ldm, B, d_rows, M, d_out, bchunk, fused); \
	else hipLaunchKernelGGL((k_muse_rows<NP, 1>), dim3(blocks), dim3(kBlock), 0, c->stream, \
		s->d_y, s->d_w, s->ld, nx, d_model, ldm, B, d_rows, M, d_out, B, none); } while (0)
	if (nx <= 512) MUSE_LAUNCH(1);
	else if (nx <= 1024) MUSE_LAUNCH(2);
	else if (nx <= 2048) MUSE_LAUNCH(4);
	else if (nx <= 4096) MUSE_LAUNCH(8);
	else
		hipLaunchKernelGGL(k_muse_rows_generic, dim3(blocks), dim3(kBlock), 0, c->stream,
		                   s->d_y, s->d_w, s->ld, nx, d_model, ldm, B, d_rows, M, d_out);
#undef MUSE_LAUNCH
	return launched("k_muse_rows");
}

}  // namespace mdns
