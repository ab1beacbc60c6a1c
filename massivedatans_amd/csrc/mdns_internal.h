// Internal declarations shared by the translation units of libmdns_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include "mdns.h"

namespace mdns {

// ---- error handling ---------------------------------------------------------------------
void set_error(const char *fmt, ...);
bool hip_ok(hipError_t e, const char *what, const char *file, int line);
#define MDNS_HIP(call) ::mdns::hip_ok((call), #call, __FILE__, __LINE__)

// Busy polls of mapped mailboxes give up after MDNS_POLL_TIMEOUT_S seconds (default 120): a wedged
// kernel or a lost mailbox store becomes an error the caller can act on instead of a core spinning
// forever.  *started_ns: 0 before the first call of a wait.
bool poll_expired(long long *started_ns);

// ---- per-process context (one process drives one GPU) -----------------------------------
struct Context {
	int device = -1;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;     // stream every launch goes to (own_stream unless overridden)
	int num_cus = 256;
	// grow-only scratch: device workspace and pinned host staging
	void *d_ws = nullptr;   size_t d_ws_bytes = 0;
	void *h_pin = nullptr;  size_t h_pin_bytes = 0;
	void *d_mask = nullptr; size_t d_mask_bytes = 0;
};
// nullptr (and mdns_last_error set) when no device can be initialised
Context *ctx();
// scratch accessors; return nullptr on allocation failure.  Contents are NOT preserved
// across a growing call.
void *device_scratch(size_t bytes);
void *pinned_scratch(size_t bytes);
void *mask_scratch(size_t bytes);      // packed bootstrap masks (separate from device_scratch)

// ---- resident spectra -------------------------------------------------------------------
}  // namespace mdns

struct mdns_spectra {
	int ndata = 0;      // number of spectra (rows)
	int nx = 0;         // channels per spectrum
	int ld = 0;         // row stride in doubles (nx rounded up to even => 16-byte aligned rows)
	double *d_y = nullptr;   // [ndata, ld]   one spectrum per row
	double *d_yT = nullptr;  // [ldT/64][cols_nx(nx)][64] channel-major replica in tiles of 64 spectra
	                         // (K1 lane kernel), or nullptr
	int ldT = 0;             // ndata rounded up to a multiple of 64 (zero padded)
	double *d_w = nullptr;   // [ndata, ld] inverse variances 1/v (K2), or nullptr
	double *d_x = nullptr;   // [nx] wavelength grid, or nullptr
	double *d_ysq = nullptr; // [ndata] sum of squares of every spectrum (K1 accept filter), or nullptr
	double *d_yG = nullptr;  // K1 on the matrix cores with operands straight from memory (k_gauss_gemm_filter): the spectra in
	                         // tiles of 16 rows, channel pair by channel pair (tiled16_at), channels padded to 16; or nullptr
	double *d_selG = nullptr; size_t selG_cap = 0;     // the same of the current selection
	double *d_model_g = nullptr; size_t model_g_cap = 0;   // templates in the same tiling
	// per-handle grow-only device buffers for the host-pointer batch API
	double *d_model = nullptr; size_t model_cap = 0;   // [B, ldm]
	double *d_params = nullptr; size_t params_cap = 0;
	int *d_rows = nullptr; size_t rows_cap = 0;
	double *d_out = nullptr; size_t out_cap = 0;
	double *d_sel = nullptr; size_t sel_cap = 0;       // compact replica of the current selection (K1 lane kernel)
	// K2 on the matrix cores (mdns_k2gemm.hip), made on first use: y w and w [ndata, ldf] with ldf = nx rounded
	// up to 16 (zero padded; d_fw is d_w itself when the strides agree), A = sum y^2 w [ndata]
	double *d_fyw = nullptr, *d_fw = nullptr, *d_fa = nullptr;
	double *d_fyw_t = nullptr, *d_fw_t = nullptr;       // the same two in tiles of 16 rows (mdns_k2gemm.hip, tiled_at)
	int ldf = 0;
	bool fw_owned = false;
};

namespace mdns {

// model row stride for nx channels: multiple of 512 doubles (zero padded), so that every
// lane / thread of the row kernels can load its channel pair without a bounds test
inline int model_ld(int nx) { return nx <= 0 ? 512 : ((nx + 511) / 512) * 512; }

// channel count of the channel-major replica / transposed templates: padded (with zeros) to
// the software-pipeline depth of k_gauss_cols
inline int cols_nx(int nx) { return ((nx + 7) / 8) * 8; }
// grow-only device buffers of a spectra handle (contents not kept): templates, compact replica
bool ensure_model(mdns_spectra *s, size_t doubles);
bool ensure_selection(mdns_spectra *s, size_t doubles);
// K1 through the lane kernel whatever the shape (mdns_core.hip)
int gauss_loglike_cols_dev(mdns_spectra *s, const double *d_params, int B, double noise_level,
                           const int *d_row_ids, int M, double *d_Lout);
// launchers implemented in mdns_like.hip (all asynchronous on ctx()->stream)
bool launch_gauss_model(const double *d_x, int nx, const double *d_params, int B,
                        double *d_model, int ldm);
// candidates per wave of k_gauss_cols for M selected spectra and B candidates (1..16)
int gauss_cols_tile(int M, int B);
// tiled templates MT[ceil(B/bt)][cols_nx(nx)][bt], zero for b >= B and j >= nx
// (d_zero, nzero): a small int buffer the kernel clears on the way (accept flags + result header)
bool launch_gauss_model_t(const double *d_x, int nx, const double *d_params, int B, int bt,
                          double *d_model_t, int *d_zero = nullptr, int nzero = 0);
// device arrays of the joint sampler state (mdns_joint.hip), all indexed by original data set
struct JointArrays {
	double *live;      // [nlive][ndata]   live-point likelihoods (multi_nested_sampler.py:111)
	double *shelfL;    // [cap][ndata]     likelihoods waiting on the shelves (:117)
	int *shelfn;       // [ndata]          how many
	double *higher;    // [ndata]          threshold of the next draw (:438-447)
	int nlive, cap, ndata;
};
// what a chunk of a constrained draw leaves for the host (mdns.h, mdns_joint_commit_dev)
struct JointHeader { int accepted; int status; long long pad; };
// What the accept pass leaves behind for the candidates it flags: per (candidate, tile of 64
// selected spectra) in which some lane beat its threshold, the ballot word and the 64
// likelihoods -- all a commit needs when nobody asks for the whole likelihood row.  Entries are
// valid when their stamp is the current one (nothing is ever cleared).
struct JointTrail {
	int *stamp_of;                 // [candidates x tiles]      (nullptr: no trail)
	unsigned long long *word;      // [candidates x tiles]
	double *L;                     // [candidates x tiles][64]
	int stamp;
};
// What a commit leaves for a caller that does not want the likelihood row, in host memory
// mapped into the device: the host polls `seq` instead of copying and synchronising.
struct JointMailbox { unsigned long long seq; int accepted; int status; unsigned long long bits[1]; /* ceil(M/64) words */ };
// a draw chunk in two launches (mdns_chunk.hip): whether the shape qualifies, and the launchers.
// Candidates (and, for the first chunk of a draw, the selection's row ids) are read from host
// memory mapped into the device; accepted candidates get flags[b] = stamp.
bool chunk_fits(const mdns_spectra *s, int M, int B);
bool launch_chunk_accept(const mdns_spectra *s, const double *d_params_mapped, int B, double scale,
                         const int *d_rows_in, int *d_rows_dev, int M, const double *d_higher,
                         int *d_flags, int stamp, const JointTrail &trail, void *d_header);
bool launch_chunk_commit(const int *d_thr_rows, int M, int B, const int *d_flags, int stamp, const JointTrail &trail,
                         const JointArrays &st, void *d_header, unsigned long long *d_fillbits, void *box_dev,
                         unsigned long long seq);
// accept test fused into the lane kernel: flags[b] = 1 when candidate b beats a threshold
bool launch_gauss_cols_accept(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int bt, int B,
                              double scale, const int *d_rows, const int *d_thr_rows, int M,
                              const double *d_higher, int *d_flags, const JointTrail &trail);
// the accept test as guarded filter (mdns_like.hip, k_gauss_cols_filter): same flags and trail as
// launch_gauss_cols_accept
// 0: the chain kernel decides; 1: vector-FMA filter; 2: matrix-core filter
int gauss_filter_pays(const mdns_spectra *s, int M, int B);
int gauss_filter_tile(int M, int B);
// d_model_g != nullptr (bt = 16): the templates also in the tiled16 layout
bool launch_gauss_model_tsq(const double *d_x, int nx, const double *d_params, int B, int bt, double *d_model_t, double *d_msq,
                            int *d_zero, int nzero, double *d_model_g = nullptr);
bool launch_gauss_cols_filter(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int bt, int B,
                              double scale, const int *d_rows, const int *d_thr_rows, int M,
                              const double *d_higher, int *d_flags, const double *d_msq, const JointTrail &trail, int *d_lowest);
bool launch_row_sumsq(const double *d_y, int ld, int nx, int ndata, double *d_out);
// element (row r, channel c) of an operand in tiles of 16 rows, inside a tile channel pair by channel pair, 16 rows x 2
// doubles: the 16 lanes of a quarter wave (the same channels of 16 rows) read 256 contiguous bytes; ncp = channels / 2
#ifdef __HIPCC__
__host__ __device__
#endif
inline size_t tiled16_at(size_t r, int c, int ncp) { return (((r >> 4) * (size_t) ncp + (size_t) (c >> 1)) << 5) + ((r & 15) << 1) + (size_t) (c & 1); }
inline int tiled16_nx(int nx) { return (nx + 15) & ~15; }
// rows [M or ndata][ld] (d_rows: which, or nullptr) -> tiled16 replica, zero padded
bool launch_tile_rows16(const double *d_y, int ld, int M, int nx, const int *d_rows, double *d_out);
// the same decision on the matrix cores (mdns_chunk.hip, k_gauss_mfma_filter + k_exact_list);
// templates tiled 16 wide; d_scratch int32[MDNS_JOINT_MAX_BATCH + 16], zeroed once
bool launch_gauss_mfma_filter(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int B, double scale,
                              const int *d_thr_rows, int M, const double *d_higher, int *d_flags,
                              const double *d_msq, const JointTrail &trail, int *d_lowest, int *d_scratch, void *d_header,
                              const double *d_yG = nullptr, const double *d_model_g = nullptr);
// which form of the matrix-core filter: 0 staged through LDS (round 3), 1 operands straight from memory in the layouts
// of the lane kernel (opt-in), 2 operands straight from memory, tiled16 (needs d_yG / d_model_g)
int gauss_mfma_form();
// first flagged candidate from the trail of the accept pass: fill bits, shelf appends, thresholds
// (flag_value: what the accept pass wrote into d_flags for an accepted candidate)
// box_dev != nullptr: the kernel's last workgroup also fills the mailbox (no k_joint_publish behind it); d_ticket: an int, zero
// between launches
bool launch_joint_commit_trail(const int *d_thr_rows, int M, int B, const int *d_flags, const JointTrail &trail,
                               const JointArrays &st, void *d_header, unsigned long long *d_fillbits, int flag_value = 1,
                               void *box_dev = nullptr, unsigned long long seq = 0, int *d_ticket = nullptr);
// first flagged candidate: its likelihood row, fill bits, shelf appends, new thresholds
bool launch_gauss_cols_commit(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int mstride, int B,
                              double scale, const int *d_rows, const int *d_thr_rows, int M, const int *d_flags,
                              const JointArrays &st, void *d_header, unsigned long long *d_fillbits, double *d_Lrow);
bool launch_gauss_cols(const mdns_spectra *s, const double *d_yT, const double *d_model_t, int bt, int B,
                       double scale, const int *d_rows, int M, double *d_out);
bool launch_muse3_model(const double *d_x, int nx, const double *d_params, int B,
                        double *d_model, int ldm);
bool launch_gauss_rows(const mdns_spectra *s, const double *d_model, int ldm, int B,
                       double scale, const int *d_rows, int M, double *d_out);
struct MuseBandFused;
// band != nullptr (pairs of candidates only: muse_rows_variant(...) == 1): the band test of the likelihood noise and its
// mailbox ride along with the scoring -- no k_joint_band behind it
bool launch_muse_rows(const mdns_spectra *s, const double *d_model, int ldm, int B,
                      const int *d_rows, int M, double *d_out, int B_shape = 0, const MuseBandFused *band = nullptr);
int muse_rows_variant(int B, int M);
// the band test of a chunk (mdns_joint.hip, k_joint_band) on K2 as two matrix products (mdns_k2gemm.hip):
// where its outcome goes (device memory; clear / maybe per candidate, listed pairs behind a counter)
struct MuseBandOut { int *counter, *clear, *maybe, *pair_b, *pair_k; double *pair_L, *pair_thr; int cap; int *zero_at; };
bool muse_filter_applies(const mdns_spectra *s, int B, int M);
int muse_filter_ld(int nx);              // row stride of its operands; templates of a chunk it may take: model_ld(nx) + 16 apart
bool launch_muse_filter(mdns_spectra *s, const double *d_model, int ldm, int B, const int *d_rows, int M,
                        const double *d_higher, const double *d_bound, const MuseBandOut &out);
void muse_filter_note(int which);          // 1: a chunk scored again exactly, 2: an exact row for a commit
// src [nx][lds] -> dst [ndata][ld] (only the nx x ndata corner is touched)
bool launch_transpose(const double *d_src, int nx, int ndata, double *d_dst, int ld,
                      bool invert, int lds);
// rows [ndata][ld] -> tiled channel-major replica [ceil(ndata/64)][cols_nx(nx)][64]
bool launch_tile_columns(const double *d_y, int ld, int ndata, int nx, const int *d_rows, double *d_yt);
bool launch_copy_rows(const double *d_src, int nx, int ndata, double *d_dst, int ld,
                      bool invert);
bool launch_pad_model(const double *d_src, int nx, int B, double *d_dst, int ldm);

// launchers implemented in mdns_neighbors.hip
// What a radius computation leaves behind: written by the LAST workgroup of the bootstrap
// kernel to finish (it alone knows that all per-round maxima are final), once into device
// memory for the membership kernel that follows in stream order and once into mapped host
// memory, where the host finds it by polling `seq` -- no event, no copy, no second stream.
struct RegionResult {
	double radius;                 // sqrt(max_b round_sq[b])            cneighbors.c:160-174
	double thresh;                 // smallest T with sqrt(T) >= radius  cneighbors.c:88,109
	unsigned long long seq;        // written last
	unsigned long long pad;
};
struct BootstrapFinish {
	unsigned *counter;             // device; zero outside a launch
	RegionResult *d_res;           // device copy
	RegionResult *h_res;           // mapped, host-coherent copy
	unsigned long long seq;        // the value `seq` takes for this computation
};

// threshold on the squared distance: thresh_sq, or -- when d_res != nullptr -- the one the
// preceding radius computation left on the device
// mdns_region_create_bootstrapped without the wait for the radius (mdns_core.hip)
mdns_region *region_begin_bootstrapped(const double *members, int K, int ndim, const unsigned *packed, int nbootstraps);

// mail != nullptr: `d_cands` and `d_counts` may be host memory mapped into the device -- the kernel
// reads the candidates from there, stores the counts there (no member split) and its last
// workgroup raises *seq_at to `seq`: the host polls instead of copying both ways
struct CountMail { int *ticket; unsigned long long *seq_at; unsigned long long seq; };
bool launch_count_within(const double *d_members, int K, int ndim, double thresh_sq,
                         const RegionResult *d_res, const double *d_cands, int M, int *d_counts,
                         const CountMail *mail = nullptr);
bool launch_bootstrap(const double *d_members, int K, int ndim, const double *d_chosen,
                      int nbootstraps, double *d_round_sq, const BootstrapFinish *finish = nullptr);
// the same with the choice already packed: bit b of d_packed[i] = point i is chosen in round b
bool launch_bootstrap_packed(const double *d_members, int K, int ndim, const unsigned *d_packed,
                             int nbootstraps, double *d_round_sq, const BootstrapFinish *finish);
bool launch_nn_maxsq(const double *d_members, int K, int ndim, double *d_out);
// K6 with the pool in Morton order and tile culling (mdns_k6sort.hip): the same radius, bit for bit
bool bootstrap_sorted_applies(int K, int ndim, int nbootstraps);
bool launch_bootstrap_sorted(const double *d_members, int K, int ndim, const unsigned *d_packed, int nbootstraps,
                             double *d_round_sq, const BootstrapFinish *finish);

// ---- the first batch of a region without a host look in between (mdns_chain.hip) --------
// what the chain kernels need of a region whose radius computation has been launched (mdns_core.hip)
struct RegionView { const double *d_members; int K, ndim; const RegionResult *d_res; };
bool region_view(mdns_region *r, RegionView *out);
static constexpr int kChainMost = 1024;        // proposals per batch at most (radfriendsregion.py:124 uses 1000)
static constexpr int kChainDim = 8;            // dimensions at most
// by-value description of what happens to a proposal between the membership test and the kernel:
// the metric's inverse transform (sdml.py), the unit-cube test (hiermetriclearn.py:113-116), the prior
// transform and the kernel's parameters (mdns_prior)
struct ChainSpec {
	int n, ndim, nparams, limit, identity;
	double mn[kChainDim], mx[kChainDim];                   // extent of the members (radfriendsregion.py:69-70)
	double mean[kChainDim], scale[kChainDim];              // y * scale + mean
	double a[kChainDim], b[kChainDim];
	int pow10[kChainDim], kernel_pow10[kChainDim];
};
// mapped host memory the chain kernels fill: {seq | nkept, B | counts | parameters of the candidates scored}
struct ChainBox {
	unsigned long long seq;
	int nkept, B;
	int counts[kChainMost];
	double params[kChainMost][3];
	double u[kChainMost * kChainDim];                      // in: the raw doubles of the proposals
};
// proposals lo + (hi - lo) u from the raw doubles in `box_dev->u`, radius and threshold from d_res;
// counts to box_dev->counts and d_counts, proposals to d_props [n][ndim]; mail != nullptr: the last
// workgroup raises box->seq (a chain that ends here)
bool launch_box_count(const RegionView &rv, const ChainSpec &spec, ChainBox *box_dev, double *d_props, int *d_counts,
                      const CountMail *mail);
// k_chunk_accept with the candidates taken from the kept proposals (first min(kept, limit) of them)
bool launch_chain_accept(const mdns_spectra *s, const ChainSpec &spec, const double *d_props, const int *d_counts,
                         ChainBox *box_dev, double scale, const int *d_rows_in, int *d_rows_dev, int M,
                         const double *d_higher, int *d_flags, int stamp, const JointTrail &trail, void *d_header);

// optional per-launch event timing (mdns_profile); which: 0 gauss rows, 1 muse rows,
// 2 count-within, 3 nearest-chosen.  Use as:  { ProfileScope ps(which); launch...; }
struct ProfileScope {
	int slot;
	explicit ProfileScope(int which);
	~ProfileScope();
};
// name of the kernel instantiation last launched for class `which` (mdns_profile_kernel)
void note_kernel(int which, const char *fmt, ...);

// Between workgroups of ONE launch: what one hands to another goes through agent-scope atomic stores / loads (or
// atomics) -- they act on memory itself -- and the hand-over is an agent-scope atomic ticket; all the fence has to
// do is wait for this wave's outstanding memory operations.  __threadfence() also writes back and invalidates the
// XCD's whole L2, per workgroup that calls it (measured in round 4: a launch of 256 workgroups with two such
// fences each took 365 us instead of 176; the folded K6 merge 264 us instead of 51).
#ifdef __HIPCC__
// (the workgroup-scope fence orders the compiler's view and emits no instruction; the wait is what makes the stores
// and atomics of THIS wave complete -- acknowledged by memory -- before the ticket that follows: on gfx9 vmcnt counts
// them too.  An agent-scope release fence is exactly `buffer_wbl2 sc1` + this wait.)
__device__ __forceinline__ void handover_release()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void handover_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// Into host memory mapped into the device, for a host that polls `seq`: system-scope stores act on that memory
// itself, so `seq` only has to wait for the stores before it (every thread: mail_store()s, handover_release(), a
// barrier; then one thread: mail_raise()).  A system-scope RELEASE would write back the L2 first -- per mailbox
// a few microseconds that nobody needs: what the host reads is all in the mailbox.
template <class T, class V> __device__ __forceinline__ void mail_store(T *at, V v) { __hip_atomic_store(at, (T) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void mail_raise(unsigned long long *seq_at, unsigned long long seq) { handover_release(); mail_store(seq_at, seq); }
#endif


// ---- the likelihood noise in band form (mdns.h: draw_band / draw_band_commit; kernels in mdns_joint.hip, mdns_like.hip) ----
// every likelihood against its threshold +- (1.01 bound[b] + 1e-12 (|L| + |thr|)): a pair above the band is beaten whatever
// the noise (clear[b] = 1), a pair inside it is listed for the host, which alone makes the exact deviates
#ifdef __HIPCC__
static constexpr int kBandCap = 4096;
struct BandBox {                   // mapped host memory
	unsigned long long seq;
	int npairs, pad;
	int status[MDNS_JOINT_MAX_BATCH];
	int pair_b[kBandCap], pair_k[kBandCap];
	double pair_L[kBandCap], pair_thr[kBandCap];
};
struct BandScratch {               // device memory
	int counter, ticket;           // listed pairs; workgroups of the band pass that are done (zero between launches)
	int clear[MDNS_JOINT_MAX_BATCH], maybe[MDNS_JOINT_MAX_BATCH];
	int pair_b[kBandCap], pair_k[kBandCap];
	double pair_L[kBandCap], pair_thr[kBandCap];
};

// what the host needs of a band pass, into mapped memory (`seq` last), and the scratch ready for the next chunk:
// by one workgroup that knows every vote is in (a kernel of its own behind the pass, or the pass's last workgroup)
__device__ __forceinline__ void band_publish(BandScratch *__restrict__ sc, int B, BandBox *__restrict__ box, unsigned long long seq)
{
	const int n = __hip_atomic_load(&sc->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const int m = n < kBandCap ? n : kBandCap;
	for (int b = threadIdx.x; b < B; b += (int) blockDim.x) {
		const int cl = __hip_atomic_load(&sc->clear[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const int mb = __hip_atomic_load(&sc->maybe[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		mail_store(&box->status[b], cl ? 1 : (mb ? 2 : 0));
		sc->clear[b] = 0; sc->maybe[b] = 0;
	}
	for (int t = threadIdx.x; t < m; t += (int) blockDim.x) {
		mail_store(&box->pair_b[t], __hip_atomic_load(&sc->pair_b[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		mail_store(&box->pair_k[t], __hip_atomic_load(&sc->pair_k[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		mail_store(&box->pair_L[t], __hip_atomic_load(&sc->pair_L[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		mail_store(&box->pair_thr[t], __hip_atomic_load(&sc->pair_thr[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	}
	handover_release();
	__syncthreads();
	if (threadIdx.x != 0) return;
	sc->counter = 0;
	sc->ticket = 0;
	mail_store(&box->npairs, n);
	mail_raise(&box->seq, seq);
}

// one (candidate, data set) pair of a band pass (votes and pairs through agent-scope stores: whoever publishes may be
// another workgroup of the same launch)
__device__ __forceinline__ void band_vote(BandScratch *__restrict__ sc, int b, int k, double v, double thr, double bnd)
{
	const double band = 1.01 * bnd + 1e-12 * (fabs(v) + fabs(thr));
	if (v > thr + band) __hip_atomic_store(&sc->clear[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (v >= thr - band) {
		__hip_atomic_store(&sc->maybe[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const int at = atomicAdd(&sc->counter, 1);
		if (at < kBandCap) {
			__hip_atomic_store(&sc->pair_b[at], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&sc->pair_k[at], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&sc->pair_L[at], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&sc->pair_thr[at], thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}
// a scoring kernel that votes and publishes by itself (launch_muse_rows): sc == nullptr: it does neither
struct MuseBandFused { BandScratch *sc; BandBox *box; unsigned long long seq; const double *higher; const double *bound; int *status_zero; };
#endif

// smallest double T with sqrt(T) >= r, so that  sqrt(d) < r  <=>  d < T  for every d >= 0
double sqrt_threshold(double r);

}  // namespace mdns
