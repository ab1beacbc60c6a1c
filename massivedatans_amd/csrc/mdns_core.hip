// libmdns_hip.so -- context, memory, resident spectra and the host-pointer entry points.
// Kernels live in mdns_like.hip and mdns_neighbors.hip.
#include "mdns_internal.h"
#include <time.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace mdns {

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static char g_error[512] = "";

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
	if (getenv("MDNS_VERBOSE")) fprintf(stderr, "[mdns] %s\n", g_error);
}

bool hip_ok(hipError_t e, const char *what, const char *file, int line)
{
	if (e == hipSuccess) return true;
	set_error("%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
	return false;
}

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
static Context g_ctx;
static bool g_ctx_ready = false;
static std::mutex g_ctx_mutex;

static int init_locked(int device)
{
	if (g_ctx_ready && (device < 0 || device == g_ctx.device)) return 0;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		set_error("no HIP device visible: libmdns_hip has no CPU path");
		return 1;
	}
	if (device < 0) {
		const char *e = getenv("MDNS_DEVICE");
		if (e && *e) device = atoi(e);
		else if ((e = getenv("LOCAL_RANK")) && *e) device = atoi(e) % count;
		else device = 0;
	}
	if (device >= count) {
		set_error("device %d requested but only %d visible", device, count);
		return 1;
	}
	if (g_ctx_ready) {
		// One process drives ONE GPU (include/mdns.h): spectra, regions, result slots, pooled
		// buffers and events all live on the first device; handing them to another one would use
		// them across devices.
		set_error("mdns_init(%d): this process already runs on device %d; one process drives one GPU "
		          "(start one process per GPU)", device, g_ctx.device);
		return 1;
	}
	if (!MDNS_HIP(hipSetDevice(device))) return 1;
	hipDeviceProp_t prop;
	if (!MDNS_HIP(hipGetDeviceProperties(&prop, device))) return 1;
	g_ctx.device = device;
	g_ctx.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (!MDNS_HIP(hipStreamCreateWithFlags(&g_ctx.own_stream, hipStreamNonBlocking))) return 1;
	g_ctx.stream = g_ctx.own_stream;
	g_ctx_ready = true;
	return 0;
}

Context *ctx()
{
	std::lock_guard<std::mutex> lock(g_ctx_mutex);
	if (!g_ctx_ready && init_locked(-1) != 0) return nullptr;
	// the runtime's current device is per thread: make sure it is ours
	(void) hipSetDevice(g_ctx.device);
	return &g_ctx;
}

void *device_scratch(size_t bytes)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (bytes <= c->d_ws_bytes) return c->d_ws;
	if (c->d_ws) {
		(void) hipStreamSynchronize(c->stream);
		(void) hipFree(c->d_ws);
		c->d_ws = nullptr; c->d_ws_bytes = 0;
	}
	size_t cap = bytes + bytes / 2 + 4096;
	if (!MDNS_HIP(hipMalloc(&c->d_ws, cap))) return nullptr;
	c->d_ws_bytes = cap;
	return c->d_ws;
}

void *mask_scratch(size_t bytes)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (bytes <= c->d_mask_bytes) return c->d_mask;
	if (c->d_mask) {
		(void) hipStreamSynchronize(c->stream);
		(void) hipFree(c->d_mask);
		c->d_mask = nullptr; c->d_mask_bytes = 0;
	}
	size_t cap = bytes + bytes / 2 + 4096;
	if (!MDNS_HIP(hipMalloc(&c->d_mask, cap))) return nullptr;
	c->d_mask_bytes = cap;
	return c->d_mask;
}

void *pinned_scratch(size_t bytes)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (bytes <= c->h_pin_bytes) return c->h_pin;
	if (c->h_pin) {
		(void) hipStreamSynchronize(c->stream);
		(void) hipHostFree(c->h_pin);
		c->h_pin = nullptr; c->h_pin_bytes = 0;
	}
	size_t cap = bytes + bytes / 2 + 4096;
	if (!MDNS_HIP(hipHostMalloc(&c->h_pin, cap, hipHostMallocDefault))) return nullptr;
	c->h_pin_bytes = cap;
	return c->h_pin;
}

// smallest non-negative double T with sqrt(T) >= r.  sqrt on doubles is correctly rounded
// and monotone, so  sqrt(d) < r  <=>  d < T.  Binary search over the bit patterns of the
// non-negative doubles (their integer order is their numeric order).
double sqrt_threshold(double r)
{
	if (std::isnan(r)) return r;             // d < NaN is false, as sqrt(d) < NaN is
	if (!(r > 0)) return 0.0;                // sqrt(d) >= 0 >= r: never inside
	if (std::isinf(r)) return r;             // every finite d is inside
	uint64_t lo = 0, hi;                     // predicate false at lo (sqrt(0) = 0 < r)
	double inf = INFINITY;
	memcpy(&hi, &inf, 8);                    // predicate true at +inf
	while (hi - lo > 1) {
		uint64_t mid = lo + (hi - lo) / 2;
		double t;
		memcpy(&t, &mid, 8);
		if (std::sqrt(t) >= r) hi = mid; else lo = mid;
	}
	double T;
	memcpy(&T, &hi, 8);
	return T;
}

// ---------------------------------------------------------------------------------------
// per-launch event timing
// ---------------------------------------------------------------------------------------
struct TimedLaunch { int which; hipEvent_t start, stop; };
static long long g_prof_seen[4];
static int g_prof_every = 1;
static int g_profiling = 0;                   // bit k: time launches of kernel class k
static std::vector<TimedLaunch> g_pending;
static std::vector<hipEvent_t> g_event_pool;
static long long g_prof_count[4] = {0, 0, 0, 0};
static double g_prof_ms[4] = {0, 0, 0, 0};

static hipEvent_t pooled_event()
{
	if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) return nullptr;
	return e;
}

static void drain_pending()
{
	for (TimedLaunch &t : g_pending) {
		float ms = 0;
		if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
			g_prof_count[t.which] += 1;
			g_prof_ms[t.which] += ms;
		}
		g_event_pool.push_back(t.start);
		g_event_pool.push_back(t.stop);
	}
	g_pending.clear();
}

ProfileScope::ProfileScope(int which) : slot(-1)
{
	if (!(g_profiling & (1 << which)) || !g_ctx_ready) return;
	if (g_prof_seen[which]++ % g_prof_every != 0) return;      // sampled: every n-th launch of a class
	TimedLaunch t{which, pooled_event(), pooled_event()};
	if (!t.start || !t.stop) return;
	(void) hipEventRecord(t.start, g_ctx.stream);
	g_pending.push_back(t);
	slot = (int) g_pending.size() - 1;
}

ProfileScope::~ProfileScope()
{
	if (slot < 0) return;
	(void) hipEventRecord(g_pending[slot].stop, g_ctx.stream);
	if (g_pending.size() >= 4096) drain_pending();      // bound the number of live events
}

template <typename T>
static bool grow(T **p, size_t *cap, size_t need)
{
	if (need <= *cap) return true;
	Context *c = ctx();
	if (!c) return false;
	if (*p) { (void) hipStreamSynchronize(c->stream); (void) hipFree(*p); *p = nullptr; *cap = 0; }
	size_t n = need + need / 2 + 64;
	if (!MDNS_HIP(hipMalloc((void **) p, n * sizeof(T)))) return false;
	*cap = n;
	return true;
}

}  // namespace mdns

using namespace mdns;

// ---------------------------------------------------------------------------------------
// library state
// ---------------------------------------------------------------------------------------
extern "C" int mdns_abi_version(void) { return 1; }
extern "C" const char *mdns_last_error(void) { return g_error; }

extern "C" int mdns_device_count(void)
{
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess) return 0;
	return count;
}

extern "C" int mdns_init(int device)
{
	std::lock_guard<std::mutex> lock(g_ctx_mutex);
	return init_locked(device);
}

// ---------------------------------------------------------------------------------------
// raw device helpers
// ---------------------------------------------------------------------------------------
extern "C" void *mdns_dev_alloc(size_t bytes)
{
	if (!ctx()) return nullptr;
	void *p = nullptr;
	if (!MDNS_HIP(hipMalloc(&p, bytes ? bytes : 8))) return nullptr;
	return p;
}
extern "C" void mdns_dev_free(void *p) { if (p && ctx()) (void) hipFree(p); }
extern "C" int mdns_h2d(void *dst, const void *src, size_t bytes)
{
	Context *c = ctx();
	if (!c) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream))) return 1;
	return MDNS_HIP(hipStreamSynchronize(c->stream)) ? 0 : 1;
}
extern "C" int mdns_d2h(void *dst, const void *src, size_t bytes)
{
	Context *c = ctx();
	if (!c) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream))) return 1;
	return MDNS_HIP(hipStreamSynchronize(c->stream)) ? 0 : 1;
}
extern "C" int mdns_d2d(void *dst, const void *src, size_t bytes)
{
	Context *c = ctx();
	if (!c) return 1;
	return MDNS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream)) ? 0 : 1;
}
extern "C" int mdns_sync(void)
{
	Context *c = ctx();
	if (!c) return 1;
	return MDNS_HIP(hipStreamSynchronize(c->stream)) ? 0 : 1;
}
extern "C" int mdns_set_stream(void *hip_stream)
{
	Context *c = ctx();
	if (!c) return 1;
	c->stream = hip_stream ? (hipStream_t) hip_stream : c->own_stream;
	return 0;
}
extern "C" void *mdns_get_stream(void)
{
	Context *c = ctx();
	return c ? (void *) c->stream : nullptr;
}
extern "C" void *mdns_event_create(void)
{
	if (!ctx()) return nullptr;
	hipEvent_t ev;
	if (!MDNS_HIP(hipEventCreate(&ev))) return nullptr;
	return (void *) ev;
}
extern "C" void mdns_event_destroy(void *ev) { if (ev) (void) hipEventDestroy((hipEvent_t) ev); }
extern "C" int mdns_event_record(void *ev)
{
	Context *c = ctx();
	if (!c || !ev) return 1;
	return MDNS_HIP(hipEventRecord((hipEvent_t) ev, c->stream)) ? 0 : 1;
}
extern "C" double mdns_event_elapsed_ms(void *a, void *b)
{
	if (!a || !b) return NAN;
	if (!MDNS_HIP(hipEventSynchronize((hipEvent_t) b))) return NAN;
	float ms = 0;
	if (!MDNS_HIP(hipEventElapsedTime(&ms, (hipEvent_t) a, (hipEvent_t) b))) return NAN;
	return (double) ms;
}

extern "C" int mdns_profile(int classes)
{
	if (!ctx()) return 1;
	drain_pending();
	for (int k = 0; k < 4; k++)
		if (classes & (1 << k)) { g_prof_count[k] = 0; g_prof_ms[k] = 0; g_prof_seen[k] = 0; }
	g_profiling = classes & 15;
	return 0;
}
extern "C" int mdns_profile_every(int n)
{
	g_prof_every = n > 1 ? n : 1;
	return 0;
}
static char g_kernel_name[4][64];
void mdns::note_kernel(int which, const char *fmt, ...)
{
	if (which < 0 || which > 3) return;
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_kernel_name[which], sizeof g_kernel_name[which], fmt, ap);
	va_end(ap);
}
extern "C" const char *mdns_profile_kernel(int which)
{
	return which >= 0 && which <= 3 ? g_kernel_name[which] : "";
}
extern "C" int mdns_profile_read(int which, long long *launches, double *total_ms)
{
	if (!ctx() || which < 0 || which > 3) return 1;
	drain_pending();
	if (launches) *launches = g_prof_count[which];
	if (total_ms) *total_ms = g_prof_ms[which];
	return 0;
}

// ---------------------------------------------------------------------------------------
// resident spectra
// ---------------------------------------------------------------------------------------
static bool upload_rows(const double *h_src, int ndata, int nx, int layout, double *d_dst, int ld,
                        bool invert)
{
	// stage the host array as it is, then re-lay it on the device into [ndata, ld] rows
	Context *c = ctx();
	const size_t n = (size_t) ndata * nx;
	double *d_tmp = nullptr;
	if (!MDNS_HIP(hipMalloc((void **) &d_tmp, (n ? n : 1) * sizeof(double)))) return false;
	bool ok = MDNS_HIP(hipMemcpyAsync(d_tmp, h_src, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
	if (ok) {
		if (layout == MDNS_LAYOUT_CHANNEL_MAJOR) ok = launch_transpose(d_tmp, nx, ndata, d_dst, ld, invert, ndata);
		else ok = launch_copy_rows(d_tmp, nx, ndata, d_dst, ld, invert);
	}
	ok = MDNS_HIP(hipStreamSynchronize(c->stream)) && ok;
	(void) hipFree(d_tmp);
	return ok;
}

extern "C" mdns_spectra *mdns_spectra_create(const double *x, const double *y, const double *v,
                                             int ndata, int nx, int layout)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (ndata < 0 || nx < 0 || !y || (layout != 0 && layout != 1)) {
		set_error("mdns_spectra_create: bad arguments (ndata=%d nx=%d layout=%d)", ndata, nx, layout);
		return nullptr;
	}
	mdns_spectra *s = new mdns_spectra();
	s->ndata = ndata; s->nx = nx; s->ld = (nx + 1) & ~1;
	// +2 doubles of slack: the row kernels read whole 16-byte pairs
	const size_t elems = (size_t) ndata * s->ld + 2;
	bool ok = MDNS_HIP(hipMalloc((void **) &s->d_y, elems * sizeof(double)));
	ok = ok && MDNS_HIP(hipMemsetAsync(s->d_y, 0, elems * sizeof(double), c->stream));
	ok = ok && upload_rows(y, ndata, nx, layout, s->d_y, s->ld, false);
	if (ok && !v && nx > 0 && ndata > 0) {
		// K1 also keeps a channel-major replica, in tiles of 64 spectra, for candidate batches
		// (the tiling kernel writes every element, padding included)
		s->ldT = ((ndata + 63) / 64) * 64;
		const size_t telems = (size_t) cols_nx(nx) * s->ldT;
		ok = MDNS_HIP(hipMalloc((void **) &s->d_yT, telems * sizeof(double)));
		ok = ok && launch_tile_columns(s->d_y, s->ld, ndata, nx, nullptr, s->d_yT);
		ok = ok && MDNS_HIP(hipMalloc((void **) &s->d_ysq, (size_t) ndata * sizeof(double)));
		ok = ok && launch_row_sumsq(s->d_y, s->ld, nx, ndata, s->d_ysq);
		ok = ok && MDNS_HIP(hipStreamSynchronize(c->stream));
	}
	if (ok && v) {
		ok = MDNS_HIP(hipMalloc((void **) &s->d_w, elems * sizeof(double)));
		ok = ok && MDNS_HIP(hipMemsetAsync(s->d_w, 0, elems * sizeof(double), c->stream));
		ok = ok && upload_rows(v, ndata, nx, layout, s->d_w, s->ld, true);
	}
	if (ok && x) {
		ok = MDNS_HIP(hipMalloc((void **) &s->d_x, (nx ? nx : 1) * sizeof(double)));
		ok = ok && MDNS_HIP(hipMemcpyAsync(s->d_x, x, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
		ok = ok && MDNS_HIP(hipStreamSynchronize(c->stream));
	}
	if (!ok) { mdns_spectra_destroy(s); return nullptr; }
	return s;
}

extern "C" void mdns_spectra_destroy(mdns_spectra *s)
{
	if (!s) return;
	Context *c = ctx();
	if (c) (void) hipStreamSynchronize(c->stream);
	void *bufs[] = {s->d_y, s->d_yT, s->d_w, s->d_x, s->d_model, s->d_params, s->d_rows, s->d_out, s->d_sel, s->d_ysq};
	for (void *b : bufs) if (b) (void) hipFree(b);
	if (s->d_yG) (void) hipFree(s->d_yG);
	if (s->d_selG) (void) hipFree(s->d_selG);
	if (s->d_model_g) (void) hipFree(s->d_model_g);
	if (s->d_fyw) (void) hipFree(s->d_fyw);
	if (s->d_fyw_t) (void) hipFree(s->d_fyw_t);
	if (s->d_fw_t) (void) hipFree(s->d_fw_t);
	if (s->d_fa) (void) hipFree(s->d_fa);
	if (s->d_fw && s->fw_owned) (void) hipFree(s->d_fw);
	delete s;
}
extern "C" int mdns_spectra_ndata(const mdns_spectra *s) { return s ? s->ndata : -1; }
extern "C" int mdns_spectra_nx(const mdns_spectra *s) { return s ? s->nx : -1; }

// ---------------------------------------------------------------------------------------
// batched scoring, device pointers
// ---------------------------------------------------------------------------------------
static bool check_batch(const mdns_spectra *s, int B, int M, const char *who)
{
	if (!s) { set_error("%s: null spectra handle", who); return false; }
	if (B < 0 || M < 0 || M > s->ndata) {
		set_error("%s: bad sizes B=%d M=%d (ndata=%d)", who, B, M, s->ndata);
		return false;
	}
	return true;
}

namespace mdns {
bool ensure_model(mdns_spectra *s, size_t doubles) { return grow(&s->d_model, &s->model_cap, doubles); }
bool ensure_selection(mdns_spectra *s, size_t doubles) { return grow(&s->d_sel, &s->sel_cap, doubles); }

// K1 with the lane kernel whatever the shape (one lane per spectrum, channels summed in
// ascending order): L[B, M] to d_Lout.  The likelihood of a (candidate, spectrum) pair computed
// here does not depend on B, M or the tile shape.
int gauss_loglike_cols_dev(mdns_spectra *s, const double *d_params, int B, double noise_level,
                           const int *d_row_ids, int M, double *d_Lout)
{
	if (!s->d_yT) { set_error("the lane kernel needs the channel-major replica (spectra without variances)"); return 1; }
	const double scale = -0.5 / (noise_level * noise_level);
	const int bt = gauss_cols_tile(M, B);
	if (!ensure_model(s, (size_t) cols_nx(s->nx) * (B + bt))) return 1;
	if (!launch_gauss_model_t(s->d_x, s->nx, d_params, B, bt, s->d_model)) return 1;
	// A selection is first copied into a compact replica (coalesced row reads, one pass)
	// when the lane kernel would otherwise gather its columns once per candidate tile:
	// many tiles, or a sparse selection (measured: 1 000 of 10 000 spectra, B = 256: 59 us
	// gathering in the kernel).
	const bool sparse = (size_t) M * 8 < (size_t) s->ndata;
	if (d_row_ids && (B >= 128 || sparse)) {
		if (!ensure_selection(s, (size_t) ((M + 63) / 64) * 64 * cols_nx(s->nx))) return 1;
		if (!launch_tile_columns(s->d_y, s->ld, M, s->nx, d_row_ids, s->d_sel)) return 1;
		return launch_gauss_cols(s, s->d_sel, s->d_model, bt, B, scale, nullptr, M, d_Lout) ? 0 : 1;
	}
	return launch_gauss_cols(s, s->d_yT, s->d_model, bt, B, scale, d_row_ids, M, d_Lout) ? 0 : 1;
}
}  // namespace mdns

extern "C" int mdns_gauss_loglike_batch_dev(mdns_spectra *s, const double *d_params, int B,
                                            double noise_level, const int *d_row_ids, int M,
                                            double *d_Lout)
{
	if (!ctx() || !check_batch(s, B, M, "mdns_gauss_loglike_batch_dev")) return 1;
	if (!s->d_x) { set_error("spectra were created without a wavelength grid"); return 1; }
	if (B == 0 || M == 0) return 0;
	const double scale = -0.5 / (noise_level * noise_level);
	// Two kernels, chosen by shape only (never by data), so a run is reproducible:
	//  * batches of 12+ candidates on dense selections (at least one spectrum in eight), and
	//    batches of 32+ on any selection, are arithmetic-bound: one LANE per spectrum on the
	//    channel-major replica, templates as scalar operands (k_gauss_cols);
	//  * few candidates, or sparse selections, are bandwidth/latency-bound: one WAVE per
	//    spectrum row, reading exactly the selected rows once (k_gauss_rows).
	// Measured on MI355X, 10 000 x 200, all spectra: rows 6.8 / 8.7 / 12.3 / 18.9 us at
	// B = 1 / 4 / 8 / 16, cols 16.9 / 17.3 / 19.2 us at B = 8 / 16 / 32; at 10 % of the spectra
	// and B = 1024: cols 59 us, rows 688 us.
	static const char *forced = getenv("MDNS_K1_PATH");      // "rows" | "cols": experiments only
	const bool dense = (size_t) M * 8 >= (size_t) s->ndata;
	// dense: the lane kernel pays a ~17 us pipeline fill, the row kernel ~6.5 us plus more per
	// eval; they cross near 1.5e5 evals (B = 12 at 10 000 spectra, B = 2 at 100 000: 27 vs 38 us)
	bool use_cols = s->d_yT && ((dense && B >= 2 && (long long) M * B >= 150000) || B >= 32);
	if (forced && !strcmp(forced, "rows")) use_cols = false;
	if (forced && !strcmp(forced, "cols") && s->d_yT) use_cols = true;
	if (use_cols) return gauss_loglike_cols_dev(s, d_params, B, noise_level, d_row_ids, M, d_Lout);
	const int ldm = model_ld(s->nx);
	if (!grow(&s->d_model, &s->model_cap, (size_t) B * ldm)) return 1;
	if (!launch_gauss_model(s->d_x, s->nx, d_params, B, s->d_model, ldm)) return 1;
	return launch_gauss_rows(s, s->d_model, ldm, B, scale, d_row_ids, M, d_Lout) ? 0 : 1;
}

extern "C" int mdns_muse_loglike_batch_dev(mdns_spectra *s, const double *d_ypred, int B,
                                           const int *d_row_ids, int M, double *d_Lout)
{
	if (!ctx() || !check_batch(s, B, M, "mdns_muse_loglike_batch_dev")) return 1;
	if (!s->d_w) { set_error("spectra were created without variances"); return 1; }
	if (B == 0 || M == 0) return 0;
	const int ldm = model_ld(s->nx);
	if (!grow(&s->d_model, &s->model_cap, (size_t) B * ldm)) return 1;
	if (!launch_pad_model(d_ypred, s->nx, B, s->d_model, ldm)) return 1;
	return launch_muse_rows(s, s->d_model, ldm, B, d_row_ids, M, d_Lout) ? 0 : 1;
}

extern "C" int mdns_muse3_loglike_batch_dev(mdns_spectra *s, const double *d_params, int B,
                                            const int *d_row_ids, int M, double *d_Lout)
{
	if (!ctx() || !check_batch(s, B, M, "mdns_muse3_loglike_batch_dev")) return 1;
	if (!s->d_w || !s->d_x) { set_error("spectra need variances and a wavelength grid"); return 1; }
	if (B == 0 || M == 0) return 0;
	const int ldm = model_ld(s->nx);
	if (!grow(&s->d_model, &s->model_cap, (size_t) B * ldm)) return 1;
	if (!launch_muse3_model(s->d_x, s->nx, d_params, B, s->d_model, ldm)) return 1;
	return launch_muse_rows(s, s->d_model, ldm, B, d_row_ids, M, d_Lout) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------
// batched scoring, host pointers
// ---------------------------------------------------------------------------------------
typedef int (*dev_batch_fn)(mdns_spectra *, const double *, int, const int *, int, double *, double);

static int host_batch(mdns_spectra *s, const double *params, int B, int nparam,
                      const int *row_ids, int M, double *Lout, double extra, dev_batch_fn fn,
                      const char *who)
{
	Context *c = ctx();
	if (!c || !check_batch(s, B, M, who)) return 1;
	if (B == 0 || M == 0) return 0;
	if (row_ids) {
		for (int k = 0; k < M; k++)
			if (row_ids[k] < 0 || row_ids[k] >= s->ndata) {
				set_error("%s: row_ids[%d]=%d outside [0,%d)", who, k, row_ids[k], s->ndata);
				return 1;
			}
	}
	const size_t np = (size_t) B * nparam, no = (size_t) B * M;
	if (!grow(&s->d_params, &s->params_cap, np)) return 1;
	if (!grow(&s->d_out, &s->out_cap, no)) return 1;
	if (row_ids && !grow(&s->d_rows, &s->rows_cap, (size_t) M)) return 1;
	// stage through pinned memory so both copies are truly asynchronous
	const size_t stage_bytes = np * 8 + (row_ids ? (size_t) M * 4 : 0);
	const size_t out_off = (stage_bytes + 15) & ~(size_t) 15;
	char *pin = (char *) pinned_scratch(out_off + no * 8);
	if (!pin) return 1;
	memcpy(pin, params, np * 8);
	bool ok = MDNS_HIP(hipMemcpyAsync(s->d_params, pin, np * 8, hipMemcpyHostToDevice, c->stream));
	if (ok && row_ids) {
		memcpy(pin + np * 8, row_ids, (size_t) M * 4);
		ok = MDNS_HIP(hipMemcpyAsync(s->d_rows, pin + np * 8, (size_t) M * 4, hipMemcpyHostToDevice, c->stream));
	}
	if (!ok) return 1;
	if (fn(s, s->d_params, B, row_ids ? s->d_rows : nullptr, M, s->d_out, extra) != 0) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(pin + out_off, s->d_out, no * 8, hipMemcpyDeviceToHost, c->stream))) return 1;
	if (!MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
	memcpy(Lout, pin + out_off, no * 8);
	return 0;
}

static int fn_gauss(mdns_spectra *s, const double *p, int B, const int *r, int M, double *o, double noise)
{ return mdns_gauss_loglike_batch_dev(s, p, B, noise, r, M, o); }
static int fn_muse(mdns_spectra *s, const double *p, int B, const int *r, int M, double *o, double)
{ return mdns_muse_loglike_batch_dev(s, p, B, r, M, o); }
static int fn_muse3(mdns_spectra *s, const double *p, int B, const int *r, int M, double *o, double)
{ return mdns_muse3_loglike_batch_dev(s, p, B, r, M, o); }

extern "C" int mdns_gauss_loglike_batch(mdns_spectra *s, const double *params, int B,
                                        double noise_level, const int *row_ids, int M, double *Lout)
{
	return host_batch(s, params, B, 3, row_ids, M, Lout, noise_level, fn_gauss, "mdns_gauss_loglike_batch");
}
extern "C" int mdns_muse_loglike_batch(mdns_spectra *s, const double *ypred, int B,
                                       const int *row_ids, int M, double *Lout)
{
	return host_batch(s, ypred, B, s ? s->nx : 0, row_ids, M, Lout, 0, fn_muse, "mdns_muse_loglike_batch");
}
extern "C" int mdns_muse3_loglike_batch(mdns_spectra *s, const double *params, int B,
                                        const int *row_ids, int M, double *Lout)
{
	return host_batch(s, params, B, 5, row_ids, M, Lout, 0, fn_muse3, "mdns_muse3_loglike_batch");
}

// ---------------------------------------------------------------------------------------
// drop-in likelihood entry points (reference argument lists)
// ---------------------------------------------------------------------------------------
struct Registered {
	const void *yy; const void *vv; int ndata; int nx; mdns_spectra *s;
};
static std::vector<Registered> g_registered;

static mdns_spectra *find_registered(const void *yy, const void *vv, int ndata, int nx)
{
	for (const Registered &r : g_registered)
		if (r.yy == yy && r.ndata == ndata && r.nx == nx && (vv == nullptr || r.vv == vv)) return r.s;
	return nullptr;
}

extern "C" int mdns_register_spectra(const void *yy, const void *vv, int ndata, int nx)
{
	mdns_unregister_spectra(yy);
	mdns_spectra *s = mdns_spectra_create(nullptr, (const double *) yy, (const double *) vv, ndata, nx,
	                                      MDNS_LAYOUT_CHANNEL_MAJOR);
	if (!s) return 1;
	g_registered.push_back(Registered{yy, vv, ndata, nx, s});
	return 0;
}

extern "C" int mdns_unregister_spectra(const void *yy)
{
	for (size_t i = 0; i < g_registered.size(); i++)
		if (g_registered[i].yy == yy) {
			mdns_spectra_destroy(g_registered[i].s);
			g_registered.erase(g_registered.begin() + i);
			return 0;
		}
	return 1;
}

// MDNS_ASSUME_STATIC_SPECTRA=1: the drop-in like() calls register every spectra array they
// see (by pointer and shape) instead of uploading it on each call -- for hosts such as the
// reference's sample.py / musefuse.py, which load their data once and never modify it
static bool assume_static_spectra()
{
	static const char *e = getenv("MDNS_ASSUME_STATIC_SPECTRA");
	return e && *e && *e != '0';
}

// mask (C bool per data set) -> ascending row ids; returns the count
static int mask_to_rows(const void *data_mask, int ndata, std::vector<int> &rows)
{
	const unsigned char *m = (const unsigned char *) data_mask;
	rows.clear();
	for (int i = 0; i < ndata; i++) if (m[i]) rows.push_back(i);
	return (int) rows.size();
}

extern "C" int mdns_gauss_like(const void *xp, const void *yyp, int ndata, int nx,
                               double A, double mu, double sig, double noise_level,
                               const void *data_maskp, void *Loutp)
{
	if (!ctx()) return 1;
	std::vector<int> rows;
	const int M = mask_to_rows(data_maskp, ndata, rows);
	if (M == 0 || nx <= 0) return 0;
	mdns_spectra *s = find_registered(yyp, nullptr, ndata, nx);
	if (!s && assume_static_spectra() && mdns_register_spectra(yyp, nullptr, ndata, nx) == 0)
		s = find_registered(yyp, nullptr, ndata, nx);
	const bool temporary = (s == nullptr);
	if (temporary) {
		s = mdns_spectra_create((const double *) xp, (const double *) yyp, nullptr, ndata, nx,
		                        MDNS_LAYOUT_CHANNEL_MAJOR);
		if (!s) return 1;
	} else {
		// the grid belongs to the call, not to the registration (clike.c:35 takes x every call)
		Context *c = ctx();
		if (!s->d_x && !MDNS_HIP(hipMalloc((void **) &s->d_x, nx * sizeof(double)))) return 1;
		if (!MDNS_HIP(hipMemcpyAsync(s->d_x, xp, nx * sizeof(double), hipMemcpyHostToDevice, c->stream))) return 1;
		if (!MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
	}
	const double params[3] = {A, mu, sig};
	std::vector<double> L((size_t) M);
	int rc = mdns_gauss_loglike_batch(s, params, 1, noise_level, M == ndata ? nullptr : rows.data(), M, L.data());
	if (temporary) mdns_spectra_destroy(s);
	if (rc != 0) return rc;
	// clike.c:72 accumulates the raw sums  sum_j ((m_j - y_ij)/noise)^2 = -2 * loglike
	double *Lout = (double *) Loutp;
	for (int k = 0; k < M; k++) Lout[k] += -2.0 * L[k];
	return 0;
}

extern "C" int mdns_muse_like(const void *yyp, const void *vvp, const void *ypredp,
                              const void *data_maskp, int ndata, int nx, void *Loutp)
{
	if (!ctx()) return 1;
	std::vector<int> rows;
	const int M = mask_to_rows(data_maskp, ndata, rows);
	if (M == 0) return 0;
	mdns_spectra *s = find_registered(yyp, vvp, ndata, nx);
	if (!s && assume_static_spectra() && mdns_register_spectra(yyp, vvp, ndata, nx) == 0)
		s = find_registered(yyp, vvp, ndata, nx);
	const bool temporary = (s == nullptr);
	if (temporary) {
		s = mdns_spectra_create(nullptr, (const double *) yyp, (const double *) vvp, ndata, nx,
		                        MDNS_LAYOUT_CHANNEL_MAJOR);
		if (!s) return 1;
	}
	std::vector<double> L((size_t) M);
	int rc = mdns_muse_loglike_batch(s, (const double *) ypredp, 1, M == ndata ? nullptr : rows.data(), M, L.data());
	if (temporary) mdns_spectra_destroy(s);
	if (rc != 0) return rc;
	double *Lout = (double *) Loutp;          // not compacted; unmasked entries untouched
	for (int k = 0; k < M; k++) Lout[rows[k]] = L[k];
	return 0;
}

// ---------------------------------------------------------------------------------------
// drop-in neighbourhood entry points
// ---------------------------------------------------------------------------------------
extern "C" int mdns_count_within_dev(const double *d_members, int K, int ndim, double maxdistance,
                                     const double *d_cands, int M, int *d_counts)
{
	Context *c = ctx();
	if (!c) return 1;
	if (K < 0 || M < 0 || ndim <= 0) { set_error("mdns_count_within_dev: bad sizes"); return 1; }
	if (M == 0) return 0;
	if (K == 0) return MDNS_HIP(hipMemsetAsync(d_counts, 0, (size_t) M * sizeof(int), c->stream)) ? 0 : 1;
	return launch_count_within(d_members, K, ndim, sqrt_threshold(maxdistance), nullptr, d_cands, M, d_counts) ? 0 : 1;
}

extern "C" int mdns_bootstrap_round_maxsq_dev(const double *d_members, int K, int ndim,
                                              const double *d_chosen, int nbootstraps,
                                              double *d_round_sq)
{
	Context *c = ctx();
	if (!c) return 1;
	if (K < 0 || nbootstraps < 0 || ndim <= 0) { set_error("mdns_bootstrap_round_maxsq_dev: bad sizes"); return 1; }
	if (nbootstraps == 0) return 0;
	// k_pack_chosen clears d_round_sq itself; an empty pool still has to report zeros
	if (K == 0) return MDNS_HIP(hipMemsetAsync(d_round_sq, 0, (size_t) nbootstraps * sizeof(double), c->stream)) ? 0 : 1;
	return launch_bootstrap(d_members, K, ndim, d_chosen, nbootstraps, d_round_sq) ? 0 : 1;
}

// upload a few host arrays into the context's device scratch (16-byte aligned slots) and
// hand back the device pointers; one H2D per array on the library stream
static bool stage_in(const void *const *src, const size_t *bytes, int n, size_t extra_bytes, char **d_ptrs,
                     char **d_extra)
{
	Context *c = ctx();
	size_t total = 0;
	for (int i = 0; i < n; i++) total += (bytes[i] + 255) & ~(size_t) 255;
	char *base = (char *) device_scratch(total + extra_bytes + 256);
	if (!base) return false;
	size_t off = 0;
	for (int i = 0; i < n; i++) {
		d_ptrs[i] = base + off;
		if (bytes[i] && !MDNS_HIP(hipMemcpyAsync(d_ptrs[i], src[i], bytes[i], hipMemcpyHostToDevice, c->stream)))
			return false;
		off += (bytes[i] + 255) & ~(size_t) 255;
	}
	*d_extra = base + off;
	return true;
}

extern "C" int mdns_count_within_distance_of(const void *xx, int nsamples, int ndim, double maxdistance,
                                             const void *yy, int nothers, void *outp, const int countmax)
{
	Context *c = ctx();
	if (!c) return 1;
	if (nothers <= 0) return 0;
	if (nsamples <= 0) return 0;               // no member: nothing is incremented
	if (ndim <= 0) { set_error("mdns_count_within_distance_of: ndim=%d", ndim); return 1; }
	const void *src[2] = {xx, yy};
	const size_t bytes[2] = {(size_t) nsamples * ndim * 8, (size_t) nothers * ndim * 8};
	char *d[2], *d_counts;
	if (!stage_in(src, bytes, 2, (size_t) nothers * 4, d, &d_counts)) return 1;
	if (mdns_count_within_dev((const double *) d[0], nsamples, ndim, maxdistance, (const double *) d[1],
	                          nothers, (int *) d_counts) != 0) return 1;
	std::vector<int> hits((size_t) nothers);
	if (!MDNS_HIP(hipMemcpyAsync(hits.data(), d_counts, (size_t) nothers * 4, hipMemcpyDeviceToHost, c->stream))) return 1;
	if (!MDNS_HIP(hipStreamSynchronize(c->stream))) return 1;
	// cneighbors.c:108-115: out[j]++ per member inside, scanning stops once out[j] >= countmax
	double *out = (double *) outp;
	for (int j = 0; j < nothers; j++) {
		int h = hits[j];
		if (countmax > 0) {
			while (h > 0) { out[j] += 1; h--; if (out[j] >= countmax) break; }
		} else {
			for (; h > 0; h--) out[j] += 1;
		}
	}
	return 0;
}

extern "C" int mdns_is_within_distance_of(const void *xx, int nsamples, int ndim, double maxdistance,
                                          const void *y)
{
	double out = 0;
	if (nsamples <= 0) return 0;
	if (mdns_count_within_distance_of(xx, nsamples, ndim, maxdistance, y, 1, &out, 1) != 0) return -1;
	return out > 0 ? 1 : 0;
}

extern "C" double mdns_bootstrapped_maxdistance(const void *xx, int nsamples, int ndim,
                                                const void *choice, int nbootstraps)
{
	Context *c = ctx();
	if (!c) return NAN;
	if (nsamples <= 0 || nbootstraps <= 0 || ndim <= 0) {
		set_error("mdns_bootstrapped_maxdistance: bad sizes (%d, %d, %d)", nsamples, ndim, nbootstraps);
		return NAN;
	}
	const void *src[2] = {xx, choice};
	const size_t bytes[2] = {(size_t) nsamples * ndim * 8, (size_t) nsamples * nbootstraps * 8};
	char *d[2], *d_round;
	if (!stage_in(src, bytes, 2, (size_t) nbootstraps * 8, d, &d_round)) return NAN;
	if (mdns_bootstrap_round_maxsq_dev((const double *) d[0], nsamples, ndim, (const double *) d[1],
	                                   nbootstraps, (double *) d_round) != 0) return NAN;
	std::vector<double> sq((size_t) nbootstraps);
	if (!MDNS_HIP(hipMemcpyAsync(sq.data(), d_round, (size_t) nbootstraps * 8, hipMemcpyDeviceToHost, c->stream))) return NAN;
	if (!MDNS_HIP(hipStreamSynchronize(c->stream))) return NAN;
	// sqrt is monotone: max_i sqrt(min_j d_ij) = sqrt(max_i min_j d_ij) (cneighbors.c:160-174)
	double best = 0;
	for (int b = 0; b < nbootstraps; b++) { const double r = std::sqrt(sq[b]); if (r > best) best = r; }
	return best;
}

extern "C" double mdns_most_distant_nearest_neighbor(const void *xx, int nsamples, int ndim)
{
	Context *c = ctx();
	if (!c) return NAN;
	if (nsamples <= 0 || ndim <= 0) { set_error("mdns_most_distant_nearest_neighbor: bad sizes"); return NAN; }
	const void *src[1] = {xx};
	const size_t bytes[1] = {(size_t) nsamples * ndim * 8};
	char *d[1], *d_out;
	if (!stage_in(src, bytes, 1, 8, d, &d_out)) return NAN;
	if (!MDNS_HIP(hipMemsetAsync(d_out, 0, 8, c->stream))) return NAN;
	if (!launch_nn_maxsq((const double *) d[0], nsamples, ndim, (double *) d_out)) return NAN;
	double sq = 0;
	if (!MDNS_HIP(hipMemcpyAsync(&sq, d_out, 8, hipMemcpyDeviceToHost, c->stream))) return NAN;
	if (!MDNS_HIP(hipStreamSynchronize(c->stream))) return NAN;
	return std::sqrt(sq);
}

// ---------------------------------------------------------------------------------------
// resident RadFriends region (members + radius)
// ---------------------------------------------------------------------------------------
struct mdns_region {
	const double *d_members = nullptr;
	double *owned = nullptr;          // == d_members when this handle allocated them
	int K = 0, ndim = 0;
	double radius = NAN;              // maxdistance, host copy
	double thresh_sq = NAN;           // membership threshold on squared distances, host copy
	// A radius computation ends on the device: the last workgroup of the bootstrap kernel
	// writes {radius, threshold} to d_res (for the membership kernel, stream order) and to
	// h_res (mapped host memory), then h_res->seq = seq.  The host polls for that.
	double *d_round = nullptr;        // per-round max of squared nearest-chosen distances (slab or own_round)
	double *own_round = nullptr;
	int round_cap = 0;
	RegionResult *d_res = nullptr;
	RegionResult *h_res = nullptr;    // hipHostMalloc (mapped, coherent)
	RegionResult *h_res_dev = nullptr;   // its device address
	unsigned *d_counter = nullptr;
	int slot = -1;                    // result slot (take_result_slot)
	unsigned long long seq = 0;
	bool on_device = false;           // the membership kernel must use d_res
	bool pending = false;             // the host copies still have to be fetched (async path)
	size_t owned_bytes = 0, own_round_bytes = 0;      // pool_take sizes
	void *d_chosen = nullptr; size_t chosen_bytes = 0;
	void *d_points = nullptr; size_t points_bytes = 0;
	void *d_counts = nullptr; size_t counts_bytes = 0;
};

// Result slots ({radius, threshold} on the device, its mapped host mirror and the ticket
// counter) come from slabs allocated once: the host code builds a region handle per region,
// and hipHostMalloc per handle was a millisecond each.
static constexpr int kSlabSlots = 64;
static constexpr int kSlotRounds = 16;    // per-round maxima kept in the slab (more: own buffer)
struct ResultSlab { RegionResult *d_res; unsigned *d_counter; RegionResult *h_res, *h_res_dev; double *d_round; };
static std::vector<ResultSlab> g_slabs;
static std::vector<int> g_free_slots;

static int take_result_slot()
{
	if (g_free_slots.empty()) {
		ResultSlab slab = {nullptr, nullptr, nullptr, nullptr, nullptr};
		const bool ok =
		    MDNS_HIP(hipMalloc((void **) &slab.d_res, kSlabSlots * sizeof(RegionResult))) &&
		    MDNS_HIP(hipMalloc((void **) &slab.d_counter, kSlabSlots * sizeof(unsigned))) &&
		    MDNS_HIP(hipMemset(slab.d_counter, 0, kSlabSlots * sizeof(unsigned))) &&
		    MDNS_HIP(hipMalloc((void **) &slab.d_round, kSlabSlots * kSlotRounds * sizeof(double))) &&
		    MDNS_HIP(hipMemset(slab.d_round, 0, kSlabSlots * kSlotRounds * sizeof(double))) &&
		    MDNS_HIP(hipHostMalloc((void **) &slab.h_res, kSlabSlots * sizeof(RegionResult), hipHostMallocMapped | hipHostMallocCoherent)) &&
		    MDNS_HIP(hipHostGetDevicePointer((void **) &slab.h_res_dev, slab.h_res, 0));
		if (!ok) return -1;
		memset(slab.h_res, 0, kSlabSlots * sizeof(RegionResult));
		const int base = (int) g_slabs.size() * kSlabSlots;
		g_slabs.push_back(slab);
		for (int i = kSlabSlots - 1; i >= 0; i--) g_free_slots.push_back(base + i);
	}
	const int slot = g_free_slots.back();
	g_free_slots.pop_back();
	return slot;
}

// Device buffers of the region handles come from a small cache: the host code builds a region
// per rebuild of a constrained draw (thousands per run), and hipMalloc / hipFree per buffer were a
// quarter of a millisecond per region.  Everything runs on one stream, so a buffer handed back
// and taken again is reused in stream order: no synchronisation is needed.
struct PoolEntry { size_t bytes; void *p; };
static std::vector<PoolEntry> g_pool;

static void *pool_take(size_t bytes, size_t *got)
{
	size_t want = 4096;
	while (want < bytes) want <<= 1;
	for (size_t i = 0; i < g_pool.size(); i++)
		if (g_pool[i].bytes == want) {
			void *p = g_pool[i].p;
			g_pool[i] = g_pool.back();
			g_pool.pop_back();
			*got = want;
			return p;
		}
	void *p = nullptr;
	if (!MDNS_HIP(hipMalloc(&p, want))) return nullptr;
	*got = want;
	return p;
}

static void pool_give(void *p, size_t bytes)
{
	if (!p) return;
	if (g_pool.size() >= 64) { (void) hipFree(p); return; }
	g_pool.push_back({bytes, p});
}

// make *p hold at least `need` bytes (contents are not kept)
static bool pool_fit(void **p, size_t *cap, size_t need)
{
	if (need <= *cap) return true;
	pool_give(*p, *cap);
	*cap = 0;
	*p = pool_take(need, cap);
	return *p != nullptr;
}

static bool region_fetch(mdns_region *r);

bool mdns::poll_expired(long long *started_ns)
{
	static const double limit_s = [] { const char *v = getenv("MDNS_POLL_TIMEOUT_S"); const double t = v ? atof(v) : 120.0; return t > 0 ? t : 120.0; }();
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	const long long now = (long long) ts.tv_sec * 1000000000LL + ts.tv_nsec;
	if (*started_ns == 0) { *started_ns = now; return false; }
	return (double) (now - *started_ns) * 1e-9 > limit_s;
}

static mdns_region *region_new(const double *d_members, double *owned, int K, int ndim)
{
	mdns_region *r = new mdns_region();
	r->d_members = d_members; r->owned = owned; r->K = K; r->ndim = ndim;
	r->slot = take_result_slot();
	if (r->slot < 0) { mdns_region_destroy(r); return nullptr; }
	const ResultSlab &slab = g_slabs[r->slot / kSlabSlots];
	const int i = r->slot % kSlabSlots;
	r->d_res = slab.d_res + i;
	r->d_counter = slab.d_counter + i;
	r->h_res = slab.h_res + i;
	r->h_res_dev = slab.h_res_dev + i;
	r->d_round = slab.d_round + (size_t) i * kSlotRounds;      // zero: every finishing computation leaves it so
	r->round_cap = kSlotRounds;
	r->seq = r->h_res->seq;            // a recycled slot keeps counting from where it was
	return r;
}

extern "C" mdns_region *mdns_region_create(const double *members, int K, int ndim)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (!members || K <= 0 || ndim <= 0) { set_error("mdns_region_create: bad arguments (K=%d ndim=%d)", K, ndim); return nullptr; }
	const size_t bytes = (size_t) K * ndim * sizeof(double);
	size_t got = 0;
	double *d = (double *) pool_take(bytes, &got);
	if (!d) return nullptr;
	if (!MDNS_HIP(hipMemcpyAsync(d, members, bytes, hipMemcpyHostToDevice, c->stream)) ||
	    !MDNS_HIP(hipStreamSynchronize(c->stream))) { pool_give(d, got); return nullptr; }
	mdns_region *r = region_new(d, d, K, ndim);
	if (r) r->owned_bytes = got; else pool_give(d, got);
	return r;
}

// Members and packed bootstrap choice staged together, K6 launched, radius waited for: what
// RadFriendsRegion.__init__ (radfriendsregion.py:59-70) needs from the device, in ONE call --
// one pinned staging copy, one H2D, one launch, one poll of the mapped result.
// wait == false: the radius stays in flight (mdns_region_radius waits for it); the pinned staging
// block is busy until then -- the caller creates no other region meanwhile
static mdns_region *create_bootstrapped(const double *members, int K, int ndim, const unsigned *packed, int nbootstraps,
                                        double *radius, bool wait)
{
	Context *c = ctx();
	if (!c) return nullptr;
	if (!members || !packed || (wait && !radius) || K <= 0 || ndim <= 0 || nbootstraps <= 0 || nbootstraps > 16) {
		set_error("mdns_region_create_bootstrapped: bad arguments (K=%d ndim=%d rounds=%d)", K, ndim, nbootstraps);
		return nullptr;
	}
	const size_t mbytes = (size_t) K * ndim * sizeof(double), pbytes = (size_t) K * sizeof(unsigned);
	const size_t poff = (mbytes + 255) & ~(size_t) 255;
	size_t got = 0;
	char *d = (char *) pool_take(poff + pbytes, &got);
	if (!d) return nullptr;
	char *pin = (char *) pinned_scratch(poff + pbytes);
	if (!pin) { pool_give(d, got); return nullptr; }
	memcpy(pin, members, mbytes);
	memcpy(pin + poff, packed, pbytes);
	if (!MDNS_HIP(hipMemcpyAsync(d, pin, poff + pbytes, hipMemcpyHostToDevice, c->stream))) { pool_give(d, got); return nullptr; }
	mdns_region *r = region_new((const double *) d, (double *) d, K, ndim);
	if (!r) { (void) hipStreamSynchronize(c->stream); pool_give(d, got); return nullptr; }
	r->owned_bytes = got;
	const BootstrapFinish fin = {r->d_counter, r->d_res, r->h_res_dev, ++r->seq};
	if (!launch_bootstrap_packed(r->d_members, K, ndim, (const unsigned *) (d + poff), nbootstraps, r->d_round, &fin)) {
		(void) hipStreamSynchronize(c->stream);
		mdns_region_destroy(r);
		return nullptr;
	}
	r->on_device = true;
	r->pending = true;
	if (!wait) return r;
	*radius = mdns_region_radius(r);          // waits: the pinned staging block is free again
	if (*radius != *radius) { mdns_region_destroy(r); return nullptr; }
	return r;
}

extern "C" mdns_region *mdns_region_create_bootstrapped(const double *members, int K, int ndim,
                                                        const unsigned *packed, int nbootstraps, double *radius)
{
	return create_bootstrapped(members, K, ndim, packed, nbootstraps, radius, true);
}

mdns_region *mdns::region_begin_bootstrapped(const double *members, int K, int ndim, const unsigned *packed, int nbootstraps)
{
	return create_bootstrapped(members, K, ndim, packed, nbootstraps, nullptr, false);
}

// what the chain kernels (mdns_chain.hip) need of a region whose radius is being computed or known on
// the device
bool mdns::region_view(mdns_region *r, RegionView *out)
{
	if (!r || !out || !r->on_device) return false;
	out->d_members = r->d_members; out->K = r->K; out->ndim = r->ndim; out->d_res = r->d_res;
	return true;
}

extern "C" mdns_region *mdns_region_wrap_dev(const double *d_members, int K, int ndim)
{
	if (!ctx()) return nullptr;
	if (!d_members || K <= 0 || ndim <= 0) { set_error("mdns_region_wrap_dev: bad arguments"); return nullptr; }
	return region_new(d_members, nullptr, K, ndim);
}

extern "C" void mdns_region_destroy(mdns_region *r)
{
	if (!r) return;
	// a radius computation still in flight writes the result slot: let it land before the slot
	// can go to another handle (the buffers themselves are reused in stream order)
	if (r->pending && ctx()) (void) region_fetch(r);
	pool_give(r->owned, r->owned_bytes);
	pool_give(r->own_round, r->own_round_bytes);
	pool_give(r->d_chosen, r->chosen_bytes);
	pool_give(r->d_points, r->points_bytes);
	pool_give(r->d_counts, r->counts_bytes);
	if (r->slot >= 0) g_free_slots.push_back(r->slot);
	delete r;
}

extern "C" int mdns_region_set_radius(mdns_region *r, double maxdistance)
{
	if (!r) { set_error("null region"); return 1; }
	// a radius computation still in flight writes this handle's result slot: let it land first
	if (r->pending && ctx()) (void) region_fetch(r);
	r->radius = maxdistance;
	r->thresh_sq = sqrt_threshold(maxdistance);
	r->on_device = false;
	r->pending = false;
	return 0;
}

// host copies of {radius, threshold}; waits for the device when they are still in flight
static bool region_fetch(mdns_region *r)
{
	if (!r->pending) return true;
	Context *c = ctx();
	volatile unsigned long long *seq = &r->h_res->seq;
	// The result usually is there already (the caller has launched other work meanwhile).
	// While polling, look at the stream now and then: once it has drained, everything the
	// kernel wrote is visible, and a failed launch shows up as an error instead of a hang.
	long long started = 0;
	for (unsigned spin = 0; *seq != r->seq; spin++) {
		if ((spin & 1023) != 1023) continue;
		const hipError_t e = hipStreamQuery(c->stream);
		if (e == hipErrorNotReady) {
			if (poll_expired(&started)) { set_error("radius computation: no result within MDNS_POLL_TIMEOUT_S"); return false; }
			continue;
		}
		if (e != hipSuccess) { set_error("radius computation failed: %s", hipGetErrorString(e)); return false; }
		if (*seq != r->seq) { set_error("radius computation finished without a result"); return false; }
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	r->radius = r->h_res->radius;
	r->thresh_sq = sqrt_threshold(r->radius);     // the device derived the same number (tested)
	r->pending = false;
	return true;
}

extern "C" double mdns_region_radius(mdns_region *r)
{
	if (!r || !ctx() || !region_fetch(r)) return NAN;
	return r->radius;
}

extern "C" int mdns_region_bootstrap_radius_async(mdns_region *r, const double *d_chosen, int nbootstraps)
{
	Context *c = ctx();
	if (!c || !r) return 1;
	if (nbootstraps <= 0) { set_error("mdns_region_bootstrap_radius: nbootstraps=%d", nbootstraps); return 1; }
	if (!region_fetch(r)) return 1;             // a previous result may still be travelling
	if (r->round_cap < nbootstraps) {
		if (!pool_fit((void **) &r->own_round, &r->own_round_bytes, (size_t) nbootstraps * sizeof(double))) return 1;
		r->d_round = r->own_round;
		// zero once: every finishing computation hands its slots back zeroed
		if (!MDNS_HIP(hipMemsetAsync(r->d_round, 0, (size_t) nbootstraps * sizeof(double), c->stream))) return 1;
		r->round_cap = nbootstraps;
	}
	const BootstrapFinish fin = {r->d_counter, r->d_res, r->h_res_dev, ++r->seq};
	if (!launch_bootstrap(r->d_members, r->K, r->ndim, d_chosen, nbootstraps, r->d_round, &fin)) return 1;
	r->on_device = true;
	r->pending = true;
	return 0;
}

extern "C" double mdns_region_bootstrap_radius_dev(mdns_region *r, const double *d_chosen, int nbootstraps)
{
	if (mdns_region_bootstrap_radius_async(r, d_chosen, nbootstraps) != 0) return NAN;
	return mdns_region_radius(r);
}

extern "C" double mdns_region_bootstrap_radius(mdns_region *r, const double *chosen, int nbootstraps)
{
	Context *c = ctx();
	if (!c || !r) return NAN;
	const size_t n = (size_t) r->K * (nbootstraps > 0 ? nbootstraps : 0);
	if (n == 0) { set_error("mdns_region_bootstrap_radius: nbootstraps=%d", nbootstraps); return NAN; }
	if (!pool_fit(&r->d_chosen, &r->chosen_bytes, n * sizeof(double))) return NAN;
	if (!MDNS_HIP(hipMemcpyAsync(r->d_chosen, chosen, n * sizeof(double), hipMemcpyHostToDevice, c->stream))) return NAN;
	return mdns_region_bootstrap_radius_dev(r, (const double *) r->d_chosen, nbootstraps);
}

extern "C" double mdns_region_bootstrap_radius_packed(mdns_region *r, const unsigned *packed, int nbootstraps)
{
	Context *c = ctx();
	if (!c || !r) return NAN;
	if (nbootstraps <= 0 || nbootstraps > 16 || !packed) { set_error("mdns_region_bootstrap_radius_packed: nbootstraps=%d (1..16)", nbootstraps); return NAN; }
	if (!region_fetch(r)) return NAN;
	const size_t bytes = (size_t) r->K * sizeof(unsigned);
	if (!pool_fit(&r->d_chosen, &r->chosen_bytes, bytes)) return NAN;
	if (!MDNS_HIP(hipMemcpyAsync(r->d_chosen, packed, bytes, hipMemcpyHostToDevice, c->stream))) return NAN;
	const BootstrapFinish fin = {r->d_counter, r->d_res, r->h_res_dev, ++r->seq};
	if (!launch_bootstrap_packed(r->d_members, r->K, r->ndim, (const unsigned *) r->d_chosen, nbootstraps, r->d_round, &fin)) return NAN;
	r->on_device = true;
	r->pending = true;
	return mdns_region_radius(r);
}

extern "C" int mdns_region_count_dev(mdns_region *r, const double *d_points, int M, int *d_counts)
{
	if (!ctx() || !r) return 1;
	if (M < 0) { set_error("mdns_region_count: M=%d", M); return 1; }
	if (M == 0) return 0;
	if (!r->on_device && r->radius != r->radius) { set_error("mdns_region_count: the region has no radius yet"); return 1; }
	// right after a radius computation the kernel takes the threshold the bootstrap kernel
	// left in device memory (stream order)
	return launch_count_within(r->d_members, r->K, r->ndim, r->thresh_sq, r->on_device ? r->d_res : nullptr,
	                           d_points, M, d_counts) ? 0 : 1;
}

// one staging block for the membership round trips of the native constrainer: pinned + mapped,
// layout { seq | counts int32[cap] | points f64[cap * ndim] }
static char *g_stage = nullptr, *g_stage_dev = nullptr;
static size_t g_stage_points = 0, g_stage_doubles = 0;
static unsigned long long g_stage_seq = 0;
static int *g_stage_ticket = nullptr;                 // device: workgroups of the current launch that are through

// K3 for host points with the least round trip: ONE kernel.  The points sit in a pinned block mapped
// into the device, the kernel reads them there, stores the counts there and its last workgroup
// raises a sequence number the host polls for.  (History, per call in a real run: pageable copies
// both ways and a stream synchronisation 42 us; pinned copy in + count + export kernel + polling
// 35 us, of which four device commands -- copy, clear, count, export -- for 6 us of counting.
// Tried: member chunks adding partial counts in a device buffer that the last workgroup exports and
// clears -- more workgroups, but 2.17 s against 1.65-1.73 s over the 73 600 calls of a C2 run.)
extern "C" int mdns_region_count_polled(mdns_region *r, const double *points, int M, int *counts)
{
	Context *c = ctx();
	if (!c || !r) return 1;
	if (M <= 0) return M < 0;
	if (!r->on_device && r->radius != r->radius) { set_error("mdns_region_count: the region has no radius yet"); return 1; }
	const size_t n = (size_t) M * r->ndim;
	if ((size_t) M > g_stage_points || n > g_stage_doubles) {
		if (g_stage) { (void) hipStreamSynchronize(c->stream); (void) hipHostFree(g_stage); g_stage = nullptr; }
		const size_t np = (size_t) M + M / 2 + 1024, nd = n + n / 2 + 4096;
		if (!MDNS_HIP(hipHostMalloc((void **) &g_stage, 64 + np * sizeof(int) + 64 + nd * sizeof(double), hipHostMallocMapped)) ||
		    !MDNS_HIP(hipHostGetDevicePointer((void **) &g_stage_dev, g_stage, 0))) { g_stage = nullptr; g_stage_points = g_stage_doubles = 0; return 1; }
		g_stage_points = np; g_stage_doubles = nd;
		*(volatile unsigned long long *) g_stage = g_stage_seq;
	}
	if (!g_stage_ticket) {
		if (!MDNS_HIP(hipMalloc((void **) &g_stage_ticket, sizeof(int))) ||
		    !MDNS_HIP(hipMemsetAsync(g_stage_ticket, 0, sizeof(int), c->stream))) { g_stage_ticket = nullptr; return 1; }
	}
	const size_t off_counts = 64, off_points = (64 + g_stage_points * sizeof(int) + 63) & ~(size_t) 63;
	memcpy(g_stage + off_points, points, n * sizeof(double));
	const unsigned long long seq = ++g_stage_seq;
	const CountMail mail = {g_stage_ticket, (unsigned long long *) g_stage_dev, seq};
	// right after a radius computation the kernel takes the threshold the bootstrap kernel left in
	// device memory (stream order)
	if (!launch_count_within(r->d_members, r->K, r->ndim, r->thresh_sq, r->on_device ? r->d_res : nullptr,
	                         (const double *) (g_stage_dev + off_points), M, (int *) (g_stage_dev + off_counts), &mail)) return 1;
	volatile unsigned long long *at = (volatile unsigned long long *) g_stage;
	long long started = 0;
	for (unsigned spin = 0; *at != seq; spin++) {
		if ((spin & 1023) != 1023) continue;
		const hipError_t e = hipStreamQuery(c->stream);
		if (e == hipErrorNotReady) {
			if (poll_expired(&started)) { set_error("membership count: no result within MDNS_POLL_TIMEOUT_S"); return 1; }
			continue;
		}
		if (e != hipSuccess) { set_error("membership count failed: %s", hipGetErrorString(e)); return 1; }
		if (*at != seq) {
			// (the stream drained and the number is not there: the launch failed; the ticket counter
			// may be anywhere)
			(void) hipMemsetAsync(g_stage_ticket, 0, sizeof(int), c->stream);
			set_error("membership count finished without a result");
			return 1;
		}
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	memcpy(counts, g_stage + off_counts, (size_t) M * sizeof(int));
	return 0;
}

extern "C" int mdns_region_count(mdns_region *r, const double *points, int M, int *counts)
{
	Context *c = ctx();
	if (!c || !r) return 1;
	if (M <= 0) return M < 0;
	const size_t n = (size_t) M * r->ndim;
	if (!pool_fit(&r->d_points, &r->points_bytes, n * sizeof(double)) ||
	    !pool_fit(&r->d_counts, &r->counts_bytes, (size_t) M * sizeof(int))) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(r->d_points, points, n * sizeof(double), hipMemcpyHostToDevice, c->stream))) return 1;
	if (mdns_region_count_dev(r, (const double *) r->d_points, M, (int *) r->d_counts) != 0) return 1;
	if (!MDNS_HIP(hipMemcpyAsync(counts, r->d_counts, (size_t) M * sizeof(int), hipMemcpyDeviceToHost, c->stream))) return 1;
	return MDNS_HIP(hipStreamSynchronize(c->stream)) ? 0 : 1;
}
