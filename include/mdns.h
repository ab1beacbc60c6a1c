/*
 * mdns.h -- C ABI of libmdns_hip.so, the MI355X (gfx950) implementation of the
 * massivedatans hot path: batched spectral-line log-likelihoods and the RadFriends
 * neighbourhood / safe-radius tests.
 *
 * Plain C: pointers and sizes only, no C++ or torch types.  Every entry point names the
 * reference interface it stands in for (paths under the reference checkout).
 *
 * The library has NO CPU fallback: every compute entry point runs HIP kernels on the
 * selected device and reports failure (return code / NaN, message via mdns_last_error())
 * when no device is usable.
 *
 * Threading: like the reference (one Python thread; its OpenMP lives inside the .so files),
 * the entry points are meant to be called from one thread at a time.  One process drives one
 * GPU; all work is issued on one HIP stream (the library's own, or the caller's through
 * mdns_set_stream).
 *
 * Part 1  drop-in entry points with the reference's argument lists (host pointers);
 *         the three shim libraries clike.so / cmuselike.so / cneighbors.so re-export them
 *         under the reference's symbol names.
 * Part 2  device-resident handles (spectra stay in HBM, candidates are scored in batches).
 * Part 3  raw device entry points, memory, stream and event helpers (bench / multi-GPU).
 */
#ifndef MDNS_H
#define MDNS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------ */
/* Library state                                                                        */
/* ------------------------------------------------------------------------------------ */

/* Select the device of this process (one process per GPU).  device < 0: MDNS_DEVICE, else
 * LOCAL_RANK modulo the device count, else 0.  Called implicitly by the first compute call.
 * Returns 0 on success. */
int mdns_init(int device);
/* Number of visible HIP devices (0 when there is none). */
int mdns_device_count(void);
/* Message of the last failure in this thread's process ("" if none). */
const char *mdns_last_error(void);
/* ABI version, bumped on any signature change. */
int mdns_abi_version(void);

/* ------------------------------------------------------------------------------------ */
/* Part 1: drop-in entry points (reference argument lists, host pointers)               */
/* ------------------------------------------------------------------------------------ */

/*
 * K1 -- stands in for `like` of clike.so: clike.c:34-40, bound at sample.py:85-96, called at
 * sample.py:106.  x f64[nx]; yy f64[nx,ndata] C-order (element (j,i) at i + j*ndata);
 * data_mask C bool[ndata]; Lout f64[mask.sum()], compacted in mask order and ACCUMULATED
 * (+=, clike.c:72) -- the caller pre-zeroes it (sample.py:104).  Returns 0 (the reference
 * always does); non-zero only on a device failure.
 * The spectra are uploaded on every call unless the yy pointer was registered with
 * mdns_register_spectra() (then the HBM-resident copy is used).
 */
int mdns_gauss_like(const void *x, const void *yy, int ndata, int nx,
                    double A, double mu, double sig, double noise_level,
                    const void *data_mask, void *Lout);

/*
 * K2 -- stands in for `like` of cmuselike.so: cmuselike.c:34-38, bound at
 * musefuse.py:509-517, called at musefuse.py:534.  yy, vv f64[nx,ndata]; ypred f64[nx];
 * Lout f64[ndata] NOT compacted, only masked entries are written (cmuselike.c:49,62).
 */
int mdns_muse_like(const void *yy, const void *vv, const void *ypred, const void *data_mask,
                   int ndata, int nx, void *Lout);

/* Keep a device copy of a host spectra array for the drop-in calls above: later calls whose
 * yy (and vv) pointer, ndata and nx match use it instead of uploading.  The caller promises
 * not to modify the array while it is registered.  vv may be NULL (K1). */
int mdns_register_spectra(const void *yy, const void *vv, int ndata, int nx);
int mdns_unregister_spectra(const void *yy);

/* K5 -- cneighbors.c:32-34 (clustering/neighbors.py:100-110). */
double mdns_most_distant_nearest_neighbor(const void *xx, int nsamples, int ndim);
/* K4 -- cneighbors.c:77-79 (clustering/neighbors.py:112-124); 1 = inside. */
int mdns_is_within_distance_of(const void *xx, int nsamples, int ndim, double maxdistance,
                               const void *y);
/* K3 -- cneighbors.c:95-98 (clustering/neighbors.py:126-159).  out f64[nothers] is
 * incremented in place, with the reference's countmax early-stop semantics. */
int mdns_count_within_distance_of(const void *xx, int nsamples, int ndim, double maxdistance,
                                  const void *yy, int nothers, void *out, const int countmax);
/* K6 -- cneighbors.c:125-130 (clustering/neighbors.py:161-177).  choice f64[nsamples,
 * nbootstraps], tested != 0. */
double mdns_bootstrapped_maxdistance(const void *xx, int nsamples, int ndim,
                                     const void *choice, int nbootstraps);

/* ------------------------------------------------------------------------------------ */
/* Part 2: device-resident spectra + batched scoring (extension)        */
/* ------------------------------------------------------------------------------------ */

typedef struct mdns_spectra mdns_spectra;

#define MDNS_LAYOUT_CHANNEL_MAJOR 0   /* reference layout [nx, ndata] (sample.py:31)        */
#define MDNS_LAYOUT_DATASET_MAJOR 1   /* [ndata, nx]: one spectrum per contiguous row        */

/* Upload spectra once; they stay in HBM as [ndata, nx] rows.  x f64[nx] (may be NULL when
 * only precomputed templates are scored); v per-pixel variances (NULL for K1-only use). */
mdns_spectra *mdns_spectra_create(const double *x, const double *y, const double *v,
                                  int ndata, int nx, int layout);
void mdns_spectra_destroy(mdns_spectra *s);
int mdns_spectra_ndata(const mdns_spectra *s);
int mdns_spectra_nx(const mdns_spectra *s);

/*
 * Score B candidate lines against M selected spectra in one pass over the spectra.
 * params f64[B,3] = (A, mu, sig) per candidate (sig already linear, sample.py:103);
 * row_ids int32[M] = indices of the selected spectra in ascending mask order (NULL = all,
 * M = ndata); Lout f64[B,M] receives the log-likelihoods  -0.5 * sum_j ((m_j - y_ij)/noise)^2
 * (what sample.py:108 returns), compacted like the reference's output.
 */
int mdns_gauss_loglike_batch(mdns_spectra *s, const double *params, int B, double noise_level,
                             const int *row_ids, int M, double *Lout);

/* Same for the scale-marginalised likelihood (cmuselike.c:45-64) against B precomputed
 * templates ypred f64[B,nx]; Lout f64[B,M] = -0.5*chi. */
int mdns_muse_loglike_batch(mdns_spectra *s, const double *ypred, int B,
                            const int *row_ids, int M, double *Lout);

/* Config-C5 template (three Gaussians on a flat continuum, massivedatans_amd/gen.py
 * muse_template) evaluated ON DEVICE for B parameter vectors params f64[B,5], then scored
 * exactly as mdns_muse_loglike_batch. */
int mdns_muse3_loglike_batch(mdns_spectra *s, const double *params, int B,
                             const int *row_ids, int M, double *Lout);

/* Live-point pool resident on the device: the members of a RadFriends region
 * (clustering/radfriendsregion.py:59-70 keeps `members` and `maxdistance`; every are_inside /
 * count_nearby_members call of a region's life re-uses them, radfriendsregion.py:82-98). */
typedef struct mdns_region mdns_region;

/* members f64[K, ndim] on the host (copied to the device) ... */
mdns_region *mdns_region_create(const double *members, int K, int ndim);
/* ... together with the packed bootstrap choice of its safe radius (bit b of packed[i]: point i
 * is chosen in round b, as mdns_region_bootstrap_radius_packed): one upload, one launch; *radius
 * receives the result (radfriendsregion.py:59-64 in one call). */
mdns_region *mdns_region_create_bootstrapped(const double *members, int K, int ndim,
                                             const unsigned *packed, int nbootstraps, double *radius);
/* ... or already on the device (borrowed, not copied: e.g. an all-gathered pool). */
mdns_region *mdns_region_wrap_dev(const double *d_members, int K, int ndim);
void mdns_region_destroy(mdns_region *r);
/* K6 on the resident members (cneighbors.c:125-179): chosen f64[K, nbootstraps] on the host /
 * on the device.  Returns the radius and keeps it as the region's maxdistance; NaN on failure. */
double mdns_region_bootstrap_radius(mdns_region *r, const double *chosen, int nbootstraps);
double mdns_region_bootstrap_radius_dev(mdns_region *r, const double *d_chosen, int nbootstraps);
/* The same with the choice matrix packed by the caller: bit b of packed[i] (uint32[K], host) is
 * set when point i is chosen in round b, i.e. chosen[i][b] != 0 (cneighbors.c:146); at most 16
 * rounds.  A twentieth of the bytes to build and to upload. */
double mdns_region_bootstrap_radius_packed(mdns_region *r, const unsigned *packed, int nbootstraps);
/* The same without waiting: radius and membership threshold are finished ON the device, so a
 * following mdns_region_count_dev needs no host round trip; mdns_region_radius() then waits
 * for and returns the value (the host needs it for the bounding box, radfriendsregion.py:69).
 * Returns 0 when enqueued. */
int mdns_region_bootstrap_radius_async(mdns_region *r, const double *d_chosen, int nbootstraps);
/* RadFriendsRegion(members, maxdistance=...) with a given radius (hiermetriclearn.py:54,90). */
int mdns_region_set_radius(mdns_region *r, double maxdistance);
double mdns_region_radius(mdns_region *r);
/* K3 with the region's radius (cneighbors.c:95-119, no early stop): points f64[M, ndim],
 * counts int32[M] overwritten.  Host pointers (synchronous) / device pointers (asynchronous). */
int mdns_region_count(mdns_region *r, const double *points, int M, int *counts);
int mdns_region_count_dev(mdns_region *r, const double *d_points, int M, int *d_counts);
/* mdns_region_count with the shortest round trip (what a native constrainer calls once per 1000
 * proposals): ONE kernel, which reads the points from and stores the counts to a pinned block
 * mapped into the device and raises a sequence number the host polls for -- no copies, no stream
 * synchronisation. */
int mdns_region_count_polled(mdns_region *r, const double *points, int M, int *counts);

/* ------------------------------------------------------------------------------------ */
/* Part 2b: the constrained draw decided on the device (extension) */
/* ------------------------------------------------------------------------------------ */

/*
 * The floating-point state of the joint sampler, resident in HBM: what the reference keeps in
 * `live_pointsL[nlive, ndata]` (multi_nested_sampler.py:111) and in the likelihoods of its
 * shelves (`self.shelves`, :117), plus what it derives from them on every draw -- the
 * thresholds `Lmins_higher` (:438-447: for a data set with n accepted points waiting, the
 * (n+1)-th smallest of its live and shelf likelihoods).  Everything is indexed by the ORIGINAL
 * data-set index of the spectra handle, also after data sets have finished (`cut_down`, :148-173).
 * The integer side (point ids, shelves' ids, data-set graph) stays with the host.
 */
typedef struct mdns_joint mdns_joint;

/* shelf_cap: accepted points a data set can hold waiting (grown on demand, see mdns_joint_reserve). */
mdns_joint *mdns_joint_create(mdns_spectra *s, int nlive, int shelf_cap);
void mdns_joint_destroy(mdns_joint *j);

/* Initial live points (multi_nested_sampler.py:88-103): scores params f64[nlive,3] = (A, mu, sig)
 * against ALL spectra and keeps the result as the live likelihood matrix; nothing is returned
 * (mdns_joint_get_live reads it back).  All data sets are running, all shelves empty. */
int mdns_joint_init_gauss(mdns_joint *j, const double *params, double noise_level);
/* The same for spectra with variances: params f64[nlive, 5] of the three-line template
 * (mdns_muse3_loglike_batch) scored with the scale-marginalised likelihood (cmuselike.c:45-64);
 * jitter f64[nlive, ndata] or NULL is added (musefuse.py:535: the tie-breaking noise the reference
 * adds to every likelihood evaluation).  A joint state over such spectra is driven through the
 * mdns_backend_* entry points (Part 5). */
int mdns_joint_init_muse3(mdns_joint *j, const double *params, const double *jitter);
/* The same from a host matrix liveL f64[nlive, ndata] (row p = live slot p). */
int mdns_joint_set_live(mdns_joint *j, const double *liveL);
/* liveL f64[nlive, ndata] <- the device matrix (the integrator's remainder, :536-563). */
int mdns_joint_get_live(mdns_joint *j, double *liveL);

/* The data sets still running (multi_nested_sampler.py:148-173): rows int32[nrun], ascending
 * original indices.  prepare/advance work on exactly these, in this order. */
int mdns_joint_set_running(mdns_joint *j, const int *rows, int nrun);

/* Start of an iteration (multi_nested_sampler.py:130-143 `prepare`): per running data set the
 * lowest live likelihood Lmin f64[nrun] and its live slot argmin int32[nrun] (first occurrence,
 * as numpy.argmin); shelf entries that no longer beat Lmin are dropped in order (:137-138) --
 * keep uint64[nrun * ceil(cap/64)] gets, per data set, bit k of word k/64 set when its entry k
 * stays (words per data set: mdns_joint_keep_words); thresholds are refreshed. */
int mdns_joint_prepare(mdns_joint *j, double *Lmin, int *argmin, unsigned long long *keep);
int mdns_joint_keep_words(const mdns_joint *j);
/* End of an iteration (multi_nested_sampler.py:494-534): every running data set gives up the
 * live point found by the last prepare and takes the head of its shelf.  Fails when a running
 * data set has an empty shelf. */
int mdns_joint_advance(mdns_joint *j);
/* Make room for shelves of at least `shelf_cap` entries (contents kept). */
int mdns_joint_reserve(mdns_joint *j, int shelf_cap);
int mdns_joint_shelf_cap(const mdns_joint *j);

/*
 * One chunk of a constrained draw (hiermetriclearn.py:181-196 + multi_nested_sampler.py:462-485):
 * the B candidates params f64[B,3] (already proposed by the host, in order) are scored against
 * the M selected spectra row_ids int32[M] (ascending original indices; NULL = all, M = ndata);
 * candidate b is acceptable when it beats the threshold of at least one selected data set
 * (`any(L > Lmins)`, hiermetriclearn.py:193).  *accepted = index of the FIRST acceptable
 * candidate, or -1.  For that candidate only: Lrow f64[M] its likelihoods, fillbits
 * uint64[ceil(M/64)] with bit k set when it beats the threshold of the k-th selected data set
 * (multi_nested_sampler.py:482-485); those data sets' shelves receive it and their thresholds
 * move up, all on the device.  Likelihoods of the other candidates never reach memory.
 * Lrow may be NULL (the row then stays on the device).  B <= 1024.
 */
int mdns_joint_draw_gauss(mdns_joint *j, const double *params, int B, double noise_level,
                          const int *row_ids, int M, int *accepted, double *Lrow,
                          unsigned long long *fillbits);
#define MDNS_JOINT_MAX_BATCH 1024
/* The two halves of that call for hosts that put something between them: `score` stages the
 * candidates and leaves one accept flag per candidate on the device (mdns_joint_flags_dev); with
 * the data sets sharded over several GPUs the ranks MAX-reduce those flags; `commit` then takes
 * the first flagged candidate and returns its index, likelihood row and fill bits for the
 * selection given to `score`. */
int mdns_joint_score(mdns_joint *j, const double *params, int B, double noise_level,
                     const int *row_ids, int M);
int mdns_joint_commit(mdns_joint *j, int *accepted, double *Lrow, unsigned long long *fillbits);

/* Current thresholds of all data sets, higher f64[ndata] (tests; NaN before the first prepare),
 * and the shelf sizes n int32[ndata]. */
int mdns_joint_get_thresholds(mdns_joint *j, double *higher, int *shelf_n);

/* ------------------------------------------------------------------------------------ */
/* Part 3: raw device entry points (all pointers are DEVICE pointers; asynchronous on the */
/* library stream; no host synchronisation)                                             */
/* ------------------------------------------------------------------------------------ */

void *mdns_dev_alloc(size_t bytes);
void mdns_dev_free(void *p);
int mdns_h2d(void *dst, const void *src, size_t bytes);   /* synchronous */
int mdns_d2h(void *dst, const void *src, size_t bytes);   /* synchronous */
int mdns_d2d(void *dst, const void *src, size_t bytes);   /* asynchronous, library stream */
int mdns_sync(void);                                      /* wait for the library stream */
/* Run on a caller-owned hipStream_t (e.g. the stream of an RCCL collective); NULL restores
 * the library's own stream. */
int mdns_set_stream(void *hip_stream);

/* HIP events on the library stream (bench.py times launches with these). */
void *mdns_event_create(void);
void mdns_event_destroy(void *ev);
int mdns_event_record(void *ev);
double mdns_event_elapsed_ms(void *ev_start, void *ev_stop);   /* synchronises on ev_stop */

/* Per-launch timing of the dominant kernels with HIP events on the library stream.
 * Kernel classes: 0 = gauss log-likelihood, 1 = muse log-likelihood, 2 = count-within,
 * 3 = bootstrap nearest-chosen.  mdns_profile(classes) takes a bit mask (bit k = class k; 15 =
 * all): it clears the statistics of the named classes and records from now on exactly those;
 * mdns_profile(0) stops.  Every timed launch costs two event records on the stream (about
 * 9 us per 90 us step of bench.py when every launch is timed), so time only what is being
 * measured; mdns_profile_every(n) times every n-th launch of a class only (default 1).
 * mdns_profile_read synchronises and returns, for kernel class `which`, the number of launches
 * timed and the sum of their durations in milliseconds.  mdns_profile_kernel names the
 * kernel instantiation last launched for that class (as rocprofv3 prints it, without the
 * namespace), e.g. "k_gauss_cols<8, 1>". */
int mdns_profile(int classes);
int mdns_profile_every(int n);
int mdns_profile_read(int which, long long *launches, double *total_ms);
const char *mdns_profile_kernel(int which);

/* d_params f64[B,3], d_row_ids int32[M] or NULL, d_Lout f64[B,M]. */
int mdns_gauss_loglike_batch_dev(mdns_spectra *s, const double *d_params, int B,
                                 double noise_level, const int *d_row_ids, int M,
                                 double *d_Lout);
int mdns_muse_loglike_batch_dev(mdns_spectra *s, const double *d_ypred, int B,
                                const int *d_row_ids, int M, double *d_Lout);
int mdns_muse3_loglike_batch_dev(mdns_spectra *s, const double *d_params, int B,
                                 const int *d_row_ids, int M, double *d_Lout);

/* The two halves of mdns_joint_draw_gauss on device pointers, nothing waits for the host:
 * score leaves one accept flag per candidate in the handle (int32[MDNS_JOINT_MAX_BATCH], 1 =
 * beats some selected data set of THIS handle's spectra; mdns_joint_flags_dev -- with the data
 * sets sharded over ranks a MAX all-reduce of that buffer in place makes every rank see every
 * rank's flags), commit takes the first flagged candidate and leaves, in the handle's result
 * buffer (mdns_joint_result_dev; device memory, valid until the next score),
 * {int32 accepted, int32 status (0 ok, 1 shelf overflow), 8 bytes unused} | fillbits
 * uint64[ceil(M/64)] | Lrow f64[M]  (mdns_joint_result_bytes(M) bytes in all). */
int mdns_joint_score_dev(mdns_joint *j, const double *d_params, int B, double noise_level,
                         const int *d_row_ids, int M);
int *mdns_joint_flags_dev(mdns_joint *j);
int mdns_joint_commit_dev(mdns_joint *j, const int *d_row_ids, int M);
/* The same without the likelihood row (Lrow in the result buffer is left as it was): who beats its
 * threshold and with which likelihood is taken from what the score pass kept of the candidates it
 * flagged, nothing is computed again.  This is what mdns_joint_draw_gauss / mdns_joint_commit do
 * when called with Lrow == NULL. */
int mdns_joint_commit_bits_dev(mdns_joint *j, const int *d_row_ids, int M);
const void *mdns_joint_result_dev(mdns_joint *j);
size_t mdns_joint_result_bytes(int M);
/* The outcome of the last mdns_joint_commit_dev without a copy: a kernel behind the commit pass
 * leaves {accepted, status, fill words} in host memory mapped into the device, `seq` last; this call
 * polls for it (no stream synchronisation) and hands over *accepted and, when a candidate was
 * accepted, fillbits uint64[ceil(M/64)] (may be NULL).  M as passed to the commit. */
int mdns_joint_fetch(mdns_joint *j, int M, int *accepted, unsigned long long *fillbits);
/* prepare / advance without the copies to the host (results stay in the handle);
 * mdns_joint_restore_live_dev overwrites the live matrix from d_liveL f64[nlive, ndata] and
 * empties the shelves (bench.py re-runs the same iteration). */
int mdns_joint_prepare_dev(mdns_joint *j);
int mdns_joint_advance_dev(mdns_joint *j);
int mdns_joint_restore_live_dev(mdns_joint *j, const double *d_liveL);
/* Takes back the last mdns_joint_advance[_dev]: every running data set's replaced live slot gets
 * the likelihood back that the preceding prepare found there; shelves emptied (bench.py). */
int mdns_joint_undo_advance_dev(mdns_joint *j);
const double *mdns_joint_live_dev(mdns_joint *j);

/* K3 on device: d_members f64[K,ndim], d_cands f64[M,ndim]; d_counts int32[M] is
 * OVERWRITTEN with the number of members strictly within maxdistance (no early stop). */
int mdns_count_within_dev(const double *d_members, int K, int ndim, double maxdistance,
                          const double *d_cands, int M, int *d_counts);
/* K6 on device: d_chosen f64[K,nbootstraps]; d_round_sq f64[nbootstraps] receives, per round,
 * max over left-out points i>=1 of the squared distance to the nearest chosen point
 * (radius = sqrt of the max over rounds). */
int mdns_bootstrap_round_maxsq_dev(const double *d_members, int K, int ndim,
                                   const double *d_chosen, int nbootstraps,
                                   double *d_round_sq);

/* ------------------------------------------------------------------------------------------
 * Part 4 -- grouping of the data sets that share live points.
 *
 * Replaces, behind `MultiNestedSampler.generate_subsets_graph` (multi_nested_sampler.py:268-355),
 * igraph's `Graph.clusters()` on the bipartite graph {data sets} -- {live points} (:319-325)
 * and the `numpy.unique(live_pointsp[:, selected])` in front of it (:279): connected components
 * by minimum-label propagation on the device, the distinct ids as a bit map.  Integer results,
 * independent of the order the hardware works in.
 *
 * The handle keeps the id matrix `live_pointsp` (int32[nlive][ndata], C order, :108) on the
 * device, indexed by the ORIGINAL data-set index for the whole run.
 * ------------------------------------------------------------------------------------------ */
typedef struct mdns_groups mdns_groups;
mdns_groups *mdns_groups_create(int nlive, int ndata);
void mdns_groups_destroy(mdns_groups *g);
/* the whole matrix, ids int32[nlive][ndata] */
int mdns_groups_set_ids(mdns_groups *g, const int32_t *ids);
int mdns_groups_get_ids(mdns_groups *g, int32_t *ids);
/* End of an iteration (multi_nested_sampler.py:510-520): data set rows[i] gives up the live point
 * in slot slots[i] and takes new_ids[i]; n entries, rows distinct. */
int mdns_groups_replace(mdns_groups *g, const int32_t *rows, const int32_t *slots,
                        const int32_t *new_ids, int n);
/* Connected components over the data sets rows[0..M) (ascending original indices; NULL with
 * M == ndata: all).  Every id must lie in [0, npoints).  *ncomponents receives their number,
 * *ndistinct the number of distinct ids the selection holds and `distinct` (int32[cap], may be
 * NULL) those ids in ascending order -- numpy.unique of the selected columns (:279).  `touched`
 * (uint64[ceil(npoints/64)], may be NULL) receives the same set as a bit map: bit q = some
 * selected data set holds live point q. */
int mdns_groups_components(mdns_groups *g, const int32_t *rows, int M, long long npoints,
                           int *ncomponents, long long *ndistinct, int32_t *distinct, long long cap,
                           unsigned long long *touched);
/* Of the last mdns_groups_components: labels int32[M] = the lowest data-set index of the
 * component rows[i] lies in (components in ascending label order are igraph's cluster order,
 * which numbers them by their first vertex); point_labels int32[npoints] = the label of the
 * component holding live point q, -1 when no selected data set holds it.  Either may be NULL. */
int mdns_groups_labels(mdns_groups *g, int32_t *labels, int32_t *point_labels);
/* The same for the ids the last mdns_groups_components listed only: id_labels int32[ndistinct], in the
 * order of that list (a full mdns_groups_labels copies one label per id of the PILE, megabytes late in
 * a run, of which the caller looks at the listed ones). */
int mdns_groups_id_labels(mdns_groups *g, int32_t *labels, int32_t *id_labels, long long ndistinct);
/* rounds of label propagation per mdns_groups_components call so far, on average */
double mdns_groups_mean_rounds(const mdns_groups *g);

/* ------------------------------------------------------------------------------------------
 * Part 5 -- one native call per constrained draw (libmdns_host.so, csrc/host_constrainer.cpp).
 *
 * The reference's MLFriends constrainer (hiermetriclearn.py:27-211 over
 * clustering/radfriendsregion.py:58-182 and clustering/sdml.py:60-88) as ONE object behind the C
 * ABI: rebuild policy (:152-166,198-211), region construction with its bootstrap draws
 * (:48-92; neighbors.py:170-177), the candidate generators (radfriendsregion.py:117-182,
 * hiermetriclearn.py:104-137) and the accept loop (:181-196).  Every random number is taken from
 * numpy's OWN Mersenne-Twister state (the address the caller passes: `mt19937_state` behind
 * numpy.random's global legacy RandomState), with numpy's legacy algorithms, in the reference's
 * call order -- the stream advances exactly as if the reference had run.  The kernels are reached
 * through a table of C function pointers: on the GPU the mdns_backend_* entry points of
 * libmdns_hip.so below; tests put the CPU oracle behind the same table.
 * ------------------------------------------------------------------------------------------ */
#define MDNS_MAX_DIM 16

struct mdns_chain_request;
/* Device work of a draw.  `user` is passed back as the first argument of every function. */
typedef struct mdns_draw_backend {
	void *user;
	/* RadFriendsRegion(members, maxdistance) (radfriendsregion.py:59-70): members f64[K, ndim] in
	 * the metric's coordinates.  packed != NULL: K6 with that bootstrap choice (bit b of packed[i] =
	 * point i chosen in round b; cneighbors.c:125-179), *radius receives the result; packed == NULL:
	 * the region gets the radius *radius holds.  Returns an opaque region, NULL on failure. */
	void *(*region_create)(void *user, const double *members, int K, int ndim,
	                       const unsigned *packed, int nbootstraps, double *radius);
	void (*region_destroy)(void *user, void *region);
	/* K3 with the region's radius (cneighbors.c:95-119, no early stop): counts int32[n]. */
	int (*region_count)(void *user, void *region, const double *points, int n, int *counts);
	/* A constrained draw over the data sets rows int32[M] (ascending ORIGINAL indices; NULL: all,
	 * M = their number) begins: called once before its chunks. */
	int (*draw_begin)(void *user, const int *rows, int M);
	/* One chunk: params f64[B, nparams] of ALREADY proposed candidates, in order.  *accepted = the
	 * first that beats the threshold of some selected data set (hiermetriclearn.py:193), or -1;
	 * fillbits uint64[ceil(M/64)]: bit k set when it beats the k-th selected data set's
	 * (multi_nested_sampler.py:482-485); those data sets take the point in.  *nscored = candidates
	 * looked at (B, or fewer when the scorer stops at the accepted one).  jitter (NULL, or f64[B, M]):
	 * added to the likelihoods before they are compared and kept (musefuse.py:535). */
	int (*draw_chunk)(void *user, const double *params, int B, const double *jitter, int *accepted,
	                  unsigned long long *fillbits, int *nscored);
	/* How many of `offered` candidates one chunk should hold for M selected data sets when the last
	 * draw of this constrainer needed `hint` tries (a speed choice: results do not depend on it). */
	int (*chunk_size)(void *user, int offered, int M, int hint);
	/* Optional (NULL: region_create is used).  region_create with packed != NULL in two halves, so
	 * that the caller can work while K6 runs: region_begin uploads the members and launches K6 and
	 * returns the region at once, region_radius waits for the radius of such a region.  At most one
	 * region per backend is between the two calls at any time. */
	void *(*region_begin)(void *user, const double *members, int K, int ndim, const unsigned *packed, int nbootstraps);
	int (*region_radius)(void *user, void *region, double *radius);
	/* Optional (NULL: not offered).  The first batch of box proposals of a region between region_begin
	 * and region_radius WITHOUT a host look in between (radfriendsregion.py:135-141 + hiermetriclearn.py:
	 * 111-119,181-196): chain_begin queues, behind K6, the proposals lo + (hi - lo) u, their membership
	 * counts and -- when it can; limit > 0 asks for it -- the first chunk of the draw begun with
	 * draw_begin: the first min(kept, limit) proposals that are inside the region and, after the metric's
	 * inverse transform, inside the unit cube, prior-transformed, scored, accepted and committed like a
	 * draw_chunk.  chain_end waits for all of it: counts int32[n]; *nkept (-1: the chunk did not ride
	 * along), *B (its size), *accepted, fillbits, params f64[B][nparams] the device scored with (may be
	 * NULL).  The caller then asks region_radius as usual. */
	/* Optional (NULL: not offered): the likelihood noise in BAND form (musefuse.py:535 without one
	 * deviate per evaluation).  draw_band scores the chunk like draw_chunk, WITHOUT noise, and compares
	 * every likelihood with its threshold +- 1.01 bound[b] (bound f64[B]: no deviate of candidate b exceeds
	 * it): status int32[B] = 1 when some selected data set is beaten whatever the noise, 0 when none can
	 * be, 2 when that hangs on the pairs listed: pair_b / pair_k int32 (candidate, position in the
	 * selection), pair_L / pair_thr f64 (likelihood without noise, threshold); *npairs their number (more
	 * than cap: the caller falls back to draw_chunk with the whole block).  Nothing is committed.
	 * draw_band_commit: candidate b is the accepted one; jitter_row f64[M] is added to its likelihoods,
	 * then shelves, thresholds and fill bits as in draw_chunk. */
	int (*draw_band)(void *user, const double *params, int B, const double *bound, int *status, int *npairs,
	                 int *pair_b, int *pair_k, double *pair_L, double *pair_thr, int cap);
	int (*draw_band_commit)(void *user, int b, const double *jitter_row, unsigned long long *fillbits);
	int (*chain_begin)(void *user, void *region, const struct mdns_chain_request *rq);
	int (*chain_end)(void *user, void *region, int *counts, int *nkept, int *B, int *accepted,
	                 unsigned long long *fillbits, double *params);
	/* Optional (all three or none): draw_band in two halves with a cheap look in between.  While a chunk
	 * is being scored the constrainer already advances the stream for the candidates that FOLLOW it in
	 * the same batch (their bounds do not depend on the outcome; if the chunk accepts somebody that work is
	 * dropped), one candidate at a time until draw_band_ready returns nonzero. */
	int (*draw_band_begin)(void *user, const double *params, int B, const double *bound);
	int (*draw_band_ready)(void *user);
	int (*draw_band_end)(void *user, int *status, int *npairs, int *pair_b, int *pair_k, double *pair_L, double *pair_thr, int cap);
} mdns_draw_backend;

struct mdns_prior;
typedef struct mdns_chain_request {
	int n, ndim;                    /* proposals, dimensions */
	const double *u;                /* f64[n][ndim]: numpy's raw doubles of uniform(lo, hi, size=(n, ndim)) */
	const double *mn, *mx;          /* f64[ndim]: members.min(axis=0), members.max(axis=0) (radfriendsregion.py:69-70) */
	int identity;                   /* the metric is the identity; else x = y * scale + mean (sdml.py) */
	const double *mean, *scale;
	const struct mdns_prior *prior;
	int limit;                      /* candidates of the first chunk at most; 0: stop after the counts */
} mdns_chain_request;

/* The prior transform of the problem (sample.py:52-58) and the kernel's parameters (sample.py:103),
 * per dimension:  x[k] = a[k] * u[k] + b[k]  (the addition skipped when b[k] == 0), then
 * x[k] = 10 ** x[k] where pow10[k];  kernel parameter k = x[k], or 10 ** x[k] where
 * kernel_pow10[k].  `custom`, when set, replaces all of that: it fills x f64[B, ndim] and
 * params f64[B, nparams] from u f64[B, ndim] (a problem definition kept in Python). */
typedef struct mdns_prior {
	int ndim, nparams;
	double a[MDNS_MAX_DIM], b[MDNS_MAX_DIM];
	int pow10[MDNS_MAX_DIM], kernel_pow10[MDNS_MAX_DIM];
	void (*custom)(void *user, const double *u, int B, double *x, double *params);
	void *user;
	/* > 0: every likelihood evaluation adds N(0, jitter_sigma) noise per selected data set, drawn from
	 * the global stream in evaluation order -- `Lout[data_mask] + numpy.random.normal(0, 1e-5,
	 * size=data_mask.sum())`, musefuse.py:535.  The constrainer draws the deviates of a chunk's
	 * candidates one candidate after the other and, when one is accepted, puts the stream back to where
	 * it stood after THAT candidate's deviates (the reference never evaluates the ones behind it). */
	double jitter_sigma;
} mdns_prior;

/* numpy operations the constrainer leaves to numpy itself so that its numbers ARE numpy's (numpy
 * dispatches power/log2 to SIMD code whose last bits differ from the C library's):
 * vec_pow: inout[i] = numpy.power(inout[i], exponent)  (radfriendsregion.py:156);
 * fit_metric: mean f64[ndim], scale f64[ndim] of sdml.py:44-49 (kind 1) / :68-82 (kind 2) fitted to
 * u f64[K, ndim] shifted to its mean (hiermetriclearn.py:63-80); returns 0. */
typedef struct mdns_numpy_ops {
	void *user;
	void (*vec_pow)(void *user, double *inout, int n, double exponent);
	int (*fit_metric)(void *user, int kind, const double *u, int K, int ndim, double *mean, double *scale);
} mdns_numpy_ops;

typedef struct mdns_constrainer mdns_constrainer;

#define MDNS_METRIC_NONE 0
#define MDNS_METRIC_SIMPLESCALING 1
#define MDNS_METRIC_TRUNCATEDSCALING 2

/* MetricLearningFriendsConstrainer(metriclearner, rebuild_every, metric_rebuild_every,
 * force_shrink) (hiermetriclearn.py:28-46). */
mdns_constrainer *mdns_constrainer_create(int ndim, int metriclearner, int rebuild_every,
                                          int metric_rebuild_every, int force_shrink);
void mdns_constrainer_destroy(mdns_constrainer *c, const mdns_draw_backend *be);
/* `constrainer.region = None` (cachedconstrainer.py:104-105): the next draw builds a new region. */
void mdns_constrainer_forget_region(mdns_constrainer *c);

/* draw_constrained (hiermetriclearn.py:173-211).  live points: rows `ids` (int32 or int64 by
 * ids_itemsize, K of them, in the caller's order) of pile_u f64[npile, ndim]; selection rows/M as in
 * mdns_draw_backend.draw_begin.  Out: u f64[ndim] the accepted unit-cube point, x f64[ndim] its
 * prior transform, *ntries the likelihood calls the reference would have made (its `ntoaccept`),
 * fillbits uint64[ceil(M/64)].  Returns 0; on failure non-zero and mdns_host_last_error(). */
int mdns_constrainer_draw(mdns_constrainer *c, const mdns_draw_backend *be, const mdns_prior *prior,
                          const mdns_numpy_ops *np, void *mt19937_state,
                          const double *pile_u, const void *ids, int ids_itemsize, int K,
                          const int *rows, int M,
                          double *u, double *x, long long *ntries, unsigned long long *fillbits);
/* counters since creation, out int64[MDNS_CONSTRAINER_COUNTERS]: [0] draws, [1] chunks, [2] candidates
 * scored, [3] (candidate, data set) pairs scored, [4] regions built, [5] radius computations (K6),
 * [6] membership calls (K3), [7] raw proposals, [8] proposals the region kept, [9] tries (the
 * reference's likelihood calls); nanoseconds spent in [10] the bootstrap choices, [11] region_create
 * (K6 + upload), [12] region_count (K3), [13] proposal random numbers + arithmetic, [14] the prior
 * transform, [15] draw_chunk, [16] mdns_constrainer_draw as a whole, [17] the likelihood jitter;
 * [18] first batches chained on the device together with their chunk (chain_begin / chain_end), [19] with
 * their membership counts only, [20] accepted candidates of chained chunks whose device parameters were
 * not bit for bit the host's (10**v), [21] nanoseconds between chain_begin and chain_end; [22] pairs the
 * device could not decide without their noise (draw_band), [23] candidates whose noise was replayed for them,
 * [24] candidates whose bound was ready before their chunk (made while the previous chunk was scored).
 * mdns_constrainer_share_stats:
 * every increment is also added to totals int64[MDNS_CONSTRAINER_COUNTERS] (the caller's: the sum
 * over a sampler's constrainers). */
#define MDNS_CONSTRAINER_COUNTERS 25
void mdns_constrainer_stats(const mdns_constrainer *c, long long *out);
void mdns_constrainer_share_stats(mdns_constrainer *c, long long *totals);
const char *mdns_host_last_error(void);
/* The cached second deviate of numpy's legacy Gaussian generator (legacy-distributions.c,
 * `has_gauss` / `gauss`): the constrainer keeps it for the process like numpy's global RandomState
 * does; callers that mix their own numpy.random.normal calls with native draws hand it over. */
void mdns_host_rng_get_gauss(int *has_gauss, double *gauss);
void mdns_host_rng_set_gauss(int has_gauss, double gauss);

/* The backend of a Gaussian-line joint state on the GPU (libmdns_hip.so): the functions to put
 * into mdns_draw_backend, user = the mdns_joint handle.  region_create = K6 + resident members
 * (mdns_region_create_bootstrapped / mdns_region_create + mdns_region_set_radius), region_count =
 * mdns_region_count, draw_chunk = mdns_joint_draw_gauss without the likelihood row. */
void *mdns_backend_region_create(void *joint, const double *members, int K, int ndim,
                                 const unsigned *packed, int nbootstraps, double *radius);
void mdns_backend_region_destroy(void *joint, void *region);
int mdns_backend_region_count(void *joint, void *region, const double *points, int n, int *counts);
int mdns_backend_draw_begin(void *joint, const int *rows, int M);
int mdns_backend_draw_chunk(void *joint, const double *params, int B, const double *jitter, int *accepted,
                            unsigned long long *fillbits, int *nscored);
int mdns_backend_chunk_size(void *joint, int offered, int M, int hint);
void *mdns_backend_region_begin(void *joint, const double *members, int K, int ndim, const unsigned *packed, int nbootstraps);
int mdns_backend_region_radius(void *joint, void *region, double *radius);
/* mdns_backend_draw_chunk in two halves, for hosts that put something between them -- with the data sets
 * sharded over ranks a MAX all-reduce of the candidates' votes: `score` (after draw_begin)
 * stages and scores the chunk like draw_chunk and leaves one int32 0 / 1 vote per candidate in device
 * memory (mdns_joint_votes_dev: int32[MDNS_JOINT_MAX_BATCH], asynchronous on the library stream);
 * `commit` takes the first candidate that has a vote THEN as the accepted one and finishes like
 * draw_chunk (shelves, thresholds, *accepted, fill bits of this handle's selected data sets).  Both
 * kinds of joint state (Gaussian line; scale-marginalised likelihood with jitter f64[B, M]). */
int mdns_backend_draw_score(void *joint, const double *params, int B, const double *jitter);
int *mdns_joint_votes_dev(mdns_joint *j);
int mdns_backend_draw_commit(void *joint, int *accepted, unsigned long long *fillbits);
/* the HIP stream the library launches on (its own, or the one given to mdns_set_stream) */
void *mdns_get_stream(void);
int mdns_backend_draw_band(void *joint, const double *params, int B, const double *bound, int *status, int *npairs,
                           int *pair_b, int *pair_k, double *pair_L, double *pair_thr, int cap);
int mdns_backend_draw_band_commit(void *joint, int b, const double *jitter_row, unsigned long long *fillbits);
int mdns_backend_draw_band_begin(void *joint, const double *params, int B, const double *bound);
int mdns_backend_draw_band_ready(void *joint);
int mdns_backend_draw_band_end(void *joint, int *status, int *npairs, int *pair_b, int *pair_k, double *pair_L, double *pair_thr, int cap);
/* mdns_backend_draw_band scores large chunks (>= 8 candidates x >= 512 spectra) as two matrix products on
 * v_mfma_f64_16x16x4_f64 with the band widened by a rigorous bound on the rounding of that form
 * (csrc/mdns_k2gemm.hip); whatever that leaves undecided is scored again by the exact row kernels, and the
 * accepted candidate's row always is.  mode: -1 by shape (default; MDNS_K2_FILTER=0 / 1 at start), 0 never,
 * 1 for every chunk.  stats: chunks filtered | of those scored again exactly | exact rows made for a
 * commit | spectra handles prepared. */
void mdns_muse_filter_mode(int mode);
/* the filter pass alone on device pointers (asynchronous on the library stream): d_ypred f64[B][nx] templates,
 * d_thr f64[ndata] thresholds by data set, d_bound f64[B]; d_out int32[2 B + 1], zeroed by the caller: [b] set to
 * 1 when candidate b certainly beats a threshold, [B + b] when a pair of its cannot be settled, [2 B] the number
 * of such pairs */
int mdns_muse_filter_dev(mdns_spectra *s, const double *d_ypred, int B, const int *d_row_ids, int M, const double *d_thr,
                         const double *d_bound, int *d_out);
void mdns_muse_filter_stats(long long *out4);
int mdns_backend_chain_begin(void *joint, void *region, const mdns_chain_request *rq);
int mdns_backend_chain_end(void *joint, void *region, int *counts, int *nkept, int *B, int *accepted,
                           unsigned long long *fillbits, double *params);

/* ------------------------------------------------------------------------------------------
 * Part 6 -- one native call per nested-sampling ITERATION (libmdns_host.so, csrc/host_sampler.cpp).
 *
 * The integer side of MultiNestedSampler between `prepare` and the replacement of the dead points
 * (multi_nested_sampler.py:365-534): the passes over the data sets whose shelf is empty, the grouping
 * of the data sets that share live points (:204-355), the constrainer that serves each group
 * (cachedconstrainer.py:19-116: four-generation cache, "similar to the last call" shortcut, one
 * constrainer per single data set), the constrained draws (mdns_constrainer_draw) and the shelves'
 * queues of point ids (:474-489), the pile of accepted points, the superpoints.  The floating-point
 * side stays where it is: the joint state behind `mdns_draw_backend`.  Python keeps the reference's
 * class (massivedatans_amd.core.NativeCoreSampler: same constructor, iterator protocol, attributes).
 * ------------------------------------------------------------------------------------------ */
/* Connected components of big selections on the device: user = an mdns_groups handle, the three
 * functions = mdns_groups_components / mdns_groups_id_labels / mdns_groups_replace (Part 4).  Optional:
 * without it (and for selections of few (data set, live point) pairs) a union-find on the host gives
 * the same partition, component order and ascending id lists. */
typedef struct mdns_group_backend {
	void *user;
	int (*components)(void *user, const int32_t *rows, int M, long long npoints, int *ncomponents,
	                  long long *ndistinct, int32_t *distinct, long long cap, unsigned long long *touched);
	int (*id_labels)(void *user, int32_t *labels, int32_t *id_labels, long long ndistinct);
	int (*replace)(void *user, const int32_t *rows, const int32_t *slots, const int32_t *new_ids, int n);
} mdns_group_backend;

typedef struct mdns_core mdns_core;

/* MultiNestedSampler(...) (multi_nested_sampler.py:58-119) with the constrainer settings of the
 * reference's driver (sample.py:133-137).  be / prior / np / mt19937_state as in mdns_constrainer_draw
 * (borrowed for the life of the object); gb may be NULL; shelf_mirror (int64[ndata], may be NULL) is
 * incremented at the ORIGINAL index of every data set that shelves a point (the joint state's mirror of
 * its shelf sizes; the joint state itself takes entries off in its `advance`); constrainer_totals as in
 * mdns_constrainer_share_stats (may be NULL). */
mdns_core *mdns_core_create(int nlive, int ndata, int ndim, int nsuperset_draws, int use_graph,
                            int metriclearner, int rebuild_every, int metric_rebuild_every, int force_shrink,
                            const mdns_draw_backend *be, const mdns_prior *prior, const mdns_numpy_ops *np,
                            void *mt19937_state, const mdns_group_backend *gb, long long *shelf_mirror,
                            long long *constrainer_totals);
void mdns_core_destroy(mdns_core *c);
const char *mdns_core_last_error(void);
/* selections of at most this many (data set, live point) pairs are grouped on the host even when a
 * device backend is there (default 32768) */
void mdns_core_set_host_edges(mdns_core *c, long long edges);
/* The focussed passes of an iteration select ever fewer data sets: their components are analysed once
 * per iteration (selections of at most edges_max pairs; 0: never) and then kept up to date as data sets
 * leave (csrc/host_sampler_inc.h); check != 0 compares every such result with a fresh computation. */
void mdns_core_set_incremental(mdns_core *c, long long edges_max, int check);
/* the nlive prior draws every data set starts from (:88-103): u, x f64[nlive][ndim] */
int mdns_core_set_initial(mdns_core *c, const double *u, const double *x);
/* `prepare` (:137-138): keep uint8[nrunning][width]; entry e of the r-th running data set's shelf stays
 * when e < width and keep[r][e] */
int mdns_core_purge(mdns_core *c, const unsigned char *keep, int width);
/* :365-491: constrained draws until no running data set has an empty shelf */
int mdns_core_fill(mdns_core *c);
/* :494-534: the r-th running data set gives up the live point in slot argmin[r] and takes the head of
 * its shelf; dead_u, dead_x f64[nrunning][ndim] / dead_ids, new_ids int32[nrunning] may be NULL */
int mdns_core_advance(mdns_core *c, const int32_t *argmin, double *dead_u, double *dead_x,
                      int32_t *dead_ids, int32_t *new_ids);
/* cut_down (:148-173): surviving uint8[nrunning] */
int mdns_core_cut_down(mdns_core *c, const unsigned char *surviving);
long long mdns_core_npoints(const mdns_core *c);
int mdns_core_nrunning(const mdns_core *c);
/* the pile of accepted points f64[npoints][ndim] (valid until the next mdns_core_fill) */
const double *mdns_core_pile_u(const mdns_core *c);
const double *mdns_core_pile_x(const mdns_core *c);
/* live_pointsp int32[nlive][nrunning] (:108) */
int mdns_core_get_ids(const mdns_core *c, int32_t *out);
/* shelf sizes int32[nrunning] and (ids != NULL) the waiting ids, queue after queue; returns their number */
long long mdns_core_get_shelves(const mdns_core *c, int32_t *sizes, int32_t *ids, long long cap);
int mdns_core_get_superpoints(const mdns_core *c, int32_t *out, int cap);
/* out int64[MDNS_CORE_COUNTERS]: [0] ndraws (likelihood calls of the reference), [1] constrained draws,
 * [2] useful (candidate, data set) evaluations, [3] points in the pile, [4] iterations, [5] running data
 * sets, [6] superpoints, [7] passes, [8] groupings, of which [9] on the host, [10] on the device, [11] walks,
 * [12] constrainers created, nanoseconds in [13] mdns_constrainer_draw, [14] grouping, [15] mdns_core_fill,
 * [16] draws served by the "similar to the last call" shortcut, [17..24] groupings by selection size
 * (fewer than 2, 8, 32, 128, 512, 2048, 8192 data sets, more) and [25..32] the nanoseconds they took;
 * focussed groupings kept up to date: [33] analyses from scratch, [34] updates, [35] updates that split a component */
#define MDNS_CORE_COUNTERS 36
void mdns_core_stats(const mdns_core *c, long long *out);

#ifdef __cplusplus
}
#endif
#endif /* MDNS_H */
