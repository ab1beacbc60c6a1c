/*
 * mdns_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the arithmetic of the reference's native hot path, used only
 * as the checker in tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline leg.
 * Nothing under massivedatans_amd/ may load, link or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle.py shows every function below to be
 * bit-identical to the reference's own C (compiled from /root/reference by oracle/Makefile
 * into oracle/_ref/) on seeded inputs, and the .npz fixtures in tests/golden (made by oracle/make_golden.py
 * from oracle/_ref) pins it where /root/reference is absent (the GPU box).
 *
 * Every function keeps the reference's evaluation ORDER, because without fast-math the C
 * source order is the definition of the result (reference Makefile:2-6: -O3 -std=c99, no
 * -march => no FMA contraction, no reassociation).  This file is built with
 * -ffp-contract=off for the same reason.
 *
 * Reference map (paths under /root/reference):
 *   orc_gauss_like                 <- clike.c:34-89       (serial branch :64-76)
 *   orc_muse_like                  <- cmuselike.c:34-66
 *   orc_nn_maxdist                 <- clustering/cneighbors.c:32-75
 *   orc_any_within                 <- clustering/cneighbors.c:77-92
 *   orc_count_within               <- clustering/cneighbors.c:95-119
 *   orc_bootstrap_maxdist          <- clustering/cneighbors.c:125-179
 */
#include <math.h>
#include <stdbool.h>
#include <stdlib.h>

/* squared Euclidean distance, summed from 0 over k ascending (cneighbors.c:55-58, :83-86,
 * :104-107, :152-155 all use this same accumulation) */
static inline double sqdist(const double *a, const double *b, int ndim)
{
	double acc = 0;
	for (int k = 0; k < ndim; k++) {
		const double diff = a[k] - b[k];
		acc += diff * diff;            /* reference: pow(diff,2), which gcc lowers to diff*diff */
	}
	return acc;
}

/* Gaussian emission line evaluated at one wavelength (clike.c:65) */
static inline double line_profile(double A, double mu, double sig, double xj)
{
	const double t = (mu - xj) / sig;
	return A * exp(-0.5 * (t * t));
}

/*
 * K1.  chi^2-like sum for a single Gaussian line against every masked spectrum.
 * clike.c:64-76: channel-outer loop, the model value is computed once per channel, and the
 * k-th masked data set accumulates with += into Lout[k] (compacted, caller pre-zeroes).
 * yy is [nx, ndata] C-order: element (j,i) at i + j*ndata (clike.c:72).
 * With ORACLE_OMP the data-set axis is split over threads; each Lout[k] still sums its
 * channels in ascending order, so the result is bit-identical to the serial loop (this is
 * the *correct* parallel form; the reference's own PARALLEL branch, clike.c:47-62, races on k
 * and is never loaded, sample.py:81).
 */
int orc_gauss_like(const double *x, const double *yy, int ndata, int nx,
                   double A, double mu, double sig, double noise_level,
                   const bool *data_mask, double *Lout)
{
#ifdef ORACLE_OMP
	int *slot = (int *) malloc(sizeof(int) * (size_t) (ndata > 0 ? ndata : 1));
	int nsel = 0;
	for (int i = 0; i < ndata; i++)
		if (data_mask[i]) slot[nsel++] = i;
	double *model = (double *) malloc(sizeof(double) * (size_t) (nx > 0 ? nx : 1));
	for (int j = 0; j < nx; j++) model[j] = line_profile(A, mu, sig, x[j]);
	#pragma omp parallel for schedule(static)
	for (int k = 0; k < nsel; k++) {
		const int i = slot[k];
		double acc = Lout[k];
		for (int j = 0; j < nx; j++) {
			const double r = (model[j] - yy[(size_t) i + (size_t) j * ndata]) / noise_level;
			acc += r * r;
		}
		Lout[k] = acc;
	}
	free(model);
	free(slot);
#else
	for (int j = 0; j < nx; j++) {
		const double model = line_profile(A, mu, sig, x[j]);
		const double *row = yy + (size_t) j * ndata;
		int k = 0;
		for (int i = 0; i < ndata; i++) {
			if (!data_mask[i]) continue;
			const double r = (model - row[i]) / noise_level;
			Lout[k] += r * r;
			k++;
		}
	}
#endif
	return 0;
}

/*
 * K2.  Amplitude-marginalised chi^2 with per-pixel variances, cmuselike.c:45-64.
 * Two passes per masked data set i: first the best scale s = s1/s2 (s2 seeded with 1e-10,
 * :52), then chi.  Output is NOT compacted and only masked entries are written (:49,:62).
 * Expression order follows :54-55 ((y*m)/v and (m*m)/v) and :60 ((y - s*m)^2 / v).
 */
int orc_muse_like(const double *yy, const double *vv, const double *ypred,
                  const bool *data_mask, int ndata, int nx, double *Lout)
{
#ifdef ORACLE_OMP
	#pragma omp parallel for schedule(static)
#endif
	for (int i = 0; i < ndata; i++) {
		if (!data_mask[i]) continue;
		double num = 0.;
		double den = 1e-10;
		for (int j = 0; j < nx; j++) {
			const size_t e = (size_t) i + (size_t) j * ndata;
			num += yy[e] * ypred[j] / vv[e];
			den += (ypred[j] * ypred[j]) / vv[e];
		}
		const double s = num / den;
		double chi = 0.;
		for (int j = 0; j < nx; j++) {
			const size_t e = (size_t) i + (size_t) j * ndata;
			const double resid = yy[e] - s * ypred[j];
			chi += (resid * resid) / vv[e];
		}
		Lout[i] = -0.5 * chi;
	}
	return 0;
}

/*
 * K5.  max_i sqrt(min_{j != i} |x_i - x_j|^2), cneighbors.c:47-74.  The nearest squared
 * distance starts at 1e300 (:51) and the square root is taken after the min (:64).
 */
double orc_nn_maxdist(const double *xx, int nsamples, int ndim)
{
	double worst = 0;
	for (int i = 0; i < nsamples; i++) {
		double nearest = 1e300;
		for (int j = 0; j < nsamples; j++) {
			if (j == i) continue;
			const double d = sqdist(xx + (size_t) i * ndim, xx + (size_t) j * ndim, ndim);
			if (d < nearest) nearest = d;
		}
		const double root = sqrt(nearest);
		/* :66-71: running max seeded with element 0, then strict > for the rest */
		if (i == 0 || root > worst) worst = root;
	}
	return worst;
}

/* K4.  1 if any member lies strictly closer than maxdistance (sqrt THEN compare, :88). */
int orc_any_within(const double *xx, int nsamples, int ndim, double maxdistance, const double *y)
{
	for (int i = 0; i < nsamples; i++)
		if (sqrt(sqdist(xx + (size_t) i * ndim, y, ndim)) < maxdistance)
			return 1;
	return 0;
}

/*
 * K3.  For each of the nothers points: out[j] is INCREMENTED (it is a double, :100,:110) once
 * per member strictly within maxdistance, members scanned in ascending order; when
 * countmax > 0 the scan of that point stops as soon as out[j] >= countmax (:112-114).
 */
int orc_count_within(const double *xx, int nsamples, int ndim, double maxdistance,
                     const double *yy, int nothers, double *out, int countmax)
{
	for (int j = 0; j < nothers; j++) {
		const double *cand = yy + (size_t) j * ndim;
		for (int i = 0; i < nsamples; i++) {
			if (!(sqrt(sqdist(xx + (size_t) i * ndim, cand, ndim)) < maxdistance)) continue;
			out[j] += 1;
			if (countmax > 0 && out[j] >= countmax) break;
		}
	}
	return 0;
}

/*
 * K6.  RadFriends bootstrapped radius, cneighbors.c:137-176.  chosen is a DOUBLE matrix
 * [nsamples, nbootstraps], element (i,b) at i*nbootstraps+b, tested against 0 (:146,:150).
 * Per round: for every left-out point the nearest chosen point (squared distance from 1e300,
 * sqrt after the min, :148-160); the round's value is the max of those roots over left-out
 * points with index >= 1 ONLY (:162 starts at i = 1 -- point 0 never contributes), starting
 * from 0 (:142).  Result = max over rounds (:170-174).
 */
double orc_bootstrap_maxdist(const double *xx, int nsamples, int ndim,
                             const double *chosen, int nbootstraps)
{
	double best = 0;
#ifdef ORACLE_OMP
	#pragma omp parallel for schedule(dynamic) reduction(max:best)
#endif
	for (int b = 0; b < nbootstraps; b++) {
		double round_max = 0;
		for (int i = 1; i < nsamples; i++) {
			if (chosen[(size_t) i * nbootstraps + b] != 0) continue;
			double nearest = 1e300;
			for (int j = 0; j < nsamples; j++) {
				if (chosen[(size_t) j * nbootstraps + b] == 0) continue;
				const double d = sqdist(xx + (size_t) i * ndim, xx + (size_t) j * ndim, ndim);
				if (d < nearest) nearest = d;
			}
			const double root = sqrt(nearest);
			if (root > round_max) round_max = root;
		}
		/* :170-174 seeds the max with round 0 and uses strict > afterwards; every round
		 * value is >= 0 so a max seeded with 0 is the same number */
		if (round_max > best) best = round_max;
	}
	return best;
}
