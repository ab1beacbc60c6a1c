/* cneighbors.so drop-in: exports the four symbols clustering/neighbors.py:100-166 binds
 * (cneighbors.c:32,77,95,125) and forwards to libmdns_hip.so.  No arithmetic here. */
#include "mdns.h"
double most_distant_nearest_neighbor(const void *xxp, int nsamples, int ndim)
{
	return mdns_most_distant_nearest_neighbor(xxp, nsamples, ndim);
}
int is_within_distance_of(const void *xxp, int nsamples, int ndim, double maxdistance, const void *yp)
{
	return mdns_is_within_distance_of(xxp, nsamples, ndim, maxdistance, yp);
}
int count_within_distance_of(const void *xxp, int nsamples, int ndim, double maxdistance,
                             const void *yyp, int nothers, void *outp, const int countmax)
{
	return mdns_count_within_distance_of(xxp, nsamples, ndim, maxdistance, yyp, nothers, outp, countmax);
}
double bootstrapped_maxdistance(const void *xxp, int nsamples, int ndim, const void *choicep,
                                int nbootstraps)
{
	return mdns_bootstrapped_maxdistance(xxp, nsamples, ndim, choicep, nbootstraps);
}
