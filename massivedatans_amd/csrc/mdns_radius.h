// shared by the kernels that finish a radius computation on the device (mdns_neighbors.hip, mdns_k6sort.hip)
#pragma once

namespace mdns {

// Radius and membership threshold of a region, on the device: radius = sqrt(max_b round_sq[b])
// (cneighbors.c:160-174; sqrt after the max, monotone) and thresh = the smallest double T
// with sqrt(T) >= radius, so that  sqrt(d) < radius  <=>  d < T  (cneighbors.c:88,109).  Same
// bisection over bit patterns as mdns::sqrt_threshold on the host; hipcc's sqrt(double) is
// correctly rounded (verified bit for bit against the host on 1.6e7 inputs, and the parity
// tests compare both paths), so the two agree exactly.  Run by one lane of the last workgroup
// of a radius computation (normally ~15 square roots).
static __device__ __forceinline__ void radius_and_threshold(double max_sq, double &radius, double &thresh)
{
	const double r = sqrt(max_sq);     // sqrt after the max: same number, sqrt is monotone
	double T;
	if (!(r > 0.0)) T = 0.0;                       // nothing is strictly within a zero radius
	else if (r == __longlong_as_double(0x7ff0000000000000LL)) T = r;
	else {
		// T lies within a few ulps of r*r: walk there, and keep the full bisection for the
		// cases where r*r leaves the normal range or the walk does not settle
		const double t0 = r * r;
		unsigned long long u = (unsigned long long) __double_as_longlong(t0);
		bool settled = false;
		if (t0 > 1e-300 && t0 < 1e300) {
			int guard = 0;
			while (sqrt(__longlong_as_double((long long) u)) < r && guard < 8) { u++; guard++; }
			while (guard < 16 && sqrt(__longlong_as_double((long long) (u - 1))) >= r) { u--; guard++; }
			settled = guard < 16 && sqrt(__longlong_as_double((long long) u)) >= r &&
			          sqrt(__longlong_as_double((long long) (u - 1))) < r;
		}
		if (!settled) {
			unsigned long long lo = 0, hi = 0x7ff0000000000000ULL;
			while (hi - lo > 1) {
				const unsigned long long mid = lo + (hi - lo) / 2;
				if (sqrt(__longlong_as_double((long long) mid)) >= r) hi = mid; else lo = mid;
			}
			u = hi;
		}
		T = __longlong_as_double((long long) u);
	}
	radius = r;
	thresh = T;
}

}  // namespace mdns
