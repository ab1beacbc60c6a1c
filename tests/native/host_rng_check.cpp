// Bit-exactness of the blocked forms of numpy's legacy Gaussian stream (csrc/host_constrainer.cpp) against the
// deviate-by-deviate forms: gauss_fill == n calls of legacy_gauss (values, cached deviate, stream position) and
// BandLook::advance == drawing a candidate's M deviates (bound >= the largest, cache and stream position equal).
// Built and run by tests/test_host_rng.py (test infrastructure).
#include "../../massivedatans_amd/csrc/host_constrainer.cpp"

int main()
{
	MT mt;
	for (int i = 0; i < 624; i++) mt.key[i] = 1812433253u * i + 12345u;
	mt.pos = 624;
	int bad = 0;
	for (int trial = 0; trial < 1500; trial++) {
		MT a = mt, b = mt;
		const int n = trial < 200 ? trial : 48 + (trial * 7919) % 9000;
		const int pre = trial % 3;                              // deviates drawn before: both cache states
		std::vector<double> x(n), y(n);
		g_has_gauss = 0; g_gauss = 0;
		for (int t = 0; t < pre; t++) legacy_gauss(&a);
		for (int t = 0; t < n; t++) x[t] = legacy_gauss(&a);
		const int hx = g_has_gauss; const double gx = g_gauss;
		g_has_gauss = 0; g_gauss = 0;
		for (int t = 0; t < pre; t++) legacy_gauss(&b);
		gauss_fill(&b, y.data(), n);
		if ((n && memcmp(x.data(), y.data(), n * sizeof(double))) || hx != g_has_gauss || memcmp(&gx, &g_gauss, 8) || mt_double(&a) != mt_double(&b)) bad++;
		mt_next(&mt);
	}
	printf("gauss_fill mismatches: %d\n", bad);
	int wrong = 0;
	for (int M : {1, 2, 3, 7, 100, 209, 1000, 6250}) {
		g_has_gauss = 0; g_gauss = 0; legacy_gauss(&mt);           // a cached deviate to start with
		BandLook L;
		L.reset(&mt, M, 1.0);
		MT ref = mt; int rh = g_has_gauss; double rg = g_gauss;
		for (int chunk = 0; chunk < 4; chunk++) {
			// a chunk of 31 candidates and some way ahead, each checked as it is made; then the chunk is done
			while (L.count() < 31 + 9) {
				L.advance();
				g_has_gauss = rh; g_gauss = rg;
				double most = 0;
				for (int k = 0; k < M; k++) { const double g = std::fabs(legacy_gauss(&ref)); if (g > most) most = g; }
				rh = g_has_gauss; rg = g_gauss;
				MT chk;
				L.restore(&chk, L.snap.back());
				bool same = g_has_gauss == rh && memcmp(&g_gauss, &rg, 8) == 0 && L.bound.back() >= most;
				MT c1 = chk, c2 = ref;
				for (int q = 0; q < 700 && same; q++) same = mt_next(&c1) == mt_next(&c2);
				if (!same) wrong++;
			}
			L.base += 31;
			L.compact();
		}
	}
	printf("advance mismatches: %d\n", wrong);
	return bad || wrong ? 1 : 0;
}
