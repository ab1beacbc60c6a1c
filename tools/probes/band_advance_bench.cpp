// The advance of numpy's legacy Gaussian stream for MUSE-style noise bounds (csrc/host_constrainer.cpp, BandLook):
// ns per deviate in chunks of 57 candidates, blocks made by the caller itself or by the two helper threads.
//   g++ -O3 -fPIC -ffp-contract=off -Iinclude -std=c++17 -fno-exceptions tools/probes/band_advance_bench.cpp \
//       massivedatans_amd/csrc/host_rng.o -o /tmp/band_bench -lm -lpthread
//   /tmp/band_bench; MDNS_BAND_THREADS=1 /tmp/band_bench
#include "../../massivedatans_amd/csrc/host_constrainer.cpp"
#include <chrono>
static double now() { return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	MT mt; for (int i = 0; i < 624; i++) mt.key[i] = 1812433253u * i + 12345u; mt.pos = 624;
	g_has_gauss = 0;
	printf("helper threads: %s\n", block_producer() ? "on" : "off");
	for (int rep = 0; rep < 3; rep++)
	for (int M : {209, 6250}) {
		BandLook L; L.reset(&mt, M, 1e-5);
		const int chunks = M == 209 ? 2000 : 80;
		const double t0 = now();
		for (int c = 0; c < chunks; c++) {
			while (L.count() < 57 + 20) L.advance();          // the chunk and some way ahead
			L.restore(&mt, L.snap[L.base + 57]);
			L.base += 57;
			L.compact();
		}
		printf("M %d: %.3f ns per deviate\n", M, (now() - t0) / chunks / 57 / M);
	}
	// a flag passed between two threads
	return 0;
}
