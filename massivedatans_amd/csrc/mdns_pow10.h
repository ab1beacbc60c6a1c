// 10**v in double-double arithmetic, shared by host and device code (plain C++ with fma()).
//
// Where it is used: the constrained draw that runs WITHOUT a host look in between (mdns_chain.hip)
// transforms its candidates' unit-cube coordinates on the device -- sample.py:52-58,103:
// A = 10**(2u - 2), sig = 10**(2u) -- while the accepted point's physical coordinates, which are
// results, are still computed by the host with the C library's pow, like the reference.  The
// likelihood of a candidate depends on the parameters to a relative 1e-16 per ulp, the same order
// as the difference between the device's exp and the C library's, so the device value only has to
// be as good as an ulp; this one is correctly rounded except in about one argument in 10^4 (internal
// relative error ~2^-67), and the host counts the accepted candidates whose device parameters are
// not bit for bit its own (mdns_constrainer_stats).
//
// Method: y = v log2(10) as a double-double; y = e + j/64 + r with integers e, j in [0, 64) and
// |r| <= 1/128; 2^(j/64) from a table of 64 double-doubles; 2^r = exp(r ln2) by its series, the terms
// above 2^-15 with their rounding errors carried; the result scaled by 2^e.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MDNS_POW10_FN __host__ __device__ inline
#define MDNS_POW10_TABLE static __device__ __constant__
#else
#define MDNS_POW10_FN inline
#define MDNS_POW10_TABLE static
#endif

namespace mdns_pow10 {

// 2^(j/64) = hi + lo
static const double kExp2Host[64][2] = {
	{0x1.0000000000000p+0, 0x0.0p+0},
	{0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56},
	{0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55},
	{0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57},
	{0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54},
	{0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59},
	{0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54},
	{0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54},
	{0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55},
	{0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55},
	{0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54},
	{0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55},
	{0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54},
	{0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55},
	{0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55},
	{0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54},
	{0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55},
	{0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54},
	{0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54},
	{0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56},
	{0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55},
	{0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58},
	{0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59},
	{0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56},
	{0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56},
	{0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54},
	{0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55},
	{0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54},
	{0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54},
	{0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54},
	{0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54},
	{0x1.6623882552225p+0, -0x1.bb60987591c34p-54},
	{0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54},
	{0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57},
	{0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55},
	{0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54},
	{0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55},
	{0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56},
	{0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54},
	{0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54},
	{0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54},
	{0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55},
	{0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57},
	{0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54},
	{0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56},
	{0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54},
	{0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54},
	{0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54},
	{0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54},
	{0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57},
	{0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56},
	{0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55},
	{0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55},
	{0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54},
	{0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56},
	{0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54},
	{0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55},
	{0x1.da9e603db3285p+0, 0x1.c2300696db532p-54},
	{0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54},
	{0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55},
	{0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54},
	{0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54},
	{0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54},
	{0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55},
};
#if defined(__HIPCC__)
MDNS_POW10_TABLE const double kExp2Dev[64][2] = {
	{0x1.0000000000000p+0, 0x0.0p+0},
	{0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56},
	{0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55},
	{0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57},
	{0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54},
	{0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59},
	{0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54},
	{0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54},
	{0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55},
	{0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55},
	{0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54},
	{0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55},
	{0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54},
	{0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55},
	{0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55},
	{0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54},
	{0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55},
	{0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54},
	{0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54},
	{0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56},
	{0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55},
	{0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58},
	{0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59},
	{0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56},
	{0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56},
	{0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54},
	{0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55},
	{0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54},
	{0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54},
	{0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54},
	{0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54},
	{0x1.6623882552225p+0, -0x1.bb60987591c34p-54},
	{0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54},
	{0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57},
	{0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55},
	{0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54},
	{0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55},
	{0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56},
	{0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54},
	{0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54},
	{0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54},
	{0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55},
	{0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57},
	{0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54},
	{0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56},
	{0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54},
	{0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54},
	{0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54},
	{0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54},
	{0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57},
	{0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56},
	{0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55},
	{0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55},
	{0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54},
	{0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56},
	{0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54},
	{0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55},
	{0x1.da9e603db3285p+0, 0x1.c2300696db532p-54},
	{0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54},
	{0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55},
	{0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54},
	{0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54},
	{0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54},
	{0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55},
};
#endif

MDNS_POW10_FN double pow10_dd(double v)
{
	const double L_hi = 0x1.a934f0979a371p+1, L_lo = 0x1.7f2495fb7fa6dp-53;      // log2(10)
	const double N_hi = 0x1.62e42fefa39efp-1, N_lo = 0x1.abc9e3b39803fp-56;      // ln(2)
	if (!(v > -300.0 && v < 300.0)) return pow(10.0, v);                       // (never on the draw path)
	const double y_hi = v * L_hi;
	const double y_lo = fma(v, L_hi, -y_hi) + v * L_lo;
	const double k = rint(y_hi * 64.0);
	const double r_hi = y_hi - k * 0.015625;                                     // exact (Sterbenz)
	const int ki = (int) k;
	const int j = ki & 63, e = (ki - j) / 64;
	// t = (r_hi + y_lo) ln2
	const double t_hi = r_hi * N_hi;
	const double t_lo = fma(r_hi, N_hi, -t_hi) + (r_hi * N_lo + y_lo * N_hi);
	// exp(t) - 1 = t + t^2/2 + t^3/6 + ...   (|t| <= 0.0055)
	const double s_hi = t_hi * t_hi;
	const double s_lo = fma(t_hi, t_hi, -s_hi) + 2.0 * (t_hi * t_lo);
	const double tail = t_hi * s_hi * (1.0 / 6 + t_hi * (1.0 / 24 + t_hi * (1.0 / 120 + t_hi * (1.0 / 720 +
	                    t_hi * (1.0 / 5040 + t_hi * (1.0 / 40320))))));
	const double q = 0.5 * s_hi;
	// a = t_hi + q as a double-double, the small terms added to its low part
	const double a_hi = t_hi + q;
	const double bb = a_hi - t_hi;
	double a_lo = (t_hi - (a_hi - bb)) + (q - bb);
	a_lo += t_lo + 0.5 * s_lo + tail;
#if defined(__HIP_DEVICE_COMPILE__)
	const double T_hi = kExp2Dev[j][0], T_lo = kExp2Dev[j][1];
#else
	const double T_hi = kExp2Host[j][0], T_lo = kExp2Host[j][1];
#endif
	// T (1 + a) = T_hi + (T_hi a_hi + (T_lo + T_hi a_lo + T_lo a_hi))
	const double p_hi = T_hi * a_hi;
	const double p_lo = fma(T_hi, a_hi, -p_hi) + (T_hi * a_lo + T_lo * a_hi);
	const double r1 = T_hi + p_hi;
	const double c = r1 - T_hi;
	const double r2 = ((T_hi - (r1 - c)) + (p_hi - c)) + (p_lo + T_lo);
	return ldexp(r1 + r2, e);
}

}  // namespace mdns_pow10
