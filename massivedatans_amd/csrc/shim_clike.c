/* clike.so drop-in: exports the reference's symbol `like` (clike.c:34-40, loaded at
 * sample.py:82-84) and forwards to libmdns_hip.so.  No arithmetic here. */
#include "mdns.h"
int like(const void *xp, const void *yyp, const int ndata, const int nx, const double A,
         const double mu, const double sig, const double noise_level, const void *data_maskp,
         void *Loutp)
{
	return mdns_gauss_like(xp, yyp, ndata, nx, A, mu, sig, noise_level, data_maskp, Loutp);
}
