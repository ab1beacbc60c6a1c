/* cmuselike.so drop-in: exports the reference's symbol `like` (cmuselike.c:34-38, loaded at
 * musefuse.py:505-508) and forwards to libmdns_hip.so.  No arithmetic here. */
#include "mdns.h"
int like(const void *yyp, const void *vvp, const void *ypredp, const void *data_maskp,
         const int ndata, const int nx, void *Loutp)
{
	return mdns_muse_like(yyp, vvp, ypredp, data_maskp, ndata, nx, Loutp);
}
