"""Grouping of the data sets that share live points, on the device.

Host side of ``include/mdns.h`` Part 4 (``csrc/mdns_groups.hip``): the id matrix
``live_pointsp`` stays on the GPU, and what the reference asks igraph for in
``generate_subsets_graph`` (multi_nested_sampler.py:268-355) -- the connected components of the
bipartite graph {data sets} -- {live points}, and ``numpy.unique`` of the selected columns --
comes back as a component count, a bit map of the ids held and, only when there is more than
one component, the labels.  No CPU fallback: constructing the object needs the HIP library and
a device.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


class DeviceGroups(object):
    """``ids`` int[nlive, ndata]: the id matrix, one column per data set, indexed by the ORIGINAL
    data-set index for the whole run (columns of data sets that have finished are never read)."""

    def __init__(self, ids):
        lib = _lib.require_device()
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.nlive, self.ndata = ids.shape
        self._lib = lib
        self._h = lib.mdns_groups_create(self.nlive, self.ndata)
        if not self._h:
            raise _lib.MdnsError(_lib.last_error())
        _lib.check(lib.mdns_groups_set_ids(self._h, _lib.ptr(ids)), "mdns_groups_set_ids")
        self._distinct = np.empty(0, dtype=np.int32)
        self._rows = None
        self._npoints = 0
        self.ncalls = 0
        #: MDNS_GROUPS_LOG=1: (selected data sets, components, distinct ids) of every call (tools/e2e_run.py prints a summary)
        self.size_log = [] if os.environ.get("MDNS_GROUPS_LOG") == "1" else None

    def close(self):
        if self._h:
            self._lib.mdns_groups_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def mean_rounds(self):
        """Rounds of label propagation per :meth:`components` call so far, on average."""
        return float(self._lib.mdns_groups_mean_rounds(self._h))

    def ids(self):
        out = np.empty((self.nlive, self.ndata), dtype=np.int32)
        _lib.check(self._lib.mdns_groups_get_ids(self._h, _lib.ptr(out)), "mdns_groups_get_ids")
        return out

    def replace(self, rows, slots, new_ids):
        """Data set rows[i] gives up the live point in slot slots[i] and takes new_ids[i]
        (the end of an iteration, multi_nested_sampler.py:510-520)."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        new_ids = np.ascontiguousarray(new_ids, dtype=np.int32)
        assert len(rows) == len(slots) == len(new_ids)
        _lib.check(self._lib.mdns_groups_replace(self._h, _lib.ptr(rows), _lib.ptr(slots), _lib.ptr(new_ids), len(rows)),
                   "mdns_groups_replace")

    def components(self, rows, npoints):
        """Connected components over the data sets ``rows`` (ascending original indices; None:
        all).  Returns (number of components, the distinct ids they hold, ascending -- what
        ``numpy.unique(live_pointsp[:, rows])`` gives)."""
        npoints = int(npoints)
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.int32)
        M = self.ndata if rows is None else len(rows)
        cap = min(M * self.nlive, npoints)
        if len(self._distinct) < cap:
            self._distinct = np.empty(cap + cap // 2 + 64, dtype=np.int32)
        ncomp, ndistinct = C.c_int(0), C.c_longlong(0)
        _lib.check(self._lib.mdns_groups_components(self._h, None if rows is None else _lib.ptr(rows), M, npoints,
                                                    C.addressof(ncomp), C.addressof(ndistinct), _lib.ptr(self._distinct),
                                                    len(self._distinct), None), "mdns_groups_components")
        self._rows, self._npoints = rows, npoints
        self.ncalls += 1
        if self.size_log is not None:
            self.size_log.append((M, int(ncomp.value), int(ndistinct.value)))
        return int(ncomp.value), self._distinct[:ndistinct.value].astype(np.int64)

    def touched(self, npoints):
        """The ids held by the selection of a fresh :meth:`components` call over all data sets,
        as the bit map the library also offers (bit q of word q // 64)."""
        nwords = (int(npoints) + 63) // 64
        bits = np.zeros(nwords, dtype=np.uint64)
        ncomp, ndistinct = C.c_int(0), C.c_longlong(0)
        _lib.check(self._lib.mdns_groups_components(self._h, None, self.ndata, int(npoints), C.addressof(ncomp),
                                                    C.addressof(ndistinct), None, 0, _lib.ptr(bits)), "mdns_groups_components")
        self._rows, self._npoints = None, int(npoints)
        return bits

    def labels(self):
        """Of the last :meth:`components`: (label of every selected data set = the lowest
        data-set index of its component, label of every live point or -1)."""
        M = self.ndata if self._rows is None else len(self._rows)
        labels = np.empty(M, dtype=np.int32)
        point_labels = np.empty(self._npoints, dtype=np.int32)
        _lib.check(self._lib.mdns_groups_labels(self._h, _lib.ptr(labels), _lib.ptr(point_labels)), "mdns_groups_labels")
        return labels, point_labels

    def labels_of_ids(self, ndistinct):
        """Of the last :meth:`components`: (label of every selected data set, label of every id IT
        LISTED, in the order of that list) -- a few thousand numbers instead of one per id of the
        pile; once per components call."""
        M = self.ndata if self._rows is None else len(self._rows)
        labels = np.empty(M, dtype=np.int32)
        id_labels = np.empty(int(ndistinct), dtype=np.int32)
        _lib.check(self._lib.mdns_groups_id_labels(self._h, _lib.ptr(labels), _lib.ptr(id_labels), int(ndistinct)),
                   "mdns_groups_id_labels")
        return labels, id_labels

    def groups(self, rows, npoints):
        """[(original indices of the member data sets, ascending; their distinct ids, ascending)],
        components in order of their lowest data set (igraph numbers clusters by their first
        vertex; data-set vertices come first in the reference's graph, multi_nested_sampler.py:177-180)."""
        ncomp, ids = self.components(rows, npoints)
        every = np.arange(self.ndata) if rows is None else np.asarray(rows)
        if ncomp == 1:
            return [(every, ids)]
        labels, of_id = self.labels_of_ids(len(ids))
        return [(every[labels == root], ids[of_id == root]) for root in np.unique(labels)]
