"""RCCL on the library's stream, called directly (ctypes over ``librccl.so``).

The exchanges of the sharded path are tiny -- B accept flags, one fill bit per data set, a few
hundred live points -- so what they cost is the host side of issuing them.  Going through
``torch.distributed`` costs ~25 us of Python/c10d per collective plus an event hop between
torch's stream and RCCL's; a direct ``ncclAllReduce`` on the stream the kernels are on is one
enqueue.  ``torch.distributed`` (any backend) is still what launches the ranks and carries the
128-byte communicator id from rank 0 to the others: plumbing.

Only what the path needs: all-gather and all-reduce on device pointers.
"""
import ctypes as C
import os

INT32, INT64, UINT64, FLOAT64 = 2, 4, 5, 8          # ncclDataType_t (rccl.h)
SUM, PROD, MAX, MIN = 0, 1, 2, 3                    # ncclRedOp_t
UNIQUE_ID_BYTES = 128


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


_LIB = None


def lib():
    """librccl.so: the copy torch ships when torch is in the process (one RCCL per process),
    otherwise the ROCm one."""
    global _LIB
    if _LIB is not None:
        return _LIB
    candidates = []
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    except ImportError:
        pass
    candidates += ["librccl.so", "/opt/rocm/lib/librccl.so"]
    last = None
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            L = C.CDLL(path)
            break
        except OSError as e:
            last = e
    else:
        raise RcclError("librccl.so not found: %s" % last)
    L.ncclGetErrorString.restype = C.c_char_p
    L.ncclGetErrorString.argtypes = [C.c_int]
    L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    _LIB = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RcclError("%s: %s" % (what, (lib().ncclGetErrorString(rc) or b"?").decode()))


class Communicator(object):
    """One communicator over all ranks.  ``exchange(payload)`` must return rank 0's payload on
    every rank (rank 0 passes its 128 id bytes, the others None).  The calling thread's current
    HIP device is the communicator's device."""

    def __init__(self, world, rank, exchange):
        L = lib()
        uid = _UniqueId()
        if rank == 0:
            _check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = exchange(bytes(bytearray(uid.internal)) if rank == 0 else None)
        if raw is None or len(raw) != UNIQUE_ID_BYTES:
            raise RcclError("communicator id exchange failed")
        C.memmove(C.byref(uid), raw, UNIQUE_ID_BYTES)
        self.world, self.rank = int(world), int(rank)
        self._comm = C.c_void_p()
        _check(L.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_gather(self, send, recv, count, dtype, stream):
        """recv[r * count : (r + 1) * count] = rank r's send[0 : count] (elements of dtype)."""
        _check(lib().ncclAllGather(C.c_void_p(send), C.c_void_p(recv), count, dtype, self._comm, C.c_void_p(stream)),
               "ncclAllGather")

    def all_reduce(self, send, recv, count, dtype, op, stream):
        """recv = op over the ranks of send (in place when the pointers are equal)."""
        _check(lib().ncclAllReduce(C.c_void_p(send), C.c_void_p(recv), count, dtype, op, self._comm, C.c_void_p(stream)),
               "ncclAllReduce")

    def destroy(self):
        if self._comm:
            lib().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()


def from_torch_distributed():
    """A communicator over the ranks of the initialised default process group (the id travels
    through it as a pickled object, whatever its backend)."""
    import torch.distributed as dist

    def exchange(payload):
        box = [payload]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    return Communicator(dist.get_world_size(), dist.get_rank(), exchange)
