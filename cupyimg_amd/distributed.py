"""Slab-parallel filtering across the GPUs of one node.

New design (the reference is single-GPU): a volume
is partitioned along axis 0 into one contiguous slab per rank, one process per
GPU.  Output plane z needs input planes z-lo .. z+hi with

    lo = w0 // 2 + origin0,   hi = w0 - 1 - lo

(offset rule of cupyimg/scipy/ndimage/_filters_core.py:10-11), so a rank
receives `lo` planes from its predecessor and `hi` planes from its successor.
That neighbour exchange -- RCCL send/recv pairs in one group, each over one
xGMI link -- is the only communication; axes 1 and 2 need none.

Every rank keeps its slab inside an *extended* buffer
``[lo halo | local planes | hi halo]``.  Any filter of this package can then be
run on the extended buffer as if it were a stand-alone volume: the local
output planes only depend on real data, and at a global edge (no halo) the
buffer edge *is* the volume edge, so the boundary mode is evaluated exactly as
in the unsplit volume.  ``wrap`` closes the chain (rank 0 <-> rank P-1).

`SlabPlan` is pure host logic (tested on CPU with a gloo world of 2);
`HaloComm` is the RCCL transport behind the C-ABI.
"""
import ctypes

import numpy as np

from . import _lib, core


def halo_widths(size, origin=0):
    """(lo, hi) planes needed below / above a slab for a filter of `size` taps."""
    lo = size // 2 + origin
    hi = size - 1 - lo
    if lo < 0 or hi < 0:
        raise ValueError("invalid origin")
    return lo, hi


class SlabPlan:
    """Partition of `nz` planes over `nranks` ranks plus the halo bookkeeping."""

    def __init__(self, nz, nranks, rank, lo, hi, wrap=False):
        if not 0 <= rank < nranks:
            raise ValueError("rank out of range")
        self.nz, self.nranks, self.rank = int(nz), int(nranks), int(rank)
        self.lo, self.hi, self.wrap = int(lo), int(hi), bool(wrap)
        base, extra = divmod(self.nz, self.nranks)
        counts = [base + (1 if r < extra else 0) for r in range(self.nranks)]
        if min(counts) < max(self.lo, self.hi, 1):
            raise ValueError("slabs of {} planes are thinner than the halo ({}, {})".format(
                min(counts), self.lo, self.hi))
        starts = np.concatenate([[0], np.cumsum(counts)])
        self.z0, self.z1 = int(starts[rank]), int(starts[rank + 1])
        self.n_local = self.z1 - self.z0
        self.counts = counts
        closed = self.wrap and self.nranks > 1
        self.prev = rank - 1 if rank > 0 else (self.nranks - 1 if closed else -1)
        self.next = rank + 1 if rank < self.nranks - 1 else (0 if closed else -1)
        # halo planes actually present in the extended buffer
        self.lo_present = self.lo if self.prev >= 0 else 0
        self.hi_present = self.hi if self.next >= 0 else 0
        self.n_ext = self.lo_present + self.n_local + self.hi_present

    # indices into the extended buffer
    @property
    def local_slice(self):
        return slice(self.lo_present, self.lo_present + self.n_local)

    def send_to_prev(self):
        """planes (as a slice of the extended buffer) the predecessor needs: my first `hi`"""
        return slice(self.lo_present, self.lo_present + self.hi) if self.prev >= 0 and self.hi else None

    def send_to_next(self):
        """my last `lo` local planes"""
        end = self.lo_present + self.n_local
        return slice(end - self.lo, end) if self.next >= 0 and self.lo else None

    def recv_from_prev(self):
        return slice(0, self.lo_present) if self.lo_present else None

    def recv_from_next(self):
        end = self.lo_present + self.n_local
        return slice(end, end + self.hi_present) if self.hi_present else None

    def global_planes_of_ext(self):
        """global plane index held by each plane of the extended buffer (mod nz for wrap)"""
        idx = np.arange(self.z0 - self.lo_present, self.z1 + self.hi_present)
        return idx % self.nz if self.wrap else idx


class HaloComm:
    """RCCL communicator for the neighbour exchange (one per process / GPU)."""

    def __init__(self, nranks, rank, exchange_id):
        """`exchange_id(id_bytes_or_None) -> id_bytes`: rank 0 passes the id it
        created, every rank gets rank 0's id back (e.g. a torch.distributed or
        MPI broadcast; this package does not depend on either)."""
        lib = _lib.load()
        uid = None
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            _lib.check(lib.mi_comm_unique_id(buf))
            uid = buf.raw
        uid = exchange_id(uid)
        self._comm = ctypes.c_void_p()
        _lib.check(lib.mi_comm_init_rank(ctypes.byref(self._comm), nranks, rank, uid))
        self.nranks, self.rank = nranks, rank

    def exchange(self, ext, plan):
        """Fill the halo planes of the extended buffer `ext` (device array,
        C-contiguous, axis 0 = planes) from the neighbours; asynchronous on the
        default stream."""
        if ext.shape[0] != plan.n_ext or not ext._is_c_contiguous():
            raise ValueError("extended buffer does not match the plan")
        plane_bytes = ext.nbytes // max(ext.shape[0], 1)
        lib = _lib.load()
        # the C entry point takes the symmetric layout [lo | local | hi]; a
        # missing neighbour simply means that side is absent (width 0 there)
        base = ext.ptr - (plan.lo - plan.lo_present) * plane_bytes
        _lib.check(lib.mi_halo_exchange(self._comm, ctypes.c_void_p(base), plane_bytes, plan.n_local,
                                        plan.lo, plan.hi, plan.prev, plan.next, None))

    def close(self):
        if self._comm:
            _lib.load().mi_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlabFilter:
    """Runs ``fn(ext_in, ext_out)`` (any filter of this package, output given)
    on a rank's extended slab after a halo exchange."""

    def __init__(self, plan, plane_shape, dtype, comm=None):
        self.plan, self.comm = plan, comm
        self.ext_in = core.empty((plan.n_ext,) + tuple(plane_shape), dtype)
        self.ext_out = core.empty((plan.n_ext,) + tuple(plane_shape), dtype)

    @property
    def local_in(self):
        return self.ext_in[self.plan.local_slice]

    @property
    def local_out(self):
        return self.ext_out[self.plan.local_slice]

    def step(self, fn):
        if self.comm is not None and self.plan.nranks > 1:
            self.comm.exchange(self.ext_in, self.plan)
        fn(self.ext_in, self.ext_out)
        return self.local_out
