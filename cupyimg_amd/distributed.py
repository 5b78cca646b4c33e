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

Three schedules of a step (exchange + filter), all bit-identical in their result:

    plain       exchange, then one launch over the local planes, one stream
    overlapped  interior planes while the exchange is in flight, edge planes afterwards (two launches, two
                cross-stream waits on the critical path: pays for large halos when every step depends on the last)
    pipelined   `SlabPipeline` (r4): two or three resident input slabs, the exchange of the NEXT input underneath the
                single launch of the current one -- for sequences of independent volumes (and benchmarks that filter
                a resident volume repeatedly); step time = max(kernel, exchange)
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

    @classmethod
    def self_loop(cls, nz, lo, hi):
        """Plan of a closed chain of ONE rank that is both neighbours of itself (a one-rank communicator): the halos
        are the periodic continuation of the rank's own planes, i.e. `wrap` along axis 0.  What single-GPU tests and
        `bench.py --self-loop` run the whole multi-rank code path on -- exchange, schedules, pipeline."""
        plan = cls.__new__(cls)
        plan.nz, plan.nranks, plan.rank = int(nz), 2, 0          # nranks > 1 selects the exchange path
        plan.lo, plan.hi, plan.wrap = int(lo), int(hi), True
        if nz < max(lo, hi, 1):
            raise ValueError("slab thinner than the halo")
        plan.z0, plan.z1, plan.n_local = 0, plan.nz, plan.nz
        plan.counts = [plan.nz]
        plan.prev = plan.next = 0
        plan.lo_present, plan.hi_present = plan.lo, plan.hi
        plan.n_ext = plan.lo + plan.nz + plan.hi
        return plan

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


    def check_reach(self, size, origin=0):
        """Raise ValueError unless an axis-0 kernel of `size` taps stays inside
        the halo this plan exchanges (otherwise the planes next to a neighbour
        would see the boundary mode at an interior slab edge)."""
        lo, hi = halo_widths(int(size), int(origin))
        if (self.prev >= 0 and lo > self.lo) or (self.next >= 0 and hi > self.hi):
            raise ValueError("axis-0 kernel of {} taps (origin {}) needs a halo of ({}, {}) planes; the slab plan "
                             "exchanges ({}, {})".format(size, origin, lo, hi, self.lo, self.hi))

    def plane_ranges(self):
        """(interior, edges): output plane ranges of the extended buffer that
        can be filtered before / only after the halo exchange."""
        a = self.lo_present
        b = a + self.n_local
        ib = a + (self.lo if self.prev >= 0 else 0)
        ie = b - (self.hi if self.next >= 0 else 0)
        if ib >= ie:                       # slab so thin that every plane touches a halo
            return [], [(a, b)]
        return [(ib, ie)], [(a, ib), (ie, b)]


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

    def exchange(self, ext, plan, stream=None):
        """Fill the halo planes of the extended buffer `ext` (device array,
        C-contiguous, axis 0 = planes) from the neighbours; asynchronous on
        `stream` (a core.Stream; default: the library's default stream)."""
        if ext.shape[0] != plan.n_ext or not ext._is_c_contiguous():
            raise ValueError("extended buffer does not match the plan")
        plane_bytes = ext.nbytes // max(ext.shape[0], 1)
        lib = _lib.load()
        # the C entry point takes the symmetric layout [lo | local | hi]; a
        # missing neighbour simply means that side is absent (width 0 there)
        base = ext.ptr - (plan.lo - plan.lo_present) * plane_bytes
        _lib.check(lib.mi_halo_exchange(self._comm, ctypes.c_void_p(base), plane_bytes, plan.n_local,
                                        plan.lo, plan.hi, plan.prev, plan.next,
                                        None if stream is None else stream.handle))

    def sendrecv(self, ops, stream=None):
        """`ops`: [(device_array_view, peer, is_send)] -- C-contiguous views; all transfers in one RCCL group
        (mi_comm_sendrecv).  Between two ranks the sends of one must come in the order of the other's receives."""
        ops = [(a, int(peer), bool(snd)) for a, peer, snd in ops if a.nbytes]
        n = len(ops)
        if n == 0:
            return
        for a, _, _ in ops:
            if not a._is_c_contiguous():
                raise ValueError("sendrecv needs C-contiguous views")
        ptrs = (ctypes.c_void_p * n)(*[a.ptr for a, _, _ in ops])
        sizes = (ctypes.c_size_t * n)(*[a.nbytes for a, _, _ in ops])
        peers = (ctypes.c_int * n)(*[p_ for _, p_, _ in ops])
        snd = (ctypes.c_int * n)(*[1 if s_ else 0 for _, _, s_ in ops])
        _lib.check(_lib.load().mi_comm_sendrecv(self._comm, n, ptrs, sizes, peers, snd,
                                                None if stream is None else stream.handle))

    def close(self):
        if self._comm:
            _lib.load().mi_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------
# Output-sharded interpolation (last sentence): map_coordinates / affine_transform have no
# neighbourhood structure to exchange halos for -- the OUTPUT (and the coordinates) are partitioned by z-slabs, every
# rank gathers from the input.  Either the input is replicated (each rank holds the whole volume: nothing to send), or
# it is slab-distributed and each rank fetches ONCE the input planes its output planes read -- the pre-image slab,
# known from the matrix (affine) or from the min / max of the rank's coordinates -- with point-to-point RCCL transfers
# from the ranks that own them.  No collective, nothing communicated per voxel.
# ----------------------------------------------------------------------------------------------------------------
def interp_reach(order):
    """(below, above): taps of an order-0 / order-1 interpolation relative to floor(coordinate)."""
    if order not in (0, 1):
        raise ValueError("the pre-image of spline orders >= 2 is the whole line (the prefilter is recursive): replicate the "
                         "input or prefilter it first")
    return (1, 1) if order == 0 else (0, 1)      # order 0 rounds: floor(c + 0.5) is floor(c) or floor(c) + 1


def preimage_planes(cmin, cmax, nz, order=1, mode="constant"):
    """[a, b): input planes along axis 0 that coordinates in [cmin, cmax] (axis 0) read.  Index-mapping modes fold a
    coordinate outside the volume back inside -- anywhere: the pre-image is then the whole axis."""
    if not (np.isfinite(cmin) and np.isfinite(cmax)):
        return 0, int(nz)
    below, above = interp_reach(order)
    hair = 1e-6 * (1.0 + max(abs(cmin), abs(cmax)))
    a = int(np.floor(cmin - hair)) - below
    b = int(np.floor(cmax + hair)) + above + 1
    if mode not in ("constant", "nearest", "grid-constant") and (a < 0 or b > nz):
        return 0, int(nz)
    a, b = max(a, 0), min(b, int(nz))
    if b <= a:                                   # everything outside: one plane keeps the call well-formed
        a = min(max(a, 0), int(nz) - 1)
        b = a + 1
    return a, b


def affine_axis0_range(matrix, offset, out_shape, z0, z1):
    """(cmin, cmax) of the axis-0 input coordinate over output planes z0 .. z1 - 1 of an affine_transform."""
    m = np.asarray(matrix, dtype=np.float64)
    off = np.asarray(offset, dtype=np.float64)
    if m.ndim == 1:
        m = np.diag(m)
    lo = hi = float(off[0])
    ext = [(z0, z1 - 1)] + [(0, int(n) - 1) for n in out_shape[1:]]
    for j, (e0, e1) in enumerate(ext):
        v0, v1 = m[0, j] * e0, m[0, j] * e1
        lo += min(v0, v1)
        hi += max(v0, v1)
    return lo, hi


class ShardedInterp:
    """Output-sharded `affine_transform` / `map_coordinates` over the ranks of one node.

    `out_plan`: SlabPlan of the OUTPUT planes (halo 0).  `in_plan`: SlabPlan of the INPUT planes when the input is
    slab-distributed (rank r holds input planes in_plan.z0 .. z1), None when every rank holds the whole input.
    `allgather(list_of_ints) -> list of per-rank lists` (host, e.g. torch.distributed.all_gather_object): only needed
    for map_coordinates on a distributed input (each rank's pre-image depends on its own coordinates)."""

    def __init__(self, out_plan, comm=None, in_plan=None, allgather=None):
        self.out_plan, self.comm, self.in_plan, self.allgather = out_plan, comm, in_plan, allgather
        if in_plan is not None and in_plan.nranks != out_plan.nranks:
            raise ValueError("input and output plans differ in the number of ranks")

    # ---- what touches the device, in one place each (tests/test_distributed_gloo.py runs the partition / pre-image /
    # pairing logic above these on host arrays, with gloo as the transport and the CPU oracle as the kernel)
    def _alloc(self, shape, dtype):
        return core.empty(shape, dtype)

    def _run_affine(self, src, m, off, shape, output, order, mode, cval, prefilter):
        from .scipy import ndimage as ndi
        return ndi.affine_transform(src, m, off, output_shape=shape, output=output, order=order, mode=mode, cval=cval,
                                    prefilter=prefilter)

    def _run_map(self, src, coordinates, output, order, mode, cval, prefilter):
        from .scipy import ndimage as ndi
        return ndi.map_coordinates(src, coordinates, output=output, order=order, mode=mode, cval=cval, prefilter=prefilter)

    def _axis0_min_max(self, coordinates):
        from .scipy.ndimage import _support as S
        return S.min_max(coordinates[0])

    def _shift_axis0(self, coordinates, a):
        """the coordinates with axis 0 shifted by -a, as a new array (exact: a is an integer below the coordinate)"""
        from .scipy.ndimage import _support as S
        c = coordinates.copy()
        c[0] = S.scale_shift(c[0], 1.0, -float(a))
        return c

    # ---- the pre-image slab of a distributed input
    def _gather(self, local_in, needs):
        """`needs[q] = (a, b)`: input planes rank q reads.  Returns (sub, a): this rank's planes a .. b as one
        contiguous device array, fetched from their owners (one RCCL group; own planes by a device copy)."""
        ip, r = self.in_plan, self.in_plan.rank
        a, b = needs[r]
        starts = np.concatenate([[0], np.cumsum(ip.counts)])
        plane_shape = tuple(local_in.shape[1:])
        sub = self._alloc((b - a,) + plane_shape, local_in.dtype)
        ops = []
        for q in range(ip.nranks):                              # ascending peer order on both sides of every pair
            if q == r:
                continue
            qa, qb = needs[q]
            s0, s1 = max(qa, ip.z0), min(qb, ip.z1)             # what q needs of mine
            if s1 > s0:
                ops.append((local_in[s0 - ip.z0:s1 - ip.z0], q, True))
            g0, g1 = max(a, int(starts[q])), min(b, int(starts[q + 1]))      # what I need of q's
            if g1 > g0:
                ops.append((sub[g0 - a:g1 - a], q, False))
        o0, o1 = max(a, ip.z0), min(b, ip.z1)
        if o1 > o0:
            sub[o0 - a:o1 - a] = local_in[o0 - ip.z0:o1 - ip.z0]
        if ops:
            if self.comm is None:
                raise ValueError("a communicator is required to fetch input planes from other ranks")
            self.comm.sendrecv(ops)
        return sub, a

    def affine_transform(self, input, matrix, offset=0.0, output_shape=None, output=None, order=1, mode="constant",
                         cval=0.0, prefilter=True):
        """This rank's planes (out_plan.z0 .. z1) of ``affine_transform(whole_input, matrix, offset, output_shape)``.
        `input`: the whole volume (replicated input) or this rank's input slab (`in_plan` given).  `output_shape` is
        the shape of the WHOLE output (default: the whole input's).  Distributed inputs: orders 0 and 1."""
        op = self.out_plan
        m = np.array(matrix.get() if isinstance(matrix, core.ndarray) else matrix, dtype=np.float64)
        nd = input.ndim
        off = np.full(nd, float(offset)) if np.isscalar(offset) else np.array(
            offset.get() if isinstance(offset, core.ndarray) else offset, dtype=np.float64)
        if m.ndim == 1:
            m = np.diag(m)
        if m.shape == (nd, nd + 1) or m.shape == (nd + 1, nd + 1):
            off, m = m[:nd, nd].copy(), m[:nd, :nd].copy()
        nz_in = self.in_plan.nz if self.in_plan is not None else input.shape[0]
        if output_shape is None:
            output_shape = (nz_in,) + tuple(input.shape[1:])
        if int(output_shape[0]) != op.nz:
            raise ValueError("output_shape[0] = {} but the output plan partitions {} planes".format(output_shape[0], op.nz))
        local_shape = (op.n_local,) + tuple(int(v) for v in output_shape[1:])
        off_local = off + m[:, 0] * op.z0                          # output plane k of this rank is global plane z0 + k
        if self.in_plan is None:
            src, a = input, 0
            if order <= 1:
                # a view of the planes this rank reads: smaller gathers' working set, same result (preimage_planes)
                a, b = preimage_planes(*affine_axis0_range(m, off, output_shape, op.z0, op.z1), nz_in, order, mode)
                src = input[a:b]
        else:
            needs = []
            starts = np.concatenate([[0], np.cumsum(op.counts)])
            for q in range(op.nranks):
                needs.append(preimage_planes(*affine_axis0_range(m, off, output_shape, int(starts[q]), int(starts[q + 1])),
                                             nz_in, order, mode))
            src, a = self._gather(input, needs)
        off_local = off_local.copy()
        off_local[0] -= a                                          # coordinates in the frame of the planes held
        return self._run_affine(src, m, off_local, local_shape, output, order, mode, cval, prefilter)

    def map_coordinates(self, input, coordinates, output=None, order=1, mode="constant", cval=0.0, prefilter=True):
        """This rank's part of ``map_coordinates(whole_input, whole_coordinates)``: `coordinates` are THIS rank's slab
        of the coordinate array (ndim, n_local, ...).  Replicated input: a plain local call.  Distributed input
        (`in_plan`): the rank's pre-image planes are fetched first (min / max of its axis-0 coordinates, all-gathered on
        the host as two ints per rank)."""
        if self.in_plan is None:
            return self._run_map(input, coordinates, output, order, mode, cval, prefilter)
        if self.allgather is None and self.in_plan.nranks > 1:
            raise ValueError("map_coordinates on a distributed input needs `allgather`")
        lo, hi = self._axis0_min_max(coordinates)
        mine = [int(v) for v in preimage_planes(float(lo), float(hi), self.in_plan.nz, order, mode)]
        needs = [tuple(v) for v in self.allgather(mine)] if self.in_plan.nranks > 1 else [tuple(mine)]
        sub, a = self._gather(input, needs)
        if a:
            coordinates = self._shift_axis0(coordinates, a)      # the frame of the planes held
        return self._run_map(sub, coordinates, output, order, mode, cval, prefilter)


class SlabPipeline:
    """The pipelined schedule (r4, `mi_slab_pipe_*`): `nbuf` resident input slabs, ONE whole-slab launch per step, the
    halo exchange of the next input on a high-priority comm stream underneath it.

        comm stream : | X(k+1) ......... | X(k+2) ......... |
        stream      : | filter(k) ...... | filter(k+1) ... |

    For sequences of independent volumes: write the local planes of input j (`local_in(j)`), `submit(j)` -- its halo
    exchange is queued behind everything written so far -- and later `compute(j)`; `run(n)` rotates over resident
    inputs (benchmarks, repeated filtering).  Results are bit-identical to the plain schedule: the same kernel on the
    same extended slab.  Created by `SlabFilter.pipeline(...)` / `.uniform_pipeline` / `.gaussian_pipeline`."""

    def __init__(self, plan, comm, ext_ins, ext_out, weights, modes, cval, origins):
        from .scipy.ndimage import _support as S
        self.plan, self.comm = plan, comm
        self.inputs, self.ext_out = list(ext_ins), ext_out
        if weights[0] is not None and len(weights[0]) > 1:
            plan.check_reach(len(weights[0]), origins[0])
        modes = S.normalize_sequence(modes, 3)
        for m in modes:
            S.check_mode(m)
        keep = [None if w is None else np.ascontiguousarray(w, dtype=np.float64) for w in weights]
        dp = ctypes.POINTER(ctypes.c_double)
        ptrs = (dp * 3)(*[ctypes.cast(None, dp) if w is None else w.ctypes.data_as(dp) for w in keep])
        descs = [a._desc() for a in self.inputs]
        arr_p = ctypes.POINTER(_lib.MiArray)
        in_ptrs = (arr_p * len(descs))(*[ctypes.pointer(d) for d in descs])
        out_desc = ext_out._desc()
        self._pipe = ctypes.c_void_p()
        _lib.check(_lib.load().mi_slab_pipe_create(
            ctypes.byref(self._pipe), comm._comm if comm is not None else None, len(descs), in_ptrs, ctypes.byref(out_desc),
            ptrs, S.c_ints([0 if w is None else len(w) for w in keep]), S.c_ints(origins),
            S.c_ints([S.mode_code(m) for m in modes]), float(cval), plan.lo, plan.hi, plan.prev, plan.next, None))
        self.nbuf = len(descs)
        # for `measure` (the same launch without the pipe); also keeps the host buffers alive
        self._kernel_args = (ptrs, S.c_ints([0 if w is None else len(w) for w in keep]), S.c_ints(origins),
                             S.c_ints([S.mode_code(m) for m in modes]), float(cval))
        self._keep = (keep, descs, out_desc)

    def local_in(self, k):
        return self.inputs[k][self.plan.local_slice]

    @property
    def local_out(self):
        return self.ext_out[self.plan.local_slice]

    def submit(self, k):
        """Input k is final (as of everything queued on the default stream so far): exchange its halos."""
        _lib.check(_lib.load().mi_slab_pipe_step(self._pipe, int(k), -1))

    def compute(self, k):
        """Filter input k (submitted before) into the output slab; returns this rank's output planes."""
        _lib.check(_lib.load().mi_slab_pipe_step(self._pipe, -1, int(k)))
        return self.local_out

    def step(self, submit, compute):
        """`submit(submit)` then `compute(compute)` in ONE native call (the steady-state step: submit the input
        nbuf - 1 steps ahead, filter the current one)."""
        _lib.check(_lib.load().mi_slab_pipe_step(self._pipe, int(submit), int(compute)))
        return self.local_out

    def run(self, nsteps, graph=0):
        """`nsteps` steps of the rotation over the resident inputs; graph > 0: replay a captured hipGraph of one
        rotation (graph = 1) or of `graph` steps where the capture works (`info()`), direct queuing otherwise."""
        _lib.check(_lib.load().mi_slab_pipe_run(self._pipe, int(nsteps), int(graph)))
        return self.local_out

    def info(self):
        g, n, pk = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().mi_slab_pipe_info(self._pipe, ctypes.byref(g), ctypes.byref(n), ctypes.byref(pk)))
        return {"graph_state": g.value, "graph_steps": n.value, "planes_ok": bool(pk.value), "nbuf": self.nbuf}

    def measure(self, reps=20):
        """(kernel_us, exchange_us) of this rank: the filter launch alone and the RCCL exchange alone, each the
        average of `reps` back-to-back repetitions on the default stream (HIP events); exchange_us is None without
        neighbours.  Collective (it exchanges halos): every rank must call it, with the pipeline idle.  bench.py
        prints both for every rank."""
        core.synchronize()
        k_us = self._kernel_only_us(reps)
        ex_us = None
        if self.comm is not None and self.plan.nranks > 1:
            for _ in range(2):
                self.comm.exchange(self.inputs[0], self.plan)
            e0, e1 = core.Event(), core.Event()
            e0.record()
            for _ in range(reps):
                self.comm.exchange(self.inputs[0], self.plan)
            e1.record()
            e1.synchronize()
            ex_us = e0.elapsed_ms(e1) / reps * 1e3
        core.synchronize()
        return k_us, ex_us

    def _kernel_only_us(self, reps):
        from . import _lib as L
        lib = L.load()
        plan = self.plan
        a = plan.lo_present
        planes = (ctypes.c_int64 * 2)(a, a + plan.n_local)
        info = self.info()
        d_in, d_out = self.inputs[0]._desc(), self.ext_out._desc()
        args = self._kernel_args
        e0, e1 = core.Event(), core.Event()

        def launch():
            if info["planes_ok"]:
                L.check(lib.mi_separable3d_f32_planes(ctypes.byref(d_in), ctypes.byref(d_out), *args, planes, 1, None))
            else:
                L.check(lib.mi_separable3d_f32(ctypes.byref(d_in), ctypes.byref(d_out), *args, 0, None))
        for _ in range(3):
            launch()
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        e1.synchronize()
        return e0.elapsed_ms(e1) / reps * 1e3

    def close(self):
        if getattr(self, "_pipe", None):
            _lib.load().mi_slab_pipe_destroy(self._pipe)
            self._pipe = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlabFilter:
    """Runs ``fn(ext_in, ext_out)`` (any filter of this package, output given)
    on a rank's extended slab after a halo exchange.

    `step` is the plain schedule: exchange, then filter the whole extended
    buffer, all on the default stream.  `step_overlapped` hides the exchange:

        comm stream    : [ wait input ready ][ RCCL send/recv of the halos ]
        default stream : [ filter interior planes ........ ][ wait ][ filter edge planes ]

    The interior planes (those whose taps stay inside the local planes) do not
    need the halos; only the `lo` + `hi` planes next to a neighbour wait for the
    exchange, and they are filtered in one small launch afterwards.
    """

    def __init__(self, plan, plane_shape, dtype, comm=None, reduce_max=None):
        """`reduce_max(list_of_floats) -> list_of_floats` (optional): element-wise maximum over the ranks (e.g. an
        all-reduce on the host); with it the measured schedule choice is agreed by all ranks from the slowest rank's
        timings instead of being taken by every rank on its own."""
        self.plan, self.comm = plan, comm
        self.reduce_max = reduce_max
        self.ext_in = core.empty((plan.n_ext,) + tuple(plane_shape), dtype)
        self.ext_out = core.empty((plan.n_ext,) + tuple(plane_shape), dtype)
        self._comm_stream = None
        self._halos_ready = None      # exchange finished (recorded on the comm stream)
        self._input_free = None       # previous readers of ext_in finished (default stream)
        self._extra_inputs = []       # further resident input slabs of the pipelined schedule
        self._overlap_refused = set() # filters (keys) whose kernels take no plane ranges: plain schedule, not re-probed
        self._prepared = {}
        self._native_refused = set()  # filters mi_slab_separable3d_f32 answered UNSUPPORTED for (generic step instead)
        self._tuning = {}             # per filter: schedule measurements / choice (see _tuned_schedule)
        self.autotune = True          # overlap=None: measure plain vs overlapped inside the first call (warm)

    @property
    def local_in(self):
        return self.ext_in[self.plan.local_slice]

    @property
    def local_out(self):
        return self.ext_out[self.plan.local_slice]

    def check_reach(self, size, origin=0):
        """See SlabPlan.check_reach."""
        self.plan.check_reach(size, origin)

    def step(self, fn):
        """Exchange the halos, then run ``fn(ext_in, ext_out)``.  Contract: along
        axis 0 `fn` may read at most `plan.lo` planes below and `plan.hi` planes
        above an output plane (`check_reach` tests a kernel length against the
        plan); a wider filter silently applies its boundary mode at the slab
        edges."""
        if self.comm is not None and self.plan.nranks > 1:
            self.comm.exchange(self.ext_in, self.plan)
        fn(self.ext_in, self.ext_out)
        return self.local_out

    def step_overlapped(self, fn, key=None):
        """Same result as `step` for the local planes, with the exchange
        overlapped with the interior filtering.  `fn` must honour
        `_support.output_planes` (the fused separable and min / max kernels do);
        anything else raises Unsupported on its first plane-restricted call and the
        step finishes on the plain schedule -- whichever launch refused, interior or
        edge, on every rank alike: exactly one exchange has been queued by then, and
        the refusal is remembered per filter (`key`), not for the whole SlabFilter."""
        from .scipy.ndimage import _support as S
        if self.comm is None or self.plan.nranks == 1 or key in self._overlap_refused:
            return self.step(fn)
        self._streams()
        interior, edges = self.plan.plane_ranges()
        # the exchange overwrites the halo planes: it has to wait for whatever
        # was queued on the default stream so far (producers of the local
        # planes, readers of the old halos)
        self._input_free.record()
        self._comm_stream.wait_event(self._input_free)
        self.comm.exchange(self.ext_in, self.plan, self._comm_stream)
        self._halos_ready.record(self._comm_stream)
        try:
            if interior:
                with S.output_planes(interior):
                    fn(self.ext_in, self.ext_out)
            core.default_stream_wait_event(self._halos_ready)
            with S.output_planes(edges):
                fn(self.ext_in, self.ext_out)
        except S.Unsupported:
            self._overlap_refused.add(key)
            core.default_stream_wait_event(self._halos_ready)
            fn(self.ext_in, self.ext_out)
        return self.local_out

    # ---------------------------------------------------------------- native step
    def _streams(self):
        if self._comm_stream is None:
            self._comm_stream = core.Stream()
            self._halos_ready, self._input_free = core.Event(), core.Event()

    def separable(self, weights, modes="reflect", cval=0.0, origins=(0, 0, 0), fallback=None, overlap=None,
                  _key=None):
        """One overlapped step of a separable filter given per-axis 1-D weights
        (None = axis not filtered) as ONE native call
        (mi_slab_separable3d_f32); the marshalled arguments are cached, so a
        repeated step costs a few microseconds of host time.  `fallback(ext_in,
        ext_out)` runs under the plain schedule when the fused kernel does not
        cover the request.  overlap: True / False; default None = measured
        inside the first call (`_tuned_schedule`, `warm`; with `autotune = False`: decided by
        the halo size in C, overlapping from ~8 MiB per direction)."""
        from .scipy.ndimage import _support as S
        plan = self.plan
        key = _key
        if key is None:
            key = (tuple(None if w is None else tuple(np.asarray(w, dtype=np.float64)) for w in weights),
                   str(modes), float(cval), tuple(int(o) for o in origins))
        prep = self._prepared.get(key)
        if prep is None:
            if weights[0] is not None and len(weights[0]) > 1:
                self.check_reach(len(weights[0]), origins[0])
            modes = S.normalize_sequence(modes, 3)
            for m in modes:
                S.check_mode(m)
            keep = [None if w is None else np.ascontiguousarray(w, dtype=np.float64) for w in weights]
            dp = ctypes.POINTER(ctypes.c_double)
            ptrs = (dp * 3)(*[ctypes.cast(None, dp) if w is None else w.ctypes.data_as(dp) for w in keep])
            a, b = self.ext_in._desc(), self.ext_out._desc()
            self._streams()
            args = (self.comm._comm if self.comm is not None else None, ctypes.byref(a), ctypes.byref(b), ptrs,
                    S.c_ints([0 if w is None else len(w) for w in keep]), S.c_ints(origins),
                    S.c_ints([S.mode_code(m) for m in modes]), float(cval), plan.lo, plan.hi, plan.prev, plan.next, 0,
                    self._comm_stream.handle, self._input_free._e, self._halos_ready._e, None)
            prep = self._prepared[key] = (args, (keep, a, b))        # second item keeps the buffers alive
        if key in self._native_refused and fallback is not None:
            return self.step(fallback)             # the fused kernels do not take this request: remembered, not re-probed
        args = prep[0]
        lib = _lib.load()
        try:
            if overlap is not None:
                mode_flag = int(bool(overlap)) if key not in self._overlap_refused else 0
            elif self.comm is None or plan.nranks == 1 or not self.autotune:
                mode_flag = -1
            else:
                mode_flag = self._tuned_schedule(key, args)
            try:
                _lib.check(lib.mi_slab_separable3d_f32(*args[:12], mode_flag, *args[13:]))
            except _lib.Unsupported:
                # refusals come before anything is queued (mi_slab_separable3d_f32): the overlapped form does not
                # exist for this kernel (no plane ranges) -> the plain native schedule, remembered per filter
                if mode_flag == 0:
                    raise
                self._overlap_refused.add(key)
                _lib.check(lib.mi_slab_separable3d_f32(*args[:12], 0, *args[13:]))
        except _lib.Unsupported:
            if fallback is None:
                raise
            self._native_refused.add(key)
            return self.step(fallback)
        return self.local_out

    def _tuned_schedule(self, key, args, samples=3, burst=10):
        """Plain or overlapped schedule for this filter, measured instead of guessed: whether hiding the exchange
        behind the interior planes pays depends on the link (xGMI latency and bandwidth for THIS halo size) and on
        the cost of the cross-stream waits, and neither can be known from a one-GPU box.  The FIRST call of a filter
        tunes (`warm()`; never inside a later step): one warm step of each schedule, then `samples` bursts of `burst`
        back-to-back steps of each, alternating, ONE host synchronisation per burst -- what a caller that issues steps
        back to back sees (r3 timed single steps with a host sync after each, i.e. latency: in the plain schedule the
        exchange of step k + 1 queues behind kernel k on one stream, in the overlapped one it runs under it).
        With `reduce_max` the per-schedule medians are maximised over the ranks before the choice, so every rank takes
        the SAME schedule, the one that is faster for the slowest rank; without it each rank decides alone (still
        correct: every step performs exactly one exchange whatever the schedule).  Whether the overlapped form exists
        is asked up front (mi_separable3d_f32_supports: same answer on every rank), so no probe ever queues an
        exchange and then refuses.  Returns the flag for mi_slab_separable3d_f32."""
        st = self._tuning.get(key)
        if st is not None:
            return st["choice"]
        lib = _lib.load()
        overlap_supported = lib.mi_separable3d_f32_supports(args[1], args[2], args[3], args[4], args[5], args[6], args[7], 1) == _lib.MI_OK
        st = {"t": [[], []], "overlap_supported": overlap_supported, "burst": int(burst)}
        flags = (0, 1) if overlap_supported else (0,)

        def run(flag, n):
            for _ in range(n):
                _lib.check(lib.mi_slab_separable3d_f32(*args[:12], flag, *args[13:]))

        for flag in flags:
            run(flag, 2)
        e0, e1 = core.Event(), core.Event()
        for _ in range(max(int(samples), 1)):
            for flag in flags:
                e0.record()
                run(flag, st["burst"])
                e1.record()
                e1.synchronize()
                st["t"][flag].append(e0.elapsed_ms(e1) / st["burst"])
        med = [float(np.median(t)) if t else 1e30 for t in st["t"]]
        st["median_ms_local"] = list(med)
        if self.reduce_max is not None:
            med = [float(v) for v in self.reduce_max(med)]
            st["agreed"] = True
        st["median_ms"] = [None if v >= 1e30 else v for v in med]
        st["choice"] = 1 if overlap_supported and med[1] < 0.97 * med[0] else 0
        self._tuning[key] = st
        return st["choice"]

    def warm(self, step):
        """Run ``step()`` once so that everything a repeated step amortises is done: argument marshalling, the
        streams / events, and -- with more than one rank and `autotune` -- the schedule measurement of
        `_tuned_schedule`.  Benchmarks call it before their own warm-up, so no tuning step can fall into a timed
        region whatever `--warmup` is.  Collective: every rank must call it (it exchanges halos)."""
        step()
        core.synchronize()

    def schedule_of(self, key_prefix):
        """{"choice": 0 plain / 1 overlapped, "median_ms": [plain, overlapped]} of the tuned filters whose key starts
        with `key_prefix` (e.g. "uniform"), or None -- what bench.py prints as the partition label."""
        for k, st in self._tuning.items():
            if isinstance(k, tuple) and k and k[0] == key_prefix:
                return {"choice": st["choice"], "median_ms": st.get("median_ms"), "median_ms_local": st.get("median_ms_local"),
                        "overlap_supported": st["overlap_supported"], "agreed_across_ranks": bool(st.get("agreed")),
                        "burst": st.get("burst")}
        return None

    def uniform_filter(self, size, mode="reflect", cval=0.0, overlap=None):
        """uniform_filter of the whole (distributed) volume; returns this rank's planes."""
        from .scipy import ndimage as ndi
        from .scipy.ndimage import _support as S
        key = ("uniform", str(size), str(mode), float(cval))
        weights = None
        if key not in self._prepared:
            sizes = [int(v) for v in S.normalize_sequence(size, 3)]
            weights = [np.full((sz,), 1.0 / sz) if sz > 1 else None for sz in sizes]
        return self.separable(weights, mode, cval, overlap=overlap, _key=key,
                              fallback=lambda a, b: ndi.uniform_filter(a, size=size, mode=mode, cval=cval, output=b))

    def gaussian_filter(self, sigma, order=0, mode="reflect", cval=0.0, truncate=4.0, overlap=None):
        """gaussian_filter of the whole (distributed) volume; returns this rank's planes."""
        from .scipy import ndimage as ndi
        from .scipy.ndimage import _support as S
        from .scipy.ndimage.filters import _gaussian_weights
        key = ("gaussian", str(sigma), str(order), str(mode), float(cval), float(truncate))
        weights = None
        if key not in self._prepared:
            sigmas, orders = S.normalize_sequence(sigma, 3), S.normalize_sequence(order, 3)
            weights = [_gaussian_weights(sg, od, truncate) if sg > 1e-15 else None for sg, od in zip(sigmas, orders)]
        return self.separable(weights, mode, cval, overlap=overlap, _key=key,
                              fallback=lambda a, b: ndi.gaussian_filter(a, sigma, order=order, mode=mode, cval=cval,
                                                                        truncate=truncate, output=b))

    # ---------------------------------------------------------------- pipelined schedule
    def pipeline(self, weights, modes="reflect", cval=0.0, origins=(0, 0, 0), nbuf=2):
        """`SlabPipeline` of a separable filter over `nbuf` resident input slabs (the first is this filter's `ext_in`,
        the others are allocated here) writing `ext_out`.  Raises Unsupported when no fused / streaming float32 kernel
        takes the request (callers then use `step`)."""
        if self.ext_in.dtype != np.float32:
            raise _lib.Unsupported("the pipelined schedule exists for the float32 separable filters")
        nbuf = int(nbuf)
        if not 1 <= nbuf <= 4:
            raise ValueError("nbuf must be 1 .. 4")
        while len(self._extra_inputs) < nbuf - 1:          # kept: a second pipeline of this filter reuses them
            self._extra_inputs.append(core.empty(self.ext_in.shape, self.ext_in.dtype))
        ins = [self.ext_in] + self._extra_inputs[:nbuf - 1]
        return SlabPipeline(self.plan, self.comm, ins, self.ext_out, weights, modes, cval, origins)

    def uniform_pipeline(self, size, mode="reflect", cval=0.0, nbuf=2):
        """Pipelined `uniform_filter(size)` of a sequence of distributed volumes."""
        from .scipy.ndimage import _support as S
        sizes = [int(v) for v in S.normalize_sequence(size, 3)]
        weights = [np.full((sz,), 1.0 / sz) if sz > 1 else None for sz in sizes]
        return self.pipeline(weights, mode, cval, nbuf=nbuf)

    def gaussian_pipeline(self, sigma, order=0, mode="reflect", cval=0.0, truncate=4.0, nbuf=2):
        """Pipelined `gaussian_filter(sigma)` of a sequence of distributed volumes."""
        from .scipy.ndimage import _support as S
        from .scipy.ndimage.filters import _gaussian_weights
        sigmas, orders = S.normalize_sequence(sigma, 3), S.normalize_sequence(order, 3)
        weights = [_gaussian_weights(sg, od, truncate) if sg > 1e-15 else None for sg, od in zip(sigmas, orders)]
        return self.pipeline(weights, mode, cval, nbuf=nbuf)

    # ---------------------------------------------------------------- filters on the plain schedule
    # Everything below runs `step(fn)` by default: one exchange, then the package's own single-GPU kernel on the extended
    # slab (the halo planes are filtered too -- (lo + hi) / n_ext of wasted work, 3 % for config E -- and are scratch);
    # the min / max family can overlap the exchange (`overlap=True`) through plane-restricted launches.

    def _minmax(self, name, size, mode, cval, origin, overlap=False, footprint=None, structure=None):
        from .scipy import ndimage as ndi
        from .scipy.ndimage import _support as S
        nd = self.ext_in.ndim
        if footprint is not None or structure is not None:
            # window shape from the footprint / structure (filters.py:1373-1419, morphology.py:1091-1118); no overlap:
            # only the flat cubic kernels take plane ranges
            fp = None if footprint is None else S.as_host(footprint).astype(bool)
            st = None if structure is None else S.as_host(structure)
            shape = (fp if fp is not None else st).shape
            if len(shape) != nd:
                raise RuntimeError("footprint / structure and input must have the same dimensionality")
            sizes = [int(v) for v in shape]
            kw = {}
            if fp is not None:
                kw["footprint"] = fp
            if st is not None:
                kw["structure"] = st
            overlap = False
        else:
            sizes = [int(v) for v in S.normalize_sequence(size, nd)]
            kw = {"size": tuple(sizes)}
        origins = [int(v) for v in S.normalize_sequence(origin, nd)]
        # grey dilation mirrors the window: its reach along axis 0 is that of the flipped kernel
        o0 = origins[0] if name != "grey_dilation" else -origins[0] - (1 if sizes[0] % 2 == 0 else 0)
        self.check_reach(sizes[0], o0)
        fn = getattr(ndi, name)
        call = lambda a, b: fn(a, mode=mode, cval=cval, origin=tuple(origins), output=b, **kw)   # noqa: E731
        # overlap=True: interior planes while the exchange is in flight, the planes next to a neighbour afterwards
        # (mi_minmax3d_u8_planes / mi_minmax3d_f32_planes: uint8 cubic 3 / 5 / 7, float32 cubic 3 .. 9); what those
        # kernels do not take falls back to the plain schedule inside step_overlapped
        key = (name, tuple(sizes), str(mode), float(cval), tuple(origins))
        return self.step_overlapped(call, key) if overlap else self.step(call)

    def minimum_filter(self, size=None, footprint=None, mode="reflect", cval=0.0, origin=0, overlap=False):
        """minimum_filter(size=... | footprint=...) of the distributed volume; returns this rank's planes.  The halo
        comes from the window's extent along axis 0 and the origin (checked against the plan)."""
        return self._minmax("minimum_filter", size, mode, cval, origin, overlap, footprint=footprint)

    def maximum_filter(self, size=None, footprint=None, mode="reflect", cval=0.0, origin=0, overlap=False):
        return self._minmax("maximum_filter", size, mode, cval, origin, overlap, footprint=footprint)

    def grey_erosion(self, size=None, footprint=None, structure=None, mode="reflect", cval=0.0, origin=0, overlap=False):
        """grey_erosion (BASELINE config C's operation: size=7) of the distributed volume; flat (`size` / `footprint`)
        or with a `structure`."""
        return self._minmax("grey_erosion", size, mode, cval, origin, overlap, footprint=footprint, structure=structure)

    def grey_dilation(self, size=None, footprint=None, structure=None, mode="reflect", cval=0.0, origin=0, overlap=False):
        return self._minmax("grey_dilation", size, mode, cval, origin, overlap, footprint=footprint, structure=structure)

    def _dense(self, convolution, weights, mode, cval, origin):
        from .scipy import ndimage as ndi
        from .scipy.ndimage import _support as S
        w = S.as_host(weights)
        nd = self.ext_in.ndim
        if w.ndim != nd:
            raise RuntimeError("filter weights array has incorrect shape.")
        origins = [int(v) for v in S.normalize_sequence(origin, nd)]
        # axis-0 reach from the weights' extent + origin (offset rule _filters_core.py:10-11); a convolution is the
        # correlation with the flipped kernel, whose origin is -origin (one less for even lengths, filters.py:441-462)
        n0 = int(w.shape[0])
        o0 = origins[0] if not convolution else -origins[0] - (1 if n0 % 2 == 0 else 0)
        self.check_reach(n0, o0)
        fn = ndi.convolve if convolution else ndi.correlate
        return self.step(lambda a, b: fn(a, w, output=b, mode=mode, cval=cval, origin=tuple(origins)))

    def correlate(self, weights, mode="reflect", cval=0.0, origin=0):
        """Dense n-D `correlate` (filters.py:65-136) of the distributed volume: one exchange of the planes the weights'
        axis-0 extent reaches, then the single-GPU stencil on the extended slab; returns this rank's planes."""
        return self._dense(False, weights, mode, cval, origin)

    def convolve(self, weights, mode="reflect", cval=0.0, origin=0):
        """Dense n-D `convolve` (filters.py:139-210) of the distributed volume."""
        return self._dense(True, weights, mode, cval, origin)

    def _binary(self, name, structure, iterations, border_value, origin, any_changed):
        from .scipy import ndimage as ndi
        from .scipy.ndimage import _support as S
        fn = getattr(ndi, name)
        if structure is None:
            structure = ndi.generate_binary_structure(3, 1)
        st = S.as_host(structure).astype(bool)
        if st.ndim != 3:
            raise RuntimeError("structure and input must have same dimensionality")
        origins = [int(v) for v in S.normalize_sequence(origin, 3)]
        n0 = st.shape[0]
        # planes an output plane reaches below / above per iteration (dilation mirrors the structure)
        o0 = origins[0] if name == "binary_erosion" else -origins[0] - (1 if n0 % 2 == 0 else 0)
        lo1, hi1 = halo_widths(n0, o0)
        iterations = int(iterations)
        if iterations >= 1:
            # ONE exchange of iterations x reach planes: what the slab edges get wrong moves one reach inwards per
            # iteration and never arrives at a local plane
            plan = self.plan
            if (plan.prev >= 0 and iterations * lo1 > plan.lo) or (plan.next >= 0 and iterations * hi1 > plan.hi):
                raise ValueError("{} iterations of a structure reaching ({}, {}) planes need a halo of ({}, {}); the slab "
                                 "plan exchanges ({}, {})".format(iterations, lo1, hi1, iterations * lo1, iterations * hi1,
                                                                  plan.lo, plan.hi))
            return self.step(lambda a, b: fn(a, structure=st, iterations=iterations, border_value=border_value,
                                             origin=tuple(origins), output=b))
        # until nothing changes anywhere: one iteration per exchange, the "changed" flags of the ranks OR-ed on the host
        if self.plan.nranks > 1 and any_changed is None:
            raise ValueError("iterations < 1 on a distributed volume needs `any_changed` (a callable that ORs one bool "
                             "over the ranks, e.g. an all-reduce of the host flag)")
        self.check_reach(n0, o0)
        while True:
            self.step(lambda a, b: fn(a, structure=st, iterations=1, border_value=border_value, origin=tuple(origins),
                                      output=b))
            changed = bool(core.arrays_differ(self.local_out, self.local_in))
            if any_changed is not None:
                changed = bool(any_changed(changed))
            if not changed:
                return self.local_out
            self.local_in[...] = self.local_out

    def binary_erosion(self, structure=None, iterations=1, border_value=0, origin=0, any_changed=None):
        """binary_erosion of the distributed volume (reference loop: cupyimg/scipy/ndimage/morphology.py:292-322).
        `iterations >= 1`: ONE halo exchange of iterations x (structure reach) planes -- the plan must have been built
        with that halo -- then the iterated single-GPU kernel on the extended slab.  `iterations < 1` (until stable):
        one iteration per exchange; `any_changed(flag) -> bool` ORs the per-rank "changed" flag over the ranks (the
        only other communication, one int per iteration).  Returns this rank's planes."""
        return self._binary("binary_erosion", structure, iterations, border_value, origin, any_changed)

    def binary_dilation(self, structure=None, iterations=1, border_value=0, origin=0, any_changed=None):
        return self._binary("binary_dilation", structure, iterations, border_value, origin, any_changed)
