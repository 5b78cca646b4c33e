"""scipy.ndimage morphology on device arrays.

Same signatures and semantics as cupyimg/scipy/ndimage/morphology.py
(binary_erosion :334, binary_dilation :396, binary_opening :464,
binary_closing :540, binary_hit_or_miss :616, binary_propagation :684,
binary_fill_holes :726, grey_erosion :769, grey_dilation :818,
generate_binary_structure :174, iterate_structure :136).

Differences in mechanism, not in results: structuring elements stay on the
host (no device sync to inspect them, cf. morphology.py:133,274); iterated
erosion/dilation keeps the "did anything change" flag on the device and reads
back a single int32 per iteration instead of reducing a full-volume comparison
on the host (morphology.py:313,321).
"""
import ctypes
import operator

import numpy as np

from ... import _lib, core
from . import _support as S
from . import filters

__all__ = [
    "binary_erosion", "binary_dilation", "binary_opening", "binary_closing", "binary_hit_or_miss",
    "binary_propagation", "binary_fill_holes", "grey_erosion", "grey_dilation", "grey_opening", "grey_closing",
    "morphological_gradient", "morphological_laplace", "white_tophat", "black_tophat",
    "generate_binary_structure", "iterate_structure",
]


# stages of ONE opening / closing launch (twice the iteration count).  The library takes up to MI_BINARY_MAX_FUSED = 8, but a
# tile's halo grows with the stage count: six stages on 1024^3 run 1.50 ms in one launch against 0.95 ms as two launches of
# three fused iterations each (profiles/r6_binary.txt); two and four stages win (512^3 opening: 103 -> 58 us)
_MAX_FUSED_STAGES = 4
_MAX_FUSED_UNTIL_STABLE = 4     # runs until nothing changes: iterations per launch (eight were measured: the wider tile halo costs more than the launches saved)
_MAX_GROUP = 4                  # ... and the flags of up to four launches are read at once (host round trips, not launches, are what a batch costs)
_MAX_FUSED = 4      # iterations per launch of mi_binary_erosion_fused (the halo of a tile grows with it; <= MI_BINARY_MAX_FUSED)


def generate_binary_structure(rank, connectivity):
    """Structuring element with squared-distance connectivity
    (morphology.py:174-201); returned as a host bool array."""
    if connectivity < 1:
        connectivity = 1
    if rank < 1:
        return np.array(True, dtype=bool)
    dist = np.abs(np.indices([3] * rank) - 1).sum(axis=0)
    return dist <= connectivity


def iterate_structure(structure, iterations, origin=None):
    """Dilate a structure with itself ``iterations - 1`` times
    (morphology.py:136-171)."""
    structure = S.as_host(structure)
    if iterations < 2:
        return structure.copy()
    ni = iterations - 1
    shape = [ii + ni * (ii - 1) for ii in structure.shape]
    pos = [ni * (structure.shape[ii] // 2) for ii in range(len(shape))]
    slc = tuple(slice(pos[ii], pos[ii] + structure.shape[ii], None) for ii in range(len(shape)))
    out = np.zeros(shape, bool)
    out[slc] = structure != 0
    out = binary_dilation(out, structure, iterations=ni).get()
    if origin is None:
        return out
    origin = S.fix_sequence_arg(origin, structure.ndim, "origin", int)
    origin = [iterations * o for o in origin]
    return out, origin


_ROOTS = {}


def _minkowski_root(structure):
    """(small, r): `structure` (bool, cubic odd extent 2 r + 1 >= 5) equals r iterations of the 3 x 3 x 3 structure `small`
    (the cube or the 6-connected cross), else (None, 1).  Memoised on the structure's bytes."""
    shape = structure.shape
    if len(shape) != 3 or shape[0] != shape[1] or shape[1] != shape[2] or shape[0] < 5 or shape[0] % 2 == 0 or shape[0] > 31:
        return None, 1
    key = (shape, structure.tobytes())
    if key not in _ROOTS:
        if len(_ROOTS) > 64:
            _ROOTS.clear()
        r = shape[0] // 2
        found = (None, 1)
        if structure.all():
            found = (np.ones((3, 3, 3), bool), r)
        else:
            g = np.abs(np.indices(shape) - r).sum(axis=0)
            if np.array_equal(structure, g <= r):
                found = (generate_binary_structure(3, 1), r)
        _ROOTS[key] = found
    return _ROOTS[key]


def _binary_erosion(input, structure, iterations, mask, output, border_value, origin, invert,
                    brute_force=True):
    """morphology.py:204-331"""
    try:
        iterations = operator.index(iterations)
    except TypeError:
        raise TypeError("iterations parameter should be an integer")
    if isinstance(input, np.ndarray) and input.dtype.kind == "c":
        raise TypeError("Complex type not supported")
    input = S.as_device(input)
    if structure is None:
        structure = generate_binary_structure(input.ndim, 1)
    else:
        structure = S.as_host(structure).astype(bool)
    if structure.ndim != input.ndim:
        raise RuntimeError("structure and input must have same dimensionality")
    if structure.size < 1:
        raise RuntimeError("structure must not be empty")
    if mask is not None:
        mask = S.as_device(mask)
        if mask.shape != input.shape:
            raise RuntimeError("mask and input must have equal sizes")
        mask = core.ascontiguousarray(mask if mask.dtype == np.bool_ else mask.astype(np.bool_))
    origin = S.fix_sequence_arg(origin, input.ndim, "origin", int)

    if isinstance(output, core.ndarray):
        if output.dtype.kind == "c":
            raise TypeError("Complex output type not supported")
    else:
        output = bool
    output = S.get_output(output, input)
    if structure.ndim == 0:
        # 0-d special case (morphology.py:262-268)
        res = input.astype(np.bool_)
        if not bool(structure):
            res = _logical_not(res)
        output[...] = res
        return output
    for o, w in zip(origin, structure.shape):
        S.check_origin(o, w)
    if input.size == 0:
        return output

    # The reference raises NotImplementedError for multi-iteration calls with
    # brute_force=False when the structure centre is set (morphology.py:297-300).
    # SciPy returns the same result either way, so every iteration simply
    # re-evaluates all voxels here (the "brute force" schedule).

    # r6: a structure that is the r-fold Minkowski sum of a 3 x 3 x 3 one -- ones((2 r + 1,) * 3) = r x ones((3, 3, 3)), the
    # octahedron of radius r = r x the 6-connected cross (iterate_structure) -- runs as r times as many iterations of the
    # small one: those have straight-line code in the bit kernel and fuse (a 5 x 5 x 5 cube on an MNI-grid mask: 45 -> 22 us).
    # Exact with either border value: erosion by B + B is erosion by B twice on the zero- / one-extended volume, and every voxel
    # a dilation by B + B reaches is reached through an intermediate voxel inside the array's box (B convex, axis-symmetric).
    # Not with a mask (the mask then applies per iteration), not for runs until stable (same fixed point, but let the count be).
    if mask is None and iterations >= 1 and input.ndim == 3 and input.dtype.itemsize == 1 and not any(origin):
        small, r = _minkowski_root(structure)
        if small is not None and iterations * r <= 64:
            structure, iterations = small, iterations * r

    st = np.ascontiguousarray(structure, dtype=np.uint8)
    stp = st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    sshape = S.c_int64s(st.shape)
    org = S.c_ints(origin)
    mdesc = mask._desc() if mask is not None else None
    lib = S.lib()

    def launch(src, dst, flag_ptr=None):
        a, b = src._desc(), dst._desc()
        S.check(lib.mi_binary_erosion(ctypes.byref(a), ctypes.byref(b), stp, sshape, org,
                                      ctypes.byref(mdesc) if mdesc is not None else None,
                                      int(bool(border_value)), int(invert), flag_ptr, None))

    src = core.ascontiguousarray(input)
    direct = output._is_c_contiguous() and not core.shares_memory(output, src)
    final = output if direct else core.empty(output.shape, output.dtype)

    def launch_fused(src_, dst_, k, flag_ptr):
        a, b = src_._desc(), dst_._desc()
        rc = lib.mi_binary_erosion_fused(ctypes.byref(a), ctypes.byref(b), stp, sshape, org,
                                         ctypes.byref(mdesc) if mdesc is not None else None,
                                         int(bool(border_value)), int(invert), int(k), flag_ptr, None)
        if rc == _lib.MI_ERR_UNSUPPORTED:
            return False
        S.check(rc)
        return True

    if (iterations == 1 and src.dtype.itemsize == 1 and src.ndim == 3 and src.shape[-1] % 16 and final.dtype.itemsize == 1
            and launch_fused(src, final, 1, None)):
        pass                    # r6: rows of any length straight through the bit kernel (no extended copy)
    elif (iterations == 1 and mask is None and src.dtype.itemsize == 1 and src.ndim in (2, 3) and src.shape[-1] % 4
            and max(structure.shape) <= 9):                     # (the tiled kernel takes rows that are multiples of four bytes)
        # rows that are not a multiple of 16 bytes: the tiled kernel on rows extended by the border value (r4b, filters.py
        # _run_on_extended_rows; an out-of-range sample IS the border value, whichever way `invert` reads it)
        from .filters import _run_on_extended_rows
        left = structure.shape[-1] // 2 + int(origin[-1])
        if _run_on_extended_rows(src, final, left, structure.shape[-1] - 1 - left, "constant", int(bool(border_value)),
                                 lambda e, o: (launch(e, o), o)[1]) is None:
            launch(src, final)
    elif iterations == 1:
        launch(src, final)
    else:
        # brute-force ping-pong (morphology.py:301-327) with an on-device "changed" flag per iteration; buffers are
        # arranged so the last write hits `final` whenever the number of launches is known.  r6: up to _MAX_FUSED
        # iterations run inside ONE launch on a bit-packed tile (mi_binary_erosion_fused, csrc/bitmorph3d.hip) -- the
        # intermediate volumes never reach HBM; shapes / dtypes outside that kernel's envelope iterate one launch each.
        other = core.empty(final.shape, final.dtype)
        bufs = [final, other]
        cur = src
        if iterations >= 1:
            sizes = [_MAX_FUSED] * (iterations // _MAX_FUSED) + ([iterations % _MAX_FUSED] if iterations % _MAX_FUSED else [])
            which = 0 if len(sizes) & 1 else 1
            fused_ok = launch_fused(cur, bufs[which], sizes[0], None)
            if fused_ok:
                cur = bufs[which]
                for k in sizes[1:]:
                    which ^= 1
                    if launch_fused(cur, bufs[which], k, None):
                        cur = bufs[which]
                        continue
                    # a shorter tail may be refused where the full batch was taken (images: single iterations belong to the
                    # byte kernel): one launch per iteration between the two buffers; `cur` may end in either, the copy
                    # below puts it where it belongs
                    for _ in range(k):
                        launch(cur, bufs[which])
                        cur = bufs[which]
                        which ^= 1
                    which ^= 1
            else:
                which = 0 if iterations & 1 else 1
                for _ in range(iterations):
                    launch(cur, bufs[which])
                    cur = bufs[which]
                    which ^= 1
        else:
            # until nothing changes: one host read of the flags per batch of _MAX_FUSED iterations (the reference: one
            # full-volume comparison + synchronisation per iteration, morphology.py:313,321); the first iteration that
            # changes nothing ends the run, and the later iterations of its batch reproduce the same volume
            # r6: the flags are read once per GROUP of batches (1, 2, then 4 batches): a propagation through a 200-voxel volume
            # is 50 batches, and what a batch costs is the host round trip, not its 25-us launch; batches queued after the volume
            # became stable reproduce it (the same buffers ping-pong), so reading late changes nothing but the count.
            # Three ways to take a step, decided by the first launch of the run: (1) r6, masked dilations (binary_propagation,
            # binary_fill_holes): a block-wise FILL launch -- every workgroup sweeps a block of the volume in place until nothing
            # inside it changes, whole runs of mask bits along x per sweep (mi_binary_propagation_step: the operator is monotone,
            # any order of local updates reaches the same fixed point), so a launch carries information across a whole block
            # instead of one voxel; (2) a batch of fused iterations; (3) one iteration per launch.
            def launch_fill(src_, dst_, flag_ptr):
                a, b = src_._desc(), dst_._desc()
                rc = lib.mi_binary_propagation_step(ctypes.byref(a), ctypes.byref(b), stp, sshape, org, ctypes.byref(mdesc),
                                                    int(bool(border_value)), flag_ptr, None)
                if rc == _lib.MI_ERR_UNSUPPORTED:
                    return False
                S.check(rc)
                return True

            group = 1
            kb = _MAX_FUSED_UNTIL_STABLE
            flags = core.zeros((_MAX_GROUP, _MAX_FUSED_UNTIL_STABLE), np.int32)
            which = 1
            mode = None                                   # "fill" | "fused" | "single"
            while True:
                flags.fill(0)
                launched = 0
                for g in range(group if mode != "single" else 1):
                    dst = bufs[which]
                    fptr = ctypes.c_void_p(flags.ptr + 4 * _MAX_FUSED_UNTIL_STABLE * g)
                    if mode is None:
                        if invert and mdesc is not None and launch_fill(cur, dst, fptr):
                            mode = "fill"
                        elif launch_fused(cur, dst, kb, fptr):
                            mode = "fused"
                        else:
                            mode = "single"
                            launch(cur, dst, fptr)
                    elif mode == "fill":
                        if not launch_fill(cur, dst, fptr):
                            break                          # (cannot happen: same arrays, same answer)
                    elif mode == "fused":
                        if not launch_fused(cur, dst, kb, fptr):
                            break
                    else:
                        launch(cur, dst, fptr)
                    cur = dst
                    which ^= 1
                    launched += 1
                got = flags.get()
                if mode == "fused":
                    stable = not got[:launched, :kb].all()
                else:
                    stable = not got[:launched, 0].all()
                if stable:
                    break
                group = min(_MAX_GROUP, group * 2)
        if cur is not final:
            final[...] = cur
    if not direct:
        output[...] = final
    return output


def binary_erosion(input, structure=None, iterations=1, mask=None, output=None, border_value=0,
                   origin=0, brute_force=False):
    """Multidimensional binary erosion (morphology.py:334-393)."""
    return _binary_erosion(input, structure, iterations, mask, output, border_value, origin, 0,
                           brute_force)


def binary_dilation(input, structure=None, iterations=1, mask=None, output=None, border_value=0,
                    origin=0, brute_force=False):
    """Multidimensional binary dilation (morphology.py:396-461): erosion of
    the complement with the mirrored structure."""
    ndim = input.ndim if hasattr(input, "ndim") else np.ndim(input)
    if structure is None:
        structure = generate_binary_structure(ndim, 1)
    structure = S.as_host(structure)
    origin = S.fix_sequence_arg(origin, ndim, "origin", int)
    structure = structure[tuple([slice(None, None, -1)] * structure.ndim)]
    for ii in range(len(origin)):
        origin[ii] = -origin[ii]
        if ii < structure.ndim and not structure.shape[ii] & 1:
            origin[ii] -= 1
    return _binary_erosion(input, structure, iterations, mask, output, border_value, origin, 1,
                           brute_force)


def _open_close_fused(input, structure, iterations, output, origin, mask, border_value, closing):
    """binary_opening / binary_closing in ONE launch (mi_binary_open_close_fused, csrc/bitmorph3d.hip): both halves on a
    bit-packed tile, no temporary volume.  Returns the output array, or None when the request is outside the kernel's
    envelope (the caller then runs the two halves as the reference does, morphology.py:464-613)."""
    try:
        iterations = operator.index(iterations)
    except TypeError:
        return None                                   # the two-call path raises the reference's TypeError
    if (input.ndim not in (2, 3) or input.dtype.itemsize != 1 or iterations < 1 or 2 * iterations > _MAX_FUSED_STAGES
            or input.size == 0):
        return None
    st = S.as_host(structure).astype(bool)
    if st.ndim != input.ndim or st.size < 1 or any(int(n) % 2 == 0 for n in st.shape):
        return None
    if any(int(o) != 0 for o in S.fix_sequence_arg(origin, input.ndim, "origin", int)):
        return None
    if mask is not None:
        mask = S.as_device(mask)
        if mask.shape != input.shape:
            return None
        mask = core.ascontiguousarray(mask if mask.dtype == np.bool_ else mask.astype(np.bool_))
    if isinstance(output, core.ndarray):
        if output.dtype.kind == "c" or output.dtype.itemsize != 1:
            return None
        out = output
    else:
        out = core.empty(input.shape, np.bool_)
    src = core.ascontiguousarray(input)
    direct = out._is_c_contiguous() and not core.shares_memory(out, src)
    dst = out if direct else core.empty(out.shape, out.dtype)
    st8 = np.ascontiguousarray(st, dtype=np.uint8)
    a, b = src._desc(), dst._desc()
    mdesc = mask._desc() if mask is not None else None
    rc = S.lib().mi_binary_open_close_fused(ctypes.byref(a), ctypes.byref(b), st8.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                            S.c_int64s(st8.shape), ctypes.byref(mdesc) if mdesc is not None else None,
                                            int(bool(border_value)), int(closing), iterations, None)
    if rc == _lib.MI_ERR_UNSUPPORTED:
        return None
    S.check(rc)
    if not direct:
        out[...] = dst
    return out


def binary_opening(input, structure=None, iterations=1, output=None, origin=0, mask=None,
                   border_value=0, brute_force=False):
    """Erosion followed by dilation (morphology.py:464-537)."""
    input = S.as_device(input)
    if structure is None:
        structure = generate_binary_structure(input.ndim, 1)
    res = _open_close_fused(input, structure, iterations, output, origin, mask, border_value, False)
    if res is not None:
        return res
    tmp = binary_erosion(input, structure, iterations, mask, None, border_value, origin, brute_force)
    return binary_dilation(tmp, structure, iterations, mask, output, border_value, origin, brute_force)


def binary_closing(input, structure=None, iterations=1, output=None, origin=0, mask=None,
                   border_value=0, brute_force=False):
    """Dilation followed by erosion (morphology.py:540-613)."""
    input = S.as_device(input)
    if structure is None:
        structure = generate_binary_structure(input.ndim, 1)
    res = _open_close_fused(input, structure, iterations, output, origin, mask, border_value, True)
    if res is not None:
        return res
    tmp = binary_dilation(input, structure, iterations, mask, None, border_value, origin, brute_force)
    return binary_erosion(tmp, structure, iterations, mask, output, border_value, origin, brute_force)


def _as_bool(a):
    """a != 0 as a bool device array"""
    return a if a.dtype == np.bool_ else a.astype(np.bool_)


def _logical_not(a):
    """~a for a bool device array (xor with ones, mi_elementwise)"""
    out = core.empty(a.shape, np.bool_)
    return S.elementwise("subtract", a, core.ones(a.shape, np.bool_), out)


def _logical_and(a, b):
    out = core.empty(a.shape, np.bool_)
    return S.elementwise("multiply", a, b, out)


def binary_hit_or_miss(input, structure1=None, structure2=None, output=None, origin1=0, origin2=None):
    """Hit-or-miss transform (morphology.py:616-681)."""
    input = S.as_device(input)
    if structure1 is None:
        structure1 = generate_binary_structure(input.ndim, 1)
    structure1 = S.as_host(structure1)
    if structure2 is None:
        structure2 = np.logical_not(structure1)
    origin1 = S.fix_sequence_arg(origin1, input.ndim, "origin1", int)
    if origin2 is None:
        origin2 = origin1
    else:
        origin2 = S.fix_sequence_arg(origin2, input.ndim, "origin2", int)
    tmp1 = _binary_erosion(input, structure1, 1, None, None, 0, origin1, 0, False)
    result = _binary_erosion(input, structure2, 1, None, None, 0, origin2, 1, False)
    res = _logical_and(_as_bool(tmp1), _logical_not(_as_bool(result)))
    if isinstance(output, core.ndarray):
        output[...] = res
        return None
    return res


def binary_propagation(input, structure=None, mask=None, output=None, border_value=0, origin=0):
    """Dilation until stable inside ``mask`` (morphology.py:684-723)."""
    return binary_dilation(input, structure, -1, mask, output, border_value, origin, brute_force=True)


def binary_fill_holes(input, structure=None, output=None, origin=0):
    """Fill holes in binary objects (morphology.py:726-766)."""
    input = S.as_device(input)
    mask = _logical_not(_as_bool(input))
    tmp = core.zeros(mask.shape, np.bool_)
    res = binary_dilation(tmp, structure, -1, mask, None, 1, origin, brute_force=True)
    res = _logical_not(_as_bool(res))
    if isinstance(output, core.ndarray):
        output[...] = res
        return None
    return res


def grey_erosion(input, size=None, footprint=None, structure=None, output=None, mode="reflect",
                 cval=0.0, origin=0):
    """Greyscale erosion = minimum filter (morphology.py:769-815)."""
    if size is None and footprint is None and structure is None:
        raise ValueError("size, footprint or structure must be specified")
    return filters._min_or_max_filter(input, size, footprint, structure, output, mode, cval, origin, "min")


def grey_dilation(input, size=None, footprint=None, structure=None, output=None, mode="reflect",
                  cval=0.0, origin=0):
    """Greyscale dilation = maximum filter with footprint / structure mirrored
    on every axis and the origin negated, minus one for even extents
    (morphology.py:818-884)."""
    if size is None and footprint is None and structure is None:
        raise ValueError("size, footprint or structure must be specified")
    ndim = input.ndim if hasattr(input, "ndim") else np.ndim(input)
    mirror = lambda a: a[tuple([slice(None, None, -1)] * a.ndim)]
    if structure is not None:
        structure = mirror(S.as_host(structure))
    if footprint is not None:
        footprint = mirror(S.as_host(footprint))
    origin = S.fix_sequence_arg(origin, ndim, "origin", int)
    for i in range(len(origin)):
        origin[i] = -origin[i]
        if footprint is not None:
            sz = footprint.shape[i]
        elif structure is not None:
            sz = structure.shape[i]
        elif np.isscalar(size):
            sz = size
        else:
            sz = size[i]
        if sz % 2 == 0:
            origin[i] -= 1
    return filters._min_or_max_filter(input, size, footprint, structure, output, mode, cval, origin, "max")


def grey_opening(input, size=None, footprint=None, structure=None, output=None, mode="reflect", cval=0.0, origin=0):
    """Greyscale opening: erosion, then dilation (morphology.py:887-935)."""
    tmp = grey_erosion(input, size, footprint, structure, None, mode, cval, origin)
    return grey_dilation(tmp, size, footprint, structure, output, mode, cval, origin)


def grey_closing(input, size=None, footprint=None, structure=None, output=None, mode="reflect", cval=0.0, origin=0):
    """Greyscale closing: dilation, then erosion (morphology.py:938-986)."""
    tmp = grey_dilation(input, size, footprint, structure, None, mode, cval, origin)
    return grey_erosion(tmp, size, footprint, structure, output, mode, cval, origin)


def morphological_gradient(input, size=None, footprint=None, structure=None, output=None, mode="reflect",
                           cval=0.0, origin=0):
    """dilation - erosion (morphology.py:989-1047)."""
    input = S.as_device(input)
    tmp = grey_dilation(input, size, footprint, structure, None, mode, cval, origin)
    ero = grey_erosion(input, size, footprint, structure, output if isinstance(output, core.ndarray) else None,
                       mode, cval, origin)
    return S.elementwise("subtract", tmp, ero, ero)


def morphological_laplace(input, size=None, footprint=None, structure=None, output=None, mode="reflect",
                          cval=0.0, origin=0):
    """dilation + erosion - 2 input (morphology.py:1050-1105)."""
    input = S.as_device(input)
    tmp1 = grey_dilation(input, size, footprint, structure, None, mode, cval, origin)
    tmp2 = grey_erosion(input, size, footprint, structure, output if isinstance(output, core.ndarray) else None,
                        mode, cval, origin)
    S.elementwise("add", tmp1, tmp2, tmp2)
    S.elementwise("subtract", tmp2, input, tmp2)
    return S.elementwise("subtract", tmp2, input, tmp2)


def white_tophat(input, size=None, footprint=None, structure=None, output=None, mode="reflect", cval=0.0, origin=0):
    """input - opening (xor for bool images) (morphology.py:1108-1166)."""
    input = S.as_device(input)
    tmp = grey_erosion(input, size, footprint, structure, None, mode, cval, origin)
    tmp = grey_dilation(tmp, size, footprint, structure, output, mode, cval, origin)
    return S.elementwise("subtract", input, tmp, tmp)       # bool - bool is xor


def black_tophat(input, size=None, footprint=None, structure=None, output=None, mode="reflect", cval=0.0, origin=0):
    """closing - input (xor for bool images) (morphology.py:1169-1226)."""
    input = S.as_device(input)
    tmp = grey_dilation(input, size, footprint, structure, None, mode, cval, origin)
    tmp = grey_erosion(tmp, size, footprint, structure, output, mode, cval, origin)
    return S.elementwise("subtract", tmp, input, tmp)
