"""Minimal device ndarray on top of the libmi355img runtime.

The reference takes ``cupy.ndarray`` everywhere (README.md:50-58); this is the
counterpart owned by the engine itself: a pointer into pooled HBM plus
shape / byte strides / dtype.  It supports what the filtering path needs
(creation, host transfer, views by basic slicing and transposition, dtype
casts, ``out[...] = in``) and deliberately nothing more -- it is plumbing, not
an array library.  No CuPy, no PyTorch.
"""
import ctypes
import threading

import numpy as np

from . import _lib
from ._lib import MiArray

_DTYPE_CODES = {
    np.dtype(np.bool_): 0, np.dtype(np.int8): 1, np.dtype(np.uint8): 2,
    np.dtype(np.int16): 3, np.dtype(np.uint16): 4, np.dtype(np.int32): 5,
    np.dtype(np.uint32): 6, np.dtype(np.int64): 7, np.dtype(np.uint64): 8,
    np.dtype(np.float32): 9, np.dtype(np.float64): 10,
    np.dtype(np.float16): 11,      # storage only (mi_copy converts); see scipy.ndimage._support.float16_aware
}


def dtype_code(dtype):
    dtype = np.dtype(dtype)
    try:
        return _DTYPE_CODES[dtype]
    except KeyError:
        raise TypeError("dtype {} is not supported by cupyimg_amd".format(dtype))


class _Memory:
    """Owner of one pooled device allocation."""

    __slots__ = ("ptr", "nbytes", "device", "__weakref__")

    def __init__(self, nbytes):
        lib = _lib.load()
        p = ctypes.c_void_p()
        dev = ctypes.c_int()
        _lib.check(lib.mi_get_device(ctypes.byref(dev)))
        _lib.check(lib.mi_malloc(ctypes.byref(p), max(int(nbytes), 1)))
        self.ptr = p.value
        self.nbytes = int(nbytes)
        self.device = dev.value

    def __del__(self):
        try:
            if self.ptr:
                _lib.load().mi_free(self.ptr)
        except Exception:  # interpreter shutdown
            pass
        self.ptr = None


class _ForeignMemory:
    """Memory owned by someone else (e.g. a torch tensor); kept alive via `owner`."""

    __slots__ = ("ptr", "nbytes", "device", "owner")

    def __init__(self, ptr, nbytes, owner, device=0):
        self.ptr = int(ptr)
        self.nbytes = int(nbytes)
        self.owner = owner
        self.device = device


class _Flags:
    __slots__ = ("c_contiguous",)

    def __init__(self, c):
        self.c_contiguous = c

    def __repr__(self):
        return "  C_CONTIGUOUS : {}".format(self.c_contiguous)


def _c_strides(shape, itemsize):
    strides = []
    st = itemsize
    for s in reversed(shape):
        strides.append(st)
        st *= max(int(s), 1)
    return tuple(reversed(strides))


class ndarray:
    """Device array: pointer + shape + byte strides + dtype."""

    __slots__ = ("_mem", "ptr", "shape", "strides", "dtype", "base", "_dc", "_v3", "_hc")

    def __init__(self, shape, dtype=np.float32, _mem=None, _ptr=None, _strides=None, _base=None):
        if np.isscalar(shape):
            shape = (int(shape),)
        self.shape = tuple(int(s) for s in shape)
        if len(self.shape) > _lib.MI_MAX_NDIM:
            raise ValueError("at most {} dimensions are supported".format(_lib.MI_MAX_NDIM))
        self.dtype = np.dtype(dtype)
        dtype_code(self.dtype)
        if _mem is None:
            _mem = _Memory(self.size * self.dtype.itemsize)
            _ptr = _mem.ptr
        self._mem = _mem
        self.ptr = int(_ptr)
        self.strides = tuple(_strides) if _strides is not None else _c_strides(self.shape, self.dtype.itemsize)
        self.base = _base
        self._dc = None         # cached C-ABI descriptor: (shape, strides, ptr, MiArray)
        self._v3 = None         # cached one-plane-volume view of an image
        self._hc = None         # host copy of a small constant array (structuring elements), see host_hint()

    # ------------------------------------------------------------- properties
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def itemsize(self):
        return self.dtype.itemsize

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def device(self):
        return self._mem.device

    @property
    def flags(self):
        return _Flags(self._is_c_contiguous())

    @property
    def real(self):
        return self

    @property
    def T(self):
        return self.transpose()

    def _is_c_contiguous(self):
        expect = self.dtype.itemsize
        for s, st in zip(reversed(self.shape), reversed(self.strides)):
            if s == 0:
                return True
            if s != 1 and st != expect:
                return False
            expect *= s
        return True

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def __repr__(self):
        return "cupyimg_amd.ndarray(shape={}, dtype={}, device={})".format(self.shape, self.dtype, self.device)

    def __array__(self, *args, **kwargs):
        raise TypeError("implicit conversion to a NumPy array is not allowed; use .get()")

    @property
    def __cuda_array_interface__(self):
        """Version 3.  By default the library's default stream is synchronised
        before the pointer is handed out and `stream` is None ("no
        synchronisation required"): torch.as_tensor ignores the `stream` entry
        (measured: it read the array before the filter had run), so naming the
        stream alone is not safe.  With ``core.EXPORT_SYNC = False`` the entry
        names the library's default stream instead and a protocol-abiding
        consumer (CuPy) orders its own stream behind it without a host sync."""
        if EXPORT_SYNC:
            _lib.check(_lib.load().mi_stream_sync(None))
        return {
            "shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False),
            "strides": None if self._is_c_contiguous() else self.strides, "version": 3,
            "stream": None if EXPORT_SYNC else default_stream_handle(),
        }

    # ------------------------------------------------------------- C-ABI view
    def _desc(self):
        # the descriptor is read-only for the library, so one per array object is enough (a filter call on a small
        # image is bound by this Python layer: building two descriptors cost 3 of its ~20 us)
        c = self._dc
        if c is not None and c[0] is self.shape and c[1] is self.strides and c[2] == self.ptr:
            return c[3]
        d = MiArray()
        d.data = self.ptr
        d.dtype = dtype_code(self.dtype)
        d.ndim = self.ndim
        for i, (s, st) in enumerate(zip(self.shape, self.strides)):
            d.shape[i] = s
            d.strides[i] = st
        self._dc = (self.shape, self.strides, self.ptr, d)
        return d

    def _as3(self):
        """An image as a one-plane volume (what the fused 3-D kernels take); cached."""
        v = self._v3
        if v is None or v.ptr != self.ptr or v.shape[1:] != self.shape:
            # no strong reference back to `self` (array -> _v3 -> base -> array would keep the device buffer alive
            # until the cyclic collector runs): the view shares `_mem`, which is what owns the allocation
            v = ndarray((1,) + self.shape, self.dtype, _mem=self._mem, _ptr=self.ptr,
                        _strides=(self.strides[0] * self.shape[0],) + self.strides, _base=self.base)
            self._v3 = v
        return v

    # ------------------------------------------------------------- transfers
    def get(self):
        """Copy to a new NumPy array (synchronises the default stream)."""
        src = self if self._is_c_contiguous() else self.copy()
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            _lib.check(_lib.load().mi_memcpy_d2h(out.ctypes.data, src.ptr, out.nbytes, None))
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        if arr.shape != self.shape:
            raise ValueError("shape mismatch")
        self._touch()
        if self._is_c_contiguous():
            if arr.nbytes:
                lib = _lib.load()
                _lib.check(lib.mi_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes, None))
                # the source is pageable host memory: wait so it may be reused
                _lib.check(lib.mi_stream_sync(None))
        else:
            self[...] = asarray(arr)

    # ------------------------------------------------------------- copies / casts
    def copy(self):
        out = ndarray(self.shape, self.dtype)
        _copy(self, out)
        return out

    def astype(self, dtype, copy=True):
        dtype = np.dtype(dtype)
        if dtype == self.dtype and not copy:
            return self
        out = ndarray(self.shape, dtype)
        _copy(self, out)
        return out

    def _touch(self):
        """The array is about to be written: drop the remembered host copy (with_host_hint) of this array and of the
        array it is a view of."""
        self._hc = None
        b = self.base
        if b is not None:
            b._hc = None

    def fill(self, value):
        self._touch()
        d = self._desc()
        _lib.check(_lib.load().mi_fill(ctypes.byref(d), float(value), None))

    # ------------------------------------------------------------- views
    def _view(self, shape, strides, ptr):
        return ndarray(shape, self.dtype, _mem=self._mem, _ptr=ptr, _strides=strides,
                       _base=self if self.base is None else self.base)

    def transpose(self, *axes):
        if not axes or axes == (None,):
            axes = tuple(reversed(range(self.ndim)))
        elif len(axes) == 1 and hasattr(axes[0], "__iter__"):
            axes = tuple(axes[0])
        if sorted(a % self.ndim for a in axes) != list(range(self.ndim)):
            raise ValueError("axes don't match array")
        axes = [a % self.ndim for a in axes]
        return self._view([self.shape[a] for a in axes], [self.strides[a] for a in axes], self.ptr)

    def reshape(self, *shape):
        if len(shape) == 1 and hasattr(shape[0], "__iter__"):
            shape = tuple(shape[0])
        shape = list(shape)
        if shape.count(-1) > 1:
            raise ValueError("can only specify one unknown dimension")
        if -1 in shape:
            known = 1
            for s in shape:
                if s != -1:
                    known *= s
            shape[shape.index(-1)] = self.size // known if known else 0
        n = 1
        for s in shape:
            n *= s
        if n != self.size:
            raise ValueError("cannot reshape array of size {} into shape {}".format(self.size, tuple(shape)))
        src = self if self._is_c_contiguous() else self.copy()
        return src._view(shape, _c_strides(shape, self.dtype.itemsize), src.ptr)

    def ravel(self):
        return self.reshape(-1)

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = [k is Ellipsis for k in key].index(True)
            nfill = self.ndim - sum(1 for k in key if k is not None and k is not Ellipsis)
            key = key[:i] + (slice(None),) * nfill + key[i + 1:]
        shape, strides, ptr, dim = [], [], self.ptr, 0
        for k in key:
            if k is None:
                shape.append(1)
                strides.append(0)
                continue
            if dim >= self.ndim:
                raise IndexError("too many indices for array")
            n, st = self.shape[dim], self.strides[dim]
            if isinstance(k, slice):
                start, stop, step = k.indices(n)
                length = len(range(start, stop, step))
                ptr += start * st
                shape.append(length)
                strides.append(st * step)
            else:
                k = int(k)
                if k < -n or k >= n:
                    raise IndexError("index out of bounds")
                ptr += (k % n) * st
            dim += 1
        shape += list(self.shape[dim:])
        strides += list(self.strides[dim:])
        return self._view(shape, strides, ptr)

    def __setitem__(self, key, value):
        self._touch()
        dst = self[key]
        if isinstance(value, ndarray):
            src = value
        elif np.isscalar(value) or (isinstance(value, np.ndarray) and value.ndim == 0):
            dst.fill(value)
            return
        else:
            src = asarray(np.asarray(value))
        if src.shape != dst.shape:
            # allow leading unit-axis broadcasting of equal sizes only
            if src.size == dst.size:
                src = src.reshape(dst.shape)
            else:
                raise ValueError("could not broadcast input array from shape {} into shape {}".format(
                    src.shape, dst.shape))
        _copy(src, dst)


def _copy(src, dst, round_half_even=False):
    dst._touch()
    a, b = src._desc(), dst._desc()
    _lib.check(_lib.load().mi_copy(ctypes.byref(a), ctypes.byref(b), int(round_half_even), None))


# --------------------------------------------------------------------- creation
def empty(shape, dtype=np.float32):
    return ndarray(shape, dtype)


def zeros(shape, dtype=np.float32):
    a = ndarray(shape, dtype)
    if a.nbytes:
        _lib.check(_lib.load().mi_memset(a.ptr, 0, a.nbytes, None))
    return a


def full(shape, value, dtype=None):
    if dtype is None:
        dtype = np.asarray(value).dtype
    a = ndarray(shape, dtype)
    a.fill(value)
    return a


def ones(shape, dtype=np.float32):
    return full(shape, 1, dtype)


def empty_like(a, dtype=None):
    return ndarray(a.shape, a.dtype if dtype is None else dtype)


def zeros_like(a, dtype=None):
    return zeros(a.shape, a.dtype if dtype is None else dtype)


def asarray(obj, dtype=None):
    """Device array from a device array (no copy), a NumPy array / sequence
    (host to device copy) or anything exposing ``__cuda_array_interface__``
    (zero copy, e.g. a torch ROCm tensor)."""
    if isinstance(obj, ndarray):
        if dtype is not None and np.dtype(dtype) != obj.dtype:
            return obj.astype(dtype)
        return obj
    if not isinstance(obj, np.ndarray) and hasattr(obj, "__cuda_array_interface__"):
        return from_cuda_array_interface(obj)
    arr = np.asarray(obj, dtype=dtype)
    if arr.dtype.kind in "cOSU":
        raise TypeError("dtype {} is not supported by cupyimg_amd".format(arr.dtype))
    arr = np.ascontiguousarray(arr)
    out = ndarray(arr.shape, arr.dtype)
    out.set(arr)
    return out


array = asarray


EXPORT_SYNC = True      # see ndarray.__cuda_array_interface__


def default_stream_handle():
    """The library's default stream (a non-blocking hipStream_t) as an integer."""
    h = ctypes.c_void_p()
    _lib.check(_lib.load().mi_default_stream(ctypes.byref(h)))
    return int(h.value or 0)


def wait_for_stream(producer):
    """Work queued on the library's default stream from now on waits for what is
    queued on `producer` (a foreign stream handle as an integer, or 1 / 2 for the
    legacy / per-thread default stream)."""
    _lib.check(_lib.load().mi_stream_wait_stream(None, ctypes.c_void_p(int(producer))))


def stream_waits_for_us(consumer):
    """The foreign stream `consumer` (integer handle, or 1 / 2) waits for the work
    queued on the library's default stream so far -- call it before another
    runtime reads an array this library wrote in place (a torch tensor passed as
    `output=`); arrays exported through __cuda_array_interface__ carry the
    stream themselves."""
    _lib.check(_lib.load().mi_stream_wait_stream(ctypes.c_void_p(int(consumer)), None))


def from_cuda_array_interface(obj):
    """Zero-copy view of a foreign device array.  Stream ordering: when the
    producer names a stream (protocol version 3) the library's default stream
    waits for the work queued there.  Without one -- torch exports version 2,
    which has no `stream` entry -- it waits for the legacy default stream,
    which is where torch's default stream runs; a producer working on a
    non-blocking side stream has to name it or synchronise itself."""
    cai = obj.__cuda_array_interface__
    producer = cai.get("stream")
    if producer is not None and int(producer) == 0:
        raise ValueError("__cuda_array_interface__: stream 0 is not allowed by the protocol")
    wait_for_stream(1 if producer is None else int(producer))
    dtype = np.dtype(cai["typestr"])
    shape = tuple(cai["shape"])
    strides = cai.get("strides") or _c_strides(shape, dtype.itemsize)
    ptr = cai["data"][0]
    n = 1
    for s in shape:
        n *= s
    mem = _ForeignMemory(ptr, n * dtype.itemsize, obj)
    return ndarray(shape, dtype, _mem=mem, _ptr=ptr, _strides=strides)


def asnumpy(a):
    if isinstance(a, ndarray):
        return a.get()
    return np.asarray(a)


def ascontiguousarray(a, dtype=None):
    a = asarray(a)
    if dtype is not None and np.dtype(dtype) != a.dtype:
        return a.astype(dtype)
    return a if a._is_c_contiguous() else a.copy()


def with_host_hint(a, host):
    """Remember the host array a small device array was uploaded from (structuring elements: the morphology calls
    need them on the host again, and fetching one back costs a stream synchronisation plus a copy, ~50 us per call).
    Every write path of this package drops the hint -- `__setitem__`, `fill`, `set`, copies into the array, and its
    use as an `output=` (scipy.ndimage._support.get_output), also through a view (`ndarray._touch`); only the
    footprint constructors of skimage.morphology set it."""
    a._hc = np.array(host, copy=True)
    a._hc.setflags(write=False)
    return a


def host_copy(a):
    """NumPy copy of a device array: the remembered upload source if there is one, else a device-to-host copy."""
    hc = a._hc
    if hc is not None and hc.shape == a.shape and hc.dtype == a.dtype:
        return hc.copy()
    return a.get()


def _bounds(a):
    lo = hi = a.ptr
    for s, st in zip(a.shape, a.strides):
        if s == 0:
            return a.ptr, a.ptr
        if st >= 0:
            hi += (s - 1) * st
        else:
            lo += (s - 1) * st
    return lo, hi + a.dtype.itemsize


def shares_memory(a, b):
    """MAY_SHARE_BOUNDS test, as the reference uses it (_filters_core.py:148)."""
    if not isinstance(a, ndarray) or not isinstance(b, ndarray):
        return False
    if a.size == 0 or b.size == 0:
        return False
    alo, ahi = _bounds(a)
    blo, bhi = _bounds(b)
    return alo < bhi and blo < ahi


# --------------------------------------------------------------------- runtime
def synchronize():
    _lib.check(_lib.load().mi_stream_sync(None))


def device_count():
    n = ctypes.c_int(0)
    rc = _lib.load().mi_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def is_available():
    try:
        return device_count() > 0
    except Exception:
        return False


def set_device(dev):
    _lib.check(_lib.load().mi_set_device(int(dev)))


def get_device():
    d = ctypes.c_int()
    _lib.check(_lib.load().mi_get_device(ctypes.byref(d)))
    return d.value


def device_name(dev=None):
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.load().mi_device_name(get_device() if dev is None else dev, buf, 256))
    return buf.value.decode()


def pool_stats():
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.load().mi_pool_stats(ctypes.byref(a), ctypes.byref(b)))
    return {"in_use": a.value, "cached": b.value}


def free_all_blocks():
    _lib.check(_lib.load().mi_pool_trim())


def arrays_differ(a, b):
    """True when two device arrays of equal dtype / shape differ anywhere
    (device-side comparison, one int32 read back)."""
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("arrays_differ needs equal shapes and dtypes")
    a, b = ascontiguousarray(a), ascontiguousarray(b)
    flag = zeros((1,), np.int32)
    da, db = a._desc(), b._desc()
    _lib.check(_lib.load().mi_any_diff(ctypes.byref(da), ctypes.byref(db), ctypes.c_void_p(flag.ptr), None))
    return bool(flag.get()[0])


class Stream:
    """A hipStream besides the library's default stream (used for the halo
    exchange so that it overlaps the interior filtering)."""

    def __init__(self):
        self._s = ctypes.c_void_p()
        _lib.check(_lib.load().mi_stream_create(ctypes.byref(self._s)))

    @property
    def handle(self):
        return self._s

    def wait_event(self, event):
        """Work submitted to this stream from now on waits for `event`."""
        _lib.check(_lib.load().mi_stream_wait_event(self._s, event._e))

    def synchronize(self):
        _lib.check(_lib.load().mi_stream_sync(self._s))

    def __del__(self):
        try:
            if self._s:
                _lib.load().mi_stream_destroy(self._s)
        except Exception:
            pass


def default_stream_wait_event(event):
    """Work submitted to the default stream from now on waits for `event`."""
    _lib.check(_lib.load().mi_stream_wait_event(None, event._e))


class Event:
    """hipEvent; recorded on the library's default stream unless a Stream is given."""

    def __init__(self):
        self._e = ctypes.c_void_p()
        _lib.check(_lib.load().mi_event_create(ctypes.byref(self._e)))

    def record(self, stream=None):
        _lib.check(_lib.load().mi_event_record(self._e, None if stream is None else stream.handle))

    def synchronize(self):
        _lib.check(_lib.load().mi_event_sync(self._e))

    def elapsed_ms(self, later):
        ms = ctypes.c_float()
        _lib.check(_lib.load().mi_event_elapsed_ms(self._e, later._e, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self._e:
                _lib.load().mi_event_destroy(self._e)
        except Exception:
            pass
