"""CPU restatement of the counter-based synthetic volume generator (cupyimg_amd/csrc/synth.hip,
mi_debug_fill_synthetic_f32) -- test / bench infrastructure like the rest of oracle/: bench.py --config E uses it to
rebuild the planes next to a slab seam on the host without ever holding the 32 GiB volume.

    x(i) = ((u0 + u1 + u2 + u3) - 2) * sqrt(3),  u_k = (splitmix64(seed + 4 i + k) >> 42) * 2^-22

bit-identical to the device kernel (exact float32 sums, one correctly rounded product)."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = z + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synthetic_f32(first_index, n, seed=0):
    """The n values of global linear indices first_index .. first_index + n - 1."""
    with np.errstate(over="ignore"):
        c = np.uint64(seed) + np.uint64(4) * (np.uint64(first_index) + np.arange(n, dtype=np.uint64))
        s = np.zeros(n, np.float32)
        for k in range(4):
            u = (_splitmix64(c + np.uint64(k)) >> np.uint64(42)).astype(np.uint32).astype(np.float32)
            s += u * np.float32(2.0 ** -22)
    return (s - np.float32(2.0)) * np.float32(1.7320508075688772)


def synthetic_f32_c(first_index, n, seed=0):
    """The same through liboracle.so (orc_synth_f32): ~100x faster than the NumPy form for the seam checks."""
    import ctypes
    from . import ndimage as orc
    out = np.empty(int(n), np.float32)
    fn = orc.lib().orc_synth_f32
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64]
    fn(out.ctypes.data, int(n), int(first_index), int(seed))
    return out


def synthetic_planes(z0, z1, plane_shape, seed=0):
    """Planes z0 .. z1-1 of a C-contiguous volume whose planes have `plane_shape`."""
    per = int(np.prod(plane_shape))
    return synthetic_f32_c(z0 * per, (z1 - z0) * per, seed).reshape((z1 - z0,) + tuple(plane_shape))
