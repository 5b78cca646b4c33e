/*
 * ndimage_oracle.c -- CPU restatement of the n-D filtering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (cupyimg_amd/)
 * may import, link or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed
 * CPU baseline.
 *
 * What it restates (reference = /root/reference/cupyimg/scipy/ndimage, whose
 * own tests define correctness as agreement with SciPy's C implementation
 * scipy.ndimage._nd_image, pinned here at SciPy 1.15.3):
 *   - boundary index maps          _util.py:170-228   (filters + interpolation)
 *   - offset rule w//2 + origin    _filters_core.py:10-11, _util.py:231-239
 *   - correlate (1-D and n-D)      _filters_core.py:190-324, filters.py:441-511
 *   - min / max with footprint and optional non-flat structure
 *                                  filters.py:1373-1557
 *   - binary erosion / dilation    morphology.py:41-128, 204-331
 *   - order 0 / 1 interpolation    _interp_kernels.py:277-592
 *
 * Parity pin: every function is checked in tests/test_oracle_*.py against
 * (a) the literal known-answer vectors of the reference's own tests
 *     (tests/golden/kat_reference.json) and
 * (b) fixtures generated from SciPy 1.15.3 (the .npz files under tests/golden).
 *
 * All value arrays are C-contiguous doubles: SciPy itself converts every line
 * to double before doing arithmetic (its line buffers), so doing the same is
 * exact, not an approximation.  Casting to the user-visible dtype happens in
 * orc_cast_from_f64 with C truncation semantics.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXDIM 8

enum {
    ORC_REFLECT = 0,       /* d c b a | a b c d | d c b a   (grid-mirror) */
    ORC_CONSTANT = 1,      /* k k k k | a b c d | k k k k                 */
    ORC_NEAREST = 2,       /* a a a a | a b c d | d d d d                 */
    ORC_MIRROR = 3,        /* d c b   | a b c d | c b a                   */
    ORC_WRAP = 4,          /* interpolation only: period n-1              */
    ORC_GRID_WRAP = 5,     /* a b c d | a b c d | a b c d                 */
    ORC_GRID_CONSTANT = 6  /* interpolation: blended with cval            */
};

/* dtype codes shared with include/mi355img.h */
enum {
    ORC_BOOL = 0, ORC_I8, ORC_U8, ORC_I16, ORC_U16, ORC_I32, ORC_U32,
    ORC_I64, ORC_U64, ORC_F32, ORC_F64
};

/* ------------------------------------------------------------------ */
/* boundary maps                                                       */
/* ------------------------------------------------------------------ */

/* Integer index map; returns -1 when the constant value must be used.
 * Follows _util.py:170-228 including C truncated '%'. */
static int64_t bmap(int64_t i, int64_t n, int mode)
{
    switch (mode) {
    case ORC_REFLECT:
        if (i < 0) i = -1 - i;
        i %= 2 * n;
        return i < 2 * n - 1 - i ? i : 2 * n - 1 - i;
    case ORC_MIRROR:
        if (n == 1) return 0;
        if (i < 0) i = -i;
        i = 1 + (i - 1) % (2 * n - 2);
        return i < 2 * n - 2 - i ? i : 2 * n - 2 - i;
    case ORC_NEAREST:
        return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    case ORC_GRID_WRAP:
        i %= n;
        return i < 0 ? i + n : i;
    case ORC_WRAP:
        /* integer form of the period n-1 wrap (only used on taps that were
         * produced from an already wrapped float coordinate) */
        if (n == 1) return 0;
        if (i < 0) i += (n - 1) * (-i / (n - 1) + 1);
        else if (i > n - 1) i -= (n - 1) * (i / (n - 1));
        return i;
    default: /* constant / grid-constant */
        return (i < 0 || i >= n) ? -1 : i;
    }
}

int64_t orc_boundary_index(int64_t i, int64_t n, int mode) { return bmap(i, n, mode); }

/* ------------------------------------------------------------------ */
/* small n-D helpers                                                   */
/* ------------------------------------------------------------------ */

static int64_t prod(const int64_t *s, int n)
{
    int64_t p = 1;
    for (int i = 0; i < n; i++) p *= s[i];
    return p;
}

static void unravel(int64_t lin, const int64_t *shape, int ndim, int64_t *idx)
{
    for (int d = ndim - 1; d >= 0; d--) {
        idx[d] = shape[d] ? lin % shape[d] : 0;
        lin = shape[d] ? lin / shape[d] : 0;
    }
}

/* ------------------------------------------------------------------ */
/* 1-D correlate along one axis (double accumulate)                    */
/* ------------------------------------------------------------------ */
/*
 * out[.., o, ..] = sum_k w[k] * ext(in)[.., o - (wlen/2 + origin) + k, ..]
 *
 * Summation order mirrors SciPy's NI_Correlate1D so that double results are
 * bit-identical: symmetric odd kernels add the centre tap first and then the
 * pairs from the outside in; antisymmetric ones subtract; everything else
 * starts from the last tap and then runs left to right.
 */
int orc_correlate1d(const double *in, double *out, const int64_t *shape,
                    int ndim, int axis, const double *w, int wlen, int origin,
                    int mode, double cval)
{
    if (ndim < 1 || ndim > ORC_MAXDIM || axis < 0 || axis >= ndim || wlen < 1)
        return -1;
    const int64_t n = shape[axis];
    int64_t inner = 1, outer = 1;
    for (int d = axis + 1; d < ndim; d++) inner *= shape[d];
    for (int d = 0; d < axis; d++) outer *= shape[d];
    const int size1 = wlen / 2, size2 = wlen - size1 - 1;
    const int off = size1 + origin;
    if (off < 0 || off >= wlen) return -2;

    int symmetric = 0;
    if (wlen & 1) {
        symmetric = 1;
        for (int i = 1; i <= size1; i++)
            if (fabs(w[size1 + i] - w[size1 - i]) > 2.220446049250313e-16) { symmetric = 0; break; }
        if (!symmetric) {
            symmetric = -1;
            for (int i = 1; i <= size1; i++)
                if (fabs(w[size1 + i] + w[size1 - i]) > 2.220446049250313e-16) { symmetric = 0; break; }
        }
    }
    if (n == 0 || inner == 0 || outer == 0) return 0;

    double *line = (double *)malloc(sizeof(double) * (size_t)(n + wlen));
    if (!line) return -3;
    for (int64_t o = 0; o < outer; o++) {
        for (int64_t q = 0; q < inner; q++) {
            const double *src = in + o * n * inner + q;
            double *dst = out + o * n * inner + q;
            /* extended line: position p <-> input index p - off */
            for (int64_t p = 0; p < n + wlen - 1; p++) {
                int64_t j = bmap(p - off, n, mode);
                line[p] = j < 0 ? cval : src[j * inner];
            }
            for (int64_t l = 0; l < n; l++) {
                const double *c = line + l + size1; /* centre tap */
                const double *fw = w + size1;
                double acc;
                if (symmetric > 0) {
                    acc = c[0] * fw[0];
                    for (int j = -size1; j < 0; j++) acc += (c[j] + c[-j]) * fw[j];
                } else if (symmetric < 0) {
                    acc = c[0] * fw[0];
                    for (int j = -size1; j < 0; j++) acc += (c[j] - c[-j]) * fw[j];
                } else {
                    acc = c[size2] * fw[size2];
                    for (int j = -size1; j < size2; j++) acc += c[j] * fw[j];
                }
                dst[l * inner] = acc;
            }
        }
    }
    free(line);
    return 0;
}

/*
 * Box mean along one axis the way SciPy 1.15 does it (NI_UniformFilter1D):
 * a running *sum* in double, divided by the window length for every sample.
 * For integer-valued data the sums are exact, which is what makes the
 * truncating integer outputs of scipy.ndimage.uniform_filter reproducible
 * (the reference's weights-based formulation is not: tests/test_filters.py:444-450).
 */
int orc_uniform1d(const double *in, double *out, const int64_t *shape, int ndim,
                  int axis, int size, int origin, int mode, double cval)
{
    if (ndim < 1 || ndim > ORC_MAXDIM || axis < 0 || axis >= ndim || size < 1)
        return -1;
    const int64_t n = shape[axis];
    int64_t inner = 1, outer = 1;
    for (int d = axis + 1; d < ndim; d++) inner *= shape[d];
    for (int d = 0; d < axis; d++) outer *= shape[d];
    const int off = size / 2 + origin;
    if (off < 0 || off >= size) return -2;
    if (n == 0 || inner == 0 || outer == 0) return 0;
    double *line = (double *)malloc(sizeof(double) * (size_t)(n + size));
    if (!line) return -3;
    for (int64_t o = 0; o < outer; o++)
        for (int64_t q = 0; q < inner; q++) {
            const double *src = in + o * n * inner + q;
            double *dst = out + o * n * inner + q;
            for (int64_t p = 0; p < n + size - 1; p++) {
                int64_t j = bmap(p - off, n, mode);
                line[p] = j < 0 ? cval : src[j * inner];
            }
            double tmp = 0.0;
            for (int k = 0; k < size; k++) tmp += line[k];
            dst[0] = tmp / (double)size;
            for (int64_t l = 1; l < n; l++) {
                tmp += line[l + size - 1] - line[l - 1];
                dst[l * inner] = tmp / (double)size;
            }
        }
    free(line);
    return 0;
}

/* ------------------------------------------------------------------ */
/* dense n-D correlate                                                 */
/* ------------------------------------------------------------------ */
/*
 * out[o] = sum over taps t (C order, zero weights skipped) of
 *          w[t] * ext(in)[o - (wshape/2 + origin) + t]
 * (_filters_core.py:298-324; zero-weight skip at :242-246).  In constant mode
 * a tap is replaced by cval as soon as any axis falls outside (:276-293).
 */
int orc_correlate_nd(const double *in, double *out, const int64_t *shape,
                     int ndim, const double *w, const int64_t *wshape,
                     const int *origins, int mode, double cval)
{
    if (ndim < 1 || ndim > ORC_MAXDIM) return -1;
    int64_t off[ORC_MAXDIM], idx[ORC_MAXDIM], tap[ORC_MAXDIM], stride[ORC_MAXDIM];
    for (int d = 0; d < ndim; d++) {
        off[d] = wshape[d] / 2 + origins[d];
        if (wshape[d] > 0 && (off[d] < 0 || off[d] >= wshape[d])) return -2;
    }
    stride[ndim - 1] = 1;
    for (int d = ndim - 2; d >= 0; d--) stride[d] = stride[d + 1] * shape[d + 1];
    const int64_t total = prod(shape, ndim), ntap = prod(wshape, ndim);
    for (int64_t lin = 0; lin < total; lin++) {
        unravel(lin, shape, ndim, idx);
        double acc = 0.0;
        for (int64_t t = 0; t < ntap; t++) {
            if (w[t] == 0.0) continue;
            unravel(t, wshape, ndim, tap);
            int64_t pos = 0;
            int oob = 0;
            for (int d = 0; d < ndim; d++) {
                int64_t j = bmap(idx[d] - off[d] + tap[d], shape[d], mode);
                if (j < 0) { oob = 1; break; }
                pos += j * stride[d];
            }
            acc += (oob ? cval : in[pos]) * w[t];
        }
        out[lin] = acc;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* min / max over a footprint, optional non-flat structure             */
/* ------------------------------------------------------------------ */

/* (T)double the way an x86-64 SciPy build does it: through a wide signed
 * integer, then truncated to the low bits. */
static double wrap_to_dtype(double a, int dtype)
{
    switch (dtype) {
    case ORC_BOOL: return (double)(uint8_t)(a != 0.0);
    case ORC_I8:   return (double)(int8_t)(int64_t)a;
    case ORC_U8:   return (double)(uint8_t)(int64_t)a;
    case ORC_I16:  return (double)(int16_t)(int64_t)a;
    case ORC_U16:  return (double)(uint16_t)(int64_t)a;
    case ORC_I32:  return (double)(int32_t)(int64_t)a;
    case ORC_U32:  return (double)(uint32_t)(int64_t)a;
    case ORC_I64:  return (double)(int64_t)a;
    case ORC_U64:  return (double)(a >= 0 ? (uint64_t)a : (uint64_t)(-(int64_t)(uint64_t)(-a)));
    case ORC_F32:  return (double)(float)a;
    default:       return a;
    }
}

/*
 * filters.py:1510-1557: running min/max over the taps whose footprint entry
 * is set; with a structure the tap value is x - s (min) or x + s (max).
 *
 * Integer exactness follows SciPy's NI_MinOrMaxFilter, which the reference's
 * tests compare against: cval is first converted to the *input* dtype
 * (uint8: 300 -> 44); the first set tap is evaluated in double, every later
 * tap adds the structure value in the input dtype (so unsigned types wrap,
 * exactly like the reference's `cast<X>(sval)` arithmetic in X), and the
 * running result is compared as double.  dtype < 0 turns all of that off
 * (pure double arithmetic, used for the separable 1-D passes whose line
 * buffers are double in SciPy).
 */
int orc_minmax_nd(const double *in, double *out, const int64_t *shape, int ndim,
                  const uint8_t *fp, const double *st, const int64_t *fshape,
                  const int *origins, int mode, double cval, int is_max, int dtype)
{
    if (ndim < 1 || ndim > ORC_MAXDIM) return -1;
    int64_t off[ORC_MAXDIM], idx[ORC_MAXDIM], tap[ORC_MAXDIM], stride[ORC_MAXDIM];
    for (int d = 0; d < ndim; d++) {
        off[d] = fshape[d] / 2 + origins[d];
        if (off[d] < 0 || off[d] >= fshape[d]) return -2;
    }
    stride[ndim - 1] = 1;
    for (int d = ndim - 2; d >= 0; d--) stride[d] = stride[d + 1] * shape[d + 1];
    const int64_t total = prod(shape, ndim), ntap = prod(fshape, ndim);
    if (dtype >= 0) cval = wrap_to_dtype(cval, dtype);
    for (int64_t lin = 0; lin < total; lin++) {
        unravel(lin, shape, ndim, idx);
        double best = 0.0;
        int have = 0;
        for (int64_t t = 0; t < ntap; t++) {
            if (fp && !fp[t]) continue;
            unravel(t, fshape, ndim, tap);
            int64_t pos = 0;
            int oob = 0;
            for (int d = 0; d < ndim; d++) {
                int64_t j = bmap(idx[d] - off[d] + tap[d], shape[d], mode);
                if (j < 0) { oob = 1; break; }
                pos += j * stride[d];
            }
            double v = oob ? cval : in[pos];
            if (st) {
                double s = is_max ? st[t] : -st[t];
                if (dtype < 0 || !have) v = v + s;
                else if (dtype == ORC_F32) v = (double)((float)v + (float)s);
                else if (dtype == ORC_F64) v = v + s;
                else v = wrap_to_dtype(v + wrap_to_dtype(s, dtype), dtype);
            }
            if (!have || (is_max ? v > best : v < best)) { best = v; have = 1; }
        }
        out[lin] = best;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* binary erosion (one iteration); dilation through `invert`           */
/* ------------------------------------------------------------------ */
/*
 * morphology.py:41-128.  `in` is nonzero/zero.  With invert=0 the output is
 * true iff every set structure tap sees a true voxel (outside the array the
 * tap sees border_value).  invert=1 computes the complement on the complement
 * (true/false swapped, border inverted), which is how binary_dilation is
 * expressed (morphology.py:443-461).  With a mask, voxels where mask==0 are
 * copied from the input.
 */
int orc_binary_erosion(const uint8_t *in, uint8_t *out, const int64_t *shape,
                       int ndim, const uint8_t *st, const int64_t *sshape,
                       const int *origins, const uint8_t *mask,
                       int border_value, int invert)
{
    if (ndim < 1 || ndim > ORC_MAXDIM) return -1;
    int64_t off[ORC_MAXDIM], idx[ORC_MAXDIM], tap[ORC_MAXDIM], stride[ORC_MAXDIM];
    for (int d = 0; d < ndim; d++) off[d] = sshape[d] / 2 + origins[d];
    stride[ndim - 1] = 1;
    for (int d = ndim - 2; d >= 0; d--) stride[d] = stride[d + 1] * shape[d + 1];
    const int64_t total = prod(shape, ndim), ntap = prod(sshape, ndim);
    const int tv = invert ? 0 : 1, fv = invert ? 1 : 0;
    const int bv = invert ? !border_value : !!border_value;
    for (int64_t lin = 0; lin < total; lin++) {
        const int cur = in[lin] != 0;
        if (mask && !mask[lin]) { out[lin] = (uint8_t)cur; continue; }
        unravel(lin, shape, ndim, idx);
        int res = tv;
        for (int64_t t = 0; t < ntap && res == tv; t++) {
            if (!st[t]) continue;
            unravel(t, sshape, ndim, tap);
            int64_t pos = 0;
            int oob = 0;
            for (int d = 0; d < ndim; d++) {
                int64_t j = idx[d] - off[d] + tap[d];
                if (j < 0 || j >= shape[d]) { oob = 1; break; }
                pos += j * stride[d];
            }
            if (oob) { if (!bv) res = fv; }
            else {
                int nn = (in[pos] != 0) ? tv : fv;
                if (!nn) res = fv;
            }
        }
        out[lin] = (uint8_t)res;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* interpolation, spline order 0 and 1                                 */
/* ------------------------------------------------------------------ */

/* float coordinate wrap with period n-1 (mode 'wrap'), _util.py:210-218 */
static double wrap_coord(double c, int64_t n)
{
    if (n <= 1) return 0.0;
    double s = (double)(n - 1);
    if (c < 0) c += s * ((double)(int64_t)(-c / s) + 1.0);
    else if (c > s) c -= s * (double)(int64_t)(c / s);
    return c;
}

/* Float coordinate folded into the array the way SciPy's map_coordinate()
 * does it (published algorithm of scipy/ndimage/src/ni_interpolation.c,
 * SciPy 1.15.3; not part of /root/reference).  Only needed for order 0, where
 * rounding happens after the fold. */
static double fold_coord(double c, int64_t n, int mode)
{
    if (n <= 1) return 0.0;
    const double dn = (double)n;
    switch (mode) {
    case ORC_MIRROR: {
        const double p = 2.0 * dn - 2.0;
        if (c < 0) { c = p * (double)(int64_t)(-c / p) + c; c = c <= 1.0 - dn ? c + p : -c; }
        else if (c > dn - 1.0) { c -= p * (double)(int64_t)(c / p); if (c >= dn) c = p - c; }
        return c;
    }
    case ORC_REFLECT: {
        const double p = 2.0 * dn;
        if (c < 0) {
            if (c < -p) c = p * (double)(int64_t)(-c / p) + c;
            c = c < -dn ? c + p : (c > -1e-15 ? 1e-15 : -c) - 1.0;
        } else if (c > dn - 1.0) {
            c -= p * (double)(int64_t)(c / p);
            if (c >= dn) c = p - c - 1.0;
        }
        return c;
    }
    case ORC_WRAP:
        return wrap_coord(c, n);
    case ORC_GRID_WRAP:
        if (c < 0) c += dn * ((double)(int64_t)((-1.0 - c) / dn) + 1.0);
        else if (c > dn - 1.0) c -= dn * (double)(int64_t)((c + 1.0) / dn);
        return c;
    case ORC_NEAREST:
        return c < 0 ? 0.0 : (c > dn - 1.0 ? dn - 1.0 : c);
    default:
        return c;
    }
}

/* ------------------------------------------------------------------ */
/* B-spline orders 2..5 (published algorithm of scipy/ndimage/src/ni_splines.c
 * and ni_interpolation.c, SciPy 1.15.3 -- a dependency whose source is not
 * under /root/reference; the reference restates the same recurrences in
 * _spline_prefilter_core.py:14-139 and _spline_kernel_weights.py).  Pinned by
 * tests/golden fixtures generated from SciPy.                            */
/* ------------------------------------------------------------------ */
static int spline_poles(int order, double *z)
{
    switch (order) {
    case 2: z[0] = -0.171572875253809902396622551580603843; return 1;
    case 3: z[0] = -0.267949192431122706472553658494127633; return 1;
    case 4: z[0] = -0.361341225900220177092212841325675255; z[1] = -0.013725429297339121360331226939128204; return 2;
    case 5: z[0] = -0.430575347099973791851434783493520110; z[1] = -0.043096288203264653822712376822550182; return 2;
    }
    return 0;
}

/* in-place prefilter of one line c[0], c[st], ...; smode: 0 mirror, 1 reflect, 2 grid-wrap */
static void spline_line(double *c, int64_t n, int64_t st, int order, int smode)
{
    double zs[2];
    const int np = spline_poles(order, zs);
    if (n <= 1) return;                       /* a single sample is left as it is */
    double gain = 1.0;
    for (int k = 0; k < np; k++) gain *= (1.0 - zs[k]) * (1.0 - 1.0 / zs[k]);
    for (int64_t i = 0; i < n; i++) c[i * st] *= gain;
    for (int k = 0; k < np; k++) {
        const double z = zs[k];
        double z_i = z;
        if (smode == 0) {
            const double z_n_1 = pow(z, (double)(n - 1));
            c[0] = c[0] + z_n_1 * c[(n - 1) * st];
            for (int64_t i = 1; i < n - 1; i++) { c[0] += z_i * (c[i * st] + z_n_1 * c[(n - 1 - i) * st]); z_i *= z; }
            c[0] /= 1 - z_n_1 * z_n_1;
        } else if (smode == 2) {
            for (int64_t i = 1; i < n; i++) { c[0] += z_i * c[(n - i) * st]; z_i *= z; }
            c[0] /= 1 - z_i;
        } else {
            const double z_n = pow(z, (double)n), c0 = c[0];
            c[0] = c[0] + z_n * c[(n - 1) * st];
            for (int64_t i = 1; i < n; i++) { c[0] += z_i * (c[i * st] + z_n * c[(n - 1 - i) * st]); z_i *= z; }
            c[0] *= z / (1 - z_n * z_n);
            c[0] += c0;
        }
        for (int64_t i = 1; i < n; i++) c[i * st] += z * c[(i - 1) * st];
        if (smode == 0) {
            c[(n - 1) * st] = (z * c[(n - 2) * st] + c[(n - 1) * st]) * z / (z * z - 1);
        } else if (smode == 2) {
            z_i = z;
            for (int64_t i = 0; i < n - 1; i++) { c[(n - 1) * st] += z_i * c[i * st]; z_i *= z; }
            c[(n - 1) * st] *= z / (z_i - 1);
        } else {
            c[(n - 1) * st] *= z / (z - 1);
        }
        for (int64_t i = n - 2; i >= 0; i--) c[i * st] = z * (c[(i + 1) * st] - c[i * st]);
    }
}

/* in-place along `axis` of a C-contiguous float64 array */
int orc_spline_filter1d(double *data, const int64_t *shape, int ndim, int axis, int order, int smode)
{
    if (ndim < 1 || ndim > ORC_MAXDIM || axis < 0 || axis >= ndim || order < 2 || order > 5) return -1;
    int64_t inner = 1, outer = 1;
    for (int d = axis + 1; d < ndim; d++) inner *= shape[d];
    for (int d = 0; d < axis; d++) outer *= shape[d];
    const int64_t n = shape[axis];
    for (int64_t o = 0; o < outer; o++)
        for (int64_t k = 0; k < inner; k++) spline_line(data + o * n * inner + k, n, inner, order, smode);
    return 0;
}

static void spline_weights(double x, int order, double *w)
{
    double y;
    switch (order) {
    case 2:
        w[1] = 0.75 - x * x; y = 0.5 - x; w[0] = 0.5 * y * y; w[2] = 1.0 - w[0] - w[1];
        break;
    case 3:
        y = 1.0 - x;
        w[1] = (x * x * (x - 2.0) * 3.0 + 4.0) / 6.0;
        w[2] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w[0] = y * y * y / 6.0;
        w[3] = 1.0 - w[0] - w[1] - w[2];
        break;
    case 4:
        y = x * x;
        w[2] = y * (y * 0.25 - 0.625) + 115.0 / 192.0;
        y = 1.0 + x;
        w[1] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        y = 1.0 - x;
        w[3] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        y = 0.5 - x; y = y * y;
        w[0] = y * y / 24.0;
        w[4] = 1.0 - w[0] - w[1] - w[2] - w[3];
        break;
    default:
        y = x * x;
        w[2] = y * (y * (0.25 - x / 12.0) - 0.5) + 0.55;
        y = 1.0 - x; y = y * y;
        w[3] = y * (y * (0.25 - (1.0 - x) / 12.0) - 0.5) + 0.55;
        y = x + 1.0;
        w[1] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        y = 2.0 - x;
        w[4] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        y = 1.0 - x; y = y * y;
        w[0] = (1.0 - x) * y * y / 120.0;
        w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
        break;
    }
}

/* spline tap index outside [0, n): the symmetry the coefficients were computed with */
static int64_t spline_tap(int64_t i, int64_t n, int mode)
{
    if (i >= 0 && i < n) return i;
    if (mode == ORC_GRID_CONSTANT) return -1;
    if (mode == ORC_REFLECT) return bmap(i, n, ORC_REFLECT);
    if (mode == ORC_NEAREST) return i < 0 ? 0 : n - 1;      /* taps are clamped, the coordinate is not */
    if (mode == ORC_GRID_WRAP) return bmap(i, n, ORC_GRID_WRAP);
    return bmap(i, n, ORC_MIRROR);
}

/* order 2..5 on prefiltered coefficients `in` (already padded by `npad` for
 * nearest / grid-constant); c = coordinates in the unpadded frame */
static double spline_point(const double *in, const int64_t *shape, const int64_t *stride, int ndim, const double *c,
                           int order, int mode, double cval, int npad)
{
    double w[ORC_MAXDIM][6];
    int64_t idx[ORC_MAXDIM][6];
    for (int d = 0; d < ndim; d++) {
        double cc = c[d] + (double)npad;
        const int64_t n = shape[d];
        if (mode == ORC_CONSTANT) {
            if (cc < 0 || cc > (double)(n - 1)) return cval;
        } else if (mode != ORC_GRID_CONSTANT && mode != ORC_NEAREST) {
            cc = fold_coord(cc, n, mode);
        }
        const double fl = (order & 1) ? floor(cc) : floor(cc + 0.5);
        const int64_t start = (int64_t)fl - order / 2;
        spline_weights(cc - fl, order, w[d]);
        for (int k = 0; k <= order; k++) idx[d][k] = spline_tap(start + k, n, mode);
    }
    int k[ORC_MAXDIM] = {0};
    double acc = 0.0;
    for (;;) {
        int64_t pos = 0;
        int oob = 0;
        for (int d = 0; d < ndim; d++) {
            if (idx[d][k[d]] < 0) oob = 1; else pos += idx[d][k[d]] * stride[d];
        }
        /* SciPy multiplies the sample by its weights one axis at a time (matters at exact ties of integer outputs) */
        double coeff = oob ? cval : in[pos];
        for (int d = 0; d < ndim; d++) coeff *= w[d][k[d]];
        acc += coeff;
        int d = ndim - 1;
        while (d >= 0 && ++k[d] > order) { k[d] = 0; d--; }
        if (d < 0) break;
    }
    return acc;
}

/* value of one output sample at float coordinates c[0..ndim) */
static double interp_point(const double *in, const int64_t *shape,
                           const int64_t *stride, int ndim, const double *c,
                           int order, int mode, double cval)
{
    if (mode == ORC_CONSTANT)
        for (int d = 0; d < ndim; d++)
            if (c[d] < 0 || c[d] > (double)(shape[d] - 1)) return cval;

    if (order == 0) {
        /* SciPy maps the *float* coordinate into the array first and rounds
         * half up afterwards (NI_GeometricTransform: map_coordinate, then
         * floor(c + 0.5)); the reference rounds first with lrint and excludes
         * ties from its own tests (tests/test_interpolation.py:362-364).  The
         * two only differ at exact half-integer coordinates outside the
         * array; the oracle follows SciPy. */
        int64_t pos = 0;
        for (int d = 0; d < ndim; d++) {
            int64_t j;
            if (mode == ORC_CONSTANT) {
                j = (int64_t)floor(c[d] + 0.5);
            } else if (mode == ORC_GRID_CONSTANT) {
                j = bmap((int64_t)floor(c[d] + 0.5), shape[d], mode);
            } else {
                double f = fold_coord(c[d], shape[d], mode);
                j = bmap((int64_t)floor(f + 0.5), shape[d], mode);
            }
            if (j < 0) return cval; /* grid-constant */
            pos += j * stride[d];
        }
        return in[pos];
    }

    /* order 1: 2^ndim taps.  Arithmetic as SciPy 1.15 does it (ni_interpolation.c / ni_splines.c), so that
     * integer outputs agree at exact ties: the coordinate is folded first, the weights are 1 - x and
     * 1 - (1 - x) with x the fraction of the folded coordinate, and each sample is multiplied by its weights
     * one axis at a time before it is added.  The upper tap is skipped at integral coordinates
     * (_interp_kernels.py:416), which only matters for non-finite samples. */
    int64_t lo[ORC_MAXDIM], hi[ORC_MAXDIM];
    double wlo[ORC_MAXDIM], whi[ORC_MAXDIM];
    int npt[ORC_MAXDIM];
    for (int d = 0; d < ndim; d++) {
        double cc = c[d];
        if (mode != ORC_CONSTANT && mode != ORC_GRID_CONSTANT && mode != ORC_NEAREST) cc = fold_coord(cc, shape[d], mode);
        double cf = floor(cc);
        npt[d] = (cc == cf) ? 1 : 2;
        wlo[d] = 1.0 - (cc - cf);
        whi[d] = 1.0 - wlo[d];
        lo[d] = (int64_t)cf;
        hi[d] = lo[d] + 1;
        if (mode != ORC_CONSTANT) {
            lo[d] = bmap(lo[d], shape[d], mode == ORC_WRAP ? ORC_MIRROR : mode);
            hi[d] = bmap(hi[d], shape[d], mode == ORC_WRAP ? ORC_MIRROR : mode);
        }
    }
    double acc = 0.0;
    const int ncorner = 1 << ndim;
    for (int m = 0; m < ncorner; m++) {
        int64_t pos = 0;
        int skip = 0, oob = 0;
        for (int d = 0; d < ndim; d++) {
            int up = (m >> (ndim - 1 - d)) & 1;
            if (up && npt[d] == 1) { skip = 1; break; }
            int64_t j = up ? hi[d] : lo[d];
            if (j < 0) oob = 1; else pos += j * stride[d];
        }
        if (skip) continue;
        double coeff = oob ? cval : in[pos];
        for (int d = 0; d < ndim; d++) coeff *= ((m >> (ndim - 1 - d)) & 1) ? whi[d] : wlo[d];
        acc += coeff;
    }
    return acc;
}

/* coords: (ndim, nout) C-contiguous, read as coords[d*nout + i]
 * (_interp_kernels.py:38-46) */
int orc_map_coordinates(const double *in, const int64_t *shape, int ndim,
                        const double *coords, int64_t nout, double *out,
                        int order, int mode, double cval, int npad)
{
    if (ndim < 1 || ndim > ORC_MAXDIM || order < 0 || order > 5) return -1;
    int64_t stride[ORC_MAXDIM];
    double c[ORC_MAXDIM];
    stride[ndim - 1] = 1;
    for (int d = ndim - 2; d >= 0; d--) stride[d] = stride[d + 1] * shape[d + 1];
    for (int64_t i = 0; i < nout; i++) {
        for (int d = 0; d < ndim; d++) c[d] = coords[d * nout + i];
        out[i] = order > 1 ? spline_point(in, shape, stride, ndim, c, order, mode, cval, npad)
                           : interp_point(in, shape, stride, ndim, c, order, mode, cval);
    }
    return 0;
}

/* mat: (ndim, ndim+1) row-major; c = mat[:, :ndim] @ o + mat[:, ndim]
 * (_interp_kernels.py:198-242) */
int orc_affine_transform(const double *in, const int64_t *shape, int ndim,
                         const double *mat, double *out, const int64_t *oshape,
                         int order, int mode, double cval, int npad)
{
    if (ndim < 1 || ndim > ORC_MAXDIM || order < 0 || order > 5) return -1;
    int64_t stride[ORC_MAXDIM], o[ORC_MAXDIM];
    double c[ORC_MAXDIM];
    stride[ndim - 1] = 1;
    for (int d = ndim - 2; d >= 0; d--) stride[d] = stride[d + 1] * shape[d + 1];
    const int64_t total = prod(oshape, ndim);
    for (int64_t lin = 0; lin < total; lin++) {
        unravel(lin, oshape, ndim, o);
        for (int d = 0; d < ndim; d++) {
            double s = 0.0;
            for (int k = 0; k < ndim; k++) s += mat[d * (ndim + 1) + k] * (double)o[k];
            c[d] = s + mat[d * (ndim + 1) + ndim];
        }
        out[lin] = order > 1 ? spline_point(in, shape, stride, ndim, c, order, mode, cval, npad)
                             : interp_point(in, shape, stride, ndim, c, order, mode, cval);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* double -> user dtype with C cast semantics                          */
/* ------------------------------------------------------------------ */
/* Truncation toward zero; negative values into unsigned types go through the
 * signed 64-bit value and wrap (what x86 SciPy builds do and what
 * _filters_core.py:173-183 spells as -(B)(-a)). */
#define CAST_LOOP(T, EXPR) do { T *d = (T *)dst; \
    for (int64_t i = 0; i < n; i++) { double a = src[i]; d[i] = (EXPR); } } while (0)

/* SciPy's rounding for integer outputs of the interpolation routines
 * (CASE_INTERP_OUT_INT / _UINT in ni_interpolation.c): half away from zero,
 * then clipped to the output range.  The reference uses rint()
 * (_interp_kernels.py:580-583), which differs only at exact .5 values. */
static double interp_round(double t, int dtype)
{
    double lo, hi;
    switch (dtype) {
    case ORC_I8:  lo = -128.0; hi = 127.0; break;
    case ORC_U8:  lo = 0.0; hi = 255.0; break;
    case ORC_I16: lo = -32768.0; hi = 32767.0; break;
    case ORC_U16: lo = 0.0; hi = 65535.0; break;
    case ORC_I32: lo = -2147483648.0; hi = 2147483647.0; break;
    case ORC_U32: lo = 0.0; hi = 4294967295.0; break;
    case ORC_I64: lo = -9223372036854775808.0; hi = 9223372036854775807.0; break;
    case ORC_U64: lo = 0.0; hi = 18446744073709551615.0; break;
    default: return t;
    }
    if (lo == 0.0) t = t > 0 ? t + 0.5 : 0.0;
    else t = t > 0 ? t + 0.5 : t - 0.5;
    if (t > hi) t = hi;
    if (t < lo) t = lo;
    return t;
}

int orc_cast_from_f64(const double *src, void *dst, int64_t n, int dtype, int interp_rounding)
{
    if (interp_rounding) {
        double *tmp = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
        if (!tmp) return -3;
        for (int64_t i = 0; i < n; i++) tmp[i] = interp_round(src[i], dtype);
        int rc = orc_cast_from_f64(tmp, dst, n, dtype, 0);
        free(tmp);
        return rc;
    }
    switch (dtype) {
    case ORC_BOOL: CAST_LOOP(uint8_t, (uint8_t)(a != 0.0)); break;
    case ORC_I8:   CAST_LOOP(int8_t, (int8_t)(int64_t)a); break;
    case ORC_U8:   CAST_LOOP(uint8_t, (uint8_t)(int64_t)a); break;
    case ORC_I16:  CAST_LOOP(int16_t, (int16_t)(int64_t)a); break;
    case ORC_U16:  CAST_LOOP(uint16_t, (uint16_t)(int64_t)a); break;
    case ORC_I32:  CAST_LOOP(int32_t, (int32_t)(int64_t)a); break;
    case ORC_U32:  CAST_LOOP(uint32_t, (uint32_t)(int64_t)a); break;
    case ORC_I64:  CAST_LOOP(int64_t, (int64_t)a); break;
    case ORC_U64:  CAST_LOOP(uint64_t, a >= 0 ? (uint64_t)a : (uint64_t)(-(int64_t)(uint64_t)(-a))); break;
    case ORC_F32:  CAST_LOOP(float, (float)a); break;
    case ORC_F64:  CAST_LOOP(double, a); break;
    default: return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* typed fast paths used only as the timed CPU baseline (bench.py)     */
/* ------------------------------------------------------------------ */
/*
 * float32 box filter along one axis of a C-contiguous 3-D volume, written the
 * way a scalar CPU port would be: one line at a time through a double line
 * buffer, running sum, result rounded to float32 per pass (what
 * scipy.ndimage.uniform_filter does for float32 arrays).  Single thread.
 */
static void box_axis_f32(const float *in, float *out, int64_t n0, int64_t n1,
                         int64_t n2, int axis, int size, int mode, double cval)
{
    const int64_t shape[3] = {n0, n1, n2};
    const int64_t n = shape[axis];
    int64_t inner = 1, outer = 1;
    for (int d = axis + 1; d < 3; d++) inner *= shape[d];
    for (int d = 0; d < axis; d++) outer *= shape[d];
    const int off = size / 2;
    double *line = (double *)malloc(sizeof(double) * (size_t)(n + size));
    for (int64_t o = 0; o < outer; o++)
        for (int64_t q = 0; q < inner; q++) {
            const float *src = in + o * n * inner + q;
            float *dst = out + o * n * inner + q;
            for (int64_t p = 0; p < n + size - 1; p++) {
                int64_t j = bmap(p - off, n, mode);
                line[p] = j < 0 ? cval : (double)src[j * inner];
            }
            double tmp = 0.0;
            for (int k = 0; k < size; k++) tmp += line[k];
            dst[0] = (float)(tmp / (double)size);
            for (int64_t l = 1; l < n; l++) {
                tmp += line[l + size - 1] - line[l - 1];
                dst[l * inner] = (float)(tmp / (double)size);
            }
        }
    free(line);
}

int orc_uniform3d_f32(const float *in, float *out, float *tmp, int64_t n0,
                      int64_t n1, int64_t n2, int size, int mode, double cval)
{
    if (size < 1) return -1;
    box_axis_f32(in, out, n0, n1, n2, 0, size, mode, cval);
    box_axis_f32(out, tmp, n0, n1, n2, 1, size, mode, cval);
    box_axis_f32(tmp, out, n0, n1, n2, 2, size, mode, cval);
    return 0;
}

/* uint8 running min/max along one axis, 3-D volume, scalar port */
static void minmax_axis_u8(const uint8_t *in, uint8_t *out, int64_t n0, int64_t n1,
                           int64_t n2, int axis, int size, int mode, int cval, int is_max)
{
    const int64_t shape[3] = {n0, n1, n2};
    const int64_t n = shape[axis];
    int64_t inner = 1, outer = 1;
    for (int d = axis + 1; d < 3; d++) inner *= shape[d];
    for (int d = 0; d < axis; d++) outer *= shape[d];
    const int off = size / 2;
    uint8_t *line = (uint8_t *)malloc((size_t)(n + size));
    for (int64_t o = 0; o < outer; o++)
        for (int64_t q = 0; q < inner; q++) {
            const uint8_t *src = in + o * n * inner + q;
            uint8_t *dst = out + o * n * inner + q;
            for (int64_t p = 0; p < n + size - 1; p++) {
                int64_t j = bmap(p - off, n, mode);
                line[p] = j < 0 ? (uint8_t)cval : src[j * inner];
            }
            for (int64_t l = 0; l < n; l++) {
                uint8_t b = line[l];
                for (int k = 1; k < size; k++) {
                    uint8_t v = line[l + k];
                    if (is_max ? v > b : v < b) b = v;
                }
                dst[l * inner] = b;
            }
        }
    free(line);
}

int orc_minmax3d_u8(const uint8_t *in, uint8_t *out, uint8_t *tmp, int64_t n0,
                    int64_t n1, int64_t n2, int size, int mode, int cval, int is_max)
{
    if (size < 1) return -1;
    minmax_axis_u8(in, out, n0, n1, n2, 0, size, mode, cval, is_max);
    minmax_axis_u8(out, tmp, n0, n1, n2, 1, size, mode, cval, is_max);
    minmax_axis_u8(tmp, out, n0, n1, n2, 2, size, mode, cval, is_max);
    return 0;
}

/* ------------------------------------------------------------------------
 * Counter-based synthetic volume (bench.py --config E): the CPU restatement of
 * cupyimg_amd/csrc/synth.hip, bit-identical to it (exact float32 sums of 22-bit
 * uniforms, one correctly rounded product).  oracle/synth.py is the front end.
 * ---------------------------------------------------------------------- */
static uint64_t orc_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int orc_synth_f32(float *out, int64_t n, uint64_t first_index, uint64_t seed)
{
    const float scale = 1.7320508075688772f;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t c = seed + 4ull * (first_index + (uint64_t)i);
        float s = 0.f;
        for (int k = 0; k < 4; k++) s += (float)(uint32_t)(orc_splitmix64(c + (uint64_t)k) >> 42) * 0x1p-22f;
        out[i] = (s - 2.0f) * scale;
    }
    return 0;
}
