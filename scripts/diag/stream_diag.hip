// Standalone diagnostic for the streaming-pass failure (> 1024 waves per launch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I include -I cupyimg_amd/csrc scripts/diag/stream_diag.hip -o scripts/diag/stream_diag.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {
void set_error(const char *, ...) {}

enum { V_NOP = 1, V_BPERM = 2, V_NOBRANCH = 4, V_NOEDGE = 8, V_WAIT = 32, V_ST0 = 64, V_ST1 = 128, V_ST3 = 256, V_ST7 = 512, V_STVOFF = 1024, V_STCOPY = 2048 };

__device__ __forceinline__ void edge_block(int side, int j, int x0, int xe, int nx, int mode, int *start, int *kind)
{
    if (side == 0) {
        if (x0 - 4 * j >= 0) { *start = x0 - 4 * j; *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = 4 * (j - 1); *kind = EDGE_REV; break;
        case MI_MODE_MIRROR:    *start = 4 * (j - 1) + 1; *kind = EDGE_REV; break;
        case MI_MODE_NEAREST:   *start = 0; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = nx - 4 * j; *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    } else {
        if (xe + 4 * j <= nx) { *start = xe + 4 * (j - 1); *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = nx - 4 * j; *kind = EDGE_REV; break;
        case MI_MODE_MIRROR:    *start = nx - 1 - 4 * j; *kind = EDGE_REV; break;
        case MI_MODE_NEAREST:   *start = nx - 4; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = 4 * (j - 1); *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    }
}

template <int VAR>
__device__ __forceinline__ float4 apply_kind(float4 t, int kind, int side, float cval)
{
    if constexpr (VAR & V_NOBRANCH) {
        const float4 rev = make_float4(t.w, t.z, t.y, t.x);
        const float s = side == 0 ? t.x : t.w;
        float4 r = t;
        r.x = kind == EDGE_REV ? rev.x : kind == EDGE_SPLAT ? s : kind == EDGE_CONST ? cval : t.x;
        r.y = kind == EDGE_REV ? rev.y : kind == EDGE_SPLAT ? s : kind == EDGE_CONST ? cval : t.y;
        r.z = kind == EDGE_REV ? rev.z : kind == EDGE_SPLAT ? s : kind == EDGE_CONST ? cval : t.z;
        r.w = kind == EDGE_REV ? rev.w : kind == EDGE_SPLAT ? s : kind == EDGE_CONST ? cval : t.w;
        return r;
    } else {
        if (kind == EDGE_REV) return make_float4(t.w, t.z, t.y, t.x);
        if (kind == EDGE_SPLAT) { const float s = side == 0 ? t.x : t.w; return make_float4(s, s, s, s); }
        if (kind == EDGE_CONST) return make_float4(cval, cval, cval, cval);
        return t;
    }
}

template <int VAR>
__device__ __forceinline__ float shl1(float keep, float v, int lane)
{
    if constexpr (VAR & V_BPERM) {
        const int r = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __float_as_int(v));
        return lane == 63 ? keep : __int_as_float(r);
    } else {
        if constexpr (VAR & V_NOP) asm volatile("s_nop 7" ::: "memory");
        float r = dpp_from_right(keep, v);
        if constexpr (VAR & V_NOP) asm volatile("s_nop 7" ::: "memory");
        return r;
    }
}
template <int VAR>
__device__ __forceinline__ float shr1(float keep, float v, int lane)
{
    if constexpr (VAR & V_BPERM) {
        const int r = __builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, __float_as_int(v));
        return lane == 0 ? keep : __int_as_float(r);
    } else {
        if constexpr (VAR & V_NOP) asm volatile("s_nop 7" ::: "memory");
        float r = dpp_from_left(keep, v);
        if constexpr (VAR & V_NOP) asm volatile("s_nop 7" ::: "memory");
        return r;
    }
}

enum { SP_CORR = 0, SP_MIN = 1, SP_MAX = 2 };
template <int OP>
__device__ __forceinline__ float pick_mm(float x, float best) { return (OP == SP_MAX ? x > best : x < best) ? x : best; }
template <int OP>
__device__ __forceinline__ F4 f4_mm(const F4 x, const F4 best)
{
    F4 r;
    r.lo = (f32x2){pick_mm<OP>(x.lo.x, best.lo.x), pick_mm<OP>(x.lo.y, best.lo.y)};
    r.hi = (f32x2){pick_mm<OP>(x.hi.x, best.hi.x), pick_mm<OP>(x.hi.y, best.hi.y)};
    return r;
}

template <int WX, int OP, int VAR>
__device__ __forceinline__ F4 xpass_hops(const float4 v, const float4 (&eL)[4], const float4 (&eR)[4], int lane, int last)
{
    constexpr int RX = WX / 2;
    constexpr int NB = (RX + 3) / 4;
    float4 blk[2 * NB + 1];
    blk[NB] = v;
    float4 l = v, r = v;
#pragma unroll
    for (int j = 1; j <= NB; j++) {
        l = make_float4(shr1<VAR>(eL[j - 1].x, l.x, lane), shr1<VAR>(eL[j - 1].y, l.y, lane), shr1<VAR>(eL[j - 1].z, l.z, lane), shr1<VAR>(eL[j - 1].w, l.w, lane));
        float4 rr = make_float4(shl1<VAR>(eR[j - 1].x, r.x, lane), shl1<VAR>(eR[j - 1].y, r.y, lane), shl1<VAR>(eR[j - 1].z, r.z, lane), shl1<VAR>(eR[j - 1].w, r.w, lane));
        r = lane == last ? eR[j - 1] : rr;
        blk[NB - j] = l;
        blk[NB + j] = r;
    }
    constexpr int NP = 2 * (2 * NB + 1);
    f32x2 A[NP];
#pragma unroll
    for (int b = 0; b < 2 * NB + 1; b++) {
        A[2 * b] = (f32x2){blk[b].x, blk[b].y};
        A[2 * b + 1] = (f32x2){blk[b].z, blk[b].w};
    }
    constexpr int BASE = 4 * NB - RX;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        auto win = [&](int t) { return (t & 1) ? A[t >> 1].y : A[t >> 1].x; };
        float best = win(BASE + c);
#pragma unroll
        for (int k = 1; k < WX; k++) best = pick_mm<OP>(win(BASE + c + k), best);
        o[c] = best;
    }
    F4 rr;
    rr.lo = (f32x2){o[0], o[1]};
    rr.hi = (f32x2){o[2], o[3]};
    return rr;
}

constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
constexpr int lcm_(int a, int b) { return a / gcd_(a, b) * b; }

template <int WX, int WA, int DEPTH, int OP, int VAR>
__device__ __forceinline__ void body(const float *__restrict__ in, float *__restrict__ out, const StreamParams &p)
{
    constexpr int RX = WX / 2;
    constexpr int NB = WX > 1 ? (RX + 3) / 4 : 0;
    constexpr int RINGN = WA - 1;
    constexpr int U = RINGN > 0 ? lcm_(RINGN, DEPTH) : DEPTH;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nother = p.axis == 0 ? ny : nz;
    const int nA = p.axis == 0 ? nz : ny;
    const int nlines = nother * p.nxt;
    const int wid = p.wid_base + blockIdx.x * p.wpb + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int oth = line / p.nxt, xt = line - oth * p.nxt;
    const int x0 = xt * 256;
    const int nlanes = min(64, (nx - x0) >> 2);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned strideA = p.axis == 0 ? plane : (unsigned)nx;
    const unsigned rowbase = p.axis == 0 ? (unsigned)oth * nx : (unsigned)oth * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 4 * lane)) * 4u : kOOB;

    unsigned evoff[4] = {kOOB, kOOB, kOOB, kOOB};
    int ekind[4] = {EDGE_FWD, EDGE_FWD, EDGE_FWD, EDGE_FWD};
    const int side = lane == 0 ? 0 : 1;
    {
        const bool is_edge_lane = lane == 0 || lane == last;
#pragma unroll
        for (int j = 1; j <= NB; j++) {
            int st, kd;
            edge_block(side, j, x0, x0 + 4 * nlanes, nx, p.mx, &st, &kd);
            ekind[j - 1] = kd;
            if (is_edge_lane && kd != EDGE_CONST) evoff[j - 1] = (rowbase + (unsigned)st) * 4u;
        }
    }

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nA);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - p.oa;

    struct Slot { float4 v; float4 e[NB > 0 ? NB : 1]; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)nA) ai = bmap<int>(ai, nA, p.ma);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * strideA * 4u;
        s.v = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0));
        if constexpr (!(VAR & V_NOEDGE)) {
#pragma unroll
            for (int j = 0; j < NB; j++)
                s.e[j] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : evoff[j], soff, 0));
        }
        if constexpr (VAR & V_WAIT) asm volatile("s_waitcnt vmcnt(0)\n s_nop 7" ::: "memory");
    };

    F4 ring[RINGN > 0 ? RINGN : 1];
#pragma unroll
    for (int k = 0; k < (RINGN > 0 ? RINGN : 1); k++) ring[k] = f4_splat(0.f);

#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    const float4 cv4 = make_float4(p.cval, p.cval, p.cval, p.cval);
    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                float4 v = s.cst ? cv4 : s.v;
                float4 eL[4] = {cv4, cv4, cv4, cv4}, eR[4] = {cv4, cv4, cv4, cv4};
                if constexpr (!(VAR & V_NOEDGE)) {
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const float4 t = s.cst ? cv4 : apply_kind<VAR>(s.e[j], ekind[j], side, p.cval);
                        eL[j] = t;
                        eR[j] = t;
                    }
                }
                const F4 xf = xpass_hops<WX, OP, VAR>(v, eL, eR, lane, last);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                if (i >= WA - 1) {
                    F4 a;
                    if constexpr (WA == 1) {
                        a = xf;
                    } else {
                        a = ring[J % RINGN];
#pragma unroll
                        for (int k = 1; k < RINGN; k++) a = f4_mm<OP>(ring[(J + k) % RINGN], a);
                        a = f4_mm<OP>(xf, a);
                    }
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * strideA * 4u;
                    if constexpr (VAR & V_STVOFF) __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(a), rout, voff + so, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(a), rout, voff, so, 0);
                    if constexpr (VAR & V_ST0) asm volatile("s_nop 0" ::: "memory");
                    if constexpr (VAR & V_ST1) asm volatile("s_nop 1" ::: "memory");
                    if constexpr (VAR & V_ST3) asm volatile("s_nop 3" ::: "memory");
                    if constexpr (VAR & V_ST7) asm volatile("s_nop 7" ::: "memory");
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = xf;
            }
        });
    }
}

template <int WX, int WA, int VAR>
__global__ void __launch_bounds__(256) k_plain(const float *__restrict__ in, float *__restrict__ out, const StreamParams p)
{
    body<WX, WA, 2, SP_MIN, VAR>(in, out, p);
}
template <int WX, int WA, int VAR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_occ1(const float *__restrict__ in, float *__restrict__ out, const StreamParams p)
{
    body<WX, WA, 2, SP_MIN, VAR>(in, out, p);
}
// dynamic LDS request limits blocks per CU
template <int WX, int WA, int VAR>
__global__ void __launch_bounds__(256) k_lds(const float *__restrict__ in, float *__restrict__ out, const StreamParams p)
{
    extern __shared__ float dummy[];
    if (p.nx < 0) dummy[threadIdx.x] = 1.f;
    body<WX, WA, 2, SP_MIN, VAR>(in, out, p);
}
}  // namespace mi

using namespace mi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static int NZ = 256, NY = 256, NX = 256;

template <typename K>
static void launch(K kern, const float *in, float *out, int slice, int wpb, size_t lds, int axis = 0)
{
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.nx = NX; p.ny = NY; p.nz = NZ;
    p.axis = axis; p.wa = 3; p.oa = 1; p.ma = MI_MODE_MIRROR; p.mx = MI_MODE_MIRROR;
    p.nxt = (NX + 255) / 256;
    const int nA = axis == 0 ? NZ : NY, nother = axis == 0 ? NY : NZ;
    p.chunk = 32;
    p.nchunks = (nA + p.chunk - 1) / p.chunk;
    const int waves = nother * p.nxt * p.nchunks;
    if (slice <= 0) slice = waves;
    p.wpb = wpb;
    for (int base = 0; base < waves; base += slice) {
        p.wid_base = base;
        const int n = std::min(slice, waves - base);
        hipLaunchKernelGGL(kern, dim3((n + wpb - 1) / wpb), dim3(64 * wpb), lds, 0, in, out, p);
        CK(hipGetLastError());
        if (slice < waves) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
}

static std::vector<float> ref;
static void report(const char *tag, const float *dout, size_t n)
{
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), dout, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    std::set<int> lanes, comps;
    for (size_t i = 0; i < n; i++)
        if (memcmp(&h[i], &ref[i], 4)) {
            bad++;
            const int x = (int)(i % NX);
            lanes.insert((x % 256) / 4 % 16);
            comps.insert(x % 4);
        }
    printf("%-40s mismatches %zu  lanes%%16 {", tag, bad);
    for (int l : lanes) printf("%d ", l);
    printf("} comps {");
    for (int c : comps) printf("%d ", c);
    printf("}\n");
    fflush(stdout);
}

int main()
{
    const size_t n = (size_t)NZ * NY * NX;
    std::vector<float> h(n);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    CK(hipMalloc(&din, n * 4));
    CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
    // reference: sliced baseline
    launch(k_plain<3, 3, 0>, din, dout, 1000, 4, 0);
    ref.resize(n);
    CK(hipMemcpy(ref.data(), dout, n * 4, hipMemcpyDeviceToHost));
    // host check of the reference on a sample of voxels (mirror mode, window z-1..z+1, x-1..x+1)
    {
        size_t bad = 0;
        for (size_t t = 0; t < 2000000; t++) {
            const size_t i = (t * 2654435761ull) % n;
            const int x = (int)(i % NX), y = (int)(i / NX % NY), z = (int)(i / ((size_t)NX * NY));
            float m = 1e30f;
            for (int dz = -1; dz <= 1; dz++)
                for (int dx = -1; dx <= 1; dx++) {
                    int zz = z + dz, xx = x + dx;
                    if (zz < 0) zz = -zz; if (zz >= NZ) zz = 2 * NZ - 2 - zz;
                    if (xx < 0) xx = -xx; if (xx >= NX) xx = 2 * NX - 2 - xx;
                    m = std::min(m, h[((size_t)zz * NY + y) * NX + xx]);
                }
            if (m != ref[i]) bad++;
        }
        printf("reference (sliced) vs host on 2M samples: %zu wrong\n", bad);
    }
#define RUN(tag, kern, slice, wpb, lds) do { CK(hipMemset(dout, 0xff, n * 4)); launch(kern, din, dout, slice, wpb, lds); report(tag, dout, n); } while (0)
    RUN("baseline unsliced", (k_plain<3, 3, 0>), 0, 4, 0);
    RUN("baseline unsliced (again)", (k_plain<3, 3, 0>), 0, 4, 0);
    RUN("baseline slice 2000", (k_plain<3, 3, 0>), 2000, 4, 0);
    RUN("s_nop around dpp", (k_plain<3, 3, V_NOP>), 0, 4, 0);
    RUN("bpermute", (k_plain<3, 3, V_BPERM>), 0, 4, 0);
    RUN("branch-free apply_kind", (k_plain<3, 3, V_NOBRANCH>), 0, 4, 0);
    RUN("no edge loads", (k_plain<3, 3, V_NOEDGE>), 0, 4, 0);
    RUN("wait after issue", (k_plain<3, 3, V_WAIT>), 0, 4, 0);
    RUN("nobranch+nop", (k_plain<3, 3, V_NOBRANCH | V_NOP>), 0, 4, 0);
    RUN("store + s_nop 0", (k_plain<3, 3, V_ST0>), 0, 4, 0);
    RUN("store + s_nop 1", (k_plain<3, 3, V_ST1>), 0, 4, 0);
    RUN("store + s_nop 3", (k_plain<3, 3, V_ST3>), 0, 4, 0);
    RUN("store + s_nop 7", (k_plain<3, 3, V_ST7>), 0, 4, 0);
    RUN("store + s_nop 7+3+1+0", (k_plain<3, 3, V_ST7 | V_ST3 | V_ST1 | V_ST0>), 0, 4, 0);
    RUN("store voffset only (soffset 0)", (k_plain<3, 3, V_STVOFF>), 0, 4, 0);
    RUN("waves_per_eu(1,1)", (k_occ1<3, 3, 0>), 0, 4, 0);
    CK(hipFuncSetAttribute((const void *)k_lds<3, 3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RUN("lds 160K (1 block/CU)", (k_lds<3, 3, 0>), 0, 4, 160 * 1024);
    RUN("lds 80K (2 blocks/CU)", (k_lds<3, 3, 0>), 0, 4, 80 * 1024);
    RUN("lds 40K (4 blocks/CU)", (k_lds<3, 3, 0>), 0, 4, 40 * 1024);
    RUN("wpb 1 unsliced", (k_plain<3, 3, 0>), 0, 1, 0);
    RUN("x only <3,1>", (k_plain<3, 1, 0>), 0, 4, 0);
    return 0;
}
