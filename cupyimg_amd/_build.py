"""Builds libmi355img.so (HIP kernels + C-ABI) for gfx950 with hipcc.

    python -m cupyimg_amd._build [--force] [--jobs N]

The library is built in-tree (cupyimg_amd/libmi355img.so) so that it travels
with the source snapshot to the GPU box; objects go to cupyimg_amd/csrc/build/.
hipcc cross-compiles for gfx950 without a GPU being present.
"""
import argparse
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmi355img.so")
# the same library with every hand-counted `s_waitcnt vmcnt(n)` turned into vmcnt(0) (csrc/common.hpp MI_VMCNT): a TEST
# artefact -- tests/test_gpu_burst.py compares the two under load, bit for bit; nothing in the product loads it
LIB_STRICT = os.path.join(HERE, "libmi355img_strict.so")
STRICT_SOURCES = ("sep3d_long.hip", "minmax3d_f32.hip", "interp_fast.hip", "interp.hip", "cubic_fast.hip")     # the files that use MI_VMCNT
ARCH = "gfx950"

# (source, extra flags).  The generic kernels are built without FMA contraction
# so that double results are bit-identical to SciPy's C code; the roofline
# kernels use the default (fast) contraction.
SOURCES = [
    ("runtime.hip", []),
    ("copy.hip", []),
    ("synth.hip", []),
    ("correlate1d.hip", ["-ffp-contract=off"]),
    ("separable3d.hip", []),
    ("stream3d.hip", []),
    ("stream_f64.hip", []),
    # MI_LONG_TUNE=1 in the environment adds the tuning variants of the 17-tap kernel (mi_debug_set_long_cfg)
    ("sep3d_long.hip", ["-DMI_LONG_TUNE"] if os.environ.get("MI_LONG_TUNE") else []),
    ("minmax3d_f32.hip", []),
    ("correlate_nd.hip", ["-ffp-contract=off"]),
    ("stencil3d.hip", ["-ffp-contract=off"]),
    ("stencil3s.hip", ["-ffp-contract=off"]),
    ("minmax.hip", ["-ffp-contract=off"]),
    ("median3d.hip", []),
    ("median3d_u8.hip", []),
    ("median3d_16.hip", []),
    ("median3d_i.hip", []),
    ("median3d_f64.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p16.hip", ["-ffp-contract=off"]),
    ("rank_sorted_med.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p32a.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p32b.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p64a.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p64b.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p64c.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p64d.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p128a.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p128b.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p128c.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p128d.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p128e.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p128f.hip", ["-ffp-contract=off"]),
    ("rank_sorted_p128g.hip", ["-ffp-contract=off"]),
    ("minmax3d_u8.hip", []),
    ("minmax3d_u8r.hip", []),
    ("minmax3d_16r.hip", []),
    ("median2d.hip", []),
    ("minmax_16.hip", []),
    ("binary.hip", []),
    ("binary3d.hip", []),
    ("bitmorph3d.hip", []),
    ("interp.hip", ["-ffp-contract=off"]),
    ("interp_fast.hip", ["-ffp-contract=off"]),
    ("spline_fast.hip", []),
    ("cubic_fast.hip", ["-ffp-contract=off"]),
    ("halo.hip", []),
    ("metrics.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(os.path.dirname(HERE), "include")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmi355img.so cannot be built")
    return exe


def _deps_mtime():
    """Newest header of the tree: the fallback dependency of an object whose own dependency file is missing."""
    newest = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for name in os.listdir(root):
            if name.endswith((".hpp", ".h")):
                newest = max(newest, os.path.getmtime(os.path.join(root, name)))
    return newest


def _obj_deps_mtime(depfile, fallback):
    """Newest in-tree header the object was built from (hipcc -MMD writes them to <object>.d), so that editing one
    kernel's header does not rebuild the whole library (a full build is minutes)."""
    try:
        text = open(depfile).read().replace("\\\n", " ")
    except OSError:
        return fallback
    newest = 0.0
    for tok in text.split()[1:]:
        if tok.endswith((".hpp", ".h")) and not tok.startswith("/opt/"):
            try:
                newest = max(newest, os.path.getmtime(os.path.join(CSRC, tok)))      # absolute tokens stay as they are
            except OSError:
                return fallback          # a header was renamed / removed: rebuild
    return newest


def _llvm_tool(name):
    """llvm-objdump / llvm-objcopy of the ROCm install hipcc belongs to (not a fixed /opt/rocm)."""
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc())))
    for cand in (os.path.join(root, "lib", "llvm", "bin", name), os.path.join(root, "llvm", "bin", name),
                 os.path.join("/opt/rocm/lib/llvm/bin", name), shutil.which(name) or ""):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("{} not found next to {}: the scratch-spill check of the vmcnt-counting kernels cannot run"
                       .format(name, hipcc()))


def _flag_stamp(flags):
    import hashlib
    return hashlib.sha256(" ".join(COMMON + list(flags)).encode()).hexdigest()


def _compile(args):
    src, flags, force, hdr_mtime = args[:4]
    strict = len(args) > 4 and args[4]
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace(".hip", ".strict.o" if strict else ".o"))
    if strict:
        flags = list(flags) + ["-DMI_STRICT_WAITS"]
    d = o + ".d"
    stamp_file = o + ".flags"
    stamp = _flag_stamp(flags)
    try:
        same_flags = open(stamp_file).read() == stamp       # e.g. MI_LONG_TUNE toggled: rebuild
    except OSError:
        same_flags = False
    if (not force and same_flags and os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(s)
            and os.path.getmtime(o) >= _obj_deps_mtime(d, hdr_mtime)):
        return o, False
    if os.path.exists(stamp_file):
        os.unlink(stamp_file)
    cmd = [hipcc()] + COMMON + flags + ["-MMD", "-MF", d, "-c", s, "-o", o]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed for {}:\n{}".format(src, proc.stdout))
    if src in NO_SCRATCH:
        try:
            spilled = _scratch_users(o, NO_SCRATCH[src])
        except BaseException:
            os.unlink(o)                 # an unchecked object must not look up to date to the next build
            raise
        if spilled:
            os.unlink(o)
            raise RuntimeError("{}: kernels that count their vector-memory operations by hand (s_waitcnt vmcnt) were "
                               "compiled with scratch-memory accesses, which add uncounted ones: {}".format(src, spilled))
    if src in REG_BUDGET:
        try:
            over = _over_budget(o, REG_BUDGET[src])
        except BaseException:
            os.unlink(o)
            raise
        if over:
            os.unlink(o)
            raise RuntimeError("{}: kernels planned for a fixed number of waves per SIMD need more registers than that "
                               "allows: {}".format(src, over))
    with open(stamp_file, "w") as f:     # written last: object compiled AND checked with these flags
        f.write(stamp)
    if proc.stdout.strip():
        sys.stderr.write(proc.stdout)
    return o, True


# Sources whose kernels wait on vmcnt by count (LDS-DMA rings): the name fragment selects the kernels that must not
# touch scratch memory (a spill store or reload is one more vector-memory operation in flight than the count assumes).
# The ablation builds of the r3 long kernel (<W, SAME, DBG = true, 0>) are exempt: timing aids, not product kernels.
NO_SCRATCH = {"sep3d_long.hip": "sep3d_long", "minmax3d_f32.hip": "mm3f32_long", "interp_fast.hip": ("zstream_kernel", "zrect_kernel"), "interp.hip": "cubic3_zstream_kernel", "cubic_fast.hip": "cubic3_zfactor_kernel"}


# Kernels whose occupancy is part of their design: (fragment of the mangled name, VGPRs + AGPRs per lane at most).
# map_coords3d_zstream_kernel<true, 4, *>: 512 threads, two workgroups per CU = four waves per SIMD = 128 registers.
REG_BUDGET = {"interp_fast.hip": [("map_coords3d_zstream_kernelILb1ELi4E", 128)]}


def _over_budget(obj, budgets):
    """[(kernel, registers)] of the kernels in `obj` that exceed their entry in `budgets` (from the code object's metadata
    note: .vgpr_count counts the accumulation registers too on gfx90a+)."""
    import tempfile
    readelf = _llvm_tool("llvm-readelf")
    with tempfile.TemporaryDirectory() as tmp:
        co = os.path.join(tmp, "dev.co")
        with open(co, "wb") as f:
            f.write(_device_code_object(obj))
        text = subprocess.run([readelf, "--notes", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    over, seen = [], set()
    # the note is YAML; the fields of one kernel sit between two ".agpr_count" lines (first key of every entry)
    for block in re.split(r"\n\s+- \.agpr_count:", text)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        regs = re.search(r"\.vgpr_count:\s+(\d+)", block)
        if not name or not regs:
            continue
        for fragment, limit in budgets:
            if fragment in name.group(1):
                seen.add(fragment)
                if int(regs.group(1)) > limit:
                    over.append("{} ({} > {})".format(name.group(1), regs.group(1), limit))
    missing = [f for f, _ in budgets if f not in seen]
    if missing:
        raise RuntimeError("register budget check: no kernel matches {} in {}".format(missing, obj))
    return over


def _device_code_object(obj):
    """The gfx950 code object out of a hipcc host object (section .hip_fatbin, clang offload bundle)."""
    import struct
    import tempfile
    objcopy = _llvm_tool("llvm-objcopy")
    with tempfile.TemporaryDirectory() as tmp:
        fb = os.path.join(tmp, "fb.bin")
        subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fb, obj], check=True)
        blob = open(fb, "rb").read()
    if blob[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
        raise RuntimeError("unexpected fat binary layout in " + obj)
    n = struct.unpack_from("<Q", blob, 24)[0]
    off = 32
    for _ in range(n):
        o, sz, ts = struct.unpack_from("<QQQ", blob, off)
        off += 24
        triple = blob[off:off + ts].decode()
        off += ts
        if ARCH in triple:
            return blob[o:o + sz]
    raise RuntimeError("no {} code object in {}".format(ARCH, obj))


def _is_ablation_build(mangled):
    """sep3d_long3_kernel<W, SAME, DBG = true, ...>: the ablation builds (timing aids behind a debug knob, not product
    kernels) may spill.  Decided from the template arguments of the Itanium-mangled name -- `Lb1E` in third position --
    not from one exact suffix, so that a new trailing parameter or a renamed parameter struct cannot silently turn the
    exemption off (or on for a product kernel)."""
    m = re.search(r"sep3d_long[34]_kernelI((?:L[a-z]n?\d+E)+)E", mangled)
    if not m:
        return False
    targs = re.findall(r"L([a-z])(n?\d+)E", m.group(1))
    return len(targs) >= 3 and targs[2] == ("b", "1")


def _scratch_users(obj, fragment):
    """Kernels of `obj` whose name contains `fragment` and whose code has scratch_* instructions."""
    import tempfile
    objdump = _llvm_tool("llvm-objdump")
    with tempfile.TemporaryDirectory() as tmp:
        co = os.path.join(tmp, "dev.co")
        with open(co, "wb") as f:
            f.write(_device_code_object(obj))
        text = subprocess.run([objdump, "-d", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    bad, name, count = [], None, 0
    def close():
        frags = fragment if isinstance(fragment, tuple) else (fragment,)
        if name and count and any(f in name for f in frags) and not _is_ablation_build(name):
            bad.append("{} ({} scratch / compiler-made AGPR instructions)".format(name, count))
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            close()
            name, count = m.group(1), 0
        elif "\tscratch_" in line or " scratch_" in line:
            count += 1
        elif name and "map_coords3d_zstream" in name and re.search(r"\bv_accvgpr_(write|mov)", line):
            # these kernels keep two planes of coordinates IN FLIGHT in a[0:50] (global_load -> AGPR, ds_write_b128 <- AGPR,
            # hand-counted waits); the compiler must not put anything of its own there (a VGPR spilled into an AGPR
            # could sit under a landing load): any AGPR write or move of the compiler's is refused like a scratch access
            count += 1
    close()
    return bad


def build(force=False, jobs=None, verbose=True, strict=False):
    """Compile and link under an exclusive file lock: N ranks importing the
    package at once (torchrun) build once, the others wait and find the library
    up to date.  The link goes to a temporary name and is moved into place, so a
    concurrent dlopen never sees a half-written file.  strict=True builds
    LIB_STRICT (the vmcnt(0) twin, a test artefact) from the same objects
    except the STRICT_SOURCES, which are compiled again with -DMI_STRICT_WAITS."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, jobs, verbose, strict)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


# sources that take minutes each: compiled first, so that a from-scratch build does not end on one of them alone
_HEAVY = ("rank_sorted_p128", "sep3d_long", "interp.hip", "cubic_fast", "interp_fast", "rank_sorted_p64", "minmax3d", "separable3d", "rank_sorted_p32",
          "median3d")


def _weight(src):
    for n, frag in enumerate(_HEAVY):
        if frag in src:
            return n
    return len(_HEAVY)


def _link(LIB, objs, results, verbose):
    rebuilt = any(r for _, r in results) or (os.path.exists(LIB) and any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs))
    if rebuilt or not os.path.exists(LIB):
        tmp = "{}.{}.tmp".format(LIB, os.getpid())
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs + [
            "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed:\n" + proc.stdout)
        os.replace(tmp, LIB)
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


def _build_locked(force, jobs, verbose, strict=False):
    """strict: False = the product library, True = the strict twin, "both" = both from ONE pool of compile jobs (the twin's five
    objects start while the product's long compilations still run: what __graft_entry__.build() uses)"""
    present = [(s, f) for s, f in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    missing = [s for s, _ in SOURCES if not os.path.exists(os.path.join(CSRC, s))]
    if missing:
        raise RuntimeError("missing kernel sources: {}".format(missing))
    hdr = _deps_mtime()
    jobs = jobs or min(8, os.cpu_count() or 1)
    want_product = strict in (False, "both")
    want_strict = strict in (True, "both")
    tasks = [(s, f, force, hdr, False) for s, f in present]            # the twin links the product's objects for everything else
    if want_strict:
        tasks += [(s, f, force, hdr, True) for s, f in present if s in STRICT_SOURCES]
    order = sorted(range(len(tasks)), key=lambda i: (_weight(tasks[i][0]), i))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        done = list(ex.map(_compile, [tasks[i] for i in order]))
    results = [None] * len(tasks)
    for i, r in zip(order, done):
        results[i] = r
    plain = {tasks[i][0]: results[i] for i in range(len(present))}
    twin = {tasks[i][0]: results[i] for i in range(len(present), len(tasks))}
    lib = None
    if want_product:
        res = [plain[s] for s, _ in present]
        lib = _link(globals()["LIB"], [o for o, _ in res], res, verbose)
    if want_strict:
        res = [twin.get(s, plain[s]) for s, _ in present]
        lib_s = _link(LIB_STRICT, [o for o, _ in res], res, verbose)
        lib = lib if want_product else lib_s
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--strict", action="store_true", help="build libmi355img_strict.so (every counted wait = vmcnt(0)) too")
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs)
    if a.strict:
        build(force=a.force, jobs=a.jobs, strict=True)
