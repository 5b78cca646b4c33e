"""Instruction histogram of the kernels of one object in cupyimg_amd/csrc/build whose mangled name holds a fragment (CPU; what
showed the scalar-spill traffic of the 5^3 tap loop):  python scripts/isa_hist.py stencil3s.o stencil3s_kernelILi5Ef [top]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cupyimg_amd import _build
obj, frag = os.path.join(_build.OBJ, sys.argv[1]), sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
with tempfile.TemporaryDirectory() as tmp:
    co = os.path.join(tmp, "dev.co")
    open(co, "wb").write(_build._device_code_object(obj))
    txt = subprocess.run([_build._llvm_tool("llvm-objdump"), "-d", co], stdout=subprocess.PIPE, text=True, check=True).stdout
for part in re.split(r"\n(?=[0-9a-f]+ <)", txt):
    head = part.split("\n", 1)[0]
    if frag not in head:
        continue
    lines = part.split("\n")[1:]
    c = collections.Counter(l.strip().split()[0] for l in lines if l.strip())
    print(head, len(lines), "instructions")
    print("   " + "  ".join("%s %d" % kv for kv in c.most_common(top)))
    if os.environ.get("DUMP"):
        open(os.environ["DUMP"], "w").write(part)
