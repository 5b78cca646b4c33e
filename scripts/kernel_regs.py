"""Registers / scratch / LDS of every kernel in an object of cupyimg_amd/csrc/build (code-object metadata note).
    python scripts/kernel_regs.py bitmorph3d.o [name-fragment]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cupyimg_amd import _build
obj = os.path.join(_build.OBJ, sys.argv[1])
frag = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    co = os.path.join(tmp, "dev.co")
    open(co, "wb").write(_build._device_code_object(obj))
    text = subprocess.run([_build._llvm_tool("llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    if os.environ.get("DISASM"):
        subprocess.run([_build._llvm_tool("llvm-objdump"), "-d", co], stdout=open(os.environ["DISASM"], "w"))
for block in re.split(r"\n\s+- \.agpr_count:", text)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", block) or [None, "?"])[1]
    if frag in g("name"):
        name = subprocess.run(["c++filt", g("name")], stdout=subprocess.PIPE, text=True).stdout.strip()
        print("vgpr {:>4} sgpr {:>4} scratch {:>5} lds {:>6} spill {:>3}  {}".format(g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count"), name[:110]))
