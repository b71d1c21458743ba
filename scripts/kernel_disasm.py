"""Disassembly of one kernel of libazmi.so (gfx950 code object): python scripts/kernel_disasm.py <mangled-name substring> > out.s"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import SO, code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    want = sys.argv[1]
    blob = open(SO, "rb").read()
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            syms = subprocess.run([OBJDUMP, "-t", f.name], capture_output=True, text=True).stdout
            names = [l.split()[-1] for l in syms.splitlines() if want in l and " F " in l and ".text" in l]
            for nm in names:
                out = subprocess.run([OBJDUMP, "-d", "--disassemble-symbols=" + nm, f.name], capture_output=True, text=True).stdout
                print(out)


if __name__ == "__main__":
    main()
