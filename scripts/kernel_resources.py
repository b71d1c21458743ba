"""Register / scratch / LDS budget of every kernel in libazmi.so, read from the code objects' metadata (what the hardware gets):
    python scripts/kernel_resources.py [filter ...] > profiles/r5_kernel_resources.txt
libazmi.so carries one clang offload bundle per translation unit in its .hip_fatbin section; each bundle's gfx950 entry is an ELF
whose notes (llvm-readelf --notes) list, per kernel: .vgpr_count, .agpr_count, .sgpr_count, .vgpr_spill_count, .sgpr_spill_count,
.private_segment_fixed_size (scratch bytes per lane), .group_segment_fixed_size (static LDS)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "alphazero-pybind11_amd", "libazmi.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernels(co):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    out = []
    for block in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        block = ".agpr_count:" + block
        rec = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                    "group_segment_fixed_size", "max_flat_workgroup_size"):
            m = re.search(r"\.%s:\s+(\d+)" % key, block)
            rec[key] = int(m.group(1)) if m else -1
        m = re.search(r"\.name:\s+(\S+)", block)
        rec["name"] = m.group(1).strip("'\"") if m else "?"
        out.append(rec)
    return out


def main():
    filters = sys.argv[1:]
    blob = open(SO, "rb").read()
    rows = []
    for co in code_objects(blob):
        rows += kernels(co)
    names = subprocess.run([CXXFILT] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    print("# libazmi.so, gfx950 code objects: vgpr (of which agpr) | sgpr | vgpr spills | sgpr spills | scratch B/lane | static LDS | threads | kernel")
    for r, nm in sorted(zip(rows, names), key=lambda x: x[1]):
        nm = re.sub(r"\(.*$", "", nm)
        if filters and not any(f in nm for f in filters):
            continue
        print("%4d (%3d) | %3d | %3d | %3d | %5d | %6d | %4d | %s" % (r["vgpr_count"], r["agpr_count"], r["sgpr_count"], r["vgpr_spill_count"],
              r["sgpr_spill_count"], r["private_segment_fixed_size"], r["group_segment_fixed_size"], r["max_flat_workgroup_size"], nm[:150]))


if __name__ == "__main__":
    main()
