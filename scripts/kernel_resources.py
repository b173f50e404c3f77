#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel, read from the gfx950 code objects inside libpgi.so (the AMDGPU
metadata notes the compiler wrote) -- the authoritative numbers; rocprofv3's kernel-trace VGPR column is in different
units.  CPU only.  Usage: kernel_resources.py [out.txt]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("PGI_LIB", os.path.join(ROOT, "pose-graph-initialization_amd", "libpgi.so"))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def main():
    data = open(LIB, "rb").read()
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data):
            b = m.start()
            n = struct.unpack_from("<Q", data, b + 24)[0]
            pos = b + 32
            for e in range(n):
                off, size, tl = struct.unpack_from("<QQQ", data, pos)
                pos += 24
                triple = data[pos:pos + tl].decode()
                pos += tl
                if "gfx950" not in triple or not size:
                    continue
                fn = os.path.join(d, "co_%d_%d.o" % (b, e))
                open(fn, "wb").write(data[b + off:b + off + size])
                txt = subprocess.run([READELF, "--notes", fn], capture_output=True, text=True).stdout
                cur = {}
                for line in txt.splitlines():
                    mm = re.match(r"\s*-?\s*\.(\w+):\s+(.*)", line)
                    if not mm:
                        continue
                    k, v = mm.group(1), mm.group(2).strip()
                    if k == "agpr_count" and cur.get("name"):
                        rows.append(cur)
                        cur = {}
                    if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                             "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                        cur[k] = v
                if cur.get("name"):
                    rows.append(cur)
    names = subprocess.run([FILT] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    out = ["kernel resources from the gfx950 code objects of libpgi.so (llvm-readelf --notes)",
           "%-78s %5s %5s %5s %9s %9s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_B", "lds_B")]
    for r, nm in sorted(zip(rows, names), key=lambda t: t[1]):
        out.append("%-78s %5s %5s %5s %9s %9s %8s %8s" % (nm[:78], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"),
                                                            r.get("vgpr_spill_count"), r.get("sgpr_spill_count"),
                                                            r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
