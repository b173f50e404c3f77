#!/usr/bin/env python3
"""Disassembles the gfx950 code object(s) inside a built file (libpgi.so or build/*.o) and prints the kernels whose
mangled name contains a substring.  CPU only.  Usage: disasm_kernel.py <file> <substring> [out.s]"""
import os, re, struct, subprocess, sys, tempfile
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
data = open(sys.argv[1], "rb").read()
want = sys.argv[2]
out = []
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
            txt = subprocess.run([OBJDUMP, "-d", fn], capture_output=True, text=True).stdout
            keep = False
            for line in txt.splitlines():
                mm = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if mm:
                    keep = want in mm.group(1)
                if keep:
                    out.append(line)
text = "\n".join(out) + "\n"
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text)
else:
    sys.stdout.write(text)
