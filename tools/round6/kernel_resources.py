#!/usr/bin/env python3
"""Registers / LDS / scratch of the gfx950 kernels inside a hipcc object or shared library (the clang offload bundle it embeds):
python tools/round6/kernel_resources.py egopack_amd/csrc/build/gemm.o [name filter ...]"""
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    at = 0
    while True:
        at = data.find(MAGIC, at)
        if at < 0:
            return
        n, = struct.unpack_from("<Q", data, at + 24)
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield data[at + off:at + off + size]
        at += 24


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
            if filters and not all(f in name for f in filters):
                continue
            print(f"vgpr {get('vgpr_count'):>4} agpr {blk.split()[0]:>4} sgpr {get('sgpr_count'):>4} lds {get('group_segment_fixed_size'):>7} "
                  f"scratch {get('private_segment_fixed_size'):>5}  {name[:140]}")


if __name__ == "__main__":
    main()
