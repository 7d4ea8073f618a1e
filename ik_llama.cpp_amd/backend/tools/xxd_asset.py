#!/usr/bin/env python3
"""What the reference's scripts/xxd.cmake does for the server's web assets (`xxd -i`): <name>.hpp with `unsigned char <name>[] = {...}; unsigned int <name>_len`.
usage: xxd_asset.py <input> <output.hpp>     (Makefile.llama, llama-server target; generated files live under oracle/_ref/llama/obj/server/)"""
import os, re, sys
src, dst = sys.argv[1], sys.argv[2]
name = re.sub(r"[.\-]", "_", os.path.basename(src)); data = open(src, "rb").read()
with open(dst, "w") as f:
    f.write("unsigned char %s[] = {%s};\nunsigned int %s_len = %d;\n" % (name, "".join("0x%02x," % b for b in data), name, len(data)))
