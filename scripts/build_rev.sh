#!/bin/bash
# build_rev.sh <git-rev> -- builds libggml-hip-cdna4.so of ANOTHER revision of this repo into ik_llama.cpp_amd/build/ab_<rev>/ (git-ignored, travels to the
# GPU box with the gpurun snapshot) for `python bench.py --ab-lib <path>`: both builds are then timed in ONE process on ONE box.
# The GPU box has no .git, so this runs in the build container.
set -e
rev=$(git rev-parse --short=12 "$1")
root=$(git rev-parse --show-toplevel)
dst=$root/ik_llama.cpp_amd/build/ab_$rev
mkdir -p "$dst/src"
git -C "$root" archive "$rev" ik_llama.cpp_amd/csrc ik_llama.cpp_amd/build.py include | tar -x -C "$dst/src"
python "$dst/src/ik_llama.cpp_amd/build.py" > "$dst/build.log" 2>&1
cp "$dst/src/ik_llama.cpp_amd/libggml-hip-cdna4.so" "$dst/libggml-hip-cdna4.so"
rm -rf "$dst/src"
echo "$dst/libggml-hip-cdna4.so"
