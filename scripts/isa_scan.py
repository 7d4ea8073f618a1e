"""Static ISA scan of the library's translation units (runs in the CPU-only container: hipcc -S cross-compiles): per kernel, the number of quarter-rate integer multiplies
(v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32: what 64-bit `index * stride` address arithmetic turns into), of `s_waitcnt vmcnt(0)` (a full drain of the memory pipe: each one in a
loop or in front of a load is a serialized round trip -- vmcnt is in-order and counts stores) and of instructions.  The two patterns behind round 3's largest wins and behind the
findings of profiles/r03_notes.md section 12.

    python scripts/isa_scan.py ops.hip gemv_dual.hip [--top 12] [-D...]          (sources under ik_llama.cpp_amd/csrc; extra -D / -I flags are passed to hipcc)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ik_llama.cpp_amd", "csrc")
QMUL = ("v_mul_lo_u32", "v_mad_u64_u32", "v_mul_hi_u32", "v_mad_i64_i32", "v_mul_hi_i32")


def scan(asm):
    name, stats = None, {}
    for line in open(asm, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); stats[name] = [0, 0, 0]; continue
        t = line.strip()
        if name is None or not t or t[0] in ";.":
            continue
        op = t.split()[0]
        stats[name][0] += op in QMUL
        stats[name][1] += t.startswith("s_waitcnt vmcnt(0)")
        stats[name][2] += bool(re.match(r"^(v|s|global|ds|buffer|flat)_", op))
        if op == "s_endpgm":
            name = None
    return stats


def main():
    args = sys.argv[1:]; top = 12
    if "--top" in args:
        i = args.index("--top"); top = int(args[i + 1]); del args[i:i + 2]
    flags = [a for a in args if a.startswith("-")]; files = [a for a in args if not a.startswith("-")]
    hipcc = "/opt/rocm/bin/hipcc"
    for f in files:
        src = f if os.path.exists(f) else os.path.join(CSRC, f)
        with tempfile.TemporaryDirectory() as tmp:
            asm = os.path.join(tmp, "out.s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I/opt/rocm/include", "-fno-slp-vectorize", "-S", "--cuda-device-only"] + flags + [src, "-o", asm],
                                  stderr=subprocess.DEVNULL)
            stats = scan(asm)
        import shutil
        filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
        demangle = subprocess.run([filt], input="\n".join(stats), capture_output=True, text=True).stdout.split("\n") if filt else list(stats)
        names = dict(zip(stats, demangle))
        print("== %s: %d kernels" % (os.path.basename(src), len(stats)))
        for k, v in sorted(stats.items(), key=lambda kv: (-kv[1][0], -kv[1][1]))[:top]:
            print("%5d quarter-rate muls  %4d vmcnt(0)  %6d instructions  %s" % (v[0], v[1], v[2], names.get(k, k)[:110]))


if __name__ == "__main__":
    main()
