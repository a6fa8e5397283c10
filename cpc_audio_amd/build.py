"""Build libcpc_hip.so for gfx950 with hipcc (in-tree, no torch C++ extension: the ABI is
plain C, so the library is immune to the torch-ROCm / system-ROCm version skew).

    python -m cpc_audio_amd.build            # incremental
    python -m cpc_audio_amd.build --force
"""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcpc_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wno-unused-result", "-fno-slp-vectorize"]


# Per-file flags (none at present).  -fno-slp-vectorize is global (FLAGS) since round 3: with compiler-formed v_pk_fma_f32 whose
# operands come from LDS reads, conv0_bwd_kernel and later conv0_fwd_kernel returned wrong bits whenever a 16-bit-MFMA GEMM
# kernel shared the chip with them (tools/probe_corun.py, DESIGN.md section 4.6: root cause open, compiler or hardware).
# Rather than trusting that no other kernel ever meets the pattern, the library is built without compiler-formed packed fp32
# altogether, and check_packed_fp32() below fails the build if a kernel outside PACKED_FP32_ALLOWED contains any.
FILE_FLAGS = {}

# Kernels that may contain v_pk_{fma,mul,add}_f32: NONE since round 4.  Until then three recurrence kernels carried explicit f32x4
# arithmetic on MFMA accumulators (36 packed adds / multiplies in all, never seen to misbehave beside the 16-bit-MFMA GEMMs in
# tests/test_gpu_corun.py); they are written component by component now (gru.hip: add4 / scale4, no measurable cost), so the
# library as a whole is free of the instruction class and the gate below has nothing to excuse.
PACKED_FP32_ALLOWED = ()


def _llvm(tool):
    for d in (os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib", "llvm", "bin"),
              "/opt/rocm/lib/llvm/bin"):
        if os.path.exists(os.path.join(d, tool)):
            return os.path.join(d, tool)
    raise RuntimeError(f"{tool} not found (needed for the packed-fp32 gate)")


def packed_fp32_kernels(obj):
    """{mangled kernel symbol: number of v_pk_{fma,mul,add}_f32} for the gfx950 code object inside a hipcc object file."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        r = subprocess.run([_llvm("llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True, text=True)
        if r.returncode != 0:
            return {}                                     # host-only translation unit
        r = subprocess.run([_llvm("clang-offload-bundler"), "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}",
                            f"--input={fat}", f"--output={co}", "--unbundle"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"cannot extract the {ARCH} code object of {obj}:\n{r.stderr}")
        dis = subprocess.run([_llvm("llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
        elif cur and re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
            out[cur] = out.get(cur, 0) + 1
    return out


def check_packed_fp32(objs):
    """The build-time gate: no packed fp32 arithmetic outside the allow-list (see FILE_FLAGS above)."""
    ok = tuple(f"{len(n)}{n}E" for n in PACKED_FP32_ALLOWED)          # Itanium-mangled name component (empty: nothing is excused)
    bad = {}
    for o in objs:
        for sym, cnt in packed_fp32_kernels(o).items():
            if not any(t in sym for t in ok):
                bad[sym] = cnt
    if bad:
        raise RuntimeError("packed fp32 arithmetic (v_pk_*_f32) in kernels outside build.PACKED_FP32_ALLOWED -- see the comment "
                           "there; either remove it or add the kernel to the list AND to tests/test_gpu_corun.py:\n  "
                           + "\n  ".join(f"{k}: {v}" for k, v in sorted(bad.items())))


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libcpc_hip.so for gfx950)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps_mtime():
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, obj, extra):
    cmd = [_hipcc(), *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return r.stderr


def build(force=False, verbose=False, extra_flags=(), variant=None, gate=True):
    """variant: build lib/libcpc_hip_<variant>.so (own object directory) with ``extra_flags`` appended -- A/B runs through
    the CPC_HIP_LIB override of _lib.py.  gate: run check_packed_fp32 on the objects."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" if not variant else f"obj_{variant}")
    LIB = os.path.join(LIBDIR, "libcpc_hip.so" if not variant else f"libcpc_hip_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    hdr_m = _deps_mtime()
    # a change of flags rebuilds everything (object mtimes do not know about it)
    stamp, flagline = os.path.join(objdir, "flags.txt"), " ".join([*FLAGS, *extra_flags, repr(sorted(FILE_FLAGS.items()))])
    if not os.path.exists(stamp) or open(stamp).read() != flagline:
        force = True
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append((src, obj))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            futs = {ex.submit(_compile, s, o, list(extra_flags)): s for s, o in jobs}
            for f in cf.as_completed(futs):
                out = f.result()
                if verbose and out:
                    print(out)
    if jobs:
        with open(stamp, "w") as f:
            f.write(flagline)
    # (objects newer than the library: a previous run compiled them and then failed the gate or the link)
    relink = bool(jobs) or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if gate and relink:
        check_packed_fp32(objs)
    if relink:
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    # python -m cpc_audio_amd.build [--force] [--variant NAME [extra hipcc flags ...]] [--no-gate]
    args = [a for a in sys.argv[1:] if a not in ("--force", "--no-gate")]
    variant, extra = None, []
    if "--variant" in args:
        i = args.index("--variant")
        variant, extra = args[i + 1], args[i + 2:]
    p = build(force="--force" in sys.argv, verbose=True, extra_flags=extra, variant=variant, gate="--no-gate" not in sys.argv)
    print(p)
