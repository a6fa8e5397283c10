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
         "-munsafe-fp-atomics", "-Wno-unused-result"]


# Per-file flags.  enc_conv0.hip: no SLP vectorisation, i.e. no compiler-formed v_pk_fma_f32 in conv0_bwd_kernel -- with
# them the kernel returns wrong partial sums whenever a 16-bit-MFMA GEMM kernel shares the chip with it (measured,
# tools/probe_corun.py; same speed without them: the kernel is bound by VALU issue either way).  conv0_fwd_kernel keeps its
# explicit f32x2 arithmetic and never runs beside such a kernel.
FILE_FLAGS = {"enc_conv0.hip": ["-fno-slp-vectorize"]}


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


def build(force=False, verbose=False, extra_flags=()):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_m = _deps_mtime()
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
    if jobs or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print(p)
