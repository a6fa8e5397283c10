"""Build the kernels for the host SIMT emulator (TEST TOOL ONLY, see include/hip/hip_runtime.h).
The same .hip sources as the product library, compiled as x86 C++ by clang."""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "cpc_audio_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libcpc_emu.so")


def _cxx():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError("clang++ not found")


def asan_runtime():
    """clang's shared ASan runtime (to LD_PRELOAD into the python that loads the sanitized library), or None."""
    hits = glob.glob(os.path.join(os.path.dirname(os.path.dirname(_cxx())), "lib", "clang", "*", "lib", "linux",
                                  "libclang_rt.asan-x86_64.so"))
    return hits[0] if hits else None


def build(force=False, sanitize=False):
    """sanitize: AddressSanitizer + UndefinedBehaviorSanitizer build of the same sources (libcpc_emu_san.so): every C entry
    point, its host logic and the kernels' index arithmetic run under the sanitizers in tests/test_emu_sanitized.py."""
    global OUT, LIB
    if sanitize:
        out_dir, lib = os.path.join(HERE, "_build", "san"), os.path.join(HERE, "_build", "san", "libcpc_emu_san.so")
        return _build(force, out_dir, lib, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                            "-fno-omit-frame-pointer", "-shared-libasan", "-g1"])
    return _build(force, OUT, LIB, [])


def _host_key(flags):
    """What the cached objects were compiled for: the compile flags and this host's CPU feature list (-march=native)."""
    import hashlib
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    cpu = line
                    break
    except OSError:
        pass
    return hashlib.sha1((" ".join(flags) + "|" + cpu).encode()).hexdigest()


def _build(force, OUT, LIB, extra):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "hipemu.cpp")]
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        glob.glob(os.path.join(HERE, "include", "hip", "*.h"))
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    flags = ["-O2", "-march=native", "-std=c++17", "-fPIC", "-x", "c++", "-I", os.path.join(HERE, "include"),
             "-ffp-contract=off", "-Wno-unused-value", "-Wno-unknown-pragmas", "-Wno-pass-failed",
             "-Wno-unused-variable", *extra]
    # objects of another flag set or another host (the -march=native code of a machine with other vector units) are stale
    key, key_file = _host_key(flags), os.path.join(OUT, "host.key")
    try:
        with open(key_file) as f:
            force = force or f.read().strip() != key
    except OSError:
        force = True
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([_cxx(), *flags, "-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"emu compile failed for {s}:\n{r.stderr[-6000:]}")

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(cc, jobs))
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([_cxx(), "-shared", "-fPIC", *extra, *objs, "-o", LIB, "-lpthread"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"emu link failed:\n{r.stderr[-4000:]}")
    with open(key_file, "w") as f:
        f.write(key)
    return LIB


if __name__ == "__main__":
    print(build())
