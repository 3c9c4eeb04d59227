"""Build libamdspeech.so for gfx950 with hipcc (cross-compiles without a GPU).

Every source is compiled to its own object under csrc/.obj/ (in parallel, only when it or a header changed),
then linked; the objects are build artefacts (git-ignored), the .so ships to the GPU box with the tree."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# dev: AMDSPEECH_LIB_OUT=<path> builds a variant library (own object directory) next to the product one
LIB = os.environ.get("AMDSPEECH_LIB_OUT") or os.path.join(HERE, "libamdspeech.so")
OBJ = os.path.join(CSRC, ".obj" + ("" if "AMDSPEECH_LIB_OUT" not in os.environ
                                   else "_" + os.path.basename(LIB).replace(".so", "")))
SOURCES = ["api.hip", "gemm.hip", "gemm_bf3.hip", "gemm_bf16p.hip", "gemm_skinny.hip", "lstm.hip", "ctc.hip", "optim.hip", "frontend.hip", "bn.hip", "beam.cpp",
           "audio_io.cpp", "comm.cpp"]
HEADERS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
          [os.path.join(os.path.dirname(HERE), "include", "amdspeech.h")]


def _flags():
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]
    if os.environ.get("AMDSPEECH_DEVTRACE"):     # dev builds: in-kernel timestamps / ablation switches
        flags.append("-DAMDSPEECH_DEVTRACE=" + os.environ["AMDSPEECH_DEVTRACE"])
    flags += os.environ.get("AMDSPEECH_CXXFLAGS", "").split()      # dev: tuning macros (-DFLOW_...=n)
    return flags


def _stamp(src, flags):
    h = hashlib.sha1(" ".join(flags).encode())
    for path in [src] + HEADERS:
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _flags()
    os.makedirs(OBJ, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        stamp = _stamp(src, flags)
        tag = obj + ".stamp"
        fresh = os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == stamp
        if force or not fresh:
            jobs.append((src, obj, tag, stamp))

    def compile_one(job):
        src, obj, tag, stamp = job
        cmd = [hipcc] + flags + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(tag, "w") as fh:
            fh.write(stamp)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
