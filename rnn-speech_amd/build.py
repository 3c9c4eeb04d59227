"""Build libamdspeech.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libamdspeech.so")
SOURCES = ["api.hip", "gemm.hip", "lstm.hip", "ctc.hip", "optim.hip", "frontend.hip", "bn.hip", "beam.cpp", "audio_io.cpp"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "amdspeech.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-pass-failed", "-o", LIB] + srcs
    if os.environ.get("AMDSPEECH_DEVTRACE"):     # dev builds: in-kernel timestamps / ablation switches
        cmd.insert(1, "-DAMDSPEECH_DEVTRACE=" + os.environ["AMDSPEECH_DEVTRACE"])
    for extra in os.environ.get("AMDSPEECH_CXXFLAGS", "").split():      # dev: tuning macros (-DFLOW_...=n)
        cmd.insert(1, extra)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
