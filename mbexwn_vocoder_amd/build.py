"""Build the HIP shared library in-tree (gfx950 only).

``python -m mbexwn_vocoder_amd.build`` or ``build_library()``; hipcc cross-compiles without a GPU.
The product path never falls back to a CPU implementation: if the library is missing, importing
the engine raises.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libmbexwn_hip.so")
SOURCES = ["conv_mfma.hip", "wn_winograd.hip", "wn_winograd4.hip", "wn_winograd4k.hip", "wn_resskip.hip", "wn_tail.hip", "elementwise.hip", "wavetable.hip", "pqmf.hip", "stft_filter.hip", "mbx_api.hip"]
HEADERS = ["mbx_kernels.h", os.path.join("..", "..", "include", "mbexwn.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    lib_time = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, ss) for ss in SOURCES + HEADERS]
    return any(os.path.getmtime(dd) > lib_time for dd in deps)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB_PATH]
    cmd += [os.path.join(CSRC, ss) for ss in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
