"""Build the HIP shared library in-tree (gfx950 only).

``python -m mbexwn_vocoder_amd.build`` or ``build_library()``; hipcc cross-compiles without a GPU.
Every source is compiled to its own object (in parallel, only when it or a header changed) and the objects are
linked into ``libmbexwn_hip.so`` next to this file, so the library travels with the tree.
The product path never falls back to a CPU implementation: if the library is missing, importing
the engine raises.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "build")
LIB_PATH = os.path.join(HERE, "libmbexwn_hip.so")
SOURCES = ["conv_mfma.hip", "wn_winograd2w.hip", "wn_winograd4w.hip", "wn_gate0.hip", "wn_resskip.hip", "wn_resskip_wide.hip", "wn_resskip_wave.hip", "wn_resskip_f16.hip", "wn_gate_f16.hip", "wn_tail.hip",
           "elementwise.hip", "wavetable.hip", "pqmf.hip", "stft_filter.hip", "mel_analysis.hip", "norm_mel.hip", "mbx_api.hip"]
HEADERS = ["mbx_kernels.h", "fft_lds.h", os.path.join("..", "..", "include", "mbexwn.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _stale_sources():
    hdr = max(_mtime(os.path.join(CSRC, hh)) for hh in HEADERS)
    hdr = max(hdr, _mtime(os.path.abspath(__file__)))
    out = []
    for ss in SOURCES:
        obj = os.path.join(OBJ_DIR, ss + ".o")
        if _mtime(obj) < max(_mtime(os.path.join(CSRC, ss)), hdr):
            out.append(ss)
    return out


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    lib_time = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, ss) for ss in SOURCES + HEADERS]
    return any(os.path.getmtime(dd) > lib_time for dd in deps)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = SOURCES if force else _stale_sources()

    def compile_one(ss):
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, ss), "-o", os.path.join(OBJ_DIR, ss + ".o")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        return ss, subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as pool:
        for ss, res in pool.map(compile_one, todo):
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed on {ss}:\n{res.stdout}\n{res.stderr}")
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH]
    cmd += [os.path.join(OBJ_DIR, ss + ".o") for ss in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
