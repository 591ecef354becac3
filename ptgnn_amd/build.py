"""In-tree build of libptgnn_amd.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m ptgnn_amd.build [--force]

The shared object is placed next to the sources (ptgnn_amd/csrc/libptgnn_amd.so); it is
git-ignored but travels to the GPU box with the working tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libptgnn_amd.so")
SOURCES = ["errors.cpp", "csr_build.hip", "gather_reduce.hip", "dense_f32.hip", "stream_gemm.hip", "edge_gemm.hip", "edge_wgrad.hip", "wgrad_stream.hip", "batching.hip", "row_epilogue.hip", "shard_index.hip", "segment_mul.hip", "weighted_pool.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "dense_common.h"), os.path.join(CSRC, "stream_gemm.h"), os.path.join(CSRC, "wgrad_stream.h"),
           os.path.join(INCLUDE, "ptgnn_amd.h")]
EXPORTS = os.path.join(CSRC, "exports.map")   # only ptgnn_amd_* leaves the library
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-deprecated-declarations"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    if force or _stale(obj, [path] + HEADERS):
        cmd = [hipcc()] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        subprocess.run(cmd, check=True)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB, objs + [EXPORTS]):
        # NEEDED libamdhip64.so.7 resolves to the copy torch already loaded in-process (same SONAME)
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={EXPORTS}", "-o", LIB] + objs
        subprocess.run(cmd, check=True)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
