"""Builds csrc/libmmr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m mmr_amd.csrc.build        # or: __graft_entry__.build()

The shared library has no torch / python dependency: it links only the HIP runtime and
exports the C ABI declared in include/mmr.h.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api_common.cpp", "comm.cpp", "search.hip", "gemm.hip", "vit_ops.hip", "tower.hip", "preprocess.hip"]
LIB = os.path.join(HERE, "libmmr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
         "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("MMR_EXTRA_HIPCC_FLAGS", "").split()


# per-source flags.  vit_ops.hip: keep MFMA results in arch VGPRs -- the attention kernels post-process every
# accumulator with VALU code, and the default AGPR form costs a v_accvgpr_read/write per element (measured: streaming
# attention 502 -> 463 us at ViT-L/14@336 B=128).  The GEMM and scan kernels already compile AGPR-free.
# -fno-honor-nans drops the canonicalising v_max x,x clang puts in front of every fmaxf on an MFMA result (30 of the
# 56 max instructions in the streaming-attention loop: 455 -> 440 us); nothing in that file tests for NaN.
PER_SOURCE_FLAGS = {"vit_ops.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"]}


def _obj(src):
    return os.path.join(HERE, "_obj", os.path.splitext(src)[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "..", "include", "mmr.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]

    def compile_one(src):
        sp, op = os.path.join(HERE, src), _obj(src)
        if force or _stale(op, [sp] + headers):
            cmd = [HIPCC] + FLAGS + PER_SOURCE_FLAGS.get(src, []) + ["-c", sp, "-o", op]
            if verbose:
                print("[mmr build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return True
        return False

    with ThreadPoolExecutor(max_workers=4) as ex:
        rebuilt = list(ex.map(compile_one, srcs))
    objs = [_obj(s) for s in srcs]
    if force or any(rebuilt) or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print("[mmr build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
