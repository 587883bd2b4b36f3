"""Build libcelebbasis_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library links only the HIP runtime; it has no torch/ATen dependency (the boundary is a plain C ABI,
include/celebbasis_hip.h). Object files are cached under celebbasis_amd/_build and rebuilt when a source
or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libcelebbasis_hip.so")
SOURCES = ["cb_gemm.hip", "cb_gemm_glds.hip", "cb_flash.hip", "cb_flash_bf16.hip", "cb_kernels.hip", "cb_engine.hip", "cb_models.hip",
           "cb_abi.hip"]
HEADERS = ["cb_common.h", "cb_device.h", "cb_gemm_tiles.h", "cb_plan_table.inc", "cb_engine.h", "cb_models.h", os.path.join("..", "..", "include", "celebbasis_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unused-lambda-capture", "-Wno-unused-result"]
# Attention kernels: MFMA results feed VALU (softmax) and the next MFMA's B operand directly, so the accumulators belong in
# ordinary VGPRs. The default AGPR form costs ~100 v_accvgpr_read/write per streamed tile and 28 .. 78 extra registers
# (flash_fwd<3> 128 -> 100, flash_bwd_dkv<3> 256 -> 178). CB_MFMA_VGPR_FORM=0 restores the default (A/B measurements).
EXTRA = {src: (["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.environ.get("CB_MFMA_VGPR_FORM", "1") != "0" else [])
         for src in ("cb_flash.hip", "cb_flash_bf16.hip")}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(src_paths, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_paths)


def build(verbose=True, force=False):
    os.makedirs(BUILD, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()

    def compile_one(src):
        s = os.path.join(CSRC, src)
        extra = EXTRA.get(src, [])
        o = os.path.join(BUILD, src.replace(".hip", "_x.o" if extra else ".o"))       # the flag set is part of the cache key
        if force or _newer([s] + hdrs, o):
            cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print("[celebbasis_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print("[celebbasis_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
