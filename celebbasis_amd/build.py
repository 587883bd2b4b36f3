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
SOURCES = ["cb_gemm.hip", "cb_gemm_glds.hip", "cb_gemm_bf16_p8.hip", "cb_flash.hip", "cb_flash_bf16.hip", "cb_flash_bf16_v2.hip", "cb_kernels.hip", "cb_engine.hip", "cb_models.hip",
           "cb_abi.hip"]
HEADERS = ["cb_common.h", "cb_device.h", "cb_gemm_tiles.h", "cb_plan_table.inc", "cb_engine.h", "cb_models.h", os.path.join("..", "..", "include", "celebbasis_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unused-lambda-capture", "-Wno-unused-result"]
# Attention kernels: MFMA results feed VALU (softmax) and the next MFMA's B operand directly, so the accumulators belong in
# ordinary VGPRs. The default AGPR form costs ~100 v_accvgpr_read/write per streamed tile and 28 .. 78 extra registers
# (flash_fwd<3> 128 -> 100, flash_bwd_dkv<3> 256 -> 178). CB_MFMA_VGPR_FORM=0 restores the default (A/B measurements).
EXTRA = {src: (["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.environ.get("CB_MFMA_VGPR_FORM", "1") != "0" else [])
         for src in ("cb_flash.hip", "cb_flash_bf16.hip", "cb_flash_bf16_v2.hip")}
# Kernels whose software pipeline COUNTS its outstanding VMEM instructions (`s_waitcnt vmcnt(N)` + raw `s_barrier`,
# cb_gemm_glds.hip): a register spill would add scratch loads / stores to that count and let a tile be read before it has landed.
# The build therefore fails if any kernel of these files has scratch, spills or a dynamic stack (-Rpass-analysis remarks).
# (The f2_* attention kernels drain with vmcnt(0) and do not depend on the count: scratch there is reported, not fatal.)
if os.environ.get("CB_FLASH_ABL"):       # timing ablations of the fp32 attention backward (cb_flash.hip, FL_ABL): results are wrong under them
    EXTRA["cb_flash.hip"] = EXTRA["cb_flash.hip"] + ["-DFL_ABL=" + str(int(os.environ["CB_FLASH_ABL"]))]
NO_SCRATCH = {"cb_gemm_glds.hip": ("gemm_glds", True), "cb_gemm_bf16_p8.hip": ("gemm_bf16_p8", True), "cb_flash_bf16_v2.hip": ("f2_", False)}


def source_digest():
    """Identity of the native code a tree builds: sha256 over the kernel sources, headers, the public header and the compile
    flags (`src_sha`) and over the measured plan table alone (`plan_table_sha`). Profiles under profiles/ carry the stamp of the
    tree they were measured on; bench.py compares it with the running tree's and marks anything else `profile_stale`.
    (A commit hash would not do: the GPU box receives a snapshot without .git.)"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(FLAGS + sum((EXTRA[k] for k in sorted(EXTRA)), [])).encode())
    with open(os.path.join(CSRC, "cb_plan_table.inc"), "rb") as f:
        pt = hashlib.sha256(f.read()).hexdigest()
    return dict(src_sha=h.hexdigest()[:16], plan_table_sha=pt[:16])


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


_probed = {}


def _flag_supported(hipcc, flags):
    """compile a trivial kernel with `flags` once: LLVM-internal options (-mllvm ...) come and go between compiler releases, and an
    unknown one aborts the whole build with 'Unknown command line argument'"""
    key = (hipcc,) + tuple(flags)
    if key not in _probed:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void probe(float* p) { p[threadIdx.x] = 1.f; }\n")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1"] + list(flags) + ["-c", src, "-o", os.path.join(d, "probe.o")],
                               capture_output=True, text=True)
            _probed[key] = r.returncode == 0
    return _probed[key]


def check_no_scratch(remarks, needle):
    """parse -Rpass-analysis=kernel-resource-usage remarks; return [(kernel, field, value)] violations for kernels matching `needle`"""
    import re
    bad, cur, seen = [], None, 0
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1) if needle in m.group(1) else None
            seen += cur is not None
            continue
        if cur is None:
            continue
        # SGPR -> VGPR spills add no memory instructions and are harmless to the count
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs Spill|Dynamic Stack): (\S+)", line)
        if m and m.group(2) not in ("0", "False"):
            bad.append((cur, m.group(1), m.group(2)))
        m = re.search(r"remark:\s+SGPRs Spill: (\S+)", line)          # SGPR spills go to VGPR lanes, not scratch: slower, not a vmcnt hazard
        if m and m.group(1) != "0":
            print(f"[celebbasis_amd.build] WARNING: {cur}: SGPRs Spill = {m.group(1)}", flush=True)
    return bad, seen


def build(verbose=True, force=False):
    """Serialised across processes: N ranks importing `lib` against a stale or missing library all arrive here; an exclusive flock on
    _build/.lock lets the first one compile while the others wait and then find everything up to date (tests/test_host_cpu.py)."""
    import fcntl
    os.makedirs(BUILD, exist_ok=True)
    with open(os.path.join(BUILD, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose, force):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()
    for flags in {tuple(v) for v in EXTRA.values() if v}:          # probe once, before the worker threads share the cache
        _flag_supported(hipcc, flags)

    def compile_one(src):
        import hashlib
        s = os.path.join(CSRC, src)
        extra = list(EXTRA.get(src, []))
        if extra and not _flag_supported(hipcc, extra):
            print(f"[celebbasis_amd.build] WARNING: {hipcc} rejects {' '.join(extra)}; building {src} without it "
                  "(attention accumulators in AGPR form: correct, a few percent slower)", flush=True)
            extra = []
        guard, fatal = NO_SCRATCH.get(src, (None, False))
        if guard:
            extra += ["-Rpass-analysis=kernel-resource-usage"]
        # the RESOLVED flag set is part of the object's cache key
        tag = "" if not extra else "_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:6]
        o = os.path.join(BUILD, src.replace(".hip", tag + ".o"))
        if force or _newer([s] + hdrs, o):
            cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print("[celebbasis_amd.build]", " ".join(cmd), flush=True)
            if guard:
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    sys.stderr.write(r.stderr)
                    raise subprocess.CalledProcessError(r.returncode, cmd)
                show = False                                        # ordinary warnings of these files are not swallowed
                for l in r.stderr.splitlines():
                    if "remark:" in l:
                        show = False
                    elif "warning:" in l or "error:" in l:
                        show = True
                    if show:
                        sys.stderr.write(l + "\n")
                bad, seen = check_no_scratch(r.stderr, guard)
                # no remarks at all = the guard could not run. Fatal for the counted-vmcnt kernels (a silent spill there reads tiles before
                # they land); CB_REQUIRE_REMARKS=0 is the explicit opt-out for a toolchain that renamed the remark
                if not seen and (not fatal or os.environ.get("CB_REQUIRE_REMARKS", "1") == "0"):
                    print(f"[celebbasis_amd.build] WARNING: {hipcc} printed no kernel-resource-usage remarks for {src}; the spill "
                          "guard could not run" + (" (CB_REQUIRE_REMARKS=0: accepted)" if fatal else ""), flush=True)
                elif bad and not fatal:
                    print(f"[celebbasis_amd.build] WARNING: {src}: scratch / spills in {bad[:4]} (slower, still correct)", flush=True)
                elif bad or not seen:
                    if os.path.exists(o):
                        os.remove(o)
                    raise RuntimeError(f"{src}: the counted-vmcnt pipeline requires spill-free kernels; "
                                       + (f"violations: {bad[:4]}" if bad else "no resource-usage remarks found for its kernels"))
                with open(o + ".resources.txt", "w") as f:
                    f.write("\n".join(l for l in r.stderr.splitlines() if "remark:" in l) + "\n")
            else:
                subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # the library is relinked when an object is newer OR when it was linked from another set of objects (a flag change — CB_MFMA_VGPR_FORM,
    # CB_FLASH_ABL — selects differently tagged objects that may all be older than the library)
    manifest = os.path.join(BUILD, "linked_objects.txt")
    want = "\n".join(os.path.basename(o) for o in objs)
    try:
        with open(manifest) as f:
            same = f.read() == want
    except OSError:
        same = False
    if force or not same or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print("[celebbasis_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(manifest, "w") as f:
            f.write(want)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
