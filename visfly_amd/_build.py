"""Builds libvisfly_amd.so (hand-written HIP for gfx950) in-tree with hipcc."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libvisfly_amd.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

# -ffp-contract=off: the kernels reproduce the reference's fp32 rounding sequence; FMAs are
# written explicitly where the reference fuses.  Division/sqrt stay IEEE (hipcc default).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


# per-source extra flags.  The single-step kernels: the dispatcher preloads the leading kernel arguments into SGPRs (gfx940+), see k_env_step
_PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
# ... and -slp-threshold=8: hipcc's SLP vectoriser pairs scalar fp32 operations into v_pk_* instructions; a lone wave issues a packed
# instruction in the slot of a scalar one, but every pair whose operands are not in adjacent registers costs v_mov's.  At the default
# threshold the sub-step loop of k_env_step is 315 instructions (126 packed, 36 moves); at 8 only the pairs that pay are formed: 301
# (82 packed, 8 moves) -- the minimum of a scan over 2 .. 24 (profiles/r04_env_timeline.txt).  Same IEEE operations: bit-identical.
_STEP = _PRELOAD + ["-mllvm", "-slp-threshold=8"]
PER_SOURCE_FLAGS = {"vf_env.hip": _STEP, "vf_dyn.hip": _STEP}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return os.path.getmtime(LIB) < max(os.path.getmtime(p) for p in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """one `hipcc -c` per source in parallel (the register-chained MLP kernels are ~1.5 min of fully unrolled code on
    their own), then one link.  Objects are kept under csrc/.obj and reused while they are newer than their source, every header
    and this file (default flags only): editing one .hip costs one compile."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    out = out or LIB
    if not force and out == LIB and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = f"{out}.{os.getpid()}.tmp"          # link to a private name, then rename: concurrent builders never see a torn file
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags) + ["-I", INCLUDE, "-I", CSRC]
    keep = not extra_flags and not force
    objdir = os.path.join(CSRC, ".obj") if keep else tempfile.mkdtemp(prefix="visfly_amd_build_")
    os.makedirs(objdir, exist_ok=True)
    common = max(os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h")) + [__file__])

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if keep and os.path.exists(obj) and os.path.getmtime(obj) > max(common, os.path.getmtime(src)):
            return obj
        part = f"{obj}.{os.getpid()}.tmp"
        cmd = [hipcc] + cflags + PER_SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", part]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(part, obj)
        return obj

    try:
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            objs = list(pool.map(compile_one, sources()))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
        if not keep:
            shutil.rmtree(objdir, ignore_errors=True)
    return out
