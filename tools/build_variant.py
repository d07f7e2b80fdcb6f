"""A/B builds of the library that differ in ONE translation unit: objects of the shipped flags are cached under tools/tmp/obj, the named
source is recompiled with extra flags and everything is relinked.
    python tools/build_variant.py vf_mlp_chain_split.hip tools/tmp/libvf_x.so -DVF_CHAIN_DEPTH=4 ..."""
import os, subprocess, sys, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visfly_amd import _build
from concurrent.futures import ThreadPoolExecutor
src_name, out, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
objdir = os.path.join(_build.CSRC, ".obj")        # the object cache of visfly_amd/_build.py (run the normal build first)
os.makedirs(objdir, exist_ok=True)
hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
cflags = [f for f in _build.HIPCC_FLAGS if f != "-shared"] + ["-I", _build.INCLUDE, "-I", _build.CSRC]
hdr_time = max([os.path.getmtime(os.path.join(_build.CSRC, f)) for f in os.listdir(_build.CSRC) if f.endswith(".hpp")] +
               [os.path.getmtime(os.path.join(_build.INCLUDE, "visfly_amd.h"))])
def obj_for(src, variant=False):
    return os.path.join(objdir, os.path.basename(src) + (".variant.o" if variant else ".o"))
def compile_one(src, flags, obj):
    subprocess.check_call([hipcc] + cflags + _build.PER_SOURCE_FLAGS.get(os.path.basename(src), []) + flags + ["-c", src, "-o", obj])
todo = []
for src in _build.sources():
    if os.path.basename(src) == src_name:
        todo.append((src, extra, obj_for(src, True)))
    else:
        o = obj_for(src)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), hdr_time):
            todo.append((src, [], o))
with ThreadPoolExecutor(max_workers=8) as pool:
    list(pool.map(lambda a: compile_one(*a), todo))
objs = [obj_for(s, os.path.basename(s) == src_name) for s in _build.sources()]
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print("built", out, "recompiled", [os.path.basename(t[0]) for t in todo])
