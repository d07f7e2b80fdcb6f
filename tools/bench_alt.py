"""bench.py on another build of the library (A/B): VF_ALT_LIB=/path/lib.so python tools/bench_alt.py <bench.py arguments>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("VF_ALT_LIB"):
    from visfly_amd import _build, _lib
    _build.LIB = _lib.LIB = os.environ["VF_ALT_LIB"]
path = os.path.join(ROOT, "bench.py")
sys.argv[0] = path
exec(compile(open(path).read(), path, "exec"), {"__name__": "__main__", "__file__": path})
