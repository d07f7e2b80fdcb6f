"""Is tests/test_shac_gpu.py::test_forward_does_not_depend_on_stale_lds able to fail?  Build the library with the two pad-column
zero-fills compiled out (-DVF_TEST_NO_PAD_ZERO) next to the real one and run that test against it:
    python tools/exp_pad_poison.py build      (here: cross-compiles tools/libvf_nopad.so)
    python tools/exp_pad_poison.py            (GPU box: expects the test to FAIL with the crippled build)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALT = os.path.join(ROOT, "tools", "libvf_nopad.so")
if sys.argv[1:] == ["build"]:
    from visfly_amd import _build
    _build.build(force=True, extra_flags=["-DVF_TEST_NO_PAD_ZERO"], out=ALT)
    print("built", ALT)
else:
    code = ("import sys; sys.path.insert(0, %r); from visfly_amd import _build, _lib; _build.LIB = _lib.LIB = %r; import pytest; "
            "sys.exit(pytest.main(['-x', '-q', %r, '-k', 'stale_lds']))" % (ROOT, ALT, os.path.join(ROOT, "tests", "test_shac_gpu.py")))
    rc = subprocess.call([sys.executable, "-c", code])
    print("crippled build: pytest rc", rc, "(expected != 0)")
