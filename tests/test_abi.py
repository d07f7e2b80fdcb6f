"""The C-ABI library loads (no GPU needed) and exports every symbol include/visfly_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_exported():
    import __graft_entry__ as ge
    ge.build()
    from visfly_amd import _lib
    L = _lib.lib()
    syms = declared_symbols()
    assert "vf_dyn_step" in syms and len(syms) >= 7
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/ but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in visfly_amd/_lib.py"
    assert L.vf_abi_version() == _lib.ABI_VERSION == 10


def test_cfg_struct_size_matches_header():
    """ctypes mirror and C struct agree (compiled probe)"""
    import subprocess, tempfile
    from visfly_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "p.c")
        open(p, "w").write('#include <stdio.h>\n#include "visfly_amd.h"\n'
                           'int main(){printf("%zu\\n", sizeof(vf_dyn_cfg));return 0;}\n')
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", exe])
        size = int(subprocess.check_output([exe]).decode())
    assert size == ctypes.sizeof(_lib.DynCfg)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from visfly_amd import Dynamics
    from visfly_amd._lib import VisflyError
    with pytest.raises(VisflyError):
        Dynamics(num=4, device="cpu")
