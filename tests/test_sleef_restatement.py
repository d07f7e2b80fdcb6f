"""Transcendentals of the reference path (oracle/vf_sleef.h == visfly_amd/csrc/vf_xmath.hpp):
  * torch.atan2 (SLEEF atan2f_u10) is restated bit for bit -- pinned against torch itself here;
  * torch.sin / cos / acos run closed-source MKL VML: SLEEF's u10 routines, restated, are the closest published algorithms --
    at most one ulp from torch, on a bounded fraction of arguments;
  * the device copy carries the identical text."""
import os
import re

import numpy as np
import torch

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_shared_text_identical():
    def shared(path):
        s = open(os.path.join(ROOT, path)).read()
        return s[s.index("/* ==== shared text"):s.index("/* ==== end of shared text ==== */")]
    a, b = shared("oracle/vf_sleef.h"), shared("visfly_amd/csrc/vf_xmath.hpp")
    assert a == b and len(a) > 4000
    assert not re.search(r"\b(sinf|cosf|atan2f|acosf)\(", a), "the shared text must not call a math library"


def _pad32(*arrs):
    """torch's vectorised CPU loop covers blocks of 2 x Vec::size() = 32 floats (AVX-512); the tail runs scalar glibc"""
    n = (arrs[0].size + 31) // 32 * 32
    return [np.concatenate([a, np.full(n - a.size, a[-1], a.dtype)]) for a in arrs]


def test_atan2_is_torch_atan2():
    assert torch.backends.cpu.get_cpu_capability() in ("AVX512", "AVX2"), "needs torch's vectorised SLEEF path"
    rng = np.random.default_rng(0)
    n = 1 << 21
    cases = [(rng.normal(0, 3, n), rng.normal(0, 3, n))]
    mag = lambda: np.exp(rng.uniform(-90, 88, n)) * rng.choice([-1, 1], n)
    cases.append((mag(), mag()))                                   # all magnitudes incl. denormal results
    y = mag()
    cases.append((y, y * rng.uniform(0.5, 2, n)))                  # near the octant boundaries
    cases.append((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n) * 1e-3))
    e = np.array([0.0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-39, -2.9e-39, 3e38, 0.5, -0.5, 1e-20, 7.0], np.float32)
    Y, X = np.meshgrid(e, e)
    cases.append((Y.ravel(), X.ravel()))
    for y, x in cases:
        y, x = _pad32(np.asarray(y, np.float32), np.asarray(x, np.float32))
        want = torch.atan2(torch.from_numpy(y), torch.from_numpy(x)).numpy()
        got = oracle.xmath("atan2", y, x)
        same = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (y[~same][:4], x[~same][:4], got[~same][:4], want[~same][:4])


def test_sin_cos_acos_one_ulp_from_torch():
    """torch runs closed-source MKL VML for these three; the restated SLEEF u10 routines (bit-identical to the Sleef_*f16_u10
    symbols of libtorch_cpu.so: oracle/probe_torch_transcendentals.py) are within one ulp of torch, for a bounded
    fraction of the arguments (measured 1.9 % / 2.3 % / 8.3 %)"""
    rng = np.random.default_rng(1)
    n = 1 << 21
    ang = np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-124, 124, n // 4)]).astype(np.float32)
    unit = np.concatenate([rng.uniform(-1, 1, n), np.array([1, -1, 0.5, -0.5, 0, -0.0, 0.99999994, -0.99999994], np.float32)]).astype(np.float32)
    for kind, x, tfn, frac in (("sin", ang, torch.sin, 0.03), ("cos", ang, torch.cos, 0.035), ("acos", unit, torch.acos, 0.10)):
        got = oracle.xmath(kind, x)
        t = tfn(torch.from_numpy(x)).numpy()
        d = np.abs(bits(got).astype(np.int64) - bits(t).astype(np.int64))
        assert d.max() <= 1, f"{kind}: more than one ulp from torch"
        assert (d != 0).mean() < frac, f"{kind}: {100 * (d != 0).mean():.1f} % of the arguments differ from torch"
    assert bits(oracle.xmath("sin", np.array([-0.0], np.float32)))[0] == 0x80000000
    assert np.isnan(oracle.xmath("acos", np.array([1.5, np.nan], np.float32))).all()
    e = np.array([0.0, 3.14159265, -3.14159265, 1.5707964, 6.2831855, -6.2831855, 100, 124.99, 1e-20, -1e-20], np.float32)
    assert np.abs(oracle.xmath("sin", e).astype(np.float64) - np.sin(e.astype(np.float64))).max() < 1.2e-7
    assert np.abs(oracle.xmath("cos", e).astype(np.float64) - np.cos(e.astype(np.float64))).max() < 1.2e-7


def test_cr_mode_is_the_rounded_fp64_result():
    """sin_cr / cos_cr / acos_cr (fdlibm's fp64 kernels restated, rounded once to fp32: the "cr" transcendental mode) ==
    torch.f(x.double()).float() -- what oracle/gen_golden.py::use_cr_trig patches the reference's sin / cos / acos to -- for
    every argument tried (8 M incl. the edges); and they are NOT what torch's fp32 routines return (MKL VML: ~5 % differ),
    which is why the reference has to be patched to be pinnable at all"""
    rng = np.random.default_rng(2)
    n = 1 << 21
    ang = np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-124, 124, n // 2), rng.uniform(-1e4, 1e4, n // 2),
                          np.array([0.0, -0.0, 3.14159265, -3.14159265, 1.5707964, 6.2831855, 1e-20, -1e-20, 1e-40, 0.78539816], np.float32)]).astype(np.float32)
    unit = np.concatenate([rng.uniform(-1, 1, n), 1 - np.exp(rng.uniform(-20, 0, n // 2)), -1 + np.exp(rng.uniform(-20, 0, n // 2)),
                           np.array([1, -1, 0.5, -0.5, 0, -0.0, 0.99999994, -0.99999994, 1e-20, -1e-30], np.float32)]).astype(np.float32)
    for kind, x, tfn in (("sin_cr", ang, torch.sin), ("cos_cr", ang, torch.cos), ("acos_cr", unit, torch.acos)):
        got = oracle.xmath(kind, x)
        want = tfn(torch.from_numpy(x).double()).float().numpy()
        assert np.array_equal(bits(got), bits(want)), f"{kind}: {(bits(got) != bits(want)).sum()} mismatches"
        t32 = tfn(torch.from_numpy(x)).numpy()
        assert 0.01 < (bits(t32) != bits(got)).mean() < 0.12, kind
    assert bits(oracle.xmath("sin_cr", np.array([-0.0], np.float32)))[0] == 0x80000000
    assert np.isnan(oracle.xmath("acos_cr", np.array([1.5, np.nan], np.float32))).all()
    assert np.isnan(oracle.xmath("sin_cr", np.array([np.inf, np.nan], np.float32))).all()


def test_strided_atan2_is_glibc_atan2f():
    """torch.atan2 runs SLEEF only in its vectorised loop (contiguous operands); strided operands -- the velocity rows of the
    velocity action type's auto-yaw whenever Dynamics.reset stored `vel.T` (dynamics.py:236,423-427) -- go through glibc's
    atan2f, restated as vfs_atan2f_glibc (fdlibm's e_atan2f.c / s_atanf.c, as glibc 2.35 ships them)"""
    rng = np.random.default_rng(3)
    n = 1 << 20

    def strided_atan2(y, x):
        a = torch.from_numpy(np.stack([x, y], 1).copy()).T
        assert not a[1].is_contiguous()
        return torch.atan2(a[1], a[0]).numpy()
    mag = lambda: (np.exp(rng.uniform(-80, 80, n)) * rng.choice([-1, 1], n)).astype(np.float32)
    y0 = mag()
    e = np.array([0.0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-39, 3e38, 0.5, 7.0, 0.4375, 0.6875, 1.1875, 2.4375,
                  2.0 ** 25, 2.0 ** -29], np.float32)
    Y, X = np.meshgrid(e, e)
    cases = [(rng.normal(0, 3, n), rng.normal(0, 3, n)), (mag(), mag()), (y0, y0 * rng.uniform(0.3, 3, n)),
             (rng.uniform(-1, 1, n), np.ones(n)), (Y.ravel(), X.ravel())]
    differs = 0
    for y, x in cases:
        y, x = np.asarray(y, np.float32), np.asarray(x, np.float32)
        want, got = strided_atan2(y, x), oracle.xmath("atan2_glibc", y, x)
        same = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (y[~same][:4], x[~same][:4], got[~same][:4], want[~same][:4])
        differs += int((bits(got) != bits(oracle.xmath("atan2", y, x))).sum())
    assert differs > 1000          # and it is NOT the SLEEF routine of the contiguous path
