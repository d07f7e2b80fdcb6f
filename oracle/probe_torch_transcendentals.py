"""TEST INFRASTRUCTURE -- which routine does torch's CPU path run for atan2 / sin / cos / acos?

Run in the build container (needs gcc, an AVX-512 host and torch's own libtorch_cpu.so, which exports the SLEEF symbols its
Vectorized<float> wrappers name).  Compiles a 20-line shim that calls Sleef_{sin,cos,acos,atan2}f16_{u10,u35} directly and
compares, on 4M random arguments each: torch's result, the SLEEF symbols, and the restatements of oracle/vf_sleef.h.

Result on this image (torch 2.10.0, SLEEF 3.8, MKL 2024.2, AVX-512), recorded in DESIGN.md:
    torch.atan2 == Sleef_atan2f16_u10 == vfs_atan2f_u10               (0 mismatches; scalar glibc tail for n % 32 != 0)
    vfs_{sin,cos,acos}f_u10 == Sleef_{sin,cos,acos}f16_u10            (0 mismatches: the restatement is exact)
    torch.sin / cos != Sleef_*_u35 (22 % / 27 % differ) and != Sleef_*_u10 (1.9 % / 2.3 % differ)
    torch.acos != Sleef_acosf16_u10 (8.3 % differ)
    contiguous and strided inputs give identical torch results          -> one routine: MKL VML (vsSin / vsCos / vsAcos)
    a correctly rounded sin differs from torch.sin for 4.8 %              -> SLEEF u10 is the closest published candidate
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

SHIM = r"""
#include <immintrin.h>
__m512 Sleef_sinf16_u35(__m512); __m512 Sleef_sinf16_u10(__m512); __m512 Sleef_cosf16_u35(__m512); __m512 Sleef_cosf16_u10(__m512);
__m512 Sleef_acosf16_u10(__m512); __m512 Sleef_atan2f16_u10(__m512, __m512);
#define U(name, fn) void name(const float* x, float* o, long n) { for (long i = 0; i < n; i += 16) _mm512_storeu_ps(o + i, fn(_mm512_loadu_ps(x + i))); }
U(s_sin35, Sleef_sinf16_u35) U(s_sin10, Sleef_sinf16_u10) U(s_cos35, Sleef_cosf16_u35) U(s_cos10, Sleef_cosf16_u10) U(s_acos10, Sleef_acosf16_u10)
void s_atan2(const float* y, const float* x, float* o, long n) { for (long i = 0; i < n; i += 16) _mm512_storeu_ps(o + i, Sleef_atan2f16_u10(_mm512_loadu_ps(y + i), _mm512_loadu_ps(x + i))); }
"""


def main():
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    print(f"torch {torch.__version__}, cpu capability {torch.backends.cpu.get_cpu_capability()}, mkl {torch.backends.mkl.is_available()}")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "shim.c"), "w").write(SHIM)
        subprocess.check_call(["gcc", "-O2", "-mavx512f", "-fPIC", "-shared", "shim.c", "-o", "shim.so", "-L" + tl, "-ltorch_cpu",
                               "-Wl,-rpath," + tl], cwd=d)
        S = C.CDLL(os.path.join(d, "shim.so"))
        fp = C.POINTER(C.c_float)

        def sleef(name, *xs):
            o = np.empty_like(xs[0])
            getattr(S, name)(*[x.ctypes.data_as(fp) for x in xs], o.ctypes.data_as(fp), C.c_long(o.size))
            return o

        def diff(a, b):
            return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())

        rng = np.random.default_rng(0)
        n = 1 << 22
        x = rng.uniform(-7, 7, n).astype(np.float32)
        u = rng.uniform(-1, 1, n).astype(np.float32)
        y2, x2 = rng.normal(0, 3, n).astype(np.float32), rng.normal(0, 3, n).astype(np.float32)
        t = lambda f, *a: f(*[torch.from_numpy(v) for v in a]).numpy()
        strided = torch.from_numpy(np.stack([x, x], 1).copy())[:, 0]
        rows = [
            ("torch.atan2 vs Sleef_atan2f16_u10", diff(t(torch.atan2, y2, x2), sleef("s_atan2", y2, x2))),
            ("vfs_atan2f_u10 vs Sleef_atan2f16_u10", diff(oracle.xmath("atan2", y2, x2), sleef("s_atan2", y2, x2))),
            ("torch.sin vs Sleef_sinf16_u35", diff(t(torch.sin, x), sleef("s_sin35", x))),
            ("torch.sin vs Sleef_sinf16_u10", diff(t(torch.sin, x), sleef("s_sin10", x))),
            ("torch.cos vs Sleef_cosf16_u35", diff(t(torch.cos, x), sleef("s_cos35", x))),
            ("torch.cos vs Sleef_cosf16_u10", diff(t(torch.cos, x), sleef("s_cos10", x))),
            ("torch.acos vs Sleef_acosf16_u10", diff(t(torch.acos, u), sleef("s_acos10", u))),
            ("vfs_acosf_u10 vs Sleef_acosf16_u10", diff(oracle.xmath("acos", u), sleef("s_acos10", u))),
            ("vfs_sinf_u10 vs Sleef_sinf16_u10", diff(oracle.xmath("sin", x), sleef("s_sin10", x))),
            ("vfs_cosf_u10 vs Sleef_cosf16_u10", diff(oracle.xmath("cos", x), sleef("s_cos10", x))),
            ("torch.sin contiguous vs strided input", diff(t(torch.sin, x), torch.sin(strided).numpy())),
            ("torch.sin vs correctly rounded", diff(t(torch.sin, x), np.sin(x.astype(np.float64)).astype(np.float32))),
            ("vfs_sinf_u10 vs torch.sin", diff(oracle.xmath("sin", x), t(torch.sin, x))),
            ("vfs_cosf_u10 vs torch.cos", diff(oracle.xmath("cos", x), t(torch.cos, x))),
            ("vfs_acosf_u10 vs torch.acos", diff(oracle.xmath("acos", u), t(torch.acos, u))),
        ]
        for name, d_ in rows:
            print(f"  {name:44s} {d_:8d} of {n} differ ({100.0 * d_ / n:.2f} %)")


if __name__ == "__main__":
    main()
