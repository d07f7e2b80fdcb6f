// Shared host-side plumbing for libvisfly_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "visfly_amd.h"

namespace vf {

char* err_buf();  // thread-local message buffer behind vf_last_error()

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define VF_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return vf::fail(VF_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                            \
    } while (0)

inline hipStream_t as_stream(vf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kBlock = 256;  // 4 wave64 per workgroup

inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

}  // namespace vf
