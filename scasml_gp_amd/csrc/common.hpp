// Shared host-side helpers of libscasml_hip.so: thread-local error string, HIP error mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "scasml_hip.h"

namespace scasml {

char *error_buffer();  // thread-local, 512 bytes (abi.hip)

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SCASML_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline int ceil_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace scasml
