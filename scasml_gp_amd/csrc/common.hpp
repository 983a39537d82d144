// Shared host-side helpers of libscasml_hip.so: thread-local error string, HIP error mapping.
#pragma once
#include <hip/hip_runtime.h>

#include "host_common.hpp"

namespace scasml {

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
