// Host-only helpers of libscasml_hip.so that need no HIP header: the thread-local error string and fail().
// plan_host.cpp (plain C++, also built with -fsanitize=address,undefined by tests/test_host_sanitizers.py) includes only this.
#pragma once
#include <stdarg.h>
#include <stdio.h>

#include "scasml_hip.h"

namespace scasml {

char *error_buffer();  // thread-local, 512 bytes (plan_host.cpp)

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace scasml
