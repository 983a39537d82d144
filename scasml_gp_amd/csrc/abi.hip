// ABI bookkeeping of libscasml_hip.so: version and the thread-local error string.
#include "common.hpp"

namespace scasml {
char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace scasml

extern "C" int scasml_abi_version(void) { return SCASML_ABI_VERSION; }
extern "C" const char *scasml_last_error(void) { return scasml::error_buffer(); }
extern "C" size_t scasml_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(scasml_problem);
        case 1: return sizeof(scasml_rng);
        case 2: return sizeof(scasml_term);
        case 3: return sizeof(scasml_plan);
        case 4: return sizeof(scasml_gp_model);
    }
    return 0;
}
