// Version / error plumbing of the C ABI (include/vidtome_hip.h).
#include "common.h"
#include "ablate.h"

namespace vtm {
char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace vtm

VTM_EXPORT int vtm_version(void) { return VTM_ABI_VERSION; }
VTM_EXPORT const char *vtm_last_error(void) { return vtm::err_buf(); }
VTM_EXPORT int64_t vtm_pad_rows(int64_t n) { return vtm::cdiv(n, VTM_MATCH_ROW_PAD) * VTM_MATCH_ROW_PAD; }
VTM_EXPORT int64_t vtm_pad_k(int64_t C) { return vtm::cdiv(C, VTM_MATCH_K_PAD) * VTM_MATCH_K_PAD; }

// Which ablation switches (ablate.h) the library was built with, as a bit mask over the translation units that have any:
// 0 for the shipped build.  tests/test_host.py asserts it on the library the tests load.
namespace vtm {
int filter_ablations();     // match_filter.hip
int linear_ablations();     // linear.hip
}  // namespace vtm
VTM_EXPORT int vtm_build_ablations(void) { return VTM_ABLATIONS | vtm::filter_ablations() | vtm::linear_ablations(); }
