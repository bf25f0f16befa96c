// bca.hip -- placeholder until the device builder lands (next milestone).
#include "ge_common.h"
extern "C" {
ge_status ge_bca_build(const ge_csr *, const ge_csr *, const ge_bca_cfg *, ge_coo **result) {
    if (result) *result = nullptr;
    return ge::fail(GE_ERR_STATE, "ge_bca_build: device builder not built into this library yet");
}
ge_status ge_coo_get(const ge_coo *, int64_t *, const int32_t **, const int32_t **, const float **, const int64_t **, double *) {
    return ge::fail(GE_ERR_STATE, "ge_coo_get: device builder not built into this library yet");
}
void ge_coo_destroy(ge_coo *) {}
}
