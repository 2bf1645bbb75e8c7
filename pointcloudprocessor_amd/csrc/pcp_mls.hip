// pcp_mls.hip -- placeholder until the MLS kernels land (next commit).
#include "pcp_internal.hpp"
using namespace pcp;
extern "C" {
int pcp_mls_process(pcp_context *ctx, const pcp_mls_params *, int64_t *) {
  return set_error(ctx, PCP_ERR_STATE, "pcp_mls_process: not built yet");
}
int pcp_mls_fetch(pcp_context *ctx, int64_t, float *, float *, float *, int32_t *) {
  return set_error(ctx, PCP_ERR_STATE, "pcp_mls_fetch: not built yet");
}
int pcp_sor(pcp_context *ctx, int32_t, double, uint8_t *, int64_t *) {
  return set_error(ctx, PCP_ERR_STATE, "pcp_sor: not built yet");
}
}
