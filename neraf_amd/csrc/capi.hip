// Context management for libneraf_hip (C ABI, include/neraf_hip.h).
#include "common.h"

extern "C" int neraf_abi_version(void) { return NERAF_ABI_VERSION; }

extern "C" int neraf_ctx_create(neraf_ctx** out, int device) {
  if (!out) return NERAF_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return NERAF_ENOGPU;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return NERAF_EHIP;
  neraf_ctx* c = new (std::nothrow) neraf_ctx();
  if (!c) return NERAF_EINVAL;
  c->device = device;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  *out = c;
  return NERAF_OK;
}

extern "C" void neraf_ctx_destroy(neraf_ctx* ctx) { delete ctx; }

extern "C" const char* neraf_last_error(neraf_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

extern "C" int neraf_prof_enable(neraf_ctx* ctx, int on) {
  if (!ctx) return NERAF_EINVAL;
  for (auto& r : ctx->recs) { ctx->free_events.push_back(r.a); ctx->free_events.push_back(r.b); }
  ctx->recs.clear();
  ctx->prof = on != 0;
  return NERAF_OK;
}

extern "C" int neraf_prof_summary(neraf_ctx* ctx, int kernel_id, double* total_ms, int* launches, double* work) {
  if (!ctx || kernel_id < 0 || kernel_id >= PROF_NUM_KERNELS) return NERAF_EINVAL;
  double ms = 0.0, w = 0.0; int n = 0;
  for (auto& r : ctx->recs) {
    if (r.kid != kernel_id) continue;
    NERAF_HIP_CHECK(ctx, hipEventSynchronize(r.b));
    float t = 0.f;
    NERAF_HIP_CHECK(ctx, hipEventElapsedTime(&t, r.a, r.b));
    ms += t; w += r.work; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (work) *work = w;
  return NERAF_OK;
}

extern "C" const char* neraf_prof_kernel_name(int kernel_id) {
  static const char* names[PROF_NUM_KERNELS] = {"gemm_f16_nt_pipe_kernel<128, 128, 4, *>", "gemm_f16_nt_pipe_kernel<64|128x64, 4, *>",
                                                      "proposal_density_kernel", "field_query_kernel",
                                                      "gemm_f16_nt_pipe_kernel<*, 4, conv loader 1|2>", "proposal_backward_kernel",
                                                      "field_backward_kernel", "field_scatter_kernel"};
  return (kernel_id >= 0 && kernel_id < PROF_NUM_KERNELS) ? names[kernel_id] : nullptr;
}
