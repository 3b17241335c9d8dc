// Context management for libneraf_hip (C ABI, include/neraf_hip.h).
#include "common.h"
#include <algorithm>
#include <stdlib.h>

__global__ void neraf_zero_kernel(unsigned* __restrict__ p, size_t n_words) {
  const size_t n4 = n_words / 4;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<u4*>(p)[i] = u4{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) p[n4 * 4 + threadIdx.x] = 0u;
}

__global__ void neraf_zero3_kernel(unsigned* __restrict__ p0, size_t n0, unsigned* __restrict__ p1, size_t n1, unsigned* __restrict__ p2, size_t n2) {
  const size_t total = n0 + n1 + n2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    if (i < n0) p0[i] = 0u;
    else if (i < n0 + n1) p1[i - n0] = 0u;
    else p2[i - n0 - n1] = 0u;
  }
}

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
  const char* g = getenv("NERAF_GRAPHS");
  c->graphs_enabled = !(g && g[0] == '0');
  *out = c;
  return NERAF_OK;
}

extern "C" void neraf_ctx_destroy(neraf_ctx* ctx) {
  if (!ctx) return;
  for (auto& e : ctx->graphs) (void)hipGraphExecDestroy(e.exec);
  if (ctx->capture_stream) (void)hipStreamDestroy(ctx->capture_stream);
  delete ctx;
}

extern "C" int neraf_graph_stats(neraf_ctx* ctx, int* captures, int* launches) {
  if (!ctx) return NERAF_EINVAL;
  if (captures) *captures = ctx->graph_captures;
  if (launches) *launches = ctx->graph_launches;
  return ctx->graphs_enabled ? 1 : 0;
}

// Launch manifest of the ResNet3D sequences (measurement aid): while enabled, the forward / backward run un-graphed and every launch
// appends {kernel, algorithmic FLOPs, bytes read, bytes written}; tools/resnet_node_roofline.py zips the list with a rocprofv3 trace.
extern "C" int neraf_manifest_enable(neraf_ctx* ctx, int on) {
  if (!ctx) return NERAF_EINVAL;
  ctx->nodes.clear();
  ctx->manifest = on != 0;
  return NERAF_OK;
}

extern "C" int neraf_manifest_get(neraf_ctx* ctx, int index, char* name, int name_cap, double* flops, double* rbytes, double* wbytes) {
  if (!ctx) return -1;
  const int n = (int)ctx->nodes.size();
  if (index < 0 || index >= n) return n;
  const NodeRec& r = ctx->nodes[index];
  if (name && name_cap > 0) { snprintf(name, (size_t)name_cap, "%s", r.name.c_str()); }
  if (flops) *flops = r.flops;
  if (rbytes) *rbytes = r.rbytes;
  if (wbytes) *wbytes = r.wbytes;
  return n;
}

extern "C" const char* neraf_last_error(neraf_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

extern "C" int neraf_prof_enable(neraf_ctx* ctx, int on) {
  if (!ctx) return NERAF_EINVAL;
  for (auto& r : ctx->recs) { ctx->free_events.push_back(r.a); ctx->free_events.push_back(r.b); }
  ctx->recs.clear();
  ctx->prof = on != 0;
  return NERAF_OK;
}

// Elapsed time of an EMPTY event pair on `stream` (median of 64): what every ProfScope interval contains besides its kernel.
// bench.py subtracts it per launch, so that intervals of few-microsecond kernels agree with rocprofv3's kernel durations.
extern "C" int neraf_prof_event_overhead(neraf_ctx* ctx, void* scratch_word, neraf_stream_t stream, double* ms) {
  if (!ctx || !ms || !scratch_word) return NERAF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  constexpr int N = 64;
  hipEvent_t ev[2 * N];
  for (int i = 0; i < 2 * N; ++i) NERAF_HIP_CHECK(ctx, hipEventCreate(&ev[i]));
  for (int i = 0; i < N; ++i) {
    neraf_zero_async(st, scratch_word, 4);                           // keeps the queue non-empty between pairs, as in a real step
    NERAF_HIP_CHECK(ctx, hipEventRecord(ev[2 * i], st));
    NERAF_HIP_CHECK(ctx, hipEventRecord(ev[2 * i + 1], st));
  }
  NERAF_HIP_CHECK(ctx, hipStreamSynchronize(st));
  float t[N];
  for (int i = 0; i < N; ++i) NERAF_HIP_CHECK(ctx, hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
  for (int i = 0; i < 2 * N; ++i) (void)hipEventDestroy(ev[i]);
  std::sort(t, t + N);
  *ms = t[N / 2];
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

extern "C" int neraf_prof_summary_ex(neraf_ctx* ctx, int kernel_id, double* total_ms, int* launches, double* work, double* exec_work) {
  if (!ctx || kernel_id < 0 || kernel_id >= PROF_NUM_KERNELS) return NERAF_EINVAL;
  double e = 0.0;
  for (auto& r : ctx->recs) if (r.kid == kernel_id) e += r.exec;
  if (exec_work) *exec_work = e;
  return neraf_prof_summary(ctx, kernel_id, total_ms, launches, work);
}

extern "C" const char* neraf_prof_kernel_name(int kernel_id) {
  // the rocprofv3 kernel-name prefix each scope covers (template arguments that vary inside a scope are written as *)
  static const char* names[PROF_NUM_KERNELS] = {"gemm_f16_nt_pipe_kernel<128, 128, *, 0, 1, *>", "gemm_f16_nt_pipe_kernel<64|32, 64|32, 4, 0, 1, false>",
                                                      "proposal_density_kernel | proposal_density_frame_kernel", "field_query_kernel | field_query_frame_kernel",
                                                      "gemm_f16_nt_pipe_kernel<64, 64, 4, 1, *, false>", "proposal_backward_kernel",
                                                      "field_backward_kernel", "field_scatter_kernel | field_slice_ids_kernel + field_scatter_owner_kernel",
                                                      "gemm_f16_nt_wide_kernel<*, 160|128, 3, *>", "wgrad_wide_tn_kernel | wgrad_grouped_tn_kernel",
                                                      "gemm_f16_nt_pipe_kernel<64|32, 64|32, 4, 0, 1, true>", "gemm_f16_nt_pipe_kernel<128, 64, 3, 0, 1, *>",
                                                      "gemm_f16_nt_pipe_kernel<64, 64, 4, 1, *, true>", "gemm_f16_nt_pipe_kernel<128, 64, 3, 1, *, *>",
                                                      "gemm_f16_nt_pipe_kernel<128, 128, 2, 1, *, *>", "gemm_f16_nt_pipe_kernel<128, 64, 3, 2, 5, false>"};
  return (kernel_id >= 0 && kernel_id < PROF_NUM_KERNELS) ? names[kernel_id] : nullptr;
}
