// Internal shared declarations for libneraf_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <new>
#include <string>
#include <vector>

#include "../../include/neraf_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { PROF_GEMM128 = 0, PROF_GEMM64 = 1, PROF_PROP_DENSITY = 2, PROF_FIELD_QUERY = 3, PROF_CONV = 4, PROF_PROP_BWD = 5, PROF_FIELD_BWD = 6,
       PROF_FIELD_SCATTER = 7, PROF_GEMM_WIDE = 8, PROF_WGRAD = 9, PROF_GEMM64_BF16 = 10, PROF_GEMM12864 = 11,
       PROF_CONV64_BF16 = 12, PROF_CONV12864 = 13, PROF_CONV128 = 14, PROF_CONV_STEM = 15, PROF_NUM_KERNELS = 16 };

struct ProfRec { hipEvent_t a, b; int kid; double work; double exec; };   // work: ALGORITHMIC FLOPs / bytes; exec: as executed (padding, zero taps)

struct GraphEntry { uint64_t key; hipGraphExec_t exec; uint64_t last_use; };
// One launch of a ResNet3D sequence as the library sees it (neraf_manifest_*, tools/resnet_node_roofline.py): algorithmic FLOPs and the
// bytes the launch must read / write as designed (operands once, results once; split-K slabs count where they are written and read).
struct NodeRec { std::string name; double flops, rbytes, wbytes; };

struct neraf_ctx {
  int device;
  int num_cus;
  std::string last_error;
  bool prof = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> free_events;
  // hipGraph cache for the launch-bound call sequences (ResNet3D forward / backward: 100-270 kernels of a few microseconds)
  bool graphs_enabled = true;          // NERAF_GRAPHS=0 disables; any capture failure disables for the context's lifetime
  hipStream_t capture_stream = nullptr;
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  int graph_captures = 0, graph_launches = 0;
  bool manifest = false;               // neraf_manifest_enable: launches are recorded (and run un-graphed) instead of replayed
  std::vector<NodeRec> nodes;
};
static inline void neraf_node(neraf_ctx* ctx, const char* name, double flops, double rbytes, double wbytes) {
  if (ctx && ctx->manifest) ctx->nodes.push_back(NodeRec{name, flops, rbytes, wbytes});
}

// FNV-1a over the argument values that end up in kernel arguments: the cache key of a captured call sequence
struct ArgHash {
  uint64_t h = 1469598103934665603ull;
  void bytes(const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; } }
  template <class T> void add(const T& v) { bytes(&v, sizeof(T)); }
  void ptrs(const void* const* a, int n) { for (int i = 0; i < n; ++i) add(a[i]); }
};

// RAII bracket: records an event pair around a launch when profiling is on.
struct ProfScope {
  neraf_ctx* ctx; hipStream_t st; ProfRec r; bool on;
  ProfScope(neraf_ctx* c, hipStream_t s, int kid, double work, double exec = -1.0) : ctx(c), st(s), on(c && c->prof) {
    if (!on) return;
    auto get = [&]() { hipEvent_t e; if (!c->free_events.empty()) { e = c->free_events.back(); c->free_events.pop_back(); }
                       else (void)hipEventCreate(&e); return e; };
    r.a = get(); r.b = get(); r.kid = kid; r.work = work; r.exec = exec >= 0.0 ? exec : work;
    (void)hipEventRecord(r.a, st);
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(r.b, st); ctx->recs.push_back(r); } }
};

static inline int neraf_fail(neraf_ctx* ctx, int code, const char* what) {
  if (ctx) ctx->last_error = what;
  return code;
}

#define NERAF_HIP_CHECK(ctx, expr)                                                  \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      char _b[512];                                                                 \
      snprintf(_b, sizeof(_b), "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return neraf_fail(ctx, NERAF_EHIP, _b);                                       \
    }                                                                               \
  } while (0)

// Run `body(stream)` -- a fixed sequence of launches whose kernel arguments are all determined by `key` -- through a cached
// hipGraph: captured once on a private stream (the legacy default stream cannot capture), replayed on the caller's stream.
// A dependent chain of small kernels costs 2.3 us per launch on the stream (host-bound) and 1.8 us inside a graph with no
// host cost (tools/microbench/graph_chain.hip).  Falls back to direct launches while profiling or after any failure.
template <class Body>
int neraf_run_graphed(neraf_ctx* ctx, hipStream_t user, uint64_t key, Body body) {
  if (!ctx || !ctx->graphs_enabled || ctx->prof || ctx->manifest) return body(user);
  const char* tr = getenv("NERAF_GRAPH_TRUNC");      // measurement only: replay just a prefix of the captured sequence
  const int trunc = tr ? atoi(tr) : -1;
  if (trunc >= 0) key = key * 1099511628211ull + (uint64_t)(trunc + 1);
  GraphEntry* hit = nullptr;
  for (auto& e : ctx->graphs) if (e.key == key) { hit = &e; break; }
  if (!hit) {
    if (!ctx->capture_stream && hipStreamCreateWithFlags(&ctx->capture_stream, hipStreamNonBlocking) != hipSuccess) {
      ctx->graphs_enabled = false; (void)hipGetLastError();
      return body(user);
    }
    if (hipStreamBeginCapture(ctx->capture_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
      ctx->graphs_enabled = false; (void)hipGetLastError();
      return body(user);
    }
    const int rc = body(ctx->capture_stream);
    hipGraph_t g = nullptr;
    const hipError_t ee = hipStreamEndCapture(ctx->capture_stream, &g);
    hipGraphExec_t ex = nullptr;
    if (trunc >= 0 && ee == hipSuccess && g) {       // measurement only (tools/graph_prefix_times.py): keep the first `trunc` nodes
      size_t n = 0;
      if (hipGraphGetNodes(g, nullptr, &n) == hipSuccess && n > (size_t)trunc) {
        std::vector<hipGraphNode_t> nodes(n);
        if (hipGraphGetNodes(g, nodes.data(), &n) == hipSuccess)
          for (size_t i = n; i-- > (size_t)trunc;) (void)hipGraphDestroyNode(nodes[i]);
      }
    }
    if (rc != NERAF_OK || ee != hipSuccess || !g || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      ctx->graphs_enabled = false;
      return rc != NERAF_OK ? rc : body(user);       // nothing captured has run: do it directly
    }
    (void)hipGraphDestroy(g);
    if (ctx->graphs.size() >= 12) {                  // evict the least recently used
      size_t lru = 0;
      for (size_t i = 1; i < ctx->graphs.size(); ++i) if (ctx->graphs[i].last_use < ctx->graphs[lru].last_use) lru = i;
      (void)hipStreamSynchronize(user);              // rare path: the evicted graph's last replay must have drained
      (void)hipGraphExecDestroy(ctx->graphs[lru].exec);
      ctx->graphs.erase(ctx->graphs.begin() + lru);
    }
    ctx->graphs.push_back(GraphEntry{key, ex, 0});
    hit = &ctx->graphs.back();
    ++ctx->graph_captures;
  }
  hit->last_use = ++ctx->graph_clock;
  ++ctx->graph_launches;
  NERAF_HIP_CHECK(ctx, hipGraphLaunch(hit->exec, user));
  return NERAF_OK;
}

// Zero-fill by a kernel (16-byte aligned pointer, size a multiple of 4): used instead of hipMemsetAsync inside the sequences
// that are captured into hipGraphs -- memset NODES did not reliably order against the following kernel nodes on replay.
__global__ void neraf_zero_kernel(unsigned* __restrict__ p, size_t n_words);
static inline void neraf_zero_async(hipStream_t st, void* p, size_t bytes) {
  const size_t words = bytes / 4;
  if (!words) return;
  size_t blocks = (words + 4 * 256 - 1) / (4 * 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(neraf_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)p, words);
}

// up to three ranges in ONE launch (every launch costs >= ~4.5 us on the stream or in a graph, whatever it does)
__global__ void neraf_zero3_kernel(unsigned* __restrict__ p0, size_t n0, unsigned* __restrict__ p1, size_t n1, unsigned* __restrict__ p2, size_t n2);
static inline void neraf_zero3_async(hipStream_t st, void* p0, size_t b0, void* p1, size_t b1, void* p2, size_t b2) {
  const size_t w0 = b0 / 4, w1 = b1 / 4, w2 = b2 / 4;
  const size_t words = w0 + w1 + w2;
  if (!words) return;
  size_t blocks = (words + 256 - 1) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(neraf_zero3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)p0, w0, (unsigned*)p1, w1, (unsigned*)p2, w2);
}

// NERAF_DETERMINISTIC=1 (read once per process): every floating-point sum whose order the hardware schedules -- BatchNorm statistics
// and BatchNorm-backward sums, bias gradients, the average pool, d feat, the appearance-embedding gradient -- is formed from
// per-workgroup partials in a FIXED order instead of by atomics (the hash-table gradients already are integer sums).  A training
// run is then bit-reproducible; costs launches and bandwidth (tests / debugging: the bench keeps the atomics).
static inline bool neraf_deterministic() {
  static const bool d = [] { const char* e = getenv("NERAF_DETERMINISTIC"); return e && atoi(e) != 0; }();
  return d;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// ---- fp16 MFMA GEMM with fused epilogue (gemm_f16.hip) ----------------------------------
// C = epi(alpha * A[M,K] . B[N,K]^T) ; A and B are fp16, K-contiguous ("NT" form).
// Buffers are tile-padded: A has >= Mpad rows, B has >= Npad rows, K % 64 == 0.  Loads are
// unguarded; every fp16 output tile is written in full with zeros outside the logical MxN
// extent so padded buffers are always valid K-/M-padding for the next GEMM.
enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_TANH10 = 2, ACT_RELU = 3 };

// Implicit-GEMM convolution geometry (cubic volumes, channels-last activations [D^3][cin] fp16).
// loader 0: plain GEMM (A is a matrix);  1: cin % 64 == 0, one filter tap per 64-wide K-step;
// 2: cin == 8, one filter tap per 16-byte chunk (the 7->64 channel stem, NeRAF_resnet3d.py:120).
struct ConvGeom {
  int loader;
  int din, dout, stride, pad, ksize, cin;   // din = edge of the SOURCE tensor, dout = edge of the result
  int tflip;     // transposed convolution (dgrad): source offset is -tap instead of +tap
  int tstride;   // transposed convolution of a stride-2 conv: a tap contributes only where (z + pad - tap) is even
  int tclass;    // (set by the dispatcher) tstride == 2 with PARITY-CLASS row order: M-tile t holds 64 result voxels of parity class
                 // t & 7 = (z&1, y&1, x&1), so the tile walks only the taps that reach that class (27 -> 1, 2, 4 or 8 of a 3x3x3 filter;
                 // 27/8 on average instead of 27 with 7/8 of the rows fed from the zero page); the epilogue maps rows back to voxels
  const half_t* zero_page;   // >= 16 bytes of zeros: source of every out-of-bounds / padding-tap chunk
};

struct GemmParams {
  const half_t* A; int lda;
  const half_t* B; int ldb;
  int M, N, K;          // logical extents (K multiple of 64)
  int Mpad, Npad;       // tile-padded extents; multiples of the chosen tile
  float alpha;
  const float* alpha_dev; // optional device scalar multiplied into alpha (backward un-scaling)
  const float* bias;    // [Npad] fp32 or null (added before act)
  int act;
  const half_t* lmask; int ldmask;   // optional: v *= (lmask[m][n] > 0 ? 1 : slope)  (leaky/relu backward)
  float mask_slope;
  const half_t* add16; int ldadd;    // optional fp16 matrix added to the result before the stores (gradient accumulation)
  half_t* C16; int ldc16;            // optional row-major fp16 out [Mpad, >=Npad]
  half_t* C16T; int ldc16t;          // optional transposed fp16 out [Npad, >=Mpad]
  float* C32; int ldc32;             // optional fp32 out, masked to M x N
  int c32_beta;                      // != 0: C32 += result instead of C32 = result (a second producer of the same gradient)
  float* colsum;                     // optional [Npad] fp32: atomically += column sums of the final values
  float* colsumsq;                   // optional [Npad] fp32: atomically += column sums of squares (BatchNorm statistics)
  int stat_rep, stat_stride;         // colsum/colsumsq are replicated stat_rep (power of two, 0 = 1) times, stat_stride floats apart; a
                                     // workgroup adds into replica (m-tile % stat_rep) -- same-line atomics serialise in the memory system
  int stat_det;                      // deterministic statistics (NERAF_DETERMINISTIC=1): every 32-row block of the result owns a slot
                                     // (slot = first row / 32, stat_stride floats apart, zero-initialised by the caller) and the workgroup
                                     // that holds the rows STORES its column sums there -- no atomics; the consumer adds the slots in a
                                     // fixed order (slot_sum_kernel), so the statistics are bit-reproducible run to run
  // fused BatchNorm-backward reduction (dgrad GEMMs of the small ResNet3D layers): the result g is the gradient w.r.t. a
  // post-activation tensor whose producer is BatchNorm + ReLU; with bnb_x set, g *= (bnb_mask > 0) AFTER add16, and colsum / colsumsq
  // receive sum g and sum g * xhat (xhat = (bnb_x - mean) * rsqrt(var + 1e-5), mean / var = bnb_fin[c] / bnb_fin[bnb_cpad + c])
  // instead of sum / sum of squares -- what bn_bwd_reduce_kernel computes in a launch of its own
  const half_t* bnb_x; int ldbnb;    // pre-BN conv output fp16 [Mpad][N]
  const half_t* bnb_mask;            // post-activation tensor in the GEMM's element type [Mpad][N] (ld = ldbnb), or null
  const float* bnb_fin; int bnb_cpad;
  ConvGeom conv;                     // conv.loader == 0 for a plain GEMM
  float* splitk_ws; size_t splitk_ws_bytes;   // optional fp32 scratch enabling split-K for under-filled grids
  int tile_n;                        // 0 = auto; 64 forces the 128x64 tile (Cout = 64 layers)
  double alg_flops;                  // ALGORITHMIC FLOPs of the operation this GEMM implements (a convolution's
                                     // 2 dout^3 taps cin cout, also for its transposed form); 0 = 2 M N K.  Profiling only.
  int bf16;                          // 1: every 16-bit operand/result (A, B, lmask, add16, C16, C16T) is bfloat16 (exported for tests)
  // grouped launch (plain loader, fp32 results only): ngroups > 1 runs ngroups GEMMs of identical padded shape (Mpad, Npad, K,
  // lda, ldb) in one grid; group g takes A/B/C32/M/N/ldc32 from grp[g].  Used for the small-output weight gradients.
  int ngroups;
  struct Group { const half_t* A; const half_t* B; float* C32; int M, N, ldc32; } grp[6];
};

int launch_gemm_f16(neraf_ctx* ctx, const GemmParams& p, hipStream_t stream);

// ---- grouped weight-gradient launch (gemm_f16.hip) ---------------------------------------------------------------------------
// All convolution weight gradients of a backward pass are independent of one another and of the dgrad chain: they are described
// here and computed by TWO launches at the end of the pass (the TN GEMM over every (convolution, tile, K-split), then one
// reducer that also writes the PyTorch layout [cout][cin_real][taps]) instead of up to three launches per convolution.
struct WgradItem {
  const void* dy;        // [K rows = voxels (padded)][cout] gradient w.r.t. the conv output (fp16, or bf16 with `bf16` below)
  const void* x;         // [din^3][cin] the conv input in the same 16-bit type (fp16: the forward's own activation tensor)
  float* out;            // fp32 [cout][cin_real][taps]
  int cout, cin, cin_real, ksize, stride, pad, din, dout;
  int K;                 // voxel rows of dy (multiple of 64)
  int alpha_idx;         // the result is multiplied by alpha_dev[alpha_idx] (1 / scale of this item's dy tensor: the fp16 chain's groups)
};
int launch_wgrad_grouped(neraf_ctx* ctx, const WgradItem* items, int n, const half_t* zero_page, float* slab_ws, size_t slab_bytes,
                         const float* alpha_dev, hipStream_t stream, bool bf16 = false);

constexpr int kAmaxRep = 32, kAmaxStride = 64;      // amax replicas per scale group of the fp16 gradient chain, 256 bytes apart (same-line atomics serialise)

