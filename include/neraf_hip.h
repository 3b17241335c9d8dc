/* libneraf_hip -- C ABI of the MI355X-native NeRAF field-query engine.
 *
 * The reference (AmandineBtto/NeRAF) is pure Python and has no FFI of its own; the
 * entry points below are what a nerfstudio-side binding for the hot path would
 * bind.  Each function cites the reference interface it
 * replaces (paths relative to the reference's NeRAF/ package).
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *    a parameter is documented "host".
 *  - the caller owns every buffer (PyTorch's caching allocator in our host layer);
 *    the library allocates nothing persistent except the opaque neraf_ctx.
 *  - every launch is asynchronous on the hipStream_t passed in; no hidden syncs.
 *  - return 0 on success or a negative NERAF_E* code; never throws across the ABI;
 *    neraf_last_error(ctx) returns a description of the last failure on that ctx.
 *  - layouts: row-major contiguous, point-major [N,.]; voxel grid [7,X,Y,Z] as in
 *    NeRAF_model.py:271-277.
 */
#ifndef NERAF_HIP_H
#define NERAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct neraf_ctx neraf_ctx;
typedef void* neraf_stream_t; /* hipStream_t */

enum {
  NERAF_OK = 0,
  NERAF_EINVAL = -1, /* bad argument / unsupported shape */
  NERAF_EHIP = -2,   /* a HIP runtime call failed */
  NERAF_ENOGPU = -3, /* no gfx950 device */
  NERAF_ESTATE = -4  /* a numerical state the call depends on could not be established (fp16 gradient-chain calibration) */
};

#define NERAF_ABI_VERSION 1

int neraf_abi_version(void);
int neraf_ctx_create(neraf_ctx** out, int device);
void neraf_ctx_destroy(neraf_ctx* ctx);
const char* neraf_last_error(neraf_ctx* ctx);

/* ------------------------------------------------------------------------------------
 * Per-kernel timing (measurement only; off by default).  When enabled, every launch of a
 * tracked kernel family is bracketed by a hipEvent pair on the launch stream and its
 * algorithmic work (FLOPs or bytes, logical un-padded extents) is recorded.
 * neraf_prof_summary synchronises those events and returns totals since the last enable.
 * kernel ids: 0 = gemm_f16 128x128 tile (work = FLOPs), 1 = gemm_f16 64x64 tile (FLOPs),
 * 2 = proposal_density (work = gathered hash-table bytes), 3 = field_query (gathered bytes),
 * 4 = implicit-GEMM convolution instances of gemm_f16 (algorithmic FLOPs; neraf_prof_summary_ex adds the executed count),
 * 5 = proposal_backward (gathered + atomically added table bytes), 6 = field_backward (MLP chain; work =
 * gathered table bytes), 7 = field_scatter (work = 8 bytes per (sample, level, corner) 64-bit atomic);
 * neraf_prof_kernel_name(id) returns NULL past the end.
 * ---------------------------------------------------------------------------------- */
/* hipGraph cache statistics: the launch-bound call sequences (neraf_resnet3d_fwd / _bwd: 100-270 kernels of a few
 * microseconds each) are captured once per distinct argument set and replayed with hipGraphLaunch (NERAF_GRAPHS=0 in the
 * environment disables; profiling disables).  Returns 1 if graphs are enabled, 0 if not. */
int neraf_graph_stats(neraf_ctx* ctx, int* captures, int* launches);
int neraf_prof_enable(neraf_ctx* ctx, int on);
int neraf_prof_summary(neraf_ctx* ctx, int kernel_id, double* total_ms, int* launches, double* work);
/* As neraf_prof_summary, plus the EXECUTED work of the same launches (SURVEY 8d asks for both): `work` is the algorithmic count
 * (a convolution, its transposed form and its weight gradient are each 2 dout^3 taps cin cout FLOPs with the real channel / tap
 * counts), `exec_work` what the grid multiplied (K padded to the tile, channels padded to 8, zero-page taps at the volume's faces,
 * padded voxel rows).  Equal for plain GEMMs and for the byte-priced gather kernels. */
int neraf_prof_summary_ex(neraf_ctx* ctx, int kernel_id, double* total_ms, int* launches, double* work, double* exec_work);
/* Median elapsed time (ms) of an empty HIP event pair on `stream` behind a 4-byte fill of `scratch_word` (device): the part of
 * every profiled interval that is not the kernel; callers subtract it per launch. */
int neraf_prof_event_overhead(neraf_ctx* ctx, void* scratch_word, neraf_stream_t stream, double* ms);
const char* neraf_prof_kernel_name(int kernel_id);

/* ------------------------------------------------------------------------------------
 * Generic fp16 MFMA GEMM with fused epilogue:  C = epi(alpha * A[M,K] . B[N,K]^T)
 * (replaces the cuBLAS nn.Linear GEMMs behind NeRAF_field.py:49-58; exported for tests
 * and for callers that want the raw contraction).  A/B are fp16, K-contiguous; rows are
 * padded to Mpad/Npad (multiples of 128) and K is a multiple of 64.  act: 0 none,
 * 1 LeakyReLU(0.1), 2 tanh*10, 3 ReLU.
 * ---------------------------------------------------------------------------------- */
int neraf_gemm_f16(neraf_ctx* ctx, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                   int Mpad, int Npad, float alpha, const float* bias, int act,
                   void* C16, int ldc16, void* C16T, int ldc16t, float* C32, int ldc32,
                   neraf_stream_t stream);

/* The same contraction with bfloat16 operands / 16-bit results (used by the deep gradient chains). */
int neraf_gemm_bf16(neraf_ctx* ctx, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                    int Mpad, int Npad, float alpha, const float* bias, int act,
                    void* C16, int ldc16, void* C16T, int ldc16t, float* C32, int ldc32,
                    neraf_stream_t stream);
/* "TN" form, bfloat16, both operands K-major and dense: C32[m][n] = sum_k A[k][m] * B[k][n]  (A [K][M], B [K][N], C32 [M][N];
 * M, N, K multiples of 64).  This is the contraction behind the Conv3d weight gradients (cuDNN wgrad in the reference,
 * NeRAF_resnet3d.py:81-86 under autograd): the operand tiles are read from LDS with gfx950's hardware-transposed
 * ds_read_b64_tr_b16, so neither dY nor the activations are ever transposed or im2col'ed in HBM.  Inside the ResNet3D
 * backward all 43 weight gradients run as ONE grid of this kernel; this entry is its plain single-matrix case.
 * splitk_ws: fp32 scratch, at least 256 + 4*M*N bytes (more lets K be split over the chip). */
int neraf_gemm_bf16_tn(neraf_ctx* ctx, const void* A, const void* B, int M, int N, int K, float* C32, void* splitk_ws,
                       size_t splitk_bytes, neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * NAcF -- neural acoustic field MLP (NeRAFAudioSoundField, NeRAF_field.py:37-65) with the
 * query prologue of NeRAFAudioModel.get_outputs (NeRAF_model.py:531-564) fused in front
 * and the tanh*10 heads behind.
 * ---------------------------------------------------------------------------------- */
typedef struct neraf_nacf_desc {
  int n_feat;  /* grid feature width (1024, NeRAF_config.py:103); 0 = no grid (NeRAF_model.py:207) */
  int n_query; /* per-row encoded query width: 21+63+63+16 = 163 (NeRAF_model.py:169-171,560) */
  int W;       /* trunk output width W_field (512, NeRAF_config.py:106) */
  int C;       /* sound_rez: 1 RAF / 2 SoundSpaces (NeRAF_model.py:129,133) */
  int F;       /* N_frequencies: 513 RAF / 257 SoundSpaces */
  int dense_l0; /* also pack layer 0 for the dense h[B,n_feat+n_query] entry points (fwd_dense/bwd_dense) */
} neraf_nacf_desc;

/* Bytes of the packed fp16 weight blob (both orientations of every layer + fp32 biases). */
size_t neraf_nacf_packed_bytes(const neraf_nacf_desc* d);
/* Bytes of the activation workspace for a batch of B rows (saved for backward if training). */
size_t neraf_nacf_workspace_bytes(const neraf_nacf_desc* d, int B, int training);

/* Pack fp32 master weights (state-dict order: soundfield.{0..4}.{weight,bias}, then
 * STFT_linear.{c}.{weight,bias} for c<C; each a device pointer to the nn.Linear tensor,
 * NeRAF_field.py:41-45) into the fp16 MFMA layout.  `weights` is a HOST array of
 * 2*(5+C) device pointers. */
int neraf_nacf_pack_weights(neraf_ctx* ctx, const neraf_nacf_desc* d, const float* const* weights,
                            void* packed, neraf_stream_t stream);

/* Query prologue (NeRAF_model.py:533-551): time/(T-1), AABB-normalise + in-box selector,
 * NeRF encodings, SH deg-4 -> q fp16 [Bpad,192] (+ transposed copy for backward) inside
 * the workspace.  aabb is a HOST array of 6 floats (min xyz, max xyz). */
int neraf_nacf_encode_queries(neraf_ctx* ctx, const neraf_nacf_desc* d, const int64_t* time_query,
                              const double* mic_pose, const double* source_pose, const double* rot,
                              const float* aabb_host, int max_len, int B, void* workspace, int training,
                              neraf_stream_t stream);
/* pose_rows = B (one pose row per query, as above) or 1: every query shares row 0 of mic_pose / source_pose / rot -- the eval branch
 * (NeRAF_model.py:648-694), where one RIR is T time queries at ONE (microphone, source, orientation); saves the three [T,3] expansions. */
int neraf_nacf_encode_queries_ex(neraf_ctx* ctx, const neraf_nacf_desc* d, const int64_t* time_query,
                                 const double* mic_pose, const double* source_pose, const double* rot, int pose_rows,
                                 const float* aabb_host, int max_len, int B, void* workspace, int training,
                                 neraf_stream_t stream);

/* Forward with the layer-0 split: feat [n_feat] fp32 is the ResNet3D feature shared by all
 * rows (NeRAF_model.py:557-558); queries must have been encoded into the workspace.
 * out: fp32 [B, C, F] (NeRAF_field.py:63).  fp32 master weights are read for the
 * feature half of layer 0 (pointer array as in pack_weights). */
int neraf_nacf_fwd(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, const float* const* weights,
                   const float* feat, int B, float* out, void* workspace, int training, neraf_stream_t stream);

/* Dense forward on an explicit h [B, n_feat+n_query] fp32 -- exactly the signature of
 * NeRAFAudioSoundField.forward(h) (NeRAF_field.py:47). */
int neraf_nacf_fwd_dense(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, const float* h, int B,
                         float* out, void* workspace, int training, neraf_stream_t stream);

/* Backward of neraf_nacf_fwd: dout fp32 [B,C,F] -> grads (HOST array of 2*(5+C) device
 * pointers, same order/shapes as `weights`, overwritten) and dfeat [n_feat] (may be null). */
int neraf_nacf_bwd(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, const float* const* weights,
                   const float* feat, int B, const float* out, const float* dout, float* const* grads,
                   float* dfeat, void* workspace, neraf_stream_t stream);

/* Backward of neraf_nacf_fwd_dense; dh fp32 [B, n_feat+n_query] may be null. */
int neraf_nacf_bwd_dense(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, int B, const float* out,
                         const float* dout, float* const* grads, float* dh, void* workspace,
                         neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * STFT loss (STFTLoss.forward, NeRAF_evaluator.py:88-108; scaling NeRAF_model.py:584-600).
 * loss_type: 0 = 'mse' (SC+SLMSE), 1 = 'l1' (SC+SLL1).  sums (device, fp32[NERAF_STFT_SUMS_FLOATS] = 4 + 4 * 256, needs no
 * initialisation): [0..3] = sum (ymag-xmag)^2, sum ymag^2, sum |x-y|^p, 0; the rest holds per-workgroup partials that a second
 * launch adds in a fixed order (no atomics: the loss and its gradient are bit-reproducible).  losses (fp32[2], device):
 * {sc, mag}, multiplied by weights[0], weights[1] (device fp32[2], the loss factors of NeRAF_model.py:592-599) when weights != NULL.
 * The *_bwd writes d(total)/dpred into dpred with d total/d sc = *g_sc * weights[0] * extra and d total/d mag = *g_mag * weights[1]
 * * extra: g_sc / g_mag are DEVICE scalars (the upstream gradients, i.e. the grad-scaler scale; NULL = 0), weights as in the forward
 * (NULL = 1), extra a host factor (the data-parallel world size, see neraf_amd/losses.py) -- no host synchronisation and no scalar
 * torch ops between forward and backward.
 * ---------------------------------------------------------------------------------- */
#define NERAF_STFT_SUMS_FLOATS (4 + 4 * 256)
int neraf_stft_loss_fwd(neraf_ctx* ctx, const float* pred, const float* gt, size_t n, int loss_type,
                        float* sums, float* losses, neraf_stream_t stream);
/* The same in two steps for data-parallel training: the spectral-convergence ratio is over the
 * GLOBAL batch (NeRAF_evaluator.py:26), so ranks compute local sums, all-reduce the 4 floats
 * (RCCL), then finalize with the global element count n_total. */
int neraf_stft_loss_sums(neraf_ctx* ctx, const float* pred, const float* gt, size_t n, int loss_type,
                         float* sums, neraf_stream_t stream);
int neraf_stft_loss_finalize(neraf_ctx* ctx, const float* sums, size_t n_total, const float* weights, float* losses,
                             neraf_stream_t stream);
int neraf_stft_loss_bwd(neraf_ctx* ctx, const float* pred, const float* gt, size_t n, size_t n_total,
                        int loss_type, const float* sums, const float* g_sc, const float* g_mag, const float* weights, float extra,
                        float* dpred, neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Radiance half (forward).  Replaces what NeRAFVisionModel (NeRAF_model.py:54-79) inherits from
 * nerfstudio's NerfactoModel and executes through tiny-cuda-nn: ProposalNetworkSampler
 * (UniformLinDispPiecewiseSampler + PDFSampler), HashMLPDensityField, NerfactoField (reached via
 * NeRAFVisionFieldValue.forward, NeRAF_field.py:33-34, and from the grid refresh
 * NeRAF_model.py:333-350), RaySamples.get_weights and the RGB/Depth/Accumulation renderers.
 * Samples of a ray are described by their S+1 bin edges: s_bins (normalised spacing) and
 * e_bins (euclidean), both fp32 [R, S+1]; sample position = o + d * (e[i]+e[i+1])/2.
 * ---------------------------------------------------------------------------------- */
typedef struct neraf_grid_desc {
  int n_levels;          /* 16 main field / 5 proposal nets            */
  int base_res;          /* 16                                         */
  int max_res;           /* 2048 main / 128, 256 proposal              */
  int log2_hashmap_size; /* 19 main / 17 proposal                      */
  int n_features;        /* features per level: 2                      */
} neraf_grid_desc;

/* Level table of a tiny-cuda-nn style multiresolution hash grid (host helper so every layer of
 * the stack agrees): scales/resolutions/sizes [n_levels], offsets [n_levels+1] in table rows. */
int neraf_grid_layout(const neraf_grid_desc* g, float* scales, int* resolutions, uint32_t* sizes, uint32_t* offsets);

/* Multiresolution hash encoding on its own (tiny-cuda-nn HashGrid forward [TCNN-recall]; the fused kernels below contain it): x01 fp32
 * [n_points,3] in [0,1]^3 (positions are used as given: no contraction, no selector) -> enc fp32 [n_points, 2 * n_levels], level-major
 * pairs, the trilinear interpolation of the level's fp16 table entries (table_f16: fp16 [rows, 2], rows from neraf_grid_layout). */
int neraf_hash_encode(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const float* x01, long long n_points,
                      float* enc, neraf_stream_t stream);

/* Initial samples: S bins uniform in piecewise lin-disp spacing between near and far, optional
 * single jitter per ray: jitter [R] in [0,1) given by the caller, or -- jitter NULL and jitter_seed != 0 -- drawn inside the
 * kernel as a pure function of (jitter_seed, ray) (training: no random-number launch in front of the sampler; the host layer passes a
 * fresh seed per call and stage); both NULL / 0 = deterministic (eval).  The same pair of arguments in neraf_pdf_resample. */
int neraf_sample_uniform(neraf_ctx* ctx, int R, int S, float near, float far, const float* jitter, uint64_t jitter_seed,
                         float* s_bins, float* e_bins, neraf_stream_t stream);

/* Proposal density: contraction -> hash grid -> MLP(2L -> 16 -> 1) -> avg_density * exp.
 * table_f16: fp16 [rows, 2]; mlp_f16: fp16 [16*16 + 16] = layer-0 [hidden][input] then layer-1 row.
 * density: fp32 [R, S].
 * coherent_rays != 0 (here and in neraf_field_query) declares that CONSECUTIVE RAYS ARE NEIGHBOURS IN SPACE -- the pixels of one
 * camera in row-major order, i.e. the chunks of Model.get_outputs_for_camera (NeRAF_model.py:70-79) -- and changes only how samples
 * are assigned to lanes: a wavefront owns a TILE of 64 (proposal) / 16 (field) neighbouring rays and walks a run of 16 consecutive
 * sample indices, so at every step its lanes fall into the same few grid cells (a gather touches a handful of cache lines instead
 * of 64), the rays' data are loaded once per run and the outputs leave as whole cache lines.
 *   coherent_rays == 1: a tile is 64 / 16 consecutive rays (a pixel ROW segment);
 *   coherent_rays  > 1: coherent_rays is the IMAGE WIDTH: the R rays are whole rows of an image that wide (width % 8 == 0,
 *                       R % width == 0), and a tile is an 8 x 8 / 4 x 4 block of PIXELS.
 * Takes effect when S % 16 == 0 (nerfacto's 256 / 96 / 48 samples per ray), otherwise the call runs in the general order.
 * Results are bit-identical to coherent_rays == 0. */
int neraf_proposal_density(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                           const float* origins, const float* dirs, const float* e_bins, int R, int S,
                           float avg_density, int coherent_rays, float* density, neraf_stream_t stream);
/* The same with an explicit row stride of e_bins (floats): S + 1, or 0 = EVERY RAY SHARES ROW 0.  Without jitter the first sampler
 * stage (neraf_sample_uniform) gives every ray the same S + 1 edges: a frame render then generates ONE row instead of R (0.4 ms and
 * 2 x 34 MB per 32,768-ray chunk) and the first proposal query / PDF resampling read it with stride 0. */
int neraf_proposal_density_ex(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                              const float* origins, const float* dirs, const float* e_bins, int64_t e_row_stride, int R, int S,
                              float avg_density, int coherent_rays, float* density, neraf_stream_t stream);

/* get_weights + PDF resampling of n_new bins (n_new+1 edges) from annealed weights; weights
 * (fp32 [R,S], may be NULL) are the un-annealed volume-rendering weights of the S input bins. */
int neraf_pdf_resample(neraf_ctx* ctx, const float* density, const float* s_bins, const float* e_bins, int R, int S,
                       float anneal, const float* jitter, uint64_t jitter_seed, int n_new, float near, float far, float* weights,
                       float* s_new, float* e_new, neraf_stream_t stream);
/* ... with an explicit row stride of the INPUT bins s_bins / e_bins (floats): S + 1, or 0 (every ray shares row 0; see above) */
int neraf_pdf_resample_ex(neraf_ctx* ctx, const float* density, const float* s_bins, const float* e_bins, int64_t bins_row_stride,
                          int R, int S, float anneal, const float* jitter, uint64_t jitter_seed, int n_new, float near, float far,
                          float* weights, float* s_new, float* e_new, neraf_stream_t stream);

/* ... and the expected-depth clip range of the composite that follows folded into the sampler (nerfstudio's DepthRenderer clips the
 * expected depth to [steps.min(), steps.max()] of the batch [NS-recall]; NerfactoModel.get_outputs, reached from NeRAF_model.py:65-68):
 * minmax_scratch holds 64 REPLICAS of the pair {bits(min), bits(max)}, 256 bytes apart (ray r updates replica r & 63: atomics on one
 * line serialise in its L2 channel), i.e. 16384 bytes, followed by up to 60 further words.  minmax_mode 1 -- the launch also SEEDS the
 * replicas = {bits(+FLT_MAX), 0} and zeroes the words behind them (bytes 16384..scratch_bytes: the loss node's four fp32 sums,
 * neraf_render_loss); minmax_mode 2 -- the launch also ACCUMULATES {min, max} of 0.5 (e_new[0] + e_new[1]) and 0.5 (e_new[n_new-1] +
 * e_new[n_new]) over its rays (n_new <= 63); 0 = neraf_pdf_resample_ex.  The two sampler stages of a render call it with 1 then 2;
 * neraf_composite_mm reduces the replicas.  scratch_bytes: 16384..16624, a multiple of 4. */
int neraf_pdf_resample_mm(neraf_ctx* ctx, const float* density, const float* s_bins, const float* e_bins, int64_t bins_row_stride,
                          int R, int S, float anneal, const float* jitter, uint64_t jitter_seed, int n_new, float near, float far,
                          float* weights, float* s_new, float* e_new, void* minmax_scratch, size_t scratch_bytes, int minmax_mode,
                          neraf_stream_t stream);

/* Fused nerfacto field query: position map (mode 0: L-inf scene contraction, mode 1: AABB
 * normalisation with aabb_host[6]) -> 16-level hash grid -> base MLP -> density; SH(dir) +
 * appearance embedding -> colour MLP -> sigmoid.  wfrag_f16: 24 MFMA weight fragments (24 KB)
 * packed by the host layer; emb_f16: fp16 [rows,32]; avg_row >= 0 selects one row for every
 * sample (eval: the mean embedding), else cam_idx[R] is used.  rgb fp32 [R,S,3], density [R,S]. */
int neraf_field_query(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                      const void* emb_f16, const float* origins, const float* dirs, const float* e_bins,
                      const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host, float avg_density,
                      int avg_row, int coherent_rays, float* rgb, float* density, neraf_stream_t stream);

/* Training form of the query: additionally stores the interpolated encoding enc_out fp16 [R*S, 32] (required) and, when denc_out is
 * not NULL, its derivatives w.r.t. the mapped sample position denc_out fp16 [R*S, 4, 24] (lane quarter, then level x feature x axis).
 * neraf_field_backward_ex reads them back instead of walking the hash table a second time (and a third for the camera-pose edge). */
int neraf_field_query_train(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                            const void* emb_f16, const float* origins, const float* dirs, const float* e_bins,
                            const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host, float avg_density,
                            int avg_row, float* rgb, float* density, void* enc_out, void* denc_out, neraf_stream_t stream);

/* Weights + composite for S <= 64 samples per ray: rgb = sum w c + c_last (1 - sum w) (clipped to
 * [0,1], NeRAF_model.py:67), median depth, expected depth, accumulation.  scratch (device, scratch_bytes >= 8 when `expected` is
 * asked for, <= 248, multiple of 4): bytes 0..7 hold the batch's {min, max} step of the expected-depth clip; bytes 8.. are ZEROED
 * by the same launch that seeds them -- a training step's loss node takes its four fp32 `sums` (neraf_render_loss) from there. */
int neraf_composite(neraf_ctx* ctx, const float* density, const float* rgb, const float* e_bins, int R, int S,
                    int training, float* weights, float* rgb_out, float* depth, float* expected, float* acc,
                    void* scratch, size_t scratch_bytes, neraf_stream_t stream);
/* The same as ONE launch: `minmax` = the 64 replicated {min, max} step pairs the sampler left (neraf_pdf_resample_mm modes 1 and 2) -- no seeding
 * launch and no reduction launch in front of the composite (an eval frame is 22 chunks: 44 launches, 0.39 ms of 11.15). */
int neraf_composite_mm(neraf_ctx* ctx, const float* density, const float* rgb, const float* e_bins, int R, int S,
                       int training, float* weights, float* rgb_out, float* depth, float* expected, float* acc,
                       const void* minmax, neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Radiance half, training: losses (V4) and backward.  These replace autograd through nerfstudio's
 * NerfactoModel.get_loss_dict (rgb MSE on the clipped colour NeRAF_model.py:67, interlevel_loss x1.0,
 * distortion_loss x0.002) and tiny-cuda-nn's backward kernels.  `up3` is a DEVICE pointer to the upstream
 * gradients of {rgb_loss, interlevel_loss, distortion_loss} (NULL = compute loss values only);
 * `sums` (device, fp32[4], caller zeroes) accumulates {sum (rgb-gt)^2, sum_rays distortion,
 * sum outer-loss, -}; the host layer divides by 3R / R / (R*S2) and applies the multipliers.
 * neraf_render_loss with up3 == NULL and d_rgb_s != NULL computes the loss values AND unit gradients in the
 * same pass: d_rgb_s / d_density for d rgb_loss = 1 and d_density_dist for d distortion_loss = 1 (the
 * gradient is linear in the upstream scalars, so the backward pass is two scalings instead of a re-run).
 * ---------------------------------------------------------------------------------- */
int neraf_render_loss(neraf_ctx* ctx, const float* density, const float* rgb_s, const float* e_bins, const float* s_bins,
                      const float* gt_rgb, int R, int S, float distortion_mult, const float* up3, float* d_rgb_s,
                      float* d_density, float* d_density_dist, float* sums, neraf_stream_t stream);
int neraf_interlevel_loss(neraf_ctx* ctx, const float* c_bins, const float* w_fine, int S2, const float* p_bins,
                          const float* p_ebins, const float* p_density, int Sp, int R, float mult, const float* up3,
                          float* d_density, float* sums, neraf_stream_t stream);
/* Proposal network backward: d_density [R,S] -> table_grad fp32 [rows,2] and w_grad fp32 [16*16+16]
 * (layer 0 then layer-1 row), both ACCUMULATED (caller zeroes).  scratch (optional, >= 2048*272*4 bytes): per-workgroup
 * weight-gradient partials, folded by a second launch instead of 272 same-line atomics per workgroup.  With a scratch of
 * neraf_proposal_backward_scratch_bytes(R, S, n_levels) bytes the table gradient takes the packed two-pass path of the main field:
 * the MLP-backward kernel stores the per-sample encoding gradients and each level's gradient mass, the scatter adds both features of
 * an entry with one 64-bit fixed-point atomic (per-level power-of-two scale, overflow-proof, bit-reproducible), an in-place pass
 * converts to fp32 -- table_grad must then be ZERO on entry (it doubles as the integer accumulator). */
size_t neraf_proposal_backward_scratch_bytes(int R, int S, int n_levels);
int neraf_proposal_backward(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                            const float* origins, const float* dirs, const float* e_bins, const float* d_density, int R,
                            int S, float avg_density, float* table_grad, float* w_grad, void* scratch, size_t scratch_bytes,
                            neraf_stream_t stream);
/* Fused field backward.  d_rgb [R,S,3], d_density [R,S] (and the forward density) -> table_grad fp32
 * [rows,2] (MUST BE ZERO on entry: during the call it holds packed 64-bit fixed-point sums, one integer
 * atomic per table entry instead of two fp32 ones, unpacked in place before returning; the result is
 * bit-reproducible), emb_grad fp32 [n_emb,32] (ACCUMULATED, caller zeroes) and the five MLP weight gradients
 * w_grads (HOST array of device pointers {base_w0 [64,32], base_w1 [16,64], head_w0 [64,64], head_w1
 * [64,64], head_w2 [16,64]}, overwritten).  wfrag_bwd_f16: 28 transposed-weight MFMA fragments packed by
 * the host layer.  dump: scratch of neraf_field_backward_dump_bytes(R,S) bytes that the caller ZEROES ONCE
 * at allocation (its padding rows must stay zero); splitk_ws: >= 8 MB fp32 scratch. */
size_t neraf_field_backward_dump_bytes(int R, int S);
int neraf_field_backward(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                         const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                         const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                         float avg_density, int avg_row, const float* density, const float* d_rgb, const float* d_density,
                         float* table_grad, float* emb_grad, float* const* w_grads, void* dump, void* splitk_ws,
                         size_t splitk_bytes, neraf_stream_t stream);

/* The same call for batches in which runs of `pos_run` consecutive rays share their single sample position (S must be 1, R a
 * multiple of pos_run) -- the grid refresh queries every cell centre under 18 directions (NeRAF_model.py:327-339): the hash-grid
 * gradient of a run is summed before it is scattered.  pos_run = 1 is neraf_field_backward. */
int neraf_field_backward_runs(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                         const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                         const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                         float avg_density, int avg_row, const float* density, const float* d_rgb, const float* d_density,
                         float* table_grad, float* emb_grad, float* const* w_grads, void* dump, void* splitk_ws,
                         size_t splitk_bytes, int pos_run, neraf_stream_t stream);

/* The two backward calls with the camera-pose edge: additionally ACCUMULATE d loss / d (ray origin, ray direction) into d_rays
 * fp32 [R,6] (caller zeroes) -- the gradient nerfstudio's CameraOptimizer (NeRAF_config.py:97, SO3xR3) receives through
 * RaySamples.frustums.get_positions() / the SH direction encoding: hash-grid input gradient (tiny-cuda-nn computes it on request),
 * L-inf contraction Jacobian, sum over the ray's samples (sample distances are constants: the PDF sampler detaches its bins), and
 * for the main field the SH input gradient.  wfrag_bwd_f16 holds 28 fragments (the last two: head layer 0's SH columns). */
int neraf_field_backward_rays(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                              const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                              const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                              float avg_density, int avg_row, const float* density, const float* d_rgb, const float* d_density,
                              float* table_grad, float* emb_grad, float* const* w_grads, void* dump, void* splitk_ws,
                              size_t splitk_bytes, float* d_rays, neraf_stream_t stream);
/* The general form of the three calls above: pos_run as neraf_field_backward_runs, d_rays as neraf_field_backward_rays (NULL = none),
 * enc_saved / denc_saved = the buffers neraf_field_query_train filled for the SAME batch and parameters (NULL = recompute from the table;
 * denc_saved is only read together with d_rays). */
int neraf_field_backward_ex(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                            const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                            const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                            float avg_density, int avg_row, const float* density, const float* d_rgb, const float* d_density,
                            float* table_grad, float* emb_grad, float* const* w_grads, void* dump, void* splitk_ws,
                            size_t splitk_bytes, int pos_run, float* d_rays, const void* enc_saved, const void* denc_saved,
                            void* acc_scratch, int accumulate, int emb_rows, neraf_stream_t stream);
/* acc_scratch / accumulate / emb_rows (neraf_field_backward_ex; acc_scratch also neraf_proposal_backward_ex): how the outputs start.
 *   acc_scratch == NULL (and every other entry point of the family): table_grad doubles as the 64-bit fixed-point accumulator and
 *     is converted in place -- the CALLER zeroes table_grad and emb_grad before the call.
 *   acc_scratch != NULL: 8 bytes per table row, ZERO on entry, left ZERO on exit (the conversion pass clears what it reads), so one
 *     persistent buffer serves every call without a fill launch.  accumulate == 0: table_grad, the five w_grads and emb_grad
 *     [emb_rows, 32] are WRITTEN (need no initialisation); accumulate != 0: the call ADDS to all of them -- the second producer of
 *     the same parameters' gradients in one backward pass (NeRAF's grid refresh next to the render batch, NeRAF_model.py:395-400)
 *     adds in place instead of handing autograd a second set of tensors to sum. */
/* neraf_proposal_backward(_rays) with the outputs in the parameters' own layouts and nothing for the caller to zero: w0_grad fp32
 * [16,16], w1_grad fp32 [16,16] (row 0 = the used output row, rows 1..15 written as zeros), table_grad written through acc_scratch
 * (see above); d_rays NULL or the ACCUMULATED fp32 [R,6] ray gradients.  scratch: neraf_proposal_backward_scratch_bytes. */
int neraf_proposal_backward_ex(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                               const float* origins, const float* dirs, const float* e_bins, const float* d_density, int R, int S,
                               float avg_density, float* table_grad, float* w0_grad, float* w1_grad, void* scratch,
                               size_t scratch_bytes, void* acc_scratch, float* d_rays, neraf_stream_t stream);
int neraf_proposal_backward_rays(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                 const float* origins, const float* dirs, const float* e_bins, const float* d_density, int R,
                                 int S, float avg_density, float* table_grad, float* w_grad, void* scratch, size_t scratch_bytes,
                                 float* d_rays, neraf_stream_t stream);

/* Grid refresh epilogue of query_grid_one_batch (NeRAF_model.py:352-357,386,395-400): mean over the ndirs
 * view directions of rgb [ndirs*n,3] / density [ndirs*n] (direction-major), alpha = clip(1-exp(-delta d)),
 * written to channels 0..3 of grid fp32 [7, nvox] at flat cells [start, start+n). */
int neraf_grid_refresh_write(neraf_ctx* ctx, const float* rgb, const float* density, int n, int ndirs, float delta,
                             float* grid, size_t nvox, size_t start, neraf_stream_t stream);

/* The same epilogue as a differentiable node (the reference keeps the autograd edge of the refreshed cells,
 * NeRAF_model.py:395-400): vals fp32 [4][n] = (mean rgb, alpha); cell_major = 1: query k = i*ndirs + j, 0: k = j*n + i.
 * With grid != NULL the same launch also writes the values into channels 0..3 of grid fp32 [7, nvox] at flat cells [start, start+n)
 * (the detached slab write of :395-400).
 * The backward maps dvals [4][n] to d_rgb [n*ndirs,3] / d_density [n*ndirs] (alpha's clip passes no gradient where active). */
int neraf_grid_refresh_vals(neraf_ctx* ctx, const float* rgb, const float* density, int n, int ndirs, int cell_major, float delta,
                            float* vals, float* grid, size_t nvox, size_t start, neraf_stream_t stream);
int neraf_grid_refresh_vals_bwd(neraf_ctx* ctx, const float* dvals, const float* density, int n, int ndirs, int cell_major,
                                float delta, float* d_rgb, float* d_density, neraf_stream_t stream);

/* World positions of the refresh queries, cell-major (NeRAF_model.py:315, :327-333): out[i*ndirs + j] = coords[i] * (aabb1 - aabb0)
 * + aabb0 with coords fp32 [n,3] (a window of coordinates_to_render) and aabb_host = {min xyz, max xyz} on the host. */
int neraf_refresh_origins(neraf_ctx* ctx, const float* coords, int n, int ndirs, const float* aabb_host, float* out,
                          neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ResNet3D scene encoder (ResNet3D_helper / ResNet3D.forward, NeRAF_resnet3d.py:116-201,266-285;
 * backbone 'resnet50' truncated after layer3, N_features = 1024; called at NeRAF_model.py:554-558
 * and :680-684): voxel grid fp32 [7,S,S,S] (S = 128 for grid_step 1/128, 64 for 1/64, 256 for 1/256) -> feat fp32
 * [n_features] (1024: layers 1-3; 2048: + resnet50's layer4, NeRAF_resnet3d.py:128-131, 53 convolutions).  conv_w: HOST array of the Conv3d weights (device pointers) in state-dict order (conv1,
 * then per Bottleneck conv1, conv2, conv3, [downsample.0]); bn: HOST array of 4 device pointers per
 * BatchNorm3d in the same order {weight, bias, running_mean, running_var}.  use_batch_stats = 1 is
 * nn.Module.train() behaviour (statistics over the voxels of the single sample), 0 uses the running
 * statistics.
 * ---------------------------------------------------------------------------------- */
typedef struct neraf_resnet3d_desc {
  int grid_size;   /* 64, 128 or 256 */
  int in_channels; /* 7: rgb, alpha, xyz (NeRAF_model.py:185) */
  int n_features;  /* 1024 or 2048 */
} neraf_resnet3d_desc;

int neraf_resnet3d_num_convs(const neraf_resnet3d_desc* d); /* 43 (1024 features) | 53 (2048) */
size_t neraf_resnet3d_packed_bytes(const neraf_resnet3d_desc* d);
size_t neraf_resnet3d_workspace_bytes(const neraf_resnet3d_desc* d);
/* Algorithmic forward FLOPs of the encoder: sum over its 43 convolutions of 2 dout^3 taps cin cout with the real channel
 * and tap counts -- 94.72e9 for the 7 x 128^3 grid.  Host arithmetic only; < 0 for an unsupported descriptor. */
double neraf_resnet3d_forward_flops(const neraf_resnet3d_desc* d);
int neraf_resnet3d_pack_weights(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const float* const* conv_w, void* packed,
                                neraf_stream_t stream);
/* win_cells > 0: only the grid cells [win_start, win_start + win_cells) (flat index, the refresh window of NeRAF_model.py:395-404)
 * are converted into the workspace's fp16 channels-last input image; the CALLER vouches that every other cell of `grid` is unchanged
 * since the previous forward on this (grid, workspace, use_batch_stats).  win_cells == 0 converts the whole grid. */
int neraf_resnet3d_fwd(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const void* packed, const float* const* bn,
                       const float* grid, void* workspace, float* feat, int use_batch_stats, size_t win_start, int win_cells,
                       neraf_stream_t stream);
/* After a train-mode forward: running_mean/var <- (1-m) running + m batch (unbiased var), as nn.BatchNorm3d; with
 * num_batches_tracked != NULL (HOST array of the 43 BatchNorms' int64 device scalars) each counter is incremented by the same launch. */
int neraf_resnet3d_update_running_stats(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const void* workspace,
                                        float* const* bn, float momentum, int64_t* const* num_batches_tracked,
                                        neraf_stream_t stream);

/* Backward of neraf_resnet3d_fwd (train-mode BatchNorm).  `workspace` is the forward's (its activations and
 * BN statistics are read); packed_t comes from neraf_resnet3d_pack_weights_bwd.  dfeat fp32 [1024] ->
 * w_grads (HOST array, 43 device pointers shaped like the Conv3d weights, overwritten) and bn_grads (HOST
 * array, 2 per BatchNorm3d: d weight, d bias).  If n_cells > 0 the gradient w.r.t. channels [0, n_ch) of
 * the grid cells [cell_start, cell_start+n_cells) (flat x-major index, the refresh window of
 * NeRAF_model.py:306-311) is written to dgrid_cells fp32 [n_ch, n_cells].
 * The gradient chain is fp16 with fp32 accumulation (the reference's AMP, NeRAF_config.py:79) under per-tensor power-of-two scales
 * that live in bwd_workspace and follow the amax each tensor's producer recorded in the previous call (delayed scaling); the first
 * call on a workspace first repeats the chain as calibration passes until none of its tensors overflows.  neraf_resnet3d_bwd_reset forgets a workspace's calibration (call it
 * when the buffer is (re)allocated or the weights are replaced wholesale); neraf_resnet3d_bwd_chain_state is a test aid: exponent
 * and last recorded amax of the first n chain tensors, info2 = {tensors unsettled in the pass before the last, passes run} (HOST
 * arrays; synchronises the stream). */
size_t neraf_resnet3d_bwd_packed_bytes(const neraf_resnet3d_desc* d);
size_t neraf_resnet3d_bwd_workspace_bytes(const neraf_resnet3d_desc* d);
int neraf_resnet3d_pack_weights_bwd(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const float* const* conv_w, void* packed_t,
                                    neraf_stream_t stream);
int neraf_resnet3d_bwd(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const void* packed_t, const float* const* conv_w,
                       const float* const* bn, void* workspace, void* bwd_workspace, const float* dfeat,
                       float* const* w_grads, float* const* bn_grads, size_t cell_start, int n_cells, int n_ch,
                       float* dgrid_cells, neraf_stream_t stream);
int neraf_resnet3d_bwd_reset(neraf_ctx* ctx, void* bwd_workspace);
int neraf_resnet3d_bwd_chain_state(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const void* bwd_workspace, int32_t* e_out,
                                   float* amax_out, int n, int32_t* info2, neraf_stream_t stream);

/* Measurement aid (tools/resnet_node_roofline.py): while enabled, neraf_resnet3d_fwd / _bwd run their launches directly (no graph
 * replay) and append one record per launch -- "<kernel-name prefix> | description", algorithmic FLOPs, bytes read, bytes written as
 * designed (operands once, results once).  neraf_manifest_get returns the number of records (and fills record `index` if in range). */
int neraf_manifest_enable(neraf_ctx* ctx, int on);
int neraf_manifest_get(neraf_ctx* ctx, int index, char* name, int name_cap, double* flops, double* rbytes, double* wbytes);

/* Test aid: one conv + BatchNorm(train) + ReLU stage, forward and backward on caller data (allocates and
 * synchronises internally; not part of the product path). */
int neraf_debug_conv_bn_relu_stage(neraf_ctx* ctx, int cin, int cin_real, int cout, int k, int stride, int pad, int din,
                                   const void* x_f16, const float* w, const float* gamma, const float* beta, const float* g,
                                   void* y_f16, float* dx, float* dw, float* dgamma, float* dbeta, neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Camera-pose refinement on a ray bundle (nerfstudio CameraOptimizer(mode="SO3xR3").apply_to_raybundle, NeRAF_config.py:97):
 * pose_adjustment fp32 [n_cameras, 6] = (translation | so(3) log); per ray origins_out = origins + t[cam], dirs_out = exp(w[cam]) dirs
 * (Rodrigues, squared angle clamped at 1e-4 as nerfstudio does).  With reg_out3 != NULL the forward also writes nerfstudio's
 * CameraOptimizer.get_loss_dict / get_metrics_dict values [NS-recall] in the same launch: reg_out3 = {sum_c |t_c| * w_trans +
 * sum_c |w_c| * w_rot, |t|_F, |w|_F} (w_trans = trans_l2_penalty / n_cameras, w_rot = rot_l2_penalty / n_cameras).
 * The backward WRITES d pose [n_cameras, 6] = g_reg * d regulariser / d pose (g_reg: device scalar or NULL = 0) + the pull-back of
 * d origins_out / d dirs_out (rows g_stride floats apart: 3 = two [R,3] arrays, 6 = the halves of neraf_field_backward_rays' [R,6]
 * d_rays; both NULL = regulariser only).
 * ---------------------------------------------------------------------------------- */
/* Camera -> ray bundle (nerfstudio Cameras.generate_rays as Model.get_outputs_for_camera calls it, NeRAF_model.py:70-79 [NS-recall]):
 * c2w fp32 [n_cams,3,4] (OpenGL camera frame), fx / fy / cx / cy fp32 [n_cams] each, distortion fp32 [n_cams,6] =
 * (k1,k2,k3,k4,p1,p2) or NULL.  Ray i belongs to camera cam_idx[i] (int64, as nerfstudio's RayBundle.camera_indices) or, with
 * cam_idx NULL, to camera cam_single; its pixel is coords[i] = (row, col) fp32 or, with coords NULL, the centre of pixel i of a
 * `width`-wide image in row-major order (a whole frame: R = height * width).  Writes origins / dirs fp32 [R,3] (unit directions)
 * and, when cam_out != NULL, the int64 camera index per ray. */
int neraf_camera_rays(neraf_ctx* ctx, const float* c2w, const float* fx, const float* fy, const float* cx, const float* cy,
                      const float* distortion, int n_cams,
                      const int64_t* cam_idx, int cam_single, const float* coords, int R, int width, float* origins, float* dirs,
                      int64_t* cam_out, neraf_stream_t stream);
int neraf_camera_apply(neraf_ctx* ctx, const float* pose_adjustment, const int32_t* cam_idx, const float* origins, const float* dirs,
                       int R, float* origins_out, float* dirs_out, int n_cameras, float w_trans, float w_rot, float* reg_out3,
                       neraf_stream_t stream);
int neraf_camera_apply_bwd(neraf_ctx* ctx, const float* pose_adjustment, const int32_t* cam_idx, const float* dirs,
                           const float* d_origins, const float* d_dirs, int g_stride, int R, int n_cameras, float w_trans, float w_rot,
                           const float* g_reg, float* d_pose, neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Scalar plumbing of a training iteration as single launches (csrc/glue.hip): what NerfactoModel.get_loss_dict / get_metrics_dict and
 * Trainer.train_iteration build from scalar torch ops [NS-recall] -- loss = reduce(add, loss_dict.values()),
 * grad_scaler.scale(loss), psnr = -10 log10(mse).
 *   neraf_loss_sum_scale:       out2 = {scale * sum_i *terms[i], sum_i *terms[i]}; terms = HOST array of n <= 12 device scalars,
 *                               added left to right; scale = device scalar or NULL (1).
 *   neraf_vision_loss_finalize: out4 = {sums[0] k3[0], sums[1] k3[1], sums[2] k3[2], -10 log10(out4[0])} from the raw sums of
 *                               neraf_render_loss / neraf_interlevel_loss (rgb squared error, distortion, interlevel).
 *   neraf_vision_bwd_prologue:  d_rgb [n,3] = u_rgb * *g_rgb;  d_dens [n] = u_dens * *g_rgb + u_dist * *g_dist (unit gradients as
 *                               neraf_render_loss wrote them; g_* device scalars, NULL = 0); up3 = {*g_rgb, *g_inter, *g_dist} for
 *                               neraf_interlevel_loss' backward; zero-fills d_rays (n_ray_floats, may be NULL) and sums4 (may be NULL).
 * ---------------------------------------------------------------------------------- */
int neraf_loss_sum_scale(neraf_ctx* ctx, const float* const* terms, int n, const float* scale, float* out2, neraf_stream_t stream);
int neraf_vision_loss_finalize(neraf_ctx* ctx, const float* sums, const float* k3, float* out4, neraf_stream_t stream);
int neraf_vision_bwd_prologue(neraf_ctx* ctx, const float* u_rgb, const float* u_dens, const float* u_dist, const float* g_rgb,
                              const float* g_inter, const float* g_dist, size_t n_samples, float* d_rgb, float* d_dens, float* up3,
                              float* d_rays, size_t n_ray_floats, float* sums4, neraf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * fp16 working copies of the radiance parameters (tiny-cuda-nn keeps fp32 masters + fp16 copies inside its optimizer; the
 * reference reaches them through nerfstudio's NerfactoField / HashMLPDensityField).  neraf_cvt_f16_segments converts up to 12
 * contiguous fp32 tensors (hash tables, embedding, proposal MLP weights) to fp16 in ONE launch: src / dst / len are HOST arrays
 * of device pointers and element counts.  neraf_gather_f16 builds the MFMA weight fragments: dst[i] = fp16(flat[index[i]]) where
 * flat is the concatenation of nsrc (<= 8) fp32 tensors (src_len elements each; an index past the end reads 0) and index a device
 * int64 array (the fragment permutation built once by the host layer).
 * ---------------------------------------------------------------------------------- */
int neraf_cvt_f16_segments(neraf_ctx* ctx, const float* const* src, void* const* dst, const long long* len, int n,
                           neraf_stream_t stream);
int neraf_gather_f16(neraf_ctx* ctx, const float* const* src, const long long* src_len, int nsrc, const long long* index, void* dst,
                     long long n, neraf_stream_t stream);

/* Test aid: where a forward tensor lives inside the forward `workspace` (byte offset, logical rows x cols; rows are voxels in
 * x-major / z-fastest order, i.e. torch's flattened [D,H,W]).  kind 0 / 1 / 2: fp16 post-activation a1 / a2 / block output of
 * Bottleneck `index` (NeRAF_resnet3d.py:97, :101, :111); 3: fp16 pooled stem activation (:188); 4: uint8 arg-max tap table of
 * the stem max-pool (tap = (dz+1)*9 + (dy+1)*3 + (dx+1), 255 = window maximum not > 0; training forward only); 5: fp16 pre-BN
 * output of convolution `index`; 6: fp32 [2][cols] batch mean / biased variance of BatchNorm `index`. */
int neraf_resnet3d_debug_locate(const neraf_resnet3d_desc* d, int kind, int index, size_t* offset, int* rows, int* cols);

/* Test hook (host only, no GPU needed): n / d as the gather kernels compute it -- the division-by-invariant constants of csrc/field_common.h
 * (make_fastdiv) with the multiply-high evaluated on the host.  d == 0 returns 0xFFFFFFFF. */
uint32_t neraf_debug_fastdiv(uint32_t n, uint32_t d);

/* ------------------------------------------------------------------------------------
 * Optimizer step (SURVEY 8f "optimizer fusion"): torch.optim.Adam as nerfstudio's Optimizers apply it to the
 * `fields` and `audio_fields` groups (NeRAF_config.py:116-127), one launch per optimizer.  `table` is a device
 * array of 48-byte records {float* p; const float* g; float* m; float* v; int64 numel; int32 group; int32 slot}
 * and group_lr a HOST array of the n_groups (<= 8) learning rates (passed by value: schedulers change them every step);
 * g_ptrs (device uint64[n], may be NULL) overrides the records' gradient pointers -- autograd hands out new gradient
 * tensors every step, and this column can be refreshed with an asynchronous copy while the rest of the table stays;
 * workgroup b updates elements [blk_chunk[b]*chunk, +chunk) of tensor blk_tensor[b] (chunk =
 * neraf_fused_adam_chunk()).  step: device float[n_slots][4], one record {t, 1/(1-b1^t), 1/sqrt(1-b2^t), -} per counter slot; a
 * tensor's record names its slot (the host layer gives every parameter tensor its own, as torch.optim.Adam keeps a `step` per
 * parameter).  The call increments t of the slots of the n_tensors records of `table` (the tensors that hold gradients this step:
 * torch.optim.Adam skips parameters without a gradient, so the proposal networks' bias correction only advances on their update
 * steps, and the NAcF / ResNet3D members of "audio_fields" start at t = 1 when the audio branch starts although the radiance-field
 * members of the same group have been stepped since iteration 0, NeRAF_pipeline.py:186, :487);
 * grad_scale / found_inf: the GradScaler's device scalars (NULL = no scaling); when *found_inf != 0 nothing is modified.
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps), exactly torch's formula (no weight decay, no amsgrad).
 * ---------------------------------------------------------------------------------- */
int neraf_fused_adam_chunk(void);
/* GradScaler's non-finite check over the same tensor table: found_inf[0] = 1.0f if any gradient element is inf / nan, else 0.
 * g_ptrs_host / n_ptrs (here and in neraf_fused_adam_dual): the per-tensor gradient pointers as a HOST array of n_ptrs <= 256 entries;
 * they then travel in the kernel arguments and g_ptrs (the same column in DEVICE memory, which costs a host-to-device copy per step
 * because a step's gradient tensors are new allocations) is not read.  NULL / 0: g_ptrs, or the table's own g fields when that is NULL
 * too.  found_is_zero != 0: the caller vouches that found_inf[0] is 0 on entry (neraf_amp_update_scale with clear_flags left it so):
 * no launch to clear it. */
int neraf_grads_nonfinite(neraf_ctx* ctx, const void* table, const void* g_ptrs, const int* blk_tensor, const int* blk_chunk,
                          int n_blocks, float* found_inf, const void* const* g_ptrs_host, int n_ptrs, int found_is_zero,
                          neraf_stream_t stream);
/* GradScaler.update for up to 8 optimizers in one launch: torch._amp_update_scale_(scale, growth_tracker, sum of found_infs, ...)
 * -- any flag set: scale *= backoff_factor, tracker = 0; else tracker += 1 and, at growth_interval, scale *= growth_factor (kept
 * finite), tracker = 0.  found_infs: HOST array of n device flags (as neraf_grads_nonfinite writes them); clear_flags != 0: every
 * flag is reset to 0 after it has been read (the next step's checks then need no clearing launch). */
int neraf_amp_update_scale(neraf_ctx* ctx, float* scale, int32_t* growth_tracker, const float* const* found_infs, int n,
                           double growth_factor, double backoff_factor, int growth_interval, int clear_flags, neraf_stream_t stream);
int neraf_fused_adam(neraf_ctx* ctx, const void* table, const void* g_ptrs, const int* blk_tensor, const int* blk_chunk,
                     int n_blocks, const float* group_lr, int n_groups, int n_tensors, double beta1, double beta2, double eps,
                     float* step, const float* grad_scale, const float* found_inf, neraf_stream_t stream);

/* neraf_fused_adam with the update of doubly-stepped tensors fused: the reference steps the radiance-field parameters with the
 * "fields" optimizer and then again with "audio_fields" (NeRAF_pipeline.py:487) -- two passes over p and g.  Here the first optimizer's
 * call leaves those tensors out of its workgroup map (their counters still advance: every record of `table` does) and the second
 * optimizer's call passes `dual`, a device array parallel to its table of 24-byte records {float* m; float* v; int32 group; int32 slot}
 * (m == NULL: none) naming the FIRST optimizer's moments, parameter group (index into group_lr0, its HOST learning rates) and counter slot in `step0` (the first optimizer's step
 * table, already advanced by its own call) and its non-finite flag `found_inf0` (may be NULL).  For such a tensor the kernel applies
 * update 0 (unless *found_inf0) and then update 1 (unless *found_inf) in registers: the same fp32 operations in the same order as
 * the two launches, bit for bit.  n_blocks may be 0 (nothing but deferred tensors).  dual == NULL: plain neraf_fused_adam.
 * n_tensors == 0: the per-tensor counters are NOT advanced (they were, by an earlier call over this table: the host layer's flush of
 * an update that was deferred to an optimizer which then never stepped). */
int neraf_fused_adam_dual(neraf_ctx* ctx, const void* table, const void* g_ptrs, const int* blk_tensor, const int* blk_chunk,
                          int n_blocks, const float* group_lr, int n_groups, int n_tensors, double beta1, double beta2, double eps,
                          float* step, const float* grad_scale, const float* found_inf, const void* dual, const float* step0,
                          const float* found_inf0, const float* group_lr0, int n_groups0, const void* const* g_ptrs_host, int n_ptrs,
                          neraf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERAF_HIP_H */
