/* fr_hip.h -- C ABI of the MI355X-native embed-and-match hot path (libfr_hip.so).
 *
 * The reference (CEIA-NoLeakIV/face-recognition) has no FFI: its hot path is `model(tensor)` on a
 * torch.nn.Module (inference.py:120-122, evaluate.py:189-192), NumPy/torch cosine formulas
 * (utils/face_utils.py:84-87, evaluate.py:554) and no 1:N search at all.  This header is therefore
 * the interface a maintainer would bind (ctypes stub shown in INTEGRATION.md) to swap that path
 * for hand-written gfx950 kernels.  Each entry point names the reference lines it replaces.
 *
 * Conventions: plain C, no C++ types or exceptions across the boundary.  Every function returns
 * 0 on success or a negative fr_status; fr_last_error() returns a thread-local message.  Unless
 * stated otherwise pointers are DEVICE pointers owned by the caller; the library never allocates
 * or frees caller outputs; launches are asynchronous on the given hipStream_t (passed as void*,
 * NULL = default stream).  The only device memory the library owns is the packed weights of a
 * plan (allocated in fr_plan_create, freed in fr_plan_destroy).
 */
#ifndef FR_HIP_H
#define FR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fr_status {
    FR_OK = 0,
    FR_ERR_ARG = -1,          /* bad argument / inconsistent description */
    FR_ERR_UNSUPPORTED = -2,  /* layer shape with no kernel */
    FR_ERR_HIP = -3,          /* HIP runtime error (message has hipGetErrorString) */
    FR_ERR_WORKSPACE = -4     /* workspace too small */
} fr_status;

/* Activation storage / arithmetic mode of a plan.
 * FR_PREC_BF16: activations stored bf16, single bf16 MFMA pass, fp32 accumulate + fp32 epilogue.
 * FR_PREC_F32 : activations stored fp32; GEMM operands split hi+lo bf16, three MFMA passes
 *               (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo), fp32 accumulate (~16 mantissa bits).
 * FR_PREC_F24 : the arithmetic of FR_PREC_F32 with activations STORED as the top 24 bits of the fp32 pattern (16
 *               significant bits -- what the split operands carry anyway), rounded to nearest, in two planes per buffer: a
 *               bf16 plane (the truncated top 16 bits, directly the hi MFMA operand) and a byte plane (the next 8 bits).
 *               3 bytes per element instead of 4 on a byte-bound path.  Depthwise-separable families only (MobileNetV1 /
 *               V2 / V3): a plan with a dense 3x3 conv fails with FR_ERR_UNSUPPORTED. */
typedef enum fr_precision { FR_PREC_BF16 = 0, FR_PREC_F32 = 1, FR_PREC_F24 = 2 } fr_precision;
/* OR into `precision`: keep every layer a separate launch (no expand->depthwise fusion), so that
 * fr_embed_tap can observe every intermediate tensor. */
#define FR_PLAN_NO_FUSE 0x100

typedef enum fr_op { FR_OP_CONV = 0, FR_OP_SE = 1, FR_OP_LINEAR = 2 } fr_op;
typedef enum fr_act { FR_ACT_NONE = 0, FR_ACT_PRELU = 1, FR_ACT_HSWISH = 2,
                      FR_ACT_RELU = 3, FR_ACT_HSIGMOID = 4 } fr_act;

/* One reference layer group, described with HOST pointers to the fp32 tensors exactly as they sit
 * in the reference state_dict (no folding, no re-layout on the caller's side).
 *   FR_OP_CONV   Conv2d [+BatchNorm2d] [+PReLU|Hardswish]   (utils/layers.py:15-84, :156-173;
 *                models/mobilenetv2.py:45-46).  weight [cout, cin/groups, k, k].
 *   FR_OP_SE     SqueezeExcitation (utils/layers.py:87-122 with Hardsigmoid): cin = channels,
 *                cout = squeeze channels, weight/bias = fc1 [S,C]/[S], weight2/bias2 = fc2 [C,S]/[C].
 *                Output = scale * input; its consumer must be a 1x1 conv.
 *   FR_OP_LINEAR Linear [+BatchNorm1d] on the NCHW-flattened output of `src`
 *                (utils/layers.py:183-186; models/sphereface.py:84,126-127).  weight [out, in].
 * src / residual are indices into the descriptor array (-1 = network input / none); the residual
 * tensor is added after BN and activation (models/mobilenetv3.py:96-97, mobilenetv2.py:55,
 * sphereface.py:48). */
typedef struct fr_layer_desc {
    int32_t op;
    int32_t src;
    int32_t residual;
    int32_t cin, cout;
    int32_t kernel, stride, pad, groups;
    int32_t act;
    const float* weight;
    const float* bias;        /* NULL if the conv/linear has no bias */
    const float* bn_weight;   /* gamma; NULL if no BatchNorm follows */
    const float* bn_bias;
    const float* bn_mean;
    const float* bn_var;
    float bn_eps;
    const float* prelu;       /* [cout] slopes when act == FR_ACT_PRELU */
    const float* weight2;     /* SE only */
    const float* bias2;       /* SE only */
} fr_layer_desc;

typedef struct fr_plan fr_plan;

const char* fr_last_error(void);

/* Replaces: nn.Module construction + load_state_dict + .to(device).eval() (inference.py:55-96).
 * Folds BatchNorm (w' = w*g/sqrt(v+eps), b' = (b-m)*g/sqrt(v+eps)+beta, in double), packs weights
 * into MFMA fragment order / NHWC tap-major order, permutes Linear columns from NCHW-flatten to
 * NHWC order, uploads them to the CURRENT device.  Network input is fp32 NCHW [batch,in_c,in_h,in_w]. */
int fr_plan_create(const fr_layer_desc* layers, int n_layers, int in_c, int in_h, int in_w,
                   int precision, fr_plan** out_plan);
int fr_plan_destroy(fr_plan* plan);

/* Bytes of caller-provided device scratch needed by fr_embed for `batch` images. */
int fr_plan_workspace_bytes(const fr_plan* plan, int batch, size_t* out_bytes);
/* Output shape of layer `layer` (c,h,w); LINEAR / SE report (features,1,1). */
int fr_plan_layer_dims(const fr_plan* plan, int layer, int* c, int* h, int* w);
/* Number of kernel launches one fr_embed call issues per (sub-)batch (diagnostics).  Where a plan computes its 3x3 stem conv inside
 * the kernel of the first depthwise + 1x1 pair (32-channel stems), fr_embed_u8 and a tap on the stem take one launch more: they run
 * the stem kernel on its own. */
int fr_plan_num_launches(const fr_plan* plan);

/* Replaces: `model(tensor)` under torch.no_grad() (inference.py:121-122, evaluate.py:191-192).
 * x: fp32 NCHW [batch,in_c,in_h,in_w].  out: fp32; if the last layer is LINEAR it is
 * [batch,out_features], otherwise the last activation as NCHW fp32.  Raw, un-normalised, as the
 * reference backbones return (utils/layers.py:188-191).
 * Asynchronous on `stream`.  A batch of 128 images or more is cut into two sub-batches, the second of
 * which runs on an auxiliary stream of the device, shared by its plans (forked from and joined back to `stream` with
 * the plan's events), so one plan serves one host thread at a time, and plans used by several threads at once queue their
 * second halves behind each other on that stream; results do not depend on the split.  A workspace must not
 * be shared by calls that can overlap on the GPU: calls enqueued on different streams need different workspaces
 * (the host layer keeps one per plan, device and stream). */
int fr_embed(const fr_plan* plan, const float* x, int batch, float* out,
             void* workspace, size_t workspace_bytes, void* stream);
/* Replaces: `get_transform()(img)` + `model(tensor)` (inference.py:98-106 and :120-122) for images that
 * are already 112x112 RGB: x_nhwc is uint8 [batch,in_h,in_w,3] (what PIL / the aligner hands over, 37.6 KB
 * per face instead of 150 KB of fp32).  ToTensor (/255) and Normalize(0.5, 0.5) are applied inside the
 * stem kernel's load with the same fp32 arithmetic, so the result is bit-identical to fr_embed on the
 * transformed tensor.  Resize is NOT performed (the reference's Resize((112,112)) is the identity for such
 * input).  Needs a plan whose first layer is the 3x3 stem (all 7 backbones), a 4-byte aligned x_nhwc and
 * in_h*in_w*3 divisible by 4; FR_ERR_UNSUPPORTED otherwise.  SURVEY.md 8(f) rank 2. */
int fr_embed_u8(const fr_plan* plan, const unsigned char* x_nhwc, int batch, float* out,
                void* workspace, size_t workspace_bytes, void* stream);
/* Replaces: face_alignment (utils/face_utils.py:56-71) = cv2.warpAffine(image, M, (size, size),
 * borderValue=0.0) with OpenCV's defaults INTER_LINEAR / BORDER_CONSTANT, for a batch of faces that may come
 * from different source images (models/onnx_model.py:82 calls it per face).  All pointers are device
 * pointers: images[i] -> uint8 HWC 3-channel source image i (rows packed, width*3 bytes), sizes_hw[2i..2i+1] =
 * its height and width, inv_matrices[6i..6i+5] = the 2x3 matrix of face i ALREADY INVERTED (cv::warpAffine
 * inverts M in double before it starts; the host layer does the same: align.invert_affine).  out: uint8
 * [n,size,size,3], channel order as in the source.  Same fixed-point arithmetic as OpenCV 4.10 (10-bit
 * coordinates, 5 fractional bits, 15-bit weights): integer-exact against oracle/align.py.  The landmark ->
 * matrix step (estimate_norm, utils/face_utils.py:19-53) is host code: align.estimate_norm.
 * SURVEY.md 8(f) rank 4. */
int fr_align_faces(const unsigned char* const* images, const int* sizes_hw, const double* inv_matrices,
                   int n, int out_size, unsigned char* out, void* stream);
/* Replay form of fr_embed / fr_embed_u8 for small batches.  Replaces: the reference's per-image loops -- `model(tensor)` once
 * per image at inference.py:150-153 (extract_batch_embeddings), :109-123 (extract_features) and evaluate.py:517-555 (two bs-1
 * forwards per LFW pair) -- where a forward is ~60 launches of a few microseconds each and the GPU idles between them.
 * fr_embed_graph_create runs the plan once eagerly on a private stream (loads the code objects, sets the kernels' LDS limits),
 * then captures the same launch sequence into a hipGraph and instantiates it.  The graph is bound to the pointers given here:
 * x (float32 [batch,3,h,w] NCHW, or uint8 [batch,h,w,3] when x_is_u8), out and workspace must stay allocated and at the same
 * addresses until fr_embed_graph_destroy; the caller refills x and reads out around each launch.  The batch is never split
 * over auxiliary streams (single-stream capture); results are bit-identical to fr_embed on the same input.
 * The graph holds the PLAN's weight pointers: the plan must outlive the graph (destroy the graph first).
 * fr_embed_graph_launch enqueues one replay on `stream` (asynchronous; replays of one graph must not overlap); the graph's
 * device must be the current device (FR_ERR_ARG otherwise).
 * fr_embed_graph_nodes returns the number of KERNEL nodes captured (diagnostics), negative on error. */
typedef struct fr_graph fr_graph;
int fr_embed_graph_create(const fr_plan* plan, const void* x, int x_is_u8, int batch, float* out,
                          void* workspace, size_t workspace_bytes, fr_graph** out_graph);
int fr_embed_graph_launch(fr_graph* graph, void* stream);
int fr_embed_graph_nodes(const fr_graph* graph);
int fr_embed_graph_destroy(fr_graph* graph);
/* Test hook (no reference counterpart): after fr_debug_fail_after(L), L >= 0, the SECOND sub-batch of every split fr_embed /
 * fr_embed_u8 call (batch >= 128) stops with FR_ERR_HIP once L layers have been enqueued on its auxiliary stream, so that the
 * join-on-error path can be tested; fr_debug_fail_after(-1) switches it off (the initial state).  Process-wide. */
int fr_debug_fail_after(int layer);
/* Same, and additionally writes the output of layer `tap_layer` as fp32 NCHW to tap_out
 * (parity tests of intermediate blocks; no reference counterpart beyond forward hooks). */
int fr_embed_tap(const fr_plan* plan, const float* x, int batch, float* out,
                 void* workspace, size_t workspace_bytes, void* stream,
                 int tap_layer, float* tap_out);

/* Diagnostics (no reference counterpart): run fr_embed with a hipEvent pair around every layer on
 * `stream`, synchronise, and return the per-layer device time in milliseconds (layer_ms[n_layers],
 * HOST pointer).  fr_plan_layer_kernel names the kernel family that executes a layer. */
int fr_embed_profile(const fr_plan* plan, const float* x, int batch, float* out,
                     void* workspace, size_t workspace_bytes, void* stream, float* layer_ms);
int fr_plan_layer_kernel(const fr_plan* plan, int layer, const char** name);

/* L2-normalise rows: y = x / max(||x||, tiny); norms[n] optional (may be NULL).  The reference
 * normalises inside its similarity formulas only; this is the gallery/query preparation step. */
int fr_l2norm(const float* x, float* y, float* norms, int n, int d, void* stream);
/* fp32 rows -> L2-normalised bf16 rows (gallery / query format of fr_topk_search). */
int fr_l2norm_bf16(const float* x, uint16_t* y, int n, int d, void* stream);

/* Replaces: compute_similarity (utils/face_utils.py:84-87; eps = 0) and the LFW pair score
 * f1.dot(f2)/(f1.norm()*f2.norm()+1e-5) (evaluate.py:554; eps = 1e-5), plus the decision
 * `float64(sim) > threshold` (evaluate.py:205-206, inference.py:175).
 * a,b: fp32 [pairs,d]; out_sim fp32 [pairs]; out_dec uint8 [pairs] or NULL. */
int fr_cosine_pairs(const float* a, const float* b, int pairs, int d, float eps, double threshold,
                    float* out_sim, uint8_t* out_dec, void* stream);

/* Synthetic gallery / query rows, bit-identical to
 * oracle/search.py.  Writes rows [row0,row0+nrows) of the global set as L2-normalised bf16.
 * plant_seed/n_plant/n_gallery_total: planted matches (n_plant = 0 for queries). */
int fr_synth_rows(uint16_t* out, uint64_t seed, int64_t row0, int64_t nrows, int dim,
                  uint64_t plant_seed, int64_t n_plant, int64_t n_gallery_total, void* stream);

/* 1:N cosine search (no reference counterpart; definitional oracle in oracle/search.py).
 * q bf16 [Q,d] and g bf16 [G,d] are L2-normalised rows; scores = q @ g^T on MFMA, exact top-k
 * per query (k <= 8), ties to the lower index.  out_scores fp32 [Q,k] descending, out_idx int64
 * [Q,k] = local row + idx_offset.  workspace: fr_topk_workspace_bytes(Q,G,k) bytes. */
int fr_topk_workspace_bytes(int64_t Q, int64_t G, int k, size_t* out_bytes);
int fr_topk_search(const uint16_t* q, const uint16_t* g, int64_t Q, int64_t G, int d, int k,
                   int64_t idx_offset, float* out_scores, int64_t* out_idx,
                   void* workspace, size_t workspace_bytes, void* stream);
/* No counterpart in the reference.  For callers that cannot allocate device memory around a search (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): fr_topk_search without the workspace
 * argument.  The partial top-k lists live in a 32 MiB file-scope device array of the library (one copy per GPU,
 * held until the process ends); the call itself makes no hipMalloc / hipFree.  Results are identical to
 * fr_topk_search.  FR_ERR_WORKSPACE when fr_topk_workspace_bytes(Q,G,k) exceeds 32 MiB (with the default tuning
 * the largest need is 25.1 MB at k = 8, 15.7 MB at k = 5, for any Q and G; only the FR_SEARCH_BLOCKS /
 * FR_SEARCH_QCHUNK tuning variables can push it beyond).  Q == 0 performs the one-time per-device set-up only
 * (address look-up of the array, kernel attributes) -- do that once before capturing a graph.  All calls on one GPU
 * share that GPU's array: calls that can overlap on one GPU (two streams, two host threads) race; serialise them. */
int fr_topk_search_noalloc(const uint16_t* q, const uint16_t* g, int64_t Q, int64_t G, int d, int k,
                           int64_t idx_offset, float* out_scores, int64_t* out_idx, void* stream);
/* Merge P per-shard results (scores [P,Q,k], idx [P,Q,k], e.g. after an RCCL all-gather) into the
 * global top-k [Q,k]; deterministic: descending score, ties to the lower global index. */
int fr_topk_merge(const float* scores, const int64_t* idx, int P, int64_t Q, int k,
                  float* out_scores, int64_t* out_idx, void* stream);

/* Same merge, reading the receive buffer of ONE all-gather as it is (no reference counterpart):
 * packed = P messages back to back, each [Q*k fp32 scores, padded to a multiple of 8 bytes | Q*k int64 global
 * indices] -- the layout fr_topk_search fills when out_scores / out_idx point into such a message
 * (face-recognition_amd/search.py: packed_bytes, exchange_topk_chunks). */
int fr_topk_merge_packed(const void* packed, int P, int64_t Q, int k,
                         float* out_scores, int64_t* out_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FR_HIP_H */
