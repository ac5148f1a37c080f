/* frx.h -- C ABI of libfrx.so: the MI355X (gfx950) device layer under the
 * face-recognition training / verification hot path.
 *
 * The reference (Lac-quan-yeu-doi/Face-Recognition-Models) is 100 % Python and has no
 * FFI of its own: every arithmetic step on its hot path is an ATen call.  Each entry
 * point below names the reference call site (file:line under main_code/) whose ATen
 * work it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (a live torch tensor) unless
 *     the parameter is documented as host;
 *   - the library never allocates or frees tensor memory; scratch comes from a
 *     caller-supplied workspace sized by the matching *_workspace_bytes query;
 *   - `device` is the HIP ordinal, `stream` a hipStream_t; calls only enqueue work
 *     (no hidden synchronisation, graph-capturable);
 *   - return 0 on success, <0 = frx_status; text via frx_last_error() (thread-local);
 *   - re-entrant; safe from the autograd engine thread.
 */
#ifndef FRX_H
#define FRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* frx_stream_t; /* hipStream_t */

enum frx_status {
  FRX_OK = 0,
  FRX_ERR_ARG = -1,      /* bad shape / null pointer / unsupported combination */
  FRX_ERR_WORKSPACE = -2,/* workspace too small */
  FRX_ERR_HIP = -3       /* a HIP runtime call failed */
};

enum frx_dtype { FRX_F32 = 0, FRX_BF16 = 1 };

/* head kinds: utils/criterion.py ArcFace:232, CosFace:137, SphereFace:12, CurricularFace:491 */
enum frx_head_kind { FRX_ARC = 0, FRX_COS = 1, FRX_SPHERE = 2, FRX_CURR = 3 };

/* ---------------------------------------------------------------- diagnostics */
int frx_version(void);
const char* frx_last_error(void);
/* props[0]=CU count, [1]=clock kHz, [2]=LDS bytes/CU, [3]=wavefront size, [4]=gcnArch is gfx950 (0/1) */
int frx_device_props(int device, int64_t props[8]);

/* ---------------------------------------------------------------- margin head
 * Replaces, per training step: F.normalize x2 + F.linear / torch.mm + the ~18
 * element-wise [N,C] passes of the head forward (criterion.py:260-301, 162-197,
 * 57-107, 527-587), nn.CrossEntropyLoss (model_utils.py:556,179), accuracy()/topk
 * (metrics.py:3-16, model_utils.py:182) and the autograd replay of all of them.
 *
 * Weight layout follows the reference parameters: ARC/SPHERE `weight`
 * is [C,D] row-major, COS/CURR `kernel` is [D,C] row-major.  x is [N,D] fp32.
 */
typedef struct frx_head_desc {
  int32_t kind;      /* frx_head_kind */
  int32_t N, D, C;   /* D % 16 == 0 */
  float s, m;        /* scale / margin (config.py:16-37); SPHERE ignores s, m must be 2 */
  float momentum;    /* CURR EMA momentum (config.py:37) */
  float lamb;        /* SPHERE annealing lambda for THIS forward (criterion.py:58-60; host state) */
} frx_head_desc;

size_t frx_head_workspace_bytes(const frx_head_desc* d);

/* Forward.  The workspace keeps what backward needs (cosines, lse, inverse norms).
 *   labels   [N] int64
 *   state_t  [1] float, CURR's `t` buffer (criterion.py:517), updated in place before use (:570-573);
 *            may be NULL for other kinds
 *   ty_sum   optional [1] float: when non-NULL the batch mean of the target cosine used in the
 *            EMA is ty_sum[0]/ty_count instead of the local mean (data-parallel: the caller
 *            all-reduces it between frx_head_fwd_cos and frx_head_fwd_loss)
 *   cos_s, logits  optional [N,C] fp32 outputs (the reference's return contract)
 *   norms    optional [N]
 *   loss     [1] mean CE;  lse optional [N];  topk [2] int32 = (#top-1 hits, #top-5 hits)
 */
int frx_head_fwd_cos(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                     const float* w, const int64_t* labels, void* ws, size_t ws_bytes,
                     float* ty_sum_out /* optional [1]: sum over rows of clamp(cos[i,y_i]) */);
int frx_head_fwd_loss(int device, frx_stream_t stream, const frx_head_desc* d, const int64_t* labels,
                      float* state_t, const float* ty_sum, int64_t ty_count, void* ws, size_t ws_bytes,
                      float* cos_s, float* logits, float* norms, float* loss, float* lse,
                      int32_t* topk);
/* both phases back to back (single GPU) */
int frx_head_fwd(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                 const float* w, const int64_t* labels, float* state_t, void* ws, size_t ws_bytes,
                 float* cos_s, float* logits, float* norms, float* loss, float* lse, int32_t* topk);
/* Backward of mean-CE through margin, cosine GEMM and both normalisations.
 *   gout   optional [1] upstream dL/dloss (NULL = 1)
 *   dx [N,D]; dw same layout as w; when accumulate_dw != 0, dw += result */
int frx_head_bwd(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                 const float* w, const int64_t* labels, const float* state_t, const float* gout,
                 void* ws, size_t ws_bytes, float* dx, float* dw, int accumulate_dw);

/* ---------------------------------------------------------------- verification
 * Replaces F.normalize(feat1) * F.normalize(feat2) .sum(1) and the threshold count of
 * evaluate()/tune_threshold_roc() (model_utils.py:370-375, 391-393). */
int frx_pair_cosine(int device, frx_stream_t stream, const float* f1, const float* f2, int64_t P,
                    int32_t D, float* cos_out);
/* correct[0] += #{ (cos[i] > thr) == same[i] }  (strict >, model_utils.py:373-374) */
int frx_threshold_count(int device, frx_stream_t stream, const float* cos, const int64_t* same,
                        int64_t P, float thr, int32_t* correct);

#ifdef __cplusplus
}
#endif
#endif /* FRX_H */
