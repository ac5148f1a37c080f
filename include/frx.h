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

/* head kinds (main_code/utils/criterion.py): ArcFace:232, CosFace:137, SphereFace:12, CurricularFace:491, and the
 * SURVEY 8(f)-3 widening: MV_Softmax:327 ('am' / 'arc' margin types), AdaFace:795, ElasticArcFace:1054,
 * ElasticCosFace:951, MagFace:1178, VPLArcFace:619 */
enum frx_head_kind { FRX_ARC = 0, FRX_COS = 1, FRX_SPHERE = 2, FRX_CURR = 3,
                     FRX_MV_AM = 4, FRX_MV_ARC = 5, FRX_ADA = 6, FRX_ELASTIC_ARC = 7, FRX_ELASTIC_COS = 8, FRX_MAG = 9,
                     FRX_VPL = 10 };

/* ---------------------------------------------------------------- diagnostics */
int frx_version(void);
/* sizeof(frx_head_desc), sizeof(frx_conv_desc), sizeof(frx_dgrad_fuse), sizeof(frx_wgrad_job), sizeof(frx_bn_tot), 0, 0, 0:
 * lets a binding verify its own struct layouts (ctypes / cgo / JNI) against the library it loaded */
int frx_struct_sizes(int64_t sizes[8]);
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
  float s, m;        /* scale / margin (config.py:16-70); SPHERE ignores s, m is an integer 1..5; ELASTIC / MAG ignore m */
  float momentum;    /* CURR EMA momentum (config.py:37) */
  float lamb;        /* SPHERE: annealing lambda for THIS forward (criterion.py:58-60; host state)
                        MAG:    frx_head_bwd: lambda_g, the weight of loss_g in the total loss (model_utils.py:180);
                                frx_head_bwd_dlogits: the upstream gradient dL/dloss_g (0 leaves loss_g out) */
  float p[4];        /* MV_*: p[0] = mv_weight (criterion.py:341)
                        ADA:  p[0] = h, p[1] = t_alpha (criterion.py:805-807)
                        MAG:  p[0] = l_margin, p[1] = u_margin, p[2] = l_a, p[3] = u_a (criterion.py:1188-1191)
                        VPL:  p[0] = lamda, p[1] = delta (criterion.py:632-633) */
  int32_t flags;     /* ARC, MAG, VPL: bit 0 = easy_margin (criterion.py:284-285, 1187, 631)
                        VPL: bit 1 = norm_training_flag, the memory is in use (criterion.py:671-679)
                        SPHERE: bit 2 = read the annealing lambda from state_t[0] instead of `lamb` (a captured
                                hipGraph then follows criterion.py:58-60 without re-capture)
                        bit 4 = reserved (rounds 3-4: a split-bf16 form of the head's GEMMs, measured no faster, removed)
                        bit 3 = class-sharded mode (ARC / COS / SPHERE / CURR / MV_*): this descriptor is the column
                                shard [class_offset, class_offset + C) of a wider head; N counts the rows of the
                                GATHERED batch and labels stay global (see "class-sharded head" below) */
  int32_t class_offset; /* flags bit 3: global index of the shard's first class; otherwise ignored */
} frx_head_desc;

/* Per-kind meaning of the `state_t` argument of the calls below (device floats owned by the caller):
 *   CURR          [1]  the `t` buffer (criterion.py:517), updated in place before use (:570-573)
 *   ADA           [2]  batch_mean, batch_std (criterion.py:838-839), EMA-updated in place before use (:873-877)
 *   ELASTIC_*     [N]  this step's per-row margins, already sampled and clamped by the caller
 *                      (torch.normal + clamp, criterion.py:1002-1004 / 1113-1115; read only)
 *   VPL           [C*D + C]  `mem` [C,D] then `life` [C] (criterion.py:660-661), both updated by frx_head_vpl_prepare
 *   SPHERE        [1]  only with flags bit 2: this forward's annealing lambda (read only)
 *   other kinds   may be NULL */

size_t frx_head_workspace_bytes(const frx_head_desc* d);

/* Forward.  The workspace keeps what backward needs (cosines, lse, inverse norms).
 *   labels   [N] int64
 *   state_t  per-kind head state, see the table above (CURR's `t`, ADA's batch statistics, ELASTIC's margins)
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
/* After frx_head_fwd_cos: ty_out [N] <- the clamped target cosine of every row.  The elastic heads' plus=True variant
 * (criterion.py:1006-1011, 1117-1122) assigns its sampled margins by the rank of these before frx_head_fwd_loss. */
int frx_head_target_cos(int device, frx_stream_t stream, const frx_head_desc* d, void* ws, size_t ws_bytes, float* ty_out);
/* VPL only, between frx_head_fwd_cos and frx_head_fwd_loss (frx_head_fwd calls it itself): per-class mean of the batch's
 * raw features into `mem`, `life` = delta for those classes, life -= 1 for all classes, cosine against the normalised
 * memory and the lamda blend of criterion.py:699-722 written over the cosines in the workspace (flags bit 1 clear: no-op). */
int frx_head_vpl_prepare(int device, frx_stream_t stream, const frx_head_desc* d, const float* x, const int64_t* labels,
                         float* state_t, void* ws, size_t ws_bytes);
/* ---- class-sharded head (the reference's dormant device_id chunking, criterion.py:268-278, is the same
 * partition without the exchanges).  Each rank owns C/world class columns (no head gradient on the wire); the batch is
 * all-gathered, so N = world x per-rank batch rows.  Sequence per step, collectives issued by the caller in between:
 *   frx_head_shard_cos    -> ty_out [N]: clamped target cosine of the rows whose label falls in this shard, 0 elsewhere
 *     [all-reduce SUM of ty_out over ranks -> every rank holds every row's target cosine]
 *   frx_head_shard_rows   (state update from the global target cosines, margin + row sweep over the local columns)
 *                         -> part [3][N]: row max, sum exp(z - row max), count of local columns ranked above the target
 *     [all-reduce MAX of a copy of part[0] -> global_max]
 *   frx_head_shard_rescale  part[1][n] *= exp(part[0][n] - global_max[n])
 *     [all-reduce SUM of part[1..2] -> global sum-exp and global rank]
 *   frx_head_shard_finish -> loss, lse, top-k (identical on every rank); the global lse stays in the workspace
 *   frx_head_bwd          -> dw of the LOCAL columns (complete) and this shard's PARTIAL dx [N, D]
 *     [reduce-scatter SUM of dx -> each rank's own rows]; pass gout = world when the optimiser rescales by 1/world */
int frx_head_shard_cos(int device, frx_stream_t stream, const frx_head_desc* d, const float* x, const float* w,
                       const int64_t* labels, void* ws, size_t ws_bytes, float* ty_out);
int frx_head_shard_rows(int device, frx_stream_t stream, const frx_head_desc* d, const int64_t* labels, float* state_t,
                        const float* ty_global, void* ws, size_t ws_bytes, float* part);
int frx_head_shard_rescale(int device, frx_stream_t stream, int N, const float* local_max, const float* global_max,
                           float* part_sum);
int frx_head_shard_finish(int device, frx_stream_t stream, const frx_head_desc* d, const float* state_t,
                          const float* global_max, const float* global_sum, const float* global_rank, void* ws,
                          size_t ws_bytes, float* norms, float* loss, float* lse, int32_t* topk);
/* After frx_head_fwd_loss: loss_g [1] (MAG: mean(x_norm / u_a^2 + 1 / x_norm), criterion.py:1235-1239; 0 for other
 * kinds) and, optionally, the per-row parameter the epilogue used, row_param [N]: ADA margin_scaler (:879-880), MAG
 * ada_margin (:1229-1233), ELASTIC_* the margins passed in; zeros otherwise.  For MAG the `norms` output of
 * frx_head_fwd_loss is the clamped x_norm the reference returns (:1246,1283). */
int frx_head_aux(int device, frx_stream_t stream, const frx_head_desc* d, const float* state_t, void* ws,
                 size_t ws_bytes, float* loss_g, float* row_param);
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

/* Same, for an ARBITRARY upstream gradient dlogits [N,C] = dL/dlogits (what autograd hands back when
 * a caller applies its own criterion to the returned logits, model_utils.py:178-185). */
int frx_head_bwd_dlogits(int device, frx_stream_t stream, const frx_head_desc* d, const float* x,
                         const float* w, const int64_t* labels, const float* state_t, const float* dlogits,
                         void* ws, size_t ws_bytes, float* dx, float* dw, int accumulate_dw);

/* ---------------------------------------------------------------- verification
 * Replaces F.normalize(feat1) * F.normalize(feat2) .sum(1) and the threshold count of
 * evaluate()/tune_threshold_roc() (model_utils.py:370-375, 391-393). */
int frx_pair_cosine(int device, frx_stream_t stream, const float* f1, const float* f2, int64_t P,
                    int32_t D, float* cos_out);
/* correct[0] += #{ (cos[i] > thr) == same[i] }  (strict >, model_utils.py:373-374) */
int frx_threshold_count(int device, frx_stream_t stream, const float* cos, const int64_t* same,
                        int64_t P, float thr, int32_t* correct);


/* ---------------------------------------------------------------- backbone: convolutions
 * Replace every nn.Conv2d / nn.Linear of the torchvision ResNet-50 the reference builds in
 * utils/backbones.py:16-18 (forward: criterion.py:320; backward: model_utils.py:185).
 * Activations are NHWC, weights K-contiguous: KRSC [Co][R][S][Ci] for forward / wgrad and CRSK
 * [Ci][R][S][Co] for dgrad (frx_weight_prep makes both from the fp32 master copy).
 * The previous layer's train-mode BatchNorm + ReLU is applied to the input while tiles are
 * staged (in_scale / in_shift / in_relu), and the per-channel sum / sum-of-squares of the
 * output needed by the NEXT BatchNorm come out of the epilogue (stat_partial). */
typedef struct frx_conv_desc {
  int32_t dtype;                 /* frx_dtype of activations and kernel-format weights */
  int32_t N, Hi, Wi, Ci;         /* input  [N,Hi,Wi,Ci] */
  int32_t Co, R, S, stride, pad; /* stride 1 or 2 */
  int32_t Ho, Wo;                /* output [N,Ho,Wo,Co] */
  int32_t stem;                  /* 1: 7x7 s2 conv on the zero-bordered NHWC4 image of frx_input_prep;
                                    weights [Co][7][8][4] (tap 8 and channel 4 are zero) */
} frx_conv_desc;

/* rows of the stat_partial buffer ([rows][2][Co] floats) frx_conv_fwd writes for this layer */
int frx_conv_stat_rows(const frx_conv_desc* d);
/* diagnostic (host logic only): the block tile (pixels x channels) frx_conv_fwd (dgrad = 0) or frx_conv_dgrad* (dgrad = 1)
 * launches for this layer -- what the profiling labels and the stat-row counts are derived from */
int frx_conv_tile(const frx_conv_desc* d, int dgrad, int* bm, int* bn);
/* diagnostic (host logic only): the kernel's row tile (128 or 64 pixels) if this layer's geometry (bf16, 3x3, stride 1, pad 1, images up to 30 pixels wide, a
 * multiple of 64 gathered channels) puts frx_conv_fwd* (dgrad = 0) / frx_conv_dgrad_bn* (dgrad = 1) on the PATCH-MODE kernel
 * (csrc/conv_kernels.h "P3") whenever the call carries a prologue and
 * no addend -- and, if it asks for per-tile partial statistics, frx_conv_tile's row tile is the same.  0 otherwise; -1
 * for a rejected descriptor.  FRX_CONV3X3=0 in the environment switches the kernel off. */
int frx_conv_patch_mode(const frx_conv_desc* d, int dgrad);
/* diagnostic (host logic only): the implicit-GEMM kernel instantiation the calling thread's last frx_conv_fwd* /
 * frx_conv_dgrad* call launched: fields12 = {BM, BN, waves, K-chunk bytes, LDS-DMA stages (0 = register ring), MODE (0 fwd,
 * 1 dgrad, 2 stem, 3 / 4 patch-mode 3x3 forward / input gradient), PRO (0 none, 1 BN+ReLU, 2 BN backward), EPI (0 plain,
 * 1 statistics, 2 / 4 masked BN-backward statistics, 3 fc), addend, persistent, staging waves, contraction length} --
 * the key of bench.py's per-class roofline table and of scripts/traffic_summary.py's PMC rows */
int frx_last_conv_launch(int* fields12);
int frx_stem_padded_dims(int Hi, int Wi, int* Hp, int* Wp);
/* Train-mode BatchNorm statistics as REPLICATED TOTALS (csrc/bn_tot.h; replaces the 53 + 53 per-layer finalize launches of
 * a training step, i.e. the statistics half of every nn.BatchNorm2d forward / backward of backbones.py:16-18).  A producing
 * kernel ADDS its per-channel sums with float atomics into `replicas` rows  totals[replicas][2][C]  (forward: sum y, sum y^2;
 * backward: sum dz, sum dz*xhat); every consuming kernel derives the constants it needs from those rows while it sets up;
 * frx_bn_finalize_batched / frx_bn_bwd_finalize_batched write the canonical per-channel arrays once per pass and zero the
 * rows.  Sums depend on arrival order in the last bits: the bit-reproducible path is stat_partial + frx_bn_finalize. */
typedef struct frx_bn_tot {
  const float* totals;   /* [replicas][2][C] */
  const float* gamma;    /* [C] */
  const float* beta;     /* [C]  forward consumers (scale / shift) */
  const float* mean;     /* [C]  backward consumers (alpha / beta / gam): the arrays frx_bn_finalize_batched wrote */
  const float* invstd;   /* [C]  backward consumers */
  int32_t replicas;      /* a power of two: few rows for layers with few row tiles (every consumer block reads all of them) */
  float count;           /* elements per channel */
  float eps;             /* forward */
  int32_t reserved;
} frx_bn_tot;
/* A 1x1 / stride-1 convolution whose input is the residual merge of the block BEFORE it, evaluated as the prologue:
 *   x = relu(s3 * y3 + b3 + (sd * idn + bd))      (sd = 1, bd = 0 when sd == NULL: a plain identity)
 * i.e. torchvision Bottleneck.forward's `out = bn3(conv3(..)); out += identity; out = relu(out)` (backbones.py:16-18)
 * feeding the next Bottleneck's conv1, without the pass over memory frx_block_merge_fwd* spends on it.  y3 / idn [M, Ci]
 * are the previous block's raw conv3 output and its identity (the block input, or the raw output of its projection);
 * the BatchNorm constants come as arrays (s3, b3[, sd, bd]) or as replicated totals (bn3[, bnd]) -- one form for both.
 * block_out [M, Ci] receives x (what the merge pass would have written, bit for bit) and mask [M * Ci / V] its > 0 bits
 * (may be NULL), both stored once by the first column of tiles; y and the statistics are frx_conv_fwd's / _tot's. */
int frx_conv_fwd_merge(int device, frx_stream_t stream, const frx_conv_desc* d, const void* y3, const void* idn,
                       const void* w, const float* s3, const float* b3, const float* sd, const float* bd,
                       const frx_bn_tot* bn3, const frx_bn_tot* bnd, void* block_out, uint8_t* mask, void* y,
                       float* stat_partial, float* stat_totals, int stat_replicas);
/* frx_conv_fwd with the prologue constants taken from `in_bn` (NULL: no prologue) and the statistics of y ADDED into
 * stat_totals [stat_replicas][2][Co] (NULL: none) */
int frx_conv_fwd_tot(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x, const void* w_krsc,
                     const frx_bn_tot* in_bn, int in_relu, void* y, float* stat_totals, int stat_replicas);
/* frx_conv_fwd / frx_conv_fwd_tot (prologue constants as arrays OR as totals, statistics as partial rows OR as totals, or
 * none) that also KEEPS the prologue's output  x_norm_out = relu(scale * x + shift)  (same shape and dtype as x) -- what the
 * 3x3 weight gradient otherwise re-evaluates once per staged tap (torch autograd saves this tensor: the input of nn.Conv2d
 * conv2 in torchvision's Bottleneck).  Patch-mode layers only (frx_conv_patch_mode(d, 0) != 0, and with stat_partial its row
 * tile = frx_conv_tile's): there every element is transformed exactly once, by the block that owns its pixel row. */
int frx_conv_fwd_keep(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x, const void* w_krsc,
                      const float* in_scale, const float* in_shift, const frx_bn_tot* in_bn, int in_relu, void* y,
                      float* stat_partial, float* stat_totals, int stat_replicas, void* x_norm_out);
/* y = conv(f(x), w) [+ bias]; out_f32 stores y as fp32 (the fc layer feeding the head) */
int frx_conv_fwd(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x, const void* w_krsc,
                 const float* in_scale, const float* in_shift, int in_relu, const float* bias, void* y,
                 int out_f32, float* stat_partial);
/* dx = conv_transpose(dy, w) [+ addend]   (addend: the residual branch's gradient) */
int frx_conv_dgrad(int device, frx_stream_t stream, const frx_conv_desc* d, const void* dy, const void* w_crsk,
                   const void* addend, void* dx);
/* BatchNorm backward fused into the convolution backward (saves two HBM passes per layer):
 *   prologue: the kernel reads  dy = alpha*dz + beta*pro_y + gam  on the fly (frx_bn_bwd_finalize's coef);
 *   epilogue: the freshly computed input gradient g is masked (ReLU of the producing layer: out > 0, or
 *             epi_scale*epi_y + epi_shift > 0), stored as dz, and  sum(dz), sum(dz*xhat)  per channel go to
 *             epi_partial [frx_conv_dgrad_stat_rows][2][Ci]  (then frx_bn_bwd_finalize as usual). */
typedef struct frx_dgrad_fuse {
  const void* pro_y;         /* raw output y of THIS conv (dtype T, [N,Ho,Wo,Co]); NULL: `dz` argument already is dy */
  const float* pro_coef;     /* [3][Co] */
  const void* epi_y;         /* raw output of the conv that produced this conv's INPUT ([N,Hi,Wi,Ci]); NULL: no epilogue */
  const void* epi_out;       /* optional block output for the merge-ReLU mask */
  const float* epi_scale;    /* mask = epi_scale*epi_y + epi_shift > 0 when epi_out is NULL */
  const float* epi_shift;
  const float* epi_mean;
  const float* epi_invstd;
  float* epi_partial;
  const void* epi_out_bits;  /* optional, instead of epi_out: the merge-ReLU mask as written by frx_block_merge_fwd_mask
                                (one byte per 16-byte channel group): the epilogue then reads 1/16 of the bytes */
  int32_t addend_stride;     /* 0 / 1: `addend` has dx's shape.  2: it is the COMPACT [N,ceil(Hi/2),ceil(Wi/2),Ci] input
                                gradient of a stride-2 1x1 branch (computed as a stride-1 conv on the coarse grid) and is
                                added at the even pixels: the zeros of the other three quarters are never materialised */
  void* pro_dy_out;          /* optional (1x1 convs, with pro_y): the prologue's dy = alpha*dz + beta*y + gam is also stored
                                here ([N,Ho,Wo,Co], dtype T) by the first column of tiles, so that frx_conv_wgrad can read
                                dy without re-evaluating the BN backward (and without reading two tensors) */
  const frx_bn_tot* pro_tot; /* instead of pro_coef (with pro_y): alpha / beta / gam derived from replicated totals */
  float* epi_totals;         /* instead of epi_partial: (sum dz, sum dz*xhat) ADDED into [epi_replicas][2][Ci] */
  int32_t epi_replicas;
} frx_dgrad_fuse;
int frx_conv_dgrad_stat_rows(const frx_conv_desc* d);
int frx_conv_dgrad_bn(int device, frx_stream_t stream, const frx_conv_desc* d, const void* dz, const void* w_crsk,
                      const void* addend, void* dx, const frx_dgrad_fuse* fuse);
int frx_conv_wgrad_bn(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x, const float* in_scale,
                      const float* in_shift, int in_relu, const void* dz, const void* pro_y, const float* pro_coef,
                      float* dw);
/* dw (fp32 KRSC, accumulated) += f(x)^T dy */
int frx_conv_wgrad(int device, frx_stream_t stream, const frx_conv_desc* d, const void* x, const float* in_scale,
                   const float* in_shift, int in_relu, const void* dy, float* dw);

/* All weight gradients of (a part of) the backward pass in ONE launch of persistent blocks (the per-layer calls
 * above each pay a launch floor, a ring fill and an atomics tail, and half of them cannot fill the GPU).
 * Pointers are captured when the table is planned: plan once, run every step.  dw is accumulated like frx_conv_wgrad. */
typedef struct frx_wgrad_job {
  frx_conv_desc d;
  const void* x;             /* conv input activation (raw when in_scale is given) */
  const float* in_scale;     /* optional BN+ReLU prologue on x, as in frx_conv_wgrad */
  const float* in_shift;
  int32_t in_relu;
  const void* dy;            /* dy -- or dz when pro_y / pro_coef are given (frx_conv_wgrad_bn) */
  const void* pro_y;
  const float* pro_coef;
  float* dw;
  /* DECOMPOSED job (both set, pro_y / pro_coef NULL; 1x1, stride 1, Ci <= 128): the weight gradient of a conv whose dy is the
   * BatchNorm backward alpha*dz + beta*y + gam of (dz, y) WITHOUT reading y: dw accumulates the plain dz^T x, gram [Ci*Ci]
   * the matrix x^T x and xsum [Ci] the column sums of x (x after its prologue); frx_wgrad_gram_finish closes
   * dW = alpha (.) dw + beta (.) (W gram) + gam (x) xsum, because y = x W^T is linear in x.  gram must be followed by one
   * zeroed int32 (the finish launch's completion counter) and, like xsum, start at zero. */
  float* gram;
  float* xsum;
} frx_wgrad_job;
int64_t frx_wgrad_group_bytes(const frx_wgrad_job* jobs, int njobs);          /* size of the device table; < 0: error */
/* Enqueue-only like every other call: the table is built in `table_host` (table_bytes of PINNED host memory owned by the
 * caller, to be kept alive until the copy has run) and copied to `table_dev` by ONE asynchronous copy on `stream`.
 * *small_tiles <- 1 when every job takes the 64 x 64 tile (Co <= 64 or Ci <= 64: layer1, the stem): hand it to
 * frx_wgrad_group_run, which then launches the instantiation that fits four persistent blocks per CU instead of two --
 * so group such layers into a list of their own.  The table also holds the launch's eight item-draw counters (one per XCD;
 * every launch leaves them at zero): one table serves one launch at a time. */
int frx_wgrad_group_plan(int device, frx_stream_t stream, const frx_wgrad_job* jobs, int njobs, void* table_host,
                         void* table_dev, int64_t table_bytes, int* nitems, int* small_tiles, int* nlayers);
/* nlayers: as returned by the plan (njobs + the decomposed jobs' internal x^T x layers) */
int frx_wgrad_group_run(int device, frx_stream_t stream, int dtype, void* table_dev, int nlayers, int nitems,
                        int small_tiles);
/* closes the decomposed jobs of a list after frx_wgrad_group_run (and after the BatchNorm backward coefficients are final):
 * table_dev [n][8] int64 = {dw, kernel-format weight [Co][Ci] in the compute dtype (what the forward multiplied by), gram,
 * xsum, coef [3][Co] (alpha | beta | gam: frx_bn_bwd_finalize*), Co, Ci, index of the layer's first block (256 weights per
 * block)}; total_blocks = sum of ceil(Co*Ci / 256).  gram and xsum are zeroed again when the layer is done. */
int frx_wgrad_gram_finish(int device, frx_stream_t stream, int dtype, int n, const int64_t* table_dev, int total_blocks);

/* ---------------------------------------------------------------- backbone: BatchNorm / ReLU / residual / pools
 * Replace nn.BatchNorm2d x53 (train: batch statistics + running-stat update, momentum 0.1, eps 1e-5;
 * eval: running statistics), nn.ReLU, the residual add, MaxPool2d(3,2,1), AdaptiveAvgPool2d(1). */
int frx_bn_finalize(int device, frx_stream_t stream, const float* partial, int rows, int C, int64_t count,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean, float* invstd, float* scale, float* shift);
int frx_bn_eval_affine(int device, frx_stream_t stream, int C, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, float eps, float* scale, float* shift);
/* out = relu(s3*y3 + b3 + idn),  idn = block input or sd*yd + bd (downsample branch) */
int frx_block_merge_fwd(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* y3,
                        const float* s3, const float* b3, const void* idn, const float* sd, const float* bd,
                        void* out);
/* the same, also writing mask [rows * C / V] bytes, V = channels per 16-byte group (8 bf16 / 4 fp32): bit j of a byte =
 * (channel j of that group > 0) -- the backward's ReLU mask at 1/16 of the block output's size */
int frx_block_merge_fwd_mask(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* y3,
                             const float* s3, const float* b3, const void* idn, const float* sd, const float* bd,
                             void* out, uint8_t* mask);
/* the same with the BatchNorm constants derived from replicated totals (bnd NULL: identity block); mask may be NULL */
int frx_block_merge_fwd_tot(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* y3,
                            const frx_bn_tot* bn3, const void* idn, const frx_bn_tot* bnd, void* out, uint8_t* mask);
/* every listed BatchNorm layer in ONE launch (one thread per channel): totals -> mean / invstd / scale / shift (+ running
 * statistics), then the rows are zeroed.  table_dev [n][16] int64 = {totals, replicas, C, count, gamma, beta,
 * running_mean | 0, running_var | 0, mean, invstd, scale, shift, index of the layer's first block (256 channels per
 * block), eps as float bits, momentum as float bits, pointer to the layer's int64 num_batches_tracked | 0 (incremented by
 * one: nn.BatchNorm2d's counter, no launch of its own)}; total_blocks = sum of ceil(C / 256). */
int frx_bn_finalize_batched(int device, frx_stream_t stream, int n, const int64_t* table_dev, int total_blocks);
/* backward twin: totals (sum dz, sum dz*xhat) -> dgamma +=, dbeta +=, coef [3][C]; rows zeroed.  table_dev [n][16] int64 =
 * {totals, replicas, C, count, gamma, mean, invstd, dgamma | 0, dbeta | 0, coef, 0, 0, first block, 0, 0, 0} */
int frx_bn_bwd_finalize_batched(int device, frx_stream_t stream, int n, const int64_t* table_dev, int total_blocks);
/* BatchNorm backward in three steps.  dz = g*mask with mask = (out>0) if out given, else
 * (scale*y+shift>0) if relu, else 1.
 *   reduce   -> partial [frx_bn_bwd_partial_rows][2][C] = (sum dz, sum dz*xhat); optional dz_out
 *   finalize -> dgamma +=, dbeta +=, coef [3][C] = (alpha, beta, gam): the BN backward is AFFINE in (dz, y),
 *               dy = alpha*dz + beta*y + gam  with alpha = gamma*invstd, beta = -alpha*invstd*mean(dz*xhat),
 *               gam = alpha*(mu*invstd*mean(dz*xhat) - mean(dz))
 *   apply    -> dy = alpha*dz + beta*y + gam   (or fused into the consumers: frx_conv_dgrad_bn / frx_conv_wgrad_bn) */
int frx_bn_bwd_partial_rows(int64_t rows, int C);
/* g_pool_hw > 0: `g` is the gradient of an average pool over g_pool_hw pixels, [rows / g_pool_hw][C]; every row takes its
 * image's entry / g_pool_hw (what frx_avgpool_bwd would have stored, without that tensor in memory).  0: g is [rows][C]. */
int frx_bn_bwd_reduce(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g, const void* y,
                      const void* out, const float* scale, const float* shift, int relu, const float* mean,
                      const float* invstd, void* dz_out, float* partial, int g_pool_hw);
/* frx_bn_bwd_reduce adding into replicated totals [replicas][2][C] instead of writing partial rows */
int frx_bn_bwd_reduce_tot(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g, const void* y,
                          const void* out, const float* scale, const float* shift, int relu, const float* mean,
                          const float* invstd, void* dz_out, float* totals, int replicas, int g_pool_hw);
/* frx_bn_bwd_apply with alpha / beta / gam derived from replicated totals */
int frx_bn_bwd_apply_tot(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g, const void* y,
                         const void* out, const float* scale, const float* shift, int relu, const frx_bn_tot* bn, void* dy);
int frx_bn_bwd_finalize(int device, frx_stream_t stream, const float* partial, int nblk, int C, int64_t count,
                        const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                        float* coef);
int frx_bn_bwd_apply(int device, frx_stream_t stream, int dtype, int64_t rows, int C, const void* g, const void* y,
                     const void* out, const float* scale, const float* shift, int relu, const float* mean,
                     const float* invstd, const float* coef, void* dy);
int frx_stem_pool_fwd(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* y,
                      const float* scale, const float* shift, void* out, uint8_t* argmax);
int frx_stem_pool_fwd_tot(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* y,
                          const frx_bn_tot* bn, void* out, uint8_t* argmax);
int frx_stem_pool_bwd(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                      const uint8_t* argmax, void* dpost);
/* The stem's backward in two passes that never store the full-resolution gradient: the max-pool gather of
 * frx_stem_pool_bwd feeds the ReLU mask (scale*y + shift > 0) and the BatchNorm backward directly.
 *   frx_stem_bwd_reduce -> partial [frx_stem_bwd_partial_rows()][2][C]  (then frx_bn_bwd_finalize -> coef)
 *   frx_stem_bwd_apply  -> dy = alpha*dz + beta*y + gam                  (the stem conv's output gradient)
 * Same values as frx_stem_pool_bwd + frx_bn_bwd_reduce(relu) + frx_bn_bwd_apply(relu). */
int frx_stem_bwd_partial_rows(void);
int frx_stem_bwd_reduce(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                        const uint8_t* argmax, const void* y, const float* scale, const float* shift, const float* mean,
                        const float* invstd, float* partial);
int frx_stem_bwd_apply(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                       const uint8_t* argmax, const void* y, const float* scale, const float* shift, const float* coef,
                       void* dy);
int frx_stem_bwd_reduce_tot(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                            const uint8_t* argmax, const void* y, const float* scale, const float* shift, const float* mean,
                            const float* invstd, float* totals, int replicas);
int frx_stem_bwd_apply_tot(int device, frx_stream_t stream, int dtype, int N, int H, int W, int C, const void* dout,
                           const uint8_t* argmax, const void* y, const float* scale, const float* shift,
                           const frx_bn_tot* bn, void* dy);
int frx_avgpool_fwd(int device, frx_stream_t stream, int dtype, int N, int HW, int C, const void* x, void* out);
int frx_avgpool_bwd(int device, frx_stream_t stream, int dtype, int N, int HW, int C, const void* dpool, void* dx);

/* ---------------------------------------------------------------- optimiser / staging
 * frx_sgd_step: optim.SGD(lr, momentum 0.9, weight_decay 5e-4) over ALL parameters in one launch
 *   (model_utils.py:557,186).  lr_dev (optional device scalar) overrides lr, so a captured
 *   hipGraph follows the CustomStepLR schedule (schedulers.py:3-14) without re-capture.
 * frx_input_prep: ToTensor + Normalize(0.5,0.5) (model_utils.py:539-547) fused with the layout
 *   change to the stem's zero-bordered NHWC4; `images` is fp32 NCHW in [-1,1] or uint8 NHWC.
 *   out_elems = element count of `out`; it must equal N*Hp*Wp*4 for this H, W (frx_stem_padded_dims),
 *   otherwise FRX_ERR_ARG (an image of another size would overrun a buffer planned for 112x112). */
int frx_sgd_step(int device, frx_stream_t stream, int64_t n, float* p, const float* g, float* buf,
                 const float* lr_dev, float lr, float momentum, float weight_decay, float grad_scale);
int frx_weight_prep(int device, frx_stream_t stream, int dtype, int Co, int RS, int Ci, const float* master_krsc,
                    void* krsc, void* crsk);
/* every layer in one launch: table_dev [n][8] int64 = {offset of the layer's KRSC master in `master` (floats), Co, RS,
 * Ci, krsc pointer or 0, crsk pointer or 0, index of the layer's first block, mode}; mode 0: a block copies 1024
 * consecutive elements (no CRSK); mode 1: a block owns one 64(co) x 64(ci) tile of one tap (Co, Ci multiples of 64) */
int frx_weight_prep_batched(int device, frx_stream_t stream, int dtype, int n, const int64_t* table_dev,
                            const float* master, int total_blocks);
/* the optimiser step (model_utils.py:186, optim.SGD :557) and the kernel-format copies of the updated weights in ONE
 * launch: frx_sgd_step's arithmetic on p / g / buf, with the blocks laid out by frx_weight_prep_batched's table -- a mode-1
 * block updates its 64 x 64 tile of the fp32 master and writes the KRSC / CRSK copies from the values it holds -- plus
 * mode 2 rows {offset (floats), length (multiple of 4), 0, 0, 0, 0, first block, 2} for the ranges that have no copy
 * (BatchNorm affine, fc bias, margin head): a block updates 1024 consecutive elements.  The rows must cover every
 * parameter exactly once.  zero_grads != 0: g is zeroed as it is consumed (optimizer.zero_grad() of the NEXT step,
 * model_utils.py:184, folded in). */
int frx_sgd_step_prep(int device, frx_stream_t stream, int dtype, int n, const int64_t* table_dev, int total_blocks,
                      float* p, float* g, float* buf, const float* lr_dev, float lr, float momentum, float weight_decay,
                      float grad_scale, int zero_grads);
int frx_input_prep(int device, frx_stream_t stream, int dtype, int N, int H, int W, const void* images,
                   int is_u8_nhwc, void* out, int64_t out_elems);
int frx_cast(int device, frx_stream_t stream, int dtype, int to_f32, int64_t n, const void* x, void* y);
int frx_colsum_f32(int device, frx_stream_t stream, int rows, int C, const float* x, float* out);

#ifdef __cplusplus
}
#endif
#endif /* FRX_H */
