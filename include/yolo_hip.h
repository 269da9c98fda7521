/*
 * yolo_hip.h — C ABI of libyolo_hip.so, the MI355X (gfx950 / CDNA4) execution of the Darknet hot path.
 *
 * The reference (SpursLipu/YOLOv3v4-ModelCompression-MultidatasetTraining-Multibackbone) is pure
 * Python: it has no FFI.  Its "operator interface" for this path is the set of torch.nn / ATen calls
 * issued by models.py and utils/utils.py.  Every entry point below names the reference call it
 * replaces (file:line relative to the reference root).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers unless a
 *    parameter is documented as host.  The library never allocates or frees caller-visible memory and
 *    keeps no reference to caller buffers after a call returns (plans keep the pointers they were
 *    given; the caller keeps those buffers alive for the life of the plan).
 *  - Activations are NHWC ("channels last"): element (n, y, x, c) of a tensor with row pitch `ld`
 *    (in elements, ld >= C, ld % 8 == 0) lives at ((n*H + y)*W + x)*ld + c.  A pitch larger than C is
 *    how route/concat is made zero-copy: producers write channel slices of a wider buffer.
 *  - dtype selects the storage/compute type of activations and packed weights: YH_F16 computes on
 *    MFMA f16 (v_mfma_f32_16x16x32_f16) with fp32 accumulation; YH_F32 computes on the exact-f32 MFMA
 *    (v_mfma_f32_16x16x4_f32).  Bias, BN folding and the epilogue are always fp32.
 *  - Every function is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, and
 *    returns 0 on success, a negative YH_E* code for a rejected argument, or a positive hipError_t.
 */
#ifndef YOLO_HIP_H
#define YOLO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_ABI_VERSION 2

enum { YH_F16 = 0, YH_F32 = 1, YH_I8 = 2 };  /* YH_I8: PTQ eval path on v_mfma_i32_16x16x64_i8 (conv, stem out, pool, copy, qadd) */

/* activation codes: models.py:102-113 (leaky 0.1 / 0.25, relu6, h_swish, relu, mish; else linear) */
enum { YH_ACT_LINEAR = 0, YH_ACT_LEAKY = 1, YH_ACT_RELU = 2, YH_ACT_RELU6 = 3, YH_ACT_HSWISH = 4, YH_ACT_MISH = 5 };

enum {
    YH_OK = 0,
    YH_EINVAL = -1,      /* bad size / null pointer / unsupported combination */
    YH_EALIGN = -2,      /* pointer or pitch not aligned as documented */
    YH_EUNSUPPORTED = -3,
    YH_ENOMEM = -4,      /* host allocation inside a plan failed */
    YH_ERANGE = -5       /* index out of range (plan op / slot) */
};

int yh_abi_version(void);
const char* yh_error_string(int code);

/* ---------------------------------------------------------------------------------------------------
 * Weight packing.  Replaces the host-side BN folding of utils/torch_utils.py:65-89 (fuse_conv_and_bn)
 * and produces the K-major, tile-padded weight image the conv kernels stream.
 *
 *   w          fp32 [cout][cin][kh][kw]   (nn.Conv2d.weight as stored in the state_dict)
 *   conv_bias  fp32 [cout] or NULL        (present iff the block has no BN, models.py:98)
 *   bn_*       fp32 [cout] or all NULL    (BatchNorm2d weight, bias, running_mean, running_var)
 *   cin_map    int32 [cin] or NULL        physical input-channel index of each logical channel
 *                                         (identity when NULL); lets a consumer read padded/concat
 *                                         buffers whose logical channels are not contiguous
 *   cin_k      padded per-tap reduction length: multiple of the K step (32 for f16, 16 for f32) and
 *              > every cin_map entry
 *   m_pad      rows of the packed image: multiple of 128, >= cout; rows >= cout are zero
 *   packed     out, dtype [m_pad][kh*kw][cin_k]   W'[co][tap][map(ci)] = w * gamma/sqrt(var+eps)
 *   bias_out   out, fp32 [m_pad]                  beta - gamma*mean/sqrt(var+eps) (+ scaled conv bias)
 */
int yh_conv_pack_weights(int dtype, const float* w, const float* conv_bias, const float* bn_gamma,
                         const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                         const int32_t* cin_map, int cout, int cin, int kh, int kw, int cin_k, int m_pad,
                         void* packed, float* bias_out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused dense convolution block: y = act(conv(x, W') + b') [+ residual], optionally written 2x
 * nearest-upsampled and/or into a channel slice of a wider buffer.
 * Replaces nn.Sequential(Conv2d, BatchNorm2d, activation) (models.py:92-113) in eval mode, plus the
 * following Shortcut add (utils/layers.py:52-72), the nn.Upsample(scale_factor=2) of models.py:225 and
 * the torch.cat of FeatureConcat (utils/layers.py:35) when the planner fuses them.
 * Implicit GEMM on MFMA: M = cout, N = n*ho*wo, K = kh*kw*cin_k.
 */
typedef struct yh_conv_desc {
    const void* x;        /* dtype, NHWC, pitch ldx; channel offset already applied to the pointer      */
    const void* w;        /* packed weights from yh_conv_pack_weights                                    */
    const float* bias;    /* fp32 [m_pad]                                                                */
    const void* res;      /* dtype residual added after the activation, pitch ldr; NULL for none         */
    void* y;              /* out: dtype (or fp32 when out_f32), pitch ldy                                 */
    int32_t n, h, w_in, cin;      /* input batch, height, width, physical channels reduced over (%8==0)  */
    int32_t ho, wo, cout;         /* output height, width, physical channels stored (%4==0)              */
    int32_t kh, kw, stride, pad;
    int32_t ldx, ldr, ldy;        /* pitches in elements                                                 */
    int32_t cin_k, m_pad;         /* as given to yh_conv_pack_weights                                    */
    int32_t act;                  /* YH_ACT_*                                                            */
    float slope;                  /* leaky slope                                                         */
    int32_t ups;                  /* 1, or 2 = write every output pixel to its 2x2 upsampled block, or   */
                                  /* 3 = phase scatter (see y_h .. y_off_w below)                         */
    int32_t out_f32;              /* store fp32 regardless of dtype (yolo head inputs)                   */
    int32_t dtype;                /* YH_F16 / YH_F32                                                     */
    int32_t tile;                 /* 0 = auto; else forces a tile config (bench/autotune only)           */
    float acc_scale;              /* YH_I8 only: s_w * s_x, turns the int32 accumulator into real units   */
    float out_scale;              /* YH_I8 only: s_a of the block's activation quantizer                  */
    int32_t y_h, y_w;             /* ups == 3 only: output pixel (n, ho, wo) is stored (and res read) at  */
    int32_t y_off_h, y_off_w;     /* (n, 2 ho + y_off_h, 2 wo + y_off_w) of a y_h x y_w tensor; ho/wo are  */
                                  /* then free (taps beyond the input read zeros).  This is one of the four */
                                  /* phases of a stride-2 data gradient, see yh_conv_pack_weights_dgrad_phase */
                                  /* ups == 4: all four phases in one pass - the cout rows are four groups of */
                                  /* cout / 4 channels, group p = 2a + b is stored at (n, 2 ho + a, 2 wo + b) */
                                  /* (skipped beyond y_h x y_w), y has cout / 4 channels; kh = kw = 2, pad 0,  */
                                  /* weight image from yh_pack_batch mode 5.  For memory-bound layers (few     */
                                  /* channels): dz is read once instead of four times and rows are written whole */
    float* stats_ws;              /* training forward: when set, the epilogue also emits per-channel partial sums of  */
    int64_t stats_ws_floats;      /* y and y*y (as stored) — yh_conv2d_stats_rows(d) rows of [2][cout] floats — which  */
                                  /* yh_bn_finalize (nparts) turns into the batch statistics: no separate pass over y */
    /* int8 only, with res != NULL: the quantised shortcut that follows the conv (COSPTQuantizedShortcut_min / _max eval,
     * quantized_ptq_cos.py:877-912) in the same epilogue, exactly yh_qadd's arithmetic on the value q the conv would have stored:
     *   y = clamp(round((round(q * q_rx) * q_scale_x + round(res * q_ra) * q_scale_a) * q_inv_scale_sum))
     * q_rx = conv activation scale / scale_x, q_ra = scale of the routed tensor / scale_a; res is int8 NHWC (pitch ldr).
     * When all five are powers of two with integer q_rx, q_ra <= 2^15 (every calibrated COS-PTQ graph with the `_min` shortcut), every
     * intermediate above is exact and the kernels evaluate clamp(round(q A + res B)), A = q_rx q_scale_x q_inv_scale_sum, B likewise:
     * the same bytes in 7 instead of 16 VALU slots per value (round 5; YH_QADD_POW2=0 keeps the general arithmetic).  */
    float q_rx, q_ra, q_scale_x, q_scale_a, q_inv_scale_sum;
    /* Training backward (round 6; ABI 2).  bwd_z != NULL: the tensor this launch stores (a 1x1 / stride-1 data gradient, with or
     * without its residual accumulate) COMPLETES the gradient dy of a BatchNorm + activation block - autograd of
     * /root/reference/models.py:100-103 - whose pre-BatchNorm output is bwd_z (dtype, pitch bwd_ldz, same pixels and channels as y).
     * The kernel then also emits that block's backward sums over the values it stores, g = dy act'(gamma xhat + beta), xhat = (z - mean)
     * invstd: rows of [sum g | sum g xhat][cout] floats in stats_ws (yh_conv2d_bwd_stats_rows(d) rows; 0 = this launch cannot carry
     * them), which yh_bn_act_bwd_reduce with nparts = rows adds into dbeta / dgamma - the separate reduction pass over dy and z
     * (two reads per element) becomes one extra read of z in a kernel that has the dy row in registers anyway. */
    const void* bwd_z;
    const float* bwd_gamma;
    const float* bwd_beta;
    const float* bwd_mean;
    const float* bwd_invstd;
    int32_t bwd_ldz, bwd_act;
    float bwd_slope;
    int32_t bwd_reserved;
} yh_conv_desc;
int64_t yh_conv2d_stats_rows(const yh_conv_desc* d);
int64_t yh_conv2d_bwd_stats_rows(const yh_conv_desc* d);

int yh_conv2d_fwd(const yh_conv_desc* d, void* stream);
/* YH_I8 form of the block = eval branch of BNFold_COSPTQuantizedConv2d_For_FPGA.forward
 * (utils/quantized/quantized_ptq_cos.py:193-212,288-296,543-567,717): x and w hold the int8 grid values
 * (x = round(x_real/s_x), w = clamp(round(W'/s_w))), bias holds q_bias in real units (int grid * s_b);
 *   y_real = act(acc_i32 * acc_scale + bias);  y_q = clamp(round_half_away(y_real / out_scale), -128, 127)
 * stored as int8, or as fp32 y_q * out_scale when out_f32 (yolo head inputs).  cin % 16, cin_k % 64, no residual.
 * yh_qconv_pack_weights builds the int8 image from the module's q_weight buffer (real units) and its scale.  */
int yh_qconv_pack_weights(const float* q_weight, float w_scale, const int32_t* cin_map, int cout, int cin, int kh, int kw,
                          int cin_k, int m_pad, void* packed, void* stream);
/* Tile configuration yh_conv2d_fwd will use for this descriptor (1 = 128x128, 2 = 64x256, 3 = 32x256,
 * 4 = 64x128, 5 = 128x64, channels x pixels; 2x = the LDS-DMA ring forms of the same tiles, 41 / 43 = halo kernels
 * for 3x3 stride 1, 6x = ping-pong forms, 71 / 72 = the streaming 1x1 / 3x3 kernels for few-channel layers on
 * large grids; bench.py TILE_NAMES has the full list): lets a profiler attribute time to kernel instantiations. */
int yh_conv2d_tile(const yh_conv_desc* d);

/* First-layer convolution straight from the caller's NCHW fp32 image batch (cin <= 4): fuses the
 * NCHW->NHWC relayout and the cast.  Replaces the first Sequential of models.py:92-113 as fed by
 * detect.py:101 / test.py:95 (img / 256 stays with the caller, as in the reference).
 *   x fp32 [n][cin][h][w];  w fp32 [kh*kw*cin][cout_pad] (tap-major, from yh_stem_pack_weights)    */
typedef struct yh_stem_desc {
    const float* x;
    const float* w;
    const float* bias;
    void* y;
    int32_t n, cin, h, w_in, ho, wo, cout, cout_pad, kh, kw, stride, pad, ldy, act;
    float slope;
    int32_t dtype;
    float out_scale;   /* YH_I8 only: the output is quantised, y_q = clamp(round_half_away(y / out_scale)) stored int8 */
    float* stats_ws;           /* training forward (as yh_conv_desc.stats_ws): yh_conv2d_stem_stats_rows(d) rows of [2][cout]   */
    int64_t stats_ws_floats;   /* partial sums of y and y*y as stored, summed by yh_bn_finalize (nparts).  NULL = none.        */
} yh_stem_desc;

int yh_stem_pack_weights(const float* w, const float* conv_bias, const float* bn_gamma, const float* bn_beta,
                         const float* bn_mean, const float* bn_var, float bn_eps, int cout, int cin, int kh,
                         int kw, int cout_pad, float* packed, float* bias_out, void* stream);
int yh_conv2d_stem_fwd(const yh_stem_desc* d, void* stream);
/* rows of partial sums the kernel this descriptor selects emits (0 = it has no statistics epilogue: run yh_bn_stats) */
int64_t yh_conv2d_stem_stats_rows(const yh_stem_desc* d);

/* ---------------------------------------------------------------------------------------------------
 * Depthwise block: y = act(dwconv(x, W') + b'), one k x k filter per channel (k = 3 or 5 in the MobilenetV3 /
 * GhostNet cfgs).  Replaces nn.Sequential(DepthWise2d, BatchNorm2d, activation) (models.py:176-197) and the
 * groups == channels form of models.py:92-98.  HBM-bound: 16-byte channel vectors, weights stay in L1/L2.
 *   yh_dw_pack_weights: w fp32 [c][1][k][k] (+ optional conv bias / BN as in yh_conv_pack_weights); ch_map maps
 *   logical to physical channel (NULL = identity); packed out is dtype [k*k][c_phys], bias_out fp32 [c_phys].   */
typedef struct yh_dw_desc {
    const void* x;
    const void* w;
    const float* bias;
    void* y;
    int32_t n, h, w_in, c, ho, wo, k, stride, pad, ldx, ldy, act;
    float slope;
    int32_t dtype;
} yh_dw_desc;
int yh_dw_pack_weights(int dtype, const float* w, const float* conv_bias, const float* bn_gamma, const float* bn_beta,
                       const float* bn_mean, const float* bn_var, float bn_eps, const int32_t* ch_map, int c, int k,
                       int c_phys, void* packed, float* bias_out, void* stream);
int yh_dwconv2d_fwd(const yh_dw_desc* d, void* stream);

/* Squeeze-excite, SE.forward (utils/layers.py:188-192): y = x * hsigmoid(W2 relu(W1 avgpool(x))), both Linear
 * layers bias-free.  w1 fp32 [cr][c], w2 fp32 [c][cr] (nn.Linear.weight layout, logical channels); pooled and
 * gate are caller-provided fp32 scratch [n][c_phys]; ch_map as above.  Three launches: pool, FC pair, scale.   */
typedef struct yh_se_desc {
    const void* x;
    void* y;
    const float* w1;
    const float* w2;
    float* pooled;
    float* gate;
    const int32_t* ch_map;
    int32_t n, h, w_in, c, c_phys, cr, ldx, ldy, dtype;
} yh_se_desc;
int yh_se_fwd(const yh_se_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * nn.MaxPool2d (models.py:207-215).  pad_lo cells of -inf are implied on the top/left, the window may
 * run past the bottom/right edge: out-of-range taps read `edge_zero ? 0 : -inf` (edge_zero=1 reproduces
 * the ZeroPad2d((0,1,0,1)) + MaxPool2d(2,1) pair of yolov3-tiny).                                       */
typedef struct yh_pool_desc {
    const void* x;
    void* y;
    int32_t n, h, w_in, c, ho, wo, k, stride, pad_lo, edge_zero, ldx, ldy, dtype;
} yh_pool_desc;
int yh_maxpool2d_fwd(const yh_pool_desc* d, void* stream);

/* Channel-slice copy with optional nearest upsample by `ups` (1 or 2): the copying form of
 * FeatureConcat (utils/layers.py:35,38) and nn.Upsample (models.py:225) for the cases the planner
 * cannot make zero-copy.  y[n, y*ups+dy, x*ups+dx, 0:c] = x[n, y, x, 0:c].                              */
typedef struct yh_copy_desc {
    const void* x;
    void* y;
    int32_t n, h, w_in, c, ups, ldx, ldy, dtype;
} yh_copy_desc;
int yh_copy_channels(const yh_copy_desc* d, void* stream);

/* Unfused Shortcut (utils/layers.py:52-72): y[..., 0:c] = a[..., 0:c] + b[..., 0:c].
 * With amap / bmap (device int32[c], both or neither) the operands are gathered per channel instead:
 * y[..., k] = a[..., amap[k]] + b[..., bmap[k]], a negative index contributing zero - the reference's sum over the
 * leading min(Ca, Cb) channels when the operands differ in width (layers.py:65-70), and operands that are concats of
 * padded pieces (GhostNet).  The gather form has no alignment requirement on c.                                      */
typedef struct yh_add_desc {
    const void* a;
    const void* b;
    void* y;
    int64_t pixels;
    int32_t c, lda, ldb, ldy, dtype;
    const int32_t* amap;
    const int32_t* bmap;
} yh_add_desc;
int yh_add_channels(const yh_add_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * int8 (PTQ eval) forms of the data-movement blocks.  Tensors are int8 grid values with one power-of-two scale
 * per tensor kept by the caller; c % 16 == 0, pitches % 16 == 0.
 *  yh_qcopy   y = clamp(round_half_away(x * ratio)) with optional 2x nearest upsample.  ratio = s_in / s_out:
 *             1 for a plain slice copy / upsample, otherwise the re-quantisation COSPTQuantizedFeatureConcat
 *             applies to every routed input (quantized_ptq_cos.py:1540-1545).
 *  yh_qpool   nn.MaxPool2d on grid values (max commutes with the positive scale); same geometry as yh_pool_desc.
 *  yh_qadd    COSPTQuantizedShortcut_min/_max eval (quantized_ptq_cos.py:877-912,1029):
 *               xq = round(x * rx), aq = round(a * ra)            (no clamp; rx = s_x_in / scale_x, ra = s_a_in / scale_a)
 *               y  = clamp(round((xq * scale_x + aq * scale_a) / scale_sum))                                   */
typedef struct yh_qcopy_desc {
    const void* x;
    void* y;
    int32_t n, h, w_in, c, ups, ldx, ldy;
    float ratio;
} yh_qcopy_desc;
int yh_qcopy(const yh_qcopy_desc* d, void* stream);
int yh_qpool(const yh_pool_desc* d, void* stream);   /* dtype field must be YH_I8 */
typedef struct yh_qadd_desc {
    const void* x;
    const void* a;
    void* y;
    int64_t pixels;
    int32_t c, ldx, lda, ldy;
    float rx, ra, scale_x, scale_a, inv_scale_sum;
} yh_qadd_desc;
int yh_qadd(const yh_qadd_desc* d, void* stream);
/* Device self-test of the Mish the int8 epilogues use (csrc/common.h mish_for_grid: a cheap form, the exact form wherever the
 * result lies next to a rounding tie of the activation grid) over the float bit patterns [bits0, bits1), |v| <= 64:
 * out[0] = values whose int8 grid index differs from the exact form's (must be 0), out[1] = largest relative difference of the
 * cheap form in units of 1e-9, out[2] = values decided by the exact form.  out: 3 x uint64 in device memory, zeroed by the caller. */
int yh_qmish_selftest(uint32_t bits0, uint32_t bits1, float inv_s, uint64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * YOLO head decode, YOLOLayer.forward eval branch (models.py:406-418, grid :367-378, anchors :362):
 *   raw[n][a][y][x][o] = p[n][y][x][a*no + o]
 *   io: xy = (sigmoid(t) + cell) * stride, wh = (exp(t) * anchor) * stride, obj/cls = sigmoid(t)
 * p is fp32 NHWC with pitch ldp (the head conv is run with out_f32).  io rows land at
 * io[n][row_off + (a*ny + y)*nx + x][0:no] of an (n, rows_total, no) fp32 tensor, i.e. the torch.cat of
 * models.py:554 is done in place.  raw may be NULL.                                                     */
typedef struct yh_decode_desc {
    const float* p;
    float* io;
    float* raw;
    int32_t n, ny, nx, na, no, ldp, rows_total, row_off;
    float stride;
    float anchor_w[8], anchor_h[8];  /* cfg anchors[mask] / stride in fp32 (models.py:362), na <= 8 */
} yh_decode_desc;
int yh_yolo_decode(const yh_decode_desc* d, void* stream);
/* Decode + step 1 of the NMS below in one pass (round 5): the candidate records yh_yolo_decode followed by yh_nms_candidates would
 * produce for this head - same values, same keys (row = row_off + (a*ny + y)*nx + x of the concatenated tensor) - without writing the
 * (n, rows_total, no) tensor: only rows whose objectness passes conf_thres are decoded (models.py:406-418 feeding utils.py:799-827).
 * d->io and d->raw are ignored; call once per head with the same cand / count buffers; cand == NULL counts only.            */
int yh_yolo_decode_candidates(const yh_decode_desc* d, float conf_thres, int multi_label, const uint8_t* class_mask,
                              float* cand, int32_t* count, int cap, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Non-maximum suppression, utils/utils.py:782-860, batched over the n images of one forward.
 * Candidate records are 8 floats: x1, y1, x2, y2, score, cls, key (int32 bits), 0 — key = row*nc + cls
 * is the position at which the reference would have emitted the candidate.  Buffers are laid out
 * [n][cap][...] with one int32 count per image (counts above cap are clamped when read).
 *
 * 1 yh_nms_candidates  conf filter (:799), wh window (:802), cls*=obj (:809), xywh->xyxy (:812),
 *                      best-class (:819-820) or multi-label expansion (:815-817), class allow-list
 *                      (:823, class_mask uint8[nc] or NULL), finite filter (:827).  Appends records with
 *                      an atomic cursor per image; with cand == NULL it only counts (sizing pass).
 *                      pred fp32 [n][rows][5+nc]; count int32 [n], zeroed by the caller.
 * 2 yh_nms_sort        orders every image's records by score descending, ties by ascending key (the
 *                      stable order of the reference's emission sequence); rank-by-counting, O(m^2).
 * 3 yh_nms_mask        bit j of mask[img][i][j/64] = IoU(box_i + off_i, box_j + off_j) > iou_thres for
 *                      j > i, off = cls*4096 unless agnostic (:840-841); mask is [n][mmax][ceil(mmax/64)].
 * 4 yh_nms_reduce      greedy scan in score order — the torchvision.ops.boxes.nms contract behind :843;
 *                      writes kept indices (ascending = score order) and their number per image.
 * 5 yh_nms_merge       out[img][k] = [merged box, score, cls] of kept box k.  For merge_lo < m < merge_hi
 *                      (reference: 1 < n < 3000, :844-852) the box is sum_j w_j box_j / sum_j w_j with
 *                      w_j = (IoU(kept_k, j) > thr) * score_j over all m records; otherwise the kept box.
 *                      out fp32 [n][cap][6]; kmax >= max kept count (grid sizing).
 * mmax >= max count over the batch (grid and mask sizing), mmax <= cap.
 *
 * Class-segmented form of steps 3 + 4 (round 5; not agnostic, nc <= 255): the class offset of :840 exists so that one
 * torchvision.nms call never suppresses across classes, so one workgroup per (image, class) runs the SAME greedy scan
 * on that class's boxes alone - no IoU bit mask in memory, n x nc scans side by side.
 * 2' yh_nms_sort_cls   = yh_nms_sort, and cls8[img][rank] = class of every sorted position (uint8 [n][cap]).
 * 2" yh_nms_sort_tiles = 2' in O(m log m): tiles of 2048 records sorted in LDS (bitonic, same 64-bit keys), global rank = rank in the
 *                      own tile + one binary search per other tile; cap must be a power of two >= 256; cls8 may be NULL;
 *                      ws: 12 bytes per record slot (n * cap * 12), 16-byte aligned.
 * 3' yh_nms_class_scan keep8 uint8 [n][cap] (1 = kept, per sorted position), then keep_idx / n_keep as step 4 writes
 *                      them.  state uint32 [n][8], initialised by the caller to {0, ~0, ~0, 0, 0, 0, 0, 0}: [0] bit 0 =
 *                      a class held more than 2048 candidates; [1..4] = order-preserving encodings of min x1, min y1,
 *                      max x2, max y2 of the image's candidates; [5] = 1 when the image needs the general steps 3 + 4
 *                      instead: a class overflowed, or the candidates span more than 4096 in x AND in y, in which case
 *                      boxes of different classes could overlap after the offset (below that they provably cannot
 *                      and the result is the general form's bit for bit).  n_keep[img] = 0 for such an image.      */
int yh_nms_candidates(const float* pred, int n, int rows, int nc, float conf_thres, int multi_label,
                      const uint8_t* class_mask, float* cand, int32_t* count, int cap, void* stream);
int yh_nms_sort(const float* cand, const int32_t* count, int n, int cap, int mmax, float* sorted, void* stream);
int yh_nms_sort_cls(const float* cand, const int32_t* count, int n, int cap, int mmax, float* sorted, uint8_t* cls8,
                    void* stream);
int yh_nms_sort_tiles(const float* cand, const int32_t* count, int n, int cap, int mmax, float* sorted, uint8_t* cls8,
                      void* ws, size_t ws_bytes, void* stream);
int yh_nms_class_scan(const float* sorted, const uint8_t* cls8, const int32_t* count, int n, int cap, int nc,
                      float iou_thres, uint8_t* keep8, uint32_t* state, int32_t* keep_idx, int32_t* n_keep, void* stream);
int yh_nms_mask(const float* sorted, const int32_t* count, int n, int cap, int mmax, float iou_thres, int agnostic,
                uint64_t* mask, void* stream);
int yh_nms_reduce(const uint64_t* mask, const int32_t* count, int n, int cap, int mmax, int32_t* keep_idx,
                  int32_t* n_keep, void* stream);
int yh_nms_merge(const float* sorted, const int32_t* count, const int32_t* keep_idx, const int32_t* n_keep, int n,
                 int cap, int kmax, float iou_thres, int agnostic, int merge_lo, int merge_hi, float* out,
                 void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Training path: train-mode BatchNorm + activation around the conv kernels above, and the
 * backward kernels.  Replaces nn.BatchNorm2d in training mode (models.py:100, momentum 0.1, eps 1e-5), the
 * activation modules and their autograd, and autograd's conv backward.  z = conv output (dtype), statistics fp32.
 *
 *  yh_bn_stats      sum[c] += sum_p z, sumsq[c] += sum_p z^2 over all pixels (caller zeroes sum/sumsq, fp32 [c]).
 *  yh_bn_finalize   mean = sum/P, var = sumsq/P - mean^2 (biased); invstd = 1/sqrt(var+eps);
 *                   running_mean = (1-m) running_mean + m mean; running_var = (1-m) running_var + m var P/(P-1).
 *  yh_bn_act_fwd    y = act(gamma (z-mean) invstd + beta) [+ res]; optional 2x upsampled write (ups).
 *  yh_bn_act_bwd_reduce   g = dy * act'(u), u = gamma xhat + beta, xhat = (z-mean) invstd:
 *                         dbeta[c] += sum_p g, dgamma[c] += sum_p g xhat   (caller zeroes, fp32)
 *  yh_bn_act_bwd_apply    dz = gamma invstd (g - dbeta/P - xhat dgamma/P)
 * For blocks without BN (gamma == NULL in the desc) fwd is y = act(z + bias) and bwd is dz = dy * act'(z + bias)
 * with dbeta = the bias gradient.                                                                              */
typedef struct yh_bn_desc {
    const void* z;          /* conv output, dtype, pitch ldz                                                  */
    const void* dy;         /* bwd: gradient w.r.t. the block output (after residual add), pitch lddy         */
    const void* res;        /* fwd: residual added after the activation (NULL for none), pitch ldr            */
    void* out;              /* fwd: y (pitch ldo); bwd apply: dz (pitch ldo)                                   */
    const float* gamma;     /* [c] or NULL (block without BN)                                                  */
    const float* beta;      /* [c] BN beta, or the conv bias when gamma == NULL (may be NULL = 0)              */
    float* mean;            /* [c] batch mean (finalize writes, others read)                                   */
    float* invstd;          /* [c]                                                                             */
    float* sum;             /* [c] stats: sum z     / bwd: dbeta  accumulator                                  */
    float* sumsq;           /* [c] stats: sum z^2   / bwd: dgamma accumulator                                  */
    float* running_mean;    /* [c] or NULL                                                                     */
    float* running_var;     /* [c] or NULL                                                                     */
    int64_t pixels;         /* n*h*w of z                                                                      */
    int32_t n, h, w_in;     /* geometry of z (needed for ups)                                                  */
    int32_t c, ldz, lddy, ldr, ldo, act, ups, dtype;
    float slope, eps, momentum;
    int32_t nparts;         /* yh_bn_finalize only: > 0 = first add `nparts` rows of [2][c] partial sums at ws (written by  */
                            /* yh_conv2d_fwd with stats_ws) into sum / sumsq                                              */
    float* ws;              /* optional workspace for the two reductions (yh_bn_stats, yh_bn_act_bwd_reduce): per-workgroup */
    int64_t ws_floats;      /* partial sums are stored there and summed by a second launch instead of contended atomics; */
                            /* size from yh_bn_reduce_workspace().  NULL / too small -> fp32 atomics.                     */
} yh_bn_desc;
int64_t yh_bn_reduce_workspace(const yh_bn_desc* d);
int yh_bn_stats(const yh_bn_desc* d, void* stream);
int yh_bn_finalize(const yh_bn_desc* d, void* stream);
int yh_bn_act_fwd(const yh_bn_desc* d, void* stream);
int yh_bn_act_bwd_reduce(const yh_bn_desc* d, void* stream);
int yh_bn_act_bwd_apply(const yh_bn_desc* d, void* stream);

/* Backward of the convolution itself (replaces autograd's conv2d backward behind nn.Conv2d, models.py:88-99).
 *
 *  data gradient   = the forward kernel run on dz with the transposed, spatially flipped weight image:
 *                    yh_conv_pack_weights_dgrad builds [m_pad rows = cin][kh*kw flipped][cout_k] from the fp32 OIHW
 *                    parameter; call yh_conv2d_fwd with x = dz, stride 1, pad = k-1-pad, act linear, res = y = the
 *                    gradient buffer of the input when it already holds another consumer's contribution.
 *                    Stride-2 layers run as four parity-phase correlations of dz (yh_conv_pack_weights_dgrad_phase +
 *                    ups = 3), or as one four-phase pass when they are memory bound (yh_pack_batch mode 5 + ups = 4);
 *                    yh_dilate2 (dz scattered onto the even positions of a zeroed 2x buffer) remains for callers that
 *                    want the plain dilated form.
 *  weight gradient = yh_conv2d_wgrad: dw[co][ci][r][s] += sum_pixels dz[p][co] * x[p shifted by tap][ci], fp32 OIHW,
 *                    accumulated (caller zeroes dw): per-split partial tiles in the workspace summed by a second
 *                    launch, or fp32 atomics without a workspace.  MFMA with the pixel index as K.
 *  yh_stem_wgrad   the same for the first layer straight from the fp32 NCHW image (cin = 3, 3x3; scalar kernel).  The engine
 *                  prefers yh_nchw_to_nhwc + yh_conv2d_wgrad(cin_w = 3), which runs the layer on the MFMA kernel (6x faster).
 *  yh_upsample2_bwd  dx[n,h,w,c] = sum of the 2x2 block of dy (backward of the fused nearest-neighbour store).
 *  yh_cast_f32     fp32 pitched rows -> dtype pitched rows (head gradients arrive from autograd as fp32).          */
int yh_conv_pack_weights_dgrad(int dtype, const float* w, int cout, int cin, int kh, int kw, int cout_k, int m_pad,
                               void* packed, void* stream);
/* Stride-2 data gradient without the zero-dilated copy: the input pixels of parity (a, b) = (h & 1, w & 1) only see
 * the taps r = a + pad - 2t, s = b + pad - 2u (t, u >= 0), i.e. a (kh_p x kw_p)-tap correlation of dz whose window
 * starts at (h >> 1, w >> 1).  This packs that phase's image [m_pad rows = cin][kh_p*kw_p][cout_k] and returns
 * kh_p / kw_p; run yh_conv2d_fwd with x = dz, kh = kh_p, kw = kw_p, stride 1, pad 0, ups = 3, y_off = (a, b).
 * For 3x3 / pad 1 the four phases have 1, 2, 2 and 4 taps: 9 tap-GEMMs instead of the 36 of the dilated form.      */
int yh_conv_pack_weights_dgrad_phase(int dtype, const float* w, int cout, int cin, int kh, int kw, int pad, int a, int b,
                                     int cout_k, int m_pad, void* packed, int* kh_p, int* kw_p, void* stream);
/* ---------------------------------------------------------------------------------------------------
 * compute_loss (utils/utils.py:368-432) and its gradient on the raw head tensors, fl_gamma == 0, mean reduction:
 *   lbox = giou_gain * sum_heads mean_i (1 - GIoU(box(ps_i), tbox_i))
 *   lobj = obj_gain  * sum_heads mean_cells BCE(p[..., 4], tobj),  tobj = (1 - gr) + gr * max(GIoU, 0) at matched cells
 *   lcls = cls_gain  * sum_heads mean_{i, c} BCE(ps_i[5 + c], cp if c == tcls_i else cn)          (nc > 1 only)
 * with box(ps) = (sigmoid(ps[0:2]), min(exp(ps[2:4]), 1e3) * anchor).  The target assignment (build_targets,
 * utils.py:725-779) happens inside the kernels, so a step needs no host round trip: candidate k = a * nt + t (anchor a,
 * label t, the reference's order) is matched when wh_iou(anchors[a], label wh in grid units) > iou_t; its cell is
 * (image, a, int(y * ny), int(x * nx)) and its box (frac(x * nx), frac(y * ny), w * nx, h * ny).  Raw tensors are
 * addressed through element strides, so the (bs, na, ny, nx, no) views of NHWC head buffers are read in place.
 *
 *  yh_yolo_loss_fwd   fills tobj (caller zeroes it) and adds the three un-normalised sums to sums[0..2]
 *                     (caller zeroes; lbox: sum (1 - giou), lobj: sum bce, lcls: sum bce) for one head; adds the number
 *                     of matched candidates to count[0] and ORs a label error mask into count[1] (1: class outside
 *                     [0, nc), 2: image or cell outside the head; such candidates are skipped) - caller zeroes both.
 *  yh_yolo_loss_bwd   writes d(loss)/d(p) for EVERY logical element of the head (zeros included) scaled by *scale
 *                     (device scalar: autograd's grad_output), gradient tensor addressed by its own strides; the mean
 *                     weights g_box / nb, g_obj / cells, g_cls / (nb * nc) use nb = count[0] from the forward.      */
typedef struct yh_loss_desc {
    const float* p;             /* raw head, element (b, a, y, x, o) at p + b*sb + a*sa + y*sy + x*sx + o          */
    float* grad;                /* bwd only, same indexing with gb, ga, gy, gx                                       */
    float* tobj;                /* (bs, na, ny, nx) dense fp32                                                       */
    int32_t* winner;            /* (bs, na, ny, nx) int32, caller fills with -1: when several candidates match one  */
                                /* cell the LAST one sets tobj, like the reference's sequential index_put on the CPU */
    const float* targets;       /* (nt, 6): image, class, x, y, w, h (normalised to the image)                       */
    const float* anchors;       /* (na, 2): anchor wh in grid units (YOLOLayer.anchor_vec)                           */
    float* sums;                /* fwd: [3] accumulators                                                             */
    int32_t* count;             /* [2]: matched candidates, label error mask                                         */
    const float* scale;         /* bwd: device scalar multiplied into every gradient                                 */
    int64_t sb, sa, sy, sx, gb, ga, gy, gx;
    int32_t bs, na, ny, nx, no, nc, nt;
    float iou_t, gr, cp, cn, cls_pw, obj_pw;
    float g_box, g_obj, g_cls;  /* hyp['giou'], hyp['obj'], hyp['cls']                                              */
} yh_loss_desc;
int yh_yolo_loss_fwd(const yh_loss_desc* d, void* stream);
int yh_yolo_loss_bwd(const yh_loss_desc* d, void* stream);

/* Backward of the depthwise block and of squeeze-excite (training of the Mobilenet / Ghost backbones).
 *  yh_dw_wgrad   dw[c][r][s] += sum_pixels dz[p][c] * x[p shifted by (r, s)][c]   (fp32 [c][k][k], caller zeroes)
 *  yh_dw_dgrad   dx[hi, wi, c] (+)= sum_taps dz[(hi + pad - r) / stride, (wi + pad - s) / stride, c] * w[r][s][c] over the
 *                taps whose source position is integral and inside dz; w is the FORWARD packed image [k*k][c]
 *  yh_se_bwd     y = x * g, g = hsigmoid(W2 relu(W1 mean(x))):  dx (+)= dy * g + W1^T(relu' (W2^T(hsig' (sum_p dy x)))) / HW,
 *                dw1 / dw2 += the two outer products (fp32, nn.Linear layout); pooled / gate are the forward's buffers,
 *                scratch is fp32 [n][c].                                                                              */
typedef struct yh_dw_bwd_desc {
    const void* x;          /* wgrad: forward input of the block                                                     */
    const void* dz;         /* gradient of the (pre-BN) depthwise output                                             */
    const void* w;          /* dgrad: forward packed weights [k*k][c] dtype                                          */
    void* dx;               /* dgrad output                                                                          */
    float* dw;              /* wgrad output                                                                          */
    int32_t n, h, w_in, c, ho, wo, k, stride, pad, ldx, lddz, lddx, accumulate, dtype;
    float* ws;              /* wgrad, optional (ABI 2): yh_dw_wgrad_workspace(d) floats - per-workgroup partial rows summed by a second */
    int64_t ws_floats;      /* launch in a fixed order; without it the workgroups meet in fp32 atomics (contended: 8 x slower)         */
} yh_dw_bwd_desc;
int64_t yh_dw_wgrad_workspace(const yh_dw_bwd_desc* d);
int yh_dw_wgrad(const yh_dw_bwd_desc* d, void* stream);
int yh_dw_dgrad(const yh_dw_bwd_desc* d, void* stream);
typedef struct yh_se_bwd_desc {
    const void* x; const void* dy; void* dx;
    const float* w1; const float* w2; const float* pooled; const float* gate;
    float* dw1; float* dw2; float* scratch;
    int32_t n, h, w_in, c, cr, ldx, lddy, lddx, accumulate, dtype;
    float* scratch2;        /* optional (ABI 2): fp32 [n][c + 2 cr] - the per-image factors of the two weight gradients, which a second  */
    int64_t scratch2_floats;/* launch sums over the images in order; without it every image adds its outer products by fp32 atomics    */
} yh_se_bwd_desc;
int yh_se_bwd(const yh_se_bwd_desc* d, void* stream);

/* All weight images of a training step in ONE launch (the parameters change every optimizer step; 75 layers x
 * (forward image + data-gradient image[s]) would otherwise be ~170 tiny launches).  `items` is a DEVICE array.       */
typedef struct yh_pack_item {
    const float* w;         /* fp32 OIHW parameter                                                              */
    const float* bias;      /* conv bias or NULL                                                                */
    void* packed;           /* destination image                                                                */
    float* bias_out;        /* modes 0 and 3: fp32 bias row (zeros when bias == NULL); else NULL                */
    int32_t mode;           /* 0 forward [m_pad][kh*kw][k_pad] (yh_conv_pack_weights without BN)                 */
                            /* 1 data gradient [m_pad = cin rows][flipped taps][k_pad] (yh_conv_pack_weights_dgrad) */
                            /* 2 one parity phase of a stride-2 data gradient (.._dgrad_phase, pa/pb)            */
                            /* 3 first layer [kh*kw*cin][cout_pad] fp32 (yh_stem_pack_weights without BN)        */
                            /* 4 depthwise [kh*kw][k_pad = c_phys] (yh_dw_pack_weights without BN), cout = channels */
                            /* 5 all four parity phases of a stride-2 data gradient: rows [4][cout_pad = channels per  */
                            /*   phase], 2x2 taps (t, u) holding w[co][ci][a + pad - 2t][b + pad - 2u] or zero (ups == 4) */
    int32_t dtype, cout, cin, kh, kw, k_pad, m_pad, pad, pa, pb, cout_pad;
} yh_pack_item;
int yh_pack_batch(const yh_pack_item* items, int n_items, void* stream);

typedef struct yh_wgrad_desc {
    const void* x;          /* forward input of the conv, NHWC dtype (stem: NCHW fp32 image)                    */
    const void* dz;         /* gradient of the conv output, NHWC dtype, pitch lddz                              */
    float* dw;              /* [cout][cin][kh][kw] fp32, accumulated                                            */
    int32_t n, h, w_in, cin, ho, wo, cout, kh, kw, stride, pad, ldx, lddz, dtype;
    int32_t splits;         /* pixel-range splits (0 = library heuristic)                                       */
    int32_t cin_w;          /* input channels of dw (0 = cin): the first layer reads the image through an 8-channel  */
                            /* NHWC copy (yh_nchw_to_nhwc) whose channels 3..7 are zero and have no dw entries       */
    float* ws;              /* optional workspace: per-split partial tiles, summed by a second launch instead of atomics; */
    int64_t ws_floats;      /* size from yh_conv2d_wgrad_workspace().  NULL / too small -> fp32 atomics.                  */
} yh_wgrad_desc;
int64_t yh_conv2d_wgrad_workspace(const yh_wgrad_desc* d);
/* The fp16 kernel splits the pixel axis over about one resident wave of workgroups (768 for the 128-row tile, 512 for the
 * 256-row tile it uses on K-heavy layers with cout % 256 == 0, 1024 for the 64-row tile) and hands consecutive splits to
 * the same XCD so that all tiles of a split share one L2.  Tuning knobs read from the environment (A/B measurements
 * only): YH_WGRAD_TARGET = workgroups aimed for (negative = round the split count down), YH_WGRAD_BM = 128 / 256 forces the
 * row tile, YH_WGRAD_BN = 256 selects the 8-wave 128 x 256 tile, YH_WGRAD_XCD = 0 restores the plain (tile, split) grid.
 * splits = -1 in the descriptor selects the register-staged fp16 kernel.
 * 3x3 / stride 1 / pad 1 layers with cout % 128 == 0, cin % 64 == 0, 16 <= W <= 190 and a workspace run on the rolling-halo kernel of
 * round 5 (csrc/conv_wgrad_roll.hip: 128 x [9 x 64] tile per workgroup, fragments refreshed between the MFMAs); knobs: YH_WGRAD_HALO =
 * 0 (im2col kernels only) / 1 (the round-3 halo kernel) / 2 (default) / 3, YH_WGRAD_ROLL_ORDER = 0..3 (K-loop forms), YH_WGRAD_ROLL_STAGES,
 * YH_WGRAD_ROLL_MFMA32 = 1 (v_mfma_f32_32x32x16_f16 form), YH_WGRAD_ROLL_REDUCE = 1 (scattering reduce), YH_WGRAD_HALO_WGS.               */
int yh_conv2d_wgrad(const yh_wgrad_desc* d, void* stream);
/* Which kernel yh_conv2d_wgrad will launch for this descriptor (host code, no launch; with d->ws == NULL the answer assumes the
 * workspace of yh_conv2d_wgrad_workspace() will be bound, as the training plan does): 91 = conv_wgrad_roll_kernel (3x3 / s1 rolling-halo
 * form, round 5), 90 = conv_wgrad_halo_kernel (3x3 / s1 halo form, round 3), 10 * TM + WNW for conv_wgrad_dma_kernel<TM, WNW> (22 = 64-row tile, 42 = 128 x 128, 44 = 128 x 256, 82 = 256 x 128,
 * 84 = 256 x 256), 1 = the register-staged kernel (fp32, or splits == -1).  bench.py uses it to name the dominant kernel of
 * the weight-gradient class and to attach that kernel's HBM counters.                                   */
int yh_conv2d_wgrad_kernel(const yh_wgrad_desc* d);
int yh_stem_wgrad(const yh_wgrad_desc* d, void* stream);

/* First conv block, whole backward in ONE pass over dy and z (round 4).  The first layer has no data gradient, so its dz is only
 * ever read by its own weight gradient; reference: autograd through models.py:92-113 (Conv2d -> BatchNorm2d -> activation) of
 * block 0, i.e. what yh_bn_act_bwd_reduce + yh_bn_act_bwd_apply + yh_nchw_to_nhwc + yh_conv2d_wgrad compute in four launches
 * (reading dy and z twice, writing and re-reading dz: 9 bytes per element more than this form).  With g = dy * act'(u),
 * xhat = (z - mean) invstd, X = im2col of the image:
 *     dbeta = S1 = sum_p g,   dgamma = S2 = sum_p g xhat,
 *     dz = gamma invstd (g - S1/P - xhat S2/P)                                   (batch-statistics BatchNorm backward)
 *     dW[c][k] = sum_p dz[c][p] X[k][p] = gamma_c invstd_c (Q[c][k] - S1_c/P SX[k] - S2_c/P R[c][k])
 * with Q = g X^T, R = xhat X^T, SX = sum_p X: all three are sums over pixels that one pass can take (fp16 MFMA operands, fp32
 * accumulation, the final combination in double).  3x3 / stride 1 / pad 1, cin <= 3, cout 16 or 32, fp16 activations; other
 * first layers keep the four-launch form.  dgamma / dbeta / dw are ADDED to (the gradient arena is zeroed per backward).      */
typedef struct yh_stem_bwd_desc {
    const float* x;         /* the fp32 NCHW image batch [n][cin][h][w]                                           */
    const void* dy;         /* gradient of the block output, NHWC f16, pitch lddy                                 */
    const void* z;          /* conv output before BatchNorm, NHWC f16, pitch ldz                                  */
    const float* gamma; const float* beta; const float* mean; const float* invstd;   /* [cout] fp32               */
    float* dgamma; float* dbeta;   /* [cout] fp32, added to                                                        */
    float* dw;              /* [cout][cin][3][3] fp32, added to                                                   */
    float* ws;              /* workspace of yh_stem_bwd_workspace() floats: per-workgroup partial rows            */
    int64_t ws_floats;
    int32_t n, cin, h, w_in, cout, lddy, ldz, act;
    float slope;
    /* Optional: the data gradient of the NEXT conv fused in (dz1 != NULL; then `dy` is not read at all).  When the first block's
     * output has exactly one consumer and that consumer is a 3x3 / stride 2 / pad 1 conv with 64 output channels (Darknet-53's
     * conv1, CSPDarknet53's conv1), dy = conv_transpose(dz1, w1) is computed per 32-pixel segment from the 1 or 2 rows of dz1 it
     * depends on (12 or 24 MFMAs) instead of being written by yh_conv2d_fwd(ups = 4) and read back: 1.5 GB less to write and 1.5 GB
     * less to read at 608 x 608 x 32 x batch 64 (reference: autograd through models.py:92-113 of blocks 0 and 1).  cout must be 32. */
    const void* dz1;        /* gradient of the next conv's output, NHWC f16 [n][h1][w1][k1 = 64], pitch lddz1      */
    const void* w1;         /* its weights as yh_conv_pack_weights_dgrad packs them: [m1_pad rows >= cout][9 flipped taps][k1_pad] f16 */
    int32_t h1, w1_in, k1, k1_pad, lddz1;
} yh_stem_bwd_desc;
int64_t yh_stem_bwd_workspace(const yh_stem_bwd_desc* d);
int yh_stem_bwd(const yh_stem_bwd_desc* d, void* stream);
typedef struct yh_resample_desc {
    const void* x; void* y;
    int32_t n, h, w_in, c;  /* geometry of the SMALL tensor (dilate2: source; upsample2_bwd: destination)       */
    int32_t big_h, big_w;   /* geometry of the LARGE tensor: 2h x 2w, or 2h-1 / 2w-1 for dilate2 onto an odd input  */
    int32_t ldx, ldy, dtype;
} yh_resample_desc;
int yh_dilate2(const yh_resample_desc* d, void* stream);
int yh_upsample2_bwd(const yh_resample_desc* d, void* stream);
typedef struct yh_cast_desc {
    const float* x; void* y; int64_t pixels; int32_t c, ldx, ldy, dtype;
} yh_cast_desc;
int yh_cast_f32(const yh_cast_desc* d, void* stream);
/* Backward of yh_maxpool2d_fwd (autograd of nn.MaxPool2d, models.py:207-215): every output element adds its gradient
 * to the FIRST maximum of its window in row-major scan order (the index torch's forward records); windows that
 * overlap (SPP, stride 1) meet in atomics, so dx must hold zeros or an earlier contribution.  With edge_zero a
 * window whose maximum is the implicit zero padding passes no gradient.                                            */
typedef struct yh_pool_bwd_desc {
    const void* x;          /* forward input, dtype, pitch ldx                                                 */
    const void* dy;         /* gradient of the pooled output, pitch lddy                                       */
    void* dx;               /* accumulated into, pitch lddx                                                    */
    int32_t n, h, w_in, c, ho, wo, k, stride, pad_lo, edge_zero, ldx, lddy, lddx, dtype;
} yh_pool_bwd_desc;
int yh_maxpool2d_bwd(const yh_pool_bwd_desc* d, void* stream);
/* fp32 NCHW image (n, c, h, w) -> dtype NHWC (n, h, w, ldy) with channels c..c_pad-1 written as zeros (c_pad <= ldy):
 * the first layer's weight gradient then runs on the MFMA kernel like every other layer.                          */
int yh_nchw_to_nhwc(const float* x, void* y, int n, int c, int h, int w, int c_pad, int ldy, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Input pipeline on the device (first slice): letterbox of ONE uint8 HWC frame into its slot of the fp32 NCHW batch.
 * Replaces utils/datasets.py letterbox (reference datasets.py:611-646: resize + constant border), the BGR->RGB / HWC->CHW shuffle
 * (datasets.py:112-113) and `torch.from_numpy(img).to(device).float() / 256.0` (detect.py:99-101).
 *   tmp[y][x][ch]     = clip8(2^21 + sum_i src[y][hbounds[2x] + i][ch] * hk[x * hksize + i]),  i < hbounds[2x+1]   (horizontal pass)
 *   v(Y, X, ch)       = clip8(2^21 + sum_i tmp[vbounds[2y] + i][x][ch] * vk[y * vksize + i])  for (y, x) = (Y - top, X - left) inside
 *                       the new_h x new_w image area, pad_value elsewhere                                         (vertical pass)
 *   dst[ch'][Y][X]    = scale * v + shift,  ch' = 2 - ch when swap_rb (3 channels), else ch
 * with clip8(s) = min(max(s >> 22, 0), 255): Pillow's 8-bit resampling; the bounds / coefficient tables come from the host
 * (engine/preprocess.py restates precompute_coeffs / normalize_coeffs_8bpc).  All pointers are device pointers.
 *
 * arith selects the resampling arithmetic.  The values other than YH_ARITH_PILLOW evaluate OpenCV's uint8 formulas - the library the
 * reference's loaders call (cv2.resize, reference datasets.py:519-526, :637) - in one fused pass (tmp unused), tables from
 * engine/imgtables.py:
 *   YH_ARITH_CV2_LINEAR     hbounds / vbounds = the two source indices per output sample, hk / vk = their 11-bit weights (ksize 2):
 *                           r_j = S[j][i0] a0 + S[j][i1] a1;  v = (((b0 (r_0 >> 4)) >> 16) + ((b1 (r_1 >> 4)) >> 16) + 2) >> 2
 *   YH_ARITH_CV2_AREA       hbounds / vbounds = (first source sample, count), hk / vk = float32 weights [ksize] (bit pattern in the
 *                           int32 array): buf = sum_u S[j][x0 + u] ka[u], sum (+)= kb[t] buf, all float32 in this order; round half even
 *   YH_ARITH_CV2_AREA_FAST  integer decimation factors hksize x vksize, no tables: integer cell sums; (s + 2) >> 2 for 2 x 2,
 *                           round(s * (1.f / area)) otherwise, ragged edge cells divided by their sample count
 * out_u8 != 0 (cv2 arithmetics only) writes a uint8 HWC image [out_h][out_w][c] to dst instead of the scaled fp32 planes.     */
enum { YH_ARITH_PILLOW = 0, YH_ARITH_CV2_LINEAR = 1, YH_ARITH_CV2_AREA = 2, YH_ARITH_CV2_AREA_FAST = 3 };
typedef struct yh_letterbox_desc {
    const uint8_t* src;          /* [h0][src_pitch bytes], c interleaved channels                                   */
    uint8_t* tmp;                /* [h0][new_w][c] scratch                                                          */
    float* dst;                  /* [c][out_h][out_w] planes of this frame                                          */
    const int32_t* hbounds;      /* [new_w][2]  (first source column, count)                                        */
    const int32_t* hk;           /* [new_w][hksize] fixed-point coefficients (22 fractional bits)                   */
    const int32_t* vbounds;      /* [new_h][2]                                                                      */
    const int32_t* vk;           /* [new_h][vksize]                                                                 */
    int32_t h0, w0, c, src_pitch, hksize, vksize;
    int32_t new_h, new_w, out_h, out_w, top, left;
    int32_t pad_value, swap_rb;
    float scale, shift;          /* detect.py:101: 1/256, 0;  --maxabsscaler: 2/256, -1                              */
    int32_t arith, out_u8;
} yh_letterbox_desc;
int yh_letterbox_fwd(const yh_letterbox_desc* d, void* stream);

/* Input pipeline on the device, second slice: ONE training item = 4-image mosaic -> random affine warp (bilinear, constant
 * border) -> HSV augmentation -> left-right flip -> planar CHW uint8 or x / divisor float.  Replaces utils/datasets.py
 * load_mosaic + random_affine + augment_hsv + the flip / transpose of __getitem__ (reference datasets.py:553-608, 649-715, 534-550,
 * 470-505) and `imgs.float() / 256.0` (train.py:345).  The random draws, the label geometry and the decode stay on the host
 * (engine/preprocess.py); the arithmetic is the host loader's operation for operation (Pillow's double-precision bilinear
 * transform, numpy's float32 / float64 HSV formulas), so the output is bit-identical to it.
 *   canvas(cy, cx)  = src[k][y1b[k] + cy - y1a[k]][x1b[k] + cx - x1a[k]] for the rectangle k with x1a <= cx < x2a, y1a <= cy < y2a,
 *                     pad_value where no rectangle covers (the canvas is never materialised); empty rectangles are skipped
 *   warp(Y, X)      = bilinear sample of canvas at (inv[0] (X'+.5) + inv[1] (Y+.5) + inv[2], inv[3] (X'+.5) + inv[4] (Y+.5) + inv[5])
 *                     with X' = out_w - 1 - X when flip_lr; pad_value when that point is outside the canvas
 *   dst[ch][Y][X]   = hsv(warp) (3 channels and hsv != 0; gains hsv_gain[0..2] on hue / saturation / value)
 *
 * arith = YH_ARITH_CV2_LINEAR: the reference's own library calls instead, restated from OpenCV (3.x - 4.10) - warpAffine's fixed
 * point bilinear (10-bit coordinates, 5-bit sub-pixel weights summing to 2^15, border = pad_value) with `inv` = OpenCV's inversion
 * of the forward matrix (engine/imgtables.py cv2_invert_affine) and X' applied as above; cvtColor(BGR2HSV) in 12-bit fixed point,
 * the three 256-entry tables `lut` = [hue][sat][val] of augment_hsv (datasets.py:539-542; hsv_gain is not read),
 * cvtColor(HSV2BGR) in float32.  src[k] are then the images already resized by yh_letterbox_fwd (cv2 arithmetic, out_u8).          */
enum { YH_MOSAIC_U8 = 0, YH_MOSAIC_F32 = 1, YH_MOSAIC_F16 = 2 };
typedef struct yh_mosaic_desc {
    const uint8_t* src[4];       /* decoded frames, [src_h][src_pitch bytes], c interleaved channels (device pointers)    */
    void* dst;                   /* [c][out_h][out_w]: uint8, or float / half holding value / divisor                      */
    double inv[6];               /* output pixel centre -> canvas point (rows of the inverse affine matrix)                */
    double hsv_gain[3];
    int32_t src_h[4], src_w[4], src_pitch[4];
    int32_t x1a[4], y1a[4], x2a[4], y2a[4];   /* placement rectangle on the canvas                                        */
    int32_t x1b[4], y1b[4];                   /* its top-left corner in the source frame                                   */
    int32_t canvas_h, canvas_w, out_h, out_w, c, pad_value, hsv, flip_lr, out_dtype;
    float divisor;               /* train.py:345: 256                                                                      */
    int32_t arith;               /* YH_ARITH_PILLOW or YH_ARITH_CV2_LINEAR                                                 */
    const uint8_t* lut;          /* [3][256] device table (cv2 arithmetic with hsv != 0)                                   */
} yh_mosaic_desc;
int yh_mosaic_affine_hsv(const yh_mosaic_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * COS-PTQ calibration on the device.  The scale search every quantiser of the reference runs per calibration batch
 * (utils/quantized/quantized_ptq_cos.py:64-93 Quantizer.forward, training branch; :838-912 COSPTQuantizedShortcut_min and
 * :1153-1197 _max; the weight / bias quantisers of BNFold_COSPTQuantizedConv2d_For_FPGA :193-212 use the same search):
 *   for j in 0 .. n-1:  scale_j = scale0 * 2^j;  q_j = clamp(round_half_away(t / scale_j), lo, hi) * scale_j  (clamp iff do_clamp)
 *   cos_out[j] = <t, q_j> / (|t| |q_j|)  (0 when a norm vanishes);  *best = the FIRST j with the largest cosine
 * in ONE pass over t: fp32 element arithmetic exactly as the modules', sums in double.  t: fp32, `count` elements in any order (a
 * dense tensor of any layout); n <= 16; ws: yh_ptq_search_workspace(count) bytes of device scratch; cos_out (double[n]) and best
 * (int32) are DEVICE pointers - the caller reads them back when it needs the decision on the host.
 * yh_absmax: max |t| - what COSPTQuantizedFeatureConcat tracks per routed tensor (:1403-1449, max(max(t), |min(t)|)).        */
int64_t yh_ptq_search_workspace(int64_t count);
int yh_ptq_cos_search(const float* t, int64_t count, float scale0, int n, float lo, float hi, int do_clamp, void* ws,
                      int64_t ws_bytes, double* cos_out, int32_t* best, void* stream);
int yh_absmax(const float* t, int64_t count, void* ws, int64_t ws_bytes, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Plans: a recorded sequence of the launches above, replayed by one native call per forward (the
 * replacement for the per-layer Python dispatch loop of models.py:524-545).  Pointers that change
 * from call to call (network input, per-call outputs) are "slots": a fixup patches one pointer field
 * of one recorded op with slot_base + byte_offset right before launch.                                  */
typedef struct yh_plan yh_plan;
enum { YH_OP_CONV = 1, YH_OP_STEM = 2, YH_OP_POOL = 3, YH_OP_COPY = 4, YH_OP_ADD = 5, YH_OP_DECODE = 6, YH_OP_DW = 7,
       YH_OP_SE = 8, YH_OP_QCOPY = 9, YH_OP_QPOOL = 10, YH_OP_QADD = 11, YH_OP_BN_STATS = 12, YH_OP_BN_FINALIZE = 13,
       YH_OP_BN_ACT_FWD = 14, YH_OP_BN_BWD_REDUCE = 15, YH_OP_BN_BWD_APPLY = 16, YH_OP_WGRAD = 17, YH_OP_STEM_WGRAD = 18,
       YH_OP_DILATE2 = 19, YH_OP_UPSAMPLE2_BWD = 20, YH_OP_CAST_F32 = 21, YH_OP_NCHW_TO_NHWC = 22, YH_OP_POOL_BWD = 23, YH_OP_PACK_BATCH = 24, YH_OP_DW_WGRAD = 25, YH_OP_DW_DGRAD = 26,
       YH_OP_SE_BWD = 27, YH_OP_STEM_BWD = 28 };
typedef struct yh_pack_batch_desc { const yh_pack_item* items; int32_t n_items; } yh_pack_batch_desc;
typedef struct yh_layout_desc { const float* x; void* y; int32_t n, c, h, w_in, c_pad, ldy, dtype; } yh_layout_desc;

yh_plan* yh_plan_create(void);
void yh_plan_destroy(yh_plan* p);
int yh_plan_add(yh_plan* p, int op_kind, const void* desc, int desc_bytes); /* returns op index or <0 */
int yh_plan_add_fixup(yh_plan* p, int op_index, int field_offset, int slot, int64_t byte_offset);
int yh_plan_bind_slot(yh_plan* p, int slot, void* ptr);
/* Binding a slot to YH_SLOT_NULL makes its fixups write NULL into their fields: an optional output the caller does not want this
 * call (the raw head copies of yh_yolo_decode).  A slot that was never bound still fails the run with YH_EINVAL.                   */
#define YH_SLOT_NULL ((void*)(intptr_t)-1)
int yh_plan_num_ops(const yh_plan* p);
int yh_plan_run(yh_plan* p, void* stream);
/* run ops [first, last) only (profiling / per-layer tests) */
int yh_plan_run_range(yh_plan* p, int first, int last, void* stream);
/* Two lanes.  Ops are replayed in index order on the caller's stream (lane 0) unless marked lane 1, which runs on a stream owned
 * by the plan; across lanes the only ordering is what yh_plan_add_dep states (op waits for the completion of dep, dep < op), and
 * every yh_plan_run / yh_plan_run_range call joins the side lane into the caller's stream before it returns.  The training
 * backward puts the weight gradients on lane 1: they only feed the optimizer, so the HBM-bound BatchNorm backward passes of the
 * next layer run underneath them (engine/train.py).                                                                        */
int yh_plan_set_lane(yh_plan* p, int op_index, int lane);
int yh_plan_add_dep(yh_plan* p, int op_index, int dep_index);
/* Per-op HIP-event timing on the launch stream: when enabled every replay brackets each op with a pair of
 * events; after the caller has synchronised the stream, yh_plan_get_timings writes the last replay's
 * per-op durations in milliseconds to the HOST array ms_out[n], n == yh_plan_num_ops.                    */
int yh_plan_set_timing(yh_plan* p, int enable);
/* round 6: the reduce launches of the plan's weight gradients (yh_conv2d_wgrad: partial tiles -> dW) run on a stream of their own, in the
 * shadow of the following ops; every yh_plan_run / _run_range joins before it returns.  Only for plans whose weight-gradient ops have a
 * workspace no other op uses.  Off under yh_plan_set_timing (per-op times stay additive). */
int yh_plan_set_async_reduce(yh_plan* p, int enable);
int yh_plan_get_timings(yh_plan* p, float* ms_out, int n);
/* hipGraph replay for launch-bound (small batch) forwards: capture records one replay with the currently bound
 * slot pointers on a non-null stream; launch replays it with a single graph launch; reset drops the graph (needed
 * before slot pointers or recorded ops change).  Weights may be re-packed in place between launches.          */
int yh_plan_graph_capture(yh_plan* p, void* stream);
int yh_plan_graph_launch(yh_plan* p, void* stream);
int yh_plan_graph_reset(yh_plan* p);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H */
