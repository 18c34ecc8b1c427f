/*
 * vpho_hip.h -- C ABI of the MI355X (gfx950) kernels behind vpho_amd's `vpho_net.forward(mode='predict')`.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`; nothing is owned or freed by the library
 *     (outputs are caller-allocated), no torch types appear in any signature;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises unless stated;
 *   - return value: 0 on success, non-zero on error with a message retrievable by vpho_last_error();
 *   - activations are fp32 NHWC ("pixel-major, channel-minor"); row-major matrices otherwise;
 *   - thread-compatible (one caller per stream), not re-entrant on the same output buffers.
 *
 * Each entry point names the reference call site (file:line under zhoujun-7/VPHO) whose arithmetic it replaces.
 */
#ifndef VPHO_HIP_H
#define VPHO_HIP_H

/* The library is built with -fvisibility=hidden: the entry points declared here are its ONLY exported symbols
 * (tests/test_abi.py compares `nm -D` with this header). */
#if defined(__GNUC__) || defined(__clang__)
#define VPHO_API __attribute__((visibility("default")))
#else
#define VPHO_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

VPHO_API const char* vpho_last_error(void);
VPHO_API int vpho_abi_version(void);   /* 12 */

/* Opt-in timing of one kernel class with HIP events recorded on the launch stream around every launch
 * (0 = conv_igemm 128x128 tile, 1 = conv_igemm 64x64 tile, 2 = fused score head, 3 = conv_igemm 128x64 tile; HBM-bound kernels,
 * reported in bytes: 4 = MANO FK, 5 = object physics score, 6 = hand cascade fuse, 7 = RoIAlign, 8 = bilinear resize; 9 = Winograd
 * convolution, 10 / 11 = weight-gradient TN GEMM with 64x64 / 128x128 tiles, 12 = pseudo-force optimiser, 13 = pose encoder).
 * vpho_prof_collect waits for the recorded events and returns the summed kernel time, the launch count, the algorithmic
 * flop (2*M*N*K) and the algorithmic bytes (operands once) issued. */
VPHO_API int vpho_prof_enable(int kernel_class, int on);
VPHO_API int vpho_prof_collect(int kernel_class, double* total_ms, long long* launches, double* total_flops, double* total_bytes);

/* ------------------------------------------------------------------------------------------------------------------
 * Convolution / linear layers as one implicit-GEMM kernel on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces every nn.Conv2d + BatchNorm(eval) + LeakyReLU/ReLU (+ residual add) group of
 *   lib/model/backbone_FPN_HFL.py:70-109,326-350, head_inplane.py:102-107, encoding.py:21-36,58-73,
 *   cross_module.py:126-127 and every nn.Linear of head_mano.py:61-70, physics.py:700-712, cross_module.py:104-134,
 *   denoiser.py:68-76 (a Linear is the 1x1 case with H=W=1).
 * y[n,oy,ox,co] = act( sum_{r,s,c} pre(x[n, oy*stride-pad_y+r, ox*stride-pad_x+s, c]) * w[co][(r*KW+s)*Cin+c]
 *                      + bias[co] + res[n,oy,ox,co] ),  act(v) = v>0 ? v : v*out_slope
 * pre(v) = lrelu(v*in_scale[c]+in_shift[c], in_slope) when in_scale != NULL (pre-activation BN of encoding.Residual),
 * zero padding is applied after pre().  BatchNorm following a conv is folded into w/bias by the caller.
 * Requirements: Cin % 4 == 0, x_ld % 4 == 0 (x_ld = floats between consecutive pixels of x), 16-byte aligned x and w.
 * The same kernel is the GEMM of the training slices (H=W=1: rows x Cin times Cout x Cin transposed).
 */
typedef struct {
    const float* x; const float* w; const float* bias;
    const float* in_scale; const float* in_shift;
    const float* res; float* y;
    int N, H, W, Cin, x_ld;
    int Cout, KH, KW, stride, pad_y, pad_x, OH, OW;
    long long y_sn, y_sy, y_sx;      /* element strides of y for (n, oy, ox); channel stride is 1 */
    long long r_sn, r_sy, r_sx;      /* same for res */
    float in_slope, out_slope;
    /* Optional (zero = off).  w_ld: floats between consecutive rows of w when a launch reduces only a K-slice of wider rows.
     * splits > 1: blockIdx.y = 0..splits-1 runs the same problem on x + i*x_split, w + i*w_split and writes y + i*y_split --
     * partial sums of a long reduction (the pixel dimension of a weight gradient) for the caller to add; no bias / res. */
    int w_ld, splits;
    long long x_split, w_split, y_split;
    /* Optional (NULL = off; ABI version 4): y = gate > 0 ? y : y * gate_slope after the epilogue, `gate` laid out like y (same
     * strides, pointing at the element that corresponds to y[0]) -- the backward of a LeakyReLU given its output, fused into the
     * input-gradient convolution (torch autograd of nn.LeakyReLU in Bottleneck / Residual, backbone_FPN_HFL.py:326, encoding.py:21-36). */
    const float* gate;
    float gate_slope;
    /* Optional (NULL = off; ABI version 5): compute only the output pixels listed in row_map[0 .. *row_count) -- each entry the
     * linear index (n*OH + oy)*OW + ox of an output pixel -- and write them as the rows of a COMPACT (*row_count, Cout) matrix
     * (y_sx = floats between rows; y_sn / y_sy unused; res, if given, is compact too).  Both live on the device: the grid is
     * sized for all N*OH*OW pixels and tiles beyond *row_count exit, so the launch needs no host round trip and replays in a
     * HIP graph with new windows.  Used for the FPN smoothing convolutions, whose maps are read only through RoIAlign
     * (VPHO.py:126-129): see vpho_roi_windows_i32.  rows_hint (0 = unknown) only feeds the profiling counters. */
    const int* row_map;
    const int* row_count;
    int rows_hint;
    int rows_scatter;                /* 1: keep y's (n, oy, ox) layout and write only the listed pixels (the rest of y is untouched) */
    /* Optional, opt-in (NULL / 0 = fp32 MFMA, the default and the path parity is stated on): w as three bf16 planes [3][Cout][K]
     * with w = plane0 + plane1 + plane2 exactly, plane_terms = 6 or 9 bf16 products per fp32 product, fp32 accumulation.  Taken for
     * Cin % 16 == 0, no prologue affine, no splits / gate, 16-byte addressable output rows; other shapes run the fp32 kernels. */
    const void* w_planes;
    int plane_terms;
    /* Optional (NULL = off; ABI version 8): the residual is a COARSER map, added after bilinear up-sampling to the output grid --
     * y = act(conv + bias + F.interpolate(res_up, size=(OH, OW), mode='bilinear', align_corners=False)[n, oy, ox, co]) -- the FPN's
     * top-down step `_upsample_add(p, lateral(c))` (backbone_FPN_HFL.py:66-68,98-104) inside the lateral 1x1 convolution's epilogue:
     * the finer map is written once instead of written, re-read and re-written by a separate pass.  res_up: (N, ru_H, ru_W, ru_ld)
     * NHWC, ru_ld % 4 == 0, 16-byte aligned; exclusive with `res`; the same arithmetic, in the same order, as
     * vpho_resize_bilinear_nhwc_f32(accumulate = 1) after the convolution: bit-identical.  Works with row_map / rows_scatter. */
    const float* res_up;
    int ru_H, ru_W, ru_ld;
    /* Optional (NULL = off; ABI version 8): a SECOND input of a 1x1 convolution, concatenated behind x along the channel axis without a
     * copy: y = act(w[:, :Cin] . x[n, oy*stride, ox*stride, :] + w[:, Cin:] . x2[n, oy*stride2, ox*stride2, :] + bias (+ res)), w packed as
     * [Cout][Cin + Cin2].  It merges the projection shortcut of a ResNet stage's first bottleneck (downsample.0/1: a 1x1 convolution +
     * BatchNorm of the block input, stride 1 or 2, backbone_FPN_HFL.py:311-350) into conv3: one launch, and the 4C-wide shortcut map is
     * neither written nor re-read.  x2: (N, H2, W2, x2_ld) NHWC; needs KH = KW = 1, no padding, Cin % 32 == 0, Cin2 % 32 == 0, no pixel
     * list / prologue / splits.  The sum over the Cin + Cin2 products is one accumulation chain: results differ from the two-launch
     * form (sum, bias, add) by fp32 rounding. */
    const float* x2;
    int Cin2, x2_ld, stride2, H2, W2;
    /* Optional (0 / 1 = off; ABI version 11): GROUPED launch -- group g = 0 .. groups-1 runs the same problem on x + g*x_group, w + g*w_group,
     * bias + g*bias_group, y + g*y_group, res + g*res_group, x2 + g*x2_group, in_scale / in_shift + g*pre_group, res_up + g*ru_group (floats;
     * multiples of 4; x_group = 0: every group reads the same input).  The hand and the object branch of the feature path are twins
     * -- layer2 / layer3, FPN laterals, heat-map heads, encoders: same shapes, different weights (backbone_FPN_HFL.py:79-109, VPHO.py:131-149)
     * -- and run as ONE launch each: twice the tiles, half the launches; every output element's k order is unchanged (bit-identical to
     * the two single launches).  No splits / pixel list / gate / bf16 planes. */
    int groups;
    long long x_group, w_group, bias_group, y_group, res_group, x2_group, pre_group, ru_group;
    /* Optional (NULL = off; ABI version 12): the reductions of a train-mode BatchNorm next to this convolution, taken in its epilogue
     * while the output tile is still on the chip (nn.BatchNorm2d under model.train() behind every convolution of Bottleneck / Residual /
     * HeadHeatmap2: backbone_FPN_HFL.py:330-350, encoding.py:21-36, head_inplane.py:40-58; train_diff_hand_obj.py:171,181).
     *   stats      device, [stats_cap][2][Cout] floats: row t = the partial sums over the output rows of M-tile t of the values this launch
     *              STORES (after bias / residual / activation / gate): plane 0 = sum v, plane 1 = sum v * v -- the statistics pass of
     *              the BatchNorm that FOLLOWS the convolution (vpho_bn_train_forward_stats_f32 finishes them in fp64).  Plain stores, one
     *              writer per element, fixed order: bit-reproducible.
     *   stats_rows HOST pointer, written before the call returns: the number of partial rows the launch writes (= its M-tiles), or 0 when
     *              the kernel chosen for this shape does not produce them (the caller then runs the stand-alone reduction).  A launch that
     *              needs more than stats_cap rows writes none and reports 0.
     *   bn_x != NULL (input-gradient convolutions): the launch's output is the gradient at the OUTPUT of act(BatchNorm(bn_x)), bn_x laid
     *              out like y.  The epilogue recomputes xhat = (bn_x - bn_mean) * bn_invstd and the activation's sign from
     *              xhat * bn_gamma + bn_beta (the expression of the forward pass: the same bits), applies the activation's backward
     *              (y = t > 0 ? y : y * gate_slope, as `gate` would with the stored activation), and plane 1 becomes sum y * xhat:
     *              the two sums of the BatchNorm backward (d beta, d gamma; vpho_bn_train_backward_stats_f32).  With `gate` set as well
     *              the sign comes from the stored activation as usual and only xhat is recomputed (bn_gamma / bn_beta unused): the
     *              closing activation of a residual block, lrelu(BatchNorm(x) + shortcut), whose sign bn_x alone does not determine.
     * Served by the direct-to-LDS kernels' 16-byte epilogue: no splits / pixel list / groups / bf16 planes; with bn_x the call fails on
     * other shapes instead of dropping the gate. */
    float* stats;
    int* stats_rows;
    int stats_cap;
    const float* bn_x; const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;
} vpho_conv_desc;
/* Limits: Cin, x_ld multiples of 4, 16-byte aligned x / w; x and w (all splits included) below 3.9 GB each (32-bit buffer offsets). */
VPHO_API int vpho_conv2d_nhwc_f32(const vpho_conv_desc* d, void* stream);
/* Winograd F(2x2, 3x3) form of a 3x3 / stride 1 / padding 1 convolution (+ bias, LeakyReLU): u = G g G^T, fp32, stage-tiled as
 * (Cin/8, 16 frequencies, Cout, 8 input channels) by model/pack.py::winograd_weights; 2.25 x fewer multiply-adds on the matrix cores, input / output transforms inside the kernel
 * (csrc/conv_winograd.hip).  Needs even H, W, Cin % 16 == 0, Cout % 64 == 0.  Results differ from vpho_conv2d_nhwc_f32 by fp32 rounding
 * of the transforms (~1e-6 relative). */
VPHO_API int vpho_conv3x3_winograd_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout,
                                   float out_slope, float* y, int y_ld, void* stream);
/* The same convolution on RoI windows only (the FPN smoothing convolutions, backbone_FPN_HFL.py:105-108, whose maps are read only
 * through RoIAlign, VPHO.py:126-129): wins = the (N,5) window table of vpho_roi_windows_i32, tile_base (N+1 ints, written by
 * vpho_winograd_window_tiles_i32) = first tile of every image on the list of 2 x 2 output tiles -- on each image's EVEN pixel grid,
 * so every pixel comes from the same 4 x 4 patch as in the full-map launch: bit-identical -- that touch its window.  y_rows is the
 * COMPACT (rows, Cout) matrix vpho_roi_align_window_nhwc_f32 reads; x is the ordinary (N,H,W,x_ld) map, of which only the windows
 * dilated by one pixel need to hold data.  Device-side lists: the grid is sized for all tiles, blocks past the last live tile exit
 * (no host round trip, replays in a HIP graph with new boxes).  tiles_hint (0 = unknown) only feeds the profiling counters. */
/* GROUPED form (ABI version 11): `groups` independent convolutions of the same shape in one launch (the twin hand / object branches):
 * group g reads x + g*x_group (floats; 0 = every group reads the same input), u + g*16*Cout*Cin, bias + g*Cout and writes images
 * [g*N, (g+1)*N) of y (groups*N, H, W, y_ld).  Bit-identical to `groups` single launches. */
VPHO_API int vpho_conv3x3_winograd_grouped_nhwc_f32(const float* x, long long x_group, const float* u, const float* bias, int groups, int N, int H, int W,
                                           int Cin, int x_ld, int Cout, float out_slope, float* y, int y_ld, void* stream);
VPHO_API int vpho_winograd_window_tiles_i32(const int* wins, int N, int* tile_base, void* stream);
/* ... and with the window pixels written IN PLACE into the ordinary (N,H,W,y_ld) map y, every other pixel of y left untouched (the
 * input gradient of an FPN smoothing convolution: non-zero only in the RoI windows dilated by the 3x3 halo; the caller zeroes y) */
VPHO_API int vpho_conv3x3_winograd_scatter_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout,
                                           float out_slope, const int* wins, const int* tile_base, float* y, int y_ld, void* stream);
/* Training (weights change every step): u on the DEVICE from the packed 3x3 weights (Cout, 9*Cin) -- for the forward convolution
 * (for_input_gradient = 0: u is (Cin/8, 16, Cout, 8)) or for its input-gradient convolution, the 3x3 convolution of dY with the
 * spatially flipped, channel-transposed weights (1: u is (Cout/8, 16, Cin, 8)); fp64 arithmetic, rounded once.  The gate variant
 * of the convolution fuses the backward of the LeakyReLU that produced the convolution's input (torch autograd of nn.LeakyReLU in
 * Bottleneck / Residual, backbone_FPN_HFL.py:326, encoding.py:21-36): y = gate > 0 ? y : gate_slope * y, gate laid out like y. */
VPHO_API int vpho_winograd_weights_f32(const float* w_packed, int Cout, int Cin, int for_input_gradient, float* u, void* stream);
/* the same for a list of weight tensors in one launch (ABI version 9).  segments: device array of n_segments records
 * { const float* w_packed; float* u; int32 Cout, Cin, for_input_gradient, first_block; } (32 bytes) with first_block = running sum of
 * ceil(Cout * Cin / 256) over the preceding records, total_blocks = that sum over all records; every record as for the single call
 * (the convolution's input channels a multiple of 16) -- not checked on the device. */
VPHO_API int vpho_winograd_weights_multi_f32(const void* segments, int n_segments, long long total_blocks, void* stream);
VPHO_API int vpho_conv3x3_winograd_gate_nhwc_f32(const float* x, const float* u, const float* gate, float gate_slope, int N, int H, int W, int Cin,
                                        int x_ld, int Cout, float* y, int y_ld, void* stream);
/* Full-map Winograd convolution with the BatchNorm reductions of vpho_conv_desc.stats in its epilogue (ABI version 12): stats =
 * [ceil(N*H*W/256)][2][Cout] floats, row t = the sums over the 256 output pixels of tile block t (64 tiles of 2 x 2); returns the number of
 * rows through *stats_rows (host).  bn_x == NULL: the forward convolution (bias, out_slope as above; sum v | sum v^2 of the stored values).
 * bn_x != NULL: the input-gradient convolution of the gate variant with the gate recomputed from bn_x (laid out like y, leading dimension
 * y_ld) and the BatchNorm vectors -- y = (xhat * gamma + beta > 0) ? y : gate_slope * y, sums y | y * xhat -- see vpho_conv_desc. */
VPHO_API int vpho_conv3x3_winograd_stats_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout,
                                         float out_slope, float* y, int y_ld, float* stats, int stats_cap, int* stats_rows, const float* bn_x,
                                         const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                         float gate_slope, void* stream);
VPHO_API int vpho_conv3x3_winograd_rows_nhwc_f32(const float* x, const float* u, const float* bias, int N, int H, int W, int Cin, int x_ld, int Cout,
                                        float out_slope, const int* wins, const int* tile_base, int tiles_hint, float* y_rows, int y_ld,
                                        void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Score network (GenPose-style conditional denoiser) and the probability-flow ODE sampler.
 * Replaces lib/model/denoiser.py:68-82 (BaseDenoiser.forward with ManoPoseHead2 :176-179 / ObjHead2 :244-247 and
 * parallel_linear.py:27-35) and lib/model/score_based_model.py:45-105 (cond_ode_sampler: scipy solve_ivp RK45,
 * rtol/atol controller, dense output on t_eval, final reverse-diffusion denoise step) with sde.py:15-28 (VE SDE).
 *
 * Weight layout (packed once by the caller from the reference state_dict):
 *   t_W[64]                     t_encoder.0.W
 *   t_w[128][128], t_b[128]     t_encoder.1.{weight,bias}            (row-major [out][in])
 *   pe0_w[256][Dp], pe0_b[256]  pose_encoder.0 (input dim D zero-padded to Dp, Dp % 4 == 0)
 *   pe2_w[256][256], pe2_b[256] pose_encoder.2
 *   w1_t[128][NH]               head.head.0.weight[n, 0:128,   j] at [k][n*256+j]         NH = nheads*256
 *   w1_p[NH][256]               head.head.0.weight[n, 128:384, j] at [n*256+j][k]
 *   w1_f[NH][1024], b1[NH]      head.head.0.weight[n, 384:1408, j] at [n*256+j][k]; head.head.0.bias
 *   w2[NH][4]                   head.head.2.weight[n, j, 0:3] padded to 4;  b2[nheads*3] head.head.2.bias
 * The 1408-wide first layer is evaluated as  feat-part (once per image, `cimg`) + t-part (once per evaluation)
 * + pose-part (per row, fp32 MFMA) -- algebraically the same sum (SURVEY.md 7 "algebraic opportunity").
 */
typedef struct {
    int D, Dp, nheads;
    const float *t_W, *t_w, *t_b, *pe0_w, *pe0_b, *pe2_w, *pe2_b, *w1_t, *w1_p, *w1_f, *b1, *w2, *b2;
    /* Optional, opt-in (NULL / 0 = the fp32-MFMA score head, the default and the path parity is stated on): w1_p as three bf16
     * planes [nheads][3][256][256] with w1_p = plane0 + plane1 + plane2 exactly (plane0 = bf16(w), plane1 = bf16(w - plane0), ...),
     * split_terms = 6 or 9 cross products per fp32 product on the bf16 matrix cores, fp32 accumulation (csrc/score_ode.hip). */
    const void* w1_p_split;
    int split_terms;
} vpho_score_weights;

/* bytes of scratch `vpho_score_eval` / `vpho_ode_sample` need for R = bs*S rows */
VPHO_API long long vpho_score_workspace_bytes(const vpho_score_weights* w, int bs, int S);

/* One score evaluation s(x, t | feat) for R = bs*S rows (row r belongs to image r / S).
 * feat_img: [bs][1024], x: [R][D] fp32, t: scalar shared by all rows, out: [R][D] fp32.  (denoiser.py:68-82) */
VPHO_API int vpho_score_eval(const vpho_score_weights* w, const float* feat_img, int bs, int S, const float* x, float t,
                    float* out, void* workspace, long long workspace_bytes, void* stream);

typedef struct {
    int nfev;            /* score-network evaluations incl. the final denoise call */
    int n_accepted, n_rejected;
    int nan_count;       /* NaN score entries replaced by 0 (score_based_model.py:69-71) */
    int status;          /* 0 ok, 1 step size underflow */
    int n_log;           /* entries written to step_log_host: (t, h, error_norm, accepted) per attempted step */
} vpho_ode_stats;

/* Full cond_ode_sampler run.  init_x: [R][D] fp32 prior draw (already scaled by sigma(T0)).
 * xs_out: [R][num_steps][D] dense output at t_eval = linspace(T0, eps, num_steps), fp64 if xs_is_f64 else fp32
 * (the reference casts the hand trajectory to fp32 right after sampling, VPHO.py:243); x_out: [R][D] final sample
 * after the denoise step, fp64 if x_is_f64 else rounded to fp32 the same way.  scipy's scalar step controller
 * (select_initial_step, accept/reject, step-size factor, t_eval stamps) runs in one-thread kernels on a control block in the
 * workspace; the call enqueues the expected number of attempts plus the denoise step and waits for `stream` ONCE (more
 * attempts are added two at a time if the solve is not finished).  With the environment variable VPHO_RK_HOST=1 (or
 * num_steps > 1024) the controller runs on the host instead and the call synchronises once per attempted step.
 * stats_host / step_log_host are host memory (step_log_host may be NULL; capacity in entries of 4 doubles). */
VPHO_API int vpho_ode_sample(const vpho_score_weights* w, const float* feat_img, int bs, int S, const float* init_x,
                    double T0, double eps, int num_steps, double rtol, double atol,
                    void* xs_out, int xs_is_f64, void* x_out, int x_is_f64,
                    void* workspace, long long workspace_bytes,
                    vpho_ode_stats* stats_host, double* step_log_host, int step_log_cap, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HBM-bound glue of the feature path (NHWC fp32).  c_off / ld* let a kernel write straight into a channel slice of a
 * wider (concatenation) buffer.
 */
/* rgb (N,C,H,W) -> (N,H,W,ldy) with channels >= C zero-filled (input of the 7x7 stem, backbone_FPN_HFL.py:206) */
VPHO_API int vpho_nchw_to_nhwc_f32(const float* x, int N, int C, int H, int W, float* y, int ldy, void* stream);
/* (N,H,W,C | ldx) -> (N,C,H,W): heat-map outputs (VPHO.py:232-233) and Encoder's x.flatten(1) (encoding.py:72) */
VPHO_API int vpho_nhwc_to_nchw_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, void* stream);
/* nn.MaxPool2d (backbone_FPN_HFL.py:209 k3 s2 p1; encoding.py:54 k2 s2) */
VPHO_API int vpho_maxpool_nhwc_f32(const float* x, int N, int H, int W, int C, int k, int stride, int pad, float* y, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False); accumulate=1 gives FPN._upsample_add (backbone_FPN_HFL.py:66-68),
 * accumulate=0 with c_off the heat-map 64->32 resize into the encoder input (VPHO.py:143-144,148-149) */
VPHO_API int vpho_resize_bilinear_nhwc_f32(const float* x, int N, int H, int W, int C, int ldx, int OH, int OW,
                                  float* y, int ldy, int c_off, int accumulate, void* stream);
/* torchvision.ops.roi_align(aligned=False, sampling_ratio=-1), one box per image, boxes (N,4) xyxy (VPHO.py:125-128);
 * flip_w[n] != 0 mirrors the output along W (flip_tensor_by_mask_index, VPHO.py:138) */
VPHO_API int vpho_roi_align_nhwc_f32(const float* feat, int N, int H, int W, int C, const float* boxes, float spatial_scale,
                            int out_size, const unsigned char* flip_w, float* out, int ldo, int c_off, void* stream);
/* Demand-driven FPN output.  The stride-4 maps of FPN.forward (backbone_FPN_HFL.py:105-109) are read ONLY by the RoIAligns of
 * VPHO.py:126-129, so each branch's last convolution is computed on the pixels its image's boxes can sample and nowhere else --
 * the same values, about half the pixels at the README config's box sizes (dexycb6.py:346-356: boxes = 1.15 / 1.10 x the tight
 * key-point boxes).  vpho_roi_windows_i32: per image the window = union over boxes_a[n], boxes_b[n] (NULL = one box) of the rows /
 * columns a bilinear sample of torchvision's roi_align can weigh; wins (N,5) = (first compact row, y0, x0, w, h), row_map = linear
 * pixel index (n*H + y)*W + x of every window pixel in compact order, *row_count their number; dilate > 0 widens every window by
 * that many pixels (the 3x3 halo: the lateral 1x1 convolution and the top-down add that feed the last convolution run on the
 * windows dilated by 1, stored in place -- vpho_conv_desc.rows_scatter).  All device-side: feed row_map /
 * row_count to vpho_conv_desc and wins to vpho_roi_align_window_nhwc_f32, which reads the compact (rows, C) matrix with the
 * arithmetic of vpho_roi_align_nhwc_f32 (bit-identical outputs, tests/test_gpu_glue.py). */
VPHO_API int vpho_roi_windows_i32(const float* boxes_a, const float* boxes_b, int N, int H, int W, float spatial_scale, int dilate,
                         int* wins, int* row_map, int* row_count, void* stream);
/* vpho_resize_bilinear_nhwc_f32 on the listed output pixels only (FPN._upsample_add inside the dilated windows) */
VPHO_API int vpho_resize_bilinear_rows_nhwc_f32(const float* x, int N, int H, int W, int C, int ldx, int OH, int OW, float* y, int ldy, int c_off,
                                       int accumulate, const int* row_map, const int* row_count, int rows_hint, void* stream);
VPHO_API int vpho_roi_align_window_nhwc_f32(const float* feat_rows, const int* wins, int N, int H, int W, int C, const float* boxes,
                                   float spatial_scale, int out_size, const unsigned char* flip_w, float* out, int ldo, int c_off, int rows_hint,
                                   void* stream);
/* the same pooling pass with TWO destinations (each with its own optional W-flip flags): the object branch pools bbox_obj_rect twice,
 * plain for the heat-map head and flipped for left hands into the encoder input (VPHO.py:126-138) -- one read of the map, values
 * bit-identical to two vpho_roi_align_window_nhwc_f32 calls */
VPHO_API int vpho_roi_align_window_dual_nhwc_f32(const float* feat_rows, const int* wins, int N, int H, int W, int C, const float* boxes,
                                        float spatial_scale, int out_size, const unsigned char* flip_w, float* out, int ldo, int c_off,
                                        const unsigned char* flip_w2, float* out2, int ldo2, int c_off2, int rows_hint, void* stream);
/* align_hm_to_bbox_rectangle (VPHO.py:333-346, transposing, quirk Q2) (+ optional W flip, VPHO.py:139) */
VPHO_API int vpho_align_heatmap_nhwc_f32(const float* hm, int N, int size, int C, const float* bbox, const float* bbox_rect,
                                const unsigned char* flip_w, float* out, void* stream);
/* NeRF embedding of gravity (cross_module.py:8-46), x negated where flip_x (VPHO.py:167); out (N,64), column 63 = 0 */
/* y = act(x . w^T + bias) of a SMALL linear layer with the products and the sum in double, rounded to fp32 once (ABI version 11): the
 * regression head head_mano.py:61-70, whose output is normalised into half of the cascade's candidates (aggregation.py:120-126) -- see
 * csrc/misc.hip.  x (rows, ld_x >= cin), w (cout, cin) packed, y (rows, ld_y >= cout); cin <= 4096. */
VPHO_API int vpho_linear_acc64_f32(const float* x, int rows, int cin, int ld_x, const float* w, const float* bias, int cout, float out_slope,
                          float* y, int ld_y, void* stream);
VPHO_API int vpho_nerf_embed_f32(const float* g, int N, const unsigned char* flip_x, float* out, void* stream);
/* (bs,65,512) token tensor of CrossModule.forward (cross_module.py:124-133) incl. the positional code of the BATCH index */
VPHO_API int vpho_cross_tokens_f32(const float* proj_hand, const float* proj_obj, const float* grav_emb, const float* pe,
                          int bs, float* out, void* stream);
/* multi-head attention core over the first axis of qkv (S,B,3E) (nn.MultiheadAttention inside cross_module.py:104-107) */
VPHO_API int vpho_mha_f32(const float* qkv, int S, int B, int E, int nhead, float* out, void* stream);
/* the same with nn.MultiheadAttention's dropout on the attention probabilities (training): drop_mask [B*nhead][S][S] = keep / (1 - p),
 * applied after the soft-max and before P V; NULL = no dropout */
VPHO_API int vpho_mha_dropout_f32(const float* qkv, int S, int B, int E, int nhead, const float* drop_mask, float* out, void* stream);
/* out = LayerNorm(x + r) (post-norm TransformerEncoderLayer) */
VPHO_API int vpho_add_layernorm_f32(const float* x, const float* r, const float* gamma, const float* beta, long long rows, int E,
                           float eps, float* out, void* stream);
/* HeadPhysics: |scale| * normalise(softmax(softmax(logits)) . friction-cone anchors) (physics.py:546-557,700-712).
 * Output row r reads token row (r / group) * group_stride + r % group (+ off_scale | off_logits) of the MLP outputs, so the
 * 32 hand / 32 object tokens are picked out of the (bs, 65, .) transformer output without a copy. */
VPHO_API int vpho_force_local_f32(const float* scale, int ld_scale, const float* logits, int ld_logits, const float* anchor,
                         float friction, long long rows, int group, int group_stride, int off_scale, int off_logits,
                         float* out, void* stream);
/* matrix_to_axis_angle(rotation_6d_to_matrix(x)) for rows of rot_per_row rotations (head_mano.py:66-69, VPHO.py:314-324) */
VPHO_API int vpho_rot6d_to_axis_angle_f32(const float* x, long long rows, int rot_per_row, int ldx, float* out, int ldo, void* stream);
/* out[row, 48:58] = betas[row / rows_per_image] (VPHO.py:318-319,325-326) */
VPHO_API int vpho_append_betas_f32(const float* betas, long long rows, long long rows_per_image, float* out, int ldo, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * MANO forward kinematics (manopth.ManoLayer as configured at head_mano.py:48-55; metres, head_mano.py:78-87).
 * posedirs_t is the pose-blend table transposed to [135][778*3] (coalesced over vertices).
 */
typedef struct {
    const float *v_template;   /* [778][3]      */
    const float *shapedirs;    /* [778][3][10]  */
    const float *posedirs_t;   /* [135][778*3]  */
    const float *J_regressor;  /* [16][778]     */
    const float *weights;      /* [778][16]     */
    /* ABI version 6, optional (NULL: read the columns from posedirs_t): the 30 pose-blend columns of the 10 finger-tip vertices
     * (5 manopth tips 745, 317, 444, 556, 673, then 5 HO3D tips 728, 353, 442, 576, 694; column = tip * 3 + c) gathered into one
     * contiguous [135][30] table -- the joints-only launches of the heat-map cascade read nothing else of the 1.26 MB table */
    const float *tip_posedirs_t;
    /* ABI version 7, optional (NULL: big launches with vertices take the packed-FMA kernel): the pose-blend table in the operand order
     * of the fp32 matrix-core kernel, zero padded to 136 rows and 800 vertices: [25 vertex tiles][17 groups of 4 k steps][3][2][32][4],
     * element (t, g, j, lh, li, e) = posedirs_t[k][(32 t + li) * 3 + c] with f = 4 j + e, k = 2 (4 g + f / 3) + lh, c = f % 3 --
     * a lane's B fragments of four k steps are three aligned 16-byte loads, consecutive lanes read consecutive 16 bytes */
    const float *posedirs_mfma;
} vpho_mano_tables;
/* per image: v_shaped (n_img,778,3), J (n_img,16,3) from betas (n_img,10) */
VPHO_API int vpho_mano_shape_f32(const vpho_mano_tables* t, const float* betas, int n_img, float* v_shaped, float* J, void* stream);
/* per hand: pose rows of ld_pose floats (first 48 = axis-angle); hand h uses image h / hands_per_image.
 * verts may be NULL (joints only).  ho3d_per_image (optional) selects hand_fn.get_joint_aligned_with_HO3D ordering. */
VPHO_API int vpho_mano_fk_f32(const vpho_mano_tables* t, const float* pose, int ld_pose, long long n_hands, int hands_per_image,
                     const float* v_shaped, const float* J, const unsigned char* ho3d_per_image,
                     float* verts, float* joints, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Aggregation (lib/model/aggregation.py HOI_Aggregator.__call__ :1167-1353 and callees; lib/utils/transform_fn.py:101-125;
 * lib/utils/physics_fn.py:224-257).  Top-k order: value descending, ties by ascending candidate index.
 */
typedef struct { const float* kpt; const float* vert; const float* com; int n_kpt, n_vert, n_obj; } vpho_obj_tables;
    /* kpt [n_obj][n_kpt][3], vert [n_obj][n_vert][3], com [n_obj][3]   (head_object.py:13-33) */
typedef struct { const int* face_idx; const float* anchor_weight; const float* vert2joint; const int* skeleton; } vpho_anchor_tables;
    /* face_idx [32][3], anchor_weight [32][2], vert2joint [21][778], skeleton [32][2]   (physics_fn.py:120-171) */

/* pose (bs,2S,48): [S diffusion | S regression with the diffusion wrist] (aggregation.py:120-126,140-143) */
VPHO_API int vpho_hand_candidates_f32(const float* diff_pose, int ld_diff, const float* reg_pose, int bs, int S, float* pose, void* stream);
/* hv (bs,C,n_obs): bicubic heat-map value of each observed joint of each candidate (aggregation.py:196-213);
 * joints (bs,C,21,3) root-relative, heatmap (bs,J,H,W) planar; observe_host is a HOST int array */
VPHO_API int vpho_hand_heat_f32(const float* joints, const float* root, const float* Kmat, const float* bbox, const float* heatmap,
                       int bs, int C, int J, int H, int W, const int* observe_host, int n_obs, float* out, void* stream);
/* one cascade level (aggregation.py:215-269): score -> top-k -> weighted quaternion mean -> broadcast into all candidates.
 * val/idx: [bs][F][k] with F = 1 (level 0) or 5, in torch.topk's order (larger score first; equal scores: smaller index first);
 * topk_pose (optional) [bs][k][F][3]; score_out (optional) [bs][C][F] = the level score of EVERY candidate exactly as ranked
 * (aggregation.py:215-218,244-247: sum over the observed joints at level 0, per-finger mean at levels 1-3) */
VPHO_API int vpho_hand_fuse_level_f32(const float* hv, int n_obs, float* pose, int bs, int C, int k, int level,
                             float* val, int* idx, float* topk_pose, float* score_out, void* stream);
/* generic wavefront top-k: element c of row (o,f) at scores[(o*n + c)*F + f]; val/idx [o][f][k] */
VPHO_API int vpho_topk_f32(const float* scores, int rows_outer, int n, int F, int k, float* val, int* idx, void* stream);
VPHO_API int vpho_topk_weights_f32(const float* val, int rows, int k, float* w, void* stream);
/* select_topk_object_by_heatmap score (aggregation.py:742-776); pose (bs,n,9) fp64 */
VPHO_API int vpho_obj_heat_score(const double* pose, int n, const double* transl_override, const float* root, const vpho_obj_tables* t,
                        const int* obj_id, const unsigned char* is_right, const float* Kmat, const float* bbox,
                        const float* heatmap, int bs, int H, int W, float* score, void* stream);
VPHO_API int vpho_obj_cross_candidates(const double* pose, int n, const int* transl_idx, const int* rot_idx, int bs, int ko, double* cand, void* stream);
/* select_topk_object_by_physics3 score (aggregation.py:947-985) */
VPHO_API int vpho_obj_physics_score(const double* cand, int n, const float* root, const vpho_obj_tables* t, const int* obj_id,
                           const unsigned char* is_right, const float* force_point, const float* force_global, int bs,
                           float* score, void* stream);
/* fuse_topk + average_rot6d in fp64 (aggregation.py:729-740,50-56); source b (if given) is used where pick_b[b] != 0 */
VPHO_API int vpho_obj_fuse_f64(const double* pose, int n, const int* idx_a, const float* w_a, const int* idx_b, const float* w_b,
                      const unsigned char* pick_b, int bs, int k, double* fused, void* stream);
VPHO_API int vpho_obj_verts_f32(const double* pose, const float* root, const vpho_obj_tables* t, const int* obj_id,
                       const unsigned char* is_right, int bs, float* out, void* stream);
/* ForceAnchor.__call__ on (verts + root) and from_local_to_global (physics_fn.py:224-257, physics.py:362-371) */
VPHO_API int vpho_force_anchor_f32(const vpho_anchor_tables* t, const float* verts, const float* root, const float* force_local,
                          long long n_hands, int hands_per_image, float* force_point, float* force_global, void* stream);
/* aggregation.py:1306-1325 / :561-596 / :598-617 */
VPHO_API int vpho_hand_phys_candidates_f32(const float* agg_pose, int ld_agg, const float* betas, const float* topk_pose, int bs, int k,
                                  float* out, void* stream);
VPHO_API int vpho_hand_phys_score_f32(const float* force_point, const float* force_global, const float* obj_vert, int n_vert,
                             int bs, int n_cand, float* finger_score, void* stream);
VPHO_API int vpho_hand_phys_fuse_f32(const float* cand, int n_cand, const int* idx, int bs, int k, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Evaluation metrics on the device: TesterHand.criterion_MJE_PAMJE (lib/engine/test.py:657-680) with
 * rigid_align_AtoB (lib/utils/transform_fn.py:43-66).  pd, gt: [n_img][n_pts][3] fp32 (metres); outputs per image the
 * mean point error, the mean error after similarity (Procrustes) alignment of pd onto gt, optionally every point's error. */
VPHO_API int vpho_hand_metrics_f32(const float* pd, const float* gt, int n_img, int n_pts, float* mean_err, float* pa_mean_err,
                          float* per_point, void* stream);

/* Object evaluation metrics on the device: TesterObject.__call__ (lib/engine/test.py:240-352) for a single
 * hypothesis per image -- criterion_MCE_OCE (:354-374), criterion_MCE2 / compute_obj_metrics_dexycb (:398-414,155-193),
 * criterion_ADD_REP (:425-458), cal_ADD01d / cal_REP5 (:505-519), criterion_FSCORE (:460-503).
 * Tables are the fp64 YCB_MESHES entries (lib/dataset/base.py:222-244); pd_rt / gt_rt [n_img][3][4] and cam_intr [n_img][3][3]
 * fp64 (obj_9D_to_mat + root joint, train_diff_hand_obj.py:594-597); obj_id [n_img] indexes the tables; max_verts >= the
 * largest per-object vertex count.  out [n_img][16] = MCE, OCE, MCE2, ADD, ADD-S, ADD<0.1d, ADD-S<0.1d, REP, REP<5px,
 * Chamfer-L2, F-score @ 2 mm, 5 mm, 10 mm, 2 cm, 5 cm, 10 cm (metres / pixels / {0,1} / [0,1]). */
typedef struct vpho_obj_metric_tables {
    const double* bbox3d;          /* [n_obj][8][3] */
    const double* verts_sampled;   /* [n_obj][n_sampled][3] */
    const double* verts;           /* [vert_offset[n_obj]][3], objects concatenated */
    const int* vert_offset;        /* [n_obj + 1] */
    const double* diameter;        /* [n_obj] */
    int n_obj, n_sampled;
} vpho_obj_metric_tables;
/* obj_9D_to_mat (lib/utils/transform_fn.py:85-90) with the root joint added to the translation (Trainer.postprocess,
 * train_diff_hand_obj.py:578-597): pose9 [n][9] fp64 = [rot6d | t], root_joint [n][3] fp32 -> rt [n][3][4] fp64. */
VPHO_API int vpho_obj_9d_to_rt_f64(const double* pose9, const float* root_joint, int n, double* rt, void* stream);
VPHO_API long long vpho_obj_metrics_workspace_bytes(const vpho_obj_metric_tables* t, int n_img, int max_verts);
VPHO_API int vpho_obj_metrics_f64(const vpho_obj_metric_tables* t, const double* pd_rt, const double* gt_rt, const double* cam_intr,
                         const int* obj_id, int n_img, int max_verts, double* out, void* workspace, long long workspace_bytes,
                         void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Pseudo-force label optimisation (force_optim.py / lib/engine/force_optimization.py:110-207).
 * vpho_anchor_frames_f32: ForceAnchor.__call__ (lib/utils/physics_fn.py:224-257) -> pts [n][32][3], frames [n][32][3][3]
 * (frame[j][i] = component j of axis i), computed ONCE (the reference recomputes them in each of its 3000 iterations).
 * vpho_force_optimize_f32: the whole AdamW loop (two optimisers: `phase1_iters` steps on the cone weights against the
 * gravity-alignment loss, then scale+weights against force-balance + moment + contact-distribution losses) as one
 * persistent workgroup per batch of B <= 64 samples (the batch-mean force loss couples the samples of a batch, :146).
 * Inputs are laid out [n_batches*B]...; gravity/com [..][3] in the flipped (right-hand) frame; outputs force_local /
 * force_global [..][32][3] (zero where !is_grasped, :199-202), final scale [..][32], weight [..][32][8],
 * losses [n_batches][4] = (force, gravity, moment, distribution) of the last iteration. */
VPHO_API int vpho_anchor_frames_f32(const vpho_anchor_tables* t, const float* verts, long long n_hands, float* pts, float* frames, void* stream);
VPHO_API int vpho_force_optimize_f32(const float* pts, const float* frames, const float* gravity, const float* com, const float* force_contact,
                            const unsigned char* is_grasped, int n_batches, int B, int iters, int phase1_iters, float lr,
                            float* force_local, float* force_global, float* scale, float* weight, float* losses, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Hand-object contact detection (lib/utils/physics_fn.py:47-117,201-221; caller
 * lib/dataset/base.py:841-912).  vpho_contact_detect_f32 is one direction of detect_hand_and_object_contact: for each of
 * n samples, every query point (with unit normal) finds its nearest target point, gates on the signed distance along the
 * normal in (normal_lo, normal_hi) and the tangential distance < vertical_thresh, and gets the double-sigmoid weight
 * normalised to 1 at distance 0; nn_index (optional) = nearest target index where in contact, else -1.  Call it hand->object
 * for hand_contact_map and object->hand for obj_contact_map / obj_contact_to_hand_vert.
 * vpho_force_contact_f32 pools a hand contact map (rows of `ld` >= 778 floats) onto the 32 CPF anchors and applies the
 * check_is_grasped rule. */
VPHO_API int vpho_contact_detect_f32(const float* query, const float* query_normals, const float* target, int n, int n_query, int n_target,
                            float normal_lo, float normal_hi, float vertical_thresh, float decay_lo, float decay_hi,
                            float* weight, int* nn_index, void* stream);
VPHO_API int vpho_force_contact_f32(const vpho_anchor_tables* t, const float* hand_contact, int ld, int n, float thresh,
                           float* force_contact, unsigned char* is_grasped, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training of the score networks (first slice): the denoising-score-matching step of
 * ScoreBasedModelAgent.get_score_loss / loss_fn (lib/model/score_based_model.py:11-42,117-128) on BaseDenoiser
 * (lib/model/denoiser.py:68-82) -- forward with saved activations, analytic backward to every denoiser parameter and to the
 * image encoding, AdamW (lib/engine/train_diff_hand_obj.py:49-52).  rows = repeat_num * batch, row r = rep * batch + b.
 * The GEMMs run through vpho_conv2d_nhwc_f32 (1x1 convolutions on [rows][C] matrices); these are the pieces around them. */
/* std = 0.01 * 5000^t (sde.py:15-18), x_t = gt_pose[b] + z * std zero-padded to Dp columns, emb = [sin, cos](t W 2 pi)
 * (denoiser.py:29-31).  t [rows], z [rows][D], fourier_W [64] -> x_t [rows][Dp], emb [rows][128], std_out [rows] */
VPHO_API int vpho_dsm_prepare_f32(const float* gt_pose, const float* t, const float* z, const float* fourier_W, int bs, int reps, int D, int Dp,
                         float* x_t, float* emb, float* std_out, void* stream);
/* second ParallelLinear (256 -> 3 per head, parallel_linear.py:27-35) + division by (std + 1e-7) (denoiser.py:80-81):
 * h [rows][nheads*256], w2 [nheads][256][3], b2 [nheads][3] -> score [rows][3*nheads] */
VPHO_API int vpho_plinear2_fwd_f32(const float* h, const float* w2, const float* b2, const float* std_rows, long long rows, int nheads,
                          float* score, void* stream);
/* loss = mean over batch*reps of sum_d std^2 (score - target)^2, target = -z std / std^2 (score_based_model.py:33-41);
 * dout = d loss / d (un-normalised head output) [rows][D]; loss: one double on the device; partial_ws: >= 1024 doubles */
VPHO_API int vpho_dsm_loss_f32(const float* score, const float* z, const float* std_rows, long long rows, int D, int batch_times_reps,
                      float* dout, double* loss, double* partial_ws, int partial_cap, void* stream);
/* Weight gradient of an NHWC convolution without materialised im2col / transposes (torch.nn.functional.conv2d's autograd for
 * every nn.Conv2d of lib/model/backbone_FPN_HFL.py, encoding.py, head_inplane.py under lib/engine/train_diff_hand_obj.py:181-182):
 * dw[co][(r*KW+s)*Cin + ci] = sum_{n,oy,ox} dy[n,oy,ox,co] * x[n, oy*stride+r-pad_y, ox*stride+s-pad_x, ci]   (zero outside the image)
 * in the packed layout of the forward weights.  x [N][H][W][x_ld], dy [N][OH][OW][dy_ld]; Cin, Cout and both leading dimensions
 * multiples of 4, pointers 16-byte aligned.  The pixel range is reduced in slices (fp32 MFMA accumulation inside a slice, slices
 * added in ascending order): workspace of vpho_conv2d_wgrad_workspace_bytes(...) bytes (may be 0 -> workspace may be NULL). */
VPHO_API long long vpho_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW);
VPHO_API int vpho_conv2d_wgrad_nhwc_f32(const float* x, int N, int H, int W, int Cin, int x_ld, const float* dy, int OH, int OW, int Cout, int dy_ld,
                               int KH, int KW, int stride, int pad_y, int pad_x, float* dw, void* workspace, void* stream);
/* The same weight gradient when dY is known to be zero outside RoI windows (the gradient of a map that is only read through RoIAlign:
 * the FPN smoothing convolutions, VPHO.py:126-129): the reduction runs over the ascending list of live 32-pixel groups that
 * vpho_window_groups_i32 builds on the device from the window table of vpho_roi_windows_i32 (a group is live when one of its pixels
 * lies in its image's window); the pixel slices cut the list instead of the pixel range.  Needs OW % 32 == 0.  Same workspace. */
VPHO_API int vpho_window_groups_i32(const int* wins, int N, int H, int W, int* group_list, int* group_count, void* stream);
VPHO_API int vpho_conv2d_wgrad_groups_nhwc_f32(const float* x, int N, int H, int W, int Cin, int x_ld, const float* dy, int OH, int OW, int Cout,
                                      int dy_ld, int KH, int KW, int stride, int pad_y, int pad_x, const int* group_list,
                                      const int* group_count, float* dw, void* workspace, void* stream);
/* Training-mode tail of HeadMano for one batch of hands (lib/model/head_mano.py:60-87 forward + get_hand_verts, :89-133 get_loss,
 * weights applied as lib/model/VPHO.py:214-219): rot6d [bs][96] (fc_pose output) -> rotation_6d_to_matrix -> ManoLayer -> root-centred
 * vertices / joints in metres (optional outputs verts [bs][778][3], joints [bs][21][3], may be NULL); the four losses against
 * gt_vert / gt_joint / gt_rot6d (= mano_aa_to_6D(gt pose)[:96]) / gt_shape (right hands only, is_right [bs]) and the gradient of
 * w_vert*vert_loss + w_joint*joint_loss + w_pose*mano_pose_loss + w_shape*mano_shape_loss w.r.t. rot6d and shape.
 * loss_parts [bs][4]: per-hand sums of squared differences (vert, joint, pose, shape) in fp64 -- the caller applies
 * weight / (bs * 2334 | bs * 63 | bs * 96 | bs * 10).  is_ho3d (per hand, optional): the regressed joints of those hands enter the joint
 * loss in HO3D's convention (get_joint_aligned_with_HO3D, VPHO.py:154-157, hand_fn.py:454-461: joints re-ordered, HO3D's own tip vertices).
 * A batch without any right hand gives a shape loss of 0 here (the reference takes the mean of an empty tensor: NaN). */
VPHO_API int vpho_mano_train_f32(const vpho_mano_tables* t, const float* rot6d, const float* shape, const float* gt_vert, const float* gt_joint,
                        const float* gt_rot6d, const float* gt_shape, const unsigned char* is_right, const unsigned char* is_ho3d, int bs,
                        float w_vert, float w_joint, float w_pose, float w_shape,
                        float* d_rot6d, float* d_shape, double* loss_parts, float* verts, float* joints, void* stream);
/* JointsMSELoss (lib/model/head_inplane.py:191-203: nn.MSELoss, mean over all elements) times its loss weight
 * (VPHO.py:214-219): loss[0] = weight * mean((pd - gt)^2) in fp64, grad = weight * 2 (pd - gt) / n.  partial_ws: >= partial_cap doubles */
VPHO_API int vpho_mse_loss_f32(const float* pd, const float* gt, long long n, float weight, float* grad, double* loss, double* partial_ws, int partial_cap,
                      void* stream);
/* backward of the second ParallelLinear and of the ReLU in front of it: dpre [rows][nheads*256] (gradient at the first
 * layer's pre-activation), dw2 [nheads][256][3], db2 [nheads][3] */
VPHO_API int vpho_plinear2_bwd_f32(const float* h, const float* dout, const float* w2, long long rows, int nheads, float* dpre, float* dw2, float* db2,
                          void* stream);
/* dx = y > 0 ? dy : 0 on [rows][cols] slices with leading dimensions (gradient through nn.ReLU given its output y) */
VPHO_API int vpho_relu_bwd_f32(const float* dy, int ld_dy, const float* y, int ld_y, long long rows, int cols, float* dx, int ld_dx, void* stream);
/* out[c] = sum_r x[r][c] (bias gradients; fp64 partial sums over row chunks, combined in a fixed order).
 * workspace: vpho_bn_workspace_bytes(cols) bytes */
VPHO_API int vpho_colsum_f32(const float* x, int ld, long long rows, int cols, float* out, void* workspace, void* stream);
/* out[b][c] = sum_rep x[rep*bs + b][c_off + c]: the encoding is shared by the repeat_num draws of an image */
VPHO_API int vpho_sum_repeats_f32(const float* x, int ld, int c_off, int bs, int reps, int cols, float* out, void* stream);
/* y[c][r] = x[r][c] (operands of the weight-gradient GEMMs) */
VPHO_API int vpho_transpose_f32(const float* x, int rows, int cols, int ldx, float* y, int ldy, void* stream);
/* Transposed im2col, the second operand of a convolution's weight gradient dW[co][(r,s,ci)] = sum_p dY^T[co][p] * out[(r,s,ci)][p]
 * (the product itself is vpho_conv2d_nhwc_f32 as a GEMM): out[(r*KW+s)*Cin + ci][p] = x[n, oy*stride+r-pad_y, ox*stride+s-pad_x, ci],
 * p = (n*OH+oy)*OW+ox, zero outside the image and in the padding columns p >= N*OH*OW of the leading dimension ldo. */
VPHO_API int vpho_im2col_t_f32(const float* x, int N, int H, int W, int Cin, int x_ld, int KH, int KW, int stride, int pad_y, int pad_x,
                      int OH, int OW, float* out, long long ldo, void* stream);
/* nn.BatchNorm2d in training mode on NHWC rows (rows = N*H*W, leading dimension ld >= C) -- every BatchNorm of
 * backbone_FPN_HFL.py / encoding.py / head_inplane.py under model.train() (train_diff_hand_obj.py:171): batch mean and biased
 * variance (fp64 two-level reduction), y = lrelu((x - mean) * invstd * gamma + beta, slope) (slope 1 = no activation), running
 * statistics updated with `momentum` and the unbiased variance; save_mean / save_invstd feed the backward:
 * dbeta = sum dy, dgamma = sum dy * xhat, dx = gamma * invstd / rows * (rows * dy - dbeta - xhat * dgamma).
 * (For a fused activation pass its backward first: vpho_lrelu_bwd_f32 on y.)  res (may be NULL; ABI version 4): the residual
 * branch of a bottleneck, y = lrelu(bn(x) + res, slope) (Bottleneck.forward, backbone_FPN_HFL.py:347-349); its gradient is the
 * gradient of the sum.  workspace: vpho_bn_workspace_bytes(C). */
VPHO_API long long vpho_bn_workspace_bytes(int C);
VPHO_API int vpho_bn_train_forward_f32(const float* x, long long rows, int C, int ld, const float* gamma, const float* beta, float eps, float momentum,
                              float slope, float* running_mean, float* running_var, float* save_mean, float* save_invstd, const float* res,
                              float* y, void* workspace, void* stream);
VPHO_API int vpho_bn_train_backward_f32(const float* x, const float* dy, long long rows, int C, int ld, const float* gamma, const float* save_mean,
                               const float* save_invstd, float* dx, float* dgamma, float* dbeta, void* workspace, void* stream);
/* The same two calls when the reductions were taken by the producing convolution's epilogue (ABI version 12; vpho_conv_desc.stats,
 * vpho_conv3x3_winograd_stats_nhwc_f32): stats = [stats_rows][2][C] float partial sums (forward: sum x | sum x^2 of x's rows; backward:
 * sum dy | sum dy * xhat), combined here in fp64 in a fixed order -- the pass over x (and dy) that vpho_bn_train_*_f32 start with is
 * not run.  Everything else as above. */
VPHO_API int vpho_bn_train_forward_stats_f32(const float* x, long long rows, int C, int ld, const float* stats, int stats_rows, const float* gamma,
                                    const float* beta, float eps, float momentum, float slope, float* running_mean, float* running_var,
                                    float* save_mean, float* save_invstd, const float* res, float* y, void* workspace, void* stream);
/* backward: stats may be NULL (stats_rows 0: the call takes its own reduction pass).  res (may be NULL): the other branch of a residual sum,
 * added to dx (encoding.Residual's identity shortcut, encoding.py:21-36: d input = BatchNorm backward + d shortcut).  dx_colsum (may be NULL;
 * C floats): the column sums of the dx it stores = the bias gradient of the convolution that produced x, taken while dx is written. */
VPHO_API int vpho_bn_train_backward_stats_f32(const float* x, const float* dy, long long rows, int C, int ld, const float* gamma, const float* save_mean,
                                     const float* save_invstd, const float* stats, int stats_rows, const float* res, float* dx, float* dgamma,
                                     float* dbeta, float* dx_colsum, void* workspace, void* stream);
/* dx = y > 0 ? dy : dy * slope: backward of nn.LeakyReLU(slope) / nn.ReLU (slope 0) given its OUTPUT y */
VPHO_API int vpho_lrelu_bwd_f32(const float* dy, const float* y, long long n, float slope, float* dx, void* stream);
/* nn.MaxPool2d backward (backbone_FPN_HFL.py:209): dx[n,iy,ix,c] = sum of dy over the windows whose first maximum (row-major)
 * is (iy,ix); x is the pooling input.  Deterministic gather, no atomics. */
VPHO_API int vpho_maxpool_bwd_nhwc_f32(const float* x, const float* dy, int N, int H, int W, int C, int k, int stride, int pad, float* dx, void* stream);
/* the same with a byte workspace (vpho_maxpool_bwd_workspace_bytes; 0 = not needed) for overlapping windows: arg-max position of
 * every window first, then the gather compares position codes instead of re-scanning windows (bit-identical results) */
VPHO_API long long vpho_maxpool_bwd_workspace_bytes(int N, int H, int W, int C, int k, int stride, int pad);
VPHO_API int vpho_maxpool_bwd_ws_nhwc_f32(const float* x, const float* dy, int N, int H, int W, int C, int k, int stride, int pad, float* dx,
                                 void* workspace, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False) backward (FPN._upsample_add, backbone_FPN_HFL.py:66-68):
 * dy [N][OH][OW][C] -> dx [N][H][W][C] */
VPHO_API int vpho_resize_bilinear_bwd_nhwc_f32(const float* dy, int N, int OH, int OW, int C, int H, int W, float* dx, void* stream);
/* torchvision.ops.roi_align backward (VPHO.py:125-128 under loss.backward()): dy [N][P][P][ldo] (channel slice c_off..c_off+C, optionally
 * W-flipped like the forward) accumulated into dfeat [N][H][W][C] (+=; zero-initialise it, or pass another gradient of the same
 * map to sum both).  One RoI per image (box n belongs to image n).  Channel counts / strides that are multiples of 4 take a gather
 * with a fixed summation order (each feature pixel sums the few bins that reach it); other shapes scatter with fp32 atomics. */
VPHO_API int vpho_roi_align_bwd_nhwc_f32(const float* dy, int ldo, int c_off, int N, int H, int W, int C, const float* boxes, float spatial_scale,
                                int out_size, const unsigned char* flip_w, float* dfeat, void* stream);
/* backward of vpho_align_heatmap_nhwc_f32 (align_hm_to_bbox_rectangle + flip, VPHO.py:333-346,139): dout [N][S][S][C] -> dhm [N][S][S][C]
 * (+=: zero-initialise dhm; a gather with a fixed summation order for S <= 120, fp32 atomics beyond) */
VPHO_API int vpho_align_heatmap_bwd_nhwc_f32(const float* dout, int N, int size, int C, const float* bbox, const float* bbox_rect,
                                    const unsigned char* flip_w, float* dhm, void* stream);
/* y = lrelu(a + b, slope): `out += residual; out = leakyrelu(out)` of Bottleneck.forward (backbone_FPN_HFL.py:347-348); slope 1 = a + b */
VPHO_API int vpho_add_lrelu_f32(const float* a, const float* b, long long n, float slope, float* y, void* stream);
/* torch.optim.AdamW single-tensor step (decoupled weight decay, bias-corrected moments); grad_scale multiplies the gradient
 * first (1 / world_size after a sum all-reduce) */
VPHO_API int vpho_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, void* stream);
/* the same step for a list of tensors in one launch.  segments: device array of n_segments records
 * { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; long long n; long long first_block; } with
 * first_block = running sum of ceil(n / 1024) over the preceding records, total_blocks = that sum over all records. */
VPHO_API int vpho_adamw_multi_f32(const void* segments, int n_segments, long long total_blocks, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float grad_scale, void* stream);

/* ---- physics branch of the training step (lib/model/VPHO.py:170-172,205-212 under loss.backward()) -------------------------------
 * backward of vpho_cross_tokens_f32 (cross_module.py:125-133: .view(bs,32,-1) of the projected maps, cat, + positional code):
 * dtok [bs][65][512] -> d_proj_hand / d_proj_obj [bs][8][8][256] NHWC (NULL for the stream the reference detaches, VPHO.py:170-171)
 * and d_grav_emb [bs][512] (may be NULL) */
VPHO_API int vpho_cross_tokens_bwd_f32(const float* dtok, int bs, float* d_proj_hand, float* d_proj_obj, float* d_grav_emb, void* stream);
/* backward of vpho_add_layernorm_f32 (norm1 / norm2 of nn.TransformerEncoderLayer, post-norm): dy -> dx = d(x + r) and dy_xhat =
 * dy * normalised input per element (d gamma = its column sums, d beta = the column sums of dy) */
VPHO_API int vpho_layernorm_bwd_f32(const float* x, const float* r, const float* gamma, const float* dy, long long rows, int E, float eps,
                           float* dx, float* dy_xhat, void* stream);
/* backward of vpho_mha_dropout_f32 (nn.MultiheadAttention inside the encoder layer, sequence axis = batch, quirk Q3):
 * qkv [S*B][3E], d_out [S*B][E], drop_mask as in the forward (NULL = none) -> dqkv [S*B][3E]; S <= 64 (one launch, tables in LDS) */
VPHO_API int vpho_mha_bwd_f32(const float* qkv, const float* d_out, int S, int B, int E, int nhead, const float* drop_mask, float* dqkv, void* stream);
/* the same for S <= 1024 (a per-rank training batch above 64 images): above 64 positions three launches over a workspace of
 * vpho_mha_bwd_workspace_bytes (two S x S tables per (b, head); 0 for S <= 64, which runs the one-launch kernel; -1 = bad argument) */
VPHO_API long long vpho_mha_bwd_workspace_bytes(int S, int B, int nhead);
VPHO_API int vpho_mha_bwd_ws_f32(const float* qkv, const float* d_out, int S, int B, int E, int nhead, const float* drop_mask, float* dqkv,
                        void* workspace, void* stream);
/* HeadPhysics tail + losses + gradient (physics.py:546-557 get_local_force with the double soft-max of :659-664, :362-371
 * from_local_to_global on the GROUND-TRUTH vertices, :456-500 get_loss; weights as VPHO.py:214-219): scale_raw [bs*32] (fc_scale
 * output), logits [bs*32][8] (fc_weight before its Softmax), com [bs*32][3] (fc_CoM output); frame [bs][32][3][3] / point
 * [bs][32][3] from vpho_anchor_frames_f32 of gt_hand_vert_flip; gt_force_local [bs][32][3]; gravity, gt_com [bs][3] in the flipped
 * frame; weights5 (HOST array) = weight_{force,gravity,torque,supervised,CoM}_loss.  Outputs: force_local [bs*32][3], the weighted
 * losses losses5 (device, fp64, same order) and d(total)/d(scale_raw | logits | com).  partial_ws: bs*5 doubles. */
VPHO_API int vpho_physics_loss_f32(const float* scale_raw, const float* logits, const float* com, const float* anchor, float friction,
                          const float* frame, const float* point, const float* gt_force_local, const float* gravity,
                          const float* gt_com, const unsigned char* is_grasped, const float* weights5, int bs,
                          float* force_local, float* d_scale, float* d_logits, float* d_com, double* losses5, double* partial_ws,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif
