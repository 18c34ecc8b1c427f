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

#ifdef __cplusplus
extern "C" {
#endif

const char* vpho_last_error(void);
int vpho_abi_version(void);

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
} vpho_conv_desc;
int vpho_conv2d_nhwc_f32(const vpho_conv_desc* d, void* stream);

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
} vpho_score_weights;

/* bytes of scratch `vpho_score_eval` / `vpho_ode_sample` need for R = bs*S rows */
long long vpho_score_workspace_bytes(const vpho_score_weights* w, int bs, int S);

/* One score evaluation s(x, t | feat) for R = bs*S rows (row r belongs to image r / S).
 * feat_img: [bs][1024], x: [R][D] fp32, t: scalar shared by all rows, out: [R][D] fp32.  (denoiser.py:68-82) */
int vpho_score_eval(const vpho_score_weights* w, const float* feat_img, int bs, int S, const float* x, float t,
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
 * (the reference casts the hand trajectory to fp32 right after sampling, VPHO.py:243); x_out: [R][D] fp64 final
 * sample after the denoise step.  SYNCHRONISES `stream` once per attempted RK step (8-byte error norm D2H) --
 * the scalar step controller runs on the host exactly as scipy's.  stats_host / step_log_host are host memory
 * (step_log_host may be NULL; capacity in entries of 4 doubles). */
int vpho_ode_sample(const vpho_score_weights* w, const float* feat_img, int bs, int S, const float* init_x,
                    double T0, double eps, int num_steps, double rtol, double atol,
                    void* xs_out, int xs_is_f64, double* x_out,
                    void* workspace, long long workspace_bytes,
                    vpho_ode_stats* stats_host, double* step_log_host, int step_log_cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif
