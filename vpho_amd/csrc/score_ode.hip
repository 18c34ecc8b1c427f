// Score network evaluation and the on-device Dormand-Prince RK45 probability-flow ODE sampler.
//
// One right-hand-side evaluation = 4 launches: time embedding (GEMV), two pose-encoder layers (conv_igemm) and the
// fused score head `score_head_kernel`:
//   block (row tile of 128 hypotheses, head n): H^T[256 x 128] = W1p[n] (256x256, "A") x P2^T ("B") on fp32 MFMA with the
//   hidden unit on the accumulator ROW (registers) and the hypothesis on the COLUMN (lane), so that the second
//   ParallelLinear (256 -> 3) is a register-local dot product followed by ONE cross-half shuffle; the (R x 8192) hidden
//   activation never leaves the CU.  8 waves: each owns 32 hypotheses x 128 hidden units (4 MFMA row tiles); the two
//   hidden halves of a hypothesis are combined through LDS.
// Roofline: fp32 MFMA; algorithmic flop per evaluation in the restructured formulation =
//   R * (2*Dp*256 + 2*256*256 + nheads*(2*256*256 + 2*256*3)) (+ once per image 2*1024*NH, once per eval 2*128*NH).
//
// The RK stage algebra (fp64 state, fp32 stage derivatives exactly as numpy stores them), the error norm and the dense
// output run as elementwise / two-pass-reduction kernels; only the 8-byte error norm crosses PCIe per attempted step.
#include "common.h"
#include <type_traits>
#include "../../include/vpho_hip.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <mutex>
#include <unordered_map>
#include <algorithm>

VPHO_STAMP_DECL(head)

namespace {

#ifndef HEAD_ABLATE
#define HEAD_ABLATE 0                          // timing experiments on head_tile (scripts/kernel_ablate.sh score_ode HEAD_ABLATE ...): 1 no stage fills after the
#endif                                         // first two, 2 no fragment reads after the first, 4 no stage barrier, 8 no epilogue -- wrong results, never in the product build
constexpr int HB_K = 16;
constexpr double SIGMA_MIN = 0.01, SIGMA_MAX = 50.0;

// --------------------------------------------------------------------------------------------- time embedding
// ct[o] = b1[o] + sum_k relu(t_b[k] + sum_i t_w[k][i] * gfp(t)[i]) * w1_t[k][o]        (denoiser.py:29-31,71-72)
struct TimeList { float t[8]; };

// Device-resident RK45 controller state (one per sampler workspace).  With it a whole solve is enqueued without a host round
// trip per attempted step: every kernel of an attempt takes its scalars (h, stage times, sigma(t), which of the two state
// buffers is y, which stage slot holds f(t, y)) from here, and rk_begin / rk_end below are scipy's scalar step controller.
struct RkCtl {
    double t, h_abs, h, t_new, err, dense_h;
    double T0, tf, rtol, atol, max_step, g_scale;      // g_scale = sqrt(2 (ln sigma_max - ln sigma_min))
    double d[4];                                       // select_initial_step: d0, d1, d2, h0
    double *te, *dense_p, *log;                        // t_eval [num_steps], stamp powers [num_steps][4], step log [cap][4]
    int done, accepted, rejected, parity, kswap, status;
    int next_idx, num_steps, dense_first, dense_last, dense_parity, dense_kswap;
    int n_accepted, n_rejected, nfev, n_log, log_cap, n_attempts;
    float ts[8], inv_std[8], coef[8];
};
// first-same-as-last: after an accepted step the roles of stage slots 0 and 6 are exchanged
__device__ __host__ inline int kslot(int j, int kswap) { return (kswap && (j == 0 || j == 6)) ? 6 - j : j; }
// ctl_mode of a kernel: 0 no controller; 1 part of an attempt (skip once the solve is done, scalars from the controller);
// 2 part of the final denoise step (skip until the solve is done)
__device__ inline bool ctl_skip(const RkCtl* ctl, int mode) {
    if (!ctl || mode == 0) return false;
    const int d = ctl->done;
    return mode == 1 ? d != 0 : d == 0;
}
__global__ __launch_bounds__(256) void time_embed_kernel(const vpho_score_weights w, TimeList tl, int NH, float* __restrict__ ct_all,
                                                         const RkCtl* __restrict__ ctl, int ctl_mode) {
    __shared__ float emb[128], tf[128];
    const int tid = threadIdx.x;
    if (ctl_skip(ctl, ctl_mode)) return;
    const float t = (ctl && ctl_mode == 1) ? ctl->ts[blockIdx.y] : tl.t[blockIdx.y];
    float* ct = ct_all + (long long)blockIdx.y * NH;
    if (tid < 64) {
        // x[:, None] * W[None, :] * 2 * np.pi : three fp32 multiplications, left to right
        float a = t * w.t_W[tid];
        a = a * 2.0f;
        a = a * 3.14159265358979323846f;
        emb[tid] = sinf(a);
        emb[tid + 64] = cosf(a);
    }
    __syncthreads();
    if (tid < 128) {
        float s = 0.f;
        for (int i = 0; i < 128; ++i) s += w.t_w[tid * 128 + i] * emb[i];
        s += w.t_b[tid];
        tf[tid] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    const int o = blockIdx.x * 256 + tid;
    if (o < NH) {
        float s = 0.f;
        for (int k = 0; k < 128; ++k) s += tf[k] * w.w1_t[(long long)k * NH + o];
        ct[o] = s;
    }
}

// --------------------------------------------------------------------------------------------- fused pose encoder
// P2 = relu(relu(X W0^T + b0) W2^T + b2)   (denoiser.py:60-65,74): both Linear layers of `pose_encoder` in one launch.
// Block = 32 hypotheses x all 256 hidden units; the 32x256 intermediate stays in LDS.
struct LinComb { double c[7]; int n; double h; };
// the seven stage-derivative buffers K_0..K_6 by LOGICAL stage index; first-same-as-last is a pointer swap on the host
struct KSlots { const float* p[7]; };

// Input rows either come from X ([R][Dp] fp32) or are formed on the fly as the RK stage state
// x = (float)(y + h * sum_j c_j K_j)  (scipy rk_step: y + dy, then ode_func's .float(), score_based_model.py:76)
struct PoseEncArgs {
    const float* X; int Dp;            // [R][Dp]
    const double* y; KSlots ks; int D; LinComb lc; double* ynew; int use_lc;
    const float *w0, *b0, *w2, *b2;    // [256][Dp], [256], [256][256], [256]
    float* out; int R;                 // [R][256]
    const RkCtl* ctl; int ctl_mode, write_ynew; const float* kbase; long long n_el; double *ybuf0, *ybuf1;
};
constexpr int PE_ROWS = 32, PE_H_LD = 260, PE_NST = 3, PE_STAGE = 256 * 32;
__device__ __attribute__((aligned(256))) float g_pe_zero_page[64];

// 8 waves; wave w owns hidden columns 32w..32w+31 (one 32x32 MFMA tile) of the block's 32 hypotheses.  Both weight
// matrices stream as ONE sequence of [256][32] chunks (W0's ceil(Dp/32), then W2's 8) through a 3-stage LDS ring filled by
// global_load_lds (unpadded 128-B rows, 16-B chunk index XOR (row>>1)&7 on the source address), two chunks in flight
// across each raw s_barrier (counted vmcnt), so the weight latency hides behind the MFMAs and the first two chunks load
// while the stage state is being formed.  Biases are staged to LDS up front: an ordinary global load consumed while a
// direct-to-LDS load is in flight would drain the ring (vmcnt is in-order).
__global__ __launch_bounds__(512) void pose_encoder_kernel(const PoseEncArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int K1 = (a.Dp + 31) / 32 * 32, X_LD = K1 + 4;
    float* Ws = smem;                              // [PE_NST][256][32]
    float* H1 = Ws + PE_NST * PE_STAGE;            // [32][PE_H_LD]
    float* Bs = H1 + PE_ROWS * PE_H_LD;            // [2][256] biases
    float* Xs = Bs + 512;                          // [32][X_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int r0 = blockIdx.x * PE_ROWS;
    const int n1 = K1 / 32, nch = n1 + 8;
    if (ctl_skip(a.ctl, a.ctl_mode)) return;
    KSlots ks = a.ks;
    const double* yv = a.y;
    double* ynew = a.ynew;
    double lch = a.lc.h;
    if (a.ctl && a.ctl_mode == 1 && a.use_lc) {
        const int par = a.ctl->parity, sw = a.ctl->kswap;
        yv = par ? a.ybuf1 : a.ybuf0;
        ynew = a.write_ynew ? (par ? a.ybuf0 : a.ybuf1) : nullptr;
        lch = a.ctl->h;
#pragma unroll
        for (int j = 0; j < 7; ++j) ks.p[j] = a.kbase + (long long)kslot(j, sw) * a.n_el;
    }

    // ring fill: one wave instruction = 8 rows x 128 B; wave w, pass j fills rows 64j + 8w .. +7
    const int frow = wave * 8 + (lane >> 3);                               // + 64 j
    const int fkq = (lane & 7) ^ ((frow >> 1) & 7);                        // logical 16-B chunk (64j keeps (row>>1)&7)
    auto issue = [&](int c) {
        float* dst = Ws + (c % PE_NST) * PE_STAGE + wave * 8 * 32;
        if (c < n1) {
            const int k = c * 32 + 4 * fkq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* src = k < a.Dp ? a.w0 + (long long)(frow + 64 * j) * a.Dp + k : g_pe_zero_page;
                __builtin_amdgcn_global_load_lds(src, dst + 64 * j * 32, 16, 0, 0);
            }
        } else {
            const float* src = a.w2 + (long long)frow * 256 + (c - n1) * 32 + 4 * fkq;
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(src + 64 * j * 256, dst + 64 * j * 32, 16, 0, 0);
        }
    };
    issue(0);
    issue(1);

    Bs[tid] = tid < 256 ? a.b0[tid] : a.b2[tid - 256];
    for (int i = tid; i < PE_ROWS * (K1 / 4); i += 512) {
        const int r = i / (K1 / 4), c = (i - r * (K1 / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < a.R && c < a.Dp) {
            if (!a.use_lc) v = *reinterpret_cast<const f32x4*>(a.X + (long long)(r0 + r) * a.Dp + c);
            else {
                // all seven stage slots are loaded up front (one memory round trip instead of lc.n dependent ones);
                // slots >= lc.n are read from slot 0 and never enter the sum, which keeps its order j = 0..n-1
                const long long e0 = (long long)(r0 + r) * a.D + c;
                float kv[7][4];
                double yl[4];
                if ((a.D & 3) == 0) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const f32x4 k4 = *reinterpret_cast<const f32x4*>(ks.p[j < a.lc.n ? j : 0] + e0);
                        kv[j][0] = k4[0]; kv[j][1] = k4[1]; kv[j][2] = k4[2]; kv[j][3] = k4[3];
                    }
                    const f64x2 y01 = *reinterpret_cast<const f64x2*>(yv + e0), y23 = *reinterpret_cast<const f64x2*>(yv + e0 + 2);
                    yl[0] = y01[0]; yl[1] = y01[1]; yl[2] = y23[0]; yl[3] = y23[1];
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const long long e = c + u < a.D ? e0 + u : e0;
#pragma unroll
                        for (int j = 0; j < 7; ++j) kv[j][u] = ks.p[j < a.lc.n ? j : 0][e];
                        yl[u] = yv[e];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double sacc = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; ++j) sacc = j < a.lc.n ? sacc + (double)kv[j][u] * a.lc.c[j] : sacc;
                    const double xv = yl[u] + sacc * lch;
                    if (c + u < a.D) {
                        v[u] = (float)xv;
                        if (ynew) ynew[e0 + u] = xv;
                    }
                }
            }
        }
        *reinterpret_cast<f32x4*>(Xs + r * X_LD + c) = v;
    }

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int sw = (li >> 1) & 7;
    struct Frags { f32x4 a[4], b[4]; };
    auto frags = [&](int c, Frags& f) {
        const float* As = c < n1 ? Xs + li * X_LD + c * 32 + 4 * lh : H1 + li * PE_H_LD + (c - n1) * 32 + 4 * lh;
        const float* Wc = Ws + (c % PE_NST) * PE_STAGE + (wave * 32 + li) * 32;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f.a[kk] = *reinterpret_cast<const f32x4*>(As + kk * 8);
            f.b[kk] = *reinterpret_cast<const f32x4*>(Wc + (((2 * kk + lh) ^ sw) << 2));
        }
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk][q], f.b[kk][q], acc, 0, 0, 0);
    };
    // Chunk c has landed for this wave once at most the 4 loads of chunk c+1 are outstanding; the barrier makes that true
    // for every wave's share and retires everybody's fragment reads of stage (c-1)%3, which chunk c+2 then overwrites.
    // The MFMAs run one phase behind the fragment reads: chunk c's ds_reads overlap chunk c-1's matrix work.
    auto phase = [&](int c, Frags& fload, const Frags& fuse) {
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 2 < nch) issue(c + 2);
        if (c != n1) {
            frags(c, fload);
            mfmas(fuse);
        } else {
            // layer boundary: h1 = relu(acc + b0) -> LDS [row][hidden] (C layout: hidden unit on the lane, hypothesis row
            // on the register) must be complete in every wave before anyone reads it as the next A operand
            mfmas(fuse);
            const int col = wave * 32 + li;
            const float bv = Bs[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = acc[e] + bv;
                H1[row * PE_H_LD + col] = v > 0.f ? v : 0.f;
                acc[e] = 0.f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            frags(c, fload);
        }
    };
    Frags f0, f1;
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(2);
    frags(0, f0);
    for (int c = 1; c < nch; c += 2) {
        phase(c, f1, f0);
        if (c + 1 < nch) phase(c + 1, f0, f1);
    }
    mfmas((nch & 1) ? f0 : f1);
    const int col = wave * 32 + li;
    const float bv = Bs[256 + col];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < a.R) {
            const float v = acc[e] + bv;
            a.out[(long long)row * 256 + col] = v > 0.f ? v : 0.f;
        }
    }
}

// Round 4: the same two layers with the weight fragments in REGISTERS.  A wave only ever reads its own 32 rows of a weight chunk,
// so the ring above used LDS as a prefetch buffer, not for sharing: 96 of the kernel's 143 KB, which made a workgroup own its CU --
// under the pipelined evaluator a pose-encoder workgroup had to wait until BOTH convolution workgroups of a CU had drained (135 us
// average against 22 us exclusive), 102 times per step on the sampler's serial chain.  Here a lane loads its B fragments
// (W[32 wave + lane&31][32 c + 8 kk + 4 (lane>>5) .. +3], 16 bytes) straight from L2 into a three-chunk register ring, three chunks
// ahead of their use; the k order of the products is the ring kernel's, so the outputs are bit-identical.  What is left in LDS: the
// A operand (stage state X, then -- aliased, the state is dead by then -- the 32 x 256 intermediate) and the biases: 35 KB, three
// workgroup barriers instead of twelve, <= 128 registers: the footprint of ONE direct-convolution workgroup, so it takes the next
// free slot of any CU instead of a whole CU.
// RT = row tiles of 32 hypotheses per workgroup.  RT = 1: the kernel as described above.  RT = 2 (round 6, launches of more than 8 192 rows --
// BASELINE cfg4's 32 768): a workgroup streams the same 352 KB of weights for 64 rows instead of 32 -- every B fragment feeds two matrix
// instructions, half the L2 -> register weight traffic per row -- with the A fragments read one 8-k slice ahead instead of a whole chunk
// ahead (16 registers for two tiles); 68 KB of LDS.  Same k order per output element: bit-identical to RT = 1.
template <int N1, int RT>
__device__ __forceinline__ void pe_reg_body(const PoseEncArgs& a) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    constexpr int K1 = N1 * 32, X_LD = K1 + 4, NCH = N1 + 8, ROWS = PE_ROWS * RT;
    float* H1 = smem;                              // [ROWS][PE_H_LD]   (layer 2's A operand)
    float* Xs = smem;                              // [ROWS][X_LD]      (layer 1's A operand; dead before H1 is written)
    float* Bs = smem + ROWS * PE_H_LD;             // [2][256] biases
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int r0 = blockIdx.x * ROWS;
    if (ctl_skip(a.ctl, a.ctl_mode)) return;
    KSlots ks = a.ks;
    const double* yv = a.y;
    double* ynew = a.ynew;
    double lch = a.lc.h;
    if (a.ctl && a.ctl_mode == 1 && a.use_lc) {
        const int par = a.ctl->parity, sw = a.ctl->kswap;
        yv = par ? a.ybuf1 : a.ybuf0;
        ynew = a.write_ynew ? (par ? a.ybuf0 : a.ybuf1) : nullptr;
        lch = a.ctl->h;
#pragma unroll
        for (int j = 0; j < 7; ++j) ks.p[j] = a.kbase + (long long)kslot(j, sw) * a.n_el;
    }
    struct WFrag { f32x4 b[4]; };
    const float* w0row = a.w0 + (long long)(wave * 32 + li) * a.Dp + 4 * lh;
    const float* w2row = a.w2 + (long long)(wave * 32 + li) * 256 + 4 * lh;
    auto wload = [&](int c, WFrag& w) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (c < N1) {
                const int k = c * 32 + 8 * kk + 4 * lh;                      // beyond Dp: load column 0 instead and drop it (no branch)
                const f32x4 v = *reinterpret_cast<const f32x4*>(k < a.Dp ? w0row + c * 32 + 8 * kk : w0row - 4 * lh);
                w.b[kk] = k < a.Dp ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                w.b[kk] = *reinterpret_cast<const f32x4*>(w2row + (c - N1) * 32 + 8 * kk);
            }
        }
    };
    WFrag wr[3];
    wload(0, wr[0]);
    wload(1, wr[1]);
    wload(2, wr[2]);
    __builtin_amdgcn_sched_barrier(0);       // the first three weight chunks are out and stay in flight (L2 hits) while the stage state is formed

    Bs[tid] = tid < 256 ? a.b0[tid] : a.b2[tid - 256];
    for (int i = tid; i < ROWS * (K1 / 4); i += 512) {
        const int r = i / (K1 / 4), c = (i - r * (K1 / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < a.R && c < a.Dp) {
            if (!a.use_lc) v = *reinterpret_cast<const f32x4*>(a.X + (long long)(r0 + r) * a.Dp + c);
            else {
                // all seven stage slots are loaded up front (one memory round trip instead of lc.n dependent ones);
                // slots >= lc.n are read from slot 0 and never enter the sum, which keeps its order j = 0..n-1
                const long long e0 = (long long)(r0 + r) * a.D + c;
                float kv[7][4];
                double yl[4];
                if ((a.D & 3) == 0) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const f32x4 k4 = *reinterpret_cast<const f32x4*>(ks.p[j < a.lc.n ? j : 0] + e0);
                        kv[j][0] = k4[0]; kv[j][1] = k4[1]; kv[j][2] = k4[2]; kv[j][3] = k4[3];
                    }
                    const f64x2 y01 = *reinterpret_cast<const f64x2*>(yv + e0), y23 = *reinterpret_cast<const f64x2*>(yv + e0 + 2);
                    yl[0] = y01[0]; yl[1] = y01[1]; yl[2] = y23[0]; yl[3] = y23[1];
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const long long e = c + u < a.D ? e0 + u : e0;
#pragma unroll
                        for (int j = 0; j < 7; ++j) kv[j][u] = ks.p[j < a.lc.n ? j : 0][e];
                        yl[u] = yv[e];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double sacc = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; ++j) sacc = j < a.lc.n ? sacc + (double)kv[j][u] * a.lc.c[j] : sacc;
                    const double xv = yl[u] + sacc * lch;
                    if (c + u < a.D) {
                        v[u] = (float)xv;
                        if (ynew) ynew[e0 + u] = xv;
                    }
                }
            }
        }
        *reinterpret_cast<f32x4*>(Xs + r * X_LD + c) = v;
    }
    __syncthreads();

    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // layer boundary: every wave's last reads of Xs are consumed (their products have issued) before H1 = relu(acc + b0) is written over
    // it, and H1 is complete in every wave before anyone reads it as the next A operand
    auto boundary = [&]() {
        __syncthreads();
        const int col = wave * 32 + li;
        const float bv = Bs[col];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = acc[t][e] + bv;
                H1[row * PE_H_LD + col] = v > 0.f ? v : 0.f;
                acc[t][e] = 0.f;
            }
        __syncthreads();
    };
    if constexpr (RT == 1) {
        struct AFrag { f32x4 a[4]; };
        auto aload = [&](int c, AFrag& f) {
            const float* As = c < N1 ? Xs + li * X_LD + c * 32 + 4 * lh : H1 + li * PE_H_LD + (c - N1) * 32 + 4 * lh;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) f.a[kk] = *reinterpret_cast<const f32x4*>(As + kk * 8);
        };
        auto mfmas = [&](const AFrag& f, const WFrag& w) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk][q], w.b[kk][q], acc[0], 0, 0, 0);
        };
        AFrag fa[2];
        aload(0, fa[0]);
        // the sched_barriers pin the order "next A fragments, this chunk's 16 MFMAs, the weight loads of chunk c + 3": left to itself the
        // scheduler sinks every weight load to just in front of its use (shorter live ranges under the 128-register cap): 4 MFMAs of cover
        // for an L2 round trip instead of 2 chunks
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH && c + 1 != N1) aload(c + 1, fa[(c + 1) & 1]);       // next chunk's A fragments while this chunk multiplies
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[c & 1], wr[c % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 3 < NCH) wload(c + 3, wr[c % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 == N1) { boundary(); aload(N1, fa[N1 & 1]); }
        }
    } else {
        // two row tiles: the A fragments of one 8-k slice (2 x 16 bytes) are read one slice ahead of the 8 matrix instructions that use them
        auto aslice = [&](int c, int kk, f32x4 (&f)[RT]) {
            const float* As = c < N1 ? Xs + li * X_LD + c * 32 + 4 * lh + kk * 8 : H1 + li * PE_H_LD + (c - N1) * 32 + 4 * lh + kk * 8;
            const int ld = c < N1 ? X_LD : PE_H_LD;
#pragma unroll
            for (int t = 0; t < RT; ++t) f[t] = *reinterpret_cast<const f32x4*>(As + 32 * t * ld);
        };
        f32x4 fa[2][RT];
        aslice(0, 0, fa[0]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int s = (c * 4 + kk) & 1;
                if (kk < 3) aslice(c, kk + 1, fa[s ^ 1]);
                else if (c + 1 < NCH && c + 1 != N1) aslice(c + 1, 0, fa[s ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][t][q], wr[c % 3].b[kk][q], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c + 3 < NCH) wload(c + 3, wr[c % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 == N1) { boundary(); aslice(N1, 0, fa[(N1 * 4) & 1]); }
        }
    }
    const int col = wave * 32 + li;
    const float bv = Bs[256 + col];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + 32 * t + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (row < a.R) {
                const float v = acc[t][e] + bv;
                a.out[(long long)row * 256 + col] = v > 0.f ? v : 0.f;
            }
        }
}
template <int N1>
__global__ __launch_bounds__(512, 4) void pose_encoder_reg_kernel(const PoseEncArgs a) { pe_reg_body<N1, 1>(a); }
template <int N1>
__global__ __launch_bounds__(512, 4) void pose_encoder_reg64_kernel(const PoseEncArgs a) { pe_reg_body<N1, 2>(a); }

// --------------------------------------------------------------------------------------------- fused score head
struct HeadArgs {
    const float* w1p;    // [NH][256]
    const float* p2;     // [R][256]
    const float* cimg;   // [bs][NH]  (feat part + b1)
    const float* ct;     // [NH]
    const float* w2;     // [NH][4]
    const float* b2;     // [nheads*3]
    float* out;          // [R][D]
    int* nan_count;
    int R, S, NH, D;
    float inv_std_den;   // std + 1e-7
    float coef;          // rhs = 0 - coef*score when rhs_mode, else score
    int rhs_mode;
    const RkCtl* ctl; int ctl_mode, stage, out_slot; float* kbase; long long n_el;
    int nheads, full_tiles, tail_tiles;   // per head: 128-row tiles, then 32-row tail tiles (launch order: all full tiles first)
    const unsigned short* w1p_split;      // [nheads][3][256][256] bf16 planes of w1p (split-bf16 kernel only)
};

// One tile of the score head: ROWS = 32*RG hypotheses of head n.  TI = MFMA row tiles (32 hidden units each) per wave:
//   TI = 4: 128 hypotheses, wave = (row group rg = wave & 3) x (hidden half hh = wave >> 2)         -- the ordinary tile
//   TI = 1:  32 hypotheses, wave = hidden units 32*wave .. 32*wave+31 of the same 32 hypotheses      -- the tail tile
// A launch whose tile count is not a multiple of the chip's workgroup slots ends in a round that keeps a few CUs busy for a whole
// tile time (6 400 rows x 32 heads = 1 600 tiles on 512 slots: 3.125 rounds cost 4).  The launch therefore cuts the rows beyond the
// last full round into quarter tiles that all CUs share: 1 536 ordinary tiles = 3 rounds exactly, then 256 tail tiles, one per CU.
// (Round 5, in-kernel stamps: a tail tile lives 13 us, 10.7 of them in its 16 stages = 0.67 us per stage for 0.21 us of matrix work, and the
// 256 of them add ~20 us to a 232-us launch.  A four-stage prefetch ring for them -- fills three stages ahead, 80 KB of LDS -- changed
// nothing (233-235 us): the stage time is the 24 KB of weights a tail workgroup streams for 32 rows, ~70 GB/s per CU from L2, not a
// latency.  Dropped.)
template <int TI, bool CB>
__device__ __forceinline__ void head_tile(const HeadArgs& a, float* smem, const int n, const int r0) {
    constexpr int ROWS = TI == 4 ? 128 : 32, PARTS = 8;              // partial sums per row: one per 32 hidden units, whatever the tile kind
    constexpr int STAGE = (256 + 128) * HB_K;
    float* Eb = smem + 2 * STAGE;
    VPHO_STAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int rg = TI == 4 ? (wave & 3) : 0, hh = TI == 4 ? (wave >> 2) : wave;   // hidden base of the wave = 32*TI*hh
    // one wave instruction fills 64 x 16 B = RPW whole tile rows; chunk swizzle f(row) = (row >> SW_SHIFT) & (CPR-1) keeps
    // the 16 lanes of every ds_read_b128 group on distinct 16-B slots of a 256-B bank row (rows of 128 B: 2 per bank row,
    // rows of 64 B: 4 per bank row)
    constexpr int CPR = HB_K / 4, RPW = 64 / CPR, RPP = 8 * RPW, SW_SHIFT = CPR == 8 ? 1 : 2;
    static_assert(CPR == 8 || CPR == 4, "BK must be 32 or 16");
    const int lrow = wave * RPW + lane / CPR;           // tile row this lane fills (per pass: + RPP*j)
    const int kq = (lane % CPR) ^ ((lrow >> SW_SHIFT) & (CPR - 1));      // logical 16-B chunk this lane fetches
    const float* Wg = a.w1p + (long long)n * 256 * 256;

    // Per-image terms of the epilogue (cimg: the encoding's share of the first layer, one 256-vector per image and head).  A tile's rows
    // span a few images (128 rows of sample_num 100: at most 3): their vectors are staged in LDS next to the epilogue table, slot stride
    // 257 floats so that the lanes of a half-wave -- same hidden unit, neighbouring images -- hit different banks.  Round 4 read them
    // with 64 global loads per lane in the epilogue: 16 waves x 64 dword loads through the CU's one texture-address path = ~7 us of a
    // 62-us tile life with the matrix pipe idle (profiles/r05_inkernel_clock.txt: epilogue 13.6 us median).  Same values, same order of
    // additions: bit-identical.  CB = sample_num >= 64 (chosen by the host: a 128-row tile then spans at most 3 images); smaller
    // sample_num (a 128-row tile of sample_num 4 spans 33 images -- and is a tiny launch) keeps the global loads.
    constexpr int CB_LD = 257;
    float* Cb = Eb + 256 * 4;                                       // [3][CB_LD] <= the 1024 floats behind the table
    const int img0 = r0 / a.S;
    const int img_last = (min(r0 + ROWS, a.R) - 1) / a.S;
    if (tid < 256) {   // epilogue table
        f32x4 e;
        e[0] = a.ct[n * 256 + tid];
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(a.w2 + (long long)(n * 256 + tid) * 4);
        e[1] = w2[0]; e[2] = w2[1]; e[3] = w2[2];
        *reinterpret_cast<f32x4*>(Eb + tid * 4) = e;
    } else if (CB) {
        const int j = tid - 256;
        for (int sl = 0; sl <= img_last - img0; ++sl) Cb[sl * CB_LD + j] = a.cimg[(long long)(img0 + sl) * a.NH + n * 256 + j];
    }

    // scalars of the epilogue's last step, requested now (a dependent chain of global loads behind the tile's last barrier otherwise).
    // (Only on the CB path: the global-load instantiation -- sample_num < 64, tiny launches -- is 2 registers short of keeping them.)
    float hd_inv_std = a.inv_std_den, hd_coef = a.coef, hd_b2 = 0.f;
    float* hd_out = a.out;
    auto epilogue_scalars = [&]() {
        if (a.ctl && a.ctl_mode == 1) {
            hd_inv_std = a.ctl->inv_std[a.stage]; hd_coef = a.ctl->coef[a.stage];
            hd_out = a.kbase + (long long)kslot(a.out_slot, a.ctl->kswap) * a.n_el;
        }
        hd_b2 = a.b2[n * 3 + tid % 3];
    };
    if (CB) epilogue_scalars();

    // tiles through buffer resources: per-lane byte offsets are loop constants, the k advance is the instruction's scalar offset;
    // hypothesis rows beyond R (or beyond the tile) carry an out-of-range offset (the hardware writes zeros)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wg), 0, 256 * 256 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p2), 0, 0xFFFFFFF0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    int woff[256 / RPP], poff[128 / RPP];
#pragma unroll
    for (int j = 0; j < 256 / RPP; ++j) woff[j] = ((lrow + RPP * j) * 256 + 4 * kq) * 4;
#pragma unroll
    for (int j = 0; j < 128 / RPP; ++j) {
        const int lr = lrow + RPP * j, r = r0 + lr;
        poff[j] = (r < a.R && lr < ROWS) ? (int)(((unsigned)r * 256u + 4u * (unsigned)kq) * 4u) : -1;
    }
    auto fill = [&](int buf, int kt) {
        float* Ws = smem + buf * STAGE + wave * RPW * HB_K;
        float* Ps = smem + buf * STAGE + 256 * HB_K + wave * RPW * HB_K;
        const int koff = kt * HB_K * 4;
#pragma unroll
        for (int j = 0; j < 256 / RPP; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(Ws + RPP * j * HB_K), 16, woff[j], koff, 0, 0);
#pragma unroll
        for (int j = 0; j < 128 / RPP; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (lds_ptr)(Ps + RPP * j * HB_K), 16, poff[j], koff, 0, 0);
    };

    f32x16 acc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int sw = (li >> SW_SHIFT) & (CPR - 1);
    constexpr int NK = 256 / HB_K, NKK = HB_K / 8;
    // Two LDS stages.  The barrier of k-tile kt sits before its LAST 8-wide MFMA group: by then every wave has its
    // fragments of stage `buf` in registers, so the stage is refilled (k-tile kt+2) right behind the barrier and the
    // load has a whole k-tile of MFMA time to land before the next barrier needs it.
    fill(0, 0);
    VPHO_STAMP_AT(1);
    VPHO_SYNC_LDS_DMA();
    fill(1, 1);
    VPHO_STAMP_AT(2);
    VPHO_PRIO_MAIN();
    f32x4 b, av[TI];
    for (int kt = 0; kt < NK; ++kt) {
        const int buf = kt & 1;
        const float* As = smem + buf * STAGE + (hh * 32 * TI + li) * HB_K;
        const float* Bs = smem + buf * STAGE + 256 * HB_K + (rg * 32 + li) * HB_K;
        auto frags = [&](int kk) {
            if ((HEAD_ABLATE & 2) && (kt | kk)) return;              // timing: the first fragments only
            const int ch = ((2 * kk + lh) ^ sw) * 4;
            b = *reinterpret_cast<const f32x4*>(Bs + ch);
#pragma unroll
            for (int i = 0; i < TI; ++i) av[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * HB_K + ch);
        };
        auto mfmas = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], b[q], acc[i], 0, 0, 0);
        };
#pragma unroll
        for (int kk = 0; kk < NKK - 1; ++kk) { frags(kk); mfmas(); }
        frags(NKK - 1);
        if (!(HEAD_ABLATE & 4)) VPHO_SYNC_LDS_DMA();
        if (kt + 2 < NK && !(HEAD_ABLATE & 1)) fill(buf, kt + 2);
        mfmas();
    }
    __syncthreads();
    VPHO_PRIO_REST();
    VPHO_STAMP_AT(3);
    if ((HEAD_ABLATE & 8) && acc[0][0] != 1.2345e-30f) return;       // timing: no epilogue

    // epilogue: hidden unit j = 32*TI*hh + 32*i + (e&3) + 8*(e>>2) + 4*lh on the register, hypothesis on the lane
    const int lrow_out = rg * 32 + li;
    const int row = r0 + lrow_out;
    const bool live = row < a.R;
    const float* cim = a.cimg + (long long)(live ? row / a.S : 0) * a.NH + n * 256;
    const float* cbl = Cb + (live ? row / a.S - img0 : 0) * CB_LD;  // this lane's image in the LDS copy (CB)
    // The 256 hidden units of a row are summed as EIGHT partial sums of 32 (hidden 32 p .. 32 p + 31 = one 32 x 32 accumulator tile: two
    // 16-term lane sums added), combined in ascending p -- in an ordinary 128-row tile (a wave holds four of the eight) exactly as in a
    // 32-row tail tile (a wave holds one).  Round 3 let an ordinary tile sum 2 x 128: which rows of a launch fall into tail tiles
    // depends on bs x sample_num and on the CU count, so a hypothesis rounded differently when the batch size or the GPU changed
    // (ADVICE r2).  Now a row's score does not depend on the tile that computed it: bit-identical across tile kinds.
    float* Ob = smem;                                               // [8 parts][ROWS][4]: the stage buffers are free (barrier above)
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = 32 * TI * hh + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const f32x4 t = *reinterpret_cast<const f32x4*>(Eb + j * 4);
            float h = acc[i][e] + (CB ? cbl[j] : cim[j]) + t[0];
            h = h > 0.f ? h : 0.f;
            o0 += h * t[1]; o1 += h * t[2]; o2 += h * t[3];
        }
        o0 += __shfl_xor(o0, 32); o1 += __shfl_xor(o1, 32); o2 += __shfl_xor(o2, 32);
        if (lh == 0) {
            float* ob = Ob + ((hh * TI + i) * ROWS + lrow_out) * 4;
            ob[0] = o0; ob[1] = o1; ob[2] = o2;
        }
        __builtin_amdgcn_sched_barrier(0);                          // one accumulator tile at a time: the four chains interleaved cost 130 more registers
    }
    VPHO_STAMP_AT(5);
    __syncthreads();
    // (Round 5, measured no better on one box, 3 interleaved runs each: the combine WITHOUT this barrier -- the waves that share a row group
    // draw an LDS ticket and the last one combines its 32 rows, 233-238 us against 234-237; s_setprio 1 / 3 around the main loop, 230 against
    // 228-230 (scripts/build_prio.sh).  With the persistent variant below that is four ways of shortening a tile's life outside its main
    // loop without moving the launch time: the 2 x 8 waves of a CU are bound by what they share, not by a workgroup's own serial parts.)
    // One output per thread: thread t -> (row t / 3, component t % 3), 3 x ROWS threads.  Round 4 gave a row's three components to ONE of
    // 128 threads (two of the eight waves: 24 strided LDS reads, three 4-byte stores 384 B apart and three divisions each, behind two
    // dependent global loads of the controller's scalars): 5.7 us of a 62-us tile life with six waves idle (profiles/r05_inkernel_clock.txt).
    // The scalars and b2 are fetched at the kernel's start now (hd_inv_std, hd_coef, hd_out, hd_b2).  Same sums in the same order.
    if (!CB) epilogue_scalars();
    if (tid < 3 * ROWS) {
        const int rl = tid / 3, dd = tid - 3 * rl, orow = r0 + rl;
        if (orow < a.R) {
            float acc2 = Ob[rl * 4 + dd];                                 // partial sums of the hidden slices, ascending
#pragma unroll
            for (int part = 1; part < PARTS; ++part) acc2 += Ob[(part * ROWS + rl) * 4 + dd];
            float sv = (acc2 + hd_b2) / hd_inv_std;
            if (a.rhs_mode) {
                // score_eval_wrapper (score_based_model.py:65-72): nan_to_num(nan=0, posinf=0, neginf=0) on the RHS evaluations of the
                // solve -- there, when ANY entry of the evaluation is NaN; here entry by entry: the same result whenever a NaN is
                // present, and +-inf without any NaN has no defined outcome in the reference (scipy's controller never recovers).
                // The bare score (vpho_score_eval, the final denoise evaluation :100) is NOT guarded, as in the reference.
                if (sv != sv) { sv = 0.f; atomicAdd(a.nan_count, 1); } else if (fabsf(sv) == INFINITY) sv = 0.f;
                sv = 0.f - hd_coef * sv;
            }
            hd_out[(long long)orow * a.D + n * 3 + dd] = sv;
        }
    }
    VPHO_STAMP_AT(4);
    VPHO_STAMP_WRITE(head, blockIdx.x);
}

template <bool CB>
__global__ __launch_bounds__(512, 4) void score_head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    // [2] stages x ([256][HB_K] weights | [128][HB_K] activations), unpadded rows filled by global_load_lds with the 16-B chunk
    // index XOR-swizzled (see conv_igemm_glds_kernel); then [256][4] {ct, w2_0, w2_1, w2_2}; the [8][ROWS][4] partial outputs of the epilogue reuse the stages
    if (ctl_skip(a.ctl, a.ctl_mode)) return;
    // (measured no better, round 5: 64-row tiles for launches with fewer 128-row tiles than CUs -- the object head, 3 heads x 50 tiles = 150
    // workgroups -> 300: 39.4-39.5 us against 39.8-40.2, step unchanged.  A tile alone on its CU is a chain of 16 barrier-separated k stages,
    // ~2.3 us each whether the stage holds 32 or 16 matrix instructions per wave: the launch is as long as one tile's life either way.)
    // (measured no better: dispatching the tail tiles first, next to ordinary tiles, 233 vs 230 us; giving each XCD whole heads of
    // tail tiles, 231 us -- a tail tile is bound by its chain of 16 barrier-separated weight stages, not by where the weights are)
    const int b = blockIdx.x, n_full = a.nheads * a.full_tiles;
    if (b < n_full) {
        head_tile<4, CB>(a, smem, b / a.full_tiles, (b % a.full_tiles) * 128);
    } else {
        const int q = b - n_full;
        head_tile<1, CB>(a, smem, q / a.tail_tiles, a.full_tiles * 128 + (q % a.tail_tiles) * 32);
    }
}

// --------------------------------------------------------------------------------------------- persistent score head (round 5 experiment, opt-in: VPHO_HEAD_PERS=1)
// The in-kernel stamps of profiles/r05_inkernel_clock.txt: a 128-row tile lives 62 us -- entry 0.7, first fill wait 1.0, main loop 45,
// epilogue 11 (partial sums 5.5, then a barrier and a combine that keeps 2 of the 8 waves busy for 5.7), slot turnover 0.8 -- and a CU has
// two workgroups in their main loops for only 46 % of the launch.  Here a workgroup WALKS its tiles (b, b + grid, ...: 128-row tiles of
// all heads first, then the 32-row tail tiles) and
//   * requests the next tile's first two k stages -- and loads its epilogue tables into registers -- right behind the current tile's
//     last barrier, so they land under the epilogue (the partial sums have an LDS region of their own: both stage buffers are free);
//   * finishes a tile with ONE output per thread (384 threads instead of 128 x 3), written by a buffer store that every thread issues
//     (an out-of-range offset for the idle ones): the counted wait for the next tile's first stage is exact;
//   * counts NaNs in a register and adds them once per workgroup.
// Arithmetic, k order, partial sums and their order are head_tile's: bit-identical scores (tests/test_gpu_sampler.py).
#define VPHO_WAIT_VMH(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
template <bool CB>
__global__ __launch_bounds__(512, 4) void score_head_pers_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    if (ctl_skip(a.ctl, a.ctl_mode)) return;
    constexpr int STAGE = (256 + 128) * HB_K;
    constexpr int CPR = HB_K / 4, RPW = 64 / CPR, RPP = 8 * RPW, SW_SHIFT = 2;
    static_assert(HB_K == 16 && RPP == 128, "fill layout");
    constexpr int NFW = 256 / RPP, NF = NFW + 1;                    // LDS-DMA instructions per stage and wave: weights + activations
    constexpr int NK = 256 / HB_K, NKK = HB_K / 8;
    constexpr int CB_LD = 257, B2_AT = 900;
    float* Eb = smem + 2 * STAGE;                                    // [256][4] {ct, w2_0, w2_1, w2_2} of the tile's head
    float* Cb = Eb + 256 * 4;                                        // [3][257] per-image terms of the tile's images | b2 of the head at [B2_AT..+2]
    float* Ob = Cb + 1024;                                           // [8 parts][128 rows][4] partial outputs
    VPHO_STAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int n_full = a.nheads * a.full_tiles, total = n_full + a.nheads * a.tail_tiles, G = gridDim.x;
    int b = blockIdx.x;
    if (b >= total) return;
    float inv_std = a.inv_std_den, coef = a.coef;
    float* outp = a.out;
    if (a.ctl && a.ctl_mode == 1) {
        inv_std = a.ctl->inv_std[a.stage]; coef = a.ctl->coef[a.stage];
        outp = a.kbase + (long long)kslot(a.out_slot, a.ctl->kswap) * a.n_el;
    }
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1p), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p2), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(outp, 0, 0xFFFFFFF0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int lrow = wave * RPW + lane / CPR;
    const int kq = (lane % CPR) ^ ((lrow >> SW_SHIFT) & (CPR - 1));
    int woff[NFW];
#pragma unroll
    for (int j = 0; j < NFW; ++j) woff[j] = ((lrow + RPP * j) * 256 + 4 * kq) * 4;

    // tile b -> head n, first row r0, rows (128 | 32)
    auto tile_of = [&](int t, int& n, int& r0, int& rows) {
        if (t < n_full) { n = t / a.full_tiles; r0 = (t - n * a.full_tiles) * 128; rows = 128; }
        else { const int q = t - n_full; n = q / a.tail_tiles; r0 = a.full_tiles * 128 + (q - n * a.tail_tiles) * 32; rows = 32; }
    };
    int poff;                                                        // activation row of this lane in the tile being filled
    auto set_poff = [&](int r0, int rows) {
        const int r = r0 + lrow;
        poff = (r < a.R && lrow < rows) ? (int)(((unsigned)r * 256u + 4u * (unsigned)kq) * 4u) : -1;
    };
    auto fill = [&](int buf, int kt, int n) {
        float* Ws = smem + buf * STAGE + wave * RPW * HB_K;
        float* Ps = smem + buf * STAGE + 256 * HB_K + wave * RPW * HB_K;
        const int koff = kt * HB_K * 4, wbase = n * (256 * 256 * 4) + koff;
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            const int wo = woff[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(Ws + RPP * j * HB_K), 16, wo, wbase, 0, 0);
        }
        const int po = poff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (lds_ptr)Ps, 16, po, koff, 0, 0);
    };
    // epilogue tables of a tile, into one f32x4 per thread: threads 0..255 the head's row {ct, w2}, threads 256..511 column j = tid - 256 of
    // the per-image terms of the (at most 3) images the tile spans, and b2 of the head in the first three of them
    // (tq: the thread index as an OPAQUE per-call value -- everything derived from a visible threadIdx.x is loop-invariant over the tiles,
    // gets hoisted out of the tile loop and spilled; a scratch reload is a vector-memory operation whose wait also waits for the fills)
    // Branch-free: four dword loads per thread from addresses chosen by selects -- a branch per table would end in a join where the
    // compiler waits for the loads, in front of the fills that have to be requested next
    auto load_tables = [&](int n, int r0, int rows) -> f32x4 {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const bool lo = tq < 256;
        const int j = lo ? tq : tq - 256;
        const int img0 = r0 / a.S, nimg = (min(r0 + rows, a.R) - 1) / a.S - img0;            // images beyond the first
        const float* w2r = a.w2 + (long long)(n * 256 + j) * 4;
        const float* cm = a.cimg + (long long)img0 * a.NH + n * 256 + j;
        const float* p0 = lo ? a.ct + n * 256 + j : cm;
        const float* p1 = lo ? w2r + 0 : cm + (nimg >= 1 ? a.NH : 0);
        const float* p2 = lo ? w2r + 1 : cm + (nimg >= 2 ? 2 * a.NH : 0);
        const float* p3 = lo ? w2r + 2 : a.b2 + n * 3 + min(j, 2);
        f32x4 v;
        v[0] = *p0; v[1] = *p1; v[2] = *p2; v[3] = *p3;
        return v;
    };
    auto store_tables = [&](const f32x4& v) {
        int tq = tid;
        asm volatile("" : "+v"(tq));
        if (tq < 256) *reinterpret_cast<f32x4*>(Eb + tq * 4) = v;
        else {
            const int j = tq - 256;
            if (CB) { Cb[j] = v[0]; Cb[CB_LD + j] = v[1]; Cb[2 * CB_LD + j] = v[2]; }
            if (j < 3) Cb[B2_AT + j] = v[3];
        }
    };

    int n, r0, rows, nans = 0;
    tile_of(b, n, r0, rows);
    {
        const f32x4 tv = load_tables(n, r0, rows);
        set_poff(r0, rows);
        fill(0, 0, n);
        fill(1, 1, n);
        store_tables(tv);
    }
    VPHO_STAMP_AT(1);
    const int sw = (li >> SW_SHIFT) & (CPR - 1);
    bool first = true;

    // one tile: TI = 4 (128 rows: wave = row group rg x hidden half hh) or TI = 1 (32 rows: wave = hidden slice)
    auto tile = [&](auto ti_tag) -> bool {
        constexpr int TI = decltype(ti_tag)::value;
        const int rg = TI == 4 ? (wave & 3) : 0, hh = TI == 4 ? (wave >> 2) : wave;
        f32x16 acc[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        // ---- stage 0 of this tile (and, behind the barrier, its tables) has landed?  Younger than it: stage 1 (NF) and, from the
        // second tile on, the previous tile's output store (1)
        if (first) VPHO_WAIT_VMH(NF); else VPHO_WAIT_VMH(NF + 1);
        VPHO_BARRIER_LDS_ONLY();                                    // (not __syncthreads(): its fence would wait for the previous tile's store)
        if (first) VPHO_STAMP_AT(2);
        VPHO_PRIO_MAIN();
        for (int kt = 0; kt < NK; ++kt) {
            const int buf = kt & 1;
            const float* As = smem + buf * STAGE + (hh * 32 * TI + li) * HB_K;
            const float* Bs = smem + buf * STAGE + 256 * HB_K + (rg * 32 + li) * HB_K;
            f32x4 bq, av[TI];
            auto frags = [&](int kk) {
                const int ch = ((2 * kk + lh) ^ sw) * 4;
                bq = *reinterpret_cast<const f32x4*>(Bs + ch);
#pragma unroll
                for (int i = 0; i < TI; ++i) av[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * HB_K + ch);
            };
            auto mfmas = [&]() {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TI; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bq[q], acc[i], 0, 0, 0);
            };
#pragma unroll
            for (int kk = 0; kk < NKK - 1; ++kk) { frags(kk); mfmas(); }
            frags(NKK - 1);
            if (kt + 1 < NK) {
                VPHO_SYNC_LDS_DMA();                                // every wave has its fragments of stage `buf`; stage kt + 1 has landed
                if (kt + 2 < NK) fill(buf, kt + 2, n);
            }
            mfmas();
        }
        __syncthreads();                                            // both stage buffers are free
        VPHO_PRIO_REST();
        if (first) VPHO_STAMP_AT(3);
        // ---- next tile: tables into registers, then its first two stages, BEFORE this tile's epilogue
        const int en = n, er0 = r0, erows = rows;
        b += G;
        const bool more = b < total;
        f32x4 tvn = {0.f, 0.f, 0.f, 0.f};
        if (more) {
            tile_of(b, n, r0, rows);
            tvn = load_tables(n, r0, rows);
            set_poff(r0, rows);
            fill(0, 0, n);
            fill(1, 1, n);
        }
        // ---- epilogue: hidden unit j = 32*TI*hh + 32*i + (e&3) + 8*(e>>2) + 4*lh on the register, hypothesis on the lane; a row's 256 hidden
        // units are summed as EIGHT partial sums of 32 in ascending order, whatever the tile kind (see head_tile)
        const int lrow_out = rg * 32 + li, row = er0 + lrow_out;
        const bool live = row < a.R;
        // (the lane's first hidden unit as an OPAQUE per-tile value: left visible, the compiler hoists the 64 loop-invariant LDS addresses
        // of this epilogue out of the tile loop and spills them: 45 registers of scratch)
        int j0 = 32 * TI * hh + 4 * lh;
        asm volatile("" : "+v"(j0));
        const float* cim = a.cimg + (long long)(live ? row / a.S : 0) * a.NH + en * 256 + j0;
        const float* cbl = Cb + (live ? row / a.S - er0 / a.S : 0) * CB_LD + j0;
        const float* ebl = Eb + j0 * 4;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = 32 * i + (e & 3) + 8 * (e >> 2);      // + j0
                const f32x4 t = *reinterpret_cast<const f32x4*>(ebl + j * 4);
                float h = acc[i][e] + (CB ? cbl[j] : cim[j]) + t[0];
                h = h > 0.f ? h : 0.f;
                o0 += h * t[1]; o1 += h * t[2]; o2 += h * t[3];
            }
            o0 += __shfl_xor(o0, 32); o1 += __shfl_xor(o1, 32); o2 += __shfl_xor(o2, 32);
            if (lh == 0) {
                float* ob = Ob + ((hh * TI + i) * 128 + lrow_out) * 4;
                ob[0] = o0; ob[1] = o1; ob[2] = o2;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (first) VPHO_STAMP_AT(5);
        VPHO_BARRIER_LDS_ONLY();                                    // partial sums complete; Eb / Cb are no longer read (no fence: the next tile's fills are in flight)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const float b2v = tq < 3 * erows ? Cb[B2_AT + tq % 3] : 0.f;
        VPHO_BARRIER_LDS_ONLY();                                    // ... b2 of THIS tile read before the tables of the next one overwrite it
        if (more) store_tables(tvn);
        // ---- one output per thread: thread t -> (row t / 3, component t % 3); every thread issues the store (out of range when idle)
        {
            const int rl = tq / 3, dd = tq - 3 * rl, orow = er0 + rl;
            int off = -1;
            float sv = 0.f;
            if (rl < erows && orow < a.R) {
                float acc2 = Ob[rl * 4 + dd];
#pragma unroll
                for (int part = 1; part < 8; ++part) acc2 += Ob[(part * 128 + rl) * 4 + dd];
                sv = (acc2 + b2v) / inv_std;
                if (a.rhs_mode) {
                    // score_eval_wrapper's nan_to_num (score_based_model.py:65-72), entry by entry; see head_tile
                    if (sv != sv) { sv = 0.f; ++nans; } else if (fabsf(sv) == INFINITY) sv = 0.f;
                    sv = 0.f - coef * sv;
                }
                off = (int)(((long long)orow * a.D + en * 3 + dd) * 4);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv), orr, off, 0, 0);
        }
        if (first) VPHO_STAMP_AT(4);
        first = false;
        return more;
    };
    while (true) {
        const bool more = rows == 128 ? tile(std::integral_constant<int, 4>{}) : tile(std::integral_constant<int, 1>{});
        if (!more) break;
    }
    if (nans) atomicAdd(a.nan_count, nans);
    VPHO_STAMP_WRITE(head, blockIdx.x);
}

// --------------------------------------------------------------------------------------------- score head, split-bf16 products (opt-in)
// VPHO_SCORE_MFMA=bf16x6 | bf16x9 (not the default; the fp32-MFMA kernel above is the product path and the one every parity claim is
// made on).  gfx950 multiplies bf16 on the matrix cores 16 x faster than fp32, and a product of two fp32 numbers can be assembled
// from bf16 products without giving up fp32 accuracy: every fp32 x is EXACTLY x = h + m + l with h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m) (3 x 8 significant bits), each bf16 x bf16 product is exact in fp32, and the MFMA accumulates in fp32 like the
// fp32 instruction does.  bf16x9 adds all nine cross products (nothing dropped: same operands, same accumulator type, 9/16 of the
// matrix-core time); bf16x6 drops m*l, l*m, l*l, each below 2^-24 |x||y| -- the size of one fp32 rounding.  The weights are split
// once on the host ([n][3][256][256] bf16), the activations (P2, fp32 in LDS) as the fragments are read: 8 values -> 3 x 8 bf16 with
// v_cvt_pk_bf16_f32 + two subtractions per level, issued in the shadow of the MFMAs.  Error study: scripts/split_error_study.py.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int SP_K = 16;                                    // k per LDS stage = one 32x32x16 MFMA step

__device__ __forceinline__ void split3(const f32x4& x0, const f32x4& x1, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? x0[e] : x1[e - 4];
        const __bf16 hb = (__bf16)x;
        const float r1 = x - (float)hb;                     // exact: |r1| <= ulp_bf16(x) / 2 needs <= 16 bits
        const __bf16 mb = (__bf16)r1;
        const float r2 = r1 - (float)mb;                    // exact, <= 8 significant bits left
        h[e] = hb; m[e] = mb; l[e] = (__bf16)r2;
    }
}

template <int TI, int TERMS>
__device__ __forceinline__ void head_tile_split(const HeadArgs& a, float* smem, const int n, const int r0) {
    constexpr int ROWS = TI == 4 ? 128 : 32, PARTS = TI == 4 ? 2 : 8;
    constexpr int W_PLANE = 256 * SP_K / 2;                 // floats per plane and stage ([256][16] bf16)
    constexpr int STAGE = 3 * W_PLANE + 128 * SP_K;         // floats per stage: three weight planes | [128][16] fp32 activations
    float* Eb = smem + 2 * STAGE;
    float* Ob = Eb + 256 * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int rg = TI == 4 ? (wave & 3) : 0, hh = TI == 4 ? (wave >> 2) : wave;
    if (tid < 256) {   // epilogue table
        f32x4 e;
        e[0] = a.ct[n * 256 + tid];
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(a.w2 + (long long)(n * 256 + tid) * 4);
        e[1] = w2[0]; e[2] = w2[1]; e[3] = w2[2];
        *reinterpret_cast<f32x4*>(Eb + tid * 4) = e;
    }
    // weight planes: one wave instruction = 64 x 16 B = 32 rows of 32 B (16 bf16): 8 waves fill the 256 rows of a plane in one
    // pass; the two 16-B chunks of row r are exchanged when (r >> 3) & 1, so that 16 consecutive rows read at one logical chunk hit
    // 16 distinct 16-B slots of the two 256-B bank rows they span
    const int wrow = wave * 32 + (lane >> 1);
    const int wchunk = (lane & 1) ^ ((wrow >> 3) & 1);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(a.w1p_split) + (long long)n * 3 * 256 * 256, 0, 3 * 256 * 256 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p2), 0, 0xFFFFFFF0u, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    int woff[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) woff[p] = ((p * 256 + wrow) * 256 + wchunk * 8) * 2;
    // activations: [128][16] fp32, rows of 64 B (4 chunks), chunk index XOR (row >> 2) & 3 (as in head_tile with HB_K = 16)
    const int prow = wave * 16 + (lane >> 2);
    const int pkq = (lane & 3) ^ ((prow >> 2) & 3);
    const int pr_row = r0 + prow;
    const int poff = (pr_row < a.R && prow < ROWS) ? (int)(((unsigned)pr_row * 256u + 4u * (unsigned)pkq) * 4u) : -1;
    auto fill = [&](int buf, int kt) {
        float* Ws = smem + buf * STAGE + wave * 32 * (SP_K / 2);
        float* Ps = smem + buf * STAGE + 3 * W_PLANE + wave * 16 * SP_K;
#pragma unroll
        for (int p = 0; p < 3; ++p) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(Ws + p * W_PLANE), 16, woff[p], kt * SP_K * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (lds_ptr)Ps, 16, poff, kt * SP_K * 4, 0, 0);
    };

    f32x16 acc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    constexpr int NK = 256 / SP_K;
    const int a_sw = (li >> 3) & 1, b_sw = (li >> 2) & 3;
    // Register double buffer: the fragments of stage kt+1 are read from LDS and split while the MFMAs of stage kt issue (its fill
    // landed before barrier kt), one tile's three weight planes per group of products; the barrier of stage kt only orders "every
    // wave has read stage kt" (done one iteration earlier) before the refill of that buffer.
    struct Frag { bf16x8 a[TI][3]; bf16x8 bh, bm, bl; };
    auto read_b = [&](int buf, Frag& f) {
        const float* Bs = smem + buf * STAGE + 3 * W_PLANE + (rg * 32 + li) * SP_K;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(Bs + ((2 * lh) ^ b_sw) * 4);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(Bs + ((2 * lh + 1) ^ b_sw) * 4);
        split3(x0, x1, f.bh, f.bm, f.bl);
    };
    auto read_a = [&](int buf, int i, Frag& f) {
        const float* Wst = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            f.a[i][p] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(Wst + p * W_PLANE) + (hh * 32 * TI + 32 * i + li) * SP_K + ((lh ^ a_sw) * 8));
    };
    auto products = [&](int i, const Frag& f) {
        // small terms first
        if (TERMS == 9) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.bl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.bm, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.bl, acc[i], 0, 0, 0);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.bm, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.bm, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.bh, acc[i], 0, 0, 0);
    };
    Frag fr[2];
    fill(0, 0);
    fill(1, 1);
    VPHO_SYNC_LDS_DMA();
    read_b(0, fr[0]);
#pragma unroll
    for (int i = 0; i < TI; ++i) read_a(0, i, fr[0]);
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        const int buf = kt & 1;
        VPHO_SYNC_LDS_DMA();                                // all waves hold stage kt in registers; fill(kt+1) has landed
        if (kt + 2 < NK) fill(buf, kt + 2);
        Frag& cur = fr[kt & 1];
        Frag& nxt = fr[(kt + 1) & 1];
        if (kt + 1 < NK) read_b(buf ^ 1, nxt);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (kt + 1 < NK) read_a(buf ^ 1, i, nxt);
            products(i, cur);
        }
    }
    __syncthreads();

    // epilogue: identical to head_tile (same accumulator layout)
    const int lrow_out = rg * 32 + li;
    const int row = r0 + lrow_out;
    const bool live = row < a.R;
    const float* cim = a.cimg + (long long)(live ? row / a.S : 0) * a.NH + n * 256;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = 32 * TI * hh + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const f32x4 t = *reinterpret_cast<const f32x4*>(Eb + j * 4);
            float h = acc[i][e] + cim[j] + t[0];
            h = h > 0.f ? h : 0.f;
            o0 += h * t[1]; o1 += h * t[2]; o2 += h * t[3];
        }
    }
    o0 += __shfl_xor(o0, 32);
    o1 += __shfl_xor(o1, 32);
    o2 += __shfl_xor(o2, 32);
    if (lh == 0) {
        float* ob = Ob + (hh * ROWS + lrow_out) * 4;
        ob[0] = o0; ob[1] = o1; ob[2] = o2;
    }
    __syncthreads();
    if (tid < ROWS) {
        const int orow = r0 + tid;
        float inv_std = a.inv_std_den, coef = a.coef;
        float* outp = a.out;
        if (a.ctl && a.ctl_mode == 1) {
            inv_std = a.ctl->inv_std[a.stage]; coef = a.ctl->coef[a.stage];
            outp = a.kbase + (long long)kslot(a.out_slot, a.ctl->kswap) * a.n_el;
        }
        if (orow < a.R) {
            int nans = 0;
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                float acc2 = Ob[tid * 4 + dd];
#pragma unroll
                for (int part = 1; part < PARTS; ++part) acc2 += Ob[(part * ROWS + tid) * 4 + dd];
                float sv = (acc2 + a.b2[n * 3 + dd]) / inv_std;
                if (a.rhs_mode) {
                    // score_eval_wrapper (score_based_model.py:65-72): nan_to_num(nan=0, posinf=0, neginf=0) on the RHS evaluations of the
                    // solve -- there, when ANY entry of the evaluation is NaN; here entry by entry: the same result whenever a NaN is
                    // present, and +-inf without any NaN has no defined outcome in the reference (scipy's controller never recovers).
                    // The bare score (vpho_score_eval, the final denoise evaluation :100) is NOT guarded, as in the reference.
                    if (sv != sv) { sv = 0.f; ++nans; } else if (fabsf(sv) == INFINITY) sv = 0.f;
                    sv = 0.f - coef * sv;
                }
                outp[(long long)orow * a.D + n * 3 + dd] = sv;
            }
            if (nans) atomicAdd(a.nan_count, nans);
        }
    }
}

template <int TERMS>
__global__ __launch_bounds__(512) void score_head_split_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    if (ctl_skip(a.ctl, a.ctl_mode)) return;
    const int b = blockIdx.x, n_full = a.nheads * a.full_tiles;
    if (b < n_full) {
        head_tile_split<4, TERMS>(a, smem, b / a.full_tiles, (b % a.full_tiles) * 128);
    } else {
        const int q = b - n_full;
        head_tile_split<1, TERMS>(a, smem, q / a.tail_tiles, a.full_tiles * 128 + (q % a.tail_tiles) * 32);
    }
}

// --------------------------------------------------------------------------------------------- RK stage algebra
// X[r][0..Dp) = (float)(y + h * sum_j c_j K_j), pad columns zero; optionally also stores the fp64 sum to ynew
__global__ void stage_input_kernel(const double* __restrict__ y, KSlots ks, long long n_el, int D, int Dp,
                                   LinComb lc, float* __restrict__ X, double* __restrict__ ynew,
                                   const RkCtl* __restrict__ ctl, int ctl_mode, const double* ybuf0, const double* ybuf1) {
    if (ctl_skip(ctl, ctl_mode)) return;
    if (ctl && ctl_mode == 1) lc.h = ctl->h;                                    // select_initial_step: y0 + h0 * direction * f0
    if (ctl && ctl_mode == 2) y = ctl->parity ? ybuf1 : ybuf0;                  // denoise step: the accepted state
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long R = n_el / D;
    if (i >= R * Dp) return;
    const long long r = i / Dp;
    const int c = (int)(i - r * Dp);
    if (c >= D) { X[i] = 0.f; return; }
    const long long e = r * D + c;
    double s = 0.0;
    for (int j = 0; j < lc.n; ++j) s += (double)ks.p[j][e] * lc.c[j];
    const double v = y[e] + s * lc.h;
    X[i] = (float)v;
    if (ynew) ynew[e] = v;
}

__global__ void f32_to_state_kernel(const float* __restrict__ x, long long n_el, int D, int Dp, double* __restrict__ y, float* __restrict__ X) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long R = n_el / D;
    if (i >= R * Dp) return;
    const long long r = i / Dp;
    const int c = (int)(i - r * Dp);
    if (c >= D) { X[i] = 0.f; return; }
    const float v = x[r * D + c];
    y[r * D + c] = (double)v;
    X[i] = v;
}

// mode 0: (y/scale)^2, scale = atol+|y|rtol; 1: (Ka/scale)^2; 2: ((Kb-Ka)/scale)^2; 3: RK45 error with scale from (y, ynew)
struct NormArgs {
    const double* y; const double* ynew; const float* Ka; const float* Kb; KSlots ks; long long n_el;
    double rtol, atol, h; double E[7]; int mode; double* partial;
    const RkCtl* ctl; const float* kbase; const double *ybuf0, *ybuf1;      // mode 3 inside a controller-driven attempt
};
__global__ __launch_bounds__(256) void norm_partial_kernel(const NormArgs a) {
    __shared__ double red[256];
    double s = 0.0;
    if (ctl_skip(a.ctl, 1)) return;
    KSlots ks = a.ks;
    const double *yv = a.y, *yn = a.ynew;
    double h = a.h;
    if (a.ctl) {
        const int par = a.ctl->parity, sw = a.ctl->kswap;
        yv = par ? a.ybuf1 : a.ybuf0; yn = par ? a.ybuf0 : a.ybuf1; h = a.ctl->h;
#pragma unroll
        for (int j = 0; j < 7; ++j) ks.p[j] = a.kbase + (long long)kslot(j, sw) * a.n_el;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n_el; i += (long long)gridDim.x * 256) {
        double v, scale;
        if (a.mode == 3) {
            double e = 0.0;
            for (int j = 0; j < 7; ++j) e += (double)ks.p[j][i] * a.E[j];
            v = e * h;
            scale = a.atol + fmax(fabs(yv[i]), fabs(yn[i])) * a.rtol;
        } else {
            scale = a.atol + fabs(yv[i]) * a.rtol;
            v = a.mode == 0 ? yv[i] : (a.mode == 1 ? (double)a.Ka[i] : (double)a.Kb[i] - (double)a.Ka[i]);
        }
        const double q = v / scale;
        s += q * q;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[blockIdx.x] = red[0];
}
// fixed-order (deterministic) tree sum of the per-block partials
__global__ __launch_bounds__(256) void norm_final_kernel(const double* __restrict__ partial, int n, double* __restrict__ out,
                                                         const RkCtl* __restrict__ ctl) {
    __shared__ double red[256];
    if (ctl_skip(ctl, 1)) return;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// dense output (scipy RkDenseOutput): y_old + h * (K^T P) . [x, x^2, x^3, x^4] for every t_eval stamp inside the step
constexpr int DENSE_MAX = 48;
struct DenseArgs { double P[7][4]; double p[DENSE_MAX][4]; int idx[DENSE_MAX]; int n; double h; };
__global__ void dense_kernel(const double* __restrict__ y_old, KSlots ks, long long n_el, int D,
                             DenseArgs da, void* __restrict__ xs, int is_f64, int num_steps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_el) return;
    double q[4];
    for (int m = 0; m < 4; ++m) {
        double s = 0.0;
        for (int j = 0; j < 7; ++j) s += (double)ks.p[j][i] * da.P[j][m];
        q[m] = s;
    }
    const double y0 = y_old[i];
    const long long r = i / D;
    const int c = (int)(i - r * D);
    for (int st = 0; st < da.n; ++st) {
        double acc = 0.0;
        for (int m = 0; m < 4; ++m) acc += q[m] * da.p[st][m];
        const double v = da.h * acc + y0;
        const long long o = (r * num_steps + da.idx[st]) * D + c;
        if (is_f64) reinterpret_cast<double*>(xs)[o] = v; else reinterpret_cast<float*>(xs)[o] = (float)v;
    }
}

// x = y + (0 - g^2 * grad) * step  (fp32 product, fp64 add; score_based_model.py:95-104)
__global__ void denoise_kernel(const double* __restrict__ y, const float* __restrict__ grad, long long n_el, float g, float step,
                               void* __restrict__ x, int is_f64, const RkCtl* __restrict__ ctl, const double* ybuf0, const double* ybuf1) {
    if (ctl_skip(ctl, 2)) return;
    if (ctl) y = ctl->parity ? ybuf1 : ybuf0;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_el) return;
    const float drift = 0.f - (g * g) * grad[i];
    const double v = y[i] + (double)(drift * step);
    if (is_f64) reinterpret_cast<double*>(x)[i] = v; else reinterpret_cast<float*>(x)[i] = (float)v;
}

// --------------------------------------------------------------------------------------------- device-side step controller
// scipy.integrate._ivp: select_initial_step (common.py), RungeKutta._step_impl (rk.py) and the t_eval bookkeeping of
// solve_ivp (ivp.py), as one-thread kernels between the stage kernels.  Same formulas as the host loop in vpho_ode_sample
// (which stays available: VPHO_RK_HOST=1); pow() / powf() are the device library's, i.e. step sizes can differ from the
// host's in the last bit.
struct RkSetup { double T0, tf, rtol, atol, g_scale; int num_steps, log_cap; double *te, *dense_p, *log; };

// float32 sigma of ve_marginal_prob on a float32 time (torch: 0.01 * 5000.0 ** t_f32, a float32 power then a float32 product).  The power
// is taken in double and rounded once: the device library's powf is good to a few ulp only (a 2-4 ulp sigma = 2.4-5e-7 of every score,
// the 2.1 x / 3.6 x of the fp64 sampler referee, round 5); the correctly rounded float32 power is what torch's CPU kernel returns.
__device__ inline float dev_sigma_f32(float t) { return (float)SIGMA_MIN * (float)pow(SIGMA_MAX / SIGMA_MIN, (double)t); }
// Scalars of one RHS evaluation at time t.  Two sigmas, as in the reference (score_based_model.py:74-83, sde.py:15-24):
//   * the score's own division uses std = ve_marginal_prob(time tensor): the time tensor is float32 (torch.ones(bs) * t), so
//     sigma_min * (sigma_max / sigma_min) ** t is a FLOAT32 power;
//   * the drift coefficient 0.5 g(t)^2 comes from sde_coeff(torch.tensor(t)) with t the solver's np.float64 time: a 0-d float64 tensor,
//     sigma and g in FLOAT64, the product 0.5 g^2 cast to float32 once (numpy's value-based casting against the f32 score).
// Rounds 1-4 took g from the float32 sigma as well: a relative error of ~3e-7 in every stage's coefficient -- what the fp64 sampler
// referee (oracle/sampler_fp64.py, round 5) found as 2.1 x (max) / 3.6 x (rms) the reference arithmetic's distance from the exact scheme.
// Only the solver's very first call hands the wrapper a Python float (-> float32 tensor): eval_rhs(..., first_call = true) on the host.
__device__ inline void ctl_stage_scalars(RkCtl* c, int i, double t) {
    const float tf = (float)t;
    const float sg = dev_sigma_f32(tf);
    const double g = SIGMA_MIN * pow(SIGMA_MAX / SIGMA_MIN, t) * c->g_scale;
    c->ts[i] = tf; c->inv_std[i] = sg + 1e-7f; c->coef[i] = (float)(0.5 * g * g);
}

__global__ void rk_setup_kernel(RkCtl* c, RkSetup p) {
    if (threadIdx.x || blockIdx.x) return;
    c->t = p.T0; c->T0 = p.T0; c->tf = p.tf; c->rtol = p.rtol; c->atol = p.atol; c->max_step = 10.0; c->g_scale = p.g_scale;
    c->h_abs = c->h = c->t_new = c->err = c->dense_h = 0.0;
    c->te = p.te; c->dense_p = p.dense_p; c->log = p.log;
    c->done = c->accepted = c->rejected = c->parity = c->kswap = c->status = 0;
    c->next_idx = 0; c->num_steps = p.num_steps; c->dense_first = c->dense_last = c->dense_parity = c->dense_kswap = 0;
    c->n_accepted = c->n_rejected = c->nfev = c->n_log = c->n_attempts = 0; c->log_cap = p.log_cap;
    // t_eval = np.linspace(T0, eps, num_steps)
    const int div = p.num_steps > 1 ? p.num_steps - 1 : 1;
    const double step = (p.tf - p.T0) / div;
    for (int i = 0; i < p.num_steps; ++i) p.te[i] = (double)i * step + p.T0;
    if (p.num_steps > 1) p.te[p.num_steps - 1] = p.tf;
}

// after d0 = ||y0/scale||, d1 = ||f0/scale||: h0 and the scalars of the probe evaluation at t0 + h0 * direction
__global__ void rk_init1_kernel(RkCtl* c, long long n_el) {
    if (threadIdx.x || blockIdx.x) return;
    const double rn = sqrt((double)n_el);
    const double d0 = sqrt(c->d[0]) / rn, d1 = sqrt(c->d[1]) / rn;
    const double interval = fabs(c->tf - c->t);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h0 = fmin(h0, interval);
    c->d[0] = d0; c->d[1] = d1; c->d[3] = h0;
    c->h = h0 * -1.0;
    ctl_stage_scalars(c, 0, c->t + h0 * -1.0);
}

__global__ void rk_init2_kernel(RkCtl* c, long long n_el) {
    if (threadIdx.x || blockIdx.x) return;
    const double h0 = c->d[3], d1 = c->d[1];
    const double d2 = (sqrt(c->d[2]) / sqrt((double)n_el)) / h0;
    const double interval = fabs(c->tf - c->t);
    const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
    c->h_abs = fmin(fmin(100 * h0, h1), fmin(interval, c->max_step));
    c->nfev = 2;
}

struct RkC { double C[6]; };
__global__ void rk_begin_kernel(RkCtl* c, RkC k) {
    if (threadIdx.x || blockIdx.x) return;
    if (c->done) { c->dense_first = c->dense_last = 0; return; }
    const double direction = -1.0;
    const double t = c->t;
    const double min_step = 10 * fabs(nextafter(t, direction * INFINITY) - t);
    double h_abs = c->h_abs;
    if (!c->rejected) {                                   // first attempt of a step: clamp (rk.py _step_impl head)
        if (h_abs > c->max_step) h_abs = c->max_step; else if (h_abs < min_step) h_abs = min_step;
    } else if (h_abs < min_step) {                        // a retry may not go below the resolution of t
        c->status = 1; c->done = 1; c->dense_first = c->dense_last = 0;
        return;
    }
    double h = h_abs * direction;
    double t_new = t + h;
    if (direction * (t_new - c->tf) > 0) t_new = c->tf;
    h = t_new - t;
    c->h = h; c->h_abs = fabs(h); c->t_new = t_new;
    for (int s = 1; s < 6; ++s) ctl_stage_scalars(c, s - 1, t + k.C[s] * h);
    ctl_stage_scalars(c, 5, t + h);
    ++c->n_attempts;
}

__global__ void rk_end_kernel(RkCtl* c, const double* __restrict__ norm2, long long n_el) {
    if (threadIdx.x || blockIdx.x) return;
    if (c->done) return;
    const double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10.0, ERR_EXP = -1.0 / 5.0;
    const double err = sqrt(*norm2) / sqrt((double)n_el);
    c->err = err;
    c->nfev += 6;
    const double t = c->t, h = c->h;
    const bool acc = err < 1;
    if (c->n_log < c->log_cap) { double* p = c->log + 4 * c->n_log; p[0] = t; p[1] = h; p[2] = err; p[3] = acc ? 1.0 : 0.0; }
    ++c->n_log;
    if (acc) {
        double factor = err == 0 ? MAX_FACTOR : fmin(MAX_FACTOR, SAFETY * pow(err, ERR_EXP));
        if (c->rejected) factor = fmin(1.0, factor);
        c->h_abs *= factor;
        ++c->n_accepted;
        // dense output on the stamps inside (t_new, t]  (decreasing time: stamps >= t_new)
        const double t_new = c->t_new;
        int ni = c->next_idx;
        c->dense_first = ni;
        while (ni < c->num_steps && c->te[ni] >= t_new) {
            const double x = (c->te[ni] - t) / h;
            double* pp = c->dense_p + 4 * ni;
            pp[0] = x; pp[1] = x * x; pp[2] = pp[1] * x; pp[3] = pp[2] * x;
            ++ni;
        }
        c->dense_last = c->next_idx = ni;
        c->dense_h = h; c->dense_parity = c->parity; c->dense_kswap = c->kswap;
        // accept: y <- y_new, f <- f_new (first-same-as-last)
        c->parity ^= 1; c->kswap ^= 1; c->t = t_new; c->rejected = 0; c->accepted = 1;
        if (t_new == c->tf) c->done = 1;
    } else {
        c->h_abs *= fmax(MIN_FACTOR, SAFETY * pow(err, ERR_EXP));
        c->rejected = 1; c->accepted = 0;
        ++c->n_rejected;
        c->dense_first = c->dense_last = 0;
    }
}

// dense output of the attempt just judged (RkDenseOutput): all of its stamps in one launch, arguments from the controller
struct DenseP { double P[7][4]; };
__global__ void dense_ctl_kernel(const RkCtl* __restrict__ c, const double* ybuf0, const double* ybuf1, const float* __restrict__ kbase,
                                 long long n_el, int D, DenseP dp, void* __restrict__ xs, int is_f64, int num_steps) {
    const int first = c->dense_first, last = c->dense_last;
    if (first >= last) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_el) return;
    const double* y_old = c->dense_parity ? ybuf1 : ybuf0;
    const int sw = c->dense_kswap;
    double q[4];
    for (int m = 0; m < 4; ++m) {
        double s = 0.0;
        for (int j = 0; j < 7; ++j) s += (double)kbase[(long long)kslot(j, sw) * n_el + i] * dp.P[j][m];
        q[m] = s;
    }
    const double y0 = y_old[i], h = c->dense_h;
    const long long r = i / D;
    const int cc = (int)(i - r * D);
    for (int st = first; st < last; ++st) {
        const double* p = c->dense_p + 4 * st;
        double acc = 0.0;
        for (int m = 0; m < 4; ++m) acc += q[m] * p[m];
        const double v = h * acc + y0;
        const long long o = (r * num_steps + st) * D + cc;
        if (is_f64) reinterpret_cast<double*>(xs)[o] = v; else reinterpret_cast<float*>(xs)[o] = (float)v;
    }
}

// --------------------------------------------------------------------------------------------- host side
inline long long align_up(long long v) { return (v + 255) / 256 * 256; }

struct Workspace {
    float *cimg, *ct, *X, *P1, *P2, *K, *tmp;
    double *y, *ynew, *partial, *result;
    int* nan_count;
    RkCtl* ctl; double *te, *dense_p, *log;
    long long bytes;
};
constexpr int CTL_MAX_STEPS = 1024, CTL_LOG_CAP = 512;

Workspace carve(const vpho_score_weights& w, int bs, int S, char* base) {
    const long long R = (long long)bs * S, NH = (long long)w.nheads * 256;
    long long off = 0;
    Workspace ws;
    auto take = [&](long long b) { char* p = base ? base + off : nullptr; off += align_up(b); return p; };
    ws.cimg = (float*)take(bs * NH * 4);
    ws.ct = (float*)take(8 * NH * 4);
    ws.X = (float*)take(R * w.Dp * 4);
    ws.P1 = (float*)take(R * 256 * 4);
    ws.P2 = (float*)take(R * 256 * 4);
    ws.K = (float*)take(7 * R * w.D * 4);
    ws.tmp = (float*)take(R * w.D * 4);
    ws.y = (double*)take(R * w.D * 8);
    ws.ynew = (double*)take(R * w.D * 8);
    ws.partial = (double*)take(1024 * 8);
    ws.result = (double*)take(64);
    ws.nan_count = (int*)take(64);
    ws.ctl = (RkCtl*)take(sizeof(RkCtl));
    ws.te = (double*)take(CTL_MAX_STEPS * 8);
    ws.dense_p = (double*)take(CTL_MAX_STEPS * 4 * 8);
    ws.log = (double*)take(CTL_LOG_CAP * 4 * 8);
    ws.bytes = off;
    return ws;
}

int check_weights(const vpho_score_weights* w) {
    VPHO_REQUIRE(w != nullptr, "score weights: null");
    VPHO_REQUIRE(w->nheads > 0 && w->D == w->nheads * 3 && w->Dp >= w->D && w->Dp % 4 == 0, "score weights: D=%d Dp=%d nheads=%d inconsistent", w->D, w->Dp, w->nheads);
    VPHO_REQUIRE(w->t_W && w->t_w && w->t_b && w->pe0_w && w->pe0_b && w->pe2_w && w->pe2_b && w->w1_t && w->w1_p && w->w1_f && w->b1 && w->w2 && w->b2, "score weights: null tensor");
    return 0;
}

int linear(const float* x, int rows, int cin, const float* wt, const float* bias, int cout, float slope, float* y, hipStream_t s) {
    vpho_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.w = wt; d.bias = bias; d.y = y;
    d.N = rows; d.H = 1; d.W = 1; d.Cin = cin; d.x_ld = cin; d.Cout = cout; d.KH = 1; d.KW = 1; d.stride = 1;
    d.OH = 1; d.OW = 1; d.y_sx = cout; d.y_sy = cout; d.y_sn = cout; d.in_slope = 1.f; d.out_slope = slope;
    return vpho_conv2d_nhwc_f32(&d, s);
}

float sigma_f32(float t) { return (float)SIGMA_MIN * (float)std::pow(SIGMA_MAX / SIGMA_MIN, (double)t); }      // see dev_sigma_f32

struct Ctx {
    const vpho_score_weights* w; Workspace ws; int bs, S; long long R, n_el; int NH; hipStream_t s;
};

int prepare_cimg(Ctx& c, const float* feat_img) {
    // cimg[img][o] = b1[o] + feat[img] . w1_f[o]
    return linear(feat_img, c.bs, 1024, c.w->w1_f, c.w->b1, c.NH, 1.f, c.ws.cimg, c.s);
}

// X (R x Dp, fp32) -> out (R x D)
// time embeddings of up to 8 evaluation times in one launch -> ct slots 0..n-1
int embed_times(Ctx& c, const float* ts, int n, int ctl_mode = 0) {
    TimeList tl;
    for (int i = 0; i < 8; ++i) tl.t[i] = (ts && i < n) ? ts[i] : 0.f;
    hipLaunchKernelGGL(time_embed_kernel, dim3((c.NH + 255) / 256, n), dim3(256), 0, c.s, *c.w, tl, c.NH, c.ws.ct,
                       ctl_mode ? c.ws.ctl : (const RkCtl*)nullptr, ctl_mode);
    return vpho::check_launch("time_embed_kernel");
}

// ct_slot < 0: embed t now into slot 0; otherwise slot ct_slot was filled by embed_times for exactly this t
// Controller-driven calls (ctl_mode 1: stage `stage` of an attempt, result into logical stage slot `out_slot`; 2: the final
// denoise evaluation, which runs only once the solve is done) take their run-time scalars from c.ws.ctl.
struct CtlCall { int mode = 0, stage = 0, out_slot = 0, write_ynew = 0; };

int eval_net(Ctx& c, const float* X, float t, int rhs_mode, float coef, float* out, int ct_slot = -1,
             const LinComb* lc = nullptr, const double* y = nullptr, double* ynew = nullptr, const KSlots* ks = nullptr,
             CtlCall cc = CtlCall()) {
    if (ct_slot < 0) {
        if (int e = embed_times(c, &t, 1, cc.mode == 2 ? 2 : 0)) return e;
        ct_slot = 0;
    }
    {
        PoseEncArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.X = X; pa.Dp = c.w->Dp; pa.w0 = c.w->pe0_w; pa.b0 = c.w->pe0_b; pa.w2 = c.w->pe2_w; pa.b2 = c.w->pe2_b;
        pa.out = c.ws.P2; pa.R = (int)c.R;
        if (lc) { pa.use_lc = 1; pa.lc = *lc; pa.y = y; if (ks) pa.ks = *ks; pa.D = c.w->D; pa.ynew = ynew; }
        pa.ctl = cc.mode ? c.ws.ctl : nullptr; pa.ctl_mode = cc.mode; pa.write_ynew = cc.write_ynew; pa.kbase = c.ws.K; pa.n_el = c.n_el;
        pa.ybuf0 = c.ws.y; pa.ybuf1 = c.ws.ynew;
        const int K1 = (pa.Dp + 31) / 32 * 32;
        const char* pe_env = getenv("VPHO_PE_RING");                            // round 3's LDS-ring kernel (A/B measurements, read per call)
        const bool pe_ring = pe_env && atoi(pe_env);
        const dim3 pe_grid((unsigned)((c.R + PE_ROWS - 1) / PE_ROWS));
        vpho::ProfScope pe_prof(vpho::PROF_POSE_ENC, c.s, 2.0 * (double)c.R * 256.0 * (K1 + 256.0), 0.0);
        if (pe_ring) {
            const size_t pe_lds = (size_t)(PE_NST * PE_STAGE + PE_ROWS * PE_H_LD + 512 + PE_ROWS * (K1 + 4)) * sizeof(float);
            VPHO_DYN_LDS(pose_encoder_kernel, 150 * 1024);
            VPHO_REQUIRE(pe_lds <= 150 * 1024, "pose encoder: input dimension %d too large for the LDS tile", pa.Dp);
            hipLaunchKernelGGL(pose_encoder_kernel, pe_grid, dim3(512), pe_lds, c.s, pa);
        } else {
            const size_t pe_lds = (size_t)(PE_ROWS * PE_H_LD + 512) * sizeof(float);            // 35 328 B
            VPHO_REQUIRE(K1 <= 128 && (pa.Dp & 3) == 0, "pose encoder: input dimension %d (padded %d) not supported (<= 128, multiple of 4)", c.w->D, pa.Dp);
            // 64-row workgroups above 8 192 rows, i.e. as soon as 32-row workgroups would be more than one per CU (measured, isolated launches, hand /
            // object network: 12 800 rows 35.4 / 33.1 -> 30.4 / 27.9 us, 16 384 rows 35.8 / 32.6 -> 30.7 / 26.8, 32 768 rows -- BASELINE cfg4 -- 67.7 / 58.7 ->
            // 57.8 / 47.8; at the README config's 6 400 rows the 32-row kernel wins, 18.7 against 28.9: 200 workgroups against 100 on 256 CUs).
            // VPHO_PE_ROWS=32 / 64 forces either (A/B aid, read per call; the two are bit-identical)
            const char* rows_env = getenv("VPHO_PE_ROWS");
            const bool rows64 = rows_env ? atoi(rows_env) == 64 : c.R > 8192;
            if (rows64) {
                const size_t lds64 = (size_t)(2 * PE_ROWS * PE_H_LD + 512) * sizeof(float);          // 68 608 B
                const dim3 grid64((unsigned)((c.R + 2 * PE_ROWS - 1) / (2 * PE_ROWS)));
                VPHO_DYN_LDS(pose_encoder_reg64_kernel<1>, lds64); VPHO_DYN_LDS(pose_encoder_reg64_kernel<2>, lds64);
                VPHO_DYN_LDS(pose_encoder_reg64_kernel<3>, lds64); VPHO_DYN_LDS(pose_encoder_reg64_kernel<4>, lds64);
                switch (K1 / 32) {
                    case 1: hipLaunchKernelGGL(pose_encoder_reg64_kernel<1>, grid64, dim3(512), lds64, c.s, pa); break;
                    case 2: hipLaunchKernelGGL(pose_encoder_reg64_kernel<2>, grid64, dim3(512), lds64, c.s, pa); break;
                    case 3: hipLaunchKernelGGL(pose_encoder_reg64_kernel<3>, grid64, dim3(512), lds64, c.s, pa); break;
                    default: hipLaunchKernelGGL(pose_encoder_reg64_kernel<4>, grid64, dim3(512), lds64, c.s, pa); break;
                }
            } else
            switch (K1 / 32) {
                case 1: hipLaunchKernelGGL(pose_encoder_reg_kernel<1>, pe_grid, dim3(512), pe_lds, c.s, pa); break;
                case 2: hipLaunchKernelGGL(pose_encoder_reg_kernel<2>, pe_grid, dim3(512), pe_lds, c.s, pa); break;
                case 3: hipLaunchKernelGGL(pose_encoder_reg_kernel<3>, pe_grid, dim3(512), pe_lds, c.s, pa); break;
                default: hipLaunchKernelGGL(pose_encoder_reg_kernel<4>, pe_grid, dim3(512), pe_lds, c.s, pa); break;
            }
        }
        if (int e = vpho::check_launch("pose_encoder_kernel")) return e;
    }
    VPHO_REQUIRE((double)c.R * 256.0 * 4.0 < 4.0e9, "score head: %lld hypothesis rows exceed the 32-bit buffer offsets", (long long)c.R);
    HeadArgs a;
    a.w1p = c.w->w1_p; a.p2 = c.ws.P2; a.cimg = c.ws.cimg; a.ct = c.ws.ct + (long long)ct_slot * c.NH; a.w2 = c.w->w2; a.b2 = c.w->b2;
    a.out = out; a.nan_count = c.ws.nan_count; a.R = (int)c.R; a.S = c.S; a.NH = c.NH; a.D = c.w->D;
    a.inv_std_den = sigma_f32(t) + 1e-7f; a.coef = coef; a.rhs_mode = rhs_mode;
    a.ctl = cc.mode ? c.ws.ctl : nullptr; a.ctl_mode = cc.mode; a.stage = cc.stage; a.out_slot = cc.out_slot; a.kbase = c.ws.K; a.n_el = c.n_el;
    // [2] stages | [256][4] epilogue table | per-image terms (3 x 257 of 1024 floats)
    size_t lds = (size_t)(2 * (256 + 128) * HB_K + 256 * 4 + 2 * 128 * 4) * sizeof(float);
    if (getenv("VPHO_HEAD_LDS")) lds = (size_t)atoi(getenv("VPHO_HEAD_LDS"));   // tuning aid: force 1 block/CU
    VPHO_DYN_LDS(score_head_kernel<true>, lds);
    VPHO_DYN_LDS(score_head_kernel<false>, lds);
    // Tile plan: 128-row tiles; when the last round of the launch would be less than half full, the rows beyond the last full round
    // become 32-row tail tiles (see head_tile).  VPHO_HEAD_TAIL=0: ordinary tiles only.
    static const int tail_on = getenv("VPHO_HEAD_TAIL") ? atoi(getenv("VPHO_HEAD_TAIL")) : 1;
    static int slots_of[64] = {0};                          // per device: one process may drive several GPUs of different sizes
    int dev = 0;
    VPHO_HIP(hipGetDevice(&dev));
    int& slots = slots_of[dev & 63];
    if (!slots) {
        hipDeviceProp_t prop;
        VPHO_HIP(hipGetDeviceProperties(&prop, dev));
        slots = 2 * prop.multiProcessorCount;               // two 57 KB workgroups per CU
    }
    const int tiles = (int)((c.R + 127) / 128), nheads = c.w->nheads;
    a.nheads = nheads; a.full_tiles = tiles; a.tail_tiles = 0;
    const long long total = (long long)tiles * nheads;
    if (tail_on && total > slots && total % slots != 0 && (total % slots) * 2 < slots) {
        const int fp = (int)(total / slots * slots / nheads);
        if (fp >= 1 && fp < tiles) { a.full_tiles = fp; a.tail_tiles = (int)((c.R - 128ll * fp + 31) / 32); }
    }
    a.w1p_split = (const unsigned short*)c.w->w1_p_split;
    if (c.w->w1_p_split && (c.w->split_terms == 6 || c.w->split_terms == 9)) {
        const size_t slds = (size_t)(2 * (3 * 256 * SP_K / 2 + 128 * SP_K) + 256 * 4 + 2 * 128 * 4) * sizeof(float);
        VPHO_DYN_LDS(score_head_split_kernel<6>, slds);
        VPHO_DYN_LDS(score_head_split_kernel<9>, slds);
        vpho::ProfScope prof(vpho::PROF_SCORE_HEAD, c.s, (double)c.R * c.w->nheads * (2.0 * 256 * 256 + 2.0 * 256 * 3));
        const dim3 grid((unsigned)(nheads * (a.full_tiles + a.tail_tiles)));
        if (c.w->split_terms == 6) hipLaunchKernelGGL(score_head_split_kernel<6>, grid, dim3(512), slds, c.s, a);
        else                       hipLaunchKernelGGL(score_head_split_kernel<9>, grid, dim3(512), slds, c.s, a);
        return vpho::check_launch("score_head_split_kernel");
    }
    {
        vpho::ProfScope prof(vpho::PROF_SCORE_HEAD, c.s, (double)c.R * c.w->nheads * (2.0 * 256 * 256 + 2.0 * 256 * 3));
        // per-image epilogue terms from an LDS copy when a 128-row tile spans at most 3 images (sample_num >= 64; VPHO_HEAD_CB=0: global loads, A/B aid -- same bits)
        const char* cb_s = getenv("VPHO_HEAD_CB");
        const int cb_env = cb_s ? atoi(cb_s) : 1;
        const dim3 grid((unsigned)(nheads * (a.full_tiles + a.tail_tiles)));
        // persistent kernel, OPT-IN (VPHO_HEAD_PERS=1; same bits): min(tiles, workgroup slots) workgroups walk the tiles.  Measured round 5
        // (profiles/r05_head_persistent_ab.txt): stand-alone 230-238 us against 232-234 for one workgroup per tile, the pipelined step
        // 29.9-30.1 ms against 29.5-29.6 -- with two workgroups per CU the other workgroup's matrix work already covers a tile's prologue,
        // and workgroups that hold their CU slot for the whole launch keep the other streams' kernels out.  Not the default.
        const char* pp = getenv("VPHO_HEAD_PERS");
        const int pers = pp ? atoi(pp) : 0;
        const bool cb = c.S >= 64 && cb_env;
        if (pers && cb && (double)c.R * c.w->D * 4.0 < 3.9e9 && (double)c.w->nheads * 256 * 256 * 4.0 < 3.9e9) {
            const size_t plds = (size_t)(2 * (256 + 128) * HB_K + 256 * 4 + 1024 + 8 * 128 * 4) * sizeof(float);     // 72 KB: two workgroups per CU
            VPHO_DYN_LDS(score_head_pers_kernel<true>, plds);
            const dim3 pgrid((unsigned)std::min<long long>((long long)grid.x, (long long)slots));
            hipLaunchKernelGGL(score_head_pers_kernel<true>, pgrid, dim3(512), plds, c.s, a);       // sample_num >= 64 (the LDS copy of the per-image terms); smaller: the one-tile kernel
            return vpho::check_launch("score_head_pers_kernel");
        }
        if (cb) hipLaunchKernelGGL(score_head_kernel<true>, grid, dim3(512), lds, c.s, a);
        else    hipLaunchKernelGGL(score_head_kernel<false>, grid, dim3(512), lds, c.s, a);
    }
    return vpho::check_launch("score_head_kernel");
}

// rhs(t, .) of the probability-flow ODE: 0 - f32(0.5 g(t)^2) * score     (score_based_model.py:74-83)
// first_call: fun(t0, y0), the one evaluation whose t is a Python float in scipy (-> float32 sigma); every other t is np.float64 (see ctl_stage_scalars)
int eval_rhs(Ctx& c, const float* X, double t, float* out, int ct_slot = -1, const LinComb* lc = nullptr,
             const double* y = nullptr, double* ynew = nullptr, const KSlots* ks = nullptr, bool first_call = false) {
    const float tf = (float)t;
    const double sigma = first_call ? (double)sigma_f32(tf) : SIGMA_MIN * std::pow(SIGMA_MAX / SIGMA_MIN, t);
    const double g = sigma * std::sqrt(2.0 * (std::log(SIGMA_MAX) - std::log(SIGMA_MIN)));
    const float coef = (float)(0.5 * g * g);
    return eval_net(c, X, tf, 1, coef, out, ct_slot, lc, y, ynew, ks);
}

double* pinned_slot() {
    static thread_local double* p = nullptr;
    if (!p) { if (hipHostMalloc((void**)&p, 64, hipHostMallocDefault) != hipSuccess) p = nullptr; }
    return p;
}

// sum of squares -> *out_dev (device); ctl: part of a controller-driven attempt
int enqueue_norm(Ctx& c, NormArgs na, double* out_dev, bool ctl = false) {
    const int nb = (int)std::min<long long>(1024, (c.n_el + 255) / 256);
    na.n_el = c.n_el; na.partial = c.ws.partial;
    na.ctl = ctl ? c.ws.ctl : nullptr; na.kbase = c.ws.K; na.ybuf0 = c.ws.y; na.ybuf1 = c.ws.ynew;
    hipLaunchKernelGGL(norm_partial_kernel, dim3(nb), dim3(256), 0, c.s, na);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, c.s, c.ws.partial, nb, out_dev, ctl ? c.ws.ctl : (const RkCtl*)nullptr);
    return vpho::check_launch("norm kernels");
}

int reduce_norm(Ctx& c, NormArgs na, double* value) {
    if (int e = enqueue_norm(c, na, c.ws.result)) return e;
    double* h = pinned_slot();
    VPHO_REQUIRE(h != nullptr, "hipHostMalloc failed");
    VPHO_HIP(hipMemcpyAsync(h, c.ws.result, 8, hipMemcpyDeviceToHost, c.s));
    VPHO_HIP(hipStreamSynchronize(c.s));
    *value = std::sqrt(*h) / std::sqrt((double)c.n_el);
    return 0;
}

const double RK_C[6] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1};
const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
const double RK_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

}  // namespace

extern "C" long long vpho_score_workspace_bytes(const vpho_score_weights* w, int bs, int S) {
    if (!w || bs <= 0 || S <= 0) return -1;
    return carve(*w, bs, S, nullptr).bytes;
}

extern "C" int vpho_score_eval(const vpho_score_weights* w, const float* feat_img, int bs, int S, const float* x, float t,
                               float* out, void* workspace, long long workspace_bytes, void* stream) {
    if (int e = check_weights(w)) return e;
    VPHO_REQUIRE(bs > 0 && S > 0 && feat_img && x && out && workspace, "vpho_score_eval: bad argument");
    Ctx c;
    c.w = w; c.bs = bs; c.S = S; c.R = (long long)bs * S; c.n_el = c.R * w->D; c.NH = w->nheads * 256; c.s = (hipStream_t)stream;
    c.ws = carve(*w, bs, S, (char*)workspace);
    VPHO_REQUIRE(workspace_bytes >= c.ws.bytes, "vpho_score_eval: workspace %lld < %lld bytes", workspace_bytes, c.ws.bytes);
    VPHO_HIP(hipMemsetAsync(c.ws.nan_count, 0, 4, c.s));
    if (int e = prepare_cimg(c, feat_img)) return e;
    const int nb = (int)((c.R * w->Dp + 255) / 256);
    hipLaunchKernelGGL(f32_to_state_kernel, dim3(nb), dim3(256), 0, c.s, x, c.n_el, w->D, w->Dp, c.ws.y, c.ws.X);
    return eval_net(c, c.ws.X, t, 0, 0.f, out);
}

namespace {

// ---- host-driven solve: the scalar controller runs on the host exactly as scipy's, one 8-byte D2H + sync per attempt ----
int ode_sample_host(Ctx& c, const float* feat_img, const float* init_x, double T0, double eps, int num_steps, double rtol, double atol,
                    void* xs_out, int xs_is_f64, void* x_out, int x_is_f64, vpho_ode_stats* st, double* step_log, int step_log_cap) {
    const vpho_score_weights* w = c.w;
    const long long n_el = c.n_el;
    const int D = w->D, Dp = w->Dp;
    const int nbX = (int)((c.R * Dp + 255) / 256), nbE = (int)((n_el + 255) / 256);
    KSlots ks;
    for (int j = 0; j < 7; ++j) ks.p[j] = c.ws.K + (long long)j * n_el;
    auto Kp = [&](int j) { return const_cast<float*>(ks.p[j]); };

    VPHO_HIP(hipMemsetAsync(c.ws.nan_count, 0, 4, c.s));
    if (int e = prepare_cimg(c, feat_img)) return e;
    double* y = c.ws.y;
    double* ynew = c.ws.ynew;
    hipLaunchKernelGGL(f32_to_state_kernel, dim3(nbX), dim3(256), 0, c.s, init_x, n_el, D, Dp, y, c.ws.X);

    // t_eval = np.linspace(T0, eps, num_steps)
    std::vector<double> te(num_steps);
    {
        const int div = num_steps > 1 ? num_steps - 1 : 1;
        const double step = (eps - T0) / div;
        for (int i = 0; i < num_steps; ++i) te[i] = (double)i * step + T0;
        if (num_steps > 1) te[num_steps - 1] = eps;
    }
    const double direction = -1.0, tf = eps, max_step = 10.0;
    double t = T0;
    int next_idx = 0;
    auto log_step = [&](double tt, double hh, double err, int acc) {
        if (step_log && st->n_log < step_log_cap) {
            double* p = step_log + 4 * st->n_log;
            p[0] = tt; p[1] = hh; p[2] = err; p[3] = acc;
        }
        ++st->n_log;
    };

    // f0 = fun(t0, y0); select_initial_step
    if (int e = eval_rhs(c, c.ws.X, t, Kp(0), -1, nullptr, nullptr, nullptr, nullptr, true)) return e;
    ++st->nfev;
    double h_abs;
    {
        NormArgs na;
        memset(&na, 0, sizeof(na));
        na.y = y; na.rtol = rtol; na.atol = atol;
        double d0, d1, d2;
        na.mode = 0;
        if (int e = reduce_norm(c, na, &d0)) return e;
        na.mode = 1; na.Ka = Kp(0);
        if (int e = reduce_norm(c, na, &d1)) return e;
        const double interval = std::fabs(tf - t);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = std::min(h0, interval);
        LinComb lc;
        memset(&lc, 0, sizeof(lc));
        lc.n = 1; lc.c[0] = 1.0; lc.h = h0 * direction;
        hipLaunchKernelGGL(stage_input_kernel, dim3(nbX), dim3(256), 0, c.s, y, ks, n_el, D, Dp, lc, c.ws.X, (double*)nullptr,
                           (const RkCtl*)nullptr, 0, (const double*)nullptr, (const double*)nullptr);
        if (int e = eval_rhs(c, c.ws.X, t + h0 * direction, Kp(1))) return e;
        ++st->nfev;
        na.mode = 2; na.Ka = Kp(0); na.Kb = Kp(1);
        if (int e = reduce_norm(c, na, &d2)) return e;
        d2 /= h0;
        double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0 * 1e-3) : std::pow(0.01 / std::max(d1, d2), 1.0 / 5.0);
        h_abs = std::min(std::min(100 * h0, h1), std::min(interval, max_step));
    }

    const double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10.0, ERR_EXP = -1.0 / 5.0;
    while (t != tf) {
        const double min_step = 10 * std::fabs(std::nextafter(t, direction * INFINITY) - t);
        if (h_abs > max_step) h_abs = max_step; else if (h_abs < min_step) h_abs = min_step;
        bool accepted = false, rejected = false;
        double h = 0, t_new = t;
        while (!accepted) {
            if (h_abs < min_step) { st->status = 1; return vpho::fail("vpho_ode_sample: step size underflow at t=%g", t); }
            h = h_abs * direction;
            t_new = t + h;
            if (direction * (t_new - tf) > 0) t_new = tf;
            h = t_new - t;
            h_abs = std::fabs(h);
            {   // the six stage times of this attempt are known up front: one embedding launch
                float ts[6];
                for (int s = 1; s < 6; ++s) ts[s - 1] = (float)(t + RK_C[s] * h);
                ts[5] = (float)(t + h);
                if (int e = embed_times(c, ts, 6)) return e;
            }
            for (int s = 1; s < 6; ++s) {
                LinComb lc;
                memset(&lc, 0, sizeof(lc));
                lc.n = s; lc.h = h;
                for (int j = 0; j < s; ++j) lc.c[j] = RK_A[s][j];
                if (int e = eval_rhs(c, nullptr, t + RK_C[s] * h, Kp(s), s - 1, &lc, y, nullptr, &ks)) return e;   // stage state formed in the pose encoder
            }
            {
                LinComb lc;
                memset(&lc, 0, sizeof(lc));
                lc.n = 6; lc.h = h;
                for (int j = 0; j < 6; ++j) lc.c[j] = RK_B[j];
                if (int e = eval_rhs(c, nullptr, t + h, Kp(6), 5, &lc, y, ynew, &ks)) return e;                    // also stores y_new (fp64)
            }
            st->nfev += 6;
            NormArgs na;
            memset(&na, 0, sizeof(na));
            na.mode = 3; na.y = y; na.ynew = ynew; na.ks = ks; na.h = h; na.rtol = rtol; na.atol = atol;
            for (int j = 0; j < 7; ++j) na.E[j] = RK_E[j];
            double err;
            if (int e = reduce_norm(c, na, &err)) return e;
            if (err < 1) {
                double factor = err == 0 ? MAX_FACTOR : std::min(MAX_FACTOR, SAFETY * std::pow(err, ERR_EXP));
                if (rejected) factor = std::min(1.0, factor);
                log_step(t, h, err, 1);
                h_abs *= factor;
                accepted = true;
                ++st->n_accepted;
            } else {
                log_step(t, h, err, 0);
                h_abs *= std::max(MIN_FACTOR, SAFETY * std::pow(err, ERR_EXP));
                rejected = true;
                ++st->n_rejected;
            }
        }
        // dense output on the stamps inside (t_new, t]  (decreasing time: stamps >= t_new)
        while (next_idx < num_steps && te[next_idx] >= t_new) {
            DenseArgs da;
            for (int j = 0; j < 7; ++j) for (int m = 0; m < 4; ++m) da.P[j][m] = RK_P[j][m];
            da.h = h;
            da.n = 0;
            while (da.n < DENSE_MAX && next_idx < num_steps && te[next_idx] >= t_new) {
                const double x = (te[next_idx] - t) / h;
                double* pp = da.p[da.n];
                pp[0] = x; pp[1] = x * x; pp[2] = pp[1] * x; pp[3] = pp[2] * x;
                da.idx[da.n++] = next_idx++;
            }
            hipLaunchKernelGGL(dense_kernel, dim3(nbE), dim3(256), 0, c.s, y, ks, n_el, D, da, xs_out, xs_is_f64, num_steps);
        }
        // accept: y <- y_new, f <- f_new (first-same-as-last)
        std::swap(y, ynew);
        std::swap(ks.p[0], ks.p[6]);
        t = t_new;
    }
    // reverse-diffusion predictor "denoise" step at t = eps
    {
        const float tfl = (float)eps;
        hipLaunchKernelGGL(stage_input_kernel, dim3(nbX), dim3(256), 0, c.s, y, ks, n_el, D, Dp, LinComb{{0}, 0, 0.0}, c.ws.X, (double*)nullptr,
                           (const RkCtl*)nullptr, 0, (const double*)nullptr, (const double*)nullptr);
        if (int e = eval_net(c, c.ws.X, tfl, 0, 0.f, c.ws.tmp)) return e;
        ++st->nfev;
        const float g = sigma_f32(tfl) * (float)std::sqrt(2.0 * (std::log(SIGMA_MAX) - std::log(SIGMA_MIN)));
        const float stepf = (float)((1.0 - eps) / num_steps);
        hipLaunchKernelGGL(denoise_kernel, dim3(nbE), dim3(256), 0, c.s, y, c.ws.tmp, n_el, g, stepf, x_out, x_is_f64,
                           (const RkCtl*)nullptr, (const double*)nullptr, (const double*)nullptr);
    }
    if (int e = vpho::check_launch("ode tail")) return e;
    int nan_host = 0;
    VPHO_HIP(hipMemcpyAsync(&nan_host, c.ws.nan_count, 4, hipMemcpyDeviceToHost, c.s));
    VPHO_HIP(hipStreamSynchronize(c.s));
    st->nan_count = nan_host;
    return 0;
}

// ---- controller-driven solve: everything is enqueued ahead, ONE sync per solve in the steady state ---------------------------
struct CtlHdr { RkCtl c; };
double* pinned_block(size_t bytes) {
    static thread_local char* p = nullptr;
    static thread_local size_t cap = 0;
    if (cap < bytes) {
        if (p) (void)hipHostFree(p);
        if (hipHostMalloc((void**)&p, bytes, hipHostMallocDefault) != hipSuccess) { p = nullptr; cap = 0; return nullptr; }
        cap = bytes;
    }
    return (double*)p;
}

// Wait for the stream without spinning: with one wait per solve the wake-up latency no longer matters, and a spinning
// hipStreamSynchronize per sampler thread eats the host's CPU quota (a throttled cgroup stalls every launch thread).
int blocking_wait(hipStream_t s) {
    static thread_local hipEvent_t ev = nullptr;
    if (!ev) VPHO_HIP(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
    VPHO_HIP(hipEventRecord(ev, s));
    VPHO_HIP(hipEventSynchronize(ev));
    return 0;
}

int ode_sample_device(Ctx& c, const float* feat_img, const float* init_x, double T0, double eps, int num_steps, double rtol, double atol,
                      void* xs_out, int xs_is_f64, void* x_out, int x_is_f64, vpho_ode_stats* st, double* step_log, int step_log_cap) {
    const vpho_score_weights* w = c.w;
    const long long n_el = c.n_el;
    const int D = w->D, Dp = w->Dp;
    const int nbX = (int)((c.R * Dp + 255) / 256), nbE = (int)((n_el + 255) / 256);
    RkCtl* ctl = c.ws.ctl;
    KSlots ks;
    for (int j = 0; j < 7; ++j) ks.p[j] = c.ws.K + (long long)j * n_el;
    const double g_scale = std::sqrt(2.0 * (std::log(SIGMA_MAX) - std::log(SIGMA_MIN)));

    VPHO_HIP(hipMemsetAsync(c.ws.nan_count, 0, 4, c.s));
    RkSetup su{T0, eps, rtol, atol, g_scale, num_steps, CTL_LOG_CAP, c.ws.te, c.ws.dense_p, c.ws.log};
    hipLaunchKernelGGL(rk_setup_kernel, dim3(1), dim3(1), 0, c.s, ctl, su);
    if (int e = prepare_cimg(c, feat_img)) return e;
    hipLaunchKernelGGL(f32_to_state_kernel, dim3(nbX), dim3(256), 0, c.s, init_x, n_el, D, Dp, c.ws.y, c.ws.X);

    // select_initial_step: f0 = fun(t0, y0) (t0 is known here), d0, d1 -> h0 on the device -> probe evaluation -> d2 -> h_abs
    if (int e = eval_rhs(c, c.ws.X, T0, c.ws.K, -1, nullptr, nullptr, nullptr, nullptr, true)) return e;
    {
        NormArgs na;
        memset(&na, 0, sizeof(na));
        na.y = c.ws.y; na.rtol = rtol; na.atol = atol;
        na.mode = 0;
        if (int e = enqueue_norm(c, na, &ctl->d[0])) return e;
        na.mode = 1; na.Ka = ks.p[0];
        if (int e = enqueue_norm(c, na, &ctl->d[1])) return e;
        hipLaunchKernelGGL(rk_init1_kernel, dim3(1), dim3(1), 0, c.s, ctl, n_el);
        LinComb lc;
        memset(&lc, 0, sizeof(lc));
        lc.n = 1; lc.c[0] = 1.0;
        hipLaunchKernelGGL(stage_input_kernel, dim3(nbX), dim3(256), 0, c.s, c.ws.y, ks, n_el, D, Dp, lc, c.ws.X, (double*)nullptr,
                           (const RkCtl*)ctl, 1, (const double*)c.ws.y, (const double*)c.ws.ynew);
        if (int e = embed_times(c, nullptr, 1, 1)) return e;
        CtlCall cc;
        cc.mode = 1; cc.stage = 0; cc.out_slot = 1;
        if (int e = eval_net(c, c.ws.X, 0.f, 1, 0.f, nullptr, 0, nullptr, nullptr, nullptr, nullptr, cc)) return e;
        na.mode = 2; na.Ka = ks.p[0]; na.Kb = ks.p[1];
        if (int e = enqueue_norm(c, na, &ctl->d[2])) return e;
        hipLaunchKernelGGL(rk_init2_kernel, dim3(1), dim3(1), 0, c.s, ctl, n_el);
    }

    RkC kc;
    for (int i = 0; i < 6; ++i) kc.C[i] = RK_C[i];
    DenseP dp;
    for (int j = 0; j < 7; ++j) for (int m = 0; m < 4; ++m) dp.P[j][m] = RK_P[j][m];
    auto enqueue_attempt = [&]() -> int {
        hipLaunchKernelGGL(rk_begin_kernel, dim3(1), dim3(1), 0, c.s, ctl, kc);
        if (int e = embed_times(c, nullptr, 6, 1)) return e;
        for (int s = 1; s <= 6; ++s) {
            LinComb lc;
            memset(&lc, 0, sizeof(lc));
            lc.n = s < 6 ? s : 6;
            for (int j = 0; j < lc.n; ++j) lc.c[j] = s < 6 ? RK_A[s][j] : RK_B[j];
            CtlCall cc;
            cc.mode = 1; cc.stage = s - 1; cc.out_slot = s; cc.write_ynew = s == 6;
            if (int e = eval_net(c, nullptr, 0.f, 1, 0.f, nullptr, s - 1, &lc, nullptr, nullptr, nullptr, cc)) return e;
        }
        NormArgs na;
        memset(&na, 0, sizeof(na));
        na.mode = 3; na.rtol = rtol; na.atol = atol;
        for (int j = 0; j < 7; ++j) na.E[j] = RK_E[j];
        if (int e = enqueue_norm(c, na, c.ws.result, true)) return e;
        hipLaunchKernelGGL(rk_end_kernel, dim3(1), dim3(1), 0, c.s, ctl, (const double*)c.ws.result, n_el);
        hipLaunchKernelGGL(dense_ctl_kernel, dim3(nbE), dim3(256), 0, c.s, (const RkCtl*)ctl, (const double*)c.ws.y, (const double*)c.ws.ynew,
                           (const float*)c.ws.K, n_el, D, dp, xs_out, xs_is_f64, num_steps);
        return vpho::check_launch("rk attempt");
    };
    // reverse-diffusion predictor "denoise" step at t = eps; its kernels are no-ops until the controller says done
    auto enqueue_final = [&]() -> int {
        const float tfl = (float)eps;
        hipLaunchKernelGGL(stage_input_kernel, dim3(nbX), dim3(256), 0, c.s, (const double*)c.ws.y, ks, n_el, D, Dp, LinComb{{0}, 0, 0.0}, c.ws.X,
                           (double*)nullptr, (const RkCtl*)ctl, 2, (const double*)c.ws.y, (const double*)c.ws.ynew);
        CtlCall cc;
        cc.mode = 2;
        if (int e = eval_net(c, c.ws.X, tfl, 0, 0.f, c.ws.tmp, -1, nullptr, nullptr, nullptr, nullptr, cc)) return e;
        const float g = sigma_f32(tfl) * (float)g_scale;
        const float stepf = (float)((1.0 - eps) / num_steps);
        hipLaunchKernelGGL(denoise_kernel, dim3(nbE), dim3(256), 0, c.s, (const double*)c.ws.y, c.ws.tmp, n_el, g, stepf, x_out, x_is_f64,
                           (const RkCtl*)ctl, (const double*)c.ws.y, (const double*)c.ws.ynew);
        return vpho::check_launch("ode tail");
    };

    // Enqueue as many attempts as the previous solve on this workspace needed, then the final step, then look ONCE.
    static std::mutex hint_mu;
    static std::unordered_map<const void*, int> hint;
    int n_first = 10;
    {
        std::lock_guard<std::mutex> lk(hint_mu);
        auto it = hint.find((const void*)ctl);
        if (it != hint.end()) n_first = it->second;
    }
    const size_t hdr_bytes = sizeof(RkCtl) + 8;
    char* host = (char*)pinned_block(hdr_bytes + (size_t)CTL_LOG_CAP * 32);
    VPHO_REQUIRE(host != nullptr, "hipHostMalloc failed");
    RkCtl* hc = (RkCtl*)host;
    int* nan_host = (int*)(host + sizeof(RkCtl));
    int enq = 0;
    for (int round = 0;; ++round) {
        const int n = round == 0 ? std::max(1, n_first) : 2;
        for (int i = 0; i < n; ++i) if (int e = enqueue_attempt()) return e;
        enq += n;
        if (int e = enqueue_final()) return e;
        VPHO_HIP(hipMemcpyAsync(hc, ctl, sizeof(RkCtl), hipMemcpyDeviceToHost, c.s));
        VPHO_HIP(hipMemcpyAsync(nan_host, c.ws.nan_count, 4, hipMemcpyDeviceToHost, c.s));
        if (int e = blocking_wait(c.s)) return e;
        if (hc->done) break;
        VPHO_REQUIRE(enq < CTL_LOG_CAP, "vpho_ode_sample: no convergence after %d attempted steps", enq);
    }
    {
        std::lock_guard<std::mutex> lk(hint_mu);
        hint[(const void*)ctl] = hc->n_attempts;
    }
    st->nfev = hc->nfev + 1;
    st->n_accepted = hc->n_accepted; st->n_rejected = hc->n_rejected; st->status = hc->status; st->n_log = hc->n_log;
    st->nan_count = *nan_host;
    if (step_log && hc->n_log > 0) {
        const int n = std::min(std::min(hc->n_log, step_log_cap), CTL_LOG_CAP);
        if (n > 0) {
            double* lg = (double*)(host + hdr_bytes);
            VPHO_HIP(hipMemcpyAsync(lg, c.ws.log, (size_t)n * 32, hipMemcpyDeviceToHost, c.s));
            if (int e = blocking_wait(c.s)) return e;
            memcpy(step_log, lg, (size_t)n * 32);
        }
    }
    if (hc->status == 1) return vpho::fail("vpho_ode_sample: step size underflow at t=%g", hc->t);
    return 0;
}

}  // namespace

extern "C" int vpho_ode_sample(const vpho_score_weights* w, const float* feat_img, int bs, int S, const float* init_x,
                               double T0, double eps, int num_steps, double rtol, double atol,
                               void* xs_out, int xs_is_f64, void* x_out, int x_is_f64, void* workspace, long long workspace_bytes,
                               vpho_ode_stats* st, double* step_log, int step_log_cap, void* stream) {
    if (int e = check_weights(w)) return e;
    VPHO_REQUIRE(bs > 0 && S > 0 && num_steps >= 1 && feat_img && init_x && xs_out && x_out && workspace && st, "vpho_ode_sample: bad argument");
    VPHO_REQUIRE(T0 > eps, "vpho_ode_sample: T0 must exceed eps (backward integration)");
    Ctx c;
    c.w = w; c.bs = bs; c.S = S; c.R = (long long)bs * S; c.n_el = c.R * w->D; c.NH = w->nheads * 256; c.s = (hipStream_t)stream;
    c.ws = carve(*w, bs, S, (char*)workspace);
    VPHO_REQUIRE(workspace_bytes >= c.ws.bytes, "vpho_ode_sample: workspace %lld < %lld bytes", workspace_bytes, c.ws.bytes);
    memset(st, 0, sizeof(*st));
    // VPHO_RK_HOST=1: scipy's controller on the host (one sync per attempted step); default: controller on the device
    static const bool host_ctl = getenv("VPHO_RK_HOST") && atoi(getenv("VPHO_RK_HOST")) != 0;
    if (host_ctl || num_steps > CTL_MAX_STEPS)
        return ode_sample_host(c, feat_img, init_x, T0, eps, num_steps, rtol, atol, xs_out, xs_is_f64, x_out, x_is_f64, st, step_log, step_log_cap);
    return ode_sample_device(c, feat_img, init_x, T0, eps, num_steps, rtol, atol, xs_out, xs_is_f64, x_out, x_is_f64, st, step_log, step_log_cap);
}
