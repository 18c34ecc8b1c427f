// Denoising-score-matching training step of a score network (first slice): the pieces around the GEMMs.
// Reference: lib/model/score_based_model.py:11-42,117-128 (loss_fn / get_score_loss), lib/model/sde.py:15-18 (ve_marginal_prob),
// lib/model/denoiser.py:19-31,68-82,176-189,244-257 (Fourier time features, BaseDenoiser.forward, ParallelLinear heads),
// lib/engine/train_diff_hand_obj.py:49-52,169-199 (AdamW, one step per batch).
// The matrix products (rows = repeat_num * batch, 1408 x nheads*256 first head layer and its two backward products, the
// encoders) run on conv_igemm's fp32-MFMA GEMM; here are the per-row preparation, the 256 -> 3 second head layer and its
// backward, the loss with its seed gradient, ReLU masks, bias gradients (deterministic column sums) and the AdamW update.
// All HBM-bound; row-major fp32.
#include "common.h"
#include "../../include/vpho_hip.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

namespace {

constexpr float SIG_MIN = 0.01f, SIG_MAX = 50.0f;
inline int nblk(long long n, int bs = 256) { return (int)((n + bs - 1) / bs); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// V consecutive floats as one access (V = 4: 16 B) -- the HBM-bound kernels below move whole channel quads per thread
template <int V>
__device__ inline void ldv(const float* __restrict__ p, float (&o)[V]) {
    if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
    else o[0] = p[0];
}
template <int V>
__device__ inline void stv(float* __restrict__ p, const float (&v)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else p[0] = v[0];
}

// rows r = rep * bs + b:  std = sigma_min (sigma_max / sigma_min)^t,  x_t = x0[b] + z * std,  emb = [sin, cos](t W 2 pi)
__global__ void dsm_prepare_kernel(const float* __restrict__ gt, const float* __restrict__ t, const float* __restrict__ z,
                                   const float* __restrict__ Wf, int bs, int rows, int D, int Dp,
                                   float* __restrict__ xt, float* __restrict__ emb, float* __restrict__ stdv) {
    const int r = blockIdx.x;
    const int b = r % bs;
    const float tt = t[r];
    const float sd = SIG_MIN * powf(SIG_MAX / SIG_MIN, tt);
    if (threadIdx.x == 0) stdv[r] = sd;
    for (int c = threadIdx.x; c < Dp; c += blockDim.x) xt[(long long)r * Dp + c] = c < D ? gt[b * D + c] + z[(long long)r * D + c] * sd : 0.f;
    for (int k = threadIdx.x; k < 64; k += blockDim.x) {
        float a = tt * Wf[k];
        a = a * 2.0f;
        a = a * 3.14159265358979323846f;
        emb[r * 128 + k] = sinf(a);
        emb[r * 128 + 64 + k] = cosf(a);
    }
}

// score[r][3n+d] = (b2[n][d] + sum_c h[r][n*256+c] * w2[n][c][d]) / (std[r] + 1e-7): one wave per (row, head)
__global__ __launch_bounds__(256) void plinear2_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const float* __restrict__ stdv, long long rows, int n, float* __restrict__ score) {
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= rows * n) return;
    const long long r = item / n;
    const int hd = (int)(item - r * n);
    const float* hp = h + (r * n + hd) * 256;
    const float* wp = w2 + (long long)hd * 256 * 3;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < 256; c += 64) {
        const float hv = hp[c];
        s0 += hv * wp[c * 3]; s1 += hv * wp[c * 3 + 1]; s2 += hv * wp[c * 3 + 2];
    }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        const float den = stdv[r] + 1e-7f;
        float* o = score + r * (3 * n) + hd * 3;
        o[0] = (s0 + b2[hd * 3]) / den; o[1] = (s1 + b2[hd * 3 + 1]) / den; o[2] = (s2 + b2[hd * 3 + 2]) / den;
    }
}

// per element: target = -z std / std^2, weight std^2, term = w (s - target)^2;  dscore = 2 w (s - target) * inv_count;
// dout = dscore / (std + 1e-7) (gradient w.r.t. the un-normalised head output); per-block partial sums of the loss
__global__ __launch_bounds__(256) void dsm_loss_kernel(const float* __restrict__ score, const float* __restrict__ z, const float* __restrict__ stdv,
                                                       long long n_el, int D, float inv_count, float* __restrict__ dout, double* __restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_el; i += (long long)gridDim.x * 256) {
        const float sd = stdv[i / D];
        const float w = sd * sd;
        const float target = (0.f - z[i]) * sd / w;
        const float diff = score[i] - target;
        acc += (double)(w * (diff * diff));
        dout[i] = (2.f * w * diff * inv_count) / (sd + 1e-7f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n, double scale, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *out = red[0] * scale;
}

// nn.MSELoss (mean) times a loss weight: per-block partial sums of (pd - gt)^2, grad = weight * 2 (pd - gt) / n
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ pd, const float* __restrict__ gt, long long n_el, float grad_scale,
                                                       float* __restrict__ grad, double* __restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_el; i += (long long)gridDim.x * 256) {
        const float diff = pd[i] - gt[i];
        acc += (double)(diff * diff);
        grad[i] = diff * grad_scale;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// dpre[r][n*256+c] = h > 0 ? sum_d dout[r][3n+d] * w2[n][c][d] : 0   (backward of the second head layer and of the ReLU before it)
__global__ void plinear2_bwd_input_kernel(const float* __restrict__ h, const float* __restrict__ dout, const float* __restrict__ w2,
                                          long long rows, int n, float* __restrict__ dpre) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n * 256) return;
    const int c = (int)(i & 255);
    const long long rn = i >> 8;
    const int hd = (int)(rn % n);
    const long long r = rn / n;
    const float* d = dout + r * (3 * n) + hd * 3;
    const float* w = w2 + ((long long)hd * 256 + c) * 3;
    const float g = d[0] * w[0] + d[1] * w[1] + d[2] * w[2];
    dpre[i] = h[i] > 0.f ? g : 0.f;
}
// dw2[n][c][d] = sum_r h[r][n*256+c] * dout[r][3n+d];  db2[n][d] = sum_r dout[r][3n+d]: one block per head, rows strided
// over 4 waves then combined in a fixed order
__global__ __launch_bounds__(256) void plinear2_bwd_weight_kernel(const float* __restrict__ h, const float* __restrict__ dout, long long rows, int n,
                                                                  float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ float part[4][256 * 3 + 3];
    const int hd = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a[4][3] = {{0}};
    float bsum[3] = {0.f, 0.f, 0.f};
    for (long long r = wave; r < rows; r += 4) {
        const float* d = dout + r * (3 * n) + hd * 3;
        const float d0 = d[0], d1 = d[1], d2 = d[2];
        const float* hp = h + (r * n + hd) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float hv = hp[q * 64 + lane];
            a[q][0] += hv * d0; a[q][1] += hv * d1; a[q][2] += hv * d2;
        }
        bsum[0] += d0; bsum[1] += d1; bsum[2] += d2;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int d = 0; d < 3; ++d) part[wave][(q * 64 + lane) * 3 + d] = a[q][d];
    if (lane == 0) for (int d = 0; d < 3; ++d) part[wave][768 + d] = bsum[d];
    __syncthreads();
    for (int i = threadIdx.x; i < 771; i += 256) {
        const float s = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
        if (i < 768) dw2[(long long)hd * 768 + i] = s; else db2[hd * 3 + (i - 768)] = s;
    }
}

// dx = y > 0 ? dy : 0 on a [rows][cols] slice with leading dimensions (masks the gradient of a ReLU output y)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y, long long rows, int cols,
                                float* __restrict__ dx, int ld_dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long long r = i / cols;
    const int c = (int)(i - r * cols);
    dx[r * ld_dx + c] = y[r * ld_y + c] > 0.f ? dy[r * ld_dy + c] : 0.f;
}

// out[b][c] = sum_rep x[rep*bs + b][c_off + c]  (gradient w.r.t. the image encoding, shared by the repeat_num draws)
__global__ void sum_repeats_kernel(const float* __restrict__ x, int ld, int c_off, int bs, int reps, int cols, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bs * cols) return;
    const int b = (int)(i / cols), c = (int)(i - (long long)b * cols);
    float s = 0.f;
    for (int r = 0; r < reps; ++r) s += x[((long long)r * bs + b) * ld + c_off + c];
    out[i] = s;
}

// y[c][r] = x[r][c] through a padded 32x32 LDS tile
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int rows, int cols, int ldx, float* __restrict__ y, int ldy) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? x[(long long)r * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) y[(long long)c * ldy + r] = tile[tx][j];
    }
}

// Transposed im2col for the weight gradient of a convolution: out[(tap*Cin + ci)][p] = x[n, oy*stride + r - pad_y, ox*stride + s - pad_x, ci]
// (0 outside the image), p = (n*OH + oy)*OW + ox, tap = r*KW + s.  32x32 tiles through LDS: reads are 128-B channel rows of
// one pixel, writes 128-B pixel runs of one (tap, channel) row; grid (Cin/32, P/32, KH*KW).
struct Im2colArgs { const float* x; int N, H, W, Cin, x_ld, KH, KW, stride, pad_y, pad_x, OH, OW; long long P, ldo; float* out; };
__global__ __launch_bounds__(256) void im2col_t_kernel(const Im2colArgs a) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32;
    const long long p0 = (long long)blockIdx.y * 32;
    const int tap = blockIdx.z, r = tap / a.KW, s = tap - r * a.KW;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const long long p = p0 + j;
        float v = 0.f;
        if (p < a.P && c0 + tx < a.Cin) {
            const int ox = (int)(p % a.OW);
            const long long q = p / a.OW;
            const int oy = (int)(q % a.OH), n = (int)(q / a.OH);
            const int iy = oy * a.stride + r - a.pad_y, ix = ox * a.stride + s - a.pad_x;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = a.x[(((long long)n * a.H + iy) * a.W + ix) * a.x_ld + c0 + tx];
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long long p = p0 + tx;
        if (c < a.Cin && p < a.ldo) a.out[((long long)tap * a.Cin + c) * a.ldo + p] = p < a.P ? tile[tx][j] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm, training mode
// nn.BatchNorm2d.train() on NHWC [rows][C] (rows = N*H*W; every BatchNorm of backbone_FPN_HFL.py / encoding.py /
// head_inplane.py under model.train(), train_diff_hand_obj.py:171): per-channel reductions are column sums -- block = 32
// channels x 8 row groups over a chunk of rows, fp64 partials [chunk][C] combined in a fixed order by the finishing kernel.
// mode 0: (sum x, sum x^2);  mode 1: (sum dy, sum dy * xhat) with xhat = (x - mean) * invstd
struct BnRedArgs {
    const float* x; const float* dy; const float* mean; const float* invstd; long long rows; int C, ld, mode, rows_per_chunk; double* part;
};
// block = 8 column groups of V channels x 32 row groups over one chunk of rows; a thread walks rows g, g+32, ... of its chunk with
// four loads in flight, fp64 partial sums; the 32 row groups are combined in a fixed order.  mode 2: plain column sum (s0 only).
template <int V>
__global__ __launch_bounds__(256) void col_reduce_kernel(const BnRedArgs a) {
    __shared__ double p0[32][8 * V], p1[32][8 * V];
    const int cq = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int c = (blockIdx.x * 8 + cq) * V;
    const long long r0 = (long long)blockIdx.y * a.rows_per_chunk, r1 = r0 + a.rows_per_chunk < a.rows ? r0 + a.rows_per_chunk : a.rows;
    double s0[V], s1[V];
    float mu[V], is[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { s0[v] = 0.0; s1[v] = 0.0; mu[v] = 0.f; is[v] = 0.f; }
    const bool live = c < a.C;                           // C % V == 0 on the vector path, so a live thread owns V valid channels
    if (live && a.mode == 1) {
#pragma unroll
        for (int v = 0; v < V; ++v) { mu[v] = a.mean[c + v]; is[v] = a.invstd[c + v]; }
    }
    if (live) {
        for (long long r = r0 + g; r < r1; r += 128) {
            float xv[4][V], dv[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long rr = r + 32 * u;
                const bool ok = rr < r1;
#pragma unroll
                for (int v = 0; v < V; ++v) { xv[u][v] = 0.f; dv[u][v] = 0.f; }
                if (ok) {
                    ldv<V>(a.x + rr * a.ld + c, xv[u]);
                    if (a.mode == 1) ldv<V>(a.dy + rr * a.ld + c, dv[u]);
                }
                if (!ok && a.mode == 1) {                 // padding rows must not contribute (x - mean) terms
#pragma unroll
                    for (int v = 0; v < V; ++v) xv[u][v] = mu[v];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (a.mode == 0) { s0[v] += (double)xv[u][v]; s1[v] += (double)xv[u][v] * (double)xv[u][v]; }
                    else if (a.mode == 1) { s0[v] += (double)dv[u][v]; s1[v] += (double)dv[u][v] * (double)((xv[u][v] - mu[v]) * is[v]); }
                    else s0[v] += (double)xv[u][v];
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) { p0[g][cq * V + v] = s0[v]; p1[g][cq * V + v] = s1[v]; }
    __syncthreads();
    if (threadIdx.x < 8 * V) {
        const int cc = blockIdx.x * 8 * V + threadIdx.x;
        if (cc < a.C) {
            double t0 = p0[0][threadIdx.x], t1 = p1[0][threadIdx.x];
            for (int k = 1; k < 32; ++k) { t0 += p0[k][threadIdx.x]; t1 += p1[k][threadIdx.x]; }
            if (a.mode == 2) a.part[(long long)blockIdx.y * a.C + cc] = t0;
            else {
                a.part[((long long)blockIdx.y * 2) * a.C + cc] = t0;
                a.part[((long long)blockIdx.y * 2 + 1) * a.C + cc] = t1;
            }
        }
    }
}
// sum over the row chunks of one or two partial-sum planes (double: the chunks of col_reduce_kernel; float: the per-tile partial sums a
// convolution's epilogue wrote, vpho_conv_desc.stats): block = 32 channels x 32 chunk groups (chunks g, g+32, ... per thread, the 32
// group sums combined in a fixed order); the result is valid in the threads with g == 0.  (Round 6: 8 groups -> 32: a finishing kernel is
// one short chain of dependent loads per thread, and a training step runs ~370 of them back to back with their consumers.)
constexpr int FIN_G = 32;
template <int PLANES, typename T>
__device__ inline bool finish_sums(const T* __restrict__ part, int chunks, int C, int& c, double& s0, double& s1) {
    __shared__ double q0[FIN_G][32], q1[FIN_G][32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    c = blockIdx.x * 32 + cl;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        for (int k = g; k < chunks; k += FIN_G) {
            a0 += (double)part[((long long)k * PLANES) * C + c];
            if (PLANES == 2) a1 += (double)part[((long long)k * PLANES + 1) * C + c];
        }
    }
    q0[g][cl] = a0; q1[g][cl] = a1;
    __syncthreads();
    if (g != 0 || c >= C) return false;
    s0 = q0[0][cl]; s1 = q1[0][cl];
    for (int k = 1; k < FIN_G; ++k) { s0 += q0[k][cl]; s1 += q1[k][cl]; }
    return true;
}
__global__ __launch_bounds__(32 * FIN_G) void colsum_finish_kernel(const double* __restrict__ part, int chunks, int C, float* __restrict__ out) {
    int c; double s, unused;
    if (finish_sums<1>(part, chunks, C, c, s, unused)) out[c] = (float)s;
}
// chunking of a [rows][C] column reduction: enough (column block, row chunk) workgroups to fill 256 CUs several times over,
// at most 256 chunks (the workspace holds 256 x 2 x C doubles), at least 32 rows per chunk (one per row group)
int col_chunks(long long rows, int C, int V, int* rows_per_chunk) {
    const int colblocks = (C + 8 * V - 1) / (8 * V);
    long long chunks = std::max(1, 2048 / colblocks);
    chunks = std::min<long long>(std::min<long long>(chunks, 256), (rows + 31) / 32);
    if (chunks < 1) chunks = 1;
    *rows_per_chunk = (int)((rows + chunks - 1) / chunks);
    return (int)((rows + *rows_per_chunk - 1) / *rows_per_chunk);
}
// launches the column reduction; returns the number of chunks written to the workspace (the caller launches its finishing kernel: the
// in-launch finish by arrival tickets of round 5 made the training step 0.5-2 ms slower -- the one-workgroup finishing kernels overlap
// the weight-gradient stream for free -- and was removed in round 6 together with its per-stream ticket allocation)
int launch_col_reduce(BnRedArgs a, hipStream_t s) {
    const bool vec = a.C % 4 == 0 && a.ld % 4 == 0 && aligned16(a.x) && (a.mode != 1 || aligned16(a.dy));
    const int V = vec ? 4 : 1;
    const int chunks = col_chunks(a.rows, a.C, V, &a.rows_per_chunk);
    const dim3 grid((a.C + 8 * V - 1) / (8 * V), chunks);
    if (vec) hipLaunchKernelGGL(col_reduce_kernel<4>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(col_reduce_kernel<1>, grid, dim3(256), 0, s, a);
    return chunks;
}
// statistics: mean, biased variance -> invstd = 1/sqrt(var + eps); running stats with the unbiased variance (torch semantics)
template <typename T>
__global__ __launch_bounds__(32 * FIN_G) void bn_finish_stats_kernel(const T* __restrict__ part, int chunks, int C, long long rows, float eps, float momentum,
                                                              float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var) {
    int c; double s, ss;
    if (!finish_sums<2>(part, chunks, C, c, s, ss)) return;
    const double m = s / (double)rows;
    double var = ss / (double)rows - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}
template <typename T>
__global__ __launch_bounds__(32 * FIN_G) void bn_finish_grads_kernel(const T* __restrict__ part, int chunks, int C, float* __restrict__ dbeta, float* __restrict__ dgamma) {
    int c; double s, ss;
    if (finish_sums<2>(part, chunks, C, c, s, ss)) { dbeta[c] = (float)s; dgamma[c] = (float)ss; }
}
// y = lrelu((x - mean) * invstd * gamma + beta, slope); a thread handles V consecutive channels of a row
template <int V>
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ res, long long rows, int C, int ld, float slope,
                                float* __restrict__ y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = C / V;
    if (i >= rows * cq) return;
    const long long r = i / cq;
    const int c = (int)(i - r * cq) * V;
    float xv[V], m[V], is[V], ga[V], be[V], o[V];
    ldv<V>(x + r * ld + c, xv); ldv<V>(mean + c, m); ldv<V>(invstd + c, is); ldv<V>(gamma + c, ga); ldv<V>(beta + c, be);
    float rv[V];
#pragma unroll
    for (int v = 0; v < V; ++v) rv[v] = 0.f;
    if (res) ldv<V>(res + r * ld + c, rv);                       // residual branch of a bottleneck, added before the activation
#pragma unroll
    for (int v = 0; v < V; ++v) { const float t = ((xv[v] - m[v]) * is[v] * ga[v] + be[v]) + rv[v]; o[v] = t > 0.f ? t : t * slope; }
    stv<V>(y + r * ld + c, o);
}
// dx = gamma * invstd / rows * (rows * dy - dbeta - xhat * dgamma)
template <int V>
__global__ void bn_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean, const float* __restrict__ invstd,
                                   const float* __restrict__ gamma, const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                   long long rows, int C, int ld, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = C / V;
    if (i >= rows * cq) return;
    const long long r = i / cq;
    const int c = (int)(i - r * cq) * V;
    float xv[V], dv[V], mu[V], is[V], ga[V], db[V], dg[V], o[V];
    ldv<V>(x + r * ld + c, xv); ldv<V>(dy + r * ld + c, dv); ldv<V>(mean + c, mu); ldv<V>(invstd + c, is); ldv<V>(gamma + c, ga);
    ldv<V>(dbeta + c, db); ldv<V>(dgamma + c, dg);
    const float m = (float)rows;
#pragma unroll
    for (int v = 0; v < V; ++v) { const float xh = (xv[v] - mu[v]) * is[v]; o[v] = ga[v] * is[v] / m * (m * dv[v] - db[v] - xh * dg[v]); }
    stv<V>(dx + r * ld + c, o);
}
// The same element-wise backward with (a) the other branch of a residual sum added to the result (encoding.Residual's identity shortcut:
// dx = BatchNorm backward + dres) and (b) the column sums of what it stores -- the bias gradient of the convolution that produced x
// (d bias = sum over pixels of the gradient at its output) -- taken on the way out instead of by a second pass over dx: the tiling of
// col_reduce_kernel (block = 8 channel groups of V x 32 row groups over a chunk of rows, fp64 partials [chunk][C], fixed order), every
// element computed by the expression of bn_backward_kernel.
struct BnBackSumArgs {
    const float* x; const float* dy; const float* mean; const float* invstd; const float* gamma; const float* dbeta; const float* dgamma; const float* res;
    long long rows; int C, ld, rows_per_chunk; float* dx; double* part;
};
template <int V>
__global__ __launch_bounds__(256) void bn_backward_sum_kernel(const BnBackSumArgs a) {
    __shared__ double p0[32][8 * V];
    const int cq = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int c = (blockIdx.x * 8 + cq) * V;
    const long long r0 = (long long)blockIdx.y * a.rows_per_chunk, r1 = r0 + a.rows_per_chunk < a.rows ? r0 + a.rows_per_chunk : a.rows;
    double s0[V];
    float mu[V], is[V], ga[V], db[V], dg[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { s0[v] = 0.0; mu[v] = is[v] = ga[v] = db[v] = dg[v] = 0.f; }
    const bool live = c < a.C;
    if (live) {
        ldv<V>(a.mean + c, mu); ldv<V>(a.invstd + c, is); ldv<V>(a.gamma + c, ga); ldv<V>(a.dbeta + c, db); ldv<V>(a.dgamma + c, dg);
        const float m = (float)a.rows;
        for (long long r = r0 + g; r < r1; r += 128) {
            float xv[4][V], dv[4][V], rv[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long rr = r + 32 * u;
#pragma unroll
                for (int v = 0; v < V; ++v) { xv[u][v] = 0.f; dv[u][v] = 0.f; rv[u][v] = 0.f; }
                if (rr < r1) {
                    ldv<V>(a.x + rr * a.ld + c, xv[u]); ldv<V>(a.dy + rr * a.ld + c, dv[u]);
                    if (a.res) ldv<V>(a.res + rr * a.ld + c, rv[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long rr = r + 32 * u;
                if (rr >= r1) continue;
                float o[V];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float xh = (xv[u][v] - mu[v]) * is[v];
                    o[v] = ga[v] * is[v] / m * (m * dv[u][v] - db[v] - xh * dg[v]);
                    if (a.res) o[v] = o[v] + rv[u][v];
                    s0[v] += (double)o[v];
                }
                stv<V>(a.dx + rr * a.ld + c, o);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) p0[g][cq * V + v] = s0[v];
    __syncthreads();
    if (threadIdx.x < 8 * V) {
        const int cc = blockIdx.x * 8 * V + threadIdx.x;
        if (cc < a.C) {
            double t0 = p0[0][threadIdx.x];
            for (int k = 1; k < 32; ++k) t0 += p0[k][threadIdx.x];
            a.part[(long long)blockIdx.y * a.C + cc] = t0;
        }
    }
}
// dx = y > 0 ? dy : dy * slope  (LeakyReLU backward given its output; slope > 0 keeps the sign of the input)
template <int V>
__global__ void lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n, float slope, float* __restrict__ dx) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= n) return;
    float d[V], yv[V], o[V];
    ldv<V>(dy + i, d); ldv<V>(y + i, yv);
#pragma unroll
    for (int v = 0; v < V; ++v) o[v] = yv[v] > 0.f ? d[v] : d[v] * slope;
    stv<V>(dx + i, o);
}

// y = lrelu(a + b, slope)  (residual add of a bottleneck; slope 1 = plain sum, e.g. of two input gradients)
template <int V>
__global__ void add_lrelu_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float slope, float* __restrict__ y) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= n) return;
    float av[V], bv[V], o[V];
    ldv<V>(a + i, av); ldv<V>(b + i, bv);
#pragma unroll
    for (int v = 0; v < V; ++v) { const float t = av[v] + bv[v]; o[v] = t > 0.f ? t : t * slope; }
    stv<V>(y + i, o);
}

// ------------------------------------------------------------------------------------------------ pooling / resize backward
// nn.MaxPool2d backward as a gather (deterministic): input pixel (iy, ix) receives dy of every window whose arg-max it is --
// the first maximum in row-major window order, as torch's CPU/GPU kernels pick it.
template <int V>
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int H, int W, int C, int k, int stride, int pad,
                                   int OH, int OW, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // V consecutive channels of one input pixel
    const int CV = C / V;
    if (i >= (long long)N * H * W * CV) return;
    const int c = (int)(i % CV) * V;
    long long p = i / CV;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const long long n = p / H;
    const float* xb = x + n * H * W * (long long)C + c;
    float g[V];
#pragma unroll
    for (int v = 0; v < V; ++v) g[v] = 0.f;
    // windows covering iy: oy*stride - pad <= iy <= oy*stride - pad + k - 1
    const int oy_lo = max(0, (iy + pad - k + stride) / stride), oy_hi = min(OH - 1, (iy + pad) / stride);
    const int ox_lo = max(0, (ix + pad - k + stride) / stride), ox_hi = min(OW - 1, (ix + pad) / stride);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            float m[V];
            bool mine[V];                                  // is (iy, ix) the FIRST maximum of this window (row-major order)?
#pragma unroll
            for (int v = 0; v < V; ++v) { m[v] = -INFINITY; mine[v] = false; }
            bool first = true;
            for (int r = 0; r < k; ++r) {
                const int yy = oy * stride - pad + r;
                if (yy < 0 || yy >= H) continue;
                for (int q = 0; q < k; ++q) {
                    const int xx = ox * stride - pad + q;
                    if (xx < 0 || xx >= W) continue;
                    float xv[V];
                    ldv<V>(xb + ((long long)yy * W + xx) * C, xv);
                    const bool here = yy == iy && xx == ix;
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (xv[v] > m[v] || first) { m[v] = xv[v]; mine[v] = here; }
                    first = false;
                }
            }
            float dv[V];
            ldv<V>(dy + ((n * OH + oy) * OW + ox) * (long long)C + c, dv);
#pragma unroll
            for (int v = 0; v < V; ++v) g[v] += mine[v] ? dv[v] : 0.f;
        }
    stv<V>(dx + ((n * H + iy) * W + ix) * (long long)C + c, g);
}

__device__ inline void lin_src_t(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}
// F.interpolate(mode='bilinear', align_corners=False) backward as a gather: dx[iy, ix] = sum over the outputs whose two source
// rows / columns include (iy, ix) of their weights * dy, outputs visited in ascending order (deterministic)
// a thread owns V consecutive channels of one input pixel: the candidate range and the weights are per pixel, the loads 4 V bytes
template <int V>
__global__ void resize_bilinear_bwd_kernel(const float* __restrict__ dy, int N, int OH, int OW, int C, int H, int W, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = C / V;
    if (i >= (long long)N * H * W * cq) return;
    const int c = (int)(i % cq) * V;
    long long p = i / cq;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const long long n = p / H;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    // outputs whose source position lies in (iy - 1, iy + 1): a conservative index range, exact weights recomputed inside
    const int oy_lo = max(0, (int)floorf(((float)iy - 1.f + 0.5f) / sy - 0.5f) - 1), oy_hi = min(OH - 1, (int)ceilf(((float)iy + 1.f + 0.5f) / sy - 0.5f) + 1);
    const int ox_lo = max(0, (int)floorf(((float)ix - 1.f + 0.5f) / sx - 0.5f) - 1), ox_hi = min(OW - 1, (int)ceilf(((float)ix + 1.f + 0.5f) / sx - 0.5f) + 1);
    float g[V];
#pragma unroll
    for (int v = 0; v < V; ++v) g[v] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1; float ly;
        lin_src_t(oy, sy, H, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            int x0, x1; float lx;
            lin_src_t(ox, sx, W, x0, x1, lx);
            const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            if (wx == 0.f) continue;
            float d[V];
            ldv<V>(dy + ((n * OH + oy) * OW + ox) * (long long)C + c, d);
            const float wgt = wy * wx;
#pragma unroll
            for (int v = 0; v < V; ++v) g[v] += wgt * d[v];
        }
    }
    stv<V>(dx + ((n * H + iy) * W + ix) * (long long)C + c, g);
}

// torchvision roi_align backward (aligned=False, adaptive sampling; one RoI per image): every sample of every bin adds its four
// bilinear weights * dy / count to the feature gradient.  Bins of one RoI overlap in the taps they touch -> atomicAdd on fp32
// (the summation order, hence the last bit, is not fixed -- as in torchvision's own GPU kernel).  dfeat must be zeroed.
__global__ void roi_align_bwd_kernel(const float* __restrict__ dy, int ldo, int c_off, int N, int H, int W, int C, const float* __restrict__ boxes,
                                     float scale, int P, const unsigned char* __restrict__ flip_w, float* __restrict__ dfeat) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * P * P * C) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int pw = (int)(p % P); p /= P;
    const int ph = (int)(p % P);
    const int n = (int)(p / P);
    const float x1 = boxes[n * 4 + 0] * scale, y1 = boxes[n * 4 + 1] * scale;
    const float x2 = boxes[n * 4 + 2] * scale, y2 = boxes[n * 4 + 3] * scale;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float cnt = fmaxf((float)(gh * gw), 1.f);
    const int ow = (flip_w && flip_w[n]) ? P - 1 - pw : pw;
    const float g = dy[(((long long)n * P + ph) * P + ow) * ldo + c_off + c] / cnt;
    float* f = dfeat + (long long)n * H * W * C + c;
    for (int iy = 0; iy < gh; ++iy) {
        float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
            float yy = y;
            if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
            if (yy <= 0.f) yy = 0.f;
            if (x <= 0.f) x = 0.f;
            int yl = (int)yy, xl = (int)x, yh, xh;
            if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
            if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
            const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
            atomicAdd(f + ((long long)yl * W + xl) * C, hy * hx * g);
            atomicAdd(f + ((long long)yl * W + xh) * C, hy * lx * g);
            atomicAdd(f + ((long long)yh * W + xl) * C, ly * hx * g);
            atomicAdd(f + ((long long)yh * W + xh) * C, ly * lx * g);
        }
    }
}

// The same gradient as a GATHER (one RoI per image, so every feature pixel has one owner): bilinear sampling is separable,
//   dfeat[n,y,x,c] += 1/count * sum_ph sum_pw WY[y][ph] * WX[x][pw] * dy[n,ph,pw,c],
// WY[y][ph] = total weight the samples of bin row ph put on feature row y (same border rules as the forward), WX likewise.
// A workgroup owns one feature row of one image: it builds WY (one row) and WX (W x P) in LDS, then every thread accumulates
// channel quads of its pixels over the few bins that reach them.  No atomics: fixed summation order, each output written once.
__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(const float* __restrict__ dy, int ldo, int c_off, int H, int W, int C,
                                                                   const float* __restrict__ boxes, float scale, int P,
                                                                   const unsigned char* __restrict__ flip_w, float* __restrict__ dfeat) {
    extern __shared__ float sm[];
    float* wy = sm;                       // [P]
    float* wx = sm + P;                   // [W][P]
    int* rng = reinterpret_cast<int*>(wx + W * P);      // [2] ph range, then [W][2] pw ranges
    const int y = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const float x1 = boxes[n * 4 + 0] * scale, y1 = boxes[n * 4 + 1] * scale;
    const float x2 = boxes[n * 4 + 2] * scale, y2 = boxes[n * 4 + 3] * scale;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float cnt = fmaxf((float)(gh * gw), 1.f);
    // weight of one sample coordinate on integer position `pos` of an axis of length L (forward's clamping rules)
    auto axis_w = [](float s, int pos, int L) -> float {
        if (s < -1.0f || s > (float)L) return 0.f;
        if (s <= 0.f) s = 0.f;
        int lo = (int)s, hi;
        if (lo >= L - 1) { hi = lo = L - 1; s = (float)lo; } else hi = lo + 1;
        const float l = s - (float)lo;
        return (pos == lo ? 1.f - l : 0.f) + (pos == hi ? l : 0.f);
    };
    if (tid < P) {
        float s = 0.f;
        for (int iy = 0; iy < gh; ++iy) s += axis_w(y1 + tid * bh + ((float)iy + 0.5f) * bh / (float)gh, y, H);
        wy[tid] = s;
    }
    for (int i = tid; i < W * P; i += 256) {
        const int x = i / P, pw = i - x * P;
        float s = 0.f;
        for (int ix = 0; ix < gw; ++ix) s += axis_w(x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw, x, W);
        wx[i] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int lo = P, hi = -1;
        for (int k = 0; k < P; ++k) if (wy[k] != 0.f) { if (lo == P) lo = k; hi = k; }
        rng[0] = lo; rng[1] = hi;
    }
    if (tid >= 64 && tid < 64 + W) {                 // W <= 192 checked by the host
        const int x = tid - 64;
        int lo = P, hi = -1;
        for (int k = 0; k < P; ++k) if (wx[x * P + k] != 0.f) { if (lo == P) lo = k; hi = k; }
        rng[2 + 2 * x] = lo; rng[3 + 2 * x] = hi;
    }
    __syncthreads();
    const int plo = rng[0], phi = rng[1];
    if (phi < plo) return;                             // no sample of this RoI touches this row
    const bool flip = flip_w && flip_w[n];
    const int C4 = C >> 2;
    const float inv = 1.f / cnt;
    for (int it = tid; it < W * C4; it += 256) {
        const int x = it / C4, c = (it - x * C4) * 4;
        const int qlo = rng[2 + 2 * x], qhi = rng[3 + 2 * x];
        if (qhi < qlo) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ph = plo; ph <= phi; ++ph) {
            const float a = wy[ph];
            if (a == 0.f) continue;
            for (int pw = qlo; pw <= qhi; ++pw) {
                const float w = a * wx[x * P + pw];
                if (w == 0.f) continue;
                const int ow = flip ? P - 1 - pw : pw;
                acc += w * *reinterpret_cast<const f32x4*>(dy + (((long long)n * P + ph) * P + ow) * ldo + c_off + c);
            }
        }
        f32x4* o = reinterpret_cast<f32x4*>(dfeat + (((long long)n * H + y) * W + x) * C + c);
        *o = *o + acc * inv;
    }
}

// backward of vpho_align_heatmap_nhwc_f32 (VPHO.py:333-346 + the W-flip of :139): out[b,i,j] = bilinear_zero(hm[b]; x(i), y(j)),
// so d hm[b, y0..y0+1, x0..x0+1] += weights * d out[b, i, flip ? S-1-j : j]; fp32 atomics (d hm zero-initialised)
__global__ void align_heatmap_bwd_kernel(const float* __restrict__ dout, int N, int S, int C, const float* __restrict__ bbox,
                                         const float* __restrict__ bbox_rect, const unsigned char* __restrict__ flip_w, float* __restrict__ dhm) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * S * S * C) return;
    const int c = (int)(idx % C);
    long long p = idx / C;
    const int j = (int)(p % S); p /= S;
    const int i = (int)(p % S);
    const int n = (int)(p / S);
    const float relw = (bbox_rect[n * 4 + 2] - bbox_rect[n * 4 + 0]) / (bbox[n * 4 + 2] - bbox[n * 4 + 0]);
    const float relh = (bbox_rect[n * 4 + 3] - bbox_rect[n * 4 + 1]) / (bbox[n * 4 + 3] - bbox[n * 4 + 1]);
    const float gx = ((float)i / (float)(S - 1) * 2.f - 1.f) * relw;
    const float gy = ((float)j / (float)(S - 1) * 2.f - 1.f) * relh;
    const float ix = ((gx + 1.f) * (float)S - 1.f) / 2.f, iy = ((gy + 1.f) * (float)S - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const int oj = (flip_w && flip_w[n]) ? S - 1 - j : j;
    const float g = dout[(((long long)n * S + i) * S + oj) * C + c];
    float* b = dhm + (long long)n * S * S * C + c;
    auto add = [&](int yy, int xx, float w) { if (yy >= 0 && yy < S && xx >= 0 && xx < S) atomicAdd(b + ((long long)yy * S + xx) * C, w * g); };
    add(y0, x0, (1.f - tx) * (1.f - ty)); add(y0, x0 + 1, tx * (1.f - ty));
    add(y0 + 1, x0, (1.f - tx) * ty); add(y0 + 1, x0 + 1, tx * ty);
}

// The same gradient as a gather (the resampling grid is separable: ix depends on i only, iy on j only): a workgroup owns one row yy of
// one image, builds WY[j] (weight of sample row j on yy) and WX[xx][i] in LDS and sums, per pixel and channel, the few samples that
// reach it in a fixed order -- no atomics.
__global__ __launch_bounds__(256) void align_heatmap_bwd_gather_kernel(const float* __restrict__ dout, int S, int C, const float* __restrict__ bbox,
                                                                       const float* __restrict__ bbox_rect, const unsigned char* __restrict__ flip_w,
                                                                       float* __restrict__ dhm) {
    extern __shared__ float sm[];
    float* wy = sm;                        // [S]
    float* wx = sm + S;                    // [S (xx)][S (i)]
    int* rng = reinterpret_cast<int*>(wx + S * S);     // [2] j range, [S][2] i ranges
    const int yy = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const float relw = (bbox_rect[n * 4 + 2] - bbox_rect[n * 4 + 0]) / (bbox[n * 4 + 2] - bbox[n * 4 + 0]);
    const float relh = (bbox_rect[n * 4 + 3] - bbox_rect[n * 4 + 1]) / (bbox[n * 4 + 3] - bbox[n * 4 + 1]);
    auto axis_w = [S](int k, float rel, int pos) -> float {        // weight of sample k of an axis on integer position pos
        const float g = ((float)k / (float)(S - 1) * 2.f - 1.f) * rel;
        const float t = ((g + 1.f) * (float)S - 1.f) / 2.f;
        const float f = floorf(t);
        const int p0 = (int)f;
        const float fr = t - f;
        return (pos == p0 ? 1.f - fr : 0.f) + (pos == p0 + 1 ? fr : 0.f);
    };
    if (tid < S) wy[tid] = axis_w(tid, relh, yy);
    for (int e = tid; e < S * S; e += 256) { const int xx = e / S, i = e - xx * S; wx[e] = axis_w(i, relw, xx); }
    __syncthreads();
    if (tid == 0) {
        int lo = S, hi = -1;
        for (int k = 0; k < S; ++k) if (wy[k] != 0.f) { if (lo == S) lo = k; hi = k; }
        rng[0] = lo; rng[1] = hi;
    }
    if (tid >= 64 && tid < 64 + S) {                   // S <= 192 checked by the host
        const int xx = tid - 64;
        int lo = S, hi = -1;
        for (int k = 0; k < S; ++k) if (wx[xx * S + k] != 0.f) { if (lo == S) lo = k; hi = k; }
        rng[2 + 2 * xx] = lo; rng[3 + 2 * xx] = hi;
    }
    __syncthreads();
    const int jlo = rng[0], jhi = rng[1];
    const bool flip = flip_w && flip_w[n];
    for (int e = tid; e < S * C; e += 256) {
        const int xx = e / C, c = e - xx * C;
        const int ilo = rng[2 + 2 * xx], ihi = rng[3 + 2 * xx];
        float acc = 0.f;
        for (int i = ilo; i <= ihi; ++i) {
            const float a = wx[xx * S + i];
            if (a == 0.f) continue;
            for (int j = jlo; j <= jhi; ++j) {
                const float w = a * wy[j];
                if (w == 0.f) continue;
                const int oj = flip ? S - 1 - j : j;
                acc += w * dout[(((long long)n * S + i) * S + oj) * C + c];
            }
        }
        dhm[(((long long)n * S + yy) * S + xx) * C + c] += acc;
    }
}

// torch.optim.AdamW (single tensor, no amsgrad): decoupled decay, then bias-corrected moments
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                             float lr, float beta1, float beta2, float eps, float wd, float step_size, float sqrt_bc2, float grad_scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gr = g[i] * grad_scale;
    float w = p[i] * (1.f - lr * wd);
    const float mi = beta1 * m[i] + (1.f - beta1) * gr;
    const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = w - step_size * (mi / denom);
}

// the same update for a whole list of tensors in ONE launch: a block updates 1024 consecutive elements of one tensor, found by
// bisecting the tensors' first-block indices (multi-tensor apply; 569 launches per step otherwise)
struct AdamSeg { float* p; const float* g; float* m; float* v; long long n; long long blk0; };
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamSeg* __restrict__ segs, int nseg, float lr, float beta1, float beta2, float eps,
                                                          float wd, float step_size, float sqrt_bc2, float grad_scale) {
    int lo = 0, hi = nseg - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {                                   // last segment with blk0 <= b
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    const AdamSeg sg = segs[lo];
    const long long base = (b - sg.blk0) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i >= sg.n) continue;
        const float gr = sg.g[i] * grad_scale;
        float w = sg.p[i] * (1.f - lr * wd);
        const float mi = beta1 * sg.m[i] + (1.f - beta1) * gr;
        const float vi = beta2 * sg.v[i] + (1.f - beta2) * gr * gr;
        sg.m[i] = mi; sg.v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        sg.p[i] = w - step_size * (mi / denom);
    }
}

}  // namespace

extern "C" int vpho_dsm_prepare_f32(const float* gt_pose, const float* t, const float* z, const float* fourier_W, int bs, int reps, int D, int Dp,
                                    float* x_t, float* emb, float* std_out, void* stream) {
    VPHO_REQUIRE(gt_pose && t && z && fourier_W && x_t && emb && std_out && bs > 0 && reps > 0 && D > 0 && Dp >= D, "vpho_dsm_prepare_f32: bad argument");
    hipLaunchKernelGGL(dsm_prepare_kernel, dim3(bs * reps), dim3(128), 0, (hipStream_t)stream, gt_pose, t, z, fourier_W, bs, bs * reps, D, Dp, x_t, emb, std_out);
    return vpho::check_launch("dsm_prepare_kernel");
}

extern "C" int vpho_plinear2_fwd_f32(const float* h, const float* w2, const float* b2, const float* std_rows, long long rows, int nheads,
                                     float* score, void* stream) {
    VPHO_REQUIRE(h && w2 && b2 && std_rows && score && rows > 0 && nheads > 0, "vpho_plinear2_fwd_f32: bad argument");
    hipLaunchKernelGGL(plinear2_fwd_kernel, dim3(nblk(rows * nheads, 4)), dim3(256), 0, (hipStream_t)stream, h, w2, b2, std_rows, rows, nheads, score);
    return vpho::check_launch("plinear2_fwd_kernel");
}

extern "C" int vpho_dsm_loss_f32(const float* score, const float* z, const float* std_rows, long long rows, int D, int batch_times_reps,
                                 float* dout, double* loss, double* partial_ws, int partial_cap, void* stream) {
    VPHO_REQUIRE(score && z && std_rows && dout && loss && partial_ws && rows > 0 && D > 0 && batch_times_reps > 0 && partial_cap >= 1, "vpho_dsm_loss_f32: bad argument");
    const int nb = std::min(partial_cap, std::min(1024, nblk(rows * D)));
    hipLaunchKernelGGL(dsm_loss_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, score, z, std_rows, rows * D, D, 1.0f / (float)batch_times_reps, dout, partial_ws);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)partial_ws, nb, 1.0 / (double)batch_times_reps, loss);
    return vpho::check_launch("dsm_loss kernels");
}

extern "C" int vpho_mse_loss_f32(const float* pd, const float* gt, long long n, float weight, float* grad, double* loss, double* partial_ws,
                                 int partial_cap, void* stream) {
    VPHO_REQUIRE(pd && gt && grad && loss && partial_ws && n > 0 && partial_cap >= 1, "vpho_mse_loss_f32: bad argument");
    const int nb = std::min(partial_cap, std::min(1024, nblk(n)));
    hipLaunchKernelGGL(mse_loss_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pd, gt, n, (float)(2.0 * (double)weight / (double)n), grad, partial_ws);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)partial_ws, nb, (double)weight / (double)n, loss);
    return vpho::check_launch("mse_loss_kernel");
}

extern "C" int vpho_plinear2_bwd_f32(const float* h, const float* dout, const float* w2, long long rows, int nheads, float* dpre, float* dw2, float* db2,
                                     void* stream) {
    VPHO_REQUIRE(h && dout && w2 && dpre && dw2 && db2 && rows > 0 && nheads > 0, "vpho_plinear2_bwd_f32: bad argument");
    hipLaunchKernelGGL(plinear2_bwd_input_kernel, dim3(nblk(rows * nheads * 256)), dim3(256), 0, (hipStream_t)stream, h, dout, w2, rows, nheads, dpre);
    hipLaunchKernelGGL(plinear2_bwd_weight_kernel, dim3(nheads), dim3(256), 0, (hipStream_t)stream, h, dout, rows, nheads, dw2, db2);
    return vpho::check_launch("plinear2_bwd kernels");
}

extern "C" int vpho_relu_bwd_f32(const float* dy, int ld_dy, const float* y, int ld_y, long long rows, int cols, float* dx, int ld_dx, void* stream) {
    VPHO_REQUIRE(dy && y && dx && rows > 0 && cols > 0 && ld_dy >= cols && ld_y >= cols && ld_dx >= cols, "vpho_relu_bwd_f32: bad argument");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblk(rows * cols)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, y, ld_y, rows, cols, dx, ld_dx);
    return vpho::check_launch("relu_bwd_kernel");
}

extern "C" int vpho_colsum_f32(const float* x, int ld, long long rows, int cols, float* out, void* workspace, void* stream) {
    VPHO_REQUIRE(x && out && workspace && rows > 0 && cols > 0 && ld >= cols, "vpho_colsum_f32: bad argument");
    BnRedArgs ra{x, nullptr, nullptr, nullptr, rows, cols, ld, 2, 0, (double*)workspace};
    const int chunks = launch_col_reduce(ra, (hipStream_t)stream);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(nblk(cols, 32)), dim3(32 * FIN_G), 0, (hipStream_t)stream, (const double*)workspace, chunks, cols, out);
    return vpho::check_launch("colsum kernels");
}

extern "C" int vpho_sum_repeats_f32(const float* x, int ld, int c_off, int bs, int reps, int cols, float* out, void* stream) {
    VPHO_REQUIRE(x && out && bs > 0 && reps > 0 && cols > 0 && c_off >= 0 && ld >= c_off + cols, "vpho_sum_repeats_f32: bad argument");
    hipLaunchKernelGGL(sum_repeats_kernel, dim3(nblk((long long)bs * cols)), dim3(256), 0, (hipStream_t)stream, x, ld, c_off, bs, reps, cols, out);
    return vpho::check_launch("sum_repeats_kernel");
}

extern "C" int vpho_transpose_f32(const float* x, int rows, int cols, int ldx, float* y, int ldy, void* stream) {
    VPHO_REQUIRE(x && y && rows > 0 && cols > 0 && ldx >= cols && ldy >= rows, "vpho_transpose_f32: bad argument");
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ldx, y, ldy);
    return vpho::check_launch("transpose_kernel");
}

extern "C" int vpho_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, float grad_scale, void* stream) {
    VPHO_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "vpho_adamw_f32: bad argument");
    // bias corrections in double on the host, as torch.optim.AdamW computes them (python floats)
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       weight_decay, (float)((double)lr / bc1), (float)std::sqrt(bc2), grad_scale);
    return vpho::check_launch("adamw_kernel");
}

extern "C" int vpho_adamw_multi_f32(const void* segments, int n_segments, long long total_blocks, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, int step, float grad_scale, void* stream) {
    VPHO_REQUIRE(segments && n_segments > 0 && total_blocks > 0 && total_blocks < (1ll << 31) && step >= 1, "vpho_adamw_multi_f32: bad argument");
    static_assert(sizeof(AdamSeg) == 48, "segment record = 4 pointers + 2 int64");
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const AdamSeg*)segments, n_segments, lr, beta1,
                       beta2, eps, weight_decay, (float)((double)lr / bc1), (float)std::sqrt(bc2), grad_scale);
    return vpho::check_launch("adamw_multi_kernel");
}

extern "C" int vpho_im2col_t_f32(const float* x, int N, int H, int W, int Cin, int x_ld, int KH, int KW, int stride, int pad_y, int pad_x,
                                 int OH, int OW, float* out, long long ldo, void* stream) {
    VPHO_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && Cin > 0 && x_ld >= Cin && KH > 0 && KW > 0 && stride > 0 && OH > 0 && OW > 0,
                 "vpho_im2col_t_f32: bad argument");
    const long long P = (long long)N * OH * OW;
    VPHO_REQUIRE(ldo >= P, "vpho_im2col_t_f32: leading dimension %lld < %lld pixels", ldo, P);
    Im2colArgs a{x, N, H, W, Cin, x_ld, KH, KW, stride, pad_y, pad_x, OH, OW, P, ldo, out};
    hipLaunchKernelGGL(im2col_t_kernel, dim3((Cin + 31) / 32, (unsigned)((ldo + 31) / 32), KH * KW), dim3(256), 0, (hipStream_t)stream, a);
    return vpho::check_launch("im2col_t_kernel");
}

extern "C" long long vpho_bn_workspace_bytes(int C) { return C > 0 ? (long long)256 * 2 * C * 8 : -1; }

// the statistics of a [rows][C] matrix from partial sums: double chunks of col_reduce_kernel, or the float partial rows of a convolution's
// epilogue -- finished directly when there are few, through one column reduction of the partial matrix itself (1/64 ... 1/256 of the data) first
static const void* reduce_partials(const float* stats, int P, int C, void* workspace, hipStream_t s, int* chunks, bool* is_double) {
    if (P <= 256) { *chunks = P; *is_double = false; return stats; }
    BnRedArgs ra{stats, nullptr, nullptr, nullptr, P, 2 * C, 2 * C, 2, 0, (double*)workspace};      // [P][2C] -> [chunks][2C] doubles = [chunks][2][C]
    *chunks = launch_col_reduce(ra, s);
    *is_double = true;
    return workspace;
}
static int bn_forward_tail(const float* x, long long rows, int C, int ld, const void* part, int chunks, bool part_double, const float* gamma, const float* beta,
                           float eps, float momentum, float slope, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                           const float* res, float* y, hipStream_t s) {
    if (part_double)
        hipLaunchKernelGGL(bn_finish_stats_kernel<double>, dim3(nblk(C, 32)), dim3(32 * FIN_G), 0, s, (const double*)part, chunks, C, rows, eps, momentum, save_mean, save_invstd,
                           running_mean, running_var);
    else
        hipLaunchKernelGGL(bn_finish_stats_kernel<float>, dim3(nblk(C, 32)), dim3(32 * FIN_G), 0, s, (const float*)part, chunks, C, rows, eps, momentum, save_mean, save_invstd,
                           running_mean, running_var);
    if (C % 4 == 0 && ld % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && aligned16(save_mean) && aligned16(save_invstd) &&
        (!res || aligned16(res)))
        hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(nblk(rows * (C / 4))), dim3(256), 0, s, x, (const float*)save_mean, (const float*)save_invstd, gamma, beta, res, rows, C, ld, slope, y);
    else
        hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(nblk(rows * C)), dim3(256), 0, s, x, (const float*)save_mean, (const float*)save_invstd, gamma, beta, res, rows, C, ld, slope, y);
    return vpho::check_launch("bn_train_forward kernels");
}
static int bn_backward_tail(const float* x, const float* dy, long long rows, int C, int ld, const void* part, int chunks, bool part_double, const float* gamma,
                            const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, hipStream_t s,
                            const float* res = nullptr, float* dx_colsum = nullptr, void* workspace = nullptr) {
    if (part_double) hipLaunchKernelGGL(bn_finish_grads_kernel<double>, dim3(nblk(C, 32)), dim3(32 * FIN_G), 0, s, (const double*)part, chunks, C, dbeta, dgamma);
    else hipLaunchKernelGGL(bn_finish_grads_kernel<float>, dim3(nblk(C, 32)), dim3(32 * FIN_G), 0, s, (const float*)part, chunks, C, dbeta, dgamma);
    if (res || dx_colsum) {
        // (the finishing kernel above has consumed the partial sums: the workspace is free for the column sums of dx)
        const bool vec = C % 4 == 0 && ld % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) && aligned16(save_mean) && aligned16(save_invstd) &&
                         aligned16(dbeta) && aligned16(dgamma) && (!res || aligned16(res));
        BnBackSumArgs a{x, dy, save_mean, save_invstd, gamma, dbeta, dgamma, res, rows, C, ld, 0, dx, (double*)workspace};
        const int ch = col_chunks(rows, C, vec ? 4 : 1, &a.rows_per_chunk);
        const dim3 grid((C + (vec ? 32 : 8) - 1) / (vec ? 32 : 8), ch);
        if (vec) hipLaunchKernelGGL(bn_backward_sum_kernel<4>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(bn_backward_sum_kernel<1>, grid, dim3(256), 0, s, a);
        if (dx_colsum) hipLaunchKernelGGL(colsum_finish_kernel, dim3(nblk(C, 32)), dim3(32 * FIN_G), 0, s, (const double*)workspace, ch, C, dx_colsum);
        return vpho::check_launch("bn_train_backward kernels");
    }
    if (C % 4 == 0 && ld % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) && aligned16(save_mean) && aligned16(save_invstd) &&
        aligned16(dbeta) && aligned16(dgamma))
        hipLaunchKernelGGL(bn_backward_kernel<4>, dim3(nblk(rows * (C / 4))), dim3(256), 0, s, x, dy, save_mean, save_invstd, gamma, (const float*)dbeta, (const float*)dgamma, rows, C, ld, dx);
    else
        hipLaunchKernelGGL(bn_backward_kernel<1>, dim3(nblk(rows * C)), dim3(256), 0, s, x, dy, save_mean, save_invstd, gamma, (const float*)dbeta, (const float*)dgamma, rows, C, ld, dx);
    return vpho::check_launch("bn_train_backward kernels");
}

extern "C" int vpho_bn_train_forward_f32(const float* x, long long rows, int C, int ld, const float* gamma, const float* beta, float eps, float momentum,
                                         float slope, float* running_mean, float* running_var, float* save_mean, float* save_invstd, const float* res,
                                         float* y, void* workspace, void* stream) {
    VPHO_REQUIRE(x && gamma && beta && save_mean && save_invstd && y && workspace && rows > 0 && C > 0 && ld >= C, "vpho_bn_train_forward_f32: bad argument");
    VPHO_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "vpho_bn_train_forward_f32: running_mean/var must come together");
    hipStream_t s = (hipStream_t)stream;
    BnRedArgs ra{x, nullptr, nullptr, nullptr, rows, C, ld, 0, 0, (double*)workspace};
    const int chunks = launch_col_reduce(ra, s);
    return bn_forward_tail(x, rows, C, ld, workspace, chunks, true, gamma, beta, eps, momentum, slope, running_mean, running_var, save_mean, save_invstd, res, y, s);
}

extern "C" int vpho_bn_train_forward_stats_f32(const float* x, long long rows, int C, int ld, const float* stats, int stats_rows, const float* gamma,
                                               const float* beta, float eps, float momentum, float slope, float* running_mean, float* running_var,
                                               float* save_mean, float* save_invstd, const float* res, float* y, void* workspace, void* stream) {
    VPHO_REQUIRE(x && stats && stats_rows > 0 && gamma && beta && save_mean && save_invstd && y && workspace && rows > 0 && C > 0 && ld >= C,
                 "vpho_bn_train_forward_stats_f32: bad argument");
    VPHO_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "vpho_bn_train_forward_stats_f32: running_mean/var must come together");
    hipStream_t s = (hipStream_t)stream;
    int chunks; bool dbl;
    const void* part = reduce_partials(stats, stats_rows, C, workspace, s, &chunks, &dbl);
    return bn_forward_tail(x, rows, C, ld, part, chunks, dbl, gamma, beta, eps, momentum, slope, running_mean, running_var, save_mean, save_invstd, res, y, s);
}

extern "C" int vpho_bn_train_backward_f32(const float* x, const float* dy, long long rows, int C, int ld, const float* gamma, const float* save_mean,
                                          const float* save_invstd, float* dx, float* dgamma, float* dbeta, void* workspace, void* stream) {
    VPHO_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace && rows > 0 && C > 0 && ld >= C, "vpho_bn_train_backward_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    BnRedArgs ra{x, dy, save_mean, save_invstd, rows, C, ld, 1, 0, (double*)workspace};
    const int chunks = launch_col_reduce(ra, s);
    return bn_backward_tail(x, dy, rows, C, ld, workspace, chunks, true, gamma, save_mean, save_invstd, dx, dgamma, dbeta, s);
}

extern "C" int vpho_bn_train_backward_stats_f32(const float* x, const float* dy, long long rows, int C, int ld, const float* gamma, const float* save_mean,
                                                const float* save_invstd, const float* stats, int stats_rows, const float* res, float* dx, float* dgamma,
                                                float* dbeta, float* dx_colsum, void* workspace, void* stream) {
    VPHO_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace && rows > 0 && C > 0 && ld >= C && (stats == nullptr) == (stats_rows <= 0),
                 "vpho_bn_train_backward_stats_f32: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int chunks; bool dbl;
    const void* part;
    if (stats) part = reduce_partials(stats, stats_rows, C, workspace, s, &chunks, &dbl);
    else {
        BnRedArgs ra{x, dy, save_mean, save_invstd, rows, C, ld, 1, 0, (double*)workspace};
        chunks = launch_col_reduce(ra, s); dbl = true; part = workspace;
    }
    return bn_backward_tail(x, dy, rows, C, ld, part, chunks, dbl, gamma, save_mean, save_invstd, dx, dgamma, dbeta, s, res, dx_colsum, workspace);
}

extern "C" int vpho_lrelu_bwd_f32(const float* dy, const float* y, long long n, float slope, float* dx, void* stream) {
    VPHO_REQUIRE(dy && y && dx && n > 0, "vpho_lrelu_bwd_f32: bad argument");
    if (n % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(dx)) hipLaunchKernelGGL(lrelu_bwd_kernel<4>, dim3(nblk(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, n, slope, dx);
    else hipLaunchKernelGGL(lrelu_bwd_kernel<1>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, dy, y, n, slope, dx);
    return vpho::check_launch("lrelu_bwd_kernel");
}

extern "C" int vpho_add_lrelu_f32(const float* a, const float* b, long long n, float slope, float* y, void* stream) {
    VPHO_REQUIRE(a && b && y && n > 0, "vpho_add_lrelu_f32: bad argument");
    if (n % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(y)) hipLaunchKernelGGL(add_lrelu_kernel<4>, dim3(nblk(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, n, slope, y);
    else hipLaunchKernelGGL(add_lrelu_kernel<1>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, slope, y);
    return vpho::check_launch("add_lrelu_kernel");
}

// Overlapping windows (k > stride: the 3x3 / stride-2 pool behind the stem) make the gather above scan k*k inputs for each of up to
// four windows per input pixel.  Two passes instead: the arg-max position of every window once (one byte per output element), then an
// input pixel compares its own position code with at most four of those bytes.  Same selection rule, same summation order.
__global__ void maxpool_argmax_kernel(const float* __restrict__ x, int N, int H, int W, int C, int k, int stride, int pad, int OH, int OW,
                                      unsigned char* __restrict__ arg) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // 4 consecutive channels of one OUTPUT pixel
    const int CV = C / 4;
    if (i >= (long long)N * OH * OW * CV) return;
    const int c = (int)(i % CV) * 4;
    long long p = i / CV;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const long long n = p / OH;
    const float* xb = x + n * H * W * (long long)C + c;
    float m[4];
    unsigned char code[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { m[v] = -INFINITY; code[v] = 255; }
    bool first = true;
    for (int r = 0; r < k; ++r) {
        const int yy = oy * stride - pad + r;
        if (yy < 0 || yy >= H) continue;
        for (int q = 0; q < k; ++q) {
            const int xx = ox * stride - pad + q;
            if (xx < 0 || xx >= W) continue;
            float xv[4];
            ldv<4>(xb + ((long long)yy * W + xx) * C, xv);
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (xv[v] > m[v] || first) { m[v] = xv[v]; code[v] = (unsigned char)(r * k + q); }
            first = false;
        }
    }
    uchar4 o; o.x = code[0]; o.y = code[1]; o.z = code[2]; o.w = code[3];
    *reinterpret_cast<uchar4*>(arg + ((n * OH + oy) * OW + ox) * (long long)C + c) = o;
}
__global__ void maxpool_bwd_arg_kernel(const unsigned char* __restrict__ arg, const float* __restrict__ dy, int N, int H, int W, int C, int k, int stride,
                                       int pad, int OH, int OW, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int CV = C / 4;
    if (i >= (long long)N * H * W * CV) return;
    const int c = (int)(i % CV) * 4;
    long long p = i / CV;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const long long n = p / H;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = max(0, (iy + pad - k + stride) / stride), oy_hi = min(OH - 1, (iy + pad) / stride);
    const int ox_lo = max(0, (ix + pad - k + stride) / stride), ox_hi = min(OW - 1, (ix + pad) / stride);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int mine = (iy - (oy * stride - pad)) * k + (ix - (ox * stride - pad));
            const long long o = ((n * OH + oy) * OW + ox) * (long long)C + c;
            const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + o);
            float dv[4];
            ldv<4>(dy + o, dv);
            g[0] += a4.x == mine ? dv[0] : 0.f; g[1] += a4.y == mine ? dv[1] : 0.f;
            g[2] += a4.z == mine ? dv[2] : 0.f; g[3] += a4.w == mine ? dv[3] : 0.f;
        }
    stv<4>(dx + ((n * H + iy) * W + ix) * (long long)C + c, g);
}

extern "C" long long vpho_maxpool_bwd_workspace_bytes(int N, int H, int W, int C, int k, int stride, int pad) {
    if (k <= stride || C % 4 != 0 || k * k > 255) return 0;            // non-overlapping windows: the one-pass gather reads each input once
    const long long OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    return (long long)N * OH * OW * C;
}

extern "C" int vpho_maxpool_bwd_ws_nhwc_f32(const float* x, const float* dy, int N, int H, int W, int C, int k, int stride, int pad, float* dx,
                                            void* workspace, void* stream) {
    VPHO_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0, "vpho_maxpool_bwd_ws_nhwc_f32: bad argument");
    if (!workspace || vpho_maxpool_bwd_workspace_bytes(N, H, W, C, k, stride, pad) == 0 || !aligned16(x) || !aligned16(dy) || !aligned16(dx) ||
        ((uintptr_t)workspace & 3))
        return vpho_maxpool_bwd_nhwc_f32(x, dy, N, H, W, C, k, stride, pad, dx, stream);
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(maxpool_argmax_kernel, dim3(nblk((long long)N * OH * OW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, k, stride, pad,
                       OH, OW, (unsigned char*)workspace);
    hipLaunchKernelGGL(maxpool_bwd_arg_kernel, dim3(nblk((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)workspace,
                       dy, N, H, W, C, k, stride, pad, OH, OW, dx);
    return vpho::check_launch("maxpool_bwd_arg_kernel");
}

extern "C" int vpho_maxpool_bwd_nhwc_f32(const float* x, const float* dy, int N, int H, int W, int C, int k, int stride, int pad, float* dx, void* stream) {
    VPHO_REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0, "vpho_maxpool_bwd_nhwc_f32: bad argument");
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    if (C % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx))
        hipLaunchKernelGGL(maxpool_bwd_kernel<4>, dim3(nblk((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, dy, N, H, W, C, k, stride, pad, OH, OW, dx);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<1>, dim3(nblk((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, dy, N, H, W, C, k, stride, pad, OH, OW, dx);
    return vpho::check_launch("maxpool_bwd_kernel");
}

extern "C" int vpho_resize_bilinear_bwd_nhwc_f32(const float* dy, int N, int OH, int OW, int C, int H, int W, float* dx, void* stream) {
    VPHO_REQUIRE(dy && dx && N > 0 && OH > 0 && OW > 0 && C > 0 && H > 0 && W > 0, "vpho_resize_bilinear_bwd_nhwc_f32: bad argument");
    if (C % 4 == 0 && aligned16(dy) && aligned16(dx))
        hipLaunchKernelGGL(resize_bilinear_bwd_kernel<4>, dim3(nblk((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, N, OH, OW, C, H, W, dx);
    else
        hipLaunchKernelGGL(resize_bilinear_bwd_kernel<1>, dim3(nblk((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, N, OH, OW, C, H, W, dx);
    return vpho::check_launch("resize_bilinear_bwd_kernel");
}

extern "C" int vpho_roi_align_bwd_nhwc_f32(const float* dy, int ldo, int c_off, int N, int H, int W, int C, const float* boxes, float spatial_scale,
                                           int out_size, const unsigned char* flip_w, float* dfeat, void* stream) {
    VPHO_REQUIRE(dy && boxes && dfeat && N > 0 && H > 0 && W > 0 && C > 0 && out_size > 0 && ldo >= c_off + C && c_off >= 0, "vpho_roi_align_bwd_nhwc_f32: bad argument");
    const size_t lds = (size_t)(out_size + W * out_size) * sizeof(float) + (size_t)(2 + 2 * W) * sizeof(int);
    static const int force_atomic = getenv("VPHO_ROI_BWD_ATOMIC") ? atoi(getenv("VPHO_ROI_BWD_ATOMIC")) : 0;      // tuning aid
    if (!force_atomic && C % 4 == 0 && ldo % 4 == 0 && c_off % 4 == 0 && aligned16(dy) && aligned16(dfeat) && W <= 192 && out_size <= 256 && lds <= 48 * 1024) {
        hipLaunchKernelGGL(roi_align_bwd_gather_kernel, dim3(H, N), dim3(256), lds, (hipStream_t)stream, dy, ldo, c_off, H, W, C, boxes, spatial_scale, out_size,
                           flip_w, dfeat);
        return vpho::check_launch("roi_align_bwd_gather_kernel");
    }
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(nblk((long long)N * out_size * out_size * C)), dim3(256), 0, (hipStream_t)stream, dy, ldo, c_off, N, H, W, C,
                       boxes, spatial_scale, out_size, flip_w, dfeat);
    return vpho::check_launch("roi_align_bwd_kernel");
}

extern "C" int vpho_align_heatmap_bwd_nhwc_f32(const float* dout, int N, int size, int C, const float* bbox, const float* bbox_rect,
                                               const unsigned char* flip_w, float* dhm, void* stream) {
    VPHO_REQUIRE(dout && bbox && bbox_rect && dhm && N > 0 && size > 1 && C > 0, "vpho_align_heatmap_bwd_nhwc_f32: bad argument");
    const size_t lds = (size_t)(size + size * size) * sizeof(float) + (size_t)(2 + 2 * size) * sizeof(int);
    if (size <= 192 && lds <= 60 * 1024) {
        hipLaunchKernelGGL(align_heatmap_bwd_gather_kernel, dim3(size, N), dim3(256), lds, (hipStream_t)stream, dout, size, C, bbox, bbox_rect, flip_w, dhm);
        return vpho::check_launch("align_heatmap_bwd_gather_kernel");
    }
    hipLaunchKernelGGL(align_heatmap_bwd_kernel, dim3(nblk((long long)N * size * size * C)), dim3(256), 0, (hipStream_t)stream, dout, N, size, C, bbox, bbox_rect, flip_w, dhm);
    return vpho::check_launch("align_heatmap_bwd_kernel");
}
