// Backward of the physics branch of the training step (lib/model/VPHO.py:170-172,205-212):
//   * CrossModule (cross_module.py:120-137): token assembly backward, LayerNorm backward and the attention backward of the
//     post-norm nn.TransformerEncoderLayer whose sequence axis is the BATCH (quirk Q3); the Linear / Conv2d layers around them
//     run on the fp32-MFMA implicit-GEMM kernels (forward, dgrad) and the TN weight-gradient kernel;
//   * HeadPhysics.get_local_force (physics.py:546-557, double soft-max, quirk Q4), from_local_to_global (:362-371) and the five
//     losses of get_loss (:456-500) fused with their hand-derived gradient down to the three MLP outputs.
// Everything here is small (4160 token rows, 64 images): HBM / latency bound, fixed summation orders (no atomics).
#include "common.h"
#include "../../include/vpho_hip.h"
#include <cmath>

namespace {

inline unsigned nblk(long long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// d tokens (bs,65,512) -> d proj_hand / d proj_obj (bs,8,8,256 NHWC; null = detached stream) and d gravity embedding (bs,512).
// Token n of a stream holds channels 8n..8n+7 of the projected map, feature f = (channel % 8) * 64 + pixel (the reference's
// .view(bs, 32, -1) of an NCHW tensor, cross_module.py:125-126); the positional code is an additive constant.
__global__ void cross_tokens_bwd_kernel(const float* __restrict__ dtok, int bs, float* __restrict__ dph, float* __restrict__ dpo,
                                        float* __restrict__ dge) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bs * 65 * 512) return;
    const int f = (int)(i % 512);
    const int tkn = (int)((i / 512) % 65);
    const int b = (int)(i / (512 * 65));
    const float v = dtok[i];
    if (tkn == 64) { if (dge) dge[b * 512 + f] = v; return; }
    float* dst = tkn < 32 ? dph : dpo;
    if (dst) dst[((long long)b * 64 + (f & 63)) * 256 + 8 * (tkn & 31) + (f >> 6)] = v;
}

// y = LayerNorm(x + r) * gamma + beta  ->  d(x + r) and the per-row products dy * xhat (column sums of which are d gamma;
// column sums of dy are d beta).  One wave per row, same mean / variance arithmetic as add_layernorm_kernel.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                            const float* __restrict__ gamma, const float* __restrict__ dy,
                                                            long long rows, int E, float eps, float* __restrict__ dx, float* __restrict__ gxhat) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* a = x + row * E;
    const float* b = r + row * E;
    const float* g = dy + row * E;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += a[i] + b[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)E;
    float v = 0.f;
    for (int i = lane; i < E; i += 64) { const float d = a[i] + b[i] - mean; v += d * d; }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / (float)E + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int i = lane; i < E; i += 64) {
        const float xh = (a[i] + b[i] - mean) * rstd, gg = g[i] * gamma[i];
        m1 += gg; m2 += gg * xh;
    }
    for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
    m1 /= (float)E; m2 /= (float)E;
    for (int i = lane; i < E; i += 64) {
        const float xh = (a[i] + b[i] - mean) * rstd, gg = g[i] * gamma[i];
        dx[row * E + i] = rstd * (gg - m1 - xh * m2);
        gxhat[row * E + i] = g[i] * xh;
    }
}

// Attention backward for sequences of at most 64 positions (the training batch size).  qkv: (S*B rows, 3E), row = s*B + b,
// columns [q | k | v], head h = columns h*hd .. ; d_out (S*B, E) -> dqkv (S*B, 3E).  One workgroup per (b, head):
// P = softmax(q k^T / sqrt(hd)) and dS = P o (dP - rowsum(dP o P)) live in LDS, then thread c owns feature column c.
constexpr int MB_S = 64;
__global__ __launch_bounds__(256) void mha_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, int S, int B, int E,
                                                      int nhead, float* __restrict__ dqkv, const float* __restrict__ drop) {
    __shared__ __attribute__((aligned(16))) float P[MB_S][MB_S + 4], dS[MB_S][MB_S + 4];          // rows 16-byte aligned: read four at a time below
    const int hd = E / nhead, b = blockIdx.x / nhead, h = blockIdx.x % nhead, tid = threadIdx.x;
    const long long rs = (long long)B * 3 * E, ro = (long long)B * E;
    const float* q = qkv + (long long)b * 3 * E + h * hd;
    const float* k = q + E;
    const float* v = q + 2 * E;
    const float* go = dout + (long long)b * E + h * hd;
    const float scl = 1.0f / sqrtf((float)hd);
    // scores q k^T / sqrt(hd) and dP = dO V^T, both 64 x 64 x hd: the four operands pass through LDS in chunks of 16 feature columns
    // (coalesced 64-byte row pieces; round 3 let every thread walk whole rows of q, k, dO, v straight from global memory: 16 MB of
    // 16-byte loads per workgroup at a row stride of B*3E floats, 600 us per launch), a thread accumulates the 4 x 4 block
    // (ti + 16 u, tj + 16 w) in registers.  Same products in the same order (columns ascending): the same bits as before.
    {
        __shared__ float Qc[MB_S][17], Kc[MB_S][17], Gc[MB_S][17], Vc[MB_S][17];
        const int ti = tid >> 4, tj = tid & 15, lr = tid >> 2, lc = (tid & 3) * 4;
        float accA[4][4], accD[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) { accA[u][w] = 0.f; accD[u][w] = 0.f; }
        for (int c0 = 0; c0 < hd; c0 += 16) {
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f}, k4 = q4, g4 = q4, v4 = q4;
            if (lr < S && c0 + lc < hd) {                                     // hd % 4 == 0: a chunk of 4 columns is in or out as a whole
                q4 = *reinterpret_cast<const f32x4*>(q + lr * rs + c0 + lc); k4 = *reinterpret_cast<const f32x4*>(k + lr * rs + c0 + lc);
                g4 = *reinterpret_cast<const f32x4*>(go + lr * ro + c0 + lc); v4 = *reinterpret_cast<const f32x4*>(v + lr * rs + c0 + lc);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { Qc[lr][lc + u] = q4[u] * scl; Kc[lr][lc + u] = k4[u]; Gc[lr][lc + u] = g4[u]; Vc[lr][lc + u] = v4[u]; }
            __syncthreads();
            const int nc = hd - c0 < 16 ? hd - c0 : 16;
            for (int c = 0; c < nc; ++c) {
                float qv[4], kv[4], gv[4], vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { qv[u] = Qc[ti + 16 * u][c]; gv[u] = Gc[ti + 16 * u][c]; kv[u] = Kc[tj + 16 * u][c]; vv[u] = Vc[tj + 16 * u][c]; }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int w = 0; w < 4; ++w) { accA[u][w] += qv[u] * kv[w]; accD[u][w] += gv[u] * vv[w]; }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) { P[ti + 16 * u][tj + 16 * w] = accA[u][w]; dS[ti + 16 * u][tj + 16 * w] = accD[u][w]; }      // dS holds dP for now
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < S; i += 4) {
        const float sc = lane < S ? P[i][lane] : -INFINITY;
        float mx = sc;
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = lane < S ? expf(sc - mx) : 0.f;
        float sum = e;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float p = e / sum;
        // attention dropout: O = (P o M) V  ->  dP = (dO V^T) o M; the dV product below uses P o M
        const float m = (drop && lane < S) ? drop[((long long)blockIdx.x * S + i) * S + lane] : 1.f;
        const float dp = lane < S ? dS[i][lane] * m : 0.f;
        float t = dp * p;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane < S) {
            P[i][lane] = p * m;
            dS[i][lane] = p * (dp - t);
        }
    }
    __syncthreads();
    float* dq = dqkv + (long long)b * 3 * E + h * hd;
    float* dk = dq + E;
    float* dv = dq + 2 * E;
    // P and dS are complete 64 x 64 tables here (rows / columns beyond S are zero: their operands were loaded as zeros), so the three
    // products read them four entries at a time (every lane the same address: a broadcast) -- a quarter of the LDS instructions of the
    // entry-by-entry loops, same sums in the same order
    for (int c = tid; c < hd; c += 256) {
        float col[MB_S];
        // dV[r][c] = sum_i P[i][r] dO[i][c]
#pragma unroll
        for (int i = 0; i < MB_S; ++i) col[i] = i < S ? go[i * ro + c] : 0.f;
        for (int r = 0; r < S; r += 4) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MB_S; ++i) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(&P[i][r]);
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] += p4[u] * col[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (r + u < S) dv[(r + u) * rs + c] = a[u];
        }
        // dQ[r][c] = scl * sum_j dS[r][j] K[j][c]
#pragma unroll
        for (int j = 0; j < MB_S; ++j) col[j] = j < S ? k[j * rs + c] : 0.f;
        for (int r = 0; r < S; ++r) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < MB_S; j += 4) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(&dS[r][j]);
#pragma unroll
                for (int u = 0; u < 4; ++u) a += p4[u] * col[j + u];
            }
            dq[r * rs + c] = a * scl;
        }
        // dK[r][c] = scl * sum_i dS[i][r] Q[i][c]
#pragma unroll
        for (int i = 0; i < MB_S; ++i) col[i] = i < S ? q[i * rs + c] : 0.f;
        for (int r = 0; r < S; r += 4) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MB_S; ++i) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(&dS[i][r]);
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] += p4[u] * col[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (r + u < S) dk[(r + u) * rs + c] = a[u] * scl;
        }
    }
}

// ---- the same backward for LONGER sequences (the sequence axis is the batch, quirk Q3: a per-rank training batch above 64 images).
// Three launches over a workspace of two S x S tables per (b, head) -- raw scores -> P, dP -> dS --, S <= 1024 like the forward kernel
// (misc.hip::mha_kernel):  A  64 x 64 blocks of scl Q K^T and dO V^T (the register blocking of mha_bwd_kernel);  B  one wave per row:
// soft-max, dropout mask, dS = P o (dP - rowsum(dP o P));  C  64-row blocks of dV = P^T dO, dQ = scl dS K, dK = scl dS^T Q, the tables
// staged through LDS 64 x 64 at a time.  Sums ascending in the reduction index, fixed: deterministic.
__global__ __launch_bounds__(256) void mha_bwd_scores_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, int S, int B, int E, int nhead,
                                                             float* __restrict__ Pw, float* __restrict__ dPw) {
    __shared__ float Qc[64][17], Kc[64][17], Gc[64][17], Vc[64][17];
    const int hd = E / nhead, bh = blockIdx.z, b = bh / nhead, h = bh % nhead, tid = threadIdx.x;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const long long rs = (long long)B * 3 * E, ro = (long long)B * E;
    const float* q = qkv + (long long)b * 3 * E + h * hd;
    const float* k = q + E;
    const float* v = q + 2 * E;
    const float* go = dout + (long long)b * E + h * hd;
    const float scl = 1.0f / sqrtf((float)hd);
    const int ti = tid >> 4, tj = tid & 15, lr = tid >> 2, lc = (tid & 3) * 4;
    float accA[4][4], accD[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) { accA[u][w] = 0.f; accD[u][w] = 0.f; }
    for (int c0 = 0; c0 < hd; c0 += 16) {
        f32x4 q4 = {0.f, 0.f, 0.f, 0.f}, k4 = q4, g4 = q4, v4 = q4;
        if (c0 + lc < hd) {
            if (i0 + lr < S) { q4 = *reinterpret_cast<const f32x4*>(q + (i0 + lr) * rs + c0 + lc); g4 = *reinterpret_cast<const f32x4*>(go + (i0 + lr) * ro + c0 + lc); }
            if (j0 + lr < S) { k4 = *reinterpret_cast<const f32x4*>(k + (j0 + lr) * rs + c0 + lc); v4 = *reinterpret_cast<const f32x4*>(v + (j0 + lr) * rs + c0 + lc); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { Qc[lr][lc + u] = q4[u] * scl; Kc[lr][lc + u] = k4[u]; Gc[lr][lc + u] = g4[u]; Vc[lr][lc + u] = v4[u]; }
        __syncthreads();
        const int nc = hd - c0 < 16 ? hd - c0 : 16;
        for (int c = 0; c < nc; ++c) {
            float qv[4], kv[4], gv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { qv[u] = Qc[ti + 16 * u][c]; gv[u] = Gc[ti + 16 * u][c]; kv[u] = Kc[tj + 16 * u][c]; vv[u] = Vc[tj + 16 * u][c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) { accA[u][w] += qv[u] * kv[w]; accD[u][w] += gv[u] * vv[w]; }
        }
        __syncthreads();
    }
    float* Pb = Pw + (long long)bh * S * S;
    float* Db = dPw + (long long)bh * S * S;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int i = i0 + ti + 16 * u, j = j0 + tj + 16 * w;
            if (i < S && j < S) { Pb[(long long)i * S + j] = accA[u][w]; Db[(long long)i * S + j] = accD[u][w]; }
        }
}
// one wave per row (i, bh): lane l owns the columns l, l + 64, ... (<= 16 of them at S <= 1024)
__global__ __launch_bounds__(256) void mha_bwd_softmax_kernel(int S, long long rows, const float* __restrict__ drop, float* __restrict__ Pw, float* __restrict__ dPw) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // = bh * S + i
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* Pr = Pw + row * S;
    float* Dr = dPw + row * S;
    float sc[16], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int j = lane + 64 * t; sc[t] = j < S ? Pr[j] : -INFINITY; mx = fmaxf(mx, sc[t]); }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { sc[t] = lane + 64 * t < S ? expf(sc[t] - mx) : 0.f; sum += sc[t]; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    float dp[16], m[16], tt = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int j = lane + 64 * t;
        sc[t] = sc[t] / sum;
        m[t] = (drop && j < S) ? drop[row * S + j] : 1.f;
        dp[t] = j < S ? Dr[j] * m[t] : 0.f;
        tt += dp[t] * sc[t];
    }
    for (int o = 32; o > 0; o >>= 1) tt += __shfl_xor(tt, o);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int j = lane + 64 * t;
        if (j < S) { Pr[j] = sc[t] * m[t]; Dr[j] = sc[t] * (dp[t] - tt); }
    }
}
// block = 64 rows r of one (b, head); thread = (column c = tid & 63 (+ 64, ...), row group rq = tid >> 6: rows 16 rq .. 16 rq + 15)
__global__ __launch_bounds__(256) void mha_bwd_grads_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, int S, int B, int E, int nhead,
                                                            const float* __restrict__ Pw, const float* __restrict__ dSw, float* __restrict__ dqkv) {
    __shared__ float Pt[64][65], St[64][65], Sr[64][65];     // P[i][r], dS[i][r] (i = reduction chunk, r = block row); dS[r][j] (j = reduction chunk)
    const int hd = E / nhead, bh = blockIdx.y, b = bh / nhead, h = bh % nhead, tid = threadIdx.x;
    const int r0 = blockIdx.x * 64, cl = tid & 63, rq = tid >> 6;
    const long long rs = (long long)B * 3 * E, ro = (long long)B * E;
    const float* q = qkv + (long long)b * 3 * E + h * hd;
    const float* k = q + E;
    const float* go = dout + (long long)b * E + h * hd;
    float* dq = dqkv + (long long)b * 3 * E + h * hd;
    float* dk = dq + E;
    float* dv = dq + 2 * E;
    const float* Pb = Pw + (long long)bh * S * S;
    const float* Sb = dSw + (long long)bh * S * S;
    const float scl = 1.0f / sqrtf((float)hd);
    for (int c0 = 0; c0 < hd; c0 += 64) {
        const int c = c0 + cl;
        float aV[16], aQ[16], aK[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { aV[u] = 0.f; aQ[u] = 0.f; aK[u] = 0.f; }
        for (int x0 = 0; x0 < S; x0 += 64) {                 // reduction chunk: i (dV, dK) or j (dQ) in [x0, x0 + 64)
            __syncthreads();
            for (int e = tid; e < 64 * 64; e += 256) {
                const int a = e >> 6, bb = e & 63;           // tables read along their rows (bb fastest)
                const bool in = x0 + a < S && r0 + bb < S;
                Pt[a][bb] = in ? Pb[(long long)(x0 + a) * S + r0 + bb] : 0.f;
                St[a][bb] = in ? Sb[(long long)(x0 + a) * S + r0 + bb] : 0.f;
                Sr[a][bb] = (r0 + a < S && x0 + bb < S) ? Sb[(long long)(r0 + a) * S + x0 + bb] : 0.f;
            }
            __syncthreads();
            if (c < hd) {
                for (int x = 0; x < 64 && x0 + x < S; ++x) {
                    const float gv = go[(long long)(x0 + x) * ro + c], kv = k[(long long)(x0 + x) * rs + c], qv = q[(long long)(x0 + x) * rs + c];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        aV[u] += Pt[x][16 * rq + u] * gv;
                        aQ[u] += Sr[16 * rq + u][x] * kv;
                        aK[u] += St[x][16 * rq + u] * qv;
                    }
                }
            }
        }
        if (c < hd) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = r0 + 16 * rq + u;
                if (r < S) { dv[(long long)r * rs + c] = aV[u]; dq[(long long)r * rs + c] = aQ[u] * scl; dk[(long long)r * rs + c] = aK[u] * scl; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ physics losses
struct PhysArgs {
    const float *scale_raw, *logits, *com;     // fc_scale output [bs*32], fc_weight logits [bs*32][8], fc_CoM output [bs*32][3]
    const float* anchor;                       // [8][3] friction-cone anchors (xy NOT yet scaled by the friction coefficient)
    const float *frame, *point;                // from_local_to_global of the GROUND-TRUTH vertices: [bs][32][3][3] (columns x,y,z), [bs][32][3]
    const float *gt_local, *gravity, *gt_com;  // [bs][32][3], [bs][3], [bs][3] (already in the flipped frame)
    const unsigned char* grasped;              // [bs]
    float w[5];                                // weights of force / gravity / torque / supervised / CoM loss
    float friction; int bs;
    float *force_local, *d_scale, *d_logits, *d_com;
    double* partial;                           // [bs][5] un-normalised per-image loss terms
};

__device__ inline float wsum32(float v) {                  // sum over the 32 anchors of an image (lanes 0..31 of the wave)
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(64) void physics_loss_kernel(const PhysArgs a) {
    const int b = blockIdx.x, an = threadIdx.x & 31;
    const bool live = threadIdx.x < 32;
    const long long r = (long long)b * 32 + an;
    const float gr = a.grasped[b] ? 1.f : 0.f, nb = (float)a.bs;
    // forward: double soft-max -> friction-cone mixture -> unit direction x |scale|          (physics.py:546-557,659-664)
    float u[8], v8[8], w8[8];
    float mx = -INFINITY;
    for (int i = 0; i < 8; ++i) { u[i] = a.logits[r * 8 + i]; mx = fmaxf(mx, u[i]); }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) { v8[i] = expf(u[i] - mx); s += v8[i]; }
    for (int i = 0; i < 8; ++i) v8[i] /= s;
    mx = -INFINITY;
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, v8[i]);
    s = 0.f;
    for (int i = 0; i < 8; ++i) { w8[i] = expf(v8[i] - mx); s += w8[i]; }
    for (int i = 0; i < 8; ++i) w8[i] /= s;
    float A[8][3], d[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) {
        A[i][0] = a.anchor[i * 3 + 0] * a.friction; A[i][1] = a.anchor[i * 3 + 1] * a.friction; A[i][2] = a.anchor[i * 3 + 2];
        for (int c = 0; c < 3; ++c) d[c] += w8[i] * A[i][c];
    }
    const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), ne = n + 1e-8f;
    const float sr = a.scale_raw[r], sa = fabsf(sr);
    float dir[3], fl[3], fg[3], F[3], rr[3], tq[3];
    for (int c = 0; c < 3; ++c) { dir[c] = d[c] / ne; fl[c] = dir[c] * sa; }
    const float* fr = a.frame + r * 9;                                   // frame[j][i]
    for (int j = 0; j < 3; ++j) fg[j] = fl[0] * fr[j * 3 + 0] + fl[1] * fr[j * 3 + 1] + fl[2] * fr[j * 3 + 2];
    for (int c = 0; c < 3; ++c) { F[c] = wsum32(live ? fg[c] : 0.f); rr[c] = a.point[r * 3 + c] - a.gt_com[b * 3 + c]; }
    const float t0 = rr[1] * fg[2] - rr[2] * fg[1], t1 = rr[2] * fg[0] - rr[0] * fg[2], t2 = rr[0] * fg[1] - rr[1] * fg[0];
    tq[0] = wsum32(live ? t0 : 0.f); tq[1] = wsum32(live ? t1 : 0.f); tq[2] = wsum32(live ? t2 : 0.f);
    const float* g = a.gravity + b * 3;
    const float R0 = F[0] + g[0], R1 = F[1] + g[1], R2 = F[2] + g[2];
    const float cosp = F[0] * g[0] + F[1] * g[1] + F[2] * g[2] + 1.f;
    float sup = 0.f, cm = 0.f, dfl[3], dcom[3];
    for (int c = 0; c < 3; ++c) {
        const float e = fl[c] - a.gt_local[r * 3 + c], ec = a.com[r * 3 + c] - a.gt_com[b * 3 + c];
        sup += e * e; cm += ec * ec;
        dfl[c] = a.w[3] * 2.f * e / (nb * 96.f);
        dcom[c] = a.w[4] * 2.f * ec / (nb * 96.f);
    }
    sup = wsum32(live ? sup : 0.f); cm = wsum32(live ? cm : 0.f);
    if (threadIdx.x == 0) {
        double* p = a.partial + (long long)b * 5;
        p[0] = (double)gr * gr * ((double)R0 * R0 + (double)R1 * R1 + (double)R2 * R2);          // (|F + g| * grasped)^2
        p[1] = (double)gr * gr * (double)cosp * cosp;
        p[2] = (double)gr * gr * ((double)tq[0] * tq[0] + (double)tq[1] * tq[1] + (double)tq[2] * tq[2]);
        p[3] = sup; p[4] = cm;
    }
    if (!live) return;
    // backward
    const float kf = a.w[0] * 2.f * gr * gr / nb, kg = a.w[1] * 2.f * gr * gr * cosp / nb, kt = a.w[2] * 2.f * gr * gr / nb;
    const float dF[3] = {kf * R0 + kg * g[0], kf * R1 + kg * g[1], kf * R2 + kg * g[2]};
    const float dT[3] = {kt * tq[0], kt * tq[1], kt * tq[2]};
    // d fg = dF + dT x r
    const float dfg[3] = {dF[0] + dT[1] * rr[2] - dT[2] * rr[1], dF[1] + dT[2] * rr[0] - dT[0] * rr[2], dF[2] + dT[0] * rr[1] - dT[1] * rr[0]};
    for (int i = 0; i < 3; ++i) dfl[i] += fr[0 * 3 + i] * dfg[0] + fr[1 * 3 + i] * dfg[1] + fr[2 * 3 + i] * dfg[2];
    const float dsa = dfl[0] * dir[0] + dfl[1] * dir[1] + dfl[2] * dir[2];
    a.d_scale[r] = sr > 0.f ? dsa : (sr < 0.f ? -dsa : 0.f);
    float ddir[3], dd[3];
    for (int c = 0; c < 3; ++c) ddir[c] = dfl[c] * sa;
    const float proj = d[0] * ddir[0] + d[1] * ddir[1] + d[2] * ddir[2];
    for (int c = 0; c < 3; ++c) dd[c] = ddir[c] / ne - (n > 0.f ? d[c] * proj / (n * ne * ne) : 0.f);
    float dw[8], acc = 0.f;
    for (int i = 0; i < 8; ++i) { dw[i] = A[i][0] * dd[0] + A[i][1] * dd[1] + A[i][2] * dd[2]; acc += dw[i] * w8[i]; }
    float dv[8], acc2 = 0.f;
    for (int i = 0; i < 8; ++i) { dv[i] = w8[i] * (dw[i] - acc); acc2 += dv[i] * v8[i]; }
    for (int i = 0; i < 8; ++i) a.d_logits[r * 8 + i] = v8[i] * (dv[i] - acc2);
    for (int c = 0; c < 3; ++c) { a.d_com[r * 3 + c] = dcom[c]; a.force_local[r * 3 + c] = fl[c]; }
}

// losses[k] = w[k] * norm[k] * sum_b partial[b][k], images in order
__global__ void physics_loss_finish_kernel(const double* __restrict__ partial, int bs, PhysArgs a, double* __restrict__ losses) {
    const int k = threadIdx.x;
    if (k >= 5) return;
    double s = 0.0;
    for (int b = 0; b < bs; ++b) s += partial[(long long)b * 5 + k];
    const double norm = k < 3 ? 1.0 / bs : 1.0 / ((double)bs * 96.0);
    losses[k] = (double)a.w[k] * norm * s;
}

}  // namespace

extern "C" int vpho_cross_tokens_bwd_f32(const float* dtok, int bs, float* d_proj_hand, float* d_proj_obj, float* d_grav_emb, void* stream) {
    VPHO_REQUIRE(dtok && bs > 0, "vpho_cross_tokens_bwd_f32: bad argument");
    hipLaunchKernelGGL(cross_tokens_bwd_kernel, dim3(nblk((long long)bs * 65 * 512)), dim3(256), 0, (hipStream_t)stream, dtok, bs, d_proj_hand, d_proj_obj, d_grav_emb);
    return vpho::check_launch("cross_tokens_bwd_kernel");
}

extern "C" int vpho_layernorm_bwd_f32(const float* x, const float* r, const float* gamma, const float* dy, long long rows, int E, float eps,
                                      float* dx, float* dy_xhat, void* stream) {
    VPHO_REQUIRE(x && r && gamma && dy && dx && dy_xhat && rows > 0 && E > 0, "vpho_layernorm_bwd_f32: bad argument");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, r, gamma, dy, rows, E, eps, dx, dy_xhat);
    return vpho::check_launch("layernorm_bwd_kernel");
}

constexpr int MB_S_MAX = 1024;
extern "C" long long vpho_mha_bwd_workspace_bytes(int S, int B, int nhead) {
    if (S <= 0 || B <= 0 || nhead <= 0 || S > MB_S_MAX) return -1;
    return S <= MB_S ? 0 : 2ll * B * nhead * S * S * 4;
}

extern "C" int vpho_mha_bwd_ws_f32(const float* qkv, const float* d_out, int S, int B, int E, int nhead, const float* drop_mask, float* dqkv,
                                   void* workspace, void* stream) {
    VPHO_REQUIRE(qkv && d_out && dqkv && S > 0 && B > 0 && nhead > 0 && E % nhead == 0 && (E / nhead) % 4 == 0, "vpho_mha_bwd_f32: bad argument");
    VPHO_REQUIRE(S <= MB_S_MAX, "vpho_mha_bwd_f32: sequence (= batch, quirk Q3) of %d exceeds the %d positions of the attention kernels", S, MB_S_MAX);
    hipStream_t s = (hipStream_t)stream;
    if (S <= MB_S) {
        hipLaunchKernelGGL(mha_bwd_kernel, dim3(B * nhead), dim3(256), 0, s, qkv, d_out, S, B, E, nhead, dqkv, drop_mask);
        return vpho::check_launch("mha_bwd_kernel");
    }
    VPHO_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0, "vpho_mha_bwd_ws_f32: %d positions need the workspace of vpho_mha_bwd_workspace_bytes", S);
    VPHO_REQUIRE((long long)B * nhead <= 65535, "vpho_mha_bwd_ws_f32: B * nhead = %lld exceeds the grid", (long long)B * nhead);
    float* Pw = (float*)workspace;
    float* Dw = Pw + (long long)B * nhead * S * S;
    const int nb = (S + 63) / 64;
    hipLaunchKernelGGL(mha_bwd_scores_kernel, dim3(nb, nb, B * nhead), dim3(256), 0, s, qkv, d_out, S, B, E, nhead, Pw, Dw);
    const long long rows = (long long)B * nhead * S;
    hipLaunchKernelGGL(mha_bwd_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, S, rows, drop_mask, Pw, Dw);
    hipLaunchKernelGGL(mha_bwd_grads_kernel, dim3(nb, B * nhead), dim3(256), 0, s, qkv, d_out, S, B, E, nhead, (const float*)Pw, (const float*)Dw, dqkv);
    return vpho::check_launch("mha_bwd kernels (long sequence)");
}

extern "C" int vpho_mha_bwd_f32(const float* qkv, const float* d_out, int S, int B, int E, int nhead, const float* drop_mask, float* dqkv,
                                void* stream) {
    VPHO_REQUIRE(S <= MB_S, "vpho_mha_bwd_f32: sequence (= batch, quirk Q3) of %d exceeds the %d positions of the one-launch kernel: vpho_mha_bwd_ws_f32 takes up to %d", S, MB_S, MB_S_MAX);
    return vpho_mha_bwd_ws_f32(qkv, d_out, S, B, E, nhead, drop_mask, dqkv, nullptr, stream);
}

extern "C" int vpho_physics_loss_f32(const float* scale_raw, const float* logits, const float* com, const float* anchor, float friction,
                                     const float* frame, const float* point, const float* gt_force_local, const float* gravity,
                                     const float* gt_com, const unsigned char* is_grasped, const float* weights5, int bs,
                                     float* force_local, float* d_scale, float* d_logits, float* d_com, double* losses5, double* partial_ws,
                                     void* stream) {
    VPHO_REQUIRE(scale_raw && logits && com && anchor && frame && point && gt_force_local && gravity && gt_com && is_grasped && weights5 &&
                 force_local && d_scale && d_logits && d_com && losses5 && partial_ws && bs > 0, "vpho_physics_loss_f32: bad argument");
    PhysArgs a;
    a.scale_raw = scale_raw; a.logits = logits; a.com = com; a.anchor = anchor; a.frame = frame; a.point = point; a.gt_local = gt_force_local;
    a.gravity = gravity; a.gt_com = gt_com; a.grasped = is_grasped; a.friction = friction; a.bs = bs;
    for (int i = 0; i < 5; ++i) a.w[i] = weights5[i];
    a.force_local = force_local; a.d_scale = d_scale; a.d_logits = d_logits; a.d_com = d_com; a.partial = partial_ws;
    hipLaunchKernelGGL(physics_loss_kernel, dim3(bs), dim3(64), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(physics_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)partial_ws, bs, a, losses5);
    return vpho::check_launch("physics_loss kernels");
}
