// Candidate aggregation kernels: heat-map scoring of projected joints / key-points with
// bicubic look-ups, wavefront-shuffle top-k, weighted quaternion means (Markley: top eigenvector of a symmetric 4x4),
// CPF anchors, pseudo-force scores against the object point cloud.  Everything here is latency/HBM-bound integer and
// small-vector work: one wavefront per (image, finger) or per (image, candidate), tables staged through LDS.
//
// Top-k order: larger value first, ties broken by the smaller candidate index (torch.topk leaves ties unspecified).
#include "common.h"
#include <cstdlib>
#include "rot.h"
#include "../../include/vpho_hip.h"

namespace {

inline int nblocks(long long n, int bs = 256) { return (int)((n + bs - 1) / bs); }

// ---------------------------------------------------------------------------------------- projection + bicubic look-up
__device__ inline float cubic1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ inline float cubic2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// F.grid_sample(mode='bicubic', padding_mode='zeros', align_corners=False) of one plane at normalised (gx, gy)
__device__ inline float bicubic_zero(const float* __restrict__ plane, int H, int W, float gx, float gy) {
    const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    const int x0 = (int)fx, y0 = (int)fy;
    const float cx[4] = {cubic2(tx + 1.f), cubic1(tx), cubic1(1.f - tx), cubic2(2.f - tx)};
    const float cy[4] = {cubic2(ty + 1.f), cubic1(ty), cubic1(1.f - ty), cubic2(2.f - ty)};
    float acc = 0.f;
    float rows[4];
    for (int i = 0; i < 4; ++i) {
        const int yy = y0 - 1 + i;
        float r = 0.f;
        for (int j = 0; j < 4; ++j) {
            const int xx = x0 - 1 + j;
            const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? plane[yy * W + xx] : 0.f;
            r += v * cx[j];
        }
        rows[i] = r;
    }
    for (int i = 0; i < 4; ++i) acc += rows[i] * cy[i];
    return acc;
}

// pinhole projection (aggregation.py:24-32) then normalisation to the bbox (aggregation.py:201-204)
__device__ inline void project_norm(const float* p3, const float* K, const float* bbox, float& gx, float& gy) {
    const float u = p3[0] * K[0] + p3[1] * K[1] + p3[2] * K[2];
    const float v = p3[0] * K[3] + p3[1] * K[4] + p3[2] * K[5];
    const float w = p3[0] * K[6] + p3[1] * K[7] + p3[2] * K[8];
    const float px = u / w - bbox[0], py = v / w - bbox[1];
    gx = 2.f * px / (bbox[2] - bbox[0]) - 1.f;
    gy = 2.f * py / (bbox[3] - bbox[1]) - 1.f;
}

// Round 6: the same look-up with the coordinate chain in DOUBLE -- root + joint, pinhole projection, normalisation to the box, grid
// coordinate, the eight cubic weights and the 16-tap sum; the result is rounded to fp32 once.  Why: the end-to-end fp64 judge
// (oracle/judge_fp64.py) found the scores of the cascade's picks 1.5 x farther (rms) from the float64 scores than the reference's own
// fp32 arithmetic at every level (1.6e-6 against 1.1e-6 of the score scale at level 0, 5.0e-6 against 3.2e-6 at level 1), and the lists
// flipping against the float64 order more often (32 against 19 of 256 images).  The noise is born where a 0.7-m camera coordinate is
// projected to a pixel of a 64 x 64 map in fp32 (each rounding of u = fx X + cx Z is 1e-5 px; mul + add without contraction round five
// times where torch's matmul rounds three) and multiplied by the heat-map's slope; ~40 fp64 operations per look-up beside 16 loads cost
// nothing.  VPHO_SCORE_FP32=1 keeps the fp32 chain (A/B aid for the judge).
__device__ inline double cubic1d(double x) { const double A = -0.75; return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0; }
__device__ inline double cubic2d(double x) { const double A = -0.75; return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A; }
__device__ inline float bicubic_zero_d(const float* __restrict__ plane, int H, int W, double gx, double gy) {
    const double ix = ((gx + 1.0) * (double)W - 1.0) / 2.0, iy = ((gy + 1.0) * (double)H - 1.0) / 2.0;
    const double fx = floor(ix), fy = floor(iy);
    const double tx = ix - fx, ty = iy - fy;
    // (a point far outside the map: every tap is outside, the value is 0 -- and the casts below stay in range)
    if (!(fx > -4.0 && fx < (double)W + 4.0 && fy > -4.0 && fy < (double)H + 4.0)) return 0.f;
    const int x0 = (int)fx, y0 = (int)fy;
    const double cx[4] = {cubic2d(tx + 1.0), cubic1d(tx), cubic1d(1.0 - tx), cubic2d(2.0 - tx)};
    const double cy[4] = {cubic2d(ty + 1.0), cubic1d(ty), cubic1d(1.0 - ty), cubic2d(2.0 - ty)};
    double acc = 0.0;
    for (int i = 0; i < 4; ++i) {
        const int yy = y0 - 1 + i;
        double r = 0.0;
        for (int j = 0; j < 4; ++j) {
            const int xx = x0 - 1 + j;
            const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? plane[yy * W + xx] : 0.f;
            r += (double)v * cx[j];
        }
        acc += r * cy[i];
    }
    return (float)acc;
}
__device__ inline void project_norm_d(const double* p3, const float* K, const float* bbox, double& gx, double& gy) {
    const double u = p3[0] * (double)K[0] + p3[1] * (double)K[1] + p3[2] * (double)K[2];
    const double v = p3[0] * (double)K[3] + p3[1] * (double)K[4] + p3[2] * (double)K[5];
    const double w = p3[0] * (double)K[6] + p3[1] * (double)K[7] + p3[2] * (double)K[8];
    const double px = u / w - (double)bbox[0], py = v / w - (double)bbox[1];
    gx = 2.0 * px / ((double)bbox[2] - (double)bbox[0]) - 1.0;
    gy = 2.0 * py / ((double)bbox[3] - (double)bbox[1]) - 1.0;
}

// hand: hv[b][c][i] = bicubic(heatmap[b][obs[i]], project(joint[b][c][obs[i]] + root[b]))          (aggregation.py:196-213)
struct ObsList { int n; int idx[21]; };
template <bool F64>
__global__ void hand_heat_kernel(const float* __restrict__ joints, const float* __restrict__ root, const float* __restrict__ Kmat,
                                 const float* __restrict__ bbox, const float* __restrict__ heatmap, int bs, int C, int J, int H, int W,
                                 ObsList obs, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bs * C * obs.n) return;
    const int o = (int)(i % obs.n);
    const long long bc = i / obs.n;
    const int b = (int)(bc / C);
    const int j = obs.idx[o];
    const float* p = joints + (bc * 21 + j) * 3;
    if (F64) {
        const double p3[3] = {(double)p[0] + (double)root[b * 3 + 0], (double)p[1] + (double)root[b * 3 + 1], (double)p[2] + (double)root[b * 3 + 2]};
        double gx, gy;
        project_norm_d(p3, Kmat + b * 9, bbox + b * 4, gx, gy);
        out[i] = bicubic_zero_d(heatmap + ((long long)b * J + j) * H * W, H, W, gx, gy);
        return;
    }
    const float p3[3] = {p[0] + root[b * 3 + 0], p[1] + root[b * 3 + 1], p[2] + root[b * 3 + 2]};
    float gx, gy;
    project_norm(p3, Kmat + b * 9, bbox + b * 4, gx, gy);
    out[i] = bicubic_zero(heatmap + ((long long)b * J + j) * H * W, H, W, gx, gy);
}

// object: score[b][c] = sum_i bicubic(heatmap[b][i], project(flip(R(pose) kpt_i + t + root)))        (aggregation.py:742-776)
// pose: (bs, n, 9) fp64; optional per-image translation override (bs,3) fp64 (aggregation.py:1219-1220)
// F64: the poses are fp64 already (quirk Q5: the sampler returns float64 object poses, the reference casts them with .float(), :753):
// rotation from the 6-D columns, key-point transform, translation + root, flip, projection, look-up and the sum over the key-points all
// in double, one rounding at the end
template <bool F64>
__global__ void obj_heat_kernel(const double* __restrict__ pose, const double* __restrict__ transl_override, const float* __restrict__ root,
                                const float* __restrict__ kpt_tab, const int* __restrict__ obj_id, const unsigned char* __restrict__ is_right,
                                const float* __restrict__ Kmat, const float* __restrict__ bbox, const float* __restrict__ heatmap,
                                int bs, int n, int J, int H, int W, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bs * n) return;
    const int b = (int)(i / n);
    const double* pp = pose + i * 9;
    const float* kp = kpt_tab + (long long)obj_id[b] * J * 3;
    if (F64) {
        double R[9], t[3];
        vpho::rot6d_to_matrix<double>(pp, R);
        for (int k = 0; k < 3; ++k) t[k] = (transl_override ? transl_override[b * 3 + k] : pp[6 + k]) + (double)root[b * 3 + k];
        const double sgn = is_right[b] ? 1.0 : -1.0;
        double acc = 0.0;
        for (int j = 0; j < J; ++j) {
            double p3[3];
            for (int r = 0; r < 3; ++r) p3[r] = ((double)kp[j * 3 + 0] * R[r * 3 + 0] + (double)kp[j * 3 + 1] * R[r * 3 + 1] + (double)kp[j * 3 + 2] * R[r * 3 + 2]) + t[r];
            p3[0] = p3[0] * sgn;
            double gx, gy;
            project_norm_d(p3, Kmat + b * 9, bbox + b * 4, gx, gy);
            acc += (double)bicubic_zero_d(heatmap + ((long long)b * J + j) * H * W, H, W, gx, gy);
        }
        out[i] = (float)acc;
        return;
    }
    float p6[6], R[9], t[3];
    for (int k = 0; k < 6; ++k) p6[k] = (float)pp[k];
    for (int k = 0; k < 3; ++k) t[k] = (float)(transl_override ? transl_override[b * 3 + k] : pp[6 + k]) + root[b * 3 + k];
    vpho::rot6d_to_matrix(p6, R);
    const float sgn = is_right[b] ? 1.f : -1.f;
    float acc = 0.f;
    for (int j = 0; j < J; ++j) {
        float p3[3];
        for (int r = 0; r < 3; ++r) p3[r] = (kp[j * 3 + 0] * R[r * 3 + 0] + kp[j * 3 + 1] * R[r * 3 + 1] + kp[j * 3 + 2] * R[r * 3 + 2]) + t[r];
        p3[0] = p3[0] * sgn;
        float gx, gy;
        project_norm(p3, Kmat + b * 9, bbox + b * 4, gx, gy);
        acc += bicubic_zero(heatmap + ((long long)b * J + j) * H * W, H, W, gx, gy);
    }
    out[i] = acc;
}

// ---------------------------------------------------------------------------------------- wavefront top-k
// One wavefront per row.  Each lane keeps SLOTS (value, index) candidates (8: up to 512 candidates -- the README and cfg4 sizes --,
// 16: up to 1024); k rounds of a 64-lane butterfly arg-max.
constexpr int TOPK_MAX_SLOTS = 16;
template <int TOPK_SLOTS>
__device__ inline void wave_topk(float (&v)[TOPK_SLOTS], int n, int k, int lane, float* val_out, int* idx_out) {
    // v[s] holds element s*64 + lane; `taken` marks elements already emitted (so -inf values are not picked twice)
    unsigned taken = 0;
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < TOPK_SLOTS; ++s) {
            const int id = s * 64 + lane;
            if (id < n && !((taken >> s) & 1u) && (v[s] > best || bi == 0x7fffffff)) { best = v[s]; bi = id; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
        }
        if (lane == 0) { val_out[r] = best; idx_out[r] = bi; }
#pragma unroll
        for (int s = 0; s < TOPK_SLOTS; ++s) if (s * 64 + lane == bi) taken |= 1u << s;
    }
}

// scores: element c of row (b, f) at scores[(b*n + c)*F + f]; outputs [b][f][k]
template <int TOPK_SLOTS>
__global__ __launch_bounds__(64) void topk_kernel(const float* __restrict__ scores, int n, int F, int k,
                                                  float* __restrict__ val, int* __restrict__ idx) {
    const int row = blockIdx.x, b = row / F, f = row % F, lane = threadIdx.x;
    float v[TOPK_SLOTS];
#pragma unroll
    for (int s = 0; s < TOPK_SLOTS; ++s) {
        const int c = s * 64 + lane;
        float sc = c < n ? scores[((long long)b * n + c) * F + f] : -INFINITY;
        if (sc != sc) sc = INFINITY;               // NaN ranks first, as in torch.topk; keeps every index in range
        v[s] = sc;
    }
    wave_topk(v, n, k, lane, val + (long long)row * k, idx + (long long)row * k);
    // a NaN score was RANKED as +inf; the value torch.topk returns for it is the NaN itself (lane 0 wrote val / idx: same lane reads them)
    if (lane == 0)
        for (int r = 0; r < k; ++r)
            if (val[(long long)row * k + r] == INFINITY) val[(long long)row * k + r] = scores[((long long)b * n + idx[(long long)row * k + r]) * F + f];
}

// Rows beyond 1024 candidates (sample_num > 512: the reference has no limit, aggregation.py:217,246,777): one 256-thread workgroup per row
// ranks by COUNTING, like hand_fuse_kernel -- candidate c's rank = the number of candidates that precede it in "larger value first, then
// smaller index" order = its position in the stable descending sort, which is what k rounds of arg-max with that tie rule produce.
__global__ __launch_bounds__(256) void topk_rank_kernel(const float* __restrict__ scores, int n, int F, int k,
                                                        float* __restrict__ val, int* __restrict__ idx) {
    extern __shared__ float tk_v[];
    const int row = blockIdx.x, b = row / F, f = row % F, tid = threadIdx.x;
    for (int c = tid; c < n; c += 256) {
        float sc = scores[((long long)b * n + c) * F + f];
        if (sc != sc) sc = INFINITY;               // NaN ranks first, as in torch.topk
        tk_v[c] = sc;
    }
    __syncthreads();
    for (int c = tid; c < n; c += 256) {
        const float v = tk_v[c];
        int rank = 0;
        for (int j = 0; j < n; ++j) { const float o = tk_v[j]; rank += (o > v || (o == v && j < c)) ? 1 : 0; }
        // (a NaN score was ranked as +inf; the value written is the score itself, as torch.topk returns it)
        if (rank < k) { val[(long long)row * k + rank] = v == INFINITY ? scores[((long long)b * n + c) * F + f] : v; idx[(long long)row * k + rank] = c; }
    }
}

// ---------------------------------------------------------------------------------------- hand cascade
// candidates: [S diffusion | S regression copies] (aggregation.py:120-126); regression copies take the diffusion wrist
// before level 0 (aggregation.py:140-143, quirk Q7)
__global__ void hand_candidates_kernel(const float* __restrict__ diff, int ld_diff, const float* __restrict__ reg, int bs, int S,
                                       float* __restrict__ pose) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bs * 2 * S * 48) return;
    const int e = (int)(i % 48);
    const long long bc = i / 48;
    const int c = (int)(bc % (2 * S)), b = (int)(bc / (2 * S));
    float v;
    if (c < S) v = diff[((long long)b * S + c) * ld_diff + e];
    else v = e < 3 ? diff[((long long)b * S + (c - S)) * ld_diff + e] : reg[b * 48 + e];
    pose[i] = v;
}

// One wavefront per (image b, slot f).  Level 0: a single slot whose score is the SUM of the observed joints and whose
// fused parameter is the wrist (joint 0).  Levels 1-3: slot = finger f, score = MEAN over the n_obs deeper joints of that
// finger (hv column o = l*5 + f), fused parameter = MANO joint `jid[f]`.          (aggregation.py:215-269)
struct FuseArgs {
    const float* hv; int n_obs_total;     // (bs, C, n_obs_total)
    float* pose;                          // (bs, C, 48) in/out
    int bs, C, k, level;
    int jid[5];
    float* val; int* idx;                 // [b][f][k]
    float* topk_pose;                     // [b][k][F][3] gathered axis-angle (aggregation.py:254) or NULL
    float* score_out;                     // [b][C][F] the score of every candidate as ranked here, or NULL
};
// One 256-thread workgroup per (image, slot).  Ranking by COUNTING instead of k rounds of arg-max: thread c holds candidate c (+256, ...)
// and counts the candidates that precede it in "larger value first, then smaller index" order -- every thread reads the same LDS
// word per step (broadcast), no dependent shuffles -- so rank < k IS the position in the stable descending sort (torch.topk's order
// wherever that is defined).  The k picks are then converted to quaternions by k lanes at once; the weighted moment matrix
// sum_r w_r q_r q_r^T is accumulated entry by entry (16 lanes) in ascending r, the same order and the same operations as a serial
// loop, so the fused rotation is bit-identical to it.
constexpr int FUSE_THREADS = 256;
template <int SLOTS>
__global__ __launch_bounds__(FUSE_THREADS) void hand_fuse_kernel(const FuseArgs a) {
    __shared__ __attribute__((aligned(16))) float s_v[SLOTS * FUSE_THREADS];
    __shared__ float s_val[64], s_w[64], s_q[64][4], s_A[16], s_aa[3];
    __shared__ int s_idx[64];
    const int F = a.level == 0 ? 1 : 5;
    const int b = blockIdx.x / F, f = blockIdx.x % F, tid = threadIdx.x;
    const int n_obs = a.level == 0 ? a.n_obs_total : a.n_obs_total / 5;
    float v[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int c = s * FUSE_THREADS + tid;
        float sc = -INFINITY;
        if (c < a.C) {
            const float* h = a.hv + ((long long)b * a.C + c) * a.n_obs_total;
            sc = 0.f;
            if (a.level == 0) { for (int o = 0; o < n_obs; ++o) sc += h[o]; }
            else { for (int l = 0; l < n_obs; ++l) sc += h[l * 5 + f]; sc = sc / (float)n_obs; }
            if (a.score_out) a.score_out[((long long)b * a.C + c) * F + f] = sc;
            if (sc != sc) sc = INFINITY;           // NaN ranks first, as in torch.topk; keeps every index in range
        }
        v[s] = sc;
        s_v[c] = sc;                               // entries past C: -inf with an index above every candidate's
    }
    __syncthreads();
    int rank[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) rank[s] = 0;
    const int n4 = (a.C + 3) & ~3;
    for (int j = 0; j < n4; j += 4) {
        const float4 o = *reinterpret_cast<const float4*>(s_v + j);
        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) rank[s] += (ov[u] > v[s] || (ov[u] == v[s] && j + u < s * FUSE_THREADS + tid)) ? 1 : 0;
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int c = s * FUSE_THREADS + tid;
        if (c < a.C && rank[s] < a.k) { s_val[rank[s]] = v[s]; s_idx[rank[s]] = c; }
    }
    __syncthreads();
    const int joint = a.level == 0 ? 0 : a.jid[f];
    float* P = a.pose + (long long)b * a.C * 48 + joint * 3;
    if (tid < a.k) {
        const int r = tid;
        float vsum = 0.f;
        for (int i = 0; i < a.k; ++i) vsum += s_val[i];
        const float* aa = P + (long long)s_idx[r] * 48;
        const float ax[3] = {aa[0], aa[1], aa[2]};
        float q[4];
        vpho::axis_angle_to_quaternion(ax, q);
        const float sg = q[0] > 0.f ? 1.f : -1.f;
        for (int i = 0; i < 4; ++i) s_q[r][i] = q[i] * sg;
        s_w[r] = (s_val[r] + 1e-8f) / (vsum + 1e-8f);
        a.val[((long long)b * F + f) * a.k + r] = s_val[r];
        a.idx[((long long)b * F + f) * a.k + r] = s_idx[r];
        if (a.topk_pose) { float* tp = a.topk_pose + (((long long)b * a.k + r) * F + f) * 3; tp[0] = ax[0]; tp[1] = ax[1]; tp[2] = ax[2]; }
    }
    __syncthreads();
    if (tid < 16) {
        const int i = tid >> 2, j = tid & 3;
        float acc = 0.f, wsum = 0.f;
        for (int r = 0; r < a.k; ++r) { acc += (s_q[r][i] * s_q[r][j]) * s_w[r]; wsum += s_w[r]; }
        s_A[tid] = acc / wsum;
    }
    __syncthreads();
    if (tid == 0) {
        float A[4][4], qa[4], aa[3];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) A[i][j] = s_A[i * 4 + j];
        vpho::sym4_top_eigenvector(A, qa);
        const float sg = qa[0] > 0.f ? 1.f : -1.f;
        for (int i = 0; i < 4; ++i) qa[i] *= sg;
        vpho::quaternion_to_axis_angle(qa, aa);
        s_aa[0] = aa[0]; s_aa[1] = aa[1]; s_aa[2] = aa[2];
    }
    __syncthreads();
    // broadcast the fused value into every candidate: x*0 + fused (NaN/Inf in x propagate like the reference)
    for (int i = tid; i < a.C * 3; i += FUSE_THREADS) { float* p = P + (long long)(i / 3) * 48 + (i % 3); *p = *p * 0.f + s_aa[i % 3]; }
}

// The same cascade level without the size limits of hand_fuse_kernel (at most 1024 candidates, k <= 64): any candidate count that fits LDS
// (16 000 per image) and any k <= C -- the reference has neither limit (aggregation.py:217,246: torch.topk of 2 x sample_num candidates,
// topk_hand free).  Candidate scores, the k picks and their quaternions live in dynamic LDS; every sum runs in the order of
// hand_fuse_kernel (the weight normaliser over ascending rank, the moment matrix entry by entry over ascending rank), so a launch
// both kernels can serve gives the same bits from either (tests/test_gpu_edge_cases.py).
__global__ __launch_bounds__(FUSE_THREADS) void hand_fuse_any_kernel(const FuseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fa_sm[];
    float* s_v = fa_sm;                                  // [C]
    float* s_val = s_v + ((a.C + 3) & ~3);               // [k]
    float* s_w = s_val + a.k;                            // [k]
    float* s_q = s_w + a.k;                              // [k][4]
    int* s_idx = reinterpret_cast<int*>(s_q + 4 * a.k);  // [k]
    __shared__ float s_A[16], s_aa[3];
    const int F = a.level == 0 ? 1 : 5;
    const int b = blockIdx.x / F, f = blockIdx.x % F, tid = threadIdx.x;
    const int n_obs = a.level == 0 ? a.n_obs_total : a.n_obs_total / 5;
    for (int c = tid; c < a.C; c += FUSE_THREADS) {
        const float* h = a.hv + ((long long)b * a.C + c) * a.n_obs_total;
        float sc = 0.f;
        if (a.level == 0) { for (int o = 0; o < n_obs; ++o) sc += h[o]; }
        else { for (int l = 0; l < n_obs; ++l) sc += h[l * 5 + f]; sc = sc / (float)n_obs; }
        if (a.score_out) a.score_out[((long long)b * a.C + c) * F + f] = sc;
        if (sc != sc) sc = INFINITY;
        s_v[c] = sc;
    }
    __syncthreads();
    for (int c = tid; c < a.C; c += FUSE_THREADS) {
        const float v = s_v[c];
        int rank = 0;
        for (int j = 0; j < a.C; ++j) { const float o = s_v[j]; rank += (o > v || (o == v && j < c)) ? 1 : 0; }
        if (rank < a.k) { s_val[rank] = v; s_idx[rank] = c; }
    }
    __syncthreads();
    const int joint = a.level == 0 ? 0 : a.jid[f];
    float* P = a.pose + (long long)b * a.C * 48 + joint * 3;
    float vsum = 0.f;
    for (int i = 0; i < a.k; ++i) vsum += s_val[i];
    for (int r = tid; r < a.k; r += FUSE_THREADS) {
        const float* aa = P + (long long)s_idx[r] * 48;
        const float ax[3] = {aa[0], aa[1], aa[2]};
        float q[4];
        vpho::axis_angle_to_quaternion(ax, q);
        const float sg = q[0] > 0.f ? 1.f : -1.f;
        for (int i = 0; i < 4; ++i) s_q[4 * r + i] = q[i] * sg;
        s_w[r] = (s_val[r] + 1e-8f) / (vsum + 1e-8f);
        a.val[((long long)b * F + f) * a.k + r] = s_val[r];
        a.idx[((long long)b * F + f) * a.k + r] = s_idx[r];
        if (a.topk_pose) { float* tp = a.topk_pose + (((long long)b * a.k + r) * F + f) * 3; tp[0] = ax[0]; tp[1] = ax[1]; tp[2] = ax[2]; }
    }
    __syncthreads();
    if (tid < 16) {
        const int i = tid >> 2, j = tid & 3;
        float acc = 0.f, wsum = 0.f;
        for (int r = 0; r < a.k; ++r) { acc += (s_q[4 * r + i] * s_q[4 * r + j]) * s_w[r]; wsum += s_w[r]; }
        s_A[tid] = acc / wsum;
    }
    __syncthreads();
    if (tid == 0) {
        float A[4][4], qa[4], aa[3];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) A[i][j] = s_A[i * 4 + j];
        vpho::sym4_top_eigenvector(A, qa);
        const float sg = qa[0] > 0.f ? 1.f : -1.f;
        for (int i = 0; i < 4; ++i) qa[i] *= sg;
        vpho::quaternion_to_axis_angle(qa, aa);
        s_aa[0] = aa[0]; s_aa[1] = aa[1]; s_aa[2] = aa[2];
    }
    __syncthreads();
    for (int i = tid; i < a.C * 3; i += FUSE_THREADS) { float* p = P + (long long)(i / 3) * 48 + (i % 3); *p = *p * 0.f + s_aa[i % 3]; }
}

// ---------------------------------------------------------------------------------------- CPF anchors / forces
// physics_fn.py:224-257 + physics.py:362-371.  One block per hand: joints21 = vert2joint . (verts + root) then the 32
// anchors (barycentric point, frame from face normal and bone direction) and force_global = frame . force_local[img]
struct AnchorArgs {
    const float* verts; const float* root; const float* force_local; int hands_per_image; long long n_hands;
    vpho_anchor_tables t;
    float* force_point; float* force_global;
};
__global__ __launch_bounds__(256) void anchor_kernel(const AnchorArgs a) {
    __shared__ float jt[21][3];
    const long long hand = blockIdx.x;
    const int img = (int)(hand / a.hands_per_image);
    const float* V = a.verts + hand * 778 * 3;
    const float rt[3] = {a.root[img * 3 + 0], a.root[img * 3 + 1], a.root[img * 3 + 2]};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o = wave; o < 63; o += 4) {
        const int j = o / 3, c = o % 3;
        float s = 0.f;
        for (int v = lane; v < 778; v += 64) s += (V[v * 3 + c] + rt[c]) * a.t.vert2joint[j * 778 + v];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) jt[j][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int an = threadIdx.x;
        float p[3][3];
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) p[k][c] = V[a.t.face_idx[an * 3 + k] * 3 + c] + rt[c];
        float b1[3], b2[3], dy[3], dz[3], dx[3];
        for (int c = 0; c < 3; ++c) { b1[c] = p[1][c] - p[0][c]; b2[c] = p[2][c] - p[0][c]; }
        const int j0 = a.t.skeleton[an * 2 + 0], j1 = a.t.skeleton[an * 2 + 1];
        for (int c = 0; c < 3; ++c) dy[c] = jt[j1][c] - jt[j0][c];
        dz[0] = b1[1] * b2[2] - b1[2] * b2[1]; dz[1] = b1[2] * b2[0] - b1[0] * b2[2]; dz[2] = b1[0] * b2[1] - b1[1] * b2[0];
        float n = sqrtf(dz[0] * dz[0] + dz[1] * dz[1] + dz[2] * dz[2]) + 1e-8f;
        for (int c = 0; c < 3; ++c) dz[c] /= n;
        n = sqrtf(dy[0] * dy[0] + dy[1] * dy[1] + dy[2] * dy[2]) + 1e-8f;
        for (int c = 0; c < 3; ++c) dy[c] /= n;
        dx[0] = dy[1] * dz[2] - dy[2] * dz[1]; dx[1] = dy[2] * dz[0] - dy[0] * dz[2]; dx[2] = dy[0] * dz[1] - dy[1] * dz[0];
        dy[0] = dz[1] * dx[2] - dz[2] * dx[1]; dy[1] = dz[2] * dx[0] - dz[0] * dx[2]; dy[2] = dz[0] * dx[1] - dz[1] * dx[0];
        n = sqrtf(dy[0] * dy[0] + dy[1] * dy[1] + dy[2] * dy[2]) + 1e-8f;
        for (int c = 0; c < 3; ++c) dy[c] /= n;
        const float w1 = a.t.anchor_weight[an * 2 + 0], w2 = a.t.anchor_weight[an * 2 + 1];
        const float* fl = a.force_local + ((long long)img * 32 + an) * 3;
        float* fp = a.force_point + (hand * 32 + an) * 3;
        float* fg = a.force_global + (hand * 32 + an) * 3;
        for (int c = 0; c < 3; ++c) {
            fp[c] = (w1 * b1[c] + w2 * b2[c]) + p[0][c];
            fg[c] = fl[0] * dx[c] + fl[1] * dy[c] + fl[2] * dz[c];
        }
    }
}

// ---------------------------------------------------------------------------------------- object point clouds
// verts_cam[b][v] = flip(R(pose_b) v + t_b + root_b)     (head_object.py:36-67, aggregation.py:1281-1284)
__global__ void obj_verts_kernel(const double* __restrict__ pose, const float* __restrict__ root, const float* __restrict__ vert_tab,
                                 const int* __restrict__ obj_id, const unsigned char* __restrict__ is_right, int bs, int nv, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bs * nv) return;
    const int b = (int)(i / nv), v = (int)(i % nv);
    float p6[6], R[9];
    for (int k = 0; k < 6; ++k) p6[k] = (float)pose[b * 9 + k];
    vpho::rot6d_to_matrix(p6, R);
    const float* p = vert_tab + ((long long)obj_id[b] * nv + v) * 3;
    for (int r = 0; r < 3; ++r) {
        float val = (p[0] * R[r * 3 + 0] + p[1] * R[r * 3 + 1] + p[2] * R[r * 3 + 2]) + ((float)pose[b * 9 + 6 + r] + root[b * 3 + r]);
        if (r == 0 && !is_right[b]) val = val * -1.f;
        out[i * 3 + r] = val;
    }
}

// select_topk_object_by_physics3 (aggregation.py:947-997): one block per (image, candidate); the candidate's 2048 vertices
// are transformed into LDS once (16-byte records: one LDS read per vertex), 32 force points search them (8 lanes per force point,
// shuffle arg-min).  The search compares SQUARED distances and takes the square root of the winner only: sqrt is monotone, so the
// nearest vertex is the same one, with the tie rule "smaller squared distance, then smaller vertex index" (the reference's
// torch.cdist + min leaves the order of equal distances to its kernel; two squared distances that differ in their last bit but
// round to the same root are told apart here and not there).
struct ObjPhysArgs {
    const double* cand; int n;                       // (bs, n, 9)
    const float* root; const float* vert_tab; const float* com_tab; const int* obj_id; const unsigned char* is_right; int nv;
    const float* force_point; const float* force_global;   // (bs,32,3)
    float* score;                                     // (bs, n)
};
__global__ __launch_bounds__(256) void obj_physics_kernel(const ObjPhysArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];               // nv x (x, y, z, -) transformed vertices
    __shared__ float s_d[32], s_r[32][3];
    const int b = blockIdx.x / a.n;
    const double* pp = a.cand + (long long)blockIdx.x * 9;
    float p6[6], R[9], t[3];
    for (int k = 0; k < 6; ++k) p6[k] = (float)pp[k];
    for (int k = 0; k < 3; ++k) t[k] = (float)pp[6 + k] + a.root[b * 3 + k];
    vpho::rot6d_to_matrix(p6, R);
    const float sgn = a.is_right[b] ? 1.f : -1.f;
    const float* tab = a.vert_tab + (long long)a.obj_id[b] * a.nv * 3;
    for (int v = threadIdx.x; v < a.nv; v += blockDim.x) {
        float o[3];
        for (int r = 0; r < 3; ++r) {
            const float val = (tab[v * 3 + 0] * R[r * 3 + 0] + tab[v * 3 + 1] * R[r * 3 + 1] + tab[v * 3 + 2] * R[r * 3 + 2]) + t[r];
            o[r] = r == 0 ? val * sgn : val;
        }
        *reinterpret_cast<float4*>(lds + v * 4) = make_float4(o[0], o[1], o[2], 0.f);
    }
    __syncthreads();
    const int an = threadIdx.x >> 3, sub = threadIdx.x & 7;       // 32 anchors x 8 lanes
    const float* fp = a.force_point + ((long long)b * 32 + an) * 3;
    const float fx = fp[0], fy = fp[1], fz = fp[2];
    float best = INFINITY; int bi = 0x7fffffff;
#pragma unroll 4
    for (int v = sub; v < a.nv; v += 8) {
        const float4 p = *reinterpret_cast<const float4*>(lds + v * 4);
        const float dx = fx - p.x, dy = fy - p.y, dz = fz - p.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < best) { best = d2; bi = v; }                      // v only grows within a lane: the first minimum stays
    }
    for (int o = 4; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    best = sqrtf(best);
    if (sub == 0) {
        // CoM of the candidate: R com + t, flipped
        const float* cm = a.com_tab + (long long)a.obj_id[b] * 3;
        float com[3];
        for (int r = 0; r < 3; ++r) com[r] = (cm[0] * R[r * 3 + 0] + cm[1] * R[r * 3 + 1] + cm[2] * R[r * 3 + 2]) + t[r];
        com[0] *= sgn;
        s_d[an] = best;
        for (int c = 0; c < 3; ++c) s_r[an][c] = (fp[c] - lds[bi * 4 + c]) - com[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float* fg = a.force_global + (long long)b * 32 * 3;
        float nrm[32], nsum = 0.f;
        for (int i = 0; i < 32; ++i) { nrm[i] = sqrtf(fg[i * 3] * fg[i * 3] + fg[i * 3 + 1] * fg[i * 3 + 1] + fg[i * 3 + 2] * fg[i * 3 + 2]); nsum += nrm[i]; }
        float sc = 0.f, L[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < 32; ++i) {
            sc += s_d[i] * (nrm[i] / nsum);
            const float u[3] = {fg[i * 3] / nrm[i], fg[i * 3 + 1] / nrm[i], fg[i * 3 + 2] / nrm[i]};
            L[0] += u[1] * s_r[i][2] - u[2] * s_r[i][1];
            L[1] += u[2] * s_r[i][0] - u[0] * s_r[i][2];
            L[2] += u[0] * s_r[i][1] - u[1] * s_r[i][0];
        }
        const float Ln = sqrtf(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
        a.score[blockIdx.x] = -(sc * Ln);
    }
}

// ---------------------------------------------------------------------------------------- object pose fusion (fp64)
// fuse_topk (aggregation.py:729-740) + average_rot6d (:50-56) in the dtype of the sampler output (fp64, quirk Q5).
// idx/weight may come from two sources selected per image by `pick_b` (is_grasped ? physics : heat-map, :1270-1275).
struct ObjFuseArgs {
    const double* pose; int n;                    // (bs, n, 9)
    const int* idx_a; const float* w_a;           // (bs, k)   w_a == NULL -> uniform 1/k
    const int* idx_b; const float* w_b; const unsigned char* pick_b;   // optional second source
    int bs, k;
    double* fused;                                // (bs, 9)
};
__global__ __launch_bounds__(64) void obj_fuse_kernel(const ObjFuseArgs a) {          // launched with 64 threads: without the bound the compiler budgets for 1024 (128 registers) and spills the fp64 eigen-solve
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.bs) return;
    const bool useb = a.pick_b && a.pick_b[b];
    const int* idx = (useb ? a.idx_b : a.idx_a) + (long long)b * a.k;
    const float* w = useb ? a.w_b : a.w_a;
    double A[4][4] = {{0}}, tr[3] = {0, 0, 0};
    float wsum = 0.f;
    for (int r = 0; r < a.k; ++r) {
        const float wf = w ? w[(long long)b * a.k + r] : (1.0f / (float)a.k);
        const double wr = (double)wf;
        const double* p = a.pose + ((long long)b * a.n + idx[r]) * 9;
        double R[9], q[4];
        vpho::rot6d_to_matrix(p, R);
        vpho::matrix_to_quaternion(R, q);
        const double sg = q[0] > 0 ? 1.0 : -1.0;
        for (int i = 0; i < 4; ++i) q[i] *= sg;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) A[i][j] += (q[i] * q[j]) * wr;
        for (int c = 0; c < 3; ++c) tr[c] += p[6 + c] * wr;
        wsum += wf;                                  // W.sum in the weights' own dtype (fp32)
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) A[i][j] /= (double)wsum;
    double qa[4], R[9];
    vpho::sym4_top_eigenvector(A, qa);
    const double sg = qa[0] > 0 ? 1.0 : -1.0;
    for (int i = 0; i < 4; ++i) qa[i] *= sg;
    vpho::quaternion_to_matrix(qa, R);
    double* o = a.fused + (long long)b * 9;
    for (int c = 0; c < 6; ++c) o[c] = R[c];
    for (int c = 0; c < 3; ++c) o[6 + c] = tr[c];
}

// top-k heat weights (aggregation.py:777-778): w = (val + 1e-8) / (sum(val) + 1e-8)
__global__ void topk_weights_kernel(const float* __restrict__ val, int rows, int k, float* __restrict__ w) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int i = 0; i < k; ++i) s += val[r * k + i];
    for (int i = 0; i < k; ++i) w[r * k + i] = (val[r * k + i] + 1e-8f) / (s + 1e-8f);
}

// cand[b][i*ko + j] = [rot6d of pose[b][rot_idx[b][j]], transl of pose[b][transl_idx[b][i]]]      (aggregation.py:1235-1242)
__global__ void obj_cross_kernel(const double* __restrict__ pose, int n, const int* __restrict__ transl_idx, const int* __restrict__ rot_idx,
                                 int bs, int ko, double* __restrict__ cand) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)bs * ko * ko * 9) return;
    const int e = (int)(i % 9);
    const long long c = i / 9;
    const int j = (int)(c % ko), ii = (int)((c / ko) % ko), b = (int)(c / ((long long)ko * ko));
    const int src = e < 6 ? rot_idx[b * ko + j] : transl_idx[b * ko + ii];
    cand[i] = pose[((long long)b * n + src) * 9 + e];
}

// ---------------------------------------------------------------------------------------- hand physics (aggregation.py:537-626)
// candidates (aggregation.py:1306-1325): agg pose with the distal joints of finger f replaced by the f-th column of the
// k-th best level-3 pose; the last candidate is the aggregated pose itself.  topk_pose: [b][k][5][3]
__constant__ int c_lvl3_joint[5] = {15, 3, 6, 12, 9};   // MANO_PARAMS_LEVEL[3] // 3  (T, I, M, R, P)
__constant__ int c_lvl2_joint[5] = {14, 2, 5, 11, 8};   // MANO_PARAMS_LEVEL[2] // 3
__global__ void hand_phys_candidates_kernel(const float* __restrict__ agg_pose, int ld_agg, const float* __restrict__ betas,
                                            const float* __restrict__ topk_pose, int bs, int k, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = k + 1;
    if (i >= (long long)bs * n * 58) return;
    const int e = (int)(i % 58);
    const long long bc = i / 58;
    const int c = (int)(bc % n), b = (int)(bc / n);
    float v;
    if (e >= 48) v = betas[b * 10 + e - 48];
    else {
        v = agg_pose[(long long)b * ld_agg + e];
        if (c < k) {
            const int joint = e / 3;
            for (int f = 0; f < 5; ++f) if (joint == c_lvl3_joint[f]) v = topk_pose[(((long long)b * k + c) * 5 + f) * 3 + e % 3];
        }
    }
    out[i] = v;
}

// per (image, candidate): 32 anchors x nearest object vertex; finger scores                (aggregation.py:561-596)
__constant__ int c_finger_anchor[5][4] = {{1, 2, 3, 4}, {8, 9, 10, 11}, {14, 15, 16, 17}, {21, 22, 23, 24}, {28, 29, 30, 31}};
__global__ __launch_bounds__(256) void hand_phys_score_kernel(const float* __restrict__ force_point, const float* __restrict__ force_global,
                                                              const float* __restrict__ obj_vert, int nv, int n_cand,
                                                              float* __restrict__ finger_score) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // nv x (x, y, z, -): one LDS read per vertex
    __shared__ float s_d[32];
    const int b = blockIdx.x / n_cand;
    const float* ov = obj_vert + (long long)b * nv * 3;
    for (int v = threadIdx.x; v < nv; v += blockDim.x) *reinterpret_cast<float4*>(lds + v * 4) = make_float4(ov[v * 3], ov[v * 3 + 1], ov[v * 3 + 2], 0.f);
    __syncthreads();
    const int an = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const float* fp = force_point + ((long long)blockIdx.x * 32 + an) * 3;
    const float fx = fp[0], fy = fp[1], fz = fp[2];
    float best = INFINITY;                                               // min of the SQUARED distances; sqrt is monotone and correctly
#pragma unroll 4                                                         // rounded, so sqrt(min d2) == min sqrt(d2) bit for bit
    for (int v = sub; v < nv; v += 8) {
        const float4 p = *reinterpret_cast<const float4*>(lds + v * 4);
        const float dx = fx - p.x, dy = fy - p.y, dz = fz - p.z;
        best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
    for (int o = 4; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o));
    if (sub == 0) s_d[an] = sqrtf(best);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float* fg = force_global + (long long)blockIdx.x * 32 * 3;
        float nrm[32], nsum = 0.f, I[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < 32; ++i) {
            nrm[i] = sqrtf(fg[i * 3] * fg[i * 3] + fg[i * 3 + 1] * fg[i * 3 + 1] + fg[i * 3 + 2] * fg[i * 3 + 2]);
            nsum += nrm[i];
            for (int c = 0; c < 3; ++c) I[c] += fg[i * 3 + c] / nrm[i];
        }
        const float In = sqrtf(I[0] * I[0] + I[1] * I[1] + I[2] * I[2]);
        for (int f = 0; f < 5; ++f) {
            float s = 0.f;
            for (int m = 0; m < 4; ++m) { const int i = c_finger_anchor[f][m]; s += -(((nrm[i] / nsum) * s_d[i]) * In); }
            finger_score[(long long)blockIdx.x * 5 + f] = s;
        }
    }
}

// fuse: per image, per finger: un-weighted quaternion mean of the top-k candidates' proximal+distal joints  (:598-617)
__global__ __launch_bounds__(256) void hand_phys_fuse_kernel(const float* __restrict__ cand, int n_cand, const int* __restrict__ idx, int bs, int k,
                                      float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= bs * 10) return;
    const int b = t / 10, f = (t % 10) / 2, which = t % 2;
    const int joint = which == 0 ? c_lvl2_joint[f] : c_lvl3_joint[f];
    float A[4][4] = {{0}};
    for (int r = 0; r < k; ++r) {
        const float* aa = cand + ((long long)b * n_cand + idx[((long long)b * 5 + f) * k + r]) * 58 + joint * 3;
        float q[4];
        vpho::axis_angle_to_quaternion(aa, q);
        const float sg = q[0] > 0.f ? 1.f : -1.f;
        for (int i = 0; i < 4; ++i) q[i] *= sg;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) A[i][j] += q[i] * q[j];
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) A[i][j] /= (float)k;
    float qa[4], aa[3];
    vpho::sym4_top_eigenvector(A, qa);
    const float sg = qa[0] > 0.f ? 1.f : -1.f;
    for (int i = 0; i < 4; ++i) qa[i] *= sg;
    vpho::quaternion_to_axis_angle(qa, aa);
    for (int e = 0; e < 3; ++e) out[(long long)b * 58 + joint * 3 + e] = aa[e];
}

__global__ void copy_rows_kernel(const float* __restrict__ src, long long ld_src, float* __restrict__ dst, long long ld_dst, int rows, int cols) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[r * ld_dst + c] = src[r * ld_src + c];
}

}  // namespace

#define LAUNCH1D(kernel, total, stream, ...) \
    hipLaunchKernelGGL((kernel), dim3(nblocks(total)), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__)
// VPHO_SCORE_FP32=1: the heat-map scores' coordinate chain in fp32 as until round 5 (A/B aid of the fp64 judge; read per call)
static bool score_fp32() { const char* e = getenv("VPHO_SCORE_FP32"); return e && e[0] == '1'; }

extern "C" int vpho_hand_candidates_f32(const float* diff_pose, int ld_diff, const float* reg_pose, int bs, int S, float* pose, void* stream) {
    VPHO_REQUIRE(diff_pose && reg_pose && pose && bs > 0 && S > 0 && ld_diff >= 48, "vpho_hand_candidates_f32: bad argument");
    LAUNCH1D(hand_candidates_kernel, (long long)bs * 2 * S * 48, stream, diff_pose, ld_diff, reg_pose, bs, S, pose);
    return vpho::check_launch("hand_candidates_kernel");
}

extern "C" int vpho_hand_heat_f32(const float* joints, const float* root, const float* Kmat, const float* bbox, const float* heatmap,
                                  int bs, int C, int J, int H, int W, const int* observe_host, int n_obs, float* out, void* stream) {
    VPHO_REQUIRE(joints && root && Kmat && bbox && heatmap && observe_host && out && bs > 0 && C > 0 && n_obs > 0 && n_obs <= 21 && J >= 21,
                 "vpho_hand_heat_f32: bad argument");
    ObsList ol;
    ol.n = n_obs;
    for (int i = 0; i < n_obs; ++i) { VPHO_REQUIRE(observe_host[i] >= 0 && observe_host[i] < 21, "vpho_hand_heat_f32: joint index out of range"); ol.idx[i] = observe_host[i]; }
    if (score_fp32()) LAUNCH1D(hand_heat_kernel<false>, (long long)bs * C * n_obs, stream, joints, root, Kmat, bbox, heatmap, bs, C, J, H, W, ol, out);
    else LAUNCH1D(hand_heat_kernel<true>, (long long)bs * C * n_obs, stream, joints, root, Kmat, bbox, heatmap, bs, C, J, H, W, ol, out);
    return vpho::check_launch("hand_heat_kernel");
}

extern "C" int vpho_hand_fuse_level_f32(const float* hv, int n_obs, float* pose, int bs, int C, int k, int level,
                                        float* val, int* idx, float* topk_pose, float* score_out, void* stream) {
    VPHO_REQUIRE(hv && pose && val && idx && bs > 0 && C > 0 && level >= 0 && level <= 3, "vpho_hand_fuse_level_f32: bad argument");
    VPHO_REQUIRE(k > 0 && k <= C, "selected index k out of range (topk_hand=%d, candidates=%d)", k, C);
    VPHO_REQUIRE(C <= 16000, "vpho_hand_fuse_level_f32: at most 16000 candidates per image (got %d)", C);
    VPHO_REQUIRE(level == 0 || n_obs % 5 == 0, "vpho_hand_fuse_level_f32: n_obs must be a multiple of 5 for finger levels");
    static const int jid[4][5] = {{0, 0, 0, 0, 0}, {13, 1, 4, 10, 7}, {14, 2, 5, 11, 8}, {15, 3, 6, 12, 9}};   // MANO_PARAMS_LEVEL // 3
    FuseArgs a;
    a.hv = hv; a.n_obs_total = n_obs; a.pose = pose; a.bs = bs; a.C = C; a.k = k; a.level = level;
    for (int f = 0; f < 5; ++f) a.jid[f] = jid[level][f];
    a.val = val; a.idx = idx; a.topk_pose = topk_pose; a.score_out = score_out;
    // algorithmic bytes: the score table read once, the fused joints (3 or 5 x 3 floats) written into every candidate's pose
    vpho::ProfScope prof(vpho::PROF_HAND_FUSE, (hipStream_t)stream, 0.0,
                         (double)bs * C * ((double)n_obs * 4 + (level == 0 ? 3 : 15) * 4) + (double)bs * (level == 0 ? 1 : 5) * k * 8);
    const dim3 grid(bs * (level == 0 ? 1 : 5));
    const char* any_env = getenv("VPHO_FUSE_ANY");                  // 1: the limit-free kernel for every launch (tests: same bits)
    if (C > 64 * TOPK_MAX_SLOTS || k > 64 || (any_env && atoi(any_env))) {
        const size_t lds = (size_t)(((C + 3) & ~3) + 7 * k) * sizeof(float);
        VPHO_REQUIRE(lds <= 150 * 1024, "vpho_hand_fuse_level_f32: %d candidates with k = %d do not fit LDS", C, k);
        VPHO_DYN_LDS(hand_fuse_any_kernel, 150 * 1024);
        hipLaunchKernelGGL(hand_fuse_any_kernel, grid, dim3(FUSE_THREADS), lds, (hipStream_t)stream, a);
        return vpho::check_launch("hand_fuse_any_kernel");
    }
    if (C <= FUSE_THREADS) hipLaunchKernelGGL(hand_fuse_kernel<1>, grid, dim3(FUSE_THREADS), 0, (hipStream_t)stream, a);
    else if (C <= 2 * FUSE_THREADS) hipLaunchKernelGGL(hand_fuse_kernel<2>, grid, dim3(FUSE_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(hand_fuse_kernel<4>, grid, dim3(FUSE_THREADS), 0, (hipStream_t)stream, a);
    return vpho::check_launch("hand_fuse_kernel");
}

extern "C" int vpho_topk_f32(const float* scores, int rows_outer, int n, int F, int k, float* val, int* idx, void* stream) {
    VPHO_REQUIRE(scores && val && idx && rows_outer > 0 && n > 0 && F > 0, "vpho_topk_f32: bad argument");
    VPHO_REQUIRE(k > 0 && k <= n, "selected index k out of range (k=%d, candidates=%d)", k, n);
    VPHO_REQUIRE(n <= 16384, "vpho_topk_f32: at most 16384 candidates per row (got %d)", n);
    if (n <= 512) hipLaunchKernelGGL(topk_kernel<8>, dim3(rows_outer * F), dim3(64), 0, (hipStream_t)stream, scores, n, F, k, val, idx);
    else if (n <= 64 * TOPK_MAX_SLOTS) hipLaunchKernelGGL(topk_kernel<16>, dim3(rows_outer * F), dim3(64), 0, (hipStream_t)stream, scores, n, F, k, val, idx);
    else VPHO_DYN_LDS(topk_rank_kernel, 16384 * sizeof(float));              // n = 16384 is exactly the 64-KB default limit: ask for it
    if (n > 64 * TOPK_MAX_SLOTS) hipLaunchKernelGGL(topk_rank_kernel, dim3(rows_outer * F), dim3(256), (size_t)n * sizeof(float), (hipStream_t)stream, scores, n, F, k, val, idx);
    return vpho::check_launch("topk_kernel");
}

extern "C" int vpho_topk_weights_f32(const float* val, int rows, int k, float* w, void* stream) {
    VPHO_REQUIRE(val && w && rows > 0 && k > 0, "vpho_topk_weights_f32: bad argument");
    LAUNCH1D(topk_weights_kernel, rows, stream, val, rows, k, w);
    return vpho::check_launch("topk_weights_kernel");
}

extern "C" int vpho_obj_heat_score(const double* pose, int n, const double* transl_override, const float* root, const vpho_obj_tables* t,
                                   const int* obj_id, const unsigned char* is_right, const float* Kmat, const float* bbox,
                                   const float* heatmap, int bs, int H, int W, float* score, void* stream) {
    VPHO_REQUIRE(pose && root && t && t->kpt && obj_id && is_right && Kmat && bbox && heatmap && score && bs > 0 && n > 0, "vpho_obj_heat_score: bad argument");
    if (score_fp32()) LAUNCH1D(obj_heat_kernel<false>, (long long)bs * n, stream, pose, transl_override, root, t->kpt, obj_id, is_right, Kmat, bbox, heatmap, bs, n, t->n_kpt, H, W, score);
    else LAUNCH1D(obj_heat_kernel<true>, (long long)bs * n, stream, pose, transl_override, root, t->kpt, obj_id, is_right, Kmat, bbox, heatmap, bs, n, t->n_kpt, H, W, score);
    return vpho::check_launch("obj_heat_kernel");
}

extern "C" int vpho_obj_cross_candidates(const double* pose, int n, const int* transl_idx, const int* rot_idx, int bs, int ko, double* cand, void* stream) {
    VPHO_REQUIRE(pose && transl_idx && rot_idx && cand && bs > 0 && n > 0 && ko > 0, "vpho_obj_cross_candidates: bad argument");
    LAUNCH1D(obj_cross_kernel, (long long)bs * ko * ko * 9, stream, pose, n, transl_idx, rot_idx, bs, ko, cand);
    return vpho::check_launch("obj_cross_kernel");
}

extern "C" int vpho_obj_physics_score(const double* cand, int n, const float* root, const vpho_obj_tables* t, const int* obj_id,
                                      const unsigned char* is_right, const float* force_point, const float* force_global, int bs,
                                      float* score, void* stream) {
    VPHO_REQUIRE(cand && root && t && t->vert && t->com && obj_id && is_right && force_point && force_global && score && bs > 0 && n > 0, "vpho_obj_physics_score: bad argument");
    VPHO_REQUIRE(t->n_vert > 0 && (size_t)t->n_vert * 16 <= 64 * 1024, "vpho_obj_physics_score: object point cloud of %d vertices does not fit LDS", t->n_vert);
    ObjPhysArgs a;
    a.cand = cand; a.n = n; a.root = root; a.vert_tab = t->vert; a.com_tab = t->com; a.obj_id = obj_id; a.is_right = is_right; a.nv = t->n_vert;
    a.force_point = force_point; a.force_global = force_global; a.score = score;
    // algorithmic bytes: the object's vertex table and the 32 force points / forces once per image, 72-byte pose + score per candidate
    vpho::ProfScope prof(vpho::PROF_OBJ_PHYSICS, (hipStream_t)stream, 0.0,
                         (double)bs * ((double)t->n_vert * 12 + 2 * 32 * 12) + (double)bs * n * (72 + 4));
    hipLaunchKernelGGL(obj_physics_kernel, dim3(bs * n), dim3(256), (size_t)t->n_vert * 16, (hipStream_t)stream, a);
    return vpho::check_launch("obj_physics_kernel");
}

extern "C" int vpho_obj_fuse_f64(const double* pose, int n, const int* idx_a, const float* w_a, const int* idx_b, const float* w_b,
                                 const unsigned char* pick_b, int bs, int k, double* fused, void* stream) {
    VPHO_REQUIRE(pose && idx_a && fused && bs > 0 && n > 0 && k > 0, "vpho_obj_fuse_f64: bad argument");
    VPHO_REQUIRE(!pick_b || idx_b, "vpho_obj_fuse_f64: pick_b needs idx_b");
    ObjFuseArgs a;
    a.pose = pose; a.n = n; a.idx_a = idx_a; a.w_a = w_a; a.idx_b = idx_b; a.w_b = w_b; a.pick_b = pick_b; a.bs = bs; a.k = k; a.fused = fused;
    hipLaunchKernelGGL(obj_fuse_kernel, dim3(nblocks(bs, 64)), dim3(64), 0, (hipStream_t)stream, a);
    return vpho::check_launch("obj_fuse_kernel");
}

extern "C" int vpho_obj_verts_f32(const double* pose, const float* root, const vpho_obj_tables* t, const int* obj_id,
                                  const unsigned char* is_right, int bs, float* out, void* stream) {
    VPHO_REQUIRE(pose && root && t && t->vert && obj_id && is_right && out && bs > 0, "vpho_obj_verts_f32: bad argument");
    LAUNCH1D(obj_verts_kernel, (long long)bs * t->n_vert, stream, pose, root, t->vert, obj_id, is_right, bs, t->n_vert, out);
    return vpho::check_launch("obj_verts_kernel");
}

extern "C" int vpho_force_anchor_f32(const vpho_anchor_tables* t, const float* verts, const float* root, const float* force_local,
                                     long long n_hands, int hands_per_image, float* force_point, float* force_global, void* stream) {
    VPHO_REQUIRE(t && t->face_idx && t->anchor_weight && t->vert2joint && t->skeleton && verts && root && force_local && force_point && force_global && n_hands > 0 && hands_per_image > 0,
                 "vpho_force_anchor_f32: bad argument");
    AnchorArgs a;
    a.verts = verts; a.root = root; a.force_local = force_local; a.hands_per_image = hands_per_image; a.n_hands = n_hands; a.t = *t;
    a.force_point = force_point; a.force_global = force_global;
    hipLaunchKernelGGL(anchor_kernel, dim3((unsigned)n_hands), dim3(256), 0, (hipStream_t)stream, a);
    return vpho::check_launch("anchor_kernel");
}

extern "C" int vpho_hand_phys_candidates_f32(const float* agg_pose, int ld_agg, const float* betas, const float* topk_pose, int bs, int k,
                                             float* out, void* stream) {
    VPHO_REQUIRE(agg_pose && betas && topk_pose && out && bs > 0 && k > 0 && ld_agg >= 48, "vpho_hand_phys_candidates_f32: bad argument");
    LAUNCH1D(hand_phys_candidates_kernel, (long long)bs * (k + 1) * 58, stream, agg_pose, ld_agg, betas, topk_pose, bs, k, out);
    return vpho::check_launch("hand_phys_candidates_kernel");
}

extern "C" int vpho_hand_phys_score_f32(const float* force_point, const float* force_global, const float* obj_vert, int n_vert,
                                        int bs, int n_cand, float* finger_score, void* stream) {
    VPHO_REQUIRE(force_point && force_global && obj_vert && finger_score && bs > 0 && n_cand > 0 && n_vert > 0, "vpho_hand_phys_score_f32: bad argument");
    VPHO_REQUIRE((size_t)n_vert * 16 <= 64 * 1024, "vpho_hand_phys_score_f32: object point cloud of %d vertices does not fit LDS", n_vert);
    hipLaunchKernelGGL(hand_phys_score_kernel, dim3(bs * n_cand), dim3(256), (size_t)n_vert * 16, (hipStream_t)stream,
                       force_point, force_global, obj_vert, n_vert, n_cand, finger_score);
    return vpho::check_launch("hand_phys_score_kernel");
}

extern "C" int vpho_hand_phys_fuse_f32(const float* cand, int n_cand, const int* idx, int bs, int k, float* out, void* stream) {
    VPHO_REQUIRE(cand && idx && out && bs > 0 && n_cand > 0 && k > 0 && k <= n_cand, "vpho_hand_phys_fuse_f32: bad argument");
    // start from candidate 0 (aggregation.py:598), then overwrite the 10 fused joints
    LAUNCH1D(copy_rows_kernel, (long long)bs * 58, stream, cand, (long long)n_cand * 58, out, 58LL, bs, 58);
    LAUNCH1D(hand_phys_fuse_kernel, (long long)bs * 10, stream, cand, n_cand, idx, bs, k, out);
    return vpho::check_launch("hand_phys_fuse_kernel");
}
