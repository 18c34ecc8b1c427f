// Device-side hand metrics: MJE / PA-MJE / MVE / PA-MVE of TesterHand.criterion_MJE_PAMJE
// (lib/engine/test.py:657-680) with the Procrustes alignment of transform_fn.rigid_transform_3D_AtoB (:43-58), so that the
// evaluation loop ships 8 floats per image to the all-gather instead of copying (bs,S,778,3) candidates to the host.
// One block per image; reductions and the 3x3 SVD (Jacobi on H^T H) in fp64.  HBM-bound: 24 B read per point.
#include <type_traits>
#include "common.h"
#include "rot.h"
#include <algorithm>
#include "../../include/vpho_hip.h"

namespace {

__device__ inline double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi); eigenvalues descending, eigenvectors in the columns of V
__device__ inline void sym3_eig(double A[3][3], double w[3], double V[3][3]) {
    // all indices are compile-time constants (unrolled pairs, a three-element sorting network on whole columns): registers, no scratch
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (!(off > 1e-300)) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (!(fabs(A[p][q]) > 1e-300)) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
            }
    }
    // eigenvalues descending with their columns: the exchanges of the former index sort (0,1), (0,2), (1,2), each on a strict ">"
    w[0] = A[0][0]; w[1] = A[1][1]; w[2] = A[2][2];
    auto cswap = [&](auto I, auto J) {
        constexpr int i = decltype(I)::value, j = decltype(J)::value;
        const bool sw = w[j] > w[i];
        const double wi = w[i], wj = w[j];
        w[i] = sw ? wj : wi; w[j] = sw ? wi : wj;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double a = V[k][i], b = V[k][j]; V[k][i] = sw ? b : a; V[k][j] = sw ? a : b; }
    };
    cswap(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    cswap(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    cswap(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
}

__global__ __launch_bounds__(256) void hand_metrics_kernel(const float* __restrict__ pd, const float* __restrict__ gt, int n,
                                                           float* __restrict__ mean_err, float* __restrict__ pa_mean_err,
                                                           float* __restrict__ per_point) {
    __shared__ double red[256];
    __shared__ double T[12];            // c*R (9) and t (3)
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* A = pd + (long long)b * n * 3;
    const float* B = gt + (long long)b * n * 3;
    double sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0}, se = 0;
    for (int i = tid; i < n; i += 256) {
        double d2 = 0;
        for (int c = 0; c < 3; ++c) { sa[c] += A[i * 3 + c]; sb[c] += B[i * 3 + c]; const double d = (double)B[i * 3 + c] - A[i * 3 + c]; d2 += d * d; }
        const double e = sqrt(d2);
        se += e;
        if (per_point) per_point[(long long)b * n + i] = (float)e;
    }
    double cA[3], cB[3];
    for (int c = 0; c < 3; ++c) { cA[c] = block_sum(sa[c], red) / n; cB[c] = block_sum(sb[c], red) / n; }
    const double me = block_sum(se, red) / n;
    double h[9] = {0}, va = 0;
    for (int i = tid; i < n; i += 256) {
        double a[3], bb[3];
        for (int c = 0; c < 3; ++c) { a[c] = A[i * 3 + c] - cA[c]; bb[c] = B[i * 3 + c] - cB[c]; va += a[c] * a[c]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) h[r * 3 + c] += a[r] * bb[c];
    }
    double H[3][3];
    for (int k = 0; k < 9; ++k) H[k / 3][k % 3] = block_sum(h[k], red) / n;
    const double varA = block_sum(va, red) / n;
    if (tid == 0) {
        // H = U S V^T; eigen of H^T H gives V and S^2, U = H V S^-1;  R = V U^T (numpy: Vh.T @ U.T)
        double M[3][3], w[3], V[3][3], U[3][3], s[3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { M[i][j] = 0; for (int k = 0; k < 3; ++k) M[i][j] += H[k][i] * H[k][j]; }
        sym3_eig(M, w, V);
        for (int k = 0; k < 3; ++k) s[k] = sqrt(w[k] > 0 ? w[k] : 0.0);
        for (int k = 0; k < 3; ++k) {
            double u[3] = {0, 0, 0};
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) u[i] += H[i][j] * V[j][k];
            const double nu = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            if (k < 2 || nu > 1e-12 * (s[0] + 1e-300)) for (int i = 0; i < 3; ++i) U[i][k] = u[i] / (nu > 0 ? nu : 1.0);
            else {   // rank-deficient H: complete the basis (sign fixed by the determinant rule below)
                U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
                U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
                U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
            }
        }
        auto build = [&](double R[3][3]) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { R[i][j] = 0; for (int k = 0; k < 3; ++k) R[i][j] += V[i][k] * U[j][k]; } };
        double R[3][3];
        build(R);
        const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                           R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
        if (det < 0) { s[2] = -s[2]; for (int i = 0; i < 3; ++i) V[i][2] = -V[i][2]; build(R); }
        const double c = (s[0] + s[1] + s[2]) / varA;
        for (int i = 0; i < 3; ++i) {
            double t = cB[i];
            for (int j = 0; j < 3; ++j) { T[i * 3 + j] = c * R[i][j]; t -= c * R[i][j] * cA[j]; }
            T[9 + i] = t;
        }
    }
    __syncthreads();
    double sp = 0;
    for (int i = tid; i < n; i += 256) {
        double d2 = 0;
        for (int r = 0; r < 3; ++r) {
            const double al = T[r * 3] * A[i * 3] + T[r * 3 + 1] * A[i * 3 + 1] + T[r * 3 + 2] * A[i * 3 + 2] + T[9 + r];
            const double d = (double)B[i * 3 + r] - al;
            d2 += d * d;
        }
        sp += sqrt(d2);
    }
    const double pa = block_sum(sp, red) / n;
    if (tid == 0) { mean_err[b] = (float)me; pa_mean_err[b] = (float)pa; }
}

// obj_9D_to_mat (lib/utils/transform_fn.py:85-90) + root joint (train_diff_hand_obj.py:594-597): [rot6d | t] -> [R | t + root]
__global__ void obj_9d_to_rt_kernel(const double* __restrict__ pose9, const float* __restrict__ root, int n, double* __restrict__ rt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double R[9];
    vpho::rot6d_to_matrix<double>(pose9 + (long long)i * 9, R);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) rt[(long long)i * 12 + r * 4 + c] = R[r * 3 + c];
        rt[(long long)i * 12 + r * 4 + 3] = (double)root[i * 3 + r] + pose9[(long long)i * 9 + 6 + r];
    }
}

// ------------------------------------------------------------------------------------------------ object metrics
// TesterObject (lib/engine/test.py:240-503) per image and single hypothesis: MCE / OCE on the 8 model-box corners (fp64,
// :354-374), MCE2 on the axis-aligned boxes of the transformed sampled vertices (:155-193,398-414), ADD / ADD-S / REP
// (:425-458), their 0.1-diameter and 5-pixel hits (:505-519), Chamfer-L2 and F-score at six thresholds on the full vertex
// set (:460-503).  Transforms are fp64 on the fp64 tables and rounded to fp32 where the reference calls .float(); the two
// nearest-neighbour searches are exact fp32 brute force through LDS tiles (the reference's torch.cdist uses the matmul
// expansion, which is only accurate to ~1e-7 in d^2 at camera-space magnitudes).  HBM/L2-bound: 24 B per table point.
struct ObjWs { float *pd_s, *gt_s, *pd_f, *gt_f, *d_s, *d_p2g, *d_g2p; long long bytes; };

__host__ __device__ inline long long ows_align(long long v) { return (v + 255) / 256 * 256; }

inline ObjWs obj_carve(int n_img, int ns, int vmax, char* base) {
    long long off = 0;
    ObjWs w;
    auto take = [&](long long b) { char* p = base ? base + off : nullptr; off += ows_align(b); return (float*)p; };
    w.pd_s = take((long long)n_img * ns * 12); w.gt_s = take((long long)n_img * ns * 12);
    w.pd_f = take((long long)n_img * vmax * 12); w.gt_f = take((long long)n_img * vmax * 12);
    w.d_s = take((long long)n_img * ns * 4); w.d_p2g = take((long long)n_img * vmax * 4); w.d_g2p = take((long long)n_img * vmax * 4);
    w.bytes = off;
    return w;
}

__device__ inline void rt_apply(const double* rt, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = v[0] * rt[r * 4 + 0] + v[1] * rt[r * 4 + 1] + v[2] * rt[r * 4 + 2] + rt[r * 4 + 3];
}

__global__ __launch_bounds__(256) void obj_transform_kernel(const vpho_obj_metric_tables t, const double* __restrict__ pd_rt,
                                                            const double* __restrict__ gt_rt, const int* __restrict__ obj_id,
                                                            int vmax, ObjWs w) {
    const int b = blockIdx.x, o = obj_id[b];
    const int nf = t.vert_offset[o + 1] - t.vert_offset[o];
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= t.n_sampled + nf) return;
    const bool full = i >= t.n_sampled;
    const int k = full ? i - t.n_sampled : i;
    const double* v = full ? t.verts + ((long long)t.vert_offset[o] + k) * 3 : t.verts_sampled + ((long long)o * t.n_sampled + k) * 3;
    double p[3], g[3];
    rt_apply(pd_rt + (long long)b * 12, v, p);
    rt_apply(gt_rt + (long long)b * 12, v, g);
    float* po = full ? w.pd_f + ((long long)b * vmax + k) * 3 : w.pd_s + ((long long)b * t.n_sampled + k) * 3;
    float* go = full ? w.gt_f + ((long long)b * vmax + k) * 3 : w.gt_s + ((long long)b * t.n_sampled + k) * 3;
    for (int c = 0; c < 3; ++c) { po[c] = (float)p[c]; go[c] = (float)g[c]; }
}

// grid (image, job, chunk of 256 source points): job 0 sampled pd->gt, 1 full pd->gt, 2 full gt->pd
__global__ __launch_bounds__(256) void obj_nn_kernel(const vpho_obj_metric_tables t, const int* __restrict__ obj_id, int vmax, ObjWs w) {
    __shared__ float tile[256 * 3];
    const int b = blockIdx.x, job = blockIdx.y, o = obj_id[b];
    const int n = job == 0 ? t.n_sampled : t.vert_offset[o + 1] - t.vert_offset[o];
    const int ld = job == 0 ? t.n_sampled : vmax;
    const float* src = (job == 0 ? w.pd_s : job == 1 ? w.pd_f : w.gt_f) + (long long)b * ld * 3;
    const float* dst = (job == 0 ? w.gt_s : job == 1 ? w.gt_f : w.pd_f) + (long long)b * ld * 3;
    float* out = (job == 0 ? w.d_s : job == 1 ? w.d_p2g : w.d_g2p) + (long long)b * ld;
    if (blockIdx.z * 256 >= n) return;
    const int i = blockIdx.z * 256 + threadIdx.x;
    const bool live = i < n;
    const float x = live ? src[i * 3] : 0.f, y = live ? src[i * 3 + 1] : 0.f, z = live ? src[i * 3 + 2] : 0.f;
    float best = INFINITY;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int m = min(256, n - j0);
        __syncthreads();
        for (int q = threadIdx.x; q < m * 3; q += 256) tile[q] = dst[(long long)j0 * 3 + q];
        __syncthreads();
        for (int j = 0; j < m; ++j) {
            const float dx = x - tile[j * 3], dy = y - tile[j * 3 + 1], dz = z - tile[j * 3 + 2];
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            best = d2 < best ? d2 : best;
        }
    }
    if (live) out[i] = sqrtf(best);
}

__device__ inline float block_minmax(float v, bool is_max, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] = is_max ? fmaxf(red[tid], red[tid + o]) : fminf(red[tid], red[tid + o]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void obj_metrics_kernel(const vpho_obj_metric_tables t, const double* __restrict__ pd_rt,
                                                          const double* __restrict__ gt_rt, const double* __restrict__ cam,
                                                          const int* __restrict__ obj_id, int vmax, ObjWs w, double* __restrict__ out) {
    __shared__ double red[256];
    __shared__ float redf[256];
    __shared__ double corner[2][8][3];
    const int b = blockIdx.x, tid = threadIdx.x, o = obj_id[b];
    const int ns = t.n_sampled, nf = t.vert_offset[o + 1] - t.vert_offset[o];
    const double* prt = pd_rt + (long long)b * 12;
    const double* grt = gt_rt + (long long)b * 12;
    const double* K = cam + (long long)b * 9;
    if (tid < 8) {
        rt_apply(prt, t.bbox3d + ((long long)o * 8 + tid) * 3, corner[0][tid]);
        rt_apply(grt, t.bbox3d + ((long long)o * 8 + tid) * 3, corner[1][tid]);
    }
    __syncthreads();
    double mce = 0, oce = 0;
    if (tid == 0) {
        double cp[3] = {0, 0, 0}, cg[3] = {0, 0, 0};
        for (int k = 0; k < 8; ++k) {
            double d2 = 0;
            for (int c = 0; c < 3; ++c) { const double d = corner[0][k][c] - corner[1][k][c]; d2 += d * d; cp[c] += corner[0][k][c]; cg[c] += corner[1][k][c]; }
            mce += sqrt(d2);
        }
        mce /= 8;
        double d2 = 0;
        for (int c = 0; c < 3; ++c) { const double d = cp[c] / 8 - cg[c] / 8; d2 += d * d; }
        oce = sqrt(d2);
    }
    // sampled vertices: ADD (fp32 points), REP (fp64 points re-derived from the table), ADD-S, boxes for MCE2
    const float* ps = w.pd_s + (long long)b * ns * 3;
    const float* gs = w.gt_s + (long long)b * ns * 3;
    double s_add = 0, s_rep = 0, s_adds = 0;
    float mn[2][3], mx[2][3];
    for (int c = 0; c < 3; ++c) { mn[0][c] = mn[1][c] = INFINITY; mx[0][c] = mx[1][c] = -INFINITY; }
    for (int i = tid; i < ns; i += 256) {
        const float dx = ps[i * 3] - gs[i * 3], dy = ps[i * 3 + 1] - gs[i * 3 + 1], dz = ps[i * 3 + 2] - gs[i * 3 + 2];
        s_add += (double)sqrtf((dx * dx + dy * dy) + dz * dz);
        s_adds += (double)w.d_s[(long long)b * ns + i];
        for (int c = 0; c < 3; ++c) {
            mn[0][c] = fminf(mn[0][c], ps[i * 3 + c]); mx[0][c] = fmaxf(mx[0][c], ps[i * 3 + c]);
            mn[1][c] = fminf(mn[1][c], gs[i * 3 + c]); mx[1][c] = fmaxf(mx[1][c], gs[i * 3 + c]);
        }
        double p[3], g[3], pp[2], gp[2];
        const double* v = t.verts_sampled + ((long long)o * ns + i) * 3;
        rt_apply(prt, v, p);
        rt_apply(grt, v, g);
        for (int r = 0; r < 2; ++r) {
            pp[r] = (p[0] * K[r * 3] + p[1] * K[r * 3 + 1] + p[2] * K[r * 3 + 2]) / (p[2] + 1e-7);
            gp[r] = (g[0] * K[r * 3] + g[1] * K[r * 3 + 1] + g[2] * K[r * 3 + 2]) / (g[2] + 1e-7);
        }
        s_rep += sqrt((pp[0] - gp[0]) * (pp[0] - gp[0]) + (pp[1] - gp[1]) * (pp[1] - gp[1]));
    }
    const double add = block_sum(s_add, red) / ns, rep = block_sum(s_rep, red) / ns, adds = block_sum(s_adds, red) / ns;
    float bmn[2][3], bmx[2][3];
    for (int k = 0; k < 2; ++k)
        for (int c = 0; c < 3; ++c) { bmn[k][c] = block_minmax(mn[k][c], false, redf); bmx[k][c] = block_minmax(mx[k][c], true, redf); }
    // full vertices: Chamfer distance and F-scores
    const float th[6] = {0.002f, 0.005f, 0.010f, 0.020f, 0.050f, 0.100f};
    double s_p = 0, s_g = 0, cp[6] = {0, 0, 0, 0, 0, 0}, cg[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < nf; i += 256) {
        const float dp = w.d_p2g[(long long)b * vmax + i], dg = w.d_g2p[(long long)b * vmax + i];
        s_p += (double)dp; s_g += (double)dg;
        for (int k = 0; k < 6; ++k) { cp[k] += dp < th[k] ? 1.0 : 0.0; cg[k] += dg < th[k] ? 1.0 : 0.0; }
    }
    const double mp = block_sum(s_p, red) / nf, mg = block_sum(s_g, red) / nf;
    double fs[6];
    for (int k = 0; k < 6; ++k) {
        const float prec = (float)(block_sum(cp[k], red) / nf), rec = (float)(block_sum(cg[k], red) / nf);
        fs[k] = (double)((2.f * prec * rec) / ((prec + rec) + 1e-6f));
    }
    if (tid == 0) {
        // corners of the two axis-aligned boxes in the reference's order (test.py:163-187)
        const int sel[3][8] = {{0, 1, 0, 0, 1, 0, 1, 1}, {0, 0, 1, 0, 1, 1, 0, 1}, {0, 0, 0, 1, 0, 1, 1, 1}};
        float s2 = 0.f;
        for (int k = 0; k < 8; ++k) {
            float d2 = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float a = sel[c][k] ? bmx[0][c] : bmn[0][c], g = sel[c][k] ? bmx[1][c] : bmn[1][c];
                d2 += (a - g) * (a - g);
            }
            s2 += sqrtf(d2);
        }
        const double diam = t.diameter[o];
        double* r = out + (long long)b * 16;
        r[0] = mce; r[1] = oce; r[2] = (double)(s2 / 8.f); r[3] = add; r[4] = adds;
        r[5] = add <= diam * 0.1 ? 1.0 : 0.0; r[6] = adds <= diam * 0.1 ? 1.0 : 0.0; r[7] = rep; r[8] = rep < 5 ? 1.0 : 0.0;
        r[9] = 0.5 * (mp + mg);
        for (int k = 0; k < 6; ++k) r[10 + k] = fs[k];
    }
}

}  // namespace

extern "C" int vpho_hand_metrics_f32(const float* pd, const float* gt, int n_img, int n_pts, float* mean_err, float* pa_mean_err,
                                     float* per_point, void* stream) {
    VPHO_REQUIRE(pd && gt && mean_err && pa_mean_err && n_img > 0 && n_pts >= 3, "vpho_hand_metrics_f32: bad argument");
    hipLaunchKernelGGL(hand_metrics_kernel, dim3(n_img), dim3(256), 0, (hipStream_t)stream, pd, gt, n_pts, mean_err, pa_mean_err, per_point);
    return vpho::check_launch("hand_metrics_kernel");
}

extern "C" long long vpho_obj_metrics_workspace_bytes(const vpho_obj_metric_tables* t, int n_img, int max_verts) {
    if (!t || n_img <= 0 || max_verts <= 0 || t->n_sampled <= 0) return -1;
    return obj_carve(n_img, t->n_sampled, max_verts, nullptr).bytes;
}

extern "C" int vpho_obj_metrics_f64(const vpho_obj_metric_tables* t, const double* pd_rt, const double* gt_rt, const double* cam_intr,
                                    const int* obj_id, int n_img, int max_verts, double* out, void* workspace, long long workspace_bytes,
                                    void* stream) {
    VPHO_REQUIRE(t && t->bbox3d && t->verts_sampled && t->verts && t->vert_offset && t->diameter && t->n_obj > 0 && t->n_sampled > 0,
                 "vpho_obj_metrics_f64: bad tables");
    VPHO_REQUIRE(pd_rt && gt_rt && cam_intr && obj_id && out && workspace && n_img > 0 && max_verts > 0, "vpho_obj_metrics_f64: bad argument");
    const ObjWs w = obj_carve(n_img, t->n_sampled, max_verts, (char*)workspace);
    VPHO_REQUIRE(workspace_bytes >= w.bytes, "vpho_obj_metrics_f64: workspace %lld < %lld bytes", workspace_bytes, w.bytes);
    hipStream_t s = (hipStream_t)stream;
    const int pts = t->n_sampled + max_verts;
    hipLaunchKernelGGL(obj_transform_kernel, dim3(n_img, (pts + 255) / 256), dim3(256), 0, s, *t, pd_rt, gt_rt, obj_id, max_verts, w);
    const int chunks = (std::max(t->n_sampled, max_verts) + 255) / 256;
    hipLaunchKernelGGL(obj_nn_kernel, dim3(n_img, 3, chunks), dim3(256), 0, s, *t, obj_id, max_verts, w);
    hipLaunchKernelGGL(obj_metrics_kernel, dim3(n_img), dim3(256), 0, s, *t, pd_rt, gt_rt, cam_intr, obj_id, max_verts, w, out);
    return vpho::check_launch("obj_metrics kernels");
}

extern "C" int vpho_obj_9d_to_rt_f64(const double* pose9, const float* root_joint, int n, double* rt, void* stream) {
    VPHO_REQUIRE(pose9 && root_joint && rt && n > 0, "vpho_obj_9d_to_rt_f64: bad argument");
    hipLaunchKernelGGL(obj_9d_to_rt_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose9, root_joint, n, rt);
    return vpho::check_launch("obj_9d_to_rt_kernel");
}
