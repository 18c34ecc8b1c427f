// Device-side hand metrics: MJE / PA-MJE / MVE / PA-MVE of TesterHand.criterion_MJE_PAMJE
// (lib/engine/test.py:657-680) with the Procrustes alignment of transform_fn.rigid_transform_3D_AtoB (:43-58), so that the
// evaluation loop ships 8 floats per image to the all-gather instead of copying (bs,S,778,3) candidates to the host.
// One block per image; reductions and the 3x3 SVD (Jacobi on H^T H) in fp64.  HBM-bound: 24 B read per point.
#include "common.h"
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
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (!(off > 1e-300)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (!(fabs(A[p][q]) > 1e-300)) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
                for (int k = 0; k < 3; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
                for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
            }
    }
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (A[order[j]][order[j]] > A[order[i]][order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    double Vs[3][3];
    for (int k = 0; k < 3; ++k) { w[k] = A[order[k]][order[k]]; for (int i = 0; i < 3; ++i) Vs[i][k] = V[i][order[k]]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = Vs[i][j];
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

}  // namespace

extern "C" int vpho_hand_metrics_f32(const float* pd, const float* gt, int n_img, int n_pts, float* mean_err, float* pa_mean_err,
                                     float* per_point, void* stream) {
    VPHO_REQUIRE(pd && gt && mean_err && pa_mean_err && n_img > 0 && n_pts >= 3, "vpho_hand_metrics_f32: bad argument");
    hipLaunchKernelGGL(hand_metrics_kernel, dim3(n_img), dim3(256), 0, (hipStream_t)stream, pd, gt, n_pts, mean_err, pa_mean_err, per_point);
    return vpho::check_launch("hand_metrics_kernel");
}
