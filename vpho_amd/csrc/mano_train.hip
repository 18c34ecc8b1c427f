// Training-mode HeadMano tail: rotation_6d_to_matrix -> ManoLayer forward -> the four MANO losses ->
// gradient w.r.t. the 96 rot6d outputs of fc_pose and the 10 outputs of fc_shape, one workgroup per hand.
// Reference: lib/model/head_mano.py:60-87 (forward, get_hand_verts), :89-133 (get_loss: vert / joint / mano_pose / mano_shape),
// lib/model/VPHO.py:147-148,197-204,214-219 (call site, loss weights); manopth.ManoLayer as restated in oracle/mano.py.
//
// The reference goes rot6d -> matrix -> axis-angle -> (manopth) Rodrigues -> matrix; the round trip through the axis-angle is
// the identity on rotation matrices, so the kernel feeds the Gram-Schmidt matrices to the kinematic chain directly and the
// mano_pose loss (matrix_to_rotation_6d(axis_angle_to_matrix(pd_pose)) vs the ground truth) reads their first two rows; autograd
// through the reference's chain agrees with this to 2e-6 of the gradient (tests/golden/make_golden_mano_train.py).
// Backward by hand, in the order of the forward reversed: losses -> root centring / finger tips -> linear-blend skinning -> the
// 16-joint chain (serial, one thread: 15 3x4 products) -> pose blend shapes -> joint regressor / shape blend shapes -> Gram-Schmidt.
// 64 hands per step: latency-bound glue, written for clarity; every reduction has a fixed order.
#include "common.h"
#include "../../include/vpho_hip.h"

namespace {

__constant__ int t_parent[16] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__constant__ int t_tips[5] = {745, 317, 444, 556, 673};
// HO3D joint convention (hand_fn.py:454-461, applied to the regressed joints of HO3D images at VPHO.py:154-157 before the losses):
// the 21 manopth joints re-ordered by MANOPTH_TO_MANOLAYER, then the five tips replaced by HO3D's own tip vertices
__constant__ int t_tips_ho3d[5] = {728, 353, 442, 576, 694};
__constant__ int t_to_manolayer[21] = {0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3, 4, 8, 12, 16, 20};
__constant__ int t_order[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
constexpr int NV = 778, NE = NV * 3;

struct ManoTrainArgs {
    vpho_mano_tables t;
    const float *rot6d, *shape, *gt_vert, *gt_joint, *gt_rot6d, *gt_shape;
    const unsigned char* is_right;
    const unsigned char* is_ho3d;             // per hand or NULL
    int bs;
    float cv, cj, cp, cs;                     // 2 * weight / element count of each mean
    float *d_rot6d, *d_shape, *verts, *joints;
    double* loss_parts;                       // [bs][4] sums of squared differences: vert, joint, pose, shape
};

// sum of N per-thread values over the 256 threads of the block, result in every thread (fixed order: lanes by xor butterfly, then waves 0..3)
template <int N>
__device__ inline void block_sum(float (&v)[N], float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = v[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        v[i] = s;
    }
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) red[wave * N + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ((red[i] + red[N + i]) + red[2 * N + i]) + red[3 * N + i];
    __syncthreads();
}

__device__ inline void mat3_mul(const float* a, const float* b, float* c) {            // c = a b (row-major 3x3)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

__global__ __launch_bounds__(256) void mano_train_kernel(const ManoTrainArgs a) {
    __shared__ float vs[NE];            // v_shaped, later the gradient of v_posed / v_shaped
    __shared__ float vp[NE];            // v_posed
    __shared__ float dv[NE];            // gradient of the un-centred vertices
    __shared__ float R[16][9], G[16][12], A[16][12], J[16][3], dA[16][12], dG[16][12], dR[16][9], dJ[16][3];
    __shared__ float pf[135], dpf[135], be[10], xt[5][3], a2s[16][3], gn1[16], gs[16], gn2[16];
    __shared__ float red[4 * 12];
    __shared__ double lsum[4][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool ho3d = a.is_ho3d && a.is_ho3d[b];
    const vpho_mano_tables& t = a.t;

    // ---- 0. rotation_6d_to_matrix (rows b1, b2, b3), intermediates kept for the backward
    if (tid < 16) {
        const float* p = a.rot6d + (long long)b * 96 + tid * 6;
        const float a1[3] = {p[0], p[1], p[2]}, a2[3] = {p[3], p[4], p[5]};
        const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
        const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
        const float s = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
        const float u[3] = {a2[0] - s * b1[0], a2[1] - s * b1[1], a2[2] - s * b1[2]};
        const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
        const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
        float* r = R[tid];
        r[0] = b1[0]; r[1] = b1[1]; r[2] = b1[2]; r[3] = b2[0]; r[4] = b2[1]; r[5] = b2[2];
        r[6] = b1[1] * b2[2] - b1[2] * b2[1]; r[7] = b1[2] * b2[0] - b1[0] * b2[2]; r[8] = b1[0] * b2[1] - b1[1] * b2[0];
        a2s[tid][0] = a2[0]; a2s[tid][1] = a2[1]; a2s[tid][2] = a2[2];
        gn1[tid] = n1; gs[tid] = s; gn2[tid] = n2;
    }
    if (tid >= 64 && tid < 74) be[tid - 64] = a.shape[(long long)b * 10 + tid - 64];
    __syncthreads();

    // ---- 1. shape blend, 2. joint regressor
    for (int i = tid; i < NE; i += 256) {
        float s = 0.f;
        for (int k = 0; k < 10; ++k) s += t.shapedirs[i * 10 + k] * be[k];
        vs[i] = s + t.v_template[i];
    }
    if (tid < 135) pf[tid] = R[1 + tid / 9][tid % 9] - ((tid % 9) % 4 == 0 ? 1.f : 0.f);
    __syncthreads();
    for (int o = wave; o < 48; o += 4) {
        const int j = o / 3, c = o % 3;
        float s = 0.f;
        for (int v = lane; v < NV; v += 64) s += t.J_regressor[j * NV + v] * vs[v * 3 + c];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) J[j][c] = s;
    }
    // ---- 3. pose blend
    for (int i = tid; i < NE; i += 256) {
        float s = 0.f;
        for (int k = 0; k < 135; ++k) s += t.posedirs_t[k * NE + i] * pf[k];
        vp[i] = vs[i] + s;
    }
    __syncthreads();
    // ---- 4. kinematic chain (root -> 5 fingers x 3 joints), A = G with the rest joint removed
    if (tid == 0) {
        for (int j = 0; j < 16; ++j) {
            const int p = t_parent[j];
            if (p < 0) {
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) G[j][r * 4 + c] = R[j][r * 3 + c]; G[j][r * 4 + 3] = J[j][r]; }
            } else {
                const float rel[3] = {J[j][0] - J[p][0], J[j][1] - J[p][1], J[j][2] - J[p][2]};
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c)
                        G[j][r * 4 + c] = G[p][r * 4] * R[j][c] + G[p][r * 4 + 1] * R[j][3 + c] + G[p][r * 4 + 2] * R[j][6 + c];
                    G[j][r * 4 + 3] = G[p][r * 4] * rel[0] + G[p][r * 4 + 1] * rel[1] + G[p][r * 4 + 2] * rel[2] + G[p][r * 4 + 3];
                }
            }
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) A[j][r * 4 + c] = G[j][r * 4 + c];
                A[j][r * 4 + 3] = G[j][r * 4 + 3] - (G[j][r * 4] * J[j][0] + G[j][r * 4 + 1] * J[j][1] + G[j][r * 4 + 2] * J[j][2]);
            }
        }
    }
    __syncthreads();
    const float cen[3] = {G[0][3], G[0][7], G[0][11]};
    // ---- 5. skinning, vertex loss and its gradient at the un-centred vertices
    double lv = 0.0;
    float dsum[3] = {0.f, 0.f, 0.f};
    for (int v = tid; v < NV; v += 256) {
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < 16; ++j) {
            const float w = t.weights[v * 16 + j];
            if (w != 0.f) for (int e = 0; e < 12; ++e) T[e] += w * A[j][e];
        }
        const float* x0 = vp + v * 3;
        float x[3];
        for (int r = 0; r < 3; ++r) x[r] = T[r * 4] * x0[0] + T[r * 4 + 1] * x0[1] + T[r * 4 + 2] * x0[2] + T[r * 4 + 3];
        for (int q = 0; q < 5; ++q) if (v == (ho3d ? t_tips_ho3d[q] : t_tips[q])) { xt[q][0] = x[0]; xt[q][1] = x[1]; xt[q][2] = x[2]; }
        for (int r = 0; r < 3; ++r) {
            const float pd = x[r] - cen[r];
            if (a.verts) a.verts[((long long)b * NV + v) * 3 + r] = pd;
            const float diff = pd - a.gt_vert[((long long)b * NV + v) * 3 + r];
            lv += (double)diff * (double)diff;
            const float g = a.cv * diff;
            dv[v * 3 + r] = g;
            dsum[r] += g;
        }
    }
    block_sum<3>(dsum, red);                         // (also orders the xt / dv writes before the joint pass)
    // ---- 6. joints: loss, gradient routed to the chain translations / the tip vertices / the centre (thread 0, 21 joints)
    if (tid == 0) {
        double lj = 0.0;
        float dc[3] = {-dsum[0], -dsum[1], -dsum[2]};                 // centre = joint 0, subtracted from every vertex and joint
        for (int j = 0; j < 16; ++j) for (int e = 0; e < 12; ++e) dG[j][e] = 0.f;
        for (int i = 0; i < 21; ++i) {
            // output joint i: manopth order, or HO3D's (joints 0-15 re-ordered, 16-20 = HO3D's own tip vertices, captured in xt)
            const int m = !ho3d ? t_order[i] : (i < 16 ? t_order[t_to_manolayer[i]] : i);
            for (int r = 0; r < 3; ++r) {
                const float raw = m < 16 ? G[m][r * 4 + 3] : xt[m - 16][r];
                const float pd = raw - cen[r];
                if (a.joints) a.joints[((long long)b * 21 + i) * 3 + r] = pd;
                const float diff = pd - a.gt_joint[((long long)b * 21 + i) * 3 + r];
                lj += (double)diff * (double)diff;
                const float g = a.cj * diff;
                if (m < 16) dG[m][r * 4 + 3] += g; else dv[(ho3d ? t_tips_ho3d[m - 16] : t_tips[m - 16]) * 3 + r] += g;
                dc[r] -= g;
            }
        }
        for (int r = 0; r < 3; ++r) dG[0][r * 4 + 3] += dc[r];
        lsum[0][1] = lj;
    }
    __syncthreads();
    // ---- 7. skinning backward: gradient of v_posed (into vs) and of the 16 A matrices
    for (int v = tid; v < NV; v += 256) {
        float Tr[9];
        for (int e = 0; e < 9; ++e) Tr[e] = 0.f;
        for (int j = 0; j < 16; ++j) {
            const float w = t.weights[v * 16 + j];
            if (w != 0.f) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Tr[r * 3 + c] += w * A[j][r * 4 + c];
        }
        const float* g = dv + v * 3;
        for (int c = 0; c < 3; ++c) vs[v * 3 + c] = Tr[c] * g[0] + Tr[3 + c] * g[1] + Tr[6 + c] * g[2];
    }
    for (int j = 0; j < 16; ++j) {
        float acc[12];
        for (int e = 0; e < 12; ++e) acc[e] = 0.f;
        for (int v = tid; v < NV; v += 256) {
            const float w = t.weights[v * 16 + j];
            if (w != 0.f) {
                const float* g = dv + v * 3;
                const float* x0 = vp + v * 3;
                for (int r = 0; r < 3; ++r) {
                    acc[r * 4] += w * g[r] * x0[0]; acc[r * 4 + 1] += w * g[r] * x0[1]; acc[r * 4 + 2] += w * g[r] * x0[2];
                    acc[r * 4 + 3] += w * g[r];
                }
            }
        }
        block_sum<12>(acc, red);
        if (tid < 12) dA[j][tid] = acc[tid];
    }
    __syncthreads();
    // ---- 8. chain backward (children before parents: every joint's parent has a smaller index)
    if (tid == 0) {
        for (int j = 0; j < 16; ++j) {
            // A.R = G.R;  A.t = G.t - G.R J   ->   dG.R += dA.R - dA.t (x) J,  dG.t += dA.t,  dJ = -G.R^T dA.t
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) dG[j][r * 4 + c] += dA[j][r * 4 + c] - dA[j][r * 4 + 3] * J[j][c];
                dG[j][r * 4 + 3] += dA[j][r * 4 + 3];
            }
            for (int c = 0; c < 3; ++c) dJ[j][c] = -(G[j][c] * dA[j][3] + G[j][4 + c] * dA[j][7] + G[j][8 + c] * dA[j][11]);
        }
        for (int j = 15; j >= 1; --j) {
            const int p = t_parent[j];
            const float rel[3] = {J[j][0] - J[p][0], J[j][1] - J[p][1], J[j][2] - J[p][2]};
            // G_j.R = G_p.R R_j;  G_j.t = G_p.R rel + G_p.t
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    dR[j][r * 3 + c] = G[p][r] * dG[j][c] + G[p][4 + r] * dG[j][4 + c] + G[p][8 + r] * dG[j][8 + c];
                    dG[p][r * 4 + c] += dG[j][r * 4] * R[j][c * 3] + dG[j][r * 4 + 1] * R[j][c * 3 + 1] + dG[j][r * 4 + 2] * R[j][c * 3 + 2]
                                        + dG[j][r * 4 + 3] * rel[c];
                }
            for (int c = 0; c < 3; ++c) {
                const float dl = G[p][c] * dG[j][3] + G[p][4 + c] * dG[j][7] + G[p][8 + c] * dG[j][11];
                dJ[j][c] += dl; dJ[p][c] -= dl;
            }
            for (int r = 0; r < 3; ++r) dG[p][r * 4 + 3] += dG[j][r * 4 + 3];
        }
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) dR[0][r * 3 + c] = dG[0][r * 4 + c]; dJ[0][r] += dG[0][r * 4 + 3]; }
    }
    __syncthreads();
    // ---- 9. pose-blend backward: dpf[k] = posedirs[:, k] . d v_posed
    for (int k = wave; k < 135; k += 4) {
        float s = 0.f;
        for (int i = lane; i < NE; i += 64) s += t.posedirs_t[k * NE + i] * vs[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) dpf[k] = s;
    }
    __syncthreads();
    if (tid < 135) dR[1 + tid / 9][tid % 9] += dpf[tid];
    // ---- 10. joint-regressor backward: d v_shaped = d v_posed + J_regressor^T dJ
    for (int i = tid; i < NE; i += 256) {
        const int v = i / 3, c = i - v * 3;
        float s = vs[i];
        for (int j = 0; j < 16; ++j) s += t.J_regressor[j * NV + v] * dJ[j][c];
        vs[i] = s;
    }
    __syncthreads();
    // ---- 11. shape-blend backward + shape loss (right hands only, head_mano.py:112-122)
    double lshape = 0.0;                               // lane 0 of each wave: its k = wave, wave + 4, ...
    for (int k = wave; k < 10; k += 4) {
        float s = 0.f;
        for (int i = lane; i < NE; i += 64) s += t.shapedirs[i * 10 + k] * vs[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) {
            if (a.is_right[b]) {
                const float diff = be[k] - a.gt_shape[(long long)b * 10 + k];
                s += a.cs * diff;
                lshape += (double)diff * (double)diff;
            }
            a.d_shape[(long long)b * 10 + k] = s;
        }
    }
    if (lane == 0) lsum[wave][3] = lshape;
    // ---- 12. pose loss on the first two rows + Gram-Schmidt backward
    if (tid < 16) {
        float d[9];
        for (int e = 0; e < 9; ++e) d[e] = dR[tid][e];
        double lp = 0.0;
        for (int e = 0; e < 6; ++e) {
            const float diff = R[tid][e] - a.gt_rot6d[(long long)b * 96 + tid * 6 + e];
            lp += (double)diff * (double)diff;
            d[e] += a.cp * diff;
        }
        const float* r = R[tid];
        const float b1[3] = {r[0], r[1], r[2]}, b2[3] = {r[3], r[4], r[5]};
        float db1[3] = {d[0], d[1], d[2]}, db2[3] = {d[3], d[4], d[5]};
        const float db3[3] = {d[6], d[7], d[8]};
        // b3 = b1 x b2
        db1[0] += b2[1] * db3[2] - b2[2] * db3[1]; db1[1] += b2[2] * db3[0] - b2[0] * db3[2]; db1[2] += b2[0] * db3[1] - b2[1] * db3[0];
        db2[0] += db3[1] * b1[2] - db3[2] * b1[1]; db2[1] += db3[2] * b1[0] - db3[0] * b1[2]; db2[2] += db3[0] * b1[1] - db3[1] * b1[0];
        // b2 = u / |u|,  u = a2 - s b1,  s = b1 . a2
        const float n2 = gn2[tid], s = gs[tid], n1 = gn1[tid];
        const float q = b2[0] * db2[0] + b2[1] * db2[1] + b2[2] * db2[2];
        float du[3], da2[3], da1[3];
        for (int c = 0; c < 3; ++c) du[c] = (db2[c] - q * b2[c]) / n2;
        const float ds = -(du[0] * b1[0] + du[1] * b1[1] + du[2] * b1[2]);
        for (int c = 0; c < 3; ++c) { da2[c] = du[c] + ds * b1[c]; db1[c] += -s * du[c] + ds * a2s[tid][c]; }
        // b1 = a1 / |a1|
        const float q1 = b1[0] * db1[0] + b1[1] * db1[1] + b1[2] * db1[2];
        for (int c = 0; c < 3; ++c) da1[c] = (db1[c] - q1 * b1[c]) / n1;
        float* o = a.d_rot6d + (long long)b * 96 + tid * 6;
        o[0] = da1[0]; o[1] = da1[1]; o[2] = da1[2]; o[3] = da2[0]; o[4] = da2[1]; o[5] = da2[2];
        // sum of the 16 joints' pose terms in joint order (thread 0 after the shuffle chain below)
        for (int off = 8; off > 0; off >>= 1) lp += __shfl_down(lp, off, 16);
        if (tid == 0) lsum[0][2] = lp;
    }
    // ---- 13. vertex loss: block sum in fp64 (lanes, then waves)
    for (int off = 32; off > 0; off >>= 1) lv += __shfl_xor(lv, off);
    if (lane == 0) lsum[wave][0] = lv;
    __syncthreads();
    if (tid == 0) {
        double* o = a.loss_parts + (long long)b * 4;
        o[0] = ((lsum[0][0] + lsum[1][0]) + lsum[2][0]) + lsum[3][0];
        o[1] = lsum[0][1];
        o[2] = lsum[0][2];
        o[3] = ((lsum[0][3] + lsum[1][3]) + lsum[2][3]) + lsum[3][3];
    }
}

}  // namespace

extern "C" int vpho_mano_train_f32(const vpho_mano_tables* t, const float* rot6d, const float* shape, const float* gt_vert, const float* gt_joint,
                                   const float* gt_rot6d, const float* gt_shape, const unsigned char* is_right, const unsigned char* is_ho3d, int bs,
                                   float w_vert, float w_joint, float w_pose, float w_shape,
                                   float* d_rot6d, float* d_shape, double* loss_parts, float* verts, float* joints, void* stream) {
    VPHO_REQUIRE(t && rot6d && shape && gt_vert && gt_joint && gt_rot6d && gt_shape && is_right && d_rot6d && d_shape && loss_parts && bs > 0,
                 "vpho_mano_train_f32: bad argument");
    ManoTrainArgs a;
    a.t = *t; a.rot6d = rot6d; a.shape = shape; a.gt_vert = gt_vert; a.gt_joint = gt_joint; a.gt_rot6d = gt_rot6d; a.gt_shape = gt_shape;
    a.is_right = is_right; a.is_ho3d = is_ho3d; a.bs = bs;
    a.cv = 2.f * w_vert / ((float)bs * NE); a.cj = 2.f * w_joint / ((float)bs * 63.f); a.cp = 2.f * w_pose / ((float)bs * 96.f);
    a.cs = 2.f * w_shape / ((float)bs * 10.f);
    a.d_rot6d = d_rot6d; a.d_shape = d_shape; a.verts = verts; a.joints = joints; a.loss_parts = loss_parts;
    hipLaunchKernelGGL(mano_train_kernel, dim3(bs), dim3(256), 0, (hipStream_t)stream, a);
    return vpho::check_launch("mano_train_kernel");
}
