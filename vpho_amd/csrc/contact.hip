// Hand-object contact detection: lib/utils/physics_fn.py:47-117 (nearest neighbour both ways, signed
// normal-distance / tangential-distance gates, double-sigmoid contact weight), :201-208 (anchor pooling) and :210-221
// (is_grasped rule).  The reference walks an sklearn ball tree on the CPU per frame; here every query point scans the
// other point cloud staged through LDS in 1024-point tiles (coalesced 12-byte rows, HBM/L2-bound: nq*nt distance
// evaluations, 12*(nq+nt) B algorithmic traffic per sample).  Ties: smaller target index.
#include "common.h"
#include "../../include/vpho_hip.h"

namespace {

constexpr int CT_TILE = 1024;

struct ContactArgs {
    const float *q, *qn, *t;
    int nq, nt;
    float nlo, nhi, vt;
    double mid1, mid2, w0;
    float* weight;      // (n, nq)
    int* nn_index;      // (n, nq) or NULL: index of the nearest target where in contact, else -1
};

__device__ inline double contact_weight(double x, double mid1, double mid2) {
    const double m1 = 1.0 + exp(-1600.0 * (x - mid1));
    const double m2 = 1.0 + exp(1600.0 * (x - mid2));
    if (!isfinite(m1) || !isfinite(m2)) return 0.0;
    return 1.0 / (m1 * m2 + 1e-10);
}

__global__ __launch_bounds__(256) void contact_kernel(const ContactArgs a) {
    __shared__ float tile[CT_TILE * 3];
    const int sample = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const bool live = qi < a.nq;
    const float* Q = a.q + (long long)sample * a.nq * 3;
    const float* T = a.t + (long long)sample * a.nt * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) { qx = Q[qi * 3]; qy = Q[qi * 3 + 1]; qz = Q[qi * 3 + 2]; }
    float best = INFINITY;
    int bi = 0;
    for (int t0 = 0; t0 < a.nt; t0 += CT_TILE) {
        const int cnt = min(CT_TILE, a.nt - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += 256) tile[i] = T[(long long)t0 * 3 + i];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float dx = qx - tile[j * 3], dy = qy - tile[j * 3 + 1], dz = qz - tile[j * 3 + 2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; bi = t0 + j; }
        }
    }
    if (!live) return;
    const float* N = a.qn + ((long long)sample * a.nq + qi) * 3;
    const float vx = qx - T[bi * 3], vy = qy - T[bi * 3 + 1], vz = qz - T[bi * 3 + 2];
    const float nd = vx * N[0] + vy * N[1] + vz * N[2];
    const float rx = vx - nd * N[0], ry = vy - nd * N[1], rz = vz - nd * N[2];
    const float vd = sqrtf(rx * rx + ry * ry + rz * rz);
    const bool in = nd > a.nlo && nd < a.nhi && vd < a.vt;
    const long long o = (long long)sample * a.nq + qi;
    a.weight[o] = in ? (float)(contact_weight((double)nd, a.mid1, a.mid2) / a.w0) : 0.f;
    if (a.nn_index) a.nn_index[o] = in ? bi : -1;
}

// force_contact[a] = sum_k hand_contact[face[a][k]] * w'[a][k] / sum(w'[a]),  w' = [1, w1, w2]   (physics_fn.py:201-208);
// is_grasped = at least two of {palm, thumb, index, middle, ring, pinky} have positive pooled contact (:210-221)
__constant__ int c_group_of_anchor[32] = {1, 1, 1, 1, 1, 0, 1, 2, 2, 2, 2, 2, 0, 3, 3, 3, 3, 3, 0, 0, 4, 4, 4, 4, 4, 0, 0, 5, 5, 5, 5, 5};
__global__ void force_contact_kernel(const float* __restrict__ hand_contact, int ld, vpho_anchor_tables t, int n, float thresh,
                                     float* __restrict__ fc, unsigned char* __restrict__ grasped) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    float grp[6] = {0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 32; ++a) {
        const float w1 = t.anchor_weight[a * 2], w2 = t.anchor_weight[a * 2 + 1], ws = 1.f + w1 + w2;
        const float* hc = hand_contact + (long long)s * ld;
        const float v = hc[t.face_idx[a * 3]] * (1.f / ws) + hc[t.face_idx[a * 3 + 1]] * (w1 / ws) + hc[t.face_idx[a * 3 + 2]] * (w2 / ws);
        fc[s * 32 + a] = v;
        grp[c_group_of_anchor[a]] += v;
    }
    int cnt = 0;
    for (int g = 0; g < 6; ++g) cnt += grp[g] > thresh ? 1 : 0;
    if (grasped) grasped[s] = cnt >= 2 ? 1 : 0;
}

}  // namespace

extern "C" int vpho_contact_detect_f32(const float* query, const float* query_normals, const float* target, int n, int n_query, int n_target,
                                       float normal_lo, float normal_hi, float vertical_thresh, float decay_lo, float decay_hi,
                                       float* weight, int* nn_index, void* stream) {
    VPHO_REQUIRE(query && query_normals && target && weight && n > 0 && n_query > 0 && n_target > 0, "vpho_contact_detect_f32: bad argument");
    VPHO_REQUIRE(normal_lo < decay_lo && decay_lo < decay_hi && decay_hi < normal_hi && vertical_thresh > 0, "vpho_contact_detect_f32: thresholds must satisfy normal_lo < decay_lo < decay_hi < normal_hi");
    ContactArgs a;
    a.q = query; a.qn = query_normals; a.t = target; a.nq = n_query; a.nt = n_target;
    a.nlo = normal_lo; a.nhi = normal_hi; a.vt = vertical_thresh;
    a.mid1 = ((double)decay_lo + (double)normal_lo) / 2; a.mid2 = ((double)decay_hi + (double)normal_hi) / 2;
    {
        const double m1 = 1.0 + std::exp(-1600.0 * (0.0 - a.mid1)), m2 = 1.0 + std::exp(1600.0 * (0.0 - a.mid2));
        a.w0 = 1.0 / (m1 * m2 + 1e-10);
    }
    a.weight = weight; a.nn_index = nn_index;
    hipLaunchKernelGGL(contact_kernel, dim3((n_query + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, a);
    return vpho::check_launch("contact_kernel");
}

extern "C" int vpho_force_contact_f32(const vpho_anchor_tables* t, const float* hand_contact, int ld, int n, float thresh,
                                      float* force_contact, unsigned char* is_grasped, void* stream) {
    VPHO_REQUIRE(t && t->face_idx && t->anchor_weight && hand_contact && force_contact && n > 0 && ld >= 778, "vpho_force_contact_f32: bad argument");
    hipLaunchKernelGGL(force_contact_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, hand_contact, ld, *t, n, thresh, force_contact, is_grasped);
    return vpho::check_launch("force_contact_kernel");
}
