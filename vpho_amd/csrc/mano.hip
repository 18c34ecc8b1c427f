// MANO linear-blend-skinning forward kinematics (manopth.ManoLayer.forward as configured at head_mano.py:48-55, scaled
// to metres like head_mano.py:78-87).  HBM/L2-bound: per hand it writes 778*3*4 + 21*3*4 B and reads the shared
// 1.26 MB pose-blend table from L2.
//
// Split by what depends on what:
//   mano_shape_kernel  (per IMAGE)  v_shaped = v_template + shapedirs . betas;  J = J_regressor . v_shaped
//   mano_fk_kernel     (per HAND)   Rodrigues, 16-joint chain, pose blend, skinning, tips, root-centring.
// All hypotheses of an image share its betas (VPHO.py:322-330, aggregation.py:124), so the shape blend and the joint
// regression are done once per image instead of once per hand; with `verts == NULL` only the 5 finger-tip vertices are
// skinned (the heat-map cascade needs joints only).
#include "common.h"
#include "rot.h"
#include "../../include/vpho_hip.h"

namespace {

__constant__ int c_parent[16] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__constant__ int c_tips[5] = {745, 317, 444, 556, 673};
__constant__ int c_tips_ho3d[5] = {728, 353, 442, 576, 694};
// manopth joint order: out[i] = [16 MANO joints, 5 tips][c_order[i]]
__constant__ int c_order[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
// hand_fn.py:8-9: MANOPTH_TO_MANOLAYER = argsort(c_order)
__constant__ int c_to_manolayer[21] = {0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3, 4, 8, 12, 16, 20};

constexpr int NV = 778;

__global__ __launch_bounds__(256) void mano_shape_kernel(const vpho_mano_tables t, const float* __restrict__ betas, int n_img,
                                                         float* __restrict__ v_shaped, float* __restrict__ J) {
    __shared__ float vs[NV * 3];
    __shared__ float be[10];
    const int b = blockIdx.x;
    if (threadIdx.x < 10) be[threadIdx.x] = betas[b * 10 + threadIdx.x];
    __syncthreads();
    for (int i = threadIdx.x; i < NV * 3; i += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < 10; ++k) s += t.shapedirs[i * 10 + k] * be[k];
        s += t.v_template[i];
        vs[i] = s;
        v_shaped[(long long)b * NV * 3 + i] = s;
    }
    __syncthreads();
    // 48 outputs, each a 778-long dot product: 4 waves x 12 outputs
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o = wave; o < 48; o += 4) {
        const int j = o / 3, c = o % 3;
        float s = 0.f;
        for (int v = lane; v < NV; v += 64) s += t.J_regressor[j * NV + v] * vs[v * 3 + c];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) J[b * 48 + o] = s;
    }
}

struct FkArgs {
    vpho_mano_tables t;
    const float* pose; int ld_pose;            // (n_hands, >=48) axis-angle rows
    const float* v_shaped; const float* J;     // per image
    long long n_hands; int hands_per_image;
    const unsigned char* ho3d;                 // per image flag or NULL
    float* verts;                              // (n_hands, 778, 3) or NULL
    float* joints;                             // (n_hands, 21, 3)
};

__global__ __launch_bounds__(256) void mano_fk_kernel(const FkArgs a) {
    __shared__ float R[16][9];
    __shared__ float G[16][12];      // global transforms (3x4 row-major)
    __shared__ float A[16][12];      // skinning transforms
    __shared__ float pm[135];
    __shared__ float jt[21][3];      // 16 MANO joints + 5 tips (un-centred, MANO order)
    __shared__ float tipv[10][3];
    const long long hand = blockIdx.x;
    const int img = (int)(hand / a.hands_per_image);
    const float* pose = a.pose + hand * a.ld_pose;
    const float* Jr = a.J + img * 48;
    const float* vsh = a.v_shaped + (long long)img * NV * 3;
    const int tid = threadIdx.x;

    if (tid < 16) {
        float r[9];
        vpho::mano_rodrigues(pose + 3 * tid, r);
        for (int k = 0; k < 9; ++k) R[tid][k] = r[k];
    }
    __syncthreads();
    for (int i = tid; i < 135; i += blockDim.x) {
        const int j = i / 9 + 1, k = i % 9;
        pm[i] = R[j][k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
    }
    if (tid == 0) {
        // root: [R0 | J0]; child: G[parent] * [R_i | J_i - J_parent]   (MANO joint order is parent-before-child)
        for (int k = 0; k < 3; ++k) { for (int c = 0; c < 3; ++c) G[0][k * 4 + c] = R[0][k * 3 + c]; G[0][k * 4 + 3] = Jr[k]; }
        for (int i = 1; i < 16; ++i) {
            const int p = c_parent[i];
            const float rel[3] = {Jr[i * 3 + 0] - Jr[p * 3 + 0], Jr[i * 3 + 1] - Jr[p * 3 + 1], Jr[i * 3 + 2] - Jr[p * 3 + 2]};
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) {
                    float s = 0.f;
                    for (int k = 0; k < 3; ++k) s += G[p][r * 4 + k] * R[i][k * 3 + c];
                    G[i][r * 4 + c] = s;
                }
                float s = 0.f;
                for (int k = 0; k < 3; ++k) s += G[p][r * 4 + k] * rel[k];
                G[i][r * 4 + 3] = s + G[p][r * 4 + 3];
            }
        }
    }
    __syncthreads();
    if (tid < 16) {
        const int i = tid;
        for (int r = 0; r < 3; ++r) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += G[i][r * 4 + k] * Jr[i * 3 + k];
            for (int c = 0; c < 3; ++c) A[i][r * 4 + c] = G[i][r * 4 + c];
            A[i][r * 4 + 3] = G[i][r * 4 + 3] - s;
            jt[i][r] = G[i][r * 4 + 3];
        }
    }
    __syncthreads();

    auto skin = [&](int v, float* out3) {
        float vp[3];
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
            const float* pd = a.t.posedirs_t + v * 3 + c;          // [k][v*3+c]
            for (int k = 0; k < 135; ++k) s += pd[(long long)k * NV * 3] * pm[k];
            vp[c] = vsh[v * 3 + c] + s;
        }
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < 16; ++j) {
            const float w = a.t.weights[v * 16 + j];
            for (int e = 0; e < 12; ++e) T[e] += A[j][e] * w;
        }
        for (int r = 0; r < 3; ++r) out3[r] = T[r * 4 + 0] * vp[0] + T[r * 4 + 1] * vp[1] + T[r * 4 + 2] * vp[2] + T[r * 4 + 3];
    };
    const bool ho = a.ho3d && a.ho3d[img];
    if (tid < 10) {
        float o[3];
        skin(tid < 5 ? c_tips[tid] : c_tips_ho3d[tid - 5], o);
        for (int c = 0; c < 3; ++c) tipv[tid][c] = o[c];
        if (tid < 5) for (int c = 0; c < 3; ++c) jt[16 + tid][c] = o[c];
    }
    __syncthreads();
    const float cx = jt[0][0], cy = jt[0][1], cz = jt[0][2];
    // centre on joint 0, mm (*1000) and back to metres (/1000) exactly as manopth + head_mano.py:85-86
    auto fin = [](float v, float c) { return ((v - c) * 1000.f) / 1000.f; };
    if (tid < 21) {
        float o[3];
        if (!ho) {
            const int s = c_order[tid];
            o[0] = fin(jt[s][0], cx); o[1] = fin(jt[s][1], cy); o[2] = fin(jt[s][2], cz);
        } else if (tid >= 16) {      // hand_fn.py:454-461: HO3D joint order with its own tip vertices
            o[0] = fin(tipv[5 + tid - 16][0], cx); o[1] = fin(tipv[5 + tid - 16][1], cy); o[2] = fin(tipv[5 + tid - 16][2], cz);
        } else {
            const int s = c_order[c_to_manolayer[tid]];
            o[0] = fin(jt[s][0], cx); o[1] = fin(jt[s][1], cy); o[2] = fin(jt[s][2], cz);
        }
        float* jo = a.joints + (hand * 21 + tid) * 3;
        jo[0] = o[0]; jo[1] = o[1]; jo[2] = o[2];
    }
    if (a.verts) {
        float* vo = a.verts + hand * NV * 3;
        for (int v = tid; v < NV; v += blockDim.x) {
            float o[3];
            skin(v, o);
            vo[v * 3 + 0] = fin(o[0], cx); vo[v * 3 + 1] = fin(o[1], cy); vo[v * 3 + 2] = fin(o[2], cz);
        }
    }
}

}  // namespace

extern "C" int vpho_mano_shape_f32(const vpho_mano_tables* t, const float* betas, int n_img, float* v_shaped, float* J, void* stream) {
    VPHO_REQUIRE(t && t->v_template && t->shapedirs && t->J_regressor && betas && v_shaped && J && n_img > 0, "vpho_mano_shape_f32: bad argument");
    hipLaunchKernelGGL(mano_shape_kernel, dim3(n_img), dim3(256), 0, (hipStream_t)stream, *t, betas, n_img, v_shaped, J);
    return vpho::check_launch("mano_shape_kernel");
}

extern "C" int vpho_mano_fk_f32(const vpho_mano_tables* t, const float* pose, int ld_pose, long long n_hands, int hands_per_image,
                                const float* v_shaped, const float* J, const unsigned char* ho3d_per_image,
                                float* verts, float* joints, void* stream) {
    VPHO_REQUIRE(t && t->posedirs_t && t->weights && pose && v_shaped && J && joints && n_hands > 0 && hands_per_image > 0 && ld_pose >= 48,
                 "vpho_mano_fk_f32: bad argument");
    VPHO_REQUIRE(n_hands < (1ll << 31), "vpho_mano_fk_f32: too many hands");
    FkArgs a;
    a.t = *t; a.pose = pose; a.ld_pose = ld_pose; a.v_shaped = v_shaped; a.J = J; a.n_hands = n_hands;
    a.hands_per_image = hands_per_image; a.ho3d = ho3d_per_image; a.verts = verts; a.joints = joints;
    hipLaunchKernelGGL(mano_fk_kernel, dim3((unsigned)n_hands), dim3(verts ? 256 : 64), 0, (hipStream_t)stream, a);
    return vpho::check_launch("mano_fk_kernel");
}
