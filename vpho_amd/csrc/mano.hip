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

// Block = HB hands (blockIdx.x) x up to 4 wave-chunks of 64 vertices (blockIdx.y; 13 chunks cover the 778 vertices).
// The kinematic part runs with one thread per (hand, joint): Rodrigues, then the 16-joint chain level by level (MANO's
// tree is root -> 5 fingers x 3 joints, so 4 levels).  The skinning part keeps HB x 3 pose-blend accumulators per
// thread-vertex, so one pass over the 1.26 MB pose-blend table (L2) serves HB hands instead of one.
template <int HB>
__global__ __launch_bounds__(256) void mano_fk_kernel(const FkArgs a) {
    __shared__ float R[HB][16][9];
    __shared__ float G[HB][16][12];                                  // global transforms (3x4 row-major)
    __shared__ __attribute__((aligned(16))) float A[HB][16][12];     // skinning transforms
    __shared__ __attribute__((aligned(16))) float pmT[135][HB];      // pose map R_j - I (j = 1..15), hand-minor
    __shared__ float jt[HB][21][3];                                  // 16 MANO joints + 5 tips (un-centred, MANO order)
    __shared__ float tipv[HB][10][3];
    const int tid = threadIdx.x;
    const long long h0 = (long long)blockIdx.x * HB;
    // hands past the end of the last group recompute the last hand and are never written
    auto hand_of = [&](int h) { const long long g = h0 + h; return g < a.n_hands ? g : a.n_hands - 1; };
    auto img_of = [&](int h) { return (int)(hand_of(h) / a.hands_per_image); };
    const int hj_h = tid >> 4, hj_j = tid & 15;                      // (hand, joint) role of the first HB*16 threads
    const bool hj = tid < HB * 16;

    if (hj) {
        float r[9];
        vpho::mano_rodrigues(a.pose + hand_of(hj_h) * a.ld_pose + 3 * hj_j, r);
        for (int k = 0; k < 9; ++k) R[hj_h][hj_j][k] = r[k];
    }
    __syncthreads();
    for (int i = tid; i < HB * 135; i += 256) {
        const int h = i / 135, k = i - h * 135;
        const int e = k % 9;
        pmT[k][h] = R[h][k / 9 + 1][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    // chain: root [R0 | J0]; child G[parent] * [R_i | J_i - J_parent]; joint i sits (i-1)%3+1 levels below the root
    for (int lvl = 0; lvl < 4; ++lvl) {
        if (hj && (hj_j == 0 ? 0 : (hj_j - 1) % 3 + 1) == lvl) {
            const float* Jr = a.J + img_of(hj_h) * 48;
            const int i = hj_j;
            float (*Gh)[12] = G[hj_h];
            const float* Ri = R[hj_h][i];
            if (i == 0) {
                for (int k = 0; k < 3; ++k) { for (int c = 0; c < 3; ++c) Gh[0][k * 4 + c] = Ri[k * 3 + c]; Gh[0][k * 4 + 3] = Jr[k]; }
            } else {
                const int p = c_parent[i];
                const float rel[3] = {Jr[i * 3 + 0] - Jr[p * 3 + 0], Jr[i * 3 + 1] - Jr[p * 3 + 1], Jr[i * 3 + 2] - Jr[p * 3 + 2]};
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) {
                        float s = 0.f;
                        for (int k = 0; k < 3; ++k) s += Gh[p][r * 4 + k] * Ri[k * 3 + c];
                        Gh[i][r * 4 + c] = s;
                    }
                    float s = 0.f;
                    for (int k = 0; k < 3; ++k) s += Gh[p][r * 4 + k] * rel[k];
                    Gh[i][r * 4 + 3] = s + Gh[p][r * 4 + 3];
                }
            }
        }
        __syncthreads();
    }
    if (hj) {
        const float* Jr = a.J + img_of(hj_h) * 48;
        const int i = hj_j;
        for (int r = 0; r < 3; ++r) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += G[hj_h][i][r * 4 + k] * Jr[i * 3 + k];
            for (int c = 0; c < 3; ++c) A[hj_h][i][r * 4 + c] = G[hj_h][i][r * 4 + c];
            A[hj_h][i][r * 4 + 3] = G[hj_h][i][r * 4 + 3] - s;
            jt[hj_h][i][r] = G[hj_h][i][r * 4 + 3];
        }
    }
    __syncthreads();

    // vp = v_shaped + posedirs . pose_map (k ascending), T = sum_j w_j A_j (j ascending), out = T [vp; 1].
    // The two contractions (135-long pose blend, 16-joint transform blend) are matrix products in the reference (torch.matmul inside
    // manopth: a BLAS kernel with fused multiply-adds in its own order), so they are written as FUSED multiply-adds here, two lanes of a
    // packed instruction at a time (v_pk_fma_f32: the fp32 rate of the matrix cores' 16x16x4 instruction, twice that of mul + add --
    // the build otherwise runs with -ffp-contract=off).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto skin_tail = [&](int h, int v, const float* wv, const float* blend, float* out3) {
        const float* vsh = a.v_shaped + (long long)img_of(h) * NV * 3;
        float vp[3];
        for (int c = 0; c < 3; ++c) vp[c] = vsh[v * 3 + c] + blend[c];
        f32x2 T[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) T[e] = f32x2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f32x2 w = {wv[j], wv[j]};
            const f32x2* Aj = reinterpret_cast<const f32x2*>(A[h][j]);
#pragma unroll
            for (int e = 0; e < 6; ++e) T[e] = __builtin_elementwise_fma(Aj[e], w, T[e]);
        }
        for (int r = 0; r < 3; ++r) out3[r] = fmaf(T[2 * r][0], vp[0], fmaf(T[2 * r][1], vp[1], fmaf(T[2 * r + 1][0], vp[2], T[2 * r + 1][1])));
    };
    // centre on joint 0, mm (*1000) and back to metres (/1000) exactly as manopth + head_mano.py:85-86
    auto fin = [](float v, float c) { return ((v - c) * 1000.f) / 1000.f; };

    if (blockIdx.y == 0) {
        for (int t = tid; t < HB * 10; t += 256) {
            const int h = t / 10, tip = t - h * 10;
            const int v = tip < 5 ? c_tips[tip] : c_tips_ho3d[tip - 5];
            float blend[3], wv[16], o[3];
            for (int c = 0; c < 3; ++c) {
                float s = 0.f;
                // the tip columns from their own compact table when the caller supplies it (neighbouring lanes then read neighbouring
                // words of one 120-byte row instead of 30 different cache lines of the full table); same values, same order
                const float* pd = a.t.tip_posedirs_t ? a.t.tip_posedirs_t + tip * 3 + c : a.t.posedirs_t + v * 3 + c;
                const long long ld = a.t.tip_posedirs_t ? 30 : (long long)NV * 3;
                for (int k = 0; k < 135; ++k) s = fmaf(pd[k * ld], pmT[k][h], s);
                blend[c] = s;
            }
            for (int j = 0; j < 16; ++j) wv[j] = a.t.weights[v * 16 + j];
            skin_tail(h, v, wv, blend, o);
            for (int c = 0; c < 3; ++c) tipv[h][tip][c] = o[c];
            if (tip < 5) for (int c = 0; c < 3; ++c) jt[h][16 + tip][c] = o[c];
        }
        __syncthreads();
        for (int t = tid; t < HB * 21; t += 256) {
            const int h = t / 21, q = t - h * 21;
            if (h0 + h >= a.n_hands) continue;
            const bool ho = a.ho3d && a.ho3d[img_of(h)];
            const float cx = jt[h][0][0], cy = jt[h][0][1], cz = jt[h][0][2];
            const float* src;
            if (!ho) src = jt[h][c_order[q]];
            else if (q >= 16) src = tipv[h][5 + q - 16];             // hand_fn.py:454-461: HO3D joint order with its own tip vertices
            else src = jt[h][c_order[c_to_manolayer[q]]];
            float* jo = a.joints + ((h0 + h) * 21 + q) * 3;
            jo[0] = fin(src[0], cx); jo[1] = fin(src[1], cy); jo[2] = fin(src[2], cz);
        }
    }
    if (a.verts) {
        const int v = (blockIdx.y * 4 + (tid >> 6)) * 64 + (tid & 63);
        if (v < NV) {
            float acc[HB][3];
            const float* pd = a.t.posedirs_t + v * 3;
            if constexpr (HB % 2 == 0) {
                // hands in pairs: one packed FMA advances coordinate c of two hands (their pose-map entries are neighbours in pmT)
                f32x2 ap[HB / 2][3];
#pragma unroll
                for (int h = 0; h < HB / 2; ++h) ap[h][0] = ap[h][1] = ap[h][2] = f32x2{0.f, 0.f};
#pragma unroll 3
                for (int k = 0; k < 135; ++k) {
                    const float p0 = pd[(long long)k * NV * 3], p1 = pd[(long long)k * NV * 3 + 1], p2 = pd[(long long)k * NV * 3 + 2];
                    const f32x2 q0 = {p0, p0}, q1 = {p1, p1}, q2 = {p2, p2};
                    const f32x2* m2 = reinterpret_cast<const f32x2*>(pmT[k]);
#pragma unroll
                    for (int h = 0; h < HB / 2; ++h) {
                        const f32x2 m = m2[h];
                        ap[h][0] = __builtin_elementwise_fma(q0, m, ap[h][0]);
                        ap[h][1] = __builtin_elementwise_fma(q1, m, ap[h][1]);
                        ap[h][2] = __builtin_elementwise_fma(q2, m, ap[h][2]);
                    }
                }
#pragma unroll
                for (int h = 0; h < HB; ++h)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[h][c] = ap[h / 2][c][h & 1];
            } else {
#pragma unroll
                for (int h = 0; h < HB; ++h) acc[h][0] = acc[h][1] = acc[h][2] = 0.f;
#pragma unroll 3
                for (int k = 0; k < 135; ++k) {
                    const float p0 = pd[(long long)k * NV * 3], p1 = pd[(long long)k * NV * 3 + 1], p2 = pd[(long long)k * NV * 3 + 2];
#pragma unroll
                    for (int h = 0; h < HB; ++h) {
                        const float m = pmT[k][h];
                        acc[h][0] = fmaf(p0, m, acc[h][0]); acc[h][1] = fmaf(p1, m, acc[h][1]); acc[h][2] = fmaf(p2, m, acc[h][2]);
                    }
                }
            }
            float wv[16];
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.t.weights + v * 16 + j);
                wv[j] = w4[0]; wv[j + 1] = w4[1]; wv[j + 2] = w4[2]; wv[j + 3] = w4[3];
            }
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                if (h0 + h >= a.n_hands) break;
                float o[3];
                skin_tail(h, v, wv, acc[h], o);
                float* vo = a.verts + ((h0 + h) * NV + v) * 3;
                vo[0] = fin(o[0], jt[h][0][0]); vo[1] = fin(o[1], jt[h][0][1]); vo[2] = fin(o[2], jt[h][0][2]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- vertices on the matrix cores
// The big launches with vertices (4 096 hands and more: the 6 400 final hypotheses): 32 hands per workgroup = the rows of a 32x32
// fp32 MFMA tile, 8 waves; both contractions are matrix products
//   blend[hand][v*3+c] = sum_k pose_map[hand][k] . posedirs[k][v*3+c]        K = 135 (+1 zero row), 3 accumulator tiles per 32 vertices
//   T_e[hand][v]       = sum_j A_e[hand][j] . w[j][v]   e = 0..11              K = 16, 12 accumulator tiles per 32 vertices
// so an operand is fetched once per 2 048 multiply-adds instead of once per multiply-add (the packed-FMA kernel above is bound by
// its LDS broadcasts of the transforms and by the latency of its table loads, not by arithmetic).  A lane ends up with the three
// blend coordinates and the twelve transform entries of ONE vertex for 16 hands: the 3x4 transform is applied in registers.
// The kinematic part and the finger tips are the code of mano_fk_kernel, so the joints are bit-identical to a joints-only launch.
#ifndef FK_ABLATE
#define FK_ABLATE 0                     // timing ablations (scripts/kernel_ablate.sh): 1 no vertex stores, 2 no table loads, 4 no blend MFMAs, 8 no skin MFMAs
#endif
constexpr int MH = 32;
struct FkMfmaLds {
    union {                      // 91 KB in all: one of these workgroups and one 64 KB convolution workgroup fit a CU together
        struct { float R[MH][16][9]; float G[MH][16][12]; } kin;                  // rotations and global transforms: dead once AT is made
        struct { float tippd[135][30]; float tipblend[MH][10][3]; float tipv[MH][10][3]; } tip;   // tip columns of the table, tip results
    } u;
    float AT[12][16][MH];        // skinning transforms, element-major / hand-minor: the A operand of the transform blend
    float pmT[136][MH];          // pose map R_j - I, hand-minor; row 135 = 0 (K padded to the MFMA's k step of 2)
    float jt[MH][21][3];
    int img[MH];
};

__global__ __launch_bounds__(512) void mano_fk_mfma_kernel(const FkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fk_smem[];
    FkMfmaLds& L = *reinterpret_cast<FkMfmaLds*>(fk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const long long h0 = (long long)blockIdx.x * MH;
    auto hand_of = [&](int h) { const long long g = h0 + h; return g < a.n_hands ? g : a.n_hands - 1; };
    const int hj_h = tid >> 4, hj_j = tid & 15;                      // (hand, joint) role of all 512 threads
    {
        float r[9];
        if (FK_ABLATE & 128) { for (int k = 0; k < 9; ++k) r[k] = (float)k; }
        else vpho::mano_rodrigues(a.pose + hand_of(hj_h) * a.ld_pose + 3 * hj_j, r);
        for (int k = 0; k < 9; ++k) L.u.kin.R[hj_h][hj_j][k] = r[k];
        if (hj_j == 0) L.img[hj_h] = (int)(hand_of(hj_h) / a.hands_per_image);
    }
    __syncthreads();
    for (int i = tid; i < MH * 136; i += 512) {
        const int k = i / MH, h = i - k * MH;
        float v = 0.f;
        if (k < 135) { const int e = k % 9; v = L.u.kin.R[h][k / 9 + 1][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f); }
        L.pmT[k][h] = v;
    }
    const float* Jr = a.J + L.img[hj_h] * 48;
    for (int lvl = 0; lvl < 4; ++lvl) {
        if ((hj_j == 0 ? 0 : (hj_j - 1) % 3 + 1) == lvl) {
            const int i = hj_j;
            float (*Gh)[12] = L.u.kin.G[hj_h];
            const float* Ri = L.u.kin.R[hj_h][i];
            if (i == 0) {
                for (int k = 0; k < 3; ++k) { for (int c = 0; c < 3; ++c) Gh[0][k * 4 + c] = Ri[k * 3 + c]; Gh[0][k * 4 + 3] = Jr[k]; }
            } else {
                const int p = c_parent[i];
                const float rel[3] = {Jr[i * 3 + 0] - Jr[p * 3 + 0], Jr[i * 3 + 1] - Jr[p * 3 + 1], Jr[i * 3 + 2] - Jr[p * 3 + 2]};
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) {
                        float s = 0.f;
                        for (int k = 0; k < 3; ++k) s += Gh[p][r * 4 + k] * Ri[k * 3 + c];
                        Gh[i][r * 4 + c] = s;
                    }
                    float s = 0.f;
                    for (int k = 0; k < 3; ++k) s += Gh[p][r * 4 + k] * rel[k];
                    Gh[i][r * 4 + 3] = s + Gh[p][r * 4 + 3];
                }
            }
        }
        __syncthreads();
    }
    {
        const int i = hj_j;
        for (int r = 0; r < 3; ++r) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += L.u.kin.G[hj_h][i][r * 4 + k] * Jr[i * 3 + k];
            for (int c = 0; c < 3; ++c) L.AT[r * 4 + c][i][hj_h] = L.u.kin.G[hj_h][i][r * 4 + c];
            L.AT[r * 4 + 3][i][hj_h] = L.u.kin.G[hj_h][i][r * 4 + 3] - s;
            L.jt[hj_h][i][r] = L.u.kin.G[hj_h][i][r * 4 + 3];
        }
    }
    __syncthreads();                                                 // R and G are dead: their space takes the tip table
    if (a.t.tip_posedirs_t) {
        for (int i = tid; i < 135 * 30; i += 512) (&L.u.tip.tippd[0][0])[i] = a.t.tip_posedirs_t[i];
        __syncthreads();
    }
    auto fin = [](float v, float c) { return ((v - c) * 1000.f) / 1000.f; };
    // finger tips: the arithmetic of mano_fk_kernel's tip path, operation for operation (one thread per blend coordinate first)
    for (int t = tid; t < ((FK_ABLATE & 32) ? 0 : MH * 30); t += 512) {
        const int h = t / 30, tc = t - h * 30, tip = tc / 3, c = tc - tip * 3;
        const int v = tip < 5 ? c_tips[tip] : c_tips_ho3d[tip - 5];
        float s = 0.f;
        if (a.t.tip_posedirs_t) {
#pragma unroll 15
            for (int k = 0; k < 135; ++k) s = fmaf(L.u.tip.tippd[k][tc], L.pmT[k][h], s);
        } else {
            const float* pd = a.t.posedirs_t + v * 3 + c;
            for (int k = 0; k < 135; ++k) s = fmaf(pd[(long long)k * NV * 3], L.pmT[k][h], s);
        }
        L.u.tip.tipblend[h][tip][c] = s;
    }
    __syncthreads();
    for (int t = tid; t < ((FK_ABLATE & 64) ? 0 : MH * 10); t += 512) {
        const int h = t / 10, tip = t - h * 10;
        const int v = tip < 5 ? c_tips[tip] : c_tips_ho3d[tip - 5];
        const float* vsh = a.v_shaped + (long long)L.img[h] * NV * 3;
        float vp[3], T[12], o[3];
        for (int c = 0; c < 3; ++c) vp[c] = vsh[v * 3 + c] + L.u.tip.tipblend[h][tip][c];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < 16; ++j) {
            const float w = a.t.weights[v * 16 + j];
            for (int e = 0; e < 12; ++e) T[e] = fmaf(L.AT[e][j][h], w, T[e]);
        }
        for (int r = 0; r < 3; ++r) o[r] = fmaf(T[4 * r], vp[0], fmaf(T[4 * r + 1], vp[1], fmaf(T[4 * r + 2], vp[2], T[4 * r + 3])));
        for (int c = 0; c < 3; ++c) L.u.tip.tipv[h][tip][c] = o[c];
        if (tip < 5) for (int c = 0; c < 3; ++c) L.jt[h][16 + tip][c] = o[c];
    }
    __syncthreads();
    for (int t = tid; t < MH * 21; t += 512) {
        const int h = t / 21, q = t - h * 21;
        if (h0 + h >= a.n_hands) continue;
        const bool ho = a.ho3d && a.ho3d[L.img[h]];
        const float cx = L.jt[h][0][0], cy = L.jt[h][0][1], cz = L.jt[h][0][2];
        const float* src;
        if (!ho) src = L.jt[h][c_order[q]];
        else if (q >= 16) src = L.u.tip.tipv[h][5 + q - 16];
        else src = L.jt[h][c_order[c_to_manolayer[q]]];
        float* jo = a.joints + ((h0 + h) * 21 + q) * 3;
        jo[0] = fin(src[0], cx); jo[1] = fin(src[1], cy); jo[2] = fin(src[2], cz);
    }

    // ---- vertices: wave w takes the 32-vertex tiles w, w + 8, ...
    // hands_per_image >= 16: a block of 32 hands spans at most three images (the middle one is lo + 1, or hi again)
    const int img_lo = L.img[0], img_hi = L.img[MH - 1], img_mid = img_lo + 1 < img_hi ? img_lo + 1 : img_hi;
    constexpr int GS = 4, NG = 68 / GS, NB = 3;     static_assert(GS == 4, "the tiled table holds groups of four k steps");                      // B fragments are requested NB - 1 groups of k steps ahead (L2 latency > one group's MFMAs)
    // gfx9 counts loads and stores in ONE in-order counter: a table load issued behind a tile's vertex stores cannot be waited for without
    // waiting for those stores, so the first fragments of the NEXT tile are requested before the current tile's stores are issued
    constexpr int NT = (NV + 31) / 32;
    float bq[NB][GS][3];
    auto load_group = [&](int tile, int g, float (*dst)[3]) {
        // [t][g][j][lh][li][4]: three aligned 16-byte loads bring the fragments of four k steps (f = 4 j + e <-> step f / 3, coordinate f % 3)
        const f32x4* p = reinterpret_cast<const f32x4*>(a.t.posedirs_mfma) + ((long long)(tile * NG + g) * 3) * 64 + lane;
        float f[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x4 x = (FK_ABLATE & 2) ? f32x4{1.f, 2.f, 3.f, 4.f} : p[j * 64];
            f[4 * j] = x[0]; f[4 * j + 1] = x[1]; f[4 * j + 2] = x[2]; f[4 * j + 3] = x[3];
        }
#pragma unroll
        for (int q = 0; q < GS; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[q][c] = f[3 * q + c];
    };
    if (wave < NT) {
#pragma unroll
        for (int g = 0; g < NB - 1; ++g) load_group(wave, g, bq[g]);
    }
    for (int t = wave; t < NT; t += 8) {
        const int v = t * 32 + li, vc = v < NV ? v : NV - 1;
        float wv[8], vlo[3], vmid[3], vhi[3];
#pragma unroll
        for (int s = 0; s < 8; ++s) wv[s] = a.t.weights[vc * 16 + 2 * s + lh];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            vlo[c] = a.v_shaped[(long long)img_lo * NV * 3 + vc * 3 + c];
            vmid[c] = a.v_shaped[(long long)img_mid * NV * 3 + vc * 3 + c];
            vhi[c] = a.v_shaped[(long long)img_hi * NV * 3 + vc * 3 + c];
        }
        f32x16 acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + NB - 1 < NG) load_group(t, g + NB - 1, bq[(g + NB - 1) % NB]);
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const float af = L.pmT[2 * (g * GS + q) + lh][li];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (FK_ABLATE & 4) { acc[c][q & 15] += af * bq[g % NB][q][c]; continue; }
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bq[g % NB][q][c], acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 8 < NT) {
#pragma unroll
            for (int g = 0; g < NB - 1; ++g) load_group(t + 8, g, bq[g]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // vp = v_shaped + blend, in place
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int h = (i >> 2) * 8 + lh * 4 + (i & 3);
            const int im = L.img[h];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c][i] = (im == img_lo ? vlo[c] : im == img_hi ? vhi[c] : vmid[c]) + acc[c][i];
        }
        float cen[16];                                                // per hand of this lane: the root coordinate of row r
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            f32x16 T[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) T[q][i] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) cen[i] = L.jt[(i >> 2) * 8 + lh * 4 + (i & 3)][0][r];
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (FK_ABLATE & 8) { T[q][s] += L.AT[r * 4 + q][2 * s + lh][li] * wv[s]; continue; }
                    T[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.AT[r * 4 + q][2 * s + lh][li], wv[s], T[q], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (v < NV) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int h = (i >> 2) * 8 + lh * 4 + (i & 3);
                    const float o = fmaf(T[0][i], acc[0][i], fmaf(T[1][i], acc[1][i], fmaf(T[2][i], acc[2][i], T[3][i])));
                    if ((FK_ABLATE & 1) && o != 1.2345e-30f) continue;
                    if (h0 + h < a.n_hands) a.verts[((h0 + h) * NV + v) * 3 + r] = fin(o, cen[i]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
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
    // hands per block: enough reuse of the pose-blend table for big batches, enough blocks for small ones
    const unsigned gy = verts ? 4 : 1;
    // algorithmic bytes: 48 pose floats read + 21 joints (+ 778 vertices) written per hand; v_shaped + J read once per image
    vpho::ProfScope prof(vpho::PROF_MANO_FK, (hipStream_t)stream, 0.0,
                         (double)n_hands * (48 * 4 + 21 * 12 + (verts ? 778 * 12 : 0)) + (double)((n_hands + hands_per_image - 1) / hands_per_image) * (778 + 16) * 12);
    static const bool no_mfma = getenv("VPHO_MANO_MFMA") && atoi(getenv("VPHO_MANO_MFMA")) == 0;      // A/B aid: the packed-FMA kernel for every launch
    // one 32-hand workgroup per CU takes ~95 us whatever the launch size: it pays from ~4 000 hands on (the packed-FMA kernel needs 61 us for
    // 1 920 hands, 166 us for 6 400)
    if (verts && t->posedirs_mfma && n_hands >= 4096 && hands_per_image >= MH / 2 && !no_mfma) {
        VPHO_DYN_LDS(mano_fk_mfma_kernel, sizeof(FkMfmaLds));
        hipLaunchKernelGGL(mano_fk_mfma_kernel, dim3((unsigned)((n_hands + MH - 1) / MH)), dim3(512), sizeof(FkMfmaLds), (hipStream_t)stream, a);
        return vpho::check_launch("mano_fk_mfma_kernel");
    }
    if (n_hands >= 1024) hipLaunchKernelGGL(mano_fk_kernel<16>, dim3((unsigned)((n_hands + 15) / 16), gy), dim3(256), 0, (hipStream_t)stream, a);
    else if (n_hands >= 128) hipLaunchKernelGGL(mano_fk_kernel<4>, dim3((unsigned)((n_hands + 3) / 4), gy), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mano_fk_kernel<1>, dim3((unsigned)n_hands, gy), dim3(256), 0, (hipStream_t)stream, a);
    return vpho::check_launch("mano_fk_kernel");
}
