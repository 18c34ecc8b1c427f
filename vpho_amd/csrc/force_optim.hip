// Pseudo-force label optimisation (BASELINE.json configs[4]): the 3000-iteration AdamW loop of
// ForceOptimizer.optimize_batch (lib/engine/force_optimization.py:110-207) as ONE persistent kernel per batch:
// forward (friction-cone mix -> anchor frames -> resultant force / moment / contact-distribution terms), analytic backward
// and the AdamW update all stay in registers; the constant anchor points and frames (the reference recomputes
// VERT2ANCHOR(vert) every iteration, :139) are computed once by anchor_frames_kernel.
// A sample's 32 anchors sit in one DPP row of 16 lanes, two per lane in packed registers; the only cross-sample coupling is
// the detached batch-mean force loss (`sum_weight`, :146), one LDS word per sample and one barrier per iteration.
// Vector-ALU work: no MFMA, no scratch and no HBM traffic inside the loop.
#include <type_traits>
#include "common.h"
#include "../../include/vpho_hip.h"

namespace {

constexpr int FO_THREADS = 1024, FO_IPT = 2, FO_MAXB = FO_THREADS * FO_IPT / 32;   // up to 64 samples per batch (the reference default)

// ForceAnchor.__call__ (physics_fn.py:224-257): anchor points and frames (frame[j][i] = component j of axis i)
struct AnchorFrameArgs { const float* verts; vpho_anchor_tables t; float* pts; float* frames; };
__global__ __launch_bounds__(256) void anchor_frames_kernel(const AnchorFrameArgs a) {
    __shared__ float jt[21][3];
    const long long hand = blockIdx.x;
    const float* V = a.verts + hand * 778 * 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o = wave; o < 63; o += 4) {
        const int j = o / 3, c = o % 3;
        float s = 0.f;
        for (int v = lane; v < 778; v += 64) s += V[v * 3 + c] * a.t.vert2joint[j * 778 + v];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) jt[j][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int an = threadIdx.x;
        float p[3][3];
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) p[k][c] = V[a.t.face_idx[an * 3 + k] * 3 + c];
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
        float* po = a.pts + (hand * 32 + an) * 3;
        float* fo = a.frames + (hand * 32 + an) * 9;
        for (int c = 0; c < 3; ++c) {
            po[c] = (w1 * b1[c] + w2 * b2[c]) + p[0][c];
            fo[c * 3 + 0] = dx[c]; fo[c * 3 + 1] = dy[c]; fo[c * 3 + 2] = dz[c];
        }
    }
}

struct FoArgs {
    const float *pts, *frames, *gravity, *com, *fc;
    const unsigned char* grasped;
    int B, iters, phase1;
    float lr, wd, beta1, beta2, eps, friction;
    float fB, invB, invB32;  // (float)B, 1/B, 1/(32 B): wave-uniform factors as kernel arguments (scalar registers), not recomputed per lane
    float cone[8][3];       // friction-cone anchors (physics.py:183-188,281-282), wave-uniform: kernel arguments stay in scalar registers
    float *fl_out, *fg_out, *scale_out, *weight_out, *losses_out;
};

typedef float f2 __attribute__((ext_vector_type(2)));       // the thread's two items side by side: v_pk_{mul,add,fma}_f32

// v + v[partner lane]; the partner is named by a DPP control inside the lane's row of 16 (folded into the v_add: one instruction)
template <int CTRL> __device__ inline float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (= one sample), the same bits in every lane: lane pairs (quad_perm [1,0,3,2]), quads
// ([2,3,0,1]), then the two quads of a half row (row_half_mirror: the quad sums are lane-uniform by now, so the mirror IS the
// exchange) and the two half rows (row_mirror).  Every step adds the same two numbers on both sides.
__device__ inline float row_sum(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    return v;
}

// EXACT = false (default): divisions, square roots, exp and log through the hardware's 1-ulp v_rcp / v_sqrt / v_exp / v_log (an IEEE
// division is ~10 instructions, the loop had 57 of them per item and iteration); EXACT = true (VPHO_FORCE_EXACT=1): correctly
// rounded `/`, sqrtf, and libm expf / logf -- the A/B that says what the fast forms do to a 3000-step trajectory.
template <bool EXACT> struct Fm {
    static __device__ inline float rcp(float x) { return EXACT ? 1.f / x : __builtin_amdgcn_rcpf(x); }
    static __device__ inline float sqrt(float x) { return EXACT ? sqrtf(x) : __builtin_amdgcn_sqrtf(x); }
    static __device__ inline float exp(float x) { return EXACT ? expf(x) : __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
    static __device__ inline float log(float x) { return EXACT ? logf(x) : __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
    static __device__ inline float div(float x, float y) { return EXACT ? x / y : x * __builtin_amdgcn_rcpf(y); }
    static __device__ inline f2 rcp(f2 x) { return f2{rcp(x.x), rcp(x.y)}; }
    static __device__ inline f2 sqrt(f2 x) { return f2{sqrt(x.x), sqrt(x.y)}; }
    static __device__ inline f2 exp(f2 x) { return f2{exp(x.x), exp(x.y)}; }
    static __device__ inline f2 log(f2 x) { return f2{log(x.x), log(x.y)}; }
    static __device__ inline f2 div(f2 x, f2 y) { return f2{div(x.x, y.x), div(x.y, y.y)}; }
};
__device__ inline f2 splat(float v) { return f2{v, v}; }
__device__ inline f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline f2 abs2(f2 a) { return __builtin_elementwise_abs(a); }
__device__ inline f2 sign2(f2 a) { return f2{a.x > 0.f ? 1.f : (a.x < 0.f ? -1.f : 0.f), a.y > 0.f ? 1.f : (a.y < 0.f ? -1.f : 0.f)}; }

// One workgroup per batch of B <= 64 samples, 16 lanes (one DPP row) per sample, a lane holds anchors l and l + 16 of its sample
// as the two halves of packed registers.  Per iteration: every per-sample sum (resultant force 3, moment 3, |scale|^2) is the lane's
// own two items added, then four DPP adds; the batch mean is one LDS word per sample, one barrier, one 16-byte read and the same four
// adds.  Parameters (1 + 8 per item), anchor frames and arms live in registers, the 18 AdamW moments per item in LDS (144 KB of the
// CU's 160), the per-step bias corrections in an LDS table refilled every 1024 iterations (no fp64 in the loop); the soft-max
// gradient is consumed by the AdamW update as it is produced.  No scratch, no HBM traffic inside the loop.
constexpr int FO_TAB = 1024;
template <bool EXACT>
__global__ __launch_bounds__(FO_THREADS) void force_optim_kernel(const FoArgs a) {
    using M_ = Fm<EXACT>;
    __shared__ __align__(16) float s_red[2][FO_MAXB];
    __shared__ __align__(16) float s_g[FO_MAXB][4];
    __shared__ __align__(16) float s_fin[4][FO_MAXB];      // row 0 is read 16 bytes at a time (batch_mean)
    __shared__ __align__(8) float s_tab[FO_TAB][2];         // per iteration: AdamW step size lr / (1 - beta1^n), and (1 - beta2^n)^(+-1/2)
    extern __shared__ __align__(16) float s_mom[];          // [9 parameters][thread] x (m item 0, m item 1, v item 0, v item 1)
    const int batch = blockIdx.x, B = a.B, tid = threadIdx.x;
    const int b = tid >> 4, l16 = tid & 15;
    const bool valid = b < B;
    const long long sb = (long long)batch * B + (valid ? b : 0), ia0 = sb * 32 + l16, ia1 = ia0 + 16;
    float4* mom = reinterpret_cast<float4*>(s_mom) + tid;
    const auto& cone = a.cone;

    // per-item constants (x = anchor l16, y = anchor l16 + 16); fcm = normalised contact force where the anchor is in contact
    // (force > 0.1, hence positive), 0 elsewhere: the contact mask is its sign
    f2 F[9], arm[3], fcm, s, w[8];
#pragma unroll
    for (int e = 0; e < 9; ++e) F[e] = valid ? f2{a.frames[ia0 * 9 + e], a.frames[ia1 * 9 + e]} : splat(0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (l16 == 0) s_g[b][c] = valid ? a.gravity[sb * 3 + c] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float cm_ = a.com[sb * 3 + c];
        arm[c] = valid ? f2{a.pts[ia0 * 3 + c] - cm_, a.pts[ia1 * 3 + c] - cm_} : splat(0.f);
    }
    {
        const f2 fc = valid ? f2{a.fc[ia0], a.fc[ia1]} : splat(0.f);
        const float fnorm = sqrtf(row_sum(fc.x * fc.x + fc.y * fc.y));
        const f2 fcn = fc / splat(fnorm + 1e-8f);
        fcm = f2{fc.x > 0.1f ? fcn.x : 0.f, fc.y > 0.1f ? fcn.y : 0.f};
    }
    s = splat(0.05f);
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = splat(0.f);
#pragma unroll
    for (int e = 0; e < 9; ++e) mom[e * FO_THREADS] = float4{0.f, 0.f, 0.f, 0.f};
    const float invB = a.invB;
    auto masked = [&](f2 v) { return f2{fcm.x > 0.f ? v.x : 0.f, fcm.y > 0.f ? v.y : 0.f}; };

    // forward of one iteration: soft-max mix of the cone anchors, its norm, and the per-sample sums
    struct Fwd { f2 p[8], vdir[3], vn, rden, se, ase; float R[3], M[3]; };
    auto forces = [&](const Fwd& f, f2 (&fl)[3], f2 (&fg)[3]) {
        // local force direction d = v / (|v| + 1e-8), local force d |scale|, global force through the anchor frame
#pragma unroll
        for (int c = 0; c < 3; ++c) fl[c] = (EXACT ? f.vdir[c] / (f.vn + splat(1e-8f)) : f.vdir[c] * f.rden) * f.ase;
#pragma unroll
        for (int j = 0; j < 3; ++j) fg[j] = fma2(fl[2], F[j * 3 + 2], fma2(fl[1], F[j * 3 + 1], fl[0] * F[j * 3 + 0]));
    };
    auto forward = [&](Fwd& f) {
        f.se = masked(s); f.ase = abs2(f.se);
        {
            f2 mx = w[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) mx = __builtin_elementwise_max(mx, w[e]);
            f2 sum = splat(0.f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { f.p[e] = M_::exp(w[e] - mx); sum += f.p[e]; }
            const f2 rs = M_::rcp(sum);
#pragma unroll
            for (int e = 0; e < 8; ++e) f.p[e] = EXACT ? f.p[e] / sum : f.p[e] * rs;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f2 t = f.p[0] * splat(cone[0][c]);
#pragma unroll
            for (int e = 1; e < 8; ++e) t = fma2(f.p[e], splat(cone[e][c]), t);
            f.vdir[c] = t;
        }
        f.vn = M_::sqrt(fma2(f.vdir[2], f.vdir[2], fma2(f.vdir[1], f.vdir[1], f.vdir[0] * f.vdir[0])));
        f.rden = M_::rcp(f.vn + splat(1e-8f));
        f2 fl[3], fg[3];
        forces(f, fl, fg);
#pragma unroll
        for (int c = 0; c < 3; ++c) f.R[c] = row_sum(fg[c].x + fg[c].y);
        const f2 m0 = arm[1] * fg[2] - arm[2] * fg[1], m1 = arm[2] * fg[0] - arm[0] * fg[2], m2 = arm[0] * fg[1] - arm[1] * fg[0];
        f.M[0] = row_sum(m0.x + m0.y); f.M[1] = row_sum(m1.x + m1.y); f.M[2] = row_sum(m2.x + m2.y);
    };
    // batch mean of the resultant-force norm (force_loss; detached as sum_weight, :146): one word per sample (rows past B hold zeros),
    // a barrier, then every lane sums four words and the row: identical bits in every thread of the workgroup
    auto batch_mean = [&](float* red, float rn) {
        if (l16 == 0) red[b] = valid ? rn : 0.f;
        __syncthreads();
        const float4 q = reinterpret_cast<const float4*>(red)[l16];
        return row_sum((q.x + q.y) + (q.z + q.w)) * invB;
    };
    // dist = log(|fcn / sn| + 1e-8) * mask, sn = se / (|se|_sample + 1e-8); GRAD: also d dist / d se =
    // mask * sign(r)/(|r|+1e-8) * (-fcn/(sn+1e-8)^2) / (snorm+1e-8)
    auto contact_dist = [&](f2 se, float snorm, f2& ddist, bool grad) {
        const float rsn = M_::rcp(snorm + 1e-8f);
        const f2 sn = EXACT ? se / splat(snorm + 1e-8f) : se * splat(rsn);
        const f2 rsq = M_::rcp(sn + splat(1e-8f));
        const f2 r = EXACT ? fcm / (sn + splat(1e-8f)) : fcm * rsq;
        const f2 ar = abs2(r) + splat(1e-8f);
        if (grad) ddist = EXACT ? (sign2(r) / ar) * (-fcm / ((sn + splat(1e-8f)) * (sn + splat(1e-8f)))) / splat(snorm + 1e-8f)
                                : (sign2(r) * M_::rcp(ar)) * (-fcm * (rsq * rsq)) * splat(rsn);
        return masked(M_::log(ar));
    };
    // The report of the last iteration (force_optimization.py:156-207 prints the four losses every iteration; the labels are the last
    // iteration's forces, :199-202): a forward of its own on the parameters as they stand BEFORE that iteration's update -- the same
    // operations as the iteration's forward, so the same values -- kept out of the two loops' bodies so that nothing it needs stays
    // alive in them.
    auto report = [&]() {
        Fwd f;
        forward(f);
        const float4 g = *reinterpret_cast<const float4*>(s_g[b]);
        const float rx = f.R[0] + g.x, ry = f.R[1] + g.y, rz = f.R[2] + g.z;
        const float rn = M_::sqrt(rx * rx + ry * ry + rz * rz);
        const float snorm = M_::sqrt(row_sum(f.se.x * f.se.x + f.se.y * f.se.y));
        const float mn = M_::sqrt(f.M[0] * f.M[0] + f.M[1] * f.M[1] + f.M[2] * f.M[2]);
        f2 unused;
        const f2 dist = contact_dist(f.se, snorm, unused, false);
        const float d2 = row_sum(dist.x * dist.x + dist.y * dist.y);
        if (l16 == 0) {
            const float cosb = -(f.R[0] * g.x + f.R[1] * g.y + f.R[2] * g.z);
            s_fin[1][b] = valid ? (cosb - 1.f) * (cosb - 1.f) : 0.f;
            s_fin[2][b] = valid ? mn : 0.f;
            s_fin[3][b] = valid ? d2 : 0.f;
        }
        const float sw = batch_mean(s_fin[0], rn);
        const float cm = M_::div(30.f, 100.f * sw * sw + 1e-8f), cd = M_::div(0.1f, 1000.f * sw * sw + 1e-8f);
        if (tid < 4) {
            float t = 0.f;
            for (int q = 0; q < B; ++q) t += s_fin[tid][q];
            t /= (float)B;
            if (tid == 2) t *= cm;
            if (tid == 3) t = t / 32.f * cd;
            a.losses_out[batch * 4 + tid] = t;
        }
        if (valid) {
            const long long sb = (long long)batch * B + b, ia0 = sb * 32 + l16, ia1 = ia0 + 16;
            const float keep = a.grasped[sb] ? 1.f : 0.f;                                   // :199-202
            f2 fl[3], fg[3];
            forces(f, fl, fg);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.fl_out[ia0 * 3 + c] = fl[c].x * keep; a.fl_out[ia1 * 3 + c] = fl[c].y * keep;
                a.fg_out[ia0 * 3 + c] = fg[c].x * keep; a.fg_out[ia1 * 3 + c] = fg[c].y * keep;
            }
        }
    };
    // One iteration, compiled twice: PH1 = phase 1 (gravity alignment only: no batch mean, no barrier) and phase 2.
    auto iteration = [&](const int it, auto phase) {
        constexpr bool ph1 = decltype(phase)::value;
        if (!ph1 && it == a.phase1) {            // optimizer2 starts with fresh moments and step count (two AdamW objects, :36-37)
#pragma unroll
            for (int e = 0; e < 9; ++e) mom[e * FO_THREADS] = float4{0.f, 0.f, 0.f, 0.f};
        }
        Fwd f;
        forward(f);
        const float4 g = *reinterpret_cast<const float4*>(s_g[b]);
        // ---------------- backward (ordered so that the per-sample sums die early: 128 registers per thread, no scratch) ----------
        // d loss / d (global force of the item): the resultant's part is per sample, the moment's part u x arm per item
        f2 h[3], gse = splat(0.f);                              // gse = d loss / d (s*mask)
        if (ph1) {
            // gravity_loss = mean_b (cos_b - 1)^2, cos_b = R_b . (-g_b)
            const float cosb = -(f.R[0] * g.x + f.R[1] * g.y + f.R[2] * g.z);
            const float c2 = 2.f * (cosb - 1.f) * invB;
            h[0] = splat(-c2 * g.x); h[1] = splat(-c2 * g.y); h[2] = splat(-c2 * g.z);
        } else {
            const float rx = f.R[0] + g.x, ry = f.R[1] + g.y, rz = f.R[2] + g.z;
            const float rn = M_::sqrt(rx * rx + ry * ry + rz * rz);
            const float sw = batch_mean(s_red[it & 1], rn);
            const float cm = M_::div(30.f, 100.f * sw * sw + 1e-8f), cd = M_::div(0.1f, 1000.f * sw * sw + 1e-8f);
            {
                // dist_loss = cd * mean(dist^2)
                const float snorm = M_::sqrt(row_sum(f.se.x * f.se.x + f.se.y * f.se.y));
                f2 ddist;
                const f2 dist = contact_dist(f.se, snorm, ddist, true);
                gse = masked(splat(cd * 2.f) * dist * splat(a.invB32) * ddist);
            }
            const float mn = M_::sqrt(f.M[0] * f.M[0] + f.M[1] * f.M[1] + f.M[2] * f.M[2]);
            const float inv = rn > 0.f ? M_::rcp(rn * a.fB) : 0.f;                     // torch.norm backward: 0 at 0
            const float mi = mn > 0.f ? M_::div(cm, mn * a.fB) : 0.f;
            const float u0 = f.M[0] * mi, u1 = f.M[1] * mi, u2 = f.M[2] * mi;
            h[0] = splat(rx * inv) + (splat(u1) * arm[2] - splat(u2) * arm[1]);
            h[1] = splat(ry * inv) + (splat(u2) * arm[0] - splat(u0) * arm[2]);
            h[2] = splat(rz * inv) + (splat(u0) * arm[1] - splat(u1) * arm[0]);
        }
        f2 gv[3];
        {
            f2 gfl[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) gfl[i] = fma2(h[2], F[2 * 3 + i], fma2(h[1], F[1 * 3 + i], h[0] * F[0 * 3 + i]));
            // gfl . d with d = v rden, and (gfl |scale|) . v, share the product gfl . v
            const f2 den = f.vn + splat(1e-8f);
            const f2 t = fma2(gfl[2], f.vdir[2], fma2(gfl[1], f.vdir[1], gfl[0] * f.vdir[0]));
            const f2 gase = EXACT ? fma2(gfl[2], f.vdir[2] / den, fma2(gfl[1], f.vdir[1] / den, gfl[0] * (f.vdir[0] / den))) : t * f.rden;
            gse = fma2(gase, sign2(f.se), gse);
            // d/dv of v / (|v| + 1e-8): gd / den - (gd . v) v / (|v| den^2), gd = gfl |scale|; the second term 0 at |v| = 0 (torch.norm backward)
            f2 q = EXACT ? (t * f.ase) / (f.vn * den * den) : (t * f.ase) * M_::rcp(f.vn * den * den);
            q = f2{f.vn.x > 0.f ? q.x : 0.f, f.vn.y > 0.f ? q.y : 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) gv[c] = (EXACT ? gfl[c] * f.ase / den : gfl[c] * f.ase * f.rden) - q * f.vdir[c];
        }
        const f2 gs = masked(gse);
        f2 ps = splat(0.f);                                     // soft-max backward: gw_e = p_e (gp_e - sum_e p_e gp_e), gp_e = gv . cone_e
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f2 gp = fma2(gv[2], splat(cone[e][2]), fma2(gv[1], splat(cone[e][1]), gv[0] * splat(cone[e][0])));
            ps = fma2(f.p[e], gp, ps);
        }
        // ---------------- AdamW (torch.optim.AdamW defaults: weight_decay 0.01) ----------------
        const float2 tb = *reinterpret_cast<const float2*>(s_tab[it & (FO_TAB - 1)]);
        const float step = tb.x, bc = tb.y, decay = 1.f - a.lr * a.wd;
        const f2 ob1 = splat(1.f - a.beta1), ob2 = splat(1.f - a.beta2);
        auto adamw = [&](f2& param, f2 grad, int e) {
            const float4 mv = mom[e * FO_THREADS];
            const f2 m = fma2(ob1, grad, splat(a.beta1) * f2{mv.x, mv.y});
            const f2 v = fma2(ob2 * grad, grad, splat(a.beta2) * f2{mv.z, mv.w});
            mom[e * FO_THREADS] = float4{m.x, m.y, v.x, v.y};
            const f2 dn = EXACT ? M_::sqrt(v) / splat(bc) + splat(a.eps) : fma2(M_::sqrt(v), splat(bc), splat(a.eps));
            param = param * splat(decay);
            param -= EXACT ? splat(step) * m / dn : splat(step) * m * M_::rcp(dn);
        };
        if (!ph1) adamw(s, gs, 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f2 gp = fma2(gv[2], splat(cone[e][2]), fma2(gv[1], splat(cone[e][1]), gv[0] * splat(cone[e][0])));
            adamw(w[e], f.p[e] * (gp - ps), e);
        }
    };
    // the iterations in runs of FO_TAB: each run starts by tabulating its bias corrections (thread t: iteration c0 + t; the step count
    // restarts with the second optimiser), in fp64 as torch.optim.AdamW computes them, stored as the two fp32 factors the update uses
    const int n1 = min(a.phase1, a.iters);
    for (int c0 = 0; c0 < a.iters; c0 += FO_TAB) {
        __syncthreads();
        {
            const int itq = c0 + tid, n = (itq < a.phase1 ? itq : itq - a.phase1) + 1;
            double p1 = 1.0, p2 = 1.0, x1 = (double)a.beta1, x2 = (double)a.beta2;      // beta^n by squaring (a dozen fp64 products)
            for (int k = n; k > 0; k >>= 1) {
                if (k & 1) { p1 *= x1; p2 *= x2; }
                x1 *= x1; x2 *= x2;
            }
            const double c1 = 1.0 - p1, c2 = sqrt(1.0 - p2);
            s_tab[tid][0] = (float)((double)a.lr / c1);
            s_tab[tid][1] = EXACT ? (float)c2 : (float)(1.0 / c2);
        }
        __syncthreads();
        const int c1 = min(c0 + FO_TAB, a.iters);
        int it = c0;
        for (; it < min(c1, n1); ++it) {
            if (it == a.iters - 1) report();
            iteration(it, std::true_type{});
        }
        for (; it < c1; ++it) {
            if (it == a.iters - 1) report();
            iteration(it, std::false_type{});
        }
    }
    if (valid) {
        const long long sb = (long long)batch * B + b, ia0 = sb * 32 + l16, ia1 = ia0 + 16;
        a.scale_out[ia0] = s.x; a.scale_out[ia1] = s.y;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a.weight_out[ia0 * 8 + e] = w[e].x; a.weight_out[ia1 * 8 + e] = w[e].y; }
    }
}

}  // namespace

extern "C" int vpho_anchor_frames_f32(const vpho_anchor_tables* t, const float* verts, long long n_hands, float* pts, float* frames, void* stream) {
    VPHO_REQUIRE(t && t->face_idx && t->anchor_weight && t->vert2joint && t->skeleton && verts && pts && frames && n_hands > 0, "vpho_anchor_frames_f32: bad argument");
    AnchorFrameArgs a;
    a.verts = verts; a.t = *t; a.pts = pts; a.frames = frames;
    hipLaunchKernelGGL(anchor_frames_kernel, dim3((unsigned)n_hands), dim3(256), 0, (hipStream_t)stream, a);
    return vpho::check_launch("anchor_frames_kernel");
}

extern "C" int vpho_force_optimize_f32(const float* pts, const float* frames, const float* gravity, const float* com, const float* force_contact,
                                       const unsigned char* is_grasped, int n_batches, int B, int iters, int phase1_iters, float lr,
                                       float* force_local, float* force_global, float* scale, float* weight, float* losses, void* stream) {
    VPHO_REQUIRE(pts && frames && gravity && com && force_contact && is_grasped && force_local && force_global && scale && weight && losses,
                 "vpho_force_optimize_f32: null tensor");
    VPHO_REQUIRE(n_batches > 0 && B > 0 && B <= FO_MAXB && iters > 0 && phase1_iters >= 0, "vpho_force_optimize_f32: batch of %d samples (max %d), %d iterations", B, FO_MAXB, iters);
    FoArgs a;
    a.pts = pts; a.frames = frames; a.gravity = gravity; a.com = com; a.fc = force_contact; a.grasped = is_grasped;
    a.fB = (float)B; a.invB = 1.f / (float)B; a.invB32 = 1.f / (float)(B * 32);
    a.B = B; a.iters = iters; a.phase1 = phase1_iters; a.lr = lr; a.wd = 0.01f; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f; a.friction = 0.8f;
    for (int k = 0; k < 8; ++k) {
        const float ang = (float)k * (2.0f * 3.14159265358979323846f / 8.0f);
        a.cone[k][0] = cosf(ang) / 8.f * a.friction; a.cone[k][1] = sinf(ang) / 8.f * a.friction; a.cone[k][2] = 1.f / 8.f;
    }
    a.fl_out = force_local; a.fg_out = force_global; a.scale_out = scale; a.weight_out = weight; a.losses_out = losses;
    // vector-ALU work of one iteration per (pair, anchor) item, counted on the reference's arithmetic (add / mul / fma = 1 / 1 / 2,
    // division, sqrt, exp, log = 1 each): forward 174 (soft-max over 8 cone anchors 39, cone mix + normalise 63, frame 15, resultant /
    // moment / norms incl. the per-sample reductions 57), backward 190, AdamW on 1 + 8 parameters 116 = 480 flop; x 32 anchors x iters per pair
    vpho::ProfScope prof(vpho::PROF_FORCE_OPTIM, (hipStream_t)stream, 480.0 * 32.0 * (double)iters * (double)n_batches * B, 0.0);
    constexpr int mom_lds = 2 * 9 * FO_IPT * FO_THREADS * (int)sizeof(float);      // 144 KB of the CU's 160 KB
    static const bool exact = [] { const char* e = getenv("VPHO_FORCE_EXACT"); return e && e[0] == '1'; }();
    if (exact) VPHO_DYN_LDS(force_optim_kernel<true>, mom_lds);
    else VPHO_DYN_LDS(force_optim_kernel<false>, mom_lds);
    if (exact) hipLaunchKernelGGL(force_optim_kernel<true>, dim3(n_batches), dim3(FO_THREADS), mom_lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(force_optim_kernel<false>, dim3(n_batches), dim3(FO_THREADS), mom_lds, (hipStream_t)stream, a);
    return vpho::check_launch("force_optim_kernel");
}
