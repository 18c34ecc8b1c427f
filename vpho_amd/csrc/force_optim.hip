// Pseudo-force label optimisation (BASELINE.json configs[4]): the 3000-iteration AdamW loop of
// ForceOptimizer.optimize_batch (lib/engine/force_optimization.py:110-207) as ONE persistent kernel per batch:
// forward (friction-cone mix -> anchor frames -> resultant force / moment / contact-distribution terms), analytic backward
// and the AdamW update all stay in registers; the constant anchor points and frames (the reference recomputes
// VERT2ANCHOR(vert) every iteration, :139) are computed once by anchor_frames_kernel.
// One thread per (sample, anchor) item: a sample's 32 anchors sit in one half-wave (shuffle reductions); the only
// cross-sample coupling is the detached batch-mean force loss (`sum_weight`, :146), one LDS reduction per iteration.
// Latency-bound scalar work: no MFMA, no HBM traffic inside the loop.
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
    float cone[8][3];       // friction-cone anchors (physics.py:183-188,281-282), wave-uniform: kernel arguments stay in scalar registers
    float *fl_out, *fg_out, *scale_out, *weight_out, *losses_out;
};

__device__ inline float half_sum(float v) {                 // sum over the 32 lanes of this half-wave (= one sample)
    // (the same butterfly through DPP operands + one ds_swizzle instead of five ds_bpermute: bit-identical, and no faster -- 81.5 vs
    // 81.7 ms per 3000 iterations: the loop is bound by vector-ALU issue, the permutes' latency is hidden by the other waves)
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(FO_THREADS) void force_optim_kernel(const FoArgs a) {
    __shared__ float s_red[2][FO_MAXB];
    __shared__ float s_fin[4][FO_MAXB];
    const int batch = blockIdx.x, B = a.B, tid = threadIdx.x;
    const long long base = (long long)batch * B;
    const int n_items = B * 32;

    const auto& cone = a.cone;

    // per-item constants and parameters
    float F[FO_IPT][9], arm[FO_IPT][3], g[FO_IPT][3], fcn[FO_IPT], mask[FO_IPT];
    float s[FO_IPT], w[FO_IPT][8];
    // AdamW moments of the 8 cone weights: touched once per iteration, 32 registers per thread that the 128-register budget of a
    // 1024-thread workgroup does not have (they went to scratch: 270 spilled registers) -> LDS, [moment][e][item], each thread its own
    // slots (consecutive threads = consecutive words: conflict-free, no barrier needed)
    extern __shared__ float s_mom[];
    auto MW = [&](int k, int e) -> float& { return s_mom[(e * FO_IPT + k) * FO_THREADS + tid]; };
    auto VW = [&](int k, int e) -> float& { return s_mom[((8 + e) * FO_IPT + k) * FO_THREADS + tid]; };
    auto MS = [&](int k) -> float& { return s_mom[(16 * FO_IPT + k) * FO_THREADS + tid]; };        // ... and of the scale
    auto VS = [&](int k) -> float& { return s_mom[(17 * FO_IPT + k) * FO_THREADS + tid]; };
    bool valid[FO_IPT];
#pragma unroll
    for (int k = 0; k < FO_IPT; ++k) {
        const int item = tid + FO_THREADS * k;
        valid[k] = item < n_items;
        const int b = valid[k] ? item >> 5 : 0, an = item & 31;
        const long long sb = base + b, ia = sb * 32 + an;
        for (int e = 0; e < 9; ++e) F[k][e] = valid[k] ? a.frames[ia * 9 + e] : 0.f;
        for (int c = 0; c < 3; ++c) {
            g[k][c] = a.gravity[sb * 3 + c];
            arm[k][c] = valid[k] ? a.pts[ia * 3 + c] - a.com[sb * 3 + c] : 0.f;
        }
        const float fc = valid[k] ? a.fc[ia] : 0.f;
        const float fnorm = sqrtf(half_sum(fc * fc));
        fcn[k] = fc / (fnorm + 1e-8f);
        mask[k] = (valid[k] && fc > 0.1f) ? 1.f : 0.f;
        s[k] = 0.05f; MS(k) = 0.f; VS(k) = 0.f;
        for (int e = 0; e < 8; ++e) { w[k][e] = 0.f; MW(k, e) = 0.f; VW(k, e) = 0.f; }
    }

    double b1t = 1.0, b2t = 1.0;
    for (int it = 0; it < a.iters; ++it) {
        if (it == a.phase1) {            // optimizer2 starts with fresh moments and step count (two AdamW objects, :36-37)
            b1t = b2t = 1.0;
#pragma unroll
            for (int k = 0; k < FO_IPT; ++k) { MS(k) = 0.f; VS(k) = 0.f; for (int e = 0; e < 8; ++e) { MW(k, e) = 0.f; VW(k, e) = 0.f; } }
        }
        const bool ph1 = it < a.phase1, last = it == a.iters - 1;
        // ---------------- forward ----------------
        float p[FO_IPT][8], vdir[FO_IPT][3], vn[FO_IPT], d[FO_IPT][3], se[FO_IPT], ase[FO_IPT];
        float R[FO_IPT][3], M[FO_IPT][3], rn[FO_IPT], snorm[FO_IPT];
#pragma unroll
        for (int k = 0; k < FO_IPT; ++k) {
            se[k] = s[k] * mask[k];
            ase[k] = fabsf(se[k]);
            float mx = w[k][0];
            for (int e = 1; e < 8; ++e) mx = fmaxf(mx, w[k][e]);
            float sum = 0.f;
            for (int e = 0; e < 8; ++e) { p[k][e] = expf(w[k][e] - mx); sum += p[k][e]; }
            for (int e = 0; e < 8; ++e) p[k][e] /= sum;
            for (int c = 0; c < 3; ++c) { float t = 0.f; for (int e = 0; e < 8; ++e) t += p[k][e] * cone[e][c]; vdir[k][c] = t; }
            vn[k] = sqrtf(vdir[k][0] * vdir[k][0] + vdir[k][1] * vdir[k][1] + vdir[k][2] * vdir[k][2]);
            float fl[3], fg[3];
            for (int c = 0; c < 3; ++c) { d[k][c] = vdir[k][c] / (vn[k] + 1e-8f); fl[c] = d[k][c] * ase[k]; }
            for (int j = 0; j < 3; ++j) fg[j] = fl[0] * F[k][j * 3 + 0] + fl[1] * F[k][j * 3 + 1] + fl[2] * F[k][j * 3 + 2];
            for (int c = 0; c < 3; ++c) R[k][c] = half_sum(fg[c]);
            M[k][0] = half_sum(arm[k][1] * fg[2] - arm[k][2] * fg[1]);
            M[k][1] = half_sum(arm[k][2] * fg[0] - arm[k][0] * fg[2]);
            M[k][2] = half_sum(arm[k][0] * fg[1] - arm[k][1] * fg[0]);
            const float rx = R[k][0] + g[k][0], ry = R[k][1] + g[k][1], rz = R[k][2] + g[k][2];
            rn[k] = sqrtf(rx * rx + ry * ry + rz * rz);
            snorm[k] = sqrtf(half_sum(se[k] * se[k]));
        }
        // batch mean of the resultant-force norm (force_loss; detached as sum_weight)
        float sw = 0.f;
        if (!ph1 || last) {
            float* red = s_red[it & 1];
#pragma unroll
            for (int k = 0; k < FO_IPT; ++k) if (valid[k] && (tid & 31) == 0) red[(tid + FO_THREADS * k) >> 5] = rn[k];
            __syncthreads();
            for (int b = 0; b < B; ++b) sw += red[b];
            sw /= (float)B;
        }
        const float cm = 30.f / (100.f * sw * sw + 1e-8f), cd = 0.1f / (1000.f * sw * sw + 1e-8f);
        // ---------------- backward ----------------
        float gs[FO_IPT], gw[FO_IPT][8], dist[FO_IPT];
#pragma unroll
        for (int k = 0; k < FO_IPT; ++k) {
            float gfg[3];
            const float sn = se[k] / (snorm[k] + 1e-8f);
            const float r = fcn[k] / (sn + 1e-8f);
            dist[k] = logf(fabsf(r) + 1e-8f) * mask[k];
            float gse = 0.f;                                    // d loss / d (s*mask)
            if (ph1) {
                // gravity_loss = mean_b (cos_b - 1)^2, cos_b = R_b . (-g_b)
                const float cosb = -(R[k][0] * g[k][0] + R[k][1] * g[k][1] + R[k][2] * g[k][2]);
                const float c2 = 2.f * (cosb - 1.f) / (float)B;
                for (int c = 0; c < 3; ++c) gfg[c] = -c2 * g[k][c];
            } else {
                const float rx = R[k][0] + g[k][0], ry = R[k][1] + g[k][1], rz = R[k][2] + g[k][2];
                const float inv = rn[k] > 0.f ? 1.f / (rn[k] * (float)B) : 0.f;            // torch.norm backward: 0 at 0
                gfg[0] = rx * inv; gfg[1] = ry * inv; gfg[2] = rz * inv;
                const float mn = sqrtf(M[k][0] * M[k][0] + M[k][1] * M[k][1] + M[k][2] * M[k][2]);
                const float mi = mn > 0.f ? cm / (mn * (float)B) : 0.f;
                const float u[3] = {M[k][0] * mi, M[k][1] * mi, M[k][2] * mi};
                gfg[0] += u[1] * arm[k][2] - u[2] * arm[k][1];                               // u x arm
                gfg[1] += u[2] * arm[k][0] - u[0] * arm[k][2];
                gfg[2] += u[0] * arm[k][1] - u[1] * arm[k][0];
                // dist_loss = cd * mean(dist^2); d dist / d se = mask * sign(r)/(|r|+1e-8) * (-fcn/(sn+1e-8)^2) / (snorm+1e-8)
                const float sgn = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
                const float ddist = mask[k] * (sgn / (fabsf(r) + 1e-8f)) * (-fcn[k] / ((sn + 1e-8f) * (sn + 1e-8f))) / (snorm[k] + 1e-8f);
                gse += cd * 2.f * dist[k] / (float)(B * 32) * ddist;
            }
            float gfl[3];
            for (int i = 0; i < 3; ++i) gfl[i] = gfg[0] * F[k][0 * 3 + i] + gfg[1] * F[k][1 * 3 + i] + gfg[2] * F[k][2 * 3 + i];
            const float gase = gfl[0] * d[k][0] + gfl[1] * d[k][1] + gfl[2] * d[k][2];
            const float sg = se[k] > 0.f ? 1.f : (se[k] < 0.f ? -1.f : 0.f);
            gse += gase * sg;
            gs[k] = gse * mask[k];
            float gd[3] = {gfl[0] * ase[k], gfl[1] * ase[k], gfl[2] * ase[k]};
            const float den = vn[k] + 1e-8f;
            const float dot = gd[0] * vdir[k][0] + gd[1] * vdir[k][1] + gd[2] * vdir[k][2];
            float gv[3];
            for (int c = 0; c < 3; ++c) gv[c] = gd[c] / den - (vn[k] > 0.f ? dot * vdir[k][c] / (vn[k] * den * den) : 0.f);
            float gp[8], ps = 0.f;
            for (int e = 0; e < 8; ++e) { gp[e] = gv[0] * cone[e][0] + gv[1] * cone[e][1] + gv[2] * cone[e][2]; ps += p[k][e] * gp[e]; }
            for (int e = 0; e < 8; ++e) gw[k][e] = p[k][e] * (gp[e] - ps);
        }
        // ---------------- report the four losses of the last iteration (:156-207 prints them every iteration) --------
        if (last) {
#pragma unroll
            for (int k = 0; k < FO_IPT; ++k) {
                const float d2 = half_sum(dist[k] * dist[k]);
                if (valid[k] && (tid & 31) == 0) {
                    const int b = (tid + FO_THREADS * k) >> 5;
                    const float cosb = -(R[k][0] * g[k][0] + R[k][1] * g[k][1] + R[k][2] * g[k][2]);
                    s_fin[0][b] = rn[k];
                    s_fin[1][b] = (cosb - 1.f) * (cosb - 1.f);
                    s_fin[2][b] = sqrtf(M[k][0] * M[k][0] + M[k][1] * M[k][1] + M[k][2] * M[k][2]);
                    s_fin[3][b] = d2;
                }
            }
            __syncthreads();
            if (tid < 4) {
                float t = 0.f;
                for (int b = 0; b < B; ++b) t += s_fin[tid][b];
                t /= (float)B;
                if (tid == 2) t *= cm;
                if (tid == 3) t = t / 32.f * cd;
                a.losses_out[batch * 4 + tid] = t;
            }
#pragma unroll
            for (int k = 0; k < FO_IPT; ++k) {
                if (!valid[k]) continue;
                const int item = tid + FO_THREADS * k;
                const long long sb = base + (item >> 5), ia = sb * 32 + (item & 31);
                const float keep = a.grasped[sb] ? 1.f : 0.f;                                   // :199-202
                float fl[3], fg[3];                                                            // the forward's values again (same operations)
                for (int c = 0; c < 3; ++c) fl[c] = d[k][c] * ase[k];
                for (int j = 0; j < 3; ++j) fg[j] = fl[0] * F[k][j * 3 + 0] + fl[1] * F[k][j * 3 + 1] + fl[2] * F[k][j * 3 + 2];
                for (int c = 0; c < 3; ++c) { a.fl_out[ia * 3 + c] = fl[c] * keep; a.fg_out[ia * 3 + c] = fg[c] * keep; }
            }
        }
        // ---------------- AdamW (torch.optim.AdamW defaults: weight_decay 0.01) ----------------
        b1t *= (double)a.beta1; b2t *= (double)a.beta2;
        const float step = (float)((double)a.lr / (1.0 - b1t)), bc2s = (float)sqrt(1.0 - b2t), decay = 1.f - a.lr * a.wd;
#pragma unroll
        for (int k = 0; k < FO_IPT; ++k) {
            if (!ph1) {
                s[k] *= decay;
                const float m = a.beta1 * MS(k) + (1.f - a.beta1) * gs[k];
                const float v = a.beta2 * VS(k) + (1.f - a.beta2) * gs[k] * gs[k];
                MS(k) = m; VS(k) = v;
                s[k] -= step * m / (sqrtf(v) / bc2s + a.eps);
            }
            for (int e = 0; e < 8; ++e) {
                w[k][e] *= decay;
                const float m = a.beta1 * MW(k, e) + (1.f - a.beta1) * gw[k][e];
                const float v = a.beta2 * VW(k, e) + (1.f - a.beta2) * gw[k][e] * gw[k][e];
                MW(k, e) = m; VW(k, e) = v;
                w[k][e] -= step * m / (sqrtf(v) / bc2s + a.eps);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < FO_IPT; ++k) {
        if (!valid[k]) continue;
        const int item = tid + FO_THREADS * k;
        const long long ia = (base + (item >> 5)) * 32 + (item & 31);
        a.scale_out[ia] = s[k];
        for (int e = 0; e < 8; ++e) a.weight_out[ia * 8 + e] = w[k][e];
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
    a.B = B; a.iters = iters; a.phase1 = phase1_iters; a.lr = lr; a.wd = 0.01f; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f; a.friction = 0.8f;
    for (int k = 0; k < 8; ++k) {
        const float ang = (float)k * (2.0f * 3.14159265358979323846f / 8.0f);
        a.cone[k][0] = cosf(ang) / 8.f * a.friction; a.cone[k][1] = sinf(ang) / 8.f * a.friction; a.cone[k][2] = 1.f / 8.f;
    }
    a.fl_out = force_local; a.fg_out = force_global; a.scale_out = scale; a.weight_out = weight; a.losses_out = losses;
    // vector-ALU work of one iteration per (pair, anchor) item, counted on the kernel above (add / mul / fma = 1 / 1 / 2, division, sqrt,
    // exp, log = 1 each): forward 174 (soft-max over 8 cone anchors 39, cone mix + normalise 63, frame 15, resultant / moment / norms
    // incl. the 7 half-wave reductions 57), backward 190, AdamW on 1 + 8 parameters 116 = 480 flop; x 32 anchors x iters per pair
    vpho::ProfScope prof(vpho::PROF_FORCE_OPTIM, (hipStream_t)stream, 480.0 * 32.0 * (double)iters * (double)n_batches * B, 0.0);
    constexpr int mom_lds = 2 * 9 * FO_IPT * FO_THREADS * (int)sizeof(float);      // 144 KB of the CU's 160 KB
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        VPHO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(force_optim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, mom_lds));
        lds_opt_in = true;
    }
    hipLaunchKernelGGL(force_optim_kernel, dim3(n_batches), dim3(FO_THREADS), mom_lds, (hipStream_t)stream, a);
    return vpho::check_launch("force_optim_kernel");
}
