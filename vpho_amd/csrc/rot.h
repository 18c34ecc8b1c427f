// Device-side rotation conversions with the exact formulas of pytorch3d.transforms.rotation_conversions (0.7.x) that the
// reference calls (VPHO.py:9-12; head_mano.py:5; head_object.py:3; aggregation.py:5-14), templated on float/double
// (the object-pose fuse path runs in fp64, quirk Q5), plus manopth's Rodrigues-via-quaternion (rodrigues_layer.py).
#pragma once
#include <hip/hip_runtime.h>

namespace vpho {

template <typename T> __device__ inline T t_sqrt(T x);
template <> __device__ inline float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ inline double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ inline T t_max(T a, T b) { return a > b ? a : b; }

// F.normalize: x / max(||x||, 1e-12)
template <typename T>
__device__ inline void normalize3(const T* v, T* o) {
    const T n = t_max(t_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), (T)1e-12);
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

// rotation_6d_to_matrix: rows b1, b2, b3 (row-major R[9])
template <typename T>
__device__ inline void rot6d_to_matrix(const T* d6, T* R) {
    T b1[3], b2[3];
    normalize3(d6, b1);
    const T dot = b1[0] * d6[3] + b1[1] * d6[4] + b1[2] * d6[5];
    T u[3] = {d6[3] - dot * b1[0], d6[4] - dot * b1[1], d6[5] - dot * b1[2]};
    normalize3(u, b2);
    R[0] = b1[0]; R[1] = b1[1]; R[2] = b1[2];
    R[3] = b2[0]; R[4] = b2[1]; R[5] = b2[2];
    R[6] = b1[1] * b2[2] - b1[2] * b2[1];
    R[7] = b1[2] * b2[0] - b1[0] * b2[2];
    R[8] = b1[0] * b2[1] - b1[1] * b2[0];
}

// matrix_to_quaternion (best-conditioned candidate, then standardize: real part >= 0)
template <typename T>
__device__ inline void matrix_to_quaternion(const T* m, T* q) {
    const T m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    T qa[4] = {(T)1 + m00 + m11 + m22, (T)1 + m00 - m11 - m22, (T)1 - m00 + m11 - m22, (T)1 - m00 - m11 + m22};
    for (int i = 0; i < 4; ++i) qa[i] = qa[i] > 0 ? t_sqrt(qa[i]) : (T)0;
    int best = 0;
    for (int i = 1; i < 4; ++i) if (qa[i] > qa[best]) best = i;       // first maximum, as torch.argmax
    T c[4];
    if (best == 0)      { c[0] = qa[0] * qa[0]; c[1] = m21 - m12; c[2] = m02 - m20; c[3] = m10 - m01; }
    else if (best == 1) { c[0] = m21 - m12; c[1] = qa[1] * qa[1]; c[2] = m10 + m01; c[3] = m02 + m20; }
    else if (best == 2) { c[0] = m02 - m20; c[1] = m10 + m01; c[2] = qa[2] * qa[2]; c[3] = m12 + m21; }
    else                { c[0] = m10 - m01; c[1] = m20 + m02; c[2] = m21 + m12; c[3] = qa[3] * qa[3]; }
    const T den = (T)2 * t_max(qa[best], (T)0.1);
    for (int i = 0; i < 4; ++i) q[i] = c[i] / den;
    if (q[0] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

template <typename T> __device__ inline T t_atan2(T y, T x);
template <> __device__ inline float t_atan2<float>(float y, float x) { return atan2f(y, x); }
template <> __device__ inline double t_atan2<double>(double y, double x) { return atan2(y, x); }
template <typename T> __device__ inline T t_sin(T x);
template <> __device__ inline float t_sin<float>(float x) { return sinf(x); }
template <> __device__ inline double t_sin<double>(double x) { return sin(x); }
template <typename T> __device__ inline T t_cos(T x);
template <> __device__ inline float t_cos<float>(float x) { return cosf(x); }
template <> __device__ inline double t_cos<double>(double x) { return cos(x); }
template <typename T> __device__ inline T t_abs(T x) { return x < 0 ? -x : x; }

template <typename T>
__device__ inline void quaternion_to_axis_angle(const T* q, T* aa) {
    const T n = t_sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const T half = t_atan2(n, q[0]);
    const T ang = (T)2 * half;
    const T s = t_abs(ang) < (T)1e-6 ? (T)0.5 - (ang * ang) / (T)48 : t_sin(half) / ang;
    aa[0] = q[1] / s; aa[1] = q[2] / s; aa[2] = q[3] / s;
}

template <typename T>
__device__ inline void axis_angle_to_quaternion(const T* aa, T* q) {
    const T ang = t_sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    const T half = ang * (T)0.5;
    const T s = t_abs(ang) < (T)1e-6 ? (T)0.5 - (ang * ang) / (T)48 : t_sin(half) / ang;
    q[0] = t_cos(half); q[1] = aa[0] * s; q[2] = aa[1] * s; q[3] = aa[2] * s;
}

template <typename T>
__device__ inline void quaternion_to_matrix(const T* q, T* R) {
    const T r = q[0], i = q[1], j = q[2], k = q[3];
    const T two_s = (T)2 / (r * r + i * i + j * j + k * k);
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}

// manopth rodrigues_layer.batch_rodrigues + quat2mat (angle = ||aa + 1e-8||)
__device__ inline void mano_rodrigues(const float* aa, float* R) {
    const float ax = aa[0] + 1e-8f, ay = aa[1] + 1e-8f, az = aa[2] + 1e-8f;
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    const float nx = aa[0] / angle, ny = aa[1] / angle, nz = aa[2] / angle;
    const float half = angle * 0.5f;
    const float c = cosf(half), s = sinf(half);
    float w = c, x = s * nx, y = s * ny, z = s * nz;
    const float qn = sqrtf(w * w + x * x + y * y + z * z);
    w /= qn; x /= qn; y /= qn; z /= qn;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;      R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;      R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;      R[7] = 2 * wx + 2 * yz;      R[8] = w2 - x2 - y2 + z2;
}

// Largest-eigenvalue eigenvector of a symmetric 4x4 (cyclic Jacobi in the matrix' own precision; sweeps until the off-diagonal mass is
// below the rounding of the diagonal -- 3-5 sweeps, quadratic convergence -- at most 12) --
// transform_fn.average_quaternion's torch.linalg.eigh(A)[1][..., -1]; sign fixed by the caller.
template <typename T>
__device__ inline void sym4_top_eigenvector(T A[4][4], T* v) {
    // every index into A and V is a compile-time constant (the (p, q) pairs and the k loops are unrolled, the final column is picked by
    // selects): the matrices stay in registers -- indexed by loop variables they lived in scratch (80-144 B per lane, 9 x the fuse
    // kernels' algorithmic HBM bytes).  Same operations in the same order: bit-identical results.
    T V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        T off = 0, dg = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            dg += A[p][p] * A[p][p];
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        }
        // converged: a rotation by an angle below eps / 4 no longer changes a digit of A or V (c = 1, s * a < ulp(a))
        if (off < (T)1e-40 || off <= dg * (sizeof(T) == 4 ? (T)2e-16 : (T)1e-33)) break;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                // exactly (or denormally) zero off-diagonal: nothing to rotate.  Matters for rank-deficient moment matrices
                // (k identical quaternions): a second sweep would otherwise evaluate 0/0 between two zero eigenvalues.
                if (!(t_abs(A[p][q]) > (sizeof(T) == 4 ? (T)1e-30 : (T)1e-290))) continue;
                const T theta = (A[q][q] - A[p][p]) / ((T)2 * A[p][q]);
                const T t = (theta >= 0 ? (T)1 : (T)-1) / (t_abs(theta) + t_sqrt(theta * theta + (T)1));
                const T c = (T)1 / t_sqrt(t * t + (T)1), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const T akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const T apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const T vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
        }
    }
    T top = A[0][0];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = V[k][0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const bool better = A[i][i] > top;                       // first of equal maxima, as before
        top = better ? A[i][i] : top;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = better ? V[k][i] : v[k];
    }
}

}  // namespace vpho
