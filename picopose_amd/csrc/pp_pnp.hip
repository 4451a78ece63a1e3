// PnP / RANSAC pose recovery on the GPU — the host step of the reference
// (utils/pose_recovery.py:68-105: gather 2D/3D coordinates, bring the 3-D points to the object
// frame, cv2.solvePnPRansac(EPNP, 150 iterations, 2 px), Rodrigues) as ONE batched launch:
// one 256-thread workgroup per (instance, hypothesis) problem, no per-problem host sync.
//
// OpenCV (opencv-python 4.9, requirements.txt:3) is not vendored in the reference and is absent
// from the build image, so this is a from-scratch restatement of the published algorithm with
// the reference's hyper-parameters: RANSAC over 5-point minimal samples, EPnP (Lepetit,
// Moreno-Noguer, Fua 2009: 4 control points, 12x12 null space, 3 beta approximations + Gauss-
// Newton, Horn alignment) as the model solver, squared reprojection error <= 2^2 as the inlier
// test, EPnP refit on the inlier set.  Sampling uses a counter-based hash, not OpenCV's RNG, and
// all 150 iterations are run (OpenCV stops early at 99 % confidence): parity with cv2 is
// "unpinned"  and is defined by known-answer tests instead.  fp64 throughout.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/picopose_hip.h"
#include "pp_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXP = 4096;
constexpr int SAMPLE = 5;  // minimal sample size of solvePnPRansac for EPNP

// ------------------------------------------------------------------ small dense helpers (double)
// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 12): a is destroyed,
// w = eigenvalues, v[k*n + i] = component i of eigenvector k; then sorted ascending.
template <int N>
__device__ void jacobi_eig(double* a, double* w, double* v) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) v[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) off += a[p * N + q] * a[p * N + q];
        double diag = 0.0;
        for (int p = 0; p < N; ++p) diag += a[p * N + p] * a[p * N + p];
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double apq = a[p * N + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (a[q * N + q] - a[p * N + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) {  // A <- A J
                    const double akp = a[k * N + p], akq = a[k * N + q];
                    a[k * N + p] = c * akp - s * akq;
                    a[k * N + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {  // A <- J^T A
                    const double apk = a[p * N + k], aqk = a[q * N + k];
                    a[p * N + k] = c * apk - s * aqk;
                    a[q * N + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {  // eigenvectors as rows of v
                    const double vpk = v[p * N + k], vqk = v[q * N + k];
                    v[p * N + k] = c * vpk - s * vqk;
                    v[q * N + k] = s * vpk + c * vqk;
                }
            }
    }
    for (int i = 0; i < N; ++i) w[i] = a[i * N + i];
    for (int i = 0; i < N - 1; ++i) {  // selection sort, ascending
        int m = i;
        for (int j = i + 1; j < N; ++j)
            if (w[j] < w[m]) m = j;
        if (m != i) {
            const double tw = w[i];
            w[i] = w[m];
            w[m] = tw;
            for (int k = 0; k < N; ++k) {
                const double tv = v[i * N + k];
                v[i * N + k] = v[m * N + k];
                v[m * N + k] = tv;
            }
        }
    }
}

// least squares  min |A x - b|  for A (6 x C), via normal equations + Gaussian elimination
template <int C>
__device__ void lstsq6(const double (*A)[C], const double* b, double* x) {
    double n[C][C + 1];
    for (int i = 0; i < C; ++i) {
        for (int j = 0; j < C; ++j) {
            double s = 0.0;
            for (int r = 0; r < 6; ++r) s += A[r][i] * A[r][j];
            n[i][j] = s;
        }
        double s = 0.0;
        for (int r = 0; r < 6; ++r) s += A[r][i] * b[r];
        n[i][C] = s;
    }
    for (int i = 0; i < C; ++i) {
        int p = i;
        for (int r = i + 1; r < C; ++r)
            if (fabs(n[r][i]) > fabs(n[p][i])) p = r;
        if (p != i)
            for (int c = 0; c <= C; ++c) {
                const double t = n[i][c];
                n[i][c] = n[p][c];
                n[p][c] = t;
            }
        const double d = fabs(n[i][i]) > 1e-300 ? n[i][i] : 1e-300;
        for (int r = i + 1; r < C; ++r) {
            const double f = n[r][i] / d;
            for (int c = i; c <= C; ++c) n[r][c] -= f * n[i][c];
        }
    }
    for (int i = C - 1; i >= 0; --i) {
        double s = n[i][C];
        for (int c = i + 1; c < C; ++c) s -= n[i][c] * x[c];
        x[i] = s / (fabs(n[i][i]) > 1e-300 ? n[i][i] : 1e-300);
    }
}

__device__ inline double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// rotation of the Horn alignment: the proper rotation closest to U V^T of the SVD of the 3x3
// cross-covariance H = sum (pc)(pw)^T (Kabsch: when det(U V^T) < 0 the direction of the smallest
// singular value flips, which is what completing both bases to right-handed triads does)
__device__ void horn_rotation(const double* H, double* R) {
    double hth[9], w[3], v[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) hth[i * 3 + j] = H[0 * 3 + i] * H[0 * 3 + j] + H[1 * 3 + i] * H[1 * 3 + j] + H[2 * 3 + i] * H[2 * 3 + j];
    jacobi_eig<3>(hth, w, v);  // ascending; rows of v = right singular vectors
    double u[9];               // columns u_k = H v_k / sigma_k for the two largest, third by cross product
    for (int k = 2; k >= 1; --k) {
        double x[3];
        for (int i = 0; i < 3; ++i) x[i] = H[i * 3 + 0] * v[k * 3 + 0] + H[i * 3 + 1] * v[k * 3 + 1] + H[i * 3 + 2] * v[k * 3 + 2];
        const double nn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        for (int i = 0; i < 3; ++i) u[k * 3 + i] = nn > 1e-300 ? x[i] / nn : (i == k ? 1.0 : 0.0);
    }
    // make u1 orthogonal to u2 (guards a tiny sigma_1), u0 = u1 x u2 ; same for v0 = v1 x v2
    const double d12 = u[3] * u[6] + u[4] * u[7] + u[5] * u[8];
    for (int i = 0; i < 3; ++i) u[3 + i] -= d12 * u[6 + i];
    const double n1 = sqrt(u[3] * u[3] + u[4] * u[4] + u[5] * u[5]);
    for (int i = 0; i < 3; ++i) u[3 + i] /= (n1 > 1e-300 ? n1 : 1.0);
    u[0] = u[4] * u[8] - u[5] * u[7];
    u[1] = u[5] * u[6] - u[3] * u[8];
    u[2] = u[3] * u[7] - u[4] * u[6];
    double v0[3] = {v[4] * v[8] - v[5] * v[7], v[5] * v[6] - v[3] * v[8], v[3] * v[7] - v[4] * v[6]};
    // R = sum_k u_k v_k^T with (u0, v0) completing right-handed triads: det(R) = +1 ...
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = u[0 + i] * v0[j] + u[3 + i] * v[3 + j] + u[6 + i] * v[6 + j];
    // ... which is the det-corrected U V^T (the smallest singular direction is the one that flips)
}

// ------------------------------------------------------------------ EPnP
// MODE 0: one thread solves alone over `n` sampled points (idx[0..n)).
// MODE 1: the whole workgroup solves over the points with use[i] != 0; every thread runs the small
//         algebra redundantly, point loops are strided and block-reduced through `red`.
struct Cam {
    double fu, fv, uc, vc;
};

template <int MODE>
struct PointSet {
    const float* p3;  // [n][3] object-frame points (LDS)
    const float* p2;  // [n][2] pixels (LDS)
    const int* idx;   // MODE 0: sample indices
    const unsigned char* use;  // MODE 1: inlier mask
    int n;            // MODE 0: sample size, MODE 1: total points
    double* red;      // MODE 1: LDS scratch [NT]
};

template <int MODE>
__device__ inline void block_sum(const PointSet<MODE>& ps, double* vals, int cnt) {
    if (MODE == 0) return;
    for (int c = 0; c < cnt; ++c) {
        __syncthreads();
        ps.red[threadIdx.x] = vals[c];
        __syncthreads();
        for (int s = NT / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) ps.red[threadIdx.x] += ps.red[threadIdx.x + s];
            __syncthreads();
        }
        vals[c] = ps.red[0];
    }
    __syncthreads();
}

#define PP_FOR_POINTS(ps, i, ...)                                                          \
    if (MODE == 0) {                                                                       \
        for (int ii_ = 0; ii_ < (ps).n; ++ii_) {                                           \
            const int i = (ps).idx[ii_];                                                   \
            __VA_ARGS__                                                                    \
        }                                                                                  \
    } else {                                                                               \
        for (int i = threadIdx.x; i < (ps).n; i += NT) {                                   \
            if (!(ps).use[i]) continue;                                                    \
            __VA_ARGS__                                                                    \
        }                                                                                  \
    }

// returns the mean reprojection error of the chosen solution; R (row-major) and t
template <int MODE>
__device__ double epnp(const PointSet<MODE>& ps, const Cam& cam, double* R, double* t) {
    // ---- control points: centroid + principal directions
    double acc[16];
    for (int k = 0; k < 4; ++k) acc[k] = 0.0;
    PP_FOR_POINTS(ps, i, { acc[0] += ps.p3[3 * i]; acc[1] += ps.p3[3 * i + 1]; acc[2] += ps.p3[3 * i + 2]; acc[3] += 1.0; })
    block_sum(ps, acc, 4);
    const double n = acc[3];
    double cws[4][3];
    for (int k = 0; k < 3; ++k) cws[0][k] = acc[k] / n;
    for (int k = 0; k < 6; ++k) acc[k] = 0.0;
    PP_FOR_POINTS(ps, i, {
        const double x = ps.p3[3 * i] - cws[0][0], y = ps.p3[3 * i + 1] - cws[0][1], z = ps.p3[3 * i + 2] - cws[0][2];
        acc[0] += x * x; acc[1] += x * y; acc[2] += x * z; acc[3] += y * y; acc[4] += y * z; acc[5] += z * z;
    })
    block_sum(ps, acc, 6);
    {
        double c3[9] = {acc[0], acc[1], acc[2], acc[1], acc[3], acc[4], acc[2], acc[4], acc[5]}, w[3], v[9];
        jacobi_eig<3>(c3, w, v);
        for (int k = 0; k < 3; ++k) {  // largest first, as the SVD ordering of OpenCV's EPnP
            const double kk = sqrt(fmax(w[2 - k], 0.0) / n);
            for (int c = 0; c < 3; ++c) cws[k + 1][c] = cws[0][c] + kk * v[(2 - k) * 3 + c];
        }
    }
    // ---- barycentric coordinates: alpha_{1..3} = CC^-1 (p - c0)
    double cc[9], cci[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) cc[r * 3 + c] = cws[c + 1][r] - cws[0][r];
    {
        const double d = det3(cc);
        const double id = fabs(d) > 1e-300 ? 1.0 / d : 0.0;
        cci[0] = (cc[4] * cc[8] - cc[5] * cc[7]) * id; cci[1] = (cc[2] * cc[7] - cc[1] * cc[8]) * id; cci[2] = (cc[1] * cc[5] - cc[2] * cc[4]) * id;
        cci[3] = (cc[5] * cc[6] - cc[3] * cc[8]) * id; cci[4] = (cc[0] * cc[8] - cc[2] * cc[6]) * id; cci[5] = (cc[2] * cc[3] - cc[0] * cc[5]) * id;
        cci[6] = (cc[3] * cc[7] - cc[4] * cc[6]) * id; cci[7] = (cc[1] * cc[6] - cc[0] * cc[7]) * id; cci[8] = (cc[0] * cc[4] - cc[1] * cc[3]) * id;
    }
    auto alphas = [&](int i, double* a) {
        const double x = ps.p3[3 * i] - cws[0][0], y = ps.p3[3 * i + 1] - cws[0][1], z = ps.p3[3 * i + 2] - cws[0][2];
        a[1] = cci[0] * x + cci[1] * y + cci[2] * z;
        a[2] = cci[3] * x + cci[4] * y + cci[5] * z;
        a[3] = cci[6] * x + cci[7] * y + cci[8] * z;
        a[0] = 1.0 - a[1] - a[2] - a[3];
    };
    // ---- M^T M (12 x 12, upper triangle accumulated)
    double mtm[144];
    {
        double up[78];
        for (int k = 0; k < 78; ++k) up[k] = 0.0;
        PP_FOR_POINTS(ps, i, {
            double a[4], r1[12], r2[12];
            alphas(i, a);
            const double u = ps.p2[2 * i], vv = ps.p2[2 * i + 1];
            for (int j = 0; j < 4; ++j) {
                r1[3 * j] = a[j] * cam.fu; r1[3 * j + 1] = 0.0; r1[3 * j + 2] = a[j] * (cam.uc - u);
                r2[3 * j] = 0.0; r2[3 * j + 1] = a[j] * cam.fv; r2[3 * j + 2] = a[j] * (cam.vc - vv);
            }
            int k = 0;
            for (int r = 0; r < 12; ++r)
                for (int c = r; c < 12; ++c) up[k++] += r1[r] * r1[c] + r2[r] * r2[c];
        })
        block_sum(ps, up, 78);
        int k = 0;
        for (int r = 0; r < 12; ++r)
            for (int c = r; c < 12; ++c) {
                mtm[r * 12 + c] = up[k];
                mtm[c * 12 + r] = up[k++];
            }
    }
    double ew[12], ev[144];
    jacobi_eig<12>(mtm, ew, ev);  // ascending: ev rows 0..3 span the (approximate) null space
    const double* vn[4] = {ev, ev + 12, ev + 24, ev + 36};
    // ---- L (6 x 10) and rho
    double L[6][10], rho[6];
    {
        const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
        double dv[4][6][3];
        for (int i = 0; i < 4; ++i)
            for (int p = 0; p < 6; ++p)
                for (int c = 0; c < 3; ++c) dv[i][p][c] = vn[i][3 * pa[p] + c] - vn[i][3 * pb[p] + c];
        auto dot = [&](int i, int j, int p) { return dv[i][p][0] * dv[j][p][0] + dv[i][p][1] * dv[j][p][1] + dv[i][p][2] * dv[j][p][2]; };
        for (int p = 0; p < 6; ++p) {
            L[p][0] = dot(0, 0, p); L[p][1] = 2 * dot(0, 1, p); L[p][2] = dot(1, 1, p); L[p][3] = 2 * dot(0, 2, p);
            L[p][4] = 2 * dot(1, 2, p); L[p][5] = dot(2, 2, p); L[p][6] = 2 * dot(0, 3, p); L[p][7] = 2 * dot(1, 3, p);
            L[p][8] = 2 * dot(2, 3, p); L[p][9] = dot(3, 3, p);
            double s = 0.0;
            for (int c = 0; c < 3; ++c) s += (cws[pa[p]][c] - cws[pb[p]][c]) * (cws[pa[p]][c] - cws[pb[p]][c]);
            rho[p] = s;
        }
    }
    // ---- three beta initialisations, Gauss-Newton, pose, keep the least reprojection error
    double best_err = 1e300;
    for (int approx = 0; approx < 3; ++approx) {
        double b[4] = {0, 0, 0, 0};
        if (approx == 0) {  // betas10 columns B11 B12 B13 B14
            double A4[6][4], x[4];
            for (int p = 0; p < 6; ++p) { A4[p][0] = L[p][0]; A4[p][1] = L[p][1]; A4[p][2] = L[p][3]; A4[p][3] = L[p][6]; }
            lstsq6<4>(A4, rho, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = -x[1] / b[0]; b[2] = -x[2] / b[0]; b[3] = -x[3] / b[0]; }
            else { b[0] = sqrt(x[0]); b[1] = x[1] / b[0]; b[2] = x[2] / b[0]; b[3] = x[3] / b[0]; }
        } else if (approx == 1) {  // B11 B12 B22
            double A3[6][3], x[3];
            for (int p = 0; p < 6; ++p) { A3[p][0] = L[p][0]; A3[p][1] = L[p][1]; A3[p][2] = L[p][2]; }
            lstsq6<3>(A3, rho, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) b[0] = -b[0];
        } else {  // B11 B12 B22 B13 B23
            double A5[6][5], x[5];
            for (int p = 0; p < 6; ++p)
                for (int c = 0; c < 5; ++c) A5[p][c] = L[p][c];
            lstsq6<5>(A5, rho, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) b[0] = -b[0];
            b[2] = fabs(b[0]) > 1e-300 ? x[3] / b[0] : 0.0;
        }
        if (!(b[0] == b[0]) || !(b[1] == b[1]) || !(b[2] == b[2]) || !(b[3] == b[3])) continue;
        for (int it = 0; it < 5; ++it) {  // Gauss-Newton on the 6 distance constraints
            double A[6][4], rb[6], dx[4];
            for (int p = 0; p < 6; ++p) {
                const double* l = L[p];
                A[p][0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                A[p][1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                A[p][2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
                A[p][3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
                rb[p] = rho[p] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] +
                                  l[4] * b[1] * b[2] + l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] +
                                  l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
            }
            lstsq6<4>(A, rb, dx);
            for (int k = 0; k < 4; ++k) b[k] += dx[k];
        }
        // control points in the camera frame, sign from the depth of the points
        double ccs[4][3];
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 3; ++c) ccs[j][c] = b[0] * vn[0][3 * j + c] + b[1] * vn[1][3 * j + c] + b[2] * vn[2][3 * j + c] + b[3] * vn[3][3 * j + c];
        double s8[16];
        for (int k = 0; k < 7; ++k) s8[k] = 0.0;
        PP_FOR_POINTS(ps, i, {
            double a[4];
            alphas(i, a);
            for (int c = 0; c < 3; ++c) s8[c] += a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c];
            s8[3] += ps.p3[3 * i]; s8[4] += ps.p3[3 * i + 1]; s8[5] += ps.p3[3 * i + 2];
        })
        block_sum(ps, s8, 6);
        if (s8[2] < 0) {  // solve_for_sign (mean depth must be positive)
            for (int j = 0; j < 4; ++j)
                for (int c = 0; c < 3; ++c) ccs[j][c] = -ccs[j][c];
            for (int c = 0; c < 3; ++c) s8[c] = -s8[c];
        }
        double pc0[3], pw0[3];
        for (int c = 0; c < 3; ++c) { pc0[c] = s8[c] / n; pw0[c] = s8[3 + c] / n; }
        double H[9];
        for (int k = 0; k < 9; ++k) H[k] = 0.0;
        PP_FOR_POINTS(ps, i, {
            double a[4], pc[3];
            alphas(i, a);
            for (int c = 0; c < 3; ++c) pc[c] = a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c] - pc0[c];
            const double w0 = ps.p3[3 * i] - pw0[0], w1 = ps.p3[3 * i + 1] - pw0[1], w2 = ps.p3[3 * i + 2] - pw0[2];
            for (int r = 0; r < 3; ++r) { H[r * 3] += pc[r] * w0; H[r * 3 + 1] += pc[r] * w1; H[r * 3 + 2] += pc[r] * w2; }
        })
        block_sum(ps, H, 9);
        double Rc[9], tc[3];
        horn_rotation(H, Rc);
        for (int r = 0; r < 3; ++r) tc[r] = pc0[r] - (Rc[r * 3] * pw0[0] + Rc[r * 3 + 1] * pw0[1] + Rc[r * 3 + 2] * pw0[2]);
        double er[2] = {0.0, 0.0};
        PP_FOR_POINTS(ps, i, {
            const double X = ps.p3[3 * i], Y = ps.p3[3 * i + 1], Z = ps.p3[3 * i + 2];
            const double xc = Rc[0] * X + Rc[1] * Y + Rc[2] * Z + tc[0], yc = Rc[3] * X + Rc[4] * Y + Rc[5] * Z + tc[1];
            const double zc = Rc[6] * X + Rc[7] * Y + Rc[8] * Z + tc[2];
            const double iz = 1.0 / zc;
            const double du = cam.uc + cam.fu * xc * iz - ps.p2[2 * i], dvv = cam.vc + cam.fv * yc * iz - ps.p2[2 * i + 1];
            er[0] += sqrt(du * du + dvv * dvv);
        })
        block_sum(ps, er, 1);
        const double err = er[0] / n;
        if (err == err && err < best_err) {
            best_err = err;
            for (int k = 0; k < 9; ++k) R[k] = Rc[k];
            for (int k = 0; k < 3; ++k) t[k] = tc[k];
        }
    }
    return best_err;
}

__device__ inline unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// One workgroup per problem.
//   tar_pts_2d (P,2,H,W), src_pts_3d (P,3,H,W), K (P,3,3), tem_pose (P,4,4), tar_pts/src_pts (P,N,2) int64
//   out: rot (P,9) f64, tvec (P,3) f64, ratio (P) f64, ok (P) int32, npts (P) int32
__global__ __launch_bounds__(NT) void pnp_ransac_kernel(const float* __restrict__ tar2d, const float* __restrict__ src3d,
                                                        const float* __restrict__ Kmat, const float* __restrict__ tem_pose,
                                                        const int64_t* __restrict__ tar_pts, const int64_t* __restrict__ src_pts,
                                                        int H, int W, int N, int iters, float thresh, double* __restrict__ rot,
                                                        double* __restrict__ tvec, double* __restrict__ ratio,
                                                        int32_t* __restrict__ ok, int32_t* __restrict__ npts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* p3 = (float*)smem;                      // [MAXP][3]
    float* p2 = p3 + 3 * MAXP;                     // [MAXP][2]
    unsigned char* use = (unsigned char*)(p2 + 2 * MAXP);  // [MAXP]
    double* red = (double*)(use + MAXP);           // [NT]
    double* hyp = red + NT;                        // [NT][12]  R, t of every hypothesis
    int* cnt = (int*)(hyp + NT * 12);              // [NT]
    __shared__ int wsum[4], base, best_h, best_c;
    const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t* tp = tar_pts + (size_t)prob * N * 2;
    const int64_t* sp = src_pts + (size_t)prob * N * 2;
    const float* f2 = tar2d + (size_t)prob * 2 * H * W;
    const float* f3 = src3d + (size_t)prob * 3 * H * W;
    const float* P = tem_pose + (size_t)prob * 16;
    const float* Kp = Kmat + (size_t)prob * 9;
    const Cam cam = {(double)Kp[0], (double)Kp[4], (double)Kp[2], (double)Kp[5]};

    // ---- gather the valid correspondences in list order (utils/torch_utils.py:257-284), object frame
    if (tid == 0) base = 0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += NT) {
        const int n = n0 + tid;
        int64_t tx = -1, ty = -1, sx = -1, sy = -1;
        if (n < N) { tx = tp[2 * n]; ty = tp[2 * n + 1]; sx = sp[2 * n]; sy = sp[2 * n + 1]; }
        const bool v = tx != -1 && ty != -1 && sx != -1 && sy != -1;
        const unsigned long long bal = __ballot(v);
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int i = 0; i < wv; ++i) off += wsum[i];
        if (v) {
            const int r = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (r < MAXP) {
                p2[2 * r] = f2[ty * W + tx];
                p2[2 * r + 1] = f2[(size_t)H * W + ty * W + tx];
                const float X = f3[sy * W + sx] - P[3], Y = f3[(size_t)H * W + sy * W + sx] - P[7], Z = f3[(size_t)2 * H * W + sy * W + sx] - P[11];
                // (X - t) @ R_tem  (pose_recovery.py:84): component j = sum_i d_i R[i][j]
                p3[3 * r] = X * P[0] + Y * P[4] + Z * P[8];
                p3[3 * r + 1] = X * P[1] + Y * P[5] + Z * P[9];
                p3[3 * r + 2] = X * P[2] + Y * P[6] + Z * P[10];
            }
        }
        __syncthreads();
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    const int np = base < MAXP ? base : MAXP;
    if (tid == 0) npts[prob] = np;
    auto fail = [&]() {
        if (tid == 0) {
            for (int k = 0; k < 9; ++k) rot[(size_t)prob * 9 + k] = (k % 4 == 0) ? 1.0 : 0.0;
            tvec[(size_t)prob * 3] = 0.0; tvec[(size_t)prob * 3 + 1] = 0.0; tvec[(size_t)prob * 3 + 2] = 1.0;
            ratio[prob] = 0.0;
            ok[prob] = 0;
        }
    };
    if (np < SAMPLE) { fail(); return; }

    // ---- RANSAC hypotheses: thread h solves EPnP on its own 5-point sample
    const int nh = iters < NT ? iters : NT;
    if (tid < nh) {
        int idx[SAMPLE];
        unsigned s = hash32(0x9E3779B9u * (unsigned)(prob + 1) ^ (unsigned)(tid * 7919 + 17));
        for (int k = 0; k < SAMPLE; ++k) {
            for (;;) {
                s = hash32(s + 0x6D2B79F5u);
                const int c = (int)(s % (unsigned)np);
                bool dup = false;
                for (int j = 0; j < k; ++j) dup |= idx[j] == c;
                if (!dup) { idx[k] = c; break; }
            }
        }
        PointSet<0> ps = {p3, p2, idx, nullptr, SAMPLE, nullptr};
        double R[9], t[3];
        const double e = epnp<0>(ps, cam, R, t);
        for (int k = 0; k < 9; ++k) hyp[tid * 12 + k] = e < 1e299 ? R[k] : 0.0;
        for (int k = 0; k < 3; ++k) hyp[tid * 12 + 9 + k] = e < 1e299 ? t[k] : 0.0;
    }
    __syncthreads();
    // ---- score every hypothesis on every point (squared reprojection error <= thresh^2)
    const double th2 = (double)thresh * (double)thresh;
    for (int h = 0; h < nh; ++h) {
        const double* M = hyp + h * 12;
        int c = 0;
        for (int i = tid; i < np; i += NT) {
            const double X = p3[3 * i], Y = p3[3 * i + 1], Z = p3[3 * i + 2];
            const double zc = M[6] * X + M[7] * Y + M[8] * Z + M[11];
            const double iz = 1.0 / zc;
            const double du = cam.uc + cam.fu * (M[0] * X + M[1] * Y + M[2] * Z + M[9]) * iz - p2[2 * i];
            const double dv = cam.vc + cam.fv * (M[3] * X + M[4] * Y + M[5] * Z + M[10]) * iz - p2[2 * i + 1];
            c += (du * du + dv * dv <= th2) ? 1 : 0;  // NaN (degenerate hypothesis) is never an inlier
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) wsum[wv] = c;
        __syncthreads();
        if (tid == 0) cnt[h] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) {
        int bh = 0, bc = -1;
        for (int h = 0; h < nh; ++h)
            if (cnt[h] > bc) { bc = cnt[h]; bh = h; }
        best_h = bh;
        best_c = bc;
    }
    __syncthreads();
    if (best_c < SAMPLE) { fail(); return; }
    {
        const double* M = hyp + best_h * 12;
        for (int i = tid; i < np; i += NT) {
            const double X = p3[3 * i], Y = p3[3 * i + 1], Z = p3[3 * i + 2];
            const double iz = 1.0 / (M[6] * X + M[7] * Y + M[8] * Z + M[11]);
            const double du = cam.uc + cam.fu * (M[0] * X + M[1] * Y + M[2] * Z + M[9]) * iz - p2[2 * i];
            const double dv = cam.vc + cam.fv * (M[3] * X + M[4] * Y + M[5] * Z + M[10]) * iz - p2[2 * i + 1];
            use[i] = (du * du + dv * dv <= th2) ? 1 : 0;
        }
    }
    __syncthreads();
    // ---- refit on the inlier set (all threads cooperate, identical small algebra in every thread)
    PointSet<1> ps = {p3, p2, nullptr, use, np, red};
    double R[9], t[3];
    const double e = epnp<1>(ps, cam, R, t);
    if (tid == 0) {
        const bool good = e < 1e299;
        const double* M = hyp + best_h * 12;
        for (int k = 0; k < 9; ++k) rot[(size_t)prob * 9 + k] = good ? R[k] : M[k];
        for (int k = 0; k < 3; ++k) tvec[(size_t)prob * 3 + k] = good ? t[k] : M[9 + k];
        ratio[prob] = (double)best_c / (double)np;
        ok[prob] = 1;
    }
}

}  // namespace

extern "C" {

int pp_pnp_ransac(const float* tar_pts_2d, const float* src_pts_3d, const float* K, const float* tem_pose,
                  const int64_t* tar_pts, const int64_t* src_pts, int P, int H, int W, int N, int iterations,
                  float reproj_threshold, double* rot, double* tvec, double* inlier_ratio, int32_t* success,
                  int32_t* num_points, void* stream) {
    if (!tar_pts_2d || !src_pts_3d || !K || !tem_pose || !tar_pts || !src_pts || !rot || !tvec || !inlier_ratio ||
        !success || !num_points)
        return PP_EINVAL;
    if (P <= 0 || H <= 0 || W <= 0 || N <= 0 || N > MAXP || iterations <= 0 || reproj_threshold <= 0.f) return PP_EINVAL;
    const size_t smem = (size_t)MAXP * (3 + 2) * sizeof(float) + MAXP + NT * sizeof(double) + (size_t)NT * 12 * sizeof(double) +
                        NT * sizeof(int);
    static bool attr_set[PP_MAX_DEVICES];   // the dynamic-LDS opt-in is per device
    if (!attr_set[pp_cur_device()]) {
        PP_CHECK_HIP(hipFuncSetAttribute((const void*)pnp_ransac_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set[pp_cur_device()] = true;
    }
    hipLaunchKernelGGL(pnp_ransac_kernel, dim3(P), dim3(NT), smem, (hipStream_t)stream, tar_pts_2d, src_pts_3d, K,
                       tem_pose, tar_pts, src_pts, H, W, N, iterations, reproj_threshold, rot, tvec, inlier_ratio,
                       success, num_points);
    return pp_last_launch();
}

}  // extern "C"
