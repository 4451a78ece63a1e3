// PnP / RANSAC pose recovery on the GPU — the host step of the reference
// (utils/pose_recovery.py:68-105: gather 2D/3D coordinates, bring the 3-D points to the object
// frame, cv2.solvePnPRansac(EPNP, 150 iterations, 2 px), Rodrigues) as ONE batched launch:
// one 512-thread workgroup per (instance, hypothesis) problem, no per-problem host sync.
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

constexpr int NT = 512;           // 8 waves, one workgroup per CU (the correspondences of a problem fill most of the LDS)
constexpr int NW = NT / 64;
constexpr int GL = 16;            // lanes of a solver group (12 of them own a column of the 12x12 system)
constexpr int NG = NT / GL;       // solver groups per workgroup: RANSAC hypotheses are solved NG at a time
constexpr int MAXP = 4096;
constexpr int MAXH = 256;         // RANSAC hypotheses kept per problem
constexpr int HB = 128;           // hypotheses per batch (their null-space vectors wait in LDS between the two phases)
constexpr int SAMPLE = 5;         // minimal sample size of solvePnPRansac for EPNP

// ------------------------------------------------------------------ small dense helpers (double)
// Everything below keeps its operands in registers: every array index is a compile-time constant after unrolling
// (a per-lane matrix with run-time indices lives in scratch memory, whose latency made the first version of this kernel
// spend 9 ms on 160 problems).

#define PP_JROT3(A, V, p, q)                                                                   \
    {                                                                                          \
        const double apq = A[p][q];                                                            \
        if (fabs(apq) >= 1e-300) {                                                             \
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);                            \
            const double t_ = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
            const double c_ = 1.0 / sqrt(t_ * t_ + 1.0), s_ = t_ * c_;                         \
            _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                 \
                const double akp = A[k_][p], akq = A[k_][q];                                   \
                A[k_][p] = c_ * akp - s_ * akq;                                                \
                A[k_][q] = s_ * akp + c_ * akq;                                                \
            }                                                                                  \
            _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                 \
                const double apk = A[p][k_], aqk = A[q][k_];                                   \
                A[p][k_] = c_ * apk - s_ * aqk;                                                \
                A[q][k_] = s_ * apk + c_ * aqk;                                                \
            }                                                                                  \
            _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                 \
                const double vpk = V[p][k_], vqk = V[q][k_];                                   \
                V[p][k_] = c_ * vpk - s_ * vqk;                                                \
                V[q][k_] = s_ * vpk + c_ * vqk;                                                \
            }                                                                                  \
        }                                                                                      \
    }

#define PP_CSWAP3(w, V, i, j)                                   \
    if (w[j] < w[i]) {                                          \
        const double tw_ = w[i]; w[i] = w[j]; w[j] = tw_;       \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {      \
            const double tv_ = V[i][k_]; V[i][k_] = V[j][k_]; V[j][k_] = tv_; \
        }                                                       \
    }

// cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix: w ascending, V[k] = eigenvector k (rows)
__device__ inline void eig3_sym(double (&A)[3][3], double (&w)[3], double (&V)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-30 * (diag + 1e-300)) break;
        PP_JROT3(A, V, 0, 1)
        PP_JROT3(A, V, 0, 2)
        PP_JROT3(A, V, 1, 2)
    }
    w[0] = A[0][0]; w[1] = A[1][1]; w[2] = A[2][2];
    // selection sort order of the first version (min to slot 0, then min of the rest to slot 1)
    if (w[1] < w[0] && w[1] <= w[2]) { PP_CSWAP3(w, V, 0, 1) } else if (w[2] < w[0] && w[2] < w[1]) { PP_CSWAP3(w, V, 0, 2) }
    PP_CSWAP3(w, V, 1, 2)
}

// least squares  min |A x - b|  for A (6 x C), via normal equations + Gaussian elimination with partial pivoting
// (the pivot row is brought up by predicated swaps, so that no run-time row index appears)
template <int C>
__device__ inline void lstsq6(const double (&A)[6][C], const double (&b)[6], double (&x)[C]) {
    double n[C][C + 1];
#pragma unroll
    for (int i = 0; i < C; ++i) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r) s += A[r][i] * A[r][j];
            n[i][j] = s;
        }
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) s += A[r][i] * b[r];
        n[i][C] = s;
    }
#pragma unroll
    for (int i = 0; i < C; ++i) {
        int p = i;
        double pv = fabs(n[i][i]);
#pragma unroll
        for (int r = i + 1; r < C; ++r)
            if (fabs(n[r][i]) > pv) { pv = fabs(n[r][i]); p = r; }
#pragma unroll
        for (int r = i + 1; r < C; ++r)
            if (r == p) {
#pragma unroll
                for (int c = 0; c <= C; ++c) {
                    const double t = n[i][c];
                    n[i][c] = n[r][c];
                    n[r][c] = t;
                }
            }
        const double d = fabs(n[i][i]) > 1e-300 ? n[i][i] : 1e-300;
#pragma unroll
        for (int r = i + 1; r < C; ++r) {
            const double f = n[r][i] / d;
#pragma unroll
            for (int c = i; c <= C; ++c) n[r][c] -= f * n[i][c];
        }
    }
#pragma unroll
    for (int i = C - 1; i >= 0; --i) {
        double s = n[i][C];
#pragma unroll
        for (int c = i + 1; c < C; ++c) s -= n[i][c] * x[c];
        x[i] = s / (fabs(n[i][i]) > 1e-300 ? n[i][i] : 1e-300);
    }
}

__device__ inline double det3(const double (&m)[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// rotation of the Horn alignment: the proper rotation closest to U V^T of the SVD of the 3x3
// cross-covariance H = sum (pc)(pw)^T (Kabsch: when det(U V^T) < 0 the direction of the smallest
// singular value flips, which is what completing both bases to right-handed triads does)
__device__ inline void horn_rotation(const double (&H)[9], double (&R)[9]) {
    double hth[3][3], w[3], v[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) hth[i][j] = H[0 * 3 + i] * H[0 * 3 + j] + H[1 * 3 + i] * H[1 * 3 + j] + H[2 * 3 + i] * H[2 * 3 + j];
    eig3_sym(hth, w, v);  // ascending; rows of v = right singular vectors
    double u[3][3];       // u_k = H v_k / sigma_k for the two largest, third by cross product
#pragma unroll
    for (int k = 2; k >= 1; --k) {
        double x[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) x[i] = H[i * 3 + 0] * v[k][0] + H[i * 3 + 1] * v[k][1] + H[i * 3 + 2] * v[k][2];
        const double nn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) u[k][i] = nn > 1e-300 ? x[i] / nn : (i == k ? 1.0 : 0.0);
    }
    // make u1 orthogonal to u2 (guards a tiny sigma_1), u0 = u1 x u2 ; same for v0 = v1 x v2
    const double d12 = u[1][0] * u[2][0] + u[1][1] * u[2][1] + u[1][2] * u[2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) u[1][i] -= d12 * u[2][i];
    const double n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) u[1][i] /= (n1 > 1e-300 ? n1 : 1.0);
    u[0][0] = u[1][1] * u[2][2] - u[1][2] * u[2][1];
    u[0][1] = u[1][2] * u[2][0] - u[1][0] * u[2][2];
    u[0][2] = u[1][0] * u[2][1] - u[1][1] * u[2][0];
    const double v0[3] = {v[1][1] * v[2][2] - v[1][2] * v[2][1], v[1][2] * v[2][0] - v[1][0] * v[2][2], v[1][0] * v[2][1] - v[1][1] * v[2][0]};
    // R = sum_k u_k v_k^T with (u0, v0) completing right-handed triads: det(R) = +1 — the det-corrected U V^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = u[0][i] * v0[j] + u[1][i] * v[1][j] + u[2][i] * v[2][j];
}

// ------------------------------------------------------------------ 12x12 symmetric eigenproblem, cooperative
// One-sided (Hestenes) Jacobi on the columns of the symmetric positive semi-definite G = M^T M, a 16-lane group per
// matrix: lane k < 12 holds column k of G (g) and of the accumulated rotations (v) in registers.  The 66 column pairs
// of a sweep are visited as 11 rounds of 6 disjoint pairs (round-robin tournament); the two lanes of a pair fetch each
// other's columns with ds_bpermute, compute the same rotation bit for bit and apply their half of it.  At convergence
// G V = [g_1 .. g_12] has orthogonal columns: g_k = lambda_k v_k, so |g_k| is the eigenvalue of the lane's vector v.
__device__ inline void hestenes12(double (&g)[12], double (&v)[12], int k) {
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = i == k ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 14; ++sweep) {
        // A column carries absolute rounding noise of ~eps * lambda_max per component from its rotations against the
        // large columns, so g_p . g_q cannot be driven below ~eps * lambda_max * (|g_p| + |g_q|): pairs under that
        // floor are converged (the absolute criterion of the symmetric eigensolvers, LAPACK's included).
        double amax2 = 0.0;
#pragma unroll
        for (int i = 0; i < 12; ++i) amax2 += g[i] * g[i];
#pragma unroll
        for (int o = 1; o < GL; o <<= 1) amax2 = fmax(amax2, __shfl_xor(amax2, o, GL));
        const double floor2 = 2.5e-29 * amax2;                  // (16 eps)^2 * 2 * lambda_max^2
        int rotated = 0;
        double alpha = 0.0;                                     // |g_k|^2, recomputed once per sweep, then updated
#pragma unroll
        for (int i = 0; i < 12; ++i) alpha += g[i] * g[i];
        for (int s = 0; s < 11; ++s) {
            int partner = k;                                    // lanes 12..15 pair with themselves: no-op
            if (k == 11) partner = s;
            else if (k == s) partner = 11;
            else if (k < 11) { partner = 2 * s - k; partner += partner < 0 ? 11 : 0; partner -= partner >= 11 ? 11 : 0; }
            double og[12], ov[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) { og[i] = __shfl(g[i], partner, GL); ov[i] = __shfl(v[i], partner, GL); }
            const double beta = __shfl(alpha, partner, GL);
            double gamma = 0.0;
#pragma unroll
            for (int i = 0; i < 12; ++i) gamma += g[i] * og[i];
            const bool low = k < partner;                        // this lane owns column p (the lower index) of the pair
            const double ap = low ? alpha : beta, aq = low ? beta : alpha;
            if (partner != k && gamma * gamma > floor2 * (ap + aq)) {
                rotated = 1;
                // The rotation must be orthogonal to fp64 accuracy (c^2 + s^2 = 1), its ANGLE need not be exact: tan is
                // evaluated in fp32 (single-instruction rcp / sqrt; an fp64 division or square root costs ~35 instructions,
                // and three of them were half of a Jacobi step).  An angle good to 1e-7 leaves g_p . g_q reduced by that
                // factor instead of annihilated, which the next sweep finishes — the sweep count does not change.
                const float zf = (float)(aq - ap) * __builtin_amdgcn_rcpf((float)(2.0 * gamma));
                const float tf = copysignf(1.0f, zf) * __builtin_amdgcn_rcpf(fabsf(zf) + __builtin_amdgcn_sqrtf(1.0f + zf * zf));
                const double t = (tf == tf && fabsf(tf) <= 1.0f) ? (double)tf : 0.0;
                const double c = rsqrt(1.0 + t * t), sn = c * t;
                const double so = low ? -sn : sn;                // p: c g_p - s g_q ; q: s g_p + c g_q
#pragma unroll
                for (int i = 0; i < 12; ++i) { g[i] = c * g[i] + so * og[i]; v[i] = c * v[i] + so * ov[i]; }
                // |g_p'|^2 = c^2 a_p - 2 c s gamma + s^2 a_q ; |g_q'|^2 = s^2 a_p + 2 c s gamma + c^2 a_q
                alpha = low ? c * c * ap - 2.0 * c * sn * gamma + sn * sn * aq : sn * sn * ap + 2.0 * c * sn * gamma + c * c * aq;
            }
        }
#pragma unroll
        for (int o = 1; o < GL; o <<= 1) rotated |= __shfl_xor(rotated, o, GL);
        if (!rotated) break;
    }
}

// ------------------------------------------------------------------ EPnP
// MODE 0: a 16-lane group solves over `n` sampled points (idx[0..n)); every lane runs the small algebra, lane k owns
//         column k of the 12x12 system.
// MODE 1: the whole workgroup solves over the points with use[i] != 0; every thread runs the small algebra
//         redundantly, point loops are strided over the workgroup and block-reduced through `red`.
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
    double* red;      // MODE 1: LDS scratch [NW][144]
    double* vn;       // MODE 0, PHASE 1 / 2: the hypothesis' four null-space vectors [4][12] in LDS
};

// sum of vals[0..CNT) over the workgroup, result in every thread (MODE 0: every lane already holds the full sums)
template <int MODE, int CNT>
__device__ inline void block_sum(const PointSet<MODE>& ps, double (&vals)[CNT]) {
    if (MODE == 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        double x = vals[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        vals[c] = x;
    }
    __syncthreads();   // (the previous use of `red` has been read by everyone)
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) ps.red[wv * CNT + c] = vals[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) x += ps.red[w * CNT + c];
        vals[c] = x;
    }
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

// returns the mean reprojection error of the chosen solution; R (row-major) and t.
// PHASE 0: the whole solve.  A RANSAC hypothesis (MODE 0) is solved in two phases so that only the 12x12 eigenproblem
// pays for a 16-lane group: PHASE 1 (a group per hypothesis) stops after the eigen-solve and leaves the four null-space
// vectors in LDS; PHASE 2 (ONE THREAD per hypothesis, 64 per wave) repeats the cheap preparation — bit for bit the same
// control points — picks the vectors up and does the scalar algebra (three beta initialisations, Gauss-Newton, Horn
// alignment), which a group would execute sixteen times over.
// dbg (MODE 1 only, may be null): the three beta-branch candidates of this solve, [3][13] = R (9), t (3), mean reprojection
// error (1e300: the branch produced no pose) — what pp_pnp_ransac_debug returns for the refit.
template <int MODE, int PHASE = 0>
__device__ double epnp(const PointSet<MODE>& ps, const Cam& cam, double (&R)[9], double (&t)[3], double* dbg = nullptr) {
    const int gk = threadIdx.x & (GL - 1);   // lane of the solver group
    // ---- control points: centroid + principal directions
    double a4[4] = {0.0, 0.0, 0.0, 0.0};
    PP_FOR_POINTS(ps, i, { a4[0] += ps.p3[3 * i]; a4[1] += ps.p3[3 * i + 1]; a4[2] += ps.p3[3 * i + 2]; a4[3] += 1.0; })
    block_sum<MODE, 4>(ps, a4);
    const double n = a4[3];
    double cws[4][3];
_Pragma("unroll")
    for (int k = 0; k < 3; ++k) cws[0][k] = a4[k] / n;
    double a6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    PP_FOR_POINTS(ps, i, {
        const double x = ps.p3[3 * i] - cws[0][0], y = ps.p3[3 * i + 1] - cws[0][1], z = ps.p3[3 * i + 2] - cws[0][2];
        a6[0] += x * x; a6[1] += x * y; a6[2] += x * z; a6[3] += y * y; a6[4] += y * z; a6[5] += z * z;
    })
    block_sum<MODE, 6>(ps, a6);
    {
        double c3[3][3] = {{a6[0], a6[1], a6[2]}, {a6[1], a6[3], a6[4]}, {a6[2], a6[4], a6[5]}}, w[3], v[3][3];
        eig3_sym(c3, w, v);
        // Solver-independent sign of each principal axis: its component of largest magnitude (lowest index among equals) is
        // positive.  EPnP's estimate depends on which side of the centroid a control point lies (on noisy data the two choices
        // give poses ~1 mm apart at equal reprojection error), and the sign a Jacobi sequence leaves is an accident of its
        // rotations — with a fixed convention this kernel and a LAPACK-based EPnP (oracle/pnp.py) agree to solver accuracy.
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a0 = fabs(v[k][0]), a1 = fabs(v[k][1]), a2 = fabs(v[k][2]);
            const double lead = (a0 >= a1 && a0 >= a2) ? v[k][0] : (a1 >= a2 ? v[k][1] : v[k][2]);
            if (lead < 0) { v[k][0] = -v[k][0]; v[k][1] = -v[k][1]; v[k][2] = -v[k][2]; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // largest first, as the SVD ordering of OpenCV's EPnP
            const double kk = sqrt(fmax(w[2 - k], 0.0) / n);
#pragma unroll
            for (int c = 0; c < 3; ++c) cws[k + 1][c] = cws[0][c] + kk * v[2 - k][c];
        }
    }
    // ---- barycentric coordinates: alpha_{1..3} = CC^-1 (p - c0)
    double cc[9], cci[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) cc[r * 3 + c] = cws[c + 1][r] - cws[0][r];
    {
        const double d = det3(cc);
        const double id = fabs(d) > 1e-300 ? 1.0 / d : 0.0;
        cci[0] = (cc[4] * cc[8] - cc[5] * cc[7]) * id; cci[1] = (cc[2] * cc[7] - cc[1] * cc[8]) * id; cci[2] = (cc[1] * cc[5] - cc[2] * cc[4]) * id;
        cci[3] = (cc[5] * cc[6] - cc[3] * cc[8]) * id; cci[4] = (cc[0] * cc[8] - cc[2] * cc[6]) * id; cci[5] = (cc[2] * cc[3] - cc[0] * cc[5]) * id;
        cci[6] = (cc[3] * cc[7] - cc[4] * cc[6]) * id; cci[7] = (cc[1] * cc[6] - cc[0] * cc[7]) * id; cci[8] = (cc[0] * cc[4] - cc[1] * cc[3]) * id;
    }
    auto alphas = [&](int i, double (&a)[4]) {
        const double x = ps.p3[3 * i] - cws[0][0], y = ps.p3[3 * i + 1] - cws[0][1], z = ps.p3[3 * i + 2] - cws[0][2];
        a[1] = cci[0] * x + cci[1] * y + cci[2] * z;
        a[2] = cci[3] * x + cci[4] * y + cci[5] * z;
        a[3] = cci[6] * x + cci[7] * y + cci[8] * z;
        a[0] = 1.0 - a[1] - a[2] - a[3];
    };
    // ---- M^T M (12 x 12): lane gk of a solver group accumulates column gk.  MODE 1: group g takes the points
    // g, g + NG, ..., then the columns are summed over the groups (wave shuffles, then `red`)
    double vn[4][12];
    if (PHASE != 2) {
    double g12[12], v12[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) g12[r] = 0.0;
    {
        const int kj = gk / 3, kc = gk - 3 * kj;   // column gk = (control point kj, coordinate kc); kj == 4 for the idle lanes
        auto add_point = [&](int i) {
            double a[4];
            alphas(i, a);
            const double du = cam.uc - ps.p2[2 * i], dv = cam.vc - ps.p2[2 * i + 1];
            const double ak = kj == 0 ? a[0] : kj == 1 ? a[1] : kj == 2 ? a[2] : kj == 3 ? a[3] : 0.0;
            // row 1 of the point: (a_j fu, 0, a_j du), row 2: (0, a_j fv, a_j dv)
            const double r1k = kc == 0 ? ak * cam.fu : kc == 2 ? ak * du : 0.0;
            const double r2k = kc == 1 ? ak * cam.fv : kc == 2 ? ak * dv : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g12[3 * j] += (a[j] * cam.fu) * r1k;
                g12[3 * j + 1] += (a[j] * cam.fv) * r2k;
                g12[3 * j + 2] += (a[j] * du) * r1k + (a[j] * dv) * r2k;
            }
        };
        if (MODE == 0) {
            for (int ii = 0; ii < ps.n; ++ii) add_point(ps.idx[ii]);
        } else {
            for (int i = threadIdx.x / GL; i < ps.n; i += NG)
                if (ps.use[i]) add_point(i);
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
            for (int r = 0; r < 12; ++r) { g12[r] += __shfl_xor(g12[r], 16); g12[r] += __shfl_xor(g12[r], 32); }
            __syncthreads();
            if (lane < 12) {
#pragma unroll
                for (int r = 0; r < 12; ++r) ps.red[(wv * 12 + lane) * 12 + r] = g12[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                double x = 0.0;
                if (gk < 12)
                    for (int w = 0; w < NW; ++w) x += ps.red[(w * 12 + gk) * 12 + r];
                g12[r] = x;
            }
        }
    }
    hestenes12(g12, v12, gk);
    // the four eigenvectors of the smallest eigenvalues, ascending, into every lane: vn[i][0..12)
    {
        double lam = 0.0;
#pragma unroll
        for (int r = 0; r < 12; ++r) lam += g12[r] * g12[r];   // |g_k|^2 = lambda_k^2: same order
        if (gk >= 12) lam = 1e300;
        int rank = 0;
#pragma unroll
        for (int o = 0; o < 12; ++o) {
            const double lo = __shfl(lam, o, GL);
            rank += (lo < lam || (lo == lam && o < gk)) ? 1 : 0;
        }
        int src[4] = {0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < 12; ++o) {
            const int ro = __shfl(rank, o, GL);
#pragma unroll
            for (int q = 0; q < 4; ++q) src[q] = ro == q ? o : src[q];
        }
        if (PHASE == 1) {          // hand the vectors over through LDS: the lane ranked q < 4 owns vector q
            if (gk < 12 && rank < 4) {
#pragma unroll
                for (int r = 0; r < 12; ++r) ps.vn[rank * 12 + r] = v12[r];
            }
            return 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 12; ++r) vn[q][r] = __shfl(v12[r], src[q], GL);
    }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 12; ++r) vn[q][r] = ps.vn[q * 12 + r];
    }
    // ---- L (6 x 10) and rho
    double L[6][10], rho[6];
    {
        constexpr int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
        double dv[4][6][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int c = 0; c < 3; ++c) dv[i][p][c] = vn[i][3 * pa[p] + c] - vn[i][3 * pb[p] + c];
#define PP_DOT(i, j, p) (dv[i][p][0] * dv[j][p][0] + dv[i][p][1] * dv[j][p][1] + dv[i][p][2] * dv[j][p][2])
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            L[p][0] = PP_DOT(0, 0, p); L[p][1] = 2 * PP_DOT(0, 1, p); L[p][2] = PP_DOT(1, 1, p); L[p][3] = 2 * PP_DOT(0, 2, p);
            L[p][4] = 2 * PP_DOT(1, 2, p); L[p][5] = PP_DOT(2, 2, p); L[p][6] = 2 * PP_DOT(0, 3, p); L[p][7] = 2 * PP_DOT(1, 3, p);
            L[p][8] = 2 * PP_DOT(2, 3, p); L[p][9] = PP_DOT(3, 3, p);
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) s += (cws[pa[p]][c] - cws[pb[p]][c]) * (cws[pa[p]][c] - cws[pb[p]][c]);
            rho[p] = s;
        }
#undef PP_DOT
    }
    // ---- three beta initialisations, Gauss-Newton, pose, keep the least reprojection error
    double best_err = 1e300;
    if (MODE == 1 && dbg && threadIdx.x == 0) {
        for (int k = 0; k < 39; ++k) dbg[k] = k % 13 == 12 ? 1e300 : 0.0;
    }
    for (int approx = 0; approx < 3; ++approx) {
        double b[4] = {0, 0, 0, 0};
        if (approx == 0) {  // betas10 columns B11 B12 B13 B14
            double A4[6][4], x[4];
#pragma unroll
            for (int p = 0; p < 6; ++p) { A4[p][0] = L[p][0]; A4[p][1] = L[p][1]; A4[p][2] = L[p][3]; A4[p][3] = L[p][6]; }
            lstsq6<4>(A4, rho, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = -x[1] / b[0]; b[2] = -x[2] / b[0]; b[3] = -x[3] / b[0]; }
            else { b[0] = sqrt(x[0]); b[1] = x[1] / b[0]; b[2] = x[2] / b[0]; b[3] = x[3] / b[0]; }
        } else if (approx == 1) {  // B11 B12 B22
            double A3[6][3], x[3];
#pragma unroll
            for (int p = 0; p < 6; ++p) { A3[p][0] = L[p][0]; A3[p][1] = L[p][1]; A3[p][2] = L[p][2]; }
            lstsq6<3>(A3, rho, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) b[0] = -b[0];
        } else {  // B11 B12 B22 B13 B23
            double A5[6][5], x[5];
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int c = 0; c < 5; ++c) A5[p][c] = L[p][c];
            lstsq6<5>(A5, rho, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) b[0] = -b[0];
            b[2] = fabs(b[0]) > 1e-300 ? x[3] / b[0] : 0.0;
        }
        // (MODE 1: b is the same in every thread, so this branch is uniform and the block sums below stay aligned)
        if (!(b[0] == b[0]) || !(b[1] == b[1]) || !(b[2] == b[2]) || !(b[3] == b[3])) continue;
        for (int it = 0; it < 5; ++it) {  // Gauss-Newton on the 6 distance constraints
            double A[6][4], rb[6], dx[4];
#pragma unroll
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
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] += dx[k];
        }
        // control points in the camera frame, sign from the depth of the points
        double ccs[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) ccs[j][c] = b[0] * vn[0][3 * j + c] + b[1] * vn[1][3 * j + c] + b[2] * vn[2][3 * j + c] + b[3] * vn[3][3 * j + c];
        double s6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        PP_FOR_POINTS(ps, i, {
            double a[4];
            alphas(i, a);
_Pragma("unroll")
            for (int c = 0; c < 3; ++c) s6[c] += a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c];
            s6[3] += ps.p3[3 * i]; s6[4] += ps.p3[3 * i + 1]; s6[5] += ps.p3[3 * i + 2];
        })
        block_sum<MODE, 6>(ps, s6);
        if (s6[2] < 0) {  // solve_for_sign (mean depth must be positive)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) ccs[j][c] = -ccs[j][c];
#pragma unroll
            for (int c = 0; c < 3; ++c) s6[c] = -s6[c];
        }
        double pc0[3], pw0[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pc0[c] = s6[c] / n; pw0[c] = s6[3 + c] / n; }
        double H[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        PP_FOR_POINTS(ps, i, {
            double a[4], pc[3];
            alphas(i, a);
_Pragma("unroll")
            for (int c = 0; c < 3; ++c) pc[c] = a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c] - pc0[c];
            const double w0 = ps.p3[3 * i] - pw0[0], w1 = ps.p3[3 * i + 1] - pw0[1], w2 = ps.p3[3 * i + 2] - pw0[2];
_Pragma("unroll")
            for (int r = 0; r < 3; ++r) { H[r * 3] += pc[r] * w0; H[r * 3 + 1] += pc[r] * w1; H[r * 3 + 2] += pc[r] * w2; }
        })
        block_sum<MODE, 9>(ps, H);
        double Rc[9], tc[3];
        horn_rotation(H, Rc);
#pragma unroll
        for (int r = 0; r < 3; ++r) tc[r] = pc0[r] - (Rc[r * 3] * pw0[0] + Rc[r * 3 + 1] * pw0[1] + Rc[r * 3 + 2] * pw0[2]);
        double er[1] = {0.0};
        PP_FOR_POINTS(ps, i, {
            const double X = ps.p3[3 * i], Y = ps.p3[3 * i + 1], Z = ps.p3[3 * i + 2];
            const double xc = Rc[0] * X + Rc[1] * Y + Rc[2] * Z + tc[0], yc = Rc[3] * X + Rc[4] * Y + Rc[5] * Z + tc[1];
            const double zc = Rc[6] * X + Rc[7] * Y + Rc[8] * Z + tc[2];
            const double iz = 1.0 / zc;
            const double du = cam.uc + cam.fu * xc * iz - ps.p2[2 * i], dvv = cam.vc + cam.fv * yc * iz - ps.p2[2 * i + 1];
            er[0] += sqrt(du * du + dvv * dvv);
        })
        block_sum<MODE, 1>(ps, er);
        const double err = er[0] / n;
        if (MODE == 1 && dbg && threadIdx.x == 0) {
            for (int k = 0; k < 9; ++k) dbg[approx * 13 + k] = Rc[k];
            for (int k = 0; k < 3; ++k) dbg[approx * 13 + 9 + k] = tc[k];
            dbg[approx * 13 + 12] = err == err ? err : 1e300;
        }
        if (err == err && err < best_err) {
            best_err = err;
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = Rc[k];
#pragma unroll
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
                                                        int32_t* __restrict__ ok, int32_t* __restrict__ npts,
                                                        double* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* p3 = (float*)smem;                      // [MAXP][3]
    float* p2 = p3 + 3 * MAXP;                     // [MAXP][2]
    unsigned char* use = (unsigned char*)(p2 + 2 * MAXP);  // [MAXP]
    double* vnb = (double*)(use + MAXP);           // [HB][4][12] null-space vectors of a batch of hypotheses
    double* red = vnb;                             // [NW][144] (the refit's reduction scratch, after the hypotheses)
    double* hyp = vnb + HB * 48;                   // [MAXH][12]  R, t of every hypothesis
    int* cnt = (int*)(hyp + MAXH * 12);            // [MAXH]
    __shared__ int wsum[NW], base, best_h, best_c;
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
    if (tid < MAXH) cnt[tid] = 0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += NT) {
        const int n = n0 + tid;
        int64_t tx = -1, ty = -1, sx = -1, sy = -1;
        if (n < N) { tx = tp[2 * n]; ty = tp[2 * n + 1]; sx = sp[2 * n]; sy = sp[2 * n + 1]; }
        const bool v = tx != -1 && ty != -1 && sx != -1 && sy != -1;
        const unsigned long long bal = __ballot(v);
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int off = base, tot = 0;
        for (int i = 0; i < NW; ++i) { off += i < wv ? wsum[i] : 0; tot += wsum[i]; }
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
        if (tid == 0) base += tot;
        __syncthreads();
    }
    const int np = base < MAXP ? base : MAXP;
    if (tid == 0) npts[prob] = np;
    auto fail = [&]() {
        if (tid == 0) {
            if (dbg)
                for (int k = 0; k < 40; ++k) dbg[(size_t)prob * 40 + k] = k == 39 ? -1.0 : k % 13 == 12 ? 1e300 : 0.0;
            for (int k = 0; k < 9; ++k) rot[(size_t)prob * 9 + k] = (k % 4 == 0) ? 1.0 : 0.0;
            tvec[(size_t)prob * 3] = 0.0; tvec[(size_t)prob * 3 + 1] = 0.0; tvec[(size_t)prob * 3 + 2] = 1.0;
            ratio[prob] = 0.0;
            ok[prob] = 0;
        }
    };
    if (np < SAMPLE) { fail(); return; }

    // ---- RANSAC hypotheses on 5-point samples, HB at a time: phase 1 — solver group g (16 lanes) runs the 12x12
    // eigen-solve of hypotheses g, g + NG, ... and leaves their null-space vectors in LDS; phase 2 — thread h finishes
    // hypothesis h (64 hypotheses per wave instead of 4)
    const int nh = iters < MAXH ? iters : MAXH;
    auto sample = [&](int h, int (&idx)[SAMPLE]) __attribute__((always_inline)) {
        unsigned s = hash32(0x9E3779B9u * (unsigned)(prob + 1) ^ (unsigned)(h * 7919 + 17));
#pragma unroll
        for (int k = 0; k < SAMPLE; ++k) {
            for (;;) {
                s = hash32(s + 0x6D2B79F5u);
                const int c = (int)(s % (unsigned)np);
                bool dup = false;
#pragma unroll
                for (int j = 0; j < SAMPLE; ++j) dup |= j < k && idx[j] == c;
                if (!dup) { idx[k] = c; break; }
            }
        }
    };
    for (int hb = 0; hb < nh; hb += HB) {
        const int hend = hb + HB < nh ? hb + HB : nh;
        for (int h0 = hb; h0 < hend; h0 += NG) {
            const int h = h0 + tid / GL;
            if (h < hend) {      // (uniform over the 16 lanes of a group; the shuffles inside stay within the group)
                int idx[SAMPLE];
                sample(h, idx);
                PointSet<0> ps = {p3, p2, idx, nullptr, SAMPLE, nullptr, vnb + (h - hb) * 48};
                double R[9], t[3];
                (void)epnp<0, 1>(ps, cam, R, t);
            }
        }
        __syncthreads();
        if (tid < HB && hb + tid < hend) {
            const int h = hb + tid;
            int idx[SAMPLE];
            sample(h, idx);
            PointSet<0> ps = {p3, p2, idx, nullptr, SAMPLE, nullptr, vnb + tid * 48};
            double R[9], t[3];
            const double e = epnp<0, 2>(ps, cam, R, t);
#pragma unroll
            for (int k = 0; k < 9; ++k) hyp[h * 12 + k] = e < 1e299 ? R[k] : 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) hyp[h * 12 + 9 + k] = e < 1e299 ? t[k] : 0.0;
        }
        __syncthreads();
    }
    // ---- score every hypothesis on every point (squared reprojection error <= thresh^2): a thread keeps its (up to 8)
    // points in registers and walks the hypotheses (the 12 doubles of a model are an LDS broadcast); counts are summed
    // per wave and added to cnt[h] with one LDS atomic per wave — no barrier inside the loop
    const double th2 = (double)thresh * (double)thresh;
    {
        constexpr int PPT = MAXP / NT;
        double X[PPT], Y[PPT], Z[PPT], U[PPT], V[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int i = tid + q * NT;
            const bool in = i < np;
            X[q] = in ? p3[3 * i] : 0.0; Y[q] = in ? p3[3 * i + 1] : 0.0; Z[q] = in ? p3[3 * i + 2] : 0.0;
            U[q] = in ? p2[2 * i] : 1e30; V[q] = in ? p2[2 * i + 1] : 1e30;      // a padding slot is never an inlier
        }
        for (int h = 0; h < nh; ++h) {
            const double* M = hyp + h * 12;
            const double m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5], m6 = M[6], m7 = M[7], m8 = M[8];
            const double t0 = M[9], t1 = M[10], t2 = M[11];
            int c = 0;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                if (q * NT >= np) break;   // uniform
                const double zc = m6 * X[q] + m7 * Y[q] + m8 * Z[q] + t2;
                const double iz = 1.0 / zc;
                const double du = cam.uc + cam.fu * (m0 * X[q] + m1 * Y[q] + m2 * Z[q] + t0) * iz - U[q];
                const double dv = cam.vc + cam.fv * (m3 * X[q] + m4 * Y[q] + m5 * Z[q] + t1) * iz - V[q];
                c += (du * du + dv * dv <= th2) ? 1 : 0;  // NaN (degenerate hypothesis) is never an inlier
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if (lane == 0 && c) atomicAdd(&cnt[h], c);
        }
    }
    __syncthreads();
    if (tid < 64) {   // arg-max of the consensus, lowest hypothesis index among equals
        int bc = -1, bh = 0;
        for (int h = tid; h < nh; h += 64)
            if (cnt[h] > bc) { bc = cnt[h]; bh = h; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int oc = __shfl_xor(bc, o), oh = __shfl_xor(bh, o);
            if (oc > bc || (oc == bc && oh < bh)) { bc = oc; bh = oh; }
        }
        if (tid == 0) { best_h = bh; best_c = bc; }
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
    PointSet<1> ps = {p3, p2, nullptr, use, np, red, nullptr};
    double R[9], t[3];
    const double e = epnp<1>(ps, cam, R, t, dbg ? dbg + (size_t)prob * 40 : nullptr);
    if (tid == 0) {
        const bool good = e < 1e299;
        if (dbg) {   // the branch the refit kept: the first one of least error (-1: none, the RANSAC winner is returned)
            double* d = dbg + (size_t)prob * 40;
            d[39] = !good ? -1.0 : d[12] == e ? 0.0 : d[25] == e ? 1.0 : 2.0;
        }
        const double* M = hyp + best_h * 12;
        for (int k = 0; k < 9; ++k) rot[(size_t)prob * 9 + k] = good ? R[k] : M[k];
        for (int k = 0; k < 3; ++k) tvec[(size_t)prob * 3 + k] = good ? t[k] : M[9 + k];
        ratio[prob] = (double)best_c / (double)np;
        ok[prob] = 1;
    }
}

}  // namespace

static int pnp_launch(const float* tar_pts_2d, const float* src_pts_3d, const float* K, const float* tem_pose,
                      const int64_t* tar_pts, const int64_t* src_pts, int P, int H, int W, int N, int iterations,
                      float reproj_threshold, double* rot, double* tvec, double* inlier_ratio, int32_t* success,
                      int32_t* num_points, double* dbg, void* stream) {
    if (!tar_pts_2d || !src_pts_3d || !K || !tem_pose || !tar_pts || !src_pts || !rot || !tvec || !inlier_ratio ||
        !success || !num_points)
        return PP_EINVAL;
    if (P <= 0 || H <= 0 || W <= 0 || N <= 0 || N > MAXP || iterations <= 0 || reproj_threshold <= 0.f) return PP_EINVAL;
    static_assert(HB * 48 >= NW * 144, "the refit's reduction scratch aliases the null-space vector buffer");
    const size_t smem = (size_t)MAXP * (3 + 2) * sizeof(float) + MAXP + (size_t)HB * 48 * sizeof(double) +
                        (size_t)MAXH * 12 * sizeof(double) + MAXH * sizeof(int);
    static bool attr_set[PP_MAX_DEVICES];   // the dynamic-LDS opt-in is per device
    if (!attr_set[pp_cur_device()]) {
        PP_CHECK_HIP(hipFuncSetAttribute((const void*)pnp_ransac_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set[pp_cur_device()] = true;
    }
    hipLaunchKernelGGL(pnp_ransac_kernel, dim3(P), dim3(NT), smem, (hipStream_t)stream, tar_pts_2d, src_pts_3d, K,
                       tem_pose, tar_pts, src_pts, H, W, N, iterations, reproj_threshold, rot, tvec, inlier_ratio,
                       success, num_points, dbg);
    return pp_last_launch();
}

extern "C" {

int pp_pnp_ransac(const float* tar_pts_2d, const float* src_pts_3d, const float* K, const float* tem_pose,
                  const int64_t* tar_pts, const int64_t* src_pts, int P, int H, int W, int N, int iterations,
                  float reproj_threshold, double* rot, double* tvec, double* inlier_ratio, int32_t* success,
                  int32_t* num_points, void* stream) {
    return pnp_launch(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, P, H, W, N, iterations, reproj_threshold, rot, tvec,
                      inlier_ratio, success, num_points, nullptr, stream);
}

int pp_pnp_ransac_debug(const float* tar_pts_2d, const float* src_pts_3d, const float* K, const float* tem_pose,
                        const int64_t* tar_pts, const int64_t* src_pts, int P, int H, int W, int N, int iterations,
                        float reproj_threshold, double* rot, double* tvec, double* inlier_ratio, int32_t* success,
                        int32_t* num_points, double* refit_branches, void* stream) {
    if (!refit_branches) return PP_EINVAL;
    return pnp_launch(tar_pts_2d, src_pts_3d, K, tem_pose, tar_pts, src_pts, P, H, W, N, iterations, reproj_threshold, rot, tvec,
                      inlier_ratio, success, num_points, refit_branches, stream);
}

}  // extern "C"
