// Per-lane Kabsch rotation (fp32 guess + fp64 polish) and the feature map + its derivatives.
// One lane = one frame; the frame's coordinates sit in LDS (load_x_tile layout).
#pragma once
#include "cvf_common.hpp"

// ------------------------------------------------------------------------------------
// Optimal rotation for the 3x3 covariance H (row-vector convention x_al = (x-c) R,
// R maximises tr(R^T H) over proper rotations):  R = U diag(1,1,sign det(U V^T)) V^T.
// Also returns Kinv = (tr(P) I - P)^-1, P = sym(R^T H): the 3x3 solve of the analytic
// derivative dR = R [Kinv ax(R^T dH)]x used by the VJP/JVP kernels.
// ------------------------------------------------------------------------------------
// fp64 reciprocal from the hardware seed (v_rcp_f64: ~2^-26 relative) + Newton steps to full double precision.  The
// IEEE division sequence the compiler emits otherwise (scale, fixup, denormal handling) is 3-4x longer.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
// ------------------------------------------------------------------------------------
// The solve, in two precisions.  (1) fp32: cyclic Jacobi on H^T H (three sweeps, fixed count: no divergence between
// lanes), the two dominant eigenvectors v1, v2, u_i = H v_i orthonormalised, R0 = [u1 u2 u1xu2][v1 v2 v1xv2]^T (the cross
// products realise the det fix) - robust for any H, exact to ~1e-7 .. 1e-3 depending on the conditioning (fp32, and
// H^T H squares the condition number).  (2) fp64: R0 is re-orthonormalised (one Newton-Schulz step) and polished by two
// Newton steps ON THE ROTATION GROUP for  max_R tr(R^T H):  with P = R^T H, K = tr(sym P) I - sym P, the maximiser near R
// is R exp([omega]x), omega = K^-1 ax(P), ax(P) = (P21 - P12, P02 - P20, P10 - P01) - the same 3x3 solve as the analytic
// derivative dR = R [K^-1 ax(R^T dH)]x the VJP / JVP kernels use, so the second step's K^-1 IS the Kinv they need.  The
// iteration converges quadratically: |R - R*| <= 1.5e-12 and Kinv to 2e-12 relative from R0 errors up to 1e-3 (20 000
// random / near-planar / planar / ill-conditioned H against an fp64 SVD, tests/test_kabsch_host.py).  Against the all-fp64
// Jacobi (four sweeps) this replaces: ~300 fp64 + ~430 fp32 instructions instead of ~830 fp64 ones, and the dependent
// chain of fp64 reciprocal square roots is gone - the solve is the serial part of every alignment kernel.
// Degenerate H (second and third singular value equal with det < 0, or rank <= 1): the optimum is not unique, K is
// singular, Kinv = 0 is returned and R0 (re-orthonormalised) stands, as before.
// ------------------------------------------------------------------------------------
// (the solver is __host__ __device__: tests/test_kabsch_host.py runs this very source in hipcc's host pass on the CPU against
//  an fp64 SVD; only the reciprocal / reciprocal-square-root seeds differ between the passes)
#define CVF_HD __host__ __device__ __forceinline__
CVF_HD float cvf_rsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
CVF_HD float cvf_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
CVF_HD double cvf_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return fast_rcp(x);
#else
  return 1.0 / x;
#endif
}
CVF_HD double cvf_rsq(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rsq(x);   // v_rsq_f64 seed (~2^-26) + two Newton steps
  double e = fma(-x * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-x * r, r, 1.0);
  return fma(0.5 * r, e, r);
#else
  return 1.0 / sqrt(x);
#endif
}

template <class T> struct KabschTiny;
template <> struct KabschTiny<float> {   // (A is H^T H of an H scaled to max |H_ij| = 1: entries O(1); these squared are still normal floats)
  static constexpr float rot = 1e-18f, norm = 1e-36f;
};
template <> struct KabschTiny<double> {
  static constexpr double rot = 1e-150, norm = 1e-300;
};

// One Jacobi rotation annihilating A[P][Q].  With d = aqq - app, b = 2 apq, r = sqrt(d^2 + b^2):
//   cos(2 phi) = |d| / r   ->   c^2 = (1 + |d|/r) / 2,   s = sgn(d) b / (2 r c),   t = s / c
// evaluated with two reciprocal square roots (1/r, 1/c) and no division.
template <int P, int Q, class T>
CVF_HD void jacobi_rot(T (&A)[3][3], T (&V)[3][3]) {
  const T apq = A[P][Q];
  const T app = A[P][P], aqq = A[Q][Q];
  const T d = aqq - app, b = T(2) * apq;
  // |d| + tiny keeps r > 0: d = b = 0 gives c = 1, s = 0 (the identity) without a guard on every quantity
  const T ad = (d < T(0) ? -d : d) + KabschTiny<T>::rot;
  const T inv_r = cvf_rsq(ad * ad + b * b);
  const T c2 = T(0.5) * ad * inv_r + T(0.5);   // in [1/2, 1]
  const T inv_c = cvf_rsq(c2);
  const T c = c2 * inv_c;
  const T s = (d < T(0) ? T(-0.5) : T(0.5)) * b * inv_r * inv_c;
  const T t = s * inv_c;
  constexpr int R = 3 - P - Q;  // the remaining index
  A[P][P] = app - t * apq;
  A[Q][Q] = aqq + t * apq;
  A[P][Q] = A[Q][P] = T(0);
  const T arp = A[R][P], arq = A[R][Q];
  A[R][P] = A[P][R] = c * arp - s * arq;
  A[R][Q] = A[Q][R] = s * arp + c * arq;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T vip = V[i][P], viq = V[i][Q];
    V[i][P] = c * vip - s * viq;
    V[i][Q] = s * vip + c * viq;
  }
}

struct KabschOut {
  float R[9];     // row-major
  float Kinv[6];  // 00 01 02 11 12 22
};

// R0 (row-major 3x3) from H in precision T with SWEEPS Jacobi sweeps.  Returns false when the eigenvalues of H^T H leave the
// third eigenvector's separation from the other two below what precision T resolves (fp32: a nearly collinear align set
// or nearly equal second and third singular values) - the caller then repeats the stage in fp64.
template <class T, int SWEEPS>
CVF_HD bool kabsch_guess(const double (&H)[3][3], T (&R0)[3][3]) {
  T h[3][3];
  T m = T(0);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      h[i][j] = (T)H[i][j];
      const T ah = h[i][j] < T(0) ? -h[i][j] : h[i][j];
      m = ah > m ? ah : m;
    }
  const T sc = cvf_rcp(m > T(1e-30) ? m : T(1e-30));   // scale invariance: the rotation of H is the rotation of H / max |H_ij|
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) h[i][j] *= sc;
  T A[3][3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) A[i][j] = A[j][i] = h[0][i] * h[0][j] + h[1][i] * h[1][j] + h[2][i] * h[2][j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? T(1) : T(0);
#pragma unroll 1
  for (int sweep = 0; sweep < SWEEPS; ++sweep) {
    jacobi_rot<0, 1>(A, V);
    jacobi_rot<0, 2>(A, V);
    jacobi_rot<1, 2>(A, V);
  }
  // v1, v2: eigenvectors of the largest and the second eigenvalue.  Written as selects on three flags (index
  // arithmetic compiles to a tree of exec-mask branches).
  const T l0 = A[0][0], l1 = A[1][1], l2 = A[2][2];
  const bool min0 = l0 <= l1 && l0 <= l2;   // column 0 belongs to the smallest eigenvalue
  const bool min1 = !min0 && l1 <= l2;      // column 1 does
  const bool min01 = min0 || min1;
  const T la = min0 ? l1 : l0, lb = min01 ? l2 : l1;   // the two that remain: columns (1,2), (0,2) or (0,1)
  const T lmin = min0 ? l0 : (min1 ? l1 : l2);
  const bool a_first = la >= lb;
  T v1[3], v2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T ca = min0 ? V[i][1] : V[i][0], cb = min01 ? V[i][2] : V[i][1];
    v1[i] = a_first ? ca : cb;
    v2[i] = a_first ? cb : ca;
  }
  T u1[3], u2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    u1[i] = h[i][0] * v1[0] + h[i][1] * v1[1] + h[i][2] * v1[2];
    u2[i] = h[i][0] * v2[0] + h[i][1] * v2[1] + h[i][2] * v2[2];
  }
  T n1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
  n1 = cvf_rsq(n1 + KabschTiny<T>::norm);   // u1 = 0 stays 0 (no branch)
#pragma unroll
  for (int i = 0; i < 3; ++i) u1[i] *= n1;
  const T pr = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) u2[i] -= pr * u1[i];
  T n2 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
  n2 = cvf_rsq(n2 + KabschTiny<T>::norm);
#pragma unroll
  for (int i = 0; i < 3; ++i) u2[i] *= n2;
  const T u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
  const T v3_[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R0[i][j] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3_[j];
  const T lmid = a_first ? lb : la, lmax = a_first ? la : lb;
  return lmid - lmin > T(3e-6) * lmax;
}

// P = R^T H, K = tr(sym P) I - sym P (symmetric, by cofactors) -> Kinv (6 entries, 0 when K is singular) and
// omega = Kinv ax(P)
CVF_HD void kabsch_newton_terms(const double (&R)[3][3], const double (&H)[3][3], double (&Kinv)[6], double (&om)[3]) {
  double Pm[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Pm[i][j] = R[0][i] * H[0][j] + R[1][i] * H[1][j] + R[2][i] * H[2][j];
  const double p01 = 0.5 * (Pm[0][1] + Pm[1][0]), p02 = 0.5 * (Pm[0][2] + Pm[2][0]), p12 = 0.5 * (Pm[1][2] + Pm[2][1]);
  const double tr = Pm[0][0] + Pm[1][1] + Pm[2][2];
  const double k00 = tr - Pm[0][0], k11 = tr - Pm[1][1], k22 = tr - Pm[2][2];
  const double k01 = -p01, k02 = -p02, k12 = -p12;
  const double c00 = k11 * k22 - k12 * k12, c01 = k02 * k12 - k01 * k22, c02 = k01 * k12 - k02 * k11;
  const double c11 = k00 * k22 - k02 * k02, c12 = k01 * k02 - k00 * k12, c22 = k00 * k11 - k01 * k01;
  double det = k00 * c00 + k01 * c01 + k02 * c02;
  // singular relative to the size of K (tr^3 bounds |det|): the rotation is not unique there (see the header comment)
  const bool regular = fabs(det) > 1e-13 * fabs(tr * tr * tr) && fabs(det) > 1e-300;
  det = cvf_rcp(regular ? det : 1.0);
  det = regular ? det : 0.0;
  Kinv[0] = c00 * det;
  Kinv[1] = c01 * det;
  Kinv[2] = c02 * det;
  Kinv[3] = c11 * det;
  Kinv[4] = c12 * det;
  Kinv[5] = c22 * det;
  const double a0 = Pm[2][1] - Pm[1][2], a1 = Pm[0][2] - Pm[2][0], a2 = Pm[1][0] - Pm[0][1];
  const double x = Kinv[0] * a0 + Kinv[1] * a1 + Kinv[2] * a2;
  const double y = Kinv[1] * a0 + Kinv[3] * a1 + Kinv[4] * a2;
  const double z = Kinv[2] * a0 + Kinv[4] * a1 + Kinv[5] * a2;
  // a step of more than ~0.5 rad is no Newton step: K is nearly singular (the rotation nearly non-unique) - keep R
  const bool small = x * x + y * y + z * z < 0.25;
  om[0] = small ? x : 0.0;
  om[1] = small ? y : 0.0;
  om[2] = small ? z : 0.0;
}

// KINV = false: the caller wants the rotation only (features without the derivative's K^-1): ONE Newton step, second order in
// omega (|R - R*| <= 2e-11 from the fp32 guess, below the fp32 rounding of the stored rotation); out.Kinv is not written.
template <bool KINV = true>
CVF_HD void kabsch_from_H(const double (&H)[3][3], KabschOut& out) {
  double r0[3][3];
  {
    float g[3][3];
    const bool resolved = kabsch_guess<float, 3>(H, g);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) r0[i][j] = (double)g[i][j];
    // the rare frames the fp32 stage cannot resolve: the same stage in fp64, four sweeps (off-diagonal / diagonal <= 3e-18
    // over random, rank-deficient and degenerate H).  Lane-divergent and skipped by waves none of whose frames needs it;
    // inlined - a call would put the whole kernel under the function-call ABI (stack, registers saved around the call).
    if (__builtin_expect(!resolved, 0)) (void)kabsch_guess<double, 4>(H, r0);
  }
  double R[3][3];
  {  // Newton-Schulz: R <- R0 (3 I - R0^T R0) / 2  (an fp32 R0 is orthogonal to ~1e-7; the Newton steps below assume a rotation)
    double Y[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) {
        const double x = r0[0][i] * r0[0][j] + r0[1][i] * r0[1][j] + r0[2][i] * r0[2][j];
        Y[i][j] = Y[j][i] = (i == j ? 1.5 : 0.0) - 0.5 * x;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[i][j] = r0[i][0] * Y[0][j] + r0[i][1] * Y[1][j] + r0[i][2] * Y[2][j];
  }
  double Kinv[6], om[3];
  {  // first Newton step, second order in omega:  R <- R (I + W + W^2 / 2),  W = [omega]x
    kabsch_newton_terms(R, H, Kinv, om);
    const double x = om[0], y = om[1], z = om[2];
    const double M[3][3] = {{1.0 - 0.5 * (y * y + z * z), -z + 0.5 * x * y, y + 0.5 * x * z},
                            {z + 0.5 * x * y, 1.0 - 0.5 * (x * x + z * z), -x + 0.5 * y * z},
                            {-y + 0.5 * x * z, x + 0.5 * y * z, 1.0 - 0.5 * (x * x + y * y)}};
    double Rn[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Rn[i][j] = R[i][0] * M[0][j] + R[i][1] * M[1][j] + R[i][2] * M[2][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[i][j] = Rn[i][j];
  }
  if constexpr (KINV) {  // second step (omega ~ 1e-6 .. 1e-12 now: first order), whose K^-1 is the one the derivative kernels use
    kabsch_newton_terms(R, H, Kinv, om);
    const double x = om[0], y = om[1], z = om[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double r0 = R[i][0], r1 = R[i][1], r2 = R[i][2];
      R[i][0] = r0 + (r1 * z - r2 * y);
      R[i][1] = r1 + (r2 * x - r0 * z);
      R[i][2] = r2 + (r0 * y - r1 * x);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out.R[3 * i + j] = (float)R[i][j];
  if constexpr (KINV) {
#pragma unroll
    for (int i = 0; i < 6; ++i) out.Kinv[i] = (float)Kinv[i];
  }
}

// ------------------------------------------------------------------------------------
// Invariant features on raw coordinates.  `at(a)` returns atom a of this lane's frame.
// grad_* return the gradient vectors w.r.t. each atom of the *scalar(s)* the feature
// emits; for the dihedral in (cos,sin) mode the two outputs share dphi: d cos = -sin dphi,
// d sin = cos dphi.
// ------------------------------------------------------------------------------------
struct BondG {
  float val;
  V3 ga, gb;
};
__device__ __forceinline__ BondG bond_eval(V3 xa, V3 xb) {
  V3 r = xb - xa;
  float d = sqrtf(dot(r, r));
  float inv = 1.0f / d;
  BondG o;
  o.val = d;
  o.gb = inv * r;
  o.ga = (-inv) * r;
  return o;
}

struct AngleG {
  float cs;  // cos of the angle at b
  V3 ga, gb, gc;  // gradient of cos
};
__device__ __forceinline__ AngleG angle_eval(V3 xa, V3 xb, V3 xc) {
  V3 r1 = xa - xb, r2 = xc - xb;
  float l1 = sqrtf(dot(r1, r1)), l2 = sqrtf(dot(r2, r2));
  float inv12 = 1.0f / (l1 * l2);
  float cs = dot(r1, r2) * inv12;
  AngleG o;
  o.cs = cs;
  o.ga = inv12 * r2 - (cs / (l1 * l1)) * r1;
  o.gc = inv12 * r1 - (cs / (l2 * l2)) * r2;
  o.gb = (-1.0f) * (o.ga + o.gc);
  return o;
}

struct DihedralG {
  float cs, sn;
  V3 g1, g2, g3, g4;  // gradient of phi
  float p, q;         // g2 = (-1 - p) g1 + q g4,  g3 = p g1 + (-1 - q) g4
};
__device__ __forceinline__ DihedralG dihedral_eval(V3 x1, V3 x2, V3 x3, V3 x4) {
  V3 b1 = x2 - x1, b2 = x3 - x2, b3 = x4 - x3;
  V3 n1 = cross(b1, b2), n2 = cross(b2, b3);
  float n1sq = dot(n1, n1), n2sq = dot(n2, n2), b2sq = dot(b2, b2);
  float l2 = sqrtf(b2sq);
  float inv = 1.0f / sqrtf(n1sq * n2sq);
  DihedralG o;
  o.cs = dot(n1, n2) * inv;
  o.sn = dot(n1, b3) * l2 * inv;
  o.g1 = (-l2 / n1sq) * n1;
  o.g4 = (l2 / n2sq) * n2;
  float p = dot(b1, b2) / b2sq, q = dot(b3, b2) / b2sq;
  o.g2 = (-1.0f - p) * o.g1 + q * o.g4;
  o.g3 = p * o.g1 + (-1.0f - q) * o.g4;
  o.p = p;
  o.q = q;
  return o;
}
