// Per-lane Kabsch rotation in fp64 and the feature map + its derivatives.
// One lane = one frame; the frame's coordinates sit in LDS (load_x_tile layout).
#pragma once
#include "cvf_common.hpp"

// ------------------------------------------------------------------------------------
// Optimal rotation for the 3x3 covariance H (row-vector convention x_al = (x-c) R,
// R maximises tr(R^T H) over proper rotations):  R = U diag(1,1,sign det(U V^T)) V^T.
// Computed as  R = [u1 u2 u1xu2] [v1 v2 v1xv2]^T  with v1,v2 the two dominant
// eigenvectors of H^T H (cyclic Jacobi, fixed sweep count -> no divergence between
// lanes) and u_i = H v_i normalised; the cross products realise the det fix.
// Also returns Kinv = (tr(P) I - P)^-1, P = sym(R^T H): the 3x3 solve of the analytic
// derivative dR = R [Kinv ax(R^T dH)]x used by the VJP/JVP kernels.
// ------------------------------------------------------------------------------------
// fp64 reciprocal / reciprocal square root / square root from the hardware seeds (v_rcp_f64, v_rsq_f64:
// ~2^-26 relative) + Newton steps to full double precision.  The IEEE division / sqrt sequences the compiler
// emits otherwise (scale, fixup, denormal handling) are 3-4x longer, and this eigen-solve is the serial part of
// a kernel that runs one wave per SIMD.  Inputs here are sums of squares of O(1..1e4) numbers: no denormals.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  // r <- r + r*(1 - x r^2)/2, twice
  double e = fma(-x * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-x * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  return r;
}
__device__ __forceinline__ double fast_sqrt(double x) { return x > 0.0 ? x * fast_rsqrt(x) : 0.0; }

// v_rsq_f64 seed + one Newton step (relative error ~1e-15: enough for a Jacobi rotation, which the next sweep corrects)
__device__ __forceinline__ double rsqrt_1step(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * r, r, 1.0);
  return fma(0.5 * r, e, r);
}

// One Jacobi rotation annihilating A[P][Q].  With d = aqq - app, b = 2 apq, r = sqrt(d^2 + b^2):
//   cos(2 phi) = |d| / r   ->   c^2 = (1 + |d|/r) / 2,   s = sgn(d) b / (2 r c),   t = s / c
// evaluated with two reciprocal square roots (1/r, 1/c) and no division: the usual
// t = sgn(d) b / (|d| + r), c = 1/sqrt(1 + t^2) needs a square root, a reciprocal and a reciprocal square root,
// and these dependent fp64 sequences are what the solve's time consists of.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot(double (&A)[3][3], double (&V)[3][3]) {
  const double apq = A[P][Q];
  const double app = A[P][P], aqq = A[Q][Q];
  const double d = aqq - app, b = 2.0 * apq;
  // |d| + tiny keeps r > 0: d = b = 0 gives c = 1, s = 0 (the identity) without a guard on every quantity
  const double ad = fabs(d) + 1e-150;
  const double inv_r = rsqrt_1step(fma(ad, ad, b * b));
  const double c2 = fma(0.5 * ad, inv_r, 0.5);   // in [1/2, 1]
  const double inv_c = rsqrt_1step(c2);
  const double c = c2 * inv_c;
  const double s = (d < 0.0 ? -0.5 : 0.5) * b * inv_r * inv_c;
  const double t = s * inv_c;
  constexpr int R = 3 - P - Q;  // the remaining index
  A[P][P] = app - t * apq;
  A[Q][Q] = aqq + t * apq;
  A[P][Q] = A[Q][P] = 0.0;
  const double arp = A[R][P], arq = A[R][Q];
  A[R][P] = A[P][R] = c * arp - s * arq;
  A[R][Q] = A[Q][R] = s * arp + c * arq;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double vip = V[i][P], viq = V[i][Q];
    V[i][P] = c * vip - s * viq;
    V[i][Q] = s * vip + c * viq;
  }
}

struct KabschOut {
  float R[9];     // row-major
  float Kinv[6];  // 00 01 02 11 12 22
};

__device__ __forceinline__ void kabsch_from_H(const double (&H)[3][3], KabschOut& out) {
  double A[3][3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
#pragma unroll 1
  // 4 sweeps: off-diagonal / diagonal <= 3e-18 over random, rank-deficient and degenerate H (7e-7 after 3)
  for (int sweep = 0; sweep < 4; ++sweep) {
    jacobi_rot<0, 1>(A, V);
    jacobi_rot<0, 2>(A, V);
    jacobi_rot<1, 2>(A, V);
  }
  // v1, v2: eigenvectors of the largest and the second eigenvalue.  Written as selects on three flags (the
  // index arithmetic this replaces compiled to a tree of exec-mask branches, a third of the solve's instructions).
  const double l0 = A[0][0], l1 = A[1][1], l2 = A[2][2];
  const bool min0 = l0 <= l1 && l0 <= l2;   // column 0 belongs to the smallest eigenvalue
  const bool min1 = !min0 && l1 <= l2;      // column 1 does
  const bool min01 = min0 || min1;
  const double la = min0 ? l1 : l0, lb = min01 ? l2 : l1;   // the two that remain: columns (1,2), (0,2) or (0,1)
  const bool a_first = la >= lb;
  double v1[3], v2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double ca = min0 ? V[i][1] : V[i][0], cb = min01 ? V[i][2] : V[i][1];
    v1[i] = a_first ? ca : cb;
    v2[i] = a_first ? cb : ca;
  }
  double u1[3], u2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    u1[i] = H[i][0] * v1[0] + H[i][1] * v1[1] + H[i][2] * v1[2];
    u2[i] = H[i][0] * v2[0] + H[i][1] * v2[1] + H[i][2] * v2[2];
  }
  double n1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
  n1 = fast_rsqrt(n1 + 1e-300);   // u1 = 0 stays 0 (no branch around the Newton steps)
#pragma unroll
  for (int i = 0; i < 3; ++i) u1[i] *= n1;
  const double pr = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) u2[i] -= pr * u1[i];
  double n2 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
  n2 = fast_rsqrt(n2 + 1e-300);
#pragma unroll
  for (int i = 0; i < 3; ++i) u2[i] *= n2;
  const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
  const double v3_[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
  double R[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i][j] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3_[j];
  // P = R^T H (symmetric at the optimum), K = tr(P) I - P
  double Pm[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Pm[i][j] = R[0][i] * H[0][j] + R[1][i] * H[1][j] + R[2][i] * H[2][j];
  const double p01 = 0.5 * (Pm[0][1] + Pm[1][0]), p02 = 0.5 * (Pm[0][2] + Pm[2][0]), p12 = 0.5 * (Pm[1][2] + Pm[2][1]);
  const double tr = Pm[0][0] + Pm[1][1] + Pm[2][2];
  const double k00 = tr - Pm[0][0], k11 = tr - Pm[1][1], k22 = tr - Pm[2][2];
  const double k01 = -p01, k02 = -p02, k12 = -p12;
  // inverse of the symmetric K by cofactors
  const double c00 = k11 * k22 - k12 * k12, c01 = k02 * k12 - k01 * k22, c02 = k01 * k12 - k02 * k11;
  const double c11 = k00 * k22 - k02 * k02, c12 = k01 * k02 - k00 * k12, c22 = k00 * k11 - k01 * k01;
  double det = k00 * c00 + k01 * c01 + k02 * c02;
  const bool regular = fabs(det) > 1e-300;
  det = fast_rcp(regular ? det : 1.0);
  det = regular ? det : 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out.R[3 * i + j] = (float)R[i][j];
  out.Kinv[0] = (float)(c00 * det);
  out.Kinv[1] = (float)(c01 * det);
  out.Kinv[2] = (float)(c02 * det);
  out.Kinv[3] = (float)(c11 * det);
  out.Kinv[4] = (float)(c12 * det);
  out.Kinv[5] = (float)(c22 * det);
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
