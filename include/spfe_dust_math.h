/*
 * spfe_dust_math.h — the arithmetic of the direct "dust" alignment, shared by the
 * GPU kernel (sp_orb_slam_amd/csrc/dust.hip) and the CPU oracle (oracle/spfe_oracle.c) so that both
 * evaluate the same sequence of IEEE operations (compile with -ffp-contract=off).
 *
 * What it restates:
 *   g2o::EdgeSE3ProjectDustOnlyPose::computeError / linearizeOplus / getPixelValue / isInImage
 *       /root/reference/orb_slam2/src/optimization/types_dust_tracking.cpp:37-140
 *   Optimizer::PoseOptimizationDust (edge set-up, intrinsics / 8, Huber delta 0.9, 40 iterations, inlier rule)
 *       /root/reference/orb_slam2/src/mapping/optimizer_dust.cpp:170-294
 *   and, of g2o (a catkin dependency of the reference, NOT vendored in /root/reference — parity unpinned,
 *   published algorithm restated): SE3Quat (exp, operator*, map, normalizeRotation), VertexSE3Expmap::oplusImpl,
 *   RobustKernelHuber::robustify, BaseUnaryEdge::constructQuadraticForm, OptimizationAlgorithmLevenberg::solve
 *   (tau = 1e-5, good-step scale in [1/3, 2/3], ni doubling, 10 trials after failure), LinearSolverDense.
 */
#ifndef SPFE_DUST_MATH_H
#define SPFE_DUST_MATH_H

#include <math.h>

#if defined(__HIPCC__)
#define SPFE_DM __host__ __device__ static inline
#else
#define SPFE_DM static inline
#endif

typedef struct {
  double q[4]; /* unit quaternion x, y, z, w (SE3Quat::_r) */
  double t[3]; /* translation (SE3Quat::_t) */
} spfe_se3;

/* if (w < 0) coeffs *= -1; normalize()  — SE3Quat::normalizeRotation */
SPFE_DM void spfe_se3_normalize(spfe_se3 *T) {
  if (T->q[3] < 0) { T->q[0] = -T->q[0]; T->q[1] = -T->q[1]; T->q[2] = -T->q[2]; T->q[3] = -T->q[3]; }
  const double n2 = T->q[0] * T->q[0] + T->q[1] * T->q[1] + T->q[2] * T->q[2] + T->q[3] * T->q[3];
  const double n = sqrt(n2);
  T->q[0] /= n; T->q[1] /= n; T->q[2] /= n; T->q[3] /= n;
}

/* Eigen::Quaterniond(Matrix3d): the trace / largest-diagonal branches of Eigen's quaternionbase_assign_impl */
SPFE_DM void spfe_quat_from_rot(const double R[9], double q[4]) {
  double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    double s = sqrt(tr + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[7] - R[5]) * s;
    q[1] = (R[2] - R[6]) * s;
    q[2] = (R[3] - R[1]) * s;
  } else {
    /* i = index of the largest diagonal element, j = (i + 1) % 3, k = (j + 1) % 3 — the three cases written out (static
     * indices: the kernel keeps R and q in registers) */
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > (i == 1 ? R[4] : R[0])) i = 2;
#define SPFE_QFR_CASE(I, J, K)                                                   \
  {                                                                              \
    double s = sqrt(R[I * 3 + I] - R[J * 3 + J] - R[K * 3 + K] + 1.0);           \
    q[I] = 0.5 * s;                                                              \
    s = 0.5 / s;                                                                 \
    q[3] = (R[K * 3 + J] - R[J * 3 + K]) * s;                                    \
    q[J] = (R[J * 3 + I] + R[I * 3 + J]) * s;                                    \
    q[K] = (R[K * 3 + I] + R[I * 3 + K]) * s;                                    \
  }
    if (i == 0) SPFE_QFR_CASE(0, 1, 2)
    else if (i == 1) SPFE_QFR_CASE(1, 2, 0)
    else SPFE_QFR_CASE(2, 0, 1)
#undef SPFE_QFR_CASE
  }
}

/* Eigen::Quaterniond::toRotationMatrix */
SPFE_DM void spfe_quat_to_rot(const double q[4], double R[9]) {
  const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* Eigen: Quaternion * Vector3 = v + w * (2 q x v) + q x (2 q x v); SE3Quat::map = _r * xyz + _t */
SPFE_DM void spfe_se3_map(const spfe_se3 *T, const double p[3], double out[3]) {
  const double *q = T->q;
  const double ux = 2.0 * (q[1] * p[2] - q[2] * p[1]);
  const double uy = 2.0 * (q[2] * p[0] - q[0] * p[2]);
  const double uz = 2.0 * (q[0] * p[1] - q[1] * p[0]);
  out[0] = (p[0] + q[3] * ux + (q[1] * uz - q[2] * uy)) + T->t[0];
  out[1] = (p[1] + q[3] * uy + (q[2] * ux - q[0] * uz)) + T->t[1];
  out[2] = (p[2] + q[3] * uz + (q[0] * uy - q[1] * ux)) + T->t[2];
}

/* SE3Quat::exp(update) * T  (VertexSE3Expmap::oplusImpl); update = (omega[3], upsilon[3]) */
SPFE_DM void spfe_se3_oplus(spfe_se3 *T, const double upd[6]) {
  const double wx = upd[0], wy = upd[1], wz = upd[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  /* Omega = skew(omega), Omega2 = Omega * Omega */
  const double O[9] = {0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0};
  double O2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
  double a, b, c1, c2;
  if (theta < 0.00001) {
    a = 1.0; b = 0.5; c1 = 0.5; c2 = 1.0 / 6.0;
  } else {
    const double st = sin(theta), ct = cos(theta);
    a = st / theta;
    b = (1.0 - ct) / (theta * theta);
    c1 = b;
    c2 = (theta - st) / (theta * theta * theta);
  }
  double R[9], V[9];
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + a * O[i] + b * O2[i];
    V[i] = I + c1 * O[i] + c2 * O2[i];
  }
  spfe_se3 E;
  spfe_quat_from_rot(R, E.q);
  for (int r = 0; r < 3; ++r) E.t[r] = V[r * 3] * upd[3] + V[r * 3 + 1] * upd[4] + V[r * 3 + 2] * upd[5];
  /* the SE3Quat(Quaternion, t) constructor normalises */
  spfe_se3_normalize(&E);
  /* result = E * T: t = E.t + E.r * T.t; r = E.r * T.r; normalizeRotation() */
  spfe_se3 Er = E;
  Er.t[0] = Er.t[1] = Er.t[2] = 0.0;
  double rt[3];
  spfe_se3_map(&Er, T->t, rt);
  spfe_se3 N;
  N.t[0] = E.t[0] + rt[0]; N.t[1] = E.t[1] + rt[1]; N.t[2] = E.t[2] + rt[2];
  const double *a4 = E.q, *b4 = T->q; /* Eigen quaternion product a * b */
  N.q[3] = a4[3] * b4[3] - a4[0] * b4[0] - a4[1] * b4[1] - a4[2] * b4[2];
  N.q[0] = a4[3] * b4[0] + a4[0] * b4[3] + a4[1] * b4[2] - a4[2] * b4[1];
  N.q[1] = a4[3] * b4[1] + a4[1] * b4[3] + a4[2] * b4[0] - a4[0] * b4[2];
  N.q[2] = a4[3] * b4[2] + a4[2] * b4[3] + a4[0] * b4[1] - a4[1] * b4[0];
  spfe_se3_normalize(&N);
  *T = N;
}

/* Converter::toSE3Quat(cv::Mat CV_32F 4x4): R, t from floats, SE3Quat(R, t) normalises (converter.cpp:36-46) */
SPFE_DM void spfe_se3_from_f32(const float Tcw[16], spfe_se3 *T) {
  double R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = (double)Tcw[r * 4 + c];
  spfe_quat_from_rot(R, T->q);
  T->t[0] = (double)Tcw[3]; T->t[1] = (double)Tcw[7]; T->t[2] = (double)Tcw[11];
  spfe_se3_normalize(T);
}
/* Converter::toCvMat(SE3Quat): to_homogeneous_matrix() cast to float (converter.cpp:48-67) */
SPFE_DM void spfe_se3_to_f32(const spfe_se3 *T, float Tcw[16]) {
  double R[9];
  spfe_quat_to_rot(T->q, R);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Tcw[r * 4 + c] = (float)R[r * 3 + c];
    Tcw[r * 4 + 3] = (float)T->t[r];
  }
  Tcw[12] = Tcw[13] = Tcw[14] = 0.0f;
  Tcw[15] = 1.0f;
}

/* EdgeSE3ProjectDustOnlyPose::isInImage (types_dust_tracking.cpp:37-42), border = 1; w, h are floats there */
SPFE_DM int spfe_dust_in_image(double u, double v, float w, float h) {
  const double border = 1.0;
  return (u >= border && u + border + 1 < w && v >= border && v + border + 1 < h);
}

/* getPixelValue (:44-58): float bilinear lookup in the hc x wc dust map */
SPFE_DM float spfe_dust_pixel(const float *dust, int wc, float x, float y) {
  const int x_f = (int)floorf(x);
  const int y_f = (int)floorf(y);
  const float xx = x - x_f;
  const float yy = y - y_f;
  return (float)((1 - xx) * (1 - yy) * dust[y_f * wc + x_f] + xx * (1 - yy) * dust[y_f * wc + x_f + 1] +
                 (1 - xx) * yy * dust[(y_f + 1) * wc + x_f] + xx * yy * dust[(y_f + 1) * wc + x_f + 1]);
}

/* One edge.  level is sticky: computeError sets it to 1 and nothing ever resets it (:72-76, :85-87). */
typedef struct {
  double err;   /* _error(0,0) */
  float u, v;   /* u_, v_: last in-image projection */
  int level;
} spfe_dust_edge;

/* computeError (:64-98) */
SPFE_DM void spfe_dust_error(const spfe_se3 *T, const double Xw[3], double fx, double fy, double cx, double cy,
                             const float *dust, int wc, int hc, spfe_dust_edge *e) {
  double xl[3];
  spfe_se3_map(T, Xw, xl);
  if (xl[2] < 0.0) { e->err = 0.0; e->level = 1; return; }
  const double x = xl[0] * fx / xl[2] + cx;
  const double y = xl[1] * fy / xl[2] + cy;
  if (!spfe_dust_in_image(x, y, (float)wc, (float)hc)) {
    e->err = 0.0;
    e->level = 1;
  } else {
    e->err = (double)spfe_dust_pixel(dust, wc, (float)x, (float)y);
    e->u = (float)x;
    e->v = (float)y;
  }
}

/* linearizeOplus (:100-140): J[6]; returns 0 where the reference would throw " should be omitted" */
SPFE_DM int spfe_dust_jacobian(const spfe_se3 *T, const double Xw[3], double fx, double fy, double cx, double cy,
                               const float *dust, int wc, int hc, int level, double J[6]) {
  if (level == 1) { for (int k = 0; k < 6; ++k) J[k] = 0.0; return 1; }
  double p[3];
  spfe_se3_map(T, Xw, p);
  const double x = p[0], y = p[1];
  const double invz = 1.0 / p[2];
  const double invz_2 = invz * invz;
  const double u = x * fx * invz + cx;
  const double v = y * fy * invz + cy;
  if (!spfe_dust_in_image(u, v, (float)wc, (float)hc)) { for (int k = 0; k < 6; ++k) J[k] = 0.0; return 0; }
  double Ju[6], Jv[6];
  Ju[0] = -x * y * invz_2 * fx;
  Ju[1] = (1 + (x * x * invz_2)) * fx;
  Ju[2] = -y * invz * fx;
  Ju[3] = invz * fx;
  Ju[4] = 0;
  Ju[5] = -x * invz_2 * fx;
  Jv[0] = -(1 + y * y * invz_2) * fy;
  Jv[1] = x * y * invz_2 * fy;
  Jv[2] = x * invz * fy;
  Jv[3] = 0;
  Jv[4] = invz * fy;
  Jv[5] = -y * invz_2 * fy;
  const double gu = (double)((spfe_dust_pixel(dust, wc, (float)(u + 1), (float)v) - spfe_dust_pixel(dust, wc, (float)(u - 1), (float)v)) / 2.0f);
  const double gv = (double)((spfe_dust_pixel(dust, wc, (float)u, (float)(v + 1)) - spfe_dust_pixel(dust, wc, (float)u, (float)(v - 1))) / 2.0f);
  for (int k = 0; k < 6; ++k) J[k] = gu * Ju[k] + gv * Jv[k];   /* (1x2) * (2x6) */
  return 1;
}

/* RobustKernelHuber::robustify(e2 = chi2, rho[3]) */
SPFE_DM void spfe_huber(double e2, double delta, double rho[3]) {
  const double dsqr = delta * delta;
  if (e2 <= dsqr) {
    rho[0] = e2; rho[1] = 1.0; rho[2] = 0.0;
  } else {
    const double sqrte = sqrt(e2);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e2;
  }
}

/* ---- the sums over the edges: a FIXED-SHAPE tree, the same in the kernel and in the oracle ---------------------------
 * g2o accumulates chi2 and the pose block's normal equations edge by edge (activeRobustChi2, constructQuadraticForm); its
 * bits are not pinned by anything in /root/reference (g2o is not vendored).  Rounds 1-4 restated that as a strict
 * edge-order chain, which a GPU can only run one lane per sum (27 chains of n dependent additions: 32 us per Levenberg
 * iteration, slower than one host core).  The contract is now a tree a 256-thread workgroup evaluates in log time:
 *   slot t (0 <= t < 256) = 0.0 + term(t) + term(t + 256) + ...            (ascending edge index, edges i < n only)
 *   wave g (0 <= g < 4)   = halving tree over its 64 slots: for m = 32, 16, 8, 4, 2, 1: s[l] += s[l + m], l < m
 *   total                 = ((wave 0 + wave 1) + wave 2) + wave 3
 * (the kernel's cross-lane butterfly forms s[l] + s[l ^ m] in every lane: IEEE addition commutes, so lane 0's value — and
 * every other lane's — is the halving tree's).  What is summed: SPFE_DUST_NSUM = 28 quantities per edge,
 *   q[0]                     rho0 = Huber(err^2)                              (activeRobustChi2)
 *   q[1 + i (i + 1) / 2 + j] (J[i] * rho1) * J[j], 0 <= j <= i < 6            (lower triangle of J^T (rho1 Omega) J, Omega = 1)
 *   q[22 + j]                -((rho1 * J[j]) * err)                            (b -= J^T rho1 Omega e)
 * The solver reads the lower triangle only (spfe_solve6); spfe_dust_unpack mirrors it. */
#define SPFE_DUST_SLOTS 256
#define SPFE_DUST_NSUM 28

SPFE_DM void spfe_dust_terms(double err, const double J[6], double delta, double q[SPFE_DUST_NSUM]) {
  double rho[3];
  spfe_huber(err * err, delta, rho);
  q[0] = rho[0];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) q[1 + i * (i + 1) / 2 + j] = (J[i] * rho[1]) * J[j];
  for (int j = 0; j < 6; ++j) q[22 + j] = -((rho[1] * J[j]) * err);
}

SPFE_DM void spfe_dust_unpack(const double tot[SPFE_DUST_NSUM], double *chi, double H[36], double b[6]) {
  *chi = tot[0];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) H[i * 6 + j] = H[j * 6 + i] = tot[1 + i * (i + 1) / 2 + j];
  for (int j = 0; j < 6; ++j) b[j] = tot[22 + j];
}

/* the tree over one quantity's 256 slot sums (host form; s is destroyed) */
static inline double spfe_dust_tree_total(double s[SPFE_DUST_SLOTS]) {
  for (int g = 0; g < 4; ++g)
    for (int m = 32; m >= 1; m >>= 1)
      for (int l = 0; l < m; ++l) s[64 * g + l] += s[64 * g + l + m];
  return ((s[0] + s[64]) + s[128]) + s[192];
}

/* (H + lambda I) x = b for the 6x6 pose block (LinearSolverDense: Eigen LDLT, solved when isPositive()).  Restated as an
 * unpivoted L D L^T with unit lower-triangular L — no square roots, ONE division per pivot (its reciprocal; everything else
 * multiplies by it), no early exit: the kernel runs this on the critical path of every trial step, and a branch per pivot
 * would stop the independent columns from overlapping.  Positive = every pivot D_j > 0 (what a Cholesky factorisation
 * checks too).  H is a sum of rho1 * J^T J (positive semi-definite) and lambda >= 0, so this and Eigen's pivoted LDLT
 * disagree on one input only: the ZERO matrix (no gradient anywhere and lambda = tau * 0), which Eigen calls positive and
 * solves to x = 0 — one trial with rho = 0, Terminate — and this form rejects — ten failed trials, Terminate.  Either way the
 * pose is untouched, the edges are re-evaluated at it and optimize() reports one iteration (tests/golden/dust_flat.npz
 * pins that).  H: symmetric 6x6, the lower triangle is read.  Returns 1 / 0; x = 0 when 0 (the caller's gain ratio reads x).
 *   U_ij = H_ij - sum_{k<j} L_ik U_jk (j <= i; U_jj = D_j, + lambda on the diagonal), L_ij = U_ij * (1 / D_j)
 *   z = L^-1 b (forward), w = z / D (as z * (1 / D)), x = L^-T w (backward); sums in ascending k. */
SPFE_DM int spfe_solve6(const double H[36], double lambda, const double b[6], double x[6]) {
  double L[36], U[36], rD[6];
  int ok = 1;
  for (int j = 0; j < 6; ++j) {
    for (int i = j; i < 6; ++i) {
      double s = H[i * 6 + j] + (i == j ? lambda : 0.0);
      for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * U[j * 6 + k];
      U[i * 6 + j] = s;
    }
    ok &= U[j * 6 + j] > 0.0;
    rD[j] = 1.0 / U[j * 6 + j];
    for (int i = j + 1; i < 6; ++i) L[i * 6 + j] = U[i * 6 + j] * rD[j];
  }
  double z[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * z[k];
    z[i] = s;
  }
  for (int i = 5; i >= 0; --i) {
    double s = z[i] * rD[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
    x[i] = s;
  }
  if (!ok)
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
  return ok;
}

/* Levenberg state + one trial's bookkeeping (OptimizationAlgorithmLevenberg::solve) */
typedef struct {
  double lambda, ni;
} spfe_lm;

#define SPFE_LM_TAU 1e-5
#define SPFE_LM_GOOD_LO (1. / 3.)
#define SPFE_LM_GOOD_HI (2. / 3.)
#define SPFE_LM_MAX_TRIALS 10

/* after a trial: rho from the chi2 pair; returns 1 if the step is accepted (lambda / ni updated either way) */
SPFE_DM int spfe_lm_judge(spfe_lm *lm, double currentChi, double tempChi, const double x[6], const double b[6],
                          double *rho_out) {
  double rho = currentChi - tempChi;
  double scale = 0;
  for (int j = 0; j < 6; ++j) scale += x[j] * (lm->lambda * x[j] + b[j]);
  scale += 1e-3;
  rho /= scale;
  *rho_out = rho;
  if (rho > 0 && isfinite(tempChi)) {
    const double d = 2 * rho - 1;
    double alpha = 1. - d * d * d;
    alpha = alpha < SPFE_LM_GOOD_HI ? alpha : SPFE_LM_GOOD_HI;
    const double sf = SPFE_LM_GOOD_LO > alpha ? SPFE_LM_GOOD_LO : alpha;
    lm->lambda *= sf;
    lm->ni = 2;
    return 1;
  }
  lm->lambda *= lm->ni;
  lm->ni *= 2;
  return 0;
}

#endif /* SPFE_DUST_MATH_H */
