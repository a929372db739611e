// HMM kernels over staged emission records: batched forward likelihood (K5, like()), forward-backward
// posterior decoding (K6, decodek2()/decode()) and the Viterbi path (K7).  Replaces the per-call recursions of
// src/relateHMM.cpp:1426-1544, 1012-1263, 1289-1420.  gfx950 only; FP64 VALU work, no MFMA (a 3x3 transition
// with per-lane parameters is not a dense contraction).
//
// Numerics.  The reference walks the SNPs in log space (3 exp + 6 log per step).  Here the recursion runs in the
// linear domain and is renormalised by exact powers of two every 16 steps; the exponent is accumulated as an
// integer and one log is taken at the end.  The two exponentials per step (e^{-a t}, e^{-a k1 t}; a is per lane,
// t per SNP) are inherent; they enter only through expm1, because the transition matrix is affine in
// (e1 - 1, e2 - 1) with NO constant term off the diagonal:
//     T01 = T21 = u = -k1 m1          m1 = expm1(-a t), m2 = expm1(-a k1 t)
//     T02 = k2 w,  T20 = k0 w         w  = (m2 - k1 m1)/(k1 - 1)
//     T10 = u k0/k1,  T12 = u k2/k1,  diagonal = 1 - row sum
// which is the reference's T (src/relateHMM.cpp:1452-1489) with its constant terms cancelled analytically
// (k0 + k1 + k2 = 1); the reference's own rounding noise in T02/T20 (which its fix_underflow clamps below 1e-15,
// :542-547) does not arise.  For |a t| < 2^-7 (the normal case at SNP-chip density) u and w are two short
// polynomials in -a t with per-lane coefficients (LaneC); a forward step is then 13 + 14 FP64 instructions per
// vector.  like() is reproduced to <= 1e-13 relative (tests/test_gpu_parity.py).
//
// Records (kernels_emis.hpp).  S_ibd1/S_ibd0, S_ibd2/S_ibd0 (16 B per pair and SNP) as two planes per group of 64
// pairs, two consecutive SNPs per 16-byte element: two wave loads bring two records of every lane.  The recursions need
// emissions only up to a per-SNP factor: S_ibd0 is taken out at staging time, its product per pair (mantissa sc_m,
// exponent sc_e) enters the log-likelihood once at the end and cancels in posteriors.  The distance of a step is the
// SHARED tsh[s] = pos[s] - pos[s-1] (scalar loads): a SNP dropped for the pair is (-1, 1), a step with emission
// (1, 1, 1) -- T(t1) T(t2) = T(t1 + t2), so the steps over dropped SNPs compose to the reference's one step between
// kept SNPs, no lane branches on missing data, and k T = k lets dropped SNPs precede the first kept one.  The sign bit
// of r1 is the "dropped" flag for the kernels that must know (posterior = -1); the sign bit of r2 marks the rare record
// that carries its own distance (a run of dropped SNPs across a chromosome break, kernels_emis.hpp): a group of records
// with such a lane goes through the general-distance code with per-lane t (rec_lane).
//
// Parallelism.  One lane = one (pair, parameter point).  When a launch has too few lanes to fill 256 CUs the SNP
// axis is cut into C chunks: the per-step matrices A_s = T(t_s) diag(S_s) compose associatively,
// chunk_matrix_kernel forms every chunk's product (three row vectors with their own power-of-two exponents) and a
// short combine kernel folds the C products per lane.  The exponentials and T are shared by the three rows, so a
// chunked step costs ~2x a plain one.  A workgroup is 256 lanes of ONE chunk, so loop bounds are scalar.
// Between the two, a launch of up to one wave per SIMD runs two-sided (like_halves_kernel): forward half . backward half.
//
// Memory.  Every lane streams its column of records, 4 records (64 B) per group, double-buffered in registers; the row
// base is wave-uniform (scalar), the lane adds its 32-bit column offset.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels_emis.hpp"

namespace relate {

static constexpr int kGroup = 4;     // records per load group (two groups in flight)

struct Rec {
  double t, r1, r2;  // distance of the step (shared, wave-uniform on the hot path), S_ibd1 / S_ibd0, S_ibd2 / S_ibd0
};
// One pair's column of the staged block: wave-uniform base (the SNP loops are uniform, so the row address is scalar)
// plus the lane's byte offset.
struct RecCol {
  const double *__restrict__ base;
  const double *__restrict__ tsh;  // shared distances, M rounded up to even + 2 entries
  size_t row;    // stride in doubles of a row of two SNPs (2 planes x padded pairs x 2)
  unsigned off;  // byte offset of the pair's column inside a row
  int hmax;      // last valid row (prefetches beyond the data are clamped to it)
  bool general;  // some lane of this wave sits on a pair with records that carry their own distance (own[col] != 0)
  __device__ RecCol(const double *rec, const double *tshared, const int *own, int RP, int col, int M)
      : base(rec), tsh(tshared), row((size_t)2 * kRecPlanes * RP), off(8u * (unsigned)rec_off(0, col, kRecPlanes, RP)),
        hmax((M - 1) >> 1), general(__builtin_amdgcn_ballot_w64(own[col] != 0) != 0) {}
  // records of SNPs 2h and 2h + 1 (h is wave-uniform: the distances are scalar loads)
  __device__ inline void at2(int h, Rec &a, Rec &b) const {
    const char *q = (const char *)(base + (size_t)h * row) + off;
    const double2 r1 = *(const double2 *)q, r2 = *(const double2 *)(q + 1024);
    const double2 t = *(const double2 *)(tsh + 2 * (size_t)h);
    a.t = t.x; a.r1 = r1.x; a.r2 = r2.x;
    b.t = t.y; b.r1 = r1.y; b.r2 = r2.y;
  }
  __device__ inline Rec at(int s) const {  // one record (tails)
    const char *q = (const char *)(base + (size_t)(s >> 1) * row) + off + 8 * (s & 1);
    Rec r;
    r.t = tsh[s];
    r.r1 = *(const double *)q;
    r.r2 = *(const double *)(q + 1024);
    return r;
  }
  // kGroup records from SNP s (a multiple of 2); rows beyond the data are clamped (their values are never used)
  __device__ inline void group(int s, Rec (&R)[4]) const {
    const int h = s >> 1;
    at2(min(h, hmax), R[0], R[1]);
    at2(min(h + 1, hmax), R[2], R[3]);
  }
};
static constexpr double kLn2 = 0.6931471805599453094;

// ---- expm1 for x <= 0 ------------------------------------------------------------------------------------
__device__ __forceinline__ double em_tiny(double x, double xx) {  // |x| < 2^-7: x + x^2 (1/2 + ... + x^4/720), rel. err < 6e-17
  double p = __builtin_fma(x, 1.0 / 720, 1.0 / 120);
  p = __builtin_fma(x, p, 1.0 / 24);
  p = __builtin_fma(x, p, 1.0 / 6);
  p = __builtin_fma(x, p, 0.5);
  return __builtin_fma(xx, p, x);
}
__device__ __forceinline__ double em_tiny5(double x, double xx) {  // |x| < 2^-8: terms through x^5/120, abs. err < 5e-18
  double p = __builtin_fma(x, 1.0 / 120, 1.0 / 24);
  p = __builtin_fma(x, p, 1.0 / 6);
  p = __builtin_fma(x, p, 0.5);
  return __builtin_fma(xx, p, x);
}
__device__ __forceinline__ double em_small(double x) {  // |x| < 2^-4: Taylor through x^10/10!
  double p = 1.0 / 3628800;
  p = __builtin_fma(x, p, 1.0 / 362880);
  p = __builtin_fma(x, p, 1.0 / 40320);
  p = __builtin_fma(x, p, 1.0 / 5040);
  p = __builtin_fma(x, p, 1.0 / 720);
  p = __builtin_fma(x, p, 1.0 / 120);
  p = __builtin_fma(x, p, 1.0 / 24);
  p = __builtin_fma(x, p, 1.0 / 6);
  p = __builtin_fma(x, p, 0.5);
  return __builtin_fma(x * x, p, x);
}
__device__ __forceinline__ double em_general(double x) {  // any x <= 0: exp by range reduction, minus one; exact 0 below -745
  double xc = __builtin_fmax(x, -1000.0);
  double n = __builtin_rint(xc * 1.4426950408889634074);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, xc);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;  // 1/13!
  p = __builtin_fma(r, p, 1.0 / 479001600.0);
  p = __builtin_fma(r, p, 1.0 / 39916800.0);
  p = __builtin_fma(r, p, 1.0 / 3628800.0);
  p = __builtin_fma(r, p, 1.0 / 362880.0);
  p = __builtin_fma(r, p, 1.0 / 40320.0);
  p = __builtin_fma(r, p, 1.0 / 5040.0);
  p = __builtin_fma(r, p, 1.0 / 720.0);
  p = __builtin_fma(r, p, 1.0 / 120.0);
  p = __builtin_fma(r, p, 1.0 / 24.0);
  p = __builtin_fma(r, p, 1.0 / 6.0);
  p = __builtin_fma(r, p, 0.5);
  double e = __builtin_fma(r * r, p, r) + 1.0;
  e = __builtin_amdgcn_ldexp(e, (int)n);
  double g = e - 1.0;
  return (x > -0.0625) ? em_small(x) : g;  // also routes NaN to the propagating branch
}

// Tiers of the per-step exponentials: 0 = |a t| < 2^-7, 1 = |a t| < 2^-4, 2 = any t.  The tier is chosen per wave and
// per group of records (lanes of a wave sit on the same SNPs, so t is uniform up to per-pair dropped SNPs and |a t|
// differs only through a): straight-line code for the whole group, no per-step branch.  The likelihood sweeps (VALU-bound)
// have a tier -1 below tier 0: |a t| < 2^-8, where both series stop one term earlier (absolute truncation error
// < 5e-18 in u and < 2.5e-17 in w, a tenth of an ulp of the vector they are added to).

// ---- per-lane constants, hoisted out of the SNP loop (src/relateHMM.cpp:1440-1490) -------------------------
// With m1 = expm1(-a t), m2 = expm1(-a k1 t), x = -a t:
//     u = T01 = T21 = -k1 m1,     T10 = u k0/k1,   T12 = u k2/k1
//     w = (m2 - k1 m1)/(k1 - 1) = sum_{j>=2} g_j x^j,   g_j = k1 (1 + k1 + ... + k1^(j-2)) / j!
//     T02 = k2 w,   T20 = k0 w
// The series form of w has no cancellation and no division; it is used in tier 0 (terms through x^6, absolute
// truncation error < 3e-18) and the closed form in the other tiers.  k1 == 1 forces k0 = k2 = 0, so T02 = T20 = 0 as
// the reference sets them (:1461-1489); k1 == 0 gives u = w = 0.
__device__ inline int hi_word(double x) { return (int)(__builtin_bit_cast(long long, x) >> 32); }

struct LaneC {
  double na, nak1;      // -a, -a*k1
  double nk1;           // -k1
  double k0, k1, k2;
  double r10, r12;      // k0/k1, k2/k1 (both 1 when k1 == 0)
  double ikm1;          // 1/(k1-1); 0 when k1 == 1
  double g2, g3, g4, g5, g6;
  int hlim0, hlim1, hlimm;  // high words of the t below which |a t| < 2^-7 / 2^-4 / 2^-8 (group_tier)
};

__device__ inline LaneC make_lane(double a, double k2, double k1, double k0) {
  LaneC C;
  C.na = -a;
  C.nak1 = -a * k1;
  C.nk1 = -k1;
  C.k0 = k0; C.k1 = k1; C.k2 = k2;
  C.hlimm = hi_word(0x1p-8 / a);
  C.hlim0 = hi_word(0x1p-7 / a);
  C.hlim1 = hi_word(0x1p-4 / a);
  C.ikm1 = (k1 == 1) ? 0.0 : 1.0 / (k1 - 1);
  if (k1 == 0) { C.r10 = 1; C.r12 = 1; } else { C.r10 = k0 / k1; C.r12 = k2 / k1; }
  double s = 1.0;               // 1 + k1 + ... + k1^(j-2)
  C.g2 = k1 * s * (1.0 / 2);
  s = __builtin_fma(s, k1, 1.0);
  C.g3 = k1 * s * (1.0 / 6);
  s = __builtin_fma(s, k1, 1.0);
  C.g4 = k1 * s * (1.0 / 24);
  s = __builtin_fma(s, k1, 1.0);
  C.g5 = k1 * s * (1.0 / 120);
  s = __builtin_fma(s, k1, 1.0);
  C.g6 = k1 * s * (1.0 / 720);
  return C;
}

// m1 = expm1(-a t) alone (double-recombination model)
template <int TIER>
__device__ __forceinline__ double m1_at(const LaneC &C, double t) {
  double x = C.na * t;
  return TIER <= 0 ? em_tiny(x, x * x) : (TIER == 1 ? em_small(x) : em_general(x));
}

// the two numbers that carry the distance t into the transition matrix
struct Trans {
  double u, w;
};
template <int TIER>
__device__ __forceinline__ Trans trans_at(const LaneC &C, double t) {
  Trans T;
  double x = C.na * t;
  if (TIER < 0) {
    double xx = x * x;
    T.u = C.nk1 * em_tiny5(x, xx);
    double q = __builtin_fma(x, C.g5, C.g4);
    q = __builtin_fma(x, q, C.g3);
    q = __builtin_fma(x, q, C.g2);
    T.w = xx * q;
  } else if (TIER == 0) {
    double xx = x * x;
    T.u = C.nk1 * em_tiny(x, xx);
    double q = __builtin_fma(x, C.g6, C.g5);
    q = __builtin_fma(x, q, C.g4);
    q = __builtin_fma(x, q, C.g3);
    q = __builtin_fma(x, q, C.g2);
    T.w = xx * q;
  } else {
    double x2 = C.nak1 * t;
    double m1 = TIER == 1 ? em_small(x) : em_general(x);
    double m2 = TIER == 1 ? em_small(x2) : em_general(x2);
    T.u = C.nk1 * m1;
    T.w = __builtin_fma(C.nk1, m1, m2) * C.ikm1;
  }
  return T;
}

// v <- (v T) o S for NV row vectors sharing one T (forward step); DR = double recombination (dbl_recom, :501-507)
//   new0 = f0 + u (f1 r10 - f0) + w (f2 k0 - f0 k2)
//   new1 = f1 + u (f0 + f2 - f1 (r10 + r12)) = f1 - u (A + D),  A = f1 r10 - f0,  D = f1 r12 - f2
//   new2 = f2 + u (f1 r12 - f2) - w (f2 k0 - f0 k2)
template <int NV, bool DR, int TIER>
__device__ __forceinline__ void fwd_step(const LaneC &C, const Rec &r, double (&x)[NV][3]) {
  if (DR) {
    double m1 = m1_at<TIER>(C, r.t);  // T_xy = (1 - e1) k_y + [x == y] e1
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double F = (x[v][0] + x[v][1]) + x[v][2];
      double g0 = __builtin_fma(-C.k0, F, x[v][0]), g1 = __builtin_fma(-C.k1, F, x[v][1]),
             g2 = __builtin_fma(-C.k2, F, x[v][2]);
      x[v][0] = __builtin_fma(m1, g0, x[v][0]);
      x[v][1] = __builtin_fma(m1, g1, x[v][1]) * __builtin_fabs(r.r1);
      x[v][2] = __builtin_fma(m1, g2, x[v][2]) * r.r2;
    }
    return;
  }
  Trans T = trans_at<TIER>(C, r.t);
#pragma unroll
  for (int v = 0; v < NV; v++) {
    double f0 = x[v][0], f1 = x[v][1], f2 = x[v][2];
    double A = __builtin_fma(f1, C.r10, -f0);
    double B = __builtin_fma(f2, C.k0, -(f0 * C.k2));
    double D = __builtin_fma(f1, C.r12, -f2);
    double u0 = __builtin_fma(T.w, B, __builtin_fma(T.u, A, f0));
    double u1 = __builtin_fma(-T.u, A + D, f1);  // f0 + f2 - f1 (r10 + r12) = -(A + D): the columns of T - I sum to zero
    double u2 = __builtin_fma(-T.w, B, __builtin_fma(T.u, D, f2));
    x[v][0] = u0;
    x[v][1] = u1 * __builtin_fabs(r.r1);  // a dropped SNP carries r1 = -1
    x[v][2] = u2 * r.r2;
  }
}

// b <- T (S o b) (backward step through the record of the LATER SNP):  n_x = q_x + sum_{y != x} T_xy (q_y - q_x)
template <bool DR, int TIER>
__device__ __forceinline__ void bwd_step(const LaneC &C, const Rec &r, double &b0, double &b1, double &b2) {
  double q0 = b0, q1 = __builtin_fabs(r.r1) * b1, q2 = r.r2 * b2;
  if (DR) {
    double m1 = m1_at<TIER>(C, r.t);
    double G = __builtin_fma(C.k2, q2, __builtin_fma(C.k1, q1, C.k0 * q0));  // sum_y k_y q_y
    b0 = __builtin_fma(m1, q0 - G, q0);
    b1 = __builtin_fma(m1, q1 - G, q1);
    b2 = __builtin_fma(m1, q2 - G, q2);
    return;
  }
  Trans T = trans_at<TIER>(C, r.t);
  double d10 = q1 - q0, d20 = q2 - q0, d21 = q2 - q1;
  double z = __builtin_fma(-C.r10, d10, C.r12 * d21);
  b0 = __builtin_fma(T.w, C.k2 * d20, __builtin_fma(T.u, d10, q0));
  b1 = __builtin_fma(T.u, z, q1);
  b2 = __builtin_fma(-T.w, C.k0 * d20, __builtin_fma(-T.u, d21, q2));
}

// exact rescale of a 3-vector by a power of two; returns the exponent removed (0 for an all-zero vector)
__device__ __forceinline__ int rescale3(double &x0, double &x1, double &x2) {
  double m = __builtin_fmax(x0, __builtin_fmax(x1, x2));
  int e = (m > 0) ? __builtin_amdgcn_frexp_exp(m) : 0;
  x0 = __builtin_amdgcn_ldexp(x0, -e);
  x1 = __builtin_amdgcn_ldexp(x1, -e);
  x2 = __builtin_amdgcn_ldexp(x2, -e);
  return e;
}

__device__ inline bool lane_params(const double *__restrict__ params, int l, double alim1, bool reject, double &a,
                                   double &k2, double &k1, double &k0) {
  a = params[3 * l]; k2 = params[3 * l + 1]; k1 = params[3 * l + 2];
  k0 = 1 - (k1 + k2);
  return !(reject && (k2 < 0 || k1 < 0 || k0 < 0 || k2 > 1 || a > alim1));  // like()'s rejection, :1436
}

// lane -> pair of the staged block: an explicit list (the optimiser's lanes), or lane / div (E points per pair in order)
struct LaneMap {
  const int32_t *pair;
  int div;
  __device__ inline int operator()(int l) const { return pair ? pair[l] : l / div; }
};

__device__ inline bool rec_dropped(const Rec &r) { return __builtin_signbit(r.r1); }

// The record as this lane must step through it: the shared distance, unless the record carries its own (sign bit of
// r2: dropped -> |r2| is the distance and the emission is 1; kept -> distance 0 and |r2| is the emission).
__device__ __forceinline__ Rec rec_lane(const Rec &r) {
  Rec e;
  const bool own = __builtin_signbit(r.r2), dropped = __builtin_signbit(r.r1);
  e.t = own ? (dropped ? __builtin_fabs(r.r2) : 0.0) : r.t;
  e.r1 = r.r1;
  e.r2 = (own && dropped) ? 1.0 : __builtin_fabs(r.r2);
  return e;
}
// does any lane of the wave hold a record with a distance of its own among these?
template <int N>
__device__ __forceinline__ bool own_distance(const Rec (&R)[N]) {
  int m = hi_word(R[0].r2);
#pragma unroll
  for (int u = 1; u < N; u++) m |= hi_word(R[u].r2);
  return __builtin_amdgcn_ballot_w64(m < 0) != 0;
}
template <int N>
__device__ __forceinline__ void rec_lane_all(const Rec (&R)[N], Rec (&E)[N]) {
#pragma unroll
  for (int u = 0; u < N; u++) E[u] = rec_lane(R[u]);
}

// wave-uniform tier of a group of N records: 0 if every lane has |a t| < 2^-7 for all of them, 1 if < 2^-4, else 2
// (-1 below 2^-8 when LOW).  t is +0.0, -0.0 (dropped) or positive, so its order is the order of its high word as a
// signed integer: the comparison runs on high words (3 integer max + compares instead of 7 canonicalising v_max_f64
// per group); a t whose high word equals the limit's counts as above it, which only ever picks the more careful tier.
// NaN has the largest high word and goes to tier 2, where it propagates.
template <int N, bool LOW = false>
__device__ __forceinline__ int group_tier_h(int hlim0, int hlim1, int hlimm, const Rec (&R)[N]) {
  int hmax = hi_word(R[0].t);
#pragma unroll
  for (int u = 1; u < N; u++) hmax = max(hmax, hi_word(R[u].t));
  if (__builtin_amdgcn_ballot_w64(!(hmax < hlim0)) == 0)
    return (LOW && __builtin_amdgcn_ballot_w64(!(hmax < hlimm)) == 0) ? -1 : 0;
  if (__builtin_amdgcn_ballot_w64(!(hmax < hlim1)) == 0) return 1;
  return 2;
}
template <int N, bool LOW = false>
__device__ __forceinline__ int group_tier(const LaneC &C, const Rec (&R)[N]) {
  return group_tier_h<N, LOW>(C.hlim0, C.hlim1, C.hlimm, R);
}

// one vector of one parameter point through one record (the NV = 1 case of fwd_step on a plain array)
template <bool DR, int TIER>
__device__ __forceinline__ void fwd_step1(const LaneC &C, const Rec &r, double (&x)[3]) {
  double (&xv)[1][3] = *reinterpret_cast<double (*)[1][3]>(&x);
  fwd_step<1, DR, TIER>(C, r, xv);
}

template <int NV, bool DR, int TIER>
__device__ __forceinline__ void fwd_group_t(const LaneC &C, const Rec (&R)[kGroup], double (&x)[NV][3]) {
#pragma unroll
  for (int u = 0; u < kGroup; u++) fwd_step<NV, DR, TIER>(C, R[u], x);
}
template <int NV, bool DR>
__device__ __forceinline__ void fwd_group(const LaneC &C, const Rec (&R)[kGroup], double (&x)[NV][3]) {
  int tier = group_tier<kGroup, !DR>(C, R);
  if (tier < 0) fwd_group_t<NV, DR, -1>(C, R, x);
  else if (tier == 0) fwd_group_t<NV, DR, 0>(C, R, x);
  else if (tier == 1) fwd_group_t<NV, DR, 1>(C, R, x);
  else fwd_group_t<NV, DR, 2>(C, R, x);
}

// Streams the records of rows [sb, se) of one column through two callbacks: groups of kGroup records (the bulk) and
// single records (a tail of < NBUF * kGroup rows).  NBUF groups are in flight per lane in a statically indexed register
// ring, so a wave always has (NBUF - 1) * 64 B per lane on the way from HBM while it computes: with 2-3 waves per SIMD
// that covers the ~2 us loaded HBM latency.
template <int NBUF, class FG, class FT>
__device__ __forceinline__ void stream_records(const RecCol &col, int sb, int se, int M, FG &&on_group, FT &&on_tail) {
  constexpr int span = NBUF * kGroup;
  int s = sb;  // even: 0 or a chunk start
  if (col.general) {  // record by record through the general-distance steps (rec_lane): rare, see kernels_emis.hpp
    for (; s < se; s++) on_tail(col.at(s), s);
    return;
  }
  Rec buf[NBUF][kGroup];
  if (s + span <= se) {
#pragma unroll
    for (int j = 0; j < NBUF; j++) col.group(s + j * kGroup, buf[j]);
  }
  for (; s + span <= se; s += span) {
#pragma unroll
    for (int j = 0; j < NBUF; j++) {
      on_group(buf[j], s + j * kGroup);
      col.group(s + span + j * kGroup, buf[j]);  // the group one ring turn ahead
    }
  }
  for (; s < se; s++) on_tail(col.at(s), s);
}

// groups in flight per lane, per kernel (register budget: 16 VGPRs per group)
static constexpr int kNbufChunk = 2;

// forward recursion over rows [sb, se) for NV vectors
template <int NV, bool DR, int NBUF, class FC>
__device__ __forceinline__ void sweep(const RecCol &col, int sb, int se, int M, const LaneC &C, double (&x)[NV][3], int (&ex)[NV],
                             FC &&at_tile) {
  stream_records<NBUF>(
      col, sb, se, M,
      [&](const Rec(&R)[kGroup], int s) {
        if ((s & 12) == 0) at_tile(s >> 4);
        fwd_group<NV, DR>(C, R, x);
        if ((s & 12) == 12) {  // every fourth group: 16 steps between renormalisations (emission ratios <= ~1e8 each way)
#pragma unroll
          for (int v = 0; v < NV; v++) ex[v] += rescale3(x[v][0], x[v][1], x[v][2]);
        }
      },
      [&](const Rec &r, int s) {
        if ((s & 15) == 0) at_tile(s >> 4);
        fwd_step<NV, DR, 2>(C, rec_lane(r), x);
        if ((s & 7) == 7) {  // the tail can be NBUF * kGroup - 1 records long
#pragma unroll
          for (int v = 0; v < NV; v++) ex[v] += rescale3(x[v][0], x[v][1], x[v][2]);
        }
      });
#pragma unroll
  for (int v = 0; v < NV; v++) ex[v] += rescale3(x[v][0], x[v][1], x[v][2]);
}

// The factor taken out of a pair's records at staging time (kernels_emis.hpp): mantissa and exponent per pair.
struct PairScale {
  const double *__restrict__ m;
  const int *__restrict__ e;
};
// -log-likelihood from the final vector sum `tot` with exponent `ex` and the pair's factor
__device__ inline double neg_loglik(double tot, long ex, const PairScale &S, int pair) {
  return -(log(tot * S.m[pair]) + (double)(ex + S.e[pair]) * kLn2);
}

// Exponents of a likelihood vector.  What a single common exponent would lose: a component that does not reach the
// likelihood may outgrow, over 10^5 SNPs, the ones that do by more than the exponent range.
//  * A state with k_x == 0 is never entered (T_yx = 0 for every y) and never held (alpha_x == 0): its beta is cleared
//    at the start and at every rescaling (k = (0, 1, 0), the poLike lane, gave beta_1 = 0 without).
//  * k1 == 0 makes T the identity (not under double recombination): states 0 and 2 are two separate chains, alpha_0 may win the first half and beta_2
//    the second.  Such a lane rescales every component on its own exponent (exact: the components never mix).
// Every other lane mixes its states at every step (T_xy >= k_y w > 0), which bounds the ratios, and uses one exponent.
struct HalfNorm {
  double z0, z1, z2;  // 0 for a state with k_x == 0
  bool iso;           // k1 == 0: independent components
  int e0 = 0, e1 = 0, e2 = 0;
  __device__ HalfNorm() : z0(1), z1(1), z2(1), iso(false) {}
  __device__ HalfNorm(double k0, double k1, double k2, bool dr)  // the double-recombination T mixes 0 and 2 whatever k1 is
      : z0(k0 == 0 ? 0.0 : 1.0), z1(k1 == 0 ? 0.0 : 1.0), z2(k2 == 0 ? 0.0 : 1.0), iso(!dr && k1 == 0) {}
  __device__ inline void operator()(double &x0, double &x1, double &x2) {
    x0 *= z0; x1 *= z1; x2 *= z2;
    if (iso) {
      int f0 = x0 > 0 ? __builtin_amdgcn_frexp_exp(x0) : 0, f2 = x2 > 0 ? __builtin_amdgcn_frexp_exp(x2) : 0;
      x0 = __builtin_amdgcn_ldexp(x0, -f0);
      x2 = __builtin_amdgcn_ldexp(x2, -f2);
      e0 += f0; e2 += f2;
    } else {
      int e = rescale3(x0, x1, x2);
      e0 += e; e1 += e; e2 += e;
    }
  }
  __device__ inline void store(double *__restrict__ q, size_t nlp, double x0, double x1, double x2) const {
    q[0] = x0; q[nlp] = x1; q[2 * nlp] = x2;
    q[3 * nlp] = (double)e0; q[4 * nlp] = (double)e1; q[5 * nlp] = (double)e2;
  }
};

// sum_c pr[c] 2^{ex[c]} as (value in [0.5, 1), exponent); NaN from a broken recursion propagates
__device__ inline double combine3(const double (&pr)[3], const int (&ex)[3], long &eout) {
  int emax = -2147483647;
#pragma unroll
  for (int c = 0; c < 3; c++)
    if (pr[c] > 0 && ex[c] > emax) emax = ex[c];
  double tot = 0;
#pragma unroll
  for (int c = 0; c < 3; c++)
    if (pr[c] > 0) tot += ldexp(pr[c], max(ex[c] - emax, -2200));
#pragma unroll
  for (int c = 0; c < 3; c++)
    if (!(pr[c] >= 0)) tot = pr[c];
  int fe = tot > 0 ? __builtin_amdgcn_frexp_exp(tot) : 0;
  eout = (long)emax + fe;
  return __builtin_amdgcn_ldexp(tot, -fe);
}

// A lane of the likelihood sweeps carries NV parameter points of ONE pair: the record ring, its loads, the tier test
// and the loop are shared by NV independent chains (the optimiser asks for 3-7 points of a pair at a time, and several
// restarts of a pair run side by side).  Per point-step that is 1.5 / NV vector loads and 1 / NV of the loop overhead,
// and NV chains interleave in the VALU.  Slots: lane l owns points [l NV, l NV + NV) of the launch (the host pads a
// pair's points to a multiple of NV with copies); a rejected point rides along on harmless parameters.
template <int NV>
struct LaneSet {
  LaneC C[NV];
  HalfNorm norm[NV];
  bool ok[NV];
  int hl0, hl1, hlm;  // tier limits: the most careful of the lane's points
  bool any;
  __device__ inline void init(const double *__restrict__ params, int slot0, int nslots, double alim1, bool dr,
                              double (&x)[NV][3]) {
    hl0 = hl1 = hlm = 2147483647;
    any = false;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double a, k2, k1, k0;
      ok[v] = lane_params(params, min(slot0 + v, nslots - 1), alim1, true, a, k2, k1, k0);
      any |= ok[v];
      if (!ok[v]) { a = 0.01; k2 = 0; k1 = 0; k0 = 1; }
      C[v] = make_lane(a, k2, k1, k0);
      norm[v] = HalfNorm(k0, k1, k2, dr);
      x[v][0] = k0; x[v][1] = k1; x[v][2] = k2;
      hl0 = min(hl0, C[v].hlim0); hl1 = min(hl1, C[v].hlim1); hlm = min(hlm, C[v].hlimm);
    }
  }
};

template <int NV, bool DR, int TIER>
__device__ __forceinline__ void fwd_group_mp(const LaneSet<NV> &L, const Rec (&R)[kGroup], double (&x)[NV][3]) {
#pragma unroll
  for (int u = 0; u < kGroup; u++)
#pragma unroll
    for (int v = 0; v < NV; v++) fwd_step1<DR, TIER>(L.C[v], R[u], x[v]);
}
template <int NV, bool DR, int TIER>
__device__ __forceinline__ void bwd_group_mp(const LaneSet<NV> &L, const Rec (&R)[kGroup], double (&b)[NV][3]) {
#pragma unroll
  for (int u = kGroup - 1; u >= 0; u--)
#pragma unroll
    for (int v = 0; v < NV; v++) bwd_step<DR, TIER>(L.C[v], R[u], b[v][0], b[v][1], b[v][2]);
}

// forward half of a likelihood sweep: rows [sb, se) from x, exponents in L.norm
template <bool DR, int NBUF, int NV>
__device__ __forceinline__ void like_forward(const RecCol &col, int sb, int se, int M, LaneSet<NV> &L, double (&x)[NV][3]) {
  auto renorm = [&]() {
#pragma unroll
    for (int v = 0; v < NV; v++) L.norm[v](x[v][0], x[v][1], x[v][2]);
  };
  stream_records<NBUF>(
      col, sb, se, M,
      [&](const Rec(&R)[kGroup], int s) {
        int tier = group_tier_h<kGroup, !DR>(L.hl0, L.hl1, L.hlm, R);
        if (tier < 0) fwd_group_mp<NV, DR, -1>(L, R, x);
        else if (tier == 0) fwd_group_mp<NV, DR, 0>(L, R, x);
        else if (tier == 1) fwd_group_mp<NV, DR, 1>(L, R, x);
        else fwd_group_mp<NV, DR, 2>(L, R, x);
        if ((s & 12) == 12) renorm();  // every fourth group: 16 steps between renormalisations
      },
      [&](const Rec &r0, int s) {
        const Rec r = rec_lane(r0);
#pragma unroll
        for (int v = 0; v < NV; v++) fwd_step1<DR, 2>(L.C[v], r, x[v]);
        if ((s & 7) == 7) renorm();  // the tail can be NBUF * kGroup - 1 records long
      });
  renorm();
}

// K5, one sweep.  out[slot] = -log-likelihood, SYS_MAX for a rejected point, NaN/inf if the recursion broke down.
// NBUF = record groups in flight per lane, chosen by the host from the waves per SIMD the launch can fill: a lane is
// a serial chain per point, so with one wave per SIMD only the depth of its own prefetch hides HBM latency; with two
// waves NBUF 4; from three waves on NBUF 2, where a deeper ring would cost the third wave.
template <bool DR, int NBUF, int NV>
__global__ void __launch_bounds__(256)
like_kernel(const double *__restrict__ rec, const double *__restrict__ tsh, const int *__restrict__ own, int M, int RP, LaneMap lane_pair, PairScale scale,
            const double *__restrict__ params, int nslots, double alim1, double *__restrict__ out) {
  const int slot0 = (blockIdx.x * blockDim.x + threadIdx.x) * NV;
  if (slot0 >= nslots) return;
  LaneSet<NV> L;
  double x[NV][3];
  L.init(params, slot0, nslots, alim1, DR, x);
  const int lp = lane_pair(slot0);
  if (L.any) like_forward<DR, NBUF, NV>(RecCol(rec, tsh, own, RP, lp, M), 0, M, M, L, x);
#pragma unroll
  for (int v = 0; v < NV; v++) {
    if (slot0 + v >= nslots) break;
    const double pr[3] = {x[v][0], x[v][1], x[v][2]};
    const int ex[3] = {L.norm[v].e0, L.norm[v].e1, L.norm[v].e2};
    long e;
    double tot = combine3(pr, ex, e);
    out[slot0 + v] = L.ok[v] ? neg_loglik(tot, e, scale, lp) : 10000000.0;
  }
}

// The same records from the last row down (the backward half of the two-sided likelihood): groups of kGroup rows
// [base, base + kGroup) are handed over in descending order of base, then single rows down to sb.
template <int NBUF, class FG, class FT>
__device__ __forceinline__ void stream_records_rev(const RecCol &col, int sb, int se, FG &&on_group, FT &&on_tail) {
  constexpr int span = NBUF * kGroup;
  int s = se;  // rows [sb, s) are still to do
  if (col.general) {
    for (s = s - 1; s >= sb; s--) on_tail(col.at(s), s);
    return;
  }
  for (; (s & (kGroup - 1)) != 0 && s > sb; s--) on_tail(col.at(s - 1), s - 1);  // down to a group boundary
  Rec buf[NBUF][kGroup];
  if (s - span >= sb) {
#pragma unroll
    for (int j = 0; j < NBUF; j++) col.group(s - (j + 1) * kGroup, buf[j]);
  }
  for (; s - span >= sb; s -= span) {
#pragma unroll
    for (int j = 0; j < NBUF; j++) {
      on_group(buf[j], s - (j + 1) * kGroup);
      col.group(max(s - span - (j + 1) * kGroup, 0), buf[j]);  // one ring turn further down; clamped to row 0
    }
  }
  for (s = s - 1; s >= sb; s--) on_tail(col.at(s), s);
}

template <bool DR, int TIER>
__device__ __forceinline__ void bwd_group_t(const LaneC &C, const Rec (&R)[kGroup], double &b0, double &b1, double &b2) {
#pragma unroll
  for (int u = kGroup - 1; u >= 0; u--) bwd_step<DR, TIER>(C, R[u], b0, b1, b2);
}

// K5, two-sided.  The likelihood is alpha_mid . beta_mid for any cut: workgroups [0, nb) sweep rows [0, mid) forward
// from the start vector k, workgroups [nb, 2 nb) sweep rows [mid, M) backward from the vector of ones, and
// like_join_kernel takes the dot product.  Twice the lanes at half the length for the same arithmetic: a launch of
// up to one wave per SIMD (a serial chain per lane, 62 % of the issue rate) becomes one of up to two.
// part: 12 x nlp doubles (alpha[3], its three exponents, beta[3], its three exponents), nlp = lanes padded to 256.
//
template <bool DR, int NBUF, int NV>
__global__ void __launch_bounds__(256)
like_halves_kernel(const double *__restrict__ rec, const double *__restrict__ tsh, const int *__restrict__ own, int M, int RP, LaneMap lane_pair,
                   const double *__restrict__ params, int nslots, int nsp, double alim1, double *__restrict__ part) {
  // nsp = slots padded to 256 NV: workgroups [0, nb) forward, [nb, 2 nb) backward; part is indexed by slot
  const int nb = nsp / (256 * NV);
  const bool back = (int)blockIdx.x >= nb;
  const int slot0 = (((int)blockIdx.x - (back ? nb : 0)) * 256 + threadIdx.x) * NV;
  if (slot0 >= nslots) return;
  LaneSet<NV> L;
  double x[NV][3];
  L.init(params, slot0, nslots, alim1, DR, x);
  if (!L.any) return;  // like_join_kernel rejects the points
  const int mid = (M / 2) & ~(kGroup - 1);
  const RecCol col(rec, tsh, own, RP, lane_pair(slot0), M);
  if (!back) {
    like_forward<DR, NBUF, NV>(col, 0, mid, M, L, x);
#pragma unroll
    for (int v = 0; v < NV; v++)
      if (slot0 + v < nslots) L.norm[v].store(part + slot0 + v, (size_t)nsp, x[v][0], x[v][1], x[v][2]);
    return;
  }
#pragma unroll
  for (int v = 0; v < NV; v++) { x[v][0] = 1; x[v][1] = 1; x[v][2] = 1; }
  auto renorm = [&]() {
#pragma unroll
    for (int v = 0; v < NV; v++) L.norm[v](x[v][0], x[v][1], x[v][2]);
  };
  stream_records_rev<NBUF>(
      col, mid, M,
      [&](const Rec(&R)[kGroup], int s) {
        int tier = group_tier_h<kGroup, !DR>(L.hl0, L.hl1, L.hlm, R);
        if (tier < 0) bwd_group_mp<NV, DR, -1>(L, R, x);
        else if (tier == 0) bwd_group_mp<NV, DR, 0>(L, R, x);
        else if (tier == 1) bwd_group_mp<NV, DR, 1>(L, R, x);
        else bwd_group_mp<NV, DR, 2>(L, R, x);
        if ((s & 12) == 0) renorm();
      },
      [&](const Rec &r0, int s) {
        const Rec r = rec_lane(r0);
#pragma unroll
        for (int v = 0; v < NV; v++) bwd_step<DR, 2>(L.C[v], r, x[v][0], x[v][1], x[v][2]);
        if ((s & 7) == 0) renorm();
      });
  renorm();
#pragma unroll
  for (int v = 0; v < NV; v++)
    if (slot0 + v < nslots) L.norm[v].store(part + (size_t)6 * nsp + slot0 + v, (size_t)nsp, x[v][0], x[v][1], x[v][2]);
}

__global__ void __launch_bounds__(256)
like_join_kernel(const double *__restrict__ part, LaneMap lane_pair, PairScale scale, const double *__restrict__ params,
                 int nlanes, int nlp, double alim1, double *__restrict__ out) {
  int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlanes) return;
  double a, k2, k1, k0;
  if (!lane_params(params, l, alim1, true, a, k2, k1, k0)) { out[l] = 10000000.0; return; }
  double pr[3];
  int ex[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    pr[c] = part[(size_t)c * nlp + l] * part[(size_t)(6 + c) * nlp + l];
    ex[c] = (int)part[(size_t)(3 + c) * nlp + l] + (int)part[(size_t)(9 + c) * nlp + l];
  }
  long e;
  double tot = combine3(pr, ex, e);
  out[l] = neg_loglik(tot, e, scale, lane_pair(l));
}

// Chunk products.  Thread (chunk c, lane l) at index at = c * nlp + l (nlp = lanes padded to 256): rows X[i][.]
// (i = unit start vector) with exponents e[i]:
//   alpha_end[y] = sum_i alpha_start[i] X[i][y] 2^{e[i]}        beta_start[i] = 2^{e[i]} sum_y X[i][y] beta_end[y]
// Arrays are struct-of-arrays over n = C * nlp.
struct ChunkOut {
  double *m;  // 9 x n
  int *e;     // 3 x n
};
// Tile checkpoints of the decode (CKPT): the partial product of the chunk up to the start of every tile of kTile = 16
// SNPs, same row/exponent form.  Per tile and group of 64 pairs: nine planes of 64 doubles, then three of 64 ints
// (5 376 B back to back).  The backward pass turns one into the forward vector entering the tile
// (alpha_chunk_start . X) and recomputes the 16 vectors inside: 84 B per 16 SNPs in place of a forward pass of its own
// over the records.
static constexpr size_t kCkptGroupBytes = 9 * 512 + 3 * 256;
struct TileCkpt {
  unsigned char *base;
  size_t tile_stride;  // bytes: groups of 64 pairs x kCkptGroupBytes
  __device__ inline double *m(int tile, int p) const {
    return (double *)(base + (size_t)tile * tile_stride + (size_t)(p >> 6) * kCkptGroupBytes) + (p & 63);
  }
  __device__ inline int *e(int tile, int p) const {
    return (int *)(base + (size_t)tile * tile_stride + (size_t)(p >> 6) * kCkptGroupBytes + 9 * 512) + (p & 63);
  }
};

template <bool DR, bool CKPT>
__global__ void __launch_bounds__(256)
chunk_matrix_kernel(const double *__restrict__ rec, const double *__restrict__ tsh, const int *__restrict__ own, int M, int RP, LaneMap lane_pair,
                    const double *__restrict__ params, int nlanes, int nlp, int nchunk, int chunk_len, double alim1,
                    int reject, ChunkOut O, TileCkpt ck, int col0) {  // col0: first column of the block's pairs (CKPT)
  int bpc = nlp / 256;  // workgroups per chunk
  int c = blockIdx.x / bpc;
  int l = (blockIdx.x % bpc) * 256 + threadIdx.x;
  if (l >= nlanes) return;
  double a, k2, k1, k0;
  if (!lane_params(params, l, alim1, reject != 0, a, k2, k1, k0)) return;  // the combine kernel rejects the lane
  LaneC C = make_lane(a, k2, k1, k0);
  int sb = c * chunk_len, se = min(M, sb + chunk_len);
  double x[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  int ex[3] = {0, 0, 0};
  const int lp = lane_pair(l);
  sweep<3, DR, kNbufChunk>(RecCol(rec, tsh, own, RP, col0 + lp, M), sb, se, M, C, x, ex, [&](int tile) {
    if (CKPT) {
      double *qm = ck.m(tile, lp);
      int *qe = ck.e(tile, lp);
#pragma unroll
      for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int y = 0; y < 3; y++) qm[(i * 3 + y) * 64] = x[i][y];
        qe[i * 64] = ex[i];
      }
    }
  });
  size_t n = (size_t)nlp * nchunk, at = (size_t)c * nlp + l;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int y = 0; y < 3; y++) O.m[(size_t)(i * 3 + y) * n + at] = x[i][y];
    O.e[(size_t)i * n + at] = ex[i];
  }
}

// one chunk product in registers; the combine kernels load the next one while they fold the current one (their loops
// are serial per lane and latency-bound: one strided load round trip per chunk otherwise)
struct ChunkProd {
  double m[9];
  int e[3];
};
__device__ inline ChunkProd load_chunk(const ChunkOut &O, size_t n, size_t at) {
  ChunkProd X;
#pragma unroll
  for (int i = 0; i < 9; i++) X.m[i] = O.m[(size_t)i * n + at];
#pragma unroll
  for (int i = 0; i < 3; i++) X.e[i] = O.e[(size_t)i * n + at];
  return X;
}

// alpha <- alpha * (chunk product); returns the exponent moved out.  The common exponent is that of the largest
// term a_i 2^{e_i}, so a component is lost only when it is 2^-1074 below the leading one (with k1 = 0 the states are
// separate chains whose products drift thousands of binades apart and back).
__device__ inline int apply_chunk_fwd(double &f0, double &f1, double &f2, const ChunkProd &X) {
  double a[3] = {f0, f1, f2};
  int et[3], emax = -2147483647;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    et[i] = a[i] > 0 ? __builtin_amdgcn_frexp_exp(a[i]) + X.e[i] : -2147483647;
    emax = max(emax, et[i]);
  }
  if (emax == -2147483647) { f0 = f1 = f2 = 0; return 0; }
  double r0 = 0, r1 = 0, r2 = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (!(a[i] > 0)) continue;
    double w = ldexp(a[i], max(X.e[i] - emax, -2200));
    r0 += w * X.m[i * 3 + 0];
    r1 += w * X.m[i * 3 + 1];
    r2 += w * X.m[i * 3 + 2];
  }
  f0 = r0; f1 = r1; f2 = r2;
  return emax;
}

// K5 combine: folds the C chunk products of every lane into -log-likelihood.  A lane with k1 = 0 (T = I, not under
// double recombination) carries its states as separate chains, each on its own exponent (HalfNorm): the products are
// diagonal and the chains may drift further apart than a common exponent holds.
__global__ void __launch_bounds__(256)
like_combine_kernel(ChunkOut O, LaneMap lane_pair, PairScale scale, const double *__restrict__ params, int nlanes,
                    int nlp, int nchunk, double alim1, int double_recom, double *__restrict__ out) {
  int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlanes) return;
  double a, k2, k1, k0;
  if (!lane_params(params, l, alim1, true, a, k2, k1, k0)) { out[l] = 10000000.0; return; }
  size_t n = (size_t)nlp * nchunk;
  double f0 = k0, f1 = k1, f2 = k2;
  long esum = 0;
  double tot;
  // the fold is serial per lane and its loads are strided: kAhead products are fetched at a time (one round trip per
  // batch instead of one per chunk -- a launch of a few hundred lanes has up to ~1 000 chunks)
  constexpr int kAhead = 8;
  const bool iso = !double_recom && k1 == 0;
  int e0 = 0, e2 = 0;
  for (int cb = 0; cb < nchunk; cb += kAhead) {
    ChunkProd X[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) X[u] = load_chunk(O, n, (size_t)min(cb + u, nchunk - 1) * nlp + l);
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      if (cb + u >= nchunk) break;
      if (iso) {
        f0 *= X[u].m[0]; f2 *= X[u].m[8];
        int g0 = f0 > 0 ? __builtin_amdgcn_frexp_exp(f0) : 0, g2 = f2 > 0 ? __builtin_amdgcn_frexp_exp(f2) : 0;
        f0 = __builtin_amdgcn_ldexp(f0, -g0); f2 = __builtin_amdgcn_ldexp(f2, -g2);
        e0 += X[u].e[0] + g0; e2 += X[u].e[2] + g2;
      } else {
        esum += apply_chunk_fwd(f0, f1, f2, X[u]);
        esum += rescale3(f0, f1, f2);
      }
    }
  }
  if (iso) {
    const double pr[3] = {f0, 0.0, f2};
    const int ex[3] = {e0, 0, e2};
    tot = combine3(pr, ex, esum);
  } else {
    tot = (f0 + f1) + f2;
  }
  out[l] = neg_loglik(tot, esum, scale, lane_pair(l));
}

// ---- K6 ----------------------------------------------------------------------------------------------------
// Pass A (chunk_matrix_kernel<DR, true>) leaves every chunk's product and the tile checkpoints; decode_combine_kernel
// folds the products into the boundary vectors: astart[(c * nlp + p) * 3 ..] = forward vector entering chunk c,
// bend[...] = backward vector at the last SNP of chunk c; pass B (decode_bwd_kernel) walks every chunk backwards.  Any
// positive scale factor cancels in the posterior, so vectors are normalised with v_rcp_f64.
__device__ inline void norm_fast(double &x0, double &x1, double &x2) {
  double s = (x0 + x1) + x2;
  double inv = (s > 0) ? __builtin_amdgcn_rcp(s) : 1.0;
  x0 *= inv; x1 *= inv; x2 *= inv;
}

__global__ void __launch_bounds__(256)
decode_combine_kernel(ChunkOut O, const double *__restrict__ params, int P, int nlp, int nchunk,
                      double *__restrict__ astart, double *__restrict__ bend) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  size_t n = (size_t)nlp * nchunk;
  double k2 = params[3 * p + 1], k1 = params[3 * p + 2];
  double k0 = 1 - k2 - k1;  // decodek2, :1018
  double f0 = k0, f1 = k1, f2 = k2;
  ChunkProd X = load_chunk(O, n, (size_t)p);
  for (int c = 0; c < nchunk; c++) {
    size_t at = (size_t)c * nlp + p;
    ChunkProd Xn = load_chunk(O, n, (size_t)min(c + 1, nchunk - 1) * nlp + p);
    astart[at * 3] = f0; astart[at * 3 + 1] = f1; astart[at * 3 + 2] = f2;
    apply_chunk_fwd(f0, f1, f2, X);
    norm_fast(f0, f1, f2);
    X = Xn;
  }
  double b0 = 1, b1 = 1, b2 = 1;
  X = load_chunk(O, n, (size_t)(nchunk - 1) * nlp + p);
  for (int c = nchunk - 1; c >= 0; c--) {
    size_t at = (size_t)c * nlp + p;
    ChunkProd Xn = load_chunk(O, n, (size_t)max(c - 1, 0) * nlp + p);
    bend[at * 3] = b0; bend[at * 3 + 1] = b1; bend[at * 3 + 2] = b2;
    // beta_start[i] = 2^{e[i]} sum_y X[i][y] beta[y]
    double v[3];
    int emax = -2147483647;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      v[i] = X.m[i * 3 + 0] * b0 + X.m[i * 3 + 1] * b1 + X.m[i * 3 + 2] * b2;
      if (v[i] > 0) emax = max(emax, __builtin_amdgcn_frexp_exp(v[i]) + X.e[i]);
    }
    if (emax != -2147483647) {
      b0 = v[0] > 0 ? ldexp(v[0], max(X.e[0] - emax, -2200)) : 0;
      b1 = v[1] > 0 ? ldexp(v[1], max(X.e[1] - emax, -2200)) : 0;
      b2 = v[2] > 0 ? ldexp(v[2], max(X.e[2] - emax, -2200)) : 0;
      norm_fast(b0, b1, b2);
    }
    X = Xn;
  }
}

// Posterior tiles.  The backward pass works on tiles of kTile = 16 SNPs (chunk starts are multiples of 16) made of
// four groups of kGroup = 4.  From the tile's checkpoint it forms the forward vector entering the tile, walks the tile
// forward once keeping the vector entering each group, then takes the groups last to first: four forward vectors
// recomputed from the group's entry vector, four backward steps, four posteriors.  Recomputing costs ~1.75 forward
// steps per SNP in an HBM-bound kernel; a stored forward matrix would cost 24 B per (pair, SNP) written and read.
static constexpr int kTile = 16;
static constexpr int kTilePad = kTile + 1;
static constexpr int kBwdThreads = 64;  // one wave per workgroup: 13.5 KiB of LDS each

// tile row j holds IBD0 (j = 0), IBD1, IBD2; only the first NR are kept (rows the caller does not write never reach LDS)
template <bool DR, int TIER, int NR>
__device__ __forceinline__ void dec_tile(const LaneC &C, const Rec (&R)[kTile], double f0, double f1, double f2, int tb, int te,
                                double &b0, double &b1, double &b2, double (&tile)[NR][kBwdThreads][kTilePad]) {
  constexpr int NG = kTile / kGroup;
  double G[NG][3];
  {
    double x[1][3] = {{f0, f1, f2}};
#pragma unroll
    for (int g = 0; g < NG; g++) {
      G[g][0] = x[0][0]; G[g][1] = x[0][1]; G[g][2] = x[0][2];
      if (g == NG - 1) break;
#pragma unroll
      for (int u = 0; u < kGroup; u++) fwd_step<1, DR, TIER>(C, R[g * kGroup + u], x);
    }
  }
#pragma unroll
  for (int g = NG - 1; g >= 0; g--) {
    const int hb = tb + g * kGroup;
    if (hb >= te) continue;  // uniform: ragged last tile
    double F[kGroup][3];
    {
      double x[1][3] = {{G[g][0], G[g][1], G[g][2]}};
#pragma unroll
      for (int u = 0; u < kGroup; u++) {
        fwd_step<1, DR, TIER>(C, R[g * kGroup + u], x);
        F[u][0] = x[0][0]; F[u][1] = x[0][1]; F[u][2] = x[0][2];
      }
    }
#pragma unroll
    for (int u = kGroup - 1; u >= 0; u--) {
      if (hb + u >= te) continue;  // uniform
      const Rec &r = R[g * kGroup + u];
      double p0, p1, p2, sum;
      {
        // no contraction here: with fewer rows kept the products feed only the sum, and a product fused into the sum
        // in one instantiation but not in another would make the rows differ in the last bit between them
#pragma clang fp contract(off)
        p0 = F[u][0] * b0; p1 = F[u][1] * b1; p2 = F[u][2] * b2;
        sum = (p0 + p1) + p2;
      }
      double inv = __builtin_amdgcn_rcp(sum);
      inv = inv * __builtin_fma(-sum, inv, 2.0);
      inv = inv * __builtin_fma(-sum, inv, 2.0);
      bool dr = rec_dropped(r);
      const int e = g * kGroup + u;
      tile[0][threadIdx.x][e] = dr ? -1.0 : p0 * inv;
      if constexpr (NR > 1) tile[1][threadIdx.x][e] = dr ? -1.0 : p1 * inv;
      if constexpr (NR > 2) tile[2][threadIdx.x][e] = dr ? -1.0 : p2 * inv;
      bwd_step<DR, TIER>(C, r, b0, b1, b2);
      norm_fast(b0, b1, b2);
    }
  }
}

// A posterior as the reference's .post text carries it: "%.4g" (POSTPRECISION, src/conf.h:8), i.e. the double rounded
// to four significant decimal digits (the caller then rounds that once to float, as a float reader of the file does).
// -1 (dropped SNP), 0 and NaN pass through.  The scaled value is exact enough that only values within ~1e-12 relative
// of a rounding boundary could round the other way than printf does.
__device__ inline double round_sig4(double x) {
  if (!(x > 0)) return x;
  int e = (int)floor(log10(x));
  double s = x * pow(10.0, (double)(3 - e));
  if (s < 1000.0) { e--; s = x * pow(10.0, (double)(3 - e)); }
  else if (s >= 10000.0) { e++; s = x * pow(10.0, (double)(3 - e)); }
  double r = rint(s);
  if (r >= 10000.0) { r = 1000.0; e++; }
  return (3 - e >= 0) ? r / pow(10.0, (double)(3 - e)) : r * pow(10.0, (double)(e - 3));
}

// Backward sweep inside one chunk + posterior.  post is P x 3 x M (pair-major, as the host writer wants it): row0
// IBD2, row1 IBD1, row2 IBD0 (decodek2, :1133-1137); -1 where the SNP is dropped (.post writer, :126-133).
// posterior_x(s) = fwd_x(s) bwd_x(s) / sum_y fwd_y(s) bwd_y(s): every scale factor cancels.
// Lanes hold consecutive pairs, whose posterior rows lie 3*M doubles apart: results of kTile SNPs are transposed
// through LDS so that every pair row is written in whole 128-byte lines.  Workgroup = one wave = 64 pairs of one chunk.
// WROWS = 0 (OutT = double): post[p][3][M], rows IBD2, IBD1, IBD0 (the reference's dMatrix).  WROWS = 1..3:
// post[p][WROWS][M] in the order the .post writer takes them (IBD0, IBD1, IBD2; :3771-3782), the first WROWS of them
// -- rows nobody reads are not written (8 B per pair and SNP each); OutT = float: each value the double result rounded
// once to float.
template <bool DR, typename OutT, int WROWS>  // WROWS: 0 = the reference's three rows; 1..3 = that many rows in the writer's order
__global__ void __launch_bounds__(kBwdThreads)
decode_bwd_kernel(const double *__restrict__ rec, const double *__restrict__ tsh, const int *__restrict__ own, int M, int P, int RP, int col0,
                  const double *__restrict__ params,
                  int nlp, int chunk_len, const double *__restrict__ astart, const double *__restrict__ bend,
                  TileCkpt ck, OutT *__restrict__ post, int sig4) {  // sig4: float rows rounded as the .post text
  constexpr bool WORDER = WROWS > 0;
  constexpr int nrow = WORDER ? WROWS : 3;
  __shared__ double tile[nrow][kBwdThreads][kTilePad];
  int bpc = (P + kBwdThreads - 1) / kBwdThreads;
  int c = blockIdx.x / bpc;
  int pbase = (blockIdx.x % bpc) * kBwdThreads;
  int p = pbase + threadIdx.x;
  int pc = p < P ? p : P - 1;  // dead lanes shadow a valid pair and write nothing
  double a = params[3 * pc], k2 = params[3 * pc + 1], k1 = params[3 * pc + 2];
  double k0 = 1 - k2 - k1;
  LaneC C = make_lane(a, k2, k1, k0);
  int sb = c * chunk_len, se = min(M, sb + chunk_len);  // sb is a multiple of kTile
  double a0 = k0, a1 = k1, a2 = k2;
  double b0 = 1, b1 = 1, b2 = 1;
  if (astart) {
    size_t at = ((size_t)c * nlp + pc) * 3;
    a0 = astart[at]; a1 = astart[at + 1]; a2 = astart[at + 2];
    b0 = bend[at]; b1 = bend[at + 1]; b2 = bend[at + 2];
  }
  const RecCol col(rec, tsh, own, RP, col0 + pc, M);
  // The records and the checkpoint of the NEXT tile are fetched before this tile is worked on: the kernel holds one or
  // two waves per SIMD (228 VGPRs, 26 KB of LDS a wave), so little else would cover its memory latency while it computes.
  auto load_tile = [&](int tb, Rec (&R)[kTile], ChunkProd &X) {
#pragma unroll
    for (int u = 0; u < kTile; u += 2) col.at2(min((tb + u) >> 1, col.hmax), R[u], R[u + 1]);
    const double *qm = ck.m(tb / kTile, pc);
    const int *qe = ck.e(tb / kTile, pc);
#pragma unroll
    for (int i = 0; i < 9; i++) X.m[i] = qm[i * 64];
#pragma unroll
    for (int i = 0; i < 3; i++) X.e[i] = qe[i * 64];
  };
  Rec R[kTile];
  ChunkProd X;
  load_tile((se - 1) & ~(kTile - 1), R, X);
  for (int tb = (se - 1) & ~(kTile - 1); tb >= sb; tb -= kTile) {
    int te = min(se, tb + kTile);
    Rec Rn[kTile];
    ChunkProd Xn;
    load_tile(max(tb - kTile, sb), Rn, Xn);  // (the chunk's first tile once more at the end: never used)
    double f0 = a0, f1 = a1, f2 = a2;
    apply_chunk_fwd(f0, f1, f2, X);
    norm_fast(f0, f1, f2);
    if (col.general && own_distance(R)) {
      Rec E[kTile];
      rec_lane_all(R, E);
      dec_tile<DR, 2, nrow>(C, E, f0, f1, f2, tb, te, b0, b1, b2, tile);
    } else {
      int tier = group_tier<kTile>(C, R);
      if (tier == 0) dec_tile<DR, 0, nrow>(C, R, f0, f1, f2, tb, te, b0, b1, b2, tile);
      else if (tier == 1) dec_tile<DR, 1, nrow>(C, R, f0, f1, f2, tb, te, b0, b1, b2, tile);
      else dec_tile<DR, 2, nrow>(C, R, f0, f1, f2, tb, te, b0, b1, b2, tile);
    }
    __syncthreads();
    // rows x 64 x kTile values; consecutive threads write consecutive SNPs of one pair row (128-byte pieces).  Output
    // row rr of a pair is tile row rr (the reference's order) or 2 - rr (the writer's order: IBD0 first); rows that
    // are not asked for are not visited.
#pragma unroll 12
    for (int j = 0; j < nrow * kTile; j++) {
      const int idx = j * kBwdThreads + threadIdx.x;
      const int e = idx % kTile, pr = (idx / kTile) % kBwdThreads, rr = idx / (kTile * kBwdThreads);
      const int s = tb + e, pp = pbase + pr;
      const double v = tile[WORDER ? rr : 2 - rr][pr][e];  // the reference's order is IBD2 first
      if (s < te && pp < P) {
        if constexpr (sizeof(OutT) == sizeof(double)) post[((size_t)pp * nrow + rr) * M + s] = v;
        else post[((size_t)pp * nrow + rr) * M + s] = (OutT)(sig4 ? round_sig4(v) : v);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kTile; u++) R[u] = Rn[u];
    X = Xn;
  }
}

// K7 Viterbi (:1289-1420), log domain as the reference (max3 / which_max tie rules kept, and the start quirk:
// every state starts from max(log k) + its emission, :1356-1359).  bp: M x P bytes (3 x 2-bit back pointers).
// The transition matrix is formed the reference's way (exp, constants, fix_underflow) because the tie rules act
// on its logarithms.
__device__ inline int vwhich3(double a, double b, double c) {
  if (a > b && a > c) return 0;
  else if (b > a && b > c) return 1;
  return 2;
}
// max3 (a > b && a > c ? a : b > a && b > c ? b : c) and which_max of the same three sums at once, without branches
__device__ __forceinline__ double vpick3(double a, double b, double c, int &which) {
  const bool fa = a > b && a > c, fb = b > a && b > c;
  which = fa ? 0 : (fb ? 1 : 2);
  return fa ? a : (fb ? b : c);
}

struct RefTrans {
  double na, nak1, k0, k1, k2, c02a, c02b, c20a, c20b, c20c, r10, r12;
  int k1_is_one, dr;
};
__device__ inline RefTrans make_ref_trans(double a, double k2, double k1, double k0, int double_recom) {
  RefTrans T;
  T.na = -a; T.nak1 = -a * k1;
  T.k0 = k0; T.k1 = k1; T.k2 = k2;
  T.k1_is_one = (k1 == 1);
  T.dr = double_recom;
  double km1 = k1 - 1;
  T.c02a = k2 / km1;
  T.c02b = k1 + k0 * k1 / km1;
  T.c20a = k0 / km1;
  T.c20b = k1 + k2 * k1 / km1;
  T.c20c = (1 - k1) * k0 / km1;
  if (k1 == 0) { T.r10 = 1; T.r12 = 1; } else { T.r10 = k0 / k1; T.r12 = k2 / k1; }
  return T;
}
__device__ inline void ref_trans_at(const RefTrans &C, double t, double (&T)[3][3]) {
  double e1 = exp(C.na * t);
  if (C.dr) {
    double o = 1.0 - e1;
    double b0 = o * C.k0, b1 = o * C.k1, b2 = o * C.k2;
    T[0][0] = b0 + e1; T[0][1] = b1; T[0][2] = b2;
    T[1][0] = b0; T[1][1] = b1 + e1; T[1][2] = b2;
    T[2][0] = b0; T[2][1] = b1; T[2][2] = b2 + e1;
    return;
  }
  double e2 = exp(C.nak1 * t);
  double t01 = (1.0 - e1) * C.k1;
  double t02 = 0, t20 = 0;
  if (!C.k1_is_one) {
    t02 = e2 * C.c02a + e1 * C.c02b + C.k2;
    t20 = e2 * C.c20a + e1 * C.c20b - C.c20c;
    if (t02 < 1e-15) t02 = 0;  // fix_underflow, :542-547
    if (t20 < 1e-15) t20 = 0;
  }
  double t10 = t01 * C.r10, t12 = t01 * C.r12;
  T[0][1] = t01; T[0][2] = t02; T[0][0] = (-t01 - t02) + 1;
  T[1][0] = t10; T[1][2] = t12; T[1][1] = (-t10 - t12) + 1;
  T[2][1] = t01; T[2][0] = t20; T[2][2] = (-t01 - t20) + 1;
}

// Back pointers of the Viterbi: one byte per pair and SNP (three two-bit choices, 0xFF = SNP dropped for the pair), and
// the state the path ends in as the byte of "SNP" M.  Pairs go in groups of kVitPairs, and within a group four SNPs of
// a pair lie together: [group][SNP / 4][pair % 16][SNP % 4] -- the walking wave of the forward kernel stores a dword per
// pair and four SNPs (64 B per store for sixteen pairs), the backward kernel reads a stretch of a group as one run.
static constexpr int kVitPairs = 16;
__host__ __device__ inline size_t vit_bp_quads(int M) { return (size_t)M / 4 + 1; }
__host__ __device__ inline size_t vit_bp_bytes(int M, int P) {
  return (size_t)((P + kVitPairs - 1) / kVitPairs) * vit_bp_quads(M) * 64;
}
__host__ __device__ inline size_t vit_bp_off(int p, int s, int M) {
  return ((size_t)(p / kVitPairs) * vit_bp_quads(M) + (s >> 2)) * 64 + (p % kVitPairs) * 4 + (s & 3);
}

// One value of a lane's quad (lanes 4q .. 4q + 3) to all four of its lanes.
template <int K>
__device__ __forceinline__ double quad_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, K * 0x55, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, K * 0x55, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true);
}

// A workgroup takes NP pairs.  The expensive part of a step -- two exponentials and twelve logarithms, none of which
// depends on the chain -- is done by 256 threads for a tile of 256 / NP SNPs at once (thread = one pair and one SNP of
// the tile; the records of the next tile are already on their way) and left in LDS.  A fifth wave walks the tile
// before, a quad of lanes per pair: lane y of the quad holds the state y the step arrives in -- three adds, the max3 /
// which_max rules, the emission -- and the quad's three new values go round by DPP; lane 0 writes the back-pointer
// byte of the SNP (0xFF = SNP dropped for the pair).  Two tile buffers, so the walk over tile t - 1 runs beside the
// logarithms of tile t.  The same library calls and the same sums in the same order as one lane per pair would make,
// so the values, the ties and the paths are the same.  A workgroup's time is its walk along the row whatever NP is:
// a short pair list takes NP = 4 (tiles of 64 SNPs, four times the workgroups).  bp: vit_bp_off() above; the end state
// is 0xFF for a pair without a kept SNP.
static constexpr int kVitThreads = 256 + 64;
template <int NP>
__global__ void __launch_bounds__(kVitThreads)
viterbi_fwd_kernel(const double *__restrict__ rec, int M, int P, int RP, const double *__restrict__ params,
                   int double_recom, uint8_t *__restrict__ bp) {  // rec: the FULL format (t, S0, S1, S2 planes)
  static_assert(NP == 4 || NP == 8 || NP == 16, "a quad of the walking wave per pair");
  constexpr int kTile = 256 / NP;
  // per SNP, pair and state y arrived in: {log S_y, log T[0][y], log T[1][y], log T[2][y]} -- what lane y of the quad reads
  __shared__ __align__(16) double lg[2][kTile][NP][3][4];
  __shared__ unsigned char dropped[2][kTile][NP];
  const int tid = threadIdx.x, wl = tid - 256;
  const bool producer = tid < 256;
  const bool walker = !producer && wl < 4 * NP;
  const int pp = producer ? tid % NP : (wl >> 2) % NP, ss = tid / NP;
  const int y = wl & 3, yy = y < 3 ? y : 2;  // the fourth lane of a quad shadows the third; nothing of it is kept
  const int p = blockIdx.x * NP + pp;
  const int pc = p < P ? p : P - 1;  // dead lanes shadow a valid pair and write nothing
  const double a = params[3 * pc], k2 = params[3 * pc + 1], k1 = params[3 * pc + 2];
  const double k0 = 1 - k2 - k1;
  const RefTrans C = make_ref_trans(a, k2, k1, k0, double_recom);
  const double mx = fmax(log(k0), fmax(log(k1), log(k2)));
  double v0 = 0, v1 = 0, v2 = 0, vy = 0;  // the walking lanes' chain; vy: the lane's own component
  bool started = false;
  const int ntiles = (M + kTile - 1) / kTile;
  double cur[4] = {-1, 0, 0, 0}, nxt[4] = {-1, 0, 0, 0};
  if (producer && ss < M) {
    const double *row = rec + rec_off(ss, pc, kRecPlanesFull, RP);
    cur[0] = row[0]; cur[1] = row[128]; cur[2] = row[256]; cur[3] = row[384];
  }
  for (int t = 0; t <= ntiles; t++) {
    if (producer) {
      const int s = t * kTile + ss;
      if (s + kTile < M) {
        const double *row = rec + rec_off(s + kTile, pc, kRecPlanesFull, RP);
        nxt[0] = row[0]; nxt[1] = row[128]; nxt[2] = row[256]; nxt[3] = row[384];
      }
      if (s < M) {
        const int w = t & 1;
        const double rt = cur[0];
        const bool dr = __builtin_signbit(rt);
        dropped[w][ss][pp] = dr ? 1 : 0;
        if (!dr) {
          double T[3][3];
          ref_trans_at(C, rt, T);
#pragma unroll
          for (int j = 0; j < 3; j++) {
            lg[w][ss][pp][j][0] = log(cur[1 + j]);
#pragma unroll
            for (int x = 0; x < 3; x++) lg[w][ss][pp][j][1 + x] = log(T[x][j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) cur[j] = nxt[j];
    } else if (walker && t > 0) {
      const int r = (t - 1) & 1, s0 = (t - 1) * kTile;
      const int n = min(kTile, M - s0);
      uint8_t *out = bp + vit_bp_off(pc, s0, M);  // s0 is a multiple of four: byte u of the tile is out[(u / 4) * 64 + u % 4]
      const bool writes = y == 0 && p < P;
      unsigned quad = 0;  // the lane's part of the bytes of four SNPs, stored together
      const unsigned f1 = y < 3 ? 1u << (2 * yy) : 0u, f2 = y < 3 ? 2u << (2 * yy) : 0u;
      struct In { double e, L0, L1, L2; int dr; };
      auto fetch = [&](int u) {
        const double *q = &lg[r][u][pp][yy][0];
        return In{q[0], q[1], q[2], q[3], dropped[r][u][pp]};
      };
      // general: the chain of a pair may still have to start (its first kept SNP takes max log k + log S)
      auto step = [&](int u, const In &in, auto general) {
        const bool kept = !in.dr;
        // the lane's two bits of the byte (none from a quad's fourth lane): which_max of the same three sums, in place
        const double sa = in.L0 + v0, sb = in.L1 + v1, sc = in.L2 + v2;
        const bool fa = sa > sb && sa > sc, fb = sb > sa && sb > sc;
        double nv = in.e + (fa ? sa : (fb ? sb : sc));
        unsigned field = fa ? 0u : (fb ? f1 : f2);
        if (decltype(general)::value) {
          if (!started) { nv = mx + in.e; field = 0; }
          started = started || kept;
        }
        field = kept ? field : 0xFFu;
        vy = kept ? nv : vy;
        v0 = quad_bcast<0>(vy); v1 = quad_bcast<1>(vy); v2 = quad_bcast<2>(vy);
        // the quad's three fields meet at the store
        if (decltype(general)::value) {
          const unsigned byte = quad_bcast<0>((int)field) | quad_bcast<1>((int)field) | quad_bcast<2>((int)field);
          if (writes) out[(u >> 2) * 64 + (u & 3)] = (uint8_t)byte;
        } else {
          quad |= field << (8 * (u & 3));
          if ((u & 3) == 3) {
            const unsigned all = quad_bcast<0>((int)quad) | quad_bcast<1>((int)quad) | quad_bcast<2>((int)quad);
            if (writes) *reinterpret_cast<unsigned *>(out + (u >> 2) * 64) = all;
            quad = 0;
          }
        }
      };
      if (n == kTile && __all(started)) {
        In c = fetch(0);  // the LDS reads of a step are issued a step ahead of the chain
#pragma unroll 4
        for (int u = 0; u < kTile; u++) {
          const In nx = fetch(u + 1 < kTile ? u + 1 : u);
          step(u, c, std::false_type());
          c = nx;
        }
      } else {
        for (int u = 0; u < n; u++) step(u, fetch(u), std::true_type());
      }
    }
    __syncthreads();  // tile t is complete, tile t - 1 is consumed: its buffer is the one tile t + 1 goes to
  }
  if (walker && y == 0 && p < P) bp[vit_bp_off(p, M, M)] = started ? (uint8_t)vwhich3(v0, v1, v2) : (uint8_t)0xFF;
}

// The walk back along the pointers, kVitPairs pairs per workgroup and kVitBack SNPs at a time: the stretch of the
// group's pointers is one run of memory (16 B a thread into LDS).  A back-pointer byte is a map of the three states
// (the identity where the SNP is dropped) and maps compose exactly, so the stretch of a pair is cut into 16 segments
// of 16 SNPs: every thread composes the map of one segment of one pair, one lane per pair takes the state through the
// 16 segment maps, and every thread walks its segment again from the state it is entered in -- 48 steps in a row
// instead of 256.  Every thread then writes one SNP of the stretch of every pair's row (path: P x M, -1 where the SNP
// is dropped for the pair).
static constexpr int kVitBack = 256, kVitSeg = 16;
__global__ void __launch_bounds__(kVitBack)
viterbi_back_kernel(const uint8_t *__restrict__ bp, int M, int P, int8_t *__restrict__ path) {
  static_assert(kVitBack == kVitPairs * kVitSeg && kVitBack / kVitSeg == 16, "a thread per pair and segment of 16 SNPs");
  __shared__ __align__(16) unsigned char pt[kVitBack * kVitPairs];  // [SNP / 4][pair][SNP % 4], as in memory
  __shared__ __align__(4) signed char out[kVitPairs][kVitBack + 4];
  __shared__ unsigned char maps[kVitSeg][kVitPairs];
  __shared__ signed char entry[kVitSeg][kVitPairs];
  const int tid = threadIdx.x, j = tid % kVitPairs, g = tid / kVitPairs;
  const int p0 = blockIdx.x * kVitPairs, np = min(kVitPairs, P - p0);
  int cur = 0;  // lanes 0 .. 15: the state of pair `tid` at the lowest SNP walked so far
  if (tid < np) {
    uint8_t e = bp[vit_bp_off(p0 + tid, M, M)];
    cur = e == 0xFF ? -1 : (int)e;
  }
  const int top = (M - 1) / kVitBack * kVitBack;
  auto stretch = [&](int lo) {  // the thread's 16 B of the stretch that starts at SNP lo (whole quads: the buffer holds them)
    const int n = min(kVitBack, M - lo);
    return tid * 16 < (n + 3) / 4 * 64 ? reinterpret_cast<const uint4 *>(bp + vit_bp_off(p0, lo, M))[tid] : uint4{0, 0, 0, 0};
  };
  uint4 ahead = stretch(top);
  for (int lo = top; lo >= 0; lo -= kVitBack) {
    const int n = min(kVitBack, M - lo);
    reinterpret_cast<uint4 *>(pt)[tid] = ahead;
    __syncthreads();
    if (lo > 0) ahead = stretch(lo - kVitBack);  // on its way during the walk
    // the thread's segment: SNPs 16 g .. 16 g + 15 of pair j, four dwords of four SNPs
    const unsigned *pt32 = reinterpret_cast<const unsigned *>(pt);
    unsigned wd[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      wd[i] = pt32[(4 * g + i) * kVitPairs + j];
      if (n < kVitBack) {  // past the end of the rows: whatever the byte holds, it is no SNP
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (16 * g + 4 * i + k >= n) wd[i] |= 0xFFu << (8 * k);
      }
    }
    unsigned f = 0x24;  // the map so far, from the segment's top downwards: f(x) in bits 2x, 2x + 1; identity to begin with
#pragma unroll
    for (int i = 3; i >= 0; i--)
#pragma unroll
      for (int k = 3; k >= 0; k--) {
        const unsigned b = (wd[i] >> (8 * k)) & 0xFF;
        const unsigned nf = ((b >> (2 * (f & 3))) & 3) | (((b >> (2 * ((f >> 2) & 3))) & 3) << 2) |
                            (((b >> (2 * ((f >> 4) & 3))) & 3) << 4);
        f = b == 0xFF ? f : nf;
      }
    maps[g][j] = (unsigned char)f;
    __syncthreads();
    if (tid < kVitPairs) {
#pragma unroll
      for (int q = kVitSeg - 1; q >= 0; q--) {
        entry[q][tid] = (signed char)cur;
        cur = cur < 0 ? cur : (int)((maps[q][tid] >> (2 * cur)) & 3);
      }
    }
    __syncthreads();
    int c = entry[g][j];
    unsigned *out32 = reinterpret_cast<unsigned *>(&out[j][0]);
#pragma unroll
    for (int i = 3; i >= 0; i--) {
      unsigned o = 0;
#pragma unroll
      for (int k = 3; k >= 0; k--) {
        const unsigned b = (wd[i] >> (8 * k)) & 0xFF;
        const bool dr = b == 0xFF;
        o |= (dr ? 0xFFu : (unsigned)c & 0xFFu) << (8 * k);
        c = dr ? c : (int)((b >> (2 * (c & 3))) & 3);
      }
      out32[4 * g + i] = o;
    }
    __syncthreads();
    if (tid < n)
      for (int q = 0; q < np; q++) path[(size_t)(p0 + q) * M + lo + tid] = out[q][tid];
    __syncthreads();
  }
}

}  // namespace relate
