// HMM kernels over staged emission records: batched forward likelihood (K5, like()), forward-backward
// posterior decoding (K6, decodek2()/decode()) and the Viterbi path (K7).  Replaces the per-call recursions of
// src/relateHMM.cpp:1426-1544, 1012-1263, 1289-1420.  gfx950 only; FP64 VALU work, no MFMA (a 3x3 transition
// with per-lane parameters is not a dense contraction).
//
// The reference walks the SNPs in log space (3 exp + 6 log per step).  Here the recursion runs in the linear
// domain and is renormalised by exact powers of two (v_frexp_exp / v_ldexp), the exponent is accumulated as an
// integer and one log is taken at the end; this reproduces like() to ~1e-15 relative (tests/test_gpu_parity.py).
// The two exp() per step for the transition terms e^{-a t}, e^{-a k1 t} are inherent: a is per lane, t per SNP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_emis.hpp"

namespace relate {

// Per-lane transition coefficients, hoisted out of the SNP loop (src/relateHMM.cpp:1440-1490).
struct Trans {
  double na, nak1;           // -a, -a*k1
  double k0, k1, k2;
  double c02a, c02b;         // T02 = e2*c02a + e1*c02b + k2      (zero when k1 == 1)
  double c20a, c20b, c20c;   // T20 = e2*c20a + e1*c20b - c20c    (zero when k1 == 1)
  double r10, r12;           // T10 = T01*r10, T12 = T01*r12      (both 1 when k1 == 0)
  int k1_is_one, dr;
};

__device__ inline Trans make_trans(double a, double k2, double k1, double k0, int double_recom) {
  Trans T;
  T.na = -a;
  T.nak1 = -a * k1;
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

// the nine transition probabilities T[x][y] (from state x to y) for inter-SNP distance t
__device__ inline void trans_at(const Trans &C, double t, double (&T)[3][3]) {
  double e1 = exp(C.na * t);
  if (C.dr) {  // dbl_recom, :501-507
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

// exact rescale of a 3-vector by a power of two; returns the exponent removed (0 if the vector is all zero)
__device__ inline int rescale3(double &x0, double &x1, double &x2) {
  double m = fmax(x0, fmax(x1, x2));
  if (!(m > 0)) return 0;
  int e = ilogb(m);
  x0 = ldexp(x0, -e);
  x1 = ldexp(x1, -e);
  x2 = ldexp(x2, -e);
  return e;
}

// K5.  One lane per (pair, parameter point): lane_pair[l] = column of the staged block, params[l] = (a,k2,k1).
// out[l] = -log-likelihood, RELATE_SYS_MAX for a rejected point (:1436), NaN if the recursion broke down.
__global__ void __launch_bounds__(256)
like_kernel(const EmisRec *__restrict__ rec, int M, int P, const int32_t *__restrict__ lane_pair,
            const double *__restrict__ params, int nlanes, double alim1, int double_recom,
            double *__restrict__ out) {
  int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlanes) return;
  double a = params[3 * l], k2 = params[3 * l + 1], k1 = params[3 * l + 2];
  double k0 = 1 - (k1 + k2);
  if (k2 < 0 || k1 < 0 || k0 < 0 || k2 > 1 || a > alim1) {
    out[l] = 10000000.0;
    return;
  }
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  const EmisRec *col = rec + lane_pair[l];
  double f0 = 0, f1 = 0, f2 = 0;
  long esum = 0;
  bool started = false;
  for (int s = 0; s < M; s++) {
    EmisRec r = col[(size_t)s * P];
    if (r.t < 0) continue;
    if (!started) {
      f0 = k0 * r.s0; f1 = k1 * r.s1; f2 = k2 * r.s2;
      started = true;
    } else {
      double T[3][3];
      trans_at(C, r.t, T);
      double n0 = (T[0][0] * f0 + T[1][0] * f1 + T[2][0] * f2) * r.s0;
      double n1 = (T[0][1] * f0 + T[1][1] * f1 + T[2][1] * f2) * r.s1;
      double n2 = (T[0][2] * f0 + T[1][2] * f1 + T[2][2] * f2) * r.s2;
      f0 = n0; f1 = n1; f2 = n2;
    }
    esum += rescale3(f0, f1, f2);
  }
  double tot = f0 + f1 + f2;
  out[l] = started ? -(log(tot) + (double)esum * 0.6931471805599453094) : nan("");
}

// K6 forward sweep.  One lane per pair of the block; params[p] = (a,k2,k1).  Writes the normalised forward
// vector (f0,f1,f2 summing to 1) of every kept SNP to fwd[s*P + p]; dropped SNPs are left untouched.
struct __attribute__((aligned(8))) Fwd3 {
  double f0, f1, f2;
};

__global__ void __launch_bounds__(256)
decode_fwd_kernel(const EmisRec *__restrict__ rec, int M, int P, const double *__restrict__ params,
                  int double_recom, Fwd3 *__restrict__ fwd, int *__restrict__ nsnp) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double a = params[3 * p], k2 = params[3 * p + 1], k1 = params[3 * p + 2];
  double k0 = 1 - k2 - k1;
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  double f0 = 0, f1 = 0, f2 = 0;
  bool started = false;
  int m = 0;
  for (int s = 0; s < M; s++) {
    EmisRec r = rec[(size_t)s * P + p];
    if (r.t < 0) continue;
    m++;
    if (!started) {
      f0 = k0 * r.s0; f1 = k1 * r.s1; f2 = k2 * r.s2;
      started = true;
    } else {
      double T[3][3];
      trans_at(C, r.t, T);
      double n0 = (T[0][0] * f0 + T[1][0] * f1 + T[2][0] * f2) * r.s0;
      double n1 = (T[0][1] * f0 + T[1][1] * f1 + T[2][1] * f2) * r.s1;
      double n2 = (T[0][2] * f0 + T[1][2] * f1 + T[2][2] * f2) * r.s2;
      f0 = n0; f1 = n1; f2 = n2;
    }
    double inv = 1.0 / (f0 + f1 + f2);
    f0 *= inv; f1 *= inv; f2 *= inv;
    Fwd3 o = {f0, f1, f2};
    fwd[(size_t)s * P + p] = o;
  }
  if (nsnp) nsnp[p] = m;
}

// K6 backward sweep + posterior.  post is P x 3 x M (pair-major, as the host writer wants it): row0 IBD2,
// row1 IBD1, row2 IBD0 (decodek2, :1133-1137); -1 where the SNP is dropped for the pair (.post writer, :126-133).
// posterior_x(s) = fwd_x(s) * bwd_x(s) / sum_y fwd_y(s) bwd_y(s): every scale factor cancels.
__global__ void __launch_bounds__(256)
decode_bwd_kernel(const EmisRec *__restrict__ rec, int M, int P, const double *__restrict__ params,
                  int double_recom, const Fwd3 *__restrict__ fwd, double *__restrict__ post) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double a = params[3 * p], k2 = params[3 * p + 1], k1 = params[3 * p + 2];
  double k0 = 1 - k2 - k1;
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  double *o2 = post + (size_t)p * 3 * M, *o1 = o2 + M, *o0 = o1 + M;
  double b0 = 1, b1 = 1, b2 = 1;
  // emission and distance of the next kept SNP (the one the backward vector currently refers to)
  double nt = 0, ns0 = 0, ns1 = 0, ns2 = 0;
  bool have_next = false;
  for (int s = M - 1; s >= 0; s--) {
    EmisRec r = rec[(size_t)s * P + p];
    if (r.t < 0) {
      o0[s] = -1; o1[s] = -1; o2[s] = -1;
      continue;
    }
    if (have_next) {  // beta_x(s) = sum_y T_xy(next) S_y(next) beta_y(next), :1116-1125
      double T[3][3];
      trans_at(C, nt, T);
      double q0 = ns0 * b0, q1 = ns1 * b1, q2 = ns2 * b2;
      double n0 = q2 * T[0][2] + q1 * T[0][1] + q0 * T[0][0];
      double n1 = q2 * T[1][2] + q1 * T[1][1] + q0 * T[1][0];
      double n2 = q2 * T[2][2] + q1 * T[2][1] + q0 * T[2][0];
      double inv = 1.0 / fmax(n0, fmax(n1, n2));
      b0 = n0 * inv; b1 = n1 * inv; b2 = n2 * inv;
    }
    Fwd3 f = fwd[(size_t)s * P + p];
    double p0 = f.f0 * b0, p1 = f.f1 * b1, p2 = f.f2 * b2;
    double inv = 1.0 / (p0 + p1 + p2);
    o0[s] = p0 * inv; o1[s] = p1 * inv; o2[s] = p2 * inv;
    nt = r.t; ns0 = r.s0; ns1 = r.s1; ns2 = r.s2;
    have_next = true;
  }
}

// K7 Viterbi (:1289-1420), log domain as the reference (max3 / which_max tie rules kept, and the start quirk:
// every state starts from max(log k) + its emission, :1356-1359).  bp: M x P bytes (3 x 2-bit back pointers).
__device__ inline double vmax3(double a, double b, double c) {
  if (a > b && a > c) return a;
  else if (b > a && b > c) return b;
  return c;
}
__device__ inline int vwhich3(double a, double b, double c) {
  if (a > b && a > c) return 0;
  else if (b > a && b > c) return 1;
  return 2;
}

__global__ void __launch_bounds__(256)
viterbi_kernel(const EmisRec *__restrict__ rec, int M, int P, const double *__restrict__ params, int double_recom,
               uint8_t *__restrict__ bp, int8_t *__restrict__ path) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double a = params[3 * p], k2 = params[3 * p + 1], k1 = params[3 * p + 2];
  double k0 = 1 - k2 - k1;
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  double v0 = 0, v1 = 0, v2 = 0;
  bool started = false;
  int last = -1;
  for (int s = 0; s < M; s++) {
    EmisRec r = rec[(size_t)s * P + p];
    if (r.t < 0) continue;
    double l0 = log(r.s0), l1 = log(r.s1), l2 = log(r.s2);
    if (!started) {
      double mx = fmax(log(k0), fmax(log(k1), log(k2)));
      v0 = mx + l0; v1 = mx + l1; v2 = mx + l2;
      started = true;
      bp[(size_t)s * P + p] = 0;
    } else {
      double T[3][3];
      trans_at(C, r.t, T);
      double L[3][3];
      for (int x = 0; x < 3; x++)
        for (int y = 0; y < 3; y++) L[x][y] = log(T[x][y]);
      double n0 = l0 + vmax3(L[0][0] + v0, L[1][0] + v1, L[2][0] + v2);
      double n1 = l1 + vmax3(L[0][1] + v0, L[1][1] + v1, L[2][1] + v2);
      double n2 = l2 + vmax3(L[0][2] + v0, L[1][2] + v1, L[2][2] + v2);
      int w0 = vwhich3(v0 + L[0][0], v1 + L[1][0], v2 + L[2][0]);
      int w1 = vwhich3(v0 + L[0][1], v1 + L[1][1], v2 + L[2][1]);
      int w2 = vwhich3(v0 + L[0][2], v1 + L[1][2], v2 + L[2][2]);
      bp[(size_t)s * P + p] = (uint8_t)(w0 | (w1 << 2) | (w2 << 4));
      v0 = n0; v1 = n1; v2 = n2;
    }
    last = s;
  }
  int8_t *out = path + (size_t)p * M;
  int cur = started ? vwhich3(v0, v1, v2) : -1;
  for (int s = M - 1; s >= 0; s--) {
    if (s > last || rec[(size_t)s * P + p].t < 0) { out[s] = -1; continue; }
    out[s] = (int8_t)cur;
    cur = (bp[(size_t)s * P + p] >> (2 * cur)) & 3;
  }
}

}  // namespace relate
