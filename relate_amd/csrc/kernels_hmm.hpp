// HMM kernels over staged emission records: batched forward likelihood (K5, like()), forward-backward
// posterior decoding (K6, decodek2()/decode()) and the Viterbi path (K7).  Replaces the per-call recursions of
// src/relateHMM.cpp:1426-1544, 1012-1263, 1289-1420.  gfx950 only; FP64 VALU work, no MFMA (a 3x3 transition
// with per-lane parameters is not a dense contraction).
//
// Numerics.  The reference walks the SNPs in log space (3 exp + 6 log per step).  Here the recursion runs in the
// linear domain and is renormalised by exact powers of two (v_frexp_exp / v_ldexp) every kRescale steps; the
// exponent is accumulated as an integer and one log is taken at the end.  This reproduces like() to ~1e-15
// relative (tests/test_gpu_parity.py).  The two exp() per step for e^{-a t}, e^{-a k1 t} are inherent: a is per
// lane, t per SNP.
//
// Parallelism.  One lane = one (pair, parameter point).  A pair block at M = 5e5 SNPs holds only a few thousand
// pairs (32 B per pair and SNP), far too few lanes for 256 CUs, so the SNP axis is cut into C chunks: the
// per-step matrices A_s = T(t_s) diag(S_s) compose associatively, chunk_matrix_kernel forms every chunk's
// product (three row vectors with their own power-of-two exponents) in parallel, and a short combine kernel
// folds the C products per lane.  With C = 1 the plain single-vector kernels are used.  Threads are laid out
// chunk-major (tid = chunk * nlanes + lane): a wave reads the records of consecutive lanes of one SNP.
//
// Memory.  rec[s * P + p] is 32 B; every thread streams its column with a software prefetch of kUnroll records.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_emis.hpp"

namespace relate {

static constexpr int kUnroll = 4;    // records in flight per thread
static constexpr int kRescale = 8;   // steps between power-of-two renormalisations (emissions >~ 1e-12 per step)
static constexpr double kLn2 = 0.6931471805599453094;

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

// v <- (v T) o S : one forward step of a row vector
__device__ inline void fwd_step(double &f0, double &f1, double &f2, const double (&T)[3][3], const EmisRec &r) {
  double n0 = (T[0][0] * f0 + T[1][0] * f1 + T[2][0] * f2) * r.s0;
  double n1 = (T[0][1] * f0 + T[1][1] * f1 + T[2][1] * f2) * r.s1;
  double n2 = (T[0][2] * f0 + T[1][2] * f1 + T[2][2] * f2) * r.s2;
  f0 = n0; f1 = n1; f2 = n2;
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

__device__ inline bool lane_params(const double *__restrict__ params, int l, double alim1, bool reject, double &a,
                                   double &k2, double &k1, double &k0) {
  a = params[3 * l]; k2 = params[3 * l + 1]; k1 = params[3 * l + 2];
  k0 = 1 - (k1 + k2);
  return !(reject && (k2 < 0 || k1 < 0 || k0 < 0 || k2 > 1 || a > alim1));  // like()'s rejection, :1436
}

// Record markers written by stage_pairs_kernel: t == kDropped: SNP dropped for the pair; t == kFirst: first kept
// SNP of the pair (the recursion starts here from k o S).
static constexpr double kDropped = -1.0, kFirst = -2.0;

// K5, C = 1.  out[l] = -log-likelihood, SYS_MAX for a rejected point, NaN/inf if the recursion broke down.
__global__ void __launch_bounds__(256)
like_kernel(const EmisRec *__restrict__ rec, int M, int P, const int32_t *__restrict__ lane_pair,
            const double *__restrict__ params, int nlanes, double alim1, int double_recom,
            double *__restrict__ out) {
  int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlanes) return;
  double a, k2, k1, k0;
  if (!lane_params(params, l, alim1, true, a, k2, k1, k0)) { out[l] = 10000000.0; return; }
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  const EmisRec *col = rec + lane_pair[l];
  double f0 = 0, f1 = 0, f2 = 0;
  long esum = 0;
  bool started = false;
  EmisRec nx[kUnroll];
#pragma unroll
  for (int u = 0; u < kUnroll; u++)
    if (u < M) nx[u] = col[(size_t)u * P];
  for (int s0 = 0; s0 < M; s0 += kUnroll) {
    EmisRec cur[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) cur[u] = nx[u];
#pragma unroll
    for (int u = 0; u < kUnroll; u++)
      if (s0 + kUnroll + u < M) nx[u] = col[(size_t)(s0 + kUnroll + u) * P];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      if (s0 + u >= M) break;
      const EmisRec &r = cur[u];
      if (r.t == kDropped) continue;
      if (r.t == kFirst) {
        f0 = k0 * r.s0; f1 = k1 * r.s1; f2 = k2 * r.s2;
        started = true;
      } else {
        double T[3][3];
        trans_at(C, r.t, T);
        fwd_step(f0, f1, f2, T, r);
      }
    }
    if (((s0 / kUnroll) & (kRescale / kUnroll - 1)) == 0) esum += rescale3(f0, f1, f2);
  }
  esum += rescale3(f0, f1, f2);
  double tot = f0 + f1 + f2;
  out[l] = started ? -(log(tot) + (double)esum * kLn2) : nan("");
}

// Chunk products.  For thread (chunk c, lane l): rows X[i][.] (i = unit start vector) with exponents e[i]:
//   alpha_end[y] = sum_i alpha_start[i] * X[i][y] * 2^{e[i]}        beta_start[i] = 2^{e[i]} sum_y X[i][y] beta_end[y]
// flag 0: matrix; 1: the chunk contains the pair's first kept SNP, row 0 is the forward VECTOR at the chunk end;
// 2: no kept SNP in the chunk (identity).  Arrays are struct-of-arrays over n = C * nlanes threads.
struct ChunkOut {
  double *m;     // 9 x n
  int *e;        // 3 x n
  int *flag;     // n
};

__global__ void __launch_bounds__(256)
chunk_matrix_kernel(const EmisRec *__restrict__ rec, int M, int P, const int32_t *__restrict__ lane_pair,
                    const double *__restrict__ params, int nlanes, int nchunk, int chunk_len, double alim1,
                    int double_recom, int reject, ChunkOut O) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long n = (long)nlanes * nchunk;
  if (tid >= n) return;
  int l = (int)(tid % nlanes), c = (int)(tid / nlanes);
  double a, k2, k1, k0;
  if (!lane_params(params, l, alim1, reject != 0, a, k2, k1, k0)) { O.flag[tid] = 2; return; }
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  const EmisRec *col = rec + (lane_pair ? lane_pair[l] : l);
  int sb = c * chunk_len, se = min(M, sb + chunk_len);
  double x[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  int ex[3] = {0, 0, 0};
  int flag = 2;
  EmisRec nx[kUnroll];
#pragma unroll
  for (int u = 0; u < kUnroll; u++)
    if (sb + u < se) nx[u] = col[(size_t)(sb + u) * P];
  int it = 0;
  for (int s0 = sb; s0 < se; s0 += kUnroll, it++) {
    EmisRec cur[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) cur[u] = nx[u];
#pragma unroll
    for (int u = 0; u < kUnroll; u++)
      if (s0 + kUnroll + u < se) nx[u] = col[(size_t)(s0 + kUnroll + u) * P];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      if (s0 + u >= se) break;
      const EmisRec &r = cur[u];
      if (r.t == kDropped) continue;
      if (r.t == kFirst) {
        x[0][0] = k0 * r.s0; x[0][1] = k1 * r.s1; x[0][2] = k2 * r.s2;
        ex[0] = 0;
        flag = 1;
      } else {
        double T[3][3];
        trans_at(C, r.t, T);
        fwd_step(x[0][0], x[0][1], x[0][2], T, r);
        if (flag != 1) {
          fwd_step(x[1][0], x[1][1], x[1][2], T, r);
          fwd_step(x[2][0], x[2][1], x[2][2], T, r);
          flag = 0;
        }
      }
    }
    if ((it & (kRescale / kUnroll - 1)) == 0) {
      ex[0] += rescale3(x[0][0], x[0][1], x[0][2]);
      if (flag == 0) {
        ex[1] += rescale3(x[1][0], x[1][1], x[1][2]);
        ex[2] += rescale3(x[2][0], x[2][1], x[2][2]);
      }
    }
  }
  ex[0] += rescale3(x[0][0], x[0][1], x[0][2]);
  if (flag == 0) {
    ex[1] += rescale3(x[1][0], x[1][1], x[1][2]);
    ex[2] += rescale3(x[2][0], x[2][1], x[2][2]);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int y = 0; y < 3; y++) O.m[(size_t)(i * 3 + y) * n + tid] = x[i][y];
    O.e[(size_t)i * n + tid] = ex[i];
  }
  O.flag[tid] = flag;
}

// alpha <- alpha * (chunk product); keeps alpha scaled, returns the exponent moved out
__device__ inline int apply_chunk_fwd(double &f0, double &f1, double &f2, const ChunkOut &O, long n, long at) {
  double a[3] = {f0, f1, f2};
  int e[3] = {O.e[at], O.e[n + at], O.e[2 * n + at]};
  int emax = -2147483647;
  for (int i = 0; i < 3; i++)
    if (a[i] > 0 && e[i] > emax) emax = e[i];
  if (emax == -2147483647) { f0 = f1 = f2 = 0; return 0; }
  double r0 = 0, r1 = 0, r2 = 0;
  for (int i = 0; i < 3; i++) {
    if (!(a[i] > 0)) continue;
    double w = ldexp(a[i], e[i] - emax);
    r0 += w * O.m[(size_t)(i * 3 + 0) * n + at];
    r1 += w * O.m[(size_t)(i * 3 + 1) * n + at];
    r2 += w * O.m[(size_t)(i * 3 + 2) * n + at];
  }
  f0 = r0; f1 = r1; f2 = r2;
  return emax;
}

// K5 combine: folds the C chunk products of every lane into -log-likelihood.
__global__ void __launch_bounds__(256)
like_combine_kernel(ChunkOut O, const double *__restrict__ params, int nlanes, int nchunk, double alim1,
                    double *__restrict__ out) {
  int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlanes) return;
  double a, k2, k1, k0;
  if (!lane_params(params, l, alim1, true, a, k2, k1, k0)) { out[l] = 10000000.0; return; }
  long n = (long)nlanes * nchunk;
  double f0 = 0, f1 = 0, f2 = 0;
  long esum = 0;
  bool started = false;
  for (int c = 0; c < nchunk; c++) {
    long at = (long)c * nlanes + l;
    int flag = O.flag[at];
    if (flag == 2) continue;
    if (flag == 1) {
      f0 = O.m[at]; f1 = O.m[n + at]; f2 = O.m[2 * n + at];
      esum = O.e[at];
      started = true;
    } else if (started) {
      esum += apply_chunk_fwd(f0, f1, f2, O, n, at);
      esum += rescale3(f0, f1, f2);
    }
  }
  double tot = f0 + f1 + f2;
  out[l] = started ? -(log(tot) + (double)esum * kLn2) : nan("");
}

// K6.  Normalised forward vectors go to fwd[s * P + p] (24 B); the boundary vectors of the chunked sweep to
// bnd: astart[(c * P + p) * 3 ..] = forward vector entering chunk c, bend[...] = backward vector at the last kept
// SNP of chunk c.
struct __attribute__((aligned(8))) Fwd3 {
  double f0, f1, f2;
};

__device__ inline void norm_sum(double &x0, double &x1, double &x2) {
  double inv = 1.0 / (x0 + x1 + x2);
  x0 *= inv; x1 *= inv; x2 *= inv;
}
__device__ inline void norm_max(double &x0, double &x1, double &x2) {
  double inv = 1.0 / fmax(x0, fmax(x1, x2));
  x0 *= inv; x1 *= inv; x2 *= inv;
}

__global__ void __launch_bounds__(256)
decode_combine_kernel(ChunkOut O, int P, int nchunk, double *__restrict__ astart, double *__restrict__ bend) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  long n = (long)P * nchunk;
  double f0 = 0, f1 = 0, f2 = 0;
  bool started = false;
  for (int c = 0; c < nchunk; c++) {
    long at = (long)c * P + p;
    astart[at * 3] = f0; astart[at * 3 + 1] = f1; astart[at * 3 + 2] = f2;
    int flag = O.flag[at];
    if (flag == 2) continue;
    if (flag == 1) {
      f0 = O.m[at]; f1 = O.m[n + at]; f2 = O.m[2 * n + at];
      started = true;
    } else if (started) {
      apply_chunk_fwd(f0, f1, f2, O, n, at);
    }
    if (started) norm_sum(f0, f1, f2);
  }
  double b0 = 1, b1 = 1, b2 = 1;
  for (int c = nchunk - 1; c >= 0; c--) {
    long at = (long)c * P + p;
    bend[at * 3] = b0; bend[at * 3 + 1] = b1; bend[at * 3 + 2] = b2;
    if (O.flag[at] != 0) continue;
    // beta_start[i] = 2^{e[i]} sum_y X[i][y] beta[y]
    double v[3];
    int e[3], emax = -2147483647;
    for (int i = 0; i < 3; i++) {
      v[i] = O.m[(size_t)(i * 3 + 0) * n + at] * b0 + O.m[(size_t)(i * 3 + 1) * n + at] * b1 +
             O.m[(size_t)(i * 3 + 2) * n + at] * b2;
      e[i] = O.e[(size_t)i * n + at];
      if (v[i] > 0 && e[i] > emax) emax = e[i];
    }
    if (emax == -2147483647) continue;
    b0 = v[0] > 0 ? ldexp(v[0], e[0] - emax) : 0;
    b1 = v[1] > 0 ? ldexp(v[1], e[1] - emax) : 0;
    b2 = v[2] > 0 ? ldexp(v[2], e[2] - emax) : 0;
    norm_max(b0, b1, b2);
  }
}

// forward sweep inside one chunk; thread = (chunk c, pair p); astart == nullptr means nchunk == 1
__global__ void __launch_bounds__(256)
decode_fwd_kernel(const EmisRec *__restrict__ rec, int M, int P, const double *__restrict__ params,
                  int double_recom, int nchunk, int chunk_len, const double *__restrict__ astart,
                  Fwd3 *__restrict__ fwd, int *__restrict__ nsnp) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)P * nchunk) return;
  int p = (int)(tid % P), c = (int)(tid / P);
  double a, k2, k1, k0;
  lane_params(params, p, 0, false, a, k2, k1, k0);
  k0 = 1 - k2 - k1;  // decodek2, :1018
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  int sb = c * chunk_len, se = min(M, sb + chunk_len);
  double f0 = 0, f1 = 0, f2 = 0;
  if (astart) { f0 = astart[tid * 3]; f1 = astart[tid * 3 + 1]; f2 = astart[tid * 3 + 2]; }
  const EmisRec *col = rec + p;
  int m = 0;
  EmisRec nx[kUnroll];
#pragma unroll
  for (int u = 0; u < kUnroll; u++)
    if (sb + u < se) nx[u] = col[(size_t)(sb + u) * P];
  for (int s0 = sb; s0 < se; s0 += kUnroll) {
    EmisRec cur[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) cur[u] = nx[u];
#pragma unroll
    for (int u = 0; u < kUnroll; u++)
      if (s0 + kUnroll + u < se) nx[u] = col[(size_t)(s0 + kUnroll + u) * P];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      if (s0 + u >= se) break;
      const EmisRec &r = cur[u];
      if (r.t == kDropped) continue;
      m++;
      if (r.t == kFirst) {
        f0 = k0 * r.s0; f1 = k1 * r.s1; f2 = k2 * r.s2;
      } else {
        double T[3][3];
        trans_at(C, r.t, T);
        fwd_step(f0, f1, f2, T, r);
      }
      norm_sum(f0, f1, f2);
      Fwd3 o = {f0, f1, f2};
      fwd[(size_t)(s0 + u) * P + p] = o;
    }
  }
  if (nsnp) atomicAdd(&nsnp[p], m);
}

// backward sweep inside one chunk + posterior.  post is P x 3 x M (pair-major, as the host writer wants it):
// row0 IBD2, row1 IBD1, row2 IBD0 (decodek2, :1133-1137); -1 where the SNP is dropped (.post writer, :126-133).
// posterior_x(s) = fwd_x(s) bwd_x(s) / sum_y fwd_y(s) bwd_y(s): every scale factor cancels.
__global__ void __launch_bounds__(256)
decode_bwd_kernel(const EmisRec *__restrict__ rec, int M, int P, const double *__restrict__ params,
                  int double_recom, int nchunk, int chunk_len, const double *__restrict__ bend,
                  const Fwd3 *__restrict__ fwd, double *__restrict__ post) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)P * nchunk) return;
  int p = (int)(tid % P), c = (int)(tid / P);
  double a, k2, k1, k0;
  lane_params(params, p, 0, false, a, k2, k1, k0);
  k0 = 1 - k2 - k1;
  Trans C = make_trans(a, k2, k1, k0, double_recom);
  int sb = c * chunk_len, se = min(M, sb + chunk_len);
  double *o2 = post + (size_t)p * 3 * M, *o1 = o2 + M, *o0 = o1 + M;
  double b0 = 1, b1 = 1, b2 = 1;
  if (bend) { b0 = bend[tid * 3]; b1 = bend[tid * 3 + 1]; b2 = bend[tid * 3 + 2]; }
  const EmisRec *col = rec + p;
  // emission and distance of the next kept SNP (the one the backward vector currently refers to)
  double nt = 0, ns0 = 0, ns1 = 0, ns2 = 0;
  bool have_next = false;
  EmisRec nx[kUnroll];
#pragma unroll
  for (int u = 0; u < kUnroll; u++)
    if (se - 1 - u >= sb) nx[u] = col[(size_t)(se - 1 - u) * P];
  for (int s0 = se - 1; s0 >= sb; s0 -= kUnroll) {
    EmisRec cur[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) cur[u] = nx[u];
#pragma unroll
    for (int u = 0; u < kUnroll; u++)
      if (s0 - kUnroll - u >= sb) nx[u] = col[(size_t)(s0 - kUnroll - u) * P];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      int s = s0 - u;
      if (s < sb) break;
      const EmisRec &r = cur[u];
      if (r.t == kDropped) {
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
        b0 = n0; b1 = n1; b2 = n2;
        norm_max(b0, b1, b2);
      }
      Fwd3 f = fwd[(size_t)s * P + p];
      double p0 = f.f0 * b0, p1 = f.f1 * b1, p2 = f.f2 * b2;
      double inv = 1.0 / (p0 + p1 + p2);
      o0[s] = p0 * inv; o1[s] = p1 * inv; o2[s] = p2 * inv;
      nt = r.t; ns0 = r.s0; ns1 = r.s1; ns2 = r.s2;
      have_next = true;
    }
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
  for (int s = 0; s < M; s++) {
    EmisRec r = rec[(size_t)s * P + p];
    if (r.t == kDropped) continue;
    double l0 = log(r.s0), l1 = log(r.s1), l2 = log(r.s2);
    if (r.t == kFirst) {
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
  }
  int8_t *out = path + (size_t)p * M;
  int cur = started ? vwhich3(v0, v1, v2) : -1;
  for (int s = M - 1; s >= 0; s--) {
    if (rec[(size_t)s * P + p].t == kDropped) { out[s] = -1; continue; }
    out[s] = (int8_t)cur;
    cur = (bp[(size_t)s * P + p] >> (2 * cur)) & 3;
  }
}

}  // namespace relate
