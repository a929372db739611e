// Pre-calculation kernels (once per data set): 2-bit packing, MAF counts (K1), LD window cells (K2).
// Replaces calcMaf (src/relateHMM.cpp:2985-3007) and snp_pair_range + get_geno_count + do_geno_cal
// (src/relateHMM.cpp:198-259, src/ld.cpp:342-569).  gfx950 only.
//
// Layout: genotypes are packed 2 bits per individual, SNP-major: row s holds W = ceil(N/32) uint64 words,
// individual n sits in word n/32 at bits 2*(n%32).  Code 0 = missing, 1/2/3 as in the reference.
// One SNP row of N = 2000 individuals is 504 B, so a whole LD window of rows stays in L1/L2 while the
// per-cell counts are formed with AND + popcount on 32 individuals at a time (HBM-bound integer work).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dd_math.hpp"

namespace relate {

static constexpr uint64_t kLoMask = 0x5555555555555555ull;

// raw: N x Mraw bytes (individual-major, as the reference's iMatrix); cols: the Msel raw column indices to
// keep (in order); out: Msel x W packed words.  One thread per (word, selected SNP).
__global__ void pack_kernel(const uint8_t *__restrict__ raw, int N, int Mraw, const int *__restrict__ cols, int Msel,
                            int W, uint64_t *__restrict__ out) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)Msel * W;
  if (tid >= total) return;
  int s = (int)(tid % Msel), w = (int)(tid / Msel);  // lanes walk consecutive SNP columns: coalesced byte reads
  int col = cols ? cols[s] : s;
  uint64_t word = 0;
  int n0 = w * 32;
  for (int b = 0; b < 32; b++) {
    int n = n0 + b;
    if (n >= N) break;
    uint64_t g = raw[(size_t)n * Mraw + col] & 3u;
    word |= g << (2 * b);
  }
  out[(size_t)s * W + w] = word;
}

// PLINK .bed ingestion without the reference's int matrix (bed_to_iMatrix, src/filereader_and_conversions.cpp:483-564).
// bed: the SNP-major payload after the 3-byte header, one row of bps = ceil(N/4) bytes per .bim line, four
// individuals per byte from the low bits up -- the same bit order as the packed layout here, so a 64-bit word of 32
// individuals is recoded with three logic operations.  PLINK code -> reference code: recode {1,0,2,3} followed by the
// 1 <-> 3 swap (:485, 530-533), i.e. 00 -> 3, 01 (missing) -> 0, 10 -> 2, 11 -> 1: out_hi = !lo, out_lo = !(hi ^ lo).
// cols: the Msel .bed rows to use, in analysis order (autosome filter doBimFile :634-640 and mysort src/asort.cpp:37-88
// are applied to the M-length position vectors by the caller).  One thread per (selected SNP, word).
__global__ void bed_recode_kernel(const uint8_t *__restrict__ bed, int bps, const int *__restrict__ cols, int Msel, int N,
                                  int W, uint64_t *__restrict__ out) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)Msel * W) return;
  int s = (int)(tid / W), w = (int)(tid % W);
  const uint8_t *row = bed + (size_t)cols[s] * bps;
  uint64_t x = 0;
  int b0 = w * 8;
#pragma unroll
  for (int b = 0; b < 8; b++)
    if (b0 + b < bps) x |= (uint64_t)row[b0 + b] << (8 * b);
  uint64_t lo = x & kLoMask, hi = (x >> 1) & kLoMask;
  uint64_t y = ((~lo & kLoMask) << 1) | (~(hi ^ lo) & kLoMask);
  int left = N - w * 32;  // individuals in this word; padding bits of the last byte are not individuals
  if (left < 32) y &= (((uint64_t)1 << (2 * left)) - 1);
  out[tid] = y;
}

// rows selected from an already packed matrix (pruning / MAF filter on the device copy)
__global__ void gather_rows_kernel(const uint64_t *__restrict__ in, const int *__restrict__ rows, int Msel, int W,
                                   uint64_t *__restrict__ out) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)Msel * W) return;
  int s = (int)(tid / W), w = (int)(tid % W);
  out[tid] = in[(size_t)rows[s] * W + w];
}

// unpack to N x M bytes (for the stripped dump the reference writes, src/relateHMM.cpp:3745)
__global__ void unpack_kernel(const uint64_t *__restrict__ in, int N, int M, int W, uint8_t *__restrict__ out) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)N * M) return;
  int n = (int)(tid / M), s = (int)(tid % M);
  out[tid] = (uint8_t)((in[(size_t)s * W + (n >> 5)] >> (2 * (n & 31))) & 3u);
}

// K1: calcMaf.  maf[s] = (sum(g-1)/n)/2 over non-missing genotypes.  One thread per SNP; a row is W words.
__global__ void maf_kernel(const uint64_t *__restrict__ G, int M, int W, double *__restrict__ maf) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const uint64_t *row = G + (size_t)s * W;
  int sum = 0, n = 0;
  for (int w = 0; w < W; w++) {
    uint64_t x = row[w];
    uint64_t lo = x & kLoMask, hi = (x >> 1) & kLoMask;
    n += __popcll(lo | hi);
    sum += __popcll(hi & ~lo) + 2 * __popcll(hi & lo);  // code 2 -> 1, code 3 -> 2
  }
  maf[s] = (((double)sum) / n) / 2;
}

// ---- one LD cell from its 3x3 count table: do_geno_cal (src/ld.cpp:342-525) --------------------------
struct LdCell {
  double pBA, pBa, pbA, pba, D, rsq2;  // snp_pair_range's names (src/relateHMM.cpp:222-242)
};

__device__ inline int quad_roots(double a, double b, double c, double *x0, double *x1) {  // src/ld.cpp:71-125
#pragma clang fp contract(off)  // keep +,-,*,/ and sqrt bit-identical to the host arithmetic of the reference
  double disc = b * b - 4 * a * c;
  if (a == 0) {
    if (b == 0) return 0;
    *x0 = -c / b;
    return 1;
  }
  if (disc > 0) {
    if (b == 0) {
      double r = fabs(0.5 * sqrt(disc) / a);
      *x0 = -r;
      *x1 = r;
    } else {
      double sgnb = (b > 0 ? 1 : -1);
      double temp = -0.5 * (b + sgnb * sqrt(disc));
      double r1 = temp / a, r2 = c / temp;
      if (r1 < r2) { *x0 = r1; *x1 = r2; } else { *x0 = r2; *x1 = r1; }
    }
    return 2;
  }
  if (disc == 0) {
    *x0 = -0.5 * b / a;
    *x1 = -0.5 * b / a;
    return 2;
  }
  return 0;
}

// acos / cos / pow: correctly rounded (dd_math.hpp), so that the roots carry the last bits the host libm gives the
// reference in all but ~0.1 % of calls -- getHighestId's strict comparison of LD measures is sensitive to them
__device__ inline int cubic_roots(double a, double b, double c, double *x) {  // src/ld.cpp:139-220
#pragma clang fp contract(off)
  double q = (a * a - 3 * b);
  double r = (2 * a * a * a - 9 * a * b + 27 * c);
  double Q = q / 9, R = r / 54;
  double Q3 = Q * Q * Q, R2 = R * R;
  double CR2 = 729 * r * r, CQ3 = 2916 * q * q * q;
  double xx = CR2 - CQ3;
  const double tol = 0.000000001;
  if (R == 0 && Q == 0) {
    x[0] = x[1] = x[2] = -a / 3;
    return 3;
  } else if (xx < tol && xx > -tol) {
    double sqrtQ = sqrt(Q);
    if (R > 0) {
      x[0] = -2 * sqrtQ - a / 3;
      x[1] = sqrtQ - a / 3;
      x[2] = sqrtQ - a / 3;
    } else {
      x[0] = -sqrtQ - a / 3;
      x[1] = -sqrtQ - a / 3;
      x[2] = 2 * sqrtQ - a / 3;
    }
    return 3;
  } else if (CR2 < CQ3) {
    double sqrtQ = sqrt(Q), sqrtQ3 = sqrtQ * sqrtQ * sqrtQ;
    double theta = ddm::acos_cr(R / sqrtQ3), norm = -2 * sqrtQ;
    double r0 = norm * ddm::cos_cr(theta / 3) - a / 3;
    double r1 = norm * ddm::cos_cr((theta + 2.0 * 3.14159265358979323846) / 3) - a / 3;
    double r2 = norm * ddm::cos_cr((theta - 2.0 * 3.14159265358979323846) / 3) - a / 3, tmp;
    if (r0 > r1) { tmp = r0; r0 = r1; r1 = tmp; }
    if (r1 > r2) {
      tmp = r1; r1 = r2; r2 = tmp;
      if (r0 > r1) { tmp = r0; r0 = r1; r1 = tmp; }
    }
    x[0] = r0; x[1] = r1; x[2] = r2;
    return 3;
  } else {
    double sgnR = (R >= 0 ? 1 : -1);
    double A = -sgnR * ddm::pow_cr(fabs(R) + sqrt(R2 - Q3), 1.0 / 3.0);
    double B = Q / A;
    x[0] = A + B - a / 3;
    return 1;
  }
}

// count[(first-1)*3 + (second-1)], first = later SNP, second = earlier SNP (src/ld.cpp:537-546)
__device__ inline LdCell ld_cell_from_counts(const int *count) {
#pragma clang fp contract(off)
  int total = 0;
  for (int c = 0; c < 9; c++) total += count[c];
  int j = count[0] * 2 + count[1] + count[3];
  int k = count[2] * 2 + count[1] + count[5];
  int l = count[6] * 2 + count[3] + count[7];
  int m = count[8] * 2 + count[5] + count[7];
  int e = count[4];
  int uv = j + k + e, wx = l + m + e, uw = j + l + e, vx = k + m + e;
  double cand[3] = {0, 0, 0};
  int nc = 3;
  if (e) {
    int tmp1 = e * (j + m - k - l), e2 = e * e, jm = j * m, kl = k * l;
    double a0 = -jm, a1 = e2 - tmp1 + jm + kl, a2 = tmp1 - 3 * e2, a3 = 2 * e2;
    if (jm == 0 || kl == 0) {
      if (jm == 0 && kl == 0) {
        cand[0] = 0.0;
        cand[1] = 1.0;
        cand[2] = ((double)(e + k + l - j - m)) / e / 2.0;
      } else if (jm == 0) {
        cand[0] = 0.0;
        nc = quad_roots(a3, a2, a1, &cand[1], &cand[2]) + 1;
      } else {
        cand[0] = 1.0;
        nc = quad_roots(a3, (a2 - 2 * tmp1), (a1 + 2 * tmp1), &cand[1], &cand[2]) + 1;
        cand[1] = 1.0 - cand[1];
        cand[2] = 1.0 - cand[2];
      }
    } else {
      nc = cubic_roots(a2 / a3, a1 / a3, a0 / a3, cand);
    }
  } else {
    int bottom = j * m + k * l;
    cand[0] = bottom != 0 ? (j * m) * 1.0 / bottom : 0;
    cand[1] = 1;
    cand[2] = 0;
  }
  double best = -1.0e308, p = 0;  // pick_best_p, src/ld.cpp:293-339
  for (int c = 0; c < nc; c++) {
    double pc = cand[c];
    if (pc > 1.0 || pc < 0.0) continue;
    double u = j + pc * e, v = k + (1.0 - pc) * e, w = l + (1.0 - pc) * e, x = m + pc * e;
    double ex[9] = {u * u, 2. * u * v, v * v, 2. * u * w, 2. * u * x + 2. * v * w, 2. * v * x, w * w, 2. * w * x, x * x};
    double sum = 0;
    for (int q = 0; q < 9; q++)
      if (count[q] && ex[q] > 0.0) sum += count[q] * ddm::log_cr(ex[q]);
    if (sum > best) { best = sum; p = pc; }
  }
  double u = j + p * e, v = k + (1.0 - p) * e, w = l + (1.0 - p) * e, x = m + p * e;
  LdCell r;
  double T2 = total * 2.0;
  double bigD = 0;
  r.rsq2 = -1.0;
  if (total) {
    bigD = u * x - w * v;
    if (uw && uv && wx && vx) r.rsq2 = (bigD / uw / uv) * (bigD / wx / vx);
  }
  r.pBA = u / T2;
  r.pBa = v / T2;
  r.pbA = w / T2;
  r.pba = x / T2;
  r.D = bigD / (T2 * T2);
  return r;
}

// K2: one thread per (SNP s, depth d) cell, s >= d+1.  Tables in forward SNP coordinates after pre_calc's
// reversals (src/relateHMM.cpp:3340-3354): mea[d][s-1] (depth x (M-1)); hap tables depth x M with the
// pBa <-> pbA swap; entries whose earlier SNP does not exist stay zero (tables are zero-filled first).
__global__ void ld_window_kernel(const uint64_t *__restrict__ G, int M, int W, int depth, int LD, int wantHap,
                                 double *__restrict__ mea, double *__restrict__ pBA, double *__restrict__ pBa,
                                 double *__restrict__ pbA, double *__restrict__ pba) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)M * depth) return;
  int d = (int)(tid / M), s = (int)(tid % M);  // consecutive lanes walk consecutive SNP rows
  if (s < d + 1) return;
  const uint64_t *ra = G + (size_t)s * W;            // first  = later SNP
  const uint64_t *rb = G + (size_t)(s - 1 - d) * W;  // second = earlier SNP
  int count[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < W; w++) {
    uint64_t xa = ra[w], xb = rb[w];
    uint64_t alo = xa & kLoMask, ahi = (xa >> 1) & kLoMask, blo = xb & kLoMask, bhi = (xb >> 1) & kLoMask;
    uint64_t A[3] = {alo & ~ahi, ahi & ~alo, ahi & alo};
    uint64_t B[3] = {blo & ~bhi, bhi & ~blo, bhi & blo};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int q = 0; q < 3; q++) count[i * 3 + q] += __popcll(A[i] & B[q]);
  }
  LdCell c = ld_cell_from_counts(count);
  mea[(size_t)d * (M - 1) + (s - 1)] = LD ? c.rsq2 : fabs(c.D);
  if (wantHap) {
    pBA[(size_t)d * M + s] = c.pBA;
    pBa[(size_t)d * M + s] = c.pbA;  // hap_global->pBa <- ld->pbA (src/relateHMM.cpp:3352)
    pbA[(size_t)d * M + s] = c.pBa;  // hap_global->pbA <- ld->pBa (:3353)
    pba[(size_t)d * M + s] = c.pba;
  }
}

}  // namespace relate
