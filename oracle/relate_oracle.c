/*
 * TEST INFRASTRUCTURE -- CPU oracle (see relate_oracle.h).  Plain C restatement of the reference's
 * all-pairs IBD HMM path; only tests/, smoke() and bench.py's cpu_baseline leg may use it.
 * Build: oracle/Makefile target `oracle` (gcc -O2 -ffp-contract=off).
 */
#include "relate_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX2(a, b) ((a) > (b) ? (a) : (b))
#define MIN2(a, b) ((a) < (b) ? (a) : (b))

/* ------------------------------------------------------------------ MAF (src/relateHMM.cpp:2985-3007) */
void orc_maf(const uint8_t *g, int N, int M, double *maf) {
  for (int s = 0; s < M; s++) {
    int sum = 0, n = 0;
    for (int i = 0; i < N; i++) {
      int v = g[(size_t)i * M + s];
      if (v == 0) continue;
      sum += v - 1;
      n++;
    }
    maf[s] = (((double)sum) / n) / 2;
  }
}

/* maf_stripper (src/relateHMM.cpp:3238-3250) */
int orc_maf_keep(const double *maf, int M, double min, int *keep) {
  int n = 0;
  for (int s = 0; s < M; s++) {
    keep[s] = (maf[s] != 0 && maf[s] > min && maf[s] < (1 - min)) ? 1 : 0;
    n += keep[s];
  }
  return n;
}

/* ------------------------------------------------------------------ LD cell (src/ld.cpp) */
/* gsl_poly_solve_quadratic as vendored at src/ld.cpp:71-125 */
static int solve_quadratic(double a, double b, double c, double *x0, double *x1) {
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

/* gsl_poly_solve_cubic with the widened double-root test of src/ld.cpp:139-220 */
static int solve_cubic(double a, double b, double c, double *x0, double *x1, double *x2) {
  double q = (a * a - 3 * b);
  double r = (2 * a * a * a - 9 * a * b + 27 * c);
  double Q = q / 9, R = r / 54;
  double Q3 = Q * Q * Q, R2 = R * R;
  double CR2 = 729 * r * r, CQ3 = 2916 * q * q * q;
  double xx = CR2 - CQ3, tol = 0.000000001;
  if (R == 0 && Q == 0) {
    *x0 = *x1 = *x2 = -a / 3;
    return 3;
  } else if (xx < tol && xx > -tol) {
    double sqrtQ = sqrt(Q);
    if (R > 0) {
      *x0 = -2 * sqrtQ - a / 3;
      *x1 = sqrtQ - a / 3;
      *x2 = sqrtQ - a / 3;
    } else {
      *x0 = -sqrtQ - a / 3;
      *x1 = -sqrtQ - a / 3;
      *x2 = 2 * sqrtQ - a / 3;
    }
    return 3;
  } else if (CR2 < CQ3) {
    double sqrtQ = sqrt(Q), sqrtQ3 = sqrtQ * sqrtQ * sqrtQ;
    double theta = acos(R / sqrtQ3), norm = -2 * sqrtQ;
    double r0 = norm * cos(theta / 3) - a / 3;
    double r1 = norm * cos((theta + 2.0 * M_PI) / 3) - a / 3;
    double r2 = norm * cos((theta - 2.0 * M_PI) / 3) - a / 3, tmp;
    if (r0 > r1) { tmp = r0; r0 = r1; r1 = tmp; }
    if (r1 > r2) {
      tmp = r1; r1 = r2; r2 = tmp;
      if (r0 > r1) { tmp = r0; r0 = r1; r1 = tmp; }
    }
    *x0 = r0; *x1 = r1; *x2 = r2;
    return 3;
  } else {
    double sgnR = (R >= 0 ? 1 : -1);
    double A = -sgnR * pow(fabs(R) + sqrt(R2 - Q3), 1.0 / 3.0);
    double B = Q / A;
    *x0 = A + B - a / 3;
    return 1;
  }
}

void orc_ld_cell(const uint8_t *first, const uint8_t *second, int n, int stride, double *out) {
  int count[9] = {0};
  for (int r = 0; r < n; r++) { /* get_geno_count, src/ld.cpp:537-546 */
    int a = first[(size_t)r * stride], b = second[(size_t)r * stride];
    if (a && b) count[(a - 1) * 3 + (b - 1)]++;
  }
  int total = 0;
  for (int c = 0; c < 9; c++) total += count[c];
  int j = count[0] * 2 + count[1] + count[3]; /* :564-567 */
  int k = count[2] * 2 + count[1] + count[5];
  int l = count[6] * 2 + count[3] + count[7];
  int m = count[8] * 2 + count[5] + count[7];
  int e = count[4];
  int uv = j + k + e, wx = l + m + e, uw = j + l + e, vx = k + m + e; /* :349-352 */

  double cand[3];
  int nc = 3;
  if (e) { /* :356-411 */
    int tmp1 = e * (j + m - k - l), e2 = e * e, jm = j * m, kl = k * l;
    double a0 = -jm, a1 = e2 - tmp1 + jm + kl, a2 = tmp1 - 3 * e2, a3 = 2 * e2;
    if (jm == 0 || kl == 0) {
      if (jm == 0 && kl == 0) {
        cand[0] = 0.0;
        cand[1] = 1.0;
        cand[2] = ((double)(e + k + l - j - m)) / e / 2.0;
      } else if (jm == 0) {
        cand[0] = 0.0;
        nc = solve_quadratic(a3, a2, a1, &cand[1], &cand[2]) + 1;
      } else {
        cand[0] = 1.0;
        cand[1] = cand[2] = 0;
        nc = solve_quadratic(a3, (a2 - 2 * tmp1), (a1 + 2 * tmp1), &cand[1], &cand[2]) + 1;
        cand[1] = 1.0 - cand[1];
        cand[2] = 1.0 - cand[2];
      }
    } else {
      nc = solve_cubic(a2 / a3, a1 / a3, a0 / a3, &cand[0], &cand[1], &cand[2]);
    }
  } else { /* :412-437 */
    int bottom = j * m + k * l;
    cand[0] = bottom != 0 ? (j * m) * 1.0 / bottom : 0;
    cand[1] = 1;
    cand[2] = 0;
  }
  /* pick_best_p, :293-339 */
  double best = -1.0e308, p = 0;
  for (int c = 0; c < nc; c++) {
    double pc = cand[c];
    if (pc > 1.0 || pc < 0.0) continue;
    double u = j + pc * e, v = k + (1.0 - pc) * e, w = l + (1.0 - pc) * e, x = m + pc * e;
    double ex[9] = {u * u, 2. * u * v, v * v, 2. * u * w, 2. * u * x + 2. * v * w, 2. * v * x, w * w, 2. * w * x, x * x};
    double sum = 0;
    for (int q = 0; q < 9; q++)
      if (count[q] && ex[q] > 0.0) sum += count[q] * log(ex[q]);
    if (sum > best) { best = sum; p = pc; }
  }
  double u = j + p * e, v = k + (1.0 - p) * e, w = l + (1.0 - p) * e, x = m + p * e; /* :441-444 */
  double bigD = 0, rsq2 = -1.0, dprime = -1.0;
  if (total) { /* :458-487 */
    bigD = u * x - w * v;
    if (bigD > 0) {
      int bottom = MIN2(vx * uv, wx * uw);
      dprime = bottom != 0 ? bigD / bottom : -1.0;
    } else {
      int bottom = MIN2(uw * uv, wx * vx);
      dprime = bottom != 0 ? -bigD / bottom : -1.0;
    }
    if (uw && uv && wx && vx) rsq2 = (bigD / uw / uv) * (bigD / wx / vx);
  }
  double lod = 0.0; /* :495-518 */
  if (j) lod += j * log(total * 2.0 * u / uw / uv);
  if (k) lod += k * log(total * 2.0 * v / vx / uv);
  if (l) lod += l * log(total * 2.0 * w / wx / uw);
  if (m) lod += m * log(total * 2.0 * x / wx / vx);
  if (e) lod += e * log(total * total * 2.0 / uv / uw * (u * x + v * w) / wx / vx);
  lod *= 0.43429448190325182765;
  double T2 = total * 2.0; /* src/relateHMM.cpp:222-227 */
  out[0] = u / T2;
  out[1] = v / T2;
  out[2] = w / T2;
  out[3] = x / T2;
  out[4] = bigD / (T2 * T2);
  out[5] = rsq2;
  out[6] = dprime;
  out[7] = lod;
  out[8] = p;
  out[9] = total;
}

/* snp_pair_range on the reversed matrix + revCols / revCols_and_extend + pBa<->pbA swap
 * (src/relateHMM.cpp:198-259, 441-478, 3340-3354), written in forward SNP coordinates. */
void orc_hap_global(const uint8_t *g, int N, int M, int depth, int LD, double *mea, double *pBA, double *pBa,
                    double *pbA, double *pba) {
  memset(mea, 0, sizeof(double) * (size_t)depth * (M - 1));
  memset(pBA, 0, sizeof(double) * (size_t)depth * M);
  memset(pBa, 0, sizeof(double) * (size_t)depth * M);
  memset(pbA, 0, sizeof(double) * (size_t)depth * M);
  memset(pba, 0, sizeof(double) * (size_t)depth * M);
  for (int d = 0; d < depth; d++)
    for (int s = d + 1; s < M; s++) {
      double c[10];
      orc_ld_cell(g + s, g + (s - 1 - d), N, M, c); /* first = later SNP s, second = SNP s-1-d */
      mea[(size_t)d * (M - 1) + (s - 1)] = LD ? c[5] : fabs(c[4]);
      pBA[(size_t)d * M + s] = c[0]; /* u: both "first" alleles                       */
      pBa[(size_t)d * M + s] = c[2]; /* hap_global->pBa <- ld->pbA (w), :3352          */
      pbA[(size_t)d * M + s] = c[1]; /* hap_global->pbA <- ld->pBa (v), :3353          */
      pba[(size_t)d * M + s] = c[3];
    }
}

/* pruning3 (src/relateHMM.cpp:669-756) on getMea (:651-666, depth = back) */
int orc_prune(const uint8_t *g, int N, int M, int LD, int back, double prune_val, int *keep) {
  int W = M - 1;
  double *mea = (double *)calloc((size_t)back * W, sizeof(double));
  for (int d = 0; d < back; d++)
    for (int s = d + 1; s < M; s++) {
      double c[10];
      orc_ld_cell(g + s, g + (s - 1 - d), N, M, c);
      mea[(size_t)d * W + (s - 1)] = LD ? c[5] : fabs(c[4]);
    }
#define MEA(r, c) mea[(size_t)(r)*W + (c)]
  int n = 1;
  keep[0] = 1;
  if (back > 1) {
    for (int nsnp = 1; nsnp < M; nsnp++) {
      int strip = 0;
      for (int r = 0; r < back; r++)
        if (MEA(r, nsnp - 1) > prune_val) { strip = 1; break; }
      if (strip) {
        keep[nsnp] = 0;
        if (back > 2) {
          for (int p = 0; p < back - 1; p++) {
            if (p + nsnp >= W) break;
            for (int q = p + 1; q < back - 1; q++) MEA(q, nsnp + p) = MEA(q + 1, p + nsnp);
            MEA(back - 1, nsnp + p) = 0;
          }
        } else
          MEA(1, nsnp) = 0; /* reference writes column nsnp unguarded (:731); W > nsnp except at the last SNP */
      } else {
        keep[nsnp] = 1;
        n++;
      }
    }
  } else {
    for (int i = 1; i < M; i++) {
      keep[i] = MEA(0, i - 1) <= prune_val ? 1 : 0;
      n += keep[i];
    }
  }
#undef MEA
  free(mea);
  return n;
}

/* ------------------------------------------------------------------ emissions */
/* error(), src/relateHMM.cpp:2828-2855 -- note the asymmetric het/hom terms, kept literally */
double orc_err(int tr, int o, double eps) {
  if (tr == 0) return o == 1 ? pow(1 - eps, 2.0) : o == 2 ? (1 - eps) * eps : pow(eps, 2.0);
  if (tr == 1) return o == 2 ? pow(1 - eps, 2.0) + pow(eps, 2.0) : 2 * (1 - eps) * eps;
  return o == 1 ? pow(eps, 2.0) : o == 2 ? (1 - eps) * eps : pow(1 - eps, 2.0);
}

/* emission2_new (:2776-2822) written term by term in the reference's order; q = 1-p given by caller
 * in the reference (maf2); here q is passed implicitly as 1-p exactly as sgl forms it (:3440-3441,
 * where p = 1 - maf is formed from q = maf).  To keep the rounding identical the caller-visible
 * entry point takes p and recovers q the way the call sites do. */
static void emis_single_pq(double p, double q, const double *ep, double *S3) {
  double r2, r1, r0;
  r2 = pow(p, 2.0) * ep[0];
  r1 = pow(p, 3) * ep[0];
  r0 = pow(p, 4) * ep[0];
  r1 += 2 * pow(p, 2) * q * ep[1];
  r0 += 4 * pow(p, 3) * q * ep[1];
  r0 += 2 * pow(p, 2) * pow(q, 2) * ep[2];
  r1 += 2 * pow(p, 2) * q * ep[3];
  r0 += 4 * pow(p, 3) * q * ep[3];
  r2 += 2 * q * p * ep[4];
  r1 += q * p * (q + p) * ep[4];
  r0 += 4 * pow(q, 2) * pow(p, 2) * ep[4];
  r1 += 2 * pow(q, 2) * p * ep[5];
  r0 += 4 * pow(q, 3) * p * ep[5];
  r0 += 2 * pow(q, 2) * pow(p, 2) * ep[6];
  r1 += 2 * pow(q, 2) * p * ep[7];
  r0 += 4 * pow(q, 3) * p * ep[7];
  r2 += pow(q, 2.0) * ep[8];
  r1 += pow(q, 3) * ep[8];
  r0 += pow(q, 4) * ep[8];
  S3[0] = r0;
  S3[1] = r1;
  S3[2] = r2;
}

static void err9(int o1, int o2, double eps, double *ep) { /* errorNoLD, :2861-2875 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) ep[i * 3 + j] = orc_err(i, o1, eps) * orc_err(j, o2, eps);
}

void orc_emis_single(double p, int o1, int o2, double eps, double *S3) {
  double ep[9];
  err9(o1, o2, eps, ep);
  emis_single_pq(p, 1.0 - p, ep, S3);
}

/*
 * LDemission_new (:2343-2704) restated by its generating rule instead of its 81 expanded polynomials
 * (rule recovered from the reference's coefficients; tests/test_oracle_emissions.py checks all 81 x 3
 * polynomials against the compiled reference and tests/golden/ld_poly_coeffs.json):
 *
 *   haplotypes h = (allele at conditioning SNP, allele at current SNP) with frequencies
 *   pBA,pbA,pBa,pba; a two-SNP genotype (gc,gp) (gc at current, gp at conditioning SNP, each the count
 *   0..2 of the a / b allele) has diplotype probability D(gc,gp) = sum over unordered haplotype pairs.
 *   For the true-genotype combination (i,j,k,l) = (ind1 cur, ind2 cur, ind1 cond, ind2 cond) the
 *   reference's polynomial is the sum over the DISTINCT orderings (s1,s2) of {i,j} and (t1,t2) of {k,l}
 *   of Q((s1,t1),(s2,t2)), i.e. which individual carries which genotype is forgotten per SNP, with
 *     IBD0: Q = D(g1) D(g2);  IBD2: Q = [g1==g2] D(g1);
 *     IBD1: Q = sum_h p_h * p_{g1 - h} * p_{g2 - h}   (g - h = the haplotype completing g given h).
 * The weights are errorLD's products (:2884-2916).
 */
static double hapfreq(const double *hp, int ac, int ap) { /* ac: 1 if allele a at current; ap: 1 if b at cond. */
  return hp[ap + 2 * ac]; /* hp = {pBA, pbA, pBa, pba} */
}
static double dip(const double *hp, int gc, int gp) {
  double s = 0; /* unordered pairs {h1,h2} with allele counts (gc,gp) */
  for (int c1 = 0; c1 <= 1; c1++)
    for (int p1 = 0; p1 <= 1; p1++) {
      int c2 = gc - c1, p2 = gp - p1;
      if (c2 < 0 || c2 > 1 || p2 < 0 || p2 > 1) continue;
      s += hapfreq(hp, c1, p1) * hapfreq(hp, c2, p2); /* ordered pairs: counts each unordered pair once or twice */
    }
  return s;
}
static double q_ibd1(const double *hp, int gc1, int gp1, int gc2, int gp2) {
  double s = 0;
  for (int c = 0; c <= 1; c++)
    for (int p = 0; p <= 1; p++) {
      int c1 = gc1 - c, p1 = gp1 - p, c2 = gc2 - c, p2 = gp2 - p;
      if (c1 < 0 || c1 > 1 || p1 < 0 || p1 > 1 || c2 < 0 || c2 > 1 || p2 < 0 || p2 > 1) continue;
      s += hapfreq(hp, c, p) * hapfreq(hp, c1, p1) * hapfreq(hp, c2, p2);
    }
  return s;
}

void orc_emis_ld(double pBA, double pbA, double pBa, double pba, int o1c, int o2c, int o1p, int o2p, double eps,
                 double *S3) {
  double hp[4] = {pBA, pbA, pBa, pba};
  double s0 = 0, s1 = 0, s2 = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          double w = orc_err(i, o1c, eps) * orc_err(j, o2c, eps) * orc_err(k, o1p, eps) * orc_err(l, o2p, eps);
          double p0 = 0, p1 = 0, p2 = 0;
          for (int sc = 0; sc < (i == j ? 1 : 2); sc++)
            for (int tc = 0; tc < (k == l ? 1 : 2); tc++) {
              int g1c = sc ? j : i, g2c = sc ? i : j, g1p = tc ? l : k, g2p = tc ? k : l;
              p0 += dip(hp, g1c, g1p) * dip(hp, g2c, g2p);
              p1 += q_ibd1(hp, g1c, g1p, g2c, g2p);
              if (g1c == g2c && g1p == g2p) p2 += dip(hp, g1c, g1p);
            }
          s0 += p0 * w;
          s1 += p1 * w;
          s2 += p2 * w;
        }
  S3[0] = s0;
  S3[1] = s1;
  S3[2] = s2;
}

/* sgl front half (src/relateHMM.cpp:3424-3472) with remove_cross/getHighestId/LD2_new in the clean
 * per-pair form: local row r = r-th previous kept SNP. */
int orc_pair_stage(const uint8_t *g, int N, int M, int i, int j, const double *maf, const double *pos, double eps,
                   int use_ld, int back, int back2, const double *mea, const double *pBA, const double *pBa,
                   const double *pbA, const double *pba, int *keep, double *t, double *S1, double *S, int *choose) {
  (void)N;
  int *idx = (int *)malloc(sizeof(int) * (size_t)M);
  int m = 0;
  for (int s = 0; s < M; s++) { /* which_arent_NA, :425-437 */
    keep[s] = (g[(size_t)i * M + s] != 0 && g[(size_t)j * M + s] != 0) ? 1 : 0;
    if (keep[s]) idx[m++] = s;
  }
  if (m == 0) { free(idx); return 0; }
  t[0] = 0; /* diff, :391-404 */
  t[m] = 0;
  for (int k = 1; k < m; k++) {
    double d = pos[idx[k]] - pos[idx[k - 1]];
    t[k] = d < 0 ? 10000000.0 : d;
  }
  for (int k = 0; k < m; k++) { /* errorNoLD + emission2_new with maf1_local = 1 - maf, :3439-3442 */
    int s = idx[k];
    double ep[9], r[3], q = maf[s], p = 1.0 - q;
    err9(g[(size_t)i * M + s], g[(size_t)j * M + s], eps, ep);
    emis_single_pq(p, q, ep, r);
    for (int x = 0; x < 3; x++) S1[(size_t)x * M + k] = r[x];
  }
  if (!use_ld || back <= 0) {
    for (int x = 0; x < 3; x++) memcpy(S + (size_t)x * M, S1 + (size_t)x * M, sizeof(double) * m);
    for (int k = 0; k < m; k++) choose[k] = -1;
    free(idx);
    return m;
  }
  choose[0] = -1;
  for (int x = 0; x < 3; x++) S[(size_t)x * M + 0] = S1[(size_t)x * M + 0]; /* LD2_new, :2955-2957 */
  for (int k = 1; k < m; k++) {
    int s = idx[k];
    /* remove_cross (:850-971) + getHighestId (:481-497) */
    int best = 0;
    double bestv = 0;
    for (int r = 0; r < back; r++) {
      double v = 0;
      if (k - 1 - r >= 0) {
        int dd = s - 1 - idx[k - 1 - r];
        if (dd < back2) v = mea[(size_t)dd * (M - 1) + (s - 1)];
      }
      if (r == 0) bestv = v;
      else if (v > bestv) { bestv = v; best = r; }
    }
    if (back == 1) best = 0;
    choose[k] = best;
    int kc = k - 1 - best;
    double hA = 0, hw = 0, hv = 0, hx = 0; /* pBA, pbA, pBa, pba of the chosen local row */
    if (kc >= 0) {
      int dd = s - 1 - idx[kc];
      if (dd < back2) {
        hA = pBA[(size_t)dd * M + s];
        hw = pbA[(size_t)dd * M + s];
        hv = pBa[(size_t)dd * M + s];
        hx = pba[(size_t)dd * M + s];
      }
    } else
      kc = 0; /* reference reads ind[-1] here (src/extractors.cpp:100-105): undefined, excluded from parity */
    int sc = idx[kc];
    int o1c = g[(size_t)i * M + s], o2c = g[(size_t)j * M + s], o1p = g[(size_t)i * M + sc], o2p = g[(size_t)j * M + sc];
    double S2[3], S1t[3], ep[9];
    orc_emis_ld(hA, hw, hv, hx, o1c, o2c, o1p, o2p, eps, S2);
    double maft = hA * 1 + hv * 1 + 0; /* math_to(1,pBAt,1,pBat,0), :2948 */
    err9(o1p, o2p, eps, ep);
    emis_single_pq(maft, 1.0 - maft, ep, S1t);
    for (int x = 0; x < 3; x++) S[(size_t)x * M + k] = S2[x] / S1t[x];
  }
  free(idx);
  return m;
}

/* ------------------------------------------------------------------ HMM kernels */
/* transition arrays shared by like/decodek2/viterbi (src/relateHMM.cpp:1440-1490); T[x*3+y][i], i in 0..m */
static double *build_trans(double a, double k2, double k1, double k0, const double *t, int n, int double_recom,
                           int zero_when_k1_is_1) {
  double *T = (double *)calloc((size_t)9 * n, sizeof(double));
#define TR(x, y) (T + (size_t)((x)*3 + (y)) * n)
  if (double_recom) { /* dbl_recom, :501-507 */
    double kk[3] = {k0, k1, k2};
    for (int x = 0; x < 3; x++)
      for (int y = 0; y < 3; y++)
        for (int i = 0; i < n; i++) TR(x, y)[i] = (1.0 - exp(-a * t[i])) * kk[y] + (x == y ? 1 : 0) * (exp(-a * t[i]));
    return T;
  }
  for (int i = 0; i < n; i++) TR(0, 1)[i] = (1.0 - exp(-a * t[i])) * k1 + 0 * (exp(-a * t[i]));
  if (k1 != 1) {
    for (int i = 0; i < n; i++) {
      double v = exp(-a * k1 * t[i]) * k2 / (k1 - 1) + exp(-a * t[i]) * k1 + exp(-a * t[i]) * k0 * k1 / (k1 - 1) + k2;
      TR(0, 2)[i] = v < 0.000000000000001 ? 0 : v; /* fix_underflow, :542-547 */
    }
  } else if (!zero_when_k1_is_1) {
    /* decodek2/viterbi leave IBD02 uninitialised (new[]) and then run fix_underflow (:1035-1041);
       like() zeroes it (:1463-1465).  Zero is used here in both cases. */
  }
  for (int i = 0; i < n; i++) TR(0, 0)[i] = TR(0, 1)[i] * -1 + TR(0, 2)[i] * -1 + 1; /* math_to, :509-514 */
  if (k1 == 0) {
    for (int i = 0; i < n; i++) TR(1, 0)[i] = TR(1, 2)[i] = TR(0, 1)[i];
  } else {
    for (int i = 0; i < n; i++) {
      TR(1, 0)[i] = (k0 / k1) * TR(0, 1)[i]; /* scale, :523-528 */
      TR(1, 2)[i] = (k2 / k1) * TR(0, 1)[i];
    }
  }
  for (int i = 0; i < n; i++) TR(1, 1)[i] = TR(1, 0)[i] * -1 + TR(1, 2)[i] * -1 + 1;
  for (int i = 0; i < n; i++) TR(2, 1)[i] = TR(0, 1)[i];
  if (k1 != 1) {
    for (int i = 0; i < n; i++) {
      double v = exp(-a * k1 * t[i]) * k0 / (k1 - 1.0) + exp(-a * t[i]) * k1 + exp(-a * t[i]) / (k1 - 1) * k2 * k1 -
                 (1 - k1) * k0 / (k1 - 1);
      TR(2, 0)[i] = v < 0.000000000000001 ? 0 : v;
    }
  }
  for (int i = 0; i < n; i++) TR(2, 2)[i] = TR(2, 1)[i] * -1 + TR(2, 0)[i] * -1 + 1;
  return T;
}

double orc_like(const double *delta, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                int double_recom, double alim1) {
  double a = delta[0], k2 = delta[1], k1 = delta[2];
  double k0 = 1 - (k1 + k2);
  if (k2 < 0 || k1 < 0 || k0 < 0 || k2 > 1 || a > alim1) return ORC_SYS_MAX; /* :1436 */
  int n = m + 1;
  double *T = build_trans(a, k2, k1, k0, t, n, double_recom, 1);
  double L0 = log(k0), L1 = log(k1), L2 = log(k2), k = 0; /* :1499-1521 */
  for (int i = 1; i < n - 1; i++) {
    double p0 = exp(L0 + Sk3[i - 1] - k), p1 = exp(L1 + Sk2[i - 1] - k), p2 = exp(L2 + Sk1[i - 1] - k);
    L0 = k + log(TR(0, 0)[i] * p0 + TR(1, 0)[i] * p1 + TR(2, 0)[i] * p2);
    L1 = k + log(TR(0, 1)[i] * p0 + TR(1, 1)[i] * p1 + TR(2, 1)[i] * p2);
    L2 = k + log(TR(0, 2)[i] * p0 + TR(1, 2)[i] * p1 + TR(2, 2)[i] * p2);
    k = MAX2(log(p1) + k, MAX2(log(p2) + k, log(p0) + k));
    if (isnan(k) || isnan(L0) || isnan(L1) || isnan(L2)) { /* reference prints and exit(0)s, :1513-1520 */
      free(T);
      return NAN;
    }
  }
  double q0 = Sk3[m - 1] + L0, q1 = Sk2[m - 1] + L1, q2 = Sk1[m - 1] + L2; /* :1523-1529 */
  k = MAX2(q0, MAX2(q1, q2));
  double l = k + log(exp(q0 - k) + exp(q1 - k) + exp(q2 - k));
  free(T);
  return -l;
}

void orc_decode(const double *par, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                int double_recom, int use_k2, double *post) {
  double a = par[0], k2 = use_k2 ? par[1] : 0, k1 = par[2];
  double k0 = use_k2 ? 1 - k2 - k1 : 1 - 0 - k1;
  int n = m + 1;
  double *T;
  if (use_k2) {
    T = build_trans(a, k2, k1, k0, t, n, double_recom, 0);
  } else { /* decode(), :1170-1189: 2-state transitions */
    T = (double *)calloc((size_t)9 * n, sizeof(double));
    for (int i = 0; i < n; i++) {
      double e = exp(-a * t[i]);
      if (double_recom) {
        TR(0, 0)[i] = (1.0 - e) * k0 + 1 * e;
        TR(0, 1)[i] = (1.0 - e) * k1 + 0 * e;
        TR(1, 0)[i] = (1.0 - e) * k0 + 0 * e;
        TR(1, 1)[i] = (1.0 - e) * k1 + 1 * e;
      } else {
        TR(0, 1)[i] = (1.0 - e) * k1 + 0 * e;
        TR(0, 0)[i] = 1.0 - TR(0, 1)[i];
        TR(1, 0)[i] = k1 == 0 ? TR(0, 1)[i] : (k0 / k1) * TR(0, 1)[i];
        TR(1, 1)[i] = 1.0 - TR(1, 0)[i];
      }
    }
  }
  int ns = use_k2 ? 3 : 2; /* states 0..ns-1 = IBD0, IBD1[, IBD2] */
  const double *Sk[3] = {Sk3, Sk2, Sk1};
  double lk[3] = {log(k0), log(k1), use_k2 ? log(k2) : 0};
  double *F = (double *)calloc((size_t)3 * m, sizeof(double)), *B = (double *)calloc((size_t)3 * m, sizeof(double));
#define FW(x, i) F[(size_t)(x)*m + (i)]
#define BW(x, i) B[(size_t)(x)*m + (i)]
  for (int x = 0; x < ns; x++) FW(x, 0) = lk[x] + Sk[x][0]; /* :1068-1070 */
  double k = 0;
  for (int i = 1; i < m; i++) { /* :1074-1083 */
    double p[3] = {0, 0, 0};
    for (int x = 0; x < ns; x++) p[x] = exp(FW(x, i - 1) - k);
    for (int y = 0; y < ns; y++) {
      double s = TR(0, y)[i] * p[0] + TR(1, y)[i] * p[1];
      if (ns == 3) s = s + TR(2, y)[i] * p[2];
      FW(y, i) = k + log(s) + Sk[y][i];
    }
    k = FW(0, i);
    for (int x = 1; x < ns; x++) k = MAX2(k, FW(x, i)); /* max is order-independent */
  }
  double q[3];
  for (int x = 0; x < ns; x++) q[x] = FW(x, m - 1);
  k = q[0];
  for (int x = 1; x < ns; x++) k = MAX2(k, q[x]);
  double se = exp(q[0] - k) + exp(q[1] - k);
  if (ns == 3) se = se + exp(q[2] - k);
  double l = k + log(se); /* :1084-1088 */
  for (int x = 0; x < ns; x++) BW(x, m - 1) = 0; /* real backward sweep, :1111-1125 */
  k = 0;
  for (int i = m - 2; i >= 0; i--) {
    double p[3] = {0, 0, 0};
    for (int x = 0; x < ns; x++) p[x] = exp(Sk[x][i + 1] + BW(x, i + 1) - k);
    for (int x = 0; x < ns; x++) {
      double s;
      if (ns == 3) s = p[2] * TR(x, 2)[i + 1] + p[1] * TR(x, 1)[i + 1] + p[0] * TR(x, 0)[i + 1];
      else s = p[1] * TR(x, 1)[i + 1] + p[0] * TR(x, 0)[i + 1];
      BW(x, i) = k + log(s);
    }
    k = BW(0, i);
    for (int x = 1; x < ns; x++) k = MAX2(k, BW(x, i));
  }
  for (int i = 0; i < m; i++) { /* :1133-1137 */
    post[(size_t)0 * m + i] = ns == 3 ? exp(FW(2, i) + BW(2, i) - l) : 0.0;
    post[(size_t)1 * m + i] = exp(FW(1, i) + BW(1, i) - l);
    post[(size_t)2 * m + i] = exp(FW(0, i) + BW(0, i) - l);
  }
  free(F);
  free(B);
  free(T);
}

static double max3(double a, double b, double c) { /* :1272-1279 */
  if (a > b && a > c) return a;
  else if (b > a && b > c) return b;
  return c;
}
static int which_max(double a, double b, double c) { /* :1280-1286 */
  if (a > b && a > c) return 0;
  else if (b > a && b > c) return 1;
  return 2;
}

void orc_viterbi(const double *par, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                 int double_recom, int *path) {
  double a = par[0], k2 = par[1], k1 = par[2], k0 = 1 - k2 - k1;
  int n = m + 1;
  double *T = build_trans(a, k2, k1, k0, t, n, double_recom, 0);
  for (size_t q = 0; q < (size_t)9 * n; q++) T[q] = log(T[q]); /* row_log, :1342-1350 */
  double *v = (double *)calloc((size_t)3 * m, sizeof(double));
  int *bp = (int *)calloc((size_t)3 * m, sizeof(int));
#define V(x, i) v[(size_t)(x)*m + (i)]
  const double *Sk[3] = {Sk3, Sk2, Sk1};
  double maxed = MAX2(log(k0), MAX2(log(k1), log(k2))); /* :1356-1359 */
  for (int x = 0; x < 3; x++) V(x, 0) = maxed + Sk[x][0];
  for (int i = 1; i < m; i++)
    for (int y = 0; y < 3; y++) { /* :1370-1377 */
      V(y, i) = Sk[y][i] + max3(TR(0, y)[i] + V(0, i - 1), TR(1, y)[i] + V(1, i - 1), TR(2, y)[i] + V(2, i - 1));
      bp[(size_t)y * m + i] = which_max(V(0, i - 1) + TR(0, y)[i], V(1, i - 1) + TR(1, y)[i], V(2, i - 1) + TR(2, y)[i]);
    }
  path[m - 1] = which_max(V(0, m - 1), V(1, m - 1), V(2, m - 1));
  for (int i = m - 1; i > 0; i--) path[i - 1] = bp[(size_t)path[i] * m + i];
  free(v);
  free(bp);
  free(T);
}

double orc_calculateA(double k0, double k1, double k2, double phi) { /* :974-1007 */
  (void)k0;
  double ma, mb;
  if (k2 == 0) {
    ma = 1 - log(k1) / log(2);
    mb = 0;
  } else {
    double pw = pow(k1 + 2 * k2, 2);
    double sq = sqrt(pw - 4 * k2);
    double xa = (k1 + 2 * k2 + sq) / 2.0;
    double xb = k2 / xa;
    ma = 1 - log(xa) / log(2);
    mb = 1 - log(xb) / log(2);
  }
  double mm = ma + mb;
  return -mm * log(1 - phi);
}

double orc_objective(int mode, const double *x, double fixA_val, double fixK2_val, double phi, const double *Sk1,
                     const double *Sk2, const double *Sk3, const double *t, int m, int double_recom, double alim1) {
  double d[3];
  if (mode == 0) { /* full_optim_fun, :1827-1830 */
    d[0] = x[0]; d[1] = x[1]; d[2] = x[2];
    return orc_like(d, Sk1, Sk2, Sk3, t, m, double_recom, alim1);
  }
  double k2, k1, a;
  if (mode == 1) { k2 = x[0]; k1 = x[1]; }            /* calcA_fun, :1564-1593 */
  else if (mode == 2) { k2 = fixK2_val; k1 = x[1]; }  /* fixK2_fun, :1680-1704 */
  else { k2 = fixK2_val; k1 = x[0]; }                 /* fixK2_calcA_fun :1736, fixK2_fixA_fun :1783 */
  double k0 = 1 - k2 - k1;
  if (pow((k1 + 2 * k2), 2.0) < 4 * k2) return ORC_SYS_MAX;
  if (mode == 1 || mode == 3) a = orc_calculateA(k0, k1, k2, phi);
  else if (mode == 2) a = x[0];
  else a = fixA_val;
  if (mode != 4 && (isnan(a) || isinf(a))) return ORC_SYS_MAX;
  d[0] = a; d[1] = k2; d[2] = k1;
  return orc_like(d, Sk1, Sk2, Sk3, t, m, double_recom, alim1);
}
