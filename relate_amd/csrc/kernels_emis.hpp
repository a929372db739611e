// Emission kernels: per-SNP shared tables (built once per data set) and the per-pair staging kernel (K4).
// Replaces error/errorNoLD/emission2_new/errorLD/LDemission_new/LD2_new and the per-pair table
// re-indexing remove_cross/getHighestId (src/relateHMM.cpp:2776-2975, 850-971, 481-497).  gfx950 only.
//
// Emission records live SNP-major in HBM, 16 B per pair and SNP: S_ibd1 / S_ibd0, S_ibd2 / S_ibd0.  Two consecutive
// SNPs share a row, and a row holds the pairs of the staged block in groups of 64 (one wave); a group keeps its two
// values as two planes of 64 x {SNP 2h, SNP 2h+1} (16 B per pair and plane, 2 048 B per group), so a wave fetches two
// records of each of its pairs with two 16-byte loads -- one vector-memory instruction per record, every one a
// contiguous KiB (the address unit does ~20 cycles of work per wave instruction whatever its width, and the sweeps
// are bound by it before they are bound by bytes):
//     rec[(s / 2) * 4 * RP + (p / 64) * 256 + c * 128 + (p % 64) * 2 + s % 2],   c = 0: S1/S0, 1: S2/S0
// (RP = pairs rounded up to a multiple of 64, chosen by the host so that consecutive rows do not lie a power of
// two apart).  Emissions matter to the recursions only up to a per-SNP factor: the factor S_ibd0 is
// taken out here, multiplied up per pair (mantissa and power-of-two exponent, scale_reduce_kernel) and added to the
// log-likelihood once; it cancels in posteriors.
//
// Distances are NOT in the record.  The reference gives a pair the distance between its consecutive KEPT SNPs
// (diff, src/relateHMM.cpp:391-404); here every SNP -- kept or dropped for the pair -- steps by the shared distance
// tsh[s] = pos[s] - pos[s-1] (negative: 1e7, as diff has it; tsh[0] = 0), a dropped SNP with emission (1, 1, 1): the
// transition matrices form a semigroup, T(t1) T(t2) = T(t1 + t2) (both of the reference's models are e^{Qt}), so the
// steps over a run of dropped SNPs compose to the reference's single step, and the start vector k is stationary
// (k T = k), so dropped SNPs in front of the first kept one change nothing.  A third of the bytes of every sweep,
// and the distance is wave-uniform (scalar loads, scalar tier test).  Record values: kept (r1 >= 0, r2 >= 0);
// dropped (-1, +1) -- the sign bit of r1 is the "dropped" flag, the recursions multiply by |r1|.
// One case does not compose: a run of dropped SNPs across a chromosome break whose two kept ends have a NON-negative
// position difference d (the reference then steps by T(d), not by the break's 1e7).  boundary_fixup_kernel finds
// those runs (a handful of genotype reads per pair and break) and marks their records with the sign bit of r2 =
// "this lane's distance is not the shared one": dropped (-1, -tau) steps by tau, kept (r1, -r2) by 0; the run's
// SNPs get tau = 0 except the last dropped one, which gets d.  The sweeps test the sign bits of a group's r2 with
// one ballot and take a per-lane-distance path for such a group.
// The "full" format (Viterbi, inspection) has four planes t, S_ibd0, S_ibd1, S_ibd2 with the pair's own distances.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace relate {

static constexpr int kRecPlanes = 2, kRecPlanesFull = 4;
// offset in doubles of SNP s, pair column col in a block of npl planes and padded pair count RP; plane c is c * 128 further
__host__ __device__ inline size_t rec_off(int s, int col, int npl, int RP) {
  return (size_t)(s >> 1) * (2 * npl) * RP + (size_t)(col >> 6) * (npl * 128) + (col & 63) * 2 + (s & 1);
}

// A table entry holds {S_ibd0, S_ibd1 / S_ibd0, S_ibd2 / S_ibd0} (correctly rounded divisions, once per entry), so the
// staging kernel divides nothing.  An all-zero emission (conditioning row beyond the data) stays all zero.
__device__ inline void store_entry(double *__restrict__ o, const double *S) {
  bool z = S[0] == 0;
  o[0] = S[0];
  o[1] = z ? (S[1] == 0 ? 0.0 : INFINITY) : S[1] / S[0];
  o[2] = z ? (S[2] == 0 ? 0.0 : INFINITY) : S[2] / S[0];
}

// running product of the factors taken out of a pair's records: mantissa and power-of-two exponent
struct ScaleAcc {
  double m = 1.0;
  int e = 0;
  __device__ inline void fold() {
    int f = (m > 0) ? __builtin_amdgcn_frexp_exp(m) : 0;
    m = __builtin_amdgcn_ldexp(m, -f);
    e += f;
  }
};

// One staged record of a lane.  state: 0 dropped, 1/2 kept with entry {s0, r1, r2}, 3 all-zero emission.  Returns the
// values of the planes (FULL: t, S0, S1, S2; else S1/S0, S2/S0) and folds the factor into acc.
template <bool FULL>
__device__ inline void make_record(int state, double t, double s0, double r1, double r2, ScaleAcc &acc, int &nkept,
                                   double (&out)[FULL ? 4 : 2]) {
  if (state == 0) { t = -0.0; s0 = 1; r1 = 1; r2 = 1; }
  else if (state == 3) { s0 = 0; r1 = 0; r2 = 0; }
  nkept += state != 0;
  acc.m *= s0;
  if constexpr (FULL) { out[0] = t; out[1] = s0; out[2] = r1 * s0; out[3] = r2 * s0; }
  else { out[0] = state == 0 ? -1.0 : r1; out[1] = r2; }
}
template <bool FULL>
__device__ inline void put_record(double *__restrict__ rec, int RP, int s, int p, int state, double t, double s0,
                                  double r1, double r2, ScaleAcc &acc, int &nkept) {
  constexpr int NPL = FULL ? kRecPlanesFull : kRecPlanes;
  double v[NPL];
  make_record<FULL>(state, t, s0, r1, r2, acc, nkept, v);
  double *q = rec + rec_off(s, p, NPL, RP);
#pragma unroll
  for (int c = 0; c < NPL; c++) q[c * 128] = v[c];
}

// P(observed code o in 1..3 | true genotype index tr in 0..2): error(), src/relateHMM.cpp:2828-2855.
// The het/hom asymmetry is the reference's and is kept literally.
__device__ __host__ inline double err_tab(int tr, int o, double eps) {
  double a = (1 - eps) * (1 - eps), b = (1 - eps) * eps, c = eps * eps;
  if (tr == 0) return o == 1 ? a : (o == 2 ? b : c);
  if (tr == 1) return o == 2 ? a + c : 2 * b;
  return o == 1 ? c : (o == 2 ? b : a);
}

// Single-SNP emission (emission2_new, :2776-2822) by its generating rule: for true genotypes (i,j) of the two
// individuals the reference sums over the distinct orderings of {i,j} of Q(g1,g2) with
//   IBD0: Q = D(g1) D(g2), IBD1: Q = sum_h p_h p_{g1-h} p_{g2-h}, IBD2: Q = [g1==g2] D(g1),
// D(g) = {p^2, 2pq, q^2}; the weights are errorNoLD's products (:2861-2875).  p = freq. for genotype index 0.
__device__ __host__ inline void emis_single(double p, double q, int o1, int o2, double eps, double *S) {
  double D[3] = {p * p, 2 * p * q, q * q};
  double hp[2] = {p, q};
  double u[3], v[3];
  for (int i = 0; i < 3; i++) {
    u[i] = err_tab(i, o1, eps);
    v[i] = err_tab(i, o2, eps);
  }
  double s0 = 0, s1 = 0, s2 = 0;
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double w = u[a] * v[b] + (a != b ? u[b] * v[a] : 0.0);
      double q1 = 0;
      for (int h = 0; h <= 1; h++) {
        int ca = a - h, cb = b - h;
        if (ca < 0 || ca > 1 || cb < 0 || cb > 1) continue;
        q1 += hp[h] * hp[ca] * hp[cb];
      }
      s0 += D[a] * D[b] * w;
      s1 += q1 * w;
      if (a == b) s2 += D[a] * w;
    }
  S[0] = s0;
  S[1] = s1;
  S[2] = s2;
}

// Two-SNP (LD) emission, LDemission_new (:2343-2704) by the same generating rule over two-SNP genotypes
// g = (gc, gp) (current / conditioning SNP); hp = {pBA, pbA, pBa, pba}, index = [b at cond.] + 2*[a at cur.].
// Wc/Wp fold errorLD's weights (:2884-2916) over the distinct orderings at each SNP separately.
struct LdPoly {
  double D[9];    // D[gc*3+gp]  diplotype probabilities
  double hp[4];
};

__device__ __host__ inline void ldpoly_init(LdPoly &L, double pBA, double pbA, double pBa, double pba) {
  L.hp[0] = pBA; L.hp[1] = pbA; L.hp[2] = pBa; L.hp[3] = pba;
  for (int gc = 0; gc < 3; gc++)
    for (int gp = 0; gp < 3; gp++) {
      double s = 0;
      for (int c1 = 0; c1 <= 1; c1++)
        for (int p1 = 0; p1 <= 1; p1++) {
          int c2 = gc - c1, p2 = gp - p1;
          if (c2 < 0 || c2 > 1 || p2 < 0 || p2 > 1) continue;
          s += L.hp[p1 + 2 * c1] * L.hp[p2 + 2 * c2];
        }
      L.D[gc * 3 + gp] = s;
    }
}

__device__ __host__ inline double ldpoly_q1(const LdPoly &L, int gc1, int gp1, int gc2, int gp2) {
  double s = 0;
  for (int c = 0; c <= 1; c++)
    for (int p = 0; p <= 1; p++) {
      int c1 = gc1 - c, p1 = gp1 - p, c2 = gc2 - c, p2 = gp2 - p;
      if (c1 < 0 || c1 > 1 || p1 < 0 || p1 > 1 || c2 < 0 || c2 > 1 || p2 < 0 || p2 > 1) continue;
      s += L.hp[p + 2 * c] * L.hp[p1 + 2 * c1] * L.hp[p2 + 2 * c2];
    }
  return s;
}

__device__ __host__ inline void emis_ld(const LdPoly &L, int o1c, int o2c, int o1p, int o2p, double eps, double *S) {
  double uc[3], vc[3], up[3], vp[3];
  for (int i = 0; i < 3; i++) {
    uc[i] = err_tab(i, o1c, eps);
    vc[i] = err_tab(i, o2c, eps);
    up[i] = err_tab(i, o1p, eps);
    vp[i] = err_tab(i, o2p, eps);
  }
  double s0 = 0, s1 = 0, s2 = 0;
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double wc = uc[a] * vc[b] + (a != b ? uc[b] * vc[a] : 0.0);
      for (int c = 0; c < 3; c++)
        for (int d = 0; d < 3; d++) {
          double w = wc * (up[c] * vp[d] + (c != d ? up[d] * vp[c] : 0.0));
          s0 += L.D[a * 3 + c] * L.D[b * 3 + d] * w;
          s1 += ldpoly_q1(L, a, c, b, d) * w;
          if (a == b && c == d) s2 += L.D[a * 3 + c] * w;
        }
    }
  S[0] = s0;
  S[1] = s1;
  S[2] = s2;
}

// conditional emission of LD2_new (:2939-2975): S2 / S1t, S1t = emission2_new with p = pBA + pBa (:2948-2953)
__device__ __host__ inline void emis_cond(double pBA, double pbA, double pBa, double pba, int o1c, int o2c, int o1p,
                                          int o2p, double eps, double *S) {
  LdPoly L;
  ldpoly_init(L, pBA, pbA, pBa, pba);
  double S2[3], S1t[3];
  emis_ld(L, o1c, o2c, o1p, o2p, eps, S2);
  double pc = pBA + pBa;
  emis_single(pc, 1.0 - pc, o1p, o2p, eps, S1t);
  S[0] = S2[0] / S1t[0];
  S[1] = S2[1] / S1t[1];
  S[2] = S2[2] / S1t[2];
}

// getHighestId (:481-497) on the first `back` rows of the global mea table: first strictly largest value.
__global__ void choose_global_kernel(const double *__restrict__ mea, int M, int back, int *__restrict__ choose) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  if (s == 0) { choose[0] = -1; return; }
  int best = 0;
  double bv = mea[s - 1];
  for (int d = 1; d < back; d++) {
    double v = mea[(size_t)d * (M - 1) + (s - 1)];
    if (v > bv) { bv = v; best = d; }
  }
  choose[s] = best;
}

// Per-SNP tables shared by every pair.  tab1[s][o1*3+o2][3]: unconditional emission (27 doubles per SNP).
// tabL[s][code][3], code = ((o1c*3+o2c)*3+o1p)*3+o2p with o in 0..2 (= genotype code - 1): emission of SNP s
// conditional on SNP s-1-choose[s] (243 doubles = 1944 B per SNP; 0.97 GB at M = 5e5).
__global__ void table_single_kernel(const double *__restrict__ maf, int M, double eps, double *__restrict__ tab1) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= M * 9) return;
  int s = tid / 9, code = tid % 9;
  double S[3];
  emis_single(1.0 - maf[s], maf[s], code / 3 + 1, code % 3 + 1, eps, S);  // maf1_local = 1 - maf2_local (:3440-3441)
  store_entry(tab1 + (size_t)tid * 3, S);
}

__global__ void table_ld_kernel(const int *__restrict__ choose, const double *__restrict__ pBA,
                                const double *__restrict__ pbA, const double *__restrict__ pBa,
                                const double *__restrict__ pba, int M, double eps, double *__restrict__ tabL) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)M * 81) return;
  int s = (int)(tid / 81), code = (int)(tid % 81);
  double S[3] = {0, 0, 0};
  if (s > 0) {
    size_t at = (size_t)choose[s] * M + s;
    int o2p = code % 3, o1p = (code / 3) % 3, o2c = (code / 9) % 3, o1c = code / 27;
    emis_cond(pBA[at], pbA[at], pBa[at], pba[at], o1c + 1, o2c + 1, o1p + 1, o2p + 1, eps, S);
  }
  store_entry(tabL + (size_t)tid * 3, S);
}

__device__ inline int geno_at(const uint64_t *__restrict__ G, int W, int s, int ind) {
  return (int)((G[(size_t)s * W + (ind >> 5)] >> (2 * (ind & 31))) & 3u);
}
// genotype of SNP s in the individual-major copy (row of one individual: SNPs 32w .. 32w+31 per word, 2 bits each)
__device__ inline int geno_row_at(const uint64_t *__restrict__ row, int s) {
  return (int)((row[s >> 5] >> (2 * (s & 31))) & 3u);
}
__device__ inline uint64_t kept_bits(uint64_t a, uint64_t b) {  // bit 2k set iff SNP k of the word is non-missing in both
  return (a | (a >> 1)) & (b | (b >> 1)) & 0x5555555555555555ull;
}

// K4: front half of sgl (which_arent_NA, diff, errorNoLD/emission2_new, remove_cross, getHighestId, LD2_new;
// src/relateHMM.cpp:3424-3472) for a block of P pairs, any back2.  Workgroup = one wave = 64 consecutive pairs x one
// run of 64 SNPs.  Fast path (no missing genotype among the previous `back` SNPs): the pair's re-indexed LD table equals
// the global one, so the emission is a lookup in tabL.  Slow path: re-index per the clean form of remove_cross (row r =
// r-th previous kept SNP, zero beyond depth back2), pick the row with getHighestId's rule and evaluate the conditional
// emission directly -- a scan of up to back2 depths and the 81-term polynomial.  After one missing genotype the next
// `back` SNPs of the pair are slow: with back = 50, 10 % of all (pair, SNP) items, so a wave that took them where they
// fall ran the slow path at practically every step for a few of its lanes.  Instead every lane first walks its run and
// does the fast items, leaving a bit per slow one; the wave's slow items are then numbered (prefix sum over the lanes)
// and taken 64 at a time, whichever lane they came from; their S0 factors go through LDS back to the owning lane, which
// folds them into its product in SNP order.
static constexpr int kStageRun = 64;
static constexpr int kSlowBuf = 1024;  // S0 factors of slow items in LDS at a time (a workgroup has up to 64 x kStageRun)
struct SlowItem {
  int state;
  double t, e0, e1, e2;
};
template <bool FULL>
__device__ inline SlowItem stage_slow_item(const uint64_t *__restrict__ Gt, int Wt, int M, int i1, int i2, int s,
                                           const double *__restrict__ pos, const double *__restrict__ mea,
                                           const int *__restrict__ chooseG,
                                           const double *__restrict__ pBA, const double *__restrict__ pbA,
                                           const double *__restrict__ pBa, const double *__restrict__ pba, int back,
                                           int back2, double eps) {
  SlowItem R{2, 0.0, 1.0, 1.0, 1.0};
  const uint64_t *r1 = Gt + (size_t)i1 * Wt, *r2 = Gt + (size_t)i2 * Wt;
  const int g1 = geno_row_at(r1, s), g2 = geno_row_at(r2, s);
  // rows r = 0..back-1 <-> r-th previous kept SNP at distance dd = s-1-s_prev (< back2).  The kept SNPs before s are
  // the set bits of the two rows' words, taken from s-1 downwards: a handful of word loads in place of two genotype
  // reads per depth.
  int bestS = -1, nfound = 0, nearest = -1;  // bestS: SNP of the first row with the strictly largest measure
  double bv = 0;
  const int lowest = max(0, s - back2);  // SNPs below are beyond depth back2
  // The measures of the depths 0 .. back-1 have a known winner, the global choice c = chooseG[s] (the first strictly
  // largest of them): when the SNP at depth c is kept by the pair, it is also the first strictly largest among the kept
  // ones of those depths, and only the depths from `back` on -- as many as SNPs were dropped above them, one or two --
  // have to be read and compared.  Otherwise (the winner itself is dropped; 2-4 % of the slow items) every depth is read.
  const int c = s > 0 ? chooseG[s] : -1;
  const int spc = s - 1 - c;
  const bool quick = c >= 0 && spc >= 0 && geno_row_at(r1, spc) && geno_row_at(r2, spc);
  const int skip_above = quick ? max(s - back, 0) : s;  // SNPs skip_above .. s-1 are counted, not read
  if (quick) { bv = mea[(size_t)c * (M - 1) + (s - 1)]; bestS = spc; }
  for (int w = (s - 1) >> 5; w >= 0 && nfound < back && w * 32 + 31 >= lowest && s > 0; w--) {
    uint64_t K = kept_bits(r1[w], r2[w]);
    if (w == ((s - 1) >> 5) && ((s - 1) & 31) != 31) K &= ((uint64_t)1 << (2 * (((s - 1) & 31) + 1))) - 1;  // SNPs < s only
    if (K && nearest < 0) nearest = w * 32 + ((63 - __clzll(K)) >> 1);
    if (w * 32 + 31 >= skip_above) {  // the part of the word at depths < back: every kept SNP there is a row, none wins
      uint64_t top = K;
      if (w * 32 < skip_above) {
        const int cut = 2 * (skip_above - w * 32);  // bits from `cut` up belong to SNPs >= skip_above
        top = K & ~(((uint64_t)1 << cut) - 1);
        K &= ((uint64_t)1 << cut) - 1;
      } else {
        K = 0;
      }
      nfound += __popcll(top);  // (never more than back: the depths 0 .. back-1 hold at most back SNPs)
    }
    while (K && nfound < back) {
      const int b = 63 - __clzll(K);  // highest set bit: the nearest kept SNP not yet taken
      K &= ~((uint64_t)1 << b);
      const int sp = w * 32 + (b >> 1);
      if (sp < lowest) { K = 0; break; }
      const int dd = s - 1 - sp;
      double v = mea[(size_t)dd * (M - 1) + (s - 1)];
      if (bestS < 0 && nfound == 0) { bv = v; bestS = sp; }
      else if (v > bv) { bv = v; bestS = sp; }
      nfound++;
    }
  }
  if (nfound == 0) { bv = 0; bestS = -1; }
  if (nfound < back && 0.0 > bv) bestS = -1;  // a zero-filled row beyond the window wins (getHighestId, :492)
  if (bestS < 0) {
    R.state = 3;  // hap frequencies of that row are zero -> S2 = 0
  } else {
    int dd = s - 1 - bestS;
    size_t at = (size_t)dd * M + s;
    double S[3], E[3];
    emis_cond(pBA[at], pbA[at], pBa[at], pba[at], g1, g2, geno_row_at(r1, bestS), geno_row_at(r2, bestS), eps, S);
    store_entry(E, S);
    R.e0 = E[0]; R.e1 = E[1]; R.e2 = E[2];
  }
  if (FULL) {  // the pair's own distance: to the last kept SNP before s (a slow item always has one)
    int sp = nearest;
    if (sp < 0)  // beyond depth back2: further down the rows
      for (int q = lowest - 1; q >= 0; q--)
        if (geno_row_at(r1, q) && geno_row_at(r2, q)) { sp = q; break; }
    if (sp >= 0) {
      double d = pos[s] - pos[sp];
      R.t = d < 0 ? 10000000.0 : d;
    }
  }
  return R;
}

template <bool FULL>
__global__ void __launch_bounds__(64)
stage_pairs_kernel(const uint64_t *__restrict__ G, const uint64_t *__restrict__ Gt, int Wt, int M, int W,
                   const int32_t *__restrict__ pairs, int P, const double *__restrict__ pos, const double *__restrict__ tab1, const double *__restrict__ tabL,
                   const int *__restrict__ chooseG, const double *__restrict__ mea, const double *__restrict__ pBA,
                   const double *__restrict__ pbA, const double *__restrict__ pBa, const double *__restrict__ pba,
                   int useLD, int back, int back2, double eps, double *__restrict__ rec, int RP, int col0,
                   double *__restrict__ sc_m, int *__restrict__ sc_e, int *__restrict__ sc_n) {
  // pair p of the list goes to column col0 + p of the block
  __shared__ double slow_s0[kSlowBuf];
  __shared__ unsigned long long masks[64];
  __shared__ int offs[65], ind[64][2];
  const int groups = (P + 63) / 64;
  const int c = blockIdx.x / groups, lane = threadIdx.x;
  const int p = (blockIdx.x % groups) * 64 + lane;
  const bool live = p < P;
  const int pc = live ? p : P - 1;
  const int i1 = pairs[2 * pc], i2 = pairs[2 * pc + 1];
  const int s0 = c * kStageRun, s1 = min(M, s0 + kStageRun);
  ScaleAcc acc;
  int nkept = 0;
  unsigned long long slow = 0;
  if (live) {
    // The two individuals' genotype words (individual-major copy) of the run and of the 64 SNPs below it, eight loads
    // issued together: every genotype the fast items need while back <= 64 -- the SNP itself, the SNP its global
    // choice conditions on, the state at the run's start -- then comes out of registers.  (s0 is a multiple of 64.)
    const uint64_t *r1 = Gt + (size_t)i1 * Wt, *r2 = Gt + (size_t)i2 * Wt;
    const int w0 = s0 >> 5;
    uint64_t wa[4], wb[4];  // words w0 - 2 .. w0 + 1
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int w = w0 - 2 + k;
      const bool in = w >= 0 && w < Wt;
      wa[k] = in ? r1[w] : 0;
      wb[k] = in ? r2[w] : 0;
    }
    // state at the run's start: index of the last kept SNP before s0 (-1: none) and the number of consecutive kept SNPs
    // immediately before s0 (saturated at back)
    int last = -1, run = 0;
    {
      const uint64_t E = 0x5555555555555555ull;
      const uint64_t K1 = kept_bits(wa[1], wb[1]), K0 = kept_bits(wa[0], wb[0]);  // SNPs s0-32 .. s0-1, s0-64 .. s0-33
      if (K1) last = s0 - 32 + ((63 - __clzll(K1)) >> 1);
      else if (K0) last = s0 - 64 + ((63 - __clzll(K0)) >> 1);
      else
        for (int q = s0 - 65; q >= 0; q--)
          if (geno_row_at(r1, q) && geno_row_at(r2, q)) { last = q; break; }
      const uint64_t n1 = ~K1 & E, n0 = ~K0 & E;  // the SNPs not kept (words that do not exist count as such)
      if (n1) run = 31 - ((63 - __clzll(n1)) >> 1);
      else if (n0) run = 63 - ((63 - __clzll(n0)) >> 1);
      else {
        run = 64;
        for (int q = s0 - 65; q >= 0 && run < back; q--) {
          if (geno_row_at(r1, q) && geno_row_at(r2, q)) run++;
          else break;
        }
      }
      if (run > back) run = back;
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const uint64_t ca = wa[2 + half], cb = wb[2 + half];
      for (int s = s0 + 32 * half; s < min(s1, s0 + 32 * half + 32); s++) {
        const int g1 = (int)((ca >> (2 * (s & 31))) & 3u), g2 = (int)((cb >> (2 * (s & 31))) & 3u);
        int state = 0;
        double t = 0, e0 = 1, e1 = 1, e2 = 1;
        if (g1 == 0 || g2 == 0) {
          run = 0;
        } else {
          state = 2;
          if (FULL && last >= 0) {  // else first kept SNP: t[0] = 0 in diff (:391-404); unconditional emission (LD2_new :2955-2957)
            double d = pos[s] - pos[last];
            t = d < 0 ? 10000000.0 : d;
          }
          if (!useLD || last < 0) {
            const double *e = tab1 + ((size_t)s * 9 + (g1 - 1) * 3 + (g2 - 1)) * 3;
            e0 = e[0]; e1 = e[1]; e2 = e[2];
          } else if (run >= min(back, s)) {
            const int sc = max(s - 1 - chooseG[s], 0);
            const int rel = sc - (s0 - 64);
            int h1, h2;
            if (rel >= 0) {
              const int k = rel >> 5;
              const uint64_t ha = k == 0 ? wa[0] : k == 1 ? wa[1] : k == 2 ? wa[2] : wa[3];
              const uint64_t hb = k == 0 ? wb[0] : k == 1 ? wb[1] : k == 2 ? wb[2] : wb[3];
              h1 = (int)((ha >> (2 * (rel & 31))) & 3u);
              h2 = (int)((hb >> (2 * (rel & 31))) & 3u);
            } else {  // back > 64
              h1 = geno_row_at(r1, sc);
              h2 = geno_row_at(r2, sc);
            }
            int code = (((g1 - 1) * 3 + (g2 - 1)) * 3 + (h1 - 1)) * 3 + (h2 - 1);
            const double *e = tabL + ((size_t)s * 81 + code) * 3;
            e0 = e[0]; e1 = e[1]; e2 = e[2];
          } else {
            state = -1;  // slow: later, with the wave's other slow items
            slow |= 1ull << (s - s0);
          }
          last = s;
          if (run < back) run++;
        }
        if (state >= 0) {
          put_record<FULL>(rec, RP, s, col0 + p, state, t, e0, e1, e2, acc, nkept);
          if ((s & 7) == 7) acc.fold();
        } else {
          nkept++;
        }
      }
    }
    acc.fold();
  }
  masks[lane] = slow;
  ind[lane][0] = i1; ind[lane][1] = i2;
  __syncthreads();
  if (lane == 0) {
    int tot = 0;
    for (int l = 0; l < 64; l++) { offs[l] = tot; tot += __popcll(masks[l]); }
    offs[64] = tot;
  }
  __syncthreads();
  const int total = offs[64];
  const int n = __popcll(slow), at = offs[lane];
  int k = 0;  // the lane's slow items folded so far
  for (int base = 0; base < total; base += kSlowBuf) {  // (one round, but for runs with over a quarter of slow items)
    const int end = min(total, base + kSlowBuf);
    for (int j = base + lane; j < end; j += 64) {
      int lo = 0, hi = 64;  // the lane L with offs[L] <= j < offs[L + 1]
      while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offs[mid] <= j) lo = mid; else hi = mid;
      }
      const int L = lo;
      unsigned long long m = masks[L];
      for (int q = j - offs[L]; q > 0; q--) m &= m - 1;
      const int s = s0 + (int)__ffsll((long long)m) - 1;
      const SlowItem R = stage_slow_item<FULL>(Gt, Wt, M, ind[L][0], ind[L][1], s, pos, mea, chooseG, pBA, pbA, pBa, pba, back, back2, eps);
      ScaleAcc dummy;
      int nk = 0;
      put_record<FULL>(rec, RP, s, col0 + (blockIdx.x % groups) * 64 + L, R.state, R.t, R.e0, R.e1, R.e2, dummy, nk);
      slow_s0[j - base] = R.state == 3 ? 0.0 : R.e0;
    }
    __syncthreads();
    if (live)
      for (; k < n && at + k < end; k++) {  // the lane's factors of this round, in SNP order
        acc.m *= slow_s0[at + k - base];
        if ((k & 7) == 7) acc.fold();
      }
    __syncthreads();
  }
  if (live) {
    acc.fold();
    sc_m[(size_t)c * RP + col0 + p] = acc.m;
    sc_e[(size_t)c * RP + col0 + p] = acc.e;
    sc_n[(size_t)c * RP + col0 + p] = nkept;
  }
}

// Folds the per-run partial products of the staging kernels in a fixed order: per pair the factor taken out of its
// records as mantissa x 2^exponent (the likelihood kernels add its logarithm once), and the number of kept SNPs.
// Workgroup = 64 pairs x kScaleSeg segments of the run axis; a thread folds its segment (loads unrolled: the loop is
// latency-bound otherwise), then the first thread row folds the segments in order.
static constexpr int kScaleSeg = 8;
__global__ void __launch_bounds__(64 * kScaleSeg)
scale_reduce_kernel(const double *__restrict__ sc_m, const int *__restrict__ sc_e, const int *__restrict__ sc_n,
                    int nruns, int RP, int col0, int P, double *__restrict__ pm, int *__restrict__ pe,
                    int *__restrict__ pn) {
  // columns [col0, col0 + P) of the partial arrays and of the per-pair results
  __shared__ double lm[kScaleSeg][64];
  __shared__ int le[kScaleSeg][64], ln[kScaleSeg][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  sc_m += col0; sc_e += col0; sc_n += col0;
  pm += col0; pe += col0; pn += col0;
  const int per = (nruns + kScaleSeg - 1) / kScaleSeg;
  const int c0 = seg * per, c1 = min(nruns, c0 + per);
  ScaleAcc acc;
  int n = 0;
  if (p < P) {
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
      double m[8];
      int e[8], k[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        size_t at = (size_t)(c + u) * RP + p;
        m[u] = sc_m[at]; e[u] = sc_e[at]; k[u] = sc_n[at];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) { acc.m *= m[u]; acc.e += e[u]; n += k[u]; }
      acc.fold();  // mantissas are in [0.5, 1): eight of them stay far from underflow
    }
    for (; c < c1; c++) {
      size_t at = (size_t)c * RP + p;
      acc.m *= sc_m[at]; acc.e += sc_e[at]; n += sc_n[at];
    }
    acc.fold();
  }
  lm[seg][lane] = acc.m; le[seg][lane] = acc.e; ln[seg][lane] = n;
  __syncthreads();
  if (seg == 0 && p < P) {
    ScaleAcc tot;
    int nt = 0;
#pragma unroll
    for (int g = 0; g < kScaleSeg; g++) { tot.m *= lm[g][lane]; tot.e += le[g][lane]; nt += ln[g][lane]; }
    tot.fold();
    pm[p] = tot.m; pe[p] = tot.e; pn[p] = nt;
  }
}

// ---- windowed staging (back2 <= 32) ---------------------------------------------------------------------------
// Individual-major copy of the packed genotypes: Gt[ind * Wt + w] holds SNPs 32w .. 32w+31 of one individual at
// 2 bits each, so a lane (= one pair) fetches 32 SNPs of both individuals with two 8-byte loads and does the rest
// of the per-SNP work (missingness, window search, genotype codes) on registers.
__global__ void transpose_geno_kernel(const uint64_t *__restrict__ G, int N, int M, int W, int Wt,
                                      uint64_t *__restrict__ Gt) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)N * Wt) return;
  int ind = (int)(tid % N), w = (int)(tid / N);  // lanes walk individuals: neighbouring bits of the same words
  uint64_t out = 0;
  int s0 = w * 32;
  for (int k = 0; k < 32; k++) {
    int s = s0 + k;
    if (s >= M) break;
    uint64_t g = (G[(size_t)s * W + (ind >> 5)] >> (2 * (ind & 31))) & 3u;
    out |= g << (2 * k);
  }
  Gt[(size_t)ind * Wt + w] = out;
}

// tabD[s][dd][code][3]: emission of SNP s conditional on SNP s-1-dd for every depth dd < back2 (the per-pair
// re-indexing of remove_cross can select any of them); zero where the earlier SNP does not exist.
__global__ void table_ld_depth_kernel(const double *__restrict__ pBA, const double *__restrict__ pbA,
                                      const double *__restrict__ pBa, const double *__restrict__ pba, int M, int back2,
                                      double eps, double *__restrict__ tabD) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)M * back2 * 81) return;
  int code = (int)(tid % 81);
  long sd = tid / 81;
  int dd = (int)(sd % back2), s = (int)(sd / back2);
  double S[3] = {0, 0, 0};
  if (s - 1 - dd >= 0) {
    size_t at = (size_t)dd * M + s;
    int o2p = code % 3, o1p = (code / 3) % 3, o2c = (code / 9) % 3, o1c = code / 27;
    emis_cond(pBA[at], pbA[at], pBa[at], pba[at], o1c + 1, o2c + 1, o1p + 1, o2p + 1, eps, S);
  }
  store_entry(tabD + (size_t)tid * 3, S);
}

__device__ inline uint64_t both_present(uint64_t a, uint64_t b) {  // bit 2k set iff SNP k is non-missing in both
  return (a | (a >> 1)) & (b | (b >> 1)) & 0x5555555555555555ull;
}
__device__ inline uint32_t win2(uint64_t lo, uint64_t hi, int q) {  // 2-bit field q of the 64-field window {lo, hi}
  uint64_t w = q < 32 ? lo : hi;
  return (uint32_t)(w >> (2 * (q & 31))) & 3u;
}

// Preference order of the conditioning depths of every SNP: order[s * back2 + i] = the depths dd < min(back2, s)
// sorted by (LD measure descending, dd ascending), 255 beyond.  getHighestId's "first strictly largest" over any
// subset of rows is the first entry of this list that belongs to the subset.
__global__ void ld_order_kernel(const double *__restrict__ mea, int M, int back2, uint8_t *__restrict__ order) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  uint8_t *o = order + (size_t)s * back2;
  int n = min(back2, s);
  for (int i = 0; i < back2; i++) o[i] = 255;
  for (int dd = 0; dd < n; dd++) {  // insertion sort, n <= 32
    double v = mea[(size_t)dd * (M - 1) + (s - 1)];
    int at = dd;
    while (at > 0) {
      double u = mea[(size_t)o[at - 1] * (M - 1) + (s - 1)];
      if (!(v > u)) break;  // strict: equal values keep the smaller depth first
      o[at] = o[at - 1];
      at--;
    }
    o[at] = (uint8_t)dd;
  }
}

// K4 for back2 <= 32.  Workgroup = 256 consecutive pairs x one run of `words` 32-SNP words, so the SNP index is
// scalar and pos / choose / order are scalar loads.  Every lane applies the clean form of remove_cross +
// getHighestId: among the previous back2 SNPs take the `back` nearest that are kept for the pair and
// choose the one with the first strictly largest LD measure; a zero-filled row beyond the window wins over negative
// measures.  All of it on the pair's 2-bit words in registers: bit 2*dd of the reversed kept-window = SNP s-1-dd.
// A pair with no missing genotype among its `back` nearest SNPs (99 % of lane-steps at 0.1 % missingness) takes the
// global choose; the others walk the SNP's preference list.  The emission is then a lookup in tabD; no lane
// evaluates the 81-term polynomial.
template <bool FULL>
__global__ void __launch_bounds__(256)
stage_pairs_win_kernel(const uint64_t *__restrict__ Gt, int M, int Wt, const int32_t *__restrict__ pairs, int P,
                       int words, const double *__restrict__ pos, const double *__restrict__ tab1,
                       const double *__restrict__ tabD, const double *__restrict__ mea,
                       const int *__restrict__ chooseG, const uint8_t *__restrict__ order, int useLD, int back,
                       int back2, double *__restrict__ rec, int RP, int col0, double *__restrict__ sc_m,
                       int *__restrict__ sc_e, int *__restrict__ sc_n) {  // pair p of the list -> column col0 + p
  // Every plane store of a wave is 64 consecutive doubles: 512 B of whole lines per instruction.
  int bpc = (P + 255) / 256;
  int c = blockIdx.x / bpc;
  int p = (blockIdx.x % bpc) * 256 + threadIdx.x;
  if (p >= P) return;
  const uint64_t *r1 = Gt + (size_t)pairs[2 * p] * Wt, *r2 = Gt + (size_t)pairs[2 * p + 1] * Wt;
  int w0 = c * words, w1 = min(Wt, w0 + words);
  const uint64_t even = 0x5555555555555555ull;
  const uint64_t wmask = back2 >= 32 ? even : (((uint64_t)1 << (2 * back2)) - 1) & even;
  const uint64_t fastmask = (((uint64_t)1 << (2 * back)) - 1) & even;  // back <= 16
  uint64_t a_lo = 0, b_lo = 0;  // previous word of the two individuals
  int last = -1;                // last kept SNP before the current one
  ScaleAcc acc;
  int nkept = 0;
  if (w0 > 0) {
    a_lo = r1[w0 - 1];
    b_lo = r2[w0 - 1];
    uint64_t K = both_present(a_lo, b_lo);
    if (K) {
      last = (w0 - 1) * 32 + (63 - __clzll(K)) / 2;
    } else {
      for (int w = w0 - 2; w >= 0; w--) {  // a run of >= 32 missing SNPs: rare
        uint64_t K2 = both_present(r1[w], r2[w]);
        if (K2) { last = w * 32 + (63 - __clzll(K2)) / 2; break; }
      }
    }
  }
  for (int w = w0; w < w1; w++) {
    uint64_t a_hi = r1[w], b_hi = r2[w];
    uint64_t K_lo = both_present(a_lo, b_lo), K_hi = both_present(a_hi, b_hi);
    int kmax = min(32, M - w * 32);
    // Batches of kBatch SNPs: first all selections (register work) and the position loads, then all table loads, then
    // all stores, so that a lane has kBatch gathers in flight instead of one memory round trip per SNP.
    constexpr int kBatch = 4;
    for (int k0 = 0; k0 < kmax; k0 += kBatch) {
      const double *e[kBatch];
      double tpos[kBatch], lpos[kBatch];
      int state[kBatch];  // 0 dropped, 1 first kept SNP, 2 emission from e[], 3 all-zero emission
#pragma unroll
      for (int u = 0; u < kBatch; u++) {
        int k = k0 + u;
        state[u] = 0;
        e[u] = tab1;
        tpos[u] = 0;
        lpos[u] = 0;
        if (k >= kmax) continue;  // uniform
        int s = w * 32 + k, q = 32 + k;
        uint32_t g1 = (uint32_t)(a_hi >> (2 * k)) & 3u, g2 = (uint32_t)(b_hi >> (2 * k)) & 3u;
        if (g1 == 0 || g2 == 0) continue;
        e[u] = tab1 + ((size_t)s * 9 + (g1 - 1) * 3 + (g2 - 1)) * 3;
        if (last < 0) {
          state[u] = 1;
        } else {
          state[u] = 2;
          if constexpr (FULL) {
            tpos[u] = pos[s];
            lpos[u] = pos[last];
          }
          if (useLD) {
            // kept flags of SNPs s-32 .. s-1 (bit 2j <-> SNP s-32+j), then reversed: bit 2*dd <-> SNP s-1-dd
            uint64_t Wk = k ? ((K_lo >> (2 * k)) | (K_hi << (64 - 2 * k))) : K_lo;
            uint64_t sel = (__brevll(Wk) >> 1) & wmask;
            int bestdd;
            if ((sel & fastmask) == fastmask) {
              bestdd = chooseG[s];
            } else {
              int n = __popcll(sel);
              while (n > back) {  // keep the `back` nearest kept SNPs
                sel &= ~((uint64_t)1 << (63 - __clzll(sel)));
                n--;
              }
              bestdd = -1;
              if (n > 0) {
                const uint8_t *o = order + (size_t)s * back2;
                for (int i = 0; i < back2; i++) {
                  int dd = o[i];
                  if (dd == 255) break;
                  if ((sel >> (2 * dd)) & 1u) { bestdd = dd; break; }
                }
                // a zero-filled row beyond the kept ones wins getHighestId over a negative measure (:492)
                if (bestdd >= 0 && n < back && 0.0 > mea[(size_t)bestdd * (M - 1) + (s - 1)]) bestdd = -1;
              }
            }
            if (bestdd < 0) {
              state[u] = 3;
            } else {
              int q2 = q - 1 - bestdd;
              uint32_t h1 = win2(a_lo, a_hi, q2), h2 = win2(b_lo, b_hi, q2);
              int code = (((g1 - 1) * 3 + (g2 - 1)) * 3 + (h1 - 1)) * 3 + (h2 - 1);
              e[u] = tabD + (((size_t)s * back2 + bestdd) * 81 + code) * 3;
            }
          }
        }
        last = s;
      }
      double v[kBatch][3];
#pragma unroll
      for (int u = 0; u < kBatch; u++) {
        v[u][0] = e[u][0]; v[u][1] = e[u][1]; v[u][2] = e[u][2];
      }
#pragma unroll
      for (int u = 0; u < kBatch; u += 2) {  // two SNPs = one 16-byte element per plane
        constexpr int NPL = FULL ? kRecPlanesFull : kRecPlanes;
        if (k0 + u >= kmax) continue;  // uniform
        double va[NPL], vb[NPL];
        double d = tpos[u] - lpos[u];
        make_record<FULL>(state[u], state[u] == 1 ? 0.0 : (d < 0 ? 10000000.0 : d), v[u][0], v[u][1], v[u][2], acc,
                          nkept, va);
        double *q = rec + rec_off(w * 32 + k0 + u, col0 + p, NPL, RP);
        if (k0 + u + 1 < kmax) {  // uniform
          d = tpos[u + 1] - lpos[u + 1];
          make_record<FULL>(state[u + 1], state[u + 1] == 1 ? 0.0 : (d < 0 ? 10000000.0 : d), v[u + 1][0], v[u + 1][1],
                            v[u + 1][2], acc, nkept, vb);
#pragma unroll
          for (int c = 0; c < NPL; c++) *(double2 *)(q + c * 128) = double2{va[c], vb[c]};
        } else {
#pragma unroll
          for (int c = 0; c < NPL; c++) q[c * 128] = va[c];
        }
      }
      if ((k0 & 4) == 4) acc.fold();  // every 8 records
    }
    a_lo = a_hi;
    b_lo = b_hi;
  }
  acc.fold();
  sc_m[(size_t)c * RP + col0 + p] = acc.m;
  sc_e[(size_t)c * RP + col0 + p] = acc.e;
  sc_n[(size_t)c * RP + col0 + p] = nkept;
}

// Runs of dropped SNPs across a chromosome break whose kept ends lie a non-negative distance apart (header): one
// thread per (pair, break).  bnd[b] = first SNP of a chromosome (pos[s] < pos[s-1]).  L = last kept SNP before the
// break, F = first kept SNP from the break on; if SNPs were dropped in between and pos[F] - pos[L] >= 0 the records
// of (L, F] are marked.  Several breaks inside one run find the same L and F and write the same values.
__global__ void boundary_fixup_kernel(const uint64_t *__restrict__ G, int M, int W, const int32_t *__restrict__ pairs,
                                      int P, const int *__restrict__ bnd, int nb, const double *__restrict__ pos,
                                      double *__restrict__ rec, int RP, int col0, int *__restrict__ own,
                                      const int *__restrict__ nsnp) {
  long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)P * nb) return;
  const int p = (int)(tid % P), b = bnd[tid / P];
  if (nsnp[col0 + p] == 0) return;  // no kept SNP: nothing to mark (and nothing to stop the scans below)
  const int i1 = pairs[2 * p], i2 = pairs[2 * p + 1];
  int L = -1, F = -1;
  for (int s = b - 1; s >= 0; s--)
    if (geno_at(G, W, s, i1) && geno_at(G, W, s, i2)) { L = s; break; }
  for (int s = b; s < M; s++)
    if (geno_at(G, W, s, i1) && geno_at(G, W, s, i2)) { F = s; break; }
  if (L < 0 || F < 0 || F == L + 1) return;
  const double d = pos[F] - pos[L];
  if (d < 0) return;  // the reference steps by 1e7 and so does the shared distance of the break
  for (int s = L + 1; s < F; s++) {
    double *q = rec + rec_off(s, col0 + p, kRecPlanes, RP);
    q[0] = -1.0;
    q[128] = s == F - 1 ? -d : -0.0;
  }
  double *q = rec + rec_off(F, col0 + p, kRecPlanes, RP);
  q[128] = -fabs(q[128]);
  own[col0 + p] = 1;
}

}  // namespace relate
