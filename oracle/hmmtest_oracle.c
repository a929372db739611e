/*
 * TEST INFRASTRUCTURE -- CPU oracle for the linkage tester HMMtest (src/HMMtest.txt of aalbrechtsen/relate).
 * Plain-C restatement in the reference's float arithmetic and summation order; used only by tests/ as the checker
 * of relate_hip_linkage.  Nothing under relate_amd/ may include, link or call it.  Pinned against outputs of the
 * reference program itself (oracle/_ref/HMMtest_ref, tests/golden/hmmtest/) by tests/test_hmmtest_oracle.py.
 * Build: part of `make -C oracle oracle` (-ffp-contract=off: every float operation rounds as the reference's).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* fix_missing, src/HMMtest.txt:550-664 (its quirks kept: difPos[j+1] for the right gap, "dif2 =+", the weights) */
static void fix_missing(float *post, int R, int M, const float *pos) {
  float *dif = (float *)calloc((size_t)M + 2, sizeof(float));
  for (int i = 1; i < M; i++) dif[i - 1] = pos[i] - pos[i - 1];
  for (int r = 0; r < R; r++) {
    float *p = post + (size_t)r * M;
    if (p[0] == -1) {
      int n = 1;
      while (n < M && p[n] == -1) n++;
      float nx = p[n < M ? n : M - 1];
      for (int j = 0; j < n; j++) p[j] = nx;
    }
    if (p[M - 1] == -1) {
      int n = 1;
      while (n < M && p[M - 1 - n] == -1) n++;
      float nx = p[M - 1 - n];
      for (int j = 0; j < n; j++) p[M - 1 - j] = nx;
    }
    for (int j = 1; j < M - 1; j++) {
      if (p[j] == -1) {
        float first = p[j - 1], next = p[j + 1];
        float dif2 = dif[j + 1];
        int o = 1;
        while (next == -1) {
          o++;
          dif2 = +dif[j + o];
          next = p[j + o];
        }
        float dif1 = dif[j - 1];
        float sum = dif1 + dif2;
        float t1 = (dif1 / sum) * first, t2 = (dif2 / sum) * next;
        if (dif1 < 0) { t1 = 0; t2 = next; }
        else if (dif2 < 0) { t2 = 0; t1 = first; }
        p[j] = t1 + t2;
      }
    }
  }
  free(dif);
}

/* getCaseList, :693-829: index[row] in {1 case pair, 2 control pair, 0/-1 unused}; returns n (individuals) */
static void case_lists(const int *cc, int n, int ccAll, int *index, int *n1, int *n0) {
  int *aff = (int *)malloc(sizeof(int) * n);
  int na = 0;
  for (int i = 0; i < n; i++) {
    aff[i] = cc[i] ? 0 : 1; /* "changed per anders request": 0 in the file = affected */
    na += aff[i];
  }
  *n1 = na;
  *n0 = n - na;
  int row = 0;
  for (int i = 0; i < n - 1; i++)
    for (int j = i + 1; j < n; j++, row++) {
      if (aff[i] && aff[j]) index[row] = 1;
      else if (aff[j]) index[row] = -1; /* exists(col1, idList) branch (:785); the discard list is always empty */
      else if (!aff[i] && !aff[j]) index[row] = 2;
      else index[row] = 0;
    }
  if (ccAll == 2) { for (int r = 0; r < row; r++) if (index[r] == 0) index[r] = 2; }
  else if (ccAll == 1) { for (int r = 0; r < row; r++) if (index[r] == 0) index[r] = 1; }
  free(aff);
}

static double lcg01(int *seed) { /* d_uniform_01, :1063-1072 */
  int k = *seed / 127773;
  *seed = 16807 * (*seed - k * 127773) - k * 2836;
  if (*seed < 0) *seed = *seed + 2147483647;
  return (double)(*seed) * 4.656612875E-10;
}

/* permutate_dFileList, :1113-1143 */
static void permute_lists(int n, int n1, int seed, int *index) {
  int *p = (int *)malloc(sizeof(int) * n), *in = (int *)calloc(n, sizeof(int));
  for (int i = 1; i <= n; i++) p[i - 1] = i;
  for (int i = 1; i <= n; i++) {
    int j = i + (int)(lcg01(&seed) * (double)(n + 1 - i));
    if (j < i) j = i;
    if (n < j) j = n;
    int t = p[i - 1]; p[i - 1] = p[j - 1]; p[j - 1] = t;
  }
  for (int i = 0; i < n1; i++) in[p[i] - 1] = 1;
  int row = 0;
  for (int i = 0; i < n - 1; i++)
    for (int j = i + 1; j < n; j++, row++) index[row] = (in[i] && in[j]) ? 1 : ((!in[i] && !in[j]) ? 2 : 0);
  free(p);
  free(in);
}

/* generate_affectLists, :455-470: overwrites the FRONT of the two lists; their lengths stay what getCaseList made
 * them, so when a permutation yields fewer case/control pairs than the observed lists hold (ccAll != 0 with mixed
 * pairs) the tail keeps entries of the previous lists.  Kept: it is what the reference computes. */
static void fill_lists(const int *index, int R, int *la, int *ln) {
  int a = 0, n = 0;
  for (int r = 0; r < R; r++) {
    if (index[r] == 1) la[a++] = r;
    else if (index[r] == 2) ln[n++] = r;
  }
}

static float list_mean(const float *K, const int *list, int n) { /* get_mean, :831-840 */
  float s = 0;
  for (int i = 0; i < n; i++) s += K[list[i]];
  return s / n;
}

static void col_means(const float *post, int M, const int *list, int n, int missing, float *U) { /* :909-935 */
  for (int s = 0; s < M; s++) {
    float sum = 0;
    int m = 0;
    for (int j = 0; j < n; j++) {
      float v = post[(size_t)list[j] * M + s];
      if (!missing || v != -1) { sum += v; m++; }
    }
    U[s] = sum / (float)(missing ? m : n);
  }
}

static float sum_cov(const float *cov, int R, const int *list, int n) { /* :972-979 */
  float s = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) s += cov[(size_t)list[i] * R + list[j]];
  return s;
}

static float pooled_bottom(int na, int nn, float varU1, float varU0) { /* :1279 */
  float t = ((nn - 1) * varU0 + (na - 1) * varU1) / (nn + na - 2);
  double u = (double)t * (1.0 / (double)(float)nn + 1.0 / (double)(float)na);
  return (float)sqrt(u);
}

static void statistic(const float *post, int R, int M, const float *K, const float *cov, int missing, const int *la,
                      int na, const int *ln, int nn, float *U1, float *U0, float *X) {
  float mn1 = list_mean(K, la, na), mn0 = list_mean(K, ln, nn);
  col_means(post, M, la, na, missing, U1);
  col_means(post, M, ln, nn, missing, U0);
  float varU1 = sum_cov(cov, R, la, na) / (na * na), varU0 = sum_cov(cov, R, ln, nn) / (nn * nn);
  float bottom = pooled_bottom(na, nn, varU1, varU0);
  for (int s = 0; s < M; s++) {
    float z1 = U1[s] + (-mn1), z0 = U0[s] + (-mn0);
    X[s] = (z1 - z0) / bottom;
  }
}

/* uber_main's case-control branch (:1195-1367).  post: R x M floats, modified in place when infer && pos.
 * Kin: NULL = row means (:941-968).  cc: the -D file's values for n_ind individuals, R must be n_ind(n_ind-1)/2.
 * summary = {pmaxsim, psim/nsim, psigsim, case pairs, control pairs}.  Returns 0, or -1 on a size mismatch. */
int orc_linkage(float *post, int R, int M, const float *pos, const float *Kin, int isK0, const int *cc, int n_ind,
                int nsim, int ccAll, double sig, int infer, unsigned seed, float *K, float *cov, float *stat, float *xmax,
                double *summary) {
  if ((long)n_ind * (n_ind - 1) / 2 != R) return -1;
  int missing = 1;
  if (infer == 1 && pos) { fix_missing(post, R, M, pos); missing = 0; }
  if (Kin) {
    for (int r = 0; r < R; r++) K[r] = isK0 ? 1 - Kin[r] : Kin[r];
  } else {
    for (int r = 0; r < R; r++) {
      float s = 0;
      int m = 0;
      for (int j = 0; j < M; j++) {
        float v = post[(size_t)r * M + j];
        if (!missing || v != -1) { s += v; m++; }
      }
      K[r] = s / (float)(missing ? m : M);
    }
  }
  for (int i = 0; i < R; i++) /* calculateCoVar, :842-907 */
    for (int j = i; j < R; j++) {
      float s = 0;
      int m = 0;
      const float *a = post + (size_t)i * M, *b = post + (size_t)j * M;
      for (int c = 0; c < M; c++)
        if (!missing || (a[c] != -1 && b[c] != -1)) { s += (a[c] - K[i]) * (b[c] - K[j]); m++; }
      float v = missing ? (float)(s / ((float)m - 1.0)) : (float)(s / (M - 1.0));
      cov[(size_t)i * R + j] = v;
      cov[(size_t)j * R + i] = v;
    }
  int *index = (int *)malloc(sizeof(int) * R), *la = (int *)malloc(sizeof(int) * R), *ln = (int *)malloc(sizeof(int) * R);
  float *U1 = (float *)malloc(sizeof(float) * M), *U0 = (float *)malloc(sizeof(float) * M);
  float *Xi = (float *)malloc(sizeof(float) * M);
  int *psim = (int *)calloc(M, sizeof(int));
  int n1, n0;
  case_lists(cc, n_ind, ccAll, index, &n1, &n0);
  int na = 0, nn = 0;
  for (int r = 0; r < R; r++) { na += index[r] == 1; nn += index[r] == 2; }
  summary[3] = na;
  summary[4] = nn;
  fill_lists(index, R, la, ln);
  statistic(post, R, M, K, cov, missing, la, na, ln, nn, U1, U0, stat);
  srand(seed);
  float mx = stat[0];
  for (int s = 1; s < M; s++) if (stat[s] > mx) mx = stat[s];
  xmax[0] = mx;
  for (int i = 1; i < nsim; i++) {
    permute_lists(n0 + n1, n1, rand(), index);
    fill_lists(index, R, la, ln);
    statistic(post, R, M, K, cov, missing, la, na, ln, nn, U1, U0, Xi);
    mx = Xi[0];
    for (int s = 0; s < M; s++) {
      if (Xi[s] >= stat[s]) psim[s]++;
      if (Xi[s] > mx) mx = Xi[s];
    }
    xmax[i] = mx;
  }
  float pmax = 1;
  for (int i = 1; i < nsim; i++) if (xmax[0] <= xmax[i]) pmax++;
  summary[0] = pmax / nsim;
  float ps = 0;
  for (int s = 0; s < M; s++) ps += psim[s];
  summary[1] = ps / (float)nsim;
  float *sorted = (float *)malloc(sizeof(float) * nsim);
  memcpy(sorted, xmax, sizeof(float) * nsim);
  for (int j = 0; j < nsim; j++) /* sort(), :1180-1193 */
    for (int i = j; i < nsim; i++)
      if (sorted[i] < sorted[j]) { float t = sorted[i]; sorted[i] = sorted[j]; sorted[j] = t; }
  float fs = (float)sig; /* functionPars2::sig is a float (src/types.h:373) */
  int at = (int)round((double)((1 - fs) * nsim)) - 1;
  summary[2] = (at >= 0 && at < nsim) ? sorted[at] : NAN;
  free(sorted); free(index); free(la); free(ln); free(U1); free(U0); free(Xi); free(psim);
  return 0;
}
