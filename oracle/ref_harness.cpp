/*
 * TEST INFRASTRUCTURE -- not part of the product.
 *
 * C-ABI harness around the *real* reference implementation.  It is compiled
 * together with the reference's own source files where they lie under
 * /root/reference/src (see oracle/Makefile, target `ref`); the result is
 * oracle/_ref/librelate_ref.so.  No reference source is copied into this
 * repository: this file only declares the prototypes of the reference's
 * external-linkage functions and marshals flat arrays into the reference's
 * row-pointer containers (src/types.h:26-180).
 *
 * Used by: tests/ (to pin oracle/relate_oracle.c and to generate
 * tests/golden/*), and bench.py's cpu_baseline leg (kind "reference").
 *
 * Determinism: the reference seeds rand() with time(0) in run_optimization
 * (src/relateHMM.cpp:1853-1858).  This harness defines its own `time` and the
 * library is linked with -Wl,-Bsymbolic-functions, so calls *inside this .so*
 * bind to refh_time below and a test can pin the seed; nothing outside the
 * .so is affected.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#ifndef _types_h
#define _types_h
#include "types.h"
#endif
#ifndef _alloc_h
#define _alloc_h
#include "alloc.h"
#endif
#include "ld.h"
#define private public /* the harness needs relateHMM's static globals */
#include "relateHMM.h"
#undef private
extern "C" {
#include "bfgs.h"
}

/* external-linkage functions of src/relateHMM.cpp that have no header */
double like(const double *delta, const dArray *Sk1, const dArray *Sk2, const dArray *Sk3, const dArray *t,
            int double_recom, const dArray *alim);                                   /* :1426 */
dMatrix *decodek2(const dArray *ptr, const dArray *Sk1, const dArray *Sk2, const dArray *Sk3, const dArray *t,
                  int double_recom);                                                  /* :1012 */
dMatrix *decode(const dArray *ptr, const dArray *Sk1, const dArray *Sk2, const dArray *Sk3, const dArray *t,
                int double_recom);                                                    /* :1163 */
iArray *viterbi(const dArray *ptr, const dArray *Sk1, const dArray *Sk2, const dArray *Sk3, const dArray *t,
                int double_recom);                                                    /* :1289 */
dMatrix *errorNoLD(iArray *ind1, iArray *ind2, int snp, double epsilon);              /* :2861 */
dMatrix *emission2_new(dArray *maf, dArray *maf2, dMatrix *dist);                     /* :2776 */
dMatrix *LDemission_new(dArray *pBA, dArray *pbA, dArray *pBa, dArray *pba, dMatrix *ep); /* :2343 */
dMatrix *errorLD(iArray *ind1, iArray *ind2, iArray *ind1t, iArray *ind2t, int snp, double epsilon); /* :2884 */
bArray *pruning3(dMatrix *mea, int back, double prune_val);                           /* :669 */
dMatrix *getMea(iMatrix *dat, int ld_choose, int back);                               /* :651 */
dMatrix *remove_cross(const dMatrix *mea, const bArray *keepList, int back);          /* :850 */
iArray *getHighestId(dMatrix *matr);                                                  /* :481 */
double calculateA(double k0, double k1, double k2, double phi);                       /* :974 */
dArray *calcMaf(iMatrix *matrix);                                                     /* :2985 */
dArray *diff(dArray *var);                                                            /* :391 */

extern int print_info;

static time_t g_fake_time = 1234567;
extern "C" time_t time(time_t *t) { /* binds only inside this .so (-Bsymbolic-functions) */
  if (t) *t = g_fake_time;
  return g_fake_time;
}

namespace {
dArray *mk_d(const double *p, int n) {
  dArray *a = allocDoubleArray(n);
  for (int i = 0; i < n; i++) a->array[i] = p[i];
  return a;
}
iArray *mk_i(const int *p, int n) {
  iArray *a = allocIntArray(n);
  for (int i = 0; i < n; i++) a->array[i] = p[i];
  return a;
}
iMatrix *mk_im(const int *p, int x, int y) {
  iMatrix *m = allocIntMatrix(x, y);
  for (int i = 0; i < x; i++)
    for (int j = 0; j < y; j++) m->matrix[i][j] = p[(size_t)i * y + j];
  return m;
}
dMatrix *mk_dm(const double *p, int x, int y) {
  dMatrix *m = allocDoubleMatrix(x, y);
  for (int i = 0; i < x; i++)
    for (int j = 0; j < y; j++) m->matrix[i][j] = p[(size_t)i * y + j];
  return m;
}
void out_dm(const dMatrix *m, double *p) {
  for (int i = 0; i < m->x; i++)
    for (int j = 0; j < m->y; j++) p[(size_t)i * m->y + j] = m->matrix[i][j];
}

struct Session {
  functionPars *pars = nullptr;
  relateHMM *obj = nullptr;
  bool live = false;
} S;
}  // namespace

extern "C" {

/* mirrors the scalars of functionPars (src/types.h:232-276) */
struct refh_opts {
  double min, epsilon, prune_val, alim0, alim1, fixA_val, fixK_val, phi, convTol, par[3];
  int back, back2, doPrune, LD, ld_adj, double_recom, fixA, fixK, calcA, timesToConverge, timesToRun, doPar;
};

void refh_set_time(long t) { g_fake_time = (time_t)t; }

/* ---- kernels on caller data -------------------------------------------------- */
double refh_like(const double *delta, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                 int double_recom, const double *alim) {
  dArray *a1 = mk_d(Sk1, m), *a2 = mk_d(Sk2, m), *a3 = mk_d(Sk3, m), *tt = mk_d(t, m + 1), *al = mk_d(alim, 2);
  double r = like(delta, a1, a2, a3, tt, double_recom, al);
  killArray(a1); killArray(a2); killArray(a3); killArray(tt); killArray(al);
  return r;
}

/* post: 3*m row-major (row0 IBD2, row1 IBD1, row2 IBD0); use_k2: 1 decodek2, 0 decode, -1 as sgl (:3514) */
void refh_decode(const double *par, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                 int double_recom, int use_k2, double *post) {
  dArray *p = mk_d(par, 3), *a1 = mk_d(Sk1, m), *a2 = mk_d(Sk2, m), *a3 = mk_d(Sk3, m), *tt = mk_d(t, m + 1);
  if (use_k2 < 0) use_k2 = (par[1] != 0);
  dMatrix *r = use_k2 ? decodek2(p, a1, a2, a3, tt, double_recom) : decode(p, a1, a2, a3, tt, double_recom);
  if (!use_k2)
    for (int i = 0; i < m; i++) r->matrix[0][i] = 0; /* decode() leaves row 0 unset (:1247-1251) */
  out_dm(r, post);
  killMatrix(r);
  killArray(p); killArray(a1); killArray(a2); killArray(a3); killArray(tt);
}

void refh_viterbi(const double *par, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                  int double_recom, int *path) {
  dArray *p = mk_d(par, 3), *a1 = mk_d(Sk1, m), *a2 = mk_d(Sk2, m), *a3 = mk_d(Sk3, m), *tt = mk_d(t, m + 1);
  iArray *r = viterbi(p, a1, a2, a3, tt, double_recom);
  for (int i = 0; i < m; i++) path[i] = r->array[i];
  killArray(r);
  killArray(p); killArray(a1); killArray(a2); killArray(a3); killArray(tt);
}

/* S1 = emission2_new(1-maf, maf, errorNoLD(...)) as in sgl (:3439-3442); maf = calcMaf value; out 3*m */
void refh_emission_single(const int *ind1, const int *ind2, const double *maf, int m, double eps, double *S1) {
  iArray *i1 = mk_i(ind1, m), *i2 = mk_i(ind2, m);
  dArray *m2 = mk_d(maf, m);
  dArray *m1 = allocDoubleArray(m);
  for (int i = 0; i < m; i++) m1->array[i] = 1.0 - m2->array[i];
  dMatrix *ep = errorNoLD(i1, i2, m, eps);
  dMatrix *r = emission2_new(m1, m2, ep);
  out_dm(r, S1);
  killMatrix(r); killMatrix(ep);
  killArray(i1); killArray(i2); killArray(m1); killArray(m2);
}

/* raw 81-term polynomial: ep is 81*n row-major, out 3*n (row0 IBD0,row1 IBD1,row2 IBD2) */
void refh_ldemission(const double *pBA, const double *pbA, const double *pBa, const double *pba, const double *ep,
                     int n, double *out) {
  dArray *a = mk_d(pBA, n), *b = mk_d(pbA, n), *c = mk_d(pBa, n), *d = mk_d(pba, n);
  dMatrix *e = mk_dm(ep, 81, n);
  dMatrix *r = LDemission_new(a, b, c, d, e);
  out_dm(r, out);
  killMatrix(r); killMatrix(e);
  killArray(a); killArray(b); killArray(c); killArray(d);
}

/* errorLD weights (81*n) for observed genotypes: cur1,cur2 current SNP, prev1,prev2 conditioning SNP */
void refh_error_ld(const int *cur1, const int *cur2, const int *prev1, const int *prev2, int n, double eps,
                   double *ep) {
  /* errorLD shifts ind1/ind2 left by one (:2885-2886): prepend a dummy element */
  std::vector<int> c1(n + 1, 1), c2(n + 1, 1);
  for (int i = 0; i < n; i++) { c1[i + 1] = cur1[i]; c2[i + 1] = cur2[i]; }
  iArray *i1 = mk_i(c1.data(), n + 1), *i2 = mk_i(c2.data(), n + 1), *p1 = mk_i(prev1, n), *p2 = mk_i(prev2, n);
  dMatrix *r = errorLD(i1, i2, p1, p2, n, eps);
  out_dm(r, ep);
  killMatrix(r);
  killArray(i1); killArray(i2); killArray(p1); killArray(p2);
}

/* one LD cell (src/ld.cpp:527): out = {pBA,pBa,pbA,pba,D,rsq2,dprime,lod,p,total} */
void refh_ld_cell(const int *first, const int *second, int n, double *out) {
  std::vector<int> a(first, first + n), b(second, second + n);
  geno_cptr r = get_geno_count(a.data(), b.data(), n);
  double T2 = r->total * 2.0;
  out[0] = r->u / T2; out[1] = r->v / T2; out[2] = r->w / T2; out[3] = r->x / T2;
  out[4] = r->bigD / (T2 * T2); out[5] = r->rsq2; out[6] = r->dprime; out[7] = r->lod;
  out[8] = r->p; out[9] = r->total;
  free(r->expt);
  free(r);
}

/* snp_pair_range on an N x M matrix (row-major); each out table depth*(M-1) */
void refh_snp_pair_range(const int *geno, int N, int M, int depth, double *pBA, double *pBa, double *pbA,
                         double *pba, double *D, double *rsq2) {
  iMatrix *g = mk_im(geno, N, M);
  snpMatrix *r = snp_pair_range(g, depth);
  out_dm(r->pBA, pBA); out_dm(r->pBa, pBa); out_dm(r->pbA, pbA); out_dm(r->pba, pba);
  out_dm(r->D, D); out_dm(r->rmisc, rsq2);
  killSnpMatrix(r);
  killMatrix(g);
}

/* pruning3 on getMea(geno) -> keep[M] ; returns number kept */
int refh_prune(const int *geno, int N, int M, int LD, int back, double prune_val, int *keep) {
  iMatrix *g = mk_im(geno, N, M);
  dMatrix *mea = getMea(g, LD, back);
  bArray *k = pruning3(mea, back, prune_val); /* frees mea */
  int n = k->numTrue;
  for (int i = 0; i < M; i++) keep[i] = k->array[i];
  killArray(k);
  killMatrix(g);
  return n;
}

/* remove_cross: mea is depth x cols row-major, keep has cols+1 entries; out back x (numTrue-1) */
void refh_remove_cross(const double *mea, int depth, int cols, const int *keep, int back, double *out) {
  dMatrix *m = mk_dm(mea, depth, cols);
  bArray *k = allocBoolArray(cols + 1);
  int nt = 0;
  for (int i = 0; i < cols + 1; i++) { k->array[i] = keep[i]; nt += keep[i] != 0; }
  k->numTrue = nt;
  dMatrix *r = remove_cross(m, k, back);
  out_dm(r, out);
  killMatrix(r); killMatrix(m); killArray(k);
}

double refh_calculateA(double k0, double k1, double k2, double phi) { return calculateA(k0, k1, k2, phi); }

void refh_calc_maf(const int *geno, int N, int M, double *maf) {
  iMatrix *g = mk_im(geno, N, M);
  dArray *r = calcMaf(g);
  for (int i = 0; i < M; i++) maf[i] = r->array[i];
  killArray(r);
  killMatrix(g);
}

/* the reference's L-BFGS-B driver on an arbitrary callback (src/bfgs.c:127) */
typedef double (*refh_fun)(double *);
double refh_findmax_bfgs(int n, double *x, refh_fun f, double *lo, double *hi) {
  return findmax_bfgs(n, x, f, NULL, lo, hi, NULL, -1);
}

/* glibc rand stream as consumed by the reference (for pinning the restated generator) */
void refh_rand_stream(unsigned seed, int n, int *out) {
  srand(seed);
  for (int i = 0; i < n; i++) out[i] = rand();
}

/* ---- whole-path session: init_globals + sgl --------------------------------- */
int refh_init(const int *geno, int N, int Mraw, const double *pos, const int *chr, const refh_opts *o) {
  if (S.live) return -1;
  functionPars *p = new functionPars();
  p->data = mk_im(geno, N, Mraw);
  p->pair = allocIntArray(2);
  p->chr = mk_i(chr, Mraw);
  p->position = mk_d(pos, Mraw);
  p->par = mk_d(o->par, 3);
  p->alim = allocDoubleArray(2);
  p->alim->array[0] = o->alim0;
  p->alim->array[1] = o->alim1;
  p->doAllPairs = 1;
  p->doPar = o->doPar;
  p->doPrune = o->doPrune;
  p->back2 = o->back2;
  p->fixA = o->fixA;
  p->fixK = o->fixK;
  p->calcA = o->calcA;
  p->double_recom = o->double_recom;
  p->LD = o->LD;
  p->ld_adj = o->ld_adj;
  p->print_results = 0;
  p->phi = o->phi;
  p->calcA_val = 0;
  p->min = o->min;
  p->epsilon = o->epsilon;
  p->back = o->back;
  p->timesToRun = o->timesToRun;
  p->timesToConverge = o->timesToConverge;
  p->prune_val = o->prune_val;
  p->fixA_val = o->fixA_val;
  p->fixK_val = o->fixK_val;
  p->convTol = o->convTol;
  p->moment = 0;
  S.pars = p;
  S.obj = new relateHMM();
  S.obj->init_globals(p);
  S.live = true;
  return relateHMM::genos_global->y;
}

/* keep[Mraw]; genos N*M; maf[M] (= calcMaf value, "maf1_global"); pos[M] */
void refh_globals(int *keep, int *genos, double *maf, double *pos) {
  const bArray *k = relateHMM::pre_calc_used_list;
  for (int i = 0; i < k->x; i++) keep[i] = k->array[i];
  const iMatrix *g = relateHMM::genos_global;
  for (int i = 0; i < g->x; i++)
    for (int j = 0; j < g->y; j++) genos[(size_t)i * g->y + j] = g->matrix[i][j];
  for (int j = 0; j < g->y; j++) {
    maf[j] = relateHMM::maf1_global->array[j];
    pos[j] = relateHMM::pos_global->array[j];
  }
}

/* hap_global after the reversals in pre_calc (:3340-3354): mea is back2 x (M-1); hap tables back2 x M */
int refh_hap_global(double *mea, double *pBA, double *pBa, double *pbA, double *pba) {
  hapStruct *h = relateHMM::hap_global;
  if (!h) return 0;
  out_dm(h->mea, mea); out_dm(h->pBA, pBA); out_dm(h->pBa, pBa); out_dm(h->pbA, pbA); out_dm(h->pba, pba);
  return h->mea->x;
}

/*
 * one pair through sgl (:3408).  scal = {a,k2,k1,k0,kLike,uLike,poLike,timesRun,timesConv,m}.
 * keep[M]; t[m+1]; S1,S,post 3*m each (row-major, row stride m); choose[m]; path[m];
 * conv: timesToRun*4 (rows beyond timesRun are garbage in the reference -> zeroed here).
 * Any output pointer may be NULL.
 */
int refh_sgl(int i, int j, long seed, double *scal, int *keep, double *t, double *S1, double *Sm, int *choose,
             double *post, int *path, double *conv) {
  if (!S.live) return -1;
  g_fake_time = (time_t)seed;
  S.pars->pair->array[0] = i;
  S.pars->pair->array[1] = j;
  fullResults *r = S.obj->sgl(S.pars);
  int m = r->snp;
  if (scal) {
    scal[0] = r->a; scal[1] = r->kResult->array[0]; scal[2] = r->kResult->array[1]; scal[3] = r->kResult->array[2];
    scal[4] = r->kLike; scal[5] = r->uLike; scal[6] = r->poLike; scal[7] = r->timesRun; scal[8] = r->timesConverged;
    scal[9] = m;
  }
  if (keep) for (int s = 0; s < r->keepList->x; s++) keep[s] = r->keepList->array[s];
  if (t) for (int s = 0; s < r->t->x; s++) t[s] = r->t->array[s];
  if (S1) out_dm(r->S1, S1);
  if (Sm) out_dm(r->S, Sm);
  if (choose) {
    if (r->choose) for (int s = 0; s < r->choose->x; s++) choose[s] = r->choose->array[s];
    else for (int s = 0; s < m; s++) choose[s] = -1;
  }
  if (post) {
    if (r->kResult->array[0] == 0) for (int s = 0; s < m; s++) r->post->matrix[0][s] = 0; /* decode(): row 0 unset */
    out_dm(r->post, post);
  }
  if (path) for (int s = 0; s < m; s++) path[s] = r->path->array[s];
  if (conv && r->convInfo) {
    for (int a = 0; a < r->convInfo->x; a++)
      for (int b = 0; b < 4; b++) conv[a * 4 + b] = a < r->timesRun ? r->convInfo->matrix[a][b] : 0.0;
  }
  killFullResults(r);
  return m;
}

void refh_fini() {
  if (!S.live) return;
  S.obj->dinit_globals(S.pars->ld_adj ? S.pars->back : 0);
  delete S.obj;
  killFunctionPars(S.pars);
  S = Session();
}

} /* extern "C" */
