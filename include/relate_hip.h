/*
 * relate_hip.h -- C ABI of the MI355X-native all-pairs IBD HMM path (librelate_hip.so).
 *
 * Drop-in boundary for aalbrechtsen/relate's per-pair loop.  The reference has no plugin/FFI layer;
 * its de-facto operator interface is relateHMM::init_globals / sgl / dinit_globals
 * (src/relateHMM.h:58-79) called once per pair from fast_all_pairs / joblist_fast_all_pairs
 * (src/relateHMM.cpp:3764-3785, 3851-3875), plus the objective callback handed to findmax_bfgs
 * (src/bfgs.h:63-66).  The entry points below replace those calls with batched ones; each cites the
 * reference interface it stands for.  INTEGRATION.md shows the binding a maintainer adds to
 * src/relateHMM.cpp.
 *
 * Conventions: every function returns 0 on success or a negative RELATE_E_* code; the message is
 * available from relate_hip_last_error().  All buffers are caller-owned host memory unless a name
 * starts with d_.  No torch types, no global state: one handle per GPU, used from one host thread.
 * Genotype codes are the reference's: 0 missing, 1/2/3 = 0/1/2 copies (src/relateHMM.cpp:2985-3007).
 * Row order of posteriors follows decodek2 (src/relateHMM.cpp:1133-1137): row0 IBD2, row1 IBD1, row2 IBD0.
 */
#ifndef RELATE_HIP_H
#define RELATE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELATE_OK 0
#define RELATE_E_ARG -1     /* bad argument / call order */
#define RELATE_E_HIP -2     /* HIP runtime error */
#define RELATE_E_NODEV -3   /* no usable gfx950 device */
#define RELATE_E_NUMERIC -4 /* NaN in the recursion (the reference prints and exit(0)s, :1513-1520) */
#define RELATE_E_NOMEM -5
#define RELATE_E_NOSNP -6   /* relate_fit.status: the pair shares no non-missing SNP (the reference stops, :3415-3421) */

#define RELATE_SYS_MAX 10000000.0 /* src/conf.h:6 -- value of an infeasible objective */

typedef struct relate_hip_ctx *relate_hip_t;

/* Scalars of functionPars (src/types.h:232-276) after getOptions (src/filereader_and_conversions.cpp:65-91)
 * and post_processing_of_option_pars (src/runRelate.cpp:76-101; back2 = 2*back is applied by the library). */
typedef struct {
  double min;          /* MAF filter: keep min < maf < 1-min                 */
  double epsilon;      /* genotyping error rate                               */
  double prune_val;    /* LD pruning threshold                                */
  double alim[2];      /* bounds of a                                         */
  double fixA_val, fixK_val;
  double phi;          /* calculateA                                          */
  double convTol;
  double par[3];       /* (a,k2,k1) used when doPar                           */
  int back;            /* LD window; back2 = 2*back                           */
  int doPrune;
  int LD;              /* 1 = r^2, 0 = |D| (already un-inverted, :69)         */
  int ld_adj;          /* LD-conditioned emissions on/off                     */
  int double_recom;
  int fixA, fixK, calcA;
  int timesToConverge, timesToRun;
  int doPar;
  int seed_mode;       /* 0: srand(seed) per pair, glibc-compatible stream (reference: seed = time(0), :1853);
                          1: srand(seed + pair index)                          */
  unsigned int seed;
} relate_opts;

/* Per-pair result record: what sgl returns in fullResults (src/relateHMM.cpp:3527-3572). */
typedef struct {
  double a, k2, k1, k0; /* kResult = {k2,k1,k0}, a                          */
  double negloglik;     /* kLike                                            */
  double uLike, poLike; /* like at (alim0,0,0) and (alim0,0,1), :3478-3481  */
  int timesRun, timesConverged;
  int nsnp;             /* SNPs kept for the pair                           */
  int status;           /* 0 ok, RELATE_E_NUMERIC if the recursion produced NaN, RELATE_E_NOSNP if nsnp == 0 (not fitted) */
} relate_fit;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* relateHMM object construction (src/runRelate.cpp:404). device_id = HIP device ordinal. */
int relate_hip_create(relate_hip_t *out, int device_id);
int relate_hip_destroy(relate_hip_t h);
const char *relate_hip_last_error(relate_hip_t h); /* h may be NULL: error of a failed create */
/* Result buffers: the reference's callee allocates them (allocDoubleMatrix / allocDoubleArray, src/alloc.cpp:62-309)
 * and the caller frees them with kill*; here buffers are caller-owned, and a caller that wants posterior rows and
 * paths to arrive at the speed of the host link allocates them with this pair (page-locked memory: the device
 * writes it directly).  Any other host memory works too and is staged through the handle's own pinned buffers. */
int relate_hip_host_alloc(void **out, size_t bytes);
int relate_hip_host_free(void *p);

/* ---- data set: init_globals -> pre_calc (src/relateHMM.cpp:3371-3392, 3293-3362) --------------- */
/* geno: N x M_raw row-major (individual-major, as iMatrix data), values 0..3; pos/chr: M_raw, already
 * sorted by (chr,pos) as mysort leaves them (src/asort.cpp:37-88). */
int relate_hip_load(relate_hip_t h, const uint8_t *geno, int N, int M_raw, const double *pos, const int *chr,
                    const relate_opts *opts);
/* PLINK ingestion without the reference's int matrix: replaces bed_to_iMatrix + extractOK(keeplist) + mysort on the
 * genotype matrix (src/filereader_and_conversions.cpp:483-564, src/runRelate.cpp:363-381).  bed: the SNP-major .bed
 * payload after its 3-byte header (0x6C 0x1B 0x01), n_snp_bed rows of ceil(N/4) bytes.  cols[M_raw]: the .bed rows to
 * analyse, in analysis order -- the caller applies doBimFile's autosome filter (:634-640) and the (chr,pos) sort
 * (src/asort.cpp:37-88) to the short per-SNP vectors and passes the resulting order; pos (already / 1e6, src/conf.h) and
 * chr are in that order.  The 2-bit fields are recoded on the GPU by relate_hip_precalc ({1,0,2,3} then 1 <-> 3). */
int relate_hip_load_bed(relate_hip_t h, const uint8_t *bed, int N, int n_snp_bed, const int *cols, int M_raw,
                        const double *pos, const int *chr, const relate_opts *opts);
/* calcMaf + maf_stripper + [getMea + pruning3] + global MAF + snp_pair_range(back2) + emission tables.
 * keep_out (M_raw ints, may be NULL) = pre_calc_used_list; *M_out = SNPs kept. */
int relate_hip_precalc(relate_hip_t h, int *keep_out, int *M_out);
/* The stripped data the reference dumps (src/relateHMM.cpp:3745-3750): geno N x M (0..3), pos[M], maf[M]
 * (calcMaf value).  Any pointer may be NULL. */
int relate_hip_get_stripped(relate_hip_t h, uint8_t *geno, double *pos, double *maf);
/* hap_global (src/relateHMM.cpp:3340-3354): mea back2 x (M-1); pBA,pBa,pbA,pba back2 x M; choose_global[M]
 * (getHighestId of the first `back` rows, -1 for SNP 0).  Any pointer may be NULL. Inspection / tests. */
int relate_hip_get_hap(relate_hip_t h, double *mea, double *pBA, double *pBa, double *pbA, double *pba,
                       int *choose_global);

/* ---- per-pair emissions: front half of sgl (src/relateHMM.cpp:3424-3472) ------------------------ */
/* Inspection of one pair: keep[M]; t[M] (t of each kept SNP in stripped coordinates, 0 for the first,
 * -1 for dropped SNPs); S 3 x M row-major (row0 IBD0,row1 IBD1,row2 IBD2; 0 where dropped). */
int relate_hip_pair_emissions(relate_hip_t h, int ind1, int ind2, int *keep, double *t, double *S);

/* ---- batched kernels ----------------------------------------------------------------------------- */
/* like() (src/relateHMM.cpp:1426-1544) for P pairs x E parameter points: params is P x E x 3 (a,k2,k1),
 * negloglik is P x E.  This is the call issued once per optimiser round. */
int relate_hip_like(relate_hip_t h, const int32_t *pairs /* P x 2 */, int P, const double *params, int E,
                    double *negloglik);

/* relate_hip_like for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc).  No
 * counterpart in the reference.  Works on the pair block the previous relate_hip_like / decode call staged (P must be
 * that block's pair count), takes DEVICE pointers (d_params P x E x 3, pair-major as relate_hip_like; d_negloglik
 * P x E) and calls no hipMalloc / hipFree: the chunk products of a launch cut along the SNP axis and the half vectors
 * of the two-sided sweep go to a 64 MiB array every GPU the library runs on holds until the process ends (a launch is
 * cut into as many chunks as fit there; values agree with relate_hip_like to rounding, <= 1e-13 relative).  At most
 * 1 048 576 lanes (P x E) per call, more is rejected.  Runs on the handle's own stream and returns with the values in
 * place (it synchronises that stream, so it is not for use under stream capture).  A relate_hip_fit / run_pairs may
 * leave NO block staged (its tail re-stages the last live pairs side by side): the call is then rejected with a
 * message that names the staged pair count, and a relate_hip_like on the block stages it again.  Calls that overlap on
 * one GPU (two handles, two threads) race on that array. */
int relate_hip_like_noalloc(relate_hip_t h, int P, const double *d_params, int E, double *d_negloglik);

/* run_optimization + *_optim + findmax_bfgs (src/relateHMM.cpp:1852-1971, 1597-1849; src/bfgs.c:127-199),
 * plus uLike/poLike, batched over pairs.  starts: optional P x timesToRun x 4 start records
 * {a,k2,k1,unused} replacing the rand() draws of get_rand/get_start (NULL = draw per opts.seed_mode).
 * conv_out: optional P x timesToRun x 4 = convInfo ({a,k2,k1,negloglik} per run; rows >= timesRun zero). */
int relate_hip_fit(relate_hip_t h, const int32_t *pairs, int P, const double *starts, relate_fit *out,
                   double *conv_out);

/* decodek2 / decode (+ viterbi) (src/relateHMM.cpp:1012-1263, 1289-1420) at given parameters.
 * params: P x 3 (a,k2,k1).  post: P x 3 x M doubles in stripped SNP coordinates, -1 where the SNP is
 * dropped for the pair (as the .post writer prints, :126-133); may be NULL.  path: P x M int8
 * (0 = IBD0 ... 2 = IBD2, -1 dropped); may be NULL.  nsnp_out: P ints, may be NULL. */
int relate_hip_decode(relate_hip_t h, const int32_t *pairs, int P, const double *params, double *post,
                      int8_t *path, int *nsnp_out);

/* The same decode, only the rows the caller reads: post is P x rows x M doubles in the order the .post writer takes
 * the rows of decodek2's matrix (write_dMatrix_to_file of row 2 = IBD0, then row 1 = IBD1 unless fixK2 = 1 with value
 * 0, src/relateHMM.cpp:3771-3782): row 0 = IBD0, row 1 = IBD1, row 2 = IBD2; rows = 1..3; values identical to
 * relate_hip_decode's.  The reference's CLI prints two of decodek2's three rows (one in the linkage set-up); a row
 * nobody reads costs 8 B per pair and SNP in HBM and again over the host link. */
int relate_hip_decode_rows(relate_hip_t h, const int32_t *pairs, int P, const double *params, int rows, double *post,
                           int *nsnp_out);

/* The same decode with the rows the .post writer keeps (write_dMatrix_to_file of row 2 = IBD0, then row 1 = IBD1
 * unless fixK2 = 1 with value 0, src/relateHMM.cpp:3771-3782) as floats: post is P x rows x M, row 0 = IBD0, row 1 =
 * IBD1, row 2 = IBD2 (rows = 1..3), each value the double posterior rounded once to float (7 digits against the 4 the
 * reference prints), -1 where the SNP is dropped.  A third of the bytes over the host link; the layout of the binary
 * shards of relate_amd.allpairs. */
int relate_hip_decode_f32(relate_hip_t h, const int32_t *pairs, int P, const double *params, int rows, float *post,
                          int *nsnp_out);

/* sgl for P pairs: fit (or doPar) then decode at the estimate -- the body of the reference's pair loop. */
int relate_hip_run_pairs(relate_hip_t h, const int32_t *pairs, int P, const double *starts, relate_fit *out,
                         double *post, int8_t *path);

/* findmax_bfgs (src/bfgs.h:63-66) with a host callback: same L-BFGS-B state machine and numeric gradient
 * (src/bfgs.c:12-100) that relate_hip_fit drives in batch; nbd = 2 for all variables, noisy = -1.
 * Returns the reference's return value (-f at the optimum) through *neg_f. No GPU involved. */
int relate_findmax_bfgs(int numpars, double *invec, double (*fun)(double *x, void *user), void *user,
                        const double *lowbound, const double *upbound, double *neg_f);

/* glibc-compatible rand() stream used for the start points (get_rand / get_start consume rand();
 * src/relateHMM.cpp:278-293, 1546-1560).  out[i] = i-th rand() after srand(seed). */
void relate_rand_stream(unsigned int seed, int n, int *out);

/* ---- output text ---------------------------------------------------------------------------------- */
/* write_dMatrix_to_file (src/relateHMM.cpp:117-136, POSTPRECISION src/conf.h:8) for a block of posteriors as
 * relate_hip_decode returns them (P x 3 x M): per pair the IBD0 row, then the IBD1 row unless k1_only (fixK2 = 1 with
 * value 0, :3771); each value as "%.4g" + tab, "-1" for dropped SNPs, newline after the row -- the bytes the reference's
 * ostream writes.  Host threads only, no GPU.  out: (P * rows) x relate_post_text_stride(M) bytes; row r starts at
 * out + r * stride and is row_len[r] bytes long (no terminator). */
size_t relate_post_text_stride(int M);
int relate_format_post_rows(const double *post, int P, int M, int k1_only, char *out, size_t *row_len);
/* The same text for nrows consecutive rows of M values each (what relate_hip_decode_rows returns: P x rows x M is
 * P * rows rows in the writer's order): row r of the input -> out + r * stride, row_len[r] bytes. */
int relate_format_rows(const double *rows, long nrows, int M, char *out, size_t *row_len);

/* ---- multi-GPU: contiguous pair-block partition (no data-path collective) ------------------------ */
/* Rank r of nranks owns pairs [*begin, *end) of a list of P pairs. */
void relate_hip_shard(int64_t P, int rank, int nranks, int64_t *begin, int64_t *end);

/* The path's one exchange: all-gather of the per-pair result records, rank r contributing the records of its
 * relate_hip_shard(P_total, r, nranks) range; `all` (P_total records) is complete on every rank when the call
 * returns.  The reference has no counterpart: its pair loop is one process (src/relateHMM.cpp:3764-3785); this is
 * the seam at which that loop is cut into pair blocks per GPU.  transport 0: RCCL (xGMI inside a node) on the
 * handle's GPU and stream -- one communicator per handle, made at the first call from a ncclUniqueId that rank 0
 * leaves in the file <rendezvous>.g<call>.id; librccl.so is looked up at run time (no link dependency).  transport 1:
 * the records travel through files <rendezvous>.g<call>.fit.<rank> on a filesystem every rank sees (h may be NULL):
 * for ranks that share one GPU, and for tests.  rendezvous may be NULL when nranks == 1.  Waits for the other ranks
 * up to RELATE_GATHER_TIMEOUT seconds (default 1800). */
int relate_hip_gather(relate_hip_t h, const relate_fit *mine, int64_t P_total, int rank, int nranks,
                      const char *rendezvous, int transport, relate_fit *all);

/* ---- linkage statistic + permutation test on posterior rows (next row of the path: the reference's HMMtest) ------ */
/* uber_main, case-control design (src/HMMtest.txt:1195-1367).  post: R x M floats, row p = the IBD0 posterior row of
 * pair p as the .post file holds it (-1 = SNP dropped for the pair), pairs in the all-pairs order of the n_ind
 * individuals the -D file lists, so R = n_ind (n_ind - 1) / 2.  pos: M floats or NULL (needed when infer = 1:
 * fix_missing, :550-664).  K: R floats or NULL (NULL = row means, :941-968; isK0: use 1 - K, :674-681).
 * cc: the -D file's values, one per individual (0 = affected, per getCaseList :717-723).
 * Outputs (any may be NULL): stat[M] = the statistic X0 written to -f; xmax[nsim] = the -x file (observed maximum
 * first, then one per permutation); covar[R x R] = the -c dump; post_out[R x M] = the -o dump (after inference).
 * Results equal the reference's float arithmetic bit for bit (same summation order per output value). */
typedef struct {
  int nsim;        /* -Nsim, permutations incl. the observed one (default 10)        */
  int ccAll;       /* -ccAll: 0 strict, 1 mixed pairs are cases, 2 mixed are controls */
  int infer;       /* -infer: 1 = interpolate missing posteriors (needs pos)          */
  int isK0;        /* K holds k0 rather than k1 (-K0)                                  */
  float sig;       /* -sig (default 0.05)                                              */
  unsigned seed;   /* srand(seed) before the permutations (reference: time(NULL), :1296) */
} relate_linkage_opts;
typedef struct {
  double pmaxsim, psim_over_nsim, psigsim; /* the three printed summaries (:1349-1362) */
  int n_case_pairs, n_control_pairs;
} relate_linkage_result;
int relate_hip_linkage(relate_hip_t h, const float *post, int R, int M, const float *pos, const float *K, const int *cc,
                       int n_ind, const relate_linkage_opts *o, float *stat, float *xmax, float *covar, float *post_out,
                       relate_linkage_result *res);

/* The same statistic with the posterior rows decoded on the spot and never leaving HBM: decode (IBD0 row as float, the
 * row the linkage pipeline's .post file holds: fixK2 = 1 with value 0, src/relateHMM.cpp:3771-3782) of the R pairs at
 * params (R x 3: a, k2, k1 -- the estimates of relate_hip_fit), block by block into one R x M matrix on the device,
 * then uber_main's steps on it (src/HMMtest.txt:1195-1367).  Replaces the hand-off through the .post file between the
 * two programs.  digits = 4: every posterior is first rounded to the 4 significant digits that file carries
 * (POSTPRECISION, src/conf.h:8), so the statistic is the one HMMtest computes from the reference's output; digits = 0:
 * float posteriors as decoded.  pos NULL = the stripped positions of the handle's data set as floats.  Pairs in the
 * all-pairs order of the n_ind individuals of cc; other arguments and outputs as relate_hip_linkage (M = SNPs kept). */
int relate_hip_linkage_pairs(relate_hip_t h, const int32_t *pairs, int R, const double *params, int digits,
                             const float *pos, const float *K, const int *cc, int n_ind, const relate_linkage_opts *o,
                             float *stat, float *xmax, float *covar, float *post_out, relate_linkage_result *res);

#ifdef __cplusplus
}
#endif
#endif
