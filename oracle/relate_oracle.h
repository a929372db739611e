/*
 * TEST INFRASTRUCTURE -- CPU oracle for the all-pairs IBD HMM path of aalbrechtsen/relate.
 *
 * A plain-C restatement of the reference algorithm, used only as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under relate_amd/ may
 * include, link or call it.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).  Pinned against the compiled reference
 * (oracle/_ref, see oracle/Makefile) and against tests/golden/ by tests/test_oracle_*.py.
 *
 * Conventions: genotypes are uint8 in {0 missing,1,2,3}; matrices are flat row-major;
 * "maf" is the value of calcMaf (frequency of the allele counted by (g-1)/2).
 * Emission rows follow the reference: row0 = IBD0, row1 = IBD1, row2 = IBD2.
 * Posterior rows follow decodek2: row0 = IBD2, row1 = IBD1, row2 = IBD0.
 */
#ifndef RELATE_ORACLE_H
#define RELATE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SYS_MAX 10000000.0 /* src/conf.h:6 */

/* calcMaf, src/relateHMM.cpp:2985-3007 ; g is N x M */
void orc_maf(const uint8_t *g, int N, int M, double *maf);
/* maf_stripper, :3238-3250 ; returns number kept */
int orc_maf_keep(const double *maf, int M, double min, int *keep);

/* get_geno_count + do_geno_cal, src/ld.cpp:342-569.
 * first/second: genotype of the later / earlier SNP for n individuals, element stride `stride`.
 * out = {pBA,pBa,pbA,pba,D,rsq2,dprime,lod,p,total} in snp_pair_range's naming (:222-242). */
void orc_ld_cell(const uint8_t *first, const uint8_t *second, int n, int stride, double *out);

/* Global tables after pre_calc's reversals (:3340-3354): mea is depth x (M-1), mea[d][c] = LD measure
 * (rsq2 if LD else |D|) between SNP c+1 and SNP c-d; hap tables depth x M, hap[d][s] = haplotype
 * frequency of SNP s (A/a) with SNP s-1-d (B/b); zero where the earlier SNP does not exist. */
void orc_hap_global(const uint8_t *g, int N, int M, int depth, int LD, double *mea, double *pBA, double *pBa,
                    double *pbA, double *pba);

/* pruning3 on getMea(depth=back), :651-756 ; keep[M]; returns number kept */
int orc_prune(const uint8_t *g, int N, int M, int LD, int back, double prune_val, int *keep);

/* error() table, :2828-2855 : P(observed code o in 1..3 | true genotype index tr in 0..2) */
double orc_err(int tr, int o, double eps);
/* errorNoLD + emission2_new, :2861-2875, :2776-2822 : p = frequency for true genotype index 0 */
void orc_emis_single(double p, int o1, int o2, double eps, double *S3);
/* errorLD + LDemission_new (81 terms), :2884-2916, :2343-2704 ; o*c current SNP, o*p conditioning SNP */
void orc_emis_ld(double pBA, double pbA, double pBa, double pba, int o1c, int o2c, int o1p, int o2p, double eps,
                 double *S3);

/* Front half of sgl (:3424-3472) for one pair: which SNPs are kept, t, S1, choose, S.
 * keep[M]; t[m+1]; S1,S are 3 x m with row stride M (caller allocates 3*M); choose[m]. Returns m.
 * mea/hap tables as produced by orc_hap_global with depth back2 (ignored if !use_ld). */
int orc_pair_stage(const uint8_t *g, int N, int M, int i, int j, const double *maf, const double *pos, double eps,
                   int use_ld, int back, int back2, const double *mea, const double *pBA, const double *pBa,
                   const double *pbA, const double *pba, int *keep, double *t, double *S1, double *S, int *choose);

/* like(), :1426-1544 ; Sk1=log IBD2 emission, Sk2=log IBD1, Sk3=log IBD0 (length m); t length m+1 */
double orc_like(const double *delta, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                int double_recom, double alim1);
/* decodek2 (:1012-1157) when use_k2, else decode (:1163-1263); post is 3 x m (row stride m) */
void orc_decode(const double *par, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                int double_recom, int use_k2, double *post);
/* viterbi(), :1289-1420 */
void orc_viterbi(const double *par, const double *Sk1, const double *Sk2, const double *Sk3, const double *t, int m,
                 int double_recom, int *path);
/* calculateA, :974-1007 */
double orc_calculateA(double k0, double k1, double k2, double phi);

/* objective wrappers (:1564-1830) as one function: mode 0 full(a,k2,k1), 1 calcA(k2,k1), 2 fixK2(a,k1),
 * 3 fixK2_calcA(k1), 4 fixK2_fixA(k1).  Returns the value the reference's *_fun returns. */
double orc_objective(int mode, const double *x, double fixA_val, double fixK2_val, double phi, const double *Sk1,
                     const double *Sk2, const double *Sk3, const double *t, int m, int double_recom, double alim1);

#ifdef __cplusplus
}
#endif
#endif
