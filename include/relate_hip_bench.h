/*
 * relate_hip_bench.h -- measurement hooks of librelate_hip.so (bench.py, tools/): kernel timings with inputs resident
 * in HBM.  Not part of the drop-in boundary (include/relate_hip.h); nothing in the reference corresponds to them.
 */
#ifndef RELATE_HIP_BENCH_H
#define RELATE_HIP_BENCH_H
#include "relate_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Device time in ms of the kernels launched by the last like / decode / pair_emissions-stage call,
 * measured with hipEvents on the library's stream: out = {stage_ms, like_ms, decode_fwd_ms, decode_bwd_ms}. */
int relate_hip_last_timings(relate_hip_t h, double *out4);
/* The likelihood launches of the last relate_hip_fit: out = {point-steps of launches that sweep every point once (single
 * and two-sided sweeps), point-steps of launches cut along the SNP axis (three-row chunk products), their kernel ms,
 * their counts}; a point-step = one (pair, SNP, parameter point) update. */
int relate_hip_last_like_stats(relate_hip_t h, double *out6);
/* Number of SNP-axis chunks the last like / decode launch was cut into (1 = single sweep; > 1 adds the chunk-product
 * pass): needed to state the algorithmic bytes of a launch. */
int relate_hip_last_chunks(relate_hip_t h, int *like_chunks, int *decode_chunks);
/* Same computation as relate_hip_like / relate_hip_decode, but on the emission block staged by the
 * previous call for the same pairs and with results left on the device: used to time kernels with
 * inputs resident in HBM.  reps = number of back-to-back launches; ms_out = average ms per launch. */
int relate_hip_bench_like(relate_hip_t h, const int32_t *pairs, int P, const double *params, int E, int reps,
                          double *ms_out);
int relate_hip_bench_decode(relate_hip_t h, const int32_t *pairs, int P, const double *params, int reps,
                            double *ms_fwd_out, double *ms_bwd_out);

/* relate_hip_bench_decode with the rows of relate_hip_decode_rows: rows = 1..3 posterior rows in writer order (IBD0,
 * IBD1, IBD2), 0 = the reference's three rows. */
int relate_hip_bench_decode_rows(relate_hip_t h, const int32_t *pairs, int P, const double *params, int rows, int reps,
                                 double *ms_fwd_out, double *ms_bwd_out);
/* Ranks that took part in the handle's last relate_hip_gather: the size of the RCCL communicator as the library
 * reports it (ncclCommCount), or the number of ranks of the file session; 0 before any gather.  Lets a scaling run
 * show that the C-ABI collective really spanned N ranks. */
int relate_hip_last_gather_ranks(relate_hip_t h);

#ifdef __cplusplus
}
#endif
#endif
