// Linkage statistic and permutation test on posterior rows: the reference's downstream tool HMMtest
// (src/HMMtest.txt) on the GPU.  The reference works in float and sums sequentially; every kernel here keeps the
// reference's operation order per output value (one thread owns one sum and walks it in the reference's order,
// multiplications and additions are not contracted), so results equal the reference's bit for bit while rows,
// SNPs, matrix tiles and permutations run in parallel.  gfx950 only; HBM/L2-bound integer-indexed float work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace relate {

// fix_missing (src/HMMtest.txt:550-664): one thread per posterior row (the interior sweep is sequential: an inferred
// value is the left neighbour of the next gap).  Quirks kept: the right gap is difPos[j+1], "dif2 =+", the weights.
__global__ void lk_fix_missing_kernel(float *__restrict__ post, int R, int M, const float *__restrict__ pos) {
#pragma clang fp contract(off)
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float *p = post + (size_t)r * M;
  auto dif = [&](int k) -> float { return k <= M - 2 ? pos[k + 1] - pos[k] : 0.0f; };
  if (p[0] == -1) {
    int n = 1;
    while (n < M && p[n] == -1) n++;
    if (n == M) return;  // the whole row is missing (a pair without a shared SNP): nothing to infer from, it stays -1
    float nx = p[n];
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
      float dif2 = dif(j + 1);
      int o = 1;
      while (next == -1 && j + o < M - 1) {
        o++;
        dif2 = +dif(j + o);
        next = p[j + o];
      }
      float dif1 = dif(j - 1);
      float sum = dif1 + dif2;
      float t1 = (dif1 / sum) * first, t2 = (dif2 / sum) * next;
      if (dif1 < 0) { t1 = 0; t2 = next; }
      else if (dif2 < 0) { t2 = 0; t1 = first; }
      p[j] = t1 + t2;
    }
  }
}

// row_wise_mean_{no,with}_missing (:941-968), or K from a file (getKthorfinn, :674-681)
__global__ void lk_row_mean_kernel(const float *__restrict__ post, int R, int M, int missing, float *__restrict__ K) {
#pragma clang fp contract(off)
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float *p = post + (size_t)r * M;
  float s = 0;
  int m = 0;
  for (int j = 0; j < M; j++) {
    float v = p[j];
    if (!missing || v != -1) { s += v; m++; }
  }
  K[r] = s / (float)(missing ? m : M);
}

// calculateCoVar (:842-907): cov[i][j] = sum_s (post[i][s]-K[i]) (post[j][s]-K[j]) / (M-1) (or over the SNPs present
// in both rows).  16 x 16 output tile per workgroup, the SNP axis staged through LDS 32 at a time; each thread
// accumulates its own element over s ascending, as the reference does.
static constexpr int kCovT = 16, kCovK = 32;
__global__ void __launch_bounds__(256)
lk_covar_kernel(const float *__restrict__ post, const float *__restrict__ K, int R, int M, int missing,
                float *__restrict__ cov) {
#pragma clang fp contract(off)
  __shared__ float A[kCovT][kCovK + 1], B[kCovT][kCovK + 1];
  int ti = blockIdx.y * kCovT, tj = blockIdx.x * kCovT;
  int ly = threadIdx.x / kCovT, lx = threadIdx.x % kCovT;
  int i = ti + ly, j = tj + lx;
  float ki = i < R ? K[i] : 0.0f, kj = j < R ? K[j] : 0.0f;
  float s = 0;
  int m = 0;
  for (int k0 = 0; k0 < M; k0 += kCovK) {
    for (int e = threadIdx.x; e < kCovT * kCovK; e += 256) {
      int row = e / kCovK, col = e % kCovK;
      int k = k0 + col;
      A[row][col] = (ti + row < R && k < M) ? post[(size_t)(ti + row) * M + k] : 0.0f;
      B[row][col] = (tj + row < R && k < M) ? post[(size_t)(tj + row) * M + k] : 0.0f;
    }
    __syncthreads();
    int kn = min(kCovK, M - k0);
    for (int k = 0; k < kn; k++) {
      float a = A[ly][k], b = B[lx][k];
      if (!missing || (a != -1 && b != -1)) {
        s += (a - ki) * (b - kj);
        m++;
      }
    }
    __syncthreads();
  }
  if (i < R && j < R) cov[(size_t)i * R + j] = missing ? (float)((double)s / ((double)(float)m - 1.0)) : (float)((double)s / ((double)M - 1.0));
}

// Per permutation the two pair lists la[perm][LA] (case pairs) and ln[perm][LN] (control pairs), exactly as the
// reference holds them (fixed lengths; see relate_hip.hip for how their tails can carry entries of the previous
// permutation).  Column means over the lists (column_wise_mean_*, :909-935): one (permutation, SNP) per thread,
// walking the list in order.
__global__ void __launch_bounds__(256)
lk_col_means_kernel(const float *__restrict__ post, int M, const int *__restrict__ la, int LA,
                    const int *__restrict__ ln, int LN, int missing, float *__restrict__ U1, float *__restrict__ U0) {
#pragma clang fp contract(off)
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int perm = blockIdx.y;
  if (s >= M) return;
  const int *a = la + (size_t)perm * LA, *n = ln + (size_t)perm * LN;
  float s1 = 0, s0 = 0;
  int m1 = 0, m0 = 0;
  for (int j = 0; j < LA; j++) {
    float v = post[(size_t)a[j] * M + s];
    if (!missing || v != -1) { s1 += v; m1++; }
  }
  for (int j = 0; j < LN; j++) {
    float v = post[(size_t)n[j] * M + s];
    if (!missing || v != -1) { s0 += v; m0++; }
  }
  U1[(size_t)perm * M + s] = s1 / (float)(missing ? m1 : LA);
  U0[(size_t)perm * M + s] = s0 / (float)(missing ? m0 : LN);
}

// Per permutation scalars: means of K over the lists (get_mean, :831-840), sums of the covariance sub-matrices
// (sum_of_covar_matrix, :972-979: row-major double loop over the list) and the pooled denominator (:1279).
// out[perm] = {mn1, mn0, bottom}.  The sums are float accumulations in the reference's order, so each is one serial
// chain of LA^2 (LN^2) additions -- but its terms are gathers that do not depend on it: one wave per permutation, the
// 64 lanes fetch the next 64 terms of the row-major order while the chain (carried redundantly by every lane) takes
// the previous 64 out of the lanes one by one.  (One lane per permutation, gathers and additions in one loop, took
// 110-180 ms for 1 000 permutations of 1 770 rows; this takes a few.)
__device__ inline float lk_ordered_sum64(float acc, float v, int n) {  // acc + v[lane 0] + v[lane 1] + ... (n <= 64 terms)
#pragma clang fp contract(off)
  if (n == 64) {
#pragma unroll
    for (int t = 0; t < 64; t++) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
  } else {
    for (int t = 0; t < n; t++) acc += __shfl(v, t, 64);
  }
  return acc;
}
// sum over the row-major double loop of cov[l[i]][l[j]], i, j < L (l in LDS)
__device__ inline float lk_ordered_cov_sum(const float *__restrict__ cov, int R, const int *l, int L, int lane) {
  const long total = (long)L * L;
  float acc = 0;
  auto term = [&](long q) -> float {
    if (q >= total) return 0.0f;
    const int i = (int)(q / L), j = (int)(q - (long)i * L);
    return cov[(size_t)l[i] * R + l[j]];
  };
  float cur = term(lane);
  for (long base = 0; base < total; base += 64) {
    const float nxt = term(base + 64 + lane);  // in flight while the chain takes this tile
    acc = lk_ordered_sum64(acc, cur, (int)(total - base < 64 ? total - base : 64));
    cur = nxt;
  }
  return acc;
}
__global__ void __launch_bounds__(64)
lk_perm_scalars_kernel(const float *__restrict__ K, const float *__restrict__ cov, int R, const int *__restrict__ la,
                       int LA, const int *__restrict__ ln, int LN, int nperm, int lds, float *__restrict__ out) {
#pragma clang fp contract(off)
  extern __shared__ int lists[];  // LA + LN entries when the launch gives the room (lds != 0), else the lists stay in global memory
  const int perm = blockIdx.x, lane = threadIdx.x;
  const int *a = la + (size_t)perm * LA, *n = ln + (size_t)perm * LN;
  if (lds) {
    for (int i = lane; i < LA; i += 64) lists[i] = a[i];
    for (int i = lane; i < LN; i += 64) lists[LA + i] = n[i];
    __syncthreads();
    a = lists;
    n = lists + LA;
  }
  float sum1 = 0, sum0 = 0;
  for (int b = 0; b < LA; b += 64) sum1 = lk_ordered_sum64(sum1, b + lane < LA ? K[a[b + lane]] : 0.0f, min(64, LA - b));
  for (int b = 0; b < LN; b += 64) sum0 = lk_ordered_sum64(sum0, b + lane < LN ? K[n[b + lane]] : 0.0f, min(64, LN - b));
  float mn1 = sum1 / LA, mn0 = sum0 / LN;
  float v1 = lk_ordered_cov_sum(cov, R, a, LA, lane);
  float v0 = lk_ordered_cov_sum(cov, R, n, LN, lane);
  float varU1 = v1 / (LA * LA), varU0 = v0 / (LN * LN);
  float t = ((LN - 1) * varU0 + (LA - 1) * varU1) / (LN + LA - 2);
  double u = (double)t * (1.0 / (double)(float)LN + 1.0 / (double)(float)LA);
  if (lane == 0) {
    out[3 * perm] = mn1;
    out[3 * perm + 1] = mn0;
    out[3 * perm + 2] = (float)sqrt(u);
  }
}

// X[perm][s] = ((U1 - mn1) - (U0 - mn0)) / bottom (:1271-1284, 1315-1322), in place over U1
__global__ void __launch_bounds__(256)
lk_statistic_kernel(float *__restrict__ U1, const float *__restrict__ U0, const float *__restrict__ scal, int M) {
#pragma clang fp contract(off)
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int perm = blockIdx.y;
  if (s >= M) return;
  float mn1 = scal[3 * perm], mn0 = scal[3 * perm + 1], bottom = scal[3 * perm + 2];
  size_t at = (size_t)perm * M + s;
  float z1 = U1[at] + (-mn1), z0 = U0[at] + (-mn0);
  U1[at] = (z1 - z0) / bottom;
}

// Per permutation maximum (getMax_elem, :1169-1176) and the per-SNP exceedance counts psim (:1328-1332).
// X0 = the observed statistic.  first = 1 for the batch whose row 0 is X0 itself (it is not counted in psim).
__global__ void __launch_bounds__(256)
lk_perm_reduce_kernel(const float *__restrict__ X, const float *__restrict__ X0, int M, int nperm, int first,
                      float *__restrict__ xmax, int *__restrict__ psim) {
  __shared__ float red[256];
  int perm = blockIdx.x;
  const float *x = X + (size_t)perm * M;
  bool count = !(first && perm == 0);
  float mx = -INFINITY;
  for (int s = threadIdx.x; s < M; s += 256) {
    float v = x[s];
    if (v > mx) mx = v;
    if (count && v >= X0[s]) atomicAdd(&psim[s], 1);
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w && red[threadIdx.x + w] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) xmax[perm] = red[0];
}

}  // namespace relate
