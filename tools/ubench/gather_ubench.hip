// Feasibility of code-indexed emission records for the decode (gfx950): per (pair, SNP) a 16-bit code instead of a
// materialised 16-byte record {S1/S0, S2/S0}; the sweep gathers the 16 bytes from the per-SNP table (L2 / Infinity
// Cache resident: 820 entries x 16 B per SNP at back2 = 10, the pairs of a wave hit the ~81 entries of one depth).
// Emulates the backward pass of the decode: one wave per workgroup = 64 pairs of one SNP chunk, tiles of 16 SNPs, the
// next tile's records in flight while the current one is worked on (~150 dependent-ish FP64 instructions per SNP),
// 16 B of posteriors out per (pair, SNP).  Variant 0: contiguous 16-byte records as the library has them
// ([s/2][group][plane][lane][s%2]); variant 1: codes ([s/8][pair] x 8 x u16) + table gathers.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_ubench gather_ubench.hip ; run on the GPU box:
//   ./gather_ubench [pairs 5120] [snps 250000] [chunks 46] [work 150] [skew 0|1]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kTile = 16, kEntries = 820;

__device__ __forceinline__ void work(double r1, double r2, double &b0, double &b1, double &b2, int nwork) {
  // nwork FP64 FMAs in three chains that depend on the record
  for (int i = 0; i < nwork; i += 3) {
    b0 = __builtin_fma(b0, 0.999999, r1 * 1e-9);
    b1 = __builtin_fma(b1, 0.999998, r2 * 1e-9);
    b2 = __builtin_fma(b2, 0.999997, b0 * 1e-12);
  }
}

template <int VARIANT>
__global__ void __launch_bounds__(64) bwd(const double *__restrict__ rec, const uint16_t *__restrict__ codes,
                                          const double2 *__restrict__ tab, int M, int P, int RP, int chunk_len, int nwork,
                                          double *__restrict__ post) {
  int bpc = (P + 63) / 64;
  int c = blockIdx.x / bpc;
  int p = (blockIdx.x % bpc) * 64 + threadIdx.x;
  int sb = c * chunk_len, se = min(M, sb + chunk_len);
  double b0 = 1, b1 = 1, b2 = 1;
  double2 R[kTile], Rn[kTile];
  uint4 C0, C1, Cn0 = {0, 0, 0, 0}, Cn1 = {0, 0, 0, 0};
  auto load_codes = [&](int tb, uint4 &a, uint4 &b) {
    const uint4 *q = (const uint4 *)(codes + ((size_t)(tb >> 3) * RP + p) * 8);
    a = q[0];
    b = q[(size_t)RP];
  };
  auto gather = [&](int tb, const uint4 &a, const uint4 &b, double2 (&X)[kTile]) {
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int u = 0; u < kTile; u++) {
      unsigned code = (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
      X[u] = tab[(size_t)(tb + u) * kEntries + code];  // scalar row base + lane's entry
    }
  };
  auto load_recs = [&](int tb, double2 (&X)[kTile]) {
#pragma unroll
    for (int u = 0; u < kTile; u += 2) {
      const double *q = rec + (size_t)((tb + u) >> 1) * 4 * RP + (size_t)(p >> 6) * 256 + (p & 63) * 2;
      double2 x = *(const double2 *)q, y = *(const double2 *)(q + 128);
      X[u] = double2{x.x, y.x};
      X[u + 1] = double2{x.y, y.y};
    }
  };
  int tb0 = (se - 1) & ~(kTile - 1);
  if (VARIANT == 1) {
    load_codes(tb0, C0, C1);
    gather(tb0, C0, C1, R);
    load_codes(max(tb0 - kTile, sb), C0, C1);
  } else {
    load_recs(tb0, R);
  }
  for (int tb = tb0; tb >= sb; tb -= kTile) {
    int tn = max(tb - kTile, sb), tnn = max(tb - 2 * kTile, sb);
    if (VARIANT == 1) {
      gather(tn, C0, C1, Rn);        // records of the next tile: their codes arrived a tile ago
      load_codes(tnn, Cn0, Cn1);     // codes of the tile after it
    } else {
      load_recs(tn, Rn);
    }
    double o0[kTile], o1[kTile];
#pragma unroll
    for (int u = kTile - 1; u >= 0; u--) {
      work(R[u].x, R[u].y, b0, b1, b2, nwork);
      o0[u] = b0 + b2;
      o1[u] = b1;
    }
    // 16 B per (pair, SNP) out, [row][s][pair] (coalesced; the real kernel transposes through LDS)
#pragma unroll
    for (int u = 0; u < kTile; u++) {
      if (tb + u < se) {
        post[(size_t)(tb + u) * P + p] = o0[u];
        post[(size_t)M * P + (size_t)(tb + u) * P + p] = o1[u];
      }
    }
#pragma unroll
    for (int u = 0; u < kTile; u++) R[u] = Rn[u];
    C0 = Cn0;
    C1 = Cn1;
  }
}


// Variant 2: codes + the SNP's dominant-depth slice of the table (81 entries x 16 B) staged through LDS by a workgroup of
// NW waves (64 x NW pairs of one SNP chunk share a tile of 16 slices = 20.7 KB); lanes whose code sits on another depth
// (a missing genotype in the pair's window) gather that record from the table in global memory.
template <int NW>
__global__ void __launch_bounds__(64 * NW) bwd_lds(const uint16_t *__restrict__ codes, const double2 *__restrict__ tab, int M, int P,
                                                   int RP, int chunk_len, int nwork, double *__restrict__ post) {
  constexpr int NT = 64 * NW, kSlice = 81, kPer = (kTile * kSlice + NT - 1) / NT;
  __shared__ double2 lt[kTile * kSlice];
  int bpc = (P + NT - 1) / NT;
  int c = blockIdx.x / bpc;
  int p = (blockIdx.x % bpc) * NT + threadIdx.x;
  const bool live = p < P;
  if (!live) p = P - 1;
  int sb = c * chunk_len, se = min(M, sb + chunk_len);
  double b0 = 1, b1 = 1, b2 = 1;
  double2 R[kTile], T[kPer];
  uint4 C0, C1;
  auto load_codes = [&](int tb, uint4 &a, uint4 &b) {
    const uint4 *q = (const uint4 *)(codes + ((size_t)(tb >> 3) * RP + p) * 8);
    a = q[0];
    b = q[(size_t)RP];
  };
  auto load_table = [&](int tb) {  // this thread's share of the 16 slices of tile tb
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      int e = threadIdx.x + j * NT;
      int u = e / kSlice, k = e - u * kSlice;
      if (e < kTile * kSlice) T[j] = tab[(size_t)(tb + u) * kEntries + 10 + ((tb + u) % 5) * kSlice + k];
    }
  };
  int tb0 = (se - 1) & ~(kTile - 1);
  load_codes(tb0, C0, C1);
  load_table(tb0);
  for (int tb = tb0; tb >= sb; tb -= kTile) {
    __syncthreads();  // everyone is done with the previous tile's slices
#pragma unroll
    for (int j = 0; j < kPer; j++) {
      int e = threadIdx.x + j * NT;
      if (e < kTile * kSlice) lt[e] = T[j];
    }
    __syncthreads();
    const unsigned w[8] = {C0.x, C0.y, C0.z, C0.w, C1.x, C1.y, C1.z, C1.w};
#pragma unroll
    for (int u = 0; u < kTile; u++) {
      unsigned code = (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
      int k = (int)code - 10 - ((tb + u) % 5) * kSlice;
      if (k >= 0 && k < kSlice) R[u] = lt[u * kSlice + k];
      else R[u] = tab[(size_t)(tb + u) * kEntries + code];
    }
    int tn = max(tb - kTile, sb);
    load_codes(tn, C0, C1);
    load_table(tn);
    double o0[kTile], o1[kTile];
#pragma unroll
    for (int u = kTile - 1; u >= 0; u--) {
      work(R[u].x, R[u].y, b0, b1, b2, nwork);
      o0[u] = b0 + b2;
      o1[u] = b1;
    }
    if (live) {
#pragma unroll
      for (int u = 0; u < kTile; u++) {
        if (tb + u < se) {
          post[(size_t)(tb + u) * P + p] = o0[u];
          post[(size_t)M * P + (size_t)(tb + u) * P + p] = o1[u];
        }
      }
    }
  }
}

int main(int argc, char **argv) {
  int P = argc > 1 ? atoi(argv[1]) : 5120, M = argc > 2 ? atoi(argv[2]) : 250000, nchunk = argc > 3 ? atoi(argv[3]) : 46;
  int nwork = argc > 4 ? atoi(argv[4]) : 150, skew = argc > 5 ? atoi(argv[5]) : 1;
  M = M / 16 * 16;
  int RP = (P + 63) / 64 * 64;
  int clen = ((M + nchunk - 1) / nchunk + 15) / 16 * 16;
  nchunk = (M + clen - 1) / clen;
  double *rec, *post;
  uint16_t *codes;
  double2 *tab;
  size_t nrec = (size_t)(M / 2) * 4 * RP, ncode = (size_t)(M / 8 + 1) * RP * 8, ntab = (size_t)M * kEntries;
  CHK(hipMalloc(&rec, nrec * 8));
  CHK(hipMalloc(&codes, ncode * 2));
  CHK(hipMalloc(&tab, ntab * 16));
  CHK(hipMalloc(&post, (size_t)2 * M * P * 8));
  CHK(hipMemset(rec, 0, nrec * 8));
  CHK(hipMemset(tab, 0, ntab * 16));
  {
    // codes: depth dd(s) = s % 5 (the global choice), combination skewed (a few genotype combinations carry most pairs)
    // or uniform over the 81; one code in 50 on another depth (pairs with a missing genotype in the window)
    std::vector<uint16_t> h(ncode);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < ncode; i++) {
      size_t s = (i / 8 / RP) * 8 + i % 8;
      unsigned r = (unsigned)(rnd() >> 33);
      unsigned combo = skew ? ((r % 81) * ((r >> 8) % 81) / 81) : r % 81;
      unsigned dd = ((r >> 20) % 50 == 0) ? (r >> 26) % 10 : s % 5;
      h[i] = (uint16_t)(10 + dd * 81 + combo);
    }
    CHK(hipMemcpy(codes, h.data(), ncode * 2, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  dim3 grid((unsigned)((size_t)((P + 63) / 64) * nchunk));
  printf("P %d, M %d, chunks %d (len %d), work %d FMA per SNP, %s combinations; records %.1f GB, codes %.1f GB, table %.1f GB\n", P, M,
         nchunk, clen, nwork, skew ? "skewed" : "uniform", nrec * 8 / 1e9, ncode * 2 / 1e9, ntab * 16 / 1e9);
  for (int v = 0; v < 4; v++) {
    for (int rep = 0; rep < 3; rep++) {
      CHK(hipEventRecord(e0));
      if (v == 0) hipLaunchKernelGGL(bwd<0>, grid, dim3(64), 0, 0, rec, codes, tab, M, P, RP, clen, nwork, post);
      else if (v == 1) hipLaunchKernelGGL(bwd<1>, grid, dim3(64), 0, 0, rec, codes, tab, M, P, RP, clen, nwork, post);
      else if (v == 2) hipLaunchKernelGGL(bwd_lds<4>, dim3((unsigned)((size_t)((P + 255) / 256) * nchunk)), dim3(256), 0, 0, codes, tab, M, P, RP, clen, nwork, post);
      else hipLaunchKernelGGL(bwd_lds<2>, dim3((unsigned)((size_t)((P + 127) / 128) * nchunk)), dim3(128), 0, 0, codes, tab, M, P, RP, clen, nwork, post);
      CHK(hipEventRecord(e1));
      CHK(hipDeviceSynchronize());
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      double units = (double)P * M;
      if (rep)
        printf("variant %d (%s): %8.3f ms  %.1f G pair*SNP/s  (x 2 for M = 5e5: %.1f ms)\n", v, v == 0 ? "16-byte records" : v == 1 ? "codes + table gather" : v == 2 ? "codes + LDS slices, 256 pairs" : "codes + LDS slices, 128 pairs",
               ms, units / ms / 1e6, ms * 5e5 / M);
    }
  }
  return 0;
}
