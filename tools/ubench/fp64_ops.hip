// FP64 VALU micro-benchmark for gfx950: per-op issue cost (independent chains) and dependent latency.
// Build: hipcc --offload-arch=gfx950 -O3 -o fp64_ops fp64_ops.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2048;

template <int OP, int CH>
__global__ void __launch_bounds__(256) k(double *out, double seed, int iters) {
  double x[CH];
  for (int c = 0; c < CH; c++) x[c] = seed + threadIdx.x * 1e-9 + c * 1e-7;
  double m = 1.0000001, a = 1e-9;
  int ei = (int)seed;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      if (OP == 0) x[c] = __builtin_fma(x[c], m, a);
      else if (OP == 1) x[c] = x[c] * m;
      else if (OP == 2) x[c] = x[c] + a;
      else if (OP == 3) x[c] = __builtin_amdgcn_ldexp(x[c], ei);
      else if (OP == 4) x[c] = __builtin_rint(x[c] + a);
      else if (OP == 5) x[c] = __builtin_amdgcn_rcp(x[c]);
      else if (OP == 6) x[c] = __builtin_fmax(x[c], a);
      else if (OP == 7) x[c] = (x[c] < a) ? m : x[c];
      else if (OP == 8) x[c] = (double)(int)x[c];
      else if (OP == 9) x[c] = (double)__builtin_amdgcn_frexp_exp(x[c]);
      else if (OP == 10) x[c] = exp(-x[c] * 1e-3);
      else if (OP == 11) x[c] = 1.0 / x[c];
      else if (OP == 12) x[c] = log(x[c]);
      else if (OP == 13) x[c] = __builtin_amdgcn_sqrt(x[c]);
      else if (OP == 14) x[c] = expm1(-x[c] * 1e-3);
    }
  }
  double s = 0;
  for (int c = 0; c < CH; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CH>
void run(const char *name, double *d, int nops_per = 1) {
  int blocks = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d, 1.5, 16);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d, 1.5, ITER);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 256 * ITER * CH;
  // cycles per wave-instruction per SIMD at 2.4 GHz: waves per SIMD = 8; time*2.4e9 / (ITER*CH*8)
  double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * CH * 8);
  printf("%-28s CH=%d  %8.3f ms  %8.2f Gop/s  ~%6.2f cycles/wave-op/SIMD (8 waves/SIMD, 2.4GHz assumed)\n", name, CH, ms, ops / ms / 1e6, cyc);
  // single wave per SIMD, dependent chain: latency
}

template <int OP>
void lat(const char *name, double *d) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<OP, 1>), dim3(1), dim3(64), 0, 0, d, 1.5, 16);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<OP, 1>), dim3(1), dim3(64), 0, 0, d, 1.5, ITER * 16);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s dependent chain, 1 wave: %7.2f ns/op\n", name, ms * 1e6 / (ITER * 16.0));
}

int main() {
  double *d; CHK(hipMalloc(&d, sizeof(double) * 256 * 8 * 256));
  run<0, 8>("v_fma_f64", d); run<1, 8>("v_mul_f64", d); run<2, 8>("v_add_f64", d);
  run<3, 8>("v_ldexp_f64", d); run<4, 8>("v_add+v_rndne_f64", d); run<5, 8>("v_rcp_f64", d);
  run<6, 8>("v_max_f64", d); run<7, 8>("v_cmp+cndmask x2", d); run<8, 8>("cvt_i32_f64+cvt_f64_i32", d);
  run<9, 8>("frexp_exp+cvt", d); run<10, 4>("ocml exp(-x*1e-3)", d); run<11, 4>("1.0/x (IEEE div)", d);
  run<12, 4>("ocml log", d); run<13, 8>("v_sqrt_f64", d); run<14, 4>("ocml expm1", d);
  run<0, 1>("v_fma_f64 (1 chain)", d); run<0, 2>("v_fma_f64 (2 chains)", d); run<0, 4>("v_fma_f64 (4 chains)", d);
  lat<0>("v_fma_f64", d); lat<1>("v_mul_f64", d); lat<3>("v_ldexp_f64", d); lat<5>("v_rcp_f64", d);
  return 0;
}
