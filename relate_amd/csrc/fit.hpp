// Host-side optimiser control for the batched fit: the numeric-gradient driver findmax_bfgs
// (src/bfgs.c:12-199), the six objective wrappers (src/relateHMM.cpp:1564-1849), the start points
// (get_rand / get_start, :278-293, 1546-1560, on a glibc-compatible rand() stream) and the multi-start rule of
// run_optimization (:1852-1971) -- all as resumable state machines, so that every active (pair, restart)
// contributes its requested points to ONE likelihood launch per L-BFGS-B iterate.
// Host code, compiled with -ffp-contract=off.  No numerics of the HMM live here.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "lbfgsb.hpp"

namespace relate {

static constexpr double kSysMax = 10000000.0;  // SYS_MAX, src/conf.h:6

// glibc random()/rand() TYPE_3 additive-feedback generator (r[i] = r[i-3] + r[i-31], seeded by the
// Lehmer LCG 16807 mod 2^31-1, first 310 outputs discarded); reproduces srand(seed); rand() of the
// reference's platform bit for bit (tests/test_fit_cpu.py pins it against the compiled reference).
class GlibcRand {
 public:
  explicit GlibcRand(unsigned seed = 1) { reseed(seed); }
  void reseed(unsigned seed) {
    if (seed == 0) seed = 1;
    r_[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
      int64_t hi = r_[i - 1] / 127773, lo = r_[i - 1] % 127773;
      int64_t w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      r_[i] = (int32_t)w;
    }
    f_ = 3;
    b_ = 0;
    for (int i = 0; i < 310; i++) next();
  }
  int next() {
    uint32_t v = (uint32_t)r_[f_] + (uint32_t)r_[b_];
    r_[f_] = (int32_t)v;
    f_ = (f_ + 1) % 31;
    b_ = (b_ + 1) % 31;
    return (int)(v >> 1);
  }
  double unit() { return (double)next() / (double)2147483647; }  // rand()/RAND_MAX

 private:
  int32_t r_[31];
  int f_, b_;
};

enum FitMode { MODE_FULL = 0, MODE_CALCA = 1, MODE_FIXK2 = 2, MODE_FIXK2_CALCA = 3, MODE_FIXK2_FIXA = 4 };

struct FitSpec {
  int mode;
  double fixA_val, fixK2_val, phi, alim0, alim1;
  int nvar() const { return mode == MODE_FULL ? 3 : (mode == MODE_CALCA || mode == MODE_FIXK2) ? 2 : 1; }
};

// branch table of run_optimization (:1868-1901); returns -1 for combinations the reference rejects or
// leaves undefined (fixA without fixK2 is broken in the reference)
inline int select_mode(int fixK2, int fixA, int calcA) {
  if (fixK2 && calcA && !fixA) return MODE_FIXK2_CALCA;
  if (fixK2 && fixA && !calcA) return MODE_FIXK2_FIXA;
  if (fixK2 && !fixA && !calcA) return MODE_FIXK2;
  if (!fixK2 && !fixA && !calcA) return MODE_FULL;
  if (!fixK2 && !fixA && calcA) return MODE_CALCA;
  return -1;
}

inline double calculate_a(double k1, double k2, double phi) {  // calculateA, :974-1007
  double ma, mb;
  if (k2 == 0) {
    ma = 1 - std::log(k1) / std::log(2);
    mb = 0;
  } else {
    double pw = std::pow(k1 + 2 * k2, 2);
    double sq = std::sqrt(pw - 4 * k2);
    double xa = (k1 + 2 * k2 + sq) / 2.0;
    double xb = k2 / xa;
    ma = 1 - std::log(xa) / std::log(2);
    mb = 1 - std::log(xb) / std::log(2);
  }
  double m = ma + mb;
  return -m * std::log(1 - phi);
}

// the *_fun wrappers: map the optimiser's variables to (a,k2,k1); false = the wrapper returns SYS_MAX
// without calling like()
inline bool map_point(const FitSpec &S, const double *x, double *akk) {
  if (S.mode == MODE_FULL) { akk[0] = x[0]; akk[1] = x[1]; akk[2] = x[2]; return true; }
  double k2, k1, a;
  if (S.mode == MODE_CALCA) { k2 = x[0]; k1 = x[1]; }
  else if (S.mode == MODE_FIXK2) { k2 = S.fixK2_val; k1 = x[1]; }
  else { k2 = S.fixK2_val; k1 = x[0]; }
  if (std::pow((k1 + 2 * k2), 2.0) < 4 * k2) return false;
  if (S.mode == MODE_CALCA || S.mode == MODE_FIXK2_CALCA) a = calculate_a(k1, k2, S.phi);
  else if (S.mode == MODE_FIXK2) a = x[0];
  else a = S.fixA_val;
  if (S.mode != MODE_FIXK2_FIXA && (std::isnan(a) || std::isinf(a))) return false;
  akk[0] = a; akk[1] = k2; akk[2] = k1;
  return true;
}

inline void mode_bounds(const FitSpec &S, double *lo, double *hi) {  // the *_optim drivers, :1597-1849
  switch (S.mode) {
    case MODE_FULL: lo[0] = S.alim0; hi[0] = S.alim1; lo[1] = 0; hi[1] = 1; lo[2] = 0; hi[2] = 1; break;
    case MODE_CALCA: lo[0] = 0; hi[0] = 1; lo[1] = 0; hi[1] = 1; break;
    case MODE_FIXK2: lo[0] = S.alim0; hi[0] = S.alim1; lo[1] = 0; hi[1] = 1; break;
    case MODE_FIXK2_CALCA:
      lo[0] = -2 * S.fixK2_val + 0.5 * std::sqrt(32 * S.fixK2_val - 16 * S.fixK2_val);
      hi[0] = 1;
      break;
    default: lo[0] = 0; hi[0] = 1; break;
  }
}

// start record {a,k2,k1,unused} -> optimiser variables
inline void start_to_x(const FitSpec &S, const double *st, double *x) {
  switch (S.mode) {
    case MODE_FULL: x[0] = st[0]; x[1] = st[1]; x[2] = st[2]; break;
    case MODE_CALCA: x[0] = st[1]; x[1] = st[2]; break;
    case MODE_FIXK2: x[0] = st[0]; x[1] = st[2]; break;
    default: x[0] = st[2]; break;
  }
}

// optimiser result -> {a,k2,k1,negloglik} as the *_optim drivers pack it
inline void x_to_result(const FitSpec &S, const double *x, double like, double *res) {
  switch (S.mode) {
    case MODE_FULL: res[0] = x[0]; res[1] = x[1]; res[2] = x[2]; break;
    case MODE_CALCA: res[0] = calculate_a(x[1], x[0], S.phi); res[1] = x[0]; res[2] = x[1]; break;
    case MODE_FIXK2: res[0] = x[0]; res[1] = S.fixK2_val; res[2] = x[1]; break;
    case MODE_FIXK2_CALCA: res[1] = S.fixK2_val; res[2] = x[0]; res[0] = calculate_a(res[2], res[1], S.phi); break;
    default: res[0] = S.fixA_val; res[1] = S.fixK2_val; res[2] = x[0]; break;
  }
  res[3] = like;
}

// one run's start record from the rand() stream: get_rand (4 draws, 3 when k2 is fixed), then for calcA
// get_start (2 draws per try until (k1+2k2)^2 >= 4k2)
inline void draw_start(GlibcRand &R, const FitSpec &S, int fixK2, double *st) {
  st[0] = S.alim0 + S.alim1 * R.unit();
  st[1] = fixK2 ? S.fixK2_val : R.unit();
  double sum = st[1];
  for (int i = 2; i < 4; i++) {
    int r = R.next();
    st[i] = (1 - sum) * ((double)r) / (double)2147483647;
    sum += st[i];
  }
  st[3] = -kSysMax;
  if (S.mode == MODE_CALCA) {
    for (;;) {
      double k2 = R.next() / (double)2147483647;
      double k1 = ((1 - k2) * R.next()) / (double)2147483647;
      if (!(std::pow((k1 + 2 * k2), 2.0) < 4 * k2)) { st[1] = k2; st[2] = k1; break; }
    }
  }
}

// findmax_bfgs with the central-difference gradient of Yanggradient/getgradient, resumable: request() lists
// the distinct points whose objective is needed; supply() feeds the values back.
class FindMax {
 public:
  int n = 0;
  bool done = false;
  double like = 0;
  Lbfgsb opt;
  int npts = 0;
  double pts[7][3];

  void begin(int n_, const double *x0, const double *lo, const double *hi, double *slab = nullptr) {
    n = n_;
    done = false;
    opt.init(n, x0, lo, hi, slab);
    // the reference evaluates fun/dfun once at the unprojected start before the first setulb_ call
    // (src/bfgs.c:165-166); those values are overwritten by the FG_START request and are not needed.
    advance();
  }

  void supply(const double *vals) {
    double f0 = vals[0];
    opt.f = f0;  // like = (*fun)(invec); getgradient's own f0 = func(invec) is the same point
    for (int i = 0; i < n; i++) {
      double gi;
      if (kind_[i] == 1) gi = (vals[ia_[i]] - f0) / (eh_[i] * 2.0);
      else if (kind_[i] == 2) gi = (f0 - vals[ia_[i]]) / (eh_[i] * 2.0);
      else gi = (vals[ia_[i]] - vals[ib_[i]]) / (eh_[i] * 2.0);
      // checkbounds, src/bfgs.c:92-98
      if (opt.x[i] <= opt.lo[i] && gi > 0.0) gi = 0.0;
      if (opt.x[i] >= opt.hi[i] && gi < 0.0) gi = 0.0;
      opt.g[i] = gi;
    }
    advance();
  }

 private:
  int kind_[3], ia_[3], ib_[3];
  double eh_[3];

  void advance() {
    for (;;) {
      Lbfgsb::Task t = opt.step();
      if (t == Lbfgsb::FG_START || t == Lbfgsb::FG_LNSRCH) { build_request(); return; }
      if (t == Lbfgsb::NEW_X) continue;
      done = true;
      like = opt.f;
      return;
    }
  }

  void build_request() {  // Yanggradient, central mode (src/bfgs.c:24-37)
    npts = 0;
    for (int j = 0; j < n; j++) pts[0][j] = opt.x[j];
    npts = 1;
    for (int i = 0; i < n; i++) {
      double eh = std::pow(1.e-8 * (std::fabs(opt.x[i]) + 1), 0.67);
      eh_[i] = eh;
      double x0 = opt.x[i], x1 = opt.x[i];
      x0 -= eh;
      x1 += eh;
      if (x0 < opt.lo[i]) {
        x1 += eh;
        kind_[i] = 1;
        ia_[i] = add_point(i, x1);
      } else if (x1 > opt.hi[i]) {
        x0 -= eh;
        kind_[i] = 2;
        ia_[i] = add_point(i, x0);
      } else {
        kind_[i] = 0;
        ia_[i] = add_point(i, x1);
        ib_[i] = add_point(i, x0);
      }
    }
  }
  int add_point(int i, double v) {
    for (int j = 0; j < n; j++) pts[npts][j] = opt.x[j];
    pts[npts][i] = v;
    return npts++;
  }
};

// multi-start rule of run_optimization for one pair; runs are launched in waves (the minimum number that could
// still be needed, or all remaining ones once the first wave has shown no agreement) and consumed strictly in index
// order, so the outcome equals the sequential loop.
struct MultiStart {
  int timesToConverge = 1, timesToRun = 1;
  double convTol = 0.1;
  int launched = 0, consumed = 0;  // runs started / runs fed through the rule
  int conv = 0;
  bool finished = false;
  double best[4] = {0, 0, 0, 0};

  int first_wave() const { return timesToConverge < timesToRun ? (timesToConverge > 0 ? timesToConverge : 1) : timesToRun; }
  // how many more runs to start now.  eager: every remaining run at once, without waiting for the wave in flight (the
  // caller has seen that first waves rarely settle a pair; runs beyond the stopping point are dropped unseen as ever)
  int want_to_launch(bool eager = false) const {
    if (finished) return 0;
    if (eager) return timesToRun - launched;
    if (launched > consumed) return 0;  // a wave is still in flight
    if (launched == 0) return first_wave();
    int need = timesToConverge - conv;
    if (need < 1) need = 1;
    int room = timesToRun - launched;
    // No two runs have agreed so far: expect to need every remaining run and start them together, so that the
    // likelihood launches keep enough lanes to fill the machine.  Results are still consumed strictly in index
    // order and the rule stops where the sequential loop stops; runs beyond that point are discarded unseen.
    if (conv <= 1 && consumed >= 2) return room;
    return need < room ? need : room;
  }
  void consume(const double *res) {  // res = {a,k2,k1,negloglik} of run `consumed`
    if (consumed == 0) {
      for (int i = 0; i < 4; i++) best[i] = res[i];
      conv = 1;
    } else {
      double diff = std::fabs(res[3] - best[3]);
      if (diff < convTol) {
        conv++;
      } else if (std::fabs(res[3]) < std::fabs(best[3])) {
        for (int i = 0; i < 4; i++) best[i] = res[i];
        conv = 1;
      }
    }
    consumed++;
    if (!(conv < timesToConverge && consumed < timesToRun)) finished = true;
  }
};

}  // namespace relate
