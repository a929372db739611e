// Re-entrant L-BFGS-B (Byrd, Lu, Nocedal, Zhu 1995; code version 2.1, "latest revision June 1996") for the
// batched optimiser: one small state object per (pair, restart), advanced by reverse communication so that
// thousands of them can share one likelihood launch per iterate.
//
// The reference drives the f2c translation of the same algorithm through setulb_ (src/bfgs.c:240-502) with
// m = 30, factr = 1e6, pgtol = 1e-4, all variables two-sided bounded (src/bfgs.h:44-46, src/bfgs.c:152-155).
// Estimates are only reproducible if the iterates are, so this restatement keeps the published algorithm's
// arithmetic order (generalised Cauchy point, subspace minimisation with backtracking to the box, More-Thuente
// line search with ftol 1e-3 / gtol 0.9 / xtol 0.1 and at most 20 trials) -- see tests/test_lbfgsb.py, which
// checks iterates bit-for-bit against the compiled reference.  Indices below are 1-based as in the papers.
// Host code, compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace relate {

class Lbfgsb {
 public:
  enum Task { FG_START, FG_LNSRCH, NEW_X, CONVERGED_PGTOL, CONVERGED_FACTR, ABNORMAL_LNSRCH, ERROR_INPUT };

  static constexpr int kMaxN = 3;
  static constexpr int kM = 30;  // MVAL, src/bfgs.h:44

  int n = 0;
  double x[kMaxN], lo[kMaxN], hi[kMaxN], g[kMaxN];
  int nbd[kMaxN];
  double f = 0;
  double factr = 1.0e6, pgtol = 1.0e-4;  // FACTR, PGTOL (src/bfgs.h:45-46)
  int iter = 0, nfgv = 0;

  // Work matrices: ws, wy (n x m), sy, ss, wt (m x m), wn, snd (2m x 2m), one after the other in a slab of kSlab doubles.
  // A caller that keeps thousands of states passes the slab (any contents; recycled from a finished run or fresh from
  // an arena of its own) -- the matrices are 80 KB per state and, allocated one by one, cost a page fault per 4 KB;
  // without one the state owns its storage.
  static constexpr size_t kSlab = (size_t)2 * kMaxN * kM + 3 * kM * kM + 8 * kM * kM;
  Lbfgsb() = default;
  Lbfgsb(const Lbfgsb &) = delete;
  Lbfgsb &operator=(const Lbfgsb &) = delete;

  void init(int n_, const double *x0, const double *l, const double *u, double *slab = nullptr) {
    n = n_;
    for (int i = 0; i < n; i++) { x[i] = x0[i]; lo[i] = l[i]; hi[i] = u[i]; nbd[i] = 2; g[i] = 0; }
    started_ = false;
    if (!slab) {
      own_.resize(kSlab);
      slab = own_.data();
    }
    ws_ = slab; wy_ = ws_ + kMaxN * kM; sy_ = wy_ + kMaxN * kM; ss_ = sy_ + kM * kM; wt_ = ss_ + kM * kM;
    wn_ = wt_ + kM * kM; snd_ = wn_ + 4 * kM * kM;
    std::memset(slab, 0, sizeof(double) * (size_t)(wn_ - slab));
    wn_ready_ = false;  // 2m x 2m each: cleared on first use
    std::memset(wa_, 0, sizeof(wa_));
  }

  // One call of setulb_: on the first call returns FG_START; afterwards the caller has put f(x) and g(x) into
  // f / g (for FG_* tasks) and calls again.  NEW_X must simply be followed by another call.
  Task step() {
    if (!started_) { start(); return task_; }
    resume();
    return task_;
  }

 private:
  // ---- persistent state (the f2c statics that mainlb saves in lsave/isave/dsave, src/bfgs.c:1272-1309) ----
  bool started_ = false;
  Task task_ = FG_START;
  bool prjctd_, cnstnd_, boxed_, updatd_;
  int iback_, nskip_, head_, col_, itail_, iupdat_, nint_, info_, ifun_, iword_, nfree_, nact_, ileave_, nenter_;
  double theta_, fold_, tol_, dnorm_, epsmch_, gd_, stpmx_, sbgnrm_, stp_, gdold_, dtd_;
  bool wrk_ = false;
  // dcsrch state
  bool brackt_;
  int stage_;
  double ginit_, gtest_, gx_, gy_, finit_, fx_, fy_, stx_, sty_, stmin_, stmax_, width_, width1_;
  enum LsTask { LS_START, LS_FG, LS_CONV, LS_WARN, LS_ERROR } lstask_ = LS_START;

  std::vector<double> own_;
  double *ws_ = nullptr, *wy_ = nullptr, *sy_ = nullptr, *ss_ = nullptr, *wt_ = nullptr, *wn_ = nullptr, *snd_ = nullptr;
  bool wn_ready_ = false;
  double z_[kMaxN + 1], r_[kMaxN + 1], d_[kMaxN + 1], t_[kMaxN + 1], xs_[kMaxN + 1];
  double wa_[8 * kM + 1];
  int index_[kMaxN + 1], iwhere_[kMaxN + 1], indx2_[kMaxN + 1];

  // 1-based accessors (column-major, leading dimensions as in the original work arrays)
  double &WS(int i, int j) { return ws_[(size_t)(i - 1) + (size_t)(j - 1) * n]; }
  double &WY(int i, int j) { return wy_[(size_t)(i - 1) + (size_t)(j - 1) * n]; }
  double &SY(int i, int j) { return sy_[(size_t)(i - 1) + (size_t)(j - 1) * kM]; }
  double &SS(int i, int j) { return ss_[(size_t)(i - 1) + (size_t)(j - 1) * kM]; }
  double &WT(int i, int j) { return wt_[(size_t)(i - 1) + (size_t)(j - 1) * kM]; }
  double &WN(int i, int j) { return wn_[(size_t)(i - 1) + (size_t)(j - 1) * 2 * kM]; }
  double &WN1(int i, int j) { return snd_[(size_t)(i - 1) + (size_t)(j - 1) * 2 * kM]; }
  double &X(int i) { return x[i - 1]; }
  double &G(int i) { return g[i - 1]; }
  double L(int i) const { return lo[i - 1]; }
  double U(int i) const { return hi[i - 1]; }
  int NBD(int i) const { return nbd[i - 1]; }

  static double ddot(int cnt, const double *a, int inca, const double *b, int incb) {
    double s = 0;  // sequential accumulation, as the reference's ddot_ does in effect (src/bfgs.c, ddot_)
    for (int i = 0; i < cnt; i++) s += a[(size_t)i * inca] * b[(size_t)i * incb];
    return s;
  }

  void reset_memory() {  // "refresh the lbfgs memory and restart the iteration"
    info_ = 0; col_ = 0; head_ = 1; theta_ = 1.; iupdat_ = 0; updatd_ = false;
  }

  // ---- LINPACK pieces on column-major storage with leading dimension ld, 1-based ----
  // dpofa: Cholesky factorisation A = R'R of the leading cnt x cnt block, upper triangle in place
  static int dpofa(double *a, int ld, int cnt) {
    auto A = [&](int i, int j) -> double & { return a[(size_t)(i - 1) + (size_t)(j - 1) * ld]; };
    for (int j = 1; j <= cnt; j++) {
      double s = 0.;
      for (int k = 1; k <= j - 1; k++) {
        double t = A(k, j) - ddot(k - 1, &A(1, k), 1, &A(1, j), 1);
        t /= A(k, k);
        A(k, j) = t;
        s += t * t;
      }
      s = A(j, j) - s;
      if (s <= 0.) return j;
      A(j, j) = std::sqrt(s);
    }
    return 0;
  }
  // dtrsl: job 01 solves T x = b, job 11 solves T' x = b, T upper triangular
  static int dtrsl(double *t, int ld, int cnt, double *b, int job) {
    auto T = [&](int i, int j) -> double & { return t[(size_t)(i - 1) + (size_t)(j - 1) * ld]; };
    auto B = [&](int i) -> double & { return b[i - 1]; };
    for (int i = 1; i <= cnt; i++)
      if (T(i, i) == 0.) return i;
    if (job == 1) {
      B(cnt) /= T(cnt, cnt);
      for (int jj = 2; jj <= cnt; jj++) {
        int j = cnt - jj + 1;
        double temp = -B(j + 1);
        if (temp != 0.)
          for (int i = 1; i <= j; i++) B(i) += temp * T(i, j + 1);
        B(j) /= T(j, j);
      }
    } else {  // job == 11
      B(1) /= T(1, 1);
      for (int j = 2; j <= cnt; j++) {
        B(j) -= ddot(j - 1, &T(1, j), 1, &B(1), 1);
        B(j) /= T(j, j);
      }
    }
    return 0;
  }

  // ---- algorithm pieces ----
  void active() {
    prjctd_ = false; cnstnd_ = false; boxed_ = true;
    for (int i = 1; i <= n; i++) {
      if (NBD(i) > 0) {
        if (NBD(i) <= 2 && X(i) <= L(i)) {
          if (X(i) < L(i)) { prjctd_ = true; X(i) = L(i); }
        } else if (NBD(i) >= 2 && X(i) >= U(i)) {
          if (X(i) > U(i)) { prjctd_ = true; X(i) = U(i); }
        }
      }
    }
    for (int i = 1; i <= n; i++) {
      if (NBD(i) != 2) boxed_ = false;
      if (NBD(i) == 0) iwhere_[i] = -1;
      else {
        cnstnd_ = true;
        iwhere_[i] = (NBD(i) == 2 && U(i) - L(i) <= 0.) ? 3 : 0;
      }
    }
  }

  void projgr() {
    sbgnrm_ = 0.;
    for (int i = 1; i <= n; i++) {
      double gi = G(i);
      if (NBD(i) != 0) {
        if (gi < 0.) {
          if (NBD(i) >= 2) gi = std::fmax(X(i) - U(i), gi);
        } else {
          if (NBD(i) <= 2) gi = std::fmin(X(i) - L(i), gi);
        }
      }
      sbgnrm_ = std::fmax(sbgnrm_, std::fabs(gi));
    }
  }

  // product of the 2m x 2m middle matrix of the compact L-BFGS formula with a 2col vector v, result in p
  int bmv(const double *v, double *p) {
    auto V = [&](int i) { return v[i - 1]; };
    auto P = [&](int i) -> double & { return p[i - 1]; };
    int col = col_;
    if (col == 0) return 0;
    P(col + 1) = V(col + 1);
    for (int i = 2; i <= col; i++) {
      int i2 = col + i;
      double sum = 0.;
      for (int k = 1; k <= i - 1; k++) sum += SY(i, k) * V(k) / SY(k, k);
      P(i2) = V(i2) + sum;
    }
    if (dtrsl(wt_, kM, col, &P(col + 1), 11)) return 1;
    for (int i = 1; i <= col; i++) P(i) = V(i) / std::sqrt(SY(i, i));
    if (dtrsl(wt_, kM, col, &P(col + 1), 1)) return 1;
    for (int i = 1; i <= col; i++) P(i) = -P(i) / std::sqrt(SY(i, i));
    for (int i = 1; i <= col; i++) {
      double sum = 0.;
      for (int k = i + 1; k <= col; k++) sum += SY(k, i) * P(col + k) / SY(i, i);
      P(i) += sum;
    }
    return 0;
  }

  // heap step used by the Cauchy search to extract breakpoints in increasing order
  static void hpsolb(int cnt, double *t, int *iorder, int iheap) {
    auto T = [&](int i) -> double & { return t[i]; };       // t, iorder are 1-based arrays here
    auto O = [&](int i) -> int & { return iorder[i]; };
    if (iheap == 0) {
      for (int k = 2; k <= cnt; k++) {
        double ddum = T(k);
        int indxin = O(k), i = k;
        while (i > 1) {
          int j = i / 2;
          if (ddum < T(j)) { T(i) = T(j); O(i) = O(j); i = j; } else break;
        }
        T(i) = ddum; O(i) = indxin;
      }
    }
    if (cnt > 1) {
      int i = 1;
      double out = T(1);
      int indxou = O(1);
      double ddum = T(cnt);
      int indxin = O(cnt);
      for (;;) {
        int j = i + i;
        if (j <= cnt - 1) {
          if (T(j + 1) < T(j)) j++;
          if (T(j) < ddum) { T(i) = T(j); O(i) = O(j); i = j; } else break;
        } else break;
      }
      T(i) = ddum; O(i) = indxin;
      T(cnt) = out; O(cnt) = indxou;
    }
  }

  // generalised Cauchy point: z_ = xcp; p = wa[1..2m], c = wa[2m+1..4m], wbp = wa[4m+1..6m], v = wa[6m+1..8m]
  int cauchy() {
    double *p = wa_, *c = wa_ + 2 * kM, *wbp = wa_ + 4 * kM, *v = wa_ + 6 * kM;  // 1-based via [i]
    int *iorder = indx2_;
    double *t = t_, *d = d_, *xcp = z_;
    int col = col_, m = kM;
    double theta = theta_;
    if (sbgnrm_ <= 0.) {
      for (int i = 1; i <= n; i++) xcp[i] = X(i);
      return 0;
    }
    bool bnded = true;
    int nfree = n + 1, nbreak = 0, ibkmin = 0, col2 = 2 * col;
    double bkmin = 0., f1 = 0., tl = 0, tu = 0;
    for (int i = 1; i <= col2; i++) p[i] = 0.;
    for (int i = 1; i <= n; i++) {
      double neggi = -G(i);
      if (iwhere_[i] != 3 && iwhere_[i] != -1) {
        if (NBD(i) <= 2) tl = X(i) - L(i);
        if (NBD(i) >= 2) tu = U(i) - X(i);
        bool xlower = NBD(i) <= 2 && tl <= 0.;
        bool xupper = NBD(i) >= 2 && tu <= 0.;
        iwhere_[i] = 0;
        if (xlower) {
          if (neggi <= 0.) iwhere_[i] = 1;
        } else if (xupper) {
          if (neggi >= 0.) iwhere_[i] = 2;
        } else {
          if (std::fabs(neggi) <= 0.) iwhere_[i] = -3;
        }
      }
      int pointr = head_;
      if (iwhere_[i] != 0 && iwhere_[i] != -1) {
        d[i] = 0.;
      } else {
        d[i] = neggi;
        f1 -= neggi * neggi;
        for (int j = 1; j <= col; j++) {
          p[j] += WY(i, pointr) * neggi;
          p[col + j] += WS(i, pointr) * neggi;
          pointr = pointr % m + 1;
        }
        if (NBD(i) <= 2 && NBD(i) != 0 && neggi < 0.) {
          ++nbreak;
          iorder[nbreak] = i;
          t[nbreak] = tl / (-neggi);
          if (nbreak == 1 || t[nbreak] < bkmin) { bkmin = t[nbreak]; ibkmin = nbreak; }
        } else if (NBD(i) >= 2 && neggi > 0.) {
          ++nbreak;
          iorder[nbreak] = i;
          t[nbreak] = tu / neggi;
          if (nbreak == 1 || t[nbreak] < bkmin) { bkmin = t[nbreak]; ibkmin = nbreak; }
        } else {
          --nfree;
          iorder[nfree] = i;
          if (std::fabs(neggi) > 0.) bnded = false;
        }
      }
    }
    if (theta != 1.)
      for (int j = 1; j <= col; j++) p[col + j] = theta * p[col + j];
    for (int i = 1; i <= n; i++) xcp[i] = X(i);
    if (nbreak == 0 && nfree == n + 1) return 0;
    for (int j = 1; j <= col2; j++) c[j] = 0.;
    double f2 = -theta * f1;
    if (col > 0) {
      if (bmv(p + 1, v + 1)) return 1;
      f2 -= ddot(col2, v + 1, 1, p + 1, 1);
    }
    double dtm = -f1 / f2, tsum = 0.;
    nint_ = 1;
    bool to_999 = false;
    if (nbreak != 0) {
      int nleft = nbreak, it = 1, ibp = 0;
      double tj = 0.;
      for (;;) {
        double tj0 = tj;
        if (it == 1) {
          tj = bkmin;
          ibp = iorder[ibkmin];
        } else {
          if (it == 2) {
            if (ibkmin != nbreak) { t[ibkmin] = t[nbreak]; iorder[ibkmin] = iorder[nbreak]; }
          }
          hpsolb(nleft, t, iorder, it - 2);
          tj = t[nleft];
          ibp = iorder[nleft];
        }
        double dt = tj - tj0;
        if (dtm < dt) break;  // -> 888
        tsum += dt;
        --nleft;
        ++it;
        double dibp = d[ibp];
        d[ibp] = 0.;
        double zibp;
        if (dibp > 0.) { zibp = U(ibp) - X(ibp); xcp[ibp] = U(ibp); iwhere_[ibp] = 2; }
        else { zibp = L(ibp) - X(ibp); xcp[ibp] = L(ibp); iwhere_[ibp] = 1; }
        if (nleft == 0 && nbreak == n) { dtm = dt; to_999 = true; break; }
        ++nint_;
        double dibp2 = dibp * dibp;
        f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
        f2 -= theta * dibp2;
        if (col > 0) {
          if (dt != 0.)
            for (int j = 1; j <= col2; j++) c[j] += dt * p[j];
          int pointr = head_;
          for (int j = 1; j <= col; j++) {
            wbp[j] = WY(ibp, pointr);
            wbp[col + j] = theta * WS(ibp, pointr);
            pointr = pointr % m + 1;
          }
          if (bmv(wbp + 1, v + 1)) return 1;
          double wmc = ddot(col2, c + 1, 1, v + 1, 1);
          double wmp = ddot(col2, p + 1, 1, v + 1, 1);
          double wmw = ddot(col2, wbp + 1, 1, v + 1, 1);
          double md = -dibp;
          if (md != 0.)
            for (int j = 1; j <= col2; j++) p[j] += md * wbp[j];
          f1 += dibp * wmc;
          f2 = f2 + dibp * 2. * wmp - dibp2 * wmw;
        }
        if (nleft > 0) {
          dtm = -f1 / f2;
          continue;
        } else if (bnded) {
          f1 = 0.; f2 = 0.; dtm = 0.;
        } else {
          dtm = -f1 / f2;
        }
        break;
      }
    }
    if (!to_999) {  // 888
      if (dtm <= 0.) dtm = 0.;
      tsum += dtm;
      if (tsum != 0.)
        for (int i = 1; i <= n; i++) xcp[i] += tsum * d[i];
    }
    if (col > 0 && dtm != 0.)  // 999
      for (int j = 1; j <= col2; j++) c[j] += dtm * p[j];
    return 0;
  }

  void freev() {
    nenter_ = 0;
    ileave_ = n + 1;
    if (iter > 0 && cnstnd_) {
      for (int i = 1; i <= nfree_; i++) {
        int k = index_[i];
        if (iwhere_[k] > 0) { --ileave_; indx2_[ileave_] = k; }
      }
      for (int i = nfree_ + 1; i <= n; i++) {
        int k = index_[i];
        if (iwhere_[k] <= 0) { ++nenter_; indx2_[nenter_] = k; }
      }
    }
    wrk_ = ileave_ < n + 1 || nenter_ > 0 || updatd_;
    nfree_ = 0;
    int iact = n + 1;
    for (int i = 1; i <= n; i++) {
      if (iwhere_[i] <= 0) { ++nfree_; index_[nfree_] = i; }
      else { --iact; index_[iact] = i; }
    }
  }

  // LEL^T factorisation of the indefinite middle matrix of the subspace problem
  int formk() {
    int m = kM, col = col_, nsub = nfree_;
    if (!wn_ready_) { std::memset(wn_, 0, sizeof(double) * 8 * kM * kM); wn_ready_ = true; }
    int upcl;
    if (updatd_) {
      if (iupdat_ > m) {
        for (int jy = 1; jy <= m - 1; jy++) {
          int js = m + jy;
          for (int q = 0; q < m - jy; q++) WN1(jy + q, jy) = WN1(jy + 1 + q, jy + 1);
          for (int q = 0; q < m - jy; q++) WN1(js + q, js) = WN1(js + 1 + q, js + 1);
          for (int q = 0; q < m - 1; q++) WN1(m + 1 + q, jy) = WN1(m + 2 + q, jy + 1);
        }
      }
      int pbegin = 1, pend = nsub, dbegin = nsub + 1, dend = n;
      int iy = col, is = m + col;
      int ipntr = head_ + col - 1;
      if (ipntr > m) ipntr -= m;
      int jpntr = head_;
      for (int jy = 1; jy <= col; jy++) {
        int js = m + jy;
        double temp1 = 0., temp2 = 0., temp3 = 0.;
        for (int k = pbegin; k <= pend; k++) {
          int k1 = index_[k];
          temp1 += WY(k1, ipntr) * WY(k1, jpntr);
        }
        for (int k = dbegin; k <= dend; k++) {
          int k1 = index_[k];
          temp2 += WS(k1, ipntr) * WS(k1, jpntr);
          temp3 += WS(k1, ipntr) * WY(k1, jpntr);
        }
        WN1(iy, jy) = temp1;
        WN1(is, js) = temp2;
        WN1(is, jy) = temp3;
        jpntr = jpntr % m + 1;
      }
      int jy = col;
      jpntr = head_ + col - 1;
      if (jpntr > m) jpntr -= m;
      ipntr = head_;
      for (int i = 1; i <= col; i++) {
        is = m + i;
        double temp3 = 0.;
        for (int k = pbegin; k <= pend; k++) {
          int k1 = index_[k];
          temp3 += WS(k1, ipntr) * WY(k1, jpntr);
        }
        ipntr = ipntr % m + 1;
        WN1(is, jy) = temp3;
      }
      upcl = col - 1;
    } else {
      upcl = col;
    }
    int ipntr = head_;
    for (int iy = 1; iy <= upcl; iy++) {
      int is = m + iy;
      int jpntr = head_;
      for (int jy = 1; jy <= iy; jy++) {
        int js = m + jy;
        double temp1 = 0., temp2 = 0., temp3 = 0., temp4 = 0.;
        for (int k = 1; k <= nenter_; k++) {
          int k1 = indx2_[k];
          temp1 += WY(k1, ipntr) * WY(k1, jpntr);
          temp2 += WS(k1, ipntr) * WS(k1, jpntr);
        }
        for (int k = ileave_; k <= n; k++) {
          int k1 = indx2_[k];
          temp3 += WY(k1, ipntr) * WY(k1, jpntr);
          temp4 += WS(k1, ipntr) * WS(k1, jpntr);
        }
        WN1(iy, jy) = WN1(iy, jy) + temp1 - temp3;
        WN1(is, js) = WN1(is, js) - temp2 + temp4;
        jpntr = jpntr % m + 1;
      }
      ipntr = ipntr % m + 1;
    }
    ipntr = head_;
    for (int is = m + 1; is <= m + upcl; is++) {
      int jpntr = head_;
      for (int jy = 1; jy <= upcl; jy++) {
        double temp1 = 0., temp3 = 0.;
        for (int k = 1; k <= nenter_; k++) {
          int k1 = indx2_[k];
          temp1 += WS(k1, ipntr) * WY(k1, jpntr);
        }
        for (int k = ileave_; k <= n; k++) {
          int k1 = indx2_[k];
          temp3 += WS(k1, ipntr) * WY(k1, jpntr);
        }
        if (is <= jy + m) WN1(is, jy) = WN1(is, jy) + temp1 - temp3;
        else WN1(is, jy) = WN1(is, jy) - temp1 + temp3;
        jpntr = jpntr % m + 1;
      }
      ipntr = ipntr % m + 1;
    }
    int m2 = 2 * m;
    for (int iy = 1; iy <= col; iy++) {
      int is = col + iy, is1 = m + iy;
      for (int jy = 1; jy <= iy; jy++) {
        int js = col + jy, js1 = m + jy;
        WN(jy, iy) = WN1(iy, jy) / theta_;
        WN(js, is) = WN1(is1, js1) * theta_;
      }
      for (int jy = 1; jy <= iy - 1; jy++) WN(jy, is) = -WN1(is1, jy);
      for (int jy = iy; jy <= col; jy++) WN(jy, is) = WN1(is1, jy);
      WN(iy, iy) += SY(iy, iy);
    }
    if (dpofa(wn_, m2, col)) return -1;
    int col2 = 2 * col;
    for (int js = col + 1; js <= col2; js++)
      if (dtrsl(wn_, m2, col, &WN(1, js), 11)) return -1;  // reference ignores this info; dpofa succeeded
    for (int is = col + 1; is <= col2; is++)
      for (int js = is; js <= col2; js++) WN(is, js) += ddot(col, &WN(1, is), 1, &WN(1, js), 1);
    if (dpofa(&WN(col + 1, col + 1), m2, col)) return -2;
    return 0;
  }

  int formt() {
    int col = col_;
    for (int j = 1; j <= col; j++) WT(1, j) = theta_ * SS(1, j);
    for (int i = 2; i <= col; i++)
      for (int j = i; j <= col; j++) {
        int k1 = (i < j ? i : j) - 1;
        double ddum = 0.;
        for (int k = 1; k <= k1; k++) ddum += SY(i, k) * SY(j, k) / SY(k, k);
        WT(i, j) = ddum + theta_ * SS(i, j);
      }
    if (dpofa(wt_, kM, col)) return -3;
    return 0;
  }

  // r = -Z'(B(xcp - x) + g) for the free variables
  int cmprlb() {
    int col = col_, m = kM;
    if (!cnstnd_ && col > 0) {
      for (int i = 1; i <= n; i++) r_[i] = -G(i);
    } else {
      for (int i = 1; i <= nfree_; i++) {
        int k = index_[i];
        r_[i] = -theta_ * (z_[k] - X(k)) - G(k);
      }
      if (bmv(wa_ + 2 * m + 1, wa_ + 1)) return -8;
      int pointr = head_;
      for (int j = 1; j <= col; j++) {
        double a1 = wa_[j], a2 = theta_ * wa_[col + j];
        for (int i = 1; i <= nfree_; i++) {
          int k = index_[i];
          r_[i] = r_[i] + WY(k, pointr) * a1 + WS(k, pointr) * a2;
        }
        pointr = pointr % m + 1;
      }
    }
    return 0;
  }

  // subspace minimisation; on entry z_ = xcp, r_ = reduced gradient; on exit z_ = subspace minimiser
  // projected back to the box along the step (the pre-3.0 behaviour the reference has)
  int subsm() {
    int nsub = nfree_, col = col_, m = kM;
    double *wv = wa_;  // 1-based
    double *d = r_;
    if (nsub <= 0) return 0;
    int pointr = head_;
    for (int i = 1; i <= col; i++) {
      double temp1 = 0., temp2 = 0.;
      for (int j = 1; j <= nsub; j++) {
        int k = index_[j];
        temp1 += WY(k, pointr) * d[j];
        temp2 += WS(k, pointr) * d[j];
      }
      wv[i] = temp1;
      wv[col + i] = theta_ * temp2;
      pointr = pointr % m + 1;
    }
    int m2 = 2 * m, col2 = 2 * col;
    if (dtrsl(wn_, m2, col2, wv + 1, 11)) return 1;
    for (int i = 1; i <= col; i++) wv[i] = -wv[i];
    if (dtrsl(wn_, m2, col2, wv + 1, 1)) return 1;
    pointr = head_;
    for (int jy = 1; jy <= col; jy++) {
      int js = col + jy;
      for (int i = 1; i <= nsub; i++) {
        int k = index_[i];
        d[i] = d[i] + WY(k, pointr) * wv[jy] / theta_ + WS(k, pointr) * wv[js];
      }
      pointr = pointr % m + 1;
    }
    for (int i = 1; i <= nsub; i++) d[i] /= theta_;
    double alpha = 1., temp1 = alpha;
    int ibd = 0;
    for (int i = 1; i <= nsub; i++) {
      int k = index_[i];
      double dk = d[i];
      if (NBD(k) != 0) {
        if (dk < 0. && NBD(k) <= 2) {
          double temp2 = L(k) - z_[k];
          if (temp2 >= 0.) temp1 = 0.;
          else if (dk * alpha < temp2) temp1 = temp2 / dk;
        } else if (dk > 0. && NBD(k) >= 2) {
          double temp2 = U(k) - z_[k];
          if (temp2 <= 0.) temp1 = 0.;
          else if (dk * alpha > temp2) temp1 = temp2 / dk;
        }
        if (temp1 < alpha) { alpha = temp1; ibd = i; }
      }
    }
    if (alpha < 1.) {
      double dk = d[ibd];
      int k = index_[ibd];
      if (dk > 0.) { z_[k] = U(k); d[ibd] = 0.; }
      else if (dk < 0.) { z_[k] = L(k); d[ibd] = 0.; }
    }
    for (int i = 1; i <= nsub; i++) {
      int k = index_[i];
      z_[k] += alpha * d[i];
    }
    iword_ = alpha < 1. ? 1 : 0;
    return 0;
  }

  // More-Thuente safeguarded step (dcstep)
  static void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp, double fp,
                     double dp, bool &brackt, double stpmin, double stpmax) {
    double sgnd = dp * (dx / std::fabs(dx));
    double stpf, stpc, stpq, theta, s, gamma, p, q, r;
    if (fp > fx) {
      theta = (fx - fp) * 3. / (stp - stx) + dx + dp;
      s = std::fmax(std::fmax(std::fabs(theta), std::fabs(dx)), std::fabs(dp));
      double d1 = theta / s;
      gamma = s * std::sqrt(d1 * d1 - dx / s * (dp / s));
      if (stp < stx) gamma = -gamma;
      p = gamma - dx + theta;
      q = gamma - dx + gamma + dp;
      r = p / q;
      stpc = stx + r * (stp - stx);
      stpq = stx + dx / ((fx - fp) / (stp - stx) + dx) / 2. * (stp - stx);
      if (std::fabs(stpc - stx) < std::fabs(stpq - stx)) stpf = stpc;
      else stpf = stpc + (stpq - stpc) / 2.;
      brackt = true;
    } else if (sgnd < 0.) {
      theta = (fx - fp) * 3. / (stp - stx) + dx + dp;
      s = std::fmax(std::fmax(std::fabs(theta), std::fabs(dx)), std::fabs(dp));
      double d1 = theta / s;
      gamma = s * std::sqrt(d1 * d1 - dx / s * (dp / s));
      if (stp > stx) gamma = -gamma;
      p = gamma - dp + theta;
      q = gamma - dp + gamma + dx;
      r = p / q;
      stpc = stp + r * (stx - stp);
      stpq = stp + dp / (dp - dx) * (stx - stp);
      if (std::fabs(stpc - stp) > std::fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      brackt = true;
    } else if (std::fabs(dp) < std::fabs(dx)) {
      theta = (fx - fp) * 3. / (stp - stx) + dx + dp;
      s = std::fmax(std::fmax(std::fabs(theta), std::fabs(dx)), std::fabs(dp));
      double d3 = theta / s;
      gamma = s * std::sqrt(std::fmax(0., d3 * d3 - dx / s * (dp / s)));
      if (stp > stx) gamma = -gamma;
      p = gamma - dp + theta;
      q = gamma + (dx - dp) + gamma;
      r = p / q;
      if (r < 0. && gamma != 0.) stpc = stp + r * (stx - stp);
      else if (stp > stx) stpc = stpmax;
      else stpc = stpmin;
      stpq = stp + dp / (dp - dx) * (stx - stp);
      if (brackt) {
        if (std::fabs(stpc - stp) < std::fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        if (stp > stx) stpf = std::fmin(stp + (sty - stp) * .66, stpf);
        else stpf = std::fmax(stp + (sty - stp) * .66, stpf);
      } else {
        if (std::fabs(stpc - stp) > std::fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        stpf = std::fmin(stpmax, stpf);
        stpf = std::fmax(stpmin, stpf);
      }
    } else {
      if (brackt) {
        theta = (fp - fy) * 3. / (sty - stp) + dy + dp;
        s = std::fmax(std::fmax(std::fabs(theta), std::fabs(dy)), std::fabs(dp));
        double d1 = theta / s;
        gamma = s * std::sqrt(d1 * d1 - dy / s * (dp / s));
        if (stp > sty) gamma = -gamma;
        p = gamma - dp + theta;
        q = gamma - dp + gamma + dy;
        r = p / q;
        stpc = stp + r * (sty - stp);
        stpf = stpc;
      } else if (stp > stx) stpf = stpmax;
      else stpf = stpmin;
    }
    if (fp > fx) {
      sty = stp; fy = fp; dy = dp;
    } else {
      if (sgnd < 0.) { sty = stx; fy = fx; dy = dx; }
      stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
  }

  // More-Thuente line search (dcsrch) with ftol = 1e-3, gtol = 0.9, xtol = 0.1, stpmin = 0 (src/bfgs.c:232-234)
  void dcsrch(double fv, double gv, double &stp, double stpmax) {
    const double ftol = .001, gtol = .9, xtol = .1, stpmin = 0.;
    if (lstask_ == LS_START) {
      if (stp < stpmin || stp > stpmax || gv >= 0. || stpmax < stpmin) { lstask_ = LS_ERROR; return; }
      brackt_ = false;
      stage_ = 1;
      finit_ = fv;
      ginit_ = gv;
      gtest_ = ftol * ginit_;
      width_ = stpmax - stpmin;
      width1_ = width_ / .5;
      stx_ = 0.; fx_ = finit_; gx_ = ginit_;
      sty_ = 0.; fy_ = finit_; gy_ = ginit_;
      stmin_ = 0.;
      stmax_ = stp + stp * 4.;
      lstask_ = LS_FG;
      return;
    }
    double ftest = finit_ + stp * gtest_;
    if (stage_ == 1 && fv <= ftest && gv >= 0.) stage_ = 2;
    LsTask nt = LS_FG;
    if (brackt_ && (stp <= stmin_ || stp >= stmax_)) nt = LS_WARN;
    if (brackt_ && stmax_ - stmin_ <= xtol * stmax_) nt = LS_WARN;
    if (stp == stpmax && fv <= ftest && gv <= gtest_) nt = LS_WARN;
    if (stp == stpmin && (fv > ftest || gv >= gtest_)) nt = LS_WARN;
    if (fv <= ftest && std::fabs(gv) <= gtol * (-ginit_)) nt = LS_CONV;
    if (nt != LS_FG) { lstask_ = nt; return; }
    if (stage_ == 1 && fv <= fx_ && fv > ftest) {
      double fm = fv - stp * gtest_, fxm = fx_ - stx_ * gtest_, fym = fy_ - sty_ * gtest_;
      double gm = gv - gtest_, gxm = gx_ - gtest_, gym = gy_ - gtest_;
      dcstep(stx_, fxm, gxm, sty_, fym, gym, stp, fm, gm, brackt_, stmin_, stmax_);
      fx_ = fxm + stx_ * gtest_;
      fy_ = fym + sty_ * gtest_;
      gx_ = gxm + gtest_;
      gy_ = gym + gtest_;
    } else {
      dcstep(stx_, fx_, gx_, sty_, fy_, gy_, stp, fv, gv, brackt_, stmin_, stmax_);
    }
    if (brackt_) {
      if (std::fabs(sty_ - stx_) >= width1_ * .66) stp = stx_ + (sty_ - stx_) * .5;
      width1_ = width_;
      width_ = std::fabs(sty_ - stx_);
    }
    if (brackt_) {
      stmin_ = std::fmin(stx_, sty_);
      stmax_ = std::fmax(stx_, sty_);
    } else {
      stmin_ = stp + (stp - stx_) * 1.1;
      stmax_ = stp + (stp - stx_) * 4.;
    }
    stp = std::fmax(stp, stpmin);
    stp = std::fmin(stp, stpmax);
    if ((brackt_ && (stp <= stmin_ || stp >= stmax_)) || (brackt_ && stmax_ - stmin_ <= xtol * stmax_)) stp = stx_;
    lstask_ = LS_FG;
  }

  // lnsrlb: returns true if a new function value is needed (task FG_LNSRCH), false when the search ended
  // (info_ != 0 or NEW_X).  `fresh` = entering for a new direction.
  bool lnsrlb(bool fresh) {
    if (fresh) {
      dtd_ = ddot(n, d_ + 1, 1, d_ + 1, 1);
      dnorm_ = std::sqrt(dtd_);
      stpmx_ = 1e10;
      if (cnstnd_) {
        if (iter == 0) {
          stpmx_ = 1.;
        } else {
          for (int i = 1; i <= n; i++) {
            double a1 = d_[i];
            if (NBD(i) != 0) {
              if (a1 < 0. && NBD(i) <= 2) {
                double a2 = L(i) - X(i);
                if (a2 >= 0.) stpmx_ = 0.;
                else if (a1 * stpmx_ < a2) stpmx_ = a2 / a1;
              } else if (a1 > 0. && NBD(i) >= 2) {
                double a2 = U(i) - X(i);
                if (a2 <= 0.) stpmx_ = 0.;
                else if (a1 * stpmx_ > a2) stpmx_ = a2 / a1;
              }
            }
          }
        }
      }
      if (iter == 0 && !boxed_) stp_ = std::fmin(1. / dnorm_, stpmx_);
      else stp_ = 1.;
      for (int i = 1; i <= n; i++) { t_[i] = X(i); xs_[i] = G(i); }  // t = x, r = g (saved copies)
      fold_ = f;
      ifun_ = 0;
      iback_ = 0;
      lstask_ = LS_START;
    }
    gd_ = ddot(n, g, 1, d_ + 1, 1);
    if (ifun_ == 0) {
      gdold_ = gd_;
      if (gd_ >= 0.) { info_ = -4; return false; }
    }
    dcsrch(f, gd_, stp_, stpmx_);
    if (lstask_ != LS_CONV && lstask_ != LS_WARN) {
      // (an LS_ERROR from dcsrch's START checks is treated like "FG" by the reference's string tests)
      ++ifun_;
      ++nfgv;
      iback_ = ifun_ - 1;
      if (stp_ == 1.) {
        for (int i = 1; i <= n; i++) X(i) = z_[i];
      } else {
        for (int i = 1; i <= n; i++) X(i) = stp_ * d_[i] + t_[i];
      }
      return true;
    }
    return false;
  }

  void matupd(double rr, double dr) {
    int m = kM;
    if (iupdat_ <= m) {
      col_ = iupdat_;
      itail_ = (head_ + iupdat_ - 2) % m + 1;
    } else {
      itail_ = itail_ % m + 1;
      head_ = head_ % m + 1;
    }
    for (int i = 1; i <= n; i++) { WS(i, itail_) = d_[i]; WY(i, itail_) = r_[i]; }
    theta_ = rr / dr;
    if (iupdat_ > m) {
      for (int j = 1; j <= col_ - 1; j++) {
        for (int q = 0; q < j; q++) SS(1 + q, j) = SS(2 + q, j + 1);
        for (int q = 0; q < col_ - j; q++) SY(j + q, j) = SY(j + 1 + q, j + 1);
      }
    }
    int pointr = head_;
    for (int j = 1; j <= col_ - 1; j++) {
      SY(col_, j) = ddot(n, d_ + 1, 1, &WY(1, pointr), 1);
      SS(j, col_) = ddot(n, &WS(1, pointr), 1, d_ + 1, 1);
      pointr = pointr % m + 1;
    }
    if (stp_ == 1.) SS(col_, col_) = dtd_;
    else SS(col_, col_) = stp_ * stp_ * dtd_;
    SY(col_, col_) = dr;
  }

  // ---- mainlb control flow ----
  void start() {
    started_ = true;
    epsmch_ = 2.220446049250313e-16;  // dpmeps_() on IEEE double
    col_ = 0; head_ = 1; theta_ = 1.; iupdat_ = 0; updatd_ = false;
    iter = 0; nfgv = 0; nint_ = 0; nskip_ = 0; nfree_ = n;
    tol_ = factr * epsmch_;
    info_ = 0;
    itail_ = 0; iback_ = 0; ifun_ = 0; iword_ = 0; nact_ = 0; ileave_ = 0; nenter_ = 0;
    for (int i = 0; i < n; i++)
      if (lo[i] > hi[i]) { task_ = ERROR_INPUT; return; }
    active();
    task_ = FG_START;
  }

  void resume() {
    if (task_ == FG_START) {
      nfgv = 1;
      projgr();
      if (sbgnrm_ <= pgtol) { task_ = CONVERGED_PGTOL; return; }
      iterate_from_222();
      return;
    }
    if (task_ == FG_LNSRCH) { line_search_666(false); return; }
    if (task_ == NEW_X) { after_new_x_777(); return; }
    // terminal tasks: nothing to do
  }

  void iterate_from_222() {
    for (;;) {  // label 222
      iword_ = -1;
      bool skip_to_333 = false;
      if (!cnstnd_ && col_ > 0) {
        for (int i = 1; i <= n; i++) z_[i] = X(i);
        wrk_ = updatd_;
        nint_ = 0;
        skip_to_333 = true;
      }
      if (!skip_to_333) {
        if (cauchy()) { reset_memory(); continue; }
        freev();
        nact_ = n - nfree_;
      }
      // 333
      if (nfree_ != 0 && col_ != 0) {
        if (wrk_) {
          int fi = formk();
          if (fi != 0) { reset_memory(); continue; }
        }
        int ci = cmprlb();
        if (ci == 0) ci = subsm();
        if (ci != 0) { reset_memory(); continue; }
      }
      // 555
      for (int i = 1; i <= n; i++) d_[i] = z_[i] - X(i);
      if (line_search_666(true)) continue;  // restart requested (RESTART_FROM_LNSRCH)
      return;
    }
  }

  // returns true when the caller (iterate_from_222) must loop again
  bool line_search_666(bool fresh) {
    bool need_f = lnsrlb(fresh);
    if (info_ != 0 || iback_ >= 20) {
      for (int i = 1; i <= n; i++) { X(i) = t_[i]; G(i) = xs_[i]; }
      f = fold_;
      if (col_ == 0) {
        if (info_ == 0) { info_ = -9; --nfgv; --ifun_; --iback_; }
        task_ = ABNORMAL_LNSRCH;
        ++iter;
        return false;
      }
      if (info_ == 0) --nfgv;
      reset_memory();
      if (fresh) return true;
      iterate_from_222();
      return false;
    }
    if (need_f) { task_ = FG_LNSRCH; return false; }
    ++iter;
    projgr();
    task_ = NEW_X;
    return false;
  }

  void after_new_x_777() {
    if (sbgnrm_ <= pgtol) { task_ = CONVERGED_PGTOL; return; }
    double ddum = std::fmax(std::fmax(std::fabs(fold_), std::fabs(f)), 1.);
    if (fold_ - f <= tol_ * ddum) {
      task_ = CONVERGED_FACTR;
      if (iback_ >= 10) info_ = -5;
      return;
    }
    for (int i = 1; i <= n; i++) r_[i] = G(i) - xs_[i];
    double rr = ddot(n, r_ + 1, 1, r_ + 1, 1);
    double dr, dd;
    if (stp_ == 1.) {
      dr = gd_ - gdold_;
      dd = -gdold_;
    } else {
      dr = (gd_ - gdold_) * stp_;
      for (int i = 1; i <= n; i++) d_[i] = stp_ * d_[i];
      dd = -gdold_ * stp_;
    }
    if (dr <= epsmch_ * dd) {
      ++nskip_;
      updatd_ = false;
    } else {
      updatd_ = true;
      ++iupdat_;
      matupd(rr, dr);
      if (formt() != 0) reset_memory();
    }
    iterate_from_222();
  }
};

}  // namespace relate
