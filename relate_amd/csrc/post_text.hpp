// Text form of posterior rows: write_dMatrix_to_file (src/relateHMM.cpp:117-136) = for every SNP
// `ostream << setprecision(POSTPRECISION=4) << value << '\t'` (i.e. printf "%.4g"), "-1" for SNPs dropped for the pair,
// '\n' after the row.  Host code only.  At 5·10^5 SNPs a block of posteriors is 10^8-10^9 values and stdio's "%.4g"
// (150-250 ns per value on one thread) would take far longer than the GPU needs to produce them, so the rows are
// formatted by the host threads with a formatter specialised to four significant digits; every value whose fourth
// digit is not decided with a wide margin by double arithmetic goes through snprintf, so the bytes are those of
// printf in all cases (tests/test_post_text.py compares them on 10^7 values, ties and decade boundaries included).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace relate_text {

constexpr int kMaxValueChars = 13;  // "-1.234e-308" + tab, with room to spare

struct Pow10Table {
  double p[617];  // 10^k for k = -308 .. 308, correctly rounded
  Pow10Table() {
    char buf[16];
    for (int k = -308; k <= 308; k++) {
      std::snprintf(buf, sizeof buf, "1e%d", k);
      p[k + 308] = std::strtod(buf, nullptr);
    }
  }
  double operator()(int k) const { return p[k + 308]; }
};
inline const Pow10Table &pow10_table() {
  static const Pow10Table t;
  return t;
}

inline char *put_slow(char *o, double v) { return o + std::snprintf(o, 32, "%.4g", v); }

// "%.4g" of v into o (no terminator); returns the end.  o must have 32 bytes of room.
inline char *put_g4(char *o, double v) {
  if (v == 0.0) {
    if (std::signbit(v)) *o++ = '-';
    *o++ = '0';
    return o;
  }
  if (!(v > 1e-290 && v < 1e290)) return put_slow(o, v);  // negative, tiny, huge, inf, nan
  const Pow10Table &T = pow10_table();
  int e2;
  (void)std::frexp(v, &e2);                                // v = m 2^e2, m in [0.5, 1)
  int X = (int)std::floor((e2 - 1) * 0.30102999566398120);  // floor(log10 v) or one less
  if (v >= T(X + 1)) X++;
  // four digits: d = round(v 10^(3-X)).  The product carries two roundings (the table entry when |3-X| > 22, and the
  // multiplication): < 4e-12 absolute at 10^4.  A value that close to a rounding boundary or a decade goes the slow way.
  double sc = v * T(3 - X);
  double fl = std::floor(sc);
  double fr = sc - fl;
  // (X is exact, so the true product is >= 1000: sc < 1000 is rounding error at a decade; >= 9999.5 rounds into the
  // next decade and changes the exponent)
  if (std::fabs(fr - 0.5) < 1e-6 || sc < 1000.0 || sc >= 9999.5 - 1e-6) return put_slow(o, v);
  int d = (int)fl + (fr > 0.5 ? 1 : 0);
  int dg[4] = {d / 1000, (d / 100) % 10, (d / 10) % 10, d % 10};
  int nd = 4;
  while (nd > 1 && dg[nd - 1] == 0) nd--;  // %g drops trailing zeros
  if (X >= -4 && X < 4) {
    if (X >= 0) {
      for (int i = 0; i <= X; i++) *o++ = (char)('0' + dg[i]);  // integer part: digits 0..X (zeros included)
      if (nd > X + 1) {
        *o++ = '.';
        for (int i = X + 1; i < nd; i++) *o++ = (char)('0' + dg[i]);
      }
    } else {
      *o++ = '0';
      *o++ = '.';
      for (int i = 0; i < -X - 1; i++) *o++ = '0';
      for (int i = 0; i < nd; i++) *o++ = (char)('0' + dg[i]);
    }
    return o;
  }
  *o++ = (char)('0' + dg[0]);
  if (nd > 1) {
    *o++ = '.';
    for (int i = 1; i < nd; i++) *o++ = (char)('0' + dg[i]);
  }
  *o++ = 'e';
  int ax = X < 0 ? -X : X;
  *o++ = X < 0 ? '-' : '+';
  if (ax >= 100) { *o++ = (char)('0' + ax / 100); ax %= 100; *o++ = (char)('0' + ax / 10); *o++ = (char)('0' + ax % 10); }
  else { *o++ = (char)('0' + ax / 10); *o++ = (char)('0' + ax % 10); }
  return o;
}

// One row of M values; returns the number of bytes written (at most M * kMaxValueChars + 1).
inline size_t format_row(const double *row, long M, char *out) {
  char *o = out;
  for (long s = 0; s < M; s++) {
    double v = row[s];
    if (v == -1.0) { *o++ = '-'; *o++ = '1'; }
    else o = put_g4(o, v);
    *o++ = '\t';
  }
  *o++ = '\n';
  return (size_t)(o - out);
}

}  // namespace relate_text
