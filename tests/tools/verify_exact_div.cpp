// verify_exact_div.cpp -- evidence for the division strength reduction in csrc/integrate.hip.
// For a divisor b with y = RN(1/b):   q = RN(a*y);  r = fma(-b, q, a) (exact);  q' = fma(r, y, q)
// equals the IEEE quotient RN(a/b) only for suitable divisors.  This tool checks it EXHAUSTIVELY over every finite
// float a for the constant divisors the kernel uses it for (32767, 255), exhaustively over the operand range
// 2^-24 <= |a| < 2^11 for the integer weights 1..256 (the dividends are weighted sums of values in [-1, 255]), and
// reports -- as the reason the kernel keeps a true IEEE division there -- the mismatch rate for typical mu values.
//   g++ -O2 -fopenmp -mfma -ffp-contract=off -o verify_exact_div verify_exact_div.cpp && ./verify_exact_div
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>

static inline float markstein(float a, float b, float y) {
  const float q = a * y;
  const float r = fmaf(-b, q, a);
  return fmaf(r, y, q);
}
static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static long long check_all(float b, uint32_t stride, int exp_lo = 0, int exp_hi = 254) {
  const float y = 1.0f / b;
  long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (long long i = 0; i < (1ll << 32); i += stride) {
    const uint32_t u = (uint32_t)i;
    const int ex = (int)((u >> 23) & 0xff);
    if (ex == 0xff || ex < exp_lo || ex > exp_hi) continue;  // inf / nan / outside the requested binades
    const float a = from_bits(u);
    const float want = a / b, got = markstein(a, b, y);
    if (bits(want) != bits(got) && !(want == 0.0f && got == 0.0f)) {
      // results in the subnormal range may differ (double rounding of q); the kernel never divides such values
      // ... and quotients beyond FLT_MAX overflow differently (inf vs nan); |a| is < 1e3 in the kernel
      if (fabsf(want) >= 1.17549435e-38f * 16.0f && fabsf(a) <= 1e30f) bad++;
    }
  }
  return bad;
}

int main() {
  long long total_bad = 0;
  const float consts[] = {32767.0f, 255.0f};
  for (float b : consts) {
    long long bad = check_all(b, 1);
    printf("divisor %-10g exhaustive: %lld mismatches\n", b, bad);
    total_bad += bad;
  }
  long long bad_w = 0;
  for (int w = 1; w <= 256; w++) bad_w += check_all((float)w, 1, 127 - 24, 127 + 10);
  printf("integer divisors 1..256, every float with 2^-24 <= |a| < 2^11: %lld mismatches\n", bad_w);
  total_bad += bad_w;
  for (float b : {0.2f, 0.02f}) printf("(not used) divisor %g: %lld mismatches -> IEEE division kept for mu\n", b, check_all(b, 1));
  return total_bad ? 1 : 0;
}
