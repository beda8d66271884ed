// verify_exact_div.cpp -- evidence for the division strength reduction in csrc/integrate.hip.
// For a divisor b with y = RN(1/b):   q = RN(a*y);  r = fma(-b, q, a) (exact);  q' = fma(r, y, q)
// is claimed to equal the IEEE quotient RN(a/b) (Markstein).  This tool checks it EXHAUSTIVELY over every finite
// float a for the constant divisors the kernel uses (32767, 255, and the mu values of the workloads), and over a
// dense sample for the integer weights 1..256.
//   g++ -O2 -fopenmp -mfma -ffp-contract=off -o verify_exact_div verify_exact_div.cpp && ./verify_exact_div
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

static inline float markstein(float a, float b, float y) {
  const float q = a * y;
  const float r = fmaf(-b, q, a);
  return fmaf(r, y, q);
}
static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static long long check_all(float b, uint32_t stride) {
  const float y = 1.0f / b;
  long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (long long i = 0; i < (1ll << 32); i += stride) {
    const uint32_t u = (uint32_t)i;
    if (((u >> 23) & 0xff) == 0xff) continue;  // inf / nan
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
  const float consts[] = {32767.0f, 255.0f, 0.02f, 0.2f, 0.08f, 0.04f, 0.005f, 0.05f, 1.0f / 3.0f};
  for (float b : consts) {
    long long bad = check_all(b, 1);
    printf("divisor %-10g exhaustive: %lld mismatches\n", b, bad);
    total_bad += bad;
  }
  long long bad_w = 0;
  for (int w = 1; w <= 256; w++) bad_w += check_all((float)w, 61);  // every 61st float (coprime stride)
  printf("integer divisors 1..256, every 61st float: %lld mismatches\n", bad_w);
  total_bad += bad_w;
  return total_bad ? 1 : 0;
}
