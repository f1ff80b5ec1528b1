// Check that q1 = fma(fma(-q0, b, a), 1/b, q0), q0 = a * (1/b), equals the IEEE quotient a / b (gn_lane.h: div_res).
// gcc -O2 -mfma -ffp-contract=off profiles/tools/div_check.c -o /tmp/div_check -lm && /tmp/div_check   ("total mismatches: 0")
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double urand(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }
int main(void) {
  const double bs[] = {10.0 / 256.0, 10.0 / 128.0, 12.0 / 200.0, 0.1, 10.0 / 400.0, 20.0 / 333.0, 1.0 / 3.0, 0.0390625 * 1.0000000001};
  long bad_total = 0;
  for (int bi = 0; bi < 8 + 2000; ++bi) {
    const double b = bi < 8 ? bs[bi] : (0.001 + urand() * 2.0);
    const double y = 1.0 / b;
    long bad = 0;
    const long N = bi < 8 ? 200000000L : 500000L;
    for (long i = 0; i < N; ++i) {
      double a;
      const uint64_t k = rnd() & 3;
      if (k == 0) a = (urand() - 0.5) * 20.0;            /* workspace coordinates */
      else if (k == 1) a = (urand() - 0.5) * 2e-3;        /* tiny */
      else if (k == 2) a = (urand() - 0.5) * 2e6;         /* large */
      else a = (double)((long)(urand() * 4096) - 2048) * b * (1.0 + (urand() - 0.5) * 1e-15);   /* near integer multiples of b */
      const double q0 = a * y;
      const double r = fma(-q0, b, a);
      const double q1 = fma(r, y, q0);
      if (q1 != a / b) ++bad;
    }
    if (bad) printf("b=%.17g: %ld mismatches of %ld\n", b, bad, N);
    bad_total += bad;
  }
  printf("total mismatches: %ld\n", bad_total);
  return 0;
}
