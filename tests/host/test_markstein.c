// TEST INFRASTRUCTURE.  The three-operation division the chain wave of csrc/cd_team.hip uses for the soft-threshold step
// (sklearn divides by Q_ii + beta, _cd_fast.pyx:667):
//     r = RN(1 / b) (a true division, once per feature);  q0 = RN(a r);  rem = RN(a - b q0) (exact: fma);  q1 = RN(q0 + rem r)
// must equal the hardware's correctly rounded a / b for every operand pair away from over / underflow (Markstein 1990).
// Brute force: random mantissas, adversarial divisors / dividends (all-ones mantissas, near powers of two, few-bit
// patterns), and structured quotients (exactly representable, near half-way).  usage: test_markstein [random_pairs]
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static inline double mk(uint64_t mant, int e) { uint64_t u = ((uint64_t)(e + 1023) << 52) | (mant & 0xFFFFFFFFFFFFFull); double d; memcpy(&d, &u, 8); return d; }
static long bad = 0, tot = 0;
static inline void chk(double a, double b) {
    double r = 1.0 / b, q0 = a * r, rem = fma(-b, q0, a), q1 = fma(rem, r, q0), ref = a / b;
    ++tot;
    if (q1 != ref) { if (bad < 20) printf("MISMATCH a=%a b=%a q1=%a ref=%a\n", a, b, q1, ref); ++bad; }
}
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 200000000L;
    for (long i = 0; i < n; ++i) {  // random mantissas, moderate exponents
        chk(mk(rnd(), (int)(rnd() % 200) - 100), mk(rnd(), (int)(rnd() % 200) - 100));
    }
    printf("random: %ld mismatches of %ld\n", bad, tot);
    // adversarial b: all-ones mantissas, near powers of two, few-bit mantissas; a likewise
    uint64_t specials[64]; int ns = 0;
    specials[ns++] = 0; specials[ns++] = 1; specials[ns++] = 2; specials[ns++] = 3;
    specials[ns++] = 0xFFFFFFFFFFFFFull; specials[ns++] = 0xFFFFFFFFFFFFEull; specials[ns++] = 0xFFFFFFFFFFFFDull;
    specials[ns++] = 0x8000000000000ull; specials[ns++] = 0x7FFFFFFFFFFFFull; specials[ns++] = 0x8000000000001ull;
    specials[ns++] = 0x5555555555555ull; specials[ns++] = 0xAAAAAAAAAAAAAull; specials[ns++] = 0x6A09E667F3BCDull; /* sqrt2 */
    for (int k = 4; k < 52; k += 3) specials[ns++] = (1ull << k) - 1;
    for (int k = 4; k < 52; k += 5) specials[ns++] = 0xFFFFFFFFFFFFFull ^ ((1ull << k) - 1);
    long bad0 = bad, tot0 = tot;
    for (int i = 0; i < ns; ++i)
        for (long j = 0; j < 3000000; ++j) {
            chk(mk(rnd(), (int)(rnd() % 60) - 30), mk(specials[i], (int)(rnd() % 60) - 30));
            chk(mk(specials[i], (int)(rnd() % 60) - 30), mk(rnd(), (int)(rnd() % 60) - 30));
            chk(mk(specials[i] + (rnd() & 7), 0), mk(specials[(i + j) % ns] ^ (rnd() & 7), 3));
        }
    printf("adversarial: %ld mismatches of %ld\n", bad - bad0, tot - tot0);
    // quotients that are exactly representable or exact midpoints-ish: a = b * small integer / power of two
    bad0 = bad; tot0 = tot;
    for (long j = 0; j < 50000000; ++j) {
        double b = mk(rnd(), 0); double k = (double)(rnd() % 4096 + 1);
        chk(b * k, b); chk(k, b); chk(b, k);
    }
    printf("structured: %ld mismatches of %ld\n", bad - bad0, tot - tot0);
    return bad != 0;
}
