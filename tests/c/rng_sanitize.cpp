// Sanitizer harness for the host-side MT19937 stream (csrc/host_rng.cpp): exact-size heap buffers
// for the state and the output, every way of crossing a 624-word state block, odd / even start
// positions.  Built with -fsanitize=address,undefined by tests/test_host_rng.py (CPU only: GPU
// sanitizer builds are not available on the pool).  Prints a checksum per case for the caller to
// compare with NumPy.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" int sw_mt19937_uniform_pm1(uint32_t *key, int32_t *pos, int64_t n, double *out);

// numpy's init_genrand(seed) (mt19937_seed)
static void seed_state(uint32_t *mt, uint32_t seed)
{
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

int main()
{
    const int64_t sizes[] = {0, 1, 2, 3, 311, 312, 313, 623, 624, 625, 5000, 1, 70001, 2};
    uint32_t *key = (uint32_t *)malloc(624 * sizeof(uint32_t));   // exactly 624 words
    seed_state(key, 12345u);
    int32_t pos = 624;                                            // numpy: "state exhausted" after seeding
    for (unsigned c = 0; c < sizeof sizes / sizeof sizes[0]; ++c) {
        const int64_t n = sizes[c];
        double *out = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));   // exactly n doubles
        const int rc = sw_mt19937_uniform_pm1(key, &pos, n, n ? out : NULL);
        if (rc != 0) { printf("case %u: rc %d\n", c, rc); return 1; }
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            if (!(out[i] >= -1.0 && out[i] < 1.0)) { printf("case %u: value out of range\n", c); return 1; }
            s += out[i] * (double)((i % 7) + 1);
        }
        printf("%lld %.17g %d\n", (long long)n, s, pos);
        free(out);
    }
    // bad arguments are refused, nothing is touched
    int32_t bad = 625;
    if (sw_mt19937_uniform_pm1(key, &bad, 4, (double *)key) != 3) return 1;
    if (sw_mt19937_uniform_pm1(NULL, &pos, 4, (double *)key) != 1) return 1;
    free(key);
    return 0;
}
