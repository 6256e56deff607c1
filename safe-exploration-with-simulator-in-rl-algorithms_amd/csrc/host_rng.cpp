// host_rng.cpp -- NumPy's legacy MT19937 uniform stream, generated natively on the host.
//
// The reference draws its exploration noise from NumPy's global legacy generator:
// np.random.seed(seed) (ars/ars_agent.py:95) and N calls of 2*np.random.rand(m, d)-1
// (ars/ars_agent.py:137-138).  Keeping that exact stream is part of the drop-in contract,
// but NumPy produces it at ~5 ns per double on one core: at 4096 directions (8 GPUs x 512)
// that is 0.33 ms per iteration on EVERY rank (every rank needs every direction's noise for the
// update) -- as long as the whole GPU iteration; at n = 6 (70 doubles per direction) four times
// that.  This file continues the same stream from the same state an order of magnitude faster
// (block-wise twist, tempering and conversion that the compiler vectorises), so the host stays
// ahead of the GPU at any rank count.  State goes in and out in the form np.random.get_state() /
// set_state() use (key[624], pos), so NumPy's generator can be advanced in lock step.
//
// The library is built on one machine and runs on another, so the vector width is chosen at RUN
// time: the same code is compiled three times (baseline x86-64, AVX2, AVX-512) and the first call
// picks the widest the CPU has (__builtin_cpu_supports).  All three produce the same bits.
//
// Algorithm: Matsumoto & Nishimura's MT19937 exactly as numpy/random/src/mt19937 implements
// it (mt19937_gen when pos == 624, standard tempering); a double is
// ((a >> 5) * 67108864 + (b >> 6)) / 9007199254740992 from two consecutive outputs
// (numpy legacy random_sample).
#include <stdint.h>
#include <string.h>

#include "../../include/swimmer_hip.h"

namespace {

constexpr int kN = 624, kM = 397;
constexpr uint32_t kMatrixA = 0x9908b0dfu, kUpper = 0x80000000u, kLower = 0x7fffffffu;

#define SW_RNG_INLINE inline __attribute__((always_inline))

SW_RNG_INLINE uint32_t twist(uint32_t u, uint32_t v, uint32_t far)
{
    const uint32_t y = (u & kUpper) | (v & kLower);
    return far ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
}

// One full regeneration of the 624-word state (numpy: mt19937_gen).  The two main loops
// have no loop-carried dependence shorter than 227 iterations, so they vectorise.
SW_RNG_INLINE void regenerate(uint32_t *mt)
{
    int i = 0;
#pragma clang loop vectorize(assume_safety)
#pragma GCC ivdep
    for (i = 0; i < kN - kM; ++i) mt[i] = twist(mt[i], mt[i + 1], mt[i + kM]);
#pragma clang loop vectorize(assume_safety)
#pragma GCC ivdep
    for (i = kN - kM; i < kN - 1; ++i) mt[i] = twist(mt[i], mt[i + 1], mt[i + (kM - kN)]);
    mt[kN - 1] = twist(mt[kN - 1], mt[0], mt[kM - 1]);
}

SW_RNG_INLINE uint32_t temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// 2 * ((a >> 5) * 2^26 + (b >> 6)) / 2^53 - 1.  Both halves are below 2^27: the SIGNED 32-bit
// conversion is exact (and a single instruction at every vector width, unlike the unsigned one);
// the product, the sum (< 2^53) and the scaling by a power of two are exact, the final subtraction
// rounds once -- the same value NumPy's 2 * random_sample() - 1 has.
SW_RNG_INLINE double pm1(uint32_t wa, uint32_t wb)
{
    const double a = (double)(int32_t)(wa >> 5), b = (double)(int32_t)(wb >> 6);
    return (a * 67108864.0 + b) * (2.0 / 9007199254740992.0) - 1.0;
}

SW_RNG_INLINE int generate(uint32_t *key, int32_t *pos, int64_t n, double *out)
{
    int p = *pos;
    alignas(64) uint32_t buf[kN];   // tempered outputs of the current state block
    int64_t done = 0;
    uint32_t carry = 0;
    bool have_carry = false;   // first half (a) of a double split across two state blocks
    while (done < n) {
        if (p == kN) {
            regenerate(key);
            p = 0;
        }
        const int avail = kN - p;
#pragma clang loop vectorize(enable)
        for (int i = 0; i < avail; ++i) buf[i] = temper(key[p + i]);
        int i = 0;
        if (have_carry && avail > 0) {
            out[done++] = pm1(carry, buf[0]);
            have_carry = false;
            i = 1;
        }
        const int64_t pairs = (avail - i) / 2;
        const int64_t take = pairs < (n - done) ? pairs : (n - done);
        const uint32_t *src = buf + i;
        double *dst = out + done;
#pragma clang loop vectorize(enable)
        for (int64_t q = 0; q < take; ++q) dst[q] = pm1(src[2 * q], src[2 * q + 1]);
        done += take;
        i += (int)(2 * take);
        p += i;
        if (done < n && kN - p == 1) {
            // one output left in this block: it is the `a` half of the next double
            carry = buf[i];
            have_carry = true;
            p = kN;
        }
    }
    *pos = p;
    return SW_OK;
}

int generate_base(uint32_t *key, int32_t *pos, int64_t n, double *out) { return generate(key, pos, n, out); }

__attribute__((target("avx2,fma"))) int generate_avx2(uint32_t *key, int32_t *pos, int64_t n, double *out)
{
    return generate(key, pos, n, out);
}

__attribute__((target("avx512f,avx512dq,avx512vl,avx512bw,avx2,fma")))
int generate_avx512(uint32_t *key, int32_t *pos, int64_t n, double *out)
{
    return generate(key, pos, n, out);
}

typedef int (*generate_fn)(uint32_t *, int32_t *, int64_t, double *);

int g_forced_isa = -1;   // test hook (sw_mt19937_force_isa): -1 = widest available

generate_fn pick()
{
    __builtin_cpu_init();
    const bool a512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
                      __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw");
    const bool a2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    int want = g_forced_isa;
    if (want < 0) want = a512 ? 2 : (a2 ? 1 : 0);
    if (want == 2 && a512) return generate_avx512;
    if (want >= 1 && a2) return generate_avx2;
    return generate_base;
}

}  // namespace

extern "C" int sw_mt19937_uniform_pm1(uint32_t *key, int32_t *pos, int64_t n, double *out)
{
    if (!key || !pos || (!out && n > 0)) return SW_ERR_NULL;
    if (n < 0 || *pos < 0 || *pos > kN) return SW_ERR_SIZE;
    static const generate_fn fn = pick();
    return (g_forced_isa < 0 ? fn : pick())(key, pos, n, out);
}

// Test hook: 0 = baseline x86-64, 1 = AVX2, 2 = AVX-512, -1 = widest the CPU has (default).  A level the
// CPU lacks falls back to the next one down.  Returns the level that will actually run.
extern "C" int sw_mt19937_force_isa(int level)
{
    g_forced_isa = (level < -1 || level > 2) ? -1 : level;
    const generate_fn fn = pick();
    return fn == generate_avx512 ? 2 : (fn == generate_avx2 ? 1 : 0);
}
