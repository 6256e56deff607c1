// host_rng.cpp -- NumPy's legacy MT19937 uniform stream, generated natively on the host.
//
// The reference draws its exploration noise from NumPy's global legacy generator:
// np.random.seed(seed) (ars/ars_agent.py:95) and N calls of 2*np.random.rand(m, d)-1
// (ars/ars_agent.py:137-138).  Keeping that exact stream is part of the drop-in contract,
// but NumPy produces it at ~5 ns per double on one core: at 4096 directions (8 GPUs x 512)
// that is 0.33 ms per iteration on EVERY rank -- as long as the whole GPU iteration.  This
// file continues the same stream from the same state 5-8x faster (block-wise twist and
// tempering that the compiler vectorises), so the host stays ahead of the GPU at any rank
// count.  State goes in and out in the form np.random.get_state() / set_state() use
// (key[624], pos), so NumPy's generator can be advanced in lock step.
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

inline uint32_t twist(uint32_t u, uint32_t v, uint32_t far)
{
    const uint32_t y = (u & kUpper) | (v & kLower);
    return far ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrixA);
}

// One full regeneration of the 624-word state (numpy: mt19937_gen).  The two main loops
// have no loop-carried dependence shorter than 227 iterations, so they vectorise.
void regenerate(uint32_t *mt)
{
    int i = 0;
#pragma clang loop vectorize(assume_safety)
    for (i = 0; i < kN - kM; ++i) mt[i] = twist(mt[i], mt[i + 1], mt[i + kM]);
#pragma clang loop vectorize(assume_safety)
    for (i = kN - kM; i < kN - 1; ++i) mt[i] = twist(mt[i], mt[i + 1], mt[i + (kM - kN)]);
    mt[kN - 1] = twist(mt[kN - 1], mt[0], mt[kM - 1]);
}

inline uint32_t temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

}  // namespace

extern "C" int sw_mt19937_uniform_pm1(uint32_t *key, int32_t *pos, int64_t n, double *out)
{
    if (!key || !pos || (!out && n > 0)) return SW_ERR_NULL;
    if (n < 0 || *pos < 0 || *pos > kN) return SW_ERR_SIZE;
    int p = *pos;
    uint32_t buf[kN];   // tempered outputs of the current state block
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
            const uint32_t a = carry >> 5, b = buf[0] >> 6;
            out[done++] = 2.0 * ((a * 67108864.0 + b) / 9007199254740992.0) - 1.0;
            have_carry = false;
            i = 1;
        }
        const int64_t pairs = (avail - i) / 2;
        const int64_t take = pairs < (n - done) ? pairs : (n - done);
#pragma clang loop vectorize(enable)
        for (int64_t q = 0; q < take; ++q) {
            const uint32_t a = buf[i + 2 * q] >> 5, b = buf[i + 2 * q + 1] >> 6;
            out[done + q] = 2.0 * ((a * 67108864.0 + b) / 9007199254740992.0) - 1.0;
        }
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
