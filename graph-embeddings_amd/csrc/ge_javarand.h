// ge_javarand.h -- java.util.Random's 48-bit LCG for host and device code.
// Public JDK algorithm; the reference draws from it through ExtendedRandom
// (J/util/rnd/ExtendedRandom.java:27-35) for parameter init (J/opt/Optimizer.java:50-57)
// and for the Fisher-Yates shuffle (ExtendedRandom.java:398-407).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define GE_HD __host__ __device__
#else
#define GE_HD
#endif

namespace ge {

struct JavaRandom {
    static constexpr uint64_t MULT = 0x5DEECE66DULL;
    static constexpr uint64_t ADD  = 0xBULL;
    static constexpr uint64_t MASK = (1ULL << 48) - 1;
    uint64_t s;

    GE_HD static uint64_t scramble(int64_t seed) { return ((uint64_t)seed ^ MULT) & MASK; }
    GE_HD int32_t next(int bits) {
        s = (s * MULT + ADD) & MASK;
        return (int32_t)(uint32_t)(s >> (48 - bits));
    }
    GE_HD float next_float() { return (float)next(24) / (float)(1 << 24); }
    GE_HD int32_t next_int(int32_t bound) {
        int32_t r = next(31);
        int32_t m = bound - 1;
        if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
        for (int32_t u = r;; u = next(31)) {
            r = u % bound;
            if ((int32_t)((uint32_t)u - (uint32_t)r + (uint32_t)m) >= 0) return r;
        }
    }
    // state after n further next() calls: the LCG is an affine map, compose by squaring.
    GE_HD static uint64_t jump(uint64_t state, uint64_t n) {
        uint64_t ra = 1, rc = 0, ba = MULT, bc = ADD;
        while (n) {
            if (n & 1) { ra = (ba * ra) & MASK; rc = (ba * rc + bc) & MASK; }
            bc = (ba * bc + bc) & MASK;
            ba = (ba * ba) & MASK;
            n >>= 1;
        }
        return (ra * state + rc) & MASK;
    }
};

}  // namespace ge
