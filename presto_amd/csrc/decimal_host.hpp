// decimal_host.hpp -- host side of the DECIMAL aggregates: a group's sum arrives as independent i64 sums of 30-bit limbs
// (pa_dec_limb, kernels/pa_device.h) and is put together here, in 192-bit two's complement, into the value the reference's
// DecimalSumAggregation / DecimalAverageAggregation would emit
// (core/trino-main/src/main/java/io/trino/operator/aggregation/DecimalSumAggregation.java:177-190,
//  DecimalAverageAggregation.java:190-226; core/trino-spi/src/main/java/io/trino/spi/type/UnscaledDecimal128Arithmetic.java).
#pragma once

#include <cstdint>
#include <cstring>

namespace pa {

constexpr int kDecLimbBits = 30;  // PA_DEC_LIMB_BITS

// limbs a value of `bits` significant bits (sign included) is cut into
inline int decimal_limbs_for_bits(int bits) { return (bits + kDecLimbBits - 1) / kDecLimbBits; }
// bits of a DECIMAL(p, s) value: |v| < 10^p
inline int decimal_bits_for_precision(int precision)
{
    // ceil(p * log2(10)) + 1 sign bit
    return (int)((precision * 3402 + 1023) / 1024) + 1;
}

struct Wide192 {
    uint64_t w[3] = {0, 0, 0};
    // += v * 2^shift (v signed)
    void add_shifted(int64_t v, int shift)
    {
        uint64_t x[3];
        const uint64_t ext = v < 0 ? ~0ULL : 0ULL;
        // sign-extended 192-bit image of v, then shifted left
        uint64_t src[3] = {(uint64_t)v, ext, ext};
        const int limbs = shift / 64, bits = shift % 64;
        for (int i = 0; i < 3; i++) {
            const int j = i - limbs;
            uint64_t lo = j >= 0 ? src[j] : 0, below = (j - 1) >= 0 ? src[j - 1] : 0;
            x[i] = bits ? ((lo << bits) | (below >> (64 - bits))) : lo;
        }
        unsigned __int128 carry = 0;
        for (int i = 0; i < 3; i++) {
            const unsigned __int128 s = (unsigned __int128)w[i] + x[i] + carry;
            w[i] = (uint64_t)s;
            carry = s >> 64;
        }
    }
    bool negative() const { return (w[2] >> 63) != 0; }
    void negate()
    {
        uint64_t carry = 1;
        for (int i = 0; i < 3; i++) {
            const uint64_t v = ~w[i] + carry;
            carry = (carry && v == 0) ? 1 : 0;
            w[i] = v;
        }
    }
};

constexpr unsigned __int128 kTen38 = ((unsigned __int128)0x4B3B4CA85A86C47AULL << 64) | 0x098A224000000000ULL;

// the limb sums of one group -> the exact total
inline Wide192 decimal_total(const uint64_t* words, int first, int limbs)
{
    Wide192 t;
    for (int k = 0; k < limbs; k++) t.add_shifted((int64_t)words[first + k], kDecLimbBits * k);
    return t;
}
// total -> (negative, magnitude) when the magnitude is below `bound`; false = NUMERIC_VALUE_OUT_OF_RANGE
inline bool decimal_fits(Wide192 t, unsigned __int128 bound, bool* negative, unsigned __int128* magnitude)
{
    *negative = t.negative();
    if (*negative) t.negate();
    if (t.w[2] != 0) return false;
    *magnitude = ((unsigned __int128)t.w[1] << 64) | t.w[0];
    return *magnitude < bound;
}
// total / count, rounded half up on the magnitude (BigDecimal.divide(count, scale, ROUND_HALF_UP))
inline Wide192 decimal_average(Wide192 t, int64_t count)
{
    const bool neg = t.negative();
    if (neg) t.negate();
    const uint64_t d = (uint64_t)count;
    unsigned __int128 rem = 0;
    for (int i = 2; i >= 0; i--) {
        const unsigned __int128 cur = (rem << 64) | t.w[i];
        t.w[i] = (uint64_t)(cur / d);
        rem = cur % d;
    }
    if (rem * 2 >= d) {
        for (int i = 0; i < 3; i++) {
            if (++t.w[i] != 0) break;
        }
    }
    if (neg) t.negate();
    return t;
}
// LongDecimalType's 16 bytes: the low 64 bits of the magnitude, then the high 63 bits with the sign on top
inline void long_decimal_store(uint8_t* out, bool negative, unsigned __int128 magnitude)
{
    const uint64_t lo = (uint64_t)magnitude, hi = (uint64_t)(magnitude >> 64) | (negative && magnitude != 0 ? 0x8000000000000000ULL : 0ULL);
    memcpy(out, &lo, 8);
    memcpy(out + 8, &hi, 8);
}

}  // namespace pa
