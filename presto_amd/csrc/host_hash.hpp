// host_hash.hpp -- host-side copies of the reference's hash arithmetic, used where an operator has to
// emit a $hashvalue column for a handful of result rows.  Product code; the test oracle
// has its own independent restatement under oracle/.
#pragma once

#include <cstdint>
#include <cstring>

namespace pa {

inline uint64_t host_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// AbstractLongType.hash (core/trino-spi/src/main/java/io/trino/spi/type/AbstractLongType.java:126-130)
inline int64_t host_hash_bigint(int64_t v)
{
    return (int64_t)(host_rotl64((uint64_t)v * 0xC2B2AE3D27D4EB4FULL, 31) * 0x9E3779B185EBCA87ULL);
}

// XXH64, seed 0 (io.airlift.slice.XxHash64)
inline uint64_t host_xxh64(const uint8_t* p, int64_t len)
{
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL,
                   P5 = 0x27D4EB2F165667C5ULL;
    auto rd64 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; };
    auto rd32 = [](const uint8_t* q) { uint32_t v; memcpy(&v, q, 4); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { return host_rotl64(acc + in * P2, 31) * P1; };
    auto merge = [&](uint64_t h, uint64_t v) { return (h ^ round(0, v)) * P1 + P4; };
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0ULL - P1;
        do {
            v1 = round(v1, rd64(p));
            v2 = round(v2, rd64(p + 8));
            v3 = round(v3, rd64(p + 16));
            v4 = round(v4, rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = host_rotl64(v1, 1) + host_rotl64(v2, 7) + host_rotl64(v3, 12) + host_rotl64(v4, 18);
        h = merge(h, v1);
        h = merge(h, v2);
        h = merge(h, v3);
        h = merge(h, v4);
    }
    else {
        h = P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) {
        h ^= round(0, rd64(p));
        h = host_rotl64(h, 27) * P1 + P4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= (uint64_t)rd32(p) * P1;
        h = host_rotl64(h, 23) * P2 + P3;
        p += 4;
    }
    while (p < end) {
        h ^= (uint64_t)(*p) * P5;
        h = host_rotl64(h, 11) * P1;
        p++;
    }
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

inline uint64_t host_xxh64_long(int64_t v)
{
    uint8_t b[8];
    memcpy(b, &v, 8);
    return host_xxh64(b, 8);
}

}  // namespace pa
