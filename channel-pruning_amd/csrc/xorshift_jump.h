// Jump-ahead for sklearn's our_rand_r (sklearn/utils/_random.pxd:20-35), the xorshift32
// generator (13, 17, 5) that picks the coordinate of every coordinate-descent step.
//
// One xorshift step is linear over GF(2): s' = T s.  With the columns of T^64 known at
// compile time, 64 lanes that hold states s_l = T^l s_0 can all advance by 64 steps with one
// GF(2) matrix-vector product each, i.e. the stream is produced 64 values at a time by vector
// code (~1.7 VALU instructions per value) instead of ~17 scalar instructions per value.
// Usable from host code too (tests/ compile it with g++ to check it against the sequential
// generator).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CP_HD __host__ __device__
#else
#define CP_HD
#endif

namespace cpx {

CP_HD constexpr uint32_t xs_step(uint32_t s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

CP_HD constexpr uint32_t xs_pow64_col(int bit) {  // T^64 applied to the unit vector e_bit
    uint32_t s = uint32_t(1) << bit;
    for (int i = 0; i < 64; ++i) s = xs_step(s);
    return s;
}

// state <- T^64 state
CP_HD inline uint32_t xs_jump64(uint32_t s) {
    uint32_t acc = 0;
#define CPX_BIT(b)                                                  \
    {                                                               \
        constexpr uint32_t col = xs_pow64_col(b);                   \
        acc ^= (uint32_t)(-(int32_t)((s >> (b)) & 1u)) & col;       \
    }
    CPX_BIT(0) CPX_BIT(1) CPX_BIT(2) CPX_BIT(3) CPX_BIT(4) CPX_BIT(5) CPX_BIT(6) CPX_BIT(7)
    CPX_BIT(8) CPX_BIT(9) CPX_BIT(10) CPX_BIT(11) CPX_BIT(12) CPX_BIT(13) CPX_BIT(14) CPX_BIT(15)
    CPX_BIT(16) CPX_BIT(17) CPX_BIT(18) CPX_BIT(19) CPX_BIT(20) CPX_BIT(21) CPX_BIT(22) CPX_BIT(23)
    CPX_BIT(24) CPX_BIT(25) CPX_BIT(26) CPX_BIT(27) CPX_BIT(28) CPX_BIT(29) CPX_BIT(30) CPX_BIT(31)
#undef CPX_BIT
    return acc;
}

// rand_int(n) = (state & 0x7fffffff) % n  (_cd_fast.pyx:30-32) via Lemire's fastmod;
// magic = floor((2^64 - 1) / n) + 1, exact for every 32-bit numerator.
CP_HD inline uint64_t fastmod_magic(uint32_t n) { return ~uint64_t(0) / n + 1; }
CP_HD inline uint32_t fastmod(uint32_t r, uint64_t magic, uint32_t n) {
    const uint64_t low = magic * r;
#if defined(__HIP_DEVICE_COMPILE__)
    return uint32_t(__umul64hi(low, uint64_t(n)));
#else
    return uint32_t(((unsigned __int128)low * n) >> 64);
#endif
}

}  // namespace cpx
