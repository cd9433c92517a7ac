// Host check (g++) of channel-pruning_amd/csrc/xorshift_jump.h against the sequential
// generator restated from sklearn/utils/_random.pxd:20-35.  Exit code 0 = all good.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "xorshift_jump.h"

static uint32_t seq_rand_int(uint32_t *s, uint32_t n) {
    if (*s == 0) *s = 1;
    *s ^= *s << 13; *s ^= *s >> 17; *s ^= *s << 5;
    return (*s % 2147483648u) % n;
}

int main() {
    const uint32_t seeds[] = {1u, 2u, 12345u, 2147483646u, 0u, 0xdeadbeefu, 987654321u};
    const uint32_t ns[] = {1u, 3u, 16u, 55u, 64u, 96u, 222u, 256u, 445u, 512u, 2048u, 65537u};
    long checked = 0;
    for (uint32_t seed : seeds)
        for (uint32_t n : ns) {
            uint32_t s = seed;
            // lane l state = T^(l+1) seed  (value consumed by step l of the first batch)
            uint32_t lane[64];
            uint32_t s0 = seed == 0 ? 1u : seed;
            for (int l = 0; l < 64; ++l) { s0 = cpx::xs_step(s0); lane[l] = s0; }
            const uint64_t magic = cpx::fastmod_magic(n);
            for (int batch = 0; batch < 50; ++batch) {
                for (int l = 0; l < 64; ++l) {
                    uint32_t want = seq_rand_int(&s, n);
                    uint32_t got = cpx::fastmod(lane[l] & 0x7fffffffu, magic, n);
                    if (want != got) {
                        printf("MISMATCH seed=%u n=%u batch=%d lane=%d want=%u got=%u\n", seed, n, batch, l, want, got);
                        return 1;
                    }
                    ++checked;
                }
                for (int l = 0; l < 64; ++l) lane[l] = cpx::xs_jump64(lane[l]);
            }
        }
    printf("ok %ld\n", checked);
    return 0;
}
