// vmx_device.h: FastDiv — the kernels' division of a path / pixel index by a launch constant — against n / d on the host:
// every d of a list of awkward divisors x numerators around every multiple boundary and across the 32-bit range.
#include <cstdint>
#include <cstdio>
#include <random>

#include "vmx_device.h"

static inline uint32_t fast_div(uint32_t n, const vmx::FastDiv &f) {  // the device function, with the host's mulhi
    const uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

int main() {
    const uint32_t ds[] = {1, 2, 3, 4, 5, 7, 16, 17, 20, 63, 64, 65, 255, 256, 257, 1000, 1024, 1080, 1920, 2160, 3840, 65535,
                           65536, 65537, 1000003, 0x7FFFFFFFu, 0x80000000u, 0x80000001u, 0xFFFFFFFEu, 0xFFFFFFFFu};
    std::mt19937 rng(7);
    unsigned long long bad = 0, n_checked = 0;
    for (uint32_t d : ds) {
        const vmx::FastDiv f = vmx::make_fastdiv(d);
        auto check = [&](uint32_t n) {
            ++n_checked;
            if (fast_div(n, f) != n / d) ++bad;
        };
        for (uint32_t n = 0; n < 100000; ++n) check(n), check(0xFFFFFFFFu - n);
        for (uint64_t q = 0; q * d <= 0xFFFFFFFFull && q < 200000; ++q) {
            const uint64_t b = q * d;
            for (int o = -2; o <= 2; ++o)
                if ((int64_t)b + o >= 0 && b + o <= 0xFFFFFFFFull) check((uint32_t)(b + o));
        }
        for (int i = 0; i < 2000000; ++i) check(rng());
    }
    for (int i = 0; i < 20000; ++i) {  // random divisors too
        const uint32_t d = rng() >> (rng() % 32);
        const vmx::FastDiv f = vmx::make_fastdiv(d ? d : 1);
        for (int k = 0; k < 200; ++k) {
            const uint32_t n = rng() >> (rng() % 32);
            ++n_checked;
            if (fast_div(n, f) != n / (d ? d : 1)) ++bad;
        }
    }
    std::printf("checked %llu mismatches %llu\n", n_checked, bad);
    return bad != 0;
}
