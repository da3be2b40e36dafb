// Division by a launch constant via multiply-high (exact for n < 2^31): the implicit-GEMM index math
// (pixel -> image / row / column) runs on the VALU next to the MFMAs and must not cost a v_div sequence.
// Pure C++ (the device-side fdiv lives in common.h) so the host sanitizer test can check the magic numbers.
#pragma once
#include <stdint.h>

namespace pdt {

struct FastDiv {
    uint32_t mlo, shift, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t L = 0;
    while ((1u << L) < d) ++L;
    const unsigned __int128 one = 1;
    const unsigned __int128 magic = ((one << (32 + L)) + d - 1) / d;  // ceil(2^(32+L) / d), in [2^32, 2^33)
    f.mlo = (uint32_t)(magic - (one << 32));
    f.shift = L;
    return f;
}
// what the device computes: (n + mulhi(n, mlo)) >> shift
inline uint32_t fastdiv_host(uint32_t n, const FastDiv& f) {
    const uint32_t hi = (uint32_t)(((uint64_t)n * f.mlo) >> 32);
    return (n + hi) >> f.shift;
}

}  // namespace pdt
