// Host-side invariants of the native runtime, meant to be built with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// (tests/test_native_host.py does that; GPU sanitizers are not available on the target pool, host ones are).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "../kernels/fastdiv.h"
#include "../reducer/bucket_plan.h"

#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

static void test_bucket_plans() {
    std::mt19937_64 rng(1234);
    for (int trial = 0; trial < 300; ++trial) {
        const int n = 1 + (int)(rng() % 200);
        std::vector<int64_t> numels(n), order;
        for (auto& v : numels) v = 1 + (int64_t)(rng() % (trial % 3 == 0 ? 3000000 : 5000));
        for (int i = n - 1; i >= 0; --i) order.push_back(i);
        if (trial % 2) std::shuffle(order.begin(), order.end(), rng);
        const int64_t elem = trial % 4 == 0 ? 2 : 4;
        const int64_t cap = 1 + (int64_t)(rng() % (16 << 20)), first = (int64_t)(rng() % (1 << 20));
        const auto plans = pdt::plan_buckets(numels, order, elem, cap, first);
        std::set<int64_t> seen;
        size_t pos = 0;
        for (size_t b = 0; b < plans.size(); ++b) {
            const auto& p = plans[b];
            CHECK(!p.params.empty() && p.params.size() == p.offsets.size());
            int64_t off = 0;
            for (size_t j = 0; j < p.params.size(); ++j) {
                CHECK(p.params[j] == order[pos++]);                    // order preserved across buckets
                CHECK(seen.insert(p.params[j]).second);                // every parameter exactly once
                CHECK(p.offsets[j] == off && off % pdt::kBucketAlign == 0);
                off += pdt::bucket_pad(numels[p.params[j]]);
            }
            CHECK(p.numel == off);
            const int64_t limit = (b == 0 && first > 0) ? first : cap;
            CHECK(p.params.size() == 1 || p.numel * elem <= limit);    // only a single oversize tensor may exceed
        }
        CHECK((int)seen.size() == n);
    }
}

static void test_fastdiv() {
    std::mt19937 rng(7);
    const uint32_t ds[] = {1, 2, 3, 5, 7, 14, 28, 49, 56, 57, 112, 115, 196, 224, 3136, 12544, 50176, 65535, 1u << 20, 2147483647u};
    for (uint32_t d : ds) {
        const pdt::FastDiv f = pdt::make_fastdiv(d);
        for (int i = 0; i < 20000; ++i) {
            const uint32_t n = i < 2000 ? (uint32_t)i : (rng() & 0x7fffffffu);
            CHECK(pdt::fastdiv_host(n, f) == n / d);
        }
        CHECK(pdt::fastdiv_host(0x7fffffffu, f) == 0x7fffffffu / d);
    }
}

int main() {
    test_bucket_plans();
    test_fastdiv();
    std::puts("host sanitize test: OK");
    return 0;
}
