// Gradient bucket planning: pure C++, no torch / HIP dependency, so the host
// sanitizer test (csrc/tests/host_sanitize_test.cpp, built with -fsanitize=address,undefined) can
// exercise exactly the code the reducer runs.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace pdt {

struct BucketPlan {
    std::vector<int64_t> params;   // parameter indices
    std::vector<int64_t> offsets;  // element offsets in the flat buffer
    int64_t numel = 0;             // padded size
};

constexpr int64_t kBucketAlign = 128;  // elements; every slot stays 256/512-B aligned for 16 B/lane access
inline int64_t bucket_pad(int64_t n) { return (n + kBucketAlign - 1) / kBucketAlign * kBucketAlign; }

// Walk `order` (gradient-ready order, last layer first) and cut a new bucket whenever the next
// parameter would push the current one over its cap; the first bucket has its own (small) cap so
// the first all-reduce starts early in backward.
inline std::vector<BucketPlan> plan_buckets(const std::vector<int64_t>& numels, const std::vector<int64_t>& order,
                                            int64_t elem_size, int64_t cap_bytes, int64_t first_cap_bytes) {
    std::vector<BucketPlan> out;
    BucketPlan cur;
    int64_t cap = first_cap_bytes > 0 ? first_cap_bytes : cap_bytes;
    for (int64_t idx : order) {
        const int64_t n = bucket_pad(numels[idx]);
        if (!cur.params.empty() && (cur.numel + n) * elem_size > cap) {
            out.push_back(std::move(cur));
            cur = BucketPlan();
            cap = cap_bytes;
        }
        cur.params.push_back(idx);
        cur.offsets.push_back(cur.numel);
        cur.numel += n;
    }
    if (!cur.params.empty()) out.push_back(std::move(cur));
    return out;
}

}  // namespace pdt
