// Yardstick for the library's own radix sort: rocPRIM's device radix sort on the two shapes the
// binning stage sorts (1 M 32-bit depth keys with 32-bit values; 5.15 M 13-bit tile keys with 64-bit
// values).  Measurement only - libgsr.so does not link rocPRIM.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

template <typename V>
static void run(size_t n, int bits, const char *what)
{
    std::vector<uint32_t> hk(n);
    std::mt19937 rng(1);
    for (auto &k : hk) k = bits == 32 ? (0x3F000000u + (rng() & 0x7FFFFF)) : (rng() % 8160u);
    uint32_t *ki, *ko; V *vi, *vo;
    hipMalloc(&ki, n * 4); hipMalloc(&ko, n * 4); hipMalloc(&vi, n * sizeof(V)); hipMalloc(&vo, n * sizeof(V));
    hipMemcpy(ki, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(vi, 0, n * sizeof(V));
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, vi, vo, n, 0, bits);
    void *tmp; hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, 0, bits);
    hipEventRecord(a);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, 0, bits);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s: n=%zu bits=%d value=%zuB: %.4f ms per sort (temp %zu bytes)\n", what, n, bits, sizeof(V), ms / reps, tmp_bytes);
    hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo); hipFree(tmp);
}

int main()
{
    run<uint32_t>(1000000, 32, "depth sort");
    run<uint64_t>(5150000, 13, "tile sort");
    run<uint64_t>(31000000, 15, "tile sort 4K");
    return 0;
}
