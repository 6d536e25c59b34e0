// What does the library's radix sort (rocPRIM: onesweep / decoupled look-back) take for the grid build's
// (cell key, row) pairs?  20-bit keys of a nearly sorted sequence (a previous order perturbed like one step).
// Build: hipcc --offload-arch=gfx950 -O3 rocprim_sort.hip -o rocprim_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

int main() {
    for (size_t n : {262144ul, 630000ul, 4194304ul, 16777216ul}) {
        std::vector<uint32_t> k(n), v(n);
        srand(7);
        const double cells = 800000.0; // occupied key range of the random cloud
        for (size_t i = 0; i < n; ++i) k[i] = (uint32_t)(cells * i / n) + 100000u;
        for (size_t i = 0; i < n; ++i) { // a tenth of the rows move to a neighbouring cell (x +-1, y +-1: +-100)
            int r = rand() % 20;
            if (r == 0) k[i] += 1; else if (r == 1) k[i] -= 100;
            v[i] = (uint32_t)i;
        }
        uint32_t *dk, *dv, *ok, *ov;
        hipMalloc(&dk, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&ok, n * 4); hipMalloc(&ov, n * 4);
        hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
        size_t tmpBytes = 0; void *tmp = nullptr;
        rocprim::radix_sort_pairs(nullptr, tmpBytes, dk, ok, dv, ov, n, 0, 20);
        hipMalloc(&tmp, tmpBytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) rocprim::radix_sort_pairs(tmp, tmpBytes, dk, ok, dv, ov, n, 0, 20);
        hipDeviceSynchronize();
        const int reps = 20;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) rocprim::radix_sort_pairs(tmp, tmpBytes, dk, ok, dv, ov, n, 0, 20);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint32_t> out(n); hipMemcpy(out.data(), ok, n * 4, hipMemcpyDeviceToHost);
        printf("n = %zu: rocprim::radix_sort_pairs(20 bits) %.1f us per sort (temp %zu KB), sorted: %s\n", n, ms / reps * 1e3, tmpBytes / 1024,
               std::is_sorted(out.begin(), out.end()) ? "yes" : "NO");
        hipFree(dk); hipFree(dv); hipFree(ok); hipFree(ov); hipFree(tmp);
    }
    return 0;
}
