// Can the read-back go through an SDMA engine instead of the runtime's blit kernel (which slows the
// kernels beside it)?  50.3 MB device -> pinned host through hsa_amd_memory_async_copy
// (the HSA runtime decides) and through each SDMA engine it reports, timed on the host.
// build: hipcc --offload-arch=gfx950 -O2 sdma_d2h.cpp -o sdma_d2h -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <vector>

static hsa_agent_t g_gpu{}, g_cpu{};
static bool have_gpu = false, have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void *) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}

int main() {
    const size_t bytes = (size_t)4194304 * 12;
    void *dev = nullptr, *host = nullptr;
    if (hipMalloc(&dev, bytes) != hipSuccess || hipHostMalloc(&host, bytes, hipHostMallocDefault) != hipSuccess) return 1;
    hipMemset(dev, 1, bytes);
    hipDeviceSynchronize();
    if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 1; }
    hsa_iterate_agents(on_agent, nullptr);
    if (!have_gpu || !have_cpu) { printf("agents not found\n"); return 1; }
    hsa_signal_t sig;
    hsa_signal_create(1, 0, nullptr, &sig);
    auto time_copy = [&](int engine, const char *label) {
        double best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hsa_signal_store_relaxed(sig, 1);
            auto t0 = std::chrono::steady_clock::now();
            hsa_status_t st = engine < 0
                ? hsa_amd_memory_async_copy(host, g_cpu, dev, g_gpu, bytes, 0, nullptr, sig)
                : hsa_amd_memory_async_copy_on_engine(host, g_cpu, dev, g_gpu, bytes, 0, nullptr, sig,
                                                      (hsa_amd_sdma_engine_id_t)engine, false);
            if (st != HSA_STATUS_SUCCESS) { printf("%s: copy call failed (status %d)\n", label, (int)st); return; }
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%s: %.3f ms = %.1f GB/s\n", label, best, bytes / best / 1e6);
    };
    time_copy(-1, "hsa_amd_memory_async_copy (runtime's choice)");
    uint32_t mask = 0;
    hsa_status_t st = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &mask);
    printf("copy_engine_status: status %d, free engine mask 0x%x\n", (int)st, mask);
    for (int e = 0; e < 8; ++e)
        if (mask & (1u << e)) {
            char label[64];
            snprintf(label, sizeof label, "SDMA engine bit %d", e);
            time_copy(1 << e, label);
        }
    // the HIP runtime's own copy, for scale
    hipStream_t s;
    hipStreamCreate(&s);
    for (int rep = 0; rep < 3; ++rep) hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 10; ++rep) hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 10;
    printf("hipMemcpyAsync: %.3f ms = %.1f GB/s\n", ms, bytes / ms / 1e6);
    return 0;
}
