// adcensus_farm_multi.cpp -- the C++ host of BASELINE.json configs[4]: a batch of independent stereo pairs farmed over every
// visible GPU of one node.  One adc_farm (3 pipelines with pinned staging, include/adcensus_c_api.h) per device, one host
// thread per device; the threads pull pair indices from ONE atomic counter (a faster GPU takes more pairs) and the join of
// the threads is the completion barrier -- pairs are independent, so there is no collective.
//   adcensus_farm_multi [pairs=64] [width=1920] [height=1080] [max_disparity=128] [gpus=all]
// Inputs are seeded noise pairs made on the host (xorshift; no dataset on the box).  Prints pairs/s of the whole batch
// (PCIe inclusive: pageable host images in, pageable host maps out), the pairs every GPU took and a checksum per pair;
// with more than one GPU every pair is ALSO computed on the next GPU afterwards and the two maps must be identical.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "adcensus_c_api.h"

static void make_pair(int id, size_t bytes, std::vector<uint8_t>& l, std::vector<uint8_t>& r)
{
    uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(id + 1);
    auto next = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint8_t)(s >> 32); };
    l.resize(bytes); r.resize(bytes);
    for (size_t i = 0; i < bytes; i++) l[i] = next();
    for (size_t i = 0; i < bytes; i++) r[i] = next();
}
static uint64_t fnv(const float* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    const uint8_t* b = reinterpret_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n * 4; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    const int pairs = argc > 1 ? atoi(argv[1]) : 64, W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
    const int D = argc > 4 ? atoi(argv[4]) : 128;
    int gpus = adc_device_count();
    if (argc > 5 && atoi(argv[5]) > 0 && atoi(argv[5]) < gpus) gpus = atoi(argv[5]);
    if (gpus < 1) { fprintf(stderr, "no HIP device (there is no CPU fallback)\n"); return 2; }
    adc_option opt;
    adc_option_default(&opt);
    opt.max_disparity = D;
    const size_t P = (size_t)W * H;
    std::vector<std::vector<uint8_t>> L(pairs), R(pairs);
    for (int i = 0; i < pairs; i++) make_pair(i, P * 3, L[i], R[i]);
    std::vector<std::vector<float>> out(pairs, std::vector<float>(P)), again(pairs);
    std::vector<int> owner(pairs, -1), failed(gpus, 0);
    std::vector<adc_farm*> farms(gpus, nullptr);
    for (int g = 0; g < gpus; g++)
        if (!(farms[g] = adc_farm_create(W, H, &opt, g, 3))) { fprintf(stderr, "GPU %d: %s\n", g, adc_last_error()); return 2; }
    std::atomic<int> next(0);
    auto worker = [&](int g, std::vector<std::vector<float>>* dst, bool shifted) {
        for (;;) { // pull queue; `shifted`: the verification leg, GPU g recomputes what GPU g - 1 delivered
            const int i = next.fetch_add(1);
            if (i >= pairs) break;
            if (shifted && owner[i] != (g + gpus - 1) % gpus) continue;
            if (!shifted) owner[i] = g;
            (*dst)[i].resize(P);
            uint64_t ticket = 0;
            const int rc = adc_farm_submit(farms[g], L[i].data(), R[i].data(), (*dst)[i].data(), &ticket);
            if (rc != 0 && rc != ADC_FARM_PREVIOUS_FAILED) { failed[g]++; break; }
            if (rc == ADC_FARM_PREVIOUS_FAILED) failed[g]++;
        }
        if (adc_farm_drain(farms[g]) < 0) failed[g]++;
    };
    auto run = [&](std::vector<std::vector<float>>* dst, bool shifted) {
        next = 0;
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int g = 0; g < gpus; g++) th.emplace_back(worker, g, dst, shifted);
        for (auto& t : th) t.join(); // the completion barrier
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    run(&out, false); // warm-up (clocks, first-Match ring / budget assumptions)
    const double dt = run(&out, false);
    std::vector<int> took(gpus, 0);
    for (int i = 0; i < pairs; i++) took[owner[i]]++;
    printf("{\"pairs\": %d, \"size\": [%d, %d, %d], \"gpus\": %d, \"pairs_per_s\": %.2f, \"pairs_per_gpu\": [", pairs, W, H, D, gpus, pairs / dt);
    for (int g = 0; g < gpus; g++) printf("%s%d", g ? ", " : "", took[g]);
    int mism = 0, nfail = 0;
    if (gpus > 1) { // (every worker scans all indices and keeps the previous GPU's pairs)
        for (int g = 0; g < gpus; g++) { next = 0; std::thread t(worker, g, &again, true); t.join(); }
        for (int i = 0; i < pairs; i++) mism += again[i].size() != P || memcmp(again[i].data(), out[i].data(), P * 4) != 0;
    }
    for (int g = 0; g < gpus; g++) nfail += failed[g];
    printf("], \"cross_gpu_mismatches\": %d, \"failures\": %d, \"checksum_pair0\": \"%016llx\"}\n", mism, nfail, (unsigned long long)fnv(out[0].data(), P));
    for (adc_farm* f : farms) adc_farm_destroy(f);
    return (mism || nfail) ? 1 : 0;
}
