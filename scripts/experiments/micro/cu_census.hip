// Which blocks of a launch share a CU?  A census kernel with the F(4,3) gate kernel's footprint (256 threads, 53 760 bytes of
// LDS: 3 blocks per CU): every block records its XCC id, HW id (SE / SH / CU) and start / end time and spins for ~20 us.
// Prints, for the first 768 block ids (the first resident round), how the three blocks of a CU are numbered.
//   hipcc --offload-arch=gfx950 -O3 -o cu_census cu_census.hip && ./cu_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };
__global__ __launch_bounds__(256, 3) void census(Rec *out, int spin_ticks) {
    __shared__ float lds[53760 / 4];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        Rec r;
        r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        r.hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        r.t0 = t0;
        r.t1 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x] = r;
    }
    if (lds[(threadIdx.x * 7) & 255] == -1.f) out[0].xcc = 0;
}
int main() {
    const int blocks = 10080;
    Rec *d; hipMalloc(&d, blocks * sizeof(Rec));
    std::vector<Rec> h(blocks);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(census, dim3(blocks), dim3(256), 0, 0, d, 2000);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (auto &r : h) tmin = std::min(tmin, r.t0);
    std::map<unsigned, std::vector<int>> by_cu;      // (xcc, se, sh, cu) -> block ids
    for (int i = 0; i < blocks; ++i) {
        const unsigned cu = (h[i].hwid >> 8) & 15, sh = (h[i].hwid >> 12) & 1, se = (h[i].hwid >> 13) & 7;
        by_cu[((h[i].xcc & 15) << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
    }
    printf("distinct (xcc, se, sh, cu): %zu\n", by_cu.size());
    int shown = 0;
    for (auto &kv : by_cu) {
        if (shown++ >= 12) break;
        printf("xcc %u se %u sh %u cu %2u: first blocks", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
        for (size_t k = 0; k < std::min<size_t>(kv.second.size(), 8); ++k) printf(" %5d(t0 %5.1f us)", kv.second[k], (h[kv.second[k]].t0 - tmin) / 100.0);
        printf("  ... %zu blocks\n", kv.second.size());
    }
    // hypothesis A: the three first-round blocks of a CU are ids with equal (id & 7) and (id >> 3) % 32 ; B: (id >> 3) / 3 equal
    int okA = 0, okB = 0, cus = 0;
    for (auto &kv : by_cu) {
        std::vector<int> first;
        for (int id : kv.second) if (id < 768) first.push_back(id);
        if (first.size() != 3) continue;
        ++cus;
        okA += ((first[0] >> 3) % 32 == (first[1] >> 3) % 32 && (first[1] >> 3) % 32 == (first[2] >> 3) % 32);
        okB += ((first[0] >> 3) / 3 == (first[1] >> 3) / 3 && (first[1] >> 3) / 3 == (first[2] >> 3) / 3);
    }
    printf("CUs with exactly 3 first-round blocks: %d; slot = (id>>3)/32 holds on %d, slot = (id>>3)%%3 holds on %d\n", cus, okA, okB);
    // start-time spread of the first round and of the whole launch
    unsigned long long t768 = 0;
    for (int i = 0; i < 768; ++i) t768 = std::max(t768, h[i].t0);
    printf("first 768 blocks start within %.2f us\n", (t768 - tmin) / 100.0);
    return 0;
}
