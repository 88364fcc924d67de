// Where do the waves of small workgroups land?  N workgroups of T threads (T / 64 waves), each wave records its HW_ID
// (gfx9 layout: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]) and XCC_ID, then spins so that
// the whole grid is resident at once.  Prints how many waves share a SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wave_place.hip -o tools/micro/bin/wave_place && tools/micro/bin/wave_place [threads=128] [wgs=512] [vgprs=200] [lds_kb=27]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int VG>
__global__ void place_kernel(unsigned* out, long long spin) {
    extern __shared__ float smem[];
    if (VG >= 200) asm volatile("v_mov_b32 v199, 0" ::: "v199");
    else if (VG >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    smem[threadIdx.x] = 1.f;
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));         // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));       // HW_REG_XCC_ID
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 512, VG = argc > 3 ? atoi(argv[3]) : 200;
    const int ldskb = argc > 4 ? atoi(argv[4]) : 27;
    const int waves = N * (T / 64);
    unsigned* d;
    hipMalloc(&d, waves * 8);
    auto k = VG >= 200 ? place_kernel<200> : (VG >= 128 ? place_kernel<128> : place_kernel<64>);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(N), dim3(T), ldskb * 1024, 0, d, 40000LL);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * waves);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu;
    for (int w = 0; w < waves; ++w) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 15u;
        const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cukey]++;
        per_simd[(cukey << 2) | simd]++;
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    printf("%d workgroups x %d threads (%d waves), %d VGPRs, %d KB LDS: %zu CUs used, %zu SIMDs used\n", N, T, waves, VG, ldskb,
           per_cu.size(), per_simd.size());
    printf("  waves per CU:  ");
    for (auto& kv : hist_cu) printf(" %d CUs hold %d;", kv.second, kv.first);
    printf("\n  waves per SIMD:");
    for (auto& kv : hist_simd) printf(" %d SIMDs hold %d;", kv.second, kv.first);
    printf("\n  first workgroups: ");
    for (int w = 0; w < 8 && w < waves; ++w) printf("[xcc %u se %u cu %u simd %u] ", h[2 * w + 1] & 15u, (h[2 * w] >> 13) & 7u, (h[2 * w] >> 8) & 15u, (h[2 * w] >> 4) & 3u);
    printf("\n");
    return 0;
}
