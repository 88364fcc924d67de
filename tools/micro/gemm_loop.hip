// Calibration of the wave-level GEMM loop of mlp_device.h (WaveGemm<1>::run, KC = 13 = a 100-wide layer) in isolation:
// ticks per call for 4 waves (one per SIMD) / 8 waves (two per SIMD) of a workgroup, one workgroup or 256.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -I<pkg>/csrc [-DEXP_NO_BLOAD ...] tools/micro/gemm_loop.hip -o tools/micro/bin/gemm_loop
#include "mlp_device.h"
#include <cstdio>

__global__ void k(const float* packed, float* out, long long* cyc, int KC, int layers, int LD) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8, half = lane >> 5;
    float* tile = smem + grp * 32 * LD;
    for (int i = tid & 255; i < 32 * LD; i += 256) tile[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    WaveGemm<1> wg;
    const float* pk = packed + (long)grp * 4 * KC * 256 * layers;
    NextFrags nx;
    nx.KCn = nx.KCnn = KC;
    nx.n0 = nx.nn0 = nx.n1 = nx.nn1 = frag_ptr(pk, 0, KC, wave, lane);
    wg.prime(frag_ptr(pk, 0, KC, wave, lane), frag_ptr(pk, 0, KC, wave, lane), KC, nx);
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    long long t0 = __builtin_readcyclecounter();
    for (int l = 0; l < layers; ++l) {
        const float4* p0 = frag_ptr(pk, (l % 4) * 4 * KC * 256, KC, wave, lane);
        nx.n0 = nx.nn0 = nx.n1 = nx.nn1 = frag_ptr(pk, ((l + 1) % 4) * 4 * KC * 256, KC, wave, lane);
        wg.run(tile + (lane & 31) * LD + half * 4, p0, p0, KC, nx, acc);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    const int KC = 13, layers = 64, LD = 132;
    float *packed, *out; long long* cyc;
    (void)hipMalloc(&packed, sizeof(float) * 2 * 4 * KC * 256 * layers);
    (void)hipMemset(packed, 0, sizeof(float) * 2 * 4 * KC * 256 * layers);
    (void)hipMalloc(&out, sizeof(float) * 512 * 256);
    (void)hipMalloc(&cyc, 8);
    for (int blocks : {1, 256})
        for (int threads : {256, 512}) {
            for (int rep = 0; rep < 2; ++rep) k<<<blocks, threads, 2 * 32 * LD * 4>>>(packed, out, cyc, KC, layers, LD);
            (void)hipDeviceSynchronize();
            long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("blocks %3d waves/SIMD %d: %.0f ticks per layer GEMM (52 MFMAs: floor 3328), %.1f per MFMA\n", blocks, threads / 256,
                   (double)c / layers, (double)c / layers / (4.0 * KC));
        }
    return 0;
}
