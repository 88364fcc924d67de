// Calibration: cycles per v_mfma_f32_32x32x2_f32 for 1 / 2 / 4 independent accumulator chains and 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/bin/mfma_rate && tools/micro/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k(float* out, long long* cyc, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % CHAINS], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// the same loop on v_mfma_f32_16x16x4_f32 (2048 FLOP per instruction)
template <int CHAINS>
__global__ void k16(float* out, long long* cyc, int iters) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CHAINS], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS>
void run16(int threads, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    k16<CHAINS><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k16<CHAINS><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 8.0 * iters;
    printf("16x16x4: chains %d  threads %4d  blocks %4d: %.1f ticks per MFMA per wave, kernel %.1f us, %.1f TFLOP/s\n",
           CHAINS, threads, blocks, c / n, ms * 1e3, n * 2048.0 * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

template <int CHAINS>
void run(int threads, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 8.0 * iters;
    const double waves_per_simd = threads / 64 / 4.0;
    printf("chains %d  waves/SIMD %.2f  blocks %4d: %.1f ticks per MFMA per wave (%.1f per SIMD-MFMA), kernel %.1f us -> %.2f ticks/ns, %.1f TFLOP/s\n",
           CHAINS, waves_per_simd, blocks, c / n, c / n / (waves_per_simd < 1 ? 1 : waves_per_simd), ms * 1e3, c / (ms * 1e6),
           n * 4096.0 * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256}) {
        for (int threads : {256, 512, 1024}) {
            run<1>(threads, blocks);
            run<2>(threads, blocks);
            run<4>(threads, blocks);
        }
    }
    for (int blocks : {1, 256})
        for (int threads : {256, 512}) {
            run16<1>(threads, blocks);
            run16<2>(threads, blocks);
        }
    return 0;
}
