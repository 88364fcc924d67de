// Feasibility calibration for the register-resident ("RR") NODE kernels: one wave owns 16 rows and runs a whole MLP
// layer chain on v_mfma_f32_16x16x4_f32 with the activations never leaving its registers (the transposed product's
// output fragment IS the next layer's B fragment under a permuted k order); weights stream from an L2-resident
// fragment-ordered pack, D float4 per lane in flight.  Measures cycles per 100-wide layer (175 MFMAs = 5.6k cycles of
// matrix-pipe time) with every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rr_chain.hip -o tools/micro/bin/rr_chain && tools/micro/bin/rr_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ldw(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

template <int NB, int R, int D, int RELU_BITS>
__global__ __launch_bounds__(256) void chain_kernel(const float4* __restrict__ pack, int layers_f, int layers_g,
                                                    int n_stages, float* out, long long* cyc, unsigned* bits_out) {
    constexpr int KS = 4 * (NB - 1) + R, NM = NB * KS, NV = (NM + 3) / 4;
    static_assert(NV % D == 0, "queue depth must divide the layer's float4 count");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_layers = (wave < 2) ? layers_f : layers_g;
    const int net_base = (wave < 2 ? 0 : layers_f) * NV * 1024;       // bytes
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pack, 0, (layers_f + layers_g) * NV * 1024, 0x00020000);
    const int voff = lane * 16;
    float H[NB * 4];
#pragma unroll
    for (int i = 0; i < NB * 4; ++i) H[i] = 0.01f * (float)((lane + i) % 7);
    f32x4 wq[D];
#pragma unroll
    for (int i = 0; i < D; ++i) wq[i] = ldw(rs, voff, net_base + i * 1024);
    unsigned bits_acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int st = 0; st < n_stages; ++st) {
        for (int l = 0; l < n_layers; ++l) {
            const int cur = net_base + l * NV * 1024;
            const int nxt = net_base + ((l + 1 == n_layers) ? 0 : l + 1) * NV * 1024;
            f32x4 acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // MFMA order: block groups of 2 (the last of 3 when NB is odd), k inner
            int m = 0;
#pragma unroll
            for (int g0 = 0; g0 < NB; g0 += 2) {
                const int gn = (NB - g0 == 3) ? 3 : 2;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        if (jj < gn) {
                            const int v = m >> 2, c = m & 3;
                            const float a = wq[v % D][c];
                            acc[g0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, H[ks], acc[g0 + jj], 0, 0, 0);
                            if (c == 3 || m == NM - 1) {          // slot v%D is free: refill with stream element v + D
                                const int vn = v + D;
                                wq[v % D] = (vn < NV) ? ldw(rs, voff, cur + vn * 1024) : ldw(rs, voff, nxt + (vn - NV) * 1024);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            ++m;
                        }
                    }
                }
                if (gn == 3) g0 += 1;
            }
            unsigned word = 0;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = fmaxf(acc[j][r], 0.f);
                    H[j * 4 + r] = h * 0.05f;       // (keeps the values bounded over many layers)
                    if (RELU_BITS) word |= (h > 0.f ? 1u : 0u) << (j * 4 + r);
                }
            if (RELU_BITS) bits_acc ^= word;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NB * 4; ++i) s += H[i];
    out[(long)blockIdx.x * 256 + threadIdx.x] = s;
    if (RELU_BITS) bits_out[(long)blockIdx.x * 256 + threadIdx.x] = bits_acc;
    if (lane == 0) cyc[(long)blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NB, int R, int D, int RB>
void run(int blocks, int n_stages, const char* tag) {
    constexpr int KS = 4 * (NB - 1) + R, NM = NB * KS, NV = (NM + 3) / 4;
    const int lf = 3, lg = 2;
    const size_t nf4 = (size_t)(lf + lg) * NV * 64;
    std::vector<float> hw(nf4 * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.02f * (float)((int)(i * 2654435761u % 201) - 100) / 100.f;
    float4* pack; float* out; long long* cyc; unsigned* bits;
    hipMalloc(&pack, nf4 * 16); hipMemcpy(pack, hw.data(), nf4 * 16, hipMemcpyHostToDevice);
    hipMalloc(&out, sizeof(float) * 256 * blocks); hipMalloc(&cyc, 8 * 4 * blocks); hipMalloc(&bits, 4 * 256 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        chain_kernel<NB, R, D, RB><<<blocks, 256>>>(pack, lf, lg, n_stages, out, cyc, bits);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(4 * blocks);
    hipMemcpy(hc.data(), cyc, 8 * 4 * blocks, hipMemcpyDeviceToHost);
    double cf = 0, cg = 0;
    for (int b = 0; b < blocks; ++b) { cf += hc[4 * b] + hc[4 * b + 1]; cg += hc[4 * b + 2] + hc[4 * b + 3]; }
    cf /= 2.0 * blocks * n_stages * lf; cg /= 2.0 * blocks * n_stages * lg;
    const double flop = 2.0 * 16 * (double)(KS * 4) * (NB * 16) * (2 * lf + 2 * lg) * n_stages * blocks;   // padded
    printf("%-28s blocks %4d stages %d: %.1f us  cycles/layer f-wave %.0f g-wave %.0f (pipe floor %d)  %.1f TFLOP/s padded\n",
           tag, blocks, n_stages, ms * 1e3, cf, cg, NM * 32, flop / (ms * 1e-3) / 1e12);
    hipFree(pack); hipFree(out); hipFree(cyc); hipFree(bits);
}

int main() {
    run<7, 1, 11, 0>(256, 6, "hid100 D11");
    run<7, 1, 11, 1>(256, 6, "hid100 D11 +bits");
    run<7, 1, 4, 0>(256, 6, "hid100 D4");
    run<7, 1, 22, 0>(256, 6, "hid100 D22");
    run<7, 1, 11, 1>(512, 6, "hid100 D11 +bits 2 WG/CU");
    run<7, 1, 11, 1>(1024, 6, "hid100 D11 +bits 4 WG/CU");
    run<4, 4, 8, 1>(256, 6, "hid64 D8 +bits");
    run<8, 4, 16, 1>(256, 6, "hid128 D16 +bits");
    return 0;
}
