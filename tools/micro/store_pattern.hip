// Stand-alone calibration: how fast does the chip absorb the END-OF-KERNEL store burst of the register-resident MLP
// kernels, by store shape?  Every workgroup (256 threads) writes a 32-row x 512-float tile (64 KB), all workgroups at
// once, as
//   mode 0: the RR layout — lane (q, r16) writes 16 B of row r16 per instruction: 16 rows x 64 B per wave instruction
//   mode 1: whole rows — a wave instruction writes 1 KB contiguous (8 full 128-B lines)
//   mode 2: the RR layout with the two 64-B halves of a line written by two consecutive instructions of the same wave
//           (what the kernels do today; mode 0 strides the halves far apart)
// hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o tools/micro/bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* out, int spin) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15, rh = wave >> 1, ch = wave & 1;
    float* tile = out + (long)blockIdx.x * 32 * 512;
    f32x4 v{(float)tid, 1.f, 2.f, 3.f};
    // some ALU time first so that every workgroup reaches its stores at about the same moment, as the real kernels do
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (MODE == 1) {
        // wave w writes rows 8w .. 8w+7, 2 KB each: 16 instructions of 1 KB
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = 8 * wave + (k >> 1);
            *reinterpret_cast<f32x4*>(tile + row * 512 + (k & 1) * 256 + lane * 4) = v;
        }
    } else {
        // wave (rh, ch): rows 16 rh + r16, panel ch of both 256-float layers: blocks j = 0..7 of 16 floats, lane quarter q
        float* row = tile + (16 * rh + r16) * 512 + 128 * ch + 4 * q;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int layer = (MODE == 2) ? (k >> 3) : (k & 1), j = (MODE == 2) ? (k & 7) : (k >> 1);
            *reinterpret_cast<f32x4*>(row + layer * 256 + 16 * j) = v;
        }
    }
}

int main() {
    const int n_wg[] = {128, 256, 384, 768};
    float* buf;
    hipMalloc(&buf, 768L * 32 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 2000}) {
        for (int g : n_wg) {
            float t[3];
            for (int mode = 0; mode < 3; ++mode) {
                std::vector<float> ts;
                for (int it = 0; it < 30; ++it) {
                    hipEventRecord(e0);
                    if (mode == 0) store_kernel<0><<<g, 256>>>(buf, spin);
                    else if (mode == 1) store_kernel<1><<<g, 256>>>(buf, spin);
                    else store_kernel<2><<<g, 256>>>(buf, spin);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    ts.push_back(ms * 1e3f);
                }
                std::sort(ts.begin(), ts.end());
                t[mode] = ts[ts.size() / 2];
            }
            printf("spin %4d  %3d workgroups (%5.1f MB): RR far halves %6.1f us   whole rows %6.1f us   RR adjacent halves %6.1f us\n",
                   spin, g, g * 65536 / 1e6, t[0], t[1], t[2]);
        }
    }
    return 0;
}
