// Register-resident ("RR") layer chains on v_mfma_f32_16x16x4_f32: ONE wave owns 16 rows and every hidden unit of
// a net, and the activations never leave its registers between layers.
//
// The product is taken transposed, D[unit][row] = sum_k W[unit][k] X[row][k]: the weight fragment is the MFMA's A
// operand (lane l supplies A[l & 15][l >> 4]), the activations its B operand (lane l supplies B[l >> 4][l & 15]), and
// the result leaves lane l = (q, row) = (l >> 4, l & 15) with D[4 q + r][row] in register r.  So after one 16-unit output
// block j a lane holds, for ITS row, the units 16 j + 4 q + r (r = 0..3) — and a k-step of the NEXT layer needs from
// lane (q, row) exactly one k value of that row.  With the k order permuted so that k-step (j, r) contracts the four
// units {16 j + 4 q + r : q = 0..3}, register r of block j IS the B operand of k-step 4 j + r: bias + ReLU are applied in
// place and the next layer's MFMAs read the same registers.  No LDS round trip, no barrier, no cross-lane move.
//
// Widths that are not multiples of 16 (the reference's NODE nets are 100 wide, U/sac_cbf_clf/model.py:186-206): the
// last block holds 4 R units (R = 1..4; 100 = 6 * 16 + 4 * 1), placed at A rows 4 q + r with r < R, i.e. unit
// 16 (NB-1) + R q + r — so that block contributes R k-steps instead of 4 and a 100-deep contraction is exactly 25
// k-steps (the 32x32x2 tiling of mlp_device.h issues 104 / 128).  Rows r >= R carry zero weights.
//
// Weights stream from an L2-resident fragment-ordered copy ("RR pack", written by nlbac_mlp_pack next to the 32x32x2
// packs): one buffer_load_dwordx4 per lane = the A fragments of four consecutive MFMAs, D loads in flight per wave in
// statically indexed registers; the stream runs on across layers and stages (the slots freed by a layer's last MFMAs
// are refilled with the next layer's first fragments).  tools/micro/rr_chain.hip (MI355X, every CU busy, one wave per
// SIMD): a 100-wide layer's 175 MFMAs (5.6k cycles of matrix-pipe time) take 6.2k cycles including bias / ReLU.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define RR_MAX_HID 128

__host__ __device__ constexpr int rr_pick_depth(int nv) {
    for (int d = 12; d >= 4; --d)
        if (nv % d == 0) return d;
    return 1;
}

template <int NB_, int R_>
struct RRShape {
    static constexpr int NB = NB_, R = R_;
    static constexpr int KS = 4 * (NB - 1) + R;     // k-steps of a hid-deep contraction (= hid / 4)
    static constexpr int HID = 4 * KS;
    static constexpr int NM = NB * KS;              // MFMAs of a hid x hid layer on 16 rows
    static constexpr int NV = (NM + 3) / 4;         // float4 per lane of its fragment stream
    static constexpr int D = rr_pick_depth(NV);     // of them in flight
    static constexpr int LAYER_BYTES = NV * 1024;
    static_assert(NB >= 2 && NB <= 8 && R >= 1 && R <= 4, "RR chains cover widths 20..128");
    static_assert(D >= 4, "no usable queue depth for this width");
};

// ---- unit <-> fragment position (shared by the pack kernel and the kernels) ------------------------------------------
// the unit that lane quarter kq contributes at k-step ks
__host__ __device__ __forceinline__ int rr_unit_in(int NB, int R, int ks, int kq) {
    const int full = 4 * (NB - 1);
    return ks < full ? 16 * (ks >> 2) + 4 * kq + (ks & 3) : 16 * (NB - 1) + R * kq + (ks - full);
}
// the unit that A row hu of output block jo computes (-1: a padding row)
__host__ __device__ __forceinline__ int rr_unit_out(int NB, int R, int jo, int hu) {
    if (jo < NB - 1) return 16 * jo + hu;
    const int q = hu >> 2, r = hu & 3;
    return r < R ? 16 * (NB - 1) + R * q + r : -1;
}
// MFMA issue order of a layer: output blocks in groups of two (the last group has three when NB is odd), k-steps inner,
// so two / three independent accumulator chains alternate (v_mfma_f32_16x16x4_f32: 32 cycles issue, 40 dependent)
__host__ __device__ __forceinline__ void rr_mfma_of(int NB, int KS, int m, int& jo, int& ks) {
    int g0 = 0;
    for (;;) {
        const int left = NB - g0, gn = (left == 3) ? 3 : (left < 2 ? left : 2);
        if (m < gn * KS || left <= gn) { ks = m / gn; jo = g0 + m % gn; return; }
        m -= gn * KS;
        g0 += gn;
    }
}
__host__ __device__ __forceinline__ bool rr_width_ok(int hid) { return hid % 4 == 0 && hid >= 20 && hid <= RR_MAX_HID; }
// floats of one layer's RR pack
__host__ __device__ __forceinline__ long rr_layer_floats(int hid) {
    const int KS = hid / 4, NB = (hid + 15) / 16;
    return (long)((NB * KS + 3) / 4) * 256;
}

__device__ __forceinline__ f32x4 rr_ldw(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rr_rsrc(const float* base, int floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, floats * 4, 0x00020000);
}

// ReLU as an integer max: negative floats (and -0) are negative integers; one VALU op, no NaN canonicalisation
__device__ __forceinline__ float rr_relu(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave's hid x hid layer on its 16 rows:  acc[jo] (+)= sum_ks A(jo, ks) * H[ks]
// ---------------------------------------------------------------------------------------------------------------------
template <class S>
struct RRGemm {
    f32x4 wq[S::D];

    // first D float4 of the stream that starts at byte offset `soff` (wave-uniform) of the pack
    __device__ __forceinline__ void prime(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
#pragma unroll
        for (int i = 0; i < S::D; ++i) wq[i] = rr_ldw(rs, voff, soff + i * 1024);
    }

    // cur / nxt: byte offsets of this layer's fragments and of the layer that follows it in the wave's stream
    __device__ __forceinline__ void run(f32x4 (&acc)[S::NB], const float (&H)[S::KS], __amdgpu_buffer_rsrc_t rs, int voff,
                                        int cur, int nxt) {
        constexpr int NB = S::NB, KS = S::KS, NM = S::NM, NV = S::NV, D = S::D;
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int m = 0;
#pragma unroll
        for (int g0 = 0; g0 < NB;) {
            constexpr int dummy = 0; (void)dummy;
            const int left = NB - g0, gn = (left == 3) ? 3 : (left < 2 ? left : 2);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    if (jj < gn) {
                        const int v = m >> 2, c = m & 3;
                        acc[g0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[v % D][c], H[ks], acc[g0 + jj], 0, 0, 0);
                        if (c == 3 || m == NM - 1) {          // slot v % D is free: refill with stream element v + D
                            const int vn = v + D;
                            wq[v % D] = (vn < NV) ? rr_ldw(rs, voff, cur + vn * 1024) : rr_ldw(rs, voff, nxt + (vn - NV) * 1024);
                            __builtin_amdgcn_sched_barrier(0);   // keep each refill right behind its slot's MFMAs
                        }
                        ++m;
                    }
                }
            }
            g0 += gn;
        }
    }
};

// this lane's 4 bias values of output block jo (zeros for padding rows)
template <class S>
__device__ __forceinline__ f32x4 rr_bias(const float* __restrict__ b, int jo, int q) {
    if (jo < S::NB - 1 || S::R == 4) return *reinterpret_cast<const f32x4*>(b + 16 * jo + 4 * q);
    f32x4 v{0.f, 0.f, 0.f, 0.f};
    const float* p = b + 16 * (S::NB - 1) + S::R * q;
#pragma unroll
    for (int r = 0; r < S::R; ++r) v[r] = p[r];
    return v;
}

// a lane's units of block jo <-> a row of a row-major [rows][hid] array (activations, dz): 4 (or R) consecutive floats
template <class S>
__device__ __forceinline__ f32x4 rr_row_load(const float* __restrict__ rowp, int jo, int q) {
    if (jo < S::NB - 1 || S::R == 4) return *reinterpret_cast<const f32x4*>(rowp + 16 * jo + 4 * q);
    f32x4 v{0.f, 0.f, 0.f, 0.f};
    const float* p = rowp + 16 * (S::NB - 1) + S::R * q;
#pragma unroll
    for (int r = 0; r < S::R; ++r) v[r] = p[r];
    return v;
}
template <class S>
__device__ __forceinline__ void rr_row_store(float* __restrict__ rowp, int jo, int q, const f32x4& v) {
    if (jo < S::NB - 1 || S::R == 4) { *reinterpret_cast<f32x4*>(rowp + 16 * jo + 4 * q) = v; return; }
    float* p = rowp + 16 * (S::NB - 1) + S::R * q;
#pragma unroll
    for (int r = 0; r < S::R; ++r) p[r] = v[r];
}
