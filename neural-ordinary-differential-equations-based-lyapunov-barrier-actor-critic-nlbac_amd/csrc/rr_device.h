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

#ifndef RR_DEPTH_MAX
#define RR_DEPTH_MAX 8       /* tools/micro/rr_chain.hip: 4 float4 in flight stream as fast as 11 or 22 */
#endif
__host__ __device__ constexpr int rr_pick_depth(int nv, int dmax = RR_DEPTH_MAX) {
    for (int d = dmax; d >= 4; --d)
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
// MFMA issue order of a layer: output blocks in groups — the first of three blocks when NB is odd, pairs otherwise —
// k-steps inner, so two / three independent accumulator chains alternate (v_mfma_f32_16x16x4_f32: 32 cycles issue, 40
// dependent) and a finished group's bias / ReLU can be issued between the next group's MFMAs
__host__ __device__ __forceinline__ constexpr int rr_group_first(int NB) { return (NB & 1) ? 3 : 2; }
__host__ __device__ __forceinline__ void rr_mfma_of(int NB, int KS, int m, int& jo, int& ks) {
    int g0 = 0, gn = rr_group_first(NB);
    while (m >= gn * KS) { m -= gn * KS; g0 += gn; gn = 2; }
    ks = m / gn;
    jo = g0 + m % gn;
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

// One mask bit shifted into a lane's word: word = (word << 1) | (h > 0) for a ReLU output h (a non-negative float: its
// bit pattern is a non-negative integer, zero only for +0; the sign of its negation is the bit).  Two VALU instructions
// (v_sub_u32, v_alignbit_b32).  After the KS values of a layer have been shifted in — in ascending order of their
// register index k — value k's bit sits at position KS-1-k (rr_mask_on).
__device__ __forceinline__ void rr_mask_push(unsigned& word, float h) {
    const unsigned neg = 0u - __builtin_bit_cast(unsigned, h);
    word = __builtin_amdgcn_alignbit(word, neg, 31);
}
// value k of a layer's KS gated by its mask word: v_bfe_i32 (all-ones / zero) + v_and_b32.  (x by value: clang's
// __builtin_bit_cast applied directly to a vector-element lvalue reads element 0.)
template <int KS>
__device__ __forceinline__ float rr_mask_gate(unsigned word, int k, float x) {
    const unsigned on = (unsigned)(((int)(word << (31 - (KS - 1 - k)))) >> 31);
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & on);
}

// ReLU as an integer max: negative floats (and -0) are negative integers; one VALU op, no NaN canonicalisation
__device__ __forceinline__ float rr_relu(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

// Empty volatile asm statements that "use and redefine" a value: volatile asms keep their program order among
// themselves, so whatever produces the value stays before the statement and whatever consumes it after it.
#ifdef RR_NO_PIN_S
#define RR_PIN_S(x_)
#else
#define RR_PIN_S(x_) asm volatile("" : "+s"(x_));
#endif
#ifdef RR_ACC_VGPR          /* built with -mllvm -amdgpu-mfma-vgpr-form: accumulators live in the VGPR half */
#define RR_ACC_C(x_) "+v"(x_)
#else
#define RR_ACC_C(x_) "+a"(x_)
#endif
#ifdef RR_NO_PIN_A
#define RR_PIN_A2(a_, b_)
#define RR_PIN_A3(a_, b_, c_)
#else
#define RR_PIN_A2(a_, b_) asm volatile("" : RR_ACC_C(a_), RR_ACC_C(b_));
#define RR_PIN_A3(a_, b_, c_) asm volatile("" : RR_ACC_C(a_), RR_ACC_C(b_), RR_ACC_C(c_));
#endif

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

    // cur / nxt: byte offsets of this layer's fragments and of the layer that follows it in the wave's stream.
    // Nothing but the MFMA stream is exposed: the per-value work around a product (bias, ReLU / mask, saving what the
    // backward needs: ~6 VALU instructions per value) is issued between MFMAs, in gaps the matrix pipe leaves free —
    //   pre(ks)     in the first group, before k-step ks reads H[ks]: finishes what the PREVIOUS product left pending
    //               (its last group of blocks, or all of a short product's outputs), just in time;
    //   fin(jo, r)  accumulator register r of output block jo, for every group but the last, one value per k-step of the
    //               group that follows;
    //   mid()       once, before the last group starts (prefetches for the next product);
    // the last group's accumulators (blocks NB-2, NB-1) are left as they are for the next product's `pre`.
    // Every one of those VALU instructions costs matrix-pipe time: v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate —
    // on the same FMA lanes — so a VALU instruction between two MFMAs is not hidden, it takes its ~4-5 cycles from the
    // product (measured: 3 extra instructions per value = +800 cycles on a 5.6k-cycle layer).  Hence: accumulators in
    // the VGPR half (no v_accvgpr_read), the bias as the first MFMA's C operand (no add), one integer max for the ReLU,
    // two instructions for a mask bit.
    template <class PRE, class FIN, class MID>
    __device__ __forceinline__ void run(f32x4 (&acc)[S::NB], const f32x4 (&cinit)[S::NB], float (&H)[S::KS],
                                        __amdgpu_buffer_rsrc_t rs, int voff, int cur, int nxt, PRE&& pre, FIN&& fin, MID&& mid) {
        constexpr int NB = S::NB, KS = S::KS, NM = S::NM, NV = S::NV, D = S::D;
        static_assert(KS >= 12 && NB >= 4, "a group's epilogue (<= 12 values) is spread over the next group's k-steps");
        // (opaque to the optimiser: otherwise every load's scalar offset is hoisted out of the caller's stage loop as
        // a loop invariant of its own — hundreds of live SGPRs, spilled — instead of one s_add next to the load)
        asm volatile("" : "+s"(cur), "+s"(nxt));
        int m = 0;
#pragma unroll
        for (int g0 = 0; g0 < NB;) {
            const int gn = (g0 == 0) ? rr_group_first(NB) : 2;
            const int pn = (g0 == 0) ? 0 : ((g0 == rr_group_first(NB)) ? rr_group_first(NB) : 2), p0 = g0 - pn;   // previous group
            if (g0 + gn == NB) mid();
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                if (jj < gn) acc[g0 + jj] = cinit[g0 + jj];          // (the first MFMA's C operand: the layer's bias, or zero)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (g0 == 0) pre(ks);
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    if (jj < gn) {
                        const int v = m >> 2, c = m & 3;
                        acc[g0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[v % D][c], H[ks], acc[g0 + jj], 0, 0, 0);
                        if (c == 3 || m == NM - 1) {          // slot v % D is free: refill with stream element v + D
                            const int vn = v + D;
                            int so = (vn < NV) ? cur + vn * 1024 : nxt + (vn - NV) * 1024;
                            RR_PIN_S(so)                         // (the load stays behind its slot's MFMAs: see RR_PIN_*)
                            wq[v % D] = rr_ldw(rs, voff, so);
                            __builtin_amdgcn_sched_barrier(0);   // keep each refill right behind its slot's MFMAs
                        }
                        ++m;
                    }
                }
                // program-order anchors: MFMAs are pure to the optimiser, which otherwise sinks whole k-steps of them
                // past the refills and the interleaved per-value work (19 loads in flight, 247 VGPRs)
                if (gn == 3) { RR_PIN_A3(acc[g0], acc[g0 + 1], acc[g0 + 2]) } else { RR_PIN_A2(acc[g0], acc[g0 + 1]) }
                if (ks < 4 * pn) fin(p0 + (ks >> 2), ks & 3);
            }
            g0 += gn;
        }
    }

    // a single-block product o = sum_ks a[ks] * H[ks] (output layers: <= 16 outputs) on two accumulator chains — one chain
    // would run at the instruction's 40-cycle dependent latency instead of its 32-cycle issue rate —, with the same
    // just-in-time `pre`
    template <class PRE>
    __device__ __forceinline__ static f32x4 block(const float (&a)[S::KS], float (&H)[S::KS], PRE&& pre) {
        f32x4 o0{0.f, 0.f, 0.f, 0.f}, o1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < S::KS; ++ks) {
            pre(ks);
            if (ks & 1) o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], H[ks], o1, 0, 0, 0);
            else o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], H[ks], o0, 0, 0, 0);
        }
        return o0 + o1;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// A contiguous range [M0, M1) of a layer's MFMA stream — whole groups of output blocks, all KS k-steps — with a weight
// queue of its own: two waves share ONE hid x hid product by output blocks (node_rr_kernels.hip: the g_net wave of a tile
// takes over part of f_net's last layer, whose waves would otherwise run one layer longer than g_net's).  Same hooks as
// RRGemm::run; the range's last group of blocks is left pending for the caller (the next product's `pre`).
// ---------------------------------------------------------------------------------------------------------------------
template <class S, int M0, int M1>
struct RRPart {
    static constexpr int NB = S::NB, KS = S::KS, G0 = rr_group_first(NB), D = 4;
    static constexpr int V0 = M0 >> 2, V1 = (M1 + 3) >> 2;                  // float4 of the layer's stream the range touches
    static constexpr int block_of(int m) { return m < G0 * KS ? 0 : G0 + 2 * ((m - G0 * KS) / (2 * KS)); }
    static constexpr int J0 = block_of(M0), J1 = (M1 >= S::NM) ? NB : block_of(M1);   // its output blocks [J0, J1)
    static constexpr int K0 = 4 * J0, K1 = (J1 == NB) ? KS : 4 * J1;         // the values (next layer's k-steps) it yields
    static constexpr int JT = (J1 - J0 > G0 || J0 > 0) ? J1 - 2 : J0;         // first block of its last group
    static_assert(M0 % KS == 0 && M1 % KS == 0 && M0 < M1 && M1 <= S::NM, "a range is whole groups of blocks");
    f32x4 wq[D];

    // cur: byte offset of the LAYER's stream (wave-uniform)
    __device__ __forceinline__ void prime(__amdgpu_buffer_rsrc_t rs, int voff, int cur) {
#pragma unroll
        for (int i = 0; i < D; ++i) wq[i] = (V0 + i < V1) ? rr_ldw(rs, voff, cur + (V0 + i) * 1024) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    template <class PRE, class FIN>
    __device__ __forceinline__ void run(f32x4 (&acc)[NB], const f32x4 (&cinit)[NB], float (&H)[KS], __amdgpu_buffer_rsrc_t rs,
                                        int voff, int cur, PRE&& pre, FIN&& fin) {
        asm volatile("" : "+s"(cur));
        int m = M0;
#pragma unroll
        for (int g0 = J0; g0 < J1;) {
            const int gn = (g0 == 0) ? G0 : 2;
            const int pn = (g0 == J0) ? 0 : ((g0 == G0) ? G0 : 2), p0 = g0 - pn;     // previous group of the range
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                if (jj < gn) acc[g0 + jj] = cinit[g0 + jj];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (g0 == J0) pre(ks);
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    if (jj < gn) {
                        const int v = m >> 2, c = m & 3;
                        acc[g0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[(v - V0) % D][c], H[ks], acc[g0 + jj], 0, 0, 0);
                        if (c == 3 || m == M1 - 1) {
                            const int vn = v + D;
                            if (vn < V1) {
                                int so = cur + vn * 1024;
                                RR_PIN_S(so)
                                wq[(v - V0) % D] = rr_ldw(rs, voff, so);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ++m;
                    }
                }
                if (gn == 3) { RR_PIN_A3(acc[g0], acc[g0 + 1], acc[g0 + 2]) } else { RR_PIN_A2(acc[g0], acc[g0 + 1]) }
                if (ks < 4 * pn) fin(p0 + (ks >> 2), ks & 3);
            }
            g0 += gn;
        }
    }

    // the single-block product over the range's values: o = sum_{ks in [K0, K1)} a[ks] * H[ks], two accumulator chains
    template <class PRE>
    __device__ __forceinline__ static f32x4 block(const float (&a)[KS], float (&H)[KS], PRE&& pre) {
        f32x4 o0{0.f, 0.f, 0.f, 0.f}, o1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = K0; ks < K1; ++ks) {
            pre(ks);
            if (ks & 1) o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], H[ks], o1, 0, 0, 0);
            else o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], H[ks], o0, 0, 0, 0);
        }
        return o0 + o1;
    }
};
// where a layer's stream is cut for two waves: the first floor(groups / 2) groups of blocks
template <class S>
constexpr int rr_split_m() {
    constexpr int G0 = rr_group_first(S::NB), groups = 1 + (S::NB - G0) / 2, first = groups / 2;
    return first == 0 ? 0 : (G0 + 2 * (first - 1)) * S::KS;
}

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


// ---------------------------------------------------------------------------------------------------------------------
// A PANEL of a hid x hid layer for the 256-wide actor / critic nets (mlp_rr_kernels.hip): NBO output blocks (one half
// of the layer's units: two waves share a 16-row tile's layer, each reading all KS k-steps of the input), in groups of
// two, k-steps inner; the same hooks as RRGemm::run.  Fragment stream: the panel's NBO * KS / 4 float4 per lane, in
// issue order ("panel pack", written by nlbac_mlp_pack).
// ---------------------------------------------------------------------------------------------------------------------
// DMAX: float4 of the stream in flight per lane at most (kernels that share a SIMD between three waves keep 4)
template <int NBO, int KS, int DMAX = RR_DEPTH_MAX>
struct RRPanel {
    static constexpr int NM = NBO * KS, NV = NM / 4, D = rr_pick_depth(NV, DMAX), BYTES = NV * 1024;
    static_assert(NBO % 2 == 0 && KS % 4 == 0 && D >= 4, "panels hold an even number of 16-unit blocks");
    f32x4 wq[D];

    __device__ __forceinline__ void prime(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
#pragma unroll
        for (int i = 0; i < D; ++i) wq[i] = rr_ldw(rs, voff, soff + i * 1024);
    }

    // (the stream ends with the panel: the last D refills re-read its own head, harmlessly)
    template <class PRE, class FIN>
    __device__ __forceinline__ void run(f32x4 (&acc)[NBO], const f32x4 (&cinit)[NBO], float (&H)[KS],
                                        __amdgpu_buffer_rsrc_t rs, int voff, int cur, PRE&& pre, FIN&& fin) {
        asm volatile("" : "+s"(cur));
        int m = 0;
#pragma unroll
        for (int g0 = 0; g0 < NBO; g0 += 2) {
            acc[g0] = cinit[g0];
            acc[g0 + 1] = cinit[g0 + 1];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (g0 == 0) pre(ks);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int v = m >> 2, c = m & 3;
                    acc[g0 + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[v % D][c], H[ks], acc[g0 + jj], 0, 0, 0);
                    if (c == 3) {
                        const int vn = v + D;
                        int so = cur + (vn < NV ? vn : vn - NV) * 1024;
                        RR_PIN_S(so)
                        wq[v % D] = rr_ldw(rs, voff, so);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ++m;
                }
                RR_PIN_A2(acc[g0], acc[g0 + 1])
                if (g0 > 0 && ks < 8) fin(g0 - 2 + (ks >> 2), ks & 3);
            }
        }
    }

    // a single-block product over KSB k-steps of H starting at H[k0], two accumulator chains, just-in-time `pre`
    template <int KSB, class PRE>
    __device__ __forceinline__ static f32x4 block(const float (&a)[KSB], const float* H, PRE&& pre) {
        f32x4 o0{0.f, 0.f, 0.f, 0.f}, o1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            pre(ks);
            if (ks & 1) o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], H[ks], o1, 0, 0, 0);
            else o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], H[ks], o0, 0, 0, 0);
        }
        return o0 + o1;
    }
};

// floats of a panel net's layer-0 fragment block: four k-steps (in_dim + 1 <= 16) x hid / 16 blocks x 64 lanes
__host__ __device__ __forceinline__ long rr_panel_l0_floats(int hid) { return 4L * (hid >> 4) * 64; }

// which nets get which RR pack (nlbac_mlp.rr_kind)
#define RR_KIND_NONE 0
#define RR_KIND_CHAIN 1      /* n_layers >= 4, hid <= 128: every hid x hid layer, RRGemm order (the NODE nets) */
#define RR_KIND_PANEL 2      /* n_layers == 3, hid % 32 == 0: the one hid x hid layer as two panels (actor / critic nets) */
__host__ __device__ __forceinline__ int rr_kind_of(int n_layers, int hid) {
    if (n_layers >= 4 && rr_width_ok(hid)) return RR_KIND_CHAIN;
    if (n_layers == 3 && hid % 64 == 0 && hid >= 64 && hid <= 256) return RR_KIND_PANEL;
    return RR_KIND_NONE;
}
