// Device-side building blocks shared by the MLP kernels (mlp_kernels.hip) and the fused
// Runge-Kutta step kernel (node_kernels.hip): MFMA tile GEMM with LDS-resident activations and
// fragment-packed weights, and the VALU skinny-layer contraction.
#pragma once
#include "common.h"

__device__ __forceinline__ int pad8(int x) { return (x + 7) & ~7; }
__device__ __forceinline__ int pad32(int x) { return (x + 31) & ~31; }

// row of accumulator register r for lane-half h in a 32x32 MFMA result (= the hidden unit within the wave's column
// tile in the transposed products of WaveGemm, = the output row in mlp_kernels.hip's weight-gradient GEMMs)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------
// C[32 x 32*NTW] += A_lds[32 x 8*KC] * Bpacked      (one wave, NTW = 1 or 2 tiles)
// Four K-chunks of operands are kept in flight in statically indexed registers
// (no register rotation), so the compiler emits counted s_waitcnt vmcnt(N) and
// every B-fragment load has three chunks of MFMA time to land.
// ---------------------------------------------------------------------------
// The WEIGHT fragment is the MFMA's first operand and the activation fragment its second: the product comes out
// transposed (D[hidden unit][row]), i.e. a lane holds, for ITS row (lane & 31), the hidden units
// 8 i + 4 (lane >> 5) + j (i, j = 0..3; register 4 i + j) of the wave's 32-column tile — four runs of four consecutive
// columns, so the epilogues move float4s (4 LDS writes per tile instead of 16, one mask word per lane in the backward)
// and a row's 32 ReLU bits are two lanes' halves.  Same products, same k order: bit-identical to the other operand
// order (measured on MI355X: fused RK forward 65 -> 60 us, backward 150 -> 138 us, update 0.88 -> 0.835 ms).
#define MFMA1(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32((b_), (a_), (c_), 0, 0, 0)
#define MFMA4(A, B0, B1)                                         \
    acc[0] = MFMA1((A).x, (B0).x, acc[0]);                       \
    if (NTW == 2) acc[1] = MFMA1((A).x, (B1).x, acc[1]);         \
    acc[0] = MFMA1((A).y, (B0).y, acc[0]);                       \
    if (NTW == 2) acc[1] = MFMA1((A).y, (B1).y, acc[1]);         \
    acc[0] = MFMA1((A).z, (B0).z, acc[0]);                       \
    if (NTW == 2) acc[1] = MFMA1((A).z, (B1).z, acc[1]);         \
    acc[0] = MFMA1((A).w, (B0).w, acc[0]);                       \
    if (NTW == 2) acc[1] = MFMA1((A).w, (B1).w, acc[1]);

// One wave's B-operand (weight) fragments form a stream that does not depend on the activations:
// chunk 0..KC-1 of this layer, then chunk 0.. of the next layer.  WaveGemm keeps B_DEPTH chunks of it
// in statically indexed registers; while a layer's last chunks are being multiplied the freed slots
// are refilled with the NEXT layer's first chunks, so after the inter-layer barrier the MFMAs start
// on registers that are already loaded (only the LDS-resident A fragments are fetched then).
// What follows the current layer in one wave's weight stream: the next layer's fragments and, for the
// slots the next layer is too short to use (layer 0 has 1-2 chunks), the layer after it.
struct NextFrags {
    const float4 *n0, *n1; int KCn;
    const float4 *nn0, *nn1; int KCnn;
    __device__ __forceinline__ const float4* at0(int i) const { return (i < KCn) ? n0 + (long)i * 64 : nn0 + (long)min(i, KCnn - 1) * 64; }
    __device__ __forceinline__ const float4* at1(int i) const { return (i < KCn) ? n1 + (long)i * 64 : nn1 + (long)min(i, KCnn - 1) * 64; }
};

// D: chunks of the weight stream in flight per column tile (a refill is issued behind its slot's MFMAs and needed D - 1
// chunks later).  Four is enough: tools/micro/gemm_loop.hip (MI355X) — a lone one-tile wave runs a 100-wide layer's 52
// MFMAs in 4.9k cycles (3.3k being the MFMA time, 3.8k with the weight loads compiled out) with four chunks in flight
// and in 5.2k with eight, so the gap is the wave's own load / LDS instruction issue between its MFMAs, not latency;
// two waves sharing a SIMD fill each other's gaps (6.5k for both GEMMs = the MFMA rate), which is how the kernels run.
#ifndef NLBAC_B_DEPTH1
#define NLBAC_B_DEPTH1 4
#endif
template <int NTW, int D = (NTW == 1) ? NLBAC_B_DEPTH1 : 4>
struct WaveGemm {
    float4 b0[D], b1[(NTW == 2) ? D : 1];

    // slot i <- chunk i of this layer, or the upcoming stream's slot-i fragment when this layer is shorter
    __device__ __forceinline__ void prime(const float4* p0, const float4* p1, int KC, const NextFrags& nx) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            b0[i] = *((i < KC) ? p0 + (long)i * 64 : nx.at0(i));
            if (NTW == 2) b1[i] = *((i < KC) ? p1 + (long)i * 64 : nx.at1(i));
        }
    }

    // acc += A_lds[32 x 8*KC] * B ; p*: this layer's fragments (already in the slots), nx: what comes next
    __device__ __forceinline__ void run(const float* __restrict__ arow, const float4* __restrict__ p0,
                                        const float4* __restrict__ p1, int KC, const NextFrags& nx,
                                        f32x16 (&acc)[2]) {
        const int last = KC - 1;
        float4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4*>(arow + min(i, last) * 8);
        int kc = 0;
        for (; kc + D <= KC; kc += D) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                MFMA4(a[i & 3], b0[i], b1[(NTW == 2) ? i : 0])
#ifndef EXP_NO_BLOAD          // (ablation: MFMAs run on stale weight registers, no L2 traffic)
                const int c = kc + i + D;
                const bool here = c <= last;
                b0[i] = *(here ? p0 + (long)c * 64 : nx.at0(i));
                if (NTW == 2) b1[i] = *(here ? p1 + (long)c * 64 : nx.at1(i));
#endif
                a[i & 3] = *reinterpret_cast<const float4*>(arow + min(kc + i + 4, last) * 8);
                __builtin_amdgcn_sched_barrier(0);   // keep each slot's refill right behind its MFMAs
            }
        }
#pragma unroll
        for (int i = 0; i < D - 1; ++i)
            if (kc + i < KC) {
                MFMA4(a[i & 3], b0[i], b1[(NTW == 2) ? i : 0])
                b0[i] = *nx.at0(i);
                if (NTW == 2) b1[i] = *nx.at1(i);
                if (i + 4 < D - 1) a[i & 3] = *reinterpret_cast<const float4*>(arow + min(kc + i + 4, last) * 8);
            }
    }
};

// Fragment pointers of one wave for a packed layer (tiles wave and wave+4)
__device__ __forceinline__ const float4* frag_ptr(const float* packed, int off, int KC, int tile, int lane) {
    return reinterpret_cast<const float4*>(packed + off) + (long)tile * KC * 64 + lane;
}

// skinny contraction helpers (VALU): NO outputs at once, every load unconditional
template <int NO>
__device__ __forceinline__ void skinny_dot(const float* __restrict__ lds_row, const float* __restrict__ W, int hid,
                                           int part, float (&acc)[4]) {
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
        const int k = part * 4 + 32 * it;
        const bool ok = k < hid;
        const int kk = ok ? k : 0;
        float4 h = *reinterpret_cast<const float4*>(lds_row + kk);
        if (!ok) h = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            const float4 w = *reinterpret_cast<const float4*>(W + (long)q * hid + kk);
            acc[q] += h.x * w.x + h.y * w.y + h.z * w.z + h.w * w.w;
        }
    }
}


// Skinny output layer, one (row, output) dot product per thread: thread t < 32*out_dim handles row t&31,
// output t>>5.  The 32 lanes of a half-wave read 32 different rows at the same k (ds_read_b128, row
// stride = 4 mod 8 dwords: conflict-free) and broadcast-read the weight row; no cross-lane reduction.
__device__ __forceinline__ float skinny_row_dot(const float* __restrict__ hrow, const float* __restrict__ w, int hid) {
    // 16 k per step: the eight ds_read_b128 are issued together (one LDS latency per step instead of one per float4;
    // the output layers of the fused RK kernel were latency bound at ~145 cycles per float4), same summation order
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = 0;
    for (; k + 16 <= hid; k += 16) {
        float4 h[4], ww[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h[q] = *reinterpret_cast<const float4*>(hrow + k + 4 * q);
            ww[q] = *reinterpret_cast<const float4*>(w + k + 4 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0 += h[q].x * ww[q].x; a1 += h[q].y * ww[q].y; a2 += h[q].z * ww[q].z; a3 += h[q].w * ww[q].w;
        }
    }
    for (; k < hid; k += 4) {
        const float4 h = *reinterpret_cast<const float4*>(hrow + k);
        const float4 ww = *reinterpret_cast<const float4*>(w + k);
        a0 += h.x * ww.x; a1 += h.y * ww.y; a2 += h.z * ww.z; a3 += h.w * ww.w;
    }
    return (a0 + a1) + (a2 + a3);
}

// ---------------------------------------------------------------------------
// Wide (MFMA) layers of one net for one 32-row tile, executed by a 4-wave group in lock step
// (every wave of the workgroup must call this with the same nwide_run: it contains the barriers).
//   forward : h_{l+1} = relu(h_l W_l^T + b_l),  l = 0..nwide-1, saved to acts when given
//   wrap    : the weight stream continues with layer 0 after the last layer (next RK stage)
// ---------------------------------------------------------------------------
// upcoming stream after wide layer l of a forward pass (wrap: layer 0 follows the last layer)
template <int NTW>
__device__ __forceinline__ NextFrags fwd_next(const nlbac_mlp& net, int l, int inp, bool wrap, int wave, int lane) {
    const int nwide = net.n_layers - 1, hidp8 = pad8(net.hid);
    const int ln = (l + 1 < nwide) ? l + 1 : (wrap ? 0 : l);
    const int lnn = (ln + 1 < nwide) ? ln + 1 : (wrap ? 0 : ln);
    NextFrags nx;
    nx.KCn = ((ln == 0) ? inp : hidp8) >> 3;
    nx.KCnn = ((lnn == 0) ? inp : hidp8) >> 3;
    nx.n0 = frag_ptr(net.packed, net.pf_off[ln], nx.KCn, wave, lane);
    nx.nn0 = frag_ptr(net.packed, net.pf_off[lnn], nx.KCnn, wave, lane);
    nx.n1 = (NTW == 2) ? frag_ptr(net.packed, net.pf_off[ln], nx.KCn, wave + 4, lane) : nx.n0;
    nx.nn1 = (NTW == 2) ? frag_ptr(net.packed, net.pf_off[lnn], nx.KCnn, wave + 4, lane) : nx.nn0;
    return nx;
}

template <int NTW, int D>
__device__ __forceinline__ void fwd_prime(WaveGemm<NTW, D>& wg, const nlbac_mlp& net, int inp, bool wrap, int wave,
                                          int lane) {
    const int KC = inp >> 3;
    wg.prime(frag_ptr(net.packed, net.pf_off[0], KC, wave, lane),
             frag_ptr(net.packed, net.pf_off[0], KC, (NTW == 2) ? wave + 4 : wave, lane), KC,
             fwd_next<NTW>(net, 0, inp, wrap, wave, lane));
}

// A barrier over ONE wave group of a workgroup (the fused NODE kernels run f_net and g_net in two groups of four waves
// with separate LDS tiles): an LDS arrival counter that only the group's own waves bump and poll, so the other group
// is free to run ahead between the stage boundaries (__syncthreads() there).  Every wave of the group must call it the
// same number of times; gb == nullptr is the plain workgroup barrier.
struct GroupBar {
    unsigned* cnt;       // LDS word, zeroed before the first use
    unsigned target;     // arrivals expected after the next sync (wave-uniform)
    unsigned n_waves;
};
__device__ __forceinline__ void tile_sync(GroupBar* gb, int lane) {
    if (!gb) { __syncthreads(); return; }
    gb->target += gb->n_waves;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // this wave's LDS writes have landed
    if (lane == 0) __hip_atomic_fetch_add(gb->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((int)(__builtin_amdgcn_readfirstlane(
                     __hip_atomic_load(gb->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - gb->target) < 0)
#ifdef EXP_NOSLEEP
        ;
#else
        __builtin_amdgcn_s_sleep(1);
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int NTW, int BITS = 0, int D = 4>
__device__ __forceinline__ void fwd_wide_layers(WaveGemm<NTW, D>& wg, const nlbac_mlp& net, bool active, int wave, int lane,
                                                int LD, int inp, float*& in, float*& out, float* acts_tile,
                                                long acts_ls, int n_rows, int nwide_run, bool wrap,
                                                long long* dbg = nullptr /* ablation builds: per-layer clock stamps */,
                                                GroupBar* gb = nullptr) {
    // BITS: acts_tile holds bit-packed ReLU masks instead of activations: uint32 word [layer][row][col tile]
    // (bit = column within the 32-wide tile), enough for a backward that needs no weight gradients
    const int hid = net.hid, hidp8 = pad8(hid), nwide = net.n_layers - 1, half = lane >> 5;
    for (int l = 0; l < nwide_run; ++l) {
        if (l < nwide && active) {
            const int KC = ((l == 0) ? inp : hidp8) >> 3;
            const float4* p0 = frag_ptr(net.packed, net.pf_off[l], KC, wave, lane);
            const float4* p1 = (NTW == 2) ? frag_ptr(net.packed, net.pf_off[l], KC, wave + 4, lane) : p0;
            const NextFrags nx = fwd_next<NTW>(net, l, inp, wrap, wave, lane);
            const float* bias = net.params + net.b_off[l];
            // biases of this lane's 4 x 4 columns per tile (hid % 4 == 0 and every parameter tensor of a net starts on a
            // 16-byte boundary of its arena), requested before the GEMM
            float4 bv[2][4];
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    bv[t][i] = *reinterpret_cast<const float4*>(bias + min((wave + 4 * t) * 32 + 8 * i + 4 * half, hid - 4));
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
            if (dbg) dbg[4 * l + 0] = (long long)__builtin_readcyclecounter();
#ifndef EXP_NO_GEMM
            wg.run(in + (lane & 31) * LD + half * 4, p0, p1, KC, nx, acc);
#endif
            if (dbg) dbg[4 * l + 1] = (long long)__builtin_readcyclecounter();
            float* acts = acts_tile ? acts_tile + (long)l * acts_ls : nullptr;
            const int m = lane & 31;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                unsigned bits = 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = (wave + 4 * t) * 32 + 8 * i + 4 * half;
                    const bool colok = c0 < hid;
                    float4 v;
                    v.x = colok ? fmaxf(acc[t][4 * i + 0] + bv[t][i].x, 0.f) : 0.f;
                    v.y = colok ? fmaxf(acc[t][4 * i + 1] + bv[t][i].y, 0.f) : 0.f;
                    v.z = colok ? fmaxf(acc[t][4 * i + 2] + bv[t][i].z, 0.f) : 0.f;
                    v.w = colok ? fmaxf(acc[t][4 * i + 3] + bv[t][i].w, 0.f) : 0.f;
                    if (c0 < LD - 4) *reinterpret_cast<float4*>(out + m * LD + c0) = v;   // (a narrow LD holds pad8(hid) columns)
                    if (BITS) {
                        bits |= (v.x > 0.f ? 1u : 0u) << (8 * i + 0);
                        bits |= (v.y > 0.f ? 1u : 0u) << (8 * i + 1);
                        bits |= (v.z > 0.f ? 1u : 0u) << (8 * i + 2);
                        bits |= (v.w > 0.f ? 1u : 0u) << (8 * i + 3);
                    } else if (acts && colok && m < n_rows) {
                        *reinterpret_cast<float4*>(acts + (long)m * hid + c0) = v;
                    }
                }
                if (BITS && acts) {
                    // row m's mask word of this column tile: this lane's 16 bits and those of lane m + 32
                    bits <<= 4 * half;
                    const unsigned word = bits | (unsigned)__shfl_xor((int)bits, 32, 64);
                    const int NTm = (hid + 31) >> 5;
                    if (lane < 32 && m < n_rows) (reinterpret_cast<unsigned*>(acts) + (wave + 4 * t))[m * NTm] = word;
                }
            }
        }
        if (dbg) dbg[4 * l + 2] = (long long)__builtin_readcyclecounter();
        tile_sync(gb, lane);
        if (dbg) dbg[4 * l + 3] = (long long)__builtin_readcyclecounter();
        if (l < nwide) { float* tmp = in; in = out; out = tmp; }
    }
}

// top (skinny) layer of the backward: s[m] += sum_o dy[m][o] W_last[o][k] for this thread's column k
template <int NO>
__device__ __forceinline__ void top_layer_bwd(const float* __restrict__ sdy, const float* __restrict__ W, int hid,
                                              int k, float (&s)[NLBAC_MLP_TILE]) {
    float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NO; ++q) w[q] = W[(long)q * hid + k];
#pragma unroll
    for (int m = 0; m < NLBAC_MLP_TILE; ++m) {
        const float4 d = *reinterpret_cast<const float4*>(sdy + m * 16);
        s[m] += d.x * w[0] + d.y * w[1] + d.z * w[2] + d.w * w[3];
    }
}

// A finished 32 x hid tile in LDS -> its rows in global memory (hid % 4 == 0): float4 per lane, coalesced.  Keeping the
// global dz stores out of the MFMA epilogues (which would hold 16 row addresses per lane) is what keeps those
// epilogues free of register spills; the copy of layer j's tile overlaps layer j-1's GEMM.
__device__ __forceinline__ void tile_to_global(const float* __restrict__ tile, int LD, float* __restrict__ g, int hid,
                                               int n_rows, int t, int nthr) {
    const int q = hid >> 2;
    for (int idx = t; idx < n_rows * q; idx += nthr) {
        const int r = idx / q, c = (idx - r * q) << 2;
        *reinterpret_cast<float4*>(g + r * hid + c) = *reinterpret_cast<const float4*>(tile + r * LD + c);
    }
}

// backward wide layers: dz[j-1] = (dz[j] W_j) * [acts[j-1] > 0] for j = nwide-1 .. 1 (backward packs)
// (wrap: after layer 1 the stream restarts at layer nwide-1 — the next RK stage of a fused backward)
template <int NTW>
__device__ __forceinline__ NextFrags bwd_next(const nlbac_mlp& net, int j, int wave, int lane, bool wrap = false) {
    const int KC = pad8(net.hid) >> 3;
    const int jn = j > 1 ? j - 1 : (wrap ? net.n_layers - 2 : j);
    NextFrags nx;
    nx.KCn = nx.KCnn = KC;
    nx.n0 = nx.nn0 = frag_ptr(net.packed, net.pb_off[jn], KC, wave, lane);
    nx.n1 = nx.nn1 = (NTW == 2) ? frag_ptr(net.packed, net.pb_off[jn], KC, wave + 4, lane) : nx.n0;
    return nx;
}

template <int NTW, int D>
__device__ __forceinline__ void bwd_prime(WaveGemm<NTW, D>& wg, const nlbac_mlp& net, int wave, int lane,
                                          bool wrap = false) {
    const int nwide = net.n_layers - 1, KC = pad8(net.hid) >> 3;
    if (nwide < 2) return;
    const int j = nwide - 1;
    wg.prime(frag_ptr(net.packed, net.pb_off[j], KC, wave, lane),
             frag_ptr(net.packed, net.pb_off[j], KC, (NTW == 2) ? wave + 4 : wave, lane), KC,
             bwd_next<NTW>(net, j, wave, lane, wrap));
}

template <int NTW, int BITS = 0, int D = 4>
__device__ __forceinline__ void bwd_wide_layers(WaveGemm<NTW, D>& wg, const nlbac_mlp& net, bool active, int wave, int lane,
                                                int LD, float*& in, float*& out, const float* acts_tile,
                                                float* dz_tile, long ls, int n_rows, int row_clamp,
                                                int n_run = -1, bool wrap = false, int nthr = 256,
                                                GroupBar* gb = nullptr, float* colsum = nullptr, long colsum_step = 0) {
    const int hid = net.hid, KC = pad8(hid) >> 3, nwide = net.n_layers - 1, half = lane >> 5;
    if (n_run < 0) n_run = nwide - 1;          // lock-step iterations (>= nwide-1 when groups differ in depth)
    for (int it = 0; it < n_run; ++it) {
        const int j = nwide - 1 - it;
        if (j >= 1 && active) {
            const float4* p0 = frag_ptr(net.packed, net.pb_off[j], KC, wave, lane);
            const float4* p1 = (NTW == 2) ? frag_ptr(net.packed, net.pb_off[j], KC, wave + 4, lane) : p0;
            const NextFrags nx = bwd_next<NTW>(net, j, wave, lane, wrap);
            // ReLU masks of this lane's columns of row (lane & 31), requested before the GEMM so they land under it:
            // one mask word per column tile, or the activations themselves as four float4
            const float* acts = acts_tile + (long)(j - 1) * ls;
            const int m = lane & 31, mc = min(m, row_clamp);
            unsigned mword[2];
            float4 av[2][4];
            if constexpr (BITS != 0) {
                const unsigned* mk = reinterpret_cast<const unsigned*>(acts_tile) + (long)(j - 1) * ls;
                const int NTm = (hid + 31) >> 5;
#pragma unroll
                for (int t = 0; t < NTW; ++t) mword[t] = mk[mc * NTm + (wave + 4 * t)] >> (4 * half);
            } else {
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        av[t][i] = *reinterpret_cast<const float4*>(
                            acts + (long)mc * hid + min((wave + 4 * t) * 32 + 8 * i + 4 * half, hid - 4));
            }
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#ifndef EXP_BWD_NO_GEMM
            wg.run(in + (lane & 31) * LD + half * 4, p0, p1, KC, nx, acc);
#endif
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = (wave + 4 * t) * 32 + 8 * i + 4 * half;
                    const bool ok = (c0 < hid) && (m < n_rows);
                    bool q0, q1, q2, q3;
                    if constexpr (BITS != 0) {
                        q0 = (mword[t] >> (8 * i + 0)) & 1u; q1 = (mword[t] >> (8 * i + 1)) & 1u;
                        q2 = (mword[t] >> (8 * i + 2)) & 1u; q3 = (mword[t] >> (8 * i + 3)) & 1u;
                    } else {
                        q0 = av[t][i].x > 0.f; q1 = av[t][i].y > 0.f; q2 = av[t][i].z > 0.f; q3 = av[t][i].w > 0.f;
                    }
                    float4 o;
                    o.x = (ok && q0) ? acc[t][4 * i + 0] : 0.f;
                    o.y = (ok && q1) ? acc[t][4 * i + 1] : 0.f;
                    o.z = (ok && q2) ? acc[t][4 * i + 2] : 0.f;
                    o.w = (ok && q3) ? acc[t][4 * i + 3] : 0.f;
                    *reinterpret_cast<float4*>(out + m * LD + c0) = o;
                }
            }
        }
        tile_sync(gb, lane);
        if (j >= 1) {
            float* tmp = in; in = out; out = tmp;
            // dz[j-1] now sits complete in `in`: stream it out while the next layer's GEMM runs (mask mode keeps no dz)
            if constexpr (BITS == 0) {
                if (dz_tile) tile_to_global(in, LD, dz_tile + (long)(j - 1) * ls, hid, n_rows, wave * 64 + lane, nthr);
                if (colsum) {      // bias-gradient partials of layer j-1: this thread's column over the tile's rows, in order,
                                   // one per 16 rows (NLBAC_SK_CHUNK), colsum_step floats apart
                    const int col = wave * 64 + lane;
#pragma unroll
                    for (int hh = 0; hh < NLBAC_MLP_TILE / 16; ++hh) {
                        float a = 0.f;
                        if (col < hid)
                            for (int m = 0; m < 16; ++m) a += in[(hh * 16 + m) * LD + col];
                        colsum[hh * colsum_step + (long)(j - 1) * 256] = a;
                    }
                }
            }
        }
    }
}

// Top (skinny) layer of the backward for one 32-row tile: dz_top[m][k] = [act_top[m][k] > 0] sum_o dy[m][o] W_last[o][k].
// RPT rows per thread: the 256 threads of a group cover (32 / RPT) row groups x (256 * RPT / 32) columns.
// ReLU masks of the top hidden layer for this thread's (row group, column): issued at the start of a stage so that their
// global latency (~2k cycles) hides under the output-gradient fill and the first barrier.
template <int RPT, int BITS, int NTHR = 256>
__device__ __forceinline__ void node_top_masks(const float* __restrict__ atop, int hid, int NT, int t, int n_rows,
                                               float (&av)[RPT]) {
    constexpr int CPG = NTHR * RPT / 32;
    const int k = t % CPG, m0 = (t / CPG) * RPT, kc = min(k, hid - 1);
    if constexpr (BITS != 0) {
        const unsigned* mtop = reinterpret_cast<const unsigned*>(atop) + (kc >> 5);
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            av[i] = ((mtop[min(m0 + i, n_rows - 1) * NT] >> (kc & 31)) & 1u) ? 1.f : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < RPT; ++i) av[i] = atop[min(m0 + i, n_rows - 1) * hid + kc];
    }
}

template <int RPT, int BITS, int NTHR = 256>
__device__ __forceinline__ void node_top_layer(const float* __restrict__ sdy, const float* __restrict__ sW, int out_dim,
                                               int hid, int hidp32, int NT, int t, int n_rows,
                                               const float (&av)[RPT], float* __restrict__ in, int LD) {
    constexpr int CPG = NTHR * RPT / 32;           // columns per row group
    const int k = t % CPG, m0 = (t / CPG) * RPT, kc = min(k, hid - 1);
    float s[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) s[i] = 0.f;
#ifndef EXP_BWD_NO_TOP
    for (int o0 = 0; o0 < out_dim; o0 += 4) {
        const int no = min(4, out_dim - o0);
        float w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (q < no) ? sW[(o0 + min(q, no - 1)) * hid + kc] : 0.f;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const float4 d = *reinterpret_cast<const float4*>(sdy + (m0 + i) * 16 + o0);
            s[i] += d.x * w[0] + d.y * w[1] + d.z * w[2] + d.w * w[3];
        }
    }
#endif
    if (k < hidp32) {
        const bool colok = k < hid;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int m = m0 + i;
            const bool ok = colok && (m < n_rows);
            const float v = (ok && av[i] > 0.f) ? s[i] : 0.f;
            in[m * LD + k] = v;        // (dz goes out from this LDS tile afterwards: tile_to_global)
        }
    }
}

