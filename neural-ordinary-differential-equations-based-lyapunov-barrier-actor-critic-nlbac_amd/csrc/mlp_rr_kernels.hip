// Register-resident forward of the actor / critic nets (QNetwork, LyaNetwork, GaussianPolicy, BarrierNetwork:
// in -> hid -> hid -> out, U/sac_cbf_clf/model.py:37-114): the launches of nlbac_mlp_fwd / nlbac_mlp_fwd_gauss for nets
// whose one hid x hid layer has a panel pack (rr_device.h, RRPanel; hid = 64 / 128 / 256).
//
// A workgroup is still one 32-row tile of one net (grid.y = net), but its four waves no longer share a layer through LDS
// tiles and barriers: wave (rh, ch) owns rows 16 rh .. 16 rh + 15 and the output blocks of panel ch.  It computes layer 0
// for ALL units itself (K <= 16: a few dozen MFMAs, cheaper than exchanging it), keeps those activations in registers as
// the B operands of its panel's MFMA stream, and contributes its half of the skinny output layer; the two halves meet in
// LDS.  The LDS-tiled kernels run such a tile as a latency chain of ~16 us whatever the grid (a 3-net launch of 384
// workgroups: 33-38 us; 6 nets: 40 us); here a wave is ~620 MFMAs = 20k cycles of matrix-pipe time and a launch costs
// what its MFMAs cost.
#include "mlp_launch.h"
#include "rr_device.h"
#include <cstdlib>

#define MRR_MAX_IN 15       /* in_dim + the bias column <= 16: four k-steps of layer 0 */

#ifdef RR_TIMING      // ablation build: wave 0 of workgroup 0 stamps the shader clock behind the first net's outputs (as int64)
#define MSTAMP(k_) if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) reinterpret_cast<long long*>(io.y + (long)B * io.y_ld)[k_] = (long long)__builtin_readcyclecounter();
// (the data backward's stamps land behind the net's two dz layers: the caller of a timing build allocates a third)
#define BSTAMP(k_) if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) reinterpret_cast<long long*>(io.dz + 2 * ls)[k_] = (long long)__builtin_readcyclecounter();
#else
#define MSTAMP(k_)
#define BSTAMP(k_)
#endif

// BITS: the launch also leaves ReLU mask words (nlbac_mlp_io::masks: per layer and row 8 words, word 2 q + ch = the
// units 16 (NBH ch + j) + 4 q + r of lane quarter q in panel ch, value t = 4 j + r at bit 4 NBH - 1 - t) — what the data
// backward gates with; nets that are only differentiated w.r.t. their inputs keep nothing else (io.acts == NULL).
template <int NBH, int BITS>
__global__ __launch_bounds__(256) void mlp_rr_fwd_kernel(const MlpLaunch L, const nlbac_gauss_head G) {
    constexpr int HID = 32 * NBH, NBA = 2 * NBH, KS = HID / 4, KSH = KS / 2;
    using P = RRPanel<NBH, KS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sO = smem;                               // [2][32][16] the panels' halves of the output layer
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), rh = wave >> 1, ch = wave & 1;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    const int q = lane >> 4, r16 = lane & 15, m = 16 * rh + r16, grow = row0 + m;
    const bool row_ok = grow < B;
    const int idim = net.in_dim, odim = net.out_dim;
    const int KL0 = (idim + 4) >> 2;                      // k-steps of layer 0 over [x | 1] (1..4)
    const float* const params = net.params;

    MSTAMP(0)
    // ---- the panel's weight stream
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wcur = (net.rr_fwd_off + ch * (HID * HID / 2)) * 4;
    P panel;
    panel.prime(rs, voff, wcur);

    // ---- this lane's inputs: component 4 k0 + q of its row, a 1 in the bias column behind the last one
    float yv[4];
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) {
        const int c = 4 * k0 + q;
        float v = (c == idim) ? 1.f : 0.f;
        if (row_ok && c < idim) v = (c < io.x0_dim) ? io.x0[(long)grow * io.x0_ld + c] : io.x1[(long)grow * io.x1_ld + (c - io.x0_dim)];
        yv[k0] = v;
    }
    // ---- this wave's half of the output layer's A fragments: lane (o, kq) supplies W_out[o][unit of k-step ks, quarter kq]
    float wo[KSH];
    {
        const float* wrow = params + net.w_off[2] + (long)min(r16, odim - 1) * HID + 16 * NBH * ch;
#pragma unroll
        for (int j = 0; j < NBH; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 16 * j + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[4 * j + r] = (r16 < odim) ? v[r] : 0.f;
        }
    }
    // the layer's biases enter as the C operands of this panel's blocks
    f32x4 cinit[NBH];
#pragma unroll
    for (int jo = 0; jo < NBH; ++jo) cinit[jo] = *reinterpret_cast<const f32x4*>(params + net.b_off[1] + 16 * (NBH * ch + jo) + 4 * q);
    MSTAMP(1)

    const long ls = io.acts_ls ? io.acts_ls : (long)B * HID;
    float* const a0row = (io.acts && ch == 0 && row_ok) ? io.acts + (long)grow * HID : nullptr;      // layer 0: one wave saves it
    float* const a1row = (io.acts && row_ok) ? io.acts + ls + (long)grow * HID + 16 * NBH * ch : nullptr;

    // ---- layer 0 for all units (bias folded into the product), ReLU in place
    float H0[KS];
    {
        // its A fragments come from the pack (four output blocks per load), layer 0's bias folded into the product
        const int l0 = (net.rr_bwd_off + HID * HID) * 4;
#pragma unroll
        for (int j4 = 0; j4 < NBA / 4; ++j4) {
            f32x4 w[4];
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) w[k0] = (k0 < KL0) ? rr_ldw(rs, voff, l0 + (k0 * (NBA / 4) + j4) * 1024) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int jo = 4 * j4 + c;
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][c], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (KL0 > 1) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][c], yv[1], a, 0, 0, 0);
                if (KL0 > 2) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2][c], yv[2], a, 0, 0, 0);
                if (KL0 > 3) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3][c], yv[3], a, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) H0[4 * jo + r] = rr_relu(a[r]);
            }
        }
    }
    MSTAMP(2)
    unsigned mw0 = 0u, mw1 = 0u;
    if constexpr (BITS != 0) {      // layer 0's mask word of this wave's panel (static register indices in both branches)
        if (ch == 0) {
#pragma unroll
            for (int t = 0; t < 4 * NBH; ++t) rr_mask_push(mw0, H0[t]);
        } else {
#pragma unroll
            for (int t = 0; t < 4 * NBH; ++t) rr_mask_push(mw0, H0[4 * NBH + t]);
        }
    }
    // ---- the hid x hid layer: this wave's panel; the last pair of blocks is finished inside the output product
    float H1[KSH];
    f32x4 acc[NBH];
    auto finish = [&](int jo, int r) __attribute__((always_inline)) {
        H1[4 * jo + r] = rr_relu(acc[jo][r]);
        if constexpr (BITS != 0) rr_mask_push(mw1, H1[4 * jo + r]);       // (called in ascending order of 4 jo + r)
    };
    panel.run(acc, cinit, H0, rs, voff, wcur, [&](int) __attribute__((always_inline)) {},
              [&](int jo, int r) __attribute__((always_inline)) { finish(jo, r); });
    MSTAMP(3)
    // ---- this half of the output layer (one block, K = the panel's units); H1's last eight values just in time
    {
        const f32x4 o = P::template block<KSH>(wo, H1, [&](int ks) __attribute__((always_inline)) {
            // (NBH == 2: the pending pair IS the panel — value ks is finished right before k-step ks reads it)
            if (ks < 8) finish(NBH - 2 + (ks >> 2), ks & 3);
        });
        *reinterpret_cast<f32x4*>(sO + (ch * NLBAC_MLP_TILE + m) * 16 + 4 * q) = o;
    }
    MSTAMP(4)
    __syncthreads();
    MSTAMP(5)
    // ---- the two halves + bias -> y; the thread that writes a row's (mean | log_std) also draws the row's action and
    //      log-probability from it (nlbac_gauss_head: gauss_fwd_kernel's arithmetic, no launch of its own)
    if (tid < NLBAC_MLP_TILE && row0 + tid < B) {
        const int row = row0 + tid;
        const float* bo = params + net.b_off[2];
        float* y = io.y + (long)row * io.y_ld;
        for (int o = 0; o < odim; ++o) y[o] = (sO[tid * 16 + o] + sO[(NLBAC_MLP_TILE + tid) * 16 + o]) + bo[o];
        if (G.eps)
            gauss_fwd_row(y, G.eps, G.scale, G.bias, G.n_u, (long)blockIdx.y * B + row, G.action, G.action_ld, G.logp);
    }
    // ---- the saved activations are the kernel's LAST instructions: nothing waits for them.  Stores share the weight
    //      loads' in-order vmcnt queue (one issued between two fragment loads makes the MFMAs behind the second load wait
    //      for the store's trip to memory), and the burst of every workgroup storing at once takes microseconds to drain:
    //      issued in front of the barrier above it stalled every tile for that long (22k of a 50k-cycle tile at 6 nets)
    if (a0row) {
#pragma unroll
        for (int jo = 0; jo < NBA; ++jo)
            *reinterpret_cast<f32x4*>(a0row + 16 * jo + 4 * q) = f32x4{H0[4 * jo], H0[4 * jo + 1], H0[4 * jo + 2], H0[4 * jo + 3]};
    }
    if (a1row) {
#pragma unroll
        for (int jo = 0; jo < NBH; ++jo)
            *reinterpret_cast<f32x4*>(a1row + 16 * jo + 4 * q) = f32x4{H1[4 * jo], H1[4 * jo + 1], H1[4 * jo + 2], H1[4 * jo + 3]};
    }
    if constexpr (BITS != 0) {
        if (io.masks && row_ok) {
            unsigned* mrow = io.masks + (long)grow * 8 + 2 * q + ch;
            mrow[0] = mw0;
            mrow[(long)B * 8] = mw1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-resident data backward of the same nets (autograd's backward through QNetwork / LyaNetwork / GaussianPolicy /
// BarrierNetwork, U/sac_cbf_clf/model.py:37-114): the launches of nlbac_mlp_bwd_data / nlbac_mlp_bwd_data_head.
//
//   dz1 = (dy W_2) * [a1 > 0]        one 16-unit block per MFMA (K = out_dim <= 16), every wave for ALL units of its rows
//   dz0 = (dz1 W_1) * [a0 > 0]       the wave's panel of W_1^T (backward RR pack), dz1 in registers as the B operands
//   dx  =  dz0 W_0                    one block over the panel's units, the two panels' halves meet in LDS
//
// Same decomposition as the forward: wave (rh, ch) = rows 16 rh .. 16 rh + 15 x panel ch.  dL/dy comes from io.dy or from
// a dy head (dy_heads.h) whose launch-wide election runs at the END of the kernel; dz rows leave in one burst behind the
// weight stream.  With nlbac_mlp_io::skinny_ws the tile's dz1 / dz0 also go through LDS once and every thread sums one
// hidden column over the tile's 32 rows — the per-tile partials of the bias / first- / last-layer gradients, the same sums
// in the same order as mlp_bwd_skinny_partial_kernel's (nlbac_mlp_bwd_weights only reduces them).
// KLO: k-steps of the top product (1: out_dim <= 4, every net the agent has; 4: out_dim <= 16).
// BITS: the ReLU gates come from the forward's mask words (nlbac_mlp_io::masks, 3 dwords per lane) instead of the saved
// activation rows (24 float4 per lane): every net of the launch has them.
template <int NBH, int KLO, int BITS>
__global__ __launch_bounds__(256, 2) void mlp_rr_bwd_kernel(const MlpLaunch L, const nlbac_dy_head H) {
    constexpr int HID = 32 * NBH, NBA = 2 * NBH, KS = HID / 4, KSH = KS / 2, LDZ = HID + 4;
    using P = RRPanel<NBH, KS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sdy = smem;                                  // [32][16] dL/dy of the tile, zero padded
    float* const sx = sdy + NLBAC_MLP_TILE * 16;              // [32][16] input rows (skinny partials) / election scratch
    float* const sO = sx + NLBAC_MLP_TILE * 16;               // [2][32][16] the panels' halves of dx
    float* const sZ1 = sO + 2 * NLBAC_MLP_TILE * 16;          // [32][LDZ] dz1 / dz0 of the tile (skinny partials only)
    float* const sZ0 = sZ1 + NLBAC_MLP_TILE * LDZ;
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), rh = wave >> 1, ch = wave & 1;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    const int q = lane >> 4, r16 = lane & 15, m = 16 * rh + r16, grow = row0 + m;
    const bool row_ok = grow < B;
    const long growc = min(grow, B - 1);
    const int idim = net.in_dim, odim = net.out_dim;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * HID;
    const bool sk = io.skinny_ws != nullptr && io.dz != nullptr;
    BSTAMP(0)

    // ---- the panel's weight stream (W_1^T), then everything else this wave reads: all of it is in flight before the
    //      first wait
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wcur = (net.rr_bwd_off + ch * (HID * HID / 2)) * 4;
    P panel;
    panel.prime(rs, voff, wcur);
    const int n0b = (int)rr_panel_l0_floats(HID) * 4;                            // bytes of one fragment block behind the panels
    const int top0 = (net.rr_bwd_off + HID * HID) * 4 + n0b, dx0 = top0 + n0b;
    f32x4 wt[KLO][NBA / 4];
#pragma unroll
    for (int k0 = 0; k0 < KLO; ++k0)
#pragma unroll
        for (int j4 = 0; j4 < NBA / 4; ++j4) wt[k0][j4] = rr_ldw(rs, voff, top0 + (k0 * (NBA / 4) + j4) * 1024);
    f32x4 a1v[NBA], a0v[NBH];
    unsigned m1w[2] = {0u, 0u}, m0w = 0u;
    if constexpr (BITS != 0) {
        const unsigned* mrow = io.masks + growc * 8 + 2 * q;
        m0w = mrow[ch];
        m1w[0] = mrow[(long)B * 8];
        m1w[1] = mrow[(long)B * 8 + 1];
    } else {
        const float* a1row = io.acts + ls + growc * HID;
#pragma unroll
        for (int j = 0; j < NBA; ++j) a1v[j] = *reinterpret_cast<const f32x4*>(a1row + 16 * j + 4 * q);
        const float* a0row = io.acts + growc * HID + 16 * NBH * ch;
#pragma unroll
        for (int j = 0; j < NBH; ++j) a0v[j] = *reinterpret_cast<const f32x4*>(a0row + 16 * j + 4 * q);
    }

    // ---- dL/dy (and, for the skinny partials, the input rows) of the tile -> LDS
    // (kind 3: nets behind the Q pairs read io.dy — but for the first of them when the head evaluates the constraint backward: cb_kind)
    const bool plain_dy = H.kind == 0 || (H.kind == 3 && (int)blockIdx.y >= 2 * H.n_prob && !(H.cb_kind && (int)blockIdx.y == 2 * H.n_prob));
    DyHeadPending pend;
    pend.v0 = pend.v1 = 0.f;
    {
        float vdy[2] = {0.f, 0.f}, vx0[2] = {0.f, 0.f}, vx1[2] = {0.f, 0.f};
        const bool has_x1 = io.x1 != nullptr && io.x1_dim > 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, r = idx >> 4, c = idx & 15;
            const long row = min(row0 + r, B - 1);
            if (plain_dy) vdy[it] = io.dy[row * io.dy_ld + min(c, odim - 1)];
            if (sk) {
                vx0[it] = io.x0[row * io.x0_ld + min(c, io.x0_dim - 1)];
                if (has_x1) vx1[it] = io.x1[row * io.x1_ld + min(max(c - io.x0_dim, 0), io.x1_dim - 1)];
            }
        }
        if (!plain_dy) dy_head_rows(H, blockIdx.y, row0, B, sdy, pend);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, r = idx >> 4, c = idx & 15, row = row0 + r;
            if (plain_dy) sdy[idx] = (row < B && c < odim) ? vdy[it] : 0.f;
            if (sk) sx[idx] = (row < B && c < idim) ? (c < io.x0_dim ? vx0[it] : vx1[it]) : 0.f;
        }
    }
    __syncthreads();
    BSTAMP(1)

    // ---- top layer for all units of the wave's rows: dz1^T[unit][row] = sum_o W_2[o][unit] dy[row][o], gated by a1
    float H1[KS];
    {
        float yv[KLO];
#pragma unroll
        for (int k0 = 0; k0 < KLO; ++k0) yv[k0] = sdy[m * 16 + 4 * k0 + q];
#pragma unroll
        for (int j4 = 0; j4 < NBA / 4; ++j4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int jo = 4 * j4 + c;
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0][j4][c], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int k0 = 1; k0 < KLO; ++k0) a = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[k0][j4][c], yv[k0], a, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (BITS != 0) H1[4 * jo + r] = rr_mask_gate<4 * NBH>(m1w[jo / NBH], 4 * (jo % NBH) + r, a[r]);
                    else H1[4 * jo + r] = (a1v[jo][r] > 0.f) ? a[r] : 0.f;
                }
            }
        }
    }
    BSTAMP(2)
    // layer 0's transposed fragments for dx: the panel's blocks (requested here, behind the top layer's operands)
    f32x4 wx[NBH];
    if (io.dx) {
#pragma unroll
        for (int j = 0; j < NBH; ++j) wx[j] = rr_ldw(rs, voff, dx0 + (NBH * ch + j) * 1024);
    }

    // ---- the hid x hid layer: this wave's panel of dz0, gated by a0 as its blocks finish
    float Hz[KSH];
    f32x4 acc[NBH];
    f32x4 czero[NBH];
#pragma unroll
    for (int jo = 0; jo < NBH; ++jo) czero[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto finish = [&](int jo, int r) __attribute__((always_inline)) {
        if constexpr (BITS != 0) Hz[4 * jo + r] = rr_mask_gate<4 * NBH>(m0w, 4 * jo + r, acc[jo][r]);
        else Hz[4 * jo + r] = (a0v[jo][r] > 0.f) ? acc[jo][r] : 0.f;
    };
    panel.run(acc, czero, H1, rs, voff, wcur, [&](int) __attribute__((always_inline)) {},
              [&](int jo, int r) __attribute__((always_inline)) { finish(jo, r); });
    BSTAMP(3)
    // ---- this half of dx (one block over the panel's units); the last pair of dz0 blocks is finished just in time
    if (io.dx) {
        float wo[KSH];
#pragma unroll
        for (int j = 0; j < NBH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[4 * j + r] = wx[j][r];
        const f32x4 o = P::template block<KSH>(wo, Hz, [&](int ks) __attribute__((always_inline)) {
            if (ks < 8) finish(NBH - 2 + (ks >> 2), ks & 3);
        });
        *reinterpret_cast<f32x4*>(sO + (ch * NLBAC_MLP_TILE + m) * 16 + 4 * q) = o;
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) finish(NBH - 2 + (ks >> 2), ks & 3);
    }
    BSTAMP(4)
    // ---- skinny partials: the tile's dz through LDS; the last layer's activations of this thread's column, requested
    //      before the barrier
    float av[NLBAC_MLP_TILE];
    const int kcol = min(tid, HID - 1);
    if (sk) {
        const float* a1 = io.acts + ls + kcol;
#pragma unroll
        for (int mm = 0; mm < NLBAC_MLP_TILE; ++mm) av[mm] = a1[(long)min(row0 + mm, B - 1) * HID];
    }
    // ---- dz rows, this wave's panel of both layers: behind the last weight load (stores share the loads' in-order vmcnt
    //      queue) and behind the column loads above, and NOTHING WAITS FOR THEM — the barrier below orders LDS only: the
    //      burst of every workgroup storing at once takes microseconds to drain (behind a __syncthreads() it stalled the
    //      whole tile for that long: 20k of a 60k-cycle tile at 3 nets); now it drains under the partial sums
    if (io.dz && row_ok) {
        float* z1 = io.dz + ls + (long)grow * HID + 16 * NBH * ch;
        float* z0p = io.dz + (long)grow * HID + 16 * NBH * ch;
        if (ch == 0) {
#pragma unroll
            for (int j = 0; j < NBH; ++j)
                *reinterpret_cast<f32x4*>(z1 + 16 * j + 4 * q) = f32x4{H1[4 * j], H1[4 * j + 1], H1[4 * j + 2], H1[4 * j + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < NBH; ++j)
                *reinterpret_cast<f32x4*>(z1 + 16 * j + 4 * q) =
                    f32x4{H1[4 * (NBH + j)], H1[4 * (NBH + j) + 1], H1[4 * (NBH + j) + 2], H1[4 * (NBH + j) + 3]};
        }
#pragma unroll
        for (int j = 0; j < NBH; ++j)
            *reinterpret_cast<f32x4*>(z0p + 16 * j + 4 * q) = f32x4{Hz[4 * j], Hz[4 * j + 1], Hz[4 * j + 2], Hz[4 * j + 3]};
    }
    if (sk) {
        float* s1 = sZ1 + m * LDZ + 16 * NBH * ch + 4 * q;
        float* s0 = sZ0 + m * LDZ + 16 * NBH * ch + 4 * q;
        if (ch == 0) {
#pragma unroll
            for (int j = 0; j < NBH; ++j)
                *reinterpret_cast<f32x4*>(s1 + 16 * j) = f32x4{H1[4 * j], H1[4 * j + 1], H1[4 * j + 2], H1[4 * j + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < NBH; ++j)
                *reinterpret_cast<f32x4*>(s1 + 16 * j) =
                    f32x4{H1[4 * (NBH + j)], H1[4 * (NBH + j) + 1], H1[4 * (NBH + j) + 2], H1[4 * (NBH + j) + 3]};
        }
#pragma unroll
        for (int j = 0; j < NBH; ++j)
            *reinterpret_cast<f32x4*>(s0 + 16 * j) = f32x4{Hz[4 * j], Hz[4 * j + 1], Hz[4 * j + 2], Hz[4 * j + 3]};
    }
    lds_barrier();
    BSTAMP(5)
    if (io.dx) {   // the two halves -> dx (columns below dx_first are not wanted: nlbac_mlp_io)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, r = idx >> 4, c = idx & 15;
            if (row0 + r < B && c >= io.dx_first && c < idim)
                io.dx[(long)(row0 + r) * io.dx_ld + c] = sO[idx] + sO[NLBAC_MLP_TILE * 16 + idx];
        }
    }
    if (sk) {      // thread = hidden column k; the sums and their order are mlp_bwd_skinny_partial_kernel's (row after row,
                   // fused multiply-adds), one set per NLBAC_SK_CHUNK = 16 rows: the column's values in registers, the rows'
                   // x / dy as broadcast float4 reads
        const bool live = tid < HID;
#pragma unroll
        for (int hh = 0; hh < NLBAC_MLP_TILE / NLBAC_SK_CHUNK; ++hh) {
            constexpr int CH = NLBAC_SK_CHUNK;
            float* w = io.skinny_ws + ((long)blockIdx.x * (NLBAC_MLP_TILE / CH) + hh) * (2 + idim + odim + 1) * 256 + tid;
            const float* z1p = sZ1 + hh * CH * LDZ, * z0p = sZ0 + hh * CH * LDZ;
            const float* xp = sx + hh * CH * 16, * dyp = sdy + hh * CH * 16;
            float z0[CH];
            float b0 = 0.f, b1 = 0.f;
#pragma unroll
            for (int mm = 0; mm < CH; ++mm) { z0[mm] = z0p[mm * LDZ + kcol]; b1 += z1p[mm * LDZ + kcol]; }
#pragma unroll
            for (int mm = 0; mm < CH; ++mm) b0 += z0[mm];
            w[0] = live ? b0 : 0.f;
            w[256] = live ? b1 : 0.f;
            for (int i0 = 0; i0 < idim; i0 += 4) {    // dW_0[k][i] = sum_m dz0[m][k] x[m][i]
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mm = 0; mm < CH; ++mm) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + mm * 16 + i0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] = __builtin_fmaf(z0[mm], xv[c], a[c]);
                }
                // (all four chains are wanted as they stand, interleaved: behind the `i0 + c < idim` guards below the
                //  compiler sinks each into its own block — serial chains of dependent FMAs instead of one pass)
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (i0 + c < idim) w[(long)(2 + i0 + c) * 256] = live ? a[c] : 0.f;
            }
            for (int o0 = 0; o0 < odim; o0 += 4) {    // dW_2[o][k] = sum_m dy[m][o] a1[m][k]
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mm = 0; mm < CH; ++mm) {
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(dyp + mm * 16 + o0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] = __builtin_fmaf(dv[c], av[hh * CH + mm], a[c]);
                }
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (o0 + c < odim) w[(long)(2 + idim + o0 + c) * 256] = live ? a[c] : 0.f;
            }
            float bl = 0.f;
            if (tid < 16)
                for (int mm = 0; mm < CH; ++mm) bl += dyp[mm * 16 + tid];
            w[(long)(2 + idim + odim) * 256] = bl;
        }
    }
    BSTAMP(6)
    // ---- the dy head's batch sums: published / finished here, off the tile's critical path
    if (!plain_dy && H.kind != 1) {
        __syncthreads();                           // (sx is the election's scratch: every read of it above is done)
        dy_head_finish(H, blockIdx.y, row0, gridDim.x, sx, gridDim.y, pend);
    }
    // ---- batch sums an earlier launch's head left to this one (nlbac_dy_head::finish)
    dy_head_jobs(H, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, sx);
}

// ---------------------------------------------------------------------------------------------------------------------
static bool mrr_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_MLP_RR"); return !(e && e[0] == '0'); }();
    return on;
}

bool nlbac_mlp_rr_eligible(const nlbac_mlp* nets, int n_nets) {
    if (!mrr_enabled()) return false;
    for (int i = 0; i < n_nets; ++i) {
        const nlbac_mlp& n = nets[i];
        if (n.rr_kind != RR_KIND_PANEL || n.rr_fwd_off < 0 || n.hid != nets[0].hid) return false;
        if (n.hid != 64 && n.hid != 128 && n.hid != 256) return false;
        if (n.in_dim > MRR_MAX_IN || n.out_dim > 16) return false;
    }
    return true;
}

int nlbac_mlp_rr_fwd_launch(const MlpLaunch& L, int n_nets, const nlbac_gauss_head& G, const char* who, hipStream_t s) {
    if (!nlbac_mlp_rr_eligible(L.net, n_nets)) return 1;
    {
        const int rq = nlbac_mlp_rrq_fwd_launch(L, n_nets, G, who, s);      // (hid 128 / 256: the quarter-panel kernels)
        if (rq <= 0) return rq;
    }
    const int hid = L.net[0].hid;
    bool bits = false;
    for (int i = 0; i < n_nets; ++i) bits = bits || L.io[i].masks != nullptr;
    const size_t lds = (size_t)(2 * NLBAC_MLP_TILE * 16) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.B, NLBAC_MLP_TILE), n_nets);
#define MRR_FWD(NBH_)                                                                                            \
    if (bits) hipLaunchKernelGGL((mlp_rr_fwd_kernel<NBH_, 1>), grid, dim3(256), lds, s, L, G);                   \
    else hipLaunchKernelGGL((mlp_rr_fwd_kernel<NBH_, 0>), grid, dim3(256), lds, s, L, G);
    switch (hid) {
        case 64: MRR_FWD(2) break;
        case 128: MRR_FWD(4) break;
        default: MRR_FWD(8)
    }
#undef MRR_FWD
    NLBAC_CHECK_LAUNCH(who);
    return 0;
}

// data backward: 0 = launched, 1 = not these nets' kernel (the LDS-tiled one takes the launch), < 0 = error
static bool mrr_bwd_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_MLP_RR_BWD"); return !(e && e[0] == '0'); }();
    return on;
}

// Every launch of a net of this shape — forward and data backward — is served by the register-resident kernels (so its
// 32x32x2 fragment packs are never read: nlbac_mlp_pack_layout then leaves them out, and the optimiser has half as many
// fragment slots to refresh per weight).  Shape and the process-wide switches only: the per-launch conditions of
// nlbac_mlp_rr_eligible beyond them compare the nets of ONE launch (equal widths), and a launch that mixes widths is
// refused by the LDS-tiled launcher for such a net instead of reading packs that do not exist.
bool nlbac_mlp_rr_serves_shape(int n_layers, int in_dim, int hid, int out_dim) {
    return mrr_enabled() && mrr_bwd_enabled() && rr_kind_of(n_layers, hid) == RR_KIND_PANEL &&
           (hid == 64 || hid == 128 || hid == 256) && in_dim <= MRR_MAX_IN && out_dim <= 16;
}

int nlbac_mlp_rr_bwd_launch(const MlpLaunch& L, int n_nets, const nlbac_dy_head& H, const char* who, hipStream_t s) {
    if (!mrr_bwd_enabled() || !nlbac_mlp_rr_eligible(L.net, n_nets)) return 1;
    {
        const int rq = nlbac_mlp_rrq_bwd_launch(L, n_nets, H, who, s);
        if (rq <= 0) return rq;
    }
    const int hid = L.net[0].hid;
    bool sk = false, wide_out = false;
    int n_bits = 0;
    for (int i = 0; i < n_nets; ++i) {
        sk = sk || (L.io[i].skinny_ws != nullptr && L.io[i].dz != nullptr);
        wide_out = wide_out || L.net[i].out_dim > 4;
        n_bits += L.io[i].masks != nullptr;
    }
    NLBAC_REQUIRE(n_bits == 0 || n_bits == n_nets, "%s: ReLU mask words (nlbac_mlp_io::masks) for all nets of a launch or for none", who);
    const size_t lds = (size_t)(4 * NLBAC_MLP_TILE * 16 + (sk ? 2 * NLBAC_MLP_TILE * (hid + 4) : 0)) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.B, NLBAC_MLP_TILE), n_nets);
#define MRR_BWD2(NBH_, KLO_)                                                                                       \
    if (n_bits) hipLaunchKernelGGL((mlp_rr_bwd_kernel<NBH_, KLO_, 1>), grid, dim3(256), lds, s, L, H);              \
    else hipLaunchKernelGGL((mlp_rr_bwd_kernel<NBH_, KLO_, 0>), grid, dim3(256), lds, s, L, H);
#define MRR_BWD(NBH_)                                                                                              \
    if (wide_out) { MRR_BWD2(NBH_, 4) } else { MRR_BWD2(NBH_, 1) }
    switch (hid) {
        case 64: MRR_BWD(2) break;
        case 128: MRR_BWD(4) break;
        default: MRR_BWD(8)
    }
#undef MRR_BWD
#undef MRR_BWD2
    NLBAC_CHECK_LAUNCH(who);
    return 0;
}

// 1 when both register-resident kernels take launches of these nets, i.e. when nlbac_mlp_io::masks may replace acts
extern "C" int nlbac_mlp_masks_ok(const nlbac_mlp* nets, int n_nets) {
    return (n_nets >= 1 && mrr_bwd_enabled() && nlbac_mlp_rr_eligible(nets, n_nets)) ? 1 : 0;
}
