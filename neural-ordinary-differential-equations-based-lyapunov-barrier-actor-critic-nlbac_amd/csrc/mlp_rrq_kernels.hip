// Quarter-panel form of the register-resident head kernels (QNetwork, LyaNetwork, GaussianPolicy, BarrierNetwork:
// in -> hid -> hid -> out with hid = 128 / 256; U/sac_cbf_clf/model.py:37-114): the launches of nlbac_mlp_fwd[_gauss] /
// nlbac_mlp_bwd_data[_head] at those widths.
//
// mlp_rr_kernels.hip gives a wave 16 rows x HALF of the hid x hid layer (~600 MFMAs) and a workgroup 32 rows: at B = 4096
// a launch is 128 workgroups per net — 384 for three nets on 256 CUs, i.e. two of them on half of the CUs and one on the
// rest, each a serial chain (prologue -> layer 0 -> panel -> output -> stores) with one wave per SIMD and nothing to overlap
// it with: a three-net launch took two full chains (25.7 us forward, 26.5 backward) for 13 us of matrix-pipe time.
// Here a workgroup is 16 rows and its four waves are the four QUARTERS of the layer (64 output units: ~280 MFMAs each): a
// net is 256 workgroups, every SIMD of the chip gets one wave per net, and the kernels are built for three waves per SIMD
// (<= 168 VGPRs: a weight queue of 4 float4) so that a three-net launch is resident at once and the waves of different
// nets fill each other's prologues, barriers and store bursts.  What the waves of a workgroup would compute redundantly
// (layer 0 forward, the top layer backward: all hid units are every wave's B operands) is computed a quarter each and
// exchanged through one LDS tile [16 rows][hid]; the same tile serves the backward's bias-gradient column sums.
//
// The panel packs are mlp_rr_kernels.hip's: a panel's stream is group-major, so a quarter is a contiguous half of it.
// ReLU mask words (nlbac_mlp_io::masks): one uint16 per (layer, row, lane quarter q, layer quarter cq) at
// ((layer * B + row) * 4 + q) * 4 + cq — value t = 4 j + r (block j of the quarter, register r) at bit 4 NBQ - 1 - t.
#include "mlp_launch.h"
#include "rr_device.h"
#include "auglag_device.h"
#include <cstdlib>

#define QT 16               /* rows per workgroup */
#define MRQ_MAX_IN 15       /* in_dim + the bias column <= 16: four k-steps of layer 0 */

#ifdef RR_TIMING      // ablation build: wave 0 of workgroup 0 stamps the shader clock (as int64) behind the first net's y / dz
#define QFSTAMP(k_) if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) reinterpret_cast<long long*>(io.y + (long)B * io.y_ld)[k_] = (long long)__builtin_readcyclecounter();
#define QBSTAMP(k_) if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) reinterpret_cast<long long*>(io.dz + 2 * ls)[k_] = (long long)__builtin_readcyclecounter();
#else
#define QFSTAMP(k_)
#define QBSTAMP(k_)
#endif

// the component of a fragment float4 that holds block j of quarter cq (NBQ = 4: the float4 IS the quarter's four blocks;
// NBQ = 2: two quarters share a float4)
template <int NBQ>
__device__ __forceinline__ float rrq_frag(const f32x4& w, int cq, int j) {
    if constexpr (NBQ == 4) return w[j];
    else return (cq & 1) ? w[2 + j] : w[j];
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <int NBQ, int BITS>
__global__ __launch_bounds__(256, 3) void mlp_rrq_fwd_kernel(const MlpLaunch L, const nlbac_gauss_head G) {
    constexpr int HID = 64 * NBQ, NBA = 4 * NBQ, KS = HID / 4, KSQ = KS / 4, LDH = HID + 4;
    using P = RRPanel<NBQ, KS, 4>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sH = smem;                               // [16][LDH] layer 0's activations: the quarters' exchange
    float* const sO = sH + QT * LDH;                      // [4][16][16] the quarters' parts of the output layer
    const nlbac_mlp& net = L.net[blockIdx.y];
    const nlbac_mlp_io& io = L.io[blockIdx.y];
    const int B = L.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int cq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * QT;
    const int q = lane >> 4, r16 = lane & 15, grow = row0 + r16;
    const bool row_ok = grow < B;
    const int idim = net.in_dim, odim = net.out_dim;
    const int KL0 = (idim + 4) >> 2;                      // k-steps of layer 0 over [x | 1] (1..4)
    const float* const params = net.params;
    const int ub = 16 * NBQ * cq;                         // first unit of this wave's quarter

    QFSTAMP(0)
    // ---- constraint head (the end of the kernel): its operands are requested now — they do not depend on this forward —
    //      so that their trip to memory is not part of the tile's tail: (row, hazard) per thread, the CLF rows behind them
    const bool cf_on = G.cf_kind == 1 && (int)blockIdx.y == G.cf_net;      // (uniform per workgroup)
    float cfv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cf_on) {
        if (tid < QT * 7) {
            const int r = tid / 7, h = tid - r * 7, i = min(row0 + r, B - 1);
            cfv[0] = G.cf_ps[i * 2]; cfv[1] = G.cf_ps[i * 2 + 1];
            cfv[2] = G.cf_ps_next[i * 2]; cfv[3] = G.cf_ps_next[i * 2 + 1];
            cfv[4] = G.cf_ps_next[(long)(B + i) * 2]; cfv[5] = G.cf_ps_next[(long)(B + i) * 2 + 1];
            cfv[6] = G.cf_hazards[h * 2]; cfv[7] = G.cf_hazards[h * 2 + 1];
        } else if (tid >= 128 && tid < 128 + QT) {
            cfv[0] = G.cf_V[min(row0 + tid - 128, B - 1)];
        }
    }
    // ---- the quarter's weight stream: panel cq / 2, its first or second half
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wcur = (net.rr_fwd_off + (cq >> 1) * (HID * HID / 2) + (cq & 1) * (HID * HID / 4)) * 4;
    P panel;
    panel.prime(rs, voff, wcur);
    // ---- layer 0's fragments of the quarter's blocks ([W_0 | b_0] over k-steps of [x | 1])
    f32x4 w0[4];
    {
        const int l0 = (net.rr_bwd_off + HID * HID) * 4;
#pragma unroll
        for (int k0 = 0; k0 < 4; ++k0)
            w0[k0] = (k0 < KL0) ? rr_ldw(rs, voff, l0 + (k0 * (NBA / 4) + (NBQ * cq) / 4) * 1024) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- this lane's inputs: component 4 k0 + q of its row, a 1 in the bias column behind the last one
    float yv[4];
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) {
        const int c = 4 * k0 + q;
        float v = (c == idim) ? 1.f : 0.f;
        if (row_ok && c < idim) v = (c < io.x0_dim) ? io.x0[(long)grow * io.x0_ld + c] : io.x1[(long)grow * io.x1_ld + (c - io.x0_dim)];
        yv[k0] = v;
    }
    // ---- this quarter of the output layer's A fragments and of the hidden layer's biases
    float wo[KSQ];
    f32x4 cinit[NBQ];
    {
        const float* wrow = params + net.w_off[2] + (long)min(r16, odim - 1) * HID + ub;
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 16 * j + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[4 * j + r] = (r16 < odim) ? v[r] : 0.f;
            cinit[j] = *reinterpret_cast<const f32x4*>(params + net.b_off[1] + ub + 16 * j + 4 * q);
        }
    }
    QFSTAMP(1)

    // ---- layer 0, this quarter's blocks (bias folded into the product), ReLU; the quarters meet in LDS
    float H0q[4 * NBQ];
    unsigned mw0 = 0u, mw1 = 0u;
#pragma unroll
    for (int j = 0; j < NBQ; ++j) {
        f32x4 a = __builtin_amdgcn_mfma_f32_16x16x4f32(rrq_frag<NBQ>(w0[0], cq, j), yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (KL0 > 1) a = __builtin_amdgcn_mfma_f32_16x16x4f32(rrq_frag<NBQ>(w0[1], cq, j), yv[1], a, 0, 0, 0);
        if (KL0 > 2) a = __builtin_amdgcn_mfma_f32_16x16x4f32(rrq_frag<NBQ>(w0[2], cq, j), yv[2], a, 0, 0, 0);
        if (KL0 > 3) a = __builtin_amdgcn_mfma_f32_16x16x4f32(rrq_frag<NBQ>(w0[3], cq, j), yv[3], a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            H0q[4 * j + r] = rr_relu(a[r]);
            if constexpr (BITS != 0) rr_mask_push(mw0, H0q[4 * j + r]);
        }
        *reinterpret_cast<f32x4*>(sH + r16 * LDH + ub + 16 * j + 4 * q) = f32x4{H0q[4 * j], H0q[4 * j + 1], H0q[4 * j + 2], H0q[4 * j + 3]};
    }
    lds_barrier();
    float H0[KS];
#pragma unroll
    for (int jo = 0; jo < NBA; ++jo) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sH + r16 * LDH + 16 * jo + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) H0[4 * jo + r] = v[r];
    }
    QFSTAMP(2)
    // ---- the hid x hid layer: this wave's quarter; the last pair of blocks is finished inside the output product
    float H1[KSQ];
    f32x4 acc[NBQ];
    auto finish = [&](int jo, int r) __attribute__((always_inline)) {
        H1[4 * jo + r] = rr_relu(acc[jo][r]);
        if constexpr (BITS != 0) rr_mask_push(mw1, H1[4 * jo + r]);       // (called in ascending order of 4 jo + r)
    };
    panel.run(acc, cinit, H0, rs, voff, wcur, [&](int) __attribute__((always_inline)) {},
              [&](int jo, int r) __attribute__((always_inline)) { finish(jo, r); });
    QFSTAMP(3)
    // ---- this quarter of the output layer (one block, K = the quarter's units); the pending pair just in time
    {
        const f32x4 o = P::template block<KSQ>(wo, H1, [&](int ks) __attribute__((always_inline)) {
            if (ks < 8) finish(NBQ - 2 + (ks >> 2), ks & 3);
        });
        *reinterpret_cast<f32x4*>(sO + (cq * QT + r16) * 16 + 4 * q) = o;
    }
    lds_barrier();
    QFSTAMP(4)
    // ---- the four quarters + bias -> y; the thread that writes a row's (mean | log_std) also draws the row's action and
    //      log-probability from it (nlbac_gauss_head: gauss_fwd_kernel's arithmetic, no launch of its own)
    if (tid < QT && row0 + tid < B) {
        const int row = row0 + tid;
        const float* bo = params + net.b_off[2];
        float* y = io.y + (long)row * io.y_ld;
        for (int o = 0; o < odim; ++o)
            y[o] = ((sO[tid * 16 + o] + sO[(QT + tid) * 16 + o]) + (sO[(2 * QT + tid) * 16 + o] + sO[(3 * QT + tid) * 16 + o])) + bo[o];
        if (G.eps)
            gauss_fwd_row(y, G.eps, G.scale, G.bias, G.n_u, (long)blockIdx.y * B + row, G.action, G.action_ld, G.logp);
        if (G.cf_kind) sH[tid] = y[0];        // (V(p(x')) of the row, for the constraint head below; sH is free by now)
    }
    // ---- what the backward needs, the kernel's last instructions (nothing waits for the stores): this quarter of the
    //      activation rows of both layers, and / or its mask bits
    if (io.acts && row_ok) {
        const long ls = io.acts_ls ? io.acts_ls : (long)B * HID;
        float* a0 = io.acts + (long)grow * HID + ub + 4 * q;
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
            *reinterpret_cast<f32x4*>(a0 + 16 * j) = f32x4{H0q[4 * j], H0q[4 * j + 1], H0q[4 * j + 2], H0q[4 * j + 3]};
            *reinterpret_cast<f32x4*>(a0 + ls + 16 * j) = f32x4{H1[4 * j], H1[4 * j + 1], H1[4 * j + 2], H1[4 * j + 3]};
        }
    }
    if constexpr (BITS != 0) {
        if (io.masks && row_ok) {
            unsigned short* mrow = reinterpret_cast<unsigned short*>(io.masks) + ((long)grow * 4 + q) * 4 + cq;
            mrow[0] = (unsigned short)mw0;
            mrow[(long)B * 16] = (unsigned short)mw1;
        }
    }
    QFSTAMP(5)
    // ---- constraint head (nlbac_gauss_head::cf_kind 1: unicycle_constraints_fwd_kernel's row arithmetic, agent_kernels.hip):
    //      the CBF terms of both controllers and the CLF term of the tile's rows — (row, hazard) per thread —, their
    //      relu-filtered column sums over the tile, published; the workgroup elected last runs the augmented-Lagrangian step
    if (cf_on) {
        constexpr int NH = 7, NC = 2 * NH + 1;
        float* const sVn = sH;                                // [16]
        float* const sT = sH + 16;                            // [16][16]: relu(term) of (row, column < NC)
        lds_barrier();
        if (tid < QT * NH) {
            const int r = tid / NH, h = tid - r * NH, i = min(row0 + r, B - 1);
            const float p0 = cfv[0], p1 = cfv[1], n0 = cfv[2], n1 = cfv[3], b0 = cfv[4], b1 = cfv[5], hx = cfv[6], hy = cfv[7];
            const float hs = 0.5f * (((p0 - hx) * (p0 - hx) + (p1 - hy) * (p1 - hy)) - G.cf_r2);
            const float hn = 0.5f * (((n0 - hx) * (n0 - hx) + (n1 - hy) * (n1 - hy)) - G.cf_r2);
            const float hb = 0.5f * (((b0 - hx) * (b0 - hx) + (b1 - hy) * (b1 - hy)) - G.cf_r2);
            const float t = -((hn - hs) / G.cf_dt) - G.cf_gamma_b * hs;
            const float tb = -((hb - hs) / G.cf_dt) - G.cf_gamma_b * hs;
            const bool ok = row0 + r < B;
            if (ok) {
                G.cf_matr[(long)i * (NH + 1) + h] = t;
                G.cf_bmatr[(long)i * NH + h] = tb;
            }
            sT[r * 16 + h] = (ok && t > 0.f) ? t : 0.f;
            sT[r * 16 + NH + 1 + h] = (ok && tb > 0.f) ? tb : 0.f;
        } else if (tid >= 128 && tid < 128 + QT) {
            const int r = tid - 128, i = min(row0 + r, B - 1);
            const float vv = cfv[0];
            const float lya = ((sVn[r] - vv) / G.cf_dt) + G.cf_gamma_l * vv;
            const bool ok = row0 + r < B;
            if (ok) G.cf_matr[(long)i * (NH + 1) + NH] = lya;
            sT[r * 16 + NH] = (ok && lya > 0.f) ? lya : 0.f;
        }
        lds_barrier();
        float mine = 0.f;
        if (tid < NC)
            for (int r = 0; r < QT; ++r) mine += sT[r * 16 + tid];
        const unsigned tile = blockIdx.x, n_tiles = gridDim.x;
        const AuglagArgs A = {G.cf_n_cbf, G.cf_n_clf, G.cf_batch_size, G.cf_do_lambda_update, G.cf_do_backup_lambda_update,
                              G.cf_ratio_mode, G.cf_backup_mode, G.cf_lam_lo, G.cf_lam_hi};
        if (G.cf_defer) {
            // no election: the tile's column sums go out as they are; the workgroups that need the step's coefficients
            // sum them themselves (nlbac_dy_head::cb_defer), a later launch commits the step (nlbac_head_sums kind 4)
            if (tid < NC) G.cf_partials[(long)tile * NC + tid] = mine;
            if (tile == 0 && tid == 0) G.cf_tiles[0] = n_tiles;
        } else if (publish_and_elect_grouped_lanes(G.cf_partials + (long)tile * NC, mine, NC, G.cf_tickets, tile, n_tiles)) {
            // the elected workgroup: the scalars block into LDS (the layer-0 exchange tile is free), the tiles' sums, then
            // the augmented-Lagrangian bookkeeping (auglag_device.h)
            auglag_from_tiles<NC, true>(G.cf_partials, n_tiles, A, G.cf_sc, sH, sH + NLBAC_SC_SIZE_ENUM, true);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// data backward
//   dz1 = (dy W_2) * [a1 > 0]        this quarter's blocks (K = out_dim <= 16); the quarters meet in LDS
//   dz0 = (dz1 W_1) * [a0 > 0]       the wave's quarter of W_1^T (backward RR pack), dz1 in registers as the B operands
//   dx  =  dz0 W_0                    one block over the quarter's units, the four parts meet in LDS
// dL/dy from io.dy or a dy head (dy_heads.h; its launch-wide election at the END of the kernel).  With
// nlbac_mlp_io::skinny_ws every thread also sums one hidden column over the tile's 16 rows: the per-16-row partials of the
// bias / first- / last-layer gradients (NLBAC_SK_CHUNK), mlp_bwd_skinny_partial_kernel's sums in its order.
// KLO: k-steps of the top product (1: out_dim <= 4; 4: out_dim <= 16).  BITS: gates from the forward's mask bits.
// ---------------------------------------------------------------------------------------------------------------------
template <int NBQ, int KLO, int BITS>
__global__ __launch_bounds__(256, 3) void mlp_rrq_bwd_kernel(const MlpLaunch L, const nlbac_dy_head H) {
    constexpr int HID = 64 * NBQ, NBA = 4 * NBQ, KS = HID / 4, KSQ = KS / 4, LDH = HID + 4;
    using P = RRPanel<NBQ, KS, 4>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sdy = smem;                              // [16][16] dL/dy of the tile, zero padded
    float* const sx = sdy + QT * 16;                      // [16][16] input rows (skinny partials) / election scratch
    float* const sO = sx + QT * 16;                       // [4][16][16] the quarters' parts of dx
    float* const sZ1 = sO + 4 * QT * 16;                  // [16][LDH] dz1: the quarters' exchange (and its column sums)
    float* const sZ0 = sZ1 + QT * LDH;                    // [16][LDH] dz0 (skinny partials only)
    // which net this workgroup serves.  With nlbac_dy_head::cb_defer the net behind the Q pairs goes FIRST: its workgroups
    // carry the private augmented-Lagrangian step (~4 us more); dispatched first they keep their slots longer and the
    // launch's second round lands on the other slots — last, they would end the launch that much later
    const int by = H.cb_defer ? (int)((blockIdx.y + 2u * (unsigned)H.n_prob) % gridDim.y) : (int)blockIdx.y;
    const nlbac_mlp& net = L.net[by];
    const nlbac_mlp_io& io = L.io[by];
    const int B = L.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int cq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * QT;
    const int q = lane >> 4, r16 = lane & 15, grow = row0 + r16;
    const bool row_ok = grow < B;
    const long growc = min(grow, B - 1);
    const int idim = net.in_dim, odim = net.out_dim;
    const long ls = io.acts_ls ? io.acts_ls : (long)B * HID;
    const bool sk = io.skinny_ws != nullptr && io.dz != nullptr;
    const bool skip0 = sk && io.dz_first > 0;
    const int ub = 16 * NBQ * cq;
    QBSTAMP(0)

    // ---- the quarter's weight stream (W_1^T), then everything else this wave reads
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wcur = (net.rr_bwd_off + (cq >> 1) * (HID * HID / 2) + (cq & 1) * (HID * HID / 4)) * 4;
    P panel;
    panel.prime(rs, voff, wcur);
    const int n0b = (int)rr_panel_l0_floats(HID) * 4;                            // bytes of one fragment block behind the panels
    const int top0 = (net.rr_bwd_off + HID * HID) * 4 + n0b, dx0 = top0 + n0b;
    f32x4 wt[KLO];
#pragma unroll
    for (int k0 = 0; k0 < KLO; ++k0) wt[k0] = rr_ldw(rs, voff, top0 + (k0 * (NBA / 4) + (NBQ * cq) / 4) * 1024);
    f32x4 a1v[NBQ], a0v[NBQ];
    unsigned m1w = 0u, m0w = 0u;
    if constexpr (BITS != 0) {
        const unsigned short* mrow = reinterpret_cast<const unsigned short*>(io.masks) + (growc * 4 + q) * 4 + cq;
        m0w = mrow[0];
        m1w = mrow[(long)B * 16];
    } else {
        const float* a0row = io.acts + growc * HID + ub + 4 * q;
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
            a0v[j] = *reinterpret_cast<const f32x4*>(a0row + 16 * j);
            a1v[j] = *reinterpret_cast<const f32x4*>(a0row + ls + 16 * j);
        }
    }
    f32x4 wx[NBQ];          // layer 0's transposed fragments for dx: the quarter's blocks
    if (io.dx) {
#pragma unroll
        for (int j = 0; j < NBQ; ++j) wx[j] = rr_ldw(rs, voff, dx0 + (NBQ * cq + j) * 1024);
    }

    // ---- dL/dy (and, for the skinny partials, the input rows) of the tile -> LDS
    // (kind 3: nets behind the Q pairs read io.dy — but for the first of them when the head evaluates the constraint backward: cb_kind)
    const bool plain_dy = H.kind == 0 || (H.kind == 3 && by >= 2 * H.n_prob && !(H.cb_kind && by == 2 * H.n_prob));
    DyHeadPending pend;
    pend.v0 = pend.v1 = 0.f;
    {
        const int r = tid >> 4, c = tid & 15;
        const long row = min(row0 + r, B - 1);
        float vdy = 0.f, vx0 = 0.f, vx1 = 0.f;
        if (plain_dy) vdy = io.dy[row * io.dy_ld + min(c, odim - 1)];
        if (sk) {
            vx0 = io.x0[row * io.x0_ld + min(c, io.x0_dim - 1)];
            if (io.x1 != nullptr && io.x1_dim > 0) vx1 = io.x1[row * io.x1_ld + min(max(c - io.x0_dim, 0), io.x1_dim - 1)];
        }
        if (!plain_dy) dy_head_rows<QT>(H, by, row0, B, sdy, pend);
        if (plain_dy) sdy[tid] = (row0 + r < B && c < odim) ? vdy : 0.f;
        if (sk) sx[tid] = (row0 + r < B && c < idim) ? (c < io.x0_dim ? vx0 : vx1) : 0.f;
    }
    lds_barrier();
    QBSTAMP(1)

    // ---- top layer, this quarter's blocks: dz1^T[unit][row] = sum_o W_2[o][unit] dy[row][o], gated by a1; the quarters
    //      meet in LDS
    float H1q[4 * NBQ];
    {
        float yv[KLO];
#pragma unroll
        for (int k0 = 0; k0 < KLO; ++k0) yv[k0] = sdy[r16 * 16 + 4 * k0 + q];
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
            f32x4 a = __builtin_amdgcn_mfma_f32_16x16x4f32(rrq_frag<NBQ>(wt[0], cq, j), yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int k0 = 1; k0 < KLO; ++k0) a = __builtin_amdgcn_mfma_f32_16x16x4f32(rrq_frag<NBQ>(wt[k0], cq, j), yv[k0], a, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (BITS != 0) H1q[4 * j + r] = rr_mask_gate<4 * NBQ>(m1w, 4 * j + r, a[r]);
                else H1q[4 * j + r] = (a1v[j][r] > 0.f) ? a[r] : 0.f;
            }
            *reinterpret_cast<f32x4*>(sZ1 + r16 * LDH + ub + 16 * j + 4 * q) = f32x4{H1q[4 * j], H1q[4 * j + 1], H1q[4 * j + 2], H1q[4 * j + 3]};
        }
    }
    lds_barrier();
    float H1[KS];
#pragma unroll
    for (int jo = 0; jo < NBA; ++jo) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sZ1 + r16 * LDH + 16 * jo + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) H1[4 * jo + r] = v[r];
    }
    QBSTAMP(2)

    // ---- the hid x hid layer: this wave's quarter of dz0, gated by a0 as its blocks finish
    float Hz[KSQ];
    f32x4 acc[NBQ];
    f32x4 czero[NBQ];
#pragma unroll
    for (int jo = 0; jo < NBQ; ++jo) czero[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto finish = [&](int jo, int r) __attribute__((always_inline)) {
        if constexpr (BITS != 0) Hz[4 * jo + r] = rr_mask_gate<4 * NBQ>(m0w, 4 * jo + r, acc[jo][r]);
        else Hz[4 * jo + r] = (a0v[jo][r] > 0.f) ? acc[jo][r] : 0.f;
    };
    panel.run(acc, czero, H1, rs, voff, wcur, [&](int) __attribute__((always_inline)) {},
              [&](int jo, int r) __attribute__((always_inline)) { finish(jo, r); });
    QBSTAMP(3)
    // ---- this quarter of dx (one block over its units); the pending pair of dz0 blocks is finished just in time
    if (io.dx) {
        float wo[KSQ];
#pragma unroll
        for (int j = 0; j < NBQ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[4 * j + r] = wx[j][r];
        const f32x4 o = P::template block<KSQ>(wo, Hz, [&](int ks) __attribute__((always_inline)) {
            if (ks < 8) finish(NBQ - 2 + (ks >> 2), ks & 3);
        });
        *reinterpret_cast<f32x4*>(sO + (cq * QT + r16) * 16 + 4 * q) = o;
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) finish(NBQ - 2 + (ks >> 2), ks & 3);
    }
    QBSTAMP(4)
    // ---- skinny partials: the last layer's activations of this thread's column (requested before the stores: loads and
    //      stores share the in-order vmcnt queue), dz0 through LDS
    float av[QT];
    const int kcol = min(tid, HID - 1);
    if (sk) {
        const float* a1 = io.acts + ls + kcol;
#pragma unroll
        for (int mm = 0; mm < QT; ++mm) av[mm] = a1[(long)min(row0 + mm, B - 1) * HID];
    }
    // ---- dz rows, this quarter of both layers: nothing waits for them (the barrier below orders LDS only)
    if (io.dz && row_ok) {
        float* z0p = io.dz + (long)grow * HID + ub + 4 * q;
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
            *reinterpret_cast<f32x4*>(z0p + ls + 16 * j) = f32x4{H1q[4 * j], H1q[4 * j + 1], H1q[4 * j + 2], H1q[4 * j + 3]};
            // (layer 0's rows only where somebody reads them: with the skinny partials below nobody does, nlbac_mlp_io::dz_first)
            if (!skip0) *reinterpret_cast<f32x4*>(z0p + 16 * j) = f32x4{Hz[4 * j], Hz[4 * j + 1], Hz[4 * j + 2], Hz[4 * j + 3]};
        }
    }
    if (sk) {
#pragma unroll
        for (int j = 0; j < NBQ; ++j)
            *reinterpret_cast<f32x4*>(sZ0 + r16 * LDH + ub + 16 * j + 4 * q) = f32x4{Hz[4 * j], Hz[4 * j + 1], Hz[4 * j + 2], Hz[4 * j + 3]};
    }
    lds_barrier();
    QBSTAMP(5)
    if (io.dx) {   // the four parts -> dx (columns below dx_first are not wanted: nlbac_mlp_io)
        const int r = tid >> 4, c = tid & 15;
        if (row0 + r < B && c >= io.dx_first && c < idim)
            io.dx[(long)(row0 + r) * io.dx_ld + c] = (sO[tid] + sO[QT * 16 + tid]) + (sO[2 * QT * 16 + tid] + sO[3 * QT * 16 + tid]);
    }
    if (sk) {      // thread = hidden column k; the sums and their order are mlp_bwd_skinny_partial_kernel's (row after row,
                   // fused multiply-adds): the column's 16 values in registers, the rows' x / dy as broadcast float4 reads
        float* w = io.skinny_ws + (long)blockIdx.x * (2 + idim + odim + 1) * 256 + tid;
        const bool live = tid < HID;
        float z0[QT];
        float b0 = 0.f, b1 = 0.f;
#pragma unroll
        for (int mm = 0; mm < QT; ++mm) { z0[mm] = sZ0[mm * LDH + kcol]; b1 += sZ1[mm * LDH + kcol]; }
#pragma unroll
        for (int mm = 0; mm < QT; ++mm) b0 += z0[mm];
        w[0] = live ? b0 : 0.f;
        w[256] = live ? b1 : 0.f;
        for (int i0 = 0; i0 < idim; i0 += 4) {    // dW_0[k][i] = sum_m dz0[m][k] x[m][i]
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mm = 0; mm < QT; ++mm) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(sx + mm * 16 + i0);
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = __builtin_fmaf(z0[mm], xv[c], a[c]);
            }
            // (all four chains interleaved as they stand: behind the guards below the compiler sinks each into a block of
            //  its own — serial chains of dependent FMAs)
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (i0 + c < idim) w[(long)(2 + i0 + c) * 256] = live ? a[c] : 0.f;
        }
        for (int o0 = 0; o0 < odim; o0 += 4) {    // dW_2[o][k] = sum_m dy[m][o] a1[m][k]
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mm = 0; mm < QT; ++mm) {
                const f32x4 dv = *reinterpret_cast<const f32x4*>(sdy + mm * 16 + o0);
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = __builtin_fmaf(dv[c], av[mm], a[c]);
            }
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (o0 + c < odim) w[(long)(2 + idim + o0 + c) * 256] = live ? a[c] : 0.f;
        }
        float bl = 0.f;
        if (tid < 16)
            for (int mm = 0; mm < QT; ++mm) bl += sdy[mm * 16 + tid];
        w[(long)(2 + idim + odim) * 256] = bl;
    }
    QBSTAMP(6)
    // ---- the dy head's batch sums: published / finished here, off the tile's critical path
    if (!plain_dy && H.kind != 1) {
        __syncthreads();                           // (sx is the election's scratch: every read of it above is done)
        dy_head_finish<QT>(H, by, row0, gridDim.x, sx, gridDim.y, pend);
    }
    // ---- batch sums an earlier launch's head left to this one (nlbac_dy_head::finish)
    dy_head_jobs(H, (int)blockIdx.x, by, (int)gridDim.x, sx);
}

// ---------------------------------------------------------------------------------------------------------------------
static bool mrq_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_MLP_RRQ"); return !(e && e[0] == '0'); }();
    return on;
}

// the widths the quarter-panel kernels serve (everything else nlbac_mlp_rr_eligible accepts stays on the half-panel ones)
bool nlbac_mlp_rrq_eligible(const nlbac_mlp* nets, int n_nets) {
    if (!mrq_enabled() || !nlbac_mlp_rr_eligible(nets, n_nets)) return false;
    return nets[0].hid == 128 || nets[0].hid == 256;
}

int nlbac_mlp_rrq_fwd_launch(const MlpLaunch& L, int n_nets, const nlbac_gauss_head& G, const char* who, hipStream_t s) {
    if (!nlbac_mlp_rrq_eligible(L.net, n_nets)) return 1;
    const int hid = L.net[0].hid;
    bool bits = false;
    for (int i = 0; i < n_nets; ++i) bits = bits || L.io[i].masks != nullptr;
    const size_t lds = (size_t)(QT * (hid + 4) + 4 * QT * 16) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.B, QT), n_nets);
#define MRQ_FWD(NBQ_)                                                                                            \
    if (bits) hipLaunchKernelGGL((mlp_rrq_fwd_kernel<NBQ_, 1>), grid, dim3(256), lds, s, L, G);                  \
    else hipLaunchKernelGGL((mlp_rrq_fwd_kernel<NBQ_, 0>), grid, dim3(256), lds, s, L, G);
    if (hid == 128) { MRQ_FWD(2) } else { MRQ_FWD(4) }
#undef MRQ_FWD
    NLBAC_CHECK_LAUNCH(who);
    return 0;
}

int nlbac_mlp_rrq_bwd_launch(const MlpLaunch& L, int n_nets, const nlbac_dy_head& H, const char* who, hipStream_t s) {
    if (!nlbac_mlp_rrq_eligible(L.net, n_nets)) return 1;
    const int hid = L.net[0].hid;
    bool sk = false, wide_out = false;
    int n_bits = 0;
    for (int i = 0; i < n_nets; ++i) {
        sk = sk || (L.io[i].skinny_ws != nullptr && L.io[i].dz != nullptr);
        wide_out = wide_out || L.net[i].out_dim > 4;
        n_bits += L.io[i].masks != nullptr;
    }
    NLBAC_REQUIRE(n_bits == 0 || n_bits == n_nets, "%s: ReLU mask words (nlbac_mlp_io::masks) for all nets of a launch or for none", who);
    const size_t lds = (size_t)(6 * QT * 16 + (sk ? 2 : 1) * QT * (hid + 4)) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.B, QT), n_nets);
#define MRQ_BWD2(NBQ_, KLO_)                                                                                       \
    if (n_bits) hipLaunchKernelGGL((mlp_rrq_bwd_kernel<NBQ_, KLO_, 1>), grid, dim3(256), lds, s, L, H);             \
    else hipLaunchKernelGGL((mlp_rrq_bwd_kernel<NBQ_, KLO_, 0>), grid, dim3(256), lds, s, L, H);
#define MRQ_BWD(NBQ_)                                                                                              \
    if (wide_out) { MRQ_BWD2(NBQ_, 4) } else { MRQ_BWD2(NBQ_, 1) }
    if (hid == 128) { MRQ_BWD(2) } else { MRQ_BWD(4) }
#undef MRQ_BWD
#undef MRQ_BWD2
    NLBAC_CHECK_LAUNCH(who);
    return 0;
}
