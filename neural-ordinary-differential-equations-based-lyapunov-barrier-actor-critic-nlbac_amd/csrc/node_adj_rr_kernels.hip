// Continuous adjoint of the control-affine NODE (odeint_adjoint; see node_adjoint_kernels.hip for the mathematics and the
// reference call sites — P/sac_cbf_clf/sac_cbf_clf.py:459,499,534 under torchdiffeq's OdeintAdjointMethod, BASELINE
// configs[3]) with REGISTER-RESIDENT layer chains: the launches of nlbac_node_adj_step for the reference's NODE shapes
// (node_rr_kernels.hip: f_net five layers, g_net four, width 64 / 100 / 128) when nothing but the step's result is kept
// (rollouts: the adjoint of the states and actions; the NODE fit's parameter quadrature stays on the LDS-tiled kernel).
//
// Same wave roles as the fused RK kernels: four waves per 32-row tile, wave = (net, 16 rows).  A stage of the augmented
// state z = [y | a_x | a_u] is, per wave, ONE uninterrupted chain
//     layer 0 -> hid x hid layers -> output layer  (forward pack)  |  top product -> hid x hid products -> dX  (backward pack)
// and the ReLU masks the backward half gates with never leave the wave: one 32-bit word per layer, assembled by the
// forward half in a register (rr_mask_push) and read by the backward half from the same register.  The LDS-tiled kernel
// keeps its masks and activations in LDS tiles, ten layer steps of GEMM + epilogue + barrier per stage: 204 us per
// attempted step at 16384 rows (0.35 of the fp32-MFMA roof).  The stage algebra on z (stage input, k_y = -(f + g u),
// k_au = g^T a_x, k_ax = dX_f + dX_g, step result and error estimate) is the LDS-tiled kernel's, operation for operation.
#include "node_adj_shared.h"
#include "rr_device.h"
#include <cstdlib>
#include <type_traits>

bool nlbac_node_rr_eligible(const nlbac_mlp* f, const nlbac_mlp* g);      // (node_rr_kernels.hip)

#define ARR_MAX_W 4        /* layer 0 + up to three hid x hid layers */

// KEEP (the NODE fit's parameter quadrature): every evaluated stage's inputs, the output-layer gradient of g_net, the
// activations and the pre-activation gradients additionally go out as rows, for nlbac_mlp_bwd_weights.
template <int NB, int R, int KEEP>
__global__ __launch_bounds__(256) void node_adj_rr_kernel(const NodeAdjLaunch L) {
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID, TB = NB - 2, NT = KS - 4 * TB, G0 = rr_group_first(NB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 1, half = wave & 1;
    const int n = L.n, ns = L.n_s, nu = L.n_u, W = L.W;
    const int WP = L.ld;          // LDS row stride of a row of z: W | 1 (odd: the lanes of a column walk distinct banks)
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    const nlbac_mlp& net = L.net[grp];
    const int nw = net.n_layers - 1;
    const int q = lane >> 4, r16 = lane & 15;
    const int m = 16 * half + r16, grow = row0 + m;
    const bool row_ok = grow < n;
    const int KS0 = (ns + 3) >> 2, KL0 = (ns + 4) >> 2;
    const int KSO = (grp == 0) ? KS0 : KS0 * nu;          // k-steps of the output layer's transposed product (<= 4)

    // ---- LDS
    float* const sKZ = smem;                                              // [stage][32][WP]
    float* const sZ0 = sKZ + L.S_total * NLBAC_MLP_TILE * WP;    // [32][WP]
    float* const sZS = sZ0 + NLBAC_MLP_TILE * WP;                     // [32][WP] stage input
    float* const sU = sZS + NLBAC_MLP_TILE * WP;                      // [32][4]
    float* const sH = sU + NLBAC_MLP_TILE * ADJ_MAX_NU;                   // [32]
    float* const sLive = sH + NLBAC_MLP_TILE;                             // [32]
    float* const sF = sLive + NLBAC_MLP_TILE;                             // [32][8]
    float* const sG = sF + NLBAC_MLP_TILE * ADJ_MAX_NS;                   // [32][32]
    float* const sDX = sG + NLBAC_MLP_TILE * ADJ_MAX_GOUT;                // [2][32][8]
    float* const sW0 = sDX + 2 * NLBAC_MLP_TILE * ADJ_MAX_NS;             // [net][k-step < 3][block < 8][lane] layer 0's A fragments
    float* const sWt = sW0 + 2 * 3 * 8 * 64;                              // [net][k-step < 4][block < 8][lane] W_out^T's A fragments
    __shared__ int s_any;
    if (tid == 0) s_any = 0;

    const float* const params = net.params;
    int boff[ARR_MAX_W];
#pragma unroll
    for (int l = 0; l < ARR_MAX_W; ++l) boff[l] = net.b_off[l];

    // ---- the wave's weight stream: forward fragments of layers 1 .. nw-1, then the backward fragments nw-1 .. 1, round and
    //      round over the stages
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int fbase = net.rr_fwd_off * 4, bbase = net.rr_bwd_off * 4;
    RRGemm<S> gemm;
    gemm.prime(rs, voff, fbase);

    // ---- constants of the launch: layer 0's and W_out^T's A fragments to LDS, the output layer's and W_0^T's into registers
    if (half == 0) {
        const float* W0 = params + net.w_off[0];
        const float* b0 = params + boff[0];
        const float* Wl = params + net.w_off[nw];
#pragma unroll
        for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int uo = rr_unit_out(NB, R, jo, r16), col = 4 * k0 + q, uc = max(uo, 0);
                const float vw = W0[uc * ns + min(col, ns - 1)], vb = b0[uc];
                sW0[((grp * 3 + k0) * 8 + jo) * 64 + lane] = (uo < 0 || col > ns) ? 0.f : (col < ns ? vw : vb);
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int o = -1;
            if (grp == 0) { const int c = 4 * e + q; if (e < KS0 && c < ns) o = c; }
            else { const int k0 = e / nu, u = e - k0 * nu, c = 4 * k0 + q; if (e < KS0 * nu && c < ns) o = c * nu + u; }
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int uo = rr_unit_out(NB, R, jo, r16);
                const float vw = Wl[(long)max(o, 0) * HID + max(uo, 0)];
                sWt[((grp * 4 + e) * 8 + jo) * 64 + lane] = (o >= 0 && uo >= 0) ? vw : 0.f;
            }
        }
    }
    float wo[KS], w0t[KS];
    {
        // forward: which output the A row r16 of the output block computes (node_rr_kernels.hip::rr_out_row)
        int orow;
        {
            const int qp = r16 >> 2, rp = r16 & 3;
            if (grp == 0) { const int c = 4 * rp + qp; orow = (rp < KS0 && c < ns) ? c : -1; }
            else { const int k0 = rp / nu, u = rp - k0 * nu, c = 4 * k0 + qp; orow = (rp < KS0 * nu && c < ns) ? c * nu + u : -1; }
        }
        const float* wrow = params + net.w_off[nw] + (long)max(orow, 0) * HID;
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) {
            const f32x4 v = rr_row_load<S>(wrow, jo, q);
#pragma unroll
            for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) wo[4 * jo + r] = (orow >= 0) ? v[r] : 0.f;
        }
        const float* W0 = params + net.w_off[0];
        const int c = 4 * (r16 & 3) + (r16 >> 2);          // A row 4 q' + r' computes dX component 4 r' + q'
        const bool ok = (r16 & 3) < KS0 && c < ns;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float vw = W0[(long)rr_unit_in(NB, R, ks, q) * ns + min(c, ns - 1)];
            w0t[ks] = ok ? vw : 0.f;
        }
    }
    int o_idx[4]; float o_bias[4];
    {
        const float* bo = params + net.b_off[nw];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int o = -1;
            if (grp == 0) { const int c = 4 * r + q; if (r < KS0 && c < ns) o = c; }
            else { const int k0 = r / nu, u = r - k0 * nu, c = 4 * k0 + q; if (r < KS0 * nu && c < ns) o = c * nu + u; }
            o_idx[r] = o;
            const float vb = bo[max(o, 0)];
            o_bias[r] = (o >= 0) ? vb : 0.f;
        }
    }

    // ---- the tile's rows of the step: z0, u, h, liveness, the stage derivatives an earlier launch left
    for (int idx = tid; idx < NLBAC_MLP_TILE * WP; idx += 256) {
        const int mm = idx / WP, c = idx - mm * WP, row = row0 + mm;
        sZ0[idx] = (row < n && c < W) ? L.Z0[(long)row * W + c] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE * ADJ_MAX_NU) {
        const int mm = tid >> 2, c = tid & 3, row = row0 + mm;
        sU[tid] = (row < n && c < nu) ? L.u[(long)row * nu + c] : 0.f;
    }
    __syncthreads();            // (s_any = 0 is there)
    if (tid < NLBAC_MLP_TILE) {
        const int row = row0 + tid, p = min(row, n - 1) / L.rpp;
        sH[tid] = L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p];
        const bool live = row < n && !(L.ctl && L.ctl[(long)p * NLBAC_DOPRI_CTL + C_DONE] > 0.0);
        sLive[tid] = live ? 1.f : 0.f;
        if (live) s_any = 1;
    }
    for (int idx = tid; idx < L.st_lo * NLBAC_MLP_TILE * WP; idx += 256) {
        const int j = idx / (NLBAC_MLP_TILE * WP), rem = idx - j * NLBAC_MLP_TILE * WP;
        const int mm = rem / WP, c = rem - mm * WP, row = row0 + mm;
        sKZ[idx] = (row < n && c < W) ? L.KZ[((long)j * n + row) * W + c] : 0.f;
    }
    __syncthreads();
    if (!s_any) return;                      // (uniform) every problem of this tile has finished its solve

    for (int st = L.st_lo; st < L.st_hi; ++st) {
        // ---- stage input  Z_st = Z0 + h sum_j beta[st][j] K_j  (all of z: y feeds the nets, a_x is the cotangent)
        for (int idx = tid; idx < NLBAC_MLP_TILE * WP; idx += 256) {
            const int mm = idx / WP, c = idx - mm * WP;
            float a = sZ0[idx];
            const float h = sH[mm];
            for (int j = 0; j < st; ++j)
                if (L.beta[st][j] != 0.f) a = a + sKZ[(j * NLBAC_MLP_TILE + mm) * WP + c] * (L.beta[st][j] * h);
            sZS[idx] = a;
            if (KEEP && c < W && sLive[mm] != 0.f) L.ZS[((long)st * n + row0 + mm) * W + c] = a;
        }
        __syncthreads();
        const long srow = (long)st * n + grow;
        const bool keep_row = KEEP && row_ok && sLive[m] != 0.f;
        (void)srow; (void)keep_row;

        // =========================== forward chain ===========================
        float Ha[KS], Hb[KS];
        f32x4 acc0[NB], acc[NB], bv[NB], bpre[3];
        unsigned wd = 0u;                 // the mask word being assembled
        unsigned mreg[ARR_MAX_W];         // the finished words, one per layer: what the backward half gates with
#pragma unroll
        for (int l = 0; l < ARR_MAX_W; ++l) mreg[l] = 0u;
        auto prefetch_bias = [&](int l) __attribute__((always_inline)) {
#pragma unroll
            for (int jo = 0; jo < G0; ++jo) bpre[jo] = rr_bias<S>(params + boff[l], jo, q);
        };
        prefetch_bias(1);
        auto keep_word = [&](int l, unsigned word) __attribute__((always_inline)) {
            // (static layer index after inlining; rows past the end gate everything off)
#pragma unroll
            for (int ll = 0; ll < ARR_MAX_W; ++ll)
                if (ll == l) mreg[ll] = row_ok ? word : 0u;
        };
        auto save_rows = [&](float* base, int l, int jo, const float (&H)[KS]) __attribute__((always_inline)) {
            if (!keep_row) return;
            f32x4 hv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) hv[rr] = H[4 * jo + rr];
            rr_row_store<S>(base + (long)l * L.acts_ls[grp] + srow * HID, jo, q, hv);
        };
        auto pre_l0 = [&](int ks) __attribute__((always_inline)) {
            const int jo = (ks < 4 * (NB - 1)) ? (ks >> 2) : NB - 1, r = ks - 4 * jo;
            const float h = rr_relu(acc0[jo][r]);
            Ha[ks] = h;
            rr_mask_push(wd, h);
            if (KEEP && r == ((jo < NB - 1) ? 3 : R - 1)) save_rows(L.acts[grp], 0, jo, Ha);
            if (ks == KS - 1) keep_word(0, wd);
        };
        auto pre_tail_f = [&](int lp, float (&H)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            const float h = rr_relu(acc[jo][r]);
            H[4 * TB + t] = h;
            rr_mask_push(wd, h);
            if (KEEP && (t == 3 || t == NT - 1)) save_rows(L.acts[grp], lp, jo, H);
            if (t == NT - 1) keep_word(lp, wd);
        };
        {   // layer 0: K = ns + 1 (one to three k-steps), straight from the stage input's y part
            float yv[3], a0[3][NB];
#pragma unroll
            for (int k0 = 0; k0 < 3; ++k0) {
                const int col = 4 * k0 + q;
                yv[k0] = (col < ns) ? sZS[m * WP + min(col, ns - 1)] : (col == ns ? 1.f : 0.f);
            }
#pragma unroll
            for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) a0[k0][jo] = (k0 < KL0) ? sW0[((grp * 3 + k0) * 8 + jo) * 64 + lane] : 0.f;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][jo], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (KL0 > 1) acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1][jo], yv[1], acc0[jo], 0, 0, 0);
                if (KL0 > 2) acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2][jo], yv[2], acc0[jo], 0, 0, 0);
            }
        }
        auto wide = [&](auto lc, float (&Hin)[KS], float (&Hout)[KS]) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) bv[jo] = (jo < G0) ? bpre[jo] : rr_bias<S>(params + boff[l], jo, q);
            __builtin_amdgcn_sched_barrier(0);
            const int cur = fbase + (l - 1) * S::LAYER_BYTES;
            // behind the last forward layer the stream turns round: the first backward product's fragments (layer nw-1)
            const int nxt = (l + 1 < nw) ? cur + S::LAYER_BYTES : bbase + (nw - 2) * S::LAYER_BYTES;
            gemm.run(acc, bv, Hin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) {
                         if (l == 1) pre_l0(ks);
                         else pre_tail_f(l - 1, Hin, ks);
                     },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         const float h = rr_relu(acc[jo][r]);
                         Hout[4 * jo + r] = h;
                         rr_mask_push(wd, h);
                         if (KEEP && r == 3) save_rows(L.acts[grp], l, jo, Hout);
                     },
                     [&]() __attribute__((always_inline)) {
                         if (l + 1 < nw) prefetch_bias(l + 1);
                     });
        };
        auto outl = [&](auto lc, float (&Hin)[KS]) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            const f32x4 o = RRGemm<S>::block(wo, Hin, [&](int ks) __attribute__((always_inline)) { pre_tail_f(l - 1, Hin, ks); });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (o_idx[r] < 0) continue;
                const float val = o[r] + o_bias[r];
                if (grp == 0) sF[m * ADJ_MAX_NS + o_idx[r]] = val;
                else sG[m * ADJ_MAX_GOUT + o_idx[r]] = val;
            }
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
        wide(I1{}, Ha, Hb);
        wide(I2{}, Hb, Ha);
        if (grp == 0) { wide(I3{}, Ha, Hb); outl(I4{}, Hb); }
        else outl(I3{}, Ha);
        __syncthreads();

        // ---- k_y = -(f + g u) ; k_au = g^T a_x
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += 256) {
            const int mm = idx / ns, r = idx - mm * ns;
            float a = sF[mm * ADJ_MAX_NS + r];
            for (int c = 0; c < nu; ++c) a += sG[mm * ADJ_MAX_GOUT + r * nu + c] * sU[mm * ADJ_MAX_NU + c];
            sKZ[(st * NLBAC_MLP_TILE + mm) * WP + r] = -a;
        }
        for (int idx = tid; idx < NLBAC_MLP_TILE * nu; idx += 256) {
            const int mm = idx / nu, c = idx - mm * nu;
            float a = 0.f;
            for (int r = 0; r < ns; ++r) a += sG[mm * ADJ_MAX_GOUT + r * nu + c] * sZS[mm * WP + ns + r];
            sKZ[(st * NLBAC_MLP_TILE + mm) * WP + 2 * ns + c] = a;
        }

        // =========================== backward chain ===========================
        // cotangents of the two output layers: f: a_x itself, g: a_x u^T — this lane's B operands of the top product
        float dy[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k0 = (grp == 0) ? e : e / nu, u = (grp == 0) ? 0 : e - k0 * nu, c = 4 * k0 + q;
            const bool ok = (grp == 0) ? (e < KS0 && c < ns) : (e < KS0 * nu && c < ns);
            const float ax = sZS[m * WP + ns + min(c, ns - 1)];
            const float uu = (grp == 0) ? 1.f : sU[m * ADJ_MAX_NU + min(u, nu - 1)];
            dy[e] = ok ? ((grp == 0) ? ax : ax * uu) : 0.f;
            if (KEEP && grp == 1 && ok && keep_row) L.dG[srow * (ns * nu) + c * nu + u] = dy[e];
        }
        float Za[KS], Zb[KS];
        f32x4 acct[NB], zero[NB];
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) zero[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned mw = 0u, mwt = 0u;
        auto word_of = [&](int l) __attribute__((always_inline)) {
            unsigned v = 0u;
#pragma unroll
            for (int ll = 0; ll < ARR_MAX_W; ++ll)
                if (ll == l) v = mreg[ll];
            return v;
        };
        auto pre_top = [&](int ks) __attribute__((always_inline)) {
            const int jo = (ks < 4 * (NB - 1)) ? (ks >> 2) : NB - 1, r = ks - 4 * jo;
            Za[ks] = rr_mask_gate<KS>(mwt, ks, acct[jo][r]);
            if (KEEP && r == ((jo < NB - 1) ? 3 : R - 1)) save_rows(L.dz[grp], nw - 1, jo, Za);
        };
        int tail_layer = 0;          // (the layer whose dz the pending tail belongs to)
        auto pre_tail_b = [&](float (&Z)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            Z[4 * TB + t] = rr_mask_gate<KS>(mwt, 4 * TB + t, acc[jo][r]);
            if (KEEP && (t == 3 || t == NT - 1)) save_rows(L.dz[grp], tail_layer, jo, Z);
        };
        mw = word_of(nw - 1);
        {   // top product: dz_top = mask_top * (W_out^T dy)
            float at[4][NB];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) at[e][jo] = (e < 2 || KSO > 2) ? sWt[((grp * 4 + e) * 8 + jo) * 64 + lane] : 0.f;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[0][jo], dy[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[1][jo], dy[1], acct[jo], 0, 0, 0);
                if (KSO > 2) {
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[2][jo], dy[2], acct[jo], 0, 0, 0);
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[3][jo], dy[3], acct[jo], 0, 0, 0);
                }
            }
        }
        auto prod = [&](auto pc, float (&Zin)[KS], float (&Zout)[KS]) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            const int lo = nw - 1 - p;                            // the layer whose dz this product yields
            mwt = mw;
            mw = word_of(lo);
            __builtin_amdgcn_sched_barrier(0);
            const int cur = bbase + lo * S::LAYER_BYTES;              // fragments of layer lo + 1 sit at index lo
            // behind the last backward product the stream goes on with the next stage's first forward layer
            const int nxt = (lo >= 1) ? cur - S::LAYER_BYTES : fbase;
            gemm.run(acc, zero, Zin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) {
                         if (p == 1) pre_top(ks);
                         else pre_tail_b(Zin, ks);
                     },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         Zout[4 * jo + r] = rr_mask_gate<KS>(mw, 4 * jo + r, acc[jo][r]);
                         if (KEEP && r == 3) save_rows(L.dz[grp], lo, jo, Zout);
                     },
                     [&]() __attribute__((always_inline)) {});
            tail_layer = lo;          // (this product's own tail is finished inside the next one)
        };
        auto dxl = [&](float (&Zin)[KS]) __attribute__((always_inline)) {
            mwt = mw;
            const f32x4 o = RRGemm<S>::block(w0t, Zin, [&](int ks) __attribute__((always_inline)) { pre_tail_b(Zin, ks); });
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = 4 * r + q;
                if (r < KS0 && c < ns) sDX[(grp * NLBAC_MLP_TILE + m) * ADJ_MAX_NS + c] = o[r];
            }
        };
        prod(I1{}, Za, Zb);
        prod(I2{}, Zb, Za);
        if (grp == 0) { prod(I3{}, Za, Zb); dxl(Zb); }
        else dxl(Za);
        __syncthreads();

        // ---- k_ax = dX_f + dX_g; the stage's derivative row goes out
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += 256) {
            const int mm = idx / ns, c = idx - mm * ns;
            sKZ[(st * NLBAC_MLP_TILE + mm) * WP + ns + c] =
                sDX[mm * ADJ_MAX_NS + c] + sDX[(NLBAC_MLP_TILE + mm) * ADJ_MAX_NS + c];
        }
        __syncthreads();
        for (int idx = tid; idx < NLBAC_MLP_TILE * W; idx += 256) {
            const int mm = idx / W, c = idx - mm * W;
            if (sLive[mm] != 0.f) L.KZ[((long)st * n + row0 + mm) * W + c] = sKZ[(st * NLBAC_MLP_TILE + mm) * WP + c];
        }
    }

    // ---- step outputs
    for (int idx = tid; idx < NLBAC_MLP_TILE * W; idx += 256) {
        const int mm = idx / W, c = idx - mm * W, row = row0 + mm;
        if (sLive[mm] == 0.f) continue;
        const float h = sH[mm];
        if (L.Z1) {
            float a = sZ0[mm * WP + c];
            for (int j = 0; j < L.n_out; ++j)
                if (L.c_out[j] != 0.f) a = a + sKZ[(j * NLBAC_MLP_TILE + mm) * WP + c] * (L.c_out[j] * h);
            L.Z1[(long)row * W + c] = a;
        }
        if (L.ERR) {
            float a = 0.f;
            for (int j = 0; j < L.n_err; ++j)
                if (L.c_err[j] != 0.f) a = a + sKZ[(j * NLBAC_MLP_TILE + mm) * WP + c] * (L.c_err[j] * h);
            L.ERR[(long)row * W + c] = a;
        }
        if (L.ip_out) {
            // the interpolant of z at t_end, should this attempt be accepted and finish the problem's solve (the
            // controller's own test and abscissa: ode_control.h; dopri_interp_fwd_kernel's arithmetic on z1 = the value
            // written to Z1 above)
            const int p = row / L.rpp;
            const double t = L.ctl[(long)p * NLBAC_DOPRI_CTL + C_T], hd = L.ctl[(long)p * NLBAC_DOPRI_CTL + C_H];
            if (t + hd >= L.t_end) {
                const float x = (float)((L.t_end - t) / hd);
                const float a0 = sZ0[mm * WP + c];
                float a1 = a0, k[7];
                for (int j = 0; j < L.n_out; ++j)
                    if (L.c_out[j] != 0.f) a1 = a1 + sKZ[(j * NLBAC_MLP_TILE + mm) * WP + c] * (L.c_out[j] * h);
#pragma unroll
                for (int j = 0; j < 7; ++j) k[j] = sKZ[(j * NLBAC_MLP_TILE + mm) * WP + c];
                L.ip_out[(long)row * W + c] = dopri_interp_value(a0, a1, k, h, x);
            }
        }
    }
}

static bool adj_rr_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_ADJ_RR"); return !(e && e[0] == '0'); }();
    return on;
}

static bool adj_rr_keep_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_ADJ_RR_KEEP"); return !(e && e[0] == '0'); }();
    return on;
}

int nlbac_node_adj_rr_launch(NodeAdjLaunch& L, hipStream_t s) {
    const bool keep = L.ZS != nullptr;
    if (!adj_rr_enabled() || (keep && !adj_rr_keep_enabled()) || !nlbac_node_rr_eligible(&L.net[0], &L.net[1])) return 1;
    using Kernel = void (*)(const NodeAdjLaunch);
    const int hid = L.net[0].hid;
    const Kernel k = keep ? (hid == 64 ? node_adj_rr_kernel<4, 4, 1> : (hid == 100 ? node_adj_rr_kernel<7, 1, 1> : node_adj_rr_kernel<8, 4, 1>))
                          : (hid == 64 ? node_adj_rr_kernel<4, 4, 0> : (hid == 100 ? node_adj_rr_kernel<7, 1, 0> : node_adj_rr_kernel<8, 4, 0>));
    L.ld = L.W | 1;
    const size_t lds = (size_t)((L.S_total + 2) * NLBAC_MLP_TILE * L.ld +
                                NLBAC_MLP_TILE * (ADJ_MAX_NU + 1 + 1 + ADJ_MAX_NS + ADJ_MAX_GOUT + 2 * ADJ_MAX_NS) +
                                2 * 3 * 8 * 64 + 2 * 4 * 8 * 64) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.n, NLBAC_MLP_TILE));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_adj_step(rr)");
    return 0;
}

extern "C" int nlbac_node_adj_interp_ok(const nlbac_mlp* f, const nlbac_mlp* g) {
    return (f && g && adj_rr_enabled() && nlbac_node_rr_eligible(f, g)) ? 1 : 0;
}
