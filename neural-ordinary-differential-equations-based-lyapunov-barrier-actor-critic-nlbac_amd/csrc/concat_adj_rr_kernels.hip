// Continuous adjoint of the single-net NODE  dx/dt = out_mu + out_sig * net(([x | c] - in_mu) * in_isig)  (SimulatedCars,
// C/sac_cbf_clf/model.py:179-205, under torchdiffeq 0.2.3's OdeintAdjointMethod; the Quadrotor-like task's normalised
// form): ONE launch per attempted RK step of the augmented system
//     z = [ y (n_s) | a_y (n_s) | a_c (n_c) ],  W = 2 n_s + n_c,  integrated in s = t1 - t:
//     dy/ds = -f(y, c),   da_y/ds = (df/dy)^T a_y,   da_c/ds = (df/dc)^T a_y
// — nlbac_concat_adj_step.  It replaces, per stage, the five launches nlbac_rk_combine -> nlbac_concat_adj_in ->
// nlbac_mlp_fwd -> nlbac_mlp_bwd_data -> nlbac_concat_adj_out (ode_kernels.hip; kept as the path of other shapes and as
// the cross-check) — about thirty launches per attempted dopri5 step — for the reference's depth (in -> hid -> hid -> hid
// -> out) at widths 64 / 100 / 128.
//
// Wave roles as concat_rr_kernels.hip: one wave owns 16 rows for the whole launch, two waves per workgroup, no barrier
// inside the stage loop.  A stage is ONE uninterrupted MFMA stream per wave (the scheme of node_adj_rr_kernels.hip):
//     layer 0 -> two hid x hid layers -> output layer  (forward pack)  |  top product -> two hid x hid products -> dX  (backward pack)
// the weight queue turns from the forward pack to the backward pack and back without draining, and the ReLU masks the
// backward half gates with stay in three registers.  The stage derivatives of z live in LDS across the stages; what a
// launch leaves in memory is written in one burst at its end.  KEEP (the NODE fit's parameter quadrature): the net's
// normalised inputs, the cotangent of its output, the activations and the pre-activation gradients of every evaluated
// stage additionally go out as rows, for nlbac_mlp_bwd_weights.
#include "concat_rk_shared.h"
#include "rr_device.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define CADJ_MAX_IN 15       /* in_dim + the bias column <= 16: four k-steps of layer 0 */

struct ConcatAdjLaunch {
    nlbac_mlp net;
    const float* c;                   // [n][n_c] carried inputs
    const float* Z0;                  // [n][W]
    float* KZ;                        // [S][n][W] stage derivatives (s-time); stages < st_lo are read, the others written
    float* Z1; float* ERR;            // [n][W] or null
    float* Xin; float* Ay;            // KEEP: [S][n][in_dim] / [S][n][n_s]
    float* acts; float* dz; long acts_ls;   // KEEP: [layer][S*n][hid]
    const float* norm;                // [in_mu | in_isig | out_mu | out_sig] or null
    int n, rpp, n_s, n_c, W, WP;
    int st_lo, st_hi, S_total;
    float beta[CK_MAX_STAGES][CK_MAX_STAGES];
    float c_out[CK_MAX_STAGES]; int n_out;
    float c_err[CK_MAX_STAGES]; int n_err;
    const double* h_dev; int h_stride; float h_val[8];
    const double* ctl;                // rows of problems whose C_DONE is set are left alone
    float* ip_out; double t_end;      // the interpolant of z at t_end, should this attempt finish the solve (as NodeAdjLaunch)
};

// NW: waves per workgroup (2 or 4), as concat_rr_kernels.hip: four-wave workgroups put one wave on every SIMD.
template <int NB, int R, int KEEP, int NW>
__global__ __launch_bounds__(64 * NW) void concat_adj_rr_kernel(const ConcatAdjLaunch L) {
    constexpr int TILE = 16 * NW;
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID, TB = NB - 2, NT = KS - 4 * TB, G0 = rr_group_first(NB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = L.n, ns = L.n_s, nc = L.n_c, W = L.W, WP = L.WP;
    const int row0 = blockIdx.x * TILE;
    const nlbac_mlp& net = L.net;
    const int idim = net.in_dim;
    const int q = lane >> 4, r16 = lane & 15, m = 16 * half + r16, grow = row0 + m;
    const bool row_ok = grow < n;
    const int KS0 = (ns + 3) >> 2;
    const float* const params = net.params;
    const float* const nrm = L.norm;
    const int S_total = L.S_total;

    float* const sKZ = smem;                                        // [stage][32][WP]
    float* const sZ0 = sKZ + S_total * TILE * WP;         // [32][WP]
    float* const sC = sZ0 + TILE * WP;                    // [32][CK_NC]
    float* const sH = sC + TILE * CK_NC;                  // [32]
    float* const sLive = sH + TILE;                       // [32]
    float* const sW0 = sLive + TILE;                      // [k-step < 4][block < 8][lane]: layer 0's A fragments
    float* const sWt = sW0 + 4 * 8 * 64;                            // [k-step < 4][block < 8][lane]: W_out^T's A fragments

    // ---- the wave's weight stream: forward fragments of layers 1, 2, backward fragments of layers 2, 1, round and round
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int fbase = net.rr_fwd_off * 4, bbase = net.rr_bwd_off * 4;
    RRGemm<S> gemm;
    gemm.prime(rs, voff, fbase);

    // ---- constants of the launch (every load unconditional: clamped index, select afterwards)
    if (half == 0) {
        const float* W0 = params + net.w_off[0];
        const float* b0 = params + net.b_off[0];
#pragma unroll
        for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int uo = rr_unit_out(NB, R, jo, r16), col = 4 * k0 + q, uc = max(uo, 0);
                const float vw = W0[uc * idim + min(col, idim - 1)], vb0 = b0[uc];
                sW0[(k0 * 8 + jo) * 64 + lane] = (uo < 0 || col > idim) ? 0.f : (col < idim ? vw : vb0);
            }
    } else {
        const float* Wl = params + net.w_off[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * e + q;
            const bool ok = e < KS0 && c < ns;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int uo = rr_unit_out(NB, R, jo, r16);
                const float v = Wl[(long)min(c, ns - 1) * HID + max(uo, 0)];
                sWt[(e * 8 + jo) * 64 + lane] = (ok && uo >= 0) ? v : 0.f;
            }
        }
    }
    float wo[KS], w0t[KS];
    {
        // forward: A row 4 q' + r' of the output block computes state component 4 r' + q'
        const int cq = 4 * (r16 & 3) + (r16 >> 2);
        const bool ok = (r16 & 3) < KS0 && cq < ns;
        const float* wrow = params + net.w_off[3] + (long)(ok ? cq : 0) * HID;
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) {
            const f32x4 v = rr_row_load<S>(wrow, jo, q);
#pragma unroll
            for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) wo[4 * jo + r] = ok ? v[r] : 0.f;
        }
        // backward: A row i (< in_dim) of dX's block is row i of W_0^T
        const float* W0 = params + net.w_off[0];
        const bool okx = r16 < idim;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float v = W0[(long)rr_unit_in(NB, R, ks, q) * idim + min(r16, idim - 1)];
            w0t[ks] = okx ? v : 0.f;
        }
    }
    float o_bias[4], o_mu[4], o_sig[4], i_mu[4], i_isig[4], x_isig[4];
    const float* const nrm_v = nrm ? nrm : params;        // (a readable address either way)
    const int nrm_n = nrm ? 2 * idim + 2 * ns : 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * r + q, cc = min(c, ns - 1);
        const bool ok = r < KS0 && c < ns;
        const float vb = params[net.b_off[3] + cc];
        const float vm = nrm_v[min(2 * idim + cc, nrm_n - 1)], vs = nrm_v[min(2 * idim + ns + cc, nrm_n - 1)];
        o_bias[r] = ok ? vb : 0.f;
        o_mu[r] = (ok && nrm) ? vm : 0.f;
        o_sig[r] = (ok && nrm) ? vs : 1.f;
        const int col = 4 * r + q, ci = min(col, idim - 1);           // layer 0's B operand: input column 4 k0 + q
        const float im = nrm_v[min(ci, nrm_n - 1)], is = nrm_v[min(idim + ci, nrm_n - 1)];
        i_mu[r] = (nrm && col < idim) ? im : 0.f;
        i_isig[r] = (nrm && col < idim) ? is : 1.f;
        const int i = 4 * q + r;                                      // dX leaves lane (q, row) with input column 4 q + r in register r
        const float xs = nrm_v[min(idim + min(i, idim - 1), nrm_n - 1)];
        x_isig[r] = (nrm && i < idim) ? xs : 1.f;
    }

    // ---- this wave's rows of the step: z0, carried inputs, h, liveness, the stage derivatives an earlier launch left
    for (int idx = lane; idx < 16 * WP; idx += 64) {
        const int mm = 16 * half + idx / WP, c = idx % WP, row = row0 + mm;
        const float v = L.Z0[(long)min(row, n - 1) * W + min(c, W - 1)];
        sZ0[mm * WP + c] = (row < n && c < W) ? v : 0.f;
    }
    {
        const int mm = 16 * half + (lane >> 2), c = lane & 3, row = row0 + mm;
        const float v = L.c[(long)min(row, n - 1) * nc + min(c, max(nc - 1, 0))];
        sC[mm * CK_NC + c] = (row < n && c < nc) ? v : 0.f;
    }
    bool live = false;
    if (lane < 16) {
        const int row = row0 + 16 * half + lane, p = min(row, n - 1) / L.rpp;
        sH[16 * half + lane] = L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p];
        live = row < n && !(L.ctl && L.ctl[(long)p * NLBAC_DOPRI_CTL + C_DONE] > 0.0);
        sLive[16 * half + lane] = live ? 1.f : 0.f;
    }
    const bool any_live = __builtin_amdgcn_ballot_w64(live) != 0ull;
    for (int idx = lane; idx < L.st_lo * 16 * WP; idx += 64) {
        const int j = idx / (16 * WP), rem = idx - j * 16 * WP;
        const int mm = 16 * half + rem / WP, c = rem % WP, row = row0 + mm;
        const float v = L.KZ[((long)j * n + min(row, n - 1)) * W + min(c, W - 1)];
        sKZ[(j * TILE + mm) * WP + c] = (row < n && c < W) ? v : 0.f;
    }
    // narrow nets keep both hid x hid layers' biases in registers (concat_rr_kernels.hip: BRES)
    constexpr bool BRES = NB <= 4;
    f32x4 bres[BRES ? 2 : 1][NB];
    if (BRES) {
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) bres[BRES ? l : 0][jo] = rr_bias<S>(params + net.b_off[1 + l], jo, q);
    }
    __syncthreads();           // (sW0 / sWt are shared by the two waves; everything else is the wave's own rows)
    if (!any_live) return;     // (wave-uniform; no barrier follows) every problem of this wave's rows has finished its solve

    f32x4 zero[NB];
#pragma unroll
    for (int jo = 0; jo < NB; ++jo) zero[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bnext[CK_MAX_STAGES];
#pragma unroll
    for (int j = 0; j < CK_MAX_STAGES; ++j) bnext[j] = L.beta[L.st_lo][j];

    for (int st = L.st_lo; st < L.st_hi; ++st) {
        float bn[CK_MAX_STAGES];
#pragma unroll
        for (int j = 0; j < CK_MAX_STAGES; ++j) bn[j] = bnext[j];
        const long srow = (long)st * n + grow;
        // ---- stage point  Z_st = Z0 + h sum_j beta[st][j] K_j : the columns this lane feeds to the net — input column
        //      4 k0 + q of [y | c | 1] and the cotangent a_y[4 e + q] — every LDS operand requested up front
        float yv[4], dy[4];
        {
            const float h = sH[m];
            float z0y[4], z0a[4], cv[4], ky[4][CK_MAX_STAGES - 1], ka[4][CK_MAX_STAGES - 1];
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) {
                const int col = 4 * k0 + q, cs = min(col, ns - 1), cc = min(max(col - ns, 0), max(nc - 1, 0));
                z0y[k0] = sZ0[m * WP + cs];
                z0a[k0] = sZ0[m * WP + ns + cs];
                cv[k0] = sC[m * CK_NC + cc];
#pragma unroll
                for (int j = 0; j < CK_MAX_STAGES - 1; ++j) {
                    const int jj = min(j, S_total - 1);
                    ky[k0][j] = sKZ[(jj * TILE + m) * WP + cs];
                    ka[k0][j] = sKZ[(jj * TILE + m) * WP + ns + cs];
                }
            }
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) {
                const int col = 4 * k0 + q;
                float a = z0y[k0], b = z0a[k0];
#pragma unroll
                for (int j = 0; j < CK_MAX_STAGES - 1; ++j) {
                    const float ta = a + ky[k0][j] * (bn[j] * h), tb = b + ka[k0][j] * (bn[j] * h);
                    const bool on = j < st && bn[j] != 0.f;
                    a = on ? ta : a;
                    b = on ? tb : b;
                }
                a = (col < ns) ? a : ((col < idim) ? cv[k0] : 0.f);
                a = (col < idim) ? (a - i_mu[k0]) * i_isig[k0] : a;
                b = (k0 < KS0 && col < ns) ? b * o_sig[k0] : 0.f;
                if (KEEP && row_ok) {
                    if (col < idim) L.Xin[srow * idim + col] = a;
                    if (k0 < KS0 && col < ns) L.Ay[srow * ns + col] = b;
                }
                yv[k0] = (col == idim) ? 1.f : a;
                dy[k0] = b;
            }
        }

        // =========================== forward chain ===========================
        float Ha[KS], Hb[KS];
        f32x4 acc0[NB], acc[NB], bv[NB], bpre[3];
        unsigned wd = 0u, mreg0 = 0u, mreg1 = 0u, mreg2 = 0u;
        auto prefetch_bias = [&](int l) __attribute__((always_inline)) {
            if (BRES) return;
#pragma unroll
            for (int jo = 0; jo < G0; ++jo) bpre[jo] = rr_bias<S>(params + net.b_off[l], jo, q);
        };
        prefetch_bias(1);
        auto keep_word = [&](int l) __attribute__((always_inline)) {      // (rows past the end gate everything off)
            const unsigned w = row_ok ? wd : 0u;
            if (l == 0) mreg0 = w; else if (l == 1) mreg1 = w; else mreg2 = w;
        };
        auto save_act = [&](int l, int jo, const float (&H)[KS]) __attribute__((always_inline)) {
            if (!KEEP || !row_ok) return;
            f32x4 hv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) hv[rr] = H[4 * jo + rr];
            rr_row_store<S>(L.acts + (long)l * L.acts_ls + srow * HID, jo, q, hv);
        };
        auto pre_l0 = [&](int ks) __attribute__((always_inline)) {
            const int jo = (ks < 4 * (NB - 1)) ? (ks >> 2) : NB - 1, r = ks - 4 * jo;
            const float h = rr_relu(acc0[jo][r]);
            Ha[ks] = h;
            rr_mask_push(wd, h);
            if (r == ((jo < NB - 1) ? 3 : R - 1)) save_act(0, jo, Ha);
            if (ks == KS - 1) keep_word(0);
        };
        auto pre_tail_f = [&](int lp, float (&H)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            const float h = rr_relu(acc[jo][r]);
            H[4 * TB + t] = h;
            rr_mask_push(wd, h);
            if (t == 3 || t == NT - 1) save_act(lp, jo, H);
            if (t == NT - 1) keep_word(lp);
        };
        {   // layer 0 (bias folded into the product): four k-steps, k-step outside
            float a0[4][NB];
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) a0[k0][jo] = sW0[(k0 * 8 + jo) * 64 + lane];
#pragma unroll
            for (int jo = 0; jo < NB; ++jo)
                acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][jo], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int k0 = 1; k0 < 4; ++k0)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo)
                    acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[k0][jo], yv[k0], acc0[jo], 0, 0, 0);
        }
        auto wide = [&](auto lc, float (&Hin)[KS], float (&Hout)[KS]) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo)
                bv[jo] = BRES ? bres[BRES ? l - 1 : 0][jo] : ((jo < G0) ? bpre[jo] : rr_bias<S>(params + net.b_off[l], jo, q));
            __builtin_amdgcn_sched_barrier(0);
            if (l == 1) {       // the next stage's tableau row, requested a stage ahead (concat_rr_kernels.hip)
                int sn = min(st + 1, S_total - 1);
                asm volatile("" : "+s"(sn));
#pragma unroll
                for (int j = 0; j < CK_MAX_STAGES; ++j) bnext[j] = L.beta[sn][j];
                __builtin_amdgcn_sched_barrier(0);
            }
            const int cur = fbase + (l - 1) * S::LAYER_BYTES;
            // behind the second forward layer the stream turns round: the first backward product's fragments (layer 2)
            const int nxt = (l == 1) ? cur + S::LAYER_BYTES : bbase + S::LAYER_BYTES;
            gemm.run(acc, bv, Hin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) {
                         if (l == 1) pre_l0(ks);
                         else pre_tail_f(l - 1, Hin, ks);
                     },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         const float h = rr_relu(acc[jo][r]);
                         Hout[4 * jo + r] = h;
                         rr_mask_push(wd, h);
                         if (r == 3) save_act(l, jo, Hout);
                     },
                     [&]() __attribute__((always_inline)) { if (l == 1) prefetch_bias(2); });
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        wide(I1{}, Ha, Hb);
        wide(I2{}, Hb, Ha);
        {   // output layer: k_y = -f
            const f32x4 o = RRGemm<S>::block(wo, Ha, [&](int ks) __attribute__((always_inline)) { pre_tail_f(2, Ha, ks); });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + q;
                if (r < KS0 && c < ns) sKZ[(st * TILE + m) * WP + c] = -((o[r] + o_bias[r]) * o_sig[r] + o_mu[r]);
            }
        }

        // =========================== backward chain ===========================
        float Za[KS], Zb[KS];
        f32x4 acct[NB];
        unsigned mw = mreg2, mwt = 0u;
        auto save_dz = [&](int l, int jo, const float (&Z)[KS]) __attribute__((always_inline)) {
            if (!KEEP || !row_ok) return;
            f32x4 zv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) zv[rr] = Z[4 * jo + rr];
            rr_row_store<S>(L.dz + (long)l * L.acts_ls + srow * HID, jo, q, zv);
        };
        auto pre_tail_b = [&](int lp, float (&Z)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            Z[4 * TB + t] = rr_mask_gate<KS>(mwt, 4 * TB + t, acc[jo][r]);
            if (t == 3 || t == NT - 1) save_dz(lp, jo, Z);
        };
        {   // top product: dz_2 = mask_2 * (W_out^T dy), finished at once
            float at[4][NB];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) at[e][jo] = sWt[(e * 8 + jo) * 64 + lane];
#pragma unroll
            for (int jo = 0; jo < NB; ++jo)
                acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[0][jo], dy[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int e = 1; e < 4; ++e)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo)
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[e][jo], dy[e], acct[jo], 0, 0, 0);
        }
        auto pre_top = [&](int ks) __attribute__((always_inline)) {
            const int jo = (ks < 4 * (NB - 1)) ? (ks >> 2) : NB - 1, r = ks - 4 * jo;
            Za[ks] = rr_mask_gate<KS>(mwt, ks, acct[jo][r]);
            if (r == ((jo < NB - 1) ? 3 : R - 1)) save_dz(2, jo, Za);
        };
        auto prod = [&](auto pc, float (&Zin)[KS], float (&Zout)[KS]) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            constexpr int lo = 2 - p;                             // the layer whose dz this product yields
            mwt = mw;
            mw = (lo == 1) ? mreg1 : mreg0;
            __builtin_amdgcn_sched_barrier(0);
            const int cur = bbase + lo * S::LAYER_BYTES;              // fragments of layer lo + 1 sit at index lo
            // behind the last backward product the stream goes on with the next stage's first forward layer
            const int nxt = (lo >= 1) ? cur - S::LAYER_BYTES : fbase;
            gemm.run(acc, zero, Zin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) {
                         if (p == 1) pre_top(ks);
                         else pre_tail_b(lo + 1, Zin, ks);
                     },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         Zout[4 * jo + r] = rr_mask_gate<KS>(mw, 4 * jo + r, acc[jo][r]);
                         if (r == 3) save_dz(lo, jo, Zout);
                     },
                     [&]() __attribute__((always_inline)) {});
        };
        prod(I1{}, Za, Zb);
        prod(I2{}, Zb, Za);
        {   // dX = W_0^T dz_0 (times in_isig): lane (q, row) holds input columns 4 q + r of its row: [k_ay | k_ac] of the stage
            mwt = mw;
            const f32x4 o = RRGemm<S>::block(w0t, Za, [&](int ks) __attribute__((always_inline)) { pre_tail_b(0, Za, ks); });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                if (i < idim) sKZ[(st * TILE + m) * WP + ns + i] = o[r] * x_isig[r];
            }
        }
    }

    // ---- this wave's rows of the results, in one burst: the evaluated stages' derivatives, the step result, the error estimate
    for (int idx = lane; idx < (L.st_hi - L.st_lo) * 16 * W; idx += 64) {
        const int j = L.st_lo + idx / (16 * W), rem = idx % (16 * W);
        const int mm = 16 * half + rem / W, c = rem % W;
        if (sLive[mm] != 0.f) L.KZ[((long)j * n + row0 + mm) * W + c] = sKZ[(j * TILE + mm) * WP + c];
    }
    if (L.Z1 || L.ERR)
        for (int idx = lane; idx < 16 * W; idx += 64) {
            const int mm = 16 * half + idx / W, c = idx % W, row = row0 + mm;
            if (sLive[mm] == 0.f) continue;
            const float h = sH[mm];
            if (L.Z1) {
                float a = sZ0[mm * WP + c];
                for (int j = 0; j < L.n_out; ++j)
                    if (L.c_out[j] != 0.f) a = a + sKZ[(j * TILE + mm) * WP + c] * (L.c_out[j] * h);
                L.Z1[(long)row * W + c] = a;
            }
            if (L.ERR) {
                float a = 0.f;
                for (int j = 0; j < L.n_err; ++j)
                    if (L.c_err[j] != 0.f) a = a + sKZ[(j * TILE + mm) * WP + c] * (L.c_err[j] * h);
                L.ERR[(long)row * W + c] = a;
            }
            if (L.ip_out) {       // (node_adj_rr_kernels.hip: the interpolant at t_end on z1 = the value written to Z1 above)
                const int p = row / L.rpp;
                const double t = L.ctl[(long)p * NLBAC_DOPRI_CTL + C_T], hd = L.ctl[(long)p * NLBAC_DOPRI_CTL + C_H];
                if (t + hd >= L.t_end) {
                    const float x = (float)((L.t_end - t) / hd);
                    const float a0 = sZ0[mm * WP + c];
                    float a1 = a0, k[7];
                    for (int j = 0; j < L.n_out; ++j)
                        if (L.c_out[j] != 0.f) a1 = a1 + sKZ[(j * TILE + mm) * WP + c] * (L.c_out[j] * h);
#pragma unroll
                    for (int j = 0; j < 7; ++j) k[j] = sKZ[(j * TILE + mm) * WP + c];
                    L.ip_out[(long)row * W + c] = dopri_interp_value(a0, a1, k, h, x);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
static bool cadj_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_CONCAT_ADJ_RR"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" int nlbac_concat_adj_step_ok(const nlbac_mlp* net) {
    return (net && cadj_enabled() && nlbac_concat_rr_eligible(net) && net->in_dim <= CADJ_MAX_IN) ? 1 : 0;
}

extern "C" int nlbac_concat_adj_step(const nlbac_mlp* net, const float* c, int P, int rows_per_problem, int st_lo, int st_hi,
                                     int n_stages_total, const float* beta, const float* c_out, int n_out,
                                     const float* c_err, int n_err, const float* h_host, const double* h_dev,
                                     int h_dev_stride, const double* ctl, const float* Z0, float* KZ, float* Z1, float* ERR,
                                     const float* norm, float* Xin, float* Ay, float* acts, long acts_ls, float* dz,
                                     float* interp_out, double t_end, nlbac_stream_t s) {
    NLBAC_REQUIRE(net && P >= 1 && P <= 8 && rows_per_problem >= 1, "nlbac_concat_adj_step: bad problem sizes");
    NLBAC_REQUIRE(nlbac_concat_adj_step_ok(net),
                  "nlbac_concat_adj_step: the net is not in -> hid -> hid -> hid -> out with hid in {64, 100, 128} "
                  "(nlbac_concat_adj_step_ok; other shapes run stage by stage: nlbac_concat_adj_in / _out)");
    NLBAC_REQUIRE(n_stages_total >= 1 && n_stages_total <= CK_MAX_STAGES && st_lo >= 0 && st_lo < st_hi &&
                      st_hi <= n_stages_total, "nlbac_concat_adj_step: bad stage range");
    NLBAC_REQUIRE(c && Z0 && KZ && beta, "nlbac_concat_adj_step: null pointer");
    NLBAC_REQUIRE(h_dev || h_host, "nlbac_concat_adj_step: no step size");
    NLBAC_REQUIRE(n_out <= n_stages_total && n_err <= n_stages_total && (!Z1 || c_out) && (!ERR || c_err),
                  "nlbac_concat_adj_step: bad coefficient counts");
    const bool keep = Xin != nullptr;
    NLBAC_REQUIRE(keep ? (Ay && acts && dz && acts_ls > 0) : (!Ay && !acts && !dz),
                  "nlbac_concat_adj_step: Xin / Ay / acts / dz go together");
    ConcatAdjLaunch L;
    memset(&L, 0, sizeof(L));
    L.net = *net;
    L.c = c; L.Z0 = Z0; L.KZ = KZ; L.Z1 = Z1; L.ERR = ERR;
    L.Xin = Xin; L.Ay = Ay; L.acts = acts; L.dz = dz; L.acts_ls = acts_ls;
    L.norm = norm;
    L.n = P * rows_per_problem; L.rpp = rows_per_problem;
    L.n_s = net->out_dim; L.n_c = net->in_dim - net->out_dim;
    NLBAC_REQUIRE(L.n_s >= 1 && L.n_s <= CK_NS && L.n_c >= 0 && L.n_c <= CK_NC, "nlbac_concat_adj_step: net is not [x (<=16) | carried (<=4)] -> dx");
    L.W = 2 * L.n_s + L.n_c;
    L.WP = L.W | 1;                       // odd LDS row stride: the lanes of a quarter hit distinct banks
    L.st_lo = st_lo; L.st_hi = st_hi; L.S_total = n_stages_total;
    for (int i = 0; i < n_stages_total; ++i)
        for (int j = 0; j < n_stages_total; ++j) L.beta[i][j] = beta[i * n_stages_total + j];
    for (int j = 0; j < n_out && c_out; ++j) L.c_out[j] = c_out[j];
    for (int j = 0; j < n_err && c_err; ++j) L.c_err[j] = c_err[j];
    L.n_out = Z1 ? n_out : 0; L.n_err = ERR ? n_err : 0;
    L.h_dev = h_dev; L.h_stride = h_dev_stride;
    for (int p = 0; p < P; ++p) L.h_val[p] = h_host ? h_host[p] : 0.f;
    L.ctl = ctl;
    if (interp_out) {
        NLBAC_REQUIRE(ctl && h_dev && Z1 && st_hi == n_stages_total && n_stages_total == 7,
                      "nlbac_concat_adj_step: interp_out goes with an attempt launch of a device-driven dopri5 solve");
        L.ip_out = interp_out; L.t_end = t_end;
    }
    using Kernel = void (*)(const ConcatAdjLaunch);
    static const Kernel kt[2][3][2] = {{{concat_adj_rr_kernel<4, 4, 0, 2>, concat_adj_rr_kernel<4, 4, 1, 2>},
                                        {concat_adj_rr_kernel<7, 1, 0, 2>, concat_adj_rr_kernel<7, 1, 1, 2>},
                                        {concat_adj_rr_kernel<8, 4, 0, 2>, concat_adj_rr_kernel<8, 4, 1, 2>}},
                                       {{concat_adj_rr_kernel<4, 4, 0, 4>, concat_adj_rr_kernel<4, 4, 1, 4>},
                                        {concat_adj_rr_kernel<7, 1, 0, 4>, concat_adj_rr_kernel<7, 1, 1, 4>},
                                        {concat_adj_rr_kernel<8, 4, 0, 4>, concat_adj_rr_kernel<8, 4, 1, 4>}}};
    const int shape = net->hid == 64 ? 0 : (net->hid == 100 ? 1 : 2);
    // (rows of done problems are skipped per row, nothing else is per tile: a tile may straddle problems)
    static const int forced_nw = [] { const char* e = getenv("NLBAC_CONCAT_NW"); return e ? atoi(e) : 0; }();
    const int nw = forced_nw == 2 ? 2 : 4, tile = 16 * nw;
    const size_t lds = (size_t)((n_stages_total + 1) * tile * L.WP + tile * (CK_NC + 2) + 2 * 4 * 8 * 64) * sizeof(float);
    const Kernel k = kt[nw == 4][shape][keep ? 1 : 0];
    if (lds > 64 * 1024) {
        static bool attr_set[2][3][2] = {};
        if (!attr_set[nw == 4][shape][keep ? 1 : 0]) {
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
            attr_set[nw == 4][shape][keep ? 1 : 0] = true;
        }
    }
    hipLaunchKernelGGL(k, dim3(nlbac_ceil_div(L.n, tile)), dim3(64 * nw), lds, (hipStream_t)s, L);
    NLBAC_CHECK_LAUNCH("nlbac_concat_adj_step");
    return 0;
}
