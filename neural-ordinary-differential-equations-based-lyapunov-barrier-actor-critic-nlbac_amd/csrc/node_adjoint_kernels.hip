// Continuous adjoint of the control-affine NODE  dx/dt = F(x, u) = f(x) + g(x) u  (odeint_adjoint): the backward-in-
// time solve of the augmented state  z = [y | a_x | a_u]  per row,
//     dy/ds = -F(y, u),   da_x/ds = (dF/dx)^T a_x,   da_u/ds = g(y)^T a_x        (s = t1 - t),
// with the parameter adjoint  d theta_bar/ds = (dF/dtheta)^T a_x  as a quadrature beside it.  Nothing of the forward
// solve is kept except y(t1): every stage RE-COMPUTES f_net / g_net on its own stage input and runs their data
// backward straight away, in ONE launch per RK step.  This is what torchdiffeq's OdeintAdjointMethod does with
// autograd.grad inside augmented_dynamics (adjoint.py; the reference pins torchdiffeq==0.2.3, README.md:33, and would
// call it at P/sac_cbf_clf/sac_cbf_clf.py:459,499,534 / P/sac_cbf_clf/model.py:259 — BASELINE configs[3]).
//
// CDNA4 mapping: the workgroup of node_kernels.hip — 512 threads, waves 0-3 own f_net, waves 4-7 own g_net, each
// group with its own LDS ping-pong tile and its own group barrier; weights stream from the L2-resident fragment packs.
// A stage is  [stage input] -> forward chain (activations stay in LDS, ReLU masks as bit words in LDS) -> k_y = -F,
// cotangent a_x -> top layer -> backward chain -> dX = J^T a_x -> k_a.  HBM traffic per row and stage: nothing in mask
// mode (the adjoint of a rollout: 2W floats in, 2W out per STEP); with parameter gradients wanted the activations and
// pre-activation gradients of the stage go out once for nlbac_mlp_bwd_weights (memory O(one step), not O(steps)).
#include "node_adj_shared.h"

// MODE 1: both nets <= 4 column tiles, 2: both 8, 0: mixed.  KEEP 0: ReLU masks in LDS (no weight gradients),
// 1: activations / dz / stage inputs to global memory for nlbac_mlp_bwd_weights.
template <int MODE, int KEEP>
__global__ __launch_bounds__(512) void node_adj_kernel(const NodeAdjLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BITS = KEEP ? 0 : 1;
    const int tid = threadIdx.x, t = tid & 255;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int lane = t & 63, wave = t >> 6;
    __shared__ unsigned s_gcnt[2];
    __shared__ int s_any;
    if (tid < 2) s_gcnt[tid] = 0u;
    if (tid == 2) s_any = 0;
    GroupBar gbar{&s_gcnt[grp], 0u, 4u};
    const int n = L.n, ns = L.n_s, nu = L.n_u, W = L.W, LD = L.ld, gout = ns * nu;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    const nlbac_mlp& net = L.net[grp];
    const int hid = net.hid, NT = pad32(hid) >> 5, hidp32 = NT * 32;
    const int nwide = net.n_layers - 1;
    const int inp = pad8(ns);
    const int n_rows = min(NLBAC_MLP_TILE, n - row0);
    const bool active = wave < NT, two = (MODE == 2) || (MODE == 0 && (wave + 4) < NT);

    // LDS carve (all offsets multiples of 16 B)
    float* buf = smem + grp * 2 * NLBAC_MLP_TILE * LD;                    // this group's ping-pong tiles
    float* sKZ = smem + 4 * NLBAC_MLP_TILE * LD;                          // [stage][32][WP]
    float* sZ0 = sKZ + ADJ_MAX_STAGES * NLBAC_MLP_TILE * ADJ_WP;          // [32][WP]
    float* sZS = sZ0 + NLBAC_MLP_TILE * ADJ_WP;                           // [32][WP] stage input
    float* sU = sZS + NLBAC_MLP_TILE * ADJ_WP;                            // [32][4]
    float* sH = sU + NLBAC_MLP_TILE * ADJ_MAX_NU;                         // [32]
    float* sLive = sH + NLBAC_MLP_TILE;                                   // [32] 1 = row exists and its problem is not done
    float* sF = sLive + NLBAC_MLP_TILE;                                   // [32][8]
    float* sG = sF + NLBAC_MLP_TILE * ADJ_MAX_NS;                         // [32][32]
    float* sDX = sG + NLBAC_MLP_TILE * ADJ_MAX_GOUT;                      // [2][32][8]
    float* sdy_all = sDX + 2 * NLBAC_MLP_TILE * ADJ_MAX_NS;               // [2][32][16]
    float* sdy = sdy_all + grp * NLBAC_MLP_TILE * 16;
    float* sMaskF = sdy_all + 2 * NLBAC_MLP_TILE * 16 + grp * L.mask_words;   // uint32 words, [layer][32][NT]
    float* sW = sdy_all + 2 * NLBAC_MLP_TILE * 16 + 2 * L.mask_words + (grp ? L.sw_off1 : 0);
    float* sBl = sW + net.out_dim * hid;                                  // output-layer bias
    float* sW0t = sBl + ((net.out_dim + 3) & ~3);                         // W_0^T [in][hid]

    {   // constants of the launch -> LDS
        const float* Wl = net.params + net.w_off[nwide];
        const float* bl = net.params + net.b_off[nwide];
        const float* W0 = net.params + net.w_off[0];
        for (int idx = t; idx < net.out_dim * hid; idx += 256) sW[idx] = Wl[idx];
        for (int idx = t; idx < net.out_dim; idx += 256) sBl[idx] = bl[idx];
        for (int idx = t; idx < net.in_dim * hid; idx += 256) {
            const int i = idx / hid, k = idx - i * hid;
            sW0t[idx] = W0[(long)k * net.in_dim + i];
        }
    }
    for (int idx = tid; idx < NLBAC_MLP_TILE * ADJ_WP; idx += 512) {
        const int m = idx / ADJ_WP, c = idx - m * ADJ_WP, row = row0 + m;
        sZ0[idx] = (row < n && c < W) ? L.Z0[(long)row * W + c] : 0.f;
    }
    for (int idx = tid; idx < NLBAC_MLP_TILE * ADJ_MAX_NU; idx += 512) {
        const int m = idx >> 2, c = idx & 3, row = row0 + m;
        sU[idx] = (row < n && c < nu) ? L.u[(long)row * nu + c] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE) {
        const int row = row0 + tid, p = min(row, n - 1) / L.rpp;
        sH[tid] = L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p];
        const bool live = row < n && !(L.ctl && L.ctl[(long)p * NLBAC_DOPRI_CTL + C_DONE] > 0.0);
        sLive[tid] = live ? 1.f : 0.f;
        if (live) s_any = 1;
    }
    for (int idx = tid; idx < L.st_lo * NLBAC_MLP_TILE * ADJ_WP; idx += 512) {
        const int j = idx / (NLBAC_MLP_TILE * ADJ_WP), rem = idx - j * NLBAC_MLP_TILE * ADJ_WP;
        const int m = rem / ADJ_WP, c = rem - m * ADJ_WP, row = row0 + m;
        sKZ[idx] = (row < n && c < W) ? L.KZ[((long)j * n + row) * W + c] : 0.f;
    }
    __syncthreads();
    if (!s_any) return;                      // (uniform) every problem of this tile has finished its solve

    WaveGemm<(MODE == 1) ? 1 : 2> wg2;
    WaveGemm<1> wg1;
    if (active) {
        if constexpr (MODE != 1) { if (two) fwd_prime<2>(wg2, net, inp, false, wave, lane); }
        if constexpr (MODE != 2) { if (!two) fwd_prime<1>(wg1, net, inp, false, wave, lane); }
    }

    for (int st = L.st_lo; st < L.st_hi; ++st) {
        // ---- stage input  Z_st = Z0 + h sum_j beta[st][j] K_j  (all of z: y feeds the nets, a_x is the cotangent)
        for (int idx = tid; idx < NLBAC_MLP_TILE * ADJ_WP; idx += 512) {
            const int m = idx / ADJ_WP, c = idx - m * ADJ_WP;
            float a = sZ0[idx];
            const float h = sH[m];
            for (int j = 0; j < st; ++j)
                if (L.beta[st][j] != 0.f) a = a + sKZ[(j * NLBAC_MLP_TILE + m) * ADJ_WP + c] * (L.beta[st][j] * h);
            sZS[idx] = a;
            if (KEEP && c < W && sLive[m] != 0.f) L.ZS[((long)st * n + row0 + m) * W + c] = a;
        }
        __syncthreads();
        float* in = buf;
        float* out = buf + NLBAC_MLP_TILE * LD;
        for (int idx = t; idx < NLBAC_MLP_TILE * inp; idx += 256) {
            const int m = idx / inp, c = idx - m * inp;
            in[m * LD + c] = (c < ns) ? sZS[m * ADJ_WP + c] : 0.f;
        }
        tile_sync(&gbar, lane);

        // ---- forward chains of f_net (group 0) and g_net (group 1); masks / activations of this stage only
        float* acts_tile = KEEP ? L.acts[grp] + ((long)st * n + row0) * hid : sMaskF;
        const long acts_ls = KEEP ? L.acts_ls[grp] : (long)NLBAC_MLP_TILE * NT;
        if constexpr (MODE == 2)
            fwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, inp, in, out, acts_tile, acts_ls, n_rows, nwide, false, nullptr, &gbar);
        else if constexpr (MODE == 1)
            fwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, inp, in, out, acts_tile, acts_ls, n_rows, nwide, false, nullptr, &gbar);
        else {
            if (two) fwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, inp, in, out, acts_tile, acts_ls, n_rows, nwide, false, nullptr, &gbar);
            else fwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, inp, in, out, acts_tile, acts_ls, n_rows, nwide, false, nullptr, &gbar);
        }
        // the weight stream turns round: the backward packs' first chunks are requested now and land under the output
        // layers and the cotangent fill below
        if (active && nwide >= 2) {
            if constexpr (MODE != 1) { if (two) bwd_prime<2>(wg2, net, wave, lane, false); }
            if constexpr (MODE != 2) { if (!two) bwd_prime<1>(wg1, net, wave, lane, false); }
        }

        // ---- skinny output layers -> sF / sG
        for (int idx = t; idx < NLBAC_MLP_TILE * net.out_dim; idx += 256) {
            const int m = idx & 31, o = idx >> 5;
            const float val = skinny_row_dot(in + m * LD, sW + o * hid, hid) + sBl[o];
            (grp == 0 ? sF + m * ADJ_MAX_NS : sG + m * ADJ_MAX_GOUT)[o] = val;
        }
        __syncthreads();

        // ---- k_y = -(f + g u) ; k_au = g^T a_x ; cotangents of the two output layers
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += 512) {
            const int m = idx / ns, r = idx - m * ns;
            float a = sF[m * ADJ_MAX_NS + r];
            for (int c = 0; c < nu; ++c) a += sG[m * ADJ_MAX_GOUT + r * nu + c] * sU[m * ADJ_MAX_NU + c];
            sKZ[(st * NLBAC_MLP_TILE + m) * ADJ_WP + r] = -a;
        }
        for (int idx = tid; idx < NLBAC_MLP_TILE * nu; idx += 512) {
            const int m = idx / nu, c = idx - m * nu;
            float a = 0.f;
            for (int r = 0; r < ns; ++r) a += sG[m * ADJ_MAX_GOUT + r * nu + c] * sZS[m * ADJ_WP + ns + r];
            sKZ[(st * NLBAC_MLP_TILE + m) * ADJ_WP + 2 * ns + c] = a;
        }
        constexpr int TOP_RPT = (MODE == 1) ? 16 : 32;
        float av_top[TOP_RPT];
        node_top_masks<TOP_RPT, BITS>(acts_tile + (long)(nwide - 1) * acts_ls, hid, NT, t, n_rows, av_top);
        for (int rem = t; rem < NLBAC_MLP_TILE * 16; rem += 256) {
            const int m = rem >> 4, o = rem & 15;
            float v = 0.f;
            if (grp == 0) {
                if (o < ns) v = sZS[m * ADJ_WP + ns + o];
            } else if (o < gout) {
                v = sZS[m * ADJ_WP + ns + o / nu] * sU[m * ADJ_MAX_NU + o % nu];
                if (KEEP && sLive[m] != 0.f) L.dG[((long)st * n + row0 + m) * gout + o] = v;
            }
            sdy[rem] = v;
        }
        tile_sync(&gbar, lane);

        // ---- top (skinny) layer of the backward, then the wide layers
        node_top_layer<TOP_RPT, BITS>(sdy, sW, net.out_dim, hid, hidp32, NT, t, n_rows, av_top, in, LD);
        tile_sync(&gbar, lane);
        if constexpr (KEEP != 0)
            tile_to_global(in, LD, L.dz[grp] + (long)(nwide - 1) * acts_ls + ((long)st * n + row0) * hid, hid, n_rows, t, 256);
        {
            float* dz_tile = KEEP ? L.dz[grp] + ((long)st * n + row0) * hid : nullptr;
            if constexpr (MODE == 2)
                bwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, acts_ls, n_rows, n_rows - 1, nwide - 1, false, 256, &gbar);
            else if constexpr (MODE == 1)
                bwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, acts_ls, n_rows, n_rows - 1, nwide - 1, false, 256, &gbar);
            else {
                if (two) bwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, acts_ls, n_rows, n_rows - 1, nwide - 1, false, 256, &gbar);
                else bwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, acts_ls, n_rows, n_rows - 1, nwide - 1, false, 256, &gbar);
            }
        }
        // (the next stage's forward packs are requested before the small phases below)
        if (active && st + 1 < L.st_hi) {
            if constexpr (MODE != 1) { if (two) fwd_prime<2>(wg2, net, inp, false, wave, lane); }
            if constexpr (MODE != 2) { if (!two) fwd_prime<1>(wg1, net, inp, false, wave, lane); }
        }

        // ---- dX = dz0 W_0 (one dot product per thread), k_ax = dX_f + dX_g
        for (int idx = t; idx < NLBAC_MLP_TILE * ns; idx += 256) {
            const int m = idx & 31, i = idx >> 5;
            sDX[(grp * NLBAC_MLP_TILE + m) * ADJ_MAX_NS + i] = skinny_row_dot(in + m * LD, sW0t + i * hid, hid);
        }
        __syncthreads();
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += 512) {
            const int m = idx / ns, c = idx - m * ns;
            sKZ[(st * NLBAC_MLP_TILE + m) * ADJ_WP + ns + c] =
                sDX[m * ADJ_MAX_NS + c] + sDX[(NLBAC_MLP_TILE + m) * ADJ_MAX_NS + c];
        }
        __syncthreads();
        for (int idx = tid; idx < NLBAC_MLP_TILE * W; idx += 512) {
            const int m = idx / W, c = idx - m * W;
            if (sLive[m] != 0.f) L.KZ[((long)st * n + row0 + m) * W + c] = sKZ[(st * NLBAC_MLP_TILE + m) * ADJ_WP + c];
        }
    }

    // ---- step outputs
    for (int idx = tid; idx < NLBAC_MLP_TILE * W; idx += 512) {
        const int m = idx / W, c = idx - m * W, row = row0 + m;
        if (sLive[m] == 0.f) continue;
        const float h = sH[m];
        if (L.Z1) {
            float a = sZ0[m * ADJ_WP + c];
            for (int j = 0; j < L.n_out; ++j)
                if (L.c_out[j] != 0.f) a = a + sKZ[(j * NLBAC_MLP_TILE + m) * ADJ_WP + c] * (L.c_out[j] * h);
            L.Z1[(long)row * W + c] = a;
        }
        if (L.ERR) {
            float a = 0.f;
            for (int j = 0; j < L.n_err; ++j)
                if (L.c_err[j] != 0.f) a = a + sKZ[(j * NLBAC_MLP_TILE + m) * ADJ_WP + c] * (L.c_err[j] * h);
            L.ERR[(long)row * W + c] = a;
        }
    }
}

extern "C" int nlbac_node_adj_step(const nlbac_mlp* f, const nlbac_mlp* g, const float* u, int P, int rows_per_problem,
                                   int st_lo, int st_hi, int n_stages_total, const float* beta, const float* c_out,
                                   int n_out, const float* c_err, int n_err, const float* h_host, const double* h_dev,
                                   int h_dev_stride, const double* ctl, const float* Z0, float* KZ, float* Z1,
                                   float* ERR, float* ZS, float* dG, float* acts_f, long acts_f_ls, float* acts_g,
                                   long acts_g_ls, float* dz_f, float* dz_g, float* interp_out, double t_end,
                                   nlbac_stream_t s) {
    NLBAC_REQUIRE(f && g && u && Z0 && KZ, "nlbac_node_adj_step: null pointer");
    NLBAC_REQUIRE(P >= 1 && P <= 8 && rows_per_problem >= 1, "nlbac_node_adj_step: bad problem sizes");
    NLBAC_REQUIRE(n_stages_total >= 1 && n_stages_total <= ADJ_MAX_STAGES && st_lo >= 0 && st_lo < st_hi &&
                      st_hi <= n_stages_total, "nlbac_node_adj_step: bad stage range");
    NLBAC_REQUIRE(f->in_dim == g->in_dim && f->in_dim <= ADJ_MAX_NS && f->out_dim == f->in_dim &&
                      g->out_dim % f->in_dim == 0 && g->out_dim / f->in_dim <= ADJ_MAX_NU && g->out_dim <= 16,
                  "nlbac_node_adj_step: f/g shapes are not a supported control-affine field");
    NLBAC_REQUIRE(f->hid % 4 == 0 && g->hid % 4 == 0 && f->hid <= 256 && g->hid <= 256, "nlbac_node_adj_step: bad hid");
    NLBAC_REQUIRE(f->n_layers >= 3 && g->n_layers >= 3, "nlbac_node_adj_step: nets need at least two wide layers");
    NLBAC_REQUIRE(h_dev || h_host, "nlbac_node_adj_step: no step size");
    NLBAC_REQUIRE(n_out <= n_stages_total && n_err <= n_stages_total, "nlbac_node_adj_step: bad coefficient counts");
    const bool keep = ZS != nullptr;
    NLBAC_REQUIRE(keep == (dG != nullptr) && keep == (acts_f != nullptr) && keep == (acts_g != nullptr) &&
                      keep == (dz_f != nullptr) && keep == (dz_g != nullptr),
                  "nlbac_node_adj_step: ZS, dG, acts_f, acts_g, dz_f, dz_g go together");
    NodeAdjLaunch L;
    memset(&L, 0, sizeof(L));
    L.net[0] = *f; L.net[1] = *g;
    L.u = u; L.Z0 = Z0; L.KZ = KZ; L.Z1 = Z1; L.ERR = ERR; L.ZS = ZS; L.dG = dG;
    L.acts[0] = acts_f; L.acts[1] = acts_g; L.acts_ls[0] = acts_f_ls; L.acts_ls[1] = acts_g_ls;
    L.dz[0] = dz_f; L.dz[1] = dz_g;
    L.n = P * rows_per_problem; L.rpp = rows_per_problem;
    L.n_s = f->in_dim; L.n_u = g->out_dim / f->in_dim; L.W = 2 * L.n_s + L.n_u;
    L.st_lo = st_lo; L.st_hi = st_hi; L.S_total = n_stages_total;
    if (beta)
        for (int i = 0; i < n_stages_total; ++i)
            for (int j = 0; j < n_stages_total; ++j) L.beta[i][j] = beta[i * n_stages_total + j];
    for (int j = 0; j < n_out; ++j) L.c_out[j] = c_out[j];
    for (int j = 0; j < n_err; ++j) L.c_err[j] = c_err[j];
    L.n_out = Z1 ? n_out : 0; L.n_err = ERR ? n_err : 0;
    L.h_dev = h_dev; L.h_stride = h_dev_stride; L.ctl = ctl;
    for (int p = 0; p < P; ++p) L.h_val[p] = h_host ? h_host[p] : 0.f;
    const int w = ((f->hid > g->hid ? f->hid : g->hid) + 31) & ~31;
    L.ld = w + 4;
    auto sw_of = [](const nlbac_mlp* m) { return (m->out_dim * m->hid + ((m->out_dim + 3) & ~3) + m->in_dim * m->hid + 3) & ~3; };
    L.sw_off1 = sw_of(f);
    const int sw_total = sw_of(f) + sw_of(g);
    const int ntf = (f->hid + 31) >> 5, ntg = (g->hid + 31) >> 5;
    const int nwmax = (f->n_layers > g->n_layers ? f->n_layers : g->n_layers) - 1;
    L.mask_words = keep ? 0 : ((nwmax * NLBAC_MLP_TILE * (ntf > ntg ? ntf : ntg) + 3) & ~3);
    const size_t lds = ((size_t)4 * NLBAC_MLP_TILE * L.ld + (ADJ_MAX_STAGES + 2) * NLBAC_MLP_TILE * ADJ_WP +
                        NLBAC_MLP_TILE * (ADJ_MAX_NU + 1 + 1 + ADJ_MAX_NS + ADJ_MAX_GOUT + 2 * ADJ_MAX_NS + 2 * 16) +
                        2 * L.mask_words + sw_total) * sizeof(float);
    NLBAC_REQUIRE(lds <= ADJ_LDS_MAX, "nlbac_node_adj_step: LDS budget exceeded (%zu B)", lds);
    using Kernel = void (*)(const NodeAdjLaunch);
    static const Kernel k[2][3] = {{node_adj_kernel<0, 0>, node_adj_kernel<1, 0>, node_adj_kernel<2, 0>},
                                   {node_adj_kernel<0, 1>, node_adj_kernel<1, 1>, node_adj_kernel<2, 1>}};
    static bool attr_set = false;
    if (!attr_set) {
        for (int b = 0; b < 2; ++b)
            for (int m = 0; m < 3; ++m)
                (void)hipFuncSetAttribute((const void*)k[b][m], hipFuncAttributeMaxDynamicSharedMemorySize, ADJ_LDS_MAX);
        attr_set = true;
    }
    if (interp_out) {
        NLBAC_REQUIRE(ctl && h_dev && Z1 && st_hi == n_stages_total && n_stages_total == 7,
                      "nlbac_node_adj_step: interp_out goes with an attempt launch of a device-driven dopri5 solve");
        L.ip_out = interp_out; L.t_end = t_end;
    }
    {       // the reference's NODE shapes: the register-resident kernel (node_adj_rr_kernels.hip), with or without kept rows
        const int rr = nlbac_node_adj_rr_launch(L, (hipStream_t)s);
        if (rr <= 0) return rr;
    }
    NLBAC_REQUIRE(!interp_out, "nlbac_node_adj_step: interp_out needs the register-resident kernel (nlbac_node_adj_interp_ok)");
    const int mode = (ntf <= 4 && ntg <= 4) ? 1 : ((ntf == 8 && ntg == 8) ? 2 : 0);
    const dim3 grid(nlbac_ceil_div(L.n, NLBAC_MLP_TILE));
    hipLaunchKernelGGL(k[keep ? 1 : 0][mode], grid, dim3(512), lds, (hipStream_t)s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_adj_step");
    return 0;
}

// ---------------------------------------------------------------------------
// Per-row glue of the adjoint solve: z = [y | a_x | a_u] packing, the mixed error norm + step controller, the
// accepted-step hand-over.  One lane per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adj_pack_kernel(const float* y, const float* ax, int ns, int nu, int n, float* Z) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int W = 2 * ns + nu;
    for (int c = 0; c < ns; ++c) {
        Z[(long)i * W + c] = y[(long)i * ns + c];
        Z[(long)i * W + ns + c] = ax[(long)i * ns + c];
    }
    for (int c = 0; c < nu; ++c) Z[(long)i * W + 2 * ns + c] = 0.f;
}

__global__ __launch_bounds__(256) void adj_unpack_kernel(const float* Z, int ns, int nu, int n, float* dy0, float* du) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int W = 2 * ns + nu;
    if (dy0) for (int c = 0; c < ns; ++c) dy0[(long)i * ns + c] = Z[(long)i * W + ns + c];
    if (du) for (int c = 0; c < nu; ++c) du[(long)i * nu + c] = Z[(long)i * W + 2 * ns + c];
}

extern "C" int nlbac_adj_pack(const float* y, const float* a_x, int n_s, int n_u, int n, float* Z, nlbac_stream_t s) {
    NLBAC_REQUIRE(y && a_x && Z && n >= 1, "nlbac_adj_pack: bad arguments");
    hipLaunchKernelGGL(adj_pack_kernel, dim3(nlbac_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)s, y, a_x, n_s, n_u, n, Z);
    NLBAC_CHECK_LAUNCH("nlbac_adj_pack");
    return 0;
}

extern "C" int nlbac_adj_unpack(const float* Z, int n_s, int n_u, int n, float* dy0, float* du, nlbac_stream_t s) {
    NLBAC_REQUIRE(Z && n >= 1, "nlbac_adj_unpack: bad arguments");
    hipLaunchKernelGGL(adj_unpack_kernel, dim3(nlbac_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)s, Z, n_s, n_u, n, dy0, du);
    NLBAC_CHECK_LAUNCH("nlbac_adj_unpack");
    return 0;
}

// Mixed norm of torchdiffeq's default adjoint norm (handle_adjoint_norm_): max(rms over the y part [x | u], rms over
// the adj_y part [a_x | a_u], max_i rms(adj_param_i)) of the scaled quantity.  partials: [P][nblk][4]
//  mode 0: cols (y: (z0/scale)^2, (f0/scale)^2 ; a: same)      a = KZ[0]
//  mode 1: cols (y: ((f1-f0)/scale)^2, - ; a: same)            a = KZ[1], b = KZ[0]
//  mode 2: cols (y: (err/tol)^2, - ; a: same)                  a = ERR, tol = atol + rtol max(|z0|, |z1|)
// The carried controls count as state columns of y with zero derivative / error (as in the reference's [x, u] state).
__device__ __forceinline__ void adj_norm_block(const float* a, const float* b, const float* Z0, const float* Z1,
                                               const float* u, int mode, float rtol, float atol, int ns, int nu,
                                               int rpp, float* partials, bool publish = false) {
    __shared__ float red[16];
    const int p = blockIdx.y, W = 2 * ns + nu;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < rpp) {
        const long row = (long)p * rpp + i;
        for (int c = 0; c < W; ++c) {
            const int part = c < ns ? 0 : 2;
            const float z = Z0[row * W + c];
            if (mode == 2) {
                const float tol = atol + rtol * fmaxf(fabsf(z), fabsf(Z1[row * W + c]));
                const float q = a[row * W + c] / tol;
                v[part] += q * q;
            } else {
                const float sc = atol + fabsf(z) * rtol;
                if (mode == 0) {
                    const float q0 = z / sc, q1 = a[row * W + c] / sc;
                    v[part] += q0 * q0; v[part + 1] += q1 * q1;
                } else {
                    const float q = (a[row * W + c] - b[row * W + c]) / sc;
                    v[part] += q * q;
                }
            }
        }
        if (mode == 0)
            for (int c = 0; c < nu; ++c) {
                const float y = u[row * nu + c];
                const float q = y / (atol + fabsf(y) * rtol);
                v[0] += q * q;
            }
    }
    block_sum_256<4>(v, red);
    float* q = partials + ((long)p * gridDim.x + blockIdx.x) * 4;
    if (!publish) {
        if (threadIdx.x == 0)
            for (int k = 0; k < 4; ++k) q[k] = v[k];
        return;
    }
    // for a reader in another workgroup of THIS launch: the four sums leave as device-scope atomic exchanges, one wave
    // instruction, and have returned when this function does — no agent-scope fence, which on gfx950 writes the XCD's L2
    // back (common.h::publish_and_elect; the fenced form made this launch 13 us against the forward controller's 9)
    if (threadIdx.x < 64) {
        float old = 0.f;
        if (threadIdx.x < 4) {
            const float mine = threadIdx.x == 0 ? v[0] : (threadIdx.x == 1 ? v[1] : (threadIdx.x == 2 ? v[2] : v[3]));
            old = __hip_atomic_exchange(q + threadIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" ::"v"(old) : "memory");
    }
}

__device__ __forceinline__ void adj_control_one(const double (&s)[4], int p, int mode, int ns, int nu, int rpp,
                                                double t_end, const float* pnorm, double* ctl) {
    const double cnt = (double)rpp * (double)(ns + nu);
    double n0 = fmax(sqrt(s[0] / cnt), sqrt(s[2] / cnt)), n1 = fmax(sqrt(s[1] / cnt), sqrt(s[3] / cnt));
    if (pnorm) { n0 = fmax(n0, (double)pnorm[0]); n1 = fmax(n1, (double)pnorm[1]); }
    dopri_control_vals(n0, n1, p, mode, t_end, ctl);
}

// norm + controller in one launch (last-workgroup ticket per problem, as dopri_norm_control_kernel); a problem whose
// solve has finished (C_DONE) is skipped by all of its blocks when an attempted step is judged (mode 2)
__global__ __launch_bounds__(256) void adj_norm_control_kernel(const float* a, const float* b, const float* Z0,
                                                               const float* Z1, const float* u, int mode, float rtol,
                                                               float atol, int ns, int nu, int rpp, double t_end,
                                                               const float* pnorm, float* partials, unsigned* tickets,
                                                               double* ctl, double* ctl_host, double host_seq) {
    const int p = blockIdx.y, nblk = (int)gridDim.x;
    if (mode == 2 && ctl[(long)p * NLBAC_DOPRI_CTL + C_DONE] > 0.0) return;
    adj_norm_block(a, b, Z0, Z1, u, mode, rtol, atol, ns, nu, rpp, partials, tickets != nullptr);
    if (!tickets) return;                              // two-call form (data parallel): nlbac_adj_control follows
    __shared__ unsigned s_last;
    __shared__ double s_red[4][256];
    if (threadIdx.x == 0) {                            // (this block's sums have been performed device-wide: see adj_norm_block)
        elect_release_();
        const unsigned ticket = __hip_atomic_fetch_add(tickets + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (ticket == gridDim.x - 1) ? 1u : 0u;
        if (s_last) __hip_atomic_store(tickets + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    elect_acquire_();
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int bb = threadIdx.x; bb < nblk; bb += 256) {
        const float* q = partials + ((long)p * nblk + bb) * 4;
        for (int k = 0; k < 4; ++k) v[k] += (double)__hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int k = 0; k < 4; ++k) s_red[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 4; ++k) s_red[k][threadIdx.x] += s_red[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double s[4] = {s_red[0][0], s_red[1][0], s_red[2][0], s_red[3][0]};
        adj_control_one(s, p, mode, ns, nu, rpp, t_end, pnorm, ctl);
        // the host's copy of this problem's block (pinned memory the device writes directly, as dopri_norm_control_kernel)
        if (ctl_host) ctl_host_post(ctl_host + (long)p * NLBAC_DOPRI_CTL, ctl + (long)p * NLBAC_DOPRI_CTL, host_seq);
    }
}

// controller alone on (all-reduced) sums [P][nblk][4]; rows_per_problem is the global count
__global__ void adj_control_kernel(const float* partials, int nblk, int mode, int ns, int nu, int rpp, double t_end,
                                   const float* pnorm, double* ctl) {
    if (threadIdx.x != 0) return;
    const int p = blockIdx.x;
    if (mode == 2 && ctl[(long)p * NLBAC_DOPRI_CTL + C_DONE] > 0.0) return;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int bb = 0; bb < nblk; ++bb)
        for (int k = 0; k < 4; ++k) s[k] += (double)partials[((long)p * nblk + bb) * 4 + k];
    adj_control_one(s, p, mode, ns, nu, rpp, t_end, pnorm, ctl);
}

extern "C" int nlbac_adj_norm_control(const float* a, const float* b, const float* Z0, const float* Z1, const float* u,
                                      int mode, float rtol, float atol, int n_s, int n_u, int rows_per_problem, int P,
                                      double t_end, const float* pnorm, float* partials, unsigned* tickets, double* ctl,
                                      double* ctl_host, double host_seq, nlbac_stream_t s) {
    NLBAC_REQUIRE(a && Z0 && partials && ctl && mode >= 0 && mode <= 2, "nlbac_adj_norm_control: bad arguments");
    NLBAC_REQUIRE((mode != 0 || u) && (mode != 1 || b) && (mode != 2 || Z1), "nlbac_adj_norm_control: missing operand");
    NLBAC_REQUIRE(P >= 1 && P <= 8, "nlbac_adj_norm_control: P %d out of range", P);
    hipLaunchKernelGGL(adj_norm_control_kernel, dim3(nlbac_ceil_div(rows_per_problem, 256), P), dim3(256), 0,
                       (hipStream_t)s, a, b, Z0, Z1, u, mode, rtol, atol, n_s, n_u, rows_per_problem, t_end, pnorm,
                       partials, tickets, ctl, tickets ? ctl_host : nullptr, host_seq);
    NLBAC_CHECK_LAUNCH("nlbac_adj_norm_control");
    return 0;
}

extern "C" int nlbac_adj_control(const float* partials, int n_blk_per_problem, int mode, int n_s, int n_u,
                                 int rows_per_problem, int P, double t_end, const float* pnorm, double* ctl,
                                 nlbac_stream_t s) {
    NLBAC_REQUIRE(partials && ctl && mode >= 0 && mode <= 2, "nlbac_adj_control: bad arguments");
    hipLaunchKernelGGL(adj_control_kernel, dim3(P), dim3(64), 0, (hipStream_t)s, partials, n_blk_per_problem, mode, n_s,
                       n_u, rows_per_problem, t_end, pnorm, ctl);
    NLBAC_CHECK_LAUNCH("nlbac_adj_control");
    return 0;
}

// Hand-over after an attempted step, decided on the device: rows of problems whose step was accepted and whose
// solve goes on take z1 as the new z0 and the last stage derivative as the next first one (FSAL).  Generic strided
// form (dst0 <- src0, dst1 <- src1, `w` floats per row) so that the parameter adjoint uses it too.
__global__ __launch_bounds__(256) void adj_commit_kernel(const double* ctl, int rpp, long n, int w, float* dst0,
                                                         const float* src0, float* dst1, const float* src1) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* c = ctl + (i / rpp) * NLBAC_DOPRI_CTL;
    if (!(c[C_ACCEPT] > 0.0) || c[C_DONE] > 0.0) return;
    for (int k = 0; k < w; ++k) {
        dst0[i * w + k] = src0[i * w + k];
        if (dst1) dst1[i * w + k] = src1[i * w + k];
    }
}

extern "C" int nlbac_adj_commit(const double* ctl, int rows_per_problem, long n_rows, int w, float* dst0,
                                const float* src0, float* dst1, const float* src1, nlbac_stream_t s) {
    NLBAC_REQUIRE(ctl && dst0 && src0 && n_rows >= 1 && w >= 1 && rows_per_problem >= 1 && (!dst1 || src1),
                  "nlbac_adj_commit: bad arguments");
    hipLaunchKernelGGL(adj_commit_kernel, dim3(nlbac_ceil_div(n_rows, 256)), dim3(256), 0, (hipStream_t)s, ctl,
                       rows_per_problem, n_rows, w, dst0, src0, dst1, src1);
    NLBAC_CHECK_LAUNCH("nlbac_adj_commit");
    return 0;
}

// ---------------------------------------------------------------------------
// Parameter adjoint  theta_bar  (one flat vector in the arena's layout): a quadrature beside the per-row state.  Its
// stage derivatives K_theta[j] = sum_rows (dF/dtheta)^T a_x at stage j come from nlbac_mlp_bwd_weights on what
// nlbac_node_adj_step kept of that stage.  This kernel forms the step result and the quantity the step-size norm
// needs, per PARAMETER TENSOR (torchdiffeq's _mixed_norm over adj_params: the maximum of the tensors' RMS norms):
//   mode 0: pnorm = (max_i rms(th0_i / scale_i), max_i rms(K[0]_i / scale_i)),  scale = atol + rtol |th0|
//   mode 1: pnorm[0] = max_i rms((K[1] - K[0])_i / scale_i)
//   mode 2: th1 = th0 + h sum_j c_sol[j] K[j];  pnorm[0] = max_i rms((h sum_j c_err[j] K[j])_i / tol_i),
//           tol = atol + rtol max(|th0|, |th1|)
// grid = one block per tensor; the last block to finish (ticket) takes the maximum over the tensors.
// ---------------------------------------------------------------------------
struct AdjParamArg { float c_sol[ADJ_MAX_STAGES]; float c_err[ADJ_MAX_STAGES]; int S; float h_host; };

__global__ __launch_bounds__(256) void adj_param_norm_kernel(int mode, const float* th0, const float* K, long stride,
                                                             const AdjParamArg A, const double* h_dev,
                                                             const int* seg_off, const int* seg_len, int n_seg,
                                                             float rtol, float atol, const double* ctl, float* th1,
                                                             float* pseg, unsigned* ticket, float* pnorm) {
    if (mode == 2 && ctl && ctl[C_DONE] > 0.0) return;
    __shared__ float red[8];
    __shared__ unsigned s_last;
    const int seg = blockIdx.x, off = seg_off[seg], len = seg_len[seg];
    const float h = h_dev ? (float)h_dev[0] : A.h_host;
    float v[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < len; i += 256) {
        const long e = off + i;
        const float z0 = th0[e];
        if (mode == 0) {
            const float sc = atol + fabsf(z0) * rtol;
            const float q0 = z0 / sc, q1 = K[e] / sc;
            v[0] += q0 * q0; v[1] += q1 * q1;
        } else if (mode == 1) {
            const float sc = atol + fabsf(z0) * rtol;
            const float q = (K[stride + e] - K[e]) / sc;
            v[0] += q * q;
        } else {
            float z1 = z0, er = 0.f;
            for (int j = 0; j < A.S; ++j) {
                const float k = K[(long)j * stride + e];
                if (A.c_sol[j] != 0.f) z1 = z1 + k * (A.c_sol[j] * h);
                if (A.c_err[j] != 0.f) er = er + k * (A.c_err[j] * h);
            }
            th1[e] = z1;
            const float q = er / (atol + rtol * fmaxf(fabsf(z0), fabsf(z1)));
            v[0] += q * q;
        }
    }
    block_sum_256<2>(v, red);
    if (threadIdx.x == 0) {
        pseg[2 * seg + 0] = v[0] / (float)len;
        pseg[2 * seg + 1] = v[1] / (float)len;
        __threadfence();
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    float m0 = 0.f, m1 = 0.f;
    for (int k = 0; k < n_seg; ++k) {
        m0 = fmaxf(m0, __hip_atomic_load(pseg + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        m1 = fmaxf(m1, __hip_atomic_load(pseg + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    pnorm[0] = sqrtf(m0); pnorm[1] = sqrtf(m1);
}

extern "C" int nlbac_adj_param_norm(int mode, const float* th0, const float* K, long k_stride, int n_stages,
                                    const float* c_sol, const float* c_err, const float* h_host, const double* h_dev,
                                    const int* seg_off, const int* seg_len, int n_seg, float rtol, float atol,
                                    const double* ctl, float* th1, float* pseg, unsigned* ticket, float* pnorm,
                                    nlbac_stream_t s) {
    NLBAC_REQUIRE(th0 && K && seg_off && seg_len && n_seg >= 1 && pseg && ticket && pnorm && mode >= 0 && mode <= 2,
                  "nlbac_adj_param_norm: bad arguments");
    NLBAC_REQUIRE(mode != 2 || (th1 && c_sol && c_err && n_stages >= 1 && n_stages <= ADJ_MAX_STAGES && (h_host || h_dev)),
                  "nlbac_adj_param_norm: mode 2 needs th1, the coefficients and a step size");
    AdjParamArg A;
    memset(&A, 0, sizeof(A));
    A.S = n_stages;
    for (int j = 0; j < n_stages && mode == 2; ++j) { A.c_sol[j] = c_sol[j]; A.c_err[j] = c_err[j]; }
    A.h_host = h_host ? h_host[0] : 0.f;
    hipLaunchKernelGGL(adj_param_norm_kernel, dim3(n_seg), dim3(256), 0, (hipStream_t)s, mode, th0, K, k_stride, A,
                       h_dev, seg_off, seg_len, n_seg, rtol, atol, ctl, th1, pseg, ticket, pnorm);
    NLBAC_CHECK_LAUNCH("nlbac_adj_param_norm");
    return 0;
}
