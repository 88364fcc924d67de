// Fused Runge-Kutta step of the control-affine NODE  dx/dt = f(x) + g(x) u  with REGISTER-RESIDENT layer chains
// (rr_device.h): the same launches, arguments and results as node_kernels.hip's LDS-tiled kernels — they are selected
// inside nlbac_node_rk_fwd / nlbac_node_rk_bwd for nets up to 128 units wide — but a 32-row tile is worked on by four
// waves, one per SIMD, each running ONE net's whole layer chain for 16 of the rows:
//     wave 0: f_net rows 0-15    wave 1: f_net rows 16-31    wave 2: g_net rows 0-15    wave 3: g_net rows 16-31
// Per stage a wave issues layer 0, its hid x hid layers and the output layer as one uninterrupted MFMA stream
// (v_mfma_f32_16x16x4_f32, weights streamed from the L2-resident RR pack, bias + ReLU + mask bits applied to the
// accumulators in place); the two nets meet at k = f + g u, through LDS and two workgroup barriers per stage.
// The LDS-tiled kernels spend 34k cycles per stage on a 32-row tile (five layer steps of GEMM + epilogue + barrier, the
// pipe's 16.6k cycles of 128-column / K=104 tiles spread over them, profiles/r02_phase_times_node_rk_fwd.txt); here a
// stage is f_net's three 5.6k-cycle layers + ~2k.
//
// Reference call sites: torchdiffeq.odeint at U/sac_cbf_clf/sac_cbf_clf.py:453,577 and U/sac_cbf_clf/model.py:252
// over NeuralODEModel.forward (model.py:208-217); the backward is what autograd does through the solver's stages.
#include "node_rk_shared.h"
#include "rr_device.h"
#include <cstdlib>

// which output the A row hu = 4 q' + r' of the (single) output block computes, so that the result leaves lane (q, row)
// with state component c = 4 r + q in register r (f_net) resp. g[c = 4 ks0 + q][u] in register e = ks0 nu + u (g_net):
// exactly the layout of layer 0's B operand.  -1: padding row.
__device__ __forceinline__ int rr_out_row(int grp, int hu, int ns, int nu) {
    const int qp = hu >> 2, rp = hu & 3, KS0 = (ns + 3) >> 2;
    if (grp == 0) {
        const int c = 4 * rp + qp;
        return (rp < KS0 && c < ns) ? c : -1;
    }
    const int k0 = rp / nu, u = rp - k0 * nu, c = 4 * k0 + qp;
    return (rp < KS0 * nu && c < ns) ? c * nu + u : -1;
}

template <int NB, int R, int BITS>
__global__ __launch_bounds__(256) void node_rr_fwd_kernel(const NodeRkLaunch L) {
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 1, half = wave & 1;
    const int n = L.n, ns = L.n_s, nu = L.n_u, gout = ns * nu;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    RkFwdWhere w;
    if (!rk_fwd_where(L, row0, w)) return;
    RkFwdTile T;
    T.carve(smem);
    float* sYin = smem + RkFwdTile::floats();            // [32][8] the stage input, columns ns..7 zero
    const nlbac_mlp& net = L.net[grp];
    const int nw = net.n_layers - 1;                     // layer 0 + (nw - 1) hid x hid layers, then the output layer
    const int n_rows = min(NLBAC_MLP_TILE, n - row0);
    const int q = lane >> 4, r16 = lane & 15;
    const int m = 16 * half + r16, grow = row0 + m;      // this lane's row: within the tile, global
    const bool row_ok = grow < n;
    const int KS0 = (ns + 3) >> 2;                       // k-steps of layer 0 (1 or 2)

    // ---- the wave's weight stream: hid x hid layers 1 .. nw-1, then layer 1 again (next stage)
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wbase = net.rr_fwd_off * 4;
    RRGemm<S> gemm;
    gemm.prime(rs, voff, wbase);

    // ---- constants of the launch in registers: layer 0's and the output layer's A fragments
    float w0[2][NB];
    {
        const float* W0 = net.params + net.w_off[0];
#pragma unroll
        for (int k0 = 0; k0 < 2; ++k0)
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int uo = rr_unit_out(NB, R, jo, r16), col = 4 * k0 + q;
                w0[k0][jo] = (uo >= 0 && col < ns) ? W0[uo * ns + col] : 0.f;
            }
    }
    float wo[KS];
    const int orow = rr_out_row(grp, r16, ns, nu);
    {
        const float* wrow = net.params + net.w_off[nw] + (long)max(orow, 0) * HID;
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) {
            const f32x4 v = rr_row_load<S>(wrow, jo, q);
#pragma unroll
            for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) wo[4 * jo + r] = (orow >= 0) ? v[r] : 0.f;
        }
    }
    // this lane's outputs of the output layer (register r): where they go in sF / sG (and G), their bias
    int o_idx[4]; float o_bias[4];
    {
        const float* bo = net.params + net.b_off[nw];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int o = -1;
            if (grp == 0) { const int c = 4 * r + q; if (r < KS0 && c < ns) o = c; }
            else { const int k0 = r / nu, u = r - k0 * nu, c = 4 * k0 + q; if (r < KS0 * nu && c < ns) o = c * nu + u; }
            o_idx[r] = o;
            o_bias[r] = (o >= 0) ? bo[o] : 0.f;
        }
    }

    rk_fwd_tile_constants<256>(L, w, T, row0, tid);

    for (int st = L.stage_begin; st < L.stage_end; ++st) {
        if (st == L.stage_begin) {
            rk_fwd_first_input(L, w, T, row0, st, 8, sYin, 8, true, tid, 256);
            __syncthreads();
        }
        float H[KS];
        f32x4 acc[NB];
        // ---- layer 0: K = ns (one or two k-steps), straight from the stage input
        {
            const float y0 = sYin[m * 8 + q], y1 = sYin[m * 8 + 4 + q];
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                acc[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[0][jo], y0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (KS0 > 1) acc[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[1][jo], y1, acc[jo], 0, 0, 0);
            }
        }
        for (int l = 0; l < nw; ++l) {
            if (l > 0) {
                const int cur = wbase + (l - 1) * S::LAYER_BYTES;
                const int nxt = (l + 1 < nw) ? cur + S::LAYER_BYTES : wbase;
                gemm.run(acc, H, rs, voff, cur, nxt);
            }
            // ---- bias + ReLU in place; what the backward needs goes out once: mask bits or the activations
            const float* bias = net.params + net.b_off[l];
            unsigned word = 0u;
            float* arow = (!BITS && L.acts[grp]) ? L.acts[grp] + w.soff + (long)l * L.acts_ls[grp] + ((long)st * n + grow) * HID : nullptr;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const f32x4 b = rr_bias<S>(bias, jo, q);
                f32x4 hv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) {
                    const float h = rr_relu(acc[jo][r] + b[r]);
                    H[4 * jo + r] = h;
                    hv[r] = h;
                    if (BITS) word |= (__builtin_bit_cast(int, h) > 0 ? 1u : 0u) << (4 * jo + r);
                }
                if (!BITS && arow && row_ok) rr_row_store<S>(arow, jo, q, hv);
            }
            if (BITS && L.acts[grp] && row_ok)
                reinterpret_cast<unsigned*>(L.acts[grp] + w.soff + (long)l * L.acts_ls[grp])[((long)st * n + grow) * 4 + q] = word;
        }
        // ---- output layer (<= 16 outputs: one block), to LDS for k = f + g u; g(x) also to global for the backward
        {
            f32x4 o{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) o = __builtin_amdgcn_mfma_f32_16x16x4f32(wo[ks], H[ks], o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (o_idx[r] < 0) continue;
                const float val = o[r] + o_bias[r];
                if (grp == 0) T.sF[m * RK_MAX_NS + o_idx[r]] = val;
                else {
                    T.sG[m * RK_MAX_GOUT + o_idx[r]] = val;
                    if (row_ok) w.gG[((long)st * n + grow) * gout + o_idx[r]] = val;
                }
            }
        }
        __syncthreads();
        rk_fwd_combine<256>(L, w, T, row0, st, 8, sYin, nullptr, 8, tid);
        __syncthreads();
    }
    rk_fwd_outputs_and_control<256>(L, w, T, row0, n_rows, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the same step, same wave roles.  Per stage (descending): the output layer's gradient enters as the B
// operand of one transposed block product, then dz_{l-1} = mask_{l-1} * (W_l^T dz_l) down the chain in registers (the
// backward RR pack), then dX = W_0^T dz_0; the two nets meet in the stage algebra (rk_bwd_stage_algebra).
// ---------------------------------------------------------------------------------------------------------------------
template <int NB, int R, int BITS>
__global__ __launch_bounds__(256) void node_rr_bwd_kernel(const NodeRkBwdLaunch L) {
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 1, half = wave & 1;
    const int n = L.n, ns = L.n_s, nu = L.n_u, gout = ns * nu;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    RkBwdWhere w;
    if (!rk_bwd_where(L, row0, w)) return;
    RkBwdTile T;
    T.carve(smem);
    const nlbac_mlp& net = L.net[grp];
    const int nw = net.n_layers - 1;
    const int q = lane >> 4, r16 = lane & 15;
    const int m = 16 * half + r16, grow = row0 + m;
    const bool row_ok = grow < n;
    const int growc = min(grow, n - 1);
    const int KS0 = (ns + 3) >> 2;
    const int KSO = (grp == 0) ? KS0 : KS0 * nu;          // k-steps of the output layer's transposed product (<= 4)
    const bool keep_dz = L.dz[0] != nullptr;

    // ---- weight stream: backward fragments of layers nw-1 .. 1, then nw-1 again (next stage)
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wbase = net.rr_bwd_off * 4;
    RRGemm<S> gemm;
    if (nw >= 2) gemm.prime(rs, voff, wbase + (nw - 2) * S::LAYER_BYTES);

    // ---- constants in registers: W_out^T (A of the top product) and W_0^T (A of dX)
    float wtop[4][NB];
    {
        const float* Wl = net.params + net.w_off[nw];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int o = -1;
            if (grp == 0) { const int c = 4 * e + q; if (e < KS0 && c < ns) o = c; }
            else { const int k0 = e / nu, u = e - k0 * nu, c = 4 * k0 + q; if (e < KS0 * nu && c < ns) o = c * nu + u; }
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int uo = rr_unit_out(NB, R, jo, r16);
                wtop[e][jo] = (o >= 0 && uo >= 0) ? Wl[(long)o * HID + uo] : 0.f;
            }
        }
    }
    float w0t[KS];
    {
        const float* W0 = net.params + net.w_off[0];
        const int c = 4 * (r16 & 3) + (r16 >> 2);          // A row 4 q' + r' computes dX component 4 r' + q'
        const bool ok = (r16 & 3) < KS0 && c < ns;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w0t[ks] = ok ? W0[(long)rr_unit_in(NB, R, ks, q) * ns + c] : 0.f;
    }

    rk_bwd_tile_constants<256>(L, w, T, row0, tid);
    __syncthreads();

    for (int st = L.st_hi - 1; st >= w.st_lo; --st) {
        const bool data = w.has_data(st);
        // ---- output-layer gradients: f: dK itself, g: dK u^T (also kept for the weight gradients), and du
        float dy[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
            if (grp == 0) {
                const int c = 4 * e + q;
                if (e < KS0 && c < ns) v = T.sDK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + c];
            } else {
                const int k0 = e / nu, u = e - k0 * nu, c = 4 * k0 + q;
                if (e < KS0 * nu && c < ns) {
                    v = T.sDK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] * T.sU[m * RK_MAX_NU + u];
                    if (w.gdG && row_ok) w.gdG[((long)st * n + grow) * gout + c * nu + u] = v;
                }
            }
            dy[e] = v;
        }
        rk_bwd_du<256>(L, w, T, row0, st, tid);
        if (!data) continue;              // uniform: nothing below is needed for this stage

        const long srow = (long)st * n + growc;
        float Z[KS];
        f32x4 acc[NB];
        // ---- top product: dz_top = mask_top * (W_out^T dy)
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) {
            acc[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < KSO) acc[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtop[e][jo], dy[e], acc[jo], 0, 0, 0);
        }
        for (int l = nw - 1; l >= 0; --l) {
            if (l < nw - 1) {
                // dz_l = mask_l * (W_{l+1}^T dz_{l+1})
                const int cur = wbase + l * S::LAYER_BYTES;              // fragments of layer l + 1 sit at index l
                const int nxt = (l >= 1) ? cur - S::LAYER_BYTES : wbase + (nw - 2) * S::LAYER_BYTES;
                gemm.run(acc, Z, rs, voff, cur, nxt);
            }
            float* zrow = (keep_dz && !BITS) ? L.dz[grp] + w.soff + (long)l * L.acts_ls[grp] + ((long)st * n + grow) * HID : nullptr;
            if (BITS) {
                const unsigned word = reinterpret_cast<const unsigned*>(L.acts[grp] + w.soff + (long)l * L.acts_ls[grp])[srow * 4 + q];
#pragma unroll
                for (int jo = 0; jo < NB; ++jo)
#pragma unroll
                    for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r)
                        Z[4 * jo + r] = (row_ok && ((word >> (4 * jo + r)) & 1u)) ? acc[jo][r] : 0.f;
            } else {
                const float* arow = L.acts[grp] + w.soff + (long)l * L.acts_ls[grp] + srow * HID;
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) {
                    const f32x4 a = rr_row_load<S>(arow, jo, q);
                    f32x4 zv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) {
                        const float z = (row_ok && a[r] > 0.f) ? acc[jo][r] : 0.f;
                        Z[4 * jo + r] = z;
                        zv[r] = z;
                    }
                    if (zrow && row_ok) rr_row_store<S>(zrow, jo, q, zv);
                }
            }
        }
        if (st == 0 && !L.dx_stage0) continue;       // only the dz of stage 0 were wanted

        // ---- dX = W_0^T dz_0 (one block), then the stage algebra
        {
            f32x4 o{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) o = __builtin_amdgcn_mfma_f32_16x16x4f32(w0t[ks], Z[ks], o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = 4 * r + q;
                if (r < KS0 && c < ns) T.sDX[(grp * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] = o[r];
            }
        }
        __syncthreads();
        rk_bwd_stage_algebra<256>(L, w, T, row0, st, tid);
        __syncthreads();
    }
    __syncthreads();
    rk_bwd_outputs<256>(L, w, T, row0, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static bool rr_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("NLBAC_NODE_RR");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on == 1;
}

static int rr_shape_index(int hid) { return hid == 64 ? 0 : (hid == 100 ? 1 : (hid == 128 ? 2 : -1)); }

bool nlbac_node_rr_eligible(const nlbac_mlp* f, const nlbac_mlp* g) {
    if (!rr_enabled() || !f || !g) return false;
    if (f->hid != g->hid || rr_shape_index(f->hid) < 0) return false;
    if (f->n_layers < 3 || g->n_layers < 3) return false;
    if (f->rr_fwd_off < 0 || g->rr_fwd_off < 0 || f->rr_bwd_off < 0 || g->rr_bwd_off < 0) return false;
    const int ns = f->in_dim, nu = g->out_dim / (ns > 0 ? ns : 1);
    if (ns < 1 || ns > RK_MAX_NS || nu < 1 || nu > RK_MAX_NU) return false;
    if (((ns + 3) >> 2) * nu > 4) return false;            // g_net's outputs must fit one 16-row output block's registers
    return true;
}

extern "C" int nlbac_node_rk_mask_words(const nlbac_mlp* f, const nlbac_mlp* g, int which) {
    // uint32 words per row and layer of the bit-packed ReLU masks nlbac_node_rk_fwd writes for net `which` (0: f, 1: g)
    if (nlbac_node_rr_eligible(f, g)) return 4;            // one word per lane quarter
    const nlbac_mlp* net = which ? g : f;
    return (net->hid + 31) >> 5;
}

int nlbac_node_rr_fwd_launch(NodeRkLaunch& L, hipStream_t s) {
    if (!nlbac_node_rr_eligible(&L.net[0], &L.net[1])) return 1;
    using KernelF = void (*)(const NodeRkLaunch);
    static const KernelF kf[3][2] = {{node_rr_fwd_kernel<4, 4, 0>, node_rr_fwd_kernel<4, 4, 1>},
                                     {node_rr_fwd_kernel<7, 1, 0>, node_rr_fwd_kernel<7, 1, 1>},
                                     {node_rr_fwd_kernel<8, 4, 0>, node_rr_fwd_kernel<8, 4, 1>}};
    const size_t lds = (size_t)(RkFwdTile::floats() + NLBAC_MLP_TILE * 8) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.n, NLBAC_MLP_TILE));
    hipLaunchKernelGGL(kf[rr_shape_index(L.net[0].hid)][L.acts_bits ? 1 : 0], grid, dim3(256), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_rk_fwd(rr)");
    return 0;
}

int nlbac_node_rr_bwd_launch(NodeRkBwdLaunch& L, hipStream_t s) {
    if (!nlbac_node_rr_eligible(&L.net[0], &L.net[1])) return 1;
    using KernelB = void (*)(const NodeRkBwdLaunch);
    static const KernelB kb[3][2] = {{node_rr_bwd_kernel<4, 4, 0>, node_rr_bwd_kernel<4, 4, 1>},
                                     {node_rr_bwd_kernel<7, 1, 0>, node_rr_bwd_kernel<7, 1, 1>},
                                     {node_rr_bwd_kernel<8, 4, 0>, node_rr_bwd_kernel<8, 4, 1>}};
    const size_t lds = (size_t)RkBwdTile::floats() * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.n, NLBAC_MLP_TILE));
    hipLaunchKernelGGL(kb[rr_shape_index(L.net[0].hid)][L.acts_bits ? 1 : 0], grid, dim3(256), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_rk_bwd(rr)");
    return 0;
}
