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
#include <type_traits>

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

#ifdef RR_TIMING      // ablation build only: waves 0 (f_net) and 2 (g_net) of workgroup 0 stamp the shader clock into L.err (as int64)
#define RSTAMP(slot_) if (L.err && blockIdx.x == 0 && lane == 0 && half == 0) reinterpret_cast<long long*>(L.err)[grp * 256 + (slot_)] = (long long)__builtin_readcyclecounter();
#else
#define RSTAMP(slot_)
#endif

#define RR_MAX_W 4        /* layer 0 + up to three hid x hid layers (n_layers <= 5: the reference's f_net) */

#ifdef RR_TIMING      // ablation build only: the backward's stamps go to a device symbol (nlbac_debug_bwd_stamps reads it back)
__device__ long long g_bwd_stamps[2 * 256];
#define BWSTAMP(slot_) if (blockIdx.x == 0 && lane == 0 && half == 0) g_bwd_stamps[grp * 256 + (slot_)] = (long long)__builtin_readcyclecounter();
extern "C" int nlbac_debug_bwd_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_stamps), sizeof(long long) * 2 * 256) == hipSuccess ? 0 : -1;
}
#else
#define BWSTAMP(slot_)
#endif

// SPLIT: f_net has one hid x hid layer more than g_net (U/sac_cbf_clf/model.py:186-206), so its waves ran ~175 MFMAs per
// stage longer and g_net's waited a quarter of every stage at the stage barrier.  With SPLIT the g_net wave of each half
// tile takes over the upper groups of output blocks of f_net's LAST hid x hid layer: the f_net wave hands its layer-2
// activations over through LDS (behind a flag only the two waves touch), both compute their blocks and their part of
// f_net's output layer, and the two partial outputs meet in the stage's k = f + g u step.
template <int NB, int R, int BITS, int SPLIT>
__device__ __forceinline__ void node_rr_fwd_body(const NodeRkLaunch& L) {
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID;
    constexpr int TB = NB - 2;                 // first block of a layer's last group, the "tail": its accumulators are
    constexpr int NT = KS - 4 * TB;            // finished inside the NEXT product (NT values: 5 at hid 100, else 8)
    constexpr int MF = rr_split_m<S>();        // (SPLIT) f_net's last layer: MFMAs [0, MF) stay with its wave, [MF, NM) go
    using PF = RRPart<S, 0, MF>;
    using PG = RRPart<S, MF, S::NM>;
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
    float* const sYin = smem + RkFwdTile::floats();      // [32][8] the stage input, columns ns..7 zero
    float* const sW0 = sYin + NLBAC_MLP_TILE * 8;        // [net][k-step < 3][block < 8][lane]: layer 0's A fragments
    // (SPLIT) the hand-over between a half tile's f_net and g_net waves
    float* const sX = sW0 + 2 * 3 * 8 * 64;              // [half][KS][lane] f_net's layer-2 activations
    float* const sF2 = sX + 2 * KS * 64;                 // [32][8] the g_net wave's part of f(x)
    unsigned* const sMw = reinterpret_cast<unsigned*>(sF2 + NLBAC_MLP_TILE * RK_MAX_NS);      // [half][f, g][lane] mask-word parts
    int* const sFlag = reinterpret_cast<int*>(sMw + 2 * 2 * 64);                               // [half] stage + 1 once sX is there
    const nlbac_mlp& net = L.net[grp];
    const int nw = net.n_layers - 1;                     // layer 0 + (nw - 1) hid x hid layers, then the output layer
    const int n_rows = min(NLBAC_MLP_TILE, n - row0);
    const int q = lane >> 4, r16 = lane & 15;
    const int m = 16 * half + r16, grow = row0 + m;      // this lane's row: within the tile, global
    const bool row_ok = grow < n;
    const int KS0 = (ns + 3) >> 2;                       // registers a lane needs for its row's state components (1 or 2)
    const int KL0 = (ns + 4) >> 2;                       // k-steps of layer 0, which contracts [y | 1] with [W_0 | b_0] (1..3)

    // ---- what the stage loop reads of the launch descriptor, once
    const float* const params = net.params;
    int boff[RR_MAX_W];
#pragma unroll
    for (int l = 0; l < RR_MAX_W; ++l) boff[l] = net.b_off[l];
    float* const acts = L.acts[grp] ? L.acts[grp] + w.soff : nullptr;
    const long acts_ls = L.acts_ls[grp];
    const int stage_end = L.stage_end, S_last = L.S_total - 1;
    // the initial-step probe with its norm fused into this launch (norm_mode 1): nothing it would leave in memory — the
    // probe point, its derivative, g there — is read by anyone (the norm comes from LDS, the step's stage 1 overwrites
    // them), and stores in front of the epilogue's atomics are waited for there (vmcnt is in order)
    const bool quiet = L.norm_mode == 1;

    // ---- the wave's weight stream: hid x hid layers 1 .. nw-1, then layer 1 again (next stage)
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wbase = net.rr_fwd_off * 4;
    RRGemm<S> gemm;
    gemm.prime(rs, voff, wbase);

    // ---- constants of the launch: layer 0's A fragments (its bias rides in the k slot behind the last state component)
    //      to LDS in fragment order, the output layer's into registers
    if (half == 0) {
        const float* W0 = params + net.w_off[0];
        const float* b0 = params + boff[0];
#pragma unroll
        for (int k0 = 0; k0 < 3; ++k0)
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                // (unconditional loads, clamped index, select afterwards: a guarded load is a branch and a round trip of its own)
                const int uo = rr_unit_out(NB, R, jo, r16), col = 4 * k0 + q, uc = max(uo, 0);
                const float vw = W0[uc * ns + min(col, ns - 1)], vb = b0[uc];
                sW0[((grp * 3 + k0) * 8 + jo) * 64 + lane] = (uo < 0 || col > ns) ? 0.f : (col < ns ? vw : vb);
            }
    }
    float wo[KS];
    {
        const int orow = rr_out_row(grp, r16, ns, nu);
        const float* wrow = params + net.w_off[nw] + (long)max(orow, 0) * HID;
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) {
            const f32x4 v = rr_row_load<S>(wrow, jo, q);
#pragma unroll
            for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) wo[4 * jo + r] = (orow >= 0) ? v[r] : 0.f;
        }
    }
    // this lane's outputs of the output layer (register r): their LDS slot in sF / sG (-1: none), their bias
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

    // (SPLIT) what the g_net wave needs of f_net: its pack, its last layer's bias, its output layer's A fragments for the
    // k-steps the wave's blocks yield, the output slots (f_net's mapping of o_idx above)
    const nlbac_mlp& netF = L.net[0];
    const __amdgpu_buffer_rsrc_t rsF = rr_rsrc(netF.packed, netF.packed_floats);
    const int curF3 = netF.rr_fwd_off * 4 + 2 * S::LAYER_BYTES;          // byte offset of f_net's layer-3 stream
    float wof[KS];
    int of_idx[4];
    if constexpr (SPLIT != 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wof[ks] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int c = 4 * r + q; of_idx[r] = (r < KS0 && c < ns) ? c : -1; }
        if (grp == 1) {
            const int orow = rr_out_row(0, r16, ns, nu);
            const float* wrow = netF.params + netF.w_off[4] + (long)max(orow, 0) * HID;
#pragma unroll
            for (int jo = PG::J0; jo < NB; ++jo) {
                const f32x4 v = rr_row_load<S>(wrow, jo, q);
#pragma unroll
                for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) wof[4 * jo + r] = (orow >= 0) ? v[r] : 0.f;
            }
        }
        if (tid < 2) sFlag[tid] = 0;
    }

    RSTAMP(0)
    rk_fwd_tile_constants<256>(L, w, T, row0, tid);
    RSTAMP(1)

    for (int st = L.stage_begin; st < stage_end; ++st) {
        const int sb = 2 + 8 * (st - L.stage_begin);
        (void)sb;
        if (st == L.stage_begin) {
            rk_fwd_first_input(L, w, T, row0, st, 8, sYin, 8, !quiet, tid, 256);
            __syncthreads();
        }
        RSTAMP(sb + 0)
        // the next stage's tableau row (scalar loads from the kernel arguments, issued now: the combine step behind the
        // layer chains was a chain of exposed load latencies)
        float bn[RK_MAX_STAGES];
        {
            const int sn = min(st + 1, S_last);
#pragma unroll
            for (int j = 0; j < RK_MAX_STAGES; ++j) bn[j] = L.beta[sn][j];
        }
        const long srow = (long)st * n + grow;
        float Ha[KS], Hb[KS];             // activations ping-pong between two register sets
        f32x4 acc0[NB], acc[NB], bv[NB], bpre[3];
        unsigned wd = 0u;                 // the mask word being assembled (values arrive in ascending register order)
        constexpr int G0 = rr_group_first(NB);
        // biases enter as the C operand of each block's first MFMA; those of a layer's first group of blocks are
        // requested one product ahead (bpre), the others at the layer's start
        auto prefetch_bias = [&](int l) __attribute__((always_inline)) {
#pragma unroll
            for (int jo = 0; jo < G0; ++jo) bpre[jo] = rr_bias<S>(params + boff[l], jo, q);
        };
        prefetch_bias(1);
        // (SPLIT) both waves' queues for their ranges of f_net's last layer: requested now, consumed two layers later
        PF partF; PG partG;
        if constexpr (SPLIT != 0) {
            if (grp == 0) partF.prime(rs, voff, curF3);
            else partG.prime(rsF, voff, curF3);
        }

        // what the backward needs of a finished value goes out once: a mask bit (word per layer), or the activation itself
        // (activation mode) a finished layer's activations leave in ONE burst, issued where the product that consumes
        // them starts its last group of blocks — by then the layer's pending tail is finished too.  Stores share the
        // loads' in-order vmcnt queue: issued one by one between the fragment loads, each made the MFMAs behind it wait
        // for its own trip to HBM; as a burst the trips overlap and the stream stalls once per layer.
        auto save_layer = [&](int l, const float (&H)[KS]) __attribute__((always_inline)) {
            if (BITS || !acts || !row_ok) return;
            float* rowp = acts + (long)l * acts_ls + srow * HID;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                f32x4 hv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) hv[rr] = H[4 * jo + rr];
                rr_row_store<S>(rowp, jo, q, hv);
            }
        };
        auto save_word = [&](int l, unsigned word) __attribute__((always_inline)) {
            if (BITS && acts && row_ok) reinterpret_cast<unsigned*>(acts + (long)l * acts_ls)[srow * 4 + q] = word;
        };
        // layer 0's value ks (no bias: folded into the product), finished just before layer 1's k-step ks reads it
        auto pre_l0 = [&](int ks) __attribute__((always_inline)) {
            const int jo = (ks < 4 * (NB - 1)) ? (ks >> 2) : NB - 1, r = ks - 4 * jo;
            const float h = rr_relu(acc0[jo][r]);
            Ha[ks] = h;
            if (BITS) rr_mask_push(wd, h);
            if (ks == KS - 1) save_word(0, wd);
        };
        // the tail of hid x hid layer lp (blocks TB, TB+1 of `acc`), finished inside the product that follows it: value t
        // at that product's k-step t (it is read at k-step 4 TB + t)
        auto pre_tail = [&](int lp, float (&H)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            const float h = rr_relu(acc[jo][r]);
            H[4 * TB + t] = h;
            if (BITS) rr_mask_push(wd, h);
            if (t == NT - 1) save_word(lp, wd);
        };

        // ---- layer 0: K = ns + 1 (one to three k-steps), straight from the stage input
        {
            float yv[3], a0[3][NB];
#pragma unroll
            for (int k0 = 0; k0 < 3; ++k0) {
                const int col = 4 * k0 + q;
                yv[k0] = (col < ns) ? sYin[m * 8 + min(col, 7)] : (col == ns ? 1.f : 0.f);
            }
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) a0[0][jo] = sW0[((grp * 3 + 0) * 8 + jo) * 64 + lane];
            if (KL0 == 1) {
#pragma unroll
                for (int jo = 0; jo < NB; ++jo)
                    acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][jo], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            } else {
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) a0[1][jo] = sW0[((grp * 3 + 1) * 8 + jo) * 64 + lane];
                if (KL0 == 2) {
#pragma unroll
                    for (int jo = 0; jo < NB; ++jo) {
                        acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][jo], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1][jo], yv[1], acc0[jo], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int jo = 0; jo < NB; ++jo) a0[2][jo] = sW0[((grp * 3 + 2) * 8 + jo) * 64 + lane];
#pragma unroll
                    for (int jo = 0; jo < NB; ++jo) {
                        acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][jo], yv[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1][jo], yv[1], acc0[jo], 0, 0, 0);
                        acc0[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2][jo], yv[2], acc0[jo], 0, 0, 0);
                    }
                }
            }
        }
        RSTAMP(sb + 1)

        // ---- the hid x hid layers and the output layer, statically unrolled (lc: the layer index as a type): wide layer l
        //      reads one activation set and writes the other
        auto wide = [&](auto lc, float (&Hin)[KS], float (&Hout)[KS]) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) bv[jo] = (jo < G0) ? bpre[jo] : rr_bias<S>(params + boff[l], jo, q);
            __builtin_amdgcn_sched_barrier(0);
            const int cur = wbase + (l - 1) * S::LAYER_BYTES;
            // (SPLIT: f_net's layer 3 has queues of its own — behind layer 2 the main stream goes on with the next stage)
            const int nxt = (l + 1 < nw && !(SPLIT != 0 && grp == 0 && l == 2)) ? cur + S::LAYER_BYTES : wbase;
            gemm.run(acc, bv, Hin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) {
                         if (l == 1) pre_l0(ks);
                         else pre_tail(l - 1, Hin, ks);
                     },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         const float h = rr_relu(acc[jo][r]);
                         Hout[4 * jo + r] = h;
                         if (BITS) rr_mask_push(wd, h);
                     },
                     [&]() __attribute__((always_inline)) {
                         if (l + 1 < nw) prefetch_bias(l + 1);
                         save_layer(l - 1, Hin);
                     });
            RSTAMP(sb + 1 + l)
        };
        // output layer (<= 16 outputs: one block), to LDS for k = f + g u; g(x) also to global for the backward
        auto outl = [&](auto lc, float (&Hin)[KS]) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;            // (= nw: the layers before it are 0 .. l-1)
            const f32x4 o = RRGemm<S>::block(wo, Hin, [&](int ks) __attribute__((always_inline)) { pre_tail(l - 1, Hin, ks); });
            save_layer(l - 1, Hin);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (o_idx[r] < 0) continue;
                const float val = o[r] + o_bias[r];
                if (grp == 0) T.sF[m * RK_MAX_NS + o_idx[r]] = val;
                else {
                    T.sG[m * RK_MAX_GOUT + o_idx[r]] = val;
                    if (row_ok && !quiet) w.gG[((long)st * n + grow) * gout + o_idx[r]] = val;
                }
            }
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
        // the reference's nets (U/sac_cbf_clf/model.py:186-206): f_net has three hid x hid layers, g_net two.  (One code
        // path per depth: a third, for two-layer nets, cost 26 more VGPRs and accumulator-file spills in all of them.)
        wide(I1{}, Ha, Hb);
        wide(I2{}, Hb, Ha);
        if constexpr (SPLIT == 0) {
            if (grp == 0) { wide(I3{}, Ha, Hb); outl(I4{}, Hb); }
            else outl(I3{}, Ha);
        } else {
            // one part of f_net's output layer over the k-steps [K0, K1) of the wave's blocks -> LDS (the f_net wave's with
            // the bias, into sF; the g_net wave's into sF2)
            auto out_part = [&](const f32x4& o, float* dst, bool with_bias) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (of_idx[r] >= 0) dst[m * RK_MAX_NS + of_idx[r]] = with_bias ? o[r] + o_bias[r] : o[r];
            };
            auto save_rows = [&](int l, const float (&H)[KS], int j0, int j1) __attribute__((always_inline)) {
                float* const actsF = L.acts[0] ? L.acts[0] + w.soff : nullptr;
                if (BITS || !actsF || !row_ok) return;
                float* rowp = actsF + (long)l * L.acts_ls[0] + srow * HID;
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) {
                    if (jo < j0 || jo >= j1) continue;
                    f32x4 hv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) hv[rr] = H[4 * jo + rr];
                    rr_row_store<S>(rowp, jo, q, hv);
                }
            };
            unsigned wp = 0u;                  // this wave's part of layer 3's mask word
            if (grp == 0) {
                // layer 2 is finished now (its tail is not deferred: the g_net wave waits for ALL of it), handed over, then
                // the blocks [0, PF::J1) of layer 3 and their part of the output layer
#pragma unroll
                for (int t = 0; t < NT; ++t) pre_tail(2, Ha, t);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) sX[(half * KS + ks) * 64 + lane] = Ha[ks];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __hip_atomic_store(sFlag + half, st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) bv[jo] = rr_bias<S>(params + boff[3], jo, q);
                auto fin3 = [&](int jo, int r) __attribute__((always_inline)) {
                    const float h = rr_relu(acc[jo][r]);
                    Hb[4 * jo + r] = h;
                    if (BITS) rr_mask_push(wp, h);
                };
                partF.run(acc, bv, Ha, rs, voff, curF3, [&](int) __attribute__((always_inline)) {}, fin3);
                RSTAMP(sb + 4)
                save_layer(2, Ha);
                const f32x4 o = PF::block(wo, Hb, [&](int ks) __attribute__((always_inline)) {
                    if (ks >= 4 * PF::JT) fin3(PF::JT + ((ks - 4 * PF::JT) >> 2), (ks - 4 * PF::JT) & 3);
                });
                save_rows(3, Hb, 0, PF::J1);
                out_part(o, T.sF, true);
                if (BITS) sMw[(half * 2 + 0) * 64 + lane] = wp << (KS - PF::K1);
            } else {
                outl(I3{}, Ha);
                // f_net's layer-2 activations of the same rows, once its wave has put them there
                while (__hip_atomic_load(sFlag + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != st + 1) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) Hb[ks] = sX[(half * KS + ks) * 64 + lane];
                RSTAMP(sb + 4)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) bv[jo] = (jo >= PG::J0) ? rr_bias<S>(netF.params + netF.b_off[3], jo, q) : f32x4{0.f, 0.f, 0.f, 0.f};
                auto fin3 = [&](int jo, int r) __attribute__((always_inline)) {
                    const float h = rr_relu(acc[jo][r]);
                    Ha[4 * jo + r] = h;
                    if (BITS) rr_mask_push(wp, h);
                };
                partG.run(acc, bv, Hb, rsF, voff, curF3, [&](int) __attribute__((always_inline)) {}, fin3);
                const f32x4 o = PG::block(wof, Ha, [&](int ks) __attribute__((always_inline)) {
                    if (ks >= 4 * PG::JT) {
                        const int t = ks - 4 * PG::JT, jo = PG::JT + (t >> 2), r = t & 3;
                        if (jo < NB - 1 || r < R) fin3(jo, r);
                    }
                });
                save_rows(3, Ha, PG::J0, NB);
                out_part(o, sF2, false);
                if (BITS) sMw[(half * 2 + 1) * 64 + lane] = wp;
            }
        }
        RSTAMP(sb + 5)
        __syncthreads();
        RSTAMP(sb + 6)
        // ---- k = f + g u (same op order as affine_fwd_kernel) and, same thread, the next stage's input
        //      Y_{st+1} = y0 + h sum_j beta[st+1][j] K_j (same op order as rk_combine_kernel): one (row, component) per
        //      thread, every LDS operand requested up front, no data-dependent branch
        {
            const int mm = tid >> 3, c = tid & 7;
            const bool more = st + 1 < stage_end, cv = c < ns, rv = row0 + mm < n;
            float a = T.sF[mm * RK_MAX_NS + c];
            if constexpr (SPLIT != 0) a += sF2[mm * RK_MAX_NS + c];
            float gv[RK_MAX_NU], uv[RK_MAX_NU], kj[RK_MAX_STAGES - 1];
#pragma unroll
            for (int u = 0; u < RK_MAX_NU; ++u) {
                gv[u] = T.sG[mm * RK_MAX_GOUT + min(c * nu + u, RK_MAX_GOUT - 1)];
                uv[u] = T.sU[mm * RK_MAX_NU + u];
            }
#pragma unroll
            for (int j = 0; j < RK_MAX_STAGES - 1; ++j) kj[j] = T.sK[(j * NLBAC_MLP_TILE + mm) * RK_MAX_NS + c];
            float y = T.sY0[mm * RK_MAX_NS + c];
            const float h = T.sH[mm];
#pragma unroll
            for (int u = 0; u < RK_MAX_NU; ++u) {
                const float t = a + gv[u] * uv[u];
                a = (u < nu) ? t : a;
            }
            float bst = 0.f;
#pragma unroll
            for (int j = 0; j < RK_MAX_STAGES - 1; ++j) {
                const float t = y + kj[j] * (bn[j] * h);
                y = (j < st && bn[j] != 0.f) ? t : y;
                bst = (j == st) ? bn[j] : bst;
            }
            {
                const float t = y + a * (bst * h);
                y = (bst != 0.f) ? t : y;
            }
            if (cv) {
                T.sK[(st * NLBAC_MLP_TILE + mm) * RK_MAX_NS + c] = a;
                if (rv && !quiet) w.gK[((long)st * n + row0 + mm) * ns + c] = a;
                if (more && rv) w.gY[((long)(st + 1) * n + row0 + mm) * ns + c] = y;
            }
            if (more) sYin[mm * 8 + c] = cv ? y : 0.f;
        }
        if constexpr (SPLIT != 0 && BITS != 0) {      // layer 3's mask word: the two waves' parts, stored by the f_net wave
            if (grp == 0 && L.acts[0] && row_ok)
                reinterpret_cast<unsigned*>(L.acts[0] + w.soff + 3 * L.acts_ls[0])[srow * 4 + q] =
                    sMw[(half * 2 + 0) * 64 + lane] | sMw[(half * 2 + 1) * 64 + lane];
        }
        __syncthreads();
        RSTAMP(sb + 7)
    }
#ifdef RR_TIMING
    if (L.err) return;
#endif
    rk_fwd_outputs_and_control<256>(L, w, T, row0, n_rows, tid);
}

template <int NB, int R, int BITS, int SPLIT>
__global__ __launch_bounds__(256) void node_rr_fwd_kernel(const NodeRkLaunch L) {
    node_rr_fwd_body<NB, R, BITS, SPLIT>(L);
}

// The three launches that open a dopri5 solve — f0 = field(y0) with Hairer's first guess, the probe f(y0 + h0 f0) with
// the initial step size, the first attempted step (stages 1..6) — as ONE launch (nlbac_node_rk_fwd_begin): the same code
// three times, a per-problem wait in between (rk_fwd_grid_wait: the workgroup that ran the phase's controller releases
// the others).  What it was meant to save is two launches' dispatch, prologue and cold weights (f0 + probe cost 44 us in
// the update for two stage evaluations of 8.7 us) — measured, it saves nothing: 108 us against 105 (the cost of the
// one-stage launches is their norm's election, and the waits here are the same elections).  Kept behind
// NLBAC_NODE_PERSIST=1, tested against the three launches (bit-identical).  Needs every workgroup of the launch resident
// at once: <= 8192 rows.
template <int NB, int R, int SPLIT>
__global__ __launch_bounds__(256) void node_rr_fwd_begin_kernel(const NodeRkLaunch LA, const NodeRkLaunch LB, const NodeRkLaunch LC) {
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    node_rr_fwd_body<NB, R, 1, SPLIT>(LA);
    if (!rk_fwd_grid_wait(LA, row0)) return;
    node_rr_fwd_body<NB, R, 1, SPLIT>(LB);
    if (!rk_fwd_grid_wait(LB, row0)) return;
    node_rr_fwd_body<NB, R, 1, SPLIT>(LC);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the same step, same wave roles.  Per stage (descending): the output layer's gradient enters as the B
// operand of one transposed block product, then dz_{l-1} = mask_{l-1} * (W_l^T dz_l) down the chain in registers (the
// backward RR pack), then dX = W_0^T dz_0; the two nets meet in the stage algebra (rk_bwd_stage_algebra).
// ---------------------------------------------------------------------------------------------------------------------
// SPLIT (see the forward): f_net's chain has one product more than g_net's — its FIRST, dz_2 = mask_2 * (W_3^T dz_3).  The
// g_net wave of the half tile computes the lower groups of output blocks of that product before its own chain starts
// (dz_3 comes over through LDS behind a flag, the blocks go back the same way), the f_net wave the upper groups.
template <int NB, int R, int BITS, int SPLIT>
__global__ __launch_bounds__(256) void node_rr_bwd_kernel(const NodeRkBwdLaunch L) {
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID, TB = NB - 2, NT = KS - 4 * TB;
    constexpr int MB = rr_split_m<S>();        // (SPLIT) f_net's first product: MFMAs [0, MB) go to the g_net wave
    using PG = RRPart<S, 0, MB>;
    using PF = RRPart<S, MB, S::NM>;
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
    float* const sWt = smem + RkBwdTile::floats();       // [net][k-step < 4][block < 8][lane]: W_out^T's A fragments
    // (SPLIT) the hand-over between a half tile's f_net and g_net waves
    float* const sX = sWt + 2 * 4 * 8 * 64;              // [half][KS][lane] f_net's dz_3
    float* const sX2 = sX + 2 * KS * 64;                 // [half][PG::K1][lane] the g_net wave's blocks of dz_2
    int* const sFlag = reinterpret_cast<int*>(sX2 + 2 * 16 * 64);      // [2][half]: sX / sX2 are there for stage key
    const nlbac_mlp& net = L.net[grp];
    const int nw = net.n_layers - 1;
    const int q = lane >> 4, r16 = lane & 15;
    const int m = 16 * half + r16, grow = row0 + m;
    const bool row_ok = grow < n;
    const int growc = min(grow, n - 1);
    const int KS0 = (ns + 3) >> 2;
    const int KSO = (grp == 0) ? KS0 : KS0 * nu;          // k-steps of the output layer's transposed product (<= 4)
    const bool keep_dz = L.dz[0] != nullptr;

    // ---- what the stage loop reads of the launch descriptor, once
    const float* const params = net.params;
    const float* const acts = L.acts[grp] + w.soff;
    float* const dz = keep_dz ? L.dz[grp] + w.soff : nullptr;
    const long acts_ls = L.acts_ls[grp];
    const int dx_stage0 = L.dx_stage0;

    // ---- weight stream: backward fragments of layers nw-1 .. 1, then nw-1 again (next stage)
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16;
    const int wbase = net.rr_bwd_off * 4;
    RRGemm<S> gemm;
    // (SPLIT: f_net's first product has queues of its own — its main stream starts with the second)
    const int first_l = (SPLIT != 0 && grp == 0) ? nw - 3 : nw - 2;
    gemm.prime(rs, voff, wbase + first_l * S::LAYER_BYTES);
    const nlbac_mlp& netF = L.net[0];
    const __amdgpu_buffer_rsrc_t rsF = rr_rsrc(netF.packed, netF.packed_floats);
    const int curB3 = netF.rr_bwd_off * 4 + 2 * S::LAYER_BYTES;          // byte offset of W_3^T's stream (f_net)
    if (SPLIT != 0 && tid < 4) sFlag[tid] = 0;

    // ---- constants: W_out^T (A of the top product) to LDS in fragment order, W_0^T (A of dX) into registers
    if (half == 0) {
        const float* Wl = params + net.w_off[nw];
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
    float w0t[KS];
    {
        const float* W0 = params + net.w_off[0];
        const int c = 4 * (r16 & 3) + (r16 >> 2);          // A row 4 q' + r' computes dX component 4 r' + q'
        const bool ok = (r16 & 3) < KS0 && c < ns;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float vw = W0[(long)rr_unit_in(NB, R, ks, q) * ns + min(c, ns - 1)];
            w0t[ks] = ok ? vw : 0.f;
        }
    }

    BWSTAMP(0)
    rk_bwd_tile_constants<256>(L, w, T, row0, tid);
    __syncthreads();
    BWSTAMP(1)

    // g(Y_st) for the du term comes from global memory: this thread's values of a stage are requested while the stage
    // before it runs (loads issued at the start of a stage would sit in front of the stage's first weight-fragment loads in
    // the in-order vmcnt queue); the tableau row (scalar loads) likewise, behind the stage's first LDS wait
    const bool du_thread = L.du && tid < NLBAC_MLP_TILE * nu;
    const int du_m = du_thread ? tid / nu : 0, du_c = du_thread ? tid - du_m * nu : 0;
    float gnext[RK_MAX_NS];
    __shared__ float sBeta[RK_MAX_STAGES * RK_MAX_STAGES];      // the tableau, once: per-stage scalar loads of its rows from the
    if (tid < RK_MAX_STAGES * RK_MAX_STAGES)                      // kernel arguments cost SGPRs (spills) and a wait per stage
        sBeta[tid] = (&L.beta[0][0])[tid];
    auto request_g = [&](int stn) __attribute__((always_inline)) {
        if (!du_thread || stn < w.st_lo) return;
        const float* gp = w.gG + ((long)stn * n + min(row0 + du_m, n - 1)) * gout + du_c;
#pragma unroll
        for (int r = 0; r < RK_MAX_NS; ++r) gnext[r] = gp[min(r, ns - 1) * nu];
    };
#pragma unroll
    for (int r = 0; r < RK_MAX_NS; ++r) gnext[r] = 0.f;
    request_g(L.st_hi - 1);
    for (int st = L.st_hi - 1; st >= w.st_lo; --st) {
        const int sbw = 2 + 8 * st;
        (void)sbw;
        BWSTAMP(sbw + 0)
        const bool data = w.has_data(st);
        float gcur[RK_MAX_NS];
#pragma unroll
        for (int r = 0; r < RK_MAX_NS; ++r) gcur[r] = gnext[r];
        request_g(st - 1);
        // ---- output-layer gradients: f: dK itself, g: dK u^T (also kept for the weight gradients), and du — every LDS
        //      operand requested with a clamped index, selects afterwards (no per-column branch)
        float dy[4];
        {
            float dk[4], uu[4], dkd[RK_MAX_NS];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k0 = (grp == 0) ? e : e / nu, u = (grp == 0) ? 0 : e - k0 * nu, c = 4 * k0 + q;
                dk[e] = T.sDK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + min(c, ns - 1)];
                uu[e] = (grp == 0) ? 1.f : T.sU[m * RK_MAX_NU + min(u, nu - 1)];
            }
#pragma unroll
            for (int r = 0; r < RK_MAX_NS; ++r) dkd[r] = T.sDK[(st * NLBAC_MLP_TILE + du_m) * RK_MAX_NS + r];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k0 = (grp == 0) ? e : e / nu, u = (grp == 0) ? 0 : e - k0 * nu, c = 4 * k0 + q;
                const bool ok = (grp == 0) ? (e < KS0 && c < ns) : (e < KS0 * nu && c < ns);
                const float v = (grp == 0) ? dk[e] : dk[e] * uu[e];
                dy[e] = ok ? v : 0.f;
                if (grp != 0 && ok && w.gdG && row_ok) w.gdG[((long)st * n + grow) * gout + c * nu + u] = v;
            }
            if (du_thread) {        // du += g(Y_st)^T dK_st (rk_bwd_du's sum, same order)
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < RK_MAX_NS; ++r) a = (r < ns) ? a + gcur[r] * dkd[r] : a;
                T.sDU[du_m * RK_MAX_NU + du_c] = T.sDU[du_m * RK_MAX_NU + du_c] + 1.0f * a;
            }
        }
        if (!data) continue;              // uniform: nothing below is needed for this stage
        BWSTAMP(sbw + 1)

        const long srow = (long)st * n + growc;
        float Za[KS], Zb[KS];              // dz ping-pong between two register sets
        f32x4 acct[NB], acc[NB], zero[NB];
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) zero[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ReLU masks of a layer's outputs for this lane's units: one word (mask mode) or the activations themselves,
        // requested before the product they gate so that they land under it; `*t`: those of the pending tail / top product
        unsigned mw = 0u, mwt = 0u;
        f32x4 avA[NB], avB[NB], avt[2];       // (activation mode) two sets: a product's masks are requested one product ahead
        auto fetch_masks = [&](int l, f32x4 (&av)[NB]) __attribute__((always_inline)) {
            if (BITS) {        // (rows past the end contribute nothing: their word is cleared once)
                mw = reinterpret_cast<const unsigned*>(acts + (long)l * acts_ls)[srow * 4 + q];
                mw = row_ok ? mw : 0u;
            }
            else {
                const float* arow = acts + (long)l * acts_ls + srow * HID;
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) av[jo] = rr_row_load<S>(arow, jo, q);
            }
        };
        // (activation mode, weight gradients wanted) a finished dz leaves in one burst inside the product that consumes it,
        // like the forward's activations (see there)
        auto save_dz = [&](int l, const float (&Z)[KS]) __attribute__((always_inline)) {
            if (BITS || !dz || !row_ok) return;
            float* rowp = dz + (long)l * acts_ls + ((long)st * n + grow) * HID;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                f32x4 zv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) zv[rr] = Z[4 * jo + rr];
                rr_row_store<S>(rowp, jo, q, zv);
            }
        };
        // the top product's value ks (mask mode: finished just before the next product's k-step ks reads it)
        auto pre_top = [&](int ks) __attribute__((always_inline)) {
            const int jo = (ks < 4 * (NB - 1)) ? (ks >> 2) : NB - 1, r = ks - 4 * jo;
            if (BITS) Za[ks] = rr_mask_gate<KS>(mwt, ks, acct[jo][r]);
            else Za[ks] = (row_ok && avA[jo][r] > 0.f) ? acct[jo][r] : 0.f;
        };
        // the tail (blocks TB, TB+1 of `acc`) of the product that produced dz of layer lp, finished inside the next one
        auto pre_tail = [&](float (&Z)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            if (BITS) Z[4 * TB + t] = rr_mask_gate<KS>(mwt, 4 * TB + t, acc[jo][r]);
            else Z[4 * TB + t] = (row_ok && avt[jo - TB][r] > 0.f) ? acc[jo][r] : 0.f;
        };

        // ---- top product: dz_top = mask_top * (W_out^T dy); the first chain product's masks are requested with its own
        fetch_masks(nw - 1, avA);
        if (!BITS) fetch_masks(nw - 2, avB);
        {
            float at[4][NB];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) at[e][jo] = (e < 2 || KSO > 2) ? sWt[((grp * 4 + e) * 8 + jo) * 64 + lane] : 0.f;
            if (KSO <= 2) {
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) {
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[0][jo], dy[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[1][jo], dy[1], acct[jo], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) {
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[0][jo], dy[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[1][jo], dy[1], acct[jo], 0, 0, 0);
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[2][jo], dy[2], acct[jo], 0, 0, 0);
                    acct[jo] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[3][jo], dy[3], acct[jo], 0, 0, 0);
                }
            }
        }
#ifdef RR_BWD_NO_DEFER_TOP
        constexpr bool defer_top = false;
        mwt = mw;
#else
        constexpr bool defer_top = BITS != 0;
#endif
        if (!defer_top) {       // (activation mode keeps one set of mask registers: the top product is finished at once)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pre_top(ks);
        }
        // ---- dz_{nw-1-p} = mask * (W_{nw-p}^T dz_{nw-p}), p = 1 .. nw-1, then dX = W_0^T dz_0 (one block); statically
        //      unrolled (pc: the product index as a type): a product reads one dz set and writes the other
        // (avC: this product's masks — already requested; avN: where the next product's go)
        auto prod = [&](auto pc, float (&Zin)[KS], float (&Zout)[KS], f32x4 (&avC)[NB], f32x4 (&avN)[NB]) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            const int lo = nw - 1 - p;                            // the layer whose dz this product yields
            mwt = mw;
            if (BITS) fetch_masks(lo, avC);
            __builtin_amdgcn_sched_barrier(0);
            const int cur = wbase + lo * S::LAYER_BYTES;              // fragments of layer lo + 1 sit at index lo
            const int nxt = (lo >= 1) ? cur - S::LAYER_BYTES : wbase + first_l * S::LAYER_BYTES;
            gemm.run(acc, zero, Zin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) {
                         if (p == 1) { if (defer_top) pre_top(ks); }
                         else pre_tail(Zin, ks);
                     },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         if (BITS) Zout[4 * jo + r] = rr_mask_gate<KS>(mw, 4 * jo + r, acc[jo][r]);
                         else Zout[4 * jo + r] = (row_ok && avC[jo][r] > 0.f) ? acc[jo][r] : 0.f;
                     },
                     [&]() __attribute__((always_inline)) {
                         save_dz(lo + 1, Zin);                        // (Zin is complete: its tail was finished in group 0)
                         if (!BITS && lo >= 1) fetch_masks(lo - 1, avN);
                     });
            avt[0] = avC[TB]; avt[1] = avC[TB + 1];               // (this product's own tail is finished in the next one)
        };
        const bool skip_dx = (st == 0 && !dx_stage0);       // only the dz of stage 0 were wanted
        auto dxl = [&](float (&Zin)[KS]) __attribute__((always_inline)) {
            mwt = mw;
            const f32x4 o = RRGemm<S>::block(w0t, Zin, [&](int ks) __attribute__((always_inline)) { pre_tail(Zin, ks); });
            save_dz(0, Zin);
            if (!skip_dx) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int c = 4 * r + q;
                    if (r < KS0 && c < ns) T.sDX[(grp * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] = o[r];
                }
            }
        };
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        BWSTAMP(sbw + 2)
        if constexpr (SPLIT == 0) {
            prod(I1{}, Za, Zb, avB, avA);                     // (f_net: three hid x hid layers, g_net: two — see the forward)
            prod(I2{}, Zb, Za, avA, avB);
            if (grp == 0) { prod(I3{}, Za, Zb, avB, avA); dxl(Zb); }
            else dxl(Za);
        } else {
            const int key = L.st_hi - st;                     // (1, 2, ...: what the flags count)
            const float* const actsF = L.acts[0] + w.soff;
            if (grp == 0) {
                // dz_3 complete (the top product is not deferred here), handed over; then the upper groups of dz_2's blocks
                if (defer_top) {
                    mwt = mw;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) pre_top(ks);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) sX[(half * KS + ks) * 64 + lane] = Za[ks];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __hip_atomic_store(sFlag + half, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                PF partF;
                partF.prime(rs, voff, curB3);
                if (BITS) fetch_masks(2, avB);                // (activation mode: layer 2's rows are in avB already)
                partF.run(acc, zero, Za, rs, voff, curB3, [&](int) __attribute__((always_inline)) {},
                          [&](int jo, int r) __attribute__((always_inline)) {
                              if (BITS) Zb[4 * jo + r] = rr_mask_gate<KS>(mw, 4 * jo + r, acc[jo][r]);
                              else Zb[4 * jo + r] = (row_ok && avB[jo][r] > 0.f) ? acc[jo][r] : 0.f;
                          });
                save_dz(3, Za);
                if (!BITS) fetch_masks(1, avA);
                avt[0] = avB[TB]; avt[1] = avB[TB + 1];
                // the lower blocks, from the g_net wave
                while (__hip_atomic_load(sFlag + 2 + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != key) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < PG::K1; ++k) Zb[k] = sX2[(half * 16 + k) * 64 + lane];
                BWSTAMP(sbw + 3)
                prod(I2{}, Zb, Za, avA, avB);
                prod(I3{}, Za, Zb, avB, avA);
                BWSTAMP(sbw + 4)
                dxl(Zb);
            } else {
                // the lower groups of blocks of f_net's dz_2 for the same rows, before this wave's own chain
                unsigned mwF = 0u;
                f32x4 avF[PG::J1];
                if (BITS) {
                    mwF = reinterpret_cast<const unsigned*>(actsF + 2 * L.acts_ls[0])[srow * 4 + q];
                    mwF = row_ok ? mwF : 0u;
                } else {
                    const float* arow = actsF + 2 * L.acts_ls[0] + srow * HID;
#pragma unroll
                    for (int jo = 0; jo < PG::J1; ++jo) avF[jo] = rr_row_load<S>(arow, jo, q);
                }
                PG partG;
                partG.prime(rsF, voff, curB3);
                while (__hip_atomic_load(sFlag + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != key) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) Zb[ks] = sX[(half * KS + ks) * 64 + lane];
                partG.run(acc, zero, Zb, rsF, voff, curB3, [&](int) __attribute__((always_inline)) {},
                          [&](int, int) __attribute__((always_inline)) {});
                // (all of the range's blocks are still pending when it is a single group; with two groups the first was
                //  never finished by a hook either: every block is gated here)
#pragma unroll
                for (int jo = 0; jo < PG::J1; ++jo)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v;
                        if (BITS) v = rr_mask_gate<KS>(mwF, 4 * jo + r, acc[jo][r]);
                        else v = (row_ok && avF[jo][r] > 0.f) ? acc[jo][r] : 0.f;
                        sX2[(half * 16 + 4 * jo + r) * 64 + lane] = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __hip_atomic_store(sFlag + 2 + half, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                BWSTAMP(sbw + 3)
                prod(I1{}, Za, Zb, avB, avA);
                prod(I2{}, Zb, Za, avA, avB);
                BWSTAMP(sbw + 4)
                dxl(Za);
            }
        }
        BWSTAMP(sbw + 5)
        if (skip_dx) continue;
        __syncthreads();
        BWSTAMP(sbw + 6)
        // ---- stage algebra (rk_bwd_stage_algebra's arithmetic): dY = [dYup at the last stage] + dX_f + dX_g; dy0 += dY;
        //      dK_j += beta[st][j] h dY for j < st — one (row, component) per thread, every operand requested up front
        if (tid < NLBAC_MLP_TILE * RK_MAX_NS) {
            const int mm = tid >> 3, c = tid & 7, row = row0 + mm;
            const bool cv = c < ns, up = (w.gdYup || w.ip) && st == L.S_total - 1;
            const float xf = T.sDX[mm * RK_MAX_NS + c], xg = T.sDX[(NLBAC_MLP_TILE + mm) * RK_MAX_NS + c];
            const float y0 = T.sDY0[mm * RK_MAX_NS + c], h = T.sH[mm];
            float kj[RK_MAX_STAGES - 1], bn[RK_MAX_STAGES - 1];
#pragma unroll
            for (int j = 0; j < RK_MAX_STAGES - 1; ++j) {
                kj[j] = T.sDK[(j * NLBAC_MLP_TILE + mm) * RK_MAX_NS + c];
                bn[j] = sBeta[st * RK_MAX_STAGES + j];
            }
            float d = 0.f;
            if (up) d = w.ip ? T.sDYup[mm * RK_MAX_NS + c] : w.gdYup[(long)min(row, n - 1) * ns + min(c, ns - 1)];      // (uniform branch)
            d = (up && row < n) ? d : 0.f;
            d += xf;
            d += xg;
            if (cv) {
                T.sDY0[mm * RK_MAX_NS + c] = y0 + d;
#pragma unroll
                for (int j = 0; j < RK_MAX_STAGES - 1; ++j) {
                    const float t = kj[j] + (bn[j] * h) * d;
                    T.sDK[(j * NLBAC_MLP_TILE + mm) * RK_MAX_NS + c] = (j < st && bn[j] != 0.f) ? t : kj[j];
                }
            }
        }
        __syncthreads();
        BWSTAMP(sbw + 7)
    }
    __syncthreads();
    BWSTAMP(2 + 8 * 7)
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
    if (f->n_layers != 5 || g->n_layers != 4) return false;      // the layer chains are unrolled for the reference's depths
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

// the f_net / g_net wave balance (SPLIT, see the kernels); NLBAC_NODE_SPLIT=0 keeps one net per wave
static bool rr_split() {
    static const bool on = [] { const char* e = getenv("NLBAC_NODE_SPLIT"); return !(e && e[0] == '0'); }();
    return on;
}

int nlbac_node_rr_fwd_launch(NodeRkLaunch& L, hipStream_t s) {
    if (!nlbac_node_rr_eligible(&L.net[0], &L.net[1])) return 1;
    using KernelF = void (*)(const NodeRkLaunch);
    static const KernelF kf[2][3][2] = {{{node_rr_fwd_kernel<4, 4, 0, 0>, node_rr_fwd_kernel<4, 4, 1, 0>},
                                         {node_rr_fwd_kernel<7, 1, 0, 0>, node_rr_fwd_kernel<7, 1, 1, 0>},
                                         {node_rr_fwd_kernel<8, 4, 0, 0>, node_rr_fwd_kernel<8, 4, 1, 0>}},
                                        {{node_rr_fwd_kernel<4, 4, 0, 1>, node_rr_fwd_kernel<4, 4, 1, 1>},
                                         {node_rr_fwd_kernel<7, 1, 0, 1>, node_rr_fwd_kernel<7, 1, 1, 1>},
                                         {node_rr_fwd_kernel<8, 4, 0, 1>, node_rr_fwd_kernel<8, 4, 1, 1>}}};
    const size_t lds = (size_t)(RkFwdTile::floats() + NLBAC_MLP_TILE * 8 + 2 * 3 * 8 * 64 +
                                2 * 32 * 64 + NLBAC_MLP_TILE * RK_MAX_NS + 2 * 2 * 64 + 4) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.n, NLBAC_MLP_TILE));
    // (the forward is split in mask mode only: with activation rows kept — the NODE fit, 32768 rows, two workgroups per
    //  CU — the two waves' store bursts and the hand-over cost more than the balance gains: 140 against 124 us per launch)
    hipLaunchKernelGGL(kf[(rr_split() && L.acts_bits) ? 1 : 0][rr_shape_index(L.net[0].hid)][L.acts_bits ? 1 : 0], grid, dim3(256), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_rk_fwd(rr)");
    return 0;
}

// 0 = launched, 1 = not these nets' launch
int nlbac_node_rr_fwd_begin_launch(NodeRkLaunch& LA, NodeRkLaunch& LB, NodeRkLaunch& LC, hipStream_t s) {
    if (!nlbac_node_rr_eligible(&LA.net[0], &LA.net[1]) || !LA.acts_bits) return 1;
    using Kernel3 = void (*)(const NodeRkLaunch, const NodeRkLaunch, const NodeRkLaunch);
    static const Kernel3 k3[2][3] = {{node_rr_fwd_begin_kernel<4, 4, 0>, node_rr_fwd_begin_kernel<7, 1, 0>, node_rr_fwd_begin_kernel<8, 4, 0>},
                                     {node_rr_fwd_begin_kernel<4, 4, 1>, node_rr_fwd_begin_kernel<7, 1, 1>, node_rr_fwd_begin_kernel<8, 4, 1>}};
    const size_t lds = (size_t)(RkFwdTile::floats() + NLBAC_MLP_TILE * 8 + 2 * 3 * 8 * 64 +
                                2 * 32 * 64 + NLBAC_MLP_TILE * RK_MAX_NS + 2 * 2 * 64 + 4) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(LA.n, NLBAC_MLP_TILE));
    hipLaunchKernelGGL(k3[rr_split() ? 1 : 0][rr_shape_index(LA.net[0].hid)], grid, dim3(256), lds, s, LA, LB, LC);
    NLBAC_CHECK_LAUNCH("nlbac_node_rk_fwd_begin(rr)");
    return 0;
}

int nlbac_node_rr_bwd_launch(NodeRkBwdLaunch& L, hipStream_t s) {
    if (!nlbac_node_rr_eligible(&L.net[0], &L.net[1])) return 1;
    using KernelB = void (*)(const NodeRkBwdLaunch);
    static const KernelB kb[2][3][2] = {{{node_rr_bwd_kernel<4, 4, 0, 0>, node_rr_bwd_kernel<4, 4, 1, 0>},
                                         {node_rr_bwd_kernel<7, 1, 0, 0>, node_rr_bwd_kernel<7, 1, 1, 0>},
                                         {node_rr_bwd_kernel<8, 4, 0, 0>, node_rr_bwd_kernel<8, 4, 1, 0>}},
                                        {{node_rr_bwd_kernel<4, 4, 0, 1>, node_rr_bwd_kernel<4, 4, 1, 1>},
                                         {node_rr_bwd_kernel<7, 1, 0, 1>, node_rr_bwd_kernel<7, 1, 1, 1>},
                                         {node_rr_bwd_kernel<8, 4, 0, 1>, node_rr_bwd_kernel<8, 4, 1, 1>}}};
    const size_t lds = (size_t)(RkBwdTile::floats() + 2 * 4 * 8 * 64 + 2 * 32 * 64 + 2 * 16 * 64 + 4) * sizeof(float);
    const dim3 grid(nlbac_ceil_div(L.n, NLBAC_MLP_TILE));
    hipLaunchKernelGGL(kb[rr_split() ? 1 : 0][rr_shape_index(L.net[0].hid)][L.acts_bits ? 1 : 0], grid, dim3(256), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_rk_bwd(rr)");
    return 0;
}
