// Fused Runge-Kutta step of the single-net NODE  dx/dt = net([x | c])  (SimulatedCars: C/sac_cbf_clf/model.py:179-205,
// odeint call sites C/sac_cbf_clf/sac_cbf_clf.py:437,458,581,603 and C/model.py:245; the Quadrotor-like task's
// normalised form) with REGISTER-RESIDENT layer chains (rr_device.h): the launches of nlbac_concat_rk_fwd / _bwd for
// the reference's depth (in -> hid -> hid -> hid -> out) and widths 64 / 100 / 128.
//
// One wave owns 16 rows for the whole launch — every stage's input, the net's four products and the stage algebra — so
// nothing is exchanged between waves: no barrier inside the stage loop (the LDS arrays of a wave's rows are private to
// it), a workgroup is just two such waves on one 32-row tile of the problem bookkeeping.  The LDS-tiled kernels give a
// 64-wide layer's two column tiles to two of a group's four waves (concat_rk_fwd 22 TFLOP/s, profiles/r02_bench_variant_cars.json).
#include "concat_rk_shared.h"
#include "rr_device.h"
#include <cstdlib>
#include <type_traits>

#define CRR_MAX_IN 15       /* in_dim + the bias column <= 16: four k-steps of layer 0 */

// this lane's state component in register r: c = 4 r + q (the layout of layer 0's B operand and of the output block)
#ifdef RR_TIMING
#define CSTAMP(slot_) if (L.err && blockIdx.x == 0 && lane == 0) reinterpret_cast<long long*>(L.err)[half * 256 + (slot_)] = (long long)__builtin_readcyclecounter();
#define BSTAMP(slot_) if (L.dyn && !L.norm && blockIdx.x == 0 && lane == 0) reinterpret_cast<long long*>(L.dyn)[half * 256 + (slot_)] = (long long)__builtin_readcyclecounter();
#else
#define CSTAMP(slot_)
#define BSTAMP(slot_)
#endif
// NW: waves per workgroup (2 or 4: 32- or 64-row tiles).  Four where the problems' row counts allow it (a tile must not
// straddle two problems of a device-driven chain): of two 2-wave workgroups on one CU the hardware puts two waves on the
// same SIMD and leaves one SIMD empty (tools/micro/wave_place.hip: 512 workgroups x 128 threads use 768 of the 1024 SIMDs,
// 256 of them twice) — the doubled-up waves ran a stage in 11-13k cycles against 8.2k, and the launch waits for them.
template <int NB, int R, int BITS, int NW>
__global__ __launch_bounds__(64 * NW) void concat_rr_fwd_kernel(const ConcatRkLaunch L) {
    constexpr int TILE = 16 * NW, NTHR = 64 * NW;
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID, TB = NB - 2, NT = KS - 4 * TB, G0 = rr_group_first(NB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = L.n, ns = L.n_s, nc = L.n_c;
    const int row0 = blockIdx.x * TILE;
    const int p_tile = row0 / L.rpp;
    long soff = 0;
    bool fsal = false, ip = false;
    float ip_x = 0.f;
    if (L.ctl) {
        const double* c = L.ctl + (long)p_tile * NLBAC_DOPRI_CTL;
        if (c[C_DONE] > 0.0) return;              // (uniform) this problem's solve has finished
        const int slot = (int)c[C_NACC];
        soff = (long)slot * L.slot_floats;
        fsal = slot > 0;
        if (L.ip_out) {          // this attempt reaches t_end: it also evaluates the solve's result (node_rk_shared.h::rk_fwd_where)
            const double t = c[C_T], hd = c[C_H];
            ip = t + hd >= L.t_end;
            ip_x = (float)((L.t_end - t) / hd);
        }
    }
    float* const gK = L.K + soff;
    float* const gY = L.Y + soff;
    float* const gErr = L.err ? L.err + soff : nullptr;
    float* const gXn = L.Xn ? L.Xn + soff : nullptr;
    const float* const gy0 = fsal ? (gY - L.slot_floats) + (long)(L.S_total - 1) * n * ns : L.y0;
    const nlbac_mlp& net = L.net;
    const int idim = net.in_dim;
    const int n_rows = min(TILE, n - row0);
    const int q = lane >> 4, r16 = lane & 15, m = 16 * half + r16, grow = row0 + m;
    const bool row_ok = grow < n;
    const int KS0 = (ns + 3) >> 2;      // registers per row of state (layer 0 always runs four k-steps over [x | c | 1 | 0..])
    const float* const params = net.params;
    float* const acts = L.acts ? L.acts + soff : nullptr;
    const long acts_ls = L.acts_ls;
    const float* const nrm = L.norm;
    const int stage_end = L.stage_end;

    float* sK = smem;                                               // [stage][32][CK_NS]
    float* sY0 = sK + CK_MAX_STAGES * TILE * CK_LD;       // [32][CK_NS]
    float* sC = sY0 + TILE * CK_LD;                       // [32][CK_NC]
    float* sH = sC + TILE * CK_NC;                        // [32]
    float* sW0 = sH + TILE;                               // [k-step < 4][block < 8][lane]: layer 0's A fragments

    // ---- the wave's weight stream: hid x hid layers 1, 2, then 1 again (next stage)
    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16, wbase = net.rr_fwd_off * 4;
    RRGemm<S> gemm;
    CSTAMP(0)
    gemm.prime(rs, voff, wbase);

    // ---- constants: layer 0's A fragments over [x | c | 1] (bias in the column behind the inputs) -> LDS, the output
    //      layer's into registers (A row 4 q' + r' computes state component 4 r' + q')
    {       // (the workgroup's waves share the job: k-step k0 by wave k0 mod NW)
        const float* W0 = params + net.w_off[0];
        const float* b0 = params + net.b_off[0];
#pragma unroll
        for (int kk = 0; kk < 4 / NW; ++kk)
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int k0 = half + NW * kk;
                const int uo = rr_unit_out(NB, R, jo, r16), col = 4 * k0 + q, uc = max(uo, 0);
                const float vw = W0[uc * idim + min(col, idim - 1)], vb0 = b0[uc];
                sW0[(k0 * 8 + jo) * 64 + lane] = (uo < 0 || col > idim) ? 0.f : (col < idim ? vw : vb0);
            }
    }
    float wo[KS];
    {
        const int cq = 4 * (r16 & 3) + (r16 >> 2);
        const bool ok = (r16 & 3) < KS0 && cq < ns;
        const float* wrow = params + net.w_off[3] + (long)(ok ? cq : 0) * HID;
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) {
            const f32x4 v = rr_row_load<S>(wrow, jo, q);
#pragma unroll
            for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) wo[4 * jo + r] = ok ? v[r] : 0.f;
        }
    }
    // (every load of the prologue is unconditional — clamped index, select afterwards — so that they are all in flight
    // together: as guarded loads each was a branch and an L2 round trip of its own, 8k cycles before the first stage)
    float o_bias[4], o_mu[4], o_sig[4];
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
    }
    // this lane's input columns 4 k0 + q: where they come from, their normalisation
    float i_mu[4], i_isig[4];
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) {
        const int col = 4 * k0 + q, cc = min(col, idim - 1);
        const float vm = nrm_v[min(cc, nrm_n - 1)], vs = nrm_v[min(idim + cc, nrm_n - 1)];
        i_mu[k0] = (nrm && col < idim) ? vm : 0.f;
        i_isig[k0] = (nrm && col < idim) ? vs : 1.f;
    }
    // ---- this wave's rows of the tile constants
    for (int idx = lane; idx < 16 * CK_NS; idx += 64) {
        const int mm = 16 * half + idx / CK_NS, c = idx % CK_NS, row = row0 + mm;
        const float v = gy0[(long)min(row, n - 1) * ns + min(c, ns - 1)];
        sY0[mm * CK_LD + c] = (row < n && c < ns) ? v : 0.f;
    }
    {
        const int mm = 16 * half + (lane >> 2), c = lane & 3, row = row0 + mm;
        const float v = L.c[(long)min(row, n - 1) * nc + min(c, max(nc - 1, 0))];
        sC[mm * CK_NC + c] = (row < n && c < nc) ? v : 0.f;
    }
    if (lane < 16) {
        const int p = min(row0 + 16 * half + lane, n - 1) / L.rpp;
        sH[16 * half + lane] = L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p];
    }
    if (L.stage_begin > 0) {      // stages of an earlier launch: all loads first, then the LDS (and FSAL) stores
        constexpr int NIT = CK_MAX_STAGES * 16 * CK_NS / 64;
        float vals[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + 64 * it;
            const int j = idx / (16 * CK_NS), rem = idx - j * 16 * CK_NS;           // (j is uniform per iteration)
            const int mm = 16 * half + rem / CK_NS, c = rem % CK_NS, row = row0 + mm;
            const long rc = (long)min(row, n - 1) * ns + min(c, ns - 1);
            float v = 0.f;
            if (j < L.stage_begin) {
                if (fsal && j == 0) v = (gK - L.slot_floats)[(long)(L.S_total - 1) * n * ns + rc];   // first stage = the previous slot's last
                else v = gK[(long)j * n * ns + rc];
            }
            vals[it] = (row < n && c < ns) ? v : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + 64 * it;
            const int j = idx / (16 * CK_NS), rem = idx - j * 16 * CK_NS;
            const int mm = 16 * half + rem / CK_NS, c = rem % CK_NS, row = row0 + mm;
            if (j < L.stage_begin) {
                sK[(j * TILE + mm) * CK_LD + c] = vals[it];
                if (fsal && j == 0 && row < n && c < ns) gK[(long)row * ns + c] = vals[it];   // kept in this slot for the interpolant
            }
        }
    }
    // narrow nets keep both hid x hid layers' biases in registers for the whole launch: a bias load inside the stage loop
    // queues behind the previous stage's stores (vmcnt is in order) and the layer's first MFMAs need it as their C operand
    constexpr bool BRES = NB <= 4;
    f32x4 bres[BRES ? 2 : 1][NB];
    if (BRES) {
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) bres[BRES ? l : 0][jo] = rr_bias<S>(params + net.b_off[1 + l], jo, q);
    }
    __syncthreads();           // (sW0 is shared by the two waves; everything else above is the wave's own rows)
    CSTAMP(1)

    // (a stage's tableau row is a scalar load from the kernel arguments: requested one stage ahead, its latency — a
    // scalar-cache miss per stage — is off the stage's critical path)
    float bnext[CK_MAX_STAGES];
#pragma unroll
    for (int j = 0; j < CK_MAX_STAGES; ++j) bnext[j] = L.beta[L.stage_begin][j];
    for (int st = L.stage_begin; st < stage_end; ++st) {
        const int sb = 2 + 8 * (st - L.stage_begin);
        (void)sb;
        CSTAMP(sb + 0)
        float bn[CK_MAX_STAGES];
#pragma unroll
        for (int j = 0; j < CK_MAX_STAGES; ++j) bn[j] = bnext[j];
        const long srow = (long)st * n + grow;
        // ---- stage input [Y_st | c | 1] in registers,  Y_st = y0 + h sum_j beta[st][j] K_j  (rk_combine_kernel's op order)
        // What a stage leaves in global memory — Y_st, K_st, the layers' mask words — is stored in ONE burst at its end:
        // vmcnt counts stores and loads in order, so a store issued between two weight-fragment loads makes the MFMAs
        // behind the second one wait for the store's trip to HBM (a 64-wide layer is 2k cycles of MFMAs: one such wait
        // per layer doubled it).  Behind the burst come the next stage's input and layer 0, which need no load.
        float yv[4], ykeep[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned wsave0 = 0u, wsave1 = 0u, wsave2 = 0u;
        {
            // (no data-dependent branch: every lane requests its operands with clamped indices, all reads in flight
            // together, selects at the end — the per-column if / else ladder was four serialised LDS round trips)
            const float h = sH[m];
            float y0v[4], cv[4], kv[4][CK_MAX_STAGES - 1];
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) {
                const int col = 4 * k0 + q, cs = min(col, ns - 1), cc = min(max(col - ns, 0), max(nc - 1, 0));
                y0v[k0] = sY0[m * CK_LD + cs];
                cv[k0] = sC[m * CK_NC + cc];
#pragma unroll
                for (int j = 0; j < CK_MAX_STAGES - 1; ++j) kv[k0][j] = sK[(j * TILE + m) * CK_LD + cs];
            }
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0) {
                const int col = 4 * k0 + q;
                float a = y0v[k0];
#pragma unroll
                for (int j = 0; j < CK_MAX_STAGES - 1; ++j) {
                    const float t = a + kv[k0][j] * (bn[j] * h);
                    a = (j < st && bn[j] != 0.f) ? t : a;
                }
                ykeep[k0] = a;
                a = (col < ns) ? a : ((col < idim) ? cv[k0] : 0.f);
                if (nrm) {
                    a = (col < idim) ? (a - i_mu[k0]) * i_isig[k0] : a;
                    if (gXn && row_ok && col < idim) gXn[srow * idim + col] = a;
                }
                yv[k0] = (col == idim) ? 1.f : a;
            }
        }
        float Ha[KS], Hb[KS];
        f32x4 acc0[NB], acc[NB], bv[NB], bpre[3];
        auto prefetch_bias = [&](int l) __attribute__((always_inline)) {
            if (BRES) return;
#pragma unroll
            for (int jo = 0; jo < G0; ++jo) bpre[jo] = rr_bias<S>(params + net.b_off[l], jo, q);
        };
        prefetch_bias(1);
        unsigned wd = 0u;                 // (mask mode) the layer's mask word, values shifted in in ascending register order
        auto save_block = [&](int l, int jo, const float (&H)[KS]) __attribute__((always_inline)) {
            if (BITS || !acts || !row_ok) return;
            f32x4 hv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) hv[rr] = H[4 * jo + rr];
            rr_row_store<S>(acts + (long)l * acts_ls + srow * HID, jo, q, hv);
        };
        auto save_word = [&](int l) __attribute__((always_inline)) {       // (kept; stored with the stage's burst)
            if (l == 0) wsave0 = wd; else if (l == 1) wsave1 = wd; else wsave2 = wd;
        };
        auto pre_l0 = [&](int ks) __attribute__((always_inline)) {
            const int jo = (ks < 4 * (NB - 1)) ? (ks >> 2) : NB - 1, r = ks - 4 * jo;
            const float h = rr_relu(acc0[jo][r]);
            Ha[ks] = h;
            if (BITS) rr_mask_push(wd, h);
            if (r == ((jo < NB - 1) ? 3 : R - 1)) save_block(0, jo, Ha);
            if (ks == KS - 1) save_word(0);
        };
        auto pre_tail = [&](int lp, float (&H)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            const float h = rr_relu(acc[jo][r]);
            H[4 * TB + t] = h;
            if (BITS) rr_mask_push(wd, h);
            if (t == 3 || t == NT - 1) save_block(lp, jo, H);
            if (t == NT - 1) save_word(lp);
        };
        CSTAMP(sb + 1)
        // ---- layer 0 (bias folded into the product): always four k-steps — fragments and inputs past the width are zero —
        //      with the k-step outside, so that the MFMAs of one block are NB issue slots apart and nothing branches
        {
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
        // ---- the two hid x hid layers, then the output layer (statically unrolled, as node_rr_kernels.hip)
        auto wide = [&](auto lc, float (&Hin)[KS], float (&Hout)[KS]) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo)
                bv[jo] = BRES ? bres[BRES ? l - 1 : 0][jo] : ((jo < G0) ? bpre[jo] : rr_bias<S>(params + net.b_off[l], jo, q));
            __builtin_amdgcn_sched_barrier(0);
            if (l == 1) {
                // the next stage's tableau row: a scalar load, requested HERE — behind the stage's last LDS wait (scalar
                // and LDS loads share lgkmcnt and scalar loads return out of order, so a wait for LDS data is a wait
                // for every scalar load in flight) and ahead of two layers of MFMAs that need neither
                int sn = min(st + 1, L.S_total - 1);
                asm volatile("" : "+s"(sn));
#pragma unroll
                for (int j = 0; j < CK_MAX_STAGES; ++j) bnext[j] = L.beta[sn][j];
                __builtin_amdgcn_sched_barrier(0);
            }
            const int cur = wbase + (l - 1) * S::LAYER_BYTES;
            const int nxt = (l == 1) ? cur + S::LAYER_BYTES : wbase;
            gemm.run(acc, bv, Hin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) {
                         if (l == 1) pre_l0(ks);
                         else pre_tail(l - 1, Hin, ks);
                     },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         const float h = rr_relu(acc[jo][r]);
                         Hout[4 * jo + r] = h;
                         if (BITS) rr_mask_push(wd, h);
                         if (r == 3) save_block(l, jo, Hout);
                     },
                     [&]() __attribute__((always_inline)) { if (l == 1) prefetch_bias(2); });
        };
        CSTAMP(sb + 2)
        wide(std::integral_constant<int, 1>{}, Ha, Hb);
        CSTAMP(sb + 3)
        wide(std::integral_constant<int, 2>{}, Hb, Ha);
        CSTAMP(sb + 4)
        {
            const f32x4 o = RRGemm<S>::block(wo, Ha, [&](int ks) __attribute__((always_inline)) { pre_tail(2, Ha, ks); });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * r + q;
                if (r < KS0 && c < ns) {
                    const float val = (o[r] + o_bias[r]) * o_sig[r] + o_mu[r];
                    sK[(st * TILE + m) * CK_LD + c] = val;
                    if (row_ok) gK[srow * ns + c] = val;
                }
            }
#pragma unroll
            for (int k0 = 0; k0 < 4; ++k0)
                if (row_ok && 4 * k0 + q < ns) gY[srow * ns + 4 * k0 + q] = ykeep[k0];
            if (BITS && acts && row_ok) {
                unsigned* wp = reinterpret_cast<unsigned*>(acts) + srow * 4 + q;
                wp[0] = wsave0; wp[acts_ls] = wsave1; wp[2 * acts_ls] = wsave2;
            }
        }
        CSTAMP(sb + 5)
    }
    __syncthreads();
    CSTAMP(2 + 8 * (stage_end - L.stage_begin))
#ifdef RR_TIMING
    if (L.err) return;
#endif

    // ---- step outputs
    for (int idx = tid; idx < TILE * ns; idx += NTHR) {
        const int mm = idx / ns, r = idx - mm * ns, row = row0 + mm;
        if (row >= n) continue;
        const float h = sH[mm];
        if (L.out) {
            float a = sY0[mm * CK_LD + r];
            for (int j = 0; j < L.n_out; ++j)
                if (L.c_out[j] != 0.f) a = a + sK[(j * TILE + mm) * CK_LD + r] * (L.c_out[j] * h);
            L.out[(long)row * ns + r] = a;
        }
        if (gErr) {
            float a = 0.f;
            for (int j = 0; j < L.n_err; ++j)
                if (L.c_err[j] != 0.f) a = a + sK[(j * TILE + mm) * CK_LD + r] * (L.c_err[j] * h);
            gErr[(long)row * ns + r] = a;
        }
    }
    if (ip && tid < n_rows) {      // the interpolant at t_end, should this attempt be accepted: one thread per row
        const int mm = tid, row = row0 + mm, sl = L.S_total - 1;
        const float h = sH[mm];
        for (int r = 0; r < ns; ++r) {
            const float a0 = sY0[mm * CK_LD + r];
            float a1 = a0, k[7];
            for (int j = 0; j < sl; ++j)
                if (L.beta[sl][j] != 0.f) a1 = a1 + sK[(j * TILE + mm) * CK_LD + r] * (L.beta[sl][j] * h);
#pragma unroll
            for (int j = 0; j < 7; ++j) k[j] = sK[(j * TILE + mm) * CK_LD + r];
            L.ip_out[(long)row * ns + r] = dopri_interp_value(a0, a1, k, h, ip_x);
        }
    }
    // ---- fused step control (as concat_rk_fwd_kernel): tile partial sums, one ticket per problem, last workgroup = controller
    if (L.norm_mode < 0) return;
    __shared__ unsigned s_last;
    if (tid < 64) {
        const int mm = tid;
        float v0 = 0.f, v1 = 0.f;
        if (mm < n_rows) {
            const float h = sH[mm];
            for (int r = 0; r < ns; ++r) {
                const float y = sY0[mm * CK_LD + r];
                if (L.norm_mode == 2) {
                    float e = 0.f, y1 = y;
                    for (int j = 0; j < L.n_err; ++j)
                        if (L.c_err[j] != 0.f) e = e + sK[(j * TILE + mm) * CK_LD + r] * (L.c_err[j] * h);
                    const int sl = L.S_total - 1;
                    for (int j = 0; j < sl; ++j)
                        if (L.beta[sl][j] != 0.f) y1 = y1 + sK[(j * TILE + mm) * CK_LD + r] * (L.beta[sl][j] * h);
                    const float qq = e / (L.atol + L.rtol * fmaxf(fabsf(y), fabsf(y1)));
                    v0 += qq * qq;
                } else {
                    const float sc = L.atol + fabsf(y) * L.rtol;
                    if (L.norm_mode == 0) {
                        const float q0 = y / sc, q1 = sK[mm * CK_LD + r] / sc;
                        v0 += q0 * q0; v1 += q1 * q1;
                    } else {
                        const float qq = (sK[(TILE + mm) * CK_LD + r] - sK[mm * CK_LD + r]) / sc;
                        v0 += qq * qq;
                    }
                }
            }
            if (L.norm_mode == 0)
                for (int c = 0; c < nc; ++c) {
                    const float y = sC[mm * CK_NC + c];
                    const float qq = y / (L.atol + fabsf(y) * L.rtol);
                    v0 += qq * qq;
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_down(v0, off, 64); v1 += __shfl_down(v1, off, 64); }
        if (tid == 0) {
            const int nblk = (L.rpp + TILE - 1) / TILE;
            const int blk = (row0 - p_tile * L.rpp) / TILE;
            float* pq = L.partials + ((long)p_tile * nblk + blk) * 2;
            const float o0 = __hip_atomic_exchange(pq + 0, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float o1 = __hip_atomic_exchange(pq + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(o0), "v"(o1) : "memory");
            const unsigned ticket = __hip_atomic_fetch_add(L.tickets + p_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (ticket == (unsigned)nblk - 1u) ? 1u : 0u;
            if (s_last) __hip_atomic_store(L.tickets + p_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last || tid >= 64) return;
    {
        const int nblk = (L.rpp + TILE - 1) / TILE;
        double d0 = 0.0, d1 = 0.0;
        for (int b = tid; b < nblk; b += 64) {
            const float* pq = L.partials + ((long)p_tile * nblk + b) * 2;
            d0 += (double)__hip_atomic_load(pq + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d1 += (double)__hip_atomic_load(pq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); d1 += __shfl_down(d1, off, 64); }
        if (tid == 0) {
            const double cnt = (double)L.rpp * (double)(ns + nc);
            double* c = L.ctl_w + (long)p_tile * NLBAC_DOPRI_CTL;
            const int slot_before = (int)c[C_NACC];
            const double h_try = c[C_H];
            dopri_control_vals(sqrt(d0 / cnt), sqrt(d1 / cnt), p_tile, L.norm_mode, L.t_end, L.ctl_w, L.n_slots);
            if (L.norm_mode == 2 && L.hslots && c[C_ACCEPT] > 0.0) L.hslots[(long)p_tile * L.n_slots + slot_before] = h_try;
            if (L.norm_mode == 2 && L.alog) {
                const int k = (int)c[C_NSTEPS] - 1;
                if (k >= 0 && k < L.alog_cap) {
                    double* a = L.alog + ((long)p_tile * L.alog_cap + k) * 3;
                    a[0] = h_try; a[1] = c[C_RATIO]; a[2] = c[C_ACCEPT];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the same step, same wave roles: per stage (descending) the output layer's gradient enters one transposed
// block product, dz runs down the chain in registers (backward RR pack), dX = W_0^T dz_0 is one more block product whose
// state columns feed the stage algebra and whose carried columns accumulate dc — all on the wave's own 16 rows.
// ---------------------------------------------------------------------------------------------------------------------
template <int NB, int R, int BITS, int NW>
__global__ __launch_bounds__(64 * NW) void concat_rr_bwd_kernel(const ConcatRkBwdLaunch L) {
    constexpr int TILE = 16 * NW;
    using S = RRShape<NB, R>;
    constexpr int KS = S::KS, HID = S::HID, TB = NB - 2, NT = KS - 4 * TB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = L.n, ns = L.n_s, nc = L.n_c;
    const int row0 = blockIdx.x * TILE;
    long soff = 0;
    int slot = 0;
    const bool chained = L.ctl != nullptr;
    if (chained) {
        slot = (int)L.ctl[(long)(row0 / L.rpp) * NLBAC_DOPRI_CTL + C_NACC] - L.back_idx;
        if (slot < 0) return;                       // (uniform) this problem took fewer steps
        soff = (long)slot * L.slot_floats;
    }
    const bool carry = chained && L.back_idx > 0;
    const bool ip = L.ip_on && chained && !carry;      // dK / dy0 / dy1 of the last step from d loss / d y(t_end): no interp launch
    float* const gdK = L.dK + soff;
    float* const gdy0 = L.dy0 ? L.dy0 + soff : nullptr;
    float* const gdyn = L.dyn ? L.dyn + soff : nullptr;
    const float* const gdYup = carry ? gdy0 + L.slot_floats : (L.dYup ? L.dYup + soff : nullptr);
    const nlbac_mlp& net = L.net;
    const int idim = net.in_dim;
    const bool keep_dz = L.dz != nullptr;
    const int q = lane >> 4, r16 = lane & 15, m = 16 * half + r16, grow = row0 + m;
    const bool row_ok = grow < n;
    const int growc = min(grow, n - 1);
    const int KS0 = (ns + 3) >> 2;
    const float* const params = net.params;
    const float* const acts = L.acts + soff;
    float* const dz = keep_dz ? L.dz + soff : nullptr;
    const long acts_ls = L.acts_ls;
    const float* const nrm = L.norm;
    const int dx_stage0 = L.dx_stage0;

    float* sDK = smem;                                              // [stage][32][CK_NS]
    float* sH = sDK + CK_MAX_STAGES * TILE * CK_LD;       // [32]
    float* sDY0 = sH + TILE;                              // [32][CK_NS] running dy0
    float* sDC = sDY0 + TILE * CK_LD;                     // [32][CK_NC] running d carried
    float* sDX = sDC + TILE * CK_NC;                      // [32][16] dX of the current stage (input columns)
    float* sWt = sDX + TILE * 16;                         // [k-step < 4][block < 8][lane]: W_out^T's A fragments
    float* sDYup = sWt + 4 * 8 * 64;                                // [32][CK_NS] dL/dy1 when the launch forms it itself (ip)

    const int st_lo = chained ? (slot == 0 ? 0 : 1) : L.st_lo;
    const bool stage0_data = dx_stage0 || keep_dz;
#define crr_has_data(st_) ((st_) >= st_lo && ((st_) > 0 || stage0_data))

    const __amdgpu_buffer_rsrc_t rs = rr_rsrc(net.packed, net.packed_floats);
    const int voff = lane * 16, wbase = net.rr_bwd_off * 4;
    RRGemm<S> gemm;
    BSTAMP(0)
    gemm.prime(rs, voff, wbase + S::LAYER_BYTES);           // (layer 2's fragments first, then layer 1's)

    {       // (k-step e by wave e mod NW)
        const float* Wl = params + net.w_off[3];
#pragma unroll
        for (int ee = 0; ee < 4 / NW; ++ee) {
            const int e = half + NW * ee;
            const int c = 4 * e + q;
            const bool ok = e < KS0 && c < ns;
#pragma unroll
            for (int jo = 0; jo < NB; ++jo) {
                const int uo = rr_unit_out(NB, R, jo, r16);
                const float v = Wl[(long)min(c, ns - 1) * HID + max(uo, 0)];      // (unconditional: see the forward's prologue)
                sWt[(e * 8 + jo) * 64 + lane] = (ok && uo >= 0) ? v : 0.f;
            }
        }
    }
    float w0t[KS];        // A of dX: row i (< in_dim) of W_0^T
    {
        const float* W0 = params + net.w_off[0];
        const bool ok = r16 < idim;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float v = W0[(long)rr_unit_in(NB, R, ks, q) * idim + min(r16, idim - 1)];
            w0t[ks] = ok ? v : 0.f;
        }
    }
    float o_sig[4], x_isig[4];
    const float* const nrm_v = nrm ? nrm : params;
    const int nrm_n = nrm ? 2 * idim + 2 * ns : 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * r + q;
        const float vs = nrm_v[min(2 * idim + ns + min(c, ns - 1), nrm_n - 1)];
        o_sig[r] = (nrm && r < KS0 && c < ns) ? vs : 1.f;
        const int i = 4 * q + r;                     // dX leaves lane (q, row) with input column 4 q + r in register r
        const float vi = nrm_v[min(idim + min(i, idim - 1), nrm_n - 1)];
        x_isig[r] = (nrm && i < idim) ? vi : 1.f;
    }
    // ---- this wave's rows of the tile constants
    {       // (uniform conditions branch; per-lane ones clamp the address and select: every load in flight at once)
        const int mm = 16 * half + (lane >> 2), c = lane & 3, row = row0 + mm;
        float v = 0.f;
        if (L.dc && L.dc_acc) v = L.dc[(long)min(row, n - 1) * nc + min(c, max(nc - 1, 0))];
        sDC[mm * CK_NC + c] = (row < n && c < nc) ? v : 0.f;
    }
    if (ip) {       // (uniform) the interpolant's backward for this wave's rows (ode_kernels.hip::dopri_interp_bwd_kernel's arithmetic)
#pragma unroll
        for (int it = 0; it < (16 * CK_LD + 63) / 64; ++it) {
            const int idx = lane + 64 * it;
            const int mm = 16 * half + idx / CK_NS, c = idx % CK_NS, row = row0 + mm, rowc = min(row, n - 1), p = rowc / L.rpp;
            const float hh = (float)L.ctl[(long)p * NLBAC_DOPRI_CTL + C_HUSED], xx = (float)L.ctl[(long)p * NLBAC_DOPRI_CTL + C_X];
            const float g = L.ip_dout[(long)rowc * ns + min(c, ns - 1)];
            float d0v, d1v, dk[7];
            dopri_interp_grad(g, hh, xx, d0v, d1v, dk);
            const bool ok = row < n && c < ns;
            if (idx < 16 * CK_NS) {
                sDY0[mm * CK_LD + c] = ok ? d0v : 0.f;
                sDYup[mm * CK_LD + c] = ok ? d1v : 0.f;
#pragma unroll
                for (int j = 0; j < 7; ++j) sDK[(j * TILE + mm) * CK_LD + c] = ok ? dk[j] : 0.f;
            }
        }
    } else {
        const bool have = gdy0 && L.dy0_in && !carry;
#pragma unroll
        for (int it = 0; it < (16 * CK_LD + 63) / 64; ++it) {
            const int idx = lane + 64 * it;
            const int mm = 16 * half + idx / CK_NS, c = idx % CK_NS, row = row0 + mm;
            float v = 0.f;
            if (have) v = gdy0[(long)min(row, n - 1) * ns + min(c, ns - 1)];
            if (idx < 16 * CK_NS) sDY0[mm * CK_LD + c] = (row < n && c < ns) ? v : 0.f;
        }
    }
    if (lane < 16) {
        const int p = min(row0 + 16 * half + lane, n - 1) / L.rpp;
        sH[16 * half + lane] = chained ? (float)L.hslots[(long)p * L.n_slots + slot]
                                       : (L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p]);
    }
    if (!ip) {   // dK of every stage into LDS: all loads first (a loop with a run-time bound and the LDS store behind each load
        // was one global round trip per iteration: 16 in a row for rk4, most of the launch's prologue)
        constexpr int NIT = CK_MAX_STAGES * 16 * CK_NS / 64;
        float vals[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + 64 * it;
            const int j = idx / (16 * CK_NS), rem = idx - j * 16 * CK_NS;           // (j is uniform: 16 * CK_NS is a multiple of 64)
            const int mm = 16 * half + rem / CK_NS, c = rem % CK_NS, row = row0 + mm;
            const long rc = (long)min(row, n - 1) * ns + min(c, ns - 1);
            float v = 0.f;
            if (j < L.st_hi) {
                if (!carry) v = gdK[(long)j * n * ns + rc];
                else if (j == L.S_total - 1) v = (gdK + L.slot_floats)[rc];     // FSAL: next slot's dK[0]
            }
            vals[it] = (row < n && c < ns) ? v : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + 64 * it;
            const int j = idx / (16 * CK_NS), rem = idx - j * 16 * CK_NS;
            const int mm = 16 * half + rem / CK_NS, c = rem % CK_NS;
            if (j < L.st_hi) sDK[(j * TILE + mm) * CK_LD + c] = vals[it];
        }
    }
    __syncthreads();           // (sWt is shared by the two waves)

    f32x4 zero[NB];
#pragma unroll
    for (int jo = 0; jo < NB; ++jo) zero[jo] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (mask mode) a stage's three mask words are requested while the stage before it runs: a load from HBM issued at the
    // start of a product would hold back every fragment load behind it (vmcnt is in order) for longer than the product's
    // own MFMAs take
    unsigned mnext0 = 0u, mnext1 = 0u, mnext2 = 0u;
    auto request_masks = [&](int stn) __attribute__((always_inline)) {
        if (!BITS || stn < st_lo) return;
        const unsigned* wp = reinterpret_cast<const unsigned*>(acts) + ((long)stn * n + growc) * 4 + q;
        mnext0 = wp[0]; mnext1 = wp[acts_ls]; mnext2 = wp[2 * acts_ls];
    };
    request_masks(L.st_hi - 1);
    float bnext[CK_MAX_STAGES];
#pragma unroll
    for (int j = 0; j < CK_MAX_STAGES; ++j) bnext[j] = L.beta[max(L.st_hi - 1, 0)][j];
    for (int st = L.st_hi - 1; st >= st_lo; --st) {
        const unsigned mcur0 = mnext0, mcur1 = mnext1, mcur2 = mnext2;
        request_masks(st - 1);
        float bn[CK_MAX_STAGES];
#pragma unroll
        for (int j = 0; j < CK_MAX_STAGES; ++j) bn[j] = bnext[j];
        {
            const int sn = max(st - 1, 0);          // (requested one stage ahead: see the forward)
#pragma unroll
            for (int j = 0; j < CK_MAX_STAGES; ++j) bnext[j] = L.beta[sn][j];
        }
        if (!crr_has_data(st)) continue;      // (uniform; the dyn of such a stage is not wanted either)
        const int sbb = 2 + 8 * st;
        (void)sbb;
        BSTAMP(sbb + 0)
        const long srow = (long)st * n + growc;
        // ---- the output layer's gradient: dK (times out_sig), also kept for the weight gradients of a normalised field
        float dy[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * e + q;
            const float raw = sDK[(st * TILE + m) * CK_LD + min(c, ns - 1)];
            const float v = (e < KS0 && c < ns) ? raw * o_sig[e] : 0.f;
            if (gdyn && nrm && row_ok && e < KS0 && c < ns) gdyn[((long)st * n + grow) * ns + c] = v;
            dy[e] = v;
        }
        float Za[KS], Zb[KS];
        f32x4 acct[NB], acc[NB], av[NB], avt[2];
        unsigned mw = 0u, mwt = 0u;
        auto fetch_masks = [&](int l) __attribute__((always_inline)) {
            if (BITS) {
                mw = (l == 0) ? mcur0 : (l == 1 ? mcur1 : mcur2);
                mw = row_ok ? mw : 0u;
            } else {
                const float* arow = acts + (long)l * acts_ls + srow * HID;
#pragma unroll
                for (int jo = 0; jo < NB; ++jo) av[jo] = rr_row_load<S>(arow, jo, q);
            }
        };
        auto save_block = [&](int l, int jo, const float (&Z)[KS]) __attribute__((always_inline)) {
            if (BITS || !dz || !row_ok) return;
            f32x4 zv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < ((jo < NB - 1) ? 4 : R); ++rr) zv[rr] = Z[4 * jo + rr];
            rr_row_store<S>(dz + (long)l * acts_ls + ((long)st * n + grow) * HID, jo, q, zv);
        };
        auto pre_tail = [&](int lp, float (&Z)[KS], int t) __attribute__((always_inline)) {
            if (t >= NT) return;
            const int jo = TB + (t >> 2), r = t & 3;
            if (BITS) Z[4 * TB + t] = rr_mask_gate<KS>(mwt, 4 * TB + t, acc[jo][r]);
            else Z[4 * TB + t] = (row_ok && avt[jo - TB][r] > 0.f) ? acc[jo][r] : 0.f;
            if (t == 3 || t == NT - 1) save_block(lp, jo, Z);
        };
        // ---- top product: dz_2 = mask_2 * (W_out^T dy), finished at once
        fetch_masks(2);
        {       // (always four k-steps: fragments and dy past the width are zero; k-step outside, nothing branches)
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
#pragma unroll
        for (int jo = 0; jo < NB; ++jo) {
#pragma unroll
            for (int r = 0; r < ((jo < NB - 1) ? 4 : R); ++r) {
                if (BITS) Za[4 * jo + r] = rr_mask_gate<KS>(mw, 4 * jo + r, acct[jo][r]);
                else Za[4 * jo + r] = (row_ok && av[jo][r] > 0.f) ? acct[jo][r] : 0.f;
            }
            save_block(2, jo, Za);
        }
        // ---- dz_1 = mask_1 * (W_2^T dz_2), dz_0 = mask_0 * (W_1^T dz_1)
        auto prod = [&](auto pc, float (&Zin)[KS], float (&Zout)[KS]) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            constexpr int lo = 2 - p;                             // the layer whose dz this product yields
            avt[0] = av[TB]; avt[1] = av[TB + 1]; mwt = mw;
            fetch_masks(lo);
            __builtin_amdgcn_sched_barrier(0);
            const int cur = wbase + lo * S::LAYER_BYTES;              // fragments of layer lo + 1 sit at index lo
            const int nxt = (lo >= 1) ? cur - S::LAYER_BYTES : wbase + S::LAYER_BYTES;
            gemm.run(acc, zero, Zin, rs, voff, cur, nxt,
                     [&](int ks) __attribute__((always_inline)) { if (p > 1) pre_tail(lo + 1, Zin, ks); },
                     [&](int jo, int r) __attribute__((always_inline)) {
                         if (BITS) Zout[4 * jo + r] = rr_mask_gate<KS>(mw, 4 * jo + r, acc[jo][r]);
                         else Zout[4 * jo + r] = (row_ok && av[jo][r] > 0.f) ? acc[jo][r] : 0.f;
                         if (r == 3) save_block(lo, jo, Zout);
                     },
                     [&]() __attribute__((always_inline)) {});
        };
        BSTAMP(sbb + 1)
        prod(std::integral_constant<int, 1>{}, Za, Zb);
        BSTAMP(sbb + 2)
        prod(std::integral_constant<int, 2>{}, Zb, Za);
        BSTAMP(sbb + 3)
        avt[0] = av[TB]; avt[1] = av[TB + 1]; mwt = mw;
        const f32x4 o = RRGemm<S>::block(w0t, Za, [&](int ks) __attribute__((always_inline)) { pre_tail(0, Za, ks); });
        BSTAMP(sbb + 4)
        if (st == 0 && !dx_stage0) continue;       // only the dz of stage 0 were wanted (uniform)
        // ---- dX (times in_isig): lane (q, row) holds input columns 4q + r of ITS row in register r — state columns go
        //      into the stage algebra (dy0 += d, dK_j += beta[st][j] h d for the earlier stages j), carried columns into
        //      dc — each (row, column) by the lane that holds it: every LDS operand requested up front with a clamped
        //      address, updated values written back under the lane's own predicate (no loop over the tile, no branch
        //      between the reads)
        {
            const float h = sH[m];
            const bool up = (gdYup || ip) && st == L.S_total - 1;       // (uniform)
            float yv0[4], dcv[4], kvv[4][CK_MAX_STAGES - 1], gup[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r, cs = min(i, ns - 1), cc = min(max(i - ns, 0), max(nc - 1, 0));
                yv0[r] = sDY0[m * CK_LD + cs];
                dcv[r] = sDC[m * CK_NC + cc];
#pragma unroll
                for (int j = 0; j < CK_MAX_STAGES - 1; ++j) kvv[r][j] = sDK[(j * TILE + m) * CK_LD + cs];
                gup[r] = up ? (ip ? sDYup[m * CK_LD + cs] : gdYup[(long)growc * ns + cs]) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                const float dxv = o[r] * x_isig[r];
                if (i < ns) {
                    float d = (up && row_ok) ? gup[r] : 0.f;
                    d += dxv;
                    sDY0[m * CK_LD + i] = yv0[r] + d;
#pragma unroll
                    for (int j = 0; j < CK_MAX_STAGES - 1; ++j) {
                        const float t = kvv[r][j] + (bn[j] * h) * d;
                        sDK[(j * TILE + m) * CK_LD + i] = (j < st && bn[j] != 0.f) ? t : kvv[r][j];
                    }
                } else if (i < ns + nc) {
                    sDC[m * CK_NC + (i - ns)] = dcv[r] + dxv;
                }
            }
        }
        BSTAMP(sbb + 5)
    }
    BSTAMP(1)
    // ---- this wave's rows of the results
    for (int idx = lane; idx < L.st_hi * 16 * ns; idx += 64) {
        const int j = idx / (16 * ns), rem = idx - j * 16 * ns;
        const int mm = 16 * half + rem / ns, c = rem % ns, row = row0 + mm;
        if (row < n) gdK[((long)j * n + row) * ns + c] = sDK[(j * TILE + mm) * CK_LD + c];
    }
    if (gdy0)
        for (int idx = lane; idx < 16 * ns; idx += 64) {
            const int mm = 16 * half + idx / ns, c = idx % ns, row = row0 + mm;
            if (row < n) gdy0[(long)row * ns + c] = sDY0[mm * CK_LD + c];
        }
    if (L.dc)
        for (int idx = lane; idx < 16 * nc; idx += 64) {
            const int mm = 16 * half + idx / nc, c = idx % nc, row = row0 + mm;
            if (row < n) L.dc[(long)row * nc + c] = sDC[mm * CK_NC + c];
        }
#undef crr_has_data
}

// ---------------------------------------------------------------------------------------------------------------------
static bool crr_enabled() {
    static const bool on = [] { const char* e = getenv("NLBAC_CONCAT_RR"); return !(e && e[0] == '0'); }();
    return on;
}
static int crr_shape_index(int hid) { return hid == 64 ? 0 : (hid == 100 ? 1 : (hid == 128 ? 2 : -1)); }
static bool crr_eligible(const nlbac_mlp& net);

bool nlbac_concat_rr_eligible(const nlbac_mlp* net) { return crr_eligible(*net); }
static bool crr_eligible(const nlbac_mlp& net) {
    return crr_enabled() && net.n_layers == 4 && crr_shape_index(net.hid) >= 0 && net.rr_kind == RR_KIND_CHAIN &&
           net.rr_fwd_off >= 0 && net.rr_bwd_off >= 0 && net.in_dim <= CRR_MAX_IN && net.out_dim <= CK_NS;
}

// waves per workgroup: four (64-row tiles) unless a tile would then straddle two problems; NLBAC_CONCAT_NW=2 keeps two
static int crr_waves(int n, int rpp) {
    static const int forced = [] { const char* e = getenv("NLBAC_CONCAT_NW"); return e ? atoi(e) : 0; }();
    if (forced == 2) return 2;
    return (rpp >= n || rpp % 64 == 0) ? 4 : 2;
}

int nlbac_concat_rr_fwd_launch(ConcatRkLaunch& L, hipStream_t s) {
    if (!crr_eligible(L.net)) return 1;
    using KernelF = void (*)(const ConcatRkLaunch);
    static const KernelF kf[2][3][2] = {{{concat_rr_fwd_kernel<4, 4, 0, 2>, concat_rr_fwd_kernel<4, 4, 1, 2>},
                                         {concat_rr_fwd_kernel<7, 1, 0, 2>, concat_rr_fwd_kernel<7, 1, 1, 2>},
                                         {concat_rr_fwd_kernel<8, 4, 0, 2>, concat_rr_fwd_kernel<8, 4, 1, 2>}},
                                        {{concat_rr_fwd_kernel<4, 4, 0, 4>, concat_rr_fwd_kernel<4, 4, 1, 4>},
                                         {concat_rr_fwd_kernel<7, 1, 0, 4>, concat_rr_fwd_kernel<7, 1, 1, 4>},
                                         {concat_rr_fwd_kernel<8, 4, 0, 4>, concat_rr_fwd_kernel<8, 4, 1, 4>}}};
    const int nw = crr_waves(L.n, L.rpp), tile = 16 * nw;
    const size_t lds = (size_t)(CK_MAX_STAGES * tile * CK_LD + tile * (CK_LD + CK_NC + 1) + 4 * 8 * 64) * sizeof(float);
    hipLaunchKernelGGL(kf[nw == 4][crr_shape_index(L.net.hid)][L.acts_bits ? 1 : 0], dim3(nlbac_ceil_div(L.n, tile)), dim3(64 * nw), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_concat_rk_fwd(rr)");
    return 0;
}

int nlbac_concat_rr_bwd_launch(ConcatRkBwdLaunch& L, hipStream_t s) {
    if (!crr_eligible(L.net)) return 1;
    using KernelB = void (*)(const ConcatRkBwdLaunch);
    static const KernelB kb[2][3][2] = {{{concat_rr_bwd_kernel<4, 4, 0, 2>, concat_rr_bwd_kernel<4, 4, 1, 2>},
                                         {concat_rr_bwd_kernel<7, 1, 0, 2>, concat_rr_bwd_kernel<7, 1, 1, 2>},
                                         {concat_rr_bwd_kernel<8, 4, 0, 2>, concat_rr_bwd_kernel<8, 4, 1, 2>}},
                                        {{concat_rr_bwd_kernel<4, 4, 0, 4>, concat_rr_bwd_kernel<4, 4, 1, 4>},
                                         {concat_rr_bwd_kernel<7, 1, 0, 4>, concat_rr_bwd_kernel<7, 1, 1, 4>},
                                         {concat_rr_bwd_kernel<8, 4, 0, 4>, concat_rr_bwd_kernel<8, 4, 1, 4>}}};
    const int nw = crr_waves(L.n, L.rpp), tile = 16 * nw;
    const size_t lds = (size_t)(CK_MAX_STAGES * tile * CK_LD + tile * (1 + CK_LD + CK_NC + 16 + CK_LD) + 4 * 8 * 64) * sizeof(float);
    hipLaunchKernelGGL(kb[nw == 4][crr_shape_index(L.net.hid)][L.acts_bits ? 1 : 0], dim3(nlbac_ceil_div(L.n, tile)), dim3(64 * nw), lds, s, L);
    NLBAC_CHECK_LAUNCH("nlbac_concat_rk_bwd(rr)");
    return 0;
}
