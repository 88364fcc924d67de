// Fused Runge-Kutta step of the control-affine NODE  dx/dt = f(x) + g(x) u :
// ONE launch evaluates all requested stages of an explicit RK step for a tile of
// 32 rows — stage algebra, f_net and g_net (run concurrently by two 4-wave groups
// of the workgroup), k = f + g u, and the step's output combinations.
//
// Replaces, per stage, the launch triple  rk_combine -> mlp_fwd[f,g] -> affine_fwd
// of the un-fused path (kept in ode_kernels.hip / mlp_kernels.hip; the backward
// still runs per stage and reads the buffers this kernel saves).  Reference call
// sites: torchdiffeq.odeint at U/sac_cbf_clf/sac_cbf_clf.py:453,577 and
// U/sac_cbf_clf/model.py:252 over NeuralODEModel.forward (model.py:208-217).
//
// CDNA4 mapping: 512-thread workgroup = 8 waves = 2 per SIMD; waves 0-3 own f_net,
// waves 4-7 own g_net, each group with its own LDS ping-pong activation tile, so the
// two nets' layer chains overlap on every SIMD (one wave's MFMAs under the other's
// LDS/L2 latency).  Weights are NOT staged through LDS: each wave owns distinct
// output columns, so a fragment-packed, L2-resident copy read straight into
// registers (1 KiB per wave-instruction, 4 chunks in flight) moves every byte once
// per wave; f_net+g_net packed are 340 KB and would not fit the 160 KB LDS anyway.
// Stage derivatives of the tile stay in LDS across stages (sK); everything the
// backward needs (Y, K, g(x), post-ReLU activations) is written once, coalesced.
#include "node_rk_shared.h"

// OCC 1: compiled for 4 waves per SIMD (128 VGPRs, a few spills) so that two workgroups share a CU and overlap their
// per-tile latency chains - pays off only for launches with well over one tile per CU (measured: 32768 rows, mask mode,
// 390 -> 329 us; 8192 rows 99 -> 104 us), so the launcher picks it by tile count.
// MODE 1: both nets <= 4 column tiles, 2: both 8, 0: mixed (see mlp_kernels.hip); BITS: save ReLU bit masks
#ifdef EXP_TIMING     // ablation build only: workgroup 0 stamps the shader clock at phase boundaries into L.err (as int64)
#define TSTAMP(slot_) if (L.err && blockIdx.x == 0 && tid == 0) reinterpret_cast<long long*>(L.err)[slot_] = (long long)__builtin_readcyclecounter();
#else
#define TSTAMP(slot_)
#endif
template <int MODE, int BITS, int OCC>
__global__ __launch_bounds__(512, OCC ? 4 : 2) void node_rk_fwd_kernel(const NodeRkLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, t = tid & 255;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);   // wave-uniform: lets L.net[grp] etc. be scalar loads
    const int lane = t & 63, wave = t >> 6;
    __shared__ unsigned s_gcnt[2];            // per-group barrier counters (GroupBar): f_net / g_net run decoupled
    if (tid < 2) s_gcnt[tid] = 0u;            // between the stage boundaries (visible after the prologue barrier)
    GroupBar gbar{&s_gcnt[grp], 0u, 4u};
#ifdef EXP_PRIO
    if (grp == 0) __builtin_amdgcn_s_setprio(EXP_PRIO);
#endif
    const int n = L.n, ns = L.n_s, nu = L.n_u, LD = L.ld;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    // device-driven chain: this tile's problem (tiles do not straddle problems there), its step slot, FSAL source
    const int p_tile = row0 / L.rpp;
    long soff = 0;
    bool fsal = false;
    if (L.ctl) {
        const double* c = L.ctl + (long)p_tile * NLBAC_DOPRI_CTL;
        if (c[C_DONE] > 0.0) return;              // (uniform) this problem's solve has finished
        const int slot = (int)c[C_NACC];
        soff = (long)slot * L.slot_floats;
        fsal = slot > 0;
    }
    float* const gK = L.K + soff;
    float* const gY = L.Y + soff;
    float* const gG = L.G + soff;
    float* const gErr = L.err ? L.err + soff : nullptr;
    const float* const gy0 = fsal ? (gY - L.slot_floats) + (long)(L.S_total - 1) * n * ns : L.y0;
    const nlbac_mlp& net = L.net[grp];
    const int hid = net.hid, NT = pad32(hid) >> 5, hidp8 = pad8(hid);
    const int nwide_own = net.n_layers - 1;
    (void)hidp8;
    const int inp = pad8(ns);
    const int n_rows = min(NLBAC_MLP_TILE, n - row0);
    const bool active = wave < NT, two = (MODE == 2) || (MODE == 0 && (wave + 4) < NT);
    WaveGemm<(MODE == 1) ? 1 : 2> wg2;
    WaveGemm<1> wg1;
    if (active) {
        if constexpr (MODE != 1) { if (two) fwd_prime<2>(wg2, net, inp, (L.stage_end - L.stage_begin > 1), wave, lane); }
        if constexpr (MODE != 2) { if (!two) fwd_prime<1>(wg1, net, inp, (L.stage_end - L.stage_begin > 1), wave, lane); }
    }

    // LDS carve (all offsets multiples of 16 B)
    float* buf = smem + grp * 2 * NLBAC_MLP_TILE * LD;           // this group's ping-pong tiles
    float* sK = smem + 4 * NLBAC_MLP_TILE * LD;                  // [stage][32][8]
    float* sY0 = sK + RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS;   // [32][8]
    float* sU = sY0 + NLBAC_MLP_TILE * RK_MAX_NS;                // [32][4]
    float* sH = sU + NLBAC_MLP_TILE * RK_MAX_NU;                 // [32] step size per row
    float* sF = sH + NLBAC_MLP_TILE;                             // [32][8]   f(x)
    float* sG = sF + NLBAC_MLP_TILE * RK_MAX_NS;                 // [32][32]  g(x)
    // output-layer weights + bias of both nets: constant over the stages, read once per launch
    float* sW = sG + NLBAC_MLP_TILE * RK_MAX_GOUT + (grp ? L.sw_off1 : 0);   // [out_dim][hid] then [out_dim]
    {
        const float* W = net.params + net.w_off[nwide_own];
        const float* bsrc = net.params + net.b_off[nwide_own];
        const int nw = net.out_dim * hid;
        for (int idx = t; idx < nw; idx += 256) sW[idx] = W[idx];
        for (int idx = t; idx < net.out_dim; idx += 256) sW[nw + idx] = bsrc[idx];
    }

    // ---- tile constants: y0, u, h, already-known stages (FSAL / f0 from an earlier launch)
    if (L.in_kind == 1 && !fsal) {
        // same arithmetic as unicycle_state_kernel (the reference takes arctan2 on the host in float64 and casts back)
        for (int idx = tid; idx < NLBAC_MLP_TILE * RK_MAX_NS; idx += 512) sY0[idx] = 0.f;
        __syncthreads();
        if (tid < NLBAC_MLP_TILE && row0 + tid < n) {
            const int row = row0 + tid, i = row % L.rpp;
            const float* o = L.in_obs + (long)i * L.in_obs_ld;
            const float th = (float)atan2((double)o[3], (double)o[2]);
            sY0[tid * RK_MAX_NS + 0] = o[0]; sY0[tid * RK_MAX_NS + 1] = o[1]; sY0[tid * RK_MAX_NS + 2] = th;
            L.y0_w[(long)row * 3 + 0] = o[0]; L.y0_w[(long)row * 3 + 1] = o[1]; L.y0_w[(long)row * 3 + 2] = th;
            if (L.in_ps && row < L.rpp) {
                L.in_ps[i * 2 + 0] = o[0] + L.in_l * cosf(th);
                L.in_ps[i * 2 + 1] = o[1] + L.in_l * sinf(th);
            }
        }
    } else
    for (int idx = tid; idx < NLBAC_MLP_TILE * RK_MAX_NS; idx += 512) {
        const int m = idx >> 3, c = idx & 7, row = row0 + m;
        sY0[idx] = (row < n && c < ns) ? gy0[(long)row * ns + c] : 0.f;
    }
    for (int idx = tid; idx < NLBAC_MLP_TILE * RK_MAX_NU; idx += 512) {
        const int m = idx >> 2, c = idx & 3, row = row0 + m;
        sU[idx] = (row < n && c < nu) ? L.u[(long)row * nu + c] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE) {
        const int p = min(row0 + tid, n - 1) / L.rpp;
        sH[tid] = L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p];
    }
    for (int idx = tid; idx < L.stage_begin * NLBAC_MLP_TILE * RK_MAX_NS; idx += 512) {
        const int j = idx / (NLBAC_MLP_TILE * RK_MAX_NS), rem = idx - j * NLBAC_MLP_TILE * RK_MAX_NS;
        const int m = rem >> 3, c = rem & 7, row = row0 + m;
        float v = 0.f;
        if (row < n && c < ns) {
            if (fsal && j == 0) {       // first stage = the previous slot's last one; kept in this slot for the interpolant
                v = (gK - L.slot_floats)[((long)(L.S_total - 1) * n + row) * ns + c];
                gK[(long)row * ns + c] = v;
            } else {
                v = gK[((long)j * n + row) * ns + c];
            }
        }
        sK[idx] = v;
    }
    __syncthreads();

    TSTAMP(0)
    for (int st = L.stage_begin; st < L.stage_end; ++st) {
        // ---- stage input  Y_st = y0 + h sum_j beta[st][j] K_j   (same op order as rk_combine_kernel): formed here for
        //      the launch's first stage only; every later stage's input is written by the threads that finish the
        //      previous stage's k (below), into both groups' tiles
        TSTAMP(1 + 8 * (st - L.stage_begin) + 0)
        float* in = buf;
        float* out = buf + NLBAC_MLP_TILE * LD;
        if (st == L.stage_begin) {
            for (int idx = t; idx < NLBAC_MLP_TILE * inp; idx += 256) {
                const int m = idx / inp, c = idx - m * inp;
                float a = 0.f;
                if (c < ns) {
                    a = sY0[m * RK_MAX_NS + c];
                    const float h = sH[m];
                    for (int j = 0; j < st; ++j)
                        if (L.beta[st][j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] * (L.beta[st][j] * h);
                    if (grp == 0 && row0 + m < n) gY[((long)st * n + row0 + m) * ns + c] = a;
                }
                in[m * LD + c] = a;
            }
            tile_sync(&gbar, lane);               // (the tile is written and read by this group only)
        }
        TSTAMP(1 + 8 * (st - L.stage_begin) + 1)

        // ---- wide layers of f_net (group 0) and g_net (group 1), each behind its own group barrier so that one
        //      group's GEMM can overlap the other's epilogue; the weight stream of each wave runs on across layers and
        //      stages (WaveGemm), only the LDS operands wait for the barriers
        {
            float* acts_tile = L.acts[grp] ? L.acts[grp] + soff + ((long)st * n + row0) * (BITS ? NT : hid) : nullptr;
            const bool wrap = st + 1 < L.stage_end;
            if constexpr (MODE == 2)
                fwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, inp, in, out, acts_tile, L.acts_ls[grp], n_rows, nwide_own, wrap, nullptr, &gbar);
            else if constexpr (MODE == 1) {
#ifdef EXP_TIMING       // per-layer stamps of wave 0 of workgroup 0, stage index 1 of the launch
                long long* dbg = (L.err && blockIdx.x == 0 && tid == 0 && st == L.stage_begin + 1)
                                     ? reinterpret_cast<long long*>(L.err) + 64 : nullptr;
                fwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, inp, in, out, acts_tile, L.acts_ls[grp], n_rows, nwide_own, wrap, dbg, &gbar);
#else
                fwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, inp, in, out, acts_tile, L.acts_ls[grp], n_rows, nwide_own, wrap, nullptr, &gbar);
#endif
            }
            else {
                if (two) fwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, inp, in, out, acts_tile, L.acts_ls[grp], n_rows, nwide_own, wrap, nullptr, &gbar);
                else fwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, inp, in, out, acts_tile, L.acts_ls[grp], n_rows, nwide_own, wrap, nullptr, &gbar);
            }
        }

        TSTAMP(1 + 8 * (st - L.stage_begin) + 2)
        // ---- skinny output layers -> sF / sG (and g(x) to global for the backward)
#ifndef EXP_NO_SKINNY
        for (int idx = t; idx < NLBAC_MLP_TILE * net.out_dim; idx += 256) {
            const int m = idx & 31, o = idx >> 5, row = row0 + m;
            const float val = skinny_row_dot(in + m * LD, sW + o * hid, hid) + sW[net.out_dim * hid + o];
            (grp == 0 ? sF + m * RK_MAX_NS : sG + m * RK_MAX_GOUT)[o] = val;
            if (grp == 1 && row < n) gG[((long)st * n + row) * (ns * nu) + o] = val;
        }
#endif
        __syncthreads();
        TSTAMP(1 + 8 * (st - L.stage_begin) + 3)

        // ---- k = f + g u   (same op order as affine_fwd_kernel), and — same thread, same (row, component) — the next
        //      stage's input Y_{st+1} = y0 + h sum_j beta[st+1][j] K_j into the first tile of BOTH groups (nothing reads
        //      the tiles between the barrier above and the one below); columns ns..inp-1 are zeroed again
        {
            const bool more = st + 1 < L.stage_end;
            float* in_f = smem;
            float* in_g = smem + 2 * NLBAC_MLP_TILE * LD;
            for (int idx = tid; idx < NLBAC_MLP_TILE * inp; idx += 512) {
                const int m = idx / inp, c = idx - m * inp;
                float y = 0.f;
                if (c < ns) {
                    float a = sF[m * RK_MAX_NS + c];
                    for (int q = 0; q < nu; ++q) a += sG[m * RK_MAX_GOUT + c * nu + q] * sU[m * RK_MAX_NU + q];
                    sK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] = a;
                    if (row0 + m < n) gK[((long)st * n + row0 + m) * ns + c] = a;
                    if (more) {
                        y = sY0[m * RK_MAX_NS + c];
                        const float h = sH[m];
                        for (int j = 0; j < st; ++j)
                            if (L.beta[st + 1][j] != 0.f) y = y + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] * (L.beta[st + 1][j] * h);
                        if (L.beta[st + 1][st] != 0.f) y = y + a * (L.beta[st + 1][st] * h);
                        if (row0 + m < n) gY[((long)(st + 1) * n + row0 + m) * ns + c] = y;
                    }
                }
                if (more) { in_f[m * LD + c] = y; in_g[m * LD + c] = y; }
            }
        }
        __syncthreads();
        TSTAMP(1 + 8 * (st - L.stage_begin) + 4)
    }

    // ---- step outputs
#ifdef EXP_TIMING
    return;
#endif
    for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += 512) {
        const int m = idx / ns, r = idx - m * ns, row = row0 + m;
        if (row >= n) continue;
        const float h = sH[m];
        if (L.out) {
            float a = sY0[m * RK_MAX_NS + r];
            for (int j = 0; j < L.n_out; ++j)
                if (L.c_out[j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r] * (L.c_out[j] * h);
            L.out[(long)row * ns + r] = a;
        }
        if (gErr) {
            float a = 0.f;
            for (int j = 0; j < L.n_err; ++j)
                if (L.c_err[j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r] * (L.c_err[j] * h);
            gErr[(long)row * ns + r] = a;
        }
    }

    // ---- fused step control: this tile's partial sums of the scaled norms (same per-entry arithmetic as
    //      dopri_norm_block), one ticket per problem; the last workgroup of a problem sums the tile partials in a fixed
    //      order and runs the controller
    if (L.norm_mode < 0) return;
    __shared__ unsigned s_last;
    if (tid < 64) {
        const int m = tid;
        float v0 = 0.f, v1 = 0.f;
        if (m < n_rows) {
            const float h = sH[m];
            for (int r = 0; r < ns; ++r) {
                const float y = sY0[m * RK_MAX_NS + r];
                if (L.norm_mode == 2) {
                    float e = 0.f, y1 = y;
                    for (int j = 0; j < L.n_err; ++j)
                        if (L.c_err[j] != 0.f) e = e + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r] * (L.c_err[j] * h);
                    const int sl = L.S_total - 1;         // y1 = the last stage's input
                    for (int j = 0; j < sl; ++j)
                        if (L.beta[sl][j] != 0.f) y1 = y1 + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r] * (L.beta[sl][j] * h);
                    const float tol = L.atol + L.rtol * fmaxf(fabsf(y), fabsf(y1));
                    const float q = e / tol;
                    v0 += q * q;
                } else {
                    const float sc = L.atol + fabsf(y) * L.rtol;
                    if (L.norm_mode == 0) {
                        const float q0 = y / sc, q1 = sK[m * RK_MAX_NS + r] / sc;
                        v0 += q0 * q0; v1 += q1 * q1;
                    } else {
                        const float q = (sK[(NLBAC_MLP_TILE + m) * RK_MAX_NS + r] - sK[m * RK_MAX_NS + r]) / sc;
                        v0 += q * q;
                    }
                }
            }
            if (L.norm_mode == 0)
                for (int c = 0; c < nu; ++c) {
                    const float y = sU[m * RK_MAX_NU + c];
                    const float q = y / (L.atol + fabsf(y) * L.rtol);
                    v0 += q * q;
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_down(v0, off, 64); v1 += __shfl_down(v1, off, 64); }
        if (tid == 0) {
            const int nblk = (L.rpp + NLBAC_MLP_TILE - 1) / NLBAC_MLP_TILE;
            const int blk = (row0 - p_tile * L.rpp) / NLBAC_MLP_TILE;
            float* q = L.partials + ((long)p_tile * nblk + blk) * 2;
            // No agent-scope fence here: on gfx950 a release at agent scope writes the XCD's whole L2 back, and this
            // workgroup has just written the step's K / Y / masks (measured: +15 us on a one-stage launch, +30 us on an
            // attempt).  The two partial sums go out as device-scope atomic exchanges — performed at the level all
            // XCDs see — and the ticket is only taken once both have RETURNED; the last workgroup reads them with
            // device-scope atomic loads.  Everything else this kernel wrote is for later launches (kernel boundary).
            const float o0 = __hip_atomic_exchange(q + 0, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float o1 = __hip_atomic_exchange(q + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(o0), "v"(o1) : "memory");
            const unsigned ticket = __hip_atomic_fetch_add(L.tickets + p_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (ticket == (unsigned)nblk - 1u) ? 1u : 0u;
            if (s_last) __hip_atomic_store(L.tickets + p_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last || tid >= 64) return;
    {
        const int nblk = (L.rpp + NLBAC_MLP_TILE - 1) / NLBAC_MLP_TILE;
        double d0 = 0.0, d1 = 0.0;
        for (int b = tid; b < nblk; b += 64) {
            const float* q = L.partials + ((long)p_tile * nblk + b) * 2;
            d0 += (double)__hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d1 += (double)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); d1 += __shfl_down(d1, off, 64); }
        if (tid == 0) {
            const double cnt = (double)L.rpp * (double)(ns + nu);
            double* c = L.ctl_w + (long)p_tile * NLBAC_DOPRI_CTL;
            const int slot_before = (int)c[C_NACC];
            const double h_try = c[C_H];
            dopri_control_vals(sqrt(d0 / cnt), sqrt(d1 / cnt), p_tile, L.norm_mode, L.t_end, L.ctl_w, L.n_slots);
            if (L.norm_mode == 2 && L.hslots && c[C_ACCEPT] > 0.0)
                L.hslots[(long)p_tile * L.n_slots + slot_before] = h_try;      // step size of the accepted step in its slot
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

// ---------------------------------------------------------------------------
// Fused backward of one RK step (discretise-then-differentiate, i.e. what autograd does through
// torchdiffeq's fixed-step / dopri5 stages): for st = st_hi-1 .. st_lo
//     dk = dK[st];  du += g(Y_st)^T dk;  df = dk, dg = dk u^T
//     dX = J_f^T df + J_g^T dg          (f_net / g_net data backward, two wave groups)
//     dY = [dYup at the last stage] + dX;  dy0 += dY;  dK[j] += beta[st][j] h dY  (j < st)
// Replaces the per-stage launch triple affine_bwd -> mlp_bwd_data[f,g] -> rk_stage_bwd.
// With dz/dG given it also leaves every stage's pre-activation grads for nlbac_mlp_bwd_weights (NODE fit).
// ---------------------------------------------------------------------------
template <int MODE, int BITS>
__global__ __launch_bounds__(512) void node_rk_bwd_kernel(const NodeRkBwdLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, t = tid & 255;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);   // wave-uniform: lets L.net[grp] etc. be scalar loads
    const int lane = t & 63, wave = t >> 6;
    __shared__ unsigned s_gcnt[2];            // per-group barrier counters, as in the forward kernel
    if (tid < 2) s_gcnt[tid] = 0u;
    GroupBar gbar{&s_gcnt[grp], 0u, 4u};
#ifdef EXP_PRIO
    if (grp == 0) __builtin_amdgcn_s_setprio(EXP_PRIO);
#endif
    const int n = L.n, ns = L.n_s, nu = L.n_u, LD = L.ld, gout = ns * nu;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    long soff = 0;
    int slot = 0;
    const bool chained = L.ctl != nullptr;
    if (chained) {
        const int p_tile = row0 / L.rpp;
        slot = (int)L.ctl[(long)p_tile * NLBAC_DOPRI_CTL + C_NACC] - L.back_idx;
        if (slot < 0) return;                       // (uniform) this problem took fewer steps
        soff = (long)slot * L.slot_floats;
    }
    const bool carry = chained && L.back_idx > 0;    // not the problem's last step: gradients arrive from the slot behind
    const float* const gG = L.G + soff;
    float* const gdG = L.dG ? L.dG + soff : nullptr;
    float* const gdK = L.dK + soff;
    float* const gdy0 = L.dy0 ? L.dy0 + soff : nullptr;
    const float* const gdYup = carry ? gdy0 + L.slot_floats : (L.dYup ? L.dYup + soff : nullptr);
    const nlbac_mlp& net = L.net[grp];
    const int hid = net.hid, NT = pad32(hid) >> 5, hidp32 = NT * 32;
    const int nwide = net.n_layers - 1;
    const int n_rows = min(NLBAC_MLP_TILE, n - row0);
    const bool keep_dz = L.dz[0] != nullptr;
    const bool active = wave < NT, two = (MODE == 2) || (MODE == 0 && (wave + 4) < NT);

    float* buf = smem + grp * 2 * NLBAC_MLP_TILE * LD;
    float* sDK = smem + 4 * NLBAC_MLP_TILE * LD;                    // [stage][32][8]
    float* sU = sDK + RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS;   // [32][4]
    float* sH = sU + NLBAC_MLP_TILE * RK_MAX_NU;                    // [32]
    float* sDY0 = sH + NLBAC_MLP_TILE;                              // [32][8]  running dy0
    float* sDU = sDY0 + NLBAC_MLP_TILE * RK_MAX_NS;                 // [32][4]  running du
    float* sDX = sDU + NLBAC_MLP_TILE * RK_MAX_NU;                  // [2][32][8]
    float* sdy_all = sDX + 2 * NLBAC_MLP_TILE * RK_MAX_NS;          // [2][32][16] output-layer grads of f / g
    float* sdy = sdy_all + grp * NLBAC_MLP_TILE * 16;
    float* sW = sdy_all + 2 * NLBAC_MLP_TILE * 16 + (grp ? L.sw_off1 : 0);   // W_last [out][hid], then W_0^T [in][hid]
    float* sW0t = sW + net.out_dim * hid;

    const int st_lo = chained ? (slot == 0 ? 0 : 1) : L.st_lo;    // (a later step's stage 0 is its predecessor's last stage)
    const bool stage0_data = L.dx_stage0 || keep_dz;
#define has_data(st_) ((st_) >= st_lo && ((st_) > 0 || stage0_data))

    WaveGemm<(MODE == 1) ? 1 : 2> wg2;
    WaveGemm<1> wg1;
    if (active && nwide >= 2 && has_data(L.st_hi - 1)) {
        const bool wrap0 = has_data(L.st_hi - 2);
        if constexpr (MODE != 1) { if (two) bwd_prime<2>(wg2, net, wave, lane, wrap0); }
        if constexpr (MODE != 2) { if (!two) bwd_prime<1>(wg1, net, wave, lane, wrap0); }
    }
    {   // constants of the launch -> LDS
        const float* Wl = net.params + net.w_off[nwide];
        const float* W0 = net.params + net.w_off[0];
        for (int idx = t; idx < net.out_dim * hid; idx += 256) sW[idx] = Wl[idx];
        for (int idx = t; idx < net.in_dim * hid; idx += 256) {
            const int i = idx / hid, k = idx - i * hid;
            sW0t[idx] = W0[(long)k * net.in_dim + i];
        }
    }
    for (int idx = tid; idx < NLBAC_MLP_TILE * RK_MAX_NU; idx += 512) {
        const int m = idx >> 2, c = idx & 3, row = row0 + m;
        const bool ok = row < n && c < nu;
        sU[idx] = ok ? L.u[(long)row * nu + c] : 0.f;
        sDU[idx] = (ok && L.du && L.du_acc) ? L.du[(long)row * nu + c] : 0.f;
    }
    for (int idx = tid; idx < NLBAC_MLP_TILE * RK_MAX_NS; idx += 512) {
        const int m = idx >> 3, c = idx & 7, row = row0 + m;
        sDY0[idx] = (row < n && c < ns && gdy0 && L.dy0_in && !carry) ? gdy0[(long)row * ns + c] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE) {
        const int p = min(row0 + tid, n - 1) / L.rpp;
        sH[tid] = chained ? (float)L.hslots[(long)p * L.n_slots + slot]
                          : (L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p]);
    }
    for (int idx = tid; idx < L.st_hi * NLBAC_MLP_TILE * RK_MAX_NS; idx += 512) {
        const int j = idx / (NLBAC_MLP_TILE * RK_MAX_NS), rem = idx - j * NLBAC_MLP_TILE * RK_MAX_NS;
        const int m = rem >> 3, c = rem & 7, row = row0 + m;
        float v = 0.f;
        if (row < n && c < ns) {
            if (!carry) v = gdK[((long)j * n + row) * ns + c];
            else if (j == L.S_total - 1) v = (gdK + L.slot_floats)[(long)row * ns + c];     // FSAL: next slot's dK[0]
        }
        sDK[idx] = v;
    }
    __syncthreads();

#if defined(EXP_TIMING) || defined(EXP_TIMING_BWD)      // stamps of workgroup 0 go to the (otherwise unused in mask mode) dz[1] pointer
#define BSTAMP(slot_) if (L.dz[1] && blockIdx.x == 0 && tid == 0) reinterpret_cast<long long*>(L.dz[1])[slot_] = (long long)__builtin_readcyclecounter();
#else
#define BSTAMP(slot_)
#endif
    for (int st = L.st_hi - 1; st >= st_lo; --st) {
        const bool data = has_data(st);
        BSTAMP(8 * st + 0)
        // ---- output-layer gradients of both nets, du
        const float* acts_tile = L.acts[grp] + soff + ((long)st * n + row0) * (BITS ? NT : hid);
        constexpr int TOP_RPT = (MODE == 1) ? 16 : 32;   // MODE 1: <= 128 padded columns -> two row groups of 16 rows
        float av_top[TOP_RPT];
        if (data) node_top_masks<TOP_RPT, BITS>(acts_tile + (long)(nwide - 1) * L.acts_ls[grp], hid, NT, t, n_rows, av_top);
        __builtin_amdgcn_sched_barrier(0);       // (issued here, not sunk to their use behind the barrier)
        for (int rem = t; rem < NLBAC_MLP_TILE * 16; rem += 256) {       // each group fills its own net's rows
            const int m = rem >> 4, o = rem & 15, row = row0 + m;
            float v = 0.f;
            if (grp == 0) {
                if (o < ns) v = sDK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + o];
            } else if (o < gout) {
                v = sDK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + o / nu] * sU[m * RK_MAX_NU + o % nu];
                if (gdG && row < n) gdG[((long)st * n + row) * gout + o] = v;
            }
            sdy[rem] = v;
        }
        if (L.du)
            for (int idx = tid; idx < NLBAC_MLP_TILE * nu; idx += 512) {
                const int m = idx / nu, c = idx - m * nu, row = min(row0 + m, n - 1);
                float a = 0.f;
                for (int r = 0; r < ns; ++r)
                    a += gG[((long)st * n + row) * gout + r * nu + c] * sDK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + r];
                sDU[m * RK_MAX_NU + c] = sDU[m * RK_MAX_NU + c] + 1.0f * a;
            }
        if (!data) continue;              // uniform: nothing below is needed for this stage
        tile_sync(&gbar, lane);
        BSTAMP(8 * st + 1)

        float* in = buf;
        float* out = buf + NLBAC_MLP_TILE * LD;
        // top (skinny) layer: thread = (row group, hidden column); narrow nets split the 32 rows over the threads that
        // would otherwise idle (hid <= 128: 2 groups of 16 rows)
        node_top_layer<TOP_RPT, BITS>(sdy, sW, net.out_dim, hid, hidp32, NT, t, n_rows, av_top, in, LD);
        tile_sync(&gbar, lane);
        BSTAMP(8 * st + 2)
        if constexpr (BITS == 0) {
            if (keep_dz) tile_to_global(in, LD, L.dz[grp] + soff + (long)(nwide - 1) * L.acts_ls[grp] + ((long)st * n + row0) * hid,
                                        hid, n_rows, t, 256);
        }

        {
            float* dz_tile = keep_dz ? L.dz[grp] + soff + ((long)st * n + row0) * hid : nullptr;
            const bool wrap = has_data(st - 1);
            if constexpr (MODE == 2)
                bwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, L.acts_ls[grp], n_rows, n_rows - 1, nwide - 1, wrap, 256, &gbar);
            else if constexpr (MODE == 1)
                bwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, L.acts_ls[grp], n_rows, n_rows - 1, nwide - 1, wrap, 256, &gbar);
            else {
                if (two) bwd_wide_layers<2, BITS>(wg2, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, L.acts_ls[grp], n_rows, n_rows - 1, nwide - 1, wrap, 256, &gbar);
                else bwd_wide_layers<1, BITS>(wg1, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, L.acts_ls[grp], n_rows, n_rows - 1, nwide - 1, wrap, 256, &gbar);
            }
        }
        BSTAMP(8 * st + 3)
        if (st == 0 && !L.dx_stage0) continue;       // only the dz of stage 0 were wanted

        // ---- dX = dz0 W_0 (one dot product per thread), then the stage algebra
        for (int idx = t; idx < NLBAC_MLP_TILE * ns; idx += 256) {
            const int m = idx & 31, i = idx >> 5;
            sDX[(grp * NLBAC_MLP_TILE + m) * RK_MAX_NS + i] = skinny_row_dot(in + m * LD, sW0t + i * hid, hid);
        }
        __syncthreads();
        BSTAMP(8 * st + 4)
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += 512) {
            const int m = idx / ns, c = idx - m * ns, row = row0 + m;
            float d = (gdYup && st == L.S_total - 1 && row < n) ? gdYup[(long)row * ns + c] : 0.f;
            d += sDX[m * RK_MAX_NS + c];
            d += sDX[(NLBAC_MLP_TILE + m) * RK_MAX_NS + c];
            sDY0[m * RK_MAX_NS + c] = sDY0[m * RK_MAX_NS + c] + d;
            const float h = sH[m];
            for (int j = 0; j < st; ++j)
                if (L.beta[st][j] != 0.f) sDK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] += (L.beta[st][j] * h) * d;
        }
        __syncthreads();
        BSTAMP(8 * st + 5)
    }
    __syncthreads();

    for (int idx = tid; idx < L.st_hi * NLBAC_MLP_TILE * ns; idx += 512) {
        const int j = idx / (NLBAC_MLP_TILE * ns), rem = idx - j * NLBAC_MLP_TILE * ns;
        const int m = rem / ns, c = rem - m * ns, row = row0 + m;
        if (row < n) gdK[((long)j * n + row) * ns + c] = sDK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c];
    }
    if (gdy0)
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += 512) {
            const int m = idx / ns, c = idx - m * ns, row = row0 + m;
            if (row < n) gdy0[(long)row * ns + c] = sDY0[m * RK_MAX_NS + c];
        }
    if (L.du)
        for (int idx = tid; idx < NLBAC_MLP_TILE * nu; idx += 512) {
            const int m = idx / nu, c = idx - m * nu, row = row0 + m;
            if (row < n) L.du[(long)row * nu + c] = sDU[m * RK_MAX_NU + c];
        }
#undef has_data
}

// A launch that differentiates stage 0 only, and that only w.r.t. the actions (Euler's backward when neither dy0 nor the
// weight gradients are wanted): du = g(Y_0)^T dK_0 per row — no net is touched, so no tile kernel either (the fused
// kernel spent 17.7 us on its prologue for it).
__global__ __launch_bounds__(256) void node_du_only_kernel(const float* __restrict__ G, const float* __restrict__ dK,
                                                           int n, int ns, int nu, float* __restrict__ du, int du_acc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * nu) return;
    const int row = i / nu, c = i - row * nu;
    float a = 0.f;
    for (int r = 0; r < ns; ++r) a += G[(long)row * ns * nu + r * nu + c] * dK[(long)row * ns + r];
    const float old = du_acc ? du[i] : 0.f;
    du[i] = old + 1.0f * a;
}

extern "C" int nlbac_node_rk_bwd(const nlbac_mlp* f, const nlbac_mlp* g, const float* u, const float* G, int P,
                                 int rows_per_problem, int n_stages_total, int st_lo, int st_hi, int dx_stage0,
                                 const float* beta, const float* h_host, const double* h_dev, int h_dev_stride,
                                 const float* acts_f, long acts_f_ls, const float* acts_g, long acts_g_ls,
                                 int acts_bits, float* dz_f, float* dz_g, float* dG, float* dK, const float* dYup,
                                 float* dy0, int dy0_in, float* du, int du_acc, const nlbac_rk_chain* chain,
                                 int back_idx, nlbac_stream_t s) {
    NLBAC_REQUIRE(f && g && u && G && acts_f && acts_g && dK, "nlbac_node_rk_bwd: null pointer");
    NLBAC_REQUIRE(!(acts_bits && dz_f), "nlbac_node_rk_bwd: weight gradients need the activations, not bit masks");
    NLBAC_REQUIRE(P >= 1 && P <= 8 && rows_per_problem >= 1, "nlbac_node_rk_bwd: bad problem sizes");
    NLBAC_REQUIRE(n_stages_total >= 1 && n_stages_total <= RK_MAX_STAGES && st_lo >= 0 && st_lo < st_hi &&
                      st_hi <= n_stages_total, "nlbac_node_rk_bwd: bad stage range");
    NLBAC_REQUIRE(f->in_dim == g->in_dim && f->in_dim <= RK_MAX_NS && f->out_dim == f->in_dim &&
                      g->out_dim % f->in_dim == 0 && g->out_dim / f->in_dim <= RK_MAX_NU && g->out_dim <= 16,
                  "nlbac_node_rk_bwd: f/g shapes are not a supported control-affine field");
    NLBAC_REQUIRE(f->hid % 4 == 0 && g->hid % 4 == 0 && f->hid <= 256 && g->hid <= 256, "nlbac_node_rk_bwd: bad hid");
#if !defined(EXP_TIMING) && !defined(EXP_TIMING_BWD)
    NLBAC_REQUIRE((dz_f == nullptr) == (dz_g == nullptr) && (dz_f == nullptr) == (dG == nullptr),
                  "nlbac_node_rk_bwd: dz_f, dz_g and dG go together");
#endif
    NLBAC_REQUIRE(h_dev || h_host || (chain && chain->hslots), "nlbac_node_rk_bwd: no step size");
    if (!(chain && chain->ctl) && st_lo == 0 && st_hi == 1 && !dx_stage0 && !dz_f) {
        if (du) {
            const int n = P * rows_per_problem, nu = g->out_dim / f->in_dim;
            hipLaunchKernelGGL(node_du_only_kernel, dim3(nlbac_ceil_div((long)n * nu, 256)), dim3(256), 0, (hipStream_t)s, G, dK, n,
                               f->in_dim, nu, du, du_acc);
            NLBAC_CHECK_LAUNCH("nlbac_node_rk_bwd(du only)");
        }
        return 0;
    }
    NodeRkBwdLaunch L;
    memset(&L, 0, sizeof(L));
    if (chain && chain->ctl) {
        NLBAC_REQUIRE(P == 1 || rows_per_problem % NLBAC_MLP_TILE == 0,
                      "nlbac_node_rk_bwd: a chained launch needs rows_per_problem %% 32 == 0 (tiles must not straddle problems)");
        NLBAC_REQUIRE(chain->hslots && chain->n_slots >= 1 && chain->slot_floats > 0 && back_idx >= 0 && dy0,
                      "nlbac_node_rk_bwd: incomplete chain description");
        L.ctl = chain->ctl; L.slot_floats = chain->slot_floats; L.back_idx = back_idx; L.n_slots = chain->n_slots;
        L.hslots = chain->hslots;
        if (chain->interp_bwd && back_idx == 0) {
            NLBAC_REQUIRE(n_stages_total == 7 && st_hi == 7, "nlbac_node_rk_bwd: interp_bwd goes with a dopri5 step");
            NLBAC_REQUIRE(chain->interp_kind == 1 ? (f->in_dim == 3 && chain->interp_dp && chain->interp_x) : (chain->interp_dout != nullptr),
                          "nlbac_node_rk_bwd: interp_bwd needs interp_dout, or out map 1 with interp_dp and interp_x (n_s == 3)");
            L.ip_on = 1; L.ip_kind = chain->interp_kind; L.ip_l = chain->interp_l; L.ip_dout = chain->interp_dout;
            L.ip_dp = chain->interp_dp; L.ip_dp2 = chain->interp_dp2; L.ip_x = chain->interp_x;
        }
    }
    L.net[0] = *f; L.net[1] = *g;
    L.u = u; L.G = G;
    L.acts[0] = acts_f; L.acts[1] = acts_g; L.acts_ls[0] = acts_f_ls; L.acts_ls[1] = acts_g_ls;
    L.acts_bits = acts_bits;
    L.dz[0] = dz_f; L.dz[1] = dz_g; L.dG = dG;
    L.dK = dK; L.dYup = dYup; L.dy0 = dy0; L.dy0_in = dy0_in; L.du = du; L.du_acc = du_acc;
    L.n = P * rows_per_problem; L.rpp = rows_per_problem;
    L.n_s = f->in_dim; L.n_u = g->out_dim / f->in_dim;
    L.S_total = n_stages_total; L.st_lo = st_lo; L.st_hi = st_hi; L.dx_stage0 = dx_stage0;
    if (beta)
        for (int i = 0; i < n_stages_total; ++i)
            for (int j = 0; j < n_stages_total; ++j) L.beta[i][j] = beta[i * n_stages_total + j];
    L.h_dev = h_dev; L.h_stride = h_dev_stride;
    for (int p = 0; p < P; ++p) L.h_val[p] = h_host ? h_host[p] : 0.f;
    {   // nets up to 128 wide run on the register-resident kernels (node_rr_kernels.hip)
        const int rr = nlbac_node_rr_bwd_launch(L, (hipStream_t)s);
        if (rr <= 0) return rr;
    }
    NLBAC_REQUIRE(!L.ip_on, "nlbac_node_rk_bwd: interp_bwd needs the register-resident kernels (nlbac_rk_interp_ok)");
    int w = ((f->hid > g->hid ? f->hid : g->hid) + 31) & ~31;
    L.ld = w + 4;
    L.sw_off1 = (((f->out_dim + f->in_dim) * f->hid) + 3) & ~3;
    const int sw_total = L.sw_off1 + ((((g->out_dim + g->in_dim) * g->hid) + 3) & ~3);
    const size_t lds = ((size_t)4 * NLBAC_MLP_TILE * L.ld + RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS +
                        NLBAC_MLP_TILE * (RK_MAX_NU + 1 + RK_MAX_NS + RK_MAX_NU + 2 * RK_MAX_NS + 2 * 16) + sw_total) *
                       sizeof(float);
    NLBAC_REQUIRE(lds <= NODE_LDS_MAX, "nlbac_node_rk_bwd: LDS budget exceeded (%zu B)", lds);
    using KernelB = void (*)(const NodeRkBwdLaunch);
    static const KernelB kb[2][3] = {{node_rk_bwd_kernel<0, 0>, node_rk_bwd_kernel<1, 0>, node_rk_bwd_kernel<2, 0>},
                                     {node_rk_bwd_kernel<0, 1>, node_rk_bwd_kernel<1, 1>, node_rk_bwd_kernel<2, 1>}};
    static bool attr_set = false;
    if (!attr_set) {
        for (int b = 0; b < 2; ++b)
            for (int m = 0; m < 3; ++m)
                (void)hipFuncSetAttribute((const void*)kb[b][m], hipFuncAttributeMaxDynamicSharedMemorySize, NODE_LDS_MAX);
        attr_set = true;
    }
    const int ntf = (f->hid + 31) >> 5, ntg = (g->hid + 31) >> 5;
    const int mode = (ntf <= 4 && ntg <= 4) ? 1 : ((ntf == 8 && ntg == 8) ? 2 : 0);
    const dim3 grid(nlbac_ceil_div(L.n, NLBAC_MLP_TILE));
    hipLaunchKernelGGL(kb[acts_bits ? 1 : 0][mode], grid, dim3(512), lds, (hipStream_t)s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_rk_bwd");
    return 0;
}

// ---------------------------------------------------------------------------
static int rk_fwd_fill(NodeRkLaunch& L, const nlbac_mlp* f, const nlbac_mlp* g, const float* y0, const float* u, int P,
                       int rows_per_problem, int stage_begin, int stage_end, int n_stages_total, const float* beta,
                       const float* c_out, int n_out, const float* c_err, int n_err, const float* h_host,
                       const double* h_dev, int h_dev_stride, float* K, float* Y, float* G, float* acts_f,
                       long acts_f_ls, float* acts_g, long acts_g_ls, int acts_bits, float* out, float* err,
                       const nlbac_rk_chain* chain, const nlbac_in_map* in_map) {
    NLBAC_REQUIRE(f && g && y0 && u && K && Y && G, "nlbac_node_rk_fwd: null pointer");
    NLBAC_REQUIRE(P >= 1 && P <= 8 && rows_per_problem >= 1, "nlbac_node_rk_fwd: bad problem sizes");
    NLBAC_REQUIRE(n_stages_total >= 1 && n_stages_total <= RK_MAX_STAGES && stage_begin >= 0 &&
                      stage_begin < stage_end && stage_end <= n_stages_total, "nlbac_node_rk_fwd: bad stage range");
    NLBAC_REQUIRE(f->in_dim == g->in_dim && f->in_dim <= RK_MAX_NS && f->out_dim == f->in_dim &&
                      g->out_dim % f->in_dim == 0 && g->out_dim / f->in_dim <= RK_MAX_NU &&
                      g->out_dim <= RK_MAX_GOUT, "nlbac_node_rk_fwd: f/g shapes are not a control-affine field");
    NLBAC_REQUIRE(f->hid % 4 == 0 && g->hid % 4 == 0 && f->hid <= 256 && g->hid <= 256, "nlbac_node_rk_fwd: bad hid");
    NLBAC_REQUIRE(h_dev || h_host, "nlbac_node_rk_fwd: no step size");
    NLBAC_REQUIRE(n_out <= n_stages_total && n_err <= n_stages_total, "nlbac_node_rk_fwd: bad coefficient counts");
    memset(&L, 0, sizeof(L));
    L.net[0] = *f; L.net[1] = *g;
    L.y0 = y0; L.u = u;
    L.in_kind = 0; L.in_obs = nullptr; L.in_obs_ld = 0; L.in_l = 0.f; L.in_ps = nullptr; L.y0_w = nullptr;
    if (in_map && in_map->kind) {
        NLBAC_REQUIRE(in_map->kind == 1 && f->in_dim == 3 && in_map->obs && in_map->obs_ld >= 4 && stage_begin == 0,
                      "nlbac_node_rk_fwd: in map 1 needs n_s == 3, observation rows and a launch that starts at stage 0");
        L.in_kind = 1; L.in_obs = in_map->obs; L.in_obs_ld = in_map->obs_ld; L.in_l = in_map->l; L.in_ps = in_map->ps;
        L.y0_w = const_cast<float*>(y0);
    }
    L.n = P * rows_per_problem; L.rpp = rows_per_problem;
    L.n_s = f->in_dim; L.n_u = g->out_dim / f->in_dim;
    L.stage_begin = stage_begin; L.stage_end = stage_end; L.S_total = n_stages_total;
    if (beta)
        for (int i = 0; i < n_stages_total; ++i)
            for (int j = 0; j < n_stages_total; ++j) L.beta[i][j] = beta[i * n_stages_total + j];
    for (int j = 0; j < n_out; ++j) L.c_out[j] = c_out[j];
    for (int j = 0; j < n_err; ++j) L.c_err[j] = c_err[j];
    L.n_out = out ? n_out : 0; L.n_err = err ? n_err : 0;
    L.h_dev = h_dev; L.h_stride = h_dev_stride;
    for (int p = 0; p < P; ++p) L.h_val[p] = h_host ? h_host[p] : 0.f;
    L.K = K; L.Y = Y; L.G = G;
    L.acts[0] = acts_f; L.acts[1] = acts_g; L.acts_ls[0] = acts_f_ls; L.acts_ls[1] = acts_g_ls;
    L.acts_bits = acts_bits;
    L.out = out; L.err = err;
    L.norm_mode = -1;
    if (chain) {
        NLBAC_REQUIRE(P == 1 || rows_per_problem % NLBAC_MLP_TILE == 0,
                      "nlbac_node_rk_fwd: a chained launch needs rows_per_problem %% 32 == 0 (tiles must not straddle problems)");
        NLBAC_REQUIRE(chain->norm_mode < 0 || (chain->norm_mode <= 2 && chain->partials && (chain->tickets || chain->norm_defer) && chain->ctl_w),
                      "nlbac_node_rk_fwd: fused step control needs partials, tickets and the control block");
        NLBAC_REQUIRE(chain->norm_mode != 2 || (err && n_err > 0), "nlbac_node_rk_fwd: norm mode 2 needs the error coefficients");
        NLBAC_REQUIRE(!chain->ctl || chain->slot_floats >= 0, "nlbac_node_rk_fwd: bad slot stride");
        L.ctl = chain->ctl; L.slot_floats = chain->slot_floats;
        L.norm_mode = chain->norm_mode; L.n_slots = chain->n_slots > 0 ? chain->n_slots : (1 << 30);
        L.rtol = chain->rtol; L.atol = chain->atol; L.t_end = chain->t_end;
        L.partials = chain->partials; L.tickets = chain->tickets; L.ctl_w = chain->ctl_w; L.hslots = chain->hslots;
        L.alog = chain->alog; L.alog_cap = chain->alog_cap;
        if (chain->norm_defer || chain->norm_pre) {
            NLBAC_REQUIRE(nlbac_node_rr_eligible(f, g),
                          "nlbac_node_rk_fwd: norm_defer / norm_pre need the register-resident kernels (nlbac_rk_interp_ok)");
            NLBAC_REQUIRE(!chain->norm_defer || (chain->norm_mode >= 0 && chain->norm_mode <= 2),
                          "nlbac_node_rk_fwd: norm_defer goes with a norm mode");
            NLBAC_REQUIRE(chain->norm_pre >= 0 && chain->norm_pre <= 2 && (!chain->norm_pre || (chain->partials_pre && chain->ctl_w)),
                          "nlbac_node_rk_fwd: norm_pre is 0, 1 or 2 and needs partials_pre and the control block");
            NLBAC_REQUIRE(!chain->norm_pre || chain->partials_pre != chain->partials,
                          "nlbac_node_rk_fwd: the partial sums read and written by one launch must be two arrays");
            L.norm_defer = chain->norm_defer ? 1 : 0; L.norm_pre = chain->norm_pre; L.partials_pre = chain->partials_pre;
        }
        if (chain->interp_out) {
            NLBAC_REQUIRE(chain->ctl && stage_end == n_stages_total && n_stages_total == 7,
                          "nlbac_node_rk_fwd: interp_out goes with an attempt launch of a device-driven dopri5 chain");
            NLBAC_REQUIRE(chain->interp_kind == 0 || (chain->interp_kind == 1 && f->in_dim == 3 && chain->interp_p),
                          "nlbac_node_rk_fwd: out map 1 needs n_s == 3 and interp_p");
            L.ip_out = chain->interp_out; L.ip_kind = chain->interp_kind; L.ip_l = chain->interp_l; L.ip_p = chain->interp_p;
        }
    }
    return 0;
}

extern "C" int nlbac_node_rk_fwd(const nlbac_mlp* f, const nlbac_mlp* g, const float* y0, const float* u, int P,
                                 int rows_per_problem, int stage_begin, int stage_end, int n_stages_total,
                                 const float* beta /* [n_stages_total][n_stages_total] row-major */,
                                 const float* c_out, int n_out, const float* c_err, int n_err,
                                 const float* h_host, const double* h_dev, int h_dev_stride, float* K, float* Y,
                                 float* G, float* acts_f, long acts_f_ls, float* acts_g, long acts_g_ls,
                                 int acts_bits, float* out, float* err, const nlbac_rk_chain* chain,
                                 const nlbac_in_map* in_map, nlbac_stream_t s) {
    NodeRkLaunch L;
    if (rk_fwd_fill(L, f, g, y0, u, P, rows_per_problem, stage_begin, stage_end, n_stages_total, beta, c_out, n_out, c_err,
                    n_err, h_host, h_dev, h_dev_stride, K, Y, G, acts_f, acts_f_ls, acts_g, acts_g_ls, acts_bits, out, err,
                    chain, in_map)) return -1;
    {   // nets up to 128 wide run on the register-resident kernels (node_rr_kernels.hip)
        const int rr = nlbac_node_rr_fwd_launch(L, (hipStream_t)s);
        if (rr <= 0) return rr;
    }
    NLBAC_REQUIRE(!L.ip_out, "nlbac_node_rk_fwd: interp_out needs the register-resident kernels (nlbac_rk_interp_ok)");
    // LDS tiles hold pad8(hid) columns (the next layer's K extent), row stride = 4 mod 8 dwords: two workgroups fit per CU
    int w = ((f->hid > g->hid ? f->hid : g->hid) + 7) & ~7;
    L.ld = w + 4;
    L.sw_off1 = ((f->out_dim * (f->hid + 1)) + 3) & ~3;
    const int sw_total = L.sw_off1 + (((g->out_dim * (g->hid + 1)) + 3) & ~3);
    const size_t lds = ((size_t)4 * NLBAC_MLP_TILE * L.ld + RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS +
                        NLBAC_MLP_TILE * (RK_MAX_NS + RK_MAX_NU + 1 + RK_MAX_NS + RK_MAX_GOUT) + sw_total) *
                       sizeof(float);
    NLBAC_REQUIRE(lds <= NODE_LDS_MAX, "nlbac_node_rk_fwd: LDS budget exceeded (%zu B)", lds);
    using KernelF = void (*)(const NodeRkLaunch);
    static const KernelF kf[2][2][3] = {
        {{node_rk_fwd_kernel<0, 0, 0>, node_rk_fwd_kernel<1, 0, 0>, node_rk_fwd_kernel<2, 0, 0>},
         {node_rk_fwd_kernel<0, 1, 0>, node_rk_fwd_kernel<1, 1, 0>, node_rk_fwd_kernel<2, 1, 0>}},
        {{node_rk_fwd_kernel<0, 0, 1>, node_rk_fwd_kernel<1, 0, 1>, node_rk_fwd_kernel<2, 0, 1>},
         {node_rk_fwd_kernel<0, 1, 1>, node_rk_fwd_kernel<1, 1, 1>, node_rk_fwd_kernel<2, 1, 1>}}};
    static bool attr_set = false;
    if (!attr_set) {
        for (int o = 0; o < 2; ++o)
            for (int b = 0; b < 2; ++b)
                for (int m = 0; m < 3; ++m)
                    (void)hipFuncSetAttribute((const void*)kf[o][b][m], hipFuncAttributeMaxDynamicSharedMemorySize, NODE_LDS_MAX);
        attr_set = true;
    }
    const int ntf = (f->hid + 31) >> 5, ntg = (g->hid + 31) >> 5;
    const int mode = (ntf <= 4 && ntg <= 4) ? 1 : ((ntf == 8 && ntg == 8) ? 2 : 0);
    const int n_tiles = nlbac_ceil_div(L.n, NLBAC_MLP_TILE);
    // two workgroups per CU when there is more than ~1.5 tiles per CU, the tiles fit twice into LDS and no
    // activations are streamed out (mask mode / no save)
    const int occ = (n_tiles > 384 && lds <= 80 * 1024 && (acts_bits || !acts_f)) ? 1 : 0;
    const dim3 grid(n_tiles);
    hipLaunchKernelGGL(kf[occ][acts_bits ? 1 : 0][mode], grid, dim3(512), lds, (hipStream_t)s, L);
    NLBAC_CHECK_LAUNCH("nlbac_node_rk_fwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The three launches that open a dopri5 solve on the device-driven chain as ONE persistent launch (node_rr_kernels.hip:
// node_rr_fwd_begin_kernel).  What the host would pass to the three nlbac_node_rk_fwd calls is derived here:
//   A  stage 0 of the step (with the in-map), norm mode 0 -> Hairer's first guess C_H0;
//   B  the probe f(y0 + h0 f0): tableau [[1]], step size C_H0, no masks kept, norm mode 1 -> the initial step C_H;
//   C  the first attempted step, stages 1..6 with step size C_H, error estimate, interpolation at t_end (chain->interp_*);
//      its norm + controller stay the separate slot-aware launch (nlbac_dopri_norm_control), as for every attempt.
// chain: the attempt's description (ctl = ctl_w = the control blocks, partials / tickets for the fused norms of A and B).
// gen: P uint32, zero before the first use; target: any value > 0 that no earlier launch on these words used + 2.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int nlbac_node_rk_fwd_begin_ok(const nlbac_mlp* f, const nlbac_mlp* g, int P, int rows_per_problem) {
    if (!f || !g || !nlbac_node_rr_eligible(f, g)) return 0;
    // every workgroup of the launch must be resident at once: one 32-row tile per CU
    return ((P == 1 || rows_per_problem % NLBAC_MLP_TILE == 0) && (long)P * rows_per_problem <= 256L * NLBAC_MLP_TILE) ? 1 : 0;
}

extern "C" int nlbac_node_rk_fwd_begin(const nlbac_mlp* f, const nlbac_mlp* g, float* y0, const float* u, int P,
                                       int rows_per_problem, const float* beta /* dopri5: [7][7] */, const float* c_err,
                                       int n_err, float* K, float* Y, float* G, float* acts_f, long acts_f_ls,
                                       float* acts_g, long acts_g_ls, float* err, const nlbac_rk_chain* chain,
                                       const nlbac_in_map* in_map, unsigned* gen, unsigned target, nlbac_stream_t s) {
    const char* who = "nlbac_node_rk_fwd_begin";
    NLBAC_REQUIRE(chain && chain->ctl && chain->ctl_w == chain->ctl && chain->partials && chain->tickets && gen && target >= 1u &&
                      target < 0xFFFFFFF0u && err && c_err && acts_f && acts_g,
                  "%s: needs the control blocks, partials, tickets, the error buffer, mask buffers and the generation words", who);
    NLBAC_REQUIRE(nlbac_node_rk_fwd_begin_ok(f, g, P, rows_per_problem), "%s: not available for these nets / sizes (nlbac_node_rk_fwd_begin_ok)", who);
    double* ctl = chain->ctl_w;
    nlbac_rk_chain ca = *chain, cb = *chain, cc = *chain;
    ca.ctl = nullptr; ca.norm_mode = 0; ca.interp_out = nullptr; ca.interp_bwd = 0;
    cb.ctl = nullptr; cb.norm_mode = 1; cb.interp_out = nullptr; cb.interp_bwd = 0;
    cc.norm_mode = -1;
    static const float probe_beta[4] = {0.f, 0.f, 1.f, 0.f};      // [[0, 0], [1, 0]]
    NodeRkLaunch LA, LB, LC;
    if (rk_fwd_fill(LA, f, g, y0, u, P, rows_per_problem, 0, 1, 7, beta, nullptr, 0, nullptr, 0, nullptr, ctl + C_H,
                    NLBAC_DOPRI_CTL, K, Y, G, acts_f, acts_f_ls, acts_g, acts_g_ls, 1, nullptr, nullptr, &ca, in_map)) return -1;
    if (rk_fwd_fill(LB, f, g, y0, u, P, rows_per_problem, 1, 2, 2, probe_beta, nullptr, 0, nullptr, 0, nullptr, ctl + C_H0,
                    NLBAC_DOPRI_CTL, K, Y, G, nullptr, acts_f_ls, nullptr, acts_g_ls, 1, nullptr, nullptr, &cb, nullptr)) return -1;
    if (rk_fwd_fill(LC, f, g, y0, u, P, rows_per_problem, 1, 7, 7, beta, nullptr, 0, c_err, n_err, nullptr, ctl + C_H,
                    NLBAC_DOPRI_CTL, K, Y, G, acts_f, acts_f_ls, acts_g, acts_g_ls, 1, nullptr, err, &cc, nullptr)) return -1;
    LA.pers_gen = gen; LA.pers_target = target;      LA.coh = 0;
    LB.pers_gen = gen; LB.pers_target = target + 1u; LB.coh = 1;
    LC.pers_gen = nullptr;                           LC.coh = 1;
    const int rr = nlbac_node_rr_fwd_begin_launch(LA, LB, LC, (hipStream_t)s);
    NLBAC_REQUIRE(rr == 0, "%s: the register-resident kernels do not take this launch", who);
    return 0;
}

bool nlbac_concat_rr_eligible(const nlbac_mlp* net);      // (concat_rr_kernels.hip)
extern "C" int nlbac_rk_interp_ok(const nlbac_mlp* f, const nlbac_mlp* g) {
    if (!f) return 0;
    return (g ? nlbac_node_rr_eligible(f, g) : nlbac_concat_rr_eligible(f)) ? 1 : 0;
}
