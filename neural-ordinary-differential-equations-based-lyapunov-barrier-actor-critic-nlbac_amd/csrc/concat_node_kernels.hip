// Fused Runge-Kutta step of the single-net NODE  dx/dt = net([x, c])  with carried inputs c = (u, t) constant over
// the step (SimulatedCars: C/sac_cbf_clf/model.py:179-205, called through torchdiffeq.odeint at
// C/sac_cbf_clf/sac_cbf_clf.py:437,458,581,603 and C/model.py:245).  ONE launch evaluates all requested stages of an
// explicit RK step for a tile of 32 rows and ONE launch differentiates them — the same scheme as node_kernels.hip for
// the control-affine field, with one net and therefore one 4-wave group per workgroup; they replace the per-stage
// launch pairs  rk_combine -> mlp_fwd  and  mlp_bwd_data -> rk_stage_bwd  (kept as the cross-check and as the path for
// nets wider than 128 units).
//
// CDNA4 mapping: 256-thread workgroup = 4 waves, one 32-column tile of the hidden layer per wave (the reference's net is
// 64 wide: two waves carry the MFMAs, all four share the VALU phases); several workgroups share a CU (LDS ~25 KB,
// < 128 VGPRs), which is what hides the per-tile layer chain here.  Weights stream from their fragment-packed,
// L2-resident copy straight into registers (WaveGemm), stage derivatives stay in LDS across stages.
#include "concat_rk_shared.h"

// NTHR = threads per workgroup.  Measured on the 64-wide reference net (two column tiles): 128-thread workgroups — only
// the two MFMA-carrying waves, twice as many workgroups per CU — are 7-12 % SLOWER than 256: the VALU phases (stage
// input, output layer, top layer, dX, stage algebra) take as long as the MFMA chain and want all four waves.
template <int NTHR>
__global__ __launch_bounds__(NTHR) void concat_rk_fwd_kernel(const ConcatRkLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = L.n, ns = L.n_s, nc = L.n_c, LD = L.ld;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
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
    float* const gErr = L.err ? L.err + soff : nullptr;
    float* const gXn = L.Xn ? L.Xn + soff : nullptr;
    const float* const gy0 = fsal ? (gY - L.slot_floats) + (long)(L.S_total - 1) * n * ns : L.y0;
    const nlbac_mlp& net = L.net;
    const int hid = net.hid, NT = pad32(hid) >> 5, nwide = net.n_layers - 1;
    const int inp = pad8(net.in_dim);
    const int n_rows = min(NLBAC_MLP_TILE, n - row0);
    const bool active = wave < NT;
    const bool multi = L.stage_end - L.stage_begin > 1;
    WaveGemm<1> wg;
    if (active) fwd_prime<1>(wg, net, inp, multi, wave, lane);

    float* buf = smem;                                              // ping-pong activation tiles
    float* sK = smem + 2 * NLBAC_MLP_TILE * LD;                     // [stage][32][CK_NS]
    float* sY0 = sK + CK_MAX_STAGES * NLBAC_MLP_TILE * CK_NS;       // [32][CK_NS]
    float* sC = sY0 + NLBAC_MLP_TILE * CK_NS;                       // [32][CK_NC]
    float* sH = sC + NLBAC_MLP_TILE * CK_NC;                        // [32]
    float* sW = sH + NLBAC_MLP_TILE;                                // output layer [out][hid], then its bias
    float* sN = sW + ((net.out_dim * (hid + 1) + 3) & ~3);          // [in_mu | in_isig | out_mu | out_sig] (norm only)
    {
        const float* W = net.params + net.w_off[nwide];
        const float* bsrc = net.params + net.b_off[nwide];
        const int nw = net.out_dim * hid;
        for (int idx = tid; idx < nw; idx += NTHR) sW[idx] = W[idx];
        for (int idx = tid; idx < net.out_dim; idx += NTHR) sW[nw + idx] = bsrc[idx];
        if (L.norm)
            for (int idx = tid; idx < 2 * net.in_dim + 2 * ns; idx += NTHR) sN[idx] = L.norm[idx];
    }
    const int idim = net.in_dim;
    for (int idx = tid; idx < NLBAC_MLP_TILE * CK_NS; idx += NTHR) {
        const int m = idx / CK_NS, c = idx - m * CK_NS, row = row0 + m;
        sY0[idx] = (row < n && c < ns) ? gy0[(long)row * ns + c] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE * CK_NC) {
        const int m = tid / CK_NC, c = tid - m * CK_NC, row = row0 + m;
        sC[tid] = (row < n && c < nc) ? L.c[(long)row * nc + c] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE) {
        const int p = min(row0 + tid, n - 1) / L.rpp;
        sH[tid] = L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p];
    }
    for (int idx = tid; idx < L.stage_begin * NLBAC_MLP_TILE * CK_NS; idx += NTHR) {      // stages of an earlier launch
        const int j = idx / (NLBAC_MLP_TILE * CK_NS), rem = idx - j * NLBAC_MLP_TILE * CK_NS;
        const int m = rem / CK_NS, c = rem - m * CK_NS, row = row0 + m;
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

    for (int st = L.stage_begin; st < L.stage_end; ++st) {
        // ---- stage input [Y_st | c],  Y_st = y0 + h sum_j beta[st][j] K_j   (same op order as rk_combine_kernel)
        float* in = buf;
        float* out = buf + NLBAC_MLP_TILE * LD;
        for (int idx = tid; idx < NLBAC_MLP_TILE * inp; idx += NTHR) {
            const int m = idx / inp, c = idx - m * inp;
            float a = 0.f;
            if (c < ns) {
                a = sY0[m * CK_NS + c];
                const float h = sH[m];
                for (int j = 0; j < st; ++j)
                    if (L.beta[st][j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * CK_NS + c] * (L.beta[st][j] * h);
                if (row0 + m < n) gY[((long)st * n + row0 + m) * ns + c] = a;
            } else if (c < ns + nc) {
                a = sC[m * CK_NC + (c - ns)];
            }
            if (L.norm && c < idim) {
                a = (a - sN[c]) * sN[idim + c];
                if (gXn && row0 + m < n) gXn[((long)st * n + row0 + m) * idim + c] = a;
            }
            in[m * LD + c] = a;
        }
        __syncthreads();
        // ---- hidden layers (MFMA), activations saved for the backward
        float* acts_tile = L.acts ? L.acts + soff + ((long)st * n + row0) * hid : nullptr;
        fwd_wide_layers<1, 0>(wg, net, active, wave, lane, LD, inp, in, out, acts_tile, L.acts_ls, n_rows, nwide,
                              st + 1 < L.stage_end);
        // ---- output layer: k_st
        for (int idx = tid; idx < NLBAC_MLP_TILE * net.out_dim; idx += NTHR) {
            const int m = idx & 31, o = idx >> 5, row = row0 + m;
            float val = skinny_row_dot(in + m * LD, sW + o * hid, hid) + sW[net.out_dim * hid + o];
            if (L.norm) val = val * sN[2 * idim + ns + o] + sN[2 * idim + o];
            sK[(st * NLBAC_MLP_TILE + m) * CK_NS + o] = val;
            if (row < n) gK[((long)st * n + row) * ns + o] = val;
        }
        __syncthreads();
    }

    // ---- step outputs
    for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += NTHR) {
        const int m = idx / ns, r = idx - m * ns, row = row0 + m;
        if (row >= n) continue;
        const float h = sH[m];
        if (L.out) {
            float a = sY0[m * CK_NS + r];
            for (int j = 0; j < L.n_out; ++j)
                if (L.c_out[j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * CK_NS + r] * (L.c_out[j] * h);
            L.out[(long)row * ns + r] = a;
        }
        if (gErr) {
            float a = 0.f;
            for (int j = 0; j < L.n_err; ++j)
                if (L.c_err[j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * CK_NS + r] * (L.c_err[j] * h);
            gErr[(long)row * ns + r] = a;
        }
    }

    // ---- fused step control (see node_rk_fwd_kernel): tile partial sums, one ticket per problem, last workgroup = controller
    if (L.norm_mode < 0) return;
    __shared__ unsigned s_last;
    if (tid < 64) {
        const int m = tid;
        float v0 = 0.f, v1 = 0.f;
        if (m < n_rows) {
            const float h = sH[m];
            for (int r = 0; r < ns; ++r) {
                const float y = sY0[m * CK_NS + r];
                if (L.norm_mode == 2) {
                    float e = 0.f, y1 = y;
                    for (int j = 0; j < L.n_err; ++j)
                        if (L.c_err[j] != 0.f) e = e + sK[(j * NLBAC_MLP_TILE + m) * CK_NS + r] * (L.c_err[j] * h);
                    const int sl = L.S_total - 1;
                    for (int j = 0; j < sl; ++j)
                        if (L.beta[sl][j] != 0.f) y1 = y1 + sK[(j * NLBAC_MLP_TILE + m) * CK_NS + r] * (L.beta[sl][j] * h);
                    const float q = e / (L.atol + L.rtol * fmaxf(fabsf(y), fabsf(y1)));
                    v0 += q * q;
                } else {
                    const float sc = L.atol + fabsf(y) * L.rtol;
                    if (L.norm_mode == 0) {
                        const float q0 = y / sc, q1 = sK[m * CK_NS + r] / sc;
                        v0 += q0 * q0; v1 += q1 * q1;
                    } else {
                        const float q = (sK[(NLBAC_MLP_TILE + m) * CK_NS + r] - sK[m * CK_NS + r]) / sc;
                        v0 += q * q;
                    }
                }
            }
            if (L.norm_mode == 0)
                for (int c = 0; c < nc; ++c) {
                    const float y = sC[m * CK_NC + c];
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

// ---------------------------------------------------------------------------
// Backward of one RK step: for st = st_hi-1 .. st_lo
//     dz = J_net^T dK[st]  (data backward of the net);  dX = dz0 W_0
//     dY = [dYup at the last stage] + dX[:, :n_s];  dy0 += dY;  dK[j] += beta[st][j] h dY;  dc += dX[:, n_s:]
// With dz given it also leaves every stage's pre-activation gradients for nlbac_mlp_bwd_weights (NODE fit).
// ---------------------------------------------------------------------------
template <int NTHR>
__global__ __launch_bounds__(NTHR) void concat_rk_bwd_kernel(const ConcatRkBwdLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = L.n, ns = L.n_s, nc = L.n_c, LD = L.ld;
    const int row0 = blockIdx.x * NLBAC_MLP_TILE;
    long soff = 0;
    int slot = 0;
    const bool chained = L.ctl != nullptr;
    if (chained) {
        slot = (int)L.ctl[(long)(row0 / L.rpp) * NLBAC_DOPRI_CTL + C_NACC] - L.back_idx;
        if (slot < 0) return;                       // (uniform) this problem took fewer steps
        soff = (long)slot * L.slot_floats;
    }
    const bool carry = chained && L.back_idx > 0;
    float* const gdK = L.dK + soff;
    float* const gdy0 = L.dy0 ? L.dy0 + soff : nullptr;
    float* const gdyn = L.dyn ? L.dyn + soff : nullptr;
    const float* const gdYup = carry ? gdy0 + L.slot_floats : (L.dYup ? L.dYup + soff : nullptr);
    const nlbac_mlp& net = L.net;
    const int hid = net.hid, NT = pad32(hid) >> 5, hidp32 = NT * 32, nwide = net.n_layers - 1;
    const int n_rows = min(NLBAC_MLP_TILE, n - row0);
    const bool keep_dz = L.dz != nullptr;
    const bool active = wave < NT;
    const long ls = L.acts_ls;

    float* buf = smem;
    float* sDK = smem + 2 * NLBAC_MLP_TILE * LD;                    // [stage][32][CK_NS]
    float* sH = sDK + CK_MAX_STAGES * NLBAC_MLP_TILE * CK_NS;       // [32]
    float* sDY0 = sH + NLBAC_MLP_TILE;                              // [32][CK_NS] running dy0
    float* sDC = sDY0 + NLBAC_MLP_TILE * CK_NS;                     // [32][CK_NC] running d carried
    float* sDX = sDC + NLBAC_MLP_TILE * CK_NC;                      // [32][CK_NS]
    float* sdy = sDX + NLBAC_MLP_TILE * CK_NS;                      // [32][16] output-layer gradient
    float* sW = sdy + NLBAC_MLP_TILE * 16;                          // W_last [out][hid], then W_0^T [in][hid]
    float* sW0t = sW + net.out_dim * hid;
    float* sN = sW + (((net.out_dim + net.in_dim) * hid + 3) & ~3);   // [in_mu | in_isig | out_mu | out_sig] (norm only)
    const int idim = net.in_dim;
    if (L.norm)
        for (int idx = tid; idx < 2 * idim + 2 * ns; idx += NTHR) sN[idx] = L.norm[idx];

    const int st_lo = chained ? (slot == 0 ? 0 : 1) : L.st_lo;
    const bool stage0_data = L.dx_stage0 || keep_dz;
#define ck_has_data(st_) ((st_) >= st_lo && ((st_) > 0 || stage0_data))
    WaveGemm<1> wg;
    if (active && nwide >= 2 && ck_has_data(L.st_hi - 1)) bwd_prime<1>(wg, net, wave, lane, ck_has_data(L.st_hi - 2));
    {
        const float* Wl = net.params + net.w_off[nwide];
        const float* W0 = net.params + net.w_off[0];
        for (int idx = tid; idx < net.out_dim * hid; idx += NTHR) sW[idx] = Wl[idx];
        for (int idx = tid; idx < net.in_dim * hid; idx += NTHR) {
            const int i = idx / hid, k = idx - i * hid;
            sW0t[idx] = W0[(long)k * net.in_dim + i];
        }
    }
    if (tid < NLBAC_MLP_TILE * CK_NC) {
        const int m = tid / CK_NC, c = tid - m * CK_NC, row = row0 + m;
        sDC[tid] = (row < n && c < nc && L.dc && L.dc_acc) ? L.dc[(long)row * nc + c] : 0.f;
    }
    for (int idx = tid; idx < NLBAC_MLP_TILE * CK_NS; idx += NTHR) {
        const int m = idx / CK_NS, c = idx - m * CK_NS, row = row0 + m;
        sDY0[idx] = (row < n && c < ns && gdy0 && L.dy0_in && !carry) ? gdy0[(long)row * ns + c] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE) {
        const int p = min(row0 + tid, n - 1) / L.rpp;
        sH[tid] = chained ? (float)L.hslots[(long)p * L.n_slots + slot]
                          : (L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p]);
    }
    for (int idx = tid; idx < L.st_hi * NLBAC_MLP_TILE * CK_NS; idx += NTHR) {
        const int j = idx / (NLBAC_MLP_TILE * CK_NS), rem = idx - j * NLBAC_MLP_TILE * CK_NS;
        const int m = rem / CK_NS, c = rem - m * CK_NS, row = row0 + m;
        float v = 0.f;
        if (row < n && c < ns) {
            if (!carry) v = gdK[((long)j * n + row) * ns + c];
            else if (j == L.S_total - 1) v = (gdK + L.slot_floats)[(long)row * ns + c];     // FSAL: next slot's dK[0]
        }
        sDK[idx] = v;
    }
    __syncthreads();

    for (int st = L.st_hi - 1; st >= st_lo; --st) {
        const bool data = ck_has_data(st);
        const float* acts_tile = L.acts + soff + ((long)st * n + row0) * hid;
        float av_top[16];            // NTHR / 2 padded columns x two row groups of 16 rows cover the tile
        if (data) node_top_masks<16, 0, NTHR>(acts_tile + (long)(nwide - 1) * ls, hid, NT, tid, n_rows, av_top);
        __builtin_amdgcn_sched_barrier(0);
        for (int rem = tid; rem < NLBAC_MLP_TILE * 16; rem += NTHR) {
            const int m = rem >> 4, o = rem & 15;
            float v = (o < ns) ? sDK[(st * NLBAC_MLP_TILE + m) * CK_NS + o] : 0.f;
            if (L.norm && o < ns) {
                v *= sN[2 * idim + ns + o];
                if (gdyn && data && row0 + m < n) gdyn[((long)st * n + row0 + m) * ns + o] = v;
            }
            sdy[rem] = v;
        }
        if (!data) continue;             // uniform
        __syncthreads();
        float* in = buf;
        float* out = buf + NLBAC_MLP_TILE * LD;
        node_top_layer<16, 0, NTHR>(sdy, sW, net.out_dim, hid, hidp32, NT, tid, n_rows, av_top, in, LD);
        __syncthreads();
        if (keep_dz) tile_to_global(in, LD, L.dz + soff + (long)(nwide - 1) * ls + ((long)st * n + row0) * hid, hid, n_rows, tid, NTHR);
        {
            float* dz_tile = keep_dz ? L.dz + soff + ((long)st * n + row0) * hid : nullptr;
            bwd_wide_layers<1, 0>(wg, net, active, wave, lane, LD, in, out, acts_tile, dz_tile, ls, n_rows, n_rows - 1,
                                  nwide - 1, ck_has_data(st - 1), NTHR);
        }
        if (st == 0 && !L.dx_stage0) continue;       // only the dz of stage 0 were wanted (uniform)

        // ---- dX = dz0 W_0: state columns -> sDX, carried columns accumulate (always the same thread per entry)
        for (int idx = tid; idx < NLBAC_MLP_TILE * net.in_dim; idx += NTHR) {
            const int m = idx & 31, i = idx >> 5;
            float v = skinny_row_dot(in + m * LD, sW0t + i * hid, hid);
            if (L.norm) v *= sN[idim + i];
            if (i < ns) sDX[m * CK_NS + i] = v;
            else sDC[m * CK_NC + (i - ns)] += v;
        }
        __syncthreads();
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += NTHR) {
            const int m = idx / ns, c = idx - m * ns, row = row0 + m;
            float d = (gdYup && st == L.S_total - 1 && row < n) ? gdYup[(long)row * ns + c] : 0.f;
            d += sDX[m * CK_NS + c];
            sDY0[m * CK_NS + c] = sDY0[m * CK_NS + c] + d;
            const float h = sH[m];
            for (int j = 0; j < st; ++j)
                if (L.beta[st][j] != 0.f) sDK[(j * NLBAC_MLP_TILE + m) * CK_NS + c] += (L.beta[st][j] * h) * d;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int idx = tid; idx < L.st_hi * NLBAC_MLP_TILE * ns; idx += NTHR) {
        const int j = idx / (NLBAC_MLP_TILE * ns), rem = idx - j * NLBAC_MLP_TILE * ns;
        const int m = rem / ns, c = rem - m * ns, row = row0 + m;
        if (row < n) gdK[((long)j * n + row) * ns + c] = sDK[(j * NLBAC_MLP_TILE + m) * CK_NS + c];
    }
    if (gdy0)
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += NTHR) {
            const int m = idx / ns, c = idx - m * ns, row = row0 + m;
            if (row < n) gdy0[(long)row * ns + c] = sDY0[m * CK_NS + c];
        }
    if (L.dc)
        for (int idx = tid; idx < NLBAC_MLP_TILE * nc; idx += NTHR) {
            const int m = idx / nc, c = idx - m * nc, row = row0 + m;
            if (row < n) L.dc[(long)row * nc + c] = sDC[m * CK_NC + c];
        }
#undef ck_has_data
}

// ---------------------------------------------------------------------------
static int concat_check(const nlbac_mlp* net, int P, int rpp, int S, const char* who) {
    NLBAC_REQUIRE(net && P >= 1 && P <= 8 && rpp >= 1, "%s: bad problem sizes", who);
    NLBAC_REQUIRE(S >= 1 && S <= CK_MAX_STAGES, "%s: bad stage count %d", who, S);
    NLBAC_REQUIRE(net->n_layers >= 2 && net->hid % 4 == 0 && net->hid <= 128, "%s: hidden width %d (fused path: <= 128)",
                  who, net->hid);
    NLBAC_REQUIRE(net->out_dim >= 1 && net->out_dim <= CK_NS && net->in_dim > net->out_dim &&
                      net->in_dim - net->out_dim <= CK_NC, "%s: net is not [x (<=16) | carried (<=4)] -> dx", who);
    return 0;
}

extern "C" int nlbac_concat_rk_fwd(const nlbac_mlp* net, const float* y0, const float* c, int P, int rows_per_problem,
                                   int stage_begin, int stage_end, int n_stages_total, const float* beta,
                                   const float* c_out, int n_out, const float* c_err, int n_err, const float* h_host,
                                   const double* h_dev, int h_dev_stride, float* K, float* Y, float* acts, long acts_ls,
                                   int acts_bits, float* out, float* err, const float* norm, float* Xn,
                                   const nlbac_rk_chain* chain, nlbac_stream_t s) {
    if (concat_check(net, P, rows_per_problem, n_stages_total, "nlbac_concat_rk_fwd")) return -1;
    NLBAC_REQUIRE(y0 && c && K && Y, "nlbac_concat_rk_fwd: null pointer");
    NLBAC_REQUIRE(stage_begin >= 0 && stage_begin < stage_end && stage_end <= n_stages_total,
                  "nlbac_concat_rk_fwd: bad stage range");
    NLBAC_REQUIRE(h_dev || h_host, "nlbac_concat_rk_fwd: no step size");
    NLBAC_REQUIRE(n_out <= n_stages_total && n_err <= n_stages_total, "nlbac_concat_rk_fwd: bad coefficient counts");
    ConcatRkLaunch L;
    memset(&L, 0, sizeof(L));
    L.net = *net;
    L.y0 = y0; L.c = c;
    L.n = P * rows_per_problem; L.rpp = rows_per_problem;
    L.n_s = net->out_dim; L.n_c = net->in_dim - net->out_dim;
    L.stage_begin = stage_begin; L.stage_end = stage_end;
    if (beta)
        for (int i = 0; i < n_stages_total; ++i)
            for (int j = 0; j < n_stages_total; ++j) L.beta[i][j] = beta[i * n_stages_total + j];
    for (int j = 0; j < n_out; ++j) L.c_out[j] = c_out[j];
    for (int j = 0; j < n_err; ++j) L.c_err[j] = c_err[j];
    L.n_out = out ? n_out : 0; L.n_err = err ? n_err : 0;
    L.h_dev = h_dev; L.h_stride = h_dev_stride;
    for (int p = 0; p < P; ++p) L.h_val[p] = h_host ? h_host[p] : 0.f;
    L.K = K; L.Y = Y; L.acts = acts; L.acts_ls = acts_ls; L.acts_bits = acts_bits;
    NLBAC_REQUIRE(!acts_bits || nlbac_concat_rr_eligible(net), "nlbac_concat_rk_fwd: mask words need the register-resident kernels (nlbac_concat_rk_mask_words)");
    L.out = out; L.err = err;
    NLBAC_REQUIRE(norm || !Xn, "nlbac_concat_rk_fwd: Xn goes with norm");
    L.norm = norm; L.Xn = Xn;
    L.S_total = n_stages_total;
    L.norm_mode = -1;
    if (chain) {
        NLBAC_REQUIRE(P == 1 || rows_per_problem % NLBAC_MLP_TILE == 0,
                      "nlbac_concat_rk_fwd: a chained launch needs rows_per_problem %% 32 == 0 (tiles must not straddle problems)");
        NLBAC_REQUIRE(chain->norm_mode < 0 || (chain->norm_mode <= 2 && chain->partials && chain->tickets && chain->ctl_w),
                      "nlbac_concat_rk_fwd: fused step control needs partials, tickets and the control block");
        NLBAC_REQUIRE(chain->norm_mode != 2 || (err && n_err > 0), "nlbac_concat_rk_fwd: norm mode 2 needs the error coefficients");
        L.ctl = chain->ctl; L.slot_floats = chain->slot_floats;
        L.norm_mode = chain->norm_mode; L.n_slots = chain->n_slots > 0 ? chain->n_slots : (1 << 30);
        L.rtol = chain->rtol; L.atol = chain->atol; L.t_end = chain->t_end;
        L.partials = chain->partials; L.tickets = chain->tickets; L.ctl_w = chain->ctl_w; L.hslots = chain->hslots;
        L.alog = chain->alog; L.alog_cap = chain->alog_cap;
        if (chain->interp_out) {
            NLBAC_REQUIRE(chain->ctl && stage_end == n_stages_total && n_stages_total == 7 && chain->interp_kind == 0,
                          "nlbac_concat_rk_fwd: interp_out goes with an attempt launch of a device-driven dopri5 chain (no out map)");
            L.ip_out = chain->interp_out;
        }
    }
    {   // the reference's depth at widths 64 / 100 / 128 runs on the register-resident kernels (concat_rr_kernels.hip)
        const int rr = nlbac_concat_rr_fwd_launch(L, (hipStream_t)s);
        if (rr <= 0) return rr;
    }
    NLBAC_REQUIRE(!L.ip_out, "nlbac_concat_rk_fwd: interp_out needs the register-resident kernels (nlbac_rk_interp_ok)");
    const int in_p = (net->in_dim + 7) & ~7, hid_p = (net->hid + 7) & ~7;
    L.ld = (hid_p > in_p ? hid_p : in_p) + 4;
    const size_t lds = ((size_t)2 * NLBAC_MLP_TILE * L.ld + CK_MAX_STAGES * NLBAC_MLP_TILE * CK_NS +
                        NLBAC_MLP_TILE * (CK_NS + CK_NC + 1) + ((net->out_dim * (net->hid + 1) + 3) & ~3) +
                        2 * (CK_NS + CK_NC) + 2 * CK_NS) * sizeof(float);
    NLBAC_REQUIRE(lds <= 64 * 1024, "nlbac_concat_rk_fwd: LDS budget exceeded (%zu B)", lds);
    hipLaunchKernelGGL(concat_rk_fwd_kernel<256>, dim3(nlbac_ceil_div(L.n, NLBAC_MLP_TILE)), dim3(256), lds, (hipStream_t)s, L);
    NLBAC_CHECK_LAUNCH("nlbac_concat_rk_fwd");
    return 0;
}

extern "C" int nlbac_concat_rk_bwd(const nlbac_mlp* net, int P, int rows_per_problem, int n_stages_total, int st_lo,
                                   int st_hi, int dx_stage0, const float* beta, const float* h_host,
                                   const double* h_dev, int h_dev_stride, const float* acts, long acts_ls, int acts_bits,
                                   float* dz, float* dK, const float* dYup, float* dy0, int dy0_in, float* dc, int dc_acc,
                                   const float* norm, float* dyn, const nlbac_rk_chain* chain, int back_idx,
                                   nlbac_stream_t s) {
    if (concat_check(net, P, rows_per_problem, n_stages_total, "nlbac_concat_rk_bwd")) return -1;
    NLBAC_REQUIRE(acts && dK, "nlbac_concat_rk_bwd: null pointer");
    NLBAC_REQUIRE(st_lo >= 0 && st_lo < st_hi && st_hi <= n_stages_total, "nlbac_concat_rk_bwd: bad stage range");
    NLBAC_REQUIRE(h_dev || h_host || (chain && chain->hslots), "nlbac_concat_rk_bwd: no step size");
    ConcatRkBwdLaunch L;
    memset(&L, 0, sizeof(L));
    if (chain && chain->ctl) {
        NLBAC_REQUIRE(P == 1 || rows_per_problem % NLBAC_MLP_TILE == 0,
                      "nlbac_concat_rk_bwd: a chained launch needs rows_per_problem %% 32 == 0");
        NLBAC_REQUIRE(chain->hslots && chain->n_slots >= 1 && chain->slot_floats > 0 && back_idx >= 0 && dy0,
                      "nlbac_concat_rk_bwd: incomplete chain description");
        L.ctl = chain->ctl; L.slot_floats = chain->slot_floats; L.back_idx = back_idx; L.n_slots = chain->n_slots;
        L.hslots = chain->hslots;
        if (chain->interp_bwd && back_idx == 0) {
            NLBAC_REQUIRE(n_stages_total == 7 && st_hi == 7 && chain->interp_kind == 0 && chain->interp_dout,
                          "nlbac_concat_rk_bwd: interp_bwd goes with a dopri5 step and interp_dout (no out map)");
            L.ip_on = 1; L.ip_dout = chain->interp_dout;
        }
    }
    L.net = *net;
    L.acts = acts; L.acts_ls = acts_ls; L.acts_bits = acts_bits; L.dz = dz;
    NLBAC_REQUIRE(!acts_bits || (nlbac_concat_rr_eligible(net) && !dz), "nlbac_concat_rk_bwd: mask words need the register-resident kernels and exclude dz");
    L.dK = dK; L.dYup = dYup; L.dy0 = dy0; L.dy0_in = dy0_in; L.dc = dc; L.dc_acc = dc_acc;
    L.n = P * rows_per_problem; L.rpp = rows_per_problem;
    L.n_s = net->out_dim; L.n_c = net->in_dim - net->out_dim;
    L.S_total = n_stages_total; L.st_lo = st_lo; L.st_hi = st_hi; L.dx_stage0 = dx_stage0;
    if (beta)
        for (int i = 0; i < n_stages_total; ++i)
            for (int j = 0; j < n_stages_total; ++j) L.beta[i][j] = beta[i * n_stages_total + j];
    L.h_dev = h_dev; L.h_stride = h_dev_stride;
    for (int p = 0; p < P; ++p) L.h_val[p] = h_host ? h_host[p] : 0.f;
    #ifndef RR_TIMING      /* (the timing build takes its stamps in dyn) */
    NLBAC_REQUIRE(norm || !dyn, "nlbac_concat_rk_bwd: dyn goes with norm");
#endif
    NLBAC_REQUIRE(!norm || !dz || dyn, "nlbac_concat_rk_bwd: weight gradients of a normalised field need dyn");
    L.norm = norm; L.dyn = dyn;
    {
        const int rr = nlbac_concat_rr_bwd_launch(L, (hipStream_t)s);
        if (rr <= 0) return rr;
    }
    NLBAC_REQUIRE(!L.ip_on, "nlbac_concat_rk_bwd: interp_bwd needs the register-resident kernels (nlbac_rk_interp_ok)");
    L.ld = ((net->hid + 31) & ~31) + 4;
    const size_t lds = ((size_t)2 * NLBAC_MLP_TILE * L.ld + CK_MAX_STAGES * NLBAC_MLP_TILE * CK_NS +
                        NLBAC_MLP_TILE * (1 + CK_NS + CK_NC + CK_NS + 16) +
                        (((net->out_dim + net->in_dim) * net->hid + 3) & ~3) + 2 * (CK_NS + CK_NC) + 2 * CK_NS) * sizeof(float);
    NLBAC_REQUIRE(lds <= 64 * 1024, "nlbac_concat_rk_bwd: LDS budget exceeded (%zu B)", lds);
    hipLaunchKernelGGL(concat_rk_bwd_kernel<256>, dim3(nlbac_ceil_div(L.n, NLBAC_MLP_TILE)), dim3(256), lds, (hipStream_t)s, L);
    NLBAC_CHECK_LAUNCH("nlbac_concat_rk_bwd");
    return 0;
}

extern "C" int nlbac_concat_rk_mask_words(const nlbac_mlp* net) {
    // uint32 words per row and layer of the ReLU masks nlbac_concat_rk_fwd can write instead of the activations
    // (acts_bits): 4 when the net runs on the register-resident kernels, 0 = not available (activations only)
    return (net && nlbac_concat_rr_eligible(net)) ? 4 : 0;
}
