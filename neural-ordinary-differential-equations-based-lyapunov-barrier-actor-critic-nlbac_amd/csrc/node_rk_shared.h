// What the two families of fused Runge-Kutta step kernels of the control-affine NODE  dx/dt = f(x) + g(x) u  share:
// the launch descriptors and the per-tile bookkeeping around the layer chains (tile constants, stage algebra, step
// outputs, the in-launch step control).  The families differ in how a 32-row tile's layer chains run:
//   node_kernels.hip     LDS-tiled: 8 waves, activations in LDS ping-pong tiles, any width up to 256
//   node_rr_kernels.hip  register-resident: 4 waves, one (net, 16 rows) chain per wave, widths <= 128
// Reference: torchdiffeq.odeint at U/sac_cbf_clf/sac_cbf_clf.py:453,577 and U/sac_cbf_clf/model.py:252 over
// NeuralODEModel.forward (model.py:208-217).
#pragma once
#include "mlp_device.h"
#include "ode_control.h"

#define NODE_LDS_MAX (160 * 1024 - 64)   /* dynamic LDS per workgroup; the rest holds the static group-barrier counters */
#define RK_MAX_STAGES 8
#define RK_MAX_NS 8
#define RK_MAX_NU 4
#define RK_MAX_GOUT (RK_MAX_NS * RK_MAX_NU)

struct NodeRkLaunch {
    nlbac_mlp net[2];                 // f, g
    const float* y0; const float* u;
    int n, rpp, n_s, n_u;
    int stage_begin, stage_end, S_total;
    float beta[RK_MAX_STAGES][RK_MAX_STAGES];
    float c_out[RK_MAX_STAGES]; int n_out;
    float c_err[RK_MAX_STAGES]; int n_err;
    const double* h_dev; int h_stride; float h_val[8];
    float* K; float* Y; float* G;
    float* acts[2]; long acts_ls[2];
    int acts_bits;                    // acts hold bit-packed ReLU masks [layer][stage*n + row][words per row] (uint32)
    float* out; float* err;
    int ld;
    int sw_off1;                      // float offset of g_net's output-layer block behind f_net's in LDS
    // device-driven dopri5 chain (nlbac_rk_chain): problems whose solve is done are skipped; the step's buffers (K, Y,
    // G, acts, err) are those of step slot C_NACC, `slot_floats` floats apart; a slot > 0 starts from its predecessor's
    // last stage (y1 = Y[6], FSAL K[6]).  norm_mode >= 0: the scaled norms of nlbac_dopri_norm_control and the step
    // controller run in this launch's epilogue (last workgroup of each problem).
    const double* ctl; long slot_floats;
    int norm_mode, n_slots; float rtol, atol; double t_end;
    float* partials; unsigned* tickets; double* ctl_w; double* hslots;
    double* alog; int alog_cap;       // attempt log [P][alog_cap][3] = (h tried, error ratio, accepted) or null
    // nlbac_in_map: the solve's initial state is formed by this launch (stage 0 of a fresh step) from observation rows
    // and written to y0 for the launches that follow; kind 1 = the Unicycle tasks' state (+ its look-ahead point)
    int in_kind; const float* in_obs; int in_obs_ld; float in_l; float* in_ps; float* y0_w;
    // nlbac_rk_chain::interp_*: an attempt whose step reaches t_end also evaluates the solve's result — the interpolant
    // at t_end, and the owner's out-map of it — for its rows (register-resident kernels; what nlbac_dopri_interp_fwd
    // does as a launch of its own after the accept decision; a rejected attempt's values are overwritten by the next)
    float* ip_out; int ip_kind; float ip_l; float* ip_p;
    // nlbac_node_rk_fwd_begin (the three launches that open a dopri5 solve as ONE persistent launch, register-resident
    // kernels): pers_gen [P] — the workgroup that runs a fused controller publishes the control block at device scope and
    // stores pers_target there, the other workgroups of the problem wait for it before their next phase; coh: this phase
    // reads the control block (h, slot, done) past the non-coherent caches (another workgroup of THIS launch wrote it)
    unsigned* pers_gen; unsigned pers_target; int coh;
    // nlbac_rk_chain::norm_defer / norm_pre: the fused norm WITHOUT its election.  norm_defer: the epilogue only leaves
    // this tile's partial sums (plain stores); norm_pre = 1 + mode: every workgroup of THIS launch sums the partials the
    // previous launch left (partials_pre, the same fixed order) and runs that mode's controller itself before its first
    // stage — the step size it yields is this launch's; the problem's first tile writes the control block.
    int norm_defer, norm_pre; const float* partials_pre;
};

struct NodeRkBwdLaunch {
    nlbac_mlp net[2];
    const float* u; const float* G;
    const float* acts[2]; long acts_ls[2];
    int acts_bits;                    // acts hold bit-packed ReLU masks (see NodeRkLaunch)
    float* dz[2]; float* dG;
    float* dK; const float* dYup;
    float* dy0; int dy0_in;
    float* du; int du_acc;
    int n, rpp, n_s, n_u, S_total, st_lo, st_hi, dx_stage0;
    float beta[RK_MAX_STAGES][RK_MAX_STAGES];
    const double* h_dev; int h_stride; float h_val[8];
    int ld, sw_off1;
    // device-driven chain: launch `back_idx` differentiates step slot C_NACC - back_idx of each problem (problems with
    // fewer accepted steps are skipped); back_idx 0 is every problem's LAST step (dK / dy0 / dYup come from the
    // interpolant's backward), the others start from the slot behind them: dK[6] = its dK[0] (FSAL), dYup = its dy0.
    // Step sizes come from hslots[p][slot].
    const double* ctl; long slot_floats; int back_idx, n_slots; const double* hslots;
    // nlbac_rk_chain::interp_*: launch back_idx 0 forms dK / dy0 / dy1 of each problem's last step itself, from
    // d loss / d y(t_end) (ip_dout) or, with the out-map, from d loss / d p (ip_dp [+ ip_dp2]) and the solve's output
    // ip_x — what nlbac_dopri_interp_bwd does as a launch of its own (register-resident kernels)
    int ip_on, ip_kind; float ip_l; const float *ip_dout, *ip_dp, *ip_dp2, *ip_x;
};

// The register-resident kernels (node_rr_kernels.hip): 0 = launched, 1 = these nets / this launch are not theirs (the
// LDS-tiled kernels take it), < 0 = error.
int nlbac_node_rr_fwd_launch(NodeRkLaunch& L, hipStream_t s);
int nlbac_node_rr_bwd_launch(NodeRkBwdLaunch& L, hipStream_t s);
int nlbac_node_rr_fwd_begin_launch(NodeRkLaunch& LA, NodeRkLaunch& LB, NodeRkLaunch& LC, hipStream_t s);
bool nlbac_node_rr_eligible(const nlbac_mlp* f, const nlbac_mlp* g);

// ---------------------------------------------------------------------------------------------------------------
// forward: the small per-tile arrays in LDS
// ---------------------------------------------------------------------------------------------------------------
struct RkFwdTile {
    float* sK;     // [stage][32][8]   stage derivatives
    float* sY0;    // [32][8]          the step's initial state
    float* sU;     // [32][4]          actions
    float* sH;     // [32]             step size per row
    float* sF;     // [32][8]          f(x) of the current stage
    float* sG;     // [32][32]         g(x) of the current stage
    __host__ __device__ __forceinline__ static constexpr int floats() {
        return RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS + NLBAC_MLP_TILE * (RK_MAX_NS + RK_MAX_NU + 1 + RK_MAX_NS + RK_MAX_GOUT);
    }
    __device__ __forceinline__ void carve(float* p) {
        sK = p;
        sY0 = sK + RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS;
        sU = sY0 + NLBAC_MLP_TILE * RK_MAX_NS;
        sH = sU + NLBAC_MLP_TILE * RK_MAX_NU;
        sF = sH + NLBAC_MLP_TILE;
        sG = sF + NLBAC_MLP_TILE * RK_MAX_NS;
    }
};

// where a tile's step lives: its problem, step slot and FSAL source (device-driven chain) — false: the problem is done
struct RkFwdWhere {
    int p_tile; long soff; bool fsal;
    float *gK, *gY, *gG, *gErr; const float* gy0;
    bool ip; float ip_x;         // this attempt reaches t_end: evaluate the interpolant at abscissa ip_x (nlbac_rk_chain::interp_out)
    double h_pre;                // norm_pre: the step size this launch's own controller produced (rk_fwd_norm_pre)
};
__device__ __forceinline__ double rk_ctl_load(const double* p, bool coh) {
    return coh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ bool rk_fwd_where(const NodeRkLaunch& L, int row0, RkFwdWhere& w) {
    w.p_tile = row0 / L.rpp;
    w.soff = 0;
    w.fsal = false;
    w.ip = false;
    w.ip_x = 0.f;
    w.h_pre = 0.0;
    if (L.ctl) {
        const double* c = L.ctl + (long)w.p_tile * NLBAC_DOPRI_CTL;
        const bool coh = L.coh != 0;
        if (rk_ctl_load(c + C_DONE, coh) > 0.0) return false;              // (uniform) this problem's solve has finished
        const int slot = (int)rk_ctl_load(c + C_NACC, coh);
        w.soff = (long)slot * L.slot_floats;
        w.fsal = slot > 0;
        if (L.ip_out && !L.norm_pre) {      // the controller's own test and abscissa (ode_control.h: accept && t + h >= t_end -> C_X)
            const double t = rk_ctl_load(c + C_T, coh), h = rk_ctl_load(c + C_H, coh);
            w.ip = t + h >= L.t_end;
            w.ip_x = (float)((L.t_end - t) / h);
        }
    }
    w.gK = L.K + w.soff;
    w.gY = L.Y + w.soff;
    w.gG = L.G + w.soff;
    w.gErr = L.err ? L.err + w.soff : nullptr;
    w.gy0 = w.fsal ? (w.gY - L.slot_floats) + (long)(L.S_total - 1) * L.n * L.n_s : L.y0;
    return true;
}

// nlbac_rk_chain::norm_pre: the controller of the norm whose tile partials the previous launch left, run by every
// workgroup for its own problem: the sums in the order the elected workgroup of the fused form takes them (same bits),
// dopri_control_vals on a private copy of the block; the problem's first tile runs it on the block itself, for the
// launches that follow.  What a launch saves this way is the previous launch's election: two device-scope atomic round
// trips per workgroup behind its stores, the last workgroup's sums and controller with the whole chip idle, ~10 us per
// one-stage launch; what it pays is this reduction (L2-resident loads, one fp64 pow) under its own prologue's loads.
// Contains a barrier.
template <int NTHR>
__device__ __forceinline__ void rk_fwd_norm_pre(const NodeRkLaunch& L, RkFwdWhere& w, int row0, int tid) {
    __shared__ double s_hpre_[2];
    const int mode = L.norm_pre - 1, p = w.p_tile;
    if (tid < 64) {
        const int nblk = (L.rpp + NLBAC_MLP_TILE - 1) / NLBAC_MLP_TILE;
        double d0 = 0.0, d1 = 0.0;
        for (int b = tid; b < nblk; b += 64) {
            const float* q = L.partials_pre + ((long)p * nblk + b) * 2;
            d0 += (double)q[0];
            d1 += (double)q[1];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_down(d0, off, 64); d1 += __shfl_down(d1, off, 64); }
        if (tid == 0) {
            const double cnt = (double)L.rpp * (double)(L.n_s + L.n_u);
            const double n0 = sqrt(d0 / cnt), n1 = sqrt(d1 / cnt);
            double* c = L.ctl_w + (long)p * NLBAC_DOPRI_CTL;
            double loc[NLBAC_DOPRI_CTL];
#pragma unroll
            for (int k = 0; k < NLBAC_DOPRI_CTL; ++k) loc[k] = 0.0;
            if (mode == 1) { loc[C_H0] = c[C_H0]; loc[C_D1] = c[C_D1]; }     // (what mode 1 reads: the previous launch's first tile wrote them)
            dopri_control_vals(n0, n1, 0, mode, L.t_end, loc, L.n_slots);
            if (row0 == p * L.rpp) dopri_control_vals(n0, n1, p, mode, L.t_end, L.ctl_w, L.n_slots);
            s_hpre_[0] = mode == 0 ? loc[C_H0] : loc[C_H];
            s_hpre_[1] = mode == 0 ? 0.0 : c[C_T];
        }
    }
    __syncthreads();
    w.h_pre = s_hpre_[0];
    if (L.ip_out) {
        const double t = s_hpre_[1], h = w.h_pre;
        w.ip = t + h >= L.t_end;
        w.ip_x = (float)((L.t_end - t) / h);
    }
}

// tile constants: y0 (or the in-map's state), u, h, already-known stages (FSAL / f0 from an earlier launch).
// Contains barriers; ends with one.
template <int NTHR>
__device__ __forceinline__ void rk_fwd_tile_constants(const NodeRkLaunch& L, RkFwdWhere& w, const RkFwdTile& T,
                                                      int row0, int tid) {
    const int n = L.n, ns = L.n_s, nu = L.n_u;
    if (L.in_kind == 1 && !w.fsal) {
        // same arithmetic as unicycle_state_kernel (the reference takes arctan2 on the host in float64 and casts back)
        for (int idx = tid; idx < NLBAC_MLP_TILE * RK_MAX_NS; idx += NTHR) T.sY0[idx] = 0.f;
        __syncthreads();
        if (tid < NLBAC_MLP_TILE && row0 + tid < n) {
            const int row = row0 + tid, i = row % L.rpp;
            const float* o = L.in_obs + (long)i * L.in_obs_ld;
            const float th = (float)atan2((double)o[3], (double)o[2]);
            T.sY0[tid * RK_MAX_NS + 0] = o[0]; T.sY0[tid * RK_MAX_NS + 1] = o[1]; T.sY0[tid * RK_MAX_NS + 2] = th;
            L.y0_w[(long)row * 3 + 0] = o[0]; L.y0_w[(long)row * 3 + 1] = o[1]; L.y0_w[(long)row * 3 + 2] = th;
            if (L.in_ps && row < L.rpp) {
                L.in_ps[i * 2 + 0] = o[0] + L.in_l * cosf(th);
                L.in_ps[i * 2 + 1] = o[1] + L.in_l * sinf(th);
            }
        }
    }
    // Every global load below is issued before the first LDS store that consumes one (clamped addresses, selects
    // afterwards): written as "sX[idx] = cond ? load : 0" statement by statement — and as a loop with a run-time bound for
    // the earlier stages — each was a global round trip of its own, several microseconds of a launch's prologue in a row.
    {
        constexpr int NY = (NLBAC_MLP_TILE * RK_MAX_NS + NTHR - 1) / NTHR, NU = (NLBAC_MLP_TILE * RK_MAX_NU + NTHR - 1) / NTHR;
        constexpr int NK = (RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS + NTHR - 1) / NTHR;
        const bool plain_y0 = !(L.in_kind == 1 && !w.fsal);
        float vy[NY], vu[NU], vk[NK], vh = 0.f;
#pragma unroll
        for (int it = 0; it < NY; ++it) {
            const int idx = tid + NTHR * it, m = (idx >> 3) & (NLBAC_MLP_TILE - 1), c = idx & 7;
            vy[it] = plain_y0 ? w.gy0[(long)min(row0 + m, n - 1) * ns + min(c, ns - 1)] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NU; ++it) {
            const int idx = tid + NTHR * it, m = (idx >> 2) & (NLBAC_MLP_TILE - 1), c = idx & 3;
            vu[it] = L.u[(long)min(row0 + m, n - 1) * nu + min(c, nu - 1)];
        }
        if (tid < NLBAC_MLP_TILE) {
            const int p = min(row0 + tid, n - 1) / L.rpp;
            vh = L.h_dev ? (float)rk_ctl_load(L.h_dev + (long)p * L.h_stride, L.coh != 0) : L.h_val[p];
        }
#pragma unroll
        for (int it = 0; it < NK; ++it) {
            const int idx = tid + NTHR * it;
            const int j = idx / (NLBAC_MLP_TILE * RK_MAX_NS), rem = idx - j * NLBAC_MLP_TILE * RK_MAX_NS;
            const int m = rem >> 3, c = rem & 7;
            const long rc = (long)min(row0 + m, n - 1) * ns + min(c, ns - 1);
            float v = 0.f;
            if (j < L.stage_begin) {          // (uniform per wave: a stage's block is a multiple of 64 entries)
                if (w.fsal && j == 0) v = (w.gK - L.slot_floats)[(long)(L.S_total - 1) * n * ns + rc];   // first stage = the previous slot's last
                else v = w.gK[(long)j * n * ns + rc];
            }
            vk[it] = v;
        }
        // nlbac_rk_chain::norm_pre: the step size of this launch comes from the previous launch's tile partials — summed
        // HERE, behind the row loads above (their trips to memory overlap; in front of them the two latencies added up)
        if (L.norm_pre) {
            rk_fwd_norm_pre<NTHR>(L, w, row0, tid);
            vh = (float)w.h_pre;
        }
#pragma unroll
        for (int it = 0; it < NY; ++it) {
            const int idx = tid + NTHR * it, m = idx >> 3, c = idx & 7;
            if (plain_y0 && idx < NLBAC_MLP_TILE * RK_MAX_NS) T.sY0[idx] = (row0 + m < n && c < ns) ? vy[it] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NU; ++it) {
            const int idx = tid + NTHR * it, m = idx >> 2, c = idx & 3;
            if (idx < NLBAC_MLP_TILE * RK_MAX_NU) T.sU[idx] = (row0 + m < n && c < nu) ? vu[it] : 0.f;
        }
        if (tid < NLBAC_MLP_TILE) T.sH[tid] = vh;
#pragma unroll
        for (int it = 0; it < NK; ++it) {
            const int idx = tid + NTHR * it;
            const int j = idx / (NLBAC_MLP_TILE * RK_MAX_NS), rem = idx - j * NLBAC_MLP_TILE * RK_MAX_NS;
            const int m = rem >> 3, c = rem & 7, row = row0 + m;
            if (j < L.stage_begin) {
                const float v = (row < n && c < ns) ? vk[it] : 0.f;
                T.sK[idx] = v;
                if (w.fsal && j == 0 && row < n && c < ns) w.gK[(long)row * ns + c] = v;      // kept in this slot for the interpolant
            }
        }
    }
    __syncthreads();
}

// stage input  Y_st = y0 + h sum_j beta[st][j] K_j  of the launch's FIRST stage (same op order as rk_combine_kernel),
// for the threads [t, t + nthr, ...) of one consumer: written to `in` (row stride LD, columns ns..inp-1 zero) and, by
// the consumer that `owns` the global copy, to Y
__device__ __forceinline__ void rk_fwd_first_input(const NodeRkLaunch& L, const RkFwdWhere& w, const RkFwdTile& T,
                                                   int row0, int st, int inp, float* in, int LD, bool owns, int t,
                                                   int nthr) {
    const int n = L.n, ns = L.n_s;
    for (int idx = t; idx < NLBAC_MLP_TILE * inp; idx += nthr) {
        const int m = idx / inp, c = idx - m * inp;
        float a = 0.f;
        if (c < ns) {
            a = T.sY0[m * RK_MAX_NS + c];
            const float h = T.sH[m];
            for (int j = 0; j < st; ++j)
                if (L.beta[st][j] != 0.f) a = a + T.sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] * (L.beta[st][j] * h);
            if (owns && row0 + m < n) w.gY[((long)st * n + row0 + m) * ns + c] = a;
        }
        in[m * LD + c] = a;
    }
}

// k = f + g u   (same op order as affine_fwd_kernel), and — same thread, same (row, component) — the next stage's input
// Y_{st+1} = y0 + h sum_j beta[st+1][j] K_j into in_f (and in_g when given; row stride LD, columns ns..inp-1 zeroed
// again).  Call between two workgroup barriers: reads sF / sG of the stage, nothing else touches the input tiles.
template <int NTHR>
__device__ __forceinline__ void rk_fwd_combine(const NodeRkLaunch& L, const RkFwdWhere& w, const RkFwdTile& T, int row0,
                                               int st, int inp, float* in_f, float* in_g, int LD, int tid) {
    const int n = L.n, ns = L.n_s, nu = L.n_u;
    const bool more = st + 1 < L.stage_end;
    for (int idx = tid; idx < NLBAC_MLP_TILE * inp; idx += NTHR) {
        const int m = idx / inp, c = idx - m * inp;
        float y = 0.f;
        if (c < ns) {
            float a = T.sF[m * RK_MAX_NS + c];
            for (int q = 0; q < nu; ++q) a += T.sG[m * RK_MAX_GOUT + c * nu + q] * T.sU[m * RK_MAX_NU + q];
            T.sK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] = a;
            if (row0 + m < n) w.gK[((long)st * n + row0 + m) * ns + c] = a;
            if (more) {
                y = T.sY0[m * RK_MAX_NS + c];
                const float h = T.sH[m];
                for (int j = 0; j < st; ++j)
                    if (L.beta[st + 1][j] != 0.f) y = y + T.sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] * (L.beta[st + 1][j] * h);
                if (L.beta[st + 1][st] != 0.f) y = y + a * (L.beta[st + 1][st] * h);
                if (row0 + m < n) w.gY[((long)(st + 1) * n + row0 + m) * ns + c] = y;
            }
        }
        if (more) { in_f[m * LD + c] = y; if (in_g) in_g[m * LD + c] = y; }
    }
}

// step outputs (solution / error combination) and, with norm_mode >= 0, the fused step control: this tile's partial
// sums of the scaled norms (same per-entry arithmetic as dopri_norm_block), one ticket per problem; the last workgroup
// of a problem sums the tile partials in a fixed order and runs the controller.  Contains a barrier when norm_mode >= 0
// (every thread of the workgroup must call it).
template <int NTHR>
__device__ __forceinline__ void rk_fwd_outputs_and_control(const NodeRkLaunch& L, const RkFwdWhere& w, const RkFwdTile& T,
                                                           int row0, int n_rows, int tid) {
    const int n = L.n, ns = L.n_s, nu = L.n_u, p_tile = w.p_tile;
    const float *sK = T.sK, *sY0 = T.sY0, *sU = T.sU, *sH = T.sH;
    for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += NTHR) {
        const int m = idx / ns, r = idx - m * ns, row = row0 + m;
        if (row >= n) continue;
        const float h = sH[m];
        if (L.out) {
            float a = sY0[m * RK_MAX_NS + r];
            for (int j = 0; j < L.n_out; ++j)
                if (L.c_out[j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r] * (L.c_out[j] * h);
            L.out[(long)row * ns + r] = a;
        }
        if (w.gErr) {
            float a = 0.f;
            for (int j = 0; j < L.n_err; ++j)
                if (L.c_err[j] != 0.f) a = a + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r] * (L.c_err[j] * h);
            w.gErr[(long)row * ns + r] = a;
            if (L.norm_mode == 2) {
                // the error norm's term of this (row, component) — one per thread here; the row sums follow below (one
                // thread per row doing all of its components' 13-term chains in a row cost the attempt launch 4 us)
                const float y = sY0[m * RK_MAX_NS + r];
                float y1 = y;
                const int sl = L.S_total - 1;         // y1 = the last stage's input
                for (int j = 0; j < sl; ++j)
                    if (L.beta[sl][j] != 0.f) y1 = y1 + sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r] * (L.beta[sl][j] * h);
                const float tol = L.atol + L.rtol * fmaxf(fabsf(y), fabsf(y1));
                const float q = a / tol;
                T.sG[m * RK_MAX_NS + r] = q * q;      // (g(x) of the last stage is no longer needed)
            }
        }
    }
    if (w.ip) {      // (uniform per tile) the solve's result, should this attempt be accepted: one (row, component) per thread
        static_assert(NTHR >= NLBAC_MLP_TILE * RK_MAX_NS, "one (row, component) per thread");
        const int m = tid >> 3, r = tid & 7, row = row0 + m, sl = L.S_total - 1;
        if (tid < NLBAC_MLP_TILE * RK_MAX_NS && r < ns) {
            const float h = sH[m];
            const float a0 = sY0[m * RK_MAX_NS + r];
            float k[7], bl[RK_MAX_STAGES];
#pragma unroll
            for (int j = 0; j < 7; ++j) k[j] = sK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + r];
#pragma unroll
            for (int j = 0; j < RK_MAX_STAGES; ++j) bl[j] = L.beta[sl][j];
            float a1 = a0;                            // y1 = the last stage's input (the expression that wrote Y[6])
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float t = a1 + k[j] * (bl[j] * h);
                a1 = (j < sl && bl[j] != 0.f) ? t : a1;
            }
            const float o = dopri_interp_value(a0, a1, k, h, w.ip_x);
            if (row < n) L.ip_out[(long)row * ns + r] = o;
            T.sF[m * RK_MAX_NS + r] = o;              // (f(x) of the last stage is no longer needed)
        }
        if (L.ip_kind == 1) {
            __syncthreads();
            if (tid < 2 * NLBAC_MLP_TILE) {          // the look-ahead point: component `which` of row mm
                const int mm = tid >> 1, which = tid & 1, rw = row0 + mm;
                const float th = T.sF[mm * RK_MAX_NS + 2];
                const float v = T.sF[mm * RK_MAX_NS + which] + L.ip_l * (which ? sinf(th) : cosf(th));
                if (rw < n) L.ip_p[(long)rw * 2 + which] = v;
            }
        }
    }
    if (L.norm_mode < 0) return;
    __shared__ unsigned s_last;
    if (L.norm_mode == 2) __syncthreads();         // (uniform) the terms in sG
    if (tid < 64) {
        const int m = tid;
        float v0 = 0.f, v1 = 0.f;
        if (m < n_rows) {
            for (int r = 0; r < ns; ++r) {
                const float y = sY0[m * RK_MAX_NS + r];
                if (L.norm_mode == 2) {
                    v0 += T.sG[m * RK_MAX_NS + r];
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
            if (L.norm_defer) { q[0] = v0; q[1] = v1; }        // (the next launch sums them: rk_fwd_norm_pre)
            else {
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
    }
    if (L.norm_defer) return;                  // (uniform)
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
            if (L.coh && L.norm_mode == 1) {      // (persistent launch: what the previous phase's controller — possibly on
                c[C_H0] = rk_ctl_load(c + C_H0, true);     //  another XCD — left there: read past this XCD's L2)
                c[C_D1] = rk_ctl_load(c + C_D1, true);
            }
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
            if (L.pers_gen) {
                // persistent launch: the other workgroups of this problem go on with the next phase once they see
                // pers_target.  What they read of the control block leaves through device-scope exchanges (performed
                // beyond the XCDs' L2s when they return — no agent-scope fence, which would write this XCD's L2 back),
                // the flag is stored behind them.
                double olds = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) olds += __hip_atomic_exchange(c + k, c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::"v"(olds) : "memory");
                __hip_atomic_store(L.pers_gen + p_tile, L.pers_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// persistent launch: wait until this problem's controller of the phase just finished has released it (rk_fwd_outputs_and_control).
// Bounded: a workgroup that never sees the flag (which needs every workgroup of the launch to be resident — the host only
// asks for this launch where that holds) gives up after ~1 s, marks the solve (C_OVF = 3) and the caller returns.
__device__ __forceinline__ bool rk_fwd_grid_wait(const NodeRkLaunch& L, int row0) {
    __shared__ int s_ok_;
    const int p = row0 / L.rpp;
    if (threadIdx.x == 0) {
        int ok = 0;
        for (long it = 0; it < (1L << 23); ++it) {
            if (__hip_atomic_load(L.pers_gen + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L.pers_target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) L.ctl_w[(long)p * NLBAC_DOPRI_CTL + C_OVF] = 3.0;
        s_ok_ = ok;
    }
    __syncthreads();
    const bool r = s_ok_ != 0;
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// backward: the small per-tile arrays in LDS
// ---------------------------------------------------------------------------------------------------------------
struct RkBwdTile {
    float* sDK;    // [stage][32][8]   dL/dK_j
    float* sU;     // [32][4]
    float* sH;     // [32]
    float* sDY0;   // [32][8]          running dy0
    float* sDU;    // [32][4]          running du
    float* sDX;    // [2][32][8]       dX of f_net / g_net for the current stage
    float* sDYup;  // [32][8]          dL/dy1 of the step when the launch forms it itself (nlbac_rk_chain::interp_*)
    __host__ __device__ __forceinline__ static constexpr int floats() {
        return RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS + NLBAC_MLP_TILE * (RK_MAX_NU + 1 + RK_MAX_NS + RK_MAX_NU + 2 * RK_MAX_NS + RK_MAX_NS);
    }
    __device__ __forceinline__ void carve(float* p) {
        sDK = p;
        sU = sDK + RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS;
        sH = sU + NLBAC_MLP_TILE * RK_MAX_NU;
        sDY0 = sH + NLBAC_MLP_TILE;
        sDU = sDY0 + NLBAC_MLP_TILE * RK_MAX_NS;
        sDX = sDU + NLBAC_MLP_TILE * RK_MAX_NU;
        sDYup = sDX + 2 * NLBAC_MLP_TILE * RK_MAX_NS;
    }
};

struct RkBwdWhere {
    long soff; int slot; bool chained, carry;
    bool ip;       // this launch forms the step's dK / dy0 / dy1 from the gradient of the solve's result (interpolant backward)
    const float* gG; float* gdG; float* gdK; float* gdy0; const float* gdYup;
    int st_lo; bool stage0_data;
    __device__ __forceinline__ bool has_data(int st) const { return st >= st_lo && (st > 0 || stage0_data); }
};
__device__ __forceinline__ bool rk_bwd_where(const NodeRkBwdLaunch& L, int row0, RkBwdWhere& w) {
    w.soff = 0;
    w.slot = 0;
    w.chained = L.ctl != nullptr;
    if (w.chained) {
        const int p_tile = row0 / L.rpp;
        w.slot = (int)L.ctl[(long)p_tile * NLBAC_DOPRI_CTL + C_NACC] - L.back_idx;
        if (w.slot < 0) return false;                       // (uniform) this problem took fewer steps
        w.soff = (long)w.slot * L.slot_floats;
    }
    w.carry = w.chained && L.back_idx > 0;    // not the problem's last step: gradients arrive from the slot behind
    w.ip = L.ip_on && w.chained && !w.carry;
    w.gG = L.G + w.soff;
    w.gdG = L.dG ? L.dG + w.soff : nullptr;
    w.gdK = L.dK + w.soff;
    w.gdy0 = L.dy0 ? L.dy0 + w.soff : nullptr;
    w.gdYup = w.carry ? w.gdy0 + L.slot_floats : (L.dYup ? L.dYup + w.soff : nullptr);
    w.st_lo = w.chained ? (w.slot == 0 ? 0 : 1) : L.st_lo;    // (a later step's stage 0 is its predecessor's last stage)
    w.stage0_data = L.dx_stage0 || L.dz[0] != nullptr;
    return true;
}

// u, running du / dy0, h, dK of the tile -> LDS.  No barrier inside.
template <int NTHR>
__device__ __forceinline__ void rk_bwd_tile_constants(const NodeRkBwdLaunch& L, const RkBwdWhere& w, const RkBwdTile& T,
                                                      int row0, int tid) {
    const int n = L.n, ns = L.n_s, nu = L.n_u;
    // (all global loads first, then the LDS stores: see rk_fwd_tile_constants)
    constexpr int NY = (NLBAC_MLP_TILE * RK_MAX_NS + NTHR - 1) / NTHR, NU = (NLBAC_MLP_TILE * RK_MAX_NU + NTHR - 1) / NTHR;
    constexpr int NK = (RK_MAX_STAGES * NLBAC_MLP_TILE * RK_MAX_NS + NTHR - 1) / NTHR;
    const bool du_in = L.du && L.du_acc, dy_in = w.gdy0 && L.dy0_in && !w.carry && !w.ip;
    float vu[NU], vdu[NU], vy[NY], vk[NK], vh = 0.f;
    float ip_g = 0.f, ip_h = 0.f, ip_xx = 0.f;
    if (w.ip) {      // (uniform) this thread's (row, component) of d loss / d y(t_end): every load up front with the others
        static_assert(NTHR == NLBAC_MLP_TILE * RK_MAX_NS, "one (row, component) per thread");
        const int m = tid >> 3, c = tid & 7, row = min(row0 + m, n - 1), p = row / L.rpp;
        ip_h = (float)L.ctl[(long)p * NLBAC_DOPRI_CTL + C_HUSED];
        ip_xx = (float)L.ctl[(long)p * NLBAC_DOPRI_CTL + C_X];
        if (L.ip_kind == 1) {        // out-map 1 (ode_kernels.hip::dopri_interp_bwd_kernel's arithmetic)
            float d0 = L.ip_dp[(long)row * 2 + 0], d1 = L.ip_dp[(long)row * 2 + 1];
            if (L.ip_dp2) { d0 += L.ip_dp2[(long)row * 2 + 0]; d1 += L.ip_dp2[(long)row * 2 + 1]; }
            const float th = L.ip_x[(long)row * ns + 2];
            const float g2 = L.ip_l * (-sinf(th) * d0 + cosf(th) * d1);
            ip_g = c == 0 ? d0 : (c == 1 ? d1 : (c == 2 ? g2 : 0.f));
        } else {
            ip_g = L.ip_dout[(long)row * ns + min(c, ns - 1)];
        }
    }
#pragma unroll
    for (int it = 0; it < NU; ++it) {
        const int idx = tid + NTHR * it, m = (idx >> 2) & (NLBAC_MLP_TILE - 1), c = idx & 3;
        const long rc = (long)min(row0 + m, n - 1) * nu + min(c, nu - 1);
        vu[it] = L.u[rc];
        vdu[it] = du_in ? L.du[rc] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NY; ++it) {
        const int idx = tid + NTHR * it, m = (idx >> 3) & (NLBAC_MLP_TILE - 1), c = idx & 7;
        vy[it] = dy_in ? w.gdy0[(long)min(row0 + m, n - 1) * ns + min(c, ns - 1)] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE) {
        const int p = min(row0 + tid, n - 1) / L.rpp;
        vh = w.chained ? (float)L.hslots[(long)p * L.n_slots + w.slot]
                       : (L.h_dev ? (float)L.h_dev[(long)p * L.h_stride] : L.h_val[p]);
    }
#pragma unroll
    for (int it = 0; it < NK; ++it) {
        const int idx = tid + NTHR * it;
        const int j = idx / (NLBAC_MLP_TILE * RK_MAX_NS), rem = idx - j * NLBAC_MLP_TILE * RK_MAX_NS;
        const int m = rem >> 3, c = rem & 7;
        const long rc = (long)min(row0 + m, n - 1) * ns + min(c, ns - 1);
        float v = 0.f;
        if (j < L.st_hi && !w.ip) {           // (uniform per wave)
            if (!w.carry) v = w.gdK[(long)j * n * ns + rc];
            else if (j == L.S_total - 1) v = (w.gdK + L.slot_floats)[rc];     // FSAL: next slot's dK[0]
        }
        vk[it] = v;
    }
    if (w.ip) {
        float d0v, d1v, dk[7];
        dopri_interp_grad(ip_g, ip_h, ip_xx, d0v, d1v, dk);
        const bool ok = row0 + (tid >> 3) < n && (tid & 7) < ns;
        vy[0] = d0v;
        T.sDYup[tid] = ok ? d1v : 0.f;
#pragma unroll
        for (int it = 0; it < NK; ++it)
            if (it < 7) vk[it] = dk[it];
    }
#pragma unroll
    for (int it = 0; it < NU; ++it) {
        const int idx = tid + NTHR * it, m = idx >> 2, c = idx & 3;
        if (idx < NLBAC_MLP_TILE * RK_MAX_NU) {
            const bool ok = row0 + m < n && c < nu;
            T.sU[idx] = ok ? vu[it] : 0.f;
            T.sDU[idx] = ok ? vdu[it] : 0.f;
        }
    }
#pragma unroll
    for (int it = 0; it < NY; ++it) {
        const int idx = tid + NTHR * it, m = idx >> 3, c = idx & 7;
        if (idx < NLBAC_MLP_TILE * RK_MAX_NS) T.sDY0[idx] = (row0 + m < n && c < ns) ? vy[it] : 0.f;
    }
    if (tid < NLBAC_MLP_TILE) T.sH[tid] = vh;
#pragma unroll
    for (int it = 0; it < NK; ++it) {
        const int idx = tid + NTHR * it;
        const int j = idx / (NLBAC_MLP_TILE * RK_MAX_NS), rem = idx - j * NLBAC_MLP_TILE * RK_MAX_NS;
        const int m = rem >> 3, c = rem & 7;
        if (j < L.st_hi) T.sDK[idx] = (row0 + m < n && c < ns) ? vk[it] : 0.f;
    }
}

// du += g(Y_st)^T dK_st for the tile's rows
template <int NTHR>
__device__ __forceinline__ void rk_bwd_du(const NodeRkBwdLaunch& L, const RkBwdWhere& w, const RkBwdTile& T, int row0,
                                          int st, int tid) {
    if (!L.du) return;
    const int n = L.n, ns = L.n_s, nu = L.n_u, gout = ns * nu;
    for (int idx = tid; idx < NLBAC_MLP_TILE * nu; idx += NTHR) {
        const int m = idx / nu, c = idx - m * nu, row = min(row0 + m, n - 1);
        float a = 0.f;
        for (int r = 0; r < ns; ++r)
            a += w.gG[((long)st * n + row) * gout + r * nu + c] * T.sDK[(st * NLBAC_MLP_TILE + m) * RK_MAX_NS + r];
        T.sDU[m * RK_MAX_NU + c] = T.sDU[m * RK_MAX_NU + c] + 1.0f * a;
    }
}

// the stage algebra once both nets' dX sit in sDX: dY = [dYup at the last stage] + dX_f + dX_g; dy0 += dY;
// dK[j] += beta[st][j] h dY (j < st)
template <int NTHR>
__device__ __forceinline__ void rk_bwd_stage_algebra(const NodeRkBwdLaunch& L, const RkBwdWhere& w, const RkBwdTile& T,
                                                     int row0, int st, int tid) {
    const int n = L.n, ns = L.n_s;
    for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += NTHR) {
        const int m = idx / ns, c = idx - m * ns, row = row0 + m;
        float d = (w.gdYup && !w.ip && st == L.S_total - 1 && row < n) ? w.gdYup[(long)row * ns + c] : 0.f;
        if (w.ip && st == L.S_total - 1) d = T.sDYup[m * RK_MAX_NS + c];
        d += T.sDX[m * RK_MAX_NS + c];
        d += T.sDX[(NLBAC_MLP_TILE + m) * RK_MAX_NS + c];
        T.sDY0[m * RK_MAX_NS + c] = T.sDY0[m * RK_MAX_NS + c] + d;
        const float h = T.sH[m];
        for (int j = 0; j < st; ++j)
            if (L.beta[st][j] != 0.f) T.sDK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c] += (L.beta[st][j] * h) * d;
    }
}

// dK, dy0, du of the tile -> global
template <int NTHR>
__device__ __forceinline__ void rk_bwd_outputs(const NodeRkBwdLaunch& L, const RkBwdWhere& w, const RkBwdTile& T,
                                               int row0, int tid) {
    const int n = L.n, ns = L.n_s, nu = L.n_u;
    for (int idx = tid; idx < L.st_hi * NLBAC_MLP_TILE * ns; idx += NTHR) {
        const int j = idx / (NLBAC_MLP_TILE * ns), rem = idx - j * NLBAC_MLP_TILE * ns;
        const int m = rem / ns, c = rem - m * ns, row = row0 + m;
        if (row < n) w.gdK[((long)j * n + row) * ns + c] = T.sDK[(j * NLBAC_MLP_TILE + m) * RK_MAX_NS + c];
    }
    if (w.gdy0)
        for (int idx = tid; idx < NLBAC_MLP_TILE * ns; idx += NTHR) {
            const int m = idx / ns, c = idx - m * ns, row = row0 + m;
            if (row < n) w.gdy0[(long)row * ns + c] = T.sDY0[m * RK_MAX_NS + c];
        }
    if (L.du)
        for (int idx = tid; idx < NLBAC_MLP_TILE * nu; idx += NTHR) {
            const int m = idx / nu, c = idx - m * nu, row = row0 + m;
            if (row < n) L.du[(long)row * nu + c] = T.sDU[m * RK_MAX_NU + c];
        }
}
