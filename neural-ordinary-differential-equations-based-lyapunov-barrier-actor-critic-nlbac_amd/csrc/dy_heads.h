// "dy heads": where nlbac_mlp_bwd_data_head takes dL/dy of its nets from.  The per-row operations that sat in launches of
// their own between a forward and the data backward that consumes their result — the squashed-Gaussian sample backward,
// the TD / Lyapunov targets, the min(Q1, Q2)(s, pi) branch terms — run in the backward's prologue instead, for the 32
// rows of the workgroup's tile, with their batch sums finished by the launch's last designated workgroup
// (publish_and_elect, common.h).  The row arithmetic is the one of gauss_bwd_kernel / td_targets_kernel /
// actor_q_terms_kernel (agent_kernels.hip), which stay for the data-parallel path and the other callers.
//
// Reference lines (U = NLBAC_Unicycle_RL_training/Unicycle_RL_training):
//   GaussianPolicy.sample            U/sac_cbf_clf/model.py:116-128
//   targets + MSE                    U/sac_cbf_clf/sac_cbf_clf.py:231-246
//   policy_loss_1 / alpha loss       U/sac_cbf_clf/sac_cbf_clf.py:258-273, 292-308
#pragma once
#include "common.h"
#include "auglag_device.h"
#include "scalars.h"

#define LOG_SIG_MAX 2.0f
#define LOG_SIG_MIN (-20.0f)
#define SAMPLE_EPS 1e-6f
#define MAX_NU 4

struct ActorScalarArgs {      // nlbac_actor_scalar_args: what nlbac_actor_scalars needs, per problem (0 primary, 1 backup)
    float target_entropy; const float* log_alpha[2]; float* g_log_alpha[2]; float* sc;
};
__device__ __forceinline__ void actor_scalars_one(float s0, float s1, int p, int B, float target_entropy,
                                                  const float* log_alpha, float* g_log_alpha, float* sc) {
    const float pl1 = s0 / (float)B;
    const float mean_lp = s1 / (float)B;
    const float la = log_alpha[0];
    const float aloss = -(la * (mean_lp + target_entropy));      // alpha_loss = -(log_alpha * (logp + H)).mean()
    sc[(p == 0) ? SC_PL1 : SC_BPL1] = pl1;
    sc[(p == 0) ? SC_ALOSS : SC_BALOSS] = aloss;
    sc[(p == 0) ? SC_MEAN_LOGP : SC_MEAN_BLOGP] = mean_lp;
    g_log_alpha[0] = -(mean_lp + target_entropy);
}

// GaussianPolicy.sample of one row: heads = the row's (mean | log_std), i = the row's index in the stacked eps / action /
// logp arrays
__device__ __forceinline__ void gauss_fwd_row(const float* heads, const float* eps, const float* scale, const float* bias,
                                              int n_u, long i, float* action, int action_ld, float* logp) {
    float lp = 0.f;
    for (int c = 0; c < n_u; ++c) {
        const float mean = heads[c];
        float ls = heads[n_u + c];
        ls = fminf(fmaxf(ls, LOG_SIG_MIN), LOG_SIG_MAX);
        const float std = expf(ls);
        const float e = eps[i * n_u + c];
        const float x = mean + e * std;
        const float y = tanhf(x);
        action[i * action_ld + c] = y * scale[c] + bias[c];
        const float var = std * std;
        const float d = x - mean;
        float l = -(d * d) / (2.0f * var) - ls - 0.91893853320467274178f;   // log(sqrt(2 pi))
        l -= logf(scale[c] * (1.0f - y * y) + SAMPLE_EPS);
        lp += l;
    }
    logp[i] = lp;
}

// d heads of one (row, action component) from d action (up to three sources) and d logp = alpha * dlogp_mul
__device__ __forceinline__ void gauss_bwd_one(const float* heads, int heads_ld, const float* eps, const float* scale,
                                              int n_u, long i, int c, const float* da0, int da0_ld, const float* da1,
                                              int da1_ld, const float* da2, int da2_ld, float dlp, float& dmean,
                                              float& dls) {
    const float mean = heads[i * heads_ld + c];
    const float ls_raw = heads[i * heads_ld + n_u + c];
    const float ls = fminf(fmaxf(ls_raw, LOG_SIG_MIN), LOG_SIG_MAX);
    const float std = expf(ls);
    const float e = eps[i * n_u + c];
    const float y = tanhf(mean + e * std);
    float da = 0.f;
    if (da0) da += da0[i * da0_ld + c];
    if (da1) da += da1[i * da1_ld + c];
    if (da2) da += da2[i * da2_ld + c];
    const float one_m = 1.0f - y * y;
    const float s1 = scale[c] * one_m;
    // dx through a = scale*tanh(x)+bias and through -log(scale(1-y^2)+eps)
    const float gx = da * s1 + dlp * (2.0f * y * s1 / (s1 + SAMPLE_EPS));
    // the -(x-mean)^2/(2 var) term is constant (-eps^2/2) under reparameterisation
    dmean = gx;
    const float dstd = gx * e;
    const bool in_range = (ls_raw >= LOG_SIG_MIN) && (ls_raw <= LOG_SIG_MAX);
    dls = in_range ? (dstd * std - dlp) : 0.f;
}

// Sum of n_tiles partials (stride floats apart) read past the non-coherent caches: thread t takes tiles t, t + 256, ...
// in order, then the fixed tree of block_sum_256 — the result does not depend on which workgroup was elected.
template <int NV>
__device__ __forceinline__ void elected_tile_sums(const float* partials, int n_tiles, int stride, float (&v)[NV], float* red,
                                                  bool coherent = true) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.f;
    for (int b = threadIdx.x; b < n_tiles; b += 256)
#pragma unroll
        for (int k = 0; k < NV; ++k) {      // (coherent: published by other workgroups of THIS launch; else an earlier launch's)
            const float* q = partials + (long)b * stride + k;
            v[k] += coherent ? coherent_load(q) : *q;
        }
    block_sum_256<NV>(v, red);
}

// Partial sums a tile's dy head leaves for the launch-wide election (kind 2: v0 = the tile's squared-error sum; kind 3:
// v0, v1 = sums of (alpha logp - min q, logp)); valid in thread 0.
struct DyHeadPending { float v0, v1; };

// Fills sdy[32][16] (dL/dy of the tile's rows, zero padded) of net `inet` of the launch.  Called by all 256 threads of an
// mlp_bwd_data workgroup before anything reads sdy.  What the tile contributes to the launch's batch sums is returned in
// `pend` for dy_head_finish — which may run at any later point of the kernel (the LDS-tiled kernel calls it right away;
// the register-resident one at its very end, so that the atomics' round trip is off the tile's critical path).
// TILE: rows of the workgroup's tile (32: the LDS-tiled and half-panel kernels; 16: the quarter-panel kernels).
template <int TILE = NLBAC_MLP_TILE>
__device__ __forceinline__ void dy_head_rows(const nlbac_dy_head& H, int inet, int row0, int B, float* sdy, DyHeadPending& pend) {
    const int tid = threadIdx.x;
    pend.v0 = pend.v1 = 0.f;
    if (H.kind == 1) {
        // GaussianPolicy.sample backward; net inet = controller inet, its rows are inet*B.. of the stacked arrays
        const float dlp = H.alpha[inet] * H.dlogp_mul;
        for (int idx = tid; idx < TILE * 16; idx += 256) {
            const int r = idx >> 4, c = idx & 15, row = row0 + r;
            float v = 0.f;
            if (row < B && c < 2 * H.n_u) {
                const long i = (long)inet * B + row;
                const int cu = (c < H.n_u) ? c : c - H.n_u;
                float dmean, dls;
                gauss_bwd_one(H.heads, H.heads_ld, H.eps, H.scale, H.n_u, i, cu, H.da[0], H.da_ld[0], H.da[1], H.da_ld[1],
                              H.da[2], H.da_ld[2], dlp, dmean, dls);
                v = (c < H.n_u) ? dmean : dls;
                if (H.dheads) H.dheads[i * H.dheads_ld + c] = v;
            }
            sdy[idx] = v;
        }
    } else if (H.kind == 2) {
        // TD / Lyapunov targets; net 0 = Q1, 1 = Q2, 2 = Lyapunov critic (3 = the learned-barrier copies' BarrierNet).
        float e2 = 0.f;
        for (int idx = tid; idx < TILE * 16; idx += 256) sdy[idx] = 0.f;
        __syncthreads();
        if (tid < TILE && row0 + tid < B) {
            const int i = row0 + tid;
            const float mk = H.mask[(long)i * H.rcm_ld];
            float y;
            if (inet < 2) {
                const float mq = fminf(H.q1t[i], H.q2t[i]) - H.alpha[0] * H.nlogp[i];
                y = H.reward[(long)i * H.rcm_ld] + mk * H.gamma * mq;
                if (inet == 0 && H.next_q) H.next_q[i] = y;
            } else if (inet == 2) {
                y = H.constraint[(long)i * H.rcm_ld] + mk * H.gamma * H.lt[i];
                if (H.next_l) H.next_l[i] = y;
            } else {
                y = H.xsig[(long)i * H.xsig_ld] + mk * H.gamma * H.xt[i];
            }
            const float norm = (float)(2.0 / (double)H.B_norm);
            const float e = (inet < 3 ? H.q[inet][i] : H.xq[i]) - y;
            const float d = norm * e;
            (inet < 3 ? H.dq[inet] : H.dxq)[i] = d;
            sdy[tid * 16] = d;
            e2 = e * e;
        }
        if (tid < 64) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) e2 += __shfl_down(e2, off, 64);
        }
        pend.v0 = e2;
    } else if (H.kind == 3 && inet >= 2 * H.n_prob) {
        // the net behind the Q pairs: its dL/dy is the constraint backward's dV_next (cb_kind 1:
        // unicycle_constraints_bwd_kernel's arithmetic, agent_kernels.hip — d ps_next of both controllers from the CBF
        // terms, dV_next from the CLF term), one thread per row of the tile
        // Two phases so that every global load of the tile is in flight at once: (row, hazard, controller) per thread
        // forms that hazard's addend, then one thread per row sums the addends in hazard order (the launch's order).
        __shared__ float s_cb[TILE * 2 * 16 * 2];
        __shared__ float s_ag[NLBAC_SC_SIZE_ENUM + 4 * 15];
        const int NH = H.cb_nh;
        const float* sc = H.cb_sc;
        if (H.cb_defer) {      // (uniform) the forward's constraint head left its tiles' column sums and ran no step: this
            //                    workgroup runs it on a private copy of the block — which this launch only reads — and takes
            //                    the coefficients from there; a later launch commits it (nlbac_head_sums kind 4)
            const AuglagArgs A = {H.cb_auglag.n_cbf, H.cb_auglag.n_clf, H.cb_auglag.batch_size, H.cb_auglag.do_lambda_update,
                                  H.cb_auglag.do_backup_lambda_update, H.cb_auglag.ratio_mode, H.cb_auglag.backup_mode,
                                  H.cb_auglag.lam_lo, H.cb_auglag.lam_hi};
            auglag_from_tiles<15, false>(H.cb_partials, H.cb_tiles[0], A, const_cast<float*>(H.cb_sc), s_ag,
                                         s_ag + NLBAC_SC_SIZE_ENUM, false);
            sc = s_ag;
            if (row0 == 0)      // the first tile leaves the stepped block for the job that commits it (nlbac_head_sums 4)
                for (int t = tid; t < NLBAC_SC_SIZE_ENUM; t += 256) H.cb_stage[t] = s_ag[t];
        }
        for (int idx = tid; idx < TILE * 16; idx += 256) sdy[idx] = 0.f;
        for (int t = tid; t < TILE * 2 * NH; t += 256) {
            const int r = t / (2 * NH), rem = t - r * 2 * NH, h = rem >> 1, which = rem & 1, i = min(row0 + r, B - 1);
            const float hx = H.cb_hazards[h * 2], hy = H.cb_hazards[h * 2 + 1];
            const long pr = which ? (long)(B + i) : (long)i;
            const float n0 = H.cb_ps_next[pr * 2], n1 = H.cb_ps_next[pr * 2 + 1];
            const float term = which ? H.cb_bmatr[(long)i * NH + h] : H.cb_matr[(long)i * (NH + 1) + h];
            const float coef = sc[(which ? SC_BCOEF : SC_COEF) + h];
            const float g = -((coef / H.cb_batch) / H.cb_dt);
            const bool on = term > 0.f;
            s_cb[((r * 2 + which) * 16 + h) * 2 + 0] = on ? g * (n0 - hx) : 0.f;
            s_cb[((r * 2 + which) * 16 + h) * 2 + 1] = on ? g * (n1 - hy) : 0.f;
        }
        float lya = 0.f, cl = 0.f;
        if (tid < TILE) {
            lya = H.cb_matr[(long)min(row0 + tid, B - 1) * (NH + 1) + NH];
            cl = sc[SC_COEF + NH];
        }
        __syncthreads();
        if (tid < TILE && row0 + tid < B) {
            const int i = row0 + tid;
            float d0 = 0.f, d1 = 0.f, e0 = 0.f, e1 = 0.f;
            for (int h = 0; h < NH; ++h) {       // (an inactive term's addend is +0: the sum is the launch's, term by term)
                d0 += s_cb[((tid * 2 + 0) * 16 + h) * 2 + 0]; d1 += s_cb[((tid * 2 + 0) * 16 + h) * 2 + 1];
                e0 += s_cb[((tid * 2 + 1) * 16 + h) * 2 + 0]; e1 += s_cb[((tid * 2 + 1) * 16 + h) * 2 + 1];
            }
            H.cb_dps_next[i * 2] = d0; H.cb_dps_next[i * 2 + 1] = d1;
            H.cb_dps_next[(long)(B + i) * 2] = e0; H.cb_dps_next[(long)(B + i) * 2 + 1] = e1;
            const float dv = (lya > 0.f) ? ((cl / H.cb_batch) / H.cb_dt) : 0.f;
            H.cb_dV[i] = dv;
            sdy[tid * 16] = dv;
        }
    } else if (H.kind == 3) {
        // min(Q1, Q2)(s, pi) branch gradients; net inet = (controller inet / 2, Q1 / Q2 = inet % 2)
        const int p = inet >> 1, which = inet & 1;
        float v0 = 0.f, v1 = 0.f;
        for (int idx = tid; idx < TILE * 16; idx += 256) sdy[idx] = 0.f;
        __syncthreads();
        if (tid < TILE && row0 + tid < B) {
            const long r = (long)p * B + row0 + tid;
            const float a = H.qa[r], b = H.qb[r];
            const float g = -(1.0f / (float)H.B_norm);
            const float da = (a < b) ? g : ((a == b) ? 0.5f * g : 0.f);
            const float db = (b < a) ? g : ((a == b) ? 0.5f * g : 0.f);
            const float d = which ? db : da;
            (which ? H.dqb : H.dqa)[r] = d;
            sdy[tid * 16] = d;
            v0 = H.alpha[p] * H.logp[r] - fminf(a, b);
            v1 = H.logp[r];
        }
        if (which == 0 && tid < 64) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) { v0 += __shfl_down(v0, off, 64); v1 += __shfl_down(v1, off, 64); }
        }
        pend.v0 = v0; pend.v1 = v1;
    }
}

// The launch-wide half of a dy head: every designated workgroup publishes its tile's sums, the last one to arrive
// finishes the batch quantities (publish_and_elect, common.h).  All 256 threads; `red`: >= 16 floats of LDS scratch that
// nothing else uses until the call returns.
// The batch sums of a kind-2 / kind-3 head from its tile partials: what the elected workgroup of the head's own launch
// does, or — nlbac_dy_head::sums_defer — a workgroup of a later launch (dy_head_jobs).  All 256 threads.
__device__ __forceinline__ void head_sums_finish(const nlbac_head_sums& J, int n_tiles, float* red, bool coh = true) {
    const int tid = threadIdx.x;
    if (J.kind == 2) {
        float s[3];
        elected_tile_sums<3>(J.partials, n_tiles, J.n_nets, s, red, coh);
        if (tid < 3) J.out[tid] = s[tid] * J.mul;
        if (J.n_nets == 4) {
            float sx[1];
            elected_tile_sums<1>(J.partials + 3, n_tiles, 4, sx, red, coh);
            if (tid == 0) J.out_x[0] = sx[0] * J.mul;
        }
    } else if (J.kind == 3) {
        for (int pp = 0; pp < J.n_nets; ++pp) {
            float s[2];
            elected_tile_sums<2>(J.partials + (long)pp * n_tiles * 2, n_tiles, 2, s, red, coh);
            if (tid == 0)
                actor_scalars_one(s[0], s[1], pp, J.B_norm, J.actor.target_entropy, J.actor.log_alpha[pp],
                                  J.actor.g_log_alpha[pp], J.actor.sc);
        }
    } else if (J.kind == 4) {
        // commit the augmented-Lagrangian step an earlier launch staged: the entries the step may change (auglag_finish_at)
        for (int t = tid; t < NLBAC_SC_SIZE_ENUM; t += 256)
            if ((t >= SC_RATIO && t <= SC_BPL2) || (t >= SC_LAMBDA && t < SC_MEAN_LOGP)) J.sc[t] = J.partials[t];
    }
}

template <int TILE = NLBAC_MLP_TILE>
__device__ __forceinline__ void dy_head_finish(const nlbac_dy_head& H, int inet, int row0, int n_tiles, float* red, int n_nets,
                                               const DyHeadPending& pend) {
#ifdef EXP_NO_DY_FINISH       /* ablation (timing only: the batch sums are not formed; 1: none, 2 / 3: not that kind's): what
                                 the elections cost the update — kind 2 (critic losses) 5.2 us, kind 3 (alpha terms) 4.3 us of
                                 550 at B = 4096, MI355X */
    if (EXP_NO_DY_FINISH == 1 || EXP_NO_DY_FINISH == H.kind) return;
#endif
    const int tid = threadIdx.x;
    const int tile = row0 / TILE;
    if (H.kind == 2) {
        // every net's workgroup publishes its own squared-error sum of the tile; the last of the n_nets * n_tiles
        // workgroups finishes the losses — or (sums_defer) a workgroup of a later launch does
        if (H.sums_defer) {
            if (tid == 0) {
                H.partials[(long)tile * n_nets + inet] = pend.v0;
                if (tile == 0 && inet == 0) H.sums_tiles[0] = (unsigned)n_tiles;
            }
            return;
        }
        const float v1[1] = {pend.v0};
        if (publish_and_elect_grouped<1>(H.partials + (long)tile * n_nets + inet, v1, H.ticket, (unsigned)(tile * n_nets + inet),
                                         (unsigned)(n_nets * n_tiles))) {
            nlbac_head_sums J;
            J.kind = 2; J.n_nets = n_nets; J.partials = H.partials; J.mul = H.mul; J.out = H.out; J.out_x = H.out_x;
            head_sums_finish(J, n_tiles, red);
        }
    } else if (H.kind == 3) {
        // the Q1 workgroups publish the tile's sums of (alpha logp - min q, logp); the last of them finishes
        // policy_loss_1 / alpha loss / d log_alpha of every controller (actor_scalars_one)
        const int p = inet >> 1, which = inet & 1, n_prob = H.n_prob;
        if (which == 0 && inet < 2 * n_prob) {        // (the net behind the Q pairs — cb_kind — has no batch sums)
            if (H.sums_defer) {
                if (tid == 0) {
                    float* q = H.partials + ((long)p * n_tiles + tile) * 2;
                    q[0] = pend.v0; q[1] = pend.v1;
                    if (tile == 0 && inet == 0) H.sums_tiles[0] = (unsigned)n_tiles;
                }
                return;
            }
            const float v2[2] = {pend.v0, pend.v1};
            if (publish_and_elect_grouped<2>(H.partials + ((long)p * n_tiles + tile) * 2, v2, H.ticket, (unsigned)(p * n_tiles + tile),
                                             (unsigned)(n_prob * n_tiles))) {
                nlbac_head_sums J;
                J.kind = 3; J.n_nets = n_prob; J.partials = H.partials; J.B_norm = H.B_norm; J.actor = H.actor;
                head_sums_finish(J, n_tiles, red);
            }
        }
    }
}

// nlbac_dy_head::finish: the sums an earlier launch deferred, by workgroup (tile j, net 0) of this one.  Every thread of
// every workgroup calls it at the END of the kernel (contains a barrier for the workgroups that have a job; uniform).
__device__ __forceinline__ void dy_head_jobs(const nlbac_dy_head& H, int tile, int inet, int n_tiles_launch, float* red) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (H.finish[j].kind != 0 && inet == 0 && tile == (j < n_tiles_launch ? j : 0)) {
            __syncthreads();                            // (red is free: every earlier use of the scratch is done)
            head_sums_finish(H.finish[j], H.finish[j].kind == 4 ? 0 : (int)H.finish[j].n_tiles[0], red, false);
        }
}

__device__ __forceinline__ void dy_head_fill(const nlbac_dy_head& H, int inet, int row0, int B, int n_tiles, float* sdy,
                                             float* red, int n_nets) {
    DyHeadPending pend;
    dy_head_rows(H, inet, row0, B, sdy, pend);
    dy_head_finish(H, inet, row0, n_tiles, red, n_nets, pend);
}
