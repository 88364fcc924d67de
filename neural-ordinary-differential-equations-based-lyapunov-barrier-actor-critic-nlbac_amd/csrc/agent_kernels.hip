// Per-sample (elementwise + reduction) kernels of the SAC / CLF / CBF update:
// squashed-Gaussian sampling, TD targets, Unicycle geometry, constraint
// terms, augmented-Lagrangian scalars.  One lane per sample, coalesced
// row reads, wavefront shuffle + LDS block reductions to per-block partials
// that a single fixed-order pass sums (deterministic).
//
// Reference lines (U = NLBAC_Unicycle_RL_training/Unicycle_RL_training):
//   GaussianPolicy.sample            U/sac_cbf_clf/model.py:116-128
//   targets + MSE                    U/sac_cbf_clf/sac_cbf_clf.py:231-246
//   policy_loss_1 / alpha loss       U/sac_cbf_clf/sac_cbf_clf.py:258-273, 292-308
//   get_state                        U/sac_cbf_clf/dynamics.py:53-58
//   get_policy_loss_2                U/sac_cbf_clf/sac_cbf_clf.py:408-530
//   backup_get_policy_loss_2         U/sac_cbf_clf/sac_cbf_clf.py:532-640
#include "common.h"
#include "scalars.h"
#include "dy_heads.h"

#include "auglag_device.h"

template <bool COHERENT>
__device__ __forceinline__ void auglag_body(const float* partials, int n_blk, const AuglagArgs A, float* sc);

// runtime-width variant of publish_and_elect (common.h), n <= 64: the n exchanges are ONE wave instruction (lane k
// publishes value k) — issued by thread 0 alone they were n dependent round trips to the memory side, ~0.6 us each
// (15 columns: unicycle_constraints_fwd 14.6 -> ~6 us)
__device__ __forceinline__ bool publish_and_elect_n(float* dst, const float* vals, int n, unsigned* ticket, unsigned n_blocks) {
    __shared__ unsigned s_elect_n_;
    __shared__ float s_vals_n_[64];
    if (threadIdx.x == 0)
        for (int k = 0; k < n; ++k) s_vals_n_[k] = vals[k];
    __syncthreads();
    if (threadIdx.x < 64) {
        float old = 0.f;
        if ((int)threadIdx.x < n)
            old = __hip_atomic_exchange(dst + threadIdx.x, s_vals_n_[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");        // every lane's exchange has returned before the ticket is taken
        elect_release_();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_elect_n_ = (t == n_blocks - 1u) ? 1u : 0u;
            if (s_elect_n_) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_elect_n_ != 0u) elect_acquire_();
    return s_elect_n_ != 0u;
}

// ---------------------------------------------------------------------------
// Self-test of the election (common.h: the gfx950 contract).  Workgroup b publishes n_vals integers-as-floats that
// depend on (b, k, salt) — a stale partial of an earlier launch (other salt) or a partial that had not landed when the
// last ticket was drawn changes the sum — through publish_and_elect<4> (n_vals == 4) or the runtime-width variant; the
// elected workgroup sums all workgroups' values with coherent loads, in order, into out[0..n_vals) and counts itself in
// out[n_vals] (exactly one workgroup per launch must be elected).  Every workgroup first spends a block-dependent
// number of cycles so that the tickets are drawn in a scrambled order.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float elect_test_value(unsigned b, unsigned k, unsigned salt) {
    return (float)((b * 31u + k * 7u + salt * 13u) % 251u);
}
__global__ __launch_bounds__(256) void elect_selftest_kernel(float* partials, unsigned* ticket, float* out, int n_vals,
                                                             unsigned salt) {
    const unsigned b = blockIdx.x;
    const unsigned spin = ((b * 2654435761u) >> 24) * 8u;        // 0 .. 2040 cycles
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (long long)spin) {}
    float v[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) v[k] = (k < n_vals) ? elect_test_value(b, k, salt) : 0.f;
    bool elected;
    if (n_vals == 4) {
        float v4[4] = {v[0], v[1], v[2], v[3]};
        elected = publish_and_elect<4>(partials + (long)b * 4, v4, ticket, gridDim.x);
    } else if (n_vals == 2) {      // the two-level election of the dy heads (ticket: 1 + ceil(n_blocks / 16) words)
        float v2[2] = {v[0], v[1]};
        elected = publish_and_elect_grouped<2>(partials + (long)b * 2, v2, ticket, b, gridDim.x);
    } else {
        __shared__ float sv[64];
        if (threadIdx.x == 0)
            for (int k = 0; k < n_vals; ++k) sv[k] = elect_test_value(b, k, salt);
        __syncthreads();
        elected = publish_and_elect_n(partials + (long)b * n_vals, sv, n_vals, ticket, gridDim.x);
    }
    if (!elected) return;
    if ((int)threadIdx.x < n_vals) {
        float acc = 0.f;                                       // (integers below 2^24: exact in any order)
        for (unsigned j = 0; j < gridDim.x; ++j) acc += coherent_load(partials + (long)j * n_vals + threadIdx.x);
        out[threadIdx.x] = acc;
    }
    if (threadIdx.x == 0) atomicAdd(out + n_vals, 1.0f);
}

extern "C" int nlbac_elect_selftest(float* partials, unsigned* ticket, float* out, int n_blocks, int n_vals, unsigned salt,
                                    nlbac_stream_t s) {
    NLBAC_REQUIRE(partials && ticket && out, "nlbac_elect_selftest: null buffer");
    NLBAC_REQUIRE(n_blocks >= 1 && n_vals >= 1 && n_vals <= 64, "nlbac_elect_selftest: n_blocks >= 1, 1 <= n_vals <= 64");
    hipLaunchKernelGGL(elect_selftest_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)s, partials, ticket, out, n_vals, salt);
    NLBAC_CHECK_LAUNCH("nlbac_elect_selftest");
    return 0;
}
// tail of a constraints_fwd kernel: plain partials, or (ticket) publish them and let the last workgroup run auglag
#define CONSTRAINTS_TAIL(NCOLS_, V_)                                                              \
    if (!ticket) {                                                                                \
        if (threadIdx.x == 0)                                                                     \
            for (int c_ = 0; c_ < (NCOLS_); ++c_) partials[(long)blockIdx.x * (NCOLS_) + c_] = (V_)[c_]; \
        return;                                                                                   \
    }                                                                                             \
    if (!publish_and_elect_n(partials + (long)blockIdx.x * (NCOLS_), (V_), (NCOLS_), ticket, gridDim.x)) return; \
    auglag_body<true>(partials, (int)gridDim.x, A, sc);


// ---------------------------------------------------------------------------
// GaussianPolicy.sample forward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss_fwd_kernel(const float* heads, int heads_ld, const float* eps,
                                                        const float* scale, const float* bias, int n_u, int n,
                                                        float* action, int action_ld, float* logp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    gauss_fwd_row(heads + (long)i * heads_ld, eps, scale, bias, n_u, i, action, action_ld, logp);
}

// backward: d heads from d action (sum of up to three sources) and d logp = alpha[p] * dlogp_mul
__global__ __launch_bounds__(256) void gauss_bwd_kernel(const float* heads, int heads_ld, const float* eps,
                                                        const float* scale, int n_u, int n, int rows_per_problem,
                                                        const float* da0, int da0_ld, const float* da1, int da1_ld,
                                                        const float* da2, int da2_ld, const float* alpha,
                                                        float dlogp_mul, float* dheads, int dheads_ld) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float dlp = alpha[i / rows_per_problem] * dlogp_mul;
    for (int c = 0; c < n_u; ++c) {
        float dmean, dls;
        gauss_bwd_one(heads, heads_ld, eps, scale, n_u, i, c, da0, da0_ld, da1, da1_ld, da2, da2_ld, dlp, dmean, dls);
        dheads[(long)i * dheads_ld + c] = dmean;
        dheads[(long)i * dheads_ld + n_u + c] = dls;
    }
}

// ---------------------------------------------------------------------------
// TD / Lyapunov targets, MSE partial sums and dL/dq
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_targets_kernel(const float* q1t, const float* q2t, const float* lt,
                                                         const float* nlogp, const float* reward,
                                                         const float* constraint, const float* mask, int rcm_ld,
                                                         const float* q1, const float* q2, const float* lf,
                                                         const float* alpha, float gamma, int B, int B_norm,
                                                         float* dq1, float* dq2, float* dlf, float* next_q,
                                                         float* next_l, float* partials, unsigned* ticket,
                                                         float mul, float* out) {
    __shared__ float red[12];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[3] = {0.f, 0.f, 0.f};
    if (i < B) {
        const float a = alpha[0];
        const float mq = fminf(q1t[i], q2t[i]) - a * nlogp[i];
        const float mk = mask[(long)i * rcm_ld];
        const float yq = reward[(long)i * rcm_ld] + mk * gamma * mq;
        const float yl = constraint[(long)i * rcm_ld] + mk * gamma * lt[i];
        const float norm = (float)(2.0 / (double)B_norm);
        const float e1 = q1[i] - yq, e2 = q2[i] - yq, e3 = lf[i] - yl;
        dq1[i] = norm * e1; dq2[i] = norm * e2; dlf[i] = norm * e3;
        if (next_q) next_q[i] = yq;
        if (next_l) next_l[i] = yl;
        v[0] = e1 * e1; v[1] = e2 * e2; v[2] = e3 * e3;
    }
    block_sum_256<3>(v, red);
    if (!ticket) {
        if (threadIdx.x == 0) {
            partials[blockIdx.x * 3 + 0] = v[0];
            partials[blockIdx.x * 3 + 1] = v[1];
            partials[blockIdx.x * 3 + 2] = v[2];
        }
        return;
    }
    // the three loss sums in this launch (same order as nlbac_sum_partials: block 0, 1, ...)
    if (!publish_and_elect<3>(partials + blockIdx.x * 3, v, ticket, gridDim.x)) return;
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (unsigned b = 0; b < gridDim.x; ++b) s += coherent_load(partials + b * 3 + threadIdx.x);
        out[threadIdx.x] = s * mul;
    }
}

// ---------------------------------------------------------------------------
// min(Q1,Q2)(s, pi) branch gradients and partial sums for policy_loss_1 / alpha loss.
// Rows: P problems (primary, backup) x B.  partials: [P][nblk][2]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void actor_q_terms_kernel(const float* q1, const float* q2, const float* logp,
                                                            const float* alpha, int B, int B_norm, float* dq1,
                                                            float* dq2, float* partials, unsigned* ticket,
                                                            const ActorScalarArgs F) {
    __shared__ float red[8];
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[2] = {0.f, 0.f};
    if (i < B) {
        const long r = (long)p * B + i;
        const float a = q1[r], b = q2[r];
        const float g = -(1.0f / (float)B_norm);
        dq1[r] = (a < b) ? g : ((a == b) ? 0.5f * g : 0.f);
        dq2[r] = (b < a) ? g : ((a == b) ? 0.5f * g : 0.f);
        v[0] = alpha[p] * logp[r] - fminf(a, b);
        v[1] = logp[r];
    }
    block_sum_256<2>(v, red);
    if (!ticket) {
        if (threadIdx.x == 0) {
            partials[((long)p * gridDim.x + blockIdx.x) * 2 + 0] = v[0];
            partials[((long)p * gridDim.x + blockIdx.x) * 2 + 1] = v[1];
        }
        return;
    }
    // policy_loss_1 / alpha loss / d log_alpha in this launch (same sums, same order as actor_scalars_kernel)
    if (!publish_and_elect<2>(partials + ((long)p * gridDim.x + blockIdx.x) * 2, v, ticket, gridDim.x * gridDim.y)) return;
    if (threadIdx.x < gridDim.y) {
        const int pp = threadIdx.x, nblk = (int)gridDim.x;
        float s0 = 0.f, s1 = 0.f;
        for (int b = 0; b < nblk; ++b) {
            s0 += coherent_load(partials + ((long)pp * nblk + b) * 2 + 0);
            s1 += coherent_load(partials + ((long)pp * nblk + b) * 2 + 1);
        }
        actor_scalars_one(s0, s1, pp, B_norm, F.target_entropy, F.log_alpha[pp], F.g_log_alpha[pp], F.sc);
    }
}

// policy_loss_1, alpha losses, d log_alpha  (one thread)
__global__ void actor_scalars_kernel(const float* partials, int nblk, int B, int p0, int P, float target_entropy,
                                     const float* log_alpha, int log_alpha_stride, float* g_log_alpha, float* sc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // problems p0 .. p0+P-1 (0 primary, 1 backup); log_alpha / g_log_alpha point at problem p0's entry
    for (int p = p0; p < p0 + P; ++p) {
        float s0 = 0.f, s1 = 0.f;
        for (int b = 0; b < nblk; ++b) {
            s0 += partials[((long)p * nblk + b) * 2 + 0];
            s1 += partials[((long)p * nblk + b) * 2 + 1];
        }
        const float pl1 = s0 / (float)B;
        const float mean_lp = s1 / (float)B;
        const float la = log_alpha[(p - p0) * log_alpha_stride];
        // alpha_loss = -(log_alpha * (logp + H)).mean()
        const float aloss = -(la * (mean_lp + target_entropy));
        sc[(p == 0) ? SC_PL1 : SC_BPL1] = pl1;
        sc[(p == 0) ? SC_ALOSS : SC_BALOSS] = aloss;
        sc[(p == 0) ? SC_MEAN_LOGP : SC_MEAN_BLOGP] = mean_lp;
        g_log_alpha[(p - p0) * log_alpha_stride] = -(mean_lp + target_entropy);
    }
}

__global__ void alpha_refresh_kernel(const float* log_alpha, int log_alpha_stride, int p0, int P, float* sc) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int p = p0; p < p0 + P; ++p) sc[SC_ALPHA + p] = expf(log_alpha[(p - p0) * log_alpha_stride]);
}

// ---------------------------------------------------------------------------
// Unicycle geometry
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unicycle_state_kernel(const float* obs, int obs_ld, int B, float l_p,
                                                             float* state, int n_copies, float* ps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float* o = obs + (long)i * obs_ld;
    // the reference computes arctan2 on the host in float64 and casts back
    const float th = (float)atan2((double)o[3], (double)o[2]);
    for (int c = 0; c < n_copies; ++c) {       // one copy per controller whose rollout starts from this state
        float* st = state + ((long)c * B + i) * 3;
        st[0] = o[0]; st[1] = o[1]; st[2] = th;
    }
    if (ps) {
        ps[i * 2 + 0] = o[0] + l_p * cosf(th);
        ps[i * 2 + 1] = o[1] + l_p * sinf(th);
    }
}

__global__ __launch_bounds__(256) void unicycle_lookahead_kernel(const float* x, int n, float l_p, float* ps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float th = x[i * 3 + 2];
    ps[i * 2 + 0] = x[i * 3 + 0] + l_p * cosf(th);
    ps[i * 2 + 1] = x[i * 3 + 1] + l_p * sinf(th);
}

__global__ __launch_bounds__(256) void unicycle_lookahead_bwd_kernel(const float* x, const float* dps,
                                                                     const float* dps2, int n, float l_p, float* dx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float d0 = dps[i * 2 + 0], d1 = dps[i * 2 + 1];
    if (dps2) { d0 += dps2[i * 2 + 0]; d1 += dps2[i * 2 + 1]; }
    const float th = x[i * 3 + 2];
    dx[i * 3 + 0] = d0;
    dx[i * 3 + 1] = d1;
    dx[i * 3 + 2] = l_p * (-sinf(th) * d0 + cosf(th) * d1);
}

// ---------------------------------------------------------------------------
// CBF / CLF terms.  ps (B,2); ps_next (2B,2): rows [0,B) primary, [B,2B) backup.
// matr (B, n_hz+1) = [cbf_1..cbf_nhz, clf],  bmatr (B, n_hz).
// partials [nblk][2*n_hz+1] = column sums of the relu-filtered terms.
// ---------------------------------------------------------------------------
template <int NH>
__global__ __launch_bounds__(256) void unicycle_constraints_fwd_kernel(const float* ps, const float* ps_next,
                                                                       const float* V, const float* V_next,
                                                                       const float* hazards, float r2, float dt,
                                                                       float gamma_b, float gamma_l, int B,
                                                                       unsigned* ticket, const AuglagArgs A, float* sc,
                                                                       float* matr, float* bmatr, float* partials) {
    __shared__ float red[4 * (2 * NH + 1)];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[2 * NH + 1];
#pragma unroll
    for (int c = 0; c < 2 * NH + 1; ++c) v[c] = 0.f;
    if (i < B) {
        const float p0 = ps[i * 2], p1 = ps[i * 2 + 1];
        const float n0 = ps_next[i * 2], n1 = ps_next[i * 2 + 1];
        const float b0 = ps_next[(long)(B + i) * 2], b1 = ps_next[(long)(B + i) * 2 + 1];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float hx = hazards[h * 2], hy = hazards[h * 2 + 1];
            const float hs = 0.5f * (((p0 - hx) * (p0 - hx) + (p1 - hy) * (p1 - hy)) - r2);
            const float hn = 0.5f * (((n0 - hx) * (n0 - hx) + (n1 - hy) * (n1 - hy)) - r2);
            const float hb = 0.5f * (((b0 - hx) * (b0 - hx) + (b1 - hy) * (b1 - hy)) - r2);
            const float t = -((hn - hs) / dt) - gamma_b * hs;
            const float tb = -((hb - hs) / dt) - gamma_b * hs;
            matr[(long)i * (NH + 1) + h] = t;
            bmatr[(long)i * NH + h] = tb;
            v[h] = t > 0.f ? t : 0.f;
            v[NH + 1 + h] = tb > 0.f ? tb : 0.f;
        }
        const float vv = V[i];
        const float lya = ((V_next[i] - vv) / dt) + gamma_l * vv;
        matr[(long)i * (NH + 1) + NH] = lya;
        v[NH] = lya > 0.f ? lya : 0.f;
    }
    block_sum_256<2 * NH + 1>(v, red);
    CONSTRAINTS_TAIL(2 * NH + 1, v)
}

template <bool COHERENT>
__device__ __forceinline__ void auglag_body(const float* partials, int n_blk, const AuglagArgs A, float* sc_global) {
    __shared__ __attribute__((aligned(8))) float sc[NLBAC_SC_SIZE_ENUM];
    __shared__ float s_stage[128 * 36];
    auglag_body_at<COHERENT>(partials, n_blk, A, sc_global, sc, s_stage, 128 * 36);
}

__global__ void auglag_kernel(const float* partials, int n_blk, const AuglagArgs A, float* sc) {
    if (blockIdx.x != 0) return;
    auglag_body<false>(partials, n_blk, A, sc);
}


// d ps_next (2B,2) from the CBF terms and dV_next (B) from the CLF term
template <int NH>
__global__ __launch_bounds__(256) void unicycle_constraints_bwd_kernel(const float* ps_next, const float* matr,
                                                                       const float* bmatr, const float* hazards,
                                                                       float dt, float batch_size, int B,
                                                                       const float* sc, float* dps_next,
                                                                       float* dV_next) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float n0 = ps_next[i * 2], n1 = ps_next[i * 2 + 1];
    const float b0 = ps_next[(long)(B + i) * 2], b1 = ps_next[(long)(B + i) * 2 + 1];
    float d0 = 0.f, d1 = 0.f, e0 = 0.f, e1 = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float hx = hazards[h * 2], hy = hazards[h * 2 + 1];
        // d required_h / d term = 1/batch_size on the active set; term = -(hn - hs)/dt - ...
        if (matr[(long)i * (NH + 1) + h] > 0.f) {
            const float g = -((sc[SC_COEF + h] / batch_size) / dt);
            d0 += g * (n0 - hx); d1 += g * (n1 - hy);
        }
        if (bmatr[(long)i * NH + h] > 0.f) {
            const float g = -((sc[SC_BCOEF + h] / batch_size) / dt);
            e0 += g * (b0 - hx); e1 += g * (b1 - hy);
        }
    }
    dps_next[i * 2] = d0; dps_next[i * 2 + 1] = d1;
    dps_next[(long)(B + i) * 2] = e0; dps_next[(long)(B + i) * 2 + 1] = e1;
    dV_next[i] = (matr[(long)i * (NH + 1) + NH] > 0.f) ? ((sc[SC_COEF + NH] / batch_size) / dt) : 0.f;
}

// ---------------------------------------------------------------------------
// SimulatedCars (C = NLBAC_SimulatedCarsFollowing_RL_training/Simulated_Car_Following_RL_training)
//   get_state / get_obs         C/sac_cbf_clf/dynamics.py:59-62, 88-91
//   relative-degree-2 CBFs+CLF  C/sac_cbf_clf/sac_cbf_clf.py:474-511 (primary), 618-645 (backup)
// State = 5 x (position, velocity); h23 = x4 - x6 - r, h34 = x6 - x8 - r.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cars_state_kernel(const float* obs, int obs_ld, int n, float* state) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 10; ++c)      // the reference scales in float64 on the host, then casts
        state[(long)i * 10 + c] = (float)((double)obs[(long)i * obs_ld + c] * ((c & 1) ? 30.0 : 100.0));
}

// Everything the two-step rollout starts from, in one launch: state, its two copies (primary / backup rows of the
// first solve) and the carried inputs [action, time] of both steps (the second step's action column is filled later).
__global__ __launch_bounds__(256) void cars_rollout_inputs_kernel(const float* mb, int ld, int t_col, int nt_col,
                                                                  const float* pi2, int B, float* state, float* y0_2,
                                                                  float* c1, float* c2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        const float v = (float)((double)mb[(long)i * ld + c] * ((c & 1) ? 30.0 : 100.0));
        state[(long)i * 10 + c] = v;
        y0_2[(long)i * 10 + c] = v;
        y0_2[(long)(B + i) * 10 + c] = v;
    }
    const float t = mb[(long)i * ld + t_col], nt = mb[(long)i * ld + nt_col];
    c1[2 * i + 0] = pi2[i];            c1[2 * i + 1] = t;
    c1[2 * (B + i) + 0] = pi2[B + i];  c1[2 * (B + i) + 1] = t;
    c2[2 * i + 1] = nt;
    c2[2 * (B + i) + 1] = nt;
}

__global__ __launch_bounds__(256) void cars_obs_kernel(const float* state, int n, float* obs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 10; ++c) obs[(long)i * 10 + c] = state[(long)i * 10 + c] / ((c & 1) ? 30.0f : 100.0f);
}

__device__ __forceinline__ float cars_cbf(float h0, float h1, float h2, float gb) {
    const float l1 = h1 - h0 + gb * h0;
    const float l2 = h2 - h1 + gb * h1;
    return -(l2 - l1) - gb * l1;
}

// state (B,10); x1, x2 (2B,10): primary rows then backup rows.  matr (B,3) = [cbf23, cbf34, clf], bmatr (B,2).
__global__ __launch_bounds__(256) void cars_constraints_fwd_kernel(const float* state, const float* x1, const float* x2,
                                                                   const float* V, const float* V1, float gamma_b,
                                                                   float gamma_l, float radius, int B,
                                                                   unsigned* ticket, const AuglagArgs A, float* sc,
                                                                   float* matr, float* bmatr, float* partials) {
    __shared__ float red[20];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < B) {
        const float* s0 = state + (long)i * 10;
        const float h23_0 = (s0[4] - s0[6]) + (-radius), h34_0 = (s0[6] - s0[8]) + (-radius);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float* a = x1 + ((long)p * B + i) * 10;
            const float* b = x2 + ((long)p * B + i) * 10;
            const float c23 = cars_cbf(h23_0, (a[4] - a[6]) + (-radius), (b[4] - b[6]) + (-radius), gamma_b);
            const float c34 = cars_cbf(h34_0, (a[6] - a[8]) + (-radius), (b[6] - b[8]) + (-radius), gamma_b);
            if (p == 0) {
                matr[(long)i * 3 + 0] = c23; matr[(long)i * 3 + 1] = c34;
                v[0] = c23 > 0.f ? c23 : 0.f; v[1] = c34 > 0.f ? c34 : 0.f;
            } else {
                bmatr[(long)i * 2 + 0] = c23; bmatr[(long)i * 2 + 1] = c34;
                v[3] = c23 > 0.f ? c23 : 0.f; v[4] = c34 > 0.f ? c34 : 0.f;
            }
        }
        const float vv = V[i];
        const float lya = (V1[i] - vv) + gamma_l * vv;
        matr[(long)i * 3 + 2] = lya;
        v[2] = lya > 0.f ? lya : 0.f;
    }
    block_sum_256<5>(v, red);
    CONSTRAINTS_TAIL(5, v)
}

// dx1, dx2 (2B,10) and dV1 (B) from the loss coefficients in sc:  cbf = -h2 + 2(1-gb) h1 - (1-gb)^2 h0
__global__ __launch_bounds__(256) void cars_constraints_bwd_kernel(const float* matr, const float* bmatr, float gamma_b,
                                                                   float batch_size, int B, const float* sc,
                                                                   float* dx1, float* dx2, float* dV1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float* m = p == 0 ? matr + (long)i * 3 : bmatr + (long)i * 2;
        const float* coef = sc + (p == 0 ? SC_COEF : SC_BCOEF);
        const float g23 = m[0] > 0.f ? coef[0] / batch_size : 0.f;
        const float g34 = m[1] > 0.f ? coef[1] / batch_size : 0.f;
        // d cbf/d h2 = -1 ; d cbf/d h1 = -(-1 + gb) + (1 - gb)  (the two autograd paths through l2 and l1)
        const float c1 = -(-1.0f + gamma_b) + (1.0f - gamma_b);
        float* a = dx1 + ((long)p * B + i) * 10;
        float* b = dx2 + ((long)p * B + i) * 10;
#pragma unroll
        for (int c = 0; c < 10; ++c) { a[c] = 0.f; b[c] = 0.f; }
        a[4] = g23 * c1; a[6] = -(g23 * c1) + g34 * c1; a[8] = -(g34 * c1);
        b[4] = -g23; b[6] = g23 - g34; b[8] = g34;
    }
    dV1[i] = matr[(long)i * 3 + 2] > 0.f ? sc[SC_COEF + 2] / batch_size : 0.f;
}

// dst[row][col0 + c] += src[row][c]
__global__ __launch_bounds__(256) void add_cols_kernel(float* dst, int dst_ld, int col0, const float* src, int src_ld,
                                                       int ncols, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < ncols; ++c) dst[(long)i * dst_ld + col0 + c] += src[(long)i * src_ld + c];
}

// dst (n x ld) <- (dst + src on its column block, rows < n_src) + add: add_cols and the axpby behind it as one launch
__global__ __launch_bounds__(256) void add_cols_plus_kernel(float* dst, int ld, int col0, const float* src, int src_ld,
                                                            int ncols, int n_src, const float* add, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int row = (int)(i / ld), c = (int)(i - (long)row * ld);
    float v = dst[i];
    if (row < n_src && c >= col0 && c < col0 + ncols) v += src[(long)row * src_ld + (c - col0)];
    dst[i] = v + add[i];
}

// MSE (mean over n*d) partial sums and gradient
__global__ __launch_bounds__(256) void mse_kernel(const float* pred, int pred_ld, const float* target, int target_ld,
                                                  int n, int n_norm, int d, float* dpred, int dpred_ld,
                                                  float* partials) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[1] = {0.f};
    if (i < n) {
        const float norm = (float)(2.0 / ((double)n_norm * d));
        for (int c = 0; c < d; ++c) {
            const float e = pred[(long)i * pred_ld + c] - target[(long)i * target_ld + c];
            dpred[(long)i * dpred_ld + c] = norm * e;
            v[0] += e * e;
        }
    }
    block_sum_256<1>(v, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static int fuse_args(AuglagArgs& A, const nlbac_auglag_args* fused, unsigned*& ticket, float* sc, const char* who) {
    memset(&A, 0, sizeof(A));
    if (!fused) { ticket = nullptr; return 0; }
    NLBAC_REQUIRE(ticket && sc, "%s: the fused augmented-Lagrangian step needs a ticket word and the scalars block", who);
    NLBAC_REQUIRE(fused->backup_mode >= 0 && fused->backup_mode <= 2 && fused->n_cbf >= 1 &&
                      fused->n_cbf + fused->n_clf <= NLBAC_NC_MAX && fused->n_clf >= 0 && fused->n_clf <= 1,
                  "%s: bad augmented-Lagrangian arguments", who);
    A.n_cbf = fused->n_cbf; A.n_clf = fused->n_clf; A.batch_size = fused->batch_size;
    A.do_lambda_update = fused->do_lambda_update; A.do_backup_lambda_update = fused->do_backup_lambda_update;
    A.ratio_mode = fused->ratio_mode; A.backup_mode = fused->backup_mode; A.lam_lo = fused->lam_lo; A.lam_hi = fused->lam_hi;
    return 0;
}

#define GRID1(n) dim3(nlbac_ceil_div((n), 256)), dim3(256), 0, (hipStream_t)s

extern "C" int nlbac_gauss_sample_fwd(const float* heads, int heads_ld, const float* eps, const float* scale,
                                      const float* bias, int n_u, int n, float* action, int action_ld,
                                      float* logp, nlbac_stream_t s) {
    NLBAC_REQUIRE(heads && eps && scale && bias && action && logp, "nlbac_gauss_sample_fwd: null pointer");
    NLBAC_REQUIRE(n_u >= 1 && n_u <= MAX_NU && n >= 1, "nlbac_gauss_sample_fwd: bad sizes");
    hipLaunchKernelGGL(gauss_fwd_kernel, GRID1(n), heads, heads_ld, eps, scale, bias, n_u, n, action, action_ld, logp);
    NLBAC_CHECK_LAUNCH("nlbac_gauss_sample_fwd");
    return 0;
}

extern "C" int nlbac_gauss_sample_bwd(const float* heads, int heads_ld, const float* eps, const float* scale,
                                      int n_u, int n, int rows_per_problem, const float* da0, int da0_ld,
                                      const float* da1, int da1_ld, const float* da2, int da2_ld,
                                      const float* alpha, float dlogp_mul, float* dheads, int dheads_ld,
                                      nlbac_stream_t s) {
    NLBAC_REQUIRE(heads && eps && scale && alpha && dheads, "nlbac_gauss_sample_bwd: null pointer");
    NLBAC_REQUIRE(n_u >= 1 && n_u <= MAX_NU && n >= 1 && rows_per_problem >= 1, "nlbac_gauss_sample_bwd: bad sizes");
    hipLaunchKernelGGL(gauss_bwd_kernel, GRID1(n), heads, heads_ld, eps, scale, n_u, n, rows_per_problem, da0,
                       da0_ld, da1, da1_ld, da2, da2_ld, alpha, dlogp_mul, dheads, dheads_ld);
    NLBAC_CHECK_LAUNCH("nlbac_gauss_sample_bwd");
    return 0;
}

extern "C" int nlbac_td_targets(const float* q1t, const float* q2t, const float* lt, const float* nlogp,
                                const float* reward, const float* constraint, const float* mask, int rcm_ld,
                                const float* q1, const float* q2, const float* lf, const float* alpha,
                                float gamma, int B, int B_norm, float* dq1, float* dq2, float* dlf, float* next_q,
                                float* next_l, float* partials, unsigned* ticket, float mul, float* out,
                                nlbac_stream_t s) {
    NLBAC_REQUIRE(q1t && q2t && lt && nlogp && reward && constraint && mask && q1 && q2 && lf && alpha && dq1 &&
                      dq2 && dlf && partials, "nlbac_td_targets: null pointer");
    NLBAC_REQUIRE(!ticket || out, "nlbac_td_targets: the fused sums need an output");
    hipLaunchKernelGGL(td_targets_kernel, GRID1(B), q1t, q2t, lt, nlogp, reward, constraint, mask, rcm_ld, q1, q2, lf,
                       alpha, gamma, B, B_norm, dq1, dq2, dlf, next_q, next_l, partials, ticket, mul, out);
    NLBAC_CHECK_LAUNCH("nlbac_td_targets");
    return 0;
}

extern "C" int nlbac_actor_q_terms(const float* q1, const float* q2, const float* logp, const float* alpha,
                                   int B, int B_norm, int P, float* dq1, float* dq2, float* partials,
                                   const nlbac_actor_scalar_args* fused, unsigned* ticket, nlbac_stream_t s) {
    NLBAC_REQUIRE(q1 && q2 && logp && alpha && dq1 && dq2 && partials && P >= 1 && P <= 2, "nlbac_actor_q_terms: bad arguments");
    ActorScalarArgs F;
    memset(&F, 0, sizeof(F));
    if (fused) {
        NLBAC_REQUIRE(ticket && fused->sc, "nlbac_actor_q_terms: the fused scalars need a ticket word and the scalars block");
        F.target_entropy = fused->target_entropy; F.sc = fused->sc;
        for (int p = 0; p < P; ++p) {
            NLBAC_REQUIRE(fused->log_alpha[p] && fused->g_log_alpha[p], "nlbac_actor_q_terms: missing log_alpha pointers");
            F.log_alpha[p] = fused->log_alpha[p]; F.g_log_alpha[p] = fused->g_log_alpha[p];
        }
    } else {
        ticket = nullptr;
    }
    hipLaunchKernelGGL(actor_q_terms_kernel, dim3(nlbac_ceil_div(B, 256), P), dim3(256), 0, (hipStream_t)s, q1, q2,
                       logp, alpha, B, B_norm, dq1, dq2, partials, ticket, F);
    NLBAC_CHECK_LAUNCH("nlbac_actor_q_terms");
    return 0;
}

extern "C" int nlbac_actor_scalars(const float* partials, int n_blk, int B, int first_problem, int P,
                                   float target_entropy, const float* log_alpha, int log_alpha_stride,
                                   float* g_log_alpha, float* sc, nlbac_stream_t s) {
    NLBAC_REQUIRE(partials && log_alpha && g_log_alpha && sc && P >= 1 && first_problem >= 0 && first_problem + P <= 2,
                  "nlbac_actor_scalars: bad arguments");
    hipLaunchKernelGGL(actor_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, partials, n_blk, B, first_problem, P,
                       target_entropy, log_alpha, log_alpha_stride, g_log_alpha, sc);
    NLBAC_CHECK_LAUNCH("nlbac_actor_scalars");
    return 0;
}

extern "C" int nlbac_alpha_refresh(const float* log_alpha, int log_alpha_stride, int first_problem, int P, float* sc,
                                   nlbac_stream_t s) {
    NLBAC_REQUIRE(log_alpha && sc && P >= 1 && first_problem >= 0 && first_problem + P <= 2,
                  "nlbac_alpha_refresh: bad arguments");
    hipLaunchKernelGGL(alpha_refresh_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, log_alpha, log_alpha_stride,
                       first_problem, P, sc);
    NLBAC_CHECK_LAUNCH("nlbac_alpha_refresh");
    return 0;
}

extern "C" int nlbac_unicycle_state(const float* obs, int obs_ld, int B, float l_p, float* state, int n_copies,
                                    float* ps, nlbac_stream_t s) {
    NLBAC_REQUIRE(obs && state && n_copies >= 1, "nlbac_unicycle_state: bad arguments");
    hipLaunchKernelGGL(unicycle_state_kernel, GRID1(B), obs, obs_ld, B, l_p, state, n_copies, ps);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_state");
    return 0;
}

extern "C" int nlbac_unicycle_lookahead(const float* x, int n, float l_p, float* ps, nlbac_stream_t s) {
    NLBAC_REQUIRE(x && ps, "nlbac_unicycle_lookahead: null pointer");
    hipLaunchKernelGGL(unicycle_lookahead_kernel, GRID1(n), x, n, l_p, ps);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_lookahead");
    return 0;
}

extern "C" int nlbac_unicycle_lookahead_bwd(const float* x, const float* dps, const float* dps2, int n, float l_p,
                                            float* dx, nlbac_stream_t s) {
    NLBAC_REQUIRE(x && dps && dx, "nlbac_unicycle_lookahead_bwd: null pointer");
    hipLaunchKernelGGL(unicycle_lookahead_bwd_kernel, GRID1(n), x, dps, dps2, n, l_p, dx);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_lookahead_bwd");
    return 0;
}

extern "C" int nlbac_unicycle_constraints_fwd(const float* ps, const float* ps_next, const float* V,
                                              const float* V_next, const float* hazards, int n_hz, float r_coll,
                                              float dt, float gamma_b, float gamma_l, int B, float* matr,
                                              float* bmatr, float* partials, const nlbac_auglag_args* fused,
                                              unsigned* ticket, float* sc, nlbac_stream_t s) {
    NLBAC_REQUIRE(ps && ps_next && V && V_next && hazards && matr && bmatr && partials,
                  "nlbac_unicycle_constraints_fwd: null pointer");
    NLBAC_REQUIRE(n_hz == 7, "nlbac_unicycle_constraints_fwd: built for n_hz == 7 (got %d)", n_hz);
    const float r2 = (float)((double)r_coll * (double)r_coll);
    AuglagArgs A;
    if (fuse_args(A, fused, ticket, sc, "nlbac_unicycle_constraints_fwd")) return -1;
    hipLaunchKernelGGL(unicycle_constraints_fwd_kernel<7>, GRID1(B), ps, ps_next, V, V_next, hazards, r2, dt,
                       gamma_b, gamma_l, B, ticket, A, sc, matr, bmatr, partials);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_constraints_fwd");
    return 0;
}

extern "C" int nlbac_auglag(const float* partials, int n_blk, int n_cbf, int n_clf, float batch_size,
                            int do_lambda_update, int do_backup_lambda_update, int ratio_mode, int backup_mode,
                            float lam_lo, float lam_hi, float* sc, nlbac_stream_t s) {
    NLBAC_REQUIRE(partials && sc, "nlbac_auglag: null pointer");
    NLBAC_REQUIRE(backup_mode >= 0 && backup_mode <= 2, "nlbac_auglag: backup_mode is 0, 1 or 2");
    NLBAC_REQUIRE(n_cbf >= 1 && n_cbf + n_clf <= NLBAC_NC_MAX && n_clf >= 0 && n_clf <= 1, "nlbac_auglag: bad constraint counts");
    AuglagArgs A = {n_cbf, n_clf, batch_size, do_lambda_update, do_backup_lambda_update, ratio_mode, backup_mode, lam_lo, lam_hi};
    hipLaunchKernelGGL(auglag_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, partials, n_blk, A, sc);
    NLBAC_CHECK_LAUNCH("nlbac_auglag");
    return 0;
}

extern "C" int nlbac_unicycle_constraints_bwd(const float* ps_next, const float* matr, const float* bmatr,
                                              const float* hazards, int n_hz, float dt, float batch_size, int B,
                                              const float* sc, float* dps_next, float* dV_next, nlbac_stream_t s) {
    NLBAC_REQUIRE(ps_next && matr && bmatr && hazards && sc && dps_next && dV_next,
                  "nlbac_unicycle_constraints_bwd: null pointer");
    NLBAC_REQUIRE(n_hz == 7, "nlbac_unicycle_constraints_bwd: built for n_hz == 7 (got %d)", n_hz);
    hipLaunchKernelGGL(unicycle_constraints_bwd_kernel<7>, GRID1(B), ps_next, matr, bmatr, hazards, dt, batch_size,
                       B, sc, dps_next, dV_next);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_constraints_bwd");
    return 0;
}

extern "C" int nlbac_mse_fwd_bwd(const float* pred, int pred_ld, const float* target, int target_ld, int n,
                                 int n_norm, int d, float* dpred, int dpred_ld, float* partials, nlbac_stream_t s) {
    NLBAC_REQUIRE(pred && target && dpred && partials, "nlbac_mse_fwd_bwd: null pointer");
    hipLaunchKernelGGL(mse_kernel, GRID1(n), pred, pred_ld, target, target_ld, n, n_norm, d, dpred, dpred_ld, partials);
    NLBAC_CHECK_LAUNCH("nlbac_mse_fwd_bwd");
    return 0;
}

extern "C" int nlbac_cars_state(const float* obs, int obs_ld, int n, float* state, nlbac_stream_t s) {
    NLBAC_REQUIRE(obs && state, "nlbac_cars_state: null pointer");
    hipLaunchKernelGGL(cars_state_kernel, GRID1(n), obs, obs_ld, n, state);
    NLBAC_CHECK_LAUNCH("nlbac_cars_state");
    return 0;
}

extern "C" int nlbac_cars_rollout_inputs(const float* mb, int ld, int t_col, int nt_col, const float* pi2, int B,
                                         float* state, float* y0_2, float* c1, float* c2, nlbac_stream_t s) {
    NLBAC_REQUIRE(mb && pi2 && state && y0_2 && c1 && c2 && B >= 1 && t_col >= 0 && nt_col >= 0 && t_col < ld &&
                      nt_col < ld && ld >= 10, "nlbac_cars_rollout_inputs: bad arguments");
    hipLaunchKernelGGL(cars_rollout_inputs_kernel, GRID1(B), mb, ld, t_col, nt_col, pi2, B, state, y0_2, c1, c2);
    NLBAC_CHECK_LAUNCH("nlbac_cars_rollout_inputs");
    return 0;
}

extern "C" int nlbac_cars_obs(const float* state, int n, float* obs, nlbac_stream_t s) {
    NLBAC_REQUIRE(obs && state, "nlbac_cars_obs: null pointer");
    hipLaunchKernelGGL(cars_obs_kernel, GRID1(n), state, n, obs);
    NLBAC_CHECK_LAUNCH("nlbac_cars_obs");
    return 0;
}

extern "C" int nlbac_cars_constraints_fwd(const float* state, const float* x1, const float* x2, const float* V,
                                          const float* V1, float gamma_b, float gamma_l, float radius, int B,
                                          float* matr, float* bmatr, float* partials, const nlbac_auglag_args* fused,
                                          unsigned* ticket, float* sc, nlbac_stream_t s) {
    NLBAC_REQUIRE(state && x1 && x2 && V && V1 && matr && bmatr && partials, "nlbac_cars_constraints_fwd: null pointer");
    AuglagArgs A;
    if (fuse_args(A, fused, ticket, sc, "nlbac_cars_constraints_fwd")) return -1;
    hipLaunchKernelGGL(cars_constraints_fwd_kernel, GRID1(B), state, x1, x2, V, V1, gamma_b, gamma_l, radius, B, ticket,
                       A, sc, matr, bmatr, partials);
    NLBAC_CHECK_LAUNCH("nlbac_cars_constraints_fwd");
    return 0;
}

extern "C" int nlbac_cars_constraints_bwd(const float* matr, const float* bmatr, float gamma_b, float batch_size,
                                          int B, const float* sc, float* dx1, float* dx2, float* dV1,
                                          nlbac_stream_t s) {
    NLBAC_REQUIRE(matr && bmatr && sc && dx1 && dx2 && dV1, "nlbac_cars_constraints_bwd: null pointer");
    hipLaunchKernelGGL(cars_constraints_bwd_kernel, GRID1(B), matr, bmatr, gamma_b, batch_size, B, sc, dx1, dx2, dV1);
    NLBAC_CHECK_LAUNCH("nlbac_cars_constraints_bwd");
    return 0;
}

extern "C" int nlbac_add_cols(float* dst, int dst_ld, int col0, const float* src, int src_ld, int ncols, int n,
                              nlbac_stream_t s) {
    NLBAC_REQUIRE(dst && src && ncols >= 1, "nlbac_add_cols: bad arguments");
    hipLaunchKernelGGL(add_cols_kernel, GRID1(n), dst, dst_ld, col0, src, src_ld, ncols, n);
    NLBAC_CHECK_LAUNCH("nlbac_add_cols");
    return 0;
}

extern "C" int nlbac_add_cols_plus(float* dst, int ld, int col0, const float* src, int src_ld, int ncols, int n_src,
                                   const float* add, int n, nlbac_stream_t s) {
    NLBAC_REQUIRE(dst && src && add && ncols >= 1 && col0 >= 0 && col0 + ncols <= ld && n_src <= n,
                  "nlbac_add_cols_plus: bad arguments");
    const long total = (long)n * ld;
    hipLaunchKernelGGL(add_cols_plus_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, dst, ld,
                       col0, src, src_ld, ncols, n_src, add, total);
    NLBAC_CHECK_LAUNCH("nlbac_add_cols_plus");
    return 0;
}


// ---------------------------------------------------------------------------
// Learned-barrier-certificate copies (NU = neural_barrier_certificate/.../Unicycle_RL_training)
//   barrier TD target + MSE     NU/sac_cbf_clf/sac_cbf_clf.py:224-233
//   get_obs (differentiable)    NU/sac_cbf_clf/dynamics.py:92-135
//   learned CBF + CLF terms     NU/sac_cbf_clf/sac_cbf_clf.py:399-420, 430-440
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_value_kernel(const float* next_target, const float* signal, int sig_ld,
                                                       const float* mask, int mask_ld, const float* pred, float gamma,
                                                       int B, int B_norm, float* dpred, float* next_out,
                                                       float* partials, unsigned* ticket, float mul, float* out) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[1] = {0.f};
    if (i < B) {
        const float y = signal[(long)i * sig_ld] + mask[(long)i * mask_ld] * gamma * next_target[i];
        const float e = pred[i] - y;
        dpred[i] = (float)(2.0 / (double)B_norm) * e;
        if (next_out) next_out[i] = y;
        v[0] = e * e;
    }
    block_sum_256<1>(v, red);
    if (!ticket) {
        if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
        return;
    }
    if (!publish_and_elect<1>(partials + blockIdx.x, v, ticket, gridDim.x)) return;
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (unsigned b = 0; b < gridDim.x; ++b) s += coherent_load(partials + b);
        out[0] = s * mul;
    }
}

__global__ __launch_bounds__(256) void unicycle_obs_fwd_kernel(const float* x, int n, float gx, float gy, float* obs,
                                                               int obs_ld) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = x[i * 3], py = x[i * 3 + 1], th = x[i * 3 + 2];
    const float c = cosf(th), s = sinf(th);
    const float rx = gx - px, ry = gy - py;
    const float dist = sqrtf(rx * rx + ry * ry);
    const float v0 = c * rx + s * ry, v1 = -s * rx + c * ry;
    const float div = sqrtf(v0 * v0 + v1 * v1) + 0.001f;
    float* o = obs + (long)i * obs_ld;
    o[0] = px; o[1] = py; o[2] = c; o[3] = s; o[4] = v0 / div; o[5] = v1 / div; o[6] = expf(-dist);
}

__global__ __launch_bounds__(256) void unicycle_obs_bwd_kernel(const float* x, const float* dobs, int dobs_ld, int n,
                                                               float gx, float gy, float* dx, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = x[i * 3], py = x[i * 3 + 1], th = x[i * 3 + 2];
    const float c = cosf(th), s = sinf(th);
    const float rx = gx - px, ry = gy - py;
    const float dist = sqrtf(rx * rx + ry * ry);
    const float v0 = c * rx + s * ry, v1 = -s * rx + c * ry;
    const float nv = sqrtf(v0 * v0 + v1 * v1), div = nv + 0.001f;
    const float* d = dobs + (long)i * dobs_ld;
    float gpx = d[0], gpy = d[1], gth = -s * d[2] + c * d[3];
    float dv0 = d[4] / div, dv1 = d[5] / div;
    const float dnv = -(d[4] * v0 + d[5] * v1) / (div * div);
    if (nv > 0.f) { dv0 += dnv * v0 / nv; dv1 += dnv * v1 / nv; }
    float drx = dv0 * c - dv1 * s, dry = dv0 * s + dv1 * c;
    gth += dv0 * v1 - dv1 * v0;
    if (dist > 0.f) {
        const float dd = -d[6] * expf(-dist);
        drx += dd * rx / dist; dry += dd * ry / dist;
    }
    gpx -= drx; gpy -= dry;
    if (accumulate) { dx[i * 3] += gpx; dx[i * 3 + 1] += gpy; dx[i * 3 + 2] += gth; }
    else { dx[i * 3] = gpx; dx[i * 3 + 1] = gpy; dx[i * 3 + 2] = gth; }
}

// matr (B,2) = [-(B' - B) - gamma_b B, (V' - V)/dt + gamma_l V]; partials [nblk][2]
__global__ __launch_bounds__(256) void barrier_constraints_fwd_kernel(const float* Bv, const float* Bn, const float* V,
                                                                      const float* Vn, float dt, float gamma_b,
                                                                      float gamma_l, int B, unsigned* ticket,
                                                                      const AuglagArgs A, float* sc, float* matr,
                                                                      float* partials) {
    __shared__ float red[8];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[2] = {0.f, 0.f};
    if (i < B) {
        const float b = Bv[i], vv = V[i];
        const float bt = -(Bn[i] - b) - gamma_b * b;
        const float lt = ((Vn[i] - vv) / dt) + gamma_l * vv;
        matr[i * 2] = bt; matr[i * 2 + 1] = lt;
        v[0] = bt > 0.f ? bt : 0.f;
        v[1] = lt > 0.f ? lt : 0.f;
    }
    block_sum_256<2>(v, red);
    CONSTRAINTS_TAIL(2, v)
}

__global__ __launch_bounds__(256) void barrier_constraints_bwd_kernel(const float* matr, float dt, float batch_size,
                                                                      int B, const float* sc, float* dBn, float* dVn) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    dBn[i] = matr[i * 2] > 0.f ? -(sc[SC_COEF] / batch_size) : 0.f;
    dVn[i] = matr[i * 2 + 1] > 0.f ? ((sc[SC_COEF + 1] / batch_size) / dt) : 0.f;
}

extern "C" int nlbac_td_value(const float* next_target, const float* signal, int sig_ld, const float* mask,
                              int mask_ld, const float* pred, float gamma, int B, int B_norm, float* dpred,
                              float* next_out, float* partials, unsigned* ticket, float mul, float* out,
                              nlbac_stream_t s) {
    NLBAC_REQUIRE(next_target && signal && mask && pred && dpred && partials, "nlbac_td_value: null pointer");
    NLBAC_REQUIRE(!ticket || out, "nlbac_td_value: the fused sum needs an output");
    hipLaunchKernelGGL(td_value_kernel, GRID1(B), next_target, signal, sig_ld, mask, mask_ld, pred, gamma, B, B_norm,
                       dpred, next_out, partials, ticket, mul, out);
    NLBAC_CHECK_LAUNCH("nlbac_td_value");
    return 0;
}

extern "C" int nlbac_unicycle_obs_fwd(const float* x, int n, float goal_x, float goal_y, float* obs, int obs_ld,
                                      nlbac_stream_t s) {
    NLBAC_REQUIRE(x && obs && obs_ld >= 7, "nlbac_unicycle_obs_fwd: bad arguments");
    hipLaunchKernelGGL(unicycle_obs_fwd_kernel, GRID1(n), x, n, goal_x, goal_y, obs, obs_ld);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_obs_fwd");
    return 0;
}

extern "C" int nlbac_unicycle_obs_bwd(const float* x, const float* dobs, int dobs_ld, int n, float goal_x,
                                      float goal_y, float* dx, int accumulate, nlbac_stream_t s) {
    NLBAC_REQUIRE(x && dobs && dx && dobs_ld >= 7, "nlbac_unicycle_obs_bwd: bad arguments");
    hipLaunchKernelGGL(unicycle_obs_bwd_kernel, GRID1(n), x, dobs, dobs_ld, n, goal_x, goal_y, dx, accumulate);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_obs_bwd");
    return 0;
}

extern "C" int nlbac_barrier_constraints_fwd(const float* Bv, const float* Bn, const float* V, const float* Vn,
                                             float dt, float gamma_b, float gamma_l, int B, float* matr,
                                             float* partials, const nlbac_auglag_args* fused, unsigned* ticket,
                                             float* sc, nlbac_stream_t s) {
    NLBAC_REQUIRE(Bv && Bn && V && Vn && matr && partials, "nlbac_barrier_constraints_fwd: null pointer");
    AuglagArgs A;
    if (fuse_args(A, fused, ticket, sc, "nlbac_barrier_constraints_fwd")) return -1;
    hipLaunchKernelGGL(barrier_constraints_fwd_kernel, GRID1(B), Bv, Bn, V, Vn, dt, gamma_b, gamma_l, B, ticket, A, sc,
                       matr, partials);
    NLBAC_CHECK_LAUNCH("nlbac_barrier_constraints_fwd");
    return 0;
}

extern "C" int nlbac_barrier_constraints_bwd(const float* matr, float dt, float batch_size, int B, const float* sc,
                                             float* dBn, float* dVn, nlbac_stream_t s) {
    NLBAC_REQUIRE(matr && sc && dBn && dVn, "nlbac_barrier_constraints_bwd: null pointer");
    hipLaunchKernelGGL(barrier_constraints_bwd_kernel, GRID1(B), matr, dt, batch_size, B, sc, dBn, dVn);
    NLBAC_CHECK_LAUNCH("nlbac_barrier_constraints_bwd");
    return 0;
}


// ---------------------------------------------------------------------------
// Pvtol (P = NLBAC_pvtol_RL_training/Pvtol_RL_training)
//   get_state / get_obs                 P/sac_cbf_clf/dynamics.py:50-66, 97-153
//   safety operator follow              P/sac_cbf_clf/sac_cbf_clf.py:462-470
//   relative-degree-3 CBFs + CLF        P/sac_cbf_clf/sac_cbf_clf.py:543-690 (primary), 880-1010 (backup)
// Dynamic state x = [x, y, theta, vx, vy, thrust]; op = x-position of the safety operator.
// Rows of x1/x2/x3: NP problems (primary, backup) x B.
// ---------------------------------------------------------------------------
#define PV_NH 5
#define PV_NC (PV_NH + 4)

__global__ __launch_bounds__(256) void pvtol_state_kernel(const float* obs, int obs_ld, int n, float* st6, float* op) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* o = obs + (long)i * obs_ld;
    float* x = st6 + (long)i * 6;
    x[0] = o[0]; x[1] = o[1];
    x[2] = (float)atan2((double)o[3], (double)o[2]);      // float64 on the host in the reference, then cast
    x[3] = o[4]; x[4] = o[5]; x[5] = o[6];
    if (op) op[i] = o[7];
}

__global__ __launch_bounds__(256) void pvtol_obs_fwd_kernel(const float* x6, const float* op_prev, int op_rows,
                                                            float follow, float gx, float gy, int n, float* obs,
                                                            int obs_ld, float* op_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* x = x6 + (long)i * 6;
    const float o0 = op_prev[i % op_rows];
    const float o1 = o0 + follow * (x[0] - o0);
    const float c = cosf(x[2]), s = sinf(x[2]);
    const float rx = gx - x[0], ry = gy - x[1];
    const float dist = sqrtf(rx * rx + ry * ry);
    const float v0 = c * rx + s * ry, v1 = -s * rx + c * ry;
    const float div = sqrtf(v0 * v0 + v1 * v1) + 0.001f;
    float* o = obs + (long)i * obs_ld;
    o[0] = x[0]; o[1] = x[1]; o[2] = c; o[3] = s; o[4] = x[3]; o[5] = x[4]; o[6] = x[5]; o[7] = o1;
    o[8] = v0 / div; o[9] = v1 / div; o[10] = expf(-dist);
    if (op_out) op_out[i] = o1;
}

__global__ __launch_bounds__(256) void pvtol_obs_bwd_kernel(const float* x6, const float* dobs, int dobs_ld,
                                                            float follow, float gx, float gy, int n, float* dx,
                                                            int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* x = x6 + (long)i * 6;
    const float c = cosf(x[2]), s = sinf(x[2]);
    const float rx = gx - x[0], ry = gy - x[1];
    const float dist = sqrtf(rx * rx + ry * ry);
    const float v0 = c * rx + s * ry, v1 = -s * rx + c * ry;
    const float nv = sqrtf(v0 * v0 + v1 * v1), div = nv + 0.001f;
    const float* d = dobs + (long)i * dobs_ld;
    float g[6];
    g[0] = d[0] + follow * d[7];          // the operator's new position follows x
    g[1] = d[1];
    g[2] = -s * d[2] + c * d[3];
    g[3] = d[4]; g[4] = d[5]; g[5] = d[6];
    float dv0 = d[8] / div, dv1 = d[9] / div;
    const float dnv = -(d[8] * v0 + d[9] * v1) / (div * div);
    if (nv > 0.f) { dv0 += dnv * v0 / nv; dv1 += dnv * v1 / nv; }
    float drx = dv0 * c - dv1 * s, dry = dv0 * s + dv1 * c;
    g[2] += dv0 * v1 - dv1 * v0;
    if (dist > 0.f) {
        const float dd = -d[10] * expf(-dist);
        drx += dd * rx / dist; dry += dd * ry / dist;
    }
    g[0] -= drx; g[1] -= dry;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (accumulate) dx[(long)i * 6 + k] += g[k]; else dx[(long)i * 6 + k] = g[k];
    }
}

// the reference's operation order (P:575-587)
__device__ __forceinline__ float pv_rd3(float h0, float h1, float h2, float h3, float gb) {
    const float t1 = h3 - h2 + gb * h2;
    const float t2 = h2 - h1 + gb * h1;
    const float t3 = h1 - h0 + gb * h0;
    return -(t1 - t2 + gb * t2 - (t2 - t3 + gb * t3) + gb * (t2 - t3 + gb * t3));
}

__global__ __launch_bounds__(256) void pvtol_constraints_fwd_kernel(
    const float* st6, const float* op0, const float* x1, const float* x2, const float* x3, const float* V,
    const float* V1, const float* hazards, float r2, float d_op, float y_max, float y_min, float follow, float gb,
    float gl, int B, int NP, unsigned* ticket, const AuglagArgs A, float* sc, float* matr, float* bmatr,
    float* partials) {
    __shared__ float red[4 * (2 * PV_NC + 1)];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[2 * PV_NC + 1];
#pragma unroll
    for (int k = 0; k < 2 * PV_NC + 1; ++k) v[k] = 0.f;
    if (i < B) {
        const float px0 = st6[(long)i * 6], py0 = st6[(long)i * 6 + 1], o0 = op0[i];
        for (int p = 0; p < NP; ++p) {
            const long r = (long)p * B + i;
            const float px1 = x1[r * 6], py1 = x1[r * 6 + 1];
            const float px2 = x2[r * 6], py2 = x2[r * 6 + 1];
            const float px3 = x3[r * 6], py3 = x3[r * 6 + 1];
            const float o1 = o0 + follow * (px1 - o0);
            const float o2 = o1 + follow * (px2 - o1);
            const float o3 = o2 + follow * (px3 - o2);
            float t[PV_NC];
#pragma unroll
            for (int h = 0; h < PV_NH; ++h) {
                const float hx = hazards[h * 2], hy = hazards[h * 2 + 1];
                const float a0 = 0.5f * ((px0 - hx) * (px0 - hx) + (py0 - hy) * (py0 - hy) - r2);
                const float a1 = 0.5f * ((px1 - hx) * (px1 - hx) + (py1 - hy) * (py1 - hy) - r2);
                const float a2 = 0.5f * ((px2 - hx) * (px2 - hx) + (py2 - hy) * (py2 - hy) - r2);
                const float a3 = 0.5f * ((px3 - hx) * (px3 - hx) + (py3 - hy) * (py3 - hy) - r2);
                t[h] = pv_rd3(a0, a1, a2, a3, gb);
            }
            t[PV_NH + 0] = pv_rd3(px0 - o0 + d_op, px1 - o1 + d_op, px2 - o2 + d_op, px3 - o3 + d_op, gb);
            t[PV_NH + 1] = pv_rd3(-px0 + o0 + d_op, -px1 + o1 + d_op, -px2 + o2 + d_op, -px3 + o3 + d_op, gb);
            t[PV_NH + 2] = pv_rd3(-py0 + y_max - 10.0f, -py1 + y_max - 10.0f, -py2 + y_max - 10.0f,
                                  -py3 + y_max - 10.0f, gb);
            t[PV_NH + 3] = pv_rd3(py0 - y_min - 10.0f, py1 - y_min - 10.0f, py2 - y_min - 10.0f,
                                  py3 - y_min - 10.0f, gb);
            if (p == 0) {
                const float vv = V[i];
                const float lt = ((V1[i] - vv) / 1.0f) + gl * vv;
#pragma unroll
                for (int k = 0; k < PV_NC; ++k) { matr[(long)i * (PV_NC + 1) + k] = t[k]; v[k] = t[k] > 0.f ? t[k] : 0.f; }
                matr[(long)i * (PV_NC + 1) + PV_NC] = lt;
                v[PV_NC] = lt > 0.f ? lt : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < PV_NC; ++k) { bmatr[(long)i * PV_NC + k] = t[k]; v[PV_NC + 1 + k] = t[k] > 0.f ? t[k] : 0.f; }
            }
        }
    }
    block_sum_256<2 * PV_NC + 1>(v, red);
    const int ncol = PV_NC + 1 + (NP == 2 ? PV_NC : 0);
    CONSTRAINTS_TAIL(ncol, v)
}

__global__ __launch_bounds__(256) void pvtol_constraints_bwd_kernel(
    const float* matr, const float* bmatr, const float* x1, const float* x2, const float* x3, const float* hazards,
    float follow, float gb, float batch_size, int B, int NP, const float* sc, float* dx1, float* dx2, float* dx3,
    float* dV1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float a = gb - 1.0f;
    const float ck[3] = {-3.0f * a * a, -3.0f * a, -1.0f};      // d term / d h(t+1), h(t+2), h(t+3)
    for (int p = 0; p < NP; ++p) {
        const long r = (long)p * B + i;
        const float* xs[3] = {x1 + r * 6, x2 + r * 6, x3 + r * 6};
        const float* m = p == 0 ? matr + (long)i * (PV_NC + 1) : bmatr + (long)i * PV_NC;
        const float* coef = sc + (p == 0 ? SC_COEF : SC_BCOEF);
        float gx[3] = {0.f, 0.f, 0.f}, gy[3] = {0.f, 0.f, 0.f}, gop[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < PV_NH; ++h) {
            if (m[h] > 0.f) {
                const float g = coef[h] / batch_size;
                const float hx = hazards[h * 2], hy = hazards[h * 2 + 1];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    gx[k] += g * ck[k] * (xs[k][0] - hx);
                    gy[k] += g * ck[k] * (xs[k][1] - hy);
                }
            }
        }
        if (m[PV_NH + 0] > 0.f) {                                 // x - op + d
            const float g = coef[PV_NH + 0] / batch_size;
#pragma unroll
            for (int k = 0; k < 3; ++k) { gx[k] += g * ck[k]; gop[k] -= g * ck[k]; }
        }
        if (m[PV_NH + 1] > 0.f) {                                 // -x + op + d
            const float g = coef[PV_NH + 1] / batch_size;
#pragma unroll
            for (int k = 0; k < 3; ++k) { gx[k] -= g * ck[k]; gop[k] += g * ck[k]; }
        }
        if (m[PV_NH + 2] > 0.f) {                                 // -y + y_max - 10
            const float g = coef[PV_NH + 2] / batch_size;
#pragma unroll
            for (int k = 0; k < 3; ++k) gy[k] -= g * ck[k];
        }
        if (m[PV_NH + 3] > 0.f) {                                 // y - y_min - 10
            const float g = coef[PV_NH + 3] / batch_size;
#pragma unroll
            for (int k = 0; k < 3; ++k) gy[k] += g * ck[k];
        }
        // op_k = op_(k-1) + follow (x_k - op_(k-1))
        gx[2] += follow * gop[2]; gop[1] += (1.0f - follow) * gop[2];
        gx[1] += follow * gop[1]; gop[0] += (1.0f - follow) * gop[1];
        gx[0] += follow * gop[0];
        float* outs[3] = {dx1 + r * 6, dx2 + r * 6, dx3 + r * 6};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            outs[k][0] = gx[k]; outs[k][1] = gy[k];
            outs[k][2] = 0.f; outs[k][3] = 0.f; outs[k][4] = 0.f; outs[k][5] = 0.f;
        }
        if (p == 0) dV1[i] = m[PV_NC] > 0.f ? ((sc[SC_COEF + PV_NC] / batch_size) / 1.0f) : 0.f;
    }
}

extern "C" int nlbac_pvtol_state(const float* obs, int obs_ld, int n, float* st6, float* op, nlbac_stream_t s) {
    NLBAC_REQUIRE(obs && st6 && obs_ld >= 8, "nlbac_pvtol_state: bad arguments");
    hipLaunchKernelGGL(pvtol_state_kernel, GRID1(n), obs, obs_ld, n, st6, op);
    NLBAC_CHECK_LAUNCH("nlbac_pvtol_state");
    return 0;
}

extern "C" int nlbac_pvtol_obs_fwd(const float* x6, const float* op_prev, int op_rows, float follow, float goal_x,
                                   float goal_y, int n, float* obs, int obs_ld, float* op_out, nlbac_stream_t s) {
    NLBAC_REQUIRE(x6 && op_prev && obs && obs_ld >= 11 && op_rows >= 1, "nlbac_pvtol_obs_fwd: bad arguments");
    hipLaunchKernelGGL(pvtol_obs_fwd_kernel, GRID1(n), x6, op_prev, op_rows, follow, goal_x, goal_y, n, obs, obs_ld,
                       op_out);
    NLBAC_CHECK_LAUNCH("nlbac_pvtol_obs_fwd");
    return 0;
}

extern "C" int nlbac_pvtol_obs_bwd(const float* x6, const float* dobs, int dobs_ld, float follow, float goal_x,
                                   float goal_y, int n, float* dx, int accumulate, nlbac_stream_t s) {
    NLBAC_REQUIRE(x6 && dobs && dx && dobs_ld >= 11, "nlbac_pvtol_obs_bwd: bad arguments");
    hipLaunchKernelGGL(pvtol_obs_bwd_kernel, GRID1(n), x6, dobs, dobs_ld, follow, goal_x, goal_y, n, dx, accumulate);
    NLBAC_CHECK_LAUNCH("nlbac_pvtol_obs_bwd");
    return 0;
}

extern "C" int nlbac_pvtol_constraints_fwd(const float* st6, const float* op0, const float* x1, const float* x2,
                                           const float* x3, const float* V, const float* V1, const float* hazards,
                                           int n_hz, float r_coll, float d_op, float y_max, float y_min, float follow,
                                           float gamma_b, float gamma_l, int B, int NP, float* matr, float* bmatr,
                                           float* partials, const nlbac_auglag_args* fused, unsigned* ticket,
                                           float* sc, nlbac_stream_t s) {
    NLBAC_REQUIRE(st6 && op0 && x1 && x2 && x3 && V && V1 && hazards && matr && partials && (NP == 1 || bmatr),
                  "nlbac_pvtol_constraints_fwd: null pointer");
    NLBAC_REQUIRE(n_hz == PV_NH && (NP == 1 || NP == 2), "nlbac_pvtol_constraints_fwd: built for 5 hazards, 1-2 problems");
    const float r2 = (float)((double)r_coll * (double)r_coll);
    AuglagArgs A;
    if (fuse_args(A, fused, ticket, sc, "nlbac_pvtol_constraints_fwd")) return -1;
    hipLaunchKernelGGL(pvtol_constraints_fwd_kernel, GRID1(B), st6, op0, x1, x2, x3, V, V1, hazards, r2, d_op, y_max,
                       y_min, follow, gamma_b, gamma_l, B, NP, ticket, A, sc, matr, bmatr, partials);
    NLBAC_CHECK_LAUNCH("nlbac_pvtol_constraints_fwd");
    return 0;
}

extern "C" int nlbac_pvtol_constraints_bwd(const float* matr, const float* bmatr, const float* x1, const float* x2,
                                           const float* x3, const float* hazards, int n_hz, float follow,
                                           float gamma_b, float batch_size, int B, int NP, const float* sc, float* dx1,
                                           float* dx2, float* dx3, float* dV1, nlbac_stream_t s) {
    NLBAC_REQUIRE(matr && x1 && x2 && x3 && hazards && sc && dx1 && dx2 && dx3 && dV1 && (NP == 1 || bmatr),
                  "nlbac_pvtol_constraints_bwd: null pointer");
    NLBAC_REQUIRE(n_hz == PV_NH && (NP == 1 || NP == 2), "nlbac_pvtol_constraints_bwd: built for 5 hazards, 1-2 problems");
    hipLaunchKernelGGL(pvtol_constraints_bwd_kernel, GRID1(B), matr, bmatr, x1, x2, x3, hazards, follow, gamma_b,
                       batch_size, B, NP, sc, dx1, dx2, dx3, dV1);
    NLBAC_CHECK_LAUNCH("nlbac_pvtol_constraints_bwd");
    return 0;
}
