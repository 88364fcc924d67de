// The augmented-Lagrangian step of the CBF / CLF constraints (sac_cbf_clf.py:502-528, 623-638) as a device function, for
// whichever workgroup holds the finished partial sums: the stand-alone nlbac_auglag launch, the last workgroup of a
// *_constraints_fwd launch (agent_kernels.hip), or the elected workgroup of an MLP forward that evaluates the constraint
// terms in its epilogue (nlbac_gauss_head::cf_kind, mlp_rrq_kernels.hip).
#pragma once
#include "common.h"
#include "scalars.h"

struct AuglagArgs {        // nlbac_auglag's scalar arguments, by value (nlbac_auglag_args in the header)
    int n_cbf, n_clf; float batch_size; int do_lambda_update, do_backup_lambda_update, ratio_mode, backup_mode;
    float lam_lo, lam_hi;
};

// Augmented-Lagrangian scalars.  One wave: lanes sum the partial columns, lane 0 does the scalar bookkeeping.
// backup_mode: 0 no backup controller (learned-barrier copies), 1 backup shares rho with the primary
// (Unicycle / SimulatedCars), 2 backup keeps its own rho (Pvtol); lambda clamp [lam_lo, lam_hi].
// required_matrix sums, ratio, lambda / rho updates and loss coefficients (sac_cbf_clf.py:502-528, 623-638) by ONE
// workgroup: the stand-alone nlbac_auglag launch, or the last workgroup of a constraints_fwd launch (COHERENT: the
// partials were published by other workgroups of the same launch).
__device__ __forceinline__ void auglag_finish_at(const AuglagArgs A, float* sc_global, float* sc, bool write_back = true);

// sc: NLBAC_SC_SIZE_ENUM floats of LDS (8-byte aligned), stage: stage_cap floats of LDS (>= one block's columns) — scratch
// of the calling workgroup, free for the duration of the call.  All threads of the workgroup call it.
template <bool COHERENT>
__device__ __forceinline__ void auglag_body_at(const float* partials, int n_blk, const AuglagArgs A, float* sc_global,
                                               float* sc, float* s_stage, int stage_cap) {
    // The scalar bookkeeping below is ~60 dependent reads / writes of the scalars block by ONE thread: on the block in
    // global memory each is a trip to L2 (unicycle_constraints_fwd: 13 us, most of it here); it runs on a copy in LDS,
    // brought in and written back by the whole workgroup.
    for (int t = threadIdx.x; t < NLBAC_SC_SIZE_ENUM; t += blockDim.x) sc[t] = sc_global[t];
    __syncthreads();
    const int n_cbf = A.n_cbf, n_clf = A.n_clf, backup_mode = A.backup_mode;
    const float batch_size = A.batch_size;
    const int nc = n_cbf + n_clf, ncol = nc + (backup_mode ? n_cbf : 0);
    // the other workgroups' partial sums come in with rounds of device-scope loads (every thread a few), staged in LDS
    // stage_cap floats at a time, then column c is summed in block order by thread c — n_blk dependent loads per column
    // took ~0.6 us each (16 blocks: unicycle_constraints_fwd 15 us, two thirds of it here)
    {
        const int per_round = max(1, stage_cap / ncol);          // blocks per round
        float s = 0.f;
        for (int b0 = 0; b0 < n_blk; b0 += per_round) {
            const int nb = min(per_round, n_blk - b0);
            for (int idx = threadIdx.x; idx < nb * ncol; idx += blockDim.x) {
                const float* q = partials + (long)b0 * ncol + idx;
                s_stage[idx] = COHERENT ? coherent_load(q) : *q;
            }
            __syncthreads();
            if ((int)threadIdx.x < ncol)
                for (int b = 0; b < nb; ++b) s += s_stage[b * ncol + threadIdx.x];
            __syncthreads();
        }
        if ((int)threadIdx.x < ncol) {
            const int c = threadIdx.x;
            s = s / batch_size;
            if (c < nc) sc[SC_REQ + c] = s; else sc[SC_BREQ + (c - nc)] = s;
        }
    }
    __syncthreads();
    auglag_finish_at(A, sc_global, sc);
}

// The second half: the scalar bookkeeping on required sums that are already in the LDS copy `sc` of the scalars block
// (sc[SC_REQ + c], sc[SC_BREQ + c], divided by the batch size), and the write-back.  All threads; starts at a point where
// every thread's writes to `sc` are complete (the caller's barrier).
// write_back = false: the step stays in the LDS copy (a workgroup that only needs the step's coefficients, while another
// launch commits it: nlbac_dy_head::cb_defer).
__device__ __forceinline__ void auglag_finish_at(const AuglagArgs A, float* sc_global, float* sc, bool write_back) {
    const int n_cbf = A.n_cbf, n_clf = A.n_clf, ratio_mode = A.ratio_mode, backup_mode = A.backup_mode;
    const int do_lambda_update = A.do_lambda_update, do_backup_lambda_update = A.do_backup_lambda_update;
    const float lam_lo = A.lam_lo, lam_hi = A.lam_hi;
    const int nc = n_cbf + n_clf;
    if (threadIdx.x == 0) {
    double* rho_p = reinterpret_cast<double*>(sc + SC_RHO_F64);
    double* brho_p = backup_mode == 1 ? rho_p : reinterpret_cast<double*>(sc + SC_BRHO_F64);
    // ---- primary (sac_cbf_clf.py:506-528)
    {
        const float* req = sc + SC_REQ;
        float* lam = sc + SC_LAMBDA;
        double ratio = 1.0;
        if (n_clf && ratio_mode) {
            float m = 0.f;
            for (int c = 0; c < n_cbf; ++c) m += req[c];
            m = fabsf(m / (float)n_cbf);
            float r = m / fabsf(req[nc - 1]);
            if (ratio_mode == 2) r = fmaxf(r, 0.002f);
            ratio = (double)r;
        }
        sc[SC_RATIO] = (float)ratio;
        double rho = *rho_p;
        if (do_lambda_update)
            for (int c = 0; c < nc; ++c)
                lam[c] = fminf(fmaxf(lam[c] + (float)rho * req[c], lam_lo), lam_hi);
        rho = fmin(rho * 1.0005, 200.0);
        *rho_p = rho;
        const float ch = (float)(rho / 2.0);
        float loss = 0.f;
        for (int c = 0; c < n_cbf; ++c) {
            const float g = req[c];
            loss += lam[c] * g + ch * g * g;
            sc[SC_COEF + c] = lam[c] + (ch * g + ch * g);
        }
        if (n_clf) {
            const float g = req[nc - 1];
            const float l1 = (float)((double)lam[nc - 1] * ratio);
            const float c2 = (float)(ratio * ratio * rho / 2.0);
            loss += l1 * g + c2 * g * g;
            sc[SC_COEF + nc - 1] = l1 + (c2 * g + c2 * g);
        }
        sc[SC_PL2] = loss;
    }
    // ---- backup (sac_cbf_clf.py:623-638)
    if (backup_mode) {
        const float* req = sc + SC_BREQ;
        float* lam = sc + SC_BLAMBDA;
        double rho = *brho_p;
        if (do_backup_lambda_update)
            for (int c = 0; c < n_cbf; ++c)
                lam[c] = fminf(fmaxf(lam[c] + (float)rho * req[c], lam_lo), lam_hi);
        rho = fmin(rho * 1.0005, 200.0);
        *brho_p = rho;
        const float ch = (float)(rho / 2.0);
        float loss = 0.f;
        for (int c = 0; c < n_cbf; ++c) {
            const float g = req[c];
            loss += lam[c] * g + ch * g * g;
            sc[SC_BCOEF + c] = lam[c] + (ch * g + ch * g);
        }
        sc[SC_BPL2] = loss;
    }
    }
    __syncthreads();
    if (!write_back) return;
    // what the step may have changed: ratio / losses (4..6), multipliers, coefficients, required sums, rho (16..115)
    for (int t = threadIdx.x; t < NLBAC_SC_SIZE_ENUM; t += blockDim.x)
        if ((t >= SC_RATIO && t <= SC_BPL2) || (t >= SC_LAMBDA && t < SC_MEAN_LOGP)) sc_global[t] = sc[t];
}

// The step on the per-TILE column sums a constraint head left (mlp_rrq_kernels.hip; NC columns = n_cbf + n_clf + n_cbf):
// the scalars block into the LDS copy `scl` (NLBAC_SC_SIZE_ENUM floats), every thread sums the tiles t, t + 256, ... of
// all columns, a fixed tree over the workgroup (red: 4 * NC floats) — whichever workgroup runs it, the same sums —, then
// the bookkeeping.  COHERENT: the sums were published by other workgroups of this launch.  All 256 threads.
template <int NC, bool COHERENT>
__device__ __forceinline__ void auglag_from_tiles(const float* partials, unsigned n_tiles, const AuglagArgs A,
                                                  float* sc_global, float* scl, float* red, bool write_back) {
    const int tid = threadIdx.x;
    for (int t = tid; t < NLBAC_SC_SIZE_ENUM; t += 256) scl[t] = sc_global[t];
    float v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = 0.f;
    for (unsigned b = tid; b < n_tiles; b += 256)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* q = partials + (long)b * NC + c;
            v[c] += COHERENT ? coherent_load(q) : *q;
        }
    block_sum_256<NC>(v, red);
    const int nc = A.n_cbf + A.n_clf;
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float sv = v[c] / A.batch_size;
            if (c < nc) scl[SC_REQ + c] = sv; else scl[SC_BREQ + (c - nc)] = sv;
        }
    }
    __syncthreads();
    auglag_finish_at(A, sc_global, scl, write_back);
}

