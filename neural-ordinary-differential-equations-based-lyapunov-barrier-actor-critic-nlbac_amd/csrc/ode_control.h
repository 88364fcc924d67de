// dopri5 step-size controller shared by the forward solve (ode_kernels.hip) and the adjoint solve
// (node_adjoint_kernels.hip).  ctl: per problem NLBAC_DOPRI_CTL doubles.
#pragma once
#include "common.h"

enum { C_H = 0, C_T = 1, C_RATIO = 2, C_ACCEPT = 3, C_DONE = 4, C_X = 5, C_H0 = 6, C_D0 = 7, C_D1 = 8, C_D2 = 9,
       C_NSTEPS = 10, C_HUSED = 11,
       C_NACC = 12,    // accepted steps so far = index of the step slot the current attempt works in (device-driven chain)
       C_OVF = 13 };   // the solve ran out of step slots (it is stopped: C_DONE is set with it; the host restarts it)

enum { C_SEQ = 15 };   // HOST copy only: the stamp of the launch that wrote the block (ctl_host_post); the device block's slot is unused

// One problem's control block into the host's copy (pinned, fine-grained memory the device writes directly), as a
// sequence lock the host can poll without an event: stamp <- -seq, fields, stamp <- +seq.  Every store is a
// system-scope (write-through) store and each group waits for its acknowledgements before the next is issued — the
// order the host sees; no release fence (a system-scope fence writes back the XCD's whole L2: the RK launch's rows are
// dirty in it, and none of that is the host's business).  seq = 0: plain copy (the host reads behind an event).
__device__ __forceinline__ void ctl_host_post(double* dst, const double* src, double seq) {
    if (seq <= 0.0) {
        for (int k = 0; k < NLBAC_DOPRI_CTL; ++k) dst[k] = src[k];
        return;
    }
    double v[NLBAC_DOPRI_CTL];
    for (int k = 0; k < NLBAC_DOPRI_CTL; ++k) v[k] = src[k];
    __hip_atomic_store(dst + C_SEQ, -seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = 0; k < NLBAC_DOPRI_CTL; ++k)
        if (k != C_SEQ) __hip_atomic_store(dst + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(dst + C_SEQ, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The controller of problem p on finished norms (torchdiffeq _select_initial_step / _compute_error_ratio /
// _optimal_step_size with safety 0.9, ifactor 10, dfactor 0.2; steps are not clipped to t_end).
//  mode 0: n0 = ||y0/scale||, n1 = ||f0/scale||          -> C_H0 (first guess), resets C_T / C_NSTEPS / C_DONE
//  mode 1: n0 = ||(f1 - f0)/scale||                        -> C_H  (initial step)
//  mode 2: n0 = ||err/tol||                                -> accept / done / next C_H
__device__ __forceinline__ void dopri_control_vals(double n0, double n1, int p, int mode, double t_end, double* ctl,
                                                   int n_slots = 1 << 30) {
    double* c = ctl + (long)p * NLBAC_DOPRI_CTL;
    if (mode == 0) {
        const double d0 = n0, d1 = n1;
        c[C_D0] = d0; c[C_D1] = d1;
        c[C_H0] = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        c[C_T] = 0.0; c[C_NSTEPS] = 0.0; c[C_DONE] = 0.0; c[C_NACC] = 0.0; c[C_OVF] = 0.0;
    } else if (mode == 1) {
        const double h0 = c[C_H0], d1 = c[C_D1];
        const double d2 = n0 / h0;
        c[C_D2] = d2;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
        else h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
        c[C_H] = fmin(100.0 * h0, h1);
    } else {
        const double ratio = n0;
        const double h = c[C_H], t = c[C_T];
        const bool accept = ratio <= 1.0;
        double fac;
        if (ratio == 0.0) fac = 10.0;
        else {
            const double dfac = (ratio < 1.0) ? 1.0 : 0.2;
            fac = fmin(10.0, fmax(0.9 / pow(ratio, 0.2), dfac));
        }
        c[C_RATIO] = ratio; c[C_ACCEPT] = accept ? 1.0 : 0.0; c[C_HUSED] = h;
        c[C_NSTEPS] += 1.0;
        if (accept && t + h >= t_end) {
            c[C_DONE] = 1.0;
            c[C_X] = (t_end - t) / h;
        } else {
            if (accept) {
                c[C_T] = t + h;
                if ((int)c[C_NACC] + 1 >= n_slots) { c[C_OVF] = 1.0; c[C_DONE] = 1.0; }   // no slot left for the next step
                else c[C_NACC] += 1.0;
            }
            c[C_H] = h * fac;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 4th-order interpolant of an accepted dopri5 step (torchdiffeq _interp_fit / _interp_evaluate with the mid-point
// coefficients DPS_C_MID), one state component:  y(t0 + x h) = a0 + x (d + x (c + x (b + x a))).  Shared by the
// interpolation launches (ode_kernels.hip) and the RK kernels that evaluate it themselves (nlbac_rk_chain::interp_*):
// the same expressions, so the same bits.
// ---------------------------------------------------------------------------------------------------------------------
#define DPM0 (6025192743.0 / 30085553152.0 / 2.0)
#define DPM2 (51252292925.0 / 65400821598.0 / 2.0)
#define DPM3 (-2691868925.0 / 45128329728.0 / 2.0)
#define DPM4 (187940372067.0 / 1594534317056.0 / 2.0)
#define DPM5 (-1776094331.0 / 19743644256.0 / 2.0)
#define DPM6 (11237099.0 / 235043384.0 / 2.0)

// k[j]: the step's stage derivatives K_0..K_6 of this component; a0 = y0, a1 = y1 (the step's result)
// (no multiply-add contraction inside: a, b, c cancel terms of size 32 |y| down to O(h^2), and whether the compiler
// fuses a given multiply into the neighbouring add depends on the code around the inlined body — 6e-6 between two call sites)
__device__ __forceinline__ float dopri_interp_value(float a0, float a1, const float (&k)[7], float h, float x) {
#pragma clang fp contract(off)
    const float cm[7] = {(float)DPM0, 0.f, (float)DPM2, (float)DPM3, (float)DPM4, (float)DPM5, (float)DPM6};
    float ym = a0;
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (cm[j] != 0.f) ym = ym + k[j] * (cm[j] * h);
    const float f0 = k[0], f1 = k[6];
    const float a = 2.f * h * (f1 - f0) - 8.f * (a1 + a0) + 16.f * ym;
    const float b = h * (5.f * f0 - 3.f * f1) + 18.f * a0 + 14.f * a1 - 32.f * ym;
    const float c = h * (f1 - 4.f * f0) - 11.f * a0 - 5.f * a1 + 16.f * ym;
    const float d = h * f0;
    return a0 + x * (d + x * (c + x * (b + x * a)));
}

// its backward for one component: g = d loss / d y(t0 + x h)  ->  d loss / d y0, d y1, d K_0..K_6
__device__ __forceinline__ void dopri_interp_grad(float g, float h, float x, float& dy0, float& dy1, float (&dk)[7]) {
#pragma clang fp contract(off)
    const float cm[7] = {(float)DPM0, 0.f, (float)DPM2, (float)DPM3, (float)DPM4, (float)DPM5, (float)DPM6};
    const float x2 = x * x, x3 = x2 * x, x4 = x2 * x2;
    const float A = x4 * g, Bc = x3 * g, C = x2 * g, D = x * g;
    const float ym = 16.f * A - 32.f * Bc + 16.f * C;
    dy0 = g - 8.f * A + 18.f * Bc - 11.f * C + ym;
    dy1 = -8.f * A + 14.f * Bc - 5.f * C;
    const float f0b = h * (-2.f * A + 5.f * Bc - 4.f * C + D);
    const float f1b = h * (2.f * A - 3.f * Bc + C);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        float v = (cm[j] * h) * ym;
        if (j == 0) v += f0b;
        if (j == 6) v += f1b;
        dk[j] = v;
    }
}
