// dopri5 step-size controller shared by the forward solve (ode_kernels.hip) and the adjoint solve
// (node_adjoint_kernels.hip).  ctl: per problem NLBAC_DOPRI_CTL doubles.
#pragma once
#include "common.h"

enum { C_H = 0, C_T = 1, C_RATIO = 2, C_ACCEPT = 3, C_DONE = 4, C_X = 5, C_H0 = 6, C_D0 = 7, C_D1 = 8, C_D2 = 9,
       C_NSTEPS = 10, C_HUSED = 11,
       C_NACC = 12,    // accepted steps so far = index of the step slot the current attempt works in (device-driven chain)
       C_OVF = 13 };   // the solve ran out of step slots (it is stopped: C_DONE is set with it; the host restarts it)

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
