// What the two kernels of the continuous-adjoint step share (node_adjoint_kernels.hip: LDS-tiled, any width up to 256,
// with or without kept activations; node_adj_rr_kernels.hip: register-resident chains, the reference's NODE shapes, mask
// mode): the launch descriptor of nlbac_node_adj_step.
#pragma once
#include "mlp_device.h"
#include "ode_control.h"

#define ADJ_LDS_MAX (160 * 1024 - 64)
#define ADJ_MAX_STAGES 8
#define ADJ_MAX_NS 8
#define ADJ_MAX_NU 4
#define ADJ_WP 24                /* padded width of a row of z = [y(ns) | a_x(ns) | a_u(nu)] in LDS */
#define ADJ_MAX_GOUT 32

struct NodeAdjLaunch {
    nlbac_mlp net[2];                 // f, g
    const float* u;                   // [n][nu]
    const float* Z0;                  // [n][W]   state at the step start
    float* KZ;                        // [S][n][W] stage derivatives (s-time); stages < st_lo are read, the others written
    float* Z1; float* ERR;            // [n][W] step result / error estimate, or null
    float* ZS;                        // [S][n][W] stage inputs, kept for the weight gradients (or null)
    float* dG;                        // [S][n][ns*nu] output-layer gradient of g_net (with ZS)
    float* acts[2]; long acts_ls[2];  // [layer][S*n][hid] activations of the stage (with ZS), else null: masks in LDS
    float* dz[2];
    int n, rpp, n_s, n_u, W;
    int st_lo, st_hi, S_total;
    float beta[ADJ_MAX_STAGES][ADJ_MAX_STAGES];
    float c_out[ADJ_MAX_STAGES]; int n_out;
    float c_err[ADJ_MAX_STAGES]; int n_err;
    const double* h_dev; int h_stride; float h_val[8];
    const double* ctl;                // rows of problems whose C_DONE is set are left alone (device-driven step chain)
    int ld, sw_off1, mask_words;      // mask_words: uint32 words of one group's LDS mask store
    float* ip_out; double t_end;      // (register-resident kernel) the interpolant of z at t_end, should this attempt finish the solve
};


// The register-resident kernel: 0 = launched, 1 = not its launch (the LDS-tiled kernel takes it), < 0 = error.
int nlbac_node_adj_rr_launch(NodeAdjLaunch& L, hipStream_t s);
