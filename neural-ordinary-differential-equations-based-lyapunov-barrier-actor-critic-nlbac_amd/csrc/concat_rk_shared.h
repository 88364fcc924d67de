// Launch descriptors of the fused RK kernels of the single-net NODE  dx/dt = net([x | c])  (SimulatedCars,
// C/sac_cbf_clf/model.py:179-205), shared by the LDS-tiled kernels (concat_node_kernels.hip) and the register-resident
// ones (concat_rr_kernels.hip).
#pragma once
#include "mlp_device.h"
#include "ode_control.h"

#define CK_MAX_STAGES 8
#define CK_NS 16          // widest state (n_s <= 16); LDS row stride of state-sized rows in the LDS-tiled kernels
#define CK_LD 17          // the same rows in the register-resident kernels: lane (q, row) reads component 4 k + q of ITS row, so
                          // sixteen lanes walk the rows of one column — a stride of 16 floats put them on two banks (8-way conflict)
#define CK_NC 4           // carried inputs per row (n_c <= 4)

struct ConcatRkLaunch {
    nlbac_mlp net;
    const float* y0; const float* c;
    int n, rpp, n_s, n_c;
    int stage_begin, stage_end;
    float beta[CK_MAX_STAGES][CK_MAX_STAGES];
    float c_out[CK_MAX_STAGES]; int n_out;
    float c_err[CK_MAX_STAGES]; int n_err;
    const double* h_dev; int h_stride; float h_val[8];
    float* K; float* Y;
    float* acts; long acts_ls;
    int acts_bits;                    // register-resident kernels only: acts hold ReLU mask words [layer][stage*n + row][4]
    float* out; float* err;
    // input normalisation / output de-normalisation of the field (the Quadrotor NODE, /root/reference/README.md:192):
    // dx/dt = out_mu + out_sig * net(([x | c] - in_mu) * in_isig); norm = [in_mu | in_isig] (in_dim each) then
    // [out_mu | out_sig] (n_s each), or null.  Xn: [stage][n][in_dim] normalised net inputs kept for the first layer's
    // weight gradient (or null).
    const float* norm; float* Xn;
    int ld;
    // device-driven dopri5 chain, as NodeRkLaunch (node_kernels.hip): step slots `slot_floats` apart, done problems
    // skipped, FSAL from the previous slot, optional fused norm + controller epilogue
    int S_total;
    const double* ctl; long slot_floats;
    int norm_mode, n_slots; float rtol, atol; double t_end;
    float* partials; unsigned* tickets; double* ctl_w; double* hslots; double* alog; int alog_cap;
    float* ip_out;                    // as NodeRkLaunch::ip_out (nlbac_rk_chain::interp_out; no out-map for this field)
};

struct ConcatRkBwdLaunch {
    nlbac_mlp net;
    const float* acts; long acts_ls;
    int acts_bits;                    // as ConcatRkLaunch::acts_bits (excludes dz)
    float* dz;
    float* dK; const float* dYup;
    float* dy0; int dy0_in;
    float* dc; int dc_acc;
    int n, rpp, n_s, n_c, S_total, st_lo, st_hi, dx_stage0;
    float beta[CK_MAX_STAGES][CK_MAX_STAGES];
    const double* h_dev; int h_stride; float h_val[8];
    const float* norm;                // as ConcatRkLaunch::norm
    float* dyn;                       // [stage][n][n_s] gradient w.r.t. the net's own output (dK * out_sig), kept with dz
    int ld;
    // device-driven chain, as NodeRkBwdLaunch: launch back_idx differentiates slot C_NACC - back_idx of each problem
    const double* ctl; long slot_floats; int back_idx, n_slots; const double* hslots;
    int ip_on; const float* ip_dout;  // as NodeRkBwdLaunch::ip_* (nlbac_rk_chain::interp_dout)
};


// The register-resident kernels (concat_rr_kernels.hip): 0 = launched, 1 = not theirs (the LDS-tiled kernels take the
// launch), < 0 = error.
int nlbac_concat_rr_fwd_launch(ConcatRkLaunch& L, hipStream_t s);
int nlbac_concat_rr_bwd_launch(ConcatRkBwdLaunch& L, hipStream_t s);
bool nlbac_concat_rr_eligible(const nlbac_mlp* net);
