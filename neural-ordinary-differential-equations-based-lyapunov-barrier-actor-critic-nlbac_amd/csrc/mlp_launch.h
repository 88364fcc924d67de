// The launch descriptor shared by the MLP kernel families (mlp_kernels.hip: LDS-tiled; mlp_rr_kernels.hip:
// register-resident panels) and what the first hands to the second.
#pragma once
#include "common.h"
#include "dy_heads.h"

// Rows per chunk of the skinny-gradient partial sums that nlbac_mlp_bwd_data leaves for nlbac_mlp_bwd_weights
// (nlbac_mlp_io::skinny_ws): the finest tile of the data-backward kernels — the 32-row kernels write two chunks per tile.
#define NLBAC_SK_CHUNK 16

struct MlpLaunch {
    nlbac_mlp net[NLBAC_MAX_NETS];
    nlbac_mlp_io io[NLBAC_MAX_NETS];
    int B;
    int ld;          // LDS row stride in floats
    int n_slabs;     // bwd_weights only
    int rows_per_slab;
    long slab_stride;
};

// The register-resident forward (mlp_rr_kernels.hip): 0 = launched, 1 = these nets are not its (the LDS-tiled kernels
// take the launch), < 0 = error.
int nlbac_mlp_rr_fwd_launch(const MlpLaunch& L, int n_nets, const nlbac_gauss_head& G, const char* who, hipStream_t s);
bool nlbac_mlp_rr_eligible(const nlbac_mlp* nets, int n_nets);
bool nlbac_mlp_rr_serves_shape(int n_layers, int in_dim, int hid, int out_dim);
// The register-resident data backward of the same nets (NLBAC_MLP_RR_BWD=0 keeps the LDS-tiled kernel): same return values.
int nlbac_mlp_rr_bwd_launch(const MlpLaunch& L, int n_nets, const nlbac_dy_head& H, const char* who, hipStream_t s);

// The one-launch weight/bias gradients of narrow nets (mlp_dw16_kernels.hip; NLBAC_MLP_DW16=0 keeps the older kernels).
bool nlbac_mlp_dw16_eligible(const nlbac_mlp* nets, int n_nets, int B);
int nlbac_mlp_dw16_launch(const MlpLaunch& L, int n_nets, hipStream_t s);

// The quarter-panel kernels (mlp_rrq_kernels.hip: 16-row workgroups, hid = 128 / 256; NLBAC_MLP_RRQ=0 keeps the half-panel
// ones): same return values as above.
bool nlbac_mlp_rrq_eligible(const nlbac_mlp* nets, int n_nets);
int nlbac_mlp_rrq_fwd_launch(const MlpLaunch& L, int n_nets, const nlbac_gauss_head& G, const char* who, hipStream_t s);
int nlbac_mlp_rrq_bwd_launch(const MlpLaunch& L, int n_nets, const nlbac_dy_head& H, const char* who, hipStream_t s);
