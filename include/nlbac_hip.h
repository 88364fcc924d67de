/*
 * nlbac_hip.h — C ABI of the MI355X (gfx950) NLBAC hot path.
 *
 * The reference (pure Python) has no native interface; what a drop-in has to
 * replace are the PyTorch/ATen op sequences behind these reference call sites
 * (U = NLBAC_Unicycle_RL_training/Unicycle_RL_training):
 *
 *   nlbac_mlp_*            QNetwork / LyaNetwork / GaussianPolicy.forward and the
 *                          autograd backward through them   U/sac_cbf_clf/model.py:37-114
 *                          f_net / g_net of NeuralODEModel   U/sac_cbf_clf/model.py:186-206
 *   nlbac_gauss_*          GaussianPolicy.sample             U/sac_cbf_clf/model.py:116-128
 *   nlbac_td_targets       target / MSE block                U/sac_cbf_clf/sac_cbf_clf.py:231-246
 *   nlbac_unicycle_*       get_state + get_policy_loss_2 / backup_get_policy_loss_2
 *                                                            U/sac_cbf_clf/dynamics.py:53-58
 *                                                            U/sac_cbf_clf/sac_cbf_clf.py:408-640
 *   nlbac_ode_*            torchdiffeq.odeint call sites     U/sac_cbf_clf/sac_cbf_clf.py:453,577
 *                                                            U/sac_cbf_clf/model.py:252
 *   nlbac_adam_* / soft    torch.optim.Adam.step, soft_update
 *                                                            U/sac_cbf_clf/sac_cbf_clf.py:249-255,284-308
 *                                                            U/sac_cbf_clf/utils.py:75-79
 *
 * Conventions: every pointer is a DEVICE pointer to fp32 unless stated; no
 * entry point allocates, frees or synchronises; all work is enqueued on the
 * caller's stream (a hipStream_t passed as void*).  Return 0 on success,
 * <0 on error (text via nlbac_last_error()).  Matrices are row-major with an
 * explicit leading dimension (`*_ld`, in floats).
 */
#ifndef NLBAC_HIP_H
#define NLBAC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define NLBAC_ABI_VERSION 16 /* bumped whenever an exported signature or struct changes; nlbac_abi_version() returns it */
#define NLBAC_MAX_LAYERS 6
#define NLBAC_MAX_NETS 8
#define NLBAC_MLP_TILE 32 /* samples per workgroup in the MLP kernels */

typedef void *nlbac_stream_t; /* hipStream_t */

int nlbac_abi_version(void);
const char *nlbac_last_error(void);
/* Test hook (no reference counterpart): one launch of the last-workgroup election the fused loss / controller kernels
 * rely on (csrc/common.h publish_and_elect: a gfx950-only ordering contract).  Workgroup b publishes n_vals values
 * ((b * 31 + k * 7 + salt * 13) % 251); the elected workgroup leaves their sums over all workgroups in out[0..n_vals)
 * and adds 1 to out[n_vals].  partials: n_blocks * n_vals floats; ticket: one zeroed unsigned (left zero again). */
int nlbac_elect_selftest(float *partials, unsigned *ticket, float *out, int n_blocks, int n_vals, unsigned salt,
                         nlbac_stream_t s);

/* ------------------------------------------------------------------------
 * ReLU MLP:  in_dim -> hid -> ... -> hid -> out_dim   (n_layers Linear layers;
 * layers 0..n_layers-2 are "wide" (hid outputs, ReLU, run on MFMA), the last
 * layer is "skinny" (out_dim <= 16, no activation, run on VALU).
 *
 * Parameters live in ONE flat fp32 buffer in the layout of
 * torch.nn.Linear: W_l [N_l][K_l] row-major at params+w_off[l], bias at
 * params+b_off[l].  Gradients use the same offsets inside a "grad slab".
 * `packed` holds MFMA-fragment-ordered copies of the wide layers, written by
 * nlbac_mlp_pack() after every optimiser step:
 *   pf_off[l]  forward pack of layer l      (B operand of  Y = X W^T)
 *   pb_off[l]  backward pack of layer l>=1  (B operand of dX = dY W ), -1 if none
 * Nets of width 20..128 (hid % 4 == 0) with at least one hid x hid layer also get "RR packs" of those layers
 * (layers 1 .. n_layers-2, contiguous, rr_layer_floats(hid) floats each): the A fragments of
 * v_mfma_f32_16x16x4_f32 in the issue order of the register-resident layer chains (csrc/rr_device.h), which the
 * fused RK kernels of the control-affine NODE stream:
 *   rr_fwd_off  of W_l   (D[unit][row] = W_l X^T),        -1 if none
 *   rr_bwd_off  of W_l^T (D[k][row]    = W_l^T dZ^T),     -1 if none
 * packed_floats = size of `packed` as nlbac_mlp_pack_layout returned it.
 * ---------------------------------------------------------------------- */
typedef struct nlbac_mlp {
    int n_layers, in_dim, hid, out_dim;
    const float *params;
    int w_off[NLBAC_MAX_LAYERS];
    int b_off[NLBAC_MAX_LAYERS];
    float *packed;
    int pf_off[NLBAC_MAX_LAYERS];
    int pb_off[NLBAC_MAX_LAYERS];
    int rr_fwd_off, rr_bwd_off, packed_floats;
    int rr_kind;   /* 0 none; 1 "chain" packs (above); 2 "panel" packs: nets with ONE hid x hid layer (n_layers == 3) and
                      hid % 32 == 0 get that layer as two panels of hid / 32 output blocks each (hid * hid floats per
                      direction), streamed by the register-resident actor / critic kernels (csrc/mlp_rr_kernels.hip) */
} nlbac_mlp;

/* Per-launch tensors of one net.  Unused pointers are NULL. */
typedef struct nlbac_mlp_io {
    const float *x0; int x0_dim, x0_ld; /* input columns [0, x0_dim)                */
    const float *x1; int x1_dim, x1_ld; /* input columns [x0_dim, in_dim) or NULL   */
    float *y; int y_ld;                 /* fwd out: (B, out_dim)                    */
    float *acts;                        /* [n_layers-1][B][hid] post-ReLU, saved by fwd */
    long acts_ls;                       /* layer stride of acts/dz in floats; 0 = B*hid.  Lets several
                                           launches (RK stages) fill row blocks of one [layer][rows][hid] buffer */
    const float *dy; int dy_ld;         /* bwd in:  (B, out_dim)                    */
    int dz_first;                       /* ABI 13 (in what was padding).  dz rows of layers [0, dz_first) are not wanted:
                                           with skinny_ws the data backward leaves layer 0's weight / bias gradients
                                           as partial sums, and nlbac_mlp_bwd_weights reads dz from layer 1 on — 1 saves
                                           the (B, hid) store of layer 0 (a third of the launch's bytes at hid 256).
                                           Honoured by the kernels that can (others store every layer); 0: all */
    float *dz;                          /* [n_layers-1][B][hid] pre-activation grads (bwd_data out, bwd_weights in) */
    float *dx; int dx_ld;               /* bwd_data out: (B, in_dim) or NULL        */
    int dx_first;                       /* first input column dx is wanted for: columns [0, dx_first) of dx are NOT
                                           written (the Q(s, pi) nets' gradient is consumed for the action columns only).
                                           Sits in what was padding: sizeof and the other offsets are unchanged */
    float *grad;                        /* bwd_weights out: slab 0 of the flat grad (same offsets as params) */
    float *skinny_ws;                   /* or NULL.  This net's block of the nlbac_mlp_bwd_weights workspace (ws +
                                           i * ws_floats / n_nets): nlbac_mlp_bwd_data then leaves the per-16-row partial
                                           sums of the bias / first- / last-layer gradients there (it has every dz tile
                                           in LDS anyway) and nlbac_mlp_bwd_weights only reduces them.  B <= 32768. */
    unsigned *masks;                    /* or NULL (ABI 4).  [n_layers-1 = 2][B][8] ReLU mask words of the register-resident
                                           kernels (3-layer nets of width 64 / 128 / 256: nlbac_mlp_masks_ok): the forward
                                           writes them, the data backward gates with them instead of reading the saved
                                           activation rows.  A net that is only differentiated w.r.t. its inputs (dx
                                           wanted, no dz / weight gradients) may then pass acts == NULL: 64 B per row
                                           instead of 8 * hid.  All nets of a data-backward launch have them, or none. */
} nlbac_mlp_io;

/* 1 when launches of these nets are served by the kernels that write / read nlbac_mlp_io::masks, else 0 */
int nlbac_mlp_masks_ok(const nlbac_mlp *nets, int n_nets);

/* number of floats nlbac_mlp_pack needs in `packed`, and the offsets it will use */
int nlbac_mlp_pack_layout(nlbac_mlp *net /* in: n_layers,in_dim,hid ; out: pf_off,pb_off */);
int nlbac_mlp_pack(const nlbac_mlp *nets, int n_nets, nlbac_stream_t s);

/* y = MLP(x) for n_nets independent nets over the same B rows (grid.y = net). */
int nlbac_mlp_fwd(const nlbac_mlp *nets, const nlbac_mlp_io *io, int n_nets, int B, nlbac_stream_t s);
/* The same for policy nets (out_dim = 2 n_u: mean | log_std) with GaussianPolicy.sample (model.py:116-128) applied to
 * every output row by the launch itself — nlbac_gauss_sample_fwd's arithmetic and outputs, no launch of its own.  Net i's
 * rows are rows i*B.. of the stacked eps (n_nets*B, n_u), action and logp arrays. */
typedef struct nlbac_gauss_head {
    const float *eps, *scale, *bias; int n_u;
    float *action; int action_ld; float *logp;
    /* ABI 7 — optional "constraint head" (cf_kind != 0; nlbac_mlp_fwd_head): net cf_net of the launch is the Lyapunov
     * critic evaluated on the predicted look-ahead points, y = V(p(x')), and each of its workgroups goes on, for its
     * rows, with the task's constraint terms — what *_constraints_fwd does as a launch of its own: the terms (kept for
     * the backward), the relu-filtered column sums per tile, and, in the workgroup elected last, the augmented-
     * Lagrangian step on the finished sums (nlbac_auglag).  cf_kind 1 = nlbac_unicycle_constraints_fwd (cf_nh == 7
     * hazards; same arguments: ps (B,2), ps_next (2B,2), V (B), hazards, r2 = r_coll^2, dt, gamma_b, gamma_l, matr
     * (B,8), bmatr (B,7)); cf_partials [n_tiles][15] with n_tiles = ceil(B / 16), cf_tickets 1 + ceil(n_tiles / 16)
     * zeroed uint32 (left zeroed), cf_auglag_* = the nlbac_auglag_args (backup_mode != 0: 15 columns), cf_sc the
     * scalars block.  Row arithmetic as the launch it replaces; the column sums are taken per 16-row tile (to rounding). */
    int cf_kind, cf_net, cf_nh;
    const float *cf_ps, *cf_ps_next, *cf_V, *cf_hazards;
    float cf_r2, cf_dt, cf_gamma_b, cf_gamma_l;
    float *cf_matr, *cf_bmatr, *cf_partials; unsigned *cf_tickets;
    int cf_n_cbf, cf_n_clf; float cf_batch_size; int cf_do_lambda_update, cf_do_backup_lambda_update, cf_ratio_mode, cf_backup_mode;
    float cf_lam_lo, cf_lam_hi;
    float *cf_sc;
    /* ABI 16 — cf_defer != 0: no election and no augmented-Lagrangian step in this launch: the tiles' column sums go to
     * cf_partials as plain stores, the launch's tile count to cf_tiles[0]; the consumers sum them themselves
     * (nlbac_dy_head::cb_defer; nlbac_head_sums kind 4 commits the step).  cf_tickets is not used. */
    int cf_defer; unsigned *cf_tiles;
} nlbac_gauss_head;
int nlbac_mlp_fwd_gauss(const nlbac_mlp *nets, const nlbac_mlp_io *io, int n_nets, int B,
                        const nlbac_gauss_head *head, nlbac_stream_t s);
/* nlbac_mlp_fwd with a constraint head (head->eps == NULL: no Gaussian sample).  nlbac_mlp_fwd_head_ok(nets, n_nets) != 0:
 * the launch is served by the kernels that evaluate it (the quarter-panel ones: 3-layer nets of width 128 / 256). */
int nlbac_mlp_fwd_head_ok(const nlbac_mlp *nets, int n_nets);
int nlbac_mlp_fwd_head(const nlbac_mlp *nets, const nlbac_mlp_io *io, int n_nets, int B,
                       const nlbac_gauss_head *head, nlbac_stream_t s);
/* dz (all wide layers) and optionally dx from dy and the saved activations. */
int nlbac_mlp_bwd_data(const nlbac_mlp *nets, const nlbac_mlp_io *io, int n_nets, int B, nlbac_stream_t s);
struct nlbac_dy_head;
/* The same with dL/dy PRODUCED in the launch (io[i].dy is not read): see nlbac_dy_head below, after the per-row
 * entry points it replaces. */
int nlbac_mlp_bwd_data_head(const nlbac_mlp *nets, const nlbac_mlp_io *io, int n_nets, int B,
                            const struct nlbac_dy_head *head, nlbac_stream_t s);
/* Weight/bias gradients from x, dy, acts, dz, as n_slabs partial gradients the caller sums in order
 * (nlbac_adam_fused / nlbac_adam_step / nlbac_reduce_slabs), deterministic.
 * Every call defines every entry of its nets' gradients in ALL n_slabs slabs — the slab sum is the gradient whatever a
 * previous call (another path, batch size or workspace on the same arena) left there.  Two kernels' layouts:
 *   - nets wider than 112, and any net whose launch of nlbac_mlp_bwd_data left the skinny-gradient partials
 *     (nlbac_mlp_io::skinny_ws == this net's block of `ws`): the hidden->hidden matrices are reduced per row range into
 *     the slabs (slab s = rows [s*rows_per_slab, ...)); the skinny first/last layers and all biases are reduced over all
 *     rows into slab 0 through `ws` (>= nlbac_mlp_bwd_weights_ws_floats() floats, always required) and their entries
 *     in slabs 1 .. n_slabs-1 are written as zeros;
 *   - nets of hid <= 112 (the NODEs) whose partials are NOT in `ws` (NLBAC_MLP_DW16=0 switches this path off): every
 *     gradient — biases and skinny layers too — is a partial in every slab (slab s = the 4-row k-steps 4s+w,
 *     4s+w+4 n_slabs, ... of wave w: mlp_dw16_kernels.hip); `ws` is not read; rows * hid must stay below 2^29 - 2^16. */
long nlbac_mlp_bwd_weights_ws_floats(const nlbac_mlp *nets, int n_nets, int B);
int nlbac_mlp_bwd_weights(const nlbac_mlp *nets, const nlbac_mlp_io *io, int n_nets, int B,
                          int n_slabs, long slab_stride, float *ws, long ws_floats, nlbac_stream_t s);

/* ------------------------------------------------------------------------
 * Optimiser (torch.optim.Adam defaults: betas .9/.999, eps 1e-8, no decay).
 * `state` = {int step; float step_size; float bc2_sqrt; uint32 ticket (zero)} on device.
 * ---------------------------------------------------------------------- */
int nlbac_adam_prepare(void *state, double lr, nlbac_stream_t s); /* ++step, bias corrections */
/* p,m,v: n floats; grad: n_slabs slabs (summed in slab order); if target!=NULL:
 * target = (1-tau) target + tau p_new  (soft_update fused; tau<0 disables). */
int nlbac_adam_step(float *p, float *m, float *v, const float *grad, int n_slabs, long slab_stride,
                    long n, const void *state, float *target, float tau, nlbac_stream_t s);
/* nlbac_adam_prepare + nlbac_adam_step + nlbac_mlp_pack of the stepped nets in one launch.  scatter /
 * scatter_target (or NULL): scatter_slots * n uint64 device addresses, [scatter_slots * i + k] = the k-th
 * MFMA-fragment slot of parameter i inside the nets' `packed` buffers (0 = none; forward / backward pack: 2 slots,
 * with RR packs: 4), for the trained and the target copy. */
int nlbac_adam_fused(float *p, float *m, float *v, const float *grad, int n_slabs, long slab_stride, long n,
                     void *state, double lr, float *target, float tau, const void *scatter,
                     const void *scatter_target, int scatter_slots /* 2 or 4 */,
                     int n_alpha /* 0..2 temperatures refreshed by the step itself: alpha_dst[k][0] = exp(p[alpha_off[k]])
                                    after the step (sac_cbf_clf.py:297, 308) */,
                     const long *alpha_off, float *const *alpha_dst,
                     const float *mirror_src, float *mirror_dst /* or NULL: pinned HOST memory; the step's last workgroup
                        writes mirror_src[0..n_mirror) there (the scalars block incl. the refreshed temperatures), so the
                        update's last launch hands its results to the host itself (sac_cbf_clf.py:312-319 `.item()`s) */,
                     int n_mirror, nlbac_stream_t s);
int nlbac_reduce_slabs(float *out, const float *grad, int n_slabs, long slab_stride, long n, nlbac_stream_t s);
int nlbac_soft_update(float *target, const float *src, long n, float tau, nlbac_stream_t s);

/* ------------------------------------------------------------------------
 * Squashed-Gaussian policy head (model.py:116-128).  heads = (n, 2*n_u):
 * [mean | log_std]; eps ~ N(0,1) supplied by the caller (n, n_u).
 * ---------------------------------------------------------------------- */
int nlbac_gauss_sample_fwd(const float *heads, int heads_ld, const float *eps, const float *scale,
                           const float *bias, int n_u, int n, float *action, int action_ld,
                           float *logp, nlbac_stream_t s);
/* d heads from d action = da0+da1+da2 (each (n,n_u) with its own ld, may be NULL)
 * and a uniform d logp = alpha[row / rows_per_problem] * dlogp_mul. */
int nlbac_gauss_sample_bwd(const float *heads, int heads_ld, const float *eps, const float *scale,
                           int n_u, int n, int rows_per_problem, const float *da0, int da0_ld,
                           const float *da1, int da1_ld, const float *da2, int da2_ld,
                           const float *alpha, float dlogp_mul, float *dheads, int dheads_ld,
                           nlbac_stream_t s);

/* ------------------------------------------------------------------------
 * Scalars block `sc` (device, NLBAC_SC_SIZE floats): alpha, lambdas, rho,
 * loss coefficients and per-update loss outputs.  Layout: csrc/scalars.h
 * (mirrored in nlbac_amd/sac_cbf_clf/_layout.py).
 * ---------------------------------------------------------------------- */
#define NLBAC_SC_SIZE 128

/* TD / Lyapunov targets, MSE partial sums and dL/dq (sac_cbf_clf.py:231-246).
 * All q-like arguments are (B) vectors; reward/constraint/mask are read with stride rcm_ld (columns of
 * the minibatch rows); alpha = sc+SC_ALPHA.  Means are over B_norm rows
 * (= B on one GPU, the global batch under data parallelism; the same holds for every *_norm below).
 * partials: [ceil(B/256)][3] squared-error sums (qf1, qf2, lf).
 * ticket (or NULL): a zeroed uint32 (left zeroed) — the launch then finishes the job itself: its last workgroup sums
 * the partials in block order (exactly nlbac_sum_partials' arithmetic) and writes out[0..2] = sum * mul.  The same
 * "(ticket, mul, out)" tail exists on nlbac_td_value; single GPU only (under data parallelism the sums are
 * all-reduced first). */
int nlbac_td_targets(const float *q1t, const float *q2t, const float *lt, const float *nlogp,
                     const float *reward, const float *constraint, const float *mask, int rcm_ld,
                     const float *q1, const float *q2, const float *lf, const float *alpha,
                     float gamma, int B, int B_norm, float *dq1, float *dq2, float *dlf, float *next_q,
                     float *next_l, float *partials, unsigned *ticket, float mul, float *out, nlbac_stream_t s);

/* policy_loss_1 pieces for P controllers (rows p*B..): d min(Q1,Q2)/dq * (-1/B) and
 * partials [P][ceil(B/256)][2] = sums of (alpha_p*logp - minq, logp)
 * (sac_cbf_clf.py:258-273). */
typedef struct nlbac_actor_scalar_args { /* nlbac_actor_scalars' arguments, per problem (0 primary, 1 backup) */
    float target_entropy;
    const float *log_alpha[2];
    float *g_log_alpha[2];
    float *sc;
} nlbac_actor_scalar_args;
/* fused (or NULL) + ticket (a zeroed uint32, left zeroed): the launch's last workgroup also does nlbac_actor_scalars'
 * job for all P problems (same sums in the same order) — single GPU only. */
int nlbac_actor_q_terms(const float *q1, const float *q2, const float *logp, const float *alpha,
                        int B, int B_norm, int P, float *dq1, float *dq2, float *partials,
                        const nlbac_actor_scalar_args *fused, unsigned *ticket, nlbac_stream_t s);
/* nlbac_mlp_bwd_data_head: the per-row step between a forward and the data backward that consumes its result runs in
 * that backward's prologue (each workgroup for its 32 rows), its batch sums are finished by the launch's last
 * designated workgroup; single GPU (the sums are not all-reduced).  Same row arithmetic, and the same outputs left in
 * memory, as the entry point it replaces:
 *   kind 1  nlbac_gauss_sample_bwd  net i = controller i (rows i*B.. of the stacked arrays); out_dim = 2 n_u
 *   kind 2  nlbac_td_targets        nets 0, 1, 2 = Q1, Q2, Lyapunov critic [, 3 = BarrierNet: nlbac_td_value];
 *                                   out[0..2] = the three losses * mul
 *   kind 3  nlbac_actor_q_terms     net i = (controller i / 2, Q1 / Q2 = i % 2) for i < 2 n_prob, nets behind them take
 *                                   io[i].dy as in nlbac_mlp_bwd_data (an independent backward sharing the launch);
 *                                   + nlbac_actor_scalars via `actor`
 * partials: n_nets * n_tiles (kind 2) / 2 * n_prob * n_tiles (kind 3) floats, n_tiles = ceil(B / 16) (the finest tile
 * of the kernels that serve the launch); ticket: 1 + ceil(n / 16) zeroed uint32 for the n = n_nets * n_tiles (kind 2) /
 * n_prob * n_tiles (kind 3) workgroups that take part in the two-level election, left zeroed. */
typedef struct nlbac_auglag_args { /* nlbac_auglag's scalar arguments */
    int n_cbf, n_clf;
    float batch_size;
    int do_lambda_update, do_backup_lambda_update, ratio_mode, backup_mode;
    float lam_lo, lam_hi;
} nlbac_auglag_args;
/* ABI 15 — the batch sums of a kind-2 / kind-3 head without the election that ends their launch: the head's workgroups
 * only leave their tile partials (nlbac_dy_head::sums_defer), and a LATER data-backward launch finishes them as a job of
 * one of its workgroups (nlbac_dy_head::finish[j]: workgroup (tile j, net 0) after its own work) — the same sums in the
 * same order, the same outputs (kind 2: out[0..2] (, out_x[0]); kind 3: nlbac_actor_scalars via `actor`). */
typedef struct nlbac_head_sums {
    int kind;                  /* 0: no job; 2 / 3: the kind of the head whose partials these are; 4 (ABI 16): commit the
                                  augmented-Lagrangian step an earlier launch ran on a private copy of the scalars block
                                  (nlbac_dy_head::cb_defer left it in cb_stage = `partials` here): the entries the step
                                  changes are copied to `sc` — the one place its new multipliers / rho become visible */
    int n_nets;                /* kind 2: the nets of that launch (3 or 4); kind 3: its n_prob */
    const float *partials;     /* that head's `partials` */
    const unsigned *n_tiles;   /* that head's `sums_tiles`: the tile count of the kernel that served it */
    float mul; float *out; float *out_x;           /* kind 2 */
    int B_norm; nlbac_actor_scalar_args actor;      /* kind 3 */
    float *sc;                                      /* kind 4 (partials = the staged block, NLBAC scalars-block sized; n_tiles,
                                                       n_nets unused) */
} nlbac_head_sums;
typedef struct nlbac_dy_head {
    int kind, B_norm;
    /* 1 */
    const float *heads; int heads_ld; const float *eps; const float *scale; int n_u;
    const float *da[3]; int da_ld[3];
    const float *alpha; /* kinds 1-3: temperatures (per controller) */
    float dlogp_mul; float *dheads; int dheads_ld;
    /* 2 */
    const float *q1t, *q2t, *lt, *nlogp, *reward, *constraint, *mask; int rcm_ld;
    const float *q[3]; float gamma; float *dq[3]; float *next_q, *next_l;
    /* 2, optional 4th net (the learned-barrier copies' BarrierNet, nlbac_td_value: NU/sac_cbf_clf/sac_cbf_clf.py:224-233):
     * target xsig + mask * gamma * xt against xq; its loss * mul goes to out_x[0] */
    const float *xt, *xsig; int xsig_ld; const float *xq; float *dxq; float *out_x;
    /* 3 */
    const float *qa, *qb, *logp; float *dqa, *dqb; int n_prob;
    nlbac_actor_scalar_args actor;
    /* 2, 3 */
    float *partials; unsigned *ticket; float mul; float *out;
    /* 3, optional (ABI 7; cb_kind != 0): the net right behind the Q pairs (index 2 n_prob) takes its dL/dy from the
     * task's constraint backward, evaluated by its workgroups for their rows instead of by a launch of its own.
     * cb_kind 1 = nlbac_unicycle_constraints_bwd (same arithmetic, same outputs left in memory: cb_dps_next (2B, 2)
     * and cb_dV (B) = that net's dL/dy; io[2 n_prob].dy is not read). */
    int cb_kind, cb_nh;
    const float *cb_ps_next, *cb_matr, *cb_bmatr, *cb_hazards, *cb_sc;
    float cb_dt, cb_batch;
    float *cb_dps_next, *cb_dV;
    /* ABI 15 (see nlbac_head_sums).  sums_defer != 0 (kinds 2, 3): partial sums only — plain stores, `ticket` unused —
     * and the launch's tile count in sums_tiles[0]; finish[j].kind != 0 (any kind of head): this launch finishes the
     * partials an EARLIER launch on the stream left. */
    int sums_defer; unsigned *sums_tiles;
    nlbac_head_sums finish[3];
    /* ABI 16 — cb_defer != 0 (with cb_kind): the constraint head of the forward launch left its tiles' column sums
     * (nlbac_gauss_head::cf_defer) and ran no augmented-Lagrangian step: every workgroup of the net behind the Q pairs
     * sums them (cb_partials, cb_tiles = that head's cf_partials / cf_tiles) and runs nlbac_auglag with cb_auglag on a
     * PRIVATE copy of the scalars block cb_sc, which this launch only reads; the loss coefficients it needs come from
     * that copy.  The workgroup of the net's first tile leaves the stepped copy in cb_stage (a scalars-block sized
     * array); a finish job of kind 4 in a LATER launch commits it to the block. */
    int cb_defer; const float *cb_partials; const unsigned *cb_tiles; nlbac_auglag_args cb_auglag; float *cb_stage;
} nlbac_dy_head;
/* policy_loss_1, alpha_loss into sc; d alpha_loss / d log_alpha into g_log_alpha
 * (sac_cbf_clf.py:292-308) for problems first_problem .. first_problem+P-1 (0 primary, 1 backup);
 * partials is the base of all problems, log_alpha / g_log_alpha point at first_problem's entry (stride between). */
int nlbac_actor_scalars(const float *partials, int n_blk, int B, int first_problem, int P, float target_entropy,
                        const float *log_alpha, int log_alpha_stride, float *g_log_alpha, float *sc,
                        nlbac_stream_t s);
/* sc[SC_ALPHA+p] = exp(log_alpha[p*stride])  (sac_cbf_clf.py:299,308) */
int nlbac_alpha_refresh(const float *log_alpha, int log_alpha_stride, int first_problem, int P, float *sc,
                        nlbac_stream_t s);

/* Unicycle geometry: obs -> state (atan2 in fp64 then cast, dynamics.py:53-58) and look-ahead
 * p(x) = xy + l_p (cos th, sin th) (sac_cbf_clf.py:429-437, 455-469). */
int nlbac_unicycle_state(const float *obs, int obs_ld, int B, float l_p,
                         float *state /*(n_copies*B,3): the B states repeated n_copies times (one per controller)*/,
                         int n_copies, float *ps /*(B,2) or NULL*/, nlbac_stream_t s);
int nlbac_unicycle_lookahead(const float *x /*(n,3)*/, int n, float l_p, float *ps /*(n,2)*/, nlbac_stream_t s);
int nlbac_unicycle_lookahead_bwd(const float *x, const float *dps, const float *dps2 /*or NULL*/, int n,
                                 float l_p, float *dx /*(n,3)*/, nlbac_stream_t s);

/* CBF/CLF terms, relu filter and column partial sums (sac_cbf_clf.py:471-504, 596-621).
 * ps (B,2); ps_next (2B,2): primary rows then backup rows.  matr (B,n_hz+1) and bmatr (B,n_hz)
 * are kept for the backward.  partials [ceil(B/256)][2*n_hz+1]. */
/* Every *_constraints_fwd takes a trailing (fused, ticket, sc): with fused != NULL (single GPU) its last workgroup
 * runs nlbac_auglag itself on the partial sums of this launch (ticket: a zeroed uint32, left zeroed) — one launch
 * less per update, same arithmetic. */
int nlbac_unicycle_constraints_fwd(const float *ps, const float *ps_next, const float *V,
                                   const float *V_next, const float *hazards, int n_hz, float r_coll,
                                   float dt, float gamma_b, float gamma_l, int B, float *matr,
                                   float *bmatr, float *partials, const nlbac_auglag_args *fused,
                                   unsigned *ticket, float *sc, nlbac_stream_t s);
/* required_matrix, ratio, lambda update (clamp [lam_lo,lam_hi]), rho *= 1.0005 (cap 200), loss values and
 * loss coefficients, primary then backup (sac_cbf_clf.py:502-528, 619-638).
 * ratio_mode: 0 none (NU), 1 plain (U), 2 clamp at 0.002 (C/P/NP).
 * backup_mode: 0 no backup controller (NU/NP: partials have n_cbf+n_clf columns), 1 backup shares augmented_term
 * with the primary (U/C), 2 backup keeps its own (P); with a backup, partials have 2*n_cbf+n_clf columns. */
int nlbac_auglag(const float *partials, int n_blk, int n_cbf, int n_clf, float batch_size,
                 int do_lambda_update, int do_backup_lambda_update /* P updates them on different schedules */,
                 int ratio_mode, int backup_mode, float lam_lo, float lam_hi, float *sc, nlbac_stream_t s);
/* d ps_next (2B,2) [CBF part] and dV_next (B) from the coefficients in sc. */
int nlbac_unicycle_constraints_bwd(const float *ps_next, const float *matr, const float *bmatr,
                                   const float *hazards, int n_hz, float dt, float batch_size, int B,
                                   const float *sc, float *dps_next, float *dV_next, nlbac_stream_t s);

/* SimulatedCars: obs <-> state scaling (C/sac_cbf_clf/dynamics.py:59-62, 88-91), relative-degree-2 CBFs between
 * cars 3-4 and 4-5 plus the CLF term over a two-step rollout (C/sac_cbf_clf/sac_cbf_clf.py:474-511, 618-645).
 * state (B,10); x1/x2 (2B,10) = rollout after one / two steps, primary rows then backup rows.
 * matr (B,3) = [cbf23, cbf34, clf], bmatr (B,2); partials [ceil(B/256)][5]; use nlbac_auglag(n_cbf=2). */
int nlbac_cars_state(const float *obs, int obs_ld, int n, float *state, nlbac_stream_t s);
/* nlbac_cars_state plus the inputs of the two-step rollout in one launch: y0_2 (2B,10) = state twice (primary /
 * backup rows), c1 (2B,2) = [pi2 (the two controllers' actions), t], c2[:, 1] = next_t (c2[:, 0] is the re-sampled
 * second action, written later) — C/sac_cbf_clf/sac_cbf_clf.py:424-437, 568-581. */
int nlbac_cars_rollout_inputs(const float *mb, int ld, int t_col, int nt_col, const float *pi2, int B,
                              float *state, float *y0_2, float *c1, float *c2, nlbac_stream_t s);
int nlbac_cars_obs(const float *state, int n, float *obs, nlbac_stream_t s);
int nlbac_cars_constraints_fwd(const float *state, const float *x1, const float *x2, const float *V,
                               const float *V1, float gamma_b, float gamma_l, float radius, int B,
                               float *matr, float *bmatr, float *partials, const nlbac_auglag_args *fused,
                               unsigned *ticket, float *sc, nlbac_stream_t s);
int nlbac_cars_constraints_bwd(const float *matr, const float *bmatr, float gamma_b, float batch_size, int B,
                               const float *sc, float *dx1, float *dx2, float *dV1, nlbac_stream_t s);
/* dst[row][col0+c] += src[row][c] */
int nlbac_add_cols(float *dst, int dst_ld, int col0, const float *src, int src_ld, int ncols, int n,
                   nlbac_stream_t s);
/* ABI 12 — dst (n x ld, dense) <- (dst + src on columns [col0, col0 + ncols) of rows < n_src) + add (n x ld): nlbac_add_cols
 * and the nlbac_axpby(1, dst, 1, add) behind it as one launch (SimulatedCars: the gradient w.r.t. x_t+1 from the
 * constraints, from V(x_t+1) and from the second solve; C/sac_cbf_clf.py:412-555 autograd accumulates the same three) */
int nlbac_add_cols_plus(float *dst, int ld, int col0, const float *src, int src_ld, int ncols, int n_src,
                        const float *add, int n, nlbac_stream_t s);

/* Learned barrier certificate (NU = neural_barrier_certificate/.../Unicycle_RL_training).
 * td_value: y = signal + mask*gamma*next_target, dpred = 2 (pred - y)/B_norm, per-block squared-error partials
 *   (BarrierNet TD step, NU/sac_cbf_clf/sac_cbf_clf.py:224-233).
 * unicycle_obs_fwd/bwd: get_obs(x') = [x, y, cos, sin, compass(2), exp(-dist to goal)] and its transpose-Jacobian
 *   product into dx (n,3) (NU/sac_cbf_clf/dynamics.py:92-135; used differentiably at sac_cbf_clf.py:408).
 * barrier_constraints: matr (B,2) = [-(B(obs',a') - B(obs,a)) - gamma_b B(obs,a), (V' - V)/dt + gamma_l V],
 *   partials [ceil(B/256)][2]; bwd gives dB(obs',a') and dV' from the coefficients in sc (:412-440). */
int nlbac_td_value(const float *next_target, const float *signal, int sig_ld, const float *mask, int mask_ld,
                   const float *pred, float gamma, int B, int B_norm, float *dpred, float *next_out /*or NULL*/,
                   float *partials, unsigned *ticket, float mul, float *out, nlbac_stream_t s);
int nlbac_unicycle_obs_fwd(const float *x /*(n,3)*/, int n, float goal_x, float goal_y, float *obs, int obs_ld,
                           nlbac_stream_t s);
int nlbac_unicycle_obs_bwd(const float *x, const float *dobs, int dobs_ld, int n, float goal_x, float goal_y,
                           float *dx /*(n,3)*/, int accumulate, nlbac_stream_t s);
int nlbac_barrier_constraints_fwd(const float *Bv, const float *Bn, const float *V, const float *Vn, float dt,
                                  float gamma_b, float gamma_l, int B, float *matr, float *partials,
                                  const nlbac_auglag_args *fused, unsigned *ticket, float *sc, nlbac_stream_t s);
int nlbac_barrier_constraints_bwd(const float *matr, float dt, float batch_size, int B, const float *sc,
                                  float *dBn, float *dVn, nlbac_stream_t s);

/* Pvtol (P = NLBAC_pvtol_RL_training/Pvtol_RL_training): dynamic state x6 = [x, y, theta, vx, vy, thrust] plus the
 * safety operator's x-position op, which follows the vehicle: op' = op + follow (x' - op) (P/sac_cbf_clf.py:462-470).
 * pvtol_state: get_state (P/sac_cbf_clf/dynamics.py:50-66).  pvtol_obs_fwd/bwd: get_obs of [x6, op'] (:97-153), obs
 *   (n,11) = [x, y, cos, sin, vx, vy, thrust, op', compass(2), exp(-dist)]; op_prev has op_rows rows (row % op_rows);
 *   bwd is the transpose-Jacobian product into dx (n,6) including the op' -> x path.
 * pvtol_constraints: relative-degree-3 CBFs over the positions at t..t+3 (5 hazards, 2 operator distances, y_max,
 *   y_min) and the CLF term (V1 - V)/1 + gamma_l V (P/sac_cbf_clf.py:543-690, 880-1010).  x1/x2/x3 (NP*B,6): primary
 *   rows, then backup rows when NP == 2.  matr (B,10), bmatr (B,9); partials [ceil(B/256)][10 or 19]. */
int nlbac_pvtol_state(const float *obs, int obs_ld, int n, float *st6, float *op /*or NULL*/, nlbac_stream_t s);
int nlbac_pvtol_obs_fwd(const float *x6, const float *op_prev, int op_rows, float follow, float goal_x, float goal_y,
                        int n, float *obs, int obs_ld, float *op_out /*or NULL*/, nlbac_stream_t s);
int nlbac_pvtol_obs_bwd(const float *x6, const float *dobs, int dobs_ld, float follow, float goal_x, float goal_y,
                        int n, float *dx, int accumulate, nlbac_stream_t s);
int nlbac_pvtol_constraints_fwd(const float *st6, const float *op0, const float *x1, const float *x2, const float *x3,
                                const float *V, const float *V1, const float *hazards, int n_hz, float r_coll,
                                float d_op, float y_max, float y_min, float follow, float gamma_b, float gamma_l,
                                int B, int NP, float *matr, float *bmatr, float *partials,
                                const nlbac_auglag_args *fused, unsigned *ticket, float *sc, nlbac_stream_t s);
int nlbac_pvtol_constraints_bwd(const float *matr, const float *bmatr, const float *x1, const float *x2,
                                const float *x3, const float *hazards, int n_hz, float follow, float gamma_b,
                                float batch_size, int B, int NP, const float *sc, float *dx1, float *dx2, float *dx3,
                                float *dV1, nlbac_stream_t s);

/* nn.MSELoss('mean') over (n,d): dpred and per-block squared-error partials [ceil(n/256)] (model.py:256). */
int nlbac_mse_fwd_bwd(const float *pred, int pred_ld, const float *target, int target_ld, int n, int n_norm,
                      int d, float *dpred, int dpred_ld, float *partials, nlbac_stream_t s);

/* ------------------------------------------------------------------------
 * Control-affine NODE field  k = f(x) + g(x) u  (model.py:208-217) and the
 * explicit RK algebra of odeint on t=[t0,t1] (euler, rk4 3/8-rule, dopri5).
 * Rows are P problems x rows_per_problem; stage derivatives are stage-major
 * K[stage][row][n_s].  A step size is per problem: h_host[P] (host floats,
 * passed by value) unless h_dev != NULL (device doubles, stride in doubles).
 * ---------------------------------------------------------------------- */
int nlbac_affine_combine_fwd(const float *f, const float *g, const float *u, int n_s, int n_u, int n,
                             float *k, nlbac_stream_t s);
/* dg[r][c] = dk[r] u[c] (dg may be NULL); du[c] (+)= mul * sum_r g[r][c] dk[r] (du may be NULL); df == dk */
int nlbac_affine_combine_bwd(const float *dk, const float *g, const float *u, int n_s, int n_u, int n,
                             float mul, float *dg, float *du, int accumulate_du, nlbac_stream_t s);
/* out = (y0 ? y0 : 0) + sum_{j<n_k} (coef[j]*h_p) K[j] */
int nlbac_rk_combine(const float *y0, const float *K, int n_k, const float *coef, const float *h_host,
                     const double *h_dev, int h_dev_stride, int P, int rows_per_problem, int n_s,
                     float *out, nlbac_stream_t s);
/* dY = dYup + dXf + dXg (any may be NULL; dX* have leading dim dx_ld);
 * dy0 (+)= dY (dy0 may be NULL); dK[j] += coef[j]*h_p*dY for j<n_k */
int nlbac_rk_stage_bwd(const float *dYup, const float *dXf, const float *dXg, int dx_ld, int n_k,
                       const float *coef, const float *h_host, const double *h_dev, int h_dev_stride,
                       int P, int rows_per_problem, int n_s, float *dK, float *dy0, int accumulate_dy0,
                       float *du_ext /* (n,n_ext) += dXf[:, n_s:n_s+n_ext]; NULL for the affine field */, int n_ext,
                       nlbac_stream_t s);
/* Fused RK step: one launch evaluates stages [stage_begin, stage_end) of an explicit RK step with
 * n_stages_total stages for P problems x rows_per_problem rows: stage inputs
 * Y_s = y0 + h_p sum_{j<s} beta[s][j] K_j, f_net and g_net (two wave groups of one workgroup),
 * K_s = f + g u, and optionally out = y0 + h sum c_out[j] K_j and err = h sum c_err[j] K_j.
 * K / Y / G are [n_stages_total][n][.] stage-major; stages < stage_begin are read from K (FSAL, f0).
 * acts_* ([layer][n_stages_total*n][hid], layer stride *_ls) may be NULL when no backward follows.
 * acts_bits != 0: the acts_* buffers receive bit-packed ReLU masks instead — uint32 words
 * [layer][n_stages_total*n][ceil(hid/32)] (bit c of word t = unit 32 t + c is active), layer stride in words: all a
 * backward without weight gradients needs, at 1/32 of the HBM traffic. */
/* Device-driven dopri5 step chain (optional, NULL = one self-contained launch).  A solve's accepted steps live in
 * STEP SLOTS — identical buffer layouts `slot_floats` floats apart; every K / Y / G / acts / err / dK / dG / dz /
 * dy0 / dYup pointer a launch is given is slot 0's.  ctl (the NLBAC_DOPRI_CTL control blocks): a problem whose `done`
 * is set is skipped, otherwise the launch works in slot ctl[12] (= accepted steps so far); a slot > 0 starts from its
 * predecessor's last stage (y1 and, FSAL, its derivative).  norm_mode 0 / 1 / 2: the scaled norms of
 * nlbac_dopri_norm_control and the step controller run in the launch's own epilogue (last workgroup per problem;
 * partials: P * ceil(rows_per_problem / 32) * 2 floats, tickets: P zeroed uint32) — an attempted step is ONE launch
 * and a fixed number of attempts can be enqueued without the host looking at any of them.  The controller records an
 * accepted step's size in hslots[p * n_slots + slot]; it stops a solve that runs out of slots (ctl[13], with done).
 * Needs rows_per_problem % 32 == 0 (or P == 1).  The backward walks the slots from each problem's last one:
 * launch back_idx differentiates slot ctl[12] - back_idx. */
typedef struct nlbac_rk_chain {
    const double *ctl;
    long slot_floats;
    int norm_mode; /* -1: none */
    int n_slots;
    float rtol, atol;
    double t_end;
    float *partials;
    unsigned *tickets;
    double *ctl_w;  /* the control blocks the fused controller updates (normally == ctl) */
    double *hslots; /* [P][n_slots] accepted step sizes */
    double *alog;   /* or NULL: attempt log [P][alog_cap][3] = (step size tried, error ratio, accepted) */
    int alog_cap;
    double *ctl_host; /* or NULL: pinned HOST memory, [P][NLBAC_DOPRI_CTL]; nlbac_dopri_norm_control's controller leaves
                         a copy of each block it updates there (no copy launch between the decision and the host) */
    /* ABI 6 — the interpolation at t_end inside the RK launches (nlbac_rk_interp_ok(f, g) != 0: the register-resident
     * kernels), instead of nlbac_dopri_interp_fwd / _bwd as launches of their own:
     *   forward (an attempt launch, stages 1..6): interp_out (n, n_s) or NULL — an attempt whose step reaches t_end
     *     (ctl: t + h >= t_end) also writes the interpolant at t_end for its rows, and with interp_kind == 1 the
     *     out-map's look-ahead point (interp_l, interp_p (n, 2)); a rejected attempt's values are overwritten by the
     *     attempt that is accepted.  Same arithmetic as nlbac_dopri_interp_fwd: same bits.
     *   backward (launch back_idx 0): interp_bwd != 0 — dK / dy0 / dy1 of each problem's last step are formed by the
     *     launch itself from interp_dout (n, n_s) = d loss / d y(t_end), or with interp_kind == 1 from interp_dp
     *     (+ interp_dp2, may be NULL) and the solve's output interp_x, as nlbac_dopri_interp_bwd would (dYup and the
     *     contents of dK / dy0 of that slot are then not read). */
    float *interp_out;
    int interp_kind; float interp_l; float *interp_p;
    int interp_bwd;
    const float *interp_dout, *interp_dp, *interp_dp2, *interp_x;
    /* ABI 10 — ctl_seq > 0: the copy in ctl_host is written as a sequence lock the host can poll instead of waiting on
     * an event behind the launch: slot 15 of the HOST block <- -ctl_seq, the other fields, slot 15 <- +ctl_seq, with
     * system-scope release fences between (a reader takes slot 15, the block, slot 15 again and accepts equal positive
     * stamps).  0: plain copy. */
    double ctl_seq;
    /* ABI 11 — the fused norm of a launch without its election (the kernels of nlbac_rk_interp_ok(f, g), g != NULL).
     * norm_defer != 0 (with a norm_mode): the launch's epilogue only leaves its tiles' partial sums in `partials`
     *   (plain stores; tickets is not used, the control block is not touched).  Modes 0 / 1 are finished by the next
     *   launch (norm_pre), mode 2 by nlbac_dopri_control_tiles.
     * norm_pre = 1 + mode (1: mode 0, 2: mode 1), in the NEXT launch on the stream: every workgroup sums
     *   partials_pre (= the previous launch's `partials`; not this launch's) for its problem in the fixed order, runs
     *   that mode's controller and takes ITS step size (mode 0: the first guess, mode 1: the initial step) instead of
     *   h_host / h_dev; the problem's first tile updates ctl_w exactly as the fused / separate controller would.
     *   With interp_out the reach-t_end test uses that step size.  Same arithmetic in the same order: same bits. */
    int norm_defer, norm_pre;
    const float *partials_pre;
} nlbac_rk_chain;
/* 1 when the fused RK kernels that serve these nets evaluate nlbac_rk_chain::interp_* (g == NULL: the single-net
 * kernels of nlbac_concat_rk_fwd / _bwd, which take interp_out / interp_dout only — no out-map) */
int nlbac_rk_interp_ok(const nlbac_mlp *f, const nlbac_mlp *g);
/* Optional: the solve's initial state is FORMED by the launch that evaluates stage 0 (and written to y0 for the launches
 * that follow) instead of by a launch of its own.  kind 1 = the Unicycle tasks' state map (sac_cbf_clf.py:400-408
 * `get_state`: (x, y, arctan2(sin, cos)) of observation row `row % rows_per_problem`, float64 arctan2 as the reference's
 * numpy call) and, when ps is given, the look-ahead point of that state (same arithmetic as nlbac_unicycle_state). */
typedef struct nlbac_in_map {
    int kind;            /* 0: none */
    const float *obs; int obs_ld;
    float l;             /* look-ahead distance */
    float *ps;           /* or NULL: (rows_per_problem, 2) */
} nlbac_in_map;
int nlbac_node_rk_fwd(const nlbac_mlp *f, const nlbac_mlp *g, const float *y0, const float *u, int P,
                      int rows_per_problem, int stage_begin, int stage_end, int n_stages_total,
                      const float *beta, const float *c_out, int n_out, const float *c_err, int n_err,
                      const float *h_host, const double *h_dev, int h_dev_stride, float *K, float *Y,
                      float *G, float *acts_f, long acts_f_ls, float *acts_g, long acts_g_ls, int acts_bits,
                      float *out, float *err, const nlbac_rk_chain *chain,
                      const nlbac_in_map *in_map /* or NULL; y0 is then written, not read */, nlbac_stream_t s);
/* ABI 9 — the three launches that open a dopri5 solve on the device-driven chain (stage 0 + Hairer's first guess; the
 * probe f(y0 + h0 f0) + the initial step size; the first attempted step, stages 1..6) as ONE persistent launch: the same
 * kernel code three times with a per-problem wait in between (the workgroup that runs a phase's fused controller
 * releases the others).  nlbac_node_rk_fwd_begin_ok: the register-resident kernels serve the nets, tiles do not straddle
 * problems and P * rows_per_problem <= 8192 (every workgroup must be resident at once).  Measured: 108 us against 105 us for the
 * three launches at 8192 rows — the host side uses it only on request (NLBAC_NODE_PERSIST=1).
 * Arguments as the three nlbac_node_rk_fwd calls it replaces: beta = dopri5's [7][7], c_err / err = the error estimate
 * of the attempt, acts_* = the mask-word buffers (acts_bits), chain = the attempt's description (ctl == ctl_w: the control
 * blocks; partials / tickets serve the fused norms of the first two phases; interp_* as for any attempt), in_map for
 * stage 0.  gen: P uint32 (zero before their first use), target: a value >= 1 not used on these words before, nor
 * target + 1.  The attempt's norm + controller remain the caller's nlbac_dopri_norm_control launch.  A workgroup that
 * waits ~1 s in vain sets ctl[13] (C_OVF) = 3 and the launch ends: results are then invalid. */
int nlbac_node_rk_fwd_begin_ok(const nlbac_mlp *f, const nlbac_mlp *g, int P, int rows_per_problem);
int nlbac_node_rk_fwd_begin(const nlbac_mlp *f, const nlbac_mlp *g, float *y0, const float *u, int P,
                            int rows_per_problem, const float *beta, const float *c_err, int n_err, float *K, float *Y,
                            float *G, float *acts_f, long acts_f_ls, float *acts_g, long acts_g_ls, float *err,
                            const nlbac_rk_chain *chain, const nlbac_in_map *in_map, unsigned *gen, unsigned target,
                            nlbac_stream_t s);
/* uint32 words per row and layer of the bit-packed ReLU masks (acts_bits) that nlbac_node_rk_fwd writes and
 * nlbac_node_rk_bwd reads for net `which` (0: f, 1: g) of this pair: ceil(hid / 32) for the LDS-tiled kernels, 4 (one
 * word per lane quarter) when the pair runs on the register-resident kernels. */
int nlbac_node_rk_mask_words(const nlbac_mlp *f, const nlbac_mlp *g, int which);
/* Fused backward of the same step (exact gradient of the discrete step): processes stages st_hi-1 .. st_lo.
 * In/out dK [n_stages_total][n][n_s] holds dL/dK_j (initialised by the caller from the step's output
 * combination / interpolant); dYup (may be NULL) is dL/d(stage input) of the last stage (FSAL y1);
 * dy0 (+= when dy0_in) and du (+= when du_acc) receive the gradients w.r.t. the step's initial state and the
 * actions.  dx_stage0: also differentiate stage 0 w.r.t. its input (needed for dy0).  When dz_f/dz_g/dG are
 * given, every stage's pre-activation grads and d g(x) are kept for nlbac_mlp_bwd_weights. */
int nlbac_node_rk_bwd(const nlbac_mlp *f, const nlbac_mlp *g, const float *u, const float *G, int P,
                      int rows_per_problem, int n_stages_total, int st_lo, int st_hi, int dx_stage0,
                      const float *beta, const float *h_host, const double *h_dev, int h_dev_stride,
                      const float *acts_f, long acts_f_ls, const float *acts_g, long acts_g_ls,
                      int acts_bits /* as written by nlbac_node_rk_fwd; excludes dz_f/dz_g/dG */,
                      float *dz_f, float *dz_g, float *dG, float *dK, const float *dYup, float *dy0,
                      int dy0_in, float *du, int du_acc, const nlbac_rk_chain *chain, int back_idx,
                      nlbac_stream_t s);
/* The same one-launch RK step for the single-net NODE dx/dt = net([x | c]) with carried inputs c = (u, t)
 * (SimulatedCars, C/sac_cbf_clf/model.py:179-205; odeint call sites C/sac_cbf_clf/sac_cbf_clf.py:437,458,581,603,
 * C/model.py:245): n_s = net->out_dim state columns, n_c = net->in_dim - n_s carried columns (c: (rows, n_c)),
 * hidden width <= 128.  Arguments as in nlbac_node_rk_fwd / _bwd; acts hold activations (floats); dc = gradient
 * w.r.t. the carried inputs (dc_acc: 0 overwrite, 1 add to the buffer's contents).
 * norm (or NULL): the field is  dx/dt = out_mu + out_sig * net(([x | c] - in_mu) * in_isig)  — inputs normalised,
 * outputs de-normalised inside the kernels (the Quadrotor NODE of /root/reference/README.md:192; BASELINE configs[4]);
 * norm = [in_mu (in_dim) | in_isig (in_dim) | out_mu (n_s) | out_sig (n_s)] on the device.  What the weight gradients
 * then need is kept on request: Xn [stage][n][in_dim] the normalised net inputs (forward), dyn [stage][n][n_s] the
 * gradient w.r.t. the net's own output (backward, with dz). */
int nlbac_concat_rk_fwd(const nlbac_mlp *net, const float *y0, const float *c, int P, int rows_per_problem,
                        int stage_begin, int stage_end, int n_stages_total, const float *beta,
                        const float *c_out, int n_out, const float *c_err, int n_err, const float *h_host,
                        const double *h_dev, int h_dev_stride, float *K, float *Y, float *acts, long acts_ls,
                        int acts_bits /* acts hold nlbac_concat_rk_mask_words() uint32 ReLU mask words per row and layer
                                         instead of the activations (rollouts differentiated w.r.t. their inputs only) */,
                        float *out, float *err, const float *norm, float *Xn, const struct nlbac_rk_chain *chain,
                        nlbac_stream_t s);
/* 4 when the net runs on the register-resident kernels (which can keep mask words), else 0 (activations only) */
int nlbac_concat_rk_mask_words(const nlbac_mlp *net);
int nlbac_concat_rk_bwd(const nlbac_mlp *net, int P, int rows_per_problem, int n_stages_total, int st_lo,
                        int st_hi, int dx_stage0, const float *beta, const float *h_host, const double *h_dev,
                        int h_dev_stride, const float *acts, long acts_ls, int acts_bits /* excludes dz */, float *dz,
                        float *dK, const float *dYup,
                        float *dy0, int dy0_in, float *dc, int dc_acc, const float *norm, float *dyn,
                        const struct nlbac_rk_chain *chain, int back_idx, nlbac_stream_t s);
/* dopri5 step control on the device.  ctl: per problem NLBAC_DOPRI_CTL doubles
 * {h, t, ratio, accept, done, x, h0, d0, d1, d2, n_steps, h_used}.
 * norm partials [P][ceil(rows/256)][2]; mode 0: (y0/scale, f0/scale) with a=f0;
 * mode 1: (f1-f0)/scale with a=f1,b=f0; mode 2: err/tol with a=err. */
#define NLBAC_DOPRI_CTL 16
int nlbac_dopri_norm_partials(const float *a, const float *b, const float *y0, const float *y1,
                              const float *u, int mode, float rtol, float atol, int n_s, int n_u,
                              int rows_per_problem, int P, float *partials,
                              const double *slot_ctl /* or NULL; mode 2 of a device-driven chain: a, y1 are slot 0's and
                                                        the attempt's are those of slot ctl[12]; done problems are skipped */,
                              long slot_floats, nlbac_stream_t s);
/* nlbac_dopri_norm_partials + nlbac_dopri_control in one launch (single-GPU path: no all-reduce between them).  tickets: P zeroed
 * uint32 words, left zeroed by the launch. */
int nlbac_dopri_norm_control(const float *a, const float *b, const float *y0, const float *y1, const float *u,
                             int mode, float rtol, float atol, int n_s, int n_u, int rows_per_problem, int P,
                             double t_end, float *partials, unsigned *tickets, double *ctl,
                             const struct nlbac_rk_chain *chain /* or NULL; device-driven chain, mode 2: a / y1 are
                                 slot 0's, the attempt judged is the one in slot ctl[12], finished problems are skipped,
                                 accepted step sizes / the attempt log are recorded as in nlbac_node_rk_fwd */,
                             nlbac_stream_t s);
int nlbac_dopri_control(const float *partials, int n_blk_per_problem, int mode, int n_s, int n_u,
                        int rows_per_problem, int P, double t_end, double *ctl,
                        int n_slots /* 0: no step slots; > 0: chained (a finished solve is left alone in mode 2) */,
                        double *hslots /* or NULL: [P][n_slots] accepted step sizes */,
                        double *alog /* or NULL: attempt log, see nlbac_rk_chain */, int alog_cap, nlbac_stream_t s);
/* ABI 14 — the controller of an attempted step whose RK launch ran norm mode 2 with norm_defer (its tiles' partial sums
 * are in chain->partials): sums them per problem in the fused form's order and does what nlbac_dopri_norm_control's
 * controller does with `chain` (step slots, attempt log, ctl_host / ctl_seq); finished problems are skipped.  A 64-thread
 * workgroup per problem instead of a pass over the error rows and an election. */
int nlbac_dopri_control_tiles(const struct nlbac_rk_chain *chain, int n_s, int n_u, int rows_per_problem, int P,
                              nlbac_stream_t s);
/* y(t_end) from the accepted step's stages (4th-order interpolant, x=(t_end-t)/h) and its backward
 * (writes dy0, dy1, dK[0..6]).  h and x come from the device control block `ctl` (h_used, x) when it is
 * non-NULL — hipGraph-replay safe — else from the host arrays. */
/* Optional per-row map of the solve's output, evaluated inside the two interpolation launches instead of in launches
 * of its own.  kind 1 = the Unicycle tasks' look-ahead point (sac_cbf_clf.py:439-447; same arithmetic as
 * nlbac_unicycle_lookahead / nlbac_unicycle_lookahead_bwd): forward also writes p (n, 2); backward takes
 * d loss / d p = dp (+ dp2) and the forward's output x instead of `dout` (which may then be NULL). */
typedef struct nlbac_out_map {
    int kind;            /* 0: none */
    float l;             /* look-ahead distance */
    float *p;            /* forward */
    const float *dp, *dp2, *x;   /* backward (dp2 may be NULL) */
} nlbac_out_map;
int nlbac_dopri_interp_fwd(const float *y0, const float *y1, const float *K, const float *h_host,
                           const float *x_host, const double *ctl, int P, int rows_per_problem, int n_s,
                           float *out, long slot_floats, const nlbac_out_map *map /* or NULL */, nlbac_stream_t s);
int nlbac_dopri_interp_bwd(const float *dout, const float *h_host, const float *x_host, const double *ctl,
                           int P, int rows_per_problem, int n_s, float *dy0, float *dy1, float *dK,
                           long slot_floats, const nlbac_out_map *map /* or NULL */, nlbac_stream_t s);
/* (slot_floats != 0: y1 / K resp. dy0 / dy1 / dK are slot 0's pointers of a device-driven chain and the step
 * interpolated is the one in slot ctl[12]; its y0 is the predecessor slot's y1.) */

/* ------------------------------------------------------------------------
 * odeint_adjoint (torchdiffeq 0.2.3 OdeintAdjointMethod, torchdiffeq/_impl/adjoint.py; pinned by the reference at
 * README.md:33, never called by it — would-be call sites P/sac_cbf_clf/sac_cbf_clf.py:459,499,534,774,812,852 and
 * P/sac_cbf_clf/model.py:259, BASELINE configs[3]).  The augmented state is kept per row as
 *     z = [ y (n_s) | a_x (n_s) | a_u (n_u) ],   W = 2 n_s + n_u floats,
 * integrated in s = t1 - t:  dy/ds = -(f + g u),  da_x/ds = (d(f + g u)/dx)^T a_x,  da_u/ds = g^T a_x.
 * nlbac_node_adj_step evaluates stages [st_lo, st_hi) of one explicit RK step of that system in ONE launch; every
 * stage re-computes f_net / g_net on its stage input and back-propagates a_x through them at once (nothing of the
 * forward solve is read).  KZ [n_stages_total][n][W] stage derivatives (stages < st_lo are read: FSAL / f0);
 * Z1 = Z0 + h sum c_out[j] KZ_j and ERR = h sum c_err[j] KZ_j when given.  ctl (or NULL): the dopri control blocks —
 * rows of problems whose `done` is set are left untouched, so a fixed chain of attempts can be enqueued without a
 * host decision per attempt.  With ZS / dG / acts_* / dz_* (all or none; P = 1) the stage inputs, the output-layer
 * gradient of g_net, the activations and the pre-activation gradients of every evaluated stage are written
 * ([stage][n][.] resp. [layer][n_stages_total*n][hid]) for nlbac_mlp_bwd_weights with x0 = ZS (ld W),
 * dy = ZS + n_s (ld W) for f_net and dy = dG for g_net: the parameter adjoint's stage derivative.  Without them the
 * ReLU masks live in LDS and the launch moves 2 W floats per row.
 * ---------------------------------------------------------------------- */
int nlbac_node_adj_step(const nlbac_mlp *f, const nlbac_mlp *g, const float *u, int P, int rows_per_problem,
                        int st_lo, int st_hi, int n_stages_total, const float *beta, const float *c_out, int n_out,
                        const float *c_err, int n_err, const float *h_host, const double *h_dev, int h_dev_stride,
                        const double *ctl, const float *Z0, float *KZ, float *Z1, float *ERR, float *ZS, float *dG,
                        float *acts_f, long acts_f_ls, float *acts_g, long acts_g_ls, float *dz_f, float *dz_g,
                        float *interp_out /* ABI 8; or NULL */, double t_end, nlbac_stream_t s);
/* interp_out [n][W] (with ctl, an attempt launch of dopri5: st_hi == n_stages_total == 7, Z1 given; nlbac_node_adj_interp_ok
 * says whether the kernel that serves the nets evaluates it): a problem whose attempted step reaches t_end (ctl: t + h >=
 * t_end) also gets the interpolant of z at t_end written for its rows — what nlbac_dopri_interp_fwd(Z0, Z1, KZ, ctl, .., W)
 * does as a launch of its own after the accept decision; a rejected attempt's values are overwritten by the next. */
int nlbac_node_adj_interp_ok(const nlbac_mlp *f, const nlbac_mlp *g);
/* Z[row] = [y | a_x | 0]  and  (dy0, du) = (Z[:, n_s:2n_s], Z[:, 2n_s:])  (either output may be NULL) */
int nlbac_adj_pack(const float *y, const float *a_x, int n_s, int n_u, int n, float *Z, nlbac_stream_t s);
int nlbac_adj_unpack(const float *Z, int n_s, int n_u, int n, float *dy0, float *du, nlbac_stream_t s);
/* Step control of the adjoint solve: torchdiffeq's default adjoint norm (handle_adjoint_norm_) — the maximum of the
 * RMS norms of the y part [x | u], the adj_y part [a_x | a_u] and, when pnorm is given, pnorm[0] (pnorm[1] for the
 * second norm of mode 0): the largest per-tensor RMS of the parameter adjoint, from nlbac_adj_param_norm.
 * Modes as nlbac_dopri_norm_control (a = KZ[0] / KZ[1] / ERR, b = KZ[0]); partials [P][ceil(rows/256)][4].
 * tickets == NULL: partial sums only (data parallel: all-reduce, then nlbac_adj_control).  A problem whose solve is
 * done is skipped in mode 2. */
int nlbac_adj_norm_control(const float *a, const float *b, const float *Z0, const float *Z1, const float *u, int mode,
                           float rtol, float atol, int n_s, int n_u, int rows_per_problem, int P, double t_end,
                           const float *pnorm, float *partials, unsigned *tickets, double *ctl,
                           double *ctl_host /* ABI 8; or NULL: pinned HOST memory [P][NLBAC_DOPRI_CTL], the controller
                                               leaves a copy of each block it updates there (with tickets only) */,
                           double host_seq /* ABI 10; > 0: that copy as a sequence lock, see nlbac_rk_chain::ctl_seq */,
                           nlbac_stream_t s);
int nlbac_adj_control(const float *partials, int n_blk_per_problem, int mode, int n_s, int n_u, int rows_per_problem,
                      int P, double t_end, const float *pnorm, double *ctl, nlbac_stream_t s);
/* After an attempted step, on the device: rows (w floats each) of problems whose step was accepted and whose solve
 * goes on take dst0 <- src0 (z0 <- z1) and dst1 <- src1 (first stage <- last stage, FSAL; dst1 may be NULL). */
int nlbac_adj_commit(const double *ctl, int rows_per_problem, long n_rows, int w, float *dst0, const float *src0,
                     float *dst1, const float *src1, nlbac_stream_t s);
/* The parameter adjoint theta_bar (flat, the arena's layout) beside the per-row state.  K [n_stages][k_stride]: its
 * stage derivatives (nlbac_mlp_bwd_weights on what nlbac_node_adj_step kept of each stage, reduced over the slabs).
 * Forms, per PARAMETER TENSOR (segments seg_off / seg_len, device int arrays), what the step-size norm needs —
 * torchdiffeq's _mixed_norm over adj_params: pnorm[0] (and pnorm[1] in mode 0) = max over tensors of the tensor's RMS:
 *   mode 0: (th0 / scale, K[0] / scale), scale = atol + rtol |th0|;   mode 1: (K[1] - K[0]) / scale;
 *   mode 2: th1 = th0 + h sum c_sol[j] K[j] is written, pnorm[0] from h sum c_err[j] K[j] / (atol + rtol max(|th0|,|th1|)).
 * h = h_dev[0] when given else h_host[0].  pseg: 2 n_seg floats of scratch, ticket: a zeroed uint32 (left zeroed).
 * Skipped in mode 2 when ctl (problem 0) says done. */
int nlbac_adj_param_norm(int mode, const float *th0, const float *K, long k_stride, int n_stages, const float *c_sol,
                         const float *c_err, const float *h_host, const double *h_dev, const int *seg_off,
                         const int *seg_len, int n_seg, float rtol, float atol, const double *ctl, float *th1,
                         float *pseg, unsigned *ticket, float *pnorm, nlbac_stream_t s);

/* The same adjoint for the single-net NODE  dx/dt = out_mu + out_sig * net(([x | c] - in_mu) * in_isig)  with carried
 * inputs c (C/sac_cbf_clf/model.py:179-205; would-be call sites C/sac_cbf_clf/sac_cbf_clf.py:437,458,581,603; the
 * normalised form is BASELINE configs[4]'s, norm as in nlbac_concat_rk_fwd or NULL), stage by stage on the MLP entry
 * points: rows of z = [y | a_y | a_c], w = 2 n_s + n_c floats.  nlbac_concat_adj_in forms, from the stage points ZS, the
 * net's input rows Xin (n, n_s + n_c) and the cotangent of its output Ay (n, n_s) = a_y (* out_sig); after
 * nlbac_mlp_fwd (x0 = Xin -> fnet) and nlbac_mlp_bwd_data (dy = Ay -> dX), nlbac_concat_adj_out writes the stage
 * derivative KZ row = [ -(out_mu + out_sig fnet) | dX[:, :n_s] in_isig | dX[:, n_s:] in_isig ].  The parameter
 * adjoint's stage derivative is nlbac_mlp_bwd_weights on (Xin, Ay) and what the two launches kept. */
int nlbac_concat_adj_in(const float *ZS, int w, const float *c, int n_s, int n_c, const float *norm, int n,
                        float *Xin, float *Ay, nlbac_stream_t s);
int nlbac_concat_adj_out(const float *fnet, const float *dX, int n_s, int n_c, const float *norm, int n, int w,
                         float *KZ, nlbac_stream_t s);
/* ONE launch per attempted RK step of that system (ABI 5; north_star: "one fused kernel per integrator step"): stages
 * [st_lo, st_hi) of the step — per stage the stage point, the net's forward on it and the vector-Jacobian product of
 * a_y through it, i.e. the five launches above — for nets of the reference's depth (in -> hid -> hid -> hid -> out) and
 * width 64 / 100 / 128 (nlbac_concat_adj_step_ok(net) != 0; other shapes take the stage-by-stage path and this entry
 * point refuses them).  Arguments as nlbac_node_adj_step: KZ [n_stages_total][n][w] (stages < st_lo are read),
 * Z1 = Z0 + h sum c_out[j] KZ_j and ERR = h sum c_err[j] KZ_j when given, beta [n_stages_total][n_stages_total], rows of
 * problems whose `done` is set in ctl are left untouched.  With Xin [S][n][in_dim], Ay [S][n][n_s], acts / dz
 * [layer][S*n][hid] (layer stride acts_ls floats; all four or none) every evaluated stage's net inputs, output
 * cotangent, activations and pre-activation gradients are kept for nlbac_mlp_bwd_weights (the parameter adjoint's stage
 * derivative); without them nothing but KZ / Z1 / ERR is written. */
int nlbac_concat_adj_step_ok(const nlbac_mlp *net);
int nlbac_concat_adj_step(const nlbac_mlp *net, const float *c, int P, int rows_per_problem, int st_lo, int st_hi,
                          int n_stages_total, const float *beta, const float *c_out, int n_out, const float *c_err,
                          int n_err, const float *h_host, const double *h_dev, int h_dev_stride, const double *ctl,
                          const float *Z0, float *KZ, float *Z1, float *ERR, const float *norm, float *Xin, float *Ay,
                          float *acts, long acts_ls, float *dz, float *interp_out /* ABI 8, as nlbac_node_adj_step; or NULL */,
                          double t_end, nlbac_stream_t s);

/* Strided block copy of 32-bit words: block b (block_len words) from src + b*src_stride to dst + b*dst_stride —
 * a row range of a stage-major solver buffer in one launch (hands a problem's first attempted dopri5 step to its
 * own solver when the problems of a joint solve stop agreeing on accept / done). */
int nlbac_copy_blocks(const void *src, long src_stride, void *dst, long dst_stride, long block_len, long n_blocks,
                      nlbac_stream_t s);
/* Replay minibatch gather on the device (replay_memory.py:21-25): dst[r] = src[idx[r]] for n_rows rows of ld floats
 * (ld % 4 == 0; idx are int64 row numbers in [0, src_rows)). */
int nlbac_gather_rows(const float *src, long src_rows, int ld, const long *idx, long n_rows, float *dst,
                      nlbac_stream_t s);
/* Device-drawn minibatch: n_rows indices uniform on [0, src_rows) with replacement (Philox4x32-10 keyed by
 * `seed`, counter = `draw`, the caller's draw number), gathered like nlbac_gather_rows, plus n_eps N(0,1)
 * floats (the update's policy noise; eps may be NULL with n_eps 0) — one launch.  Replaces the host
 * random.sample + np.stack of replay_memory.py:21-25 when the caller does not need the host's index stream. */
int nlbac_sample_rows(const float *src, long src_rows, int ld, long n_rows, float *dst, float *eps, long n_eps,
                      unsigned long long seed, unsigned long long draw, nlbac_stream_t s);

/* ------------------------------------------------------------------------
 * Batched device simulators (row f3): n independent environments advanced by one launch, one lane each, float64 like
 * the reference's numpy simulators — U/envs/unicycle_env.py:57-152 (+ barrier signal NU/envs/unicycle_env.py:116-144),
 * P/envs/pvtol_env.py:85-216 (+ NP/envs/pvtol_env.py:144-220), C/envs/simulated_cars_env.py:66-146.  All arrays are
 * float64 device arrays (ep_step / done: int32); `state`, `ep_step`, `last_dist` / `t` are advanced in place; every
 * other array is an output of the step: obs, reward, constraint, barrier signal, the Lyapunov inputs before / after the
 * step, done, info = [goal_met | reached, number of safety violations, safety cost] per environment.
 * ---------------------------------------------------------------------- */
int nlbac_unicycle_env_step(int n, const double *params /* dt, goal x, goal y, goal size, goal reward, hazard radius,
                            l_p, barrier signal off, barrier signal on */, int max_steps, const double *hazards,
                            int n_hz, const double *action /*(n,2)*/, double *state /*(n,3)*/, int *ep_step,
                            double *last_dist, double *obs /*(n,7)*/, double *reward, double *constraint, double *signal,
                            double *center /*(n,2)*/, double *next_center /*(n,2)*/, int *done, double *info /*(n,3)*/,
                            nlbac_stream_t s);
int nlbac_pvtol_env_step(int n, const double *params /* dt, goal x, goal y, goal size, goal reward, hazard radius,
                         operator follow, barrier signal off, barrier signal on */, int max_steps,
                         const double *hazards, int n_hz, const double *action /*(n,2)*/, double *state /*(n,7)*/,
                         int *ep_step, double *obs /*(n,11)*/, double *reward, double *constraint, double *signal,
                         double *lya_pre /*(n,11): the observation before the step*/, int *done, double *info,
                         nlbac_stream_t s);
int nlbac_cars_env_step(int n, const double *params /* dt, kp, k_brake, should_keep, keep threshold, goal reward */,
                        int max_steps, const double *action /*(n)*/, double *state /*(n,10)*/, double *t, int *ep_step,
                        double *obs /*(n,10)*/, double *reward, double *constraint, double *lya_pre /*(n,4)*/,
                        double *lya_next /*(n,4)*/, int *done, double *info, nlbac_stream_t s);

/* small utilities */
int nlbac_axpby(float a, const float *x, float b, const float *y /*or NULL*/, long n, float *out, nlbac_stream_t s);
int nlbac_fill(float *p, float v, long n, nlbac_stream_t s);
int nlbac_sum_partials(const float *partials, int n_blk, int n_cols, float mul, float *out, nlbac_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* NLBAC_HIP_H */
