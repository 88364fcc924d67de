// Device-resident scalars block of the agent (floats; NLBAC_SC_SIZE of them).
// Mirrors the Python-side state the reference keeps on the host:
// alpha / backup_alpha (sac_cbf_clf.py:32-33,299,308), lambda lists
// (:119-127), augmented_term (:59), plus per-update loss outputs.
// nlbac_amd/sac_cbf_clf/_layout.py holds the same constants for the host.
#pragma once

#define NLBAC_NC_MAX 16   // max constraints per controller

enum {
    SC_ALPHA = 0,        // temperature used in the losses (primary)
    SC_BALPHA = 1,       // backup
    SC_RATIO = 4,
    SC_PL2 = 5,          // policy_loss_2
    SC_BPL2 = 6,
    SC_QF1 = 7, SC_QF2 = 8, SC_LF = 9,
    SC_PL1 = 10, SC_BPL1 = 11,
    SC_ALOSS = 12, SC_BALOSS = 13,
    SC_NODE_LOSS = 14,
    SC_XLOSS = 15,       // TD loss of the extra critic-type net (BarrierNet)
    SC_LAMBDA = 16,                       // [NC_MAX]
    SC_BLAMBDA = SC_LAMBDA + NLBAC_NC_MAX,   // 32
    SC_COEF = SC_BLAMBDA + NLBAC_NC_MAX,     // 48  dLoss/d required_i
    SC_BCOEF = SC_COEF + NLBAC_NC_MAX,       // 64
    SC_REQ = SC_BCOEF + NLBAC_NC_MAX,        // 80  required_matrix
    SC_BREQ = SC_REQ + NLBAC_NC_MAX,         // 96
    SC_RHO_F64 = 112,    // double augmented_term      (floats 112,113)
    SC_BRHO_F64 = 114,   // double backup_augmented_term (Pvtol keeps its own)
    SC_MEAN_LOGP = 116, SC_MEAN_BLOGP = 117,
    NLBAC_SC_SIZE_ENUM = 128
};
