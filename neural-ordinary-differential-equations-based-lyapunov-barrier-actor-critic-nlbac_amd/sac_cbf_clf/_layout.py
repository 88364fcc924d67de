"""Offsets inside the device scalars block (mirror of csrc/scalars.h)."""
NC_MAX = 16
SC_ALPHA, SC_BALPHA = 0, 1
SC_RATIO, SC_PL2, SC_BPL2 = 4, 5, 6
SC_QF1, SC_QF2, SC_LF = 7, 8, 9
SC_PL1, SC_BPL1 = 10, 11
SC_ALOSS, SC_BALOSS = 12, 13
SC_NODE_LOSS = 14
SC_XLOSS = 15      # TD loss of the extra critic-type net (BarrierNet)
SC_LAMBDA = 16
SC_BLAMBDA = SC_LAMBDA + NC_MAX
SC_COEF = SC_BLAMBDA + NC_MAX
SC_BCOEF = SC_COEF + NC_MAX
SC_REQ = SC_BCOEF + NC_MAX
SC_BREQ = SC_REQ + NC_MAX
SC_RHO_F64, SC_BRHO_F64 = 112, 114      # doubles stored in float slots (112,113) / (114,115)
SC_MEAN_LOGP, SC_MEAN_BLOGP = 116, 117
SC_SIZE = 128
