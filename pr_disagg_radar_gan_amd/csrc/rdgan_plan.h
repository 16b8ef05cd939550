// Gather plans: every conv-like contraction on the hot path (Conv3D forward with optional
// folded nearest-upsample, its input gradient by parity phases, the Dense layers, the 1x1
// "column" GEMMs) is one implicit GEMM   C[m][n] = sum_taps sum_c A_gather[m][tap][c] * W[tap_w][c][n]
// described by this batch-independent plan.  Rows m enumerate (sample, ld, lh, lw) of a
// loop space; per axis  src_pre = l*s_mul + tap_off  (valid iff 0 <= src_pre < S<<s_shift,
// src = src_pre >> s_shift)  and  dst = l*o_mul + o_off.
#pragma once
#include <stdint.h>

#define RD_MAX_TAPS 64
#define RD_MAX_PHASES 8

struct RdPhase {
  int L, LD, LH, LW;           // rows per sample and loop extents
  int s_mul[3];
  int o_mul[3], o_off[3];
  int ntaps;
  int w_off;                   // element offset added to W for this phase
  int8_t tap_off[RD_MAX_TAPS][4];
  int16_t tap_w[RD_MAX_TAPS];  // weight row block = tap_w * w_rows_per_tap
};

struct RdPlan {
  int nphases;
  int SD, SH, SW;              // source spatial dims (before the folded upsample)
  int s_shift;                 // 1: nearest x2 upsample folded into the gather
  int s_cstride;               // floats per source pixel
  int SC;                      // channels gathered per tap (GEMM K per tap)
  int w_rows_per_tap;          // W row stride between taps
  int DD, DH, DW;              // destination spatial dims
  int d_cstride;               // floats per destination pixel
  int N;                       // GEMM N
  int pad_;
  RdPhase ph[RD_MAX_PHASES];
};

// epilogue modes of the conv GEMM
enum { RD_EPI_PLAIN = 0, RD_EPI_BIAS = 1, RD_EPI_BIAS_LRELU = 2, RD_EPI_BIAS_LRELU_DROP = 3, RD_EPI_GATE_AUX = 4 };

struct RdEpi {
  int mode;
  int use_drop;                // dropout active (key valid)
  uint32_t key;                // rd_make_key(seed, stream)
  uint32_t idx_base;           // added to the flat destination index for the mask counter
  const float* bias;
  const float* aux;            // RD_EPI_GATE_AUX: activation whose sign/mask gates the result
};
