// Gather plans: every conv-like contraction on the hot path (Conv3D forward with optional
// folded nearest-upsample, its input gradient by parity phases, the Dense layers, the 1x1
// "column" GEMMs) is one implicit GEMM   C[m][n] = sum_taps sum_c A_gather[m][tap][c] * W[tap_w][c][n]
// described by this batch-independent plan.  Rows m enumerate (sample, ld, lh, lw) of a
// loop space; per axis  src_pre = l*s_mul + s_off + tap_off  (valid iff 0 <= src_pre < S<<s_shift,
// src = src_pre >> s_shift)  and  dst = l*o_mul + o_off.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RD_PLAN_HD __host__ __device__ __forceinline__
#else
#define RD_PLAN_HD inline
#endif

#define RD_MAX_TAPS 64
#define RD_MAX_PHASES 32

struct RdTap { int mask, delta, w, code; };

struct RdPhase {
  int L, LD, LH, LW;           // rows per sample and loop extents
  int s_mul[3];
  int s_off[3];                // source coordinate of loop index 0 before the tap offset: src_pre = l * s_mul + s_off + tap_off (host side)
  int o_mul[3], o_off[3];
  int ntaps;
  int w_off;                   // element offset added to W for this phase
  int tab;                     // first entry of this phase's row table in RdPlan::tab
  int8_t tap_off[RD_MAX_TAPS][4];   // per-axis source offset of the tap, each in [-1, 2] (host side only)
  // per-tap scalars, one 16-byte scalar load per K chunk:
  //   x = validity bits the tap needs: bit (axis*4 + off + 1) per axis
  //   y = s_shift == 0: BYTE offset of the tap relative to the row base
  //   z = weight row block: W rows start at z * w_rows_per_tap
  //   w = s_shift == 1: the three 2-bit-code shift amounts, packed sd | sh<<8 | sw<<16
  RdTap tap[RD_MAX_TAPS];
};

// Row table entry (one per row l of a sample, per phase), built on the host at plan time:
//   x = element offset of the row's source base pixel inside its sample (coords l*s_mul, >> s_shift)
//   y = validity bits: bit (axis*4 + off + 1) set iff 0 <= l*s_mul + off < S << s_shift, off in [-1,2]
//   z = element offset of the row's destination pixel inside its sample
//   w = s_shift == 1 only: 2-bit codes ((l+off)>>1) - (l>>1) + 1 at bit axis*6 + (off+1)*2, off in [-1,1]
struct RdRow { int x, y, z, w; };

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
  int interleave;              // all phases congruent: tile order is (row tile, phase) with the phase fastest, so
                               // the phases that re-read the same source rows run together
  int boxes;                   // phases = border-class boxes of ONE output grid (plan_conv_fwd_boxes): same rows as the one-phase plan
  long src_sample, dst_sample; // floats per sample of the source / destination tensor
  const RdRow* tab;            // device pointer to the row tables of all phases
  int phL[RD_MAX_PHASES];      // ph[i].L again, contiguous: a workgroup of a plan with unequal phases finds its phase by walking these
  int phT[RD_MAX_PHASES];      // ph[i].ntaps, the same way (weight gradients over border-class boxes)
  signed char tapinv[RD_MAX_PHASES][64];   // [phase][weight tap w < 64] -> index of that tap in the phase's list, -1 = not listed
  unsigned long long wmask;    // boxes: the weight taps the PARENT plan listed (bit w): the fold writes exactly these
  RdPhase ph[RD_MAX_PHASES];
};

// epilogue modes of the conv GEMM
// RD_EPI_BIAS_PN_LRELU: bias, PixelNormalization over the N channels of the row, LeakyReLU (needs BN == N)
// RD_EPI_TAPGATHER (BN == 32, last generator conv 64 -> 1 as a column GEMM over its 27 taps): the tile's columns
//   P[row][tap] are summed over the taps whose neighbour rows lie inside the tile -- over (kh,kw) when the tile holds
//   whole (h,w) planes (gq = 3 sums per row, one per kd), over kw only when it holds whole w rows (gq = 9, one per
//   (kd,kh)) -- and written as Q[plane][gq][h][w]; k_tapsum_softmax finishes the sum and the softmax
enum { RD_EPI_PLAIN = 0, RD_EPI_BIAS = 1, RD_EPI_BIAS_LRELU = 2, RD_EPI_BIAS_LRELU_DROP = 3, RD_EPI_GATE_AUX = 4,
       RD_EPI_BIAS_PN_LRELU = 5, RD_EPI_TAPGATHER = 6 };

struct RdEpi {
  int mode;
  int use_drop;                // dropout active (key valid)
  uint32_t key;                // rd_make_key(seed, stream)
  uint32_t idx_base;           // added to the flat destination index for the mask counter
  const float* bias;
  const float* aux;            // RD_EPI_GATE_AUX: activation whose sign/mask gates the result
  // split-K (small-M, large-K layers): gridDim.y = ksplit workgroups share one output tile, each writes its raw
  // partial sums to kpart[split * kstride + idx]; k_splitk_finish adds them up and applies the epilogue
  int ksplit;
  float* kpart;
  long kstride;
  float* rinv;                 // RD_EPI_BIAS_PN_LRELU: per-pixel 1/sqrt(mean(y^2)+eps), kept for the backward pass
  int gw, ghw, gq;             // RD_EPI_TAPGATHER: plane width W, plane size H*W, sums per row (3 or 9)
  // shared-centre forward (second GEMM): before the mode's own work add T[b][plane >> 1][...] to the row, where
  // plane = (offset of the row inside its sample) / addt_plane is the output hour plane; T holds one plane per PAIR
  int nametag;                 // 1: launch under the dominant launch's own kernel symbol (profiling only)
  const float* addt;
  int addt_plane;              // floats per output hour plane (2H * 2W * Cout)
  int out16;                   // bf16-operand kernels: 1 = the destination, aux and addt tensors are bf16 (storage mode)
};

// weight-gradient tiling (rdgan_gemm.hip.h: k_wgrad_gemm*): dW[tap_w*wrpt + c][n] = sum_m A_gather[m][tap][c] * dY[m][n];
// partial sums go to `partial[phase][split][RT*BR][N]` and are folded by k_wgrad_reduce (deterministic, no atomics).
// Chosen on the host (rdgan_hostplan.h: wgrad_tiling), passed to the kernels by value.
struct RdWgradTiling {
  int RT, NT;            // tiles over (tap,c) rows and over N
  int tiles_per_tap;     // SC >= BR: ceil(SC/BR); else 0
  int cw;                // c extent per tap inside a tile (BR, or padded SC < BR)
  int taps_per_tile;     // 1 or BR/cw
  int rows_per_split;    // multiple of 32
  int nsplit, nphases;   // grid = RT*NT * nsplit * nphases workgroups (1-D)
  // border-class boxes (RdPlan::boxes): phases with their own tap and row counts that share ONE set of weights.  Phase p has
  // rd_wgrad_phase_rt(T, phT[p]) row tiles and ceil(B * phL[p] / rows_per_split) splits (rows_per_split = 1 << rps_log2); its
  // workgroups and partial slabs follow those of the phases in front of it; k_wgrad_reduce_box adds up, per WEIGHT tap, the
  // slabs of every phase that lists the tap.
  int box, rps_log2, tpt_log2;
  // direct (k_wgrad_gemm only; one phase, one split): the tile goes straight to its place in the weight gradient -- `partial` is
  // then dW + w_off, ldw its leading dimension -- instead of a partial slab that k_wgrad_reduce would read and write again (the
  // Dense layer of ndomain 64: 825 MB of gradient, a 64-sample K loop)
  int direct, ldw;
};

RD_PLAN_HD int rd_wgrad_phase_rt(const RdWgradTiling& T, int ntaps) {
  return T.tiles_per_tap > 0 ? ntaps * T.tiles_per_tap : (ntaps + T.taps_per_tile - 1) >> T.tpt_log2;
}

// element offset of W[n][k] inside one tap block [N][K] of a FRAGMENT-ORDER bf16 weight image (rdgan_gemm_f16.hip.h; K % 16 == 0,
// N % 32 == 0): fragment (n / 32, k / 16) = 512 contiguous elements = one MFMA operand of a wave, lane (k % 16 / 8) * 32 + n % 32
// holds its 8 consecutive k
RD_PLAN_HD long rd_wfrag_index(int n, int k, int K) {
  return ((long)(n >> 5) * (K >> 4) + (k >> 4)) * 512 + ((((k >> 3) & 1) << 5) + (n & 31)) * 8 + (k & 7);
}

// shared-centre form (rdgan_elem.hip.h: k_weight_transform): U[u] = sum_k c[u][k] W[k], c in {-1, 0, 1}
struct RdWeightMap { int8_t c[48][27]; };
