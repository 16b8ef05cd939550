// fp32 MFMA implicit-GEMM kernels for gfx950 (CDNA4): the conv-like contractions of the
// cWGAN-GP step (gan_train_cwgangp_pixelnorm.py:286-299 critic Conv3D, :326-345 generator
// Dense / UpSampling3D+Conv3D), their input gradients and their weight gradients.
//
// v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD): lane l supplies A[i=l&31][k=l>>5]
// and B[k=l>>5][j=l&31]; D register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
// 256-thread workgroups = 4 waves (one per SIMD); operands staged global -> registers -> LDS
// (double buffered, one barrier per K chunk); A is gathered on the fly from NDHWC
// activations through an RdPlan (im2col never materialised, nearest-upsample folded in).
#pragma once
#include <hip/hip_runtime.h>
#include "rdgan_plan.h"
#include "rdgan_rng.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RD_LRELU_ALPHA 0.2f

__device__ __forceinline__ float rd_lrelu(float x) { return x > 0.f ? x : RD_LRELU_ALPHA * x; }
// slope of LeakyReLU recovered from its (possibly dropout-scaled) output: TF's LeakyReluGrad
// uses features > 0 ? 1 : alpha, and sign(output) == sign(features) for kept elements.
__device__ __forceinline__ float rd_lrelu_slope_from_out(float h) { return h > 0.f ? 1.f : RD_LRELU_ALPHA; }

struct RdRowDecode {
  int b, ld, lh, lw;
};
__device__ __forceinline__ RdRowDecode rd_decode_row(int m, const RdPhase& P) {
  RdRowDecode r;
  r.b = m / P.L;
  int rem = m - r.b * P.L;
  int hw = P.LH * P.LW;
  r.ld = rem / hw;
  rem -= r.ld * hw;
  r.lh = rem / P.LW;
  r.lw = rem - r.lh * P.LW;
  return r;
}

// ------------------------------------------------------------------------------------
// C[m][n] = sum_{tap,c} A_gather[m][tap][c] * W[tap_w*wrpt + c][n]  (+ fused epilogue)
// ------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK>
__global__ void __launch_bounds__(256)
k_conv_gemm(const RdPlan* __restrict__ plan, int B, const float* __restrict__ src,
            const float* __restrict__ W, int ldw, float* __restrict__ dst, RdEpi epi) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  constexpr int AST = BK + 4;                 // b128 fragment reads conflict-free (stride 36 / 12 dwords)
  constexpr int BST = BN + 4;
  constexpr int A_F4 = BK / 4, A_RPP = 256 / A_F4, A_P = (BM + A_RPP - 1) / A_RPP;
  constexpr int B_F4 = BN / 4, B_RPP = 256 / B_F4, B_P = (BK + B_RPP - 1) / B_RPP;
  constexpr int STAGE = BM * AST + BK * BST;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhalf = lane >> 5;

  // ---- which phase / tile
  int mt = blockIdx.x, pidx = 0;
  for (int p = 0; p < plan->nphases; ++p) {
    int nt = (B * plan->ph[p].L + BM - 1) / BM;
    if (mt < nt) { pidx = p; break; }
    mt -= nt;
  }
  const RdPhase& P = plan->ph[pidx];
  const int rows = B * P.L;
  const int m0 = mt * BM;
  const int n0 = blockIdx.y * BN;
  const int SD = plan->SD, SH = plan->SH, SW = plan->SW, sh = plan->s_shift;
  const int limD = SD << sh, limH = SH << sh, limW = SW << sh;
  const int cstride = plan->s_cstride, SC = plan->SC, wrpt = plan->w_rows_per_tap;
  const bool partial_c = (SC & 3) != 0;
  const float* Wp = W + P.w_off;

  // ---- per-thread A rows
  int rb[A_P], rc[A_P];
  const int a_c4 = (tid % A_F4) * 4;
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    int r = tid / A_F4 + i * A_RPP;
    int m = m0 + r;
    if (r < BM && m < rows) {
      RdRowDecode d = rd_decode_row(m, P);
      rb[i] = d.b;
      rc[i] = (d.ld * P.s_mul[0]) | ((d.lh * P.s_mul[1]) << 8) | ((d.lw * P.s_mul[2]) << 16);
    } else {
      rb[i] = -1; rc[i] = 0;
    }
  }
  const int b_kk = tid / B_F4, b_n4 = (tid % B_F4) * 4;

  const int CPT = (SC + BK - 1) / BK;
  const int nchunks = P.ntaps * CPT;

  f32x4 ra[A_P], rw[B_P];
  auto load_chunk = [&](int q) {
    const int tap = q / CPT, cc = q - tap * CPT;
    const int od = P.tap_off[tap][0], oh = P.tap_off[tap][1], ow = P.tap_off[tap][2];
    const int c = cc * BK + a_c4;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      int pd = (rc[i] & 255) + od, ph = ((rc[i] >> 8) & 255) + oh, pw = ((rc[i] >> 16) & 255) + ow;
      bool ok = rb[i] >= 0 && (unsigned)pd < (unsigned)limD && (unsigned)ph < (unsigned)limH &&
                (unsigned)pw < (unsigned)limW && c < SC;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        long pix = (((long)rb[i] * SD + (pd >> sh)) * SH + (ph >> sh)) * SW + (pw >> sh);
        v = *(const f32x4*)(src + pix * cstride + c);
        if (partial_c) {
          if (c + 1 >= SC) v.y = 0.f;
          if (c + 2 >= SC) v.z = 0.f;
          if (c + 3 >= SC) v.w = 0.f;
        }
      }
      ra[i] = v;
    }
    const long krow0 = (long)P.tap_w[tap] * wrpt + cc * BK;
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      int kk = b_kk + i * B_RPP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kk < BK && cc * BK + kk < SC) v = *(const f32x4*)(Wp + (krow0 + kk) * ldw + n0 + b_n4);
      rw[i] = v;
    }
  };
  auto store_chunk = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * AST;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      int r = tid / A_F4 + i * A_RPP;
      if (r < BM) *(f32x4*)&As[r * AST + a_c4] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      int kk = b_kk + i * B_RPP;
      if (kk < BK) *(f32x4*)&Bs[kk * BST + b_n4] = rw[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
    if (q + 1 < nchunks) load_chunk(q + 1);
    const float* As = smem + buf * STAGE;
    const float* Bs = As + BM * AST;
#pragma unroll
    for (int j8 = 0; j8 < BK / 8; ++j8) {
      f32x4 a[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *(const f32x4*)&As[(wm * WTM + i * 32 + l31) * AST + j8 * 8 + lhalf * 4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = Bs[(j8 * 8 + lhalf * 4 + s) * BST + wn * WTN + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    if (q + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue
  const int DD = plan->DD, DH = plan->DH, DW = plan->DW, dcs = plan->d_cstride;
  const int mode = epi.mode;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      const int m = m0 + row;
      if (m < rows) {
        RdRowDecode d = rd_decode_row(m, P);
        long pix = (((long)d.b * DD + d.ld * P.o_mul[0] + P.o_off[0]) * DH + d.lh * P.o_mul[1] + P.o_off[1]) * DW +
                   d.lw * P.o_mul[2] + P.o_off[2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * WTN + j * 32 + l31;
          const long idx = pix * dcs + col;
          float v = acc[i][j][r];
          if (mode == RD_EPI_BIAS) {
            v += epi.bias[col];
          } else if (mode == RD_EPI_BIAS_LRELU) {
            v = rd_lrelu(v + epi.bias[col]);
          } else if (mode == RD_EPI_BIAS_LRELU_DROP) {
            v = rd_lrelu(v + epi.bias[col]);
            if (epi.use_drop) v *= rd_drop_scale(epi.key, (uint32_t)idx + epi.idx_base);
          } else if (mode == RD_EPI_GATE_AUX) {
            float g = rd_lrelu_slope_from_out(epi.aux[idx]);
            if (epi.use_drop) g *= rd_drop_scale(epi.key, (uint32_t)idx + epi.idx_base);
            v *= g;
          }
          dst[idx] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// weight gradient: dW[tap_w*wrpt + c][n] = sum_m A_gather[m][tap][c] * dY[m][n]
// grid.x = r_tile * NT + n_tile, grid.y = split over rows m, grid.z = plan phase; partial sums go to
// `partial[phase][split][RT*BR][N]` and are folded by k_wgrad_reduce (deterministic, no atomics).
// ------------------------------------------------------------------------------------
struct RdWgradTiling {
  int RT, NT;            // tiles over (tap,c) rows and over N
  int tiles_per_tap;     // SC >= BR: ceil(SC/BR); else 0
  int cw;                // c extent per tap inside a tile (BR, or padded SC < BR)
  int taps_per_tile;     // 1 or BR/cw
  int rows_per_split;    // multiple of 32
};

__device__ __forceinline__ void rd_wgrad_tile_row(const RdWgradTiling& T, int BR, int rt, int r, int& tap, int& c) {
  if (T.tiles_per_tap > 0) {
    tap = rt / T.tiles_per_tap;
    c = (rt - tap * T.tiles_per_tap) * BR + r;
  } else {
    int tl = r / T.cw;
    tap = rt * T.taps_per_tile + tl;
    c = r - tl * T.cw;
  }
}

template <int BR, int BN>
__global__ void __launch_bounds__(256)
k_wgrad_gemm(const RdPlan* __restrict__ plan, int B, const float* __restrict__ src,
             const float* __restrict__ dy, float* __restrict__ partial, RdWgradTiling T) {
  constexpr int BKP = 32;
  constexpr int WTM = BR / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int AST = BR + 4, BST = BN + 4;
  constexpr int A_F4 = BR / 4, A_PPP = 256 / A_F4, A_P = BKP / A_PPP;   // positions per pass
  constexpr int B_F4 = BN / 4, B_PPP = 256 / B_F4, B_P = BKP / B_PPP;
  constexpr int STAGE = BKP * AST + BKP * BST;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhalf = lane >> 5;
  const RdPhase& P = plan->ph[blockIdx.z];
  const int rows = B * P.L;
  const int rt = blockIdx.x / T.NT, ntile = blockIdx.x - rt * T.NT;
  const int n0 = ntile * BN;
  const int mbeg = blockIdx.y * T.rows_per_split;
  const int mend = min(rows, mbeg + T.rows_per_split);
  const int SD = plan->SD, SH = plan->SH, SW = plan->SW, sh = plan->s_shift;
  const int limD = SD << sh, limH = SH << sh, limW = SW << sh;
  const int cstride = plan->s_cstride, SC = plan->SC;
  const int DD = plan->DD, DH = plan->DH, DW = plan->DW, dcs = plan->d_cstride;
  const bool partial_c = (SC & 3) != 0;

  // this thread's A column group (tap, c) is fixed over the whole loop
  int a_tap, a_c;
  const int a_r = (tid % A_F4) * 4;
  rd_wgrad_tile_row(T, BR, rt, a_r, a_tap, a_c);
  const bool a_ok = a_tap < P.ntaps && a_c < SC;
  int od = 0, oh = 0, ow = 0;
  if (a_ok) { od = P.tap_off[a_tap][0]; oh = P.tap_off[a_tap][1]; ow = P.tap_off[a_tap][2]; }
  const int b_n4 = (tid % B_F4) * 4;

  f32x4 ra[A_P], rg[B_P];
  auto load_chunk = [&](int mb) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      int m = mb + tid / A_F4 + i * A_PPP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ok && m < mend) {
        RdRowDecode d = rd_decode_row(m, P);
        int pd = d.ld * P.s_mul[0] + od, ph = d.lh * P.s_mul[1] + oh, pw = d.lw * P.s_mul[2] + ow;
        if ((unsigned)pd < (unsigned)limD && (unsigned)ph < (unsigned)limH && (unsigned)pw < (unsigned)limW) {
          long pix = (((long)d.b * SD + (pd >> sh)) * SH + (ph >> sh)) * SW + (pw >> sh);
          v = *(const f32x4*)(src + pix * cstride + a_c);
          if (partial_c) {
            if (a_c + 1 >= SC) v.y = 0.f;
            if (a_c + 2 >= SC) v.z = 0.f;
            if (a_c + 3 >= SC) v.w = 0.f;
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      int m = mb + tid / B_F4 + i * B_PPP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < mend) {
        RdRowDecode d = rd_decode_row(m, P);
        long pix = (((long)d.b * DD + d.ld * P.o_mul[0] + P.o_off[0]) * DH + d.lh * P.o_mul[1] + P.o_off[1]) * DW +
                   d.lw * P.o_mul[2] + P.o_off[2];
        v = *(const f32x4*)(dy + pix * dcs + n0 + b_n4);
      }
      rg[i] = v;
    }
  };
  auto store_chunk = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BKP * AST;
#pragma unroll
    for (int i = 0; i < A_P; ++i) *(f32x4*)&As[(tid / A_F4 + i * A_PPP) * AST + a_r] = ra[i];
#pragma unroll
    for (int i = 0; i < B_P; ++i) *(f32x4*)&Bs[(tid / B_F4 + i * B_PPP) * BST + b_n4] = rg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = (mend - mbeg + BKP - 1) / BKP;
  if (nchunks > 0) {
    load_chunk(mbeg);
    store_chunk(0);
  }
  __syncthreads();
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
    if (q + 1 < nchunks) load_chunk(mbeg + (q + 1) * BKP);
    const float* As = smem + buf * STAGE;
    const float* Bs = As + BKP * AST;
#pragma unroll
    for (int s = 0; s < BKP / 2; ++s) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[(2 * s + lhalf) * AST + wm * WTM + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[(2 * s + lhalf) * BST + wn * WTN + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (q + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  const int N = plan->N;
  float* out = partial + (((long)blockIdx.z * gridDim.y + blockIdx.y) * T.RT + rt) * BR * N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
#pragma unroll
      for (int j = 0; j < TN; ++j) out[(long)row * N + n0 + wn * WTN + j * 32 + l31] = acc[i][j][r];
    }
}

// fold the split partials and scatter rows (tap,c) to their place in the weight gradient
__global__ void k_wgrad_reduce(const RdPlan* __restrict__ plan, const float* __restrict__ partial, int nsplit,
                               RdWgradTiling T, int BR, float* __restrict__ dW, int ldw) {
  const int N = plan->N;
  const int n4s = N / 4;
  const long total = (long)T.RT * BR * n4s;
  const RdPhase& P = plan->ph[blockIdx.y];
  const long stride = (long)T.RT * BR * N;
  const float* pbase = partial + (long)blockIdx.y * nsplit * stride;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    int R = (int)(f / n4s), n = (int)(f - (long)R * n4s) * 4;
    int rt = R / BR, r = R - rt * BR, tap, c;
    rd_wgrad_tile_row(T, BR, rt, r, tap, c);
    if (tap >= P.ntaps || c >= plan->SC) continue;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* p = pbase + (long)R * N + n;
    for (int k = 0; k < nsplit; ++k) s += *(const f32x4*)(p + k * stride);
    *(f32x4*)(dW + P.w_off + ((long)P.tap_w[tap] * plan->w_rows_per_tap + c) * ldw + n) = s;
  }
}
