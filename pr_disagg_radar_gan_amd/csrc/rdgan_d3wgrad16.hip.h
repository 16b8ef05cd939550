// bf16 storage mode, ndomain 16: weight gradient of the critic's third layer (backward of T:295, Conv3D(256, 3x3x3, stride 2, 'same')
// on the 6 x 4 x 4 x 128 output of layer 2 -> 3 x 2 x 2 x 256; no padding in front: source position 2 o + tap):
// dW[tap][128 ci][256 co] = sum over samples and output positions o of h2[2 o + tap][ci] * dy[o][co], in the pattern of
// k_d2_wgrad_slab16 (rdgan_d2wgrad16.hip.h).  As tiles of k_wgrad_gemm_ws16<128,128>: 0.29 ms at 6144 samples, 0.18 of the bf16 roof.
//   * a tap product is [128 x 256] = 32 MFMA tiles: a WAVE owns (tap, QUARTER of the output channels): [128 x 64], 128 accumulator
//     registers, kept over the workgroup's whole share of the batch;
//   * by input parity the 27 taps fall into 8 classes (taps 0 and 2 of an axis read the even positions 2 o and 2 o + 2, tap 1 the odd
//     positions 2 o + 1); every class is a dense 3 x 2 x 2 sub-grid of a sample's 6 x 4 x 4 positions (12 of its 96 rows) on which its
//     taps are shifts by 0 / +1.  Workgroup types = the tap groups {8}, {4,4}, {4,2,2}, {2,1} of k_d2_wgrad_slab16 x 4 channel quarters;
//   * a sample has only 12 output positions, so a work item is FOUR samples: 48 positions = 3 k-steps of 16; per stage the type's
//     class sub-grids (12 KB each) and the quarter's output-gradient rows (48 x 128 B); two stages.
// partial[group][tap][128][256], folded in a fixed order.
#pragma once
#include "rdgan_d2wgrad16.hip.h"

#define RD_D3W_S 4                                    // samples per item
#define RD_D3W_DY (RD_D3W_S * 12 * 128)               // output-gradient rows of the quarter: 48 x 128 B
#define RD_D3W_CLS (RD_D3W_S * 12 * 256)              // one class sub-grid of the item: 48 rows of 256 B
#define RD_D3W_STAGE (RD_D3W_DY + 3 * RD_D3W_CLS)
#define RD_D3W_ZERO (2 * RD_D3W_STAGE)                // a 256-byte row of zeros
#define RD_D3W_LDS (RD_D3W_ZERO + 256)
#define RD_D3W_TILE (128 * 256)                       // floats per tap

// x [B][6][4][4][128] bf16 (layer 2's output; penalty third: the second sweep's r2), dy [B][3][2][2][256] bf16 -> partial [G][27][128][256].
// grid: 16 G workgroups of 512 threads, blockIdx = g_lo + 8 (type + 16 g_hi), type = tap group * 4 + channel quarter,
// group = g_lo + 8 g_hi; group g walks items g, g + G, ... < ceil(B / 4).  Dynamic LDS RD_D3W_LDS.
__global__ void __launch_bounds__(512, 1)
k_d3_wgrad_slab16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ dy, float* __restrict__ partial, int B, int G) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int g_lo = blockIdx.x & 7, type = (blockIdx.x >> 3) & 15, group = g_lo + 8 * (blockIdx.x >> 7);
  const int tgroup = type >> 2, quarter = type & 3;
  if (tid < 64) *(float*)(lds + RD_D3W_ZERO + tid * 4) = 0.f;

  // ---- this wave's tap: the wave-th tap of the group's classes (class bit 1 = an axis with two taps, 0 and 2, on the even positions)
  int cls = -1, cslot = 0, tsel = wave, ncls = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = rd_d2w_type_classes[tgroup][k];
    if (c < 0) continue;
    const int nt = 1 << __builtin_popcount(c);
    if (cls < 0 && tsel < nt) { cls = c; cslot = ncls; }
    if (cls < 0) tsel -= nt;
    ++ncls;
  }
  const bool has_tap = cls >= 0;
  int sft[3] = {0, 0, 0}, tap = 0;
  if (has_tap) {
    int rem = tsel, t3[3];
    for (int a = 2; a >= 0; --a) {
      const int two = (cls >> (2 - a)) & 1;
      int t = 1;
      if (two) { t = 2 * (rem & 1); rem >>= 1; }
      t3[a] = t; sft[a] = t == 2 ? 1 : 0;             // tap 2 reads even position o + 1
    }
    tap = (t3[0] * 3 + t3[1]) * 3 + t3[2];
  }
  // ---- transposed-read addresses: position k = 16 kk + 8 lhalf + q4 (+ 4 for the second read) of the item = (sample k / 12,
  // o = k % 12 = (od, oh, ow) of 3 x 2 x 2); the tap's source row in its class image = sample * 12 + the shifted sub-grid position
  const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  int a_row[3][2];                 // byte offset of the row inside the class image, or -1
#pragma unroll
  for (int kk = 0; kk < 3; ++kk)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int k = 16 * kk + 8 * lhalf + q4 + 4 * rd;
      const int s = k / 12, o = k - s * 12;
      const int jd = (o >> 2) + sft[0], jh = ((o >> 1) & 1) + sft[1], jw = (o & 1) + sft[2];
      a_row[kk][rd] = (jd < 3 && jh < 2 && jw < 2) ? (s * 12 + jd * 4 + jh * 2 + jw) * 256 : -1;
    }
  const int a_colb = ((2 * g16 + (p4 >> 1)) << 4) + (p4 & 1) * 8;
  int b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
    b_off[j] = (8 * lhalf + q4) * 128 + (((j * 4 + 2 * g16 + (p4 >> 1)) ^ rd_tr_swz<128>(q4)) << 4) + (p4 & 1) * 8;

  // ---- DMA sources relative to the item's first sample, once: instruction i: i < 6: output-gradient rows 8 i .. (128 B of the quarter);
  // i >= 6: rows 4 (i - 6) .. of the concatenated class images (256 B each, gathered from layer 2's output)
  const int ndma = 6 + ncls * 12;
  int dma_off[6]; int dma_s[6];    // byte offset inside the sample, sample of the item (-1: nothing)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int i = wave + 8 * k;
    int off = 0, smp = -1;
    if (i < 6) {
      const int r = i * 8 + (lane >> 3);              // position of the item
      smp = r / 12;
      off = ((r - smp * 12) * 256 + quarter * 64) * 2 + (((lane & 7) ^ rd_tr_swz<128>(r)) << 4);
    } else if (i < ndma) {
      const int R = (i - 6) * 4 + (lane >> 4);        // row of the concatenated class images
      const int ci = R / 48, r = R - ci * 48;
      const int c = rd_d2w_type_classes[tgroup][ci];
      smp = r / 12;
      const int j = r - smp * 12, jd = j >> 2, jh = (j >> 1) & 1, jw = j & 1;
      const int srow = ((2 * jd + 1 - ((c >> 2) & 1)) * 4 + 2 * jh + 1 - ((c >> 1) & 1)) * 4 + 2 * jw + 1 - (c & 1);
      off = srow * 256 + (((lane & 15) ^ rd_tr_swz<256>(r)) << 4);
    }
    dma_off[k] = off; dma_s[k] = smp;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nitems = (B + RD_D3W_S - 1) / RD_D3W_S;
  auto load_item = [&](int item, int stage) {
    const int b0 = item * RD_D3W_S;
    const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc((const float*)(x + (long)b0 * (96 * 128)));
    const __amdgpu_buffer_rsrc_t rsY = rd_make_rsrc((const float*)(dy + (long)b0 * (12 * 256)));
    char* st = lds + stage * RD_D3W_STAGE;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = wave + 8 * k;                    // wave-uniform
      const bool ok = dma_s[k] >= 0 && b0 + dma_s[k] < B;
      if (i < 6) {
        unsigned voff = ok ? (unsigned)(dma_s[k] * (12 * 256 * 2) + dma_off[k]) : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rsY, (float*)(st + i * 1024), (int)voff, 0);
      } else if (i < ndma) {
        unsigned voff = ok ? (unsigned)(dma_s[k] * (96 * 128 * 2) + dma_off[k]) : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rsX, (float*)(st + RD_D3W_DY + (i - 6) * 1024), (int)voff, 0);
      }
    }
  };

  int item = group, stage = 0;
  if (item < nitems) load_item(item, 0);
  rd_dma_landed();
  __syncthreads();
  for (; item < nitems; item += G, stage ^= 1) {
    if (item + G < nitems) load_item(item + G, stage ^ 1);
    if (has_tap) {
      const char* st = lds + stage * RD_D3W_STAGE;
      const int cbase = RD_D3W_DY + cslot * RD_D3W_CLS;
      const int zoff = RD_D3W_ZERO - stage * RD_D3W_STAGE + a_colb;
      rd_bf16x8 fa[2][4], fb[2][2];
      auto load_frag = [&](int slot, int kk) {
        int o[2][4];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int r = a_row[kk][rd];
          const int swz = ((r >> 8) & 3) << 2;        // rd_tr_swz<256> of the row inside its 48-row class image
#pragma unroll
          for (int i = 0; i < 4; ++i) o[rd][i] = r >= 0 ? cbase + r + ((i * 64 + a_colb) ^ (swz << 4)) : zoff;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[slot][i] = rd_tr_frag(st, o[0][i], o[1][i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[slot][j] = rd_tr_frag(st, b_off[j] + kk * 16 * 128, b_off[j] + (kk * 16 + 4) * 128);
      };
      load_frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int cur = kk & 1;
        if (kk + 1 < 3) load_frag(cur ^ 1, kk + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      }
    }
    rd_dma_landed();
    __syncthreads();
  }
  if (has_tap) {
    float* o = partial + ((long)group * 27 + tap) * RD_D3W_TILE + quarter * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 256 + j * 32 + l31] = acc[i][j][r];
  }
}

// dW[i] = sum over groups of partial[g][i], i < 27 * RD_D3W_TILE (fixed order)
__global__ void __launch_bounds__(256)
k_d3_wgrad_fold(const float* __restrict__ partial, int G, float* __restrict__ dW) {
  const long i4 = blockIdx.x * 256L + threadIdx.x;
  if (i4 >= 27L * RD_D3W_TILE / 4) return;
  f32x4 s = *(const f32x4*)(partial + i4 * 4);
  for (int g = 1; g < G; ++g) s += *(const f32x4*)(partial + (long)g * 27 * RD_D3W_TILE + i4 * 4);
  *(f32x4*)(dW + i4 * 4) = s;
}
