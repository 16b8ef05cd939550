// bf16 storage mode, ndomain 16: weight gradient of generator block 2 in the collapsed form (backward of T:335-336:
// dWc[phase * 8 + tap][256 ci][128 co] over the 6 x 4 x 4 x 256 block input and the 12 x 8 x 8 x 128 output gradient), the slab kernel
// of block 3 (rdgan_upwgrad16.hip.h) on this block's geometry.  A [256 x 128] tap product is 32 MFMA tiles, so a workgroup owns
// (phase, QUARTER of the input channels): eight waves = eight taps, [64 ci x 128 co] = 8 tiles = 128 accumulator registers each.
// Work item = one sample: the seven source planes the phase's taps touch (6 + one halo; 16 positions x 64 channels of the quarter =
// 2 KB each) and the phase's 96 output-gradient rows (256 B each, 24 KB) per stage; a 16-position k-step is one source plane.
// 32 workgroup types x G groups; the types of a group share an XCD.  partial[group][phase * 8 + tap][256][128].
#pragma once
#include "rdgan_upwgrad16.hip.h"

#define RD_UW2_XPLANE 2048                           // 16 rows of 128 B (one channel quarter)
#define RD_UW2_DY (7 * RD_UW2_XPLANE)                // offset of the output-gradient rows inside a stage
#define RD_UW2_STAGE (RD_UW2_DY + 96 * 256)
#define RD_UW2_ZERO (2 * RD_UW2_STAGE)
#define RD_UW2_LDS (RD_UW2_ZERO + 128)
#define RD_UW2_TILE (256 * 128)                      // floats per (phase, tap) product

// grid: 32 G workgroups of 512 threads, blockIdx = g_lo + 8 (type + 32 g_hi), type = phase * 4 + quarter, group = g_lo + 8 g_hi;
// group g walks samples g, g + G, ... < B.  bias_partial (optional): [G][8 phases][128], written by the quarter-0 workgroups.
__global__ void __launch_bounds__(512, 1)
k_upconv2_wgrad_slab16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ dy, float* __restrict__ partial, int B, int G,
                       float* __restrict__ bias_partial = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the wave's tap (td, th, tw)
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int g_lo = blockIdx.x & 7, type = (blockIdx.x >> 3) & 31, group = g_lo + 8 * (blockIdx.x >> 8);
  const int phase = type >> 2, quarter = type & 3;
  const int pd = phase >> 2, ph = (phase >> 1) & 1, pw = phase & 1;
  const int td = wave >> 2, th = (wave >> 1) & 1, tw = wave & 1;
  const int oh = ph - 1 + th, ow = pw - 1 + tw;
  if (tid < 32) *(float*)(lds + RD_UW2_ZERO + tid * 4) = 0.f;

  // transposed-read addresses: a 16-position k-step is source plane d = kk: position 8 lhalf + q4 (+4): h = 2 lhalf + rd, w = q4
  const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  int a_row[2];                    // byte offset of the tap's source row inside its plane, or -1
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int hh = 2 * lhalf + rd + oh, ww = q4 + ow;
    a_row[rd] = ((unsigned)hh < 4u && (unsigned)ww < 4u) ? (hh * 4 + ww) * 128 : -1;
  }
  const int a_swz = (((q4 + ow) >> 1) & 1) << 2;     // rd_tr_swz<128> of the source row: planes and h rows are multiples of 4 rows
  const int a_colb = ((2 * g16 + (p4 >> 1)) << 4) + (p4 & 1) * 8;
  int a_col[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_col[i] = (i * 64 + a_colb) ^ (a_swz << 4);
  int b_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    b_off[j] = RD_UW2_DY + (8 * lhalf + q4) * 256 + (((j * 4 + 2 * g16 + (p4 >> 1)) ^ rd_tr_swz<256>(q4)) << 4) + (p4 & 1) * 8;

  // DMA sources, once: instruction i: i < 14: rows 8 (i & 1) .. of source plane pd - 1 + (i >> 1) (this quarter's 128 B of a row);
  // 14 <= i < 38: output-gradient rows 4 (i - 14) .. + 3 of the phase
  int dma_off[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = wave + 8 * k;
    int off = (int)RD_OOB;
    if (i < 14) {
      const int slot = i >> 1, r = (i & 1) * 8 + (lane >> 3), d = pd - 1 + slot;
      if ((unsigned)d < 6u) off = (d * 16 + r) * 512 + quarter * 128 + (((lane & 7) ^ rd_tr_swz<128>(r)) << 4);
    } else if (i < 38) {
      const int kpos = (i - 14) * 4 + (lane >> 4);
      const int d = kpos >> 4, hh = (kpos >> 2) & 3, ww = kpos & 3;
      const int orow = ((2 * d + pd) * 8 + 2 * hh + ph) * 8 + 2 * ww + pw;
      off = orow * 256 + (((lane & 15) ^ rd_tr_swz<256>(kpos)) << 4);
    }
    dma_off[k] = off;
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = bias_partial != nullptr && wave == 0 && quarter == 0;

  auto load_item = [&](int b, int stage) {
    const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc((const float*)(x + (long)b * (96 * 256)));
    const __amdgpu_buffer_rsrc_t rsY = rd_make_rsrc((const float*)(dy + (long)b * (768 * 128)));
    char* st = lds + stage * RD_UW2_STAGE;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = wave + 8 * k;                    // wave-uniform
      if (i < 14) rd_lds_dma16(rsX, (float*)(st + i * 1024), dma_off[k], 0);
      else if (i < 38) rd_lds_dma16(rsY, (float*)(st + RD_UW2_DY + (i - 14) * 1024), dma_off[k], 0);
    }
  };

  int b = group, stage = 0;
  if (b < B) load_item(b, 0);
  rd_dma_landed();
  __syncthreads();
  for (; b < B; b += G, stage ^= 1) {
    if (b + G < B) load_item(b + G, stage ^ 1);
    const char* st = lds + stage * RD_UW2_STAGE;
    const int zoff = RD_UW2_ZERO - stage * RD_UW2_STAGE + a_colb;
    rd_bf16x8 fa[2][2], fb[2][4];
    auto load_frag = [&](int slot, int kk) {
      const int poff = (kk + td) * RD_UW2_XPLANE;    // plane slot of the tap: d + td (slot 0 = plane pd - 1)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o0 = a_row[0] >= 0 ? poff + a_row[0] + a_col[i] : zoff;
        const int o1 = a_row[1] >= 0 ? poff + a_row[1] + a_col[i] : zoff;
        fa[slot][i] = rd_tr_frag(st, o0, o1);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[slot][j] = rd_tr_frag(st, b_off[j] + kk * 16 * 256, b_off[j] + (kk * 16 + 4) * 256);
    };
    load_frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      const int cur = kk & 1;
      if (kk + 1 < 6) load_frag(cur ^ 1, kk + 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4_t w = __builtin_bit_cast(u32x4_t, fb[cur][j]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            bsum[j] += __builtin_bit_cast(float, w[e] << 16) + __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
        }
      }
    }
    rd_dma_landed();
    __syncthreads();
  }
  float* o = partial + (((long)group * 64 + phase * 8 + wave) * RD_UW2_TILE) + (long)quarter * 64 * 128;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 128 + j * 32 + l31] = acc[i][j][r];
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = bsum[j] + __shfl_xor(bsum[j], 32, 64);
      if (lhalf == 0) bias_partial[((long)group * 8 + phase) * 128 + j * 32 + l31] = v;
    }
  }
}
// dWc[i] = sum over groups of partial[g][i], i < 64 * RD_UW2_TILE (fixed order)
__global__ void __launch_bounds__(256)
k_upconv2_wgrad_fold(const float* __restrict__ partial, int G, float* __restrict__ dWc) {
  const long i4 = blockIdx.x * 256L + threadIdx.x;
  if (i4 >= 64L * RD_UW2_TILE / 4) return;
  f32x4 s = *(const f32x4*)(partial + i4 * 4);
  for (int g = 1; g < G; ++g) s += *(const f32x4*)(partial + (long)g * 64 * RD_UW2_TILE + i4 * 4);
  *(f32x4*)(dWc + i4 * 4) = s;
}
