// bf16 storage mode, ndomain 16: weight gradient of generator block 3 in the collapsed form (backward of T:340-341:
// dWc[phase * 8 + tap][128 ci][64 co] = sum over samples and source positions r of x[r + off(phase, tap)][ci] * dy[out(r, phase)][co];
// k_fold_collapsed_wgrad turns the 64 entries into the 27-tap gradient) as a SLAB kernel (round 3).
//
// Why.  As tiles of the streaming kernel (k_wgrad_gemm_ws16<256, 64>) every 64-position chunk of a 256 x 64 tile pulls 40 KB into
// LDS for 16 MFMAs per wave -- 80 B/clk per CU where a CU takes in ~30 -- and the gathered rows are fetched again for every one of the
// 32 row tiles: 2.6 ms at 2048 samples, 0.25 of the bf16 roof, the largest launch of the generator's backward pass.  Here a
// workgroup OWNS ONE PHASE (pd, ph, pw) and keeps all eight tap products of it -- eight [128 x 64] fp32 tiles, one per wave, 128
// accumulator registers each -- in registers over its whole share of the batch, and walks work items = (sample, two source hour
// planes): the three source planes the phase's taps touch (48 KB) and the 128 output-gradient rows of the phase (16 KB) come in by
// LDS-DMA into one of two stages while the other stage is multiplied; every wave multiplies the SAME 128 positions (8 k-steps of
// 16) for its own tap: a shifted view of the resident planes, read transposed (ds_read_b64_tr_b16: 8 consecutive positions of one
// channel per lane, as in k_wgrad_gemm_ws16).  64 KB per 512 MFMAs instead of 40 KB per 64.  The eight phases of a group of
// workgroups walk the same items at the same time on the same XCD, so a sample's source planes reach seven of them from L2.
// One barrier per item; no epilogue until the end: partial[group][phase * 8 + tap][128][64], folded by k_reduce_partials in a fixed
// order (deterministic).
#pragma once
#include "rdgan_gemm_ws16.hip.h"

#define RD_UWG_XPLANE 16384                          // 64 rows of 256 B
#define RD_UWG_STAGE (3 * RD_UWG_XPLANE + 128 * 128) // three source planes + 128 output-gradient rows of 128 B
#define RD_UWG_ZERO (2 * RD_UWG_STAGE)               // a 256-byte row of zeros: taps outside the (h, w) picture
#define RD_UWG_LDS (RD_UWG_ZERO + 256)
#define RD_UWG_TILE (128 * 64)                       // floats per (phase, tap) product

// x [B][12][8][8][128] bf16 (block input h2), dy [B][24][16][16][64] bf16 (gradient at the block's conv output)
// -> partial [G][64][128][64] fp32.  grid: 8 G workgroups of 512 threads, blockIdx = g_lo + 8 (phase + 8 g_hi), group = g_lo + 8 g_hi
// (G a multiple of 8: the eight phases of a group share an XCD); group g walks items g, g + G, ... < 6 B.  Dynamic LDS RD_UWG_LDS.
__global__ void __launch_bounds__(512, 1)
k_upconv_wgrad_slab16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ dy, float* __restrict__ partial, int B, int G,
                      float* __restrict__ bias_partial = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the wave's tap (td, th, tw)
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int g_lo = blockIdx.x & 7, phase = (blockIdx.x >> 3) & 7, group = g_lo + 8 * (blockIdx.x >> 6);
  const int pd = phase >> 2, ph = (phase >> 1) & 1, pw = phase & 1;
  const int td = wave >> 2, th = (wave >> 1) & 1, tw = wave & 1;
  const int oh = ph - 1 + th, ow = pw - 1 + tw;                   // the tap's (h, w) offset; its plane slot is dc + td
  if (tid < 64) *(float*)(lds + RD_UWG_ZERO + tid * 4) = 0.f;

  // ---- transposed-read addresses (k_wgrad_gemm_ws16): this lane is lane 4 q4 + p4 of 16-lane group g16 in half lhalf and supplies
  // the address of position 8 lhalf + q4 (+ 4 for the second read) of a 16-position step, columns 4 p4 .. of its 16-column group
  const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  // A = x^T: position pp = 16 (kk & 3) + 8 lhalf + q4 (+4) of plane dc = kk >> 2: h = 2 (kk & 3) + lhalf, w = q4 (+4)
  int a_row[4][2];                 // byte offset of the tap's source row inside a stage for kk & 3 and the two reads, or -1
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int hh = 2 * k4 + lhalf + oh, ww = q4 + 4 * rd + ow;
      a_row[k4][rd] = ((unsigned)hh < 8u && (unsigned)ww < 8u) ? (td * 64 + hh * 8 + ww) * 256 : -1;
    }
  const int a_swz = ((q4 + ow) & 3) << 2;           // rd_tr_swz<256> of the source row: (w + ow) & 3, the same for both reads
  int a_col[4];                    // byte offset of this lane's 8 bytes inside the row, per block of 32 channels
#pragma unroll
  for (int i = 0; i < 4; ++i) a_col[i] = (((i * 4 + 2 * g16 + (p4 >> 1)) ^ a_swz) << 4) + (p4 & 1) * 8;
  const int z_col = ((2 * g16 + (p4 >> 1)) << 4) + (p4 & 1) * 8;     // inside the zero row
  // B = dy: position row 16 kk + 8 lhalf + q4 (+4) of the 128-row image, 128-byte rows
  int b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
    b_off[j] = 3 * RD_UWG_XPLANE + (8 * lhalf + q4) * 128 + ((((j * 32) / 8 + 2 * g16 + (p4 >> 1)) ^ rd_tr_swz<128>(q4)) << 4) + (p4 & 1) * 8;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // bias gradient = column sums of dy: the eight phases partition the block's output positions, so the tap-0 wave of every
  // workgroup also sums the output-gradient fragments it multiplies (lane: column 32 j + l31, the 8 positions of its k-group)
  // -> bias_partial[group][phase][64], folded by k_reduce_partials: the column-sum pass over dy (1.6 GB at 2048 samples,
  // 0.78 ms beside the GEMMs) is gone
  float bsum[2] = {0.f, 0.f};
  const bool do_bias = bias_partial != nullptr && wave == 0;
  const int nitems = 6 * B;
  auto load_item = [&](int item, int stage) {        // 64 DMA instructions of 1 KB, 8 per wave
    const int b = item / 6, d0 = (item - b * 6) * 2;
    const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc((const float*)(x + (long)b * (12 * 64 * 128)));
    const __amdgpu_buffer_rsrc_t rsY = rd_make_rsrc((const float*)(dy + (long)b * (24 * 256 * 64)));
    char* st = lds + stage * RD_UWG_STAGE;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = wave * 8 + k;                    // wave-uniform
      if (i < 48) {                                  // source planes d0 + pd - 1 + slot: 4 rows of 256 B per instruction
        const int slot = i >> 4, row = (i & 15) * 4 + (lane >> 4);
        const int d = d0 + pd - 1 + slot;
        const int cl = (lane & 15) ^ ((row & 3) << 2);
        unsigned voff = (unsigned)d < 12u ? (unsigned)((d * 64 + row) * 256 + cl * 16) : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rsX, (float*)(st + i * 1024), (int)voff, 0);
      } else {                                       // output-gradient rows of the phase: 8 rows of 128 B per instruction
        const int kpos = (i - 48) * 8 + (lane >> 3);
        const int dc = kpos >> 6, hh = (kpos >> 3) & 7, ww = kpos & 7;
        const int orow = ((2 * (d0 + dc) + pd) * 16 + 2 * hh + ph) * 16 + 2 * ww + pw;
        const int cl = (lane & 7) ^ rd_tr_swz<128>(kpos);
        rd_lds_dma16(rsY, (float*)(st + 3 * RD_UWG_XPLANE + (i - 48) * 1024), orow * 128 + cl * 16, 0);
      }
    }
  };

  int item = group, stage = 0;
  if (item < nitems) load_item(item, 0);
  rd_dma_landed();
  __syncthreads();
  for (; item < nitems; item += G, stage ^= 1) {
    if (item + G < nitems) load_item(item + G, stage ^ 1);
    const char* st = lds + stage * RD_UWG_STAGE;
    rd_bf16x8 fa[2][4], fb[2][2];
    auto load_frag = [&](int slot, int kk) {
      const int dcoff = (kk >> 2) * RD_UWG_XPLANE;
      const int r0 = a_row[kk & 3][0], r1 = a_row[kk & 3][1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o0 = r0 >= 0 ? r0 + dcoff + a_col[i] : RD_UWG_ZERO - stage * RD_UWG_STAGE + z_col;
        const int o1 = r1 >= 0 ? r1 + dcoff + a_col[i] : RD_UWG_ZERO - stage * RD_UWG_STAGE + z_col;
        fa[slot][i] = rd_tr_frag(st, o0, o1);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[slot][j] = rd_tr_frag(st, b_off[j] + kk * 16 * 128, b_off[j] + (kk * 16 + 4) * 128);
    };
    load_frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int cur = kk & 1;
      if (kk + 1 < 8) load_frag(cur ^ 1, kk + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u32x4_t w = __builtin_bit_cast(u32x4_t, fb[cur][j]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            bsum[j] += __builtin_bit_cast(float, w[e] << 16) + __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
        }
      }
    }
    rd_dma_landed();
    __syncthreads();                                 // the next item has landed; this stage may be overwritten by the one after
  }
  // ---- the wave's [128 ci][64 co] product: register r of block (i, j) = row 32 i + (r & 3) + 8 (r >> 2) + 4 lhalf, column 32 j + l31
  float* o = partial + (((long)group * 64 + phase * 8 + wave) * RD_UWG_TILE);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 64 + j * 32 + l31] = acc[i][j][r];
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float v = bsum[j] + __shfl_xor(bsum[j], 32, 64);        // the other k-group's eight positions
      if (lhalf == 0) bias_partial[((long)group * 8 + phase) * 64 + j * 32 + l31] = v;
    }
  }
}

// dWc[i] = sum over groups of partial[g][i], i < 64 * RD_UWG_TILE, in the order of the groups (deterministic)
__global__ void __launch_bounds__(256)
k_upconv_wgrad_fold(const float* __restrict__ partial, int G, float* __restrict__ dWc) {
  const long i4 = blockIdx.x * 256L + threadIdx.x;
  if (i4 >= 64L * RD_UWG_TILE / 4) return;
  f32x4 s = *(const f32x4*)(partial + i4 * 4);
  for (int g = 1; g < G; ++g) s += *(const f32x4*)(partial + (long)g * 64 * RD_UWG_TILE + i4 * 4);
  *(f32x4*)(dWc + i4 * 4) = s;
}
