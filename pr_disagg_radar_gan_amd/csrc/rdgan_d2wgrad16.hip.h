// bf16 storage mode, ndomain 16: weight gradient of the critic's second layer (backward of T:291, Conv3D(128, 3x3x3, stride 2, 'same')
// on the 11 x 7 x 7 x 64 output of layer 1): dW[tap][64 ci][128 co] = sum over samples and output positions o of
// h1[2 o + tap - 1][ci] * dy[o][co], as a SLAB kernel in the pattern of k_upconv_wgrad_slab16 (round 3).
//
// Why.  As tiles of the streaming kernel (k_wgrad_gemm_ws16<128,128>, 14 row tiles x position splits) every 64-position chunk pulls
// 32 KB into LDS for 16 MFMAs per wave, and each of the 14 row tiles streams the whole output gradient again: 0.48 ms at 6144
// samples, 0.22 of the bf16 roof, five launches per iteration of BASELINE configs[2].  Here a wave OWNS ONE TAP: its [64 x 128] fp32
// product (8 MFMA tiles, 128 accumulator registers) stays in registers over the workgroup's whole share of the batch.  With
// stride 2 a tap reads only ONE parity class of layer 1's positions per axis (tap 1: even positions 2 o, taps 0 and 2: odd positions
// 2 o -+ 1), so the 27 taps fall into 8 classes of 8, 4, 4, 2, 4, 2, 2, 1 taps, each with its own dense sub-grid (5|6 x 3|4 x 3|4
// positions) on which its taps are shifts by 0 / -1 -- the collapsed-upsample picture again.  Four workgroup TYPES of eight waves
// carry the classes {8}, {4, 4}, {4, 2, 2}, {2, 1}; a work item is one sample: the type's sub-grids (6-26 KB of the sample's 69 KB)
// and the sample's 96 x 128 output gradient (24 KB) arrive by LDS-DMA into one of two stages while the other is multiplied (6 k-steps
// of 16 positions = one output hour plane each); both MFMA operands are read transposed from the position-major images
// (ds_read_b64_tr_b16).  The four types of a group walk the same samples at the same time on the same XCD (output gradient from L2 for
// three of them).  partial[group][tap][64][128], folded in a fixed order (k_d2_wgrad_fold).
// (Measured and not kept: three stages with the sample two items ahead in flight and a counted `s_waitcnt vmcnt(n)` at the end of an
// item -- 0.314 against 0.307 ms at 6144 samples: the waves do not wait for the DMA's latency.  SQ counters of the two-stage
// kernel, profiles/r03_sq_counters_bf16_bs256.csv: matrix pipe busy 0.27 of the SIMD cycles -- 0.84 of the waves have a tap -- 0.46
// of the wave cycles waiting at the per-item barrier or for an issue slot.)
#pragma once
#include "rdgan_gemm_ws16.hip.h"

#define RD_D2W_DY 24576                               // 96 output-gradient rows of 256 B
#define RD_D2W_XROWS 216                              // rows of 128 B reserved for a type's sub-grids (the largest type holds 206)
#define RD_D2W_STAGE (RD_D2W_DY + RD_D2W_XROWS * 128)
#define RD_D2W_ZERO (2 * RD_D2W_STAGE)                // a 128-byte row of zeros: taps outside the picture
#define RD_D2W_LDS (RD_D2W_ZERO + 128)
#define RD_D2W_TILE (64 * 128)                        // floats per tap

// TILED variant (k_d2_wgrad_slab_t16, round 4: ndomain 32 / 48 / 64): a work item is one (h, w) tile of 4 x 4 OUTPUT positions of a
// sample (all 6 output hours: the same 96 output-gradient rows, a k-step is still one output hour plane) and the layer-1 positions
// its taps reach: an even class holds positions 2 (o0 + j), j = 0..3, an odd class 2 (o0 + j) - 1, j = 0..4 (one halo position: tap 0
// of output o reads j = o - o0, tap 2 reads j + 1), i.e. sub-grids of 5|6 x 4|5 x 4|5 positions, at most 350 rows per type; positions
// outside the picture arrive as zeros (DMA offset out of range), so no tap is masked along h and w.
struct RdD2wGeom { int IH, IW, OH, OW, TH, TW; };     // layer-1 grid 11 x IH x IW, layer-2 grid 6 x OH x OW, TH x TW tiles
#define RD_D2WT_XROWS 352
#define RD_D2WT_STAGE (RD_D2W_DY + RD_D2WT_XROWS * 128)
#define RD_D2WT_ZERO (2 * RD_D2WT_STAGE)
#define RD_D2WT_LDS (RD_D2WT_ZERO + 128)

// class c = (cd, ch, cw) parity bits (1 = odd positions: taps 0 and 2 on that axis, 0 = even positions: tap 1); the classes of a type
// and their first row in the type's image.  Types: {7}, {6, 5}, {3, 4, 2}, {1, 0}.
template <bool TILED = false>
__device__ __forceinline__ void rd_d2w_class_dims(int c, int& nD, int& nH, int& nW) {
  nD = (c & 4) ? 5 : 6;
  nH = TILED ? ((c & 2) ? 5 : 4) : ((c & 2) ? 3 : 4);
  nW = TILED ? ((c & 1) ? 5 : 4) : ((c & 1) ? 3 : 4);
}
__constant__ int rd_d2w_type_classes[4][3] = {{7, -1, -1}, {6, 5, -1}, {3, 4, 2}, {1, 0, -1}};

// x [B][11][7][7][64] bf16 (layer 1's output; the penalty third holds the second sweep's r1), dy [B][6][4][4][128] bf16
// -> partial [G][27][64][128] fp32.  grid: 4 G workgroups of 512 threads, blockIdx = g_lo + 8 (type + 4 g_hi), group = g_lo + 8 g_hi
// (G a multiple of 8); group g walks samples g, g + G, ... < B.  Dynamic LDS RD_D2W_LDS.
template <bool TILED>
__device__ __forceinline__ void rd_d2w_body(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ dy, float* __restrict__ partial,
                                            int B, int G, const RdD2wGeom geo) {
  constexpr int STAGE = TILED ? RD_D2WT_STAGE : RD_D2W_STAGE, ZERO = TILED ? RD_D2WT_ZERO : RD_D2W_ZERO;
  constexpr int NSLOT = TILED ? 9 : 7;                 // DMA instructions per wave and item (24 + ceil(xrows / 8) over 8 waves)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int g_lo = blockIdx.x & 7, type = (blockIdx.x >> 3) & 3, group = g_lo + 8 * (blockIdx.x >> 5);
  if (tid < 32) *(float*)(lds + ZERO + tid * 4) = 0.f;

  // ---- this wave's tap: the wave-th tap of the type's classes in order (within a class: d, h, w with w fastest over the odd axes)
  int cls = -1, cbase = 0, tsel = wave, xrows = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int c = rd_d2w_type_classes[type][k];
    if (c < 0) continue;
    int nD, nH, nW; rd_d2w_class_dims<TILED>(c, nD, nH, nW);
    const int nt = 1 << __builtin_popcount(c);
    if (cls < 0 && tsel < nt) { cls = c; cbase = xrows; }
    if (cls < 0) tsel -= nt;
    xrows += nD * nH * nW;
  }
  const bool has_tap = cls >= 0;                      // (type 3 has three taps: five waves only load and wait)
  int nD = 6, nH = 4, nW = 4, sd = 0, sh = 0, sw = 0, tap = 0;
  if (has_tap) {
    rd_d2w_class_dims<TILED>(cls, nD, nH, nW);
    int rem = tsel, t3[3];
    // bits of tsel go to the odd axes, w first: bit 0 -> tap 0 (shift -1), bit 1 -> tap 2 (shift 0); an even axis has tap 1 (shift 0)
    for (int a = 2; a >= 0; --a) {
      const int odd = (cls >> (2 - a)) & 1;
      int t = 1;
      if (odd) { t = 2 * (rem & 1); rem >>= 1; }
      t3[a] = t;
    }
    tap = (t3[0] * 3 + t3[1]) * 3 + t3[2];
    sd = t3[0] == 0 ? -1 : 0; sh = t3[1] == 0 ? -1 : 0; sw = t3[2] == 0 ? -1 : 0;
    if (TILED) { sh += (cls >> 1) & 1; sw += cls & 1; }       // (an odd class starts at its halo position)
  }

  // ---- transposed-read addresses: this lane is lane 4 q4 + p4 of 16-lane group g16 in half lhalf; a 16-position k-step is one
  // output hour plane od = kk: position 8 lhalf + q4 (+ 4 for the second read): oh = 2 lhalf + rd, ow = q4
  const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  int a_hw[2];                     // row offset (jh * nW + jw) of the tap's source position inside its plane, or -1
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int jh = 2 * lhalf + rd + sh, jw = q4 + sw;
    a_hw[rd] = ((unsigned)jh < (unsigned)nH && (unsigned)jw < (unsigned)nW) ? jh * nW + jw : -1;
  }
  const int a_colb = ((2 * g16 + (p4 >> 1)) << 4) + (p4 & 1) * 8;     // this lane's 8 bytes inside a 32-channel block, unswizzled
  int b_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    b_off[j] = (8 * lhalf + q4) * 256 + (((j * 4 + 2 * g16 + (p4 >> 1)) ^ rd_tr_swz<256>(q4)) << 4) + (p4 & 1) * 8;

  // ---- DMA sources, once: instruction i of the item: i < 24: output-gradient rows 4 i .. 4 i + 3 (256 B each);
  // i >= 24: rows 8 (i - 24) .. + 7 of the type's sub-grid image (128 B each, gathered from layer 1's output).
  // TILED: offsets relative to the tile's first position; x rows keep their (h, w) relative to it for the per-item range check
  const int ndma = 24 + (xrows + 7) / 8;
  const int OHs = TILED ? geo.OH : 4, OWs = TILED ? geo.OW : 4, IHs = TILED ? geo.IH : 7, IWs = TILED ? geo.IW : 7;
  int dma_off[NSLOT], dma_hw[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int i = wave + 8 * k;
    int off = (int)RD_OOB, hw = -1;
    if (i < 24) {
      const int r = i * 4 + (lane >> 4);
      const int srow = TILED ? ((r >> 4) * OHs + ((r >> 2) & 3)) * OWs + (r & 3) : r;
      off = srow * 256 + (((lane & 15) ^ rd_tr_swz<256>(r)) << 4);
    } else if (i < ndma) {
      const int R = (i - 24) * 8 + (lane >> 3);
      if (R < xrows) {
        int c = -1, r = R;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int cc = rd_d2w_type_classes[type][q];
          if (cc < 0 || c >= 0) continue;
          int d, h, w; rd_d2w_class_dims<TILED>(cc, d, h, w);
          if (r < d * h * w) c = cc; else r -= d * h * w;
        }
        int d, h, w; rd_d2w_class_dims<TILED>(c, d, h, w);
        const int jd = r / (h * w), q = r - jd * h * w, jh = q / w, jw = q - jh * w;
        // nd16: odd class position j is 2 j + 1; tiled: 2 (o0 + j) - 1 relative to 2 o0: 2 j - 1
        const int hrel = 2 * jh + (((c >> 1) & 1) ? (TILED ? -1 : 1) : 0), wrel = 2 * jw + ((c & 1) ? (TILED ? -1 : 1) : 0);
        const int srow = ((2 * jd + ((c >> 2) & 1)) * IHs + hrel) * IWs + wrel;
        off = srow * 128 + (((lane & 7) ^ rd_tr_swz<128>(R)) << 4);
        hw = (hrel + 1) | ((wrel + 1) << 8);
      }
    }
    dma_off[k] = off; dma_hw[k] = hw;
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int ntile = TILED ? geo.TH * geo.TW : 1;
  auto load_item = [&](int item, int stage) {
    int b = item, oh0 = 0, ow0 = 0;
    if (TILED) { b = item / ntile; const int t = item - b * ntile; oh0 = (t / geo.TW) * 4; ow0 = (t - (t / geo.TW) * geo.TW) * 4; }
    const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc((const float*)(x + (long)b * (11 * IHs * IWs * 64)));
    const __amdgpu_buffer_rsrc_t rsY = rd_make_rsrc((const float*)(dy + (long)b * (6 * OHs * OWs * 128)));
    const int ysh = TILED ? (oh0 * OWs + ow0) * 256 : 0, xsh = TILED ? (2 * oh0 * IWs + 2 * ow0) * 128 : 0;
    char* st = lds + stage * STAGE;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int i = wave + 8 * k;                    // wave-uniform
      if (i < 24) rd_lds_dma16(rsY, (float*)(st + i * 1024), dma_off[k] + ysh, 0);
      else if (i < ndma) {
        unsigned voff = (unsigned)(dma_off[k] + xsh);
        if (TILED) {
          const int hh = 2 * oh0 + (dma_hw[k] & 255) - 1, ww = 2 * ow0 + (dma_hw[k] >> 8) - 1;
          if (dma_hw[k] < 0 || (unsigned)hh >= (unsigned)IHs || (unsigned)ww >= (unsigned)IWs) voff = RD_OOB;
          asm volatile("" : "+v"(voff));
        }
        rd_lds_dma16(rsX, (float*)(st + RD_D2W_DY + (i - 24) * 1024), (int)voff, 0);
      }
    }
  };

  const int nitems = B * ntile;
  int b = group, stage = 0;
  if (b < nitems) load_item(b, 0);
  rd_dma_landed();
  __syncthreads();
  for (; b < nitems; b += G, stage ^= 1) {
    if (b + G < nitems) load_item(b + G, stage ^ 1);
    if (has_tap) {
      const char* st = lds + stage * STAGE;
      const int zoff = ZERO - stage * STAGE + a_colb;
      rd_bf16x8 fa[2][2], fb[2][4];
      auto load_frag = [&](int slot, int kk) {
        const int jd = kk + sd;                        // wave-uniform
        const bool dok = (unsigned)jd < (unsigned)nD;
        const int rbase = cbase + jd * nH * nW;
        int o[2][2];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int R = rbase + a_hw[rd];
          const bool ok = dok && a_hw[rd] >= 0;
          const int swz = rd_tr_swz<128>(R);
#pragma unroll
          for (int i = 0; i < 2; ++i)
            o[rd][i] = ok ? RD_D2W_DY + R * 128 + ((i * 64 + a_colb) ^ (swz << 4)) : zoff;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[slot][i] = rd_tr_frag(st, o[0][i], o[1][i]);
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
      }
    }
    rd_dma_landed();
    __syncthreads();
  }
  if (has_tap) {
    float* o = partial + ((long)group * 27 + tap) * RD_D2W_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 128 + j * 32 + l31] = acc[i][j][r];
  }
}

__global__ void __launch_bounds__(512, 1)
k_d2_wgrad_slab16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ dy, float* __restrict__ partial, int B, int G) {
  rd_d2w_body<false>(x, dy, partial, B, G, RdD2wGeom());
}
// x [B][11][IH][IW][64] bf16, dy [B][6][OH][OW][128] bf16 (OH, OW multiples of 4) -> partial [G][27][64][128]; group g walks the items
// (sample, tile) g, g + G, ... < B TH TW.  Dynamic LDS RD_D2WT_LDS.
__global__ void __launch_bounds__(512, 1)
k_d2_wgrad_slab_t16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ dy, float* __restrict__ partial, int B, int G,
                    RdD2wGeom geo) {
  rd_d2w_body<true>(x, dy, partial, B, G, geo);
}

// dW[i] = sum over groups of partial[g][i], i < 27 * RD_D2W_TILE, in the order of the groups (deterministic)
__global__ void __launch_bounds__(256)
k_d2_wgrad_fold(const float* __restrict__ partial, int G, float* __restrict__ dW) {
  const long i4 = blockIdx.x * 256L + threadIdx.x;
  if (i4 >= 27L * RD_D2W_TILE / 4) return;
  f32x4 s = *(const f32x4*)(partial + i4 * 4);
  for (int g = 1; g < G; ++g) s += *(const f32x4*)(partial + (long)g * 27 * RD_D2W_TILE + i4 * 4);
  *(f32x4*)(dW + i4 * 4) = s;
}
