// bf16 storage mode, ndomain 16: forward of the critic's second layer (T:291-293: Conv3D(128, 3x3x3, stride 2, 'same') on the
// 11 x 7 x 7 x 64 output of layer 1 + bias + LeakyReLU + dropout -> 6 x 4 x 4 x 128) as a SLAB kernel in the pattern of
// k_upconv2_slab16 (rdgan_upconv16b.hip.h): a work item is ONE SAMPLE, its layer-1 output (539 rows x 128 B = 69 KB) resident in LDS
// for all 27 taps -- a tap is a stride-2 view of those rows, positions in the padding read a zero row; the four waves of a workgroup
// split the 128 output channels (wave nb: channels 32 nb ..., all 96 output positions = 3 accumulator tiles) and stream their own
// weight fragments global -> VGPR (1 KB per k-step of 16 input channels, a queue of eight k-steps, taps in pairs: 27 taps + one
// zero tap = 14 x 8 k-steps); operands swapped (weights = MFMA A operand), so bias + LeakyReLU + dropout + bf16 rounding run in
// registers on 16 channels of one output row per lane.  Two 256-thread workgroups per CU.  As tiles of the streaming kernel
// (k_conv_gemm_ws<128,128,...,bf16>) the layer runs at 0.29 of the bf16 roof (0.35 ms at 6144 samples).
// MEASURED: 0.33-0.35 ms at 6144 samples -- NO faster than the streaming GEMM (option "d2_fwd_slab", default OFF; parity-tested
// like the others).  Unlike block 2 of the generator (8 phases x 8 taps x 256 channels = 3072 MFMAs per wave behind one 48 KB
// slab load) a sample here is 336 MFMAs per wave behind a 69 KB load and two barriers, and only two workgroups fit a CU to cover
// each other's loads; neither the bank-friendly row order below nor taking the per-tap position arithmetic out of the loop
// moved the time (0.345 -> 0.334).  What it would need is the next sample's slab in flight during this one's taps (138 KB: one
// workgroup per CU) -- not built.
//
// LDS image: the 32 lanes of a fragment read take source rows TWO apart along w (stride 2) -- all of one parity, and a 128-byte row
// covers half the banks, so in source order a 16-lane group could reach 8 of the 16 bank groups (first version: 0.35 ms at 6144
// samples, no faster than the streaming GEMM).  Rows are therefore stored even rows first, odd rows behind them: row r at
// p = (r >> 1) + 272 (r & 1), 16-byte chunk c of it at c ^ ((p >> 1) & 7); rows two apart are neighbours in LDS.
#pragma once
#include "rdgan_upconv16b.hip.h"

#define RD_D2F_ROWS 539
#define RD_D2F_IMG (544 * 128)                       // 68 DMA instructions of 8 rows
#define RD_D2F_ZERO RD_D2F_IMG                       // a 128-byte row of zeros
#define RD_D2F_BIAS (RD_D2F_ZERO + 128)              // 128 floats
#define RD_D2F_LDS (RD_D2F_BIAS + 512)
#define RD_D2F_KSTEPS 112                            // 28 taps (the last one zero) x 4 steps of 16 input channels

// Weight image from the layer's kernel w2 [27][64 ci][128 co] (fp32): for k-step g = tap * 4 + j and channel block nb, lane l holds
// the 8 bf16 w2[tap][16 j + 8 (l >> 5) + e][32 nb + (l & 31)] (tap 27: zeros): 1 KB per (g, nb), 448 KB.
__global__ void k_d2f_wimg(const float* __restrict__ w2, unsigned short* __restrict__ wimg) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                 // (g, nb, lane)
  if (idx >= RD_D2F_KSTEPS * 4 * 64) return;
  const int lane = idx & 63, nb = (idx >> 6) & 3, g = idx >> 8;
  const int tap = g >> 2, j = g & 3;
  const int n = nb * 32 + (lane & 31), k0 = j * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = tap < 27 ? w2[((long)tap * 64 + k0 + e) * 128 + n] : 0.f;
  u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
  *(u32x4_t*)(wimg + (long)idx * 8) = o;
}

// x [B][11][7][7][64] bf16 -> out [B][6][4][4][128] bf16 = dropout(LeakyReLU(conv(x) + bias)); dropout counter = flat index of out
// + idx_base (rd_drop_word / rd_drop_apply_w: one hash word per channel quad, +0.0 = dropped).
// grid: min(B, 2 per CU) persistent workgroups of 256 threads; dynamic LDS RD_D2F_LDS.
__global__ void __launch_bounds__(256, 2)
k_d2_fwd_slab16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ wimg, const float* __restrict__ bias,
                rd_bf16_t* __restrict__ out, int B, int use_drop, uint32_t key, uint32_t idx_base) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nb = __builtin_amdgcn_readfirstlane(tid >> 6);         // the wave's channel block
  const int l31 = lane & 31, lhalf = lane >> 5;
  if (tid < 128) *(float*)(lds + RD_D2F_BIAS + tid * 4) = bias[tid];
  if (tid < 32) *(float*)(lds + RD_D2F_ZERO + tid * 4) = 0.f;
  const unsigned wvoff = (unsigned)lane * 16u;
  // this lane's three output positions o = 32 mb + l31 of the 6 x 4 x 4 grid: source position of tap (0,0,0) = 2 o - 1 per axis
  // (the same for every sample: the source row of tap (0,0,0) -- may be negative -- and a 28-bit mask of the taps that land inside
  // the picture are computed once; per tap the row is base + (td * 49 + th * 7 + tw).  A first version recomputed positions and
  // range checks per tap: ~120 VALU instructions per tap pair beside its 24 MFMAs)
  int rbase[3]; unsigned tmask[3];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb) {
    const int o = 32 * mb + l31;
    const int id0 = 2 * (o >> 4) - 1, ih0 = 2 * ((o >> 2) & 3) - 1, iw0 = 2 * (o & 3) - 1;
    rbase[mb] = (id0 * 7 + ih0) * 7 + iw0;
    unsigned m = 0;
    for (int t = 0; t < 27; ++t) {
      const int dd = id0 + t / 9, hh = ih0 + (t / 3) % 3, ww = iw0 + t % 3;
      if ((unsigned)dd < 11u && (unsigned)hh < 7u && (unsigned)ww < 7u) m |= 1u << t;
    }
    tmask[mb] = m;
  }
  const char* wbase = (const char*)wimg + (long)nb * 1024;         // k-step g of this wave at wbase + g * 4 KB

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();                                  // every wave has left the previous sample (and bias / zero rows are in)
    {
      const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(x + (long)b * (RD_D2F_ROWS * 64)));
#pragma unroll
      for (int k = 0; k < 17; ++k) {
        const int i = nb * 17 + k;                    // wave-uniform: 68 instructions of 8 rows of the LDS image
        const int p = i * 8 + (lane >> 3);            // row of the image = source row 2 p (p < 272) or 2 (p - 272) + 1
        const int row = p < 272 ? 2 * p : 2 * (p - 272) + 1;
        const int cl = (lane & 7) ^ ((p >> 1) & 7);
        unsigned voff = row < RD_D2F_ROWS ? (unsigned)(row * 128 + cl * 16) : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rs, (float*)(lds + i * 1024), (int)voff, 0);
      }
    }
    rd_dma_landed();
    __syncthreads();

    f32x16 acc[3];
    {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *(const f32x4*)(lds + RD_D2F_BIAS + (nb * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
          acc[mb][4 * g + 0] = b4.x; acc[mb][4 * g + 1] = b4.y; acc[mb][4 * g + 2] = b4.z; acc[mb][4 * g + 3] = b4.w;
        }
      }
    }
    u32x4_t bq[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) rd_up2_wload(bq[s], wbase + (long)s * 4096, wvoff);
    // (no `continue` and no branch around a load or its wait in this loop -- rdgan_d2slab16.hip.h)
#pragma unroll 1
    for (int tp = 0; tp < 14; ++tp) {
      int abase[2][3], aswz[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * tp + u;                     // (tap 27 does not exist: its weights are zeros, its rows the zero row)
        const int toff = (t / 9) * 49 + ((t / 3) % 3) * 7 + t % 3;       // wave-uniform
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
          const bool ok = (tmask[mb] >> t) & 1u;
          const int rs = rbase[mb] + toff;
          const int p = (rs >> 1) + 272 * (rs & 1);
          abase[u][mb] = ok ? p * 128 : RD_D2F_ZERO;
          aswz[u][mb] = ok ? ((p >> 1) & 7) : 0;
        }
      }
      u32x4_t afr[2][3];
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) afr[0][mb] = *(const u32x4_t*)(lds + abase[0][mb] + ((lhalf ^ aswz[0][mb]) << 4));
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) {
          const int u = (s + 1) >> 2, c = 2 * ((s + 1) & 3) + lhalf;          // tap of the pair, logical chunk of the next k-step
#pragma unroll
          for (int mb = 0; mb < 3; ++mb) afr[(s + 1) & 1][mb] = *(const u32x4_t*)(lds + abase[u][mb] + ((c ^ aswz[u][mb]) << 4));
        }
        rd_up2_wait<7>(bq[s]);                        // the oldest of the eight loads in flight
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[s]),
                                                            __builtin_bit_cast(rd_bf16x8, afr[s & 1][mb]), acc[mb], 0, 0, 0);
        {
          const int gn = tp * 8 + s + 8;              // refill with k-step + 8 (past the end: the last one again, never used)
          rd_up2_wload(bq[s], wbase + (long)(gn < RD_D2F_KSTEPS ? gn : RD_D2F_KSTEPS - 1) * 4096, wvoff);
        }
      }
    }
    // the clamped refills are still in flight: wait for them HERE, naming their registers, before anything else is allocated
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]), "+v"(bq[7]));
    // ---- epilogue in registers: register r = channel 32 nb + 8 (r >> 2) + 4 lhalf + (r & 3) of output row 32 mb + l31
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
      const long row = (long)b * 96 + 32 * mb + l31;
      const uint32_t ibase = (uint32_t)(row * 128) + idx_base + nb * 32 + 4 * lhalf;
      char* orow = (char*)out + row * 256 + nb * 64 + lhalf * 16;
#pragma unroll
      for (int G = 0; G < 4; G += 2) {
        unsigned lo[2], hi[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint32_t w = rd_drop_word(key, ibase + 8 * (G + u));
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y = rd_lrelu(acc[mb][4 * (G + u) + e]);
            if (use_drop) y = rd_drop_apply_w(y, w, e);
            v[e] = y;
          }
          lo[u] = rd_pack_bf16(v[0], v[1]); hi[u] = rd_pack_bf16(v[2], v[3]);
        }
        const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], lo[1], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(hi[0], hi[1], false, false);
        const u32x4_t o = {sx[0], sy[0], sx[1], sy[1]};
        *(u32x4_t*)(orow + G * 16) = o;
      }
    }
  }
}
