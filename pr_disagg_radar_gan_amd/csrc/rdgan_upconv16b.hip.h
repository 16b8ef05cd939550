// bf16 storage mode, ndomain 16: generator block 2 forward (T:335-338: UpSampling3D(2) + Conv3D(256 -> 128, 3x3x3, 'same') +
// PixelNorm + LeakyReLU onto the 12 x 8 x 8 grid) in the collapsed form, as a SLAB kernel: the structure of k_upconv_slab16
// (rdgan_upconv16.hip.h) on a layer with N = 128 output channels.
//
// Block 3's wave tile (128 rows x 64 channels = all of a row's channels) does not fit: 128 channels of 96 rows are 12 accumulator
// tiles.  Here the four waves of a workgroup SPLIT THE CHANNELS: wave nb owns channels 32 nb .. 32 nb + 31 of all 96 rows of a
// phase (3 accumulator tiles), streams only its own weight fragments (1 KB per k-step, a queue of eight k-steps), and the
// PixelNorm row sum of squares crosses the waves once per phase through LDS (96 x 4 floats, one barrier; the four waves run the
// same K loop, so they arrive together).  A work item is ONE SAMPLE: its whole block input (6 x 4 x 4 positions x 256 channels = 48 KB)
// resident for all 8 phases x 8 taps; every tap is a shifted read of those rows, rows outside the picture read a zero row.
// Two 256-thread workgroups per CU.  Operands swapped as in block 3 (weights = MFMA A operand): a lane holds 16 channels of one
// output row, bias + normalisation + LeakyReLU + bf16 rounding in registers, 16-byte stores.
//
// LDS image: 96 rows of 512 B, 16-byte chunk c of row r at (c & 16) | ((c ^ r) & 15) (swizzle on the DMA's source side).
#pragma once
#include "rdgan_upconv16.hip.h"

#define RD_UP2_IMG (96 * 512)
#define RD_UP2_ZERO RD_UP2_IMG                       // a 512-byte row of zeros
#define RD_UP2_SS (RD_UP2_ZERO + 512)                // [2][96][4] floats: per-wave partial sums of squares of a phase's rows
#define RD_UP2_BIAS (RD_UP2_SS + 2 * 96 * 4 * 4)     // 128 floats
#define RD_UP2_LDS (RD_UP2_BIAS + 512)
#ifndef RD_UP2_WGS
#define RD_UP2_WGS 3                                 // workgroups per CU (50 KB of LDS and <= 168 VGPRs each)
#endif
#define RD_UP2_KSTEPS 1024                           // 64 (phase, tap) x 16 steps of 16 input channels

// Weight image from the collapsed forms Wc [64 = phase*8 + tap][256 ci][128 co] (fp32): for k-step g = (phase*8 + tap)*16 + j and
// channel block nb, lane l holds the 8 bf16 Wc[phase*8 + tap][16 j + 8 (l >> 5) + e][32 nb + (l & 31)]: 1 KB per (g, nb), 4 MB.
__global__ void k_upconv2_wimg(const float* __restrict__ Wc, unsigned short* __restrict__ wimg) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                 // (g, nb, lane)
  if (idx >= RD_UP2_KSTEPS * 4 * 64) return;
  const int lane = idx & 63, nb = (idx >> 6) & 3, g = idx >> 8;
  const int pt = g >> 4, j = g & 15;
  const int n = nb * 32 + (lane & 31), k0 = j * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = Wc[((long)pt * 256 + k0 + e) * 128 + n];
  u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
  *(u32x4_t*)(wimg + (long)idx * 8) = o;
}

// one 1 KB weight fragment global -> VGPR (see rd_upc_wload: inline asm so that hipcc neither re-schedules nor drains the queue;
// s_nop 4: the SGPR base has just been computed by SALU instructions)
__device__ __forceinline__ void rd_up2_wload(u32x4_t& d, const char* base, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void rd_up2_wait(u32x4_t& d) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "i"(N)); }

// x [B][6][4][4][256] bf16 -> out [B][12][8][8][128] bf16 = LeakyReLU(PixelNorm(upconv(x) + bias)), rinv [B][12][8][8] = 1/l2.
// grid: min(B, 2 per CU) persistent workgroups of 256 threads; dynamic LDS RD_UP2_LDS.
__global__ void __launch_bounds__(256, RD_UP2_WGS)
k_upconv2_slab16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ wimg, const float* __restrict__ bias,
                 rd_bf16_t* __restrict__ out, float* __restrict__ rinv, int B) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nb = __builtin_amdgcn_readfirstlane(tid >> 6);         // the wave's channel block
  const int l31 = lane & 31, lhalf = lane >> 5;
  if (tid < 128) {
    *(float*)(lds + RD_UP2_BIAS + tid * 4) = bias[tid];
    *(float*)(lds + RD_UP2_ZERO + tid * 4) = 0.f;
  }
  const unsigned wvoff = (unsigned)lane * 16u;
  // this lane's three rows (source positions r = 32 mb + l31 of the 6 x 4 x 4 grid)
  int rd_[3], rh_[3], rw_[3];
#pragma unroll
  for (int mb = 0; mb < 3; ++mb) { const int r = 32 * mb + l31; rd_[mb] = r >> 4; rh_[mb] = (r >> 2) & 3; rw_[mb] = r & 3; }

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();                                  // every wave has left the previous sample (and bias / zero rows are in)
    {
      // 48 KB, contiguous: 48 DMA instructions of 1 KB (2 rows), 12 per wave
      const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(x + (long)b * (96 * 256)));
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int i = nb * 12 + k;                    // wave-uniform
        const int row = i * 2 + (lane >> 5);
        const int cp = lane & 31;                     // physical chunk
        const int cl = (cp & 16) | ((cp ^ row) & 15); // logical chunk stored there
        rd_lds_dma16(rs, (float*)(lds + i * 1024), row * 512 + cl * 16, 0);
      }
    }
    rd_dma_landed();
    __syncthreads();

#pragma unroll 1
    for (int phase = 0; phase < 8; ++phase) {
      const int pd = phase >> 2, ph = (phase >> 1) & 1, pw = phase & 1;
      f32x16 acc[3];
      {
        // accumulators start at the bias of their channel: register r = channel 32 nb + 8 (r >> 2) + 4 lhalf + (r & 3)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *(const f32x4*)(lds + RD_UP2_BIAS + (nb * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
          for (int mb = 0; mb < 3; ++mb) {
            acc[mb][4 * g + 0] = b4.x; acc[mb][4 * g + 1] = b4.y; acc[mb][4 * g + 2] = b4.z; acc[mb][4 * g + 3] = b4.w;
          }
        }
      }
      // weight fragments of this wave: k-step g of the phase at ((phase * 128 + g) * 4 + nb) KB; a queue of eight k-steps
      const char* wph = (const char*)wimg + ((long)phase * 128 * 4 + nb) * 1024;          // wave-uniform
      u32x4_t bq[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) rd_up2_wload(bq[s], wph + (long)s * 4096, wvoff);
      // (no `continue` and no branch around a load or its wait in this loop: the queue is carried around it with loads in
      // flight -- rdgan_d2slab16.hip.h)
#pragma unroll 1
      for (int t = 0; t < 8; ++t) {
        const int od = pd - 1 + (t >> 2), oh = ph - 1 + ((t >> 1) & 1), ow = pw - 1 + (t & 1);
        int abase[3], aswz[3];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
          const int dd = rd_[mb] + od, hh = rh_[mb] + oh, ww = rw_[mb] + ow;
          const bool ok = (unsigned)dd < 6u && (unsigned)hh < 4u && (unsigned)ww < 4u;
          const int rs = (dd * 4 + hh) * 4 + ww;
          abase[mb] = ok ? rs * 512 : RD_UP2_ZERO;
          aswz[mb] = ok ? (rs & 15) : 0;
        }
        u32x4_t afr[2][3];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) afr[0][mb] = *(const u32x4_t*)(lds + abase[mb] + (((lhalf ^ aswz[mb]) & 15) << 4));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (j + 1 < 16) {
            const int c = 2 * (j + 1) + lhalf;        // logical chunk of the next k-step
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
              afr[(j + 1) & 1][mb] = *(const u32x4_t*)(lds + abase[mb] + (((c & 16) | ((c ^ aswz[mb]) & 15)) << 4));
          }
          rd_up2_wait<7>(bq[j & 7]);                  // the oldest of the eight loads in flight
#pragma unroll
          for (int mb = 0; mb < 3; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[j & 7]),
                                                              __builtin_bit_cast(rd_bf16x8, afr[j & 1][mb]), acc[mb], 0, 0, 0);
          {
            // refill with k-step + 8 of the phase (past its end: its last k-step again, never used)
            const int gn = t * 16 + j + 8;
            rd_up2_wload(bq[j & 7], wph + (long)(gn < 127 ? gn : 127) * 4096, wvoff);
          }
        }
      }
      // the clamped refills are still in flight and nobody will read them: wait for them HERE, naming their registers, before
      // anything else is allocated (rdgan_d2slab16.hip.h)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]), "+v"(bq[7]));
      // ---- PixelNorm across the four waves: partial sums of squares of this wave's 32 channels per row -> LDS -> all waves
      float* ssb = (float*)(lds + RD_UP2_SS) + (phase & 1) * (96 * 4);
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) {
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ss = fmaf(acc[mb][r], acc[mb][r], ss);
        ss += __shfl_xor(ss, 32, 64);
        if (lhalf == 0) ssb[(32 * mb + l31) * 4 + nb] = ss;
      }
      __syncthreads();
#pragma unroll
      for (int mb = 0; mb < 3; ++mb) {
        const f32x4 s4 = *(const f32x4*)(ssb + (32 * mb + l31) * 4);
        const float ri = __builtin_amdgcn_rsqf((s4.x + s4.y + s4.z + s4.w) * (1.0f / 128.0f) + 1.0e-8f);     // T:255-266
        const long pix = (((long)b * 12 + 2 * rd_[mb] + pd) * 8 + 2 * rh_[mb] + ph) * 8 + 2 * rw_[mb] + pw;
        if (nb == 0 && lhalf == 0) rinv[pix] = ri;
        char* orow = (char*)out + pix * 256 + nb * 64 + lhalf * 16;
#pragma unroll
        for (int G = 0; G < 4; G += 2) {             // channel groups 8 G .. and 8 (G + 1) .. of the wave's block
          unsigned lo[2], hi[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = acc[mb][4 * (G + u) + e] * ri;
              v[e] = fmaxf(y, RD_LRELU_ALPHA * y);
            }
            lo[u] = rd_pack_bf16(v[0], v[1]); hi[u] = rd_pack_bf16(v[2], v[3]);
          }
          // lanes 0-31 keep their group G and take the upper half's group G; lanes 32-63 take the lower half's group G + 1
          const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], lo[1], false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(hi[0], hi[1], false, false);
          const u32x4_t o = {sx[0], sy[0], sx[1], sy[1]};
          *(u32x4_t*)(orow + G * 16) = o;
        }
      }
    }
  }
}
