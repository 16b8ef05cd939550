// bf16 storage mode, ndomain 16, one condition channel: forward (and the gradient penalty's second sweep) of the critic's first layer
// (T:286-289: Conv3D(64, 3x3x3, stride 2, 'valid') on the 24 x 16 x 16 x 2 volume -> 11 x 7 x 7 x 64) with a SAMPLE resident in LDS.
//
// Why.  k_d1_gemm_fwd (rdgan_edge.hip.h) builds every 128-row operand tile from 9 x 128 gathered 24-byte segments (three 8-byte
// loads each: an input voxel is fetched ~3.4 times, in pieces), stages it through registers into LDS, multiplies, sends the
// product tile through LDS again for the epilogue -- four barriers per tile, 0.20 ms at 6144 samples for 0.77 GB = 3.8 TB/s, and
// 2.2 TB/s for the second sweep.  The layer is HBM-bound (14 FLOP per byte), so the kernel should look like a copy:
//   * a workgroup takes one sample: its 48 KB input volume arrives ONCE by LDS-DMA, 48 fully coalesced 1 KB pieces;
//   * a lane builds its im2col row from LDS: a row's nine (kd, kh) segments are 24 contiguous bytes at 16-byte aligned addresses
//     (ds_read_b128 + ds_read_b64); lane half 0 takes segments 0-4 (k = 0..29), half 1 segments 5-8 (k = 30..53) -- together the
//     two halves of a lane pair hold the row's B fragments of v_mfma_f32_32x32x16_bf16 (positions as the B operand, as in the
//     slab kernels: the contraction index may be dealt to the (half, k-step, element) slots in any order as long as the kernel
//     image uses the same one), rounded to bf16 exactly as k_d1_gemm_fwd rounds them;
//   * the kernel [54][64] sits in registers as A fragments (8 x 16 bytes per lane, loaded once per workgroup);
//   * operands swapped, so a lane ends up with 32 channels of ONE output row: bias / LeakyReLU / dropout (mode 0) or the gate
//     (mode 1) and the bf16 rounding in registers, 16-byte stores; the layer's 2-bit gate codes (option "d2_gate_bits") leave as
//     one 16-byte store per row (mode 0) and are the gate's source in mode 1.
// Two barriers per sample, three workgroups per CU (49 KB).  Same products as k_d1_gemm_fwd<bf16>, another summation order inside
// the MFMA: outputs agree to one bf16 ulp.
#pragma once
#include "rdgan_edge.hip.h"

#define RD_D1S_IN (24 * 16 * 16 * 2 * 4)          // bytes of a sample's input volume
#define RD_D1S_BIAS RD_D1S_IN                     // 64 floats
#define RD_D1S_LDS (RD_D1S_BIAS + 256)
#define RD_D1S_NPOS 539                           // 11 x 7 x 7 output positions per sample

// MODE 0: out = dropout(LeakyReLU(conv + bias)), gbits (optional) written.  MODE 1: out = gate * conv, gate from gbits (required).
// cin [NB][24][16][16][2] fp32; w [54][64] fp32 (k = (kd*3 + kh)*6 + kw*2 + ci); out [NB][539][64] bf16; gbits [NB][539][16].
template <int MODE>
__global__ void __launch_bounds__(256, 3)
k_d1_fwd_sample16(const float* __restrict__ cin, const float* __restrict__ w, const float* __restrict__ bias,
                  rd_bf16_t* __restrict__ out, unsigned char* __restrict__ gbits, int NB, int use_drop, uint32_t key,
                  uint32_t idx_base) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;

  // ---- the kernel as A fragments: k-step ks, channel block mt: lane (co = 32 mt + l31, half) holds w[30 half + 8 ks + e][co],
  // e = 0..7 (zero past the half's 30 / 24 values)
  u32x4_t wf[4][2];
  {
    const int nk = lhalf ? 24 : 30;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = 8 * ks + e;
          v[e] = i < nk ? w[(30 * lhalf + i) * 64 + 32 * mt + l31] : 0.f;
        }
        wf[ks][mt] = (u32x4_t){rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
      }
  }
  if (tid < 64) *(float*)(lds + RD_D1S_BIAS + tid * 4) = MODE == 0 ? bias[tid] : 0.f;
  const float s1 = use_drop ? (1.0f / 0.75f) : 1.0f, s2 = RD_LRELU_ALPHA * s1;

  for (int s = blockIdx.x; s < NB; s += gridDim.x) {
    __syncthreads();                                  // every wave has left the previous sample (and the bias row is in)
    {
      const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc(cin + (long)s * (RD_D1S_IN / 4));
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int i = wave * 12 + k;                  // wave-uniform: 1 KB piece i
        unsigned voff = (unsigned)(i * 1024 + lane * 16);
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rs, (float*)(lds + i * 1024), (int)voff, 0);
      }
    }
    rd_dma_landed();
    __syncthreads();

#pragma unroll 1
    for (int rb = wave; rb < (RD_D1S_NPOS + 31) / 32; rb += 4) {
      const int row = rb * 32 + l31;
      const bool ok = row < RD_D1S_NPOS;
      const int p = ok ? row : 0;
      const int od = p / 49, q = p - od * 49, oh = q / 7, ow = q - oh * 7;
      // byte offset of voxel (2 od, 2 oh, 2 ow): 8 bytes per voxel; segment (kd, kh) lies (kd * 256 + kh * 16) voxels further
      const int base = (((2 * od) * 16 + 2 * oh) * 16 + 2 * ow) * 8;
      float v[32];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        // half 0: segments 0..4; half 1: segments 5..8 and a fifth that does not exist (zeros)
        const int sg = min(5 * lhalf + j, 8);
        const int kd = sg / 3, kh = sg - 3 * kd;
        const char* a = lds + base + (kd * 256 + kh * 16) * 8;
        const f32x4 x4 = *(const f32x4*)a;
        const float2 x2 = *(const float2*)(a + 16);
        const bool live = j < 4 || lhalf == 0;
        v[6 * j + 0] = live ? x4.x : 0.f; v[6 * j + 1] = live ? x4.y : 0.f; v[6 * j + 2] = live ? x4.z : 0.f;
        v[6 * j + 3] = live ? x4.w : 0.f; v[6 * j + 4] = live ? x2.x : 0.f; v[6 * j + 5] = live ? x2.y : 0.f;
      }
      v[30] = 0.f; v[31] = 0.f;
      f32x16 acc[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *(const f32x4*)(lds + RD_D1S_BIAS + (mt * 32 + 8 * g + 4 * lhalf) * 4);
          acc[mt][4 * g + 0] = b4.x; acc[mt][4 * g + 1] = b4.y; acc[mt][4 * g + 2] = b4.z; acc[mt][4 * g + 3] = b4.w;
        }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4_t bf = {rd_pack_bf16(v[8 * ks + 0], v[8 * ks + 1]), rd_pack_bf16(v[8 * ks + 2], v[8 * ks + 3]),
                            rd_pack_bf16(v[8 * ks + 4], v[8 * ks + 5]), rd_pack_bf16(v[8 * ks + 6], v[8 * ks + 7])};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, wf[ks][mt]), __builtin_bit_cast(rd_bf16x8, bf),
                                                            acc[mt], 0, 0, 0);
      }
      // ---- epilogue in registers: lane (l31, lhalf) holds channels 32 mt + 8 g + 4 lhalf + 0..3 of its row
      const long m = (long)s * RD_D1S_NPOS + p;
      u32x4_t gc = {0u, 0u, 0u, 0u};
      if (MODE == 1) gc = *(const u32x4_t*)(gbits + m * 16);
      unsigned gpart[4] = {0u, 0u, 0u, 0u};
      char* op = (char*)out + m * 128 + lhalf * 16;
#pragma unroll
      for (int G = 0; G < 8; G += 2) {
        unsigned lo[2], hi[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int mt = (G + u) >> 2, g = (G + u) & 3;
          float t[4];
          if (MODE == 0) {
            const uint32_t idx = (uint32_t)(m * 64) + (uint32_t)(32 * mt + 8 * g + 4 * lhalf);
            const uint32_t word = use_drop ? rd_drop_word(key, idx + idx_base) : 0u;
            unsigned code = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = rd_lrelu(acc[mt][4 * g + e]);
              if (use_drop) x = rd_drop_apply_w(x, word, e);
              t[e] = x;
              code |= ((x > 0.f ? 1u : 0u) | ((use_drop && __builtin_bit_cast(unsigned, x) == 0u) ? 2u : 0u)) << (2 * e);
            }
            // quad 2 (G + u) + lhalf of the row = byte (2 ((G + u) & 1) + lhalf) of dword (G + u) >> 1
            gpart[(G + u) >> 1] |= code << (16 * ((G + u) & 1) + 8 * lhalf);
          } else {
            const unsigned byte = gc[(G + u) >> 1] >> (16 * ((G + u) & 1) + 8 * lhalf);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned code = (byte >> (2 * e)) & 3u;
              t[e] = acc[mt][4 * g + e] * ((code & 2u) ? 0.f : ((code & 1u) ? s1 : s2));
            }
          }
          lo[u] = rd_pack_bf16(t[0], t[1]); hi[u] = rd_pack_bf16(t[2], t[3]);
        }
        // lanes 0-31 keep their group G and take the upper half's group G; lanes 32-63 take the lower half's group G + 1
        const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], lo[1], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(hi[0], hi[1], false, false);
        const u32x4_t o = {sx[0], sy[0], sx[1], sy[1]};
        if (ok) *(u32x4_t*)(op + G * 16) = o;
      }
      if (MODE == 0 && gbits) {
        u32x4_t full;
#pragma unroll
        for (int j = 0; j < 4; ++j) full[j] = gpart[j] | (unsigned)__shfl_xor((int)gpart[j], 32, 64);
        if (ok && lhalf == 0) *(u32x4_t*)(gbits + m * 16) = full;
      }
    }
  }
}
