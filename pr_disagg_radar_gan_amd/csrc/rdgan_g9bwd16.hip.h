// bf16 storage mode: input gradient of the last generator conv (64 -> 1, 3x3x3 'same', T:345) fused with the PixelNorm + LeakyReLU
// backward of block 3 -- what k_g9_bwd_pairs (rdgan_elem.hip.h) does with 27 x 64 VALU FMAs per grid point (1.28 ms at 2048 samples:
// VALU- and LDS-broadcast-bound at 2.6 TB/s, once the bytes were halved) -- on the fp32 MATRIX pipe:
//   gh3[u][c] = sum_tap dl[u - off(tap)] * w9[tap][c]     =     (W9^T [64 c][28 k]) x (A^T [28 k][positions])
// with v_mfma_f32_32x32x2_f32: EXACT fp32 products of the fp32 dlogits and the fp32 kernel (the dlogits are the root of the whole
// generator backward; rounding them to bf16 moved the step's gradients by 1e-2, splitting them in two bf16 parts cost the packing
// what the bf16 pipe gained -- DESIGN.md 4.7), 28 MFMAs per 32 grid points.  Operands swapped as in the slab kernels: the kernel is
// the A operand (28 registers per lane, loaded once per workgroup), a lane supplies the dlogits neighbour of ITS grid point for tap
// 2 ks + half from the staged planes (one ds_read_b32 per MFMA pair), and ends up with 32 channels of one grid point (lane ^ 32:
// the other 32): the row's PixelNorm + LeakyReLU backward (one cross-half exchange for the mean) and the bf16 rounding run in
// registers, h3 comes in and dy goes out in 16-byte pieces.  No plane-pair sums (the shared-centre backward keeps the VALU kernel).
// A workgroup walks units (sample, PP hour-plane pairs: four at ndomain 16 -- one pair per unit meant two barriers per 16 tiles)
// and keeps the kernel fragments; H * W % 32 == 0, (D / 2) % PP == 0.
#pragma once
#include "rdgan_edge.hip.h"

__global__ void __launch_bounds__(256, 3)
k_g9_bwd_mfma16(const float* __restrict__ dl, const float* __restrict__ w9 /* [27][64] */, const rd_bf16_t* __restrict__ h3,
                const float* __restrict__ rinv, rd_bf16_t* __restrict__ dy, int nunits, int D, int H, int W, int PP) {
  extern __shared__ __attribute__((aligned(16))) float dls[];     // [2 PP + 2][H + 2][W + 2]: planes 2 s - 1 .. 2 (s + PP) with a zero halo
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int Ds = D / 2 / PP, PW = W + 2, PHW = (H + 2) * PW, HW = H * W;

  // kernel fragments: k-step ks, channel block ct: lane (c = 32 ct + l31, half) holds w9[2 ks + half][c] (tap 27: zero);
  // and this lane's staged-plane offset of tap 2 ks + half relative to its grid point in plane A (pl = 1)
  float wf[14][2];
  int toff[14];
#pragma unroll
  for (int ks = 0; ks < 14; ++ks) {
    const int t = 2 * ks + lhalf;
    const int tt = t < 27 ? t : 26;
    const int kd = tt / 9, kh = (tt / 3) % 3, kw = tt % 3;
    toff[ks] = (2 - kd) * PHW + (2 - kh) * PW + (2 - kw);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wf[ks][ct] = t < 27 ? w9[t * 64 + 32 * ct + l31] : 0.f;
  }

  for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    const long b = unit / Ds;
    const int s = (unit - (int)b * Ds) * PP;          // first plane pair of the unit
    __syncthreads();                                  // every wave has left the previous unit's planes
    for (int i = tid; i < (2 * PP + 2) * PHW; i += 256) {
      const int pl = i / PHW, r = i - pl * PHW, hh = r / PW - 1, ww = r % PW - 1, d = 2 * s - 1 + pl;
      float v = 0.f;
      if ((unsigned)d < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v = dl[((b * D + d) * H + hh) * W + ww];
      dls[i] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int tile = wave; tile < 2 * PP * HW / 32; tile += 4) {
      const int pos = tile * 32 + l31;
      const int plane = pos / HW, it = pos - plane * HW;
      const int hh = it / W, ww = it - hh * W;
      const long pix = (b * D + 2 * s + plane) * HW + it;
      // h3 row: 16 bytes per lane (half h takes the 8-channel chunks 2 P + h), then each half hands the other the four channels it
      // does not own: hq[G] = channels 8 G + 4 half + 0..3 of the row (as in k_d2_dgrad_slab16)
      rd_u32x2 hq[8];
      {
        const rd_bf16_t* hrow = h3 + pix * 64 + 8 * lhalf;
#pragma unroll
        for (int P = 0; P < 4; ++P) {
          const u32x4_t w4 = *(const u32x4_t*)(hrow + 16 * P);
          const auto sx = __builtin_amdgcn_permlane32_swap(w4.x, w4.z, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(w4.y, w4.w, false, false);
          hq[2 * P].x = sx[0]; hq[2 * P + 1].x = sx[1];
          hq[2 * P].y = sy[0]; hq[2 * P + 1].y = sy[1];
        }
      }
      const float ri = rinv[pix];
      f32x16 acc[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
      const float* dp = dls + plane * PHW + hh * PW + ww;
#pragma unroll
      for (int ks = 0; ks < 14; ++ks) {
        const float bv = dp[toff[ks]];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ks][ct], bv, acc[ct], 0, 0, 0);
      }
      // ---- backward of [PixelNorm -> LeakyReLU] on the row: n = h > 0 ? h : h / alpha; gn = gh * slope(h);
      // dy = rinv * (gn - n * mean_c(gn * n))     (rd_pn_lrelu_bwd_row)
      float dot = 0.f;
#pragma unroll
      for (int G = 0; G < 8; ++G) {
        const f32x4 hv = rd_unpack_bf16x4(hq[G]);
        const int ct = G >> 2, g = G & 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float hval = hv[e];
          const float n = hval > 0.f ? hval : hval * (1.0f / RD_LRELU_ALPHA);
          const float gn = acc[ct][4 * g + e] * rd_lrelu_slope_from_out(hval);
          acc[ct][4 * g + e] = gn;
          dot = fmaf(gn, n, dot);
        }
      }
      dot += __shfl_xor(dot, 32, 64);
      dot *= (1.0f / 64.0f);
      char* op = (char*)dy + pix * 128 + lhalf * 16;
#pragma unroll
      for (int G = 0; G < 8; G += 2) {
        unsigned lo[2], hi[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 hv = rd_unpack_bf16x4(hq[G + u]);
          const int ct = (G + u) >> 2, g = (G + u) & 3;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hval = hv[e];
            const float n = hval > 0.f ? hval : hval * (1.0f / RD_LRELU_ALPHA);
            o[e] = ri * (acc[ct][4 * g + e] - n * dot);
          }
          lo[u] = rd_pack_bf16(o[0], o[1]); hi[u] = rd_pack_bf16(o[2], o[3]);
        }
        // lanes 0-31 keep their group G and take the upper half's group G; lanes 32-63 take the lower half's group G + 1
        const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], lo[1], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(hi[0], hi[1], false, false);
        const u32x4_t o4 = {sx[0], sy[0], sx[1], sy[1]};
        *(u32x4_t*)(op + G * 16) = o4;
      }
    }
  }
}
