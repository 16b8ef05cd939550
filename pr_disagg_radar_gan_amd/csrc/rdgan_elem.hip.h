// HBM-bound elementwise / reduction kernels of the cWGAN-GP step (gfx950, wave64).
// T = gan_train_cwgangp_pixelnorm.py in the reference.
#pragma once
#include <hip/hip_runtime.h>
#include "rdgan_gemm.hip.h"

#define RD_PIXELNORM_EPS 1.0e-8f

__device__ __forceinline__ float rd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LP>
__device__ __forceinline__ float rd_seg_sum(float v) {   // sum over aligned groups of LP lanes
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void rd_store_bf16x4(unsigned short* p, f32x4 v) { rd_st4(p, v); }
// block-wide sum, result valid in thread 0 (256 threads)
__device__ __forceinline__ float rd_block_sum(float v, float* red) {
  v = rd_wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  __syncthreads();
  return s;
}

// G0 (T:322-323): out[b] = [z[b] (nz) | cond[b].flatten (nc)]
// (also clears the non-finite flag of the call it opens: a 4-byte hipMemsetAsync is a launch of its own, ~9 us)
// out16 (optional, bf16 storage mode with the Dense layer on the bf16 matrix pipe): the same rows rounded to bf16, row length KP >= nz + nc
// (a multiple of 64: the K chunk of the bf16 GEMM), zero padded
__global__ void k_concat(const float* __restrict__ z, const float* __restrict__ cond, float* __restrict__ out,
                         int B, int nz, int nc, int* __restrict__ zero_flag, unsigned short* __restrict__ out16 = nullptr, int KP = 0) {
  if (zero_flag && blockIdx.x == 0 && threadIdx.x == 0) *zero_flag = 0;
  const int w = nz + nc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)B * w; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / w), j = (int)(i - (long)b * w);
    const float v = j < nz ? z[(long)b * nz + j] : cond[(long)b * nc + (j - nz)];
    out[i] = v;
    if (out16) {
      out16[(long)b * KP + j] = __builtin_bit_cast(unsigned short, (__bf16)v);
      if (j < KP - w) out16[(long)b * KP + w + j] = 0;          // the pad columns (KP - w <= 63 < w)
    }
  }
}

// G7+G8 (T:255-266, T:333): h = LeakyReLU(y / sqrt(mean_c(y^2) + 1e-8)); rinv = 1/sqrt(..) kept for backward.
// In place allowed (h == y).  LP = C/4 lanes per pixel.
template <int LP, typename T = float>
__global__ void k_pixelnorm_lrelu_fwd(const T* y, T* h, float* __restrict__ rinv, long npix) {
  constexpr int C = LP * 4;
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long pix = gid / LP;
  const int sub = (int)(gid % LP);
  const bool ok = pix < npix;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok) v = rd_ld4(y + pix * C + sub * 4);
  float ss = rd_seg_sum<LP>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
  float l2 = sqrtf(ss * (1.0f / C) + RD_PIXELNORM_EPS);
  if (ok) {
    f32x4 o;
    o.x = rd_lrelu(v.x / l2); o.y = rd_lrelu(v.y / l2); o.z = rd_lrelu(v.z / l2); o.w = rd_lrelu(v.w / l2);
    rd_st4(h + pix * C + sub * 4, o);
    if (rinv && sub == 0) rinv[pix] = 1.0f / l2;
  }
}

// backward of [PixelNorm -> LeakyReLU] given h (the block output) and rinv:
//   n = h>0 ? h : h/alpha ; gn = gh*slope(h) ; dy = rinv*(gn - n*mean_c(gn*n))
// gh is either given on the same grid (POOL=0) or as the gradient on the 2x upsampled grid of the
// next block's conv input, in which case the 8 children are summed first (adjoint of UpSampling3D, T:335).
template <int LP>
__device__ __forceinline__ f32x4 rd_pn_lrelu_bwd_row(f32x4 gh, f32x4 hv, float ri) {
  constexpr int C = LP * 4;
  f32x4 n, gn;
  n.x = hv.x > 0.f ? hv.x : hv.x * (1.0f / RD_LRELU_ALPHA);
  n.y = hv.y > 0.f ? hv.y : hv.y * (1.0f / RD_LRELU_ALPHA);
  n.z = hv.z > 0.f ? hv.z : hv.z * (1.0f / RD_LRELU_ALPHA);
  n.w = hv.w > 0.f ? hv.w : hv.w * (1.0f / RD_LRELU_ALPHA);
  gn.x = gh.x * rd_lrelu_slope_from_out(hv.x);
  gn.y = gh.y * rd_lrelu_slope_from_out(hv.y);
  gn.z = gh.z * rd_lrelu_slope_from_out(hv.z);
  gn.w = gh.w * rd_lrelu_slope_from_out(hv.w);
  const float dot = rd_seg_sum<LP>(gn.x * n.x + gn.y * n.y + gn.z * n.z + gn.w * n.w) * (1.0f / C);
  f32x4 o;
  o.x = ri * (gn.x - n.x * dot); o.y = ri * (gn.y - n.y * dot);
  o.z = ri * (gn.z - n.z * dot); o.w = ri * (gn.w - n.w * dot);
  return o;
}
template <int LP, int POOL, typename T = float>
__global__ void k_pn_lrelu_bwd(const T* __restrict__ g, const T* __restrict__ h, const float* __restrict__ rinv,
                               T* __restrict__ dy, long npix, int D, int H, int W) {
  constexpr int C = LP * 4;
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long pix = gid / LP;
  const int sub = (int)(gid % LP);
  const bool ok = pix < npix;
  f32x4 gh = {0.f, 0.f, 0.f, 0.f}, hv = {0.f, 0.f, 0.f, 0.f};
  float ri = 0.f;
  if (ok) {
    hv = rd_ld4(h + pix * C + sub * 4);
    ri = rinv[pix];
    if (POOL) {
      long t = pix;
      int w = (int)(t % W); t /= W;
      int hh = (int)(t % H); t /= H;
      int d = (int)(t % D);
      long b = t / D;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        long up = (((b * (2 * D) + 2 * d + (e >> 2)) * (2 * H) + 2 * hh + ((e >> 1) & 1)) * (2 * W) + 2 * w + (e & 1));
        gh += rd_ld4(g + up * C + sub * 4);
      }
    } else {
      gh = rd_ld4(g + pix * C + sub * 4);
    }
  }
  const f32x4 o = rd_pn_lrelu_bwd_row<LP>(gh, hv, ri);
  if (ok) rd_st4(dy + pix * C + sub * 4, o);
}
// same (POOL = 0) over PAIRS of hour planes (2s, 2s+1) of a block output, additionally writing their sum
// gS[b][s][h][w][:] = dy[b][2s][h][w][:] + dy[b][2s+1][h][w][:] for the shared-centre backward (k_presum_d fused in).
// npair = B * Ds * HW pixel pairs, HW = pixels per hour plane.
template <int LP, typename T = float>
__global__ void k_pn_lrelu_bwd_pairs(const T* __restrict__ g, const T* __restrict__ h, const float* __restrict__ rinv,
                                     T* __restrict__ dy, T* __restrict__ gS, long npair, long HW) {
  constexpr int C = LP * 4;
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long pr = gid / LP;
  const int sub = (int)(gid % LP);
  const bool ok = pr < npair;
  const long bs = pr / HW, hw = pr - bs * HW;
  const long pixA = (2 * bs) * HW + hw, pixB = pixA + HW;
  f32x4 ga = {0.f, 0.f, 0.f, 0.f}, ha = ga, gb = ga, hb = ga;
  float ra = 0.f, rb = 0.f;
  if (ok) {
    ga = rd_ld4(g + pixA * C + sub * 4); ha = rd_ld4(h + pixA * C + sub * 4); ra = rinv[pixA];
    gb = rd_ld4(g + pixB * C + sub * 4); hb = rd_ld4(h + pixB * C + sub * 4); rb = rinv[pixB];
  }
  const f32x4 oa = rd_pn_lrelu_bwd_row<LP>(ga, ha, ra);
  const f32x4 ob = rd_pn_lrelu_bwd_row<LP>(gb, hb, rb);
  if (ok) {
    rd_st4(dy + pixA * C + sub * 4, oa);
    rd_st4(dy + pixB * C + sub * 4, ob);
    rd_st4(gS + pr * C + sub * 4, oa + ob);
  }
}

// Backward of the last generator conv (64 -> 1, 3^3 'same', T:345) fused with the PixelNorm+LeakyReLU backward of block 3,
// straight from the 1-channel dlogits dl (no im2col matrix, no intermediate gradient tensor):
//   gh3[u][c] = sum_tap dl[u - off(tap)] * w9[tap][c];   dy = pn_lrelu_bwd(gh3, h3, rinv);   gS = dy[2s] + dy[2s+1]
// One workgroup per (sample, hour-plane pair s); the four dl planes 2s-1 .. 2s+2 sit in LDS with a zero halo, a thread
// owns one channel quad (its 27 x 4 kernel weights in registers) and walks the plane's pixels 16 at a time.
// gS is optional (shared-centre backward).  H*W % 16 == 0.  T = element type of h3 / dy / gS (bf16 storage mode).
template <typename T = float>
__global__ void __launch_bounds__(256)
k_g9_bwd_pairs(const float* __restrict__ dl, const float* __restrict__ w9, const T* __restrict__ h3,
               const float* __restrict__ rinv, T* __restrict__ dy, T* __restrict__ gS, int D, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float dls[];     // [4][H+2][W+2]
  const int Ds = D / 2, PW = W + 2, PHW = (H + 2) * PW, HW = H * W;
  const long b = blockIdx.x / Ds;
  const int s = blockIdx.x % Ds;
  for (int i = threadIdx.x; i < 4 * PHW; i += 256) {
    const int pl = i / PHW, r = i - pl * PHW, hh = r / PW - 1, ww = r % PW - 1, d = 2 * s - 1 + pl;
    float v = 0.f;
    if ((unsigned)d < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v = dl[((b * D + d) * H + hh) * W + ww];
    dls[i] = v;
  }
  const int c4 = (threadIdx.x & 15) * 4;
  f32x4 wq[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) wq[t] = *(const f32x4*)(w9 + t * 64 + c4);
  __syncthreads();
  for (int it = threadIdx.x >> 4; it < HW; it += 16) {
    const int hh = it / W, ww = it - hh * W;
    f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gb = ga;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int o = (hh + 2 - kh) * PW + (ww + 2 - kw);
          const float da = dls[(2 - kd) * PHW + o], db = dls[(3 - kd) * PHW + o];
          const f32x4 w = wq[(kd * 3 + kh) * 3 + kw];
          ga += da * w; gb += db * w;
        }
    const long pixA = (b * D + 2 * s) * HW + it, pixB = pixA + HW, pr = (b * Ds + s) * HW + it;
    const f32x4 ha = rd_ld4(h3 + pixA * 64 + c4), hb = rd_ld4(h3 + pixB * 64 + c4);
    const f32x4 oa = rd_pn_lrelu_bwd_row<16>(ga, ha, rinv[pixA]);
    const f32x4 ob = rd_pn_lrelu_bwd_row<16>(gb, hb, rinv[pixB]);
    rd_st4(dy + pixA * 64 + c4, oa);
    rd_st4(dy + pixB * 64 + c4, ob);
    if (gS) rd_st4(gS + pr * 64 + c4, oa + ob);
  }
}
// Weight gradient of the same conv without the im2col matrix: dW9[tap][c] = sum_u dl[u - off(tap)] * h3[u][c].
// Same decomposition; each thread accumulates its channel quad over its pixels, the 16 pixel slots of the workgroup are
// folded by shuffles and LDS, and partial[blockIdx][27][64] is folded by k_reduce_partials (deterministic).
// A workgroup walks the units (sample, plane pair) blockIdx.x, blockIdx.x + gridDim.x, ... < nunits and keeps its sums in
// registers, so the number of partial slabs is the grid size, not B * D/2 (24 576 slabs at bs 2048 cost 0.35 ms to fold).
template <typename T = float>
__global__ void __launch_bounds__(256)
k_g9_wgrad_pairs(const float* __restrict__ dl, const T* __restrict__ h3, float* __restrict__ partial, int D, int H, int W,
                 int nunits) {
  extern __shared__ __attribute__((aligned(16))) float dls[];     // [4][H+2][W+2], reused for the fold
  const int Ds = D / 2, PW = W + 2, PHW = (H + 2) * PW, HW = H * W;
  const int c4 = (threadIdx.x & 15) * 4;
  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
  const long b = unit / Ds;
  const int s = unit % Ds;
  __syncthreads();                                       // the previous unit's plane reads are done
  for (int i = threadIdx.x; i < 4 * PHW; i += 256) {
    const int pl = i / PHW, r = i - pl * PHW, hh = r / PW - 1, ww = r % PW - 1, d = 2 * s - 1 + pl;
    float v = 0.f;
    if ((unsigned)d < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v = dl[((b * D + d) * H + hh) * W + ww];
    dls[i] = v;
  }
  __syncthreads();
  // (tried: batches of 2 / 4 pixels with their activation loads up front -- 218 / 256+ VGPRs, lower occupancy, no gain)
  for (int it = threadIdx.x >> 4; it < HW; it += 16) {
    const int hh = it / W, ww = it - hh * W;
    const long pixA = (b * D + 2 * s) * HW + it;
    const f32x4 ha = rd_ld4(h3 + pixA * 64 + c4), hb = rd_ld4(h3 + (pixA + HW) * 64 + c4);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int o = (hh + 2 - kh) * PW + (ww + 2 - kw);
          acc[(kd * 3 + kh) * 3 + kw] += dls[(2 - kd) * PHW + o] * ha + dls[(3 - kd) * PHW + o] * hb;
        }
  }
  }
  // fold the 4 pixel slots of a wave (lanes l, l^16, l^32 share a channel quad), then the 4 waves through LDS
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[t][e];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[t][e] = v;
    }
  __syncthreads();                                       // dl planes no longer needed
  f32x4* red = (f32x4*)dls;                               // [4 waves][27][16 quads]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 16)
#pragma unroll
    for (int t = 0; t < 27; ++t) red[(wave * 27 + t) * 16 + lane] = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < 27 * 16; i += 256) {
    f32x4 v = red[i] + red[27 * 16 + i] + red[2 * 27 * 16 + i] + red[3 * 27 * 16 + i];
    *(f32x4*)(partial + (long)blockIdx.x * 1728 + (long)i * 4) = v;
  }
}

// gradient wrt the Dense pre-activation (T:326-328): sum the 8 children of the first block's
// upsampled-grid gradient and apply LeakyReLU' from the stored output h0.  C = 256.
__global__ void k_pool_lrelu_bwd(const float* __restrict__ gup, const float* __restrict__ h0, float* __restrict__ out,
                                 long npix, int D, int H, int W, int C) {
  const int c4s = C / 4;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < npix * c4s; f += (long)gridDim.x * blockDim.x) {
    long pix = f / c4s;
    int c = (int)(f - pix * c4s) * 4;
    long t = pix;
    int w = (int)(t % W); t /= W;
    int hh = (int)(t % H); t /= H;
    int d = (int)(t % D);
    long b = t / D;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      long up = (((b * (2 * D) + 2 * d + (e >> 2)) * (2 * H) + 2 * hh + ((e >> 1) & 1)) * (2 * W) + 2 * w + (e & 1));
      s += *(const f32x4*)(gup + up * C + c);
    }
    f32x4 hv = *(const f32x4*)(h0 + pix * C + c);
    s.x *= rd_lrelu_slope_from_out(hv.x); s.y *= rd_lrelu_slope_from_out(hv.y);
    s.z *= rd_lrelu_slope_from_out(hv.z); s.w *= rd_lrelu_slope_from_out(hv.w);
    *(f32x4*)(out + pix * C + c) = s;
  }
}

// G9+G10+G11 (T:345-350): logits[pos] = bias + sum_tap P[pos+tap-1][tap] from the column GEMM
// P[pos][32] = h3[pos][:] . W9[:, tap]; then softmax over the 24 hours of each grid point.
// One thread per (sample, h, w) column.  Sets *nonfinite if any output is NaN/Inf.
__global__ void k_colgather_softmax(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ out,
                                    int B, int D, int H, int W, int* __restrict__ nonfinite) {
  const long ncol = (long)B * H * W;
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (gid >= ncol) return;
  const int w = (int)(gid % W), h = (int)((gid / W) % H);
  const long b = gid / ((long)W * H);
  const float bv = bias[0];
  const long hw = (long)H * W;
  float* o = out + b * D * hw + (long)h * W + w;
  float mx = -3.0e38f;
  for (int d = 0; d < D; ++d) {
    float s = bv;
    for (int td = 0; td < 3; ++td) {
      int sd = d + td - 1;
      if ((unsigned)sd >= (unsigned)D) continue;
      for (int th = 0; th < 3; ++th) {
        int shh = h + th - 1;
        if ((unsigned)shh >= (unsigned)H) continue;
        for (int tw = 0; tw < 3; ++tw) {
          int sw = w + tw - 1;
          if ((unsigned)sw >= (unsigned)W) continue;
          long pos = ((b * D + sd) * H + shh) * W + sw;
          s += P[pos * 32 + (td * 3 + th) * 3 + tw];
        }
      }
    }
    o[d * hw] = s;               // raw logit, re-read by this same thread below
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
  for (int d = 0; d < D; ++d) { float e = expf(o[d * hw] - mx); o[d * hw] = e; den += e; }
  bool bad = false;
  for (int d = 0; d < D; ++d) {
    float p = o[d * hw] / den;
    bad |= !(fabsf(p) <= 3.0e38f);
    o[d * hw] = p;
  }
  if (bad) atomicOr(nonfinite, 1);
}

// G9+G10+G11 after the tap-gathering column GEMM (RD_EPI_TAPGATHER): Q[b][d][NQ][h][w] holds, per grid point, the sums
// over (kh,kw) for each kd (NQ = 3) or over kw for each (kd,kh) (NQ = 9).  logits = bias + the remaining sum over
// kd (and kh), then softmax over the 24 hours.  FOUR lanes per (sample, h, w) column, six hours each (lanes l, l+16, l+32,
// l+48 of a wave share a column: 16 consecutive w per quarter wave, max and sum folded by two xor-shuffles): with one thread
// per column the launch was 1 wave per SIMD walking 72 loads (34 us for 25 MB).
template <int D, int NQ>
__global__ void __launch_bounds__(256)
k_tapsum_softmax(const float* __restrict__ Q, const float* __restrict__ bias, float* __restrict__ out,
                 int B, int H, int W, int* __restrict__ nonfinite) {
  static_assert(D % 4 == 0, "hours per lane");
  constexpr int DP = D / 4;
  const long ncol = (long)B * H * W;
  const int lane = threadIdx.x & 63, part = lane >> 4;
  const long col = (blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (lane & 15);
  const bool live = col < ncol;
  const long gid = live ? col : 0;
  const int w = (int)(gid % W), h = (int)((gid / W) % H);
  const long b = gid / ((long)W * H);
  const float bv = bias[0];
  const long hw = (long)H * W;
  const float* q = Q + b * D * NQ * hw;
  float lg[DP];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    const int d = part * DP + i;
    float s = bv;
#pragma unroll
    for (int td = 0; td < 3; ++td) {
      const int sd = d + td - 1;
      if (sd < 0 || sd >= D) continue;
      if constexpr (NQ == 3) {
        s += q[((long)sd * 3 + td) * hw + (long)h * W + w];
      } else {
#pragma unroll
        for (int th = 0; th < 3; ++th) {
          const int shh = h + th - 1;
          if ((unsigned)shh < (unsigned)H) s += q[((long)sd * 9 + td * 3 + th) * hw + (long)shh * W + w];
        }
      }
    }
    lg[i] = s;
    mx = fmaxf(mx, s);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < DP; ++i) { lg[i] = expf(lg[i] - mx); den += lg[i]; }
  // (fixed order of the four partial sums: (p0 + p1) + (p2 + p3) on every lane)
  den += __shfl_xor(den, 16, 64);
  den += __shfl_xor(den, 32, 64);
  bool bad = false;
  float* o = out + b * D * hw + (long)h * W + w;
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    const float p = lg[i] / den;
    bad |= !(fabsf(p) <= 3.0e38f);
    if (live) o[(part * DP + i) * hw] = p;
  }
  if (bad && live) atomicOr(nonfinite, 1);
}

// The same tail behind the slab kernel with the fused last conv (k_upconv_slab16<.., G9 = true>, bf16 storage mode, ndomain 16):
// QT[b][item][tp][p][q][64]: work item `item` = planes 4 item .. 4 item + 3; tp = 0..5 = TARGET plane 4 item - 1 + tp; the sum over
// (kh, kw) and over the hour taps inside the item of the tap products that SOURCE parity class p = 2 (y & 1) + (x & 1) sends to the
// grid points of target class q (class position (y >> 1) * 8 + (x >> 1)).  logit(d, y, x) = bias + [the item below's tp 5] + the
// own item's tp (d & 3) + 1 + [the item above's tp 0], four source classes each, in that fixed order; then the softmax over hours.
// A lane owns FOUR consecutive class positions (one 16-byte load per Q12 row) and six hours; lanes l, l + 16, l + 32, l + 48 share
// the columns as in k_tapsum_softmax (with one column and 4-byte loads per lane the 0.6 GB of Q12 came in at 3.5 TB/s).
template <int D>
__global__ void __launch_bounds__(256)
k_tapsum_softmax12(const float* __restrict__ Q12, const float* __restrict__ bias, float* __restrict__ out, int B,
                   int* __restrict__ nonfinite) {
  static_assert(D % 4 == 0, "hours per lane");
  constexpr int DP = D / 4, H = 16, W = 16;
  const long ngrp = (long)B * (H * W / 4);                   // groups of four columns
  const int lane = threadIdx.x & 63, part = lane >> 4;
  const long grp = (blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (lane & 15);
  const bool live = grp < ngrp;
  const long gid = live ? grp : 0;
  // columns of a sample in class-major order: cidx = q * 64 + class position
  const int cidx = (int)(gid % (W * H / 4)) * 4, qc = cidx >> 6, pos = cidx & 63;
  const int h = 2 * (pos >> 3) + (qc >> 1), w0 = 2 * (pos & 7) + (qc & 1);      // the four columns: w0, w0 + 2, w0 + 4, w0 + 6
  const long b = gid / (W * H / 4);
  const float bv = bias[0];
  const float* q = Q12 + b * (D / 4 * 6 * 1024) + cidx;
  f32x4 lg[DP];
  f32x4 mx = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    const int d = part * DP + i;
    f32x4 s = {bv, bv, bv, bv};
    const int it = d >> 2, k = d & 3;                 // the plane's work item and its place in it
    if (k == 0 && it > 0) {                           // the item below reaches it through its last plane (hour tap kd = 0)
#pragma unroll
      for (int p = 0; p < 4; ++p) s += *(const f32x4*)(q + ((long)((it - 1) * 6 + 5) * 4 + p) * 256);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) s += *(const f32x4*)(q + ((long)(it * 6 + k + 1) * 4 + p) * 256);
    if (k == 3 && it < D / 4 - 1) {                   // the item above through its first plane (kd = 2)
#pragma unroll
      for (int p = 0; p < 4; ++p) s += *(const f32x4*)(q + ((long)((it + 1) * 6) * 4 + p) * 256);
    }
    lg[i] = s;
#pragma unroll
    for (int c = 0; c < 4; ++c) mx[c] = fmaxf(mx[c], s[c]);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], 16, 64));
    mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], 32, 64));
  }
  f32x4 den = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < DP; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { lg[i][c] = expf(lg[i][c] - mx[c]); den[c] += lg[i][c]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    den[c] += __shfl_xor(den[c], 16, 64);
    den[c] += __shfl_xor(den[c], 32, 64);
  }
  bool bad = false;
  const long hw = (long)H * W;
  float* o = out + b * D * hw + (long)h * W + w0;
#pragma unroll
  for (int i = 0; i < DP; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float pr = lg[i][c] / den[c];
      bad |= !(fabsf(pr) <= 3.0e38f);
      if (live) o[(part * DP + i) * hw + 2 * c] = pr;
    }
  if (bad && live) atomicOr(nonfinite, 1);
}

// softmax-over-hours backward: dl = p * (g - sum_d p*g), one thread per (sample,h,w) column.
__global__ void k_softmax_bwd(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ dl,
                              int B, int D, int H, int W) {
  const long ncol = (long)B * H * W;
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (gid >= ncol) return;
  const long hw = (long)H * W;
  const long b = gid / hw, r = gid - b * hw;
  const long base = b * D * hw + r;
  float pv[24], dot = 0.f;
#pragma unroll
  for (int d = 0; d < 24; ++d) { pv[d] = p[base + d * hw]; dot += pv[d] * g[base + d * hw]; }
#pragma unroll
  for (int d = 0; d < 24; ++d) dl[base + d * hw] = pv[d] * (g[base + d * hw] - dot);
}

// im2col of the 1-channel dlogits for the G9 input/weight gradients:
// col[pos][tap] = dl[pos + 1 - tap] (zero outside), columns 27..31 zero.
__global__ void k_dl_im2col(const float* __restrict__ dl, float* __restrict__ col, int B, int D, int H, int W) {
  const long total = (long)B * D * H * W * 8;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    long pos = f >> 3;
    int t0 = (int)(f & 7) * 4;
    long t = pos;
    int w = (int)(t % W); t /= W;
    int h = (int)(t % H); t /= H;
    int d = (int)(t % D);
    long b = t / D;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int tap = t0 + e;
      float v = 0.f;
      if (tap < 27) {
        int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
        int sd = d + 1 - td, shh = h + 1 - th, sw = w + 1 - tw;
        if ((unsigned)sd < (unsigned)D && (unsigned)shh < (unsigned)H && (unsigned)sw < (unsigned)W)
          v = dl[((b * D + sd) * H + shh) * W + sw];
      }
      o[e] = v;
    }
    *(f32x4*)(col + pos * 32 + t0) = o;
  }
}

// D0 + X1 (T:275-282, T:221-224): critic input with CP floats per voxel = (sample | nc condition channels repeated
// over the 24 hours | zero padding): CP = 2 for nc = 1, CP = 4 for nc = 2 or 3 (revision1/additional_inputs variants).
// mode 0: out[0:B] = real, out[B:2B] = fake, out[2B:3B] = alpha*real + (1-alpha)*fake, alpha = uniform(key, alpha_base + b)
// (alpha_base = global index of this rank's first sample, option "sample_offset")
// mode 1: out[0:B] = fake only (generator step);  mode 2: out[0:B] = real only (critic.predict).
__global__ void k_build_critic_input(const float* __restrict__ real, const float* __restrict__ fake,
                                     const float* __restrict__ cond, float* __restrict__ out, int B, int D, int HW,
                                     int nc, int CP, int mode, uint32_t alpha_key, uint32_t alpha_base) {
  const long per = (long)D * HW;
  const long total = (long)B * per;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    long b = f / per;
    long r = f - b * per;
    int hw = (int)(r % HW);
    float c[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < nc; ++k) c[k] = cond[(b * HW + hw) * nc + k];
    auto put = [&](long vox, float x) {
      float* o = out + vox * CP;
      o[0] = x;
      for (int k = 1; k < CP; ++k) o[k] = k - 1 < nc ? c[k - 1] : 0.f;
    };
    if (mode == 0) {
      float rl = real[f], fk = fake[f];
      float a = rd_uniform(alpha_key, alpha_base + (uint32_t)b);
      put(f, rl); put(total + f, fk); put(2 * total + f, a * rl + (1.0f - a) * fk);
    } else if (mode == 1) {
      put(f, fake[f]);
    } else {
      put(f, real[f]);
    }
  }
}

// The same for one condition channel (CP = 2, nc = 1), four voxels per thread: 16-byte loads of real / fake / cond, two 16-byte
// stores per destination, 32-bit index arithmetic (the kernel above divides two 64-bit numbers per voxel and moves 4 bytes per
// instruction: 4.4 TB/s at 6144 samples).  HW % 4 == 0, B * D * HW < 2^31.
__global__ void k_build_critic_input_v4(const float* __restrict__ real, const float* __restrict__ fake,
                                        const float* __restrict__ cond, float* __restrict__ out, int B, int D, int HW,
                                        int mode, uint32_t alpha_key, uint32_t alpha_base) {
  const unsigned per4 = (unsigned)(D * HW) / 4u, hw4 = (unsigned)HW / 4u, total4 = (unsigned)B * per4;
  for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += gridDim.x * blockDim.x) {
    const unsigned b = q / per4, r = q - b * per4, h4 = r % hw4;
    const f32x4 c = *(const f32x4*)(cond + ((long)b * hw4 + h4) * 4);
    auto put = [&](long q4, f32x4 x) {
      float* o = out + q4 * 8;
      *(f32x4*)o = (f32x4){x.x, c.x, x.y, c.y};
      *(f32x4*)(o + 4) = (f32x4){x.z, c.z, x.w, c.w};
    };
    if (mode == 0) {
      const f32x4 rl = *(const f32x4*)(real + (long)q * 4), fk = *(const f32x4*)(fake + (long)q * 4);
      const float a = rd_uniform(alpha_key, alpha_base + b);
      put(q, rl); put((long)total4 + q, fk); put(2L * total4 + q, a * rl + (1.0f - a) * fk);
    } else if (mode == 1) {
      put(q, *(const f32x4*)(fake + (long)q * 4));
    } else {
      put(q, *(const f32x4*)(real + (long)q * 4));
    }
  }
}

// D6 (T:303-304): v[b] = h4[b,:] . w + bias; one block per sample.
template <typename T = float>
__global__ void k_critic_dense_fwd(const T* __restrict__ h4, const float* __restrict__ w, const float* __restrict__ bias,
                                   float* __restrict__ v, int F) {
  __shared__ float red[4];
  const long b = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < F; i += blockDim.x) s += rd_ld1(h4 + b * F + i) * w[i];
  s = rd_block_sum(s, red);
  if (threadIdx.x == 0) v[b] = s + bias[0];
}

// per-sample output gradient of the three critic passes (T:388-392, T:452-454):
// real: d(mean(-v))/dv = -1/B ; fake: +1/B ; interpolated: 1 (dD/dx_hat for the penalty) ;
// generator step (mode 1): d(mean(-v))/dv = -1/B.
__device__ __forceinline__ float rd_dv(int b, int B, int mode) {
  if (mode == 1) return -1.0f / B;
  return b < B ? -1.0f / B : (b < 2 * B ? 1.0f / B : 1.0f);
}

// top of the critic's input-gradient chain: u4 = gate(h4) * w6 * dv(sample)
template <typename T = float>
__global__ void k_critic_top_bwd(const T* __restrict__ h4, const float* __restrict__ w, T* __restrict__ u4,
                                 int NB, int F, int B, int mode, int use_drop, uint32_t key) {
  const long total = (long)NB * F;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    int b = (int)(f / F), i = (int)(f - (long)b * F);
    const float g = rd_gate_from_out(rd_ld1(h4 + f), use_drop);
    rd_st1(u4 + f, g * w[i] * rd_dv(b, B, mode));
  }
}

// dW6[i] = sum_b buf[b][i] * dv(b) over the 3B batch whose last third holds r4;
// block = 16 columns x 16 sample groups (F / 16 workgroups: the matrix is short and wide); gridDim.y = RS row slices, slice y
// writes out[y][F] (RS == 1: the gradient itself; RS > 1 -- thousands of samples, where one slice is a 98-us chain of
// dependent loads -- partial sums folded by k_reduce_partials)
template <typename T = float>
__global__ void __launch_bounds__(256)
k_critic_dense_wgrad(const T* __restrict__ buf, float* __restrict__ out, int NB, int F, int B) {
  __shared__ float red[256];
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
  const int per = (NB + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(NB, b0 + per);
  float s0 = 0.f, s1 = 0.f;
  if (i < F) {
    int b = b0 + g;
    for (; b + 16 < b1; b += 32) {
      s0 += rd_ld1(buf + (long)b * F + i) * rd_dv(b, B, 0);
      s1 += rd_ld1(buf + (long)(b + 16) * F + i) * rd_dv(b + 16, B, 0);
    }
    if (b < b1) s0 += rd_ld1(buf + (long)b * F + i) * rd_dv(b, B, 0);
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (g == 0 && i < F) {
    float s = red[threadIdx.x];
    for (int j = 1; j < 16; ++j) s += red[j * 16 + threadIdx.x];
    out[(long)blockIdx.y * F + i] = s;
  }
}

// column sums of rows [0,rows) of a [rows][C] matrix, two deterministic stages.
// stage 1: CG = C/4 float4 column groups x RG = 256/CG row groups per block; each thread streams float4s
// down its rows, the row groups are folded through LDS, and partial[blk][C] is written.
// NT = threads per workgroup: 256, or 1024 for long inputs -- at most 256 partial rows keep the fold a single round trip, and a
// tensor of hundreds of MB then needs the bytes in flight of 1024 threads per CU (256 workgroups x 256 threads x four 8-byte
// loads read the 805 MB bf16 gradient of ndomain 64's block 3 at 1.9 TB/s)
template <int CG, typename T = float, int NT = 256>
__global__ void __launch_bounds__(NT)
k_colsum_partial(const T* __restrict__ src, long rows, float* __restrict__ partial, long rows_per_blk) {
  constexpr int RG = NT / CG, C = CG * 4;
  __shared__ f32x4 red[NT];
  const int cg = threadIdx.x % CG, rg = threadIdx.x / CG;
  const long r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  long r = r0 + rg;
  // four independent chains: four 16-byte loads in flight per thread (round 4: at most 256 workgroups, so that the fold of the
  // partial rows is ONE round trip -- with 1024 workgroups and two chains the fold kernel was 24 us on 4-16 workgroups)
  for (; r + 3 * RG < r1; r += 4 * RG) {
    const f32x4 a0 = rd_ld4(src + r * C + cg * 4), a1 = rd_ld4(src + (r + RG) * C + cg * 4);
    const f32x4 a2 = rd_ld4(src + (r + 2 * RG) * C + cg * 4), a3 = rd_ld4(src + (r + 3 * RG) * C + cg * 4);
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; r < r1; r += RG) s0 += rd_ld4(src + r * C + cg * 4);
  s0 = (s0 + s1) + (s2 + s3);
  s1 = f32x4{0.f, 0.f, 0.f, 0.f};
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (rg == 0) {
    f32x4 t = red[cg];
#pragma unroll 4
    for (int k = 1; k < RG; ++k) t += red[k * CG + cg];
    *(f32x4*)(partial + (long)blockIdx.x * C + cg * 4) = t;
  }
}
// generic (any C): one thread per column, used for the few wide-and-short cases (Dense bias: 256 x 3072);
// grid = (row blocks, column chunks of blockDim.x)
__global__ void k_colsum_partial_any(const float* __restrict__ src, long rows, int C, float* __restrict__ partial,
                                     long rows_per_blk) {
  const long r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f;
  long r = r0;
  for (; r + 1 < r1; r += 2) { s0 += src[r * C + c]; s1 += src[(r + 1) * C + c]; }
  if (r < r1) s0 += src[r * C + c];
  partial[(long)blockIdx.x * C + c] = s0 + s1;
}
// stage 2: out[c] = sum_k partial[k][c]; block = 16 columns x G partial groups (G = blockDim.x / 16: 16, or 64 when there are
// hundreds of partial rows -- the fold is a chain of dependent loads: with 16 groups, 1024 rows of bias-gradient partials took 20 us
// on 4-16 workgroups), folded in a fixed order
__global__ void __launch_bounds__(1024)
k_reduce_partials(const float* __restrict__ partial, int nsplit, int n, float* __restrict__ out) {
  __shared__ float red[1024];
  const int G = blockDim.x >> 4;
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    int k = g;
    // four independent partial sums per thread: four loads in flight (with two, folding 768 slabs took 27 us of latency)
    for (; k + 3 * G < nsplit; k += 4 * G) {
      const float a0 = partial[(long)k * n + c], a1 = partial[(long)(k + G) * n + c];
      const float a2 = partial[(long)(k + 2 * G) * n + c], a3 = partial[(long)(k + 3 * G) * n + c];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; k < nsplit; k += G) s0 += partial[(long)k * n + c];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < n) {
    float s = red[threadIdx.x];
    for (int j = 1; j < G; ++j) s += red[j * 16 + threadIdx.x];
    out[c] = s;
  }
}
static inline int rd_reduce_threads(int nsplit) { return nsplit >= 192 ? 1024 : 256; }

// The same fold for the column sums (n % 4 == 0, nsplit <= 1024), shaped for the gaps beside a running GEMM: n / 4 workgroups of
// 256 threads (k_reduce_partials folds 1024 x 64 partials on FOUR workgroups of 1024 threads, each of which needs sixteen free
// wave slots on one CU while the GEMMs own the CUs: 20-24 us on the side stream).  Thread g adds rows g, g + 256, g + 512, g + 768
// of its workgroup's four columns -- four independent 16-byte loads, one round trip -- then a fixed-order tree over the 256
// threads: ((p[g] + p[g+256]) + (p[g+512] + p[g+768])), then strides 128 ... 1.  Deterministic, no atomics.
__global__ void __launch_bounds__(256)
k_reduce_partials4(const float* __restrict__ partial, int nsplit, int n, float* __restrict__ out) {
  __shared__ f32x4 red[256];
  const int c = blockIdx.x * 4, g = threadIdx.x;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 a0 = g < nsplit ? *(const f32x4*)(partial + (long)g * n + c) : z;
  const f32x4 a1 = g + 256 < nsplit ? *(const f32x4*)(partial + (long)(g + 256) * n + c) : z;
  const f32x4 a2 = g + 512 < nsplit ? *(const f32x4*)(partial + (long)(g + 512) * n + c) : z;
  const f32x4 a3 = g + 768 < nsplit ? *(const f32x4*)(partial + (long)(g + 768) * n + c) : z;
  red[g] = (a0 + a1) + (a2 + a3);
  __syncthreads();
#pragma unroll
  for (int st = 128; st >= 1; st >>= 1) {
    if (g < st) red[g] = red[g] + red[g + st];
    __syncthreads();
  }
  if (g == 0) *(f32x4*)(out + c) = red[0];
}

// col2im of the D1 input gradient restricted to the sample channel (channel 0):
// g0[b][pos] = sum over taps t with (pos - t) even and o = (pos - t)/2 inside D1's output of
// P[b][o][(t, ci=0)], where P[row][tap*2+ci] = u1[row][:] . W1[tap][ci][:]   (D1: stride 2, 'valid', T:286)
__global__ void k_d1_col2im(const float* __restrict__ P, float* __restrict__ g0, int B, int D, int H, int W, int Do,
                            int Ho, int Wo, int Cin, int ldp) {
  const long total = (long)B * D * H * W;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    long t = f;
    int w = (int)(t % W); t /= W;
    int h = (int)(t % H); t /= H;
    int d = (int)(t % D);
    long b = t / D;
    float s = 0.f;
    for (int td = (d & 1); td < 3; td += 2) {
      int od = (d - td) >> 1;
      if (d - td < 0 || od >= Do) continue;
      for (int th = (h & 1); th < 3; th += 2) {
        int oh = (h - th) >> 1;
        if (h - th < 0 || oh >= Ho) continue;
        for (int tw = (w & 1); tw < 3; tw += 2) {
          int ow = (w - tw) >> 1;
          if (w - tw < 0 || ow >= Wo) continue;
          long row = ((b * Do + od) * Ho + oh) * Wo + ow;
          s += P[row * ldp + ((td * 3 + th) * 3 + tw) * Cin];
        }
      }
    }
    g0[f] = s;
  }
}

// X2 (T:238-241): per-sample n = ||g0||_2 ; gp = n - 1.  Then r0 = d(10*mean(gp^2))/dg0 =
// (10/B) * 2 (n-1)/n * g0, written as the 2-channel (r0, 0) input of the second forward sweep into the
// interpolated third of the critic-input buffer.  S blocks per sample (grid B * S): with S == 1 the block folds the sample's
// sum of squares itself; with S > 1 (few samples of many elements: ndomain 64 at 64 samples was 64 workgroups walking 98 304
// elements each, 231 us) k_gp_norm_part has left S partial sums per sample, added here in a fixed order.
__global__ void k_gp_norm_part(const float* __restrict__ g0, float* __restrict__ part, int per, int S) {
  __shared__ float red[4];
  const long b = blockIdx.x / S;
  const int c = blockIdx.x - (int)b * S;
  const int len = (per + S - 1) / S, i0 = c * len, i1 = min(per, i0 + len);
  float s = 0.f;
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) { float v = g0[b * per + i]; s += v * v; }
  s = rd_block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void k_gp_norm_r0(const float* __restrict__ g0, float* __restrict__ cin_hat, float* __restrict__ gp_out,
                             int per, int B, float gp_weight, int CP, int S, const float* __restrict__ part) {
  __shared__ float red[4];
  __shared__ float coef_s;
  const long b = blockIdx.x / S;
  const int c = blockIdx.x - (int)b * S;
  const int len = (per + S - 1) / S, i0 = c * len, i1 = min(per, i0 + len);
  float s = 0.f;
  if (S == 1) {
    for (int i = threadIdx.x; i < per; i += blockDim.x) { float v = g0[b * per + i]; s += v * v; }
    s = rd_block_sum(s, red);
  } else if (threadIdx.x == 0) {
    for (int k = 0; k < S; ++k) s += part[b * S + k];
  }
  if (threadIdx.x == 0) {
    float n = sqrtf(s);
    if (c == 0) gp_out[b] = n - 1.0f;
    coef_s = (gp_weight / B) * 2.0f * (n - 1.0f) / n;
  }
  __syncthreads();
  const float coef = coef_s;
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    float* o = cin_hat + (long)CP * (b * per + i);
    o[0] = coef * g0[b * per + i];
    for (int k = 1; k < CP; ++k) o[k] = 0.f;
  }
}

// X3 (T:215-216, T:388-392): losses of the critic step as Keras reports them:
// out[0] = total, out[1] = mean(-v_real), out[2] = mean(v_fake), out[3] = mean(gp^2); out[4] = non-finite flag
// (the flag also carries the generator's check_numerics, T:349-350: *gflag != 0 when its softmax produced NaN/Inf)
// (out[5..7] and *zero1 -- the gradient of the critic's last bias, which the loss does not reach -- are cleared here rather
// than by memsets of their own)
__global__ void k_critic_losses(const float* __restrict__ v, const float* __restrict__ gp, float* __restrict__ out,
                                int B, float gp_weight, const int* __restrict__ gflag, float* __restrict__ zero1) {
  __shared__ float red[4];
  float a = 0.f, f = 0.f, g = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) { a -= v[i]; f += v[B + i]; g += gp[i] * gp[i]; }
  a = rd_block_sum(a, red); f = rd_block_sum(f, red); g = rd_block_sum(g, red);
  if (threadIdx.x == 0) {
    a /= B; f /= B; g /= B;
    float tot = a + f + gp_weight * g;
    out[0] = tot; out[1] = a; out[2] = f; out[3] = g;
    out[4] = (fabsf(tot) <= 3.0e38f && *gflag == 0) ? 0.f : 1.f;
    out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
    if (zero1) *zero1 = 0.f;
  }
}
// generator step loss (T:408): out[0] = mean(-v); out[4] = non-finite flag
__global__ void k_gen_loss(const float* __restrict__ v, float* __restrict__ out, int B, const int* __restrict__ gflag) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) a -= v[i];
  a = rd_block_sum(a, red);
  if (threadIdx.x == 0) {
    a /= B;
    out[0] = a; out[1] = 0.f; out[2] = 0.f; out[3] = 0.f;
    out[4] = (fabsf(a) <= 3.0e38f && *gflag == 0) ? 0.f : 1.f;
    out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
  }
}

// X4 (T:385): Keras Adam with beta_1 = 0 (m = g, no m slab): v = b2 v + (1-b2) g^2 ;
// p -= lr_t * g / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t) supplied by the host; g is first scaled by
// grad_scale (1/world after the RCCL sum).
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v, long n, float lr_t,
                       float beta2, float eps, float grad_scale) {
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 gv = *(const f32x4*)(g + 4 * i) * grad_scale;
    f32x4 vv = *(const f32x4*)(v + 4 * i);
    f32x4 pv = *(const f32x4*)(p + 4 * i);
    vv = beta2 * vv + (1.0f - beta2) * gv * gv;
    pv.x -= lr_t * gv.x / (sqrtf(vv.x) + eps); pv.y -= lr_t * gv.y / (sqrtf(vv.y) + eps);
    pv.z -= lr_t * gv.z / (sqrtf(vv.z) + eps); pv.w -= lr_t * gv.w / (sqrtf(vv.w) + eps);
    *(f32x4*)(v + 4 * i) = vv;
    *(f32x4*)(p + 4 * i) = pv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long i = (n4 << 2) + threadIdx.x;
    float gv = g[i] * grad_scale;
    float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
    v[i] = vv;
    p[i] -= lr_t * gv / (sqrtf(vv) + eps);
  }
}

// out[t][c][r] (row stride ldo, zero padded) = in[t][r][c]; 32x32 tiles through LDS
__global__ void k_transpose(const float* __restrict__ in, float* __restrict__ out, int R, int C, int ldo) {
  __shared__ float tile[32][33];
  const long t = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? in[(t * R + r) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, r = r0 + tx;
    if (c < C && r < ldo) out[(t * C + c) * ldo + r] = tile[tx][i];
  }
}

// ------------------------------------------------------------------------------------
// Upsample collapse (exact algebra): UpSampling3D(2) followed by a 3-tap 'same'
// conv equals, per output parity p and per axis, a 2-tap conv on the un-upsampled grid with taps
// p=0: (W0 | W1+W2) at offsets (-1, 0);  p=1: (W0+W1 | W2) at offsets (0, +1).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void rd_collapse_range(int p, int a, int& lo, int& hi) {
  lo = a == 0 ? 0 : (p == 0 ? 1 : 2);
  hi = a == 1 ? 2 : (p == 0 ? 0 : 1);
}
// Wc[phase(pd,ph,pw)*8 + tap(ad,ah,aw)][ci][co] = sum of the original taps that fall on the same source voxel
__global__ void k_collapse_weights(const float* __restrict__ W, float* __restrict__ Wc, int CC /* Cin*Cout */) {
  const int c4s = CC / 4;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < 64L * c4s; f += (long)gridDim.x * blockDim.x) {
    int pa = (int)(f / c4s);
    int e = (int)(f - (long)pa * c4s) * 4;
    int ph = pa >> 3, tp = pa & 7;
    int lo[3], hi[3];
    rd_collapse_range(ph >> 2, tp >> 2, lo[0], hi[0]);
    rd_collapse_range((ph >> 1) & 1, (tp >> 1) & 1, lo[1], hi[1]);
    rd_collapse_range(ph & 1, tp & 1, lo[2], hi[2]);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int td = lo[0]; td <= hi[0]; ++td)
      for (int th = lo[1]; th <= hi[1]; ++th)
        for (int tw = lo[2]; tw <= hi[2]; ++tw) s += *(const f32x4*)(W + (long)((td * 3 + th) * 3 + tw) * CC + e);
    *(f32x4*)(Wc + (long)pa * CC + e) = s;
  }
}
// adjoint of the above: dW[t] = sum over the 8 collapsed entries (p,a) whose range contains t on every axis
__global__ void k_fold_collapsed_wgrad(const float* __restrict__ dWc, float* __restrict__ dW, int CC) {
  const int c4s = CC / 4;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < 27L * c4s; f += (long)gridDim.x * blockDim.x) {
    int t = (int)(f / c4s);
    int e = (int)(f - (long)t * c4s) * 4;
    int tt[3] = {t / 9, (t / 3) % 3, t % 3};
    // per axis the two (p,a) pairs containing tap t: t=0: (0,0),(1,0); t=1: (0,1),(1,0); t=2: (0,1),(1,1)
    int pp[3][2], aa[3][2];
    for (int x = 0; x < 3; ++x) {
      pp[x][0] = 0; aa[x][0] = tt[x] == 0 ? 0 : 1;
      pp[x][1] = 1; aa[x][1] = tt[x] == 2 ? 1 : 0;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) {
          int ph = pp[0][i] * 4 + pp[1][j] * 2 + pp[2][k];
          int tp = aa[0][i] * 4 + aa[1][j] * 2 + aa[2][k];
          s += *(const f32x4*)(dWc + (long)(ph * 8 + tp) * CC + e);
        }
    *(f32x4*)(dW + (long)t * CC + e) = s;
  }
}
// out[q][c][r] = in[map[q]][r][c] for q < 64 (per-slice transpose through LDS); builds the collapsed
// input-gradient weights Wd[q][Cout][Cin] from Wc
struct RdSliceMap { int16_t src[64]; };
__global__ void k_transpose_map(const float* __restrict__ in, float* __restrict__ out, int R, int C, RdSliceMap map) {
  __shared__ float tile[32][33];
  const long q = blockIdx.z, t = map.src[blockIdx.z];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? in[(t * R + r) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) out[(q * C + c) * R + r] = tile[tx][i];
  }
}
// out = g * LeakyReLU'(h) (from the stored output h)
template <typename T = float>
__global__ void k_lrelu_bwd(const T* __restrict__ g, const T* __restrict__ h, float* __restrict__ out, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 gv = rd_ld4(g + 4 * i), hv = rd_ld4(h + 4 * i);
    gv.x *= rd_lrelu_slope_from_out(hv.x); gv.y *= rd_lrelu_slope_from_out(hv.y);
    gv.z *= rd_lrelu_slope_from_out(hv.z); gv.w *= rd_lrelu_slope_from_out(hv.w);
    *(f32x4*)(out + 4 * i) = gv;
  }
}

// first critic layer with CP > Cin: zero rows for the padding channels (forward copy) and their removal (gradient)
__global__ void k_pad_w1(const float* __restrict__ w, float* __restrict__ wp, int Cin, int CP) {
  const int total = 27 * CP * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int co = i % 64, c = (i / 64) % CP, t = i / (64 * CP);
    wp[i] = c < Cin ? w[(t * Cin + c) * 64 + co] : 0.f;
  }
}
__global__ void k_unpad_w1(const float* __restrict__ wp, float* __restrict__ w, int Cin, int CP) {
  const int total = 27 * Cin * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int co = i % 64, c = (i / 64) % Cin, t = i / (64 * Cin);
    w[i] = wp[(t * CP + c) * 64 + co];
  }
}

// ------------------------------------------------------------------------------------
// Shared-centre form of UpSampling3D(2)+Conv3D(3^3,'same') along the hour axis d.
// Per axis the two outputs of a source position are  out[2s] = W0 x[s-1] + (W1+W2) x[s],  out[2s+1] = (W0+W1) x[s] +
// W2 x[s+1].  With E[j] = x[j] - x[j-1] (x zero-extended, j = 0..D) and S = W0+W1+W2 this is
//   out[2s] = S x[s] - W0 E[s],   out[2s+1] = S x[s] + W2 E[s+1]:
// the S x[s] product is shared by both outputs, so the weight and input gradients need 3 instead of 4 tap products per
// source position on that axis (48 instead of 64 per position overall; the other two axes keep the collapsed form).
// U[u] = sum_k c[u][k] W[k] with u = (group A',S,D) x (ph,th) x (pw,tw) and c in {-1,0,1}.
// ------------------------------------------------------------------------------------
// (struct RdWeightMap: rdgan_plan.h)

__global__ void k_weight_transform(const float* __restrict__ W, float* __restrict__ U, int CC, int nu, RdWeightMap T) {
  const int c4s = CC / 4;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < (long)nu * c4s; f += (long)gridDim.x * blockDim.x) {
    const int u = (int)(f / c4s), e = (int)(f - (long)u * c4s) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 27; ++k) {
      const int c = T.c[u][k];
      if (c) { const f32x4 w = *(const f32x4*)(W + (long)k * CC + e); s += c > 0 ? w : -w; }
    }
    *(f32x4*)(U + (long)u * CC + e) = s;
  }
}
// adjoint: dW[k] = sum_u c[u][k] dU[u]
__global__ void k_weight_transform_adj(const float* __restrict__ dU, float* __restrict__ dW, int CC, int nu, RdWeightMap T) {
  const int c4s = CC / 4;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < 27L * c4s; f += (long)gridDim.x * blockDim.x) {
    const int k = (int)(f / c4s), e = (int)(f - (long)k * c4s) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < nu; ++u) {
      const int c = T.c[u][k];
      if (c) { const f32x4 w = *(const f32x4*)(dU + (long)u * CC + e); s += c > 0 ? w : -w; }
    }
    *(f32x4*)(dW + (long)k * CC + e) = s;
  }
}
// E[b][j][:] = x[b][j][:] - x[b][j-1][:], j = 0..D, x zero outside [0,D); P = floats per d-plane (multiple of 4)
template <typename T = float>
__global__ void k_diff_d(const T* __restrict__ x, T* __restrict__ E, int B, int D, long P) {
  const long p4 = P / 4, total = (long)B * (D + 1) * p4;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const long i = f % p4; const long bj = f / p4;
    const int j = (int)(bj % (D + 1)); const long b = bj / (D + 1);
    const T* xb = x + (b * D) * P + i * 4;
    f32x4 hi = {0.f, 0.f, 0.f, 0.f}, lo = hi;
    if (j < D) hi = rd_ld4(xb + (long)j * P);
    if (j > 0) lo = rd_ld4(xb + (long)(j - 1) * P);
    rd_st4(E + f * 4, hi - lo);
  }
}
// dx[b][d][:] += dE[b][d][:] - dE[b][d+1][:]   (adjoint of k_diff_d added to the shared-centre part already in dx)
template <typename T = float>
__global__ void k_combine_dx(T* __restrict__ dx, const T* __restrict__ dE, int B, int D, long P) {
  const long p4 = P / 4, total = (long)B * D * p4;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const long i = f % p4; const long bd = f / p4;
    const int d = (int)(bd % D); const long b = bd / D;
    const T* e = dE + ((b * (D + 1) + d) * P) + i * 4;
    f32x4 v = rd_ld4(dx + f * 4);
    v += rd_ld4(e) - rd_ld4(e + P);
    rd_st4(dx + f * 4, v);
  }
}

// ------------------------------------------------------------------------------------
// bf16 operand copies for the bf16-MFMA conv GEMM (fp32 -> bf16, round to nearest even; v_cvt_pk_bf16_f32)
// ------------------------------------------------------------------------------------
// n % 8 == 0
__global__ void k_to_bf16(const float* __restrict__ in, unsigned short* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n / 8; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(in + 8 * i), b = *(const f32x4*)(in + 8 * i + 4);
    u32x4_t o = {rd_pack_bf16(a.x, a.y), rd_pack_bf16(a.z, a.w), rd_pack_bf16(b.x, b.y), rd_pack_bf16(b.z, b.w)};
    *(u32x4_t*)(out + 8 * i) = o;
  }
}
// out[t][n][k] = bf16(in[t][k][n]): the weight layout of the bf16 conv GEMM (K contiguous); grid (N/32, K/32, T).  outf (may be
// null): the same tap blocks in fragment order
__global__ void k_weights_to_bf16_t(const float* __restrict__ in, unsigned short* __restrict__ out, int K, int N,
                                    unsigned short* __restrict__ outf = nullptr) {
  __shared__ float tile[32][33];
  const long t = blockIdx.z;
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, n = n0 + tx;
    tile[i][tx] = (k < K && n < N) ? in[(t * K + k) * N + n] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, k = k0 + tx;
    if (n < N && k < K) {
      const __bf16 v = (__bf16)tile[tx][i];
      out[(t * N + n) * K + k] = __builtin_bit_cast(unsigned short, v);
      if (outf) outf[t * N * K + rd_wfrag_index(n, k, K)] = __builtin_bit_cast(unsigned short, v);
    }
  }
}
// out[n][k] = bf16(in[k][n]) for k < K, 0 for K <= k < KP: the Dense kernel [K][N] as the [N][KP] operand of the bf16 GEMM, K padded
// to the GEMM's chunk of 64; grid (N/32, KP/32)
__global__ void k_dense_w16(const float* __restrict__ in, unsigned short* __restrict__ out, int K, int N, int KP) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, n = n0 + tx;
    tile[i][tx] = (k < K && n < N) ? in[(long)k * N + n] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, k = k0 + tx;
    if (n < N && k < KP) out[(long)n * KP + k] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][i]);
  }
}
// The critic's layers 2-4 in one launch: per layer l (blockIdx.z / 27) and tap (blockIdx.z % 27) the transposed bf16 image
// outT[tap][n][k] (forward GEMMs) AND the plain bf16 copy outC[tap][k][n] (input-gradient GEMMs) of in[tap][k][n]; 32 x 32 tiles
// through LDS, grid (8, 8, 81) covers K, N <= 256 in tile-strided loops
// outTf / outCf (may be null): the two images again in fragment order
struct RdW3 { const float* in[3]; unsigned short* outT[3]; unsigned short* outC[3]; unsigned short* outTf[3]; unsigned short* outCf[3]; int K[3], N[3]; };
__global__ void k_weights3_to_bf16(RdW3 a) {
  __shared__ float tile[32][33];
  const int l = blockIdx.z / 27;
  const long t = blockIdx.z % 27;
  const int K = a.K[l], N = a.N[l];
  const float* in = a.in[l];
  unsigned short* outT = a.outT[l];
  unsigned short* outC = a.outC[l];
  unsigned short* outTf = a.outTf[l];
  unsigned short* outCf = a.outCf[l];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k0 = blockIdx.y * 32; k0 < K; k0 += gridDim.y * 32)
    for (int n0 = blockIdx.x * 32; n0 < N; n0 += gridDim.x * 32) {
      __syncthreads();
      for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, n = n0 + tx;
        const float v = (k < K && n < N) ? in[(t * K + k) * N + n] : 0.f;
        tile[i][tx] = v;
        if (k < K && n < N) {
          outC[(t * K + k) * N + n] = __builtin_bit_cast(unsigned short, (__bf16)v);
          if (outCf) outCf[t * K * N + rd_wfrag_index(k, n, N)] = __builtin_bit_cast(unsigned short, (__bf16)v);     // ([N' = K][K' = N])
        }
      }
      __syncthreads();
      for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, k = k0 + tx;
        if (n < N && k < K) {
          outT[(t * N + n) * K + k] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][i]);
          if (outTf) outTf[t * N * K + rd_wfrag_index(n, k, K)] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][i]);
        }
      }
    }
}
// out[q][:] = bf16(in[map[q]][:]) for q < gridDim.y, cc floats per block (cc % 8 == 0): the input-gradient weight forms of the
// bf16 conv GEMM are the forward forms U themselves ([Cin][Cout] = [N][K] of that GEMM), re-ordered by tap
// outf (may be null; K = the blocks' row length, K % 16 == 0): the same blocks in fragment order
__global__ void k_blocks_to_bf16(const float* __restrict__ in, unsigned short* __restrict__ out, long cc, RdSliceMap map,
                                 unsigned short* __restrict__ outf = nullptr, int K = 0) {
  const long q = blockIdx.y, t = map.src[blockIdx.y];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < cc / 8; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(in + t * cc + 8 * i), b = *(const f32x4*)(in + t * cc + 8 * i + 4);
    u32x4_t o = {rd_pack_bf16(a.x, a.y), rd_pack_bf16(a.z, a.w), rd_pack_bf16(b.x, b.y), rd_pack_bf16(b.z, b.w)};
    *(u32x4_t*)(out + q * cc + 8 * i) = o;
    if (outf) {
      const int n = (int)(8 * i / K), k = (int)(8 * i - (long)n * K);
      *(u32x4_t*)(outf + q * cc + rd_wfrag_index(n, k, K)) = o;
    }
  }
}

// bf16 -> fp32 (test hook rdgan_debug_activation in the bf16 storage mode)
__global__ void k_bf16_to_f32(const rd_bf16_t* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = rd_ld1(in + i);
}
// first critic kernel W1 [27*Cin][64] -> bf16 [ldp][64], rows 27*Cin.. zero: the [N][K] weight image of the column GEMM of
// the critic's input gradient (P1[row][(tap,ci)] = u1[row][:] . W1[(tap,ci)][:]) in the bf16 storage mode
__global__ void k_w1_to_bf16(const float* __restrict__ w, rd_bf16_t* __restrict__ out, int rows, int ldp) {
  const int total = ldp * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
    rd_st1(out + i, i / 64 < rows ? w[i] : 0.f);
}
