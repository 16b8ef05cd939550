// Input pipeline either side of the training step, HBM-bound integer/float work:
//  * k_gather_tiles: what generate_real_samples / generate_latent_points do on the host in the reference
//    (gan_train_cwgangp_pixelnorm.py:149-166, :181-190): gather (24, nd, nd) windows of the (n_days, 24, ny, nx)
//    radar array at (tidx, yidx, xidx), daily sum = condition, tile / daily sum = hourly fractions, condition / 127.4.
//  * k_valid_tiles: compute_valid_indices.py:74-92: a box is valid if its daily sum has no NaN and at least n_thresh
//    points above tp_thresh_daily.
// Results are bit-identical to the numpy restatement (same fp32 operation order, correctly rounded division).
#pragma once
#include <hip/hip_runtime.h>

// one thread per (sample, y, x) pixel of the tile; coalesced along x.  flags[0] |= 1 if any output is non-finite
// (the reference asserts ~isnan, T:169-170), flags[0] |= 2 if a fraction is outside [0,1] (T:171-172).
__global__ void k_gather_tiles(const float* __restrict__ data, int n_days, int nh, int ny, int nx,
                               const int* __restrict__ idx, int n, int nd, float norm_scale, float* __restrict__ batch,
                               float* __restrict__ cond, int* __restrict__ flags) {
  const long total = (long)n * nd * nd;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const int j = (int)(f % nd), i = (int)((f / nd) % nd);
    const long s = f / ((long)nd * nd);
    const int t = idx[3 * s], y = idx[3 * s + 1], x = idx[3 * s + 2];
    const float* p = data + (((long)t * nh) * ny + (y + i)) * nx + (x + j);
    const long hstride = (long)ny * nx;
    float sum = 0.f;
    for (int h = 0; h < nh; ++h) sum += p[h * hstride];          // np.sum(batch, axis=1): sequential over the hours
    int bad = 0;
    if (batch) {
      for (int h = 0; h < nh; ++h) {
        float v = p[h * hstride] / sum;                             // batch[i] / batch_cond[i]
        if (!(fabsf(v) <= 3.0e38f)) bad |= 1;
        else if (v > 1.f || v < 0.f) bad |= 2;
        batch[((s * nh + h) * nd + i) * nd + j] = v;
      }
    }
    const float c = sum / norm_scale;
    if (!(fabsf(c) <= 3.0e38f)) bad |= 1;
    cond[f] = c;
    if (bad) atomicOr(flags, bad);
  }
}

// one 256-thread block per (day, box row ii, box column jj) of the stride grid: valid[...] = 1/0
__global__ void k_valid_tiles(const float* __restrict__ data, int nh, int ny, int nx, int nd, int stride, int nbi, int nbj,
                              float thresh, int n_thresh, int* __restrict__ valid) {
  __shared__ int s_nan, s_cnt;
  const int bj = blockIdx.x % nbj, bi = (blockIdx.x / nbj) % nbi;
  const long t = blockIdx.x / ((long)nbj * nbi);
  if (threadIdx.x == 0) { s_nan = 0; s_cnt = 0; }
  __syncthreads();
  int nan = 0, cnt = 0;
  const long hstride = (long)ny * nx;
  for (int pix = threadIdx.x; pix < nd * nd; pix += blockDim.x) {
    const int i = pix / nd, j = pix % nd;
    const float* p = data + ((t * nh) * ny + (bi * stride + i)) * (long)nx + (bj * stride + j);
    float sum = 0.f;
    for (int h = 0; h < nh; ++h) sum += p[h * hstride];
    if (sum != sum) nan = 1;
    if (sum > thresh) cnt += 1;
  }
  if (nan) atomicOr(&s_nan, 1);
  if (cnt) atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (threadIdx.x == 0) valid[blockIdx.x] = (!s_nan && s_cnt >= n_thresh) ? 1 : 0;
}

// Ensemble CRPS per grid point (generate_and_evaluate_crps.py:188, properscoring.crps_ensemble(obs, ens, axis=0)):
//   crps = mean_i |x_i - y| - 0.5 * mean_{i,j} |x_i - x_j|
// evaluated with the sorted-ensemble identity sum_{i<j} (x_(j) - x_(i)) = sum_i (2i - n - 1) x_(i), i = 1..n.
// One 256-thread block per grid point: the n members (stride npix floats) are sorted by a bitonic network in LDS
// (padded to the next power of two with +inf), sums in fp64-free Kahan-free fp32 pairwise block reduction.
// `scale` (nullable, per point) multiplies the members first -- fractions -> mm/h by cond*norm_scale (:186).
__global__ void __launch_bounds__(256)
k_crps_ensemble(const float* __restrict__ ens, const float* __restrict__ obs, const float* __restrict__ scale,
                float* __restrict__ crps, int n, int npow2, long npix) {
  extern __shared__ float xs[];
  __shared__ float red[8];
  const long p = blockIdx.x;
  const float sc = scale ? scale[p] : 1.0f;
  for (int i = threadIdx.x; i < npow2; i += 256) xs[i] = i < n ? ens[(long)i * npix + p] * sc : __builtin_inff();
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += 256) {
        int l = i ^ j;
        if (l > i) {
          float a = xs[i], b = xs[l];
          bool up = (i & k) == 0;
          if ((a > b) == up) { xs[i] = b; xs[l] = a; }
        }
      }
      __syncthreads();
    }
  const float y = obs[p];
  float s_abs = 0.f, s_spread = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float x = xs[i];
    s_abs += fabsf(x - y);
    s_spread += (float)(2 * i + 1 - n) * x;          // (2(i+1) - n - 1) x_(i+1)
  }
  s_abs = rd_block_sum(s_abs, red);
  s_spread = rd_block_sum(s_spread, red + 4);
  if (threadIdx.x == 0) crps[p] = s_abs / n - s_spread / ((float)n * (float)n);
}
