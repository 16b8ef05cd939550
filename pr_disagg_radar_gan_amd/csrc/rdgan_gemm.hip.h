// fp32 MFMA implicit-GEMM kernels for gfx950 (CDNA4): the conv-like contractions of the
// cWGAN-GP step (gan_train_cwgangp_pixelnorm.py:286-299 critic Conv3D, :326-345 generator
// Dense / UpSampling3D+Conv3D), their input gradients and their weight gradients.
//
// v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD): lane l supplies A[i=l&31][k=l>>5]
// and B[k=l>>5][j=l&31]; D register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
// 256-thread workgroups = 4 waves (one per SIMD); operands staged global -> registers -> LDS
// (double buffered, one barrier per K chunk).  A is gathered on the fly from NDHWC
// activations (im2col never materialised).  The gather costs 4 VALU instructions per row and
// K chunk: the per-row source offset and a 12-bit validity mask come from a host-built row
// table (RdRow), the per-tap offset and mask are wave-uniform scalars from the plan, and
// out-of-image taps are turned into an out-of-range buffer offset so the hardware bounds
// check of buffer_load returns the zero padding.
#pragma once
#include <hip/hip_runtime.h>
#include "rdgan_plan.h"
#include "rdgan_rng.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RD_LRELU_ALPHA 0.2f
#define RD_OOB 0x80000000u            // >= num_records of every descriptor below -> load returns 0
#define RD_RSRC_BYTES 0x7FFFFFF0u

__device__ __forceinline__ float rd_lrelu(float x) { return x > 0.f ? x : RD_LRELU_ALPHA * x; }
// slope of LeakyReLU recovered from its (possibly dropout-scaled) output: TF's LeakyReluGrad
// uses features > 0 ? 1 : alpha, and sign(output) == sign(features) for kept elements.
__device__ __forceinline__ float rd_lrelu_slope_from_out(float h) { return h > 0.f ? 1.f : RD_LRELU_ALPHA; }

__device__ __forceinline__ f32x4 rd_buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rd_make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, RD_RSRC_BYTES, 0x00020000);
}
// s_shift == 1 (direct form of the folded upsample): per-row element delta of a tap from the 2-bit codes
__device__ __forceinline__ int rd_shift_delta(int w, int sd, int sh_, int sw, int SH, int SW, int cstride) {
  int dd = ((w >> sd) & 3) - 1, dh = ((w >> sh_) & 3) - 1, dw = ((w >> sw) & 3) - 1;
  return ((dd * SH + dh) * SW + dw) * cstride;
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2).  This bijective remap
// gives every XCD a contiguous range of the logical tile order, so tiles that share operand rows (halo
// rows of neighbouring M tiles, the tap tiles of one wgrad split) hit the same L2.  Placement only
// affects speed, never correctness.
__device__ __forceinline__ int rd_xcd_swizzle(int lin, int total) {
  const int xcd = lin & 7, idx = lin >> 3;
  const int q = total >> 3, r = total & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ------------------------------------------------------------------------------------
// C[m][n] = sum_{tap,c} A_gather[m][tap][c] * W[tap_w*wrpt + c][n]  (+ fused epilogue)
// ------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK, bool PARTIAL, bool SHIFT>
__global__ void __launch_bounds__(256)
k_conv_gemm(const RdPlan* __restrict__ plan, int B, const float* __restrict__ src,
            const float* __restrict__ W, int ldw, float* dst, RdEpi epi) {
  // PARTIAL: SC % 4 != 0 (the last float4 of a tap is masked element-wise); SHIFT: s_shift == 1 (direct
  // form of the folded nearest upsample).  Both are compile-time so the hot loop has no uniform branches.
  static_assert(WM * WN == 4, "4 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  // A image in LDS: BK = 32 -> unpadded 128-byte rows whose 16-byte chunk c sits at c ^ ((row >> 1) & 7), which
  // makes the b128 fragment reads of every 16-lane group conflict-free; BK = 8 -> rows padded to 12 dwords.
  constexpr bool SWZ = BK == 32;
  constexpr int AST = SWZ ? BK : BK + 4;
  constexpr int BST = BN;                     // B rows are read 32 consecutive floats at a time: no padding needed
  constexpr int A_F4 = BK / 4, A_RPP = 256 / A_F4, A_P = (BM + A_RPP - 1) / A_RPP;
  constexpr int B_F4 = BN / 4, B_RPP = 256 / B_F4, B_P = (BK + B_RPP - 1) / B_RPP;
  constexpr int STAGE = BM * AST + BK * BST;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhalf = lane >> 5;

  // ---- which phase / tile (all wave-uniform); 1-D grid, N tile fastest so both N tiles of an M tile run together
  const int NTn = plan->N / BN;
  const int swz = rd_xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = swz % NTn;
  int mt = swz / NTn, pidx = 0;
  if (plan->interleave) {
    pidx = mt % plan->nphases;
    mt /= plan->nphases;
  } else {
    for (int p = 0; p < plan->nphases; ++p) {
      int nt = (B * plan->ph[p].L + BM - 1) / BM;
      if (mt < nt) { pidx = p; break; }
      mt -= nt;
    }
  }
  const RdPhase& P = plan->ph[pidx];
  const int L = P.L;
  const int rows = B * L;
  const int m0 = mt * BM;
  const int n0 = ntile * BN;
  const int b0 = m0 / L, l0 = m0 - b0 * L;
  const int SH = plan->SH, SW = plan->SW;
  const int cstride = plan->s_cstride, SC = plan->SC, wrpt = plan->w_rows_per_tap;
  const int ssample = (int)plan->src_sample;
  const int ntaps = P.ntaps;
  const RdRow* __restrict__ tab = plan->tab + P.tab;
  const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc(src + (long)b0 * plan->src_sample);
  const __amdgpu_buffer_rsrc_t rsB = rd_make_rsrc(W + P.w_off);

  // ---- per-thread A rows: byte offset relative to sample b0 (incl. this thread's channel group), validity bits
  int roff[A_P], rbits[A_P], rcode[A_P];
  const int a_c4 = (tid % A_F4) * 4;
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    int r = tid / A_F4 + i * A_RPP;
    roff[i] = 0; rbits[i] = 0; rcode[i] = 0;
    if (r < BM && m0 + r < rows) {
      int l = l0 + r, bb = 0;
      if (L >= BM) { if (l >= L) { l -= L; bb = 1; } }
      else { bb = l / L; l -= bb * L; }
      RdRow e = tab[l];
      roff[i] = (bb * ssample + e.x + a_c4) * 4;
      rbits[i] = e.y; rcode[i] = e.w;
    }
  }
  // ---- per-thread B (weight) rows
  int boff[B_P];
  const int b_kk = tid / B_F4, b_n4 = (tid % B_F4) * 4;
#pragma unroll
  for (int i = 0; i < B_P; ++i) boff[i] = ((b_kk + i * B_RPP) * ldw + n0 + b_n4) * 4;

  const int CPT = (SC + BK - 1) / BK;
  // split-K: this workgroup multiplies chunks [q0, q0 + nchunks) of the ntaps*CPT chunks of the K loop
  const int nch_all = ntaps * CPT;
  const int ksplit = epi.ksplit > 1 ? epi.ksplit : 1;
  const int per_split = (nch_all + ksplit - 1) / ksplit;
  const int q0 = (int)blockIdx.y * per_split;
  const int nchunks = max(0, min(nch_all, q0 + per_split) - q0);

  // K order: channel chunk outer, tap inner (the taps of one chunk re-read the same 128-byte pixel segments of
  // neighbouring rows).  (ld_tap, ld_cc) is the chunk the NEXT load_chunk call fetches; its tap descriptor `ti`
  // was scalar-loaded one call earlier, so no load_chunk waits on a scalar load it has just issued.
  int ld_cc = q0 / ntaps, ld_tap = q0 - ld_cc * ntaps;
  RdTap ti = P.tap[ld_tap];

  // one register set: the gather of chunk q+1 is in flight while chunk q is multiplied (a second set, i.e. a
  // distance-2 prefetch, measured no gain: the per-chunk cost is issue/barrier structure, not load latency)
  f32x4 ra[A_P], rw[B_P];
  auto load_chunk = [&]() {
    const int tmask = ti.mask;
    const int c = ld_cc * BK + a_c4;
    const bool c_ok = c < SC;
    const int sdelta = ld_cc * BK * 4 + (SHIFT ? 0 : ti.delta);
    const int sd = ti.code & 255, sh_ = (ti.code >> 8) & 255, sw = ti.code >> 16;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      unsigned voff = (unsigned)(roff[i] + sdelta);
      if (SHIFT) voff += (unsigned)(rd_shift_delta(rcode[i], sd, sh_, sw, SH, SW, cstride) * 4);
      bool ok = c_ok && ((rbits[i] & tmask) == tmask);
#ifdef RD_ABL_L1
      voff = (unsigned)(roff[i] & 0x3FFF);     // diagnostic build: every gather hits a 16 KiB window
#endif
      f32x4 v = rd_buf_load4(rsA, ok ? voff : RD_OOB);
      if (PARTIAL) {
        if (c + 1 >= SC) v.y = 0.f;
        if (c + 2 >= SC) v.z = 0.f;
        if (c + 3 >= SC) v.w = 0.f;
      }
      ra[i] = v;
    }
    const int sB = (ti.w * wrpt + ld_cc * BK) * ldw * 4;
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      int kk = b_kk + i * B_RPP;
      bool ok = (BK % B_RPP == 0 || kk < BK) && ld_cc * BK + kk < SC;
      rw[i] = rd_buf_load4(rsB, ok ? (unsigned)(boff[i] + sB) : RD_OOB);
    }
    // advance to the next chunk and fetch its tap descriptor (consumed by the next call)
    if (++ld_tap == ntaps) { ld_tap = 0; ++ld_cc; }
    ti = P.tap[ld_tap];
  };
  // LDS write offset of this thread's A chunk; A_RPP is a multiple of 16, so the swizzle term is the same for all i
  const int a_wr = SWZ ? (((tid % A_F4) ^ (((tid / A_F4) >> 1) & 7)) * 4) : a_c4;
  auto store_chunk = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * AST;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      int r = tid / A_F4 + i * A_RPP;
      if (BM % A_RPP == 0 || r < BM) *(f32x4*)&As[r * AST + a_wr] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      int kk = b_kk + i * B_RPP;
      if (BK % B_RPP == 0 || kk < BK) *(f32x4*)&Bs[kk * BST + b_n4] = rw[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // swizzled fragment read offsets (floats) of this lane for the logical chunks 2*j8 + lhalf
  const int a_sw = (l31 >> 1) & 7;
  if (nchunks > 0) {
    load_chunk();
    store_chunk(0);
  }
  __syncthreads();
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
    if (q + 1 < nchunks) load_chunk();
    const float* As = smem + buf * STAGE + (wm * WTM + l31) * AST;
    const float* Bs = smem + buf * STAGE + BM * AST + lhalf * 4 * BST + wn * WTN + l31;
    // LDS fragments double-buffered in registers: the reads of k-group j8+1 are issued before the MFMAs of j8
    constexpr int NJ = BK / 8;
    f32x4 fa[2][TM];
    float fb[2][4][TN];
    auto load_frag = [&](int slot, int j8) {
      const int acol = SWZ ? (((j8 * 2 + lhalf) ^ a_sw) * 4) : (j8 * 8 + lhalf * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[slot][i] = *(const f32x4*)&As[i * 32 * AST + acol];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][s][j] = Bs[(j8 * 8 + s) * BST + j * 32];
    };
    load_frag(0, 0);
#pragma unroll
    for (int j8 = 0; j8 < NJ; ++j8) {
      const int cur = j8 & 1;
      if (j8 + 1 < NJ) load_frag(cur ^ 1, j8 + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][s], fb[cur][s][j], acc[i][j], 0, 0, 0);
    }
    if (q + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue
  const long dsample = plan->dst_sample;
  const int mode = epi.mode;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      if (m0 + row < rows) {
        int l = l0 + row, bb = b0;
        if (L >= BM) { if (l >= L) { l -= L; bb += 1; } }
        else { int qd = l / L; l -= qd * L; bb += qd; }
        const long rowbase = (long)bb * dsample + tab[l].z;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * WTN + j * 32 + l31;
          const long idx = rowbase + col;
          float v = acc[i][j][r];
          if (ksplit > 1) {
            epi.kpart[(long)blockIdx.y * epi.kstride + idx] = v;
            continue;
          }
          if (mode == RD_EPI_BIAS) {
            v += epi.bias[col];
          } else if (mode == RD_EPI_BIAS_LRELU) {
            v = rd_lrelu(v + epi.bias[col]);
          } else if (mode == RD_EPI_BIAS_LRELU_DROP) {
            v = rd_lrelu(v + epi.bias[col]);
            if (epi.use_drop) v *= rd_drop_scale(epi.key, (uint32_t)idx + epi.idx_base);
          } else if (mode == RD_EPI_GATE_AUX) {
            float g = rd_lrelu_slope_from_out(epi.aux[idx]);
            if (epi.use_drop) g *= rd_drop_scale(epi.key, (uint32_t)idx + epi.idx_base);
            v *= g;
          }
          dst[idx] = v;
        }
      }
    }
  }
}

// split-K finish: dst[idx] = epilogue(sum_s kpart[s][idx]); the destination is dense with N floats per pixel
__global__ void k_splitk_finish(float* dst, long total, int N, RdEpi epi) {
  const int mode = epi.mode;
  for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < total / 4; i4 += (long)gridDim.x * blockDim.x) {
    const long idx0 = i4 * 4;
    f32x4 v = *(const f32x4*)(epi.kpart + idx0);
    for (int s = 1; s < epi.ksplit; ++s) v += *(const f32x4*)(epi.kpart + s * epi.kstride + idx0);
    const int col0 = (int)(idx0 % N);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long idx = idx0 + e;
      float x = v[e];
      if (mode == RD_EPI_BIAS) {
        x += epi.bias[col0 + e];
      } else if (mode == RD_EPI_BIAS_LRELU) {
        x = rd_lrelu(x + epi.bias[col0 + e]);
      } else if (mode == RD_EPI_BIAS_LRELU_DROP) {
        x = rd_lrelu(x + epi.bias[col0 + e]);
        if (epi.use_drop) x *= rd_drop_scale(epi.key, (uint32_t)idx + epi.idx_base);
      } else if (mode == RD_EPI_GATE_AUX) {
        float g = rd_lrelu_slope_from_out(epi.aux[idx]);
        if (epi.use_drop) g *= rd_drop_scale(epi.key, (uint32_t)idx + epi.idx_base);
        x *= g;
      }
      v[e] = x;
    }
    *(f32x4*)(dst + idx0) = v;
  }
}

// ------------------------------------------------------------------------------------
// weight gradient: dW[tap_w*wrpt + c][n] = sum_m A_gather[m][tap][c] * dY[m][n]
// grid.x = r_tile * NT + n_tile, grid.y = split over rows m, grid.z = plan phase; partial sums go to
// `partial[phase][split][RT*BR][N]` and are folded by k_wgrad_reduce (deterministic, no atomics).
// ------------------------------------------------------------------------------------
struct RdWgradTiling {
  int RT, NT;            // tiles over (tap,c) rows and over N
  int tiles_per_tap;     // SC >= BR: ceil(SC/BR); else 0
  int cw;                // c extent per tap inside a tile (BR, or padded SC < BR)
  int taps_per_tile;     // 1 or BR/cw
  int rows_per_split;    // multiple of 32
  int nsplit, nphases;   // grid = RT*NT * nsplit * nphases workgroups (1-D)
};

__device__ __forceinline__ void rd_wgrad_tile_row(const RdWgradTiling& T, int BR, int rt, int r, int& tap, int& c) {
  if (T.tiles_per_tap > 0) {
    tap = rt / T.tiles_per_tap;
    c = (rt - tap * T.tiles_per_tap) * BR + r;
  } else {
    int tl = r / T.cw;
    tap = rt * T.taps_per_tile + tl;
    c = r - tl * T.cw;
  }
}

template <int BR, int BN, bool PARTIAL, bool SHIFT>
__global__ void __launch_bounds__(256)
k_wgrad_gemm(const RdPlan* __restrict__ plan, int B, const float* __restrict__ src,
             const float* __restrict__ dy, float* __restrict__ partial, RdWgradTiling T) {
  constexpr int BKP = 32;
  constexpr int WTM = BR / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int AST = BR, BST = BN;          // operands are read 32 consecutive floats at a time: no padding needed
  constexpr int A_F4 = BR / 4, A_PPP = 256 / A_F4, A_P = BKP / A_PPP;   // positions per pass
  constexpr int B_F4 = BN / 4, B_PPP = 256 / B_F4, B_P = BKP / B_PPP;
  constexpr int STAGE = BKP * AST + BKP * BST;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhalf = lane >> 5;
  // 1-D grid, (tap,c) tile fastest: the tiles of one row split read the same source rows and run on one XCD
  const int swz = rd_xcd_swizzle(blockIdx.x, gridDim.x);
  const int tiles = T.RT * T.NT;
  const int bx = swz % tiles, by = (swz / tiles) % T.nsplit, bz = swz / (tiles * T.nsplit);
  const RdPhase& P = plan->ph[bz];
  const int L = P.L;
  const int rows = B * L;
  const int rt = bx / T.NT, ntile = bx - rt * T.NT;
  const int n0 = ntile * BN;
  const int mbeg = by * T.rows_per_split;
  const int mend = min(rows, mbeg + T.rows_per_split);
  const int SH = plan->SH, SW = plan->SW;
  const int cstride = plan->s_cstride, SC = plan->SC;
  const int ssample = (int)plan->src_sample, dsample = (int)plan->dst_sample;
  const RdRow* __restrict__ tab = plan->tab + P.tab;
  // descriptors are based at the first sample this block touches (wave-uniform)
  const int bb0 = mbeg / L;
  const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc(src + (long)bb0 * plan->src_sample);
  const __amdgpu_buffer_rsrc_t rsB = rd_make_rsrc(dy + (long)bb0 * plan->dst_sample);

  // this thread's A column group (tap, c) is fixed over the whole loop
  int a_tap, a_c;
  const int a_r = (tid % A_F4) * 4;
  rd_wgrad_tile_row(T, BR, rt, a_r, a_tap, a_c);
  const bool a_ok = a_tap < P.ntaps && a_c < SC;
  int tmask = 0x7FFF, tdelta = 0, sd = 0, sh_ = 0, sw = 0;   // tmask never matches when !a_ok
  if (a_ok) {
    const RdTap t = P.tap[a_tap];
    tmask = t.mask;
    if (SHIFT) { sd = t.code & 255; sh_ = (t.code >> 8) & 255; sw = t.code >> 16; }
    else tdelta = t.delta;
  }
  const int a_const = tdelta + a_c * 4;
  const int b_const = (n0 + (tid % B_F4) * 4) * 4;

  // row cursors (sample index relative to bb0, row inside the sample) of this thread's A and B positions
  int ab[A_P], al[A_P], gb[B_P], gl[B_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) { int m = mbeg + tid / A_F4 + i * A_PPP; ab[i] = m / L; al[i] = m - ab[i] * L; ab[i] -= bb0; }
#pragma unroll
  for (int i = 0; i < B_P; ++i) { int m = mbeg + tid / B_F4 + i * B_PPP; gb[i] = m / L; gl[i] = m - gb[i] * L; gb[i] -= bb0; }

  // row-table entries of the NEXT chunk's rows, fetched one load_chunk call ahead so the gather never waits
  // on a table load it has just issued
  RdRow ea[A_P];
  int ez[B_P];
  auto fetch_rows = [&]() {
#pragma unroll
    for (int i = 0; i < A_P; ++i) ea[i] = tab[al[i]];
#pragma unroll
    for (int i = 0; i < B_P; ++i) ez[i] = tab[gl[i]].z;
  };
  fetch_rows();

  f32x4 ra[A_P], rg[B_P];
  auto load_chunk = [&](int mb) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      int m = mb + tid / A_F4 + i * A_PPP;
      unsigned voff = RD_OOB;
      {
        const RdRow e = ea[i];
        int off = (ab[i] * ssample + e.x) * 4 + a_const;
        if (SHIFT) off += rd_shift_delta(e.w, sd, sh_, sw, SH, SW, cstride) * 4;
        if (m < mend && (e.y & tmask) == tmask) voff = (unsigned)off;
      }
      f32x4 v = rd_buf_load4(rsA, voff);
      if (PARTIAL) {
        if (a_c + 1 >= SC) v.y = 0.f;
        if (a_c + 2 >= SC) v.z = 0.f;
        if (a_c + 3 >= SC) v.w = 0.f;
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      int m = mb + tid / B_F4 + i * B_PPP;
      unsigned voff = m < mend ? (unsigned)((gb[i] * dsample + ez[i]) * 4 + b_const) : RD_OOB;
      rg[i] = rd_buf_load4(rsB, voff);
    }
    // advance the cursors by one chunk (BKP rows)
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      al[i] += BKP;
      if (L >= BKP) { if (al[i] >= L) { al[i] -= L; ab[i] += 1; } }
      else { int qd = al[i] / L; al[i] -= qd * L; ab[i] += qd; }
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      gl[i] += BKP;
      if (L >= BKP) { if (gl[i] >= L) { gl[i] -= L; gb[i] += 1; } }
      else { int qd = gl[i] / L; gl[i] -= qd * L; gb[i] += qd; }
    }
    fetch_rows();        // al/gl < L always, so the table reads stay in range even past the last chunk
  };
  auto store_chunk = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BKP * AST;
#pragma unroll
    for (int i = 0; i < A_P; ++i) *(f32x4*)&As[(tid / A_F4 + i * A_PPP) * AST + a_r] = ra[i];
#pragma unroll
    for (int i = 0; i < B_P; ++i) *(f32x4*)&Bs[(tid / B_F4 + i * B_PPP) * BST + (tid % B_F4) * 4] = rg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = (mend - mbeg + BKP - 1) / BKP;
  if (nchunks > 0) {
    load_chunk(mbeg);                       // the row cursors advance one chunk per call: calls must stay in order
    store_chunk(0);
  }
  __syncthreads();
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
#ifndef RD_ABL_W
    if (q + 1 < nchunks) load_chunk(mbeg + (q + 1) * BKP);
#endif
    const float* As = smem + buf * STAGE + lhalf * AST + wm * WTM + l31;
    const float* Bs = smem + buf * STAGE + BKP * AST + lhalf * BST + wn * WTN + l31;
    // operands of 4 k-steps (8 rows) per group, double-buffered in registers
    constexpr int NG = BKP / 8;
    float fa[2][4][TM], fb[2][4][TN];
    auto load_frag = [&](int slot, int g) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][s][i] = As[(g * 8 + 2 * s) * AST + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][s][j] = Bs[(g * 8 + 2 * s) * BST + j * 32];
      }
    };
    load_frag(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int cur = g & 1;
      if (g + 1 < NG) load_frag(cur ^ 1, g + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s][i], fb[cur][s][j], acc[i][j], 0, 0, 0);
    }
    if (q + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  const int N = plan->N;
  float* out = partial + (((long)bz * T.nsplit + by) * T.RT + rt) * BR * N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
#pragma unroll
      for (int j = 0; j < TN; ++j) out[(long)row * N + n0 + wn * WTN + j * 32 + l31] = acc[i][j][r];
    }
}

// fold the split partials and scatter rows (tap,c) to their place in the weight gradient
__global__ void k_wgrad_reduce(const RdPlan* __restrict__ plan, const float* __restrict__ partial, int nsplit,
                               RdWgradTiling T, int BR, float* __restrict__ dW, int ldw) {
  const int N = plan->N;
  const int n4s = N / 4;
  const long total = (long)T.RT * BR * n4s;
  const RdPhase& P = plan->ph[blockIdx.y];
  const long stride = (long)T.RT * BR * N;
  const float* pbase = partial + (long)blockIdx.y * nsplit * stride;
  for (long f = blockIdx.x * (long)blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    int R = (int)(f / n4s), n = (int)(f - (long)R * n4s) * 4;
    int rt = R / BR, r = R - rt * BR, tap, c;
    rd_wgrad_tile_row(T, BR, rt, r, tap, c);
    if (tap >= P.ntaps || c >= plan->SC) continue;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* p = pbase + (long)R * N + n;
    for (int k = 0; k < nsplit; ++k) s += *(const f32x4*)(p + k * stride);
    *(f32x4*)(dW + P.w_off + ((long)P.tap[tap].w * plan->w_rows_per_tap + c) * ldw + n) = s;
  }
}
