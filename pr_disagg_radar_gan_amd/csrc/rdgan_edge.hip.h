// Dedicated kernels for the two "edge" convolutions of the step, whose GEMM shapes are too thin for the tiled conv kernels:
// the generator's last Conv3D (64 -> 1: K = 64, N = 27; T:345) and the critic's first Conv3D (2 -> 64 channels, stride 2
// 'valid': K = 54; T:286).  The tiled kernels spend their time in per-tile prologues and (first critic layer) in nine K chunks
// of 8 with a barrier each.  What did NOT work for the first critic layer (round 2): a whole sample in LDS with FMAs on
// LDS-broadcast inputs (LDS-bound), FMAs fed from scalar loads (SGPR spills in the forward, scalar-cache misses in the weight
// gradient).  What does: the im2col row as ONE K = 64 operand row staged through registers, the matrix pipe, persistent tiles.
#pragma once
#include "rdgan_gemm_ws16.hip.h"

// ------------------------------------------------------------------------------------
// Last generator conv, forward (T:345): per grid point the 27 column products P[pos][tap] = h3[pos][:] . W9[tap][:], with the
// sum over the taps whose neighbour lies inside the 256-row tile taken right away (the tile holds whole (h,w) planes for
// ndomain 8 / 16 -> NQ = 3 sums per point, one per kd; whole w rows for ndomain 32 / 64 / 128 -> NQ = 9, one per (kd,kh)).
// Output Q[plane][NQ][h][w] as RD_EPI_TAPGATHER of k_conv_gemm writes it; k_tapsum_softmax finishes the sum + softmax.
//
// One 256-row tile per workgroup (4 waves), two workgroups per CU: the whole tile [256][64] comes in by LDS-DMA (16-byte
// chunk c of row r stored at c ^ (r & 15) for fp32 rows, c ^ ((r >> 1) & 7) for bf16 rows: conflict-free b128 fragment
// reads), one wave multiplies 64 rows x 32 taps on the matrix pipe (fp32: v_mfma_f32_32x32x2_f32, exact; bf16 storage mode:
// v_mfma_f32_32x32x16_bf16 against the bf16-rounded kernel), the tile of products goes back through LDS for the tap sums.
// Algorithmic bytes: the h3 tensor once (256 B or 128 B per grid point) + NQ floats per grid point written.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256, 2)
k_g9_fwd(const T* __restrict__ h3, const float* __restrict__ w9 /* [27][64] */, float* __restrict__ Q, long rows, int Wd,
         int HW, int NQ) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int ROWB = BF ? 128 : 256;                       // bytes per row of 64 channels
  constexpr int CST = 33;                                     // product tile row stride (floats)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                           // [256] rows of ROWB bytes (swizzled), later the product tile [256][33]
  float* Ws = smem + 256 * ROWB / 4;                          // fp32: W9T [64 k][32 n]; bf16: [32 n] rows of 128 bytes (swizzled like As)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const long m0 = (long)blockIdx.x * 256;

  // ---- the tile: 64 rows per wave by DMA (out-of-range rows -> zeros)
  {
    const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(h3 + m0 * 64));
    constexpr int RPI = 1024 / ROWB;                          // rows per DMA instruction
    constexpr int CPR = ROWB / 16;                            // 16-byte chunks per row
#pragma unroll
    for (int k = 0; k < 64 / RPI; ++k) {
      const int r = wave * 64 + k * RPI + lane / CPR;
      const int p = lane % CPR;
      const int c_log = BF ? (p ^ ((r >> 1) & 7)) : (p ^ (r & 15));
      unsigned voff = m0 + r < rows ? (unsigned)(r * ROWB + c_log * 16) : RD_OOB;
      asm volatile("" : "+v"(voff));
      rd_lds_dma16(rs, As + (wave * 64 + k * RPI) * (ROWB / 4), (int)voff, 0);
    }
  }
  // ---- the kernel (L2-resident), written to LDS in the operand layout
  if constexpr (BF) {
    // row n (tap, zero for n >= 27): 64 bf16 = 8 chunks, chunk c at c ^ ((n >> 1) & 7); thread -> (n = tid / 8, chunk = tid % 8)
    const int n = tid >> 3, c = tid & 7;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (n < 27) { a = *(const f32x4*)(w9 + n * 64 + c * 8); b = *(const f32x4*)(w9 + n * 64 + c * 8 + 4); }
    u32x4_t o = {rd_pack_bf16(a.x, a.y), rd_pack_bf16(a.z, a.w), rd_pack_bf16(b.x, b.y), rd_pack_bf16(b.z, b.w)};
    *(u32x4_t*)((char*)Ws + n * 128 + ((c ^ ((n >> 1) & 7)) * 16)) = o;
  } else {
    // Ws[k][n] = w9[n][k]: thread -> k = tid / 4, n = (tid % 4) * 8 .. +7
    const int k = tid >> 2, n0 = (tid & 3) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) Ws[k * 32 + n0 + e] = n0 + e < 27 ? w9[(n0 + e) * 64 + k] : 0.f;
  }
  rd_dma_landed();
  __syncthreads();                                            // the tile has landed

  // ---- P[64 rows of this wave][32 taps]
  f32x16 acc[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  if constexpr (BF) {
    const char* Ab = (const char*)As;
    const char* Wb = (const char*)Ws + l31 * 128;
    const int w_sw = (l31 >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 fb = *(const f32x4*)(Wb + (((kk * 2 + lhalf) ^ w_sw) * 16));
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int i = wave * 64 + rb * 32 + l31;
        const f32x4 fa = *(const f32x4*)(Ab + i * 128 + (((kk * 2 + lhalf) ^ ((i >> 1) & 7)) * 16));
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, fa), __builtin_bit_cast(rd_bf16x8, fb),
                                                          acc[rb], 0, 0, 0);
      }
    }
  } else {
    const float* Wl = Ws + lhalf * 4 * 32 + l31;
#pragma unroll
    for (int j8 = 0; j8 < 8; ++j8) {
      f32x4 fa[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int i = wave * 64 + rb * 32 + l31;
        fa[rb] = *(const f32x4*)&As[i * 64 + (((j8 * 2 + lhalf) ^ (i & 15)) * 4)];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float fb = Wl[(j8 * 8 + s) * 32];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rb][s], fb, acc[rb], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                            // every wave has read its rows: the tile region becomes the product tile
  float* Cs = smem;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      Cs[row * CST + l31] = acc[rb][r];
    }
  __syncthreads();
  // ---- sums over the taps whose neighbour rows lie inside the tile (same arithmetic and order as RD_EPI_TAPGATHER)
  const int Hd = HW / Wd;
  for (int o = tid; o < 256 * NQ; o += 256) {
    const int r = o & 255, j = o >> 8;
    const long m = m0 + r;
    if (m >= rows) continue;
    const long pl = m / HW;
    const int hw = (int)(m - pl * HW), hh = hw / Wd, ww = hw - hh * Wd;
    float s = 0.f;
    if (NQ == 9) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        if ((unsigned)(ww + kw - 1) < (unsigned)Wd) s += Cs[(r + kw - 1) * CST + j * 3 + kw];
    } else {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        if ((unsigned)(hh + kh - 1) >= (unsigned)Hd) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(ww + kw - 1) < (unsigned)Wd) s += Cs[(r + (kh - 1) * Wd + kw - 1) * CST + (j * 3 + kh) * 3 + kw];
      }
    }
    Q[(pl * NQ + j) * HW + hw] = s;
  }
}

// ------------------------------------------------------------------------------------
// Last generator conv, forward, fp32 storage (T:345), pipelined: the same products as k_g9_fwd<float> / the tap-gathering GEMM
// on 128-pixel tiles (R = 128 / W whole w rows, W a power of two <= 128) by persistent workgroups, three per CU.  A tile's
// fragments are read into registers, then -- behind a barrier -- the DMA of the workgroup's next tile is issued, and only then
// come the 32 MFMAs, the product tile P[128][32 taps] and the sums over kw (NQ = 9 sums per grid point in the tap-gather
// format, whatever the domain size): the next 32 KB fly while this tile is worked on.  k_g9_fwd<float> (one 64 KB tile per
// workgroup, nothing in flight while it computes) and the tiled GEMM both need ~150 us for the 403 MB at bs 256.
// The kernel's 27 x 64 weights sit in registers (32 per lane) for the whole launch.
// ------------------------------------------------------------------------------------
// TP = 128 (256 threads, three workgroups per CU): any W; nine kw-sums per grid point.  TP = 256 (512 threads, one workgroup per
// CU with 64 KB in flight): tiles of whole (h,w) planes (ndomain 8 / 16), three (kh,kw)-sums per grid point -- k_tapsum_softmax
// then reads 3 instead of 9 values per tap plane (8 instead of 32 us at bs 256).
template <int TP>
__global__ void __launch_bounds__(TP * 2, TP == 128 ? 3 : 1)
k_g9_fwd_mfma(const float* __restrict__ h3, const float* __restrict__ w9 /* [27][64] */, float* __restrict__ Q, long rows, int Wd,
              int hwlog2) {
  constexpr int CST = 33, NT = TP * 2, NQ = TP == 128 ? 9 : 3;
  static_assert(TP == 128 || TP == 256, "pixels per tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                                           // [TP][64], 16-byte chunk c of row r stored at c ^ (r & 15)
  float* Pt = smem + TP * 64;                                 // [TP][CST] products
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int HW = 1 << hwlog2;
  // B operand, constant: k-step s multiplies channel c = 32 lhalf + s; lane n = tap l31 (taps 27..31: zero)
  float wv[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) wv[s] = l31 < 27 ? w9[l31 * 64 + 32 * lhalf + s] : 0.f;
  auto issue = [&](unsigned m0) {         // 32 rows per wave by DMA, swizzled on the source side (rows beyond the tensor -> zeros)
    const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc(h3 + (long)m0 * 64);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = wave * 32 + k * 4 + (lane >> 4);
      const int c_log = (lane & 15) ^ (r & 15);
      unsigned voff = m0 + r < (unsigned)rows ? (unsigned)(r * 256 + c_log * 16) : RD_OOB;
      asm volatile("" : "+v"(voff));
      rd_lds_dma16(rs, Hs + (wave * 32 + k * 4) * 64, (int)voff, 0);
    }
  };
  const long ntiles = (rows + TP - 1) / TP;
  long tile = blockIdx.x;
  if (tile < ntiles) issue((unsigned)(tile * TP));
  rd_dma_landed();
  for (; tile < ntiles; tile += gridDim.x) {
    const unsigned m0 = (unsigned)(tile * TP);                // (the host keeps rows < 2^31)
    __syncthreads();                                          // the tile has landed (every wave waited for its DMA: below, and in
    //                                                           front of the loop for the first tile); the last tile's sums are out
    const int i = wave * 32 + l31;                            // this lane's pixel (MFMA row)
    f32x4 fa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = *(const f32x4*)&Hs[i * 64 + (((8 * lhalf + j) ^ (i & 15)) * 4)];
    __syncthreads();                                          // every wave holds its fragments: the tile's LDS is free
    const long next = tile + gridDim.x;
    if (next < ntiles) issue((unsigned)(next * TP));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s >> 2][s & 3], wv[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Pt[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * CST + l31] = acc[r];
    rd_dma_landed();                // the next tile's DMA had the 32 MFMAs to land; the Q stores below stay in flight over the loop head
    __syncthreads();
    // sums over the taps whose neighbours lie inside the tile (same arithmetic and order as RD_EPI_TAPGATHER)
    const int Hd = HW / Wd;
    for (int o = tid; o < TP * NQ; o += NT) {
      const int r = o & (TP - 1), j = o / TP;
      const unsigned m = m0 + (unsigned)r;
      if (m >= (unsigned)rows) continue;
      const unsigned pl = m >> hwlog2, hw = m & (unsigned)(HW - 1);
      const int ww = (int)(hw & (unsigned)(Wd - 1)), hh = (int)(hw / (unsigned)Wd);
      float sum = 0.f;
      if constexpr (NQ == 9) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if ((unsigned)(ww + kw - 1) < (unsigned)Wd) sum += Pt[(r + kw - 1) * CST + j * 3 + kw];
      } else {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          if ((unsigned)(hh + kh - 1) >= (unsigned)Hd) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            if ((unsigned)(ww + kw - 1) < (unsigned)Wd) sum += Pt[(r + (kh - 1) * Wd + kw - 1) * CST + (j * 3 + kh) * 3 + kw];
        }
      }
      Q[((long)pl * NQ + j) * HW + hw] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------
// Last generator conv, weight gradient (T:345 backward): dW9[tap][c] = sum_u dl[u - off(tap)] * h3[u][c] as a GEMM on the matrix
// pipe, C[32 taps][64 c] += A^T[32][128 pixels] x B[128][64] per 128-pixel tile (W a power of two <= 128: the tile is R = 128 / W
// whole w rows).  B = the h3 rows of the tile, in by LDS-DMA: the 403 MB tensor is read once.  A is never materialised: for each
// (kd, kh) and tile row the dlogits row at (d + 1 - kd, h + 1 - kh) is staged in LDS with a zero halo in w (9 R rows of W + 2
// floats, border rows zero; dl is 1 channel, 6 MB, L2-resident), and a lane reads A[tap][pixel] = stage[(kd, kh)][row][w + 2 - kw]
// at (lane-constant tap offset) + (pixel offset).  Persistent workgroups (four per CU), each wave multiplies 32 of the tile's
// pixels and keeps its [32][64] sums in registers; per workgroup one partial [27][64] slab, folded by k_reduce_partials
// (deterministic).  fp32 MFMAs (v_mfma_f32_32x32x2_f32) also in the bf16 storage mode -- the bf16 rows are widened on the
// fragment read -- so the products are exactly those of the scalar kernel it replaces (k_g9_wgrad_pairs: one activation load
// pair per pixel in front of 54 FMAs fed by LDS broadcasts, 2.4-2.7 TB/s).  First version of this kernel: the 27 neighbours of
// every pixel gathered from global memory into an A^T tile -- ~700 VALU instructions per pixel and tile for the index
// arithmetic, slower than the MFMAs and the DMA together (105 us at bs 256; 64-pixel tiles: 127 us).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256, 3)        // (four workgroups per CU would fit in LDS, but 128 VGPRs spill the fragments)
k_g9_wgrad_mfma(const float* __restrict__ dl, const T* __restrict__ h3, float* __restrict__ partial, long rows, int D, int H, int W,
                int wlog2) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int TP = 128;                                      // pixels per tile
  constexpr int ROWB = BF ? 128 : 256;                        // bytes per row of 64 channels
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                                            // [TP] rows of ROWB bytes
  float* St = smem + TP * ROWB / 4;                            // [9][R][W + 2] staged dlogits rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int R = TP >> wlog2, RS = W + 2, rlog2 = 7 - wlog2;
  // lane-constant part of the A address: tap = l31 = (kd*3 + kh)*3 + kw -> stage row block (kd*3 + kh), column shift 2 - kw
  const bool tapok = l31 < 27;
  const int tkk = l31 / 3, tkw = l31 - tkk * 3;
  const int tapoff = tapok ? tkk * R * RS + 2 - tkw : 0;
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  const long ntiles = (rows + TP - 1) / TP;
  // Software pipeline over the workgroup's tiles: the fragments of tile i are read into registers, then -- behind a barrier --
  // the DMA of tile i+1 and the dlogits loads of its staged rows are issued, and only then come tile i's 32 MFMAs: the h3 rows
  // and the dlogits of the next tile fly while the matrix pipe works (the first version waited for each tile's DMA with
  // nothing else to do: 104 us at bs 256).
  auto issue_b = [&](unsigned m0) {       // B: 32 rows per wave by DMA (rows beyond the tensor -> zeros)
    const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(h3 + (long)m0 * 64));
    constexpr int RPI = 1024 / ROWB, CPR = ROWB / 16;
#pragma unroll
    for (int k = 0; k < 32 / RPI; ++k) {
      const int r = wave * 32 + k * RPI + lane / CPR;
      unsigned voff = m0 + r < (unsigned)rows ? (unsigned)(r * ROWB + (lane % CPR) * 16) : RD_OOB;
      asm volatile("" : "+v"(voff));
      rd_lds_dma16(rs, Bs + (wave * 32 + k * RPI) * (ROWB / 4), (int)voff, 0);
    }
  };
  // staged rows: pass q = (kd, kh) stages R rows of W + 2 floats, thread -> (tile row r = tid / 2W, column wc = tid % 2W);
  // row (q, r) is dl[b, d+1-kd, h+1-kh, :] of the tile row's (b, d, h), zero outside the picture
  const int sr = tid >> (wlog2 + 1), swc = tid & (2 * W - 1);
  float sv[9];
  unsigned sok = 0;                       // bit q: row (q, r) exists (the select waits until store_stage: the loads stay in flight)
  auto load_stage = [&](unsigned m0) {
    sok = 0;
    const unsigned rowid = (m0 >> wlog2) + (unsigned)sr;
    const unsigned t = rowid / (unsigned)H;
    const int h_ = (int)(rowid - t * (unsigned)H);
    const unsigned b = t / (unsigned)D;
    const int d_ = (int)(t - b * (unsigned)D);
    const bool live = (long)rowid * W < rows && swc >= 1 && swc <= W;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int dd = d_ + 1 - q / 3, hh = h_ + 1 - q % 3;
      const bool ok = live && (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H;
      const unsigned idx = ok ? ((b * (unsigned)D + (unsigned)dd) * (unsigned)H + (unsigned)hh) * (unsigned)W + (unsigned)(swc - 1) : 0u;
      sv[q] = dl[idx];
      sok |= ok ? 1u << q : 0u;
    }
  };
  auto store_stage = [&]() {
    if (swc < RS) {
#pragma unroll
      for (int q = 0; q < 9; ++q) St[(q * R + sr) * RS + swc] = (sok >> q & 1u) ? sv[q] : 0.f;
    }
  };
  long tile = blockIdx.x;
  if (tile < ntiles) {
    issue_b((unsigned)(tile * TP));
    load_stage((unsigned)(tile * TP));
    store_stage();
  }
  for (; tile < ntiles; tile += gridDim.x) {
    rd_dma_landed();
    __syncthreads();                                           // tile: h3 rows landed, rows staged
    // this wave's 32 pixels: k-step s multiplies pixels base + s (lanes 0-31) and base + 16 + s (lanes 32-63)
    const int base = wave * 32 + 16 * lhalf;
    float a[16], b0[16], b1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int p = base + s;
      const float av = St[tapoff + (p >> wlog2) * RS + (p & (W - 1))];
      a[s] = tapok ? av : 0.f;
      if constexpr (BF) {
        const unsigned short* Bh = (const unsigned short*)Bs;
        b0[s] = __builtin_bit_cast(float, (unsigned)Bh[p * 64 + l31] << 16);
        b1[s] = __builtin_bit_cast(float, (unsigned)Bh[p * 64 + 32 + l31] << 16);
      } else {
        b0[s] = Bs[p * 64 + l31];
        b1[s] = Bs[p * 64 + 32 + l31];
      }
    }
    __syncthreads();                                           // every wave holds its fragments: the tile's LDS is free
    const long next = tile + gridDim.x;
    if (next < ntiles) {            // dlogits loads first: loads return in order, and store_stage must not wait for the DMA behind them
      load_stage((unsigned)(next * TP));
      issue_b((unsigned)(next * TP));
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], acc[1], 0, 0, 0);
    }
    if (next < ntiles) store_stage();
  }
  __syncthreads();
  // fold the four waves through LDS: red[wave][tap][c]; accumulator register r of a lane = row (r&3) + 8 (r>>2) + 4 lhalf
  float* red = smem;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tap = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      red[(wave * 32 + tap) * 64 + j * 32 + l31] = acc[j][r];
    }
  __syncthreads();
  for (int i = tid; i < 27 * 64; i += 256)
    partial[(long)blockIdx.x * 1728 + i] = red[i] + red[32 * 64 + i] + red[2 * 32 * 64 + i] + red[3 * 32 * 64 + i];
}

// ------------------------------------------------------------------------------------
// First critic layer (T:286-289): Conv3D(64, 3x3x3, stride 2, 'valid') on the 2-channel volume (sample | condition),
// as ONE K = 64 GEMM per tile: an output position's im2col row is 9 (kd,kh) segments of 6 contiguous floats (kw, ci) =
// 54 values, padded to 64.  The tiled conv kernel walks the 9 segments as 9 K chunks of 8 (6 used) with a barrier each;
// here a workgroup stages the whole [128 rows][64] operand once (each segment = three 8-byte loads), multiplies it against
// the kernel [64][64] (rows 54.. zero) on the fp32 matrix pipe (exact, k in the reference's tap order), and runs the
// layer's epilogue on the tile.  Any ndomain; one condition channel (2 floats per voxel).
// MODE 0: out = dropout(LeakyReLU(conv + bias))                        (forward, T:286-289)
// MODE 1: out = gate(aux) * conv, gate = LeakyReLU'(aux) * dropout     (second forward sweep of the gradient penalty; aux may be out)
// TO = element type of out / aux (bf16 storage mode).  idx_base: added to the flat output index for the dropout counter.
// rows = samples * NPOS; cin [samples][D][nd][nd][2].
// ------------------------------------------------------------------------------------
// A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the kernel image is staged once, and the next tile's
// segment loads are issued (into registers) before the current tile's epilogue, which hides their latency.
template <typename TO, int MODE>
__global__ void __launch_bounds__(256, 3)
k_d1_gemm_fwd(const float* __restrict__ cin, const float* __restrict__ w, const float* __restrict__ bias, TO* out, const TO* aux,
              long rows, int nd, int Do, int Ho, int Wo, int use_drop, uint32_t key, uint32_t idx_base,
              unsigned char* __restrict__ gbits = nullptr) {
  constexpr int BM = 128;
  constexpr bool BF = sizeof(TO) == 2;    // bf16 storage mode: operands rounded to bf16, v_mfma_f32_32x32x16_bf16
  constexpr int NSG = (BM * 9 + 255) / 256;             // segments per thread (5; the last one only for tid < 128)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // fp32: As [128 rows][64 k] floats, 16-byte chunk c of row r at c ^ (r & 15);  Ws [64 k][64 n]
  // bf16: As [128 rows] of 128 bytes,  chunk c at c ^ ((r >> 1) & 7);            Ws [64 n] rows of 128 bytes (k), same swizzle
  // the As region is reused for the fp32 output tile [128][64] (32 KiB); behind Ws: two row-offset tables
  float* As = smem;
  float* Ws = smem + BM * 64;
  long* rowoff = (long*)(smem + BM * 64 + 64 * 64);     // [2][128] element offset of the row's window in cin, -1 = no row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int NPOS = Do * Ho * Wo, HoWo = Ho * Wo;
  const long nin = (long)RDGAN_NHOURS * nd * nd * 2;
  const long ntiles = (rows + BM - 1) / BM;
  auto decode_rows = [&](long tile, int slot) {        // threads 0 .. 127: one row each
    if (tid < BM) {
      const long m = tile * BM + tid;
      long off = -1;
      if (m < rows) {
        const long b = m / NPOS;
        const int p = (int)(m - b * NPOS);
        const int od = p / HoWo, q = p - od * HoWo, oh = q / Wo, ow = q - oh * Wo;
        off = b * nin + ((long)(2 * od * nd + 2 * oh) * nd + 2 * ow) * 2;
      }
      rowoff[slot * BM + tid] = off;
    }
  };
  float2 sa[NSG][3];
  auto load_segments = [&](int slot) {                 // consecutive lanes: consecutive rows of one segment (kd,kh)
#pragma unroll
    for (int u = 0; u < NSG; ++u) {
      const int sgi = tid + u * 256;
      float2 a = {0.f, 0.f}, c = a, e = a;
      if (sgi < BM * 9) {
        const int sg = sgi >> 7, r = sgi & (BM - 1);
        const long off = rowoff[slot * BM + r];
        if (off >= 0) {
          const int kd = sg / 3, kh = sg - kd * 3;
          const float* src = cin + off + (kd * nd + kh) * nd * 2;
          a = *(const float2*)src; c = *(const float2*)(src + 2); e = *(const float2*)(src + 4);
        }
      }
      sa[u][0] = a; sa[u][1] = c; sa[u][2] = e;
    }
  };
  // kernel [54][64] (+ zero rows) in the operand layout, once
  if constexpr (BF) {
    for (int i = tid; i < 64 * 8; i += 256) {            // (n, chunk c): 8 consecutive k
      const int n = i >> 3, c = i & 7;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const int k = c * 8 + e; v[e] = k < 54 ? w[k * 64 + n] : 0.f; }
      u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
      *(u32x4_t*)((char*)Ws + n * 128 + ((c ^ ((n >> 1) & 7)) * 16)) = o;
    }
  } else {
    for (int i = tid; i < 64 * 16; i += 256) {
      const int k = i >> 4;
      f32x4 wq = {0.f, 0.f, 0.f, 0.f};
      if (k < 54) wq = *(const f32x4*)(w + 4 * i);
      *(f32x4*)&Ws[4 * i] = wq;
    }
  }
  const int c4 = (tid & 15) * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) bias4 = *(const f32x4*)(bias + c4);
  long tile = blockIdx.x;
  int slot = 0;
  if (tile < ntiles) decode_rows(tile, 0);
  __syncthreads();
  if (tile < ntiles) load_segments(0);
  for (; tile < ntiles; tile += gridDim.x, slot ^= 1) {
    const long m0 = tile * BM;
    // ---- this tile's operand rows: registers -> LDS (segment sg -> logical k = 6 sg .. 6 sg + 5, then zero to 63)
#pragma unroll
    for (int u = 0; u < NSG; ++u) {
      const int sgi = tid + u * 256;
      if (sgi < BM * 9) {
        const int sg = sgi >> 7, r = sgi & (BM - 1), k0 = sg * 6;
        if constexpr (BF) {
          char* rowp = (char*)As + r * 128;
          const int sw = (r >> 1) & 7;
          // bf16 pair holding k, k + 1: byte (k & 7) * 2 of chunk k >> 3
          *(unsigned*)(rowp + ((((k0) >> 3) ^ sw) * 16) + ((k0) & 7) * 2) = rd_pack_bf16(sa[u][0].x, sa[u][0].y);
          *(unsigned*)(rowp + ((((k0 + 2) >> 3) ^ sw) * 16) + ((k0 + 2) & 7) * 2) = rd_pack_bf16(sa[u][1].x, sa[u][1].y);
          *(unsigned*)(rowp + ((((k0 + 4) >> 3) ^ sw) * 16) + ((k0 + 4) & 7) * 2) = rd_pack_bf16(sa[u][2].x, sa[u][2].y);
        } else {
          float* rowp = As + r * 64;
          const int sw = r & 15;
          *(float2*)(rowp + ((((k0) >> 2) ^ sw) << 2) + ((k0) & 3)) = sa[u][0];
          *(float2*)(rowp + ((((k0 + 2) >> 2) ^ sw) << 2) + ((k0 + 2) & 3)) = sa[u][1];
          *(float2*)(rowp + ((((k0 + 4) >> 2) ^ sw) << 2) + ((k0 + 4) & 3)) = sa[u][2];
        }
      }
    }
    for (int i = tid; i < BM * 5; i += 256) {            // k = 54 .. 63: zero (five pairs per row)
      const int r = i / 5, k = 54 + 2 * (i - r * 5);
      if constexpr (BF) *(unsigned*)((char*)As + r * 128 + (((k >> 3) ^ ((r >> 1) & 7)) * 16) + (k & 7) * 2) = 0u;
      else *(float2*)(As + r * 64 + (((k >> 2) ^ (r & 15)) << 2) + (k & 3)) = (float2){0.f, 0.f};
    }
    const long next = tile + gridDim.x;
    if (next < ntiles) decode_rows(next, slot ^ 1);
    __syncthreads();
    // ---- each wave: 32 rows x 64 columns
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    {
      const int i = wave * 32 + l31;
      if constexpr (BF) {
        const char* Ab = (const char*)As + i * 128;
        const int a_sw = (i >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f32x4 fa = *(const f32x4*)(Ab + (((kk * 2 + lhalf) ^ a_sw) * 16));
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = j * 32 + l31;
            const f32x4 fb = *(const f32x4*)((const char*)Ws + n * 128 + (((kk * 2 + lhalf) ^ ((n >> 1) & 7)) * 16));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, fa), __builtin_bit_cast(rd_bf16x8, fb),
                                                             acc[j], 0, 0, 0);
          }
        }
      } else {
        const float* Wl = Ws + lhalf * 4 * 64 + l31;
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) {
          const f32x4 fa = *(const f32x4*)&As[i * 64 + (((j8 * 2 + lhalf) ^ (i & 15)) << 2)];
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], Wl[(j8 * 8 + s) * 64 + j * 32], acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();                         // operand reads done (the region becomes the output tile); next row offsets visible
    if (next < ntiles) load_segments(slot ^ 1);        // in flight during the epilogue below
    float* Cs = smem;                        // [128][64]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        Cs[row * 64 + j * 32 + l31] = acc[j][r];
      }
    __syncthreads();
    // ---- epilogue: thread -> channel quad tid % 16, rows tid / 16 + 16 k
#pragma unroll 4
    for (int row = tid >> 4; row < BM; row += 16) {
      const long m = m0 + row;
      if (m >= rows) break;
      f32x4 v = *(const f32x4*)&Cs[row * 64 + c4];
      const long idx = m * 64 + c4;
      if (MODE == 0) {
        v += bias4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = rd_lrelu(v[e]);
          if (use_drop) t = rd_drop_apply_w(t, rd_drop_word(key, (uint32_t)idx + idx_base), e);
          v[e] = t;
        }
        // MODE 0, optional: the layer's gate in 2 bits per element for k_d2_dgrad_slab16 -- bit 0: output > 0 (LeakyReLU' = 1),
        // bit 1: dropped (+0.0 under dropout, rd_drop_apply); one byte per channel quad, 16 bytes per row
        if (gbits) {
          unsigned code = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ve = v[e];       // (by value: __builtin_bit_cast on the vector ELEMENT v[e] read element 0 for every e)
            code |= ((ve > 0.f ? 1u : 0u) | ((use_drop && __builtin_bit_cast(unsigned, ve) == 0u) ? 2u : 0u)) << (2 * e);
          }
          gbits[m * 16 + (tid & 15)] = (unsigned char)code;
        }
      } else {
        const f32x4 a4 = rd_ld4(aux + idx);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = rd_gate_from_out(a4[e], use_drop);
          v[e] *= g;
        }
      }
      rd_st4(out + idx, v);
    }
    __syncthreads();                         // output tile read: the region may take the next operand rows
  }
}

// Weight gradient of the same layer: dW1[(tap,ci)][co] = sum over rows of im2col[row][(tap,ci)] * u1[row][co], as a GEMM
// with the rows as K: a workgroup owns a contiguous slice of rows, stages 32 rows at a time (the im2col rows as above, u1 as
// fp32) and accumulates the [64][64] product (wave = one 32x32 quadrant) on the fp32 matrix pipe; partial[blockIdx.x][64][64]
// (rows 54.. unused) goes to k_d1_wgrad_fold.  Fixed order: deterministic.  TG = element type of u1.
template <typename TG>
__global__ void __launch_bounds__(256, 4)
k_d1_gemm_wgrad(const float* __restrict__ cin, const TG* __restrict__ u1, float* __restrict__ partial, long rows,
                long rows_per_wg, int nd, int Do, int Ho, int Wo) {
  constexpr int BKR = 32;                   // rows per chunk
  constexpr int AST = 68;                   // im2col row stride in floats: consecutive rows 4 banks apart (64 would put a column in one bank)
  __shared__ __attribute__((aligned(16))) float As[2][BKR * AST];    // [row][k] (k >= 54: zero)
  __shared__ __attribute__((aligned(16))) float Gs[2][BKR * 64];     // [row][co]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const long mbeg = (long)blockIdx.x * rows_per_wg, mend = min(rows, mbeg + rows_per_wg);
  const int NPOS = Do * Ho * Wo, HoWo = Ho * Wo;
  const long nin = (long)RDGAN_NHOURS * nd * nd * 2;
  const int nchunks = mend > mbeg ? (int)((mend - mbeg + BKR - 1) / BKR) : 0;
  for (int i = tid; i < 2 * BKR * 10; i += 256) {      // the zero columns k = 54 .. 63 of both stages, once
    const int st = i / (BKR * 10), j = i - st * (BKR * 10), r = j / 10;
    As[st][r * AST + 54 + (j - r * 10)] = 0.f;
  }
  // element offset of each row's window in cin (-1 = no row), decoded by 32 threads one chunk ahead of its loads
  __shared__ long rowoff[2][BKR];
  auto decode_rows = [&](int q) {
    if (tid < BKR) {
      const long m = mbeg + (long)q * BKR + tid;
      long off = -1;
      if (m < mend) {
        const long b = m / NPOS;
        const int p = (int)(m - b * NPOS);
        const int od = p / HoWo, qq = p - od * HoWo, oh = qq / Wo, ow = qq - oh * Wo;
        off = b * nin + ((long)(2 * od * nd + 2 * oh) * nd + 2 * ow) * 2;
      }
      rowoff[q & 1][tid] = off;
    }
  };
  // staging is split: the loads of chunk q + 1 go out in front of chunk q's MFMAs, their LDS writes come behind them
  constexpr int NSG = (BKR * 9 + 255) / 256;          // im2col segments per thread (2)
  constexpr int NGQ = BKR * 16 / 256;                 // u1 quads per thread (2)
  float2 sa[NSG][3];
  f32x4 sg_[NGQ];
  auto load_regs = [&](int q) {
    const long mb = mbeg + (long)q * BKR;
#pragma unroll
    for (int u = 0; u < NSG; ++u) {
      const int sgi = tid + u * 256;
      const int sg = sgi >> 5, r = sgi & (BKR - 1);           // (lanes along the rows of one segment: see k_d1_gemm_fwd)
      float2 a = {0.f, 0.f}, c = a, e = a;
      const long off = sgi < BKR * 9 ? rowoff[q & 1][r] : -1;
      if (off >= 0) {
        const int kd = sg / 3, kh = sg - kd * 3;
        const float* src = cin + off + (kd * nd + kh) * nd * 2;
        a = *(const float2*)src; c = *(const float2*)(src + 2); e = *(const float2*)(src + 4);
      }
      sa[u][0] = a; sa[u][1] = c; sa[u][2] = e;
    }
#pragma unroll
    for (int u = 0; u < NGQ; ++u) {
      const int i = tid + u * 256;
      const int r = i >> 4, c4 = (i & 15) * 4;
      const long m = mb + r;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (m < mend) g = rd_ld4(u1 + m * 64 + c4);
      sg_[u] = g;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NSG; ++u) {
      const int sgi = tid + u * 256;
      if (sgi < BKR * 9) {
        const int sg = sgi >> 5, r = sgi & (BKR - 1);
        float* d = &As[buf][r * AST + sg * 6];
        *(float2*)d = sa[u][0]; *(float2*)(d + 2) = sa[u][1]; *(float2*)(d + 4) = sa[u][2];
      }
    }
#pragma unroll
    for (int u = 0; u < NGQ; ++u) {
      const int i = tid + u * 256;
      *(f32x4*)&Gs[buf][(i >> 4) * 64 + (i & 15) * 4] = sg_[u];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  decode_rows(0); decode_rows(1);                      // (rows beyond mend decode to -1)
  __syncthreads();
  if (nchunks > 0) { load_regs(0); store_lds(0); }
  __syncthreads();
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
    if (q + 1 < nchunks) load_regs(q + 1);
    decode_rows(q + 2);                                  // slot q & 1: last read by load_regs(q), a barrier ago
    // A operand = im2col^T: lane (i = l31, k = lhalf) <- As[row = 2 s + lhalf][wm * 32 + l31]; B = u1[row][wn * 32 + l31]
    const float* Al = &As[buf][lhalf * AST + wm * 32 + l31];
    const float* Gl = &Gs[buf][lhalf * 64 + wn * 32 + l31];
#pragma unroll
    for (int s = 0; s < BKR / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Al[s * 2 * AST], Gl[s * 128], acc, 0, 0, 0);
    if (q + 1 < nchunks) store_lds(buf ^ 1);
    __syncthreads();
  }
  float* o = partial + (long)blockIdx.x * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
    o[row * 64 + wn * 32 + l31] = acc[r];
  }
}
// Input gradient of the same layer with respect to the SAMPLE channel (the gradient penalty's dD/dx_hat, T:238-241, and the
// generator step's dL/dfake, T:395-408), bf16 storage mode, ndomain 16: g0[b][d][h][w] = sum over (tap, o: 2 o + tap = (d,h,w)) of
// sum_c u1[b][o][c] * w1[tap][channel 0][c].  The GEMM path writes the whole column matrix P [rows][64 columns (tap, ci)] as fp32
// (283 MB at 2048 samples, half of its columns -- the condition channel's -- never read) and folds it in a second pass (k_d1_col2im):
// 0.11 + 0.13 ms per call, six calls per iteration of BASELINE configs[2].  Here a workgroup owns one sample: its 539 x 27 products
// (every u1 row against the 27 sample-channel taps, one v_mfma_f32_32x32x16_bf16 pass per 32 rows, A fragments straight from
// global memory) stay in LDS and the 24 x 16 x 16 outputs gather their <= 8 terms from there, in k_d1_col2im's order: the same
// fp32 sums, bit for bit.  LDS: [544 rows][33] floats (row stride 33: the 32 taps of a row and the rows of a register quad fall
// into different banks).  grid: min(B, 2 per CU) persistent workgroups.
#define RD_D1DG_LDS (544 * 33 * 4)
// the outputs of ONE parity class (d, h, w) = (2 d' + PD, 2 h' + PH, 2 w' + PW): an axis of parity 1 takes tap 1 at o = its half
// coordinate, an axis of parity 0 taps 0 (o = half coordinate, < the layer's extent) and 2 (o = half coordinate - 1, >= 0) -- the
// same terms in the same order (td, th, tw ascending) as the scalar loop this replaces, without its divergent branches
template <int PD, int PH, int PW>
__device__ __forceinline__ void rd_d1dg_class(const float* __restrict__ Dl, float* __restrict__ gb, int tid) {
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int i = tid + it * 256;                     // 12 x 8 x 8 half coordinates
    const int dq = i >> 6, hq = (i >> 3) & 7, wq = i & 7;
    float sum = 0.f;
#pragma unroll
    for (int jd = 0; jd < (PD ? 1 : 2); ++jd) {
      const int td = PD ? 1 : 2 * jd, od = PD ? dq : dq - jd;
      const bool vd = od >= 0 && od < 11;
#pragma unroll
      for (int jh = 0; jh < (PH ? 1 : 2); ++jh) {
        const int th = PH ? 1 : 2 * jh, oh = PH ? hq : hq - jh;
        const bool vh = vd && oh >= 0 && oh < 7;
#pragma unroll
        for (int jw = 0; jw < (PW ? 1 : 2); ++jw) {
          const int tw = PW ? 1 : 2 * jw, ow = PW ? wq : wq - jw;
          const bool ok = vh && ow >= 0 && ow < 7;
          const float v = Dl[(ok ? (od * 7 + oh) * 7 + ow : 0) * 33 + (td * 3 + th) * 3 + tw];
          if (ok) sum += v;
        }
      }
    }
    gb[((2 * dq + PD) * 16 + 2 * hq + PH) * 16 + 2 * wq + PW] = sum;
  }
}
__global__ void __launch_bounds__(256, 2)
k_d1_dgrad_sample16(const rd_bf16_t* __restrict__ u1, const float* __restrict__ w1, float* __restrict__ g0, int B) {
  extern __shared__ __attribute__((aligned(16))) float Dl[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  // B operand, once: tap n = l31 (27 used), channels 16 kk + 8 lhalf + e of the sample channel's kernel w1[tap][0][c]
  rd_bf16x8 wf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = l31 < 27 ? w1[(l31 * 2) * 64 + kk * 16 + lhalf * 8 + e] : 0.f;
    const u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
    wf[kk] = __builtin_bit_cast(rd_bf16x8, o);
  }
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const rd_bf16_t* ub = u1 + (long)b * (539 * 64);
    // ---- products: 17 blocks of 32 rows dealt to the four waves (5, 4, 4, 4); rows past 538 are zeros.  All of a wave's rows are
    // requested before the first product (one memory round trip per sample instead of one per block: a first version loaded block
    // by block, 0.10 ms per launch at 2048 samples of which most was load latency)
    u32x4_t a[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = (wave + 4 * i) * 32 + l31;
      const rd_bf16_t* rp = ub + (long)(row < 539 ? row : 0) * 64 + lhalf * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[i][kk] = *(const u32x4_t*)(rp + kk * 16);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int blk = wave + 4 * i;
      if (blk < 17) {
        const bool ok = blk * 32 + l31 < 539;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const u32x4_t z = {0u, 0u, 0u, 0u};
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, ok ? a[i][kk] : z), wf[kk], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Dl[(blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 33 + l31] = acc[r];
      }
    }
    __syncthreads();
    // ---- outputs: voxel (d, h, w), <= 8 terms, in the order of k_d1_col2im; one parity class at a time
    float* gb = g0 + (long)b * 6144;
    rd_d1dg_class<0, 0, 0>(Dl, gb, tid); rd_d1dg_class<0, 0, 1>(Dl, gb, tid); rd_d1dg_class<0, 1, 0>(Dl, gb, tid);
    rd_d1dg_class<0, 1, 1>(Dl, gb, tid); rd_d1dg_class<1, 0, 0>(Dl, gb, tid); rd_d1dg_class<1, 0, 1>(Dl, gb, tid);
    rd_d1dg_class<1, 1, 0>(Dl, gb, tid); rd_d1dg_class<1, 1, 1>(Dl, gb, tid);
    __syncthreads();                                  // the products are read: the next sample may overwrite them
  }
}

// The same for domains larger than 16 x 16 (round 4; the large-domain variant L:286: ndomain 64 -> layer-1 grid 11 x 31 x 31): a work
// item is a TILE of 24 x 16 x 8 input voxels of one sample.  Its outputs need the layer-1 rows o = (i - tap) / 2, i.e. the 11 x 9 x 5
// rows (oh0 .. oh0 + 8) x (ow0 .. ow0 + 4) with oh0 = 8 ti - 1, ow0 = 4 tj - 1 (495 rows, those outside the grid are zeros): their
// products with the 27 sample-channel taps stay in LDS ([512][33] floats, 66 KB: two workgroups per CU) and the tile's 3072 outputs
// gather their <= 8 terms from there in the order of k_d1_col2im.  Against the column GEMM + k_d1_col2im (ndomain 64, 64 samples:
// 346 MB of fp32 column matrix written and read again, 0.071 + 0.095 ms per call, six calls per iteration) a call reads u1 1.4 times
// (the overlap of the tiles' row sets) and writes the 1.6 MB result.
#define RD_D1DT_LDS (512 * 33 * 4)
template <int PD, int PH, int PW>
__device__ __forceinline__ void rd_d1dt_class(const float* __restrict__ Dl, float* __restrict__ gb, int tid, int nd, int O, int oh0, int ow0,
                                              int h0, int w0) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = tid + it * 256;                     // 12 x 8 x 4 half coordinates of the tile
    if (i >= 384) break;
    const int dq = i >> 5, hq = (i >> 2) & 7, wq = i & 3;
    float sum = 0.f;
#pragma unroll
    for (int jd = 0; jd < (PD ? 1 : 2); ++jd) {
      const int td = PD ? 1 : 2 * jd, od = PD ? dq : dq - jd;
      const bool vd = od >= 0 && od < 11;
#pragma unroll
      for (int jh = 0; jh < (PH ? 1 : 2); ++jh) {
        const int th = PH ? 1 : 2 * jh, loh = PH ? hq + 1 : hq + 1 - jh;           // local row: o_h - oh0, always in [0, 8]
        const bool vh = vd && (unsigned)(oh0 + loh) < (unsigned)O;
#pragma unroll
        for (int jw = 0; jw < (PW ? 1 : 2); ++jw) {
          const int tw = PW ? 1 : 2 * jw, low = PW ? wq + 1 : wq + 1 - jw;
          const bool ok = vh && (unsigned)(ow0 + low) < (unsigned)O;
          const float v = Dl[(ok ? (od * 9 + loh) * 5 + low : 0) * 33 + (td * 3 + th) * 3 + tw];
          if (ok) sum += v;
        }
      }
    }
    gb[((long)(2 * dq + PD) * nd + h0 + 2 * hq + PH) * nd + w0 + 2 * wq + PW] = sum;
  }
}
// u1 [B][11][O][O][64] bf16 (O = nd / 2 - 1), w1 [27][2][64] fp32 (channel 0 used), g0 [B][24][nd][nd]; nd % 16 == 0.
// grid: persistent workgroups of 256 threads over B * (nd / 16) * (nd / 8) tiles; dynamic LDS RD_D1DT_LDS.
__global__ void __launch_bounds__(256, 2)
k_d1_dgrad_tile16(const rd_bf16_t* __restrict__ u1, const float* __restrict__ w1, float* __restrict__ g0, int B, int nd) {
  extern __shared__ __attribute__((aligned(16))) float Dl[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int O = nd / 2 - 1, TI = nd / 16, TJ = nd / 8;
  rd_bf16x8 wf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = l31 < 27 ? w1[(l31 * 2) * 64 + kk * 16 + lhalf * 8 + e] : 0.f;
    const u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
    wf[kk] = __builtin_bit_cast(rd_bf16x8, o);
  }
  const int nitem = B * TI * TJ;
  for (int item = blockIdx.x; item < nitem; item += gridDim.x) {
    const int tj = item % TJ, ti = (item / TJ) % TI, b = item / (TJ * TI);
    const int oh0 = 8 * ti - 1, ow0 = 4 * tj - 1;
    const rd_bf16_t* ub = u1 + (long)b * 11 * O * O * 64;
    // ---- products: 16 blocks of 32 rows, four per wave, all requested before the first product (one memory round trip per tile)
    u32x4_t a[4][4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave + 4 * i) * 32 + l31;                  // (od * 9 + loh) * 5 + low
      const int od = row / 45, rr = row - od * 45, loh = rr / 5, low = rr - loh * 5;
      const int oh = oh0 + loh, ow = ow0 + low;
      ok[i] = row < 495 && (unsigned)oh < (unsigned)O && (unsigned)ow < (unsigned)O;
      const rd_bf16_t* rp = ub + (ok[i] ? ((long)(od * O + oh) * O + ow) * 64 : 0) + lhalf * 8;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[i][kk] = *(const u32x4_t*)(rp + kk * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = wave + 4 * i;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, ok[i] ? a[i][kk] : z), wf[kk], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Dl[(blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf) * 33 + l31] = acc[r];
    }
    __syncthreads();
    float* gb = g0 + (long)b * 24 * nd * nd;
    const int h0 = 16 * ti, w0 = 8 * tj;
    rd_d1dt_class<0, 0, 0>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0); rd_d1dt_class<0, 0, 1>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0);
    rd_d1dt_class<0, 1, 0>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0); rd_d1dt_class<0, 1, 1>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0);
    rd_d1dt_class<1, 0, 0>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0); rd_d1dt_class<1, 0, 1>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0);
    rd_d1dt_class<1, 1, 0>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0); rd_d1dt_class<1, 1, 1>(Dl, gb, tid, nd, O, oh0, ow0, h0, w0);
    __syncthreads();                                  // the products are read: the next tile may overwrite them
  }
}

// The same weight gradient in the bf16 storage mode, on the bf16 matrix pipe (round 3).  k_d1_gemm_wgrad<bf16> multiplies on the
// fp32 pipe (v_mfma_f32_32x32x2f32: 64 cycles per TWO rows of the contraction) and is bound by it -- 0.38 ms at 6144 samples, 0.46
// of the fp32 MFMA roof, for a launch whose bytes (u1 424 MB bf16 + the 2-channel input 302 MB) would pass in 0.15 ms.  Here both
// operands are bf16 images in LDS, position-major as they arrive -- u1 rows by LDS-DMA straight from HBM, im2col rows rounded to
// bf16 on their way from registers (the forward GEMM k_d1_gemm_fwd<bf16> rounds them the same way) -- and both MFMA operands (8
// consecutive positions of one column) are read with ds_read_b64_tr_b16 exactly as in k_wgrad_gemm_ws16: 32 cycles per SIXTEEN
// rows.  Column 54 of the im2col image holds 1 for the first `bias_rows` rows (the real | fake thirds: the penalty third does not
// reach the bias, T:382) and 0 behind: row 54 of the product is the layer's bias gradient -- the column-sum pass over u1
// (k_colsum_partial, 0.11 ms beside the GEMMs) is gone.  partial[blockIdx.x][64][64], folded by k_d1_wgrad_fold in a fixed order.
// LDS per stage: im2col [64 rows][128 B] + u1 [64 rows][128 B]; 16-byte chunk c of row r at c ^ (((r >> 1) & 1) << 2) (rd_tr_swz<128>).
__global__ void __launch_bounds__(256, 4)
k_d1_wgrad16(const float* __restrict__ cin, const rd_bf16_t* __restrict__ u1, float* __restrict__ partial, long rows,
             long rows_per_wg, long bias_rows, int nd, int Do, int Ho, int Wo) {
  constexpr int BKR = 64;                   // rows per chunk: four 16-deep MFMA steps
  constexpr int IMG = BKR * 128;            // bytes per image
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * IMG];     // [stage][im2col | u1]
  __shared__ long rowoff[2][BKR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const long mbeg = (long)blockIdx.x * rows_per_wg, mend = min(rows, mbeg + rows_per_wg);
  const int NPOS = Do * Ho * Wo, HoWo = Ho * Wo;
  const long nin = (long)RDGAN_NHOURS * nd * nd * 2;
  const int nchunks = mend > mbeg ? (int)((mend - mbeg + BKR - 1) / BKR) : 0;
  auto decode_rows = [&](int q) {          // element offset of each row's window in cin (-1 = no row): threads 0 .. 63
    if (tid < BKR) {
      const long m = mbeg + (long)q * BKR + tid;
      long off = -1;
      if (m < mend) {
        const long b = m / NPOS;
        const int p = (int)(m - b * NPOS);
        const int od = p / HoWo, qq = p - od * HoWo, oh = qq / Wo, ow = qq - oh * Wo;
        off = b * nin + ((long)(2 * od * nd + 2 * oh) * nd + 2 * ow) * 2;
      }
      rowoff[q & 1][tid] = off;
    }
  };
  constexpr int NSG = (BKR * 9 + 255) / 256;          // im2col segments per thread (3; the last one only for tid < 64)
  float2 sa[NSG][3];
  const __amdgpu_buffer_rsrc_t rsG = rd_make_rsrc((const float*)(u1 + mbeg * 64));
  auto load_regs = [&](int q) {
#pragma unroll
    for (int u = 0; u < NSG; ++u) {
      const int sgi = tid + u * 256;
      const int sg = sgi >> 6, r = sgi & (BKR - 1);           // lanes along the rows of one segment
      float2 a = {0.f, 0.f}, c = a, e = a;
      const long off = sgi < BKR * 9 ? rowoff[q & 1][r] : -1;
      if (off >= 0) {
        const int kd = sg / 3, kh = sg - kd * 3;
        const float* src = cin + off + (kd * nd + kh) * nd * 2;
        a = *(const float2*)src; c = *(const float2*)(src + 2); e = *(const float2*)(src + 4);
      }
      sa[u][0] = a; sa[u][1] = c; sa[u][2] = e;
    }
  };
  auto dma_u1 = [&](int q, int buf) {                 // 64 rows x 128 B: 8 DMA instructions of 8 rows, 2 per wave
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = wave * 2 + k;
      const int r = i * 8 + (lane >> 3);
      const long m = (long)q * BKR + r;               // relative to mbeg
      const int cl = (lane & 7) ^ rd_tr_swz<128>(r);
      unsigned voff = mbeg + m < mend ? (unsigned)(m * 128 + cl * 16) : RD_OOB;
      asm volatile("" : "+v"(voff));
      rd_lds_dma16(rsG, (float*)(lds + buf * 2 * IMG + IMG + i * 1024), (int)voff, 0);
    }
  };
  auto store_lds = [&](int q, int buf) {
    char* Xs = lds + buf * 2 * IMG;
#pragma unroll
    for (int u = 0; u < NSG; ++u) {
      const int sgi = tid + u * 256;
      if (sgi < BKR * 9) {
        const int sg = sgi >> 6, r = sgi & (BKR - 1), k0 = sg * 6;
        char* rowp = Xs + r * 128;
        const int sw = rd_tr_swz<128>(r);
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const int k = k0 + 2 * e;
          *(unsigned*)(rowp + (((k >> 3) ^ sw) * 16) + (k & 7) * 2) = rd_pack_bf16(sa[u][e].x, sa[u][e].y);
        }
      }
    }
    if (tid < BKR) {                                  // columns 54 .. 63: the ones column of the bias gradient, then zeros
      const int r = tid;
      const long m = mbeg + (long)q * BKR + r;
      char* rowp = Xs + r * 128;
      const int sw = rd_tr_swz<128>(r);
      *(unsigned*)(rowp + ((6 ^ sw) * 16) + 12) = (m < mend && m < bias_rows) ? 0x00003F80u : 0u;      // k = 54 (1.0), 55
      *(u32x4_t*)(rowp + ((7 ^ sw) * 16)) = (u32x4_t){0u, 0u, 0u, 0u};                                   // k = 56 .. 63
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // transposed-read addresses (k_wgrad_gemm_ws16): this lane is lane 4 q4 + p4 of 16-lane group g in half lhalf
  const int g = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int ckA = (wm * 32) / 8 + 2 * g + (p4 >> 1), ckB = (wn * 32) / 8 + 2 * g + (p4 >> 1);
  const int a_off = (8 * lhalf + q4) * 128 + ((ckA ^ rd_tr_swz<128>(q4)) * 16) + (p4 & 1) * 8;
  const int b_off = IMG + (8 * lhalf + q4) * 128 + ((ckB ^ rd_tr_swz<128>(q4)) * 16) + (p4 & 1) * 8;
  decode_rows(0); decode_rows(1);
  __syncthreads();
  if (nchunks > 0) { dma_u1(0, 0); load_regs(0); store_lds(0, 0); }
  rd_dma_landed();
  __syncthreads();
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
    if (q + 1 < nchunks) { dma_u1(q + 1, buf ^ 1); load_regs(q + 1); }
    decode_rows(q + 2);                                // slot q & 1: last read by load_regs(q), a barrier ago
    const char* st = lds + buf * 2 * IMG;
#pragma unroll
    for (int kk = 0; kk < BKR / 16; ++kk) {
      const rd_bf16x8 fa = rd_tr_frag(st, a_off + kk * 16 * 128, a_off + (kk * 16 + 4) * 128);
      const rd_bf16x8 fb = rd_tr_frag(st, b_off + kk * 16 * 128, b_off + (kk * 16 + 4) * 128);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    if (q + 1 < nchunks) store_lds(q + 1, buf ^ 1);
    rd_dma_landed();
    __syncthreads();
  }
  float* o = partial + (long)blockIdx.x * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
    o[row * 64 + wn * 32 + l31] = acc[r];
  }
}
// dW1[i] = sum over workgroups of partial[g][i], i < 54 * 64 (fixed order)
// (block = 16 outputs x 16 slices of the workgroup range, folded through LDS in a fixed order)
// (db != nullptr, grid 55 * 4 blocks: row 54 of the product = the layer's bias gradient, k_d1_wgrad16)
__global__ void __launch_bounds__(256)
k_d1_wgrad_fold(const float* __restrict__ partial, int G, float* __restrict__ dw, float* __restrict__ db = nullptr) {
  __shared__ float red[256];
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
  const int n = db ? 55 * 64 : 54 * 64;
  float s = 0.f;
  if (i < n)
    for (int g = sl; g < G; g += 16) s += partial[(long)g * 4096 + i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    float t = red[threadIdx.x];
    for (int j = 1; j < 16; ++j) t += red[j * 16 + threadIdx.x];
    if (i < 54 * 64) dw[i] = t; else db[i - 54 * 64] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The generator's Dense layer (T:326-327 / L:333-334: Dense(256 * s * s * 3) + LeakyReLU) for SMALL batches in the bf16 storage
// mode (round 4).  At ndomain 64 the layer is a 64-row GEMM against a 4224 x 49152 kernel: 26 GFLOP behind 415 MB of bf16
// weights, read six times per iteration (five critic steps + the generator step) -- HBM-bound, and as three launches of the
// tiled producer/consumer GEMM (one 128 x 64 tile per workgroup, 256 workgroups per launch) it streamed them at 2.0 TB/s.
// k_dense16_skinny: no LDS, no barriers; a WAVE owns 32 output columns, keeps all rows' accumulators (RB blocks of 32 samples) and
// streams its weights and the input rows through registers in MFMA-fragment order -- both images are stored so that one wave
// instruction reads 1 KB contiguous (k_dense_wimg, k_concat16f) -- two sets of four k-steps in flight.  Operands swapped as in the
// slab kernels (weights = A): a lane ends up with 16 channels of one sample, bias + LeakyReLU + bf16 rounding in registers.
// Weight image: [n-block = N / 32][k-step = KP / 16][lane][8 bf16] = W[16 ks + 8 (lane >> 5) + e][32 nb + (lane & 31)] (zero for
// k >= K); input image: [k-step][row block][lane][8 bf16] = x[32 rb + (lane & 31)][16 ks + 8 (lane >> 5) + e] (zero rows / columns).
// ------------------------------------------------------------------------------------------------------------------------------
// One thread per (n-block, k-step, k half, group of 4 columns): eight 16-byte loads (rows k0 .. k0 + 7, columns n .. n + 3) and the four
// lanes' 16-byte outputs, which are 64 contiguous bytes of the image (a first version read 4 bytes per lane and instruction: the 825 MB
// kernel of ndomain 64 came in at 3.8 TB/s)
__global__ void k_dense_wimg(const float* __restrict__ W /* [K][N] */, unsigned short* __restrict__ img, int K, int N, int KS) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;       // (nb, ks, half, q)
  if (idx >= (long)(N / 32) * KS * 16) return;
  const int q = (int)(idx & 7), half = (int)((idx >> 3) & 1);
  const long t = idx >> 4;
  const int ks = (int)(t % KS), nb = (int)(t / KS);
  const int n = nb * 32 + q * 4, k0 = ks * 16 + half * 8;
  f32x4 v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = k0 + e < K ? *(const f32x4*)(W + (long)(k0 + e) * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned short* o = img + ((t * 64) + half * 32 + q * 4) * 8;       // lane = half * 32 + (n & 31)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32x4_t w = {rd_pack_bf16(v[0][j], v[1][j]), rd_pack_bf16(v[2][j], v[3][j]), rd_pack_bf16(v[4][j], v[5][j]), rd_pack_bf16(v[6][j], v[7][j])};
    *(u32x4_t*)(o + j * 8) = w;
  }
}
// the rows [z | cond] of k_concat as that input image, RB row blocks (rows >= B and columns >= nz + nc are zero)
__global__ void k_concat16f(const float* __restrict__ z, const float* __restrict__ cond, unsigned short* __restrict__ img, int B, int nz,
                            int nc, int KS, int RB) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;       // (ks, rb, lane)
  if (idx >= (long)KS * RB * 64) return;
  const int lane = (int)(idx & 63);
  const int rb = (int)((idx >> 6) % RB), ks = (int)((idx >> 6) / RB);
  const int b = rb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    v[e] = b < B ? (k < nz ? z[(long)b * nz + k] : (k < nz + nc ? cond[(long)b * nc + (k - nz)] : 0.f)) : 0.f;
  }
  u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
  *(u32x4_t*)(img + idx * 8) = o;
}

// out [B][N] bf16 = LeakyReLU(x W + bias); B <= 32 RB, KS = k-steps (a multiple of 8), N a multiple of 32; grid = N / 32
// workgroups of TWO waves that split the k-steps of their 32 columns in halves (the upper half's accumulators cross through LDS at
// the end): 1536 workgroups at ndomain 64 deal evenly to 256 CUs, and twelve waves per CU keep 48 KB of weight loads in flight
// per CU (one wave per 32 columns: 24 KB, 3.3 TB/s -- the launch was latency-bound, not bandwidth-bound).
template <int RB>
__global__ void __launch_bounds__(128)
k_dense16_skinny(const unsigned short* __restrict__ ximg, const unsigned short* __restrict__ wimg, const float* __restrict__ bias,
                 rd_bf16_t* __restrict__ out, int B, int KS, int N) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhalf = lane >> 5;
  __shared__ f32x16 xch[RB][64];
  const int nb = blockIdx.x;
  const int kw = threadIdx.x >> 6;                                           // which half of the k-steps
  const int KH = KS >> 1, kbeg = kw * KH;
  const u32x4_t* wp = (const u32x4_t*)wimg + ((long)nb * KS + kbeg) * 64 + lane;      // + 64 per k-step
  const u32x4_t* xp = (const u32x4_t*)ximg + (long)kbeg * RB * 64 + lane;             // + 64 RB per k-step
  f32x16 acc[RB];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    // register r of a block = channel 32 nb + 8 (r >> 2) + 4 lhalf + (r & 3): the lower half's accumulators start at the bias
    f32x4 b4 = *(const f32x4*)(bias + nb * 32 + 8 * g + 4 * lhalf);
    if (kw) b4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { acc[rb][4 * g] = b4.x; acc[rb][4 * g + 1] = b4.y; acc[rb][4 * g + 2] = b4.z; acc[rb][4 * g + 3] = b4.w; }
  }
  u32x4_t wq[2][4], xq[2][4][RB];
  auto fetch = [&](int set, int ks0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wq[set][u] = __builtin_nontemporal_load(wp + (long)(ks0 + u) * 64);       // read once: keep the L2 for the input rows
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xq[set][u][rb] = xp[((long)(ks0 + u) * RB + rb) * 64];
    }
  };
  auto mul = [&](int set) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, wq[set][u]),
                                                          __builtin_bit_cast(rd_bf16x8, xq[set][u][rb]), acc[rb], 0, 0, 0);
  };
  fetch(0, 0);
  for (int ks = 0; ks < KH; ks += 8) {
    if (ks + 4 < KH) fetch(1, ks + 4);
    mul(0);
    if (ks + 4 >= KH) break;
    if (ks + 8 < KH) fetch(0, ks + 8);
    mul(1);
  }
  if (kw) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xch[rb][lane] = acc[rb];
  }
  __syncthreads();
  if (kw) return;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] += xch[rb][lane];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int b = rb * 32 + l31;
    if (b >= B) continue;
    rd_bf16_t* o = out + (long)b * N + nb * 32 + 4 * lhalf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rd_lrelu(acc[rb][4 * g + e]);
      const u32x2_t pk = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3])};
      *(u32x2_t*)(o + 8 * g) = pk;
    }
  }
}
