// bf16-operand weight-gradient kernel of the mixed mode ("mfma_bf16"):
//   dW[(tap,c)][n] += sum_m X[m][tap][c] * dY[m][n],   X, dY bf16, fp32 accumulation, fp32 partial slabs
// Same grid, row split and deterministic fold (k_wgrad_reduce) as k_wgrad_gemm_ws.  The contraction runs over the
// gathered positions m, and both LDS images are position-major (a DMA instruction writes whole rows), so both
// v_mfma_f32_32x32x16_bf16 operands -- 8 consecutive positions of ONE channel per lane -- are columns of the images:
// they are read with ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of a 4-row x 16-column block
// (lane 4q+p supplies the address of row q, columns 4p..4p+3), two reads per 8-position fragment.
// Rows are 16-byte-chunk swizzled on the SOURCE side of the DMA so that the four rows of a block fall into different
// banks: chunk c of row r sits at c ^ ((r & 3) << 2) (rows of 256 bytes or more) or c ^ (((r >> 1) & 1) << 2) (128-byte rows).
#pragma once
#include "rdgan_gemm_ws.hip.h"

typedef short rd_s16x4 __attribute__((ext_vector_type(4)));
typedef short rd_s16x8 __attribute__((ext_vector_type(8)));

template <int RB>
__device__ __forceinline__ int rd_tr_swz(int row) {      // XOR on the 16-byte chunk index of `row`
  return RB == 128 ? ((row >> 1) & 1) << 2 : (row & 3) << 2;
}
__device__ __forceinline__ rd_bf16x8 rd_tr_frag(const char* lds, int off0, int off1) {
  const rd_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rd_s16x4*)(lds + off0));
  const rd_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rd_s16x4*)(lds + off1));
  return __builtin_bit_cast(rd_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// <256, 128> (round 4, "wgrad_wide", DEFAULT OFF): THREE stages of 48 KB, one workgroup per CU (compute waves: 128 accumulator
// registers).  The idea: the streaming weight gradient keeps one 32 KB stage per workgroup in flight (64 KB per CU) at 512 B per
// MFMA and runs at 0.20-0.29 of the bf16 roof; this shape keeps two stages = 96 KB in flight and needs 384 B per MFMA.  MEASURED
// (scripts/gpu_r04_o.sh, ndomain 64, bs 64): critic layer 2 0.250 -> 0.417 ms, layer 3 0.125 -> 0.195, generator block 2 0.527 -> 1.036
// -- the third time a one-workgroup-per-CU variant of a producer/consumer kernel loses (512 x 64 conv tiles; 4.6: the
// first slab kernel): with one compute wave per SIMD, all of them phase-locked by the chunk barrier, nobody's MFMAs cover anybody's
// transposed fragment reads.  Kept behind the option with its tests (correct: 1e-5 against the oracle); what would have to change is
// the structure, as in k_conv_gemm_f16 (no loader waves, two workgroups per CU).  The loader waves here publish chunk q + 1
// while chunk q + 2's DMAs stay in flight: a counted wait (the row-table loads of chunk q + 3 are issued BEFORE chunk q + 2's
// DMAs, so those DMAs are exactly the youngest NI_A + NI_B vector-memory operations) and a bare s_barrier in one asm statement --
// hipcc's own wait in front of __syncthreads() covers every DMA it has issued.
template <int N>
__device__ __forceinline__ void rd_ws16_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(N) : "memory"); }

template <int BR, int BN>
__global__ void __launch_bounds__(512, (BR == 256 && BN == 128) ? 2 : 4)
k_wgrad_gemm_ws16(const RdPlan* __restrict__ plan, int B, const unsigned short* __restrict__ src,
                  const unsigned short* __restrict__ dy, float* __restrict__ partial, RdWgradTiling T) {
  constexpr int BKP = 64;                                   // positions per chunk (four 16-deep MFMA steps)
  constexpr int WTM = BR / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int RBA = BR * 2, RBB = BN * 2;                 // bytes per position row of the A / B image
  constexpr int STAGE = BKP * RBA + BKP * RBB;              // bytes per LDS stage
  constexpr int A_L = RBA / 16, A_PPI = 64 / A_L;           // lanes per gathered position, positions per DMA instruction
  constexpr int NI_A = BKP / A_PPI / 4;                     // A DMA instructions per loader wave and chunk
  constexpr int B_L = RBB / 16, B_PPI = 64 / B_L;
  constexpr int NI_B = BKP / B_PPI / 4;
  constexpr int NV = A_PPI == 2 ? 2 : 1;                    // distinct (position & 3) patterns of a lane over the instructions
  constexpr int NST = (BR == 256 && BN == 128) ? 3 : 2;     // LDS stages
  static_assert(BR >= 128 && (BR == 128 || BR == 256) && (BN == 64 || BN == 128), "tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = (char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wave < 4;
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int swz = rd_xcd_swizzle(blockIdx.x, gridDim.x);
  int by, bz, rt, ntile, slab;
  if (T.box) {          // border-class boxes: every phase has its own tile and split counts (k_wgrad_gemm_ws)
    rd_wgrad_box_decode(plan, T, B, swz, bz, by, rt, ntile, slab);
  } else {
    const int tiles = T.RT * T.NT;
    const int bx = swz % tiles;
    by = (swz / tiles) % T.nsplit; bz = swz / (tiles * T.nsplit);
    rt = bx / T.NT; ntile = bx - rt * T.NT;
    slab = (bz * T.nsplit + by) * T.RT;
  }
  const RdPhase& P = plan->ph[bz];
  const int L = P.L;
  const int rows = B * L;
  const int n0 = ntile * BN;
  const int mbeg = by * T.rows_per_split;
  const int mend = min(rows, mbeg + T.rows_per_split);
  const int nchunks = (mend - mbeg + BKP - 1) / BKP;
  const int N = plan->N;

  f32x16 acc[TM][TN];

  if (!is_compute) {
    const int wl = wave - 4;
    const int SC = plan->SC;
    const int ssample = (int)plan->src_sample, dsample = (int)plan->dst_sample;
    const RdRowTab tab = rd_row_tab(plan, P.tab);
    const int bb0 = mbeg / L;
    const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc((const float*)(src + (long)bb0 * plan->src_sample));
    const __amdgpu_buffer_rsrc_t rsB = rd_make_rsrc((const float*)(dy + (long)bb0 * plan->dst_sample));
    // this lane fills physical chunk lane % A_L of its position's row; the logical chunk it fetches depends on
    // (position & 3), which takes NV patterns over the instructions k: v = k & (NV - 1)
    int tmask[NV], a_const[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int r3 = (v * A_PPI + lane / A_L) & 3;           // (position & 3): the instruction bases are multiples of 4
      const int cl = (lane % A_L) ^ rd_tr_swz<RBA>(r3);
      int a_tap, a_c;
      rd_wgrad_tile_row(T, BR, rt, cl * 8, a_tap, a_c);
      tmask[v] = 0x7FFF; a_const[v] = 0;                     // the mask never matches when the column is outside the plan
      if (a_tap < P.ntaps && a_c < SC) { const RdTap t = P.tap[a_tap]; tmask[v] = t.mask; a_const[v] = (t.delta >> 1) + a_c * 2; }
    }
    int b_const;
    {
      const int r3 = (lane / B_L) & 3;                       // B_PPI is a multiple of 4: the same pattern for every instruction
      b_const = (n0 + ((lane % B_L) ^ rd_tr_swz<RBB>(r3)) * 8) * 2;
    }
    int ab[NI_A], al[NI_A], gb[NI_B], gl[NI_B];
#pragma unroll
    for (int k = 0; k < NI_A; ++k) {
      int m = mbeg + (wl * NI_A + k) * A_PPI + lane / A_L;
      ab[k] = m / L; al[k] = m - ab[k] * L; ab[k] -= bb0;
    }
#pragma unroll
    for (int j = 0; j < NI_B; ++j) {
      int m = mbeg + (wl * NI_B + j) * B_PPI + lane / B_L;
      gb[j] = m / L; gl[j] = m - gb[j] * L; gb[j] -= bb0;
    }
    int ex[NI_A], ey[NI_A], ez[NI_B];
    auto fetch_rows = [&]() {
#pragma unroll
      for (int k = 0; k < NI_A; ++k) { ex[k] = tab[al[k]].x; ey[k] = tab[al[k]].y; }
#pragma unroll
      for (int j = 0; j < NI_B; ++j) ez[j] = tab[gl[j]].z;
    };
    fetch_rows();
    auto load_chunk = [&](int mb, int stage) {
      float* As = (float*)(lds + stage * STAGE) + wl * NI_A * 256;
      float* Bs = (float*)(lds + stage * STAGE + BKP * RBA) + wl * NI_B * 256;
#pragma unroll
      for (int k = 0; k < NI_A; ++k) {
        const int v = k & (NV - 1);
        const int m = mb + (wl * NI_A + k) * A_PPI + lane / A_L;
        const int off = (ab[k] * ssample + ex[k]) * 2 + a_const[v];
        unsigned voff = (m < mend && (ey[k] & tmask[v]) == tmask[v]) ? (unsigned)off : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rsA, As + k * 256, (int)voff, 0);
      }
#pragma unroll
      for (int j = 0; j < NI_B; ++j) {
        const int m = mb + (wl * NI_B + j) * B_PPI + lane / B_L;
        unsigned voff = m < mend ? (unsigned)((gb[j] * dsample + ez[j]) * 2 + b_const) : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rsB, Bs + j * 256, (int)voff, 0);
      }
#pragma unroll
      for (int k = 0; k < NI_A; ++k) {
        al[k] += BKP;
        if (L >= BKP) { if (al[k] >= L) { al[k] -= L; ab[k] += 1; } }
        else { int qd = al[k] / L; al[k] -= qd * L; ab[k] += qd; }
      }
#pragma unroll
      for (int j = 0; j < NI_B; ++j) {
        gl[j] += BKP;
        if (L >= BKP) { if (gl[j] >= L) { gl[j] -= L; gb[j] += 1; } }
        else { int qd = gl[j] / L; gl[j] -= qd * L; gb[j] += qd; }
      }
      fetch_rows();
    };
    if constexpr (NST == 3) {
      // two sets of row-table registers: set c & 1 holds chunk c's entries (and its sample indices) from the moment chunk c - 1's
      // DMAs have been issued
      int sx[2][NI_A], sy[2][NI_A], sab[2][NI_A], sz[2][NI_B], sgb[2][NI_B];
      auto fetch = [&](int set) {                        // entries of the cursor's chunk -> set; the cursor moves on
#pragma unroll
        for (int k = 0; k < NI_A; ++k) { sx[set][k] = tab[al[k]].x; sy[set][k] = tab[al[k]].y; sab[set][k] = ab[k]; }
#pragma unroll
        for (int j = 0; j < NI_B; ++j) { sz[set][j] = tab[gl[j]].z; sgb[set][j] = gb[j]; }
#pragma unroll
        for (int k = 0; k < NI_A; ++k) {
          al[k] += BKP;
          if (L >= BKP) { if (al[k] >= L) { al[k] -= L; ab[k] += 1; } }
          else { int qd = al[k] / L; al[k] -= qd * L; ab[k] += qd; }
        }
#pragma unroll
        for (int j = 0; j < NI_B; ++j) {
          gl[j] += BKP;
          if (L >= BKP) { if (gl[j] >= L) { gl[j] -= L; gb[j] += 1; } }
          else { int qd = gl[j] / L; gl[j] -= qd * L; gb[j] += qd; }
        }
      };
      auto dma = [&](int c, int set) {                   // chunk c from set -> stage c % 3: NI_A + NI_B instructions
        const int mb = mbeg + c * BKP, stage = c % 3;
        float* As = (float*)(lds + stage * STAGE) + wl * NI_A * 256;
        float* Bs = (float*)(lds + stage * STAGE + BKP * RBA) + wl * NI_B * 256;
#pragma unroll
        for (int k = 0; k < NI_A; ++k) {
          const int v = k & (NV - 1);
          const int m = mb + (wl * NI_A + k) * A_PPI + lane / A_L;
          const int off = (sab[set][k] * ssample + sx[set][k]) * 2 + a_const[v];
          unsigned voff = (m < mend && (sy[set][k] & tmask[v]) == tmask[v]) ? (unsigned)off : RD_OOB;
          asm volatile("" : "+v"(voff));
          rd_lds_dma16(rsA, As + k * 256, (int)voff, 0);
        }
#pragma unroll
        for (int j = 0; j < NI_B; ++j) {
          const int m = mb + (wl * NI_B + j) * B_PPI + lane / B_L;
          unsigned voff = m < mend ? (unsigned)((sgb[set][j] * dsample + sz[set][j]) * 2 + b_const) : RD_OOB;
          asm volatile("" : "+v"(voff));
          rd_lds_dma16(rsB, Bs + j * 256, (int)voff, 0);
        }
      };
      // (fetch_rows() above left chunk 0's entries in ex / ey / ez: not used here -- the sets are filled from the cursor, which
      // still points at chunk 0)
      fetch(0); fetch(1);
      if (nchunks > 0) dma(0, 0);
      fetch(0);                                          // chunk 2
      if (nchunks > 1) { dma(1, 1); rd_ws16_barrier<NI_A + NI_B>(); }        // chunk 0 has landed; chunk 1 may be in flight
      else rd_ws16_barrier<0>();
      for (int q = 0; q < nchunks; ++q) {
        if (q + 2 < nchunks) {
          fetch((q + 3) & 1);                            // chunk q + 3 (its set was last read by chunk q + 1's DMAs)
          dma(q + 2, q & 1);
          rd_ws16_barrier<NI_A + NI_B>();                // chunk q + 1 has landed
        } else rd_ws16_barrier<0>();
      }
    } else {
    if (nchunks > 0) load_chunk(mbeg, 0);
    __syncthreads();
    for (int q = 0; q < nchunks; ++q) {
      if (q + 1 < nchunks) load_chunk(mbeg + (q + 1) * BKP, (q + 1) & 1);
      __syncthreads();
    }
    }
  } else {
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transposed-read addresses: this lane is lane 4q+p of 16-lane group g (columns 16g.. of the 32-column block) in half h
    const int g = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ck = (wm * WTM + i * 32) / 8 + 2 * g + (p4 >> 1);
      a_off[i] = (8 * lhalf + q4) * RBA + ((ck ^ rd_tr_swz<RBA>(q4)) * 16) + (p4 & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ck = (wn * WTN + j * 32) / 8 + 2 * g + (p4 >> 1);
      b_off[j] = BKP * RBA + (8 * lhalf + q4) * RBB + ((ck ^ rd_tr_swz<RBB>(q4)) * 16) + (p4 & 1) * 8;
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    for (int q = 0; q < nchunks; ++q) {
      const char* st = lds + (NST == 3 ? q % 3 : q & 1) * STAGE;
      rd_bf16x8 fa[2][TM], fb[2][TN];
      auto load_frag = [&](int slot, int kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = rd_tr_frag(st, a_off[i] + kk * 16 * RBA, a_off[i] + (kk * 16 + 4) * RBA);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = rd_tr_frag(st, b_off[j] + kk * 16 * RBB, b_off[j] + (kk * 16 + 4) * RBB);
      };
      load_frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < BKP / 16; ++kk) {
        const int cur = kk & 1;
        if (kk + 1 < BKP / 16) load_frag(cur ^ 1, kk + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // ---- partial tile [BR][BN] through LDS, coalesced float4 stores by all 8 waves
  float* Cs = smem;
  if (is_compute) {
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
#pragma unroll
        for (int j = 0; j < TN; ++j) Cs[row * BN + wn * WTN + j * 32 + l31] = acc[i][j][r];
      }
  }
  __syncthreads();
  float* out = partial + ((long)slab + rt) * BR * N;
  constexpr int F4R = BN / 4, RPP = 512 / F4R;
  const int c4 = (tid % F4R) * 4;
  for (int row = tid / F4R; row < BR; row += RPP)
    *(f32x4*)(out + (long)row * N + n0 + c4) = *(const f32x4*)&Cs[row * BN + c4];
}
