// Wave-specialised (producer / consumer) variant of k_conv_gemm for the big, "clean" plans
// (SC % 32 == 0, no folded-upsample shift): 512-thread workgroups, waves 0-3 multiply, waves 4-7 load.
//
// Why: in-kernel stamps on k_conv_gemm show that next to a partner wave streaming 64-cycle fp32 MFMAs every
// non-MFMA instruction of a wave waits ~55 cycles for an issue slot, so a wave that alternates between a load
// phase (~60 instructions) and an MFMA phase leaves the matrix pipe idle ~25 % of the time.  Here a compute
// wave issues only ds_read + MFMA, and a loader wave streams the next K chunk global -> LDS with LDS-DMA
// (buffer_load ... lds, 16 B per lane): no VGPR staging, no ds_write, one instruction per KiB.  Out-of-image taps
// use an out-of-range buffer offset, for which the DMA writes zeros (verified on gfx950: scratch/probe/ldsdma.hip).
//
// LDS images are the ones of k_conv_gemm (A: unpadded 128-byte rows, 16-byte chunk c of row r stored at
// c ^ ((r>>1)&7); B: [k][BN]).  LDS-DMA writes lane-linearly, so the swizzle is applied on the SOURCE side: the
// lane that fills physical chunk p of row r fetches logical chunk p ^ ((r>>1)&7).
#pragma once
#include "rdgan_gemm.hip.h"

#ifndef RD_WGRAD_SCHED
#define RD_WGRAD_SCHED 0
#endif
#ifndef RD_CONV_SCHED
#define RD_CONV_SCHED 0
#endif

#ifdef RD_STAMP
__device__ unsigned long long rd_stamp_ws[8];   // diagnostic build only: cumulative s_memtime marks of k_conv_gemm_ws (scratch/stamp_ws.py)
#endif
// 16 bytes per lane, global -> LDS, no VGPR destination: LDS address = wave-uniform `lds` + lane*16
__device__ __forceinline__ void rd_lds_dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// A wave's LDS-DMA transfers have landed once its vmcnt is zero.  Waited for EXPLICITLY in front of every barrier that publishes
// DMA'd bytes to other waves: hipcc's own wait in front of __syncthreads() is derived from the issue order it sees on ONE path
// into the barrier (round 3: in k_g9_wgrad_mfma's loop -- nine plain loads, then the DMAs -- it emitted the prologue's
// `s_waitcnt vmcnt(9)`, DMAs first, at the loop head; a wave whose lanes all skip the staging stores then read the previous
// tile's rows: run-to-run differences of 1e-3 in the last conv's weight gradient at ndomain 64 / 128).  The weight-gradient
// loaders keep hipcc's own counted wait: there the DMAs are the OLDER operations on every path into the barrier (the next chunk's
// row-table loads follow them), its `vmcnt(N)` is right (checked in the ISA), and a full drain in front of the barrier, or the
// row-table loads moved in front of the DMAs, cost 2-20 % (measured, scratch/ab_libs.py).
__device__ __forceinline__ void rd_dma_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// BF = true: bf16 operands, fp32 accumulation and output (v_mfma_f32_32x32x16_bf16).  `src` then points at bf16
// activations in the same NDHWC layout and `W` at bf16 weights stored [tap block][N][K per tap] (K contiguous, so both
// operands are 16-byte fragments); a K chunk is 64 elements, i.e. the same 128-byte LDS rows, DMA pattern and swizzle as
// the fp32 kernel, and B uses A's row image.  Plans need SC % 64 == 0.  Everything else (tile decode, row tables, tap
// masks, epilogue) is shared.
// BF kernels are the GEMMs of the bf16 storage mode: with epi.out16 the destination and the tensors the epilogue reads
// (epi.aux, epi.addt) are bf16 too, rounded once from the fp32 accumulator after the whole epilogue (bias, T, PixelNorm,
// LeakyReLU, dropout, gating); split-K partial slabs stay fp32.  out16 = 0 (fp32 destination) serves the op-level parity
// tests and the column GEMM of the critic's input gradient.
typedef __bf16 rd_bf16x8 __attribute__((ext_vector_type(8)));
// S3 ("split3", optional data point, never the headline): fp32 operands in memory and LDS exactly as in the fp32 kernel, but the
// products run on the bf16 matrix pipe: each fp32 value is split in registers into three bf16 parts x = x1 + x2 + x3 (round to
// nearest even at each step: 3 x 8 significant bits, the whole fp32 mantissa) and a product x*y is summed from the six partial
// products x1y1 + x1y2 + x2y1 + x1y3 + x2y2 + x3y1 (each exact in fp32; the dropped ones are below 2^-24 of the product), fp32
// accumulation: 6 x 32 matrix-pipe cycles per 16 k instead of 8 x 64.  Loader waves, LDS images, tap masks, epilogue: unchanged.
typedef float rd_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rd_bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rd_split3(const float* x, u32x4_t& p1, u32x4_t& p2, u32x4_t& p3) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const rd_f32x2 v = {x[2 * i], x[2 * i + 1]};
    const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rd_bf16x2v));
    const rd_f32x2 v1 = {__builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xFFFF0000u)};
    const rd_f32x2 r = v - v1;
    const unsigned u2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, rd_bf16x2v));
    const rd_f32x2 v2 = {__builtin_bit_cast(float, u2 << 16), __builtin_bit_cast(float, u2 & 0xFFFF0000u)};
    const rd_f32x2 t = r - v2;
    p1[i] = u1; p2[i] = u2; p3[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, rd_bf16x2v));
  }
}
// NAMETAG only gives a launch its own kernel symbol (the dominant launch, so that per-name profiler statistics
// describe exactly that launch); it does not change the code.
// RES (bf16 kernels, plans whose taps of a phase are (h,w)-shifted views of the same source planes: the forward GEMMs of the
// shared-centre form): the tile's gathered rows stay RESIDENT in LDS -- each source row is fetched once per channel chunk
// instead of once per tap and channel chunk (64 KB instead of 256 KB per 256x64 tile of generator block 3) -- and only the
// weights stream through the two stages.  A compute lane reads its fragment for tap t from LDS row r + dh*SW + dw (the
// source planes hold whole (h,w) planes, H*W divides BM) and zeroes it where the tap falls outside the image.  With the K loop
// down to 8 chunks the streaming form is bound by what a CU can pull from L2 into LDS; this one is not.
template <int BM, int BN, int WM, int WN, int TG, bool BF = false, int NAMETAG = 0, bool RES = false, bool S3 = false>
__global__ void __launch_bounds__(512, 4)
k_conv_gemm_ws(const RdPlan* __restrict__ plan, int B, const float* __restrict__ src,
               const float* __restrict__ W, int ldw, float* dst, RdEpi epi) {
  static_assert(WM * WN == 4, "4 compute waves");
  static_assert(!RES || BF, "resident-tile mode: bf16 kernels only");
  static_assert(!S3 || (!BF && !RES), "split mode: fp32 operands in memory and in LDS");
  constexpr int BK = 32;
  static_assert(TG == 4 || TG == 8, "taps per register group");
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int STAGE = BM * BK + BK * BN;                 // floats per LDS stage
  constexpr int NI_A = BM / 32;                            // A DMA instructions (1 KiB = 8 rows) per loader wave and chunk
  constexpr int NI_B = BN / 32;                            // B DMA instructions (1 KiB = 256/BN rows) per loader wave
  constexpr int B_LPR = BN / 4;                            // lanes per B row
  constexpr int B_RPI = 64 / B_LPR;                        // B rows per DMA instruction
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wave < 4;
  const int l31 = lane & 31, lhalf = lane >> 5;
  // everything outside the MFMA loop runs at raised priority: beside the other resident workgroup's MFMA stream a
  // priority-0 wave gets about one issue slot per 64-cycle MFMA
  __builtin_amdgcn_s_setprio(3);
#ifdef RD_STAMP
  const unsigned long long ws_t0 = rd_stamp();
  const bool ws_st = BM == 256 && epi.addt != nullptr && lane == 0;
#endif

  // ---- which phase / tile (wave-uniform)
  // (every instruction in front of the first DMA delays the first MFMA by ~50 cycles beside the other resident
  // workgroup's MFMA stream, so the decode avoids integer divisions: N / BN and the phase count of an interleaved plan
  // are powers of two)
  constexpr int BN_LOG2 = BN == 128 ? 7 : 6;
  static_assert(BN == 128 || BN == 64, "BN");
  const int ntn_log2 = __builtin_ctz(plan->N >> BN_LOG2);
  // phases of unequal length (stride-2 input gradients: 8, 4, 4, 2, 4, 2, 2, 1 taps) are laid out one after the other:
  // a contiguous range per XCD would hand one XCD all the 8-tap tiles, so those plans keep the round-robin dealing
  const int swz = (plan->nphases > 1 && !plan->interleave) ? (int)blockIdx.x : rd_xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = swz & ((1 << ntn_log2) - 1);
  int mt = swz >> ntn_log2, pidx = 0;
  const int nph = plan->nphases;
  if (plan->interleave && (nph & (nph - 1)) == 0) {
    pidx = mt & (nph - 1);
    mt >>= __builtin_ctz(nph);
  } else if (plan->interleave) {
    pidx = mt % nph;
    mt /= nph;
  } else {
    for (int p = 0; p < plan->nphases; ++p) {
      int nt = (B * plan->phL[p] + BM - 1) / BM;
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
  const int SC = plan->SC, wrpt = plan->w_rows_per_tap;
  const int ssample = (int)plan->src_sample;
  const int ntaps = P.ntaps;
  const RdRowTab tab = rd_row_tab(plan, P.tab);
  constexpr int ESZ = BF ? 2 : 4;                        // operand element size
  constexpr int KCH = BK * 4 / ESZ;                      // elements of K per chunk (one 128-byte row)
  const int CPT = SC / KCH;
  const int nch_all = ntaps * CPT;
  const int ksplit = epi.ksplit > 1 ? epi.ksplit : 1;
  int q0 = 0, nchunks = nch_all;
  if (ksplit > 1) {
    const int per_split = (nch_all + ksplit - 1) / ksplit;
    q0 = (int)blockIdx.y * per_split;
    nchunks = max(0, min(nch_all, q0 + per_split) - q0);
  }

  f32x16 acc[TM][TN];

  if (!is_compute) {
    // =============================== loader waves ===============================
    const int wl = wave - 4;
    const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc((const float*)((const char*)src + (long)b0 * plan->src_sample * ESZ));
    const __amdgpu_buffer_rsrc_t rsB = rd_make_rsrc((const float*)((const char*)W + (long)P.w_off * ESZ));
    // A: instruction k of this wave fills rows wl*(BM/4) + 8k .. +7; this lane: row +(lane>>3), physical chunk lane&7
    // (branch-free, so that the NI_A row-table loads go out back to back and are waited for once: with a branch per
    // row hipcc serialises them, one memory round trip each, and the first DMA leaves ~10 us late)
    int roff[NI_A], rbits[NI_A];
    {
      int rl[NI_A], rb_[NI_A];
#pragma unroll
      for (int k = 0; k < NI_A; ++k) {
        const int r = wl * (BM / 4) + k * 8 + (lane >> 3);
        int l = l0 + r, bb = 0;
        if (L >= BM) { if (l >= L) { l -= L; bb = 1; } }
        else { bb = l / L; l -= bb * L; }
        rl[k] = m0 + r < rows ? l : 0;
        rb_[k] = bb;
      }
      int ex[NI_A], ey[NI_A];
#pragma unroll
      for (int k = 0; k < NI_A; ++k) { ex[k] = tab[rl[k]].x; ey[k] = tab[rl[k]].y; }
#pragma unroll
      for (int k = 0; k < NI_A; ++k) {
        const int r = wl * (BM / 4) + k * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((r >> 1) & 7);
        const bool ok = m0 + r < rows;
        roff[k] = ok ? (rb_[k] * ssample + ex[k] + c_log * (16 / ESZ)) * ESZ : 0;
        rbits[k] = ok ? ey[k] : 0;
      }
    }
    int boff[NI_B];
#pragma unroll
    for (int j = 0; j < NI_B; ++j) {
      if constexpr (BF) {      // B image = BN rows (n) of 128 bytes (64 k), swizzled like A; instruction j of this wave fills 8 rows
        const int n = (wl * NI_B + j) * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((n >> 1) & 7);
        boff[j] = ((n0 + n) * wrpt + c_log * 8) * 2;
      } else {
        const int kk = (wl * NI_B + j) * B_RPI + lane / B_LPR;
        boff[j] = (kk * ldw + n0 + (lane % B_LPR) * 4) * 4;
      }
    }
    // gather offsets of the current group of TG taps (TG = 4 for the 4-tap plans of the shared-centre form: half the
    // mask/select work in front of the first DMA)
    unsigned voffs[NI_A][TG];
    int tapw[TG];
    auto build_group = [&](int g) {
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        const int tap = min(g * TG + t, ntaps - 1);
        const RdTap ti = P.tap[tap];
        tapw[t] = BF ? ti.w * wrpt * plan->N * 2 : ti.w * wrpt * ldw * 4;
        if constexpr (!RES) {
          const int tdelta = BF ? ti.delta >> 1 : ti.delta;     // plan deltas are fp32 byte offsets
#pragma unroll
          for (int k = 0; k < NI_A; ++k)
            voffs[k][t] = ((rbits[k] & ti.mask) == ti.mask) ? (unsigned)(roff[k] + tdelta) : RD_OOB;
        }
      }
    };
    // RES: the un-shifted source rows of the tile, one 128-byte row per tile row and channel chunk, at the hour plane the
    // phase's taps share (all taps of a phase have the same d offset); rows whose plane lies outside the tensor read zeros
    unsigned resoff[NI_A];
    if constexpr (RES) {
      const RdTap t0 = P.tap[0];
      const int d_off = (t0.code & 255) / 2 - 1;
      const int pl_delta = d_off * plan->SH * plan->SW * plan->s_cstride * ESZ;
      const int dmask = t0.mask & 0xF;
#pragma unroll
      for (int k = 0; k < NI_A; ++k) {
        const int r = wl * (BM / 4) + k * 8 + (lane >> 3);
        resoff[k] = (m0 + r < rows && (rbits[k] & dmask) == dmask) ? (unsigned)(roff[k] + pl_delta) : RD_OOB;
      }
    }
    auto load_resident = [&](int cc) {         // channel chunk cc of every tile row -> A_res[cc]
      float* Ar = smem + cc * (BM * BK) + wl * (BM / 4) * BK;
#pragma unroll
      for (int k = 0; k < NI_A; ++k) rd_lds_dma16(rsA, Ar + k * 8 * BK, (int)resoff[k], cc * BK * 4);
    };
    int ld_g = 0, ld_cc = 0, ld_t = 0, ld_gt = min(TG, ntaps);
    if (q0 != 0) {
      const int full = TG * CPT;
      ld_g = q0 / full;
      const int rem = q0 - ld_g * full;
      ld_gt = min(TG, ntaps - ld_g * TG);
      ld_cc = ld_gt > 0 ? rem / ld_gt : 0;
      ld_t = ld_gt > 0 ? rem - ld_cc * ld_gt : 0;
    }
    if (nchunks > 0) build_group(ld_g);
    auto issue = [&](int stage, auto t_c) {
      constexpr int t = decltype(t_c)::value;
      float* As = smem + stage * STAGE + wl * (BM / 4) * BK;
      float* Bs = RES ? smem + CPT * (BM * BK) + stage * (BK * BN) + wl * NI_B * 256
                      : smem + stage * STAGE + BM * BK + wl * NI_B * 256;
      const int sA = ld_cc * BK * 4;
      const int sB = BF ? tapw[t] + ld_cc * BK * 4 : tapw[t] + ld_cc * BK * ldw * 4;
#ifdef RD_ABL_NODMA                 // diagnostic build only (scratch/abl16.py): the K loop of the bf16 kernels without its loads
      if constexpr (BF) return;
#endif
      if constexpr (!RES) {
#pragma unroll
        for (int k = 0; k < NI_A; ++k)
          rd_lds_dma16(rsA, As + k * 8 * BK, (int)voffs[k][t], sA);
      }
#pragma unroll
      for (int j = 0; j < NI_B; ++j)
        rd_lds_dma16(rsB, Bs + j * 256, boff[j], sB);
    };
    auto load_chunk = [&](int stage) {
      if constexpr (TG == 4) {
        switch (ld_t) {
          case 0: issue(stage, std::integral_constant<int, 0>{}); break;
          case 1: issue(stage, std::integral_constant<int, 1>{}); break;
          case 2: issue(stage, std::integral_constant<int, 2>{}); break;
          default: issue(stage, std::integral_constant<int, 3>{}); break;
        }
      } else {
        switch (ld_t) {
          case 0: issue(stage, std::integral_constant<int, 0>{}); break;
          case 1: issue(stage, std::integral_constant<int, 1>{}); break;
          case 2: issue(stage, std::integral_constant<int, 2>{}); break;
          case 3: issue(stage, std::integral_constant<int, 3>{}); break;
          case 4: issue(stage, std::integral_constant<int, 4>{}); break;
          case 5: issue(stage, std::integral_constant<int, 5>{}); break;
          case 6: issue(stage, std::integral_constant<int, 6>{}); break;
          default: issue(stage, std::integral_constant<int, 7>{}); break;
        }
      }
      if (++ld_t == ld_gt) {
        ld_t = 0;
        if (++ld_cc == CPT) {
          ld_cc = 0;
          ++ld_g;
          ld_gt = min(TG, ntaps - ld_g * TG);
          if (ld_gt > 0) build_group(ld_g);
        }
      }
    };
    if constexpr (RES) { if (nchunks > 0) load_resident(0); }
    if (nchunks > 0) load_chunk(0);
#ifdef RD_STAMP
    if (ws_st && wave == 4) atomicAdd(&rd_stamp_ws[4], rd_stamp() - ws_t0);
#endif
    rd_dma_landed();
    __syncthreads();                                     // chunk 0 has landed
    for (int q = 0; q < nchunks; ++q) {
      // stage (q+1)&1 was last read during chunk q-1, whose closing barrier every wave has passed
      if (q + 1 < nchunks) load_chunk((q + 1) & 1);
      // RES: channel chunk cc is first read by chunk q = cc * ntaps; chunks 1 .. CPT-1 come in behind chunk 0's weights
      if constexpr (RES) { if (q + 1 < CPT) load_resident(q + 1); }
      rd_dma_landed();
      __syncthreads();
    }
  } else {
    // =============================== compute waves ===============================
    const int wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int a_sw = (l31 >> 1) & 7;
    __syncthreads();
#ifdef RD_STAMP
    if (ws_st && wave == 0) atomicAdd(&rd_stamp_ws[0], rd_stamp() - ws_t0);
#endif
    __builtin_amdgcn_s_setprio(0);
    if constexpr (BF && RES) {
      // validity bits of this lane's TM tile rows (row table), and the tap walk of the loaders: chunk q = (cc, t), t fastest
      int rbits_c[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wm * WTM + i * 32 + l31;
        int l = l0 + r;
        if (L >= BM) { if (l >= L) l -= L; }
        else l -= (l / L) * L;
        rbits_c[i] = m0 + r < rows ? tab[l].y : 0;
      }
      const int SWs = plan->SW;
      int cq_t = 0, cq_cc = 0;
      for (int q = 0; q < nchunks; ++q) {
        const int buf = q & 1;
        const RdTap ti = P.tap[cq_t];
        const int h_off = (((ti.code >> 8) & 255) - 6) / 2 - 1, w_off = ((ti.code >> 16) - 12) / 2 - 1;
        const int shift = h_off * SWs + w_off;
        const float* Ar = smem + cq_cc * (BM * BK);
        const float* Bs = smem + CPT * (BM * BK) + buf * (BK * BN) + (wn * WTN + l31) * BK;
        const float* arow[TM];
        int asw[TM];
        unsigned amask[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const bool ok = (rbits_c[i] & ti.mask) == ti.mask;
          const int rp = ok ? wm * WTM + i * 32 + l31 + shift : wm * WTM + i * 32 + l31;   // (a valid tap stays inside the row's plane)
          arow[i] = Ar + rp * BK;
          asw[i] = (rp >> 1) & 7;
          amask[i] = ok ? 0xFFFFFFFFu : 0u;
        }
        f32x4 fa[2][TM], fb[2][TN];
        auto load_frag = [&](int slot, int kk) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            u32x4_t v = *(const u32x4_t*)&arow[i][((kk * 2 + lhalf) ^ asw[i]) * 4];
            v.x &= amask[i]; v.y &= amask[i]; v.z &= amask[i]; v.w &= amask[i];
            fa[slot][i] = __builtin_bit_cast(f32x4, v);
          }
          const int col = ((kk * 2 + lhalf) ^ a_sw) * 4;
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[slot][j] = *(const f32x4*)&Bs[j * 32 * BK + col];
        };
        load_frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int cur = kk & 1;
          if (kk + 1 < 4) load_frag(cur ^ 1, kk + 1);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, fa[cur][i]),
                                                                  __builtin_bit_cast(rd_bf16x8, fb[cur][j]), acc[i][j], 0, 0, 0);
        }
        if (++cq_t == ntaps) { cq_t = 0; ++cq_cc; }
        __syncthreads();
      }
    } else if constexpr (BF) {
      for (int q = 0; q < nchunks; ++q) {
#ifdef RD_ABL_NOMFMA                // diagnostic build only: the K loop without its fragment reads and MFMAs
        __syncthreads();
        continue;
#endif
        const int buf = q & 1;
        const float* As = smem + buf * STAGE + (wm * WTM + l31) * BK;
        const float* Bs = smem + buf * STAGE + BM * BK + (wn * WTN + l31) * BK;
        // k-step kk covers 16 elements = the 16-byte chunks 2kk (lanes 0-31) and 2kk+1 (lanes 32-63) of a row
        f32x4 fa[2][TM], fb[2][TN];
        auto load_frag = [&](int slot, int kk) {
          const int col = ((kk * 2 + lhalf) ^ a_sw) * 4;
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[slot][i] = *(const f32x4*)&As[i * 32 * BK + col];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[slot][j] = *(const f32x4*)&Bs[j * 32 * BK + col];
        };
        load_frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int cur = kk & 1;
          if (kk + 1 < 4) load_frag(cur ^ 1, kk + 1);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, fa[cur][i]),
                                                                  __builtin_bit_cast(rd_bf16x8, fb[cur][j]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
      }
    } else if constexpr (S3) {
      for (int q = 0; q < nchunks; ++q) {
        const int buf = q & 1;
        const float* As = smem + buf * STAGE + (wm * WTM + l31) * BK;
        const float* Bs = smem + buf * STAGE + BM * BK + wn * WTN + l31;
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
          // this lane's 8 k of the 16: k16*16 + lhalf*8 .. +7 (A: two 16-byte chunks of the row; B: eight rows of the [k][BN] image)
          u32x4_t b1[TN], b2[TN], b3[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = Bs[(k16 * 16 + lhalf * 8 + e) * BN + j * 32];
            rd_split3(x, b1[j], b2[j], b3[j]);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const f32x4 lo = *(const f32x4*)&As[i * 32 * BK + (((k16 * 4 + lhalf * 2) ^ a_sw) * 4)];
            const f32x4 hi = *(const f32x4*)&As[i * 32 * BK + (((k16 * 4 + lhalf * 2 + 1) ^ a_sw) * 4)];
            const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            u32x4_t a1, a2, a3;
            rd_split3(x, a1, a2, a3);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#define RD_MF3(A_, B_) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, A_), __builtin_bit_cast(rd_bf16x8, B_), acc[i][j], 0, 0, 0)
              RD_MF3(a3, b1[j]); RD_MF3(a2, b2[j]); RD_MF3(a1, b3[j]); RD_MF3(a2, b1[j]); RD_MF3(a1, b2[j]); RD_MF3(a1, b1[j]);
#undef RD_MF3
            }
          }
        }
        __syncthreads();
      }
    } else
    for (int q = 0; q < nchunks; ++q) {
      const int buf = q & 1;
      const float* As = smem + buf * STAGE + (wm * WTM + l31) * BK;
      const float* Bs = smem + buf * STAGE + BM * BK + lhalf * 4 * BN + wn * WTN + l31;
      constexpr int NJ = BK / 8;
      f32x4 fa[2][TM];
      float fb[2][4][TN];
      auto load_frag = [&](int slot, int j8) {
        const int acol = ((j8 * 2 + lhalf) ^ a_sw) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *(const f32x4*)&As[i * 32 * BK + acol];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[slot][s][j] = Bs[(j8 * 8 + s) * BN + j * 32];
      };
      load_frag(0, 0);
#pragma unroll
      for (int j8 = 0; j8 < NJ; ++j8) {
        const int cur = j8 & 1;
        if (j8 + 1 < NJ) load_frag(cur ^ 1, j8 + 1);
#if RD_CONV_SCHED == 1     // (round 4 experiment, see k_wgrad_gemm_ws: default off)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][s], fb[cur][s][j], acc[i][j], 0, 0, 0);
#if RD_CONV_SCHED == 1
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_setprio(3);
  }

#ifdef RD_STAMP
  const unsigned long long ws_t2 = rd_stamp();
  if (ws_st && wave == 0) { atomicAdd(&rd_stamp_ws[1], ws_t2 - ws_t0); atomicAdd(&rd_stamp_ws[5], 1ull); }
#endif
#ifdef RD_ABL_NOEPI                   // diagnostic build only: no epilogue (nothing is stored)
  if constexpr (BF) { if (acc[0][0][0] != 12345.678f) return; }
#endif
  // ---- epilogue (all 8 waves): tile through LDS, coalesced float4 rows.
  // Next to the other resident workgroup's MFMA stream every instruction here waits for an issue slot (stamps: ~50
  // cycles each), so the row loop is kept short: 32-bit byte offsets inside buffer windows based at sample b0 (rows
  // without a destination carry RD_OOB: their loads return 0 and their stores are dropped, no branch), v_rsq, DPP sums.
  const int dsample = (int)plan->dst_sample;
  const int mode = epi.mode;
  float* Cs = smem;
  unsigned* Rb = (unsigned*)(smem + BM * BN);            // [BM] byte offset of the row in the destination window
  unsigned* Tb = Rb + BM;                                // [BM] byte offset of the row in the epi.addt window
  if (is_compute) {
    const int wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
#pragma unroll
        for (int j = 0; j < TN; ++j) Cs[row * BN + wn * WTN + j * 32 + l31] = acc[i][j][r];
      }
  } else if (tid - 256 < BM) {
    const int row = tid - 256;
    unsigned rb = RD_OOB, tb = RD_OOB;
    if (m0 + row < rows) {
      int l = l0 + row, bb = 0;
      if (L >= BM) { if (l >= L) { l -= L; bb = 1; } }
      else { bb = l / L; l -= bb * L; }
      const int z = tab[l].z;
      rb = (unsigned)(bb * dsample + z) * 4u;      // byte offset of the row as fp32; halved below for a bf16 destination
      if (epi.addt) tb = (unsigned)(bb * (dsample >> 1) + z - ((z / epi.addt_plane + 1) >> 1) * epi.addt_plane) * 4u;
    }
    Rb[row] = rb; Tb[row] = tb;
  }
  __syncthreads();
#ifdef RD_STAMP
  if (ws_st && wave == 0) atomicAdd(&rd_stamp_ws[2], rd_stamp() - ws_t0);
#endif
  constexpr int F4R = BN / 4;
  constexpr int RPP = 512 / F4R;
  const int c4 = (tid % F4R) * 4;
  const unsigned colb = (unsigned)(n0 + c4) * 4u;
  const long dbase = (long)b0 * dsample;
  if (ksplit > 1) {
    const __amdgpu_buffer_rsrc_t rsK = rd_make_rsrc(epi.kpart + (long)blockIdx.y * epi.kstride + dbase);
#pragma unroll 4
    for (int row = tid / F4R; row < BM; row += RPP)
      rd_buf_store4(rsK, Rb[row] + colb, *(const f32x4*)&Cs[row * BN + c4]);
    return;
  }
  const bool has_t = epi.addt != nullptr;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (mode == RD_EPI_BIAS || mode == RD_EPI_BIAS_LRELU || mode == RD_EPI_BIAS_LRELU_DROP || mode == RD_EPI_BIAS_PN_LRELU)
    bias4 = *(const f32x4*)(epi.bias + n0 + c4);
  if constexpr (BF) {
    if (epi.out16) {
      // bf16 destination (and bf16 aux / T): same row loop on 8-byte accesses; the byte offsets of the fp32 layout are halved
      // (an out-of-range marker stays out of range: RD_OOB >> 1 is still far above every window)
      const __amdgpu_buffer_rsrc_t rsD = rd_make_rsrc((const float*)((const rd_bf16_t*)dst + dbase));
      const __amdgpu_buffer_rsrc_t rsT = rd_make_rsrc(has_t ? (const float*)((const rd_bf16_t*)epi.addt + (long)b0 * (dsample >> 1)) : dst);
      const unsigned colh = colb >> 1;
      auto oob = [](unsigned rb) -> unsigned { return (rb >> 1) | (rb & RD_OOB); };
      if (mode == RD_EPI_BIAS_PN_LRELU) {
        const __amdgpu_buffer_rsrc_t rsR = rd_make_rsrc(epi.rinv ? epi.rinv + dbase / BN : dst);
        const unsigned rmask = (c4 == 0 && epi.rinv) ? 0u : RD_OOB;
#pragma unroll 4
        for (int row = tid / F4R; row < BM; row += RPP) {
          const unsigned rb = Rb[row];
          f32x4 v = *(const f32x4*)&Cs[row * BN + c4];
          if (has_t) v += rd_buf_load4_bf16(rsT, oob(Tb[row]) + colh);
          v += bias4;
          const float ss = rd_lanes_sum<F4R>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
          const float ri = __builtin_amdgcn_rsqf(ss * (1.0f / BN) + 1.0e-8f);
          v *= ri;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], RD_LRELU_ALPHA * v[e]);
          rd_buf_store1(rsR, (rb / BN) | (rb & RD_OOB) | rmask, ri);
          rd_buf_store4_bf16(rsD, oob(rb) + colh, v);
        }
      } else {
        const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc(mode == RD_EPI_GATE_AUX ? (const float*)((const rd_bf16_t*)epi.aux + dbase) : dst);
        const uint32_t ibase = (uint32_t)dbase + epi.idx_base + (uint32_t)(n0 + c4);
#pragma unroll 4
        for (int row = tid / F4R; row < BM; row += RPP) {
          const unsigned rb = Rb[row];
          f32x4 v = *(const f32x4*)&Cs[row * BN + c4];
          if (has_t) v += rd_buf_load4_bf16(rsT, oob(Tb[row]) + colh);
          if (mode == RD_EPI_BIAS) {
            v += bias4;
          } else if (mode == RD_EPI_BIAS_LRELU || mode == RD_EPI_BIAS_LRELU_DROP) {
            v += bias4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = rd_lrelu(v[e]);
              if (mode == RD_EPI_BIAS_LRELU_DROP && epi.use_drop) x = rd_drop_apply_w(x, rd_drop_word(epi.key, ibase + (rb >> 2)), e);
              v[e] = x;
            }
          } else if (mode == RD_EPI_GATE_AUX) {
            const f32x4 a4 = rd_buf_load4_bf16(rsX, oob(rb) + colh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float g = rd_gate_from_out(a4[e], epi.use_drop);
              v[e] *= g;
            }
          }
          rd_buf_store4_bf16(rsD, oob(rb) + colh, v);
        }
      }
      return;
    }
  }
  const __amdgpu_buffer_rsrc_t rsD = rd_make_rsrc(dst + dbase);
  const __amdgpu_buffer_rsrc_t rsT = rd_make_rsrc(epi.addt ? epi.addt + (long)b0 * (dsample >> 1) : dst);
  if (mode == RD_EPI_BIAS_PN_LRELU) {
    // PixelNormalization (T:255-266) + LeakyReLU (T:333): the F4R lanes holding a row are an aligned lane group
    const __amdgpu_buffer_rsrc_t rsR = rd_make_rsrc(epi.rinv ? epi.rinv + dbase / BN : dst);
    const unsigned rmask = (c4 == 0 && epi.rinv) ? 0u : RD_OOB;      // one lane per row stores 1/l2
#pragma unroll 4
    for (int row = tid / F4R; row < BM; row += RPP) {
      const unsigned rb = Rb[row];
      f32x4 v = *(const f32x4*)&Cs[row * BN + c4];
      if (has_t) v += rd_buf_load4(rsT, Tb[row] + colb);
      v += bias4;
      const float ss = rd_lanes_sum<F4R>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
      const float ri = __builtin_amdgcn_rsqf(ss * (1.0f / BN) + 1.0e-8f);      // v_rsq_f32: 1 ulp
      v *= ri;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], RD_LRELU_ALPHA * v[e]);     // LeakyReLU for alpha < 1
      rd_buf_store1(rsR, (rb / BN) | (rb & RD_OOB) | rmask, ri);
      rd_buf_store4(rsD, rb + colb, v);
    }
  } else {
    const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc(mode == RD_EPI_GATE_AUX ? epi.aux + dbase : dst);
    const uint32_t ibase = (uint32_t)dbase + epi.idx_base + (uint32_t)(n0 + c4);
#pragma unroll 4
    for (int row = tid / F4R; row < BM; row += RPP) {
      const unsigned rb = Rb[row];
      f32x4 v = *(const f32x4*)&Cs[row * BN + c4];
      if (has_t) v += rd_buf_load4(rsT, Tb[row] + colb);
      if (mode == RD_EPI_BIAS) {
        v += bias4;
      } else if (mode == RD_EPI_BIAS_LRELU || mode == RD_EPI_BIAS_LRELU_DROP) {
        v += bias4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = rd_lrelu(v[e]);
          if (mode == RD_EPI_BIAS_LRELU_DROP && epi.use_drop) x = rd_drop_apply_w(x, rd_drop_word(epi.key, ibase + (rb >> 2)), e);
          v[e] = x;
        }
      } else if (mode == RD_EPI_GATE_AUX) {
        const f32x4 a4 = rd_buf_load4(rsX, rb + colb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = rd_gate_from_out(a4[e], epi.use_drop);
          v[e] *= g;
        }
      }
      rd_buf_store4(rsD, rb + colb, v);
    }
  }
#ifdef RD_STAMP
  if (ws_st && wave == 0) atomicAdd(&rd_stamp_ws[3], rd_stamp() - ws_t0);
#endif
}

// ------------------------------------------------------------------------------------
// Producer / consumer weight-gradient kernel (see k_wgrad_gemm for the maths and the grid):
// waves 0-3 multiply, waves 4-7 stream the next 32 gathered positions (A: [32][BR]) and output-gradient rows
// (B: [32][BN]) into LDS by DMA.  Clean plans only (SC % 4 == 0, no shift).
// ------------------------------------------------------------------------------------
template <int BR, int BN>
__global__ void __launch_bounds__(512, 4)
k_wgrad_gemm_ws(const RdPlan* __restrict__ plan, int B, const float* __restrict__ src,
                const float* __restrict__ dy, float* __restrict__ partial, RdWgradTiling T) {
  constexpr int BKP = 32;
  constexpr int WTM = BR / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int STAGE = BKP * BR + BKP * BN;
  constexpr int A_F4 = BR / 4;                   // lanes per gathered position
  constexpr int A_PPI = A_F4 >= 64 ? 1 : 64 / A_F4;          // positions per A DMA instruction
  constexpr int A_IPP = A_F4 >= 64 ? A_F4 / 64 : 1;          // A DMA instructions per position (BR = 256: 1)
  static_assert(A_IPP == 1, "BR <= 256");
  constexpr int NI_A = BKP / A_PPI / 4;          // A DMA instructions per loader wave and chunk
  constexpr int B_F4 = BN / 4;
  constexpr int B_PPI = 64 / B_F4;
  constexpr int NI_B = BKP / B_PPI / 4;
  constexpr bool UNI = A_PPI == 1;               // a whole wave gathers one position: cursors on the scalar unit
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wave < 4;
  const int l31 = lane & 31, lhalf = lane >> 5;
#ifndef RD_WGRAD_NOPRIO
  // as in k_conv_gemm_ws: everything but the MFMA loop at raised priority (beside the other resident workgroup's MFMA
  // stream a priority-0 loader wave gets about one issue slot per 64-cycle MFMA)
  __builtin_amdgcn_s_setprio(3);
#endif
  const int swz = rd_xcd_swizzle(blockIdx.x, gridDim.x);
  int by, bz, rt, ntile, slab;
  if (T.box) {          // border-class boxes: every phase has its own tile and split counts
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
    const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc(src + (long)bb0 * plan->src_sample);
    const __amdgpu_buffer_rsrc_t rsB = rd_make_rsrc(dy + (long)bb0 * plan->dst_sample);
    // this lane's (tap, channel) column group of the A tile
    int a_tap, a_c;
    rd_wgrad_tile_row(T, BR, rt, (lane % A_F4) * 4, a_tap, a_c);
    const bool a_ok = a_tap < P.ntaps && a_c < SC;
    int tmask = 0x7FFF, a_const = 0;
    if (a_ok) { const RdTap t = P.tap[a_tap]; tmask = t.mask; a_const = t.delta + a_c * 4; }
    const int b_const = (n0 + (lane % B_F4) * 4) * 4;
    // row cursors: A instruction k gathers position (wl*NI_A + k)*A_PPI + lane/A_F4 of the chunk
    int ab[NI_A], al[NI_A], gb[NI_B], gl[NI_B];
#pragma unroll
    for (int k = 0; k < NI_A; ++k) {
      int pos = (wl * NI_A + k) * A_PPI + (UNI ? 0 : lane / A_F4);
      int m = mbeg + pos;
      ab[k] = m / L; al[k] = m - ab[k] * L; ab[k] -= bb0;
    }
#pragma unroll
    for (int j = 0; j < NI_B; ++j) {
      int m = mbeg + (wl * NI_B + j) * B_PPI + lane / B_F4;
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
      float* As = smem + stage * STAGE + wl * NI_A * 256;
      float* Bs = smem + stage * STAGE + BKP * BR + wl * NI_B * 256;
#pragma unroll
      for (int k = 0; k < NI_A; ++k) {
        const int m = mb + (wl * NI_A + k) * A_PPI + (UNI ? 0 : lane / A_F4);
        const int off = (ab[k] * ssample + ex[k]) * 4 + a_const;
        unsigned voff = (m < mend && (ey[k] & tmask) == tmask) ? (unsigned)off : RD_OOB;
#ifdef RD_ABL_L1W
        voff = (unsigned)(lane * 16 + (k & 3) * 1024);    // diagnostic build: every gather hits a 4 KiB window
#endif
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rsA, As + k * 256, (int)voff, 0);
      }
#pragma unroll
      for (int j = 0; j < NI_B; ++j) {
        const int m = mb + (wl * NI_B + j) * B_PPI + lane / B_F4;
        unsigned voff = m < mend ? (unsigned)((gb[j] * dsample + ez[j]) * 4 + b_const) : RD_OOB;
#ifdef RD_ABL_L1W
        voff = (unsigned)(lane * 16 + j * 1024);
#endif
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
    if (nchunks > 0) load_chunk(mbeg, 0);
    __syncthreads();
    for (int q = 0; q < nchunks; ++q) {
      if (q + 1 < nchunks) load_chunk(mbeg + (q + 1) * BKP, (q + 1) & 1);
      __syncthreads();
    }
  } else {
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();
#ifndef RD_WGRAD_NOPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    for (int q = 0; q < nchunks; ++q) {
      const int buf = q & 1;
      const float* As = smem + buf * STAGE + lhalf * BR + wm * WTM + l31;
      const float* Bs = smem + buf * STAGE + BKP * BR + lhalf * BN + wn * WTN + l31;
      constexpr int NG = BKP / 8;
      float fa[2][4][TM], fb[2][4][TN];
      auto load_frag = [&](int slot, int g) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[slot][s][i] = As[(g * 8 + 2 * s) * BR + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[slot][s][j] = Bs[(g * 8 + 2 * s) * BN + j * 32];
        }
      };
      load_frag(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int cur = g & 1;
        if (g + 1 < NG) load_frag(cur ^ 1, g + 1);
        // Round 4, measured and NOT kept (default RD_WGRAD_SCHED 0).  hipcc sinks every fragment read of group g + 1 down to its
        // first use: the ISA is `ds_read2_b32; s_waitcnt lgkmcnt(0); v_mfma; v_mfma` all the way through, two fragment registers in
        // all -- an LDS latency exposed in front of every pair of MFMAs, covered only by the SIMD's other compute wave.  Pinning the
        // reads of the next eight positions in front of this group's sixteen MFMAs (RD_WGRAD_SCHED 1: sched_barrier; 0 of 64 MFMAs
        // then wait for a read, 112 VGPRs) made the <256,64> launches 5-7 % SLOWER (0.419-0.427 -> 0.446-0.453 ms); RD_WGRAD_SCHED 2
        // spreads the same prefetch one read per MFMA (sched_group_barrier).  See DESIGN.md section 6, round 4.
#if RD_WGRAD_SCHED == 1
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s][i], fb[cur][s][j], acc[i][j], 0, 0, 0);
#if RD_WGRAD_SCHED == 1
        __builtin_amdgcn_sched_barrier(0);
#elif RD_WGRAD_SCHED == 2
        {
          constexpr int NDS = 2 * (TM + TN), NMF = 4 * TM * TN, PER = NMF / NDS > 0 ? NMF / NDS : 1;
#pragma unroll
          for (int k = 0; k < NDS; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // one LDS read of the next group
          }
          __builtin_amdgcn_sched_group_barrier(0x008, NMF - PER * NDS > 0 ? NMF - PER * NDS : 0, 0);
        }
#endif
      }
      __syncthreads();
    }
#ifndef RD_WGRAD_NOPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
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
