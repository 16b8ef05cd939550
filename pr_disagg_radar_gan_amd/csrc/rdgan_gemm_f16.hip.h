// The gather GEMM of the bf16 storage mode in the structure of the slab kernels (round 4).
//
// k_conv_gemm_ws<.., BF> (rdgan_gemm_ws.hip.h) moves BOTH operands through LDS behind loader waves and a barrier per 64-k chunk:
// DESIGN.md 4.5 measured its parts (prologue, fill, MFMA loop, epilogue) to ADD rather than overlap, and named the way out -- "compute
// waves that fetch their own weight fragments into registers with deep vmcnt pipelining" -- which the slab kernels (4.6 ff.) then
// took for the layers whose operand fits LDS.  This is the same structure for the launches that stay GATHERED (row tables, tap
// masks, border boxes, any stride):
//   * 256 x 128 tiles, four waves, NO loader waves: wave (wm, wn) owns 128 rows x 64 columns = 4 x 2 MFMA tiles (128 accumulator
//     registers; two workgroups per CU at <= 256 VGPRs);
//   * the gathered rows (A) go global -> LDS by DMA as before (128-byte rows, same swizzle), issued by the compute waves themselves
//     one chunk ahead: a chunk is 32 MFMAs per wave between barriers instead of 16, so the fill has twice the cover;
//   * the weights (B) never touch LDS: each wave streams ITS fragments global -> VGPR from an image in fragment order
//     (rd_wfrag_index: 1 KB per (32 columns, 16 k), one coalesced dwordx4 per lane), a queue of four k-steps = one chunk ahead,
//     counted vmcnt waits, inline asm (see rd_upc_wload in rdgan_upconv16.hip.h for why);
//   * the epilogue is per WAVE: each wave turns its four 32 x 64 accumulator tiles through 8 KB of LDS of its own (no block
//     barrier) and applies bias / LeakyReLU / dropout / gate / PixelNorm on rows, 4 rows x 128 contiguous bytes per load and
//     store instruction; the destination offsets of its rows were read in the prologue.
// Per MFMA the kernel reads 0.5 KB of row fragments from LDS (the streaming kernel: 1 KB) and writes 0.25 KB of DMA into it (0.5 KB).
// Same chunk order (tap group, channel chunk, tap), same k order inside a chunk as the streaming kernel: the accumulators see the
// same sequence of products, results are bit-identical except through PixelNorm (another order of the 128 squares).
// Everything the plans describe is kept: phases, interleave, boxes, split-K partial slabs (k_splitk_finish), bf16 destinations.
// Not here (the host falls back to the streaming kernel): fp32 destinations, the shared-centre T term, N % 128 != 0.
// (Measured and not kept: s_setprio 3 around prologue and epilogue as in k_conv_gemm_ws -- no launch faster, critic layer 2's
// forward 3 % slower, scratch/f16_ab.py.)
#pragma once

// (rd_wfrag_index, the element order of a fragment-order image: rdgan_plan.h)
// [T][N][K] bf16 (the streaming kernel's weight image) -> the same tap blocks in fragment order; one thread per 8 k (16 bytes)
__global__ void k_wfrag_image(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, long T, int N, int K) {
  const long per = (long)N * K / 8, total = T * per;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i / per, r = i - t * per;
    const int n = (int)(r / (K / 8)), k = (int)(r - (long)n * (K / 8)) * 8;
    *(u32x4_t*)(out + t * N * K + rd_wfrag_index(n, k, K)) = *(const u32x4_t*)(in + t * N * K + (long)n * K + k);
  }
}

// two weight fragments (N blocks j = 0, 1 of the wave's 64 columns) of one k-step: `base` wave-uniform, v0 / v1 = lane * 16 (+ the
// distance between the two N blocks).  s_nop 4: see rd_upc_wload.
__device__ __forceinline__ void rd_f16_wload(u32x4_t& d0, u32x4_t& d1, const char* base, unsigned v0, unsigned v1) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4"
               : "=&v"(d0), "=&v"(d1) : "v"(v0), "v"(v1), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void rd_f16_wait(u32x4_t& d0, u32x4_t& d1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(d0), "+v"(d1) : "i"(N));
}
// The barrier that publishes a chunk's DMA'd rows: all but the N youngest vector-memory operations of this wave (the weight loads
// queued behind the DMAs) have completed.  One asm statement, not rd_dma_landed() + __syncthreads(): the fence inside
// __syncthreads() makes hipcc wait vmcnt(0) -- it knows the DMAs write LDS, not that eight register loads it cannot see follow
// them -- and the weight queue would drain at every chunk.  The "memory" clobber keeps every LDS access on its side of the barrier;
// a wave's own fragment reads of the stage about to be overwritten have all been consumed by MFMAs in front of it.
template <int N>
__device__ __forceinline__ void rd_f16_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(N) : "memory"); }

#define RD_F16_BM 256
#define RD_F16_BN 128
#define RD_F16_STAGE (RD_F16_BM * 128)                   // bytes per A stage: 256 rows of 64 bf16
#define RD_F16_LDS (2 * RD_F16_STAGE + 3 * 2048)         // + destination offsets and 1/l2 of the rows, the PixelNorm exchange (256 rows x 2 halves)

template <bool PN>
__global__ void __launch_bounds__(256, 2)
k_conv_gemm_f16(const RdPlan* __restrict__ plan, int B, const rd_bf16_t* __restrict__ src, const rd_bf16_t* __restrict__ wfrag,
                rd_bf16_t* dst, RdEpi epi, int tg) {
  constexpr int BM = RD_F16_BM, BN = RD_F16_BN;
  extern __shared__ __attribute__((aligned(16))) char f16_lds[];
  char* const lds = f16_lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhalf = lane >> 5;

  // ---- which phase / tile (wave-uniform; as k_conv_gemm_ws)
  const int ntn_log2 = __builtin_ctz(plan->N >> 7);
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
    for (int p = 0; p < nph; ++p) {
      const int nt = (B * plan->phL[p] + BM - 1) / BM;
      if (mt < nt) { pidx = p; break; }
      mt -= nt;
    }
  }
  const RdPhase& P = plan->ph[pidx];
  const int L = P.L;
  const int rows = B * L;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int b0 = m0 / L, l0 = m0 - b0 * L;
  const int SC = plan->SC, wrpt = plan->w_rows_per_tap, N = plan->N;
  const int ssample = (int)plan->src_sample;
  const int ntaps = P.ntaps;
  const RdRowTab tab = rd_row_tab(plan, P.tab);
  const int CPT = SC >> 6;
  const int nch_all = ntaps * CPT;
  const int ksplit = epi.ksplit > 1 ? epi.ksplit : 1;
  int q0 = 0, nchunks = nch_all;
  if (ksplit > 1) {
    const int per_split = (nch_all + ksplit - 1) / ksplit;
    q0 = (int)blockIdx.y * per_split;
    nchunks = max(0, min(nch_all, q0 + per_split) - q0);
  }

  // ---- the taps of the phase, one per lane (read back with v_readlane: no scalar load inside the K loop, whose lgkmcnt the
  // fragment reads share)
  int tp_mask = 0, tp_delta = 0, tp_w = 0;
  if (lane < ntaps) {
    const RdTap ti = P.tap[lane];
    tp_mask = ti.mask; tp_delta = ti.delta >> 1; tp_w = ti.w * wrpt * N * 2;     // (plan deltas are fp32 byte offsets)
  }
  // ---- the 64 rows this wave gathers: DMA k fills rows wave*64 + 8k .. +7; this lane: row +(lane >> 3), physical 16-byte chunk lane & 7
  const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc((const float*)(src + (long)b0 * plan->src_sample));
  int roff[8], rbits[8];
  {
    int rl[8], rb_[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = wave * 64 + k * 8 + (lane >> 3);
      int l = l0 + r, bb = 0;
      if (L >= BM) { if (l >= L) { l -= L; bb = 1; } }
      else { bb = l / L; l -= bb * L; }
      rl[k] = m0 + r < rows ? l : 0;
      rb_[k] = bb;
    }
    int ex[8], ey[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { ex[k] = tab[rl[k]].x; ey[k] = tab[rl[k]].y; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = wave * 64 + k * 8 + (lane >> 3);
      const int c_log = (lane & 7) ^ ((r >> 1) & 7);
      const bool ok = m0 + r < rows;
      roff[k] = ok ? (rb_[k] * ssample + ex[k] + c_log * 8) * 2 : 0;
      rbits[k] = ok ? ey[k] : 0;
    }
  }
  // ---- where this lane's four accumulator rows (wm*128 + i*32 + l31) go: read here so that no memory round trip stands between
  // the last MFMA and the epilogue.  Element offset in the window based at sample b0; RD_OOB: no such row
  const int dsample = (int)plan->dst_sample;
  unsigned rb[4];
  {
    int rl[4], rbb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 128 + i * 32 + l31;
      int l = l0 + row, bb = 0;
      if (L >= BM) { if (l >= L) { l -= L; bb = 1; } }
      else { bb = l / L; l -= bb * L; }
      rl[i] = m0 + row < rows ? l : 0;
      rbb[i] = bb;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int z = tab[rl[i]].z;
      rb[i] = m0 + wm * 128 + i * 32 + l31 < rows ? (unsigned)(rbb[i] * dsample + z) : RD_OOB;
    }
  }
  // ---- weights: this wave's two N blocks of 32 columns; a tap block holds N / 32 x wrpt / 16 fragments of 1 KB
  const char* const wbase = (const char*)(wfrag + P.w_off) + (long)((n0 + wn * 64) >> 5) * (wrpt >> 4) * 1024;
  const unsigned wv0 = (unsigned)lane * 16u, wv1 = wv0 + (unsigned)(wrpt >> 4) * 1024u;

  // ---- the chunk walk: (tap group g of `tg` taps, channel chunk cc, tap t of the group), t fastest
  int n_g = 0, n_cc = 0, n_t = 0, n_gt = min(tg, ntaps);
  if (q0 != 0) {
    const int full = tg * CPT;
    n_g = q0 / full;
    const int rem = q0 - n_g * full;
    n_gt = min(tg, ntaps - n_g * tg);
    n_cc = n_gt > 0 ? rem / n_gt : 0;
    n_t = n_gt > 0 ? rem - n_cc * n_gt : 0;
  }
  auto advance = [&]() {
    if (++n_t == n_gt) {
      n_t = 0;
      if (++n_cc == CPT) { n_cc = 0; ++n_g; n_gt = min(tg, ntaps - n_g * tg); }
    }
  };
  auto issue_rows = [&](int stage) {            // the chunk (n_g, n_cc, n_t): 8 DMAs of 1 KB
    const int tap = n_g * tg + n_t;
    const int tm = __builtin_amdgcn_readlane(tp_mask, tap), td = __builtin_amdgcn_readlane(tp_delta, tap);
    char* As = lds + stage * RD_F16_STAGE + wave * 64 * 128;
#ifdef RD_F16_ABL_NODMA              // (diagnostic builds, scratch/f16_abl.py: the kernel without its row gather; vmcnt bookkeeping kept by dummies)
#pragma unroll
    for (int k = 0; k < 8; ++k) rd_lds_dma16(rsA, (float*)(As + k * 1024), (int)RD_OOB, 0);
    return;
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsigned voff = ((rbits[k] & tm) == tm) ? (unsigned)(roff[k] + td) : RD_OOB;
      rd_lds_dma16(rsA, (float*)(As + k * 1024), (int)voff, n_cc * 128);
    }
  };
  auto wchunk_base = [&]() -> const char* {     // first fragment of the chunk (n_g, n_cc, n_t) for this wave
    const int tap = n_g * tg + n_t;
    return wbase + __builtin_amdgcn_readlane(tp_w, tap) + n_cc * 4096;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4_t wq[4][2];

  const int a_sw = (l31 >> 1) & 7;
  const char* const arow = lds + (wm * 128 + l31) * 128;

  auto chunk = [&](int q, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    const char* As = arow + (q & 1) * RD_F16_STAGE;
    const char* wb = nullptr;
    if constexpr (!LAST) {
      advance();
      issue_rows((q + 1) & 1);
      wb = wchunk_base();
    }
    u32x4_t afr[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) afr[0][i] = *(const u32x4_t*)(As + i * 32 * 128 + ((lhalf ^ a_sw) << 4));
    auto kstep = [&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      if constexpr (kk + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          afr[(kk + 1) & 1][i] = *(const u32x4_t*)(As + i * 32 * 128 + ((((kk + 1) * 2 + lhalf) ^ a_sw) << 4));
      }
      // in flight, oldest first: this chunk's k-steps kk .. 3 (2 loads each), then -- unless LAST -- the next chunk's 8 DMAs and its
      // k-steps 0 .. kk-1: 14 loads behind the two waited for; LAST: 2 (3 - kk)
      if constexpr (LAST) rd_f16_wait<2 * (3 - kk)>(wq[kk][0], wq[kk][1]);
      else rd_f16_wait<14>(wq[kk][0], wq[kk][1]);
#ifdef RD_F16_ABL_NOMFMA             // (diagnostic build: loads and waits only)
      if (kk == 0) { acc[0][0][0] += __builtin_bit_cast(float, wq[kk][0].x ^ afr[kk & 1][0].x); }
      if constexpr (false)
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, wq[kk][0]),
                                                            __builtin_bit_cast(rd_bf16x8, afr[kk & 1][i]), acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, wq[kk][1]),
                                                            __builtin_bit_cast(rd_bf16x8, afr[kk & 1][i]), acc[i][1], 0, 0, 0);
      }
      if constexpr (!LAST) rd_f16_wload(wq[kk][0], wq[kk][1], wb + kk * 1024, wv0, wv1);
    };
    kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{});
    kstep(std::integral_constant<int, 2>{}); kstep(std::integral_constant<int, 3>{});
    if constexpr (!LAST) {
      rd_f16_barrier<8>();                      // the next chunk's rows have landed (its 8 weight loads are younger)
    }
  };
  if (nchunks > 0) {
    issue_rows(0);
    {
      const char* wb = wchunk_base();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) rd_f16_wload(wq[kk][0], wq[kk][1], wb + kk * 1024, wv0, wv1);
    }
    rd_f16_barrier<8>();                        // the 8 DMAs are older than the 8 weight loads
    // (one block with the prologue: a separate `if (nchunks > 0)` around the last chunk gives the loop exit a path around its waits
    // that is never taken, but that scripts/check_isa.py, which does not correlate branches, has to assume)
#pragma unroll 1
    for (int q = 0; q + 1 < nchunks; ++q) chunk(q, std::false_type{});
    chunk(nchunks - 1, std::true_type{});
  }

#ifdef RD_F16_ABL_NOEPI               // (diagnostic build: nothing is stored)
  if (acc[0][0][0] != 12345.678f && acc[3][1][15] != 12345.678f) return;
#endif
  // ---- epilogue.  Accumulator register r of lane (l31, lhalf), tile (i, j): row wm*128 + i*32 + l31, column wn*64 + j*32 + 8 (r >> 2) +
  // 4 lhalf + (r & 3): a lane holds ONE row's columns, so loads and stores straight from this layout touch 32 rows per instruction,
  // 16-32 bytes of each (a first version did: with 64 KB of destination rows per workgroup going through a 32 KB L1 the gate's
  // `aux` reads and the stores cost 0.10 of critic layer 3's 0.19 ms input gradient).  Each wave therefore turns its 32 x 64 tile
  // i through 8 KB of LDS of its own -- in the A stage the last chunk did not use, no barrier -- and works on rows: lane
  // (rq = lane >> 4, cq = lane & 15) takes columns 4 cq .. 4 cq + 3 of rows 4 p + rq, p = 0 .. 7: every load and store instruction
  // covers 4 rows x 128 contiguous bytes, the bias quad is one register quad per lane for the whole tile.
  const long dbase = (long)b0 * dsample;
  unsigned* const rbs = (unsigned*)(lds + 2 * RD_F16_STAGE) + wave * 128;        // [128 rows of the wave] destination offsets
  float* const ris = (float*)(lds + 2 * RD_F16_STAGE + 2048) + wave * 128;       // PixelNorm: 1/l2 of the wave's rows
  if (lhalf == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rbs[i * 32 + l31] = rb[i];
  }
  const int mode = epi.mode;
  const bool has_bias = mode == RD_EPI_BIAS || mode == RD_EPI_BIAS_LRELU || mode == RD_EPI_BIAS_LRELU_DROP || mode == RD_EPI_BIAS_PN_LRELU;
  if constexpr (PN) {
    // PixelNormalization over the row's 128 columns: 32 squares per lane, the other half of the wave, the other column half of the
    // tile (wave wn ^ 1) through LDS
    float* xs = (float*)(lds + 2 * RD_F16_STAGE + 4096);
    const int ncol = n0 + wn * 64 + 4 * lhalf;
    float ss[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
      int nc = ncol;
      asm volatile("" : "+v"(nc));              // (the 8 bias quads are re-read per row block: hoisted, they are 32 registers beside 128 accumulators)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *(const f32x4*)(epi.bias + nc + j * 32 + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[i][j][4 * g + e] + b4[e];      // (added again below: written back, hipcc keeps both copies and spills)
            s += v * v;
          }
        }
      s += __shfl_xor(s, 32, 64);
      ss[i] = s;
      if (lhalf == 0) xs[(wm * 128 + i * 32 + l31) * 2 + wn] = s;
      __builtin_amdgcn_sched_barrier(0);        // (keeps hipcc from clustering all 32 bias loads in front: 41 spilled registers)
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsR = rd_make_rsrc(epi.rinv ? epi.rinv + dbase / BN : (const float*)dst);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 128 + i * 32 + l31;
      const float tot = wn == 0 ? ss[i] + xs[row * 2 + 1] : xs[row * 2] + ss[i];       // (column half 0 first in both waves)
      const float ri = __builtin_amdgcn_rsqf(tot * (1.0f / BN) + 1.0e-8f);
      if (lhalf == 0) ris[i * 32 + l31] = ri;
      if (epi.rinv && wn == 0 && lhalf == 0) rd_buf_store1(rsR, (rb[i] & RD_OOB) | ((rb[i] / BN) * 4u), ri);
    }
  }
  char* const Ts = lds + (nchunks & 1) * RD_F16_STAGE + wave * 8192;             // (nchunks == 0: both stages are free)
  const int rq = lane >> 4, cq = lane & 15;
  const unsigned colq = (unsigned)(n0 + wn * 64 + cq * 4);
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (has_bias && ksplit == 1) bias4 = *(const f32x4*)(epi.bias + colq);
  const __amdgpu_buffer_rsrc_t rsK = rd_make_rsrc(ksplit > 1 ? epi.kpart + (long)blockIdx.y * epi.kstride + dbase : (const float*)dst);
  const __amdgpu_buffer_rsrc_t rsD = rd_make_rsrc((const float*)(dst + dbase));
  const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc(mode == RD_EPI_GATE_AUX ? (const float*)((const rd_bf16_t*)epi.aux + dbase) : (const float*)dst);
  const bool drop = mode == RD_EPI_BIAS_LRELU_DROP && epi.use_drop;
  const uint32_t ibase = (uint32_t)dbase + epi.idx_base + colq;
  // destination offsets of the 8 x 4 rows of tile i and -- gate mode -- their `aux` quads, fetched one tile AHEAD of the stores:
  // hipcc cannot move a load above an earlier store through another buffer descriptor, so a load issued per pass behind the
  // previous pass's store is one memory round trip per pass (32 per tile: 0.10 of critic layer 3's 0.19 ms input gradient)
  unsigned rbv[2][8];
  rd_u32x2 ax[2][8];
  const bool gate = mode == RD_EPI_GATE_AUX && ksplit == 1;
  auto fetch = [&](int i, int slot) {
#pragma unroll
    for (int p = 0; p < 8; ++p) rbv[slot][p] = rbs[i * 32 + p * 4 + rq];
    if (gate) {
#pragma unroll
      for (int p = 0; p < 8; ++p)
        ax[slot][p] = __builtin_bit_cast(rd_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsX, (int)((rbv[slot][p] & RD_OOB) | ((rbv[slot][p] + colq) * 2u)), 0, 0));
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // tile i -> LDS [32 rows][64 columns] fp32; the 16-byte slot s of row r sits at slot s ^ (r & 7) (conflict-free both ways)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        *(f32x4*)(Ts + l31 * 256 + (((j * 8 + 2 * g + lhalf) ^ (l31 & 7)) << 4)) = v;
      }
    if (i + 1 < 4) fetch(i + 1, (i + 1) & 1);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = p * 4 + rq;
      f32x4 v = *(const f32x4*)(Ts + row * 256 + ((cq ^ (row & 7)) << 4));
      const unsigned rbvp = rbv[i & 1][p];
      const unsigned oob = rbvp & RD_OOB;
      if (ksplit > 1) {
        rd_buf_store4(rsK, oob | ((rbvp + colq) * 4u), v);
        continue;
      }
      if constexpr (PN) {
        const float ri = ris[i * 32 + row];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float x = (v[e] + bias4[e]) * ri; v[e] = fmaxf(x, RD_LRELU_ALPHA * x); }
      } else {
        v += bias4;
        if (mode == RD_EPI_BIAS_LRELU || mode == RD_EPI_BIAS_LRELU_DROP) {
          const uint32_t word = drop ? rd_drop_word(epi.key, ibase + rbvp) : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = rd_lrelu(v[e]);
            if (drop) x = rd_drop_apply_w(x, word, e);
            v[e] = x;
          }
        } else if (mode == RD_EPI_GATE_AUX) {
          const f32x4 a4 = rd_unpack_bf16x4(ax[i & 1][p]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= rd_gate_from_out(a4[e], epi.use_drop);
        }
      }
      rd_buf_store4_bf16(rsD, oob | ((rbvp + colq) * 2u), v);
    }
  }
}
