// bf16 storage mode, generator block 3 forward (T:340-343: UpSampling3D(2) + Conv3D(128 -> 64, 3x3x3, 'same') + PixelNorm +
// LeakyReLU onto the 24 x 16 x 16 grid) in the collapsed form (8 output-parity phases x 8 taps of 128 channels on
// the un-upsampled 12 x 8 x 8 grid) -- the dominant launch of BASELINE configs[2..3] -- as a SLAB kernel (round 3).
//
// Why another kernel.  As a streaming GEMM (k_conv_gemm_ws<256, 64, ..., bf16>) this layer has N = 64: every 256-row x 64-k
// chunk of gathered rows (32 KB) is used for 16 MFMAs per wave and then thrown away, 40 KB pulled from L2 into LDS per 512
// matrix-pipe cycles = 80 B/clk per CU, where a CU takes in 30-50 B/clk: 0.31 of the MFMA roof, and the one-barrier-per-chunk
// lock-step between loader and compute waves adds what is left.  The three re-designs of round 2 kept that
// structure.  This one does not:
//   * A workgroup owns a SLAB: 4 source hour planes of one sample (256 positions x 128 channels) + the two halo planes, resident in
//     LDS (6 x 16 KB) for ALL 8 phases x 8 taps: each source row is fetched once per slab instead of 64 times (96 KB per slab
//     against 4 MB).  A tap is a shifted read of the same rows ((h, w) shifts inside the plane, d shifts = another plane slot);
//     rows outside the picture read a zero row.
//   * Weights never touch LDS: they are stored in HBM in MFMA-fragment order (k_upconv_wimg: 1 KB per wave-instruction, fully
//     coalesced) and every wave streams its own fragments global -> VGPR four k-steps ahead of their use.  No loader waves, no
//     staging, NO barrier inside a tile: the eight waves of a workgroup drift apart, and one wave's epilogue, fragment-read
//     latency or weight wait is covered by its SIMD partner's MFMAs.
//   * Each wave computes a 128-row x 64-channel tile of ONE phase with the operands swapped (D^T = W^T X^T: weights as the
//     MFMA's A operand, activations as B), so a lane ends up with 32 channels of ONE output row and lane ^ 32 with the other 32:
//     bias + PixelNorm (sum of squares: 32 FMAs + one cross-half exchange) + LeakyReLU + bf16 rounding run in registers, and the
//     row goes out in 16-byte stores (v_permlane32_swap pairs).  No LDS round trip, no epilogue barrier.
// Per CU and 512 matrix-pipe cycles the kernel moves 16 KB of weight fragments through L1 (8 waves x 2 KB; 8 KB from L2) and
// reads 64 KB of row fragments from LDS (128 B/clk, half the LDS rate).
//
// LDS image of a plane slot: 64 rows (h*8 + w) of 256 B (128 channels), 16-byte chunk c of row r stored at c ^ (r & 15): the 32
// lanes of a fragment read take 32 consecutive rows (shifted or not), so every 16-lane group of a ds_read_b128 hits 16 different
// bank groups.  The DMA writes lane-linearly, so the swizzle is applied to the SOURCE address (as in k_conv_gemm_ws).
#pragma once
#include "rdgan_gemm_ws.hip.h"

#define RD_UPC_SLOT 16384                 // bytes per plane slot
#define RD_UPC_NSLOT 4                    // plane slots per workgroup: 2 centre planes + 2 halo planes
#define RD_UPC_ZERO (RD_UPC_NSLOT * RD_UPC_SLOT)     // a 256-byte row of zeros: taps that fall outside the (h, w) picture
#define RD_UPC_BIAS (RD_UPC_ZERO + 256)   // 64 floats
#define RD_UPC_LDS (RD_UPC_BIAS + 256)
// fused last conv (G9 = true): per-wave scratch [64 class rows][11: 9 (kh, kw) products + dummy + pad] fp32 behind the bias row
#define RD_UPC_PW RD_UPC_LDS
#define RD_UPC_PW_WAVE (64 * 11 * 4)          // row stride 11 floats: 9 products, a dummy column, odd for the banks
#define RD_UPC_LDS_G9 (RD_UPC_PW + 4 * RD_UPC_PW_WAVE)

// Weight image of the slab kernel from the collapsed forms Wc [64 = phase*8 + tap][128 ci][64 co] (fp32, k_collapse_weights):
// for k-step g = (phase*8 + tap)*8 + j (j = 16-channel step 0..7) and column block nb, lane l holds the 8 bf16
// Wc[phase*8 + tap][16 j + 8 (l >> 5) + e][32 nb + (l & 31)], e = 0..7: the A fragment of v_mfma_f32_32x32x16_bf16, 1 KB per
// (g, nb), 1 MB in all.
__global__ void k_upconv_wimg(const float* __restrict__ Wc, unsigned short* __restrict__ wimg) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                 // (g, nb, lane)
  if (idx >= 64 * 8 * 2 * 64) return;
  const int lane = idx & 63, nb = (idx >> 6) & 1, g = idx >> 7;
  const int pt = g >> 3, j = g & 7;
  const int n = nb * 32 + (lane & 31), k0 = j * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = Wc[((long)pt * 128 + k0 + e) * 64 + n];
  u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
  *(u32x4_t*)(wimg + (long)idx * 8) = o;
}

// Fused last generator conv (T:345, Conv3D 64 -> 1, 3 x 3 x 3 'same'; template flag G9).  The block's output rows are in registers
// as packed bf16 when the epilogue stores them -- lane (l31, lhalf) of row block mb holds channels 32 nb + 8 g + 4 lhalf + 0..3 of
// row l31 -- and the pair of channel groups (nb, 2 gp), (nb, 2 gp + 1) that makes one 16-byte store is exactly one B fragment of
// v_mfma_f32_32x32x16_bf16 if the contraction index is read in that order.  k_g9_wimg stores the 27 x 64 kernel as the matching A
// fragments (k-step ks = 2 nb + gp; lane (MFMA row m = l & 31, half = l >> 5) holds W9[tap(m)][32 nb + 16 gp + 4 half + e] for
// e = 0..3 and W9[tap(m)][32 nb + 16 gp + 8 + 4 half + e - 4] for e = 4..7): four MFMAs per 32-row block (128 for the block itself)
// give the tap products P[m][row] the separate pass over h3 (k_g9_fwd, 1.6 GB read at 2048 samples) recomputed.  MFMA row m <-> tap:
// m = 8 kd + j for the first eight (kh, kw) taps j = 3 kh + kw of hour offset kd, m = 24 + kd for the ninth (2, 2), rows 27..31
// zero -- accumulator register e of lane (l31, half) is row (e & 3) + 8 (e >> 2) + 4 half, so the products of one kd are
// registers 4 kd .. 4 kd + 3 of both halves plus register 12 + kd of the lower half: five unconditional LDS stores per kd.
// The sums over (kh, kw) stay inside an output plane, and a wave's tile covers ONE (h, w) parity class of two planes: per plane
// and kd the wave passes its [64 class rows][9] products through a private 2.3 KB of LDS, and lane L = class position (L >> 3, L & 7)
// of each of the four TARGET classes adds up the 1 / 2 / 2 / 4 products that land on it (a source of class p feeds a target of the
// same class through the centre tap only, a target of the other parity along an axis through the two outer taps of that axis: one
// from its own class position, one from the neighbour's).  The sum over kd is taken as far as the work item reaches: its four planes feed
// six target planes (the two outer ones belong to the neighbouring items), Out: QT[sample][item][6 target planes][source class p][target
// class][64] fp32 -- 24 B per grid point instead of the 128 B of h3; k_tapsum_softmax12 adds the four source classes and the
// neighbouring items' outer planes in a fixed order.  No barrier, no cross-wave traffic.  With ST = false (critic steps: nothing differentiates through the generator)
// h3 and 1/l2 are not stored at all.
__global__ void k_g9_wimg(const float* __restrict__ w9 /* [27][64] */, unsigned short* __restrict__ img /* [4][64][8] */) {
  const int idx = threadIdx.x;                    // (ks, lane)
  const int lane = idx & 63, ks = idx >> 6, nb = ks >> 1, gp = ks & 1;
  const int m = lane & 31, half = lane >> 5;
  const int tap = m < 24 ? 9 * (m >> 3) + (m & 7) : 9 * (m - 24) + 8;        // rows 24, 25, 26: tap (kd, 2, 2)
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 32 * nb + 16 * gp + 8 * (e >> 2) + 4 * half + (e & 3);
    v[e] = m < 27 ? w9[tap * 64 + c] : 0.f;
  }
  u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
  *(u32x4_t*)(img + (long)idx * 8) = o;
}

// A weight fragment (1 KB per wave-instruction): global -> VGPR by an inline-asm load that hipcc does not see, so that it cannot
// re-schedule the software pipeline (its own version of this loop issued every load one k-step in front of its use and waited
// for it with vmcnt(0): the queue of four k-steps was gone) and does not drain the queue with its own waits.  The statement that
// waits (rd_upc_wait) names the destination registers, so no consumer is scheduled above it.  `base` is wave-uniform (SGPR pair),
// `voff` = lane * 16.
__device__ __forceinline__ void rd_upc_wload(u32x4_t& d0, u32x4_t& d1, const char* base, unsigned voff) {
  // (s_nop 4: `base` is an SGPR pair the compiler has just computed with SALU instructions; a vector-memory instruction that
  // reads it needs 5 wait states behind the write, which hipcc pads for its own loads but not for an asm statement -- without
  // the pad the load now and then went out with the previous k-step's address)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
               : "=&v"(d0), "=&v"(d1) : "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void rd_upc_wait(u32x4_t& d0, u32x4_t& d1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(d0), "+v"(d1) : "i"(N));
}

// x [B][12][8][8][128] bf16 -> out [B][24][16][16][64] bf16 = LeakyReLU(PixelNorm(upconv(x) + bias)), rinv [B][24][16][16] = 1/l2.
// Work item = HALF a slab: 2 source hour planes (128 positions) of one sample + their two halo planes (4 plane slots, 64 KB), all
// 8 phases: four waves, wave q computes the 128 rows x 64 channels of phase (pd, q >> 1, q & 1) for pd = 0, then for pd = 1.
// 256-thread workgroups, TWO per CU (66 KB of LDS, <= 256 VGPRs): a first version with one 512-thread workgroup per CU (4 centre
// planes, eight waves) ran every wave of a CU in lock-step -- same work, same start behind the slab barrier -- so all eight
// epilogues (~700 VALU instructions per wave and tile) came together and the matrix pipe idled through them (with no loads, no
// LDS reads and no stores at all that version still needed 0.143 ms against 0.082 ms of MFMAs at bs 256).  Two independent
// workgroups per CU drift apart: one's epilogue, slab load and barriers run beside the other's MFMAs.  (Starting the CUs' second
// workgroups half a tile late on purpose changed nothing: 0.1774 -> 0.1746 ms at the best delay.)
// Where the time goes (diagnostic builds -DRD_UPC_ABL_*, scratch/upc_abl.py; bs 2048, one box): K loops without weight loads,
// LDS reads and epilogue 1.01 ms = 1.65 PFLOP/s -- the matrix pipe at the clock the chip holds under an MFMA-dense loop on random
// data (1.5-1.7 GHz, MI355X_MICROARCH.md DVFS item 5), i.e. the attainable roof is ~0.66 of the nominal 2.5 PFLOP/s; + weight
// stream and fragment reads 1.20; + epilogue arithmetic 1.28; + stores 1.36 = the kernel (0.50 of nominal, 0.74 of attainable).
// grid: min(6 B, 2 CUs) persistent workgroups; dynamic LDS RD_UPC_LDS.  NAMETAG: own symbol for the profiler.
template <int NAMETAG, bool G9 = false, bool ST = true>
__global__ void __launch_bounds__(256, 2)
k_upconv_slab16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ wimg, const float* __restrict__ bias,
                rd_bf16_t* __restrict__ out, float* __restrict__ rinv, int B, float* __restrict__ dbg = nullptr,
                const unsigned short* __restrict__ w9img = nullptr, float* __restrict__ Q12 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int q = wave;                                // phase slot
  const int ph = q >> 1, pw = q & 1;

  if (tid < 64) {
    *(float*)(lds + RD_UPC_BIAS + tid * 4) = bias[tid];
    *(float*)(lds + RD_UPC_ZERO + tid * 4) = 0.f;
  }
  const unsigned wvoff = (unsigned)lane * 16u;

  for (int slab = blockIdx.x; slab < 6 * B; slab += gridDim.x) {
    const int b = slab / 6, d0 = (slab - b * 6) * 2;
    __syncthreads();                                  // every wave has left the previous slab (and the bias / zero rows are in)
    {
      // the four planes d0 - 1 .. d0 + 2 -> slots 0 .. 3: 64 DMA instructions of 1 KB (4 rows), 16 per wave
      const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(x + (long)b * (12 * 64 * 128)));
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int i = wave * 16 + k;                  // wave-uniform
        const int pi = i >> 4, ii = i & 15;
        const int d = d0 - 1 + pi;
        const int row = ii * 4 + (lane >> 4);
        const int c_log = (lane & 15) ^ (row & 15);
        unsigned voff = (unsigned)d < 12u ? (unsigned)(((d * 64 + row) * 256) + c_log * 16) : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rs, (float*)(lds + pi * RD_UPC_SLOT + ii * 1024), (int)voff, 0);
      }
    }
    rd_dma_landed();
    __syncthreads();

#pragma unroll 1
    for (int pd = 0; pd < 2; ++pd) {
      // ---- one tile: the 128 rows of the two centre planes, phase (pd, ph, pw), all 64 channels
      const int g0 = (pd * 4 + ph * 2 + pw) * 64;     // first k-step of the phase in the weight image
      f32x16 acc[4][2];
      {
        // accumulators start at the bias of their channel: register r of block nb = channel 32 nb + 8 (r >> 2) + 4 lhalf + (r & 3)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(const f32x4*)(lds + RD_UPC_BIAS + (nb * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
              acc[mb][nb][4 * g + 0] = b4.x; acc[mb][nb][4 * g + 1] = b4.y;
              acc[mb][nb][4 * g + 2] = b4.z; acc[mb][nb][4 * g + 3] = b4.w;
            }
          }
      }
      // weight fragments: a queue of four k-steps (8 loads in flight per wave), refilled behind the MFMAs that consumed a slot
      const char* wph = (const char*)wimg + (long)g0 * 2048;            // wave-uniform
      u32x4_t bq[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s) rd_upc_wload(bq[s][0], bq[s][1], wph + s * 2048, wvoff);
#pragma unroll 1
      for (int t = 0; t < 8; ++t) {
        // tap t = (td, th, tw): source offsets (pd - 1 + td, ph - 1 + th, pw - 1 + tw)
        const int od = pd - 1 + (t >> 2), oh = ph - 1 + ((t >> 1) & 1), ow = pw - 1 + (t & 1);
        int abase[4], aswz[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const int r = 32 * (mb & 1) + l31;          // row inside the plane: h = r >> 3, w = r & 7
          const bool ok = (unsigned)((r >> 3) + oh) < 8u && (unsigned)((r & 7) + ow) < 8u;
          const int rsft = r + oh * 8 + ow;
          const int slot = (mb >> 1) + 1 + od;
          abase[mb] = ok ? slot * RD_UPC_SLOT + rsft * 256 : RD_UPC_ZERO;
          aswz[mb] = ok ? ((rsft & 15) ^ lhalf) : lhalf;
        }
        u32x4_t afr[2][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) afr[0][mb] = *(const u32x4_t*)(lds + abase[mb] + (aswz[mb] << 4));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#ifndef RD_UPC_ABL_NOA               // (diagnostic builds, scratch/upc_abl.py: the K loop without its LDS fragment reads)
          if (j + 1 < 8) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
              afr[(j + 1) & 1][mb] = *(const u32x4_t*)(lds + abase[mb] + (((2 * (j + 1)) ^ aswz[mb]) << 4));
          }
#else
          if (j + 1 < 8) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) { afr[(j + 1) & 1][mb] = afr[j & 1][mb]; asm volatile("" : "+v"(afr[(j + 1) & 1][mb])); }
          }
#endif
#ifndef RD_UPC_ABL_NOW               // (diagnostic builds: the K loop without its weight stream)
          rd_upc_wait<6>(bq[j & 3][0], bq[j & 3][1]);          // the two oldest of the eight loads in flight
#endif
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[j & 3][0]),
                                                                 __builtin_bit_cast(rd_bf16x8, afr[j & 1][mb]), acc[mb][0], 0, 0, 0);
            acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[j & 3][1]),
                                                                 __builtin_bit_cast(rd_bf16x8, afr[j & 1][mb]), acc[mb][1], 0, 0, 0);
          }
#ifndef RD_UPC_ABL_NOW
          {
            // refill the slot with k-step t*8 + j + 4 of the tile (past the tile's end: its last k-step again, never used)
            const int gn = t * 8 + j + 4;
            rd_upc_wload(bq[j & 3][0], bq[j & 3][1], wph + (long)(gn < 63 ? gn : 63) * 2048, wvoff);
          }
#else
          asm volatile("" : "+v"(bq[j & 3][0]), "+v"(bq[j & 3][1]));
#endif
        }
      }
      // The clamped refills of the last four k-steps are still in flight and nobody will read them: wait for them HERE, naming
      // their destination registers -- to the compiler those registers are dead behind the loop, and without the operands it
      // may hand them to the epilogue's temporaries (or schedule epilogue arithmetic above a bare wait) while the loads land.
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1]), "+v"(bq[3][0]),
                     "+v"(bq[3][1]));
#ifdef RD_UPC_ABL_NOEPI              // (diagnostic builds: K loops only; one element of every accumulator keeps the MFMAs alive)
      {
        float tsum = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) tsum += acc[mb][0][3] + acc[mb][1][7];
        if (tsum == 12345.678f) rinv[0] = tsum;
        continue;
      }
#endif
      // ---- epilogue, in registers: lane (l31, lhalf) of block mb holds 32 channels of output row m = 32 mb + l31 of the wave's
      // 128 rows (channels 32 nb + 8 g + 4 lhalf + 0..3), lane ^ 32 the other 32
      u32x4_t w9f[4];
      if constexpr (G9) {
        // the last conv's A fragments (4 KB image, L1 / L2 resident): loaded here, per tile -- as loop invariants they would sit in
        // 16 registers the K loop does not have
        const unsigned short* wp = w9img;
        asm volatile("" : "+s"(wp));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w9f[ks] = *(const u32x4_t*)(wp + (ks * 64 + lane) * 8);
      }
      f32x16 pacc[2];
      // The item's four planes A0..A3 = 2 d0 .. 2 d0 + 3 (this pass: A_pd and A_(pd+2)) feed six target planes A0 - 1 .. A3 + 1
      // (source plane A_a, hour tap kd -> target plane index tp = a + 2 - kd): QT[sample][item][tp][source class][target class][64].
      // pd = 0 stores its share; pd = 1 adds to the four targets both passes reach (read back here, one round trip in front of the
      // epilogue arithmetic; the same lane wrote them, behind a K loop that ended in vmcnt(0)).  Fixed order per target.
      float hold[4] = {0.f, 0.f, 0.f, 0.f}, rmw[8];
      // (the offset goes through an asm statement so that hipcc computes it HERE, behind the K loop's final wait: as a per-slab
      // invariant it and the store addresses derived from it were kept in ~60 registers across the K loop, i.e. spilled)
      int qtoff = (((b * 6 + (d0 >> 1)) * 6) * 4 + q) * 256 + lane;
      if constexpr (G9) asm volatile("" : "+v"(qtoff));
      float* qt = Q12 + qtoff;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        if constexpr (G9) {
          if ((mb & 1) == 0) {      // the two targets plane mb >> 1 of this pass adds to (tp = 2 (mb >> 1) + 1, + 2): two row blocks ahead
#pragma unroll
            for (int i = 0; i < 8; ++i) rmw[i] = 0.f;
            if (pd == 1) {
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int qi = 0; qi < 4; ++qi)
                  rmw[t * 4 + qi] = qt[(2 * (mb >> 1) + 1 + t) * 1024 + ((ph ^ (qi >> 1)) * 2 + (pw ^ (qi & 1))) * 64];
            }
          }
        }
        // (two-wide float arithmetic -- v_pk_fma_f32 / v_pk_mul_f32, 416 instead of ~700 instructions per tile -- was measured:
        // 3 % SLOWER, scratch/upc_abl.py; packed fp32 issues at half rate beside the partner's MFMAs)
        float ss = 0.f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) ss = fmaf(acc[mb][nb][r], acc[mb][nb][r], ss);
        ss += __shfl_xor(ss, 32, 64);                       // + the other half's 32 channels of the same row (lane ^ 32)
        const float ri = __builtin_amdgcn_rsqf(ss * (1.0f / 64.0f) + 1.0e-8f);       // PixelNormalization (T:255-266)
        const int r = 32 * (mb & 1) + l31;
        const int dsrc = d0 + (mb >> 1);
        const long pix = (((long)b * 24 + 2 * dsrc + pd) * 16 + 2 * (r >> 3) + ph) * 16 + 2 * (r & 7) + pw;
        if (ST && lhalf == 0) rinv[pix] = ri;
        if (dbg) { dbg[pix * 4 + lhalf] = ss; dbg[pix * 4 + 2 + lhalf] = ri; }      // (op-level test hook: both halves' row sums)
        char* orow = (char*)out + pix * 128 + lhalf * 16;
        if constexpr (G9) {
#pragma unroll
          for (int e = 0; e < 16; ++e) pacc[mb & 1][e] = 0.f;
        }
#pragma unroll
        for (int G = 0; G < 8; G += 2) {             // channel groups 8 G .. 8 G + 7 and 8 (G + 1) .. : one 16-byte store per lane
          unsigned lo[2], hi[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int nb = (G + u) >> 2, g = (G + u) & 3;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = acc[mb][nb][4 * g + e] * ri;
              v[e] = fmaxf(y, RD_LRELU_ALPHA * y);
            }
            lo[u] = rd_pack_bf16(v[0], v[1]); hi[u] = rd_pack_bf16(v[2], v[3]);
          }
          if constexpr (G9) {
            // k-step G / 2 of the last conv: this lane's eight channels (groups G, G + 1) of its row against the kernel image
            const u32x4_t bfr = {lo[0], hi[0], lo[1], hi[1]};
            pacc[mb & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, w9f[G >> 1]),
                                                                  __builtin_bit_cast(rd_bf16x8, bfr), pacc[mb & 1], 0, 0, 0);
          }
          if constexpr (ST) {
            // lanes 0-31 keep their group G and take the upper half's group G; lanes 32-63 take the lower half's group G + 1
            const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], lo[1], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi[0], hi[1], false, false);
            const u32x4_t o = {sx[0], sy[0], sx[1], sy[1]};
#ifdef RD_UPC_ABL_NOST               // (diagnostic builds: no output stores)
            if (o.x == 0x12345678u)
#endif
            *(u32x4_t*)(orow + G * 16) = o;    // (non-temporal stores: 25 % slower -- the four 16-byte pieces of a row's 128-byte line
            //                                     leave this lane in four instructions and want to meet in L2)
          }
        }
        if constexpr (G9) {
          if (mb & 1) {
            // ---- the plane (d0 + (mb >> 1), pd) is complete in pacc[0] (class rows 0..31) and pacc[1] (32..63): register e of
            // lane (l31, lhalf) = MFMA row (e & 3) + 8 (e >> 2) + 4 lhalf of class row l31
            float* Pw = (float*)(lds + RD_UPC_PW + wave * RD_UPC_PW_WAVE);
            const int th = lane >> 3, tw = lane & 7;
            // the (up to) nine sources of this lane's four targets: class position + validity, the same for every kd
            int srow[2][2]; bool sok[2][2];         // [axis][outer tap 0 / 2]: offset of the source class position
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int dh = i ? 1 - ph : -ph, dw = i ? 1 - pw : -pw;
              srow[0][i] = dh; sok[0][i] = (unsigned)(th + dh) < 8u;
              srow[1][i] = dw; sok[1][i] = (unsigned)(tw + dw) < 8u;
            }
            const int pl = mb >> 1;
            float* pwr = Pw + l31 * 11 + 4 * lhalf;
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) pwr[u * 32 * 11 + e] = pacc[u][4 * kd + e];
                Pw[(32 * u + l31) * 11 + 8 + lhalf] = pacc[u][12 + kd];          // upper half: a zero row into the dummy column
              }
              asm volatile("" ::: "memory");        // (one wave: the LDS executes its accesses in program order)
#pragma unroll
              for (int eh = 0; eh < 2; ++eh)
#pragma unroll
                for (int ew = 0; ew < 2; ++ew) {
                  float sum = 0.f;
#pragma unroll
                  for (int ih = 0; ih <= eh; ++ih) {
                    const int kh = eh ? 2 * ih : 1;
                    const int dh = eh ? srow[0][ih] : 0;
                    const bool okh = eh ? sok[0][ih] : true;
#pragma unroll
                    for (int iw = 0; iw <= ew; ++iw) {
                      const int kw = ew ? 2 * iw : 1;
                      const int dw = ew ? srow[1][iw] : 0;
                      const bool ok = okh && (ew ? sok[1][iw] : true);
                      const float pv = Pw[(ok ? lane + dh * 8 + dw : lane) * 11 + kh * 3 + kw];
                      sum += ok ? pv : 0.f;
                    }
                  }
                  const int qi = eh * 2 + ew, off = ((ph ^ eh) * 2 + (pw ^ ew)) * 64;
                  if (pl == 0 && kd == 0) hold[qi] = sum;                      // completed by plane A_(pd+2), tap kd = 2
                  else if (pd == 0) {
                    if (pl == 1 && kd == 2) qt[2 * 1024 + off] = hold[qi] + sum;
                    else qt[(2 * pl + 2 - kd) * 1024 + off] = sum;
                  } else {
                    if (pl == 1 && kd == 2) qt[3 * 1024 + off] = rmw[qi] + (hold[qi] + sum);
                    else if (pl == 1 && kd == 0) qt[5 * 1024 + off] = sum;
                    else qt[(2 * pl + 3 - kd) * 1024 + off] = rmw[(2 - kd) * 4 + qi] + sum;
                  }
                }
              asm volatile("" ::: "memory");
            }
          }
        }
      }
    }
  }
}
