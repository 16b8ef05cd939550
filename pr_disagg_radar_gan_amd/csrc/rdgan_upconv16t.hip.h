// bf16 storage mode, generator block 3 forward (T:340-343 / L:355-358: UpSampling3D(2) + Conv3D(128 -> 64, 3x3x3, 'same') + PixelNorm +
// LeakyReLU) in the collapsed form, as a slab kernel for source planes LARGER than 8 x 8: the large-domain variant
// (alternative_domains/...largedomain.py, ndomain 64: 12 x 32 x 32 x 128 -> 24 x 64 x 64 x 64; ndomain 32: 16 x 16 planes).  Round 4.
//
// k_upconv_slab16 (rdgan_upconv16.hip.h) keeps two whole 8 x 8 source planes + two halo planes resident (64 KB)
// and assumes the plane IS the tile.  Here a work item is an (h, w) TILE of 8 x 8 positions of two source hour planes:
//   * the tile is resident WITH ITS HALO: 10 x 10 positions x 4 planes (d0 - 1 .. d0 + 2), so every one of the 8 phases x 8 taps is
//     a plain shifted read -- no border test in the K loop; positions outside the picture are DMA'd as zeros (out-of-range offset);
//   * with all 128 channels that image would be 100 KB -- one workgroup per CU, and round 3 measured what a lone workgroup costs
//     (all waves of a CU in lock-step: epilogue beside epilogue, 1.75x the matrix time).  The image therefore holds HALF of the
//     channels (4 x 100 rows x 128 B = 50 KB, two workgroups per CU as in 4.6) and a tile's K loop runs in two halves that add into
//     the same accumulators: pd = 0: channels 0-63 then 64-127; pd = 1: 64-127 (already resident) then 0-63 -- three image loads
//     per item, 150 KB against 4 MB for the same rows x taps as a streaming GEMM;
//   * everything else is 4.6: weights global -> VGPR in MFMA-fragment order through a queue of four k-steps (inline-asm loads,
//     hand-counted vmcnt, checked by scripts/check_isa.py), operands swapped (a lane ends up with 32 channels of one output row),
//     bias + PixelNorm + LeakyReLU + bf16 rounding in registers, 16-byte stores, two 256-thread workgroups per CU.
//
// LDS image: row R = hh * 10 + ww (hh, ww = 0..9: tile position + 1) of plane slot s at s * 12800 + R * 128; 16-byte chunk c
// (8 channels) of a row at position c ^ (((hh & 1) << 2) | ((ww >> 1) & 3)): the 16 lanes of a fragment-read group take two tile
// rows x eight columns, whose (ww & 1, position) pairs are then all different -- 16 different 16-byte bank groups.  The DMA
// writes lane-linearly (8 rows per instruction), so the swizzle is applied to the SOURCE address.
#pragma once
#include "rdgan_upconv16.hip.h"

#define RD_UPT_ROWS 100                         // (8 + 2) x (8 + 2) positions
#define RD_UPT_SLOT (RD_UPT_ROWS * 128)         // bytes per plane slot: 64 channels of every position
#define RD_UPT_IMG (4 * RD_UPT_SLOT)            // 51,200 B = 50 DMA instructions of 1 KB
#define RD_UPT_BIAS RD_UPT_IMG                  // 64 floats
#define RD_UPT_LDS (RD_UPT_BIAS + 256)
// fused last conv (G9 = true): per-wave scratch [64 class rows][11: 9 (kh, kw) products + dummy + pad] fp32 behind the bias row
#define RD_UPT_PW RD_UPT_LDS
#define RD_UPT_LDS_G9 (RD_UPT_PW + 4 * RD_UPC_PW_WAVE)
// Q12 of the tiled kernel: per item (sample, plane pair, tile) the MAIN sums [6 target planes][4 source classes][4 target classes][64
// class positions of the tile] exactly as k_upconv_slab16 writes them, followed by the HALO terms [2 pd][2 planes][3 kd][4 source
// classes][36 (33 used)]: what the item's source pixels send to target positions one step outside the tile (a row, a column, a
// corner per target class), kept apart per (pd, plane, kd) -- k_tapsum_softmax12t adds them to the neighbouring tiles' sums
#define RD_UPT_QMAIN (6 * 4 * 4 * 64)
#define RD_UPT_QHALO (2 * 2 * 3 * 4 * 36)
#define RD_UPT_QITEM (RD_UPT_QMAIN + RD_UPT_QHALO)

// Weight image for the tiled kernel from the collapsed forms Wc [64 = phase*8 + tap][128 ci][64 co] (fp32, k_collapse_weights):
// k-steps in the order the kernel consumes them, [phase][half][tap][j] (j = 16-channel step inside the half): for k-step
// g = ((phase*2 + half)*8 + tap)*4 + j and column block nb, lane l holds the 8 bf16
// Wc[phase*8 + tap][64 half + 16 j + 8 (l >> 5) + e][32 nb + (l & 31)], e = 0..7 (the A fragment of v_mfma_f32_32x32x16_bf16):
// 1 KB per (g, nb), 64 KB contiguous per (phase, half), 1 MB in all.
__global__ void k_upconv_wimg_t(const float* __restrict__ Wc, unsigned short* __restrict__ wimg) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                 // (g, nb, lane)
  if (idx >= 512 * 2 * 64) return;
  const int lane = idx & 63, nb = (idx >> 6) & 1, g = idx >> 7;
  const int j = g & 3, tap = (g >> 2) & 7, half = (g >> 5) & 1, phase = g >> 6;
  const int n = nb * 32 + (lane & 31), k0 = half * 64 + j * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = Wc[((long)(phase * 8 + tap) * 128 + k0 + e) * 64 + n];
  u32x4_t o = {rd_pack_bf16(v[0], v[1]), rd_pack_bf16(v[2], v[3]), rd_pack_bf16(v[4], v[5]), rd_pack_bf16(v[6], v[7])};
  *(u32x4_t*)(wimg + (long)idx * 8) = o;
}

// x [B][12][H][W][128] bf16 -> out [B][24][2H][2W][64] bf16 = LeakyReLU(PixelNorm(upconv(x) + bias)), rinv [B][24][2H][2W] = 1/l2.
// H, W multiples of 8.  Work item = (sample, source plane pair d0 = 2 dp, tile (th, tw)); wave q computes the 128 rows (2 planes x
// 8 x 8 positions) x 64 channels of phase (pd, q >> 1, q & 1), pd = 0 then 1.  grid: persistent workgroups of 256 threads (two per
// CU); consecutive items -- neighbouring tiles, which share their halo columns -- go to workgroups of the same XCD (same L2).
// dynamic LDS RD_UPT_LDS.
// G9 (fused last conv, T:345; see k_upconv_slab16): the tap products of the 64 -> 1 conv are taken from the rows while they are in
// registers (four more MFMAs per 32 rows against the k_g9_wimg image) and summed over (kh, kw) inside the wave: Q12 (layout above)
// replaces the separate pass over h3 (k_g9_fwd: 805 MB read per call at ndomain 64 / 64 samples).  A tile's source pixels also feed
// target pixels one step OUTSIDE the tile; those sums leave as halo terms.  ST = false (critic steps): h3 and 1/l2 are not stored.
template <bool G9 = false, bool ST = true>
__global__ void __launch_bounds__(256, 2)
k_upconv_slab_t16(const rd_bf16_t* __restrict__ x, const rd_bf16_t* __restrict__ wimg, const float* __restrict__ bias,
                  rd_bf16_t* __restrict__ out, float* __restrict__ rinv, int B, int H, int W, float* __restrict__ dbg = nullptr,
                  const unsigned short* __restrict__ w9img = nullptr, float* __restrict__ Q12 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  const int ph = wave >> 1, pw = wave & 1;
  if (tid < 64) *(float*)(lds + RD_UPT_BIAS + tid * 4) = bias[tid];
  const unsigned wvoff = (unsigned)lane * 16u;
  const int TH = H >> 3, TW = W >> 3;
  const int nitem = B * 6 * TH * TW;
  // workgroups of one XCD (blockIdx % 8 under round-robin placement: speed only) take consecutive items
  const int G = gridDim.x;
  const int vid = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const long plane_b = (long)H * W * 256;                    // bytes per source plane

  for (int item = vid; item < nitem; item += G) {
    const int tw = item % TW, th = (item / TW) % TH, dp = (item / (TW * TH)) % 6, b = item / (6 * TH * TW);
    const int d0 = 2 * dp, h0 = 8 * th, w0 = 8 * tw;
    const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(x + (long)b * 12 * H * W * 128));
    f32x16 acc[4][2];
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
      // s = 0: (pd 0, channels 0-63)   1: (pd 0, 64-127) + epilogue   2: (pd 1, 64-127: resident)   3: (pd 1, 0-63) + epilogue
      const int pd = s >> 1, hf = (s ^ (s >> 1)) & 1;
      // weight fragments: a queue of four k-steps (8 loads in flight), STARTED HERE, in front of the image load: they do not depend on
      // the image, and their round trip then hides behind the DMA's (scripts/check_isa.py watches the queue registers across the
      // branch below: nothing may touch them before the counted waits of the K loop)
      const char* wph = (const char*)wimg + (long)(((pd * 4 + ph * 2 + pw) * 2 + hf) * 32) * 2048;      // wave-uniform
      u32x4_t bq[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) rd_upc_wload(bq[q][0], bq[q][1], wph + q * 2048, wvoff);
      if (s != 2) {
        __syncthreads();                                  // every wave has left the image (and the bias row is in)
        // the four planes d0 - 1 .. d0 + 2, positions (h0 - 1 .. h0 + 8) x (w0 - 1 .. w0 + 8), channels 64 hf ..: 50 instructions
        // (the lane index passes through an asm statement: the per-lane image geometry below is the same for every item, and
        // hipcc otherwise keeps all of it -- ~50 registers -- alive across the K loops, beside 128 accumulators: 41 spilled
        // registers in the first build, flagged by scripts/check_isa.py)
        int lq = lane;
        asm volatile("" : "+v"(lq));
#pragma unroll 1
        for (int k = 0; k < 13; ++k) {
          const int i = wave + 4 * k;                     // wave-uniform
          if (i < 50) {
            const int rg = i * 8 + (lq >> 3);             // row of the image, 0..399
            const int slot = rg / RD_UPT_ROWS, R = rg - slot * RD_UPT_ROWS;
            const int hh = R / 10, ww = R - hh * 10;
            const int c_log = (lq & 7) ^ (((hh & 1) << 2) | ((ww >> 1) & 3));
            const int d = d0 - 1 + slot, hs = h0 + hh - 1, ws = w0 + ww - 1;
            const bool ok = (unsigned)d < 12u && (unsigned)hs < (unsigned)H && (unsigned)ws < (unsigned)W;
            unsigned voff = ok ? (unsigned)(d * plane_b + ((long)(hs * W + ws) * 256) + hf * 128 + c_log * 16) : RD_OOB;
            asm volatile("" : "+v"(voff));
            rd_lds_dma16(rs, (float*)(lds + i * 1024), (int)voff, 0);
          }
        }
        rd_dma_landed();
        __syncthreads();
      }
      if ((s & 1) == 0) {
        // accumulators start at the bias of their channel: register r of block nb = channel 32 nb + 8 (r >> 2) + 4 lhalf + (r & 3)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(const f32x4*)(lds + RD_UPT_BIAS + (nb * 32 + 8 * g + 4 * lhalf) * 4);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
              acc[mb][nb][4 * g + 0] = b4.x; acc[mb][nb][4 * g + 1] = b4.y;
              acc[mb][nb][4 * g + 2] = b4.z; acc[mb][nb][4 * g + 3] = b4.w;
            }
          }
      }
      // ---- half a K loop: 8 taps x 4 k-steps of 16 channels
      int abase[4], aswz[4];
      auto tap_rows = [&](int t) {
        // tap t = (td, th, tw): source offsets (pd - 1 + td, ph - 1 + th, pw - 1 + tw); the halo is in the image: always valid
        const int od = pd - 1 + (t >> 2), oh = ph - 1 + ((t >> 1) & 1), ow = pw - 1 + (t & 1);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const int r = 32 * (mb & 1) + l31;
          const int hh = (r >> 3) + 1 + oh, ww = (r & 7) + 1 + ow;
          abase[mb] = ((mb >> 1) + 1 + od) * RD_UPT_SLOT + (hh * 10 + ww) * 128;
          aswz[mb] = (((hh & 1) << 2) | ((ww >> 1) & 3)) ^ lhalf;
        }
      };
      u32x4_t afr[2][4];
      tap_rows(0);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) afr[0][mb] = *(const u32x4_t*)(lds + abase[mb] + (aswz[mb] << 4));
#pragma unroll 1
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < 3) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
              afr[(j + 1) & 1][mb] = *(const u32x4_t*)(lds + abase[mb] + (((2 * (j + 1)) ^ aswz[mb]) << 4));
          } else {
            // the next tap's first fragments behind this tap's last k-step (tap 7: tap 7 again, never used)
            tap_rows(t < 7 ? t + 1 : 7);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) afr[0][mb] = *(const u32x4_t*)(lds + abase[mb] + (aswz[mb] << 4));
          }
          rd_upc_wait<6>(bq[j][0], bq[j][1]);               // the two oldest of the eight loads in flight
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[j][0]),
                                                                 __builtin_bit_cast(rd_bf16x8, afr[j & 1][mb]), acc[mb][0], 0, 0, 0);
            acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[j][1]),
                                                                 __builtin_bit_cast(rd_bf16x8, afr[j & 1][mb]), acc[mb][1], 0, 0, 0);
          }
          {
            // refill the slot with k-step t*4 + j + 4 of this half (past its end: the last k-step again, never used)
            const int gn = t * 4 + j + 4;
            rd_upc_wload(bq[j][0], bq[j][1], wph + (long)(gn < 31 ? gn : 31) * 2048, wvoff);
          }
        }
      }
      // the clamped refills of the last four k-steps are still in flight and nobody reads them: waited for HERE, naming their
      // registers (rdgan_upconv16.hip.h: to the compiler they are dead behind the loop)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1]), "+v"(bq[3][0]),
                     "+v"(bq[3][1]));
      if ((s & 1) == 0) continue;

      // ---- epilogue, in registers (as in k_upconv_slab16): lane (l31, lhalf) of block mb holds 32 channels of output row
      // m = 32 mb + l31 of the wave's 128 rows (channels 32 nb + 8 g + 4 lhalf + 0..3), lane ^ 32 the other 32
      u32x4_t w9f[4];
      if constexpr (G9) {
        const unsigned short* wp = w9img;
        asm volatile("" : "+s"(wp));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w9f[ks] = *(const u32x4_t*)(wp + (ks * 64 + lane) * 8);
      }
      f32x16 pacc[2];
      float hold[4] = {0.f, 0.f, 0.f, 0.f}, rmw[8];
      // (the offset goes through an asm statement so that hipcc computes it HERE, behind the K loop's final wait: rdgan_upconv16.hip.h)
      int qtoff = item;
      if constexpr (G9) asm volatile("" : "+v"(qtoff));
      float* qt = Q12 + (long)qtoff * RD_UPT_QITEM + wave * 256 + lane;          // main sums: [tp][p = wave][target class][64]
      float* qh = Q12 + (long)qtoff * RD_UPT_QITEM + RD_UPT_QMAIN + wave * 36 + lane;      // halo terms: [pd][pl][kd][p = wave][36]
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        if constexpr (G9) {
          if ((mb & 1) == 0) {      // the two targets plane mb >> 1 of this pass adds to (tp = 2 (mb >> 1) + 1, + 2)
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
        float ss = 0.f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) ss = fmaf(acc[mb][nb][r], acc[mb][nb][r], ss);
        ss += __shfl_xor(ss, 32, 64);                       // + the other half's 32 channels of the same row
        const float ri = __builtin_amdgcn_rsqf(ss * (1.0f / 64.0f) + 1.0e-8f);       // PixelNormalization (T:255-266)
        const int r = 32 * (mb & 1) + l31;
        const int dsrc = d0 + (mb >> 1);
        const long pix = (((long)b * 24 + 2 * dsrc + pd) * (2 * H) + 2 * (h0 + (r >> 3)) + ph) * (2 * W) + 2 * (w0 + (r & 7)) + pw;
        if (ST && lhalf == 0) rinv[pix] = ri;
        if (dbg) { dbg[pix * 4 + lhalf] = ss; dbg[pix * 4 + 2 + lhalf] = ri; }      // (op-level test hook: both halves' row sums)
        char* orow = (char*)out + pix * 128 + lhalf * 16;
        if constexpr (G9) {
#pragma unroll
          for (int e = 0; e < 16; ++e) pacc[mb & 1][e] = 0.f;
        }
#pragma unroll
        for (int Gc = 0; Gc < 8; Gc += 2) {             // channel groups 8 Gc .. and 8 (Gc + 1) ..: one 16-byte store per lane
          unsigned lo[2], hi[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int nb = (Gc + u) >> 2, g = (Gc + u) & 3;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = acc[mb][nb][4 * g + e] * ri;
              v[e] = fmaxf(y, RD_LRELU_ALPHA * y);
            }
            lo[u] = rd_pack_bf16(v[0], v[1]); hi[u] = rd_pack_bf16(v[2], v[3]);
          }
          if constexpr (G9) {
            // k-step Gc / 2 of the last conv: this lane's eight channels (groups Gc, Gc + 1) of its row against the kernel image
            const u32x4_t bfr = {lo[0], hi[0], lo[1], hi[1]};
            pacc[mb & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, w9f[Gc >> 1]),
                                                                  __builtin_bit_cast(rd_bf16x8, bfr), pacc[mb & 1], 0, 0, 0);
          }
          if constexpr (ST) {
            // lanes 0-31 keep their group Gc and take the upper half's group Gc; lanes 32-63 take the lower half's group Gc + 1
            const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], lo[1], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi[0], hi[1], false, false);
            const u32x4_t o = {sx[0], sy[0], sx[1], sy[1]};
            *(u32x4_t*)(orow + Gc * 16) = o;
          }
        }
        if constexpr (G9) {
          if (mb & 1) {
            // ---- the plane (d0 + (mb >> 1), pd) is complete in pacc[0] (class rows 0..31) and pacc[1] (32..63): register e of
            // lane (l31, lhalf) = MFMA row (e & 3) + 8 (e >> 2) + 4 lhalf of class row l31 (rdgan_upconv16.hip.h)
            float* Pw = (float*)(lds + RD_UPT_PW + wave * RD_UPC_PW_WAVE);
            int lq2 = lane;
            asm volatile("" : "+v"(lq2));
            const int th_ = lq2 >> 3, tw_ = lq2 & 7;
            int srow[2][2]; bool sok[2][2];         // [axis][outer tap 0 / 2]: offset of the source class position, inside the TILE?
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int dh = i ? 1 - ph : -ph, dw = i ? 1 - pw : -pw;
              srow[0][i] = dh; sok[0][i] = (unsigned)(th_ + dh) < 8u;
              srow[1][i] = dw; sok[1][i] = (unsigned)(tw_ + dw) < 8u;
            }
            // halo lanes: what this tile's sources send one step outside it.  Lane hl < 33: 0-7 target class (other h parity, same w
            // parity): the row yx = (ph ? 8 : -1), tx = hl; 8-15 (same h, other w): the column xx = (pw ? 8 : -1), ty = hl - 8;
            // 16-24 (other, other): row yx, tx = xlo + hl - 16 (xlo = pw ? 0 : -1); 25-32 (other, other): column xx, ty = hl - 25.
            // A row target has ONE source row (ph ? (7, kh 0) : (0, kh 2)), a column target one source column; the other axis sums
            // its one (same parity: centre tap) or two (other parity: outer taps) sources inside the tile.
            const int hsy = ph ? 7 : 0, hkh = ph ? 0 : 2, hsx = pw ? 7 : 0, hkw = pw ? 0 : 2;
            int ho[2]; bool hv[2];                  // the (up to) two Pw entries of this lane's halo target
            {
              int hl = lane;
              asm volatile("" : "+v"(hl));          // (recomputed per plane: hoisted out of the item loop these offsets were spilled)
              const bool rowt = hl < 8 || (hl >= 16 && hl < 25);
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                int sy, sx, kh, kw; bool v;
                if (hl < 8) { sy = hsy; kh = hkh; sx = hl; kw = 1; v = i == 0; }
                else if (hl < 16) { sx = hsx; kw = hkw; sy = hl - 8; kh = 1; v = i == 0; }
                else if (rowt) { sy = hsy; kh = hkh; const int tx = (pw ? 0 : -1) + hl - 16; sx = tx + (i ? 1 - pw : -pw); kw = 2 * i; v = (unsigned)sx < 8u; }
                else { sx = hsx; kw = hkw; const int ty = hl - 25; sy = ty + (i ? 1 - ph : -ph); kh = 2 * i; v = hl < 33 && (unsigned)sy < 8u; }
                hv[i] = v;
                ho[i] = v ? (sy * 8 + sx) * 11 + kh * 3 + kw : 0;
              }
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
              {
                const float h0v = Pw[ho[0]], h1v = Pw[ho[1]];
                const float hs = (hv[0] ? h0v : 0.f) + (hv[1] ? h1v : 0.f);
                if (lane < 33) qh[((pd * 2 + pl) * 3 + kd) * 144] = hs;
              }
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

// The tail behind k_upconv_slab_t16<G9 = true>, in two steps.
// (1) k_g9_halo_fold: every MAIN entry on a tile edge that a neighbouring tile's sources reach takes that neighbour's HALO terms --
//     in place, one thread per receiving entry (per item, target plane tp and source class p: the edge row of the target classes of
//     the other h parity, the edge column of those of the other w parity; the corner position of the class that is both takes the
//     row term, the column term and the diagonal tile's corner term), in a fixed order: (pd, plane, kd) triples ascending, row
//     before column before corner.  A few hundred entries per item: the launch is a few microseconds.
// (2) k_tapsum_softmax12t: k_tapsum_softmax12 on tiles -- logit = bias + [the item below's tp 5] + the own item's tp (d & 3) + 1 +
//     [the item above's tp 0], four source classes each, then the softmax over the hours (T:345-347).  A lane owns four consecutive
//     class positions (16-byte loads) and six hours.  (A first version did the halo sums inside this kernel, one column and 4-byte
//     loads per lane: 0.27 ms per call at ndomain 64 / 64 samples, more than the pass over h3 it replaced; with 16-byte loads 0.11.)
// H, W = class-grid extents (the source plane of block 3), multiples of 8.
__global__ void __launch_bounds__(256)
k_g9_halo_fold(float* __restrict__ Q12, int B, int TH, int TW) {
  // (32-bit index arithmetic: 4.7 M threads at ndomain 64 / 64 samples; with 64-bit divisions the launch took 49 us)
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;          // (item, tp, p, e): e < 31 used of 32
  const int e = (int)(gid & 31u);
  const unsigned t = gid >> 5;
  const int p = (int)(t & 3u), tp = (int)((t >> 2) % 6u);
  const unsigned item = t / 24u;
  if (item >= (unsigned)(B * 6 * TH * TW) || e >= 31) return;
  const int tw = (int)(item % (unsigned)TW), th = (int)((item / (unsigned)TW) % (unsigned)TH);
  const int ph = p >> 1, pw = p & 1;
  const int nth = th + (ph ? -1 : 1), ntw = tw + (pw ? -1 : 1);
  const bool hasr = (unsigned)nth < (unsigned)TH, hasc = (unsigned)ntw < (unsigned)TW;
  const int lye = ph ? 0 : 7, lxe = pw ? 0 : 7;
  // e 0-7: class (other h, same w), edge row, lx = e; 8-15: (same h, other w), edge column, ly = e - 8; 16-23: (other, other), edge
  // row, lx = e - 16; 24-30: (other, other), edge column without the corner, ly = the (e - 24)-th row that is not lye
  int eh, ew, ly, lx;
  if (e < 8) { eh = 1; ew = 0; ly = lye; lx = e; }
  else if (e < 16) { eh = 0; ew = 1; ly = e - 8; lx = lxe; }
  else if (e < 24) { eh = 1; ew = 1; ly = lye; lx = e - 16; }
  else { eh = 1; ew = 1; ly = (e - 24) + (ph ? 1 : 0); lx = lxe; }
  const bool row = eh && ly == lye && hasr, col = ew && lx == lxe && hasc;
  if (!row && !col) return;
  const int q = ((ph ^ eh) << 1) | (pw ^ ew);
  float* It = Q12 + (long)(item - (unsigned)(th * TW + tw)) * RD_UPT_QITEM;  // tile (0, 0) of this (sample, plane pair)
  float* dst = It + (long)(th * TW + tw) * RD_UPT_QITEM + tp * 1024 + p * 256 + q * 64 + ly * 8 + lx;
  // (all loads issued before the first add: with a data-dependent loop around them the launch was a chain of round trips, 48 us)
  float rv[4], cv[4], kv[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int kd = a + 2 - tp, pd = a & 1, pl = a >> 1;
    const bool va = kd >= 0 && kd <= 2;
    const int base = RD_UPT_QMAIN + ((pd * 2 + pl) * 3 + (va ? kd : 0)) * 144 + p * 36;
    rv[a] = row && va ? It[(long)(nth * TW + tw) * RD_UPT_QITEM + base + (ew ? 16 + (pw ? 0 : 1) : 0) + lx] : 0.f;
    cv[a] = col && va ? It[(long)(th * TW + ntw) * RD_UPT_QITEM + base + (eh ? 25 : 8) + ly] : 0.f;
    kv[a] = row && col && va ? It[(long)(nth * TW + ntw) * RD_UPT_QITEM + base + 16 + (pw ? 8 : 0)] : 0.f;
  }
  float s = *dst;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int kd = a + 2 - tp;
    if (kd < 0 || kd > 2) continue;
    if (row) s += rv[a];
    if (col) s += cv[a];
    if (row && col) s += kv[a];
  }
  *dst = s;
}
template <int D>
__global__ void __launch_bounds__(256)
k_tapsum_softmax12t(const float* __restrict__ Q12, const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W,
                    int* __restrict__ nonfinite) {
  static_assert(D % 4 == 0, "hours per lane");
  constexpr int DP = D / 4;
  const int TH = H >> 3, TW = W >> 3;
  const long ngrp = (long)B * H * W;                         // groups of four columns
  const int lane = threadIdx.x & 63, part = lane >> 4;
  const long grp = (blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (lane & 15);
  const bool live = grp < ngrp;
  const long gid = live ? grp : 0;
  const int per = H * W;                                     // groups per sample: tiles x 4 classes x 16
  const int b = (int)(gid / per), rest = (int)(gid - (long)b * per);
  const int pq = rest & 15, q = (rest >> 4) & 3, tile = rest >> 6, tw = tile % TW, th = tile / TW;
  const int qh_ = q >> 1, qw_ = q & 1, ly = pq >> 1, lx0 = (pq & 1) * 4;
  const float bv = bias[0];
  const long istride = (long)TH * TW * RD_UPT_QITEM;         // floats per (sample, plane pair)
  const float* qb = Q12 + (long)b * 6 * istride + (long)tile * RD_UPT_QITEM + q * 64 + ly * 8 + lx0;
  f32x4 lg[DP];
  f32x4 mx = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    const int d = part * DP + i, it = d >> 2, k = d & 3;
    f32x4 s = {bv, bv, bv, bv};
    if (k == 0 && it > 0) {                           // the item below reaches it through its last plane (hour tap kd = 0)
#pragma unroll
      for (int p = 0; p < 4; ++p) s += *(const f32x4*)(qb + (it - 1) * istride + 5 * 1024 + p * 256);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) s += *(const f32x4*)(qb + it * istride + (k + 1) * 1024 + p * 256);
    if (k == 3 && it < D / 4 - 1) {                   // the item above through its first plane (kd = 2)
#pragma unroll
      for (int p = 0; p < 4; ++p) s += *(const f32x4*)(qb + (it + 1) * istride + p * 256);
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
  const int Y = 2 * (8 * th + ly) + qh_, X0 = 2 * (8 * tw + lx0) + qw_;          // the four columns: X0, X0 + 2, X0 + 4, X0 + 6
  const long hw = 4L * H * W;
  float* o = out + (long)b * D * hw + (long)Y * (2 * W) + X0;
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
