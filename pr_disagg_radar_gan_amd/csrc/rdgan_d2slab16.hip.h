// bf16 storage mode, ndomain 16: input gradient of the critic's second layer (T:291: Conv3D(128, 3x3x3, stride 2, 'same') on
// the 11 x 7 x 7 x 64 output of layer 1 -> 6 x 4 x 4 x 128) as a SLAB kernel in the pattern of k_upconv_slab16.
//
// Why.  By input parity the gradient splits into 8 phases of 8, 4, 4, 2, 4, 2, 2, 1 taps (plan_conv_dgrad_s2), each a GEMM with
// N = 64 output channels and K = taps x 128.  As tiles of the streaming kernel (k_conv_gemm_ws<256, 64, ..., bf16>) that is
// 12 936 workgroups of 2-16 K chunks each at 6144 samples: all prologue and epilogue, 0.58 ms = 0.11 of the bf16 roof, the slowest
// launch per FLOP of BASELINE configs[2] (2.9 ms of a 45.8 ms iteration).  Here:
//   * a workgroup owns TWO samples' output gradient (2 x 96 positions x 128 channels = 48 KB, one contiguous piece of HBM) resident
//     in LDS for all 8 phases x 27 taps; every tap of every phase is a shifted read of those rows.  With 'same' padding (1,1,1)
//     and these extents every tap of every phase lands inside the picture: no masks.
//   * weights: MFMA-fragment order in HBM (k_d2s_wimg, 432 KB, L2-resident), streamed global -> VGPR four k-steps ahead by the
//     wave that uses them; no barrier inside an item besides the one that publishes the slab.
//   * operands swapped (weights = A, positions = B): a lane ends up with 32 channels of ONE position, so LeakyReLU' x dropout of
//     layer 1 (the gate, RD_EPI_GATE_AUX of the streaming kernel: same dropout counter = flat index of the destination) and the
//     bf16 rounding run in registers, 16-byte stores.
// Rows of a phase = (sample in the item, position in the phase's 5|6 x 3|4 x 3|4 sub-grid), flattened: 90 ... 192 rows, cut into
// tiles of 2-4 row blocks of 32 and dealt to the four waves so that taps x blocks + epilogues balance (table rd_d2s_tiles).
// Same tap order, same k order, same fp32 accumulation as the streaming kernel.
#pragma once
#include "rdgan_upconv16.hip.h"

#define RD_D2S_S 2                                    // samples per work item
#define RD_D2S_SROWS 96                               // source positions per sample (6 x 4 x 4)
#define RD_D2S_IMG (RD_D2S_S * RD_D2S_SROWS * 256)    // bytes of the resident image
#define RD_D2S_ZERO RD_D2S_IMG                        // a 256-byte row of zeros: rows past the item's last position
#define RD_D2S_LDS (RD_D2S_IMG + 256)
#define RD_D2S_KSTEPS 216                             // 27 taps x 8 steps of 16 channels
#define RD_D2S_OPOS 539                               // destination positions per sample (11 x 7 x 7)

// first tap slot of phase cls (phases in plan order: cls = (pd, ph, pw) parity bits of position + pad; taps 8,4,4,2,4,2,2,1)
__host__ __device__ __forceinline__ int rd_d2s_tap0(int cls) {
  return (0x1A181612100C0800ull >> (8 * cls)) & 0xFF;   // 0, 8, 12, 16, 18, 22, 24, 26
}

// Weight image from the layer's kernel w2 [27][64 ci][128 co] (fp32): for k-step g = slot * 8 + j (slot = tap in phase order,
// j = 16-channel step of co) and block nb of 32 ci, lane l holds the 8 bf16 w2[tap][32 nb + (l & 31)][16 j + 8 (l >> 5) + e]:
// the A fragment of v_mfma_f32_32x32x16_bf16 for gx^T = W gy^T.
__global__ void k_d2s_wimg(const float* __restrict__ w2, unsigned short* __restrict__ wimg) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                 // (g, nb, lane)
  if (idx >= RD_D2S_KSTEPS * 2 * 64) return;
  const int lane = idx & 63, nb = (idx >> 6) & 1, g = idx >> 7;
  const int j = g & 7, slot = g >> 3;
  int cls = 7;
  while (rd_d2s_tap0(cls) > slot) --cls;
  int rem = slot - rd_d2s_tap0(cls), wtap = 0, mul = 1;
  for (int a = 2; a >= 0; --a) {                  // axes w, h, d: an axis of parity 0 takes taps 0 and 2 (w fastest), parity 1 tap 1
    const int pi = (cls >> (2 - a)) & 1;
    int ta = 1;
    if (!pi) { ta = 2 * (rem & 1); rem >>= 1; }
    wtap += ta * mul; mul *= 3;
  }
  const int n = nb * 32 + (lane & 31), k0 = j * 16 + (lane >> 5) * 8;
  const float* s = w2 + ((long)wtap * 64 + n) * 128 + k0;
  u32x4_t o = {rd_pack_bf16(s[0], s[1]), rd_pack_bf16(s[2], s[3]), rd_pack_bf16(s[4], s[5]), rd_pack_bf16(s[6], s[7])};
  *(u32x4_t*)(wimg + (long)idx * 8) = o;
}

// tiles of an item: wave w runs rd_d2s_tiles[w][0..3] = cls | first row << 4 | row blocks << 12 (0 blocks: no tile)
// (taps x blocks per wave: 24+3+3, 8+8+6+4, 8+8+6+4, 8+8+6+4; row blocks, i.e. epilogues: 9, 9, 9, 9.  At most 3 blocks per tile:
// with 4 the accumulators (128), the weight queue (32) and the tile's gate rows (64) do not fit 256 registers)
#define RD_D2S_T(cls, row0, mbs) ((cls) | ((row0) << 4) | ((mbs) << 12))
__constant__ int rd_d2s_tiles[4][4] = {
  {RD_D2S_T(0, 0, 3), RD_D2S_T(7, 0, 3), RD_D2S_T(7, 96, 3), 0},
  {RD_D2S_T(1, 0, 2), RD_D2S_T(1, 64, 2), RD_D2S_T(3, 0, 3), RD_D2S_T(3, 96, 2)},
  {RD_D2S_T(2, 0, 2), RD_D2S_T(2, 64, 2), RD_D2S_T(5, 0, 3), RD_D2S_T(5, 96, 2)},
  {RD_D2S_T(4, 0, 2), RD_D2S_T(4, 64, 2), RD_D2S_T(6, 0, 3), RD_D2S_T(6, 96, 2)},
};

// Geometry of the TILED variant (k_d2_dgrad_slab_t16, round 4: domains larger than 16 x 16): an item is the same (h, w) tile of 8 x 8
// destination positions of TWO samples; per sample the 6 x 5 x 5 output-gradient positions the tile's taps reach are resident
// (150 rows; position (od, i, j) = source (od, oh0 + i, ow0 + j)), so a tap is a shifted read with strides 25 / 5 / 1.
struct RdD2sGeom {
  int last_h, last_w;      // the tile is the last one along h / w: 7 destination positions there (sub-grid 3 + parity instead of 4)
  int IH, IW;              // destination grid (layer 1's output): 11 x IH x IW positions per sample
  int h0, w0;              // first destination position of the tile (8 ti, 8 tj)
};
#define RD_D2T_SROWS 150                              // resident source positions per sample of an item: 6 x 5 x 5
#define RD_D2T_IMG (RD_D2S_S * RD_D2T_SROWS * 256)
#define RD_D2T_ZERO RD_D2T_IMG
#define RD_D2T_LDS (RD_D2T_IMG + 256)

// one tile: rows row0 .. row0 + 32 MB - 1 of phase cls, all 64 channels
template <int MB, bool TILED = false>
__device__ __forceinline__ void rd_d2s_tile(const char* lds, const char* wimg, unsigned wvoff, int cls, int row0, int ns, long b0,
                                            const rd_bf16_t* __restrict__ aux, rd_bf16_t* __restrict__ out, int use_drop,
                                            int l31, int lhalf, const unsigned char* __restrict__ gbits, const RdD2sGeom G = RdD2sGeom()) {
  constexpr int SROWS = TILED ? RD_D2T_SROWS : RD_D2S_SROWS, SD = TILED ? 25 : 16, SH = TILED ? 5 : 4;
  constexpr int ZERO = TILED ? RD_D2T_ZERO : RD_D2S_ZERO;
  const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
  // the phase's sub-grid (positions 2 l + 1 - parity): 3 + parity where the destination extent is 7, 4 in a full tile of 8
  const int cd = 5 + pd, chh = (!TILED || G.last_h) ? 3 + ph : 4, cw = (!TILED || G.last_w) ? 3 + pw : 4;
  const int IH = TILED ? G.IH : 7, IW = TILED ? G.IW : 7, OPOS = TILED ? 11 * G.IH * G.IW : RD_D2S_OPOS;
  const int h0 = TILED ? G.h0 : 0, w0 = TILED ? G.w0 : 0;
  const int chw = chh * cw, cnt = cd * chw;
  const int rows = ns * cnt;
  const int ntaps = 8 >> (pd + ph + pw);
  const int nks = ntaps * 8;
  int srow[MB];                 // source row of tap offset (0,0,0), or -1
  int orow[MB];                 // destination row
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int r = row0 + 32 * mb + l31;
    const int s = r >= cnt ? 1 : 0;
    const int p = r - s * cnt;
    const int ld = p / chw, q = p - ld * chw, lh = q / cw, lw = q - lh * cw;
    srow[mb] = r < rows ? s * SROWS + ld * SD + lh * SH + lw : -1;
    orow[mb] = ((int)b0 + s) * OPOS + ((2 * ld + 1 - pd) * IH + h0 + 2 * lh + 1 - ph) * IW + w0 + 2 * lw + 1 - pw;
  }
  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const char* wph = wimg + (long)rd_d2s_tap0(cls) * 8 * 2048;        // wave-uniform
  u32x4_t bq[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) rd_upc_wload(bq[s][0], bq[s][1], wph + s * 2048, wvoff);
  // (no `continue` in this loop: the weight queue is carried around it with loads in flight, and a path that skips the body would
  // make its registers phi values -- see the refill below)
#pragma unroll 1
  for (int ti = 0; ti < ntaps; ++ti) {
    // tap ti of the phase: the bits of ti go to the axes of parity 0, w first (an axis of parity 1 has one tap)
    int rem = ti;
    const int jw = pw ? 0 : (rem & 1); rem >>= pw ? 0 : 1;
    const int jh = ph ? 0 : (rem & 1); rem >>= ph ? 0 : 1;
    const int jd = pd ? 0 : (rem & 1);
    // source offset of the tap: 1 - j on an axis of parity 0, 0 on an axis of parity 1
    const int shift = (pd ? 0 : 1 - jd) * SD + (ph ? 0 : 1 - jh) * SH + (pw ? 0 : 1 - jw);
    int abase[MB], aswz[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int rs = srow[mb] + shift;
      abase[mb] = srow[mb] >= 0 ? rs * 256 : ZERO;
      aswz[mb] = srow[mb] >= 0 ? ((rs & 15) ^ lhalf) : lhalf;
    }
    u32x4_t afr[2][MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) afr[0][mb] = *(const u32x4_t*)(lds + abase[mb] + (aswz[mb] << 4));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#ifndef RD_D2S_ABL_NOA               // (diagnostic builds, scratch/d2s_abl.py: the K loop without its LDS fragment reads)
      if (j + 1 < 8) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          afr[(j + 1) & 1][mb] = *(const u32x4_t*)(lds + abase[mb] + (((2 * (j + 1)) ^ aswz[mb]) << 4));
      }
#else
      if (j + 1 < 8) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { afr[(j + 1) & 1][mb] = afr[j & 1][mb]; asm volatile("" : "+v"(afr[(j + 1) & 1][mb])); }
      }
#endif
#ifndef RD_D2S_ABL_NOW               // (diagnostic builds: the K loop without its weight stream)
      rd_upc_wait<6>(bq[j & 3][0], bq[j & 3][1]);          // the two oldest of the eight loads in flight
#endif
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[j & 3][0]),
                                                             __builtin_bit_cast(rd_bf16x8, afr[j & 1][mb]), acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rd_bf16x8, bq[j & 3][1]),
                                                             __builtin_bit_cast(rd_bf16x8, afr[j & 1][mb]), acc[mb][1], 0, 0, 0);
      }
#ifndef RD_D2S_ABL_NOW
      {
        // refill with k-step + 4 of the tile (past its end: the last one again, never used).  The SAME two statements for every
        // k-step, no branch between a load and its wait: a first attempt to stop the refills at the tile's end (waits of 4, 2, 0
        // in the last tap, chosen by a wave-uniform branch) made the queue registers phi values, and hipcc placed the copies that
        // reconcile them IN FRONT of the wait -- copies of registers a load was still in flight to (phase 0 wrong, round 3)
        const int gn = ti * 8 + j + 4;
        rd_upc_wload(bq[j & 3][0], bq[j & 3][1], wph + (long)(gn < nks ? gn : nks - 1) * 2048, wvoff);
      }
#else
      asm volatile("" : "+v"(bq[j & 3][0]), "+v"(bq[j & 3][1]));
#endif
    }
  }
  // The clamped refills of the last four k-steps are still in flight and nobody will read them: wait for them HERE, naming their
  // destination registers, before anything else is allocated -- to hipcc those registers are dead behind the loop.  (With the
  // gate loads below in front of this wait it handed them the queue registers for their ADDRESSES; a late weight fragment then
  // overwrote an address between its computation and the load that used it: memory fault, round 3.)
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1]), "+v"(bq[3][0]),
                 "+v"(bq[3][1]));
  // The gate rows of the WHOLE tile (layer 1's stored output at this tile's destination rows, 64 B per lane and row) are requested
  // here, in one go: one memory round trip per tile.  (A first version loaded them block by block
  // inside the epilogue loop -- a divergent `continue` for rows past the item's end kept hipcc from hoisting them -- and paid a
  // round trip per 32-row block: 0.16 of the kernel's 0.38 ms at 6144 samples, scratch/d2s_abl.py.)  Rows past the end read the
  // item's first row instead of branching.
  // gbits != nullptr: the forward of layer 1 (k_d1_gemm_fwd) left the gate in 2 bits per element, 16 bytes per row: ONE 16-byte
  // load per row block instead of four + eight half exchanges, an eighth of the bytes (the epilogue was 0.17 of this kernel's 0.34 ms)
  rd_u32x2 gate[MB][8];
  u32x4_t gcode[MB];
  if (gbits) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      gcode[mb] = *(const u32x4_t*)(gbits + (long)(srow[mb] >= 0 ? orow[mb] : (int)b0 * OPOS) * 16);
  } else {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    // 16 bytes per lane (lane half h takes the 8-channel chunks 2 P + h), then each half hands the other the four channels it
    // does not own (the inverse of the store path below): 4 instead of 8 load instructions per row block, 32 B per row and
    // instruction instead of 16
    const rd_bf16_t* arow = aux + (long)(srow[mb] >= 0 ? orow[mb] : (int)b0 * OPOS) * 64 + 8 * lhalf;
#pragma unroll
#ifndef RD_D2S_ABL_NOGATE            // (diagnostic builds: no gate loads)
    for (int P = 0; P < 4; ++P) {
      const u32x4_t w = *(const u32x4_t*)(arow + 16 * P);
      const auto sx = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
      gate[mb][2 * P].x = sx[0]; gate[mb][2 * P + 1].x = sx[1];
      gate[mb][2 * P].y = sy[0]; gate[mb][2 * P + 1].y = sy[1];
    }
#else
    for (int G = 0; G < 8; ++G) { gate[mb][G].x = 0x3F803F80u + G + orow[mb]; gate[mb][G].y = 0xBF803F80u; }
#endif
  }
  }
#ifdef RD_D2S_ABL_NOEPI              // (diagnostic builds: K loops only; one element of every accumulator keeps the MFMAs alive)
  {
    float tsum = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) tsum += acc[mb][0][3] + acc[mb][1][7];
    if (tsum == 12345.678f) out[0] = 1;
    return;
  }
#endif
  // ---- epilogue in registers: lane (l31, lhalf) of block mb holds channels 32 nb + 8 g + 4 lhalf + 0..3 of its row
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const bool ok = srow[mb] >= 0;                // (lanes l and l ^ 32 share their row: the swaps below see both or neither)
    char* op = (char*)out + (long)orow[mb] * 128 + lhalf * 16;
#pragma unroll
    for (int G = 0; G < 8; G += 2) {
      unsigned lo[2], hi[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int nb = (G + u) >> 2, g = (G + u) & 3;
        float v[4];
        if (gbits) {
          // quad 2 (G + u) + lhalf of the row: byte (2 ((G + u) & 1) + lhalf) of dword (G + u) >> 1
          const unsigned byte = gcode[mb][(G + u) >> 1] >> (16 * ((G + u) & 1) + 8 * lhalf);
          const float s1 = use_drop ? (1.0f / 0.75f) : 1.0f, s2 = RD_LRELU_ALPHA * s1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned code = (byte >> (2 * e)) & 3u;
            v[e] = acc[mb][nb][4 * g + e] * ((code & 2u) ? 0.f : ((code & 1u) ? s1 : s2));
          }
        } else {
          const f32x4 ga = rd_unpack_bf16x4(gate[mb][G + u]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mb][nb][4 * g + e] * rd_gate_from_out(ga[e], use_drop);
        }
        lo[u] = rd_pack_bf16(v[0], v[1]); hi[u] = rd_pack_bf16(v[2], v[3]);
      }
      const auto sx = __builtin_amdgcn_permlane32_swap(lo[0], lo[1], false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(hi[0], hi[1], false, false);
      const u32x4_t o = {sx[0], sy[0], sx[1], sy[1]};
#ifdef RD_D2S_ABL_NOST               // (diagnostic builds: no output stores)
      if (o.x == 0x12345678u)
#endif
      if (ok) *(u32x4_t*)(op + G * 16) = o;
    }
  }
}

// gy [B][6][4][4][128] bf16 (gradient at layer 2's pre-activation... i.e. h->du[2]) -> gx [B][11][7][7][64] bf16
// = (sum over taps W[t]^T gy) * rd_gate_from_out(aux): LeakyReLU' x dropout factor read from aux = layer 1's stored output (layout
// of gx; +0.0 = dropped).
// grid: min((B + 1) / 2, 2 per CU) persistent workgroups of 256 threads; dynamic LDS RD_D2S_LDS.
__global__ void __launch_bounds__(256, 2)
k_d2_dgrad_slab16(const rd_bf16_t* __restrict__ gy, const rd_bf16_t* __restrict__ wimg, const rd_bf16_t* __restrict__ aux,
                  rd_bf16_t* __restrict__ gx, int B, int use_drop, const unsigned char* __restrict__ gbits = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  if (tid < 64) *(float*)(lds + RD_D2S_ZERO + tid * 4) = 0.f;
  const unsigned wvoff = (unsigned)lane * 16u;
  const int nitems = (B + RD_D2S_S - 1) / RD_D2S_S;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const long b0 = (long)item * RD_D2S_S;
    const int ns = min(RD_D2S_S, B - (int)b0);
    __syncthreads();                                  // every wave has left the previous item (and the zero row is in)
    {
      // 48 KB, contiguous: 48 DMA instructions of 1 KB (4 rows), 12 per wave; chunk swizzle c ^ (row & 15) on the source side
      const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(gy + b0 * (RD_D2S_SROWS * 128)));
#pragma unroll
      for (int k = 0; k < RD_D2S_S * RD_D2S_SROWS / 16; ++k) {
        const int i = wave * (RD_D2S_S * RD_D2S_SROWS / 16) + k;      // wave-uniform
        const int row = i * 4 + (lane >> 4);
        const int c_log = (lane & 15) ^ (row & 15);
        unsigned voff = row < ns * RD_D2S_SROWS ? (unsigned)(row * 256 + c_log * 16) : RD_OOB;
        asm volatile("" : "+v"(voff));
        rd_lds_dma16(rs, (float*)(lds + i * 1024), (int)voff, 0);
      }
    }
    rd_dma_landed();
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      const int desc = rd_d2s_tiles[wave][t];
      const int cls = desc & 15, row0 = (desc >> 4) & 255, mbs = desc >> 12;
      if (mbs == 0) break;
      if (mbs == 3) rd_d2s_tile<3>(lds, (const char*)wimg, wvoff, cls, row0, ns, b0, aux, gx, use_drop, l31, lhalf, gbits);
      else rd_d2s_tile<2>(lds, (const char*)wimg, wvoff, cls, row0, ns, b0, aux, gx, use_drop, l31, lhalf, gbits);
    }
  }
}


// ---- the same on tiles (round 4): domains larger than 16 x 16 (ndomain 32, 48, 64: multiples of 16; the large-domain variant
// L:291-293 has 11 x 31 x 31 x 64 -> 6 x 16 x 16 x 128).  There the streaming kernel ran this launch at 0.13 of the bf16 roof (1.65 of a
// 25.9 ms iteration at ndomain 64 / 64 samples).  Item = (pair of samples, tile of 8 x 8 destination positions); 2 x 150 output-gradient
// rows resident (77 KB: two workgroups per CU), everything else -- weight stream, operand swap, gate + rounding in registers -- is
// rd_d2s_tile.  Tiles of a phase over the item's two samples (interior tile: 5|6 x 4 x 4 positions per sample, 160 | 192 rows), dealt
// to the four waves so that K loops (taps x row blocks) AND epilogues (row blocks, ~3.3 tap-blocks each) balance:
// 24+6+4 / 11, 16+20+6 / 10, 20+12 / 11, 24+12 / 12.
__constant__ int rd_d2t_tiles[4][4] = {
  {RD_D2S_T(0, 0, 3), RD_D2S_T(7, 0, 3), RD_D2S_T(7, 96, 3), RD_D2S_T(3, 96, 2)},
  {RD_D2S_T(0, 96, 2), RD_D2S_T(1, 0, 3), RD_D2S_T(1, 96, 2), RD_D2S_T(3, 0, 3)},
  {RD_D2S_T(2, 0, 3), RD_D2S_T(2, 96, 2), RD_D2S_T(5, 0, 3), RD_D2S_T(5, 96, 3)},
  {RD_D2S_T(4, 0, 3), RD_D2S_T(4, 96, 3), RD_D2S_T(6, 0, 3), RD_D2S_T(6, 96, 3)},
};
// gy [B][6][OH][OW][128] bf16 -> gx [B][11][IH][IW][64] bf16, IH = 2 OH - 1 = 8 TI - 1 (IW alike); aux = layer 1's stored output.
// grid: persistent workgroups of 256 threads over ceil(B / 2) * TI * TJ items; dynamic LDS RD_D2T_LDS.
__global__ void __launch_bounds__(256, 2)
k_d2_dgrad_slab_t16(const rd_bf16_t* __restrict__ gy, const rd_bf16_t* __restrict__ wimg, const rd_bf16_t* __restrict__ aux,
                    rd_bf16_t* __restrict__ gx, int B, int OH, int OW, int use_drop, const unsigned char* __restrict__ gbits = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhalf = lane >> 5;
  if (tid < 64) *(float*)(lds + RD_D2T_ZERO + tid * 4) = 0.f;
  const unsigned wvoff = (unsigned)lane * 16u;
  const int TI = OH >> 2, TJ = OW >> 2;
  const int npair = (B + RD_D2S_S - 1) / RD_D2S_S, nitems = npair * TI * TJ;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int tj = item % TJ, ti = (item / TJ) % TI, pr = item / (TJ * TI);
    const long b0 = (long)pr * RD_D2S_S;
    const int ns = min(RD_D2S_S, B - (int)b0);
    RdD2sGeom G;
    G.last_h = ti == TI - 1; G.last_w = tj == TJ - 1; G.IH = 2 * OH - 1; G.IW = 2 * OW - 1; G.h0 = 8 * ti; G.w0 = 8 * tj;
    __syncthreads();                                  // every wave has left the previous item (and the zero row is in)
    {
      // 2 x 150 rows of 256 B: 75 DMA instructions of 1 KB (4 rows), 19 per wave; chunk swizzle c ^ (row & 15) on the source side
      const __amdgpu_buffer_rsrc_t rs = rd_make_rsrc((const float*)(gy + b0 * ((long)6 * OH * OW * 128)));
      int lq = lane;
      asm volatile("" : "+v"(lq));                    // (per-lane geometry recomputed per item, not kept in registers across the tiles)
#pragma unroll 1
      for (int k = 0; k < 19; ++k) {
        const int i = wave + 4 * k;                   // wave-uniform
        if (i < 75) {
          const int row = i * 4 + (lq >> 4);          // row of the image: sample s, position (od, i5, j5)
          const int sidx = row / RD_D2T_SROWS, rr = row - sidx * RD_D2T_SROWS;
          const int od = rr / 25, r2 = rr - od * 25, i5 = r2 / 5, j5 = r2 - i5 * 5;
          const int oh = 4 * ti + i5, ow = 4 * tj + j5;
          const int c_log = (lq & 15) ^ (row & 15);
          const bool ok = sidx < ns && oh < OH && ow < OW;
          unsigned voff = ok ? (unsigned)((((sidx * 6 + od) * OH + oh) * OW + ow) * 256 + c_log * 16) : RD_OOB;
          asm volatile("" : "+v"(voff));
          rd_lds_dma16(rs, (float*)(lds + i * 1024), (int)voff, 0);
        }
      }
    }
    rd_dma_landed();
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      const int desc = rd_d2t_tiles[wave][t];
      const int cls = desc & 15, row0 = (desc >> 4) & 255, mbs = desc >> 12;
      // (a tile that starts behind the phase's last row -- border tiles, a single sample -- has nothing to do)
      const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
      const int cnt = (5 + pd) * (G.last_h ? 3 + ph : 4) * (G.last_w ? 3 + pw : 4);
      if (row0 >= ns * cnt) continue;
      if (mbs == 3) rd_d2s_tile<3, true>(lds, (const char*)wimg, wvoff, cls, row0, ns, b0, aux, gx, use_drop, l31, lhalf, gbits, G);
      else rd_d2s_tile<2, true>(lds, (const char*)wimg, wvoff, cls, row0, ns, b0, aux, gx, use_drop, l31, lhalf, gbits, G);
    }
  }
}
