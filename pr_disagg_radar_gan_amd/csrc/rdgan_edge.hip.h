// Dedicated kernel for the generator's last Conv3D (64 -> 1, T:345), whose GEMM shape (K = 64, N = 27) is too thin for the
// tiled conv kernels: it is an HBM-bound stream over the block-3 output, and the tiled kernel spends its time in per-tile
// prologues instead.  (Round 2 also tried dedicated kernels for the critic's first Conv3D (2 -> 64 channels, K = 54): a
// sample-in-LDS version is bound by LDS broadcasts, versions feeding the FMAs from scalar loads run into SGPR spills (forward)
// or scalar-cache misses (weight gradient); none beat the implicit-GEMM path, so that layer stays on it.)
#pragma once
#include "rdgan_gemm_ws.hip.h"

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
  __syncthreads();                                            // (hipcc waits vmcnt(0) in front of the barrier: the tile has landed)

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
