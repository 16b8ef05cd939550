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
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#define RD_LRELU_ALPHA 0.2f
#ifndef RD_PRIO_LOAD
#define RD_PRIO_LOAD 2
#endif
#define RD_OOB 0x80000000u            // >= num_records of every descriptor below -> load returns 0
#define RD_RSRC_BYTES 0x7FFFFFF0u

// bf16 storage mode ("bf16" option): activations and activation gradients live in HBM as bf16 (rd_bf16_t), all
// arithmetic on them is fp32.  Overloaded accessors so that the elementwise kernels are written once for both types.
typedef unsigned short rd_bf16_t;
typedef __bf16 rd_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int rd_u32x2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 pair (round to nearest even, NaN stays NaN: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned rd_pack_bf16(float a, float b) {
  rd_bf16x2 r; r[0] = (__bf16)a; r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ f32x4 rd_unpack_bf16x4(rd_u32x2 u) {
  f32x4 v;
  v.x = __builtin_bit_cast(float, u.x << 16); v.y = __builtin_bit_cast(float, u.x & 0xFFFF0000u);
  v.z = __builtin_bit_cast(float, u.y << 16); v.w = __builtin_bit_cast(float, u.y & 0xFFFF0000u);
  return v;
}
__device__ __forceinline__ f32x4 rd_ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 rd_ld4(const rd_bf16_t* p) { return rd_unpack_bf16x4(*(const rd_u32x2*)p); }
__device__ __forceinline__ void rd_st4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void rd_st4(rd_bf16_t* p, f32x4 v) {
  rd_u32x2 o = {rd_pack_bf16(v.x, v.y), rd_pack_bf16(v.z, v.w)};
  *(rd_u32x2*)p = o;
}
__device__ __forceinline__ float rd_ld1(const float* p) { return *p; }
__device__ __forceinline__ float rd_ld1(const rd_bf16_t* p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
__device__ __forceinline__ void rd_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void rd_st1(rd_bf16_t* p, float v) { *p = __builtin_bit_cast(unsigned short, (__bf16)v); }

__device__ __forceinline__ float rd_lrelu(float x) { return x > 0.f ? x : RD_LRELU_ALPHA * x; }
// slope of LeakyReLU recovered from its (possibly dropout-scaled) output: TF's LeakyReluGrad
// uses features > 0 ? 1 : alpha, and sign(output) == sign(features) for kept elements.
__device__ __forceinline__ float rd_lrelu_slope_from_out(float h) { return h > 0.f ? 1.f : RD_LRELU_ALPHA; }
// Inverted dropout (T:288-300, rate 0.25) on a LeakyReLU output, leaving the mask readable in the stored value: a DROPPED element
// is stored as +0.0 whatever its sign was, a kept element that is exactly zero as -0.0.  The backward kernels then take the
// whole gate -- LeakyReLU' x mask x 1/0.75 -- from the stored activation (rd_gate_from_out) instead of hashing the element's
// counter again: two rd_mix32 = four quarter-rate v_mul_lo_u32 per element were the largest VALU item of every gating epilogue
// (0.18 ms of pure VALU time per sweep over critic layer 1's output at 6144 samples).  Same mask, same arithmetic as
// slope * rd_drop_scale; as a value -0.0 is 0.0 to everything downstream.  (bf16 storage: a kept value below 2^-133 would round
// to +0.0 and read as dropped; not reachable from this network's magnitudes.)
__device__ __forceinline__ float rd_drop_apply(float x, uint32_t key, uint32_t idx) {
  const float s = rd_drop_scale(key, idx);
  const float y = x * s;
  return s != 0.f ? (y == 0.f ? -0.0f : y) : 0.0f;
}
// the same for element idx0 + e of an aligned quad (idx0 % 4 == 0) whose hash word is `word` = rd_drop_word(key, idx0)
__device__ __forceinline__ float rd_drop_apply_w(float x, uint32_t word, int e) {
  const float y = x * (1.0f / 0.75f);
  return rd_drop_keep(word, (uint32_t)e) ? (y == 0.f ? -0.0f : y) : 0.0f;
}
// LeakyReLU'(features) x dropout factor from the layer's stored output h (TF: features > 0 ? 1 : alpha; dropout grad = mask/0.75)
__device__ __forceinline__ float rd_gate_from_out(float h, int use_drop) {
  float g = h > 0.f ? 1.f : RD_LRELU_ALPHA;
  if (use_drop) g = __builtin_bit_cast(unsigned, h) == 0u ? 0.f : g * (1.0f / 0.75f);
  return g;
}

__device__ __forceinline__ f32x4 rd_buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
}
// Gather load with the out-of-image case folded into the offset.  The empty asm pins the selected offset in a
// VGPR: without it hipcc (ROCm 7.2) turns load(ok ? off : OOB) into two exec-masked loads into the same
// registers with an s_waitcnt vmcnt(0) between them -- a full memory round trip per gathered row.
__device__ __forceinline__ f32x4 rd_buf_load4_if(__amdgpu_buffer_rsrc_t rsrc, bool ok, unsigned voff) {
  unsigned off = ok ? voff : RD_OOB;
  asm volatile("" : "+v"(off));
  return rd_buf_load4(rsrc, off);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rd_make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, RD_RSRC_BYTES, 0x00020000);
}
__device__ __forceinline__ void rd_buf_store4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, (int)voff, 0, 0);
}
// 4 bf16 (8 bytes) through a buffer descriptor, as fp32
__device__ __forceinline__ f32x4 rd_buf_load4_bf16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff = 0) {
  return rd_unpack_bf16x4(__builtin_bit_cast(rd_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, soff, 0)));
}
__device__ __forceinline__ void rd_buf_store4_bf16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, f32x4 v) {
  rd_u32x2 o = {rd_pack_bf16(v.x, v.y), rd_pack_bf16(v.z, v.w)};
  __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, (int)voff, 0, 0);
}
__device__ __forceinline__ void rd_buf_store1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)voff, 0, 0);
}
// sum over an aligned group of N lanes (N = 16: one DPP row, four rotate-adds; N = 32: plus one cross-row exchange)
template <int N>
__device__ __forceinline__ float rd_lanes_sum(float v) {
  static_assert(N == 16 || N == 32, "row group");
#define RD_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  RD_DPP_ADD(0x128); RD_DPP_ADD(0x124); RD_DPP_ADD(0x122); RD_DPP_ADD(0x121);   // row_ror:8, 4, 2, 1
#undef RD_DPP_ADD
  if (N == 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// s_shift == 1 (direct form of the folded upsample): per-row element delta of a tap from the 2-bit codes
__device__ __forceinline__ int rd_shift_delta(int w, int sd, int sh_, int sw, int SH, int SW, int cstride) {
  int dd = ((w >> sd) & 3) - 1, dh = ((w >> sh_) & 3) - 1, dw = ((w >> sw) & 3) - 1;
  return ((dd * SH + dh) * SW + dw) * cstride;
}

// The row tables are written by the host before any launch and never by a kernel: reading them through the
// constant address space lets hipcc use scalar loads (s_load) wherever the index is wave-uniform.
typedef const RdRow __attribute__((address_space(4)))* RdRowTab;
__device__ __forceinline__ RdRowTab rd_row_tab(const RdPlan* plan, int first) {
  return (RdRowTab)(unsigned long long)(plan->tab + first);
}
__device__ __forceinline__ RdRow rd_row(RdRowTab t, int l) {
  RdRow e;
  e.x = t[l].x; e.y = t[l].y; e.z = t[l].z; e.w = t[l].w;
  return e;
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

#ifdef RD_STAMP
// diagnostic build only: per-segment cycle totals of the K loop (never compiled into the shipped library)
__device__ unsigned long long rd_stamp_acc[8];
__device__ __forceinline__ unsigned long long rd_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif

// ------------------------------------------------------------------------------------
// C[m][n] = sum_{tap,c} A_gather[m][tap][c] * W[tap_w*wrpt + c][n]  (+ fused epilogue)
// ------------------------------------------------------------------------------------
// SRC16 / OUT16 (bf16 storage mode, the few GEMMs that stay on the fp32 matrix pipe there): the gathered tensor is bf16
// (converted to fp32 on its way into LDS) / the destination and the epi.aux / epi.addt tensors are bf16.  Arithmetic
// unchanged.  OUT16 launches never split K.
template <int BM, int BN, int WM, int WN, int BK, bool PARTIAL, bool SHIFT, bool SRC16 = false, bool OUT16 = false>
__global__ void __launch_bounds__(256)
k_conv_gemm(const RdPlan* __restrict__ plan, int B, const float* __restrict__ src,
            const float* __restrict__ W, int ldw, float* dst, RdEpi epi) {
  // PARTIAL: SC % 4 != 0 (the last float4 of a tap is masked element-wise); SHIFT: s_shift == 1 (direct
  // form of the folded nearest upsample).  Both are compile-time so the hot loop has no uniform branches.
  static_assert(WM * WN == 4, "4 waves");
  static_assert(!(SRC16 && (SHIFT || PARTIAL)), "bf16 source: clean plans only");
  constexpr int AESZ = SRC16 ? 2 : 4;         // bytes per gathered element
  constexpr unsigned OSZ = OUT16 ? 2u : 4u;   // bytes per destination / aux / addt element
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
  constexpr int TG = 8;                       // taps per group whose gather offsets live in registers
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhalf = lane >> 5;
#ifdef RD_STAMP
  const unsigned long long st_kernel0 = rd_stamp();
#endif

  // ---- which phase / tile (all wave-uniform); 1-D grid, N tile fastest so both N tiles of an M tile run together
  const int NTn = plan->N / BN;
  // phases of unequal length (stride-2 input gradients: 8, 4, 4, 2, 4, 2, 2, 1 taps) are laid out one after the other:
  // a contiguous range per XCD would hand one XCD all the 8-tap tiles, so those plans keep the round-robin dealing
  const int swz = (plan->nphases > 1 && !plan->interleave) ? (int)blockIdx.x : rd_xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = swz % NTn;
  int mt = swz / NTn, pidx = 0;
  if (plan->interleave) {
    pidx = mt % plan->nphases;
    mt /= plan->nphases;
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
  const int SH = plan->SH, SW = plan->SW;
  const int cstride = plan->s_cstride, SC = plan->SC, wrpt = plan->w_rows_per_tap;
  const int ssample = (int)plan->src_sample;
  const int ntaps = P.ntaps;
  const RdRowTab tab = rd_row_tab(plan, P.tab);
  const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc((const float*)((const char*)src + (long)b0 * plan->src_sample * AESZ));
  const __amdgpu_buffer_rsrc_t rsB = rd_make_rsrc(W + P.w_off);

  // ---- per-thread A rows: byte offset relative to sample b0 (incl. this thread's channel group), validity bits
  // (branch-free: the A_P row-table loads are issued back to back and waited for once)
  int roff[A_P], rbits[A_P], rcode[A_P];
  const int a_c4 = (tid % A_F4) * 4;
  {
    int rl[A_P], rb_[A_P];
    bool rok[A_P];
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int r = tid / A_F4 + i * A_RPP;
      int l = l0 + r, bb = 0;
      if (L >= BM) { if (l >= L) { l -= L; bb = 1; } }
      else { bb = l / L; l -= bb * L; }
      rok[i] = r < BM && m0 + r < rows;
      rl[i] = rok[i] ? l : 0;
      rb_[i] = bb;
    }
    RdRow e[A_P];
#pragma unroll
    for (int i = 0; i < A_P; ++i) e[i] = rd_row(tab, rl[i]);
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      roff[i] = rok[i] ? (rb_[i] * ssample + e[i].x + a_c4) * AESZ : 0;
      rbits[i] = rok[i] ? e[i].y : 0;
      rcode[i] = rok[i] ? e[i].w : 0;
    }
  }
  // ---- per-thread B (weight) rows
  int boff[B_P];
  const int b_kk = tid / B_F4, b_n4 = (tid % B_F4) * 4;
#pragma unroll
  for (int i = 0; i < B_P; ++i) boff[i] = ((b_kk + i * B_RPP) * ldw + n0 + b_n4) * 4;

  const int CPT = (SC + BK - 1) / BK;
  const bool k_tail = (SC % BK) != 0;         // the last channel chunk of a tap is partial (Dense 356, column GEMM 27)
  // split-K: this workgroup multiplies chunks [q0, q0 + nchunks) of the ntaps*CPT chunks of the K loop
  const int nch_all = ntaps * CPT;
  const int ksplit = epi.ksplit > 1 ? epi.ksplit : 1;
  const int per_split = (nch_all + ksplit - 1) / ksplit;
  const int q0 = (int)blockIdx.y * per_split;
  const int nchunks = max(0, min(nch_all, q0 + per_split) - q0);

  // K order: tap group (8 taps) outer, channel chunk, tap inner -- the taps of one chunk re-read the same 128-byte
  // pixel segments of neighbouring rows.  For the taps of the current group the byte offset of every gathered row
  // sits in registers with the image-border test already folded in (out-of-image -> an out-of-range offset that the
  // buffer bounds check turns into zeros), and the channel-chunk offset rides in the load's scalar offset: the hot
  // loop issues ONE instruction per gathered row.  (In-kernel stamps: with a partner wave streaming 64-cycle MFMAs
  // on the SIMD every dependent short instruction of the load phase waits for an MFMA slot -- ~100 instructions
  // cost ~4000 cycles per chunk against 700 alone -- so instruction count is what matters here.)
  unsigned voffs[A_P][TG];
  int tapw[TG];
  auto build_group = [&](int g) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const int tap = min(g * TG + t, ntaps - 1);
      const RdTap ti = P.tap[tap];
      tapw[t] = ti.w * wrpt * ldw * 4;
      const int sd = ti.code & 255, sh_ = (ti.code >> 8) & 255, sw = ti.code >> 16;
#pragma unroll
      for (int i = 0; i < A_P; ++i) {
        int off = roff[i] + (SHIFT ? rd_shift_delta(rcode[i], sd, sh_, sw, SH, SW, cstride) * 4 : (SRC16 ? ti.delta >> 1 : ti.delta));
        voffs[i][t] = ((rbits[i] & ti.mask) == ti.mask) ? (unsigned)off : RD_OOB;
      }
    }
  };
  // position of the NEXT chunk to load: group, channel chunk, tap inside the group
  int ld_g, ld_cc, ld_t, ld_gt;
  {
    const int full = TG * CPT;
    ld_g = q0 / full;
    const int rem = q0 - ld_g * full;
    ld_gt = min(TG, ntaps - ld_g * TG);
    ld_cc = ld_gt > 0 ? rem / ld_gt : 0;
    ld_t = ld_gt > 0 ? rem - ld_cc * ld_gt : 0;
  }
  if (nchunks > 0) build_group(ld_g);

  f32x4 ra[A_P], rw[B_P];
  auto issue_loads = [&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
    const int sA = ld_cc * BK * AESZ;
    const bool tail = k_tail && ld_cc == CPT - 1;
    const int c = ld_cc * BK + a_c4;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      unsigned voff = voffs[i][t];
      if (tail && c >= SC) voff = RD_OOB;                      // uniform `tail` is false in every hot launch
      f32x4 v;
      if constexpr (SRC16) v = rd_buf_load4_bf16(rsA, voff, sA);
      else v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)voff, sA, 0));
      if (PARTIAL) {
        if (c + 1 >= SC) v.y = 0.f;
        if (c + 2 >= SC) v.z = 0.f;
        if (c + 3 >= SC) v.w = 0.f;
      }
      ra[i] = v;
    }
    const int sB = tapw[t] + ld_cc * BK * ldw * 4;
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      const int kk = b_kk + i * B_RPP;
      unsigned voff = (unsigned)boff[i];
      if ((BK % B_RPP != 0 && kk >= BK) || (tail && ld_cc * BK + kk >= SC)) voff = RD_OOB;
      rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)voff, sB, 0));
    }
  };
  auto load_chunk = [&]() {
    switch (ld_t) {                                             // wave-uniform: selects the register column
      case 0: issue_loads(std::integral_constant<int, 0>{}); break;
      case 1: issue_loads(std::integral_constant<int, 1>{}); break;
      case 2: issue_loads(std::integral_constant<int, 2>{}); break;
      case 3: issue_loads(std::integral_constant<int, 3>{}); break;
      case 4: issue_loads(std::integral_constant<int, 4>{}); break;
      case 5: issue_loads(std::integral_constant<int, 5>{}); break;
      case 6: issue_loads(std::integral_constant<int, 6>{}); break;
      default: issue_loads(std::integral_constant<int, 7>{}); break;
    }
    if (++ld_t == ld_gt) {
      ld_t = 0;
      if (++ld_cc == CPT) {
        ld_cc = 0;
        ++ld_g;
        ld_gt = min(TG, ntaps - ld_g * TG);
        if (ld_gt > 0) build_group(ld_g);                       // once per 8*CPT chunks; never for <= 8 taps
      }
    }
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
#ifdef RD_STAMP
  unsigned long long st_load = 0, st_mfma = 0, st_store = 0, st_bar = 0, st_t0 = rd_stamp();
  const unsigned long long st_begin = st_t0;
#endif
  for (int q = 0; q < nchunks; ++q) {
    const int buf = q & 1;
    if (q + 1 < nchunks) load_chunk();
#ifdef RD_STAMP
    { unsigned long long t = rd_stamp(); st_load += t - st_t0; st_t0 = t; }
#endif
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
#ifdef RD_MFMA16_TIMING
          {   // timing-only diagnostic: two 16x16x4 MFMAs (32 cycles each) instead of one 32x32x2 (64 cycles); results are WRONG
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
            f32x4_ c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            f32x4_ c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][s], fb[cur][s][j], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][s], fb[cur][s][j], c1, 0, 0, 0);
            acc[i][j][0] = c0[0]; acc[i][j][1] = c0[1]; acc[i][j][2] = c0[2]; acc[i][j][3] = c0[3];
            acc[i][j][4] = c1[0]; acc[i][j][5] = c1[1]; acc[i][j][6] = c1[2]; acc[i][j][7] = c1[3];
          }
#else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][s], fb[cur][s][j], acc[i][j], 0, 0, 0);
#endif
    }
#ifdef RD_STAMP
    { unsigned long long t = rd_stamp(); st_mfma += t - st_t0; st_t0 = t; }
#endif
    if (q + 1 < nchunks) store_chunk(buf ^ 1);
#ifdef RD_STAMP
    { unsigned long long t = rd_stamp(); st_store += t - st_t0; st_t0 = t; }
#endif
    __syncthreads();
#ifdef RD_STAMP
    { unsigned long long t = rd_stamp(); st_bar += t - st_t0; st_t0 = t; }
#endif
  }
#ifdef RD_STAMP
  const unsigned long long st_loop_end = st_t0;
  if (lane == 0 && BM == 256) {
    atomicAdd(&rd_stamp_acc[6], st_begin - st_kernel0);
    atomicAdd(&rd_stamp_acc[0], st_load); atomicAdd(&rd_stamp_acc[1], st_mfma); atomicAdd(&rd_stamp_acc[2], st_store);
    atomicAdd(&rd_stamp_acc[3], st_bar); atomicAdd(&rd_stamp_acc[4], st_t0 - st_begin); atomicAdd(&rd_stamp_acc[5], 1ull);
  }
#endif

  // ---- epilogue
  const long dsample = plan->dst_sample;
  const int mode = epi.mode;
  // LDS_EPI: the accumulator tile goes through the (now idle) staging LDS so that every thread stores whole float4s of
  // one output row: BM*BN/1024 coalesced dwordx4 stores + 1 row-table load per thread instead of 16*TM*TN scalar
  // stores + 16*TM dependent table loads (VMEM instructions issue slowly beside a partner wave's MFMA stream; stamps
  // put the scalar epilogue at 21 % of a workgroup's lifetime).
  constexpr bool LDS_EPI = BK == 32;
  if constexpr (LDS_EPI && BN == 32) {
    if (mode == RD_EPI_TAPGATHER) {
      // rows padded to 33 floats: the gather below walks rows with consecutive lanes
      constexpr int CST = BN + 1;
      float* Cs = smem;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
#pragma unroll
          for (int j = 0; j < TN; ++j) Cs[row * CST + wn * WTN + j * 32 + l31] = acc[i][j][r];
        }
      __syncthreads();
      const int Wd = epi.gw, HW = epi.ghw, NQ = epi.gq, Hd = HW / Wd;
      for (int o = tid; o < BM * NQ; o += 256) {
        const int r = o % BM, j = o / BM;
        const long m = (long)m0 + r;
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
        dst[(pl * NQ + j) * HW + hw] = s;
      }
      return;
    }
  }
  if constexpr (LDS_EPI) {
    // same lean row loop as k_conv_gemm_ws: 32-bit byte offsets inside buffer windows based at sample b0, rows without a
    // destination carry RD_OOB (loads return 0, stores are dropped), v_rsq and DPP row sums for PixelNorm
    float* Cs = smem;                                   // [BM][BN]
    unsigned* Rb = (unsigned*)(smem + BM * BN);         // [BM] byte offset of the row in the destination window
    unsigned* Tb = Rb + BM;                             // [BM] byte offset of the row in the epi.addt window
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
#pragma unroll
        for (int j = 0; j < TN; ++j) Cs[row * BN + wn * WTN + j * 32 + l31] = acc[i][j][r];
      }
    const int dsmp = (int)dsample;
    if (tid < BM) {
      unsigned rb = RD_OOB, tb = RD_OOB;
      if (m0 + tid < rows) {
        int l = l0 + tid, bb = 0;
        if (L >= BM) { if (l >= L) { l -= L; bb = 1; } }
        else { bb = l / L; l -= bb * L; }
        const int z = tab[l].z;
        rb = (unsigned)(bb * dsmp + z) * OSZ;
        if (epi.addt) tb = (unsigned)(bb * (dsmp >> 1) + z - ((z / epi.addt_plane + 1) >> 1) * epi.addt_plane) * OSZ;
      }
      Rb[tid] = rb; Tb[tid] = tb;
    }
    __syncthreads();
    constexpr int F4R = BN / 4;                         // float4s per row; 256 % F4R == 0, so a thread's columns are fixed
    constexpr int RPP = 256 / F4R;                      // rows per pass
    const int c4 = (tid % F4R) * 4;
    const unsigned colb = (unsigned)(n0 + c4) * OSZ;
    const long dbase = (long)b0 * dsample;
    // destination-typed accessors of this epilogue (bf16 storage mode: 8-byte accesses)
    auto ld_act = [](__amdgpu_buffer_rsrc_t r, unsigned off) -> f32x4 {
      if constexpr (OUT16) return rd_buf_load4_bf16(r, off); else return rd_buf_load4(r, off);
    };
    auto st_act = [](__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
      if constexpr (OUT16) rd_buf_store4_bf16(r, off, v); else rd_buf_store4(r, off, v);
    };
    auto act_base = [](const float* p, long elems) -> const float* { return (const float*)((const char*)p + elems * (long)OSZ); };
    if (!OUT16 && ksplit > 1) {
      const __amdgpu_buffer_rsrc_t rsK = rd_make_rsrc(epi.kpart + (long)blockIdx.y * epi.kstride + dbase);
#pragma unroll 4
      for (int row = tid / F4R; row < BM; row += RPP)
        rd_buf_store4(rsK, Rb[row] + colb, *(const f32x4*)&Cs[row * BN + c4]);
      return;
    }
    const __amdgpu_buffer_rsrc_t rsD = rd_make_rsrc(act_base(dst, dbase));
    const __amdgpu_buffer_rsrc_t rsT = rd_make_rsrc(epi.addt ? act_base(epi.addt, (long)b0 * (dsmp >> 1)) : dst);
    const bool has_t = epi.addt != nullptr;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (mode == RD_EPI_BIAS || mode == RD_EPI_BIAS_LRELU || mode == RD_EPI_BIAS_LRELU_DROP || mode == RD_EPI_BIAS_PN_LRELU)
      bias4 = *(const f32x4*)(epi.bias + n0 + c4);
    bool pn_done = false;
    if constexpr (F4R >= 16) {
      if (mode == RD_EPI_BIAS_PN_LRELU) {
        // PixelNormalization (T:255-266) + LeakyReLU (T:333): the F4R lanes holding a row are an aligned lane group
        pn_done = true;
        const __amdgpu_buffer_rsrc_t rsR = rd_make_rsrc(epi.rinv ? epi.rinv + dbase / BN : dst);
        const unsigned rmask = (c4 == 0 && epi.rinv) ? 0u : RD_OOB;      // one lane per row stores 1/l2
#pragma unroll 4
        for (int row = tid / F4R; row < BM; row += RPP) {
          const unsigned rb = Rb[row];
          f32x4 v = *(const f32x4*)&Cs[row * BN + c4];
          if (has_t) v += ld_act(rsT, Tb[row] + colb);
          v += bias4;
          const float ss = rd_lanes_sum<F4R>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
          const float ri = __builtin_amdgcn_rsqf(ss * (1.0f / BN) + 1.0e-8f);      // v_rsq_f32: 1 ulp
          v *= ri;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], RD_LRELU_ALPHA * v[e]);     // LeakyReLU for alpha < 1
          rd_buf_store1(rsR, ((rb / BN) * (4u / OSZ)) | (rb & RD_OOB) | rmask, ri);
          st_act(rsD, rb + colb, v);
        }
      }
    }
    if (!pn_done) {
      const __amdgpu_buffer_rsrc_t rsX = rd_make_rsrc(mode == RD_EPI_GATE_AUX ? act_base(epi.aux, dbase) : dst);
      const uint32_t ibase = (uint32_t)dbase + epi.idx_base + (uint32_t)(n0 + c4);
#pragma unroll 4
      for (int row = tid / F4R; row < BM; row += RPP) {
        const unsigned rb = Rb[row];
        f32x4 v = *(const f32x4*)&Cs[row * BN + c4];
        if (has_t) v += ld_act(rsT, Tb[row] + colb);
        if (mode == RD_EPI_BIAS) {
          v += bias4;
        } else if (mode == RD_EPI_BIAS_LRELU || mode == RD_EPI_BIAS_LRELU_DROP) {
          v += bias4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = rd_lrelu(v[e]);
            if (mode == RD_EPI_BIAS_LRELU_DROP && epi.use_drop) x = rd_drop_apply_w(x, rd_drop_word(epi.key, ibase + (rb / OSZ)), e);
            v[e] = x;
          }
        } else if (mode == RD_EPI_GATE_AUX) {
          const f32x4 a4 = ld_act(rsX, rb + colb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = rd_gate_from_out(a4[e], epi.use_drop);
            v[e] *= g;
          }
        }
        st_act(rsD, rb + colb, v);
      }
    }
  } else {
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
          if (!OUT16 && ksplit > 1) {
            epi.kpart[(long)blockIdx.y * epi.kstride + idx] = v;
            continue;
          }
          if (mode == RD_EPI_BIAS) {
            v += epi.bias[col];
          } else if (mode == RD_EPI_BIAS_LRELU) {
            v = rd_lrelu(v + epi.bias[col]);
          } else if (mode == RD_EPI_BIAS_LRELU_DROP) {
            v = rd_lrelu(v + epi.bias[col]);
            if (epi.use_drop) v = rd_drop_apply(v, epi.key, (uint32_t)idx + epi.idx_base);
          } else if (mode == RD_EPI_GATE_AUX) {
            const float g = rd_gate_from_out(OUT16 ? rd_ld1((const rd_bf16_t*)epi.aux + idx) : epi.aux[idx], epi.use_drop);
            v *= g;
          }
          if constexpr (OUT16) rd_st1((rd_bf16_t*)dst + idx, v); else dst[idx] = v;
        }
      }
    }
  }
  }
#ifdef RD_STAMP
  { unsigned long long t = rd_stamp(); if (lane == 0 && BM == 256) atomicAdd(&rd_stamp_acc[7], t - st_loop_end); }
#endif
}

// split-K finish: dst[idx] = epilogue(sum_s kpart[s][idx]); the destination is dense with N floats per pixel
// (OUT16: dst and epi.aux are bf16 -- the bf16 storage mode; the partial slabs are always fp32)
template <bool OUT16 = false>
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
        if (epi.use_drop) x = rd_drop_apply_w(x, rd_drop_word(epi.key, (uint32_t)idx0 + epi.idx_base), e);
      } else if (mode == RD_EPI_GATE_AUX) {
        const float g = rd_gate_from_out(OUT16 ? rd_ld1((const rd_bf16_t*)epi.aux + idx) : epi.aux[idx], epi.use_drop);
        x *= g;
      }
      v[e] = x;
    }
    if constexpr (OUT16) rd_st4((rd_bf16_t*)dst + idx0, v); else *(f32x4*)(dst + idx0) = v;
  }
}

// ------------------------------------------------------------------------------------
// weight gradient: dW[tap_w*wrpt + c][n] = sum_m A_gather[m][tap][c] * dY[m][n]
// grid.x = r_tile * NT + n_tile, grid.y = split over rows m, grid.z = plan phase; partial sums go to
// `partial[phase][split][RT*BR][N]` and are folded by k_wgrad_reduce (deterministic, no atomics).
// ------------------------------------------------------------------------------------
// box plans: workgroup wg -> phase bz, split by, row tile rt, n tile; slab = index of the first partial slab of (bz, by)
__device__ __forceinline__ void rd_wgrad_box_decode(const RdPlan* __restrict__ plan, const RdWgradTiling& T, int B, int wg,
                                                    int& bz, int& by, int& rt, int& ntile, int& slab) {
  int prefix = 0;
  bz = 0; by = 0; rt = 0; ntile = 0; slab = 0;
  const int np = plan->nphases;
  for (int p = 0; p < np; ++p) {
    const int rtp = rd_wgrad_phase_rt(T, plan->phT[p]);
    const int nsp = (B * plan->phL[p] + (1 << T.rps_log2) - 1) >> T.rps_log2;
    const int tiles = rtp * T.NT;
    const int cnt = tiles * nsp;
    if (wg < cnt) {
      bz = p; by = wg / tiles;
      const int bx = wg - by * tiles;
      rt = bx / T.NT; ntile = bx - rt * T.NT;
      slab = prefix + by * rtp;
      return;
    }
    wg -= cnt; prefix += rtp * nsp;
  }
}

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

// DY16 (bf16 storage mode, first critic layer): the output gradient dy is bf16, the gathered tensor fp32.
template <int BR, int BN, bool PARTIAL, bool SHIFT, bool DY16 = false>
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
  const RdRowTab tab = rd_row_tab(plan, P.tab);
  // descriptors are based at the first sample this block touches (wave-uniform)
  const int bb0 = mbeg / L;
  const __amdgpu_buffer_rsrc_t rsA = rd_make_rsrc(src + (long)bb0 * plan->src_sample);
  constexpr int GSZ = DY16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rsB = rd_make_rsrc((const float*)((const char*)dy + (long)bb0 * plan->dst_sample * GSZ));

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
  const int b_const = (n0 + (tid % B_F4) * 4) * GSZ;

  // row cursors (sample index relative to bb0, row inside the sample) of this thread's A and B positions.
  // With BR = 256 a whole wave gathers the same positions (A_F4 == 64), so the A cursors, the row-table lookups and the
  // offset arithmetic are wave-uniform: readfirstlane makes that provable and they run on the scalar unit (SMEM/SALU)
  // instead of costing vector-memory instructions next to the partner wave's MFMA stream.
  constexpr bool UNI = A_F4 == 64;
  const int a_pos0 = UNI ? __builtin_amdgcn_readfirstlane(tid / A_F4) : tid / A_F4;
  int ab[A_P], al[A_P], gb[B_P], gl[B_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) { int m = mbeg + a_pos0 + i * A_PPP; ab[i] = m / L; al[i] = m - ab[i] * L; ab[i] -= bb0; }
#pragma unroll
  for (int i = 0; i < B_P; ++i) { int m = mbeg + tid / B_F4 + i * B_PPP; gb[i] = m / L; gl[i] = m - gb[i] * L; gb[i] -= bb0; }

  // row-table entries of the NEXT chunk's rows, fetched one load_chunk call ahead so the gather never waits
  // on a table load it has just issued
  RdRow ea[A_P];
  int ez[B_P];
  auto fetch_rows = [&]() {
#pragma unroll
    for (int i = 0; i < A_P; ++i) ea[i] = rd_row(tab, al[i]);
#pragma unroll
    for (int i = 0; i < B_P; ++i) ez[i] = tab[gl[i]].z;
  };
  fetch_rows();

  f32x4 ra[A_P], rg[B_P];
  auto load_chunk = [&](int mb) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      int m = mb + a_pos0 + i * A_PPP;
      const RdRow e = ea[i];
      int off = (ab[i] * ssample + e.x) * 4 + a_const;
      if (SHIFT) off += rd_shift_delta(e.w, sd, sh_, sw, SH, SW, cstride) * 4;
      f32x4 v = rd_buf_load4_if(rsA, m < mend && (e.y & tmask) == tmask, (unsigned)off);
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
      if constexpr (DY16) {
        unsigned off = m < mend ? (unsigned)((gb[i] * dsample + ez[i]) * 2 + b_const) : RD_OOB;
        asm volatile("" : "+v"(off));
        rg[i] = rd_buf_load4_bf16(rsB, off);
      } else
      rg[i] = rd_buf_load4_if(rsB, m < mend, (unsigned)((gb[i] * dsample + ez[i]) * 4 + b_const));
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
    __builtin_amdgcn_s_setprio(RD_PRIO_LOAD);          // see k_conv_gemm
    if (q + 1 < nchunks) load_chunk(mbeg + (q + 1) * BKP);
    __builtin_amdgcn_s_setprio(0);
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
    __builtin_amdgcn_s_setprio(RD_PRIO_LOAD);
    if (q + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  const int N = plan->N;
  if (T.direct) {
    // one phase, one split: the (tap, c) rows of the tile straight to dW (rows past the tap's channels do not exist there)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        int tap, c;
        rd_wgrad_tile_row(T, BR, rt, row, tap, c);
        if (tap < P.ntaps && c < SC) {
          float* o = partial + ((long)P.tap[tap].w * plan->w_rows_per_tap + c) * T.ldw + n0 + wn * WTN + l31;
#pragma unroll
          for (int j = 0; j < TN; ++j) o[j * 32] = acc[i][j][r];
        }
      }
    return;
  }
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

// fold the split partials and scatter rows (tap,c) to their place in the weight gradient.
// 256 threads = `outs` output float4s x (256/outs) slices of the split range; the slices are folded through LDS in a
// fixed order (deterministic).  Few outputs with many splits (D1, the last generator conv) take outs = 16.
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const RdPlan* __restrict__ plan, const float* __restrict__ partial, int nsplit,
               RdWgradTiling T, int BR, float* __restrict__ dW, int ldw, int outs) {
  __shared__ f32x4 red[256];
  const int N = plan->N;
  const int n4s = N / 4;
  const long total = (long)T.RT * BR * n4s;
  const RdPhase& P = plan->ph[blockIdx.y];
  const long stride = (long)T.RT * BR * N;
  const float* pbase = partial + (long)blockIdx.y * nsplit * stride;
  const int ks = 256 / outs, o = threadIdx.x % outs, sl = threadIdx.x / outs;
  const long f = (long)blockIdx.x * outs + o;
  int tap = 0, c = 0, n = 0;
  bool ok = f < total;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    int R = (int)(f / n4s);
    n = (int)(f - (long)R * n4s) * 4;
    int rt = R / BR, r = R - rt * BR;
    rd_wgrad_tile_row(T, BR, rt, r, tap, c);
    ok = tap < P.ntaps && c < plan->SC;
    if (ok) {
      const float* p = pbase + (long)R * N + n;
      for (int k = sl; k < nsplit; k += ks) s += *(const f32x4*)(p + k * stride);
    }
  }
  if (ks > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0)
      for (int j = 1; j < ks; ++j) s += red[j * outs + o];
  }
  if (ok && sl == 0) *(f32x4*)(dW + P.w_off + ((long)P.tap[tap].w * plan->w_rows_per_tap + c) * ldw + n) = s;
}

// The fold for border-class boxes: one thread per (weight tap w, channel c, four columns) adds, phase by phase and split by split
// in a fixed order, the partial slabs of every phase that lists tap w; a tap no phase lists (it never lands inside the picture)
// gets its exact zero.  nw = weight taps to look at (at most 64); RdPlan::wmask says which of them this plan owns.
__global__ void __launch_bounds__(256)
k_wgrad_reduce_box(const RdPlan* __restrict__ plan, const float* __restrict__ partial, RdWgradTiling T, int BR, int B, int nw,
                   float* __restrict__ dW, int ldw) {
  const int N = plan->N, n4s = N / 4, SC = plan->SC;
  const long f = (long)blockIdx.x * 256 + threadIdx.x;
  if (f >= (long)nw * SC * n4s) return;
  const int n = (int)(f % n4s) * 4;
  const int c = (int)((f / n4s) % SC), w = (int)(f / ((long)n4s * SC));
  if (!((plan->wmask >> w) & 1ull)) return;          // not a tap of this plan (another launch owns that row block of dW)
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int prefix = 0;
  const int np = plan->nphases;
  for (int p = 0; p < np; ++p) {
    const int rtp = rd_wgrad_phase_rt(T, plan->phT[p]);
    const int nsp = (B * plan->phL[p] + (1 << T.rps_log2) - 1) >> T.rps_log2;
    const int k = plan->tapinv[p][w];
    if (k >= 0) {
      int rt, r;
      if (T.tiles_per_tap > 0) { const int ct = c / BR; rt = k * T.tiles_per_tap + ct; r = c - ct * BR; }
      else { rt = k >> T.tpt_log2; r = (k & (T.taps_per_tile - 1)) * T.cw + c; }
      const float* q = partial + ((long)(prefix + rt) * BR + r) * N + n;
      const long st = (long)rtp * BR * N;
      int by = 0;
      for (; by + 1 < nsp; by += 2) { s0 += *(const f32x4*)(q + by * st); s1 += *(const f32x4*)(q + (by + 1) * st); }
      if (by < nsp) s0 += *(const f32x4*)(q + by * st);
    }
    prefix += rtp * nsp;
  }
  *(f32x4*)(dW + plan->ph[0].w_off + ((long)w * plan->w_rows_per_tap + c) * ldw + n) = s0 + s1;
}
