// Host-side planner of librdgan_hip.so: network geometry, gather plans (rdgan_plan.h), row tables, weight-gradient tilings and
// the partial-slab workspace bound.  Plain C++17 -- no HIP call, no device pointer is dereferenced -- so that this translation
// unit also compiles with g++ -fsanitize=address,undefined on a box without a GPU: tests/host/plan_check.cpp builds every plan
// of every supported configuration under the sanitizers and checks the coverage / bounds invariants the kernels rely on
// (tests/test_host_plan.py).  rdgan_api.hip includes this header and adds the launches.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "rdgan_plan.h"

#ifndef RDGAN_NHOURS
#define RDGAN_NHOURS 24
#endif
#ifndef RDGAN_LATENT_DIM
#define RDGAN_LATENT_DIM 100
#endif

static int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

// ------------------------------------------------------------------------------------
// plan builders
// ------------------------------------------------------------------------------------
static void phase_defaults(RdPhase& ph, int LD, int LH, int LW) {
  memset(&ph, 0, sizeof(ph));
  ph.LD = LD; ph.LH = LH; ph.LW = LW; ph.L = LD * LH * LW;
  for (int a = 0; a < 3; ++a) { ph.s_mul[a] = 1; ph.o_mul[a] = 1; ph.o_off[a] = 0; }
}

// Conv3D forward (T:286-299, T:331-345); D,H,W = source extents before the folded upsample
static RdPlan plan_conv_fwd(int D, int H, int W, int Cin, int Cout, int Do, int Ho, int Wo, int stride,
                            int pd, int ph_, int pw, int up) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 1; p.SD = D; p.SH = H; p.SW = W; p.s_shift = up; p.s_cstride = Cin; p.SC = Cin;
  p.w_rows_per_tap = Cin; p.DD = Do; p.DH = Ho; p.DW = Wo; p.d_cstride = Cout; p.N = Cout;
  RdPhase& q = p.ph[0];
  phase_defaults(q, Do, Ho, Wo);
  for (int a = 0; a < 3; ++a) q.s_mul[a] = stride;
  q.ntaps = 27;
  for (int t = 0; t < 27; ++t) {
    q.tap_off[t][0] = (int8_t)(t / 9 - pd); q.tap_off[t][1] = (int8_t)((t / 3) % 3 - ph_);
    q.tap_off[t][2] = (int8_t)(t % 3 - pw); q.tap[t].w = t;
  }
  return p;
}

// A plan with its loop spaces cut into boxes by BORDER CLASS (option "border_boxes").  A strided 'same' conv on a small grid
// multiplies many zeros: a tap that leaves the picture is a zero row of the implicit GEMM, and on the critic's 6x4x4 / 3x2x2 /
// 2x1x1 output grids that is 38 % / 38 % / 70 % of all (position, tap) pairs (T:291-299), the same share in the input gradients.
// Per phase and axis the loop indices fall into runs with the same set of valid tap offsets (first / middle / last); a box = a
// product of such runs, a phase of its own with the parent's weights and only the taps valid somewhere in it (rows for which a
// listed tap is invalid keep their zero row through the validity mask), so the products that remain are the same, in the same tap
// order: results identical to the parent plan up to the K-split boundaries.  More than RD_MAX_PHASES boxes are merged greedily
// (same parent phase, adjacent pair with the least added (row, tap) work first).  Phases come out longest tap list first (the
// launchers run unequal phases in that order).  Returns the plan unchanged when no box drops a tap.
static RdPlan plan_boxes(const RdPlan& src) {
  if (src.s_shift) return src;
  const int S[3] = {src.SD, src.SH, src.SW};
  struct Box { int parent, lo[3], cnt[3]; unsigned mask[3]; };
  struct Run { int lo, cnt; unsigned mask; };
  auto ntaps_of = [&](const Box& b) {
    const RdPhase& q = src.ph[b.parent];
    int n = 0;
    for (int t = 0; t < q.ntaps; ++t) {
      bool ok = true;
      for (int a = 0; a < 3; ++a) ok = ok && ((b.mask[a] >> (q.tap_off[t][a] + 1)) & 1u);
      n += ok;
    }
    return n;
  };
  auto cost = [&](const Box& b) { return (long)b.cnt[0] * b.cnt[1] * b.cnt[2] * ntaps_of(b); };
  std::vector<Box> boxes;
  long full = 0;
  for (int pi = 0; pi < src.nphases; ++pi) {
    const RdPhase& q = src.ph[pi];
    const int LL[3] = {q.LD, q.LH, q.LW};
    full += (long)q.L * q.ntaps;
    std::vector<Run> runs[3];
    for (int a = 0; a < 3; ++a) {
      unsigned used = 0;
      for (int t = 0; t < q.ntaps; ++t) used |= 1u << (q.tap_off[t][a] + 1);
      for (int l = 0; l < LL[a]; ++l) {
        unsigned m = 0;
        for (int off = -1; off <= 2; ++off) {
          const int v = l * q.s_mul[a] + q.s_off[a] + off;
          if (((used >> (off + 1)) & 1u) && v >= 0 && v < S[a]) m |= 1u << (off + 1);
        }
        if (!runs[a].empty() && runs[a].back().mask == m) runs[a].back().cnt++;
        else runs[a].push_back({l, 1, m});
      }
    }
    for (const Run& rd : runs[0]) for (const Run& rh : runs[1]) for (const Run& rw : runs[2])
      boxes.push_back({pi, {rd.lo, rh.lo, rw.lo}, {rd.cnt, rh.cnt, rw.cnt}, {rd.mask, rh.mask, rw.mask}});
  }
  while ((int)boxes.size() > RD_MAX_PHASES) {
    long best = -1; size_t bi = 0, bj = 0; Box bm{};
    for (size_t i = 0; i < boxes.size(); ++i)
      for (size_t j = 0; j < boxes.size(); ++j) {
        if (i == j || boxes[i].parent != boxes[j].parent) continue;
        for (int a = 0; a < 3; ++a) {
          bool ok = boxes[i].lo[a] + boxes[i].cnt[a] == boxes[j].lo[a];
          for (int x = 0; x < 3; ++x) if (x != a && (boxes[i].lo[x] != boxes[j].lo[x] || boxes[i].cnt[x] != boxes[j].cnt[x])) ok = false;
          if (!ok) continue;
          Box m = boxes[i];
          for (int x = 0; x < 3; ++x) m.mask[x] |= boxes[j].mask[x];
          m.cnt[a] += boxes[j].cnt[a];
          const long inc = cost(m) - cost(boxes[i]) - cost(boxes[j]);
          if (best < 0 || inc < best) { best = inc; bi = i; bj = j; bm = m; }
        }
      }
    if (best < 0) return src;
    boxes.erase(boxes.begin() + std::max(bi, bj)); boxes.erase(boxes.begin() + std::min(bi, bj));
    boxes.push_back(bm);
  }
  long left = 0;
  for (const Box& b : boxes) left += cost(b);
  if (left >= full) return src;                       // nothing to skip
  std::stable_sort(boxes.begin(), boxes.end(), [&](const Box& x, const Box& y) { return ntaps_of(x) > ntaps_of(y); });
  RdPlan p = src;
  p.nphases = (int)boxes.size();
  p.boxes = 1;
  p.wmask = 0;
  for (int pi = 0; pi < src.nphases; ++pi)
    for (int t = 0; t < src.ph[pi].ntaps; ++t) if (src.ph[pi].tap[t].w >= 0 && src.ph[pi].tap[t].w < 64) p.wmask |= 1ull << src.ph[pi].tap[t].w;
  for (int i = 0; i < p.nphases; ++i) {
    const Box& b = boxes[i];
    const RdPhase& par = src.ph[b.parent];
    RdPhase& q = p.ph[i];
    q = par;
    q.LD = b.cnt[0]; q.LH = b.cnt[1]; q.LW = b.cnt[2]; q.L = q.LD * q.LH * q.LW;
    for (int a = 0; a < 3; ++a) { q.s_off[a] = par.s_off[a] + b.lo[a] * par.s_mul[a]; q.o_off[a] = par.o_off[a] + b.lo[a] * par.o_mul[a]; }
    q.ntaps = 0;
    for (int t = 0; t < par.ntaps; ++t) {
      bool ok = true;
      for (int a = 0; a < 3; ++a) ok = ok && ((b.mask[a] >> (par.tap_off[t][a] + 1)) & 1u);
      if (!ok) continue;
      const int n = q.ntaps++;
      for (int a = 0; a < 4; ++a) q.tap_off[n][a] = par.tap_off[t][a];
      q.tap[n] = par.tap[t];
    }
    if (q.ntaps == 0) {       // (a box no tap reaches still owns its output rows: one tap, every row of it masked)
      for (int a = 0; a < 4; ++a) q.tap_off[0][a] = par.tap_off[0][a];
      q.tap[0] = par.tap[0]; q.ntaps = 1;
    }
  }
  return p;
}
static RdPlan plan_conv_fwd_boxes(int D, int H, int W, int Cin, int Cout, int Do, int Ho, int Wo, int stride,
                                  int pd, int ph_, int pw) {
  return plan_boxes(plan_conv_fwd(D, H, W, Cin, Cout, Do, Ho, Wo, stride, pd, ph_, pw, 0));
}

// D1 (T:286): 2-channel input, stride 2, 'valid'.  (kw, ci) is contiguous in NDHWC with C = 2, so the
// 27 x CP taps are gathered as 9 taps (kd,kh) x 3*CP contiguous floats (CP = floats per voxel: 2, or 4 with the
// extra condition channels of the revision-1 variants).
static RdPlan plan_d1_fwd(int nd, int Do, int Ho, int Wo, int CP) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 1; p.SD = RDGAN_NHOURS; p.SH = nd; p.SW = nd; p.s_shift = 0; p.s_cstride = CP; p.SC = 3 * CP;
  p.w_rows_per_tap = 3 * CP; p.DD = Do; p.DH = Ho; p.DW = Wo; p.d_cstride = 64; p.N = 64;
  RdPhase& q = p.ph[0];
  phase_defaults(q, Do, Ho, Wo);
  for (int a = 0; a < 3; ++a) q.s_mul[a] = 2;
  q.ntaps = 9;
  for (int t = 0; t < 9; ++t) {
    q.tap_off[t][0] = (int8_t)(t / 3); q.tap_off[t][1] = (int8_t)(t % 3); q.tap_off[t][2] = 0;
    q.tap[t].w = t;
  }
  return p;
}

// input gradient of a stride-1 'same' conv on its own grid: gx[i] = sum_t W[t]^T gy[i + 1 - t]
static RdPlan plan_conv_dgrad_s1(int D, int H, int W, int Cin, int Cout) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 1; p.SD = D; p.SH = H; p.SW = W; p.s_shift = 0; p.s_cstride = Cout; p.SC = Cout;
  p.w_rows_per_tap = Cout; p.DD = D; p.DH = H; p.DW = W; p.d_cstride = Cin; p.N = Cin;
  RdPhase& q = p.ph[0];
  phase_defaults(q, D, H, W);
  q.ntaps = 27;
  for (int t = 0; t < 27; ++t) {
    q.tap_off[t][0] = (int8_t)(1 - t / 9); q.tap_off[t][1] = (int8_t)(1 - (t / 3) % 3);
    q.tap_off[t][2] = (int8_t)(1 - t % 3); q.tap[t].w = t;
  }
  return p;
}

// input gradient of a stride-2 conv, by parity phases: input position i = 2l + c0 receives
// W[t]^T gy[o] for 2o + t - pad = i, i.e. taps t = pi + 2j (pi = (i+pad)&1) at o = l + base - j.
static RdPlan plan_conv_dgrad_s2(int D, int H, int W, int Cin, int Do, int Ho, int Wo, int Cout, const int pad[3]) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.SD = Do; p.SH = Ho; p.SW = Wo; p.s_shift = 0; p.s_cstride = Cout; p.SC = Cout; p.w_rows_per_tap = Cout;
  p.DD = D; p.DH = H; p.DW = W; p.d_cstride = Cin; p.N = Cin;
  const int n[3] = {D, H, W};
  int np = 0;
  for (int cls = 0; cls < 8; ++cls) {
    int pi[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
    int c0[3], cnt[3], base[3], nt[3];
    bool empty = false;
    for (int a = 0; a < 3; ++a) {
      c0[a] = (pi[a] + pad[a]) & 1;
      cnt[a] = n[a] > c0[a] ? (n[a] - c0[a] + 1) / 2 : 0;
      base[a] = (c0[a] + pad[a] - pi[a]) / 2;
      nt[a] = pi[a] == 0 ? 2 : 1;
      if (cnt[a] == 0) empty = true;
    }
    if (empty) continue;
    RdPhase& q = p.ph[np++];
    phase_defaults(q, cnt[0], cnt[1], cnt[2]);
    for (int a = 0; a < 3; ++a) { q.o_mul[a] = 2; q.o_off[a] = c0[a]; }
    q.ntaps = 0;
    for (int jd = 0; jd < nt[0]; ++jd)
      for (int jh = 0; jh < nt[1]; ++jh)
        for (int jw = 0; jw < nt[2]; ++jw) {
          int k = q.ntaps++;
          q.tap_off[k][0] = (int8_t)(base[0] - jd); q.tap_off[k][1] = (int8_t)(base[1] - jh);
          q.tap_off[k][2] = (int8_t)(base[2] - jw);
          q.tap[k].w = ((pi[0] + 2 * jd) * 3 + (pi[1] + 2 * jh)) * 3 + (pi[2] + 2 * jw);
        }
  }
  p.nphases = np;
  return p;
}

// UpSampling3D(2)+Conv3D 3^3 'same' collapsed onto the un-upsampled grid:
// 8 output-parity phases x 8 taps; weights Wc[phase*8 + tap][Cin][Cout] from k_collapse_weights.
static RdPlan plan_upconv_fwd_collapsed(int D, int H, int W, int Cin, int Cout) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 8; p.SD = D; p.SH = H; p.SW = W; p.s_shift = 0; p.s_cstride = Cin; p.SC = Cin;
  p.w_rows_per_tap = Cin; p.DD = 2 * D; p.DH = 2 * H; p.DW = 2 * W; p.d_cstride = Cout; p.N = Cout;
  for (int ph = 0; ph < 8; ++ph) {
    RdPhase& q = p.ph[ph];
    phase_defaults(q, D, H, W);
    const int par[3] = {ph >> 2, (ph >> 1) & 1, ph & 1};
    for (int a = 0; a < 3; ++a) { q.o_mul[a] = 2; q.o_off[a] = par[a]; }
    q.ntaps = 8;
    for (int t = 0; t < 8; ++t) {
      q.tap_off[t][0] = (int8_t)(par[0] - 1 + (t >> 2)); q.tap_off[t][1] = (int8_t)(par[1] - 1 + ((t >> 1) & 1));
      q.tap_off[t][2] = (int8_t)(par[2] - 1 + (t & 1));
      q.tap[t].w = ph * 8 + t;
    }
  }
  return p;
}
// its input gradient straight onto the un-upsampled grid (upsample adjoint included): per axis the four
// upsampled-grid positions o = 2j + q - 1, q = 0..3, i.e. (phase,tap) = (1,1),(0,1),(1,0),(0,0);
// weights Wd[q3][Cout][Cin] from k_transpose_map.  D,H,W = un-upsampled extents.
static void collapsed_dgrad_slice_map(int16_t map[64]) {
  const int qp[4] = {1, 0, 1, 0}, qa[4] = {1, 1, 0, 0};
  for (int q = 0; q < 64; ++q) {
    int qd = q >> 4, qh = (q >> 2) & 3, qw = q & 3;
    int ph = qp[qd] * 4 + qp[qh] * 2 + qp[qw], tp = qa[qd] * 4 + qa[qh] * 2 + qa[qw];
    map[q] = (int16_t)(ph * 8 + tp);
  }
}
static RdPlan plan_upconv_dgrad_collapsed(int D, int H, int W, int Cin, int Cout) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 1; p.SD = 2 * D; p.SH = 2 * H; p.SW = 2 * W; p.s_shift = 0; p.s_cstride = Cout; p.SC = Cout;
  p.w_rows_per_tap = Cout; p.DD = D; p.DH = H; p.DW = W; p.d_cstride = Cin; p.N = Cin;
  RdPhase& q = p.ph[0];
  phase_defaults(q, D, H, W);
  for (int a = 0; a < 3; ++a) q.s_mul[a] = 2;
  q.ntaps = 64;
  for (int t = 0; t < 64; ++t) {
    q.tap_off[t][0] = (int8_t)((t >> 4) - 1); q.tap_off[t][1] = (int8_t)(((t >> 2) & 3) - 1);
    q.tap_off[t][2] = (int8_t)((t & 3) - 1); q.tap[t].w = t;
  }
  return p;
}

// ---- shared-centre form along d for the backward pass of a generator block (rdgan_elem.hip.h).
// U block index: u = g*16 + (ph*2+th)*4 + (pw*2+tw), g = 0 (A': -W0 on E[s]), 1 (S: W0+W1+W2 on x[s]), 2 (D: W2 on E[s+1]).
// On a collapsed axis (p,t) reads source offset p-1+t and sums the kernel taps {0},{1,2},{0,1},{2}.
static void fastd_weight_map(RdWeightMap& T) {
  static const int td[3][3] = {{-1, 0, 0}, {1, 1, 1}, {0, 0, 1}};
  static const int tc[4][3] = {{1, 0, 0}, {0, 1, 1}, {1, 1, 0}, {0, 0, 1}};
  memset(&T, 0, sizeof(T));
  for (int g = 0; g < 3; ++g)
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 27; ++k)
          T.c[g * 16 + a * 4 + b][k] = (int8_t)(td[g][k / 9] * tc[a][(k / 3) % 3] * tc[b][k % 3]);
}
// weight gradient of group g: A = E at j = s against the even output planes, S = x against the plane sums gS,
// D = E at j = s+1 against the odd output planes.  D,H,W = un-upsampled extents.
static RdPlan plan_fastd_wgrad(int D, int H, int W, int Cin, int Cout, int g) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 4; p.SD = g == 1 ? D : D + 1; p.SH = H; p.SW = W; p.s_shift = 0; p.s_cstride = Cin; p.SC = Cin;
  p.w_rows_per_tap = Cin; p.DD = g == 1 ? D : 2 * D; p.DH = 2 * H; p.DW = 2 * W; p.d_cstride = Cout; p.N = Cout;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      RdPhase& q = p.ph[ph * 2 + pw];
      phase_defaults(q, D, H, W);
      q.o_mul[0] = g == 1 ? 1 : 2; q.o_off[0] = g == 2 ? 1 : 0;
      q.o_mul[1] = 2; q.o_off[1] = ph; q.o_mul[2] = 2; q.o_off[2] = pw;
      q.ntaps = 4;
      for (int t = 0; t < 4; ++t) {
        const int th = t >> 1, tw = t & 1;
        q.tap_off[t][0] = (int8_t)(g == 2 ? 1 : 0); q.tap_off[t][1] = (int8_t)(ph - 1 + th); q.tap_off[t][2] = (int8_t)(pw - 1 + tw);
        q.tap[t].w = g * 16 + (ph * 2 + th) * 4 + (pw * 2 + tw);
      }
    }
  return p;
}
// forward, difference part: E (D+1,H,W) -> out (2D,2H,2W); phase (pd,ph,pw): pd = 0 reads E[s] with A' = -W0,
// pd = 1 reads E[s+1] with D = W2; the shared S x[s] part comes from plan_fastd_wgrad(g = 1) run as a forward plan
// into T (one hour plane per output plane pair) and is added in the epilogue (RdEpi::addt)
static RdPlan plan_fastd_fwd_e(int D, int H, int W, int Cin, int Cout) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 8; p.SD = D + 1; p.SH = H; p.SW = W; p.s_shift = 0; p.s_cstride = Cin; p.SC = Cin;
  p.w_rows_per_tap = Cin; p.DD = 2 * D; p.DH = 2 * H; p.DW = 2 * W; p.d_cstride = Cout; p.N = Cout;
  for (int i = 0; i < 8; ++i) {
    const int pd = i >> 2, ph = (i >> 1) & 1, pw = i & 1;
    RdPhase& q = p.ph[i];
    phase_defaults(q, D, H, W);
    for (int a = 0; a < 3; ++a) q.o_mul[a] = 2;
    q.o_off[0] = pd; q.o_off[1] = ph; q.o_off[2] = pw;
    q.ntaps = 4;
    for (int t = 0; t < 4; ++t) {
      const int th = t >> 1, tw = t & 1;
      q.tap_off[t][0] = (int8_t)pd; q.tap_off[t][1] = (int8_t)(ph - 1 + th); q.tap_off[t][2] = (int8_t)(pw - 1 + tw);
      q.tap[t].w = (pd ? 32 : 0) + (ph * 2 + th) * 4 + (pw * 2 + tw);
    }
  }
  return p;
}
// input gradient, shared-centre part: gS (D,2H,2W) -> dxS (D,H,W), 16 taps (positions 2u+q-1 on h and w); weights UT[0..16)
static RdPlan plan_fastd_dgrad_s(int D, int H, int W, int Cin, int Cout) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 1; p.SD = D; p.SH = 2 * H; p.SW = 2 * W; p.s_shift = 0; p.s_cstride = Cout; p.SC = Cout;
  p.w_rows_per_tap = Cout; p.DD = D; p.DH = H; p.DW = W; p.d_cstride = Cin; p.N = Cin;
  RdPhase& q = p.ph[0];
  phase_defaults(q, D, H, W);
  q.s_mul[1] = 2; q.s_mul[2] = 2;
  q.ntaps = 16;
  for (int t = 0; t < 16; ++t) {
    q.tap_off[t][0] = 0; q.tap_off[t][1] = (int8_t)((t >> 2) - 1); q.tap_off[t][2] = (int8_t)((t & 3) - 1);
    q.tap[t].w = t;
  }
  return p;
}
// input gradient, difference part: g (2D,2H,2W) -> dE (D+1,H,W): E[j] feeds output plane 2j through A' and plane 2j-1
// through D; 32 taps, weights UT[16..48)
static RdPlan plan_fastd_dgrad_e(int D, int H, int W, int Cin, int Cout) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 1; p.SD = 2 * D; p.SH = 2 * H; p.SW = 2 * W; p.s_shift = 0; p.s_cstride = Cout; p.SC = Cout;
  p.w_rows_per_tap = Cout; p.DD = D + 1; p.DH = H; p.DW = W; p.d_cstride = Cin; p.N = Cin;
  RdPhase& q = p.ph[0];
  phase_defaults(q, D + 1, H, W);
  for (int a = 0; a < 3; ++a) q.s_mul[a] = 2;
  q.ntaps = 32;
  for (int t = 0; t < 32; ++t) {
    q.tap_off[t][0] = (int8_t)(t < 16 ? 0 : -1); q.tap_off[t][1] = (int8_t)(((t >> 2) & 3) - 1); q.tap_off[t][2] = (int8_t)((t & 3) - 1);
    q.tap[t].w = 16 + t;
  }
  return p;
}
// UT[slice] = U[map[slice]]^T for the two plans above: per axis position q = 0..3 is (p,t) = (1,1),(0,1),(1,0),(0,0)
static void fastd_dgrad_slice_map(int16_t map[64]) {
  const int qi[4] = {3, 1, 2, 0};    // p*2 + t
  for (int i = 0; i < 64; ++i) map[i] = 0;
  for (int t = 0; t < 16; ++t) {
    const int hw = qi[t >> 2] * 4 + qi[t & 3];
    map[t] = (int16_t)(16 + hw); map[16 + t] = (int16_t)hw; map[32 + t] = (int16_t)(32 + hw);
  }
}

// plain row GEMM: rows (D,H,W) of `cstride` floats, first SC used -> [rows][dcs], N columns
static RdPlan plan_rows(int D, int H, int W, int SC, int cstride, int N, int dcs) {
  RdPlan p; memset(&p, 0, sizeof(p));
  p.nphases = 1; p.SD = D; p.SH = H; p.SW = W; p.s_shift = 0; p.s_cstride = cstride; p.SC = SC;
  p.w_rows_per_tap = SC; p.DD = D; p.DH = H; p.DW = W; p.d_cstride = dcs; p.N = N;
  RdPhase& q = p.ph[0];
  phase_defaults(q, D, H, W);
  q.ntaps = 1;
  return p;
}

// Row tables + per-tap scalars (see RdRow in rdgan_plan.h).  Appends this plan's rows to `out` and records
// each phase's first entry in ph.tab (relative to the plan's own table start).  Returns false if a tap
// offset falls outside the [-1, 2] range the 12-bit validity mask encodes.
static bool plan_build_tables(RdPlan& p, std::vector<RdRow>& out) {
  const int S[3] = {p.SD, p.SH, p.SW};
  const int Dd[3] = {p.DD, p.DH, p.DW};
  p.interleave = p.nphases > 1;
  for (int pi = 1; pi < p.nphases; ++pi)
    if (p.ph[pi].L != p.ph[0].L || p.ph[pi].ntaps != p.ph[0].ntaps) p.interleave = 0;   // (unequal tap counts: longest phases first)
  p.src_sample = (long)p.SD * p.SH * p.SW * p.s_cstride;
  p.dst_sample = (long)p.DD * p.DH * p.DW * p.d_cstride;
  int first = 0;
  for (int pi = 0; pi < p.nphases; ++pi) {
    RdPhase& q = p.ph[pi];
    q.tab = first;
    p.phL[pi] = q.L;
    p.phT[pi] = q.ntaps;
    for (int w = 0; w < 64; ++w) p.tapinv[pi][w] = -1;
    for (int t = 0; t < q.ntaps; ++t) if (q.tap[t].w >= 0 && q.tap[t].w < 64) p.tapinv[pi][q.tap[t].w] = (signed char)t;
    for (int t = 0; t < q.ntaps; ++t) {
      int mask = 0;
      for (int a = 0; a < 3; ++a) {
        int off = q.tap_off[t][a];
        if (off < -1 || off > 2 || (p.s_shift && off > 1)) return false;
        mask |= 1 << (a * 4 + off + 1);
      }
      q.tap[t].mask = mask;
      q.tap[t].code = ((q.tap_off[t][0] + 1) * 2) | ((6 + (q.tap_off[t][1] + 1) * 2) << 8) | ((12 + (q.tap_off[t][2] + 1) * 2) << 16);
      q.tap[t].delta = ((q.tap_off[t][0] * p.SH + q.tap_off[t][1]) * p.SW + q.tap_off[t][2]) * p.s_cstride * 4;
    }
    const int LL[3] = {q.LD, q.LH, q.LW};
    for (int ld = 0; ld < q.LD; ++ld)
      for (int lh = 0; lh < q.LH; ++lh)
        for (int lw = 0; lw < q.LW; ++lw) {
          const int l[3] = {ld, lh, lw};
          RdRow e = {0, 0, 0, 0};
          long so = 0, dof = 0;
          for (int a = 0; a < 3; ++a) {
            int pre = l[a] * q.s_mul[a] + q.s_off[a];
            so = so * S[a] + (pre >> p.s_shift);
            dof = dof * Dd[a] + (l[a] * q.o_mul[a] + q.o_off[a]);
            for (int off = -1; off <= 2; ++off) {
              int v = pre + off;
              if (v >= 0 && v < (S[a] << p.s_shift)) e.y |= 1 << (a * 4 + off + 1);
            }
            if (p.s_shift)
              for (int off = -1; off <= 1; ++off) {
                int code = (((pre + off) >> 1) - (pre >> 1)) + 1;      // arithmetic shift: (-1)>>1 = -1
                e.w |= (code & 3) << (a * 6 + (off + 1) * 2);
              }
          }
          e.x = (int)(so * p.s_cstride);
          e.z = (int)(dof * p.d_cstride);
          out.push_back(e);
        }
    (void)LL;
    first += q.L;
  }
  return true;
}

// ------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------
enum {
  PL_GDENSE = 0, PL_G1F, PL_G2F, PL_G3F, PL_G9F, PL_G1B, PL_G2B, PL_G3B, PL_G9B,
  PL_D1F, PL_D2F, PL_D3F, PL_D4F, PL_D2B, PL_D3B, PL_D4B, PL_D1B,
  PL_G1FC, PL_G2FC, PL_G3FC, PL_G1BC, PL_G2BC, PL_G3BC,
  PL_F1WA, PL_F2WA, PL_F3WA, PL_F1WS, PL_F2WS, PL_F3WS, PL_F1WD, PL_F2WD, PL_F3WD,   // shared-centre backward (fast_bwd)
  PL_F1BS, PL_F2BS, PL_F3BS, PL_F1BE, PL_F2BE, PL_F3BE, PL_F1FE, PL_F2FE, PL_F3FE,
  PL_D2FX, PL_D3FX, PL_D4FX,          // critic layers 2-4 forward, output grid cut into border-class boxes (plan_boxes)
  PL_D2BX, PL_D3BX, PL_D4BX,          // their input gradients, every parity phase cut the same way
  PL_F1WAX, PL_F2WAX, PL_F3WAX, PL_F1WSX, PL_F2WSX, PL_F3WSX, PL_F1WDX, PL_F2WDX, PL_F3WDX,   // shared-centre weight gradients on boxes
  PL_G1FCX, PL_G2FCX, PL_G3FCX, PL_G1BCX, PL_G2BCX, PL_G3BCX,     // collapsed generator blocks, forward / input gradient, on boxes
  PL_GDENSE16,                        // the Dense layer with its K padded to the bf16 GEMM's chunk of 64
  PL_COUNT
};

static double plan_flops(const RdPlan& p, int B) {
  double rt = 0;
  for (int i = 0; i < p.nphases; ++i) rt += (double)B * p.ph[i].L * p.ph[i].ntaps;
  return 2.0 * rt * p.SC * p.N;
}

static long plan_tiles(const RdPlan& p, int B, int BM) {
  long t = 0;
  for (int i = 0; i < p.nphases; ++i) t += ((long)B * p.ph[i].L + BM - 1) / BM;
  return t;
}

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// workgroups a weight-gradient launch aims for (two rounds of the 512 slots of 256 CUs x 2); RDGAN_WGRAD_WGS: diagnostic override
static long wgrad_target_wgs() {
  static const long v = [] { const char* e = getenv("RDGAN_WGRAD_WGS"); long x = e ? atol(e) : 0; return x >= 64 && x <= 8192 ? x : 1024L; }();
  return v;
}
// wide16 (bf16 kernels only, k_wgrad_gemm_ws16<256, 128>, option "wgrad_wide", default off: measured slower): N % 128 == 0 layers
// with plenty of rows take 256 x 128 tiles on three stages of 48 KB, one workgroup per CU
static RdWgradTiling wgrad_tiling(const RdPlan& p, int B, int& BR, int& BN, int& nsplit, bool wide16 = false) {
  RdWgradTiling T; memset(&T, 0, sizeof(T));
  const RdPhase& q = p.ph[0];
  BR = p.SC >= 128 ? 128 : 64;
  BN = (p.N % 128 == 0) ? 128 : 64;
  // 64 input channels against >= 128 output channels (critic layer 2): two taps per 128-row tile, so every wave owns a
  // 64x64 tile (4 fragment reads per 4 MFMAs instead of 3 per 2)
  if (p.SC == 64 && BN == 128 && q.ntaps >= 2 && (p.nphases == 1 || p.boxes) && !p.s_shift) BR = 128;
  if (BN == 64 && p.SC == 128 && q.ntaps % 2 == 0 && (long)B * q.L >= 65536) BR = 256;   // two taps per tile, 4 accumulators per wave
  if (wide16 && BR == 128 && BN == 128 && !p.s_shift && p.SC % 64 == 0 && (p.SC >= 256 ? p.SC % 256 == 0 : true)) {
    long rows_all = 0, taps_min = q.ntaps;
    for (int i = 0; i < p.nphases; ++i) { rows_all += (long)B * p.ph[i].L; taps_min = std::min<long>(taps_min, p.ph[i].ntaps); }
    if (rows_all >= 32768 && taps_min * p.SC >= 256) BR = 256;
  }
  if (p.SC >= BR) {
    T.tiles_per_tap = (p.SC + BR - 1) / BR; T.cw = BR; T.taps_per_tile = 1; T.RT = q.ntaps * T.tiles_per_tap;
  } else {
    T.cw = next_pow2(std::max(p.SC, 4)); T.taps_per_tile = BR / T.cw; T.tiles_per_tap = 0;
    T.RT = (q.ntaps + T.taps_per_tile - 1) / T.taps_per_tile;
  }
  T.NT = p.N / BN;
  T.tpt_log2 = T.taps_per_tile >= 4 ? 2 : (T.taps_per_tile >= 2 ? 1 : 0);
  if (p.boxes) {
    // border-class boxes: ONE power-of-two row count per split for all phases, the smallest that keeps the launch at about a
    // thousand workgroups; nsplit = workgroups in all, RT = partial slabs in all (what the callers size and launch with)
    T.box = 1;
    for (int lg = 5; lg < 31; ++lg) {
      long wgs = 0, slabs = 0;
      for (int i = 0; i < p.nphases; ++i) {
        const long nsp = ((long)B * p.ph[i].L + (1L << lg) - 1) >> lg;
        const long rtp = rd_wgrad_phase_rt(T, p.ph[i].ntaps);
        wgs += rtp * T.NT * nsp; slabs += rtp * nsp;
      }
      T.rps_log2 = lg; T.rows_per_split = 1 << lg; T.nsplit = (int)wgs; T.RT = (int)slabs;
      if (wgs <= wgrad_target_wgs() * 5 / 4) break;
    }
    T.nphases = p.nphases;
    nsplit = T.nsplit;
    return T;
  }
  long rows = (long)B * q.L;
  long tiles = (long)T.RT * T.NT * p.nphases;
  long want = std::max(1L, (wgrad_target_wgs() + tiles - 1) / tiles);
  long maxs = std::max(1L, (rows + 127) / 128);
  long s = std::min(want, maxs);
  long rps = (rows + s - 1) / s;
  rps = (rps + 31) / 32 * 32;
  T.rows_per_split = (int)rps;
  nsplit = (int)((rows + rps - 1) / rps);
  T.nsplit = nsplit; T.nphases = p.nphases;
  return T;
}

static size_t wgrad_partial_need(const RdPlan& hp, int B, bool wide16 = false) {
  int BR, BN, nsplit;
  RdWgradTiling T = wgrad_tiling(hp, B, BR, BN, nsplit, wide16);
  if (T.box) return (size_t)T.RT * BR * hp.N;
  return (size_t)hp.nphases * nsplit * T.RT * BR * hp.N;
}
// border-class boxes in a weight gradient: the producer/consumer kernels (k_wgrad_gemm_ws / ws16), clean plans
static bool wgrad_box_ok(const RdPlan& hp) {
  if (!hp.boxes || hp.s_shift || (hp.SC & 3) || hp.N % 64) return false;
  for (int i = 1; i < hp.nphases; ++i) if (hp.ph[i].w_off != hp.ph[0].w_off) return false;     // one weight block, taps told apart by tap.w
  return true;
}

// Upper bound of wgrad_partial_need(hp, B) over every B in [1, maxB] (the need of a box plan is NOT monotone in
// B -- rps_log2 is chosen per call).  Box plans: at the smallest split (lg = 5) the need grows with B, and whenever a larger lg
// is chosen the launch has at most wgrad_target_wgs() * 5 / 4 workgroups = that many BR x BN slab tiles; the need is below
// both.  One-phase / congruent plans: nsplit <= want, which depends on B only through the 128 -> 256 row-tile switch, so the
// bound is taken over both tile choices.  tests/host/plan_check.cpp compares it with the need of EVERY B.
static size_t wgrad_partial_bound1(const RdPlan& hp, int maxB, bool wide16);
static size_t wgrad_partial_bound(const RdPlan& hp, int maxB) {      // (both tilings of the bf16 kernels: one workspace)
  return std::max(wgrad_partial_bound1(hp, maxB, false), wgrad_partial_bound1(hp, maxB, true));
}
static size_t wgrad_partial_bound1(const RdPlan& hp, int maxB, bool wide16) {
  int BR, BN, nsplit;
  RdWgradTiling T = wgrad_tiling(hp, maxB, BR, BN, nsplit, wide16);
  if (T.box) {
    size_t slabs5 = 0;
    for (int i = 0; i < hp.nphases; ++i)
      slabs5 += (size_t)rd_wgrad_phase_rt(T, hp.ph[i].ntaps) * (size_t)(((long)maxB * hp.ph[i].L + 31) >> 5);
    size_t slabs1 = 0;                                  // one split per phase: what the largest lg leaves
    for (int i = 0; i < hp.nphases; ++i) slabs1 += (size_t)rd_wgrad_phase_rt(T, hp.ph[i].ntaps);
    const size_t at5 = slabs5 * BR * hp.N;
    const size_t capped = std::max((size_t)(wgrad_target_wgs() * 5 / 4) * BR * BN, slabs1 * BR * hp.N);
    return std::min(at5, capped);
  }
  size_t bound = 0;
  for (int B : {1, maxB}) {          // (the two row-tile choices: below / above the 65536-row switch)
    T = wgrad_tiling(hp, B, BR, BN, nsplit, wide16);
    const long tiles = (long)T.RT * T.NT * hp.nphases;
    const long want = std::max(1L, (wgrad_target_wgs() + tiles - 1) / tiles);
    const long maxs = std::max(1L, ((long)maxB * hp.ph[0].L + 127) / 128);
    bound = std::max(bound, (size_t)hp.nphases * (size_t)std::min(want, maxs) * T.RT * BR * hp.N);
  }
  return bound;
}

static void tf_same(int n, int& out, int& before) {
  out = (n + 1) / 2;
  int total = std::max((out - 1) * 2 + 3 - n, 0);
  before = total / 2;   // the extra pad goes at the END
}

// ------------------------------------------------------------------------------------
// network geometry (T:286-299, T:318-345; L:317-364 for ndomain 64) and parameter layout (Keras get_weights() order)
// ------------------------------------------------------------------------------------
struct RdGeom {
  int nd = 0, s = 0, MB = 0, NB = 0;
  int nc = 0, Cin = 0, CP = 0, ldp1 = 0;   // condition channels, critic input channels 1+nc, floats per input voxel, P1 columns
  long goff[10], gsz[10], doff[10], dsz[10], n_gen = 0, n_critic = 0;
  int n_in = 0, n_nodes = 0;
  int gdim[4][3];              // generator grids: h0, h1, h2, h3
  long gpix[4];
  int gch[4];                  // channels of h0..h3
  int ddim[5][3];              // critic: input grid + 4 conv outputs
  long dL[5];
  int dch[5];
  int dpad[4][3];
  int F = 0;                   // critic Dense fan-in
  int KP0 = 0;                 // the Dense layer's K padded to the bf16 GEMM's chunk of 64
  bool dense16_ok = false;
};

static bool rd_geometry_ok(int ndomain, int n_cond_channels, int max_batch) {
  return !(ndomain < 8 || ndomain % 8 || ndomain > 120 || n_cond_channels < 1 || n_cond_channels > 3 || max_batch < 1);
}

static void rd_geometry(RdGeom* h, int ndomain, int n_cond_channels, int max_batch) {
  h->nd = ndomain; h->s = ndomain / 8; h->MB = max_batch; h->NB = 3 * max_batch;
  h->nc = n_cond_channels; h->Cin = 1 + n_cond_channels; h->CP = n_cond_channels == 1 ? 2 : 4;
  h->ldp1 = (27 * h->Cin + 63) / 64 * 64;
  const int nd = ndomain, s = h->s;
  h->n_in = RDGAN_LATENT_DIM + nd * nd * n_cond_channels;   // T:322-323
  h->n_nodes = 256 * s * s * 3;                    // T:318, L:325
  // generator grids (T:328-341)
  const int gch[4] = {256, 256, 128, 64};
  for (int l = 0; l < 4; ++l) {
    h->gdim[l][0] = 3 << l; h->gdim[l][1] = s << l; h->gdim[l][2] = s << l;
    h->gpix[l] = (long)h->gdim[l][0] * h->gdim[l][1] * h->gdim[l][2];
    h->gch[l] = gch[l];
  }
  // critic grids (T:286-299)
  const int dch[5] = {h->Cin, 64, 128, 256, 256};
  h->ddim[0][0] = RDGAN_NHOURS; h->ddim[0][1] = nd; h->ddim[0][2] = nd;
  for (int l = 1; l <= 4; ++l)
    for (int a = 0; a < 3; ++a) {
      if (l == 1) { h->ddim[1][a] = (h->ddim[0][a] - 3) / 2 + 1; h->dpad[0][a] = 0; }
      else tf_same(h->ddim[l - 1][a], h->ddim[l][a], h->dpad[l - 1][a]);
    }
  for (int l = 0; l <= 4; ++l) { h->dL[l] = (long)h->ddim[l][0] * h->ddim[l][1] * h->ddim[l][2]; h->dch[l] = dch[l]; }
  h->F = (int)(h->dL[4] * 256);
  // parameter layouts (Keras weight order)
  {
    long gs[10] = {(long)h->n_in * h->n_nodes, h->n_nodes, 27L * 256 * 256, 256, 27L * 256 * 128, 128,
                   27L * 128 * 64, 64, 27L * 64, 1};
    long ds[10] = {27L * h->Cin * 64, 64, 27L * 64 * 128, 128, 27L * 128 * 256, 256, 27L * 256 * 256, 256, h->F, 1};
    long o = 0;
    for (int i = 0; i < 10; ++i) { h->goff[i] = o; h->gsz[i] = gs[i]; o += gs[i]; }
    h->n_gen = o; o = 0;
    for (int i = 0; i < 10; ++i) { h->doff[i] = o; h->dsz[i] = ds[i]; o += ds[i]; }
    h->n_critic = o;
  }
  h->KP0 = (h->n_in + 63) / 64 * 64;
  {
    // (the producer/consumer kernel wants N / 128 to be a power of two; n_nodes = 3 * 2^k * 256 for ndomain 8 / 16 / 32 / 64 / 128:
    // three launches of a third of the columns each, destination rows n_nodes apart)
    const int n3 = h->n_nodes / 3;
    h->dense16_ok = h->n_nodes % 3 == 0 && n3 % 128 == 0 && ((n3 / 128) & (n3 / 128 - 1)) == 0;
  }
}

// every plan of the handle (index = PL_*), `tab` = the row tables of all plans one behind the other, first[i] = plan i's first
// entry.  RdPlan::tab is left null (the caller points it into its device copy).  false: a tap offset outside the table's range.
static bool rd_build_plans(const RdGeom* h, std::vector<RdPlan>& plans, std::vector<RdRow>& tab, std::vector<size_t>& first) {
  const int nd = h->nd;
  const int* gch = h->gch; const int* dch = h->dch;
  plans.assign(PL_COUNT, RdPlan());
  for (auto& p : plans) memset(&p, 0, sizeof(p));
  plans[PL_GDENSE] = plan_rows(1, 1, 1, h->n_in, h->n_in, h->n_nodes, h->n_nodes);
  plans[PL_GDENSE16] = plan_rows(1, 1, 1, h->KP0, h->KP0, h->n_nodes / 3, h->n_nodes);
  for (int l = 1; l <= 3; ++l) {
    const int* sd = h->gdim[l - 1]; const int* od = h->gdim[l];
    plans[PL_G1F + l - 1] = plan_conv_fwd(sd[0], sd[1], sd[2], gch[l - 1], gch[l], od[0], od[1], od[2], 1, 1, 1, 1, 1);
    plans[PL_G1B + l - 1] = plan_conv_dgrad_s1(od[0], od[1], od[2], gch[l - 1], gch[l]);
    plans[PL_G1FC + l - 1] = plan_upconv_fwd_collapsed(sd[0], sd[1], sd[2], gch[l - 1], gch[l]);
    plans[PL_G1BC + l - 1] = plan_upconv_dgrad_collapsed(sd[0], sd[1], sd[2], gch[l - 1], gch[l]);
    plans[PL_G1FCX + l - 1] = plan_boxes(plans[PL_G1FC + l - 1]);
    plans[PL_G1BCX + l - 1] = plan_boxes(plans[PL_G1BC + l - 1]);
    for (int g = 0; g < 3; ++g) {
      plans[PL_F1WA + 3 * g + l - 1] = plan_fastd_wgrad(sd[0], sd[1], sd[2], gch[l - 1], gch[l], g);
      plans[PL_F1WAX + 3 * g + l - 1] = plan_boxes(plans[PL_F1WA + 3 * g + l - 1]);
    }
    plans[PL_F1BS + l - 1] = plan_fastd_dgrad_s(sd[0], sd[1], sd[2], gch[l - 1], gch[l]);
    plans[PL_F1BE + l - 1] = plan_fastd_dgrad_e(sd[0], sd[1], sd[2], gch[l - 1], gch[l]);
    plans[PL_F1FE + l - 1] = plan_fastd_fwd_e(sd[0], sd[1], sd[2], gch[l - 1], gch[l]);
  }
  {
    const int* g3 = h->gdim[3];
    plans[PL_G9F] = plan_rows(g3[0], g3[1], g3[2], 64, 64, 32, 32);
    plans[PL_G9B] = plan_rows(g3[0], g3[1], g3[2], 27, 32, 64, 64);
  }
  plans[PL_D1F] = plan_d1_fwd(nd, h->ddim[1][0], h->ddim[1][1], h->ddim[1][2], h->CP);
  for (int l = 2; l <= 4; ++l) {
    const int* id = h->ddim[l - 1]; const int* od = h->ddim[l]; const int* pd = h->dpad[l - 1];
    plans[PL_D2F + l - 2] = plan_conv_fwd(id[0], id[1], id[2], dch[l - 1], dch[l], od[0], od[1], od[2], 2, pd[0], pd[1], pd[2], 0);
    plans[PL_D2B + l - 2] = plan_conv_dgrad_s2(id[0], id[1], id[2], dch[l - 1], od[0], od[1], od[2], dch[l], pd);
    plans[PL_D2FX + l - 2] = plan_conv_fwd_boxes(id[0], id[1], id[2], dch[l - 1], dch[l], od[0], od[1], od[2], 2, pd[0], pd[1], pd[2]);
    plans[PL_D2BX + l - 2] = plan_boxes(plans[PL_D2B + l - 2]);
  }
  plans[PL_D1B] = plan_rows(h->ddim[1][0], h->ddim[1][1], h->ddim[1][2], 64, 64, h->ldp1, h->ldp1);
  tab.clear();
  first.assign(PL_COUNT, 0);
  for (int i = 0; i < PL_COUNT; ++i) {
    first[i] = tab.size();
    if (!plan_build_tables(plans[i], tab)) return false;
  }
  return true;
}

// the plans whose weight gradient runs through launch_wgrad / launch_wgrad16 (the streaming kernels with partial slabs), by the
// batch they run over: generator plans over at most MB samples, critic plans over the 3B batch [real; fake; x_hat]
static const int RD_WGRAD_GEN_PLANS[] = {PL_GDENSE, PL_G1F, PL_G2F, PL_G3F, PL_G9B, PL_G1FC, PL_G2FC, PL_G3FC, PL_G1FCX, PL_G2FCX, PL_G3FCX,
                                         PL_F1WA, PL_F2WA, PL_F3WA, PL_F1WS, PL_F2WS, PL_F3WS, PL_F1WD, PL_F2WD, PL_F3WD,
                                         PL_F1WAX, PL_F2WAX, PL_F3WAX, PL_F1WSX, PL_F2WSX, PL_F3WSX, PL_F1WDX, PL_F2WDX, PL_F3WDX};
static const int RD_WGRAD_CRITIC_PLANS[] = {PL_D1F, PL_D2F, PL_D3F, PL_D4F, PL_D2FX, PL_D3FX, PL_D4FX};

// floats of partial-slab workspace that cover every streaming weight-gradient launch of every batch size up to max_batch
static size_t rd_wgrad_workspace_floats(const RdGeom* h, const std::vector<RdPlan>& plans) {
  size_t wneed = 0;
  for (int id : RD_WGRAD_GEN_PLANS) wneed = std::max(wneed, wgrad_partial_bound(plans[id], h->MB));
  for (int id : RD_WGRAD_CRITIC_PLANS) wneed = std::max(wneed, wgrad_partial_bound(plans[id], h->NB));
  return wneed;
}
