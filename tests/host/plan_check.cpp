// Host planner of librdgan_hip.so under -fsanitize=address,undefined (no GPU, no HIP): builds every gather plan of every supported
// configuration and checks the invariants the kernels rely on.  Test infrastructure; compiled and run by tests/test_host_plan.py.
//
//   1. tables: every (row, tap) the validity mask lets through reads inside its source sample and writes inside its destination
//      sample; a row's destination offset appears exactly once per plan (no output row is computed twice or dropped);
//   2. border-class boxes (plan_boxes): the box plan computes, for every destination row, exactly the parent plan's set of
//      (weight tap, source offset) products -- the products dropped are the zero rows and nothing else;
//   3. tap lookup tables (tapinv / phL / phT / wmask) agree with the phases;
//   4. weight-gradient tilings: the workgroup -> (phase, split, row tile, n tile, slab) decode of the box kernels covers every
//      slab exactly NT times, and for EVERY batch size up to max_batch the partial-slab need stays within the bound the
//      workspace is sized with (the need of a box plan is not monotone in B).
#include <stdio.h>

#include "../../pr_disagg_radar_gan_amd/csrc/rdgan_hostplan.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                                     \
  do {                                                                       \
    if (!(cond)) {                                                           \
      if (g_fail < 20) { fprintf(stderr, "FAIL %s:%d %s: ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } \
      ++g_fail;                                                              \
    }                                                                        \
  } while (0)

struct Products {                       // what a plan computes, as flat sorted lists (cheap under the sanitizers)
  std::vector<long> dest;               // destination offset of every row
  struct P { long dest; int wrow; long src; bool operator<(const P& o) const { return dest != o.dest ? dest < o.dest : (wrow != o.wrow ? wrow < o.wrow : src < o.src); }
             bool operator==(const P& o) const { return dest == o.dest && wrow == o.wrow && src == o.src; } };
  std::vector<P> prod;                  // (destination, weight row block, source offset) of every product the mask lets through
};

// what the kernels see: row table + per-tap scalars.  s_shift plans (direct 27-tap form with the upsample folded into the
// gather) address their source through 2-bit codes; only their row bases are range-checked.
static void plan_products(const RdPlan& p, const std::vector<RdRow>& tab, size_t first, const char* name, Products& out) {
  CHECK(p.nphases >= 1 && p.nphases <= RD_MAX_PHASES, "%s: nphases %d", name, p.nphases);
  size_t rows = 0;
  for (int pi = 0; pi < p.nphases; ++pi) {
    const RdPhase& q = p.ph[pi];
    CHECK(q.ntaps >= 1 && q.ntaps <= RD_MAX_TAPS, "%s phase %d: ntaps %d", name, pi, q.ntaps);
    CHECK(q.L == q.LD * q.LH * q.LW && q.L > 0, "%s phase %d: L", name, pi);
    CHECK(p.phL[pi] == q.L && p.phT[pi] == q.ntaps, "%s phase %d: phL / phT", name, pi);
    CHECK((size_t)q.tab == rows, "%s phase %d: table start %d != %zu", name, pi, q.tab, rows);
    for (int w = 0; w < 64; ++w) {
      int want = -1;
      for (int t = 0; t < q.ntaps; ++t) if (q.tap[t].w == w) want = t;
      CHECK(p.tapinv[pi][w] == want, "%s phase %d: tapinv[%d] = %d, want %d", name, pi, w, p.tapinv[pi][w], want);
    }
    for (int l = 0; l < q.L; ++l) {
      const RdRow& r = tab[first + rows + l];
      CHECK(r.z >= 0 && (long)r.z + std::min(p.N, p.d_cstride) <= p.dst_sample,
            "%s phase %d row %d: destination offset %d outside %ld", name, pi, l, r.z, p.dst_sample);
      // (a row's source BASE may lie outside the tensor -- the extra hour plane of the shared-centre input gradient -- as long as
      // every tap the mask lets through lands inside; the folded-upsample plans address relative to the base, which must be inside)
      if (p.s_shift) CHECK(r.x >= 0 && (long)r.x + p.SC <= p.src_sample, "%s phase %d row %d: source base %d outside %ld", name, pi, l, r.x, p.src_sample);
      out.dest.push_back(r.z);
      for (int t = 0; t < q.ntaps; ++t) {
        if ((r.y & q.tap[t].mask) != q.tap[t].mask) continue;          // the zero row of a tap that leaves the picture
        if (p.s_shift) { out.prod.push_back({r.z, q.w_off + q.tap[t].w * p.w_rows_per_tap, (long)t}); continue; }
        CHECK(q.tap[t].delta % 4 == 0, "%s: tap delta not a float offset", name);
        const long so = r.x + q.tap[t].delta / 4;
        CHECK(so >= 0 && so + p.SC <= p.src_sample, "%s phase %d row %d tap %d: source offset %ld outside %ld", name, pi, l, t, so, p.src_sample);
        out.prod.push_back({r.z, q.w_off + q.tap[t].w * p.w_rows_per_tap, so});
      }
    }
    rows += q.L;
  }
  std::sort(out.dest.begin(), out.dest.end());
  std::sort(out.prod.begin(), out.prod.end());
  for (size_t i = 1; i < out.dest.size(); ++i) CHECK(out.dest[i] != out.dest[i - 1], "%s: destination %ld written by two rows", name, out.dest[i]);
  for (size_t i = 1; i < out.prod.size(); ++i) CHECK(!(out.prod[i] == out.prod[i - 1]), "%s: destination %ld: a product listed twice", name, out.prod[i].dest);
}

static void check_boxes(const RdPlan& one, const RdPlan& box, const std::vector<RdRow>& tab, size_t f1, size_t fb, const char* name) {
  Products a, b;
  plan_products(one, tab, f1, name, a);
  if (!box.boxes) return;                                   // plan_boxes found nothing to drop: the plan is the parent
  plan_products(box, tab, fb, name, b);
  CHECK(a.dest == b.dest, "%s: boxes cover %zu destination rows, the parent %zu (or other rows)", name, b.dest.size(), a.dest.size());
  CHECK(a.prod == b.prod, "%s: %zu products in the boxes, %zu in the parent (or other products)", name, b.prod.size(), a.prod.size());
  unsigned long long wm = 0;
  for (int pi = 0; pi < one.nphases; ++pi) for (int t = 0; t < one.ph[pi].ntaps; ++t) if (one.ph[pi].tap[t].w < 64) wm |= 1ull << one.ph[pi].tap[t].w;
  CHECK(box.wmask == wm, "%s: wmask", name);
  double f1s = plan_flops(one, 1), fbs = plan_flops(box, 1);
  CHECK(fbs < f1s, "%s: boxes do not drop work (%g vs %g)", name, fbs, f1s);
}

// host copy of rd_wgrad_box_decode (rdgan_gemm.hip.h): every slab must be reached by exactly NT workgroups
static void check_wgrad1(const RdPlan& p, int B, size_t bound, const char* name, bool full_decode, bool wide16);
static void check_wgrad(const RdPlan& p, int B, size_t bound, const char* name, bool full_decode) {
  check_wgrad1(p, B, bound, name, full_decode, false);
  check_wgrad1(p, B, bound, name, full_decode, true);       // (the 256 x 128 tiling of the bf16 kernels where it applies)
}
static void check_wgrad1(const RdPlan& p, int B, size_t bound, const char* name, bool full_decode, bool wide16) {
  int BR, BN, nsplit;
  RdWgradTiling T = wgrad_tiling(p, B, BR, BN, nsplit, wide16);
  const size_t need = wgrad_partial_need(p, B, wide16);
  CHECK(need <= bound, "%s B=%d: partial slabs need %zu floats, workspace bound %zu", name, B, need, bound);
  CHECK(p.N % BN == 0 || p.N < BN, "%s: N %d vs BN %d", name, p.N, BN);
  if (!T.box) {
    CHECK(nsplit >= 1 && (long)nsplit * T.rows_per_split >= (long)B * p.ph[0].L, "%s B=%d: splits do not cover the rows", name, B);
    CHECK(T.rows_per_split % 32 == 0, "%s: rows_per_split %% 32", name);
    return;
  }
  long wgs = 0, slabs = 0;
  for (int i = 0; i < p.nphases; ++i) {
    const long nsp = ((long)B * p.phL[i] + (1L << T.rps_log2) - 1) >> T.rps_log2;
    const long rtp = rd_wgrad_phase_rt(T, p.phT[i]);
    wgs += rtp * T.NT * nsp; slabs += rtp * nsp;
  }
  CHECK(wgs == T.nsplit && slabs == T.RT, "%s B=%d: box tiling counts %ld/%d workgroups, %ld/%d slabs", name, B, wgs, T.nsplit, slabs, T.RT);
  CHECK((size_t)slabs * BR * p.N == need, "%s: need", name);
  if (!full_decode) return;
  std::vector<int> hits((size_t)slabs, 0);
  for (int wg0 = 0; wg0 < T.nsplit; ++wg0) {
    int wg = wg0, prefix = 0, bz = -1, by = 0, rt = 0, slab = -1;
    for (int q = 0; q < p.nphases; ++q) {
      const int rtp = rd_wgrad_phase_rt(T, p.phT[q]);
      const int nsp = (B * p.phL[q] + (1 << T.rps_log2) - 1) >> T.rps_log2;
      const int tiles = rtp * T.NT, cnt = tiles * nsp;
      if (wg < cnt) { bz = q; by = wg / tiles; const int bx = wg - by * tiles; rt = bx / T.NT; slab = prefix + by * rtp; break; }
      wg -= cnt; prefix += rtp * nsp;
    }
    CHECK(bz >= 0 && slab >= 0 && slab + rt < slabs, "%s B=%d wg %d: decode out of range", name, B, wg0);
    if (bz >= 0 && slab + rt < slabs) hits[slab + rt]++;
    (void)by;
  }
  for (long i = 0; i < slabs; ++i) CHECK(hits[i] == T.NT, "%s B=%d: slab %ld written by %d workgroups, want %d", name, B, i, hits[i], T.NT);
}

int main(int argc, char** argv) {
  const int nds[] = {8, 16, 24, 32, 64, 120};
  const int batches[] = {1, 9, 96, 256, 2048};
  static const int pairs[][2] = {{PL_D2F, PL_D2FX}, {PL_D3F, PL_D3FX}, {PL_D4F, PL_D4FX}, {PL_D2B, PL_D2BX}, {PL_D3B, PL_D3BX}, {PL_D4B, PL_D4BX},
                                 {PL_G1FC, PL_G1FCX}, {PL_G2FC, PL_G2FCX}, {PL_G3FC, PL_G3FCX}, {PL_G1BC, PL_G1BCX}, {PL_G2BC, PL_G2BCX}, {PL_G3BC, PL_G3BCX},
                                 {PL_F1WA, PL_F1WAX}, {PL_F2WA, PL_F2WAX}, {PL_F3WA, PL_F3WAX}, {PL_F1WS, PL_F1WSX}, {PL_F2WS, PL_F2WSX}, {PL_F3WS, PL_F3WSX},
                                 {PL_F1WD, PL_F1WDX}, {PL_F2WD, PL_F2WDX}, {PL_F3WD, PL_F3WDX}};
  long nplans = 0, ntilings = 0;
  for (int nd : nds)
    for (int nc = 1; nc <= 3; nc += (nd == 120 ? 2 : 1)) {      // (ndomain 120, the largest the library accepts: 1 and 3 condition channels)
      CHECK(rd_geometry_ok(nd, nc, 1), "geometry %d %d", nd, nc);
      std::vector<RdPlan> plans; std::vector<RdRow> tab; std::vector<size_t> first;      // (independent of the batch size)
      {
        RdGeom g1;
        rd_geometry(&g1, nd, nc, 1);
        CHECK(rd_build_plans(&g1, plans, tab, first), "nd %d nc %d: a tap offset outside the table's range", nd, nc);
      }
      for (int mb : batches) {
        RdGeom g;
        rd_geometry(&g, nd, nc, mb);
        char name[96];
        if (mb == batches[0]) {        // plans and tables do not depend on the batch size: checked once per (nd, nc)
          bool paired[PL_COUNT] = {false};
          for (auto& pr : pairs) {
            snprintf(name, sizeof name, "nd%d nc%d plan %d/%d", nd, nc, pr[0], pr[1]);
            check_boxes(plans[pr[0]], plans[pr[1]], tab, first[pr[0]], first[pr[1]], name);
            paired[pr[0]] = paired[pr[1]] = true;
          }
          for (int i = 0; i < PL_COUNT; ++i) {
            if (paired[i]) continue;
            snprintf(name, sizeof name, "nd%d nc%d plan %d", nd, nc, i);
            Products m;
            plan_products(plans[i], tab, first[i], name, m);
          }
          nplans += PL_COUNT;
        }
        const size_t cap = rd_wgrad_workspace_floats(&g, plans);
        auto sweep = [&](const int* ids, size_t n, int maxB) {
          for (size_t k = 0; k < n; ++k) {
            const RdPlan& p = plans[ids[k]];
            if (p.N % 64) continue;                                            // (PL_G9B etc.: not a streaming weight gradient)
            const size_t bound = wgrad_partial_bound(p, maxB);
            CHECK(bound <= cap, "cap");
            for (int B = 1; B <= maxB; B += (B < 300 ? 1 : 13)) {
              snprintf(name, sizeof name, "nd%d nc%d mb%d wgrad plan %d", nd, nc, mb, ids[k]);
              check_wgrad(p, B, bound, name, B <= 3 || B % 211 == 0);
              ++ntilings;
            }
            check_wgrad(p, maxB, bound, name, true);
          }
        };
        sweep(RD_WGRAD_GEN_PLANS, sizeof(RD_WGRAD_GEN_PLANS) / sizeof(int), g.MB);
        sweep(RD_WGRAD_CRITIC_PLANS, sizeof(RD_WGRAD_CRITIC_PLANS) / sizeof(int), g.NB);
      }
    }
  // rejected configurations stay rejected
  CHECK(!rd_geometry_ok(4, 1, 1) && !rd_geometry_ok(20, 1, 1) && !rd_geometry_ok(128, 1, 1) && !rd_geometry_ok(16, 4, 1) && !rd_geometry_ok(16, 1, 0), "geometry guard");
  printf("plan_check: %ld plans, %ld weight-gradient tilings, %d failures\n", nplans, ntilings, g_fail);
  (void)argc; (void)argv;
  return g_fail ? 1 : 0;
}
