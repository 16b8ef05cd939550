// Host side of librdgan_hip.so: plans, workspace, step orchestration and the C ABI of
// include/rdgan.h.  gfx950 only.  T = gan_train_cwgangp_pixelnorm.py,
// L = alternative_domains/gan_train_cwgangp_pixelnorm_largedomain.py in the reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <algorithm>
#include <unordered_set>

#include "../../include/rdgan.h"
#include "rdgan_plan.h"
#include "rdgan_hostplan.h"
#include "rdgan_rng.h"
#include "rdgan_gemm.hip.h"
#include "rdgan_gemm_ws.hip.h"
#include "rdgan_gemm_ws16.hip.h"
#include "rdgan_gemm_f16.hip.h"
#include "rdgan_upconv16.hip.h"
#include "rdgan_upconv16b.hip.h"
#include "rdgan_upconv16t.hip.h"
#include "rdgan_elem.hip.h"
#include "rdgan_data.hip.h"
#include "rdgan_edge.hip.h"
#include "rdgan_d2slab16.hip.h"
#include "rdgan_d2fwd16.hip.h"
#include "rdgan_upwgrad16.hip.h"
#include "rdgan_upwgrad16b.hip.h"
#include "rdgan_d2wgrad16.hip.h"
#include "rdgan_d3wgrad16.hip.h"
#include "rdgan_d1fwd16.hip.h"
#include "rdgan_g9bwd16.hip.h"
static_assert(RDGAN_LOSS_SLOTS == 8, "k_critic_losses / k_gen_loss write slots 0..7");
// k_g9_wgrad_mfma: W a power of two in [8, 128]; dynamic LDS = tile + staged dlogits rows + row descriptors (>= the 32 KB fold)
static bool g9w_mfma_ok(int nd, long npix) { return nd >= 8 && nd <= 128 && (nd & (nd - 1)) == 0 && npix < 0x7FFFFFFFL; }
static size_t g9w_mfma_lds(bool bf16) { return std::max<size_t>((size_t)128 * (bf16 ? 128 : 256) + (1440 + 144) * sizeof(float), 32768); }

#define RD_GP_WEIGHT 10.0f   // the literal at T:392



// one GEMM launch under rdgan_profile_launches: which plan, through which kernel, how many samples and algorithmic FLOPs
struct RdLaunchRec { int plan, kind, batch; double flops; char kernel[48]; hipEvent_t e0, e1; };

struct rdgan_handle : RdGeom {     // geometry + parameter layout: rdgan_hostplan.h
  std::string err;
  // plans
  std::vector<RdPlan> plans;
  RdPlan* d_plans = nullptr;
  RdRow* d_tab = nullptr;
  // workspace
  char* ws = nullptr;
  size_t ws_bytes = 0;
  float *xcat, *h0, *h1, *r1, *h2, *r2, *h3, *r3, *P9, *fake, *dl, *gh3, *gup3, *dy2, *gup2, *dy1, *gup1, *ga0;
  float *cin, *dh[5], *du[5], *v, *P1, *g0, *gpv;
  float* ubias_part = nullptr;    // k_upconv_wgrad_slab16's bias-gradient partials [32 groups][8 phases][64]
  float *wpartial, *cpartial, *kpartial;
  size_t wpartial_cap = 0, cpartial_cap = 0, kpartial_cap = 0;
  float *DWT[5], *W1T, *GWT[4], *W9T, *W1P, *dW1P;
  // shared-centre form: per block the hour differences E of its input and the 48 weight forms U (written by the forward,
  // reused by the backward); T / plane sums gS, the gradient wrt E, weight-form gradients and transposes (scratch)
  float *fE[4], *fU[4], *fgS, *fdE, *fdU, *fUT;
  // "bf16" storage mode: activations and activation gradients live in the workspace as bf16 (the same float* fields then
  // point at bf16 data; act_off() does their pointer arithmetic); bf16 weight images of every bf16-MFMA GEMM, rebuilt from
  // the fp32 master weights inside each call: shared-centre forms [48][N][K] and their tap-reordered input-gradient
  // stack, critic layers 2-4 forward [27][Cout][Cin] / input gradient [27][Cin][Cout], generator block 1 collapsed forms
  // [64][Cout][Cin] and (re-ordered by tap) [64][Cin][Cout], the first critic kernel [ldp1][64]
  void *bU[4], *bUT;
  void *bWF[5], *bWB[5];
  void *bG1F[4], *bG1B, *bW1B;
  void *fWF[5], *fWB[5], *fG1F[4], *fG1B;    // the same images in fragment order (rdgan_gemm_f16.hip.h: rd_wfrag_index), written beside them
  int wgrad_wide = 0;             // 1: bf16 weight gradients of N % 128 == 0 layers with >= 32768 rows on 256 x 128 tiles, three stages (k_wgrad_gemm_ws16<256,128>): measured SLOWER, default off
  int conv_f16 = 1;               // 1: bf16 storage mode: the large gather GEMMs by k_conv_gemm_f16 (weights global -> VGPR, 256 x 128 tiles)
  void* bW3I;                     // weight image of the slab kernel of generator block 3 (rdgan_upconv16.hip.h): 1 MB, MFMA-fragment order
  int upconv_slab = 1;            // 1: bf16 storage mode, ndomain 16: block 3 forward (collapsed form) by the slab kernel k_upconv_slab16
  void* bW3T = nullptr;           // weight image of the TILED slab kernel of generator block 3 (rdgan_upconv16t.hip.h): 1 MB, [phase][half][tap][j]
  int upconv_slab_t = 1;          // 1: bf16 storage mode, source planes larger than 8 x 8 (ndomain 32, 64, ...: multiples of 16): block 3 forward by k_upconv_slab_t16
  void* bW2I = nullptr;           // weight image of the slab kernel of generator block 2 (rdgan_upconv16b.hip.h): 4 MB
  int upconv2_slab = 1;           // 1: the same for block 2 (k_upconv2_slab16)
  int g9_fused = 1;               // 1: with the block-3 slab kernel, the last conv's tap products come out of that kernel's epilogue
                                  // (no pass over h3; critic steps do not store h3 at all)
  void* bW9I;                     // A-fragment image of the 64 -> 1 kernel for that epilogue (k_g9_wimg): 4 KB
  void* bW2S;                     // weight image of the slab kernel of critic layer 2's input gradient (rdgan_d2slab16.hip.h): 432 KB
  int d3_wgrad_slab = 1;          // 1: bf16 storage mode, ndomain 16: weight gradient of critic layer 3 by k_d3_wgrad_slab16
  int d2_wgrad_slab = 1;          // 1: bf16 storage mode, ndomain 16: weight gradient of critic layer 2 by k_d2_wgrad_slab16
  int upwgrad_slab = 1;           // 1: bf16 storage mode, ndomain 16, collapsed form: weight gradient of generator block 3 by k_upconv_wgrad_slab16
  int d1_dgrad_fused = 1;         // 1: bf16 storage mode, ndomain 16: dD/d(sample) of layer 1 in one pass per sample (k_d1_dgrad_sample16)
  int d1_wgrad16 = 1;             // 1: bf16 storage mode: layer-1 weight gradient + bias gradient on the bf16 matrix pipe (k_d1_wgrad16)
  void* bW2F = nullptr;           // weight image of the slab kernel of critic layer 2's forward (rdgan_d2fwd16.hip.h): 448 KB
  int d2_fwd_slab = 0;            // 1: bf16 storage mode, ndomain 16: forward of critic layer 2 by k_d2_fwd_slab16 (measured: no faster than the streaming GEMM, default off)
  unsigned char* g1bits = nullptr; // layer 1's gate in 2 bits per element (written by k_d1_gemm_fwd, read by k_d2_dgrad_slab16): 16 B per row
  int dense16 = 1;                // 1: bf16 storage mode: the generator's Dense layer on the bf16 matrix pipe (inputs and kernel rounded to bf16, K padded to 64)
  int dense_skinny = 1;           // 1: ... and, on a handle of max_batch <= 128, by k_dense16_skinny (weights and inputs streamed in fragment order, no LDS)
  void* xcat16 = nullptr;         // [MB][KP0] bf16
  void* bW0 = nullptr;            // [n_nodes][KP0] bf16
  int g9_bwd_mfma = 1;            // 1: bf16 storage mode: input gradient of the last conv + block 3's PixelNorm backward on the fp32 matrix pipe (k_g9_bwd_mfma16)
  int d1_fwd_sample = 1;          // 1: bf16 storage mode, ndomain 16: layer-1 forward / second sweep with a sample resident in LDS (k_d1_fwd_sample16)
  int wgrad_boxes = 1;            // 1: the weight gradients of critic layers 2-4 (streaming kernels) on the border-class boxes too
  int border_boxes = 1;           // 1: forward / second-sweep GEMMs of critic layers 2-4 skip the taps that leave the picture (plan_conv_fwd_boxes)
  int d2_gate_bits = 1;           // 1: the slab kernel of layer 2's input gradient reads the packed gate instead of layer 1's output
  int d2_slab = 1;                // 1: bf16 storage mode, ndomain 16: input gradient of critic layer 2 by k_d2_dgrad_slab16
  int a16 = 0;                    // 1: bf16 storage mode (option "bf16"; needs the collapsed + shared-centre forms)
  int g9_direct = 1;              // 1: backward of the 64 -> 1 conv straight from the dlogits (no im2col matrix), fused with block 3's PixelNorm backward
  int fast_fwd = -1;              // 1: forward of generator blocks 2, 3 as shared part T = S x + difference part (48 instead of 64 tap products); -1: by storage mode
  int fast_bwd = -1;              // 1: generator blocks' weight/input gradients in the shared-centre form along d (48 instead of 64 tap products); -1: by storage mode
  float *GWC[4], *GWD[4], *dWc;   // collapsed generator weights, their dgrad form, collapsed wgrad scratch
  int collapse = 1;               // 1: 8-tap collapsed generator blocks (default); 0: direct 27-tap form
  int tapgather = 1;              // 1: last generator conv sums its in-tile taps in the GEMM epilogue; 0: full column matrix + gather kernel
  int ws_ksplit = 1;              // 1: split K of mid-size producer/consumer launches to fill whole rounds of workgroups; 0: off; >1: force (tests)
  int wave_spec = 1;              // 1: producer/consumer (wave-specialised, LDS-DMA) kernel for the big clean GEMMs
  int resident = 1;               // 1: bf16 forward GEMMs of the shared-centre form keep the tile's source rows resident in LDS (k_conv_gemm_ws<..., RES>)
  int edge_kernels = 1;           // 1: dedicated streaming kernels for the generator's last conv (rdgan_edge.hip.h); 0: the tiled GEMM kernels
  int sample_offset = 0;          // global index of this rank's first sample: RandomWeightedAverage's alpha of sample k is uniform(key, sample_offset + k)
  float* g9b_tmp;                 // 64 partial sums of the last conv's bias gradient
  float* gp_part;                 // [max_batch][64] partial sums of squares of the penalty's per-sample gradient norm
  float* dw6_part;                // [16][F] row-slice partial sums of the critic Dense weight gradient
  // Side stream (option "side_stream", default on): weight-only kernels (generator weight forms, critic weight transposes /
  // bf16 images) and the bias-gradient column sums run beside the caller's stream, ordered by events: ~50 launches of 5-30 us
  // per iteration that would otherwise sit between the GEMMs.  Same kernels, same arithmetic: results are bit-identical.
  int side_on = 1;
  int split3 = 0;                 // 1: fp32 conv GEMMs of the producer/consumer kernel multiply on the bf16 matrix pipe from 3-way split operands (optional data point)
  int dense_slices = 0;           // tests: force the row-slice count of the critic Dense weight gradient (0 = by batch size)
  // test hook "keep_gates": the critic step's second sweep overwrites the x_hat third of h_l in place; with the option on, that
  // third is copied here first, so that rdgan_debug_activation can return the activations (= LeakyReLU / dropout pattern) of
  // all 3B samples of the last critic step.  Allocated when the option is set, never inside a step.
  // Weight-form cache (rdgan_set_weight_versions): the forward forms of the generator (W9T, collapsed / shared-centre forms, their
  // bf16 images) and the critic's forms (transposes, padded / bf16 images) are functions of the weights alone.  The caller may
  // assert a content version for each slab; a call whose (pointer, version, form options) equal those the forms in the
  // workspace were built from skips the weight-only kernels.  Version 0 = unknown: rebuild (the default).
  uint64_t gver_in = 0, cver_in = 0;
  const float* gcache_ptr = nullptr; uint64_t gcache_ver = 0; int gcache_cfg = -1;
  const float* ccache_ptr = nullptr; uint64_t ccache_ver = 0; int ccache_cfg = -1;
  long form_builds[2] = {0, 0};   // how many times the generator / critic forms were (re)built (tests)
  int keep_gates = 0;
  void* gate_keep[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int gate_keep_B = 0;            // samples held (0: the last call was not a critic step)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_cw = nullptr, ev_g[4] = {nullptr, nullptr, nullptr, nullptr};
  int* d_flag;
  // kernels whose dynamic-LDS limit has been raised on THIS handle's device (hipFuncSetAttribute is per device; keeping the
  // record per handle rather than per process makes two handles on two devices, or on two threads, independent)
  std::unordered_set<const void*> lds_attr_done;
  // per-launch profiling (rdgan_profile_launches / rdgan_launch_table): HIP events around every GEMM launch
  int prof_launches = 0;
  std::vector<RdLaunchRec> launch_recs;
  size_t launch_used = 0;
  char cur_kernel[48] = {0};      // tile name of the launch being issued (set by the launchers while prof_launches is on)
  // profiling
  double flops_acc = 0;           // algorithmic FLOPs (2 * rows * taps * K * N of the forms actually run) of every GEMM launched so far
  unsigned prof_mask = 0;
  std::vector<hipEvent_t> ev_start[RDGAN_NUM_TAGS], ev_stop[RDGAN_NUM_TAGS];
  size_t ev_used[RDGAN_NUM_TAGS] = {0};
};

#define RD_CHECK(h, call)                                                        \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                      \
      char b_[512];                                                              \
      snprintf(b_, sizeof b_, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      if (h) (h)->err = b_;                                                      \
      return (int)e_;                                                            \
    }                                                                            \
  } while (0)
#define RD_TRY(expr)            \
  do {                          \
    int r_ = (expr);            \
    if (r_ != 0) return r_;     \
  } while (0)

static int bad_arg(rdgan_handle* h, const char* msg) {
  if (h) h->err = msg;
  return -2;
}

// side stream ordered behind everything issued on `st` so far (or `st` itself with the option off)
static hipStream_t side_fork(rdgan_handle* h, hipStream_t st) {
  if (!h || !h->side_on || !h->side) return st;
  if (hipEventRecord(h->ev_fork, st) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_fork, 0) != hipSuccess) return st;
  return h->side;
}
// `st` waits for everything issued on the side stream so far
static int side_join(rdgan_handle* h, hipStream_t st, hipEvent_t ev);

// raise a kernel's dynamic shared-memory limit once per handle (always, for the handle-less op-level entry points)
static int ensure_lds(rdgan_handle* h, const void* kern, size_t lds) {
  if (h && h->lds_attr_done.count(kern)) return 0;
  RD_CHECK(h, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (h) h->lds_attr_done.insert(kern);
  return 0;
}

static int side_join(rdgan_handle* h, hipStream_t st, hipEvent_t ev) {
  if (!h || !h->side_on || !h->side || st == h->side) return 0;
  RD_CHECK(h, hipEventRecord(ev, h->side));
  RD_CHECK(h, hipStreamWaitEvent(st, ev, 0));
  return 0;
}

struct ProfScope {
  rdgan_handle* h; int tag; hipStream_t st; bool on;
  ProfScope(rdgan_handle* h_, int tag_, hipStream_t st_) : h(h_), tag(tag_), st(st_), on(false) {
    if (h && tag >= 0 && (h->prof_mask >> tag & 1u) && h->ev_used[tag] < h->ev_start[tag].size()) {
      on = true;
      (void)hipEventRecord(h->ev_start[tag][h->ev_used[tag]], st);
    }
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(h->ev_stop[tag][h->ev_used[tag]], st); h->ev_used[tag]++; }
  }
};

// plan index of a plan that lives in the handle's table (-1: a temporary plan of the op-level entry points)
static int plan_index(const rdgan_handle* h, const RdPlan& hp) {
  if (!h || h->plans.empty()) return -1;
  const RdPlan* b = h->plans.data();
  return (&hp >= b && &hp < b + h->plans.size()) ? (int)(&hp - b) : -1;
}
// kinds of a recorded launch
enum { RD_KIND_CONV = 0, RD_KIND_WGRAD = 1, RD_KIND_EDGE = 2 };
struct LaunchScope {
  rdgan_handle* h; hipStream_t st; RdLaunchRec* r;
  LaunchScope(rdgan_handle* h_, int plan, int kind, int batch, double flops, hipStream_t st_) : h(h_), st(st_), r(nullptr) {
    if (h && h->prof_launches && h->launch_used < h->launch_recs.size()) {
      r = &h->launch_recs[h->launch_used++];
      r->plan = plan; r->kind = kind; r->batch = batch; r->flops = flops; r->kernel[0] = 0;
      h->cur_kernel[0] = 0;
      (void)hipEventRecord(r->e0, st);
    }
  }
  ~LaunchScope() {
    if (r) { (void)hipEventRecord(r->e1, st); memcpy(r->kernel, h->cur_kernel, sizeof(r->kernel)); }
  }
};
#define RD_KNAME(h, ...) do { if ((h) && (h)->prof_launches) snprintf((h)->cur_kernel, sizeof((h)->cur_kernel), __VA_ARGS__); } while (0)

// ------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------
// algorithmic FLOPs of one pass over a plan: 2 * sum over phases of (rows * taps) * K per tap * N

template <int BM, int BN, int WM, int WN, int BK, bool PARTIAL, bool SHIFT, bool SRC16 = false, bool OUT16 = false>
static int launch_conv_cfg(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const float* src,
                           const float* W, int ldw, float* dst, const RdEpi& epi, hipStream_t st) {
  constexpr int AST = BK == 32 ? BK : BK + 4, BST = BN;
#ifdef RD_ONE_BLOCK_PER_CU
  constexpr size_t lds = 150 * 1024;      // diagnostic build: one workgroup per CU (no partner wave on a SIMD)
#else
  // the BK = 32 kernels reuse the staging LDS for the output tile + BM row bases in the epilogue
  constexpr size_t lds_loop = 2 * (size_t)(BM * AST + BK * BST) * sizeof(float);
  constexpr size_t lds_epi = BK == 32 ? ((size_t)BM * BN * sizeof(float) + (size_t)BM * 16) : 0;
  constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
#endif
  auto kern = k_conv_gemm<BM, BN, WM, WN, BK, PARTIAL, SHIFT, SRC16, OUT16>;
  RD_KNAME(h, "k_conv_gemm<%d,%d,BK%d>", BM, BN, BK);
  RD_TRY(ensure_lds(h, (const void*)kern, lds));
  long tm = plan_tiles(hp, B, BM);
  if (tm <= 0) return 0;
  {  // a tile's buffer descriptor is based at its first sample: its span must stay below 2 GiB
    long minL = hp.ph[0].L;
    for (int i = 1; i < hp.nphases; ++i) minL = std::min<long>(minL, hp.ph[i].L);
    if (std::min<long>(B, BM / minL + 2) * hp.src_sample * 4 >= 0x7FFFFFF0L) return bad_arg(h, "conv: source tile span exceeds 2 GiB");
    if (std::min<long>(B, BM / minL + 2) * hp.dst_sample * 4 >= 0x7FFFFFF0L) return bad_arg(h, "conv: destination tile span exceeds 2 GiB");
  }
  const long blocks = tm * (hp.N / BN);
  // split-K for small-M / large-K layers (critic tail, Dense, first generator block): few workgroups, long K loops
  RdEpi e2 = epi;
  e2.ksplit = 1; e2.kpart = nullptr; e2.kstride = 0;
  const long nch = (long)hp.ph[0].ntaps * ((hp.SC + BK - 1) / BK);
  const long total = (long)B * hp.dst_sample;
  if (epi.addt && BK != 32) return bad_arg(h, "conv: the shared-centre epilogue needs the BK = 32 kernels");
  if (h && !OUT16 && blocks < 256 && nch >= 16 && hp.d_cstride == hp.N && hp.N % 4 == 0 && epi.mode != RD_EPI_BIAS_PN_LRELU && !epi.addt) {
    long ks = std::min<long>(std::min<long>(8, 640 / blocks), nch / 4);
    for (int i = 1; i < hp.nphases; ++i) ks = std::min<long>(ks, (long)hp.ph[i].ntaps * ((hp.SC + BK - 1) / BK) / 2);
    if (ks >= 2 && (size_t)(ks * total) <= h->kpartial_cap) { e2.ksplit = (int)ks; e2.kpart = h->kpartial; e2.kstride = total; }
  }
  dim3 grid((unsigned)blocks, (unsigned)e2.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, dp, B, src, W, ldw, dst, e2);
  if (e2.ksplit > 1)
    hipLaunchKernelGGL(k_splitk_finish<false>, dim3((unsigned)std::min<long>((total / 4 + 255) / 256, 2048)), dim3(256), 0, st, dst,
                       total, hp.N, e2);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

template <int BM, int BN, int WM, int WN, int TG, bool BF>
static int launch_conv_ws_tg(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const float* src,
                             const float* W, int ldw, float* dst, const RdEpi& epi, hipStream_t st, bool res = false);
// Resident-tile mode of the bf16 kernel (k_conv_gemm_ws<..., RES>): every phase's taps are (h,w)-shifted views of the same
// source planes (same d offset, offsets in [-1,1], unit stride, loop grid = source (h,w) grid), whole planes per tile, and
// the tile's rows for all channel chunks plus two weight stages fit the 80 KiB a workgroup may use at two per CU
static bool conv16_resident_ok(const RdPlan& hp, int BM, int BN) {
  if (hp.s_shift || hp.SC % 64) return false;
  const int plane = hp.SH * hp.SW;
  if (plane < 1 || plane > BM || BM % plane) return false;
  if ((size_t)(hp.SC / 64) * BM * 128 + 2 * (size_t)BN * 128 > 80 * 1024) return false;
  for (int i = 0; i < hp.nphases; ++i) {
    const RdPhase& q = hp.ph[i];
    if (q.ntaps < 1 || q.ntaps > 4 || q.LH != hp.SH || q.LW != hp.SW || q.L % plane) return false;
    for (int a = 0; a < 3; ++a) if (q.s_mul[a] != 1) return false;
    for (int t = 0; t < q.ntaps; ++t) {
      if (q.tap_off[t][0] != q.tap_off[0][0]) return false;
      if (q.tap_off[t][1] < -1 || q.tap_off[t][1] > 1 || q.tap_off[t][2] < -1 || q.tap_off[t][2] > 1) return false;
    }
  }
  return true;
}
// TG = taps whose gather offsets a loader wave keeps in registers: 4 when no phase has more (shared-centre plans).
// BF = bf16 operands (src / W point at bf16 data, W stored [tap block][N][K]); see k_conv_gemm_ws.
template <int BM, int BN, int WM, int WN, bool BF = false>
static int launch_conv_ws_cfg(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const float* src,
                              const float* W, int ldw, float* dst, const RdEpi& epi, hipStream_t st, bool res = false) {
  int maxtaps = 0;
  for (int i = 0; i < hp.nphases; ++i) maxtaps = std::max(maxtaps, hp.ph[i].ntaps);
  if (maxtaps <= 4) return launch_conv_ws_tg<BM, BN, WM, WN, 4, BF>(h, hp, dp, B, src, W, ldw, dst, epi, st, res);
  return launch_conv_ws_tg<BM, BN, WM, WN, 8, BF>(h, hp, dp, B, src, W, ldw, dst, epi, st);
}
template <int BM, int BN, int WM, int WN, int TG, bool BF>
static int launch_conv_ws_tg(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const float* src,
                             const float* W, int ldw, float* dst, const RdEpi& epi, hipStream_t st, bool res) {
  constexpr size_t lds_loop = 2 * (size_t)(BM * 32 + 32 * BN) * sizeof(float);
  constexpr size_t lds_epi = (size_t)BM * BN * sizeof(float) + (size_t)BM * 16;
  constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  auto kern = k_conv_gemm_ws<BM, BN, WM, WN, TG, BF>;
  if constexpr (BM == 256 && BN == 64 && (TG == 4 || BF)) {     // the dominant launch runs under its own symbol (same code)
    if (epi.nametag == 1) kern = k_conv_gemm_ws<BM, BN, WM, WN, TG, BF, 1>;
    if constexpr (BF && TG == 4) {
      if (res) kern = epi.nametag == 1 ? k_conv_gemm_ws<BM, BN, WM, WN, TG, true, 1, true> : k_conv_gemm_ws<BM, BN, WM, WN, TG, true, 0, true>;
    } else res = false;
  } else res = false;
  if constexpr (!BF) {
    // "split3": the same launch on the bf16 matrix pipe from operands split three ways in registers (rdgan_gemm_ws.hip.h)
    if (h && h->split3) kern = k_conv_gemm_ws<BM, BN, WM, WN, TG, false, 0, false, true>;
  }
  RD_KNAME(h, "k_conv_gemm_ws<%d,%d,TG%d%s%s>", BM, BN, TG, BF ? ",bf16" : "", res ? ",res" : "");
  RD_TRY(ensure_lds(h, (const void*)kern, lds));
  constexpr int KCH = BF ? 64 : 32;                    // K elements per chunk
  if (BF && hp.SC % 64) return bad_arg(h, "conv: bf16 operands need a multiple of 64 channels per tap");
  // (the kernel decodes its tile with shifts: round 3's first bf16 Dense launch, 24 column tiles, computed 8 of them -- silently)
  if (hp.N % BN || ((hp.N / BN) & (hp.N / BN - 1))) return bad_arg(h, "conv: the producer/consumer kernel needs N / BN to be a power of two");
  long tm = plan_tiles(hp, B, BM);
  if (tm <= 0) return 0;
  long minL = hp.ph[0].L;
  for (int i = 1; i < hp.nphases; ++i) minL = std::min<long>(minL, hp.ph[i].L);
  if (std::min<long>(B, BM / minL + 2) * hp.src_sample * 4 >= 0x7FFFFFF0L) return bad_arg(h, "conv: source tile span exceeds 2 GiB");
  if (std::min<long>(B, BM / minL + 2) * hp.dst_sample * 4 >= 0x7FFFFFF0L) return bad_arg(h, "conv: destination tile span exceeds 2 GiB");
  RdEpi e2 = epi;
  e2.ksplit = 1; e2.kpart = nullptr; e2.kstride = 0;
  const long blocks = tm * (hp.N / BN);
  // Wave quantisation: a mid-size launch of long workgroups (say 384 on 256 CUs) leaves a third of the chip idle in
  // its last round.  Split K so that the workgroup count fills whole rounds; the partial sums cost one extra pass over
  // the (small) output, priced at 3 % per split.
  const long total = (long)B * hp.dst_sample;
  if (h && h->ws_ksplit && !res && blocks < 1024 && hp.d_cstride == hp.N && epi.mode != RD_EPI_BIAS_PN_LRELU && !epi.addt) {
    long nch = (long)hp.ph[0].ntaps * (hp.SC / KCH);
    for (int i = 1; i < hp.nphases; ++i) nch = std::min<long>(nch, (long)hp.ph[i].ntaps * (hp.SC / KCH));
    auto cost = [&](long ks) { double r = (double)(blocks * ks) / 256.0; return std::ceil(r) / r * (1.0 + 0.03 * (ks - 1)); };
    long best = 1;
    for (long ks = 2; ks <= 8; ++ks)
      if (nch / ks >= 12 && (size_t)(ks * total) <= h->kpartial_cap && cost(ks) < cost(best) - 0.02) best = ks;
    if (h->ws_ksplit > 1) best = std::min<long>(h->ws_ksplit, std::max<long>(1, nch / 4));   // tests: force a split
    if (best > 1 && (size_t)(best * total) <= h->kpartial_cap) { e2.ksplit = (int)best; e2.kpart = h->kpartial; e2.kstride = total; }
  }
  if (!BF) e2.out16 = 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)e2.ksplit), dim3(512), lds, st, dp, B, src, W, ldw, dst, e2);
  if (e2.ksplit > 1) {
    const dim3 fg((unsigned)std::min<long>((total / 4 + 255) / 256, 2048));
    if (e2.out16) hipLaunchKernelGGL(k_splitk_finish<true>, fg, dim3(256), 0, st, dst, total, hp.N, e2);
    else hipLaunchKernelGGL(k_splitk_finish<false>, fg, dim3(256), 0, st, dst, total, hp.N, e2);
  }
  RD_CHECK(h, hipGetLastError());
  return 0;
}

static int launch_conv(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const float* src, const float* W,
                       int ldw, float* dst, const RdEpi& epi, hipStream_t st, int tag) {
  ProfScope ps(h, tag, st);
  LaunchScope ls(h, plan_index(h, hp), RD_KIND_CONV, B, plan_flops(hp, B), st);
  if (h) h->flops_acc += plan_flops(hp, B);
  const bool partial = (hp.SC & 3) != 0, shift = hp.s_shift != 0;
  if (partial && shift) return bad_arg(h, "conv: SC % 4 != 0 with a folded upsample is not supported");
  if (hp.SC < 32 && hp.SC != 27) {   // small-K taps (D1): BK = 8
    if (hp.N % 64 || shift) return bad_arg(h, "conv: unsupported BK=8 case");
    if (partial) return launch_conv_cfg<128, 64, 2, 2, 8, true, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
    return launch_conv_cfg<128, 64, 2, 2, 8, false, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
  }
#define RD_CONV(BM_, BN_, WM_, WN_)                                                                                  \
  do {                                                                                                               \
    if (partial) return launch_conv_cfg<BM_, BN_, WM_, WN_, 32, true, false>(h, hp, dp, B, src, W, ldw, dst, epi, st); \
    if (shift) return launch_conv_cfg<BM_, BN_, WM_, WN_, 32, false, true>(h, hp, dp, B, src, W, ldw, dst, epi, st);   \
    return launch_conv_cfg<BM_, BN_, WM_, WN_, 32, false, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);             \
  } while (0)
  const bool ws_ok = h && h->wave_spec && !partial && !shift && hp.SC % 32 == 0;
  if (ws_ok && h->wave_spec == 2) {      // test mode: the producer/consumer kernel regardless of the problem size
    if (hp.N % 128 == 0) return launch_conv_ws_cfg<128, 128, 2, 2>(h, hp, dp, B, src, W, ldw, dst, epi, st);
    if (hp.N == 64) return launch_conv_ws_cfg<256, 64, 4, 1>(h, hp, dp, B, src, W, ldw, dst, epi, st);
  }
  // at most 64 rows (the generator's Dense layer at ndomain 64 with 64 samples: 64 x 4196 against 4196 x 49152 weights): a
  // 128-row tile would multiply 64 rows of zeros -- the launch streams its 825 MB of weights at 1.5 TB/s, the 64-row tile at 2.6
  if (hp.nphases == 1 && (long)B * hp.ph[0].L <= 64 && hp.N % 64 == 0 && !shift) RD_CONV(64, 64, 2, 2);
  // phases of unequal length (stride-2 input gradients: 8 ... 1 taps): the 8-tap workgroups set the launch time, so a
  // mid-size launch takes the narrower tile (twice the workgroups, half the work each)
  int tmin = hp.ph[0].ntaps, tmax = tmin;
  for (int i = 1; i < hp.nphases; ++i) { tmin = std::min(tmin, hp.ph[i].ntaps); tmax = std::max(tmax, hp.ph[i].ntaps); }
  const bool uneven = tmax >= 2 * tmin;
  if (hp.N % 128 == 0 && plan_tiles(hp, B, 128) * (hp.N / 128) >= 200 &&
      !(uneven && ws_ok && plan_tiles(hp, B, 128) * (hp.N / 128) < 1024)) {
    if (ws_ok) return launch_conv_ws_cfg<128, 128, 2, 2>(h, hp, dp, B, src, W, ldw, dst, epi, st);
    RD_CONV(128, 128, 2, 2);
  }
  if (hp.N % 64 == 0) {
    // N = 64 layers with plenty of rows: 256-row tile so every wave owns a 64x64 tile (64 MFMAs per barrier)
    if (hp.N == 64 && !shift && plan_tiles(hp, B, 256) >= 1024) {   // (the SHIFT variant of this tile would spill)
      if (ws_ok) return launch_conv_ws_cfg<256, 64, 4, 1>(h, hp, dp, B, src, W, ldw, dst, epi, st);
      if (partial) return launch_conv_cfg<256, 64, 4, 1, 32, true, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
      return launch_conv_cfg<256, 64, 4, 1, 32, false, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
    }
    if (plan_tiles(hp, B, 128) * (hp.N / 64) >= 200) {
      if (ws_ok) return launch_conv_ws_cfg<128, 64, 2, 2>(h, hp, dp, B, src, W, ldw, dst, epi, st);
      RD_CONV(128, 64, 2, 2);
    }
    // few rows, very long K (input gradient of the first generator block): producer/consumer kernel with its K split
    if (ws_ok && h->ws_ksplit && (hp.nphases == 1 || hp.boxes) && hp.d_cstride == hp.N && (long)hp.ph[0].ntaps * (hp.SC / 32) >= 96 &&
        plan_tiles(hp, B, 128) * (hp.N / 64) >= 32)
      return launch_conv_ws_cfg<128, 64, 2, 2>(h, hp, dp, B, src, W, ldw, dst, epi, st);
    RD_CONV(64, 64, 2, 2);
  }
  if (hp.N == 32 && !partial && !shift && epi.mode == RD_EPI_TAPGATHER)
    return launch_conv_cfg<256, 32, 4, 1, 32, false, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
  if (hp.N == 32 && !partial && !shift)
    return launch_conv_cfg<128, 32, 4, 1, 32, false, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
#undef RD_CONV
  return bad_arg(h, "conv: unsupported N");
}

// bf16 storage mode: the few GEMMs that stay on the fp32 matrix pipe (the generator's Dense layer and the first critic
// layer read fp32 inputs and write bf16; the 64 -> 1 conv reads bf16 and writes its fp32 tap sums)
static int launch_conv_a16(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const float* src, const float* W,
                           int ldw, float* dst, const RdEpi& epi, hipStream_t st, int tag, bool src16, bool out16) {
  ProfScope ps(h, tag, st);
  LaunchScope ls(h, plan_index(h, hp), RD_KIND_CONV, B, plan_flops(hp, B), st);
  if (h) h->flops_acc += plan_flops(hp, B);
  const bool partial = (hp.SC & 3) != 0;
  if (hp.s_shift) return bad_arg(h, "conv (bf16 storage): folded upsample is not supported");
  if (!src16 && out16) {
    if (hp.SC < 32 && hp.SC != 27) {      // first critic layer: BK = 8
      if (hp.N % 64) return bad_arg(h, "conv (bf16 storage): unsupported BK=8 case");
      if (partial) return launch_conv_cfg<128, 64, 2, 2, 8, true, false, false, true>(h, hp, dp, B, src, W, ldw, dst, epi, st);
      return launch_conv_cfg<128, 64, 2, 2, 8, false, false, false, true>(h, hp, dp, B, src, W, ldw, dst, epi, st);
    }
    if (!partial && hp.N % 64 == 0)        // Dense
      return launch_conv_cfg<64, 64, 2, 2, 32, false, false, false, true>(h, hp, dp, B, src, W, ldw, dst, epi, st);
  }
  if (src16 && !out16 && hp.N == 32 && !partial) {     // last generator conv as a column GEMM
    if (epi.mode == RD_EPI_TAPGATHER) return launch_conv_cfg<256, 32, 4, 1, 32, false, false, true, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
    return launch_conv_cfg<128, 32, 4, 1, 32, false, false, true, false>(h, hp, dp, B, src, W, ldw, dst, epi, st);
  }
  return bad_arg(h, "conv (bf16 storage): unsupported fp32-pipe GEMM");
}

static inline int ew_blocks(long n, int per = 256);
// critic input (k_build_critic_input): the four-voxel kernel where the layout allows it
static void launch_build_critic_input(rdgan_handle* h, const float* real, const float* fake, const float* cond, int B, int mode,
                                      uint32_t akey, uint32_t abase, hipStream_t st) {
  const int D = h->ddim[0][0], HW = h->nd * h->nd;
  const long total = (long)B * D * HW;
  if (h->nc == 1 && h->CP == 2 && HW % 4 == 0 && total < 0x7FFFFFFFL)
    hipLaunchKernelGGL(k_build_critic_input_v4, dim3(ew_blocks(total / 4)), dim3(256), 0, st, real, fake, cond, h->cin, B, D, HW, mode,
                       akey, abase);
  else
    hipLaunchKernelGGL(k_build_critic_input, dim3(ew_blocks(total)), dim3(256), 0, st, real, fake, cond, h->cin, B, D, HW, h->nc, h->CP,
                       mode, akey, abase);
}
static inline int ew_blocks(long n, int per) { return (int)std::min<long>((n + per - 1) / per, 8192); }

// k_conv_gemm_f16 (rdgan_gemm_f16.hip.h): when a launch may take it
static bool conv_f16_ok(const rdgan_handle* h, const RdPlan& hp, int B, const RdEpi& epi) {
  if (h && !h->conv_f16) return false;
  if (hp.s_shift || hp.SC % 64 || hp.N % 128 || ((hp.N / 128) & (hp.N / 128 - 1)) || hp.w_rows_per_tap != hp.SC) return false;
  if (!epi.out16 || epi.addt) return false;
  if (epi.mode == RD_EPI_BIAS_PN_LRELU ? hp.N != 128 : epi.mode > RD_EPI_GATE_AUX) return false;
  for (int i = 0; i < hp.nphases; ++i) if (hp.ph[i].ntaps > 64) return false;
  // 256-row tiles, 512 resident at a time: below ~640 workgroups the streaming kernel (half the tile, its own K split) was the
  // faster one on every launch measured (scratch/f16_ab.py); conv_f16 = 2 (tests): regardless
  return (h && h->conv_f16 == 2) || !h || plan_tiles(hp, B, RD_F16_BM) * (hp.N / 128) >= 640;
}
static int launch_conv_f16(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const void* src16, const void* wfrag, float* dst,
                           const RdEpi& epi, hipStream_t st) {
  const bool pn = epi.mode == RD_EPI_BIAS_PN_LRELU;
  auto kern = pn ? k_conv_gemm_f16<true> : k_conv_gemm_f16<false>;
  int maxtaps = 0;
  for (int i = 0; i < hp.nphases; ++i) maxtaps = std::max(maxtaps, hp.ph[i].ntaps);
  const int tg = maxtaps <= 4 ? 4 : 8;                 // (the streaming kernel's tap groups: same chunk order, same sums)
  RD_KNAME(h, "k_conv_gemm_f16<256,128,bf16%s>", pn ? ",+pn" : "");
  RD_TRY(ensure_lds(h, (const void*)kern, RD_F16_LDS));
  long tm = plan_tiles(hp, B, RD_F16_BM);
  if (tm <= 0) return 0;
  long minL = hp.ph[0].L;
  for (int i = 1; i < hp.nphases; ++i) minL = std::min<long>(minL, hp.ph[i].L);
  if (std::min<long>(B, RD_F16_BM / minL + 2) * hp.src_sample * 4 >= 0x7FFFFFF0L) return bad_arg(h, "conv: source tile span exceeds 2 GiB");
  if (std::min<long>(B, RD_F16_BM / minL + 2) * hp.dst_sample * 4 >= 0x7FFFFFF0L) return bad_arg(h, "conv: destination tile span exceeds 2 GiB");
  RdEpi e2 = epi;
  e2.ksplit = 1; e2.kpart = nullptr; e2.kstride = 0;
  const long blocks = tm * (hp.N / 128);
  const long total = (long)B * hp.dst_sample;
  if (h && h->ws_ksplit > 1 && !pn && hp.d_cstride == hp.N) {        // tests: force a split
    long nch = (long)hp.ph[0].ntaps * (hp.SC / 64);
    for (int i = 1; i < hp.nphases; ++i) nch = std::min<long>(nch, (long)hp.ph[i].ntaps * (hp.SC / 64));
    const long best = std::min<long>(h->ws_ksplit, std::max<long>(1, nch / 4));
    if (best > 1 && (size_t)(best * total) <= h->kpartial_cap) { e2.ksplit = (int)best; e2.kpart = h->kpartial; e2.kstride = total; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)e2.ksplit), dim3(256), RD_F16_LDS, st, dp, B, (const rd_bf16_t*)src16,
                     (const rd_bf16_t*)wfrag, (rd_bf16_t*)dst, e2, tg);
  if (e2.ksplit > 1)
    hipLaunchKernelGGL(k_splitk_finish<true>, dim3((unsigned)std::min<long>((total / 4 + 255) / 256, 2048)), dim3(256), 0, st, dst, total, hp.N, e2);
  RD_CHECK(h, hipGetLastError());
  return 0;
}
// [T][N][K] bf16 -> fragment order (the twin image k_conv_gemm_f16 reads)
static int launch_wfrag_image(rdgan_handle* h, const void* in, void* out, long T, int N, int K, hipStream_t st) {
  hipLaunchKernelGGL(k_wfrag_image, dim3((unsigned)std::min<long>((T * N * K / 8 + 255) / 256, 2048)), dim3(256), 0, st,
                     (const unsigned short*)in, (unsigned short*)out, T, N, K);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// bf16-operand conv GEMM (fp32 accumulate / output): src16 = bf16 NDHWC activations, w16 = bf16 weights [tap block][N][K];
// wfrag = the same weights in fragment order, or nullptr
static int launch_conv16(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const void* src16, const void* w16,
                         float* dst, const RdEpi& epi, hipStream_t st, int tag, const void* wfrag = nullptr) {
  ProfScope ps(h, tag, st);
  LaunchScope ls(h, plan_index(h, hp), RD_KIND_CONV, B, plan_flops(hp, B), st);
  if (h) h->flops_acc += plan_flops(hp, B);
  if (hp.s_shift || hp.SC % 64 || hp.N % 64) return bad_arg(h, "conv16: needs SC % 64 == 0, N % 64 == 0, no folded upsample");
  if (wfrag && conv_f16_ok(h, hp, B, epi)) return launch_conv_f16(h, hp, dp, B, src16, wfrag, dst, epi, st);
  const float* s = (const float*)src16; const float* w = (const float*)w16;
  if (hp.N % 128 == 0) return launch_conv_ws_cfg<128, 128, 2, 2, true>(h, hp, dp, B, s, w, 0, dst, epi, st);
  if (hp.N == 64 && plan_tiles(hp, B, 256) >= 512) {
    const bool res = (!h || h->resident) && conv16_resident_ok(hp, 256, 64);
    return launch_conv_ws_cfg<256, 64, 4, 1, true>(h, hp, dp, B, s, w, 0, dst, epi, st, res);
  }
  return launch_conv_ws_cfg<128, 64, 2, 2, true>(h, hp, dp, B, s, w, 0, dst, epi, st);
}
// fp32 -> bf16 copies (round to nearest even) of an activation tensor and of a weight-form stack [T][K][N] -> [T][N][K]
static int launch_to_bf16(rdgan_handle* h, const float* in, void* out, long n, hipStream_t st) {
  hipLaunchKernelGGL(k_to_bf16, dim3(ew_blocks(n / 8)), dim3(256), 0, st, in, (unsigned short*)out, n);
  RD_CHECK(h, hipGetLastError());
  return 0;
}
static int launch_weights_to_bf16_t(rdgan_handle* h, const float* in, void* out, int T, int K, int N, hipStream_t st, void* outf = nullptr) {
  hipLaunchKernelGGL(k_weights_to_bf16_t, dim3((N + 31) / 32, (K + 31) / 32, T), dim3(256), 0, st, in, (unsigned short*)out, K, N,
                     (unsigned short*)outf);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// true when launch_conv will pick a tile whose BN equals the plan's N, i.e. a workgroup owns whole output rows
static bool conv_rows_owned(const RdPlan& hp, int B) {
  if (hp.s_shift || (hp.SC & 3) || hp.SC < 32) return false;
  if (hp.N == 128) return plan_tiles(hp, B, 128) >= 200;
  if (hp.N == 64) return plan_tiles(hp, B, 256) >= 1024 || plan_tiles(hp, B, 128) >= 200;
  return false;
}

// the same for launch_conv16 (bf16 storage mode): tiles 128x128 (N % 128 == 0), 256x64 / 128x64 (N == 64)
static bool conv16_rows_owned(const RdPlan& hp) { return hp.N == 128 || hp.N == 64; }


template <int BR, int BN, bool PARTIAL, bool SHIFT, bool DY16 = false>
static int launch_wgrad_cfg(rdgan_handle* h, const RdPlan* dp, int nphases, int B, const float* src, const float* dy,
                            float* partial, const RdWgradTiling& T, int nsplit, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)(32 * BR + 32 * BN) * sizeof(float);
  auto kern = k_wgrad_gemm<BR, BN, PARTIAL, SHIFT, DY16>;
  RD_KNAME(h, "k_wgrad_gemm<%d,%d>", BR, BN);
  RD_TRY(ensure_lds(h, (const void*)kern, lds));
  dim3 grid((unsigned)(T.RT * T.NT * nsplit * nphases));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, dp, B, src, dy, partial, T);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

template <int BR, int BN>
static int launch_wgrad_ws_cfg(rdgan_handle* h, const RdPlan* dp, int nphases, int B, const float* src, const float* dy,
                               float* partial, const RdWgradTiling& T, int nsplit, hipStream_t st) {
  constexpr size_t lds_loop = 2 * (size_t)(32 * BR + 32 * BN) * sizeof(float);
  constexpr size_t lds_epi = (size_t)BR * BN * sizeof(float);
  constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  auto kern = k_wgrad_gemm_ws<BR, BN>;
  RD_KNAME(h, "k_wgrad_gemm_ws<%d,%d>", BR, BN);
  RD_TRY(ensure_lds(h, (const void*)kern, lds));
  dim3 grid((unsigned)(T.box ? T.nsplit : T.RT * T.NT * nsplit * nphases));
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, dp, B, src, dy, partial, T);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// dW (rows tap_w*wrpt + c, leading dimension ldw = N) from src (gathered through the plan) and dy
static int launch_wgrad(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const float* src, const float* dy,
                        float* dW, float* partial_ws, size_t partial_cap, hipStream_t st, int tag, bool dy16 = false) {
  ProfScope ps(h, tag, st);
  LaunchScope ls(h, plan_index(h, hp), RD_KIND_WGRAD, B, plan_flops(hp, B), st);
  if (h) h->flops_acc += plan_flops(hp, B);
  if (hp.N % 64) return bad_arg(h, "wgrad: N % 64 != 0");
  if (!hp.boxes)
  for (int i = 1; i < hp.nphases; ++i)
    if (hp.ph[i].ntaps != hp.ph[0].ntaps || hp.ph[i].L != hp.ph[0].L) return bad_arg(h, "wgrad: phases must be congruent");
  int BR, BN, nsplit;
  RdWgradTiling T = wgrad_tiling(hp, B, BR, BN, nsplit);
  size_t need = T.box ? (size_t)T.RT * BR * hp.N : (size_t)hp.nphases * nsplit * T.RT * BR * hp.N;
  if (T.box && !(h && h->wave_spec && (hp.SC & 3) == 0 && !hp.s_shift && !dy16))
    return bad_arg(h, "wgrad: border-class boxes only through k_wgrad_gemm_ws");
  if (need > partial_cap) return bad_arg(h, "wgrad: partial workspace too small");
  long minL = hp.ph[0].L;
  for (int i = 1; i < hp.nphases; ++i) minL = std::min<long>(minL, hp.ph[i].L);
  if (std::min<long>(B, T.rows_per_split / minL + 2) * std::max(hp.src_sample, hp.dst_sample) * 4 >= 0x7FFFFFF0L)
    return bad_arg(h, "wgrad: split span exceeds 2 GiB");
  const int np = hp.nphases;
  const bool partial = (hp.SC & 3) != 0, shift = hp.s_shift != 0;
  if (partial && (shift || BR != 64 || BN != 64)) return bad_arg(h, "wgrad: SC % 4 != 0 only with the 64x64 tile");
#define RD_WG(BR_, BN_)                                                                                           \
  do {                                                                                                            \
    if (shift) RD_TRY((launch_wgrad_cfg<BR_, BN_, false, true>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st))); \
    else RD_TRY((launch_wgrad_cfg<BR_, BN_, false, false>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));      \
  } while (0)
  float* partial_buf = partial_ws;
  const long wrows = (long)B * hp.ph[0].L;
  const bool ws = h && h->wave_spec && !partial && !shift && !dy16 && (h->wave_spec == 2 || wrows * hp.nphases >= 1024 || T.box);
  // one phase in one split through k_wgrad_gemm: the tiles go straight to dW, no partial slab, no fold (the Dense layer's 64-row K
  // loop at ndomain 64: the slab was 825 MB written, read and written again)
  const bool direct = !ws && !T.box && !dy16 && !partial && np == 1 && nsplit == 1 && h != nullptr && hp.ph[0].w_off == 0;
  if (direct) { T.direct = 1; T.ldw = hp.N; partial_buf = dW; }
  if (dy16) {       // bf16 storage mode, first critic layer: fp32 gathered input against the bf16 output gradient
    if (shift || BR != 64 || BN != 64) return bad_arg(h, "wgrad: bf16 output gradient only with the 64x64 tile");
    if (partial) RD_TRY((launch_wgrad_cfg<64, 64, true, false, true>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
    else RD_TRY((launch_wgrad_cfg<64, 64, false, false, true>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
  }
  else if (ws && BR == 256) RD_TRY((launch_wgrad_ws_cfg<256, 64>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
  else if (ws && BR == 128 && BN == 128) RD_TRY((launch_wgrad_ws_cfg<128, 128>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
  else if (ws && BR == 128) RD_TRY((launch_wgrad_ws_cfg<128, 64>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
  else if (ws && BN == 128) RD_TRY((launch_wgrad_ws_cfg<64, 128>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
  else if (ws) RD_TRY((launch_wgrad_ws_cfg<64, 64>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
  else if (partial) RD_TRY((launch_wgrad_cfg<64, 64, true, false>(h, dp, np, B, src, dy, partial_buf, T, nsplit, st)));
  else if (BR == 256) RD_WG(256, 64);
  else if (BR == 128 && BN == 128) RD_WG(128, 128);
  else if (BR == 128) RD_WG(128, 64);
  else if (BN == 128) RD_WG(64, 128);
  else RD_WG(64, 64);
#undef RD_WG
  if (direct) { RD_CHECK(h, hipGetLastError()); return 0; }
  if (T.box) {      // per weight tap: the slabs of every phase that lists it
    const int nw = 64 - __builtin_clzll(hp.wmask | 1ull);
    const long nout = (long)nw * hp.SC * (hp.N / 4);
    hipLaunchKernelGGL(k_wgrad_reduce_box, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, dp, partial_ws, T, BR, B, nw, dW, hp.N);
    RD_CHECK(h, hipGetLastError());
    return 0;
  }
  long total = (long)T.RT * BR * (hp.N / 4);
  int outs = 256;                       // output float4s per workgroup; the other 256/outs thread slices split the fold
  while (outs > 16 && (total + outs - 1) / outs * np < 512 && 256 / outs < nsplit) outs >>= 1;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + outs - 1) / outs), np), dim3(256), 0, st, dp, partial_ws,
                     nsplit, T, BR, dW, hp.N, outs);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// bf16-operand weight gradient (mixed mode): src16 / dy16 = bf16 copies of the gathered tensor and of the output gradient;
// same tiling, partial slabs and fold as launch_wgrad.  Tiles 256x64, 128x128, 128x64.
template <int BR, int BN>
static int launch_wgrad16_cfg(rdgan_handle* h, const RdPlan* dp, int nphases, int B, const void* src16, const void* dy16,
                              float* partial, const RdWgradTiling& T, int nsplit, hipStream_t st) {
  constexpr size_t lds_loop = ((BR == 256 && BN == 128) ? 3 : 2) * (size_t)(64 * BR * 2 + 64 * BN * 2);
  constexpr size_t lds_epi = (size_t)BR * BN * sizeof(float);
  constexpr size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  auto kern = k_wgrad_gemm_ws16<BR, BN>;
  RD_KNAME(h, "k_wgrad_gemm_ws16<%d,%d>", BR, BN);
  RD_TRY(ensure_lds(h, (const void*)kern, lds));
  dim3 grid((unsigned)(T.box ? T.nsplit : T.RT * T.NT * nsplit * nphases));
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, dp, B, (const unsigned short*)src16, (const unsigned short*)dy16, partial, T);
  RD_CHECK(h, hipGetLastError());
  return 0;
}
static bool wgrad16_ok(const RdPlan& hp, int B) {
  int BR, BN, nsplit;
  if (hp.s_shift || hp.SC % 64 || hp.N % 64) return false;
  wgrad_tiling(hp, B, BR, BN, nsplit);
  return BR >= 128;
}
static bool wgrad16_wide(const rdgan_handle* h) { return !h || h->wgrad_wide; }     // (op-level entries: on, so that the tile stays tested)
static int launch_wgrad16(rdgan_handle* h, const RdPlan& hp, const RdPlan* dp, int B, const void* src16, const void* dy16,
                          float* dW, float* partial_ws, size_t partial_cap, hipStream_t st, int tag) {
  ProfScope ps(h, tag, st);
  LaunchScope ls(h, plan_index(h, hp), RD_KIND_WGRAD, B, plan_flops(hp, B), st);
  if (h) h->flops_acc += plan_flops(hp, B);
  if (!wgrad16_ok(hp, B)) return bad_arg(h, "wgrad16: needs SC % 64 == 0, N % 64 == 0, a 128- or 256-row tile, no folded upsample");
  if (!hp.boxes)
  for (int i = 1; i < hp.nphases; ++i)
    if (hp.ph[i].ntaps != hp.ph[0].ntaps || hp.ph[i].L != hp.ph[0].L) return bad_arg(h, "wgrad: phases must be congruent");
  int BR, BN, nsplit;
  RdWgradTiling T = wgrad_tiling(hp, B, BR, BN, nsplit, wgrad16_wide(h));
  size_t need = T.box ? (size_t)T.RT * BR * hp.N : (size_t)hp.nphases * nsplit * T.RT * BR * hp.N;
  if (need > partial_cap) return bad_arg(h, "wgrad: partial workspace too small");
  long minL16 = hp.ph[0].L;
  for (int i = 1; i < hp.nphases; ++i) minL16 = std::min<long>(minL16, hp.ph[i].L);
  if (std::min<long>(B, T.rows_per_split / minL16 + 2) * std::max(hp.src_sample, hp.dst_sample) * 2 >= 0x7FFFFFF0L)
    return bad_arg(h, "wgrad: split span exceeds 2 GiB");
  const int np = hp.nphases;
  if (BR == 256 && BN == 128) RD_TRY((launch_wgrad16_cfg<256, 128>(h, dp, np, B, src16, dy16, partial_ws, T, nsplit, st)));
  else if (BR == 256) RD_TRY((launch_wgrad16_cfg<256, 64>(h, dp, np, B, src16, dy16, partial_ws, T, nsplit, st)));
  else if (BN == 128) RD_TRY((launch_wgrad16_cfg<128, 128>(h, dp, np, B, src16, dy16, partial_ws, T, nsplit, st)));
  else RD_TRY((launch_wgrad16_cfg<128, 64>(h, dp, np, B, src16, dy16, partial_ws, T, nsplit, st)));
  if (T.box) {      // per weight tap: the slabs of every phase that lists it
    const int nw = 64 - __builtin_clzll(hp.wmask | 1ull);
    const long nout = (long)nw * hp.SC * (hp.N / 4);
    hipLaunchKernelGGL(k_wgrad_reduce_box, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, dp, partial_ws, T, BR, B, nw, dW, hp.N);
    RD_CHECK(h, hipGetLastError());
    return 0;
  }
  long total = (long)T.RT * BR * (hp.N / 4);
  int outs = 256;
  while (outs > 16 && (total + outs - 1) / outs * np < 512 && 256 / outs < nsplit) outs >>= 1;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + outs - 1) / outs), np), dim3(256), 0, st, dp, partial_ws,
                     nsplit, T, BR, dW, hp.N, outs);
  RD_CHECK(h, hipGetLastError());
  return 0;
}


// out[c] = sum over rows of src[rows][C]
static int launch_colsum(rdgan_handle* h, const float* src, long rows, int C, float* out, hipStream_t st, bool src16 = false) {
  ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
  const bool any = C != 64 && C != 128 && C != 256;
  if (src16 && any) return bad_arg(h, "colsum: bf16 input only for 64 / 128 / 256 columns");
  // (round 4 tried at most 256 partial rows with 1024-thread workgroups so that the fold is one round trip: beside the GEMMs of the
  // main stream such a workgroup waits for sixteen free wave slots on one CU -- block 3's sums went 62 -> 120-250 us; the fold got
  // small workgroups instead: k_reduce_partials4)
  long nblk = std::min<long>(1024, std::max<long>(1, rows / (any ? 8 : 32)));
  if ((size_t)nblk * C > h->cpartial_cap) nblk = std::max<long>(1, (long)(h->cpartial_cap / C));
  long rpb = (rows + nblk - 1) / nblk;
  nblk = (rows + rpb - 1) / rpb;
  dim3 grid((unsigned)nblk);
  const rd_bf16_t* s16 = (const rd_bf16_t*)src;
  const bool big = false;
#define RD_COLSUM(CG_, T_, P_)                                                                                                    \
  do {                                                                                                                            \
    if (big) hipLaunchKernelGGL((k_colsum_partial<CG_, T_, 1024>), grid, dim3(1024), 0, st, P_, rows, h->cpartial, rpb);          \
    else hipLaunchKernelGGL((k_colsum_partial<CG_, T_, 256>), grid, dim3(256), 0, st, P_, rows, h->cpartial, rpb);                \
  } while (0)
  if (src16 && C == 64) RD_COLSUM(16, rd_bf16_t, s16);
  else if (src16 && C == 128) RD_COLSUM(32, rd_bf16_t, s16);
  else if (src16) RD_COLSUM(64, rd_bf16_t, s16);
  else if (C == 64) RD_COLSUM(16, float, src);
  else if (C == 128) RD_COLSUM(32, float, src);
  else if (C == 256) RD_COLSUM(64, float, src);
#undef RD_COLSUM
  else {
    const int bt = std::max(64, std::min(256, (C + 63) / 64 * 64));
    hipLaunchKernelGGL(k_colsum_partial_any, dim3((unsigned)nblk, (C + bt - 1) / bt), dim3(bt), 0, st, src, rows, C, h->cpartial, rpb);
  }
  if (!any && nblk <= 1024)
    hipLaunchKernelGGL(k_reduce_partials4, dim3(C / 4), dim3(256), 0, st, h->cpartial, (int)nblk, C, out);
  else
  hipLaunchKernelGGL(k_reduce_partials, dim3((C + 15) / 16), dim3(rd_reduce_threads((int)nblk)), 0, st, h->cpartial, (int)nblk, C, out);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

static int launch_transpose(rdgan_handle* h, const float* in, float* out, int T, int R, int C, int ldo, hipStream_t st) {
  dim3 grid((C + 31) / 32, (std::max(R, ldo) + 31) / 32, T);
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, st, in, out, R, C, ldo);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

static int launch_pn_fwd(rdgan_handle* h, const float* y, float* hout, float* rinv, long npix, int C, hipStream_t st, bool a16 = false) {
  long threads = npix * (C / 4);
  dim3 grid((unsigned)((threads + 255) / 256));
  const rd_bf16_t* y16 = (const rd_bf16_t*)y; rd_bf16_t* h16 = (rd_bf16_t*)hout;
  if (a16) {
    if (C == 256) hipLaunchKernelGGL((k_pixelnorm_lrelu_fwd<64, rd_bf16_t>), grid, dim3(256), 0, st, y16, h16, rinv, npix);
    else if (C == 128) hipLaunchKernelGGL((k_pixelnorm_lrelu_fwd<32, rd_bf16_t>), grid, dim3(256), 0, st, y16, h16, rinv, npix);
    else if (C == 64) hipLaunchKernelGGL((k_pixelnorm_lrelu_fwd<16, rd_bf16_t>), grid, dim3(256), 0, st, y16, h16, rinv, npix);
    else return bad_arg(h, "pixelnorm: C must be 64/128/256");
  }
  else if (C == 256) hipLaunchKernelGGL(k_pixelnorm_lrelu_fwd<64>, grid, dim3(256), 0, st, y, hout, rinv, npix);
  else if (C == 128) hipLaunchKernelGGL(k_pixelnorm_lrelu_fwd<32>, grid, dim3(256), 0, st, y, hout, rinv, npix);
  else if (C == 64) hipLaunchKernelGGL(k_pixelnorm_lrelu_fwd<16>, grid, dim3(256), 0, st, y, hout, rinv, npix);
  else return bad_arg(h, "pixelnorm: C must be 64/128/256");
  RD_CHECK(h, hipGetLastError());
  return 0;
}
static int launch_pn_bwd(rdgan_handle* h, const float* g, const float* hh, const float* rinv, float* dy, long npix, int C,
                         int pool, int D, int H, int W, hipStream_t st, bool a16 = false) {
  long threads = npix * (C / 4);
  dim3 grid((unsigned)((threads + 255) / 256));
#define RD_PNB(LP, PO) hipLaunchKernelGGL((k_pn_lrelu_bwd<LP, PO>), grid, dim3(256), 0, st, g, hh, rinv, dy, npix, D, H, W)
#define RD_PNB16(LP) hipLaunchKernelGGL((k_pn_lrelu_bwd<LP, 0, rd_bf16_t>), grid, dim3(256), 0, st, (const rd_bf16_t*)g, \
                                        (const rd_bf16_t*)hh, rinv, (rd_bf16_t*)dy, npix, D, H, W)
  if (a16) {
    if (pool) return bad_arg(h, "pixelnorm bwd: the bf16 storage mode needs the collapsed form");
    if (C == 256) RD_PNB16(64); else if (C == 128) RD_PNB16(32); else if (C == 64) RD_PNB16(16);
    else return bad_arg(h, "pixelnorm bwd: C must be 64/128/256");
  }
  else if (C == 256) { if (pool) RD_PNB(64, 1); else RD_PNB(64, 0); }
  else if (C == 128) { if (pool) RD_PNB(32, 1); else RD_PNB(32, 0); }
  else if (C == 64) { if (pool) RD_PNB(16, 1); else RD_PNB(16, 0); }
  else return bad_arg(h, "pixelnorm bwd: C must be 64/128/256");
#undef RD_PNB
#undef RD_PNB16
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// PixelNorm+LeakyReLU backward over hour-plane pairs, also writing the pair sums gS (shared-centre backward)
static int launch_pn_bwd_pairs(rdgan_handle* h, const float* g, const float* hh, const float* rinv, float* dy, float* gS,
                               long npair, long HW, int C, hipStream_t st, bool a16 = false) {
  long threads = npair * (C / 4);
  dim3 grid((unsigned)((threads + 255) / 256));
#define RD_PNP16(LP) hipLaunchKernelGGL((k_pn_lrelu_bwd_pairs<LP, rd_bf16_t>), grid, dim3(256), 0, st, (const rd_bf16_t*)g, \
                                        (const rd_bf16_t*)hh, rinv, (rd_bf16_t*)dy, (rd_bf16_t*)gS, npair, HW)
  if (a16) {
    if (C == 256) RD_PNP16(64); else if (C == 128) RD_PNP16(32); else if (C == 64) RD_PNP16(16);
    else return bad_arg(h, "pixelnorm bwd: C must be 64/128/256");
  }
  else if (C == 256) hipLaunchKernelGGL(k_pn_lrelu_bwd_pairs<64>, grid, dim3(256), 0, st, g, hh, rinv, dy, gS, npair, HW);
  else if (C == 128) hipLaunchKernelGGL(k_pn_lrelu_bwd_pairs<32>, grid, dim3(256), 0, st, g, hh, rinv, dy, gS, npair, HW);
  else if (C == 64) hipLaunchKernelGGL(k_pn_lrelu_bwd_pairs<16>, grid, dim3(256), 0, st, g, hh, rinv, dy, gS, npair, HW);
  else return bad_arg(h, "pixelnorm bwd: C must be 64/128/256");
#undef RD_PNP16
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------

extern "C" int rdgan_create(rdgan_handle** out, int ndomain, int n_cond_channels, int max_batch) {
  if (!out) return -2;
  *out = nullptr;
  if (!rd_geometry_ok(ndomain, n_cond_channels, max_batch)) return -2;
  rdgan_handle* h = new rdgan_handle();
  rd_geometry(h, ndomain, n_cond_channels, max_batch);
  const int nd = ndomain;
  const int* gch = h->gch; const int* dch = h->dch;
  hipError_t e = hipSuccess;
  {
    std::vector<RdRow> tab;
    std::vector<size_t> first;
    if (!rd_build_plans(h, h->plans, tab, first)) { delete h; return -2; }
    e = hipMalloc((void**)&h->d_tab, sizeof(RdRow) * tab.size());
    if (e == hipSuccess) e = hipMemcpy(h->d_tab, tab.data(), sizeof(RdRow) * tab.size(), hipMemcpyHostToDevice);
    for (int i = 0; i < PL_COUNT; ++i) h->plans[i].tab = h->d_tab + first[i];
  }
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_plans, sizeof(RdPlan) * PL_COUNT);
  if (e == hipSuccess) e = hipMemcpy(h->d_plans, h->plans.data(), sizeof(RdPlan) * PL_COUNT, hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (h->d_tab) (void)hipFree(h->d_tab); if (h->d_plans) (void)hipFree(h->d_plans); delete h; return (int)e; }

  // workspace carve (two passes: size, then assign)
  const long MB = h->MB, NB = h->NB;
  // partial slabs of the streaming weight-gradient kernels: the worst case over every batch size up to max_batch
  size_t wneed = rd_wgrad_workspace_floats(h, h->plans);
  wneed = std::max(wneed, (size_t)MB * (RDGAN_NHOURS / 2) * 1728);        // k_g9_wgrad_pairs: [27][64] per (sample, plane pair)
  if (g9w_mfma_ok(nd, (long)MB * h->gpix[3]))                              // k_g9_wgrad_mfma: [27][64] per persistent workgroup
    wneed = std::max(wneed, (size_t)std::min<long>(((long)MB * h->gpix[3] + 127) / 128, 768) * 1728);
  if (nd == 16) wneed = std::max(wneed, (size_t)16 * 27 * RD_D3W_TILE);    // k_d3_wgrad_slab16: [16 groups][27][128][256]
  if (nd % 16 == 0) wneed = std::max(wneed, (size_t)64 * 27 * RD_D2W_TILE);    // k_d2_wgrad_slab16 / _t16: [64 groups][27][64][128]
  if (nd == 16) wneed = std::max(wneed, (size_t)8 * 64 * RD_UW2_TILE);      // k_upconv2_wgrad_slab16: [8 groups][64][256][128]
  if (nd == 16) wneed = std::max(wneed, (size_t)32 * 64 * RD_UWG_TILE);     // k_upconv_wgrad_slab16: [32 groups][64][128][64]
  h->wpartial_cap = wneed;
  h->cpartial_cap = (size_t)1024 * std::max(h->n_nodes, 256);
  {  // split-K partials: up to 8 copies of the largest small-M destination (critic layers 3/4, Dense, generator block 1)
    size_t m = std::max<size_t>((size_t)NB * h->dL[3] * 256, (size_t)MB * h->n_nodes);
    m = std::max<size_t>(m, (size_t)NB * h->dL[2] * 128);
    m = std::max<size_t>(m, (size_t)MB * h->gpix[1] * 256);
    h->kpartial_cap = 8 * m;
  }
  for (int pass = 0; pass < 2; ++pass) {
    size_t off = 0;
    auto carve = [&](float*& p, size_t nfloats) {
      off = (off + 255) & ~(size_t)255;
      if (pass == 1) p = (float*)(h->ws + off);
      off += nfloats * sizeof(float);
    };
    carve(h->xcat, MB * h->n_in);
    carve(h->h0, MB * h->gpix[0] * 256);
    carve(h->h1, MB * h->gpix[1] * 256); carve(h->r1, MB * h->gpix[1]);
    carve(h->h2, MB * h->gpix[2] * 128); carve(h->r2, MB * h->gpix[2]);
    carve(h->h3, MB * h->gpix[3] * 64); carve(h->r3, MB * h->gpix[3]);
    carve(h->P9, MB * h->gpix[3] * 32);
    carve(h->fake, MB * h->gpix[3]);
    carve(h->dl, MB * h->gpix[3]);
    carve(h->gh3, MB * h->gpix[3] * 64);
    carve(h->gup3, MB * h->gpix[3] * 128);
    carve(h->dy2, MB * h->gpix[2] * 128);
    carve(h->gup2, MB * h->gpix[2] * 256);
    carve(h->dy1, MB * h->gpix[1] * 256);
    carve(h->gup1, MB * h->gpix[1] * 256);
    carve(h->ga0, MB * h->gpix[0] * 256);
    carve(h->cin, NB * h->dL[0] * h->CP);
    h->dh[0] = nullptr; h->du[0] = nullptr;
    for (int l = 1; l <= 4; ++l) { carve(h->dh[l], NB * h->dL[l] * dch[l]); carve(h->du[l], NB * h->dL[l] * dch[l]); }
    carve(h->v, NB);
    carve(h->P1, MB * h->dL[1] * h->ldp1);
    carve(h->g0, MB * h->dL[0]);
    carve(h->gpv, MB);
    carve(h->wpartial, h->wpartial_cap);
    carve(h->cpartial, h->cpartial_cap);
    carve(h->kpartial, h->kpartial_cap);
    h->DWT[0] = h->DWT[1] = nullptr;
    for (int l = 2; l <= 4; ++l) carve(h->DWT[l], 27L * dch[l - 1] * dch[l]);
    carve(h->W1T, 64 * h->ldp1);
    carve(h->W1P, 27L * h->CP * 64); carve(h->dW1P, 27L * h->CP * 64);
    h->GWT[0] = nullptr;
    for (int l = 1; l <= 3; ++l) carve(h->GWT[l], 27L * gch[l - 1] * gch[l]);
    carve(h->W9T, 64 * 32);
    h->GWC[0] = h->GWD[0] = nullptr;
    for (int l = 1; l <= 3; ++l) { carve(h->GWC[l], 64L * gch[l - 1] * gch[l]); carve(h->GWD[l], 64L * gch[l - 1] * gch[l]); }
    carve(h->dWc, 64L * 256 * 256);
    {
      size_t ne = 0, ns = 0;
      for (int l = 1; l <= 3; ++l) {
        const int* sd = h->gdim[l - 1];
        ne = std::max(ne, (size_t)(sd[0] + 1) * sd[1] * sd[2] * gch[l - 1]);
        ns = std::max(ns, (size_t)sd[0] * 4 * sd[1] * sd[2] * gch[l]);
      }
      h->fE[0] = h->fU[0] = nullptr;
      for (int l = 1; l <= 3; ++l) {
        const int* sd = h->gdim[l - 1];
        carve(h->fE[l], MB * (size_t)(sd[0] + 1) * sd[1] * sd[2] * gch[l - 1]);
        carve(h->fU[l], 48L * gch[l - 1] * gch[l]);
      }
      carve(h->fdE, MB * ne); carve(h->fgS, MB * ns);
      {   // bf16 weight images (2 bytes per element: half the floats)
        float* p = nullptr;
        h->bU[0] = nullptr;
        for (int l = 1; l <= 3; ++l) { carve(p, 24L * gch[l - 1] * gch[l] + 8); h->bU[l] = p; }
        carve(p, 24L * 256 * 256 + 8); h->bUT = p;
        const long KP128 = (h->n_in + 127) / 128 * 128;                               // (the skinny Dense kernel's images: K in whole 128s)
        carve(p, (long)((MB + 31) / 32 * 32) * KP128 / 2 + 8); h->xcat16 = p;         // (whole 32-row blocks)
        carve(p, (long)h->n_nodes * KP128 / 2 + 8); h->bW0 = p;
        h->bWF[0] = h->bWF[1] = h->bWB[0] = h->bWB[1] = nullptr;
        for (int l = 2; l <= 4; ++l) {
          carve(p, 27L * dch[l - 1] * dch[l] / 2 + 8); h->bWF[l] = p;
          carve(p, 27L * dch[l - 1] * dch[l] / 2 + 8); h->bWB[l] = p;
        }
        h->bG1F[0] = nullptr;
        for (int l = 1; l <= 3; ++l) { carve(p, 32L * gch[l - 1] * gch[l] + 8); h->bG1F[l] = p; }
        carve(p, 32L * 256 * 256 + 8); h->bG1B = p;
        h->fWF[0] = h->fWF[1] = h->fWB[0] = h->fWB[1] = h->fG1F[0] = nullptr;
        for (int l = 2; l <= 4; ++l) {
          carve(p, 27L * dch[l - 1] * dch[l] / 2 + 8); h->fWF[l] = p;
          carve(p, 27L * dch[l - 1] * dch[l] / 2 + 8); h->fWB[l] = p;
        }
        for (int l = 1; l <= 3; ++l) { carve(p, 32L * gch[l - 1] * gch[l] + 8); h->fG1F[l] = p; }
        carve(p, 32L * 256 * 256 + 8); h->fG1B = p;
        carve(p, 32L * h->ldp1 + 8); h->bW1B = p;
        carve(p, 64L * 8 * 2 * 64 * 4 + 8); h->bW3I = p;
        carve(p, 64L * 8 * 2 * 64 * 4 + 8); h->bW3T = p;
        carve(p, 4L * 64 * 4 + 8); h->bW9I = p;
        carve(p, (long)RD_UP2_KSTEPS * 4 * 64 * 4 + 8); h->bW2I = p;
        carve(p, (long)RD_D2S_KSTEPS * 2 * 64 * 4 + 8); h->bW2S = p;
        if (nd % 16 == 0) { carve(p, NB * h->dL[1] * 4 + 8); if (pass == 1) h->g1bits = (unsigned char*)p; }     // (the domains with a layer-2 input-gradient slab kernel)
        carve(p, (long)RD_D2F_KSTEPS * 4 * 64 * 4 + 8); h->bW2F = p;
      }
      carve(h->fdU, 48L * 256 * 256); carve(h->fUT, 48L * 256 * 256);
    }
    { float* f = nullptr; carve(f, 64); if (pass == 1) h->d_flag = (int*)f; }
    carve(h->g9b_tmp, 64);
    carve(h->ubias_part, 32 * 8 * 64);
    carve(h->gp_part, (size_t)MB * 64);
    carve(h->dw6_part, (size_t)16 * h->F);
    if (pass == 0) {
      h->ws_bytes = off + 256;
      e = hipMalloc((void**)&h->ws, h->ws_bytes);
      if (e != hipSuccess) { (void)hipFree(h->d_plans); delete h; return (int)e; }
      (void)hipMemset(h->ws, 0, h->ws_bytes);
    }
  }
  // side stream + ordering events (no timing); failure to create them only turns the option off
  {
    bool ok = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t* evs[] = {&h->ev_fork, &h->ev_join, &h->ev_cw, &h->ev_g[1], &h->ev_g[2], &h->ev_g[3]};
    for (hipEvent_t* e2 : evs) ok = ok && hipEventCreateWithFlags(e2, hipEventDisableTiming) == hipSuccess;
    if (!ok) { h->side_on = 0; (void)hipGetLastError(); }
  }
  *out = h;
  return 0;
}

extern "C" void rdgan_destroy(rdgan_handle* h) {
  if (!h) return;
  if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
  for (hipEvent_t e2 : {h->ev_fork, h->ev_join, h->ev_cw, h->ev_g[1], h->ev_g[2], h->ev_g[3]}) if (e2) (void)hipEventDestroy(e2);
  for (auto& r : h->launch_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (int t = 0; t < RDGAN_NUM_TAGS; ++t) {
    for (auto e : h->ev_start[t]) (void)hipEventDestroy(e);
    for (auto e : h->ev_stop[t]) (void)hipEventDestroy(e);
  }
  for (void* g : h->gate_keep) if (g) (void)hipFree(g);
  if (h->ws) (void)hipFree(h->ws);
  if (h->d_plans) (void)hipFree(h->d_plans);
  if (h->d_tab) (void)hipFree(h->d_tab);
  delete h;
}

extern "C" const char* rdgan_last_error(const rdgan_handle* h) { return h ? h->err.c_str() : "null handle"; }
extern "C" long rdgan_gen_param_count(const rdgan_handle* h) { return h ? h->n_gen : -1; }
extern "C" long rdgan_critic_param_count(const rdgan_handle* h) { return h ? h->n_critic : -1; }
extern "C" long rdgan_workspace_bytes(const rdgan_handle* h) { return h ? (long)h->ws_bytes : -1; }
extern "C" int rdgan_gen_param_layout(const rdgan_handle* h, long* offsets, long* sizes) {
  if (!h) return -2;
  for (int i = 0; i < 10; ++i) { if (offsets) offsets[i] = h->goff[i]; if (sizes) sizes[i] = h->gsz[i]; }
  return 10;
}
extern "C" int rdgan_critic_param_layout(const rdgan_handle* h, long* offsets, long* sizes) {
  if (!h) return -2;
  for (int i = 0; i < 10; ++i) { if (offsets) offsets[i] = h->doff[i]; if (sizes) sizes[i] = h->dsz[i]; }
  return 10;
}

extern "C" int rdgan_set_option(rdgan_handle* h, const char* name, int value) {
  if (!h || !name) return -2;
  if (!strcmp(name, "collapse")) { h->collapse = value ? 1 : 0; return 0; }
  if (!strcmp(name, "wave_specialized")) { h->wave_spec = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }   // 2 = also for small problems (tests)
  if (!strcmp(name, "bf16") || !strcmp(name, "mfma_bf16")) { h->a16 = value ? 1 : 0; return 0; }   // ("mfma_bf16": round-1 name)
  if (!strcmp(name, "g9_direct")) { h->g9_direct = value ? 1 : 0; return 0; }
  if (!strcmp(name, "fast_fwd")) { h->fast_fwd = value < 0 ? -1 : (value ? 1 : 0); return 0; }     // -1 = by storage mode
  if (!strcmp(name, "fast_bwd")) { h->fast_bwd = value < 0 ? -1 : (value ? 1 : 0); return 0; }
  if (!strcmp(name, "tapgather")) { h->tapgather = value ? 1 : 0; return 0; }
  if (!strcmp(name, "resident")) { h->resident = value ? 1 : 0; return 0; }
  if (!strcmp(name, "upconv_slab")) { h->upconv_slab = value ? 1 : 0; return 0; }
  if (!strcmp(name, "upconv_slab_t")) { h->upconv_slab_t = value ? 1 : 0; return 0; }
  if (!strcmp(name, "upconv2_slab")) { h->upconv2_slab = value ? 1 : 0; return 0; }
  if (!strcmp(name, "g9_fused")) { h->g9_fused = value ? 1 : 0; return 0; }
  if (!strcmp(name, "d1_fwd_sample")) { h->d1_fwd_sample = value ? 1 : 0; return 0; }
  if (!strcmp(name, "g9_bwd_mfma")) { h->g9_bwd_mfma = value ? 1 : 0; return 0; }
  if (!strcmp(name, "dense16")) { h->dense16 = value ? 1 : 0; return 0; }
  if (!strcmp(name, "dense_skinny")) { h->dense_skinny = value ? 1 : 0; return 0; }
  if (!strcmp(name, "wgrad_boxes")) { h->wgrad_boxes = value ? 1 : 0; return 0; }
  if (!strcmp(name, "border_boxes")) { h->border_boxes = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }    // 2 = at every size (tests)
  if (!strcmp(name, "d3_wgrad_slab")) { h->d3_wgrad_slab = value ? 1 : 0; return 0; }
  if (!strcmp(name, "d2_wgrad_slab")) { h->d2_wgrad_slab = value ? 1 : 0; return 0; }
  if (!strcmp(name, "upwgrad_slab")) { h->upwgrad_slab = value ? 1 : 0; return 0; }
  if (!strcmp(name, "d1_dgrad_fused")) { h->d1_dgrad_fused = value ? 1 : 0; return 0; }
  if (!strcmp(name, "d1_wgrad16")) { h->d1_wgrad16 = value ? 1 : 0; return 0; }
  if (!strcmp(name, "wgrad_wide")) { h->wgrad_wide = value ? 1 : 0; return 0; }
  if (!strcmp(name, "conv_f16")) {      // (the forms are rebuilt: their fragment-order twins exist only while the option is on)
    if (value < 0 || value > 2) return bad_arg(h, "conv_f16: 0, 1 or 2 (2: regardless of the launch size)");
    h->conv_f16 = value; h->ccache_ver = 0; h->gcache_ver = 0; return 0;
  }
  if (!strcmp(name, "d2_fwd_slab")) { h->d2_fwd_slab = value ? 1 : 0; h->ccache_ver = 0; return 0; }
  if (!strcmp(name, "d2_gate_bits")) { h->d2_gate_bits = value ? 1 : 0; return 0; }
  if (!strcmp(name, "d2_slab")) { h->d2_slab = value ? 1 : 0; h->ccache_ver = 0; return 0; }
  if (!strcmp(name, "dense_wgrad_slices")) { h->dense_slices = value; return 0; }
  if (!strcmp(name, "keep_gates")) {
    h->keep_gates = value ? 1 : 0;
    h->gate_keep_B = 0;
    for (int l = 1; l <= 4 && h->keep_gates; ++l)
      if (!h->gate_keep[l]) RD_CHECK(h, hipMalloc(&h->gate_keep[l], (size_t)h->MB * h->dL[l] * h->dch[l] * sizeof(float)));
    return 0;
  }
  if (!strcmp(name, "split3")) { h->split3 = value ? 1 : 0; return 0; }
  if (!strcmp(name, "side_stream")) { h->side_on = (value && h->side) ? 1 : 0; return 0; }
  if (!strcmp(name, "edge_kernels")) { h->edge_kernels = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
  if (!strcmp(name, "sample_offset")) { if (value < 0) return bad_arg(h, "set_option: sample_offset < 0"); h->sample_offset = value; return 0; }
  if (!strcmp(name, "ws_ksplit")) { h->ws_ksplit = value < 0 ? 0 : (value > 8 ? 8 : value); return 0; }   // > 1 = force (tests)
  return bad_arg(h, "set_option: unknown option");
}

extern "C" int rdgan_set_weight_versions(rdgan_handle* h, uint64_t gen_version, uint64_t critic_version) {
  if (!h) return -2;
  h->gver_in = gen_version; h->cver_in = critic_version;
  return 0;
}
extern "C" int rdgan_form_builds(const rdgan_handle* h, long* gen_builds, long* critic_builds) {
  if (!h) return -2;
  if (gen_builds) *gen_builds = h->form_builds[0];
  if (critic_builds) *critic_builds = h->form_builds[1];
  return 0;
}

extern "C" int rdgan_profile(rdgan_handle* h, unsigned tag_mask) {
  if (!h) return -2;
  h->prof_mask = tag_mask;
  for (int t = 0; t < RDGAN_NUM_TAGS; ++t) {
    h->ev_used[t] = 0;
    if ((tag_mask >> t & 1u) && h->ev_start[t].empty()) {
      h->ev_start[t].resize(4096); h->ev_stop[t].resize(4096);
      for (size_t i = 0; i < 4096; ++i) {
        RD_CHECK(h, hipEventCreate(&h->ev_start[t][i]));
        RD_CHECK(h, hipEventCreate(&h->ev_stop[t][i]));
      }
    }
  }
  return 0;
}
static const char* const RD_PLAN_NAMES[PL_COUNT] = {
  "gen dense", "gen block1 fwd (direct)", "gen block2 fwd (direct)", "gen block3 fwd (direct)", "gen conv 64->1 fwd", "gen block1 dgrad (direct)",
  "gen block2 dgrad (direct)", "gen block3 dgrad (direct)", "gen conv 64->1 bwd",
  "critic layer1", "critic layer2", "critic layer3", "critic layer4", "critic layer2 dgrad", "critic layer3 dgrad", "critic layer4 dgrad",
  "critic layer1 dgrad (column GEMM)",
  "gen block1 (collapsed)", "gen block2 (collapsed)", "gen block3 (collapsed)", "gen block1 dgrad (collapsed)", "gen block2 dgrad (collapsed)",
  "gen block3 dgrad (collapsed)",
  "gen block1 shared-centre E[s]", "gen block2 shared-centre E[s]", "gen block3 shared-centre E[s]",
  "gen block1 shared-centre S", "gen block2 shared-centre S", "gen block3 shared-centre S",
  "gen block1 shared-centre E[s+1]", "gen block2 shared-centre E[s+1]", "gen block3 shared-centre E[s+1]",
  "gen block1 dgrad shared part", "gen block2 dgrad shared part", "gen block3 dgrad shared part",
  "gen block1 dgrad difference part", "gen block2 dgrad difference part", "gen block3 dgrad difference part",
  "gen block1 fwd difference part", "gen block2 fwd difference part", "gen block3 fwd difference part",
  "critic layer2 (border boxes)", "critic layer3 (border boxes)", "critic layer4 (border boxes)",
  "critic layer2 dgrad (border boxes)", "critic layer3 dgrad (border boxes)", "critic layer4 dgrad (border boxes)",
  "gen block1 shared-centre E[s] (boxes)", "gen block2 shared-centre E[s] (boxes)", "gen block3 shared-centre E[s] (boxes)",
  "gen block1 shared-centre S (boxes)", "gen block2 shared-centre S (boxes)", "gen block3 shared-centre S (boxes)",
  "gen block1 shared-centre E[s+1] (boxes)", "gen block2 shared-centre E[s+1] (boxes)", "gen block3 shared-centre E[s+1] (boxes)",
  "gen block1 (collapsed, boxes)", "gen block2 (collapsed, boxes)", "gen block3 (collapsed, boxes)",
  "gen block1 dgrad (collapsed, boxes)", "gen block2 dgrad (collapsed, boxes)", "gen block3 dgrad (collapsed, boxes)",
  "gen dense (bf16 pipe)"};

extern "C" int rdgan_profile_launches(rdgan_handle* h, int on) {
  if (!h) return -2;
  if (on && h->launch_recs.empty()) {
    h->launch_recs.resize(8192);
    for (auto& r : h->launch_recs) { RD_CHECK(h, hipEventCreate(&r.e0)); RD_CHECK(h, hipEventCreate(&r.e1)); }
  }
  h->prof_launches = on ? 1 : 0;
  if (on) h->launch_used = 0;
  return 0;
}
extern "C" int rdgan_launch_table(rdgan_handle* h, rdgan_launch_stat* out, int cap, int* n_out) {
  if (!h || !out || cap < 1 || !n_out) return -2;
  RD_CHECK(h, hipDeviceSynchronize());
  int n = 0;
  for (size_t i = 0; i < h->launch_used; ++i) {
    const RdLaunchRec& r = h->launch_recs[i];
    float ms = 0;
    RD_CHECK(h, hipEventElapsedTime(&ms, r.e0, r.e1));
    int k = 0;
    for (; k < n; ++k)
      if (out[k].plan == r.plan && out[k].kind == r.kind && out[k].batch == r.batch && !strcmp(out[k].kernel, r.kernel)) break;
    if (k == n) {
      if (n == cap) continue;
      memset(&out[n], 0, sizeof(out[n]));
      out[n].plan = r.plan; out[n].kind = r.kind; out[n].batch = r.batch;
      memcpy(out[n].kernel, r.kernel, sizeof(r.kernel));
      snprintf(out[n].name, sizeof(out[n].name), "%s", r.plan >= 0 && r.plan < PL_COUNT ? RD_PLAN_NAMES[r.plan] : "op");
      ++n;
    }
    out[k].launches += 1; out[k].gflop += r.flops * 1e-9; out[k].ms += ms;
  }
  *n_out = n;
  return 0;
}

extern "C" int rdgan_flop_count(rdgan_handle* h, double* flops, int reset) {
  if (!h) return -2;
  if (flops) *flops = h->flops_acc;
  if (reset) h->flops_acc = 0;
  return 0;
}
extern "C" int rdgan_profile_read(rdgan_handle* h, int tag, double* total_ms, long* launches) {
  if (!h || tag < 0 || tag >= RDGAN_NUM_TAGS) return -2;
  RD_CHECK(h, hipDeviceSynchronize());
  double tot = 0;
  for (size_t i = 0; i < h->ev_used[tag]; ++i) {
    float ms = 0;
    RD_CHECK(h, hipEventElapsedTime(&ms, h->ev_start[tag][i], h->ev_stop[tag][i]));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (long)h->ev_used[tag];
  return 0;
}

// ------------------------------------------------------------------------------------
// generator forward (T:312-357)
// ------------------------------------------------------------------------------------
static RdEpi epi_make(int mode, const float* bias = nullptr, const float* aux = nullptr, int use_drop = 0,
                      uint32_t key = 0, uint32_t idx_base = 0) {
  RdEpi e; e.mode = mode; e.use_drop = use_drop; e.key = key; e.idx_base = idx_base; e.bias = bias; e.aux = aux;
  e.ksplit = 1; e.kpart = nullptr; e.kstride = 0; e.rinv = nullptr; e.gw = e.ghw = e.gq = 0;
  e.addt = nullptr; e.addt_plane = 0; e.nametag = 0; e.out16 = 0;
  return e;
}
// pointer arithmetic on an activation tensor of the workspace: bf16 elements in the storage mode, floats otherwise
static inline float* act_off(const rdgan_handle* h, float* p, long elems) {
  return (float*)((char*)p + elems * (h->a16 ? 2 : 4));
}
// bf16 storage mode needs the forms whose GEMMs all exist as bf16 kernels
static int a16_check(rdgan_handle* h) {
  if (!h->a16) return 0;
  if (!h->collapse || !h->g9_direct)
    return bad_arg(h, "bf16 storage mode needs the options collapse and g9_direct at 1");
  if (4 * (size_t)(h->nd + 2) * (h->nd + 2) * sizeof(float) > 96 * 1024) return bad_arg(h, "bf16 storage mode: ndomain too large");
  return 0;
}

// the shared-centre form pays where the hour axis of the block input is long enough for its (D+1)/D boundary plane
// "fast_fwd" / "fast_bwd" at their default (-1) follow the storage mode: the shared-centre form saves a quarter of the tap
// products, which pays where the matrix pipe is the bound (fp32: 64 FLOP/clk/SIMD); with bf16 operands the pipe is 16x faster
// and nowhere near busy, so the plain collapsed form wins there -- one GEMM per block instead of two, no hour-difference /
// plane-sum / recombination passes (measured at bs 256: 3.89 vs 4.09 ms per iteration)
static int fast_fwd_on(const rdgan_handle* h) { return h->fast_fwd < 0 ? !h->a16 : h->fast_fwd; }
static int fast_bwd_on(const rdgan_handle* h) { return h->fast_bwd < 0 ? !h->a16 : h->fast_bwd; }
static bool gen_block_fast(const rdgan_handle* h, int l, int enabled) {
  return h->collapse && enabled && h->gdim[l - 1][0] >= 6;
}

// generator block 3 forward by the slab kernel (rdgan_upconv16.hip.h): bf16 storage mode, collapsed form, the 12 x 8 x 8 x 128
// -> 24 x 16 x 16 x 64 block of ndomain 16
static bool upconv_slab_on(const rdgan_handle* h, int l) {
  return h->upconv_slab && h->a16 && h->collapse && l == 3 && h->nd == 16 && !gen_block_fast(h, 3, fast_fwd_on(h));
}
// the same block on (h, w) tiles of 8 x 8 source positions with their halo (rdgan_upconv16t.hip.h): source planes of 16 x 16, 32 x 32, ...
static bool upconv_slab_t_on(const rdgan_handle* h, int l) {
  return h->upconv_slab_t && h->a16 && h->collapse && l == 3 && h->nd > 16 && h->gdim[2][1] % 8 == 0 && h->gdim[2][2] % 8 == 0 &&
         h->bW3T && !gen_block_fast(h, 3, fast_fwd_on(h));
}
static bool upconv2_slab_on(const rdgan_handle* h, int l) {
  return h->upconv2_slab && h->a16 && h->collapse && l == 2 && h->nd == 16 && !gen_block_fast(h, 2, fast_fwd_on(h));
}

// `ws`: stream of the weight-only kernels (the handle's side stream, forked by the caller, or `st` itself): the weight forms of
// block l are complete behind event ev_g[l], which `st` waits for in front of the block's GEMM
// plan of the forward / second-sweep GEMM of critic layer l >= 2 over n samples: the border-class boxes where they pay -- with
// few rows the boxes' short tap lists lose the K splits that fill the chip (layer 3 at 256 samples: 60 -> 68 us), so a small
// launch keeps the one-phase plan unless the boxes drop more than half of the work (layer 4: 70 %)
static int critic_box_plan(const rdgan_handle* h, int one, int box, long rows) {
  if (!h->border_boxes || !h->plans[box].boxes) return one;
  if (std::max(h->plans[box].src_sample, h->plans[box].dst_sample) * 4 * 258 >= 0x7FFFFFF0L) return one;      // (32-bit offsets of a tile)
  if (h->border_boxes >= 2 || rows >= 8192 || 2.0 * plan_flops(h->plans[box], 1) <= plan_flops(h->plans[one], 1)) return box;
  return one;
}
static int critic_fwd_plan(const rdgan_handle* h, int l, int n) {
  return critic_box_plan(h, PL_D2F + l - 2, PL_D2FX + l - 2, (long)n * h->dL[l]);
}
// weight gradient of layer l >= 2 (streaming kernels): every box is its own set of partial slabs and short-K workgroups, so the
// boxes must drop at least 30 % of the work to pay (ndomain 64, layer 3: 18 % dropped, 0.130 -> 0.142 ms)
static int critic_wgrad_plan(const rdgan_handle* h, int l, int n) {
  const int one = PL_D2F + l - 2, box = PL_D2FX + l - 2;
  if (!h->wgrad_boxes || !wgrad_box_ok(h->plans[box])) return one;
  if (h->border_boxes < 2 && plan_flops(h->plans[box], 1) > 0.7 * plan_flops(h->plans[one], 1)) return one;
  return critic_fwd_plan(h, l, n);
}
// the same choice for the input gradient of layer l >= 2 (rows = the layer's input positions)
static int critic_dgrad_plan(const rdgan_handle* h, int l, int n) {
  return critic_box_plan(h, PL_D2B + l - 2, PL_D2BX + l - 2, (long)n * h->dL[l - 1]);
}

// collapsed generator block (forward or input gradient): the border-class boxes where the launch runs several rounds of workgroups
// -- in a single round the interior box's workgroups, which keep every tap, set the time -- and the boxes drop 30 % of the work
// (block 1, 3 x 2 x 2 source grid: 47 % of its row-tap products are products with a zero row; blocks 2 / 3: 30 % / 16 %)
static bool box_span_ok(const RdPlan& p, int B) {      // a tile of the smallest box must stay inside the kernels' 32-bit offsets
  long minL = p.ph[0].L;
  for (int i = 1; i < p.nphases; ++i) minL = std::min<long>(minL, p.ph[i].L);
  return std::min<long>(B, 256 / minL + 2) * std::max(p.src_sample, p.dst_sample) * 4 < 0x7FFFFFF0L;
}
static int gen_box_plan(const rdgan_handle* h, int one, int box, int B) {
  if (!h->border_boxes || !h->plans[box].boxes || !box_span_ok(h->plans[box], B)) return one;
  if (h->border_boxes >= 2) return box;
  if (plan_flops(h->plans[box], 1) > 0.7 * plan_flops(h->plans[one], 1)) return one;
  return plan_tiles(h->plans[one], B, 128) * (h->plans[one].N / 128 > 0 ? h->plans[one].N / 128 : 1) >= 1024 ? box : one;
}

// the Dense layer by k_dense16_skinny (rdgan_edge.hip.h): the weight image is then in fragment order, so the choice is made per
// HANDLE (every call has B <= max_batch <= 128), not per call
static int dense_skinny_ks(const rdgan_handle* h) { return (h->n_in + 127) / 128 * 8; }      // k-steps of 16, a multiple of 8 (two waves x sets of four)
static bool dense_skinny_on(const rdgan_handle* h) {
  return h->a16 && h->dense16 && h->dense_skinny && h->MB <= 128 && h->bW0 && h->xcat16 && h->n_nodes % 32 == 0 && h->KP0 % 64 == 0;
}
static bool g9_fused_on(const rdgan_handle* h) { return h->g9_fused && h->tapgather && upconv_slab_on(h, 3); }
// the same inside the TILED block-3 kernel (ndomain 32, 64: k_upconv_slab_t16<G9>, k_tapsum_softmax12t)
static bool g9_fused_t_on(const rdgan_handle* h) { return h->g9_fused && h->tapgather && upconv_slab_t_on(h, 3); }

// keep_h3: block 3's output and 1/l2 are needed afterwards (generator step: its backward; rdgan_gen_forward: the test hook) --
// a critic step passes false, and with the fused last conv the 1.6 GB tensor (2048 samples) is then never written
static int gen_forward_impl(rdgan_handle* h, const float* gp, const float* z, const float* cond, float* out, int B,
                            hipStream_t st, hipStream_t ws, bool keep_h3 = true) {
  const int nd = h->nd;
  const bool a16 = h->a16 != 0;
  RD_TRY(a16_check(h));
  // ---- weight forms (read only the weights): skipped when the caller vouches that the forms in the workspace were built from
  // these very weights (same slab, same content version, same form options)
  const int gcfg = (h->collapse ? 1 : 0) | (fast_fwd_on(h) ? 2 : 0) | (a16 ? 4 : 0) | (h->upconv_slab ? 8 : 0) | (h->upconv2_slab ? 16 : 0) |
                   (h->g9_fused && h->tapgather ? 32 : 0) | (h->dense16 ? 64 : 0) | (h->upconv_slab_t ? 128 : 0) | (h->dense_skinny ? 256 : 0);
  const bool forms_cached = h->gver_in != 0 && gp == h->gcache_ptr && h->gver_in == h->gcache_ver && gcfg == h->gcache_cfg;
  if (!forms_cached) {
  h->form_builds[0]++;
  h->gcache_ptr = nullptr; h->gcache_ver = 0; h->gcache_cfg = -1;     // (valid again only once every form kernel has been issued)
  // W9T [64][32] = W9[tap][ci]^T (zero padded taps 27..31)
  RD_TRY(launch_transpose(h, gp + h->goff[8], h->W9T, 1, 27, 64, 32, ws));
  if (dense_skinny_on(h)) {
    const long nimg = (long)(h->n_nodes / 32) * dense_skinny_ks(h) * 16;      // (threads: four lanes' fragments each)
    hipLaunchKernelGGL(k_dense_wimg, dim3((unsigned)((nimg + 255) / 256)), dim3(256), 0, ws, gp + h->goff[0], (unsigned short*)h->bW0,
                       h->n_in, h->n_nodes, dense_skinny_ks(h));
  } else
  if (a16 && h->dense16 && h->dense16_ok && h->bW0)
    hipLaunchKernelGGL(k_dense_w16, dim3((h->n_nodes + 31) / 32, (h->KP0 + 31) / 32), dim3(256), 0, ws, gp + h->goff[0],
                       (unsigned short*)h->bW0, h->n_in, h->n_nodes, h->KP0);
  if (g9_fused_on(h) || g9_fused_t_on(h)) hipLaunchKernelGGL(k_g9_wimg, dim3(1), dim3(256), 0, ws, gp + h->goff[8], (unsigned short*)h->bW9I);
  for (int l = 1; l <= 3; ++l) {
    const float* Wl = gp + h->goff[2 * l];
    const long cc = (long)h->gch[l - 1] * h->gch[l];
    if (gen_block_fast(h, l, fast_fwd_on(h))) {
      RdWeightMap wm;
      fastd_weight_map(wm);
      ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, ws);
      hipLaunchKernelGGL(k_weight_transform, dim3(ew_blocks(48L * cc / 4)), dim3(256), 0, ws, Wl, h->fU[l], (int)cc, 48, wm);
      if (a16) RD_TRY(launch_weights_to_bf16_t(h, h->fU[l], h->bU[l], 48, h->gch[l - 1], h->gch[l], ws));
    } else if (h->collapse) {
      hipLaunchKernelGGL(k_collapse_weights, dim3(ew_blocks(16L * cc)), dim3(256), 0, ws, Wl, h->GWC[l], (int)cc);
      if (upconv_slab_on(h, l)) hipLaunchKernelGGL(k_upconv_wimg, dim3(256), dim3(256), 0, ws, h->GWC[l], (unsigned short*)h->bW3I);
      else if (upconv_slab_t_on(h, l)) hipLaunchKernelGGL(k_upconv_wimg_t, dim3(256), dim3(256), 0, ws, h->GWC[l], (unsigned short*)h->bW3T);
      else if (upconv2_slab_on(h, l)) hipLaunchKernelGGL(k_upconv2_wimg, dim3(RD_UP2_KSTEPS), dim3(256), 0, ws, h->GWC[l], (unsigned short*)h->bW2I);
      else if (a16) {
        RD_TRY(launch_weights_to_bf16_t(h, h->GWC[l], h->bG1F[l], 64, h->gch[l - 1], h->gch[l], ws, h->conv_f16 ? h->fG1F[l] : nullptr));
      }
    }
    if (ws != st) RD_CHECK(h, hipEventRecord(h->ev_g[l], ws));
  }
  RD_CHECK(h, hipGetLastError());
  h->gcache_ptr = gp; h->gcache_ver = h->gver_in; h->gcache_cfg = gcfg;
  }
  // ---- activations
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    const bool d16 = a16 && h->dense16 && h->dense16_ok && h->xcat16 && !dense_skinny_on(h);
    hipLaunchKernelGGL(k_concat, dim3(ew_blocks((long)B * h->n_in)), dim3(256), 0, st, z, cond, h->xcat, B,
                       RDGAN_LATENT_DIM, nd * nd * h->nc, h->d_flag,       // (first kernel of every entry: clears the non-finite flag)
                       d16 ? (unsigned short*)h->xcat16 : (unsigned short*)nullptr, h->KP0);
  }
  // Dense + LeakyReLU (T:326-327); the Reshape (T:328) is a view
  if (dense_skinny_on(h)) {
    // small batch: the kernel is the launch (415 MB of bf16 weights at ndomain 64) -- weights and input rows streamed in fragment order
    if (ws != st && !forms_cached) RD_CHECK(h, hipStreamWaitEvent(st, h->ev_g[1], 0));      // (the image is built in front of block 1's forms)
    const int RB = (B + 31) / 32, KS = dense_skinny_ks(h);
    ProfScope ps(h, -1, st);
    LaunchScope ls(h, PL_GDENSE16, RD_KIND_CONV, B, 3.0 * plan_flops(h->plans[PL_GDENSE16], B), st);
    RD_KNAME(h, "k_dense16_skinny<bf16>");
    h->flops_acc += 3.0 * plan_flops(h->plans[PL_GDENSE16], B);         // (the plan is a third of the columns)
    hipLaunchKernelGGL(k_concat16f, dim3((unsigned)(((long)KS * RB * 64 + 255) / 256)), dim3(256), 0, st, z, cond, (unsigned short*)h->xcat16, B,
                       RDGAN_LATENT_DIM, nd * nd * h->nc, KS, RB);
    const dim3 dg((unsigned)(h->n_nodes / 32));
    const unsigned short* xi = (const unsigned short*)h->xcat16; const unsigned short* wi = (const unsigned short*)h->bW0;
    const float* bs = gp + h->goff[1];
    if (RB == 1) hipLaunchKernelGGL(k_dense16_skinny<1>, dg, dim3(128), 0, st, xi, wi, bs, (rd_bf16_t*)h->h0, B, KS, h->n_nodes);
    else if (RB == 2) hipLaunchKernelGGL(k_dense16_skinny<2>, dg, dim3(128), 0, st, xi, wi, bs, (rd_bf16_t*)h->h0, B, KS, h->n_nodes);
    else if (RB == 3) hipLaunchKernelGGL(k_dense16_skinny<3>, dg, dim3(128), 0, st, xi, wi, bs, (rd_bf16_t*)h->h0, B, KS, h->n_nodes);
    else hipLaunchKernelGGL(k_dense16_skinny<4>, dg, dim3(128), 0, st, xi, wi, bs, (rd_bf16_t*)h->h0, B, KS, h->n_nodes);
    RD_CHECK(h, hipGetLastError());
  } else
  if (a16 && h->dense16 && h->dense16_ok && h->xcat16) {
    // the Dense layer on the bf16 matrix pipe: at ndomain 64 its 825 MB fp32 kernel is the launch (HBM-bound); the bf16 image halves it
    if (ws != st && !forms_cached) RD_CHECK(h, hipStreamWaitEvent(st, h->ev_g[1], 0));      // (the image is built in front of block 1's forms)
    const int n3 = h->n_nodes / 3;
    for (int j = 0; j < 3; ++j) {
      RdEpi ed = epi_make(RD_EPI_BIAS_LRELU, gp + h->goff[1] + (long)j * n3);
      ed.out16 = 1;
      const unsigned short* wj = (const unsigned short*)h->bW0 + (long)j * n3 * h->KP0;
      float* dj = (float*)((rd_bf16_t*)h->h0 + (long)j * n3);
      if (B <= 128) {       // one row tile: the 64-column tile doubles the workgroups that stream the kernel (ndomain 64: 138 MB per launch)
        ProfScope ps(h, -1, st);
        LaunchScope ls(h, PL_GDENSE16, RD_KIND_CONV, B, plan_flops(h->plans[PL_GDENSE16], B), st);
        h->flops_acc += plan_flops(h->plans[PL_GDENSE16], B);
        RD_TRY((launch_conv_ws_cfg<128, 64, 2, 2, true>(h, h->plans[PL_GDENSE16], h->d_plans + PL_GDENSE16, B, (const float*)h->xcat16,
                                                        (const float*)wj, 0, dj, ed, st)));
      } else
      RD_TRY(launch_conv16(h, h->plans[PL_GDENSE16], h->d_plans + PL_GDENSE16, B, h->xcat16, wj, dj, ed, st, -1));
    }
  } else
  if (a16) RD_TRY(launch_conv_a16(h, h->plans[PL_GDENSE], h->d_plans + PL_GDENSE, B, h->xcat, gp + h->goff[0], h->n_nodes, h->h0,
                                  epi_make(RD_EPI_BIAS_LRELU, gp + h->goff[1]), st, -1, false, true));
  else
  RD_TRY(launch_conv(h, h->plans[PL_GDENSE], h->d_plans + PL_GDENSE, B, h->xcat, gp + h->goff[0], h->n_nodes, h->h0,
                     epi_make(RD_EPI_BIAS_LRELU, gp + h->goff[1]), st, -1));
  float* hs[4] = {h->h0, h->h1, h->h2, h->h3};
  float* rs[4] = {nullptr, h->r1, h->r2, h->r3};
  for (int l = 1; l <= 3; ++l) {
    // UpSampling3D + Conv3D + bias (T:330-331), then PixelNorm + LeakyReLU (T:332-333)
    const float* Wl = gp + h->goff[2 * l];
    int pl = PL_G1F + l - 1;
    if (ws != st && !forms_cached) RD_CHECK(h, hipStreamWaitEvent(st, h->ev_g[l], 0));       // this block's weight forms
    if (gen_block_fast(h, l, fast_fwd_on(h))) {
      // shared-centre form along the hour axis: T = S x[s] once per output plane pair, then the difference part
      const int* sd = h->gdim[l - 1];
      const long P = (long)sd[1] * sd[2] * h->gch[l - 1];
      {
        ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
        const dim3 dg(ew_blocks((long)B * (sd[0] + 1) * P / 4));
        if (a16) hipLaunchKernelGGL(k_diff_d<rd_bf16_t>, dg, dim3(256), 0, st, (const rd_bf16_t*)hs[l - 1], (rd_bf16_t*)h->fE[l], B, sd[0], P);
        else hipLaunchKernelGGL(k_diff_d<float>, dg, dim3(256), 0, st, (const float*)hs[l - 1], h->fE[l], B, sd[0], P);
      }
      const int pls = PL_F1WS + l - 1, ple = PL_F1FE + l - 1;
      RdEpi et = epi_make(RD_EPI_PLAIN);
      et.out16 = a16;
      if (a16) RD_TRY(launch_conv16(h, h->plans[pls], h->d_plans + pls, B, hs[l - 1], h->bU[l], h->fgS, et, st, RDGAN_TAG_GCONV_FWD));
      else
      RD_TRY(launch_conv(h, h->plans[pls], h->d_plans + pls, B, hs[l - 1], h->fU[l], h->gch[l], h->fgS, et, st,
                         RDGAN_TAG_GCONV_FWD));
      const bool fuse = a16 ? conv16_rows_owned(h->plans[ple]) : conv_rows_owned(h->plans[ple], B);
      RdEpi ep = epi_make(fuse ? RD_EPI_BIAS_PN_LRELU : RD_EPI_BIAS, gp + h->goff[2 * l + 1]);
      ep.rinv = rs[l];
      ep.addt = h->fgS; ep.addt_plane = 4 * sd[1] * sd[2] * h->gch[l];
      ep.nametag = l == 3;
      ep.out16 = a16;
      if (a16) RD_TRY(launch_conv16(h, h->plans[ple], h->d_plans + ple, B, h->fE[l], h->bU[l], hs[l], ep, st,
                                    l == 3 ? RDGAN_TAG_GCONV3_FWD : RDGAN_TAG_GCONV_FWD));
      else
      RD_TRY(launch_conv(h, h->plans[ple], h->d_plans + ple, B, h->fE[l], h->fU[l], h->gch[l], hs[l], ep, st,
                         l == 3 ? RDGAN_TAG_GCONV3_FWD : RDGAN_TAG_GCONV_FWD));
      if (!fuse) {
        ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
        RD_TRY(launch_pn_fwd(h, hs[l], hs[l], rs[l], (long)B * h->gpix[l], h->gch[l], st, a16));
      }
      continue;
    }
    if (h->collapse) {
      Wl = h->GWC[l];
      pl = PL_G1FC + l - 1;
      if (!upconv_slab_on(h, l) && !upconv2_slab_on(h, l) && !upconv_slab_t_on(h, l)) pl = gen_box_plan(h, pl, PL_G1FCX + l - 1, B);
    }
    const bool fuse = a16 ? conv16_rows_owned(h->plans[pl]) : conv_rows_owned(h->plans[pl], B);       // PixelNorm+LeakyReLU in the GEMM epilogue
    RdEpi ep = epi_make(fuse ? RD_EPI_BIAS_PN_LRELU : RD_EPI_BIAS, gp + h->goff[2 * l + 1]);
    ep.rinv = rs[l];
    ep.out16 = a16;
    ep.nametag = a16 && l == 3;
    if (upconv_slab_on(h, l)) {     // block 3, bf16 storage: source slab resident in LDS, weights streamed in fragment order
      ProfScope ps(h, RDGAN_TAG_GCONV3_FWD, st);
      // (with the fused last conv the launch also carries that layer's 2 * rows * 64 * 27 FLOPs)
      const double fl3 = plan_flops(h->plans[pl], B) + (g9_fused_on(h) ? 2.0 * B * h->gpix[3] * 64 * 27 : 0.0);
      LaunchScope ls(h, pl, RD_KIND_CONV, B, fl3, st);
      RD_KNAME(h, g9_fused_on(h) ? "k_upconv_slab16<bf16, +conv 64->1>" : "k_upconv_slab16<bf16>");
      h->flops_acc += fl3;
      const dim3 ug((unsigned)std::min(6 * B, 512));
      if (g9_fused_on(h)) {         // + the last conv's tap products (Q12 in P9) from the rows while they are in registers
        float* nodbg = nullptr;
        if (keep_h3) {
          RD_TRY(ensure_lds(h, (const void*)k_upconv_slab16<1, true, true>, RD_UPC_LDS_G9));
          hipLaunchKernelGGL((k_upconv_slab16<1, true, true>), ug, dim3(256), RD_UPC_LDS_G9, st, (const rd_bf16_t*)hs[l - 1],
                             (const rd_bf16_t*)h->bW3I, gp + h->goff[2 * l + 1], (rd_bf16_t*)hs[l], rs[l], B, nodbg,
                             (const unsigned short*)h->bW9I, h->P9);
        } else {
          RD_TRY(ensure_lds(h, (const void*)k_upconv_slab16<1, true, false>, RD_UPC_LDS_G9));
          hipLaunchKernelGGL((k_upconv_slab16<1, true, false>), ug, dim3(256), RD_UPC_LDS_G9, st, (const rd_bf16_t*)hs[l - 1],
                             (const rd_bf16_t*)h->bW3I, gp + h->goff[2 * l + 1], (rd_bf16_t*)hs[l], rs[l], B, nodbg,
                             (const unsigned short*)h->bW9I, h->P9);
        }
        RD_CHECK(h, hipGetLastError());
        continue;
      }
      RD_TRY(ensure_lds(h, (const void*)k_upconv_slab16<1>, RD_UPC_LDS));
      hipLaunchKernelGGL(k_upconv_slab16<1>, ug, dim3(256), RD_UPC_LDS, st, (const rd_bf16_t*)hs[l - 1],
                         (const rd_bf16_t*)h->bW3I, gp + h->goff[2 * l + 1], (rd_bf16_t*)hs[l], rs[l], B);
      RD_CHECK(h, hipGetLastError());
      continue;
    }
    if (upconv_slab_t_on(h, l)) {   // block 3, bf16 storage, planes larger than 8 x 8: (h, w) tiles with their halo resident, K in two halves
      ProfScope ps(h, RDGAN_TAG_GCONV3_FWD, st);
      const bool g9t = g9_fused_t_on(h);
      const double fl3 = plan_flops(h->plans[pl], B) + (g9t ? 2.0 * B * h->gpix[3] * 64 * 27 : 0.0);
      LaunchScope ls(h, pl, RD_KIND_CONV, B, fl3, st);
      RD_KNAME(h, g9t ? "k_upconv_slab_t16<bf16, +conv 64->1>" : "k_upconv_slab_t16<bf16>");
      h->flops_acc += fl3;
      const int Hs = h->gdim[2][1], Ws = h->gdim[2][2];
      const long items = (long)B * 6 * (Hs / 8) * (Ws / 8);
      const dim3 tg((unsigned)std::min<long>(items, 512));
      float* nodbg = nullptr;
      if (g9t && keep_h3) {
        RD_TRY(ensure_lds(h, (const void*)k_upconv_slab_t16<true, true>, RD_UPT_LDS_G9));
        hipLaunchKernelGGL((k_upconv_slab_t16<true, true>), tg, dim3(256), RD_UPT_LDS_G9, st, (const rd_bf16_t*)hs[l - 1], (const rd_bf16_t*)h->bW3T,
                           gp + h->goff[2 * l + 1], (rd_bf16_t*)hs[l], rs[l], B, Hs, Ws, nodbg, (const unsigned short*)h->bW9I, h->P9);
      } else if (g9t) {
        RD_TRY(ensure_lds(h, (const void*)k_upconv_slab_t16<true, false>, RD_UPT_LDS_G9));
        hipLaunchKernelGGL((k_upconv_slab_t16<true, false>), tg, dim3(256), RD_UPT_LDS_G9, st, (const rd_bf16_t*)hs[l - 1], (const rd_bf16_t*)h->bW3T,
                           gp + h->goff[2 * l + 1], (rd_bf16_t*)hs[l], rs[l], B, Hs, Ws, nodbg, (const unsigned short*)h->bW9I, h->P9);
      } else {
        RD_TRY(ensure_lds(h, (const void*)k_upconv_slab_t16<false, true>, RD_UPT_LDS));
        hipLaunchKernelGGL((k_upconv_slab_t16<false, true>), tg, dim3(256), RD_UPT_LDS, st, (const rd_bf16_t*)hs[l - 1], (const rd_bf16_t*)h->bW3T,
                           gp + h->goff[2 * l + 1], (rd_bf16_t*)hs[l], rs[l], B, Hs, Ws, nodbg, (const unsigned short*)nullptr, (float*)nullptr);
      }
      RD_CHECK(h, hipGetLastError());
      continue;
    }
    if (upconv2_slab_on(h, l)) {    // block 2, bf16 storage: a sample resident in LDS, the four waves split the 128 channels
      ProfScope ps(h, RDGAN_TAG_GCONV_FWD, st);
      LaunchScope ls(h, pl, RD_KIND_CONV, B, plan_flops(h->plans[pl], B), st);
      RD_KNAME(h, "k_upconv2_slab16<bf16>");
      h->flops_acc += plan_flops(h->plans[pl], B);
      RD_TRY(ensure_lds(h, (const void*)k_upconv2_slab16, RD_UP2_LDS));
      hipLaunchKernelGGL(k_upconv2_slab16, dim3((unsigned)std::min(B, 256 * RD_UP2_WGS)), dim3(256), RD_UP2_LDS, st, (const rd_bf16_t*)hs[l - 1],
                         (const rd_bf16_t*)h->bW2I, gp + h->goff[2 * l + 1], (rd_bf16_t*)hs[l], rs[l], B);
      RD_CHECK(h, hipGetLastError());
      continue;
    }
    if (a16) {     // collapsed form (64 taps) on the bf16 matrix pipe
      RD_TRY(launch_conv16(h, h->plans[pl], h->d_plans + pl, B, hs[l - 1], h->bG1F[l], hs[l], ep, st,
                           l == 3 ? RDGAN_TAG_GCONV3_FWD : RDGAN_TAG_GCONV_FWD, h->conv_f16 ? h->fG1F[l] : nullptr));
    } else
    RD_TRY(launch_conv(h, h->plans[pl], h->d_plans + pl, B, hs[l - 1], Wl, h->gch[l], hs[l], ep, st,
                       l == 3 ? RDGAN_TAG_GCONV3_FWD : RDGAN_TAG_GCONV_FWD));
    if (!fuse) {
      ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
      RD_TRY(launch_pn_fwd(h, hs[l], hs[l], rs[l], (long)B * h->gpix[l], h->gch[l], st, a16));
    }
  }
  // Conv3D 64->1 (T:345) as column GEMM + gather, bias, Softmax(axis=1) (T:347), check_numerics (T:349-350).
  // When a 256-row tile holds whole (h,w) planes (nd = 8, 16) or whole w rows (nd = 32, 64, 128) the GEMM's epilogue
  // sums the in-tile taps itself and writes 3 (9) floats per grid point instead of 32.
  const int gq = 256 % (nd * nd) == 0 ? 3 : (256 % nd == 0 ? 9 : 0);
  const long ncol = (long)B * nd * nd;
  if (g9_fused_on(h)) {
    // the tap products left the block-3 slab kernel as Q12: the sums over kd and the source classes, bias, softmax
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    hipLaunchKernelGGL((k_tapsum_softmax12<RDGAN_NHOURS>), dim3((unsigned)((ncol / 4 + 63) / 64)), dim3(256), 0, st, h->P9,
                       gp + h->goff[9], out, B, h->d_flag);
  } else if (g9_fused_t_on(h)) {
    // the same behind the tiled kernel: main sums per tile + the halo terms of the neighbouring tiles
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    {
      const int TH = h->gdim[2][1] / 8, TW = h->gdim[2][2] / 8;
      const long nthr = (long)B * 6 * TH * TW * 24 * 32;
      hipLaunchKernelGGL(k_g9_halo_fold, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, h->P9, B, TH, TW);
    }
    hipLaunchKernelGGL((k_tapsum_softmax12t<RDGAN_NHOURS>), dim3((unsigned)((ncol / 4 + 63) / 64)), dim3(256), 0, st, h->P9,
                       gp + h->goff[9], out, B, h->gdim[2][1], h->gdim[2][2], h->d_flag);
  } else if (!a16 && h->tapgather && h->edge_kernels == 1 && g9w_mfma_ok(nd, (long)B * h->gpix[3])) {
    // fp32 storage: pipelined streaming kernel on 128-pixel tiles (rdgan_edge.hip.h), nine kw-sums per grid point
    ProfScope ps(h, RDGAN_TAG_GCONV_FWD, st);
    const long rows9 = (long)B * h->gpix[3];
    LaunchScope ls(h, PL_G9F, RD_KIND_EDGE, B, 2.0 * rows9 * 64 * 27, st);
    RD_KNAME(h, "k_g9_fwd_mfma + k_tapsum_softmax");
    h->flops_acc += 2.0 * rows9 * 64 * 27;
    if (nd * nd <= 256) {       // whole (h,w) planes in a 256-pixel tile: three sums per grid point
      const size_t lds9 = (size_t)(256 * 64 + 256 * 33) * sizeof(float);
      RD_TRY(ensure_lds(h, (const void*)k_g9_fwd_mfma<256>, lds9));
      hipLaunchKernelGGL(k_g9_fwd_mfma<256>, dim3((unsigned)std::min<long>((rows9 + 255) / 256, 256)), dim3(512), lds9, st,
                         (const float*)h->h3, gp + h->goff[8], h->P9, rows9, nd, 2 * ilog2(nd));
      hipLaunchKernelGGL((k_tapsum_softmax<RDGAN_NHOURS, 3>), dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, st, h->P9,
                         gp + h->goff[9], out, B, nd, nd, h->d_flag);
    } else {
      const size_t lds9 = (size_t)(128 * 64 + 128 * 33) * sizeof(float);
      RD_TRY(ensure_lds(h, (const void*)k_g9_fwd_mfma<128>, lds9));
      hipLaunchKernelGGL(k_g9_fwd_mfma<128>, dim3((unsigned)std::min<long>((rows9 + 127) / 128, 768)), dim3(256), lds9, st,
                         (const float*)h->h3, gp + h->goff[8], h->P9, rows9, nd, 2 * ilog2(nd));
      hipLaunchKernelGGL((k_tapsum_softmax<RDGAN_NHOURS, 9>), dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, st, h->P9,
                         gp + h->goff[9], out, B, nd, nd, h->d_flag);
    }
  } else if (gq && h->tapgather && (h->edge_kernels >= 2 || (h->edge_kernels && a16))) {
    // dedicated streaming kernel (rdgan_edge.hip.h): same tiles, same arithmetic and output as the tap-gathering GEMM below.
    // Default in the bf16 storage mode (one pass over h3 at 5.4 TB/s: 37 us against 153 us at bs 256); with fp32 storage its
    // one-tile-per-workgroup form (64 KB tiles, two workgroups per CU in lockstep) only matches the GEMM (159 vs 147 us),
    // so fp32 keeps the GEMM unless "edge_kernels" is 2 (tests).
    ProfScope ps(h, RDGAN_TAG_GCONV_FWD, st);
    const long rows9 = (long)B * h->gpix[3];
    LaunchScope ls(h, PL_G9F, RD_KIND_EDGE, B, 2.0 * rows9 * 64 * 27, st);
    RD_KNAME(h, "k_g9_fwd + k_tapsum_softmax");
    h->flops_acc += 2.0 * rows9 * 64 * 27;
    RD_TRY(ensure_lds(h, a16 ? (const void*)k_g9_fwd<rd_bf16_t> : (const void*)k_g9_fwd<float>, a16 ? 36 * 1024 : 72 * 1024));
    const dim3 g9((unsigned)((rows9 + 255) / 256));
    if (a16) hipLaunchKernelGGL(k_g9_fwd<rd_bf16_t>, g9, dim3(256), 36 * 1024, st, (const rd_bf16_t*)h->h3, gp + h->goff[8], h->P9, rows9, nd,
                                nd * nd, gq);
    else hipLaunchKernelGGL(k_g9_fwd<float>, g9, dim3(256), 72 * 1024, st, (const float*)h->h3, gp + h->goff[8], h->P9, rows9, nd,
                            nd * nd, gq);
    if (gq == 3) hipLaunchKernelGGL((k_tapsum_softmax<RDGAN_NHOURS, 3>), dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, st, h->P9,
                                    gp + h->goff[9], out, B, nd, nd, h->d_flag);
    else hipLaunchKernelGGL((k_tapsum_softmax<RDGAN_NHOURS, 9>), dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, st, h->P9,
                            gp + h->goff[9], out, B, nd, nd, h->d_flag);
  } else if (gq && h->tapgather) {
    RdEpi e = epi_make(RD_EPI_TAPGATHER);
    e.gw = nd; e.ghw = nd * nd; e.gq = gq;
    if (a16) RD_TRY(launch_conv_a16(h, h->plans[PL_G9F], h->d_plans + PL_G9F, B, h->h3, h->W9T, 32, h->P9, e, st, -1, true, false));
    else
    RD_TRY(launch_conv(h, h->plans[PL_G9F], h->d_plans + PL_G9F, B, h->h3, h->W9T, 32, h->P9, e, st, -1));
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    if (gq == 3) hipLaunchKernelGGL((k_tapsum_softmax<RDGAN_NHOURS, 3>), dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, st, h->P9,
                                    gp + h->goff[9], out, B, nd, nd, h->d_flag);
    else hipLaunchKernelGGL((k_tapsum_softmax<RDGAN_NHOURS, 9>), dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, st, h->P9,
                            gp + h->goff[9], out, B, nd, nd, h->d_flag);
  } else {
    if (a16) RD_TRY(launch_conv_a16(h, h->plans[PL_G9F], h->d_plans + PL_G9F, B, h->h3, h->W9T, 32, h->P9, epi_make(RD_EPI_PLAIN), st, -1,
                                    true, false));
    else
    RD_TRY(launch_conv(h, h->plans[PL_G9F], h->d_plans + PL_G9F, B, h->h3, h->W9T, 32, h->P9, epi_make(RD_EPI_PLAIN), st, -1));
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    hipLaunchKernelGGL(k_colgather_softmax, dim3((unsigned)((ncol + 63) / 64)), dim3(64), 0, st, h->P9, gp + h->goff[9],
                       out, B, RDGAN_NHOURS, nd, nd, h->d_flag);
  }
  RD_CHECK(h, hipGetLastError());
  return 0;
}

extern "C" int rdgan_gen_forward(rdgan_handle* h, const float* gen_params, const float* z, const float* cond,
                                 float* out, int B, void* stream) {
  if (!h || !gen_params || !z || !cond || !out) return bad_arg(h, "gen_forward: null pointer");
  if (B < 1 || B > h->MB) return bad_arg(h, "gen_forward: B outside [1, max_batch]");
  return gen_forward_impl(h, gen_params, z, cond, out, B, (hipStream_t)stream, side_fork(h, (hipStream_t)stream));
}

// tf.debugging.check_numerics behind the generator's softmax (T:349-350): the softmax kernels raise a device flag on
// NaN/Inf; this waits for `stream` and reports the flag of the calls issued since the last reset (every public
// generator-forward / gradient entry resets it first).
extern "C" int rdgan_check_numerics(rdgan_handle* h, void* stream) {
  if (!h) return -2;
  RD_CHECK(h, hipStreamSynchronize((hipStream_t)stream));
  int flag = 0;
  RD_CHECK(h, hipMemcpy(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) { h->err = "check_numerics: the generator output contains NaN or Inf"; return -1; }
  return 0;
}

// ------------------------------------------------------------------------------------
// critic
// ------------------------------------------------------------------------------------
static bool d2_slab_on(const rdgan_handle* h) { return h->d2_slab && h->a16 && h->nd == 16; }
// the same on tiles of 8 x 8 destination positions (k_d2_dgrad_slab_t16): ndomain 32, 48, 64, ... (layer 2's output grid a multiple of 4 x 4)
static bool d2_slab_t_on(const rdgan_handle* h) {
  return h->d2_slab && h->a16 && h->nd > 16 && h->nd % 16 == 0 && h->ddim[2][1] % 4 == 0 && h->ddim[2][2] % 4 == 0 && h->ddim[2][0] == 6 &&
         h->ddim[1][1] == 2 * h->ddim[2][1] - 1 && h->ddim[1][2] == 2 * h->ddim[2][2] - 1 && h->dpad[1][0] == 1 && h->dpad[1][1] == 1 && h->dpad[1][2] == 1;
}
static bool d2_fwd_slab_on(const rdgan_handle* h) { return h->d2_fwd_slab && h->a16 && h->nd == 16; }
static int prep_critic_weights(rdgan_handle* h, const float* dp, hipStream_t st) {
  // (skipped when the forms in the workspace were built from these very weights: see rdgan_set_weight_versions)
  const int ccfg = (h->a16 ? 1 : 0) | (h->d2_slab ? 2 : 0) | (h->d2_fwd_slab ? 4 : 0);
  if (h->cver_in != 0 && dp == h->ccache_ptr && h->cver_in == h->ccache_ver && ccfg == h->ccache_cfg) return 0;
  h->form_builds[1]++;
  h->ccache_ptr = nullptr; h->ccache_ver = 0; h->ccache_cfg = -1;     // (valid again only once every form kernel has been issued)
  if (!h->a16) {       // fp32 storage: transposed kernels of the input-gradient GEMMs (the bf16 storage mode reads bf16 images instead)
    for (int l = 2; l <= 4; ++l)   // [27][Cin][Cout] -> [27][Cout][Cin]
      RD_TRY(launch_transpose(h, dp + h->doff[2 * (l - 1)], h->DWT[l], 27, h->dch[l - 1], h->dch[l], h->dch[l - 1], st));
    // W1 [27*Cin][64] -> W1T [64][ldp1] (columns (tap,ci), zero padded)
    RD_TRY(launch_transpose(h, dp + h->doff[0], h->W1T, 1, 27 * h->Cin, 64, h->ldp1, st));
  }
  if (h->CP != h->Cin) hipLaunchKernelGGL(k_pad_w1, dim3(27), dim3(256), 0, st, dp + h->doff[0], h->W1P, h->Cin, h->CP);
  if (h->a16) {
    // layers 2-4 in ONE launch per image kind (forward [27][Cout][Cin], input gradient [27][Cin][Cout]): 3 launches per build
    // instead of 11 -- the critic's forms are rebuilt after each of its n_critic updates per iteration
    RdW3 a;
    for (int l = 2; l <= 4; ++l) {
      a.in[l - 2] = dp + h->doff[2 * (l - 1)];
      a.outT[l - 2] = (unsigned short*)h->bWF[l]; a.outC[l - 2] = (unsigned short*)h->bWB[l];
      a.outTf[l - 2] = h->conv_f16 ? (unsigned short*)h->fWF[l] : nullptr; a.outCf[l - 2] = h->conv_f16 ? (unsigned short*)h->fWB[l] : nullptr;
      a.K[l - 2] = h->dch[l - 1]; a.N[l - 2] = h->dch[l];
    }
    hipLaunchKernelGGL(k_weights3_to_bf16, dim3(8, 8, 3 * 27), dim3(256), 0, st, a);
    if (d2_fwd_slab_on(h))
      hipLaunchKernelGGL(k_d2f_wimg, dim3(RD_D2F_KSTEPS), dim3(256), 0, st, dp + h->doff[2], (unsigned short*)h->bW2F);
    if (d2_slab_on(h) || d2_slab_t_on(h))
      hipLaunchKernelGGL(k_d2s_wimg, dim3((RD_D2S_KSTEPS * 2 * 64 + 255) / 256), dim3(256), 0, st, dp + h->doff[2], (unsigned short*)h->bW2S);
    hipLaunchKernelGGL(k_w1_to_bf16, dim3(ew_blocks(64L * h->ldp1)), dim3(256), 0, st, dp + h->doff[0], (rd_bf16_t*)h->bW1B,
                       27 * h->Cin, h->ldp1);
  }
  RD_CHECK(h, hipGetLastError());
  h->ccache_ptr = dp; h->ccache_ver = h->cver_in; h->ccache_cfg = ccfg;
  return 0;
}

// D1's forward weights: the caller's [27][Cin][64] kernel, or its zero-padded [27][CP][64] copy when CP > Cin
static inline const float* d1_weights(const rdgan_handle* h, const float* dp) {
  return h->CP != h->Cin ? h->W1P : dp + h->doff[0];
}

// First critic layer as one K = 64 GEMM per tile (rdgan_edge.hip.h): one condition channel (2 floats per voxel), any ndomain
static bool d1_gemm_ok(const rdgan_handle* h) { return h->edge_kernels && h->CP == 2 && h->Cin == 2; }
// critic layer 2's weight gradient on tiles of 4 x 4 output positions (k_d2_wgrad_slab_t16): the geometry of 'valid' layer 1 +
// stride-2 'same' layer 2 on an ndomain that is a multiple of 16 (11 x (2 OH - 1)^2 -> 6 x OH^2, OH % 4 == 0)
static bool d2_wgrad_slab_t_on(const rdgan_handle* h) {
  return h->d2_wgrad_slab && h->a16 && h->nd > 16 && h->nd % 16 == 0 && h->ddim[1][0] == 11 && h->ddim[2][0] == 6 &&
         h->ddim[2][1] % 4 == 0 && h->ddim[2][2] % 4 == 0 && h->ddim[1][1] == 2 * h->ddim[2][1] - 1 && h->ddim[1][2] == 2 * h->ddim[2][2] - 1 &&
         h->dch[1] == 64 && h->dch[2] == 128;
}
static bool d2_gate_bits_on(const rdgan_handle* h) {
  return h->d2_gate_bits && h->d2_slab && h->a16 && (h->nd == 16 || d2_slab_t_on(h)) && h->g1bits;
}
static int launch_d1_fwd(rdgan_handle* h, const float* in, const float* w, const float* bias, float* out, const float* aux, int NBt,
                         int mode, int use_drop, uint32_t key, uint32_t idx_base, hipStream_t st) {
  unsigned char* gbits = mode == 0 && d2_gate_bits_on(h) ? h->g1bits : nullptr;
  ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
  const long rows = (long)NBt * h->dL[1];
  LaunchScope ls(h, PL_D1F, RD_KIND_EDGE, NBt, 2.0 * rows * 54 * 64, st);
  if (h->d1_fwd_sample && h->a16 && h->nd == 16 && h->CP == 2 && h->dL[1] == RD_D1S_NPOS && (mode == 0 || d2_gate_bits_on(h))) {
    // a sample's input volume resident in LDS (rdgan_d1fwd16.hip.h).  Mode 1 (second sweep, in place over the x_hat third) takes the
    // gate from the 2-bit codes the forward left for the rows at element offset idx_base
    RD_KNAME(h, "k_d1_fwd_sample16<bf16,%d>", mode);
    h->flops_acc += 2.0 * rows * 54 * 64;
    const dim3 grid((unsigned)std::min(NBt, 768));
    if (mode == 0) {
      RD_TRY(ensure_lds(h, (const void*)k_d1_fwd_sample16<0>, RD_D1S_LDS));
      hipLaunchKernelGGL(k_d1_fwd_sample16<0>, grid, dim3(256), RD_D1S_LDS, st, in, w, bias, (rd_bf16_t*)out, gbits, NBt, use_drop, key, idx_base);
    } else {
      RD_TRY(ensure_lds(h, (const void*)k_d1_fwd_sample16<1>, RD_D1S_LDS));
      hipLaunchKernelGGL(k_d1_fwd_sample16<1>, grid, dim3(256), RD_D1S_LDS, st, in, w, bias, (rd_bf16_t*)out,
                         h->g1bits + (size_t)(idx_base / 64) * 16, NBt, use_drop, key, idx_base);
    }
    RD_CHECK(h, hipGetLastError());
    return 0;
  }
  RD_KNAME(h, "k_d1_gemm_fwd<%s,%d>", h->a16 ? "bf16" : "f32", mode);
  h->flops_acc += 2.0 * rows * 54 * 64;
  constexpr size_t lds = (size_t)(128 * 64 + 64 * 64) * 4 + 2 * 128 * 8;
  const dim3 grid((unsigned)std::min<long>((rows + 127) / 128, 768));       // three workgroups per CU, each walks its tiles
  const int nd = h->nd, Do = h->ddim[1][0], Ho = h->ddim[1][1], Wo = h->ddim[1][2];
#define RD_D1F(TO, MODE)                                                                                                   \
  do {                                                                                                                     \
    RD_TRY(ensure_lds(h, (const void*)k_d1_gemm_fwd<TO, MODE>, lds));                                                      \
    hipLaunchKernelGGL((k_d1_gemm_fwd<TO, MODE>), grid, dim3(256), lds, st, in, w, bias, (TO*)out, (const TO*)aux, rows, nd, \
                       Do, Ho, Wo, use_drop, key, idx_base, gbits);                                                        \
  } while (0)
  if (h->a16) { if (mode == 0) RD_D1F(rd_bf16_t, 0); else RD_D1F(rd_bf16_t, 1); }
  else { if (mode == 0) RD_D1F(float, 0); else RD_D1F(float, 1); }
#undef RD_D1F
  RD_CHECK(h, hipGetLastError());
  return 0;
}
// bf16 storage mode: the layer-1 weight gradient on the bf16 matrix pipe, bias gradient (rows < bias_rows) in the same pass
static bool d1_wgrad16_on(const rdgan_handle* h) { return h->a16 && h->d1_wgrad16; }
static int launch_d1_wgrad(rdgan_handle* h, const float* in, const float* u1, float* dW, int NBt, hipStream_t st,
                           float* db = nullptr, long bias_rows = 0) {
  ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
  const long rows = (long)NBt * h->dL[1];
  if (d1_wgrad16_on(h)) {
    LaunchScope ls(h, PL_D1F, RD_KIND_WGRAD, NBt, 2.0 * rows * 54 * 64, st);
    RD_KNAME(h, "k_d1_wgrad16<bf16>");
    h->flops_acc += 2.0 * rows * 54 * 64;
    long G = std::min<long>(1024, (rows + 63) / 64);
    long rpw = ((rows + G - 1) / G + 63) / 64 * 64;
    G = (rows + rpw - 1) / rpw;
    if ((size_t)G * 4096 > h->wpartial_cap) return bad_arg(h, "d1 wgrad: partial workspace too small");
    hipLaunchKernelGGL(k_d1_wgrad16, dim3((unsigned)G), dim3(256), 0, st, in, (const rd_bf16_t*)u1, h->wpartial, rows, rpw,
                       db ? bias_rows : 0L, h->nd, h->ddim[1][0], h->ddim[1][1], h->ddim[1][2]);
    hipLaunchKernelGGL(k_d1_wgrad_fold, dim3((db ? 55 : 54) * 64 / 16), dim3(256), 0, st, h->wpartial, (int)G, dW, db);
    RD_CHECK(h, hipGetLastError());
    return 0;
  }
  LaunchScope ls(h, PL_D1F, RD_KIND_WGRAD, NBt, 2.0 * rows * 54 * 64, st);
  RD_KNAME(h, "k_d1_gemm_wgrad<%s>", h->a16 ? "bf16" : "f32");
  h->flops_acc += 2.0 * rows * 54 * 64;
  // one slice of rows per workgroup: four workgroups per CU, whole 32-row chunks
  long G = std::min<long>(1024, (rows + 31) / 32);
  long rpw = ((rows + G - 1) / G + 31) / 32 * 32;
  G = (rows + rpw - 1) / rpw;
  if ((size_t)G * 4096 > h->wpartial_cap) return bad_arg(h, "d1 wgrad: partial workspace too small");
  const int nd = h->nd, Do = h->ddim[1][0], Ho = h->ddim[1][1], Wo = h->ddim[1][2];
  if (h->a16) hipLaunchKernelGGL(k_d1_gemm_wgrad<rd_bf16_t>, dim3((unsigned)G), dim3(256), 0, st, in, (const rd_bf16_t*)u1, h->wpartial, rows,
                                 rpw, nd, Do, Ho, Wo);
  else hipLaunchKernelGGL(k_d1_gemm_wgrad<float>, dim3((unsigned)G), dim3(256), 0, st, in, u1, h->wpartial, rows, rpw, nd, Do, Ho, Wo);
  hipLaunchKernelGGL(k_d1_wgrad_fold, dim3(54 * 64 / 16), dim3(256), 0, st, h->wpartial, (int)G, dW);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// forward over NBt samples already laid out in h->cin; writes h->dh[1..4], h->v
static int critic_forward_impl(rdgan_handle* h, const float* dp, int NBt, uint64_t seed, hipStream_t st) {
  const int use_drop = seed != 0;
  const bool a16 = h->a16 != 0;
  const float* in = h->cin;
  for (int l = 1; l <= 4; ++l) {
    int pl = l == 1 ? PL_D1F : critic_fwd_plan(h, l, NBt);
    RdEpi ep = epi_make(RD_EPI_BIAS_LRELU_DROP, dp + h->doff[2 * (l - 1) + 1], nullptr, use_drop,
                        rd_make_key(seed, RD_STREAM_D1 + l - 1), 0);
    ep.out16 = a16;
    if (l == 1 && d1_gemm_ok(h))
      RD_TRY(launch_d1_fwd(h, in, dp + h->doff[0], dp + h->doff[1], h->dh[1], nullptr, NBt, 0, use_drop, ep.key, 0, st));
    else if (a16 && l == 1)
      RD_TRY(launch_conv_a16(h, h->plans[pl], h->d_plans + pl, NBt, in, d1_weights(h, dp), h->dch[l], h->dh[l], ep, st,
                             RDGAN_TAG_CRITIC_GEMM, false, true));
    else if (l == 2 && d2_fwd_slab_on(h)) {      // a sample's layer-1 output resident in LDS, the four waves split the 128 channels
      ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
      LaunchScope ls(h, pl, RD_KIND_CONV, NBt, plan_flops(h->plans[pl], NBt), st);
      RD_KNAME(h, "k_d2_fwd_slab16<bf16>");
      h->flops_acc += plan_flops(h->plans[pl], NBt);
      RD_TRY(ensure_lds(h, (const void*)k_d2_fwd_slab16, RD_D2F_LDS));
      hipLaunchKernelGGL(k_d2_fwd_slab16, dim3((unsigned)std::min(NBt, 512)), dim3(256), RD_D2F_LDS, st, (const rd_bf16_t*)in,
                         (const rd_bf16_t*)h->bW2F, dp + h->doff[3], (rd_bf16_t*)h->dh[2], NBt, use_drop, ep.key, 0u);
      RD_CHECK(h, hipGetLastError());
    } else if (a16)
      RD_TRY(launch_conv16(h, h->plans[pl], h->d_plans + pl, NBt, in, h->bWF[l], h->dh[l], ep, st, RDGAN_TAG_CRITIC_GEMM,
                           h->conv_f16 ? h->fWF[l] : nullptr));
    else
    RD_TRY(launch_conv(h, h->plans[pl], h->d_plans + pl, NBt, in, l == 1 ? d1_weights(h, dp) : dp + h->doff[2 * (l - 1)],
                       h->dch[l], h->dh[l], ep, st, RDGAN_TAG_CRITIC_GEMM));
    in = h->dh[l];
  }
  ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
  if (a16) hipLaunchKernelGGL(k_critic_dense_fwd<rd_bf16_t>, dim3(NBt), dim3(256), 0, st, (const rd_bf16_t*)h->dh[4], dp + h->doff[8],
                              dp + h->doff[9], h->v, h->F);
  else hipLaunchKernelGGL(k_critic_dense_fwd<float>, dim3(NBt), dim3(256), 0, st, (const float*)h->dh[4], dp + h->doff[8],
                          dp + h->doff[9], h->v, h->F);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// input-gradient chain u4 -> u1 over NBt samples (mode 0: critic step 3B batch, 1: generator step)
static int critic_dgrad_chain(rdgan_handle* h, const float* dp, int NBt, int B, int mode, uint64_t seed, hipStream_t st) {
  const int use_drop = seed != 0;
  const bool a16 = h->a16 != 0;
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    const dim3 g(ew_blocks((long)NBt * h->F));
    if (a16) hipLaunchKernelGGL(k_critic_top_bwd<rd_bf16_t>, g, dim3(256), 0, st, (const rd_bf16_t*)h->dh[4], dp + h->doff[8],
                                (rd_bf16_t*)h->du[4], NBt, h->F, B, mode, use_drop, rd_make_key(seed, RD_STREAM_D1 + 3));
    else hipLaunchKernelGGL(k_critic_top_bwd<float>, g, dim3(256), 0, st, (const float*)h->dh[4], dp + h->doff[8],
                            h->du[4], NBt, h->F, B, mode, use_drop, rd_make_key(seed, RD_STREAM_D1 + 3));
  }
  for (int l = 4; l >= 2; --l) {
    int pl = critic_dgrad_plan(h, l, NBt);
    RdEpi ep = epi_make(RD_EPI_GATE_AUX, nullptr, h->dh[l - 1], use_drop, rd_make_key(seed, RD_STREAM_D1 + l - 2), 0);
    ep.out16 = a16;
    if (l == 2 && d2_slab_on(h)) {      // two samples' output gradient resident in LDS, weights streamed in fragment order
      ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
      LaunchScope ls(h, pl, RD_KIND_CONV, NBt, plan_flops(h->plans[pl], NBt), st);
      RD_KNAME(h, "k_d2_dgrad_slab16<bf16>");
      h->flops_acc += plan_flops(h->plans[pl], NBt);
      RD_TRY(ensure_lds(h, (const void*)k_d2_dgrad_slab16, RD_D2S_LDS));
      hipLaunchKernelGGL(k_d2_dgrad_slab16, dim3((unsigned)std::min((NBt + 1) / 2, 512)), dim3(256), RD_D2S_LDS, st,
                         (const rd_bf16_t*)h->du[2], (const rd_bf16_t*)h->bW2S, (const rd_bf16_t*)h->dh[1], (rd_bf16_t*)h->du[1], NBt,
                         use_drop, d2_gate_bits_on(h) && d1_gemm_ok(h) ? h->g1bits : nullptr);
      RD_CHECK(h, hipGetLastError());
      continue;
    }
    if (l == 2 && d2_slab_t_on(h)) {    // the same on tiles: two samples' 6 x 5 x 5 output-gradient positions per tile resident
      ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
      const int one = PL_D2B;               // (FLOPs of the one-phase plan: the slab kernel multiplies every (position, tap) pair that lands inside)
      LaunchScope ls(h, one, RD_KIND_CONV, NBt, plan_flops(h->plans[one], NBt), st);
      RD_KNAME(h, "k_d2_dgrad_slab_t16<bf16>");
      h->flops_acc += plan_flops(h->plans[one], NBt);
      const int OH = h->ddim[2][1], OW = h->ddim[2][2];
      const long items = (long)((NBt + 1) / 2) * (OH / 4) * (OW / 4);
      RD_TRY(ensure_lds(h, (const void*)k_d2_dgrad_slab_t16, RD_D2T_LDS));
      hipLaunchKernelGGL(k_d2_dgrad_slab_t16, dim3((unsigned)std::min<long>(items, 512)), dim3(256), RD_D2T_LDS, st,
                         (const rd_bf16_t*)h->du[2], (const rd_bf16_t*)h->bW2S, (const rd_bf16_t*)h->dh[1], (rd_bf16_t*)h->du[1], NBt,
                         OH, OW, use_drop, d2_gate_bits_on(h) && d1_gemm_ok(h) ? h->g1bits : nullptr);
      RD_CHECK(h, hipGetLastError());
      continue;
    }
    if (a16)
      RD_TRY(launch_conv16(h, h->plans[pl], h->d_plans + pl, NBt, h->du[l], h->bWB[l], h->du[l - 1], ep, st, RDGAN_TAG_CRITIC_GEMM,
                           h->conv_f16 ? h->fWB[l] : nullptr));
    else
    RD_TRY(launch_conv(h, h->plans[pl], h->d_plans + pl, NBt, h->du[l], h->DWT[l], h->dch[l - 1], h->du[l - 1], ep, st,
                       RDGAN_TAG_CRITIC_GEMM));
  }
  return 0;
}

// dD/d(sample channel) for `B` samples whose u1 starts at u1: column GEMM + col2im -> h->g0
static int critic_input_grad(rdgan_handle* h, const float* dp, const float* u1, int B, hipStream_t st) {
  if (h->a16 && h->d1_dgrad_fused && h->nd == 16 && d1_gemm_ok(h)) {      // one pass, no column matrix (k_d1_dgrad_sample16)
    ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
    LaunchScope ls(h, PL_D1B, RD_KIND_CONV, B, 2.0 * B * h->dL[1] * 27 * 64, st);
    RD_KNAME(h, "k_d1_dgrad_sample16<bf16>");
    h->flops_acc += 2.0 * B * h->dL[1] * 27 * 64;
    RD_TRY(ensure_lds(h, (const void*)k_d1_dgrad_sample16, RD_D1DG_LDS));
    hipLaunchKernelGGL(k_d1_dgrad_sample16, dim3((unsigned)std::min(B, 512)), dim3(256), RD_D1DG_LDS, st, (const rd_bf16_t*)u1,
                       dp + h->doff[0], h->g0, B);
    RD_CHECK(h, hipGetLastError());
    return 0;
  }
  if (h->a16 && h->d1_dgrad_fused && h->nd > 16 && h->nd % 16 == 0 && d1_gemm_ok(h)) {      // the same on 24 x 16 x 8 input tiles (k_d1_dgrad_tile16)
    ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
    LaunchScope ls(h, PL_D1B, RD_KIND_CONV, B, 2.0 * B * h->dL[1] * 27 * 64, st);
    RD_KNAME(h, "k_d1_dgrad_tile16<bf16>");
    h->flops_acc += 2.0 * B * h->dL[1] * 27 * 64;
    RD_TRY(ensure_lds(h, (const void*)k_d1_dgrad_tile16, RD_D1DT_LDS));
    const long items = (long)B * (h->nd / 16) * (h->nd / 8);
    hipLaunchKernelGGL(k_d1_dgrad_tile16, dim3((unsigned)std::min<long>(items, 1024)), dim3(256), RD_D1DT_LDS, st, (const rd_bf16_t*)u1,
                       dp + h->doff[0], h->g0, B, h->nd);
    RD_CHECK(h, hipGetLastError());
    return 0;
  }
  if (h->a16)     // bf16 u1 against the bf16 first kernel [ldp1][64], fp32 column matrix
    RD_TRY(launch_conv16(h, h->plans[PL_D1B], h->d_plans + PL_D1B, B, u1, h->bW1B, h->P1, epi_make(RD_EPI_PLAIN), st,
                         RDGAN_TAG_CRITIC_GEMM));
  else
  RD_TRY(launch_conv(h, h->plans[PL_D1B], h->d_plans + PL_D1B, B, u1, h->W1T, h->ldp1, h->P1, epi_make(RD_EPI_PLAIN), st,
                     RDGAN_TAG_CRITIC_GEMM));
  ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
  hipLaunchKernelGGL(k_d1_col2im, dim3(ew_blocks((long)B * h->dL[0])), dim3(256), 0, st, h->P1, h->g0, B, h->ddim[0][0],
                     h->ddim[0][1], h->ddim[0][2], h->ddim[1][0], h->ddim[1][1], h->ddim[1][2], h->Cin, h->ldp1);
  RD_CHECK(h, hipGetLastError());
  return 0;
}

extern "C" int rdgan_critic_forward(rdgan_handle* h, const float* critic_params, const float* sample, const float* cond,
                                    float* out, int B, uint64_t seed, void* stream) {
  if (!h || !critic_params || !sample || !cond || !out) return bad_arg(h, "critic_forward: null pointer");
  if (B < 1 || B > h->NB) return bad_arg(h, "critic_forward: B outside [1, 3*max_batch]");
  hipStream_t st = (hipStream_t)stream;
  h->gate_keep_B = 0;
  RD_TRY(a16_check(h));
  if (h->a16) RD_TRY(prep_critic_weights(h, critic_params, st));      // (also makes the bf16 kernels of layers 2-4)
  else if (h->CP != h->Cin && !(h->cver_in != 0 && critic_params == h->ccache_ptr && h->cver_in == h->ccache_ver)) {
    // fp32 storage only needs the padded layer-1 kernel here.  W1P now belongs to THESE weights: whatever forms were cached
    // (a trainer's critic on the shared engine, models.get_engine) no longer describe the workspace
    hipLaunchKernelGGL(k_pad_w1, dim3(27), dim3(256), 0, st, critic_params + h->doff[0], h->W1P, h->Cin, h->CP);
    h->ccache_ptr = nullptr; h->ccache_ver = 0; h->ccache_cfg = -1;
  }
  launch_build_critic_input(h, sample, nullptr, cond, B, 2, 0u, 0u, st);
  RD_TRY(critic_forward_impl(h, critic_params, B, seed, st));
  RD_CHECK(h, hipMemcpyAsync(out, h->v, sizeof(float) * B, hipMemcpyDeviceToDevice, st));
  return 0;
}

extern "C" int rdgan_critic_grad(rdgan_handle* h, const float* dp, const float* gp, const float* x_real,
                                 const float* cond, const float* z, uint64_t seed, float* grad, int B, void* stream) {
  return rdgan_critic_grad_after(h, dp, gp, x_real, cond, z, seed, grad, B, nullptr, stream);
}

extern "C" int rdgan_critic_grad_after(rdgan_handle* h, const float* dp, const float* gp, const float* x_real,
                                       const float* cond, const float* z, uint64_t seed, float* grad, int B,
                                       void* critic_ready_event, void* stream) {
  if (!h || !dp || !gp || !x_real || !cond || !z || !grad) return bad_arg(h, "critic_grad: null pointer");
  if (B < 1 || B > h->MB) return bad_arg(h, "critic_grad: B outside [1, max_batch]");
  hipStream_t st = (hipStream_t)stream;
  const int NBt = 3 * B;
  const int use_drop = seed != 0;
  // fake = G(z, cond), generator frozen (T:363,370): reads no critic weight, so it is issued in front of the wait for
  // them -- the previous critic update's all-reduce + Adam (on the caller's other stream) hide behind it
  // The weight-only kernels (generator weight forms, then the critic's transposes / bf16 images behind the "critic ready"
  // event) go to the side stream beside the generator forward; the compute stream waits for them where it needs them.
  hipStream_t ws = side_fork(h, st);
  RD_TRY(gen_forward_impl(h, gp, z, cond, h->fake, B, st, ws, false));    // nothing differentiates through the generator here
  if (critic_ready_event) RD_CHECK(h, hipStreamWaitEvent(ws, (hipEvent_t)critic_ready_event, 0));
  RD_TRY(prep_critic_weights(h, dp, ws));
  RD_TRY(side_join(h, st, h->ev_cw));
  // [real; fake; alpha*real + (1-alpha)*fake] with the condition as 2nd channel (T:275-282, T:376)
  const bool a16 = h->a16 != 0;
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    launch_build_critic_input(h, x_real, h->fake, cond, B, 0, rd_make_key(seed, RD_STREAM_ALPHA), (uint32_t)h->sample_offset, st);
  }
  RD_TRY(critic_forward_impl(h, dp, NBt, seed, st));           // T:372,373,379 as one batch
  RD_TRY(critic_dgrad_chain(h, dp, NBt, B, 0, seed, st));       // dL/dh for real|fake, dD/dh for x_hat
  // bias gradients: only the real|fake passes reach the loss through the bias (the penalty term does not).  Column sums of
  // the output gradients the chain has just left: on the side stream, beside the penalty's second sweep and the weight-gradient
  // GEMMs below (nothing below writes du[l]; the column-sum scratch is used by these launches only)
  {
    hipStream_t cs = side_fork(h, st);
    for (int l = 1; l <= 4; ++l) {
      if (l == 1 && d1_gemm_ok(h) && d1_wgrad16_on(h)) continue;      // comes out of the weight-gradient GEMM (k_d1_wgrad16)
      RD_TRY(launch_colsum(h, h->du[l], (long)2 * B * h->dL[l], h->dch[l], grad + h->doff[2 * (l - 1) + 1], cs, h->a16 != 0));
    }
  }
  // gradient penalty (T:238-241, T:382): g0 = dD/dx_hat, n = ||g0||, r0 = d(10 mean((n-1)^2))/dg0
  RD_TRY(critic_input_grad(h, dp, act_off(h, h->du[1], (long)2 * B * h->dL[1] * 64), B, st));
  float* cin_hat = h->cin + (long)2 * B * h->dL[0] * h->CP;
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    // S blocks per sample when there are few samples of many elements (ndomain 64)
    const int per = (int)h->dL[0];
    const int S = std::max(1, std::min({(1024 + B - 1) / B, per / 4096, 64}));
    if (S > 1) hipLaunchKernelGGL(k_gp_norm_part, dim3(B * S), dim3(256), 0, st, h->g0, h->gp_part, per, S);
    hipLaunchKernelGGL(k_gp_norm_r0, dim3(B * S), dim3(256), 0, st, h->g0, cin_hat, h->gpv, per, B, RD_GP_WEIGHT, h->CP, S, h->gp_part);
  }
  // second forward sweep of the double backward: r_l = gate_l * conv_l(r_{l-1}), in place over the x_hat third
  h->gate_keep_B = 0;
  if (h->keep_gates) {       // test hook: keep the x_hat third of every h_l (see rdgan_debug_activation)
    for (int l = 1; l <= 4; ++l) {
      const long third = (long)2 * B * h->dL[l] * h->dch[l];
      RD_CHECK(h, hipMemcpyAsync(h->gate_keep[l], act_off(h, h->dh[l], third), (size_t)(third / 2) * (a16 ? 2 : 4),
                                 hipMemcpyDeviceToDevice, st));
    }
    h->gate_keep_B = B;
  }
  {
    const float* in = cin_hat;
    for (int l = 1; l <= 4; ++l) {
      int pl = l == 1 ? PL_D1F : critic_fwd_plan(h, l, B);
      long third = (long)2 * B * h->dL[l] * h->dch[l];
      float* dst = act_off(h, h->dh[l], third);
      RdEpi ep = epi_make(RD_EPI_GATE_AUX, nullptr, dst, use_drop, rd_make_key(seed, RD_STREAM_D1 + l - 1), (uint32_t)third);
      ep.out16 = a16;
      if (l == 1 && d1_gemm_ok(h))
        RD_TRY(launch_d1_fwd(h, in, dp + h->doff[0], nullptr, dst, dst, B, 1, use_drop, ep.key, (uint32_t)third, st));
      else if (a16 && l == 1)
        RD_TRY(launch_conv_a16(h, h->plans[pl], h->d_plans + pl, B, in, d1_weights(h, dp), h->dch[l], dst, ep, st,
                               RDGAN_TAG_CRITIC_GEMM, false, true));
      else if (a16)
        RD_TRY(launch_conv16(h, h->plans[pl], h->d_plans + pl, B, in, h->bWF[l], dst, ep, st, RDGAN_TAG_CRITIC_GEMM,
                             h->conv_f16 ? h->fWF[l] : nullptr));
      else
      RD_TRY(launch_conv(h, h->plans[pl], h->d_plans + pl, B, in, l == 1 ? d1_weights(h, dp) : dp + h->doff[2 * (l - 1)],
                         h->dch[l], dst, ep, st, RDGAN_TAG_CRITIC_GEMM));
      in = dst;
    }
  }
  // weight gradients over the 3B batch: inputs [h_real; h_fake; r_hat], output grads [u_real; u_fake; u_hat]
  for (int l = 1; l <= 4; ++l) {
    int pl = l == 1 ? PL_D1F : PL_D2F + l - 2;
    const float* in = l == 1 ? h->cin : h->dh[l - 1];
    const bool padded = l == 1 && h->CP != h->Cin;   // D1 with padding channels: gradient of the padded kernel, then drop the pad rows
    if (l == 1 && d1_gemm_ok(h)) {
      RD_TRY(launch_d1_wgrad(h, in, h->du[1], grad + h->doff[0], NBt, st, grad + h->doff[1], (long)2 * B * h->dL[1]));
    } else if (a16 && l == 2 && h->d2_wgrad_slab && h->nd == 16) {
      // a wave owns one tap: its [64 x 128] product stays in registers over the workgroup's share of the batch
      ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
      LaunchScope ls(h, pl, RD_KIND_WGRAD, NBt, plan_flops(h->plans[pl], NBt), st);
      RD_KNAME(h, "k_d2_wgrad_slab16<bf16>");
      h->flops_acc += plan_flops(h->plans[pl], NBt);
      const int G = NBt >= 64 ? 64 : 8;
      if ((size_t)G * 27 * RD_D2W_TILE > h->wpartial_cap) return bad_arg(h, "d2 wgrad: partial workspace too small");
      RD_TRY(ensure_lds(h, (const void*)k_d2_wgrad_slab16, RD_D2W_LDS));
      hipLaunchKernelGGL(k_d2_wgrad_slab16, dim3(4 * G), dim3(512), RD_D2W_LDS, st, (const rd_bf16_t*)in, (const rd_bf16_t*)h->du[2],
                         h->wpartial, NBt, G);
      hipLaunchKernelGGL(k_d2_wgrad_fold, dim3((27 * RD_D2W_TILE / 4 + 255) / 256), dim3(256), 0, st, h->wpartial, G, grad + h->doff[2]);
      RD_CHECK(h, hipGetLastError());
    } else if (a16 && l == 2 && d2_wgrad_slab_t_on(h)) {
      // the same on (h, w) tiles of 4 x 4 output positions (ndomain 32 / 48 / 64)
      ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
      LaunchScope ls(h, pl, RD_KIND_WGRAD, NBt, plan_flops(h->plans[pl], NBt), st);
      RD_KNAME(h, "k_d2_wgrad_slab_t16<bf16>");
      h->flops_acc += plan_flops(h->plans[pl], NBt);
      const RdD2wGeom geo = {h->ddim[1][1], h->ddim[1][2], h->ddim[2][1], h->ddim[2][2], h->ddim[2][1] / 4, h->ddim[2][2] / 4};
      const int G = (long)NBt * geo.TH * geo.TW >= 64 ? 64 : 8;
      if ((size_t)G * 27 * RD_D2W_TILE > h->wpartial_cap) return bad_arg(h, "d2 wgrad: partial workspace too small");
      RD_TRY(ensure_lds(h, (const void*)k_d2_wgrad_slab_t16, RD_D2WT_LDS));
      hipLaunchKernelGGL(k_d2_wgrad_slab_t16, dim3(4 * G), dim3(512), RD_D2WT_LDS, st, (const rd_bf16_t*)in, (const rd_bf16_t*)h->du[2],
                         h->wpartial, NBt, G, geo);
      hipLaunchKernelGGL(k_d2_wgrad_fold, dim3((27 * RD_D2W_TILE / 4 + 255) / 256), dim3(256), 0, st, h->wpartial, G, grad + h->doff[2]);
      RD_CHECK(h, hipGetLastError());
    } else if (a16 && l == 3 && h->d3_wgrad_slab && h->nd == 16) {
      // a wave owns (tap, quarter of the output channels); items of four samples (12 output positions each)
      ProfScope ps(h, RDGAN_TAG_CRITIC_GEMM, st);
      LaunchScope ls(h, pl, RD_KIND_WGRAD, NBt, plan_flops(h->plans[pl], NBt), st);
      RD_KNAME(h, "k_d3_wgrad_slab16<bf16>");
      h->flops_acc += plan_flops(h->plans[pl], NBt);
      const int G = NBt >= 256 ? 16 : 8;
      if ((size_t)G * 27 * RD_D3W_TILE > h->wpartial_cap) return bad_arg(h, "d3 wgrad: partial workspace too small");
      RD_TRY(ensure_lds(h, (const void*)k_d3_wgrad_slab16, RD_D3W_LDS));
      hipLaunchKernelGGL(k_d3_wgrad_slab16, dim3(16 * G), dim3(512), RD_D3W_LDS, st, (const rd_bf16_t*)in, (const rd_bf16_t*)h->du[3],
                         h->wpartial, NBt, G);
      hipLaunchKernelGGL(k_d3_wgrad_fold, dim3((27 * RD_D3W_TILE / 4 + 255) / 256), dim3(256), 0, st, h->wpartial, G, grad + h->doff[4]);
      RD_CHECK(h, hipGetLastError());
    } else if (a16 && l >= 2) {      // layers 2-4: bf16 activations against bf16 output gradients (on the border-class boxes)
      pl = critic_wgrad_plan(h, l, NBt);
      if (!wgrad16_ok(h->plans[pl], NBt)) return bad_arg(h, "bf16 storage mode: no bf16 weight-gradient tile for this critic layer");
      RD_TRY(launch_wgrad16(h, h->plans[pl], h->d_plans + pl, NBt, in, h->du[l], grad + h->doff[2 * (l - 1)], h->wpartial,
                            h->wpartial_cap, st, RDGAN_TAG_CRITIC_GEMM));
    } else {
    // fp32 storage, layers 2-4: the border-class boxes of the layer's forward plan (k_wgrad_gemm_ws<128,128> + k_wgrad_reduce_box)
    if (l >= 2 && !a16 && h->wave_spec) pl = critic_wgrad_plan(h, l, NBt);
    RD_TRY(launch_wgrad(h, h->plans[pl], h->d_plans + pl, NBt, in, h->du[l], padded ? h->dW1P : grad + h->doff[2 * (l - 1)],
                        h->wpartial, h->wpartial_cap, st, RDGAN_TAG_CRITIC_GEMM, a16));
    }
    if (padded) hipLaunchKernelGGL(k_unpad_w1, dim3(27), dim3(256), 0, st, h->dW1P, grad + h->doff[0], h->Cin, h->CP);
  }
  RD_TRY(side_join(h, st, h->ev_join));                 // the bias-gradient sums issued on the side stream above
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    const int RS = h->dense_slices > 0 ? std::min(16, h->dense_slices) : std::max(1, std::min(16, NBt / 384));   // row slices of the Dense weight gradient
    float* dwo = RS > 1 ? h->dw6_part : grad + h->doff[8];
    const dim3 dg((h->F + 15) / 16, RS);
    if (a16) hipLaunchKernelGGL(k_critic_dense_wgrad<rd_bf16_t>, dg, dim3(256), 0, st, (const rd_bf16_t*)h->dh[4], dwo, NBt, h->F, B);
    else hipLaunchKernelGGL(k_critic_dense_wgrad<float>, dg, dim3(256), 0, st, (const float*)h->dh[4], dwo, NBt, h->F, B);
    if (RS > 1) hipLaunchKernelGGL(k_reduce_partials, dim3((h->F + 15) / 16), dim3(256), 0, st, h->dw6_part, RS, h->F, grad + h->doff[8]);
    // (grad[doff[9]] = sum of dv over real|fake = 0, cleared by the loss kernel together with the unused loss slots)
    hipLaunchKernelGGL(k_critic_losses, dim3(1), dim3(256), 0, st, h->v, h->gpv, grad + h->n_critic, B, RD_GP_WEIGHT, h->d_flag,
                       grad + h->doff[9]);
  }
  RD_CHECK(h, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------
// generator step gradients (T:395-408)
// ------------------------------------------------------------------------------------
extern "C" int rdgan_gen_grad(rdgan_handle* h, const float* dp, const float* gp, const float* z, const float* cond,
                              uint64_t seed, float* grad, int B, void* stream) {
  return rdgan_gen_grad_after(h, dp, gp, z, cond, seed, grad, B, nullptr, stream);
}

extern "C" int rdgan_gen_grad_after(rdgan_handle* h, const float* dp, const float* gp, const float* z, const float* cond,
                                    uint64_t seed, float* grad, int B, void* critic_ready_event, void* stream) {
  if (!h || !dp || !gp || !z || !cond || !grad) return bad_arg(h, "gen_grad: null pointer");
  if (B < 1 || B > h->MB) return bad_arg(h, "gen_grad: B outside [1, max_batch]");
  hipStream_t st = (hipStream_t)stream;
  const int nd = h->nd;
  h->gate_keep_B = 0;
  if (!h->collapse)
    for (int l = 1; l <= 3; ++l)
      RD_TRY(launch_transpose(h, gp + h->goff[2 * l], h->GWT[l], 27, h->gch[l - 1], h->gch[l], h->gch[l - 1], st));
  // the generator forward reads no critic weight: the last critic update (all-reduce + Adam on the caller's other
  // stream) hides behind it; everything below the wait reads them
  hipStream_t ws = side_fork(h, st);                    // weight-only kernels beside the generator forward (see rdgan_critic_grad_after)
  RD_TRY(gen_forward_impl(h, gp, z, cond, h->fake, B, st, ws));
  if (critic_ready_event) RD_CHECK(h, hipStreamWaitEvent(ws, (hipEvent_t)critic_ready_event, 0));
  RD_TRY(prep_critic_weights(h, dp, ws));
  RD_TRY(side_join(h, st, h->ev_cw));
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    launch_build_critic_input(h, nullptr, h->fake, cond, B, 1, 0u, 0u, st);
  }
  RD_TRY(critic_forward_impl(h, dp, B, seed, st));              // critic frozen, dropout active (T:395,405)
  RD_TRY(critic_dgrad_chain(h, dp, B, B, 1, seed, st));
  RD_TRY(critic_input_grad(h, dp, h->du[1], B, st));                 // g0 = dL/d fake
  const bool a16 = h->a16 != 0;
  const long npix3 = (long)B * h->gpix[3];
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    long ncol = (long)B * nd * nd;
    hipLaunchKernelGGL(k_softmax_bwd, dim3((unsigned)((ncol + 63) / 64)), dim3(64), 0, st, h->fake, h->g0, h->dl, B,
                       RDGAN_NHOURS, nd, nd);
  }
  // last conv (64 -> 1, T:345): weight grad [27][64], bias grad, input grad.  Direct form (h->g9_direct): straight from the
  // 1-channel dlogits with the four neighbouring hour planes in LDS -- no im2col matrix, and the input gradient goes
  // through block 3's PixelNorm+LeakyReLU backward in the same kernel (below); needs the planes to fit in LDS.
  const size_t g9_lds = 4 * (size_t)(nd + 2) * (nd + 2) * sizeof(float);
  const bool g9_direct = h->g9_direct && g9_lds <= 96 * 1024 &&
                         (size_t)B * (RDGAN_NHOURS / 2) * 1728 <= h->wpartial_cap;
  if (a16 && !g9_direct) return bad_arg(h, "bf16 storage mode needs the direct backward of the last conv (g9_direct)");
  if (g9_direct) {
    ProfScope ps(h, RDGAN_TAG_GCONV_WGRAD, st);
    const size_t lds = std::max<size_t>(g9_lds, 4 * 27 * 16 * sizeof(f32x4));
    RD_TRY(ensure_lds(h, a16 ? (const void*)k_g9_bwd_pairs<rd_bf16_t> : (const void*)k_g9_bwd_pairs<float>, 96 * 1024));
    int nwg;
    if (h->edge_kernels && g9w_mfma_ok(nd, npix3) && (size_t)std::min<long>((npix3 + 127) / 128, 768) * 1728 <= h->wpartial_cap) {
      // weight gradient on the matrix pipe, the h3 tensor streamed once (rdgan_edge.hip.h)
      const size_t lds_m = g9w_mfma_lds(a16);
      RD_TRY(ensure_lds(h, a16 ? (const void*)k_g9_wgrad_mfma<rd_bf16_t> : (const void*)k_g9_wgrad_mfma<float>, lds_m));
      nwg = (int)std::min<long>((npix3 + 127) / 128, 768);          // persistent: three workgroups per CU
      LaunchScope ls(h, PL_G9B, RD_KIND_WGRAD, B, 2.0 * npix3 * 64 * 27, st);
      RD_KNAME(h, "k_g9_wgrad_mfma<%s>", a16 ? "bf16" : "f32");
      h->flops_acc += 2.0 * npix3 * 64 * 27;
      if (a16) hipLaunchKernelGGL(k_g9_wgrad_mfma<rd_bf16_t>, dim3(nwg), dim3(256), lds_m, st, h->dl, (const rd_bf16_t*)h->h3, h->wpartial,
                                  npix3, RDGAN_NHOURS, nd, nd, ilog2(nd));
      else hipLaunchKernelGGL(k_g9_wgrad_mfma<float>, dim3(nwg), dim3(256), lds_m, st, h->dl, (const float*)h->h3, h->wpartial,
                              npix3, RDGAN_NHOURS, nd, nd, ilog2(nd));
    } else {
      RD_TRY(ensure_lds(h, a16 ? (const void*)k_g9_wgrad_pairs<rd_bf16_t> : (const void*)k_g9_wgrad_pairs<float>, 96 * 1024));
      const int nunits = B * (RDGAN_NHOURS / 2);
      nwg = std::min(nunits, 3072);            // (bs 256: one unit per workgroup, as before)
      if (a16) hipLaunchKernelGGL(k_g9_wgrad_pairs<rd_bf16_t>, dim3(nwg), dim3(256), lds, st, h->dl, (const rd_bf16_t*)h->h3, h->wpartial,
                                  RDGAN_NHOURS, nd, nd, nunits);
      else hipLaunchKernelGGL(k_g9_wgrad_pairs<float>, dim3(nwg), dim3(256), lds, st, h->dl, (const float*)h->h3, h->wpartial,
                              RDGAN_NHOURS, nd, nd, nunits);
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3((1728 + 15) / 16), dim3(rd_reduce_threads(nwg)), 0, st, h->wpartial, nwg, 1728, grad + h->goff[8]);
  } else {
    {
      ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
      hipLaunchKernelGGL(k_dl_im2col, dim3(ew_blocks(npix3 * 8)), dim3(256), 0, st, h->dl, h->P9, B, RDGAN_NHOURS, nd, nd);
    }
    RD_TRY(launch_wgrad(h, h->plans[PL_G9B], h->d_plans + PL_G9B, B, h->P9, h->h3, grad + h->goff[8], h->wpartial,
                        h->wpartial_cap, st, RDGAN_TAG_GCONV_WGRAD));
    RD_TRY(launch_conv(h, h->plans[PL_G9B], h->d_plans + PL_G9B, B, h->P9, gp + h->goff[8], 64, h->gh3,
                       epi_make(RD_EPI_PLAIN), st, RDGAN_TAG_GCONV_DGRAD));
  }
  {   // bias gradient of the last conv = sum of dl: on the side stream (dl is not written again in this call)
    hipStream_t cs = side_fork(h, st);
    RD_TRY(launch_colsum(h, h->dl, npix3 / 64, 64, h->g9b_tmp, cs));   // 64 partial sums of dl (npix3 % 64 == 0)
    RD_TRY(launch_colsum(h, h->g9b_tmp, 64, 1, grad + h->goff[9], cs));
  }
  // three [upsample, conv, pixelnorm, lrelu] blocks, last to first
  float* hs[4] = {h->h0, h->h1, h->h2, h->h3};
  float* rs[4] = {nullptr, h->r1, h->r2, h->r3};
  float* dys[4] = {nullptr, h->dy1, h->dy2, h->gh3};
  float* gups[4] = {nullptr, h->gup1, h->gup2, h->gup3};   // direct: gradient on the upsampled grid; collapsed: on the source grid
  const int col = h->collapse;
  if (col) {
    RdSliceMap map;
    collapsed_dgrad_slice_map(map.src);
    for (int l = 1; l <= 3; ++l) {   // Wd[q][Cout][Cin] <- Wc (written by gen_forward_impl above unless that block ran in the shared-centre form)
      if (gen_block_fast(h, l, fast_bwd_on(h))) continue;
      if (gen_block_fast(h, l, fast_fwd_on(h)))
        hipLaunchKernelGGL(k_collapse_weights, dim3(ew_blocks(16L * h->gch[l - 1] * h->gch[l])), dim3(256), 0, st,
                           gp + h->goff[2 * l], h->GWC[l], h->gch[l - 1] * h->gch[l]);
      if (a16) continue;              // (bf16 storage: the input gradient reads the bf16 image of Wc re-ordered by tap, below)
      hipLaunchKernelGGL(k_transpose_map, dim3((h->gch[l] + 31) / 32, (h->gch[l - 1] + 31) / 32, 64), dim3(256), 0, st,
                         h->GWC[l], h->GWD[l], h->gch[l - 1], h->gch[l], map);
    }
  }
  for (int l = 3; l >= 1; --l) {
    // shared-centre backward only where the hour axis is long enough to pay for its (D+1)/D boundary plane
    const bool fast = gen_block_fast(h, l, fast_bwd_on(h));
    if (l == 3 && g9_direct) {
      // input gradient of the 64 -> 1 conv + block 3's PixelNorm+LeakyReLU backward (+ plane-pair sums)
      ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
      if (a16 && !fast && h->g9_bwd_mfma && (nd * nd) % 32 == 0) {
        // exact fp32 products on the matrix pipe, the row's backward in registers (rdgan_g9bwd16.hip.h)
        const int PP = nd <= 16 ? 4 : 1;        // plane pairs per unit (ten staged planes = 13 KB at ndomain 16; 3 B units deal evenly to 768 workgroups)
        const size_t lds9 = (size_t)(2 * PP + 2) * (nd + 2) * (nd + 2) * sizeof(float);
        const int nunits = B * (RDGAN_NHOURS / 2 / PP);
        RD_TRY(ensure_lds(h, (const void*)k_g9_bwd_mfma16, lds9));
        hipLaunchKernelGGL(k_g9_bwd_mfma16, dim3((unsigned)std::min(nunits, PP > 1 ? 768 : 1536)), dim3(256), lds9, st, h->dl, gp + h->goff[8],
                           (const rd_bf16_t*)hs[3], rs[3], (rd_bf16_t*)dys[3], nunits, RDGAN_NHOURS, nd, nd, PP);
      } else
      if (a16) hipLaunchKernelGGL(k_g9_bwd_pairs<rd_bf16_t>, dim3(B * (RDGAN_NHOURS / 2)), dim3(256), g9_lds, st, h->dl, gp + h->goff[8],
                                  (const rd_bf16_t*)hs[3], rs[3], (rd_bf16_t*)dys[3], fast ? (rd_bf16_t*)h->fgS : (rd_bf16_t*)nullptr,
                                  RDGAN_NHOURS, nd, nd);
      else hipLaunchKernelGGL(k_g9_bwd_pairs<float>, dim3(B * (RDGAN_NHOURS / 2)), dim3(256), g9_lds, st, h->dl, gp + h->goff[8],
                              (const float*)hs[3], rs[3], dys[3], fast ? h->fgS : (float*)nullptr, RDGAN_NHOURS, nd, nd);
      RD_CHECK(h, hipGetLastError());
    } else if (fast) {
      ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
      const long HW = (long)h->gdim[l][1] * h->gdim[l][2];
      RD_TRY(launch_pn_bwd_pairs(h, l == 3 ? h->gh3 : gups[l + 1], hs[l], rs[l], dys[l], h->fgS, (long)B * h->gdim[l - 1][0] * HW,
                                 HW, h->gch[l], st, a16));
    } else {
      ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
      if (l == 3) RD_TRY(launch_pn_bwd(h, h->gh3, hs[3], rs[3], dys[3], npix3, 64, 0, 0, 0, 0, st, a16));
      else RD_TRY(launch_pn_bwd(h, gups[l + 1], hs[l], rs[l], dys[l], (long)B * h->gpix[l], h->gch[l], col ? 0 : 1,
                                h->gdim[l][0], h->gdim[l][1], h->gdim[l][2], st, a16));
    }
    const long cc = (long)h->gch[l - 1] * h->gch[l];
    if (fast) {
      // shared-centre form along d: E = d-differences of the block input, gS = sums of the output-gradient plane pairs
      const int* sd = h->gdim[l - 1];
      const int D = sd[0];
      const long P = (long)sd[1] * sd[2] * h->gch[l - 1];
      RdWeightMap wm;
      fastd_weight_map(wm);
      {
        ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
        if (!gen_block_fast(h, l, fast_fwd_on(h))) {      // (otherwise the forward pass above has left both)
          const dim3 dg(ew_blocks((long)B * (D + 1) * P / 4));
          if (a16) hipLaunchKernelGGL(k_diff_d<rd_bf16_t>, dg, dim3(256), 0, st, (const rd_bf16_t*)hs[l - 1], (rd_bf16_t*)h->fE[l], B, D, P);
          else
          hipLaunchKernelGGL(k_diff_d<float>, dg, dim3(256), 0, st, (const float*)hs[l - 1],
                             h->fE[l], B, D, P);
          hipLaunchKernelGGL(k_weight_transform, dim3(ew_blocks(48L * cc / 4)), dim3(256), 0, st, gp + h->goff[2 * l], h->fU[l],
                             (int)cc, 48, wm);
        }
      }
      // weight gradients of the three tap groups: E[s] x even planes, x x plane sums, E[s+1] x odd planes
      const float* wsrc[3] = {h->fE[l], hs[l - 1], h->fE[l]};
      const float* wdy[3] = {dys[l], h->fgS, dys[l]};
      for (int g = 0; g < 3; ++g) {
        int pl = PL_F1WA + 3 * g + l - 1;
        // fp32 storage: the (h, w) border boxes of the block (12 % of its row-tap products are products with a zero row)
        // -- where they drop at least a fifth of the work (block 2: 23 %, 0.237 -> 0.219 ms per launch; block 3's 12 % do not pay
        // for the extra partial slabs and short row tiles of its 256 x 64 tiling: 0.43 -> 0.51 ms, measured)
        if (!a16 && h->wgrad_boxes && h->border_boxes && h->wave_spec && wgrad_box_ok(h->plans[PL_F1WAX + 3 * g + l - 1]) &&
            (h->border_boxes >= 2 || plan_flops(h->plans[PL_F1WAX + 3 * g + l - 1], 1) < 0.8 * plan_flops(h->plans[pl], 1)))
          pl = PL_F1WAX + 3 * g + l - 1;
        if (a16) {
          if (!wgrad16_ok(h->plans[pl], B)) return bad_arg(h, "bf16 storage mode: no bf16 weight-gradient tile for this block");
          RD_TRY(launch_wgrad16(h, h->plans[pl], h->d_plans + pl, B, wsrc[g], wdy[g], h->fdU, h->wpartial, h->wpartial_cap, st,
                                RDGAN_TAG_GCONV_WGRAD));
        } else
        RD_TRY(launch_wgrad(h, h->plans[pl], h->d_plans + pl, B, wsrc[g], wdy[g], h->fdU, h->wpartial, h->wpartial_cap, st,
                            RDGAN_TAG_GCONV_WGRAD));
      }
      hipLaunchKernelGGL(k_weight_transform_adj, dim3(ew_blocks(27L * cc / 4)), dim3(256), 0, st, h->fdU, grad + h->goff[2 * l],
                         (int)cc, 48, wm);
      RD_TRY(launch_colsum(h, dys[l], (long)B * h->gpix[l], h->gch[l], grad + h->goff[2 * l + 1], side_fork(h, st), a16));   // bias gradient, beside the GEMMs (dys[l] is read-only from here on)
      const int pbs = PL_F1BS + l - 1, pbe = PL_F1BE + l - 1;
      RdEpi eb = epi_make(RD_EPI_PLAIN);
      eb.out16 = a16;
      if (a16) {
        // the forward forms U re-ordered by tap ([Cin][Cout] is already the [N][K] layout of these GEMMs)
        RdSliceMap map;
        fastd_dgrad_slice_map(map.src);
        {
          ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
          hipLaunchKernelGGL(k_blocks_to_bf16, dim3((unsigned)std::min<long>((cc / 8 + 255) / 256, 64), 48), dim3(256), 0, st,
                             h->fU[l], (unsigned short*)h->bUT, cc, map);
        }
        RD_TRY(launch_conv16(h, h->plans[pbs], h->d_plans + pbs, B, h->fgS, h->bUT, gups[l], eb, st, RDGAN_TAG_GCONV_DGRAD));
        RD_TRY(launch_conv16(h, h->plans[pbe], h->d_plans + pbe, B, dys[l], h->bUT, h->fdE, eb, st, RDGAN_TAG_GCONV_DGRAD));
      } else {
        {
          RdSliceMap map;
          fastd_dgrad_slice_map(map.src);
          hipLaunchKernelGGL(k_transpose_map, dim3((h->gch[l] + 31) / 32, (h->gch[l - 1] + 31) / 32, 48), dim3(256), 0, st, h->fU[l],
                             h->fUT, h->gch[l - 1], h->gch[l], map);
        }
        RD_TRY(launch_conv(h, h->plans[pbs], h->d_plans + pbs, B, h->fgS, h->fUT, h->gch[l - 1], gups[l], eb, st,
                           RDGAN_TAG_GCONV_DGRAD));
        RD_TRY(launch_conv(h, h->plans[pbe], h->d_plans + pbe, B, dys[l], h->fUT, h->gch[l - 1], h->fdE, eb, st,
                           RDGAN_TAG_GCONV_DGRAD));
      }
      {
        ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
        const dim3 cg(ew_blocks((long)B * D * P / 4));
        if (a16) hipLaunchKernelGGL(k_combine_dx<rd_bf16_t>, cg, dim3(256), 0, st, (rd_bf16_t*)gups[l], (const rd_bf16_t*)h->fdE, B, D, P);
        else hipLaunchKernelGGL(k_combine_dx<float>, cg, dim3(256), 0, st, gups[l], (const float*)h->fdE, B, D, P);
      }
    } else if (col) {
      int plf = PL_G1FC + l - 1, plb = gen_box_plan(h, PL_G1BC + l - 1, PL_G1BCX + l - 1, B);
      // weight gradient through the streaming kernels: on the boxes where they drop a fifth of the work (as the shared-centre ones)
      const int plfx = PL_G1FCX + l - 1;
      const bool wbox = h->wgrad_boxes && h->border_boxes && h->wave_spec && wgrad_box_ok(h->plans[plfx]) &&
                        (h->border_boxes >= 2 || plan_flops(h->plans[plfx], 1) < 0.8 * plan_flops(h->plans[plf], 1));
      bool bias_done = false;         // (the slab kernel delivers the bias gradient too)
      if (a16 && l == 3 && h->upwgrad_slab && h->nd == 16) {
        // each workgroup owns one phase and keeps its eight tap products in registers over its share of the batch
        ProfScope ps(h, RDGAN_TAG_GCONV_WGRAD, st);
        LaunchScope ls(h, plf, RD_KIND_WGRAD, B, plan_flops(h->plans[plf], B), st);
        RD_KNAME(h, "k_upconv_wgrad_slab16<bf16>");
        h->flops_acc += plan_flops(h->plans[plf], B);
        const int G = 6 * B >= 64 ? 32 : 8;
        if ((size_t)G * 64 * RD_UWG_TILE > h->wpartial_cap) return bad_arg(h, "upconv wgrad: partial workspace too small");
        RD_TRY(ensure_lds(h, (const void*)k_upconv_wgrad_slab16, RD_UWG_LDS));
        hipLaunchKernelGGL(k_upconv_wgrad_slab16, dim3(8 * G), dim3(512), RD_UWG_LDS, st, (const rd_bf16_t*)hs[l - 1],
                           (const rd_bf16_t*)dys[l], h->wpartial, B, G, h->ubias_part);
        hipLaunchKernelGGL(k_upconv_wgrad_fold, dim3(64 * RD_UWG_TILE / 4 / 256), dim3(256), 0, st, h->wpartial, G, h->dWc);
        hipLaunchKernelGGL(k_reduce_partials, dim3(64 / 16), dim3(rd_reduce_threads(8 * G)), 0, st, h->ubias_part, 8 * G, 64, grad + h->goff[2 * l + 1]);      // (a serial fold of the 256 partial rows took 61 us)
        bias_done = true;
        RD_CHECK(h, hipGetLastError());
      } else if (a16 && l == 2 && h->upwgrad_slab && h->nd == 16) {
        // the same on block 2's geometry: a workgroup owns (phase, quarter of the 256 input channels)
        ProfScope ps(h, RDGAN_TAG_GCONV_WGRAD, st);
        LaunchScope ls(h, plf, RD_KIND_WGRAD, B, plan_flops(h->plans[plf], B), st);
        RD_KNAME(h, "k_upconv2_wgrad_slab16<bf16>");
        h->flops_acc += plan_flops(h->plans[plf], B);
        const int G = 8;
        if ((size_t)G * 64 * RD_UW2_TILE > h->wpartial_cap) return bad_arg(h, "upconv wgrad: partial workspace too small");
        RD_TRY(ensure_lds(h, (const void*)k_upconv2_wgrad_slab16, RD_UW2_LDS));
        hipLaunchKernelGGL(k_upconv2_wgrad_slab16, dim3(32 * G), dim3(512), RD_UW2_LDS, st, (const rd_bf16_t*)hs[l - 1],
                           (const rd_bf16_t*)dys[l], h->wpartial, B, G, h->ubias_part);
        hipLaunchKernelGGL(k_upconv2_wgrad_fold, dim3(64 * RD_UW2_TILE / 4 / 256), dim3(256), 0, st, h->wpartial, G, h->dWc);
        hipLaunchKernelGGL(k_reduce_partials, dim3(128 / 16), dim3(rd_reduce_threads(8 * G)), 0, st, h->ubias_part, 8 * G, 128, grad + h->goff[2 * l + 1]);
        bias_done = true;
        RD_CHECK(h, hipGetLastError());
      } else if (a16) {
        if (wbox && wgrad16_ok(h->plans[plfx], B)) plf = plfx;
        if (!wgrad16_ok(h->plans[plf], B)) return bad_arg(h, "bf16 storage mode: no bf16 weight-gradient tile for this block");
        RD_TRY(launch_wgrad16(h, h->plans[plf], h->d_plans + plf, B, hs[l - 1], dys[l], h->dWc, h->wpartial, h->wpartial_cap, st,
                              RDGAN_TAG_GCONV_WGRAD));
      } else {
      if (wbox) plf = plfx;
      RD_TRY(launch_wgrad(h, h->plans[plf], h->d_plans + plf, B, hs[l - 1], dys[l], h->dWc, h->wpartial, h->wpartial_cap,
                          st, RDGAN_TAG_GCONV_WGRAD));
      }
      hipLaunchKernelGGL(k_fold_collapsed_wgrad, dim3(ew_blocks(27L * cc / 4)), dim3(256), 0, st, h->dWc,
                         grad + h->goff[2 * l], (int)cc);
      if (!bias_done)
      RD_TRY(launch_colsum(h, dys[l], (long)B * h->gpix[l], h->gch[l], grad + h->goff[2 * l + 1], side_fork(h, st), a16));   // bias gradient, beside the GEMMs (dys[l] is read-only from here on)
      if (a16) {      // the collapsed forms re-ordered by tap are already [N = Cin][K = Cout]
        RdSliceMap map;
        collapsed_dgrad_slice_map(map.src);
        hipLaunchKernelGGL(k_blocks_to_bf16, dim3((unsigned)std::min<long>((cc / 8 + 255) / 256, 64), 64), dim3(256), 0, st,
                           h->GWC[l], (unsigned short*)h->bG1B, cc, map, h->conv_f16 ? (unsigned short*)h->fG1B : nullptr, h->gch[l]);
        RdEpi eb = epi_make(RD_EPI_PLAIN);
        eb.out16 = 1;
        RD_TRY(launch_conv16(h, h->plans[plb], h->d_plans + plb, B, dys[l], h->bG1B, gups[l], eb, st, RDGAN_TAG_GCONV_DGRAD,
                             h->conv_f16 ? h->fG1B : nullptr));
      } else
      RD_TRY(launch_conv(h, h->plans[plb], h->d_plans + plb, B, dys[l], h->GWD[l], h->gch[l - 1], gups[l],
                         epi_make(RD_EPI_PLAIN), st, RDGAN_TAG_GCONV_DGRAD));
    } else {
      int plf = PL_G1F + l - 1, plb = PL_G1B + l - 1;
      RD_TRY(launch_wgrad(h, h->plans[plf], h->d_plans + plf, B, hs[l - 1], dys[l], grad + h->goff[2 * l], h->wpartial,
                          h->wpartial_cap, st, RDGAN_TAG_GCONV_WGRAD));
      RD_TRY(launch_colsum(h, dys[l], (long)B * h->gpix[l], h->gch[l], grad + h->goff[2 * l + 1], side_fork(h, st)));
      RD_TRY(launch_conv(h, h->plans[plb], h->d_plans + plb, B, dys[l], h->GWT[l], h->gch[l - 1], gups[l],
                         epi_make(RD_EPI_PLAIN), st, RDGAN_TAG_GCONV_DGRAD));
    }
  }
  // Dense (T:326): (pool the upsample adjoint,) LeakyReLU', then dW = xcat^T ga0, db = colsum(ga0)
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    const dim3 lg(ew_blocks((long)B * h->gpix[0] * 64));
    if (col && a16)
      hipLaunchKernelGGL(k_lrelu_bwd<rd_bf16_t>, lg, dim3(256), 0, st, (const rd_bf16_t*)h->gup1, (const rd_bf16_t*)h->h0, h->ga0,
                         (long)B * h->gpix[0] * 64);
    else if (col)
      hipLaunchKernelGGL(k_lrelu_bwd<float>, lg, dim3(256), 0, st, (const float*)h->gup1, (const float*)h->h0, h->ga0,
                         (long)B * h->gpix[0] * 64);
    else
      hipLaunchKernelGGL(k_pool_lrelu_bwd, lg, dim3(256), 0, st, h->gup1, h->h0,
                         h->ga0, (long)B * h->gpix[0], h->gdim[0][0], h->gdim[0][1], h->gdim[0][2], 256);
  }
  RD_TRY(launch_wgrad(h, h->plans[PL_GDENSE], h->d_plans + PL_GDENSE, B, h->xcat, h->ga0, grad + h->goff[0], h->wpartial,
                      h->wpartial_cap, st, RDGAN_TAG_GCONV_WGRAD));
  RD_TRY(launch_colsum(h, h->ga0, B, h->n_nodes, grad + h->goff[1], side_fork(h, st)));   // (same stream as the other column sums: they share their scratch)
  RD_TRY(side_join(h, st, h->ev_join));                 // the bias-gradient sums issued on the side stream above
  {
    ProfScope ps(h, RDGAN_TAG_ELEMENTWISE, st);
    hipLaunchKernelGGL(k_gen_loss, dim3(1), dim3(256), 0, st, h->v, grad + h->n_gen, B, h->d_flag);
  }
  RD_CHECK(h, hipGetLastError());
  return 0;
}

extern "C" int rdgan_adam(float* params, const float* grad, float* v, long n, int t, float lr, float beta2, float eps,
                          float grad_scale, void* stream) {
  if (!params || !grad || !v || n < 1 || t < 1) return -2;
  float lr_t = lr * sqrtf(1.0f - powf(beta2, (float)t));
  hipLaunchKernelGGL(k_adam, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, params, grad, v, n, lr_t, beta2,
                     eps, grad_scale);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// op-level entry points for the parity tests
// ------------------------------------------------------------------------------------
struct TmpPlan {
  RdPlan host; RdPlan* dev = nullptr; RdRow* tab = nullptr;
  int upload() {
    std::vector<RdRow> t;
    if (!plan_build_tables(host, t)) return -2;
    hipError_t e = hipMalloc((void**)&tab, sizeof(RdRow) * t.size());
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(tab, t.data(), sizeof(RdRow) * t.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
    host.tab = tab;
    e = hipMalloc((void**)&dev, sizeof(RdPlan));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpy(dev, &host, sizeof(RdPlan), hipMemcpyHostToDevice);
  }
  ~TmpPlan() { if (dev) (void)hipFree(dev); if (tab) (void)hipFree(tab); }
};

extern "C" int rdgan_op_conv3d(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W,
                               int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h, int pad_w,
                               int upsample, void* stream) {
  if (!x || !w || !y || Cin % 4 || (Cout % 64 && Cout != 32)) return -2;
  TmpPlan tp;
  tp.host = plan_conv_fwd(D, H, W, Cin, Cout, Do, Ho, Wo, stride, pad_d, pad_h, pad_w, upsample);
  RD_TRY(tp.upload());
  hipStream_t st = (hipStream_t)stream;
  RD_TRY(launch_conv(nullptr, tp.host, tp.dev, B, x, w, Cout, y, epi_make(bias ? RD_EPI_BIAS : RD_EPI_PLAIN, bias), st, -1));
  return (int)hipStreamSynchronize(st);
}

// weight gradient with bf16 operands (tests): x and gy rounded to bf16 on the device, fp32 accumulation.  Cin % 128 == 0,
// Cout % 64 == 0, no folded upsample.
extern "C" int rdgan_op_conv3d_wgrad_bf16(const float* x, const float* gy, float* dw, int B, int D, int H, int W, int Cin,
                                          int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h, int pad_w,
                                          void* stream) {
  if (!x || !gy || !dw || Cin % 64 || Cout % 64) return -2;
  hipStream_t st = (hipStream_t)stream;
  TmpPlan tp;
  tp.host = plan_conv_fwd(D, H, W, Cin, Cout, Do, Ho, Wo, stride, pad_d, pad_h, pad_w, 0);
  RD_TRY(tp.upload());
  if (!wgrad16_ok(tp.host, B)) return -2;
  const long nx = (long)B * D * H * W * Cin, ng = (long)B * Do * Ho * Wo * Cout;
  size_t need = std::max(wgrad_partial_need(tp.host, B), wgrad_partial_need(tp.host, B, true));
  void *xb = nullptr, *gb = nullptr; float* partial = nullptr;
  hipError_t e = hipMalloc(&xb, nx * 2);
  if (e == hipSuccess) e = hipMalloc(&gb, ng * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&partial, need * sizeof(float));
  int rc = (int)e;
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, gy, gb, ng, st);
  if (rc == 0) rc = launch_wgrad16(nullptr, tp.host, tp.dev, B, xb, gb, dw, partial, need, st, -1);
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  if (xb) (void)hipFree(xb);
  if (gb) (void)hipFree(gb);
  if (partial) (void)hipFree(partial);
  return rc;
}

// bf16-operand variant of the above for stride-1/2 convs without folded upsample (tests): x and w are rounded to
// bf16 (nearest even) on the device, products accumulate in fp32.  Cin % 64 == 0, Cout % 64 == 0.
extern "C" int rdgan_op_conv3d_bf16(const float* x, const float* w, const float* bias, float* y, int B, int D, int H, int W,
                                    int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h, int pad_w,
                                    int out_bf16, void* stream) {
  if (!x || !w || !y || Cin % 64 || Cout % 64) return -2;
  hipStream_t st = (hipStream_t)stream;
  TmpPlan tp;
  tp.host = plan_conv_fwd(D, H, W, Cin, Cout, Do, Ho, Wo, stride, pad_d, pad_h, pad_w, 0);
  RD_TRY(tp.upload());
  const long nx = (long)B * D * H * W * Cin, nw = 27L * Cin * Cout;
  void *xb = nullptr, *wb = nullptr;
  hipError_t e = hipMalloc(&xb, nx * 2);
  if (e == hipSuccess) e = hipMalloc(&wb, nw * 2);
  int rc = (int)e;
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) rc = launch_weights_to_bf16_t(nullptr, w, wb, 27, Cin, Cout, st);
  RdEpi ep = epi_make(bias ? RD_EPI_BIAS : RD_EPI_PLAIN, bias);
  ep.out16 = out_bf16 ? 1 : 0;
  void* wf = nullptr;
  if (out_bf16 == 2) {       // the fragment kernel (k_conv_gemm_f16), which must accept the launch
    if (!conv_f16_ok(nullptr, tp.host, B, ep)) rc = -2;
    if (rc == 0) rc = (int)hipMalloc(&wf, nw * 2);
    if (rc == 0) rc = launch_wfrag_image(nullptr, wb, wf, 27, Cout, Cin, st);
  }
  if (rc == 0) rc = launch_conv16(nullptr, tp.host, tp.dev, B, xb, wb, y, ep, st, -1, wf);
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  if (xb) (void)hipFree(xb);
  if (wb) (void)hipFree(wb);
  if (wf) (void)hipFree(wf);
  return rc;
}

extern "C" int rdgan_op_conv3d_dgrad(const float* gy, const float* w, float* gx, int B, int D, int H, int W, int Cin,
                                     int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h, int pad_w,
                                     void* stream) {
  if (!gy || !w || !gx || Cout % 4 || Cin % 64) return -2;
  hipStream_t st = (hipStream_t)stream;
  float* wt = nullptr;
  hipError_t e = hipMalloc((void**)&wt, sizeof(float) * 27 * Cin * Cout);
  if (e != hipSuccess) return (int)e;
  int rc = launch_transpose(nullptr, w, wt, 27, Cin, Cout, Cin, st);
  TmpPlan tp;
  if (stride == 1) {
    if (pad_d != 1 || pad_h != 1 || pad_w != 1 || Do != D || Ho != H || Wo != W) { (void)hipFree(wt); return -2; }
    tp.host = plan_conv_dgrad_s1(D, H, W, Cin, Cout);
  } else {
    int pad[3] = {pad_d, pad_h, pad_w};
    tp.host = plan_conv_dgrad_s2(D, H, W, Cin, Do, Ho, Wo, Cout, pad);
  }
  if (rc == 0) rc = tp.upload();
  if (rc == 0) rc = launch_conv(nullptr, tp.host, tp.dev, B, gy, wt, Cin, gx, epi_make(RD_EPI_PLAIN), st, -1);
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  (void)hipFree(wt);
  return rc;
}

// input gradient with bf16 operands (tests): gy and w rounded to bf16, fp32 accumulation.  The [tap][Cin][Cout] kernel IS
// the [tap block][N][K] layout of this GEMM (N = Cin, K = Cout), so the weights are only converted.
extern "C" int rdgan_op_conv3d_dgrad_bf16(const float* gy, const float* w, float* gx, int B, int D, int H, int W, int Cin,
                                          int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h, int pad_w,
                                          void* stream) {
  if (!gy || !w || !gx || Cout % 64 || Cin % 64) return -2;
  hipStream_t st = (hipStream_t)stream;
  TmpPlan tp;
  if (stride == 1) {
    if (pad_d != 1 || pad_h != 1 || pad_w != 1 || Do != D || Ho != H || Wo != W) return -2;
    tp.host = plan_conv_dgrad_s1(D, H, W, Cin, Cout);
  } else {
    int pad[3] = {pad_d, pad_h, pad_w};
    tp.host = plan_conv_dgrad_s2(D, H, W, Cin, Do, Ho, Wo, Cout, pad);
  }
  const long ng = (long)B * Do * Ho * Wo * Cout, nw = 27L * Cin * Cout;
  void *gb = nullptr, *wb = nullptr;
  hipError_t e = hipMalloc(&gb, ng * 2);
  if (e == hipSuccess) e = hipMalloc(&wb, nw * 2);
  int rc = (int)e;
  if (rc == 0) rc = tp.upload();
  if (rc == 0) rc = launch_to_bf16(nullptr, gy, gb, ng, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, w, wb, nw, st);
  if (rc == 0) rc = launch_conv16(nullptr, tp.host, tp.dev, B, gb, wb, gx, epi_make(RD_EPI_PLAIN), st, -1);
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  if (gb) (void)hipFree(gb);
  if (wb) (void)hipFree(wb);
  return rc;
}

extern "C" int rdgan_op_conv3d_wgrad(const float* x, const float* gy, float* dw, int B, int D, int H, int W, int Cin,
                                     int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h, int pad_w,
                                     int upsample, void* stream) {
  if (!x || !gy || !dw || Cin % 4 || Cout % 64) return -2;
  hipStream_t st = (hipStream_t)stream;
  TmpPlan tp;
  tp.host = plan_conv_fwd(D, H, W, Cin, Cout, Do, Ho, Wo, stride, pad_d, pad_h, pad_w, upsample);
  RD_TRY(tp.upload());
  size_t need = wgrad_partial_need(tp.host, B);
  float* partial = nullptr;
  hipError_t e = hipMalloc((void**)&partial, need * sizeof(float));
  if (e != hipSuccess) return (int)e;
  int rc = launch_wgrad(nullptr, tp.host, tp.dev, B, x, gy, dw, partial, need, st, -1);
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  (void)hipFree(partial);
  return rc;
}

// Weight gradient of one tap group of the shared-centre form through the production plan
// (plan_fastd_wgrad: 4 output-parity phases x 4 taps, an EVEN tap count, which is what lets the launcher pick the 256-row
// tile at B * D*H*W >= 65536): group g = 0: src = E [B,D+1,H,W,Cin] against the even planes of dy [B,2D,2H,2W,Cout];
// g = 1: src = x [B,D,H,W,Cin] against the plane-pair sums gS [B,D,2H,2W,Cout]; g = 2: E at j = s+1 against the odd planes.
// dU [48][Cin][Cout]: the 16 forms g*16 .. g*16+15 are written.  bf16 = 1: operands rounded to bf16 on the device
// (k_wgrad_gemm_ws16), fp32 accumulation and output.
extern "C" int rdgan_op_fastd_wgrad(const float* src, const float* dy, float* dU, int B, int D, int H, int W, int Cin, int Cout,
                                    int g, int bf16, void* stream) {
  if (!src || !dy || !dU || g < 0 || g > 2 || Cin % 4 || Cout % 64) return -2;
  hipStream_t st = (hipStream_t)stream;
  TmpPlan tp;
  tp.host = plan_fastd_wgrad(D, H, W, Cin, Cout, g);
  RD_TRY(tp.upload());
  size_t need = std::max(wgrad_partial_need(tp.host, B), wgrad_partial_need(tp.host, B, true));
  float* partial = nullptr;
  void *xb = nullptr, *gb = nullptr;
  hipError_t e = hipMalloc((void**)&partial, need * sizeof(float));
  int rc = (int)e;
  if (rc == 0 && bf16) {
    if (!wgrad16_ok(tp.host, B)) rc = -2;
    const long nx = (long)B * tp.host.src_sample, ng = (long)B * tp.host.dst_sample;
    if (rc == 0) rc = (int)hipMalloc(&xb, nx * 2);
    if (rc == 0) rc = (int)hipMalloc(&gb, ng * 2);
    if (rc == 0) rc = launch_to_bf16(nullptr, src, xb, nx, st);
    if (rc == 0) rc = launch_to_bf16(nullptr, dy, gb, ng, st);
    if (rc == 0) rc = launch_wgrad16(nullptr, tp.host, tp.dev, B, xb, gb, dU, partial, need, st, -1);
  } else if (rc == 0) {
    rdgan_handle fake_h;                   // (wave_spec = 1: the producer/consumer kernel, as in production)
    rc = launch_wgrad(&fake_h, tp.host, tp.dev, B, src, dy, dU, partial, need, st, -1);
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  if (partial) (void)hipFree(partial);
  if (xb) (void)hipFree(xb);
  if (gb) (void)hipFree(gb);
  return rc;
}

// Weight gradient of the last generator conv (64 -> 1) alone, through the production kernels: kernel = 1 the matrix-pipe kernel
// (k_g9_wgrad_mfma), 0 the scalar kernel (k_g9_wgrad_pairs); bf16 = 1 rounds h3 to bf16 first (storage mode)
extern "C" int rdgan_op_g9_wgrad(const float* dl, const float* h3, float* dW, int B, int nd, int bf16, int kernel, void* stream) {
  if (!dl || !h3 || !dW || B < 1 || nd < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const int D = RDGAN_NHOURS;
  const long npix = (long)B * D * nd * nd;
  if (npix >= 0x7FFFFFFFL) return -2;
  const int nunits = B * (D / 2);
  if (kernel && !g9w_mfma_ok(nd, npix)) return -2;
  const int nwg = kernel ? (int)std::min<long>((npix + 127) / 128, 768) : std::min(nunits, 3072);
  float* partial = nullptr;
  void* hb = nullptr;
  int rc = (int)hipMalloc((void**)&partial, (size_t)nwg * 1728 * sizeof(float));
  if (rc == 0 && bf16) {
    rc = (int)hipMalloc(&hb, npix * 64 * 2);
    if (rc == 0) rc = launch_to_bf16(nullptr, h3, hb, npix * 64, st);
  }
  if (rc == 0 && kernel) {
    const size_t lds = g9w_mfma_lds(bf16 != 0);
    rc = ensure_lds(nullptr, bf16 ? (const void*)k_g9_wgrad_mfma<rd_bf16_t> : (const void*)k_g9_wgrad_mfma<float>, lds);
    if (rc == 0) {
      if (bf16) hipLaunchKernelGGL(k_g9_wgrad_mfma<rd_bf16_t>, dim3(nwg), dim3(256), lds, st, dl, (const rd_bf16_t*)hb, partial, npix, D, nd, nd, ilog2(nd));
      else hipLaunchKernelGGL(k_g9_wgrad_mfma<float>, dim3(nwg), dim3(256), lds, st, dl, h3, partial, npix, D, nd, nd, ilog2(nd));
    }
  } else if (rc == 0) {
    const size_t g9_lds = 4 * (size_t)(nd + 2) * (nd + 2) * sizeof(float);
    const size_t lds = std::max<size_t>(g9_lds, 4 * 27 * 16 * sizeof(f32x4));
    if (lds > 96 * 1024) rc = -2;
    if (rc == 0) rc = ensure_lds(nullptr, bf16 ? (const void*)k_g9_wgrad_pairs<rd_bf16_t> : (const void*)k_g9_wgrad_pairs<float>, 96 * 1024);
    if (rc == 0) {
      if (bf16) hipLaunchKernelGGL(k_g9_wgrad_pairs<rd_bf16_t>, dim3(nwg), dim3(256), lds, st, dl, (const rd_bf16_t*)hb, partial, D, nd, nd, nunits);
      else hipLaunchKernelGGL(k_g9_wgrad_pairs<float>, dim3(nwg), dim3(256), lds, st, dl, h3, partial, D, nd, nd, nunits);
    }
  }
  if (rc == 0) {
    hipLaunchKernelGGL(k_reduce_partials, dim3((1728 + 15) / 16), dim3(rd_reduce_threads(nwg)), 0, st, partial, nwg, 1728, dW);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  if (partial) (void)hipFree(partial);
  if (hb) (void)hipFree(hb);
  return rc;
}

// Generator block 3 forward of the bf16 storage mode through the slab kernel alone (rdgan_upconv16.hip.h): x [B,12,8,8,128] and the
// Conv3D kernel w [3,3,3,128,64] are rounded to bf16 on the device (the kernel after the upsample collapse), y [B,24,16,16,64] =
// LeakyReLU(PixelNorm(upconv(x) + bias)) comes back as fp32 (the kernel's bf16 output widened), rinv [B,24,16,16]; dbg (optional,
// [B*24*16*16][4]): the row sums of squares and 1/l2 as both lane halves computed them.
extern "C" int rdgan_op_upconv_slab16(const float* x, const float* w, const float* bias, float* y, float* rinv, float* dbg, int B,
                                      void* stream) {
  if (!x || !w || !bias || !y || !rinv || B < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long nx = (long)B * 12 * 64 * 128, ny = (long)B * 24 * 256 * 64;
  void *xb = nullptr, *yb = nullptr, *wi = nullptr; float* wc = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc(&wi, 64L * 8 * 2 * 64 * 16);
  if (rc == 0) rc = (int)hipMalloc((void**)&wc, 64L * 128 * 64 * sizeof(float));
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) {
    hipLaunchKernelGGL(k_collapse_weights, dim3(ew_blocks(16L * 128 * 64)), dim3(256), 0, st, w, wc, 128 * 64);
    hipLaunchKernelGGL(k_upconv_wimg, dim3(256), dim3(256), 0, st, wc, (unsigned short*)wi);
    rc = ensure_lds(nullptr, (const void*)k_upconv_slab16<0>, RD_UPC_LDS);
  }
  if (rc == 0) {
    hipLaunchKernelGGL(k_upconv_slab16<0>, dim3((unsigned)std::min(6 * B, 512)), dim3(256), RD_UPC_LDS, st, (const rd_bf16_t*)xb,
                       (const rd_bf16_t*)wi, bias, (rd_bf16_t*)yb, rinv, B, dbg);
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(ew_blocks(ny)), dim3(256), 0, st, (const rd_bf16_t*)yb, y, ny);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, wi, (void*)wc}) if (p) (void)hipFree(p);
  return rc;
}

// The same block on source planes of H x W positions (multiples of 8) through the TILED slab kernel alone (rdgan_upconv16t.hip.h):
// x [B,12,H,W,128] -> y [B,24,2H,2W,64], rinv [B,24,2H,2W]; dbg (optional, [B*24*2H*2W][4]) as above.
extern "C" int rdgan_op_upconv_slab_t16(const float* x, const float* w, const float* bias, float* y, float* rinv, float* dbg, int B,
                                        int H, int W, void* stream) {
  if (!x || !w || !bias || !y || !rinv || B < 1 || H < 8 || W < 8 || (H & 7) || (W & 7)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long nx = (long)B * 12 * H * W * 128, ny = (long)B * 24 * 4 * H * W * 64;
  if (12L * H * W * 256 >= 0x7FFFFFF0L) return -2;
  void *xb = nullptr, *yb = nullptr, *wi = nullptr; float* wc = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc(&wi, 64L * 8 * 2 * 64 * 16);
  if (rc == 0) rc = (int)hipMalloc((void**)&wc, 64L * 128 * 64 * sizeof(float));
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) {
    hipLaunchKernelGGL(k_collapse_weights, dim3(ew_blocks(16L * 128 * 64)), dim3(256), 0, st, w, wc, 128 * 64);
    hipLaunchKernelGGL(k_upconv_wimg_t, dim3(256), dim3(256), 0, st, wc, (unsigned short*)wi);
    rc = ensure_lds(nullptr, (const void*)k_upconv_slab_t16<false, true>, RD_UPT_LDS);
  }
  if (rc == 0) {
    const long items = (long)B * 6 * (H / 8) * (W / 8);
    hipLaunchKernelGGL((k_upconv_slab_t16<false, true>), dim3((unsigned)std::min<long>(items, 512)), dim3(256), RD_UPT_LDS, st, (const rd_bf16_t*)xb,
                       (const rd_bf16_t*)wi, bias, (rd_bf16_t*)yb, rinv, B, H, W, dbg, (const unsigned short*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(ew_blocks(ny)), dim3(256), 0, st, (const rd_bf16_t*)yb, y, ny);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, wi, (void*)wc}) if (p) (void)hipFree(p);
  return rc;
}

// Collapsed weight gradient of generator block 3 through the slab kernel alone (rdgan_upwgrad16.hip.h), ndomain 16: x [B,12,8,8,128]
// and dy [B,24,16,16,64] are rounded to bf16 on the device; dWc [64 = phase*8 + tap][128][64] fp32.
extern "C" int rdgan_op_upconv_wgrad_slab16(const float* x, const float* dy, float* dWc, int B, void* stream) {
  if (!x || !dy || !dWc || B < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long nx = (long)B * 12 * 64 * 128, ny = (long)B * 24 * 256 * 64;
  const int G = 6 * B >= 64 ? 32 : 8;
  void *xb = nullptr, *yb = nullptr; float* part = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc((void**)&part, (size_t)G * 64 * RD_UWG_TILE * sizeof(float));
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, dy, yb, ny, st);
  if (rc == 0) rc = ensure_lds(nullptr, (const void*)k_upconv_wgrad_slab16, RD_UWG_LDS);
  if (rc == 0) {
    hipLaunchKernelGGL(k_upconv_wgrad_slab16, dim3(8 * G), dim3(512), RD_UWG_LDS, st, (const rd_bf16_t*)xb, (const rd_bf16_t*)yb, part, B, G);
    hipLaunchKernelGGL(k_upconv_wgrad_fold, dim3(64 * RD_UWG_TILE / 4 / 256), dim3(256), 0, st, part, G, dWc);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, (void*)part}) if (p) (void)hipFree(p);
  return rc;
}

// Forward of the critic's second layer through the slab kernel alone (rdgan_d2fwd16.hip.h), ndomain 16: x [B,11,7,7,64] and the layer's
// kernel w [3,3,3,64,128] are rounded to bf16 on the device; y [B,6,4,4,128] = dropout(LeakyReLU(conv(x; stride 2, 'same') + bias))
// comes back as fp32 (the bf16 output widened); seed = 0: no dropout, otherwise the layer-2 mask of `seed` (counter = flat index of y).
extern "C" int rdgan_op_d2_fwd_slab16(const float* x, const float* w, const float* bias, float* y, int B, uint64_t seed, void* stream) {
  if (!x || !w || !bias || !y || B < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long nx = (long)B * 539 * 64, ny = (long)B * 96 * 128;
  void *xb = nullptr, *yb = nullptr, *wi = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc(&wi, (long)RD_D2F_KSTEPS * 4 * 64 * 16);
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) {
    hipLaunchKernelGGL(k_d2f_wimg, dim3(RD_D2F_KSTEPS), dim3(256), 0, st, w, (unsigned short*)wi);
    rc = ensure_lds(nullptr, (const void*)k_d2_fwd_slab16, RD_D2F_LDS);
  }
  if (rc == 0) {
    hipLaunchKernelGGL(k_d2_fwd_slab16, dim3((unsigned)std::min(B, 512)), dim3(256), RD_D2F_LDS, st, (const rd_bf16_t*)xb,
                       (const rd_bf16_t*)wi, bias, (rd_bf16_t*)yb, B, seed != 0, rd_make_key(seed, RD_STREAM_D1 + 1), 0u);
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(ew_blocks(ny)), dim3(256), 0, st, (const rd_bf16_t*)yb, y, ny);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, wi}) if (p) (void)hipFree(p);
  return rc;
}

// Weight gradient of the critic's third layer through the slab kernel alone (rdgan_d3wgrad16.hip.h), ndomain 16: x [B,6,4,4,128]
// (layer 2's output) and dy [B,3,2,2,256] are rounded to bf16 on the device; dW [3,3,3,128,256] fp32.
extern "C" int rdgan_op_d3_wgrad_slab16(const float* x, const float* dy, float* dW, int B, void* stream) {
  if (!x || !dy || !dW || B < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long nx = (long)B * 96 * 128, ny = (long)B * 12 * 256;
  const int G = B >= 256 ? 16 : 8;
  void *xb = nullptr, *yb = nullptr; float* part = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc((void**)&part, (size_t)G * 27 * RD_D3W_TILE * sizeof(float));
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, dy, yb, ny, st);
  if (rc == 0) rc = ensure_lds(nullptr, (const void*)k_d3_wgrad_slab16, RD_D3W_LDS);
  if (rc == 0) {
    hipLaunchKernelGGL(k_d3_wgrad_slab16, dim3(16 * G), dim3(512), RD_D3W_LDS, st, (const rd_bf16_t*)xb, (const rd_bf16_t*)yb, part, B, G);
    hipLaunchKernelGGL(k_d3_wgrad_fold, dim3((27 * RD_D3W_TILE / 4 + 255) / 256), dim3(256), 0, st, part, G, dW);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, (void*)part}) if (p) (void)hipFree(p);
  return rc;
}

// Weight gradient of the critic's second layer through the slab kernel alone (rdgan_d2wgrad16.hip.h), ndomain 16: x [B,11,7,7,64]
// (layer 1's output) and dy [B,6,4,4,128] are rounded to bf16 on the device; dW [3,3,3,64,128] fp32.
extern "C" int rdgan_op_d2_wgrad_slab16(const float* x, const float* dy, float* dW, int B, void* stream) {
  if (!x || !dy || !dW || B < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long nx = (long)B * 539 * 64, ny = (long)B * 96 * 128;
  const int G = B >= 64 ? 64 : 8;
  void *xb = nullptr, *yb = nullptr; float* part = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc((void**)&part, (size_t)G * 27 * RD_D2W_TILE * sizeof(float));
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, dy, yb, ny, st);
  if (rc == 0) rc = ensure_lds(nullptr, (const void*)k_d2_wgrad_slab16, RD_D2W_LDS);
  if (rc == 0) {
    hipLaunchKernelGGL(k_d2_wgrad_slab16, dim3(4 * G), dim3(512), RD_D2W_LDS, st, (const rd_bf16_t*)xb, (const rd_bf16_t*)yb, part, B, G);
    hipLaunchKernelGGL(k_d2_wgrad_fold, dim3((27 * RD_D2W_TILE / 4 + 255) / 256), dim3(256), 0, st, part, G, dW);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, (void*)part}) if (p) (void)hipFree(p);
  return rc;
}

// The same through the tiled kernel (k_d2_wgrad_slab_t16): x [B,11,2 OH - 1,2 OW - 1,64], dy [B,6,OH,OW,128], OH and OW multiples of 4.
extern "C" int rdgan_op_d2_wgrad_slab_t16(const float* x, const float* dy, float* dW, int B, int OH, int OW, void* stream) {
  if (!x || !dy || !dW || B < 1 || OH < 4 || OW < 4 || OH % 4 || OW % 4) return -2;
  hipStream_t st = (hipStream_t)stream;
  const RdD2wGeom geo = {2 * OH - 1, 2 * OW - 1, OH, OW, OH / 4, OW / 4};
  const long nx = (long)B * 11 * geo.IH * geo.IW * 64, ny = (long)B * 6 * OH * OW * 128;
  const int G = (long)B * geo.TH * geo.TW >= 64 ? 64 : 8;
  void *xb = nullptr, *yb = nullptr; float* part = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc((void**)&part, (size_t)G * 27 * RD_D2W_TILE * sizeof(float));
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, dy, yb, ny, st);
  if (rc == 0) rc = ensure_lds(nullptr, (const void*)k_d2_wgrad_slab_t16, RD_D2WT_LDS);
  if (rc == 0) {
    hipLaunchKernelGGL(k_d2_wgrad_slab_t16, dim3(4 * G), dim3(512), RD_D2WT_LDS, st, (const rd_bf16_t*)xb, (const rd_bf16_t*)yb, part, B, G, geo);
    hipLaunchKernelGGL(k_d2_wgrad_fold, dim3((27 * RD_D2W_TILE / 4 + 255) / 256), dim3(256), 0, st, part, G, dW);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, (void*)part}) if (p) (void)hipFree(p);
  return rc;
}

// Input gradient of the critic's second layer through the slab kernel alone (rdgan_d2slab16.hip.h), ndomain 16: gy [B,6,4,4,128],
// the layer's kernel w [3,3,3,64,128] and the gating activation aux [B,11,7,7,64] (layer 1's output) are rounded to bf16 on the
// device; gx [B,11,7,7,64] = conv3d_input_grad(gy, w; stride 2, 'same') * gate(aux) comes back as fp32 (the kernel's bf16 output
// widened).  gate = LeakyReLU'(aux); with use_drop, aux is a stored post-dropout activation (rd_drop_apply: +0.0 = dropped) and
// gate = 0 for dropped elements, LeakyReLU'(aux) / 0.75 for kept ones (rd_gate_from_out).
extern "C" int rdgan_op_d2_dgrad_slab16(const float* gy, const float* w, const float* aux, float* gx, int B, int use_drop, void* stream) {
  if (!gy || !w || !aux || !gx || B < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long ny = (long)B * RD_D2S_SROWS * 128, nx = (long)B * RD_D2S_OPOS * 64;
  void *yb = nullptr, *ab = nullptr, *xb = nullptr, *wi = nullptr;
  int rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc(&ab, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&wi, (long)RD_D2S_KSTEPS * 2 * 64 * 16);
  if (rc == 0) rc = launch_to_bf16(nullptr, gy, yb, ny, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, aux, ab, nx, st);
  if (rc == 0) {
    hipLaunchKernelGGL(k_d2s_wimg, dim3((RD_D2S_KSTEPS * 2 * 64 + 255) / 256), dim3(256), 0, st, w, (unsigned short*)wi);
    rc = ensure_lds(nullptr, (const void*)k_d2_dgrad_slab16, RD_D2S_LDS);
  }
  if (rc == 0) {
    hipLaunchKernelGGL(k_d2_dgrad_slab16, dim3((unsigned)std::min((B + 1) / 2, 512)), dim3(256), RD_D2S_LDS, st, (const rd_bf16_t*)yb,
                       (const rd_bf16_t*)wi, (const rd_bf16_t*)ab, (rd_bf16_t*)xb, B, use_drop != 0);
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(ew_blocks(nx)), dim3(256), 0, st, (const rd_bf16_t*)xb, gx, nx);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {yb, ab, xb, wi}) if (p) (void)hipFree(p);
  return rc;
}

// The same through the TILED kernel (k_d2_dgrad_slab_t16): gy [B,6,OH,OW,128], aux / gx [B,11,2 OH - 1,2 OW - 1,64]; OH, OW multiples of 4.
extern "C" int rdgan_op_d2_dgrad_slab_t16(const float* gy, const float* w, const float* aux, float* gx, int B, int OH, int OW, int use_drop,
                                          void* stream) {
  if (!gy || !w || !aux || !gx || B < 1 || OH < 4 || OW < 4 || (OH & 3) || (OW & 3)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long ny = (long)B * 6 * OH * OW * 128, nx = (long)B * 11 * (2 * OH - 1) * (2 * OW - 1) * 64;
  if (nx * 2 >= 0x7FFFFFF0L) return -2;
  void *yb = nullptr, *ab = nullptr, *xb = nullptr, *wi = nullptr;
  int rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc(&ab, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&wi, (long)RD_D2S_KSTEPS * 2 * 64 * 16);
  if (rc == 0) rc = launch_to_bf16(nullptr, gy, yb, ny, st);
  if (rc == 0) rc = launch_to_bf16(nullptr, aux, ab, nx, st);
  if (rc == 0) rc = (int)hipMemsetAsync(xb, 0xFF, nx * 2, st);          // (NaN pattern: every destination must be written)
  if (rc == 0) {
    hipLaunchKernelGGL(k_d2s_wimg, dim3((RD_D2S_KSTEPS * 2 * 64 + 255) / 256), dim3(256), 0, st, w, (unsigned short*)wi);
    rc = ensure_lds(nullptr, (const void*)k_d2_dgrad_slab_t16, RD_D2T_LDS);
  }
  if (rc == 0) {
    const long items = (long)((B + 1) / 2) * (OH / 4) * (OW / 4);
    hipLaunchKernelGGL(k_d2_dgrad_slab_t16, dim3((unsigned)std::min<long>(items, 512)), dim3(256), RD_D2T_LDS, st, (const rd_bf16_t*)yb,
                       (const rd_bf16_t*)wi, (const rd_bf16_t*)ab, (rd_bf16_t*)xb, B, OH, OW, use_drop != 0, (const unsigned char*)nullptr);
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(ew_blocks(nx)), dim3(256), 0, st, (const rd_bf16_t*)xb, gx, nx);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {yb, ab, xb, wi}) if (p) (void)hipFree(p);
  return rc;
}

// Generator block 2 forward of the bf16 storage mode through the slab kernel alone (rdgan_upconv16b.hip.h): x [B,6,4,4,256] and the
// Conv3D kernel w [3,3,3,256,128] are rounded to bf16 on the device (the kernel after the upsample collapse), y [B,12,8,8,128] =
// LeakyReLU(PixelNorm(upconv(x) + bias)) comes back as fp32 (the kernel's bf16 output widened), rinv [B,12,8,8].
extern "C" int rdgan_op_upconv2_slab16(const float* x, const float* w, const float* bias, float* y, float* rinv, int B, void* stream) {
  if (!x || !w || !bias || !y || !rinv || B < 1) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long nx = (long)B * 96 * 256, ny = (long)B * 768 * 128;
  void *xb = nullptr, *yb = nullptr, *wi = nullptr; float* wc = nullptr;
  int rc = (int)hipMalloc(&xb, nx * 2);
  if (rc == 0) rc = (int)hipMalloc(&yb, ny * 2);
  if (rc == 0) rc = (int)hipMalloc(&wi, (long)RD_UP2_KSTEPS * 4 * 64 * 16);
  if (rc == 0) rc = (int)hipMalloc((void**)&wc, 64L * 256 * 128 * sizeof(float));
  if (rc == 0) rc = launch_to_bf16(nullptr, x, xb, nx, st);
  if (rc == 0) {
    hipLaunchKernelGGL(k_collapse_weights, dim3(ew_blocks(16L * 256 * 128)), dim3(256), 0, st, w, wc, 256 * 128);
    hipLaunchKernelGGL(k_upconv2_wimg, dim3(RD_UP2_KSTEPS), dim3(256), 0, st, wc, (unsigned short*)wi);
    rc = ensure_lds(nullptr, (const void*)k_upconv2_slab16, RD_UP2_LDS);
  }
  if (rc == 0) {
    hipLaunchKernelGGL(k_upconv2_slab16, dim3((unsigned)std::min(B, 256 * RD_UP2_WGS)), dim3(256), RD_UP2_LDS, st, (const rd_bf16_t*)xb,
                       (const rd_bf16_t*)wi, bias, (rd_bf16_t*)yb, rinv, B);
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(ew_blocks(ny)), dim3(256), 0, st, (const rd_bf16_t*)yb, y, ny);
    rc = (int)hipGetLastError();
  }
  if (rc == 0) rc = (int)hipStreamSynchronize(st);
  for (void* p : {xb, yb, wi, (void*)wc}) if (p) (void)hipFree(p);
  return rc;
}

extern "C" int rdgan_op_pixelnorm_lrelu(const float* y, float* hout, float* rinv, long npix, int C, void* stream) {
  if (!y || !hout) return -2;
  RD_TRY(launch_pn_fwd(nullptr, y, hout, rinv, npix, C, (hipStream_t)stream));
  return (int)hipStreamSynchronize((hipStream_t)stream);
}
extern "C" int rdgan_op_pixelnorm_lrelu_bwd(const float* gh, const float* hh, const float* rinv, float* dy, long npix, int C,
                                            void* stream) {
  if (!gh || !hh || !rinv || !dy) return -2;
  RD_TRY(launch_pn_bwd(nullptr, gh, hh, rinv, dy, npix, C, 0, 0, 0, 0, (hipStream_t)stream));
  return (int)hipStreamSynchronize((hipStream_t)stream);
}

// test hook: the activations the last forward left in the workspace (generator h0..h3: which = 0..3, critic layers 1..4
// after LeakyReLU and dropout: which = 4..7), first n floats, as fp32
__global__ void k_bytes_to_f32(const unsigned char* in, float* out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
extern "C" int rdgan_debug_activation(rdgan_handle* h, int which, float* out, long n, void* stream) {
  if (h && out && which == 8 && n >= 1) {      // test hook: layer 1's packed gate bytes (16 per row) as floats
    if (!h->g1bits || n > (long)h->NB * h->dL[1] * 16) return bad_arg(h, "debug_activation: no gate bytes");
    hipLaunchKernelGGL(k_bytes_to_f32, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, h->g1bits, out, n);
    RD_CHECK(h, hipGetLastError());
    return 0;
  }
  if (!h || !out || which < 0 || which > 7 || n < 1) return bad_arg(h, "debug_activation: bad argument");
  const float* src; long cap;
  if (which < 4) {
    float* hs[4] = {h->h0, h->h1, h->h2, h->h3};
    src = hs[which]; cap = (long)h->MB * h->gpix[which] * h->gch[which];
  } else {
    const int l = which - 3;
    src = h->dh[l]; cap = (long)h->NB * h->dL[l] * h->dch[l];
  }
  if (n > cap) return bad_arg(h, "debug_activation: n exceeds the tensor");
  hipStream_t st = (hipStream_t)stream;
  auto copy = [&](const void* from, long first, long cnt) -> int {     // `cnt` elements of `from` -> out[first ..)
    if (cnt <= 0) return 0;
    if (h->a16) {
      hipLaunchKernelGGL(k_bf16_to_f32, dim3(ew_blocks(cnt)), dim3(256), 0, st, (const rd_bf16_t*)from, out + first, cnt);
      RD_CHECK(h, hipGetLastError());
    } else
      RD_CHECK(h, hipMemcpyAsync(out + first, from, sizeof(float) * cnt, hipMemcpyDeviceToDevice, st));
    return 0;
  };
  RD_TRY(copy(src, 0, n));
  if (which >= 4 && h->keep_gates && h->gate_keep_B > 0) {
    // the last call was a critic step: the x_hat third of this layer as it was before the second sweep overwrote it
    const int l = which - 3;
    const long per = h->dL[l] * h->dch[l], first = 2L * h->gate_keep_B * per;
    RD_TRY(copy(h->gate_keep[l], first, std::min<long>(n, first + h->gate_keep_B * per) - first));
  }
  return 0;
}

__global__ void k_rng_probe(uint32_t key, float* mask, float* uni, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (mask) mask[i] = rd_drop_scale(key, (uint32_t)i);
    if (uni) uni[i] = rd_uniform(key, (uint32_t)i);
  }
}
extern "C" int rdgan_op_rng(uint64_t seed, uint32_t stream_id, float* mask_out, float* uniform_out, long n, void* stream) {
  hipLaunchKernelGGL(k_rng_probe, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, rd_make_key(seed, stream_id),
                     mask_out, uniform_out, n);
  return (int)hipStreamSynchronize((hipStream_t)stream);
}

// ------------------------------------------------------------------------------------
// input pipeline
// ------------------------------------------------------------------------------------
extern "C" int rdgan_data_gather(const float* data, int n_days, int ny, int nx, const int* indices, int n, int ndomain,
                                 float norm_scale, float* batch_out, float* cond_out, int* flags, void* stream) {
  if (!data || !indices || !cond_out || !flags || n < 1 || ndomain < 1 || ndomain > ny || ndomain > nx) return -2;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(flags, 0, sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_gather_tiles, dim3(ew_blocks((long)n * ndomain * ndomain)), dim3(256), 0, st, data, n_days, RDGAN_NHOURS,
                     ny, nx, indices, n, ndomain, norm_scale, batch_out, cond_out, flags);
  return (int)hipGetLastError();
}

extern "C" int rdgan_data_valid_tiles(const float* data, int n_days, int ny, int nx, int ndomain, int stride,
                                      float tp_thresh_daily, int n_thresh, int* valid_out, void* stream) {
  if (!data || !valid_out || n_days < 1 || stride < 1 || ndomain < 1) return -2;
  const int nbi = (ny - ndomain + stride - 1) / stride, nbj = (nx - ndomain + stride - 1) / stride;   // len(range(0, ny-nd, stride))
  if (nbi < 1 || nbj < 1) return 0;
  const long blocks = (long)n_days * nbi * nbj;
  if (blocks > 0x7FFFFFFFL) return -2;
  hipLaunchKernelGGL(k_valid_tiles, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, data, RDGAN_NHOURS, ny, nx,
                     ndomain, stride, nbi, nbj, tp_thresh_daily, n_thresh, valid_out);
  return (int)hipGetLastError();
}

extern "C" int rdgan_crps_ensemble(const float* ens, const float* obs, const float* scale, float* crps_out, int n,
                                   long npix, void* stream) {
  if (!ens || !obs || !crps_out || n < 1 || n > 8192 || npix < 1 || npix > 0x7FFFFFFFL) return -2;
  int npow2 = 2;
  while (npow2 < n) npow2 <<= 1;
  hipLaunchKernelGGL(k_crps_ensemble, dim3((unsigned)npix), dim3(256), npow2 * sizeof(float), (hipStream_t)stream, ens, obs,
                     scale, crps_out, n, npow2, npix);
  return (int)hipGetLastError();
}

#ifdef RD_STAMP
extern "C" int rdgan_debug_stamps(unsigned long long* out, int reset) {
  unsigned long long z[8] = {0};
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && out) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(rd_stamp_acc), sizeof(z));
  if (e == hipSuccess && reset) e = hipMemcpyToSymbol(HIP_SYMBOL(rd_stamp_acc), z, sizeof(z));
  return (int)e;
}
extern "C" int rdgan_debug_stamps_ws(unsigned long long* out, int reset) {
  unsigned long long z[8] = {0};
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && out) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(rd_stamp_ws), sizeof(z));
  if (e == hipSuccess && reset) e = hipMemcpyToSymbol(HIP_SYMBOL(rd_stamp_ws), z, sizeof(z));
  return (int)e;
}
#endif
