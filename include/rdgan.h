/* C ABI of librdgan_hip.so -- the MI355X-native (gfx950) cWGAN-GP hot path of RainDisaggGAN.
 *
 * The reference (sipposip/pr-disagg-radar-gan) has no native boundary: its hot path is a set of
 * Keras calls into TensorFlow 2.1.  Each entry point below names the reference call it replaces
 * (T = gan_train_cwgangp_pixelnorm.py, P = raindisagg_gan_pretrained.py).  The binding a
 * maintainer would add on the reference side is the ctypes stub in INTEGRATION.md /
 * pr_disagg_radar_gan_amd/_lib.py.
 *
 * Conventions: plain C, no torch types.  Every pointer is a DEVICE pointer owned by the caller
 * (fp32, 16-byte aligned); `stream` is a hipStream_t passed as void*; calls are asynchronous on
 * that stream and never allocate (the workspace is sized at rdgan_create).  Return value: 0 = ok,
 * >0 = hipError_t, -2 = bad argument.  One handle per device and per stream user; a handle is not
 * thread-safe.  Layouts are the reference's: activations NDHWC, Conv3D kernels
 * (kd,kh,kw,Cin,Cout), Dense kernels (in,out); parameter slabs are the Keras weights concatenated
 * in model.get_weights() order (kernel, bias per layer; see rdgan_*_param_layout).
 */
#ifndef RDGAN_H
#define RDGAN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rdgan_handle rdgan_handle;

#define RDGAN_LATENT_DIM 100   /* T:69 */
#define RDGAN_NHOURS 24        /* T:134 */
#define RDGAN_LOSS_SLOTS 8     /* floats appended behind the gradients in a gradient slab */

/* Build the per-(ndomain, max_batch) plans and allocate the activation workspace.
 * Replaces model construction at T:361-362 (create_generator / create_discriminator).
 * ndomain must be a multiple of 8 (L:324); n_cond_channels = 1 (T:129: the daily sum), 2 (+ longitude index,
 * revision1/additional_inputs/gan_train_cwgangp_pixelnorm_lon.py:136) or 3 (+ sin/cos day of year, …_doy.py:135):
 * every `cond` below is then [B,nd,nd,n_cond_channels], the generator's Dense has 100 + nd*nd*n_cond_channels
 * inputs and the critic's first Conv3D 1 + n_cond_channels input channels. */
int rdgan_create(rdgan_handle** out, int ndomain, int n_cond_channels, int max_batch);
void rdgan_destroy(rdgan_handle* h);
const char* rdgan_last_error(const rdgan_handle* h);
long rdgan_gen_param_count(const rdgan_handle* h);      /* 3 974 273 at ndomain 16 */
long rdgan_critic_param_count(const rdgan_handle* h);   /* 2 880 065 at ndomain 16 */
long rdgan_workspace_bytes(const rdgan_handle* h);

/* generator.predict([latent, cond]) (P:60, T:205,212; graph T:312-357).
 * z [B,100], cond [B,nd,nd,1] normalised daily sums -> out [B,24,nd,nd,1] hourly fractions. */
int rdgan_gen_forward(rdgan_handle* h, const float* gen_params, const float* z, const float* cond,
                      float* out, int B, void* stream);

/* tf.debugging.check_numerics(..., 'generator has nans or infs') behind the generator's softmax (T:349-350): waits for
 * `stream`, then returns -1 (and sets rdgan_last_error) if the generator output of the calls issued since the last
 * rdgan_gen_forward / rdgan_critic_grad / rdgan_gen_grad began contained NaN or Inf, else 0.  The gradient entries also
 * fold the same flag into slot 4 (nonfinite_flag) of their loss tail, so a training loop needs no extra sync (T:487-488). */
int rdgan_check_numerics(rdgan_handle* h, void* stream);

/* critic.predict([sample, cond]) (graph T:272-309).  seed == 0: dropout off (inference);
 * seed != 0: dropout masks of a train_on_batch pass (streams D1..D4 of rdgan_rng.h). out [B,1]. */
int rdgan_critic_forward(rdgan_handle* h, const float* critic_params, const float* sample,
                         const float* cond, float* out, int B, uint64_t seed, void* stream);

/* Gradient half of critic_model.train_on_batch([X_real, cond_real, latent], [valid, fake, dummy])
 * (T:472; graph T:363-392): generator forward (frozen), RandomWeightedAverage, three critic passes
 * as one 3B batch, the gradient-penalty double backward, loss = mean(-D(x)) + mean(D(G(z))) +
 * 10*mean((||grad_xhat D(xhat)||-1)^2).  grad_out[0:n_critic_params] = d loss / d critic weights,
 * grad_out[n .. n+8) = {total, valid, fake, gp, nonfinite_flag, 0, 0, 0}.  The caller all-reduces
 * grad_out over ranks (RCCL) and then calls rdgan_adam. */
int rdgan_critic_grad(rdgan_handle* h, const float* critic_params, const float* gen_params,
                      const float* x_real, const float* cond, const float* z, uint64_t seed,
                      float* grad_out, int B, void* stream);

/* The same, for callers that update the critic on another stream (data-parallel training: gradient all-reduce + Adam
 * on a communication stream).  The frozen generator's forward reads no critic weight, so it is issued first; `stream`
 * then waits for `critic_ready_event` (a hipEvent_t the caller recorded behind its last write of critic_params; NULL =
 * no wait) before the first kernel that reads critic_params.  The previous update thus overlaps the generator forward. */
int rdgan_critic_grad_after(rdgan_handle* h, const float* critic_params, const float* gen_params,
                            const float* x_real, const float* cond, const float* z, uint64_t seed,
                            float* grad_out, int B, void* critic_ready_event, void* stream);

/* Gradient half of generator_model.train_on_batch([latent, cond], valid) (T:482; graph T:395-408):
 * loss = mean(-D(G(z,c))), critic frozen, its dropout active.
 * grad_out[0:n_gen_params], grad_out[n .. n+8) = {loss, 0, 0, 0, nonfinite_flag, 0, 0, 0}. */
int rdgan_gen_grad(rdgan_handle* h, const float* critic_params, const float* gen_params,
                   const float* z, const float* cond, uint64_t seed, float* grad_out, int B,
                   void* stream);

/* rdgan_gen_grad with the same late wait for the critic weights (see rdgan_critic_grad_after): the generator forward
 * runs in front of it. */
int rdgan_gen_grad_after(rdgan_handle* h, const float* critic_params, const float* gen_params,
                         const float* z, const float* cond, uint64_t seed, float* grad_out, int B,
                         void* critic_ready_event, void* stream);

/* tf.optimizers.Adam(lr, beta_1=0, beta_2) apply step (T:385): v = b2 v + (1-b2) g^2,
 * p -= lr*sqrt(1-b2^t) * g / (sqrt(v)+eps), g = grad*grad_scale (1/world after the all-reduce sum).
 * t = the optimizer's shared iteration counter after increment (both models share it, T:391,408). */
int rdgan_adam(float* params, const float* grad, float* v, long n, int t, float lr, float beta2,
               float eps, float grad_scale, void* stream);

/* Weight-form cache.  Every gradient / forward entry first derives "weight forms" from the parameter slabs it is given
 * (generator: the transposed last kernel and the collapsed or shared-centre forms of the three block kernels, plus their bf16
 * images in the storage mode; critic: transposed kernels, padded / bf16 images) -- weight-only kernels that are wasted work
 * while a network is frozen: the generator across the n_critic critic steps of an iteration (T:468-476), the critic across
 * the generator step and the critic step that follows it (T:482, next T:472).  The caller may vouch for the CONTENT of the
 * slabs it passes: versions set here hold for the following calls, and a call whose (slab pointer, version, form options)
 * equal those the forms in the workspace were built from skips the rebuild.  A version is any non-zero number the caller
 * changes whenever it writes the slab (the trainer: a fresh number after every Adam update or checkpoint load); 0 = unknown,
 * always rebuild (the default, and what a caller that cannot vouch passes).  Results are bit-identical either way.
 * rdgan_form_builds: how many times each network's forms have been built by this handle (tests). */
int rdgan_set_weight_versions(rdgan_handle* h, uint64_t gen_version, uint64_t critic_version);
int rdgan_form_builds(const rdgan_handle* h, long* gen_builds, long* critic_builds);

/* Parameter slab layout: fills offsets[0..10) / sizes with the element offset and size of
 * {kernel,bias} x 5 layers in Keras weight order; returns the number of tensors (10). */
int rdgan_gen_param_layout(const rdgan_handle* h, long* offsets, long* sizes);
int rdgan_critic_param_layout(const rdgan_handle* h, long* offsets, long* sizes);

/* Options.  "collapse" (default 1): evaluate each generator block UpSampling3D(2)+Conv3D(3x3x3,'same')
 * (T:330-331) as 8 parity phases of 2x2x2 taps on the un-upsampled grid with pre-summed weights --
 * algebraically identical, 3.375x fewer FLOPs in forward, input gradient and weight gradient; only the
 * fp32 rounding of the pre-summed weights differs.  0 = the reference's direct 27-tap form.
 * "wave_specialized" (default 1): big GEMMs run the producer/consumer kernel (4 loader waves streaming tiles
 * into LDS by DMA, 4 compute waves issuing only ds_read + MFMA); 0 = the single-role kernel everywhere;
 * 2 = producer/consumer kernel for every eligible shape regardless of size (tests).
 * "ws_ksplit" (default 1): mid-size producer/consumer launches whose workgroup count would leave part of the 256 CUs
 * idle in the last round split their K loop over 2..8 workgroups (partials folded by a finish kernel, fixed order);
 * 0 = never, n > 1 = force n-way splits wherever the shape allows (tests).
 * "fast_fwd" / "fast_bwd" (default -1 = by storage mode: on with fp32 storage, off in the bf16 storage mode, where the matrix
 * pipe is 16x faster and the plain collapsed form -- one GEMM per block, no difference / plane-sum passes -- is quicker;
 * 1 / 0 force it; need "collapse" 1): generator blocks 2 and 3 (block inputs with >= 6 hour
 * planes) in the shared-centre form along the hour axis: out[2s] = S x[s] - W0 E[s], out[2s+1] = S x[s] + W2 E[s+1]
 * with E[j] = x[j] - x[j-1] and S = W0+W1+W2, so both outputs of a source position share the S x[s] product -- 48
 * instead of 64 tap products per position, algebraically identical.  Forward: T = S x once per output plane pair,
 * then the difference part adds T in its epilogue; backward: plane-pair sums of the output gradient feed the S part,
 * the gradient wrt E is folded back by the adjoint of the differencing.  0 = the 64-tap collapsed form.
 * "mfma_bf16" (default 0; needs the shared-centre form): mixed mode for BASELINE configs 3-5.  The forward,
 * input-gradient, second-sweep and weight-gradient GEMMs of generator blocks 1-3 and critic layers 2-4 read bf16
 * copies of their operands (weight forms stored [N][K]) and run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation;
 * every tensor the caller or another kernel sees, all gradients and the optimizer state stay fp32.  Results move by
 * bf16 rounding (2^-9 per operand), so it is NOT the mode the fp32 metric is measured in.
 * "g9_direct" (default 1): backward of the last generator conv (64 -> 1) straight from the 1-channel dlogits with the
 * neighbouring hour planes in LDS: weight gradient without the [rows][27] im2col matrix, input gradient fused with
 * block 3's PixelNorm+LeakyReLU backward (no intermediate gradient tensor).  Needs 4 (nd+2)^2 floats of LDS (nd <= 72);
 * 0 (and larger domains) = im2col + column GEMMs + separate PixelNorm backward.
 * "resident" (default 1; bf16 storage mode): the forward GEMMs of the shared-centre form whose tile holds whole source
 * planes keep the tile's source rows resident in LDS and stream only the weights (each source row is fetched once per
 * channel chunk instead of once per tap and chunk); same arithmetic in the same order, bit-identical to 0.
 * "upconv_slab" (default 1; bf16 storage mode, ndomain 16, collapsed form): the forward of generator block 3 (128 -> 64
 * channels onto the 24 x 16 x 16 grid, the dominant launch of the mode) runs in the slab kernel k_upconv_slab16: two source hour
 * planes + their halo resident in LDS for all 8 phases x 8 taps, weights streamed global -> VGPR in MFMA-fragment order, bias +
 * PixelNorm + LeakyReLU + bf16 rounding in registers (rdgan_upconv16.hip.h).  Same products as the streaming GEMM it replaces,
 * summed in another order: outputs agree to one bf16 ulp.  0 = the streaming GEMM (k_conv_gemm_ws<256, 64, ..., bf16>).
 * "upconv2_slab" (default 1; same conditions): the forward of generator block 2 (256 -> 128 channels onto the 12 x 8 x 8 grid) in the
 * slab kernel k_upconv2_slab16: a sample's whole block input (48 KB) resident in LDS for all 8 phases x 8 taps, the four waves of a
 * workgroup split the 128 output channels (each streams its own weight fragments) and exchange the PixelNorm row sums of squares
 * through LDS once per phase (rdgan_upconv16b.hip.h).  0 = the streaming GEMM (k_conv_gemm_ws<128, 128, ..., bf16>).
 * "g9_fused" (default 1; with "upconv_slab" and "tapgather"): the tap products of the generator's last Conv3D (64 -> 1, T:345) are
 * formed in the epilogue of the block-3 slab kernel from the rows it holds in registers (four more MFMAs per 32 rows) and summed over
 * (kh, kw) and the hour taps inside the work item; k_tapsum_softmax12 adds the source parity classes and the neighbouring items and
 * takes the softmax.  No pass over block 3's output (k_g9_fwd); a critic step does not store that output at all.  Same bf16 products
 * as 0, another fp32 summation order: fractions agree to ~3e-7 of the largest.
 * "upconv_slab_t" (default 1; bf16 storage mode, collapsed form, ndomain > 16 whose block-3 source planes are multiples of 8 x 8:
 * 32, 48, 64, ... -- the large-domain variant L:355-358): block 3 forward in the TILED slab kernel k_upconv_slab_t16: (h, w) tiles of
 * 8 x 8 source positions with their halo resident in LDS, the K loop in two channel halves (rdgan_upconv16t.hip.h); with "g9_fused"
 * the last conv's tap products leave its epilogue too, the tiles' edge sums as halo terms (k_g9_halo_fold, k_tapsum_softmax12t).
 * One bf16 ulp from the streaming GEMM; fractions within 2e-5 of the separate pass.
 * "conv_f16" (default 1; bf16 storage mode): the gather GEMMs with N % 128 == 0 and a bf16 destination that launch >= 640 workgroups
 * of 256 x 128 -- generator blocks 1 / 2 forward and input gradients, critic layers 2-4 forward and input gradients where no slab
 * kernel takes them -- by k_conv_gemm_f16 (rdgan_gemm_f16.hip.h): four waves and no loader waves, weights global -> VGPR from
 * fragment-order twin images (written beside the [tap][N][K] ones), gathered rows by LDS-DMA one 64-k chunk ahead, per-wave epilogue.
 * Same chunk and k order as the streaming kernel: bit-identical results except through a fused PixelNorm (one ulp of 1/l2).
 * 0 = k_conv_gemm_ws everywhere; 2 (tests) = regardless of the launch size.
 * "wgrad_wide" (default 0; bf16 storage mode): the streaming weight gradients of N % 128 == 0 layers with >= 32768 gathered rows on
 * 256 x 128 tiles with three LDS stages, one workgroup per CU (k_wgrad_gemm_ws16<256, 128>).  Correct and SLOWER (1.6-2x): kept for
 * the record and its tests.
 * "dense_skinny" (default 1; bf16 storage mode with "dense16", handles of max_batch <= 128): the Dense layer by k_dense16_skinny --
 * weights and input rows streamed in MFMA-fragment order, no LDS (the large domain's 415 MB kernel at 3.4-3.7 TB/s).  One bf16 ulp.
 * "wgrad_boxes" (default 1; with "border_boxes"): the weight gradients of critic layers 2-4 that run in the streaming kernels (fp32
 * storage: all three; bf16 storage: the layers without a slab kernel) use the border-class boxes too: per-phase tile and split
 * counts, one fold per weight tap over every phase that lists it (k_wgrad_reduce_box).  Same products, same results to ~1e-9.
 * ("d1_dgrad_fused" at ndomain > 16, multiples of 16: the same one-pass layer-1 input gradient on tiles of 24 x 16 x 8 input voxels,
 * k_d1_dgrad_tile16; bit-identical to the column GEMM + col2im.)
 * "d1_fwd_sample" (default 1; bf16 storage mode, ndomain 16, one condition channel): forward and second sweep of the critic's first
 * layer with a sample's input volume resident in LDS (k_d1_fwd_sample16); the second sweep then takes its gate from the 2-bit codes
 * ("d2_gate_bits"; without them it keeps the tile kernel).  One bf16 ulp from 0 in ~4e-5 of the activations.
 * "dense16" (default 1; bf16 storage mode): the generator's Dense layer on the bf16 matrix pipe: input rows and kernel rounded to
 * bf16, K padded to a multiple of 64, three launches of a third of the columns each (the streaming kernel wants N / 128 to be a power
 * of two).  0 = the fp32-pipe kernel with bf16 output.
 * "g9_bwd_mfma" (default 1; bf16 storage mode, collapsed backward): the input gradient of the generator's last conv + block 3's
 * PixelNorm backward on the fp32 matrix pipe (k_g9_bwd_mfma16: exact fp32 products); 0 = the VALU kernel k_g9_bwd_pairs.
 * "border_boxes" (default 1; 2 = at every size, 0 = off): the forward, second-sweep and input-gradient GEMMs of critic layers 2-4
 * (stride-2 'same' convs on 6x4x4 / 3x2x2 / 2x1x1 output grids) run on plans whose loop spaces are cut into border-class boxes,
 * each listing only the taps that can land inside the picture: 38 / 38 / 70 % fewer (row, tap) products at ndomain 16, all of them
 * products with a zero row.  Same products in the same tap order as the one-phase plans; small launches (few rows) keep the
 * one-phase plan, whose long K the K split needs.
 * "d2_gate_bits" (default 1; with "d2_slab" and the layer-1 edge kernels): the forward of layer 1 also writes its gate -- per element
 * LeakyReLU' = 1 or alpha, dropped or kept -- as 2 bits (16 bytes per row), and the slab kernel of layer 2's input gradient reads
 * that instead of layer 1's stored output (128 bytes per row): identical results.
 * "d2_fwd_slab" (default 0 -- measured no faster than the streaming GEMM, kept parity-tested; bf16 storage mode, ndomain 16): the forward of the critic's second layer in the slab kernel
 * k_d2_fwd_slab16: a sample's layer-1 output (69 KB) resident in LDS for all 27 taps, the four waves of a workgroup split the 128
 * output channels and stream their own weight fragments, bias + LeakyReLU + dropout in registers (rdgan_d2fwd16.hip.h); the
 * penalty's second sweep keeps the streaming GEMM.  Same dropout counter as 0 = k_conv_gemm_ws<128, 128, ..., bf16>.
 * ("d2_slab" at ndomain 32 / 48 / 64: k_d2_dgrad_slab_t16, the same on 8 x 8 tiles of destination positions of two samples.)
 * "d2_slab" (default 1; bf16 storage mode, ndomain 16): the input gradient of the critic's second layer (128 -> 64 channels onto
 * the 11 x 7 x 7 grid, eight parity phases of 8 ... 1 taps) runs in the slab kernel k_d2_dgrad_slab16: two samples' output
 * gradient resident in LDS for all phases and taps, weights streamed in MFMA-fragment order, LeakyReLU' x dropout gate + bf16
 * rounding in registers (rdgan_d2slab16.hip.h).  Same taps and k order as the streaming GEMM (0).
 * "d2_wgrad_slab" (default 1; bf16 storage mode, ndomain 16): the weight gradient of the critic's second layer runs in the slab
 * kernel k_d2_wgrad_slab16: a wave owns one of the 27 taps and keeps its 64 x 128 product in registers over the workgroup's share
 * of the batch; by input parity the taps fall into 8 classes, each a dense sub-grid of layer 1's output on which its taps are shifts
 * (rdgan_d2wgrad16.hip.h).  0 = k_wgrad_gemm_ws16<128,128>.  At ndomain 32 / 48 / 64 the same kernel body with work items = 4 x 4 tiles
 * of output positions and one halo position per odd class (k_d2_wgrad_slab_t16).
 * "d3_wgrad_slab" (default 1; same conditions): critic layer 3's weight gradient the same way (k_d3_wgrad_slab16: a wave owns a tap
 * and a quarter of the 256 output channels; a sample has 12 output positions, so an item is four samples); 0 = k_wgrad_gemm_ws16.
 * "upwgrad_slab" (default 1; bf16 storage mode, ndomain 16, collapsed form): the weight gradient of generator block 3 runs in the slab
 * kernel k_upconv_wgrad_slab16: a workgroup owns one output-parity phase and keeps its eight tap products (eight 128 x 64 fp32
 * tiles, one per wave) in registers over its share of the batch; source planes and output-gradient rows arrive by LDS-DMA in two
 * stages, both MFMA operands are read transposed from the position-major images (rdgan_upwgrad16.hip.h); block 2 the same way with a
 * workgroup per (phase, quarter of the 256 input channels) (k_upconv2_wgrad_slab16).  Both deliver the block's bias gradient from
 * the output-gradient fragments they multiply.  0 = k_wgrad_gemm_ws16 + column-sum passes.
 * "d1_dgrad_fused" (default 1; bf16 storage mode, ndomain 16, one condition channel): the first critic layer's input gradient
 * with respect to the sample channel (the penalty's dD/dx_hat and the generator step's dL/dfake) in one pass per sample
 * (k_d1_dgrad_sample16: the 539 x 27 tap products of a sample stay in LDS, the outputs gather from there); same sums in the same
 * order as 0 = column GEMM [rows][64] in HBM + k_d1_col2im.
 * "d1_wgrad16" (default 1; bf16 storage mode, one condition channel): the first critic layer's weight gradient runs on the bf16
 * matrix pipe (k_d1_wgrad16: im2col rows rounded to bf16 as in the layer's forward GEMM, both operands read transposed from
 * position-major LDS images) and delivers the layer's bias gradient from a ones column of the same product; 0 = the fp32-pipe
 * kernel k_d1_gemm_wgrad<bf16> + a column-sum pass.
 * "edge_kernels" (default 1): the weight gradient of the generator's last conv (64 -> 1) runs on the matrix pipe with the block-3
 * output streamed once (k_g9_wgrad_mfma; ndomain a power of two, otherwise the scalar kernel); the first critic layer (2 -> 64
 * channels, K = 54; one condition channel) runs as one K = 64 GEMM
 * per 128-row tile -- forward, the penalty's second sweep and the weight gradient (rdgan_edge.hip.h) -- instead of nine K
 * chunks of the tiled kernel; and in the bf16 storage mode the generator's last conv (64 -> 1) runs in its dedicated streaming
 * kernel (one pass over the block-3 output at HBM speed, same arithmetic and tap-sum format as "tapgather"); 2 = also with
 * fp32 storage (bit-identical to the tiled GEMM, not faster there); 0 = the tiled GEMM kernel everywhere.
 * "dense_wgrad_slices" (default 0 = by batch size; tests): row slices of the critic Dense weight gradient (1..16).
 * "split3" (default 0; optional data point, not the BASELINE metric): with fp32 storage, the forward / input-gradient conv GEMMs
 * of the producer/consumer kernel multiply on the bf16 matrix pipe -- every fp32 operand is split in registers into three bf16
 * parts (x = x1 + x2 + x3, 24 significant bits) and a product is the sum of six partial products, accumulated in fp32.  Memory,
 * LDS images, epilogues and the weight-gradient kernels are those of the fp32 path; errors against the fp64 oracle stay within
 * the fp32 path's tolerances (observed 1-3x its error).
 * "side_stream" (default 1): inside a call the weight-only kernels (generator weight forms, critic weight transposes and
 * bf16 images) and the bias-gradient column sums are issued on a second stream owned by the handle, beside the GEMMs on the
 * caller's stream and ordered against it by events (fork at entry, joins in front of the first reader / at the end of the
 * call): ~50 launches of 5-30 us per iteration leave the critical path.  The call's contract is unchanged (everything is
 * complete when `stream` is); same kernels on the same data, results bit-identical to 0.
 * "sample_offset" (default 0): global index of this rank's first sample.  RandomWeightedAverage's alpha (T:222-223) of
 * local sample k is uniform(key(seed, ALPHA), sample_offset + k), so ranks that share a seed draw the alphas of the
 * global batch (used by the data-parallel equivalence tests; the dropout masks stay keyed by the local element index).
 * "keep_gates" (default 0; test hook): a critic step keeps a copy of the interpolated third of its activations before the
 * penalty's second sweep overwrites it in place, so that rdgan_debug_activation returns the LeakyReLU / dropout pattern of all
 * 3B samples; allocates its buffers when set (not inside a step).
 * "tapgather" (default 1): the last generator conv (64 -> 1, T:345) runs as a column GEMM over its 27 taps whose
 * epilogue already sums the taps that fall inside the 256-row tile (ndomain 8/16: whole planes, 32/64/128: whole
 * rows), writing 3 or 9 floats per grid point instead of 32; 0 (and every other ndomain) = full column matrix +
 * separate gather kernel. */
int rdgan_set_option(rdgan_handle* h, const char* name, int value);

/* Per-kernel HIP-event timing for bench.py's roofline line.  tag_mask: bit i enables timing of
 * kernel class i (RDGAN_TAG_*).  rdgan_profile_read synchronises the device. */
enum {
  RDGAN_TAG_GCONV_FWD = 0,   /* fused upsample+Conv3D forward GEMMs of the generator */
  RDGAN_TAG_GCONV_DGRAD = 1,
  RDGAN_TAG_GCONV_WGRAD = 2,
  RDGAN_TAG_CRITIC_GEMM = 3,
  RDGAN_TAG_ELEMENTWISE = 4,
  RDGAN_TAG_GCONV3_FWD = 5,  /* the single dominant launch: forward of the 128->64 block at 24x16x16 (its difference part in the shared-centre form) */
  RDGAN_NUM_TAGS = 8
};
int rdgan_profile(rdgan_handle* h, unsigned tag_mask);
/* Algorithmic FLOPs (2 * rows * taps * K * N, of the algebraic forms actually run: collapsed / shared-centre) of every GEMM
 * this handle has launched since the last reset -- the numerator of bench.py's whole-iteration roofline fraction. */
int rdgan_flop_count(rdgan_handle* h, double* flops, int reset);
int rdgan_profile_read(rdgan_handle* h, int tag, double* total_ms, long* launches);
/* Per-launch table for bench.py's `roofline.launches`: with rdgan_profile_launches(h, 1) every GEMM launch (conv / input-gradient /
 * weight-gradient plans and the dedicated first- and last-layer kernels) is bracketed by HIP events on its launch stream and
 * recorded with its plan, kernel tile, batch and algorithmic FLOPs (2 * rows * taps * K * N of the form actually run; a launch
 * with split K includes its finish kernel, a weight gradient its partial-slab fold).  rdgan_launch_table synchronises the
 * device and returns one row per (plan, kind, batch, kernel): launches, summed GFLOP and summed milliseconds since the
 * recording was switched on.  kind: 0 = forward-type GEMM over the plan, 1 = weight gradient, 2 = dedicated edge kernel. */
typedef struct {
  int plan, kind, batch, launches;
  double gflop, ms;
  char name[48];       /* what the plan computes, e.g. "gen block3 fwd difference part" */
  char kernel[48];     /* kernel and tile, e.g. "k_conv_gemm_ws<256,64,TG4>" */
} rdgan_launch_stat;
int rdgan_profile_launches(rdgan_handle* h, int on);
int rdgan_launch_table(rdgan_handle* h, rdgan_launch_stat* out, int cap, int* n_out);

/* Input pipeline either side of the step (device-resident radar array data[n_days][24][ny][nx], fp32).
 * rdgan_data_gather: the tile gather + normalisation of generate_real_samples / generate_latent_points
 * (T:149-166, T:181-190): indices[n][3] = (tidx, yidx, xidx) int32 on the device; batch_out [n,24,nd,nd,1] =
 * hourly fractions of the daily sum (NULL: condition only), cond_out [n,nd,nd,1] = daily sum / norm_scale;
 * *flags |= 1 for a non-finite value (the reference asserts none, T:169-170), |= 2 for a fraction outside [0,1].
 * rdgan_data_valid_tiles: compute_valid_indices.py:74-92 -- valid_out[day][ii/stride][jj/stride] (int32 0/1) for the
 * boxes ii in range(0, ny-ndomain, stride), jj in range(0, nx-ndomain, stride): no NaN in the daily sum and at
 * least n_thresh points above tp_thresh_daily.  Bit-identical to the numpy forms. */
int rdgan_data_gather(const float* data, int n_days, int ny, int nx, const int* indices, int n, int ndomain,
                      float norm_scale, float* batch_out, float* cond_out, int* flags, void* stream);
int rdgan_data_valid_tiles(const float* data, int n_days, int ny, int nx, int ndomain, int stride,
                           float tp_thresh_daily, int n_thresh, int* valid_out, void* stream);

/* Ensemble CRPS per grid point, properscoring.crps_ensemble(obs, ens, axis=0) of generate_and_evaluate_crps.py:188:
 * ens [n][npix] (member-major), obs [npix], optional scale [npix] applied to the members first (fractions -> mm/h,
 * :186), crps_out [npix] = mean|x_i - y| - 0.5 mean|x_i - x_j|.  n <= 8192. */
int rdgan_crps_ensemble(const float* ens, const float* obs, const float* scale, float* crps_out, int n, long npix,
                        void* stream);

/* Op-level entry points used by the parity tests (tests/test_hip_ops.py). */
/* Conv3D forward, TF semantics.  x [B,D,H,W,Cin] -> y [B,Do,Ho,Wo,Cout]; upsample=1 folds
 * UpSampling3D(2) in front (T:330-331); pad = zero padding before each axis; Cin%4==0,
 * Cout%64==0 (or 32). */
int rdgan_op_conv3d(const float* x, const float* w, const float* bias, float* y, int B, int D, int H,
                    int W, int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d,
                    int pad_h, int pad_w, int upsample, void* stream);
/* the same contraction with bf16 operands (x and w rounded to nearest-even bf16 on the device, fp32 accumulation;
 * v_mfma_f32_32x32x16_bf16): the GEMM of the "bf16" storage mode.  out_bf16 = 0: y is fp32 (the arithmetic, checked at
 * 1e-5 against the oracle on bf16-rounded operands); 1: y is bf16 (2 bytes per element), the accumulator rounded once
 * after the epilogue, as the storage mode writes it; 2: as 1 through k_conv_gemm_f16 ("conv_f16"; Cout % 128 == 0, else -2).
 * No folded upsample; Cin, Cout % 64 == 0. */
int rdgan_op_conv3d_bf16(const float* x, const float* w, const float* bias, float* y, int B, int D, int H,
                         int W, int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d,
                         int pad_h, int pad_w, int out_bf16, void* stream);
/* input gradient of the above for stride 2 (parity-phase plan) or stride 1: gy -> gx (same dims as x,
 * on the upsampled grid when the forward had upsample=1: D,H,W are the conv's input extents). */
int rdgan_op_conv3d_dgrad(const float* gy, const float* w, float* gx, int B, int D, int H, int W,
                          int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h,
                          int pad_w, void* stream);
/* the same input gradient with bf16 operands (gy, w rounded to bf16; fp32 accumulation): Cin, Cout % 64 == 0. */
int rdgan_op_conv3d_dgrad_bf16(const float* gy, const float* w, float* gx, int B, int D, int H, int W,
                               int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h,
                               int pad_w, void* stream);
/* weight gradient dW [3,3,3,Cin,Cout] of the forward above. */
int rdgan_op_conv3d_wgrad(const float* x, const float* gy, float* dw, int B, int D, int H, int W,
                          int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h,
                          int pad_w, int upsample, void* stream);
/* weight gradient with bf16 operands (x, gy rounded to bf16; fp32 accumulation and output): Cin % 128 == 0 and
 * Cout % 64 == 0, or Cin == 64 and Cout % 128 == 0 (tiles of 128 or 256 rows). */
int rdgan_op_conv3d_wgrad_bf16(const float* x, const float* gy, float* dw, int B, int D, int H, int W,
                               int Cin, int Cout, int Do, int Ho, int Wo, int stride, int pad_d, int pad_h,
                               int pad_w, void* stream);
/* Weight gradient of one tap group (g = 0, 1, 2) of the shared-centre form of a generator block through the production
 * plan (4 parity phases x 4 taps: the even tap count that selects the 256-row tile at B*D*H*W >= 65536).  g = 0: src = E
 * [B,D+1,H,W,Cin], dy [B,2D,2H,2W,Cout] (even planes used); g = 1: src = x [B,D,H,W,Cin], dy = plane-pair sums
 * [B,D,2H,2W,Cout]; g = 2: E at j = s+1 against the odd planes.  dU [48][Cin][Cout], forms g*16 .. g*16+15 written.
 * bf16 = 1: operands rounded to bf16 on the device, fp32 accumulation (Cin % 128 == 0). */
int rdgan_op_fastd_wgrad(const float* src, const float* dy, float* dU, int B, int D, int H, int W, int Cin, int Cout,
                         int g, int bf16, void* stream);
/* Weight gradient of the last generator conv (64 -> 1; backward of T:345) alone, through the production kernels:
 * dW[27][64] from dl [B][24][nd][nd] and h3 [B][24][nd][nd][64].  kernel = 1: the matrix-pipe kernel (k_g9_wgrad_mfma),
 * 0: the scalar kernel it replaced (k_g9_wgrad_pairs; nd <= 72); bf16 = 1: h3 rounded to bf16 first (storage mode). */
int rdgan_op_g9_wgrad(const float* dl, const float* h3, float* dW, int B, int nd, int bf16, int kernel, void* stream);
/* Generator block 3 forward of the bf16 storage mode (T:340-343 on the 12 x 8 x 8 x 128 input of ndomain 16) through the slab
 * kernel alone (k_upconv_slab16): x and w are rounded to bf16 on the device, y = LeakyReLU(PixelNorm(upconv(x) + bias)) comes
 * back as fp32 (the bf16 output widened), rinv [B,24,16,16] = 1/l2 per grid point; dbg: NULL, or [B*24*16*16][4] floats (test
 * hook: row sum of squares and 1/l2 as the two lane halves of a row computed them). */
int rdgan_op_upconv_slab16(const float* x, const float* w, const float* bias, float* y, float* rinv, float* dbg, int B, void* stream);
/* The same block of the large-domain variant (L:355-358; source planes of H x W positions, multiples of 8: 32 x 32 at ndomain 64)
 * through the tiled slab kernel alone (k_upconv_slab_t16: 8 x 8 tiles with their halo resident, K in two halves):
 * x [B,12,H,W,128], y [B,24,2H,2W,64], rinv [B,24,2H,2W], dbg NULL or [B*24*2H*2W][4]. */
int rdgan_op_upconv_slab_t16(const float* x, const float* w, const float* bias, float* y, float* rinv, float* dbg, int B, int H, int W,
                             void* stream);
/* Generator block 2 forward of the bf16 storage mode (T:335-338 on the 6 x 4 x 4 x 256 input of ndomain 16) through the slab kernel
 * alone (k_upconv2_slab16): x and w [3,3,3,256,128] are rounded to bf16 on the device, y [B,12,8,8,128] = LeakyReLU(PixelNorm(
 * upconv(x) + bias)) comes back as fp32 (the bf16 output widened), rinv [B,12,8,8] = 1/l2 per grid point. */
int rdgan_op_upconv2_slab16(const float* x, const float* w, const float* bias, float* y, float* rinv, int B, void* stream);
/* Weight gradient of generator block 3 in the collapsed form (backward of T:340-341 on the 12 x 8 x 8 x 128 input of ndomain 16)
 * through the slab kernel of the bf16 storage mode alone (k_upconv_wgrad_slab16): x [B,12,8,8,128] and dy [B,24,16,16,64] (gradient
 * at the conv output) are rounded to bf16 on the device; dWc [64 = phase*8 + tap][128][64] fp32 -- entry (phase, tap) is the sum over
 * samples and source positions r of x[r + off(phase, tap)] (outer) dy[2 r + phase], off = phase - 1 + tap per axis. */
int rdgan_op_upconv_wgrad_slab16(const float* x, const float* dy, float* dWc, int B, void* stream);
/* Forward of the critic's second layer (T:291-293, Conv3D(128, 3x3x3, stride 2, 'same') on 11 x 7 x 7 x 64 + bias + LeakyReLU +
 * dropout) through the slab kernel of the bf16 storage mode alone (k_d2_fwd_slab16): x [B,11,7,7,64] and w [3,3,3,64,128] are rounded
 * to bf16 on the device, y [B,6,4,4,128] comes back as fp32 (the bf16 output widened); seed = 0: dropout off. */
int rdgan_op_d2_fwd_slab16(const float* x, const float* w, const float* bias, float* y, int B, uint64_t seed, void* stream);
/* Weight gradient of the critic's second layer (backward of T:291, Conv3D(128, 3x3x3, stride 2, 'same') on 11 x 7 x 7 x 64) through
 * the slab kernel of the bf16 storage mode alone (k_d2_wgrad_slab16): x [B,11,7,7,64] (layer 1's output) and dy [B,6,4,4,128] are
 * rounded to bf16 on the device; dW [3,3,3,64,128] fp32 = sum over samples and output positions o of x[2 o + tap - 1] (outer) dy[o]. */
int rdgan_op_d2_wgrad_slab16(const float* x, const float* dy, float* dW, int B, void* stream);
/* the same through the tiled kernel of the larger domains (k_d2_wgrad_slab_t16: work items = 4 x 4 tiles of output positions with the
 * layer-1 positions their taps reach, one halo position per odd class): x [B,11,2 OH - 1,2 OW - 1,64], dy [B,6,OH,OW,128], OH and OW
 * multiples of 4 (ndomain 32 / 48 / 64: OH = ndomain / 4). */
int rdgan_op_d2_wgrad_slab_t16(const float* x, const float* dy, float* dW, int B, int OH, int OW, void* stream);
/* Weight gradient of the critic's third layer (backward of T:295, Conv3D(256, 3x3x3, stride 2, 'same') on 6 x 4 x 4 x 128 ->
 * 3 x 2 x 2 x 256) through the slab kernel of the bf16 storage mode alone (k_d3_wgrad_slab16): x [B,6,4,4,128] (layer 2's output) and
 * dy [B,3,2,2,256] are rounded to bf16 on the device; dW [3,3,3,128,256] fp32 = sum over samples and o of x[2 o + tap] (outer) dy[o]. */
int rdgan_op_d3_wgrad_slab16(const float* x, const float* dy, float* dW, int B, void* stream);
/* Input gradient of the critic's second layer (backward of T:291, Conv3D(128, 3x3x3, stride 2, 'same') on the 11 x 7 x 7 x 64
 * output of layer 1, ndomain 16) through the slab kernel of the bf16 storage mode alone (k_d2_dgrad_slab16): gy [B,6,4,4,128], the
 * layer's kernel w [3,3,3,64,128] and aux [B,11,7,7,64] (layer 1's output) are rounded to bf16 on the device;
 * gx [B,11,7,7,64] = conv3d_input_grad(gy, w) * gate(aux), as fp32; gate = LeakyReLU'(aux), and with use_drop != 0 aux is read as
 * a stored post-dropout activation: +0.0 = dropped (gate 0), anything else kept (gate LeakyReLU'(aux) / 0.75). */
int rdgan_op_d2_dgrad_slab16(const float* gy, const float* w, const float* aux, float* gx, int B, int use_drop, void* stream);
/* The same through the tiled kernel of the larger domains (k_d2_dgrad_slab_t16; L:291-293): gy [B,6,OH,OW,128], aux and gx
 * [B,11,2 OH - 1,2 OW - 1,64]; OH, OW multiples of 4 (ndomain 32 / 48 / 64: 8 / 12 / 16). */
int rdgan_op_d2_dgrad_slab_t16(const float* gy, const float* w, const float* aux, float* gx, int B, int OH, int OW, int use_drop,
                               void* stream);
/* PixelNormalization + LeakyReLU(0.2) forward (T:255-266, T:333) and its backward. C in {64,128,256}. */
int rdgan_op_pixelnorm_lrelu(const float* y, float* h, float* rinv, long npix, int C, void* stream);
int rdgan_op_pixelnorm_lrelu_bwd(const float* gh, const float* h, const float* rinv, float* dy,
                                 long npix, int C, void* stream);
/* Test hook: copies the first n floats of an activation tensor the last forward pass left in the workspace into `out`
 * (device, fp32): which = 0..3 generator h0..h3 [B, D, H, W, C] after LeakyReLU; 4..7 critic layers 1..4 after LeakyReLU
 * and dropout ([B] samples after a generator step or a critic forward; [3B] = real | fake | interpolated after a critic step,
 * whose interpolated third is only meaningful with the option "keep_gates": without it the second sweep of the gradient penalty
 * has overwritten it).  The gradient parity tests take the LeakyReLU slope pattern of the run from here, so that the fp64
 * oracle differentiates the same piecewise-linear branch.  which = 8 (bf16 storage mode, ndomain 16, option "d2_gate_bits"): the packed
 * gate bytes of critic layer 1 as floats, 16 per row: byte q of a row = channels 4 q .. 4 q + 3, two bits each (bit 0: output > 0,
 * bit 1: dropped). */
int rdgan_debug_activation(rdgan_handle* h, int which, float* out, long n, void* stream);
/* dropout keep-scale mask (0 or 1/0.75) and uniforms of the counter RNG, for pinning it to oracle/rng.py */
int rdgan_op_rng(uint64_t seed, uint32_t stream_id, float* mask_out, float* uniform_out, long n,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif
