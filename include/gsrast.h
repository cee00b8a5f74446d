/*
 * gsrast.h -- C ABI of libgsrast.so, the MI355X-native (gfx950) differentiable 3D-Gaussian rasterizer.
 *
 * This is the drop-in boundary for the reference's native layer
 *   $RAST = submodules/gaustudio-diff-gaussian-rasterization   (in GAP-LAB-CUHK-SZ/gaustudio)
 *   $RAST/cuda_rasterizer/rasterizer.h:20-92   class CudaRasterizer::Rasterizer { markVisible, forward, backward }
 * Each entry point below replaces one of those static methods: same argument meaning and order, plain
 * pointers and sizes, plus an explicit HIP stream (the reference launches on the legacy default stream).
 * The torch extension `_C` (gaustudio_amd/csrc/torch_binding.cpp) is a thin adapter over these, replacing
 * $RAST/rasterize_points.cu:35-231 / ext.cpp:15-19.
 *
 * Conventions shared with the reference:
 *   - all tensors float32, row-major, device memory of the current HIP device;
 *   - an absent optional input is a NULL pointer (forward.cu:205,241; backward.cu:406,410);
 *   - viewmatrix / projmatrix are float[16] holding the column-major 4x4 (i.e. the transposed
 *     torch tensors gaustudio's Camera builds, datasets/__init__.py:154-159);
 *   - outputs are CHW planes (forward.cu:387-395).
 * Differences, all deliberate:
 *   - `background`, `viewmatrix`, `projmatrix`, `cam_pos` may live in HOST or DEVICE memory
 *     (gaustudio's renderers pass a CPU `bg`, renderers/vanilla_renderer.py:23);
 *   - the three opaque buffers are obtained through C callbacks instead of std::function
 *     (rasterizer.h:38-40); their internal layout is private to this library;
 *   - gsr_backward takes uninitialised OUTPUT pointers plus one scratch buffer; the reference wanted
 *     ten pre-zeroed tensors of which two (dL_dconic, dL_ddepth) were scratch (rasterize_points.cu:160-169);
 *   - errors are return codes + gsr_last_error() instead of C++ exceptions.
 */
#ifndef GSRAST_H_INCLUDED
#define GSRAST_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_ERR_HIP (-1)         /* a HIP runtime call or kernel failed; see gsr_last_error() */
#define GSR_ERR_ARG (-2)         /* invalid argument (e.g. neither SH nor colours; rasterizer_impl.cu:245-248) */
#define GSR_ERR_PREFILTERED (-3) /* a point was culled although `prefiltered` is set (auxiliary.h:156-160) */
#define GSR_ERR_ALLOC (-4)       /* an allocator callback returned NULL */

/* Replaces std::function<char*(size_t)> (rasterizer.h:38-40): must return device memory of at
 * least `bytes` bytes, 256-byte aligned, that stays valid until the matching backward has run.
 * gsr_forward may call the BINNING allocator twice (first with its remembered capacity, then -- only if that turned
 * out too small -- with the exact size; the second result replaces the first, which may be released in stream
 * order, as torch's resize_ does). */
typedef char* (*gsr_alloc_fn)(void* ctx, size_t bytes);

/* ABI version of this header (bumped on any signature change). */
int gsr_abi_version(void);

/* Message of the last error on the calling thread ("" if none). */
const char* gsr_last_error(void);

/* Replaces Rasterizer::markVisible (rasterizer.h:24-29; rasterizer_impl.cu:141-153).
 * present[i] = 1 iff Gaussian i passes the near-plane test p_view.z > 0.2 (auxiliary.h:154). */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* stream);

/* Replaces Rasterizer::forward (rasterizer.h:31-59; rasterizer_impl.cu:198-343).
 * Returns num_rendered (>= 0) or a negative GSR_ERR_*.  num_rendered keeps the reference's definition -- the sum
 * over visible Gaussians of the tiles of their getRect square (rasterizer_impl.cu:280-284) -- although fewer
 * instances are actually binned (only tiles that can hold a pixel with alpha >= 1/255; gsr_inspect_counts).
 * D = active SH degree, M = SH coefficients per Gaussian as stored (row stride of `shs`).
 * Exactly one of {shs, colors_precomp} and one of {scales+rotations, cov3D_precomp} must be non-NULL.
 * out_color[3,H,W], out_depth[1,H,W], out_median_depth[3,H,W], out_opacity[1,H,W], radii[P] are
 * fully overwritten (no pre-zeroing needed).  The host waits once for num_rendered (the reference blocks the
 * device as well, rasterizer_impl.cu:283-284); here the remaining kernels are already enqueued by then. */
int gsr_forward(gsr_alloc_fn geometry_alloc, void* geometry_ctx,
                gsr_alloc_fn binning_alloc, void* binning_ctx,
                gsr_alloc_fn image_alloc, void* image_ctx,
                int P, int D, int M,
                const float* background,
                int width, int height,
                const float* means3D,
                const float* shs,
                const float* colors_precomp,
                const float* opacities,
                const float* scales,
                float scale_modifier,
                const float* rotations,
                const float* cov3D_precomp,
                const float* viewmatrix,
                const float* projmatrix,
                const float* cam_pos,
                float tan_fovx, float tan_fovy,
                int prefiltered,
                float* out_color,
                float* out_depth,
                float* out_median_depth,
                float* out_opacity,
                int* radii,
                int debug,
                void* stream);

/* Bytes of device scratch gsr_backward needs for P Gaussians and R = num_rendered instances
 * (49 B per instance: the per-instance partial-gradient rows that replace the reference's float
 * atomics, backward.cu:559-607, plus one validity byte each; the per-Gaussian row offsets live in the
 * geometry buffer). */
size_t gsr_backward_scratch_bytes(int P, int R);

/* Lower bounds on the sizes of the geometry / image buffers a gsr_forward with these dimensions requests
 * (adapters use them to reject buffers that cannot belong to the backward they are handed to). */
size_t gsr_geometry_bytes(int P);
size_t gsr_image_bytes(int width, int height);

/* Replaces Rasterizer::backward (rasterizer.h:61-91; rasterizer_impl.cu:347-452).
 * R = num_rendered returned by the matching gsr_forward; geom/binning/image buffers are the ones its
 * allocators returned, unmodified and in full (besides the sorted lists the binning buffer carries the forward's
 * 16-bit block mask of every list entry, which the compositing backward reads instead of recomputing a cull; a
 * flag in the image buffer says whether the forward left them).  Outputs (all fully overwritten, rows of culled Gaussians = 0):
 *   dL_dmean2D[P,3] (xy used, already scaled by 0.5*W / 0.5*H, backward.cu:493-494,598-599),
 *   dL_dopacity[P], dL_dcolor[P,3], dL_dmean3D[P,3], dL_dcov3D[P,6] (may be NULL when cov3D_precomp is NULL: the gradient
 *   of a covariance the operator built itself from scale / rotation has no reader), dL_dsh[P,M,3] (may be NULL if M==0),
 *   dL_dscale[P,3], dL_drot[P,4].
 * Only channel 0 of dL_dpix_median_depth[3,H,W] is read (backward.cu:481-482).
 * Each of dL_dpix[3,H,W], dL_dpix_depth[1,H,W], dL_dpix_median_depth, dL_dpix_final_opacity[1,H,W] may be NULL: the loss does not
 * use that output, its gradient is zero -- nothing is loaded for it and no zero plane has to be materialised by the caller (the
 * reference reads all four, backward.cu:476-483, so its binding fills zeros); results are bit-equal to a call with explicit zero
 * planes.  Colour alone (the other three NULL: the usual 3DGS training loss) runs a compositing kernel specialised on it. */
int gsr_backward(int P, int D, int M, int R,
                 const float* background,
                 int width, int height,
                 const float* means3D,
                 const float* shs,
                 const float* colors_precomp,
                 const float* scales,
                 float scale_modifier,
                 const float* rotations,
                 const float* cov3D_precomp,
                 const float* viewmatrix,
                 const float* projmatrix,
                 const float* campos,
                 float tan_fovx, float tan_fovy,
                 const int* radii,
                 const char* geom_buffer,
                 const char* binning_buffer,
                 const char* image_buffer,
                 const float* dL_dpix,
                 const float* dL_dpix_depth,
                 const float* dL_dpix_median_depth,
                 const float* dL_dpix_final_opacity,
                 float* dL_dmean2D,
                 float* dL_dopacity,
                 float* dL_dcolor,
                 float* dL_dmean3D,
                 float* dL_dcov3D,
                 float* dL_dsh,
                 float* dL_dscale,
                 float* dL_drot,
                 char* scratch,
                 int debug,
                 void* stream);

/* gsr_backward in stages (new; lets the caller overlap a collective with the tail of the backward,
 * gaustudio_amd/parallel.py): `parts` selects GSR_BWD_PART_MAIN (compositing backward + the per-Gaussian geometry
 * stage: every output except dL_dsh, and dL_dmean3D still lacks its SH term) and / or GSR_BWD_PART_SH (the SH
 * stage for the Gaussians [sh_g0, sh_g1), sh_g0 a multiple of 256: writes those rows of dL_dsh and adds their SH
 * term to dL_dmean3D; needs MAIN to have run on the same buffers).  gsr_backward == both parts over [0, P). */
#define GSR_BWD_PART_MAIN 1
#define GSR_BWD_PART_SH 2
/* with GSR_BWD_PART_SH (gsr_backward_ex only): the SH stage in its FACTORED form for the multi-GPU gradient exchange
 * (gaustudio_amd/parallel.py FactoredGradExchange) -- dL_dsh is not written (may be NULL); dL_dcolor[P,3] is overwritten
 * in place with dRGB, the clamp-masked colour gradient the SH basis is multiplied with (backward.cu:35-40); dL_dmean3D
 * gets its SH term as usual.  SH colours in one [P,M,3] tensor only. */
#define GSR_BWD_PART_SH_COLORS 4
/* with GSR_BWD_PART_SH_COLORS, on every call of one backward (gsr_backward_ex only): the GEOMETRY stage of GSR_BWD_PART_MAIN
 * already leaves dRGB in dL_dcolor (the clamp mask is in the forward's record: it needs no SH coefficients), and the SH stage
 * does not write dL_dcolor again.  A caller that runs MAIN and SH as two calls can start the all-gather of this view's colour
 * gradients between them -- before the SH-direction stage has read 192 B of coefficients per Gaussian -- without a race on
 * the slot.  Same bits as the one-call form. */
#define GSR_BWD_PART_COLORS_EARLY 8
/* BANDED backward (round 6, gsr_backward_ex only; gaustudio_amd/parallel.py FactoredGradExchange(bands=2)): the compositing backward
 * and the per-Gaussian geometry stage of GSR_BWD_PART_MAIN run as TWO calls on the same buffers and the same scratch, so that a
 * caller can start exchanging the finished part of the gradients while the other half of the image is still composited:
 *   parts = MAIN | GSR_BWD_PART_BAND_FIRST  [| SH_COLORS | COLORS_EARLY], sh_g0 = split tile row S (0 <= S <= tile rows):
 *       composites the tile rows [0, S) and writes every output row except dL_dsh of the Gaussians of CLASS 1 = those that are
 *       invisible (radii == 0: zeros) or whose tile rect ends at or before row S -- all of their per-instance rows exist now;
 *   parts = MAIN | GSR_BWD_PART_BAND_SECOND [| ...], sh_g0 = the same S: composites the rows [S, tile rows) and writes the
 *       Gaussians of CLASS 2 (the others).
 * Together the two calls write every Gaussian exactly once, each from exactly the rows, in exactly the order, of the one-call
 * backward: BIT-IDENTICAL outputs.  No SH part in a band call (it follows the second band: GSR_BWD_PART_SH over [0, P));
 * gsr_band_classes tells the classes apart.  (With a forward that itself rendered a tile band, S counts tile rows of the
 * whole image as the forward's tile_row_lo / tile_row_hi do.) */
#define GSR_BWD_PART_BAND_FIRST 16
#define GSR_BWD_PART_BAND_SECOND 32
/* first[g] = 1 if Gaussian g is visible and of class 1 for split row S (its gradient rows are final after the FIRST band call),
 * second[g] = 1 if visible and of class 2; 0 otherwise (int32[P] each, device memory; either may be NULL).  Fed to
 * gsr_visible_index they give the headers of the two packed colour messages of a banded view. */
int gsr_band_classes(int P, const int* radii, const char* geom_buffer, int split_tile_row, int* first, int* second, void* stream);
int gsr_backward_parts(int parts, int sh_g0, int sh_g1, int P, int D, int M, int R, const float* background, int width,
                       int height, const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp, float tan_fovx,
                       float tan_fovy, const int* radii, const char* geom_buffer, const char* binning_buffer,
                       const char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth,
                       const float* dL_dpix_median_depth, const float* dL_dpix_final_opacity, float* dL_dmean2D,
                       float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                       float* dL_dscale, float* dL_drot, char* scratch, int debug, void* stream);

/* The other half of the factored exchange: dL_dsh[P,M,3] = sum over the N views r = 0 .. N-1, in this order, of
 * basis_D(normalize(means3D[g] - campos[r])) (x) colors[r][g], i.e. the SH gradient of a multi-view step rebuilt from
 * the per-view dRGB (what GSR_BWD_PART_SH_COLORS leaves in dL_dcolor; rows of Gaussians culled in a view are zero there)
 * and the views' camera centres campos[N,3] (device memory).  Same arithmetic as the per-view SH backward and a fixed
 * order of views: bit-identical to accumulating the N per-view gradients one after the other.  colors[N,P,3].
 * (new; the reference has no multi-GPU path at all) */
int gsr_sh_grad_from_colors(int P, int D, int M, int N, const float* means3D, const float* campos, const float* colors,
                            float* dL_dsh, void* stream);

/* ---- compacted rows for the multi-GPU exchange (new; gaustudio_amd/parallel.py FactoredGradExchange(compact="view")) ----
 * A view of a real capture sees a fraction of the scene and the gradient rows of a Gaussian culled in a view are exactly zero,
 * so a view's colour gradients travel as a MESSAGE of 32-bit words:
 *   [0] K = visible Gaussians (radii > 0), [1] P, [2..3] 0; ceil(P/256) block bases (visible Gaussians in front of each
 *   256-Gaussian block); ceil(P/32) mask words (bit g & 31 of word g >> 5); padding to gsr_msg_header_words(P); then K rows.
 * gsr_visible_index builds the header from `radii` (right after the forward, off the critical path; scratch_counts:
 * ceil(P/256) words of device scratch), gsr_union_index the header of the OR of N messages' masks (message r starts at word
 * msg_offsets[r] of msgs; offsets in device memory); gsr_pack_rows copies the rows in[P,C] of the header's Gaussians to out[row * out_stride + col0 ..], gsr_unpack_rows
 * the other way (rows of other Gaussians are left untouched); gsr_sh_grad_from_packed is gsr_sh_grad_from_colors reading N
 * messages (rows of 3 floats behind each header) instead of dense [N,P,3] colours: same arithmetic, same order, same bits. */
size_t gsr_msg_header_words(int P);
int gsr_visible_index(int P, const int* radii, uint32_t* msg, uint32_t* scratch_counts, void* stream);
int gsr_union_index(int P, int N, const uint32_t* msgs, const unsigned long long* msg_offsets, uint32_t* out_hdr, uint32_t* scratch_counts,
                    void* stream);
int gsr_pack_rows(int P, int C, const uint32_t* hdr, const float* in, float* out, int out_stride, int col0, void* stream);
int gsr_unpack_rows(int P, int C, const uint32_t* hdr, const float* in, int in_stride, int col0, float* out, void* stream);
/* The geometry block of the exchange ([means3D 3 | opacity 1 | scales 3 | rotations 4] = 11 floats per Gaussian, four tensors) in one
 * pass each way: gsr_pack_geometry writes rows[r * 11 ..] = the 11 floats of the r-th Gaussian of `hdr` (rows must hold hdr[0] rows)
 * and ORs 1 into *flag_outside (device word, cleared by the caller) when a Gaussian OUTSIDE the header has a non-zero value -- a
 * gradient the rasterizer cannot have produced (it writes zeros for culled Gaussians): the caller must then exchange the dense
 * block; gsr_unpack_geometry copies the rows back (other Gaussians untouched). */
int gsr_pack_geometry(int P, const uint32_t* hdr, const float* g_means3D, const float* g_opacity, const float* g_scales,
                      const float* g_rotations, float* rows, uint32_t* flag_outside, void* stream);
int gsr_unpack_geometry(int P, const uint32_t* hdr, const float* rows, float* g_means3D, float* g_opacity, float* g_scales,
                        float* g_rotations, void* stream);
int gsr_sh_grad_from_packed(int P, int D, int M, int N, const float* means3D, const float* campos, const uint32_t* msgs,
                            const unsigned long long* msg_offsets, float* dL_dsh, void* stream);

/* Process-wide tunables (also read from the environment at load: GSR_TIGHT_BINNING, GSR_CULL, GSR_FWD_VARIANT,
 * GSR_BWD_VARIANT, GSR_SPECULATIVE).  The first five never change a bit of the forward (the backward variants add the
 * same terms in another order); they exist for A/B measurements and parity tests:
 *   "tight_binning" 1|0  bin each Gaussian into the tight sub-rect of the reference's getRect square (default 1);
 *   "cull"          1|0  block-level culling + pcut pre-test in composite_fwd (default 1);
 *   "fwd_variant"   0 = composite_fwd with per-quarter (4x4 pixel) instance lists (default), 1 = per-wave (8x8) walk
 *                        (A/B builds only, see "ab_variants");
 *   "speculative"   1|0  enqueue binning + compositing before the host has read the instance count (default 1);
 *   "bwd_variant"   -1 = auto (gsr_selftest), bit 0 = keep the select on T in composite_bwd, bit 1 = the per-wave
 *                        (8x8) kernel instead of the per-quarter one (A/B builds only);
 *   "ab_variants"   read-only: 1 if the superseded per-wave compositing kernels were compiled in (csrc/Makefile AB=1).  The
 *                        shipped library is built without them: asking for fwd_variant 1 / bwd_variant bit 1 is then GSR_ERR_ARG;
 *   "fast_exp"      0|1  process default of gsr_options.fast_exp (below);
 *   "tile_order"    1|0  (GSR_TILE_ORDER) backward of a SKEWED frame (longest tile list > 1024 entries and > 4x the mean): run
 *                        the tiles longest walk first instead of in XCD bands (two small extra launches; changes no result bit);
 *   "roctx"         0|1  (GSR_ROCTX) wrap every stage of gsr_forward / gsr_backward in a roctx range ("gsr.preprocess_fwd",
 *                        "gsr.scan", ... ) for rocprofv3 --marker-trace timelines; the marker library is dlopen()ed, get
 *                        returns 1 only if it was found;
 *   "forget_forwards" (set only) drop the host-side memory of which mode each live forward ran in: the next gsr_backward of such buffers
 *                        reads the forward's own 4-byte control word from the image buffer instead (tests; always correct, one small sync);
 *   "bin_capacity"  n    binning capacity (instances) assumed by the next gsr_forward on the current device
 *                        (0 = forget; tests use a small n to force the re-allocate-and-relaunch path);
 *   "tile_row_lo", "tile_row_hi"  tile-grid sharding of ONE view across processes: only the 16-pixel
 *                        tile rows [lo, hi) are binned, composited and differentiated; pixels outside the band come
 *                        back as an empty scene's, gradients are the band's partial sums (SUM them over the ranks);
 *                        radii and num_rendered keep describing the whole view.  hi <= 0: the whole image. */
int gsr_set_option(const char* name, int value);
int gsr_get_option(const char* name);

/* ---- per-call options (ABI v6).  The process-wide switches above change the behaviour of every caller in the
 * process; two configurations in one process (two tile bands from two threads, a bit-exact evaluation pass next to a
 * fast_exp training loop) need them per call.  Every field: -1 = take the process default (gsr_set_option /
 * environment).  gsr_forward_ex / gsr_backward_ex are the supersets of the plain and the raw entry points (an absent
 * input is NULL: `shs_rest` NULL = `shs` holds all M coefficients; activation_flags 0 = activated inputs) with the
 * options in front; opt == NULL = all defaults, and gsr_forward(...) == gsr_forward_ex(NULL, ...).
 * The backward of a forward must be given the same `fast_exp` (the adapters keep the options of the forward with the
 * graph; the forward also records what it ran with in the image buffer, and a debug-mode backward checks it).
 *   fast_exp  0|1   exp on the transcendental unit (v_exp_f32) in both compositing kernels instead of the reproducible
 *                   9-instruction polynomial: not bit-reproducible against the CPU oracle any more (values within
 *                   ~1e-6 relative, threshold flips attributed by tests/test_gpu_fastexp.py); default 0. */
typedef struct gsr_options {
	int32_t struct_bytes;   /* sizeof(gsr_options) of the caller: fields beyond it are taken as -1 */
	int32_t tight_binning;
	int32_t cull;
	int32_t fwd_variant;
	int32_t bwd_variant;
	int32_t speculative;
	int32_t tile_row_lo;    /* tile band of THIS call: [lo, hi), hi <= 0 with lo >= 0 = the whole image */
	int32_t tile_row_hi;
	int32_t fast_exp;
	int32_t forward_only;   /* 0|1  (default 0) this forward will have no backward: skip what only a backward reads (the 36 B per Gaussian
	                         * of d(rgb)/d(view direction) that preprocess_fwd leaves for the SH backward).  gsr_backward on the buffers of
	                         * such a forward (any `parts`) is refused: from a host-side memory of the last 64 forwards, and -- debug = 1 --
	                         * from the forward's own record in the image buffer.  The Python adapters set it for calls under
	                         * torch.no_grad() (the grad mode is captured by the module wrapper: ctx.needs_input_grad ignores it) and for
	                         * calls none of whose inputs requires a gradient. */
} gsr_options;
void gsr_options_init(gsr_options* opt);   /* struct_bytes = sizeof, every field -1 */

int gsr_forward_ex(const gsr_options* opt, gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc,
                   void* binning_ctx, gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                   const float* background, int width, int height, const float* means3D, const float* shs,
                   const float* shs_rest, const float* colors_precomp, const float* opacities, const float* scales,
                   float scale_modifier, const float* rotations, const float* cov3D_precomp, int activation_flags,
                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                   int prefiltered, float* out_color, float* out_depth, float* out_median_depth, float* out_opacity,
                   int* radii, int debug, void* stream);
/* parts / sh_g0 / sh_g1 as in gsr_backward_parts; dL_dsh_rest NULL unless shs_rest is given. */
int gsr_backward_ex(const gsr_options* opt, int parts, int sh_g0, int sh_g1, int P, int D, int M, int R,
                    const float* background, int width, int height, const float* means3D, const float* shs,
                    const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
                    const float* rotations, const float* cov3D_precomp, int activation_flags, float tan_fovx,
                    float tan_fovy, const int* radii, const char* geom_buffer, const char* binning_buffer,
                    const char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth,
                    const float* dL_dpix_median_depth, const float* dL_dpix_final_opacity, float* dL_dmean2D,
                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                    float* dL_dsh_rest, float* dL_dscale, float* dL_drot, char* scratch, int debug, void* stream);

/* Device self-test of the arithmetic identities composite_bwd relies on: bit 0: v_rcp_f32(1.0) == 1.0,
 * bit 1: t * v_rcp_f32(1.0) == t.  Synchronises the stream.  Negative = GSR_ERR_*. */
int gsr_selftest(void* stream);

/* ---- fused parameter activations (new, not in the reference's interface) ----
 * gsr_forward_raw / gsr_backward_raw take GauStudio's RAW point-cloud attributes -- f_dc[P,1,3] and f_rest[P,M-1,3]
 * instead of the concatenated sh[P,M,3] (models/vanilla_sg.py:103-106), pre-activation opacity / scale / rotation --
 * and apply VanillaPointCloud's activations inside the kernels (models/vanilla_sg.py:33-37: exp, sigmoid,
 * F.normalize) according to activation_flags.  Gradients are returned w.r.t. the raw attributes. */
#define GSR_ACT_OPACITY_SIGMOID 1
#define GSR_ACT_SCALE_EXP 2
#define GSR_ACT_ROT_NORMALIZE 4

int gsr_forward_raw(gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc, void* binning_ctx,
                    gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                    int height, const float* means3D, const float* f_dc, const float* f_rest, const float* raw_opacities,
                    const float* raw_scales, float scale_modifier, const float* raw_rotations, int activation_flags,
                    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                    float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_median_depth,
                    float* out_opacity, int* radii, int debug, void* stream);

int gsr_backward_raw(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                     const float* f_dc, const float* f_rest, const float* raw_scales, float scale_modifier,
                     const float* raw_rotations, int activation_flags, float tan_fovx, float tan_fovy,
                     const int* radii, const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                     const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_median_depth,
                     const float* dL_dpix_final_opacity, float* dL_dmean2D, float* dL_draw_opacity, float* dL_dcolor,
                     float* dL_dmean3D, float* dL_dcov3D, float* dL_df_dc, float* dL_df_rest, float* dL_draw_scale,
                     float* dL_draw_rot, char* scratch, int debug, void* stream);

/* ---- introspection of the opaque buffers (tests, debugging).  The reference exposes the same state
 * only implicitly through GeometryState/BinningState/ImageState (rasterizer_impl.h:33-64).
 * Any output pointer may be NULL.  All outputs are device memory. ---- */

/* means2D[P,2], depths[P], conic_opacity[P,4], rgb[P,3], clamped[P,3] (0/1 bytes), tiles_touched[P].
 * Rows of culled Gaussians (radii==0) are written as zeros. */
int gsr_inspect_geometry(const char* geom_buffer, int P, const int* radii, float* means2D, float* depths,
                         float* conic_opacity, float* rgb, unsigned char* clamped,
                         uint32_t* tiles_touched, void* stream);

/* After gsr_backward: sums[P,10] = the per-Gaussian totals of the compositing stage that feed the
 * per-Gaussian stage, in the order {dL_dmean2D.x, .y, dL_dconic a, b, c, dL_dopacity, dL_dcolor r, g, b,
 * dL_ddepth} (what the reference accumulates with atomicAdd, backward.cu:559-607), recomputed from
 * the scratch rows with the same fixed summation order gsr_backward used. */
int gsr_inspect_backward_sums(const char* geom_buffer, const char* scratch, int P, int R, const int* radii,
                               float* sums, void* stream);

/* out = { instances binned (tight rects; the length of point_list), longest tile list, the reference-defined
 * num_rendered (what gsr_forward returned), error flags: bit 0 = the reference-defined count overflowed 2^31 - 1, bit 1 = the
 * long-list sort overflowed one of its work queues (a capacity bound violated: point_list is not completely sorted; debug-mode
 * gsr_forward / gsr_backward calls fail on it) }.  `out` is HOST memory; synchronises the stream. */
int gsr_inspect_counts(const char* image_buffer, int width, int height, uint32_t out[4], void* stream);

/* Instances composite_fwd actually STAGED for this frame, summed over the tiles: a tile's workgroup stops fetching its list once
 * every pixel of the tile has saturated (forward.cu:278-285 `done`), 256 entries at a time -- on dense frames (C4: lists of 3.4 k
 * entries, 99 % of the pixels saturate after a few hundred) that is a fraction of the binned instances, and it is the count the
 * kernel's byte traffic follows (bench.py's `roofline`).  Left by the kernel in the image buffer's (by then dead) tile counters.
 * `staged_total` is HOST memory; synchronises the stream.  Valid after a gsr_forward with P > 0 and at least one instance. */
int gsr_inspect_staged(const char* image_buffer, int width, int height, unsigned long long* staged_total, void* stream);

/* point_list[R] (Gaussian ids, tile-major, depth-sorted; R = instances binned, see gsr_inspect_counts),
 * ranges[T,2] ([start,end) per tile). */
int gsr_inspect_binning(const char* binning_buffer, const char* image_buffer, int R, int width, int height,
                        uint32_t* point_list, uint32_t* ranges, void* stream);

/* final_T[H,W], n_contrib[H,W] in pixel-major order (forward.cu:385-386). */
int gsr_inspect_image(const char* image_buffer, int width, int height, float* final_T,
                      uint32_t* n_contrib, void* stream);

/* ---- post-render epilogue: the step that follows the operator in gs-extract-mesh /
 * gs-extract-pcd.  `intrinsics` = 3x3 row-major K and `world_to_camera` = 4x4 row-major (Camera.extrinsics,
 * gaustudio/datasets/__init__.py:225-237) are HOST pointers (25 floats); depth / outputs are device memory. ---- */

/* Replaces Camera.depth2point(depth, 'camera' | 'world') (datasets/__init__.py:307-339 with ndc_2_cam :106-112):
 * points[H,W,3]; world_to_camera == NULL -> camera coordinates. */
int gsr_depth_to_points(const float* depth, int width, int height, const float* intrinsics,
                        const float* world_to_camera, float* points, void* stream);

/* Replaces Camera.depth2normal(depth, k, d_min, d_max, 'camera' | 'world') (datasets/__init__.py:342-380): five-tap
 * cross-product normals[H,W,3] of the unprojected depth, (-1,-1,-1) where any tap is outside (d_min, d_max) or
 * outside the image. */
int gsr_depth_to_normals(const float* depth, int width, int height, const float* intrinsics, int k, float d_min,
                         float d_max, const float* world_to_camera, float* normals, void* stream);
/* Both of the above in one pass, with the opacity mask of gs-extract-mesh in front (extract_mesh.py:104-110: the depth of a
 * pixel with opacity < min_opacity counts as 0, so its point is the camera centre and its normal invalid): depth[H,W],
 * opacity[H,W] or NULL (no mask), points[H,W,3] and normals[H,W,3] in the coordinates world_to_camera selects (NULL =
 * camera), either output may be NULL.  Tap distance (k - 1) / 2 <= 2.  Same arithmetic per value as the separate calls. */
int gsr_depth_epilogue(const float* depth, const float* opacity, float min_opacity, int width, int height,
                       const float* intrinsics, int k, float d_min, float d_max, const float* world_to_camera, float* points,
                       float* normals, void* stream);

/* Replaces masked_bilateral_filter (gaustudio/scripts/extract_pcd.py:185-238: numpy + cv2.dilate + cv2.bilateralFilter
 * on the CPU, between the render and depth2point in gs-extract-pcd).  mask[H,W] u8 (non-zero = valid) -> new_mask[H,W]
 * u8 (valid iff the whole d x d window is valid), filtered[H,W]: the bilateral filter (OpenCV's float32 definition:
 * radius max(d/2,1), circular window, reflect-101 border) of the depth normalised over new_mask with everything else
 * set to 0, de-normalised; pixels outside new_mask keep their input depth.  d odd (or 0 = from sigma_space);
 * scratch2 = 8 bytes of device memory.  Parity with cv2 is unpinned (the library is not in this image): restated
 * from its published algorithm, colour weight by expf instead of cv2's 4096-bin interpolated table. */
int gsr_masked_bilateral(const float* depth, const unsigned char* mask, int width, int height, int d, float sigma_color,
                         float sigma_space, float* filtered, unsigned char* new_mask, unsigned int* scratch2, void* stream);

/* ---- TSDF fusion + iso-surface extraction: what gs-extract-mesh does with the rendered
 * depth points (gaustudio/scripts/extract_mesh.py:86,115,145 -> vdbfusion.VDBVolume.integrate /
 * .extract_triangle_mesh, a CPU library the reference pip-installs).  Stateless: the volume is caller-owned device
 * memory --
 *   block_keys[capacity] u64, all bits set = empty (capacity a power of two; 8x8x8-voxel blocks, open addressing);
 *   voxels[capacity * 512] u64, zero-initialised: (sum_q << 24) | count with sum_q the sum of tsdf / sdf_trunc in
 *       2^-15 fixed point (a block's voxels live at its hash slot; the fields hold up to 2^24 - 1 observations
 *       of a voxel);
 *   status[1] u32, zero-initialised: bit 0 set when the hash table overflowed (the ray that hit it is dropped from there
 *       on), bit 1 when a voxel was offered more than 2^24 - 2^20 = 15 728 640 observations (the surplus is dropped; the
 *       count is declared full 2^20 below the field's capacity so that the overflow guard needs no second atomic).
 * Algorithm and parity status: gaustudio_amd/csrc/gsr_tsdf.hip, DESIGN.md s8. ---- */

/* VDBVolume::Integrate(points, origin) with the default weighting (weight 1): points[num_points,3] device,
 * origin[3] host.  Points within 1e-3 voxels of the origin (and non-finite ones) are skipped: that is what depth2point
 * makes of a masked pixel (depth 0), so a whole point map can be passed without compacting the valid pixels. */
int gsr_tsdf_integrate(const float* points, int num_points, const float origin[3], float voxel_size, float sdf_trunc,
                       int space_carving, uint64_t* block_keys, uint64_t capacity, uint64_t* voxels, uint32_t* status,
                       void* stream);
/* The same for an image-shaped point map [num_points / row_width][row_width] (what gsr_depth_to_points returns for a
 * frame; row_width must divide num_points, 0 = plain list): workgroups take 32 x 32 patches of the map, whose rays share
 * voxels in both image directions, and commit several times fewer atomics (2.3x faster at 1080p).  The volume is bit-identical to gsr_tsdf_integrate's. */
int gsr_tsdf_integrate_map(const float* points, int num_points, int row_width, const float origin[3], float voxel_size,
                           float sdf_trunc, int space_carving, uint64_t* block_keys, uint64_t capacity, uint64_t* voxels,
                           uint32_t* status, void* stream);

/* Test / inspection: for the listed hash slots, per voxel (x fastest) the observation count, the mean tsdf
 * (+sdf_trunc where the count is 0) and the raw fixed-point sum. */
int gsr_tsdf_export_blocks(const uint64_t* voxels, const uint32_t* block_slots, int num_blocks, float sdf_trunc,
                           uint32_t* counts, float* tsdf, int64_t* sums, void* stream);

/* VDBVolume::ExtractTriangleMesh(fill_holes, min_weight) in two steps around a caller-side exclusive scan.
 * block_slots[num_blocks]: the occupied hash slots in the order the mesh should be emitted; slot_to_block[capacity]:
 * its inverse.  classify writes cases[num_blocks*512] (u8), edge_flags[num_blocks*512] (u32) and the per-block
 * vertex / triangle counts; emit takes their exclusive scans and writes vertices[nv,3] (f32, world units),
 * triangles[nt,3] (i32, outward winding: normals point towards positive tsdf) and vertex_base[num_blocks*512]. */
int gsr_tsdf_mc_classify(const uint64_t* block_keys, uint64_t capacity, const uint64_t* voxels, const uint32_t* block_slots,
                         int num_blocks, const uint32_t* slot_to_block, float sdf_trunc, float min_weight, int fill_holes,
                         uint8_t* cases, uint32_t* edge_flags, uint32_t* block_num_vertices, uint32_t* block_num_triangles,
                         void* stream);
int gsr_tsdf_mc_emit(const uint64_t* block_keys, uint64_t capacity, const uint64_t* voxels, const uint32_t* block_slots,
                     int num_blocks, const uint32_t* slot_to_block, float voxel_size, float sdf_trunc, const uint8_t* cases,
                     const uint32_t* edge_flags, const uint32_t* block_vertex_offset, const uint32_t* block_triangle_offset,
                     uint32_t* vertex_base, float* vertices, int* triangles, void* stream);

/* Per-stage GPU time, averaged over every gsr_forward / gsr_backward call made in this process (any thread) since
 * gsr_set_profiling(1): milliseconds for {preprocess, scan (tile histogram + scans + row offsets), scatter, sort, composite} (forward)
 * or {composite_bwd, preprocess_bwd} (backward), measured with HIP events recorded on the launch stream.
 * Recording costs one event per stage boundary (nine per forward + backward: ~3 % of a 1-ms step) and no synchronisation; the
 * getters synchronise on the last recorded event and return the number of calls averaged (0 = nothing recorded).
 * gsr_set_profiling(2) records the two boundaries of composite_fwd only (the other stages then read 0); 0 switches it off. */
void gsr_set_profiling(int enable);
int gsr_last_forward_ms(float ms[5]);
int gsr_last_backward_ms(float ms[2]);

#ifdef __cplusplus
}
#endif
#endif /* GSRAST_H_INCLUDED */
