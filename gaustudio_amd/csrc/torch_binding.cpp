// torch_binding.cpp -- the native torch adapter over the C ABI of libgsrast.so (include/gsrast.h).
//
// Replaces $RAST/rasterize_points.cu:35-231 + $RAST/ext.cpp:15-19: same three entry points, same argument
// order, same return tuples.  Host code only (no device code, no hipify): shape checks, output allocation
// through torch's caching allocator, torch's CURRENT HIP stream.  Built by `make -C gaustudio_amd/csrc` into
// gaustudio_amd/_Cnative*.so and loaded by gaustudio_amd/_C.py.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <algorithm>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gsrast.h"

namespace {

// replaces resizeFunctional (rasterize_points.cu:27-33)
char* resize_cb(void* ctx, size_t n)
{
	auto* t = static_cast<torch::Tensor*>(ctx);
	try {
		// a grow replaces the storage instead of resize_()-ing it: resize_ would copy the old (dead) contents device to
		// device -- the binning buffer is asked for twice when the speculated capacity turned out too small
		if ((size_t)t->numel() < n) *t = torch::empty({(long long)n}, t->options());
		else t->resize_({(long long)n});
	} catch (...) {
		return nullptr;
	}
	return reinterpret_cast<char*>(t->data_ptr());
}

// "absent" = empty tensor (the reference passes torch.Tensor([]) whose data_ptr is null, __init__.py:200-210)
const float* fptr(const torch::Tensor& t, const char* name)
{
	if (t.numel() == 0) return nullptr;
	TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32 (got ", t.scalar_type(), ")");
	TORCH_CHECK(t.is_contiguous(), name, " must be contiguous here (made so by the caller)");
	return t.data_ptr<float>();
}

void require_device(const torch::Tensor& t, const char* name)
{
	TORCH_CHECK(t.is_cuda(), name, " is on '", t.device(), "': gaustudio_amd runs on ROCm devices only "
	            "(hand-written HIP kernels, no CPU fallback)");
}

[[noreturn]] void fail(int rc)
{
	throw std::runtime_error(std::string(gsr_last_error()) + " [gsrast rc=" + std::to_string(rc) + "]");
}

// per-call options (include/gsrast.h gsr_options) as Python hands them over: [tight_binning, cull, fwd_variant,
// bwd_variant, speculative, tile_row_lo, tile_row_hi, fast_exp, forward_only], -1 = process default; a shorter (or empty) list
// leaves the remaining fields at their defaults
gsr_options make_options(const std::vector<int>& v)
{
	gsr_options o;
	gsr_options_init(&o);
	int32_t* f[9] = {&o.tight_binning, &o.cull, &o.fwd_variant, &o.bwd_variant, &o.speculative, &o.tile_row_lo, &o.tile_row_hi,
	                 &o.fast_exp, &o.forward_only};
	for (size_t i = 0; i < v.size() && i < 9; i++) *f[i] = v[i];
	return o;
}

// The image size of a backward: from whichever upstream gradient is present ([C,H,W]); an ABSENT one (empty tensor: the loss
// does not use that output -- its gradient is zero, nothing is read for it) carries none, so the adapters may also pass the
// size explicitly (image_height / image_width, > 0).
void backward_image_size(const torch::Tensor* const g[4], int image_height, int image_width, int& H, int& W)
{
	H = image_height; W = image_width;
	for (int i = 0; i < 4 && (H <= 0 || W <= 0); i++)
		if (g[i]->numel() != 0) {
			TORCH_CHECK(g[i]->ndimension() == 3, "upstream gradients must have dimensions (channels, height, width)");
			H = (int)g[i]->size(1); W = (int)g[i]->size(2);
		}
	TORCH_CHECK(H > 0 && W > 0, "rasterize_gaussians_backward: every upstream gradient is absent and no image_height / image_width was given");
}

void* current_stream(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

// One-shot output arena for the NEXT backward of ONE specific graph (gaustudio_amd/parallel.py): five caller-owned
// tensors [dL_dmeans3D, dL_dsh, dL_dopacity, dL_dscales, dL_drotations], typically slices of one flat all-reduce
// buffer, so that the gradients are born where the collective reads them (no pack copy).  The arena is keyed to the
// data pointers of the operator inputs it was armed for (means3D, sh, scales, rotations -- the ones the backward
// sees; 0 = wildcard): an unrelated backward that happens to have the same P (a second model, an eval pass,
// torch.autograd.grad) does not match and allocates as usual.  All five slots are validated up front; on any
// mismatch the arena is left armed and untouched.  Optionally the SH stage of that backward runs in `sh_chunks`
// Gaussian ranges and `chunk_hook(c, g0, g1)` is called after chunk c has been enqueued, which lets the caller start
// the all-reduce of that slice of dL_dsh while the next chunk computes.
struct GradArena {
	std::vector<torch::Tensor> outs;
	std::vector<int64_t> keys;   // data_ptr of [means3D, sh, scales, rotations]
	int sh_chunks = 1;
	py::object hook;             // None or callable
	// factored gradient exchange (gaustudio_amd/parallel.py FactoredGradExchange): when defined, that backward runs its
	// SH stage in the GSR_BWD_PART_SH_COLORS form -- the clamp-masked colour gradient [P,3] is written HERE (a slot of
	// the all-gather buffer) and no dL_dsh is produced (the binding returns None for it).  With a hook, the backward runs as
	// two calls (GSR_BWD_PART_COLORS_EARLY): the geometry stage already leaves dRGB in the slot, `hook()` is called -- the
	// caller starts the all-gather of this view's slot -- and only then the SH-direction stage is enqueued
	torch::Tensor colors_out;
	// BANDED form of that (round 6; FactoredGradExchange(bands=2)): band_split > 0 = the tile row the backward is cut at.  (1) The
	// FORWARD of these parameters tells the two Gaussian classes of the cut apart and hands them to class_hook(first[P], second[P]) --
	// the caller builds the headers of the view's two colour messages from them --, (2) the backward runs the first band (compositing of the
	// tile rows above the split + the per-Gaussian stage of the Gaussians that end there) and calls band_hook(): their dRGB rows are
	// final, the caller packs and all-gathers them WHILE (3) the second band runs; then hook() and the SH-direction stage as above
	int band_split = 0;
	py::object band_hook, class_hook;
};
GradArena& g_arena = *new GradArena();   // never destroyed: holds a Python object, must not outlive the interpreter's teardown
std::mutex g_arena_mutex;                // armed on the caller's thread, consumed on an autograd worker thread
// A backward whose SH chunks were reduced from inside it (sh_chunks > 1) leaves gradients that are PARTLY summed over the
// ranks already: a second local backward of the same parameters before the tail reduction would make autograd add a
// local view into reduced data -- silently wrong sums.  The keys of such a backward stay "pending" until the caller
// finishes the reduction (any set_grad_arena call, which parallel.allreduce_gaussian_grads issues); a backward of the
// same parameters in between is refused.
int64_t g_pending_keys[4] = {0, 0, 0, 0};
bool g_pending = false;

bool arena_matches(const GradArena& a, int64_t P, int64_t M, const torch::TensorOptions& fo, const int64_t keys[4])
{
	const bool factored = a.colors_out.defined();
	if (a.outs.size() != 5 && !(factored && a.outs.empty())) return false;
	if (factored) {
		const torch::Tensor& t = a.colors_out;
		if (t.sizes() != at::IntArrayRef({P, 3}) || t.device() != fo.device() || t.scalar_type() != torch::kFloat32 || !t.is_contiguous())
			return false;
	}
	const std::vector<int64_t> shapes[5] = {{P, 3}, {P, M, 3}, {P, 1}, {P, 3}, {P, 4}};
	for (int i = 0; i < (int)a.outs.size(); i++) {
		if (factored && i == 1) continue;   // no dL_dsh in the factored form
		const torch::Tensor& t = a.outs[i];
		if (!t.defined() || t.sizes() != at::IntArrayRef(shapes[i]) || t.device() != fo.device() ||
		    t.scalar_type() != torch::kFloat32 || !t.is_contiguous())
			return false;
	}
	for (size_t i = 0; i < a.keys.size() && i < 4; i++)
		if (a.keys[i] != 0 && a.keys[i] != keys[i]) return false;
	return true;
}

void check_rows(const torch::Tensor& t, int64_t P, int64_t per_row, const char* name)
{
	if (t.numel() == 0) return;   // absent
	TORCH_CHECK(t.numel() == P * per_row, name, " must hold ", per_row, " values for each of the ", P, " Gaussians (got ",
	            t.numel(), " elements)");
}

void check_small(const torch::Tensor& t, int64_t n, const char* name)
{
	TORCH_CHECK(t.numel() >= n, name, " must hold at least ", n, " values (got ", t.numel(), ")");
	TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
}

}  // namespace

void set_grad_arena(std::vector<torch::Tensor> outs, std::vector<int64_t> keys, int sh_chunks, py::object hook,
                    c10::optional<torch::Tensor> colors_out, int band_split, py::object band_hook, py::object class_hook)
{
	TORCH_CHECK(outs.empty() || outs.size() == 5, "set_grad_arena expects [means3D, sh, opacity, scales, rotations] gradients or []");
	TORCH_CHECK(keys.empty() || keys.size() == 4, "set_grad_arena keys: data_ptr of [means3D, sh, scales, rotations] or []");
	std::lock_guard<std::mutex> lock(g_arena_mutex);
	g_arena.outs = std::move(outs);
	g_arena.keys = std::move(keys);
	g_arena.sh_chunks = sh_chunks > 1 ? sh_chunks : 1;
	g_arena.hook = std::move(hook);
	g_arena.colors_out = colors_out.has_value() ? *colors_out : torch::Tensor();
	g_arena.band_split = band_split > 0 ? band_split : 0;
	g_arena.band_hook = std::move(band_hook);
	g_arena.class_hook = std::move(class_hook);
	g_pending = false;
}

// the other half of the factored exchange: dL_dsh[P,M,3] from every view's colour gradients (include/gsrast.h)
void sh_grad_from_colors(const torch::Tensor& means3D, const torch::Tensor& campos, const torch::Tensor& colors, int degree,
                         torch::Tensor& dL_dsh)
{
	require_device(means3D, "means3D");
	const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
	const int64_t P = means3D.size(0);
	TORCH_CHECK(dL_dsh.dim() == 3 && dL_dsh.size(0) == P && dL_dsh.size(2) == 3 && dL_dsh.is_contiguous() && dL_dsh.is_cuda() &&
	            dL_dsh.scalar_type() == torch::kFloat32, "dL_dsh must be a contiguous float32 [P,M,3] device tensor");
	TORCH_CHECK(colors.dim() == 3 && colors.size(1) == P && colors.size(2) == 3 && colors.is_contiguous() && colors.is_cuda() &&
	            colors.scalar_type() == torch::kFloat32, "colors must be a contiguous float32 [N,P,3] device tensor");
	const int64_t N = colors.size(0);
	TORCH_CHECK(campos.numel() == N * 3 && campos.is_cuda() && campos.is_contiguous() && campos.scalar_type() == torch::kFloat32,
	            "campos must be a contiguous float32 [N,3] device tensor");
	const auto m = means3D.contiguous();
	const int rc = gsr_sh_grad_from_colors((int)P, degree, (int)dL_dsh.size(1), (int)N, fptr(m, "means3D"), fptr(campos, "campos"),
	                                       fptr(colors, "colors"), dL_dsh.data_ptr<float>(), current_stream(means3D));
	if (rc < 0) fail(rc);
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussians(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& colors_,
                   const torch::Tensor& opacity_, const torch::Tensor& scales_, const torch::Tensor& rotations_,
                   const float scale_modifier, const torch::Tensor& cov3D_precomp_, const torch::Tensor& viewmatrix_,
                   const torch::Tensor& projmatrix_, const float tan_fovx, const float tan_fovy, const int image_height,
                   const int image_width, const torch::Tensor& sh_, const int degree, const torch::Tensor& campos_,
                   const bool prefiltered, const bool debug, const std::vector<int>& options)
{
	if (means3D_.ndimension() != 2 || means3D_.size(1) != 3) {
		AT_ERROR("means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:57-59
	}
	require_device(means3D_, "means3D");
	const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D_.device());
	const int P = means3D_.size(0), H = image_height, W = image_width;
	const auto means3D = means3D_.contiguous(), colors = colors_.contiguous(), opacity = opacity_.contiguous();
	const auto scales = scales_.contiguous(), rotations = rotations_.contiguous();
	const auto cov3D_precomp = cov3D_precomp_.contiguous(), sh = sh_.contiguous();
	const auto viewmatrix = viewmatrix_.contiguous(), projmatrix = projmatrix_.contiguous();
	const auto campos = campos_.contiguous(), bg = background.contiguous();
	if (colors.numel() != 0 && (colors.ndimension() != 2 || colors.size(1) != 3))
		AT_ERROR("colors_precomp must have dimensions (num_points, 3)");   // NUM_CHANNELS == 3, config.h:15
	// the reference checks means3D only and lets the kernels read out of bounds; malformed input is an error here
	check_rows(colors, P, 3, "colors_precomp");
	// (PCDRenderer passes torch.ones_like(xyz), i.e. [P,3], as opacities -- renderers/pcd_renderer.py:26 -- and the
	// reference reads its first P floats: at least P values, not exactly P)
	TORCH_CHECK(opacity.numel() >= P, "opacities must hold at least one value per Gaussian (got ", opacity.numel(), " for ", P, ")");
	check_rows(scales, P, 3, "scales");
	check_rows(rotations, P, 4, "rotations");
	check_rows(cov3D_precomp, P, 6, "cov3D_precomp");
	if (sh.numel() != 0) {
		TORCH_CHECK(sh.ndimension() == 3 && sh.size(0) == P && sh.size(2) == 3, "sh must have dimensions (num_points, M, 3)");
	}
	check_small(viewmatrix, 16, "viewmatrix");
	check_small(projmatrix, 16, "projmatrix");
	check_small(campos, 3, "campos");
	check_small(bg, 3, "bg");
	TORCH_CHECK(H > 0 && W > 0, "image size must be positive");
	for (const auto& p : {std::make_pair(&colors, "colors_precomp"), std::make_pair(&opacity, "opacities"),
	                      std::make_pair(&scales, "scales"), std::make_pair(&rotations, "rotations"),
	                      std::make_pair(&cov3D_precomp, "cov3D_precomp"), std::make_pair(&sh, "sh")})
		if (p.first->numel() != 0) require_device(*p.first, p.second);

	const auto fo = means3D.options().dtype(torch::kFloat32);
	torch::Tensor out_color = torch::empty({3, H, W}, fo);      // fully overwritten by the library
	torch::Tensor out_depth = torch::empty({1, H, W}, fo);
	torch::Tensor out_median = torch::empty({3, H, W}, fo);
	torch::Tensor out_opacity = torch::empty({1, H, W}, fo);
	torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
	const auto bo = torch::TensorOptions(torch::kByte).device(means3D.device());
	torch::Tensor geom = torch::empty({0}, bo), binning = torch::empty({0}, bo), img = torch::empty({0}, bo);
	const int M = (sh.numel() != 0 && sh.size(0) != 0) ? (int)sh.size(1) : 0;   // rasterize_points.cu:86-90

	const gsr_options opt = make_options(options);
	const int rc = gsr_forward_ex(&opt, resize_cb, &geom, resize_cb, &binning, resize_cb, &img, P, degree, M, fptr(bg, "bg"), W, H,
	                           fptr(means3D, "means3D"), fptr(sh, "sh"), nullptr, fptr(colors, "colors_precomp"),
	                           fptr(opacity, "opacities"), fptr(scales, "scales"), scale_modifier,
	                           fptr(rotations, "rotations"), fptr(cov3D_precomp, "cov3D_precomp"), 0,
	                           fptr(viewmatrix, "viewmatrix"), fptr(projmatrix, "projmatrix"), fptr(campos, "campos"),
	                           tan_fovx, tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(),
	                           out_depth.data_ptr<float>(), out_median.data_ptr<float>(), out_opacity.data_ptr<float>(),
	                           P ? radii.data_ptr<int>() : nullptr, debug ? 1 : 0, current_stream(means3D));
	if (rc < 0) fail(rc);
	// A BANDED backward is armed for these parameters (set_grad_arena(band_split > 0, class_hook)): the two Gaussian classes of its cut
	// are known as soon as the forward is -- computed here and handed to class_hook(first, second) on the CALLER'S thread, right
	// behind the forward (as FactoredGradExchange.visible() gets the radii of an unbanded view): the caller's headers, count gathers
	// and their copy to the host are long finished when the backward's first band reports.  The arena itself stays armed.
	if (P != 0) {
		py::object hook = py::none();
		int S = 0;
		{
			const int64_t keys[4] = {(int64_t)(uintptr_t)means3D_.data_ptr(), (int64_t)(uintptr_t)sh_.data_ptr(),
			                         (int64_t)(uintptr_t)scales_.data_ptr(), (int64_t)(uintptr_t)rotations_.data_ptr()};
			std::lock_guard<std::mutex> lock(g_arena_mutex);
			bool match = g_arena.band_split > 0 && !g_arena.class_hook.is_none() && g_arena.colors_out.defined();
			for (size_t i = 0; match && i < g_arena.keys.size() && i < 4; i++)
				if (g_arena.keys[i] != 0 && g_arena.keys[i] != keys[i]) match = false;
			if (match) { hook = g_arena.class_hook; S = g_arena.band_split; }
		}
		if (!hook.is_none()) {
			const auto io = means3D.options().dtype(torch::kInt32);
			torch::Tensor first = torch::empty({P}, io), second = torch::empty({P}, io);
			const int rc2 = gsr_band_classes(P, radii.data_ptr<int>(), reinterpret_cast<const char*>(geom.data_ptr()), S, first.data_ptr<int>(),
			                                 second.data_ptr<int>(), current_stream(means3D));
			if (rc2 < 0) fail(rc2);
			hook(first, second);
		}
	}
	return std::make_tuple(rc, out_color, out_depth, out_median, out_opacity, radii, geom, binning, img);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackward(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& radii_,
                           const torch::Tensor& colors_, const torch::Tensor& scales_, const torch::Tensor& rotations_,
                           const float scale_modifier, const torch::Tensor& cov3D_precomp_,
                           const torch::Tensor& viewmatrix_, const torch::Tensor& projmatrix_, const float tan_fovx,
                           const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_depth,
                           const torch::Tensor& dL_dout_median_depth, const torch::Tensor& dL_dout_final_opacity,
                           const torch::Tensor& sh_, const int degree, const torch::Tensor& campos_,
                           const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer,
                           const torch::Tensor& imageBuffer, const bool debug, const std::vector<int>& options,
                           const int image_height, const int image_width)
{
	require_device(means3D_, "means3D");
	const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D_.device());
	const gsr_options opt = make_options(options);
	const int P = means3D_.size(0);
	int H, W;
	{
		const torch::Tensor* const g[4] = {&dL_dout_color, &dL_dout_depth, &dL_dout_median_depth, &dL_dout_final_opacity};
		backward_image_size(g, image_height, image_width, H, W);
	}
	const auto means3D = means3D_.contiguous(), colors = colors_.contiguous(), scales = scales_.contiguous();
	const auto rotations = rotations_.contiguous(), cov3D_precomp = cov3D_precomp_.contiguous(), sh = sh_.contiguous();
	const auto viewmatrix = viewmatrix_.contiguous(), projmatrix = projmatrix_.contiguous();
	const auto campos = campos_.contiguous(), bg = background.contiguous(), radii = radii_.contiguous();
	const auto g_color = dL_dout_color.contiguous(), g_depth = dL_dout_depth.contiguous();
	const auto g_median = dL_dout_median_depth.contiguous(), g_op = dL_dout_final_opacity.contiguous();
	const int M = (sh.numel() != 0 && sh.size(0) != 0) ? (int)sh.size(1) : 0;

	const auto fo = means3D.options().dtype(torch::kFloat32);
	for (const auto& p : {std::make_pair(&g_color, (int64_t)3), std::make_pair(&g_depth, (int64_t)1),
	                      std::make_pair(&g_median, (int64_t)3), std::make_pair(&g_op, (int64_t)1)})
		TORCH_CHECK(p.first->numel() == 0 || (p.first->numel() == p.second * H * W && p.first->is_cuda()), "upstream gradients must be "
		            "device tensors of the rendered image size (", H, "x", W, "), or empty (absent = zero)");
	check_rows(colors, P, 3, "colors_precomp");
	check_rows(scales, P, 3, "scales");
	check_rows(rotations, P, 4, "rotations");
	check_rows(cov3D_precomp, P, 6, "cov3D_precomp");
	check_rows(radii, P, 1, "radii");
	if (sh.numel() != 0)
		TORCH_CHECK(sh.ndimension() == 3 && sh.size(0) == P && sh.size(2) == 3, "sh must have dimensions (num_points, M, 3)");
	check_small(bg, 3, "bg");
	TORCH_CHECK(R >= 0, "num_rendered must be non-negative");
	if (P != 0) {
		TORCH_CHECK(geomBuffer.is_cuda() && imageBuffer.is_cuda() && binningBuffer.is_cuda(), "the three opaque buffers must be the "
		            "device tensors rasterize_gaussians returned");
		TORCH_CHECK((size_t)geomBuffer.numel() >= gsr_geometry_bytes(P) && (size_t)imageBuffer.numel() >= gsr_image_bytes(W, H),
		            "geomBuffer / imgBuffer are smaller than what a forward with P = ", P, ", ", W, "x", H, " produces");
	}
	// torch::empty: the library writes every row (zeros for culled Gaussians); the reference needed torch::zeros
	GradArena arena;
	{
		const int64_t keys[4] = {(int64_t)(uintptr_t)means3D_.data_ptr(), (int64_t)(uintptr_t)sh_.data_ptr(),
		                         (int64_t)(uintptr_t)scales_.data_ptr(), (int64_t)(uintptr_t)rotations_.data_ptr()};
		std::lock_guard<std::mutex> lock(g_arena_mutex);
		TORCH_CHECK(!(g_pending && std::equal(keys, keys + 4, g_pending_keys)),
		            "rasterize_gaussians_backward: the previous backward of these parameters reduced its SH gradients chunk by chunk "
		            "(FlatGradBucket.arm(overlap_chunks > 1)) and the tail reduction has not run yet: another local backward now would "
		            "be accumulated into partly reduced gradients.  Use overlap_chunks only for the LAST local view of a step.");
		if (arena_matches(g_arena, P, M, fo, keys)) {   // one-shot, and only for the graph it was armed for
			arena = std::move(g_arena);
			g_arena = GradArena();
			// "pending" only when this backward really reduces chunk by chunk (the condition of `chunked` below): without SH
			// coefficients, with precomputed colours or with P = 0 the hook is never called, nothing is partly reduced and
			// nothing would ever clear the flag
			if (arena.sh_chunks > 1 && !arena.hook.is_none() && arena.outs.size() == 5 && !arena.colors_out.defined() && M > 0 &&
			    sh.numel() != 0 && P != 0) {
				std::copy(keys, keys + 4, g_pending_keys);
				g_pending = true;
			}
		}
	}
	const bool in_arena = arena.outs.size() == 5;
	const bool factored = arena.colors_out.defined() && M > 0 && sh.numel() != 0;
	torch::Tensor dL_dmeans3D = in_arena ? arena.outs[0] : torch::empty({P, 3}, fo), dL_dmeans2D = torch::empty({P, 3}, fo);
	// dL_dcov3D only when the caller supplied covariances (rasterize_points.cu:129 allocates it always; with scale / rotation
	// inputs nothing reads it: 24 B per Gaussian less to write)
	const bool want_dcov = cov3D_precomp.numel() != 0;
	torch::Tensor dL_dcolors = factored ? arena.colors_out : torch::empty({P, 3}, fo);
	torch::Tensor dL_dcov3D = want_dcov ? torch::empty({P, 6}, fo) : torch::empty({0, 6}, fo);
	torch::Tensor dL_dsh = factored ? torch::Tensor() : (in_arena ? arena.outs[1] : torch::empty({P, M, 3}, fo));
	torch::Tensor dL_dopacity = in_arena ? arena.outs[2] : torch::empty({P, 1}, fo);
	torch::Tensor dL_dscales = in_arena ? arena.outs[3] : torch::empty({P, 3}, fo);
	torch::Tensor dL_drotations = in_arena ? arena.outs[4] : torch::empty({P, 4}, fo);
	if (P != 0) {
		const auto bo = torch::TensorOptions(torch::kByte).device(means3D.device());
		torch::Tensor scratch = torch::empty({(long long)gsr_backward_scratch_bytes(P, R)}, bo);
		auto run = [&](int parts, int g0, int g1) {
			const int rc = gsr_backward_ex(
			    &opt, parts, g0, g1, P, degree, M, R, fptr(bg, "bg"), W, H, fptr(means3D, "means3D"), fptr(sh, "sh"), nullptr,
			    fptr(colors, "colors_precomp"), fptr(scales, "scales"), scale_modifier, fptr(rotations, "rotations"),
			    fptr(cov3D_precomp, "cov3D_precomp"), 0, tan_fovx, tan_fovy, radii.data_ptr<int>(),
			    reinterpret_cast<const char*>(geomBuffer.data_ptr()), reinterpret_cast<const char*>(binningBuffer.data_ptr()),
			    reinterpret_cast<const char*>(imageBuffer.data_ptr()), fptr(g_color, "dL_dout_color"),
			    fptr(g_depth, "dL_dout_depth"), fptr(g_median, "dL_dout_median_depth"), fptr(g_op, "dL_dout_final_opacity"),
			    dL_dmeans2D.data_ptr<float>(), dL_dopacity.data_ptr<float>(), dL_dcolors.data_ptr<float>(),
			    dL_dmeans3D.data_ptr<float>(), want_dcov ? dL_dcov3D.data_ptr<float>() : nullptr, (M && !factored) ? dL_dsh.data_ptr<float>() : nullptr, nullptr,
			    dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(), reinterpret_cast<char*>(scratch.data_ptr()),
			    debug ? 1 : 0, current_stream(means3D));
			if (rc < 0) fail(rc);
		};
		const bool chunked = in_arena && !factored && arena.sh_chunks > 1 && !arena.hook.is_none() && M > 0 && sh.numel() != 0;
		if (factored && !arena.hook.is_none() && arena.band_split > 0 && !arena.band_hook.is_none()) {
			// banded (the classes of the cut went to class_hook behind the forward): the two bands, a callback after each
			const int S = arena.band_split;
			run(GSR_BWD_PART_MAIN | GSR_BWD_PART_SH_COLORS | GSR_BWD_PART_COLORS_EARLY | GSR_BWD_PART_BAND_FIRST, S, 0);
			arena.band_hook();
			run(GSR_BWD_PART_MAIN | GSR_BWD_PART_SH_COLORS | GSR_BWD_PART_COLORS_EARLY | GSR_BWD_PART_BAND_SECOND, S, 0);
			arena.hook();
			run(GSR_BWD_PART_SH | GSR_BWD_PART_SH_COLORS | GSR_BWD_PART_COLORS_EARLY, 0, P);
		} else if (factored && !arena.hook.is_none()) {
			run(GSR_BWD_PART_MAIN | GSR_BWD_PART_SH_COLORS | GSR_BWD_PART_COLORS_EARLY, 0, 0);
			arena.hook();
			run(GSR_BWD_PART_SH | GSR_BWD_PART_SH_COLORS | GSR_BWD_PART_COLORS_EARLY, 0, P);
		} else if (factored) {
			run(GSR_BWD_PART_MAIN | GSR_BWD_PART_SH | GSR_BWD_PART_SH_COLORS, 0, P);
		} else if (!chunked) {
			run(GSR_BWD_PART_MAIN | GSR_BWD_PART_SH, 0, P);
		} else {
			// SH stage in Gaussian ranges (multiples of 256): the hook sees each range as soon as it is enqueued
			run(GSR_BWD_PART_MAIN, 0, 0);
			const int per = (int)(((int64_t)(P + arena.sh_chunks - 1) / arena.sh_chunks + 255) / 256 * 256);
			for (int c = 0, g0 = 0; g0 < P; c++, g0 += per) {
				const int g1 = std::min(P, g0 + per);
				run(GSR_BWD_PART_SH, g0, g1);
				arena.hook(c, g0, g1);
			}
		}
	}
	return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

torch::Tensor markVisible(torch::Tensor& means3D_, torch::Tensor& viewmatrix_, torch::Tensor& projmatrix_)
{
	require_device(means3D_, "means3D");
	const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D_.device());
	const int P = means3D_.size(0);
	const auto means3D = means3D_.contiguous(), viewmatrix = viewmatrix_.contiguous(), projmatrix = projmatrix_.contiguous();
	torch::Tensor present = torch::full({P}, false, means3D.options().dtype(at::kBool));
	if (P != 0) {
		const int rc = gsr_mark_visible(P, fptr(means3D, "means3D"), fptr(viewmatrix, "viewmatrix"),
		                                fptr(projmatrix, "projmatrix"), reinterpret_cast<unsigned char*>(present.data_ptr()),
		                                current_stream(means3D));
		if (rc < 0) fail(rc);
	}
	return present;
}

// ---- fused parameter activations: raw point-cloud attributes in, gradients w.r.t. them out ----
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansRaw(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& f_dc_,
                      const torch::Tensor& f_rest_, const torch::Tensor& raw_opacity_, const torch::Tensor& raw_scales_,
                      const torch::Tensor& raw_rotations_, const float scale_modifier, const int activation_flags,
                      const torch::Tensor& viewmatrix_, const torch::Tensor& projmatrix_, const float tan_fovx,
                      const float tan_fovy, const int image_height, const int image_width, const int degree,
                      const torch::Tensor& campos_, const bool prefiltered, const bool debug, const std::vector<int>& options)
{
	if (means3D_.ndimension() != 2 || means3D_.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
	require_device(means3D_, "means3D");
	const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D_.device());
	const int P = means3D_.size(0), H = image_height, W = image_width;
	const auto means3D = means3D_.contiguous(), f_dc = f_dc_.contiguous(), f_rest = f_rest_.contiguous();
	const auto opacity = raw_opacity_.contiguous(), scales = raw_scales_.contiguous(), rotations = raw_rotations_.contiguous();
	const auto viewmatrix = viewmatrix_.contiguous(), projmatrix = projmatrix_.contiguous();
	const auto campos = campos_.contiguous(), bg = background.contiguous();
	for (const auto& p : {std::make_pair(&f_dc, "f_dc"), std::make_pair(&f_rest, "f_rest"), std::make_pair(&opacity, "opacity"),
	                      std::make_pair(&scales, "scales"), std::make_pair(&rotations, "rotations")})
		if (p.first->numel() != 0) require_device(*p.first, p.second);
	TORCH_CHECK(f_dc.numel() == (int64_t)P * 3, "f_dc must hold 3 values per Gaussian");
	TORCH_CHECK(f_rest.numel() % ((int64_t)std::max(P, 1) * 3) == 0, "f_rest must be [P, M-1, 3]");
	const int M = 1 + (P ? (int)(f_rest.numel() / ((int64_t)P * 3)) : 0);

	const auto fo = means3D.options().dtype(torch::kFloat32);
	torch::Tensor out_color = torch::empty({3, H, W}, fo), out_depth = torch::empty({1, H, W}, fo);
	torch::Tensor out_median = torch::empty({3, H, W}, fo), out_opacity = torch::empty({1, H, W}, fo);
	torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
	const auto bo = torch::TensorOptions(torch::kByte).device(means3D.device());
	torch::Tensor geom = torch::empty({0}, bo), binning = torch::empty({0}, bo), img = torch::empty({0}, bo);
	TORCH_CHECK(P == 0 || M == 1 || f_rest.numel() != 0, "f_rest is required when M > 1");
	const gsr_options opt = make_options(options);
	const int rc = gsr_forward_ex(&opt, resize_cb, &geom, resize_cb, &binning, resize_cb, &img, P, degree, M, fptr(bg, "bg"), W, H,
	                               fptr(means3D, "means3D"), fptr(f_dc, "f_dc"), M > 1 ? fptr(f_rest, "f_rest") : nullptr, nullptr,
	                               fptr(opacity, "opacity"), fptr(scales, "scales"), scale_modifier,
	                               fptr(rotations, "rotations"), nullptr, activation_flags, fptr(viewmatrix, "viewmatrix"),
	                               fptr(projmatrix, "projmatrix"), fptr(campos, "campos"), tan_fovx, tan_fovy,
	                               prefiltered ? 1 : 0, out_color.data_ptr<float>(), out_depth.data_ptr<float>(),
	                               out_median.data_ptr<float>(), out_opacity.data_ptr<float>(),
	                               P ? radii.data_ptr<int>() : nullptr, debug ? 1 : 0, current_stream(means3D));
	if (rc < 0) fail(rc);
	return std::make_tuple(rc, out_color, out_depth, out_median, out_opacity, radii, geom, binning, img);
}

// returns (dL_dmeans2D, dL_draw_opacity, dL_dmeans3D, dL_df_dc, dL_df_rest, dL_draw_scales, dL_draw_rotations)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansRawBackward(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& radii_,
                              const torch::Tensor& f_dc_, const torch::Tensor& f_rest_, const torch::Tensor& raw_scales_,
                              const torch::Tensor& raw_rotations_, const float scale_modifier, const int activation_flags,
                              const float tan_fovx, const float tan_fovy, const torch::Tensor& dL_dout_color,
                              const torch::Tensor& dL_dout_depth, const torch::Tensor& dL_dout_median_depth,
                              const torch::Tensor& dL_dout_final_opacity, const int degree,
                              const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer,
                              const torch::Tensor& imageBuffer, const bool debug, const std::vector<int>& options,
                              const int image_height, const int image_width)
{
	require_device(means3D_, "means3D");
	const c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D_.device());
	const int P = means3D_.size(0);
	int H, W;
	{
		const torch::Tensor* const g[4] = {&dL_dout_color, &dL_dout_depth, &dL_dout_median_depth, &dL_dout_final_opacity};
		backward_image_size(g, image_height, image_width, H, W);
	}
	const auto means3D = means3D_.contiguous(), f_dc = f_dc_.contiguous(), f_rest = f_rest_.contiguous();
	const auto scales = raw_scales_.contiguous(), rotations = raw_rotations_.contiguous(), bg = background.contiguous();
	const auto radii = radii_.contiguous();
	const auto g_color = dL_dout_color.contiguous(), g_depth = dL_dout_depth.contiguous();
	const auto g_median = dL_dout_median_depth.contiguous(), g_op = dL_dout_final_opacity.contiguous();
	const int M = 1 + (P ? (int)(f_rest.numel() / ((int64_t)P * 3)) : 0);
	const auto fo = means3D.options().dtype(torch::kFloat32);
	torch::Tensor dL_dmeans3D = torch::empty({P, 3}, fo), dL_dmeans2D = torch::empty({P, 3}, fo);
	torch::Tensor dL_dcolors = torch::empty({P, 3}, fo), dL_dopacity = torch::empty_like(raw_scales_.new_empty({P, 1}));
	torch::Tensor dL_df_dc = torch::empty_like(f_dc), dL_df_rest = torch::empty_like(f_rest);
	torch::Tensor dL_dscales = torch::empty({P, 3}, fo), dL_drotations = torch::empty({P, 4}, fo);
	if (P != 0) {
		const auto bo = torch::TensorOptions(torch::kByte).device(means3D.device());
		torch::Tensor scratch = torch::empty({(long long)gsr_backward_scratch_bytes(P, R)}, bo);
		const gsr_options opt = make_options(options);
		const int rc = gsr_backward_ex(
		    &opt, GSR_BWD_PART_MAIN | GSR_BWD_PART_SH, 0, P, P, degree, M, R, fptr(bg, "bg"), W, H, fptr(means3D, "means3D"),
		    fptr(f_dc, "f_dc"), M > 1 ? fptr(f_rest, "f_rest") : nullptr, nullptr,
		    fptr(scales, "scales"), scale_modifier, fptr(rotations, "rotations"), nullptr, activation_flags, tan_fovx, tan_fovy,
		    radii.data_ptr<int>(), reinterpret_cast<const char*>(geomBuffer.data_ptr()),
		    reinterpret_cast<const char*>(binningBuffer.data_ptr()), reinterpret_cast<const char*>(imageBuffer.data_ptr()),
		    fptr(g_color, "dL_dout_color"), fptr(g_depth, "dL_dout_depth"), fptr(g_median, "dL_dout_median_depth"),
		    fptr(g_op, "dL_dout_final_opacity"), dL_dmeans2D.data_ptr<float>(), dL_dopacity.data_ptr<float>(),
		    dL_dcolors.data_ptr<float>(), dL_dmeans3D.data_ptr<float>(), nullptr /* dL_dcov3D: no reader */,
		    dL_df_dc.data_ptr<float>(), M > 1 ? dL_df_rest.data_ptr<float>() : nullptr, dL_dscales.data_ptr<float>(),
		    dL_drotations.data_ptr<float>(), reinterpret_cast<char*>(scratch.data_ptr()), debug ? 1 : 0,
		    current_stream(means3D));
		if (rc < 0) fail(rc);
	}
	return std::make_tuple(dL_dmeans2D, dL_dopacity, dL_dmeans3D, dL_df_dc, dL_df_rest, dL_dscales, dL_drotations);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
	// the reference's positional signatures, plus optional trailing arguments: the per-call options (gsr_options) and, for the
	// backwards, the image size (needed only when every upstream gradient is absent = an empty tensor)
	const std::vector<int> no_opts;
	m.def("rasterize_gaussians", &RasterizeGaussians, py::arg("background"), py::arg("means3D"), py::arg("colors"),
	      py::arg("opacity"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"),
	      py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("image_height"),
	      py::arg("image_width"), py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("prefiltered"), py::arg("debug"),
	      py::arg("options") = no_opts);
	m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackward, py::arg("background"), py::arg("means3D"),
	      py::arg("radii"), py::arg("colors"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"),
	      py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"),
	      py::arg("dL_dout_color"), py::arg("dL_dout_depth"), py::arg("dL_dout_median_depth"), py::arg("dL_dout_final_opacity"),
	      py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("geomBuffer"), py::arg("R"), py::arg("binningBuffer"),
	      py::arg("imageBuffer"), py::arg("debug"), py::arg("options") = no_opts, py::arg("image_height") = -1,
	      py::arg("image_width") = -1);
	m.def("mark_visible", &markVisible);
	m.def("set_grad_arena", &set_grad_arena, py::arg("outs"), py::arg("keys") = std::vector<int64_t>(), py::arg("sh_chunks") = 1,
	      py::arg("hook") = py::none(), py::arg("colors_out") = py::none(), py::arg("band_split") = 0, py::arg("band_hook") = py::none(),
	      py::arg("class_hook") = py::none());
	m.def("sh_grad_from_colors", &sh_grad_from_colors, py::arg("means3D"), py::arg("campos"), py::arg("colors"), py::arg("degree"),
	      py::arg("dL_dsh"));
	m.def("rasterize_gaussians_raw", &RasterizeGaussiansRaw, py::arg("background"), py::arg("means3D"), py::arg("f_dc"),
	      py::arg("f_rest"), py::arg("raw_opacity"), py::arg("raw_scales"), py::arg("raw_rotations"), py::arg("scale_modifier"),
	      py::arg("activation_flags"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"),
	      py::arg("image_height"), py::arg("image_width"), py::arg("degree"), py::arg("campos"), py::arg("prefiltered"),
	      py::arg("debug"), py::arg("options") = no_opts);
	m.def("rasterize_gaussians_raw_backward", &RasterizeGaussiansRawBackward, py::arg("background"), py::arg("means3D"),
	      py::arg("radii"), py::arg("f_dc"), py::arg("f_rest"), py::arg("raw_scales"), py::arg("raw_rotations"),
	      py::arg("scale_modifier"), py::arg("activation_flags"), py::arg("tan_fovx"), py::arg("tan_fovy"),
	      py::arg("dL_dout_color"), py::arg("dL_dout_depth"), py::arg("dL_dout_median_depth"), py::arg("dL_dout_final_opacity"),
	      py::arg("degree"), py::arg("geomBuffer"), py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"),
	      py::arg("debug"), py::arg("options") = no_opts, py::arg("image_height") = -1, py::arg("image_width") = -1);
}
