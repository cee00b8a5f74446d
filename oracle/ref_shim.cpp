// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.  C-ABI driver around the REFERENCE's own rasterizer
// (CudaRasterizer::Rasterizer::{forward,backward,markVisible}, $RAST/cuda_rasterizer/rasterizer.h:20-92),
// whose CUDA sources are translated on the fly with hipify-perl by oracle/build_ref.sh and compiled for
// gfx950 into oracle/_ref/libgsref.so.  Nothing of the reference is copied into this repository: the
// translated files live in a mktemp directory during the build and only the .so is kept (git-ignored).
// It exists so that tests can pin the CPU oracle and the HIP kernels against the reference's actual
// kernels running on an MI355X.  The product never loads it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <functional>
#include "rasterizer.h"   // the (hipified) reference header

namespace {
struct Buf {
	char* p = nullptr;
	size_t n = 0;
	char* get(size_t N)
	{
		if (N > n) {
			if (p) (void)hipFree(p);
			if (hipMalloc((void**)&p, N) != hipSuccess) { p = nullptr; n = 0; return nullptr; }
			n = N;
		}
		return p;
	}
	~Buf() { if (p) (void)hipFree(p); }
};
struct Ctx {
	Buf geom, binning, img;
	int R = 0;
};
}  // namespace

extern "C" {

void* ref_create(void) { return new Ctx; }
void ref_destroy(void* h) { delete static_cast<Ctx*>(h); }

// all pointers are device pointers; NULL = absent (forward.cu:205,241)
int ref_forward(void* h, int P, int D, int M, const float* background, int W, int H, const float* means3D,
                const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, float* out_depth, float* out_median, float* out_opacity, int* radii)
{
	Ctx* c = static_cast<Ctx*>(h);
	std::function<char*(size_t)> g = [c](size_t N) { return c->geom.get(N); };
	std::function<char*(size_t)> b = [c](size_t N) { return c->binning.get(N); };
	std::function<char*(size_t)> i = [c](size_t N) { return c->img.get(N); };
	// the torch glue zero-fills the outputs first (rasterize_points.cu:68-72)
	const size_t HW = (size_t)W * H;
	(void)hipMemset(out_color, 0, 3 * HW * 4); (void)hipMemset(out_depth, 0, HW * 4);
	(void)hipMemset(out_median, 0, 3 * HW * 4); (void)hipMemset(out_opacity, 0, HW * 4);
	(void)hipMemset(radii, 0, (size_t)P * 4);
	c->R = CudaRasterizer::Rasterizer::forward(g, b, i, P, D, M, background, W, H, means3D, shs, colors_precomp,
	                                           opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
	                                           projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered != 0, out_color,
	                                           out_depth, out_median, out_opacity, radii, false);
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	return c->R;
}

// gradient outputs must be ZEROED by the caller (rasterize_points.cu:160-169)
int ref_backward(void* h, int P, int D, int M, const float* background, int W, int H, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_median,
                 const float* dL_dpix_opacity, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                 float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot)
{
	Ctx* c = static_cast<Ctx*>(h);
	CudaRasterizer::Rasterizer::backward(P, D, M, c->R, background, W, H, means3D, shs, colors_precomp, scales,
	                                     scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos,
	                                     tan_fovx, tan_fovy, radii, c->geom.p, c->binning.p, c->img.p, dL_dpix,
	                                     dL_dpix_depth, dL_dpix_median, dL_dpix_opacity, dL_dmean2D, dL_dconic,
	                                     dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
	                                     dL_drot, false);
	return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

int ref_mark_visible(int P, float* means3D, float* viewmatrix, float* projmatrix, bool* present)
{
	CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, present);
	return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

}  // extern "C"
