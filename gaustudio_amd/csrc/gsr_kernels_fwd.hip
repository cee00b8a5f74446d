// gsr_kernels_fwd.hip -- forward kernels of libgsrast for gfx950 (MI355X, wave64).
//
//   preprocess_fwd   one lane per Gaussian: cull, project, cov3D, EWA cov2D, conic, radius, tile rect,
//                    SH->RGB; writes one 64-B record per visible Gaussian and its tile count
//                    (replaces forward.cu:155-256 preprocessCUDA)
//   bin_chunk<hist>  per-chunk tile histograms in LDS, bin_colscan turns them into per-chunk offsets + tile totals
//   tile_scan        exclusive scan of the per-tile counts -> per-tile [start,end) ranges, R
//                    (replaces cub InclusiveSum over P + identifyTileRanges, rasterizer_impl.cu:280,116-138)
//   bin_chunk<scat>  emits (depth bits<<32 | id) keys straight into each tile's segment, slots from LDS cursors
//                    (replaces duplicateWithKeys, rasterizer_impl.cu:70-111; bin_scatter = global-atomic fallback
//                    for tile grids too large for an LDS histogram)
//   tile_sort        one WAVE per tile sorts its segment by (depth, id) in registers; tile_radix_sort for lists
//                    longer than 1024 keys
//                    (replaces the global 45..47-bit cub::DeviceRadixSort, rasterizer_impl.cu:306-311)
//   composite_fwd    one workgroup (4 waves, one 8x8 pixel block each) per 16x16 tile, front-to-back alpha
//                    compositing with per-wave ballot culling (replaces forward.cu:261-397 renderCUDA)
#include "gsr_internal.h"

namespace gsr {

__device__ __constant__ float kSH_C0 = 0.28209479177387814f;
__device__ __constant__ float kSH_C1 = 0.4886025119029199f;
__device__ __constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                            -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                            0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                            -0.5900435899266435f};

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ view,
                                                           unsigned char* __restrict__ present)
{
	int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	float3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
	float3 pv = xform4x3(p, view);
	present[idx] = !(pv.z <= 0.2f);   // auxiliary.h:154 (lateral test is commented out in the reference)
}

void launch_mark_visible(int P, const float* means3D, const float* view, unsigned char* present, hipStream_t s)
{
	if (P <= 0) return;
	hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
}

// ------------------------------------------------------------------------------------------------
// Tile-rect walkers.  A lane whose rect is small walks it alone; rects with more than
// GSR_COOP_TILES tiles are walked by the whole wave (64 tiles per step) so that one screen-filling
// Gaussian does not serialise 63 idle lanes behind thousands of atomics.
#define GSR_COOP_TILES 32

template <typename F>
__device__ __forceinline__ void for_each_tile(bool active, int rminx, int rminy, int rmaxx, int rmaxy, uint32_t dead,
                                              uint32_t pay0, uint32_t pay1, F&& f)
{
	// f(x, y, p0, p1): p0/p1 are the OWNING lane's payload (shuffled in the cooperative path; the
	// shuffles sit outside the divergent tile loop so that the source lane is always active).
	// dead: corner tiles of the rect that are not binned (gs_dead_corners; 0 for rects under 2 x 2).
	const int w = rmaxx - rminx, h = rmaxy - rminy;
	const int n = active ? w * h : 0;
	const bool big = n > GSR_COOP_TILES;
	if (n > 0 && !big) {
		// one loop for all lanes of the wave (lanes with and without dead corners take the same path): a dead corner
		// shortens the first / last row by a tile at its end
		for (int y = rminy; y < rmaxy; y++) {
			const uint32_t drow = y == rminy ? dead : (y == rmaxy - 1 ? dead >> 2 : 0u);
			for (int x = rminx + (int)(drow & 1u); x < rmaxx - (int)((drow >> 1) & 1u); x++) f(x, y, pay0, pay1);
		}
	}
	unsigned long long m = __ballot(big);
	const int lane = threadIdx.x & 63;
	while (m) {
		const int src = __ffsll((long long)m) - 1;
		m &= m - 1;
		const int sx = __shfl(rminx, src, 64), sy = __shfl(rminy, src, 64);
		const int sw = __shfl(w, src, 64), sn = __shfl(n, src, 64);
		const uint32_t sd = (uint32_t)__shfl((int)dead, src, 64);
		const uint32_t p0 = (uint32_t)__shfl((int)pay0, src, 64), p1 = (uint32_t)__shfl((int)pay1, src, 64);
		const int sh = sn / sw;
		for (int t = lane; t < sn; t += 64) {
			const int cx = t % sw, cy = t / sw;
			if (sd != 0u && (cx == 0 || cx == sw - 1) && (cy == 0 || cy == sh - 1) && ((sd >> ((cx == 0 ? 0 : 1) + (cy == 0 ? 0 : 2))) & 1u))
				continue;
			f(sx + cx, sy + cy, p0, p1);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// SH -> RGB for one Gaussian (computeColorFromSH, forward.cu:20-71); returns the clamp mask.
template <int D>
__device__ __forceinline__ uint32_t gs_sh_to_rgb(const float* sh, float3 p_orig, const GsCam* __restrict__ cam, float* rgb)
{
	float3 dir = {p_orig.x - cam->campos[0], p_orig.y - cam->campos[1], p_orig.z - cam->campos[2]};
	const float len = sqrtf(FMA(dir.z, dir.z, FMA(dir.y, dir.y, dir.x * dir.x)));
	dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
	const float x = dir.x, y = dir.y, z = dir.z;
	uint32_t clamped = 0;
#pragma unroll
	for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
		float r = kSH_C0 * SH(0);
		if (D > 0) {
			r = FMA(-(kSH_C1 * y), SH(1), r);
			r = FMA(kSH_C1 * z, SH(2), r);
			r = FMA(-(kSH_C1 * x), SH(3), r);
			if (D > 1) {
				const float xx = x * x, yy = y * y, zz = z * z;
				const float xy = x * y, yz = y * z, xz = x * z;
				r = FMA(kSH_C2[0] * xy, SH(4), r);
				r = FMA(kSH_C2[1] * yz, SH(5), r);
				r = FMA(kSH_C2[2] * (FMA(2.0f, zz, -xx) - yy), SH(6), r);
				r = FMA(kSH_C2[3] * xz, SH(7), r);
				r = FMA(kSH_C2[4] * (xx - yy), SH(8), r);
				if (D > 2) {
					r = FMA(kSH_C3[0] * y * FMA(3.0f, xx, -yy), SH(9), r);
					r = FMA(kSH_C3[1] * xy * z, SH(10), r);
					r = FMA(kSH_C3[2] * y * (FMA(4.0f, zz, -xx) - yy), SH(11), r);
					r = FMA(kSH_C3[3] * z * FMA(-3.0f, yy, FMA(-3.0f, xx, 2.0f * zz)), SH(12), r);
					r = FMA(kSH_C3[4] * x * (FMA(4.0f, zz, -xx) - yy), SH(13), r);
					r = FMA(kSH_C3[5] * z * (xx - yy), SH(14), r);
					r = FMA(kSH_C3[6] * x * FMA(-3.0f, yy, xx), SH(15), r);
				}
			}
		}
#undef SH
		r += 0.5f;
		if (r < 0) clamped |= 1u << ch;
		rgb[ch] = fmaxf(r, 0.0f);
	}
	return clamped;
}

// RAW: the f1 interface (raw attributes: activations in-kernel, SH in two tensors).  A compile-time switch:
// as a run-time one it cost the standard path 40 us at C3.
// Memory-level parallelism is arranged by hand: every per-Gaussian input of the geometry is requested up front; the
// 192-B SH row only once the Gaussian is known to be visible (see `load_sh` below).
#ifndef GSR_PRE_WAVES
#define GSR_PRE_WAVES 4   // waves per SIMD the register allocation is held to (round 5: 5 gains 0.015 ms at C4-inside and loses at C3 / C4, 6 spills)
#endif
#ifndef GSR_PRE_COOP
#define GSR_PRE_COOP 1    // wave-cooperative SH rows through LDS (0: every lane fetches its own row, rounds 1-5)
#endif
#ifndef GSR_PRE_COOP_LD
#define GSR_PRE_COOP_LD(p) gs_ld_stream(p)   // nontemporal: every line is read once, by one instruction (A/B: (*(p)) = plain loads)
#endif
#ifndef GSR_PRE_COOP_MIN
#define GSR_PRE_COOP_MIN 32   // visible Gaussians a wave needs for the cooperative copy; below, each visible lane fetches its own row
#endif
template <int D, bool RAW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GSR_PRE_WAVES, GSR_PRE_WAVES))) void preprocess_fwd_kernel(
    int P, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ shs_rest, int act_arg, const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const GsCam* __restrict__ cam, int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y,
    int gx, int gy, int prefiltered, int sh_vec4, int sh_coop, int tight, int band_lo, int band_hi, int* __restrict__ radii,
    GsRec* __restrict__ recs, float* __restrict__ shjac, uint4* __restrict__ binfo,
    uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ bsums, uint32_t* __restrict__ refsums,
    uint32_t* __restrict__ tile_count, GsCtl* __restrict__ ctl)
{
	constexpr int NC = (D + 1) * (D + 1);
	constexpr int PRE_SLAB = (GSR_PRE_COOP && D == 3 && !RAW) ? 32 * 52 : 4;   // floats per wave: 32 padded coefficient rows (cooperative copy below)
	__shared__ __attribute__((aligned(16))) float s_rows[4][PRE_SLAB];
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const int act = RAW ? act_arg : 0;
	bool vis = false;
	int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
	uint32_t ref_tiles = 0;   // area of the reference's getRect square: what the reference calls tiles_touched
	int mr = 0;
	float3 p_orig = {0.f, 0.f, 0.f};
	float pix_x = 0.f, pix_y = 0.f, conic_x = 0.f, conic_y = 0.f, conic_z = 0.f, depth = 0.f, op_raw = 0.f;
	float sh[NC * 3];
	// the coefficient row of this Gaussian (12 x (D+1)^2 B, up to 192) is requested only once the Gaussian has turned out VISIBLE
	// (round 5).  Rounds 2-4 requested it right after the near-plane test so that it was in flight during the covariance arithmetic
	// (71 against 92 us at C3 in round 2); since the kernel also forms the SH-direction Jacobian (round 4: 115 VGPRs, held to 4 waves per
	// SIMD) the late request is the faster one everywhere -- preprocess C3 0.0691 -> 0.0683 ms, C4 0.344 -> 0.339 -- and a Gaussian in
	// front of the camera but outside the image (a camera inside the scene sees 16 % of a 360-degree capture; half of the rest is in
	// front of it; the reference has no x / y frustum test, auxiliary.h:147-161) no longer fetches 192 B for nothing: C4-inside
	// 0.262 -> 0.169 ms (profiles/r05_preprocess_fwd_experiments.txt: early / estimate-gated / late, 4-7 waves per SIMD)
	auto load_sh = [&]() {
		if (RAW) {
			// split storage (f_dc [P,1,3] + f_rest [P,M-1,3], models/vanilla_sg.py:103-106): no torch.cat copy
			const float* dcp = shs + 3 * (size_t)idx;
			const float* rp = shs_rest + (size_t)idx * (M - 1) * 3;
			sh[0] = dcp[0]; sh[1] = dcp[1]; sh[2] = dcp[2];
#pragma unroll
			for (int i = 3; i < NC * 3; i++) sh[i] = rp[i - 3];
		} else {
			const float* shp = shs + (size_t)idx * M * 3;
			if (sh_vec4 && (NC * 3) % 4 == 0) {
#pragma unroll
				for (int i = 0; i < NC * 3 / 4; i++) {
					const float4 v = reinterpret_cast<const float4*>(shp)[i];   // (not nontemporal: a lane's 12 loads share cache lines)
					sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
				}
			} else {
#pragma unroll
				for (int i = 0; i < NC * 3; i++) sh[i] = shp[i];
			}
		}
	};
	if (idx < P) {
		do {
			const float* view = cam->view;
			const float* proj = cam->proj;
			p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
			float3 sc_raw = {0.f, 0.f, 0.f};
			float4 q_raw = {0.f, 0.f, 0.f, 0.f};
			if (cov3D_precomp == nullptr) {
				sc_raw = {scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]};
				q_raw = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)idx);
			}
			op_raw = opacities[idx];
			// in_frustum (auxiliary.h:139-164)
			const float4 p_hom = xform4x4(p_orig, proj);
			const float p_w = 1.0f / (p_hom.w + 0.0000001f);
			const float p_proj_x = p_hom.x * p_w, p_proj_y = p_hom.y * p_w;
			const float3 p_view = xform4x3(p_orig, view);
			if (p_view.z <= 0.2f) {
				if (prefiltered) ctl->err_prefiltered = 1;
				break;
			}
			float cov3D[6];
			if (cov3D_precomp != nullptr) {
#pragma unroll
				for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
			} else {
				float inv_len;
				const float3 sc = gs_act_scale(sc_raw, act);
				const float4 q = gs_act_rot(q_raw, act, &inv_len);
				cov3d_from_scale_rot(sc, scale_modifier, q, cov3D);
			}
			Cov2D c;
			cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, c);
			const float cov_x = c.cov.m[0][0] + 0.3f;
			const float cov_y = c.cov.m[0][1];
			const float cov_z = c.cov.m[1][1] + 0.3f;
			const float det = FMA(-cov_y, cov_y, cov_x * cov_z);
			if (det == 0.0f) break;
			const float det_inv = 1.f / det;
			conic_x = cov_z * det_inv; conic_y = -cov_y * det_inv; conic_z = cov_x * det_inv;
			const float mid = 0.5f * (cov_x + cov_z);
			const float disc = sqrtf(fmaxf(0.1f, FMA(mid, mid, -det)));
			const float lambda1 = mid + disc, lambda2 = mid - disc;
			const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
			// ndc2Pix in double (auxiliary.h:41-44)
			pix_x = (float)((((double)p_proj_x + 1.0) * (double)W - 1.0) * 0.5);
			pix_y = (float)((((double)p_proj_y + 1.0) * (double)H - 1.0) * 0.5);
			// getRect (auxiliary.h:46-56)
			const int r_ = (int)my_radius;
			rminx = min(gx, max(0, (int)((pix_x - r_) / GSR_BLOCK_X)));
			rminy = min(gy, max(0, (int)((pix_y - r_) / GSR_BLOCK_Y)));
			rmaxx = min(gx, max(0, (int)((pix_x + r_ + GSR_BLOCK_X - 1) / GSR_BLOCK_X)));
			rmaxy = min(gy, max(0, (int)((pix_y + r_ + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y)));
			if ((rmaxx - rminx) * (rmaxy - rminy) == 0) break;
			ref_tiles = (uint32_t)((rmaxx - rminx) * (rmaxy - rminy));
			mr = r_;
			depth = p_view.z;
			vis = true;
		} while (0);
	}
	// wave-uniform choice between the two ways of fetching the coefficient rows (below)
	const unsigned long long vmask = __ballot(vis && colors_precomp == nullptr);
	const bool coop_wave = GSR_PRE_COOP && !RAW && D == 3 && sh_coop && __popcll(vmask) >= GSR_PRE_COOP_MIN;
	if (vis && colors_precomp == nullptr && !coop_wave) load_sh();
#if GSR_PRE_COOP
	// Round 6 (north_star: "coalesced HBM loads of xyz/scale/rot/SH"): WAVE-COOPERATIVE coefficient rows.  With one
	// lane fetching its own 192-B row, each of the lane's twelve dwordx4 loads touches 64 different rows (16 B of each): 768 cache-line
	// requests per wave for 96 lines of data.  Here the wave copies the rows of its 64 consecutive Gaussians as ONE contiguous
	// 12-KiB chunk (twelve fully coalesced 1-KiB loads; a float4 whose row is not visible is not requested -- the late,
	// visibility-gated request of round 5 is kept), transposes it through a padded LDS slab of 32 rows in two halves (a 64-row slab
	// would hold the kernel to 3 waves per SIMD: 13.3 KiB x 16 waves > 160 KiB), and every lane picks its row out of the slab.
	// Only when a row is exactly the (D+1)^2 coefficients used (M == NC, 16-B aligned): lower degrees of a 16-coefficient tensor
	// would over-fetch and keep the per-lane path, and so does a wave with fewer than GSR_PRE_COOP_MIN visible Gaussians (a camera
	// inside the scene sees 16 %: the copy's barriers and LDS passes for ten rows cost more than they save: C4-inside 0.176 ->
	// 0.190 ms without this gate).  Same values in the same registers: no result bit changes.
	// Measured (profiles/r06_preprocess_fwd_experiments.txt): C3 0.0712 -> 0.0688 (plain loads) -> 0.0657 ms (nontemporal);
	// C4 0.373 -> 0.362 -> 0.3446; C5 0.179 -> 0.155; C4-inside 0.176 -> 0.177.  Not kept: the 64-B records of a dense wave through
	// the same slab as four coalesced 1-KiB stores (C3 0.0642, C4 0.363: +5 % where it matters); gates of 16 / 48 instead of 32
	// visible Gaussians (no difference).
	if constexpr (!RAW && D == 3) if (coop_wave) {
		constexpr int RF = NC * 3, RFP = gs_row_stride<RF>(), NV = (RF * 64) / 4 / 64;   // NV float4 per lane for the wave's 64 rows
		static_assert(32 * RFP <= PRE_SLAB, "slab");
		{
			const int lane = threadIdx.x & 63;
			float* slab = s_rows[threadIdx.x >> 6];
			const float4* chunk = reinterpret_cast<const float4*>(shs + (size_t)(blockIdx.x * 256 + (threadIdx.x & ~63)) * RF);
			float4 st[NV];
#pragma unroll
			for (int it = 0; it < NV; it++) {
				const int j = it * 64 + lane;           // float4 j of the chunk lies inside row (4 j) / RF (RF % 4 == 0)
				st[it] = make_float4(0.f, 0.f, 0.f, 0.f);
				if ((vmask >> ((4 * j) / RF)) & 1ull) st[it] = GSR_PRE_COOP_LD(chunk + j);
			}
#pragma unroll
			for (int half = 0; half < 2; half++) {
#pragma unroll
				for (int it = 0; it < NV / 2; it++) {
					const int j = it * 64 + lane;       // float4 j of this half's 32 rows
					const int r = (4 * j) / RF;
					*reinterpret_cast<float4*>(slab + r * RFP + (4 * j - r * RF)) = st[half * (NV / 2) + it];
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				if ((lane >> 5) == half) gs_row_from_lds<RF>(slab + (lane & 31) * RFP, sh);
				__builtin_amdgcn_wave_barrier();
			}
		}
	}
#endif

	uint32_t my_tiles = 0, dead = 0;
	if (vis) {
		float rgb[3];
		uint32_t clamped = 0;
		if (colors_precomp == nullptr) {
			clamped = gs_sh_to_rgb<D>(sh, p_orig, cam, rgb);
			if (D > 0 && shjac != nullptr) {
				// d(rgb) / d(view direction) for the backward, while the coefficients are in registers (gs_sh_dir_jacobian;
				// the direction exactly as gs_sh_backward forms it)
				const float3 dir_orig = {p_orig.x - cam->campos[0], p_orig.y - cam->campos[1], p_orig.z - cam->campos[2]};
				const float len = sqrtf(FMA(dir_orig.z, dir_orig.z, FMA(dir_orig.y, dir_orig.y, dir_orig.x * dir_orig.x)));
				float J[9];
				gs_sh_dir_jacobian<D>(kSH_C1, kSH_C2, kSH_C3, sh, dir_orig.x / len, dir_orig.y / len, dir_orig.z / len, J);
				// read next by the backward.  Small scenes (36 B x P well inside the 256 MiB Infinity Cache): plain stores, the
				// backward finds them cached; large ones: streamed past the caches, or they displace the records the binning
				// passes are about to read (measured at 5 M Gaussians: scan 0.101 -> 0.109 ms with plain stores)
				if (P > GSR_SHJAC_STREAM_P) {
					asm volatile("" ::: "memory");   // (keeps the optimiser from merging the two arms into plain stores)
#pragma unroll
					for (int k = 0; k < 9; k++) __builtin_nontemporal_store(J[k], shjac + 9 * (size_t)idx + k);
					asm volatile("" ::: "memory");
				} else {
#pragma unroll
					for (int k = 0; k < 9; k++) shjac[9 * (size_t)idx + k] = J[k];
				}
			}
		} else {
			rgb[0] = colors_precomp[3 * (size_t)idx];
			rgb[1] = colors_precomp[3 * (size_t)idx + 1];
			rgb[2] = colors_precomp[3 * (size_t)idx + 2];
		}
		const float op = gs_act_opacity(op_raw, act);
		// bin into the tight sub-rect of the reference's square (radii and the reported num_rendered keep the
		// reference's definition); a Gaussian that cannot reach alpha >= 1/255 anywhere is binned nowhere
		float qmax = -1.0f;
		if (tight && !gs_tight_rect(pix_x, pix_y, conic_x, conic_y, conic_z, op, gx, gy, rminx, rminy, rmaxx, rmaxy, qmax))
			rminx = rminy = rmaxx = rmaxy = 0;
		// tile-grid sharding of one view across GPUs (gsr_set_option("tile_row_lo" / "tile_row_hi")): this process only
		// bins -- and therefore only composites and differentiates -- the tile rows of its band
		rminy = max(rminy, band_lo);
		rmaxy = min(rmaxy, band_hi);
		if (rmaxy <= rminy) rminx = rminy = rmaxx = rmaxy = 0;
		// ... and not into the corner tiles of that rect the ellipse does not reach
		dead = gs_dead_corners(pix_x, pix_y, conic_x, conic_y, conic_z, qmax, rminx, rminy, rmaxx, rmaxy, W, H);
		my_tiles = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx)) - (uint32_t)__popc(dead);
		// pcut: power < pcut  ==>  op*exp(power) < 1/255 with a 1e-3 margin, so skipping the pair is
		// bit-identical to evaluating it and failing `alpha < 1/255` (forward.cu:346).  Clamped to the
		// domain of gs_exp.  op <= 0 -> +inf (always skipped); NaN -> -80 (never skipped by pcut).
		const float pcut = fmaxf(-__logf(255.0f * op) - 0.001f, -80.0f);
		GsRec rec;
		rec.q0 = make_float4(pix_x, pix_y, -0.5f * conic_x, -conic_y);
		rec.q1 = make_float4(-0.5f * conic_z, op, depth, pcut);
		rec.q2 = make_float4(rgb[0], rgb[1], rgb[2], __int_as_float(mr));
		rec.q3 = make_uint4((uint32_t)rminx | ((uint32_t)rminy << 16), (uint32_t)rmaxx | ((uint32_t)rmaxy << 16),
		                    clamped | (dead << GSR_Q3Z_DEAD_SHIFT), my_tiles);
		recs[idx] = rec;
		if (binfo != nullptr) binfo[idx] = make_uint4(rec.q3.x, rec.q3.y, rec.q3.z, (uint32_t)__float_as_int(depth));
	}
	if (idx < P) {
		radii[idx] = vis ? mr : 0;
		tiles_touched[idx] = my_tiles;
	}
	// per-block partial sums of the binned and of the reference-defined tile counts (scanned by tile_scan's second
	// workgroup): the Gaussian-major row offsets of the backward and the reference's num_rendered
	{
		__shared__ uint32_t s_sum[2][4];
		uint32_t a = my_tiles, b = ref_tiles;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			a += (uint32_t)__shfl_xor((int)a, o, 64);
			b += (uint32_t)__shfl_xor((int)b, o, 64);
		}
		if ((threadIdx.x & 63) == 0) { s_sum[0][threadIdx.x >> 6] = a; s_sum[1][threadIdx.x >> 6] = b; }
		__syncthreads();
		if (threadIdx.x == 0) {
			bsums[blockIdx.x] = s_sum[0][0] + s_sum[0][1] + s_sum[0][2] + s_sum[0][3];
			refsums[blockIdx.x] = s_sum[1][0] + s_sum[1][1] + s_sum[1][2] + s_sum[1][3];
		}
	}
	// fallback binning only (tile grids too large for the LDS histogram): count instances per tile with
	// device-scope atomics.  The default path counts in bin_hist_kernel without global atomics.
	if (tile_count != nullptr)
		for_each_tile(my_tiles > 0, rminx, rminy, rmaxx, rmaxy, dead, 0u, 0u,
		              [&](int x, int y, uint32_t, uint32_t) { atomicAdd(&tile_count[y * gx + x], 1u); });
}

void launch_preprocess_fwd(const FwdArgs& a, const GsCam* cam, const ImgLayout& il, int* radii, GsRec* recs, float* shjac, uint4* binfo,
                           uint32_t* tiles_touched, uint32_t* bsums, uint32_t* refsums, uint32_t* tile_count,
                           GsCtl* ctl, hipStream_t s)
{
	const float focal_y = a.H / (2.0f * a.tan_fovy);   // rasterizer_impl.cu:225-226
	const float focal_x = a.W / (2.0f * a.tan_fovx);
	const int sh_vec4 = (a.shs != nullptr && ((uintptr_t)a.shs % 16 == 0) && ((size_t)a.M * 12) % 16 == 0) ? 1 : 0;
	const int D = a.colors_precomp ? 0 : a.D;
	// cooperative rows: a row must be exactly the coefficients the degree uses (else the copy over-fetches) and 16-B aligned
	const int sh_coop = (sh_vec4 && !a.colors_precomp && a.shs_rest == nullptr && a.act == 0 && a.M == (D + 1) * (D + 1) && D == 3) ? 1 : 0;   // (degree 3 of a 16-coefficient tensor: the training / extraction case)
	dim3 grid((a.P + 255) / 256), block(256);
#define GSR_LAUNCH_PRE(DEG, RAW)                                                                                   \
	hipLaunchKernelGGL((preprocess_fwd_kernel<DEG, RAW>), grid, block, 0, s, a.P, a.M, a.means3D, a.scales,        \
	                   a.scale_modifier, a.rotations, a.opacities, a.shs, a.shs_rest, a.act, a.cov3D_precomp,       \
	                   a.colors_precomp, cam, a.W, a.H, a.tan_fovx, a.tan_fovy, focal_x, focal_y, il.gx, il.gy,     \
	                   a.prefiltered, sh_vec4, sh_coop, a.tight, a.band_lo, a.band_hi > 0 ? a.band_hi : il.gy, radii, recs, shjac, binfo,   \
	                   tiles_touched, bsums, refsums, tile_count, ctl)
#define GSR_LAUNCH_PRE_D(RAW)                          \
	switch (D) {                                       \
		case 0: GSR_LAUNCH_PRE(0, RAW); break;         \
		case 1: GSR_LAUNCH_PRE(1, RAW); break;         \
		case 2: GSR_LAUNCH_PRE(2, RAW); break;         \
		default: GSR_LAUNCH_PRE(3, RAW); break;        \
	}
	if (a.shs_rest != nullptr || a.act != 0) { GSR_LAUNCH_PRE_D(true) }
	else { GSR_LAUNCH_PRE_D(false) }
#undef GSR_LAUNCH_PRE_D
#undef GSR_LAUNCH_PRE
}

// ------------------------------------------------------------------------------------------------
// tile_scan: two 1024-thread workgroups.
//   block 0: exclusive scan over the T tile counts -> per-tile [start, end) ranges and the number of binned
//            instances (replaces identifyTileRanges + the InclusiveSum total); also resets the counters so that the
//            atomic-fallback bin_scatter can reuse them as per-tile cursors;
//   block 1: exclusive scan (in place) of the per-256-Gaussian block sums left by preprocess_fwd -> base row offset
//            of every block (goff_apply finishes the scan), and the sum of the reference-defined tile counts = the
//            reference's num_rendered (rasterizer_impl.cu:280-284).
// Both mirror their control words into pinned host memory (host_ctl): the host reads them after an event wait
// without a copy command sitting in the stream.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(v, o, 64);
		if (lane >= o) v += t;
	}
	return v;
}

__global__ __launch_bounds__(1024) void tile_scan_kernel(int T, uint32_t* __restrict__ tile_count,
                                                         uint2* __restrict__ ranges, int nblk,
                                                         uint32_t* __restrict__ bsums,
                                                         const uint32_t* __restrict__ refsums,
                                                         GsCtl* __restrict__ ctl, GsCtl* __restrict__ host_ctl)
{
	__shared__ uint32_t s_wave[16];
	__shared__ uint32_t s_max[16], s_long[16];
	__shared__ uint64_t s_wave64[16];
	const int tid = threadIdx.x;
	const int lane = tid & 63, wv = tid >> 6;
	if (blockIdx.x == 1) {
		const int chunk = (nblk + 1023) / 1024;
		const int b = min(nblk, tid * chunk), e = min(nblk, b + chunk);
		uint32_t sum = 0;
		uint64_t rsum = 0;
#pragma unroll 4
		for (int i = b; i < e; i++) {
			sum += bsums[i];
			rsum += refsums[i];
		}
		const uint32_t incl = wave_incl_scan(sum, lane);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) rsum += (uint64_t)__shfl_xor((long long)rsum, o, 64);
		if (lane == 63) s_wave[wv] = incl;
		if (lane == 0) s_wave64[wv] = rsum;
		__syncthreads();
		uint32_t base = 0, total = 0;
		uint64_t rtotal = 0;
		for (int w = 0; w < 16; w++) {
			if (w < wv) base += s_wave[w];
			total += s_wave[w];
			rtotal += s_wave64[w];
		}
		uint32_t run = base + incl - sum;
#pragma unroll 4
		for (int i = b; i < e; i++) {
			const uint32_t c = bsums[i];
			bsums[i] = run;
			run += c;
		}
		if (tid == 0) {
			bsums[nblk] = total;
			const uint32_t ovf = rtotal > 0x7fffffffull ? 1u : 0u;   // the reference returns an int
			ctl->ref_rendered = (uint32_t)rtotal;
			ctl->err_overflow = ovf;
			if (host_ctl) {
				host_ctl->ref_rendered = (uint32_t)rtotal;
				host_ctl->err_overflow = ovf;
				host_ctl->err_prefiltered = ctl->err_prefiltered;   // set by preprocess_fwd, an earlier kernel
				__threadfence_system();
			}
		}
		return;
	}
	const int chunk = (T + 1023) / 1024;
	const int b = min(T, tid * chunk), e = min(T, b + chunk);
	uint32_t sum = 0, mx = 0, nlong = 0;
#pragma unroll 8
	for (int i = b; i < e; i++) {   // (unrolled: eight independent loads in flight; 32 serial round trips took 59 us at 4K)
		const uint32_t c = tile_count[i];
		sum += c;
		mx = max(mx, c);
		nlong += c > GSR_SORT_LDS_MAX ? 1u : 0u;
	}
	// inclusive scan across the wave, then across the 16 waves
	const uint32_t incl = wave_incl_scan(sum, lane);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64)); nlong += (uint32_t)__shfl_xor((int)nlong, o, 64); }
	uint64_t wsum = sum;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) wsum += (uint64_t)__shfl_xor((long long)wsum, o, 64);
	if (lane == 63) s_wave[wv] = incl;
	if (lane == 0) { s_max[wv] = mx; s_long[wv] = nlong; s_wave64[wv] = wsum; }
	__syncthreads();
	uint32_t base = 0, total = 0, gmax = 0, glong = 0;
	uint64_t total64 = 0;
	for (int w = 0; w < 16; w++) {
		if (w < wv) base += s_wave[w];
		total += s_wave[w];
		total64 += s_wave64[w];
		gmax = max(gmax, s_max[w]);
		glong += s_long[w];
	}
	// the u32 scan wraps past 2^32 instances (the host rejects the frame: the reference count is at least as large);
	// report a count no capacity can hold so that every kernel enqueued ahead of the host's check leaves at once
	if (total64 > 0x7fffffffull) total = 0xffffffffu;
	uint32_t run = base + incl - sum;
#pragma unroll 8
	for (int i = b; i < e; i++) {
		const uint32_t c = tile_count[i];
		ranges[i] = make_uint2(run, run + c);
		run += c;
		tile_count[i] = 0;
	}
	if (tid == 0) {
		ctl->num_binned = total;           // <= the reference count, whose overflow block 1 checks
		ctl->max_tile_count = gmax;
		ctl->n_long = glong;
		if (host_ctl) {
			host_ctl->num_binned = total;
			host_ctl->max_tile_count = gmax;
			__threadfence_system();
		}
	}
}

void launch_tile_scan(int T, uint32_t* tile_count, uint2* ranges, int nblk, uint32_t* bsums, const uint32_t* refsums,
                      GsCtl* ctl, GsCtl* host_ctl, hipStream_t s)
{
	hipLaunchKernelGGL(tile_scan_kernel, dim3(2), dim3(1024), 0, s, T, tile_count, ranges, nblk, bsums, refsums, ctl,
	                   host_ctl);
}

// goff[g] = bsums[block of g] + exclusive scan of tiles_touched inside the 256-Gaussian block (4 Gaussians per thread:
// a 256-thread workgroup finishes four blocks).  The chunked scatter (bin_chunk_kernel<true>) does this on its way;
// goff_apply_kernel is for the paths that do not run it.
__device__ __forceinline__ void gs_goff_block(int P, int blk, int lane, const uint32_t* __restrict__ tiles_touched,
                                              const uint32_t* __restrict__ bsums, uint32_t* __restrict__ goff)
{
	// one wave per 256-Gaussian block: lane l holds Gaussians 4 l .. 4 l + 3
	const int base = blk * GSR_PRE_BLOCK + 4 * lane;
	if (blk * GSR_PRE_BLOCK >= P) return;
	uint32_t v[4];
	if (base + 3 < P) {
		const uint4 q = *reinterpret_cast<const uint4*>(tiles_touched + base);
		v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++) v[i] = base + i < P ? tiles_touched[base + i] : 0u;
	}
	const uint32_t sum = v[0] + v[1] + v[2] + v[3];
	uint32_t run = bsums[blk] + wave_incl_scan(sum, lane) - sum;
	uint32_t o[4];
#pragma unroll
	for (int i = 0; i < 4; i++) { o[i] = run; run += v[i]; }
	if (base + 3 < P) {
		*reinterpret_cast<uint4*>(goff + base) = make_uint4(o[0], o[1], o[2], o[3]);
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++) if (base + i < P) goff[base + i] = o[i];
	}
	if (base <= P - 1 && P - 1 < base + 4) goff[P] = o[P - 1 - base] + v[P - 1 - base];
}

__global__ __launch_bounds__(256) void goff_apply_kernel(int P, const uint32_t* __restrict__ tiles_touched,
                                                         const uint32_t* __restrict__ bsums, uint32_t* __restrict__ goff)
{
	// wave w of the workgroup owns the 256-Gaussian block 4 * blockIdx.x + w
	gs_goff_block(P, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, tiles_touched, bsums, goff);
}

void launch_goff_apply(int P, const uint32_t* tiles_touched, const uint32_t* bsums, uint32_t* goff, hipStream_t s)
{
	const int nblk = (P + GSR_PRE_BLOCK - 1) / GSR_PRE_BLOCK;
	hipLaunchKernelGGL(goff_apply_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, P, tiles_touched, bsums, goff);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bin_scatter_kernel(int P, int gx, const int* __restrict__ radii,
                                                          const uint32_t* __restrict__ tiles_touched,
                                                          const GsRec* __restrict__ recs,
                                                          const uint2* __restrict__ ranges,
                                                          uint32_t* __restrict__ cursor,
                                                          uint64_t* __restrict__ keys, const GsCtl* __restrict__ ctl,
                                                          uint32_t cap)
{
	if (ctl->num_binned > cap) return;   // launched ahead of the host's read-back: the buffer is too small, the host re-launches
	const int idx = blockIdx.x * 256 + threadIdx.x;
	bool vis = false;
	int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
	uint32_t dbits = 0, dead = 0;
	if (idx < P && radii[idx] > 0 && tiles_touched[idx] > 0) {
		vis = true;
		const uint4 q3 = recs[idx].q3;
		rminx = q3.x & 0xffff; rminy = q3.x >> 16;
		rmaxx = q3.y & 0xffff; rmaxy = q3.y >> 16;
		dead = (q3.z >> GSR_Q3Z_DEAD_SHIFT) & 15u;
		dbits = (uint32_t)__float_as_int(recs[idx].q1.z);
	}
	for_each_tile(vis, rminx, rminy, rmaxx, rmaxy, dead, dbits, (uint32_t)idx, [&](int x, int y, uint32_t d, uint32_t id) {
		// key = | depth bits | id |: within a tile the order is (depth, id) exactly as the stable
		// radix sort of rasterizer_impl.cu:98-109,306-311 produces (depth > 0.2 so bits order as uint)
		const int tile = y * gx + x;
		const uint32_t pos = ranges[tile].x + atomicAdd(&cursor[tile], 1u);
		keys[pos] = ((uint64_t)d << 32) | id;
	});
}

void launch_bin_scatter(int P, int gx, const int* radii, const uint32_t* tiles_touched, const GsRec* recs,
                        const uint2* ranges, uint32_t* cursor, uint64_t* keys, const GsCtl* ctl, uint32_t cap,
                        hipStream_t s)
{
	hipLaunchKernelGGL(bin_scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, radii, tiles_touched, recs,
	                   ranges, cursor, keys, ctl, cap);
}

// ------------------------------------------------------------------------------------------------
// Binning without global atomics.  Device-scope float/int atomics top out near 20 G/s on MI355X (8 XCDs
// with private L2s), which made the one-atomic-per-instance counting + scatter cost 0.45 ms at C3.
// Instead the Gaussians are cut into G chunks, one workgroup per chunk:
//   bin_hist      per-chunk histogram over tiles in LDS (ds_add), written as row g of Hm[G][T];
//   bin_colscan   per tile: exclusive prefix over the chunks (Hm becomes per-chunk offsets) + tile totals;
//   tile_scan     (existing) prefix over tiles -> ranges, R;
//   bin_scatter2  per chunk: LDS cursors = ranges[t].x + Hm[g][t]; every instance takes its slot with a
//                 returning LDS atomic and stores its (depth, id) key.
// Order inside a tile is arbitrary here; tile_sort makes it (depth, id) as before.
// One 1024-thread workgroup per chunk (16 waves per CU instead of 4 hide the record-fetch latency: the kernels are
// latency-, not bandwidth-bound).
#define GSR_BIN_THREADS 1024
#ifndef GSR_BIN_BATCH
#define GSR_BIN_BATCH 4
#endif
template <bool SCATTER>
__global__ __launch_bounds__(GSR_BIN_THREADS) void bin_chunk_kernel(int P, int chunk, int gx, int T,
                                                        const uint32_t* __restrict__ tiles_touched,
                                                        const GsRec* __restrict__ recs, const uint4* __restrict__ binfo,
                                                        uint32_t* __restrict__ Hm, const uint2* __restrict__ ranges,
                                                        uint64_t* __restrict__ keys, const uint32_t* __restrict__ bsums,
                                                        uint32_t* __restrict__ goff, const GsCtl* __restrict__ ctl,
                                                        uint32_t cap)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	uint32_t* cnt = reinterpret_cast<uint32_t*>(smem_raw);
	const int tid = threadIdx.x;
	const int g = blockIdx.x;   // (an XCD-banded chunk order was measured: no effect on the scatter)
	if (SCATTER && goff != nullptr) {
		// the backward's Gaussian-major row offsets (goff_apply_kernel's job) for this chunk, ahead of the capacity guard:
		// a chunk is a whole number of 256-Gaussian blocks, one wave per block and pass
		for (int b0 = g * chunk; b0 < (g + 1) * chunk && b0 < P; b0 += 4 * GSR_BIN_THREADS) {
			const int blk = b0 / GSR_PRE_BLOCK + (tid >> 6);
			if (blk * GSR_PRE_BLOCK < (g + 1) * chunk) gs_goff_block(P, blk, tid & 63, tiles_touched, bsums, goff);
		}
	}
	if (SCATTER && ctl->num_binned > cap) return;   // see bin_scatter_kernel
	uint32_t* row = Hm + (size_t)g * T;
	for (int i = tid; i < T; i += GSR_BIN_THREADS) cnt[i] = SCATTER ? ranges[i].x + row[i] : 0u;
	__syncthreads();
	// The scatter takes NB = 4 Gaussians per thread and round: their visibility words are fetched together, then their records
	// together (culled ones read the chunk's first record: one cached line), then the tiles are walked.  One Gaussian per round
	// is a chain of dependent round trips in which every load's wait also waits for the scattered stores issued before it
	// (vmcnt counts stores): C4 0.256 -> 0.240 ms, C5 0.174 -> 0.154, C3 unchanged (8 gave C5 0.138 but C4 0.246; the counting
	// pass, which stores nothing, lost 2.5 us at C3 with batches and keeps one).  What bounds the scatter beyond that is the
	// issue rate of uncoalesced 8-B stores -- one lane-address per ~4 cycles and CU: 16 M keys = 0.1-0.2 ms at C4 -- and
	// twice the workgroups (512 chunks) only shortens the per-(chunk, tile) runs: measured slower (C3 0.0525 -> 0.055, C4 0.240 -> 0.253).
	constexpr int NB = SCATTER ? GSR_BIN_BATCH : 1;
	const int base = g * chunk;
	// the record culled lanes read instead of their own: the chunk's first, or -- for the trailing chunks that start behind the
	// last Gaussian (chunk is rounded up to 1024, so G * chunk can reach ~2 P) -- the last record of the buffer
	const int spare = base < P ? base : P - 1;
	for (int off = 0; off < chunk; off += NB * GSR_BIN_THREADS) {
		int idxs[NB];
		bool vis[NB];
#pragma unroll
		for (int u = 0; u < NB; u++) {
			idxs[u] = base + off + u * GSR_BIN_THREADS + tid;
			vis[u] = off + u * GSR_BIN_THREADS < chunk && idxs[u] < P && tiles_touched[idxs[u]] > 0;
		}
		uint4 q3s[NB];
		uint32_t dbs[NB];
#pragma unroll
		for (int u = 0; u < NB; u++) {
			const int at = vis[u] ? idxs[u] : spare;
			if (binfo != nullptr) {   // the 16-B binning record: rect, dead corners and depth in one half sector
				const uint4 b = binfo[at];
				q3s[u] = b;
				dbs[u] = b.w;
			} else {
				const GsRec* r = recs + at;
				q3s[u] = r->q3;
				dbs[u] = SCATTER ? (uint32_t)__float_as_int(r->q1.z) : 0u;
			}
		}
#pragma unroll
		for (int u = 0; u < NB; u++) {
			const uint4 q3 = q3s[u];
			const int rminx = q3.x & 0xffff, rminy = q3.x >> 16, rmaxx = q3.y & 0xffff, rmaxy = q3.y >> 16;
			const uint32_t dead = (q3.z >> GSR_Q3Z_DEAD_SHIFT) & 15u;
			for_each_tile(vis[u], rminx, rminy, rmaxx, rmaxy, dead, dbs[u], (uint32_t)idxs[u], [&](int x, int y, uint32_t d, uint32_t id) {
				const uint32_t slot = atomicAdd(&cnt[y * gx + x], 1u);   // ds_add(_rtn)_u32
				if (SCATTER) keys[slot] = ((uint64_t)d << 32) | id;
			});
		}
	}
	if (!SCATTER) {
		__syncthreads();
		for (int i = tid; i < T; i += GSR_BIN_THREADS) row[i] = cnt[i];
	}
}

// thread (t, q): tile t, sixteenth q of the chunks; exclusive prefix over chunks in place, totals out
#define GSR_COLSCAN_Q 16
__global__ __launch_bounds__(64 * GSR_COLSCAN_Q) void bin_colscan_kernel(int G, int T, uint32_t* __restrict__ Hm,
                                                                         uint32_t* __restrict__ tile_count)
{
	__shared__ uint32_t s_q[GSR_COLSCAN_Q][64];
	const int tl = threadIdx.x & 63, q = threadIdx.x >> 6;
	const int t = blockIdx.x * 64 + tl;
	const int per = (G + GSR_COLSCAN_Q - 1) / GSR_COLSCAN_Q;
	const int g0 = min(G, q * per), g1 = min(G, g0 + per);
	uint32_t sum = 0;
	if (t < T) {
#pragma unroll 8
		for (int g = g0; g < g1; g++) sum += Hm[(size_t)g * T + t];
	}
	s_q[q][tl] = sum;
	__syncthreads();
	uint32_t run = 0;
	for (int k = 0; k < q; k++) run += s_q[k][tl];
	if (t < T) {
#pragma unroll 8
		for (int g = g0; g < g1; g++) {
			const uint32_t v = Hm[(size_t)g * T + t];
			Hm[(size_t)g * T + t] = run;
			run += v;
		}
		if (q == GSR_COLSCAN_Q - 1) tile_count[t] = run;
	}
}

#ifndef GSR_BIN_CHUNKS2
#define GSR_BIN_CHUNKS2 256
#endif
// chunks = workgroups: one per CU, or (GSR_BIN_CHUNKS2 = 512) two per CU where two tile histograms fit the LDS
int bin_chunks(int P, int T)
{
	const int gmax = (size_t)T * sizeof(uint32_t) <= 72 * 1024 ? GSR_BIN_CHUNKS2 : 256;
	return P >= gmax * GSR_BIN_THREADS ? gmax : (P >= 256 * GSR_BIN_THREADS ? 256 : (P + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS);
}
size_t bin_hist_bytes(int P, int T) { return sizeof(uint32_t) * (size_t)bin_chunks(P, T) * (size_t)T; }
bool bin_lds_path_ok(int T) { return (size_t)T * sizeof(uint32_t) <= 150 * 1024; }

static void set_dyn_lds(const void* fn, size_t bytes)
{
	if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void launch_bin_hist(int P, int gx, int T, const uint32_t* tiles_touched, const GsRec* recs, const uint4* binfo, uint32_t* Hm,
                     uint32_t* tile_count, hipStream_t s)
{
	const int G = bin_chunks(P, T);
	const int chunk = ((P + G - 1) / G + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS * GSR_BIN_THREADS;
	const size_t lds = (size_t)T * sizeof(uint32_t);
	set_dyn_lds((const void*)bin_chunk_kernel<false>, lds);
	hipLaunchKernelGGL(bin_chunk_kernel<false>, dim3(G), dim3(GSR_BIN_THREADS), lds, s, P, chunk, gx, T, tiles_touched, recs, binfo, Hm,
	                   (const uint2*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const GsCtl*)nullptr, 0u);
	hipLaunchKernelGGL(bin_colscan_kernel, dim3((T + 63) / 64), dim3(64 * GSR_COLSCAN_Q), 0, s, G, T, Hm, tile_count);
}

void launch_bin_scatter2(int P, int gx, int T, const uint32_t* tiles_touched, const GsRec* recs, const uint4* binfo, uint32_t* Hm,
                         const uint2* ranges, uint64_t* keys, const uint32_t* bsums, uint32_t* goff, const GsCtl* ctl,
                         uint32_t cap, hipStream_t s)
{
	const int G = bin_chunks(P, T);
	const int chunk = ((P + G - 1) / G + GSR_BIN_THREADS - 1) / GSR_BIN_THREADS * GSR_BIN_THREADS;
	const size_t lds = (size_t)T * sizeof(uint32_t);
	set_dyn_lds((const void*)bin_chunk_kernel<true>, lds);
	hipLaunchKernelGGL(bin_chunk_kernel<true>, dim3(G), dim3(GSR_BIN_THREADS), lds, s, P, chunk, gx, T, tiles_touched, recs, binfo, Hm,
	                   ranges, keys, bsums, goff, ctl, cap);
}

// ------------------------------------------------------------------------------------------------
// Per-tile sort of short lists (n <= GSR_SORT_LDS_MAX = 1024 keys): ONE wave per tile, the keys live in registers
// (KPL = 1..16 64-bit keys per lane, blocked: key index = lane*KPL + e), and the whole bitonic network runs without
// LDS storage or barriers: compare-exchanges at strides < KPL are register-to-register (fully unrolled), strides
// >= KPL exchange with lane ^ m through ds_bpermute.  For ~500 keys that is ~1.5 k wave-instructions per tile
// against ~3.6 k plus 45 workgroup barriers for the four-wave LDS network it replaces (78 us at C3).
// Unused slots hold +inf keys; (depth, id) keys are unique, so there are no ties.
__device__ __forceinline__ void gs_cex(uint64_t& a, uint64_t& b, bool asc)
{
	const bool sw = (a > b) == asc;
	const uint64_t lo = sw ? b : a, hi = sw ? a : b;
	a = lo;
	b = hi;
}

// the network on 64 x KPL keys in registers; result in blocked order (key index = lane * KPL + e)
template <int KPL>
__device__ __forceinline__ void gs_wave_sort_regs(uint64_t (&k)[KPL], int lane)
{
	// sizes 2 .. KPL/2: entirely inside a lane, direction known at compile time (bit `size` of e)
#pragma unroll
	for (int size = 2; size < KPL; size <<= 1) {
#pragma unroll
		for (int j = size >> 1; j > 0; j >>= 1) {
#pragma unroll
			for (int e = 0; e < KPL; e++)
				if ((e & j) == 0) gs_cex(k[e], k[e | j], (e & size) == 0);
		}
	}
	// sizes KPL << ls, ls = 0 .. 6: direction = bit ls of the lane (0 for the final merge).  Fully unrolled so that
	// the partner exchange of the 18 (of 21) stages with lane strides 1, 2, 4 and 8 is a DPP move (quad permute / row
	// rotate) instead of a ds_bpermute (a trip through the LDS crossbar); strides 16 and 32 keep the bpermute.
#pragma unroll
	for (int ls = 0; ls <= 6; ls++) {
		const bool asc = ((lane >> ls) & 1) == 0;
#pragma unroll
		for (int m = (1 << ls) >> 1; m > 0; m >>= 1) {   // cross-lane strides m*KPL
			const bool keep_min = ((lane & m) == 0) == asc;
			const int src = (lane ^ m) << 2;
#pragma unroll
			for (int e = 0; e < KPL; e++) {
				uint32_t olo, ohi;
				if (m == 1) {
					olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k[e], 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
					ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k[e] >> 32), 0xB1, 0xF, 0xF, true);
				} else if (m == 2) {
					olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k[e], 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
					ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k[e] >> 32), 0x4E, 0xF, 0xF, true);
				} else if (m == 4) {
					// lane ^ 4 inside a 16-lane row: rotate by 4 for one half of the lanes, by 12 for the other (the second
					// move only writes the banks -- groups of four lanes -- with bit 2 of the lane set)
					const int lo_ = (int)(uint32_t)k[e], hi_ = (int)(uint32_t)(k[e] >> 32);
					olo = (uint32_t)__builtin_amdgcn_update_dpp(__builtin_amdgcn_update_dpp(0, lo_, 0x12C, 0xF, 0xF, true), lo_, 0x124, 0xF, 0xA, false);
					ohi = (uint32_t)__builtin_amdgcn_update_dpp(__builtin_amdgcn_update_dpp(0, hi_, 0x12C, 0xF, 0xF, true), hi_, 0x124, 0xF, 0xA, false);
				} else if (m == 8) {
					olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k[e], 0x128, 0xF, 0xF, true);          // row_ror:8
					ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k[e] >> 32), 0x128, 0xF, 0xF, true);
				} else {
					olo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)k[e]);
					ohi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(k[e] >> 32));
				}
				const uint64_t o = ((uint64_t)ohi << 32) | olo;
				k[e] = ((o < k[e]) == keep_min) ? o : k[e];
			}
		}
#pragma unroll
		for (int j = KPL >> 1; j > 0; j >>= 1) {
#pragma unroll
			for (int e = 0; e < KPL; e++)
				if ((e & j) == 0) gs_cex(k[e], k[e | j], asc);
		}
	}
}
template <int KPL>
__device__ __forceinline__ void gs_wave_sort_tile(const uint64_t* __restrict__ keys, uint32_t* __restrict__ out, uint32_t n,
                                                  int lane)
{
	uint64_t k[KPL];
	// the input order is irrelevant to a sort: fetch striped (coalesced, 512 B per instruction); only the result
	// is in blocked order
#pragma unroll
	for (int e = 0; e < KPL; e++) {
		const uint32_t i = (uint32_t)e * 64 + lane;
		k[e] = i < n ? keys[i] : ~0ull;
	}
	gs_wave_sort_regs<KPL>(k, lane);
#pragma unroll
	for (int e = 0; e < KPL; e++) {
		const uint32_t i = (uint32_t)lane * KPL + e;
		if (i < n) out[i] = (uint32_t)k[e];
	}
}

// With `q` (a frame of the queue pipeline below, long_level 2), the wave of a LONG list -- idle here otherwise -- prepares its
// first cut: it appends the list's slices to the slice queue and sorts the list's 1024 evenly spaced sample keys (the same
// register network, KPL = 16: the work of one 1024-key tile) into the head of the list's `keys2` range, where slice_hist
// picks the splitters up before slice_scatter overwrites them.
struct GsSortQ;
__device__ __forceinline__ void gs_enter_long_list(uint2 range, GsSortQ* __restrict__ q, uint4* __restrict__ slice_items, uint32_t slice_cap,
                                                   const uint64_t* __restrict__ keys, uint64_t* __restrict__ keys2, int lane);
// (90 VGPRs, 5 waves per SIMD; held to 6 / 7 / 8 waves it measured the same within noise -- C3 0.0258 / 0.0256 / 0.028 / 0.0252 ms: round 5)
__global__ __launch_bounds__(256) void tile_sort_kernel(int T, const uint2* __restrict__ ranges, const uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ point_list, const GsCtl* __restrict__ ctl,
                                                        uint32_t cap, GsSortQ* __restrict__ q, uint4* __restrict__ slice_items,
                                                        uint32_t slice_cap, uint64_t* __restrict__ keys2)
{
	if (ctl->num_binned > cap) return;   // see bin_scatter_kernel
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per tile, no cross-wave traffic
	if (tile >= T) return;
	const int lane = threadIdx.x & 63;
	const uint2 range = ranges[tile];
	const uint32_t n = range.y - range.x;
	if (n == 0) return;
	if (n > GSR_SORT_LDS_MAX) {   // handled by tile_radix_sort_kernel, or (q) by the queue pipeline -- see launch_tile_sort
		if (q != nullptr && (n > GSR_PART_REGS || ctl->n_long < GSR_SORT_MANY)) gs_enter_long_list(range, q, slice_items, slice_cap, keys, keys2, lane);
		return;
	}
	const uint64_t* src = keys + range.x;
	uint32_t* dst = point_list + range.x;
	if (n <= 64) gs_wave_sort_tile<1>(src, dst, n, lane);
	else if (n <= 128) gs_wave_sort_tile<2>(src, dst, n, lane);
	else if (n <= 256) gs_wave_sort_tile<4>(src, dst, n, lane);
	else if (n <= 512) gs_wave_sort_tile<8>(src, dst, n, lane);
	else gs_wave_sort_tile<16>(src, dst, n, lane);
}

// Long lists (> GSR_SORT_LDS_MAX keys, e.g. 5 M Gaussians in a 1297x840 frame: 4.7 k per tile on average): a
// bitonic network is O(n log^2 n) and took 2.5 ms there.  tile_radix_sort is a per-tile stable LSD radix sort on
// the 32 depth bits (4 passes of 8 bits) ping-ponging between `keys` and `keys2` in global memory (a tile's
// segment stays in L2), one workgroup per tile:
//   each wave owns a contiguous quarter of the list; per pass: per-wave digit histograms (LDS atomics),
//   exclusive scan over (digit, wave), then every wave walks its quarter in order 64 keys at a time, ranking
//   equal digits inside the group with 8 ballots (match-any) -- stable by construction.
// Equal depths come out in arbitrary id order (the scatter order is arbitrary), so a final pass orders every run
// of equal depth by id: the key (depth, id) ordering of the reference's stable radix sort.
// One stable counting pass on the 8 bits at `shift` from src to dst (see tile_radix_sort_kernel).
__device__ __forceinline__ void tile_radix_pass(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int shift,
                                                uint32_t wb, uint32_t we, uint32_t (*whist)[256], uint32_t* s_tot)
{
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	for (int i = tid; i < 1024; i += 256) (&whist[0][0])[i] = 0;
	__syncthreads();
	for (uint32_t i = wb + lane; i < we; i += 64) atomicAdd(&whist[wv][(uint32_t)(src[i] >> shift) & 255u], 1u);
	__syncthreads();
	// thread d: exclusive offsets of digit d for the four waves; then add the prefix over digits
	{
		const uint32_t c0 = whist[0][tid], c1 = whist[1][tid], c2 = whist[2][tid], c3 = whist[3][tid];
		const uint32_t tot = c0 + c1 + c2 + c3;
		uint32_t incl = tot;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
			if (lane >= o) incl += t;
		}
		if (lane == 63) s_tot[wv] = incl;
		__syncthreads();
		uint32_t base = incl - tot;
		for (int w = 0; w < wv; w++) base += s_tot[w];
		whist[0][tid] = base;
		whist[1][tid] = base + c0;
		whist[2][tid] = base + c0 + c1;
		whist[3][tid] = base + c0 + c1 + c2;
	}
	__syncthreads();
	for (uint32_t g0 = wb; g0 < we; g0 += 64) {
		const uint32_t i = g0 + lane;
		const bool act = i < we;
		const uint64_t k = act ? src[i] : 0ull;
		const uint32_t d = act ? (uint32_t)(k >> shift) & 255u : 256u;
		// lanes holding the same digit (match-any over 8 bits)
		unsigned long long same = __ballot(act);
#pragma unroll
		for (int b = 0; b < 8; b++) {
			const unsigned long long bm = __ballot((d >> b) & 1u);
			same &= ((d >> b) & 1u) ? bm : ~bm;
		}
		if (act) {
			const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
			const uint32_t base = whist[wv][d];
			dst[base + rank] = k;
			if (rank == 0) whist[wv][d] = base + (uint32_t)__popcll(same);   // group leader advances the cursor
		}
	}
	__syncthreads();
}

// ---- long lists, round 4: partition by sample splitters, then sort the buckets ALL OVER THE CHIP ------------------
// Until round 3 one workgroup did everything for its tile: cut the list into <= 1024 equal-width depth buckets, sort the
// buckets with its four waves -- and, when one bucket overflowed (depths piled up: the surface a tile looks at), fall back
// to four counting passes over the whole list.  On a clustered scene (bench.py --workload C2-clustered: per-tile lists
// p50 10 / p99 11 k / max 58 k) that kernel was 55 % of the step: 0.98 ms for the workgroup that owned the 58 k list.
// Now the stage is a small pipeline over a WORK QUEUE in the binning buffer (GsSortQ + item arrays):
//   tile_sort (long-list branch) / slice_hist / slice_scatter
//                      the first cut of every long list, by ceil(n / 8192) workgroups (see there): 1024 evenly spaced keys of
//                      the list are sorted and every (1024 / B)-th becomes a splitter (sample sort; B = 2^k <= 512 buckets of
//                      ~128 keys: equal COUNT, whatever the depth distribution), histogram + scatter into `keys2`; every
//                      non-empty bucket becomes a queue item: <= 1024 keys -> sort item, more (lists beyond 64 k keys,
//                      sampling noise) -> a segment for the next round.  (64-bit keys (depth << 32 | id) are distinct: a pile
//                      of equal depths splits on the id bits.)
//   segment_partition  the same cut (gs_partition_segment: one workgroup sorts the segment's own samples) applied to the oversized
//                      buckets, into the other key buffer; what is STILL oversized goes to the fallback queue.
//   bucket_sort        every wave of the chip sorts buckets from the queue in registers (gs_wave_sort_tile) -> point_list; in
//                      front of that its workgroups take the fallback segments (normally none): the counting sort that used to
//                      take the whole tile (tile_radix_pass).
// The order is (depth, id) as before; point_list is bit-identical to the oracle's.
struct GsSortQ {
	uint32_t n_sort, n_seg, n_fall, err, n_slice, pad[3];
};
#define GSR_Q_IN_KEYS 0x80000000u    // item flag (in .y): the bucket's keys are in `keys` (second round), else in `keys2`
#define GSR_PART_THREADS 1024
#define GSR_PART_AVG 128u            // target keys per bucket

// One workgroup (NT threads) cuts the n > NT keys at src into buckets written to dst (same offsets); items are positions
// relative to the binning arrays (`abs0` = index of src[0] in them).  item_flag marks where the buckets live.
//
// The cut is by SPLITTERS TAKEN FROM THE LIST (sample sort), not by equal key width: NT evenly spaced keys are sorted by the
// workgroup (in-wave stages through ds_bpermute, the ten cross-wave stages through LDS), every (NT / B)-th of them is a
// splitter, a key's bucket is found by binary search (log2 B LDS reads).  Until this change the buckets were B equal slices
// of [min key, max key]: on the surface a tile looks at the depths pile up, 44 % of the clustered scene's 58 k-key list fell
// into 4 of 512 buckets -- 12 oversized buckets holding 55 k keys went through a second cut, and in the first one ~10 lanes of
// every wave met in one LDS counter (same-address atomics serialise).  Buckets of equal COUNT (~128 +- sampling noise) need
// no second cut, spread the atomics, and make the min / max sweep unnecessary.
// NT threads, one sample each: ascending bitonic sort, result in s_samp[0 .. NT) (s_samp holds 2 NT words: the cross-wave
// stages alternate between its halves, one barrier per stage; the 45 in-wave stages exchange through ds_bpermute)
template <int NT>
__device__ __forceinline__ void gs_sort_samples(unsigned long long v, unsigned long long* s_samp)
{
	const int tid = threadIdx.x;
	int pp = 0;
	for (int k = 2; k <= NT; k <<= 1) {
		const bool asc = (tid & k) == 0;
		for (int j = k >> 1; j > 0; j >>= 1) {
			unsigned long long o;
			if (j >= 64) {
				unsigned long long* b = s_samp + (pp ? NT : 0);
				pp ^= 1;
				b[tid] = v;
				__syncthreads();
				o = b[tid ^ j];
			} else {
				o = (unsigned long long)__shfl_xor((long long)v, j, 64);
			}
			const bool take_min = ((tid & j) == 0) == asc;   // the lower index of an ascending pair keeps the smaller key
			v = take_min ? min(v, o) : max(v, o);
		}
	}
	__syncthreads();   // the last LDS stage's reads
	s_samp[tid] = v;
	__syncthreads();
}
// The B - 1 splitters (every spb-th of the sorted samples) as a search tree in breadth-first order: node 1 is the middle
// splitter, node v has the children 2 v and 2 v + 1; level L sits in tree[2^L .. 2^(L+1)).  A binary search over the SORTED
// array reads addresses that are multiples of large powers of two on its first levels -- all in one LDS bank, a 2^L-way
// conflict at level L (measured: the histogram kernel of the first cut 33 us, LDS-bound) -- here the 2^L nodes of a level are
// neighbours.  Thread v < B fills node v from `samples` (LDS or global).
__device__ __forceinline__ void gs_build_splitter_tree(unsigned long long* tree, const unsigned long long* samples, uint32_t B, uint32_t spb)
{
	const uint32_t v = threadIdx.x;
	if (v >= 1u && v < B) {
		const int L = 31 - __clz((int)v);
		const uint32_t i = v - (1u << L);
		tree[v] = samples[(((2u * i + 1u) * B) >> (L + 1)) * spb];
	}
}
// bucket of key k = number of splitters <= k  (B a power of two, log2 B steps)
__device__ __forceinline__ uint32_t gs_tree_bucket(const unsigned long long* tree, uint32_t B, unsigned long long k)
{
	uint32_t v = 1;
	while (v < B) v = 2u * v + (tree[v] <= k ? 1u : 0u);
	return v - B;
}
// samples per list: four per bucket, 64 .. 1024 (a 3.7 k-key list has 32 buckets: 128 samples, sorted by its wave at 1/30
// of the cost of 1024 -- with 1024 for every list the sample sorts of a C4-like frame took 81 us)
__device__ __forceinline__ uint32_t gs_sample_count(uint32_t B) { return min(1024u, max(64u, 4u * B)); }
__device__ __forceinline__ uint32_t gs_bucket_count(uint32_t n, uint32_t nt)   // B * GSR_PART_AVG >= n, 2 <= B <= nt / 2
{
	uint32_t B = 2;
	while (B < nt / 2u && B * GSR_PART_AVG < n) B <<= 1;
	return B;
}

template <int NT>
__device__ __forceinline__ void gs_partition_segment(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint32_t abs0,
                                                     uint32_t n, uint32_t item_flag, GsSortQ* __restrict__ q,
                                                     uint2* __restrict__ sort_items, uint32_t sort_cap,
                                                     uint2* __restrict__ over_items, uint32_t over_cap, uint32_t* over_count,
                                                     uint32_t* s_off, uint32_t* s_cur, uint32_t* s_misc, unsigned long long* s_samp)
{
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	constexpr int NW = NT / 64, RK = (int)(GSR_PART_REGS / NT);
	if (tid == 0) { s_misc[0] = 0u; s_misc[1] = 0u; }
	const bool in_regs = n <= (uint32_t)NT * RK;
	uint64_t kreg[RK];
	if (in_regs) {
#pragma unroll
		for (int r = 0; r < RK; r++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)NT * r;
			kreg[r] = i < n ? src[i] : ~0ull;
		}
	}
	gs_sort_samples<NT>(src[(uint32_t)(((unsigned long long)tid * n) / NT)], s_samp);   // n > NT: distinct positions, distinct keys
	const uint32_t B = gs_bucket_count(n, (uint32_t)NT);
	const uint32_t spb = (uint32_t)NT / B;                 // samples per bucket (>= 2)
	for (uint32_t b = tid; b <= B; b += NT) s_off[b] = 0u;
	unsigned long long* tree = s_samp + NT;                // the sort's second buffer: B <= NT / 2 nodes
	gs_build_splitter_tree(tree, s_samp, B, spb);
	__syncthreads();
	auto bucket_of = [&](const unsigned long long k) -> uint32_t { return gs_tree_bucket(tree, B, k); };
	uint32_t breg[RK];
	if (in_regs) {
#pragma unroll
		for (int r = 0; r < RK; r++) {
			breg[r] = 0u;
			if ((uint32_t)tid + (uint32_t)NT * r < n) { breg[r] = bucket_of(kreg[r]); atomicAdd(&s_off[breg[r]], 1u); }
		}
	} else {
		// long lists: eight independent loads in flight per thread and sweep (one load at a time made the 58 k-key list of
		// the clustered scene take 58 us here: sweeps of 57 dependent round trips)
		for (uint32_t i0 = tid; i0 < n; i0 += 8u * NT) {
			unsigned long long k8[8];
#pragma unroll
			for (int u = 0; u < 8; u++) { const uint32_t i = i0 + (uint32_t)u * NT; k8[u] = i < n ? src[i] : 0ull; }
#pragma unroll
			for (int u = 0; u < 8; u++) if (i0 + (uint32_t)u * NT < n) atomicAdd(&s_off[bucket_of(k8[u])], 1u);
		}
	}
	__syncthreads();
	// exclusive scan of the B counts (B <= NT / 2: at most one per thread); how many sort / oversized items this cut produces
	{
		const uint32_t c = (uint32_t)tid < B ? s_off[tid] : 0u;
		const bool is_s = c > 0u && c <= GSR_SORT_LDS_MAX, is_o = c > GSR_SORT_LDS_MAX;
		uint32_t incl = c;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
			if (lane >= o) incl += t;
		}
		__shared__ uint32_t s_wtot[NW];
		if (lane == 63) s_wtot[wv] = incl;
		// local item slots (order irrelevant)
		uint32_t ls = 0, lo_ = 0;
		if (is_s) ls = atomicAdd(&s_misc[0], 1u);
		if (is_o) lo_ = atomicAdd(&s_misc[1], 1u);
		__syncthreads();
		uint32_t run = incl - c;
		for (int w = 0; w < wv; w++) run += s_wtot[w];
		if (tid == 0) {   // one reservation per workgroup and queue
			const uint32_t ts = s_misc[0], to = s_misc[1];
			s_misc[2] = ts ? atomicAdd(&q->n_sort, ts) : 0u;
			s_misc[3] = to ? atomicAdd(over_count, to) : 0u;
		}
		__syncthreads();
		const uint32_t gs = s_misc[2], go = s_misc[3];
		if (gs + s_misc[0] > sort_cap || go + s_misc[1] > over_cap) { if (tid == 0) q->err = 1u; }
		if ((uint32_t)tid < B) { s_off[tid] = run; s_cur[tid] = run; }
		if ((uint32_t)tid == B - 1) s_off[B] = run + c;
		if (is_s) { if (gs + ls < sort_cap) sort_items[gs + ls] = make_uint2(abs0 + run, c | item_flag); }
		else if (is_o) { if (go + lo_ < over_cap) over_items[go + lo_] = make_uint2(abs0 + run, c | item_flag); }
	}
	__syncthreads();
	if (in_regs) {
#pragma unroll
		for (int r = 0; r < RK; r++)
			if ((uint32_t)tid + (uint32_t)NT * r < n) dst[atomicAdd(&s_cur[breg[r]], 1u)] = kreg[r];
	} else {
		for (uint32_t i0 = tid; i0 < n; i0 += 8u * NT) {
			unsigned long long k8[8];
#pragma unroll
			for (int u = 0; u < 8; u++) { const uint32_t i = i0 + (uint32_t)u * NT; k8[u] = i < n ? src[i] : 0ull; }
#pragma unroll
			for (int u = 0; u < 8; u++)
				if (i0 + (uint32_t)u * NT < n) dst[atomicAdd(&s_cur[bucket_of(k8[u])], 1u)] = k8[u];
		}
	}
	__syncthreads();
}

// held to 4 waves per SIMD (128 VGPRs, no spill; left alone the compiler takes 151 -> 3 waves): with the two length classes below
// C4's sort 0.1937 -> 0.1648 ms (profiles/r05_radix_sort_experiments.txt; 5 / 6 waves: 72-76 B of scratch, 0.173 / 0.1645)
#ifndef GSR_RADIX_WAVES
#define GSR_RADIX_WAVES 4
#endif
#define GSR_RADIX_ATTR __attribute__((amdgpu_waves_per_eu(GSR_RADIX_WAVES, GSR_RADIX_WAVES)))
// RK: keys a thread holds in registers (lists of up to 256 RK keys are cut without re-reading them).  Round 5: two instantiations,
// launched one after the other, each taking the lists of ITS length class (lo < n <= hi) -- 16 for lists up to 4096 keys (the C4 regime:
// every tile ~3.4 k), 32 beyond.  With 32 for everything the kernel needed 76 VGPRs (6 waves per SIMD); a list's cut is one long
// dependent chain (load -> min / max -> histogram -> scan -> scatter -> bucket sorts), so the workgroups a CU can hold set the pace.
template <int RK>
__global__ __launch_bounds__(256) GSR_RADIX_ATTR void tile_radix_sort_kernel(const uint2* __restrict__ ranges,
                                                              uint64_t* __restrict__ keys, uint64_t* __restrict__ keys2,
                                                              uint32_t* __restrict__ point_list, uint32_t lo, uint32_t hi,
                                                              GsSortQ* __restrict__ q, uint2* __restrict__ seg_items, uint32_t seg_cap,
                                                              const GsCtl* __restrict__ ctl, uint32_t cap)
{
	__shared__ uint32_t whist[4][256];   // per-wave digit counts, then per-wave running offsets
	__shared__ uint32_t s_tot[4];
	__shared__ uint32_t s_maxrun;
	if (ctl->num_binned > cap || ctl->max_tile_count <= lo) return;   // see bin_scatter_kernel; no long list at all
	const uint2 range = ranges[blockIdx.x];
	const uint32_t n = range.y - range.x;
	if (n <= lo || n > hi) return;
	if (q != nullptr && (n > GSR_PART_REGS || ctl->n_long < GSR_SORT_MANY)) return;   // the queue pipeline's (tile_sort_kernel entered it)
	if (q == nullptr && ctl->max_tile_count > GSR_SORT_GIANT) return;   // launched for the wrong regime: the host re-launches with the pipeline
	const int tid = threadIdx.x, wv = tid >> 6;
	uint64_t* src = keys + range.x;
	uint64_t* dst = keys2 + range.x;

	// ---- fast path (sample-sort style): cut the list into B = 2^k depth buckets of equal key width (~<= 256 keys on
	// average), scatter the keys bucket by bucket into `keys2`, then every wave sorts whole buckets in registers
	// (gs_wave_sort_tile).  Two sweeps over the keys and one register sort instead of four counting passes; a bucket
	// with more than 1024 keys (depths piled up on a few values) sends the tile down the radix path below.
	{
		constexpr uint32_t BMAX = 1024;
		constexpr uint32_t GSR_BUCKET_AVG = 256;   // target keys per bucket (128 measured the same, 64 slower)
		__shared__ uint32_t s_off[BMAX + 1], s_cur[BMAX];
		__shared__ uint32_t s_min, s_max, s_over;
		if (tid == 0) { s_min = 0xffffffffu; s_max = 0u; s_over = 0u; }
		__syncthreads();
		// lists of up to 256 * RK keys are held in registers (RK independent loads per thread in flight at once): the
		// three sweeps below -- min/max, histogram, scatter -- then cost no further memory round trips
		const bool in_regs = n <= 256u * RK;
		uint64_t kreg[RK];
		if (in_regs) {
#pragma unroll
			for (int r = 0; r < RK; r++) {
				const uint32_t i = (uint32_t)tid + 256u * r;
				kreg[r] = i < n ? src[i] : ~0ull;
			}
		}
		uint32_t mn = 0xffffffffu, mx = 0u;
		if (in_regs) {
#pragma unroll
			for (int r = 0; r < RK; r++) {
				if ((uint32_t)tid + 256u * r < n) {
					const uint32_t d = (uint32_t)(kreg[r] >> 32);
					mn = min(mn, d); mx = max(mx, d);
				}
			}
		} else {
			for (uint32_t i = tid; i < n; i += 256) {
				const uint32_t d = (uint32_t)(src[i] >> 32);
				mn = min(mn, d); mx = max(mx, d);
			}
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
			mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
		}
		if ((tid & 63) == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
		__syncthreads();
		const uint32_t dmin = s_min, span = s_max - s_min;
		if (span > 0) {
			uint32_t B = 2;
			while (B < BMAX && B * GSR_BUCKET_AVG < n) B <<= 1;
			const int span_bits = 32 - __clz((int)span), logB = 31 - __clz((int)B);
			const int shift = max(0, span_bits - logB);            // (d - dmin) >> shift < B
			for (uint32_t b = tid; b <= B; b += 256) s_off[b] = 0u;
			__syncthreads();
			if (in_regs) {
#pragma unroll
				for (int r = 0; r < RK; r++)
					if ((uint32_t)tid + 256u * r < n) atomicAdd(&s_off[((uint32_t)(kreg[r] >> 32) - dmin) >> shift], 1u);
			} else {
				for (uint32_t i = tid; i < n; i += 256) atomicAdd(&s_off[((uint32_t)(src[i] >> 32) - dmin) >> shift], 1u);
			}
			__syncthreads();
			// exclusive scan of the B counts (4 per thread), and the largest count
			{
				uint32_t c[4], sum = 0, big = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const uint32_t b = 4 * tid + k;
					c[k] = b < B ? s_off[b] : 0u;
					sum += c[k];
					big = max(big, c[k]);
				}
				uint32_t incl = sum;
				const int lane = tid & 63;
#pragma unroll
				for (int o = 1; o < 64; o <<= 1) {
					const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
					if (lane >= o) incl += t;
				}
				if (lane == 63) s_tot[wv] = incl;
				if (big > GSR_SORT_LDS_MAX) s_over = 1u;
				__syncthreads();
				uint32_t run = incl - sum;
				for (int w = 0; w < wv; w++) run += s_tot[w];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const uint32_t b = 4 * tid + k;
					if (b < B) { s_off[b] = run; s_cur[b] = run; }
					run += c[k];
				}
				if (tid == 255) s_off[B] = n;
			}
			__syncthreads();
			if (!s_over) {
				if (in_regs) {
#pragma unroll
					for (int r = 0; r < RK; r++) {
						if ((uint32_t)tid + 256u * r < n) {
							const uint64_t k = kreg[r];
							dst[atomicAdd(&s_cur[((uint32_t)(k >> 32) - dmin) >> shift], 1u)] = k;
						}
					}
				} else {
					for (uint32_t i = tid; i < n; i += 256) {
						const uint64_t k = src[i];
						dst[atomicAdd(&s_cur[((uint32_t)(k >> 32) - dmin) >> shift], 1u)] = k;
					}
				}
				__threadfence_block();
				__syncthreads();
				const int lane = tid & 63;
				for (uint32_t b = wv; b < B; b += 4) {
					const uint32_t start = s_off[b], cnt = s_off[b + 1] - start;
					const uint64_t* bs = dst + start;
					uint32_t* bo = point_list + range.x + start;
					if (cnt == 0) continue;
					if (cnt <= 64) gs_wave_sort_tile<1>(bs, bo, cnt, lane);
					else if (cnt <= 128) gs_wave_sort_tile<2>(bs, bo, cnt, lane);
					else if (cnt <= 256) gs_wave_sort_tile<4>(bs, bo, cnt, lane);
					else if (cnt <= 512) gs_wave_sort_tile<8>(bs, bo, cnt, lane);
					else gs_wave_sort_tile<16>(bs, bo, cnt, lane);
				}
				return;
			}
		}
		__syncthreads();
	}
	if (q != nullptr) {
		// an overflowing cut while the queue pipeline runs anyway (a list > GSR_PART_REGS keys exists in this frame): the
		// list becomes a segment of its second stage (cut by its own sample splitters, buckets sorted all over the chip) -- its keys are
		// still where the scatter left them, in `keys`
		if (tid == 0) {
			const uint32_t at = atomicAdd(&q->n_seg, 1u);
			if (at < seg_cap) seg_items[at] = make_uint2(range.x, n | GSR_Q_IN_KEYS);
			else q->err = 1u;
		}
		return;
	}
	// wave w owns [wb, we): quarters rounded to multiples of 64 so that groups never straddle waves
	const uint32_t per = ((n + 3) / 4 + 63) / 64 * 64;
	const uint32_t wb = min(n, (uint32_t)wv * per), we = min(n, wb + per);
	if (tid == 0) s_maxrun = 0;
	for (int pass = 0; pass < 4; pass += 2) {   // an even number of passes leaves the data in `keys`
		tile_radix_pass(src, dst, 32 + 8 * pass, wb, we, whist, s_tot);
		tile_radix_pass(dst, src, 40 + 8 * pass, wb, we, whist, s_tot);
	}
	// Runs of equal depth are in scatter order.  Measure the longest one (capped): short runs (the normal case:
	// isolated float ties) are put in id order by one thread each; long runs (a fronto-parallel sheet of Gaussians
	// all at one depth) would make that quadratic in one thread, so the whole list is re-sorted as a full 64-bit
	// stable LSD sort instead: four passes on the id bits, then the four depth passes again.
	const uint32_t RUN_CAP = 32;
	uint32_t myrun = 0;
	for (uint32_t i = tid; i < n; i += 256) {
		const uint32_t dep = (uint32_t)(src[i] >> 32);
		if (i == 0 || (uint32_t)(src[i - 1] >> 32) != dep) {
			uint32_t e = i + 1;
			while (e < n && e - i <= RUN_CAP && (uint32_t)(src[e] >> 32) == dep) e++;
			myrun = max(myrun, e - i);
		}
	}
	if (myrun > 1) atomicMax(&s_maxrun, myrun);
	__syncthreads();
	const uint32_t maxrun = s_maxrun;
	if (maxrun > RUN_CAP) {
		for (int pass = 0; pass < 8; pass += 2) {
			tile_radix_pass(src, dst, 8 * pass, wb, we, whist, s_tot);
			tile_radix_pass(dst, src, 8 * pass + 8, wb, we, whist, s_tot);
		}
	} else if (maxrun > 1) {
		for (uint32_t i = tid; i < n; i += 256) {
			const uint32_t dep = (uint32_t)(src[i] >> 32);
			const bool starts = (i == 0 || (uint32_t)(src[i - 1] >> 32) != dep) && (i + 1 < n) && (uint32_t)(src[i + 1] >> 32) == dep;
			if (starts) {
				uint32_t e = i + 1;
				while (e < n && (uint32_t)(src[e] >> 32) == dep) e++;
				for (uint32_t a = i + 1; a < e; a++) {   // insertion sort of the run [i, e) by full key (= by id)
					const uint64_t v = src[a];
					uint32_t b = a;
					while (b > i && src[b - 1] > v) { src[b] = src[b - 1]; b--; }
					src[b] = v;
				}
			}
		}
		__syncthreads();
	}
	for (uint32_t i = tid; i < n; i += 256) point_list[range.x + i] = (uint32_t)src[i];
}


// ---- first cut of every long list, by SLICES: a list of n keys is cut by ceil(n / 8192) workgroups ------------------
// (One workgroup per list, round 4's first form, left the longest list of a skewed frame to one CU: 52 us for the 58 k-key
// list of the clustered scene, three sweeps of 57 keys per thread.)  A slice is <= 8192 consecutive list positions, held in
// registers (8 keys per thread).  Every slice workgroup of a list uses the same splitters -- the list's 1024 evenly spaced
// sample keys, sorted once per list -- and
//   tile_sort       the wave that has nothing to sort for a long list (tile_sort_kernel) appends the list's slices to the
//                   slice queue (one device atomic per long list) and sorts its samples: gs_enter_long_list;
//   slice_hist      per slice: bucket of every key (kept for the scatter: 4 B per key in the not yet written point_list),
//                   LDS histogram -> row of u16 counts in the queue memory;
//   slice_scatter   per slice: bucket starts = scan over the list's bucket totals, plus what the slices in front of this one
//                   put into each bucket; keys scattered into `keys2`; slice 0 enters the list's buckets into the work queue.
#define GSR_SLICE_KEYS GSR_PART_REGS
#define GSR_SLICE_ROW (GSR_PART_THREADS / 2)     // u16 counts per slice row
template <int KPL>
__device__ __forceinline__ void gs_sort_list_samples(const uint64_t* __restrict__ src, uint32_t n, uint64_t* __restrict__ out, int lane)
{
	uint64_t k[KPL];
#pragma unroll
	for (int e = 0; e < KPL; e++) k[e] = src[(uint32_t)(((unsigned long long)(e * 64 + lane) * n) / (64u * KPL))];
	gs_wave_sort_regs<KPL>(k, lane);
#pragma unroll
	for (int e = 0; e < KPL; e++) out[lane * KPL + e] = k[e];
}
__device__ __forceinline__ void gs_enter_long_list(uint2 range, GsSortQ* __restrict__ q, uint4* __restrict__ slice_items, uint32_t slice_cap,
                                                   const uint64_t* __restrict__ keys, uint64_t* __restrict__ keys2, int lane)
{
	const uint32_t n = range.y - range.x;
	const uint32_t nsl = (n + GSR_SLICE_KEYS - 1u) / GSR_SLICE_KEYS;
	uint32_t base = 0;
	if (lane == 0) base = atomicAdd(&q->n_slice, nsl);   // one device atomic per long list
	base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
	if (base + nsl > slice_cap) { if (lane == 0) q->err = 1u; return; }   // cannot happen: SortQueueLayout::slice_cap
	for (uint32_t sl = lane; sl < nsl; sl += 64) slice_items[base + sl] = make_uint4(range.x, n, sl, base);
	// samples: the keys at positions floor(i n / S), i = 0 .. S - 1 (n > 1024 >= S: distinct positions, distinct keys), sorted
	const uint32_t S = gs_sample_count(gs_bucket_count(n, (uint32_t)GSR_PART_THREADS));
	if (S <= 64) gs_sort_list_samples<1>(keys + range.x, n, keys2 + range.x, lane);
	else if (S <= 128) gs_sort_list_samples<2>(keys + range.x, n, keys2 + range.x, lane);
	else if (S <= 256) gs_sort_list_samples<4>(keys + range.x, n, keys2 + range.x, lane);
	else if (S <= 512) gs_sort_list_samples<8>(keys + range.x, n, keys2 + range.x, lane);
	else gs_sort_list_samples<16>(keys + range.x, n, keys2 + range.x, lane);
}

// bucket starts from the bucket totals (thread b < B: tot, what slices in front hold: pre), the buckets entered into the work
// queues when `enter` (<= 1024 keys -> sort item, more -> a segment for the next level; flag 0: the buckets live in keys2),
// the slice's keys scattered into dst.  Returns this thread's bucket start.
template <int NT, int RK>
__device__ __forceinline__ uint32_t gs_slice_place(uint32_t tot, uint32_t pre, bool enter, uint32_t B, uint32_t abs0, uint32_t cnt,
                                                   const uint64_t (&kreg)[RK], const uint32_t (&breg)[RK], uint64_t* __restrict__ dst,
                                                   GsSortQ* __restrict__ q, uint2* __restrict__ sort_items, uint32_t sort_cap,
                                                   uint2* __restrict__ seg_items, uint32_t seg_cap, uint32_t* s_cur, uint32_t* s_wtot,
                                                   uint32_t* s_misc)
{
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0) { s_misc[0] = 0u; s_misc[1] = 0u; }
	uint32_t incl = tot;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
		if (lane >= o) incl += t;
	}
	if (lane == 63) s_wtot[wv] = incl;
	__syncthreads();
	uint32_t start = incl - tot;
	for (int w = 0; w < wv; w++) start += s_wtot[w];
	if ((uint32_t)tid < B) s_cur[tid] = start + pre;
	const bool is_s = enter && tot > 0u && tot <= GSR_SORT_LDS_MAX, is_o = enter && tot > GSR_SORT_LDS_MAX;
	uint32_t ls = 0, lo_ = 0;
	if (is_s) ls = atomicAdd(&s_misc[0], 1u);
	if (is_o) lo_ = atomicAdd(&s_misc[1], 1u);
	__syncthreads();
	if (tid == 0) {   // one reservation per list and queue
		const uint32_t ts = s_misc[0], to = s_misc[1];
		s_misc[2] = ts ? atomicAdd(&q->n_sort, ts) : 0u;
		s_misc[3] = to ? atomicAdd(&q->n_seg, to) : 0u;
	}
	__syncthreads();
	const uint32_t gs = s_misc[2], go = s_misc[3];
	if (gs + s_misc[0] > sort_cap || go + s_misc[1] > seg_cap) { if (tid == 0) q->err = 1u; }
	if (is_s) { if (gs + ls < sort_cap) sort_items[gs + ls] = make_uint2(abs0 + start, tot); }
	else if (is_o) { if (go + lo_ < seg_cap) seg_items[go + lo_] = make_uint2(abs0 + start, tot); }
#pragma unroll
	for (int r = 0; r < RK; r++)
		if ((uint32_t)tid + (uint32_t)NT * r < cnt) dst[atomicAdd(&s_cur[breg[r]], 1u)] = kreg[r];
	return start;
}

// (A one-slice list finished by its slice_hist workgroup -- starts, scatter and the sort of its buckets by the workgroup's own
// sixteen waves, no queue round trip -- was built and measured SLOWER: clustered-scene sort 0.076 -> 0.097 ms, C4 through the
// pipeline 0.337 -> 0.351 ms: a 3.7 k-key list has 32 buckets, two per wave, while the chip-wide bucket_sort runs them all at once.)
__global__ __launch_bounds__(GSR_PART_THREADS) void slice_hist_kernel(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys2,
                                                                      uint32_t* __restrict__ bucket_ids, const GsSortQ* __restrict__ q,
                                                                      const uint4* __restrict__ slice_items, uint32_t slice_cap,
                                                                      uint16_t* __restrict__ rows, const GsCtl* __restrict__ ctl, uint32_t cap)
{
	constexpr int NT = GSR_PART_THREADS, RK = (int)(GSR_SLICE_KEYS / NT);
	__shared__ uint32_t s_cnt[GSR_SLICE_ROW];
	__shared__ unsigned long long s_tree[GSR_SLICE_ROW];
	// the guard words, the queue length and this workgroup's first item are fetched together: one round trip instead of three
	const uint32_t nb_ = ctl->num_binned, mt_ = ctl->max_tile_count, nq_ = q->n_slice;
	const uint4 item0 = blockIdx.x < slice_cap ? slice_items[blockIdx.x] : make_uint4(0u, 0u, 0u, 0u);
	if (nb_ > cap || mt_ <= GSR_SORT_LDS_MAX) return;   // see bin_scatter_kernel; no long list at all
	const int tid = threadIdx.x;
	const uint32_t nsl_all = min(nq_, slice_cap);
	for (uint32_t it = blockIdx.x; it < nsl_all; it += gridDim.x) {
		const uint4 item = it == blockIdx.x ? item0 : slice_items[it];
		const uint64_t* src = keys + item.x;
		const uint32_t n = item.y, lo = item.z * GSR_SLICE_KEYS, cnt = min(GSR_SLICE_KEYS, n - lo);
		const uint32_t B = gs_bucket_count(n, (uint32_t)NT), spb = gs_sample_count(B) / B;
		gs_build_splitter_tree(s_tree, reinterpret_cast<const unsigned long long*>(keys2 + item.x), B, spb);   // the list's sorted samples (gs_enter_long_list)
		uint64_t kreg[RK];
#pragma unroll
		for (int r = 0; r < RK; r++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)NT * r;
			kreg[r] = i < cnt ? src[lo + i] : ~0ull;
		}
		if ((uint32_t)tid < B) s_cnt[tid] = 0u;
		__syncthreads();
		// eight independent searches per thread (padding keys land in the last bucket and are dropped below)
		uint32_t breg[RK];
#pragma unroll
		for (int r = 0; r < RK; r++) breg[r] = 1u;
		for (uint32_t lv = 1; lv < B; lv <<= 1) {
#pragma unroll
			for (int r = 0; r < RK; r++) breg[r] = 2u * breg[r] + (s_tree[breg[r]] <= kreg[r] ? 1u : 0u);
		}
#pragma unroll
		for (int r = 0; r < RK; r++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)NT * r;
			breg[r] -= B;
			if (i < cnt) {
				atomicAdd(&s_cnt[breg[r]], 1u);
				bucket_ids[item.x + lo + i] = breg[r];
			}
		}
		__syncthreads();
		if ((uint32_t)tid < B) rows[(size_t)(item.w + item.z) * GSR_SLICE_ROW + tid] = (uint16_t)s_cnt[tid];   // <= 8192
		__syncthreads();
	}
}

__global__ __launch_bounds__(GSR_PART_THREADS) void slice_scatter_kernel(const uint64_t* __restrict__ keys, uint64_t* __restrict__ keys2,
                                                                         const uint32_t* __restrict__ bucket_ids, GsSortQ* __restrict__ q,
                                                                         const uint4* __restrict__ slice_items, uint32_t slice_cap,
                                                                         const uint16_t* __restrict__ rows, uint2* __restrict__ sort_items,
                                                                         uint32_t sort_cap, uint2* __restrict__ seg_items, uint32_t seg_cap,
                                                                         const GsCtl* __restrict__ ctl, uint32_t cap)
{
	constexpr int NT = GSR_PART_THREADS, NW = NT / 64, RK = (int)(GSR_SLICE_KEYS / NT);
	__shared__ uint32_t s_cur[GSR_SLICE_ROW], s_wtot[NW], s_misc[4];
	const uint32_t nb_ = ctl->num_binned, mt_ = ctl->max_tile_count, nq_ = q->n_slice;   // one round trip, see slice_hist_kernel
	const uint4 item0 = blockIdx.x < slice_cap ? slice_items[blockIdx.x] : make_uint4(0u, 0u, 0u, 0u);
	if (nb_ > cap || mt_ <= GSR_SORT_LDS_MAX) return;
	const int tid = threadIdx.x;
	const uint32_t nsl_all = min(nq_, slice_cap);
	for (uint32_t it = blockIdx.x; it < nsl_all; it += gridDim.x) {
		const uint4 item = it == blockIdx.x ? item0 : slice_items[it];
		const uint32_t n = item.y, sl = item.z, lo = sl * GSR_SLICE_KEYS, cnt = min(GSR_SLICE_KEYS, n - lo);
		const uint32_t nsl = (n + GSR_SLICE_KEYS - 1u) / GSR_SLICE_KEYS;
		const uint32_t B = gs_bucket_count(n, (uint32_t)NT);
		uint64_t kreg[RK];
		uint32_t breg[RK];
#pragma unroll
		for (int r = 0; r < RK; r++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)NT * r;
			kreg[r] = i < cnt ? keys[item.x + lo + i] : 0ull;
			breg[r] = i < cnt ? bucket_ids[item.x + lo + i] : 0u;
		}
		// thread b: the list's total of bucket b and what the slices in front of this one hold of it
		uint32_t tot = 0, pre = 0;
		if ((uint32_t)tid < B) {
			const uint16_t* col = rows + (size_t)item.w * GSR_SLICE_ROW + tid;
#pragma unroll 8
			for (uint32_t r = 0; r < nsl; r++) {
				const uint32_t c = col[(size_t)r * GSR_SLICE_ROW];
				tot += c;
				pre += r < sl ? c : 0u;
			}
		}
		gs_slice_place<NT, RK>(tot, pre, sl == 0u, B, item.x, cnt, kreg, breg, keys2 + item.x, q, sort_items, sort_cap, seg_items, seg_cap,
		                       s_cur, s_wtot, s_misc);   // slice 0 enters the list's buckets into the queues
		__syncthreads();
	}
}

// second level: the oversized buckets of the first cut (and the overflowing lists tile_radix_sort_kernel hands over) cut again,
// each by its OWN sample splitters (gs_partition_segment), into the other key buffer; workgroups stride over the segment queue
// `in_items`, what is still oversized goes to `out_items` (the fallback's queue)
__global__ __launch_bounds__(GSR_PART_THREADS) void segment_partition_kernel(uint64_t* __restrict__ keys, uint64_t* __restrict__ keys2,
                                                                            GsSortQ* __restrict__ q, uint2* __restrict__ sort_items,
                                                                            uint32_t sort_cap, const uint2* __restrict__ in_items,
                                                                            const uint32_t* __restrict__ in_count, uint32_t seg_cap,
                                                                            uint2* __restrict__ out_items, uint32_t* __restrict__ out_count,
                                                                            const GsCtl* __restrict__ ctl, uint32_t cap)
{
	__shared__ uint32_t s_off[GSR_PART_THREADS / 2 + 1], s_cur[GSR_PART_THREADS / 2], s_misc[4];
	__shared__ unsigned long long s_samp[2 * GSR_PART_THREADS];
	const uint32_t nb_ = ctl->num_binned, mt_ = ctl->max_tile_count, nq_ = *in_count;   // one round trip
	if (nb_ > cap || mt_ <= GSR_SORT_LDS_MAX) return;
	const uint32_t nseg = min(nq_, seg_cap);
	for (uint32_t it = blockIdx.x; it < nseg; it += gridDim.x) {
		const uint2 sg = in_items[it];
		const bool in_keys = (sg.y & GSR_Q_IN_KEYS) != 0u;     // where the segment's keys are; its buckets go to the other buffer
		const uint32_t cnt = sg.y & ~GSR_Q_IN_KEYS;
		gs_partition_segment<GSR_PART_THREADS>((in_keys ? keys : keys2) + sg.x, (in_keys ? keys2 : keys) + sg.x, sg.x, cnt,
		                                       in_keys ? 0u : GSR_Q_IN_KEYS, q, sort_items, sort_cap, out_items, seg_cap,
		                                       out_count, s_off, s_cur, s_misc, s_samp);
	}
}

// last resort for a segment that two levels of cuts could not break up: stable LSD counting sort on the 32
// depth bits (ping-pong src <-> tmp, data ends in src), runs of equal depth put in id order afterwards; one workgroup
// (256 threads) per segment.
__device__ __forceinline__ void gs_radix_sort_segment(uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint32_t* __restrict__ out,
                                                      uint32_t n, uint32_t (*whist)[256], uint32_t* s_tot, uint32_t* s_maxrun)
{
	const int tid = threadIdx.x, wv = tid >> 6;
	// wave w owns [wb, we): quarters rounded to multiples of 64 so that groups never straddle waves
	const uint32_t per = ((n + 3) / 4 + 63) / 64 * 64;
	const uint32_t wb = min(n, (uint32_t)wv * per), we = min(n, wb + per);
	if (tid == 0) *s_maxrun = 0;
	for (int pass = 0; pass < 4; pass += 2) {   // an even number of passes leaves the data in `src`
		tile_radix_pass(src, dst, 32 + 8 * pass, wb, we, whist, s_tot);
		tile_radix_pass(dst, src, 40 + 8 * pass, wb, we, whist, s_tot);
	}
	// Runs of equal depth are in scatter order.  Measure the longest one (capped): short runs (the normal case:
	// isolated float ties) are put in id order by one thread each; long runs (a fronto-parallel sheet of Gaussians
	// all at one depth) would make that quadratic in one thread, so the whole list is re-sorted as a full 64-bit
	// stable LSD sort instead: four passes on the id bits, then the four depth passes again.
	const uint32_t RUN_CAP = 32;
	uint32_t myrun = 0;
	for (uint32_t i = tid; i < n; i += 256) {
		const uint32_t dep = (uint32_t)(src[i] >> 32);
		if (i == 0 || (uint32_t)(src[i - 1] >> 32) != dep) {
			uint32_t e = i + 1;
			while (e < n && e - i <= RUN_CAP && (uint32_t)(src[e] >> 32) == dep) e++;
			myrun = max(myrun, e - i);
		}
	}
	if (myrun > 1) atomicMax(s_maxrun, myrun);
	__syncthreads();
	const uint32_t maxrun = *s_maxrun;
	if (maxrun > RUN_CAP) {
		for (int pass = 0; pass < 8; pass += 2) {
			tile_radix_pass(src, dst, 8 * pass, wb, we, whist, s_tot);
			tile_radix_pass(dst, src, 8 * pass + 8, wb, we, whist, s_tot);
		}
	} else if (maxrun > 1) {
		for (uint32_t i = tid; i < n; i += 256) {
			const uint32_t dep = (uint32_t)(src[i] >> 32);
			const bool starts = (i == 0 || (uint32_t)(src[i - 1] >> 32) != dep) && (i + 1 < n) && (uint32_t)(src[i + 1] >> 32) == dep;
			if (starts) {
				uint32_t e = i + 1;
				while (e < n && (uint32_t)(src[e] >> 32) == dep) e++;
				for (uint32_t a = i + 1; a < e; a++) {   // insertion sort of the run [i, e) by full key (= by id)
					const uint64_t v = src[a];
					uint32_t b = a;
					while (b > i && src[b - 1] > v) { src[b] = src[b - 1]; b--; }
					src[b] = v;
				}
			}
		}
		__syncthreads();
	}
	for (uint32_t i = tid; i < n; i += 256) out[i] = (uint32_t)src[i];
	__syncthreads();
}

// every wave of the launch sorts buckets (<= 1024 keys each) from the queue in registers; in front of that, workgroups take
// the fallback segments (normally none)
__global__ __launch_bounds__(256) void bucket_sort_kernel(uint64_t* __restrict__ keys, uint64_t* __restrict__ keys2,
                                                          uint32_t* __restrict__ point_list, const GsSortQ* __restrict__ q,
                                                          const uint2* __restrict__ sort_items, uint32_t sort_cap,
                                                          const uint2* __restrict__ fall_items, uint32_t fall_cap,
                                                          const GsCtl* ctl, uint32_t cap, uint32_t* err_dst, uint32_t* host_err)
{
	// (ctl and err_dst point into the same control block: neither is __restrict__)
	__shared__ uint32_t whist[4][256];   // per-wave digit counts, then per-wave running offsets
	__shared__ uint32_t s_tot[4];
	__shared__ uint32_t s_maxrun;
	const uint32_t nb_ = ctl->num_binned, mt_ = ctl->max_tile_count, nfq_ = q->n_fall, nsq_ = q->n_sort;   // one round trip
	if (nb_ > cap || mt_ <= GSR_SORT_LDS_MAX) return;
	// the pipeline's overflow flag (a SortQueueLayout bound violated: items were dropped, point_list is not sorted) is final
	// here -- every kernel that can raise it ran before this one: mirrored into the frame's control words (bit 1 of
	// err_overflow), where debug-mode calls and gsr_inspect_counts see it, and into the device's sticky host word, which the NEXT
	// gsr_forward / gsr_backward of the process checks on entry: a non-debug run fails loudly too, one call late
	if (blockIdx.x == 0 && threadIdx.x == 0 && q->err != 0u) {
		atomicOr(err_dst, 2u);
		if (host_err != nullptr) *host_err = 2u;
	}
	const uint32_t nf = min(nfq_, fall_cap);
	for (uint32_t it = blockIdx.x; it < nf; it += gridDim.x) {
		const uint2 sg = fall_items[it];
		const bool in_keys = (sg.y & GSR_Q_IN_KEYS) != 0u;
		gs_radix_sort_segment((in_keys ? keys : keys2) + sg.x, (in_keys ? keys2 : keys) + sg.x, point_list + sg.x, sg.y & ~GSR_Q_IN_KEYS,
		                      whist, s_tot, &s_maxrun);
	}
	const int lane = threadIdx.x & 63;
	const uint32_t nitems = min(nsq_, sort_cap);
	const uint32_t nwaves = gridDim.x * 4u;
	for (uint32_t it = blockIdx.x * 4u + (threadIdx.x >> 6); it < nitems; it += nwaves) {
		const uint2 item = sort_items[it];
		const uint32_t cnt = item.y & ~GSR_Q_IN_KEYS;
		const uint64_t* bs = ((item.y & GSR_Q_IN_KEYS) ? keys : keys2) + item.x;
		uint32_t* bo = point_list + item.x;
		if (cnt <= 64) gs_wave_sort_tile<1>(bs, bo, cnt, lane);
		else if (cnt <= 128) gs_wave_sort_tile<2>(bs, bo, cnt, lane);
		else if (cnt <= 256) gs_wave_sort_tile<4>(bs, bo, cnt, lane);
		else if (cnt <= 512) gs_wave_sort_tile<8>(bs, bo, cnt, lane);
		else gs_wave_sort_tile<16>(bs, bo, cnt, lane);
	}
}

// queue storage behind keys2 (BinLayout::queue): [GsSortQ][sort items][segment items][fallback items][slice items][slice rows]
struct SortQueueLayout {
	uint32_t sort_cap, seg_cap, slice_cap;
	size_t sort_items, seg_items, slice_items, rows, total;
	SortQueueLayout(size_t R, int T)
	{
		sort_cap = (uint32_t)(R / 16 + 4 * (size_t)T + 4096);
		seg_cap = (uint32_t)(R / 512 + (size_t)T + 64);
		// sum over the long lists of ceil(n / 8192) <= R / 8192 + their number, and a long list has more than 1024 keys
		slice_cap = (uint32_t)(R / GSR_SLICE_KEYS + std::min((size_t)T, R / GSR_SORT_LDS_MAX) + 1);
		sort_items = 256;
		seg_items = sort_items + sizeof(uint2) * (size_t)sort_cap;
		slice_items = (seg_items + sizeof(uint2) * 2 * (size_t)seg_cap + 15) & ~(size_t)15;
		rows = slice_items + sizeof(uint4) * (size_t)slice_cap;
		total = rows + sizeof(uint16_t) * GSR_SLICE_ROW * (size_t)slice_cap + 256;
	}
};
size_t sort_queue_bytes(size_t R, int T) { return SortQueueLayout(R, T).total; }

#ifndef GSR_RADIX_RK_SHORT
#define GSR_RADIX_RK_SHORT 16   // the per-list kernel's register-resident key count for the shorter class of long lists (x 256 keys)
#endif
void launch_tile_sort(int T, bool with_short, int long_level, const uint2* ranges, uint64_t* keys, uint64_t* keys2,
                      uint32_t* point_list, char* queue, size_t R, GsCtl* ctl, uint32_t cap, uint32_t* host_err, hipStream_t s)
{
	// <= GSR_SORT_LDS_MAX keys: register bitonic network, one wave per tile (tile_sort_kernel).
	// long_level 1 (lists up to GSR_SORT_GIANT keys, e.g. the C4 regime: every tile ~3.7 k): one 256-thread workgroup per tile
	// cuts and sorts its list (tile_radix_sort_kernel, the round-2/3 kernel: 0.19 ms at C4 against 0.34 for the pipeline).
	// long_level 2 (a longer list exists in the frame): the queue pipeline -- lists cut in slices of 8192 keys by 1024-thread
	// workgroups, an overflowing bucket cut again, the buckets sorted by all waves of the chip -- for the lists beyond 8192 keys
	// and, unless the frame has MANY long lists, for all the others beyond 1024 too.
	if (long_level <= 1) {
		if (with_short)
			hipLaunchKernelGGL(tile_sort_kernel, dim3((T + 3) / 4), dim3(256), 0, s, T, ranges, keys, point_list, ctl, cap, (GsSortQ*)nullptr,
			                   (uint4*)nullptr, 0u, (uint64_t*)nullptr);
		if (long_level == 1) {
			hipLaunchKernelGGL(tile_radix_sort_kernel<GSR_RADIX_RK_SHORT>, dim3(T), dim3(256), 0, s, ranges, keys, keys2, point_list, GSR_SORT_LDS_MAX,
			                   256u * GSR_RADIX_RK_SHORT, (GsSortQ*)nullptr, (uint2*)nullptr, 0u, ctl, cap);
			hipLaunchKernelGGL(tile_radix_sort_kernel<32>, dim3(T), dim3(256), 0, s, ranges, keys, keys2, point_list, 256u * GSR_RADIX_RK_SHORT,
			                   0xffffffffu, (GsSortQ*)nullptr, (uint2*)nullptr, 0u, ctl, cap);
		}
		return;
	}
	const SortQueueLayout ql(R, T);
	const uint32_t sort_cap = ql.sort_cap, seg_cap = ql.seg_cap, slice_cap = ql.slice_cap;
	GsSortQ* q = reinterpret_cast<GsSortQ*>(queue);
	uint2* sort_items = reinterpret_cast<uint2*>(queue + ql.sort_items);
	uint2* seg_items = reinterpret_cast<uint2*>(queue + ql.seg_items);
	uint2* fall_items = seg_items + seg_cap;
	uint4* slice_items = reinterpret_cast<uint4*>(queue + ql.slice_items);
	uint16_t* rows = reinterpret_cast<uint16_t*>(queue + ql.rows);
	(void)hipMemsetAsync(q, 0, sizeof(GsSortQ), s);
	// short lists sorted; long lists entered into the slice queue, their samples sorted (gs_enter_long_list)
	hipLaunchKernelGGL(tile_sort_kernel, dim3((T + 3) / 4), dim3(256), 0, s, T, ranges, keys, point_list, ctl, cap, q, slice_items, slice_cap, keys2);
	// a frame with MANY long lists (>= GSR_SORT_MANY, e.g. C4 plus one giant tile): those of up to 8192 keys go to the one-workgroup-
	// per-list kernel (throughput: 0.19 ms for C4's 4346 lists against 0.34 ms through the queues); a cut of it that overflows
	// (depths piled up) becomes a segment of the pipeline.  With few long lists (clustered scene: 341) that kernel would be a
	// latency -- its longest list, alone on a CU, 42 us -- in front of a pipeline that takes them in its stride: it leaves at once.
	hipLaunchKernelGGL(tile_radix_sort_kernel<GSR_RADIX_RK_SHORT>, dim3(T), dim3(256), 0, s, ranges, keys, keys2, point_list, GSR_SORT_LDS_MAX,
	                   256u * GSR_RADIX_RK_SHORT, q, seg_items, seg_cap, ctl, cap);
	hipLaunchKernelGGL(tile_radix_sort_kernel<32>, dim3(T), dim3(256), 0, s, ranges, keys, keys2, point_list, 256u * GSR_RADIX_RK_SHORT, 0xffffffffu, q,
	                   seg_items, seg_cap, ctl, cap);
	hipLaunchKernelGGL(slice_hist_kernel, dim3(512), dim3(GSR_PART_THREADS), 0, s, keys, keys2, point_list, q, slice_items, slice_cap, rows, ctl, cap);
	hipLaunchKernelGGL(slice_scatter_kernel, dim3(512), dim3(GSR_PART_THREADS), 0, s, keys, keys2, point_list, q, slice_items, slice_cap, rows,
	                   sort_items, sort_cap, seg_items, seg_cap, ctl, cap);
	// oversized buckets (lists beyond 64 k keys; sampling noise) cut once more, what is STILL oversized: counting sort
	hipLaunchKernelGGL(segment_partition_kernel, dim3(128), dim3(GSR_PART_THREADS), 0, s, keys, keys2, q, sort_items, sort_cap,
	                   seg_items, &q->n_seg, seg_cap, fall_items, &q->n_fall, ctl, cap);
	hipLaunchKernelGGL(bucket_sort_kernel, dim3(2048), dim3(256), 0, s, keys, keys2, point_list, q, sort_items, sort_cap, fall_items, seg_cap, ctl, cap,
	                   &ctl->err_overflow, host_err);
}


// ------------------------------------------------------------------------------------------------
// composite_fwd: 256 threads = 4 wave64; wave w owns one 8x8 pixel block of the tile (gs_pixel_of_thread).
// Instances are staged 256 at a time through LDS as three float4 planes.  Each wave then culls the
// staged list for ITS block 64 instances at a time: lane l tests instance l against the block's box
// (gs_box_may_touch), the ballot is the list of instances worth evaluating, and the wave walks its set
// bits in order (s_ff1) reading the records with wave-uniform (broadcast) ds_read_b128.  In the C3
// scene ~3/4 of the (wave, instance) pairs die in the ballot at ~0.5 VALU op per pixel instead of ~20.
// Per-lane early termination (`done`); a wave stops when all its lanes are done, the workgroup when
// all four waves are.
// XCD-aware tile order: workgroup b runs on XCD b%8 (observed placement, speed only); each XCD is
// given a contiguous band of tiles so that neighbouring tiles, which share Gaussians, gather the
// same records from the same 4 MiB L2.
// NOCULL (debugging / parity A/B, gsr_set_option("cull", 0)): every staged instance is evaluated by every wave and
// the pcut pre-test is replaced by the domain bound of gs_exp -- the culling must not change a single bit.
#ifdef GSR_AB_VARIANTS   // the per-wave (8x8) walk: superseded by the per-quarter kernel below, kept for A/B builds only (make AB=1)
template <bool NOCULL>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    int T, int chunk, int gx, int W, int H, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const GsRec* __restrict__ recs, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_median, float* __restrict__ out_opacity,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ med_pos_out,
    const GsCtl* __restrict__ ctl, uint32_t cap, uint32_t max_sorted, const uint32_t* __restrict__ tile_order,
    uint32_t* __restrict__ staged_out)
{
	__shared__ float4 sA[256];
	__shared__ float4 sB[256];
	__shared__ float4 sC[256];
	// launched ahead of the host's read-back (see bin_scatter_kernel): leave when the binning buffer was too small or
	// a list is longer than what the sort kernels launched with this call handle (its point_list is not written)
	if (ctl->num_binned > cap || ctl->max_tile_count > max_sorted) return;
	// XCD-banded static order, or (skewed frames) longest list first: launch_tile_order_fwd
	const int tile = tile_order ? ((int)blockIdx.x < T ? (int)tile_order[blockIdx.x] : T) : (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	if ((!tile_order && (int)(blockIdx.x >> 3) >= chunk) || tile >= T) return;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int tx = tile % gx, ty = tile / gx;
	int lx, ly;
	gs_pixel_of_thread(tid, lx, ly);
	const int px = tx * GSR_BLOCK_X + lx, py = ty * GSR_BLOCK_Y + ly;
	const bool inside = px < W && py < H;
	const float pixfx = (float)px, pixfy = (float)py;
	// this wave's pixel box (pixel centres), clipped to the image
	const float bx0 = (float)(tx * GSR_BLOCK_X + (((tid >> 6) & 1) << 3));
	const float by0 = (float)(ty * GSR_BLOCK_Y + (((tid >> 6) >> 1) << 3));
	const float bx1 = fminf(bx0 + 7.f, (float)(W - 1)), by1 = fminf(by0 + 7.f, (float)(H - 1));
	const uint2 range = ranges[tile];
	const int total = (int)(range.y - range.x);
	bool done = !inside;
	float T_ = 1.0f;
	uint32_t last_contributor = 0;
	v2f acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};   // (R, G), (B, depth)
	// median candidate: list position (+1) and transmittance in front of the LAST applied instance that saw
	// T > 0.5; whether it really crossed 0.5 -- and its depth / weight / id -- is settled once per pixel after
	// the walk (two selects and one compare per pair instead of three and two)
	uint32_t med_pos = 0;
	float med_T = 0.f;

	int staged = 0;   // instances this workgroup staged before every pixel of the tile had saturated (one store at the end: the bench's byte count)
	for (int base = 0; base < total; base += 256) {
		if (__syncthreads_and(done)) break;
		const int cnt = min(256, total - base);
		staged += cnt;
		if (tid < cnt) {
			const uint32_t id = point_list[range.x + base + tid];
			const GsRec* r = recs + id;
			sA[tid] = r->q0;
			// staged planes: B = {-0.5c, op, id, pcut}, C = {r, g, b, depth}: the four blended channels sit in one
			// aligned register quad, so the accumulation is two packed FMAs
			float4 b = r->q1;
			float4 c = r->q2;
			c.w = b.z;
			b.z = __int_as_float((int)id);
			sB[tid] = b;
			sC[tid] = c;
		}
		__syncthreads();
		for (int sub = 0; sub < cnt; sub += 64) {
			if (__ballot(!done) == 0ull) break;   // every pixel of this wave has saturated
			const int jl = sub + lane;
			bool hit = false;
			if (jl < cnt) hit = NOCULL ? true : gs_box_may_touch(sA[jl], sB[jl], bx0, by0, bx1, by1);
			unsigned long long m = __ballot(hit);
			// walk the surviving instances in list order; the body is straight-line predicated code (no
			// per-test branches, no short-circuit evaluation: within a hit block about half of the lanes are
			// live, so branches would almost never be wave-uniformly skippable); records come from LDS with
			// wave-uniform reads.
			while (m) {
				const int j = sub + __ffsll((long long)m) - 1;
				m &= m - 1;
				const float4 A = sA[j], B = sB[j], Cc = sC[j];
				const float dx = A.x - pixfx, dy = A.y - pixfy;
				// power = -0.5(a dx^2 + c dy^2) - b dx dy with pre-scaled conic (forward.cu:338)
				const float power = FMA(A.w * dx, dy, FMA(B.x * dy, dy, (A.z * dx) * dx));
				const float alpha = fminf(0.99f, B.y * gs_exp(power));
				// forward.cu:339 (+ pcut), :346
				const bool valid = (!done) & (power <= 0.0f) & (power >= (NOCULL ? -80.0f : B.w)) & (!(alpha < 1.0f / 255.0f));
				const float test_T = T_ * (1 - alpha);
				const bool stop = valid & (test_T < 0.0001f);                        // forward.cu:357-361
				done = done | stop;
				const bool apply = valid & (!stop);
				// a skipped pair contributes with weight exactly 0: fma(c, 0, acc) == acc for finite c, so one select
				// on the weight replaces four on the accumulators (results stay bit-identical)
				const float w = apply ? alpha * T_ : 0.f;
				acc01 = vfma(v2f{Cc.x, Cc.y}, v2f{w, w}, acc01);
				acc23 = vfma(v2f{Cc.z, Cc.w}, v2f{w, w}, acc23);
				const bool medc = apply & (T_ > 0.5f);                               // forward.cu:368 (first half)
				med_pos = medc ? (uint32_t)(base + j + 1) : med_pos;
				med_T = medc ? T_ : med_T;
				T_ = apply ? test_T : T_;
				last_contributor = apply ? (uint32_t)(base + j + 1) : last_contributor;
				// (no per-pair "whole wave saturated" test: it cost two VALU and a scalar dependency per pair, 0.035 ms
				// at C3, to save at most the tail of one 64-instance batch; saturation is checked per batch above)
			}
		}
	}
	final_T[(size_t)tile * GSR_TILE_PIX + tid] = T_;
	n_contrib[(size_t)tile * GSR_TILE_PIX + tid] = last_contributor;
	uint32_t med_final = 0;   // list position (+1) of the median Gaussian, 0 = the transmittance never crossed 0.5
	if (inside) {
		// median depth / weight / id (forward.cu:368-373): the candidate crossed 0.5 iff its test_T < 0.5; alpha is
		// recomputed with the very operations of the walk, so the result is the bit pattern the walk would have kept
		float median_D = 15.0f, median_weight = 0.f;
		int median_id = 0;
		if (med_pos != 0) {
			const uint32_t id = point_list[range.x + med_pos - 1];
			const GsRec* r = recs + id;
			const float4 A = r->q0, B = r->q1;
			const float dx = A.x - pixfx, dy = A.y - pixfy;
			const float power = FMA(A.w * dx, dy, FMA(B.x * dy, dy, (A.z * dx) * dx));
			const float alpha = fminf(0.99f, B.y * gs_exp(power));
			if (med_T * (1 - alpha) < 0.5f) {
				median_D = B.z;
				median_weight = alpha * med_T;
				median_id = (int)id;
				med_final = med_pos;
			}
		}
		const size_t HW = (size_t)H * W;
		const size_t pix_id = (size_t)W * py + px;
		out_color[pix_id] = acc01.x;
		out_color[HW + pix_id] = acc01.y;
		out_color[2 * HW + pix_id] = acc23.x;
		out_depth[pix_id] = acc23.y;
		out_median[pix_id] = median_D;
		out_median[HW + pix_id] = median_weight;
		out_median[2 * HW + pix_id] = (float)median_id;
		out_opacity[pix_id] = 1 - T_;
	}
	med_pos_out[(size_t)tile * GSR_TILE_PIX + tid] = med_final;
	if (tid == 0 && staged_out != nullptr) staged_out[tile] = (uint32_t)staged;
}

#endif   // GSR_AB_VARIANTS

// ------------------------------------------------------------------------------------------------
// composite_fwd with per-quarter instance lists.  In the kernel above a wave (one 8x8 pixel block) walks every staged
// instance that may touch its block, and within such a block only about half of the lanes are live.  Here every
// 16-lane quarter of the wave owns a 4x4 pixel block and walks ITS OWN list of instances: the staging thread, which
// holds the record in registers anyway, tests it against the tile's sixteen 4x4 blocks at once (gs_quarter_mask), each
// wave compacts the four lists of its quarters (ballot + mbcnt) and the walk then takes list entry i of every quarter
// in the same instruction -- records come from LDS with per-quarter addresses, which a b128 read serves in its four
// 16-lane passes at no extra cost.  A wave walks max over its quarters instead of the 8x8 count: 0.725x the iterations
// at C3 (census: tools/scene_stats.py), quarters re-synchronise once per 256-instance batch.  Exhausted quarters read a
// sentinel record of opacity 0.  Per pixel the sequence of applied instances is unchanged: results are bit-identical.
#define GSR_FWD_PLANE 257                      // records per staged plane (256 + the sentinel)
#define GSR_FWD_SENT_OFF (256 * 16)            // byte offset of the sentinel record inside a plane
#define GSR_FWD_LIST 264                       // list entries per quarter (u16 byte offsets): 256 + two groups of sentinels
#define GSR_FWD_NONE 0xffffu

// FX: exp on the transcendental unit (gs_exp_hw) instead of the reproducible 9-instruction gs_exp -- the opt-in
// `fast_exp` mode (docs/DESIGN_history_r1-r4.md s4.5): outputs then agree with the bit-exact mode to ~1e-6 relative except at threshold
// flips (tests/test_gpu_fastexp.py attributes every one of them), and the backward must run in the same mode.
#ifndef GSR_FWD_WAVES
#define GSR_FWD_WAVES 7      // waves per SIMD the register allocation is held to
#endif
template <bool NOCULL, bool FX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GSR_FWD_WAVES, GSR_FWD_WAVES))) void composite_fwd_quarter_kernel(
    int T, int chunk, int gx, int W, int H, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const GsRec* __restrict__ recs, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_median, float* __restrict__ out_opacity,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ med_pos_out,
    GsCtl* __restrict__ ctl, uint32_t cap, uint32_t max_sorted, const uint32_t* __restrict__ tile_order,
    uint32_t* __restrict__ staged_out)
{
	__shared__ float4 sRec[3 * GSR_FWD_PLANE];   // planes A, B, C (as above), slot 256 of each = the sentinel
	__shared__ uint16_t sMask[256];
	__shared__ __attribute__((aligned(16))) uint16_t sList[4][4][GSR_FWD_LIST];
	const uint32_t num_binned = ctl->num_binned;
	if (num_binned > cap || ctl->max_tile_count > max_sorted) return;   // see composite_fwd_kernel
	// the block masks computed below stay behind for composite_bwd (gs_qmask_ptr)
	uint16_t* __restrict__ qmask = gs_qmask_ptr(point_list, num_binned);
	if (!NOCULL && blockIdx.x == 0 && threadIdx.x == 0) ctl->has_qmask = 1u;
	if (FX && blockIdx.x == 0 && threadIdx.x == 0) ctl->opts |= GSR_CTL_OPT_FAST_EXP;
	// XCD-banded static order, or (skewed frames) longest list first: launch_tile_order_fwd
	const int tile = tile_order ? ((int)blockIdx.x < T ? (int)tile_order[blockIdx.x] : T) : (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	if ((!tile_order && (int)(blockIdx.x >> 3) >= chunk) || tile >= T) return;
	const int tid = threadIdx.x;
	const int lane = tid & 63, w = tid >> 6, q = lane >> 4;
	const int tx = tile % gx, ty = tile / gx;
	// lane -> pixel: quarter q of wave w is the 4x4 block (2*(w&1) + (q&1), 2*(w>>1) + (q>>1)) of the tile
	const int lx = ((w & 1) << 3) + ((q & 1) << 2) + (lane & 3);
	const int ly = ((w >> 1) << 3) + ((q >> 1) << 2) + ((lane >> 2) & 3);
	const int slot = (w << 6) + ((ly & 7) << 3) + (lx & 7);   // this pixel's place in the tile-major per-pixel arrays
	const int px = tx * GSR_BLOCK_X + lx, py = ty * GSR_BLOCK_Y + ly;
	const bool inside = px < W && py < H;
	const float pixfx = (float)px, pixfy = (float)py;
	const float tx0 = (float)(tx * GSR_BLOCK_X), ty0 = (float)(ty * GSR_BLOCK_Y);
	uint32_t allq = 0;   // 4x4 blocks of this tile that hold at least one pixel of the image
#pragma unroll
	for (int b = 0; b < 16; b++)
		if (tx * GSR_BLOCK_X + 4 * (b & 3) < W && ty * GSR_BLOCK_Y + 4 * (b >> 2) < H) allq |= 1u << b;
	const uint2 range = ranges[tile];
	const int total = (int)(range.y - range.x);
	bool done = !inside;
	float T_ = 1.0f;
	uint32_t last_contributor = 0;
	v2f acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
	uint32_t med_pos = 0;
	float med_T = 0.f;
	if (tid < 3) sRec[tid * GSR_FWD_PLANE + 256] = make_float4(0.f, 0.f, 0.f, 0.f);   // opacity 0: alpha = 0, never valid
	const char* rec_base = reinterpret_cast<const char*>(sRec);
	const uint16_t* my_list = &sList[w][q][0];

	int staged = 0;   // instances this workgroup staged before every pixel of the tile had saturated (one store at the end: the bench's byte count)
	for (int base = 0; base < total; base += 256) {
		if (__syncthreads_and(done)) break;
		const int cnt = min(256, total - base);
		staged += cnt;
		uint32_t mk = 0;
		if (tid < cnt) {
			// (round 6: requesting the NEXT batch's records before the walk of the current one, as composite_bwd does, needs 14 more
			// VGPRs at 72 of 72 -- 11 spilled at 7 waves / SIMD, or 6 waves: +3 % at C3, +10 % at C4, +4 % at C5 either way;
			// profiles/r06_composite_fwd_experiments.txt.  Seven workgroups per CU already overlap one another's gathers.)
			const uint32_t id = point_list[range.x + base + tid];
			const GsRec* rr = recs + id;
			const float4 a = rr->q0;
			float4 b = rr->q1;
			float4 c = rr->q2;
			mk = NOCULL ? allq : gs_quarter_mask<4>(a, b, tx0, ty0, allq);
			if (!NOCULL) qmask[range.x + base + tid] = (uint16_t)mk;
			c.w = b.z;
			b.z = __int_as_float((int)id);
			sRec[tid] = a;
			sRec[GSR_FWD_PLANE + tid] = b;
			sRec[2 * GSR_FWD_PLANE + tid] = c;
		}
		sMask[tid] = (uint16_t)mk;
		// this wave's four lists: all sentinels, then the hits of each quarter in list order
		{
			uint4* l4 = reinterpret_cast<uint4*>(&sList[w][0][0]);
			const uint32_t ss = GSR_FWD_SENT_OFF | (GSR_FWD_SENT_OFF << 16);
			for (int k = lane; k < 4 * GSR_FWD_LIST / 8; k += 64) l4[k] = make_uint4(ss, ss, ss, ss);
		}
		__syncthreads();
		int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
		const int wbit = 8 * (w >> 1) + 2 * (w & 1);   // bit of this wave's quarter 0; quarters 1, 2, 3 are bits +1, +4, +5
		for (int sub = 0; sub < cnt; sub += 64) {
			const uint32_t m16 = (uint32_t)sMask[sub + lane] >> wbit;
			const uint32_t off = (uint32_t)(sub + lane) << 4;
#define GSR_APPEND(QQ, SH, CNT)                                                                          \
	{                                                                                                    \
		const bool h = (m16 >> (SH)) & 1u;                                                               \
		const unsigned long long bm = __ballot(h);                                                       \
		const int pos = CNT + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)); \
		if (h) sList[w][QQ][pos] = (uint16_t)off;                                                        \
		CNT += __popcll(bm);                                                                             \
	}
			GSR_APPEND(0, 0, c0)
			GSR_APPEND(1, 1, c1)
			GSR_APPEND(2, 4, c2)
			GSR_APPEND(3, 5, c3)
#undef GSR_APPEND
		}
		__builtin_amdgcn_wave_barrier();
		const int n = max(max(c0, c1), max(c2, c3));
		uint32_t last_off = GSR_FWD_NONE, med_off = GSR_FWD_NONE;
		// four list entries per step, fetched one step ahead of their use; the scheduling barrier keeps the read up here
		// (left alone, the scheduler sinks it to its use and exposes a full LDS latency per step)
		uint2 pk = *reinterpret_cast<const uint2*>(my_list);
		for (int i = 0; i < n; i += 4) {
			if ((i & 31) == 0 && __ballot(!done) == 0ull) break;   // every pixel of this wave has saturated
			const uint32_t offs[4] = {pk.x & 0xffffu, pk.x >> 16, pk.y & 0xffffu, pk.y >> 16};
			pk = *reinterpret_cast<const uint2*>(my_list + i + 4);
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t off = offs[k];
				const float4 A = *reinterpret_cast<const float4*>(rec_base + off);
				const float4 B = *reinterpret_cast<const float4*>(rec_base + off + GSR_FWD_PLANE * 16);
				const float4 Cc = *reinterpret_cast<const float4*>(rec_base + off + 2 * GSR_FWD_PLANE * 16);
				const float dx = A.x - pixfx, dy = A.y - pixfy;
				const float power = FMA(A.w * dx, dy, FMA(B.x * dy, dy, (A.z * dx) * dx));
				const float alpha = fminf(0.99f, B.y * (FX ? gs_exp_hw(power) : gs_exp(power)));
				// FX: gs_exp_hw is valid for every power <= 0 and power < pcut implies alpha < 1/255 (preprocess): the pcut
				// pre-test is implied by the alpha test, as in composite_bwd's FX arm -- one compare and one LDS read less per step
				const bool valid = (!done) & (power <= 0.0f) & (FX ? true : (power >= (NOCULL ? -80.0f : B.w))) & (!(alpha < 1.0f / 255.0f));
				const float test_T = T_ * (1 - alpha);
				const bool stop = valid & (test_T < 0.0001f);
				done = done | stop;
				const bool apply = valid & (!stop);
				const float wgt = apply ? alpha * T_ : 0.f;
				acc01 = vfma(v2f{Cc.x, Cc.y}, v2f{wgt, wgt}, acc01);
				acc23 = vfma(v2f{Cc.z, Cc.w}, v2f{wgt, wgt}, acc23);
				const bool medc = apply & (T_ > 0.5f);
				med_off = medc ? off : med_off;
				med_T = medc ? T_ : med_T;
				T_ = apply ? test_T : T_;
				last_off = apply ? off : last_off;
			}
		}
		if (last_off != GSR_FWD_NONE) last_contributor = (uint32_t)base + (last_off >> 4) + 1u;
		if (med_off != GSR_FWD_NONE) med_pos = (uint32_t)base + (med_off >> 4) + 1u;
	}
	final_T[(size_t)tile * GSR_TILE_PIX + slot] = T_;
	n_contrib[(size_t)tile * GSR_TILE_PIX + slot] = last_contributor;
	uint32_t med_final = 0;
	if (inside) {
		float median_D = 15.0f, median_weight = 0.f;
		int median_id = 0;
		if (med_pos != 0) {
			const uint32_t id = point_list[range.x + med_pos - 1];
			const GsRec* rr = recs + id;
			const float4 A = rr->q0, B = rr->q1;
			const float dx = A.x - pixfx, dy = A.y - pixfy;
			const float power = FMA(A.w * dx, dy, FMA(B.x * dy, dy, (A.z * dx) * dx));
			const float alpha = fminf(0.99f, B.y * (FX ? gs_exp_hw(power) : gs_exp(power)));
			if (med_T * (1 - alpha) < 0.5f) {
				median_D = B.z;
				median_weight = alpha * med_T;
				median_id = (int)id;
				med_final = med_pos;
			}
		}
		const size_t HW = (size_t)H * W;
		const size_t pix_id = (size_t)W * py + px;
		out_color[pix_id] = acc01.x;
		out_color[HW + pix_id] = acc01.y;
		out_color[2 * HW + pix_id] = acc23.x;
		out_depth[pix_id] = acc23.y;
		out_median[pix_id] = median_D;
		out_median[HW + pix_id] = median_weight;
		out_median[2 * HW + pix_id] = (float)median_id;
		out_opacity[pix_id] = 1 - T_;
	}
	med_pos_out[(size_t)tile * GSR_TILE_PIX + slot] = med_final;
	if (tid == 0 && staged_out != nullptr) staged_out[tile] = (uint32_t)staged;
}

// Longest-first tile order for composite_fwd on skewed frames (see launch_tile_order in gsr_kernels_bwd.hip for the
// backward's, which knows the true walk lengths): tiles by descending list-length class, one small workgroup.
__device__ __forceinline__ uint32_t gs_len_class(uint32_t m)   // 4 classes per octave, monotone; < 128
{
	if (m < 4u) return m;
	const int e = 31 - __clz((int)m);
	return (uint32_t)(4 * (e - 1)) + ((m >> (e - 2)) & 3u);
}
__global__ __launch_bounds__(1024) void tile_order_fwd_kernel(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ order,
                                                             const GsCtl* __restrict__ ctl, uint32_t cap)
{
	__shared__ uint32_t s_cnt[128], s_cur[128];
	if (ctl->num_binned > cap) return;
	const int tid = threadIdx.x;
	if (tid < 128) s_cnt[tid] = 0u;
	__syncthreads();
	for (int t = tid; t < T; t += 1024) atomicAdd(&s_cnt[gs_len_class(ranges[t].y - ranges[t].x)], 1u);
	__syncthreads();
	if (tid == 0) {
		uint32_t run = 0;
		for (int c = 127; c >= 0; c--) { s_cur[c] = run; run += s_cnt[c]; }
	}
	__syncthreads();
	for (int t = tid; t < T; t += 1024) order[atomicAdd(&s_cur[gs_len_class(ranges[t].y - ranges[t].x)], 1u)] = (uint32_t)t;
}
void launch_tile_order_fwd(int T, const uint2* ranges, uint32_t* order, const GsCtl* ctl, uint32_t cap, hipStream_t s)
{
	hipLaunchKernelGGL(tile_order_fwd_kernel, dim3(1), dim3(1024), 0, s, T, ranges, order, ctl, cap);
}

void launch_composite_fwd(const ImgLayout& il, int W, int H, const uint2* ranges, const uint32_t* point_list,
                          const GsRec* recs, float* out_color, float* out_depth, float* out_median,
                          float* out_opacity, float* final_T, uint32_t* n_contrib, uint32_t* med_pos,
                          GsCtl* ctl_, uint32_t cap, uint32_t max_sorted, bool nocull, bool wave_lists, bool fast_exp,
                          const uint32_t* tile_order, uint32_t* staged_out, hipStream_t s)
{
	const int chunk = (il.T + 7) / 8;
#define GSR_LAUNCH_FWD(K)                                                                                              \
	hipLaunchKernelGGL(K, dim3(chunk * 8), dim3(256), 0, s, il.T, chunk, il.gx, W, H, ranges, point_list, recs, out_color, \
	                   out_depth, out_median, out_opacity, final_T, n_contrib, med_pos, ctl, cap, max_sorted, tile_order, staged_out)
#ifdef GSR_AB_VARIANTS
	if (wave_lists) {
		const GsCtl* ctl = ctl_;
		if (nocull) GSR_LAUNCH_FWD(composite_fwd_kernel<true>);
		else GSR_LAUNCH_FWD(composite_fwd_kernel<false>);
	} else
#endif
	{   // (a build without GSR_AB_VARIANTS refuses fwd_variant 1 in gsr_forward: wave_lists is false here)
		GsCtl* ctl = ctl_;
		if (fast_exp) {
			if (nocull) GSR_LAUNCH_FWD((composite_fwd_quarter_kernel<true, true>));
			else GSR_LAUNCH_FWD((composite_fwd_quarter_kernel<false, true>));
		} else {
			if (nocull) GSR_LAUNCH_FWD((composite_fwd_quarter_kernel<true, false>));
			else GSR_LAUNCH_FWD((composite_fwd_quarter_kernel<false, false>));
		}
	}
#undef GSR_LAUNCH_FWD
}

}  // namespace gsr
