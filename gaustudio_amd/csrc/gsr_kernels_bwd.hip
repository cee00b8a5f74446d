// gsr_kernels_bwd.hip -- backward kernels of libgsrast for gfx950 (MI355X, wave64).
//
//   composite_bwd    one workgroup of four wave64 per tile, one pixel per lane, back to front; every 16-lane quarter
//                    of a wave (a 4x4 pixel block) walks its own list of the staged instances; two per-pixel
//                    scalars per (pixel, instance) go through an LDS slab and are turned into the ten per-Gaussian
//                    sums by lanes re-mapped to (quarter, list step, pixel row) -- a transposition instead of a
//                    cross-lane reduction tree; each (tile, Gaussian) instance leaves one plain-stored 48-B row in
//                    Gaussian-major order -- no global atomics
//                    (replaces backward.cu:415-610 renderCUDA: 11-12 atomicAdd per (pixel, Gaussian))
//   (row offsets goff: per-block sums in preprocess_fwd + tile_scan block 1 + goff_apply, gsr_kernels_fwd.hip)
//   preprocess_bwd   one lane per Gaussian: fixed-order sum of its rows (fetched wave-cooperatively through LDS),
//                    dL/dconic -> dL/dcov3D, dL/dmean (projection + depth), cov3D -> scale / raw-quaternion
//                    backward, activation chain rules of the raw interface
//                    (replaces backward.cu:144-274 computeCov2DCUDA + :346-412 preprocessCUDA)
//   preprocess_bwd_sh  SH backward (dL/dSH, dL/dmean through the view direction), rows staged through LDS
//                    (replaces backward.cu:20-139 computeColorFromSH backward)
#include "gsr_internal.h"
#include "gsr_bwd_timing.h"

namespace gsr {

__device__ __constant__ float bSH_C0 = 0.28209479177387814f;
__device__ __constant__ float bSH_C1 = 0.4886025119029199f;
__device__ __constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                            -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                            0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                            -0.5900435899266435f};

// Lane-swap folds used by phase 2 of composite_bwd (lanes that belong to the same pair sit 8, 16 and 32 lanes apart):
//   half_swap_sum(a, b): lanes 0-31 get a[l] + a[l+32], lanes 32-63 get b[l-32] + b[l]   (1 swap + 1 add for two
//   quantities); row_swap_sum(p, q): 16-lane rows (0,1,2,3) get p.r0+p.r1, q.r0+q.r1, p.r2+p.r3, q.r2+q.r3.
// Doing these "transposing" steps on pairs of quantities shrinks ten per-lane quantities to three registers whose
// rows carry different quantities; GSR_DPP_ADD(v, ctrl) is a fused v_add_f32_dpp.
#define GSR_DPP_ADD(v, ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
#define GSR_DPP_OF(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float half_swap_sum(float a, float b)
{
	// v_permlane32_swap(X, Y): X.rows23 <-> Y.rows01
	const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float row_swap_sum(float p, float q)
{
	// v_permlane16_swap(X, Y): X.row1 <-> Y.row0, X.row3 <-> Y.row2
	const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
	return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// Gaussians with MORE than GSR_SUM_LONG instance rows (a splat tens of pixels wide sits in hundreds of tile lists) are not
// summed by their own lane -- one lane adding 400 rows while 63 wait made preprocess_bwd 4x slower on a scene with 1 % of
// such splats (bench.py --workload C2-clustered: 0.199 ms against 0.05) -- but by the whole WAVE: lane l adds the rows
// b + l, b + l + 64, ... (coalesced 3-KiB reads), and six xor-butterfly steps add the 64 partial sums.  A fixed order
// again (the butterfly is symmetric: every lane ends with the same bits), used by preprocess_bwd and by the inspection
// kernel alike, so the result stays independent of scheduling and the two agree bit for bit.
#define GSR_SUM_LONG 96
__device__ __forceinline__ void gs_sum_rows_wave(uint32_t b, uint32_t e, const float* __restrict__ rows,
                                                 const uint8_t* __restrict__ row_flags, int lane, float* acc)
{
#pragma unroll
	for (int i = 0; i < 10; i++) acc[i] = 0.f;
	for (uint32_t r = b + (uint32_t)lane; r < e; r += 64) {
		if (row_flags != nullptr && !row_flags[r]) continue;
		const float4* ar = reinterpret_cast<const float4*>(rows + (size_t)r * GSR_ROW_STRIDE);
		const float4 v0 = gs_ld_stream(ar), v1 = gs_ld_stream(ar + 1), v2 = gs_ld_stream(ar + 2);
		acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w; acc[4] += v1.x; acc[5] += v1.y;
		acc[6] += v1.z; acc[7] += v1.w; acc[8] += v2.x; acc[9] += v2.y;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
		for (int i = 0; i < 10; i++) acc[i] += __shfl_xor(acc[i], o, 64);
	}
}
// the long Gaussians of a wave, one after the other (all 64 lanes must call; `b`, `e` = the lane's own row span or b == e)
__device__ __forceinline__ void gs_sum_long_rows(uint32_t b, uint32_t e, const float* __restrict__ rows,
                                                 const uint8_t* __restrict__ row_flags, int lane, float* a_)
{
	unsigned long long longs = __ballot(e - b > (uint32_t)GSR_SUM_LONG);
	while (longs) {
		const int l = __builtin_ctzll(longs);
		longs &= longs - 1;
		const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)b, l), el = (uint32_t)__builtin_amdgcn_readlane((int)e, l);
		float acc[10];
		gs_sum_rows_wave(bl, el, rows, row_flags, lane, acc);
		if (lane == l) {
#pragma unroll
			for (int i = 0; i < 10; i++) a_[i] = acc[i];
		}
	}
}

// Sum of the per-instance rows of Gaussian idx in ascending tile order (fixed order: the result does
// not depend on scheduling); spans longer than GSR_SUM_LONG by the whole wave (above).  Test-only inspection kernel;
// preprocess_bwd forms the same sums in the same order from its LDS slabs.
__global__ __launch_bounds__(256) void inspect_sums_kernel(int P, const int* __restrict__ radii,
                                                           const GsRec* __restrict__ recs,
                                                           const uint32_t* __restrict__ goff,
                                                           const float* __restrict__ rows,
                                                           const uint8_t* __restrict__ row_flags, float* __restrict__ out)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const int lane = threadIdx.x & 63;
	const bool vis = idx < P && radii[idx] > 0;
	float a_[GSR_ROW_STRIDE];
#pragma unroll
	for (int i = 0; i < GSR_ROW_STRIDE; i++) a_[i] = 0.f;
	const uint32_t b = vis ? goff[idx] : 0u, e = vis ? goff[idx + 1] : 0u;
	if (e - b <= (uint32_t)GSR_SUM_LONG) {
		for (uint32_t r = b; r < e; r++) {
			if (row_flags != nullptr && !row_flags[r]) continue;   // flags mode: not written (never reached by the walk)
			const float4* ar = reinterpret_cast<const float4*>(rows + (size_t)r * GSR_ROW_STRIDE);
			const float4 v0 = ar[0], v1 = ar[1], v2 = ar[2];
			a_[0] += v0.x; a_[1] += v0.y; a_[2] += v0.z; a_[3] += v0.w; a_[4] += v1.x; a_[5] += v1.y;
			a_[6] += v1.z; a_[7] += v1.w; a_[8] += v2.x; a_[9] += v2.y;
		}
	}
	gs_sum_long_rows(b, e, rows, row_flags, lane, a_);
	if (idx >= P) return;
#pragma unroll
	for (int i = 0; i < 10; i++) out[10 * (size_t)idx + i] = a_[i];
}

void launch_inspect_sums(int P, const int* radii, const GsRec* recs, const uint32_t* goff, const float* rows,
                         const uint8_t* row_flags, float* sums10, hipStream_t s)
{
	hipLaunchKernelGGL(inspect_sums_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, recs, goff, rows, row_flags,
	                   sums10);
}

// ------------------------------------------------------------------------------------------------
// composite_bwd: ONE workgroup of 4 wave64 per 16x16 tile; wave w owns the 8x8 pixel block w of the tile, one pixel
// per lane -- the forward's layout (gs_pixel_of_thread), walked back to front.
//
// The expensive part of this kernel used to be the cross-lane reduction: every (wave, instance) pair has to turn 64
// per-pixel contributions into ten per-Gaussian sums, and a lane-swap / DPP tree for ten quantities costs ~130 VALU
// cycles per pair however it is arranged (rounds 1-2: 45 % of the kernel).  It is now done by TRANSPOSITION
// through LDS instead:
//   phase 1 (lane = pixel): per pair only the two per-pixel scalars everything else is linear in are computed,
//           q = G * dL_dalpha   and   w = alpha * T_before,
//           and stored as one 8-byte LDS write per lane into the pair's row of the wave's slab;
//   phase 2 (lane = (pair u, pixel row g), every GSR_BWD_UNITS pairs): each lane walks the 8 pixels of ITS row of
//           ITS pair and accumulates the ten sums with plain FMAs against per-pixel constants (upstream gradients
//           from a small LDS table, dx / dy recomputed) -- 64 pixels x 10 FMAs = 10 wave-instructions per pair
//           instead of a reduction tree; three swap / DPP steps fold the 8 rows (lanes of the same pair only).
// No LDS float atomics, and the per-pixel products (q*dx, w*dL_dpixel, ...) leave phase 1 as well.  The four waves
// write their sums into four private planes that the flush adds in a fixed order: gradients stay bit-reproducible
// from run to run.
// The median-depth gradient (backward.cu:566-569) goes to the Gaussian the FORWARD recorded as the pixel's median
// (med_pos, same decision as forward.cu:368-373 on the forward's own transmittances): one integer compare per pixel
// in phase 2.  The reference re-derives the crossing from a transmittance reconstructed by division, which is
// ill-conditioned at the threshold; the two can only disagree inside the window tests/ grant as `flip9`.
// Staging is software-pipelined: the records of the NEXT batch (and the ids of the one after) are requested before
// the walk of the current one, four threads per 64-B record, so the dependent point_list -> record gathers are not
// waited for between batches.
// ------------------------------------------------------------------------------------------------
#define GSR_BWD_THREADS 256
#define GSR_BWD_BATCH 64      // instances staged per round: one lane-test per (wave, instance) in a single ballot
#ifndef GSR_BWD_UNITS
#define GSR_BWD_UNITS 8       // pairs per transposed reduction; lane = (pair u = lane % UNITS, pixel group g = lane / UNITS)
#endif
#define GSR_SLAB_STRIDE 65    // float2 per pair row: 64 pixels + 1 pad (conflict-free b64 reads across pairs)
#define GSR_PLANE_STRIDE 10   // floats per instance in a wave's plane of sums
#define GSR_TAB_ROW 17        // float4 per pixel row of the constants table: 8 pixels x 2 + 1 pad (rows on distinct banks)

#ifdef GSR_AB_VARIANTS   // the per-wave (8x8) walk: superseded by the per-quarter kernel below, kept for A/B builds only (make AB=1)
// Phase 2: the wave's slab holds (q, w) of `n` pairs; writes their ten sums into the wave's plane.
//   sums: 0 M10 = sum q dx, 1 M01 = sum q dy, 2 M20, 3 M11, 4 M02, 5 sum (w dL_dopacity + q), 6..8 sum w dL_dpixel[c],
//         9 sum w dL_ddepth + (median gradient of the pixels whose median this Gaussian is)
// tab: per pixel (sx, row) two float4 at [row * GSR_TAB_ROW + 2 sx]: {dL_dpixel rgb, dL_ddepth}, {dL_dopacity,
// dL_dmedian, med_pos bits, -}.
__device__ __forceinline__ void gs_bwd_phase2(const int n, const float2* __restrict__ slab, const unsigned long long unit_js,
                                              const float4* __restrict__ sA, const float4* __restrict__ tab, const float fbx,
                                              const float fby, const int top, float* __restrict__ plane, const int lane)
{
	constexpr int NG = 64 / GSR_BWD_UNITS;      // pixel groups: 8 (one pixel row each) or 16 (half a row each)
	constexpr int PPL = GSR_BWD_UNITS;          // pixels per lane
	const int u = lane & (GSR_BWD_UNITS - 1), g = lane / GSR_BWD_UNITS;
	const bool act = u < n;
	// batch indices of the slab's pairs, 8 bits each, packed by phase 1 with scalar instructions (wave-uniform)
	const int j = act ? (int)((unit_js >> (8 * u)) & 0xffull) : 0;
	const float2 ctr = *reinterpret_cast<const float2*>(&sA[j]);   // Gaussian centre
	const uint32_t pos1 = (uint32_t)(top - j);   // list position + 1 of the pair's instance
	const int prow = NG == 8 ? g : (g >> 1), pcol0 = NG == 8 ? 0 : ((g & 1) << 2);   // first pixel of the lane's group
	const float dy = ctr.y - (fby + (float)prow);
	const float X = ctr.x - (fbx + (float)pcol0);          // dx of the lane's first pixel; dx of pixel sx is X - sx
	const float2* row = slab + u * GSR_SLAB_STRIDE + prow * 8 + pcol0;
	const float4* trow = tab + prow * GSR_TAB_ROW + 2 * pcol0;
	// All pixels of a lane share dy, so dy factors out of M01, M11, M02 (exactly up to one rounding each, relative to
	// the magnitude of the factored sum) and only sum q, sum q dx, sum q dx^2 are accumulated per pixel.
	float S0 = 0.f, a0 = 0.f, a2 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f, a8 = 0.f, a9 = 0.f;
#pragma unroll
	for (int sx = 0; sx < PPL; sx++) {
		const float2 v = row[sx];                           // (q, w) of the pixel
		const float4 c0 = trow[2 * sx], c1 = trow[2 * sx + 1];
		const float dx = X - (float)sx;
		const float t = v.x * dx;
		S0 += v.x;
		a0 += t;
		a2 = FMA(t, dx, a2);
		a5 = FMA(v.y, c1.x, a5);                             // backward.cu:575 (+ :607 below: + sum q)
		a6 = FMA(v.y, c0.x, a6);
		a7 = FMA(v.y, c0.y, a7);
		a8 = FMA(v.y, c0.z, a8);
		a9 = FMA(v.y, c0.w, a9) + ((__float_as_uint(c1.z) == pos1) ? c1.y : 0.f);
	}
	a5 += S0;
	const float a1 = dy * S0, a3 = dy * a0, a4 = dy * a1;
	// fold the pixel groups of every pair: lanes u + UNITS * g (partners always belong to the same pair, so stale
	// slab rows of inactive pairs never leak into active ones)
	const float r0 = half_swap_sum(a0, a1);
	const float r1 = half_swap_sum(a2, a3);
	const float r2 = half_swap_sum(a4, a5);
	const float r3 = half_swap_sum(a6, a7);
	const float r4 = half_swap_sum(a8, a9);
	float s0 = row_swap_sum(r0, r1), s1 = row_swap_sum(r2, r3), s2 = row_swap_sum(r4, 0.f);
	GSR_DPP_ADD(s0, 0x128); GSR_DPP_ADD(s1, 0x128); GSR_DPP_ADD(s2, 0x128);   // row_ror:8: lanes l and l ^ 8
	// UNITS == 4: lanes u + 4 k of a row still differ; after the previous step the values are symmetric under l ^ 8,
	// so (l + 4) % 16 is as good as l ^ 4
	if (NG == 16) { GSR_DPP_ADD(s0, 0x124); GSR_DPP_ADD(s1, 0x124); GSR_DPP_ADD(s2, 0x124); }   // row_ror:4
	// 16-lane row r of s0 / s1 / s2 now holds component {0,2,1,3}[r] / 4+{0,2,1,3}[r] / {8,-,9,-}[r] of pair u
	if (act && (lane & (16 - GSR_BWD_UNITS) & 15) == 0) {
		const int rw = lane >> 4;
		float* dst = plane + j * GSR_PLANE_STRIDE + (((rw & 1) << 1) | (rw >> 1));
		dst[0] = s0;
		dst[4] = s1;
		if ((rw & 1) == 0) dst[8] = s2;
	}
}

// FLAGS: long-list regime (per-row validity bytes); compile-time so that the short-list kernel carries none of it.
// TSEL: keep a select on T for devices where v_rcp_f32(1.0) != 1.0 (gsr_selftest).
template <bool FLAGS, bool TSEL>
__global__ __launch_bounds__(GSR_BWD_THREADS) void composite_bwd_kernel(
    int T, int chunk, int gx, int W, int H, const GsBg bgv, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const GsRec* __restrict__ recs, const uint32_t* __restrict__ goff,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ med_pos,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth, const float* __restrict__ dL_dpix_median,
    const float* __restrict__ dL_dpix_opacity, float* __restrict__ rows, uint8_t* __restrict__ row_flags,
    const GsCtl* __restrict__ ctl)
{
	__shared__ float4 sA[GSR_BWD_BATCH];   // q0: px, py, -a/2, -b
	__shared__ float4 sB[GSR_BWD_BATCH];   // q1: -c/2, opacity, depth, pcut
	__shared__ float4 sC[GSR_BWD_BATCH];   // q2: r, g, b, -
	__shared__ uint32_t s_row[GSR_BWD_BATCH];   // Gaussian-major row of each staged instance
	__shared__ __attribute__((aligned(16))) float s_plane[4][GSR_BWD_BATCH * GSR_PLANE_STRIDE];   // per-wave sums of the batch
	__shared__ float2 s_slab[4][GSR_BWD_UNITS * GSR_SLAB_STRIDE];
	__shared__ float4 s_tab[4][8 * GSR_TAB_ROW];            // per-pixel constants of phase 2
	__shared__ int s_max[4];
	if (blockIdx.x == 0 && threadIdx.x == 0 && bgv.flag_dst != nullptr) *bgv.flag_dst = bgv.flag;   // regime word (GsBg)
	// XCD-banded static order, or (skewed frames) longest walk first: launch_tile_order
	// (a banded backward, GSR_BWD_PART_BAND_*, walks the tiles [tile_lo, tile_hi) only: the XCD bands are cut out of THAT range, so
	// that a band still runs on all eight XCDs -- cut out of the whole image, the upper half of the tiles lives on four of them)
	const int tb_lo = (int)bgv.tile_lo, tb_n = (int)min(bgv.tile_hi, (uint32_t)T) - tb_lo;
	const int local = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	const int tile = bgv.tile_order ? ((int)blockIdx.x < T ? (int)bgv.tile_order[blockIdx.x] : T) : tb_lo + local;
	if ((!bgv.tile_order && ((int)(blockIdx.x >> 3) >= chunk || local >= tb_n)) || tile >= T) return;
	if ((uint32_t)tile < bgv.tile_lo || (uint32_t)tile >= bgv.tile_hi) return;   // (longest-first order of a skewed frame: every tile is offered to both bands)
	const int tid = threadIdx.x;
	const int lane = tid & 63, wv = tid >> 6;
	const int tx = tile % gx, ty = tile / gx;
	int lx, ly;
	gs_pixel_of_thread(tid, lx, ly);
	const int px = tx * GSR_BLOCK_X + lx, py = ty * GSR_BLOCK_Y + ly;
	const bool inside = px < W && py < H;
	const float pixfx = (float)px, pixfy = (float)py;
	// this wave's pixel block (pixel centres), clipped to the image
	const float fbx = (float)(tx * GSR_BLOCK_X + ((wv & 1) << 3)), fby = (float)(ty * GSR_BLOCK_Y + ((wv >> 1) << 3));
	const float bx1 = fminf(fbx + 7.f, (float)(W - 1)), by1 = fminf(fby + 7.f, (float)(H - 1));
	const uint2 range = ranges[tile];

	const size_t sidx = (size_t)tile * GSR_TILE_PIX + tid;   // the forward's tile-major pixel state
	const float T_final = inside ? final_T[sidx] : 0.f;
	const int lc = inside ? (int)n_contrib[sidx] : 0;          // last_contributor
	float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f, dLo = 0.f;
	{
		float dLm = 0.f;
		uint32_t mpos = 0u;
		if (inside) {
			const size_t HW = (size_t)H * W;
			const size_t pix_id = (size_t)W * py + px;
			// an upstream gradient the caller did not supply (NULL: an output the loss does not use) is zero: not loaded
			if (dL_dpix) {
				dLp0 = dL_dpix[pix_id];
				dLp1 = dL_dpix[HW + pix_id];
				dLp2 = dL_dpix[2 * HW + pix_id];
			}
			if (dL_dpix_depth) dLd = dL_dpix_depth[pix_id];
			if (dL_dpix_median) dLm = dL_dpix_median[pix_id];   // channel 0 only (backward.cu:481-482)
			if (dL_dpix_opacity) dLo = dL_dpix_opacity[pix_id];
			mpos = med_pos[sidx];
		}
		// lane = pixel (lx & 7, ly & 7) of the wave's block
		float4* t = s_tab[wv] + (lane >> 3) * GSR_TAB_ROW + 2 * (lane & 7);
		t[0] = make_float4(dLp0, dLp1, dLp2, dLd);
		t[1] = make_float4(dLo, dLm, __uint_as_float(mpos), 0.f);
	}
	// bg . dL_dpixel (backward.cu:584-586), loop invariant
	const float bg0 = bgv.dptr ? bgv.dptr[0] : bgv.host[0], bg1 = bgv.dptr ? bgv.dptr[1] : bgv.host[1], bg2 = bgv.dptr ? bgv.dptr[2] : bgv.host[2];
	const float bg_dot = FMA(bg2, dLp2, FMA(bg1, dLp1, FMA(bg0, dLp0, 0.f)));
	// wave-uniform: with a black background (the common case) the term is skipped by a scalar branch
	const bool any_bg = __ballot(bg_dot != 0.f) != 0ull;
	float T_ = T_final;
	// The reference carries five back-to-front recurrences accum_rec[ch] (3 colours, depth, opacity;
	// backward.cu:541-573) but dL_dalpha only needs their dot product with this pixel's upstream gradients, and the
	// recurrence is linear: one scalar S = <accum_rec, dL_dpixel> is carried, updated EAGERLY with this Gaussian's
	// alpha (the reference folds last_alpha / last_color in at the next contributor: the same fma one step later).
	float S = 0.f;

	// block / tile maxima of last_contributor: list entries at or beyond them are dead for every pixel
	int wmax = lc;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
	wmax = __builtin_amdgcn_readfirstlane(wmax);
	if (lane == 0) s_max[wv] = wmax;
	__syncthreads();
	const int bmax = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

	// List entries at or beyond bmax contribute to no pixel of this tile.  Two regimes, chosen by the launcher from
	// the average list length:
	//   short lists (most of a list is walked): their rows are zeroed here, under the shadow of the main loop, and
	//     preprocess_bwd adds every row unconditionally;
	//   long lists (real scenes: a few thousand entries of which ~15 % are reached): row_flags != nullptr -- written
	//     rows set a validity byte, unreached entries are left alone and preprocess_bwd skips them WITHOUT reading
	//     them (clearing + re-reading 48 B per unreached entry was the larger cost: C4 share 1.42 -> 0.59 ms here).
	for (int i = bmax + tid; !FLAGS && i < (int)(range.y - range.x); i += GSR_BWD_THREADS) {
		const uint32_t id = point_list[range.x + i];
		const uint4 q3 = recs[id].q3;
		float4* dst = reinterpret_cast<float4*>(rows + (size_t)(goff[id] + gs_row_in_rect(q3.x, q3.y, (q3.z >> GSR_Q3Z_DEAD_SHIFT) & 15u, tx, ty)) * GSR_ROW_STRIDE);
		dst[0] = make_float4(0.f, 0.f, 0.f, 0.f);
		dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
		dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
	}
	float2* slab = s_slab[wv];
	float* plane = s_plane[wv];
	// ---- software-pipelined staging: thread t fetches 16-B part (t & 3) of the record of staged instance t >> 2 ----
	const int srec = tid >> 2, spart = tid & 3;
	auto load_id = [&](int t) -> uint32_t {   // id of list position t-1-srec (0 when outside the walk)
		return (t > 0 && srec < min(GSR_BWD_BATCH, t)) ? point_list[range.x + (uint32_t)(t - 1 - srec)] : 0u;
	};
	// part 3 (q3: tile rect) also fetches the Gaussian's first row goff[id] into the unused .w
	auto load_part = [&](int t, uint32_t id) -> float4 {
		float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
		if (t > 0 && srec < min(GSR_BWD_BATCH, t)) {
			v = reinterpret_cast<const float4*>(recs + id)[spart];
			if (spart == 3) v.w = __uint_as_float(goff[id]);
		}
		return v;
	};
	uint32_t id_next = load_id(bmax - GSR_BWD_BATCH);
	float4 part_cur = load_part(bmax, load_id(bmax));
	// walk positions pos = bmax-1 ... 0, staged GSR_BWD_BATCH at a time
	for (int top = bmax; top > 0; top -= GSR_BWD_BATCH) {
		const int cnt = min(GSR_BWD_BATCH, top);
		__syncthreads();   // the previous flush has read sA / sB / the planes
		if (srec < cnt) {
			if (spart == 0) sA[srec] = part_cur;
			else if (spart == 1) sB[srec] = part_cur;
			else if (spart == 2) sC[srec] = part_cur;
			else {
				// Gaussian-major row of this (tile, Gaussian) instance: goff[g] (fetched with the record) + raster index
				// of the tile inside the Gaussian's tile rect
				const uint32_t q3x = __float_as_uint(part_cur.x), q3y = __float_as_uint(part_cur.y), q3w = __float_as_uint(part_cur.w);
				s_row[srec] = q3w + gs_row_in_rect(q3x, q3y, (__float_as_uint(part_cur.z) >> GSR_Q3Z_DEAD_SHIFT) & 15u, tx, ty);
			}
		}
		// request the next batch's records and the ids of the one after: in flight during this batch's walk
		const float4 part_next = load_part(top - GSR_BWD_BATCH, id_next);
		id_next = load_id(top - 2 * GSR_BWD_BATCH);
		// zero the four planes of sums (4 x 64 x 10 floats = 640 float4)
		for (int i = tid; i < 4 * GSR_BWD_BATCH * GSR_PLANE_STRIDE / 4; i += GSR_BWD_THREADS)
			reinterpret_cast<float4*>(&s_plane[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
		__syncthreads();
		// per-wave cull of the staged batch: lane l tests instance l against the wave's pixel block (see composite_fwd)
		bool hit = false;
		if (lane < cnt && top - 1 - lane < wmax) hit = gs_box_may_touch(sA[lane], sB[lane], fbx, fby, bx1, by1);
		unsigned long long m = __ballot(hit);
		int nu = 0;        // pairs in the slab (wave-uniform)
		unsigned long long unit_js = 0ull;   // batch indices of the slab's pairs, 8 bits each (wave-uniform: SGPRs)
		// one (wave, instance) pair; returns with the pair's (q, w) in the slab, or without a trace if no pixel is live
		auto pair = [&](const float4 A, const float4 B, const float4 Cc, const int j) {
			const int pos = top - 1 - j;   // == `contributor` after decrement (backward.cu:520)
			const float dx = A.x - pixfx, dy = A.y - pixfy;
			const float power = FMA(A.w * dx, dy, FMA(B.x * dy, dy, (A.z * dx) * dx));
			const float G = gs_exp(power);
			const float a0 = B.y * G;
			// no short-circuit evaluation: `&` keeps the body free of exec-mask branches
			const bool live = (pos < lc) & (power <= 0.0f) & (power >= B.w) & (!(a0 < 1.0f / 255.0f));
			if (__ballot(live) == 0ull) return;
			// a dead pixel is carried through with G masked to 0, which makes everything derived from it exactly
			// neutral: alpha = 0, 1/(1-alpha) = 1 (v_rcp_f32(1.0) == 1.0, gsr_selftest), w = 0, q = 0, S <- fma(0,.,S)
			const float Gm = live ? G : 0.f;
			const float alpha = fminf(0.99f, B.y * Gm);
			// 1/(1-alpha) once, by v_rcp_f32 (1 ulp) instead of two IEEE divisions (backward.cu:536,587): the backward
			// is tolerance-checked (its sums are order-dependent in the reference as well)
			const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
			const float test_T = T_ * rinv;
			const float w = alpha * test_T;   // dchannel_dcolor = dpixel_depth_ddepth = dpixel_opacity_dopacity
			// <colour of this Gaussian, dL_dpixel> over the 5 blended channels (rgb, depth, opacity == 1)
			const float cd = FMA(Cc.x, dLp0, FMA(Cc.y, dLp1, FMA(Cc.z, dLp2, FMA(B.z, dLd, dLo))));
			const float diff = cd - S;
			float dL_dalpha = diff * test_T;
			if (any_bg) {                                                         // backward.cu:584-587
				asm volatile("");   // not speculated: keeps this a scalar branch instead of compute-always + select
				dL_dalpha = FMA(-(T_final * rinv), bg_dot, dL_dalpha);
			}
			const float q = Gm * dL_dalpha;                                       // dL_dG * G / opacity
			S = FMA(alpha, diff, S);
			T_ = TSEL ? (live ? test_T : T_) : test_T;
			slab[nu * GSR_SLAB_STRIDE + lane] = make_float2(q, w);
			unit_js |= (unsigned long long)j << (8 * nu);
			nu++;
			if (nu == GSR_BWD_UNITS) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				gs_bwd_phase2(GSR_BWD_UNITS, slab, unit_js, sA, s_tab[wv], fbx, fby, top, plane, lane);
				__builtin_amdgcn_wave_barrier();
				nu = 0;
				unit_js = 0ull;
			}
		};
		// the record of the NEXT surviving instance is read (wave-uniform LDS reads) while the current pair computes:
		// two copies of the body alternate between two register sets instead of moving 12 registers per pair
		if (m) {
			int j0 = __ffsll((long long)m) - 1, j1 = 0;
			m &= m - 1;
			float4 A0 = sA[j0], B0 = sB[j0], C0 = sC[j0], A1, B1, C1;
			while (true) {
				const bool more1 = m != 0ull;
				if (more1) {
					j1 = __ffsll((long long)m) - 1;
					m &= m - 1;
					A1 = sA[j1]; B1 = sB[j1]; C1 = sC[j1];
				}
				pair(A0, B0, C0, j0);
				if (!more1) break;
				const bool more0 = m != 0ull;
				if (more0) {
					j0 = __ffsll((long long)m) - 1;
					m &= m - 1;
					A0 = sA[j0]; B0 = sB[j0]; C0 = sC[j0];
				}
				pair(A1, B1, C1, j1);
				if (!more0) break;
			}
		}
		if (nu > 0) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			gs_bwd_phase2(nu, slab, unit_js, sA, s_tab[wv], fbx, fby, top, plane, lane);
		}
		// flush: one thread per staged instance adds the four planes in fixed order and stores the 48-B row (plain
		// stores, no global atomics), zeros included: every row of the scratch is written exactly once per backward,
		// so nobody has to clear it
		__syncthreads();
		if (tid < cnt) {
			float v[10];
#pragma unroll
			for (int k = 0; k < 10; k++)
				v[k] = ((s_plane[0][tid * GSR_PLANE_STRIDE + k] + s_plane[1][tid * GSR_PLANE_STRIDE + k]) +
				        s_plane[2][tid * GSR_PLANE_STRIDE + k]) + s_plane[3][tid * GSR_PLANE_STRIDE + k];
			// moments -> gradients (once per (tile, Gaussian)): with q = G*dL_dalpha summed over pixels,
			//   dL_dmean2D.x = -0.5W * op * (a*M10 + b*M01)      (backward.cu:593-599)
			//   dL_dconic    = -0.5 * op * (M20, M11, M02)       (backward.cu:602-604)
			// where conic (a, b, c) = (-2*q0.z, -q0.w, -2*q1.x), op = q1.y
			const float4 A = sA[tid], B = sB[tid];
			const float ca = -2.f * A.z, cb = -A.w, cc = -2.f * B.x, op = B.y;
			const float M10 = v[0], M01 = v[1], M20 = v[2], M11 = v[3], M02 = v[4];
			const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);   // backward.cu:493-494
			const uint32_t my_row = s_row[tid];
			float4* dst = reinterpret_cast<float4*>(rows + (size_t)my_row * GSR_ROW_STRIDE);
			dst[0] = make_float4(-(op * FMA(cb, M01, ca * M10)) * ddelx_dx, -(op * FMA(cb, M10, cc * M01)) * ddely_dy,
			                     -0.5f * op * M20, -0.5f * op * M11);
			dst[1] = make_float4(-0.5f * op * M02, v[5], v[6], v[7]);
			dst[2] = make_float4(v[8], v[9], 0.f, 0.f);
			if (FLAGS) row_flags[my_row] = 1;
		}
		part_cur = part_next;
	}
}

#endif   // GSR_AB_VARIANTS

// ------------------------------------------------------------------------------------------------
// composite_bwd with per-quarter instance lists (the default; the kernel above is kept as variant bit 1 for A/B).
// Same transposition idea, finer granularity -- the forward's (composite_fwd_quarter_kernel): every 16-lane quarter
// of a wave owns a 4x4 pixel block and walks its OWN list of the staged instances, compacted from a per-instance
// 4-bit mask (gs_quarter_mask<2> against the wave's four blocks, cut by each quarter's own last_contributor bound).
//   phase 1 (lane = pixel): list step i of all four quarters in one instruction stream; (q, w) go to the slab row of
//           (quarter, step);
//   phase 2 (every GSR_BWQ_U steps; lane = (quarter, step u, pixel row r)): the lane accumulates the 4 pixels of its
//           row -- their upstream gradients live in REGISTERS (20 VGPRs; with 8 pixels per lane they had to be
//           re-read from an LDS table for every pair: 80 % of phase 2's LDS traffic), folds the 4 rows with two DPP
//           butterflies and adds the ten sums to the instance's slot in the wave's plane.  Quarters of a wave may
//           meet in an instance, so the four quarters take turns with plain read / add / write around the steps of
//           the NEXT group (program order between the turns, the read's latency hidden behind a step); across waves
//           there are four planes, added by the flush in fixed order as before: bit-reproducible from run to run.
//   Measured and not kept: LDS float atomics for the plane update (they serialise their lanes: 1.13 ms, the LDS pipe
//   2.4x as busy); 8 steps per reduction with 8 pixels per lane (halves the fold, needs 165 VGPRs: 0.68 ms at 3
//   waves/SIMD against 0.59 ms at 4); one float4 per lane and turn instead of three floats (spills: 0.69 ms); a fused
//   v_add_f32_dpp fold by inline assembly and skipping the steps past the longest list (no change); branch-free turns
//   (every lane reads / adds / writes in every turn, the other quarters' lanes into a dummy word, so that a group of four
//   steps is one basic block: bit-equal, 0.5111 against 0.5097 ms); round 6: s_setprio by phase (walk above or below the staging /
//   flush phases: +-0.3 %) and a length-balanced assignment of the sixteen quarters to the four waves (census: -2.8 % wave-steps at
//   C3, profiles/r06_quarter_balance_census.txt; profiles/r06_composite_bwd_experiments.txt).  profiles/r03_composite_bwd_phases.txt: a wave walks
//   for 68 % of its life, and the walk is sensitive to VALU and LDS at once with neither saturated.
// A wave walks max over its quarters (C3: 0.73x the steps of the 8x8 walk), and phase 2 touches only hit quarters.
// The median-depth gradient (one Gaussian per pixel, the list position the forward recorded) is added once per pixel,
// by an LDS atomic before the walk of the batch that position falls into.
#define GSR_BWQ_U 4          // list steps per transposed reduction
#ifndef GSR_BWQ_BATCH
#define GSR_BWQ_BATCH 128    // instances staged per round (64 or 128): the four waves and their quarters re-synchronise
                             // once per round -- census at C3: 30.1 k workgroup steps at 64, 28.3 k at 128 (20.9 k ideal)
#endif
#define GSR_BWQ_HALVES ((GSR_BWQ_BATCH + 63) / 64)
#define GSR_BWQ_LIST (GSR_BWQ_BATCH + 8)   // bytes per quarter list: the entries + 8 sentinels
#define GSR_BWQ_SENT GSR_BWQ_BATCH         // batch index of the sentinel record (opacity 0)
#define GSR_BWQ_QSTRIDE 66   // float2 per quarter in the slab: 4 steps x 16 pixels + 2 pad (16-B aligned, banks shifted)

// FX: exp on the transcendental unit (gs_exp_hw), after a forward that ran in fast_exp mode: same instruction, same
// bits, same alpha >= 1/255 decisions as that forward.  (Measured at C3: 0.5108 -> 0.5084 ms -- nine VALU instructions
// fewer per step, but v_exp_f32 issues at a quarter of their rate; a third mode -- hardware exp after a
// BIT-EXACT forward, with a re-evaluation by gs_exp wherever opacity * G came within 4e-6 of 1/255 so that the decisions
// stayed the forward's -- was built, passed the summation-bound tests and measured 0.5346 ms: removed.)
// TM_DECL / TM(k) / TM_END: phase timing of a diagnostic build (gsr_bwd_timing.h); empty in the product
#ifndef GSR_BWQ_WAVES
#define GSR_BWQ_WAVES 4      // waves per SIMD the register allocation is held to (experiment: 5 with GSR_BWQ_BATCH=96, docs/DESIGN_history_r1-r4.md s4.3)
#endif
// CONLY: only the colour gradient is supplied (dL_dpix_depth / _median / _opacity NULL = zero: a colour-only loss, the usual
// 3DGS training call): their loads, their two FMAs of the step and the four of phase 2 are left out.  The remaining
// operations are the general kernel's with literal zeros, so the rows are bit-equal to a call with explicit zero planes.
template <bool FLAGS, bool TSEL, bool FX, bool CONLY>
__global__ __launch_bounds__(GSR_BWD_THREADS) __attribute__((amdgpu_waves_per_eu(GSR_BWQ_WAVES, GSR_BWQ_WAVES))) void composite_bwd_quarter_kernel(
    int T, int chunk, int gx, int W, int H, const GsBg bgv, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const GsRec* __restrict__ recs, const uint32_t* __restrict__ goff,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ med_pos,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_depth, const float* __restrict__ dL_dpix_median,
    const float* __restrict__ dL_dpix_opacity, float* __restrict__ rows, uint8_t* __restrict__ row_flags,
    const GsCtl* __restrict__ ctl)
{
	__shared__ float4 sA[GSR_BWQ_BATCH + 1];   // q0: px, py, -a/2, -b        (last slot: the sentinel)
	__shared__ float4 sB[GSR_BWQ_BATCH + 1];   // q1: -c/2, opacity, depth, pcut
	__shared__ float4 sC[GSR_BWQ_BATCH + 1];   // q2: r, g, b, -
	__shared__ uint32_t s_row[GSR_BWQ_BATCH];
	__shared__ uint16_t s_qmask[GSR_BWQ_BATCH];   // the forward's block masks of the staged instances (when it left them)
	__shared__ __attribute__((aligned(16))) float s_plane[4][GSR_BWQ_BATCH * GSR_PLANE_STRIDE];
	__shared__ __attribute__((aligned(16))) float2 s_slab[4][4 * GSR_BWQ_QSTRIDE];
	__shared__ __attribute__((aligned(16))) uint8_t s_list[4][4][GSR_BWQ_LIST];
	__shared__ int s_max[4];
	if (blockIdx.x == 0 && threadIdx.x == 0 && bgv.flag_dst != nullptr) *bgv.flag_dst = bgv.flag;   // regime word (GsBg)
	// XCD-banded static order, or (skewed frames) longest walk first: launch_tile_order
	// (a banded backward, GSR_BWD_PART_BAND_*, walks the tiles [tile_lo, tile_hi) only: the XCD bands are cut out of THAT range, so
	// that a band still runs on all eight XCDs -- cut out of the whole image, the upper half of the tiles lives on four of them)
	const int tb_lo = (int)bgv.tile_lo, tb_n = (int)min(bgv.tile_hi, (uint32_t)T) - tb_lo;
	const int local = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	const int tile = bgv.tile_order ? ((int)blockIdx.x < T ? (int)bgv.tile_order[blockIdx.x] : T) : tb_lo + local;
	if ((!bgv.tile_order && ((int)(blockIdx.x >> 3) >= chunk || local >= tb_n)) || tile >= T) return;
	if ((uint32_t)tile < bgv.tile_lo || (uint32_t)tile >= bgv.tile_hi) return;   // (longest-first order of a skewed frame: every tile is offered to both bands)
	const int tid = threadIdx.x;
	const int lane = tid & 63, wv = tid >> 6, qd = lane >> 4;
	TM_DECL
	const int tx = tile % gx, ty = tile / gx;
	// lane -> pixel as in composite_fwd_quarter_kernel: quarter qd of wave wv is one 4x4 block
	const int lx = ((wv & 1) << 3) + ((qd & 1) << 2) + (lane & 3);
	const int ly = ((wv >> 1) << 3) + ((qd >> 1) << 2) + ((lane >> 2) & 3);
	const int px = tx * GSR_BLOCK_X + lx, py = ty * GSR_BLOCK_Y + ly;
	const bool inside = px < W && py < H;
	const float pixfx = (float)px, pixfy = (float)py;
	const float fbx = (float)(tx * GSR_BLOCK_X + ((wv & 1) << 3)), fby = (float)(ty * GSR_BLOCK_Y + ((wv >> 1) << 3));
	const uint2 range = ranges[tile];
	// the forward's per-instance block masks (gs_qmask_ptr), if it left them: the cull below then costs two shifts
	const bool have_qmask = ctl->has_qmask != 0u;
	const uint16_t* __restrict__ qmask = gs_qmask_ptr(point_list, ctl->num_binned);

	const size_t sidx = (size_t)tile * GSR_TILE_PIX + (wv << 6) + ((ly & 7) << 3) + (lx & 7);   // tile-major pixel state
	const float T_final = inside ? final_T[sidx] : 0.f;
	const int lc = inside ? (int)n_contrib[sidx] : 0;          // last_contributor
	float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f, dLo = 0.f, dLm = 0.f;
	uint32_t mpos = 0u;
	if (inside) {
		const size_t HW = (size_t)H * W;
		const size_t pix_id = (size_t)W * py + px;
		// an upstream gradient the caller did not supply (NULL: an output the loss does not use) is zero: not loaded
		if (dL_dpix) {
			dLp0 = dL_dpix[pix_id];
			dLp1 = dL_dpix[HW + pix_id];
			dLp2 = dL_dpix[2 * HW + pix_id];
		}
		if (!CONLY) {
			if (dL_dpix_depth) dLd = dL_dpix_depth[pix_id];
			if (dL_dpix_median) dLm = dL_dpix_median[pix_id];   // channel 0 only (backward.cu:481-482)
			if (dL_dpix_opacity) dLo = dL_dpix_opacity[pix_id];
			mpos = med_pos[sidx];
		}
	}
	float2* slab = s_slab[wv];
	float* plane = s_plane[wv];
	// phase-2 role of this lane: step u2 of its own quarter, pixel row r2 of the 4x4 block; the upstream gradients of
	// that row's four pixels move into registers once, through the (still unused) slab
	const int u2 = (lane >> 2) & 3, r2 = lane & 3;
	float4 g_pix[4];   // dL_dpixel rgb, dL_ddepth of pixel (sx, r2)
	float g_op[4];     // dL_dopacity
	{
		float4* tmp = reinterpret_cast<float4*>(slab);   // 64 pixels x 2 float4 = 2 KiB <= the wave's slab
		tmp[2 * lane] = make_float4(dLp0, dLp1, dLp2, dLd);
		if (!CONLY) tmp[2 * lane + 1] = make_float4(dLo, 0.f, 0.f, 0.f);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int sx = 0; sx < 4; sx++) {
			const int pl = (qd << 4) + (r2 << 2) + sx;   // lane that owns pixel (sx, r2) of this quarter
			const float4 gp = tmp[2 * pl];
			// XOR Latin square for the transposed fold of phase 2: slot k of row r2 holds component k ^ r2 of (r, g, b, depth)
			const float c0 = gp.x, c1 = gp.y, c2 = gp.z, c3 = gp.w;
			g_pix[sx] = make_float4(r2 == 0 ? c0 : r2 == 1 ? c1 : r2 == 2 ? c2 : c3, r2 == 0 ? c1 : r2 == 1 ? c0 : r2 == 2 ? c3 : c2,
			                        r2 == 0 ? c2 : r2 == 1 ? c3 : r2 == 2 ? c0 : c1, r2 == 0 ? c3 : r2 == 1 ? c2 : r2 == 2 ? c1 : c0);
			g_op[sx] = CONLY ? 0.f : tmp[2 * pl + 1].x;
		}
		__builtin_amdgcn_wave_barrier();
	}
	const float y2 = fby + (float)(((qd >> 1) << 2) + r2), x2 = fbx + (float)((qd & 1) << 2);   // first pixel of that row
	// bg . dL_dpixel (backward.cu:584-586), loop invariant
	const float bg0 = bgv.dptr ? bgv.dptr[0] : bgv.host[0], bg1 = bgv.dptr ? bgv.dptr[1] : bgv.host[1], bg2 = bgv.dptr ? bgv.dptr[2] : bgv.host[2];
	const float bg_dot = FMA(bg2, dLp2, FMA(bg1, dLp1, FMA(bg0, dLp0, 0.f)));
	const bool any_bg = __ballot(bg_dot != 0.f) != 0ull;
	float T_ = T_final;
	float S = 0.f;   // <accum_rec, dL_dpixel>, see composite_bwd_kernel

	// quarter / block / tile maxima of last_contributor: list entries at or beyond them are dead there
	int qmax = lc;
	qmax = max(qmax, __shfl_xor(qmax, 1, 64));
	qmax = max(qmax, __shfl_xor(qmax, 2, 64));
	qmax = max(qmax, __shfl_xor(qmax, 4, 64));
	qmax = max(qmax, __shfl_xor(qmax, 8, 64));
	const int qm0 = __builtin_amdgcn_readlane(qmax, 0), qm1 = __builtin_amdgcn_readlane(qmax, 16);
	const int qm2 = __builtin_amdgcn_readlane(qmax, 32), qm3 = __builtin_amdgcn_readlane(qmax, 48);
	const int wmax = max(max(qm0, qm1), max(qm2, qm3));
	if (lane == 0) s_max[wv] = wmax;
	TM(0)
	if (tid < 3) (tid == 0 ? sA : tid == 1 ? sB : sC)[GSR_BWQ_SENT] = make_float4(0.f, 0.f, 0.f, 0.f);
	__syncthreads();
	const int bmax = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
	TM(1)

	// rows of list entries no pixel of the tile reaches (short-list regime; see composite_bwd_kernel)
	for (int i = bmax + tid; !FLAGS && i < (int)(range.y - range.x); i += GSR_BWD_THREADS) {
		const uint32_t id = point_list[range.x + i];
		const uint4 q3 = recs[id].q3;
		float4* dst = reinterpret_cast<float4*>(rows + (size_t)(goff[id] + gs_row_in_rect(q3.x, q3.y, (q3.z >> GSR_Q3Z_DEAD_SHIFT) & 15u, tx, ty)) * GSR_ROW_STRIDE);
		dst[0] = make_float4(0.f, 0.f, 0.f, 0.f);
		dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
		dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
	}
	TM(2)
	// ---- software-pipelined staging (as in composite_bwd_kernel) ----
	// thread t fetches 16-B part (t & 3) of the records of staged instances (t >> 2) + 64 h
	const int srec = tid >> 2, spart = tid & 3;
	auto load_id = [&](int t, int h) -> uint32_t {   // id of list position t-1-(srec + 64 h) (0 when outside the walk)
		const int sr = srec + 64 * h;
		return (t > 0 && sr < min(GSR_BWQ_BATCH, t)) ? point_list[range.x + (uint32_t)(t - 1 - sr)] : 0u;
	};
	auto load_part = [&](int t, int h, uint32_t id) -> float4 {
		float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
		if (t > 0 && srec + 64 * h < min(GSR_BWQ_BATCH, t)) {
			v = reinterpret_cast<const float4*>(recs + id)[spart];
			if (spart == 3) {   // q3 = {rect min, rect max, clamp bits | dead corners, tiles}: .w <- first row, .z <- dead corners << 16 | the forward's block mask
				v.w = __uint_as_float(goff[id]);
				const uint32_t dead = (__float_as_uint(v.z) >> GSR_Q3Z_DEAD_SHIFT) & 15u;
				v.z = __uint_as_float((dead << 16) | (have_qmask ? (uint32_t)qmask[range.x + (uint32_t)(t - 1 - (srec + 64 * h))] : 0u));
			}
		}
		return v;
	};
	uint32_t id_next[GSR_BWQ_HALVES];
	float4 part_cur[GSR_BWQ_HALVES];
#pragma unroll
	for (int h = 0; h < GSR_BWQ_HALVES; h++) {
		id_next[h] = load_id(bmax - GSR_BWQ_BATCH, h);
		part_cur[h] = load_part(bmax, h, load_id(bmax, h));
	}
	const uint8_t* my_list = &s_list[wv][qd][0];
	const char* recA = reinterpret_cast<const char*>(sA);
	const char* recB = reinterpret_cast<const char*>(sB);
	const char* recC = reinterpret_cast<const char*>(sC);
	float2* my_slab_w = slab + qd * GSR_BWQ_QSTRIDE + (lane & 15);                      // + 16 * step
	const float4* my_slab_r = reinterpret_cast<const float4*>(slab + qd * GSR_BWQ_QSTRIDE + u2 * 16 + r2 * 4);

	for (int top = bmax; top > 0; top -= GSR_BWQ_BATCH) {
		const int cnt = min(GSR_BWQ_BATCH, top);
		TM(3)
		__syncthreads();   // the previous flush has read sA / sB / the planes
		TM(4)
#pragma unroll
		for (int h = 0; h < GSR_BWQ_HALVES; h++) {
			const int sr = srec + 64 * h;
			if (sr < cnt) {
				if (spart == 0) sA[sr] = part_cur[h];
				else if (spart == 1) sB[sr] = part_cur[h];
				else if (spart == 2) sC[sr] = part_cur[h];
				else {
					const uint32_t q3x = __float_as_uint(part_cur[h].x), q3y = __float_as_uint(part_cur[h].y), q3w = __float_as_uint(part_cur[h].w);
					const uint32_t q3z = __float_as_uint(part_cur[h].z);
					s_row[sr] = q3w + gs_row_in_rect(q3x, q3y, (q3z >> 16) & 15u, tx, ty);
					s_qmask[sr] = (uint16_t)q3z;
				}
			}
		}
		TM(5)
		float4 part_next[GSR_BWQ_HALVES];
#pragma unroll
		for (int h = 0; h < GSR_BWQ_HALVES; h++) {
			part_next[h] = load_part(top - GSR_BWQ_BATCH, h, id_next[h]);
			id_next[h] = load_id(top - 2 * GSR_BWQ_BATCH, h);
		}
		for (int i = tid; i < 4 * GSR_BWQ_BATCH * GSR_PLANE_STRIDE / 4; i += GSR_BWD_THREADS)
			reinterpret_cast<float4*>(&s_plane[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
		// this wave's four lists: sentinels, then (below) the hits of each quarter in list order
		if (lane < 4 * GSR_BWQ_LIST / 16) {
			const uint32_t ss = GSR_BWQ_SENT * 0x01010101u;
			reinterpret_cast<uint4*>(&s_list[wv][0][0])[lane] = make_uint4(ss, ss, ss, ss);
		}
		TM(6)
		__syncthreads();
		TM(7)
		// median-depth gradient (backward.cu:566-569): to the Gaussian the forward recorded as this pixel's median
		// (list position mpos, 1-based), when it is in this batch -- once per pixel per backward
		if (mpos != 0u && dLm != 0.f && (uint32_t)top >= mpos && (uint32_t)top - mpos < (uint32_t)cnt)
			atomicAdd(&plane[((uint32_t)top - mpos) * GSR_PLANE_STRIDE + 7], dLm);   // slot 7: the depth sum
		// lane l: which of the wave's four 4x4 blocks can staged instance l + 64 h touch, and is it still in front of
		// the quarter's last contributor; the hits of each quarter are appended to its list in list order
		int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#define GSR_APPEND(QQ, CNT)                                                                                          \
	{                                                                                                                \
		const bool h_ = (mk >> (QQ)) & 1u;                                                                           \
		const unsigned long long bm = __ballot(h_);                                                                  \
		const int pos_ = CNT + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)); \
		if (h_) s_list[wv][QQ][pos_] = (uint8_t)jl;                                                                  \
		CNT += __popcll(bm);                                                                                         \
	}
#pragma unroll
		for (int h = 0; h < GSR_BWQ_HALVES; h++) {
			const int jl = lane + 64 * h;
			if (64 * h >= cnt) break;
			uint32_t mk = 0;
			if (jl < cnt) {
				if (have_qmask) {   // bits 4*row + col of the tile's blocks -> bits 2*row + col of this wave's
					const uint32_t m16 = (uint32_t)s_qmask[jl] >> (8 * (wv >> 1) + 2 * (wv & 1));
					mk = (m16 & 3u) | ((m16 >> 2) & 12u);
				} else {
					mk = gs_quarter_mask<2>(sA[jl], sB[jl], fbx, fby, 0xfu);
				}
				const int pos = top - 1 - jl;
				mk &= (pos < qm0 ? 1u : 0u) | (pos < qm1 ? 2u : 0u) | (pos < qm2 ? 4u : 0u) | (pos < qm3 ? 8u : 0u);
			}
			GSR_APPEND(0, c0)
			GSR_APPEND(1, c1)
			GSR_APPEND(2, c2)
			GSR_APPEND(3, c3)
		}
#undef GSR_APPEND
		__builtin_amdgcn_wave_barrier();
		TM(8)
		const int n = max(max(c0, c1), max(c2, c3));
		const int my_cnt = qd == 0 ? c0 : qd == 1 ? c1 : qd == 2 ? c2 : c3;
		// one list step of every quarter: (q, w) of the lane's pixel for the quarter's entry, into slab row `st`
		struct Rec { float4 A, B; float3 C; };
		auto fetch = [&](const uint32_t j) -> Rec {
			Rec r;
			r.A = *reinterpret_cast<const float4*>(recA + j * 16);
			r.B = *reinterpret_cast<const float4*>(recB + j * 16);
			r.C = *reinterpret_cast<const float3*>(recC + j * 16);
			return r;
		};
		auto step = [&](const Rec& rc, const uint32_t j, const int st) {
			const float4 A = rc.A, B = rc.B;
			const float3 Cc = rc.C;
			const int pos = top - 1 - (int)j;   // == `contributor` after decrement (backward.cu:520)
			const float dx = A.x - pixfx, dy = A.y - pixfy;
			const float power = FMA(A.w * dx, dy, FMA(B.x * dy, dy, (A.z * dx) * dx));
			float G, a0;
			bool live;
			if (!FX) {
				G = gs_exp(power);
				a0 = B.y * G;
				live = (pos < lc) & (power <= 0.0f) & (power >= B.w) & (!(a0 < 1.0f / 255.0f));
			} else {
				// gs_exp_hw is valid for every power <= 0, and power < pcut implies opacity * exp(power) < (1/255) exp(-1e-3):
				// the forward's pcut pre-test is implied by the alpha test
				G = gs_exp_hw(power);
				a0 = B.y * G;
				live = (pos < lc) & (power <= 0.0f) & (!(a0 < 1.0f / 255.0f));
			}
			// a dead pixel (and the sentinel of an exhausted quarter) is carried through with G masked to 0: alpha = 0,
			// 1/(1-alpha) = 1 (gsr_selftest), w = 0, q = 0, S <- fma(0, ., S)
			const float Gm = live ? G : 0.f;
			const float alpha = fminf(0.99f, B.y * Gm);
			const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
			const float test_T = T_ * rinv;
			const float w = alpha * test_T;
			const float cd = FMA(Cc.x, dLp0, FMA(Cc.y, dLp1, FMA(Cc.z, dLp2, CONLY ? 0.f : FMA(B.z, dLd, dLo))));   // CONLY: dLd = dLo = 0, depth >= 0: that FMA is +0
			const float diff = cd - S;
			float dL_dalpha = diff * test_T;
			if (any_bg) {                                                         // backward.cu:584-587
				asm volatile("");
				dL_dalpha = FMA(-(T_final * rinv), bg_dot, dL_dalpha);
			}
			const float q = Gm * dL_dalpha;
			S = FMA(alpha, diff, S);
			T_ = TSEL ? (live ? test_T : T_) : test_T;
			my_slab_w[16 * st] = make_float2(q, w);
		};
		// four list entries per read, fetched one group ahead of their use; the scheduling barrier keeps the read up
		// here (left alone, the scheduler sinks it to its use and exposes a full LDS latency per group)
		// The sums of a (quarter, instance) go into the instance's slot of the wave's plane; quarters may meet in an
		// instance, so the four quarters take turns (LDS float atomics would serialise every lane: measured 2x the
		// kernel time): the results of one phase 2 stay pending and are added by quarter k around step k of the NEXT
		// group -- plain read / add / write, the read's latency hidden behind the step, program order between the turns.
		float pn0 = 0.f, pn1 = 0.f, pn2 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f;
		float* pn_dst = plane;
		bool pn_act = false;
		const bool r2_hi = (r2 & 2) != 0;
		auto turn_read = [&](const int qq) {
			if (pn_act & (qd == qq)) {
				t0 = pn_dst[0];
				t1 = pn_dst[4];
				if (!r2_hi) t2 = pn_dst[8];
			}
		};
		auto turn_write = [&](const int qq) {
			if (pn_act & (qd == qq)) {
				pn_dst[0] = t0 + pn0;
				pn_dst[4] = t1 + pn1;
				if (!r2_hi) pn_dst[8] = t2 + pn2;
			}
		};
		uint32_t pk = *reinterpret_cast<const uint32_t*>(my_list);
		for (int i = 0; i < n; i += GSR_BWQ_U) {
			const uint32_t cur = pk;
			pk = *reinterpret_cast<const uint32_t*>(my_list + i + GSR_BWQ_U);
			__builtin_amdgcn_sched_barrier(0);
			// (requesting the record of step k + 1 before evaluating step k -- two alternating register sets, what bought
			// 7 % in the per-wave kernel -- changes nothing here: measured 0.528 against 0.527 ms)
			turn_read(0); step(fetch(cur & 0xffu), cur & 0xffu, 0); turn_write(0);
			turn_read(1); step(fetch((cur >> 8) & 0xffu), (cur >> 8) & 0xffu, 1); turn_write(1);
			turn_read(2); step(fetch((cur >> 16) & 0xffu), (cur >> 16) & 0xffu, 2); turn_write(2);
			turn_read(3); step(fetch(cur >> 24), cur >> 24, 3); turn_write(3);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			// ---- phase 2: lane = (quarter qd, step u2, pixel row r2) ----
			{
				const bool act = i + u2 < my_cnt;
				const uint32_t j = act ? (cur >> (8 * u2)) & 0xffu : 0u;
				const float2 ctr = *reinterpret_cast<const float2*>(recA + j * 16);
				const float4 v01 = my_slab_r[0], v23 = my_slab_r[1];   // (q, w) of the row's four pixels
				const float dy = ctr.y - y2, X = ctr.x - x2;
				const float qv[4] = {v01.x, v01.z, v23.x, v23.z}, wv_[4] = {v01.y, v01.w, v23.y, v23.w};
				float S0 = 0.f, a0 = 0.f, a2 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f, a8 = 0.f, a9 = 0.f;
#pragma unroll
				for (int sx = 0; sx < 4; sx++) {
					const float dx = X - (float)sx;
					const float t = qv[sx] * dx;
					S0 += qv[sx];
					a0 += t;
					a2 = FMA(t, dx, a2);
					if (!CONLY) a5 = FMA(wv_[sx], g_op[sx], a5);      // backward.cu:575 (+ :607 below: + sum q); CONLY: w * 0 added to +0 stays +0
					a6 = FMA(wv_[sx], g_pix[sx].x, a6);
					a7 = FMA(wv_[sx], g_pix[sx].y, a7);
					a8 = FMA(wv_[sx], g_pix[sx].z, a8);
					a9 = FMA(wv_[sx], g_pix[sx].w, a9);
				}
				a5 += S0;
				float a1 = dy * S0, a3 = dy * a0;
				float a4 = dy * a1;
				// fold the four rows (lanes r2 = 0..3 of the same quarter and step): quad butterflies
				GSR_DPP_ADD(a0, 0xB1); GSR_DPP_ADD(a1, 0xB1); GSR_DPP_ADD(a2, 0xB1); GSR_DPP_ADD(a3, 0xB1); GSR_DPP_ADD(a4, 0xB1);
				GSR_DPP_ADD(a5, 0xB1);
				GSR_DPP_ADD(a0, 0x4E); GSR_DPP_ADD(a1, 0x4E); GSR_DPP_ADD(a2, 0x4E); GSR_DPP_ADD(a3, 0x4E); GSR_DPP_ADD(a4, 0x4E);
				GSR_DPP_ADD(a5, 0x4E);
				// the four colour / depth sums: a6 .. a9 of row r2 hold component k ^ r2 (g_pix above), so two cross-register
				// quad exchanges and one more leave component r2, summed over the four rows, in row r2 -- 3 DPP adds for four
				// sums, and no select: it is the sum this row owns (slot 4 + r2)
				const float k1 = a6 + GSR_DPP_OF(a7, 0xB1), k2 = a8 + GSR_DPP_OF(a9, 0xB1);
				// lane r2 takes sums r2, 4 + r2 and (r2 < 2) 8 + r2 to the instance's slot of the wave's plane (pending):
				// slots 0..3 = a0..a3, 4..7 = colour r, g, b and depth, 8 = a4, 9 = a5
				const bool lo1 = (r2 & 1) != 0;
				pn0 = r2_hi ? (lo1 ? a3 : a2) : (lo1 ? a1 : a0);
				pn1 = k1 + GSR_DPP_OF(k2, 0x4E);
				pn2 = lo1 ? a5 : a4;
				pn_act = act;
				pn_dst = plane + j * GSR_PLANE_STRIDE + r2;
			}
			__builtin_amdgcn_wave_barrier();
		}
#pragma unroll
		for (int qq = 0; qq < 4; qq++) {   // the last group's sums
			turn_read(qq);
			turn_write(qq);
		}
		// flush: one thread per staged instance adds the four planes in fixed order and stores the 48-B row
		TM(9)
		__syncthreads();
		TM(10)
		for (int fj = tid; fj < cnt; fj += GSR_BWD_THREADS) {
			float v[10];
#pragma unroll
			for (int k = 0; k < 10; k++)
				v[k] = ((s_plane[0][fj * GSR_PLANE_STRIDE + k] + s_plane[1][fj * GSR_PLANE_STRIDE + k]) +
				        s_plane[2][fj * GSR_PLANE_STRIDE + k]) + s_plane[3][fj * GSR_PLANE_STRIDE + k];
			const float4 A = sA[fj], B = sB[fj];
			const float ca = -2.f * A.z, cb = -A.w, cc = -2.f * B.x, op = B.y;
			const float M10 = v[0], M01 = v[1], M20 = v[2], M11 = v[3], M02 = v[8];
			const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);   // backward.cu:493-494
			const uint32_t my_row = s_row[fj];
			float4* dst = reinterpret_cast<float4*>(rows + (size_t)my_row * GSR_ROW_STRIDE);
			dst[0] = make_float4(-(op * FMA(cb, M01, ca * M10)) * ddelx_dx, -(op * FMA(cb, M10, cc * M01)) * ddely_dy,
			                     -0.5f * op * M20, -0.5f * op * M11);
			dst[1] = make_float4(-0.5f * op * M02, v[9], v[4], v[5]);   // row layout unchanged: (.., opacity, r, g), (b, depth)
			dst[2] = make_float4(v[6], v[7], 0.f, 0.f);
			if (FLAGS) row_flags[my_row] = 1;
		}
#pragma unroll
		for (int h = 0; h < GSR_BWQ_HALVES; h++) part_cur[h] = part_next[h];
		TM(11)
	}
	TM_END
}

// v_rcp_f32(1.0) == 1.0 (and a few neighbours behave): composite_bwd relies on it to carry dead pixels through
// without a select on T (TSEL = false).  Checked once per process by gsr_selftest; a failing device gets TSEL = true.
__global__ void bwd_selftest_kernel(const float* __restrict__ in, uint32_t* __restrict__ out)
{
	const float one = in[0];                       // runtime 1.0f: not folded by the compiler
	const float r = __builtin_amdgcn_rcpf(one - in[1]);   // in[1] = 0.0f
	uint32_t ok = (__float_as_uint(r) == 0x3f800000u) ? 1u : 0u;
	const float t = 0.37f * in[0];
	if (t * r == t) ok |= 2u;
	out[0] = ok;
	out[1] = __float_as_uint(r);
}
void launch_bwd_selftest(const float* in, uint32_t* out, hipStream_t s)
{
	hipLaunchKernelGGL(bwd_selftest_kernel, dim3(1), dim3(1), 0, s, in, out);
}

// ---- banded backward: the two Gaussian classes of a split row (include/gsrast.h gsr_band_classes; same predicate as preprocess_bwd_kernel) ----
__global__ __launch_bounds__(256) void band_classes_kernel(int P, const int* __restrict__ radii, const GsRec* __restrict__ recs, int split,
                                                           int* __restrict__ first, int* __restrict__ second)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	const bool vis = radii[idx] > 0;
	const bool c1 = vis && (int)(recs[idx].q3.y >> 16) <= split;
	if (first != nullptr) first[idx] = c1 ? 1 : 0;
	if (second != nullptr) second[idx] = (vis && !c1) ? 1 : 0;
}
void launch_band_classes(int P, const int* radii, const GsRec* recs, int split, int* first, int* second, hipStream_t s)
{
	hipLaunchKernelGGL(band_classes_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, recs, split, first, second);
}

// ---- longest-first tile order (skewed frames) ----
__device__ __forceinline__ uint32_t gs_work_class(uint32_t m)   // 4 classes per octave, monotone in m; < 128
{
	if (m < 4u) return m;
	const int e = 31 - __clz((int)m);
	return (uint32_t)(4 * (e - 1)) + ((m >> (e - 2)) & 3u);
}
__global__ __launch_bounds__(256) void tile_work_kernel(int T, const uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ work)
{
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (tile >= T) return;
	const uint4 v = reinterpret_cast<const uint4*>(n_contrib + (size_t)tile * GSR_TILE_PIX)[lane];
	uint32_t m = max(max(v.x, v.y), max(v.z, v.w));
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
	if (lane == 0) work[tile] = m;
}
__global__ __launch_bounds__(1024) void tile_order_kernel(int T, const uint32_t* __restrict__ work, uint32_t* __restrict__ order)
{
	__shared__ uint32_t s_cnt[128], s_cur[128];
	const int tid = threadIdx.x;
	if (tid < 128) s_cnt[tid] = 0u;
	__syncthreads();
	for (int t = tid; t < T; t += 1024) atomicAdd(&s_cnt[gs_work_class(work[t])], 1u);
	__syncthreads();
	if (tid == 0) {   // descending classes: the longest walks get the first workgroups
		uint32_t run = 0;
		for (int c = 127; c >= 0; c--) { s_cur[c] = run; run += s_cnt[c]; }
	}
	__syncthreads();
	for (int t = tid; t < T; t += 1024) order[atomicAdd(&s_cur[gs_work_class(work[t])], 1u)] = (uint32_t)t;
}
void launch_tile_order(int T, const uint32_t* n_contrib, uint32_t* tile_work, uint32_t* tile_order, hipStream_t s)
{
	hipLaunchKernelGGL(tile_work_kernel, dim3((T + 3) / 4), dim3(256), 0, s, T, n_contrib, tile_work);
	hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, T, tile_work, tile_order);
}

void launch_composite_bwd(const ImgLayout& il, int W, int H, const GsBg& bg, const uint2* ranges,
                          const uint32_t* point_list, const GsRec* recs, const uint32_t* goff, const float* final_T,
                          const uint32_t* n_contrib, const uint32_t* med_pos, const float* dL_dpix, const float* dL_dpix_depth,
                          const float* dL_dpix_median, const float* dL_dpix_opacity, float* rows, uint8_t* row_flags,
                          const GsCtl* ctl, int variant, hipStream_t s)
{
	const int tb_n = (int)std::min(bg.tile_hi, (uint32_t)il.T) - (int)bg.tile_lo;   // tiles of this call (a band, or all of them)
	const int chunk = bg.tile_order ? (il.T + 7) / 8 : std::max(1, (tb_n + 7) / 8);      // (at least one workgroup: it stores the regime word)
	const bool tsel = (variant & 1) != 0, wave_lists = (variant & 2) != 0;
	const bool fx = (variant & 8) != 0;   // the forward ran in fast_exp mode (per-quarter kernels only)
	const bool flags = row_flags != nullptr;
#define GSR_LAUNCH_CB(...)                                                                                         \
	hipLaunchKernelGGL((__VA_ARGS__), dim3(chunk * 8), dim3(GSR_BWD_THREADS), 0, s, il.T, chunk, il.gx, W, H, bg,  \
	                   ranges, point_list, recs, goff, final_T, n_contrib, med_pos, dL_dpix, dL_dpix_depth, dL_dpix_median, \
	                   dL_dpix_opacity, rows, row_flags, ctl)
#ifdef GSR_AB_VARIANTS
	if (wave_lists) {
		if (flags) { if (tsel) GSR_LAUNCH_CB(composite_bwd_kernel<true, true>); else GSR_LAUNCH_CB(composite_bwd_kernel<true, false>); }
		else { if (tsel) GSR_LAUNCH_CB(composite_bwd_kernel<false, true>); else GSR_LAUNCH_CB(composite_bwd_kernel<false, false>); }
	} else
#else
	(void)wave_lists;   // refused by gsr_backward in a build without GSR_AB_VARIANTS
#endif
	if (dL_dpix_depth == nullptr && dL_dpix_median == nullptr && dL_dpix_opacity == nullptr && !tsel) {
		// colour-only loss: the specialised instantiation (a device that needs TSEL takes the general kernel, which treats NULL as zero)
		if (fx) { if (flags) GSR_LAUNCH_CB(composite_bwd_quarter_kernel<true, false, true, true>); else GSR_LAUNCH_CB(composite_bwd_quarter_kernel<false, false, true, true>); }
		else { if (flags) GSR_LAUNCH_CB(composite_bwd_quarter_kernel<true, false, false, true>); else GSR_LAUNCH_CB(composite_bwd_quarter_kernel<false, false, false, true>); }
	} else if (fx) {
		if (flags) { if (tsel) GSR_LAUNCH_CB(composite_bwd_quarter_kernel<true, true, true, false>); else GSR_LAUNCH_CB(composite_bwd_quarter_kernel<true, false, true, false>); }
		else { if (tsel) GSR_LAUNCH_CB(composite_bwd_quarter_kernel<false, true, true, false>); else GSR_LAUNCH_CB(composite_bwd_quarter_kernel<false, false, true, false>); }
	} else {
		if (flags) { if (tsel) GSR_LAUNCH_CB(composite_bwd_quarter_kernel<true, true, false, false>); else GSR_LAUNCH_CB(composite_bwd_quarter_kernel<true, false, false, false>); }
		else { if (tsel) GSR_LAUNCH_CB(composite_bwd_quarter_kernel<false, true, false, false>); else GSR_LAUNCH_CB(composite_bwd_quarter_kernel<false, false, false, false>); }
	}
#undef GSR_LAUNCH_CB
}

// ------------------------------------------------------------------------------------------------
template <int D, bool FLAGS>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
    int P, int M, const float* __restrict__ means3D, const int* __restrict__ radii, const float* __restrict__ shs,
    const float* __restrict__ scales, const float* __restrict__ rotations, float scale_modifier,
    const float* __restrict__ cov3D_precomp, const GsCam* __restrict__ cam, int W, int H, float tan_fovx,
    float tan_fovy, float h_x, float h_y, int sh_vec4, int act, const GsRec* __restrict__ recs,
    const uint32_t* __restrict__ goff, const float* __restrict__ rows, const uint8_t* __restrict__ row_flags,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor,
    float* __restrict__ dL_dmeans, float* __restrict__ dL_dcov, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dscale, float* __restrict__ dL_drot, int mask_colors, int cls_mode, int cls_split)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	// banded backward: this launch writes the Gaussians of ONE class only (BwdArgs::cls_mode) -- class 1 = invisible, or the tile
	// rect ends at or before tile row cls_split (all their rows exist after the first band); class 2 = the rest.  The two launches of a
	// banded backward together write every Gaussian exactly once, each from exactly the rows, in exactly the order, of the
	// single-call backward: bit-identical outputs.
	bool mine = true;
	if (cls_mode != 0) {
		const bool v_ = idx < P && radii[idx] > 0;
		const int rmaxy = v_ ? (int)(recs[idx].q3.y >> 16) : 0;
		mine = ((!v_ || rmaxy <= cls_split) ? 1 : 2) == cls_mode;
		if (__ballot(mine && idx < P) == 0ull) return;   // wave-uniform: nothing of this wave belongs to the class
	}
	float a_[GSR_ROW_STRIDE];
	{
		// Same sums, same order as gs_sum_rows, but the rows are fetched wave-cooperatively: the rows of 64
		// consecutive Gaussians are ONE contiguous span [goff[g0], goff[g0 + 64]) of 48-B records (~14 KB at C3),
		// copied through LDS in slabs of GSR_SUM_SLAB rows with 1 KiB per load instruction; each lane then adds
		// its own rows out of the slab in ascending order.  (Per-lane gathers touched 64 cache lines per
		// instruction: 130 us at C3.)
		__shared__ float4 s_rows[4][GSR_SUM_SLAB * 3];
		__shared__ uint8_t s_flag[4][(GSR_SUM_SLAB + 63) / 64 * 64];   // validity bytes of the slab's rows
		const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
		const int g0 = blockIdx.x * 256 + wv * 64;
		if (g0 >= P) return;   // wave-uniform
		const uint32_t b = goff[min(idx, P)], e = (idx < P && mine) ? goff[idx + 1] : b;   // goff has P + 1 entries; a Gaussian of the other class: no rows
		const uint32_t wb = goff[g0], we = goff[min(g0 + 64, P)];
		const bool is_long = e - b > (uint32_t)GSR_SUM_LONG;
#pragma unroll
		for (int i = 0; i < GSR_ROW_STRIDE; i++) a_[i] = 0.f;
		float4* slab = s_rows[wv];
		for (uint32_t base = wb; base < we; base += GSR_SUM_SLAB) {
			const uint32_t cnt = min((uint32_t)GSR_SUM_SLAB, we - base);
			const float4* src = reinterpret_cast<const float4*>(rows + (size_t)base * GSR_ROW_STRIDE);
			uint8_t* fl = s_flag[wv];
			constexpr bool flagged = FLAGS;   // long-list regime (see composite_bwd)
			if (flagged) {
#pragma unroll
				for (int it = 0; it < (GSR_SUM_SLAB + 63) / 64; it++) {
					const uint32_t r = it * 64 + lane;
					fl[r] = r < cnt ? row_flags[base + r] : (uint8_t)0;
				}
				__builtin_amdgcn_wave_barrier();
			}
#pragma unroll
			for (int it = 0; it < (GSR_SUM_SLAB * 3 + 63) / 64; it++) {
				const uint32_t j = it * 64 + lane;
				if (j < cnt * 3 && (!flagged || fl[j / 3])) slab[j] = gs_ld_stream(src + j);   // unwritten rows are not fetched; read once
			}
			__builtin_amdgcn_wave_barrier();
			const uint32_t lo = max(b, base), hi = is_long ? lo : min(e, base + cnt);   // long spans: by the whole wave, below
			for (uint32_t r = lo; r < hi; r++) {
				if (flagged && !fl[r - base]) continue;
				const float4* ar = slab + (r - base) * 3;
				const float4 v0 = ar[0], v1 = ar[1], v2 = ar[2];
				a_[0] += v0.x; a_[1] += v0.y; a_[2] += v0.z; a_[3] += v0.w; a_[4] += v1.x; a_[5] += v1.y;
				a_[6] += v1.z; a_[7] += v1.w; a_[8] += v2.x; a_[9] += v2.y;
			}
			__builtin_amdgcn_wave_barrier();
		}
		gs_sum_long_rows(b, e, rows, FLAGS ? row_flags : nullptr, lane, a_);   // spans > GSR_SUM_LONG rows: the whole wave per Gaussian
	}
	if (idx >= P || !mine) return;
	const bool vis = radii[idx] > 0;
	// user-facing copies of the composite-stage gradients (rasterize_points.cu:209 returns them)
	dL_dmean2D[3 * (size_t)idx] = a_[0];
	dL_dmean2D[3 * (size_t)idx + 1] = a_[1];
	dL_dmean2D[3 * (size_t)idx + 2] = 0.f;
	// f1: d/d(raw opacity) = dL_dopacity * op * (1 - op), op = sigmoid(raw) kept in the record
	dL_dopacity[idx] = (vis && (act & GSR_ACT_OPACITY_SIGMOID)) ? a_[5] * recs[idx].q1.y * (1.f - recs[idx].q1.y) : a_[5];
	if (mask_colors && vis) {
		// GSR_PART_COLORS_EARLY (factored multi-GPU exchange): leave dRGB -- the colour gradient with the clamped channels zeroed,
		// the very product gs_sh_backward forms (backward.cu:35-40) -- here already, so that the all-gather of this view's colour
		// slot can start before the SH-direction stage has read 192 B of coefficients per Gaussian
		const uint32_t cl = recs[idx].q3.z;
		dL_dcolor[3 * (size_t)idx] = a_[6] * ((cl & 1u) ? 0.f : 1.f);
		dL_dcolor[3 * (size_t)idx + 1] = a_[7] * (((cl >> 1) & 1u) ? 0.f : 1.f);
		dL_dcolor[3 * (size_t)idx + 2] = a_[8] * (((cl >> 2) & 1u) ? 0.f : 1.f);
	} else {
		dL_dcolor[3 * (size_t)idx] = a_[6];
		dL_dcolor[3 * (size_t)idx + 1] = a_[7];
		dL_dcolor[3 * (size_t)idx + 2] = a_[8];
	}

	float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float dscale[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
	if (vis) {
		const float* view = cam->view;
		const float* proj = cam->proj;
		const float3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
		float cov3D[6];
		if (cov3D_precomp != nullptr) {
#pragma unroll
			for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
		} else {
			float inv_len;
			const float3 sc = gs_act_scale({scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]}, act);
			const float4 q = gs_act_rot(*reinterpret_cast<const float4*>(rotations + 4 * (size_t)idx), act, &inv_len);
			cov3d_from_scale_rot(sc, scale_modifier, q, cov3D);   // recomputed, bit-identical to forward
		}
		// ---- computeCov2DCUDA (backward.cu:144-274) ----
		const float dLc_x = a_[2], dLc_y = a_[3], dLc_z = a_[4];
		Cov2D c;
		cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, c);
		const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
		const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
		const float a = c.cov.m[0][0] + 0.3f, b = c.cov.m[0][1], cc = c.cov.m[1][1] + 0.3f;
		const float denom = FMA(-b, b, a * cc);
		float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
		const float denom2inv = 1.0f / FMA(denom, denom, 0.0000001f);
#define T_(i, j) c.T.m[i][j]
#define V_(i, j) c.Vrk.m[i][j]
#define W_(i, j) c.W.m[i][j]
		if (denom2inv != 0) {
			const float dmac = FMA(-a, cc, denom);
			dL_da = denom2inv * FMA(dmac, dLc_z, FMA(2 * b * cc, dLc_y, -cc * cc * dLc_x));
			dL_dc = denom2inv * FMA(dmac, dLc_x, FMA(2 * a * b, dLc_y, -a * a * dLc_z));
			dL_db = denom2inv * 2 * FMA(a * b, dLc_z, FMA(-FMA(2 * b, b, denom), dLc_y, b * cc * dLc_x));
			dcov[0] = FMA(T_(1, 0) * T_(1, 0), dL_dc, FMA(T_(0, 0) * T_(1, 0), dL_db, T_(0, 0) * T_(0, 0) * dL_da));
			dcov[3] = FMA(T_(1, 1) * T_(1, 1), dL_dc, FMA(T_(0, 1) * T_(1, 1), dL_db, T_(0, 1) * T_(0, 1) * dL_da));
			dcov[5] = FMA(T_(1, 2) * T_(1, 2), dL_dc, FMA(T_(0, 2) * T_(1, 2), dL_db, T_(0, 2) * T_(0, 2) * dL_da));
			dcov[1] = FMA(2 * T_(1, 0) * T_(1, 1), dL_dc, FMA(FMA(T_(0, 1), T_(1, 0), T_(0, 0) * T_(1, 1)), dL_db, 2 * T_(0, 0) * T_(0, 1) * dL_da));
			dcov[2] = FMA(2 * T_(1, 0) * T_(1, 2), dL_dc, FMA(FMA(T_(0, 2), T_(1, 0), T_(0, 0) * T_(1, 2)), dL_db, 2 * T_(0, 0) * T_(0, 2) * dL_da));
			dcov[4] = FMA(2 * T_(1, 1) * T_(1, 2), dL_dc, FMA(FMA(T_(0, 2), T_(1, 1), T_(0, 1) * T_(1, 2)), dL_db, 2 * T_(0, 2) * T_(0, 1) * dL_da));
		}
#define TV(i, k) FMA(T_(i, 2), V_(k, 2), FMA(T_(i, 1), V_(k, 1), T_(i, 0) * V_(k, 0)))
		const float dL_dT00 = FMA(TV(1, 0), dL_db, 2 * TV(0, 0) * dL_da);
		const float dL_dT01 = FMA(TV(1, 1), dL_db, 2 * TV(0, 1) * dL_da);
		const float dL_dT02 = FMA(TV(1, 2), dL_db, 2 * TV(0, 2) * dL_da);
		const float dL_dT10 = FMA(TV(0, 0), dL_db, 2 * TV(1, 0) * dL_dc);
		const float dL_dT11 = FMA(TV(0, 1), dL_db, 2 * TV(1, 1) * dL_dc);
		const float dL_dT12 = FMA(TV(0, 2), dL_db, 2 * TV(1, 2) * dL_dc);
#undef TV
		const float dL_dJ00 = FMA(W_(0, 2), dL_dT02, FMA(W_(0, 1), dL_dT01, W_(0, 0) * dL_dT00));
		const float dL_dJ02 = FMA(W_(2, 2), dL_dT02, FMA(W_(2, 1), dL_dT01, W_(2, 0) * dL_dT00));
		const float dL_dJ11 = FMA(W_(1, 2), dL_dT12, FMA(W_(1, 1), dL_dT11, W_(1, 0) * dL_dT10));
		const float dL_dJ12 = FMA(W_(2, 2), dL_dT12, FMA(W_(2, 1), dL_dT11, W_(2, 0) * dL_dT10));
#undef T_
#undef V_
#undef W_
		const float tz = 1.f / c.t.z;
		const float tz2 = tz * tz;
		const float tz3 = tz2 * tz;
		const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
		const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
		const float dL_dtz = FMA((2 * h_y * c.t.y) * tz3, dL_dJ12,
		                         FMA((2 * h_x * c.t.x) * tz3, dL_dJ02, FMA(-(h_y * tz2), dL_dJ11, -h_x * tz2 * dL_dJ00)));
		// transformVec4x3Transpose (auxiliary.h:89-97)
		dmean[0] = FMA(view[2], dL_dtz, FMA(view[1], dL_dty, view[0] * dL_dtx));
		dmean[1] = FMA(view[6], dL_dtz, FMA(view[5], dL_dty, view[4] * dL_dtx));
		dmean[2] = FMA(view[10], dL_dtz, FMA(view[9], dL_dty, view[8] * dL_dtx));

		// ---- preprocessCUDA (backward.cu:346-412) ----
		const float3 m = mean;
		const float4 m_hom = xform4x4(m, proj);
		const float m_w = 1.0f / (m_hom.w + 0.0000001f);
		const float mul1 = (FMA(proj[8], m.z, FMA(proj[4], m.y, proj[0] * m.x)) + proj[12]) * m_w * m_w;
		const float mul2 = (FMA(proj[9], m.z, FMA(proj[5], m.y, proj[1] * m.x)) + proj[13]) * m_w * m_w;
		const float g2x = a_[0], g2y = a_[1];
		dmean[0] += FMA(FMA(-proj[3], mul2, proj[1] * m_w), g2y, FMA(-proj[3], mul1, proj[0] * m_w) * g2x);
		dmean[1] += FMA(FMA(-proj[7], mul2, proj[5] * m_w), g2y, FMA(-proj[7], mul1, proj[4] * m_w) * g2x);
		dmean[2] += FMA(FMA(-proj[11], mul2, proj[9] * m_w), g2y, FMA(-proj[11], mul1, proj[8] * m_w) * g2x);
		const float mul3 = FMA(view[10], m.z, FMA(view[6], m.y, view[2] * m.x)) + view[14];
		const float gd = a_[9];
		dmean[0] += FMA(-view[3], mul3, view[2]) * gd;
		dmean[1] += FMA(-view[7], mul3, view[6]) * gd;
		dmean[2] += FMA(-view[11], mul3, view[10]) * gd;

		if (scales != nullptr) {
			// computeCov3D backward (backward.cu:278-341)
			float inv_len;
			const float4 q = gs_act_rot(*reinterpret_cast<const float4*>(rotations + 4 * (size_t)idx), act, &inv_len);
			const float3 sa = gs_act_scale({scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]}, act);
			const float r = q.x, x = q.y, y = q.z, z = q.w;
			const M3 R = quat_to_R(q);
			const float s[3] = {scale_modifier * sa.x, scale_modifier * sa.y, scale_modifier * sa.z};
			M3 S;
#pragma unroll
			for (int ci = 0; ci < 3; ci++)
#pragma unroll
				for (int ri = 0; ri < 3; ri++) S.m[ci][ri] = 0.f;
			S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
			const M3 Mm = m3_mul(S, R);
			M3 dSig;
			dSig.m[0][0] = dcov[0]; dSig.m[0][1] = 0.5f * dcov[1]; dSig.m[0][2] = 0.5f * dcov[2];
			dSig.m[1][0] = 0.5f * dcov[1]; dSig.m[1][1] = dcov[3]; dSig.m[1][2] = 0.5f * dcov[4];
			dSig.m[2][0] = 0.5f * dcov[2]; dSig.m[2][1] = 0.5f * dcov[4]; dSig.m[2][2] = dcov[5];
			M3 M2;
#pragma unroll
			for (int ci = 0; ci < 3; ci++)
#pragma unroll
				for (int ri = 0; ri < 3; ri++) M2.m[ci][ri] = 2.0f * Mm.m[ci][ri];
			const M3 dL_dM = m3_mul(M2, dSig);
			const M3 Rt = m3_t(R);
			M3 dMt = m3_t(dL_dM);
#pragma unroll
			for (int i = 0; i < 3; i++)
				dscale[i] = FMA(Rt.m[i][2], dMt.m[i][2], FMA(Rt.m[i][1], dMt.m[i][1], Rt.m[i][0] * dMt.m[i][0]));
#pragma unroll
			for (int i = 0; i < 3; i++)
#pragma unroll
				for (int j = 0; j < 3; j++) dMt.m[i][j] *= s[i];
#define D_(i, j) dMt.m[i][j]
			dq[0] = FMA(2 * x, D_(1, 2) - D_(2, 1), FMA(2 * y, D_(2, 0) - D_(0, 2), 2 * z * (D_(0, 1) - D_(1, 0))));
			dq[1] = FMA(-4 * x, D_(2, 2) + D_(1, 1), FMA(2 * r, D_(1, 2) - D_(2, 1), FMA(2 * z, D_(2, 0) + D_(0, 2), 2 * y * (D_(1, 0) + D_(0, 1)))));
			dq[2] = FMA(-4 * y, D_(2, 2) + D_(0, 0), FMA(2 * z, D_(1, 2) + D_(2, 1), FMA(2 * r, D_(2, 0) - D_(0, 2), 2 * x * (D_(1, 0) + D_(0, 1)))));
			dq[3] = FMA(-4 * z, D_(1, 1) + D_(0, 0), FMA(2 * y, D_(1, 2) + D_(2, 1), FMA(2 * x, D_(2, 0) + D_(0, 2), 2 * r * (D_(0, 1) - D_(1, 0)))));
#undef D_
			// f1 chain rules: scale = exp(raw) -> * scale ; rot = raw / |raw| -> (g - q (q.g)) / |raw|
			if (act & GSR_ACT_SCALE_EXP) { dscale[0] *= sa.x; dscale[1] *= sa.y; dscale[2] *= sa.z; }
			if (act & GSR_ACT_ROT_NORMALIZE) {
				const float qg = q.x * dq[0] + q.y * dq[1] + q.z * dq[2] + q.w * dq[3];
				dq[0] = (dq[0] - q.x * qg) * inv_len; dq[1] = (dq[1] - q.y * qg) * inv_len;
				dq[2] = (dq[2] - q.z * qg) * inv_len; dq[3] = (dq[3] - q.w * qg) * inv_len;
			}
		}
	}
#pragma unroll
	for (int i = 0; i < 3; i++) dL_dmeans[3 * (size_t)idx + i] = dmean[i];   // the SH stage adds to it next: stays cached
	if (dL_dcov != nullptr) {   // optional when the covariance came from scale / rotation: nobody reads it then
#pragma unroll
		for (int i = 0; i < 6; i++) gs_st_stream(dL_dcov + 6 * (size_t)idx + i, dcov[i]);
	}
#pragma unroll
	for (int i = 0; i < 3; i++) gs_st_stream(dL_dscale + 3 * (size_t)idx + i, dscale[i]);
	gs_st_stream(reinterpret_cast<float4*>(dL_drot + 4 * (size_t)idx), make_float4(dq[0], dq[1], dq[2], dq[3]));
}

// ------------------------------------------------------------------------------------------------
// SH part of the per-Gaussian backward (computeColorFromSH backward, backward.cu:20-139), its own kernel:
// fused into preprocess_bwd it pushed that kernel to 146 VGPRs (3 waves/SIMD) for work that lives on
// memory-level parallelism.  Runs after preprocess_bwd: reads dL_dcolor, adds its mean gradient to dL_dmeans.
//
// gs_sh_backward: from J = d(rgb)/d(direction) of the Gaussian (9 floats the forward left, gs_sh_dir_jacobian; all zero at
// degree 0) returns dc[k] = d(rgb)/d(sh_k) (so that
// dL_dsh[k][ch] = dc[k] * dRGB[ch]), dRGB = dL_dcolor with the clamped channels zeroed (Q12), and dmean_sh, the
// gradient reaching the mean through the view direction (backward.cu:126-138).
// dc[k] = d(rgb) / d(sh_k) for the unit view direction (x, y, z): the SH basis of the active degree (forward.cu:20-71,
// backward.cu:60-123 dRGBdsh*).  Shared by the per-view SH backward below and by sh_grad_from_colors_kernel, which
// rebuilds the SH gradients of OTHER ranks' views from their colour gradients with the very same arithmetic.
template <int D>
__device__ __forceinline__ void gs_sh_basis(const float x, const float y, const float z, float* dc)
{
	dc[0] = bSH_C0;
	if (D > 0) {
		dc[1] = -bSH_C1 * y; dc[2] = bSH_C1 * z; dc[3] = -bSH_C1 * x;
		if (D > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
			dc[4] = bSH_C2[0] * xy; dc[5] = bSH_C2[1] * yz;
			dc[6] = bSH_C2[2] * (FMA(2.f, zz, -xx) - yy);
			dc[7] = bSH_C2[3] * xz; dc[8] = bSH_C2[4] * (xx - yy);
			if (D > 2) {
				dc[9] = bSH_C3[0] * y * FMA(3.f, xx, -yy);
				dc[10] = bSH_C3[1] * xy * z;
				dc[11] = bSH_C3[2] * y * (FMA(4.f, zz, -xx) - yy);
				dc[12] = bSH_C3[3] * z * FMA(-3.f, yy, FMA(-3.f, xx, 2.f * zz));
				dc[13] = bSH_C3[4] * x * (FMA(4.f, zz, -xx) - yy);
				dc[14] = bSH_C3[5] * z * (xx - yy);
				dc[15] = bSH_C3[6] * x * FMA(-3.f, yy, xx);
			}
		}
	}
}

template <int D>
__device__ __forceinline__ void gs_sh_backward(const float3 m, const GsCam* __restrict__ cam, uint32_t clamped,
                                               const float* __restrict__ dLc, const float* J, float* dc, float* dRGB,
                                               float* dmean_sh)
{
		const float3 dir_orig = {m.x - cam->campos[0], m.y - cam->campos[1], m.z - cam->campos[2]};
		const float len = sqrtf(FMA(dir_orig.z, dir_orig.z, FMA(dir_orig.y, dir_orig.y, dir_orig.x * dir_orig.x)));
		const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
#pragma unroll
		for (int ch = 0; ch < 3; ch++) dRGB[ch] = dLc[ch] * (((clamped >> ch) & 1u) ? 0.f : 1.f);
		// d(rgb) / d(direction): evaluated by preprocess_fwd from the coefficients it held (gs_sh_dir_jacobian, gsr_common.h)
		const float* dRGBdx = J;
		const float* dRGBdy = J + 3;
		const float* dRGBdz = J + 6;
		gs_sh_basis<D>(x, y, z, dc);
		const float ddx = FMA(dRGBdx[2], dRGB[2], FMA(dRGBdx[1], dRGB[1], dRGBdx[0] * dRGB[0]));
		const float ddy = FMA(dRGBdy[2], dRGB[2], FMA(dRGBdy[1], dRGB[1], dRGBdy[0] * dRGB[0]));
		const float ddz = FMA(dRGBdz[2], dRGB[2], FMA(dRGBdz[1], dRGB[1], dRGBdz[0] * dRGB[0]));
		// dnormvdv (auxiliary.h:107-117)
		const float3 v = dir_orig;
		const float sum2 = FMA(v.z, v.z, FMA(v.y, v.y, v.x * v.x));
		const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
		dmean_sh[0] = (FMA(-(v.z * v.x), ddz, FMA(-(v.y * v.x), ddy, FMA(-v.x, v.x, sum2) * ddx))) * invsum32;
		dmean_sh[1] = (FMA(-(v.z * v.y), ddz, FMA(FMA(-v.y, v.y, sum2), ddy, (-v.x * v.y) * ddx))) * invsum32;
		dmean_sh[2] = (FMA(FMA(-v.z, v.z, sum2), ddz, FMA(-(v.y * v.z), ddy, (-v.x * v.z) * ddx))) * invsum32;
	
}

// preprocess_bwd_sh with the SH rows and their gradients staged through LDS wave-cooperatively (gs_wave_rows_to_lds):
// used when the stored rows hold exactly the active degree (M == (D+1)^2, the steady state) and the bases are
// 16-B aligned; every global access is then a full 1 KiB-per-instruction stream.  Same arithmetic, same results.
// floats per staged SH row (>= 1 so that the templates stay well-formed at degree 0 with split storage)
constexpr int gs_sh_row_floats(int deg, bool split)
{
	const int rf = split ? ((deg + 1) * (deg + 1) - 1) * 3 : (deg + 1) * (deg + 1) * 3;
	return rf > 0 ? rf : 1;
}

// COLORS (both SH kernels): the FACTORED form of the stage for the multi-GPU gradient exchange (gaustudio_amd/parallel.py
// FactoredGradExchange): dL_dsh is not written at all; instead dL_dcolor is overwritten in place with dRGB, the
// clamp-masked colour gradient the SH basis gets multiplied with (12 B per Gaussian and view travel instead of the
// (D+1)^2 x 12 B of dL_dsh; sh_grad_from_colors_kernel rebuilds the sum over all views).  dL_dmeans gets its SH term
// as usual.
template <int D, bool SPLIT, bool COLORS>
__global__ __launch_bounds__(256) void preprocess_bwd_sh_coop_kernel(
    int g_base, int P, const float* __restrict__ means3D, const int* __restrict__ radii, const float* __restrict__ shjac,
    const GsCam* __restrict__ cam, const uint32_t* __restrict__ clampw,
    const float* __restrict__ dL_dcolor, float* __restrict__ dL_dmeans, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dsh_rest, int write_colors)
{
	extern __shared__ __attribute__((aligned(16))) float sh_slab[];
	constexpr int NC = (D + 1) * (D + 1);
	constexpr int RF = SPLIT ? (NC - 1) * 3 : NC * 3;   // floats per staged row
	constexpr int RFA = RF > 0 ? RF : 1;
	// this launch covers the Gaussians [g_base, P): g_base is a multiple of 256 (16-B aligned row chunks)
	const int idx = g_base + blockIdx.x * 256 + threadIdx.x;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int g0 = g_base + blockIdx.x * 256 + wv * 64;
	const int nrows = min(64, P - g0);
	if (nrows <= 0) return;   // wave-uniform
	const bool vis = idx < P && radii[idx] > 0;
	constexpr int RFP = gs_row_stride<RFA>();   // padded row stride in the slab (bank conflicts)
	float* slab = sh_slab + wv * 64 * RFP;
	float* row = slab + lane * RFP;
	if (vis) {
		float J[9], dc[NC], dRGB[3], dmean_sh[3];
		if (D > 0) {
			gs_load_shjac(shjac, P, idx, J);
		} else {
#pragma unroll
			for (int k = 0; k < 9; k++) J[k] = 0.f;
		}
		const float3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
		gs_sh_backward<D>(m, cam, clampw[(size_t)idx * 4], dL_dcolor + 3 * (size_t)idx, J, dc, dRGB, dmean_sh);
#define OSH(i) (dc[(i) / 3] * dRGB[(i) % 3])
		if (COLORS) {
			if (write_colors) {   // (0: the geometry stage left dRGB already and a collective may be reading the slot by now)
				float* dcol = const_cast<float*>(dL_dcolor) + 3 * (size_t)idx;
				dcol[0] = dRGB[0]; dcol[1] = dRGB[1]; dcol[2] = dRGB[2];
			}
		} else if (SPLIT) {
			float* ddc = dL_dsh + 3 * (size_t)idx;
			ddc[0] = OSH(0); ddc[1] = OSH(1); ddc[2] = OSH(2);
#pragma unroll
			for (int i = 3; i < NC * 3; i++) row[i - 3] = OSH(i);
		} else if (RF % 4 == 0) {
#pragma unroll
			for (int i = 0; i < RF / 4; i++)
				reinterpret_cast<float4*>(row)[i] = make_float4(OSH(4 * i), OSH(4 * i + 1), OSH(4 * i + 2), OSH(4 * i + 3));
		} else {
#pragma unroll
			for (int i = 0; i < RF; i++) row[i] = OSH(i);
		}
#undef OSH
#pragma unroll
		for (int i = 0; i < 3; i++) dL_dmeans[3 * (size_t)idx + i] += dmean_sh[i];
	} else if (idx < P && !COLORS) {
		if (SPLIT) {
			float* ddc = dL_dsh + 3 * (size_t)idx;
			ddc[0] = ddc[1] = ddc[2] = 0.f;
		}
		if (RF % 4 == 0) {
#pragma unroll
			for (int i = 0; i < RF / 4; i++) reinterpret_cast<float4*>(row)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
		} else {
#pragma unroll
			for (int i = 0; i < RF; i++) row[i] = 0.f;
		}
	}
	if (RF > 0 && !COLORS) {
		__builtin_amdgcn_wave_barrier();
		gs_wave_lds_to_rows<RFA>((SPLIT ? dL_dsh_rest : dL_dsh) + (size_t)g0 * RF, nrows, slab, lane);
	}
}

// The same for rows that hold MORE coefficients than the active degree uses (MS > (D+1)^2: the first iterations of a
// training run, which raise the degree every 1000 steps over [P,16,3] storage): the few active coefficients are read per
// lane, the gradient rows -- active part, zeros above -- leave through the LDS slab as full streams.  (The per-lane kernel
// below writes a 192-B row with twelve 16-B stores at a stride of 192 B per lane: 0.127 ms at C3 against 0.090.)
template <int D, int MS>
__global__ __launch_bounds__(256) void preprocess_bwd_sh_wide_kernel(
    int g_base, int P, const float* __restrict__ means3D, const int* __restrict__ radii, const float* __restrict__ shjac,
    const GsCam* __restrict__ cam, const uint32_t* __restrict__ clampw, const float* __restrict__ dL_dcolor,
    float* __restrict__ dL_dmeans, float* __restrict__ dL_dsh)
{
	extern __shared__ __attribute__((aligned(16))) float sh_slab[];
	constexpr int NC = (D + 1) * (D + 1);
	constexpr int RF = MS * 3;                     // floats per stored row
	static_assert(NC < MS && RF % 4 == 0, "wide rows only");
	const int idx = g_base + blockIdx.x * 256 + threadIdx.x;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int g0 = g_base + blockIdx.x * 256 + wv * 64;
	const int nrows = min(64, P - g0);
	if (nrows <= 0) return;   // wave-uniform
	const bool vis = idx < P && radii[idx] > 0;
	constexpr int RFP = gs_row_stride<RF>();
	float* slab = sh_slab + wv * 64 * RFP;
	float* row = slab + lane * RFP;
#pragma unroll
	for (int i = 0; i < RF / 4; i++) reinterpret_cast<float4*>(row)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	if (vis) {
		float J[9], dc[NC], dRGB[3], dmean_sh[3];
		if (D > 0) gs_load_shjac(shjac, P, idx, J);
		else { for (int k = 0; k < 9; k++) J[k] = 0.f; }
		const float3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
		gs_sh_backward<D>(m, cam, clampw[(size_t)idx * 4], dL_dcolor + 3 * (size_t)idx, J, dc, dRGB, dmean_sh);
#pragma unroll
		for (int i = 0; i < NC * 3; i++) row[i] = dc[i / 3] * dRGB[i % 3];
#pragma unroll
		for (int i = 0; i < 3; i++) dL_dmeans[3 * (size_t)idx + i] += dmean_sh[i];
	}
	__builtin_amdgcn_wave_barrier();
	gs_wave_lds_to_rows<RF>(dL_dsh + (size_t)g0 * RF, nrows, slab, lane);
}

template <int D, bool SPLIT, bool COLORS>
__global__ __launch_bounds__(256) void preprocess_bwd_sh_kernel(
    int g_base, int P, int M, const float* __restrict__ means3D, const int* __restrict__ radii, const float* __restrict__ shjac,
    const GsCam* __restrict__ cam, int sh_vec4, const uint32_t* __restrict__ clampw,
    const float* __restrict__ dL_dcolor, float* __restrict__ dL_dmeans, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dsh_rest, int write_colors)
{
	const int idx = g_base + blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	constexpr int NC = (D + 1) * (D + 1);
	const bool vis = radii[idx] > 0;
	float J[9], dc[NC], dRGB[3], dmean_sh[3];
	if (!vis) {
		if (COLORS) return;                     // the row of dL_dcolor of a culled Gaussian is zero already
		if (SPLIT) {
			// split storage (f1): dL_dsh -> dL_df_dc [P,1,3], dL_dsh_rest -> dL_df_rest [P,M-1,3]
			float* ddc = dL_dsh + 3 * (size_t)idx;
			float* drest = dL_dsh_rest + (size_t)idx * (M - 1) * 3;
			ddc[0] = ddc[1] = ddc[2] = 0.f;
			for (int i = 0; i < (M - 1) * 3; i++) drest[i] = 0.f;
		} else {
			float* dsh = dL_dsh + (size_t)idx * M * 3;
			if (sh_vec4)
				for (int i = 0; i < M * 3 / 4; i++) reinterpret_cast<float4*>(dsh)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
			else
				for (int i = 0; i < M * 3; i++) dsh[i] = 0.f;
		}
		return;
	}
	if (D > 0) gs_load_shjac(shjac, P, idx, J);
	else { for (int k = 0; k < 9; k++) J[k] = 0.f; }
	const float3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
	gs_sh_backward<D>(m, cam, clampw[(size_t)idx * 4], dL_dcolor + 3 * (size_t)idx, J, dc, dRGB, dmean_sh);
#define OSH(i) (dc[(i) / 3] * dRGB[(i) % 3])
	if (COLORS) {
		if (write_colors) {
			float* dcol = const_cast<float*>(dL_dcolor) + 3 * (size_t)idx;
			dcol[0] = dRGB[0]; dcol[1] = dRGB[1]; dcol[2] = dRGB[2];
		}
	} else if (SPLIT) {
		float* ddc = dL_dsh + 3 * (size_t)idx;
		float* drest = dL_dsh_rest + (size_t)idx * (M - 1) * 3;
		ddc[0] = dc[0] * dRGB[0]; ddc[1] = dc[0] * dRGB[1]; ddc[2] = dc[0] * dRGB[2];
#pragma unroll
		for (int i = 3; i < NC * 3; i++) drest[i - 3] = OSH(i);
		for (int i = NC * 3; i < M * 3; i++) drest[i - 3] = 0.f;   // coefficients above the active degree
	} else {
		float* dsh = dL_dsh + (size_t)idx * M * 3;
		if (sh_vec4 && (NC * 3) % 4 == 0) {
#pragma unroll
			for (int i = 0; i < NC * 3 / 4; i++)
				reinterpret_cast<float4*>(dsh)[i] = make_float4(OSH(4 * i), OSH(4 * i + 1), OSH(4 * i + 2), OSH(4 * i + 3));
			for (int i = NC * 3 / 4; i < M * 3 / 4; i++) reinterpret_cast<float4*>(dsh)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
		} else {
#pragma unroll
			for (int i = 0; i < NC * 3; i++) dsh[i] = OSH(i);
			for (int i = NC * 3; i < M * 3; i++) dsh[i] = 0.f;   // coefficients above the active degree
		}
	}
#undef OSH
#pragma unroll
	for (int i = 0; i < 3; i++) dL_dmeans[3 * (size_t)idx + i] += dmean_sh[i];
}

void launch_preprocess_bwd(const BwdArgs& a, const GsCam* cam, const GsRec* recs, const uint32_t* clampw, const float* shjac, const uint32_t* goff,
                           const float* rows, const uint8_t* row_flags, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                           float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest, float* dL_dscale,
                           float* dL_drot, int parts, int sh_g0, int sh_g1, hipStream_t s)
{
	const float h_y = a.H / (2.0f * a.tan_fovy);   // rasterizer_impl.cu:391-392
	const float h_x = a.W / (2.0f * a.tan_fovx);
	// 16-B vector access to the SH rows / dL_dsh rows needs 16-B aligned bases and a row size multiple of 16 B
	const int sh_vec4 = (a.shs != nullptr && a.shs_rest == nullptr && dL_dsh != nullptr && ((uintptr_t)a.shs % 16 == 0) &&
	                     ((uintptr_t)dL_dsh % 16 == 0) && ((size_t)a.M * 12) % 16 == 0) ? 1 : 0;
	dim3 grid((a.P + 255) / 256), block(256);
#define GSR_LAUNCH_PB(DEG, FL)                                                                                     \
	hipLaunchKernelGGL((preprocess_bwd_kernel<DEG, FL>), grid, block, 0, s, a.P, a.M, a.means3D, a.radii, a.shs, a.scales, \
	                   a.rotations, a.scale_modifier, a.cov3D_precomp, cam, a.W, a.H, a.tan_fovx, a.tan_fovy, h_x,   \
	                   h_y, sh_vec4, a.act, recs, goff, rows, row_flags, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,  \
	                   dL_drot, (parts & GSR_PART_COLORS_EARLY) ? 1 : 0, a.cls_mode, a.cls_split)
	if (parts & GSR_PART_GEOM) {
		if (row_flags != nullptr) { GSR_LAUNCH_PB(0, true); } else { GSR_LAUNCH_PB(0, false); }
	}
#undef GSR_LAUNCH_PB
	if (!(parts & GSR_PART_SH)) return;
	// the SH part over the Gaussians [sh_g0, sh_g1)
	sh_g0 = max(0, sh_g0);
	sh_g1 = min(a.P, sh_g1);
	if (sh_g1 <= sh_g0) return;
	const int sh_end = sh_g1;
	const int write_colors = (parts & GSR_PART_COLORS_EARLY) ? 0 : 1;
	grid = dim3((sh_g1 - sh_g0 + 255) / 256);
	if (a.shs != nullptr) {
#define GSR_LAUNCH_SH(DEG)                                                                                       \
	hipLaunchKernelGGL((preprocess_bwd_sh_kernel<DEG, SPLIT, COLORS>), grid, block, 0, s, sh_g0, sh_end, a.M, a.means3D, a.radii, shjac, \
	                   cam, sh_vec4, clampw, dL_dcolor, dL_dmean3D, dL_dsh, dL_dsh_rest, write_colors)
#define GSR_LAUNCH_SH_D()                        \
		switch (a.D) {                            \
			case 0: GSR_LAUNCH_SH(0); break;      \
			case 1: GSR_LAUNCH_SH(1); break;      \
			case 2: GSR_LAUNCH_SH(2); break;      \
			default: GSR_LAUNCH_SH(3); break;     \
		}
		// cooperative (LDS-staged) variant: rows hold exactly the active degree and every streamed base is 16-B aligned
		const bool split = a.shs_rest != nullptr;
		const bool colors = (parts & GSR_PART_SH_COLORS) != 0;   // factored form: dRGB into dL_dcolor, no dL_dsh (not with split storage)
		const int NCd = (a.D + 1) * (a.D + 1);
		// (the coefficient rows are no longer read here -- the forward left d(rgb)/d(direction), gs_sh_dir_jacobian -- so only the
		// gradient rows stream through the slab)
		const float* stream_out = split ? dL_dsh_rest : dL_dsh;
		const bool coop = a.M == NCd && (sh_g0 % 256 == 0) && (colors || ((uintptr_t)stream_out % 16 == 0)) &&
		                  (colors || !split || NCd == 1 || stream_out != nullptr);
#define GSR_LAUNCH_SHC(DEG, SPL, COL)                                                                             \
	hipLaunchKernelGGL((preprocess_bwd_sh_coop_kernel<DEG, SPL, COL>), grid, block,                                  \
	                   sizeof(float) * 256 * gs_row_stride<gs_sh_row_floats(DEG, SPL)>(), s, sh_g0, sh_end, \
	                   a.means3D, a.radii, shjac, cam, clampw, dL_dcolor, dL_dmean3D, dL_dsh, dL_dsh_rest, write_colors)
		// stored rows wider than the active degree ([P,16,3] storage while the degree is still being raised): the wide kernel
		const bool wide = !colors && !split && a.M == 16 && NCd < 16 && (sh_g0 % 256 == 0) && ((uintptr_t)dL_dsh % 16 == 0);
#define GSR_LAUNCH_SHW(DEG)                                                                                            \
	hipLaunchKernelGGL((preprocess_bwd_sh_wide_kernel<DEG, 16>), grid, block, sizeof(float) * 256 * gs_row_stride<48>(), s, sh_g0, \
	                   sh_end, a.means3D, a.radii, shjac, cam, clampw, dL_dcolor, dL_dmean3D, dL_dsh)
		if (wide) {
			switch (a.D) {
				case 0: GSR_LAUNCH_SHW(0); break;
				case 1: GSR_LAUNCH_SHW(1); break;
				default: GSR_LAUNCH_SHW(2); break;
			}
		} else if (colors && coop) {
			switch (a.D) {
				case 0: GSR_LAUNCH_SHC(0, false, true); break;
				case 1: GSR_LAUNCH_SHC(1, false, true); break;
				case 2: GSR_LAUNCH_SHC(2, false, true); break;
				default: GSR_LAUNCH_SHC(3, false, true); break;
			}
		} else if (colors) {
			constexpr bool SPLIT = false, COLORS = true;
			GSR_LAUNCH_SH_D()
		} else if (coop && split) {
			switch (a.D) {
				case 0: GSR_LAUNCH_SHC(0, true, false); break;
				case 1: GSR_LAUNCH_SHC(1, true, false); break;
				case 2: GSR_LAUNCH_SHC(2, true, false); break;
				default: GSR_LAUNCH_SHC(3, true, false); break;
			}
		} else if (coop) {
			switch (a.D) {
				case 0: GSR_LAUNCH_SHC(0, false, false); break;
				case 1: GSR_LAUNCH_SHC(1, false, false); break;
				case 2: GSR_LAUNCH_SHC(2, false, false); break;
				default: GSR_LAUNCH_SHC(3, false, false); break;
			}
		} else if (split) {
			constexpr bool SPLIT = true, COLORS = false;
			GSR_LAUNCH_SH_D()
		} else {
			constexpr bool SPLIT = false, COLORS = false;
			GSR_LAUNCH_SH_D()
		}
#undef GSR_LAUNCH_SHW
#undef GSR_LAUNCH_SHC
#undef GSR_LAUNCH_SH_D
#undef GSR_LAUNCH_SH
	} else if (dL_dsh != nullptr && a.M > 0) {
		(void)hipMemsetAsync(dL_dsh + (size_t)sh_g0 * 3 * a.M, 0, sizeof(float) * 3 * (size_t)a.M * (size_t)(sh_g1 - sh_g0), s);
	}
}

// ------------------------------------------------------------------------------------------------
// sh_grad_from_colors: dL_dsh[g] = sum over the N views r = 0 .. N-1 (in this order) of basis(dir_r(g)) (x) dRGB_r[g], the
// SH gradient of a multi-view step rebuilt from the per-view clamp-masked colour gradients (what the COLORS form of the
// SH stage leaves in dL_dcolor) and the views' camera centres.  Every term is computed with the arithmetic of the
// per-view SH backward (gs_sh_basis, the same normalisation of the view direction), and the views are added in
// ascending order: the result is bit-identical to running the N backwards on one device and letting autograd
// accumulate them.  One lane per Gaussian; rows leave through LDS in 1 KiB pieces when they hold exactly the active
// degree (gs_wave_lds_to_rows), else lane by lane.
template <int D, bool COOP>
__global__ __launch_bounds__(256) void sh_grad_from_colors_kernel(int P, int M, int N, const float* __restrict__ means3D,
                                                                  const float* __restrict__ campos,   // [N,3]
                                                                  const float* __restrict__ colors,   // [N,P,3], or nullptr:
                                                                  const uint32_t* __restrict__ msgs,  // N packed messages (gsr_comm.hip) at
                                                                  const unsigned long long* __restrict__ msg_off,   // word offsets msg_off[r]
                                                                  uint32_t hdr_words,
                                                                  float* __restrict__ dL_dsh)
{
	extern __shared__ __attribute__((aligned(16))) float sh_slab[];
	constexpr int NC = (D + 1) * (D + 1);
	constexpr int RF = NC * 3;
	constexpr int RFP = gs_row_stride<RF>();
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int g0 = blockIdx.x * 256 + wv * 64;
	const int nrows = min(64, P - g0);
	if (nrows <= 0) return;   // wave-uniform
	float acc[RF];
#pragma unroll
	for (int i = 0; i < RF; i++) acc[i] = 0.f;
	if (idx < P) {
		const float3 m = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
		bool first = true;
		for (int r = 0; r < N; r++) {
			const float* c;
			if (msgs != nullptr) {   // compacted view: the Gaussian has a row only where it was visible
				const uint32_t* m = msgs + msg_off[r];
				uint32_t row;
				if (!gs_msg_lookup(m, P, idx, row)) continue;
				c = reinterpret_cast<const float*>(m + hdr_words) + (size_t)row * 3;
			} else {
				c = colors + ((size_t)r * P + idx) * 3;
			}
			const float c0 = c[0], c1 = c[1], c2 = c[2];
			if (c0 == 0.f && c1 == 0.f && c2 == 0.f) continue;   // culled in view r (or no gradient): contributes +0
			const float3 d = {m.x - campos[3 * r], m.y - campos[3 * r + 1], m.z - campos[3 * r + 2]};
			const float len = sqrtf(FMA(d.z, d.z, FMA(d.y, d.y, d.x * d.x)));
			float dc[NC];
			gs_sh_basis<D>(d.x / len, d.y / len, d.z / len, dc);
			const float cc[3] = {c0, c1, c2};
			if (first) {
#pragma unroll
				for (int i = 0; i < RF; i++) acc[i] = dc[i / 3] * cc[i % 3];
				first = false;
			} else {
#pragma unroll
				for (int i = 0; i < RF; i++) acc[i] += dc[i / 3] * cc[i % 3];
			}
		}
	}
	if (COOP) {
		float* slab = sh_slab + wv * 64 * RFP;
		float* row = slab + lane * RFP;
		if (RF % 4 == 0) {
#pragma unroll
			for (int i = 0; i < RF / 4; i++)
				reinterpret_cast<float4*>(row)[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
		} else {
#pragma unroll
			for (int i = 0; i < RF; i++) row[i] = acc[i];
		}
		__builtin_amdgcn_wave_barrier();
		gs_wave_lds_to_rows<RF>(dL_dsh + (size_t)g0 * RF, nrows, slab, lane);
	} else if (idx < P) {
		float* dsh = dL_dsh + (size_t)idx * M * 3;
#pragma unroll
		for (int i = 0; i < RF; i++) dsh[i] = acc[i];
		for (int i = RF; i < M * 3; i++) dsh[i] = 0.f;   // coefficients above the active degree
	}
}

void launch_sh_grad_from_colors(int P, int D, int M, int N, const float* means3D, const float* campos, const float* colors,
                                const uint32_t* msgs, const unsigned long long* msg_off, uint32_t hdr_words, float* dL_dsh, hipStream_t s)
{
	const dim3 grid((P + 255) / 256), block(256);
	const bool coop = M == (D + 1) * (D + 1) && ((uintptr_t)dL_dsh % 16 == 0);
#define GSR_LAUNCH_SGC(DEG, CO)                                                                                    \
	hipLaunchKernelGGL((sh_grad_from_colors_kernel<DEG, CO>), grid, block,                                             \
	                   (CO) ? sizeof(float) * 256 * gs_row_stride<(DEG + 1) * (DEG + 1) * 3>() : 0, s, P, M, N, means3D, campos, \
	                   colors, msgs, msg_off, hdr_words, dL_dsh)
	switch (D) {
		case 0: if (coop) GSR_LAUNCH_SGC(0, true); else GSR_LAUNCH_SGC(0, false); break;
		case 1: if (coop) GSR_LAUNCH_SGC(1, true); else GSR_LAUNCH_SGC(1, false); break;
		case 2: if (coop) GSR_LAUNCH_SGC(2, true); else GSR_LAUNCH_SGC(2, false); break;
		default: if (coop) GSR_LAUNCH_SGC(3, true); else GSR_LAUNCH_SGC(3, false); break;
	}
#undef GSR_LAUNCH_SGC
}

}  // namespace gsr
