// gsr_internal.h -- private buffer layouts and kernel launchers of libgsrast.
#pragma once
#include "gsr_common.h"
#include <stddef.h>

namespace gsr {

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// geometry buffer: [GsCam][GsRec x P][tiles_touched u32 x P][goff u32 x (P+1)][block sums][reference block sums]
// (replaces GeometryState, rasterizer_impl.h:33-48: depths/clamped/radii/means2D/cov3D/conic_opacity/rgb/
// point_offsets/tiles_touched/scan space = 79 B/Gaussian in 9 arrays; here one 64-B record, cov3D is recomputed
// in backward instead of stored).  goff = exclusive scan of tiles_touched (the reference's point_offsets, shifted):
// the backward's Gaussian-major row index.  preprocess_fwd leaves one partial sum per 256-Gaussian block (bsums:
// binned tiles, refsums: tiles of the reference's getRect squares), tile_scan's second workgroup scans them and
// goff_apply finishes the scan inside each block -- no separate multi-kernel scan.
#define GSR_PRE_BLOCK 256
struct GeomLayout {
	size_t cam, recs, tiles_touched, goff, bsums, refsums, shjac, binfo, total;
	size_t nblk;
	explicit GeomLayout(size_t P)
	{
		nblk = (P + GSR_PRE_BLOCK - 1) / GSR_PRE_BLOCK;
		cam = 0;
		recs = align_up(sizeof(GsCam));
		tiles_touched = recs + align_up(sizeof(GsRec) * P);          // compact u32[P] (0 for culled)
		goff = tiles_touched + align_up(sizeof(uint32_t) * P);
		bsums = goff + align_up(sizeof(uint32_t) * (P + 1));
		refsums = bsums + align_up(sizeof(uint32_t) * (nblk + 1));
		// d(rgb) / d(view direction) of every visible Gaussian's SH colour, 9 floats: left by preprocess_fwd for the SH
		// backward (gs_sh_dir_jacobian), which then does not read the 192-B coefficient rows again
		shjac = refsums + align_up(sizeof(uint32_t) * (nblk + 1));
		// 16-B binning record {rect min, rect max, clamp | dead corners, depth bits}: what the two binning passes need of a Gaussian,
		// two per 32-B sector instead of two sectors of the 64-B record (round 5)
		binfo = shjac + align_up(sizeof(float) * 9 * P);
		total = binfo + align_up(sizeof(uint4) * P);
	}
};

// image buffer: [GsCtl][ranges uint2 x T][tile_count/cursor u32 x T][final_T f32 x T*256][n_contrib u32 x T*256]
//               [med_pos u32 x T*256]
// (replaces ImageState, rasterizer_impl.h:50-57; per-pixel state is tile-major so that a tile's 256
// threads read/write 1 KiB contiguous; ranges are sized per tile, not per pixel).  med_pos = 1-based position in
// the tile's list of the Gaussian at which the pixel's transmittance crossed 0.5 (0 = none): the forward's own
// median decision (forward.cu:368-373), which the backward uses for the median-depth gradient.
struct ImgLayout {
	size_t ctl, ranges, tile_count, final_T, n_contrib, med_pos, tile_work, tile_order, total;
	int gx, gy, T;
	ImgLayout(int W, int H)
	{
		gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X;
		gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
		T = gx * gy;
		ctl = 0;
		ranges = align_up(sizeof(GsCtl));
		tile_count = ranges + align_up(sizeof(uint2) * (size_t)T);
		final_T = tile_count + align_up(sizeof(uint32_t) * (size_t)T);
		n_contrib = final_T + align_up(sizeof(float) * (size_t)T * GSR_TILE_PIX);
		med_pos = n_contrib + align_up(sizeof(uint32_t) * (size_t)T * GSR_TILE_PIX);
		tile_work = med_pos + align_up(sizeof(uint32_t) * (size_t)T * GSR_TILE_PIX);     // backward only, skewed frames (launch_tile_order)
		tile_order = tile_work + align_up(sizeof(uint32_t) * (size_t)T);
		total = tile_order + align_up(sizeof(uint32_t) * (size_t)T);
	}
};

// binning buffer: [point_list u32 x R][keys u64 x R][keys2 u64 x R, only when a tile list is long enough for
// the radix path]   (replaces BinningState, rasterizer_impl.h:59-68: 2x u64 keys + 2x u32 values + CUB temp =
// 24 B/instance + temp; here 12 B/instance, 20 B with the radix ping-pong buffer).  Backward reads point_list only.
size_t sort_queue_bytes(size_t R, int T);   // work queue of the long-list sort (gsr_kernels_fwd.hip)
struct BinLayout {
	size_t point_list, keys, keys2, queue, total;
	explicit BinLayout(size_t R, bool with_tmp = false, int T = 0)
	{
		point_list = 0;
		keys = align_up(sizeof(uint32_t) * R);
		keys2 = keys + align_up(sizeof(uint64_t) * R);
		queue = keys2 + align_up(sizeof(uint64_t) * R);
		total = with_tmp ? queue + align_up(sort_queue_bytes(R, T)) : keys2;
		if (total == 0) total = 256;
	}
};
// per-tile lists longer than this are sorted by the radix path (needs the keys2 ping-pong buffer)
#define GSR_SORT_LDS_MAX 1024u

struct FwdArgs {
	int P, D, M, W, H;
	const float* means3D;
	const float* shs;
	const float* colors_precomp;
	const float* opacities;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* cov3D_precomp;
	float tan_fovx, tan_fovy;
	int prefiltered;
	const float* shs_rest;   // f1: SH given as [P,1,3] (shs) + [P,M-1,3] (shs_rest); nullptr = shs holds all M
	int act;                 // f1: GSR_ACT_* flags
	int tight;               // 1: bin into the tight rect (gs_tight_rect); 0: the reference's square (A/B, debugging)
	int band_lo, band_hi;    // only tile rows [band_lo, band_hi) are binned (tile-grid sharding of one view); hi <= 0: all
};

// --- launchers (gsr_kernels_fwd.hip) ---
// `cap`: capacity (instances) of the binning buffer the kernels were launched against; they are enqueued BEFORE the
// host knows the instance count and leave without touching memory when ctl->num_binned exceeds it (the host then
// re-allocates and re-launches; gsr_api.hip forward_impl).
void launch_mark_visible(int P, const float* means3D, const float* view, unsigned char* present, hipStream_t s);
void launch_preprocess_fwd(const FwdArgs& a, const GsCam* cam, const ImgLayout& il, int* radii, GsRec* recs, float* shjac, uint4* binfo,
                           uint32_t* tiles_touched, uint32_t* bsums, uint32_t* refsums, uint32_t* tile_count,
                           GsCtl* ctl, hipStream_t s);
void launch_tile_scan(int T, uint32_t* tile_count, uint2* ranges, int nblk, uint32_t* bsums, const uint32_t* refsums,
                      GsCtl* ctl, GsCtl* host_ctl, hipStream_t s);
void launch_goff_apply(int P, const uint32_t* tiles_touched, const uint32_t* bsums, uint32_t* goff, hipStream_t s);
void launch_bin_scatter(int P, int gx, const int* radii, const uint32_t* tiles_touched, const GsRec* recs,
                        const uint2* ranges, uint32_t* cursor, uint64_t* keys, const GsCtl* ctl, uint32_t cap,
                        hipStream_t s);
// binning without global atomics (default when the tile grid fits an LDS histogram)
int bin_chunks(int P, int T);
size_t bin_hist_bytes(int P, int T);
bool bin_lds_path_ok(int T);
void launch_bin_hist(int P, int gx, int T, const uint32_t* tiles_touched, const GsRec* recs, const uint4* binfo, uint32_t* Hm,
                     uint32_t* tile_count, hipStream_t s);
void launch_bin_scatter2(int P, int gx, int T, const uint32_t* tiles_touched, const GsRec* recs, const uint4* binfo, uint32_t* Hm,
                         const uint2* ranges, uint64_t* keys, const uint32_t* bsums, uint32_t* goff, const GsCtl* ctl, uint32_t cap, hipStream_t s);
// long_level: 0 = no list beyond GSR_SORT_LDS_MAX, 1 = lists up to GSR_SORT_GIANT keys (one workgroup per tile), 2 = longer ones too
#define GSR_PART_REGS 8192u     // keys a 1024-thread workgroup holds in registers: a slice of the queue pipeline; the longest list of the per-list kernel
#define GSR_SORT_MANY 1024u     // long lists in a frame from which the lists of up to GSR_SORT_GIANT keys take the per-list kernel
#ifndef GSR_SORT_GIANT
#define GSR_SORT_GIANT 8192u
#endif
void launch_tile_sort(int T, bool with_short, int long_level, const uint2* ranges, uint64_t* keys, uint64_t* keys2,
                      uint32_t* point_list, char* queue, size_t R, GsCtl* ctl, uint32_t cap, uint32_t* host_err, hipStream_t s);
void launch_composite_fwd(const ImgLayout& il, int W, int H, const uint2* ranges, const uint32_t* point_list,
                          const GsRec* recs, float* out_color, float* out_depth, float* out_median,
                          float* out_opacity, float* final_T, uint32_t* n_contrib, uint32_t* med_pos, GsCtl* ctl, uint32_t cap, uint32_t max_sorted, bool nocull, bool wave_lists, bool fast_exp,
                          const uint32_t* tile_order, uint32_t* staged_out, hipStream_t s);
void launch_tile_order_fwd(int T, const uint2* ranges, uint32_t* order, const GsCtl* ctl, uint32_t cap, hipStream_t s);

// --- launchers (gsr_kernels_bwd.hip) ---
struct BwdArgs {
	int P, D, M, W, H;
	const float* means3D;
	const float* shs;
	const float* colors_precomp;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* cov3D_precomp;
	float tan_fovx, tan_fovy;
	const int* radii;
	const float* shs_rest;
	int act;
	// banded backward (gsr_backward_ex with a GSR_BWD_PART_BAND_* bit): the per-Gaussian geometry stage writes only the Gaussians of one
	// CLASS -- 1: invisible, or the tile rect ends at or before tile row `cls_split` (every row of theirs exists once the first band
	// has been composited); 2: the others; 0: all
	int cls_mode = 0, cls_split = 0;
};
// Backward data flow (no global atomics): composite_bwd leaves ONE 48-B row of partial sums per
// (tile, Gaussian) instance, stored in GAUSSIAN-MAJOR order: row index = goff[g] + k, where goff is
// the exclusive scan of tiles_touched over Gaussians and k the raster index of the tile inside the
// Gaussian's tile rect.  preprocess_bwd then adds the rows of each Gaussian in ascending k -- a
// contiguous read and a fixed summation order.
//   row: 0,1 dL_dmean2D.xy | 2,3,4 dL_dconic a,b,c | 5 dL_dopacity | 6,7,8 dL_dcolor | 9 dL_ddepth | 10,11 unused
#define GSR_ROW_STRIDE 12
#define GSR_FLAG_AVG 1024   // average tile list length above which the backward uses per-row validity flags
#ifndef GSR_SUM_SLAB
// rows per LDS slab of preprocess_bwd's cooperative row fetch: 4.5 KiB per wave.  Round 5: 160 -> 96 -- the kernel is bound by
// the latency of a workgroup's dependent chain (row fetch -> sums -> cov2D / cov3D backward) times the generations of workgroups a
// CU holds, not by bytes (C4-inside: 0.64 GB in 0.28 ms; compacting the visible Gaussians of a block into fewer waves and wide
// transposed output stores were both built and measured: +13 % / no change), so LDS per workgroup = occupancy is what moves it:
// 5 -> 8 workgroups per CU; preprocess_bwd 0.1164 -> 0.1098 ms (C3), 0.539 -> 0.493 (C4), 0.507 -> 0.484 (C4-inside); 64 / 80
// measured the same within 1 %, 32 worse (profiles/r05_preprocess_bwd_experiments.txt)
#define GSR_SUM_SLAB 96
#endif
// scratch: [bg f32 x 4][rows f32 x R*12]
struct BwdLayout {
	size_t bg, flags, rows, total;
	BwdLayout(size_t P, size_t R)
	{
		(void)P;
		bg = 0;
		flags = bg + 256;                                           // u8 x R: 1 = the row was written by composite_bwd
		rows = flags + align_up(R > 0 ? R : 1);
		total = rows + align_up(sizeof(float) * GSR_ROW_STRIDE * (R > 0 ? R : 1));
	}
};
// The backward's own `background` (backward.cu:584-587 reads it; the forward never does) travels as a kernel
// argument: a device pointer is read by the kernel, host values ride in the argument block -- no staging launch.  The
// same argument carries the regime word gsr_inspect_backward_sums reads (stored by the kernel's first thread).
struct GsBg {
	const float* dptr;     // device-resident background, or nullptr
	float host[3];         // the values when dptr == nullptr (absent background: zeros)
	uint32_t* flag_dst;    // word 8 of the scratch's background block
	uint32_t flag;
	const uint32_t* tile_order;   // nullptr: workgroup b -> tile by XCD band; else workgroup b -> tile_order[b] (longest walk first)
	uint32_t tile_lo, tile_hi;    // banded backward: only the tiles [tile_lo, tile_hi) are walked (whole image: 0, 0xffffffff)
};
void launch_band_classes(int P, const int* radii, const GsRec* recs, int split, int* first, int* second, hipStream_t s);
// Longest-first tile order for composite_bwd on SKEWED frames (a few tiles with walks many times the mean: a workgroup that
// starts such a tile late is the kernel's tail): work[t] = the tile's longest walk (max n_contrib), tiles ordered by
// descending work class (counting sort, 4 classes per octave).  Two small launches, only when the forward saw skew.
void launch_tile_order(int T, const uint32_t* n_contrib, uint32_t* tile_work, uint32_t* tile_order, hipStream_t s);
// variant: 0 = default; other values select A/B variants of the kernel (gsr_set_option("bwd_variant", v))
void launch_composite_bwd(const ImgLayout& il, int W, int H, const GsBg& bg, const uint2* ranges,
                          const uint32_t* point_list, const GsRec* recs, const uint32_t* goff, const float* final_T,
                          const uint32_t* n_contrib, const uint32_t* med_pos, const float* dL_dpix, const float* dL_dpix_depth,
                          const float* dL_dpix_median, const float* dL_dpix_opacity, float* rows, uint8_t* row_flags,
                          const GsCtl* ctl, int variant, hipStream_t s);
// parts: GSR_PART_GEOM = the per-Gaussian geometry kernel (all P Gaussians); GSR_PART_SH = the SH kernel over
// the Gaussians [sh_g0, sh_g1) (a multiple-of-256 start; lets a caller interleave a collective per chunk)
#define GSR_PART_GEOM 1
#define GSR_PART_SH 2
#define GSR_PART_SH_COLORS 4   // with GSR_PART_SH: the factored form (dRGB into dL_dcolor in place, dL_dsh untouched)
#define GSR_PART_COLORS_EARLY 8   // the GEOMETRY kernel leaves dRGB (clamp-masked) in dL_dcolor; the COLORS SH kernel then does not write it
void launch_preprocess_bwd(const BwdArgs& a, const GsCam* cam, const GsRec* recs, const uint32_t* clampw, const float* shjac, const uint32_t* goff,
                           const float* rows, const uint8_t* row_flags, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                           float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest, float* dL_dscale,
                           float* dL_drot, int parts, int sh_g0, int sh_g1, hipStream_t s);
void launch_bwd_selftest(const float* in, uint32_t* out, hipStream_t s);
// dL_dsh[P,M,3] = sum over N views of basis(dir) (x) colors[r][g] (gaustudio_amd/parallel.py FactoredGradExchange)
void launch_sh_grad_from_colors(int P, int D, int M, int N, const float* means3D, const float* campos, const float* colors,
                                const uint32_t* msgs, const unsigned long long* msg_off, uint32_t hdr_words, float* dL_dsh, hipStream_t s);
void launch_inspect_sums(int P, const int* radii, const GsRec* recs, const uint32_t* goff, const float* rows,
                         const uint8_t* row_flags,
                         float* sums10, hipStream_t s);

}  // namespace gsr
