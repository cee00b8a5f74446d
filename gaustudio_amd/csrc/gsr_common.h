// gsr_common.h -- shared device helpers and private buffer layouts of libgsrast (gfx950 only).
//
// Floating-point contract: this library is compiled with -ffp-contract=off; every fused
// multiply-add is written explicitly (FMA()).  The arithmetic below is the "pinned contraction"
// of the reference kernels documented in DESIGN.md s3; it makes the forward pass reproducible to
// the bit against the CPU oracle used by tests/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GSR_BLOCK_X 16   // $RAST/cuda_rasterizer/config.h:16-17
#define GSR_BLOCK_Y 16
#define GSR_TILE_PIX 256

#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// ---------------------------------------------------------------------------------------------
// Per-Gaussian record written by preprocess_fwd, gathered by the compositing kernels.
// 64 bytes, 64-byte aligned: one gather touches exactly one 64-B sector.
//   q0 = { px, py, -0.5*conic_a, -conic_b }          (pre-scaled: exact power-of-two / sign changes)
//   q1 = { -0.5*conic_c, opacity, depth, pcut }
//   q2 = { r, g, b, radius(int bits) }
//   q3 = { rect_min (x | y<<16), rect_max (x | y<<16), clamped bits, goff }   (uint bits)
// rect = the TIGHT tile rect the Gaussian is binned into (gs_tight_rect below; a sub-rect of the reference's
// getRect square), goff = first row of the Gaussian in the backward's Gaussian-major row array (exclusive scan of
// tiles_touched, patched in by goff_apply_kernel).
// pcut: conservative lower bound on `power` below which alpha < 1/255 is certain, see preprocess.
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(64))) GsRec {
	float4 q0, q1, q2;
	uint4 q3;
};
static_assert(sizeof(GsRec) == 64, "GsRec must be 64 bytes");

// Camera block at the head of the geometry buffer (device copy of the host/device inputs).
struct GsCam {
	float view[16];
	float proj[16];
	float campos[4];
	float bg[4];
};

// Control words living in the image buffer (mirrored into pinned host memory by tile_scan_kernel).
struct GsCtl {
	uint32_t num_binned;     // instances actually binned (tight rects): sizes every per-instance buffer
	uint32_t max_tile_count; // longest per-tile list
	uint32_t err_prefiltered;
	uint32_t err_overflow;   // bit 0: the reference-defined count does not fit the reference's int num_rendered; bit 1: the long-list sort overflowed a work queue
	uint32_t ref_rendered;   // the reference's num_rendered: sum of getRect areas (rasterizer_impl.cu:280-284)
	uint32_t has_qmask;      // composite_fwd left one 16-bit block mask per list entry behind the list (see gs_qmask_ptr)
	uint32_t opts;           // options the forward ran with (GSR_CTL_OPT_*): the backward of this image buffer must agree
	uint32_t n_long;         // tiles whose list is beyond the one-wave register sort (> GSR_SORT_LDS_MAX keys)
};

#define GSR_CTL_OPT_FAST_EXP 1u     // compositing used gs_exp_hw: the backward must take its decisions with it as well
#define GSR_CTL_OPT_TIGHT 2u
#define GSR_CTL_OPT_CULL 4u
#define GSR_CTL_OPT_WAVE_LISTS 8u
#define GSR_CTL_OPT_BAND 16u
#define GSR_CTL_OPT_FORWARD_ONLY 32u   // the forward kept nothing for a backward (gsr_options.forward_only)

// Per-instance block masks of the forward (bit 4*row + col: which 4x4 pixel blocks of the tile the instance can touch,
// gs_quarter_mask<4>), kept for the backward: u16 per list entry, in the binning buffer right behind the list's
// num_binned ids (256-B aligned) -- over the sort keys, which are dead once the lists are sorted (6 B per instance of a
// buffer that holds at least 12).
__device__ __forceinline__ uint16_t* gs_qmask_ptr(const uint32_t* point_list, uint32_t num_binned)
{
	const size_t off = ((size_t)num_binned * 4 + 255) / 256 * 256;
	return reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(const_cast<uint32_t*>(point_list)) + off);
}

__device__ __forceinline__ float gs_exp(float p)
{
	// exp(p), p in [-80, 0]: 2^(p*log2e) with round-to-nearest-even split through the 1.5*2^23
	// constant, degree-5 minimax polynomial for 2^f on [-0.5, 0.5], exponent added as integer bits.
	const float LOG2E = 0x1.715476p+0f;
	const float MAGIC = 12582912.0f;
	float tm = FMA(p, LOG2E, MAGIC);
	float nf = tm - MAGIC;
	float f = FMA(p, LOG2E, -nf);
	float y = 0x1.5c08e6p-10f;
	y = FMA(y, f, 0x1.3d0c52p-7f);
	y = FMA(y, f, 0x1.c6b6e4p-5f);
	y = FMA(y, f, 0x1.ebf918p-3f);
	y = FMA(y, f, 0x1.62e428p-1f);
	y = FMA(y, f, 0x1.000002p+0f);
	return __int_as_float(__float_as_int(y) + (__float_as_int(tm) << 23));
}

// exp(p) on the transcendental unit: v_mul_f32 + v_exp_f32 (2 VALU instead of 9).  1 ulp of 2^x plus the rounding of
// p * log2(e): within 1e-6 relative of gs_exp for p in [-6, 0], the range in which the alpha >= 1/255 decision is taken.
// NOT reproducible on a CPU: used only in the opt-in `fast_exp` mode of both compositing kernels (docs/DESIGN_history_r1-r4.md s4.5) and,
// with an exact-decision fallback (FX = 2 in composite_bwd), in the backward of a bit-exact forward.
__device__ __forceinline__ float gs_exp_hw(float p) { return __builtin_amdgcn_exp2f(p * 0x1.715476p+0f); }

// Reproducible natural logarithm (positive normal x): x = m * 2^e, m in [1, 2), ln x = e ln2 + 2 atanh((m-1)/(m+1))
// with the series cut after s^9 (|error| < 1.5e-6 + rounding).  Only IEEE +,*,/,fma: the CPU oracle evaluates the
// same operations and gets the same bits (the tight tile rects must agree exactly between the two).
__device__ __forceinline__ float gs_log(float x)
{
	const int xb = __float_as_int(x);
	const int e = ((xb >> 23) & 0xff) - 127;
	const float m = __int_as_float((xb & 0x007fffff) | 0x3f800000);
	const float s = (m - 1.0f) / (m + 1.0f);
	const float s2 = s * s;
	float p = 0.11111111f;
	p = FMA(p, s2, 0.14285715f);
	p = FMA(p, s2, 0.2f);
	p = FMA(p, s2, 0.33333334f);
	p = FMA(p, s2, 1.0f);
	return FMA((float)e, 0.69314718f, (2.0f * s) * p);
}

// Tight tile rect (SnugBox-style, opacity-aware).  The reference bins a Gaussian into every tile of the square
// of side 2*ceil(3 sigma_max) around its centre (auxiliary.h:46-56); a pixel can only pass `alpha >= 1/255`
// (forward.cu:346) inside the ellipse  a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 opacity),  whose axis-aligned
// bounding box is |dx| <= sqrt(2 t c / det), |dy| <= sqrt(2 t a / det).  Tiles of the reference rect that hold no
// pixel centre of that box cannot contribute and are not binned: images, radii and every gradient are unchanged,
// the lists get ~23 % shorter at C3.  Conservative by construction: t carries +0.01 (exp polynomial, thresholds),
// a slack proportional to a*c/det (rounding of the per-pixel `power`, which cancels for elongated conics) and
// +0.05; the box another 0.01 px.  Degenerate / non-positive-definite / NaN inputs keep the reference rect.
// In/out: the reference rect; returns false when no tile can hold a live pixel.  Mirrored bit for bit by
// oracle/gsr_oracle.c:orc_tight_rects (explicit ternaries instead of fmin/fmax so that NaN behaves identically).
// qmax (out): 2 * teff, the bound on a dx^2 + 2 b dx dy + c dy^2 inside which every pixel that can pass lies -- for the
// corner-tile test below; negative when the rect was kept as it was (no usable ellipse).
__device__ __forceinline__ bool gs_tight_rect(float px, float py, float ca, float cb, float cc, float op, int gx, int gy,
                                              int& rminx, int& rminy, int& rmaxx, int& rmaxy, float& qmax)
{
	qmax = -1.0f;
	if (op <= 0.0f) return false;                       // alpha = min(0.99, op * G) <= 0 < 1/255 everywhere
	const float t = gs_log(255.0f * op) + 0.01f;
	if (t <= 0.0f) return false;                        // 255 * op <= 0.99: alpha < 1/255 everywhere
	const float det = FMA(-cb, cb, ca * cc);
	if (!(det > 0.0f && ca > 0.0f && cc > 0.0f)) return true;
	const float rel = (ca * cc) / det;
	if (!(rel < 1.0e5f)) return true;
	const float teff = FMA(t * 4.0e-5f, rel, t) + 0.05f;
	if (!(teff > 0.0f)) return true;
	qmax = 2.0f * teff;
	const float ex = sqrtf((2.0f * teff) * (cc / det)) + 0.01f;
	const float ey = sqrtf((2.0f * teff) * (ca / det)) + 0.01f;
	// tile column k holds pixel centres 16k .. 16k+15
	float fx0 = ceilf((px - ex - 15.0f) * 0.0625f), fx1 = floorf((px + ex) * 0.0625f) + 1.0f;
	float fy0 = ceilf((py - ey - 15.0f) * 0.0625f), fy1 = floorf((py + ey) * 0.0625f) + 1.0f;
	const float fgx = (float)gx, fgy = (float)gy;
	fx0 = fx0 > 0.0f ? fx0 : 0.0f; fx0 = fx0 < fgx ? fx0 : fgx;
	fx1 = fx1 > 0.0f ? fx1 : 0.0f; fx1 = fx1 < fgx ? fx1 : fgx;
	fy0 = fy0 > 0.0f ? fy0 : 0.0f; fy0 = fy0 < fgy ? fy0 : fgy;
	fy1 = fy1 > 0.0f ? fy1 : 0.0f; fy1 = fy1 < fgy ? fy1 : fgy;
	rminx = max(rminx, (int)fx0); rmaxx = min(rmaxx, (int)fx1);
	rminy = max(rminy, (int)fy0); rmaxy = min(rmaxy, (int)fy1);
	return rmaxx > rminx && rmaxy > rminy;
}

// Corner tiles of the tight rect.  The rect is the bounding box of the ellipse q(d) = a dx^2 + 2 b dx dy + c dy^2 <= qmax
// (gs_tight_rect); the ellipse does not reach into the corners of its box, and at C3 7 % of the binned instances are
// corner tiles of a >= 2 x 2 rect that hold no live pixel (86 % of all dead instances, tools/scene_stats.py).  A corner
// tile is dropped when the minimum of q over the tile's box of pixel centres exceeds qmax: the minimum of a convex
// quadratic over a box that does not contain the centre lies on the edges facing the centre and is found in closed form
// per edge.  Conservative: the box is the continuous hull of the pixel centres, qmax already carries the slack for the
// rounding of the per-pixel `power` (gs_tight_rect), and this evaluation gets 1e-5 of its terms' magnitude + 0.05 of its
// own; anything unordered (NaN) keeps the tile.  IEEE +, *, / only, mirrored operation for operation by
// oracle/gsr_oracle.c:tile_may_touch -- the two must take the same decision for every tile.
__device__ __forceinline__ bool gs_tile_may_touch(float px, float py, float ca, float cb, float cc, float qmax, int tx, int ty,
                                                  int W, int H)
{
	const float bx0 = (float)(16 * tx), by0 = (float)(16 * ty);
	float bx1 = bx0 + 15.0f, by1 = by0 + 15.0f;
	const float wm = (float)(W - 1), hm = (float)(H - 1);
	bx1 = bx1 < wm ? bx1 : wm;
	by1 = by1 < hm ? by1 : hm;
	const float X0 = px - bx1, X1 = px - bx0;   // range of dx = centre - pixel over the box
	const float Y0 = py - by1, Y1 = py - by0;
	const float xn = X0 > 0.0f ? X0 : (X1 < 0.0f ? X1 : 0.0f);   // point of the range nearest to 0
	const float yn = Y0 > 0.0f ? Y0 : (Y1 < 0.0f ? Y1 : 0.0f);
	if (xn == 0.0f && yn == 0.0f) return true;                   // centre inside the box
	float best = 3.0e38f, mag = 0.0f;
	if (xn != 0.0f) {   // edge dx = xn: q is minimal at dy = -b xn / c, clamped to the edge
		float dy = -(cb * xn) / cc;
		dy = dy < Y0 ? Y0 : dy;
		dy = dy > Y1 ? Y1 : dy;
		const float t0 = (ca * xn) * xn, t1 = ((2.0f * cb) * xn) * dy, t2 = (cc * dy) * dy;
		best = (t0 + t2) + t1;
		mag = (t0 + t2) + (t1 < 0.0f ? -t1 : t1);
	}
	if (yn != 0.0f) {   // edge dy = yn
		float dx = -(cb * yn) / ca;
		dx = dx < X0 ? X0 : dx;
		dx = dx > X1 ? X1 : dx;
		const float t0 = (cc * yn) * yn, t1 = ((2.0f * cb) * yn) * dx, t2 = (ca * dx) * dx;
		const float q = (t0 + t2) + t1, m = (t0 + t2) + (t1 < 0.0f ? -t1 : t1);
		if (q < best) { best = q; mag = m; }
	}
	return !(best > qmax + (1.0e-5f * mag + 0.05f));
}

// Bits 0..3: the top-left, top-right, bottom-left, bottom-right tile of the rect [rminx, rmaxx) x [rminy, rmaxy) holds no
// pixel that can pass (rects of at least 2 x 2 tiles only; qmax < 0: no usable ellipse, nothing dropped).
__device__ __forceinline__ uint32_t gs_dead_corners(float px, float py, float ca, float cb, float cc, float qmax, int rminx,
                                                    int rminy, int rmaxx, int rmaxy, int W, int H)
{
	if (!(qmax > 0.0f) || rmaxx - rminx < 2 || rmaxy - rminy < 2) return 0u;
	uint32_t dead = 0u;
	if (!gs_tile_may_touch(px, py, ca, cb, cc, qmax, rminx, rminy, W, H)) dead |= 1u;
	if (!gs_tile_may_touch(px, py, ca, cb, cc, qmax, rmaxx - 1, rminy, W, H)) dead |= 2u;
	if (!gs_tile_may_touch(px, py, ca, cb, cc, qmax, rminx, rmaxy - 1, W, H)) dead |= 4u;
	if (!gs_tile_may_touch(px, py, ca, cb, cc, qmax, rmaxx - 1, rmaxy - 1, W, H)) dead |= 8u;
	return dead;
}

// q3 of a record = {rect min (x | y << 16), rect max, clamp bits (0..2) | dead corners << 8, binned tiles}.
#define GSR_Q3Z_DEAD_SHIFT 8
// Position of tile (tx, ty) among the BINNED tiles of the Gaussian (row-major over its rect, dead corners left out): the
// instance's row behind goff[id] in the backward's Gaussian-major row buffer.
__device__ __forceinline__ uint32_t gs_row_in_rect(uint32_t q3x, uint32_t q3y, uint32_t dead, int tx, int ty)
{
	const int rminx = q3x & 0xffff, rminy = q3x >> 16, rw = (int)(q3y & 0xffff) - rminx;
	int idx = (ty - rminy) * rw + (tx - rminx);
	if (dead != 0u) {
		const int rh = (int)(q3y >> 16) - rminy;
		// dead corners in front of idx (the bottom-right one is last: never in front)
		idx -= (int)((dead & 1u) && idx > 0) + (int)(((dead >> 1) & 1u) && idx > rw - 1) + (int)(((dead >> 2) & 1u) && idx > (rh - 1) * rw);
	}
	return (uint32_t)idx;
}

// Streaming accesses (read once / written once, never re-read by the same kernel): the nontemporal hint keeps
// them from displacing the records, keys and rows that ARE re-read in L2 / Infinity Cache.  Measured at C3: SH
// backward + the next forward's preprocess -0.015 ms together.
typedef float gs_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gs_ld_stream(const float4* p)
{
	const gs_f4v v = __builtin_nontemporal_load(reinterpret_cast<const gs_f4v*>(p));
	return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gs_st_stream(float4* p, const float4 v)
{
	__builtin_nontemporal_store(gs_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<gs_f4v*>(p));
}
__device__ __forceinline__ void gs_st_stream(float* p, const float v) { __builtin_nontemporal_store(v, p); }

// two-component FP32 vector and its fused multiply-add (v_pk_fma_f32): composite_fwd blends (R, G) and (B, depth)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

struct M3 { float m[3][3]; };   // m[col][row]

__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B)
{
	M3 R;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int r = 0; r < 3; r++)
			R.m[c][r] = FMA(A.m[2][r], B.m[c][2], FMA(A.m[1][r], B.m[c][1], A.m[0][r] * B.m[c][0]));
	return R;
}
__device__ __forceinline__ M3 m3_t(const M3& A)
{
	M3 R;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
	return R;
}

__device__ __forceinline__ float3 xform4x3(const float3 p, const float* m)
{
	float3 r;
	r.x = FMA(m[8], p.z, FMA(m[4], p.y, m[0] * p.x)) + m[12];
	r.y = FMA(m[9], p.z, FMA(m[5], p.y, m[1] * p.x)) + m[13];
	r.z = FMA(m[10], p.z, FMA(m[6], p.y, m[2] * p.x)) + m[14];
	return r;
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const float* m)
{
	float4 r;
	r.x = FMA(m[8], p.z, FMA(m[4], p.y, m[0] * p.x)) + m[12];
	r.y = FMA(m[9], p.z, FMA(m[5], p.y, m[1] * p.x)) + m[13];
	r.z = FMA(m[10], p.z, FMA(m[6], p.y, m[2] * p.x)) + m[14];
	r.w = FMA(m[11], p.z, FMA(m[7], p.y, m[3] * p.x)) + m[15];
	return r;
}

__device__ __forceinline__ M3 quat_to_R(const float4 q)   // (r,x,y,z), not normalised
{
	const float r = q.x, x = q.y, y = q.z, z = q.w;
	M3 R;
	R.m[0][0] = FMA(-2.f, FMA(z, z, y * y), 1.f);
	R.m[0][1] = 2.f * FMA(-r, z, x * y);
	R.m[0][2] = 2.f * FMA(r, y, x * z);
	R.m[1][0] = 2.f * FMA(r, z, x * y);
	R.m[1][1] = FMA(-2.f, FMA(z, z, x * x), 1.f);
	R.m[1][2] = 2.f * FMA(-r, x, y * z);
	R.m[2][0] = 2.f * FMA(-r, y, x * z);
	R.m[2][1] = 2.f * FMA(r, x, y * z);
	R.m[2][2] = FMA(-2.f, FMA(y, y, x * x), 1.f);
	return R;
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float3 scale, float mod, const float4 rot, float* cov3D)
{
	M3 S;
#pragma unroll
	for (int c = 0; c < 3; c++)
#pragma unroll
		for (int r = 0; r < 3; r++) S.m[c][r] = 0.f;
	S.m[0][0] = mod * scale.x;
	S.m[1][1] = mod * scale.y;
	S.m[2][2] = mod * scale.z;
	M3 R = quat_to_R(rot);
	M3 M = m3_mul(S, R);
	M3 Mt = m3_t(M);
	M3 Sg = m3_mul(Mt, M);
	cov3D[0] = Sg.m[0][0];
	cov3D[1] = Sg.m[0][1];
	cov3D[2] = Sg.m[0][2];
	cov3D[3] = Sg.m[1][1];
	cov3D[4] = Sg.m[1][2];
	cov3D[5] = Sg.m[2][2];
}

// ---------------------------------------------------------------------------------------------
// Fused parameter activations: the operator can read GauStudio's RAW point-cloud
// attributes and apply VanillaPointCloud's activations itself (gaustudio/models/vanilla_sg.py:33-37,58-63 ->
// models/utils.py:6-31: exp / sigmoid / F.normalize) instead of having torch materialise activated copies.
// ---------------------------------------------------------------------------------------------
#define GSR_ACT_OPACITY_SIGMOID 1   // opacity = 1 / (1 + exp(-x))
#define GSR_ACT_SCALE_EXP 2         // scale   = exp(x)
#define GSR_ACT_ROT_NORMALIZE 4     // rot     = x / max(||x||_2, 1e-12)      (F.normalize defaults)

__device__ __forceinline__ float gs_act_opacity(float x, int act) { return (act & GSR_ACT_OPACITY_SIGMOID) ? 1.0f / (1.0f + expf(-x)) : x; }
__device__ __forceinline__ float3 gs_act_scale(float3 s, int act)
{
	if (act & GSR_ACT_SCALE_EXP) { s.x = expf(s.x); s.y = expf(s.y); s.z = expf(s.z); }
	return s;
}
// returns the activated quaternion; *inv_len = 1 / max(||q||, eps) (1 when no activation)
__device__ __forceinline__ float4 gs_act_rot(float4 q, int act, float* inv_len)
{
	*inv_len = 1.0f;
	if (act & GSR_ACT_ROT_NORMALIZE) {
		const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
		*inv_len = 1.0f / n;
		q.x /= n; q.y /= n; q.z /= n; q.w /= n;
	}
	return q;
}

struct Cov2D {
	float3 t;
	float txtz, tytz, limx, limy;
	M3 W, T, Vrk, cov;
};

__device__ __forceinline__ void cov2d_common(const float3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                             const float* cov3D, const float* view, Cov2D& c)
{
	float3 t = xform4x3(mean, view);
	c.limx = 1.3f * tan_fovx;
	c.limy = 1.3f * tan_fovy;
	c.txtz = t.x / t.z;
	c.tytz = t.y / t.z;
	t.x = fminf(c.limx, fmaxf(-c.limx, c.txtz)) * t.z;
	t.y = fminf(c.limy, fmaxf(-c.limy, c.tytz)) * t.z;
	c.t = t;
	M3 J;
	J.m[0][0] = fx / t.z; J.m[0][1] = 0.0f; J.m[0][2] = -(fx * t.x) / (t.z * t.z);
	J.m[1][0] = 0.0f; J.m[1][1] = fy / t.z; J.m[1][2] = -(fy * t.y) / (t.z * t.z);
	J.m[2][0] = 0.f; J.m[2][1] = 0.f; J.m[2][2] = 0.f;
	c.W.m[0][0] = view[0]; c.W.m[0][1] = view[4]; c.W.m[0][2] = view[8];
	c.W.m[1][0] = view[1]; c.W.m[1][1] = view[5]; c.W.m[1][2] = view[9];
	c.W.m[2][0] = view[2]; c.W.m[2][1] = view[6]; c.W.m[2][2] = view[10];
	c.Vrk.m[0][0] = cov3D[0]; c.Vrk.m[0][1] = cov3D[1]; c.Vrk.m[0][2] = cov3D[2];
	c.Vrk.m[1][0] = cov3D[1]; c.Vrk.m[1][1] = cov3D[3]; c.Vrk.m[1][2] = cov3D[4];
	c.Vrk.m[2][0] = cov3D[2]; c.Vrk.m[2][1] = cov3D[4]; c.Vrk.m[2][2] = cov3D[5];
	c.T = m3_mul(c.W, J);
	M3 Tt = m3_t(c.T);
	M3 Vt = m3_t(c.Vrk);
	M3 A = m3_mul(Tt, Vt);
	c.cov = m3_mul(A, c.T);
}

// ---------------------------------------------------------------------------------------------
// Pixel <-> slot mapping of the tile-major per-pixel state (final_T, n_contrib): slot s of a 16x16 tile is
// pixel (lx, ly) of 8x8 quadrant q = s>>6 = (ly>>3)*2 + (lx>>3), lane l = s&63 = (ly&7)*8 + (lx&7).
// composite_fwd and composite_bwd both run 4 wave64 per tile, wave q on quadrant q (slot = thread id).
// (A 2-pixel-per-lane forward was measured in round 1: 0.55 ms vs 0.46 ms at C3 -- the larger culling box costs
// more than packed arithmetic saves; round 2 moved the backward to this layout as well, docs/DESIGN_history_r1-r4.md s4.3.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void gs_pixel_of_thread(int tid, int& lx, int& ly)
{
	const int w = tid >> 6, l = tid & 63;
	lx = ((w & 1) << 3) + (l & 7);
	ly = ((w >> 1) << 3) + (l >> 3);
}

// Conservative wave-level cull.  Returns false only if NO pixel centre (x,y) in the closed box
// [bx0,bx1] x [by0,by1] can pass the per-pixel test `0 >= power >= pcut`, where
//   power(d) = ha*dx^2 + nb*dx*dy + hc*dy^2,  d = centre - pixel   (ha,nb,hc,pcut from the record).
// For a negative-definite form the maximum over the box lies on the edges nearest to the centre
// (power decreases along every ray from the centre); it is found in closed form per near edge.
// A slack of 0.05 + 1e-5*|terms| in `power` covers every rounding difference to the per-pixel
// evaluation, so culling a (wave, Gaussian) pair is exactly equivalent to evaluating its 64 pixels
// and rejecting each: results stay bit-identical, only work is removed.
__device__ __forceinline__ bool gs_box_may_touch(const float4 A, const float4 B, float bx0, float by0, float bx1,
                                                 float by1)
{
	const float ha = A.z, nb = A.w, hc = B.x, pcut = B.w;
	if (!(pcut <= 0.f)) return false;            // opacity < 1/255: no pixel can reach alpha >= 1/255
	const float X0 = A.x - bx1, X1 = A.x - bx0;  // range of dx over the box
	const float Y0 = A.y - by1, Y1 = A.y - by0;
	const float xn = fminf(fmaxf(0.f, X0), X1);  // point of the range nearest to 0
	const float yn = fminf(fmaxf(0.f, Y0), Y1);
	if (xn == 0.f && yn == 0.f) return true;     // centre inside the box
	if (!(ha < 0.f && hc < 0.f && 4.f * ha * hc - nb * nb > 0.f)) return true;   // not negative definite: keep
	float best = -3.0e38f;
	float mag = 0.f;
	if (xn != 0.f) {
		const float dy = fminf(fmaxf(-0.5f * nb * xn * __builtin_amdgcn_rcpf(hc), Y0), Y1);
		const float t0 = ha * xn * xn, t1 = (hc * dy + nb * xn) * dy;
		best = t0 + t1;
		mag = fabsf(t0) + fabsf(hc * dy * dy) + fabsf(nb * xn * dy);
	}
	if (yn != 0.f) {
		const float dx = fminf(fmaxf(-0.5f * nb * yn * __builtin_amdgcn_rcpf(ha), X0), X1);
		const float t0 = hc * yn * yn, t1 = (ha * dx + nb * yn) * dx;
		best = fmaxf(best, t0 + t1);
		mag = fmaxf(mag, fabsf(t0) + fabsf(ha * dx * dx) + fabsf(nb * yn * dx));
	}
	return best >= pcut - (0.05f + 1e-5f * mag);
}

// Which of the NB x NB 4x4-pixel blocks (bit NB*row + col) of the square of pixels starting at (tx0, ty0) can hold a
// pixel with 0 >= power >= pcut?  (NB = 4: the 16x16 tile, composite_fwd; NB = 2: a wave's 8x8 block, composite_bwd.)  Conservative
// like gs_box_may_touch, organised by rows of blocks: for the band of dy a block row spans, the dx-extent
// [lo, hi] of the region {power >= pc} is found in closed form (the roots of the quadratic in dx at the band's two
// ends, plus the region's extreme points in x when they lie in the band -- the extent is a concave/convex function of
// dy, so those three candidates contain its extremum); a block is hit iff its dx range meets [lo, hi].  pc = pcut minus
// a slack of 0.05 + 1e-5 * (largest |term| over the tile), the ranges are widened by 0.01 px.
template <int NB>
__device__ __forceinline__ uint32_t gs_quarter_mask(const float4 A, const float4 B, float tx0, float ty0, uint32_t allq)
{
	const float ha = A.z, nb = A.w, hc = B.x, pcut = B.w;
	if (!(pcut <= 0.f)) return 0u;               // opacity < 1/255
	const float hh = 4.f * ha * hc;
	const float D = hh - nb * nb;
	if (!(ha < 0.f && hc < 0.f && D > 1e-4f * hh)) return allq;   // not (safely) negative definite: keep everywhere
	// d = centre - pixel relative to the tile's first pixel; block column i spans dx in [rx - (4i + 3), rx - 4i]
	// (blocks cut by the image border are tested whole: conservative, and the constants stay literals)
	const float rx = A.x - tx0, ry = A.y - ty0;
	const float Dx = fmaxf(fabsf(rx), fabsf(rx - (4.f * NB - 1.f))), Dy = fmaxf(fabsf(ry), fabsf(ry - (4.f * NB - 1.f)));
	const float mag = fabsf(ha) * Dx * Dx + fabsf(hc) * Dy * Dy + fabsf(nb) * Dx * Dy;
	const float pc = pcut - (0.05f + 1e-5f * mag);
	const float rD = __builtin_amdgcn_rcpf(D);
	const float c4 = 4.f * ha * pc;                                        // > 0
	const float ex = __builtin_amdgcn_sqrtf(4.f * pc * hc * rD);           // half extent in dx of {power >= pc}
	const float dys = -0.5f * nb * ex * __builtin_amdgcn_rcpf(hc);         // dy where dx = +ex is reached
	const float r = -0.5f * __builtin_amdgcn_rcpf(ha);                     // 1 / (2|ha|)
	const float BIG = 3.0e38f;
	uint32_t mask = 0;
#pragma unroll
	for (int j = 0; j < NB; j++) {
		const float Y0 = ry - (4.f * j + 3.f), Y1 = ry - 4.f * j;           // dy over the block row
		const float d0 = FMA(-D * Y0, Y0, c4), d1 = FMA(-D * Y1, Y1, c4);   // discriminants / 1 at both ends
		const float s0 = __builtin_amdgcn_sqrtf(fmaxf(d0, 0.f)), s1 = __builtin_amdgcn_sqrtf(fmaxf(d1, 0.f));
		const float u0 = nb * Y0, u1 = nb * Y1;
		const float h0 = d0 >= 0.f ? (u0 + s0) * r : -BIG, l0 = d0 >= 0.f ? (u0 - s0) * r : BIG;
		const float h1 = d1 >= 0.f ? (u1 + s1) * r : -BIG, l1 = d1 >= 0.f ? (u1 - s1) * r : BIG;
		const float hs = (Y0 <= dys && dys <= Y1) ? ex : -BIG;
		const float ls = (Y0 <= -dys && -dys <= Y1) ? -ex : BIG;
		const float hi = fmaxf(fmaxf(h0, h1), hs) - rx;   // compared against -(4i + 3) - 0.01 <= hi - rx, ...
		const float lo = fminf(fminf(l0, l1), ls) - rx;
#pragma unroll
		for (int i = 0; i < NB; i++)
			mask |= (-(4.f * i + 3.01f) <= hi && -(4.f * i - 0.01f) >= lo) ? (1u << (NB * j + i)) : 0u;
	}
	return mask & allq;
}

// ---- wave-cooperative row staging (SH rows: 192 B per Gaussian at degree 3) -------------------------------
// A lane that reads "its" row with per-lane dwordx4 loads makes every load instruction touch 64 different rows
// (16 B out of each); staged through LDS instead, the wave copies its 64 consecutive rows as ONE contiguous
// 64*RF-float chunk, 1 KiB per load instruction, and each lane then picks its row out of the wave's LDS slab.
// `chunk` must be 16-B aligned (base 16-B aligned; 64 rows are a multiple of 256 B).  Rows whose bit in `mask`
// is clear are not fetched.  Caller brackets these with __builtin_amdgcn_wave_barrier() (LDS is in-order per wave).
// Rows sit in the slab with a PADDED stride gs_row_stride<RF>(): a stride of 48 floats (degree 3) maps the
// 64 lanes' b128 row accesses onto 8 banks out of 32 (round 2: 78 % of the SH kernel's LDS cycles were bank
// conflicts); 52 floats spread every 8 lanes over all 32 banks.  Strides that are not multiples of 4 floats are odd
// enough already and keep the plain layout (their float4 copies straddle rows).
template <int RF>
__device__ __host__ constexpr int gs_row_stride() { return (RF % 4 == 0 && RF > 0) ? RF + 4 : RF; }

template <int RF>
__device__ __forceinline__ void gs_wave_rows_to_lds(const float* __restrict__ chunk, int nrows, unsigned long long mask,
                                                    float* __restrict__ slab, int lane)
{
	constexpr int RFP = gs_row_stride<RF>();
	const int nfl = nrows * RF, nv = nfl >> 2;
#pragma unroll
	for (int it = 0; it < (16 * RF + 63) / 64; it++) {
		const int j = it * 64 + lane;
		if (j < nv) {
			const int r0 = (4 * j) / RF, r1 = (4 * j + 3) / RF;
			if (((mask >> r0) | (mask >> r1)) & 1ull) {
				const int at = (RFP == RF) ? 4 * j : r0 * RFP + (4 * j - r0 * RF);   // RF % 4 == 0: a float4 stays inside its row
				*reinterpret_cast<float4*>(slab + at) = gs_ld_stream(reinterpret_cast<const float4*>(chunk) + j);
			}
		}
	}
	const int t = (nv << 2) + lane;   // < 4 trailing floats, only in the last (partial) wave of a launch
	if (t < nfl && ((mask >> (t / RF)) & 1ull)) slab[(t / RF) * RFP + t % RF] = chunk[t];
}

template <int RF>
__device__ __forceinline__ void gs_wave_lds_to_rows(float* __restrict__ chunk, int nrows, const float* __restrict__ slab,
                                                    int lane)
{
	constexpr int RFP = gs_row_stride<RF>();
	const int nfl = nrows * RF, nv = nfl >> 2;
#pragma unroll
	for (int it = 0; it < (16 * RF + 63) / 64; it++) {
		const int j = it * 64 + lane;
		if (j < nv) {
			const int r0 = (4 * j) / RF;
			const int at = (RFP == RF) ? 4 * j : r0 * RFP + (4 * j - r0 * RF);
			gs_st_stream(reinterpret_cast<float4*>(chunk) + j, *reinterpret_cast<const float4*>(slab + at));
		}
	}
	const int t = (nv << 2) + lane;
	if (t < nfl) chunk[t] = slab[(t / RF) * RFP + t % RF];
}

// lane-private row in the slab <-> registers (float4 accesses when rows are 16-B multiples)
template <int RF>
__device__ __forceinline__ void gs_row_from_lds(const float* __restrict__ row, float* __restrict__ out)
{
	if (RF % 4 == 0) {
#pragma unroll
		for (int i = 0; i < RF / 4; i++) {
			const float4 v = reinterpret_cast<const float4*>(row)[i];
			out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
		}
	} else {
#pragma unroll
		for (int i = 0; i < RF; i++) out[i] = row[i];
	}
}

// ---- d(rgb) / d(unit view direction) of the SH colour (backward.cu:60-123: dRGBdx, dRGBdy, dRGBdz) ----
// The one part of the SH backward that needs the SH COEFFICIENTS.  Round 4: preprocess_fwd, which holds the 48 coefficients
// in registers anyway, evaluates these nine numbers and leaves them in the geometry buffer (36 B per Gaussian); the SH
// backward reads them instead of the 192-B coefficient row -- 128 MB less to read at C3, 0.65 GB at 5 M Gaussians -- and
// contracts them with dRGB as before.  Same expressions, same order as the reference-following backward used to evaluate
// them: bit-identical gradients (tests: the per-Gaussian stage against the CPU oracle, bit for bit).
// J = {dRGBdx[0..2], dRGBdy[0..2], dRGBdz[0..2]}; sh[(k) * 3 + ch]; C1, C2[5], C3[7] = the SH constants of auxiliary.h:22-38.
#define GSR_SHJAC_STREAM_P 2000000   // above this many Gaussians the nine floats bypass the caches (fwd store, bwd load)
// the backward's read of the nine floats of Gaussian idx
__device__ __forceinline__ void gs_load_shjac(const float* __restrict__ shjac, int P, int idx, float* J)
{
	if (P > GSR_SHJAC_STREAM_P) {
		asm volatile("" ::: "memory");   // (keeps the optimiser from merging the two arms into plain loads)
#pragma unroll
		for (int k = 0; k < 9; k++) J[k] = __builtin_nontemporal_load(shjac + 9 * (size_t)idx + k);
		asm volatile("" ::: "memory");
	} else {
#pragma unroll
		for (int k = 0; k < 9; k++) J[k] = shjac[9 * (size_t)idx + k];
	}
}
template <int D>
__device__ __forceinline__ void gs_sh_dir_jacobian(const float C1, const float* __restrict__ C2, const float* __restrict__ C3,
                                                   const float* sh, const float x, const float y, const float z, float* J)
{
	float* dRGBdx = J;
	float* dRGBdy = J + 3;
	float* dRGBdz = J + 6;
#pragma unroll
	for (int k = 0; k < 9; k++) J[k] = 0.f;
#define SH(k) sh[(k) * 3 + ch]
	if (D > 0) {
#pragma unroll
		for (int ch = 0; ch < 3; ch++) {
			dRGBdx[ch] = -C1 * SH(3);
			dRGBdy[ch] = -C1 * SH(1);
			dRGBdz[ch] = C1 * SH(2);
		}
		if (D > 1) {
			const float xx = x * x, yy = y * y, zz = z * z;
			const float xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
			for (int ch = 0; ch < 3; ch++) {
				dRGBdx[ch] += __builtin_fmaf(C2[4] * 2.f * x, SH(8), __builtin_fmaf(C2[3] * z, SH(7), __builtin_fmaf(C2[2] * 2.f * -x, SH(6), C2[0] * y * SH(4))));
				dRGBdy[ch] += __builtin_fmaf(C2[4] * 2.f * -y, SH(8), __builtin_fmaf(C2[2] * 2.f * -y, SH(6), __builtin_fmaf(C2[1] * z, SH(5), C2[0] * x * SH(4))));
				dRGBdz[ch] += __builtin_fmaf(C2[3] * x, SH(7), __builtin_fmaf(C2[2] * 2.f * 2.f * z, SH(6), C2[1] * y * SH(5)));
			}
			if (D > 2) {
#pragma unroll
				for (int ch = 0; ch < 3; ch++) {
					dRGBdx[ch] += __builtin_fmaf(C3[6] * SH(15) * 3.f, xx - yy,
					              __builtin_fmaf(C3[5] * SH(14) * 2.f, xz,
					              __builtin_fmaf(C3[4] * SH(13), __builtin_fmaf(4.f, zz, -3.f * xx) - yy,
					              __builtin_fmaf(C3[3] * SH(12) * -3.f * 2.f, xz,
					              __builtin_fmaf(C3[2] * SH(11) * -2.f, xy,
					              __builtin_fmaf(C3[1] * SH(10), yz, C3[0] * SH(9) * 3.f * 2.f * xy))))));
					dRGBdy[ch] += __builtin_fmaf(C3[6] * SH(15) * -3.f * 2.f, xy,
					              __builtin_fmaf(C3[5] * SH(14) * -2.f, yz,
					              __builtin_fmaf(C3[4] * SH(13) * -2.f, xy,
					              __builtin_fmaf(C3[3] * SH(12) * -3.f * 2.f, yz,
					              __builtin_fmaf(C3[2] * SH(11), __builtin_fmaf(4.f, zz, -3.f * yy) - xx,
					              __builtin_fmaf(C3[1] * SH(10), xz, C3[0] * SH(9) * 3.f * (xx - yy)))))));
					dRGBdz[ch] += __builtin_fmaf(C3[5] * SH(14), xx - yy,
					              __builtin_fmaf(C3[4] * SH(13) * 4.f * 2.f, xz,
					              __builtin_fmaf(C3[3] * SH(12) * 3.f, __builtin_fmaf(2.f, zz, -xx) - yy,
					              __builtin_fmaf(C3[2] * SH(11) * 4.f * 2.f, yz, C3[1] * SH(10) * xy))));
				}
			}
		}
	}
#undef SH
}

// ---- packed row messages of the multi-GPU exchange (layout: gsr_comm.hip) ----
__device__ __forceinline__ uint32_t gs_msg_nb(int P) { return (uint32_t)((P + 255) / 256); }
__device__ __forceinline__ uint32_t gs_msg_nw(int P) { return (uint32_t)((P + 31) / 32); }
// row of Gaussian idx in the message `msg` (false: not visible in that view, it has no row)
__device__ __forceinline__ bool gs_msg_lookup(const uint32_t* __restrict__ msg, int P, int idx, uint32_t& row)
{
	const uint32_t nb = gs_msg_nb(P);
	const uint32_t* mask = msg + 4 + nb;
	const uint32_t w = (uint32_t)idx >> 5, word = mask[w];
	if (!((word >> (idx & 31)) & 1u)) return false;
	uint32_t r = msg[4 + ((uint32_t)idx >> 8)];
	for (uint32_t k = ((uint32_t)idx >> 8) * 8; k < w; k++) r += (uint32_t)__popc(mask[k]);
	row = r + (uint32_t)__popc(word & ((1u << (idx & 31)) - 1u));
	return true;
}
