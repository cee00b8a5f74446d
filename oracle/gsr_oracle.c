/*
 * gsr_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never on the product path).
 *
 * A plain-C float32 restatement of the reference rasterizer
 *   $RAST = /root/reference/submodules/gaustudio-diff-gaussian-rasterization
 *   $RAST/cuda_rasterizer/{auxiliary.h,forward.cu,backward.cu,rasterizer_impl.cu}
 * Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY PIN: the reference ships no golden vectors / tests.  This oracle is
 * pinned against (1) the reference's own CUDA kernels, hipified test-only into oracle/_ref/
 * (oracle/build_ref.sh) and run on an MI355X -- fixtures in tests/golden/ref_*.npz; (2) the
 * reference's Python SH / covariance helpers (gaustudio/utils/sh_utils.py, gaustudio/models/utils.py)
 * -- fixtures tests/golden/py_*.npz; (3) a float64 torch autograd restatement (oracle/torch_f64.py).
 *
 * Floating-point contract ("pinned contraction").  The reference is built by nvcc with its default
 * --fmad=true, i.e. mul+add chains are contracted to FMA at the compiler's discretion; no CPU
 * restatement can be bit-identical to that binary.  This file pins ONE contraction, compiled with
 * -ffp-contract=off so that only the fmaf() written below fuses:
 *   * a left-to-right sum of products fuses every product after the first into the running sum
 *     through its final multiplication:  a*b + c*d + e*f  ->  fma(e,f, fma(c,d, a*b));
 *   * product +/- scalar fuses:  a*b + c -> fma(a,b,c);  c - a*b -> fma(-a,b,c).
 * Divisions, sqrt are IEEE correctly rounded.  exp() is the explicit polynomial gs_exp() below
 * (max rel. error 2.9e-7 on [-5.6,0] vs CUDA expf's 2 ulp) so that the HIP kernels can reproduce
 * the forward pass BIT-EXACTLY.  Two re-associations are made in the compositing loops and are
 * documented where they occur (weight = alpha*T formed once).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16   /* config.h:16 */
#define BLOCK_Y 16   /* config.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* auxiliary.h:22-38 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

#define FMA(a, b, c) fmaf((a), (b), (c))

typedef struct { float x, y, z; } f3;
typedef struct { float m[3][3]; } mat3; /* glm convention: m[col][row] */

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* exp() used by both compositing passes (forward.cu:345, backward.cu:531).  Explicit, reproducible:
 * exp(p) = 2^(p*log2e) ; n = round-to-nearest-even(p*log2e) through the 1.5*2^23 magic constant;
 * f = p*log2e - n (single rounding) ; 2^f by a degree-5 minimax polynomial ; scale by 2^n through
 * the exponent bits.  Defined as 0 below -80 (alpha is then < 1/255 for any sane opacity). */
float orc_exp(float p)
{
	if (p < -80.0f) return 0.0f;
	const float LOG2E = 0x1.715476p+0f; /* 1.4426950216293335 */
	const float MAGIC = 12582912.0f;    /* 1.5 * 2^23 */
	float tm = FMA(p, LOG2E, MAGIC);
	float nf = tm - MAGIC;
	float f = FMA(p, LOG2E, -nf);
	float y = 0x1.5c08e6p-10f;
	y = FMA(y, f, 0x1.3d0c52p-7f);
	y = FMA(y, f, 0x1.c6b6e4p-5f);
	y = FMA(y, f, 0x1.ebf918p-3f);
	y = FMA(y, f, 0x1.62e428p-1f);
	y = FMA(y, f, 0x1.000002p+0f);
	uint32_t yb, tb;
	memcpy(&yb, &y, 4);
	memcpy(&tb, &tm, 4);
	yb += tb << 23;
	memcpy(&y, &yb, 4);
	return y;
}

/* auxiliary.h:41-44 -- evaluated in double */
static inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56 */
static inline void getRect(float px, float py, int max_radius, int gx, int gy, int* rminx, int* rminy,
                           int* rmaxx, int* rmaxy)
{
	*rminx = imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
	*rminy = imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
	*rmaxx = imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
	*rmaxy = imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* auxiliary.h:58-66 */
static inline f3 transformPoint4x3(f3 p, const float* m)
{
	f3 r;
	r.x = FMA(m[8], p.z, FMA(m[4], p.y, m[0] * p.x)) + m[12];
	r.y = FMA(m[9], p.z, FMA(m[5], p.y, m[1] * p.x)) + m[13];
	r.z = FMA(m[10], p.z, FMA(m[6], p.y, m[2] * p.x)) + m[14];
	return r;
}
/* auxiliary.h:68-77 */
static inline void transformPoint4x4(f3 p, const float* m, float out[4])
{
	out[0] = FMA(m[8], p.z, FMA(m[4], p.y, m[0] * p.x)) + m[12];
	out[1] = FMA(m[9], p.z, FMA(m[5], p.y, m[1] * p.x)) + m[13];
	out[2] = FMA(m[10], p.z, FMA(m[6], p.y, m[2] * p.x)) + m[14];
	out[3] = FMA(m[11], p.z, FMA(m[7], p.y, m[3] * p.x)) + m[15];
}
/* auxiliary.h:89-97 */
static inline f3 transformVec4x3Transpose(f3 p, const float* m)
{
	f3 r;
	r.x = FMA(m[2], p.z, FMA(m[1], p.y, m[0] * p.x));
	r.y = FMA(m[6], p.z, FMA(m[5], p.y, m[4] * p.x));
	r.z = FMA(m[10], p.z, FMA(m[9], p.y, m[8] * p.x));
	return r;
}
/* auxiliary.h:107-117 */
static inline f3 dnormvdv3(f3 v, f3 dv)
{
	float sum2 = FMA(v.z, v.z, FMA(v.y, v.y, v.x * v.x));
	float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	f3 r;
	r.x = (FMA(-(v.z * v.x), dv.z, FMA(-(v.y * v.x), dv.y, FMA(-v.x, v.x, sum2) * dv.x))) * invsum32;
	r.y = (FMA(-(v.z * v.y), dv.z, FMA(FMA(-v.y, v.y, sum2), dv.y, (-v.x * v.y) * dv.x))) * invsum32;
	r.z = (FMA(FMA(-v.z, v.z, sum2), dv.z, FMA(-(v.y * v.z), dv.y, (-v.x * v.z) * dv.x))) * invsum32;
	return r;
}

/* glm mat3 * mat3 (glm/detail/type_mat3x3.inl operator*): Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] */
static inline mat3 mat3_mul(const mat3* A, const mat3* B)
{
	mat3 R;
	for (int c = 0; c < 3; c++)
		for (int r = 0; r < 3; r++)
			R.m[c][r] = FMA(A->m[2][r], B->m[c][2], FMA(A->m[1][r], B->m[c][1], A->m[0][r] * B->m[c][0]));
	return R;
}
static inline mat3 mat3_transpose(const mat3* A)
{
	mat3 R;
	for (int c = 0; c < 3; c++)
		for (int r = 0; r < 3; r++) R.m[c][r] = A->m[r][c];
	return R;
}

/* rotation matrix of forward.cu:134-138 / backward.cu:287-291, glm column fill; q NOT normalised */
static inline mat3 quat_to_R(const float* q)
{
	float r = q[0], x = q[1], y = q[2], z = q[3];
	mat3 R;
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

/* forward.cu:118-152 computeCov3D */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
	mat3 S;
	memset(&S, 0, sizeof(S));
	S.m[0][0] = mod * scale[0];
	S.m[1][1] = mod * scale[1];
	S.m[2][2] = mod * scale[2];
	mat3 R = quat_to_R(rot);
	mat3 M = mat3_mul(&S, &R);
	mat3 Mt = mat3_transpose(&M);
	mat3 Sigma = mat3_mul(&Mt, &M);
	cov3D[0] = Sigma.m[0][0];
	cov3D[1] = Sigma.m[0][1];
	cov3D[2] = Sigma.m[0][2];
	cov3D[3] = Sigma.m[1][1];
	cov3D[4] = Sigma.m[1][2];
	cov3D[5] = Sigma.m[2][2];
}

/* shared by forward.cu:74-113 (computeCov2D) and backward.cu:159-201: builds t (clamped), J, W, T, Vrk, cov2D */
typedef struct {
	f3 t;
	float txtz, tytz, limx, limy;
	mat3 J, W, T, Vrk, cov;
} cov2d_ctx;

static void cov2d_common(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         const float* cov3D, const float* view, cov2d_ctx* c)
{
	f3 t = transformPoint4x3(mean, view);
	c->limx = 1.3f * tan_fovx;
	c->limy = 1.3f * tan_fovy;
	c->txtz = t.x / t.z;
	c->tytz = t.y / t.z;
	t.x = fminf_(c->limx, fmaxf_(-c->limx, c->txtz)) * t.z;
	t.y = fminf_(c->limy, fmaxf_(-c->limy, c->tytz)) * t.z;
	c->t = t;
	mat3 J;
	J.m[0][0] = focal_x / t.z; J.m[0][1] = 0.0f; J.m[0][2] = -(focal_x * t.x) / (t.z * t.z);
	J.m[1][0] = 0.0f; J.m[1][1] = focal_y / t.z; J.m[1][2] = -(focal_y * t.y) / (t.z * t.z);
	J.m[2][0] = 0; J.m[2][1] = 0; J.m[2][2] = 0;
	mat3 W;
	W.m[0][0] = view[0]; W.m[0][1] = view[4]; W.m[0][2] = view[8];
	W.m[1][0] = view[1]; W.m[1][1] = view[5]; W.m[1][2] = view[9];
	W.m[2][0] = view[2]; W.m[2][1] = view[6]; W.m[2][2] = view[10];
	mat3 Vrk;
	Vrk.m[0][0] = cov3D[0]; Vrk.m[0][1] = cov3D[1]; Vrk.m[0][2] = cov3D[2];
	Vrk.m[1][0] = cov3D[1]; Vrk.m[1][1] = cov3D[3]; Vrk.m[1][2] = cov3D[4];
	Vrk.m[2][0] = cov3D[2]; Vrk.m[2][1] = cov3D[4]; Vrk.m[2][2] = cov3D[5];
	mat3 T = mat3_mul(&W, &J);
	mat3 Tt = mat3_transpose(&T);
	mat3 Vt = mat3_transpose(&Vrk);
	mat3 A = mat3_mul(&Tt, &Vt);
	mat3 cov = mat3_mul(&A, &T);
	c->J = J; c->W = W; c->T = T; c->Vrk = Vrk; c->cov = cov;
}

/* forward.cu:20-71 computeColorFromSH */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped, float* out_rgb)
{
	f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	f3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
	float len = sqrtf(FMA(dir.z, dir.z, FMA(dir.y, dir.y, dir.x * dir.x)));
	dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
	const float* sh = shs + (size_t)idx * max_coeffs * 3;
	float res[3];
	float x = dir.x, y = dir.y, z = dir.z;
	for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
		float r = SH_C0 * SH(0);
		if (deg > 0) {
			r = FMA(-(SH_C1 * y), SH(1), r);
			r = FMA(SH_C1 * z, SH(2), r);
			r = FMA(-(SH_C1 * x), SH(3), r);
			if (deg > 1) {
				float xx = x * x, yy = y * y, zz = z * z;
				float xy = x * y, yz = y * z, xz = x * z;
				r = FMA(SH_C2[0] * xy, SH(4), r);
				r = FMA(SH_C2[1] * yz, SH(5), r);
				r = FMA(SH_C2[2] * (FMA(2.0f, zz, -xx) - yy), SH(6), r);
				r = FMA(SH_C2[3] * xz, SH(7), r);
				r = FMA(SH_C2[4] * (xx - yy), SH(8), r);
				if (deg > 2) {
					r = FMA(SH_C3[0] * y * FMA(3.0f, xx, -yy), SH(9), r);
					r = FMA(SH_C3[1] * xy * z, SH(10), r);
					r = FMA(SH_C3[2] * y * (FMA(4.0f, zz, -xx) - yy), SH(11), r);
					r = FMA(SH_C3[3] * z * FMA(-3.0f, yy, FMA(-3.0f, xx, 2.0f * zz)), SH(12), r);
					r = FMA(SH_C3[4] * x * (FMA(4.0f, zz, -xx) - yy), SH(13), r);
					r = FMA(SH_C3[5] * z * (xx - yy), SH(14), r);
					r = FMA(SH_C3[6] * x * FMA(-3.0f, yy, xx), SH(15), r);
				}
			}
		}
#undef SH
		r += 0.5f;
		clamped[3 * idx + ch] = (r < 0);
		res[ch] = fmaxf_(r, 0.0f);
	}
	out_rgb[0] = res[0]; out_rgb[1] = res[1]; out_rgb[2] = res[2];
}

/* ------------------------------------------------------------------------------------------------
 * forward.cu:155-256 preprocessCUDA (+ auxiliary.h:139-164 in_frustum).  Returns 0, or -1 when a
 * point is culled although prefiltered is set (the reference printf+__trap()s, auxiliary.h:156-160).
 * Output arrays are caller-owned: radii[P], means2D[2P], depths[P], cov3D[6P], rgb[3P],
 * conic_opacity[4P], tiles_touched[P], clamped[3P].  Entries of culled Gaussians are left untouched
 * except radii/tiles_touched = 0 (forward.cu:190-191).
 * ---------------------------------------------------------------------------------------------- */
int orc_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                   const float* rotations, const float* opacities, const float* shs,
                   const float* cov3D_precomp, const float* colors_precomp, const float* view,
                   const float* proj, const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                   int prefiltered, int* radii, float* means2D, float* depths, float* cov3Ds, float* rgb,
                   float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped)
{
	const float focal_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:225-226 */
	const float focal_x = W / (2.0f * tan_fovx);
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	int err = 0;
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++) {
		radii[idx] = 0;
		tiles_touched[idx] = 0;
		f3 p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
		float p_hom[4];
		transformPoint4x4(p_orig, proj, p_hom);
		float p_w = 1.0f / (p_hom[3] + 0.0000001f);
		float p_proj_x = p_hom[0] * p_w, p_proj_y = p_hom[1] * p_w;
		f3 p_view = transformPoint4x3(p_orig, view);
		if (p_view.z <= 0.2f) {
			if (prefiltered) err = -1;
			continue;
		}
		const float* cov3D;
		if (cov3D_precomp != NULL) {
			cov3D = cov3D_precomp + (size_t)idx * 6;
		} else {
			computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx,
			             cov3Ds + (size_t)idx * 6);
			cov3D = cov3Ds + (size_t)idx * 6;
		}
		cov2d_ctx c;
		cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, &c);
		float cov_x = c.cov.m[0][0] + 0.3f; /* forward.cu:110-112 */
		float cov_y = c.cov.m[0][1];
		float cov_z = c.cov.m[1][1] + 0.3f;
		float det = FMA(-cov_y, cov_y, cov_x * cov_z);
		if (det == 0.0f) continue;
		float det_inv = 1.f / det;
		float conic_x = cov_z * det_inv, conic_y = -cov_y * det_inv, conic_z = cov_x * det_inv;
		float mid = 0.5f * (cov_x + cov_z);
		float disc = sqrtf(fmaxf_(0.1f, FMA(mid, mid, -det)));
		float lambda1 = mid + disc;
		float lambda2 = mid - disc;
		float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
		float pix_x = ndc2Pix(p_proj_x, W), pix_y = ndc2Pix(p_proj_y, H);
		int rminx, rminy, rmaxx, rmaxy;
		getRect(pix_x, pix_y, (int)my_radius, gx, gy, &rminx, &rminy, &rmaxx, &rmaxy);
		if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
		if (colors_precomp == NULL)
			computeColorFromSH(idx, D, M, means3D, campos, shs, clamped, rgb + 3 * (size_t)idx);
		depths[idx] = p_view.z;
		radii[idx] = (int)my_radius;
		means2D[2 * idx] = pix_x;
		means2D[2 * idx + 1] = pix_y;
		conic_opacity[4 * idx + 0] = conic_x;
		conic_opacity[4 * idx + 1] = conic_y;
		conic_opacity[4 * idx + 2] = conic_z;
		conic_opacity[4 * idx + 3] = opacities[idx];
		tiles_touched[idx] = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
	}
	return err;
}

/* rasterizer_impl.cu:54-66 checkFrustum / :141-153 markVisible */
void orc_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present)
{
	(void)proj;
	for (int idx = 0; idx < P; idx++) {
		f3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
		f3 pv = transformPoint4x3(p, view);
		present[idx] = !(pv.z <= 0.2f);
	}
}

/* ------------------------------------------------------------------------------------------------
 * Binning: rasterizer_impl.cu:70-111 duplicateWithKeys, :303-311 radix sort on (tile<<32 | depth bits),
 * :116-138 identifyTileRanges.  CUB's sort is stable and instances are emitted in ascending Gaussian
 * id, so ties on (tile, depth) resolve by ascending id.  Restated as a comparison sort
 * on (tile, depth bits, id).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t key; uint32_t id; } kv_t;
static int kv_cmp(const void* a, const void* b)
{
	const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
	if (x->key != y->key) return x->key < y->key ? -1 : 1;
	if (x->id != y->id) return x->id < y->id ? -1 : 1;
	return 0;
}

int64_t orc_count_rendered(int P, const uint32_t* tiles_touched)
{
	int64_t R = 0;
	for (int i = 0; i < P; i++) R += tiles_touched[i];
	return R;
}

/* ------------------------------------------------------------------------------------------------
 * NOT in the reference: the product bins each Gaussian into a TIGHT sub-rect of the reference's getRect square
 * (gaustudio_amd/csrc/gsr_common.h gs_tight_rect: tiles that cannot hold a pixel with alpha >= 1/255 are left
 * out).  The oracle restates that function operation for operation (IEEE +,*,/,sqrt,fma only; explicit ternaries)
 * so that tests can (1) prove on the CPU that the images of the tight and of the reference binning are
 * bit-identical and the tight list is a sub-sequence of the reference list holding every contributor, and
 * (2) compare the product's lists / ranges / n_contrib bit for bit against the oracle run on the same rects.
 * rects[4P] = (rminx, rminy, rmaxx, rmaxy), all 0 for Gaussians binned nowhere.
 * ---------------------------------------------------------------------------------------------- */
static inline float orc_log(float x)
{
	uint32_t xb;
	memcpy(&xb, &x, 4);
	const int e = (int)((xb >> 23) & 0xff) - 127;
	uint32_t mb = (xb & 0x007fffffu) | 0x3f800000u;
	float m;
	memcpy(&m, &mb, 4);
	const float s = (m - 1.0f) / (m + 1.0f);
	const float s2 = s * s;
	float p = 0.11111111f;
	p = FMA(p, s2, 0.14285715f);
	p = FMA(p, s2, 0.2f);
	p = FMA(p, s2, 0.33333334f);
	p = FMA(p, s2, 1.0f);
	return FMA((float)e, 0.69314718f, (2.0f * s) * p);
}

static int tight_rect(float px, float py, float ca, float cb, float cc, float op, int gx, int gy, int* rminx,
                      int* rminy, int* rmaxx, int* rmaxy, float* qmax)
{
	*qmax = -1.0f;
	if (op <= 0.0f) return 0;
	const float t = orc_log(255.0f * op) + 0.01f;
	if (t <= 0.0f) return 0;
	const float det = FMA(-cb, cb, ca * cc);
	if (!(det > 0.0f && ca > 0.0f && cc > 0.0f)) return 1;
	const float rel = (ca * cc) / det;
	if (!(rel < 1.0e5f)) return 1;
	const float teff = FMA(t * 4.0e-5f, rel, t) + 0.05f;
	if (!(teff > 0.0f)) return 1;
	*qmax = 2.0f * teff;
	const float ex = sqrtf((2.0f * teff) * (cc / det)) + 0.01f;
	const float ey = sqrtf((2.0f * teff) * (ca / det)) + 0.01f;
	float fx0 = ceilf((px - ex - 15.0f) * 0.0625f), fx1 = floorf((px + ex) * 0.0625f) + 1.0f;
	float fy0 = ceilf((py - ey - 15.0f) * 0.0625f), fy1 = floorf((py + ey) * 0.0625f) + 1.0f;
	const float fgx = (float)gx, fgy = (float)gy;
	fx0 = fx0 > 0.0f ? fx0 : 0.0f; fx0 = fx0 < fgx ? fx0 : fgx;
	fx1 = fx1 > 0.0f ? fx1 : 0.0f; fx1 = fx1 < fgx ? fx1 : fgx;
	fy0 = fy0 > 0.0f ? fy0 : 0.0f; fy0 = fy0 < fgy ? fy0 : fgy;
	fy1 = fy1 > 0.0f ? fy1 : 0.0f; fy1 = fy1 < fgy ? fy1 : fgy;
	*rminx = imax(*rminx, (int)fx0); *rmaxx = imin(*rmaxx, (int)fx1);
	*rminy = imax(*rminy, (int)fy0); *rmaxy = imin(*rmaxy, (int)fy1);
	return *rmaxx > *rminx && *rmaxy > *rminy;
}

/* gsr_common.h gs_tile_may_touch, operation for operation: can a pixel centre of tile (tx, ty) lie inside the ellipse
 * a dx^2 + 2 b dx dy + c dy^2 <= qmax?  (The product does not bin the corner tiles of a tight rect for which this says no.) */
static int tile_may_touch(float px, float py, float ca, float cb, float cc, float qmax, int tx, int ty, int W, int H)
{
	const float bx0 = (float)(16 * tx), by0 = (float)(16 * ty);
	float bx1 = bx0 + 15.0f, by1 = by0 + 15.0f;
	const float wm = (float)(W - 1), hm = (float)(H - 1);
	bx1 = bx1 < wm ? bx1 : wm;
	by1 = by1 < hm ? by1 : hm;
	const float X0 = px - bx1, X1 = px - bx0;
	const float Y0 = py - by1, Y1 = py - by0;
	const float xn = X0 > 0.0f ? X0 : (X1 < 0.0f ? X1 : 0.0f);
	const float yn = Y0 > 0.0f ? Y0 : (Y1 < 0.0f ? Y1 : 0.0f);
	if (xn == 0.0f && yn == 0.0f) return 1;
	float best = 3.0e38f, mag = 0.0f;
	if (xn != 0.0f) {
		float dy = -(cb * xn) / cc;
		dy = dy < Y0 ? Y0 : dy;
		dy = dy > Y1 ? Y1 : dy;
		const float t0 = (ca * xn) * xn, t1 = ((2.0f * cb) * xn) * dy, t2 = (cc * dy) * dy;
		best = (t0 + t2) + t1;
		mag = (t0 + t2) + (t1 < 0.0f ? -t1 : t1);
	}
	if (yn != 0.0f) {
		float dx = -(cb * yn) / ca;
		dx = dx < X0 ? X0 : dx;
		dx = dx > X1 ? X1 : dx;
		const float t0 = (cc * yn) * yn, t1 = ((2.0f * cb) * yn) * dx, t2 = (ca * dx) * dx;
		const float q = (t0 + t2) + t1, m = (t0 + t2) + (t1 < 0.0f ? -t1 : t1);
		if (q < best) { best = q; mag = m; }
	}
	return !(best > qmax + (1.0e-5f * mag + 0.05f));
}

/* gs_dead_corners: bits 0..3 = top-left, top-right, bottom-left, bottom-right tile of the rect not binned */
static uint32_t dead_corners(float px, float py, float ca, float cb, float cc, float qmax, const int* r, int W, int H)
{
	if (!(qmax > 0.0f) || r[2] - r[0] < 2 || r[3] - r[1] < 2) return 0u;
	uint32_t dead = 0u;
	if (!tile_may_touch(px, py, ca, cb, cc, qmax, r[0], r[1], W, H)) dead |= 1u;
	if (!tile_may_touch(px, py, ca, cb, cc, qmax, r[2] - 1, r[1], W, H)) dead |= 2u;
	if (!tile_may_touch(px, py, ca, cb, cc, qmax, r[0], r[3] - 1, W, H)) dead |= 4u;
	if (!tile_may_touch(px, py, ca, cb, cc, qmax, r[2] - 1, r[3] - 1, W, H)) dead |= 8u;
	return dead;
}

static inline int corner_is_dead(const int32_t* r, uint32_t dead, int x, int y)
{
	if (dead == 0u) return 0;
	if (!((x == r[0] || x == r[2] - 1) && (y == r[1] || y == r[3] - 1))) return 0;
	return (int)((dead >> ((x == r[0] ? 0 : 1) + (y == r[1] ? 0 : 2))) & 1u);
}

/* tight != 0: gs_tight_rect + gs_dead_corners (dead[P], may be NULL when tight == 0); tight == 0: the reference's getRect
 * squares.  Returns the number of instances. */
int64_t orc_rects(int P, int W, int H, const float* means2D, const float* conic_opacity, const int* radii, int tight,
                  int32_t* rects, uint32_t* tiles, uint8_t* dead)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	int64_t R = 0;
	for (int idx = 0; idx < P; idx++) {
		int r[4] = {0, 0, 0, 0};
		uint32_t dc = 0u;
		if (radii[idx] > 0) {
			getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &r[0], &r[1], &r[2], &r[3]);
			const float* co = conic_opacity + 4 * (size_t)idx;
			float qmax = -1.0f;
			if (tight && !tight_rect(means2D[2 * idx], means2D[2 * idx + 1], co[0], co[1], co[2], co[3], gx, gy, &r[0],
			                         &r[1], &r[2], &r[3], &qmax))
				r[0] = r[1] = r[2] = r[3] = 0;
			if (tight) dc = dead_corners(means2D[2 * idx], means2D[2 * idx + 1], co[0], co[1], co[2], qmax, r, W, H);
		}
		if (dead) dead[idx] = (uint8_t)dc;
		for (int k = 0; k < 4; k++) rects[4 * (size_t)idx + k] = r[k];
		const uint32_t n = (uint32_t)((r[2] - r[0]) * (r[3] - r[1])) - (uint32_t)__builtin_popcount(dc);
		if (tiles) tiles[idx] = n;
		R += n;
	}
	return R;
}

/* orc_bin_sort over explicit rects and their dead corners (see orc_rects; dead may be NULL) */
void orc_bin_sort_rects(int P, int W, int H, const float* depths, const int32_t* rects, const uint8_t* dead, int64_t R,
                        uint32_t* point_list, uint32_t* ranges)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(R > 0 ? R : 1));
	int64_t off = 0;
	for (int idx = 0; idx < P; idx++) {
		const int32_t* r = rects + 4 * (size_t)idx;
		uint32_t dbits;
		memcpy(&dbits, &depths[idx], 4);
		for (int y = r[1]; y < r[3]; y++)
			for (int x = r[0]; x < r[2]; x++) {
				if (dead && corner_is_dead(r, dead[idx], x, y)) continue;
				uint64_t key = (uint64_t)(y * gx + x);
				key <<= 32;
				key |= dbits;
				kv[off].key = key;
				kv[off].id = (uint32_t)idx;
				off++;
			}
	}
	qsort(kv, (size_t)R, sizeof(kv_t), kv_cmp);
	memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
	for (int64_t i = 0; i < R; i++) {
		point_list[i] = kv[i].id;
		uint32_t currtile = (uint32_t)(kv[i].key >> 32);
		if (i == 0) ranges[2 * currtile] = 0;
		else {
			uint32_t prevtile = (uint32_t)(kv[i - 1].key >> 32);
			if (currtile != prevtile) {
				ranges[2 * prevtile + 1] = (uint32_t)i;
				ranges[2 * currtile] = (uint32_t)i;
			}
		}
		if (i == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
	}
	free(kv);
}

/* point_list[R], ranges[2T] (start,end) ; tiles with no instance keep (0,0) (rasterizer_impl.cu:313) */
void orc_bin_sort(int P, int W, int H, const float* means2D, const float* depths, const int* radii,
                  int64_t R, uint32_t* point_list, uint32_t* ranges)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(R > 0 ? R : 1));
	int64_t off = 0;
	for (int idx = 0; idx < P; idx++) {
		if (radii[idx] > 0) {
			int rminx, rminy, rmaxx, rmaxy;
			getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &rminx, &rminy, &rmaxx, &rmaxy);
			uint32_t dbits;
			memcpy(&dbits, &depths[idx], 4);
			for (int y = rminy; y < rmaxy; y++)
				for (int x = rminx; x < rmaxx; x++) {
					uint64_t key = (uint64_t)(y * gx + x);
					key <<= 32;
					key |= dbits;
					kv[off].key = key;
					kv[off].id = (uint32_t)idx;
					off++;
				}
		}
	}
	qsort(kv, (size_t)R, sizeof(kv_t), kv_cmp);
	memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
	for (int64_t i = 0; i < R; i++) {
		point_list[i] = kv[i].id;
		uint32_t currtile = (uint32_t)(kv[i].key >> 32);
		if (i == 0) ranges[2 * currtile] = 0;
		else {
			uint32_t prevtile = (uint32_t)(kv[i - 1].key >> 32);
			if (currtile != prevtile) {
				ranges[2 * prevtile + 1] = (uint32_t)i;
				ranges[2 * currtile] = (uint32_t)i;
			}
		}
		if (i == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
	}
	free(kv);
}

/* ------------------------------------------------------------------------------------------------
 * forward.cu:261-397 renderCUDA.  One "thread" per pixel, sequential over the tile's sorted list.
 * Quirks kept: no background blend in out_color (Q1), contributor counting (Q8), depth / median
 * (Q9), CHW planes (Q10).  final_T / n_contrib are pixel-major [H*W].
 * Re-association (documented): weight w = alpha*T is formed once and colour/depth accumulate as
 * fma(feature, w, acc); the reference writes features*alpha*T (forward.cu:365-366), <= 1 ulp/term.
 * tile_step > 1 renders only tiles with (tile_id % tile_step == 0) (bench sampling); other pixels untouched.
 * ---------------------------------------------------------------------------------------------- */
void orc_composite_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                       const float* means2D, const float* features, const float* depths,
                       const float* conic_opacity, float* out_color, float* out_depth,
                       float* out_median, float* out_opacity, float* final_T, uint32_t* n_contrib,
                       int tile_step)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const size_t HW = (size_t)H * W;
	if (tile_step < 1) tile_step = 1;
#pragma omp parallel for schedule(dynamic, 1)
	for (int tile = 0; tile < gx * gy; tile += tile_step) {
		const int tx = tile % gx, ty = tile / gx;
		const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
		for (int ly = 0; ly < BLOCK_Y; ly++)
			for (int lx = 0; lx < BLOCK_X; lx++) {
				const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue;
				const size_t pix_id = (size_t)W * py + px;
				const float pixfx = (float)px, pixfy = (float)py;
				float T = 1.0f;
				uint32_t contributor = 0, last_contributor = 0;
				float C0 = 0, C1 = 0, C2 = 0, Dacc = 0;
				float median_D = 15.0f, median_weight = 0, median_id = 0;
				for (uint32_t k = r0; k < r1; k++) {
					contributor++;
					const uint32_t id = point_list[k];
					const float dx = means2D[2 * id] - pixfx, dy = means2D[2 * id + 1] - pixfy;
					const float* co = conic_opacity + 4 * (size_t)id;
					/* forward.cu:338 power = -0.5f*(a dx dx + c dy dy) - b dx dy */
					const float power = FMA(-(co[1] * dx), dy, -0.5f * FMA(co[2] * dy, dy, (co[0] * dx) * dx));
					if (power > 0.0f) continue;
					const float alpha = fminf_(0.99f, co[3] * orc_exp(power));
					if (alpha < 1.0f / 255.0f) continue;
					const float test_T = T * (1 - alpha);
					if (test_T < 0.0001f) break; /* done = true (forward.cu:357-361) */
					const float w = alpha * T;
					C0 = FMA(features[3 * (size_t)id + 0], w, C0);
					C1 = FMA(features[3 * (size_t)id + 1], w, C1);
					C2 = FMA(features[3 * (size_t)id + 2], w, C2);
					Dacc = FMA(depths[id], w, Dacc);
					if (T > 0.5f && test_T < 0.5f) {
						median_D = depths[id];
						median_weight = w;
						median_id = (float)id; /* int -> float, forward.cu:372 */
					}
					T = test_T;
					last_contributor = contributor;
				}
				final_T[pix_id] = T;
				n_contrib[pix_id] = last_contributor;
				out_color[0 * HW + pix_id] = C0;
				out_color[1 * HW + pix_id] = C1;
				out_color[2 * HW + pix_id] = C2;
				out_depth[pix_id] = Dacc;
				out_median[pix_id] = median_D;
				out_median[HW + pix_id] = median_weight;
				out_median[2 * HW + pix_id] = median_id;
				out_opacity[pix_id] = 1 - T;
			}
	}
}

/* ------------------------------------------------------------------------------------------------
 * backward.cu:415-610 renderCUDA (backward).  Per pixel, back-to-front.  The reference accumulates
 * the per-Gaussian sums with float atomicAdd in a nondeterministic order; here they are summed in
 * DOUBLE (acc[10P]) together with the sum of |contribution| (accabs[10P]) so that tests can bound a
 * float32 any-order sum rigorously.  Component order in acc: 0,1 dL_dmean2D.xy; 2,3,4 dL_dconic
 * (a,b,c) ; 5 dL_dopacity ; 6,7,8 dL_dcolor ; 9 dL_ddepth.
 * ---------------------------------------------------------------------------------------------- */
void orc_composite_bwd(int W, int H, const float* bg, const uint32_t* ranges, const uint32_t* point_list,
                       const float* means2D, const float* conic_opacity, const float* colors,
                       const float* depths, const float* final_Ts, const uint32_t* n_contrib,
                       const float* dL_dpixels, const float* dL_dpixel_depths,
                       const float* dL_dpixel_median_depths, const float* dL_dpixel_final_opacitys,
                       double* acc, double* accabs, double* flip9, int tile_step)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	const size_t HW = (size_t)H * W;
	const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H); /* backward.cu:493-494 */
	if (tile_step < 1) tile_step = 1;
#define ACC(id, k, v)                                        \
	do {                                                      \
		double v_ = (double)(v);                              \
		_Pragma("omp atomic") acc[10 * (size_t)(id) + (k)] += v_; \
		if (accabs) { _Pragma("omp atomic") accabs[10 * (size_t)(id) + (k)] += fabs(v_); } \
	} while (0)
/* a term whose value v comes out of a cancellation: `m` is the magnitude of what was subtracted (>= |v|) and is
 * what any differently-rounded evaluation is accurate relative to */
#define ACCM(id, k, v, m)                                    \
	do {                                                      \
		double v_ = (double)(v);                              \
		_Pragma("omp atomic") acc[10 * (size_t)(id) + (k)] += v_; \
		if (accabs) { _Pragma("omp atomic") accabs[10 * (size_t)(id) + (k)] += fabs((double)(m)); } \
	} while (0)
#pragma omp parallel for schedule(dynamic, 1)
	for (int tile = 0; tile < gx * gy; tile += tile_step) {
		const int tx = tile % gx, ty = tile / gx;
		const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
		for (int ly = 0; ly < BLOCK_Y; ly++)
			for (int lx = 0; lx < BLOCK_X; lx++) {
				const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
				if (!(px < W && py < H)) continue;
				const size_t pix_id = (size_t)W * py + px;
				const float pixfx = (float)px, pixfy = (float)py;
				const float T_final = final_Ts[pix_id];
				float T = T_final;
				uint32_t contributor = r1 - r0;
				const uint32_t last_contributor = n_contrib[pix_id];
				float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
				float dL_dpixel[3] = {dL_dpixels[pix_id], dL_dpixels[HW + pix_id], dL_dpixels[2 * HW + pix_id]};
				const float dL_dpixel_depth = dL_dpixel_depths[pix_id];
				const float dL_dpixel_median_depth = dL_dpixel_median_depths[pix_id]; /* channel 0 only (Q15) */
				const float dL_dpixel_final_opacity = dL_dpixel_final_opacitys[pix_id];
				float accum_depth_rec = 0, accum_final_opacity_rec = 0;
				float last_alpha = 0, last_depth = 0, last_final_opacity = 0;
				/* bg . dL_dpixel is loop invariant (backward.cu:584-586) */
				float bg_dot_dpixel = 0;
				for (int i = 0; i < 3; i++) bg_dot_dpixel = FMA(bg[i], dL_dpixel[i], bg_dot_dpixel);
				for (uint32_t kk = r1; kk-- > r0;) {
					contributor--;
					if (contributor >= last_contributor) continue;
					const uint32_t id = point_list[kk];
					const float dx = means2D[2 * id] - pixfx, dy = means2D[2 * id + 1] - pixfy;
					const float* co = conic_opacity + 4 * (size_t)id;
					const float power = FMA(-(co[1] * dx), dy, -0.5f * FMA(co[2] * dy, dy, (co[0] * dx) * dx));
					if (power > 0.0f) continue;
					const float G = orc_exp(power);
					const float alpha = fminf_(0.99f, co[3] * G);
					if (alpha < 1.0f / 255.0f) continue;
					const float test_T = T / (1.f - alpha);
					const float w = alpha * test_T; /* dchannel_dcolor = dpixel_depth_ddepth = dpixel_opacity_dopacity */
					float dL_dalpha = 0.0f;
					double mag = 0.0; /* sum of the magnitudes dL_dalpha is a (possibly cancelling) combination of */
					const float one_m_la = 1.f - last_alpha;
					for (int ch = 0; ch < 3; ch++) {
						const float c = colors[3 * (size_t)id + ch];
						accum_rec[ch] = FMA(last_alpha, last_color[ch], one_m_la * accum_rec[ch]);
						last_color[ch] = c;
						const float dL_dchannel = dL_dpixel[ch];
						dL_dalpha = FMA(c - accum_rec[ch], dL_dchannel, dL_dalpha);
						mag += ((double)fabsf(c) + fabsf(accum_rec[ch])) * fabsf(dL_dchannel);
						ACC(id, 6 + ch, w * dL_dchannel);
					}
					const float c_d = depths[id];
					accum_depth_rec = FMA(last_alpha, last_depth, one_m_la * accum_depth_rec);
					last_depth = c_d;
					dL_dalpha = FMA(c_d - accum_depth_rec, dL_dpixel_depth, dL_dalpha);
					mag += ((double)fabsf(c_d) + fabsf(accum_depth_rec)) * fabsf(dL_dpixel_depth);
					ACC(id, 9, w * dL_dpixel_depth);
					if (test_T > 0.5f && T < 0.5f) ACC(id, 9, dL_dpixel_median_depth);
					/* The median test runs on a T RECONSTRUCTED by division (backward.cu:536,566): when it lands within
					 * rounding distance of 0.5 its outcome is ill-conditioned -- in the reference as well, whose backward can
					 * disagree with its own forward there.  flip9[id] collects |dL_dmedian| of such events so that a
					 * comparison can grant exactly that much slack to component 9 of the Gaussian. */
					if (flip9 && (fabsf(test_T - 0.5f) < 2e-5f || fabsf(T - 0.5f) < 2e-5f)) {
						const double v_ = fabs((double)dL_dpixel_median_depth);
						_Pragma("omp atomic") flip9[id] += v_;
					}
					accum_final_opacity_rec = FMA(last_alpha, last_final_opacity, one_m_la * accum_final_opacity_rec);
					last_final_opacity = 1.f;
					dL_dalpha = FMA(1.f - accum_final_opacity_rec, dL_dpixel_final_opacity, dL_dalpha);
					mag += (1.0 + fabsf(accum_final_opacity_rec)) * fabsf(dL_dpixel_final_opacity);
					ACC(id, 5, w * dL_dpixel_final_opacity); /* Q14 extra opacity term */
					dL_dalpha *= test_T;
					mag *= test_T;
					T = test_T;
					last_alpha = alpha;
					dL_dalpha = FMA(-T_final / (1.f - alpha), bg_dot_dpixel, dL_dalpha);
					mag += fabs((double)(T_final / (1.f - alpha)) * bg_dot_dpixel);
					const float dL_dG = co[3] * dL_dalpha;
					const double mG = fabsf(co[3]) * mag; /* magnitude behind dL_dG */
					const float gdx = G * dx, gdy = G * dy;
					const float dG_ddelx = FMA(-gdy, co[1], -gdx * co[0]);
					const float dG_ddely = FMA(-gdx, co[1], -gdy * co[2]);
					const double mx_ = (fabs((double)gdy * co[1]) + fabs((double)gdx * co[0])) * ddelx_dx;
					const double my_ = (fabs((double)gdx * co[1]) + fabs((double)gdy * co[2])) * ddely_dy;
					ACCM(id, 0, dL_dG * dG_ddelx * ddelx_dx, mG * mx_);
					ACCM(id, 1, dL_dG * dG_ddely * ddely_dy, mG * my_);
					ACCM(id, 2, -0.5f * gdx * dx * dL_dG, 0.5 * fabs((double)gdx * dx) * mG);
					ACCM(id, 3, -0.5f * gdx * dy * dL_dG, 0.5 * fabs((double)gdx * dy) * mG);
					ACCM(id, 4, -0.5f * gdy * dy * dL_dG, 0.5 * fabs((double)gdy * dy) * mG);
					ACCM(id, 5, G * dL_dalpha, G * mag);
				}
			}
	}
#undef ACCM
#undef ACC
}

/* ------------------------------------------------------------------------------------------------
 * backward.cu:144-274 computeCov2DCUDA, then backward.cu:346-412 preprocessCUDA (with :20-139 SH
 * backward and :278-341 cov3D backward), run per Gaussian in that order (backward.cu:641,658).
 * Inputs: dL_dmean2D[3P] (xy used), dL_dconic[4P] (indices 0,1,3 used, Q16), dL_dcolor[3P], dL_ddepth[P].
 * Outputs (caller zero-initialised, as rasterize_points.cu:160-169): dL_dmeans3D[3P], dL_dcov3D[6P],
 * dL_dsh[3MP], dL_dscale[3P], dL_drot[4P].
 * ---------------------------------------------------------------------------------------------- */
void orc_preprocess_bwd(int P, int D, int M, const float* means3D, const int* radii, const float* shs,
                        const uint8_t* clamped, const float* scales, const float* rotations,
                        float scale_modifier, const float* cov3Ds, const float* view, const float* proj,
                        int W, int H, float tan_fovx, float tan_fovy, const float* campos,
                        const float* dL_dmean2D, const float* dL_dconics, float* dL_dmeans,
                        const float* dL_dcolor, const float* dL_ddepth, float* dL_dcov, float* dL_dsh,
                        float* dL_dscale, float* dL_drot)
{
	const float h_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:391-392 */
	const float h_x = W / (2.0f * tan_fovx);
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++) {
		if (!(radii[idx] > 0)) continue;
		/* ---- computeCov2DCUDA ---- */
		const float* cov3D = cov3Ds + 6 * (size_t)idx;
		f3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
		const float dLc_x = dL_dconics[4 * idx], dLc_y = dL_dconics[4 * idx + 1], dLc_z = dL_dconics[4 * idx + 3];
		cov2d_ctx c;
		cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, &c);
		const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
		const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
		const float a = c.cov.m[0][0] + 0.3f, b = c.cov.m[0][1], cc = c.cov.m[1][1] + 0.3f;
		const float denom = FMA(-b, b, a * cc);
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		const float denom2inv = 1.0f / FMA(denom, denom, 0.0000001f);
#define T_(i, j) c.T.m[i][j]
#define V_(i, j) c.Vrk.m[i][j]
#define W_(i, j) c.W.m[i][j]
		float* dcov = dL_dcov + 6 * (size_t)idx;
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
		} else {
			for (int i = 0; i < 6; i++) dcov[i] = 0;
		}
		/* backward.cu:230-241: row k of (T Vrk) products */
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
		const float dL_dtz = FMA((2 * h_y * c.t.y) * tz3, dL_dJ12, FMA((2 * h_x * c.t.x) * tz3, dL_dJ02,
		                         FMA(-(h_y * tz2), dL_dJ11, -h_x * tz2 * dL_dJ00)));
		f3 dLt = {dL_dtx, dL_dty, dL_dtz};
		f3 dmean = transformVec4x3Transpose(dLt, view); /* dL_dmeans[idx] = ... (assignment, Q17) */

		/* ---- preprocessCUDA (backward.cu:346-412) ---- */
		f3 m = mean;
		float m_hom[4];
		transformPoint4x4(m, proj, m_hom);
		const float m_w = 1.0f / (m_hom[3] + 0.0000001f);
		const float mul1 = (FMA(proj[8], m.z, FMA(proj[4], m.y, proj[0] * m.x)) + proj[12]) * m_w * m_w;
		const float mul2 = (FMA(proj[9], m.z, FMA(proj[5], m.y, proj[1] * m.x)) + proj[13]) * m_w * m_w;
		const float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
		f3 d1;
		d1.x = FMA(FMA(-proj[3], mul2, proj[1] * m_w), g2y, FMA(-proj[3], mul1, proj[0] * m_w) * g2x);
		d1.y = FMA(FMA(-proj[7], mul2, proj[5] * m_w), g2y, FMA(-proj[7], mul1, proj[4] * m_w) * g2x);
		d1.z = FMA(FMA(-proj[11], mul2, proj[9] * m_w), g2y, FMA(-proj[11], mul1, proj[8] * m_w) * g2x);
		dmean.x += d1.x; dmean.y += d1.y; dmean.z += d1.z;
		const float mul3 = FMA(view[10], m.z, FMA(view[6], m.y, view[2] * m.x)) + view[14];
		const float gd = dL_ddepth[idx];
		dmean.x += FMA(-view[3], mul3, view[2]) * gd;
		dmean.y += FMA(-view[7], mul3, view[6]) * gd;
		dmean.z += FMA(-view[11], mul3, view[10]) * gd;

		if (shs) {
			/* backward.cu:20-139 computeColorFromSH (backward) */
			f3 dir_orig = {m.x - campos[0], m.y - campos[1], m.z - campos[2]};
			float len = sqrtf(FMA(dir_orig.z, dir_orig.z, FMA(dir_orig.y, dir_orig.y, dir_orig.x * dir_orig.x)));
			float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
			const float* sh = shs + (size_t)idx * M * 3;
			float* dsh = dL_dsh + (size_t)idx * M * 3;
			float dRGB[3];
			for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0.f : 1.f);
			float ddir[3] = {0, 0, 0}; /* dL_ddir = (dot(dRGBdx,dL_dRGB), ...) accumulated channel by channel */
			float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(k) sh[(k) * 3 + ch]
#define DSH(k, coef) for (int ch = 0; ch < 3; ch++) dsh[(k) * 3 + ch] = (coef) * dRGB[ch]
			DSH(0, SH_C0);
			if (D > 0) {
				const float d1_ = -SH_C1 * y, d2_ = SH_C1 * z, d3_ = -SH_C1 * x;
				DSH(1, d1_); DSH(2, d2_); DSH(3, d3_);
				for (int ch = 0; ch < 3; ch++) {
					dRGBdx[ch] = -SH_C1 * SH(3);
					dRGBdy[ch] = -SH_C1 * SH(1);
					dRGBdz[ch] = SH_C1 * SH(2);
				}
				if (D > 1) {
					const float xx = x * x, yy = y * y, zz = z * z;
					const float xy = x * y, yz = y * z, xz = x * z;
					const float d4_ = SH_C2[0] * xy, d5_ = SH_C2[1] * yz;
					const float d6_ = SH_C2[2] * (FMA(2.f, zz, -xx) - yy);
					const float d7_ = SH_C2[3] * xz, d8_ = SH_C2[4] * (xx - yy);
					DSH(4, d4_); DSH(5, d5_); DSH(6, d6_); DSH(7, d7_); DSH(8, d8_);
					for (int ch = 0; ch < 3; ch++) {
						dRGBdx[ch] += FMA(SH_C2[4] * 2.f * x, SH(8), FMA(SH_C2[3] * z, SH(7), FMA(SH_C2[2] * 2.f * -x, SH(6), SH_C2[0] * y * SH(4))));
						dRGBdy[ch] += FMA(SH_C2[4] * 2.f * -y, SH(8), FMA(SH_C2[2] * 2.f * -y, SH(6), FMA(SH_C2[1] * z, SH(5), SH_C2[0] * x * SH(4))));
						dRGBdz[ch] += FMA(SH_C2[3] * x, SH(7), FMA(SH_C2[2] * 2.f * 2.f * z, SH(6), SH_C2[1] * y * SH(5)));
					}
					if (D > 2) {
						const float d9_ = SH_C3[0] * y * FMA(3.f, xx, -yy);
						const float d10_ = SH_C3[1] * xy * z;
						const float d11_ = SH_C3[2] * y * (FMA(4.f, zz, -xx) - yy);
						const float d12_ = SH_C3[3] * z * FMA(-3.f, yy, FMA(-3.f, xx, 2.f * zz));
						const float d13_ = SH_C3[4] * x * (FMA(4.f, zz, -xx) - yy);
						const float d14_ = SH_C3[5] * z * (xx - yy);
						const float d15_ = SH_C3[6] * x * FMA(-3.f, yy, xx);
						DSH(9, d9_); DSH(10, d10_); DSH(11, d11_); DSH(12, d12_); DSH(13, d13_); DSH(14, d14_); DSH(15, d15_);
						for (int ch = 0; ch < 3; ch++) {
							dRGBdx[ch] += FMA(SH_C3[6] * SH(15) * 3.f, xx - yy,
							              FMA(SH_C3[5] * SH(14) * 2.f, xz,
							              FMA(SH_C3[4] * SH(13), FMA(4.f, zz, -3.f * xx) - yy,
							              FMA(SH_C3[3] * SH(12) * -3.f * 2.f, xz,
							              FMA(SH_C3[2] * SH(11) * -2.f, xy,
							              FMA(SH_C3[1] * SH(10), yz, SH_C3[0] * SH(9) * 3.f * 2.f * xy))))));
							dRGBdy[ch] += FMA(SH_C3[6] * SH(15) * -3.f * 2.f, xy,
							              FMA(SH_C3[5] * SH(14) * -2.f, yz,
							              FMA(SH_C3[4] * SH(13) * -2.f, xy,
							              FMA(SH_C3[3] * SH(12) * -3.f * 2.f, yz,
							              FMA(SH_C3[2] * SH(11), FMA(4.f, zz, -3.f * yy) - xx,
							              FMA(SH_C3[1] * SH(10), xz, SH_C3[0] * SH(9) * 3.f * (xx - yy)))))));
							dRGBdz[ch] += FMA(SH_C3[5] * SH(14), xx - yy,
							              FMA(SH_C3[4] * SH(13) * 4.f * 2.f, xz,
							              FMA(SH_C3[3] * SH(12) * 3.f, FMA(2.f, zz, -xx) - yy,
							              FMA(SH_C3[2] * SH(11) * 4.f * 2.f, yz, SH_C3[1] * SH(10) * xy))));
						}
					}
				}
			}
#undef SH
#undef DSH
			ddir[0] = FMA(dRGBdx[2], dRGB[2], FMA(dRGBdx[1], dRGB[1], dRGBdx[0] * dRGB[0]));
			ddir[1] = FMA(dRGBdy[2], dRGB[2], FMA(dRGBdy[1], dRGB[1], dRGBdy[0] * dRGB[0]));
			ddir[2] = FMA(dRGBdz[2], dRGB[2], FMA(dRGBdz[1], dRGB[1], dRGBdz[0] * dRGB[0]));
			f3 dd = {ddir[0], ddir[1], ddir[2]};
			f3 dm = dnormvdv3(dir_orig, dd);
			dmean.x += dm.x; dmean.y += dm.y; dmean.z += dm.z;
		}
		dL_dmeans[3 * idx] = dmean.x; dL_dmeans[3 * idx + 1] = dmean.y; dL_dmeans[3 * idx + 2] = dmean.z;

		if (scales) {
			/* backward.cu:278-341 computeCov3D (backward) */
			const float* q = rotations + 4 * (size_t)idx;
			const float r = q[0], x = q[1], y = q[2], z = q[3];
			mat3 R = quat_to_R(q);
			float s[3] = {scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1], scale_modifier * scales[3 * idx + 2]};
			mat3 S; memset(&S, 0, sizeof(S));
			S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
			mat3 Mm = mat3_mul(&S, &R);
			const float* g = dcov;
			mat3 dSig;
			dSig.m[0][0] = g[0]; dSig.m[0][1] = 0.5f * g[1]; dSig.m[0][2] = 0.5f * g[2];
			dSig.m[1][0] = 0.5f * g[1]; dSig.m[1][1] = g[3]; dSig.m[1][2] = 0.5f * g[4];
			dSig.m[2][0] = 0.5f * g[2]; dSig.m[2][1] = 0.5f * g[4]; dSig.m[2][2] = g[5];
			mat3 M2; /* 2.0f * M */
			for (int cI = 0; cI < 3; cI++) for (int rI = 0; rI < 3; rI++) M2.m[cI][rI] = 2.0f * Mm.m[cI][rI];
			mat3 dL_dM = mat3_mul(&M2, &dSig);
			mat3 Rt = mat3_transpose(&R);
			mat3 dMt = mat3_transpose(&dL_dM);
			for (int i = 0; i < 3; i++)
				dL_dscale[3 * idx + i] = FMA(Rt.m[i][2], dMt.m[i][2], FMA(Rt.m[i][1], dMt.m[i][1], Rt.m[i][0] * dMt.m[i][0]));
			for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) dMt.m[i][j] *= s[i];
#define D_(i, j) dMt.m[i][j]
			float* dq = dL_drot + 4 * (size_t)idx;
			dq[0] = FMA(2 * x, D_(1, 2) - D_(2, 1), FMA(2 * y, D_(2, 0) - D_(0, 2), 2 * z * (D_(0, 1) - D_(1, 0))));
			dq[1] = FMA(-4 * x, D_(2, 2) + D_(1, 1), FMA(2 * r, D_(1, 2) - D_(2, 1), FMA(2 * z, D_(2, 0) + D_(0, 2), 2 * y * (D_(1, 0) + D_(0, 1)))));
			dq[2] = FMA(-4 * y, D_(2, 2) + D_(0, 0), FMA(2 * z, D_(1, 2) + D_(2, 1), FMA(2 * r, D_(2, 0) - D_(0, 2), 2 * x * (D_(1, 0) + D_(0, 1)))));
			dq[3] = FMA(-4 * z, D_(1, 1) + D_(0, 0), FMA(2 * y, D_(1, 2) + D_(2, 1), FMA(2 * x, D_(2, 0) + D_(0, 2), 2 * r * (D_(0, 1) - D_(1, 0)))));
#undef D_
		}
	}
}

int orc_num_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
	omp_set_num_threads(n);
#else
	(void)n;
#endif
}
