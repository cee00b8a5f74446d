// gsr_post.hip -- post-render epilogue kernels (the step right after the operator in
// gs-extract-mesh / gs-extract-pcd): depth map -> 3-D points, depth map -> normals.
// Replaces the ~15 torch ops of Camera.depth2point / Camera.depth2normal
// (gaustudio/datasets/__init__.py:106-112 ndc_2_cam, :307-339, :342-380) with two HBM-streaming kernels:
// 4 B read + 12 B written per pixel, no [H,W,3] intermediates.
#include "../../include/gsrast.h"
#include "gsr_internal.h"

namespace {

struct PostCam {
	float kinv[9];   // inverse intrinsics, row-major
	float c2w[12];   // rows 0..2 of the camera-to-world matrix
	int to_world;
};

// ndc_2_cam (datasets/__init__.py:106-112) on the grid of depth2point (:314-317): the reference forms
// x_ndc = i/(W-1), multiplies by (W-1) again and by z, then applies K^-1.
__device__ __forceinline__ float3 unproject(int x, int y, float z, int W, int H, const PostCam& c)
{
	const float wx = (float)(W - 1), hy = (float)(H - 1);
	const float xs = ((float)x / wx) * wx * z;
	const float ys = ((float)y / hy) * hy * z;
	float3 p;
	p.x = FMA(c.kinv[2], z, FMA(c.kinv[1], ys, c.kinv[0] * xs));
	p.y = FMA(c.kinv[5], z, FMA(c.kinv[4], ys, c.kinv[3] * xs));
	p.z = FMA(c.kinv[8], z, FMA(c.kinv[7], ys, c.kinv[6] * xs));
	return p;
}

__global__ __launch_bounds__(256) void depth_to_points_kernel(const float* __restrict__ depth, int W, int H, PostCam c,
                                                              float* __restrict__ points)
{
	const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (x >= W || y >= H) return;
	const size_t i = (size_t)y * W + x;
	float3 p = unproject(x, y, depth[i], W, H, c);
	if (c.to_world) {
		const float3 q = p;
		p.x = FMA(c.c2w[2], q.z, FMA(c.c2w[1], q.y, c.c2w[0] * q.x)) + c.c2w[3];
		p.y = FMA(c.c2w[6], q.z, FMA(c.c2w[5], q.y, c.c2w[4] * q.x)) + c.c2w[7];
		p.z = FMA(c.c2w[10], q.z, FMA(c.c2w[9], q.y, c.c2w[8] * q.x)) + c.c2w[11];
	}
	points[3 * i] = p.x; points[3 * i + 1] = p.y; points[3 * i + 2] = p.z;
}

// depth2normal (:342-380): five-tap cross product of the camera-space points, zero padding (so border pixels
// are invalid), validity d_min < z < d_max on all five taps, normalise with eps 1e-12, invalid -> (-1,-1,-1).
__device__ __forceinline__ void normal_from_taps(float zc, float3 pt, float3 pb, float3 pl, float3 pr, float d_min,
                                                 float d_max, const PostCam& c, float* __restrict__ out)
{
	const bool valid = (zc > d_min) & (zc < d_max) & (pt.z > d_min) & (pt.z < d_max) & (pb.z > d_min) & (pb.z < d_max) &
	                   (pl.z > d_min) & (pl.z < d_max) & (pr.z > d_min) & (pr.z < d_max);
	const float3 v = {pt.x - pb.x, pt.y - pb.y, pt.z - pb.z};   // vertical: top - bottom
	const float3 h = {pl.x - pr.x, pl.y - pr.y, pl.z - pr.z};   // horizontal: left - right
	float3 n = {-(v.y * h.z - v.z * h.y), -(v.z * h.x - v.x * h.z), -(v.x * h.y - v.y * h.x)};
	const float len = fmaxf(sqrtf(FMA(n.z, n.z, FMA(n.y, n.y, n.x * n.x))), 1e-12f);
	n.x /= len; n.y /= len; n.z /= len;
	if (c.to_world) {
		const float3 q = n;
		n.x = FMA(c.c2w[2], q.z, FMA(c.c2w[1], q.y, c.c2w[0] * q.x));
		n.y = FMA(c.c2w[6], q.z, FMA(c.c2w[5], q.y, c.c2w[4] * q.x));
		n.z = FMA(c.c2w[10], q.z, FMA(c.c2w[9], q.y, c.c2w[8] * q.x));
	}
	if (!valid) n.x = n.y = n.z = -1.f;
	out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

// Default path (tap distance k <= GSR_POST_KH, the reference's default k=3 gives 1): a 64x8-pixel block unprojects
// its (64+2k) x (8+2k) halo tile ONCE into LDS -- 1.3 unprojections per pixel instead of five -- and every pixel
// takes its taps from there.  Same arithmetic per point, same results.
#define GSR_POST_KH 2
#define GSR_POST_BW 64
#define GSR_POST_BH 8
// FUSED (gsr_depth_epilogue): the whole step between the render and the fusion in ONE pass over the frame -- the opacity
// mask of extract_mesh.py:104-107 (depth <- 0 where opacity < min_opacity) applied while the tile is read, the tile's own
// points written out as Camera.depth2point(..., coordinate) would (the masked pixels land on the camera centre), and the
// normals: 8 B read and 24 B written per pixel instead of three kernels and two [H,W] intermediates.
template <int k, bool FUSED>
__global__ __launch_bounds__(256) void depth_to_normals_tile_kernel(const float* __restrict__ depth, int W, int H, PostCam c,
                                                                    float d_min, float d_max, float* __restrict__ normals,
                                                                    const float* __restrict__ opacity, float min_opacity,
                                                                    float* __restrict__ points)
{
	constexpr int tw = GSR_POST_BW + 2 * k, th = GSR_POST_BH + 2 * k;
	__shared__ float s_x[th * tw], s_y[th * tw], s_z[th * tw];
	const int ox = blockIdx.x * GSR_POST_BW - k, oy = blockIdx.y * GSR_POST_BH - k;
	for (int t = threadIdx.x; t < tw * th; t += 256) {
		const int lx = t % tw, ly = t / tw, gx = ox + lx, gy = oy + ly;   // tw is a compile-time constant
		float3 p = {0.f, 0.f, 0.f};   // F.pad(..., value=0)
		if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
			float z = depth[(size_t)gy * W + gx];
			if (FUSED && opacity != nullptr && opacity[(size_t)gy * W + gx] < min_opacity) z = 0.f;
			p = unproject(gx, gy, z, W, H, c);
		}
		s_x[t] = p.x; s_y[t] = p.y; s_z[t] = p.z;
	}
	__syncthreads();
	const int lx = (threadIdx.x & 63) + k, x = ox + lx;
#pragma unroll
	for (int r = 0; r < GSR_POST_BH / 4; r++) {
		const int ly = (threadIdx.x >> 6) + 4 * r + k, y = oy + ly;
		if (x >= W || y >= H) continue;
		auto at = [&](int ax, int ay) { const int t = ay * tw + ax; return float3{s_x[t], s_y[t], s_z[t]}; };
		if (normals != nullptr)
			normal_from_taps(s_z[ly * tw + lx], at(lx, ly - k), at(lx, ly + k), at(lx - k, ly), at(lx + k, ly), d_min, d_max, c,
			                 normals + 3 * ((size_t)y * W + x));
		if (FUSED && points != nullptr) {
			float3 p = at(lx, ly);
			if (c.to_world) {   // depth_to_points_kernel's arithmetic
				const float3 q = p;
				p.x = FMA(c.c2w[2], q.z, FMA(c.c2w[1], q.y, c.c2w[0] * q.x)) + c.c2w[3];
				p.y = FMA(c.c2w[6], q.z, FMA(c.c2w[5], q.y, c.c2w[4] * q.x)) + c.c2w[7];
				p.z = FMA(c.c2w[10], q.z, FMA(c.c2w[9], q.y, c.c2w[8] * q.x)) + c.c2w[11];
			}
			float* o = points + 3 * ((size_t)y * W + x);
			o[0] = p.x; o[1] = p.y; o[2] = p.z;
		}
	}
}

// any tap distance: every pixel unprojects its own five taps
__global__ __launch_bounds__(256) void depth_to_normals_kernel(const float* __restrict__ depth, int W, int H, PostCam c,
                                                               int k, float d_min, float d_max,
                                                               float* __restrict__ normals)
{
	const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (x >= W || y >= H) return;
	auto tap = [&](int tx, int ty) {
		float3 p = {0.f, 0.f, 0.f};   // F.pad(..., value=0)
		if (tx >= 0 && tx < W && ty >= 0 && ty < H) p = unproject(tx, ty, depth[(size_t)ty * W + tx], W, H, c);
		return p;
	};
	normal_from_taps(tap(x, y).z, tap(x, y - k), tap(x, y + k), tap(x - k, y), tap(x + k, y), d_min, d_max, c,
	                 normals + 3 * ((size_t)y * W + x));
}

bool invert3(const float* m, float* out)
{
	const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
	const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
	if (det == 0.0) return false;
	const double r[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
	                     (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
	                     (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
	for (int k = 0; k < 9; k++) out[k] = (float)r[k];
	return true;
}

// inverse of a rigid-or-affine 4x4 [A t; 0 1] given row-major (world-to-camera): rows 0..2 of the inverse
bool invert_affine(const float* w2c, float* c2w12)
{
	const float A[9] = {w2c[0], w2c[1], w2c[2], w2c[4], w2c[5], w2c[6], w2c[8], w2c[9], w2c[10]};
	float Ai[9];
	if (!invert3(A, Ai)) return false;
	for (int r = 0; r < 3; r++) {
		for (int cc = 0; cc < 3; cc++) c2w12[4 * r + cc] = Ai[3 * r + cc];
		c2w12[4 * r + 3] = -(float)((double)Ai[3 * r] * w2c[3] + (double)Ai[3 * r + 1] * w2c[7] + (double)Ai[3 * r + 2] * w2c[11]);
	}
	return true;
}

int setup(const float* intrinsics, const float* world_to_camera, PostCam& c)
{
	if (!intrinsics) return GSR_ERR_ARG;
	if (!invert3(intrinsics, c.kinv)) return GSR_ERR_ARG;
	c.to_world = world_to_camera != nullptr;
	for (int k = 0; k < 12; k++) c.c2w[k] = 0.f;
	if (c.to_world && !invert_affine(world_to_camera, c.c2w)) return GSR_ERR_ARG;
	return GSR_OK;
}

// ---- masked bilateral filter of a depth map (gs-extract-pcd: extract_pcd.py:185-238 masked_bilateral_filter, which
// round-trips through numpy + cv2.dilate + cv2.bilateralFilter on the CPU).  Three small kernels on the stream:
//   1. new_mask = mask eroded by the d x d window (a pixel stays valid iff every in-image pixel of its window is valid:
//      1 - cv2.dilate(1 - mask, ones(d, d)), border pixels of the dilation ignored) + min / max of depth over new_mask;
//   2. bilateral filter (OpenCV's float32 algorithm: radius = max(d/2, 1), circular window r <= radius, BORDER_REFLECT_101,
//      weights exp(-r^2 / (2 sigma_space^2)) * exp(-dv^2 / (2 sigma_color^2))) of the depth NORMALISED to [0, 1] over the
//      valid region with everything outside it set to 0 -- the reference filters exactly that image, zeros included --
//      de-normalised, invalid pixels restored to their input depth.
// OpenCV evaluates the colour weight through a 4096-bin interpolated table; here it is expf directly (|dw| < 1e-7).
__device__ __forceinline__ uint32_t f2ord(float f)
{
	const uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // order-preserving float -> uint
}
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ __launch_bounds__(256) void bilateral_mask_minmax_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                                    int W, int H, int d, uint8_t* __restrict__ new_mask,
                                                                    uint32_t* __restrict__ mm)
{
	const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
	uint32_t lo = 0xffffffffu, hi = 0u;
	if (x < W && y < H) {
		const int r = d / 2;
		bool ok = true;
		for (int j = -r; j <= r; j++)
			for (int i = -r; i <= r; i++) {
				const int xx = x + i, yy = y + j;
				if (xx >= 0 && xx < W && yy >= 0 && yy < H && mask[(size_t)yy * W + xx] == 0) ok = false;
			}
		new_mask[(size_t)y * W + x] = ok ? 1 : 0;
		if (ok) lo = hi = f2ord(depth[(size_t)y * W + x]);
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
		hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
	}
	if ((threadIdx.x & 63) == 0 && hi != 0u) {
		atomicMin(&mm[0], lo);
		atomicMax(&mm[1], hi);
	}
}

__global__ __launch_bounds__(256) void bilateral_filter_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ new_mask,
                                                               int W, int H, int radius, float space_coeff, float color_coeff,
                                                               const uint32_t* __restrict__ mm, float* __restrict__ out)
{
	const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (x >= W || y >= H) return;
	const size_t p = (size_t)y * W + x;
	const float d0 = depth[p];
	if (mm[1] == 0u || !new_mask[p]) { out[p] = d0; return; }   // no valid pixel at all / invalid pixel: unchanged
	const float vmin = ord2f(mm[0]), vmax = ord2f(mm[1]);
	const float range = vmax - vmin;
	auto norm_at = [&](int xx, int yy) -> float {   // the image cv2.bilateralFilter is handed, BORDER_REFLECT_101
		xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
		yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
		xx = min(max(xx, 0), W - 1); yy = min(max(yy, 0), H - 1);   // (images narrower than the radius)
		const size_t q = (size_t)yy * W + xx;
		return new_mask[q] ? (depth[q] - vmin) / range : 0.f;
	};
	const float v0 = (d0 - vmin) / range;
	// cv2: a constant image is copied through (|max - min| < FLT_EPSILON of the NORMALISED image: never for range > 0,
	// and range == 0 makes the normalisation 0/0 -- the reference then returns NaNs; here the input depth)
	if (!(range > 0.f)) { out[p] = d0; return; }
	float sum = 0.f, wsum = 0.f;
	for (int j = -radius; j <= radius; j++)
		for (int i = -radius; i <= radius; i++) {
			const float r2 = (float)(i * i + j * j);
			if (r2 > (float)(radius * radius)) continue;   // circular window
			const float v = norm_at(x + i, y + j);
			const float dv = v - v0;
			const float w = __expf(r2 * space_coeff) * __expf(dv * dv * color_coeff);
			sum += v * w;
			wsum += w;
		}
	out[p] = (sum / wsum) * range + vmin;
}

}  // namespace

extern "C" {

int gsr_masked_bilateral(const float* depth, const unsigned char* mask, int width, int height, int d, float sigma_color,
                         float sigma_space, float* filtered, unsigned char* new_mask, unsigned int* scratch2, void* stream)
{
	if (!depth || !mask || !filtered || !new_mask || !scratch2 || width <= 0 || height <= 0) return GSR_ERR_ARG;
	if (d != 0 && (d < 1 || (d & 1) == 0)) return GSR_ERR_ARG;   // odd window sizes (cv2's even sizes shift the anchor)
	hipStream_t s = (hipStream_t)stream;
	if (sigma_color <= 0.f) sigma_color = 1.f;
	if (sigma_space <= 0.f) sigma_space = 1.f;
	int radius = d <= 0 ? (int)lrintf(sigma_space * 1.5f) : d / 2;
	if (radius < 1) radius = 1;
	const int dw = d <= 0 ? 2 * radius + 1 : d;   // the dilation window of the reference is d x d
	// min / max words: two 32-bit fills (a memcpy from a stack array would be a pageable, i.e. host-synchronous, H2D copy)
	if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scratch2), (int)0xffffffffu, 1, s) != hipSuccess) return GSR_ERR_HIP;
	if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scratch2 + 1), 0, 1, s) != hipSuccess) return GSR_ERR_HIP;
	const dim3 grid((width + 63) / 64, (height + 3) / 4);
	hipLaunchKernelGGL(bilateral_mask_minmax_kernel, grid, dim3(256), 0, s, depth, mask, width, height, dw, new_mask, scratch2);
	hipLaunchKernelGGL(bilateral_filter_kernel, grid, dim3(256), 0, s, depth, new_mask, width, height, radius,
	                   -0.5f / (sigma_space * sigma_space), -0.5f / (sigma_color * sigma_color), scratch2, filtered);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_depth_to_points(const float* depth, int width, int height, const float* intrinsics,
                        const float* world_to_camera, float* points, void* stream)
{
	if (!depth || !points || width <= 0 || height <= 0) return GSR_ERR_ARG;
	PostCam c;
	const int rc = setup(intrinsics, world_to_camera, c);
	if (rc != GSR_OK) return rc;
	hipLaunchKernelGGL(depth_to_points_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, (hipStream_t)stream,
	                   depth, width, height, c, points);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_depth_to_normals(const float* depth, int width, int height, const float* intrinsics, int k, float d_min,
                         float d_max, const float* world_to_camera, float* normals, void* stream)
{
	if (!depth || !normals || width <= 0 || height <= 0 || k < 1) return GSR_ERR_ARG;
	PostCam c;
	const int rc = setup(intrinsics, world_to_camera, c);
	if (rc != GSR_OK) return rc;
	const int kd = (k - 1) / 2;   // tap distance (depth2normal: k = (k - 1) // 2)
	const dim3 tgrid((width + GSR_POST_BW - 1) / GSR_POST_BW, (height + GSR_POST_BH - 1) / GSR_POST_BH);
	const float* no_op = nullptr;
	float* no_pts = nullptr;
	if (kd == 0)
		hipLaunchKernelGGL((depth_to_normals_tile_kernel<0, false>), tgrid, dim3(256), 0, (hipStream_t)stream, depth, width, height, c,
		                   d_min, d_max, normals, no_op, 0.f, no_pts);
	else if (kd == 1)
		hipLaunchKernelGGL((depth_to_normals_tile_kernel<1, false>), tgrid, dim3(256), 0, (hipStream_t)stream, depth, width, height, c,
		                   d_min, d_max, normals, no_op, 0.f, no_pts);
	else if (kd == 2)
		hipLaunchKernelGGL((depth_to_normals_tile_kernel<2, false>), tgrid, dim3(256), 0, (hipStream_t)stream, depth, width, height, c,
		                   d_min, d_max, normals, no_op, 0.f, no_pts);
	else
		hipLaunchKernelGGL(depth_to_normals_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, (hipStream_t)stream,
	                   depth, width, height, c, kd, d_min, d_max, normals);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_depth_epilogue(const float* depth, const float* opacity, float min_opacity, int width, int height,
                       const float* intrinsics, int k, float d_min, float d_max, const float* world_to_camera, float* points,
                       float* normals, void* stream)
{
	if (!depth || (!points && !normals) || width <= 0 || height <= 0 || k < 1) return GSR_ERR_ARG;
	const int kd = (k - 1) / 2;
	if (kd > GSR_POST_KH) return GSR_ERR_ARG;   // larger tap distances: the separate entry points
	PostCam c;
	const int rc = setup(intrinsics, world_to_camera, c);
	if (rc != GSR_OK) return rc;
	const dim3 tgrid((width + GSR_POST_BW - 1) / GSR_POST_BW, (height + GSR_POST_BH - 1) / GSR_POST_BH);
	if (kd == 0)
		hipLaunchKernelGGL((depth_to_normals_tile_kernel<0, true>), tgrid, dim3(256), 0, (hipStream_t)stream, depth, width, height, c,
		                   d_min, d_max, normals, opacity, min_opacity, points);
	else if (kd == 1)
		hipLaunchKernelGGL((depth_to_normals_tile_kernel<1, true>), tgrid, dim3(256), 0, (hipStream_t)stream, depth, width, height, c,
		                   d_min, d_max, normals, opacity, min_opacity, points);
	else
		hipLaunchKernelGGL((depth_to_normals_tile_kernel<2, true>), tgrid, dim3(256), 0, (hipStream_t)stream, depth, width, height, c,
		                   d_min, d_max, normals, opacity, min_opacity, points);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

}  // extern "C"
