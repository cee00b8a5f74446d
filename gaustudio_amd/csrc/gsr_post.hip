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
// The grid factors (i/(W-1))*(W-1) and (j/(H-1))*(H-1) of the taps are IEEE divisions that depend on the column /
// row only: a block computes the 64+2k columns and 4+2k rows it needs once into LDS (ten divisions per pixel
// made the kernel VALU-bound: 66 us at 4K; the remaining three are the normalisation).
#define GSR_POST_KMAX 32
__global__ __launch_bounds__(256) void depth_to_normals_kernel(const float* __restrict__ depth, int W, int H, PostCam c,
                                                               int k, float d_min, float d_max,
                                                               float* __restrict__ normals)
{
	__shared__ float s_col[64 + 2 * GSR_POST_KMAX], s_row[4 + 2 * GSR_POST_KMAX];
	const int bx0 = blockIdx.x * 64 - k, by0 = blockIdx.y * 4 - k;
	const float wx = (float)(W - 1), hy = (float)(H - 1);
	const bool tab = k <= GSR_POST_KMAX;
	if (tab) {
		for (int t = threadIdx.x; t < 64 + 2 * k; t += 256) s_col[t] = ((float)(bx0 + t) / wx) * wx;
		for (int t = threadIdx.x; t < 4 + 2 * k; t += 256) s_row[t] = ((float)(by0 + t) / hy) * hy;
	}
	__syncthreads();
	const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (x >= W || y >= H) return;
	const size_t i = (size_t)y * W + x;
	auto tap = [&](int tx, int ty, bool& valid) {
		float3 p = {0.f, 0.f, 0.f};   // F.pad(..., value=0)
		if (tx >= 0 && tx < W && ty >= 0 && ty < H) {
			const float z = depth[(size_t)ty * W + tx];
			if (tab) {
				const float xs = s_col[tx - bx0] * z, ys = s_row[ty - by0] * z;   // same products as unproject()
				p.x = FMA(c.kinv[2], z, FMA(c.kinv[1], ys, c.kinv[0] * xs));
				p.y = FMA(c.kinv[5], z, FMA(c.kinv[4], ys, c.kinv[3] * xs));
				p.z = FMA(c.kinv[8], z, FMA(c.kinv[7], ys, c.kinv[6] * xs));
			} else {
				p = unproject(tx, ty, z, W, H, c);
			}
		}
		valid = valid && (p.z > d_min) && (p.z < d_max);
		return p;
	};
	bool valid = true;
	const float3 pc = tap(x, y, valid);
	(void)pc;
	const float3 pt = tap(x, y - k, valid), pb = tap(x, y + k, valid);
	const float3 pl = tap(x - k, y, valid), pr = tap(x + k, y, valid);
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
	normals[3 * i] = n.x; normals[3 * i + 1] = n.y; normals[3 * i + 2] = n.z;
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

}  // namespace

extern "C" {

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
	hipLaunchKernelGGL(depth_to_normals_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, (hipStream_t)stream,
	                   depth, width, height, c, (k - 1) / 2, d_min, d_max, normals);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

}  // extern "C"
