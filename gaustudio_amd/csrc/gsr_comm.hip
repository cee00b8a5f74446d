// gsr_comm.hip -- row compaction for the multi-GPU gradient exchange (gaustudio_amd/parallel.py FactoredGradExchange,
// compact="view"; new design: the reference has no multi-GPU path).
//
// A view of a real capture sees a fraction of the scene (cameras INSIDE a 360-degree scene: 14-19 % of the Gaussians per
// view, 52-59 % in the union of 8 views -- tools/comm_model.py), and the gradient rows of a Gaussian that is culled in a
// view are exactly zero.  Instead of all-gathering P x 12 B of colour gradients per view, a rank sends a MESSAGE:
//
//   word 0        K = number of visible Gaussians (radii > 0) of the view          words 1..3 reserved
//   words 4 ..    nb = ceil(P / 256) block bases: visible Gaussians in front of each 256-Gaussian block (exclusive scan)
//   then          nw = ceil(P / 32) mask words: bit (g & 31) of word (g >> 5) = Gaussian g is visible
//   then (16-B aligned) K rows of C floats, in ascending Gaussian order
//
// The header depends on `radii` only, so it is built right after the FORWARD (gsr_visible_index), off the critical path;
// the rows are packed when the backward has produced them (gsr_pack_rows).  A receiver finds Gaussian g's row at
// base[g >> 8] + popcount(mask bits of the block below g): the SH-gradient rebuild reads packed messages directly
// (gsr_sh_grad_from_packed, gsr_kernels_bwd.hip), nothing is expanded.  The OR of all views' masks (gsr_union_index) is the
// set of rows with a non-zero geometry gradient anywhere in the step: the geometry all-reduce runs on those rows only.
#include "../../include/gsrast.h"
#include "gsr_internal.h"

#include <hip/hip_runtime.h>

namespace gsr {


// one workgroup per 256 Gaussians: mask words + the block's count
__global__ __launch_bounds__(256) void visible_mask_kernel(int P, const int* __restrict__ radii, uint32_t* __restrict__ msg,
                                                           uint32_t* __restrict__ counts)
{
	__shared__ uint32_t s_cnt[4];
	const int idx = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const bool vis = idx < P && radii[idx] > 0;
	const unsigned long long bm = __ballot(vis);
	uint32_t* mask = msg + 4 + gs_msg_nb(P);
	const uint32_t w0 = (uint32_t)(blockIdx.x * 8 + wv * 2);
	if (lane == 0) {
		if (w0 < gs_msg_nw(P)) mask[w0] = (uint32_t)bm;
		if (w0 + 1 < gs_msg_nw(P)) mask[w0 + 1] = (uint32_t)(bm >> 32);
		s_cnt[wv] = (uint32_t)__popcll(bm);
	}
	__syncthreads();
	if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
// the same from the OR of N messages' masks (the union of the views)
__global__ __launch_bounds__(256) void union_mask_kernel(int P, int N, const uint32_t* __restrict__ msgs,
                                                         const unsigned long long* __restrict__ msg_off,
                                                         uint32_t* __restrict__ out, uint32_t* __restrict__ counts)
{
	__shared__ uint32_t s_cnt[8];
	const uint32_t nb = gs_msg_nb(P), nw = gs_msg_nw(P);
	const uint32_t w = blockIdx.x * 8 + threadIdx.x;          // 8 mask words per 256-Gaussian block, threads 0..7
	uint32_t m = 0;
	if (threadIdx.x < 8 && w < nw)
		for (int r = 0; r < N; r++) m |= msgs[msg_off[r] + 4 + nb + w];
	if (threadIdx.x < 8) {
		if (w < nw) out[4 + nb + w] = m;
		s_cnt[threadIdx.x] = (uint32_t)__popc(m);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t c = 0;
		for (int k = 0; k < 8; k++) c += s_cnt[k];
		counts[blockIdx.x] = c;
	}
}
// exclusive scan of the nb block counts (one workgroup) -> bases in the header, K in word 0
__global__ __launch_bounds__(1024) void block_base_kernel(int P, const uint32_t* __restrict__ counts, uint32_t* __restrict__ msg)
{
	__shared__ uint32_t s_part[1024];
	const uint32_t nb = gs_msg_nb(P);
	const int tid = threadIdx.x;
	const uint32_t per = (nb + 1023u) / 1024u;
	const uint32_t b0 = (uint32_t)tid * per, b1 = min(nb, b0 + per);
	uint32_t sum = 0;
	for (uint32_t b = b0; b < b1; b++) sum += counts[b];
	s_part[tid] = sum;
	__syncthreads();
	if (tid == 0) {
		uint32_t run = 0;
		for (int t = 0; t < 1024; t++) { const uint32_t c = s_part[t]; s_part[t] = run; run += c; }
		msg[0] = run; msg[1] = (uint32_t)P; msg[2] = 0u; msg[3] = 0u;
	}
	__syncthreads();
	uint32_t run = s_part[tid];
	for (uint32_t b = b0; b < b1; b++) { msg[4 + b] = run; run += counts[b]; }
}

// rows of the visible Gaussians, in order: out[row * stride + col0 + c] = in[idx * C + c]
__global__ __launch_bounds__(256) void pack_rows_kernel(int P, int C, const uint32_t* __restrict__ hdr, const float* __restrict__ in,
                                                        float* __restrict__ out, int stride, int col0)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	uint32_t row;
	if (!gs_msg_lookup(hdr, P, idx, row)) return;
	for (int c = 0; c < C; c++) out[(size_t)row * stride + col0 + c] = in[(size_t)idx * C + c];
}
__global__ __launch_bounds__(256) void unpack_rows_kernel(int P, int C, const uint32_t* __restrict__ hdr, const float* __restrict__ in,
                                                          int stride, int col0, float* __restrict__ out)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	uint32_t row;
	if (!gs_msg_lookup(hdr, P, idx, row)) return;
	for (int c = 0; c < C; c++) out[(size_t)idx * C + c] = in[(size_t)row * stride + col0 + c];
}

// The geometry block of the exchange -- means3D 3 | opacity 1 | scales 3 | rotations 4 = 11 floats per Gaussian, four tensors -- in
// ONE pass (round 5: four pack_rows launches with strided columns, four unpack launches and a torch-level check cost 0.45 ms per
// step at 1 M Gaussians, 0.77 ms at 5 M: more than the wire time the compaction saves).  pack: row r of `rows` = the 11 floats of
// the r-th Gaussian of the header; a Gaussian OUTSIDE the header whose 11 floats are not all zero raises *flag (a loss term
// other than the rasterizer put a gradient on a Gaussian no view of the step sees: the caller then sums the dense block).
__global__ __launch_bounds__(256) void pack_geometry_kernel(int P, const uint32_t* __restrict__ hdr, const float* __restrict__ g_mean,
                                                            const float* __restrict__ g_op, const float* __restrict__ g_scale,
                                                            const float* __restrict__ g_rot, float* __restrict__ rows,
                                                            uint32_t* __restrict__ flag)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	float v[11];
#pragma unroll
	for (int c = 0; c < 3; c++) { v[c] = g_mean[3 * (size_t)idx + c]; v[4 + c] = g_scale[3 * (size_t)idx + c]; }
	v[3] = g_op[idx];
#pragma unroll
	for (int c = 0; c < 4; c++) v[7 + c] = g_rot[4 * (size_t)idx + c];   // (the block's rotations start at float 7 P: 16-B aligned only if P % 4 == 0)
	uint32_t row;
	if (gs_msg_lookup(hdr, P, idx, row)) {
#pragma unroll
		for (int c = 0; c < 11; c++) rows[(size_t)row * 11 + c] = v[c];
	} else {
		bool any = false;
#pragma unroll
		for (int c = 0; c < 11; c++) any |= v[c] != 0.0f;
		if (any) atomicOr(flag, 1u);
	}
}
__global__ __launch_bounds__(256) void unpack_geometry_kernel(int P, const uint32_t* __restrict__ hdr, const float* __restrict__ rows,
                                                              float* __restrict__ g_mean, float* __restrict__ g_op,
                                                              float* __restrict__ g_scale, float* __restrict__ g_rot)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	uint32_t row;
	if (!gs_msg_lookup(hdr, P, idx, row)) return;
	const float* v = rows + (size_t)row * 11;
#pragma unroll
	for (int c = 0; c < 3; c++) { g_mean[3 * (size_t)idx + c] = v[c]; g_scale[3 * (size_t)idx + c] = v[4 + c]; }
	g_op[idx] = v[3];
#pragma unroll
	for (int c = 0; c < 4; c++) g_rot[4 * (size_t)idx + c] = v[7 + c];
}

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_msg_header_words(int P)
{
	const size_t w = 4 + (size_t)((P + 255) / 256) + (size_t)((P + 31) / 32);
	return (w + 3) / 4 * 4;
}

int gsr_visible_index(int P, const int* radii, uint32_t* msg, uint32_t* scratch_counts, void* stream)
{
	if (P <= 0) return GSR_OK;
	if (!radii || !msg || !scratch_counts) return GSR_ERR_ARG;
	hipStream_t s = (hipStream_t)stream;
	const int nb = (P + 255) / 256;
	hipLaunchKernelGGL(visible_mask_kernel, dim3(nb), dim3(256), 0, s, P, radii, msg, scratch_counts);
	hipLaunchKernelGGL(block_base_kernel, dim3(1), dim3(1024), 0, s, P, scratch_counts, msg);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_union_index(int P, int N, const uint32_t* msgs, const unsigned long long* msg_offsets, uint32_t* out_hdr, uint32_t* scratch_counts,
                    void* stream)
{
	if (P <= 0) return GSR_OK;
	if (!msgs || !msg_offsets || !out_hdr || !scratch_counts || N <= 0) return GSR_ERR_ARG;
	hipStream_t s = (hipStream_t)stream;
	const int nb = (P + 255) / 256;
	hipLaunchKernelGGL(union_mask_kernel, dim3(nb), dim3(256), 0, s, P, N, msgs, msg_offsets, out_hdr, scratch_counts);
	hipLaunchKernelGGL(block_base_kernel, dim3(1), dim3(1024), 0, s, P, scratch_counts, out_hdr);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_pack_rows(int P, int C, const uint32_t* hdr, const float* in, float* out, int out_stride, int col0, void* stream)
{
	if (P <= 0) return GSR_OK;
	if (!hdr || !in || !out || C <= 0 || out_stride < col0 + C) return GSR_ERR_ARG;
	hipLaunchKernelGGL(pack_rows_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, C, hdr, in, out, out_stride, col0);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_unpack_rows(int P, int C, const uint32_t* hdr, const float* in, int in_stride, int col0, float* out, void* stream)
{
	if (P <= 0) return GSR_OK;
	if (!hdr || !in || !out || C <= 0 || in_stride < col0 + C) return GSR_ERR_ARG;
	hipLaunchKernelGGL(unpack_rows_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, C, hdr, in, in_stride, col0, out);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_pack_geometry(int P, const uint32_t* hdr, const float* g_means3D, const float* g_opacity, const float* g_scales,
                      const float* g_rotations, float* rows, uint32_t* flag_outside, void* stream)
{
	if (P <= 0) return GSR_OK;
	if (!hdr || !g_means3D || !g_opacity || !g_scales || !g_rotations || !rows || !flag_outside) return GSR_ERR_ARG;
	hipLaunchKernelGGL(pack_geometry_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, hdr, g_means3D, g_opacity, g_scales,
	                   g_rotations, rows, flag_outside);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_unpack_geometry(int P, const uint32_t* hdr, const float* rows, float* g_means3D, float* g_opacity, float* g_scales,
                        float* g_rotations, void* stream)
{
	if (P <= 0) return GSR_OK;
	if (!hdr || !g_means3D || !g_opacity || !g_scales || !g_rotations || !rows) return GSR_ERR_ARG;
	hipLaunchKernelGGL(unpack_geometry_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, hdr, rows, g_means3D, g_opacity,
	                   g_scales, g_rotations);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

}  // extern "C"
