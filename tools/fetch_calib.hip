// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the
// compositing kernels (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access
// pattern before trusting an absolute").  Four kernels with exactly known byte counts over buffers far larger
// than the 256 MiB Infinity Cache:
//   calib_stream_read     16 B per lane, coalesced                     (the guide's calibrated case: reads 1/2)
//   calib_gather_read     the staging pattern of composite_fwd/bwd: 4 lanes fetch the four 16-B parts of one
//                         64-B record, records visited once each in a pseudo-random order
//   calib_stream_write    16 B per lane, coalesced
//   calib_row_write       the flush pattern of composite_bwd: one thread stores a 48-B row (3 x 16 B) at a
//                         pseudo-random row index, every row written once
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o k -- tools/fetch_calib.bin     (then WRITE_SIZE)
// The program prints the true byte count of every kernel; tools/pmc_dump.py prints the counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void calib_stream_read(const float4* __restrict__ src, size_t n, float* __restrict__ sink)
{
	float acc = 0.f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float4 v = src[i];
		acc += v.x + v.y + v.z + v.w;
	}
	if (acc == 123.456f) sink[0] = acc;
}

// record index permutation: multiply by an odd constant modulo a power of two (a bijection)
__device__ __forceinline__ size_t perm(size_t i, size_t mask) { return (i * 0x9E3779B1ull + 0x7F4A7C15ull) & mask; }

__global__ void calib_gather_read(const float4* __restrict__ recs, size_t nrec_mask, float* __restrict__ sink)
{
	float acc = 0.f;
	const size_t nthreads = (size_t)gridDim.x * blockDim.x;
	for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < (nrec_mask + 1) * 4; t += nthreads) {
		const size_t r = perm(t >> 2, nrec_mask);
		const float4 v = recs[r * 4 + (t & 3)];
		acc += v.x + v.y + v.z + v.w;
	}
	if (acc == 123.456f) sink[0] = acc;
}

__global__ void calib_stream_write(float4* __restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

__global__ void calib_row_write(float4* __restrict__ rows, size_t nrow_mask)
{
	const size_t nthreads = (size_t)gridDim.x * blockDim.x;
	for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t <= nrow_mask; t += nthreads) {
		float4* d = rows + perm(t, nrow_mask) * 3;
		d[0] = make_float4((float)t, 1.f, 2.f, 3.f);
		d[1] = make_float4(4.f, 5.f, 6.f, 7.f);
		d[2] = make_float4(8.f, 9.f, 0.f, 0.f);
	}
}

int main()
{
	const size_t bytes = 1ull << 30;                 // 1 GiB per buffer: 4x the Infinity Cache
	float4 *a = nullptr, *b = nullptr;
	float* sink = nullptr;
	CHECK(hipMalloc(&a, bytes));
	CHECK(hipMalloc(&b, bytes));
	CHECK(hipMalloc(&sink, 256));
	CHECK(hipMemset(a, 1, bytes));
	CHECK(hipMemset(b, 0, bytes));
	const size_t n16 = bytes / 16, nrec = bytes / 64, nrow = (size_t)1 << 24;   // 16 M rows x 48 B = 768 MiB
	const dim3 grid(256 * 16), block(256);
	for (int rep = 0; rep < 3; rep++) {
		hipLaunchKernelGGL(calib_stream_read, grid, block, 0, 0, a, n16, sink);
		hipLaunchKernelGGL(calib_gather_read, grid, block, 0, 0, a, nrec - 1, sink);
		hipLaunchKernelGGL(calib_stream_write, grid, block, 0, 0, b, n16);
		hipLaunchKernelGGL(calib_row_write, grid, block, 0, 0, b, nrow - 1);
	}
	CHECK(hipDeviceSynchronize());
	printf("true bytes per launch: calib_stream_read %zu  calib_gather_read %zu  calib_stream_write %zu  calib_row_write %zu\n",
	       bytes, bytes, bytes, nrow * 48);
	return 0;
}
