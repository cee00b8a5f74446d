// gsr_tsdf.hip -- TSDF fusion of rendered depth points and iso-surface extraction on the GPU.
//
// Replaces, for gs-extract-mesh (gaustudio/scripts/extract_mesh.py:86,115,145), the CPU library the reference calls:
//   vdbfusion.VDBVolume(voxel_size, sdf_trunc, space_carving).integrate(points, origin)   and
//   .extract_triangle_mesh(fill_holes, min_weight)
// vdbfusion (PRBonn/vdbfusion, pip-installed by the reference, not vendored and not present here) is restated from
// its published algorithm (Vizzo et al., "VDBFusion", Sensors 2022, Alg. 1 and the marching cubes of its
// VDBVolume::ExtractTriangleMesh): per point a ray is walked through the voxels of the truncation band with a
// 3-D DDA, every visited voxel whose signed distance is > -sdf_trunc receives tsdf = min(sdf_trunc, sdf) with
// weight 1 as a running average.  PARITY UNPINNED against the library itself.
//
// MI355X design:
//   * block-sparse volume: 8^3-voxel blocks in an open-addressing hash (64-bit keys, atomicCAS insert); the voxel
//     storage of a block lives AT its hash slot (vox[slot*512 + local]), so there is no allocator and no
//     publish race -- HBM is sized for it (4 KiB per slot; the Python owner picks the capacity);
//   * a voxel is ONE 64-bit word  (sum_q << 24) | count,  sum_q = sum of tsdf/sdf_trunc in 2^-15 fixed point:
//     one integer atomicAdd per update, order-independent => bit-deterministic fusion (a float running average is
//     order dependent); mean tsdf = sum_q / count * sdf_trunc / 2^15.  Field widths: 24-bit count and 40-bit
//     signed sum hold 2^24 - 1 observations of a voxel at full magnitude (|tsdf| = sdf_trunc) -- the limit of the format;
//   * extraction: marching cubes with tables derived in gen_mc_tables.py, shared vertices (each voxel owns the
//     three edges leaving its minimum corner), count -> scan -> emit, no atomics in the emit passes.
// All state is caller-owned device memory (torch tensors in gaustudio_amd/tsdf.py); the entry points are stateless.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsrast.h"
#include "gsr_mc_tables.h"

namespace {

constexpr uint64_t EMPTY = ~0ull;
constexpr int BLOCK_VOX = 512;
constexpr float QSCALE = 32768.0f;   // 2^15: 2^24 observations x 2^15 fit the 40-bit signed sum field

__device__ __constant__ uint8_t d_ntris[256];
__device__ __constant__ uint16_t d_edge_mask[256];
__device__ __constant__ uint8_t d_tris[256][3 * GSR_MC_MAX_TRIS];
// cube corner offsets and, per cube edge, the corner that owns it (the edge's minimum corner) and its axis
__device__ __constant__ int d_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__device__ __constant__ uint8_t d_edge_owner[12] = {0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3};
__device__ __constant__ uint8_t d_edge_axis[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};

__host__ __device__ __forceinline__ uint64_t block_key(int bx, int by, int bz)
{
	const uint64_t B = 1u << 20;
	return ((uint64_t)(bx + (int)B) << 42) | ((uint64_t)(by + (int)B) << 21) | (uint64_t)(bz + (int)B);
}
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
	x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
	x ^= x >> 27; x *= 0x94d049bb133111ebull;
	x ^= x >> 31;
	return x;
}

// returns the slot of `key`, inserting it if absent; -1 when the table is full (probe limit)
__device__ int64_t find_or_insert(unsigned long long* keys, uint64_t mask, uint64_t key)
{
	uint64_t slot = mix64(key) & mask;
	for (uint64_t probe = 0; probe <= mask; probe++) {
		unsigned long long k = __atomic_load_n(&keys[slot], __ATOMIC_RELAXED);
		if (k == key) return (int64_t)slot;
		if (k == EMPTY) {
			const unsigned long long old = atomicCAS(&keys[slot], (unsigned long long)EMPTY, (unsigned long long)key);
			if (old == EMPTY || old == key) return (int64_t)slot;
		}
		slot = (slot + 1) & mask;
	}
	return -1;
}
__device__ int64_t find_slot(const unsigned long long* keys, uint64_t mask, uint64_t key)
{
	uint64_t slot = mix64(key) & mask;
	for (uint64_t probe = 0; probe <= mask; probe++) {
		const unsigned long long k = keys[slot];
		if (k == key) return (int64_t)slot;
		if (k == EMPTY) return -1;
		slot = (slot + 1) & mask;
	}
	return -1;
}

// OpenVDB math::MinIndex: index of the smallest component with its tie-breaking table
__device__ __forceinline__ int min_index(float a, float b, float c)
{
	// table {2, 1, 9, 1, 2, 9, 0, 0} (9 = impossible combination) packed in nibbles
	const uint32_t packed = 2u | (1u << 4) | (9u << 8) | (1u << 12) | (2u << 16) | (9u << 20);
	return (int)((packed >> (4 * (((a < b) << 2) + ((a < c) << 1) + (b < c)))) & 15u);
}

#define GSR_TSDF_COUNT_FULL 0xf00000ull   // observations a voxel takes: 2^24 - 2^20 (the headroom makes the overflow guard race-free)
// One packed voxel update (`add` = (sum_q << 24) + count, possibly the aggregate of several observations) into the
// global volume.  Returns false when the block hash is full.
__device__ __forceinline__ bool tsdf_commit(unsigned long long* __restrict__ keys, uint64_t mask, unsigned long long* __restrict__ vox,
                                            uint32_t* __restrict__ status, int vx, int vy, int vz, unsigned long long add,
                                            int64_t& cached_slot, int& cbx, int& cby, int& cbz)
{
	const int bxk = vx >> 3, byk = vy >> 3, bzk = vz >> 3;
	if (bxk != cbx || byk != cby || bzk != cbz) {
		cached_slot = find_or_insert(keys, mask, block_key(bxk, byk, bzk));
		cbx = bxk; cby = byk; cbz = bzk;
	}
	if (cached_slot < 0) { atomicOr(&status[0], 1u); return false; }   // table full
	const int local = ((vz & 7) << 6) | ((vy & 7) << 3) | (vx & 7);
	unsigned long long* cell = &vox[(size_t)cached_slot * BLOCK_VOX + local];
	// ONE returning add (a read of the cell in front of it was measured: 0.29 -> 0.48 ms per 1080p frame).  The 24-bit
	// count is declared full 2^20 observations BELOW its capacity: an add that takes it past that mark is undone and
	// reported (status bit 1; the voxel keeps what it had).  Between such an add and its undo the count is transiently too
	// high, but a carry into the sum field would need another million observations of the same voxel to land in that
	// window; concurrent updates that see the transient value are past the mark themselves and undo theirs as well.
	const unsigned long long prev = atomicAdd(cell, add);
	if ((prev & 0xffffffull) + (add & 0xffffffull) > GSR_TSDF_COUNT_FULL) {
		atomicAdd(cell, (unsigned long long)(-(long long)add));
		atomicOr(&status[0], 2u);
	}
	return true;
}

// VDBFusion Alg. 1, one thread per point.  Plain float arithmetic, one rounding per operation (compiled with
// -ffp-contract=off): the CPU oracle (oracle/tsdf_oracle.c) performs the same operations in the same order.
//
// Voxel updates are PRE-AGGREGATED per workgroup: neighbouring points of a depth map (consecutive in the list) walk
// through the same voxels -- ~5 pixels per 1 cm voxel along an image row at 3 m -- and a device-scope atomic per
// update made the kernel atomic-bound (round 1: 1.1 ms per 1080p frame = the ~20 G/s ceiling of the chip).  Each
// workgroup accumulates its 256 rays' updates in an LDS hash of voxels near its first point (32-bit local keys:
// 10 bits per axis around that voxel; returning ds_cmpst to claim a slot, ds_add_u64 for the packed word) and commits
// every touched voxel ONCE.  Updates that find no slot (table full after a few probes, voxel outside the local window:
// long space-carving rays) go straight to the volume as before.  Integer adds commute, so the volume is bit-identical
// to the unaggregated one (tests/test_tsdf.py compares the integer state with the oracle's).
#define GSR_TSDF_LNS 2048   // slots of the workgroup-local table (24 KiB of LDS)
// row_w > 0: `points` is an image-shaped point map [N / row_w][row_w] (what depth2point returns) and a workgroup takes a
// square PATCH of it instead of consecutive points of one row: neighbouring rows walk through the same voxels as
// neighbouring columns do, so the workgroup's table holds several times fewer distinct voxels per ray and commits that
// many fewer device atomics (C3-extract: 0.297 ms as a list, 0.150 in 16 x 16 patches, 0.131 in 32 x 32 with a table
// of twice the size).  The volume is the same to the bit either way: integer adds commute.
template <int NT>   // threads per workgroup: 256 (plain list) or 1024 (32 x 32 patches of a map)
__global__ __launch_bounds__(NT) void tsdf_integrate_kernel(const float* __restrict__ points, int N, int row_w, float ox, float oy, float oz,
                                                             float voxel_size, float sdf_trunc, int space_carving,
                                                             unsigned long long* __restrict__ keys, uint64_t mask,
                                                             unsigned long long* __restrict__ vox, uint32_t* __restrict__ status)
{
	constexpr int LNS = NT == 1024 ? 2 * GSR_TSDF_LNS : GSR_TSDF_LNS, LNB = NT == 1024 ? 12 : 11;   // table slots, log2
	__shared__ uint32_t lkeys[LNS];
	__shared__ unsigned long long lvals[LNS];
	__shared__ int s_org[3];
	__shared__ int s_first;
	const int tid = threadIdx.x;
	for (int k = tid; k < LNS; k += NT) { lkeys[k] = 0xffffffffu; lvals[k] = 0ull; }
	if (tid == 0) { s_first = NT; s_org[0] = s_org[1] = s_org[2] = 0; }
	const float inv_vs = 1.0f / voxel_size;
	int64_t cached_slot = -1;
	int cbx = 0x7fffffff, cby = 0, cbz = 0;
	int i = blockIdx.x * NT + tid;
	bool active = i < N;
	if (row_w > 0) {
		constexpr int PS = NT == 1024 ? 32 : 16, PB = NT == 1024 ? 5 : 4;   // patch side
		const int px_tiles = (row_w + PS - 1) >> PB, rows = N / row_w;
		const int x = (int)(blockIdx.x % px_tiles) * PS + (tid & (PS - 1)), y = (int)(blockIdx.x / px_tiles) * PS + (tid >> PB);
		active = x < row_w && y < rows;
		i = y * row_w + x;
	}
	float px = 0.f, py = 0.f, pz = 0.f, depth = 0.f;
	if (active) {
		px = points[3 * (size_t)i]; py = points[3 * (size_t)i + 1]; pz = points[3 * (size_t)i + 2];
		const float dx = px - ox, dy = py - oy, dz = pz - oz;
		depth = sqrtf(dx * dx + dy * dy + dz * dz);
		// degenerate / non-finite point.  A point AT the sensor origin is what depth2point makes of a masked pixel (depth 0):
		// a whole depth map can be integrated without compacting the valid pixels first.  "At" = within a thousandth of a
		// voxel: the unprojection's camera centre and the caller's `origin` agree to rounding only, and a ray of 1e-7
		// units would otherwise carve +-sdf_trunc around the sensor, hundreds of thousands of times into the same voxels.
		// That rounding scales with the COORDINATES (the unprojection inverts a view matrix: ~1e-6 of |origin|), not with
		// the voxel: for a sensor at coordinates of a few hundred and 2-cm voxels it exceeds a thousandth of a voxel, so
		// the threshold is also 1e-5 of the largest origin coordinate.  No real surface sample is that close.
		const float at_origin = fmaxf(1.0e-3f * voxel_size, 1.0e-5f * fmaxf(fabsf(ox), fmaxf(fabsf(oy), fabsf(oz))));
		if (!(depth > at_origin) || !(depth < 3.0e38f)) active = false;
	}
	__syncthreads();
	// centre of the local window: the voxel of the workgroup's first ACTIVE point (anything nearby would do)
	const bool windowable = active && fabsf(px * inv_vs) < 1.0e9f && fabsf(py * inv_vs) < 1.0e9f && fabsf(pz * inv_vs) < 1.0e9f;
	if (windowable) atomicMin(&s_first, tid);
	__syncthreads();
	if (tid == s_first) {
		s_org[0] = (int)floorf(px * inv_vs); s_org[1] = (int)floorf(py * inv_vs); s_org[2] = (int)floorf(pz * inv_vs);
	}
	__syncthreads();
	const int orgx = s_org[0] - 512, orgy = s_org[1] - 512, orgz = s_org[2] - 512;
	if (active) {
		const float dx = px - ox, dy = py - oy, dz = pz - oz;
		const float dirx = dx / depth, diry = dy / depth, dirz = dz / depth;
		// ray in index space (uniform scale map): eye / voxel_size, same direction, times / voxel_size
		const float ex = ox * inv_vs, ey = oy * inv_vs, ez = oz * inv_vs;
		const float t0 = (space_carving ? 0.0f : depth - sdf_trunc) * inv_vs;
		const float t1 = (depth + sdf_trunc) * inv_vs;
		// DDA (openvdb::math::DDA<Ray, 0>::init)
		const float posx = ex + dirx * t0, posy = ey + diry * t0, posz = ez + dirz * t0;
		int vx = (int)floorf(posx), vy = (int)floorf(posy), vz = (int)floorf(posz);
		const float BIG = 3.4028235e38f;
		float nx, ny, nz, ddx, ddy, ddz;
		int sx, sy, sz;
#define DDA_AXIS(dir, pos, v, s, nxt, dlt)                                                         \
		if (dir == 0.f) { s = 0; nxt = BIG; dlt = BIG; }                                               \
		else { const float inv = 1.0f / dir;                                                           \
			if (inv > 0.f) { s = 1; nxt = t0 + ((float)(v + 1) - pos) * inv; dlt = inv; }               \
			else { s = -1; nxt = t0 + ((float)v - pos) * inv; dlt = -inv; } }
		DDA_AXIS(dirx, posx, vx, sx, nx, ddx)
		DDA_AXIS(diry, posy, vy, sy, ny, ddy)
		DDA_AXIS(dirz, posz, vz, sz, nz, ddz)
#undef DDA_AXIS
		const float half = voxel_size * 0.5f;
		const float qs = QSCALE / sdf_trunc;
		for (int guard = 0; guard < (1 << 20); guard++) {
			// voxel centre (GetVoxelCenter) and projective signed distance (ComputeSDF)
			const float cx = (float)vx * voxel_size + half, cy = (float)vy * voxel_size + half, cz = (float)vz * voxel_size + half;
			const float ax = cx - ox, ay = cy - oy, az = cz - oz;      // voxel - origin
			const float bx = px - cx, by = py - cy, bz = pz - cz;      // point - voxel
			const float dist = sqrtf(bx * bx + by * by + bz * bz);
			const float proj = ax * bx + ay * by + az * bz;
			const float sdf = (proj / fabsf(proj)) * dist;             // NaN when proj == 0: skipped below
			if (sdf > -sdf_trunc) {
				const float tsdf = fminf(sdf_trunc, sdf);
				const long long q = (long long)__float2int_rn(tsdf * qs);
				const unsigned long long add = (unsigned long long)(q * (1ll << 24) + 1);
				// workgroup-local aggregation first
				const unsigned int rx = (unsigned int)(vx - orgx), ry = (unsigned int)(vy - orgy), rz = (unsigned int)(vz - orgz);
				bool placed = false;
				if ((rx | ry | rz) < 1024u) {
					const uint32_t lkey = rx | (ry << 10) | (rz << 20);
					uint32_t h = (lkey * 2654435761u) >> (32 - LNB);
#pragma unroll 1
					for (int probe = 0; probe < 8 && !placed; probe++) {
						const uint32_t old = atomicCAS(&lkeys[h], 0xffffffffu, lkey);
						if (old == 0xffffffffu || old == lkey) {
							atomicAdd(&lvals[h], add);
							placed = true;
						}
						h = (h + 1) & (LNS - 1);
					}
				}
				if (!placed && !tsdf_commit(keys, mask, vox, status, vx, vy, vz, add, cached_slot, cbx, cby, cbz)) break;
			}
			// DDA::step
			const int axis = min_index(nx, ny, nz);
			float t;
			if (axis == 0) { t = nx; nx += ddx; vx += sx; }
			else if (axis == 1) { t = ny; ny += ddy; vy += sy; }
			else { t = nz; nz += ddz; vz += sz; }
			if (!(t <= t1)) break;
		}
	}
	// commit the workgroup's aggregated voxels, one device atomic each
	__syncthreads();
	for (int k = tid; k < LNS; k += NT) {
		const uint32_t lkey = lkeys[k];
		if (lkey == 0xffffffffu) continue;
		const int vx = orgx + (int)(lkey & 1023u), vy = orgy + (int)((lkey >> 10) & 1023u), vz = orgz + (int)(lkey >> 20);
		tsdf_commit(keys, mask, vox, status, vx, vy, vz, lvals[k], cached_slot, cbx, cby, cbz);
	}
}

__device__ __forceinline__ void unpack(unsigned long long w, float sdf_trunc, float& f, uint32_t& count)
{
	count = (uint32_t)(w & 0xffffffull);
	const long long s = (long long)w >> 24;
	f = count ? ((float)s / (float)count) * (sdf_trunc / QSCALE) : sdf_trunc;   // background = +sdf_trunc
}

// dump of one block for tests: per voxel (count, mean tsdf)
__global__ __launch_bounds__(512) void tsdf_export_kernel(const unsigned long long* __restrict__ vox, const uint32_t* __restrict__ slots,
                                                          float sdf_trunc, uint32_t* __restrict__ counts, float* __restrict__ tsdf,
                                                          long long* __restrict__ sums)
{
	const size_t src = (size_t)slots[blockIdx.x] * BLOCK_VOX + threadIdx.x, dst = (size_t)blockIdx.x * BLOCK_VOX + threadIdx.x;
	float f;
	uint32_t c;
	unpack(vox[src], sdf_trunc, f, c);
	counts[dst] = c;
	tsdf[dst] = f;
	sums[dst] = (long long)vox[src] >> 24;
}

// ---- marching cubes over the occupied blocks (blocks[] = hash slots in a caller-chosen, deterministic order) ----
// s_nb[8]: hash slots of the 2x2x2 blocks starting at the workgroup's block (-1 = absent), index dz*4+dy*2+dx
__device__ __forceinline__ void decode_key(uint64_t key, int& bx, int& by, int& bz)
{
	const int B = 1 << 20;
	bx = (int)((key >> 42) & 0x1fffff) - B;
	by = (int)((key >> 21) & 0x1fffff) - B;
	bz = (int)(key & 0x1fffff) - B;
}

// value of the voxel at local coordinates (lx,ly,lz) in [0,8] of the 2x2x2 block neighbourhood; false = block absent
__device__ __forceinline__ bool fetch(const unsigned long long* __restrict__ vox, const int64_t* nb, int lx, int ly, int lz,
                                      float sdf_trunc, float& f, uint32_t& c)
{
	const int64_t s = nb[((lz >> 3) << 2) | ((ly >> 3) << 1) | (lx >> 3)];
	if (s < 0) { f = sdf_trunc; c = 0; return false; }
	unpack(vox[(size_t)s * BLOCK_VOX + (((lz & 7) << 6) | ((ly & 7) << 3) | (lx & 7))], sdf_trunc, f, c);
	return true;
}

// case index of the cube whose minimum corner is this voxel; 0 when the cube is not extractable
__device__ __forceinline__ int cube_case(const unsigned long long* __restrict__ vox, const int64_t* nb, int lx, int ly, int lz,
                                         float sdf_trunc, uint32_t min_count, int fill_holes, float* fout)
{
	int idx = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		float f;
		uint32_t c;
		// a corner in a block that was never allocated makes the cube non-extractable (its edge owners have no storage)
		if (!fetch(vox, nb, lx + d_corner[i][0], ly + d_corner[i][1], lz + d_corner[i][2], sdf_trunc, f, c)) return 0;
		if ((!fill_holes && c == 0) || c < min_count) return 0;
		fout[i] = f;
		if (f < 0.0f) idx |= 1 << i;
	}
	return idx == 255 ? 0 : idx;
}

// pass A: per voxel the cube case (u8) and, OR-ed into the owning voxels, which of their three edges carry a vertex.
// One workgroup per occupied block, one thread per voxel.  flags / cases are indexed by COMPACT block index;
// cidx maps hash slot -> compact index.
__global__ __launch_bounds__(512) void mc_classify_kernel(const unsigned long long* __restrict__ keys, uint64_t mask,
                                                          const unsigned long long* __restrict__ vox, const uint32_t* __restrict__ blocks,
                                                          const uint32_t* __restrict__ cidx, float sdf_trunc, uint32_t min_count,
                                                          int fill_holes, uint8_t* __restrict__ cases, uint32_t* __restrict__ flags)
{
	__shared__ int64_t s_nb[8];
	const uint32_t slot = blocks[blockIdx.x];
	if (threadIdx.x < 8) {
		int bx, by, bz;
		decode_key(keys[slot], bx, by, bz);
		s_nb[threadIdx.x] = threadIdx.x == 0 ? (int64_t)slot
		                                     : find_slot(keys, mask, block_key(bx + (threadIdx.x & 1), by + ((threadIdx.x >> 1) & 1), bz + (threadIdx.x >> 2)));
	}
	__syncthreads();
	const int lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x >> 6;
	float f[8];
	const int cs = cube_case(vox, s_nb, lx, ly, lz, sdf_trunc, min_count, fill_holes, f);
	cases[(size_t)blockIdx.x * BLOCK_VOX + threadIdx.x] = (uint8_t)cs;
	if (cs == 0) return;
	const uint32_t em = d_edge_mask[cs];
	for (int e = 0; e < 12; e++) {
		if (!((em >> e) & 1)) continue;
		const int o = d_edge_owner[e];
		const int ox = lx + d_corner[o][0], oy = ly + d_corner[o][1], oz = lz + d_corner[o][2];
		const int64_t s = s_nb[((oz >> 3) << 2) | ((oy >> 3) << 1) | (ox >> 3)];   // present: the cube was extractable
		const uint32_t cb = cidx[s];
		atomicOr(&flags[(size_t)cb * BLOCK_VOX + (((oz & 7) << 6) | ((oy & 7) << 3) | (ox & 7))], 1u << d_edge_axis[e]);
	}
}

// pass B: per block totals (vertices, triangles) for the host-side exclusive scan
__global__ __launch_bounds__(512) void mc_count_kernel(const uint8_t* __restrict__ cases, const uint32_t* __restrict__ flags,
                                                       uint32_t* __restrict__ block_nv, uint32_t* __restrict__ block_nt)
{
	__shared__ uint32_t s_v[8], s_t[8];
	const size_t i = (size_t)blockIdx.x * BLOCK_VOX + threadIdx.x;
	uint32_t nv = __popc(flags[i] & 7u), nt = d_ntris[cases[i]];
	for (int o = 32; o > 0; o >>= 1) { nv += __shfl_xor((int)nv, o, 64); nt += __shfl_xor((int)nt, o, 64); }
	if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = nv; s_t[threadIdx.x >> 6] = nt; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t a = 0, b = 0;
		for (int w = 0; w < 8; w++) { a += s_v[w]; b += s_t[w]; }
		block_nv[blockIdx.x] = a;
		block_nt[blockIdx.x] = b;
	}
}

// exclusive scan of a per-voxel count inside a 512-thread block; returns this thread's offset
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w)
{
	uint32_t incl = v;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
		if (lane >= o) incl += t;
	}
	if (lane == 63) s_w[wv] = incl;
	__syncthreads();
	uint32_t base = 0;
	for (int w = 0; w < wv; w++) base += s_w[w];
	__syncthreads();
	return base + incl - v;
}

// pass C: vertices.  vbase[voxel] = index of the voxel's first vertex.
__global__ __launch_bounds__(512) void mc_vertices_kernel(const unsigned long long* __restrict__ keys, uint64_t mask,
                                                          const unsigned long long* __restrict__ vox, const uint32_t* __restrict__ blocks,
                                                          const uint32_t* __restrict__ flags, const uint32_t* __restrict__ block_voff,
                                                          float voxel_size, float sdf_trunc, uint32_t* __restrict__ vbase,
                                                          float* __restrict__ vertices)
{
	__shared__ int64_t s_nb[8];
	__shared__ uint32_t s_w[8];
	__shared__ int s_b[3];
	const uint32_t slot = blocks[blockIdx.x];
	if (threadIdx.x < 8) {
		int bx, by, bz;
		decode_key(keys[slot], bx, by, bz);
		if (threadIdx.x == 0) { s_b[0] = bx; s_b[1] = by; s_b[2] = bz; }
		s_nb[threadIdx.x] = threadIdx.x == 0 ? (int64_t)slot
		                                     : find_slot(keys, mask, block_key(bx + (threadIdx.x & 1), by + ((threadIdx.x >> 1) & 1), bz + (threadIdx.x >> 2)));
	}
	__syncthreads();
	const size_t i = (size_t)blockIdx.x * BLOCK_VOX + threadIdx.x;
	const uint32_t fl = flags[i] & 7u;
	const uint32_t off = block_voff[blockIdx.x] + block_excl_scan(__popc(fl), s_w);
	vbase[i] = off;
	if (!fl) return;
	const int lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x >> 6;
	float f0;
	uint32_t c0;
	fetch(vox, s_nb, lx, ly, lz, sdf_trunc, f0, c0);
	// voxel centre in world units (vdbfusion: half_voxel_length + voxel_length * index)
	const float half = voxel_size * 0.5f;
	const float base[3] = {half + voxel_size * (float)(s_b[0] * 8 + lx), half + voxel_size * (float)(s_b[1] * 8 + ly),
	                       half + voxel_size * (float)(s_b[2] * 8 + lz)};
	uint32_t k = off;
	for (int a = 0; a < 3; a++) {
		if (!((fl >> a) & 1)) continue;
		float f1;
		uint32_t c1;
		fetch(vox, s_nb, lx + (a == 0), ly + (a == 1), lz + (a == 2), sdf_trunc, f1, c1);
		const float a0 = fabsf(f0), a1 = fabsf(f1);
		float p[3] = {base[0], base[1], base[2]};
		p[a] += a0 * voxel_size / (a0 + a1);
		vertices[3 * (size_t)k] = p[0]; vertices[3 * (size_t)k + 1] = p[1]; vertices[3 * (size_t)k + 2] = p[2];
		k++;
	}
}

// pass D: triangles (vertex indices through the owners' vbase)
__global__ __launch_bounds__(512) void mc_triangles_kernel(const unsigned long long* __restrict__ keys, uint64_t mask,
                                                           const uint32_t* __restrict__ blocks, const uint32_t* __restrict__ cidx,
                                                           const uint8_t* __restrict__ cases, const uint32_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ vbase, const uint32_t* __restrict__ block_toff,
                                                           int* __restrict__ triangles)
{
	__shared__ int64_t s_nb[8];
	__shared__ uint32_t s_w[8];
	const uint32_t slot = blocks[blockIdx.x];
	if (threadIdx.x < 8) {
		int bx, by, bz;
		decode_key(keys[slot], bx, by, bz);
		s_nb[threadIdx.x] = threadIdx.x == 0 ? (int64_t)slot
		                                     : find_slot(keys, mask, block_key(bx + (threadIdx.x & 1), by + ((threadIdx.x >> 1) & 1), bz + (threadIdx.x >> 2)));
	}
	__syncthreads();
	const size_t i = (size_t)blockIdx.x * BLOCK_VOX + threadIdx.x;
	const int cs = cases[i];
	const uint32_t nt = d_ntris[cs];
	uint32_t t = block_toff[blockIdx.x] + block_excl_scan(nt, s_w);
	if (!nt) return;
	const int lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x >> 6;
	for (uint32_t k = 0; k < nt; k++, t++) {
		for (int c = 0; c < 3; c++) {
			const int e = d_tris[cs][3 * k + c];
			const int o = d_edge_owner[e], ax = d_edge_axis[e];
			const int ox = lx + d_corner[o][0], oy = ly + d_corner[o][1], oz = lz + d_corner[o][2];
			const int64_t s = s_nb[((oz >> 3) << 2) | ((oy >> 3) << 1) | (ox >> 3)];
			const size_t ov = (size_t)cidx[s] * BLOCK_VOX + (((oz & 7) << 6) | ((oy & 7) << 3) | (ox & 7));
			const uint32_t fl = flags[ov] & 7u;
			triangles[3 * (size_t)t + c] = (int)(vbase[ov] + __popc(fl & ((1u << ax) - 1u)));
		}
	}
}

bool g_tables_loaded[16] = {};
int load_tables()
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GSR_ERR_HIP;
	if (g_tables_loaded[dev]) return GSR_OK;
	if (hipMemcpyToSymbol(HIP_SYMBOL(d_ntris), gsr_mc_ntris, sizeof(gsr_mc_ntris)) != hipSuccess) return GSR_ERR_HIP;
	if (hipMemcpyToSymbol(HIP_SYMBOL(d_edge_mask), gsr_mc_edge_mask, sizeof(gsr_mc_edge_mask)) != hipSuccess) return GSR_ERR_HIP;
	if (hipMemcpyToSymbol(HIP_SYMBOL(d_tris), gsr_mc_tris, sizeof(gsr_mc_tris)) != hipSuccess) return GSR_ERR_HIP;
	g_tables_loaded[dev] = true;
	return GSR_OK;
}

bool pow2(uint64_t v) { return v && !(v & (v - 1)); }

}  // namespace

extern "C" {

int gsr_tsdf_integrate_map(const float* points, int num_points, int row_width, const float origin[3], float voxel_size,
                           float sdf_trunc, int space_carving, uint64_t* block_keys, uint64_t capacity, uint64_t* voxels,
                           uint32_t* status, void* stream)
{
	if (num_points <= 0) return GSR_OK;
	if (!points || !origin || !block_keys || !voxels || !status || !pow2(capacity) || !(voxel_size > 0.f) || !(sdf_trunc > 0.f))
		return GSR_ERR_ARG;
	if (row_width < 0 || (row_width > 0 && num_points % row_width != 0)) return GSR_ERR_ARG;
	if (row_width > 0) {   // 32 x 32 patches, 1024 threads, 4096-slot table (measured at C3-extract: 0.297 ms as a list, 0.150 in 16 x 16 patches, 0.131 in 32 x 32)
		const unsigned grid = (unsigned)(((row_width + 31) / 32) * ((num_points / row_width + 31) / 32));
		hipLaunchKernelGGL(tsdf_integrate_kernel<1024>, dim3(grid), dim3(1024), 0, (hipStream_t)stream, points,
		                   num_points, row_width, origin[0], origin[1], origin[2], voxel_size, sdf_trunc, space_carving,
		                   reinterpret_cast<unsigned long long*>(block_keys), capacity - 1,
		                   reinterpret_cast<unsigned long long*>(voxels), status);
		return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
	}
	const unsigned grid = (unsigned)((num_points + 255) / 256);
	hipLaunchKernelGGL(tsdf_integrate_kernel<256>, dim3(grid), dim3(256), 0, (hipStream_t)stream, points,
	                   num_points, row_width, origin[0], origin[1], origin[2], voxel_size, sdf_trunc, space_carving,
	                   reinterpret_cast<unsigned long long*>(block_keys), capacity - 1,
	                   reinterpret_cast<unsigned long long*>(voxels), status);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_tsdf_integrate(const float* points, int num_points, const float origin[3], float voxel_size, float sdf_trunc,
                       int space_carving, uint64_t* block_keys, uint64_t capacity, uint64_t* voxels, uint32_t* status,
                       void* stream)
{
	return gsr_tsdf_integrate_map(points, num_points, 0, origin, voxel_size, sdf_trunc, space_carving, block_keys, capacity,
	                              voxels, status, stream);
}

int gsr_tsdf_export_blocks(const uint64_t* voxels, const uint32_t* block_slots, int num_blocks, float sdf_trunc,
                           uint32_t* counts, float* tsdf, int64_t* sums, void* stream)
{
	if (num_blocks <= 0) return GSR_OK;
	if (!voxels || !block_slots || !counts || !tsdf || !sums) return GSR_ERR_ARG;
	hipLaunchKernelGGL(tsdf_export_kernel, dim3(num_blocks), dim3(512), 0, (hipStream_t)stream,
	                   reinterpret_cast<const unsigned long long*>(voxels), block_slots, sdf_trunc, counts, tsdf,
	                   reinterpret_cast<long long*>(sums));
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_tsdf_mc_classify(const uint64_t* block_keys, uint64_t capacity, const uint64_t* voxels, const uint32_t* block_slots,
                         int num_blocks, const uint32_t* slot_to_block, float sdf_trunc, float min_weight, int fill_holes,
                         uint8_t* cases, uint32_t* edge_flags, uint32_t* block_num_vertices, uint32_t* block_num_triangles,
                         void* stream)
{
	if (num_blocks <= 0) return GSR_OK;
	if (!block_keys || !voxels || !block_slots || !slot_to_block || !cases || !edge_flags || !block_num_vertices ||
	    !block_num_triangles || !pow2(capacity))
		return GSR_ERR_ARG;
	const int rc = load_tables();
	if (rc != GSR_OK) return rc;
	hipStream_t s = (hipStream_t)stream;
	// weights are integer counts here: weight < min_weight  <=>  count < ceil(min_weight)
	const uint32_t min_count = min_weight <= 0.f ? 0u : (uint32_t)ceilf(min_weight);
	if (hipMemsetAsync(edge_flags, 0, sizeof(uint32_t) * (size_t)num_blocks * BLOCK_VOX, s) != hipSuccess) return GSR_ERR_HIP;
	hipLaunchKernelGGL(mc_classify_kernel, dim3(num_blocks), dim3(512), 0, s, reinterpret_cast<const unsigned long long*>(block_keys),
	                   capacity - 1, reinterpret_cast<const unsigned long long*>(voxels), block_slots, slot_to_block, sdf_trunc,
	                   min_count, fill_holes, cases, edge_flags);
	hipLaunchKernelGGL(mc_count_kernel, dim3(num_blocks), dim3(512), 0, s, cases, edge_flags, block_num_vertices,
	                   block_num_triangles);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

int gsr_tsdf_mc_emit(const uint64_t* block_keys, uint64_t capacity, const uint64_t* voxels, const uint32_t* block_slots,
                     int num_blocks, const uint32_t* slot_to_block, float voxel_size, float sdf_trunc, const uint8_t* cases,
                     const uint32_t* edge_flags, const uint32_t* block_vertex_offset, const uint32_t* block_triangle_offset,
                     uint32_t* vertex_base, float* vertices, int* triangles, void* stream)
{
	if (num_blocks <= 0) return GSR_OK;
	if (!block_keys || !voxels || !block_slots || !slot_to_block || !cases || !edge_flags || !block_vertex_offset ||
	    !block_triangle_offset || !vertex_base || !vertices || !triangles || !pow2(capacity))
		return GSR_ERR_ARG;
	const int rc = load_tables();
	if (rc != GSR_OK) return rc;
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(mc_vertices_kernel, dim3(num_blocks), dim3(512), 0, s, reinterpret_cast<const unsigned long long*>(block_keys),
	                   capacity - 1, reinterpret_cast<const unsigned long long*>(voxels), block_slots, edge_flags,
	                   block_vertex_offset, voxel_size, sdf_trunc, vertex_base, vertices);
	hipLaunchKernelGGL(mc_triangles_kernel, dim3(num_blocks), dim3(512), 0, s, reinterpret_cast<const unsigned long long*>(block_keys),
	                   capacity - 1, block_slots, slot_to_block, cases, edge_flags, vertex_base, block_triangle_offset, triangles);
	return hipGetLastError() == hipSuccess ? GSR_OK : GSR_ERR_HIP;
}

}  // extern "C"
