// tsdf_oracle.c -- CPU ORACLE for the TSDF fusion path.  TEST INFRASTRUCTURE ONLY: only
// tests/ and measurement scripts may load it; the product (gaustudio_amd) never does.
//
// PARITY UNPINNED.  The reference calls vdbfusion.VDBVolume (gaustudio/scripts/extract_mesh.py:86,115,145), a
// pip dependency (PRBonn/vdbfusion, unpinned in the reference's requirements) that is neither vendored under
// /root/reference nor installed here.  This file restates its published algorithm -- Vizzo, Guadagnino, Behley,
// Stachniss, "VDBFusion: Flexible and Efficient TSDF Integration of Range Sensor Data", Sensors 22(3), 2022, Alg. 1,
// and the public VDBVolume::Integrate / ComputeSDF / GetVoxelCenter it describes -- as a sequential float program:
//     for each point:  ray origin->point, restricted to [depth - trunc, depth + trunc]  (or [0, ..] with space carving)
//                      walked with OpenVDB's DDA (init / step / MinIndex tie-breaking);
//                      per voxel: sdf = sign(<voxel-origin, point-voxel>) * |point-voxel|;
//                      if sdf > -trunc: tsdf = min(trunc, sdf); weight 1;
//                      new_tsdf = (old_tsdf*old_w + tsdf*w) / (old_w + w)            (running average, in order)
// Next to the float running average it keeps, per voxel, the observation count and the sum of tsdf/trunc in 2^-15
// fixed point: the quantities the HIP kernel accumulates with integer atomics (csrc/gsr_tsdf.hip), which must match
// this file EXACTLY (same float operations, one rounding each: build with -ffp-contract=off).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	int32_t x, y, z;
	int32_t used;
	float tsdf, weight;   // vdbfusion's running average
	int64_t sum_q;
	uint32_t count;
} Vox;

typedef struct {
	Vox* tab;
	uint64_t cap, n;
	float voxel_size, sdf_trunc;
	int space_carving;
} Vol;

static uint64_t hash3(int32_t x, int32_t y, int32_t z)
{
	uint64_t h = ((uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)y * 0xC2B2AE3D27D4EB4Full) ^
	             ((uint64_t)(uint32_t)z * 0x165667B19E3779F9ull);
	h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
	return h;
}

static Vox* lookup(Vol* v, int32_t x, int32_t y, int32_t z, int create);

static void grow(Vol* v)
{
	Vox* old = v->tab;
	const uint64_t oc = v->cap;
	v->cap = oc ? oc * 2 : (1u << 16);
	v->tab = (Vox*)calloc(v->cap, sizeof(Vox));
	v->n = 0;
	for (uint64_t i = 0; i < oc; i++)
		if (old[i].used) {
			Vox* d = lookup(v, old[i].x, old[i].y, old[i].z, 1);
			*d = old[i];
		}
	free(old);
}

static Vox* lookup(Vol* v, int32_t x, int32_t y, int32_t z, int create)
{
	if (create && (v->n + 1) * 2 > v->cap) grow(v);
	if (!v->cap) return NULL;
	uint64_t s = hash3(x, y, z) & (v->cap - 1);
	for (;;) {
		Vox* e = &v->tab[s];
		if (!e->used) {
			if (!create) return NULL;
			e->used = 1; e->x = x; e->y = y; e->z = z;
			e->tsdf = v->sdf_trunc;   // grid background
			e->weight = 0.f; e->sum_q = 0; e->count = 0;
			v->n++;
			return e;
		}
		if (e->x == x && e->y == y && e->z == z) return e;
		s = (s + 1) & (v->cap - 1);
	}
}

void* tso_create(float voxel_size, float sdf_trunc, int space_carving)
{
	Vol* v = (Vol*)calloc(1, sizeof(Vol));
	v->voxel_size = voxel_size; v->sdf_trunc = sdf_trunc; v->space_carving = space_carving;
	return v;
}
void tso_destroy(void* h)
{
	Vol* v = (Vol*)h;
	free(v->tab);
	free(v);
}

// OpenVDB math::MinIndex
static int min_index(float a, float b, float c)
{
	static const int t[8] = {2, 1, 9, 1, 2, 9, 0, 0};
	return t[((a < b) << 2) + ((a < c) << 1) + (b < c)];
}

void tso_integrate(void* h, const float* points, int64_t N, const float* origin)
{
	Vol* v = (Vol*)h;
	const float voxel_size = v->voxel_size, sdf_trunc = v->sdf_trunc;
	const float ox = origin[0], oy = origin[1], oz = origin[2];
	const float BIG = 3.4028235e38f;
	for (int64_t i = 0; i < N; i++) {
		const float px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
		const float dx = px - ox, dy = py - oy, dz = pz - oz;
		const float depth = sqrtf(dx * dx + dy * dy + dz * dz);
		if (!(depth > 0.f) || !(depth < 3.0e38f)) continue;
		const float dirx = dx / depth, diry = dy / depth, dirz = dz / depth;
		const float inv_vs = 1.0f / voxel_size;
		const float ex = ox * inv_vs, ey = oy * inv_vs, ez = oz * inv_vs;
		const float t0 = (v->space_carving ? 0.0f : depth - sdf_trunc) * inv_vs;
		const float t1 = (depth + sdf_trunc) * inv_vs;
		const float posx = ex + dirx * t0, posy = ey + diry * t0, posz = ez + dirz * t0;
		int vx = (int)floorf(posx), vy = (int)floorf(posy), vz = (int)floorf(posz);
		float nx, ny, nz, ddx, ddy, ddz;
		int sx, sy, sz;
#define DDA_AXIS(dir, pos, vv, s, nxt, dlt)                                                \
		if (dir == 0.f) { s = 0; nxt = BIG; dlt = BIG; }                                   \
		else { const float inv = 1.0f / dir;                                               \
			if (inv > 0.f) { s = 1; nxt = t0 + ((float)(vv + 1) - pos) * inv; dlt = inv; } \
			else { s = -1; nxt = t0 + ((float)vv - pos) * inv; dlt = -inv; } }
		DDA_AXIS(dirx, posx, vx, sx, nx, ddx)
		DDA_AXIS(diry, posy, vy, sy, ny, ddy)
		DDA_AXIS(dirz, posz, vz, sz, nz, ddz)
#undef DDA_AXIS
		const float half = voxel_size * 0.5f;
		const float qs = 32768.0f / sdf_trunc;
		for (int guard = 0; guard < (1 << 20); guard++) {
			const float cx = (float)vx * voxel_size + half, cy = (float)vy * voxel_size + half, cz = (float)vz * voxel_size + half;
			const float ax = cx - ox, ay = cy - oy, az = cz - oz;
			const float bx = px - cx, by = py - cy, bz = pz - cz;
			const float dist = sqrtf(bx * bx + by * by + bz * bz);
			const float proj = ax * bx + ay * by + az * bz;
			const float sdf = (proj / fabsf(proj)) * dist;
			if (sdf > -sdf_trunc) {
				const float tsdf = fminf(sdf_trunc, sdf);
				Vox* e = lookup(v, vx, vy, vz, 1);
				const float weight = 1.0f;
				const float new_weight = weight + e->weight;
				e->tsdf = (e->tsdf * e->weight + tsdf * weight) / new_weight;
				e->weight = new_weight;
				e->sum_q += (int64_t)lrintf(tsdf * qs);   // round-to-nearest-even, as v_cvt_i32_f32 / __float2int_rn
				e->count += 1;
			}
			const int axis = min_index(nx, ny, nz);
			float t;
			if (axis == 0) { t = nx; nx += ddx; vx += sx; }
			else if (axis == 1) { t = ny; ny += ddy; vy += sy; }
			else { t = nz; nz += ddz; vz += sz; }
			if (!(t <= t1)) break;
		}
	}
}

int64_t tso_num_voxels(void* h) { return (int64_t)((Vol*)h)->n; }

static int cmp_vox(const void* a, const void* b)
{
	const Vox* p = (const Vox*)a; const Vox* q = (const Vox*)b;
	if (p->z != q->z) return p->z < q->z ? -1 : 1;
	if (p->y != q->y) return p->y < q->y ? -1 : 1;
	if (p->x != q->x) return p->x < q->x ? -1 : 1;
	return 0;
}

// all observed voxels sorted by (z, y, x): coords[n,3], tsdf[n] (running average), weight[n], sum_q[n]
void tso_export(void* h, int32_t* coords, float* tsdf, int32_t* weight, int64_t* sum_q)
{
	Vol* v = (Vol*)h;
	Vox* tmp = (Vox*)malloc(sizeof(Vox) * (v->n ? v->n : 1));
	uint64_t k = 0;
	for (uint64_t i = 0; i < v->cap; i++)
		if (v->tab[i].used) tmp[k++] = v->tab[i];
	qsort(tmp, k, sizeof(Vox), cmp_vox);
	for (uint64_t i = 0; i < k; i++) {
		coords[3 * i] = tmp[i].x; coords[3 * i + 1] = tmp[i].y; coords[3 * i + 2] = tmp[i].z;
		tsdf[i] = tmp[i].tsdf; weight[i] = (int32_t)tmp[i].count; sum_q[i] = tmp[i].sum_q;
	}
	free(tmp);
}
