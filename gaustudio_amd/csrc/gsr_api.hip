// gsr_api.hip -- host side of libgsrast: the C ABI declared in include/gsrast.h.
// Workspace planning, kernel sequencing, error handling, introspection.  No torch types here.
#include "../../include/gsrast.h"
#include "gsr_internal.h"

#include <dlfcn.h>

#include <atomic>
#include <unordered_map>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace gsr;

namespace {

thread_local std::string g_err;
// ---- per-stage profiling: HIP events recorded on the launch stream, resolved lazily ----
struct ProfSet {
	hipEvent_t ev[6];
	uint32_t have = 0;   // bit k: boundary k of the call was recorded (stage k lies between boundaries k and k + 1)
};
struct ProfLog {
	std::vector<ProfSet*> sets;   // pool, reused across resets
	size_t used = 0;
	ProfSet* next()
	{
		if (used == sets.size()) {
			ProfSet* p = new ProfSet;
			for (auto& e : p->ev) (void)hipEventCreate(&e);
			sets.push_back(p);
		}
		ProfSet* p = sets[used++];
		p->have = 0;
		return p;
	}
	// mean stage times over every recorded call; returns the number of calls
	int mean(float* out, int k)
	{
		for (int i = 0; i < k; i++) out[i] = 0.f;
		if (used == 0) return 0;
		int calls = 0;
		int per_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (size_t c = 0; c < used; c++) {
			ProfSet* p = sets[c];
			if (p->have == 0) continue;
			const int last = 31 - __builtin_clz(p->have);
			if (hipEventSynchronize(p->ev[last]) != hipSuccess) continue;
			for (int i = 0; i < k && i + 1 < 6; i++) {
				if (((p->have >> i) & 3u) != 3u) continue;   // both boundaries of stage i recorded (level 2 records one stage only)
				float ms = 0.f;
				(void)hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]);
				out[i] += ms;
				per_stage[i]++;
			}
			calls++;
		}
		for (int i = 0; i < k; i++) if (per_stage[i]) out[i] /= (float)per_stage[i];
		return calls;
	}
};
// process-wide (autograd runs backward on its own thread); guarded by g_prof_mu
std::mutex g_prof_mu;
int g_prof = 0;                  // 0 off, 1 every stage boundary, 2 the two boundaries of composite_fwd only
uint32_t g_prof_fwd_mask = 0, g_prof_bwd_mask = 0;   // boundaries recorded at the current level
ProfLog g_fwd_log, g_bwd_log;
ProfSet* prof_next(ProfLog& log)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	const uint32_t mask = &log == &g_fwd_log ? g_prof_fwd_mask : g_prof_bwd_mask;
	return (g_prof && mask) ? log.next() : nullptr;
}

int fail(int code, const char* what, const char* file, int line, hipError_t e = hipSuccess)
{
	char buf[512];
	if (e != hipSuccess)
		snprintf(buf, sizeof(buf), "[gsrast] %s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
	else
		snprintf(buf, sizeof(buf), "[gsrast] %s (%s:%d)", what, file, line);
	g_err = buf;
	return code;
}

#define HIP_TRY(expr)                                                                  \
	do {                                                                               \
		hipError_t e_ = (expr);                                                        \
		if (e_ != hipSuccess) return fail(GSR_ERR_HIP, #expr, __FILE__, __LINE__, e_); \
	} while (0)

// after a kernel launch: catch launch errors; in debug mode also synchronise and surface
// execution errors per stage (the reference's CHECK_CUDA, auxiliary.h:166-173)
#define STAGE_CHECK(name, debug, stream)                                                       \
	do {                                                                                       \
		hipError_t e_ = hipGetLastError();                                                     \
		if (e_ != hipSuccess) return fail(GSR_ERR_HIP, name " launch", __FILE__, __LINE__, e_); \
		if (debug) {                                                                           \
			e_ = hipStreamSynchronize(stream);                                                 \
			if (e_ != hipSuccess) return fail(GSR_ERR_HIP, name, __FILE__, __LINE__, e_);      \
		}                                                                                      \
	} while (0)

bool query_device_ptr(const void* p)
{
	hipPointerAttribute_t attr;
	hipError_t e = hipPointerGetAttributes(&attr, p);
	if (e != hipSuccess) {
		(void)hipGetLastError();   // clear: plain host memory is reported as an error by some runtimes
		return false;
	}
	return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// hipPointerGetAttributes is a driver query of several microseconds and sits on the hot path (camera matrices and
// background of every call): the classification of the last few DEVICE pointers is remembered per thread (a training
// loop passes the same few tensors every iteration).  Host pointers (gaustudio's CPU `bg`) are queried every time.
bool is_device_ptr(const void* p)
{
	// only DEVICE answers are remembered: a device address handed out by hipMalloc / torch's caching allocator never
	// becomes a host heap address (separate address ranges), whereas a freed host address may well be returned by a
	// later malloc -- caching "host" would rest on allocator behaviour this library does not control
	thread_local const void* cache[8] = {};
	thread_local int next = 0;
	if (p == nullptr) return false;
	for (const void* e : cache)
		if (e == p) return true;
	const bool dev = query_device_ptr(p);
	if (dev) {
		cache[next] = p;
		next = (next + 1) & 7;
	}
	return dev;
}

// ---- tunables (gsr_set_option / environment), process-wide ----
int env_int(const char* name, int dflt)
{
	const char* v = getenv(name);
	return (v && *v) ? atoi(v) : dflt;
}
std::atomic<int> g_opt_tight{env_int("GSR_TIGHT_BINNING", 1)};     // bin into gs_tight_rect (0: the reference's squares)
std::atomic<int> g_opt_cull{env_int("GSR_CULL", 1)};               // composite_fwd wave culling + pcut pre-test
std::atomic<int> g_opt_fwd_variant{env_int("GSR_FWD_VARIANT", 0)}; // 0: per-quarter (4x4) instance lists, 1: per-wave (8x8)
std::atomic<int> g_opt_bwd_variant{env_int("GSR_BWD_VARIANT", -1)};   // -1: from gsr_selftest; bit 0: select on T
std::atomic<int> g_opt_speculative{env_int("GSR_SPECULATIVE", 1)}; // launch binning + compositing before R is known
std::atomic<int> g_opt_tile_order{env_int("GSR_TILE_ORDER", 1)};   // backward of a skewed frame: tiles longest walk first (0: always XCD-banded)
std::atomic<int> g_opt_bininfo{env_int("GSR_BININFO", 1)};         // binning passes read the 16-B binning record instead of the 64-B one (A/B; no result bit)
std::atomic<int> g_opt_band_lo{0}, g_opt_band_hi{0};               // tile rows [lo, hi) this process renders (hi <= 0: all)
// exp on the transcendental unit (v_exp_f32) in both compositing kernels.  DEFAULT ON since round 4: pinned directly against the
// reference's kernels and the CPU oracle (tests/test_gpu_ref.py, tests/test_gpu_fastexp_oracle.py), it differs from the
// reference at no more pixels than two builds of the reference differ from each other (profiles/r04_parity.json).
// GSR_FAST_EXP=0 / gsr_set_option("fast_exp", 0) / gsr_options.fast_exp = 0: the reproducible polynomial exp, bit-identical
// to the CPU oracle (the test suite's mode).
std::atomic<int> g_opt_fast_exp{env_int("GSR_FAST_EXP", 1)};

// What the forwards of this process ran with, by image buffer: a backward handed buffers of a forward in the OTHER exp mode would
// take other alpha >= 1/255 decisions than its forward (silently inconsistent gradients), and a forward_only forward kept nothing
// for a backward.  The authoritative record is the forward's own control word in the image buffer (GsCtl::opts); this host-side
// map makes the check free for a backward that follows its forward in the same process.  Round 6 (VERDICT r5 #7 / ADVICE r5): it was
// a 64-entry ring -- a plain-C-ABI caller with more than 64 forwards outstanding silently got the process default for the evicted
// ones.  Now: one entry per live image-buffer address (a newer forward into the same address replaces it), and a backward on an
// address the map does not know READS THE 4-BYTE CONTROL WORD from the device once (lookup_forward) instead of assuming anything.
// The map is dropped wholesale beyond kFwdModesMax entries (a caller that never reuses addresses): its forwards then take the
// read-back path, which is always correct.
struct FwdMode { int fast_exp; int skew; int forward_only; };
constexpr size_t kFwdModesMax = 16384;
std::unordered_map<const void*, FwdMode> g_fwd_modes;
std::mutex g_fwd_modes_mutex;
void remember_forward_mode(const void* img, int fast_exp, int forward_only)
{
	std::lock_guard<std::mutex> lock(g_fwd_modes_mutex);
	if (g_fwd_modes.size() >= kFwdModesMax && g_fwd_modes.find(img) == g_fwd_modes.end()) g_fwd_modes.clear();
	g_fwd_modes[img] = FwdMode{fast_exp, 0, forward_only};
}
// skew: the frame has a tile list many times longer than the mean -- its backward orders the tiles longest walk first (speed only)
void remember_forward_skew(const void* img, int skew)
{
	std::lock_guard<std::mutex> lock(g_fwd_modes_mutex);
	auto it = g_fwd_modes.find(img);
	if (it != g_fwd_modes.end()) it->second.skew = skew;
}
// The forward's record for these buffers: from the map, else from the device (one 4-byte read-back behind the stream's work; the
// result is remembered).  < 0: HIP error.
int lookup_forward(const char* image_buffer, size_t ctl_offset, hipStream_t s, FwdMode* out)
{
	{
		std::lock_guard<std::mutex> lock(g_fwd_modes_mutex);
		auto it = g_fwd_modes.find(image_buffer);
		if (it != g_fwd_modes.end()) { *out = it->second; return 0; }
	}
	uint32_t opts = 0;
	HIP_TRY(hipMemcpyAsync(&opts, image_buffer + ctl_offset + offsetof(GsCtl, opts), sizeof(opts), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	*out = FwdMode{(opts & GSR_CTL_OPT_FAST_EXP) ? 1 : 0, 0, (opts & GSR_CTL_OPT_FORWARD_ONLY) ? 1 : 0};
	std::lock_guard<std::mutex> lock(g_fwd_modes_mutex);
	if (g_fwd_modes.size() >= kFwdModesMax) g_fwd_modes.clear();
	g_fwd_modes[image_buffer] = *out;
	return 0;
}

// options of ONE call: the caller's gsr_options where given (>= 0), the process defaults elsewhere
struct Resolved {
	int tight, cull, fwd_variant, bwd_variant, speculative, band_lo, band_hi, fast_exp, forward_only;
	bool fast_exp_explicit;   // asked for by the caller's gsr_options (not the process default)
};
Resolved resolve_options(const gsr_options* o)
{
	gsr_options v;
	gsr_options_init(&v);
	if (o) {
		const size_t n = o->struct_bytes > 0 ? (size_t)o->struct_bytes : 0;
		memcpy(&v, o, n < sizeof(v) ? n : sizeof(v));
	}
	Resolved r;
	r.tight = v.tight_binning >= 0 ? v.tight_binning : g_opt_tight.load();
	r.cull = v.cull >= 0 ? v.cull : g_opt_cull.load();
	r.fwd_variant = v.fwd_variant >= 0 ? v.fwd_variant : g_opt_fwd_variant.load();
	r.bwd_variant = v.bwd_variant >= 0 ? v.bwd_variant : g_opt_bwd_variant.load();
	r.speculative = v.speculative >= 0 ? v.speculative : g_opt_speculative.load();
	if (v.tile_row_lo >= 0) { r.band_lo = v.tile_row_lo; r.band_hi = v.tile_row_hi; }
	else { r.band_lo = g_opt_band_lo.load(); r.band_hi = g_opt_band_hi.load(); }
	r.fast_exp = v.fast_exp >= 0 ? v.fast_exp : g_opt_fast_exp.load();
	r.fast_exp_explicit = v.fast_exp >= 0;
	r.forward_only = v.forward_only > 0 ? 1 : 0;
	return r;
}

// per device: capacity (instances) the binning buffer is allocated with while R is still in flight, and whether
// the last frame had a tile list long enough for the radix path (which needs the second key buffer)
struct DevState {
	std::atomic<uint32_t> cap{0};
	std::atomic<int> long_lists{0};
	std::atomic<int> skew{0};        // the last frame was skewed (a tile list > 4x the mean): composite_fwd runs the tiles longest list first
	std::atomic<int> selftest{-1};
	// pinned, device-mapped word the long-list sort stores 2 into when a work queue overflowed (SortQueueLayout bound violated: the
	// frame's point_list is not sorted).  Sticky: checked and cleared on entry of the next gsr_forward / gsr_backward (check_sticky).
	std::atomic<uint32_t*> sticky{nullptr};
};
DevState g_dev[16];
DevState& dev_state()
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
	return g_dev[dev];
}

uint32_t* sticky_word()
{
	DevState& ds = dev_state();
	uint32_t* p = ds.sticky.load();
	if (p) return p;
	static std::mutex m;
	std::lock_guard<std::mutex> lock(m);
	p = ds.sticky.load();
	if (!p) {
		if (hipHostMalloc((void**)&p, 64, hipHostMallocDefault) != hipSuccess) return nullptr;
		memset(p, 0, 64);
		ds.sticky.store(p);
	}
	return p;
}
// an error a kernel of an EARLIER call of this process raised after its call had returned
int check_sticky(const char* who)
{
	uint32_t* p = dev_state().sticky.load();
	if (p && *reinterpret_cast<volatile uint32_t*>(p) != 0u) {
		*reinterpret_cast<volatile uint32_t*>(p) = 0u;
		return fail(GSR_ERR_HIP, (std::string(who) + ": the long-list sort of an earlier gsr_forward on this device overflowed a work queue "
		                          "(SortQueueLayout bound violated): that frame's point_list was not sorted, results computed from it are invalid").c_str(),
		            __FILE__, __LINE__);
	}
	return 0;
}

// Camera staging: the four small inputs (view 16, proj 16, campos 3, bg 3 floats) may each live in host or
// device memory.  Host-resident ones travel as kernel arguments, device-resident ones are read by the kernel:
// ONE tiny launch instead of four serialized copies (and no pageable-memory H2D copy).
struct StageArgs {
	const float* dptr[4];   // device source or nullptr
	float host[4][16];      // host values when dptr[i] == nullptr
	int n[4];
	uint32_t* word_dst;     // optional: one 32-bit word stored by the same launch (the forward's options word)
	uint32_t word;
	uint32_t* zero_dst;     // optional: up to 64 words cleared by the same launch, BEFORE word_dst is stored (the control block)
	int zero_n;
};
__global__ void stage_cam_kernel(StageArgs a, float* dst0, float* dst1, float* dst2, float* dst3)
{
	float* dst[4] = {dst0, dst1, dst2, dst3};
	const int which = threadIdx.x >> 4, i = threadIdx.x & 15;
	if (dst[which] != nullptr && i < a.n[which]) dst[which][i] = a.dptr[which] ? a.dptr[which][i] : a.host[which][i];
	if (a.zero_dst != nullptr && (int)threadIdx.x < a.zero_n) a.zero_dst[threadIdx.x] = 0u;
	__syncthreads();
	if (threadIdx.x == 0 && a.word_dst != nullptr) *a.word_dst = a.word;
}
hipError_t stage_small(const float* const src[4], float* const dst[4], const int n[4], hipStream_t s,
                       uint32_t* word_dst = nullptr, uint32_t word = 0, uint32_t* zero_dst = nullptr, int zero_n = 0)
{
	StageArgs a;
	a.word_dst = word_dst;
	a.word = word;
	a.zero_dst = zero_dst;
	a.zero_n = zero_n;
	for (int k = 0; k < 4; k++) {
		a.n[k] = (dst[k] != nullptr) ? n[k] : 0;
		a.dptr[k] = nullptr;
		for (int i = 0; i < 16; i++) a.host[k][i] = 0.f;
		if (dst[k] == nullptr || src[k] == nullptr) continue;   // absent source -> zeros
		if (is_device_ptr(src[k])) a.dptr[k] = src[k];
		else for (int i = 0; i < n[k]; i++) a.host[k][i] = src[k][i];
	}
	hipLaunchKernelGGL(stage_cam_kernel, dim3(1), dim3(64), 0, s, a, dst[0], dst[1], dst[2], dst[3]);
	return hipGetLastError();
}

// One event per host thread and device for the forward's read-back: the host waits for the 16-byte copy only,
// not for what was enqueued behind it.
hipEvent_t readback_event()
{
	thread_local hipEvent_t ev[16] = {};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
	if (!ev[dev] && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) ev[dev] = nullptr;
	return ev[dev];
}

// pinned, device-mapped host block tile_scan_kernel mirrors the control words into (one per host thread)
GsCtl* pinned_ctl()
{
	thread_local GsCtl* p = nullptr;
	if (!p) {
		if (hipHostMalloc((void**)&p, 256, hipHostMallocDefault) != hipSuccess) p = nullptr;
	}
	return p;
}

// ---- optional roctx ranges around the stages: gsr_set_option("roctx", 1) / GSR_ROCTX=1 makes every
// stage of gsr_forward / gsr_backward a named range in rocprofv3 --marker-trace / rocprof-sys timelines.  The marker
// library is looked up at run time (libroctx64.so, then librocprofiler-sdk-roctx.so): no link-time dependency, and a
// machine without it simply gets no ranges. ----
std::atomic<int> g_opt_roctx{env_int("GSR_ROCTX", 0)};
struct Roctx {
	int (*push)(const char*) = nullptr;
	int (*pop)() = nullptr;
	Roctx()
	{
		for (const char* lib : {"libroctx64.so", "librocprofiler-sdk-roctx.so", "libroctx64.so.4"}) {
			void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
			if (!h) continue;
			push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
			pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
			if (push && pop) return;
			push = nullptr; pop = nullptr;
		}
	}
};
Roctx& roctx()
{
	static Roctx r;
	return r;
}

// stage boundaries of one call: HIP events for the per-stage times (gsr_set_profiling) and, when asked for, roctx ranges
struct Timer {
	ProfSet* set;
	hipStream_t s;
	const char* const* names;
	int k = 0, ord = 0;
	uint32_t mask = 0x3fu;
	bool open = false, ranges;
	Timer(ProfSet* p, hipStream_t st, const char* const* stage_names = nullptr, uint32_t boundary_mask = 0x3fu)
	    : set(p), s(st), names(stage_names), mask(boundary_mask), ranges(stage_names != nullptr && g_opt_roctx.load() != 0 && roctx().push != nullptr) {}
	~Timer()
	{
		if (open) roctx().pop();
	}
	void mark()
	{
		if (set && ord < 6 && ((mask >> ord) & 1u)) {
			(void)hipEventRecord(set->ev[ord], s);
			set->have |= 1u << ord;
		}
		ord++;
		if (ranges) {
			if (open) roctx().pop();
			open = names[k] != nullptr;
			if (open) roctx().push(names[k++]);
		}
	}
};
const char* const kFwdStages[] = {"gsr.preprocess_fwd", "gsr.scan", "gsr.scatter", "gsr.sort", "gsr.composite_fwd", nullptr, nullptr, nullptr};
const char* const kBwdStages[] = {"gsr.composite_bwd", "gsr.preprocess_bwd", nullptr, nullptr};

__global__ __launch_bounds__(256) void fill_empty_outputs_kernel(size_t HW, float* out_color, float* out_depth,
                                                                 float* out_median, float* out_opacity)
{
	// P == 0: the reference launches nothing and returns its torch::full(0.0) images (rasterize_points.cu:67-84);
	// the 15.0 median sentinel only appears when the render kernel runs (P > 0 with empty tiles)
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= HW) return;
	out_color[i] = 0.f; out_color[HW + i] = 0.f; out_color[2 * HW + i] = 0.f;
	out_depth[i] = 0.f;
	out_median[i] = 0.f; out_median[HW + i] = 0.f; out_median[2 * HW + i] = 0.f;
	out_opacity[i] = 0.f;
}

__global__ __launch_bounds__(256) void inspect_geometry_kernel(int P, const int* radii, const GsRec* recs,
                                                               const uint32_t* tt, float* means2D, float* depths,
                                                               float* conic_opacity, float* rgb, unsigned char* clamped,
                                                               uint32_t* tiles_touched)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= P) return;
	const bool vis = radii[idx] > 0;
	GsRec r;
	if (vis) r = recs[idx];
	if (means2D) { means2D[2 * idx] = vis ? r.q0.x : 0.f; means2D[2 * idx + 1] = vis ? r.q0.y : 0.f; }
	if (depths) depths[idx] = vis ? r.q1.z : 0.f;
	if (conic_opacity) {
		conic_opacity[4 * idx + 0] = vis ? -2.f * r.q0.z : 0.f;
		conic_opacity[4 * idx + 1] = vis ? -r.q0.w : 0.f;
		conic_opacity[4 * idx + 2] = vis ? -2.f * r.q1.x : 0.f;
		conic_opacity[4 * idx + 3] = vis ? r.q1.y : 0.f;
	}
	if (rgb) { rgb[3 * idx] = vis ? r.q2.x : 0.f; rgb[3 * idx + 1] = vis ? r.q2.y : 0.f; rgb[3 * idx + 2] = vis ? r.q2.z : 0.f; }
	if (clamped) {
		for (int ch = 0; ch < 3; ch++) clamped[3 * idx + ch] = vis ? (unsigned char)((r.q3.z >> ch) & 1u) : 0;
	}
	if (tiles_touched) tiles_touched[idx] = vis ? tt[idx] : 0u;
}

__global__ __launch_bounds__(256) void inspect_image_kernel(int gx, int W, int H, const float* tT, const uint32_t* tN,
                                                            float* final_T, uint32_t* n_contrib)
{
	const int tile = blockIdx.x, tid = threadIdx.x;
	int lx, ly;
	gs_pixel_of_thread(tid, lx, ly);
	const int px = (tile % gx) * GSR_BLOCK_X + lx, py = (tile / gx) * GSR_BLOCK_Y + ly;
	if (px >= W || py >= H) return;
	const size_t pix = (size_t)W * py + px;
	if (final_T) final_T[pix] = tT[(size_t)tile * GSR_TILE_PIX + tid];
	if (n_contrib) n_contrib[pix] = tN[(size_t)tile * GSR_TILE_PIX + tid];
}

}  // namespace

extern "C" {

int gsr_abi_version(void) { return 6; }

void gsr_options_init(gsr_options* opt)
{
	if (!opt) return;
	opt->struct_bytes = (int32_t)sizeof(gsr_options);
	opt->tight_binning = opt->cull = opt->fwd_variant = opt->bwd_variant = opt->speculative = -1;
	opt->tile_row_lo = opt->tile_row_hi = -1;
	opt->fast_exp = -1;
	opt->forward_only = -1;
}

const char* gsr_last_error(void) { return g_err.c_str(); }

void gsr_set_profiling(int enable)
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	g_prof = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
	g_prof_fwd_mask = g_prof == 1 ? 0x3fu : (g_prof == 2 ? 0x30u : 0u);   // level 2: boundaries 4 and 5 = composite_fwd
	g_prof_bwd_mask = g_prof == 1 ? 0x3fu : 0u;
	g_fwd_log.used = 0;
	g_bwd_log.used = 0;
}

int gsr_last_forward_ms(float ms[5])
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	return g_fwd_log.mean(ms, 5);
}
int gsr_last_backward_ms(float ms[2])
{
	std::lock_guard<std::mutex> lk(g_prof_mu);
	return g_bwd_log.mean(ms, 2);
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* stream)
{
	(void)projmatrix;   // the reference computes p_proj but only tests p_view.z (auxiliary.h:147-154)
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (P <= 0) return GSR_OK;
	if (!means3D || !viewmatrix || !present) return fail(GSR_ERR_ARG, "gsr_mark_visible: NULL argument", __FILE__, __LINE__);
	const float* view = viewmatrix;
	float* tmp = nullptr;
	if (!is_device_ptr(viewmatrix)) {
		HIP_TRY(hipMallocAsync((void**)&tmp, 16 * sizeof(float), s));
		HIP_TRY(hipMemcpyAsync(tmp, viewmatrix, 16 * sizeof(float), hipMemcpyHostToDevice, s));
		view = tmp;
	}
	launch_mark_visible(P, means3D, view, present, s);
	hipError_t e = hipGetLastError();
	if (tmp) (void)hipFreeAsync(tmp, s);
	if (e != hipSuccess) return fail(GSR_ERR_HIP, "mark_visible launch", __FILE__, __LINE__, e);
	return GSR_OK;
}

static int forward_impl(const gsr_options* opt, gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc, void* binning_ctx,
                        gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                        int width, int height, const float* means3D, const float* shs, const float* shs_rest,
                        const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, int activation_flags,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                        float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_median_depth,
                        float* out_opacity, int* radii, int debug, void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (width <= 0 || height <= 0) return fail(GSR_ERR_ARG, "gsr_forward: bad image size", __FILE__, __LINE__);
	if (!out_color || !out_depth || !out_median_depth || !out_opacity)
		return fail(GSR_ERR_ARG, "gsr_forward: NULL output", __FILE__, __LINE__);
	const size_t HW = (size_t)width * height;
	if (P <= 0) {
		// rasterize_points.cu:84: nothing is launched, images stay at their fill value
		hipLaunchKernelGGL(fill_empty_outputs_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, s, HW,
		                   out_color, out_depth, out_median_depth, out_opacity);
		STAGE_CHECK("fill_empty_outputs", debug, s);
		return 0;
	}
	if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos || !radii)
		return fail(GSR_ERR_ARG, "gsr_forward: NULL required input", __FILE__, __LINE__);
	if (!colors_precomp && !shs)
		return fail(GSR_ERR_ARG, "For non-RGB, provide precomputed Gaussian colors!", __FILE__, __LINE__);
	if (!cov3D_precomp && (!scales || !rotations))
		return fail(GSR_ERR_ARG, "gsr_forward: need scales+rotations or cov3D_precomp", __FILE__, __LINE__);
	if (!colors_precomp && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
		return fail(GSR_ERR_ARG, "gsr_forward: SH degree does not fit the stored coefficients", __FILE__, __LINE__);
	if (!geometry_alloc || !binning_alloc || !image_alloc)
		return fail(GSR_ERR_ARG, "gsr_forward: NULL allocator", __FILE__, __LINE__);

	{
		const int rc = check_sticky("gsr_forward");
		if (rc < 0) return rc;
	}
	Resolved ro = resolve_options(opt);
#ifndef GSR_AB_VARIANTS
	if (ro.fwd_variant == 1)
		return fail(GSR_ERR_ARG, "gsr_forward: fwd_variant 1 (the per-wave A/B kernel) is not in this build (make AB=1; gsr_get_option(\"ab_variants\"))", __FILE__, __LINE__);
#endif
	if (ro.fast_exp && ro.fwd_variant == 1) {
		// the per-wave A/B kernel has no v_exp_f32 form.  Asked for both: an error.  fast_exp merely inherited from the
		// process default (on since round 4): the A/B switch wins and the call runs in the reproducible mode;
		// the backward follows the forward's recorded mode the same way
		if (ro.fast_exp_explicit)
			return fail(GSR_ERR_ARG, "gsr_forward: fast_exp needs the per-quarter compositing kernels (fwd_variant 0)", __FILE__, __LINE__);
		ro.fast_exp = 0;
	}
	const GeomLayout gl((size_t)P);
	const ImgLayout il(width, height);
	if (il.gx > 65535 || il.gy > 65535) return fail(GSR_ERR_ARG, "gsr_forward: image too large", __FILE__, __LINE__);
	char* geom = geometry_alloc(geometry_ctx, gl.total);
	// the per-chunk tile histogram of the atomics-free binning lives behind the image state proper
	const bool lds_bin = bin_lds_path_ok(il.T);
	const size_t hm_off = il.total;
	char* img = image_alloc(image_ctx, il.total + (lds_bin ? align_up(bin_hist_bytes(P, il.T)) : 0));
	if (!geom || !img) return fail(GSR_ERR_ALLOC, "gsr_forward: allocator returned NULL", __FILE__, __LINE__);

	remember_forward_mode(img, ro.fast_exp != 0, ro.forward_only);

	GsCam* cam = reinterpret_cast<GsCam*>(geom + gl.cam);
	GsRec* recs = reinterpret_cast<GsRec*>(geom + gl.recs);
	GsCtl* ctl = reinterpret_cast<GsCtl*>(img + il.ctl);
	uint2* ranges = reinterpret_cast<uint2*>(img + il.ranges);
	uint32_t* tile_count = reinterpret_cast<uint32_t*>(img + il.tile_count);
	float* final_T = reinterpret_cast<float*>(img + il.final_T);
	uint32_t* n_contrib = reinterpret_cast<uint32_t*>(img + il.n_contrib);
	uint32_t* med_pos = reinterpret_cast<uint32_t*>(img + il.med_pos);

	Timer tm(prof_next(g_fwd_log), s, kFwdStages, g_prof_fwd_mask);
	// control words (+ tile counters: only the atomic-counter binning path needs them cleared -- the chunked path
	// writes every tile's count and range itself), then the camera block and the options this call runs with: one tiny
	// launch, which also clears the 8 control words
	static_assert(sizeof(GsCtl) == 8 * sizeof(uint32_t), "stage_cam_kernel clears GsCtl word by word");
	if (!lds_bin) HIP_TRY(hipMemsetAsync(img + il.ctl, 0, il.final_T - il.ctl, s));   // ctl + ranges + tile_count
	{
		const float* const src[4] = {viewmatrix, projmatrix, cam_pos, background};
		float* const dst[4] = {cam->view, cam->proj, cam->campos, cam->bg};
		const int n[4] = {16, 16, 3, 3};
		const bool banded = ro.band_hi > 0 || ro.band_lo > 0;
		const uint32_t word = (ro.fast_exp ? GSR_CTL_OPT_FAST_EXP : 0u) | (ro.tight ? GSR_CTL_OPT_TIGHT : 0u) |
		                      (ro.cull ? GSR_CTL_OPT_CULL : 0u) | (ro.fwd_variant == 1 ? GSR_CTL_OPT_WAVE_LISTS : 0u) |
		                      (banded ? GSR_CTL_OPT_BAND : 0u) | (ro.forward_only ? GSR_CTL_OPT_FORWARD_ONLY : 0u);
		HIP_TRY(stage_small(src, dst, n, s, &ctl->opts, word, reinterpret_cast<uint32_t*>(ctl), 8));
	}

	FwdArgs a;
	a.P = P; a.D = D; a.M = M; a.W = width; a.H = height;
	a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.opacities = opacities;
	a.scales = scales; a.scale_modifier = scale_modifier; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp;
	a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.prefiltered = prefiltered;
	a.shs_rest = shs_rest; a.act = activation_flags;
	a.tight = ro.tight != 0;
	a.band_lo = ro.band_lo > 0 ? ro.band_lo : 0;
	a.band_hi = ro.band_hi;

	tm.mark();
	uint32_t* tiles_touched = reinterpret_cast<uint32_t*>(geom + gl.tiles_touched);
	uint32_t* bsums = reinterpret_cast<uint32_t*>(geom + gl.bsums);
	uint32_t* refsums = reinterpret_cast<uint32_t*>(geom + gl.refsums);
	uint32_t* Hm = reinterpret_cast<uint32_t*>(img + hm_off);
	uint4* binfo_w = reinterpret_cast<uint4*>(geom + gl.binfo);   // always written: the SH backward reads its clamp word
	const uint4* binfo = (lds_bin && g_opt_bininfo.load() != 0) ? binfo_w : nullptr;   // read by the binning passes (A/B)
	launch_preprocess_fwd(a, cam, il, radii, recs, ro.forward_only ? nullptr : reinterpret_cast<float*>(geom + gl.shjac), binfo_w, tiles_touched, bsums, refsums,
	                      lds_bin ? nullptr : tile_count, ctl, s);
	STAGE_CHECK("preprocess_fwd", debug, s);
	tm.mark();
	if (lds_bin) {
		launch_bin_hist(P, il.gx, il.T, tiles_touched, recs, binfo, Hm, tile_count, s);
		STAGE_CHECK("bin_hist", debug, s);
	}
	GsCtl* host = pinned_ctl();
	if (!host) return fail(GSR_ERR_HIP, "hipHostMalloc", __FILE__, __LINE__);
	launch_tile_scan(il.T, tile_count, ranges, (int)gl.nblk, bsums, refsums, ctl, host, s);
	STAGE_CHECK("tile_scan", debug, s);
	// The instance counts are now in pinned host memory (written by the kernel itself: no copy command in the
	// stream).  The host needs them to return num_rendered and to check that the binning buffer is large enough --
	// but the DEVICE does not have to wait for the host: everything below is enqueued against a buffer of the
	// remembered capacity before the event is waited on, so the GPU never idles across the read-back (the reference
	// blocks mid-forward, rasterizer_impl.cu:283-284).  The kernels leave without touching memory if the capacity
	// turns out too small; the host then allocates the real size and enqueues them again (rare: the capacity only
	// grows).
	hipEvent_t ev = readback_event();
	if (ev) HIP_TRY(hipEventRecord(ev, s));
	// Gaussian-major row offsets for the backward: computed by the chunked scatter kernel on its way (same Gaussians, same
	// chunks); the other paths launch the small kernel of their own
	uint32_t* goff = reinterpret_cast<uint32_t*>(geom + gl.goff);
	bool goff_done = false;
	tm.mark();

	DevState& ds = dev_state();
	const bool nocull = ro.cull == 0;
	const bool wave_lists = ro.fwd_variant == 1;
	bool skew_now = g_opt_tile_order.load() != 0 && ds.skew.load() != 0;
	auto launch_rest = [&](uint32_t cap, int long_level) -> int {
		const bool with_long = long_level > 0;
		const BinLayout bl((size_t)cap, with_long, il.T);
		char* bin = binning_alloc(binning_ctx, bl.total);
		if (!bin) return fail(GSR_ERR_ALLOC, "gsr_forward: binning allocator returned NULL", __FILE__, __LINE__);
		uint64_t* keys = reinterpret_cast<uint64_t*>(bin + bl.keys);
		uint64_t* keys2 = reinterpret_cast<uint64_t*>(bin + bl.keys2);
		uint32_t* point_list = reinterpret_cast<uint32_t*>(bin + bl.point_list);
		if (cap > 0) {
			if (lds_bin) {
				launch_bin_scatter2(P, il.gx, il.T, tiles_touched, recs, binfo, Hm, ranges, keys, bsums, goff, ctl, cap, s);
				goff_done = true;
			}
			else
				launch_bin_scatter(P, il.gx, radii, tiles_touched, recs, ranges, tile_count, keys, ctl, cap, s);
			STAGE_CHECK("bin_scatter", debug, s);
		}
		if (!goff_done) {
			launch_goff_apply(P, tiles_touched, bsums, goff, s);
			STAGE_CHECK("goff_apply", debug, s);
			goff_done = true;
		}
		tm.mark();
		if (cap > 0) {
			launch_tile_sort(il.T, true, long_level, ranges, keys, keys2, point_list, bin + bl.queue, (size_t)cap, ctl, cap, long_level > 0 ? sticky_word() : nullptr, s);
			STAGE_CHECK("tile_sort", debug, s);
		}
		tm.mark();
		const uint32_t* order = nullptr;
		if (skew_now && cap > 0) {   // (any permutation of the tiles is valid: a wrong guess of the regime costs or wastes two small launches, nothing else)
			uint32_t* o = reinterpret_cast<uint32_t*>(img + il.tile_order);
			launch_tile_order_fwd(il.T, ranges, o, ctl, cap, s);
			order = o;
		}
		launch_composite_fwd(il, width, height, ranges, point_list, recs, out_color, out_depth, out_median_depth,
		                     out_opacity, final_T, n_contrib, med_pos, ctl, cap,
		                     long_level >= 2 ? 0xffffffffu : (long_level == 1 ? GSR_SORT_GIANT : GSR_SORT_LDS_MAX), nocull, wave_lists, ro.fast_exp != 0, order,
		                     tile_count /* dead after binning: instances staged per tile, gsr_inspect_staged */, s);
		STAGE_CHECK("composite_fwd", debug, s);
		tm.mark();
		return 0;
	};

	const uint32_t cap0 = ds.cap.load();
	const int long0 = ds.long_lists.load();   // sort regime of the previous frame: 0 / 1 (lists > 1024 keys) / 2 (> GSR_SORT_GIANT keys)
	const bool speculate = !debug && ro.speculative != 0 && cap0 > 0;
	if (speculate) {
		const int rc = launch_rest(cap0, long0);
		if (rc < 0) return rc;
	}
	if (ev) HIP_TRY(hipEventSynchronize(ev));
	else HIP_TRY(hipStreamSynchronize(s));
	const uint32_t Rb = host->num_binned, max_tile = host->max_tile_count;
	if (host->err_prefiltered)
		return fail(GSR_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!",
		            __FILE__, __LINE__);
	if ((host->err_overflow & 1u) || host->ref_rendered > 0x7fffffffu || Rb > 0x7fffffffu)
		return fail(GSR_ERR_ARG, "gsr_forward: more than 2^31 - 1 (tile, Gaussian) instances", __FILE__, __LINE__);
	const int need_long = max_tile > GSR_SORT_GIANT ? 2 : (max_tile > GSR_SORT_LDS_MAX ? 1 : 0);
	if (!speculate || Rb > cap0 || need_long > long0) {
		// first call on this device, debug mode, or the speculation missed: exact size, launched now
		const int rc = launch_rest(Rb, need_long);
		if (rc < 0) return rc;
	}
	// capacity for the next frame: this frame's count + 25 % headroom; a larger remembered capacity (one 4K close-up in
	// between training views) decays by 1/8 per frame towards it instead of pinning 12-20 B x cap per live forward for
	// the rest of the process
	const uint32_t want = Rb + Rb / 4 + 4096u;
	uint32_t cur = ds.cap.load();
	for (;;) {
		const uint32_t next = cur < want ? want : (cur - (cur - want) / 8 > want + 4096u ? cur - (cur - want) / 8 : want);
		if (next == cur || ds.cap.compare_exchange_weak(cur, next)) break;
	}
	ds.long_lists.store(need_long);
	if (debug) {
		// every stage above was synchronised (STAGE_CHECK): the long-list sort's queue-overflow flag is final
		GsCtl c;
		HIP_TRY(hipMemcpyAsync(&c, img + il.ctl, sizeof(GsCtl), hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		if (c.err_overflow & 2u)
			return fail(GSR_ERR_HIP, "gsr_forward: the long-list sort overflowed a work queue (SortQueueLayout bound violated)", __FILE__, __LINE__);
	}
	// a skewed frame (longest list > 1024 keys and > 4x the mean): its backward runs the tiles longest walk first
	const int skew = (g_opt_tile_order.load() != 0 && max_tile > GSR_SORT_LDS_MAX && (uint64_t)max_tile * (uint64_t)il.T > 4ull * Rb) ? 1 : 0;
	remember_forward_skew(img, skew);
	ds.skew.store(skew);
	return (int)host->ref_rendered;
}

int gsr_forward(gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc, void* binning_ctx,
                gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth,
                float* out_median_depth, float* out_opacity, int* radii, int debug, void* stream)
{
	return forward_impl(nullptr, geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M,
	                    background, width, height, means3D, shs, nullptr, colors_precomp, opacities, scales,
	                    scale_modifier, rotations, cov3D_precomp, 0, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
	                    prefiltered, out_color, out_depth, out_median_depth, out_opacity, radii, debug, stream);
}

int gsr_forward_raw(gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc, void* binning_ctx,
                    gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                    int height, const float* means3D, const float* f_dc, const float* f_rest, const float* raw_opacities,
                    const float* raw_scales, float scale_modifier, const float* raw_rotations, int activation_flags,
                    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                    float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_median_depth,
                    float* out_opacity, int* radii, int debug, void* stream)
{
	if (P > 0 && (!f_dc || (M > 1 && !f_rest) || !raw_scales || !raw_rotations))
		return fail(GSR_ERR_ARG, "gsr_forward_raw: f_dc, f_rest, raw_scales and raw_rotations are required", __FILE__, __LINE__);
	return forward_impl(nullptr, geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M,
	                    background, width, height, means3D, f_dc, M > 1 ? f_rest : nullptr, nullptr, raw_opacities,
	                    raw_scales, scale_modifier, raw_rotations, nullptr, activation_flags, viewmatrix, projmatrix, cam_pos,
	                    tan_fovx, tan_fovy, prefiltered, out_color, out_depth, out_median_depth, out_opacity, radii, debug,
	                    stream);
}

int gsr_forward_ex(const gsr_options* opt, gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc,
                   void* binning_ctx, gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                   const float* background, int width, int height, const float* means3D, const float* shs,
                   const float* shs_rest, const float* colors_precomp, const float* opacities, const float* scales,
                   float scale_modifier, const float* rotations, const float* cov3D_precomp, int activation_flags,
                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                   int prefiltered, float* out_color, float* out_depth, float* out_median_depth, float* out_opacity,
                   int* radii, int debug, void* stream)
{
	return forward_impl(opt, geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M,
	                    background, width, height, means3D, shs, shs_rest, colors_precomp, opacities, scales, scale_modifier,
	                    rotations, cov3D_precomp, activation_flags, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
	                    prefiltered, out_color, out_depth, out_median_depth, out_opacity, radii, debug, stream);
}

size_t gsr_geometry_bytes(int P) { return GeomLayout((size_t)(P > 0 ? P : 0)).total; }
size_t gsr_image_bytes(int width, int height) { return (width > 0 && height > 0) ? ImgLayout(width, height).total : 0; }

size_t gsr_backward_scratch_bytes(int P, int R)
{
	return BwdLayout((size_t)(P > 0 ? P : 0), (size_t)(R > 0 ? R : 0)).total;
}

// composite_bwd variant: bit 0 = keep a select on T (devices where v_rcp_f32(1.0) != 1.0)
static int bwd_variant(int opt, hipStream_t s)
{
	if (opt >= 0) return opt;
	DevState& ds = dev_state();
	int st = ds.selftest.load();
	if (st < 0) {
		st = gsr_selftest((void*)s);
		if (st < 0) st = 0;
		ds.selftest.store(st);
	}
	return (st & 3) == 3 ? 0 : 1;
}

static int backward_impl(const gsr_options* opt, int parts, int sh_g0, int sh_g1, int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
                         const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         int activation_flags, float tan_fovx, float tan_fovy, const int* radii, const char* geom_buffer,
                         const char* binning_buffer, const char* image_buffer, const float* dL_dpix,
                         const float* dL_dpix_depth, const float* dL_dpix_median_depth,
                         const float* dL_dpix_final_opacity, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                         float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest, float* dL_dscale,
                         float* dL_drot, char* scratch, int debug, void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (P <= 0) return GSR_OK;   // rasterize_points.cu:171
	if (!geom_buffer || !image_buffer || !binning_buffer || !scratch || !radii || !means3D)
		return fail(GSR_ERR_ARG, "gsr_backward: NULL buffer", __FILE__, __LINE__);
	// any of the four upstream image gradients may be NULL = "the loss does not use that output" = zero: nothing is loaded for
	// it (and the caller materialises no zero plane); colour alone takes a compositing kernel specialised on it
	const bool sh_colors = (parts & GSR_BWD_PART_SH_COLORS) != 0;
	const int colors_early = ((parts & GSR_BWD_PART_COLORS_EARLY) && sh_colors) ? GSR_PART_COLORS_EARLY : 0;
	if (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || (!dL_dcov3D && cov3D_precomp) || !dL_dscale || !dL_drot ||
	    (M > 0 && !dL_dsh && !sh_colors))
		return fail(GSR_ERR_ARG, "gsr_backward: NULL output", __FILE__, __LINE__);
	if (sh_colors && (!shs || shs_rest || colors_precomp))
		return fail(GSR_ERR_ARG, "gsr_backward: GSR_BWD_PART_SH_COLORS needs SH colours in one [P,M,3] tensor", __FILE__, __LINE__);
	if (shs && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
		return fail(GSR_ERR_ARG, "gsr_backward: SH degree does not fit the stored coefficients", __FILE__, __LINE__);

	const GeomLayout gl((size_t)P);
	const ImgLayout il(width, height);
	const BinLayout bl((size_t)(R > 0 ? R : 0));
	const GsCam* cam = reinterpret_cast<const GsCam*>(geom_buffer + gl.cam);
	const GsRec* recs = reinterpret_cast<const GsRec*>(geom_buffer + gl.recs);
	const uint2* ranges = reinterpret_cast<const uint2*>(image_buffer + il.ranges);
	const float* final_T = reinterpret_cast<const float*>(image_buffer + il.final_T);
	const uint32_t* n_contrib = reinterpret_cast<const uint32_t*>(image_buffer + il.n_contrib);
	const uint32_t* med_pos = reinterpret_cast<const uint32_t*>(image_buffer + il.med_pos);
	const uint32_t* point_list = reinterpret_cast<const uint32_t*>(binning_buffer + bl.point_list);
	const BwdLayout wl((size_t)P, (size_t)(R > 0 ? R : 0));
	const uint32_t* goff = reinterpret_cast<const uint32_t*>(geom_buffer + gl.goff);   // scanned by the forward
	// the clamp bits of every visible Gaussian (word 2 of its 16-B binning record, which the forward always writes): what the SH stage
	// needs of the record -- a quarter sector instead of a 32-B sector of the 64-B record per Gaussian
	const uint32_t* clampw = reinterpret_cast<const uint32_t*>(geom_buffer + gl.binfo) + 2;
	float* bg_dev = reinterpret_cast<float*>(scratch + wl.bg);
	float* rows = reinterpret_cast<float*>(scratch + wl.rows);
	uint8_t* row_flags = reinterpret_cast<uint8_t*>(scratch + wl.flags);

	// banded backward (GSR_BWD_PART_BAND_FIRST / _SECOND; sh_g0 carries the split tile row): see include/gsrast.h
	const int band_phase = (parts & GSR_BWD_PART_BAND_FIRST) ? 0 : ((parts & GSR_BWD_PART_BAND_SECOND) ? 1 : -1);
	int band_split = 0;
	if (band_phase >= 0) {
		if ((parts & GSR_BWD_PART_BAND_FIRST) && (parts & GSR_BWD_PART_BAND_SECOND))
			return fail(GSR_ERR_ARG, "gsr_backward: GSR_BWD_PART_BAND_FIRST and _SECOND are two calls", __FILE__, __LINE__);
		if (!(parts & GSR_BWD_PART_MAIN) || (parts & GSR_BWD_PART_SH))
			return fail(GSR_ERR_ARG, "gsr_backward: a band call runs GSR_BWD_PART_MAIN only (the SH stage follows the second band as its own call)", __FILE__, __LINE__);
		band_split = sh_g0 < 0 ? 0 : (sh_g0 > il.gy ? il.gy : sh_g0);
	}
	BwdArgs a;
	a.cls_mode = band_phase + 1;
	a.cls_split = band_split;
	a.P = P; a.D = D; a.M = M; a.W = width; a.H = height;
	a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.scales = scales;
	a.scale_modifier = scale_modifier; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp;
	a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.radii = radii;
	a.shs_rest = shs_rest; a.act = activation_flags;

	// a forward_only forward kept nothing for a backward (no shjac, no per-pixel state): refused whichever stages are asked for --
	// the SH stage alone reads shjac too.  Host-side memory of the recent forwards; debug mode reads the
	// forward's own record in the image buffer as well (an entry evicted from the host table is otherwise unchecked).
	FwdMode fwd;
	{
		int rc = check_sticky("gsr_backward");
		if (rc < 0) return rc;
		rc = lookup_forward(image_buffer, il.ctl, s, &fwd);
		if (rc < 0) return rc;
	}
	if (fwd.forward_only)
		return fail(GSR_ERR_ARG, "gsr_backward: these buffers come from a forward_only forward (gsr_options.forward_only): it kept nothing for a backward", __FILE__, __LINE__);
	if (debug) {
		GsCtl c;
		HIP_TRY(hipMemcpyAsync(&c, image_buffer + il.ctl, sizeof(GsCtl), hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		if (c.opts & GSR_CTL_OPT_FORWARD_ONLY)
			return fail(GSR_ERR_ARG, "gsr_backward: these buffers come from a forward_only forward (gsr_options.forward_only): it kept nothing for a backward", __FILE__, __LINE__);
		if (c.err_overflow & 2u)
			return fail(GSR_ERR_HIP, "gsr_backward: the forward's long-list sort overflowed a work queue (point_list is not sorted)", __FILE__, __LINE__);
	}
	if (!(parts & GSR_BWD_PART_MAIN)) {
		// SH stage alone over a Gaussian range (the caller interleaves a collective per chunk, gaustudio_amd/parallel.py)
		launch_preprocess_bwd(a, cam, recs, clampw, reinterpret_cast<const float*>(geom_buffer + gl.shjac), goff, rows, nullptr, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D,
		                      dL_dsh, dL_dsh_rest, dL_dscale, dL_drot, GSR_PART_SH | (sh_colors ? GSR_PART_SH_COLORS : 0) | colors_early, sh_g0, sh_g1, s);
		STAGE_CHECK("preprocess_bwd_sh", debug, s);
		return GSR_OK;
	}

	// The background is re-staged here because the reference reads the backward's own `background`
	// argument (backward.cu:584-587), which the forward never dereferences.
	Timer tm(prof_next(g_bwd_log), s, kBwdStages, g_prof_bwd_mask);
	// long lists (average > GSR_FLAG_AVG entries per tile): one validity byte per instance row, set by composite_bwd for the
	// rows it writes; short lists: no flags, composite_bwd zeroes the rows of the entries its walk does not reach
	const bool flagged = (size_t)(R > 0 ? R : 0) > (size_t)il.T * GSR_FLAG_AVG;
	// the background and the regime word (word 8 of the scratch's bg block, for gsr_inspect_backward_sums) ride in
	// composite_bwd's argument block (GsBg): a device-resident background is read by the kernel where it lies, host
	// values are copied into the arguments -- no staging launch in front of the kernel.  Without instances the kernel
	// does not run: the word is then stored by the small staging launch.
	GsBg bgv;
	bgv.dptr = (background != nullptr && is_device_ptr(background)) ? background : nullptr;
	for (int i = 0; i < 3; i++) bgv.host[i] = (background != nullptr && bgv.dptr == nullptr) ? background[i] : 0.f;
	bgv.flag_dst = reinterpret_cast<uint32_t*>(bg_dev + 8);
	bgv.flag = flagged ? 1u : 0u;
	bgv.tile_order = nullptr;
	bgv.tile_lo = band_phase == 1 ? (uint32_t)band_split * (uint32_t)il.gx : 0u;
	bgv.tile_hi = band_phase == 0 ? (uint32_t)band_split * (uint32_t)il.gx : 0xffffffffu;
	if (R <= 0) {
		const float* const src[4] = {nullptr, nullptr, nullptr, nullptr};
		float* const dst[4] = {nullptr, nullptr, nullptr, nullptr};
		const int n[4] = {0, 0, 0, 0};
		HIP_TRY(stage_small(src, dst, n, s, bgv.flag_dst, bgv.flag));
	}
	if (flagged) {
		if (band_phase != 1) HIP_TRY(hipMemsetAsync(row_flags, 0, (size_t)R, s));   // (the second band adds to the first band's flags)
	} else
		row_flags = nullptr;

	tm.mark();
	if (R > 0) {
		Resolved ro = resolve_options(opt);
		{
			const int fwd_mode = fwd.fast_exp;
			// fast_exp not named by the caller: the mode the forward of these buffers ran in (which may itself have left the
			// process default for an A/B variant without a v_exp_f32 kernel, see forward_impl) -- from the host-side map or, for
			// buffers it does not know, from the forward's own control word (lookup_forward): never the process default
			if (!ro.fast_exp_explicit) ro.fast_exp = fwd_mode;
			if (fwd_mode != (ro.fast_exp != 0))
				return fail(GSR_ERR_ARG, "gsr_backward: fast_exp differs from the forward that produced these buffers", __FILE__, __LINE__);
		}
		if (debug) {
			// the forward recorded what it ran with: a backward in another exp mode would take other alpha >= 1/255 decisions
			GsCtl c;
			HIP_TRY(hipMemcpyAsync(&c, image_buffer + il.ctl, sizeof(GsCtl), hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			if (!ro.fast_exp_explicit) ro.fast_exp = (c.opts & GSR_CTL_OPT_FAST_EXP) ? 1 : 0;   // (the host-side memory may have forgotten this forward)
			if (((c.opts & GSR_CTL_OPT_FAST_EXP) != 0u) != (ro.fast_exp != 0))
				return fail(GSR_ERR_ARG, "gsr_backward: fast_exp differs from the forward that produced these buffers", __FILE__, __LINE__);
		}
		int variant = bwd_variant(ro.bwd_variant, s);
#ifndef GSR_AB_VARIANTS
		if (variant & 2)
			return fail(GSR_ERR_ARG, "gsr_backward: bwd_variant bit 1 (the per-wave A/B kernel) is not in this build (make AB=1; gsr_get_option(\"ab_variants\"))", __FILE__, __LINE__);
#endif
		if (ro.fast_exp) {
			if (variant & 2) return fail(GSR_ERR_ARG, "gsr_backward: fast_exp needs the per-quarter kernel (bwd_variant bit 1 clear)", __FILE__, __LINE__);
			variant |= 8;   // bit 3: the forward used the hardware exp -- the backward takes the same decisions with it
		}
		if (fwd.skew) {
			uint32_t* tw = reinterpret_cast<uint32_t*>(const_cast<char*>(image_buffer) + il.tile_work);
			uint32_t* to = reinterpret_cast<uint32_t*>(const_cast<char*>(image_buffer) + il.tile_order);
			launch_tile_order(il.T, n_contrib, tw, to, s);
			bgv.tile_order = to;
		}
		launch_composite_bwd(il, width, height, bgv, ranges, point_list, recs, goff, final_T, n_contrib, med_pos, dL_dpix,
		                     dL_dpix_depth, dL_dpix_median_depth, dL_dpix_final_opacity, rows, row_flags,
		                     reinterpret_cast<const GsCtl*>(image_buffer + il.ctl), variant, s);
		STAGE_CHECK("composite_bwd", debug, s);
	}
	tm.mark();
	launch_preprocess_bwd(a, cam, recs, clampw, reinterpret_cast<const float*>(geom_buffer + gl.shjac), goff, rows, row_flags, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
	                      dL_dsh_rest, dL_dscale, dL_drot,
	                      GSR_PART_GEOM | ((parts & GSR_BWD_PART_SH) ? GSR_PART_SH : 0) | (sh_colors ? GSR_PART_SH_COLORS : 0) | colors_early,
	                      sh_g0, sh_g1, s);
	STAGE_CHECK("preprocess_bwd", debug, s);
	tm.mark();
	return GSR_OK;
}

int gsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                 const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_median_depth,
                 const float* dL_dpix_final_opacity, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 char* scratch, int debug, void* stream)
{
	(void)viewmatrix; (void)projmatrix; (void)campos;   // the device copies made by gsr_forward are used
	return backward_impl(nullptr, GSR_BWD_PART_MAIN | GSR_BWD_PART_SH, 0, P, P, D, M, R, background, width, height, means3D, shs, nullptr, colors_precomp, scales,
	                     scale_modifier, rotations, cov3D_precomp, 0, tan_fovx, tan_fovy, radii, geom_buffer,
	                     binning_buffer, image_buffer, dL_dpix, dL_dpix_depth, dL_dpix_median_depth,
	                     dL_dpix_final_opacity, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
	                     nullptr, dL_dscale, dL_drot, scratch, debug, stream);
}

int gsr_backward_parts(int parts, int sh_g0, int sh_g1, int P, int D, int M, int R, const float* background, int width,
                       int height, const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp, float tan_fovx,
                       float tan_fovy, const int* radii, const char* geom_buffer, const char* binning_buffer,
                       const char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth,
                       const float* dL_dpix_median_depth, const float* dL_dpix_final_opacity, float* dL_dmean2D,
                       float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                       float* dL_dscale, float* dL_drot, char* scratch, int debug, void* stream)
{
	if (!(parts & (GSR_BWD_PART_MAIN | GSR_BWD_PART_SH)))
		return fail(GSR_ERR_ARG, "gsr_backward_parts: nothing to do", __FILE__, __LINE__);
	if ((parts & GSR_BWD_PART_SH) && sh_g0 % 256 != 0)
		return fail(GSR_ERR_ARG, "gsr_backward_parts: sh_g0 must be a multiple of 256", __FILE__, __LINE__);
	return backward_impl(nullptr, parts, sh_g0, sh_g1, P, D, M, R, background, width, height, means3D, shs, nullptr, colors_precomp,
	                     scales, scale_modifier, rotations, cov3D_precomp, 0, tan_fovx, tan_fovy, radii, geom_buffer,
	                     binning_buffer, image_buffer, dL_dpix, dL_dpix_depth, dL_dpix_median_depth,
	                     dL_dpix_final_opacity, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
	                     nullptr, dL_dscale, dL_drot, scratch, debug, stream);
}

int gsr_backward_ex(const gsr_options* opt, int parts, int sh_g0, int sh_g1, int P, int D, int M, int R,
                    const float* background, int width, int height, const float* means3D, const float* shs,
                    const float* shs_rest, const float* colors_precomp, const float* scales, float scale_modifier,
                    const float* rotations, const float* cov3D_precomp, int activation_flags, float tan_fovx,
                    float tan_fovy, const int* radii, const char* geom_buffer, const char* binning_buffer,
                    const char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth,
                    const float* dL_dpix_median_depth, const float* dL_dpix_final_opacity, float* dL_dmean2D,
                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                    float* dL_dsh_rest, float* dL_dscale, float* dL_drot, char* scratch, int debug, void* stream)
{
	if (!(parts & (GSR_BWD_PART_MAIN | GSR_BWD_PART_SH)))
		return fail(GSR_ERR_ARG, "gsr_backward_ex: nothing to do", __FILE__, __LINE__);
	if ((parts & GSR_BWD_PART_SH) && !(parts & (GSR_BWD_PART_BAND_FIRST | GSR_BWD_PART_BAND_SECOND)) && sh_g0 % 256 != 0)
		return fail(GSR_ERR_ARG, "gsr_backward_ex: sh_g0 must be a multiple of 256", __FILE__, __LINE__);
	return backward_impl(opt, parts, sh_g0, sh_g1, P, D, M, R, background, width, height, means3D, shs, shs_rest,
	                     colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, activation_flags, tan_fovx, tan_fovy,
	                     radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dpix_depth, dL_dpix_median_depth,
	                     dL_dpix_final_opacity, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dsh_rest,
	                     dL_dscale, dL_drot, scratch, debug, stream);
}

int gsr_band_classes(int P, const int* radii, const char* geom_buffer, int split_tile_row, int* first, int* second, void* stream)
{
	g_err.clear();
	if (P <= 0) return GSR_OK;
	if (!radii || !geom_buffer) return fail(GSR_ERR_ARG, "gsr_band_classes: NULL argument", __FILE__, __LINE__);
	const GeomLayout gl((size_t)P);
	launch_band_classes(P, radii, reinterpret_cast<const GsRec*>(geom_buffer + gl.recs), split_tile_row, first, second, (hipStream_t)stream);
	STAGE_CHECK("band_classes", 0, (hipStream_t)stream);
	return GSR_OK;
}

int gsr_sh_grad_from_colors(int P, int D, int M, int N, const float* means3D, const float* campos, const float* colors,
                            float* dL_dsh, void* stream)
{
	g_err.clear();
	if (P <= 0 || M <= 0) return GSR_OK;
	if (N < 0 || D < 0 || D > 3 || (D + 1) * (D + 1) > M)
		return fail(GSR_ERR_ARG, "gsr_sh_grad_from_colors: bad N / SH degree", __FILE__, __LINE__);
	if (!means3D || !dL_dsh || (N > 0 && (!campos || !colors)))
		return fail(GSR_ERR_ARG, "gsr_sh_grad_from_colors: NULL argument", __FILE__, __LINE__);
	if (!is_device_ptr(campos)) return fail(GSR_ERR_ARG, "gsr_sh_grad_from_colors: campos must be device memory", __FILE__, __LINE__);
	launch_sh_grad_from_colors(P, D, M, N, means3D, campos, colors, nullptr, nullptr, 0u, dL_dsh, (hipStream_t)stream);
	STAGE_CHECK("sh_grad_from_colors", 0, (hipStream_t)stream);
	return GSR_OK;
}

int gsr_sh_grad_from_packed(int P, int D, int M, int N, const float* means3D, const float* campos, const uint32_t* msgs,
                            const unsigned long long* msg_offsets, float* dL_dsh, void* stream)
{
	g_err.clear();
	if (P <= 0 || M <= 0) return GSR_OK;
	if (N < 0 || D < 0 || D > 3 || (D + 1) * (D + 1) > M)
		return fail(GSR_ERR_ARG, "gsr_sh_grad_from_packed: bad N / SH degree", __FILE__, __LINE__);
	if (!means3D || !dL_dsh || (N > 0 && (!campos || !msgs || !msg_offsets)))
		return fail(GSR_ERR_ARG, "gsr_sh_grad_from_packed: NULL argument", __FILE__, __LINE__);
	if (!is_device_ptr(campos)) return fail(GSR_ERR_ARG, "gsr_sh_grad_from_packed: campos must be device memory", __FILE__, __LINE__);
	launch_sh_grad_from_colors(P, D, M, N, means3D, campos, nullptr, msgs, msg_offsets, (uint32_t)gsr_msg_header_words(P), dL_dsh,
	                           (hipStream_t)stream);
	STAGE_CHECK("sh_grad_from_packed", 0, (hipStream_t)stream);
	return GSR_OK;
}

int gsr_set_option(const char* name, int value)
{
	g_err.clear();
	if (!name) return fail(GSR_ERR_ARG, "gsr_set_option: NULL name", __FILE__, __LINE__);
	const std::string n(name);
	if (n == "tight_binning") g_opt_tight.store(value);
	else if (n == "cull") g_opt_cull.store(value);
	else if (n == "fwd_variant") g_opt_fwd_variant.store(value);
	else if (n == "bwd_variant") g_opt_bwd_variant.store(value);
	else if (n == "speculative") g_opt_speculative.store(value);
	else if (n == "fast_exp") g_opt_fast_exp.store(value != 0);
	else if (n == "tile_order") g_opt_tile_order.store(value != 0);
	else if (n == "bininfo") g_opt_bininfo.store(value != 0);
	else if (n == "roctx") g_opt_roctx.store(value != 0);
	else if (n == "tile_row_lo") g_opt_band_lo.store(value > 0 ? value : 0);
	else if (n == "tile_row_hi") g_opt_band_hi.store(value);
	else if (n == "forget_forwards") {   // drop the host-side map of forward modes (tests: the next backward reads the forward's control word)
		std::lock_guard<std::mutex> lock(g_fwd_modes_mutex);
		g_fwd_modes.clear();
	}
	else if (n == "bin_capacity") {   // capacity assumed for the NEXT forward on the current device (tests: force the re-launch path)
		DevState& ds = dev_state();
		ds.cap.store(value > 0 ? (uint32_t)value : 0u);
		if (value <= 0) ds.long_lists.store(0);
	} else return fail(GSR_ERR_ARG, "gsr_set_option: unknown option", __FILE__, __LINE__);
	return GSR_OK;
}

int gsr_get_option(const char* name)
{
	if (!name) return -1;
	const std::string n(name);
	if (n == "tight_binning") return g_opt_tight.load();
	if (n == "cull") return g_opt_cull.load();
	if (n == "fwd_variant") return g_opt_fwd_variant.load();
	if (n == "bwd_variant") return g_opt_bwd_variant.load();
	if (n == "speculative") return g_opt_speculative.load();
	if (n == "fast_exp") return g_opt_fast_exp.load();
	if (n == "tile_order") return g_opt_tile_order.load();
	if (n == "bininfo") return g_opt_bininfo.load();
	if (n == "roctx") return g_opt_roctx.load() != 0 && roctx().push != nullptr;
	if (n == "tile_row_lo") return g_opt_band_lo.load();
	if (n == "tile_row_hi") return g_opt_band_hi.load();
	if (n == "bin_capacity") return (int)dev_state().cap.load();
	if (n == "ab_variants") {   // read-only: were the superseded per-wave compositing kernels (fwd_variant 1, bwd_variant bit 1) compiled in?
#ifdef GSR_AB_VARIANTS
		return 1;
#else
		return 0;
#endif
	}
	return -1;
}

int gsr_selftest(void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	float* in = nullptr;
	HIP_TRY(hipMalloc((void**)&in, 64));
	const float h_in[2] = {1.0f, 0.0f};
	uint32_t h_out[2] = {0u, 0u};
	hipError_t e = hipMemcpyAsync(in, h_in, sizeof(h_in), hipMemcpyHostToDevice, s);
	if (e == hipSuccess) {
		launch_bwd_selftest(in, reinterpret_cast<uint32_t*>(in + 4), s);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(h_out, in + 4, sizeof(h_out), hipMemcpyDeviceToHost, s);
	if (e == hipSuccess) e = hipStreamSynchronize(s);
	(void)hipFree(in);
	if (e != hipSuccess) return fail(GSR_ERR_HIP, "gsr_selftest", __FILE__, __LINE__, e);
	return (int)h_out[0];
}

int gsr_inspect_counts(const char* image_buffer, int width, int height, uint32_t out[4], void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (!image_buffer || !out) return fail(GSR_ERR_ARG, "gsr_inspect_counts: NULL argument", __FILE__, __LINE__);
	const ImgLayout il(width, height);
	GsCtl c;
	HIP_TRY(hipMemcpyAsync(&c, image_buffer + il.ctl, sizeof(GsCtl), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	out[0] = c.num_binned; out[1] = c.max_tile_count; out[2] = c.ref_rendered; out[3] = c.err_overflow;
	return GSR_OK;
}

int gsr_inspect_staged(const char* image_buffer, int width, int height, unsigned long long* staged_total, void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (!image_buffer || !staged_total) return fail(GSR_ERR_ARG, "gsr_inspect_staged: NULL argument", __FILE__, __LINE__);
	const ImgLayout il(width, height);
	std::vector<uint32_t> per_tile((size_t)il.T);
	HIP_TRY(hipMemcpyAsync(per_tile.data(), image_buffer + il.tile_count, sizeof(uint32_t) * (size_t)il.T, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	unsigned long long sum = 0;
	for (uint32_t v : per_tile) sum += v;
	*staged_total = sum;
	return GSR_OK;
}

int gsr_backward_raw(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                     const float* f_dc, const float* f_rest, const float* raw_scales, float scale_modifier,
                     const float* raw_rotations, int activation_flags, float tan_fovx, float tan_fovy,
                     const int* radii, const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                     const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_median_depth,
                     const float* dL_dpix_final_opacity, float* dL_dmean2D, float* dL_draw_opacity, float* dL_dcolor,
                     float* dL_dmean3D, float* dL_dcov3D, float* dL_df_dc, float* dL_df_rest, float* dL_draw_scale,
                     float* dL_draw_rot, char* scratch, int debug, void* stream)
{
	if (P > 0 && (!f_dc || (M > 1 && (!f_rest || !dL_df_rest)) || !dL_df_dc))
		return fail(GSR_ERR_ARG, "gsr_backward_raw: f_dc / f_rest and their gradient outputs are required", __FILE__, __LINE__);
	return backward_impl(nullptr, GSR_BWD_PART_MAIN | GSR_BWD_PART_SH, 0, P, P, D, M, R, background, width, height, means3D, f_dc, M > 1 ? f_rest : nullptr, nullptr, raw_scales,
	                     scale_modifier, raw_rotations, nullptr, activation_flags, tan_fovx, tan_fovy, radii, geom_buffer,
	                     binning_buffer, image_buffer, dL_dpix, dL_dpix_depth, dL_dpix_median_depth,
	                     dL_dpix_final_opacity, dL_dmean2D, dL_draw_opacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_df_dc,
	                     M > 1 ? dL_df_rest : nullptr, dL_draw_scale, dL_draw_rot, scratch, debug, stream);
}

int gsr_inspect_geometry(const char* geom_buffer, int P, const int* radii, float* means2D, float* depths,
                         float* conic_opacity, float* rgb, unsigned char* clamped, uint32_t* tiles_touched,
                         void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (P <= 0) return GSR_OK;
	if (!geom_buffer || !radii) return fail(GSR_ERR_ARG, "gsr_inspect_geometry: NULL buffer", __FILE__, __LINE__);
	const GeomLayout gl((size_t)P);
	hipLaunchKernelGGL(inspect_geometry_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii,
	                   reinterpret_cast<const GsRec*>(geom_buffer + gl.recs),
	                   reinterpret_cast<const uint32_t*>(geom_buffer + gl.tiles_touched), means2D, depths, conic_opacity, rgb,
	                   clamped, tiles_touched);
	STAGE_CHECK("inspect_geometry", 0, s);
	return GSR_OK;
}

int gsr_inspect_backward_sums(const char* geom_buffer, const char* scratch, int P, int R, const int* radii,
                               float* sums, void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (P <= 0) return GSR_OK;
	if (!geom_buffer || !scratch || !radii || !sums)
		return fail(GSR_ERR_ARG, "gsr_inspect_backward_sums: NULL argument", __FILE__, __LINE__);
	const GeomLayout gl((size_t)P);
	const BwdLayout wl((size_t)P, (size_t)(R > 0 ? R : 0));
	uint32_t flagged = 0;   // regime the backward ran in (see backward_impl)
	HIP_TRY(hipMemcpyAsync(&flagged, scratch + wl.bg + 8 * sizeof(float), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	launch_inspect_sums(P, radii, reinterpret_cast<const GsRec*>(geom_buffer + gl.recs),
	                    reinterpret_cast<const uint32_t*>(geom_buffer + gl.goff),
	                    reinterpret_cast<const float*>(scratch + wl.rows), flagged ? reinterpret_cast<const uint8_t*>(scratch + wl.flags) : nullptr,
	                    sums, s);
	STAGE_CHECK("inspect_backward_sums", 0, s);
	return GSR_OK;
}

int gsr_inspect_binning(const char* binning_buffer, const char* image_buffer, int R, int width, int height,
                        uint32_t* point_list, uint32_t* ranges, void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (!image_buffer) return fail(GSR_ERR_ARG, "gsr_inspect_binning: NULL buffer", __FILE__, __LINE__);
	const ImgLayout il(width, height);
	const BinLayout bl((size_t)(R > 0 ? R : 0));
	if (ranges) HIP_TRY(hipMemcpyAsync(ranges, image_buffer + il.ranges, sizeof(uint2) * (size_t)il.T, hipMemcpyDeviceToDevice, s));
	if (point_list && R > 0) {
		if (!binning_buffer) return fail(GSR_ERR_ARG, "gsr_inspect_binning: NULL binning buffer", __FILE__, __LINE__);
		HIP_TRY(hipMemcpyAsync(point_list, binning_buffer + bl.point_list, sizeof(uint32_t) * (size_t)R, hipMemcpyDeviceToDevice, s));
	}
	return GSR_OK;
}

int gsr_inspect_image(const char* image_buffer, int width, int height, float* final_T, uint32_t* n_contrib,
                      void* stream)
{
	hipStream_t s = (hipStream_t)stream;
	g_err.clear();
	if (!image_buffer) return fail(GSR_ERR_ARG, "gsr_inspect_image: NULL buffer", __FILE__, __LINE__);
	const ImgLayout il(width, height);
	hipLaunchKernelGGL(inspect_image_kernel, dim3(il.T), dim3(256), 0, s, il.gx, width, height,
	                   reinterpret_cast<const float*>(image_buffer + il.final_T),
	                   reinterpret_cast<const uint32_t*>(image_buffer + il.n_contrib), final_T, n_contrib);
	STAGE_CHECK("inspect_image", 0, s);
	return GSR_OK;
}

}  // extern "C"
