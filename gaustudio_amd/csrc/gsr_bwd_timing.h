// gsr_bwd_timing.h -- DIAGNOSTIC scaffolding of composite_bwd_quarter_kernel, compiled in only with
// `make BWD_EXTRA=-DGSR_BWD_TIMING` (tools/bwd_phase_timing.py; profiles/r03_composite_bwd_phases.txt): s_memtime around the phases
// of a wave's life, accumulated in registers and stored to the wave's own slot at the end.  A product build sees four empty
// macros.  (A first version added the sums with device atomics to twelve shared words: 420 k same-address atomics per launch
// stretched every workgroup's tail and made the memory phases look four times as long as they are.)
#pragma once
#ifdef GSR_BWD_TIMING
#define GSR_TM_SLOTS 40000
namespace gsr { __device__ unsigned long long g_bwd_phase_ticks[GSR_TM_SLOTS * 12]; }
#define TM_DECL unsigned long long tm_last = __builtin_amdgcn_s_memtime(); unsigned long long tm_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
#define TM(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm_acc[k] += now_ - tm_last; tm_last = now_; }
#define TM_END { const unsigned w_ = blockIdx.x * 4 + wv; if (lane == 0 && w_ < GSR_TM_SLOTS) { for (int k_ = 0; k_ < 12; k_++) g_bwd_phase_ticks[w_ * 12 + k_] += tm_acc[k_]; } }
// out: GSR_TM_SLOTS x 12 tick sums (slot = workgroup * 4 + wave); reset: clear them afterwards
extern "C" __attribute__((used, visibility("default"))) int gsr_debug_bwd_phase_ticks(unsigned long long* out, int reset)
{
	void* p = nullptr;
	if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&p, HIP_SYMBOL(gsr::g_bwd_phase_ticks)) != hipSuccess) return -1;
	if (out && hipMemcpy(out, p, sizeof(unsigned long long) * GSR_TM_SLOTS * 12, hipMemcpyDeviceToHost) != hipSuccess) return -2;
	if (reset && hipMemset(p, 0, sizeof(unsigned long long) * GSR_TM_SLOTS * 12) != hipSuccess) return -3;
	return 0;
}
#else
#define TM_DECL
#define TM(k)
#define TM_END
#endif
