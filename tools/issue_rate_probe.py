#!/usr/bin/env python
"""Instruction issue-rate probe for gfx950 (MI355X): how many cycles does a SIMD need per wave64 instruction?

Generates one kernel per pattern (an unrolled block of inline-asm instructions, 8 waves per SIMD unless --waves says
otherwise, every CU busy), times it with HIP events and prints ns x 2.4 per instruction per SIMD (= cycles at 2.4 GHz;
the sustained clock is a little lower, so "2.4" reads as 2 issue cycles and "4.2" as 4).  Used to build the VALU
cost model in docs/DESIGN_history_r1-r4.md s4.4: plain FP32 / logic ops with VGPR, inline-constant or literal sources issue at ~2 cycles,
everything else (compares, selects, min/max, shifts, conversions, DPP, packed FP32, ANY op with an SGPR source) at ~4,
transcendentals and lane swaps at ~8; scalar ALU instructions cost ~4 cycles of a SIMD's issue turns and overlap VALU.

    python tools/issue_rate_probe.py [--waves 8] [--keep]      # needs hipcc + an MI355X
"""
import argparse
import os
import subprocess
import sys
import tempfile

A = [f"%[a{i}]" for i in range(8)]
S = "%[s]"
SALU = "s_and_b64 s[24:25], s[24:25], s[22:23]"


def fma(i):
    return f"v_fma_f32 {A[i]}, {A[i]}, {S}, {A[i]}"


def each(fmt):
    return [fmt.format(a=A[i], b=A[(i + 1) % 8], s=S, p=f"%[p{i % 4}]", i=i) for i in range(8)]


def alt(x, y):
    return [(x if i % 2 == 0 else y).format(a=A[i], b=A[(i + 1) % 8], s=S, p=f"%[p{i % 4}]", i=i) for i in range(8)]


PATTERNS = {
    # ---- fast class
    "v_fma_f32 independent": each("v_fma_f32 {a}, {a}, {s}, {a}"),
    "v_fma_f32 dependent chain": [fma(0)] * 8,
    "v_fmaak_f32 (literal) dependent chain": [f"v_fmaak_f32 {A[0]}, {A[0]}, {S}, 0x3f317214"] * 8,
    "v_mul_f32 inline constant": each("v_mul_f32_e32 {a}, 0.5, {a}"),
    "v_add_f32": each("v_add_f32_e32 {a}, {a}, {s}"),
    "v_and_b32 / v_xor_b32 / v_or_b32": alt("v_and_b32_e32 {a}, {a}, {s}", "v_xor_b32_e32 {a}, {a}, {s}"),
    "v_add_u32": each("v_add_u32_e32 {a}, {a}, {s}"),
    "v_mov_b32 vgpr": each("v_mov_b32_e32 {a}, {s}"),
    # ---- the same ops with an SGPR source
    "v_fma_f32 SGPR source": each("v_fma_f32 {a}, {a}, s22, {a}"),
    "v_mul_f32 SGPR source": each("v_mul_f32_e32 {a}, s22, {a}"),
    "v_mov_b32 from SGPR": each("v_mov_b32_e32 {a}, s22"),
    # ---- slow class
    "v_cmp_gt_f32 -> SGPR pair": each("v_cmp_gt_f32_e64 s[26:27], {a}, {s}"),
    "v_cmp_gt_f32 -> vcc": each("v_cmp_gt_f32_e32 vcc, {a}, {s}"),
    "v_cndmask_b32 (SGPR-pair mask)": each("v_cndmask_b32_e64 {a}, {a}, {s}, s[20:21]"),
    "v_min_f32 / v_max_f32": alt("v_min_f32_e32 {a}, {a}, {s}", "v_max_f32_e32 {a}, {a}, {s}"),
    "v_med3_f32": each("v_med3_f32 {a}, {a}, {s}, {s}"),
    "v_lshl_add_u32": each("v_lshl_add_u32 {a}, {a}, 1, {s}"),
    "v_lshlrev_b32": each("v_lshlrev_b32_e32 {a}, 1, {a}"),
    "v_bfi_b32": each("v_bfi_b32 {a}, {s}, {a}, {a}"),
    "v_cvt_f32_u32": each("v_cvt_f32_u32_e32 {a}, {a}"),
    "v_ldexp_f32": each("v_ldexp_f32 {a}, {a}, {s}"),
    "v_readfirstlane_b32": each("v_readfirstlane_b32 s26, {a}"),
    "v_pk_fma_f32 (per instruction = 2 values)": each("v_pk_fma_f32 {p}, {p}, %[ps], {p}"),
    "v_pk_mul_f32 / v_pk_add_f32": alt("v_pk_mul_f32 {p}, {p}, %[ps]", "v_pk_add_f32 {p}, {p}, %[ps]"),
    "v_pk_fma_f32 SGPR-pair source": each("v_pk_fma_f32 {p}, {p}, %[ps], s[22:23] op_sel_hi:[1,1,0]"),
    "v_add_f32 DPP row_shr:1": each("v_add_f32_dpp {a}, {a}, {a} row_shr:1 row_mask:0xf bank_mask:0xf"),
    "v_mov_b32 DPP quad_perm": each("v_mov_b32_dpp {a}, {b} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    # ---- 8-cycle class
    "v_exp_f32": each("v_exp_f32_e32 {a}, {a}"),
    "v_rcp_f32": each("v_rcp_f32_e32 {a}, {a}"),
    "v_permlane32_swap_b32": each("v_permlane32_swap_b32_e32 {a}, {b}"),
    "v_permlane16_swap_b32": each("v_permlane16_swap_b32_e32 {a}, {b}"),
    # ---- mixes
    "fma / cndmask alternating": alt("v_fma_f32 {a}, {a}, {s}, {a}", "v_cndmask_b32_e64 {a}, {a}, {s}, s[20:21]"),
    "fma / v_cmp alternating": alt("v_fma_f32 {a}, {a}, {s}, {a}", "v_cmp_gt_f32_e64 s[26:27], {a}, {s}"),
    "3 fma + 1 cndmask": [fma(i) if i % 4 != 3 else f"v_cndmask_b32_e64 {A[i]}, {A[i]}, {S}, s[20:21]" for i in range(8)],
    # ---- scalar unit (per SALU instruction) and its overlap with VALU (per VALU instruction)
    "s_and_b64 (per SALU)": [SALU] * 8,
    "s_ff1_i32_b64 + s_bitset0_b64 (per SALU)": ["s_ff1_i32_b64 s26, s[22:23]", "s_bitset0_b64 s[24:25], s26"] * 4,
    "s_nop 0 (per instruction)": ["s_nop 0"] * 8,
    "1 fma + 1 SALU alternating (per VALU)": sum([[fma(i), SALU] for i in range(8)], []),
    "2 fma + 1 SALU (per VALU)": sum([[fma(2 * i), fma(2 * i + 1), SALU] for i in range(4)], []),
    "fma + cndmask + 2 SALU (per VALU)": sum([[fma(2 * i), SALU, f"v_cndmask_b32_e64 {A[2 * i + 1]}, {A[2 * i + 1]}, {S}, s[20:21]", SALU]
                                             for i in range(4)], []),
    # ---- LDS with a wave-uniform (broadcast) address, all four SIMDs of the CU reading (per LDS instruction)
    "ds_read_b32 broadcast (per LDS)": ["ds_read_b32 %[w], %[z]"] * 8 + ["s_waitcnt lgkmcnt(0)"],
    "ds_read_b64 broadcast (per LDS)": ["ds_read_b64 %[p], %[z]"] * 8 + ["s_waitcnt lgkmcnt(0)"],
    "ds_read_b128 broadcast (per LDS)": ["ds_read_b128 %[q], %[z]"] * 8 + ["s_waitcnt lgkmcnt(0)"],
    "ds_swizzle_b32 (per LDS)": [f"ds_swizzle_b32 %[w], {A[i]} offset:swizzle(SWAP,16)" for i in range(8)] + ["s_waitcnt lgkmcnt(0)"],
    "ds_bpermute_b32 (per LDS)": [f"ds_bpermute_b32 %[w], %[z], {A[i]}" for i in range(8)] + ["s_waitcnt lgkmcnt(0)"],
}

KERNEL = '''__global__ void k{k}(float* out, int iters, float s)
{{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	v2f p0 = {{a0, a1}}, p1 = {{a2, a3}}, p2 = {{a4, a5}}, p3 = {{a6, a7}}, ps = {{s, s}};
	__shared__ v4f lds[64];
	lds[threadIdx.x & 63] = v4f{{a0, a1, a2, a3}};
	__syncthreads();
	v4f q = {{0, 0, 0, 0}}; float wv = 0; v2f pq = {{0, 0}};
	int z = 16 * (int)(s);
	asm volatile("s_mov_b64 vcc, exec\\n s_mov_b64 s[20:21], exec\\n s_mov_b64 s[22:23], exec" ::: "vcc", "s20", "s21", "s22", "s23");
	for (int i = 0; i < iters; i++) {{
#pragma unroll
		for (int r = 0; r < 8; r++)
			asm volatile("{body}"
			             : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),
			               [a7] "+v"(a7), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [q] "+v"(q), [w] "+v"(wv), [p] "+v"(pq)
			             : [s] "v"(s), [ps] "v"(ps), [z] "v"(z)
			             : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc", "scc");
	}}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + q.x + wv + pq.x;
}}
'''

MAIN = '''
typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t kf, int n, float* out, int wps)
{
	const int iters = 1000;
	dim3 grid(256 * wps), block(256);   // 256 CUs x wps workgroups of 4 waves: wps waves on every SIMD
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kf, grid, block, 0, 0, out, iters, 1.0f);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(kf, grid, block, 0, 0, out, iters, 1.0f);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	printf("%-46s %7.3f ms  %6.2f\\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * n * wps));
	fflush(stdout);
}
int main(int argc, char** argv)
{
	const int wps = argc > 1 ? atoi(argv[1]) : 8;
	float* out; (void)hipMalloc(&out, 256 * 16 * 256 * 4);
	printf("# waves per SIMD = %d; last column = ns x 2.4 per counted instruction per SIMD\\n", wps);
'''


def generate():
    src = ["#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdlib>",
           "typedef float v2f __attribute__((ext_vector_type(2)));", "typedef float v4f __attribute__((ext_vector_type(4)));"]
    calls = []
    for k, (name, lines) in enumerate(PATTERNS.items()):
        per_salu = "(per SALU)" in name or "(per instruction)" in name
        per_lds = "(per LDS)" in name
        if per_lds:
            n = sum(1 for l in lines if l.startswith("ds_"))
        elif per_salu:
            n = len(lines)
        else:
            n = sum(1 for l in lines if l.startswith("v_"))
        src.append(KERNEL.format(k=k, body="\\n ".join(lines)))
        calls.append(f'	run("{name}", k{k}, {n * 8}, out, wps);')
    return "\n".join(src) + MAIN + "\n".join(calls) + "\n	return 0;\n}\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", type=int, default=8, help="waves per SIMD (1..8)")
    ap.add_argument("--keep", action="store_true", help="keep the generated source next to this script")
    a = ap.parse_args()
    d = os.path.dirname(os.path.abspath(__file__)) if a.keep else tempfile.mkdtemp(prefix="issue_probe_")
    hip, exe = os.path.join(d, "issue_rate_probe.hip"), os.path.join(d, "issue_rate_probe")
    with open(hip, "w") as f:
        f.write(generate())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", "-o", exe, hip])
    sys.exit(subprocess.call(["timeout", "120", exe, str(a.waves)]))


if __name__ == "__main__":
    main()
