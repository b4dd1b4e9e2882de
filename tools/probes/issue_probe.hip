// Issue rate of VALU instructions from ONE wavefront per SIMD against several: dependent chains of 1 / 2 / 4 / 8 interleaved
// independent streams (v_pk_add_u16, v_pk_max_i16, v_add_u32, v_alignbit_b32).  hipcc --offload-arch=gfx950 -O2 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template<int OP, int CH>
__global__ void __launch_bounds__(1024) probe(uint32_t *out, uint64_t *cyc, int iters)
{
	uint32_t a[8];
	for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
	const uint32_t k = 0x00010003u + blockIdx.x;
	uint64_t t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < REP / CH; ++r) {
#pragma unroll
			for (int c = 0; c < CH; ++c) {
				if (OP == 0) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[c]) : "v"(k));
				if (OP == 1) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[c]) : "v"(k));
				if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(k));
				if (OP == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[c]) : "v"(k));
			}
		}
	}
	uint64_t t1 = __builtin_readcyclecounter();
	uint32_t s = 0;
	for (int i = 0; i < 8; ++i) s ^= a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template<int OP, int CH>
static void run(const char *name, int waves_per_simd, uint32_t *out, uint64_t *cyc)
{
	const int iters = 20000, threads = 256 * waves_per_simd;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	probe<OP, CH><<<256, threads>>>(out, cyc, 100);
	hipEventRecord(e0);
	probe<OP, CH><<<256, threads>>>(out, cyc, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double inst = (double)iters * REP * waves_per_simd;        // per SIMD
	printf("%-16s chains %d  waves/SIMD %d : %6.2f ns-cycles(2.4GHz)/inst/SIMD  (per wave %6.2f)\n", name, CH, waves_per_simd, ms * 1e-3 * 2.4e9 / inst, ms * 1e-3 * 2.4e9 / ((double)iters * REP));
}
int main()
{
	uint32_t *out; uint64_t *cyc;
	hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
	for (int w = 1; w <= 4; w *= 2) {
#define ALL(OP, NAME) run<OP, 1>(NAME, w, out, cyc); run<OP, 2>(NAME, w, out, cyc); run<OP, 4>(NAME, w, out, cyc); run<OP, 8>(NAME, w, out, cyc);
		ALL(0, "v_pk_add_u16") ALL(1, "v_pk_max_i16") ALL(2, "v_add_u32") ALL(3, "v_alignbit_b32")
	}
	return 0;
}
