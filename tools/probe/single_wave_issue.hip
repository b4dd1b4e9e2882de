// How fast can ONE wavefront per SIMD issue VALU instructions on gfx950, as a function of the instruction-level
// parallelism it offers?  Each wavefront runs ITER x 16 instructions of one kind over REP independent accumulators
// (REP = 1: every instruction depends on the previous one; REP = 16: sixteen independent chains).  W wavefronts per SIMD.
//   build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/swi tools/probe/single_wave_issue.hip && /tmp/swi
// Why: launches of <= 1 wavefront per SIMD (1 024 long reads) run at about half the per-SIMD rate of full launches
// (profiles/r3_small_launch_ab.txt); this separates "a lone wavefront cannot issue faster" from "its chains are too serial".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 8192

template<int KIND, int REP>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t seed)
{
	uint32_t a[REP];
#pragma unroll
	for (int r = 0; r < REP; ++r) a[r] = (seed + threadIdx.x * 7u + r) & 0x3fff3fffu;
	uint32_t b = (seed * 3u + 1u) & 0x3fff3fffu, c = (seed ^ 0x55u) & 0x3fff3fffu;
	for (int i = 0; i < ITER; ++i) {
#pragma unroll
		for (int x = 0; x < 16; ++x) {
			const int r = x % REP;
			if (KIND == 0) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 1) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 3) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 4) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 5) {     /* the cell update's chain shape: max3 -> sub -> max, alternating kinds on one chain */
				if (x % 3 == 0) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
				else if (x % 3 == 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
				else asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			}
		}
	}
	uint32_t s = 0;
#pragma unroll
	for (int r = 0; r < REP; ++r) s ^= a[r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
template<int KIND> static void run(const char *name, int cus)
{
	kern_t k[4] = { probe<KIND, 1>, probe<KIND, 2>, probe<KIND, 4>, probe<KIND, 16> };
	const int reps[4] = { 1, 2, 4, 16 };
	for (int wps = 1; wps <= 4; wps *= 2)
		for (int v = 0; v < 4; ++v) {
			const int blocks = cus * wps;
			uint32_t *out;
			hipMalloc(&out, (size_t)blocks * 256 * 4);
			hipEvent_t e0, e1;
			hipEventCreate(&e0); hipEventCreate(&e1);
			hipLaunchKernelGGL(k[v], dim3(blocks), dim3(256), 0, 0, out, 1u);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(k[v], dim3(blocks), dim3(256), 0, 0, out, 2u);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			const double inst = (double)ITER * 16;
			printf("%-20s waves/SIMD %d  chains %2d  %8.4f ms  %6.2f cycles per instruction of one wavefront, %5.2f per SIMD\n", name, wps, reps[v], ms,
			       ms * 1e-3 * 2.4e9 / inst, ms * 1e-3 * 2.4e9 / inst / wps);
			hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
		}
}

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	printf("device %s, %d CUs (cycles at the nominal 2.4 GHz)\n", prop.gcnArchName, cus);
	run<0>("v_pk_max_i16", cus);
	run<1>("v_pk_maximum3_f16", cus);
	run<2>("v_add_u32", cus);
	run<3>("v_bfi_b32", cus);
	run<4>("v_pk_sub_i16", cus);
	run<5>("max3/sub/max chain", cus);
	return 0;
}
