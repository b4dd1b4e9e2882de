// VALU issue-rate probe for gfx950: how many cycles does one wave64 VALU instruction of a given kind occupy a SIMD?
// Every wavefront runs a long unrolled chain of independent instructions of one kind; with W wavefronts per SIMD the
// SIMD's issue rate saturates, and cycles-per-instruction = shader cycles / (instructions per SIMD).
//   build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/probe/valu_rate.hip && /tmp/valu_rate
// Result on MI355X (profiles/r1d_valu_rate.txt): integer and packed-int16 ops issue one wave64 instruction per 4 cycles
// per SIMD, v_fma_f32 / v_pk_fma_f32 likewise ... see the file.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

#define REP 16            /* independent accumulators per lane */
#define ITER 16384        /* loop trips, each issuing REP instructions */

template<int KIND>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t seed, unsigned long long *cyc)
{
	uint32_t a[REP];
#pragma unroll
	for (int r = 0; r < REP; ++r) a[r] = seed + threadIdx.x * 7u + r;
	uint32_t b = seed * 3u + 1u, c = seed ^ 0x55u;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 d[REP], d2 = { (float)seed, 1.0f }, d3 = { 0.5f, (float)seed };
#pragma unroll
	for (int r = 0; r < REP; ++r) d[r] = f2{ (float)r, (float)threadIdx.x };
	unsigned long long lm = 0x5555aaaa0f0ff0f0ull ^ seed, lm2[4] = { 0, 0, 0, 0 };
	asm volatile("" : "+s"(lm));
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < ITER; ++i) {
#pragma unroll
		for (int r = 0; r < REP; ++r) {
			if (KIND == 0) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 1) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 2) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 3) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 4) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 5) asm volatile("v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "+v"(a[r]));
			if (KIND == 6) asm volatile("v_pk_lshlrev_b16 %0, 1, %0 op_sel_hi:[0,1]" : "+v"(a[r]));
			if (KIND == 7) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 8) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 9) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 10) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 13) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 14) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 15) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 16) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 18) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 19) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 20) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 21) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 22) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 23) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a[r]));
			if (KIND == 24) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 25) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[r]));
			if (KIND == 26) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[r]));
			if (KIND == 27) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 28) asm volatile("v_mov_b32 %0, %1" : "=v"(a[r]) : "v"(b));
			if (KIND == 29) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[r]) : "v"(b));
			if (KIND == 30) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[r]) : "v"(b));
			if (KIND == 31) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 33) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 34) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 35) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 36) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 37) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 38) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 39) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 40) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 41) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[r]) : "v"(d2));
			if (KIND == 42) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[r]) : "v"(d2), "v"(d3));
			if (KIND == 43) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 44) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 45) asm volatile("v_max_f16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 46) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 47) asm volatile("v_msad_u8 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 48) asm volatile("v_dot2_i32_i16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 49) asm volatile("v_dot4_i32_i8 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 50) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "s"(lm));
			if (KIND == 51) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 52) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 53) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" : : "v"(a[r]), "v"(b) : "vcc");
			if (KIND == 54) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(lm2[r & 3]) : "v"(a[r]), "v"(b));
			if (KIND == 55) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[r]) : "v"(c), "v"(b), "s"(lm));
			if (KIND == 56) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[r]) : "v"(c), "v"(b));
			if (KIND == 57) asm volatile("v_pk_max_u16 %0, %0, %2\n\ts_mov_b64 %1, %3" : "+v"(a[r]), "=s"(lm2[r & 3]) : "v"(b), "s"(lm));
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	uint32_t s = 0;
#pragma unroll
	for (int r = 0; r < REP; ++r) s ^= a[r] ^ (uint32_t)d[r].x ^ (uint32_t)d[r].y ^ (uint32_t)lm2[r & 3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(uint32_t*, uint32_t, unsigned long long*);

int main(int argc, char **argv)
{
	const int kfirst = argc > 1 ? atoi(argv[1]) : 0;
	const char *names[] = { "v_pk_max_i16", "v_pk_add_u16", "v_pk_sub_i16", "v_pk_mad_i16", "v_pk_min_u16", "v_pk_ashrrev_i16", "v_pk_lshlrev_b16", "v_max_i32", "v_min_i32", "v_max_u32", "v_max3_i32", "v_add_u32", "v_sub_u32", "v_add3_u32", "v_mad_u32_u24", "v_mad_i32_i24", "v_xor_b32", "v_and_b32", "v_or_b32", "v_or3_b32", "v_and_or_b32", "v_bfi_b32", "v_bitop3_b32", "v_bfe_i32", "v_perm_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_lshl_add_u32", "v_mov_b32", "v_cndmask_b32", "v_mov_dpp_ror", "v_fma_f32", "v_add_f32", "v_max_f32", "v_min_f32", "v_max3_f32", "v_pk_add_f16", "v_pk_max_f16", "v_pk_min_f16", "v_pk_fma_f16", "v_pk_mul_f16", "v_pk_add_f32", "v_pk_fma_f32", "v_max_i16", "v_add_u16", "v_max_f16", "v_sad_u16", "v_msad_u8", "v_dot2_i32_i16", "v_dot4_i32_i8", "v_cndmask_e64_sgpr", "v_mul_lo_u32", "v_pk_mul_lo_u16", "v_cmp_lt_i32_vcc", "v_cmp_lt_i32_sgpr", "v_cndmask_e64_nodep", "v_cndmask_vcc_nodep", "pk_max+s_mov_b64" };
	kern_t kern[] = { probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>, probe<11>, probe<12>, probe<13>, probe<14>, probe<15>, probe<16>, probe<17>, probe<18>, probe<19>, probe<20>, probe<21>, probe<22>, probe<23>, probe<24>, probe<25>, probe<26>, probe<27>, probe<28>, probe<29>, probe<30>, probe<31>, probe<32>, probe<33>, probe<34>, probe<35>, probe<36>, probe<37>, probe<38>, probe<39>, probe<40>, probe<41>, probe<42>, probe<43>, probe<44>, probe<45>, probe<46>, probe<47>, probe<48>, probe<49>, probe<50>, probe<51>, probe<52>, probe<53>, probe<54>, probe<55>, probe<56>, probe<57> };
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	int clk_khz = 0;
	hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
	printf("device %s, %d CUs, clock attribute %d MHz\n", prop.gcnArchName, cus, clk_khz / 1000);
	printf("%-18s %6s %12s %12s %10s\n", "instruction", "waves", "ms", "cyc/inst", "Tinst/s");
	for (int k = kfirst; k < 58; ++k) {
		for (int wps = 2; wps <= 8; wps *= 4) {                   /* wavefronts per SIMD */
			const int blocks = cus * wps;                          /* 256 threads = 4 waves = one per SIMD of a CU */
			uint32_t *out;
			unsigned long long *cyc;
			hipMalloc(&out, (size_t)blocks * 256 * 4);
			hipMalloc(&cyc, (size_t)blocks * 8);
			hipEvent_t e0, e1;
			hipEventCreate(&e0); hipEventCreate(&e1);
			hipLaunchKernelGGL(kern[k], dim3(blocks), dim3(256), 0, 0, out, 1u, cyc);      /* warm-up */
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(kern[k], dim3(blocks), dim3(256), 0, 0, out, 2u, cyc);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			std::vector<unsigned long long> h(blocks);
			hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
			double avg = 0;
			for (int b = 0; b < blocks; ++b) avg += (double)h[b];
			avg /= blocks;                                           /* cycle counter ticks one wavefront spent in the loop */
			const double inst_per_simd = (double)wps * ITER * REP;
			/* s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9: use wall time x nominal clock instead */
			const double cyc_wall = ms * 1e-3 * 2.4e9;
			printf("%-18s %6d %12.4f %12.3f %10.2f   (counter ticks per wave: %.0f)\n", names[k], wps, ms, cyc_wall / inst_per_simd,
			       inst_per_simd * cus * 4 * 64 / (ms * 1e-3) / 1e12, avg);
			hipFree(out); hipFree(cyc);
			hipEventDestroy(e0); hipEventDestroy(e1);
		}
	}
	return 0;
}
