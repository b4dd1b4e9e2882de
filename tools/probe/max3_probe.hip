// gfx950 probe: can v_pk_maximum3_f16 / v_pk_minimum3_f16 serve as a packed 16-bit INTEGER max3 / min3?
// For bit patterns of positive finite halves (0x0000..0x7BFF) the IEEE order equals the unsigned integer order, so if the
// instruction returns the selected operand bit for bit (no denormal flush, no canonicalisation) one instruction replaces two
// v_pk_max_u16.  Part 1 checks that on random patterns per range; part 2 measures issue rates next to v_pk_max_u16.
//   build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/max3_probe tools/probe/max3_probe.hip && /tmp/max3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 16
#define ITER 16384

__global__ void sem(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *omax3, uint32_t *omin3, uint32_t *omax2, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t x = a[i], y = b[i], z = c[i], r;
	asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
	omax3[i] = r;
	asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
	omin3[i] = r;
	asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
	omax2[i] = r;
}

template<int KIND>
__global__ void __launch_bounds__(256) rate(uint32_t *out, uint32_t seed)
{
	uint32_t a[REP];
#pragma unroll
	for (int r = 0; r < REP; ++r) a[r] = 0x10001000u + ((seed + threadIdx.x * 7u + r) & 0x0fff0fffu);
	uint32_t b = 0x20002000u + (seed & 0xff), c = 0x18001800u ^ (seed & 0xf0);
	for (int i = 0; i < ITER; ++i) {
#pragma unroll
		for (int r = 0; r < REP; ++r) {
			if (KIND == 0) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 1) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 2) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 3) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 4) asm volatile("v_max3_u16 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 5) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[r]) : "v"(b));
			if (KIND == 6) asm volatile("v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[r]) : "v"(b));
			if (KIND == 7) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2\n\tv_sub_u32 %0, %0, %3" : "+v"(a[r]) : "v"(b), "v"(c), "v"(seed));   /* max3 + a 2-cycle op */
			if (KIND == 8) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
			if (KIND == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
			if (KIND == 10) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(a[r]) : "v"(b));
			if (KIND == 11) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[r]) : "v"(b));
		}
	}
	uint32_t s = 0;
#pragma unroll
	for (int r = 0; r < REP; ++r) s ^= a[r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }
static uint32_t umax(uint32_t x, uint32_t y) { return x > y ? x : y; }
static uint32_t umin(uint32_t x, uint32_t y) { return x < y ? x : y; }

int main()
{
	const int n = 1 << 20;
	struct { const char *name; uint32_t lo, hi; } ranges[] = {
		{ "normal positive   [0x0400,0x7BFF]", 0x0400, 0x7BFF }, { "with denormals    [0x0000,0x7BFF]", 0x0000, 0x7BFF },
		{ "denormals only    [0x0000,0x03FF]", 0x0000, 0x03FF }, { "with inf / NaN    [0x0400,0x7FFF]", 0x0400, 0x7FFF },
		{ "negative normals  [0x8400,0xFBFF]", 0x8400, 0xFBFF } };
	uint32_t *da, *db, *dc, *o3, *om, *o2;
	hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&o3, n * 4); hipMalloc(&om, n * 4); hipMalloc(&o2, n * 4);
	std::vector<uint32_t> a(n), b(n), c(n), r3(n), rm(n), r2(n);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
	printf("part 1: bit patterns as unsigned 16-bit integers, per half: mismatches against integer max3 / min3 / max2 over %d random triples\n", n);
	for (auto &R : ranges) {
		const uint32_t span = R.hi - R.lo + 1;
		for (int i = 0; i < n; ++i) {
			a[i] = (R.lo + rnd() % span) | ((R.lo + rnd() % span) << 16);
			b[i] = (R.lo + rnd() % span) | ((R.lo + rnd() % span) << 16);
			c[i] = (R.lo + rnd() % span) | ((R.lo + rnd() % span) << 16);
			if (i % 7 == 0) b[i] = a[i];                           /* ties */
			if (i % 11 == 0) c[i] = (a[i] & 0xffffu) | (b[i] & 0xffff0000u);
		}
		hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(sem, dim3(n / 256), dim3(256), 0, 0, da, db, dc, o3, om, o2, n);
		hipMemcpy(r3.data(), o3, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rm.data(), om, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, n * 4, hipMemcpyDeviceToHost);
		long bad3 = 0, badm = 0, bad2 = 0;
		int shown = 0;
		for (int i = 0; i < n; ++i)
			for (int h = 0; h < 2; ++h) {
				const uint32_t x = (a[i] >> (16 * h)) & 0xffff, y = (b[i] >> (16 * h)) & 0xffff, z = (c[i] >> (16 * h)) & 0xffff;
				const uint32_t e3 = umax(umax(x, y), z), em = umin(umin(x, y), z), e2 = umax(x, y);
				const uint32_t g3 = (r3[i] >> (16 * h)) & 0xffff, gm = (rm[i] >> (16 * h)) & 0xffff, g2 = (r2[i] >> (16 * h)) & 0xffff;
				if (g3 != e3) { ++bad3; if (shown < 3) { printf("    max3(%04x, %04x, %04x) = %04x, integer %04x\n", x, y, z, g3, e3); ++shown; } }
				if (gm != em) ++badm;
				if (g2 != e2) ++bad2;
			}
		printf("  %s  v_pk_maximum3_f16: %ld  v_pk_minimum3_f16: %ld  v_pk_max_f16: %ld\n", R.name, bad3, badm, bad2);
	}
	printf("part 2: issue rate, cycles per wave64 instruction per SIMD at 2 / 4 / 8 wavefronts per SIMD (2.4 GHz nominal)\n");
	typedef void (*kern_t)(uint32_t*, uint32_t);
	const char *names[] = { "v_pk_maximum3_f16", "v_pk_minimum3_f16", "v_pk_max_u16", "v_lshl_or_b32", "v_max3_u16", "v_alignbit_b32", "v_max_u16_sdwa(hi)",
	                        "pk_maximum3+v_sub_u32 (2 instr)", "v_max_u32", "v_and_or_b32", "v_pk_add_u16 clamp", "v_pk_sub_u16 clamp" };
	kern_t kern[] = { rate<0>, rate<1>, rate<2>, rate<3>, rate<4>, rate<5>, rate<6>, rate<7>, rate<8>, rate<9>, rate<10>, rate<11> };
	const int cus = prop.multiProcessorCount;
	for (int k = 0; k < 12; ++k) {
		printf("  %-32s", names[k]);
		for (int wps = 2; wps <= 8; wps *= 2) {
			const int blocks = cus * wps;
			uint32_t *out;
			hipMalloc(&out, (size_t)blocks * 256 * 4);
			hipEvent_t e0, e1;
			hipEventCreate(&e0); hipEventCreate(&e1);
			hipLaunchKernelGGL(kern[k], dim3(blocks), dim3(256), 0, 0, out, 1u);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(kern[k], dim3(blocks), dim3(256), 0, 0, out, 2u);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			printf("  %d: %6.3f", wps, ms * 1e-3 * 2.4e9 / ((double)wps * ITER * REP));
			hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
		}
		printf("\n");
	}
	return 0;
}
