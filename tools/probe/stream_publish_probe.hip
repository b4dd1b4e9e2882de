// Can a kernel that is already running consume an upload piece by piece?  One persistent launch (every wavefront slot of the
// device taken), a host that uploads a buffer in pieces on ANOTHER stream and publishes "pieces <= k have landed" in a device word
// behind each piece, wavefronts that poll that word (system-scope loads + s_sleep, bounded) and then checksum their slice.
//   build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/spp tools/probe/stream_publish_probe.hip && /tmp/spp
// Questions (round 4, the streamed plans of ksw2_host_plan.c):
//   1. does hipStreamWriteValue32 on plain hipMalloc memory work, and how long after the copy does a poller see the value?
//   2. does a one-wavefront "publish" kernel on the upload stream get a slot while the device is full of pollers
//      (a) at 2 wavefronts per SIMD of 224 VGPRs (64 registers left per SIMD), (b) at 4 x 128 VGPRs (none left)?
//   3. are the bytes a poller reads right after it saw the word the uploaded ones (fresh lines), and what if it had touched
//      the line BEFORE the copy landed (stale L2 line)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template<int VG>
__device__ __forceinline__ void burn_regs(uint32_t &acc)
{
	// keep VG registers alive across the poll loop so the kernel really occupies them
	uint32_t r[VG];
#pragma unroll
	for (int i = 0; i < VG; ++i) r[i] = acc * (i + 1);
#pragma unroll
	for (int i = 0; i < VG; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(acc));
#pragma unroll
	for (int i = 0; i < VG; ++i) acc ^= r[i];
}

// every wavefront: piece = wave % npieces; wait until *flag > piece; checksum its 4 KB slice of the piece
template<int VG, bool PRETOUCH>
__global__ void __launch_bounds__(256) poller(const uint32_t *flag_, const uint32_t *data, size_t piece_words, int npieces, uint64_t *seen_at,
                                               uint32_t *sums, uint64_t t_limit)
{
	const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int piece = wave % npieces;
	const uint32_t *src = data + (size_t)piece * piece_words + (size_t)(wave / npieces) * 1024 % piece_words;
	// flag_ == 0: no flag; watch the LAST word of the piece change from the old pattern (copies land in address order?)
	const uint32_t *flag = flag_ ? flag_ : data + (size_t)(piece + 1) * piece_words - 1;
	const bool sentinel = flag_ == 0;
	uint32_t acc = 1;
	uint32_t pre = 0;
	if (PRETOUCH) pre = src[lane];                       // the line is now in this CU's L1 and this XCD's L2 with the OLD bytes
	const uint64_t t0 = wall_clock64();
	uint32_t v = 0;
	int aborted = 0;
	for (;;) {
		v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		if (sentinel ? v != 0x11111111u : v > (uint32_t)piece) break;
		if (wall_clock64() - t0 > t_limit) { aborted = 1; break; }
		__builtin_amdgcn_s_sleep(32);
	}
	const uint64_t t1 = wall_clock64();
	if (!aborted) {
		uint32_t s = 0;
		for (int i = lane; i < 1024; i += 64) s += src[i] * (uint32_t)(i + 1);
		for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
		acc = s;
	} else acc = 0xdeadbeefu;
	if (VG > 8) burn_regs<VG>(acc), acc = aborted ? 0xdeadbeefu : acc;   // (dead value: only the register pressure matters)
	if (lane == 0) { seen_at[wave] = t1 - t0; sums[wave] = aborted ? 0xdeadbeefu : acc + pre * 0u; }
}

__global__ void publish(uint32_t *flag, uint32_t v) { if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

template<int VG, bool PRETOUCH>
static void run(const char *name, int mode, int blocks, int delay_us)
{
	const int npieces = 8;
	const size_t piece_bytes = 8u << 20, piece_words = piece_bytes / 4, total = piece_bytes * npieces;
	uint32_t *h, *d, *flag, *sums;
	uint64_t *seen;
	CK(hipHostMalloc(&h, total));
	uint32_t *hvals;
	CK(hipHostMalloc(&hvals, 256));
	CK(hipMalloc(&d, total));
	CK(hipMalloc(&flag, mode >= 16 ? mode : 256));
	uint32_t *hblk = 0;
	if (mode >= 16) { CK(hipHostMalloc(&hblk, (size_t)mode * 8)); for (int p = 0; p < 8; ++p) for (int i = 0; i < mode / 4; ++i) hblk[(size_t)p * (mode / 4) + i] = (uint32_t)(p + 1); }
	CK(hipMalloc(&sums, (size_t)blocks * 4 * 4));
	CK(hipMalloc(&seen, (size_t)blocks * 4 * 8));
	CK(hipMemset(d, 0x11, total));                       // the OLD bytes
	CK(hipMemset(flag, 0, mode >= 16 ? mode : 256));
	for (size_t i = 0; i < total / 4; ++i) h[i] = (uint32_t)(i * 2654435761u) ^ 0x5bd1e995u;
	hipStream_t sk, su;
	CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
	CK(hipDeviceSynchronize());
	int wclk = 0;
	CK(hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0));      // kHz
	const uint64_t t_limit = (uint64_t)wclk * 50;                              // 50 ms
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0, sk));
	hipLaunchKernelGGL((poller<VG, PRETOUCH>), dim3(blocks), dim3(256), 0, sk, mode == 3 ? (uint32_t*)0 : flag, d, piece_words, npieces, seen, sums, t_limit);
	CK(hipGetLastError());
	CK(hipEventRecord(e1, sk));
	usleep(2000);                                        // the pollers are all resident and spinning by now
	for (int p = 0; p < npieces; ++p) {
		CK(hipMemcpyAsync((char*)d + p * piece_bytes, (char*)h + p * piece_bytes, piece_bytes, hipMemcpyHostToDevice, su));
		if (mode == 0) CK(hipStreamWriteValue32(su, flag, (uint32_t)(p + 1), 0));
		else if (mode == 1) hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, su, flag, (uint32_t)(p + 1));
		else if (mode == 2) { hvals[p] = (uint32_t)(p + 1); CK(hipMemcpyAsync(flag, &hvals[p], 4, hipMemcpyHostToDevice, su)); }
		else if (mode >= 16) CK(hipMemcpyAsync(flag, hblk + (size_t)p * (mode / 4), mode, hipMemcpyHostToDevice, su));
		if (delay_us) { CK(hipStreamSynchronize(su)); usleep(delay_us); }
	}
	CK(hipStreamSynchronize(su));
	CK(hipStreamSynchronize(sk));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<uint32_t> hs((size_t)blocks * 4);
	std::vector<uint64_t> ht((size_t)blocks * 4);
	CK(hipMemcpy(hs.data(), sums, hs.size() * 4, hipMemcpyDeviceToHost));
	CK(hipMemcpy(ht.data(), seen, ht.size() * 8, hipMemcpyDeviceToHost));
	int bad = 0, aborted = 0;
	double first[8], last[8];
	for (int p = 0; p < npieces; ++p) { first[p] = 1e30; last[p] = 0; }
	for (size_t w = 0; w < hs.size(); ++w) {
		const int piece = (int)(w % npieces);
		const uint32_t *src = h + (size_t)piece * piece_words + (size_t)(w / npieces) * 1024 % piece_words;
		uint32_t s = 0;
		for (int i = 0; i < 1024; ++i) s += src[i] * (uint32_t)(i + 1);
		if (hs[w] == 0xdeadbeefu) ++aborted;
		else if (VG <= 8 && hs[w] != s) ++bad;
		const double us = (double)ht[w] / wclk * 1e3;
		if (us < first[piece]) first[piece] = us;
		if (us > last[piece]) last[piece] = us;
	}
	printf("%-44s blocks=%d kernel %.3f ms, aborted waves %d, wrong checksums %d%s\n", name, blocks, ms, aborted, bad, VG > 8 ? " (checksums not compared: register burner)" : "");
	printf("    piece seen after (us, first..last poller):");
	for (int p = 0; p < npieces; ++p) printf(" %.0f..%.0f", first[p], last[p]);
	printf("\n");
	hipFree(d); hipFree(flag); hipFree(sums); hipFree(seen); hipHostFree(h);
	hipStreamDestroy(sk); hipStreamDestroy(su);
}

int main()
{
	hipDeviceProp_t pr;
	CK(hipGetDeviceProperties(&pr, 0));
	const int cus = pr.multiProcessorCount;
	printf("%s, %d CUs\n", pr.gcnArchName, cus);
	int can = 0;
	hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
	printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
	run<8, false>("write-value, small kernel, 2 WG/CU", 0, cus * 2, 0);
	run<8, false>("write-value, small kernel, 7 WG/CU", 0, cus * 7, 0);
	run<8, false>("write-value, small kernel, 8 WG/CU", 0, cus * 8, 0);
	run<8, false>("publish-kernel, small kernel, 8 WG/CU", 1, cus * 8, 0);
	run<8, false>("flag by 4-byte copy, small kernel, 8 WG/CU", 2, cus * 8, 0);
	run<8, false>("data sentinel, small kernel, 8 WG/CU", 3, cus * 8, 0);
	run<8, false>("data sentinel, small kernel, 2 WG/CU", 3, cus * 2, 0);
	run<8, false>("flag by 4-byte copy, small kernel, 2 WG/CU", 2, cus * 2, 0);
	run<8, false>("flag by 1 KB copy, small kernel, 8 WG/CU", 1024, cus * 8, 0);
	run<8, false>("flag by 4 KB copy, small kernel, 8 WG/CU", 4096, cus * 8, 0);
	run<8, false>("flag by 16 KB copy, small kernel, 8 WG/CU", 16384, cus * 8, 0);
	run<8, false>("flag by 64 KB copy, small kernel, 8 WG/CU", 65536, cus * 8, 0);
	run<8, false>("flag by 256 KB copy, small kernel, 8 WG/CU", 262144, cus * 8, 0);
	run<8, false>("flag by 1 MB copy, small kernel, 8 WG/CU", 1048576, cus * 8, 0);
	run<116, false>("flag by 64 KB copy, 128 VGPR pollers, 4 WG/CU", 65536, cus * 4, 0);
	run<116, false>("write-value, 128 VGPR pollers, 4 WG/CU", 0, cus * 4, 0);
	run<116, false>("flag by 4-byte copy, 128 VGPR pollers, 4 WG/CU", 2, cus * 4, 0);
	run<116, false>("data sentinel, 128 VGPR pollers, 4 WG/CU", 3, cus * 4, 0);
	run<8, true>("write-value, pollers touched the line before", 0, cus * 2, 100);
	run<100, false>("write-value, ~128 VGPR pollers, 4 WG/CU", 0, cus * 4, 0);
	run<100, false>("publish-kernel, ~128 VGPR pollers, 4 WG/CU", 1, cus * 4, 0);
	run<200, false>("publish-kernel, ~224 VGPR pollers, 2 WG/CU", 1, cus * 2, 0);
	return 0;
}
