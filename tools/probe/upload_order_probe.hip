// Why did a pageable 3.6 MB hipMemcpyAsync block its caller for 15-25 ms (in steps of 5) when it followed two pieces of a page-locked
// (hipHostRegister) arena on the same stream -- in every process but the first one on a box (round 4, streamed plans of ksw2_host_plan.c)?
// One stream, per step: [zero block 64 KB] [2 pieces of the arena + 64 KB blocks] [small arrays] [the other pieces + blocks]; the time
// the small-array call takes on the host is printed for: arena registered vs hipHostMalloc'ed, small arrays pageable vs pinned.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/uop tools/probe/upload_order_probe.hip && /tmp/uop && /tmp/uop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void run(const char *name, bool arena_registered, bool small_pinned, bool with_blocks)
{
	const size_t total = 67u << 20, piece = total / 12, small = 3600u << 10, blk = 65536;
	char *arena, *smallsrc, *blocks, *d, *dsmall, *dblk;
	if (arena_registered) { arena = (char*)malloc(total); memset(arena, 1, total); CK(hipHostRegister(arena, total, hipHostRegisterDefault)); }
	else { CK(hipHostMalloc(&arena, total)); memset(arena, 1, total); }
	if (small_pinned) CK(hipHostMalloc(&smallsrc, small)); else smallsrc = (char*)malloc(small);
	memset(smallsrc, 2, small);
	CK(hipHostMalloc(&blocks, blk * 16)); memset(blocks, 3, blk * 16);
	CK(hipMalloc(&d, total)); CK(hipMalloc(&dsmall, small)); CK(hipMalloc(&dblk, blk));
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	std::vector<double> t_small, t_all;
	for (int step = 0; step < 12; ++step) {
		const double t0 = now();
		if (with_blocks) CK(hipMemcpyAsync(dblk, blocks, blk, hipMemcpyHostToDevice, s));
		for (int p = 0; p < 2; ++p) {
			CK(hipMemcpyAsync(d + p * piece, arena + p * piece, piece, hipMemcpyHostToDevice, s));
			if (with_blocks) CK(hipMemcpyAsync(dblk, blocks + (p + 1) * blk, blk, hipMemcpyHostToDevice, s));
		}
		const double t1 = now();
		CK(hipMemcpyAsync(dsmall, smallsrc, small, hipMemcpyHostToDevice, s));
		CK(hipMemsetAsync(dsmall, 0, 4u << 20 > small ? small : 4u << 20, s));
		const double t2 = now();
		for (int p = 2; p < 12; ++p) {
			CK(hipMemcpyAsync(d + p * piece, arena + p * piece, piece, hipMemcpyHostToDevice, s));
			if (with_blocks) CK(hipMemcpyAsync(dblk, blocks + (p + 1) * blk, blk, hipMemcpyHostToDevice, s));
		}
		CK(hipStreamSynchronize(s));
		t_small.push_back(t2 - t1); t_all.push_back(now() - t0);
	}
	printf("%-70s small-array calls %.3f / %.3f / %.3f ms (steps 2, 6, 11), whole step %.3f / %.3f / %.3f ms\n", name, t_small[2], t_small[6], t_small[11], t_all[2], t_all[6], t_all[11]);
	if (arena_registered) { hipHostUnregister(arena); free(arena); } else hipHostFree(arena);
	if (small_pinned) hipHostFree(smallsrc); else free(smallsrc);
	hipHostFree(blocks); hipFree(d); hipFree(dsmall); hipFree(dblk); hipStreamDestroy(s);
}

int main()
{
	run("arena hipHostMalloc, small arrays pageable, watermark blocks", false, false, true);
	run("arena registered,    small arrays pageable, watermark blocks", true, false, true);
	run("arena registered,    small arrays pageable, no blocks", true, false, false);
	run("arena registered,    small arrays pinned,   watermark blocks", true, true, true);
	run("arena hipHostMalloc, small arrays pinned,   watermark blocks", false, true, true);
	return 0;
}
