// Do two tiny long-latency kernels from two host threads (a stream each) overlap on the device, or does the second wait for the first?
// (round 4, coalesced single-pair calls: a one-pair plan's device time read 0.33 ms alone and 0.66 ms next to another thread's batch.)
// Per thread: [h2d 4 KB] [kernel: one wavefront spinning ~300 us] [d2h 4 KB] [stream sync], in a loop; prints the time per iteration for
// 1, 2, 4, 8 threads, with the streams created plain / non-blocking, and with the copies left out.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cskp tools/probe/concurrent_small_kernels_probe.hip -lpthread && /tmp/cskp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void spin_kernel(int *out, long long ticks)
{
	const long long t0 = wall_clock64();
	int x = 0;
	while (wall_clock64() - t0 < ticks) x += 1;
	if (threadIdx.x == 0) out[blockIdx.x] = x;
}

static void worker(int copies, int nonblocking, int iters, double *ms, int events)
{
	hipStream_t s;
	char *h; int *d;
	hipEvent_t e0, e1;
	if (nonblocking) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); else CK(hipStreamCreate(&s));
	CK(hipHostMalloc(&h, 8192)); CK(hipMalloc(&d, 8192));
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int it = -20; it < iters; ++it) {
		if (it == 0) *ms = now();
		if (copies) CK(hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s));
		if (events) CK(hipEventRecord(e0, s));
		spin_kernel<<<1, 64, 0, s>>>(d + 1024, 30000);      // 100 MHz wall clock: 300 us
		if (events) CK(hipEventRecord(e1, s));
		if (copies) CK(hipMemcpyAsync(h + 4096, d + 1024, 4096, hipMemcpyDeviceToHost, s));
		CK(hipStreamSynchronize(s));
	}
	*ms = (now() - *ms) / iters;
	hipStreamDestroy(s); hipHostFree(h); hipFree(d);
}

int main()
{
	for (int events = 0; events < 2; ++events)
	for (int nonblocking = 0; nonblocking < 2; ++nonblocking)
		for (int copies = 0; copies < 2; ++copies)
			for (int T : {1, 2, 4, 8, 16}) {
				std::vector<std::thread> th; std::vector<double> ms(T);
				for (int t = 0; t < T; ++t) th.emplace_back(worker, copies, nonblocking, 200, &ms[t], events);
				for (auto &t : th) t.join();
				double mx = 0; for (double m : ms) mx = m > mx ? m : mx;
				printf("events %d, streams %s, copies %d, %2d threads: %.3f ms per iteration (slowest thread)\n", events, nonblocking ? "non-blocking" : "plain", copies, T, mx);
			}
	return 0;
}
