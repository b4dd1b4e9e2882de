// gfx950 probe: every register primitive of the packed kernels (ksw2_lane_pk.h) on the device against its C twin (what the
// simulator build runs), bit for bit, on 4 M random operand triples per primitive plus every v_perm_b32 selector byte value.
//   build + run (GPU box):
//     g++ -O2 -std=c++17 -c -o /tmp/lane_ops_twin.o tools/probe/lane_ops_twin.cpp
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -o /tmp/lane_ops_probe.o tools/probe/lane_ops_probe.hip
//     hipcc --offload-arch=gfx950 -o /tmp/lane_ops_probe /tmp/lane_ops_probe.o /tmp/lane_ops_twin.o && /tmp/lane_ops_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "lane_ops_probe.h"

extern "C" uint32_t lane_op_host(int op, uint32_t a, uint32_t b, uint32_t c);

__global__ void run(int op, const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = lane_op(op, a[i], b[i], c[i]);
}

static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }

int main()
{
	const int n = 1 << 22;
	std::vector<uint32_t> a(n), b(n), c(n), o(n);
	uint32_t *da, *db, *dc, *dout;
	if (hipMalloc(&da, 4 * n) != hipSuccess || hipMalloc(&db, 4 * n) != hipSuccess || hipMalloc(&dc, 4 * n) != hipSuccess || hipMalloc(&dout, 4 * n) != hipSuccess) { printf("no device memory\n"); return 2; }
	int bad_total = 0;
	for (int op = 0; op < OP_COUNT; ++op) {
		for (int i = 0; i < n; ++i) {
			a[i] = rnd(); b[i] = rnd(); c[i] = rnd();
			if (op == OP_PERM) {                       // selector bytes: every value 0..15 often, anything else sometimes
				uint32_t s = 0;
				for (int k = 0; k < 4; ++k) { const uint32_t r = rnd(); s |= ((r & 0x300) ? (r & 15) : (r & 0xff)) << (8 * k); }
				c[i] = s;
				if (i < 256) c[i] = (uint32_t)i * 0x01010101u;      // ... and each of the 256 byte values in all four positions
			}
			if (op == OP_MAX3U) {                      // the patterns the kernels feed it: 0x0000 .. 0x7BFF per half (denormals included)
				a[i] = (a[i] & 0x7fff7fffu); b[i] &= 0x7fff7fffu; c[i] &= 0x7fff7fffu;
				for (uint32_t *p : { &a[i], &b[i], &c[i] }) { if ((*p & 0xffffu) > 0x7BFFu) *p = (*p & 0xffff0000u) | (*p & 0x3fffu); if ((*p >> 16) > 0x7BFFu) *p = (*p & 0xffffu) | ((*p >> 16 & 0x3fffu) << 16); }
			}
			if (op == OP_SELV || op == OP_SEL) { if (i & 1) a[i] = ((a[i] & 1) ? 0xffffu : 0u) | ((a[i] & 2) ? 0xffff0000u : 0u); }      // half masks as the kernels use them, and arbitrary ones
		}
		hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 4 * n, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(run, dim3(n / 256), dim3(256), 0, 0, op, da, db, dc, dout, n);
		if (hipMemcpy(o.data(), dout, 4 * n, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
		int bad = 0;
		for (int i = 0; i < n; ++i) {
			const uint32_t h = lane_op_host(op, a[i], b[i], c[i]);
			if (h != o[i] && bad++ < 4) printf("  MISMATCH %s: a=%08x b=%08x c=%08x device=%08x twin=%08x\n", lane_op_name[op], a[i], b[i], c[i], o[i], h);
		}
		printf("%-58s %8d operand triples  %s\n", lane_op_name[op], n, bad ? "DIFFERS FROM ITS TWIN" : "bit-identical to its twin");
		bad_total += bad;
	}
	printf(bad_total ? "FAILED: %d mismatches\n" : "all primitives agree with their simulator twins (%d mismatches)\n", bad_total);
	return bad_total ? 1 : 0;
}
