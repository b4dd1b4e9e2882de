// host side of tools/probe/lane_ops_probe.hip: the C twins of the packed kernels' primitives, compiled by g++ (no HIP): exactly what
// the lock-step simulator (tests/sim) executes.
#include "lane_ops_probe.h"
extern "C" uint32_t lane_op_host(int op, uint32_t a, uint32_t b, uint32_t c) { return lane_op(op, a, b, c); }
