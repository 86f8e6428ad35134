// Experiment: what does a barrier among work-groups of ONE XCD cost with device-scope atomics (performed at the memory side)
// against atomics that stay in the XCD's L2 (no sc1) read back with sc1 loads -- and is the second kind coherent there?
// 16 work-groups on one XCD (blocks b with b % 8 == 0 of a 128-block launch), N barriers; every member bumps GE once per barrier and
// spins until GE reaches 16 * round.  Prints ticks (100 MHz) per barrier and checks the final count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int MODE>
__global__ void barrier_kernel(unsigned* sy, int rounds, unsigned long long* out, unsigned* xccOut, unsigned long long limitTicks) {
  const int b = blockIdx.x;
  if (b % 8 != 0 || threadIdx.x != 0) return;
  const int member = b / 8, G = gridDim.x / 8;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xccOut[member] = xcc & 15u;
  unsigned* ge = sy;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned target = 0;
  bool ok = true;
  for (int r = 0; r < rounds && ok; ++r) {
    target += (unsigned)G;
    if (MODE == 0) __hip_atomic_fetch_add(ge, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(ge, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (unsigned spin = 0;; ++spin) {
      unsigned v;
      if (MODE == 2) v = __hip_atomic_load(ge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else v = __hip_atomic_load(ge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(v - target) >= 0) break;
      if ((spin & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > limitTicks) { ok = false; break; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[member] = ok ? t1 - t0 : ~0ull;
}
int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 16, rounds = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned* sy; unsigned long long* out; unsigned* xcc;
  hipMalloc(&sy, 1 << 20); hipMalloc(&out, 64 * 8); hipMalloc(&xcc, 64 * 4);
  for (int mode = 0; mode < 3; ++mode)
    for (int off = 0; off < 5; ++off) {
      unsigned* p = sy + off * 1024;   // 4 KB apart
      hipMemset(p, 0, 256); hipMemset(out, 0, 64 * 8);
      hipDeviceSynchronize();
      if (mode == 0) barrier_kernel<0><<<8 * G, 64>>>(p, rounds, out, xcc, 300000000ull);
      if (mode == 1) barrier_kernel<1><<<8 * G, 64>>>(p, rounds, out, xcc, 300000000ull);
      if (mode == 2) barrier_kernel<2><<<8 * G, 64>>>(p, rounds, out, xcc, 300000000ull);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(G); std::vector<unsigned> hx(G); unsigned ge = 0;
      hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost); hipMemcpy(&ge, p, 4, hipMemcpyDeviceToHost);
      unsigned long long worst = 0; bool ok = true; unsigned mask = 0;
      for (int i = 0; i < G; ++i) { if (h[i] == ~0ull) ok = false; else if (h[i] > worst) worst = h[i]; mask |= 1u << hx[i]; }
      printf("mode %d (%s) offset %5d: %s, %.3f us per barrier, final count %u (want %u), XCC mask 0x%x\n", mode,
             mode == 0 ? "agent atomics + agent loads" : mode == 1 ? "workgroup-scope atomics + agent loads" : "workgroup-scope atomics + loads",
             off * 4096, ok ? "ok" : "TIMED OUT", worst / 100.0 / rounds, ge, (unsigned)(G * rounds), mask);
      fflush(stdout);
    }
  return 0;
}
