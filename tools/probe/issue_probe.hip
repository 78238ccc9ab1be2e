// Issue-rate probe: cycles per wave64 instruction of a few VALU forms on one SIMD (1 and 2 waves per
// SIMD), independent and dependent chains.  Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void probe(unsigned long long* out, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float2v p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed};
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
  const float m = 1.0001f, c = 0.5f;
  const float2v pm = {m, m}, pc = {c, c};
  const double dm = 1.0001, dc = 0.5;
  __builtin_amdgcn_s_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 16; ++it) {
    if (MODE == 0) {  // 8 independent f32 fma chains
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (MODE == 1) {  // one dependent f32 chain
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)
    } else if (MODE == 2) {  // 4 independent packed chains
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));)
    } else if (MODE == 3) {  // one dependent packed chain
      REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pc));)
    } else if (MODE == 4) {  // 4 independent f64 chains
      REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(dc));)
    } else if (MODE == 5) {  // one dependent f64 chain
      REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(dm), "v"(dc));)
    } else if (MODE == 6) {  // independent v_rcp_f32
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 7) {  // v_pk_mul_f32 + v_pk_add_f32 independent
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                        "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));)
    } else if (MODE == 8) {  // f64 add + mul independent
      REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %5\n"
                        "v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %5\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(dc));)
    } else if (MODE == 9) {  // v_cndmask / v_cmp pairs
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                        "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + (float)(d0 + d1 + d2 + d3);
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) | (acc == 12345.f ? 1ull << 63 : 0);
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  unsigned long long* d;
  const int waves = 4 * waves_per_simd;  // one workgroup on one CU
  hipMalloc(&d, waves * 8);
  probe<MODE><<<1, 64 * waves>>>(d, 1.0f);
  probe<MODE><<<1, 64 * waves>>>(d, 1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(waves);
  hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += (double)(v & ~(1ull << 63));
  avg /= waves;
  printf("%-34s waves/SIMD %d: %.2f cycles per instruction per wave (%.0f cycles / 1024 instr)\n", name, waves_per_simd, avg / 1024.0, avg);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 3, 4, 8}) {
    run<0>("v_fma_f32 independent", w);
    run<1>("v_fma_f32 dependent chain", w);
    run<2>("v_pk_fma_f32 independent", w);
    run<3>("v_pk_fma_f32 dependent chain", w);
    run<4>("v_fma_f64 independent", w);
    run<5>("v_fma_f64 dependent chain", w);
    run<6>("v_rcp_f32 independent", w);
    run<7>("v_pk_mul/add_f32 independent", w);
    run<8>("v_mul/add_f64 independent", w);
    run<9>("v_cmp + v_cndmask", w);
  }
  return 0;
}
