// Issue-rate probe, round 3: SIMD-level cost (cycles per wave64 instruction with every resident wave
// issuing the same independent stream) of the instruction kinds the NS frame kernel is made of, at
// 1 / 2 / 4 / 8 waves per SIMD.  A frame step is priced with these (tools/ns_valu_budget.py --price).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/issue_probe3 tools/probe/issue_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8 instructions on 4 independent register sets per asm block, 8 blocks per iteration, 16 iterations
#define V4(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" \
                     op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail

template <int MODE>
__global__ void probe(unsigned long long* out, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  f32x2 p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed}, p3 = {seed, seed + 5};
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
  int s0 = 0;
  const float m = 1.0001f, c = 0.5f;
  const f32x2 pm = {1.0001f, 0.9999f};
  const double dm = 1.0001, dc = 0.5;
  unsigned long long mask = 0x5555555555555555ull;
  __shared__ float lds[4096];
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned la = (threadIdx.x & 63) * 8;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 16; ++it) {
#define BODY(txt, ...) REP8(asm volatile(txt __VA_ARGS__);)
    if (MODE == 0) { BODY(V4("v_add_f32", ", %4"), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)) }
    else if (MODE == 1) { BODY(V4("v_fma_f32", ", %4, %5"), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c)) }
    else if (MODE == 2) { BODY(V4("v_pk_fma_f32", ", %4, %4"), : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm)) }
    else if (MODE == 3) { BODY(V4("v_pk_mul_f32", ", %4"), : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm)) }
    else if (MODE == 4) { BODY(V4("v_pk_add_f32", ", %4"), : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm)) }
    else if (MODE == 5) { BODY(V4("v_fma_f64", ", %4, %5"), : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(dc)) }
    else if (MODE == 6) { BODY(V4("v_mul_f64", ", %4"), : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm)) }
    else if (MODE == 7) { BODY(V4("v_add_f64", ", %4"), : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc)) }
    else if (MODE == 8) { BODY("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)) }
    else if (MODE == 9) { BODY("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5\n v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "s"(mask)) }
    else if (MODE == 10) { BODY("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc") }
    else if (MODE == 11) { BODY("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc") }
    else if (MODE == 12) { BODY("v_readlane_b32 %4, %0, 3\n v_readlane_b32 %4, %1, 5\n v_readlane_b32 %4, %2, 7\n v_readlane_b32 %4, %3, 9\n v_readlane_b32 %4, %0, 3\n v_readlane_b32 %4, %1, 5\n v_readlane_b32 %4, %2, 7\n v_readlane_b32 %4, %3, 9", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0)) }
    else if (MODE == 13) { BODY("v_writelane_b32 %0, %4, 3\n v_writelane_b32 %1, %4, 5\n v_writelane_b32 %2, %4, 7\n v_writelane_b32 %3, %4, 9\n v_writelane_b32 %0, %4, 3\n v_writelane_b32 %1, %4, 5\n v_writelane_b32 %2, %4, 7\n v_writelane_b32 %3, %4, 9", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0)) }
    else if (MODE == 14) { BODY("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)) }
    else if (MODE == 15) { BODY(V4("v_xor_b32", ", %4"), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)) }
    else if (MODE == 16) { BODY("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5", : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1)) }
    else if (MODE == 17) { BODY("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %4\n v_cvt_f32_f64 %3, %5\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %4\n v_cvt_f32_f64 %3, %5", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1)) }
    else if (MODE == 18) { BODY(V4("v_rcp_f32", ""), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)) }
    else if (MODE == 19) { BODY(V4("v_rsq_f32", ""), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)) }
    else if (MODE == 20) { BODY("s_mov_b32 %0, 0x3f800001\n s_mov_b32 %0, 0x3f800002\n s_mov_b32 %0, 0x3f800003\n s_mov_b32 %0, 0x3f800004\n s_mov_b32 %0, 0x3f800001\n s_mov_b32 %0, 0x3f800002\n s_mov_b32 %0, 0x3f800003\n s_mov_b32 %0, 0x3f800004", : "+s"(s0)) }
    else if (MODE == 21) { BODY("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0", : "+s"(s0)) }
    else if (MODE == 22) { BODY("v_permlane32_swap_b32_e32 %0, %1\n v_permlane32_swap_b32_e32 %2, %3\n v_permlane32_swap_b32_e32 %0, %1\n v_permlane32_swap_b32_e32 %2, %3\n v_permlane32_swap_b32_e32 %0, %1\n v_permlane32_swap_b32_e32 %2, %3\n v_permlane32_swap_b32_e32 %0, %1\n v_permlane32_swap_b32_e32 %2, %3", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)) }
    else if (MODE == 23) { BODY("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)\n ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2560\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3584\n s_waitcnt lgkmcnt(0)", : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(la)) }
    else if (MODE == 24) { BODY(V4("v_max_f32", ", %4"), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)) }
    else if (MODE == 25) { BODY(V4("v_pk_mov_b32", ", %4"), : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm)) }
    else if (MODE == 26) { BODY(V4("v_lshrrev_b32", ", 3"), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)) }
    else if (MODE == 27) { BODY("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_mirror row_mask:0xf bank_mask:0xf", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)) }
    else if (MODE == 28) { BODY("v_readfirstlane_b32 %4, %0\n v_readfirstlane_b32 %4, %1\n v_readfirstlane_b32 %4, %2\n v_readfirstlane_b32 %4, %3\n v_readfirstlane_b32 %4, %0\n v_readfirstlane_b32 %4, %1\n v_readfirstlane_b32 %4, %2\n v_readfirstlane_b32 %4, %3", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0)) }
    else if (MODE == 29) { BODY("v_cmp_lt_f32 %5, %0, %4\n v_cmp_lt_f32 %5, %1, %4\n v_cmp_lt_f32 %5, %2, %4\n v_cmp_lt_f32 %5, %3, %4\n v_cmp_lt_f32 %5, %0, %4\n v_cmp_lt_f32 %5, %1, %4\n v_cmp_lt_f32 %5, %2, %4\n v_cmp_lt_f32 %5, %3, %4", : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "s"(mask)) }
    else if (MODE == 30) { BODY("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %4\n v_cvt_f64_i32 %3, %5\n v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %4\n v_cvt_f64_i32 %3, %5", : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1)) }
    else if (MODE == 31) { BODY(V4("v_mul_f32", ", %4"), : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m)) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3) + (float)s0 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) | (acc == 12345.f ? 1ull << 63 : 0);
}

template <int MODE>
void run(const char* name) {
  printf("%-30s", name);
  for (int w : {1, 2, 4, 8}) {
    if (w == 8 && false) continue;
    unsigned long long* d;
    const int waves = 4 * w;  // one workgroup on one CU
    (void)hipMalloc(&d, waves * 8);
    probe<MODE><<<1, 64 * waves>>>(d, 1.0f);
    probe<MODE><<<1, 64 * waves>>>(d, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)(v & ~(1ull << 63));
    avg /= waves;
    printf("  w%d: %5.2f/wave %5.2f/SIMD", w, avg / 1024.0, avg / 1024.0 / w);
    (void)hipFree(d);
  }
  printf("\n");
}

int main() {
  printf("cycles per instruction: per wave, and per SIMD (= per wave / resident waves)\n");
  run<0>("v_add_f32");
  run<31>("v_mul_f32");
  run<1>("v_fma_f32");
  run<24>("v_max_f32");
  run<2>("v_pk_fma_f32");
  run<3>("v_pk_mul_f32");
  run<4>("v_pk_add_f32");
  run<25>("v_pk_mov_b32");
  run<5>("v_fma_f64");
  run<6>("v_mul_f64");
  run<7>("v_add_f64");
  run<16>("v_cvt_f64_f32");
  run<17>("v_cvt_f32_f64");
  run<30>("v_cvt_f64_i32");
  run<8>("v_mov_b32");
  run<15>("v_xor_b32");
  run<26>("v_lshrrev_b32");
  run<9>("v_cndmask_b32 (sgpr mask)");
  run<10>("v_cmp_lt_f32 -> vcc");
  run<29>("v_cmp_lt_f32 -> sgpr pair");
  run<11>("v_cmp -> vcc -> v_cndmask");
  run<12>("v_readlane_b32");
  run<28>("v_readfirstlane_b32");
  run<13>("v_writelane_b32");
  run<14>("v_add_f32_dpp");
  run<27>("v_mov_b32_dpp");
  run<22>("v_permlane32_swap");
  run<18>("v_rcp_f32");
  run<19>("v_rsq_f32");
  run<20>("s_mov_b32");
  run<21>("s_nop 0");
  run<23>("ds_read_b64 x4 + wait");
  return 0;
}
