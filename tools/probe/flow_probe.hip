// flow_probe.hip -- round 4: what a frame step costs at the launch boundary, and whether the boundary can
// be replaced by a per-stream hand-off in memory.
//
//   part 1  the dependent-launch gap: an empty kernel of the NS step's grid, K launches on one HIP
//           stream, on two streams side by side, and with hipExtAnyOrderLaunch (is it honoured here?)
//   part 2  does an any-order launch start before its predecessor on the same stream has ended?
//   part 3  the NS step's memory shape (tools/probe/mem_probe.hip: per stream 12 state rows of 512 B, the
//           scalar row, the sliding buffers, 640 B in and out) plus a compute stand-in, run three ways:
//             chains   the product's shape: one launch per step and chain, plain loads / stores
//             flowA    ONE stream, any-order launches; stream s of step k waits for seq[s] == k
//             flowS    step k on HIP stream k % NS (plain launches, no cross-stream events); same wait
//           In the flow forms every state access is sc1 (write-through stores, L1-bypassing loads), the
//           wave drains its stores (s_waitcnt vmcnt(0)) and lane 0 publishes seq[s] = k + 1 with an sc1
//           store; the spin is bounded and sets an abort word (MI355X_MICROARCH.md, visibility; guide
//           section 6 G16 R1).  Every step adds 1 to every state word, so a stale read shows as a lost
//           increment: the probe counts them.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/flow_probe tools/probe/flow_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kStreamDwords = 2432;
typedef __attribute__((address_space(1))) unsigned gu32;

__global__ __launch_bounds__(256, 4) void empty_kernel(float* p) {
  if (p != nullptr && threadIdx.x == 9999) p[0] = 1.f;
}

__global__ void long_kernel(unsigned long long* t, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { t[0] = t0; t[1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void stamp_kernel(unsigned long long* t) {
  if (threadIdx.x == 0) t[2] = __builtin_amdgcn_s_memrealtime();
}

struct Rsrc { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ Rsrc make_rsrc(float* p, int bytes) {
  Rsrc b;
  b.r = __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
  return b;
}

// WORK: number of dependent-chain FMA rounds (8 accumulators each), 0 = none
template <bool FLOW, int WORK, int NCH = 8>
__global__ __launch_bounds__(256, 4) void step_kernel(float* __restrict__ state, unsigned* seq, unsigned* abort_w,
                                                      const float* __restrict__ in, float* __restrict__ out,
                                                      int S, unsigned want) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int stream = blockIdx.x * 4 + wv;
  if (stream >= S) return;
  float* st = state + (size_t)stream * kStreamDwords;
  constexpr int AUX = FLOW ? 16 : 0;  // sc1
  if (FLOW) {
    gu32* f = (gu32*)(seq + stream);
    unsigned spins = 0;
    for (;;) {
      const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_readfirstlane(v) == want) break;
      ++spins;
      if ((spins & 63) == 0) {
        const unsigned a = __hip_atomic_load((gu32*)abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(a) != 0) return;
      }
      if (spins > (1u << 17)) {
        if (lane == 0) __hip_atomic_store((gu32*)abort_w, 1u + stream, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const Rsrc b = make_rsrc(st, kStreamDwords * 4);
  float sv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, lane * 4, 0, AUX));
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  f4 s4;
  if (lane < 24) s4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (64 + 4 * lane) * 4, 0, AUX));
  else s4 = *reinterpret_cast<const f4*>(in + (size_t)stream * 160 + 4 * (lane - 24));
  f2 carry = {0.f, 0.f};
  if (lane < 48) carry = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(b.r, (160 + 2 * lane) * 4, 0, AUX));
  f2 r[12];
#pragma unroll
  for (int f = 0; f < 12; ++f)
    r[f] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(b.r, (256 + f * 128 + 2 * lane) * 4, 0, AUX));
  if (WORK > 0) {
    float a[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) a[i] = r[i].y + sv * 0.f;
#pragma unroll 4
    for (int k = 0; k < WORK; ++k) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) a[i] = __builtin_fmaf(a[i], 0.999f, r[i].y);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc += a[i];
    if (acc == 123.456f) r[0].x += 1.f;  // never true for these inputs; keeps the chain alive
  }
  if (lane >= 40) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, s4), b.r, (64 + 4 * (lane - 40)) * 4, 0, AUX);
#pragma unroll
  for (int f = 0; f < 12; ++f) {
    f2 v = {r[f].x + 1.f, r[f].y};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), b.r, (256 + f * 128 + 2 * lane) * 4, 0, AUX);
  }
  if (lane < 48) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, carry), b.r, (160 + 2 * lane) * 4, 0, AUX);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv + 1.f), b.r, lane * 4, 0, AUX);
  if (lane < 40) *reinterpret_cast<f4*>(out + (size_t)stream * 160 + 4 * lane) = s4;
  if (FLOW) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store((gu32*)(seq + stream), want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <bool FLOW, int WORK, int NCH>
static void launch_step(int grid, hipStream_t s, bool any, float* sp, unsigned* seq, unsigned* ab, const float* ip, float* op,
                        int n, unsigned want) {
  if (any)
    hipExtLaunchKernelGGL((step_kernel<FLOW, WORK, NCH>), dim3(grid), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, sp, seq, ab,
                          ip, op, n, want);
  else
    hipLaunchKernelGGL((step_kernel<FLOW, WORK, NCH>), dim3(grid), dim3(256), 0, s, sp, seq, ab, ip, op, n, want);
}

// mode 0: chains (plain, `nq` chains over the batch); 1: flowA (one stream, any-order); 2: flowS (step k on stream k % nq)
template <int WORK, int NCH = 8>
static void run_part3(int S, int mode, int nq, int steps) {
  float *state, *in, *out;
  unsigned *seq, *ab;
  CK(hipMalloc(&state, (size_t)S * kStreamDwords * 4));
  CK(hipMalloc(&in, (size_t)S * 160 * 4 * 8));
  CK(hipMalloc(&out, (size_t)S * 160 * 4 * 8));
  CK(hipMalloc(&seq, (size_t)S * 4));
  CK(hipMalloc(&ab, 64));
  CK(hipMemset(in, 0, (size_t)S * 160 * 4 * 8));
  std::vector<hipStream_t> st(nq);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork;
  std::vector<hipEvent_t> join(nq);
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (auto& j : join) CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  double best = 1e30;
  long bad_total = 0; unsigned abort_host = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(state, 0, (size_t)S * kStreamDwords * 4));
    CK(hipMemset(seq, 0, (size_t)S * 4));
    CK(hipMemset(ab, 0, 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st[0]));
    CK(hipEventRecord(fork, st[0]));
    for (int c = 1; c < nq; ++c) CK(hipStreamWaitEvent(st[c], fork, 0));
    for (int k = 0; k < steps; ++k) {
      const float* ip = in + (size_t)(k % 8) * S * 160;
      float* op = out + (size_t)(k % 8) * S * 160;
      if (mode == 0) {
        const int per = S / nq;
        for (int c = 0; c < nq; ++c)
          launch_step<false, WORK, NCH>((per + 3) / 4, st[c], false, state + (size_t)c * per * kStreamDwords, seq, ab, ip + (size_t)c * per * 160,
                                   op + (size_t)c * per * 160, per, (unsigned)k);
      } else if (mode == 1) {
        launch_step<true, WORK, NCH>((S + 3) / 4, st[0], true, state, seq, ab, ip, op, S, (unsigned)k);
      } else {
        launch_step<true, WORK, NCH>((S + 3) / 4, st[k % nq], false, state, seq, ab, ip, op, S, (unsigned)k);
      }
    }
    for (int c = 1; c < nq; ++c) { CK(hipEventRecord(join[c], st[c])); CK(hipStreamWaitEvent(st[0], join[c], 0)); }
    CK(hipEventRecord(e1, st[0]));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms * 1e3 / steps < best) best = ms * 1e3 / steps;
    // verification: every state word of the rows' .x, the scalar row == steps
    std::vector<float> h((size_t)S * kStreamDwords);
    CK(hipMemcpy(h.data(), state, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&abort_host, ab, 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int s = 0; s < S; ++s) {
      const float* p = h.data() + (size_t)s * kStreamDwords;
      for (int l = 0; l < 64; ++l) bad += p[l] != (float)steps;
      for (int f = 0; f < 12; ++f) for (int l = 0; l < 64; ++l) bad += p[256 + f * 128 + 2 * l] != (float)steps;
    }
    bad_total += bad;
    if (abort_host) break;
  }
  const char* names[3] = {"chains", "flowA ", "flowS "};
  printf("S %5d work %4d x %d %s nq %d: %.2f us/step  %.0f GB/s  lost-increment words %ld  abort %u\n", S, WORK, NCH, names[mode], nq, best,
         15716.0 * S / best / 1e3, bad_total, abort_host);
  fflush(stdout);
  for (auto& s : st) CK(hipStreamDestroy(s));
  CK(hipFree(state)); CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(seq)); CK(hipFree(ab));
}

int main(int argc, char** argv) {
  const int steps = 400;
  // ---- part 1
  for (int grid : {512, 1024}) {
    for (int mode = 0; mode < 3; ++mode) {  // 0: one stream; 1: two streams; 2: one stream any-order
      hipStream_t s[2];
      for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
      hipEvent_t e0, e1, fork, join;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s[0]));
        if (mode == 1) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); }
        for (int k = 0; k < steps; ++k) {
          if (mode == 2) hipExtLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s[0], nullptr, nullptr, hipExtAnyOrderLaunch, (float*)nullptr);
          else {
            hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s[0], (float*)nullptr);
            if (mode == 1) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s[1], (float*)nullptr);
          }
        }
        if (mode == 1) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
        CK(hipEventRecord(e1, s[0]));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3 / steps < best) best = ms * 1e3 / steps;
      }
      const char* nm[3] = {"one stream", "two streams (per pair of launches)", "one stream, any-order"};
      printf("empty kernel, grid %d x 256: %.2f us per launch, %s\n", grid, best, nm[mode]);
      for (auto& x : s) CK(hipStreamDestroy(x));
    }
  }
  // ---- part 2
  {
    unsigned long long* t; CK(hipMalloc(&t, 64)); CK(hipMemset(t, 0, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int any = 0; any < 2; ++any) {
      hipLaunchKernelGGL(long_kernel, dim3(1), dim3(64), 0, s, t, 5000);  // 50 us
      if (any) hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, t);
      else hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, t);
      CK(hipStreamSynchronize(s));
      unsigned long long h[3]; CK(hipMemcpy(h, t, 24, hipMemcpyDeviceToHost));
      printf("%s second launch: started %.2f us after the first one's START, %.2f us after its END\n", any ? "any-order" : "plain    ",
             (double)(long long)(h[2] - h[0]) / 100.0, (double)(long long)(h[2] - h[1]) / 100.0);
    }
    CK(hipStreamDestroy(s)); CK(hipFree(t));
  }
  fflush(stdout);
  // ---- part 3
  const bool quick = argc > 1;
  for (int S : {4096, 8192}) {
    if (!quick) {
      run_part3<0>(S, 0, 1, steps);
      run_part3<0>(S, 0, 2, steps);
      run_part3<0>(S, 1, 1, steps);
      run_part3<0>(S, 2, 2, steps);
      run_part3<0>(S, 2, 3, steps);
      run_part3<180>(S, 0, 1, steps);
      run_part3<180>(S, 0, 2, steps);
      run_part3<180>(S, 1, 1, steps);
      run_part3<180>(S, 2, 2, steps);
      run_part3<180>(S, 2, 3, steps);
    }
    // the same instruction count with fewer independent chains (the real step is a dependent chain:
    // 1454 VALU at ~8.7 cycles each = 5.3 us for a wave alone)
    run_part3<720, 2>(S, 0, 2, steps);
    run_part3<720, 2>(S, 2, 2, steps);
    run_part3<1440, 1>(S, 0, 2, steps);
    run_part3<1440, 1>(S, 2, 2, steps);
    run_part3<1440, 1>(S, 2, 3, steps);
    run_part3<2000, 1>(S, 0, 2, steps);
    run_part3<2000, 1>(S, 2, 2, steps);
  }
  return 0;
}
