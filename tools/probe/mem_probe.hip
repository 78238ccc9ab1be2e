// mem_probe.hip -- the memory floor of the NS frame step: a kernel that moves exactly the bytes of
// ns_frame1_kernel (per stream: 12 state rows of 512 B, the 256-B scalar row, two 384-B sliding
// buffers, 640 B of samples in and out) with no arithmetic, launched like the product path
// (one launch per step and chain, `chains` HIP streams).  Prints us per step of all streams.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/mem_probe tools/probe/mem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kStreamDwords = 2432;

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(float* __restrict__ state, const float* __restrict__ in, float* __restrict__ out, int S) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int stream = blockIdx.x * 4 + wv;
  if (stream >= S) return;
  float* st = state + (size_t)stream * kStreamDwords;
  float sv = st[lane];
  const float* src = lane < 24 ? st + 64 + 4 * lane : in + (size_t)stream * 160 + 4 * (lane - 24);
  float4 s4 = *reinterpret_cast<const float4*>(src);
  float2 carry = lane < 48 ? *reinterpret_cast<const float2*>(st + 160 + 2 * lane) : make_float2(0.f, 0.f);
  float2 r[12];
#pragma unroll
  for (int f = 0; f < 12; ++f) r[f] = *reinterpret_cast<const float2*>(st + 256 + f * 128 + 2 * lane);
  if (MODE == 1) {  // a dependent pause of about 4 us between loads and stores (s_sleep 64-cycle units)
    float acc = s4.x + sv + carry.x;
#pragma unroll
    for (int f = 0; f < 12; ++f) acc += r[f].x;
    if (acc == 123.456f) out[0] = acc;
    for (int i = 0; i < 150; ++i) __builtin_amdgcn_s_sleep(1);
  }
  if (lane >= 40) *reinterpret_cast<float4*>(st + 64 + 4 * (lane - 40)) = s4;
#pragma unroll
  for (int f = 0; f < 12; ++f) *reinterpret_cast<float2*>(st + 256 + f * 128 + 2 * lane) = make_float2(r[f].x + 1.f, r[f].y);
  if (lane < 48) *reinterpret_cast<float2*>(st + 160 + 2 * lane) = carry;
  st[lane] = sv + 1.f;
  if (lane < 40) *reinterpret_cast<float4*>(out + (size_t)stream * 160 + 4 * lane) = s4;
}

int main(int argc, char** argv) {
  const int steps = 400;
  for (int mode = 0; mode < 2; ++mode)
  for (int S : {1024, 2048, 4096, 8192}) {
    for (int chains : {1, 2, 4}) {
      float *state, *in, *out;
      CK(hipMalloc(&state, (size_t)S * kStreamDwords * 4));
      CK(hipMalloc(&in, (size_t)S * 160 * 4 * 8));
      CK(hipMalloc(&out, (size_t)S * 160 * 4 * 8));
      CK(hipMemset(state, 0, (size_t)S * kStreamDwords * 4));
      CK(hipMemset(in, 0, (size_t)S * 160 * 4 * 8));
      std::vector<hipStream_t> st(chains);
      for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      hipEvent_t e0, e1, fork; std::vector<hipEvent_t> join(chains);
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
      for (auto& j : join) CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
      const int per = S / chains;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st[0]));
        CK(hipEventRecord(fork, st[0]));
        for (int c = 1; c < chains; ++c) CK(hipStreamWaitEvent(st[c], fork, 0));
        for (int k = 0; k < steps; ++k)
          for (int c = 0; c < chains; ++c) {
            float* sp = state + (size_t)c * per * kStreamDwords;
            const float* ip = in + ((size_t)(k % 8) * S + (size_t)c * per) * 160;
            float* op = out + ((size_t)(k % 8) * S + (size_t)c * per) * 160;
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3((per + 3) / 4), dim3(256), 0, st[c], sp, ip, op, per);
            else hipLaunchKernelGGL(probe<1>, dim3((per + 3) / 4), dim3(256), 0, st[c], sp, ip, op, per);
          }
        for (int c = 1; c < chains; ++c) { CK(hipEventRecord(join[c], st[c])); CK(hipStreamWaitEvent(st[0], join[c], 0)); }
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / steps;
      printf("mode %d S %5d chains %d: %.2f us/step  %.0f GB/s (15716 B/stream)\n", mode, S, chains, us, 15716.0 * S / us / 1e3);
      for (auto& s : st) CK(hipStreamDestroy(s));
      CK(hipFree(state)); CK(hipFree(in)); CK(hipFree(out));
    }
  }
  return 0;
}
