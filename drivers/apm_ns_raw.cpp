// apm_ns_raw -- exercises include/apm_ns.h (the reference's libapm APM_NS class) on a raw
// interleaved capture file, 10 ms at a time, the way a libapm client calls it
// (WebRtc_AMP_Port/libapm/src/apm_ns.cpp:47-130).
//
//   apm_ns_raw <in.raw> <out.raw> <channels> <mode 0|1|2> <s16|f32> [frequency = 16000 | 32000 | 48000]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "apm_ns.h"

int main(int argc, char** argv) {
  if (argc != 6 && argc != 7) {
    fprintf(stderr, "usage: %s in.raw out.raw channels mode s16|f32\n", argv[0]);
    return 2;
  }
  const int channels = atoi(argv[3]);
  const int mode = atoi(argv[4]);
  const bool is_float = strcmp(argv[5], "f32") == 0;
  const unsigned frequency = argc == 7 ? (unsigned)atoi(argv[6]) : 16000u;
  const int spc = (int)(frequency / 100);  // samples per channel per 10 ms
  APM_NS ns;
  // An uninitialised module must pass data through untouched (apm_ns.cpp:49-51).
  short probe[4] = {1, 2, 3, 4};
  ns.processCaptureStream(probe, 2, 2);
  if (probe[0] != 1 || probe[3] != 4) return 3;
  if (ns.initNsModule(8000, mode, 80, channels)) return 4;  // not covered by this build
  if (!ns.initNsModule(frequency, mode, spc, channels)) {
    fprintf(stderr, "initNsModule failed: %s\n", AspNs_last_error());
    return 1;
  }
  FILE* fi = fopen(argv[1], "rb");
  FILE* fo = fopen(argv[2], "wb");
  if (!fi || !fo) return 1;
  const size_t per_frame = (size_t)spc * channels;
  const size_t width = is_float ? sizeof(float) : sizeof(short);
  std::vector<char> buf(per_frame * width);
  while (fread(buf.data(), width, per_frame, fi) == per_frame) {
    if (is_float)
      ns.processCaptureStream(reinterpret_cast<float*>(buf.data()), spc, channels);
    else
      ns.processCaptureStream(reinterpret_cast<short*>(buf.data()), spc, channels);
    fwrite(buf.data(), width, per_frame, fo);
  }
  fclose(fi);
  fclose(fo);
  return 0;
}
