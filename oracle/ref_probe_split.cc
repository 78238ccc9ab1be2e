// ref_probe_split.cc -- glue compiled INTO oracle/_ref/libsplit_ref.so next to the reference's own
// SplittingFilter sources (compiled in place from /root/reference; see oracle/Makefile).  TEST
// INFRASTRUCTURE ONLY; contains no algorithm: it feeds one channel of int16 samples through
// SplittingFilter::Analysis / Synthesis (modules/audio_processing/splitting_filter.cc:28-170), the
// calls AudioBuffer::SplitIntoFrequencyBands / MergeFrequencyBands make (audio_buffer.cc:455-463).
#include <vector>
#include <string.h>
#include "webrtc/modules/audio_processing/splitting_filter.h"
#include "webrtc/modules/audio_processing/channel_buffer.h"
using namespace webrtc;
struct RefSplit {
  SplittingFilter f; IFChannelBuffer in; IFChannelBuffer b1, b2, b3;
  RefSplit(int n, int nb) : f(1), in(n, 1), b1(n / nb, 1), b2(n / nb, 1), b3(n / nb, 1) {}
};
extern "C" {
void* ref_split_create(int samples, int num_bands) { return new RefSplit(samples, num_bands); }
void ref_split_free(void* h) { delete (RefSplit*)h; }
void ref_split_analysis(void* h, const int16_t* x, int n, int nb, int16_t* bands) {
  RefSplit* r = (RefSplit*)h;
  memcpy(r->in.ibuf()->channel(0), x, n * sizeof(int16_t));
  std::vector<IFChannelBuffer*> v; v.push_back(&r->b1); v.push_back(&r->b2); if (nb == 3) v.push_back(&r->b3);
  r->f.Analysis(&r->in, v);
  for (int k = 0; k < nb; ++k) memcpy(bands + k * (n / nb), v[k]->ibuf_const()->channel(0), (n / nb) * sizeof(int16_t));
}
void ref_split_synthesis(void* h, const int16_t* bands, int n, int nb, int16_t* out) {
  RefSplit* r = (RefSplit*)h;
  std::vector<IFChannelBuffer*> v; v.push_back(&r->b1); v.push_back(&r->b2); if (nb == 3) v.push_back(&r->b3);
  for (int k = 0; k < nb; ++k) memcpy(v[k]->ibuf()->channel(0), bands + k * (n / nb), (n / nb) * sizeof(int16_t));
  r->f.Synthesis(v, &r->in);
  memcpy(out, r->in.ibuf_const()->channel(0), n * sizeof(int16_t));
}
}
