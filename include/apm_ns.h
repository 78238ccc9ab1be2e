/*
 * apm_ns.h -- the reference's multi-channel capture wrapper, class APM_NS
 * (WebRtc_AMP_Port/libapm/include/apm_ns.h:15-56, src/apm_ns.cpp:7-157), as a
 * header-only C++ layer over the C-ABI of asp_ns.h.  Same class name, method
 * names, argument meaning and "silently ignore when not initialised" behaviour.
 *
 * The reference keeps one NsHandle per channel and loops over channels on the
 * host; here the channels are the streams of one AspNsBatch and a 10 ms frame
 * of every channel is denoised by one fused launch.  The interleave <-> planar
 * shuffles stay on the host, as in the reference.
 *
 * Data path per frame (apm_ns.cpp:46-86 / 88-130): interleaved samples ->
 * planar int16 (FloatToS16 for the float overload) -> float-S16 ->
 * WebRtcNs_Analyze + WebRtcNs_Process -> FloatS16ToS16 -> interleaved.
 *
 * This build covers 16 kHz (one band, 160 samples per channel), 32 kHz (320 per channel, two
 * bands) and 48 kHz (480 per channel, three bands): AudioBuffer::SplitIntoFrequencyBands /
 * MergeFrequencyBands (audio_buffer.cc:455-463) are the SplittingFilter of asp_split.h, the
 * suppressor runs with one or two high bands.  8 kHz makes initNsModule return false.
 */
#ifndef ASP_APM_NS_H_
#define ASP_APM_NS_H_

#include <cstdint>
#include <vector>

#include "asp_ns.h"
#include "asp_split.h"

enum {
  NS_Mode_Mild = 0,
  Ns_Mode_Mideum,
  Ns_Mode_Aggressive,
};

class APM_NS {
 public:
  APM_NS() : m_batch(nullptr), m_qmf(nullptr), m_bands(1), m_frequency(0), m_channels(0), m_ns_mode(0), m_device(0),
             init_flag(false) {}
  ~APM_NS() { release(); }
  APM_NS(const APM_NS&) = delete;
  APM_NS& operator=(const APM_NS&) = delete;

  /* Extension: choose the GPU before initNsModule (default 0). */
  void setDevice(int device) { m_device = device; }

  /* apm_ns.cpp:7-45.  input_frames is the per-channel frame size (160). */
  bool initNsModule(unsigned int frequency, int ns_mode, int input_frames, int input_channels) {
    release();
    m_frequency = frequency;
    m_channels = input_channels;
    m_ns_mode = ns_mode;
    if (m_channels <= 0) return false;
    if (frequency != 16000 && frequency != 32000 && frequency != 48000) return false;
    m_bands = (int)(frequency / 16000);
    if (input_frames != m_bands * ASP_NS_BLOCKL) return false;
    const bool two_bands = m_bands > 1;
    if (AspNsBatch_Create(&m_batch, m_channels, m_device) != ASP_OK) return false;
    if (two_bands && AspSplitBatch_Create(&m_qmf, m_channels, m_bands, m_device) != ASP_OK) {
      release();
      return false;
    }
    if (AspNsBatch_Init(m_batch, frequency) != ASP_OK ||
        AspNsBatch_set_policy(m_batch, ns_mode) != ASP_OK) {
      release();
      return false;
    }
    m_planar.assign((size_t)m_channels * input_frames, 0);
    if (two_bands) {
      m_low.assign((size_t)m_bands * m_channels * ASP_NS_BLOCKL, 0);      // [band][channel][160]
      m_lowf.assign((size_t)m_bands * m_channels * ASP_NS_BLOCKL, 0.f);
    }
    init_flag = true;
    return true;
  }

  /* apm_ns.cpp:47-86: float samples in [-1, 1], interleaved, processed in place. */
  void processCaptureStream(float* data, int samples_per_channel, int input_channels) {
    if (!usable(samples_per_channel, input_channels)) return;
    for (int c = 0; c < input_channels; ++c)
      for (int j = 0; j < samples_per_channel; ++j)
        m_planar[(size_t)c * samples_per_channel + j] = floatToS16(data[(size_t)j * input_channels + c]);
    if (!denoisePlanar()) return;
    for (int c = 0; c < input_channels; ++c)
      for (int j = 0; j < samples_per_channel; ++j)
        data[(size_t)j * input_channels + c] = s16ToFloat(m_planar[(size_t)c * samples_per_channel + j]);
  }

  /* apm_ns.cpp:88-130: int16 PCM, interleaved, processed in place. */
  void processCaptureStream(short* data, int samples_per_channel, int input_channels) {
    if (!usable(samples_per_channel, input_channels)) return;
    for (int c = 0; c < input_channels; ++c)
      for (int j = 0; j < samples_per_channel; ++j)
        m_planar[(size_t)c * samples_per_channel + j] = data[(size_t)j * input_channels + c];
    if (!denoisePlanar()) return;
    for (int c = 0; c < input_channels; ++c)
      for (int j = 0; j < samples_per_channel; ++j)
        data[(size_t)j * input_channels + c] = m_planar[(size_t)c * samples_per_channel + j];
  }

 private:
  /* common_audio/include/audio_util.h:27-33 */
  static int16_t floatToS16(float v) {
    if (v > 0) return v >= 1 ? (int16_t)32767 : (int16_t)(v * 32767 + 0.5f);
    return v <= -1 ? (int16_t)-32768 : (int16_t)(-v * -32768 - 0.5f);
  }
  /* common_audio/include/audio_util.h:35-39 */
  static float s16ToFloat(int16_t v) {
    const float kMaxInt16Inverse = 1.f / 32767;
    const float kMinInt16Inverse = 1.f / -32768;
    return v * (v > 0 ? kMaxInt16Inverse : -kMinInt16Inverse);
  }
  /* common_audio/include/audio_util.h:41-49 */
  static int16_t floatS16ToS16(float v) {
    const float kMaxRound = 32767 - 0.5f, kMinRound = -32768 + 0.5f;
    if (v > 0) return v >= kMaxRound ? (int16_t)32767 : (int16_t)(v + 0.5f);
    return v <= kMinRound ? (int16_t)-32768 : (int16_t)(v - 0.5f);
  }
  bool usable(int samples_per_channel, int input_channels) const {
    return init_flag && input_channels == m_channels &&
           samples_per_channel == m_bands * ASP_NS_BLOCKL;
  }
  /* m_planar [channels][samples] int16, in place: apm_ns.cpp:66-77 */
  bool denoisePlanar() {
    if (!m_qmf)
      return AspNsBatch_AnalyzeProcessS16(m_batch, m_planar.data(), m_planar.data(), 1, ASP_MEM_HOST) == ASP_OK;
    const size_t n = (size_t)m_channels * ASP_NS_BLOCKL;  // one band of every channel
    if (AspSplitBatch_Analysis(m_qmf, m_planar.data(), m_low.data(), ASP_MEM_HOST) != ASP_OK) return false;
    // int16 -> float-S16 views of the bands (channel_buffer.cc:43-53)
    for (size_t i = 0; i < n * m_bands; ++i) m_lowf[i] = (float)m_low[i];
    if (AspNsBatch_AnalyzeProcessBands(m_batch, m_lowf.data(), m_lowf.data() + n, m_lowf.data(),
                                       m_lowf.data() + n, 1, ASP_MEM_HOST) != ASP_OK)
      return false;
    // back to the int16 bands (channel_buffer.cc:55-61)
    for (size_t i = 0; i < n * m_bands; ++i) m_low[i] = floatS16ToS16(m_lowf[i]);
    return AspSplitBatch_Synthesis(m_qmf, m_low.data(), m_planar.data(), ASP_MEM_HOST) == ASP_OK;
  }
  void release() {
    if (m_batch) AspNsBatch_Free(m_batch);
    if (m_qmf) AspSplitBatch_Free(m_qmf);
    m_batch = nullptr;
    m_qmf = nullptr;
    init_flag = false;
  }

  AspNsBatch* m_batch;
  AspSplitBatch* m_qmf;
  int m_bands;
  std::vector<int16_t> m_planar, m_low;
  std::vector<float> m_lowf;
  unsigned int m_frequency;
  int m_channels;
  int m_ns_mode;
  int m_device;
  bool init_flag;  // ns module has been initialised successfully
};

#endif /* ASP_APM_NS_H_ */
