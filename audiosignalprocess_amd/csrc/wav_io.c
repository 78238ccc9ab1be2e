/*
 * wav_io.c -- host-side WAV header / int16 sample I/O (no GPU involvement).
 * Behaviour follows the reference's WebRtc_AMP_Port/wav_io.c:32-128; see
 * include/wav_io.h for the contract.
 */
#include "wav_io.h"

#include <string.h>

/* First offset in buf[0..buf_size) where the bytes of ID occur (wav_io.c:14-30). */
int search_ID(const char* ID, char* buf, int buf_size, int* loc) {
  const int n = (int)strlen(ID);
  for (int off = 0; off + n <= buf_size; ++off) {
    if (memcmp(buf + off, ID, (size_t)n) == 0) {
      *loc = off;
      return 0;
    }
  }
  return -1;
}

/* Finds chunk `id` at or after *pos, copies `bytes` of it to dst and advances
 * *pos past the copied struct; -1 when absent or truncated. */
static int take_chunk(const char* id, char* buf, int len, int* pos, void* dst, int bytes) {
  int loc = 0;
  if (search_ID(id, buf + *pos, len - *pos, &loc) != 0) return -1;
  *pos += loc;
  if (len - *pos < bytes) return -1;
  memcpy(dst, buf + *pos, (size_t)bytes);
  *pos += bytes;
  return 0;
}

int read_header(WAV_HEADER* header, FILE* file) {
  char buf[BUFFER_SIZE];
  int pos = 0;
  const int len = (int)fread(buf, 1, BUFFER_SIZE, file);
  if (len <= 0) return -1;
  if (take_chunk("RIFF", buf, len, &pos, &header->riff, (int)sizeof(RIFF_CHUNK)) != 0) return -1;
  if (memcmp(header->riff.type, "WAVE", 4) != 0) return -1;
  if (take_chunk("fmt ", buf, len, &pos, &header->format, (int)sizeof(FORMAT_CHUNK)) != 0)
    return -1;
  if (take_chunk("data", buf, len, &pos, &header->data, (int)sizeof(DATA_CHUNK)) != 0) return -1;
  /* leave the file positioned on the first sample (wav_io.c:83) */
  fseek(file, -(long)(len - pos), SEEK_CUR);
  return 0;
}

int write_header(WAV_HEADER* header, FILE* file) {
  header->format.size = 16;
  fwrite(header, sizeof(WAV_HEADER), 1, file);
  return 0;
}

void print_header(WAV_HEADER* header) {
  printf("RIFF_CHUNK:\n    ID: RIFF\n    SIZE: 0x%x\n    TYPE: WAVE\n", header->riff.size);
  printf("FORMAT_CHUNK:\n    ID: fmt \n    SIZE: %d\n    FormatTag:0x%x\n", header->format.size,
         header->format.format);
  printf("    Channels: %d\n    SamplePerSec: %d\n    AvgBytesPerSec: %d\n",
         header->format.channels, header->format.sample_per_sec,
         header->format.avg_bytes_per_sec);
  printf("    BlockAlign: %d\n    BitsPerSample: %d\n", header->format.blockAlign,
         header->format.bits_per_sample);
  printf("DATA_CHUNK:\n    ID: data\n    SIZE: 0x%x\n", header->data.size);
}

int read_samples(short* buf, int num_samples, WAV_HEADER* header, FILE* file) {
  (void)header;
  return (int)fread(buf, sizeof(short), (size_t)num_samples, file);
}

int write_samples(short* buf, int num_samples, WAV_HEADER* header, FILE* file) {
  (void)header;
  return (int)fwrite(buf, sizeof(short), (size_t)num_samples, file);
}
