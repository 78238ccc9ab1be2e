/*
 * wav_io.h -- minimal RIFF/WAVE header + int16 sample I/O, host C.
 *
 * Same types, names and semantics as the reference's wav_io
 * (WebRtc_AMP_Port/wav_io.h:10-44 == common/wav_parser/wav_io.h), so the
 * reference's test_*_module drivers compile against it unchanged:
 *   - the header is three packed chunks (12 + 24 + 8 = 44 bytes) taken from the
 *     first 256 bytes of the file by chunk-ID search (wav_io.c:32-85);
 *   - write_header forces fmt.size = 16 and copies every other field,
 *     including stale riff/data sizes, verbatim (wav_io.c:87-93);
 *   - read_samples / write_samples move raw int16, ignoring the channel count
 *     (wav_io.c:115-128).
 */
#ifndef ASP_WAV_IO_H_
#define ASP_WAV_IO_H_

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BUFFER_SIZE 256
#define ID_LEN 4

typedef struct {
  char ID[ID_LEN];
  int size;
  char type[ID_LEN];
} RIFF_CHUNK;

typedef struct {
  char ID[ID_LEN];
  int size;
  short format;
  short channels;
  int sample_per_sec;
  int avg_bytes_per_sec;
  short blockAlign;
  short bits_per_sample;
} FORMAT_CHUNK;

typedef struct {
  char ID[ID_LEN];
  int size;
} DATA_CHUNK;

typedef struct {
  RIFF_CHUNK riff;
  FORMAT_CHUNK format;
  DATA_CHUNK data;
} WAV_HEADER;

int search_ID(const char* ID, char* buf, int buf_size, int* loc);
int read_header(WAV_HEADER* header, FILE* file);
int write_header(WAV_HEADER* header, FILE* file);
int read_samples(short* buf, int num_samples, WAV_HEADER* header, FILE* file);
int write_samples(short* buf, int num_samples, WAV_HEADER* header, FILE* file);
void print_header(WAV_HEADER* header);

#ifdef __cplusplus
}
#endif
#endif
