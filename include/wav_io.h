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

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { BUFFER_SIZE = 256, ID_LEN = 4 }; /* bytes scanned for chunk IDs; length of a chunk ID */

/* The three chunks of a canonical 44-byte header.  Field names are the reference's (its drivers
 * read header.format.sample_per_sec, .channels, .bits_per_sample ...); widths are spelled out. */
typedef struct RiffChunk {   /* bytes  0..11 */
  char ID[ID_LEN];           /* "RIFF"                 */
  int32_t size;              /* file size - 8          */
  char type[ID_LEN];         /* "WAVE"                 */
} RIFF_CHUNK;

typedef struct FormatChunk { /* bytes 12..35 */
  char ID[ID_LEN];           /* "fmt "                 */
  int32_t size;              /* 16 for PCM             */
  int16_t format;            /* 1 = PCM                */
  int16_t channels;
  int32_t sample_per_sec;
  int32_t avg_bytes_per_sec;
  int16_t blockAlign;
  int16_t bits_per_sample;
} FORMAT_CHUNK;

typedef struct DataChunk {   /* bytes 36..43 */
  char ID[ID_LEN];           /* "data"                 */
  int32_t size;              /* payload bytes          */
} DATA_CHUNK;

typedef struct WavHeader {
  RIFF_CHUNK riff;
  FORMAT_CHUNK format;
  DATA_CHUNK data;
} WAV_HEADER;

#ifndef __cplusplus
_Static_assert(sizeof(RIFF_CHUNK) == 12 && sizeof(FORMAT_CHUNK) == 24 && sizeof(DATA_CHUNK) == 8 &&
                   sizeof(WAV_HEADER) == 44,
               "the header structs are copied from the file image byte for byte");
#endif

/* chunk search inside a memory image of the file's first BUFFER_SIZE bytes */
int search_ID(const char* ID, char* buf, int buf_size, int* loc);
/* header in / out; both return 0 on success */
int read_header(WAV_HEADER* header, FILE* file);
int write_header(WAV_HEADER* header, FILE* file);
/* raw int16 samples; the return value is the fread / fwrite element count */
int read_samples(short* buf, int num_samples, WAV_HEADER* header, FILE* file);
int write_samples(short* buf, int num_samples, WAV_HEADER* header, FILE* file);
void print_header(WAV_HEADER* header);

#ifdef __cplusplus
}
#endif
#endif
