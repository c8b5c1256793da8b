/*
 * csc_mi355x.h -- C ABI of libcsc_mi355x.so, the MI355X-native drop-in for the libcsc
 * encode/decode path of fusiyuan2010/CSC.
 *
 * The eleven CSCEnc_ / CSCDec_ entry points are exactly the ones the reference exports
 * (inside EXTERN_C_BEGIN/END, compiled with -D_7Z_TYPES_); each declaration cites the reference
 * interface it replaces.  Plain pointers and sizes only -- no C++ or torch types cross this line,
 * and no C++ exception crosses it either.
 *
 *   reference header                      what it declares
 *   src/libcsc/csc_common.h:11-63         CSC_PROP_SIZE, error codes, CSC_WRITE_ABORT, CSCProps
 *   src/libcsc/Types.h:137-154,220-231    ISeqInStream, ISeqOutStream, ICompressProgress, ISzAlloc
 *   src/libcsc/csc_enc.h:11-30            CSCEncProps_Init .. CSCEnc_Encode_Flush
 *   src/libcsc/csc_dec.h:8-21             CSCDec_ReadProperties .. CSCDec_Decode
 *
 * A program that already includes the reference's csc_enc.h / csc_dec.h / Types.h can keep
 * including those and just link against this library: the struct layouts below are identical.
 */
#ifndef CSC_MI355X_H_
#define CSC_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSC_PROP_SIZE (4 + 3 + 3)          /* csc_common.h:11 */
#define DECODE_ERROR (-96)                 /* csc_common.h:13 */
#define WRITE_ERROR (-97)                  /* csc_common.h:14 */
#define READ_ERROR (-98)                   /* csc_common.h:15 */
#define CSC_WRITE_ABORT ((size_t)-1)       /* csc_common.h:17 */
/* Not in the reference: the GPU side failed (no device, HIP error, output arena exhausted, or the watchdog of the
 * multi-wavefront parser tripped -- a bug, reported instead of hanging the GPU).
 * Returned by CSCEnc_Encode / CSCEnc_Encode_Flush only; details go to stderr. */
#define CSCMI_DEVICE_ERROR (-95)

#ifndef CSC_MI355X_NO_7Z_TYPES            /* define this if the reference's Types.h is included already */
typedef int SRes;
typedef struct { SRes (*Read)(void *p, void *buf, size_t *size); } ISeqInStream;          /* Types.h:137-142 */
typedef struct { size_t (*Write)(void *p, const void *buf, size_t size); } ISeqOutStream; /* Types.h:149-154 */
typedef struct { SRes (*Progress)(void *p, uint64_t inSize, uint64_t outSize); } ICompressProgress; /* Types.h:220-225 */
typedef struct {                                                                            /* Types.h:227-231 */
    void *(*Alloc)(void *p, size_t size);
    void (*Free)(void *p, void *address); /* address can be 0 */
} ISzAlloc;
#endif

typedef struct _CSCProps {                 /* csc_common.h:19-63, field for field */
    size_t dict_size;
    uint32_t csc_blocksize;
    uint32_t raw_blocksize;
    uint8_t hash_bits;
    uint8_t hash_width;
    uint8_t bt_hash_bits;
    uint32_t bt_size;
    uint32_t bt_cyc;
    uint8_t good_len;
    uint8_t lz_mode;
    uint8_t DLTFilter;
    uint8_t TXTFilter;
    uint8_t EXEFilter;
} CSCProps;

typedef void *CSCEncHandle;                /* csc_enc.h:17 */
typedef void *CSCDecHandle;                /* csc_dec.h:10 */

/* ---- encoder, csc_enc.h ---- */
void CSCEncProps_Init(CSCProps *p, uint32_t dict_size, int level);                 /* csc_enc.h:11  (C++ defaults 64000000 / 2) */
void CSCEnc_WriteProperties(const CSCProps *props, uint8_t *stream, int full);     /* csc_enc.h:13 */
uint64_t CSCEnc_EstMemUsage(const CSCProps *props);                                /* csc_enc.h:15 */
CSCEncHandle CSCEnc_Create(const CSCProps *props, ISeqOutStream *outstream, ISzAlloc *alloc); /* csc_enc.h:20-22 */
void CSCEnc_Destroy(CSCEncHandle p);                                               /* csc_enc.h:24 */
int CSCEnc_Encode(CSCEncHandle p, ISeqInStream *instream, ICompressProgress *progress);       /* csc_enc.h:26-28 */
int CSCEnc_Encode_Flush(CSCEncHandle p);                                           /* csc_enc.h:30 */

/* ---- decoder, csc_dec.h ---- */
void CSCDec_ReadProperties(CSCProps *props, uint8_t *stream);                      /* csc_dec.h:8 */
CSCDecHandle CSCDec_Create(const CSCProps *props, ISeqInStream *instream, ISzAlloc *alloc);   /* csc_dec.h:13-15 */
void CSCDec_Destroy(CSCDecHandle p);                                               /* csc_dec.h:17 */
int CSCDec_Decode(CSCDecHandle p, ISeqOutStream *outstream, ICompressProgress *progress);     /* csc_dec.h:19-21 */

/* ---- extensions of this library (measurement + device-resident input); not in the reference ---- */
typedef struct {
    uint64_t chunks;              /* CSCEncoder::Compress calls served */
    uint64_t input_bytes;
    uint64_t output_bytes;        /* coder payload bytes (GetCompressedSize, csc_encoder_main.cpp:174) */
    uint64_t encode_launches;     /* k_encode_runs launches */
    double encode_kernel_ms;      /* HIP-event time of those launches, on the stream they ran on */
    double analyze_kernel_ms;     /* k_analyze + k_dup_check */
    uint64_t find_match_calls, slide_positions, bt_steps, literals, matches;
} CSCMIStats;

/* Same as one iteration of CSCEnc_Encode's read loop (csc_enc.cpp:170-181), but the <= raw_blocksize
 * chunk is already resident in device memory (bench.py keeps inputs in HBM).  Returns 0 or an error. */
int CSCMI_EncodeDeviceChunk(CSCEncHandle p, const void *device_ptr, size_t size);
/* n independent handles (tasks of a -p / per-extension split) advanced by one chunk each with ONE kernel
 * launch, one workgroup per stream; handles must live on the current device.  sizes[i] == 0 skips handle i. */
int CSCMI_EncodeDeviceChunkBatch(int n, CSCEncHandle *hs, const void *const *device_ptrs, const size_t *sizes);
/* CSCEnc_Encode_Flush (csc_enc.cpp:193-203) for n handles of the current device with one round trip: the EOF kernels queued on one
 * stream, one wait, the last coder blocks handed to the handles' output streams on this thread, in handle order.  Returns 0 or
 * the first error; the streams are byte for byte what n CSCEnc_Encode_Flush calls write. */
int CSCMI_FlushBatch(int n, CSCEncHandle *hs);
/* CSCDec_Decode for n independent handles at once: one kernel launch per round advances every stream (one
 * workgroup each); block reads and Write calls happen on the calling thread, per stream in the order CSCDec_Decode
 * would make them.  rcs[i] = what CSCDec_Decode(hs[i], oss[i], NULL) would return.  Returns 0 or CSCMI_DEVICE_ERROR. */
int CSCMI_DecodeBatch(int n, CSCDecHandle *hs, ISeqOutStream *const *oss, int *rcs);
/* Host-memory variant used by CSCEnc_Encode itself. */
int CSCMI_EncodeHostChunk(CSCEncHandle p, const void *host_ptr, size_t size);
void CSCMI_GetStats(CSCEncHandle p, CSCMIStats *out);
/* 0 if a usable gfx950-class HIP device is visible, else a negative code (and a message on stderr). */
int CSCMI_DeviceCheck(void);

#ifdef __cplusplus
}
#endif
#endif
