/*
 * oracle/orc_api.h -- C view of the libcsc boundary, used ONLY by the oracle.
 *
 * TEST INFRASTRUCTURE.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Restates the public types of the reference boundary:
 *   CSCProps             /root/reference/src/libcsc/csc_common.h:19-63
 *   ISeqInStream         /root/reference/src/libcsc/Types.h:137-142
 *   ISeqOutStream        /root/reference/src/libcsc/Types.h:149-154
 *   ICompressProgress    /root/reference/src/libcsc/Types.h:220-225
 *   ISzAlloc             /root/reference/src/libcsc/Types.h:227-231
 *   error codes          /root/reference/src/libcsc/csc_common.h:13-17
 */
#ifndef ORC_API_H_
#define ORC_API_H_

#include <stddef.h>
#include <stdint.h>

#define CSC_PROP_SIZE 10
#define DECODE_ERROR (-96)
#define WRITE_ERROR (-97)
#define READ_ERROR (-98)
#define CSC_WRITE_ABORT ((size_t)-1)

typedef int SRes;

typedef struct { SRes (*Read)(void *p, void *buf, size_t *size); } ISeqInStream;
typedef struct { size_t (*Write)(void *p, const void *buf, size_t size); } ISeqOutStream;
typedef struct { SRes (*Progress)(void *p, uint64_t in_size, uint64_t out_size); } ICompressProgress;
typedef struct {
    void *(*Alloc)(void *p, size_t size);
    void (*Free)(void *p, void *address);
} ISzAlloc;

typedef struct _CSCProps {
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

typedef void *CSCEncHandle;
typedef void *CSCDecHandle;

void CSCEncProps_Init(CSCProps *p, uint32_t dict_size, int level);
void CSCEnc_WriteProperties(const CSCProps *props, uint8_t *stream, int full);
uint64_t CSCEnc_EstMemUsage(const CSCProps *props);
CSCEncHandle CSCEnc_Create(const CSCProps *props, ISeqOutStream *outstream, ISzAlloc *alloc);
void CSCEnc_Destroy(CSCEncHandle p);
int CSCEnc_Encode(CSCEncHandle p, ISeqInStream *instream, ICompressProgress *progress);
int CSCEnc_Encode_Flush(CSCEncHandle p);

void CSCDec_ReadProperties(CSCProps *props, uint8_t *stream);
CSCDecHandle CSCDec_Create(const CSCProps *props, ISeqInStream *instream, ISzAlloc *alloc);
void CSCDec_Destroy(CSCDecHandle p);
int CSCDec_Decode(CSCDecHandle p, ISeqOutStream *outstream, ICompressProgress *progress);

/* ---- oracle-only probes (intermediate goldens; not part of the boundary) ---- */
/* Analyzer::Analyze on one <=8 KiB block: returns type, writes bpb (untouched for DT_SKIP). */
uint32_t orc_analyze_block(const uint8_t *src, uint32_t size, uint32_t *bpb);
uint32_t orc_dlt_bpb(const uint8_t *src, uint32_t size, uint32_t chn);
void orc_forward_e89(uint8_t *buf, uint32_t size);
void orc_inverse_e89(uint8_t *buf, uint32_t size);
uint32_t orc_forward_dict(uint8_t *buf, uint32_t size);
void orc_inverse_dict(uint8_t *buf, uint32_t size);
void orc_forward_delta(uint8_t *buf, uint32_t size, uint32_t chn);
void orc_inverse_delta(uint8_t *buf, uint32_t size, uint32_t chn);
/* the two float-built tables */
void orc_tables(uint32_t p2bits[512], uint32_t logtab[513]);

#endif
