/* tests/abi_client.c -- a C (not C++) client that includes ONLY the public header: custom
 * ISzAlloc, short-read ISeqInStream, collecting / failing ISeqOutStream.  Mirrors how
 * csa_worker.cpp:23-56 drives the encoder.  Usage: abi_client <in> <out> <level> <dict> <max_read> <fail_after>
 * exit code = 0 on success, 10 + (-rc) on an encoder error. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "csc_mi355x.h"

typedef struct { ISeqInStream s; FILE *f; size_t max_read; } In;
typedef struct { ISeqOutStream s; FILE *f; size_t written, fail_after; } Out;
static size_t g_allocs = 0;

static SRes in_read(void *p, void *buf, size_t *size)
{
    In *in = (In *)p;                       /* the interface pointer itself is `p` (csc_enc.cpp:171) */
    size_t want = *size;
    if (in->max_read && want > in->max_read) want = in->max_read;
    *size = fread(buf, 1, want, in->f);
    return 0;
}
static size_t out_write(void *p, const void *buf, size_t size)
{
    Out *o = (Out *)p;
    if (o->fail_after && o->written + size > o->fail_after) return 0;
    o->written += size;
    return fwrite(buf, 1, size, o->f);
}
static void *my_alloc(void *p, size_t n) { (void)p; g_allocs++; return calloc(1, n ? n : 1); }
static void my_free(void *p, void *a) { (void)p; free(a); }

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    In in; Out out; ISzAlloc al; CSCProps props; unsigned char hdr[CSC_PROP_SIZE];
    in.s.Read = in_read; in.f = fopen(argv[1], "rb"); in.max_read = (size_t)atol(argv[5]);
    out.s.Write = out_write; out.f = fopen(argv[2], "wb"); out.written = 0; out.fail_after = (size_t)atol(argv[6]);
    al.Alloc = my_alloc; al.Free = my_free;
    if (!in.f || !out.f) return 3;
    CSCEncProps_Init(&props, (uint32_t)atol(argv[4]), atoi(argv[3]));
    CSCEncHandle h = CSCEnc_Create(&props, &out.s, &al);
    if (!h) return 4;
    CSCEnc_WriteProperties(&props, hdr, 0);
    fwrite(hdr, 1, CSC_PROP_SIZE, out.f);     /* the caller writes the 10-byte header (csa_worker.cpp:38-42) */
    int rc = CSCEnc_Encode(h, &in.s, NULL);
    int rc2 = CSCEnc_Encode_Flush(h);
    CSCEnc_Destroy(h);
    fclose(in.f); fclose(out.f);
    if (rc < 0) return 10 + (-rc);
    if (rc2 < 0) return 10 + (-rc2);
    return g_allocs > 0 ? 0 : 5;
}
