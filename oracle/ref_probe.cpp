/*
 * oracle/ref_probe.cpp -- extern "C" probes into the REFERENCE's own classes.
 *
 * TEST INFRASTRUCTURE.  Compiled only into oracle/_ref/libcsc_ref.so, next to
 * the reference sources taken from /root/reference where they lie (never
 * copied).  It lets tools/make_golden.py pin the intermediate stages
 * (analyzer verdicts, filter outputs) and not only whole streams.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <csc_analyzer.h>
#include <csc_filters.h>
#include <csc_default_alloc.h>
/* the encoder's match finder sits behind two `private:` sections (CSCEncoder::lz_, LZ::mf_); MatchFinder::pos_ itself is public */
#define private public
#include <csc_encoder_main.h>
#undef private

/* csc_enc.cpp:8-14 keeps this struct to itself; a CSCEncHandle points at one */
struct ProbeEncInstance { CSCEncoder *encoder; MemIO *io; ISzAlloc *alloc; uint32_t raw_blocksize; };

extern "C" {

/* Start the REFERENCE's position counter somewhere else (right after CSCEnc_Create: the tables are zeroed, every entry reads as out
 * of range whatever pos_ is) -- so that MatchFinder::normalize (csc_mf.cpp:108-114, pos_ >= 0xFFFFFFF0) runs inside a test-sized
 * input and the goldens of tools/make_golden_renorm.py pin the restatement's and the HIP path's version of it. */
void ref_debug_set_pos(void *h, uint32_t pos) { ((ProbeEncInstance *)h)->encoder->lz_.mf_.pos_ = pos; }
uint32_t ref_debug_get_pos(void *h) { return ((ProbeEncInstance *)h)->encoder->lz_.mf_.pos_; }

uint32_t ref_analyze_block(uint8_t *src, uint32_t size, uint32_t *bpb)
{
    Analyzer a;
    a.Init();
    return a.Analyze(src, size, bpb);
}

uint32_t ref_dlt_bpb(uint8_t *src, uint32_t size, uint32_t chn)
{
    Analyzer a;
    a.Init();
    return a.GetDltBpb(src, size, chn);
}

static Filters *new_filters()
{
    Filters *f = (Filters *)calloc(1, sizeof(Filters));
    f->Init(default_alloc);
    return f;
}
static void del_filters(Filters *f) { f->Destroy(); free(f); }

void ref_forward_e89(uint8_t *buf, uint32_t size) { Filters *f = new_filters(); f->Forward_E89(buf, size); del_filters(f); }
void ref_inverse_e89(uint8_t *buf, uint32_t size) { Filters *f = new_filters(); f->Inverse_E89(buf, size); del_filters(f); }
uint32_t ref_forward_dict(uint8_t *buf, uint32_t size) { Filters *f = new_filters(); uint32_t r = f->Foward_Dict(buf, size); del_filters(f); return r; }
void ref_inverse_dict(uint8_t *buf, uint32_t size) { Filters *f = new_filters(); f->Inverse_Dict(buf, size); del_filters(f); }
void ref_forward_delta(uint8_t *buf, uint32_t size, uint32_t chn) { Filters *f = new_filters(); f->Forward_Delta(buf, size, chn); del_filters(f); }
void ref_inverse_delta(uint8_t *buf, uint32_t size, uint32_t chn) { Filters *f = new_filters(); f->Inverse_Delta(buf, size, chn); del_filters(f); }

}
