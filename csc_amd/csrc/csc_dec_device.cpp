// csc_dec_device.cpp -- CSCDec_* of libcsc_mi355x.so served by k_decode_run (csc_dec_kernels.hip).
//
// The host thread owns the callbacks and the block framing, the GPU owns the stream state and all
// decoding work:
//   * MemIO::ReadBlock (csc_memio.cpp:5-81) is reproduced here read for read: when the kernel reports
//     that it ran out of RC (or BC) bytes, blocks are read from the ISeqInStream -- flag byte, optional
//     3-byte size, ONE Read for the payload -- until one of the wanted kind turns up; every block read
//     on the way is uploaded to its ring on the device, which is the reference's queue of the
//     "other" kind.
//   * CSCDec_Decode (csc_dec.cpp:740-777) loops over Decompress calls; each is one or more launches
//     of the resumable kernel; the decoded run is copied back and handed to ISeqOutStream::Write.
// No CPU decoding path exists in this library.
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/csc_mi355x.h"
#include "csc_device.h"

namespace cscmi {
void launch_decode_init(DecState *D, hipStream_t st);
void launch_decode_run(DecState *D, hipStream_t st);
void launch_decode_run_multi(DecState *const *states, uint32_t n, hipStream_t st);
hipStream_t pooled_stream(int device);                     // csc_host.cpp
void pooled_stream_release(int device, hipStream_t s);
}
using namespace cscmi;

namespace {

constexpr uint32_t kMagicDec = 0x43534344;   // "CSCD"

void *def_alloc(void *, size_t n) { return malloc(n); }
void def_free(void *, void *a) { free(a); }
ISzAlloc g_default_alloc = {def_alloc, def_free};

const char kWords[122][8] = {   // csc_filters.cpp:8-38; symbol 0x82+i expands to kWords[i]
    "ac","ad","ai","al","am","an","ar","as","at","ea","ec","ed","ee","el","en","er","es","et","id","ie",
    "ig","il","in","io","is","it","of","ol","on","oo","or","os","ou","ow","ul","un","ur","us","ba","be",
    "ca","ce","co","ch","de","di","ge","gh","ha","he","hi","ho","ra","re","ri","ro","rs","la","le","li",
    "lo","ld","ll","ly","se","si","so","sh","ss","st","ma","me","mi","ne","nc","nd","ng","nt","pa","pe",
    "ta","te","ti","to","th","tr","wa","ve",
    "all","and","but","dow","for","had","hav","her","him","his","man","mor","not","now","one","out",
    "she","the","was","wer","whi","whe","wit","you","any","are",
    "that","said","with","have","this","from","were","tion",
};

// A decoder's device state is ONE allocation (zero-filled on every use) and its pinned staging ONE allocation;
// both, and the HIP stream, are recycled through a per-process cache keyed by (device, sizes): an archive reader
// creates one decoder per task.
struct DecRes {
    int device;
    size_t dsize, hsize;
    uint8_t *dslab, *hslab;
    hipStream_t stream;
};
std::mutex g_dec_mu;
std::vector<DecRes *> g_dec_cache;
size_t g_dec_cache_bytes = 0;
constexpr size_t kDecCacheMaxBytes = 32ull << 30;

void dec_res_destroy(DecRes *r)
{
    if (!r) return;
    if (r->dslab) (void)hipFree(r->dslab);
    if (r->hslab) (void)hipHostFree(r->hslab);
    if (r->stream) cscmi::pooled_stream_release(r->device, r->stream);
    delete r;
}
void dec_cache_trim()
{
    std::vector<DecRes *> drop;
    {
        std::lock_guard<std::mutex> lk(g_dec_mu);
        drop.swap(g_dec_cache);
        g_dec_cache_bytes = 0;
    }
    for (DecRes *r : drop) dec_res_destroy(r);
}
DecRes *dec_res_get(int device, size_t dsize, size_t hsize)
{
    {
        std::lock_guard<std::mutex> lk(g_dec_mu);
        for (size_t i = 0; i < g_dec_cache.size(); i++) {
            DecRes *r = g_dec_cache[i];
            if (r->device == device && r->dsize == dsize && r->hsize == hsize) {
                g_dec_cache.erase(g_dec_cache.begin() + i);
                g_dec_cache_bytes -= dsize;
                r->stream = cscmi::pooled_stream(device);      // (a recycled entry holds none while it sits in the cache)
                if (!r->stream) { dec_res_destroy(r); return nullptr; }
                return r;
            }
        }
    }
    DecRes *r = new DecRes();
    memset(r, 0, sizeof(*r));
    r->device = device; r->dsize = dsize; r->hsize = hsize;
    bool ok = (r->stream = cscmi::pooled_stream(device)) != nullptr;     // shared, csc_host.cpp: creating a stream costs ~4 ms here
    if (ok && hipMalloc((void **)&r->dslab, dsize) != hipSuccess) {
        (void)hipGetLastError();
        dec_cache_trim();                                  // give cached slabs back and try once more
        ok = hipMalloc((void **)&r->dslab, dsize) == hipSuccess;
    }
    ok = ok && hipHostMalloc((void **)&r->hslab, hsize, hipHostMallocDefault) == hipSuccess;
    if (!ok) { dec_res_destroy(r); return nullptr; }
    return r;
}
void dec_res_put(DecRes *r)
{
    if (!r) return;
    cscmi::pooled_stream_release(r->device, r->stream);
    r->stream = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_dec_mu);
        if (g_dec_cache_bytes + r->dsize <= kDecCacheMaxBytes && g_dec_cache.size() < 1024) {
            g_dec_cache.push_back(r);
            g_dec_cache_bytes += r->dsize;
            return;
        }
    }
    dec_res_destroy(r);
}

struct DecInstance {
    uint32_t magic;
    DecRes *res;
    ISzAlloc *alloc;
    ISeqInStream *is;
    int device;
    hipStream_t stream;
    DecState *d_state;
    DecState h;               // host mirror: configuration + device pointers
    DecState *h_read;         // pinned read-back of the whole (small) state
    uint8_t *h_block;         // pinned staging for one block
    uint8_t *h_out;           // pinned decoded run
    uint32_t *h_qsize[2];     // host mirrors of the ring slot sizes
    uint32_t avail[2], taken[2];
    uint64_t consumed_view;   // GetCompressedSize as of the last launch
};

void free_all(DecInstance *x)
{
    (void)hipSetDevice(x->device);
    if (x->res) { (void)hipStreamSynchronize(x->res->stream); dec_res_put(x->res); }
    free(x->h_qsize[0]); free(x->h_qsize[1]);
    x->magic = 0;
    ISzAlloc *a = x->alloc;
    a->Free(a, x);
}

// MemIO::ReadBlock for a block that is not queued yet (csc_memio.cpp:17-79): read stream blocks until one of
// `kind` (1 RC, 0 BC) arrives; each block read goes to the device ring of its own kind.
int read_block(DecInstance *x, int kind)
{
    for (;;) {
        uint8_t fb, sb[3];
        size_t n = 1;
        x->is->Read(x->is, &fb, &n);
        if (n != 1) return -1;
        uint32_t cur = x->h.bsize;
        if (!((fb >> 6) & 1)) {
            n = 3;
            x->is->Read(x->is, sb, &n);
            if (n != 3) return -1;
            cur = ((uint32_t)sb[0] << 16) + ((uint32_t)sb[1] << 8) + sb[2];
        }
        if (!cur || cur > x->h.bsize) return -1;
        n = cur;
        x->is->Read(x->is, x->h_block, &n);       // ONE Read per payload; a short one is an error (:47-50)
        if (n != cur) return -1;
        int k2 = (fb >> 7) & 1;
        if (x->avail[k2] - x->taken[k2] >= x->h.qslots) {
            fprintf(stderr, "csc-mi355x: decoder block ring overflow\n");
            return -1;
        }
        uint32_t slot = x->avail[k2] % x->h.qslots;
        if (hipMemcpy(x->h.q[k2] + (size_t)slot * x->h.bsize, x->h_block, cur, hipMemcpyHostToDevice) != hipSuccess) return -1;
        x->h_qsize[k2][slot] = cur;
        x->avail[k2]++;
        if (k2 == kind) return 0;
    }
}

// run the kernel until the current Decompress call is complete; returns 0 / error code, *size = run length
int decompress(DecInstance *x, uint32_t *size)
{
    *size = 0;
    for (;;) {
        // publish what has been uploaded so far
        for (int k = 0; k < 2; k++)
            if (hipMemcpyAsync(x->h.qsize[k], x->h_qsize[k], sizeof(uint32_t) * x->h.qslots, hipMemcpyHostToDevice, x->stream) != hipSuccess) return CSCMI_DEVICE_ERROR;
        if (hipMemcpyAsync(&x->d_state->avail[0], x->avail, sizeof(x->avail), hipMemcpyHostToDevice, x->stream) != hipSuccess) return CSCMI_DEVICE_ERROR;
        launch_decode_run(x->d_state, x->stream);
        if (hipGetLastError() != hipSuccess) return CSCMI_DEVICE_ERROR;
        if (hipMemcpyAsync(x->h_read, x->d_state, offsetof(DecState, probs), hipMemcpyDeviceToHost, x->stream) != hipSuccess) return CSCMI_DEVICE_ERROR;
        if (hipStreamSynchronize(x->stream) != hipSuccess) return CSCMI_DEVICE_ERROR;
        const DecState &r = *x->h_read;
        x->taken[0] = r.taken[0]; x->taken[1] = r.taken[1];
        x->consumed_view = r.consumed + r.rd[0] + r.rd[1];
        switch (r.status) {
        case DEC_DONE:
            *size = r.out_size;
            return 0;
        case DEC_NEED_RC:
        case DEC_NEED_BC:
            if (read_block(x, r.status == DEC_NEED_RC ? 1 : 0) < 0)
                return r.phase == DEC_PH_PRIME ? -1 : READ_ERROR;     // csc_dec.cpp:669-671 vs :17-18
            break;
        case DEC_ERR_MINUS1: return -1;
        default: return DECODE_ERROR;
        }
    }
}

}  // namespace

extern "C" {

void CSCDec_ReadProperties(CSCProps *props, uint8_t *s)   // csc_dec.cpp:733-738
{
    props->dict_size = ((uint32_t)s[0] << 24) + ((uint32_t)s[1] << 16) + ((uint32_t)s[2] << 8) + s[3];
    props->csc_blocksize = ((uint32_t)s[4] << 16) + ((uint32_t)s[5] << 8) + s[6];
    props->raw_blocksize = ((uint32_t)s[7] << 16) + ((uint32_t)s[8] << 8) + s[9];
}

CSCDecHandle CSCDec_Create(const CSCProps *props, ISeqInStream *instream, ISzAlloc *alloc)   // csc_dec.cpp:692-720
{
    if (alloc == NULL) alloc = &g_default_alloc;
    if (props->dict_size > 1024 * kMB || props->dict_size < 32 * kKB) return NULL;
    if (props->csc_blocksize == 0 || props->raw_blocksize == 0) return NULL;
    if (CSCMI_DeviceCheck() != 0) return NULL;
    DecInstance *x = (DecInstance *)alloc->Alloc(alloc, sizeof(DecInstance));
    if (!x) return NULL;
    memset(x, 0, sizeof(*x));
    x->magic = kMagicDec; x->alloc = alloc; x->is = instream;
    DecState &h = x->h;
    h.wnd_size = (uint32_t)props->dict_size; h.bsize = props->csc_blocksize; h.raw_blocksize = props->raw_blocksize;
    h.qslots = 2 * (props->raw_blocksize / props->csc_blocksize + 1) + 16;
    bool ok = hipGetDevice(&x->device) == hipSuccess;
    // one device slab (zero-filled) and one pinned slab
    size_t doff = 0, hoff = 0;
    auto dtake = [&](size_t bytes) { size_t o = doff; doff += (bytes + 255) & ~(size_t)255; return o; };
    auto htake = [&](size_t bytes) { size_t o = hoff; hoff += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_wnd = dtake((size_t)h.wnd_size + 256), o_plit = dtake(2 * 65536 * sizeof(uint32_t));
    const size_t o_out = dtake((size_t)h.raw_blocksize + 256), o_swap = dtake(2 * (size_t)h.raw_blocksize + 256);
    const size_t o_q0 = dtake((size_t)h.qslots * h.bsize + 256), o_q1 = dtake((size_t)h.qslots * h.bsize + 256);
    const size_t o_qs0 = dtake(sizeof(uint32_t) * h.qslots), o_qs1 = dtake(sizeof(uint32_t) * h.qslots);
    const size_t o_ua = dtake(sizeof(uint32_t) * kDecUndoCap), o_uv = dtake(sizeof(uint32_t) * kDecUndoCap);
    const size_t o_words = dtake(sizeof(kWords)), o_state = dtake(sizeof(DecState));
    const size_t p_read = htake(sizeof(DecState)), p_block = htake(h.bsize), p_out = htake(h.raw_blocksize);
    if (ok) { x->res = dec_res_get(x->device, doff, hoff); ok = x->res != nullptr; }
    if (ok) {
        uint8_t *D = x->res->dslab, *H = x->res->hslab;
        x->stream = x->res->stream;
        h.wnd = D + o_wnd; h.p_lit = (uint32_t *)(D + o_plit); h.out = D + o_out; h.swap = D + o_swap;
        h.q[0] = D + o_q0; h.q[1] = D + o_q1; h.qsize[0] = (uint32_t *)(D + o_qs0); h.qsize[1] = (uint32_t *)(D + o_qs1);
        h.undo_addr = (uint32_t *)(D + o_ua); h.undo_val = (uint32_t *)(D + o_uv);
        h.words = D + o_words;
        x->d_state = (DecState *)(D + o_state);
        x->h_read = (DecState *)(H + p_read); x->h_block = H + p_block; x->h_out = H + p_out;
        ok = hipMemsetAsync(D, 0, doff, x->stream) == hipSuccess
          && hipMemcpyAsync(D + o_words, kWords, sizeof(kWords), hipMemcpyHostToDevice, x->stream) == hipSuccess;
    }
    x->h_qsize[0] = (uint32_t *)calloc(h.qslots, sizeof(uint32_t));
    x->h_qsize[1] = (uint32_t *)calloc(h.qslots, sizeof(uint32_t));
    ok = ok && x->h_qsize[0] && x->h_qsize[1];
    if (ok) {
        h.p_delta = h.p_lit + 65536;
        h.phase = DEC_PH_PRIME0;
        ok = hipMemcpyAsync(x->d_state, &h, sizeof(DecState), hipMemcpyHostToDevice, x->stream) == hipSuccess;
        if (ok) {
            launch_decode_init(x->d_state, x->stream);
            ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(x->stream) == hipSuccess;
        }
    }
    // CSCDecoder::Init already reads the first RC and BC block (csc_dec.cpp:336-337)
    if (!ok || read_block(x, 1) < 0 || read_block(x, 0) < 0) {
        if (!ok) fprintf(stderr, "csc-mi355x: decoder device allocation failed\n");
        free_all(x);
        return NULL;
    }
    return (CSCDecHandle)x;
}

void CSCDec_Destroy(CSCDecHandle p)   // csc_dec.cpp:722-731
{
    DecInstance *x = (DecInstance *)p;
    if (x && x->magic == kMagicDec) free_all(x);
}

int CSCDec_Decode(CSCDecHandle p, ISeqOutStream *os, ICompressProgress *progress)   // csc_dec.cpp:740-777
{
    DecInstance *x = (DecInstance *)p;
    if (hipSetDevice(x->device) != hipSuccess) return CSCMI_DEVICE_ERROR;
    int ret = 0;
    uint64_t outsize = 0;
    for (;;) {
        uint32_t size = 0;
        ret = decompress(x, &size);
        if (ret == 0) outsize += size;
        if (progress) progress->Progress(progress, x->consumed_view, outsize);
        if (size == 0 || ret < 0) break;
        if (hipMemcpy(x->h_out, x->h.out, size, hipMemcpyDeviceToHost) != hipSuccess) { ret = CSCMI_DEVICE_ERROR; break; }
        size_t wrote = os->Write(os, x->h_out, size);
        if (wrote == CSC_WRITE_ABORT) break;
        if (wrote < size) { ret = WRITE_ERROR; break; }
    }
    return ret;
}

#ifdef CSCMI_TIMERS
// development aid (section timers): only in the -DCSCMI_TIMERS build, never in the product library
void CSCMI_DebugDecTimers(CSCDecHandle p, uint64_t *out16)
{
    DecInstance *x = (DecInstance *)p;
    (void)hipMemcpy(out16, (const uint8_t *)x->d_state + offsetof(DecState, dbg), 16 * sizeof(uint64_t), hipMemcpyDeviceToHost);
}
#endif

// Not in the reference: CSCDec_Decode for n independent handles at once (the tasks of an archive).  Every round
// is ONE launch of k_decode_run_multi -- workgroup b advances stream b until it needs a block or has completed a
// Decompress call -- followed, per stream and on this thread, by exactly the callbacks CSCDec_Decode would make:
// the block reads MemIO::ReadBlock does for that stream, or the Write of its decoded run.  rcs[i] receives what
// CSCDec_Decode(hs[i], oss[i], NULL) would have returned.  Returns 0, or CSCMI_DEVICE_ERROR if the GPU side failed.
int CSCMI_DecodeBatch(int n, CSCDecHandle *hs, ISeqOutStream *const *oss, int *rcs)
{
    if (n <= 0) return 0;
    DecInstance *lead = (DecInstance *)hs[0];
    if (hipSetDevice(lead->device) != hipSuccess) return CSCMI_DEVICE_ERROR;
    hipStream_t st = lead->stream;
    DecState **d_list = nullptr, **h_list = nullptr;
    if (hipMalloc((void **)&d_list, sizeof(DecState *) * n) != hipSuccess) return CSCMI_DEVICE_ERROR;
    if (hipHostMalloc((void **)&h_list, sizeof(DecState *) * n, hipHostMallocDefault) != hipSuccess) { (void)hipFree(d_list); return CSCMI_DEVICE_ERROR; }
    std::vector<int> live;
    for (int i = 0; i < n; i++) { rcs[i] = 0; live.push_back(i); }
    int dev_rc = 0;
    while (!live.empty() && !dev_rc) {
        bool ok = true;
        for (size_t k = 0; k < live.size() && ok; k++) {
            DecInstance *x = (DecInstance *)hs[live[k]];
            for (int q = 0; q < 2 && ok; q++)
                ok = hipMemcpyAsync(x->h.qsize[q], x->h_qsize[q], sizeof(uint32_t) * x->h.qslots, hipMemcpyHostToDevice, st) == hipSuccess;
            ok = ok && hipMemcpyAsync(&x->d_state->avail[0], x->avail, sizeof(x->avail), hipMemcpyHostToDevice, st) == hipSuccess;
            h_list[k] = x->d_state;
        }
        ok = ok && hipMemcpyAsync(d_list, h_list, sizeof(DecState *) * live.size(), hipMemcpyHostToDevice, st) == hipSuccess;
        if (ok) {
            launch_decode_run_multi(d_list, (uint32_t)live.size(), st);
            ok = hipGetLastError() == hipSuccess;
        }
        for (size_t k = 0; k < live.size() && ok; k++) {
            DecInstance *x = (DecInstance *)hs[live[k]];
            ok = hipMemcpyAsync(x->h_read, x->d_state, offsetof(DecState, probs), hipMemcpyDeviceToHost, st) == hipSuccess;
        }
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (!ok) { dev_rc = CSCMI_DEVICE_ERROR; break; }
        // decoded runs back in one go
        for (int i : live) {
            DecInstance *x = (DecInstance *)hs[i];
            const DecState &r = *x->h_read;
            if (r.status == DEC_DONE && r.out_size)
                ok = ok && hipMemcpyAsync(x->h_out, x->h.out, r.out_size, hipMemcpyDeviceToHost, st) == hipSuccess;
        }
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (!ok) { dev_rc = CSCMI_DEVICE_ERROR; break; }
        std::vector<int> next;
        for (int i : live) {
            DecInstance *x = (DecInstance *)hs[i];
            const DecState &r = *x->h_read;
            x->taken[0] = r.taken[0]; x->taken[1] = r.taken[1];
            x->consumed_view = r.consumed + r.rd[0] + r.rd[1];
            bool keep = false;
            switch (r.status) {
            case DEC_DONE:
                if (r.out_size) {                                           // csc_dec.cpp:760-771
                    size_t wrote = oss[i]->Write(oss[i], x->h_out, r.out_size);
                    if (wrote == CSC_WRITE_ABORT) break;
                    if (wrote < r.out_size) { rcs[i] = WRITE_ERROR; break; }
                    keep = true;
                }
                break;
            case DEC_NEED_RC:
            case DEC_NEED_BC:
                if (read_block(x, r.status == DEC_NEED_RC ? 1 : 0) < 0) rcs[i] = r.phase == DEC_PH_PRIME ? -1 : READ_ERROR;
                else keep = true;
                break;
            case DEC_ERR_MINUS1: rcs[i] = -1; break;
            default: rcs[i] = DECODE_ERROR;
            }
            if (keep) next.push_back(i);
        }
        live.swap(next);
    }
    (void)hipFree(d_list);
    (void)hipHostFree(h_list);
    return dev_rc;
}

}  // extern "C"
