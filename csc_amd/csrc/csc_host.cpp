// csc_host.cpp -- host side of libcsc_mi355x.so: the reference's C API (csc_enc.h) served by
// the gfx950 kernels of csc_kernels.hip.
//
// What stays on the host (SURVEY.md section 3.1 puts the device boundary inside
// CSCEncoder::Compress): the user callbacks (always on the calling thread), the 2 MiB read loop
// (csc_enc.cpp:160-191), the run segmentation over the analyzer's per-block verdicts
// (csc_encoder_main.cpp:85-147 -- a few integer compares per 8 KiB block plus the one
// `>= bpb * 0.95` double compare) and the RC/BC block framing (csc_memio.cpp:83-108).  Everything
// that touches the data -- analyzer, filters, match finder, parser, model, range/bit coder -- runs
// in HIP kernels on state that lives in HBM.  There is NO CPU fallback: without a HIP device
// CSCEnc_Create fails loudly and returns NULL.
#include <hip/hip_runtime.h>
#include <chrono>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/csc_mi355x.h"
#include "csc_device.h"

namespace cscmi {
hipError_t upload_tables();
void launch_init_state(EncState *S, hipStream_t st);
void launch_analyze(EncState *S, uint32_t chunk_size, const double *ent_coef, hipStream_t st);
void launch_dup_check(EncState *S, uint32_t chunk_size, uint32_t first, uint32_t count, hipStream_t st);
void launch_encode_runs(int parser, EncState *S, const RunDesc *runs, uint32_t nruns, uint32_t reset_arena, hipStream_t st);
void launch_encode_eof(EncState *S, hipStream_t st);
void launch_encode_runs_multi(int parser, uint32_t nstreams, EncState *const *states, const RunDesc *const *runs,
                              const uint32_t *nruns, const uint32_t *reset, hipStream_t st);
#ifdef CSCMI_STAGE_TEST
void launch_stage_filter(EncState *S, uint32_t kind, uint32_t size, uint32_t chn, uint32_t *result, hipStream_t st);
#endif
hipStream_t pooled_stream(int device);
void pooled_stream_release(int device, hipStream_t s);
}  // namespace cscmi

using namespace cscmi;

namespace {

constexpr uint32_t kMagicEnc = 0x43534345;           // "CSCE"
constexpr int kEventPairs = 32;

void *def_alloc(void *, size_t n) { return malloc(n); }   // csc_default_alloc.cpp:5-17
void def_free(void *, void *a) { free(a); }
ISzAlloc g_default_alloc = {def_alloc, def_free};

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t err__ = (call);                                                                \
        if (err__ != hipSuccess) {                                                                \
            fprintf(stderr, "csc-mi355x: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(err__), \
                    __FILE__, __LINE__);                                                          \
            return CSCMI_DEVICE_ERROR;                                                            \
        }                                                                                         \
    } while (0)

// the 122 words of the dictionary filter -- data the stream format is defined by (csc_filters.cpp:8-38)
const char *const kWords[122] = {
    "ac","ad","ai","al","am","an","ar","as","at","ea","ec","ed","ee","el","en","er","es","et","id","ie",
    "ig","il","in","io","is","it","of","ol","on","oo","or","os","ou","ow","ul","un","ur","us","ba","be",
    "ca","ce","co","ch","de","di","ge","gh","ha","he","hi","ho","ra","re","ri","ro","rs","la","le","li",
    "lo","ld","ll","ly","se","si","so","sh","ss","st","ma","me","mi","ne","nc","nd","ng","nt","pa","pe",
    "ta","te","ti","to","th","tr","wa","ve",
    "all","and","but","dow","for","had","hav","her","him","his","man","mor","not","now","one","out",
    "she","the","was","wer","whi","whe","wit","you","any","are",
    "that","said","with","have","this","from","were","tion",
};

// Filters::MakeWordTree, csc_filters.cpp:87-111: node numbering follows insertion order
void build_trie(uint16_t *next, uint8_t *sym)
{
    memset(next, 0, sizeof(uint16_t) * 300 * 26);
    memset(sym, 0, 300);
    uint32_t nodes = 1;
    uint8_t code = 0x82;
    for (int w = 0; w < 122; w++) {
        uint32_t pos = 0;
        for (const char *p = kWords[w]; *p; p++) {
            uint32_t idx = (uint32_t)(*p - 'a');
            if (next[pos * 26 + idx]) pos = next[pos * 26 + idx];
            else { next[pos * 26 + idx] = (uint16_t)nodes; pos = nodes; nodes++; }
        }
        sym[pos] = code++;
    }
}


// Per-handle resources that do not depend on the stream's content are recycled: creating a handle
// costs a pinned allocation, a stream and ~70 events otherwise, and an archive job creates one handle
// per task (hundreds).  HostRes = pinned staging slab + HIP stream + events, keyed by (device, size);
// DevSlab = the ONE device allocation all of a handle's HBM state is carved from, keyed by
// (device, size) and zero-filled again on every reuse (the reference's determinism condition).
struct HostRes {
    int device;
    size_t hsize;
    uint8_t *hslab;
    uint8_t *h_in;              // lazily allocated: only CSCEnc_Encode / CSCMI_EncodeHostChunk stage input on the host
    hipStream_t stream;
    hipEvent_t ev[32][2];       // created on first use (host_events): a handle that is only ever driven through a batch launch records none
    hipEvent_t ev_an[2];
    bool have_events;
};
struct DevSlab { int device; size_t size; void *p; };

std::mutex g_cache_mu;
std::vector<HostRes *> g_host_cache;
std::vector<DevSlab> g_dev_cache;
size_t g_dev_cache_bytes = 0;
constexpr size_t kHostCacheMax = 4096;            // entries (~90 KiB pinned each; coder blocks are read back through the calling thread's buffer, thread_pinned)
constexpr size_t kDevCacheMaxBytes = 48ull << 30;

}  // namespace

// Streams are shared: creating one costs ~4 ms on this runtime (tools/gpu_setup_probe.py), an archive job has thousands of handles
// alive at once, and a handle's own stream only carries its set-up, its single-handle calls and its EOF -- batch calls run on the
// lead handle's stream.  Up to kStreamPool non-blocking streams per device, handed out least-used first, created when every
// existing one has a user, never destroyed.  Handles that share a stream stay correct (every call waits for its own work; it may
// wait for a neighbour's too): a host that drives more than kStreamPool handles from threads of its own loses overlap, not bytes.
namespace cscmi {
namespace {
constexpr int kStreamPool = 16;
struct StreamPool { hipStream_t s[kStreamPool]; uint32_t users[kStreamPool]; int n; };
StreamPool g_streams[64];
std::mutex g_streams_mu;
}
hipStream_t pooled_stream(int device)
{
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    StreamPool &p = g_streams[device];
    int best = -1;
    for (int i = 0; i < p.n; i++) if (best < 0 || p.users[i] < p.users[best]) best = i;
    if ((best < 0 || p.users[best] > 0) && p.n < kStreamPool) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) { best = p.n++; p.s[best] = st; p.users[best] = 0; }
        else (void)hipGetLastError();
    }
    if (best < 0) return nullptr;
    p.users[best]++;
    return p.s[best];
}
void pooled_stream_release(int device, hipStream_t s)
{
    if (device < 0 || device >= 64 || !s) return;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    StreamPool &p = g_streams[device];
    for (int i = 0; i < p.n; i++) if (p.s[i] == s) { if (p.users[i]) p.users[i]--; return; }
}
}  // namespace cscmi

namespace {

void host_res_destroy(HostRes *r)
{
    if (!r) return;
    if (r->hslab) (void)hipHostFree(r->hslab);
    if (r->h_in) (void)hipHostFree(r->h_in);
    for (auto &pr : r->ev) for (auto &e : pr) if (e) (void)hipEventDestroy(e);
    for (auto &e : r->ev_an) if (e) (void)hipEventDestroy(e);
    if (r->stream) pooled_stream_release(r->device, r->stream);
    delete r;
}

HostRes *host_res_get(int device, size_t hsize)
{
    HostRes *r = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_host_cache.size(); i++)
            if (g_host_cache[i]->device == device && g_host_cache[i]->hsize == hsize) {
                r = g_host_cache[i];
                g_host_cache.erase(g_host_cache.begin() + i);
                break;
            }
    }
    bool ok = true;
    if (!r) {
        r = new HostRes();
        memset(r, 0, sizeof(*r));
        r->device = device; r->hsize = hsize;
        ok = hipHostMalloc((void **)&r->hslab, hsize, hipHostMallocDefault) == hipSuccess;
    }
    r->stream = ok ? pooled_stream(device) : nullptr;      // (a recycled entry holds none while it sits in the cache)
    if (!ok || !r->stream) { host_res_destroy(r); return nullptr; }
    return r;
}

hipError_t host_events(HostRes *r)
{
    if (r->have_events) return hipSuccess;
    for (auto &pr : r->ev) for (auto &e : pr) if (!e) { hipError_t err = hipEventCreate(&e); if (err != hipSuccess) return err; }
    for (auto &e : r->ev_an) if (!e) { hipError_t err = hipEventCreate(&e); if (err != hipSuccess) return err; }
    r->have_events = true;
    return hipSuccess;
}

void host_res_put(HostRes *r)
{
    if (!r) return;
    pooled_stream_release(r->device, r->stream);
    r->stream = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (g_host_cache.size() < kHostCacheMax) { g_host_cache.push_back(r); return; }
    }
    host_res_destroy(r);
}

void dev_cache_trim()
{
    std::vector<DevSlab> drop;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        drop.swap(g_dev_cache);
        g_dev_cache_bytes = 0;
    }
    for (DevSlab &d : drop) (void)hipFree(d.p);
}

void *dev_slab_get(int device, size_t size)
{
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_dev_cache.size(); i++)
            if (g_dev_cache[i].device == device && g_dev_cache[i].size == size) {
                void *p = g_dev_cache[i].p;
                g_dev_cache.erase(g_dev_cache.begin() + i);
                g_dev_cache_bytes -= size;
                return p;
            }
    }
    void *p = nullptr;
    if (hipMalloc(&p, size) == hipSuccess) return p;
    (void)hipGetLastError();
    dev_cache_trim();                              // give the cached slabs back and try once more
    if (hipMalloc(&p, size) == hipSuccess) return p;
    return nullptr;
}

void dev_slab_put(int device, size_t size, void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (g_dev_cache_bytes + size <= kDevCacheMaxBytes) {
            g_dev_cache.push_back(DevSlab{device, size, p});
            g_dev_cache_bytes += size;
            return;
        }
    }
    (void)hipFree(p);
}

// per device, once: the constant tables of the kernels + the two tables built on the host -- the word trie of the dictionary filter
// (next: u16[7800], sym: u8[300]) and log2(diffNum-2)-0.6 for diffNum 6..15 (csc_analyzer.cpp:223).  Never freed.
struct DevTables { uint8_t *trie; double *coef; bool ready; };
DevTables g_dev_tables[64];
std::mutex g_dev_tables_mu;
hipError_t device_tables(int device, DevTables &out)
{
    if (device < 0 || device >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_dev_tables_mu);
    DevTables &t = g_dev_tables[device];
    if (!t.ready) {
        hipError_t err = upload_tables();
        if (err != hipSuccess) return err;
        std::vector<uint8_t> trie(300 * 26 * 2 + 320, 0);
        build_trie((uint16_t *)trie.data(), trie.data() + 300 * 26 * 2);
        double coef[16] = {0};
        for (int d = 6; d < 16; d++) coef[d - 6] = log((double)d - 2) / log((double)2) - 0.6;
        uint8_t *p = nullptr;
        const size_t trie_bytes = (trie.size() + 255) & ~(size_t)255;
        if ((err = hipMalloc((void **)&p, trie_bytes + sizeof(coef))) != hipSuccess) return err;
        if ((err = hipMemcpy(p, trie.data(), trie.size(), hipMemcpyHostToDevice)) != hipSuccess
            || (err = hipMemcpy(p + trie_bytes, coef, sizeof(coef), hipMemcpyHostToDevice)) != hipSuccess) { (void)hipFree(p); return err; }
        t.trie = p; t.coef = (double *)(p + trie_bytes); t.ready = true;
    }
    out = t;
    return hipSuccess;
}

struct EncInstance {
    uint32_t magic;
    ISzAlloc *alloc;
    ISeqOutStream *os;
    CSCProps props;
    int device;
    hipStream_t stream;
    EncState *d_state;
    EncState h;                 // host mirror of the configuration + device pointers
    RunDesc *d_runs;
    double *d_entcoef;
    uint8_t *d_trie;            // next (u16[7800]) + sym (u8[300])
    void *dslab;                // everything above and in `h` is carved from this one allocation
    size_t dsize;
    HostRes *res;               // stream, events and the pinned slab the pointers below point into
    // pinned staging
    uint8_t *h_in;
    BlockInfo *h_binfo;
    RunDesc *h_runs;
    uint32_t *h_dup;
    uint32_t *h_small;
    hipEvent_t (*ev)[2];        // res->ev
    hipEvent_t *ev_an;          // res->ev_an
    // accounting
    int64_t outsize;            // GetCompressedSize, csc_encoder_main.cpp:174
    CSCMIStats stats;
    int parser;
    // a chunk whose final launch is deferred to a batch launch (CSCMI_EncodeDeviceChunkBatch)
    uint32_t pend_a, pend_b;
    bool pend_first;
    int pend_ev;
    // the run segmentation of the chunk in hand, resumable (seg_advance): it stops where LZ::IsDuplicateBlock must look at the tables
    // as the runs so far leave them (csc_encoder_main.cpp:123-126), so that a batch of streams can take that step together
    struct Seg {
        uint32_t nruns, launched, last_type, last_begin, last_size, bpb, dup_from, dup_to, blk, i, need_blk, need_cnt;
        bool first_launch;
        int ev_used;
    } seg;
};
// Per-thread resources, recycled through process-wide free lists: a thread that ends hands them back (its thread_local holder's
// destructor makes no HIP call -- it may run while the runtime shuts down), the next thread that needs one takes it over.  So
// short-lived worker threads do not accumulate pinned memory, streams or device tables.
//  * PinBuf: the pinned buffer coder blocks are read back through -- one handle's (drain_arena: a chunk's blocks, or the few bytes
//    of an EOF) and a batch's (drain_batch).  Grown on demand to what is actually read (power of two, >= 256 KiB), so a
//    CSCEnc_Encode_Flush pins no arena-sized buffer and a handle pins none at all.
//  * BatchArgs: argument arrays of the multi-stream launch ([4 * kMaxBatch] pointer-sized words: states, run lists, run counts,
//    reset flags), one set per calling thread and device, + the {bytes, error} pair of every stream of a batch.
struct PinBuf { uint8_t *p = nullptr; size_t cap = 0; };
struct BatchArgs {
    int device = -1; void **d = nullptr; void **h = nullptr; void **d2 = nullptr; void **h2 = nullptr; hipStream_t side = nullptr;
    uint32_t *small = nullptr;
};
std::mutex g_thread_res_mu;
std::vector<PinBuf> g_pin_free;
std::vector<BatchArgs> g_batch_free;
struct ThreadRes {
    PinBuf pin;
    BatchArgs batch;
    ~ThreadRes()
    {
        std::lock_guard<std::mutex> lk(g_thread_res_mu);
        if (pin.p) g_pin_free.push_back(pin);
        if (batch.d) g_batch_free.push_back(batch);
    }
};
thread_local ThreadRes t_res;
#define t_batch (t_res.batch)
constexpr int kMaxBatch = 2048;
constexpr size_t kBatchPool = 64ull << 20;       // most a batch reads back per wait; >= the arena of the largest raw_blocksize (3 x 16 MiB + 1 MiB)
constexpr size_t kPinMin = 256 * 1024;

// the calling thread's pinned read-back buffer, at least `need` bytes (nullptr: allocation failed)
uint8_t *thread_pinned(size_t need)
{
    PinBuf &b = t_res.pin;
    if (b.cap >= need) return b.p;
    PinBuf take;
    {
        std::lock_guard<std::mutex> lk(g_thread_res_mu);
        if (b.p) { g_pin_free.push_back(b); b = PinBuf(); }
        size_t best = g_pin_free.size();
        for (size_t i = 0; i < g_pin_free.size(); i++)
            if (g_pin_free[i].cap >= need && (best == g_pin_free.size() || g_pin_free[i].cap < g_pin_free[best].cap)) best = i;
        if (best < g_pin_free.size()) { take = g_pin_free[best]; g_pin_free.erase(g_pin_free.begin() + best); }
        else if (g_pin_free.size() > 8) {        // nothing fits and the list is long: give the smallest back to the system
            size_t sm = 0;
            for (size_t i = 1; i < g_pin_free.size(); i++) if (g_pin_free[i].cap < g_pin_free[sm].cap) sm = i;
            (void)hipHostFree(g_pin_free[sm].p);
            g_pin_free.erase(g_pin_free.begin() + sm);
        }
    }
    if (!take.p) {
        size_t cap = kPinMin;
        while (cap < need) cap <<= 1;
        if (hipHostMalloc((void **)&take.p, cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        take.cap = cap;
    }
    b = take;
    return b.p;
}

void free_device(EncInstance *e)
{
    (void)hipSetDevice(e->device);
    if (e->res) (void)hipStreamSynchronize(e->res->stream);
    dev_slab_put(e->device, e->dsize, e->dslab);
    host_res_put(e->res);
    e->dslab = nullptr; e->res = nullptr; e->stream = nullptr;
}

// MemIO::WriteBlock, csc_memio.cpp:83-108: flag byte, [3-byte BE size], payload -- 2-3 Write calls
int write_block(EncInstance *e, const uint8_t *buf, uint32_t size, uint32_t rc1bc0)
{
    uint8_t fb = (uint8_t)(rc1bc0 << 7);
    if (size == e->props.csc_blocksize) fb |= (1 << 6);
    if (e->os->Write(e->os, &fb, 1) != 1) return -1;
    if (size != e->props.csc_blocksize) {
        uint8_t sb[3] = {(uint8_t)(size >> 16), (uint8_t)(size >> 8), (uint8_t)size};
        if (e->os->Write(e->os, sb, 3) != 3) return -1;
    }
    if (size && e->os->Write(e->os, buf, size) != size) return -1;
    return 0;
}

// launch [a, b) of the chunk's run list
int launch_runs(EncInstance *e, uint32_t a, uint32_t b, bool &first_launch, int &ev_used)
{
    if (a == b) return 0;
    HIPCHK(hipMemcpyAsync(e->d_runs + a, e->h_runs + a, sizeof(RunDesc) * (b - a), hipMemcpyHostToDevice, e->stream));
    bool timed = ev_used < kEventPairs;
    if (timed) HIPCHK(hipEventRecord(e->ev[ev_used][0], e->stream));
    launch_encode_runs(e->parser, e->d_state, e->d_runs + a, b - a, first_launch ? 1u : 0u, e->stream);
    HIPCHK(hipGetLastError());
    if (timed) { HIPCHK(hipEventRecord(e->ev[ev_used][1], e->stream)); ev_used++; }
    e->stats.encode_launches++;
    first_launch = false;
    return 0;
}

// hand the coder blocks of one chunk (a host copy of the stream's arena) to the user's stream, in the order they were finished
int write_arena(EncInstance *e, const uint8_t *h_arena, uint32_t used)
{
    for (uint32_t off = 0; off < used;) {
        const ArenaRec *rec = (const ArenaRec *)(h_arena + off);
        e->outsize += rec->size;
        e->stats.output_bytes += rec->size;
        if (write_block(e, h_arena + off + 16, rec->size, rec->kind) < 0) return WRITE_ERROR;
        off += 16 + ((rec->size + 15) & ~15u);
    }
    return 0;
}

int report_device_error(uint32_t err)
{
    // anything else is the watchdog of the multi-wavefront parser (csc_kernels_dp2.inc): the value says which wait gave up
    fprintf(stderr, "csc-mi355x: device encoder error 0x%x (%s)\n", err,
            err == ERR_ARENA_FULL ? "output arena exhausted" : err == ERR_BAD_TYPE ? "bad block type" : "parse wavefronts lost step");
    return CSCMI_DEVICE_ERROR;
}

// read the finished coder blocks of this chunk back and hand them to the user's stream
int drain_arena(EncInstance *e, int ev_used)
{
    HIPCHK(hipMemcpyAsync(e->h_small, &e->d_state->arena_used, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < ev_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->ev[i][0], e->ev[i][1]) == hipSuccess) e->stats.encode_kernel_ms += ms;
    }
    uint32_t used = e->h_small[0], err = e->h_small[1];
    if (err != ERR_NONE) return report_device_error(err);
    uint8_t *pin = nullptr;
    if (used) {
        pin = thread_pinned(used);
        if (!pin) { fprintf(stderr, "csc-mi355x: pinned read-back buffer of %u bytes failed\n", used); return CSCMI_DEVICE_ERROR; }
        HIPCHK(hipMemcpyAsync(pin, e->h.arena, used, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return write_arena(e, pin, used);
}

// CSCEncoder::Compress, csc_encoder_main.cpp:85-147, with the data work on the device.
// Three stages so that many handles can be driven through one multi-stream launch:
//   chunk_begin   upload the chunk, launch the analyzer, start the verdict read-back (no wait)
//   chunk_segment wait for the verdicts, build the run list (may launch + wait for duplicate-block
//                 checks), then either launch the remaining runs or leave them pending
//   chunk_finish  read the coder blocks back and call the user's Write
// The chunk's blocks are walked by the encode kernel itself (enc_compress_chunk, csc_kernels_blocks.inc): typing, duplicate-block
// checks and run formation need no host round trip.  CSCMI_HOST_SEGMENT=1 (diagnostics) keeps the walk on the host as rounds
// 1-3 had it: the run list is built here, every IsDuplicateBlock verdict costs a launch boundary.
bool host_segment()
{
    static const bool v = [] { const char *s = getenv("CSCMI_HOST_SEGMENT"); return s && atoi(s) != 0; }();
    return v;
}

int chunk_begin(EncInstance *e, const void *src, size_t size, bool on_device, hipStream_t st = nullptr)
{
    if (size == 0 || size > e->props.raw_blocksize) return -1;
    HIPCHK(hipSetDevice(e->device));
    const bool timed = st == nullptr;                // (a batch queues every stream's analyzer on the lead's stream: not timed one by one)
    if (!st) st = e->stream;
    if (!host_segment()) {
        HIPCHK(hipMemcpyAsync(e->h.inbuf, src, size, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        if ((e->props.DLTFilter + e->props.EXEFilter + e->props.TXTFilter) != 0) {
            if (timed) HIPCHK(hipEventRecord(e->ev_an[0], st));
            launch_analyze(e->d_state, (uint32_t)size, e->d_entcoef, st);
            HIPCHK(hipGetLastError());
            if (timed) HIPCHK(hipEventRecord(e->ev_an[1], st));
        }
        return 0;
    }
    HIPCHK(hipMemcpyAsync(e->h.inbuf, src, size, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
    const uint32_t csize = (uint32_t)size;
    const uint32_t nblk = (csize + kMinBlock - 1) / kMinBlock;
    const bool use_filters = (e->props.DLTFilter + e->props.EXEFilter + e->props.TXTFilter) != 0;   // csc_encoder_main.cpp:27-31
    if (use_filters) {
        HIPCHK(hipEventRecord(e->ev_an[0], e->stream));
        launch_analyze(e->d_state, csize, e->d_entcoef, e->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e->ev_an[1], e->stream));
        HIPCHK(hipMemcpyAsync(e->h_binfo, e->h.binfo, sizeof(BlockInfo) * nblk, hipMemcpyDeviceToHost, e->stream));
    }
    return 0;
}

// start the segmentation of the chunk chunk_begin has uploaded: waits for the analyzer's verdicts
int seg_begin(EncInstance *e)
{
    const bool use_filters = (e->props.DLTFilter + e->props.EXEFilter + e->props.TXTFilter) != 0;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (use_filters) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->ev_an[0], e->ev_an[1]) == hipSuccess) e->stats.analyze_kernel_ms += ms;
    }
    EncInstance::Seg &g = e->seg;
    g = EncInstance::Seg();
    g.first_launch = true;
    g.last_type = DT_NORMAL;
    return 0;
}

// Walk the chunk's blocks on (CSCEncoder::Compress, csc_encoder_main.cpp:85-147).  Returns 1 when block g.need_blk (and the
// g.need_cnt - 1 blocks behind it that may need the same) must be tested by IsDuplicateBlock against the tables as the runs
// [g.launched, g.nruns) leave them: the caller launches those runs, then k_dup_check, reads the flags back (h_dup), sets
// g.launched = g.nruns, g.dup_from / g.dup_to, and calls again.  Returns 0 when the run list is complete.
int seg_advance(EncInstance *e, size_t size)
{
    EncInstance::Seg &g = e->seg;
    const uint32_t csize = (uint32_t)size;
    const uint32_t nblk = (csize + kMinBlock - 1) / kMinBlock;
    const bool use_filters = (e->props.DLTFilter + e->props.EXEFilter + e->props.TXTFilter) != 0;
    auto close_run = [&](uint32_t tail) {
        RunDesc &r = e->h_runs[g.nruns++];
        r.type = g.last_type; r.offset = g.last_begin; r.size = g.last_size; r.tail = tail;
    };
    while (g.i < csize) {
        const uint32_t blk = g.blk;
        uint32_t cur = csize - g.i < kMinBlock ? csize - g.i : kMinBlock;
        uint32_t this_type = DT_NORMAL;
        const BlockInfo *bi = use_filters ? &e->h_binfo[blk] : nullptr;
        uint32_t bpb = g.bpb;
        if (use_filters) {
            this_type = bi->type;
            if (this_type != DT_SKIP) bpb = bi->bpb;
        }
        if (this_type == DT_SKIP) this_type = g.last_type;
        if (this_type != DT_NORMAL) {
            if (this_type == DT_EXE && e->props.EXEFilter == 0) this_type = DT_NORMAL;
            else if (this_type == DT_ENGTXT && e->props.TXTFilter == 0) this_type = DT_NORMAL;
            else if (this_type >= DT_DLT && e->props.DLTFilter == 0) this_type = DT_NORMAL;
        }
        if (this_type >= DT_DLT && (double)bi->dlt_bpb[this_type - DT_DLT] >= bpb * 0.95)   // :117-121
            this_type = DT_NORMAL;
        if (this_type >= DT_NO_LZ) {   // :123-126 LZ::IsDuplicateBlock against the tables as of the pending run
            if (!(g.launched == g.nruns && blk >= g.dup_from && blk < g.dup_to)) {
                uint32_t cnt = 1;   // test the whole stretch of blocks that may need it under this state
                while (blk + cnt < nblk && (e->h_binfo[blk + cnt].type >= DT_NO_LZ)) cnt++;
                g.need_blk = blk; g.need_cnt = cnt;
                return 1;           // (nothing of this block has been committed: the walk resumes at it)
            }
            if (e->h_dup[blk]) this_type = DT_NORMAL;
        }
        g.bpb = bpb;
        if (g.last_type != this_type || g.last_size + cur > e->props.raw_blocksize) {
            if (g.last_size) close_run(0);
            g.last_begin = g.i;
            g.last_size = 0;
        }
        g.last_type = this_type;
        g.last_size += cur;
        g.i += cur;
        g.blk++;
    }
    if (g.last_size) { close_run(1); g.last_size = 0; }
    return 0;
}

int chunk_segment(EncInstance *e, size_t size, bool defer_final)
{
    int rc = seg_begin(e);
    if (rc) return rc;
    EncInstance::Seg &g = e->seg;
    const uint32_t csize = (uint32_t)size;
    while ((rc = seg_advance(e, size)) == 1) {
        rc = launch_runs(e, g.launched, g.nruns, g.first_launch, g.ev_used);
        if (rc) return rc;
        g.launched = g.nruns;
        launch_dup_check(e->d_state, csize, g.need_blk, g.need_cnt, e->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(e->h_dup + g.need_blk, e->h.dup_flags + g.need_blk, sizeof(uint32_t) * g.need_cnt, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        g.dup_from = g.need_blk; g.dup_to = g.need_blk + g.need_cnt;
    }
    if (rc < 0) return rc;
    e->stats.chunks++;
    e->stats.input_bytes += size;
    if (defer_final) {
        e->pend_a = g.launched; e->pend_b = g.nruns; e->pend_first = g.first_launch; e->pend_ev = g.ev_used;
        return 0;
    }
    rc = launch_runs(e, g.launched, g.nruns, g.first_launch, g.ev_used);
    e->pend_ev = g.ev_used;
    return rc;
}

int encode_chunk(EncInstance *e, const void *src, size_t size, bool on_device)
{
    if (size == 0) return 0;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(host_events(e->res));
    int rc = chunk_begin(e, src, size, on_device);
    if (rc) return rc;
    if (!host_segment()) {
        HIPCHK(hipEventRecord(e->ev[0][0], e->stream));
        launch_encode_runs(e->parser, e->d_state, e->d_runs, kSelfSegment | (uint32_t)size, 1u, e->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e->ev[0][1], e->stream));
        e->stats.encode_launches++;
        e->stats.chunks++;
        e->stats.input_bytes += size;
        rc = drain_arena(e, 1);
        float ms = 0;
        if ((e->props.DLTFilter + e->props.EXEFilter + e->props.TXTFilter) != 0 && hipEventElapsedTime(&ms, e->ev_an[0], e->ev_an[1]) == hipSuccess)
            e->stats.analyze_kernel_ms += ms;
        return rc;
    }
    rc = chunk_segment(e, size, false);
    if (rc) return rc;
    return drain_arena(e, e->pend_ev);
}

// the calling thread's launch tables + read-back pool, on `device`
int batch_args_ready(int device)
{
    if (t_batch.device == device) return 0;
    {
        std::lock_guard<std::mutex> lk(g_thread_res_mu);
        if (t_batch.d) { g_batch_free.push_back(t_batch); t_batch = BatchArgs(); }
        for (size_t i = 0; i < g_batch_free.size(); i++)
            if (g_batch_free[i].device == device) { t_batch = g_batch_free[i]; g_batch_free.erase(g_batch_free.begin() + i); return 0; }
    }
    BatchArgs b;
    auto fail = [&](hipError_t err) {
        fprintf(stderr, "csc-mi355x: batch tables: %s\n", hipGetErrorString(err));
        if (b.d) (void)hipFree(b.d);
        if (b.h) (void)hipHostFree(b.h);
        if (b.d2) (void)hipFree(b.d2);
        if (b.h2) (void)hipHostFree(b.h2);
        if (b.side) (void)hipStreamDestroy(b.side);
        if (b.small) (void)hipHostFree(b.small);
        return CSCMI_DEVICE_ERROR;
    };
    hipError_t err;
    if ((err = hipMalloc((void **)&b.d, sizeof(void *) * 4 * kMaxBatch)) != hipSuccess) return fail(err);
    if ((err = hipHostMalloc((void **)&b.h, sizeof(void *) * 4 * kMaxBatch, hipHostMallocDefault)) != hipSuccess) return fail(err);
    if ((err = hipMalloc((void **)&b.d2, sizeof(void *) * 4 * kMaxBatch)) != hipSuccess) return fail(err);
    if ((err = hipHostMalloc((void **)&b.h2, sizeof(void *) * 4 * kMaxBatch, hipHostMallocDefault)) != hipSuccess) return fail(err);
    if ((err = hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking)) != hipSuccess) return fail(err);
    if ((err = hipHostMalloc((void **)&b.small, sizeof(uint32_t) * 2 * kMaxBatch, hipHostMallocDefault)) != hipSuccess) return fail(err);
    b.device = device;
    t_batch = b;
    return 0;
}

// Read the coder blocks of hs[i] (every i with skip == nullptr || skip[i] != 0) back and hand them to the users' streams, in handle
// order: the {bytes, error} pairs of all streams with one wait, then the arenas packed into the thread's pinned pool, one wait per
// pool-full.  All copies on `st`, the stream the kernels that filled the arenas were launched on.
int drain_batch(int n, CSCEncHandle *hs, const size_t *skip, hipStream_t st)
{
    uint32_t *sm = t_batch.small;
    for (int i = 0; i < n; i++) {
        if (skip && !skip[i]) continue;
        EncInstance *e = (EncInstance *)hs[i];
        HIPCHK(hipMemcpyAsync(sm + 2 * i, &e->d_state->arena_used, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    int rc = 0;
    for (int i = 0; i < n; i++) {
        if (skip && !skip[i]) continue;
        if (sm[2 * i + 1] != ERR_NONE) rc = report_device_error(sm[2 * i + 1]);
        else if ((size_t)sm[2 * i] > kBatchPool) { fprintf(stderr, "csc-mi355x: a stream's coder blocks exceed the read-back pool\n"); rc = CSCMI_DEVICE_ERROR; }
    }
    if (rc) return rc;
    // the pool: what this batch reads back (a round's worth at most), not a fixed 64 MiB -- a CSCMI_FlushBatch needs a few KiB
    size_t want = 0, biggest = 0;
    for (int i = 0; i < n; i++) {
        if (skip && !skip[i]) continue;
        const size_t used = sm[2 * i];
        want += (used + 255) & ~(size_t)255;
        if (used > biggest) biggest = used;
    }
    if (want > kBatchPool) want = kBatchPool;
    if (want < biggest) want = biggest;
    if (!want) return 0;
    uint8_t *pool = thread_pinned(want);
    if (!pool) { fprintf(stderr, "csc-mi355x: pinned read-back pool of %zu bytes failed\n", want); return CSCMI_DEVICE_ERROR; }
    const size_t pool_cap = t_res.pin.cap;
    std::vector<size_t> at((size_t)n, 0);
    for (int i = 0; i < n;) {
        size_t off = 0;
        int j = i;
        for (; j < n; j++) {
            if (skip && !skip[j]) continue;
            const size_t used = sm[2 * j];
            if (off + used > pool_cap) break;
            at[j] = off;
            if (used) HIPCHK(hipMemcpyAsync(pool + off, ((EncInstance *)hs[j])->h.arena, used, hipMemcpyDeviceToHost, st));
            off += (used + 255) & ~(size_t)255;
        }
        HIPCHK(hipStreamSynchronize(st));
        for (int k = i; k < j; k++) {
            if (skip && !skip[k]) continue;
            int r = write_arena((EncInstance *)hs[k], pool + at[k], sm[2 * k]);
            if (r && !rc) rc = r;
        }
        i = j;
    }
    return rc;
}

}  // namespace

// =============================================================================================
extern "C" {

int CSCMI_DeviceCheck(void)
{
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0) {
        fprintf(stderr, "csc-mi355x: no HIP device visible (%s); this library has no CPU fallback\n",
                err == hipSuccess ? "device count 0" : hipGetErrorString(err));
        return CSCMI_DEVICE_ERROR;
    }
    return 0;
}

// CSCEncProps_Init, csc_enc.cpp:16-97 (pure host arithmetic; the 10 KiB head-room, the hash-bit
// classes and the per-level overrides are part of the format's de-facto contract)
void CSCEncProps_Init(CSCProps *p, uint32_t dict_size, int level)
{
    dict_size += 10 * kKB;
    if (dict_size < 32 * kKB) dict_size = 32 * kKB;
    if (dict_size > 1024 * kMB) dict_size = 1024 * kMB;
    p->dict_size = dict_size;
    level = level < 1 ? 1 : (level > 5 ? 5 : level);
    p->DLTFilter = p->TXTFilter = p->EXEFilter = 1;
    p->csc_blocksize = 64 * kKB;
    p->raw_blocksize = 2 * kMB;
    uint32_t hbits = dict_size < kMB ? 19 : dict_size <= 4 * kMB ? 20 : dict_size <= 16 * kMB ? 21
                   : dict_size <= 64 * kMB ? 22 : dict_size <= 256 * kMB ? 23 : 24;
    while (((uint32_t)1 << hbits) > dict_size) hbits--;
    if (dict_size <= 16 * kMB) p->bt_size = dict_size;
    else if (dict_size <= 64 * kMB) p->bt_size = (dict_size - 16 * kMB) / 2 + 16 * kMB;
    else if (dict_size <= 256 * kMB) p->bt_size = (dict_size - 64 * kMB) / 4 + 40 * kMB;
    else p->bt_size = (dict_size - 256 * kMB) / 8 + 88 * kMB;
    p->good_len = 32;
    p->hash_bits = (uint8_t)hbits;
    p->bt_hash_bits = (uint8_t)(hbits + 1);
    struct { uint8_t width, mode, good; int dbits; } lv[5] = {
        {1, 2, 32, +1}, {8, 2, 24, -1}, {2, 3, 16, +1}, {8, 3, 24, -1}, {0, 3, 48, 0}};
    const auto &L = lv[level - 1];
    p->hash_width = L.width; p->lz_mode = L.mode; p->good_len = L.good;
    p->hash_bits = (uint8_t)(p->hash_bits + L.dbits);
    if (level < 5) p->bt_size = 0;          // levels 1-4 leave bt_cyc as the caller had it (:57-84)
    else p->bt_cyc = 32;
    if (p->bt_size == p->dict_size) p->hash_width = 0;
}

void CSCEnc_WriteProperties(const CSCProps *props, uint8_t *s, int full)   // csc_enc.cpp:145-158
{
    (void)full;
    uint32_t d = (uint32_t)props->dict_size;
    s[0] = (uint8_t)(d >> 24); s[1] = (uint8_t)(d >> 16); s[2] = (uint8_t)(d >> 8); s[3] = (uint8_t)d;
    s[4] = (uint8_t)(props->csc_blocksize >> 16); s[5] = (uint8_t)(props->csc_blocksize >> 8); s[6] = (uint8_t)props->csc_blocksize;
    s[7] = (uint8_t)(props->raw_blocksize >> 16); s[8] = (uint8_t)(props->raw_blocksize >> 8); s[9] = (uint8_t)props->raw_blocksize;
}

uint64_t CSCEnc_EstMemUsage(const CSCProps *p)   // csc_enc.cpp:99-112 (same 32-bit intermediate arithmetic)
{
    uint64_t ret = p->dict_size;
    ret += p->csc_blocksize * 2;
    if (p->bt_size) ret += (uint64_t)(uint32_t)((1u << p->bt_hash_bits) + 2 * p->bt_size) * sizeof(uint32_t);
    if (p->hash_width) ret += (uint64_t)(int64_t)(int)(p->hash_width * (1 << p->hash_bits)) * sizeof(uint32_t);
    ret += 80 * kKB * sizeof(uint32_t);
    ret += 256 * 256 * sizeof(uint32_t) * 2;
    ret += 2 * kMB;
    return ret;
}

CSCEncHandle CSCEnc_Create(const CSCProps *props, ISeqOutStream *outstream, ISzAlloc *alloc)   // csc_enc.cpp:114-133
{
    if (alloc == NULL) alloc = &g_default_alloc;
    if (CSCMI_DeviceCheck() != 0) return NULL;
    if (props->csc_blocksize == 0 || props->csc_blocksize >= 16 * kMB || props->raw_blocksize == 0
        || props->raw_blocksize >= 16 * kMB || props->hash_width > 32
        || (props->lz_mode != 1 && props->lz_mode != 2 && props->lz_mode != 3)
        || (props->hash_width && (props->hash_bits < 1 || props->hash_bits > 28))
        || (props->bt_size && props->bt_hash_bits && (props->bt_hash_bits > 28))) {
        fprintf(stderr, "csc-mi355x: unsupported CSCProps\n");
        return NULL;
    }

    EncInstance *e = (EncInstance *)alloc->Alloc(alloc, sizeof(EncInstance));
    if (!e) return NULL;
    memset(e, 0, sizeof(*e));
    e->magic = kMagicEnc; e->alloc = alloc; e->os = outstream; e->props = *props;
    e->parser = props->lz_mode == 3 ? 3 : 2;
    {   // hash-table-only configuration (levels 1-4): the specialised kernels (bit 4 of the launch kind)
        bool bt = props->bt_hash_bits && props->bt_size;
        bool ht = props->hash_bits && props->hash_width;
        if (!bt && ht && props->hash_width <= 9) e->parser |= 4;
        // level-3 geometry: the pipeline form of the advanced parser (csc_kernels_dp4.inc)
        if ((e->parser & 4) && props->lz_mode == 3 && props->hash_width <= 2 && props->good_len >= 2 && props->good_len <= 16) e->parser |= 8;
        // level-5 geometry -- binary tree, no bucket, advanced parser: the inserter form (csc_kernels_bt.inc)
        if (bt && !ht && props->lz_mode == 3 && props->bt_cyc <= 32) e->parser |= 16;
        // lazy / greedy parser over a bucket of up to eight (levels 1, 2): the inserter form (csc_kernels_hp.inc)
        if ((e->parser & 4) && props->lz_mode != 3 && props->hash_width <= 8) e->parser |= 32;
    }
    bool ok = hipGetDevice(&e->device) == hipSuccess;
    // each process/thread may sit on another device: the constant tables are per device
    DevTables dt = {nullptr, nullptr, false};
    if (ok) ok = device_tables(e->device, dt) == hipSuccess;

    EncState &h = e->h;
    // LZ::Init / MatchFinder::Init geometry, csc_lz.cpp:15-33, csc_mf.cpp:45-106
    uint32_t wnd = (uint32_t)(props->dict_size > 0xFFFFFFFFull ? 0xFFFFFFFFull : props->dict_size);
    if (wnd < 32 * kKB) wnd = 32 * kKB;
    if (wnd > 1024 * kMB) wnd = 1024 * kMB;
    h.wnd_size = wnd;
    h.vld_rge = wnd - kMinBlock - 4;
    h.bsize = props->csc_blocksize; h.raw_blocksize = props->raw_blocksize;
    h.ht_bits = props->hash_bits; h.ht_width = props->hash_width;
    h.bt_bits = props->bt_hash_bits; h.bt_size = props->bt_size;
    if (!h.bt_bits || !h.bt_size) h.bt_bits = h.bt_size = 0;
    if (!h.ht_bits || !h.ht_width) h.ht_bits = h.ht_width = 0;
    h.lz_mode = props->lz_mode; h.lz_good_len = props->good_len;
    h.lz_bt_cyc = props->bt_cyc; h.lz_ht_cyc = props->hash_width;
    h.mf_size = (uint64_t)kHT2Size + kHT3Size + ((uint64_t)h.ht_width << h.ht_bits);
    if (h.bt_bits) h.mf_size += ((uint64_t)1 << h.bt_bits) + (uint64_t)h.bt_size * 2;
    h.arena_cap = 3 * props->raw_blocksize + kMB;
    h.filt_flags = (props->DLTFilter ? 1u : 0u) | (props->TXTFilter ? 2u : 0u) | (props->EXEFilter ? 4u : 0u);

    // ---- one device slab, zero-filled: memset(wnd_, 0, ..) csc_lz.cpp:50, the tables csc_mf.cpp:73, and the
    // coder buffers, which the reference reads before writing (App. C #1)
    size_t doff = 0;
    auto dtake = [&](size_t bytes) { size_t o = doff; doff += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_wnd = dtake((size_t)wnd + 256), o_mf = dtake((h.mf_size + 64) * sizeof(uint32_t));
    const size_t o_plit = dtake(256 * 256 * sizeof(uint32_t)), o_pdelta = dtake(256 * 256 * sizeof(uint32_t));
    const size_t o_rc = dtake((size_t)h.bsize + 64), o_bc = dtake((size_t)h.bsize + 64);
    const size_t o_in = dtake((size_t)h.raw_blocksize + 256), o_swap = dtake(4 * (size_t)h.raw_blocksize + 512);
    const size_t o_arena = dtake((size_t)h.arena_cap + 64), o_binfo = dtake(sizeof(BlockInfo) * kMaxBlocksPerChunk);
    const size_t o_dup = dtake(sizeof(uint32_t) * kMaxBlocksPerChunk);
    const size_t o_runs = dtake(sizeof(RunDesc) * (kMaxBlocksPerChunk + 2));
    const size_t o_undo = dtake(h.bt_bits ? kBtUndoBytes : 0);          // (its own region: the filter scratch can be smaller than the log when raw_blocksize is)
    const size_t o_state = dtake(sizeof(EncState));
    e->dsize = doff;
    // ---- one pinned slab
    size_t hoff = 0;
    auto htake = [&](size_t bytes) { size_t o = hoff; hoff += (bytes + 255) & ~(size_t)255; return o; };
    const size_t p_binfo = htake(sizeof(BlockInfo) * kMaxBlocksPerChunk);
    const size_t p_runs = htake(sizeof(RunDesc) * (kMaxBlocksPerChunk + 2)), p_dup = htake(sizeof(uint32_t) * kMaxBlocksPerChunk);
    const size_t p_small = htake(64);

    if (ok) { e->res = host_res_get(e->device, hoff); ok = e->res != nullptr; }
    if (ok) { e->dslab = dev_slab_get(e->device, e->dsize); ok = e->dslab != nullptr; }
    if (ok) {
        e->stream = e->res->stream; e->ev = e->res->ev; e->ev_an = e->res->ev_an;
        uint8_t *D = (uint8_t *)e->dslab, *H = e->res->hslab;
        h.wnd = D + o_wnd; h.mfbuf = (uint32_t *)(D + o_mf); h.p_lit = (uint32_t *)(D + o_plit); h.p_delta = (uint32_t *)(D + o_pdelta);
        h.rc_buf = D + o_rc; h.bc_buf = D + o_bc; h.inbuf = D + o_in; h.swapbuf = D + o_swap; h.arena = D + o_arena;
        h.bt_undo = h.bt_bits ? D + o_undo : nullptr;
        h.binfo = (BlockInfo *)(D + o_binfo); h.dup_flags = (uint32_t *)(D + o_dup); e->d_trie = dt.trie;
        e->d_runs = (RunDesc *)(D + o_runs); e->d_entcoef = dt.coef; e->d_state = (EncState *)(D + o_state);
        e->h_in = e->res->h_in;
        e->h_binfo = (BlockInfo *)(H + p_binfo); e->h_runs = (RunDesc *)(H + p_runs);
        e->h_dup = (uint32_t *)(H + p_dup); e->h_small = (uint32_t *)(H + p_small);
        ok = hipMemsetAsync(e->dslab, 0, e->dsize, e->stream) == hipSuccess;
    }
    hipStream_t st = e->stream;
    if (!ok) {
        fprintf(stderr, "csc-mi355x: device allocation failed (%s)\n", hipGetErrorString(hipGetLastError()));
        free_device(e);
        alloc->Free(alloc, e);
        return NULL;
    }
    uint64_t cpos = 0;
    h.ht2 = h.mfbuf + cpos; cpos += kHT2Size;
    h.ht3 = h.mfbuf + cpos; cpos += kHT3Size;
    if (h.ht_width) { h.ht6 = h.mfbuf + cpos; cpos += (uint64_t)h.ht_width << h.ht_bits; } else h.ht6 = nullptr;
    if (h.bt_bits) { h.bt_head = h.mfbuf + cpos; cpos += (uint64_t)1 << h.bt_bits; h.bt_nodes = h.mfbuf + cpos; }
    else { h.bt_head = nullptr; h.bt_nodes = nullptr; }
    h.trie_next = (const uint16_t *)e->d_trie;
    h.trie_sym = e->d_trie + 300 * 26 * 2;

    ok = hipMemcpyAsync(e->d_state, &h, sizeof(EncState), hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
        launch_init_state(e->d_state, st);
        ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(st) == hipSuccess;   // (`h` may change once this returns)
    if (!ok) {
        fprintf(stderr, "csc-mi355x: device initialisation failed (%s)\n", hipGetErrorString(hipGetLastError()));
        free_device(e);
        alloc->Free(alloc, e);
        return NULL;
    }
    return (CSCEncHandle)e;
}

void CSCEnc_Destroy(CSCEncHandle p)   // csc_enc.cpp:135-143
{
    EncInstance *e = (EncInstance *)p;
    if (!e || e->magic != kMagicEnc) return;
    free_device(e);
    e->magic = 0;
    ISzAlloc *a = e->alloc;
    a->Free(a, e);
}

int CSCMI_EncodeHostChunk(CSCEncHandle p, const void *host_ptr, size_t size)
{
    return encode_chunk((EncInstance *)p, host_ptr, size, false);
}
int CSCMI_EncodeDeviceChunk(CSCEncHandle p, const void *device_ptr, size_t size)
{
    return encode_chunk((EncInstance *)p, device_ptr, size, true);
}

// Many independent streams (the archiver's -p / per-extension tasks, csarc.cpp:532-557) advanced by one
// chunk each with ONE kernel launch: one workgroup per stream.  hs[i] encodes sizes[i] bytes at
// device_ptrs[i]; the Write callbacks of all handles run on this thread, in handle order.
int CSCMI_EncodeDeviceChunkBatch(int n, CSCEncHandle *hs, const void *const *device_ptrs, const size_t *sizes)
{
    if (n <= 0) return 0;
    if (n > kMaxBatch) return -1;
    EncInstance *lead = (EncInstance *)hs[0];
    HIPCHK(hipSetDevice(lead->device));
    { int r = batch_args_ready(lead->device); if (r) return r; }
    HIPCHK(host_events(lead->res));
    void **const d_batch = t_batch.d, **const h_batch = t_batch.h;
    int rc = 0;
    static const bool trace = getenv("CSCMI_BATCH_TRACE") != nullptr;      // development: one line per round on stderr
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_begin = tnow();
    if (!host_segment()) {
        // every stream walks its chunk itself: upload + analyzer per stream, then ONE launch per kernel flavour, one workgroup a stream
        for (int i = 0; i < n && !rc; i++) {
            if (!sizes[i]) continue;
            EncInstance *e = (EncInstance *)hs[i];
            if (e->stats.chunks == 0 && e->stream != lead->stream) HIPCHK(hipStreamSynchronize(e->stream));   // (its state was initialised on its own stream)
            rc = chunk_begin(e, device_ptrs[i], sizes[i], true, lead->stream);
        }
        if (rc) return rc;
        for (int parser = 2; parser <= 63; parser++) {
            if ((parser & 3) < 2) continue;
            uint32_t m = 0;
            EncState **st = (EncState **)h_batch;
            const RunDesc **rl = (const RunDesc **)(h_batch + kMaxBatch);
            uint32_t *cnt = (uint32_t *)(h_batch + 2 * kMaxBatch);
            uint32_t *rst = (uint32_t *)(h_batch + 3 * kMaxBatch);
            for (int i = 0; i < n; i++) {
                EncInstance *e = (EncInstance *)hs[i];
                if (!sizes[i] || e->parser != parser) continue;
                st[m] = e->d_state; rl[m] = e->d_runs; cnt[m] = kSelfSegment | (uint32_t)sizes[i]; rst[m] = 1u;
                m++;
            }
            if (!m) continue;
            HIPCHK(hipMemcpyAsync(d_batch, h_batch, sizeof(void *) * 4 * kMaxBatch, hipMemcpyHostToDevice, lead->stream));
            HIPCHK(hipEventRecord(lead->ev[0][0], lead->stream));
            launch_encode_runs_multi(parser, m, (EncState *const *)d_batch, (const RunDesc *const *)(d_batch + kMaxBatch),
                                     (const uint32_t *)(d_batch + 2 * kMaxBatch), (const uint32_t *)(d_batch + 3 * kMaxBatch), lead->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(lead->ev[0][1], lead->stream));
            HIPCHK(hipStreamSynchronize(lead->stream));          // (the pointer tables are reused by the next flavour)
            float ms = 0;
            if (hipEventElapsedTime(&ms, lead->ev[0][0], lead->ev[0][1]) == hipSuccess) lead->stats.encode_kernel_ms += ms;
            lead->stats.encode_launches++;
        }
        const auto t_dr = tnow();
        for (int i = 0; i < n; i++) {
            if (!sizes[i]) continue;
            EncInstance *e = (EncInstance *)hs[i];
            e->stats.chunks++;
            e->stats.input_bytes += sizes[i];
        }
        rc = drain_batch(n, hs, sizes, lead->stream);
        if (trace) fprintf(stderr, "batch trace: %d streams, kernels walk their chunks: launches %.1f ms, drain %.1f ms\n", n, tms(t_begin, t_dr), tms(t_dr, tnow()));
        return rc;
    }
    for (int i = 0; i < n; i++) HIPCHK(host_events(((EncInstance *)hs[i])->res));
    for (int i = 0; i < n && !rc; i++) rc = sizes[i] ? chunk_begin((EncInstance *)hs[i], device_ptrs[i], sizes[i], true) : 0;
    const auto t_cb = tnow();
    for (int i = 0; i < n && !rc; i++) rc = sizes[i] ? seg_begin((EncInstance *)hs[i]) : 0;
    if (rc) return rc;
    if (trace) fprintf(stderr, "batch trace: %d streams: chunk_begin %.1f ms, seg_begin %.1f ms\n", n, tms(t_begin, t_cb), tms(t_cb, tnow()));
    double ms_dup = 0;
    // Rounds: every stream's run segmentation goes on until it needs an IsDuplicateBlock verdict (csc_encoder_main.cpp:123-126: the
    // runs so far must have been encoded first) or is complete.  The streams that wait for a verdict have their runs so far
    // launched TOGETHER (one launch per parser flavour, one workgroup per stream), then their duplicate checks, one wait for all;
    // the last round launches what is left of every stream.  (Stream by stream -- as CSCEnc_Encode does it for one handle -- a
    // batch of streams with high-entropy / delta blocks ran one workgroup at a time.)
    std::vector<uint8_t> state(n, 0);                // 0 walking, 1 waits for a verdict, 2 run list complete
    int round = 0;
    uint32_t side_used = 0;
    bool side_any = false;
    for (;;) {
        const auto t_round = std::chrono::steady_clock::now();
        uint32_t tr_streams = 0, tr_wait = 0;
        round++;
        bool any_wait = false, all_done = true;
        for (int i = 0; i < n; i++) {
            if (!sizes[i] || state[i] == 2) continue;
            EncInstance *e = (EncInstance *)hs[i];
            int r = seg_advance(e, sizes[i]);
            if (r < 0) return r;
            state[i] = r == 1 ? 1 : 2;
            any_wait = any_wait || r == 1;
        }
        for (int i = 0; i < n; i++) all_done = all_done && (!sizes[i] || state[i] == 2);
        // this round's launches.  A stream whose run list is COMPLETE needs nothing from the host any more: its remaining runs go out
        // at once on a stream of their own (`side`) and run while the streams that wait for verdicts go through their rounds (the
        // reference's workers do not wait for each other either, csarc.cpp:361-398); the waiting streams' runs so far go out on the
        // lead's stream, which every round waits for.
        for (int pass = 0; pass < 2; pass++) {                    // 0: complete streams -> side, 1: waiting streams -> lead's stream
            for (int parser = 2; parser <= 63; parser++) {
                if ((parser & 3) < 2) continue;
                hipStream_t lst = pass == 0 ? t_batch.side : lead->stream;
                void **hb = pass == 0 ? t_batch.h2 + side_used : h_batch, **db = pass == 0 ? t_batch.d2 + side_used : d_batch;
                uint32_t m = 0;
                EncState **st = (EncState **)hb;
                const RunDesc **rl = (const RunDesc **)(hb + kMaxBatch);
                uint32_t *cnt = (uint32_t *)(hb + 2 * kMaxBatch);
                uint32_t *rst = (uint32_t *)(hb + 3 * kMaxBatch);
                for (int i = 0; i < n; i++) {
                    EncInstance *e = (EncInstance *)hs[i];
                    if (!sizes[i] || e->parser != parser) continue;
                    if (state[i] != (pass == 0 ? 2 : 1)) continue;
                    EncInstance::Seg &g = e->seg;
                    if (g.launched == g.nruns) continue;
                    HIPCHK(hipMemcpyAsync(e->d_runs + g.launched, e->h_runs + g.launched, sizeof(RunDesc) * (g.nruns - g.launched),
                                          hipMemcpyHostToDevice, lst));
                    st[m] = e->d_state; rl[m] = e->d_runs + g.launched; cnt[m] = g.nruns - g.launched; rst[m] = g.first_launch ? 1u : 0u;
                    g.launched = g.nruns; g.first_launch = false;
                    m++;
                }
                if (!m) continue;
                tr_streams += m;
                if (pass == 0) {
                    // (the four arrays of this launch start at entry side_used of the second table; nothing overwrites them before the end)
                    for (int q = 0; q < 4; q++)
                        HIPCHK(hipMemcpyAsync(db + q * kMaxBatch, hb + q * kMaxBatch, sizeof(void *) * m, hipMemcpyHostToDevice, lst));
                    if (!side_any) HIPCHK(hipEventRecord(lead->ev[1][0], lst));
                    launch_encode_runs_multi(parser, m, (EncState *const *)db, (const RunDesc *const *)(db + kMaxBatch),
                                             (const uint32_t *)(db + 2 * kMaxBatch), (const uint32_t *)(db + 3 * kMaxBatch), lst);
                    HIPCHK(hipGetLastError());
                    side_used += m; side_any = true;
                    lead->stats.encode_launches++;
                    continue;
                }
                HIPCHK(hipMemcpyAsync(d_batch, h_batch, sizeof(void *) * 4 * kMaxBatch, hipMemcpyHostToDevice, lead->stream));
                HIPCHK(hipEventRecord(lead->ev[0][0], lead->stream));
                launch_encode_runs_multi(parser, m, (EncState *const *)d_batch, (const RunDesc *const *)(d_batch + kMaxBatch),
                                         (const uint32_t *)(d_batch + 2 * kMaxBatch), (const uint32_t *)(d_batch + 3 * kMaxBatch), lead->stream);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(lead->ev[0][1], lead->stream));
                HIPCHK(hipStreamSynchronize(lead->stream));          // (the pointer tables are reused by the next flavour / round)
                float ms = 0;
                if (hipEventElapsedTime(&ms, lead->ev[0][0], lead->ev[0][1]) == hipSuccess) lead->stats.encode_kernel_ms += ms;
                lead->stats.encode_launches++;
            }
        }
        if (trace) {
            for (int i = 0; i < n; i++) tr_wait += sizes[i] && state[i] == 1;
            fprintf(stderr, "batch trace: round %d: %u streams launched, %u wait for a verdict, %.1f ms%s\n", round, tr_streams, tr_wait,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count(), all_done ? " (last)" : "");
        }
        if (all_done) break;
        // the verdicts the waiting streams asked for, all in flight before the one wait
        const auto t_dup = tnow();
        for (int i = 0; i < n; i++) {
            if (!sizes[i] || state[i] != 1) continue;
            EncInstance *e = (EncInstance *)hs[i];
            EncInstance::Seg &g = e->seg;
            launch_dup_check(e->d_state, (uint32_t)sizes[i], g.need_blk, g.need_cnt, lead->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(e->h_dup + g.need_blk, e->h.dup_flags + g.need_blk, sizeof(uint32_t) * g.need_cnt, hipMemcpyDeviceToHost, lead->stream));
        }
        HIPCHK(hipStreamSynchronize(lead->stream));
        for (int i = 0; i < n; i++) {
            if (!sizes[i] || state[i] != 1) continue;
            EncInstance::Seg &g = ((EncInstance *)hs[i])->seg;
            g.dup_from = g.need_blk; g.dup_to = g.need_blk + g.need_cnt;
            state[i] = 0;
        }
        ms_dup += tms(t_dup, tnow());
    }
    if (side_any) {
        HIPCHK(hipEventRecord(lead->ev[1][1], t_batch.side));
        HIPCHK(hipStreamSynchronize(t_batch.side));
        float ms = 0;
        if (hipEventElapsedTime(&ms, lead->ev[1][0], lead->ev[1][1]) == hipSuccess) lead->stats.encode_kernel_ms += ms;
    }
    const auto t_drain = tnow();
    for (int i = 0; i < n; i++) {
        if (!sizes[i]) continue;
        EncInstance *e = (EncInstance *)hs[i];
        e->stats.chunks++;
        e->stats.input_bytes += sizes[i];
    }
    for (int i = 0; i < n; i++) {
        if (!sizes[i]) continue;
        int r = drain_arena((EncInstance *)hs[i], 0);
        if (r && !rc) rc = r;
    }
    if (trace) fprintf(stderr, "batch trace: %d rounds, duplicate checks %.1f ms, drain %.1f ms, whole call %.1f ms\n", round, ms_dup, tms(t_drain, tnow()), tms(t_begin, tnow()));
    return rc;
}

// CSCEnc_Encode, csc_enc.cpp:160-191: exactly one Read of raw_blocksize per chunk; a short read
// IS the chunk; Progress after every chunk, return value ignored.
int CSCEnc_Encode(CSCEncHandle p, ISeqInStream *is, ICompressProgress *progress)
{
    EncInstance *e = (EncInstance *)p;
    int ret = 0;
    uint64_t insize = 0;
    if (!e->h_in) {   // pinned staging for the caller's Read, kept with the recycled host resources
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipHostMalloc((void **)&e->res->h_in, e->props.raw_blocksize, hipHostMallocDefault));
        e->h_in = e->res->h_in;
    }
    for (;;) {
        size_t size = e->props.raw_blocksize;
        ret = is->Read(is, e->h_in, &size);
        if (ret >= 0 && size) {
            insize += size;
            ret = encode_chunk(e, e->h_in, size, false);
            if (progress) progress->Progress(progress, insize, (uint64_t)e->outsize);
        } else if (ret < 0) {
            ret = READ_ERROR;
        }
        if (ret < 0 || size == 0) break;
    }
    return ret;
}

// CSCEnc_Encode_Flush, csc_enc.cpp:193-203: WriteEOF (EncodeInt(SIG_EOF)) + Coder::Flush
int CSCEnc_Encode_Flush(CSCEncHandle p)
{
    EncInstance *e = (EncInstance *)p;
    HIPCHK(hipSetDevice(e->device));
    launch_encode_eof(e->d_state, e->stream);
    HIPCHK(hipGetLastError());
    int ev_used = 0;
    return drain_arena(e, ev_used);
}

// CSCEnc_Encode_Flush for many handles of one device: every EOF kernel queued on one stream, one wait, the last blocks handed to
// the users' streams in handle order.  (2048 task streams of the archiver: one round trip instead of 2048.)
int CSCMI_FlushBatch(int n, CSCEncHandle *hs)
{
    if (n <= 0) return 0;
    EncInstance *lead = (EncInstance *)hs[0];
    HIPCHK(hipSetDevice(lead->device));
    for (int i = 0; i < n; i += kMaxBatch) {
        const int m = n - i < kMaxBatch ? n - i : kMaxBatch;
        { int r = batch_args_ready(lead->device); if (r) return r; }
        for (int k = 0; k < m; k++) {
            EncInstance *e = (EncInstance *)hs[i + k];
            if (e->device != lead->device) return -1;
            launch_encode_eof(e->d_state, lead->stream);
        }
        HIPCHK(hipGetLastError());
        int r = drain_batch(m, hs + i, nullptr, lead->stream);
        if (r) return r;
    }
    return 0;
}

void CSCMI_GetStats(CSCEncHandle p, CSCMIStats *out)
{
    EncInstance *e = (EncInstance *)p;
    KernelStats ks;
    memset(&ks, 0, sizeof(ks));
    (void)hipSetDevice(e->device);
    if (hipMemcpy(&ks, &e->d_state->stats, sizeof(ks), hipMemcpyDeviceToHost) == hipSuccess) {
        e->stats.find_match_calls = ks.find_match_calls; e->stats.slide_positions = ks.slide_positions;
        e->stats.bt_steps = ks.bt_steps; e->stats.literals = ks.literals; e->stats.matches = ks.matches;
    }
    *out = e->stats;
}

#ifdef CSCMI_STAGE_TEST
// Test-only entry points (tests/stage/libcsc_stage.so -- the same sources built with -DCSCMI_STAGE_TEST; the product library does
// not have them): the analyzer kernel and the three forward filters on a caller-supplied buffer, one stage at a time.

// out: 7 words per 8 KiB block: type, bpb, dlt_bpb[5] (csc_analyzer.cpp:184-239, :166-182)
int CSCST_Analyze(CSCEncHandle p, const void *host, size_t size, uint32_t *out)
{
    EncInstance *e = (EncInstance *)p;
    if (!e || size == 0 || size > e->props.raw_blocksize) return -1;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->h.inbuf, host, size, hipMemcpyHostToDevice, e->stream));
    const uint32_t nblk = ((uint32_t)size + kMinBlock - 1) / kMinBlock;
    HIPCHK(hipMemsetAsync(e->h.binfo, 0xEE, sizeof(BlockInfo) * nblk, e->stream));
    launch_analyze(e->d_state, (uint32_t)size, e->d_entcoef, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->h_binfo, e->h.binfo, sizeof(BlockInfo) * nblk, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (uint32_t b = 0; b < nblk; b++) {
        out[b * 7] = e->h_binfo[b].type; out[b * 7 + 1] = e->h_binfo[b].bpb;
        for (int k = 0; k < 5; k++) out[b * 7 + 2 + k] = e->h_binfo[b].dlt_bpb[k];
    }
    return 0;
}

// kind 0 Forward_E89, 1 Foward_Dict (*result = its return value), 2 Forward_Delta with `chn` channels; buf is transformed in place
int CSCST_Filter(CSCEncHandle p, int kind, void *buf, size_t size, uint32_t chn, uint32_t *result)
{
    EncInstance *e = (EncInstance *)p;
    if (!e || size == 0 || size > e->props.raw_blocksize) return -1;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->h.inbuf, buf, size, hipMemcpyHostToDevice, e->stream));
    launch_stage_filter(e->d_state, (uint32_t)kind, (uint32_t)size, chn, e->h.dup_flags, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(buf, e->h.inbuf, size, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_small, e->h.dup_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (result) *result = e->h_small[0];
    return 0;
}
// the match finder's position counter (EncState::pos; MatchFinder::Init leaves it at vld_rge_ over zeroed tables): a start close to
// 0xFFFFFFF0 brings the renormalisation of csc_mf.cpp:108-114 within reach of a test.  Right after CSCEnc_Create.
int CSCST_SetPos(CSCEncHandle p, uint32_t pos)
{
    EncInstance *e = (EncInstance *)p;
    if (!e) return -1;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(&e->d_state->pos, &pos, sizeof(pos), hipMemcpyHostToDevice));
    return 0;
}
#endif   // CSCMI_STAGE_TEST

#ifdef CSCMI_TIMERS
// development aids (tools/gpu_timers.py, tools/gpu_trace.py): only in the -DCSCMI_TIMERS build, never in the product library
void CSCMI_DebugTimers(CSCEncHandle p, uint64_t *out16)
{
    EncInstance *e = (EncInstance *)p;
    KernelStats ks;
    memset(&ks, 0, sizeof(ks));
    (void)hipSetDevice(e->device);
    (void)hipMemcpy(&ks, &e->d_state->stats, sizeof(ks), hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) out16[i] = ks.tm[i];
}

void CSCMI_DebugSetMask(CSCEncHandle p, uint64_t mask)      // csc_kernels_dp4.inc: D5DBG
{
    EncInstance *e = (EncInstance *)p;
    (void)hipSetDevice(e->device);
    (void)hipMemcpy(&e->d_state->stats.trace[63][0], &mask, sizeof(mask), hipMemcpyHostToDevice);
}

void CSCMI_DebugTrace(CSCEncHandle p, uint64_t *out768)
{
    EncInstance *e = (EncInstance *)p;
    static KernelStats ks;
    memset(&ks, 0, sizeof(ks));
    (void)hipSetDevice(e->device);
    (void)hipMemcpy(&ks, &e->d_state->stats, sizeof(ks), hipMemcpyDeviceToHost);
    memcpy(out768, ks.trace, sizeof(ks.trace));
}
#endif   // CSCMI_TIMERS

}  // extern "C"
