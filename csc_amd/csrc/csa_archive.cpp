// csa_archive.cpp -- the `.csa` container around the libcsc streams (C ABI in
// include/csa_mi355x.h).  Host code, like the reference's src/archiver: directory scan, task split,
// block table, index, file I/O.  The streams themselves come from the HIP encoder/decoder behind
// CSCEnc_* / CSCDec_* in this same library, and the per-fragment adler32 from k_adler_pieces.
//
// Layout contract: CSA_Add writes what `csarc a -t1` writes -- tasks in task-id order, each chopped
// into archive blocks by the 1 MiB coalescing rule of the reference's writer -- while encoding many
// tasks at once on the GPU (one workgroup per task stream, CSCMI_EncodeDeviceChunkBatch).  A task's
// output is held in host memory until every lower-numbered task has been written.
#include <hip/hip_runtime.h>

#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include <utime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/csc_mi355x.h"
#include "../../include/csa_mi355x.h"

void launch_adler_pieces(const void *pieces, void *out, uint32_t npieces, hipStream_t st);

namespace {

constexpr uint32_t kMagicNum = 0x20130331u;          // csarc.cpp:283,593
constexpr uint64_t kHeaderSize = 24;                 // csarc.cpp:561
constexpr uint64_t kBlockCap = 1048576;              // csa_io.h:158
constexpr uint32_t kAdlerBase = 65521;
constexpr uint32_t kAdlerPiece = 16384;              // bytes per k_adler_pieces workgroup
constexpr int kMaxStreams = 2048;        // = kMaxBatch of csc_host.cpp: one launch takes them all; the GPU starts the next workgroup where one ends
constexpr int kStageBufs = 8;
constexpr uint32_t kMaxFrags = CSA_MAX_FRAGMENTS;
const char *const kDummyName = "****";               // csa_common.h:79

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// index model -- csa_typedef.h:12-90
// ---------------------------------------------------------------------------------------------
struct Entry {
    int64_t edate = 0, esize = 0, eattr = 0;
    char ext[4] = {0, 0, 0, 0};
    std::vector<CSAFrag> frags;
};
typedef std::map<std::string, Entry> Index;

struct FilePiece {                 // FileBlock: one file (or slice of it) inside a task
    std::string path;
    Index::iterator it;
    uint32_t checksum = 0;
    uint64_t off = 0, size = 0, posblock = 0;
};
struct Task {
    uint64_t total = 0;
    std::vector<FilePiece> files;
    uint32_t ab_id = 0;
    void add(const std::string &path, uint64_t off, uint64_t size, uint64_t posblock, uint32_t checksum, Index::iterator it)
    {
        FilePiece p;
        p.path = path; p.off = off; p.size = size; p.posblock = posblock; p.checksum = checksum; p.it = it;
        files.push_back(p);
        total += size;
    }
};
struct Extent { uint64_t off, size; };
typedef std::map<uint64_t, std::vector<Extent>> BlockIndex;

// ---------------------------------------------------------------------------------------------
// little-endian fields -- csa_indexpack.cpp:5-66
// ---------------------------------------------------------------------------------------------
void put_le(std::vector<uint8_t> &v, uint64_t x, int n) { for (int i = 0; i < n; i++) v.push_back((uint8_t)(x >> (8 * i))); }
void store_le(uint8_t *p, uint64_t x, int n) { for (int i = 0; i < n; i++) p[i] = (uint8_t)(x >> (8 * i)); }
uint64_t load_le(const uint8_t *p, int n) { uint64_t x = 0; for (int i = n - 1; i >= 0; i--) x = (x << 8) | p[i]; return x; }

// PackIndex, csa_indexpack.cpp:166-189.  The reference sizes the buffer with ArchiveBlocksSize (:127-134),
// which still counts `4 + arcname.size()` per task for a name ArchiveBlocksToBuf (:136-150) stopped writing;
// the surplus sits at the end of the buffer and is compressed with it.  Here it is zero.
std::vector<uint8_t> pack_index(const Index &index, const BlockIndex &abindex, const std::string &arcname)
{
    std::vector<uint8_t> out;
    uint64_t total = 4;
    put_le(out, index.size(), 4);
    for (const auto &kv : index) {
        const Entry &e = kv.second;
        total += 4 + kv.first.size() + 3 * 8 + 1 + e.frags.size() * 32;
        put_le(out, kv.first.size(), 4);
        out.insert(out.end(), kv.first.begin(), kv.first.end());
        put_le(out, (uint64_t)e.edate, 8);
        put_le(out, (uint64_t)e.esize, 8);
        put_le(out, (uint64_t)e.eattr, 8);
        out.push_back((uint8_t)std::min<size_t>(e.frags.size(), 255));   // plan_add refuses plans with more than 127
        for (const CSAFrag &f : e.frags) {
            put_le(out, f.bid, 4); put_le(out, f.checksum, 4);
            put_le(out, f.posblock, 8); put_le(out, f.size, 8); put_le(out, f.posfile, 8);
        }
    }
    total += 4;
    put_le(out, abindex.size(), 4);
    for (const auto &kv : abindex) {
        total += 8 + 4 + arcname.size() + 4 + kv.second.size() * 16;
        put_le(out, kv.first, 8);
        put_le(out, kv.second.size(), 4);
        for (const Extent &x : kv.second) { put_le(out, x.off, 8); put_le(out, x.size, 8); }
    }
    out.resize(total, 0);
    return out;
}

// UnpackIndex, csa_indexpack.cpp:191-211 -- bounds-checked here (the reference trusts the buffer)
bool unpack_index(Index &index, BlockIndex &abindex, const uint8_t *buf, uint64_t size)
{
    uint64_t pos = 0;
    auto need = [&](uint64_t n) { return pos + n <= size; };
    index.clear();
    abindex.clear();
    if (!need(4)) return false;
    uint32_t n = (uint32_t)load_le(buf + pos, 4); pos += 4;
    for (uint32_t i = 0; i < n; i++) {
        if (!need(4)) return false;
        uint32_t ln = (uint32_t)load_le(buf + pos, 4); pos += 4;
        if (!need((uint64_t)ln + 25)) return false;
        std::string name((const char *)buf + pos, ln); pos += ln;
        Entry e;
        e.edate = (int64_t)load_le(buf + pos, 8); pos += 8;
        e.esize = (int64_t)load_le(buf + pos, 8); pos += 8;
        e.eattr = (int64_t)load_le(buf + pos, 8); pos += 8;
        int nfr = (int8_t)buf[pos]; pos += 1;                       // int8_t: csa_indexpack.cpp:105
        if (nfr < 0) return false;                                  // > 127 fragments: unreadable (there too)
        for (int k = 0; k < nfr; k++) {
            if (!need(32)) return false;
            CSAFrag f;
            f.bid = (uint32_t)load_le(buf + pos, 4); f.checksum = (uint32_t)load_le(buf + pos + 4, 4);
            f.posblock = load_le(buf + pos + 8, 8); f.size = load_le(buf + pos + 16, 8); f.posfile = load_le(buf + pos + 24, 8);
            pos += 32;
            e.frags.push_back(f);
        }
        index.insert(std::make_pair(name, e));
    }
    if (!need(4)) return false;
    n = (uint32_t)load_le(buf + pos, 4); pos += 4;
    for (uint32_t i = 0; i < n; i++) {
        if (!need(12)) return false;
        uint64_t id = load_le(buf + pos, 8); pos += 8;
        uint32_t nb = (uint32_t)load_le(buf + pos, 4); pos += 4;
        if (!need((uint64_t)nb * 16)) return false;
        std::vector<Extent> v(nb);
        for (uint32_t k = 0; k < nb; k++) { v[k].off = load_le(buf + pos, 8); v[k].size = load_le(buf + pos + 8, 8); pos += 16; }
        abindex.insert(std::make_pair(id, v));
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// path matching -- csarc.cpp:16-35 (ispath; unix build: case sensitive), :807-816 (isselected)
// ---------------------------------------------------------------------------------------------
bool path_match(const char *a, const char *b)
{
    for (; *a; ++a, ++b) {
        if (*a == '*') {
            for (;; ++b) {
                if (path_match(a + 1, b)) return true;
                if (!*b) return false;
            }
        } else if (*a == '?') {
            if (!*b) return false;
        } else if (*a == *b && *a == '/' && a[1] == 0) {
            return true;
        } else if (*a != *b) {
            return false;
        }
    }
    return *b == 0 || *b == '/';
}

struct Selection {
    std::vector<std::string> names;
    bool has(const char *name) const
    {
        if (names.empty()) return true;
        for (const std::string &f : names) if (path_match(f.c_str(), name)) return true;
        return false;
    }
};

// ---------------------------------------------------------------------------------------------
// archive-block writer -- csa_io.h:145-201 (AsyncWriter::flush / Write), :596-603 (Finish).
// Same coalescing rule, kept in memory: `sizes` is the task's block table (offsets are assigned when
// the task is appended to the archive).  First member is the ISeqOutStream the encoder calls.
// ---------------------------------------------------------------------------------------------
struct BlockSink {
    ISeqOutStream os;
    std::vector<uint8_t> data;
    std::vector<uint64_t> sizes;
    uint64_t cur = 0, cap = kBlockCap;
    BlockSink() { os.Write = &BlockSink::write_cb; }
    void put(const void *buf, size_t size)
    {
        if (cur + size > cap) {
            if (cur) sizes.push_back(cur);
            cur = 0;
            cap = std::max<uint64_t>(kBlockCap, size);
        }
        const uint8_t *p = (const uint8_t *)buf;
        data.insert(data.end(), p, p + size);
        cur += size;
    }
    void finish() { if (cur) sizes.push_back(cur); cur = 0; }
    static size_t write_cb(void *p, const void *buf, size_t size) { ((BlockSink *)p)->put(buf, size); return size; }
};

struct MemSource {                 // MemReader, csa_io.h:428-443
    ISeqInStream is;
    const uint8_t *ptr; uint64_t size, pos;
    MemSource(const uint8_t *p, uint64_t n, uint64_t at) : ptr(p), size(n), pos(at) { is.Read = &MemSource::read_cb; }
    static SRes read_cb(void *p, void *buf, size_t *size)
    {
        MemSource *r = (MemSource *)p;
        uint64_t s = std::min<uint64_t>(*size, r->size - r->pos);
        memcpy(buf, r->ptr + r->pos, s);
        r->pos += s;
        *size = s;
        return 0;
    }
};

struct MemSink {                   // MemWriter, csa_io.h:445-460
    ISeqOutStream os;
    std::vector<uint8_t> buf; uint64_t pos = 0, cap;
    // `cap` comes from an archive header: the buffer grows with what is actually decoded, never beyond it
    explicit MemSink(uint64_t cap_) : cap(cap_) { buf.reserve((size_t)std::min<uint64_t>(cap_, 1u << 20)); os.Write = &MemSink::write_cb; }
    static size_t write_cb(void *p, const void *src, size_t size)
    {
        MemSink *w = (MemSink *)p;
        uint64_t s = std::min<uint64_t>(size, w->cap - w->pos);
        const uint8_t *b = (const uint8_t *)src;
        w->buf.insert(w->buf.end(), b, b + s);
        w->pos += s;
        return s;
    }
};

// ---------------------------------------------------------------------------------------------
// scan -- csarc.cpp:709-750 (unix scandir), :798-805 (addfile)
// ---------------------------------------------------------------------------------------------
void add_entry(Index &index, const Selection &sel, const std::string &name, int64_t edate, int64_t esize, int64_t eattr)
{
    if (!sel.has(name.c_str())) return;
    Entry &e = index[name];
    e.edate = edate; e.esize = esize; e.eattr = eattr;
}

void scan_path(Index &index, const Selection &sel, std::string name, bool recurse)
{
    while (name.size() > 1 && name[name.size() - 1] == '/') name.resize(name.size() - 1);
    struct stat sb;
    if (lstat(name.c_str(), &sb) != 0) {
        if (recurse || errno != ENOENT) perror(name.c_str());
        return;
    }
    const int64_t attr = 'u' + ((int64_t)sb.st_mode << 8);
    if (S_ISREG(sb.st_mode)) add_entry(index, sel, name, CSA_DecimalTime(sb.st_mtime), sb.st_size, attr);
    if (S_ISDIR(sb.st_mode)) {
        add_entry(index, sel, name == "/" ? std::string("/") : name + "/", CSA_DecimalTime(sb.st_mtime), 0, attr);
        if (!recurse) return;
        DIR *d = opendir(name.c_str());
        if (!d) { perror(name.c_str()); return; }
        while (dirent *dp = readdir(d)) {
            if (!strcmp(dp->d_name, ".") || !strcmp(dp->d_name, "..")) continue;
            scan_path(index, sel, (name == "/" ? name : name + "/") + dp->d_name, recurse);
        }
        closedir(d);
    }
}

// ---------------------------------------------------------------------------------------------
// task split -- csarc.cpp:490-557; comparators :76-92.  std::sort on purpose: the reference's order
// of equal keys is whatever libstdc++'s introsort leaves, and this is the same libstdc++.
// ---------------------------------------------------------------------------------------------
bool ext_less(Index::iterator a, Index::iterator b)
{
    int r = memcmp(a->second.ext, b->second.ext, 4);
    if (r != 0) return r < 0;
    if (a->second.esize > 64 * 1024 || b->second.esize > 64 * 1024) return a->second.esize < b->second.esize;
    return a->first < b->first;
}

// Not in the reference (CSAOptions::task_bytes): cut tasks so that none exceeds `cap` bytes.  A file larger than
// the room left in a task continues in the next one as another fragment; a file is never cut into more than 127
// fragments (the index stores the count in one signed byte, csa_indexpack.cpp:84,105).
std::vector<Task> cap_tasks(const std::vector<Task> &in, uint64_t cap)
{
    std::vector<Task> out;
    // slices a file already has (the -p split): every slice adds at most one short tail piece, so pieces of at least
    // esize / (127 - slices) bytes keep the file's total at or below 127 (plan_add refuses what still exceeds it)
    std::map<const void *, uint64_t> slices;
    for (const Task &t : in) for (const FilePiece &f : t.files) slices[(const void *)&*f.it]++;
    for (const Task &t : in) {
        Task cur;
        for (const FilePiece &f : t.files) {
            const uint64_t ns = slices[(const void *)&*f.it], whole = (uint64_t)f.it->second.esize;
            const uint64_t room127 = ns <= 1 ? 127 : (ns < 127 ? 127 - ns : 0);              // an unsplit file has no slice tails
            const uint64_t min_piece = room127 ? (whole + room127 - 1) / room127 : whole;
            uint64_t off = f.off, left = f.size;
            if (left == 0) { cur.add(f.path, off, 0, 0, 0, f.it); continue; }
            while (left) {
                uint64_t room = cur.total < cap ? cap - cur.total : 0;
                if (room < std::min<uint64_t>(left, std::max<uint64_t>(min_piece, 1)) && cur.total) {
                    out.push_back(cur);
                    cur = Task();
                    continue;
                }
                uint64_t n = std::min<uint64_t>(left, std::max<uint64_t>(room, min_piece));
                cur.add(f.path, off, n, 0, 0, f.it);
                off += n; left -= n;
            }
        }
        if (!cur.files.empty()) out.push_back(cur);
    }
    std::sort(out.begin(), out.end(), [](const Task &a, const Task &b) { return a.total > b.total; });
    return out;
}

std::vector<Task> plan_tasks(Index &index, int split_count)
{
    std::vector<Index::iterator> order;
    for (Index::iterator it = index.begin(); it != index.end(); ++it) {
        const std::string &n = it->first;
        if (n.empty() || n[n.size() - 1] == '/') continue;
        order.push_back(it);
        memset(it->second.ext, 0, 4);
        size_t dot = n.find_last_of('.'), slash = n.find_last_of('/');
        if (dot != std::string::npos && !(slash != std::string::npos && dot < slash))
            for (size_t i = 0; i < 4 && dot + 1 + i < n.size(); i++) it->second.ext[i] = (char)tolower(n[dot + 1 + i]);
    }
    std::sort(order.begin(), order.end(), ext_less);

    // csarc.cpp:516-530: the "single file" is the LAST entry visited while the count of non-empty files
    // is still one -- an empty file sorted behind the only non-empty one takes its place (and yields no task)
    uint32_t nonempty = 0;
    bool single = false;
    Index::iterator only = index.end();
    for (Index::iterator it : order) {
        if (it->second.esize > 0) nonempty++;
        if (nonempty == 1) { single = true; only = it; }
        else if (nonempty > 1) { single = false; break; }
    }

    std::vector<Task> tasks;
    if (single) {
        const uint64_t esize = (uint64_t)only->second.esize;
        uint64_t slice = esize / (uint64_t)split_count;
        if (slice < 1048576) slice = 1048576;
        slice += 4;
        for (uint64_t off = 0; off < esize;) {
            uint64_t n = std::min<uint64_t>(slice, esize - off);
            Task t;
            t.add(only->first, off, n, 0, 0, only);
            tasks.push_back(t);
            off += n;
        }
    } else {
        Task cur;
        for (size_t i = 0; i < order.size(); i++) {
            if (i && strncmp(order[i]->second.ext, order[i - 1]->second.ext, 4) && cur.total > 64 * 1024) {
                tasks.push_back(cur);
                cur = Task();
            }
            cur.add(order[i]->first, 0, (uint64_t)order[i]->second.esize, 0, 0, order[i]);
        }
        if (cur.total) tasks.push_back(cur);
    }
    std::sort(tasks.begin(), tasks.end(), [](const Task &a, const Task &b) { return a.total > b.total; });   // csarc.cpp:355
    return tasks;
}

// ---------------------------------------------------------------------------------------------
// device side of one Add: chunk staging, adler pieces
// ---------------------------------------------------------------------------------------------
struct AdlerPieceH { const uint8_t *ptr; uint32_t len; uint32_t pad; };
struct AdlerSumsH { uint64_t a, b; };

struct PieceRef { uint32_t slot; uint32_t file; uint32_t len; };     // which fragment a piece belongs to

void adler_fold(uint32_t &adler, uint64_t A, uint64_t B, uint64_t len)
{
    uint64_t a = adler & 0xFFFFu, b = (adler >> 16) & 0xFFFFu;
    b = (b + (len % kAdlerBase) * a + B % kAdlerBase) % kAdlerBase;
    a = (a + A % kAdlerBase) % kAdlerBase;
    adler = (uint32_t)(a | (b << 16));
}

struct Slot {
    int task = -1;
    CSCEncHandle h = nullptr;
    BlockSink *sink = nullptr;
    uint8_t *d_chunk = nullptr, *d_chunk2 = nullptr;   // the chunk being encoded / the one the read-ahead fills meanwhile
    // reader cursor (AsyncFileReader, csa_io.h:215-272)
    size_t fi = 0;
    uint64_t fprog = 0, cum = 0;
    int fd = -1;
    uint64_t mem = 0;
    int64_t pre_n = -1;          // >= 0: the next chunk is already in d_chunk2 (that many bytes); < -1: the read-ahead failed with that code
};

// What one thread needs to read chunks: a HIP stream, a ring of pinned staging buffers, the adler pieces of what it has read.
// Lane 0 is the caller's; lane 1 belongs to the read-ahead thread that runs while the caller sits in the batch encode call --
// the reference's reader thread per worker (csa_io.h:215-272), here one for all streams of the job.
struct FillLane {
    hipStream_t st = nullptr;
    uint8_t *stage[kStageBufs] = {nullptr};
    hipEvent_t stage_ev[kStageBufs] = {nullptr};
    bool stage_used[kStageBufs] = {false};
    int stage_next = 0;
    std::vector<AdlerPieceH> pieces;
    std::vector<PieceRef> refs;
    void *d_pieces = nullptr, *d_sums = nullptr;
    AdlerSumsH *h_sums = nullptr;
    size_t piece_cap = 0;
    uint64_t raw_bytes = 0;
};

struct AddJob {
    std::vector<Task> *tasks = nullptr;
    FillLane lane[2];
    uint32_t raw_blocksize = 2u << 20;
    std::vector<uint8_t *> chunk_groups;          // the slots' device chunks, kChunkGroup slots (x 2 buffers) per allocation
};
constexpr size_t kChunkGroup = 64;

#define HIP_OK(x) ((x) == hipSuccess)

// Fill the slot's device chunk with the next <= raw_blocksize bytes of its task (files back to back, what
// AsyncReader::Read hands CSCEnc_Encode: full chunks until the task ends, csa_io.h:66-96) and queue the
// adler pieces of every fragment part in it.  Returns the chunk size, or < 0.
int64_t fill_chunk(AddJob &J, FillLane &Ln, uint32_t slot_idx, Slot &s, uint8_t *d_dst)
{
    Task &t = (*J.tasks)[s.task];
    uint64_t filled = 0;
    uint8_t *stage = nullptr;
    int sb = -1;
    while (s.fi < t.files.size()) {
        FilePiece &f = t.files[s.fi];
        if (s.fd < 0) {
            s.fd = open(f.path.c_str(), O_RDONLY);
            if (s.fd < 0) { f.size = 0; s.fi++; continue; }        // csa_io.h:232-236
            f.posblock = s.cum;                                     // csa_io.h:239
            s.fprog = 0;
        }
        if (s.fprog == f.size) { close(s.fd); s.fd = -1; s.fi++; continue; }   // csa_io.h:266-270 (also empty files)
        if (filled == J.raw_blocksize) break;
        if (!stage) {
            sb = Ln.stage_next; Ln.stage_next = (Ln.stage_next + 1) % kStageBufs;
            if (Ln.stage_used[sb] && !HIP_OK(hipEventSynchronize(Ln.stage_ev[sb]))) return CSCMI_DEVICE_ERROR;
            stage = Ln.stage[sb];
        }
        uint64_t want = std::min<uint64_t>(f.size - s.fprog, J.raw_blocksize - filled);
        uint64_t got = 0;
        while (got < want) {
            ssize_t r = pread(s.fd, stage + filled + got, want - got, (off_t)(f.off + s.fprog + got));
            if (r <= 0) break;
            got += (uint64_t)r;
        }
        if (got < want) {
            fprintf(stderr, "csa-mi355x: %s: short read (file changed while archiving?)\n", f.path.c_str());
            return READ_ERROR;
        }
        for (uint64_t o = 0; o < got; o += kAdlerPiece) {
            uint32_t n = (uint32_t)std::min<uint64_t>(kAdlerPiece, got - o);
            Ln.pieces.push_back(AdlerPieceH{d_dst + filled + o, n, 0});
            Ln.refs.push_back(PieceRef{slot_idx, (uint32_t)s.fi, n});
        }
        filled += got; s.fprog += got; s.cum += got;
    }
    if (filled) {
        if (!HIP_OK(hipMemcpyAsync(d_dst, stage, filled, hipMemcpyHostToDevice, Ln.st))) return CSCMI_DEVICE_ERROR;
        if (!HIP_OK(hipEventRecord(Ln.stage_ev[sb], Ln.st))) return CSCMI_DEVICE_ERROR;
        Ln.stage_used[sb] = true;
        Ln.raw_bytes += filled;
    }
    return (int64_t)filled;
}

int run_adler(AddJob &J, FillLane &Ln, std::vector<Slot> &slots)
{
    size_t n = Ln.pieces.size();
    if (!n) return HIP_OK(hipStreamSynchronize(Ln.st)) ? 0 : CSCMI_DEVICE_ERROR;      // (still: the lane's uploads must have landed)
    if (n > Ln.piece_cap) {
        size_t cap = std::max<size_t>(n * 2, 4096);
        if (Ln.d_pieces) (void)hipFree(Ln.d_pieces);
        if (Ln.d_sums) (void)hipFree(Ln.d_sums);
        if (Ln.h_sums) (void)hipHostFree(Ln.h_sums);
        Ln.d_pieces = Ln.d_sums = nullptr; Ln.h_sums = nullptr;
        if (!HIP_OK(hipMalloc(&Ln.d_pieces, cap * sizeof(AdlerPieceH))) || !HIP_OK(hipMalloc(&Ln.d_sums, cap * sizeof(AdlerSumsH)))
            || !HIP_OK(hipHostMalloc((void **)&Ln.h_sums, cap * sizeof(AdlerSumsH), hipHostMallocDefault)))
            return CSCMI_DEVICE_ERROR;
        Ln.piece_cap = cap;
    }
    if (!HIP_OK(hipMemcpyAsync(Ln.d_pieces, Ln.pieces.data(), n * sizeof(AdlerPieceH), hipMemcpyHostToDevice, Ln.st))) return CSCMI_DEVICE_ERROR;
    launch_adler_pieces(Ln.d_pieces, Ln.d_sums, (uint32_t)n, Ln.st);
    if (!HIP_OK(hipGetLastError())) return CSCMI_DEVICE_ERROR;
    if (!HIP_OK(hipMemcpyAsync(Ln.h_sums, Ln.d_sums, n * sizeof(AdlerSumsH), hipMemcpyDeviceToHost, Ln.st))) return CSCMI_DEVICE_ERROR;
    if (!HIP_OK(hipStreamSynchronize(Ln.st))) return CSCMI_DEVICE_ERROR;
    for (size_t i = 0; i < n; i++) {
        const PieceRef &r = Ln.refs[i];
        FilePiece &f = (*J.tasks)[slots[r.slot].task].files[r.file];
        adler_fold(f.checksum, Ln.h_sums[i].a, Ln.h_sums[i].b, r.len);       // csa_io.h:250, piecewise
    }
    Ln.pieces.clear();
    Ln.refs.clear();
    return 0;
}

void free_job(AddJob &J, std::vector<Slot> &slots)
{
    for (Slot &s : slots) {
        if (s.h) CSCEnc_Destroy(s.h);
        if (s.fd >= 0) close(s.fd);
        delete s.sink;
        s = Slot();
    }
    for (FillLane &Ln : J.lane) {
        for (int i = 0; i < kStageBufs; i++) {
            if (Ln.stage[i]) (void)hipHostFree(Ln.stage[i]);
            if (Ln.stage_ev[i]) (void)hipEventDestroy(Ln.stage_ev[i]);
            Ln.stage[i] = nullptr; Ln.stage_ev[i] = nullptr;
        }
        if (Ln.d_pieces) (void)hipFree(Ln.d_pieces);
        if (Ln.d_sums) (void)hipFree(Ln.d_sums);
        if (Ln.h_sums) (void)hipHostFree(Ln.h_sums);
        if (Ln.st) (void)hipStreamDestroy(Ln.st);
    }
    for (uint8_t *g : J.chunk_groups) (void)hipFree(g);
    J = AddJob();
}

bool write_all(int fd, const uint8_t *p, uint64_t n, uint64_t off)
{
    while (n) {
        ssize_t w = pwrite(fd, p, n, (off_t)off);
        if (w <= 0) return false;
        p += w; n -= (uint64_t)w; off += (uint64_t)w;
    }
    return true;
}

// compress_mt, csarc.cpp:338-409, with ONE logical worker order but many streams in flight:
// every task stream is what CompressionWorker::do_work (csa_worker.cpp:23-56) produces.
// `emit(task, sink)` receives every finished task in task order (the order a single worker would have written them).
typedef std::function<int(size_t, BlockSink &)> EmitFn;
int encode_tasks(std::vector<Task> &tasks, const EmitFn &emit, const CSAOptions &o, CSAStats *st)
{
    const size_t nt = tasks.size();
    if (!nt) return 0;
    AddJob J;
    J.tasks = &tasks;
    std::vector<Slot> slots;
    std::vector<BlockSink *> done(nt, nullptr);
    size_t next_admit = 0, next_write = 0;
    int rc = 0;

    size_t mem_free = 0, mem_total = 0;
    if (!HIP_OK(hipMemGetInfo(&mem_free, &mem_total))) return CSCMI_DEVICE_ERROR;
    const uint64_t budget = o.hbm_budget ? o.hbm_budget : (uint64_t)mem_free / 4 * 3;
    int max_streams = o.device_streams > 0 ? o.device_streams : kMaxStreams;
    if (max_streams > kMaxStreams) max_streams = kMaxStreams;
    uint64_t mem_used = 0;

    bool ok = true;
    for (FillLane &Ln : J.lane) {
        ok = ok && HIP_OK(hipStreamCreateWithFlags(&Ln.st, hipStreamNonBlocking));
        for (int i = 0; ok && i < kStageBufs; i++)
            ok = HIP_OK(hipHostMalloc((void **)&Ln.stage[i], J.raw_blocksize, hipHostMallocDefault)) && HIP_OK(hipEventCreate(&Ln.stage_ev[i]));
    }
    if (!ok) { free_job(J, slots); return CSCMI_DEVICE_ERROR; }
    int device = 0;
    (void)hipGetDevice(&device);
    // CSA_READAHEAD=0 (diagnostics): every chunk is read by the calling thread between two encode calls, as in rounds 1-4
    static const bool readahead = [] { const char *e = getenv("CSA_READAHEAD"); return !(e && atoi(e) == 0); }();
    std::vector<uint8_t> fin;

    std::vector<CSCEncHandle> hs;
    std::vector<const void *> ptrs;
    std::vector<size_t> sizes;
    std::vector<uint32_t> live;

    auto flush_written = [&]() -> int {
        // hand finished tasks on in task-id order: the archive a single worker would have written
        while (next_write < nt && done[next_write]) {
            BlockSink *k = done[next_write];
            int r = emit(next_write, *k);
            if (r) return r;
            if (st) st->n_blocks += (uint32_t)k->sizes.size();
            delete k;
            done[next_write] = nullptr;
            next_write++;
        }
        return 0;
    };

    while (rc == 0) {
        // ---- admit tasks into free slots, in task-id order (largest first, csarc.cpp:355)
        double ts = now_s();
        while (next_admit < nt) {
            CSCProps p;
            CSCEncProps_Init(&p, (uint32_t)std::min<uint64_t>(o.dict_size, tasks[next_admit].total), o.level);   // csa_worker.cpp:35
            uint64_t mem = CSCEnc_EstMemUsage(&p) + (24ull << 20);
            size_t active = 0, free_slot = slots.size();
            for (size_t i = 0; i < slots.size(); i++) { if (slots[i].task >= 0) active++; else if (free_slot == slots.size()) free_slot = i; }
            if ((int)active >= max_streams) break;
            if (active && mem_used + mem > budget) break;
            if (free_slot == slots.size()) slots.push_back(Slot());
            Slot &s = slots[free_slot];
            if (!s.d_chunk) {
                const size_t stride = ((size_t)J.raw_blocksize + 64 + 255) & ~(size_t)255;
                if (free_slot / kChunkGroup >= J.chunk_groups.size()) {
                    uint8_t *g = nullptr;
                    if (!HIP_OK(hipMalloc((void **)&g, stride * kChunkGroup * 2))) { rc = CSCMI_DEVICE_ERROR; break; }
                    J.chunk_groups.push_back(g);
                }
                s.d_chunk = J.chunk_groups[free_slot / kChunkGroup] + (free_slot % kChunkGroup) * 2 * stride;
                s.d_chunk2 = s.d_chunk + stride;
            }
            s.sink = new BlockSink();
            s.h = CSCEnc_Create(&p, &s.sink->os, NULL);
            if (!s.h) { delete s.sink; s.sink = nullptr; rc = CSCMI_DEVICE_ERROR; break; }
            uint8_t hdr[CSC_PROP_SIZE];
            CSCEnc_WriteProperties(&p, hdr, 0);
            s.sink->put(hdr, CSC_PROP_SIZE);                               // csa_worker.cpp:38-42
            s.task = (int)next_admit; s.fi = 0; s.fprog = 0; s.cum = 0; s.fd = -1; s.mem = mem; s.pre_n = -1;
            mem_used += mem;
            next_admit++;
            if (st) st->peak_streams = std::max<uint32_t>(st->peak_streams, (uint32_t)active + 1);
        }
        if (rc) break;
        if (st) st->seconds_setup += now_s() - ts;

        // ---- one chunk per live stream: the one the read-ahead brought in during the last encode call, or read + upload now
        // (a stream's first chunk); adler32 of both lanes' pieces (lane 1's are the ones of the chunks swapped in here)
        ts = now_s();
        hs.clear(); ptrs.clear(); sizes.clear(); live.clear();
        bool any_pre = false;
        for (size_t i = 0; i < slots.size(); i++) {
            Slot &s = slots[i];
            if (s.task < 0) continue;
            int64_t n;
            if (s.pre_n != -1) {
                n = s.pre_n; s.pre_n = -1;
                if (n >= 0) { std::swap(s.d_chunk, s.d_chunk2); any_pre = true; }
            } else n = fill_chunk(J, J.lane[0], (uint32_t)i, s, s.d_chunk);
            if (n < 0) { rc = (int)n; break; }
            live.push_back((uint32_t)i);
            if (n) { hs.push_back(s.h); ptrs.push_back(s.d_chunk); sizes.push_back((size_t)n); }
        }
        if (rc) break;
        if (live.empty()) break;
        rc = run_adler(J, J.lane[0], slots);                                // also waits for the uploads
        if (rc == 0 && any_pre) rc = run_adler(J, J.lane[1], slots);
        if (rc) break;
        if (st) st->seconds_io += now_s() - ts;
        // which streams end with this chunk: decided HERE -- the read-ahead below moves the cursors of the others on
        fin.assign(slots.size(), 0);
        for (uint32_t i : live) fin[i] = slots[i].fi >= tasks[slots[i].task].files.size() ? 1 : 0;

        // ---- advance every stream by its chunk with one launch per parser flavour; meanwhile a second thread reads the NEXT
        // chunk of every stream that goes on into the slots' other buffers (csa_io.h:215-272: the reference's reader thread)
        if (!hs.empty()) {
            double t0 = now_s();
            std::thread reader;
            bool more = false;
            for (uint32_t i : live) more = more || !fin[i];
            if (readahead && more) {
                // (nothing may leave the lambda by exception: it would cross the C ABI as std::terminate -- a failed allocation in the reader becomes READ_ERROR for its stream)
                auto body = [&]() {
                    (void)hipSetDevice(device);
                    for (uint32_t i : live) {
                        Slot &s = slots[i];
                        if (fin[i]) continue;
                        int64_t n;
                        try { n = fill_chunk(J, J.lane[1], i, s, s.d_chunk2); } catch (...) { n = -1; }
                        s.pre_n = n < 0 ? (n == -1 ? (int64_t)READ_ERROR : n) : n;
                        if (n < 0) break;
                    }
                };
                try { reader = std::thread(body); } catch (...) { body(); }          // (no thread to be had: read ahead on this one, before the launch)
            }
            rc = CSCMI_EncodeDeviceChunkBatch((int)hs.size(), hs.data(), ptrs.data(), sizes.data());
            const double t1 = now_s();
            if (reader.joinable()) reader.join();
            if (st) { st->seconds_encode += t1 - t0; st->seconds_io += now_s() - t1; }      // (time spent waiting for the reader after the launch is I/O, not encode)
            if (rc) break;
        }

        // ---- streams whose task has no bytes left: EOF, flush, hand the blocks to the writer
        ts = now_s();
        hs.clear();
        for (uint32_t i : live) if (fin[i]) hs.push_back(slots[i].h);
        rc = CSCMI_FlushBatch((int)hs.size(), hs.data());                   // csa_worker.cpp:49, every finished stream in one round trip
        if (rc) break;
        for (uint32_t i : live) {
            Slot &s = slots[i];
            if (!fin[i]) continue;
            CSCEnc_Destroy(s.h);
            s.h = nullptr;
            s.sink->finish();                                               // csa_io.h:596-603
            done[s.task] = s.sink;
            s.sink = nullptr;
            mem_used -= s.mem;
            s.task = -1;
        }
        if (rc) break;
        if (st) st->seconds_setup += now_s() - ts;
        rc = flush_written();
    }
    if (rc == 0) rc = flush_written();
    if (st) st->raw_bytes += J.lane[0].raw_bytes + J.lane[1].raw_bytes;
    for (BlockSink *k : done) delete k;
    free_job(J, slots);
    return rc;
}

// compress_index, csarc.cpp:219-288 (without the header fields, which the caller writes)
int encode_index(const std::vector<uint8_t> &raw, std::vector<uint8_t> &out)
{
    BlockSink sink;
    CSCProps p;
    CSCEncProps_Init(&p, 256 * 1024, 2);                                    // csarc.cpp:251
    CSCEncHandle h = CSCEnc_Create(&p, &sink.os, NULL);
    if (!h) return CSCMI_DEVICE_ERROR;
    uint8_t hdr[CSC_PROP_SIZE];
    CSCEnc_WriteProperties(&p, hdr, 0);
    sink.put(hdr, CSC_PROP_SIZE);
    MemSource src(raw.data(), raw.size(), 0);
    int rc = CSCEnc_Encode(h, &src.is, NULL);
    int rc2 = CSCEnc_Encode_Flush(h);
    CSCEnc_Destroy(h);
    if (rc < 0) return rc;
    if (rc2 < 0) return rc2;
    out.swap(sink.data);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// reading side
// ---------------------------------------------------------------------------------------------
// check_header, csarc.cpp:577-598
bool header_ok(const uint8_t *h)
{
    return h[0] == 'C' && h[1] == 'S' && h[2] == 'A' && h[7] == '1' && (uint32_t)load_le(h + 3, 4) == kMagicNum;
}

// decompress_index, csarc.cpp:290-336.  0 ok, 1 cannot open / bad header, -1 undecodable index.
int read_index(const char *arcname, Index &index, BlockIndex &abindex, std::vector<uint8_t> *raw_out = nullptr)
{
    int fd = open(arcname, O_RDONLY);
    if (fd < 0) { fprintf(stderr, "Can not open file %s\n", arcname); return 1; }
    uint8_t hdr[kHeaderSize];
    if (pread(fd, hdr, kHeaderSize, 0) != (ssize_t)kHeaderSize || !header_ok(hdr)) {
        fprintf(stderr, "Invalid csarc file\n");
        close(fd);
        return 1;
    }
    uint64_t index_pos = load_le(hdr + 8, 8);
    uint32_t csize = (uint32_t)load_le(hdr + 16, 4), rsize = (uint32_t)load_le(hdr + 20, 4);
    {
        struct stat sb;
        if (fstat(fd, &sb) != 0 || index_pos > (uint64_t)sb.st_size || csize > (uint64_t)sb.st_size - index_pos) {
            fprintf(stderr, "Invalid csarc file\n");
            close(fd);
            return -1;
        }
    }
    std::vector<uint8_t> comp(csize);
    uint64_t got = 0;
    while (got < csize) {
        ssize_t r = pread(fd, comp.data() + got, csize - got, (off_t)(index_pos + got));
        if (r <= 0) break;
        got += (uint64_t)r;
    }
    close(fd);
    if (got < csize || csize < CSC_PROP_SIZE) { fprintf(stderr, "Invalid csarc file\n"); return -1; }
    CSCProps p;
    CSCDec_ReadProperties(&p, comp.data());
    MemSource src(comp.data(), csize, CSC_PROP_SIZE);
    MemSink sink(rsize);
    CSCDecHandle h = CSCDec_Create(&p, &src.is, NULL);
    if (!h) return -1;
    int rc = CSCDec_Decode(h, &sink.os, NULL);
    CSCDec_Destroy(h);
    if (rc < 0 || sink.pos != rsize || sink.buf.size() != rsize || !unpack_index(index, abindex, sink.buf.data(), rsize)) {
        fprintf(stderr, "Invalid csarc file (index)\n");
        return -1;
    }
    if (raw_out) raw_out->swap(sink.buf);
    return 0;
}

// AsyncArchiveReader, csa_io.h:466-542: the task's archive blocks, back to back
struct ExtentSource {
    ISeqInStream is;
    int fd;
    const std::vector<Extent> *ext;
    size_t bi = 0;
    uint64_t bprog = 0;
    ExtentSource(int f, const std::vector<Extent> *e) : fd(f), ext(e) { is.Read = &ExtentSource::read_cb; }
    static SRes read_cb(void *p, void *buf, size_t *size)
    {
        ExtentSource *r = (ExtentSource *)p;
        size_t done = 0;
        while (done < *size && r->bi < r->ext->size()) {
            const Extent &x = (*r->ext)[r->bi];
            uint64_t n = std::min<uint64_t>(*size - done, x.size - r->bprog);
            ssize_t g = n ? pread(r->fd, (uint8_t *)buf + done, n, (off_t)(x.off + r->bprog)) : 0;
            if (g < 0) return -1;
            if (n && g == 0) break;                                          // truncated archive: EOF
            done += (size_t)g; r->bprog += (uint64_t)g;
            if (r->bprog == x.size) { r->bi++; r->bprog = 0; }
        }
        *size = done;
        return 0;
    }
};

void set_file_meta(const std::string &path, int64_t date, int64_t attr)      // OutputFile::close, csa_file.h:123-142
{
    if (date > 0) {
        struct utimbuf ub;
        ub.actime = time(NULL);
        ub.modtime = (time_t)CSA_UnixTime(date);
        utime(path.c_str(), &ub);
    }
    if ((attr & 255) == 'u') chmod(path.c_str(), (mode_t)(attr >> 8));
}

// makepath, csa_file.cpp:4-61 (unix half)
void make_path(std::string path, int64_t date, int64_t attr)
{
    for (size_t i = 0; i < path.size(); i++) {
        if (path[i] == '\\' || path[i] == '/') {
            path[i] = 0;
            mkdir(path.c_str(), 0777);
            path[i] = '/';
        }
    }
    if (!path.empty() && path[path.size() - 1] == '/') set_file_meta(path.substr(0, path.size() - 1), date, attr);
}

// AsyncFileWriter, csa_io.h:274-408: the decoded task stream is cut into its fragments (sorted by posblock);
// bytes between fragments are skipped, each fragment is written at its file offset and verified.
struct FragSink {
    ISeqOutStream os;
    std::vector<FilePiece> *files;
    size_t fi = 0;
    uint64_t fprog = 0, cum = 0;
    int fd = -1;
    bool dummy = false, open_ = false, done = false;
    uint32_t sum = 0;
    std::atomic<uint32_t> *failures;
    FragSink(std::vector<FilePiece> *f, std::atomic<uint32_t> *fl) : files(f), failures(fl)
    {
        os.Write = &FragSink::write_cb;
        if (files->empty()) done = true;
    }
    void close_current(bool verify)
    {
        FilePiece &f = (*files)[fi];
        if (!dummy && fd >= 0) close(fd);
        fd = -1; open_ = false;
        if (!dummy) set_file_meta(f.path, f.it->second.edate, f.it->second.eattr);
        if (verify && f.checksum != sum) {
            fprintf(stderr, "******** %s extraction/verify failed\n", f.it->first.c_str());   // csa_io.h:332,398
            failures->fetch_add(1);
        }
    }
    size_t put(const uint8_t *buf, size_t size)
    {
        if (done) return CSC_WRITE_ABORT;                                   // csa_io.h:176-179 once finished_
        uint64_t bprog = 0;
        while (!done && bprog < size) {
            FilePiece &f = (*files)[fi];
            if (cum + size <= f.posblock) break;                            // csa_io.h:306,347: nothing of it here
            if (!open_) {
                dummy = f.path == kDummyName;
                if (!dummy) {
                    fd = open(f.path.c_str(), O_RDWR | O_CREAT, 0666);
                    if (fd < 0) {                                           // csa_io.h:309-318
                        perror(f.path.c_str());
                        if (++fi >= files->size()) done = true;
                        continue;
                    }
                }
                open_ = true; fprog = 0; sum = 0;
                if (cum + bprog < f.posblock) bprog = f.posblock - cum;     // csa_io.h:322-324
            }
            uint64_t n = std::min<uint64_t>(size - bprog, f.size - fprog);
            if (!dummy && n && !write_all(fd, buf + bprog, n, f.off + fprog)) perror(f.path.c_str());
            sum = CSA_Adler32(sum, buf + bprog, n);
            fprog += n; bprog += n;
            if (fprog == f.size) {
                close_current(true);
                if (++fi >= files->size()) done = true;
            }
        }
        cum += size;
        return size;
    }
    void finish()                                                            // csa_io.h:390-406
    {
        if (open_) close_current(true);
    }
    static size_t write_cb(void *p, const void *buf, size_t size) { return ((FragSink *)p)->put((const uint8_t *)buf, size); }
};

// DecompressionWorker::do_work, csa_worker.cpp:59-90, for a WAVE of tasks: every task gets its own source, sink
// and decoder handle exactly as one worker would set them up; then all of them are decoded together, one kernel
// launch per round (CSCMI_DecodeBatch).  Returns the worst return code of the wave (0 = all fine).
int decode_wave(int arc_fd, std::vector<Task> &tasks, size_t first, size_t count, const BlockIndex &abindex,
                std::atomic<uint32_t> *failures)
{
    static const std::vector<Extent> kNoBlocks;
    std::vector<ExtentSource *> srcs;
    std::vector<FragSink *> sinks;
    std::vector<CSCDecHandle> hs;
    std::vector<ISeqOutStream *> oss;
    int worst = 0;
    for (size_t t = first; t < first + count; t++) {
        auto ab = abindex.find(tasks[t].ab_id);
        ExtentSource *src = new ExtentSource(arc_fd, ab == abindex.end() ? &kNoBlocks : &ab->second);
        FragSink *sink = new FragSink(&tasks[t].files, failures);
        srcs.push_back(src);
        sinks.push_back(sink);
        uint8_t hdr[CSC_PROP_SIZE];
        size_t n = CSC_PROP_SIZE;
        src->is.Read(&src->is, hdr, &n);                                    // csa_worker.cpp:77-80
        CSCDecHandle h = NULL;
        if (n == CSC_PROP_SIZE) {
            CSCProps p;
            CSCDec_ReadProperties(&p, hdr);
            h = CSCDec_Create(&p, &src->is, NULL);
        }
        if (!h) { worst = DECODE_ERROR; continue; }
        hs.push_back(h);
        oss.push_back(&sink->os);
    }
    std::vector<int> rcs(hs.size(), 0);
    int rc = CSCMI_DecodeBatch((int)hs.size(), hs.data(), oss.data(), rcs.data());
    if (rc < 0) worst = rc;
    for (int r : rcs) if (r < 0) worst = r;
    for (CSCDecHandle h : hs) CSCDec_Destroy(h);
    for (FragSink *k : sinks) { k->finish(); delete k; }
    for (ExtentSource *x : srcs) delete x;
    return worst;
}

// A stored name Extract may follow: not empty, no `..` component (with either separator).  Absolute names and drive letters are
// re-rooted under to_dir by the reference's own rule (csarc.cpp:613-627) and stay allowed.
bool name_is_safe(const std::string &n)
{
    if (n.empty()) return false;
    size_t i = 0;
    while (i <= n.size()) {
        size_t j = i;
        while (j < n.size() && n[j] != '/' && n[j] != '\\') j++;
        if (j - i == 2 && n[i] == '.' && n[i + 1] == '.') return false;
        i = j + 1;
    }
    return true;
}

// Extract / Test share everything but the output names: csarc.cpp:600-650, :667-700, decompress_mt :411-470
int read_archive(const char *arcname, const Selection &sel, const CSAOptions &o, bool extract, CSAStats *st)
{
    double t0 = now_s();
    Index index;
    BlockIndex abindex;
    int rc = read_index(arcname, index, abindex);
    if (rc) return rc;

    std::vector<Task> tasks;
    std::map<uint64_t, size_t> idmap;
    std::string to_dir = o.to_dir && o.to_dir[0] ? o.to_dir : "./";
    uint32_t unsafe = 0;
    for (Index::iterator it = index.begin(); it != index.end(); ++it) {
        if (!sel.names.empty() && !sel.has(it->first.c_str())) continue;
        if (!name_is_safe(it->first)) {
            fprintf(stderr, "csa-mi355x: entry \"%s\" skipped: empty name or `..` component\n", it->first.c_str());
            unsafe++;
            continue;
        }
        std::string out_name = kDummyName;
        if (extract) {
            out_name = it->first;                                            // csarc.cpp:613-627
            if (out_name.size() > 1 && out_name[1] == ':') {
                if (out_name.size() > 2 && (out_name[2] == '/' || out_name[2] == '\\')) out_name = out_name.substr(0, 1) + out_name.substr(2);
                else out_name[1] = '/';
            }
            if (out_name[0] != '/' && to_dir[to_dir.size() - 1] != '/') out_name = to_dir + '/' + out_name;
            else out_name = to_dir + out_name;
            for (char &c : out_name) if (c == '\\') c = '/';
        }
        for (const CSAFrag &f : it->second.frags) {
            size_t ti;
            auto m = idmap.find(f.bid);
            if (m == idmap.end()) {
                ti = tasks.size();
                idmap[f.bid] = ti;
                tasks.push_back(Task());
                tasks[ti].ab_id = f.bid;
            } else ti = m->second;
            if (f.size) tasks[ti].add(out_name, f.posfile, f.size, f.posblock, f.checksum, it);
        }
        if (extract) {
            make_path(out_name, it->second.edate, it->second.eattr);
            if (out_name[out_name.size() - 1] != '/') {                      // csarc.cpp:643-648: create/empty the file
                int fd = open(out_name.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
                if (fd >= 0) close(fd); else perror(out_name.c_str());
                set_file_meta(out_name, it->second.edate, it->second.eattr);
            }
        }
    }
    std::sort(tasks.begin(), tasks.end(), [](const Task &a, const Task &b) { return a.total > b.total; });    // csarc.cpp:430
    for (Task &t : tasks)
        std::sort(t.files.begin(), t.files.end(), [](const FilePiece &a, const FilePiece &b) { return a.posblock < b.posblock; });

    int arc_fd = open(arcname, O_RDONLY);
    if (arc_fd < 0) return -1;
    // waves of tasks, largest first (csarc.cpp:430): as many streams per launch as there are CUs, within the HBM budget
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { close(arc_fd); return CSCMI_DEVICE_ERROR; }
    const uint64_t budget = o.hbm_budget ? o.hbm_budget : (uint64_t)mem_free / 4 * 3;
    const size_t width = o.device_streams > 0 ? (size_t)o.device_streams : 256;
    std::atomic<uint32_t> failures(0);
    int worst = 0;
    double t1 = now_s();
    for (size_t first = 0; first < tasks.size();) {
        size_t count = 0;
        uint64_t mem = 0;
        while (first + count < tasks.size() && count < width) {
            // a decoder holds the dictionary (unknown before its header is read; bounded by the task's raw size
            // and by 1 GiB) plus ~24 MiB of rings and buffers
            uint64_t need = std::min<uint64_t>(std::max<uint64_t>(tasks[first + count].total, 1u << 20), 1ull << 30) + (24ull << 20);
            if (count && mem + need > budget) break;
            mem += need;
            count++;
        }
        int r = decode_wave(arc_fd, tasks, first, count, abindex, &failures);
        if (r < 0) worst = r;
        first += count;
    }
    close(arc_fd);
    if (st) {
        st->n_entries = (uint32_t)index.size();
        st->n_tasks = (uint32_t)tasks.size();
        st->verify_failures = failures.load();
        for (const Task &t : tasks) st->raw_bytes += t.total;
        st->seconds_encode = now_s() - t1;
        st->seconds_total = now_s() - t0;
    }
    if (worst < 0) {
        fprintf(stderr, "Extraction error, archive corrupted\n");           // csarc.cpp:464-467
        return -1;
    }
    return unsafe ? CSA_UNSAFE_NAME : 0;
}

Selection make_selection(const char *const *names, int n)
{
    Selection s;
    for (int i = 0; i < n; i++) s.names.push_back(names[i]);
    return s;
}

}   // namespace

extern "C" {

void CSA_OptionsInit(CSAOptions *o)                                          // csarc.cpp:139-147
{
    memset(o, 0, sizeof(*o));
    o->level = 2;
    o->dict_size = 32000000;
    o->mt_count = 1;
    o->split_count = 1;
    o->to_dir = "./";
}

// csa_adler32.cpp:63-129 -- zlib's adler32 with the running value passed in.  Deferred modulo over
// blocks of 5552 bytes (the largest n with 255 n (n+1)/2 + (n+1)(BASE-1) < 2^32).
uint32_t CSA_Adler32(uint32_t adler, const uint8_t *buf, uint64_t len)
{
    uint32_t a = adler & 0xFFFFu, b = (adler >> 16) & 0xFFFFu;
    while (len) {
        uint32_t n = len > 5552 ? 5552u : (uint32_t)len;
        len -= n;
        while (n--) { a += *buf++; b += a; }
        a %= kAdlerBase;
        b %= kAdlerBase;
    }
    return a | (b << 16);
}

int CSAMI_Adler32Device(uint32_t adler, const void *device_ptr, uint64_t len, uint32_t *out)
{
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;
    std::vector<AdlerPieceH> pieces;
    for (uint64_t o = 0; o < len; o += kAdlerPiece)
        pieces.push_back(AdlerPieceH{(const uint8_t *)device_ptr + o, (uint32_t)std::min<uint64_t>(kAdlerPiece, len - o), 0});
    std::vector<AdlerSumsH> sums(pieces.size());
    if (!pieces.empty()) {
        void *dp = nullptr, *ds = nullptr;
        bool ok = HIP_OK(hipMalloc(&dp, pieces.size() * sizeof(AdlerPieceH))) && HIP_OK(hipMalloc(&ds, pieces.size() * sizeof(AdlerSumsH)));
        ok = ok && HIP_OK(hipMemcpy(dp, pieces.data(), pieces.size() * sizeof(AdlerPieceH), hipMemcpyHostToDevice));
        if (ok) {
            launch_adler_pieces(dp, ds, (uint32_t)pieces.size(), nullptr);
            ok = HIP_OK(hipGetLastError()) && HIP_OK(hipMemcpy(sums.data(), ds, pieces.size() * sizeof(AdlerSumsH), hipMemcpyDeviceToHost));
        }
        if (dp) (void)hipFree(dp);
        if (ds) (void)hipFree(ds);
        if (!ok) return CSCMI_DEVICE_ERROR;
    }
    for (size_t i = 0; i < pieces.size(); i++) adler_fold(adler, sums[i].a, sums[i].b, pieces[i].len);
    *out = adler;
    return 0;
}

// decimal_time, csa_common.cpp:3-25: a calendar in which every 4th year from 1972 is a leap year
// (exact for 1970..2099, the range the reference documents).
int64_t CSA_DecimalTime(int64_t tt)
{
    if (tt == -1) tt = 0;
    int64_t t = tt;
    const int64_t second = t % 60, minute = t / 60 % 60, hour = t / 3600 % 24;
    int64_t days = t / 86400;
    int64_t year = 1970 + 4 * (days / 1461);
    days %= 1461;
    static const int ylen[4] = {365, 365, 366, 365};
    int yi = 0;
    while (days >= ylen[yi]) { days -= ylen[yi]; yi++; year++; }
    static const int mlen[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    int month = 0;
    for (;; month++) {
        int ml = mlen[month] + ((month == 1 && yi == 2) ? 1 : 0);
        if (days < ml) break;
        days -= ml;
    }
    return year * 10000000000LL + (int64_t)(month + 1) * 100000000 + (days + 1) * 1000000 + hour * 10000 + minute * 100 + second;
}

// unix_time, csa_common.cpp:27-39
int64_t CSA_UnixTime(int64_t date)
{
    if (date <= 0) return -1;
    static const int before[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    const int64_t year = date / 10000000000LL % 10000;
    const int64_t month = (date / 100000000 % 100 - 1) % 12;
    const int64_t day = date / 1000000 % 100, hour = date / 10000 % 100, minute = date / 100 % 100, second = date % 100;
    int64_t days = day - 1 + before[month] + ((year % 4 == 0 && month > 1) ? 1 : 0) + ((year - 1970) * 1461 + 1) / 4;
    return days * 86400 + hour * 3600 + minute * 60 + second;
}

}   // extern "C"

// ---- pieces of Add shared by the one-process path and the sharded (one process per GPU) path ----
namespace {

struct AddPlan {
    CSAOptions o;
    Index index;
    std::vector<Task> tasks;
};

int plan_add(AddPlan &P, const char *const *filenames, int nfilenames, const CSAOptions *opt)
{
    if (opt) P.o = *opt; else CSA_OptionsInit(&P.o);
    if (P.o.split_count <= 0) P.o.split_count = 1;                           // csarc.cpp:199-200
    Selection sel = make_selection(filenames, nfilenames);
    for (const std::string &f : sel.names) scan_path(P.index, sel, f, P.o.recurse != 0);
    P.tasks = plan_tasks(P.index, P.o.split_count);
    if (P.o.task_bytes) P.tasks = cap_tasks(P.tasks, P.o.task_bytes);
    // the index stores a file's fragment count in one signed byte (csa_indexpack.cpp:84,105): more than 127 would write an
    // archive nobody can read back -- refuse before anything is written
    std::map<const void *, uint32_t> nfr;
    for (const Task &t : P.tasks)
        for (const FilePiece &f : t.files)
            if (++nfr[(const void *)&*f.it] > kMaxFrags) {
                fprintf(stderr, "csa-mi355x: %s would be stored as more than %u fragments (split_count / task_bytes too fine); nothing written\n",
                        f.it->first.c_str(), kMaxFrags);
                return CSA_TOO_MANY_FRAGMENTS;
            }
    return 0;
}

int open_archive(const char *arcname, const CSAOptions &o, int &fd)
{
    struct stat sb;
    if (!o.overwrite && stat(arcname, &sb) == 0) {                           // csarc.cpp:474-483
        fprintf(stderr, "Archive %s already exists, use -f to force overwrite\n", arcname);
        return 1;
    }
    fd = open(arcname, O_RDWR | O_CREAT, 0666);                              // OutputFile::open: "rb+" else "wb+"
    if (fd < 0) { perror(arcname); return WRITE_ERROR; }
    if (ftruncate(fd, (off_t)kHeaderSize) != 0) { perror("ftruncate"); close(fd); fd = -1; return WRITE_ERROR; }   // csarc.cpp:559-564
    return 0;
}

// one finished task -> archive blocks at the end of the archive (csa_io.h:559-573; id == task index, csarc.cpp:393-394)
int append_task(int fd, uint64_t &arc_end, BlockIndex &abindex, size_t id, const uint64_t *sizes, size_t nsizes,
                const uint8_t *data, uint64_t len)
{
    std::vector<Extent> &ext = abindex[id];
    uint64_t pos = arc_end;
    for (size_t i = 0; i < nsizes; i++) { ext.push_back(Extent{pos, sizes[i]}); pos += sizes[i]; }
    if (pos - arc_end != len) return WRITE_ERROR;
    if (!write_all(fd, data, len, arc_end)) return WRITE_ERROR;
    arc_end = pos;
    return 0;
}

// fragments, index stream, header: csarc.cpp:371-386, 219-288
int finish_archive(int fd, uint64_t arc_end, AddPlan &P, const BlockIndex &abindex, const char *arcname, CSAStats *st)
{
    // fragments in task-id order, files in task order: what compress_mt records with one worker
    for (size_t ti = 0; ti < P.tasks.size(); ti++)
        for (const FilePiece &f : P.tasks[ti].files)
            f.it->second.frags.push_back(CSAFrag{(uint32_t)ti, f.checksum, f.posblock, f.size, f.off});
    std::vector<uint8_t> raw = pack_index(P.index, abindex, arcname), comp;
    int rc = encode_index(raw, comp);
    if (rc == 0 && !write_all(fd, comp.data(), comp.size(), arc_end)) rc = WRITE_ERROR;
    if (rc) return rc;
    uint8_t hdr[kHeaderSize];                                                // csarc.cpp:268-287
    hdr[0] = 'C'; hdr[1] = 'S'; hdr[2] = 'A'; hdr[7] = '1';
    store_le(hdr + 3, kMagicNum, 4);
    store_le(hdr + 8, arc_end, 8);
    store_le(hdr + 16, comp.size(), 4);
    store_le(hdr + 20, raw.size(), 4);
    if (!write_all(fd, hdr, kHeaderSize, 0)) return WRITE_ERROR;
    if (st) {
        st->index_raw_size = raw.size();
        st->index_compressed_size = comp.size();
        st->archive_bytes = arc_end + comp.size();
        st->n_entries = (uint32_t)P.index.size();
        st->n_tasks = (uint32_t)P.tasks.size();
    }
    return 0;
}

// shard blob: what one rank hands to the rank that writes the archive.  All fields little-endian.
//   u32 'CSAS' | u32 tasks in the whole plan | u32 tasks in this blob | u32 hash of the task -> rank deal this rank computed
//   per task: u32 id | u32 nfiles | nfiles x { u32 adler32, u64 posblock, u64 size } | u32 nblocks | nblocks x u64 size | u64 len | bytes
constexpr uint32_t kShardMagic = 0x53415343u;

}   // namespace

extern "C" {

int CSA_Add(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *opt, CSAStats *st)
{
    double t0 = now_s();
    if (st) memset(st, 0, sizeof(*st));
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;
    AddPlan P;
    {
        CSAOptions o;
        if (opt) o = *opt; else CSA_OptionsInit(&o);
        struct stat sb;
        if (!o.overwrite && stat(arcname, &sb) == 0) {                       // csarc.cpp:474-483, before the scan
            fprintf(stderr, "Archive %s already exists, use -f to force overwrite\n", arcname);
            return 1;
        }
    }
    int rc = plan_add(P, filenames, nfilenames, opt);
    if (rc) return rc;
    int fd = -1;
    rc = open_archive(arcname, P.o, fd);
    if (rc) return rc;
    uint64_t arc_end = kHeaderSize;
    BlockIndex abindex;
    rc = encode_tasks(P.tasks, [&](size_t id, BlockSink &k) {
        return append_task(fd, arc_end, abindex, id, k.sizes.data(), k.sizes.size(), k.data.data(), k.data.size());
    }, P.o, st);
    if (rc == 0) rc = finish_archive(fd, arc_end, P, abindex, arcname, st);
    close(fd);
    if (st) st->seconds_total = now_s() - t0;
    return rc;
}

// ---- sharded Add: one process per GPU ----
// Which rank encodes which task.  The reference's workers take the next task of the size-sorted list whenever one of them is free
// (csarc.cpp:361-398) -- load-aware by construction.  Ranks do not talk while they encode, so every rank computes the same deal up
// front: longest-processing-time-first over an estimated cost = bytes x a factor for the kind of data (a look at the first 8 KiB
// of the task's first file, in the spirit of Analyzer::Analyze csc_analyzer.cpp:184-239: the level-3 kernels encode English-text-like
// data ~2.2 x faster than binary / delta data), each task to the rank with the least cost so far (ties: lowest rank).  The deal is
// a schedule, not data: the archive is the same whatever it is (tasks are appended in id order).  CSA_DEAL=mod restores round 3's
// `task i -> rank i mod world`.
static double task_cost_factor(const Task &t, int level)
{
    if (level != 3) return 1.0;                     // (the factors are level-3 measurements: other levels deal by size alone)
    for (const FilePiece &f : t.files) {
        if (!f.size) continue;
        int fd = open(f.path.c_str(), O_RDONLY);
        if (fd < 0) continue;
        uint8_t buf[8192];
        ssize_t n = pread(fd, buf, sizeof(buf), (off_t)f.off);
        close(fd);
        if (n < 512) return 1.6;
        uint32_t hi = 0, alpha = 0, sp = 0;
        for (ssize_t i = 0; i < n; i++) { hi += buf[i] >= 128; alpha += buf[i] >= 'a' && buf[i] <= 'z'; sp += buf[i] == ' ' || buf[i] == '\n'; }
        const bool text = hi < (uint32_t)n / 8 && alpha > (uint32_t)n / 3 && sp > (uint32_t)n / 16;
        return text ? 1.0 : 2.2;
    }
    return 1.0;
}
static void deal_tasks(const std::vector<Task> &tasks, int world, int level, std::vector<uint32_t> &rank_of, std::vector<double> &cost)
{
    const size_t nt = tasks.size();
    rank_of.assign(nt, 0); cost.assign(nt, 0.0);
    const char *mode = getenv("CSA_DEAL");
    const bool mod = mode && !strcmp(mode, "mod");
    for (size_t i = 0; i < nt; i++) cost[i] = (double)tasks[i].total * (mod ? 1.0 : task_cost_factor(tasks[i], level));
    if (mod || world <= 1) { for (size_t i = 0; i < nt; i++) rank_of[i] = (uint32_t)(i % (size_t)std::max(1, world)); return; }
    std::vector<uint32_t> order(nt);
    for (size_t i = 0; i < nt; i++) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    std::vector<double> load((size_t)world, 0.0);
    for (uint32_t id : order) {
        int best = 0;
        for (int r = 1; r < world; r++) if (load[r] < load[best]) best = r;
        rank_of[id] = (uint32_t)best;
        load[best] += cost[id];
    }
}

int CSAMI_PlanShards(const char *const *filenames, int nfilenames, const CSAOptions *opt, int world, uint32_t *rank_of, double *cost, uint32_t cap)
{
    if (world <= 0) return -1;
    AddPlan P;
    if (int prc = plan_add(P, filenames, nfilenames, opt)) return prc < 0 ? prc : -prc;
    std::vector<uint32_t> ro;
    std::vector<double> co;
    deal_tasks(P.tasks, world, opt ? opt->level : 2, ro, co);
    for (size_t i = 0; i < ro.size() && i < cap; i++) { if (rank_of) rank_of[i] = ro[i]; if (cost) cost[i] = co[i]; }
    return (int)ro.size();
}

int CSAMI_AddShardEncode(const char *const *filenames, int nfilenames, const CSAOptions *opt, int rank, int world,
                         uint8_t **blob, uint64_t *blob_len, CSAStats *st)
{
    double t0 = now_s();
    if (st) memset(st, 0, sizeof(*st));
    if (!blob || !blob_len || world <= 0 || rank < 0 || rank >= world) return -1;
    *blob = nullptr; *blob_len = 0;
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;
    AddPlan P;
    if (int prc = plan_add(P, filenames, nfilenames, opt)) return prc;
    std::vector<Task> mine;
    std::vector<uint32_t> ids;
    std::vector<uint32_t> rank_of;
    std::vector<double> cost;
    deal_tasks(P.tasks, world, opt ? opt->level : 2, rank_of, cost);
    for (size_t i = 0; i < P.tasks.size(); i++)
        if ((int)rank_of[i] == rank) { mine.push_back(P.tasks[i]); ids.push_back((uint32_t)i); }   // (id order within a rank = dispatch order, csarc.cpp:355)
    std::vector<uint8_t> out;
    put_le(out, kShardMagic, 4);
    put_le(out, P.tasks.size(), 4);
    put_le(out, mine.size(), 4);
    {
        // every rank computes the deal by itself (first 8 KiB of files, CSA_DEAL): a hash of it travels with the blob so that ranks that
        // disagreed are told apart from a damaged blob at assembly
        uint32_t h = 2166136261u ^ (uint32_t)world;
        for (uint32_t r : rank_of) h = (h ^ r) * 16777619u;
        put_le(out, h, 4);
    }
    int rc = encode_tasks(mine, [&](size_t k, BlockSink &s) {
        put_le(out, ids[k], 4);
        put_le(out, mine[k].files.size(), 4);
        for (const FilePiece &f : mine[k].files) { put_le(out, f.checksum, 4); put_le(out, f.posblock, 8); put_le(out, f.size, 8); }
        put_le(out, s.sizes.size(), 4);
        for (uint64_t z : s.sizes) put_le(out, z, 8);
        put_le(out, s.data.size(), 8);
        out.insert(out.end(), s.data.begin(), s.data.end());
        return 0;
    }, P.o, st);
    if (rc) return rc;
    uint8_t *b = (uint8_t *)malloc(out.size());
    if (!b) return -1;
    memcpy(b, out.data(), out.size());
    *blob = b; *blob_len = out.size();
    if (st) { st->n_tasks = (uint32_t)mine.size(); st->seconds_total = now_s() - t0; }
    return 0;
}

void CSAMI_FreeBlob(uint8_t *blob) { free(blob); }

int CSAMI_AddShardAssemble(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *opt,
                           const uint8_t *const *blobs, const uint64_t *blob_lens, int nblobs, CSAStats *st)
{
    double t0 = now_s();
    if (st) memset(st, 0, sizeof(*st));
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;              // the index stream is encoded on the GPU
    AddPlan P;
    if (int prc = plan_add(P, filenames, nfilenames, opt)) return prc;
    struct Part { const uint8_t *files, *sizes, *data; uint32_t nfiles, nblocks; uint64_t len; };
    std::vector<Part> parts(P.tasks.size(), Part{nullptr, nullptr, nullptr, 0, 0, 0});
    for (int b = 0; b < nblobs; b++) {
        const uint8_t *p = blobs[b], *end = p + blob_lens[b];
        if (blob_lens[b] < 16 || load_le(p, 4) != kShardMagic || load_le(p + 4, 4) != P.tasks.size()) {
            fprintf(stderr, "csa-mi355x: shard %d does not belong to this plan (input changed between ranks?)\n", b);
            return -1;
        }
        if (b > 0 && load_le(p + 12, 4) != load_le(blobs[0] + 12, 4)) {
            fprintf(stderr, "csa-mi355x: shards 0 and %d were encoded under different task -> rank deals (the ranks disagreed: CSA_DEAL, or a file's first block, differs between them)\n", b);
            return -3;
        }
        uint32_t n = (uint32_t)load_le(p + 8, 4);
        p += 16;
        for (uint32_t k = 0; k < n; k++) {
            if (end - p < 8) return -1;
            uint32_t id = (uint32_t)load_le(p, 4), nf = (uint32_t)load_le(p + 4, 4);
            p += 8;
            if (id >= parts.size() || parts[id].data || nf != P.tasks[id].files.size() || (uint64_t)(end - p) < (uint64_t)nf * 20 + 4) return -1;
            Part &q = parts[id];
            q.files = p; q.nfiles = nf; p += (uint64_t)nf * 20;
            q.nblocks = (uint32_t)load_le(p, 4); p += 4;
            if ((uint64_t)(end - p) < (uint64_t)q.nblocks * 8 + 8) return -1;
            q.sizes = p; p += (uint64_t)q.nblocks * 8;
            q.len = load_le(p, 8); p += 8;
            if ((uint64_t)(end - p) < q.len) return -1;
            q.data = p; p += q.len;
        }
    }
    for (size_t i = 0; i < parts.size(); i++)
        if (!parts[i].data) { fprintf(stderr, "csa-mi355x: task %zu missing from the shards\n", i); return -1; }
    int fd = -1;
    int rc = open_archive(arcname, P.o, fd);
    if (rc) return rc;
    uint64_t arc_end = kHeaderSize;
    BlockIndex abindex;
    std::vector<uint64_t> sizes;
    for (size_t i = 0; rc == 0 && i < parts.size(); i++) {
        const Part &q = parts[i];
        for (uint32_t f = 0; f < q.nfiles; f++) {
            FilePiece &fp = P.tasks[i].files[f];
            fp.checksum = (uint32_t)load_le(q.files + f * 20, 4);
            fp.posblock = load_le(q.files + f * 20 + 4, 8);
            fp.size = load_le(q.files + f * 20 + 12, 8);
        }
        sizes.resize(q.nblocks);
        for (uint32_t k = 0; k < q.nblocks; k++) sizes[k] = load_le(q.sizes + k * 8, 8);
        rc = append_task(fd, arc_end, abindex, i, sizes.data(), sizes.size(), q.data, q.len);
        if (st) st->n_blocks += q.nblocks;
    }
    if (rc == 0) rc = finish_archive(fd, arc_end, P, abindex, arcname, st);
    close(fd);
    if (st) st->seconds_total = now_s() - t0;
    return rc;
}

int CSA_Extract(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *opt, CSAStats *st)
{
    CSAOptions o;
    if (opt) o = *opt; else CSA_OptionsInit(&o);
    if (st) memset(st, 0, sizeof(*st));
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;
    return read_archive(arcname, make_selection(filenames, nfilenames), o, true, st);
}

int CSA_Test(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *opt, CSAStats *st)
{
    CSAOptions o;
    if (opt) o = *opt; else CSA_OptionsInit(&o);
    if (st) memset(st, 0, sizeof(*st));
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;
    int rc = read_archive(arcname, make_selection(filenames, nfilenames), o, false, st);
    return rc == 1 ? -1 : rc;                                                // csarc.cpp:669-670
}

int CSA_List(const char *arcname, const char *const *filenames, int nfilenames, CSAListFn fn, void *ctx)
{
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;
    Index index;
    BlockIndex abindex;
    if (read_index(arcname, index, abindex) != 0) return -1;
    Selection sel = make_selection(filenames, nfilenames);
    for (const auto &kv : index) {
        if (!sel.names.empty() && !sel.has(kv.first.c_str())) continue;
        if (fn) fn(ctx, kv.first.c_str(), kv.second.esize, kv.second.edate, kv.second.eattr, (int)kv.second.frags.size(),
                   kv.second.frags.empty() ? nullptr : kv.second.frags.data());
    }
    return 0;
}

// the plan CSA_Add would execute, without executing it (host only): number of tasks and the largest fragment count of any file
int CSAMI_PlanInfo(const char *const *filenames, int nfilenames, const CSAOptions *opt, uint32_t *n_tasks, uint32_t *max_frags)
{
    AddPlan P;
    int rc = plan_add(P, filenames, nfilenames, opt);
    std::map<const void *, uint32_t> nfr;
    uint32_t mx = 0;
    for (const Task &t : P.tasks)
        for (const FilePiece &f : t.files) mx = std::max(mx, ++nfr[(const void *)&*f.it]);
    if (n_tasks) *n_tasks = (uint32_t)P.tasks.size();
    if (max_frags) *max_frags = mx;
    return rc;
}

int64_t CSA_IndexRoundTrip(const uint8_t *buf, uint64_t size, uint8_t *out, uint64_t cap)
{
    Index index;
    BlockIndex abindex;
    if (!unpack_index(index, abindex, buf, size)) return -1;
    for (const auto &kv : index) if (!name_is_safe(kv.first)) return CSA_UNSAFE_NAME;
    for (const auto &kv : index) if (kv.second.frags.size() > kMaxFrags) return -1;
    std::vector<uint8_t> raw = pack_index(index, abindex, "");
    if (out && cap) memcpy(out, raw.data(), (size_t)std::min<uint64_t>(cap, raw.size()));
    return (int64_t)raw.size();
}

int64_t CSA_ReadIndex(const char *arcname, uint8_t *buf, uint64_t cap)
{
    if (CSCMI_DeviceCheck() != 0) return CSCMI_DEVICE_ERROR;
    Index index;
    BlockIndex abindex;
    std::vector<uint8_t> raw;
    if (read_index(arcname, index, abindex, &raw) != 0) return -1;
    if (buf) memcpy(buf, raw.data(), (size_t)std::min<uint64_t>(cap, raw.size()));
    return (int64_t)raw.size();
}

}   // extern "C"
