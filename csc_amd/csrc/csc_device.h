// csc_device.h -- data laid out in HBM for one libcsc stream, shared by the host
// orchestration (csc_host.cpp) and the gfx950 kernels (csc_kernels.hip).
//
// One `EncState` per CSCEncHandle.  Everything the serial protocol of the
// reference keeps in its C++ objects (window, hash tables / binary tree,
// probability tables, range+bit coder buffers: SURVEY.md section 8a rows a10-a22)
// lives in device memory and never leaves it; the host only moves the 2 MiB
// input chunk in and the finished RC/BC blocks out.
#pragma once
#include <stdint.h>

namespace cscmi {

constexpr uint32_t kKB = 1024u;
constexpr uint32_t kMB = 1048576u;
constexpr uint32_t kMinBlock = 8u * kKB;      // csc_typedef.h:9 MinBlockSize
constexpr uint32_t kHT2Size = 16u * kKB;      // csc_mf.h:18
constexpr uint32_t kHT3Size = 64u * kKB;      // csc_mf.h:17
constexpr uint32_t kMFCandLimit = 32;         // csc_mf.h:34
constexpr uint32_t kAPLimit = 2048;           // csc_lz.h:43
constexpr uint32_t kMaxBlocksPerChunk = 2048; // 16 MiB raw_blocksize bound / 8 KiB
constexpr uint32_t kBtUndoBytes = 256 * 36 * 8; // EncState::bt_undo: record ring x undo entries a position x {slot, old word} (csc_kernels_bt.inc)

// block types, csc_typedef.h:20-40
enum : uint32_t {
    DT_NORMAL = 1, DT_ENGTXT = 2, DT_EXE = 3, DT_FAST = 4, DT_NO_LZ = 5,
    DT_ENTROPY = 7, DT_BAD = 8, SIG_EOF = 9, DT_DLT = 0x10, DT_SKIP = 0x1E
};

// offsets of the small adaptive-probability tables inside EncState::probs
// (kept in LDS while a kernel runs).  csc_model.h:84-122
enum : uint32_t {
    P_STATE = 0,                       // [64*3]
    P_REPDIST = P_STATE + 192,         // [64*3] used (reference declares 64*4)
    P_DIST = P_REPDIST + 192,          // [8 + 16*2 + 32*4]
    P_LEN_SLOT = P_DIST + 168,         // [2]
    P_LEN_X1 = P_LEN_SLOT + 2,         // [8]
    P_LEN_X2 = P_LEN_X1 + 8,           // [8]
    P_LEN_X3 = P_LEN_X2 + 8,           // [128]
    P_DIST_EXTRA = P_LEN_X3 + 128,     // [29*16]
    P_LONGLEN = P_DIST_EXTRA + 464,
    P_RLE_FLAG = P_LONGLEN + 1,
    P_COUNT = P_RLE_FLAG + 1
};

enum : uint32_t { ERR_NONE = 0, ERR_ARENA_FULL = 1, ERR_BAD_TYPE = 2, ERR_PAIR_STALL = 3 };
constexpr uint32_t kSelfSegment = 0x80000000u;   // k_encode_runs*: nruns = kSelfSegment | chunk bytes -- the kernel walks the chunk's blocks itself (enc_compress_chunk)

// one finished coder block in the output arena: 16-byte header, payload padded to 16
struct ArenaRec {
    uint32_t kind;   // 1 = range-coder block, 0 = bit-coder block (csc_memio.cpp:86-88)
    uint32_t size;   // payload bytes
    uint32_t pad[2];
};

// a run of equal-type 8 KiB blocks, csc_encoder_main.cpp:35-83
struct RunDesc {
    uint32_t type;
    uint32_t offset;   // into the chunk buffer
    uint32_t size;
    uint32_t tail;     // EncodeInt written after the run: 0 between runs, 1 + coder flush at chunk end
};

// per-8 KiB-block analyzer verdict, csc_analyzer.cpp:184-239 + :166-182
struct BlockInfo {
    uint32_t type;        // Analyze() result (DT_SKIP for < 512 B)
    uint32_t bpb;         // entropy / size (undefined for DT_SKIP)
    uint32_t dlt_bpb[5];  // GetDltBpb for channel k: filled for the verdict's channel, or all 5 for DT_SKIP blocks
    uint32_t pad;
};

struct KernelStats {          // counters for bench / DESIGN.md (SURVEY section 5 tracing row)
    unsigned long long find_match_calls;
    unsigned long long slide_positions;
    unsigned long long bt_steps;
    unsigned long long literals;
    unsigned long long matches;
    unsigned long long tm[16];   // cycle accumulators, filled only by -DCSCMI_TIMERS development builds
    unsigned long long trace[64][12];   // event times of one DP window's nodes (-DCSCMI_TIMERS builds, tools/gpu_trace.py)
};

struct EncState {
    // ---- configuration (written once by the host) ----
    uint32_t wnd_size, vld_rge, bsize, raw_blocksize;
    uint32_t ht_bits, ht_width, bt_bits, bt_size;
    uint32_t lz_mode, lz_good_len, lz_bt_cyc, lz_ht_cyc;
    uint32_t arena_cap, filt_flags;   // filt_flags: bit 0 DLTFilter, 1 TXTFilter, 2 EXEFilter (csc_common.h:52-56)
    uint64_t mf_size;          // words in mfbuf (ht2|ht3|ht6|bt_head|bt_nodes)

    // ---- HBM arrays ----
    uint8_t *wnd;              // wnd_size + slack, circular dictionary
    uint32_t *mfbuf, *ht2, *ht3, *ht6, *bt_head, *bt_nodes;
    uint32_t *p_lit;           // [256*256] order-1 literal probabilities
    uint32_t *p_delta;         // [256*256] RLE/delta literal probabilities (lazily initialised)
    uint32_t *ap_rep;          // [(kAPLimit+1)*4] rep distances of the parser's DP nodes
    uint8_t *rc_buf, *bc_buf;  // the two persistent csc_blocksize coder buffers
    uint8_t *inbuf;            // raw_blocksize + slack: the chunk being encoded (filters work in place)
    uint8_t *swapbuf;          // 4*raw_blocksize + slack: filter scratch
    uint8_t *bt_undo;          // undo log of the binary-tree inserter (csc_kernels_bt.inc: kBtRec x kBtULog x 8 bytes); null without a tree
    uint8_t *arena;            // finished blocks of the current launch (ArenaRec + payload)*
    const uint16_t *trie_next; // [300*26] word trie, csc_filters.cpp:87-111
    const uint8_t *trie_sym;   // [300]
    BlockInfo *binfo;          // [kMaxBlocksPerChunk]
    uint32_t *dup_flags;       // [kMaxBlocksPerChunk] IsDuplicateBlock results

    // ---- dynamic scalar state carried between launches ----
    uint32_t pos, bt_pos, wnd_curpos, p_delta_ready;
    uint32_t rep_dist[4];
    uint32_t state, ctx, lp_rebuild_int, rc_range;
    uint64_t rc_low;
    uint32_t rc_cache, rc_cachesize, rc_size, bc_size;
    uint32_t bc_curbits, bc_curval, arena_used, error;
    uint32_t dict_result, pad1;
    uint32_t len_price[32];
    uint32_t probs[P_COUNT + 4];
    KernelStats stats;
    uint32_t pf_sink[64];      // keeps the cache-warming loads of k_encode_runs alive
};

// ---------------------------------------------------------------------------------------------
// decoder (csc_dec_kernels.hip / csc_dec_device.cpp)
constexpr uint32_t kDecUndoCap = 32768;   // probability updates journalled per packet (RLE run lengths can take ~14.6 K long-length bits)
enum : uint32_t { DEC_RUNNING = 0, DEC_DONE = 1, DEC_NEED_RC = 2, DEC_NEED_BC = 3, DEC_ERR_DECODE = 4, DEC_ERR_MINUS1 = 5 };
enum : uint32_t { DEC_PH_PRIME0 = 0, DEC_PH_TYPE, DEC_PH_SIZE, DEC_PH_LZ, DEC_PH_RAW, DEC_PH_POST, DEC_PH_TAIL, DEC_PH_PRIME };

struct DecState {
    // configuration
    uint32_t wnd_size, bsize, raw_blocksize, qslots;
    uint8_t *wnd;              // dict_size + slack
    uint32_t *p_lit, *p_delta; // ONE allocation of 2 x 64 Ki words: p_delta = p_lit + 65536
    uint8_t *out, *swap;       // decoded run (raw_blocksize + slack), filter scratch
    uint8_t *q[2];             // [1] RC / [0] BC block rings: qslots x bsize bytes
    uint32_t *qsize[2];        // payload size per ring slot
    uint32_t *undo_addr, *undo_val;
    const uint8_t *words;      // 122 x 8 bytes: the dictionary filter's word list
    // written by the host before every launch
    uint32_t avail[2];         // blocks uploaded so far per kind
    // stream state carried between launches
    uint32_t taken[2], rd[2], fill[2];
    uint32_t range, code, bc_bits, bc_val;
    uint32_t state, ctx, rep[4], wnd_pos, p_delta_ready;
    uint32_t phase, type, run_size, i, copied, copied_from;
    uint32_t status, out_size;
    uint64_t consumed;
    uint32_t probs[P_COUNT + 4];
    uint64_t dbg[16];          // development-only section timers (-DCSCMI_TIMERS builds); zero otherwise
};

}  // namespace cscmi
