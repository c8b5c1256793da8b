/*
 * csa_mi355x.h -- C ABI of the `.csa` container layer of libcsc_mi355x.so.
 *
 * The reference's archiver (src/archiver) is a command-line program, not a library: its interface
 * is the four operations of `class CSArc` (csarc.cpp:37-70) and their options (ParseArg,
 * csarc.cpp:137-208).  This header gives each of them a C entry point with the same meaning, the
 * same return value and -- for `CSA_Add` -- the same bytes on disk as `csarc a -t1`:
 *
 *   reference                                    here
 *   CSArc::Add      csarc.cpp:472-575            CSA_Add
 *   CSArc::Extract  csarc.cpp:600-650            CSA_Extract
 *   CSArc::Test     csarc.cpp:667-700            CSA_Test
 *   CSArc::List     csarc.cpp:652-665            CSA_List
 *   adler32()       csa_adler32.cpp:63-129       CSA_Adler32 (host), CSAMI_Adler32Device (HIP kernel)
 *   PackIndex / UnpackIndex csa_indexpack.cpp:166-211   inside CSA_Add / the readers; CSA_ReadIndex exposes the raw bytes
 *
 * What runs where: the task streams and the index stream are libcsc streams produced by the HIP
 * encoder behind CSCEnc_* (many task streams advance together, one workgroup per stream); the
 * per-fragment adler32 is a HIP reduction over the chunk that is already in HBM for the encoder;
 * decoding uses the HIP decoder behind CSCDec_*.  Directory scan, task split, block table, index
 * packing and file I/O are host code, as in the reference.  No CPU codec exists in this library.
 */
#ifndef CSA_MI355X_H_
#define CSA_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Options of `csarc` (ParseArg, csarc.cpp:137-208).  CSA_OptionsInit sets the reference defaults. */
typedef struct CSAOptions {
    int level;              /* -m#      1..5, default 2                         csarc.cpp:141,150-156 */
    uint32_t dict_size;     /* -d##[k|m] 32 KiB..1 GiB, default 32000000        csarc.cpp:140,157-168 */
    int recurse;            /* -r                                               csarc.cpp:169-170 */
    int overwrite;          /* -f                                               csarc.cpp:171-172 */
    int verbose;            /* -v (List: report fragments)                      csarc.cpp:173-174 */
    int mt_count;           /* -t#  accepted for parity with the reference and otherwise unused: Add always
                               writes the single-worker (task-id order) layout, Extract/Test decode
                               `device_streams` tasks per kernel launch                                  */
    int split_count;        /* -p##  single-file split, default 1               csarc.cpp:190-191 */
    const char *to_dir;     /* -o dir, default "./"                             csarc.cpp:147,183-189 */
    /* --- not in the reference --- */
    int device_streams;     /* task streams advanced per kernel launch; 0 = as many as fit the HBM budget,
                               at most 1024 (Add) / 256 = one per CU (Extract, Test)                    */
    uint64_t hbm_budget;    /* Add: bytes of HBM the concurrent task encoders may use; 0 = 3/4 of
                               the free device memory                                                */
    uint64_t task_bytes;    /* Add: 0 = the reference's task split (archive identical to csarc's).
                               > 0: tasks are additionally cut so that none exceeds this many bytes
                               (files are split into fragments where needed).  The archive is then NOT
                               the one `csarc a` would write, but it is a valid .csa that `csarc x/t/l`
                               read; it gives the GPU many independent streams on any input.  At most
                               127 fragments per file are produced (the index format's limit).          */
} CSAOptions;

typedef struct CSAStats {
    uint64_t raw_bytes;              /* bytes read from the input files */
    uint64_t archive_bytes;          /* final size of the archive */
    uint64_t index_raw_size;
    uint64_t index_compressed_size;
    uint32_t n_entries;              /* index entries (files and directories) */
    uint32_t n_tasks;
    uint32_t n_blocks;               /* archive blocks of all tasks */
    uint32_t verify_failures;        /* Extract/Test: fragments whose adler32 did not match */
    double seconds_total;
    double seconds_encode;           /* Add: inside the batched encoder calls; Extract/Test: decode wall time */
    uint32_t peak_streams;           /* Add: most task streams in flight at once */
    uint32_t reserved;
    double seconds_io;               /* Add: file reads + uploads + adler32 kernel on the calling thread -- a stream's FIRST chunk; the later
                                        chunks are read by a second thread during the encode calls (round 5) and do not count here */
    double seconds_setup;            /* Add: creating / flushing / destroying the task encoders */
} CSAStats;

typedef struct CSAFrag {             /* FileEntry::Frag, csa_typedef.h:18-24 */
    uint32_t bid;                    /* task (archive-blocks) id */
    uint32_t checksum;               /* adler32, seed 0 */
    uint64_t posblock;               /* offset inside the task's raw stream */
    uint64_t size;
    uint64_t posfile;                /* offset inside the file */
} CSAFrag;

/* One call per index entry in name order (what `csarc l` prints, csarc.cpp:656-663). */
typedef void (*CSAListFn)(void *ctx, const char *name, int64_t esize, int64_t edate, int64_t eattr,
                          int nfrags, const CSAFrag *frags);

void CSA_OptionsInit(CSAOptions *o);

/* Not in the reference: a file that the chosen split_count / task_bytes would cut into more than 127 fragments is refused
 * before anything is written (the index keeps the count in one signed byte, csa_indexpack.cpp:84,105 -- the reference CLI
 * writes such an archive and cannot read it back). */
#define CSA_MAX_FRAGMENTS 127
#define CSA_TOO_MANY_FRAGMENTS (-94)
/* Extract / Test: entries whose stored name is empty or has a `..` component are skipped with a message and make the call
 * return CSA_UNSAFE_NAME (after everything else was processed); the reference would follow them out of to_dir. */
#define CSA_UNSAFE_NAME (-93)

/* `csarc a [opts] arcname filenames...`.  Returns 0; 1 if the archive exists and !overwrite
 * (csarc.cpp:474-483); CSCMI_DEVICE_ERROR / READ_ERROR / WRITE_ERROR style negatives when the
 * encoder or the file system fails (the reference ignores those). */
int CSA_Add(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *o, CSAStats *st);

/* Sharded Add -- one process per GPU (the reference's counterpart is compress_mt, csarc.cpp:338-409,
 * whose worker threads each take the next task of the size-sorted list and hand their archive blocks to
 * one writer).  Every rank plans the same tasks from the same file names and makes the same deal (CSAMI_PlanShards: the next
 * task of the cost-sorted list to the least loaded rank -- what csarc.cpp:361-398 does dynamically); rank r encodes its tasks on ITS GPU and returns them as one
 * opaque blob (malloc'ed; release with CSAMI_FreeBlob).  The caller moves the blobs to the writing rank
 * (RCCL over xGMI in csc_amd/sharded.py -- the only exchange on this path) and calls CSAMI_AddShardAssemble
 * there with all of them: the archive is byte for byte the one CSA_Add / `csarc a -t1` writes.
 * Return values as CSA_Add; -1 for blobs that do not match the plan. */
int CSAMI_AddShardEncode(const char *const *filenames, int nfilenames, const CSAOptions *o, int rank, int world,
                         uint8_t **blob, uint64_t *blob_len, CSAStats *st);
void CSAMI_FreeBlob(uint8_t *blob);
/* The deal CSAMI_AddShardEncode uses (host only, no GPU): task i of the plan -> rank_of[i], with the cost estimate it was made from
 * (bytes x a data-kind factor; longest-processing-time-first, deterministic on every rank; CSA_DEAL=mod in the environment: i mod
 * world).  Fills at most `cap` entries; returns the number of tasks, or < 0. */
int CSAMI_PlanShards(const char *const *filenames, int nfilenames, const CSAOptions *o, int world, uint32_t *rank_of, double *cost, uint32_t cap);
int CSAMI_AddShardAssemble(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *o,
                           const uint8_t *const *blobs, const uint64_t *blob_lens, int nblobs, CSAStats *st);

/* `csarc x`: 0 ok, 1 bad header (csarc.cpp:602-603), -1 decode error (csarc.cpp:464-468), CSA_UNSAFE_NAME (above).
 * A failed adler32 is reported on stderr like the reference does and counted in st->verify_failures;
 * it does not change the return value (csa_io.h:331-332). */
int CSA_Extract(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *o, CSAStats *st);

/* `csarc t`: 0 ok, -1 bad header or decode error (csarc.cpp:669-670,697-700). */
int CSA_Test(const char *arcname, const char *const *filenames, int nfilenames, const CSAOptions *o, CSAStats *st);

/* `csarc l`: 0 ok, -1 bad header (csarc.cpp:654-655). */
int CSA_List(const char *arcname, const char *const *filenames, int nfilenames, CSAListFn fn, void *ctx);

/* The plan CSA_Add would execute for these names and options, without executing it (host only, no GPU): number of task
 * streams and the largest fragment count any file gets.  Returns what CSA_Add would return at that point (0 or
 * CSA_TOO_MANY_FRAGMENTS). */
int CSAMI_PlanInfo(const char *const *filenames, int nfilenames, const CSAOptions *o, uint32_t *n_tasks, uint32_t *max_frags);

/* Index pack / unpack round trip on a raw index buffer, for tests and fuzzing (no GPU involved): parses `buf` with the
 * bounds-checked reader and, if it parses, re-packs it.  Returns the re-packed size (<= cap bytes are copied to out, out
 * may be NULL), -1 if the buffer does not parse (truncated, counts beyond the buffer, negative fragment count),
 * CSA_UNSAFE_NAME if it parses but holds a name Extract would refuse. */
int64_t CSA_IndexRoundTrip(const uint8_t *buf, uint64_t size, uint8_t *out, uint64_t cap);

/* The raw (unpacked) index bytes of an archive, for tools and tests: returns the raw size, or -1.
 * Copies at most cap bytes into buf (buf may be NULL to query the size). */
int64_t CSA_ReadIndex(const char *arcname, uint8_t *buf, uint64_t cap);

/* adler32 of csa_adler32.cpp:63-129 (zlib's, with the running value passed in; the archiver seeds
 * fragments with 0).  Host version, and the HIP reduction over a device buffer. */
uint32_t CSA_Adler32(uint32_t adler, const uint8_t *buf, uint64_t len);
int CSAMI_Adler32Device(uint32_t adler, const void *device_ptr, uint64_t len, uint32_t *out);

/* YYYYMMDDHHMMSS <-> time_t, csa_common.cpp:3-39 */
int64_t CSA_DecimalTime(int64_t unix_seconds);
int64_t CSA_UnixTime(int64_t decimal_date);

#ifdef __cplusplus
}
#endif
#endif /* CSA_MI355X_H_ */
