/* oracle/zero_heap.cpp -- TEST INFRASTRUCTURE, linked only into oracle/_ref/csarc_ref.
 *
 * The reference archiver compresses two kinds of uninitialised memory:
 *   - libcsc's tables come from malloc() through csc_default_alloc.cpp:5-8 when the caller passes
 *     a NULL ISzAlloc (csa_worker.cpp:36, csarc.cpp:252), and the encoder reads some of them
 *     before writing;
 *   - PackIndex() sizes its buffer 4 + arcname.size() bytes too large per task
 *     (csa_indexpack.cpp:129-134 counts a field ArchiveBlocksToBuf :136-150 no longer writes) and
 *     the unwritten tail of that `new char[]` is compressed into the index.
 * Both make the reference's output depend on heap history.  The deterministic reading of the
 * reference -- the one this build reproduces -- is "every allocation is zero-filled".  This file
 * gives the reference binary exactly that, without touching its sources: the link recipe in
 * oracle/Makefile wraps malloc (-Wl,--wrap=malloc) and this TU replaces operator new / new[].
 */
#include <stdlib.h>
#include <new>

extern "C" void *__wrap_malloc(size_t n) { return calloc(1, n ? n : 1); }

void *operator new(size_t n) { void *p = calloc(1, n ? n : 1); if (!p) throw std::bad_alloc(); return p; }
void *operator new[](size_t n) { void *p = calloc(1, n ? n : 1); if (!p) throw std::bad_alloc(); return p; }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
