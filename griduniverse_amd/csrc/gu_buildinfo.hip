// gu_buildinfo.hip -- which sources this libgu.so was built from.  The Makefile passes the sha256 (16 hex digits) of all
// library sources as GU_SOURCE_HASH and rebuilds this unit whenever one of them changes; the Python binding recomputes
// the hash from the files on disk and refuses a stale binary (the .so travels to the GPU box next to the sources).
#include <cstring>

#include "../../include/gu.h"

#ifndef GU_SOURCE_HASH
#define GU_SOURCE_HASH "unknown"
#endif

// The same hash as a marker in the file's bytes, so that the binding can read it WITHOUT mapping the library (a dlopen'ed
// image cannot be replaced in the process: glibc hands the old one back for the same path, and a rebuild would never be seen).
#ifdef GU_EXPERIMENTS
#define GU_BUILD_KIND "experiments"
#else
#define GU_BUILD_KIND "product"
#endif
extern "C" __attribute__((used, visibility("default"))) const char gu_build_marker[] = "GU_SRCHASH=" GU_SOURCE_HASH ";GU_BUILD=" GU_BUILD_KIND ";";

extern "C" int gu_source_hash(char *buf, size_t len)
{
    const size_t n = strlen(GU_SOURCE_HASH);
    if (buf && len) {
        const size_t m = n < len - 1 ? n : len - 1;
        memcpy(buf, GU_SOURCE_HASH, m);
        buf[m] = 0;
    }
    return (int)n;
}
