/*
 * msvc_compat.h -- force-included (-include) when the reference's OWN, unmodified Main.c and
 * comparator.c are compiled in place on Linux (oracle/Makefile, target ref_main).  The reference is a
 * Visual Studio project; these are the MSVC CRT names its two caller files use:
 *   errno_t, fopen_s   Main.c:45, comparator.c:30-31
 *   CLK_TCK            Main.c:57 (obsolete alias glibc no longer defines)
 * Test infrastructure only: nothing here is part of the library or of its headers.
 */
#ifndef VIT_TESTS_MSVC_COMPAT_H
#define VIT_TESTS_MSVC_COMPAT_H

#include <errno.h>
#include <stdio.h>
#include <time.h>

typedef int errno_t;

static inline errno_t fopen_s(FILE **f, const char *name, const char *mode)
{
    *f = fopen(name, mode);
    return *f ? 0 : errno;
}

#ifndef CLK_TCK
#define CLK_TCK CLOCKS_PER_SEC
#endif

#endif
