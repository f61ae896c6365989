// abi.cpp -- version + thread-local error string of libe2eslam_hip.so
#include <stdarg.h>
#include <stdio.h>

#include "../../include/e2eslam.h"

static thread_local char g_err[512] = "";

void e2e_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int e2e_version(void) { return 100; }  /* 0.1.0 */
const char* e2e_last_error(void) { return g_err; }
}
