/* LD_PRELOAD helper (diagnostics only): a native backtrace of whoever ends the process abnormally, to stderr.
 *   - interposes abort(): std::terminate, failed assertions of C++ runtimes, explicit aborts (callers that go through the PLT);
 *   - SIGABRT / SIGSEGV / SIGBUS handlers for the rest (abrt_bt_install(): call it late through ctypes if another library
 *     installs handlers of its own at load time).
 * Run with LIBC_FATAL_STDERR_=1 so that glibc's own messages (heap corruption: it calls its internal abort) reach stderr too.
 * build: make -C tools/dbg ; run: LD_PRELOAD=tools/dbg/abrt_bt.so ABRT_BT_FILE=/tmp/bt.txt python -m pytest -p no:faulthandler ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
static int out_fd = 2;   /* ABRT_BT_FILE: a file of its own (pytest redirects fd 2 into a capture file while a test runs) */
static void dump(const char *m) {
    void *bt[96];
    if (write(out_fd, m, strlen(m)) < 0) return;
    int n = backtrace(bt, 96);
    backtrace_symbols_fd(bt, n, out_fd);
}
static void on_sig(int sig) {
    dump(sig == SIGABRT ? "\n[abrt_bt] SIGABRT, native backtrace of the raising thread:\n" : "\n[abrt_bt] SIGSEGV/SIGBUS, native backtrace:\n");
    signal(sig, SIG_DFL);
    raise(sig);
}
void abrt_bt_install(void) {
    signal(SIGABRT, on_sig);
    signal(SIGSEGV, on_sig);
    signal(SIGBUS, on_sig);
}
void abort(void) {
    dump("\n[abrt_bt] abort() called, native backtrace of the caller:\n");
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
    _exit(134);
}
__attribute__((constructor)) static void init(void) {
    const char *f = getenv("ABRT_BT_FILE");
    if (f) {
        int fd = open(f, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) out_fd = fd;
    }
    abrt_bt_install();
}
