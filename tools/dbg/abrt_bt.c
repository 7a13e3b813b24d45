/* LD_PRELOAD helper (diagnostics only): native backtrace of the thread that raises SIGABRT / SIGSEGV, to stderr.
 * build: gcc -shared -fPIC -o tools/dbg/abrt_bt.so tools/dbg/abrt_bt.c ; run pytest with -p no:faulthandler */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_sig(int sig) {
    void *bt[64];
    const char *m = sig == SIGABRT ? "\n[abrt_bt] SIGABRT, native backtrace of the raising thread:\n" : "\n[abrt_bt] SIGSEGV/SIGBUS, native backtrace:\n";
    write(2, m, strlen(m));
    int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
void abrt_bt_install(void) {
    signal(SIGABRT, on_sig);
    signal(SIGSEGV, on_sig);
    signal(SIGBUS, on_sig);
}
__attribute__((constructor)) static void init(void) { abrt_bt_install(); }
