// Diagnostic preload (tools/round4/repro_abort.sh): on SIGABRT print the C backtrace of the ABORTING thread (faulthandler
// only shows Python frames, and the round-3 abort comes from a thread without any), then let the default action run.
//   gcc -shared -fPIC -O1 -o abort_bt.so abort_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static struct sigaction prev_abrt;

static void on_abort(int sig, siginfo_t* info, void* uc) {
    void* frames[96];
    char head[128];
    int n = backtrace(frames, 96);
    int len = snprintf(head, sizeof head, "\n[abort_bt] SIGABRT in tid %ld, %d frames:\n", (long)syscall(SYS_gettid), n);
    if (write(2, head, len) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
    if (write(2, "[abort_bt] end\n", 15) < 0) {}
    // chain to whoever was installed before us (faulthandler), else the default action
    if (prev_abrt.sa_flags & SA_SIGINFO) {
        if (prev_abrt.sa_sigaction) prev_abrt.sa_sigaction(sig, info, uc);
    } else if (prev_abrt.sa_handler != SIG_DFL && prev_abrt.sa_handler != SIG_IGN) {
        prev_abrt.sa_handler(sig);
    }
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

// called from Python AFTER faulthandler.enable() so that this handler runs first and chains to faulthandler's
void abort_bt_install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGABRT, &sa, &prev_abrt);
}
