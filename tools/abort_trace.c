/* LD_PRELOAD helper for one diagnostic run: prints the C call stack of the thread that raises SIGABRT (which library
 * called abort()) to stderr, then lets the default action proceed.  Build: gcc -shared -fPIC -O1 -o abort_trace.so
 * abort_trace.c ; run: LD_PRELOAD=tools/abort_trace.so python -m pytest -p no:faulthandler ...                      */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
  void *frames[64];
  static const char head[] = "\n==== abort_trace: SIGABRT raised on this thread; C stack: ====\n";
  (void)!write(2, head, sizeof head - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  void *warm[4];
  backtrace(warm, 4); /* loads libgcc now, not inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_abort;
  sigaction(SIGABRT, &sa, 0);
}
