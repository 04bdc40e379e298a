// Diagnostic only (profiles/probes/capture_defer_probe.py): a SIGSEGV/SIGBUS/SIGABRT handler that
// prints the NATIVE call stack of the faulting thread to stderr (glibc backtrace), then lets the
// default action run.  Python's faulthandler shows Python frames only; the crash of round 2
// (gpurun_out/r2_t30.log) was inside torch/HIP native code under CUDAGraph.capture_end.
//   gcc -O1 -g -shared -fPIC -o segv_bt.so segv_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* uc) {
  (void)uc;
  void* bt[160];
  char head[128];
  int n = snprintf(head, sizeof head, "\n=== native backtrace: signal %d, fault address %p ===\n", sig,
                   si ? si->si_addr : (void*)0);
  if (n > 0) (void)!write(2, head, (size_t)n);
  n = backtrace(bt, 160);
  backtrace_symbols_fd(bt, n, 2);
  const char tail[] = "=== end of native backtrace ===\n";
  (void)!write(2, tail, sizeof tail - 1);
  signal(sig, SIG_DFL);
  raise(sig);
}

void segv_bt_install(void) {
  static char stack[1 << 17];
  stack_t ss;
  memset(&ss, 0, sizeof ss);
  ss.ss_sp = stack;
  ss.ss_size = sizeof stack;
  sigaltstack(&ss, 0);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
  void* warm[4];
  backtrace(warm, 4);      // loads libgcc now, not inside the handler
}
