// Debugging aid: LD_PRELOAD this to print a native backtrace on SIGSEGV (run pytest with -p no:faulthandler).
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <ucontext.h>

static void on_segv(int sig, siginfo_t* si, void* uc_) {
  void* bt[64];
  ucontext_t* uc = (ucontext_t*)uc_;
  char buf[128];
  int n = snprintf(buf, sizeof buf, "\n== SIGSEGV at address %p, rip %p ==\n", si->si_addr,
                   (void*)uc->uc_mcontext.gregs[REG_RIP]);
  write(2, buf, n);
  int k = backtrace(bt, 64);
  backtrace_symbols_fd(bt, k, 2);
  FILE* f = fopen("/proc/self/maps", "r");
  if (f) {
    char line[512];
    while (fgets(line, sizeof line, f))
      if (strstr(line, "r-xp") && (strstr(line, "hip") || strstr(line, "hsa") || strstr(line, "mmlrec"))) write(2, line, strlen(line));
    fclose(f);
  }
  signal(SIGSEGV, SIG_DFL);
  raise(SIGSEGV);
}

__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, NULL);
}
