// Probe only (not product code): prints the native call stack on SIGSEGV, so a crash inside a HIP runtime call can be placed.
// Build: g++ -O1 -g -shared -fPIC -o benchmarks/probes/libsegvbt.so benchmarks/probes/segv_backtrace.cpp ; load with ctypes.CDLL.
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig) {
  void* frames[96];
  int n = backtrace(frames, 96);
  const char msg[] = "\n==== native backtrace (SIGSEGV) ====\n";
  write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

extern "C" void install_segv_backtrace(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = handler;
  sigaction(SIGSEGV, &sa, nullptr);
}

// ---- all-thread native stack dump after N seconds (hang diagnosis without a debugger) ----
#include <dirent.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/syscall.h>

static void dump_handler(int) {
  void* frames[64];
  int n = backtrace(frames, 64);
  char head[64];
  int len = snprintf(head, sizeof(head), "\n==== thread %ld ====\n", (long)syscall(SYS_gettid));
  if (write(2, head, len) < 0) return;
  backtrace_symbols_fd(frames, n, 2);
}

static void* watchdog(void* arg) {
  sleep((unsigned)(long)arg);
  pid_t pid = getpid(), self = (pid_t)syscall(SYS_gettid);
  DIR* d = opendir("/proc/self/task");
  if (d) {
    struct dirent* e;
    while ((e = readdir(d))) {
      pid_t tid = (pid_t)atoi(e->d_name);
      if (tid > 0 && tid != self) {
        syscall(SYS_tgkill, pid, tid, SIGUSR2);
        usleep(100000);
      }
    }
    closedir(d);
  }
  sleep(1);
  _exit(3);
  return nullptr;
}

extern "C" void install_all_thread_dump(int seconds) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = dump_handler;
  sigaction(SIGUSR2, &sa, nullptr);
  pthread_t t;
  pthread_create(&t, nullptr, watchdog, (void*)(long)seconds);
  pthread_detach(t);
}
