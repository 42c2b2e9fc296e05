// How fast can several threads (and several processes) fill a FRESH /dev/shm file -- the shared host matrix of
// sharded.gather_scores(to="host")?  Variants: memcpy into the mapping (first-touch faults), madvise(MADV_POPULATE_WRITE)
// first, pwrite() instead of stores, one file per writer.  Host only:  g++ -O2 -pthread shm_fill.cpp -o shm_fill && ./shm_fill [GB]
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum Mode { STORE, POPULATE_THEN_STORE, PWRITE, FALLOCATE_THEN_STORE };
static const char* kName[] = {"memcpy into the mapping", "MADV_POPULATE_WRITE, then memcpy", "pwrite", "posix_fallocate, then memcpy"};

// one process: nt threads fill [off, off + len) of the file `path`
static void fill(const char* path, size_t total, size_t off, size_t len, int nt, Mode mode, const char* src, size_t src_len) {
  int fd = open(path, O_RDWR);
  char* base = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (mode == FALLOCATE_THEN_STORE) posix_fallocate(fd, (off_t)off, (off_t)len);
  std::vector<std::thread> th;
  const size_t per = ((len / nt) + 4095) & ~(size_t)4095;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=] {
      const size_t b = off + (size_t)t * per, e = std::min(off + len, b + per);
      if (b >= e) return;
      if (mode == POPULATE_THEN_STORE) {
        const size_t pb = (b + 4095) & ~(size_t)4095, pe = e & ~(size_t)4095;
        if (pe > pb) madvise(base + pb, pe - pb, MADV_POPULATE_WRITE);
      }
      for (size_t o = b; o < e; o += src_len) {
        const size_t n = std::min(src_len, e - o);
        if (mode == PWRITE) { if (pwrite(fd, src, n, (off_t)o) < 0) perror("pwrite"); }
        else memcpy(base + o, src, n);
      }
    });
  for (auto& t : th) t.join();
  munmap(base, total);
  close(fd);
}

int main(int argc, char** argv) {
  const size_t GB = argc > 1 ? (size_t)atol(argv[1]) : 16;
  const size_t total = GB << 30, src_len = (size_t)64 << 20;
  char* src = (char*)malloc(src_len);
  memset(src, 7, src_len);
  printf("filling a fresh %zu GB /dev/shm file; hardware threads %u\n", GB, std::thread::hardware_concurrency());
  for (int procs : {1, 4})
    for (int nt : {4, 16, 32})
      for (Mode mode : {STORE, POPULATE_THEN_STORE, PWRITE, FALLOCATE_THEN_STORE})
        for (int files : {1, 0}) {   // 1: one shared file; 0: one file per process
          if (procs == 1 && files == 0) continue;
          std::vector<std::string> paths;
          for (int p = 0; p < (files ? 1 : procs); ++p) paths.push_back("/dev/shm/plaidhip_ubench_" + std::to_string(getpid()) + "_" + std::to_string(p));
          const size_t per_file = files ? total : total / procs;
          for (auto& pa : paths) { int fd = open(pa.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600); if (ftruncate(fd, (off_t)per_file) != 0) perror("ftruncate"); close(fd); }
          const double t0 = now();
          std::vector<pid_t> kids;
          for (int p = 0; p < procs; ++p) {
            pid_t k = fork();
            if (k == 0) {
              const size_t len = total / procs;
              if (files) fill(paths[0].c_str(), total, (size_t)p * len, len, nt, mode, src, src_len);
              else fill(paths[(size_t)p].c_str(), per_file, 0, len, nt, mode, src, src_len);
              _exit(0);
            }
            kids.push_back(k);
          }
          for (pid_t k : kids) { int st; waitpid(k, &st, 0); }
          const double dt = now() - t0;
          printf("%d process(es) x %2d threads, %-34s %s: %7.2f s  %6.1f GB/s\n", procs, nt, kName[mode], files ? "one file      " : "file per proc.", dt, total / dt * 1e-9);
          fflush(stdout);
          for (auto& pa : paths) unlink(pa.c_str());
        }
  return 0;
}
