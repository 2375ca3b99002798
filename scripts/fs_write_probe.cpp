// fs_write_probe.cpp -- how fast can ONE large file be written on this box?  (the 22 GB index.dat of config 3, p = 1,
// moves at ~10 GB/s through 4 pwrite threads; 8 files side by side reach 38 GB/s.)  Buffered pwrite with T threads on
// disjoint ranges, with and without a preceding fallocate, and O_DIRECT.
// Build: g++ -O2 -pthread -o scripts/fs_write_probe scripts/fs_write_probe.cpp ; run: scripts/fs_write_probe <dir> [GiB]
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double run(const std::string &path, uint64_t bytes, int threads, bool direct, bool prealloc, const char *buf, uint64_t buf_bytes)
{
    int flags = O_WRONLY | O_CREAT | O_TRUNC | (direct ? O_DIRECT : 0);
    int fd = open(path.c_str(), flags, 0644);
    if (fd < 0) return -1.0;
    if (prealloc && posix_fallocate(fd, 0, (off_t)bytes) != 0) { close(fd); return -2.0; }
    const uint64_t piece = 8ull << 20;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    bool failed = false;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            // interleaved pieces: thread t writes pieces t, t + T, ... (what a round-robin copy-back would hand it)
            for (uint64_t o = (uint64_t)t * piece; o < bytes; o += (uint64_t)threads * piece) {
                const uint64_t nb = std::min(piece, bytes - o);
                const char *src = buf + (o % (buf_bytes - piece));
                src = (const char *)((uintptr_t)src & ~(uintptr_t)4095);
                for (uint64_t d = 0; d < nb;) {
                    ssize_t w = pwrite(fd, src + d, nb - d, (off_t)(o + d));
                    if (w <= 0) { failed = true; return; }
                    d += (uint64_t)w;
                }
            }
        });
    for (auto &x : th) x.join();
    close(fd);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unlink(path.c_str());
    return failed ? -3.0 : bytes / s / 1e9;
}

#include <sys/mman.h>
// the same pieces through a shared mapping of the pre-sized file: page faults of different threads do not take the inode lock
static double run_mmap(const std::string &path, uint64_t bytes, int threads, bool prealloc, const char *buf, uint64_t buf_bytes)
{
    int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -1.0;
    auto t0 = std::chrono::steady_clock::now();
    if (prealloc ? posix_fallocate(fd, 0, (off_t)bytes) != 0 : ftruncate(fd, (off_t)bytes) != 0) { close(fd); return -2.0; }
    char *m = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { close(fd); return -4.0; }
    const uint64_t piece = 8ull << 20;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            for (uint64_t o = (uint64_t)t * piece; o < bytes; o += (uint64_t)threads * piece) {
                const uint64_t nb = std::min(piece, bytes - o);
                const char *src = buf + (o % (buf_bytes - piece));
                memcpy(m + o, src, nb);
            }
        });
    for (auto &x : th) x.join();
    munmap(m, bytes);
    close(fd);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unlink(path.c_str());
    return bytes / s / 1e9;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const uint64_t bytes = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 8ull) << 30;
    const uint64_t buf_bytes = 1ull << 30;
    char *buf = nullptr;
    if (posix_memalign((void **)&buf, 4096, buf_bytes)) return 1;
    memset(buf, 0x5a, buf_bytes);
    const std::string path = dir + "/fs_write_probe.bin";
    for (int pre = 0; pre < 2; pre++)
        for (int t : {1, 4, 8, 16}) {
            double g = run_mmap(path, bytes, t, pre, buf, buf_bytes);
            printf("mmap     %s threads %2d: %7.2f GB/s\n", pre ? "fallocate " : "ftruncate ", t, g);
            fflush(stdout);
        }
    for (int pre = 0; pre < 2; pre++)
        for (int direct = 0; direct < 2; direct++)
            for (int t : {1, 4, 8, 16, 32}) {
                double g = run(path, bytes, t, direct, pre, buf, buf_bytes);
                printf("%s%s threads %2d: %7.2f GB/s\n", direct ? "O_DIRECT " : "buffered ", pre ? "fallocate " : "          ", t, g);
                fflush(stdout);
            }
    return 0;
}
