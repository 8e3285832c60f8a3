// how fast T threads bring a page-cached file into a buffer: pread (what pa_process_reads does) vs memcpy from a populated mapping vs the same with streaming stores.
// build: g++ -O2 -pthread -o host_read host_read.cpp ; usage: host_read <file> [threads]   (measurement only; nothing in the product uses it)
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void nt_copy(uint8_t* d, const uint8_t* s, size_t n) {   // d 64-byte aligned, n multiple of 64
    for (size_t i = 0; i < n; i += 64) {
        __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 16)), c = _mm_loadu_si128((const __m128i*)(s + i + 32)), e = _mm_loadu_si128((const __m128i*)(s + i + 48));
        _mm_stream_si128((__m128i*)(d + i), a); _mm_stream_si128((__m128i*)(d + i + 16), b); _mm_stream_si128((__m128i*)(d + i + 32), c); _mm_stream_si128((__m128i*)(d + i + 48), e);
    }
    _mm_sfence();
}
int main(int argc, char** argv) {
    const char* path = argv[1];
    const int T = argc > 2 ? atoi(argv[2]) : 16;
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    const size_t size = (size_t)st.st_size / (2 << 20) * (2 << 20);
    const size_t W = 64ull << 20, PIECE = 2ull << 20;
    uint8_t* dst = (uint8_t*)mmap(nullptr, 4 * W, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    const uint8_t* map = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
    const uint8_t* map2 = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);   // never populated as a whole: modes 3 - 5 pay for their pages
    for (int mode = 0; mode < 6; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            std::atomic<size_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t off = next.fetch_add(PIECE);
                        if (off >= size) return;
                        uint8_t* d = dst + (off % (4 * W));
                        if (mode == 0) { size_t p = 0; while (p < PIECE) { ssize_t g = pread(fd, d + p, PIECE - p, (off_t)(off + p)); if (g <= 0) return; p += (size_t)g; } }
                        else if (mode == 1) memcpy(d, map + off, PIECE);
                        else if (mode == 2) nt_copy(d, map + off, PIECE);
                        else {   // 3: populate the piece's pages, copy, drop them again; 4: the same without dropping; 5: plain faults (no populate), drop
                            if (mode != 5 && madvise((void*)(map2 + off), PIECE, 22 /* MADV_POPULATE_READ */) != 0) { perror("madvise"); return; }
                            memcpy(d, map2 + off, PIECE);
                            if (mode != 4) madvise((void*)(map2 + off), PIECE, MADV_DONTNEED);
                        }
                    }
                });
            for (auto& x : th) x.join();
            if (rep) printf("%-28s %2d threads: %6.1f GB/s\n", mode == 0 ? "pread" : mode == 1 ? "memcpy from mapping" : mode == 2 ? "streaming stores from mapping" : mode == 3 ? "populate + memcpy + drop" : mode == 4 ? "populate + memcpy" : "fault + memcpy + drop", T, size / (now() - t0) / 1e9);
        }
    return 0;
}
