// Page-cache write ceiling of the box: T threads fill one file (pwrite or a shared mapping) or T files, G GiB in 16 MiB pieces.
//   gcc -O2 -pthread tests/tools/pagecache_micro.c -o /tmp/pcm && /tmp/pcm <dir> <threads> <GiB>
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

static int g_fd[64], g_mode, g_threads;
static size_t g_total, g_piece = 16u << 20;
static unsigned char *g_map, *g_src;
static volatile long g_next;

static void *worker(void *arg)
{
    long id = (long)arg;
    for (;;) {
        long i = __sync_fetch_and_add(&g_next, 1);
        size_t off = (size_t)i * g_piece;
        if (off >= g_total) break;
        if (g_mode == 0) { if (pwrite(g_fd[0], g_src, g_piece, off) != (ssize_t)g_piece) perror("pwrite"); }
        else if (g_mode == 1) memcpy(g_map + off, g_src, g_piece);
        else { if (pwrite(g_fd[id], g_src, g_piece, off / g_threads) != (ssize_t)g_piece) perror("pwrite"); }
    }
    return 0;
}

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    g_threads = argc > 2 ? atoi(argv[2]) : 16;
    g_total = (size_t)(argc > 3 ? atoi(argv[3]) : 8) << 30;
    g_src = malloc(g_piece);
    memset(g_src, 7, g_piece);
    const char *names[3] = {"one file, pwrite", "one file, shared mapping", "one file per thread, pwrite"};
    for (g_mode = 0; g_mode < 3; ++g_mode) {
        char path[512];
        int nf = g_mode == 2 ? g_threads : 1;
        for (int f = 0; f < nf; ++f) {
            snprintf(path, sizeof path, "%s/pcm_%d_%d", dir, g_mode, f);
            g_fd[f] = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (g_fd[f] < 0) { perror(path); return 1; }
        }
        if (g_mode == 1) {
            if (fallocate(g_fd[0], 0, 0, g_total)) perror("fallocate");
            g_map = mmap(0, g_total, PROT_READ | PROT_WRITE, MAP_SHARED, g_fd[0], 0);
            if (g_map == MAP_FAILED) { perror("mmap"); return 1; }
        }
        g_next = 0;
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        pthread_t th[64];
        for (long t = 0; t < g_threads; ++t) pthread_create(&th[t], 0, worker, (void *)t);
        for (int t = 0; t < g_threads; ++t) pthread_join(th[t], 0);
        clock_gettime(CLOCK_MONOTONIC, &b);
        double s = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
        printf("%-32s %2d threads: %.2f GB in %.3f s = %.1f GB/s\n", names[g_mode], g_threads, g_total / 1e9, s, g_total / 1e9 / s);
        if (g_mode == 1) munmap(g_map, g_total);
        for (int f = 0; f < nf; ++f) {
            close(g_fd[f]);
            snprintf(path, sizeof path, "%s/pcm_%d_%d", dir, g_mode, f);
            unlink(path);
        }
    }
    return 0;
}
