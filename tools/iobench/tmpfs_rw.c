// tmpfs_rw.c — how fast can N threads pwrite / pread 64 MiB blocks of a NEW file in /dev/shm (the CLI's end-to-end bound)?
// gcc -O2 -pthread -o tmpfs_rw tmpfs_rw.c ; ./tmpfs_rw /dev/shm/x 32
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
typedef struct { int fd; char *buf; size_t n; off_t off; int wr; } job;
static void *run(void *a) { job *j = a; size_t d = 0; while (d < j->n) { ssize_t k = j->wr ? pwrite(j->fd, j->buf + d, j->n - d, j->off + d) : pread(j->fd, j->buf + d, j->n - d, j->off + d); if (k <= 0) break; d += k; } return NULL; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(int argc, char **argv)
{
    const char *path = argv[1];
    const int nblk = argc > 2 ? atoi(argv[2]) : 32;
    const size_t B = 64u << 20;
    char *buf = malloc(B);
    memset(buf, 0x5a, B);
    for (int mode = 0; mode < 3; ++mode)                       // 0: plain, 1: fallocate first, 2: ftruncate first
        for (int T = 1; T <= 32; T *= 2) {
            unlink(path);
            int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
            double t0 = now();
            if (mode == 1) posix_fallocate(fd, 0, (off_t)nblk * B);
            if (mode == 2) ftruncate(fd, (off_t)nblk * B);
            double t1 = now();
            for (int b = 0; b < nblk; ++b) {
                pthread_t th[32]; job jb[32];
                for (int i = 0; i < T; ++i) { jb[i] = (job){fd, buf + i * (B / T), B / T, (off_t)b * B + i * (B / T), 1}; pthread_create(&th[i], NULL, run, &jb[i]); }
                for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
            }
            double t2 = now();
            for (int b = 0; b < nblk; ++b) {
                pthread_t th[32]; job jb[32];
                for (int i = 0; i < T; ++i) { jb[i] = (job){fd, buf + i * (B / T), B / T, (off_t)b * B + i * (B / T), 0}; pthread_create(&th[i], NULL, run, &jb[i]); }
                for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
            }
            double t3 = now();
            printf("mode %d threads %2d: prealloc %.3f s, write %.2f GiB/s, read %.2f GiB/s\n", mode, T, t1 - t0,
                   nblk * 0.0625 / (t2 - t1), nblk * 0.0625 / (t3 - t2));
            close(fd);
        }
    unlink(path);
    return 0;
}
