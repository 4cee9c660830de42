// tmpfs_mmap.c — a NEW file in /dev/shm written through a shared mapping: ftruncate per block, mmap the block's range,
// T threads memcpy into it (the page faults allocate the pages without the inode's write lock), munmap.  Compare with
// pwrite (tmpfs_rw.c).   gcc -O2 -pthread -o tmpfs_mmap tmpfs_mmap.c ; ./tmpfs_mmap /dev/shm/x 32
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
typedef struct { char *dst; const char *src; size_t n; } job;
static void *run(void *a) { job *j = a; memcpy(j->dst, j->src, j->n); return NULL; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(int argc, char **argv)
{
    const char *path = argv[1];
    const int nblk = argc > 2 ? atoi(argv[2]) : 32;
    const size_t B = (64u << 20) - 4096 * 3 + 123;            // block sizes that are not page multiples, as .shaf blocks are
    char *buf = malloc(B);
    memset(buf, 0x5a, B);
    for (int populate = 0; populate < 2; ++populate)
        for (int T = 1; T <= 32; T *= 2) {
            unlink(path);
            int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
            double t1 = now();
            size_t off = 0;
            for (int b = 0; b < nblk; ++b) {
                if (ftruncate(fd, (off_t)(off + B))) return 1;
                const size_t a0 = off & ~(size_t)4095, len = off + B - a0;
                char *m = mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd, (off_t)a0);
                if (m == MAP_FAILED) { perror("mmap"); return 1; }
                char *dst = m + (off - a0);
                pthread_t th[32]; job jb[32];
                const size_t per = (B / T + 4095) & ~(size_t)4095;
                int nt = 0;
                for (size_t s = 0; s < B; s += per, ++nt) { jb[nt] = (job){dst + s, buf + s, s + per <= B ? per : B - s}; pthread_create(&th[nt], NULL, run, &jb[nt]); }
                for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
                munmap(m, len);
                off += B;
            }
            double t2 = now();
            printf("mmap%s threads %2d: write %.2f GiB/s\n", populate ? "+populate" : "", T, nblk * (B / 1073741824.0) / (t2 - t1));
            close(fd);
        }
    unlink(path);
    return 0;
}
