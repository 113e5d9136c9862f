// Regression guard for the I/O pool (csrc/common.cpp, IoPool::run): a Batch lives on its waiter's stack, so a worker must
// be done with it before the waiter can leave wait_all().  Round 6's fuzz campaign found the worker notifying AFTER it
// had unlocked the Batch: the waiter (woken by another worker) had returned, and pthread_cond_broadcast ran on whatever
// the thread's next frames had put there -- rewritten stack words ("stack smashing detected" in the Writer's record
// thread) or a thread asleep for good (a hang at exit).
//
// This program does what the record thread does, fast: a Batch on the stack, eight small reads, wait_all, return -- and
// then a frame at the same depth filled with ones, checked for a while.  Exit code 0 = nothing rewritten; the test also
// bounds the run time (the other way the old code failed).  Built by tests/test_host.py with g++ against common.cpp.
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

#include "common.h"

using namespace pss;

static __attribute__((noinline)) int one_round(int fd, char *buf)
{
    IoPool::Batch b;
    for (int k = 0; k < 8; ++k) IoPool::get().submit(&b, fd, false, buf + k * 64, 64, k * 64);
    return IoPool::wait_all(&b);
}

static __attribute__((noinline)) int ones_stay_ones()
{
    volatile unsigned long long a[96];
    for (int i = 0; i < 96; ++i) a[i] = ~0ull;
    for (int k = 0; k < 300; ++k)
        for (int i = 0; i < 96; ++i)
            if (a[i] != ~0ull) return 1 + i;
    return 0;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 40000;
    char path[] = "/tmp/pss_iopool_XXXXXX";
    const int fd = mkstemp(path);
    if (fd < 0) return 2;
    const char zeros[512] = {};
    if (write(fd, zeros, sizeof zeros) != (ssize_t)sizeof zeros) return 2;
    (void)unlink(path);
    char buf[512];
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (one_round(fd, buf)) return 3;
        if (const int w = ones_stay_ones()) {
            if (++bad < 4) printf("round %d: word %d of the frame after the Batch was rewritten\n", r, w - 1);
        }
    }
    printf("%d rounds, %d with the stack rewritten\n", rounds, bad);
    return bad ? 1 : 0;
}
