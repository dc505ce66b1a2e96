// copy_pool_harness.cpp -- CPU check of vppstereo_amd/csrc/copy_pool.h (the frame stream's staging copies): every pool size copies
// several jobs of odd sizes (empty and NULL ones among them) correctly, hundreds of times, with pauses long enough for the workers to
// fall asleep in between.  tests/test_copy_pool_cpu.py builds it twice: plain, and with -fsanitize=thread.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "copy_pool.h"

using vppx_host::CopyJob;
using vppx_host::CopyPool;

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    for (int nt : {0, 1, 3, 7}) {
        CopyPool pool(nt);
        const size_t N = 1555200 + 13, M = 518400 * 4 + 1; // an image and a hint map of a 540 x 960 frame, not multiples of the piece size
        std::vector<unsigned char> s(N), d(N), s2(M), d2(M), tiny_s(5), tiny_d(5);
        for (size_t i = 0; i < N; i++) s[i] = (unsigned char)(i * 7 + nt);
        for (size_t i = 0; i < M; i++) s2[i] = (unsigned char)(i * 13 + 1);
        for (int i = 0; i < 5; i++) tiny_s[i] = (unsigned char)(40 + i);
        for (int it = 0; it < iters; it++) {
            d[0] = d[N - 1] = d2[0] = d2[M - 1] = tiny_d[4] = 0;
            s[it % N] ^= 0x5a; // (sources change between copies)
            CopyJob j[5] = {{d.data(), s.data(), N}, {d2.data(), s2.data(), M}, {nullptr, s.data(), 5}, {tiny_d.data(), tiny_s.data(), 5}, {d.data(), s.data(), 0}};
            pool.copy(j, 5);
            if (memcmp(d.data(), s.data(), N) || memcmp(d2.data(), s2.data(), M) || memcmp(tiny_d.data(), tiny_s.data(), 5)) {
                printf("MISMATCH threads %d iteration %d\n", nt + 1, it);
                return 1;
            }
            if (it % 60 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));
        }
        CopyJob one = {tiny_d.data(), tiny_s.data(), 5};
        pool.copy(&one, 1);   // a single piece: copied by the caller alone
        pool.copy(&one, 0);   // nothing
    }
    printf("COPY_POOL_OK\n");
    return 0;
}
