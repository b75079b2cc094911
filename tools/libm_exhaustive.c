// rs_pbrt_amd/csrc/glibc_libm.h — the restatement of glibc's sinf / cosf / logf / log2f / expf / acosf / atanf / atan2f that librspt
// evaluates on the device — compiled for the host and compared with the host's libm (what Rust's f32 methods call):
// every function over all 2^32 floats (sinf / cosf where the restatement applies, |x| < 120), atan2f over 2^30 pairs.
//   g++ -O2 -ffp-contract=off -mfma -o libm_exhaustive -x c++ tools/libm_exhaustive.c -lm -lpthread && ./libm_exhaustive [stride]
// stride 1 (default) = every float: about 4 core-minutes; the CPU test suite runs it with stride 5 (every fifth bit pattern: all exponents,
// all mantissa residues).
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
#define GL_FN static inline
#define GL_FN_COLD static inline
#define GL_TABLE static const
#define GL_F2U(x) f2u(x)
#define GL_U2F(x) u2f(x)
#define GL_D2U(x) d2u(x)
#define GL_U2D(x) u2d(x)
#include "../rs_pbrt_amd/csrc/glibc_libm.h"

enum { N_FN = 8 };
static const char* NAME[N_FN] = {"sinf", "cosf", "logf", "log2f", "expf", "acosf", "atanf", "atan2f"};
typedef struct { uint64_t lo, hi, bad[N_FN]; uint32_t first[N_FN], first_y; } job;
static uint64_t STRIDE = 1;
static int same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }
#define CHECK(k, mine, ref) do { if (!same((mine), (ref))) { if (!j->bad[k]) j->first[k] = v; j->bad[k]++; } } while (0)
static void* run(void* a) {
    job* j = (job*)a;
    uint64_t st = 0x9E3779B97F4A7C15ull * (j->lo + 1);
    for (uint64_t u = j->lo + (STRIDE - j->lo % STRIDE) % STRIDE; u < j->hi; u += STRIDE) {
        const uint32_t v = (uint32_t)u; const float x = u2f(v);
        if (fabsf(x) < 120.0f) { CHECK(0, rspt_sinf(x), sinf(x)); CHECK(1, rspt_cosf(x), cosf(x)); }
        CHECK(2, rspt_logf(x), logf(x)); CHECK(3, rspt_log2f(x), log2f(x)); CHECK(4, rspt_expf(x), expf(x));
        CHECK(5, rspt_acosf(x), acosf(x)); CHECK(6, rspt_atanf(x), atanf(x));
        if (((u / STRIDE) & 3) == 0) {  // 2^30 / stride pairs: half of them magnitudes 2^-20 .. 2^20 with random signs, half raw bit patterns
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const uint32_t p = (uint32_t)st, q = (uint32_t)(st >> 32);
            float yy, xx;
            if ((u / STRIDE) & 4) { yy = u2f(p); xx = u2f(q); }
            else { yy = u2f((p & 0x807fffffu) | (((p >> 23) % 40 + 107) << 23)); xx = u2f((q & 0x807fffffu) | (((q >> 23) % 40 + 107) << 23)); }
            if (!same(rspt_atan2f(yy, xx), atan2f(yy, xx))) { if (!j->bad[7]) { j->first[7] = f2u(xx); j->first_y = f2u(yy); } j->bad[7]++; }
        }
    }
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 1) { STRIDE = strtoull(argv[1], 0, 10); if (!STRIDE) STRIDE = 1; }
    enum { NT = 16 };
    pthread_t th[NT]; job jb[NT];
    for (int i = 0; i < NT; i++) { memset(&jb[i], 0, sizeof jb[i]); jb[i].lo = (uint64_t)i << 28; jb[i].hi = (uint64_t)(i + 1) << 28; pthread_create(&th[i], 0, run, &jb[i]); }
    uint64_t bad[N_FN] = {0};
    for (int i = 0; i < NT; i++) {
        pthread_join(th[i], 0);
        for (int f = 0; f < N_FN; f++) { bad[f] += jb[i].bad[f]; if (jb[i].bad[f]) printf("first %s mismatch at bits %08x\n", NAME[f], jb[i].first[f]); }
    }
    int rc = 0;
    for (int f = 0; f < N_FN; f++) { printf("%s mismatches %llu\n", NAME[f], (unsigned long long)bad[f]); rc |= bad[f] != 0; }
    printf(rc ? "MISMATCH\n" : "all eight functions equal the host libm on every input tested (stride %llu)\n", (unsigned long long)STRIDE);
    return rc;
}
