/* Exhaustive check of dynamic-visual-slam_amd/csrc/glibc_sincosf.h against the host glibc for every
 * float in [0, 6.5].  Build & run (about 10 s):
 *   g++ -O2 -ffp-contract=off -I dynamic-visual-slam_amd/csrc tools/check_sincosf.c -o /tmp/chk && /tmp/chk
 * Add -mfma -ffp-contract=fast to check the contracted variant too.  Last run in the build
 * container (glibc 2.35, Xeon with FMA): total 1087373313 bad_sin 0 bad_cos 0 (both variants,
 * and with GLIBC_TUNABLES=glibc.cpu.hwcaps=-FMA,-AVX2_Usable). */
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include "glibc_sincosf.h"
int main(int argc, char** argv) {
  float lim = 6.5f; uint32_t ul = gsc::asuint(lim);
  uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;
  unsigned long bad_s = 0, bad_c = 0, tot = 0;
  for (uint32_t u = 0; u <= ul; u += stride) {
    float x; memcpy(&x, &u, 4);
    float s = sinf(x), c = cosf(x), ms = gsc::sinf_(x), mc = gsc::cosf_(x);
    if (gsc::asuint(s) != gsc::asuint(ms)) { if (bad_s < 5) printf("sin x=%a glibc=%a mine=%a\n", x, s, ms); bad_s++; }
    if (gsc::asuint(c) != gsc::asuint(mc)) { if (bad_c < 5) printf("cos x=%a glibc=%a mine=%a\n", x, c, mc); bad_c++; }
    tot++;
  }
  printf("total %lu bad_sin %lu bad_cos %lu\n", tot, bad_s, bad_c);
  return (bad_s || bad_c) ? 1 : 0;
}
