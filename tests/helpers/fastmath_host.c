/* Host harness of tests/test_fastmath_host.py: runs mcmc-symreg_amd/csrc/bsr_fastmath.h (the device interpreter's
 * sin / cos / exp, compiled here by gcc) over a file of doubles.
 *   usage: fastmath_host <which: 0 sin, 1 cos, 2 exp> <in.bin> <out.bin>                                          */
#include <stdio.h>
#include <stdlib.h>
#define BSR_TABLE_QUALIFIER static const
#include "bsr_tables.h"
#include "bsr_fastmath.h"

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  const int which = atoi(argv[1]);
  FILE* fi = fopen(argv[2], "rb");
  FILE* fo = fopen(argv[3], "wb");
  if (!fi || !fo) return 3;
  double x;
  while (fread(&x, sizeof x, 1, fi) == 1) {
    const double r = which == 2 ? bsr_exp(x, bsr_tables_src) : bsr_sincos(x, which, bsr_tables_src);
    fwrite(&r, sizeof r, 1, fo);
  }
  fclose(fi);
  fclose(fo);
  return 0;
}
