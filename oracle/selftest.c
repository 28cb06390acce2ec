/* oracle/selftest.c -- the oracle's whole path on one frame read from a raw float32 file, for the sanitizer build
 * (`make -C oracle sanitize`: -fsanitize=address,undefined).  TEST INFRASTRUCTURE ONLY.
 * usage: selftest_san <frame.bin> <cx> <cy> <cz> <solver 0|1> [n_th n_ty n_tz]
 * prints: status n_roi n_cluster n_plane n_corners grid_index iters_a iters_b then the corners. */
#include <stdio.h>
#include <stdlib.h>

#include "ilcc_oracle.h"

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const int32_t n = (int32_t)(bytes / 16);
  float* xyzi = (float*)malloc((size_t)(n > 0 ? n : 1) * 16);
  if (fread(xyzi, 16, (size_t)n, f) != (size_t)n) return 2;
  fclose(f);
  const float click[3] = {(float)atof(argv[2]), (float)atof(argv[3]), (float)atof(argv[4])};
  orc_params p;
  orc_default_params(&p);
  p.solver = atoi(argv[5]);
  if (argc >= 9) {   /* a smaller grid keeps the exhaustive search short under the sanitizers */
    p.n_th = atoi(argv[6]);
    p.n_ty = atoi(argv[7]);
    p.n_tz = atoi(argv[8]);
    p.th_min = -0.5 * (p.n_th - 1) * p.th_step;
  }
  orc_result r;
  float* cb = (float*)malloc((size_t)(n > 0 ? n : 1) * 16);
  float* pc = (float*)malloc((size_t)(n > 0 ? n : 1) * 16);
  orc_extract(xyzi, n, click, &p, &r, cb, pc);
  printf("%d %d %d %d %d %d %d %d\n", r.status, r.n_roi, r.n_cluster, r.n_plane, r.n_corners, r.grid_index, r.iters_a, r.iters_b);
  for (int i = 0; i < r.n_corners; ++i) printf("%.9g %.9g %.9g\n", r.corners[3 * i], r.corners[3 * i + 1], r.corners[3 * i + 2]);
  /* the online caller and the consumer-side solver on the same data */
  orc_result r2;
  uint8_t* cl = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
  orc_chessboard_by_point(xyzi, n, click, &p, 500, &r2, cb, cl);
  printf("online %d %d\n", r2.status, r2.n_plane);
  free(xyzi);
  free(cb);
  free(pc);
  free(cl);
  return 0;
}
