/*
 * c_abi_check.c -- a plain-C client of libzernike_hip.so (no Python, no torch in the process): what a
 * non-Python host would link.  Driven by tests/test_gpu_parity.py::test_plain_c_client, which writes
 * the operands to a file, runs this program and compares its output with the oracle.
 *
 * file format (all little-endian): int32 size, n_poly, n_patches, H, W; int32 n[n_poly], m[n_poly];
 *   double basis[n_poly*size*size]; float patches[n_patches*size*size]; float image[H*W]
 * output: double out_patches[n_patches*n_poly]; double out_frame[n_poly*H*W]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "zernike_hip.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != 0) {                                                                   \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, zk_last_error_string());          \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)

static void *slurp(FILE *f, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read\n");
    exit(3);
  }
  return p;
}

int main(int argc, char **argv) {
  if (argc != 3) return 1;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[5];
  if (fread(hdr, sizeof(int32_t), 5, f) != 5) return 3;
  const int size = hdr[0], n_poly = hdr[1], n_patches = hdr[2], H = hdr[3], W = hdr[4];
  int32_t *n = slurp(f, sizeof(int32_t) * n_poly), *m = slurp(f, sizeof(int32_t) * n_poly);
  double *basis = slurp(f, sizeof(double) * n_poly * size * size);
  float *patches = slurp(f, sizeof(float) * (size_t)n_patches * size * size);
  float *image = slurp(f, sizeof(float) * (size_t)H * W);
  fclose(f);

  if (zk_abi_version() != ZK_ABI_VERSION) return 4;
  if (zk_device_count() < 1) {
    fprintf(stderr, "no device\n");
    return 5;
  }
  zk_plan *plan = NULL;
  CHECK(zk_plan_create(size, n_poly, n, m, basis, 0, &plan));
  double *out_p = malloc(sizeof(double) * (size_t)n_patches * n_poly);
  double *out_f = malloc(sizeof(double) * (size_t)n_poly * H * W);
  CHECK(zk_transform_patches(plan, patches, ZK_F32, n_patches, out_p));
  CHECK(zk_transform_frame(plan, image, ZK_F32, H, W, out_f));
  /* argument errors come back as codes, never as crashes */
  if (zk_transform_patches(plan, patches, 42, n_patches, out_p) != ZK_E_BADARG) return 6;
  if (zk_transform_frame(plan, NULL, ZK_F32, H, W, out_f) != ZK_E_BADARG) return 6;
  zk_plan_destroy(plan);

  FILE *o = fopen(argv[2], "wb");
  if (!o) return 1;
  fwrite(out_p, sizeof(double), (size_t)n_patches * n_poly, o);
  fwrite(out_f, sizeof(double), (size_t)n_poly * H * W, o);
  fclose(o);
  printf("c client ok: %d patches, %dx%d frame, disk pixels via ABI\n", n_patches, H, W);
  return 0;
}
