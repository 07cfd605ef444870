/*
 * c_abi_check.c -- a plain-C client of libzernike_hip.so (no Python, no torch in the process): what a
 * non-Python host would link.  Driven by tests/test_gpu_parity.py::test_plain_c_client, which writes
 * the operands to a file, runs this program and compares its output with the oracle.
 *
 * Besides the two host-buffer transforms it walks the torch-free device-resident / multi-GPU flow at world size 1:
 * zk_device_malloc / _copy, an RCCL communicator through the file rendezvous, patch moments and dense row bands written in
 * place into the full result arrays (zk_transform_patches_dev, zk_transform_frame_dev_strided), zk_allgather_rows,
 * zk_comm_join -- and checks that these reproduce the host-buffer results bit for bit; and a uint16 batch through the
 * widening host path.
 *
 * file format (all little-endian): int32 size, n_poly, n_patches, H, W; int32 n[n_poly], m[n_poly];
 *   double basis[n_poly*size*size]; float patches[n_patches*size*size]; float image[H*W]
 * output: double out_patches[n_patches*n_poly]; double out_frame[n_poly*H*W]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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

  /* ---- device-resident, sharded flow at world size 1 (what one rank of an N-GPU job does) ---------------------- */
  {
    const size_t pb = sizeof(float) * (size_t)n_patches * size * size, mb = sizeof(double) * (size_t)n_patches * n_poly;
    const size_t ib = sizeof(float) * (size_t)H * W, fb = sizeof(double) * (size_t)n_poly * H * W;
    void *d_p = NULL, *d_full = NULL, *d_img = NULL, *d_frame = NULL;
    CHECK(zk_device_malloc(0, (int64_t)pb, &d_p));
    CHECK(zk_device_malloc(0, (int64_t)mb, &d_full));
    CHECK(zk_device_malloc(0, (int64_t)ib, &d_img));
    CHECK(zk_device_malloc(0, (int64_t)fb, &d_frame));
    CHECK(zk_device_copy(0, d_p, patches, (int64_t)pb, 1));
    CHECK(zk_device_copy(0, d_img, image, (int64_t)ib, 1));
    char path[64];
    snprintf(path, sizeof path, "/tmp/zk_c_client_%d.id", (int)getpid());
    zk_comm *comm = NULL;
    CHECK(zk_comm_init_file(0, 0, 1, path, 30.0, &comm));
    if (zk_comm_rank(comm) != 0 || zk_comm_world(comm) != 1) return 7;
    /* two chunks of the batch, each gathered after its kernel; NULL = HIP's default stream */
    const int64_t half = n_patches / 2;
    CHECK(zk_transform_patches_dev(plan, d_p, ZK_F32, half, (double *)d_full, NULL));
    CHECK(zk_allgather_rows(comm, (double *)d_full, 1, n_patches, n_poly, n_patches, 0, half, NULL));
    CHECK(zk_transform_patches_dev(plan, (const char *)d_p + (size_t)half * size * size * sizeof(float), ZK_F32, n_patches - half,
                                   (double *)d_full + (size_t)half * n_poly, NULL));
    CHECK(zk_allgather_rows(comm, (double *)d_full, 1, n_patches, n_poly, n_patches, half, n_patches - half, NULL));
    /* three row bands of the frame, in place in the (n_poly, H, W) array */
    for (int b = 0; b < 3; ++b) {
      const int64_t r0 = (int64_t)H * b / 3, r1 = (int64_t)H * (b + 1) / 3;
      CHECK(zk_transform_frame_dev_strided(plan, d_img, ZK_F32, H, W, r0, r1 - r0, (double *)d_frame + (size_t)r0 * W,
                                           (int64_t)H * W, NULL));
      CHECK(zk_allgather_rows(comm, (double *)d_frame, n_poly, H, W, H, r0, r1 - r0, NULL));
    }
    CHECK(zk_comm_join(comm, NULL));
    CHECK(zk_device_synchronize(0));
    double *chk_p = malloc(mb), *chk_f = malloc(fb);
    CHECK(zk_device_copy(0, chk_p, d_full, (int64_t)mb, 2));
    CHECK(zk_device_copy(0, chk_f, d_frame, (int64_t)fb, 2));
    if (memcmp(chk_f, out_f, fb) != 0) return 8;            /* dense: same kernel, same bits */
    for (size_t k = 0; k < (size_t)n_patches * n_poly; ++k) { /* batch: the kernel variant may depend on the count */
      const double d = chk_p[k] - out_p[k];
      if (d > 1e-12 || d < -1e-12) return 9;
    }
    double t = 1.5, all[1];
    CHECK(zk_comm_allgather_host(comm, &t, all, sizeof t));
    if (all[0] != 1.5) return 10;
    CHECK(zk_comm_destroy(comm));
    CHECK(zk_device_free(0, d_p));
    CHECK(zk_device_free(0, d_full));
    CHECK(zk_device_free(0, d_img));
    CHECK(zk_device_free(0, d_frame));
    /* a 16-bit detector batch through the widening host path: integers 0..65535 of the same patches */
    uint16_t *p16 = malloc(sizeof(uint16_t) * (size_t)n_patches * size * size);
    float *pf = malloc(pb);
    for (size_t k = 0; k < (size_t)n_patches * size * size; ++k) {
      p16[k] = (uint16_t)(patches[k] * 65535.0f);
      pf[k] = (float)p16[k];
    }
    double *o16 = malloc(mb), *of = malloc(mb);
    CHECK(zk_transform_patches(plan, p16, ZK_U16, n_patches, o16));
    CHECK(zk_transform_patches(plan, pf, ZK_F32, n_patches, of));
    if (memcmp(o16, of, mb) != 0) return 11;
    if (zk_transform_patches_dev(plan, d_p, ZK_U16, n_patches, (double *)d_full, NULL) != ZK_E_BADARG) return 12;
    free(chk_p), free(chk_f), free(p16), free(pf), free(o16), free(of);
  }
  zk_plan_destroy(plan);

  FILE *o = fopen(argv[2], "wb");
  if (!o) return 1;
  fwrite(out_p, sizeof(double), (size_t)n_patches * n_poly, o);
  fwrite(out_f, sizeof(double), (size_t)n_poly * H * W, o);
  fclose(o);
  printf("c client ok: %d patches, %dx%d frame; device-resident sharded flow at world 1 and uint16 path verified\n", n_patches, H, W);
  return 0;
}
