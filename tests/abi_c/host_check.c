/* Plain C99 host program against include/fdes_abi.h: what a maintainer of the reference's host code would link.
 * No GPU needed: parameters, .cnf reader, consistency, sub-slicing, and the loud failure of fdes_create without a device
 * (or a working context with one).  Usage: host_check <file.cnf> [image.bin]; prints one line of key=value pairs.
 * With a second argument and a GPU it also runs fdes_build_measurements and writes the float32 image stack. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdes_abi.h"

int main(int argc, char** argv)
{
    fdes_params p;
    fdes_atoms a;
    fdes_ctx* ctx = NULL;
    int rc, ratio, created;
    if (argc < 2) return 2;
    memset(&a, 0, sizeof a);
    if (fdes_abi_version() != FDES_ABI_VERSION) return 3;
    rc = fdes_params_init(&p, 1000);
    if (rc != FDES_OK) return 4;
    rc = fdes_read_cnf(argv[1], &p, &a, FDES_CNF_BUG_COMPATIBLE);
    if (rc != FDES_OK) { fprintf(stderr, "read_cnf: %d\n", rc); return 5; }
    rc = fdes_params_consistent(&p);
    if (rc != FDES_OK) return 6;
    printf("nAt=%d m1=%d m2=%d m3=%d n3=%d lambda=%.6e sigma=%.6e", a.nAt, p.m1, p.m2, p.m3, p.n3, (double)p.lambda, (double)p.sigma);
    ratio = fdes_params_sub_slices(&p);
    printf(" ratio=%d m3_sub=%d", ratio, p.m3);
    created = fdes_create(&ctx, 0);
    printf(" gpu_available=%d create=%d\n", fdes_gpu_available(), created);
    if (fdes_gpu_available() && created != FDES_OK) return 7;
    if (!fdes_gpu_available() && created == FDES_OK) return 8; /* must not pretend to have a device */
    if (ctx && argc > 2) { /* the drop-in call of INTEGRATION.md, section 2 (parameters as read, before sub-slicing) */
        fdes_params q;
        fdes_atoms b;
        float* image;
        size_t n;
        memset(&b, 0, sizeof b);
        if (fdes_params_init(&q, 1000) != FDES_OK || fdes_read_cnf(argv[1], &q, &b, FDES_CNF_BUG_COMPATIBLE) != FDES_OK ||
            fdes_params_consistent(&q) != FDES_OK) return 9;
        n = (size_t)q.n1 * (size_t)q.n2 * (size_t)q.n3;
        image = (float*)malloc(sizeof(float) * n);
        if (!image) return 10;
        rc = fdes_build_measurements(ctx, &q, &b, image, NULL, NULL);
        if (rc != FDES_OK) { fprintf(stderr, "build_measurements: %d %s\n", rc, fdes_last_error(ctx)); return 11; }
        if (fdes_write_binary(argv[2], image, n) != FDES_OK) return 12;
        free(image);
        fdes_atoms_release(&b);
        fdes_params_release(&q);
    }
    if (ctx) fdes_destroy(ctx);
    fdes_atoms_release(&a);
    fdes_params_release(&p);
    return 0;
}
