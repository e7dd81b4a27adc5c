/* One MLP 784-128-10 training step driven through the C ABI alone (include/taper_hip.h): what a host in
 * any language does after binding the library -- the two launches of the fused step
 * (th_linear_fwd_ex, th_mlp_tail), no C++ host, no Python.
 * Plain C11: `gcc -std=c11 cabi_step.c -ltaper_hip`.  Prints the loss of a few steps on a fixed batch. */
#include <stdio.h>
#include <stdlib.h>

#include "../include/taper_hip.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        if ((call) != 0) {                                                 \
            fprintf(stderr, "%s failed: %s\n", #call, th_last_error());    \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static float frand(unsigned *s) {  /* LCG in [0,1) */
    *s = *s * 1664525u + 1013904223u;
    return (float)(*s >> 8) / 16777216.0f;
}

int main(void) {
    enum { B = 64, IN = 784, HID = 128, OUT = 10, N1 = HID * IN + HID, N2 = OUT * HID + OUT };
    th_ctx *ctx = NULL;
    CHECK(th_ctx_create(0, &ctx));
    unsigned seed = 1;
    float *hx = malloc(sizeof(float) * B * IN), *hy = malloc(sizeof(float) * B), *hp1 = malloc(sizeof(float) * N1),
          *hp2 = malloc(sizeof(float) * N2);
    for (int i = 0; i < B * IN; ++i) hx[i] = frand(&seed);
    for (int i = 0; i < B; ++i) hy[i] = (float)(i % OUT);
    for (int i = 0; i < N1; ++i) hp1[i] = i < HID * IN ? (frand(&seed) - 0.5f) * 0.1f : 0.0f;   /* W1 | b1 */
    for (int i = 0; i < N2; ++i) hp2[i] = i < OUT * HID ? (frand(&seed) - 0.5f) * 0.25f : 0.0f;  /* W2 | b2 */

    float *x, *y, *p1, *g1, *m1, *v1, *p2, *g2, *m2, *v2, *h, *loss, *lr;
    int32_t *tick;
    void *q;
#define DMALLOC(ptr, n) CHECK(th_malloc(ctx, sizeof(float) * (n), &q)); ptr = q
    DMALLOC(x, B * IN); DMALLOC(y, B); DMALLOC(p1, N1); DMALLOC(g1, N1); DMALLOC(m1, N1); DMALLOC(v1, N1);
    DMALLOC(p2, N2); DMALLOC(g2, N2); DMALLOC(m2, N2); DMALLOC(v2, N2); DMALLOC(h, B * HID);
    DMALLOC(loss, 4); DMALLOC(lr, 4);
    CHECK(th_malloc(ctx, 16, &q)); tick = q;
    CHECK(th_memcpy_h2d(ctx, x, hx, sizeof(float) * B * IN));
    CHECK(th_memcpy_h2d(ctx, y, hy, sizeof(float) * B));
    CHECK(th_memcpy_h2d(ctx, p1, hp1, sizeof(float) * N1));
    CHECK(th_memcpy_h2d(ctx, p2, hp2, sizeof(float) * N2));
    CHECK(th_fill_f32(ctx, m1, 0.f, N1)); CHECK(th_fill_f32(ctx, v1, 0.f, N1));
    CHECK(th_fill_f32(ctx, m2, 0.f, N2)); CHECK(th_fill_f32(ctx, v2, 0.f, N2));
    CHECK(th_fill_f32(ctx, (float *)tick, 0.f, 4));            /* Adam's t = 0 */
    const float lr_h = 1e-3f;
    CHECK(th_memcpy_h2d(ctx, lr, &lr_h, sizeof lr_h));

    /* optim.rs:99-110 applied by the kernels that complete each gradient */
    th_adam_fuse w1f = {p1, m1, v1, tick, lr, 0.9f, 0.999f, 1e-8f, 1e-4f};
    th_adam_fuse b1f = {p1 + HID * IN, m1 + HID * IN, v1 + HID * IN, tick, lr, 0.9f, 0.999f, 1e-8f, 1e-4f};
    th_adam_slice head[2] = {{g2, OUT * HID, {p2, m2, v2, tick, lr, 0.9f, 0.999f, 1e-8f, 1e-4f}},
                             {g2 + OUT * HID, OUT, {p2 + OUT * HID, m2 + OUT * HID, v2 + OUT * HID, tick, lr, 0.9f, 0.999f, 1e-8f, 1e-4f}}};
    float first = 0.f, last = 0.f;
    for (int step = 0; step < 50; ++step) {
        /* nn.rs:54-60 + ReLU; the same launch applies the head's W2 / b2 updates the PREVIOUS step left behind
         * (every workgroup of its th_mlp_tail read them) and then opens this step: t += 1 (optim.rs:84) */
        CHECK(th_linear_fwd_ex(ctx, x, p1, p1 + HID * IN, h, B, IN, HID, 1, head, step ? 2 : 0, tick));
        /* head + loss (loss.rs:136-195) + every gradient + Adam(W1, b1) in the epilogue */
        CHECK(th_mlp_tail(ctx, x, h, p2, p2 + OUT * HID, y, B, IN, HID, OUT, loss, NULL, g1, g1 + HID * IN, g2, g2 + OUT * HID, NULL, NULL,
                          NULL, 0, NULL, 0, &w1f, &b1f));
        if (step == 0 || step == 49) {
            float l;
            CHECK(th_memcpy_d2h(ctx, &l, loss, sizeof l));
            printf("step %2d: loss = %.4f\n", step, l);
            if (step == 0) first = l; else last = l;
        }
    }
    CHECK(th_adam_slices(ctx, head, 2));   /* the last step's W2 / b2 update: nobody is left to carry it */
    int32_t t = 0;
    CHECK(th_memcpy_d2h(ctx, &t, tick, sizeof t));
    printf("adam t = %d\n%s\n", t, (last < first && t == 50) ? "loss decreased" : "UNEXPECTED");
    CHECK(th_ctx_destroy(ctx));
    free(hx); free(hy); free(hp1); free(hp2);
    return !(last < first && t == 50);
}
