/* taper_hip_debug.h -- test hooks of libtaper_hip.so.  NOT part of the drop-in boundary (include/taper_hip.h): a host binding never needs
 * them.  The parity tests use them to assert WHICH kernel instance / step form a call took, and the measurement tools to force one. */
#ifndef TAPER_HIP_DEBUG_H
#define TAPER_HIP_DEBUG_H

#include "taper_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* how many th_mlp3_xent / th_mlp2_xent calls this thread has enqueued (which form a Trainer step took) */
int th_debug_mlp3_calls(int64_t *out);
int th_debug_mlp2_calls(int64_t *out);
/* measurement hook: 1 / 2 / 3 = th_mlp2_xent enqueues only that launch (rows / dW1 / finish) on this thread, so that each can be timed with
 * events on its own (the pool hands the same workspace back from call to call: launches 2 and 3 read what an earlier full call left); 0 = the step */
int th_debug_mlp2_only(int which);
/* 1 .. 8 = th_mlp2_xent splits a 16-row block's k chunks over that many workgroups on this thread whatever the cap says (the repro tests
 * walk the hand-off's forms); 0 = the default choice */
int th_debug_mlp2_ksplit(int ksplit);
/* 1 = conv chains over more than 256 images run as min(n, 256) workgroups that WALK the images (r05: bit-identical, measured 3 - 5 % slower
 * than one workgroup per image, off by default); 0 = never; -1 = default */
int th_debug_set_chain_loop(int on);
/* 1: th_conv_chain_mlp3_xent enqueues its first launch (the chain with the classifier's rows) alone -- per-launch timing; 0: both */
int th_debug_chain_mlp3_only(int which);
/* 1 = the compiled chain instances are not used on this thread (their nets take the run-time-described kernel, id 3); 0 = default */
int th_debug_set_chain_generic(int on);
/* launch configuration of the most recent matrix-core 3x3 convolution this thread enqueued (the parity
 * tests assert which kernel instance a shape takes): out6 = {16-channel tiles per workgroup (1/2/4), 1 if the
 * operands are staged by LDS-DMA (2-5: the image-resident kernel; 6: a conv chain, out6[0] = its instance id), waves per workgroup / 4, grid.x, grid.y, 1 if the epilogue is the fused 2x2 pool}. */
int th_debug_last_conv_config(th_ctx *ctx, int *out6);
/* which matrix-core kernel takes a 3x3 launch.  -1 (default): the image-resident kernel (whole images per
 * workgroup, every output tile in registers; out6[1] == 2 in th_debug_last_conv_config, out6[2] = pixel tiles per wave,
 * out6[4] = images per unit) when the launch has at least one unit per two CUs, the 128-pixel kernel otherwise; 0: never;
 * 1: whenever the shape fits it. */
int th_debug_set_conv_img(th_ctx *ctx, int mode);

/* post-mortem of the in-launch exchange (csrc/dp_dev.h) on stderr: the communicator's state words and, per parity and source block of the
 * receive region, the slots that hold words.  Trainer::check_comm calls it under TAPER_DP_POSTMORTEM=1 when a time-out is reported. */
int th_comm_debug_dump(th_comm *comm, th_ctx *ctx);
#ifdef __cplusplus
}
#endif

#endif /* TAPER_HIP_DEBUG_H */
