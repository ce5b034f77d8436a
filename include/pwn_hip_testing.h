/* pwn_hip_testing.h -- entry points that exist for the test suite only.  NOT part of the drop-in boundary (include/pwn_hip.h): a
 * reference-side binding never calls them.  They are exported by the same library so that the tests exercise the product build. */
#ifndef PWN_HIP_TESTING_H
#define PWN_HIP_TESTING_H
#include "pwn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The converter's integral-image kernels hand the running sums of a strip to the strip on its right through tagged words,
 * polled with a bound: a word that never arrives raises a fault flag and the convert call returns PWN_HIP_ERR_LAUNCH ("strip hand-over
 * timed out") instead of hanging the device.  This call makes that happen on purpose: the word (strip, band, chain) of every frame of
 * rows x * images is withheld and the poll bound is lowered to spin_limit polls (0 = the default).  strip < 0 switches the hook off.
 * While the hook is on every convert call of the context fails; the word index is the same in both kernels that hand over
 * (k_unproject_integral, k_unproject_integral_rows: 10 planes x band rows chains per (strip, band)). */
int pwn_hip_debug_withhold_carry(pwn_hip_ctx* ctx, int strip, int band, int chain, int rows, int spin_limit);
/* spin_limit < 0: |spin_limit| polls and ONE disturbed launch only -- the hook switches itself off when that launch has timed out, so
 * that the library's own recovery can be watched: a convert call (and the one-submission step) whose launch timed out is made again, once,
 * before an error is reported (the words are epoch-tagged: a failed launch leaves nothing behind).  Number of such repeats so far: */
int pwn_hip_debug_convert_retries(pwn_hip_ctx* ctx, int* retries);

/* An alignment does not project a cloud where the cloud's own index image (the one DepthImageConverter::compute produced for it) is known
 * to be what that projection returns: same camera matrix, image size and range, identity pose (the current cloud always; the reference cloud
 * in the first outer iteration of an identity guess).  enabled = 0 makes every alignment of the context project everything, so that tests can
 * hold the shortcut against the projection it replaces; 1 (the default) switches it back on. */
int pwn_hip_debug_set_index_shortcut(pwn_hip_ctx* ctx, int enabled);

/* PinholePointProjector::project (pinholepointprojector.cpp:54-63) keeps the nearest point per pixel, ties to the lowest index.  The aligner's
 * projection kernel settles two points of one projection that meet in a pixel with a compare-and-swap loop; a thread that has not settled after
 * `rounds` rounds (default 4096: tens of thousands of points in one pixel) raises the call's fault word, and the library repeats the whole call
 * with a two-pass projection (nearest depth per pixel, then the lowest index among the points that have it) that needs no loop -- the images are
 * the reference's either way.  rounds = 0 makes every collision give up, so that the repeat can be watched; rounds < 0 restores the default.
 * Number of calls repeated so far: */
int pwn_hip_debug_set_settle_guard(pwn_hip_ctx* ctx, int rounds);
int pwn_hip_debug_projection_fallbacks(pwn_hip_ctx* ctx, int* calls);

#ifdef __cplusplus
}
#endif
#endif
